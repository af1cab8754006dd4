// Host-side set-up for the MI355X constrained-assembly backend: the data the
// kernels read.  Restates (not copies) the reference's one-off builders on
// flat arrays:
//   cpp/MultiPointConstraint.h:36-126   -> mpcx_mpc_finalize
//   cpp/mpc_helpers.h:19-94             -> mpcx_cell_to_slaves
//   cpp/utils.h:381-496 (+finalize)     -> mpcx_pattern_*
// plus the row-block plan used by the LDS-privatised matrix kernel.
#include "mpcx.h"
#include "mpcx_internal.h"

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

namespace
{
thread_local std::string g_error;
}

void mpcx_set_error(const std::string& msg) { g_error = msg; }

extern "C" const char* mpcx_last_error(void) { return g_error.c_str(); }
extern "C" int mpcx_version(void) { return MPCX_VERSION; }

// ---------------------------------------------------------------------------
extern "C" int mpcx_mpc_finalize(int32_t num_dofs, int32_t num_owned_dofs, int32_t num_slaves,
                                 const int32_t* slaves, const int64_t* masters,
                                 const double* coeffs, const int32_t* owners,
                                 const int32_t* offsets, int8_t* is_slave,
                                 int32_t* sorted_slaves, int32_t* num_local_slaves,
                                 int32_t* masters_offsets, int32_t* masters_out,
                                 double* coeffs_out, int32_t* owners_out)
{
  // slave marker over all local dofs
  std::fill_n(is_slave, num_dofs, int8_t(0));
  for (int32_t i = 0; i < num_slaves; ++i)
  {
    if (slaves[i] < 0 || slaves[i] >= num_dofs)
    {
      mpcx_set_error("mpcx_mpc_finalize: slave index out of range");
      return -1;
    }
    is_slave[slaves[i]] = 1;
  }
  // adjacency with one node per local dof; only slaves have links
  std::vector<int32_t> num_masters(num_dofs, 0);
  for (int32_t i = 0; i < num_slaves; ++i)
    num_masters[slaves[i]] = offsets[i + 1] - offsets[i];
  masters_offsets[0] = 0;
  for (int32_t d = 0; d < num_dofs; ++d)
    masters_offsets[d + 1] = masters_offsets[d] + num_masters[d];
  std::fill(num_masters.begin(), num_masters.end(), 0);
  for (int32_t i = 0; i < num_slaves; ++i)
  {
    const int32_t s = slaves[i];
    for (int32_t j = offsets[i]; j < offsets[i + 1]; ++j)
    {
      const int32_t pos = masters_offsets[s] + num_masters[s]++;
      if (masters[j] < 0 || masters[j] >= num_dofs)
      {
        mpcx_set_error("mpcx_mpc_finalize: master index out of range (single-process backend: "
                       "global master index must equal a local dof)");
        return -2;
      }
      masters_out[pos] = static_cast<int32_t>(masters[j]);
      coeffs_out[pos] = coeffs[j];
      owners_out[pos] = owners[j];
    }
  }
  // sorted slave list and the number of owned slaves
  int32_t c = 0, nloc = 0;
  for (int32_t d = 0; d < num_dofs; ++d)
    if (is_slave[d])
    {
      sorted_slaves[c++] = d;
      if (d < num_owned_dofs)
        ++nloc;
    }
  *num_local_slaves = nloc;
  return 0;
}

// ---------------------------------------------------------------------------
extern "C" int64_t mpcx_cell_to_slaves(int64_t num_cells, int32_t nd, int32_t bs,
                                       const int32_t* dofmap, const int8_t* is_slave,
                                       int32_t* c2s_offsets, int32_t* c2s)
{
  // The reference inverts a dof->cells map whose nodes are visited in
  // ascending dof order, so each cell's slaves come out sorted by dof.
  int64_t total = 0;
  int32_t tmp[256];
  if (nd * bs > 256)
  {
    mpcx_set_error("mpcx_cell_to_slaves: element too large");
    return -1;
  }
  c2s_offsets[0] = 0;
  for (int64_t c = 0; c < num_cells; ++c)
  {
    int n = 0;
    const int32_t* dofs = dofmap + c * nd;
    for (int i = 0; i < nd; ++i)
      for (int k = 0; k < bs; ++k)
      {
        const int32_t d = dofs[i] * bs + k;
        if (is_slave[d])
          tmp[n++] = d;
      }
    if (c2s && n)
    {
      std::sort(tmp, tmp + n);
      std::copy(tmp, tmp + n, c2s + total);
    }
    total += n;
    if (total > INT32_MAX)
    {
      mpcx_set_error("mpcx_cell_to_slaves: more than 2^31 links");
      return -2;
    }
    c2s_offsets[c + 1] = static_cast<int32_t>(total);
  }
  return total;
}

// ---------------------------------------------------------------------------
namespace
{
struct Pattern
{
  int32_t nrows = 0;
  std::vector<int64_t> rowptr;
  std::vector<int32_t> cols;
};

template <typename F>
void parallel_ranges(int64_t n, int nthreads, F&& f)
{
  nthreads = std::max(1, nthreads);
  if (nthreads == 1 || n < 4096)
  {
    f(0, int64_t(0), n);
    return;
  }
  std::vector<std::thread> pool;
  for (int t = 0; t < nthreads; ++t)
  {
    const int64_t lo = n * t / nthreads, hi = n * (t + 1) / nthreads;
    pool.emplace_back([&f, t, lo, hi]() { f(t, lo, hi); });
  }
  for (auto& th : pool)
    th.join();
}
} // namespace

extern "C" void* mpcx_pattern_build(int64_t num_cells, const int32_t* dofmap0, int32_t nd0,
                                    int32_t bs0, int32_t num_blocks0, const int32_t* dofmap1,
                                    int32_t nd1, int32_t bs1, int32_t num_blocks1,
                                    const int32_t* c2s_offsets0, const int32_t* c2s0,
                                    const int32_t* masters_offsets0, const int32_t* masters0,
                                    const int32_t* c2s_offsets1, const int32_t* c2s1,
                                    const int32_t* masters_offsets1, const int32_t* masters1,
                                    int32_t num_threads)
{
  (void)num_blocks1;
  // 1. row-block -> cells adjacency: every cell for each of its row blocks,
  //    plus, for cells holding row slaves, every master block of those slaves
  //    (the `flattened_masters` rows of cpp/utils.h:484-488).
  std::vector<int64_t> adj_off(size_t(num_blocks0) + 1, 0);
  for (int64_t c = 0; c < num_cells; ++c)
  {
    const int32_t* d = dofmap0 + c * nd0;
    for (int i = 0; i < nd0; ++i)
      ++adj_off[d[i] + 1];
    for (int32_t p = c2s_offsets0[c]; p < c2s_offsets0[c + 1]; ++p)
    {
      const int32_t s = c2s0[p];
      for (int32_t q = masters_offsets0[s]; q < masters_offsets0[s + 1]; ++q)
        ++adj_off[masters0[q] / bs0 + 1];
    }
  }
  for (int32_t r = 0; r < num_blocks0; ++r)
    adj_off[r + 1] += adj_off[r];
  const int64_t nadj = adj_off[num_blocks0];
  if (nadj > INT32_MAX * int64_t(4))
  {
    mpcx_set_error("mpcx_pattern_build: adjacency too large");
    return nullptr;
  }
  std::vector<int32_t> adj(nadj);
  {
    std::vector<int64_t> cur(adj_off.begin(), adj_off.end() - 1);
    for (int64_t c = 0; c < num_cells; ++c)
    {
      const int32_t* d = dofmap0 + c * nd0;
      for (int i = 0; i < nd0; ++i)
        adj[cur[d[i]]++] = static_cast<int32_t>(c);
      for (int32_t p = c2s_offsets0[c]; p < c2s_offsets0[c + 1]; ++p)
      {
        const int32_t s = c2s0[p];
        for (int32_t q = masters_offsets0[s]; q < masters_offsets0[s + 1]; ++q)
          adj[cur[masters0[q] / bs0]++] = static_cast<int32_t>(c);
      }
    }
  }

  // 2. per row block: union of the column sets of its cells
  const int nt = std::max(1, num_threads);
  std::vector<std::vector<int32_t>> tcols(nt);
  std::vector<int32_t> row_count(size_t(num_blocks0), 0);
  std::vector<int64_t> tlo(nt + 1, 0);
  parallel_ranges(num_blocks0, nt,
                  [&](int t, int64_t lo, int64_t hi)
                  {
                    tlo[t] = lo;
                    std::vector<int32_t>& out = tcols[t];
                    std::vector<int32_t> buf;
                    for (int64_t r = lo; r < hi; ++r)
                    {
                      buf.clear();
                      for (int64_t a = adj_off[r]; a < adj_off[r + 1]; ++a)
                      {
                        const int64_t c = adj[a];
                        const int32_t* d = dofmap1 + c * nd1;
                        buf.insert(buf.end(), d, d + nd1);
                        for (int32_t p = c2s_offsets1[c]; p < c2s_offsets1[c + 1]; ++p)
                        {
                          const int32_t s = c2s1[p];
                          for (int32_t q = masters_offsets1[s]; q < masters_offsets1[s + 1]; ++q)
                            buf.push_back(masters1[q] / bs1);
                        }
                      }
                      std::sort(buf.begin(), buf.end());
                      buf.erase(std::unique(buf.begin(), buf.end()), buf.end());
                      row_count[r] = static_cast<int32_t>(buf.size());
                      out.insert(out.end(), buf.begin(), buf.end());
                    }
                  });
  adj.clear();
  adj.shrink_to_fit();

  // 3. expand blocks to a scalar CSR
  auto* P = new Pattern;
  const int64_t nrows = int64_t(num_blocks0) * bs0;
  if (nrows > INT32_MAX)
  {
    mpcx_set_error("mpcx_pattern_build: too many rows");
    delete P;
    return nullptr;
  }
  P->nrows = static_cast<int32_t>(nrows);
  P->rowptr.resize(nrows + 1);
  int64_t nnz = 0;
  P->rowptr[0] = 0;
  for (int32_t r = 0; r < num_blocks0; ++r)
    for (int k = 0; k < bs0; ++k)
    {
      nnz += int64_t(row_count[r]) * bs1;
      P->rowptr[size_t(r) * bs0 + k + 1] = nnz;
    }
  P->cols.resize(nnz);
  // block-row start inside each thread's buffer
  std::vector<int64_t> blk_start(size_t(num_blocks0) + 1, 0);
  for (int32_t r = 0; r < num_blocks0; ++r)
    blk_start[r + 1] = blk_start[r] + row_count[r];
  // thread t's buffer starts at block row tlo[t]
  parallel_ranges(num_blocks0, nt,
                  [&](int t, int64_t lo, int64_t hi)
                  {
                    const std::vector<int32_t>& in = tcols[t];
                    const int64_t base = blk_start[lo];
                    for (int64_t r = lo; r < hi; ++r)
                    {
                      const int32_t* cb = in.data() + (blk_start[r] - base);
                      const int32_t n = row_count[r];
                      for (int k = 0; k < bs0; ++k)
                      {
                        int32_t* dst = P->cols.data() + P->rowptr[r * bs0 + k];
                        for (int32_t j = 0; j < n; ++j)
                          for (int l = 0; l < bs1; ++l)
                            *dst++ = cb[j] * bs1 + l;
                      }
                    }
                  });
  return P;
}

extern "C" int64_t mpcx_pattern_nnz(void* p) { return static_cast<Pattern*>(p)->cols.size(); }
extern "C" int32_t mpcx_pattern_nrows(void* p) { return static_cast<Pattern*>(p)->nrows; }
extern "C" int mpcx_pattern_copy(void* p, mpcx_nnz_t* rowptr, int32_t* cols)
{
  auto* P = static_cast<Pattern*>(p);
  std::memcpy(rowptr, P->rowptr.data(), P->rowptr.size() * sizeof(mpcx_nnz_t));
  std::memcpy(cols, P->cols.data(), P->cols.size() * sizeof(int32_t));
  return 0;
}
extern "C" void mpcx_pattern_free(void* p) { delete static_cast<Pattern*>(p); }

// ---------------------------------------------------------------------------
namespace
{
struct RowBlockPlan
{
  std::vector<int32_t> block_row0;
  std::vector<int64_t> block_ent_off;
  std::vector<int32_t> block_ents;
};
} // namespace

namespace
{
// greedy contiguous partition of the rows, boundaries on dof-block (bs0) multiples; a block is cut at the last hinted
// row (tile start of the numbering) inside its capacity window so that blocks do not straddle tiles.
// rowptr == NULL: one entry per row (the vector plans).  Returns false when a single dof block exceeds the capacity.
bool greedy_block_ranges(int32_t nrows, const mpcx_nnz_t* rowptr, int32_t max_rows, int32_t max_nnz, int32_t bs0,
                         const int32_t* row_hints, int32_t n_hints, std::vector<int32_t>& block_row0)
{
  block_row0.clear();
  block_row0.push_back(0);
  int32_t r0 = 0;
  int32_t h = 0; // first hint > r0
  while (r0 < nrows)
  {
    int32_t r1 = r0;
    if (!rowptr)
    {
      const int32_t cap = std::min(max_rows, max_nnz);
      r1 = std::min(nrows, r0 + (cap / bs0) * bs0);
      if (nrows - r0 <= cap)
        r1 = nrows;
    }
    else
      while (r1 < nrows)
      {
        const int32_t rn = std::min(r1 + bs0, nrows);
        if (rn - r0 > max_rows || rowptr[rn] - rowptr[r0] > max_nnz)
          break;
        r1 = rn;
      }
    if (r1 == r0)
      return false;
    if (row_hints && r1 < nrows)
    {
      while (h < n_hints && row_hints[h] <= r0)
        ++h;
      int32_t cut = -1;
      for (int32_t k = h; k < n_hints && row_hints[k] <= r1; ++k)
        cut = row_hints[k];
      if (cut > r0 && cut % bs0 == 0)
        r1 = cut;
    }
    block_row0.push_back(r1);
    r0 = r1;
  }
  return true;
}
} // namespace

extern "C" int64_t mpcx_block_ranges(int32_t nrows, const mpcx_nnz_t* rowptr, int32_t max_rows, int32_t max_nnz, int32_t bs,
                                     const int32_t* row_hints, int32_t n_hints, int32_t* block_row0, int64_t capacity)
{
  if (nrows < 0 || max_rows <= 0 || max_nnz <= 0 || bs <= 0)
  {
    mpcx_set_error("mpcx_block_ranges: invalid arguments");
    return -1;
  }
  std::vector<int32_t> r;
  if (!greedy_block_ranges(nrows, rowptr, max_rows, max_nnz, bs, row_hints, n_hints, r))
  {
    mpcx_set_error("mpcx_block_ranges: a single dof block exceeds the block capacity");
    return -1;
  }
  if (block_row0 && int64_t(r.size()) <= capacity)
    std::memcpy(block_row0, r.data(), r.size() * sizeof(int32_t));
  return int64_t(r.size()) - 1;
}

extern "C" void* mpcx_rowblock_plan_build(int32_t nrows, const mpcx_nnz_t* rowptr, int32_t max_rows,
                                          int32_t max_nnz, int64_t n_entities, int32_t estride,
                                          const int32_t* entities0, const int32_t* dofmap0,
                                          int32_t nd0, int32_t bs0, const int32_t* row_hints,
                                          int32_t n_hints, int32_t num_threads)
{
  (void)num_threads;
  auto* P = new RowBlockPlan;
  if (!greedy_block_ranges(nrows, rowptr, max_rows, max_nnz, bs0, row_hints, n_hints, P->block_row0))
  {
    mpcx_set_error("mpcx_rowblock_plan_build: a single dof block exceeds the block capacity");
    delete P;
    return nullptr;
  }
  const int32_t nb = static_cast<int32_t>(P->block_row0.size()) - 1;
  // dof block -> row block
  const int32_t ndb = nrows / bs0;
  std::vector<int32_t> blk_of(ndb);
  for (int32_t b = 0; b < nb; ++b)
    for (int32_t r = P->block_row0[b] / bs0; r < P->block_row0[b + 1] / bs0; ++r)
      blk_of[r] = b;
  // count, then fill (entities ascending inside each block)
  P->block_ent_off.assign(size_t(nb) + 1, 0);
  int32_t seen[64];
  if (nd0 > 64)
  {
    mpcx_set_error("mpcx_rowblock_plan_build: element too large");
    delete P;
    return nullptr;
  }
  for (int pass = 0; pass < 2; ++pass)
  {
    std::vector<int64_t> cur;
    if (pass == 1)
    {
      for (int32_t b = 0; b < nb; ++b)
        P->block_ent_off[b + 1] += P->block_ent_off[b];
      P->block_ents.resize(P->block_ent_off[nb]);
      cur.assign(P->block_ent_off.begin(), P->block_ent_off.end() - 1);
    }
    for (int64_t e = 0; e < n_entities; ++e)
    {
      const int64_t c = entities0[e * estride];
      const int32_t* d = dofmap0 + c * nd0;
      int ns = 0;
      for (int i = 0; i < nd0; ++i)
      {
        const int32_t b = blk_of[d[i]];
        bool dup = false;
        for (int k = 0; k < ns; ++k)
          dup |= (seen[k] == b);
        if (!dup)
        {
          seen[ns++] = b;
          if (pass == 0)
            ++P->block_ent_off[b + 1];
          else
            P->block_ents[cur[b]++] = static_cast<int32_t>(e);
        }
      }
    }
  }
  return P;
}

extern "C" int32_t mpcx_rowblock_plan_num_blocks(void* p)
{
  return static_cast<int32_t>(static_cast<RowBlockPlan*>(p)->block_row0.size()) - 1;
}
extern "C" int64_t mpcx_rowblock_plan_num_ents(void* p)
{
  return static_cast<int64_t>(static_cast<RowBlockPlan*>(p)->block_ents.size());
}
extern "C" int mpcx_rowblock_plan_copy(void* p, int32_t* block_row0, int64_t* block_ent_off,
                                       int32_t* block_ents)
{
  auto* P = static_cast<RowBlockPlan*>(p);
  std::memcpy(block_row0, P->block_row0.data(), P->block_row0.size() * sizeof(int32_t));
  std::memcpy(block_ent_off, P->block_ent_off.data(), P->block_ent_off.size() * sizeof(int64_t));
  if (block_ents && !P->block_ents.empty())
    std::memcpy(block_ents, P->block_ents.data(), P->block_ents.size() * sizeof(int32_t));
  return 0;
}
extern "C" void mpcx_rowblock_plan_free(void* p) { delete static_cast<RowBlockPlan*>(p); }

// ---------------------------------------------------------------------------
// Scatter plan of the master contributions of slave entities: the index logic of modify_mpc_cell
// (cpp/assemble_matrix.cpp:182-267) evaluated once on the host -- which entry (p, q) of the element
// tensor goes, times which coefficient, to which position of the CSR values -- so that the device
// kernel is a list of multiply-adds instead of chains of dependent CSR searches.
namespace
{
struct MpcPlan
{
  // gathered by target: target k (position tgt_pos[k] of vals) sums coef * Ae_ent[pq] over its tuples
  std::vector<int64_t> tgt_pos; // [n_targets] distinct positions, ascending
  std::vector<int64_t> off;     // [n_targets + 1] into the tuple arrays
  std::vector<int32_t> ent;     // entity (index into the integral's entity list) of every tuple
  std::vector<int32_t> pq;      // p * N1 + q
  std::vector<double> coef;
  // scratch while building
  std::vector<int64_t> pos;
};
inline int64_t host_csr_find(const int32_t* cols, int64_t lo, int64_t hi, int col)
{
  const int32_t* b = cols + lo;
  const int32_t* e = cols + hi;
  const int32_t* it = std::lower_bound(b, e, col);
  return (it != e && *it == col) ? int64_t(it - cols) : int64_t(-1);
}
} // namespace

extern "C" void* mpcx_mpc_plan_build(int64_t n_slave_entities, const int32_t* slave_entities, int32_t estride,
                                     const int32_t* entities0, const int32_t* entities1, const int32_t* dofmap0,
                                     int32_t nd0, int32_t bs0, const int32_t* dofmap1, int32_t nd1, int32_t bs1,
                                     const int8_t* bc0, const int8_t* bc1, const int8_t* is_slave0,
                                     const int32_t* m_off0, const int32_t* masters0, const double* coeffs0,
                                     const int8_t* is_slave1, const int32_t* m_off1, const int32_t* masters1,
                                     const double* coeffs1, const mpcx_nnz_t* rowptr, const int32_t* cols)
{
  auto* P = new MpcPlan;
  const int N0 = nd0 * bs0, N1 = nd1 * bs1;
  std::vector<int32_t> rows(N0), colsd(N1);
  std::vector<char> rbc(N0), cbc(N1), rsl(N0), csl(N1);
  for (int64_t t = 0; t < n_slave_entities; ++t)
  {
    const int64_t e = slave_entities[t];
    const int64_t cell0 = entities0 ? entities0[e * estride] : e;
    const int64_t cell1 = entities1 ? entities1[e * estride] : e;
    for (int i = 0; i < nd0; ++i)
      for (int k = 0; k < bs0; ++k)
      {
        const int32_t r = dofmap0[cell0 * nd0 + i] * bs0 + k;
        rows[i * bs0 + k] = r;
        rbc[i * bs0 + k] = bc0 && bc0[r];
        rsl[i * bs0 + k] = is_slave0[r];
      }
    for (int j = 0; j < nd1; ++j)
      for (int k = 0; k < bs1; ++k)
      {
        const int32_t c = dofmap1[cell1 * nd1 + j] * bs1 + k;
        colsd[j * bs1 + k] = c;
        cbc[j * bs1 + k] = bc1 && bc1[c];
        csl[j * bs1 + k] = is_slave1[c];
      }
    auto emit = [&](int p, int q, int64_t pos, double c)
    {
      if (pos < 0 || rbc[p] || cbc[q]) // Dirichlet rows/cols of the element tensor are zero (:510-533)
        return;
      P->ent.push_back(int32_t(e));
      P->pq.push_back(p * N1 + q);
      P->pos.push_back(pos);
      P->coef.push_back(c);
    };
    // row masters (:214-246)
    for (int p = 0; p < N0; ++p)
    {
      if (!rsl[p])
        continue;
      for (int mi = m_off0[rows[p]]; mi < m_off0[rows[p] + 1]; ++mi)
      {
        const int32_t m = masters0[mi];
        const double ci = coeffs0[mi];
        const int64_t lo = rowptr[m], hi = rowptr[m + 1];
        for (int q = 0; q < N1; ++q)
        {
          if (csl[q])
          {
            // master-master term from the un-stripped tensor (:239-245)
            for (int mj = m_off1[colsd[q]]; mj < m_off1[colsd[q] + 1]; ++mj)
              emit(p, q, host_csr_find(cols, lo, hi, masters1[mj]), ci * coeffs1[mj]);
          }
          else
            emit(p, q, host_csr_find(cols, lo, hi, colsd[q]), ci); // stripped row (:226-236)
        }
      }
    }
    // column masters (:251-267)
    for (int q = 0; q < N1; ++q)
    {
      if (!csl[q])
        continue;
      for (int mj = m_off1[colsd[q]]; mj < m_off1[colsd[q] + 1]; ++mj)
      {
        const int32_t m = masters1[mj];
        const double cj = coeffs1[mj];
        for (int p = 0; p < N0; ++p)
        {
          if (rsl[p])
            continue;
          emit(p, q, host_csr_find(cols, rowptr[rows[p]], rowptr[rows[p] + 1], m), cj);
        }
      }
    }
  }
  // group the tuples by target position (stable: entity order inside a target is kept)
  const size_t n = P->pos.size();
  std::vector<int64_t> order(n);
  std::iota(order.begin(), order.end(), int64_t(0));
  std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return P->pos[x] < P->pos[y]; });
  std::vector<int32_t> ent(n), pq(n);
  std::vector<double> coef(n);
  P->off.push_back(0);
  for (size_t k = 0; k < n; ++k)
  {
    const int64_t src = order[k];
    ent[k] = P->ent[src];
    pq[k] = P->pq[src];
    coef[k] = P->coef[src];
    if (k == 0 || P->pos[src] != P->tgt_pos.back())
    {
      if (k > 0)
        P->off.push_back(int64_t(k));
      P->tgt_pos.push_back(P->pos[src]);
    }
  }
  if (n > 0)
    P->off.push_back(int64_t(n));
  P->ent.swap(ent);
  P->pq.swap(pq);
  P->coef.swap(coef);
  P->pos.clear();
  P->pos.shrink_to_fit();
  return P;
}

extern "C" int64_t mpcx_mpc_plan_size(void* plan) { return int64_t(static_cast<MpcPlan*>(plan)->pq.size()); }
extern "C" int64_t mpcx_mpc_plan_num_targets(void* plan) { return int64_t(static_cast<MpcPlan*>(plan)->tgt_pos.size()); }
extern "C" int mpcx_mpc_plan_copy(void* plan, mpcx_nnz_t* tgt_pos, int64_t* off, int32_t* ent, int32_t* pq, double* coef)
{
  auto* P = static_cast<MpcPlan*>(plan);
  std::memcpy(tgt_pos, P->tgt_pos.data(), P->tgt_pos.size() * sizeof(mpcx_nnz_t));
  std::memcpy(off, P->off.data(), P->off.size() * sizeof(int64_t));
  std::memcpy(ent, P->ent.data(), P->ent.size() * sizeof(int32_t));
  std::memcpy(pq, P->pq.data(), P->pq.size() * sizeof(int32_t));
  std::memcpy(coef, P->coef.data(), P->coef.size() * sizeof(double));
  return 0;
}
extern "C" void mpcx_mpc_plan_free(void* plan) { delete static_cast<MpcPlan*>(plan); }

// ---------------------------------------------------------------------------
// Dictionary compression of the scatter-offset table: distinct rows -> ids.
extern "C" int32_t mpcx_compress_offsets(const uint8_t* rows, int64_t n, int32_t noff, int32_t max_patterns,
                                         uint16_t* pattern_ids, uint8_t* table)
{
  if (max_patterns > 65536)
    max_patterns = 65536;
  // open-addressing hash set over the rows already copied into `table`
  constexpr int LOG2 = 18;
  std::vector<int32_t> slots(size_t(1) << LOG2, -1);
  int32_t npat = 0;
  for (int64_t e = 0; e < n; ++e)
  {
    const uint8_t* r = rows + e * noff;
    uint64_t h = 1469598103934665603ull; // FNV-1a
    for (int k = 0; k < noff; ++k)
      h = (h ^ r[k]) * 1099511628211ull;
    size_t s = size_t(h >> (64 - LOG2));
    int32_t id;
    for (;;)
    {
      id = slots[s];
      if (id < 0)
      {
        if (npat >= max_patterns)
          return -1;
        id = npat++;
        std::memcpy(table + size_t(id) * noff, r, size_t(noff));
        slots[s] = id;
        break;
      }
      if (std::memcmp(table + size_t(id) * noff, r, size_t(noff)) == 0)
        break;
      s = (s + 1) & ((size_t(1) << LOG2) - 1);
    }
    pattern_ids[e] = static_cast<uint16_t>(id);
  }
  return npat;
}
