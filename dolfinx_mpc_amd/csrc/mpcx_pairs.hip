// matrix_pairs_kernel: LDS row blocks whose unit of work is one (entity, local row dof) PAIR described by a
// self-contained record, with the entity's geometric context cached in HBM.
//
// Why: the thread-per-entity row-block kernel evaluates every entity that touches a block and
// masks the rows outside it.  A 74 KB block holds ~40 P2 nodes, so an entity keeps 3.8 of its 10 rows on average and
// 60 % of the lanes of every `ds_add_f64` are idle: the LDS pipe, not HBM, bounds the kernel (P2 Poisson 246^3: 14.3 ms,
// 0.29 of the HBM roof).  The row-pair kernel of round 2 removed the masked lanes but recomputed the context (four
// coordinate gathers, a Jacobian inverse) and re-read the masked dofmap row and ten offset bytes through three dependent
// gathers for every pair: 22.6 ms.  Here
//   * everything a pair needs travels in ONE coalesced record (16 B for P2: entity, local row, LDS slot of the row,
//     the nd1 scatter offsets) -- the reference's per-row column search (MatSetValuesLocal behind
//     cpp/assemble_matrix.cpp:546) hoisted to set-up, laid out in the order the kernel consumes it;
//   * the context of an entity is a 48 B (P2 stiffness) / 80 B load from a per-entity array cached per geometry version
//     (mpcx_pair_context), L2-resident across the pairs of an entity;
//   * a wave runs ONE unrolled row body (pairs ordered by local row inside a block), every lane keeps what it
//     computes, neighbouring lanes add into different CSR rows;
//   * record and context of the next pairs are in flight while a pair is computed (two-stage software pipeline).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>
#include <type_traits>

#include "mpcx_elements.hpp"
#include "mpcx_internal.h"

namespace mpcx
{
namespace
{
constexpr int PAIRS_MAX_THREADS = 1024;
constexpr int PAIR_MASK_SHIFT = 28;
constexpr uint32_t PAIR_ENTITY_MASK = (1u << 27) - 1;

__host__ __device__ constexpr int pair_words(int nd1) { return 1 + (nd1 + 2 + 3) / 4; }
// words per entry of the pattern dictionary: W - 1 rounded up to a power of two (one dwordx2 / dwordx4 load per entry)
__host__ __device__ constexpr int pair_dict_stride(int nd1) { return pair_words(nd1) - 1 <= 2 ? 2 : 4; }

inline int check(hipError_t err, const char* what)
{
  if (err != hipSuccess)
  {
    mpcx_set_error(std::string(what) + ": " + hipGetErrorString(err));
    return -100;
  }
  return 0;
}

inline unsigned grid_for(int64_t n, int block) { return static_cast<unsigned>((n + block - 1) / block); }

template <int W>
struct PairRec
{
  uint32_t w[W];
};

template <int W>
__device__ inline void load_rec(const uint32_t* __restrict__ p, PairRec<W>& r)
{
  if constexpr (W == 4)
  {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    r.w[0] = v.x;
    r.w[1] = v.y;
    r.w[2] = v.z;
    r.w[3] = v.w;
  }
  else if constexpr (W == 2)
  {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    r.w[0] = v.x;
    r.w[1] = v.y;
  }
  else
  {
#pragma unroll
    for (int k = 0; k < W; ++k)
      r.w[k] = p[k];
  }
}

template <int N>
__device__ inline void load_ctx(const double* __restrict__ p, double (&c)[N])
{
  // N doubles, 16-byte aligned when N is even (6, 10): dwordx4 loads
  if constexpr (N % 2 == 0)
  {
#pragma unroll
    for (int k = 0; k < N / 2; ++k)
    {
      const double2 v = reinterpret_cast<const double2*>(p)[k];
      c[2 * k] = v.x;
      c[2 * k + 1] = v.y;
    }
  }
  else
  {
#pragma unroll
    for (int k = 0; k < N; ++k)
      c[k] = p[k];
  }
}

// DICT: compact records (two words: word 0 as above, word 1 = row slot | pattern id << 16) + a table of the DISTINCT
// offset patterns (words 1 .. W-1 of a full record with the slot bits cleared), mpcx_pair_compress.  A box mesh has a
// few thousand distinct patterns whatever its size (P2 on a tiled Kuhn mesh: 1 500), so the table stays in L1 / L2 and
// a P2 record shrinks from 16 to 8 bytes -- the kernel is HBM-bound, records are a third of its read traffic.
// CACHED: the context of a pair is gathered from the per-entity array pair_ctx; otherwise computed from the coordinates.
// (Measured and removed: staging the contexts of a block's entities through LDS once per block -- one gather per (block,
// entity) visit instead of one per pair -- P2 Poisson 246^3 9.4 -> 13.8 ms: the extra dependent phase in front of every
// block and the LDS it takes from the resident workgroups cost more than the gathers it saves.)
template <class Op, bool CACHED, bool DICT>
__global__ void __launch_bounds__(PAIRS_MAX_THREADS) matrix_pairs_kernel(mpcx_matrix_args_t a)
{
  constexpr int ND0 = Op::ND0, ND1 = Op::ND1, BS0 = Op::BS0, BS1 = Op::BS1, NV = Op::NV;
  constexpr int W = pair_words(ND1), CN = Op::CTXN;
  static_assert(!Op::DIAG, "component-diagonal forms take the node-block kernel");
  static_assert(ND1 * BS1 <= 32 && ND0 <= 16, "record layout");
  const int NT = blockDim.x;
  extern __shared__ __align__(16) unsigned char smem[];
  // XCD-aware order as in matrix_rowblock_kernel: contiguous runs of row blocks per XCD
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (b >= nb)
    return;
  const int tid = threadIdx.x;
  const int r0 = a.plan.block_row0[b], r1 = a.plan.block_row0[b + 1];
  const int64_t nnz0 = a.rowptr[r0];
  const int nnzb = int(a.rowptr[r1] - nnz0);
  double* s_vals = reinterpret_cast<double*>(smem);
  int32_t* s_rowlo = reinterpret_cast<int32_t*>(s_vals + a.plan.max_nnz); // bs0 > 1 only
  for (int i = tid; i < nnzb; i += NT)
    s_vals[i] = 0.0;
  if constexpr (BS0 > 1)
  {
    const int nrow = r1 - r0;
    for (int rl = tid; rl <= nrow; rl += NT)
      s_rowlo[rl] = int(a.rowptr[r0 + rl] - nnz0);
  }
  __syncthreads();

  const int64_t e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  const uint32_t* __restrict__ recs = a.pair_recs;
  const double* __restrict__ ctxs = a.pair_ctx;
  const uint32_t* __restrict__ dict = a.pair_dict;
  // the offset words of a compact record: the pattern its id selects (a dependent load that hits L1 / L2)
  auto expand = [&](PairRec<W>& r)
  {
    if constexpr (DICT)
    {
      constexpr int DS = pair_dict_stride(ND1);
      const uint32_t w1 = r.w[1];
      const uint32_t* __restrict__ p = dict + size_t(w1 >> 16) * DS;
      uint32_t e[DS];
      if constexpr (DS == 2)
      {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        e[0] = v.x;
        e[1] = v.y;
      }
      else
      {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        e[0] = v.x;
        e[1] = v.y;
        e[2] = v.z;
        e[3] = v.w;
      }
      r.w[1] = (w1 & 0xffffu) | e[0];
#pragma unroll
      for (int k = 2; k < W; ++k)
        r.w[k] = e[k - 1];
    }
  };
  auto load = [&](int64_t t, PairRec<W>& r)
  {
    if constexpr (DICT)
    {
      const uint2 v = *reinterpret_cast<const uint2*>(recs + t * 2);
      r.w[0] = v.x;
      r.w[1] = v.y;
    }
    else
      load_rec<W>(recs + t * W, r);
  };

  auto fetch_ctx = [&](const PairRec<W>& r, double (&c)[CN])
  {
    const int64_t e = r.w[0] & PAIR_ENTITY_MASK;
    if constexpr (CACHED)
      load_ctx<CN>(ctxs + e * CN, c);
    else
    {
      const int64_t cell = a.entities ? a.entities[e * a.estride] : e;
      double cd[NV * 3];
#pragma unroll
      for (int v = 0; v < NV; ++v)
      {
        const int64_t n = a.x_dofmap[cell * NV + v];
#pragma unroll
        for (int k = 0; k < 3; ++k)
          cd[3 * v + k] = a.x[3 * n + k];
      }
      Op::ctx_store(c, cd);
    }
  };
  auto process = [&](const PairRec<W>& r, const double (&c)[CN])
  {
    const uint32_t w0 = r.w[0];
    const int i = int((w0 >> 27) & 15u);
    const uint32_t slot = r.w[1] & 0xffffu;
    if constexpr (BS0 == 1)
    {
      if (slot == 0xffffu)
        return; // Dirichlet / slave row: stays zero
    }
    uint32_t cm = 0; // bit j * BS1 + q: column (j, q) is masked
    if (w0 >> 31)
    {
      const int64_t e = w0 & PAIR_ENTITY_MASK;
      const int64_t cell1 = a.entities1 ? a.entities1[e * a.estride] : e;
#pragma unroll
      for (int j = 0; j < ND1; ++j)
      {
        const uint32_t m = uint32_t(a.mdofmap1[cell1 * ND1 + j]) >> PAIR_MASK_SHIFT;
#pragma unroll
        for (int q = 0; q < BS1; ++q)
          cm |= ((m >> q) & 1u) << (j * BS1 + q);
      }
    }
    typename Op::Lazy lz;
    Op::ctx_load(lz, a.constants, c);
    // MASKED = std::false_type: no column of the entity is masked (all but the entities at Dirichlet / slave dofs):
    // straight-line row bodies, no per-entry test (the tests cost five instructions per entry: 211 -> ~120 VALU
    // instructions per P2 pair)
    auto rows = [&](auto masked_t)
    {
      constexpr bool MASKED = decltype(masked_t)::value;
#pragma unroll
      for (int I = 0; I < ND0; ++I)
      {
        if (i != I)
          continue;
#pragma unroll
        for (int k = 0; k < BS0; ++k)
        {
          int base;
          if constexpr (BS0 == 1)
            base = int(slot);
          else
          {
            if ((slot >> (13 + k)) & 1u)
              continue;
            base = s_rowlo[int(slot & 0x1fffu) * BS0 + k];
          }
          double* row = s_vals + base;
#pragma unroll
          for (int j = 0; j < ND1; ++j)
          {
            const int off = int((r.w[(6 + j) >> 2] >> (8 * ((6 + j) & 3))) & 0xffu) * BS1;
#pragma unroll
            for (int q = 0; q < BS1; ++q)
            {
              if constexpr (MASKED)
              {
                if ((cm >> (j * BS1 + q)) & 1u)
                  continue;
              }
              __hip_atomic_fetch_add(row + off + q, Op::entry(lz, I, k, j, q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
        }
      }
    };
    if (cm == 0)
      rows(std::false_type{});
    else
      rows(std::true_type{});
  };

  int64_t t = e0 + tid;
  if constexpr (CACHED)
  {
    // two-stage pipeline: record of t + 2 NT and context (+ offset pattern) of t + NT in flight while pair t is computed
    PairRec<W> cur, nxt;
    double cc[CN];
    if (t < e1)
      load(t, cur);
    if (t + NT < e1)
      load(t + NT, nxt);
    if (t < e1)
    {
      fetch_ctx(cur, cc);
      expand(cur);
    }
    for (; t < e1; t += NT)
    {
      PairRec<W> nn = nxt;
      double cn[CN];
#pragma unroll
      for (int k = 0; k < CN; ++k)
        cn[k] = cc[k];
      if (t + NT < e1)
      {
        fetch_ctx(nxt, cn);
        expand(nxt);
      }
      if (t + 2 * NT < e1)
        load(t + 2 * NT, nn);
      process(cur, cc);
      cur = nxt;
      nxt = nn;
#pragma unroll
      for (int k = 0; k < CN; ++k)
        cc[k] = cn[k];
    }
  }
  else
  {
    PairRec<W> cur;
    if (t < e1)
      load(t, cur);
    for (; t < e1; t += NT)
    {
      PairRec<W> nxt = cur;
      if (t + NT < e1)
        load(t + NT, nxt);
      double cc[CN];
      fetch_ctx(cur, cc);
      expand(cur);
      process(cur, cc);
      cur = nxt;
    }
  }
  __syncthreads();
  MPCX_WRITE_OUT(a, nnz0, nnzb, s_vals, tid, NT);
}

// constant-free context of every entity (ElementOp::ctx_store), one thread per entity
template <class Op>
__global__ void __launch_bounds__(256) pair_context_kernel(int64_t n, int estride, const int32_t* __restrict__ entities,
                                                           const double* __restrict__ x, const int32_t* __restrict__ x_dofmap,
                                                           double* __restrict__ ctx)
{
  constexpr int NV = Op::NV, CN = Op::CTXN;
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n)
    return;
  const int64_t cell = entities ? entities[e * estride] : e;
  double cd[NV * 3];
#pragma unroll
  for (int v = 0; v < NV; ++v)
  {
    const int64_t nd = x_dofmap[cell * NV + v];
#pragma unroll
    for (int k = 0; k < 3; ++k)
      cd[3 * v + k] = x[3 * nd + k];
  }
  double c[CN];
  Op::ctx_store(c, cd);
#pragma unroll
  for (int k = 0; k < CN; ++k)
    ctx[e * CN + k] = c[k];
}

// one record per pair id (set-up; any element shape)
__global__ void __launch_bounds__(256)
    pair_records_kernel(int64_t n_pairs, const uint32_t* __restrict__ pair_ids, int estride, const int32_t* __restrict__ entities0,
                        const int32_t* __restrict__ entities1, const int32_t* __restrict__ dofmap0, int nd0, int bs0,
                        const int32_t* __restrict__ dofmap1, int nd1, int bs1, const int8_t* __restrict__ bc0,
                        const int8_t* __restrict__ slave0, const int8_t* __restrict__ bc1, const int8_t* __restrict__ slave1,
                        const mpcx_nnz_t* __restrict__ rowptr, const int32_t* __restrict__ cols, int num_blocks,
                        const int32_t* __restrict__ block_row0, uint32_t* __restrict__ recs, int32_t* __restrict__ overflow)
{
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n_pairs)
    return;
  const int W = pair_words(nd1);
  const uint32_t id = pair_ids[t];
  const int64_t e = id / uint32_t(nd0);
  const int i = int(id - uint32_t(e) * uint32_t(nd0));
  int bad = 0;
  if (e > int64_t(PAIR_ENTITY_MASK) || nd0 > 16)
    bad |= 4;
  const int64_t cell0 = entities0 ? entities0[e * estride] : e;
  const int64_t cell1 = entities1 ? entities1[e * estride] : e;
  const int32_t d0 = dofmap0[cell0 * nd0 + i];
  const int64_t r = int64_t(d0) * bs0;
  // row block of the pair: last block whose first row is <= r
  int lo = 0, hi = num_blocks;
  while (hi - lo > 1)
  {
    const int mid = (lo + hi) >> 1;
    if (block_row0[mid] <= r)
      lo = mid;
    else
      hi = mid;
  }
  const int64_t rb0 = block_row0[lo];
  uint32_t rmask = 0;
  for (int k = 0; k < bs0; ++k)
    rmask |= uint32_t((bc0 && bc0[r + k]) || (slave0 && slave0[r + k])) << k;
  uint32_t slot;
  if (bs0 == 1)
  {
    const int64_t s = rowptr[r] - rowptr[rb0];
    if (s >= 0xffff)
      bad |= 2;
    slot = rmask ? 0xffffu : uint32_t(s);
  }
  else
  {
    const int64_t ln = (r - rb0) / bs0;
    if (ln >= (1 << 13) || bs0 > 3)
      bad |= 2;
    slot = uint32_t(ln) | (rmask << 13);
  }
  uint32_t words[1 + (32 + 2 + 3) / 4];
  for (int k = 0; k < W; ++k)
    words[k] = 0;
  words[1] = slot & 0xffffu;
  uint32_t anymask = 0;
  const int64_t p0 = rowptr[r], p1 = rowptr[r + 1];
  for (int j = 0; j < nd1; ++j)
  {
    const int64_t c = int64_t(dofmap1[cell1 * nd1 + j]) * bs1;
    for (int q = 0; q < bs1; ++q)
      anymask |= uint32_t((bc1 && bc1[c + q]) || (slave1 && slave1[c + q]));
    int64_t l = p0, h = p1;
    while (l < h)
    {
      const int64_t m = (l + h) >> 1;
      if (cols[m] < c)
        l = m + 1;
      else
        h = m;
    }
    int64_t off = -1;
    if (l < p1 && cols[l] == c)
      off = (l - p0) / bs1;
    if (off < 0 || off > 255)
    {
      bad |= 1;
      off = 0;
    }
    words[(6 + j) >> 2] |= uint32_t(off) << (8 * ((6 + j) & 3));
  }
  words[0] = (uint32_t(e) & PAIR_ENTITY_MASK) | (uint32_t(i) << 27) | (anymask << 31);
  for (int k = 0; k < W; ++k)
    recs[t * W + k] = words[k];
  if (bad)
    atomicOr(overflow, bad);
}

// ---------------------------------------------------------------------------------------------------------------
// Dictionary of offset patterns (set-up).  A pattern = words 1 .. W-1 of a record with the slot bits cleared.
//   pair_dict_insert_kernel : every pair inserts the 64-bit hash of its pattern into an open-addressing table
//                             (PAIR_DICT_SLOTS slots, atomicCAS on the hash); the thread that claims a slot writes the
//                             pattern next to it.  More than PAIR_DICT_MAX distinct hashes: *count passes the limit and
//                             the caller keeps the full records.
//   pair_dict_number_kernel : dense ids for the occupied slots (one workgroup), compact table.
//   pair_dict_records_kernel: every pair looks its slot up, CHECKS its pattern against the table entry (two patterns
//                             with one hash would otherwise share an entry: *mismatch is set and the caller keeps
//                             the full records) and writes its compact record.
// ---------------------------------------------------------------------------------------------------------------
constexpr int PAIR_DICT_SLOTS = 1 << 17;
constexpr int PAIR_DICT_MAX = 65535;

__device__ inline uint64_t pair_pattern_hash(const uint32_t* __restrict__ rec, int W)
{
  uint64_t h = 0x9e3779b97f4a7c15ull;
  for (int k = 1; k < W; ++k)
  {
    const uint64_t v = (k == 1) ? (rec[1] & 0xffff0000u) : rec[k];
    h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33;
  }
  return h ? h : 1;
}

__global__ void __launch_bounds__(256)
    pair_dict_insert_kernel(int64_t n_pairs, const uint32_t* __restrict__ recs, int W, unsigned long long* __restrict__ keys,
                            uint32_t* __restrict__ raw, int32_t* __restrict__ count)
{
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n_pairs)
    return;
  if (*reinterpret_cast<volatile int32_t*>(count) > PAIR_DICT_MAX)
    return; // already too many: the caller gives up
  const uint32_t* __restrict__ r = recs + t * W;
  const uint64_t h = pair_pattern_hash(r, W);
  uint32_t s = uint32_t(h >> 20) & (PAIR_DICT_SLOTS - 1);
  for (int probe = 0; probe < PAIR_DICT_SLOTS; ++probe)
  {
    const unsigned long long cur = keys[s];
    if (cur == h)
      return;
    if (cur == 0)
    {
      const unsigned long long old = atomicCAS(keys + s, 0ull, (unsigned long long)h);
      if (old == 0)
      {
        raw[size_t(s) * (W - 1)] = r[1] & 0xffff0000u;
        for (int k = 2; k < W; ++k)
          raw[size_t(s) * (W - 1) + k - 1] = r[k];
        atomicAdd(count, 1);
        return;
      }
      if (old == h)
        return;
    }
    s = (s + 1) & (PAIR_DICT_SLOTS - 1);
  }
}

__global__ void __launch_bounds__(1024)
    pair_dict_number_kernel(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ raw, int W, int DS,
                            int32_t* __restrict__ ids, uint32_t* __restrict__ table)
{
  // one workgroup: exclusive scan of the occupancy flags in chunks of 1024 slots
  __shared__ int32_t s_scan[1024];
  __shared__ int32_t s_base;
  const int tid = threadIdx.x;
  if (tid == 0)
    s_base = 0;
  __syncthreads();
  for (int c = 0; c < PAIR_DICT_SLOTS; c += 1024)
  {
    const int slot = c + tid;
    const int f = keys[slot] != 0 ? 1 : 0;
    s_scan[tid] = f;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1)
    {
      const int v = tid >= d ? s_scan[tid - d] : 0;
      __syncthreads();
      s_scan[tid] += v;
      __syncthreads();
    }
    const int id = s_base + s_scan[tid] - f;
    ids[slot] = f ? id : -1;
    if (f && id < PAIR_DICT_MAX)
      for (int k = 0; k < DS; ++k)
        table[size_t(id) * DS + k] = k < W - 1 ? raw[size_t(slot) * (W - 1) + k] : 0u;
    __syncthreads();
    if (tid == 1023)
      s_base += s_scan[1023];
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256)
    pair_dict_records_kernel(int64_t n_pairs, const uint32_t* __restrict__ recs, int W, int DS, const unsigned long long* __restrict__ keys,
                             const int32_t* __restrict__ ids, const uint32_t* __restrict__ table, uint32_t* __restrict__ out,
                             int32_t* __restrict__ mismatch)
{
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n_pairs)
    return;
  const uint32_t* __restrict__ r = recs + t * W;
  const uint64_t h = pair_pattern_hash(r, W);
  uint32_t s = uint32_t(h >> 20) & (PAIR_DICT_SLOTS - 1);
  int id = -1;
  for (int probe = 0; probe < PAIR_DICT_SLOTS; ++probe)
  {
    const unsigned long long cur = keys[s];
    if (cur == h)
    {
      id = ids[s];
      break;
    }
    if (cur == 0)
      break;
    s = (s + 1) & (PAIR_DICT_SLOTS - 1);
  }
  bool ok = id >= 0 && id < PAIR_DICT_MAX;
  if (ok)
  {
    const uint32_t* __restrict__ p = table + size_t(id) * DS;
    ok = p[0] == (r[1] & 0xffff0000u);
    for (int k = 2; k < W; ++k)
      ok = ok && p[k - 1] == r[k];
  }
  if (!ok)
  {
    *mismatch = 1;
    id = 0;
  }
  out[2 * t] = r[0];
  out[2 * t + 1] = (r[1] & 0xffffu) | (uint32_t(id) << 16);
}

// the operators with a compact context (ElementOp::LAZY) that are not component-diagonal
#define MPCX_FOR_PAIR_OPS(X)                                                                                          \
  if (k.form == MPCX_FORM_STIFFNESS && k.bs == 1 && k.bs1 == 1 && k.degree == k.degree1 && k.coeff_degree == 0)       \
  {                                                                                                                   \
    if (k.celltype == MPCX_CELL_TETRAHEDRON && k.degree == 2)                                                         \
      X((ElementOp<3, 2, 1, 2, 1, MPCX_FORM_STIFFNESS>));                                                             \
    if (k.celltype == MPCX_CELL_TRIANGLE && k.degree == 2)                                                            \
      X((ElementOp<2, 2, 1, 2, 1, MPCX_FORM_STIFFNESS>));                                                             \
    if (k.celltype == MPCX_CELL_TETRAHEDRON && k.degree == 1)                                                         \
      X((ElementOp<3, 1, 1, 1, 1, MPCX_FORM_STIFFNESS>));                                                             \
    if (k.celltype == MPCX_CELL_TRIANGLE && k.degree == 1)                                                            \
      X((ElementOp<2, 1, 1, 1, 1, MPCX_FORM_STIFFNESS>));                                                             \
  }                                                                                                                   \
  if (k.form == MPCX_FORM_ELASTICITY && k.degree == k.degree1 && k.bs == k.bs1)                                       \
  {                                                                                                                   \
    if (k.celltype == MPCX_CELL_TETRAHEDRON && k.degree == 1 && k.bs == 3)                                            \
      X((ElementOp<3, 1, 3, 1, 3, MPCX_FORM_ELASTICITY>));                                                            \
    if (k.celltype == MPCX_CELL_TRIANGLE && k.degree == 1 && k.bs == 2)                                               \
      X((ElementOp<2, 1, 2, 1, 2, MPCX_FORM_ELASTICITY>));                                                            \
    if (k.celltype == MPCX_CELL_TETRAHEDRON && k.degree == 2 && k.bs == 3)                                            \
      X((ElementOp<3, 2, 3, 2, 3, MPCX_FORM_ELASTICITY>));                                                            \
    if (k.celltype == MPCX_CELL_TRIANGLE && k.degree == 2 && k.bs == 2)                                               \
      X((ElementOp<2, 2, 2, 2, 2, MPCX_FORM_ELASTICITY>));                                                            \
  }                                                                                                                   \
  if (k.form == MPCX_FORM_DIV_TEST && k.degree == 2 && k.degree1 == 1 && k.bs1 == 1 && k.coeff_degree == 0)           \
  {                                                                                                                   \
    if (k.celltype == MPCX_CELL_TETRAHEDRON && k.bs == 3)                                                             \
      X((ElementOp<3, 2, 3, 1, 1, MPCX_FORM_DIV_TEST>));                                                              \
    if (k.celltype == MPCX_CELL_TRIANGLE && k.bs == 2)                                                                \
      X((ElementOp<2, 2, 2, 1, 1, MPCX_FORM_DIV_TEST>));                                                              \
  }                                                                                                                   \
  if (k.form == MPCX_FORM_DIV_TRIAL && k.degree == 1 && k.degree1 == 2 && k.bs == 1 && k.coeff_degree == 0)           \
  {                                                                                                                   \
    if (k.celltype == MPCX_CELL_TETRAHEDRON && k.bs1 == 3)                                                            \
      X((ElementOp<3, 1, 1, 2, 3, MPCX_FORM_DIV_TRIAL>));                                                             \
    if (k.celltype == MPCX_CELL_TRIANGLE && k.bs1 == 2)                                                               \
      X((ElementOp<2, 1, 1, 2, 2, MPCX_FORM_DIV_TRIAL>));                                                             \
  }

#define MPCX_UNPAREN(...) __VA_ARGS__

template <class Op>
int launch_pairs(const mpcx_matrix_args_t& a)
{
  if (a.nd0 != Op::ND0 || a.nd1 != Op::ND1 || a.bs0 != Op::BS0 || a.bs1 != Op::BS1 || a.nv != Op::NV)
  {
    mpcx_set_error("mpcx_assemble_matrix: dofmap shapes do not match the element kernel");
    return -12;
  }
  if (!Op::lazy_applies(a.kernel) || a.coeffs || a.estride != 1)
  {
    mpcx_set_error("mpcx_assemble_matrix: pair records need a cell integral of an operator with a compact context, "
                   "without coefficients");
    return -8;
  }
  if ((Op::FORM == MPCX_FORM_ELASTICITY && !a.constants))
  {
    mpcx_set_error("mpcx_assemble_matrix: elasticity needs the constants (mu, lambda)");
    return -8;
  }
  size_t lds = size_t(a.plan.max_nnz) * 8 + (Op::BS0 > 1 ? size_t(a.plan.max_rows + 1) * 4 : 0);
  if (lds > 160 * 1024)
  {
    mpcx_set_error("mpcx_assemble_matrix: row-block plan exceeds 160 KiB of LDS");
    return -4;
  }
  const size_t lds_plan = lds; // (the thread count below follows the plan, not a per-launch occupancy cap)
  if (a.lds_floor > 0 && size_t(a.lds_floor) > lds && a.lds_floor <= 160 * 1024)
    lds = size_t(a.lds_floor);
  const bool cached = a.pair_ctx != nullptr;
  const bool dictm = a.pair_dict != nullptr;
  auto pick = [&](auto dict_t) -> const void*
  {
    constexpr bool D = decltype(dict_t)::value;
    return cached ? reinterpret_cast<const void*>(matrix_pairs_kernel<Op, true, D>)
                  : reinterpret_cast<const void*>(matrix_pairs_kernel<Op, false, D>);
  };
  const void* kern = dictm ? pick(std::true_type{}) : pick(std::false_type{});
  if (int rc = check(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)), "hipFuncSetAttribute"))
    return rc;
  // Threads per workgroup: the CU should hold as many waves as the registers allow (8 per SIMD up to 64 VGPRs), spread
  // over the workgroups the LDS of the plan admits -- P2 Poisson 246^3, 37 KB blocks (four per CU): 128 threads 14.0 ms,
  // 256 11.2, 512 9.5; 74 KB blocks: 512 11.2, 1024 10.2; 18 KB blocks with 256 threads 10.0.  MPCX_PAIRS_THREADS overrides.
  const int env_threads = []
  {
    const char* e = std::getenv("MPCX_PAIRS_THREADS");
    const int t = e ? std::atoi(e) : 0;
    return (t >= 64 && t <= PAIRS_MAX_THREADS && t % 64 == 0) ? t : 0;
  }();
  int threads = env_threads;
  if (threads == 0)
  {
    hipFuncAttributes attr;
    if (int rc = check(hipFuncGetAttributes(&attr, kern), "hipFuncGetAttributes"))
      return rc;
    const int per_simd = attr.numRegs <= 64 ? 8 : (attr.numRegs <= 72 ? 7 : (attr.numRegs <= 80 ? 6 : (attr.numRegs <= 96 ? 5 : 4)));
    const int wgs = int((160 * 1024) / (lds_plan + 1024)) < 1 ? 1 : int((160 * 1024) / (lds_plan + 1024));
    threads = 64 * ((4 * per_simd) / (wgs > 16 ? 16 : wgs));
    threads = threads < 128 ? 128 : (threads > PAIRS_MAX_THREADS ? PAIRS_MAX_THREADS : threads);
  }
  const unsigned grid = 8u * unsigned((a.plan.num_blocks + 7) / 8);
  hipStream_t stream = static_cast<hipStream_t>(a.stream);
  void* kargs[] = {const_cast<mpcx_matrix_args_t*>(&a)};
  if (int rc = check(hipLaunchKernel(kern, dim3(grid), dim3(threads), kargs, lds, stream), "pairs kernel launch"))
    return rc;
  return check(hipGetLastError(), "pairs kernel launch");
}
} // namespace

int launch_matrix_pairs(const mpcx_matrix_args_t& a)
{
  const mpcx_kernel_t& k = a.kernel;
  if (!a.pair_recs || a.plan.num_blocks <= 0 || !a.plan.block_row0 || !a.plan.block_ent_off)
  {
    mpcx_set_error("mpcx_assemble_matrix: plan.row_pairs == 2 needs pair_recs (mpcx_pair_records) and the row blocks");
    return -3;
  }
#define X(OP) return launch_pairs<MPCX_UNPAREN OP>(a)
  MPCX_FOR_PAIR_OPS(X)
#undef X
  mpcx_set_error("mpcx_assemble_matrix: no pair-record kernel for this operator");
  return -10;
}
} // namespace mpcx

using namespace mpcx;

extern "C" int32_t mpcx_pair_words(int32_t nd1) { return pair_words(nd1); }
extern "C" int32_t mpcx_pair_dict_stride(int32_t nd1) { return pair_dict_stride(nd1); }

extern "C" int mpcx_pair_records(int64_t n_pairs, const uint32_t* pair_ids, int32_t estride, const int32_t* entities0,
                                 const int32_t* entities1, const int32_t* dofmap0, int32_t nd0, int32_t bs0,
                                 const int32_t* dofmap1, int32_t nd1, int32_t bs1, const int8_t* bc0, const int8_t* slave0,
                                 const int8_t* bc1, const int8_t* slave1, const mpcx_nnz_t* rowptr, const int32_t* cols,
                                 int32_t num_blocks, const int32_t* block_row0, uint32_t* recs, int32_t* overflow, void* stream)
{
  if (nd0 <= 0 || nd0 > 16 || nd1 <= 0 || nd1 * bs1 > 32 || bs0 < 1 || bs0 > 3 || num_blocks <= 0)
  {
    mpcx_set_error("mpcx_pair_records: element shape outside the record layout (nd0 <= 16, nd1 * bs1 <= 32, bs0 <= 3)");
    return -1;
  }
  if (n_pairs <= 0)
    return 0;
  hipLaunchKernelGGL(pair_records_kernel, dim3(grid_for(n_pairs, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n_pairs,
                     pair_ids, estride, entities0, entities1, dofmap0, nd0, bs0, dofmap1, nd1, bs1, bc0, slave0, bc1, slave1,
                     rowptr, cols, num_blocks, block_row0, recs, overflow);
  return check(hipGetLastError(), "pair_records_kernel launch");
}

extern "C" int32_t mpcx_pair_context_size(const mpcx_kernel_t* kernel)
{
  const mpcx_kernel_t& k = *kernel;
#define X(OP) return MPCX_UNPAREN OP ::CTXN
  MPCX_FOR_PAIR_OPS(X)
#undef X
  return 0;
}

extern "C" int mpcx_pair_context(const mpcx_kernel_t* kernel, int64_t n_entities, int32_t estride, const int32_t* entities,
                                 const double* x, const int32_t* x_dofmap, int32_t nv, double* ctx, void* stream)
{
  const mpcx_kernel_t& k = *kernel;
  if (n_entities <= 0)
    return 0;
#define X(OP)                                                                                                         \
  {                                                                                                                   \
    using Op = MPCX_UNPAREN OP;                                                                                       \
    if (nv != Op::NV)                                                                                                 \
    {                                                                                                                 \
      mpcx_set_error("mpcx_pair_context: geometry dofmap does not match the cell type");                             \
      return -12;                                                                                                     \
    }                                                                                                                 \
    hipLaunchKernelGGL(pair_context_kernel<Op>, dim3(grid_for(n_entities, 256)), dim3(256), 0,                        \
                       static_cast<hipStream_t>(stream), n_entities, estride, entities, x, x_dofmap, ctx);            \
    return check(hipGetLastError(), "pair_context_kernel launch");                                                    \
  }
  MPCX_FOR_PAIR_OPS(X)
#undef X
  mpcx_set_error("mpcx_pair_context: no compact context for this operator");
  return -10;
}

extern "C" int64_t mpcx_pair_compress_workspace(int32_t nd1)
{
  const int W = pair_words(nd1);
  // keys [SLOTS] u64, raw [SLOTS][W-1] u32, ids [SLOTS] i32, counters [2] i32
  return int64_t(PAIR_DICT_SLOTS) * 8 + int64_t(PAIR_DICT_SLOTS) * (W - 1) * 4 + int64_t(PAIR_DICT_SLOTS) * 4 + 16;
}

extern "C" int mpcx_pair_compress(int64_t n_pairs, const uint32_t* recs, int32_t nd1, uint32_t* recs2, uint32_t* table,
                                  int32_t* num_patterns, void* workspace, void* stream_)
{
  const int W = pair_words(nd1);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(ws);
  uint32_t* raw = reinterpret_cast<uint32_t*>(ws + size_t(PAIR_DICT_SLOTS) * 8);
  int32_t* ids = reinterpret_cast<int32_t*>(ws + size_t(PAIR_DICT_SLOTS) * 8 + size_t(PAIR_DICT_SLOTS) * (W - 1) * 4);
  int32_t* counters = ids + PAIR_DICT_SLOTS; // [0] distinct patterns, [1] mismatch
  if (int rc = check(hipMemsetAsync(ws, 0, size_t(mpcx_pair_compress_workspace(nd1)), stream), "hipMemsetAsync"))
    return rc;
  if (n_pairs > 0)
  {
    hipLaunchKernelGGL(pair_dict_insert_kernel, dim3(grid_for(n_pairs, 256)), dim3(256), 0, stream, n_pairs, recs, W, keys, raw,
                       counters);
    hipLaunchKernelGGL(pair_dict_number_kernel, dim3(1), dim3(1024), 0, stream, keys, raw, W, pair_dict_stride(nd1), ids, table);
    hipLaunchKernelGGL(pair_dict_records_kernel, dim3(grid_for(n_pairs, 256)), dim3(256), 0, stream, n_pairs, recs, W,
                       pair_dict_stride(nd1), keys, ids,
                       table, recs2, counters + 1);
  }
  if (int rc = check(hipGetLastError(), "pair dictionary kernels"))
    return rc;
  int32_t host[2] = {0, 0};
  if (int rc = check(hipMemcpyAsync(host, counters, 8, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync"))
    return rc;
  if (int rc = check(hipStreamSynchronize(stream), "hipStreamSynchronize"))
    return rc;
  // more distinct patterns than 16-bit ids, or two patterns with one hash: no dictionary (the caller keeps the full records)
  *num_patterns = (host[0] > PAIR_DICT_MAX || host[1] != 0) ? -1 : host[0];
  return 0;
}

// (mpcx_preload, csrc/mpcx_kernels.hip: the first launch from a translation unit loads its code object)
namespace
{
__global__ void preload_pairs_kernel() {}
} // namespace
extern "C" int mpcx_preload_pairs(void* stream)
{
  hipLaunchKernelGGL(preload_pairs_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream));
  return hipGetLastError() == hipSuccess ? 0 : -100;
}
