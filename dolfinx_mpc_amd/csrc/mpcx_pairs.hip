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

template <class Op, bool CACHED>
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
#pragma unroll
        for (int j = 0; j < ND1; ++j)
        {
          const int off = int((r.w[(6 + j) >> 2] >> (8 * ((6 + j) & 3))) & 0xffu) * BS1;
#pragma unroll
          for (int q = 0; q < BS1; ++q)
          {
            if ((cm >> (j * BS1 + q)) & 1u)
              continue;
            __hip_atomic_fetch_add(s_vals + base + off + q, Op::entry(lz, I, k, j, q), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
  };

  int64_t t = e0 + tid;
  if constexpr (CACHED)
  {
    // two-stage pipeline: record of t + 2 NT and context of t + NT in flight while pair t is computed
    PairRec<W> cur, nxt;
    double cc[CN];
    if (t < e1)
      load_rec<W>(recs + t * W, cur);
    if (t + NT < e1)
      load_rec<W>(recs + (t + NT) * W, nxt);
    if (t < e1)
      fetch_ctx(cur, cc);
    for (; t < e1; t += NT)
    {
      PairRec<W> nn = nxt;
      double cn[CN];
#pragma unroll
      for (int k = 0; k < CN; ++k)
        cn[k] = cc[k];
      if (t + NT < e1)
        fetch_ctx(nxt, cn);
      if (t + 2 * NT < e1)
        load_rec<W>(recs + (t + 2 * NT) * W, nn);
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
      load_rec<W>(recs + t * W, cur);
    for (; t < e1; t += NT)
    {
      PairRec<W> nxt = cur;
      if (t + NT < e1)
        load_rec<W>(recs + (t + NT) * W, nxt);
      double cc[CN];
      fetch_ctx(cur, cc);
      process(cur, cc);
      cur = nxt;
    }
  }
  __syncthreads();
  if (a.store_mode)
    for (int i = tid; i < nnzb; i += NT)
      a.vals[nnz0 + i] = s_vals[i];
  else
    for (int i = tid; i < nnzb; i += NT)
      a.vals[nnz0 + i] += s_vals[i];
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
  const size_t lds = size_t(a.plan.max_nnz) * 8 + (Op::BS0 > 1 ? size_t(a.plan.max_rows + 1) * 4 : 0);
  if (lds > 160 * 1024)
  {
    mpcx_set_error("mpcx_assemble_matrix: row-block plan exceeds 160 KiB of LDS");
    return -4;
  }
  const void* kern = a.pair_ctx ? reinterpret_cast<const void*>(matrix_pairs_kernel<Op, true>)
                                : reinterpret_cast<const void*>(matrix_pairs_kernel<Op, false>);
  if (int rc = check(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)), "hipFuncSetAttribute"))
    return rc;
  // workgroups per CU by LDS; threads so that the CU holds 16 waves (registers allow: <= 128) -- MPCX_PAIRS_THREADS overrides
  static const int env_threads = []
  {
    const char* e = std::getenv("MPCX_PAIRS_THREADS");
    const int t = e ? std::atoi(e) : 0;
    return (t >= 64 && t <= PAIRS_MAX_THREADS && t % 64 == 0) ? t : 0;
  }();
  int threads = env_threads;
  if (threads == 0)
  {
    const int wgs = int((160 * 1024) / (lds + 512));
    threads = wgs >= 4 ? 256 : (wgs >= 2 ? 512 : 1024);
  }
  const unsigned grid = 8u * unsigned((a.plan.num_blocks + 7) / 8);
  hipStream_t stream = static_cast<hipStream_t>(a.stream);
  if (a.pair_ctx)
    hipLaunchKernelGGL((matrix_pairs_kernel<Op, true>), dim3(grid), dim3(threads), lds, stream, a);
  else
    hipLaunchKernelGGL((matrix_pairs_kernel<Op, false>), dim3(grid), dim3(threads), lds, stream, a);
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
