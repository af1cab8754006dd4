// UFCx import path: the reference's element seam is a C function pointer with the
// UFCx signature
//     void tabulate_tensor(T* A, const T* w, const T* c, const U* coordinate_dofs,
//                          const int* entity_local_index, const uint8_t* quadrature_permutation, void* custom_data)
// (cpp/assemble_matrix.cpp:291-292, 438-439; numba calls it the same way, numba/assemble_matrix.py:282-290).
// A host pointer cannot be called from a kernel, so the seam here is the SOURCE of such a function: it is
// compiled for gfx950 at run time with hipRTC as a __device__ function, together with generic per-entity
// assembly kernels (one thread per entity, element tensor in private memory, Dirichlet masking, the
// K^T A_e K elimination of cpp/assemble_matrix.cpp:99-268, CSR search + device atomics) specialised for the
// element shape through -D options.  Any form FFCx can generate for float64 can be assembled this way; the
// built-in operators remain the fast path (LDS row blocks, closed-form entries).
#include "mpcx.h"
#include "mpcx_internal.h"

#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <mutex>
#include <algorithm>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdio>
#include <string>
#include <vector>

namespace
{
// include/mpcx.h as text (the kernels take the very same argument structs by value)
const char* const MPCX_H_TEXT =
#include "mpcx_h_embed.inc"
    ;
// csrc/mpcx_ufcx_math.hpp as text: sin / cos / exp for the imported kernels (libm's cost 189 / 189 / 62 instructions each)
const char* const MPCX_UFCX_MATH_TEXT =
#include "mpcx_ufcx_math_embed.inc"
    ;

// csrc/mpcx_fan.hpp as text: the six-tet cluster tables and record formats of the cluster kernels
const char* const MPCX_FAN_TEXT =
#include "mpcx_fan_embed.inc"
    ;

const char* const KERNELS_TEXT = R"MPCXK(
#define N0 (ND0 * BS0)
#define N1 (ND1 * BS1)
// dof transformations of the imported element (include/mpcx.h mpcx_ufcx_desc_t::transform0_name / transform1_name): applied to
// the element tensor right after the kernel call, like cpp/assemble_matrix.cpp:507-508 / cpp/assemble_vector.cpp:184
#ifdef UFCX_T0
#define UFCX_POST0(A, info, cell, n) UFCX_T0(A, info, (int)(cell), n)
#else
#define UFCX_POST0(A, info, cell, n)
#endif
#ifdef UFCX_T1
#define UFCX_POST1(A, info, cell, n) UFCX_T1(A, info, (int)(cell), n)
#else
#define UFCX_POST1(A, info, cell, n)
#endif

__device__ inline void atomic_add_f64(double* p, double v)
{
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline long long csr_find(const int* __restrict__ cols, long long lo, long long hi, int col)
{
  const long long end = hi;
  while (lo < hi)
  {
    const long long mid = (lo + hi) >> 1;
    if (cols[mid] < col)
      lo = mid + 1;
    else
      hi = mid;
  }
  return (lo < end && cols[lo] == col) ? lo : -1;
}
__device__ inline void gather(const double* __restrict__ x, const int* __restrict__ xd, long long cell, double* cd)
{
  for (int i = 0; i < NV; ++i)
  {
    const long long v = xd[cell * NV + i];
    for (int k = 0; k < 3; ++k)
      cd[3 * i + k] = x[3 * v + k];
  }
}
// element tensor of entity e through the imported function (caller-zeroed, accumulated into: cpp/assemble_matrix.cpp:504)
__device__ inline void tabulate(double* Ae, int n, const double* coeffs, int cstride, const double* constants,
                                const double* cd, long long e, int lf)
{
  for (int i = 0; i < n; ++i)
    Ae[i] = 0.0;
  const unsigned char perm = 0;
  UFCX_FN(Ae, coeffs ? coeffs + e * cstride : (const double*)0, constants, cd, &lf, &perm, (void*)0);
}

// The element tensor of the per-entity kernels: N0 * N1 doubles of the thread's stack, or -- UFCX_BIG, tensors beyond
// 96 KiB such as vector-valued Q3 hexahedra (192 x 192) -- a slab of a global scratch array the host allocates per kernel
// (one slab per launched thread; the kernels then stride over their work items with a bounded grid).  Big tensors have
// the per-entity kernels only: no LDS row blocks, no master-contribution plan.
#if UFCX_BIG
extern "C" __device__ double* ufcx_scratch_matrix;
extern "C" __device__ double* ufcx_scratch_mpc;
extern "C" __device__ double* ufcx_scratch_lifting;
__device__ double* ufcx_scratch_matrix = 0;
__device__ double* ufcx_scratch_mpc = 0;
__device__ double* ufcx_scratch_lifting = 0;
#define UFCX_ITEMS(t, n, scratch, body)                                                                                                \
  {                                                                                                                                  \
    const long long tid_ = (long long)blockIdx.x * blockDim.x + threadIdx.x, nt_ = (long long)gridDim.x * blockDim.x;                  \
    double* Ae = scratch + (unsigned long long)tid_ * (unsigned long long)(N0 * N1);                                                   \
    for (long long t = tid_; t < (n); t += nt_)                                                                                      \
      body(a, t, Ae);                                                                                                                \
  }
#else
#define UFCX_ITEMS(t, n, scratch, body)                                                                                                \
  {                                                                                                                                  \
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;                                                              \
    if (t < (n))                                                                                                                     \
    {                                                                                                                                \
      double Ae[N0 * N1];                                                                                                            \
      body(a, t, Ae);                                                                                                                \
    }                                                                                                                                \
  }
#endif

#if UFCX_RANK == 2
__device__ inline void matrix_item(const mpcx_matrix_args_t& a, long long e, double* Ae)
{
  const long long l = e * a.estride;
  const long long cell = a.entities ? a.entities[l] : e;
  const long long cell0 = a.entities0 ? a.entities0[l] : e;
  const long long cell1 = a.entities1 ? a.entities1[l] : e;
  const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
  double cd[NV * 3];
  gather(a.x, a.x_dofmap, cell, cd);
  tabulate(Ae, N0 * N1, a.coeffs, a.cstride, a.constants, cd, e, lf);
  UFCX_POST0(Ae, a.cell_info0, cell0, N1);
  UFCX_POST1(Ae, a.cell_info1, cell1, N0);
  // bulk part: Dirichlet and slave rows / columns masked (cpp/assemble_matrix.cpp:510-533, 165-178)
  for (int p = 0; p < N0; ++p)
  {
    const int r = a.dofmap0[cell0 * ND0 + p / BS0] * BS0 + p % BS0;
    if ((a.bc0 && a.bc0[r]) || a.mpc0.is_slave[r])
      continue;
    const long long lo = a.rowptr[r], hi = a.rowptr[r + 1];
    for (int q = 0; q < N1; ++q)
    {
      const int c = a.dofmap1[cell1 * ND1 + q / BS1] * BS1 + q % BS1;
      if ((a.bc1 && a.bc1[c]) || a.mpc1.is_slave[c])
        continue;
      const long long pos = csr_find(a.cols, lo, hi, c);
      if (pos >= 0)
        atomic_add_f64(a.vals + MPCX_VAL_POS(a, pos), Ae[p * N1 + q]);
    }
  }
}
extern "C" __global__ void __launch_bounds__(64) ufcx_matrix_kernel(mpcx_matrix_args_t a)
UFCX_ITEMS(e, a.n_entities, ufcx_scratch_matrix, matrix_item)

// master contributions of the slave entities: cpp/assemble_matrix.cpp:182-267
__device__ inline void matrix_mpc_item(const mpcx_matrix_args_t& a, long long t, double* Ae)
{
  const long long e = a.slave_entities[t];
  const long long l = e * a.estride;
  const long long cell = a.entities ? a.entities[l] : e;
  const long long cell0 = a.entities0 ? a.entities0[l] : e;
  const long long cell1 = a.entities1 ? a.entities1[l] : e;
  const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
  double cd[NV * 3];
  gather(a.x, a.x_dofmap, cell, cd);
  tabulate(Ae, N0 * N1, a.coeffs, a.cstride, a.constants, cd, e, lf);
  UFCX_POST0(Ae, a.cell_info0, cell0, N1);
  UFCX_POST1(Ae, a.cell_info1, cell1, N0);
  int rows[N0], colsd[N1];
  bool rbc[N0], cbc[N1], rsl[N0], csl[N1];
  for (int p = 0; p < N0; ++p)
  {
    const int r = a.dofmap0[cell0 * ND0 + p / BS0] * BS0 + p % BS0;
    rows[p] = r;
    rbc[p] = a.bc0 && a.bc0[r];
    rsl[p] = a.mpc0.is_slave[r];
  }
  for (int q = 0; q < N1; ++q)
  {
    const int c = a.dofmap1[cell1 * ND1 + q / BS1] * BS1 + q % BS1;
    colsd[q] = c;
    cbc[q] = a.bc1 && a.bc1[c];
    csl[q] = a.mpc1.is_slave[c];
  }
  for (int p = 0; p < N0; ++p)
  {
    if (!rsl[p])
      continue;
    for (int mi = a.mpc0.masters_offsets[rows[p]]; mi < a.mpc0.masters_offsets[rows[p] + 1]; ++mi)
    {
      const int m = a.mpc0.masters[mi];
      const double ci = a.mpc0.coeffs[mi];
      const long long lo = a.rowptr[m], hi = a.rowptr[m + 1];
      for (int q = 0; q < N1; ++q)
      {
        const double v = (rbc[p] || cbc[q]) ? 0.0 : Ae[p * N1 + q];
        if (csl[q])
        {
          for (int mj = a.mpc1.masters_offsets[colsd[q]]; mj < a.mpc1.masters_offsets[colsd[q] + 1]; ++mj)
          {
            const long long pos = csr_find(a.cols, lo, hi, a.mpc1.masters[mj]);
            if (pos >= 0)
              atomic_add_f64(a.vals + MPCX_VAL_POS(a, pos), ci * a.mpc1.coeffs[mj] * v);
          }
        }
        else if (!cbc[q])
        {
          const long long pos = csr_find(a.cols, lo, hi, colsd[q]);
          if (pos >= 0)
            atomic_add_f64(a.vals + MPCX_VAL_POS(a, pos), ci * v);
        }
      }
    }
  }
  for (int q = 0; q < N1; ++q)
  {
    if (!csl[q])
      continue;
    for (int mj = a.mpc1.masters_offsets[colsd[q]]; mj < a.mpc1.masters_offsets[colsd[q] + 1]; ++mj)
    {
      const int m = a.mpc1.masters[mj];
      const double cj = a.mpc1.coeffs[mj];
      for (int p = 0; p < N0; ++p)
      {
        if (rsl[p] || rbc[p])
          continue;
        const long long pos = csr_find(a.cols, a.rowptr[rows[p]], a.rowptr[rows[p] + 1], m);
        if (pos >= 0)
          atomic_add_f64(a.vals + MPCX_VAL_POS(a, pos), cj * (cbc[q] ? 0.0 : Ae[p * N1 + q]));
      }
    }
  }
}
extern "C" __global__ void __launch_bounds__(64) ufcx_matrix_mpc_kernel(mpcx_matrix_args_t a)
UFCX_ITEMS(t, a.n_slave_entities, ufcx_scratch_mpc, matrix_mpc_item)

// cpp/lifting.h:77-133: raw element tensor, b -= scale * Ae[:, j] (g_j - x0_j), slaves moved to their masters
__device__ inline void lifting_item(const mpcx_lifting_args_t& a, long long t, double* Ae)
{
  const long long e = a.lift_entities[t];
  const long long l = e * a.estride;
  const long long cell = a.entities ? a.entities[l] : e;
  const long long cell0 = a.entities0 ? a.entities0[l] : e;
  const long long cell1 = a.entities1 ? a.entities1[l] : e;
  const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
  double cd[NV * 3];
  gather(a.x, a.x_dofmap, cell, cd);
  tabulate(Ae, N0 * N1, a.coeffs, a.cstride, a.constants, cd, e, lf);
  UFCX_POST0(Ae, a.cell_info0, cell0, N1);
  UFCX_POST1(Ae, a.cell_info1, cell1, N0);
  double be[N0];
  for (int m = 0; m < N0; ++m)
    be[m] = 0.0;
  for (int q = 0; q < N1; ++q)
  {
    const int jj = a.dofmap1[cell1 * ND1 + q / BS1] * BS1 + q % BS1;
    if (a.bc_markers1[jj])
    {
      const double g = a.scale * (a.bc_values1[jj] - (a.x0 ? a.x0[jj] : 0.0));
      for (int m = 0; m < N0; ++m)
        be[m] -= Ae[m * N1 + q] * g;
    }
  }
  for (int p = 0; p < N0; ++p)
  {
    const int d = a.dofmap0[cell0 * ND0 + p / BS0] * BS0 + p % BS0;
    double v = be[p];
    if (a.mpc0.is_slave[d])
    {
      const int m0 = a.mpc0.masters_offsets[d], m1 = a.mpc0.masters_offsets[d + 1];
      for (int mi = m0; mi < m1; ++mi)
        atomic_add_f64(a.b + MPCX_ROW_POS(a, a.mpc0.masters[mi]), a.mpc0.coeffs[mi] * v);
      if (m1 > m0)
        v = 0.0;
    }
    if (v != 0.0)
      atomic_add_f64(a.b + MPCX_ROW_POS(a, d), v);
  }
}
extern "C" __global__ void __launch_bounds__(64) ufcx_lifting_kernel(mpcx_lifting_args_t a)
UFCX_ITEMS(t, a.n_lift_entities, ufcx_scratch_lifting, lifting_item)
#else
// cpp/assemble_vector.cpp:65-90 + modify_mpc_vec (cpp/assemble_vector.h:35-69)
extern "C" __global__ void __launch_bounds__(64) ufcx_vector_kernel(mpcx_vector_args_t a)
{
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.n_entities)
    return;
  const long long l = e * a.estride;
  const long long cell = a.entities ? a.entities[l] : e;
  const long long cell0 = a.entities0 ? a.entities0[l] : e;
  const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
  double cd[NV * 3];
  gather(a.x, a.x_dofmap, cell, cd);
  double be[N0];
  tabulate(be, N0, a.coeffs, a.cstride, a.constants, cd, e, lf);
  UFCX_POST0(be, a.cell_info0, cell0, 1);
  for (int p = 0; p < N0; ++p)
  {
    const int d = a.dofmap[cell0 * ND0 + p / BS0] * BS0 + p % BS0;
    double v = be[p];
    if (a.mpc.is_slave[d])
    {
      const int m0 = a.mpc.masters_offsets[d], m1 = a.mpc.masters_offsets[d + 1];
      for (int mi = m0; mi < m1; ++mi)
        atomic_add_f64(a.b + MPCX_ROW_POS(a, a.mpc.masters[mi]), a.mpc.coeffs[mi] * v);
      if (m1 > m0)
        v = 0.0;
    }
    if (v != 0.0)
      atomic_add_f64(a.b + MPCX_ROW_POS(a, d), v);
  }
}
#endif
)MPCXK";


const char* const ROWBLOCK_KERNELS_TEXT = R"MPCXR(
// ---------------------------------------------------------------------------------------------------------
// The imported tabulate_tensor inside the LDS row-block kernels (the fast path of the built-in operators,
// csrc/mpcx_kernels.hip matrix_rowblock_kernel / vector_rowblock_kernel / vector_ownblock_kernel): the
// per-entity loop of cpp/assemble_matrix.cpp:488-547 with the element tensor in registers (small elements:
// every index is a compile-time constant after unrolling) or private memory (large ones), the Dirichlet /
// slave masks folded into the dofmaps, the CSR positions from the plan's 8-bit scatter offsets, ds_add_f64
// into the workgroup's copy of its row block and one coalesced write of the finished block.
// ---------------------------------------------------------------------------------------------------------
#define MASK_SHIFT 28
#define DOF_MASK ((1 << MASK_SHIFT) - 1)
#define NOFF (ND0 * ND1)
#if UFCX_SMALL
#define UFCX_UNROLL _Pragma("unroll")
#else
#define UFCX_UNROLL _Pragma("nounroll")
#endif

#if UFCX_RANK == 2
#if !UFCX_BIG
extern "C" __global__ void __launch_bounds__(UFCX_RB_THREADS) ufcx_matrix_rowblock_kernel(mpcx_matrix_args_t a)
{
  extern __shared__ __align__(16) unsigned char smem[];
  const int NT = blockDim.x;
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3); // contiguous runs of row blocks per XCD
  if (b >= nb)
    return;
  const int tid = threadIdx.x;
  const int r0 = a.plan.block_row0[b], r1 = a.plan.block_row0[b + 1];
  const int nrow = r1 - r0;
  const long long nnz0 = a.rowptr[r0];
  const int nnzb = (int)(a.rowptr[r1] - nnz0);
  double* s_vals = (double*)smem;                       // [max_nnz]
  int* s_rowlo = (int*)(s_vals + a.plan.max_nnz);       // [max_rows + 1]
  for (int i = tid; i < nnzb; i += NT)
    s_vals[i] = 0.0;
  for (int rl = tid; rl <= nrow; rl += NT)
    s_rowlo[rl] = (int)(a.rowptr[r0 + rl] - nnz0);
  __syncthreads();
  const bool same_maps = (ND0 == ND1) && (a.mdofmap1 == a.mdofmap0) && (a.entities1 == a.entities0);
  const bool geom_is_dofmap = (NV == ND0) && (a.x_dofmap == a.dofmap0) && (a.entities0 == a.entities);
  const long long e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  const int* __restrict__ ents = a.plan.block_ents;
  // per-entity index data: everything that is read through the entity index
  struct Ent
  {
    long long e;
    int lf;
    int xd[NV];
    int m0[ND0], m1[ND1];
    unsigned ow[(NOFF + 3) / 4]; // scatter offsets, 4 per word
  };
  auto load_ent = [&](long long e, Ent& E)
  {
    E.e = e;
    const long long l = e * a.estride;
    const long long cell = a.entities ? a.entities[l] : e;
    const long long cell0 = a.entities0 ? a.entities0[l] : e;
    const long long cell1 = a.entities1 ? a.entities1[l] : e;
    E.lf = a.estride == 2 ? a.entities[l + 1] : 0;
#pragma unroll
    for (int i = 0; i < ND0; ++i)
      E.m0[i] = a.mdofmap0[cell0 * ND0 + i];
#pragma unroll
    for (int j = 0; j < ND1; ++j)
      E.m1[j] = same_maps ? E.m0[j < ND0 ? j : 0] : a.mdofmap1[cell1 * ND1 + j];
#pragma unroll
    for (int i = 0; i < NV; ++i)
      E.xd[i] = geom_is_dofmap ? (E.m0[i < ND0 ? i : 0] & DOF_MASK) : a.x_dofmap[cell * NV + i];
    // scatter offsets of this entity: NOFF bytes, contiguous
    const unsigned char* po = a.plan.ent_offs + e * NOFF;
#if NOFF % 16 == 0
#pragma unroll
    for (int w = 0; w < NOFF / 16; ++w)
    {
      const uint4 v = ((const uint4*)po)[w];
      E.ow[4 * w] = v.x, E.ow[4 * w + 1] = v.y, E.ow[4 * w + 2] = v.z, E.ow[4 * w + 3] = v.w;
    }
#elif NOFF % 4 == 0
#pragma unroll
    for (int w = 0; w < NOFF / 4; ++w)
      E.ow[w] = ((const unsigned*)po)[w];
#else
#pragma unroll
    for (int w = 0; w < (NOFF + 3) / 4; ++w)
    {
      unsigned u = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (4 * w + q < NOFF)
          u |= (unsigned)po[4 * w + q] << (8 * q);
      E.ow[w] = u;
    }
#endif
  };
  // small elements: software pipeline -- while one entity is computed, the index data of the next one and the
  // entity index of the one after it are in flight, so an iteration only waits for its own coordinate gather
#if NOFF <= 16
#define UFCX_PIPE 1
#else
#define UFCX_PIPE 0
#endif
  long long t = e0 + tid;
  Ent cur;
  int i1 = 0;
#if UFCX_PIPE
  if (t < e1)
    load_ent(ents[t], cur);
  if (t + NT < e1)
    i1 = ents[t + NT];
#endif
  for (; t < e1; t += NT)
  {
#if !UFCX_PIPE
    load_ent(ents[t], cur);
#endif
    double cd[NV * 3];
#pragma unroll
    for (int i = 0; i < NV; ++i)
    {
      const long long v = cur.xd[i];
#pragma unroll
      for (int k = 0; k < 3; ++k)
        cd[3 * i + k] = a.x[3 * v + k];
    }
#if UFCX_PIPE
    Ent nxt = cur;
    if (t + NT < e1)
      load_ent(i1, nxt);
    if (t + 2 * NT < e1)
      i1 = ents[t + 2 * NT];
#endif
    const long long e = cur.e;
    const int lf = cur.lf;
    const int (&m0)[ND0] = cur.m0;
    const int (&m1)[ND1] = cur.m1;
    const unsigned (&ow)[(NOFF + 3) / 4] = cur.ow;
#if UFCX_ROWWISE
    // Row by row (element tensors of more than 36 entries, round 6): the imported function is inlined ONCE PER ROW of the tensor
    // and only that row is read afterwards, so every other entry -- and whatever feeds only them -- is dead code in that copy
    // (the loops of the text are unrolled: mpcx_ufcx.cpp adds the pragmas): ND1 * BS1 accumulators live instead of N0 * N1
    // (scalar P2: 330 -> ~80 VGPRs; P2^3: no 7 KB of scratch per thread), and a copy runs only for the rows the entity keeps
    // in this block (the entities of a block are ordered by that set, so a wave takes the same copies together).
    _Pragma("unroll")
    for (int i = 0; i < ND0; ++i)
    {
      // one copy per local NODE row: its BS0 component rows come out of the same copy (component-diagonal forms: the same
      // values, the structural zeros of the other components fold away at compile time) -- a third of the copies and of
      // the instruction-cache footprint for the Taylor-Hood velocity block
      const int rn = (m0[i] & DOF_MASK) * BS0;
      if (rn < r0 || rn >= r1 || ((m0[i] >> MASK_SHIFT) & ((1 << BS0) - 1)) == ((1 << BS0) - 1))
        continue;
      double Ar[N0 * N1];
      _Pragma("unroll")
      for (int z = 0; z < BS0 * N1; ++z)
        Ar[i * BS0 * N1 + z] = 0.0; // (the other rows stay unset: nothing reads them)
      {
        // the coordinates of THIS copy behind an empty asm: the copies must not share their geometry / basis-gradient
        // temporaries (hoisted in front of the row tests they would all be live at once: 512 VGPRs + spills for scalar P2)
        double cr[NV * 3];
        _Pragma("unroll")
        for (int z = 0; z < NV * 3; ++z)
        {
          cr[z] = cd[z];
          asm volatile("" : "+v"(cr[z]));
        }
        const unsigned char perm = 0;
        ufcx_rw::UFCX_FN(Ar, a.coeffs ? a.coeffs + e * a.cstride : (const double*)0, a.constants, cr, &lf, &perm, (void*)0);
      }
      _Pragma("unroll")
      for (int k = 0; k < BS0; ++k)
      {
        if ((m0[i] >> (MASK_SHIFT + k)) & 1)
          continue;
        const int base = s_rowlo[rn + k - r0];
        _Pragma("unroll")
        for (int j = 0; j < ND1; ++j)
        {
          const int off = (int)((ow[(i * ND1 + j) >> 2] >> (8 * ((i * ND1 + j) & 3))) & 0xff) * BS1;
          _Pragma("unroll")
          for (int q = 0; q < BS1; ++q)
          {
            if ((m1[j] >> (MASK_SHIFT + q)) & 1)
              continue;
            const double v = Ar[(i * BS0 + k) * N1 + j * BS1 + q];
            if (v != 0.0)
              __hip_atomic_fetch_add(s_vals + base + off + q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
#else
    double Ae[N0 * N1];
    UFCX_UNROLL
    for (int i = 0; i < N0 * N1; ++i)
      Ae[i] = 0.0;
    {
      const unsigned char perm = 0;
      UFCX_FN(Ae, a.coeffs ? a.coeffs + e * a.cstride : (const double*)0, a.constants, cd, &lf, &perm, (void*)0);
    }
#if defined(UFCX_T0) || defined(UFCX_T1)
    {
      const long long lt = e * a.estride;
      UFCX_POST0(Ae, a.cell_info0, a.entities0 ? a.entities0[lt] : e, N1);
      UFCX_POST1(Ae, a.cell_info1, a.entities1 ? a.entities1[lt] : e, N0);
    }
#endif
    UFCX_UNROLL
    for (int i = 0; i < ND0; ++i)
    {
      UFCX_UNROLL
      for (int k = 0; k < BS0; ++k)
      {
        const int r = (m0[i] & DOF_MASK) * BS0 + k;
        if (r < r0 || r >= r1 || ((m0[i] >> (MASK_SHIFT + k)) & 1))
          continue;
        const int base = s_rowlo[r - r0];
        UFCX_UNROLL
        for (int j = 0; j < ND1; ++j)
        {
          const int off = (int)((ow[(i * ND1 + j) >> 2] >> (8 * ((i * ND1 + j) & 3))) & 0xff) * BS1;
          UFCX_UNROLL
          for (int q = 0; q < BS1; ++q)
          {
            if ((m1[j] >> (MASK_SHIFT + q)) & 1)
              continue;
            const double v = Ae[(i * BS0 + k) * N1 + j * BS1 + q];
            if (v != 0.0) // structural zeros of blocked forms cost no LDS atomic
              __hip_atomic_fetch_add(s_vals + base + off + q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
#endif // UFCX_ROWWISE
#if UFCX_PIPE
    cur = nxt;
#endif
  }
  __syncthreads();
  MPCX_WRITE_OUT(a, nnz0, nnzb, s_vals, tid, NT);
}

#if UFCX_ROWWISE
// Pair records for imported text (round 6; plan.row_pairs == 2, the records of mpcx_pair_records, no cached context -- a
// black-box function has none): the unit of work is one (entity, local NODE row) pair whose rows lie in the block, read from
// ONE coalesced record; the pairs of a block are ordered by local row, so a wave runs ONE of the row-wise copies of the text
// (ufcx_matrix_rowblock_kernel above runs, per visit of an entity, the copies of all rows the entity keeps, and a wave the
// union over its lanes), no lane is masked, and masked dofmaps / offset tables are not read.  The geometry is gathered per pair.
#define UFCX_PW (1 + (ND1 + 2 + 3) / 4)
extern "C" __global__ void __launch_bounds__(UFCX_RB_THREADS) ufcx_matrix_pairs_kernel(mpcx_matrix_args_t a)
{
  extern __shared__ __align__(16) unsigned char smem[];
  const int NT = blockDim.x;
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (b >= nb)
    return;
  const int tid = threadIdx.x;
  const int r0 = a.plan.block_row0[b], r1 = a.plan.block_row0[b + 1];
  const long long nnz0 = a.rowptr[r0];
  const int nnzb = (int)(a.rowptr[r1] - nnz0);
  double* s_vals = (double*)smem;
  int* s_rowlo = (int*)(s_vals + a.plan.max_nnz); // BS0 > 1 only
  for (int i = tid; i < nnzb; i += NT)
    s_vals[i] = 0.0;
#if BS0 > 1
  for (int rl = tid; rl <= r1 - r0; rl += NT)
    s_rowlo[rl] = (int)(a.rowptr[r0 + rl] - nnz0);
#endif
  __syncthreads();
  const long long e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  const unsigned* __restrict__ recs = a.pair_recs;
  for (long long t = e0 + tid; t < e1; t += NT)
  {
    unsigned w[UFCX_PW];
#if UFCX_PW == 4
    {
      const uint4 v = *(const uint4*)(recs + t * 4);
      w[0] = v.x, w[1] = v.y, w[2] = v.z, w[3] = v.w;
    }
#else
#pragma unroll
    for (int q = 0; q < UFCX_PW; ++q)
      w[q] = recs[t * UFCX_PW + q];
#endif
    const unsigned w0 = w[0];
    const long long e = w0 & ((1u << 27) - 1);
    const int irow = (int)((w0 >> 27) & 15u);
    const unsigned slot = w[1] & 0xffffu;
#if BS0 == 1
    if (slot == 0xffffu)
      continue; // Dirichlet / slave row: stays zero
#endif
    const long long cell = a.entities ? a.entities[e * a.estride] : e;
    double cd[NV * 3];
#pragma unroll
    for (int v = 0; v < NV; ++v)
    {
      const long long n = a.x_dofmap[cell * NV + v];
#pragma unroll
      for (int k = 0; k < 3; ++k)
        cd[3 * v + k] = a.x[3 * n + k];
    }
    unsigned cm0 = 0, cm1 = 0; // bit j * BS1 + q (low / high 32): column (j, q) is masked
    if (w0 >> 31)
    {
      const long long cell1 = a.entities1 ? a.entities1[e * a.estride] : e;
#pragma unroll
      for (int j = 0; j < ND1; ++j)
      {
        const unsigned m = (unsigned)a.mdofmap1[cell1 * ND1 + j] >> MASK_SHIFT;
#pragma unroll
        for (int q = 0; q < BS1; ++q)
        {
          const int bit = j * BS1 + q;
          if (bit < 32)
            cm0 |= ((m >> q) & 1u) << bit;
          else
            cm1 |= ((m >> q) & 1u) << (bit - 32);
        }
      }
    }
    const int lf = 0;
    _Pragma("unroll")
    for (int I = 0; I < ND0; ++I)
    {
      if (irow != I)
        continue;
      double Ar[N0 * N1];
      _Pragma("unroll")
      for (int z = 0; z < BS0 * N1; ++z)
        Ar[I * BS0 * N1 + z] = 0.0;
      {
        double cr[NV * 3];
        _Pragma("unroll")
        for (int z = 0; z < NV * 3; ++z)
        {
          cr[z] = cd[z];
          asm volatile("" : "+v"(cr[z]));
        }
        const unsigned char perm = 0;
        ufcx_rw::UFCX_FN(Ar, a.coeffs ? a.coeffs + e * a.cstride : (const double*)0, a.constants, cr, &lf, &perm, (void*)0);
      }
      _Pragma("unroll")
      for (int k = 0; k < BS0; ++k)
      {
        int base;
#if BS0 == 1
        base = (int)slot;
#else
        if ((slot >> (13 + k)) & 1u)
          continue;
        base = s_rowlo[(int)(slot & 0x1fffu) * BS0 + k];
#endif
        _Pragma("unroll")
        for (int j = 0; j < ND1; ++j)
        {
          const int off = (int)((w[(6 + j) >> 2] >> (8 * ((6 + j) & 3))) & 0xffu) * BS1;
          _Pragma("unroll")
          for (int q = 0; q < BS1; ++q)
          {
            const int bit = j * BS1 + q;
            if (bit < 32 ? ((cm0 >> bit) & 1u) : ((cm1 >> (bit - 32)) & 1u))
              continue;
            const double v = Ar[(I * BS0 + k) * N1 + j * BS1 + q];
            if (v != 0.0)
              __hip_atomic_fetch_add(s_vals + base + off + q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
  }
  __syncthreads();
  MPCX_WRITE_OUT(a, nnz0, nnzb, s_vals, tid, NT);
}
#endif // UFCX_ROWWISE

// Master contributions from the plan gathered by target position (mpcx_mpc_plan_device): G lanes share one
// target and stride over its tuples; a lane tabulates an entity once for its consecutive tuples (the plan
// lists a target's tuples by entity).  No device atomics, no CSR searches.
extern "C" __global__ void __launch_bounds__(64) ufcx_matrix_mpc_plan_kernel(mpcx_matrix_args_t a)
{
  const int G = a.mpc_plan_group >= 16 ? 16 : (a.mpc_plan_group >= 4 ? 4 : 1);
  const int lane = threadIdx.x & (G - 1);
  const long long t = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / G;
  if (t >= a.mpc_plan_targets)
    return; // the whole group leaves together
  double sum = 0.0;
  double Ae[N0 * N1];
  long long last = -1;
  for (long long k = a.mpc_plan_off[t] + lane; k < a.mpc_plan_off[t + 1]; k += G)
  {
    const long long e = a.mpc_plan_ent[k];
    if (e != last)
    {
      const long long l = e * a.estride;
      const long long cell = a.entities ? a.entities[l] : e;
      const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
      double cd[NV * 3];
      gather(a.x, a.x_dofmap, cell, cd);
      tabulate(Ae, N0 * N1, a.coeffs, a.cstride, a.constants, cd, e, lf);
      UFCX_POST0(Ae, a.cell_info0, a.entities0 ? a.entities0[l] : e, N1);
      UFCX_POST1(Ae, a.cell_info1, a.entities1 ? a.entities1[l] : e, N0);
      last = e;
    }
    sum += a.mpc_plan_coef[k] * Ae[a.mpc_plan_pq[k]];
  }
  for (int m = G >> 1; m > 0; m >>= 1)
    sum += __shfl_xor(sum, m, G);
  if (lane == 0)
    a.vals[MPCX_VAL_POS(a, a.mpc_plan_tgt[t])] += sum;
}

// The same with every slave entity tabulated once (slave_tensors + mpc_plan_slot, include/mpcx.h): a thread per slave
// entity stores its tensor entry-major (coalesced), then G lanes per target sum coef * entry over the target's tuples.
extern "C" __global__ void __launch_bounds__(64) ufcx_slave_tensors_kernel(mpcx_matrix_args_t a)
{
  const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.n_slave_entities)
    return;
  const long long e = a.slave_entities[s];
  const long long l = e * a.estride;
  const long long cell = a.entities ? a.entities[l] : e;
  const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
  double cd[NV * 3];
  gather(a.x, a.x_dofmap, cell, cd);
  double Ae[N0 * N1];
  tabulate(Ae, N0 * N1, a.coeffs, a.cstride, a.constants, cd, e, lf);
  UFCX_POST0(Ae, a.cell_info0, a.entities0 ? a.entities0[l] : e, N1);
  UFCX_POST1(Ae, a.cell_info1, a.entities1 ? a.entities1[l] : e, N0);
#pragma unroll
  for (int i = 0; i < N0 * N1; ++i)
    a.slave_tensors[(long long)i * a.n_slave_entities + s] = Ae[i];
}

extern "C" __global__ void __launch_bounds__(64) ufcx_matrix_mpc_gather_kernel(mpcx_matrix_args_t a)
{
  const int G = a.mpc_plan_group >= 16 ? 16 : (a.mpc_plan_group >= 4 ? 4 : 1);
  const int lane = threadIdx.x & (G - 1);
  const long long t = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / G;
  if (t >= a.mpc_plan_targets)
    return;
  double sum = 0.0;
  for (long long k = a.mpc_plan_off[t] + lane; k < a.mpc_plan_off[t + 1]; k += G)
    sum += a.mpc_plan_coef[k] * a.slave_tensors[(long long)a.mpc_plan_pq[k] * a.n_slave_entities + a.mpc_plan_slot[k]];
  for (int m = G >> 1; m > 0; m >>= 1)
    sum += __shfl_xor(sum, m, G);
  if (lane == 0)
    a.vals[MPCX_VAL_POS(a, a.mpc_plan_tgt[t])] += sum;
}
#endif // !UFCX_BIG
#else
// rank 1: row blocks of b in LDS.  own_lmap == NULL: every block evaluates the entities touching it and keeps its
// own rows (vector_rowblock_kernel); own_lmap != NULL: owner-computes (vector_ownblock_kernel): every entity once,
// the LDS copy holds the block's rows followed by its halo, the halo part is written out for
// vector_spill_reduce_kernel.  Rows of slave dofs are skipped (flag in the masked dofmap / position table) and
// handled by ufcx_vector_mpc_kernel.
extern "C" __global__ void __launch_bounds__(UFCX_RB_THREADS) ufcx_vector_rowblock_kernel(mpcx_vector_args_t a)
{
  extern __shared__ __align__(16) unsigned char smem[];
  double* s_b = (double*)smem;
  const int NT = blockDim.x;
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (b >= nb)
    return;
  const int tid = threadIdx.x;
  const int r0 = a.plan.block_row0[b], r1 = a.plan.block_row0[b + 1];
  const bool owner = a.own_lmap != (const int*)0;
  const long long h0 = owner ? a.own_hoff[b] : 0, h1 = owner ? a.own_hoff[b + 1] : 0;
  const int nown = r1 - r0, nhalo = (int)(h1 - h0) * BS0;
  for (int i = tid; i < nown + nhalo; i += NT)
    s_b[i] = 0.0;
  __syncthreads();
  const long long e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  const int* __restrict__ ents = a.plan.block_ents;
  for (long long t = e0 + tid; t < e1; t += NT)
  {
    const long long e = ents[t];
    const long long l = e * a.estride;
    const long long cell = a.entities ? a.entities[l] : e;
    const long long cell0 = a.entities0 ? a.entities0[l] : e;
    const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
    double cd[NV * 3];
    gather(a.x, a.x_dofmap, cell, cd);
    double be[N0];
    tabulate(be, N0, a.coeffs, a.cstride, a.constants, cd, e, lf);
    UFCX_POST0(be, a.cell_info0, cell0, 1);
    UFCX_UNROLL
    for (int i = 0; i < ND0; ++i)
    {
      // owner-computes: LDS position from the table (read after the element kernel); else the masked dof
      const int w = owner ? a.own_lmap[e * ND0 + i] : a.mdofmap[cell0 * ND0 + i];
      UFCX_UNROLL
      for (int k = 0; k < BS0; ++k)
      {
        if ((w >> (MASK_SHIFT + k)) & 1)
          continue;
        int pos = (w & DOF_MASK) * BS0 + k;
        if (!owner)
        {
          if (pos < r0 || pos >= r1)
            continue;
          pos -= r0;
        }
        __hip_atomic_fetch_add(s_b + pos, be[i * BS0 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < nown; i += NT)
    a.b[MPCX_ROW_POS(a, r0 + i)] += s_b[i];
  for (int i = tid; i < nhalo; i += NT)
    a.own_spill[h0 * BS0 + i] = s_b[nown + i];
}

// slave rows of the entities that have any (modify_mpc_vec, cpp/assemble_vector.h:35-69)
// (contributions merged per target row in an LDS hash table before the device atomics, as vector_mpc_kernel does)
#define UFCX_VMPC_LOG2H 10
#define UFCX_VMPC_H (1 << UFCX_VMPC_LOG2H)
extern "C" __global__ void __launch_bounds__(64) ufcx_vector_mpc_kernel(mpcx_vector_args_t a)
{
  __shared__ int s_key[UFCX_VMPC_H];
  __shared__ double s_val[UFCX_VMPC_H];
  for (int i = threadIdx.x; i < UFCX_VMPC_H; i += 64)
  {
    s_key[i] = -1;
    s_val[i] = 0.0;
  }
  __syncthreads();
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < a.n_slave_entities)
  {
    const long long e = a.slave_entities[t];
    const long long l = e * a.estride;
    const long long cell = a.entities ? a.entities[l] : e;
    const long long cell0 = a.entities0 ? a.entities0[l] : e;
    const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
    double cd[NV * 3];
    gather(a.x, a.x_dofmap, cell, cd);
    double be[N0];
    tabulate(be, N0, a.coeffs, a.cstride, a.constants, cd, e, lf);
    UFCX_POST0(be, a.cell_info0, cell0, 1);
    for (int p = 0; p < N0; ++p)
    {
      const int d = a.dofmap[cell0 * ND0 + p / BS0] * BS0 + p % BS0;
      if (!a.mpc.is_slave[d])
        continue;
      const int m0 = a.mpc.masters_offsets[d], m1 = a.mpc.masters_offsets[d + 1];
      for (int mi = m0 - (m1 == m0 ? 1 : 0); mi < m1; ++mi)
      {
        const int row = mi < m0 ? d : a.mpc.masters[mi];
        const double v = mi < m0 ? be[p] : a.mpc.coeffs[mi] * be[p];
        unsigned h = ((unsigned)row * 2654435761u) >> (32 - UFCX_VMPC_LOG2H);
        int probe = 0;
        for (; probe < 32; ++probe)
        {
          const int old = atomicCAS(&s_key[h], -1, row);
          if (old == -1 || old == row)
          {
            __hip_atomic_fetch_add(&s_val[h], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
          }
          h = (h + 1) & (UFCX_VMPC_H - 1);
        }
        if (probe == 32)
          atomic_add_f64(a.b + MPCX_ROW_POS(a, row), v);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < UFCX_VMPC_H; i += 64)
    if (s_key[i] >= 0)
      atomic_add_f64(a.b + MPCX_ROW_POS(a, s_key[i]), s_val[i]);
}
#endif
)MPCXR";

const char* const CUBE_KERNELS_TEXT = R"MPCXC(
// ---------------------------------------------------------------------------------------------------------
// The imported tabulate_tensor inside the CLUSTER kernels (csrc/mpcx_cubes.hip matrix_cube_kernel /
// vector_cube_own_kernel; scalar P1 on tetrahedra): one thread takes the six tets round a shared edge, calls the
// imported function once per tet with that tet's coordinates in the mesh's own vertex order (the caller vouches for
// it: mpcx_cluster_ordered), sums the six tensors per vertex pair in registers and scatters 46 values per cluster
// instead of 96 (vectors: 8 instead of 24).  Nothing is assumed about the function: no symmetry, no closed form;
// coefficients are taken per cell through cube_cells.  The loop being replaced: cpp/assemble_matrix.cpp:488-547,
// cpp/assemble_vector.cpp:65-90.
// ---------------------------------------------------------------------------------------------------------
#if UFCX_CUBE
using namespace mpcx_fan;
#if UFCX_RANK == 2
template <bool NARROW>
__device__ __attribute__((always_inline)) inline void ufcx_matrix_cube_body(const mpcx_matrix_args_t& a, unsigned char* smem)
{
  const int NT = blockDim.x;
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3); // contiguous runs of row blocks per XCD
  if (b >= nb)
    return;
  const int tid = threadIdx.x;
  const int bb = a.cube_block_ids ? a.cube_block_ids[b] : b;
  const int r0 = a.plan.block_row0[bb], r1 = a.plan.block_row0[bb + 1];
  const int nrow = r1 - r0;
  const long long nnz0 = a.rowptr[r0];
  const int nnzb = (int)(a.rowptr[r1] - nnz0);
  double* s_vals = (double*)smem;
  int* s_rowlo = (int*)(s_vals + a.plan.max_nnz);
  constexpr int NW = NARROW ? 4 : 6; // 16-byte words per record
  const uint4* __restrict__ recs = (const uint4*)a.cube_recs;
  const long long e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  const int* __restrict__ ridx = a.cube_rec_index;
  const bool has_w = a.coeffs != (const double*)0; // (uniform: forms with coefficients read the cells of their cluster)
  auto load = [&](long long t, uint4 (&w)[NW])
  {
    const uint4* p = recs + (ridx ? (long long)ridx[t] : t) * NW;
#pragma unroll
    for (int i = 0; i < NW; ++i)
      w[i] = p[i];
  };
  auto offset_of = [&](const uint4 (&w)[NW], int i, int j) -> int
  {
    unsigned ow[4 * (NW - 2)];
#pragma unroll
    for (int q = 0; q < NW - 2; ++q)
    {
      ow[4 * q] = w[2 + q].x, ow[4 * q + 1] = w[2 + q].y, ow[4 * q + 2] = w[2 + q].z, ow[4 * q + 3] = w[2 + q].w;
    }
    if constexpr (NARROW)
    {
      const int p = fan_pair_index(i, j);
      return (int)((ow[p >> 3] >> (4 * (p & 7))) & 0xf);
    }
    else
      return (int)((ow[(i * 8 + j) >> 2] >> (8 * ((i * 8 + j) & 3))) & 0xff);
  };
  auto gather8 = [&](const uint4 (&w)[NW], double (&X)[8][3])
  {
    const int v[8] = {(int)w[0].x, (int)w[0].y, (int)w[0].z, (int)w[0].w, (int)w[1].x, (int)w[1].y, (int)w[1].z, (int)w[1].w};
#pragma unroll
    for (int i = 0; i < 8; ++i)
    {
      const long long n = v[i] & DOF_MASK;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        X[i][k] = a.x[3 * n + k];
    }
  };
  // software pipeline as in matrix_cube_kernel: the coordinates of slot t + NT and the record of slot t + 2 NT travel
  // while slot t is computed
  uint4 cur[NW];
  double X[8][3];
#if UFCX_CUBE_PIPE
  uint4 nxt[NW];
  double Xn[8][3];
#endif
  long long t = e0 + tid;
  if (t < e1)
  {
    load(t, cur);
#if UFCX_CUBE_PIPE
    if (t + NT < e1)
      load(t + NT, nxt);
    gather8(cur, X);
#endif
  }
  for (int i = tid; i < nnzb; i += NT)
    s_vals[i] = 0.0;
  for (int rl = tid; rl < nrow; rl += NT)
    s_rowlo[rl] = (int)(a.rowptr[r0 + rl] - nnz0);
  __syncthreads();
  for (; t < e1; t += NT)
  {
    const int v[8] = {(int)cur[0].x, (int)cur[0].y, (int)cur[0].z, (int)cur[0].w,
                      (int)cur[1].x, (int)cur[1].y, (int)cur[1].z, (int)cur[1].w};
    const bool has_next = t + NT < e1;
#if UFCX_CUBE_PIPE
    if (has_next)
      gather8(nxt, Xn);
#else
    // no software pipeline (imported functions are heavy: the registers go to occupancy instead): the record of the next
    // slot travels across the arithmetic, the coordinate round trip is covered by the other waves of the SIMD
    gather8(cur, X);
    uint4 nxt[NW];
    if (has_next)
      load(t + NT, nxt);
#endif
    int cells[6] = {0, 0, 0, 0, 0, 0};
    if (has_w)
    {
      const int* pc = a.cube_cells + 6ll * a.plan.block_ents[t];
#pragma unroll
      for (int q = 0; q < 6; ++q)
        cells[q] = pc[q];
    }
    int base[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
    {
      const int r = v[i] & DOF_MASK;
      const bool mine = r >= r0 && r < r1 && !(v[i] >> MASK_SHIFT);
      base[i] = mine ? s_rowlo[mine ? r - r0 : 0] : -1;
    }
    // the six tensors summed per ORDERED vertex pair (static indices: registers); a pair is scattered as soon as its
    // last tet is done
    double A[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        A[i][j] = 0.0;
#pragma unroll
    for (int step = 0; step < 6; ++step)
    {
      const int tet = fan_order(step);
      double cd[12];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k)
          cd[3 * i + k] = X[fan_vertex(tet, i)][k];
      double Ae[16];
#pragma unroll
      for (int i = 0; i < 16; ++i)
        Ae[i] = 0.0;
      {
        const int lf = 0;
        const unsigned char perm = 0;
        UFCX_FN(Ae, has_w ? a.coeffs + (long long)cells[tet] * a.cstride : (const double*)0, a.constants, cd, &lf, &perm, (void*)0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          A[fan_vertex(tet, i)][fan_vertex(tet, j)] += Ae[4 * i + j];
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
        {
          if (fan_last_step(i, j) != step)
            continue;
          if (base[i] >= 0 && !(v[j] >> MASK_SHIFT))
            __hip_atomic_fetch_add(s_vals + base[i] + offset_of(cur, i, j), A[i][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (has_next)
    {
#pragma unroll
      for (int i = 0; i < NW; ++i)
        cur[i] = nxt[i];
#if UFCX_CUBE_PIPE
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k)
          X[i][k] = Xn[i][k];
      if (t + 2 * NT < e1)
        load(t + 2 * NT, nxt);
#endif
    }
  }
  __syncthreads();
  MPCX_WRITE_OUT(a, nnz0, nnzb, s_vals, tid, NT);
}
#if UFCX_CUBE_WAVES
#define UFCX_CUBE_OCC __attribute__((amdgpu_waves_per_eu(UFCX_CUBE_WAVES)))
#else
#define UFCX_CUBE_OCC
#endif
extern "C" __global__ void __launch_bounds__(UFCX_CUBE_THREADS) UFCX_CUBE_OCC ufcx_matrix_cube_wide_kernel(mpcx_matrix_args_t a)
{
  extern __shared__ __align__(16) unsigned char smem[];
  ufcx_matrix_cube_body<false>(a, smem);
}
extern "C" __global__ void __launch_bounds__(UFCX_CUBE_THREADS) UFCX_CUBE_OCC ufcx_matrix_cube_narrow_kernel(mpcx_matrix_args_t a)
{
  extern __shared__ __align__(16) unsigned char smem[];
  ufcx_matrix_cube_body<true>(a, smem);
}
#else
// owner-computes row blocks over the clusters (the plan of vector_cube_own_kernel: plan.block_ents = the clusters a
// block owns, own_lmap = LDS position of every cluster vertex with the slave flag in bit 28)
#if UFCX_VCUBE_WAVES
#define UFCX_VCUBE_OCC __attribute__((amdgpu_waves_per_eu(UFCX_VCUBE_WAVES)))
#else
#define UFCX_VCUBE_OCC
#endif
extern "C" __global__ void __launch_bounds__(UFCX_VCUBE_THREADS) UFCX_VCUBE_OCC ufcx_vector_cube_own_kernel(mpcx_vector_args_t a)
{
  extern __shared__ __align__(16) unsigned char smem[];
  double* s_b = (double*)smem;
  const int NT = blockDim.x;
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (b >= nb)
    return;
  const int tid = threadIdx.x;
  const int r0 = a.plan.block_row0[b], r1 = a.plan.block_row0[b + 1];
  const long long h0 = a.own_hoff[b], h1 = a.own_hoff[b + 1];
  const int nown = r1 - r0, nhalo = (int)(h1 - h0);
  for (int i = tid; i < nown + nhalo; i += NT)
    s_b[i] = 0.0;
  __syncthreads();
  const bool has_w = a.coeffs != (const double*)0;
  const long long e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  const int* __restrict__ ents = a.plan.block_ents;
  for (long long t = e0 + tid; t < e1; t += NT)
  {
    const long long c = ents[t];
    int v[8];
    {
      const uint4* p = (const uint4*)(a.cube_verts + c * 8);
      const uint4 w0 = p[0], w1 = p[1];
      v[0] = w0.x, v[1] = w0.y, v[2] = w0.z, v[3] = w0.w, v[4] = w1.x, v[5] = w1.y, v[6] = w1.z, v[7] = w1.w;
    }
    int cells[6] = {0, 0, 0, 0, 0, 0};
    if (has_w)
    {
#pragma unroll
      for (int q = 0; q < 6; ++q)
        cells[q] = a.cube_cells[6 * c + q];
    }
    double X[8][3];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k)
        X[i][k] = a.x[3 * (long long)v[i] + k];
    double be8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      be8[i] = 0.0;
#pragma unroll
    for (int tet = 0; tet < 6; ++tet)
    {
      double cd[12];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k)
          cd[3 * i + k] = X[fan_vertex(tet, i)][k];
      double be[4] = {0.0, 0.0, 0.0, 0.0};
      {
        const int lf = 0;
        const unsigned char perm = 0;
        UFCX_FN(be, has_w ? a.coeffs + (long long)cells[tet] * a.cstride : (const double*)0, a.constants, cd, &lf, &perm, (void*)0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        be8[fan_vertex(tet, i)] += be[i];
    }
    int w[8];
    {
      const uint4* p = (const uint4*)(a.own_lmap + c * 8);
      const uint4 w0 = p[0], w1 = p[1];
      w[0] = w0.x, w[1] = w0.y, w[2] = w0.z, w[3] = w0.w, w[4] = w1.x, w[5] = w1.y, w[6] = w1.z, w[7] = w1.w;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (!(w[i] >> MASK_SHIFT))
        __hip_atomic_fetch_add(s_b + (w[i] & DOF_MASK), be8[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  for (int i = tid; i < nown; i += NT)
    a.b[MPCX_ROW_POS(a, r0 + i)] += s_b[i];
  for (int i = tid; i < nhalo; i += NT)
    a.own_spill[h0 + i] = s_b[nown + i];
}
#endif
#endif // UFCX_CUBE
)MPCXC";

struct Rtc
{
  void* lib = nullptr;
  int (*create)(void**, const char*, const char*, int, const char**, const char**) = nullptr;
  int (*compile)(void*, int, const char**) = nullptr;
  int (*log_size)(void*, size_t*) = nullptr;
  int (*log)(void*, char*) = nullptr;
  int (*code_size)(void*, size_t*) = nullptr;
  int (*code)(void*, char*) = nullptr;
  int (*destroy)(void**) = nullptr;
};

Rtc& rtc()
{
  static Rtc r;
  static std::once_flag once;
  std::call_once(once,
                 []
                 {
                   // the copy that belongs to the HIP runtime already in the process (torch ships one), else ROCm's
                   for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"})
                     if ((r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL)))
                       break;
                   if (!r.lib)
                     return;
                   auto sym = [&](const char* n) { return dlsym(r.lib, n); };
                   r.create = reinterpret_cast<decltype(r.create)>(sym("hiprtcCreateProgram"));
                   r.compile = reinterpret_cast<decltype(r.compile)>(sym("hiprtcCompileProgram"));
                   r.log_size = reinterpret_cast<decltype(r.log_size)>(sym("hiprtcGetProgramLogSize"));
                   r.log = reinterpret_cast<decltype(r.log)>(sym("hiprtcGetProgramLog"));
                   r.code_size = reinterpret_cast<decltype(r.code_size)>(sym("hiprtcGetCodeSize"));
                   r.code = reinterpret_cast<decltype(r.code)>(sym("hiprtcGetCode"));
                   r.destroy = reinterpret_cast<decltype(r.destroy)>(sym("hiprtcDestroyProgram"));
                 });
  return r;
}

struct UfcxKernel
{
  mpcx_ufcx_desc_t desc{};
  std::vector<char> code; // gfx950 code object
  hipModule_t module = nullptr;
  hipFunction_t matrix = nullptr, matrix_mpc = nullptr, lifting = nullptr, vector = nullptr;
  hipFunction_t matrix_rowblock = nullptr, matrix_mpc_plan = nullptr, vector_rowblock = nullptr, vector_mpc = nullptr;
  hipFunction_t slave_tensors = nullptr, matrix_mpc_gather = nullptr;
  bool rowwise = false;                 // row-wise copies of the text compiled in: ufcx_matrix_pairs_kernel exists
  hipFunction_t matrix_pairs = nullptr;
  bool t0 = false, t1 = false; // compiled with dof transformations: the calls need cell_info0 / cell_info1
  // scalar P1 on tetrahedra: the cluster kernels (MPCX_ALG_CUBE)
  bool cube = false;
  hipFunction_t matrix_cube_wide = nullptr, matrix_cube_narrow = nullptr, vector_cube_own = nullptr;
  int cube_threads = 256;
  int rb_threads = 256; // threads per workgroup of the row-block kernels (their launch bound)
  // element tensors beyond UFCX_BIG_ENTRIES doubles: per-entity kernels only, the tensor of a thread lives in a slab of a
  // global scratch array (one per kernel, so that the matrix and the lifting call of a step may overlap on two streams)
  bool big = false;
  void* scratch[3] = {nullptr, nullptr, nullptr}; // matrix, master contributions, lifting
  int64_t scratch_threads = 0;
};
// 12288 doubles = 96 KiB of the 128 KiB a thread's stack may hold next to the kernel's own arrays
constexpr int UFCX_BIG_ENTRIES = 12288;

int hip_check(hipError_t err, const char* what)
{
  if (err != hipSuccess)
  {
    mpcx_set_error(std::string(what) + ": " + hipGetErrorString(err));
    return -100;
  }
  return 0;
}

// the code object is loaded on first launch: compiling needs no device (hipRTC cross-compiles)
int ensure_loaded(UfcxKernel* k)
{
  if (k->module)
    return 0;
  if (int rc = hip_check(hipModuleLoadData(&k->module, k->code.data()), "hipModuleLoadData"))
    return rc;
  auto get = [&](hipFunction_t* f, const char* name)
  { return hip_check(hipModuleGetFunction(f, k->module, name), "hipModuleGetFunction"); };
  if (k->desc.rank == 2 && k->big)
  {
    if (int rc = get(&k->matrix, "ufcx_matrix_kernel"))
      return rc;
    if (int rc = get(&k->matrix_mpc, "ufcx_matrix_mpc_kernel"))
      return rc;
    if (int rc = get(&k->lifting, "ufcx_lifting_kernel"))
      return rc;
    // slabs for as many threads as 256 MiB per kernel hold (at least one wave, at most 16 384 threads)
    const size_t slab = size_t(k->desc.nd0) * k->desc.bs0 * k->desc.nd1 * k->desc.bs1 * 8;
    k->scratch_threads = std::min<int64_t>(16384, std::max<int64_t>(64, int64_t((size_t(256) << 20) / slab) / 64 * 64));
    const char* names[3] = {"ufcx_scratch_matrix", "ufcx_scratch_mpc", "ufcx_scratch_lifting"};
    for (int i = 0; i < 3; ++i)
    {
      if (int rc = hip_check(hipMalloc(&k->scratch[i], slab * size_t(k->scratch_threads)), "hipMalloc (element tensor scratch)"))
        return rc;
      hipDeviceptr_t sym = nullptr;
      size_t bytes = 0;
      if (int rc = hip_check(hipModuleGetGlobal(&sym, &bytes, k->module, names[i]), "hipModuleGetGlobal"))
        return rc;
      if (int rc = hip_check(hipMemcpyHtoD(sym, &k->scratch[i], sizeof(void*)), "hipMemcpyHtoD"))
        return rc;
    }
    return 0;
  }
  if (k->desc.rank == 2)
  {
    if (int rc = get(&k->matrix, "ufcx_matrix_kernel"))
      return rc;
    if (int rc = get(&k->matrix_mpc, "ufcx_matrix_mpc_kernel"))
      return rc;
    if (int rc = get(&k->matrix_rowblock, "ufcx_matrix_rowblock_kernel"))
      return rc;
    if (k->rowwise)
      if (int rc = get(&k->matrix_pairs, "ufcx_matrix_pairs_kernel"))
        return rc;
    if (int rc = get(&k->matrix_mpc_plan, "ufcx_matrix_mpc_plan_kernel"))
      return rc;
    if (int rc = get(&k->slave_tensors, "ufcx_slave_tensors_kernel"))
      return rc;
    if (int rc = get(&k->matrix_mpc_gather, "ufcx_matrix_mpc_gather_kernel"))
      return rc;
    if (k->cube)
    {
      if (int rc = get(&k->matrix_cube_wide, "ufcx_matrix_cube_wide_kernel"))
        return rc;
      if (int rc = get(&k->matrix_cube_narrow, "ufcx_matrix_cube_narrow_kernel"))
        return rc;
    }
    return get(&k->lifting, "ufcx_lifting_kernel");
  }
  if (k->cube)
    if (int rc = get(&k->vector_cube_own, "ufcx_vector_cube_own_kernel"))
      return rc;
  if (int rc = get(&k->vector_rowblock, "ufcx_vector_rowblock_kernel"))
    return rc;
  if (int rc = get(&k->vector_mpc, "ufcx_vector_mpc_kernel"))
    return rc;
  return get(&k->vector, "ufcx_vector_kernel");
}

// ---------------------------------------------------------------------------------------------------------------
// FFCx file layout.  What FFCx writes to disk is more than the tabulate_tensor function: after the functions come the
// descriptor objects DOLFINx reads -- `ufcx_integral integral_<hash> = { ..., .tabulate_tensor_float64 = <function>, ... };`
// (with `#ifndef __STDC_NO_COMPLEX__` members inside the initialiser), the arrays a form object points to
// (`finite_element_hashes_form_<hash>`, `form_integral_offsets_...`, `static ufcx_integral* form_integrals_...[] = {&integral_...}`),
// `ufcx_form form_<hash> = { ..., .form_integrals = form_integrals_form_<hash>, ... };` and an alias
// `ufcx_form* form_<file>_<name> = &form_<hash>;`.  The reference reaches a kernel only through these objects
// (cpp/assemble_matrix.cpp:438-439: `a.kernel(IntegralType::cell, i, 0)`, filled by DOLFINx from
// `form->form_integrals[k]->tabulate_tensor_float64`).  A device compiler has no use for host descriptor objects (C
// designated initialisers in any order are not C++), so the importer
//   1. walks the top-level items of the text, records what the objects say (integral -> function, list -> integrals,
//      form -> list, alias -> form) and the functions the text defines,
//   2. blanks the descriptor items out (line numbers of the compiler log stay those of the file),
//   3. resolves the name it was given: a function of the text; else a ufcx_integral object; else a ufcx_form object or
//      an alias of one (its first integral with a float64 kernel); no name: the file's only ufcx_integral.
// ---------------------------------------------------------------------------------------------------------------
struct FfcxObjects
{
  std::vector<std::string> functions;                              // functions the text defines
  std::vector<std::pair<std::string, std::string>> integrals;      // (object, tabulate_tensor_float64 or "")
  std::vector<std::pair<std::string, std::vector<std::string>>> lists; // (array, the integral objects it points to)
  std::vector<std::pair<std::string, std::string>> forms;          // (object, .form_integrals array)
  std::vector<std::pair<std::string, std::string>> aliases;        // (pointer, the form object it points to)
};

inline bool ident_char(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_'; }

// identifiers of a piece of text, comments and string literals skipped
std::vector<std::string> identifiers(const std::string& t)
{
  std::vector<std::string> out;
  for (size_t i = 0; i < t.size();)
  {
    if (t.compare(i, 2, "//") == 0)
    {
      while (i < t.size() && t[i] != '\n')
        ++i;
    }
    else if (t.compare(i, 2, "/*") == 0)
    {
      const size_t e = t.find("*/", i + 2);
      i = e == std::string::npos ? t.size() : e + 2;
    }
    else if (t[i] == '"')
    {
      for (++i; i < t.size() && t[i] != '"'; ++i)
        if (t[i] == '\\')
          ++i;
      ++i;
    }
    else if (ident_char(t[i]) && !(t[i] >= '0' && t[i] <= '9'))
    {
      size_t j = i;
      while (j < t.size() && ident_char(t[j]))
        ++j;
      out.push_back(t.substr(i, j - i));
      i = j;
    }
    else
      ++i;
  }
  return out;
}

// value of designated member `.member = <identifier>` inside an initialiser ("" if absent or not an identifier: NULL, 0)
std::string member_value(const std::string& t, const std::string& member)
{
  const std::string key = "." + member;
  for (size_t p = 0; (p = t.find(key, p)) != std::string::npos; p += key.size())
  {
    size_t q = p + key.size();
    if (q < t.size() && ident_char(t[q]))
      continue; // a longer member name
    while (q < t.size() && (t[q] == ' ' || t[q] == '\t' || t[q] == '\n'))
      ++q;
    if (q >= t.size() || t[q] != '=')
      continue;
    ++q;
    while (q < t.size() && (t[q] == ' ' || t[q] == '\t' || t[q] == '\n' || t[q] == '&'))
      ++q;
    size_t e = q;
    while (e < t.size() && ident_char(t[e]))
      ++e;
    const std::string v = t.substr(q, e - q);
    return (v == "NULL" || v == "0" || v == "nullptr") ? std::string() : v;
  }
  return std::string();
}

void ffcx_scan(std::string& user, FfcxObjects& o)
{
  static const char* const prefixes[] = {"finite_element_hashes_", "form_integral_offsets_", "form_integral_ids_", "form_integrals_",
                                         "original_coefficient_position", "coefficient_names_", "constant_names_", "constant_ranks_",
                                         "constant_shapes_", "function_spaces_", "enabled_coefficients_"};
  const size_t n = user.size();
  size_t i = 0;
  while (i < n)
  {
    // skip white space and comments between items
    if (user[i] == ' ' || user[i] == '\t' || user[i] == '\n' || user[i] == '\r')
    {
      ++i;
      continue;
    }
    if (user.compare(i, 2, "//") == 0)
    {
      while (i < n && user[i] != '\n')
        ++i;
      continue;
    }
    if (user.compare(i, 2, "/*") == 0)
    {
      const size_t e = user.find("*/", i + 2);
      i = e == std::string::npos ? n : e + 2;
      continue;
    }
    if (user[i] == '#') // a preprocessor line between items (continuation lines included): kept
    {
      while (i < n && user[i] != '\n')
      {
        if (user[i] == '\\' && i + 1 < n && user[i + 1] == '\n')
          ++i;
        ++i;
      }
      continue;
    }
    // one item: up to the ';' at depth 0, or the '}' that closes a function body
    const size_t start = i;
    int depth = 0;
    bool function = false, body = false;
    size_t first_paren = std::string::npos;
    char last = 0; // last significant character outside braces
    for (; i < n; ++i)
    {
      const char c = user[i];
      if (user.compare(i, 2, "//") == 0)
      {
        while (i < n && user[i] != '\n')
          ++i;
        continue;
      }
      if (user.compare(i, 2, "/*") == 0)
      {
        const size_t e = user.find("*/", i + 2);
        i = e == std::string::npos ? n : e + 1;
        continue;
      }
      if (c == '"' || c == '\'')
      {
        for (++i; i < n && user[i] != c; ++i)
          if (user[i] == '\\')
            ++i;
        continue;
      }
      if (c == '{')
      {
        if (depth == 0 && last == ')')
          function = body = true;
        ++depth;
      }
      else if (c == '}')
      {
        --depth;
        if (depth == 0 && body)
        {
          ++i;
          break;
        }
      }
      else if (c == ';' && depth == 0)
      {
        ++i;
        break;
      }
      if (depth == 0 && c == '(' && first_paren == std::string::npos)
        first_paren = i;
      if (depth == 0 && c != ' ' && c != '\t' && c != '\n' && c != '\r')
        last = c;
    }
    const std::string item = user.substr(start, i - start);
    if (function)
    {
      if (first_paren != std::string::npos)
      {
        size_t e = first_paren;
        while (e > start && (user[e - 1] == ' ' || user[e - 1] == '\t' || user[e - 1] == '\n'))
          --e;
        size_t b = e;
        while (b > start && ident_char(user[b - 1]))
          --b;
        if (e > b)
          o.functions.push_back(user.substr(b, e - b));
      }
      continue;
    }
    // a declaration: its declared name = the identifier before '=' (or before ';'), array brackets skipped
    const size_t eq = item.find('=');
    std::string decl = item.substr(0, eq == std::string::npos ? item.size() : eq);
    if (const size_t br = decl.find('['); br != std::string::npos)
      decl = decl.substr(0, br);
    const std::vector<std::string> ids = identifiers(decl);
    const std::string name = ids.empty() ? std::string() : ids.back();
    const bool pointer = decl.find('*') != std::string::npos;
    bool mentions_integral = false, mentions_form = false, mentions_ufcx = false;
    for (const std::string& id : ids)
    {
      mentions_integral = mentions_integral || id == "ufcx_integral";
      mentions_form = mentions_form || id == "ufcx_form";
      mentions_ufcx = mentions_ufcx || id.compare(0, 5, "ufcx_") == 0;
    }
    bool drop = mentions_ufcx;
    for (const char* pre : prefixes)
      drop = drop || name.compare(0, std::strlen(pre), pre) == 0;
    if (!drop || ids.empty() || ids.front() == "typedef")
      continue;
    const std::string init = eq == std::string::npos ? std::string() : item.substr(eq + 1);
    if (mentions_integral && !pointer)
      o.integrals.emplace_back(name, member_value(init, "tabulate_tensor_float64"));
    else if (mentions_integral && pointer)
    {
      std::vector<std::string> members;
      for (const std::string& id : identifiers(init))
        if (id != "NULL")
          members.push_back(id);
      o.lists.emplace_back(name, members);
    }
    else if (mentions_form && !pointer)
      o.forms.emplace_back(name, member_value(init, "form_integrals"));
    else if (mentions_form && pointer)
    {
      const std::vector<std::string> t = identifiers(init);
      o.aliases.emplace_back(name, t.empty() ? std::string() : t.front());
    }
    for (size_t q = start; q < i; ++q) // blanked, line structure kept
      if (user[q] != '\n')
        user[q] = ' ';
  }
}

// the tabulate_tensor function `name` stands for ("" + *err on failure)
std::string ffcx_resolve(const FfcxObjects& o, const char* name_, std::string* err)
{
  const std::string name = name_ ? name_ : "";
  auto has_fn = [&](const std::string& f) { return std::find(o.functions.begin(), o.functions.end(), f) != o.functions.end(); };
  auto of_integral = [&](const std::string& obj) -> std::string
  {
    for (const auto& p : o.integrals)
      if (p.first == obj)
        return p.second;
    return std::string();
  };
  auto of_form = [&](const std::string& form) -> std::string
  {
    for (const auto& f : o.forms)
      if (f.first == form)
        for (const auto& l : o.lists)
          if (l.first == f.second)
            for (const std::string& integral : l.second)
              if (const std::string fn = of_integral(integral); !fn.empty())
                return fn;
    return std::string();
  };
  std::string fn;
  if (name.empty())
  {
    if (o.integrals.size() == 1)
      fn = o.integrals.front().second;
    else if (o.integrals.empty() && o.functions.size() == 1)
      fn = o.functions.front();
    if (fn.empty())
    {
      *err = "no function name given and the text does not hold exactly one ufcx_integral object (" + std::to_string(o.integrals.size())
             + " found): name the function, the ufcx_integral or the ufcx_form";
      return fn;
    }
  }
  else if (has_fn(name))
    fn = name;
  else if (!(fn = of_integral(name)).empty())
    ;
  else if (!(fn = of_form(name)).empty())
    ;
  else
    for (const auto& a : o.aliases)
      if (a.first == name)
        fn = of_form(a.second);
  if (fn.empty())
    *err = "'" + name + "' is neither a function the text defines nor a ufcx_integral / ufcx_form object (or alias) with a float64 kernel";
  else if (!has_fn(fn))
  {
    *err = "the object '" + name + "' names tabulate_tensor_float64 = " + fn + ", which the text does not define";
    fn.clear();
  }
  return fn;
}

// ---- code objects on disk (opt-in: MPCX_UFCX_CACHE = directory): a form's kernels are then compiled once per (source,
// element shapes, options, library build), like the reference's FFCx / CFFI JIT cache (~/.cache/fenics).  Files are
// written to a temporary name and renamed, so concurrent processes never read a partial file.
std::string cache_dir()
{
  // MPCX_UFCX_CACHE=<dir> | 0 (off).  Unset (round 5): a per-user directory under the temporary directory -- the reference's
  // JIT caches by default too (~/.cache/fenics); a form's kernels take 0.5-5 s to compile and every process of a run (bench
  // children, test workers, the ranks of a multi-GPU job) would otherwise compile them again
  const char* e = std::getenv("MPCX_UFCX_CACHE");
  if (e && *e)
    return std::string(e) != "0" ? std::string(e) : std::string();
  const char* t = std::getenv("TMPDIR");
  return std::string(t && *t ? t : "/tmp") + "/mpcx_ufcx_cache-" + std::to_string(static_cast<long long>(::getuid()));
}
uint64_t fnv1a(const std::string& text, uint64_t h = 1469598103934665603ull)
{
  for (unsigned char c : text)
    h = (h ^ c) * 1099511628211ull;
  return h;
}
std::string cache_path(const std::string& src, const std::vector<std::string>& opts)
{
  const std::string dir = cache_dir();
  if (dir.empty())
    return {};
  uint64_t h = fnv1a(src);
  for (const auto& o : opts)
    h = fnv1a(o, h);
  h = fnv1a(std::string(__DATE__ " " __TIME__), h); // the library build (the embedded kernel text changes with it)
  char name[32];
  std::snprintf(name, sizeof(name), "%016llx.co", static_cast<unsigned long long>(h));
  return dir + "/" + name;
}
bool read_file(const std::string& path, std::vector<char>& out)
{
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f)
    return false;
  std::fseek(f, 0, SEEK_END);
  const long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  out.resize(n > 0 ? size_t(n) : 0);
  const bool ok = n > 0 && std::fread(out.data(), 1, size_t(n), f) == size_t(n);
  std::fclose(f);
  return ok;
}
void write_file_atomically(const std::string& path, const std::vector<char>& data)
{
  const auto slash = path.rfind('/');
  if (slash != std::string::npos)
  {
    std::string dir = path.substr(0, slash);
    for (size_t i = 1; i <= dir.size(); ++i) // mkdir -p
      if (i == dir.size() || dir[i] == '/')
        (void)::mkdir(dir.substr(0, i).c_str(), 0755);
  }
  const std::string tmp = path + ".tmp" + std::to_string(static_cast<long long>(::getpid()));
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f)
    return; // (a read-only home: no cache)
  const bool ok = std::fwrite(data.data(), 1, data.size(), f) == data.size();
  std::fclose(f);
  if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0)
    (void)std::remove(tmp.c_str());
}

template <class Args>
int launch(hipFunction_t f, int64_t n, const Args& a, void* stream, int64_t max_threads = 0)
{
  if (n == 0)
    return 0;
  Args copy = a;
  void* params[] = {&copy};
  unsigned grid = static_cast<unsigned>((n + 63) / 64);
  if (max_threads > 0) // big element tensors: the kernel strides over its items with as many threads as it has scratch slabs
    grid = std::min<unsigned>(grid, unsigned(max_threads / 64));
  return hip_check(hipModuleLaunchKernel(f, grid, 1, 1, 64, 1, 1, 0, static_cast<hipStream_t>(stream), params, nullptr),
                   "hipModuleLaunchKernel");
}

// one workgroup per row block (grid rounded up to a multiple of 8: the kernels map blockIdx to XCD-contiguous runs)
template <class Args>
int launch_blocks(hipFunction_t f, int num_blocks, int threads, size_t lds, const Args& a, void* stream)
{
  if (num_blocks <= 0)
    return 0;
  Args copy = a;
  void* params[] = {&copy};
  if (a.lds_floor > 0 && size_t(a.lds_floor) > lds && a.lds_floor <= 160 * 1024) // per-launch occupancy cap (include/mpcx.h)
    lds = size_t(a.lds_floor);
  const unsigned grid = 8u * unsigned((num_blocks + 7) / 8);
  return hip_check(hipModuleLaunchKernel(f, grid, 1, 1, unsigned(threads), 1, 1, unsigned(lds), static_cast<hipStream_t>(stream),
                                         params, nullptr),
                   "hipModuleLaunchKernel (row blocks)");
}
} // namespace

extern "C" int mpcx_ufcx_rowwise(void* handle) { return handle && static_cast<UfcxKernel*>(handle)->rowwise ? 1 : 0; }

extern "C" int mpcx_ufcx_resolve(const char* source, const char* name, char* out, int32_t out_len)
{
  if (!source || !out || out_len <= 0)
  {
    mpcx_set_error("mpcx_ufcx_resolve: source and out are required");
    return -1;
  }
  std::string user(source);
  FfcxObjects objs;
  ffcx_scan(user, objs);
  std::string err;
  const std::string fn = ffcx_resolve(objs, name, &err);
  if (fn.empty())
  {
    mpcx_set_error("mpcx_ufcx_resolve: " + err);
    return -2;
  }
  if (int32_t(fn.size()) + 1 > out_len)
  {
    mpcx_set_error("mpcx_ufcx_resolve: out too small");
    return -3;
  }
  std::memcpy(out, fn.c_str(), fn.size() + 1);
  return 0;
}

extern "C" void* mpcx_ufcx_compile(const mpcx_ufcx_desc_t* d)
{
  if (!d->source || (d->rank != 1 && d->rank != 2) || d->nd0 <= 0 || d->bs0 <= 0 || d->nv <= 0
      || (d->rank == 2 && (d->nd1 <= 0 || d->bs1 <= 0)))
  {
    mpcx_set_error("mpcx_ufcx_compile: incomplete descriptor");
    return nullptr;
  }
  // translation unit: fixed-width types, the C-ABI structs, the imported C function as a __device__ function
  // (FFCx emits C99: `restrict`, plain functions -- force_cuda_host_device makes them callable from kernels),
  // then the assembly kernels
  std::string hdr(MPCX_H_TEXT);
  for (const std::string inc : {"#include <stdint.h>", "#include <stddef.h>"})
    if (auto p = hdr.find(inc); p != std::string::npos)
      hdr.replace(p, inc.size(), "");
  std::string src = "typedef signed char int8_t;\ntypedef unsigned char uint8_t;\ntypedef short int16_t;\ntypedef unsigned short uint16_t;\n"
                    "typedef int int32_t;\ntypedef unsigned int uint32_t;\ntypedef long long int64_t;\n"
                    "typedef unsigned long long uint64_t;\ntypedef __SIZE_TYPE__ size_t;\n#define restrict __restrict__\n";
  src += hdr;
  // the user's text: #include lines are dropped (hipRTC supplies the fixed-width types above and the math functions
  // as built-ins; FFCx output includes <math.h>, <stdint.h>, <ufcx.h> ...), every function it defines becomes a
  // __host__ __device__ function that is always inlined into the kernels (element tensor in registers)
  std::string user(d->source);
  for (size_t p = 0; (p = user.find("#include", p)) != std::string::npos;)
  {
    size_t b = user.rfind('\n', p);
    b = b == std::string::npos ? 0 : b + 1;
    bool only_space = true;
    for (size_t q = b; q < p; ++q)
      only_space = only_space && (user[q] == ' ' || user[q] == '\t');
    if (only_space)
    {
      size_t e = user.find('\n', p);
      e = e == std::string::npos ? user.size() : e;
      user.replace(b, e - b, "");
      p = b;
    }
    else
      p += 8;
  }
  // a whole FFCx file: the descriptor objects behind the functions are read and blanked out, the name resolved through them
  FfcxObjects objs;
  ffcx_scan(user, objs);
  std::string fn_err;
  std::string fn_name = ffcx_resolve(objs, d->function_name, &fn_err);
  if (fn_name.empty() && d->function_name && d->function_name[0] && user.find(std::string(d->function_name) + "(") != std::string::npos)
    fn_name = d->function_name; // (a definition the item walk did not recognise: taken at its word, the compiler decides)
  if (fn_name.empty())
  {
    mpcx_set_error("mpcx_ufcx_compile: " + fn_err);
    return nullptr;
  }
  // sin / cos / exp of the imported text go to the library's full-range fp64 routines (mpcx_ufcx_math.hpp: <= 2.5 ulp,
  // libm itself outside their fast ranges) unless MPCX_UFCX_LIBM=1 asks for the device libm
  const char* libm = std::getenv("MPCX_UFCX_LIBM");
  if (!(libm && libm[0] == '1'))
  {
    src += "\nextern \"C\" __device__ double __ocml_sin_f64(double);\nextern \"C\" __device__ double __ocml_cos_f64(double);\n"
           "extern \"C\" __device__ double __ocml_exp_f64(double);\n"
           // (out of line and cold: inlined, libm's general paths doubled the kernel and took it over 128 registers)
           "static __device__ __attribute__((noinline, cold)) double mpcx_slow_sin(double x) { return __ocml_sin_f64(x); }\n"
           "static __device__ __attribute__((noinline, cold)) double mpcx_slow_cos(double x) { return __ocml_cos_f64(x); }\n"
           "static __device__ __attribute__((noinline, cold)) double mpcx_slow_exp(double x) { return __ocml_exp_f64(x); }\n"
           "#define MPCX_FM_LIBM_SIN(x) mpcx_slow_sin(x)\n#define MPCX_FM_LIBM_COS(x) mpcx_slow_cos(x)\n"
           "#define MPCX_FM_LIBM_EXP(x) mpcx_slow_exp(x)\n"
           "#define MPCX_UFCX_MATH_FN static __device__ __attribute__((always_inline)) inline\n#define MPCX_FM_DEVICE_TABLE 1\n";
    std::string m(MPCX_UFCX_MATH_TEXT);
    if (auto q = m.find("#pragma once"); q != std::string::npos)
      m.replace(q, 12, "");
    src += m;
    src += "\n#define sin(x) mpcx_fast_sin(x)\n#define cos(x) mpcx_fast_cos(x)\n#define exp(x) mpcx_fast_exp(x)\n";
  }
  src += "\n#pragma clang force_cuda_host_device begin\n";
  src += "#pragma clang attribute push(__attribute__((always_inline)), apply_to = function)\n";
  // Floating-point semantics of the imported text, MPCX_UFCX_FP (the counterpart of the cffi_extra_compile_args a dolfinx user
  // hands FFCx's JIT; rounding of every operation stays IEEE in all modes but "fast"):
  //   strict      nothing relaxed (cc -O2)
  //   reciprocal  x / y may be evaluated as x * (1 / y): one reciprocal for the nine entries of an inverse Jacobian
  //               K = adj(J) / detJ, a multiplication for ``/ 0.02`` -- <= 1 ulp more per division; an IEEE fp64 division is a
  //               13-instruction sequence on gfx950 and FFCx text divides per cell and per point
  //   finite      (default) reciprocal + the values are assumed finite and the sign of a zero not to matter
  //               (-ffinite-math-only -fno-signed-zeros): products with the exact zeros and ones of baked tables fold away
  //               (``K[d][a] * 0.0`` may not under strict semantics: it is NaN for an infinite K); no reassociation, the
  //               result of every surviving operation is the strict one.  Config 2's imported stiffness kernel: 1204 -> 849
  //               VALU instructions per cluster
  //   fast        finite + reassociation (cc -Ofast)
  const char* fpmode = std::getenv("MPCX_UFCX_FP");
  const std::string fp = fpmode ? fpmode : "finite";
  if (fp == "reciprocal" || fp == "finite")
    src += "#pragma clang fp reciprocal(on)\n";
  else if (fp == "fast") // + reassociation, and (compile options below) no NaN / infinity / signed-zero semantics: what -ffast-math
    src += "#pragma clang fp reciprocal(on) reassociate(on)\n"; // gives an FFCx kernel built with cffi_extra_compile_args=["-Ofast"]
  // row-wise mode of the row-block matrix kernel (element tensors of more than 36 entries, no dof transformations): the text's
  // loops must be unrolled so that the entries of the tensor become independent values and the unused rows of a copy are
  // eliminated -- a `#pragma unroll` in front of every for statement of the text (MPCX_UFCX_ROWWISE=0 switches the mode off)
  const int tensor_size = d->rank == 2 ? d->nd0 * d->bs0 * d->nd1 * d->bs1 : 0;
  const bool has_tr = (d->transform0_name && d->transform0_name[0]) || (d->transform1_name && d->transform1_name[0]);
  const char* rw_env = std::getenv("MPCX_UFCX_ROWWISE");
  // (simplices, up to 30 x 30 -- Taylor-Hood velocity block: the copies of larger / tensor-product elements, 27 x 27 for Q2
  // hexahedra with the geometry inside a 27-point loop, take minutes to compile and gain less)
  const bool rowwise = d->rank == 2 && tensor_size > 36 && tensor_size <= 900 && d->nd0 <= 10 && d->nd1 <= 10 && d->nd0 * d->bs0 <= 30
                       && d->nv <= 4 && !has_tr && !(rw_env && rw_env[0] == '0');
  if (rowwise)
  {
    std::string t;
    t.reserve(user.size() + 4096);
    for (size_t i = 0; i < user.size(); ++i)
    {
      if (user.compare(i, 3, "for") == 0 && (i == 0 || !ident_char(user[i - 1])) && i + 3 < user.size() && !ident_char(user[i + 3]))
      {
        size_t q = i + 3;
        while (q < user.size() && (user[q] == ' ' || user[q] == '\t'))
          ++q;
        if (q < user.size() && user[q] == '(')
        {
          // only loops with a literal bound of at most 16 trips (`...; X < N; ...`): the dof / component / geometry loops of an
          // element of up to ten nodes -- a 45-point quadrature loop stays rolled (it does not index the tensor)
          const size_t semi = user.find(';', q);
          const size_t lt = semi == std::string::npos ? std::string::npos : user.find('<', semi);
          const size_t semi2 = semi == std::string::npos ? std::string::npos : user.find(';', semi + 1);
          long bound = -1;
          if (lt != std::string::npos && semi2 != std::string::npos && lt < semi2)
          {
            size_t b = lt + 1;
            if (b < user.size() && user[b] == '=')
              ++b;
            while (b < user.size() && user[b] == ' ')
              ++b;
            size_t e = b;
            while (e < user.size() && user[e] >= '0' && user[e] <= '9')
              ++e;
            size_t f = e;
            while (f < user.size() && user[f] == ' ')
              ++f;
            if (e > b && f == semi2)
              bound = std::atol(user.substr(b, e - b).c_str());
          }
          if (bound >= 0 && bound <= 16)
            t += "_Pragma(\"unroll\") ";
        }
      }
      t += user[i];
    }
    // a SECOND copy of the text, in a namespace of its own: the per-entity / plan / lifting kernels of this translation unit
    // keep the text as it is (fully unrolled, a 900-entry tensor would be 900 live values there)
    src += user;
    src += "\nnamespace ufcx_rw {\n" + t + "\n}\n";
  }
  else
    src += user;
  src += "\n#undef sin\n#undef cos\n#undef exp\n";
  src += "\n#pragma clang attribute pop\n";
  src += "#pragma clang force_cuda_host_device end\n";
  src += KERNELS_TEXT;
  src += ROWBLOCK_KERNELS_TEXT;
  // scalar P1 on tetrahedra (P1 x P1 for bilinear forms): the cluster kernels as well
  const bool has_transform = (d->transform0_name && d->transform0_name[0]) || (d->transform1_name && d->transform1_name[0]);
  // (an element with dof transformations is never P1: the cluster kernels do not carry the hook)
  const bool cube = !has_transform && d->nv == 4 && d->nd0 == 4 && d->bs0 == 1 && (d->rank == 1 || (d->nd1 == 4 && d->bs1 == 1));
  if (cube)
  {
    std::string f(MPCX_FAN_TEXT);
    if (auto q = f.find("#pragma once"); q != std::string::npos)
      f.replace(q, 12, "");
    src += f;
    src += CUBE_KERNELS_TEXT;
  }
  // threads per workgroup / software pipeline / occupancy target of the cluster kernels (experiments: MPCX_UFCX_CUBE_THREADS,
  // MPCX_UFCX_CUBE_PIPE, MPCX_UFCX_CUBE_WAVES for the matrix kernel, MPCX_UFCX_VCUBE_THREADS / _WAVES for the vector kernel)
  auto env_int = [](const char* name, int dflt)
  {
    const char* e = std::getenv(name);
    return e ? std::atoi(e) : dflt;
  };
  int cube_threads = env_int("MPCX_UFCX_CUBE_THREADS", 256);
  if (cube_threads < 64 || cube_threads > 1024 || cube_threads % 64)
    cube_threads = 256;
  int vcube_threads = env_int("MPCX_UFCX_VCUBE_THREADS", 256);
  if (vcube_threads < 64 || vcube_threads > 1024 || vcube_threads % 64)
    vcube_threads = 256;
  const int cube_pipe = env_int("MPCX_UFCX_CUBE_PIPE", 0);
  const int cube_waves = env_int("MPCX_UFCX_CUBE_WAVES", 0), vcube_waves = env_int("MPCX_UFCX_VCUBE_WAVES", 0);
  // element tensors up to 36 entries (P1 scalar, P1 x P1 on triangles / tets, bs <= ...): 512 threads, 128 VGPRs;
  // up to 144 entries: fully unrolled at 256 threads (256 VGPRs); larger ones stay rolled in private memory
  const int size = d->rank == 2 ? d->nd0 * d->bs0 * d->nd1 * d->bs1 : d->nd0 * d->bs0;
  // (row-wise copies: ~90-150 VGPRs instead of 300+: three waves per SIMD from two resident workgroups; MPCX_UFCX_RB_THREADS overrides)
  // (measured, round 6: scalar P2 stiffness text 246^3: 256 threads 26.5 ms, 384 / 512 37 ms (the copies then spill); P1^3
  // elasticity text (144 entries): 512 threads 1.61 ms, 384 1.96, whole tensor at 256 1.73)
  // (few, short copies -- vector P1: 512 threads; ten or more copies -- P2: 256)
  int rb_threads = (d->rank == 1 || size <= 36) ? 512 : ((rowwise && d->nd0 <= 4) ? 512 : 256);
  if (const int t = env_int("MPCX_UFCX_RB_THREADS", 0); t >= 64 && t <= 1024 && t % 64 == 0)
    rb_threads = t;
  const int small = size <= 144 ? 1 : 0;
  const int big = (d->rank == 2 && size > UFCX_BIG_ENTRIES) ? 1 : 0; // element tensor beyond the per-thread scratch limit
  std::vector<std::string> opts
      = {"--offload-arch=gfx950", "-O3", "-munsafe-fp-atomics", "-DUFCX_FN=" + fn_name, "-DUFCX_BIG=" + std::to_string(big),
         "-DUFCX_RB_THREADS=" + std::to_string(rb_threads), "-DUFCX_SMALL=" + std::to_string(small),
         "-DUFCX_ROWWISE=" + std::to_string(rowwise ? 1 : 0),
         "-DUFCX_RANK=" + std::to_string(d->rank), "-DND0=" + std::to_string(d->nd0), "-DBS0=" + std::to_string(d->bs0),
         "-DND1=" + std::to_string(d->rank == 2 ? d->nd1 : 1), "-DBS1=" + std::to_string(d->rank == 2 ? d->bs1 : 1),
         "-DNV=" + std::to_string(d->nv), "-DUFCX_CUBE=" + std::to_string(cube ? 1 : 0),
         "-DUFCX_CUBE_THREADS=" + std::to_string(cube_threads), "-DUFCX_CUBE_PIPE=" + std::to_string(cube_pipe),
         "-DUFCX_CUBE_WAVES=" + std::to_string(cube_waves), "-DUFCX_VCUBE_WAVES=" + std::to_string(vcube_waves),
         "-DUFCX_VCUBE_THREADS=" + std::to_string(vcube_threads)};
  if (d->transform0_name && d->transform0_name[0])
    opts.push_back("-DUFCX_T0=" + std::string(d->transform0_name));
  if (d->rank == 2 && d->transform1_name && d->transform1_name[0])
    opts.push_back("-DUFCX_T1=" + std::string(d->transform1_name));
  if (rowwise)
  {
    // (the unrolled nests of a copy are large before the dead rows go: no size limit on the pragma)
    opts.push_back("-mllvm");
    opts.push_back("-pragma-unroll-threshold=1000000");
  }
  if (fp == "fast" || fp == "finite")
  {
    opts.push_back("-fno-signed-zeros");
    opts.push_back("-ffinite-math-only");
  }
  auto make_handle = [&]()
  {
    auto* k = new UfcxKernel;
    k->rb_threads = rb_threads;
    k->rowwise = rowwise;
    k->cube = cube && !big;
    k->cube_threads = d->rank == 2 ? cube_threads : vcube_threads;
    k->big = big != 0;
    k->desc = *d;
    k->desc.source = nullptr;
    k->desc.function_name = nullptr;
    k->t0 = d->transform0_name && d->transform0_name[0];
    k->t1 = d->rank == 2 && d->transform1_name && d->transform1_name[0];
    k->desc.transform0_name = k->desc.transform1_name = nullptr;
    return k;
  };
  const std::string cached = cache_path(src, opts);
  if (!cached.empty())
  {
    std::vector<char> code;
    if (read_file(cached, code))
    {
      auto* k = make_handle();
      k->code = std::move(code);
      return k;
    }
  }
  Rtc& r = rtc(); // (loaded only when something has to be compiled)
  if (!r.lib || !r.create || !r.compile || !r.code)
  {
    mpcx_set_error("mpcx_ufcx_compile: libhiprtc.so not found");
    return nullptr;
  }
  void* prog = nullptr;
  if (r.create(&prog, src.c_str(), "mpcx_ufcx.hip", 0, nullptr, nullptr) != 0)
  {
    mpcx_set_error("mpcx_ufcx_compile: hiprtcCreateProgram failed");
    return nullptr;
  }
  std::vector<const char*> copts;
  for (auto& o : opts)
    copts.push_back(o.c_str());
  const int rc = r.compile(prog, int(copts.size()), copts.data());
  if (rc != 0)
  {
    size_t n = 0;
    r.log_size(prog, &n);
    std::string log(n + 1, '\0');
    if (n)
      r.log(prog, log.data());
    r.destroy(&prog);
    mpcx_set_error("mpcx_ufcx_compile: hipRTC compilation failed:\n" + log.substr(0, 4000));
    return nullptr;
  }
  auto* k = make_handle();
  size_t n = 0;
  r.code_size(prog, &n);
  k->code.resize(n);
  r.code(prog, k->code.data());
  r.destroy(&prog);
  if (!cached.empty())
    write_file_atomically(cached, k->code);
  return k;
}

extern "C" int mpcx_ufcx_big_tensor(void* handle) { return handle && static_cast<UfcxKernel*>(handle)->big ? 1 : 0; }
extern "C" int64_t mpcx_ufcx_code_size(void* handle) { return handle ? int64_t(static_cast<UfcxKernel*>(handle)->code.size()) : 0; }

extern "C" int mpcx_ufcx_code(void* handle, void* out)
{
  auto* k = static_cast<UfcxKernel*>(handle);
  if (!k || !out)
    return -1;
  std::memcpy(out, k->code.data(), k->code.size());
  return 0;
}

extern "C" void mpcx_ufcx_free(void* handle)
{
  auto* k = static_cast<UfcxKernel*>(handle);
  if (!k)
    return;
  for (void* p : k->scratch)
    if (p)
      (void)hipFree(p);
  if (k->module)
    (void)hipModuleUnload(k->module);
  delete k;
}

namespace mpcx
{
int launch_matrix_ufcx(const mpcx_matrix_args_t& a)
{
  auto* k = static_cast<UfcxKernel*>(const_cast<void*>(a.kernel.ufcx));
  if (!k || k->desc.rank != 2 || a.nd0 != k->desc.nd0 || a.bs0 != k->desc.bs0 || a.nd1 != k->desc.nd1 || a.bs1 != k->desc.bs1
      || a.nv != k->desc.nv)
  {
    mpcx_set_error("mpcx_assemble_matrix: the imported kernel was compiled for other element shapes (or is not bilinear)");
    return -12;
  }
  if ((k->t0 && !a.cell_info0) || (k->t1 && !a.cell_info1))
  {
    mpcx_set_error("mpcx_assemble_matrix: the imported kernel was compiled with dof transformations: cell_info0 / cell_info1 needed");
    return -5;
  }
  if (a.algorithm == MPCX_ALG_CUBE && !k->cube)
  {
    mpcx_set_error("mpcx_assemble_matrix: imported (UFCx) kernels take MPCX_ALG_CUBE for scalar P1 forms on tetrahedra only "
                   "(nd0 = nd1 = nv = 4, bs = 1); MPCX_ALG_ROWBLOCK or MPCX_ALG_ATOMIC otherwise");
    return -3;
  }
  if (int rc = ensure_loaded(k))
    return rc;
  int alg = a.algorithm;
  if (alg == MPCX_ALG_AUTO)
    alg = (a.plan.num_blocks > 0 && !k->big) ? MPCX_ALG_ROWBLOCK : MPCX_ALG_ATOMIC;
  if (k->big)
  {
    if (alg == MPCX_ALG_ROWBLOCK || a.mpc_plan_off)
    {
      mpcx_set_error("mpcx_assemble_matrix (UFCx): an element tensor of more than 12288 entries runs the per-entity kernels only "
                     "(MPCX_ALG_ATOMIC, no master-contribution plan; mpcx_ufcx_big_tensor tells)");
      return -4;
    }
    if (int rc = launch(k->matrix, a.n_entities, a, a.stream, k->scratch_threads))
      return rc;
    return launch(k->matrix_mpc, a.n_slave_entities, a, a.stream, k->scratch_threads);
  }
  if (alg == MPCX_ALG_CUBE)
  {
    // the six tets of a cluster through the imported function, 46 scatter-adds per cluster (include/mpcx.h cube_cells)
    if (a.estride != 1 || a.plan.num_blocks <= 0 || !a.plan.block_row0 || !a.plan.block_ent_off
        || (a.cube_rec_bytes != 0 && a.cube_rec_bytes != 64 && a.cube_rec_bytes != 96))
    {
      mpcx_set_error("mpcx_assemble_matrix (UFCx, clusters): needs a row-block plan over the cluster slots and their records "
                     "(mpcx_cube_records; cube_rec_bytes 96 or 64), cell integrals only");
      return -3;
    }
    if (a.coeffs && (!a.cube_cells || !a.plan.block_ents))
    {
      mpcx_set_error("mpcx_assemble_matrix (UFCx, clusters): a form with coefficients needs cube_cells and plan.block_ents "
                     "(the cluster of every slot)");
      return -5;
    }
    const size_t lds = size_t(a.plan.max_nnz) * 8 + size_t(a.plan.max_rows) * 4;
    if (lds > 160 * 1024)
    {
      mpcx_set_error("mpcx_assemble_matrix: row-block plan exceeds 160 KiB of LDS");
      return -4;
    }
    if (int rc = launch_blocks(a.cube_rec_bytes == 64 ? k->matrix_cube_narrow : k->matrix_cube_wide, a.plan.num_blocks,
                               k->cube_threads, lds, a, a.stream))
      return rc;
  }
  else if (alg == MPCX_ALG_ROWBLOCK && a.n_entities > 0 && a.plan.row_pairs == 2)
  {
    // pair records (mpcx_pair_records; no cached contexts): the row-wise copies of the text, one pair per lane
    if (!k->rowwise || !k->matrix_pairs)
    {
      mpcx_set_error("mpcx_assemble_matrix (UFCx, pair records): this kernel was not compiled with row-wise copies (element tensors of "
                     "37 .. 900 entries on simplices of up to ten nodes, no dof transformations; mpcx_ufcx_rowwise tells)");
      return -8;
    }
    if (a.plan.num_blocks <= 0 || !a.pair_recs || !a.plan.block_row0 || !a.plan.block_ent_off || a.estride != 1 || a.pair_dict
        || !a.mdofmap1)
    {
      mpcx_set_error("mpcx_assemble_matrix (UFCx, pair records): needs full records (mpcx_pair_records, no dictionary), the row "
                     "blocks, the column-masked dofmap (mdofmap1) and a cell integral");
      return -5;
    }
    const size_t lds = size_t(a.plan.max_nnz) * 8 + (a.bs0 > 1 ? size_t(a.plan.max_rows + 1) * 4 : 0);
    if (lds > 160 * 1024)
    {
      mpcx_set_error("mpcx_assemble_matrix: row-block plan exceeds 160 KiB of LDS");
      return -4;
    }
    if (int rc = launch_blocks(k->matrix_pairs, a.plan.num_blocks, k->rb_threads, lds, a, a.stream))
      return rc;
  }
  else if (alg == MPCX_ALG_ROWBLOCK && a.n_entities > 0)
  {
    if (a.plan.num_blocks <= 0 || !a.mdofmap0 || !a.mdofmap1 || !a.plan.ent_offs || a.plan.row_pairs || a.plan.ent_pattern
        || a.lean || a.slot_mask)
    {
      mpcx_set_error("mpcx_assemble_matrix (UFCx, row blocks): needs a plain entity plan with its scatter-offset table "
                     "(mpcx_scatter_offsets, rotate = 0) and the masked dofmaps (mpcx_mask_dofmap, rotate = 0)");
      return -5;
    }
    const size_t lds = size_t(a.plan.max_nnz) * 8 + size_t(a.plan.max_rows + 1) * 4;
    if (lds > 160 * 1024)
    {
      mpcx_set_error("mpcx_assemble_matrix: row-block plan exceeds 160 KiB of LDS");
      return -4;
    }
    if (int rc = launch_blocks(k->matrix_rowblock, a.plan.num_blocks, k->rb_threads, lds, a, a.stream))
      return rc;
  }
  else if (int rc = launch(k->matrix, a.n_entities, a, a.stream))
    return rc;
  if (a.n_slave_entities > 0 && a.mpc_plan_off)
  {
    if (a.mpc_plan_targets <= 0)
      return 0;
    const int g = a.mpc_plan_group >= 16 ? 16 : (a.mpc_plan_group >= 4 ? 4 : 1);
    if (a.slave_tensors && a.mpc_plan_slot)
    {
      if (int rc = launch(k->slave_tensors, a.n_slave_entities, a, a.stream))
        return rc;
      return launch(k->matrix_mpc_gather, a.mpc_plan_targets * g, a, a.stream);
    }
    return launch(k->matrix_mpc_plan, a.mpc_plan_targets * g, a, a.stream);
  }
  return launch(k->matrix_mpc, a.n_slave_entities, a, a.stream);
}
int launch_vector_ufcx(const mpcx_vector_args_t& a)
{
  auto* k = static_cast<UfcxKernel*>(const_cast<void*>(a.kernel.ufcx));
  if (!k || k->desc.rank != 1 || a.nd != k->desc.nd0 || a.bs != k->desc.bs0 || a.nv != k->desc.nv)
  {
    mpcx_set_error("mpcx_assemble_vector: the imported kernel was compiled for another element shape (or is not linear)");
    return -12;
  }
  if (k->t0 && !a.cell_info0)
  {
    mpcx_set_error("mpcx_assemble_vector: the imported kernel was compiled with a dof transformation: cell_info0 needed");
    return -5;
  }
  if (int rc = ensure_loaded(k))
    return rc;
  int alg = a.algorithm;
  if (alg == MPCX_ALG_AUTO)
    alg = a.plan.num_blocks > 0 ? MPCX_ALG_ROWBLOCK : MPCX_ALG_ATOMIC;
  if (alg == MPCX_ALG_CUBE)
  {
    // one thread per cluster, owner-computes row blocks (the plan of vector_cube_own_kernel), six calls of the imported
    // function per cluster; then the halo rows and the rows of slave dofs (ufcx_vector_mpc_kernel over slave_entities)
    if (!k->cube)
    {
      mpcx_set_error("mpcx_assemble_vector: imported (UFCx) kernels take MPCX_ALG_CUBE for scalar P1 forms on tetrahedra only");
      return -3;
    }
    if (a.n_cubes == 0)
      return launch(k->vector_mpc, a.n_slave_entities, a, a.stream);
    if (!a.cube_verts || !a.own_lmap || a.plan.num_blocks <= 0 || !a.plan.block_ents || !a.own_hoff || !a.own_spill || !a.own_seg
        || (a.n_own_rows > 0 && (!a.own_rows || !a.own_src)) || (a.coeffs && !a.cube_cells))
    {
      mpcx_set_error("mpcx_assemble_vector (UFCx, clusters): needs cube_verts and the owner-computes plan over the clusters "
                     "(own_lmap ...), and cube_cells for a form with coefficients");
      return -5;
    }
    const size_t lds = size_t(a.plan.max_rows) * 8;
    if (lds > 96 * 1024)
    {
      mpcx_set_error("mpcx_assemble_vector: cluster owner plan exceeds the LDS budget");
      return -4;
    }
    if (int rc = launch_blocks(k->vector_cube_own, a.plan.num_blocks, k->cube_threads, lds, a, a.stream))
      return rc;
    if (a.n_own_rows > 0)
      if (int rc = launch_vector_spill_reduce(a, 1))
        return rc;
    return launch(k->vector_mpc, a.n_slave_entities, a, a.stream);
  }
  if (alg == MPCX_ALG_ROWBLOCK && a.n_entities > 0)
  {
    const bool owner = a.own_lmap != nullptr;
    if (a.plan.num_blocks <= 0 || (!a.mdofmap && !owner))
    {
      mpcx_set_error("mpcx_assemble_vector (UFCx, row blocks): needs a plan and the slave-masked dofmap");
      return -3;
    }
    if (owner && (!a.own_hoff || !a.own_spill || !a.own_seg || (a.n_own_rows > 0 && (!a.own_rows || !a.own_src))))
    {
      mpcx_set_error("mpcx_assemble_vector: incomplete owner-computes plan");
      return -5;
    }
    const size_t lds = size_t(a.plan.max_rows) * 8;
    if (lds > 96 * 1024)
    {
      mpcx_set_error("mpcx_assemble_vector: row-block plan exceeds the LDS budget");
      return -4;
    }
    if (int rc = launch_blocks(k->vector_rowblock, a.plan.num_blocks, k->rb_threads, lds, a, a.stream))
      return rc;
    if (owner && a.n_own_rows > 0)
      if (int rc = launch_vector_spill_reduce(a, k->desc.bs0))
        return rc;
    return launch(k->vector_mpc, a.n_slave_entities, a, a.stream);
  }
  if (alg == MPCX_ALG_ROWBLOCK)
    // no bulk entities: only the slave rows of a.slave_entities go to their masters (the bulk was assembled by another
    // call that skipped them, e.g. the built-in hexahedron kernel of MPCX_ALG_CUBE)
    return launch(k->vector_mpc, a.n_slave_entities, a, a.stream);
  return launch(k->vector, a.n_entities, a, a.stream);
}
int launch_lifting_ufcx(const mpcx_lifting_args_t& a)
{
  auto* k = static_cast<UfcxKernel*>(const_cast<void*>(a.kernel.ufcx));
  if (!k || k->desc.rank != 2 || a.nd0 != k->desc.nd0 || a.bs0 != k->desc.bs0 || a.nd1 != k->desc.nd1 || a.bs1 != k->desc.bs1
      || a.nv != k->desc.nv)
  {
    mpcx_set_error("mpcx_apply_lifting: the imported kernel was compiled for other element shapes (or is not bilinear)");
    return -12;
  }
  if ((k->t0 && !a.cell_info0) || (k->t1 && !a.cell_info1))
  {
    mpcx_set_error("mpcx_apply_lifting: the imported kernel was compiled with dof transformations: cell_info0 / cell_info1 needed");
    return -5;
  }
  if (int rc = ensure_loaded(k))
    return rc;
  return launch(k->lifting, a.n_lift_entities, a, a.stream, k->big ? k->scratch_threads : 0);
}
} // namespace mpcx
