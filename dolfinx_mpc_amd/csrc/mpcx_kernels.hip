// HIP kernels (gfx950 / CDNA4) for constrained finite-element assembly.
//
// What the reference does serially per MPI rank -- cpp/assemble_matrix.cpp:488
// (cells), :343 (exterior facets), cpp/assemble_vector.cpp:65,
// cpp/lifting.h:77 -- is done here with one thread per integration entity:
//   gather coordinates -> element tensor in VGPRs -> Dirichlet mask ->
//   slave mask -> scatter-add into a pre-built CSR / vector.
// The K^T A_e K elimination of cpp/assemble_matrix.cpp:99-268 is split the way
// the reference's numba assembler splits it (numba/assemble_matrix.py:100):
// the bulk kernel handles every entity with slave rows/cols masked out, and a
// second kernel over the compact list of slave entities adds the master
// row/column contributions.  Test and trial spaces may differ (rectangular
// blocks, cpp/assemble_matrix.cpp:537-541).  All arithmetic is fp64; the path is
// HBM-bound (no dense contraction), so there is no MFMA here.
#include "mpcx.h"
#include "mpcx_elements.hpp"
#include "mpcx_internal.h"

#include <algorithm>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <string>
#include <type_traits>

namespace mpcx
{

__device__ inline void atomic_add_f64(double* p, double v)
{
  // hardware global_atomic_add_f64 (device scope): rows may be touched from any XCD
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// position of `col` in the sorted row [lo, hi) of the CSR, or -1
__device__ inline int64_t csr_find(const int32_t* __restrict__ cols, int64_t lo, int64_t hi, int col)
{
  const int64_t end = hi;
  while (lo < hi)
  {
    const int64_t mid = (lo + hi) >> 1;
    if (cols[mid] < col)
      lo = mid + 1;
    else
      hi = mid;
  }
  return (lo < end && cols[lo] == col) ? lo : -1;
}

template <int NV>
__device__ inline void gather_coords(const double* __restrict__ x, const int32_t* __restrict__ x_dofmap,
                                     int64_t cell, double (&cd)[NV * 3])
{
#pragma unroll
  for (int i = 0; i < NV; ++i)
  {
    const int64_t v = x_dofmap[cell * NV + i];
#pragma unroll
    for (int k = 0; k < 3; ++k)
      cd[3 * i + k] = x[3 * v + k];
  }
}

// ---------------------------------------------------------------------------
// Bulk matrix kernel, atomic scatter.  cpp/assemble_matrix.cpp:488-547 with the
// slave rows/cols of Ae zeroed (:165-178) for every entity.
// ---------------------------------------------------------------------------
template <class Op>
__global__ void __launch_bounds__(256) matrix_atomic_kernel(mpcx_matrix_args_t a)
{
  constexpr int N0 = Op::N0, N1 = Op::N1, ND0 = Op::ND0, ND1 = Op::ND1, BS0 = Op::BS0, BS1 = Op::BS1,
                NV = Op::NV;
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= a.n_entities)
    return;
  const int64_t l = e * a.estride;
  const int64_t cell = (a.entities ? a.entities[l] : e);
  const int64_t cell0 = (a.entities0 ? a.entities0[l] : e);
  const int64_t cell1 = (a.entities1 ? a.entities1[l] : e);
  const int lf = a.estride == 2 ? a.entities[l + 1] : 0;

  double cd[NV * 3];
  gather_coords<NV>(a.x, a.x_dofmap, cell, cd);
  double Ae[Op::SIZE];
  Op::tabulate(Ae, a.coeffs ? a.coeffs + e * a.cstride : nullptr, a.constants, cd, lf, a.kernel);

  int32_t rows[N0], colsd[N1];
  bool rmask[N0], cmask[N1];
#pragma unroll
  for (int i = 0; i < ND0; ++i)
  {
    const int32_t d0 = a.dofmap0[cell0 * ND0 + i];
#pragma unroll
    for (int k = 0; k < BS0; ++k)
    {
      const int32_t r = d0 * BS0 + k;
      rows[i * BS0 + k] = r;
      rmask[i * BS0 + k] = (a.bc0 && a.bc0[r]) || a.mpc0.is_slave[r];
    }
  }
#pragma unroll
  for (int j = 0; j < ND1; ++j)
  {
    const int32_t d1 = a.dofmap1[cell1 * ND1 + j];
#pragma unroll
    for (int k = 0; k < BS1; ++k)
    {
      const int32_t c = d1 * BS1 + k;
      colsd[j * BS1 + k] = c;
      cmask[j * BS1 + k] = (a.bc1 && a.bc1[c]) || a.mpc1.is_slave[c];
    }
  }
#pragma unroll
  for (int p = 0; p < N0; ++p)
  {
    if (rmask[p])
      continue;
    const int64_t lo = a.rowptr[rows[p]], hi = a.rowptr[rows[p] + 1];
#pragma unroll
    for (int q = 0; q < N1; ++q)
    {
      if (cmask[q])
        continue;
      if constexpr (Op::DIAG)
      {
        if ((p % BS0) != (q % BS1))
          continue; // structurally zero entry of a component-diagonal form
      }
      const int64_t pos = csr_find(a.cols, lo, hi, colsd[q]);
      if (pos >= 0)
        atomic_add_f64(a.vals + MPCX_VAL_POS(a, pos), Op::get(Ae, p, q));
    }
  }
}

// ---------------------------------------------------------------------------
// Master contributions of slave entities: cpp/assemble_matrix.cpp:182-267.
// One thread per slave entity; Ae is indexed dynamically (scratch), which is
// fine for the <1 % of entities that reach this kernel.
// ---------------------------------------------------------------------------
template <class Op>
__global__ void __launch_bounds__(64) matrix_mpc_kernel(mpcx_matrix_args_t a)
{
  constexpr int N0 = Op::N0, N1 = Op::N1, ND0 = Op::ND0, ND1 = Op::ND1, BS0 = Op::BS0, BS1 = Op::BS1,
                NV = Op::NV;
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= a.n_slave_entities)
    return;
  const int64_t e = a.slave_entities[t];
  const int64_t l = e * a.estride;
  const int64_t cell = (a.entities ? a.entities[l] : e);
  const int64_t cell0 = (a.entities0 ? a.entities0[l] : e);
  const int64_t cell1 = (a.entities1 ? a.entities1[l] : e);
  const int lf = a.estride == 2 ? a.entities[l + 1] : 0;

  double cd[NV * 3];
  gather_coords<NV>(a.x, a.x_dofmap, cell, cd);
  double Ae[Op::SIZE];
  Op::tabulate(Ae, a.coeffs ? a.coeffs + e * a.cstride : nullptr, a.constants, cd, lf, a.kernel);

  int32_t rows[N0], colsd[N1];
  bool rbc[N0], cbc[N1], rsl[N0], csl[N1];
  for (int i = 0; i < ND0; ++i)
  {
    const int32_t d0 = a.dofmap0[cell0 * ND0 + i];
    for (int k = 0; k < BS0; ++k)
    {
      const int32_t r = d0 * BS0 + k;
      rows[i * BS0 + k] = r;
      rbc[i * BS0 + k] = a.bc0 && a.bc0[r];
      rsl[i * BS0 + k] = a.mpc0.is_slave[r];
    }
  }
  for (int j = 0; j < ND1; ++j)
  {
    const int32_t d1 = a.dofmap1[cell1 * ND1 + j];
    for (int k = 0; k < BS1; ++k)
    {
      const int32_t c = d1 * BS1 + k;
      colsd[j * BS1 + k] = c;
      cbc[j * BS1 + k] = a.bc1 && a.bc1[c];
      csl[j * BS1 + k] = a.mpc1.is_slave[c];
    }
  }
  // Dirichlet rows/cols are zeroed before the MPC modification (:510-533)
  auto entry = [&](int p, int q) -> double { return (rbc[p] || cbc[q]) ? 0.0 : Op::get(Ae, p, q); };

  // row masters (:214-246)
  for (int p = 0; p < N0; ++p)
  {
    if (!rsl[p])
      continue;
    for (int mi = a.mpc0.masters_offsets[rows[p]]; mi < a.mpc0.masters_offsets[rows[p] + 1]; ++mi)
    {
      const int32_t m = a.mpc0.masters[mi];
      const double ci = a.mpc0.coeffs[mi];
      const int64_t lo = a.rowptr[m], hi = a.rowptr[m + 1];
      for (int q = 0; q < N1; ++q)
      {
        const double v = entry(p, q);
        if (csl[q])
        {
          // master-master term uses the un-stripped original (:239-245)
          for (int mj = a.mpc1.masters_offsets[colsd[q]]; mj < a.mpc1.masters_offsets[colsd[q] + 1]; ++mj)
          {
            const int64_t pos = csr_find(a.cols, lo, hi, a.mpc1.masters[mj]);
            if (pos >= 0)
              atomic_add_f64(a.vals + MPCX_VAL_POS(a, pos), ci * a.mpc1.coeffs[mj] * v);
          }
        }
        else if (!cbc[q])
        {
          // stripped row: slave-slave entries removed (:226-236)
          const int64_t pos = csr_find(a.cols, lo, hi, colsd[q]);
          if (pos >= 0)
            atomic_add_f64(a.vals + MPCX_VAL_POS(a, pos), ci * v);
        }
      }
    }
  }
  // column masters (:251-267)
  for (int q = 0; q < N1; ++q)
  {
    if (!csl[q])
      continue;
    for (int mj = a.mpc1.masters_offsets[colsd[q]]; mj < a.mpc1.masters_offsets[colsd[q] + 1]; ++mj)
    {
      const int32_t m = a.mpc1.masters[mj];
      const double cj = a.mpc1.coeffs[mj];
      for (int p = 0; p < N0; ++p)
      {
        if (rsl[p] || rbc[p])
          continue;
        const int64_t pos = csr_find(a.cols, a.rowptr[rows[p]], a.rowptr[rows[p] + 1], m);
        if (pos >= 0)
          atomic_add_f64(a.vals + MPCX_VAL_POS(a, pos), cj * entry(p, q));
      }
    }
  }
}

// Same contributions from the plan built once on the host (mpcx_mpc_plan_build), gathered by target:
// one thread per CSR position that receives master contributions sums coef * Ae[pq] over its tuples
// (re-tabulating the tuple's entity) and adds the sum with one plain read-modify-write -- no device
// atomics (4.7 M of them at config 2 took 0.2 ms) and no CSR searches.
template <class Op>
__global__ void __launch_bounds__(64) matrix_mpc_plan_small_kernel(mpcx_matrix_args_t a)
{
  // small element tensors (P1 scalar: 16 entries): every tuple simply tabulates its entity; the tuples of a
  // thread are independent, so their load chains overlap
  constexpr int NV = Op::NV, N1 = Op::N1;
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= a.mpc_plan_targets)
    return;
  double sum = 0.0;
  for (int64_t k = a.mpc_plan_off[t]; k < a.mpc_plan_off[t + 1]; ++k)
  {
    const int64_t e = a.mpc_plan_ent[k];
    const int64_t l = e * a.estride;
    const int64_t cell = (a.entities ? a.entities[l] : e);
    const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
    double cd[NV * 3];
    gather_coords<NV>(a.x, a.x_dofmap, cell, cd);
    double Ae[Op::SIZE];
    Op::tabulate(Ae, a.coeffs ? a.coeffs + e * a.cstride : nullptr, a.constants, cd, lf, a.kernel);
    const int pq = a.mpc_plan_pq[k];
    sum += a.mpc_plan_coef[k] * Op::get(Ae, pq / N1, pq % N1);
  }
  if (a.mpc_plan_out)
    a.mpc_plan_out[t] += sum; // block-scalar storage: the couplings live in an overlay, one entry per target
  else
    a.vals[MPCX_VAL_POS(a, a.mpc_plan_tgt[t])] += sum;
}

template <class Op, int G, bool USE_LAZY> // G lanes share one target position
__global__ void __launch_bounds__(64) matrix_mpc_plan_kernel(mpcx_matrix_args_t a)
{
  // Larger element tensors (vector P1: 144 entries, P2: 100-900): a group of G lanes takes one target position
  // and strides over its tuples (17 on average, up to ~150 for the contact benchmark: one thread per target left
  // the longest list on the critical path; 3-4 for slip walls, where G = 4: the caller passes G = the average list
  // length rounded to 1 / 4 / 16), each lane evaluating its entry straight from the compact per-cell
  // context where the operator has one (USE_LAZY, Op::prepare / Op::entry: a few dozen flops instead of a full
  // element tensor in scratch memory); the partial sums are combined with a shuffle reduction.
  // The entry index is a run-time value here, so the context is indexed dynamically: it lives in LDS (one slot
  // per lane).  Left in a local array it went to scratch memory -- 104 bytes written to HBM per tuple, 4.8 GB
  // per launch on the contact benchmark, which made this kernel HBM-bound on its own spills -- and sharing a
  // kernel with the full-tensor branch cost the lazy one the registers of both (369 VGPRs, one wave per SIMD).
  constexpr int NV = Op::NV, N1 = Op::N1, BS0 = Op::BS0, BS1 = Op::BS1;
  const int lane = threadIdx.x & (G - 1);
  const int64_t t = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  if (t >= a.mpc_plan_targets)
    return; // the whole group leaves together
  [[maybe_unused]] __shared__ typename Op::Lazy s_lz[USE_LAZY ? 64 : 1];
  double sum = 0.0;
  for (int64_t k = a.mpc_plan_off[t] + lane; k < a.mpc_plan_off[t + 1]; k += G)
  {
    const int64_t e = a.mpc_plan_ent[k];
    const int64_t l = e * a.estride;
    const int64_t cell = (a.entities ? a.entities[l] : e);
    double cd[NV * 3];
    gather_coords<NV>(a.x, a.x_dofmap, cell, cd);
    const int pq = a.mpc_plan_pq[k];
    const int p = pq / N1, q = pq % N1;
    double v = 0.0;
    if constexpr (USE_LAZY)
    {
      {
        typename Op::Lazy lz;
        Op::prepare(lz, a.constants, cd);
        s_lz[threadIdx.x] = lz;
      }
      v = Op::entry(s_lz[threadIdx.x], p / BS0, p % BS0, q / BS1, q % BS1);
    }
    else
    {
      const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
      double Ae[Op::SIZE];
      Op::tabulate(Ae, a.coeffs ? a.coeffs + e * a.cstride : nullptr, a.constants, cd, lf, a.kernel);
      v = Op::get(Ae, p, q);
    }
    sum += a.mpc_plan_coef[k] * v;
  }
  if constexpr (G > 1)
  {
#pragma unroll
    for (int m = G / 2; m > 0; m >>= 1)
      sum += __shfl_xor(sum, m, G);
  }
  if (lane == 0)
  {
    if (a.mpc_plan_out)
      a.mpc_plan_out[t] += sum;
    else
      a.vals[MPCX_VAL_POS(a, a.mpc_plan_tgt[t])] += sum;
  }
}

// ---------------------------------------------------------------------------
// The plan itself on the device (host version: mpcx_mpc_plan_build).  One thread per
// slave entity walks the index logic of modify_mpc_cell (cpp/assemble_matrix.cpp:182-267) twice: a
// counting pass, then -- after the caller's scan -- a pass that writes the tuples
// (position in vals, entity, tensor entry p*N1+q, coefficient).  The caller sorts them by position
// (stable, so a target's tuples stay ordered by entity).  Entries of Dirichlet rows / columns are zero
// (:510-533) and emit nothing; neither do the structural zeros of component-diagonal forms (diag != 0).
// ---------------------------------------------------------------------------
constexpr int MPC_PLAN_MAXN = 32; // unrolled dofs per element side (P2 vector tets: 30)

template <bool FILL>
__global__ void __launch_bounds__(64)
mpc_plan_device_kernel(int64_t n_slave_entities, const int32_t* __restrict__ slave_entities, int estride,
                       const int32_t* __restrict__ entities0, const int32_t* __restrict__ entities1,
                       const int32_t* __restrict__ dofmap0, int nd0, int bs0, const int32_t* __restrict__ dofmap1,
                       int nd1, int bs1, const int8_t* __restrict__ bc0, const int8_t* __restrict__ bc1, mpcx_mpc_t mpc0,
                       mpcx_mpc_t mpc1, const mpcx_nnz_t* __restrict__ rowptr, const int32_t* __restrict__ cols, int diag,
                       int64_t* __restrict__ counts, const int64_t* __restrict__ offsets, int64_t* __restrict__ out_pos,
                       int32_t* __restrict__ out_ent, int32_t* __restrict__ out_pq, double* __restrict__ out_coef)
{
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n_slave_entities)
    return;
  const int64_t e = slave_entities[t];
  const int64_t cell0 = entities0 ? entities0[e * estride] : e;
  const int64_t cell1 = entities1 ? entities1[e * estride] : e;
  const int N0 = nd0 * bs0, N1 = nd1 * bs1;
  int32_t rows[MPC_PLAN_MAXN], colsd[MPC_PLAN_MAXN];
  uint32_t rbc = 0, cbc = 0, rsl = 0, csl = 0; // one bit per unrolled local dof
  for (int i = 0; i < nd0; ++i)
    for (int k = 0; k < bs0; ++k)
    {
      const int32_t r = dofmap0[cell0 * nd0 + i] * bs0 + k;
      rows[i * bs0 + k] = r;
      rbc |= uint32_t(bc0 && bc0[r]) << (i * bs0 + k);
      rsl |= uint32_t(mpc0.is_slave[r] != 0) << (i * bs0 + k);
    }
  for (int j = 0; j < nd1; ++j)
    for (int k = 0; k < bs1; ++k)
    {
      const int32_t c = dofmap1[cell1 * nd1 + j] * bs1 + k;
      colsd[j * bs1 + k] = c;
      cbc |= uint32_t(bc1 && bc1[c]) << (j * bs1 + k);
      csl |= uint32_t(mpc1.is_slave[c] != 0) << (j * bs1 + k);
    }
  int64_t n = 0;
  const int64_t base = FILL ? offsets[t] : 0;
  auto emit = [&](int p, int q, int64_t lo, int64_t hi, int32_t col, double c)
  {
    if (((rbc >> p) & 1) || ((cbc >> q) & 1) || (diag && (p % bs0) != (q % bs1)))
      return;
    if constexpr (FILL)
    {
      out_pos[base + n] = csr_find(cols, lo, hi, col); // -1 (outside the pattern) is dropped by the caller
      out_ent[base + n] = int32_t(e);
      out_pq[base + n] = p * N1 + q;
      out_coef[base + n] = c;
    }
    ++n;
  };
  // row masters (:214-246)
  for (int p = 0; p < N0; ++p)
  {
    if (!((rsl >> p) & 1))
      continue;
    for (int mi = mpc0.masters_offsets[rows[p]]; mi < mpc0.masters_offsets[rows[p] + 1]; ++mi)
    {
      const int32_t m = mpc0.masters[mi];
      const double ci = mpc0.coeffs[mi];
      const int64_t lo = rowptr[m], hi = rowptr[m + 1];
      for (int q = 0; q < N1; ++q)
      {
        if ((csl >> q) & 1)
        {
          // master-master term from the un-stripped tensor (:239-245)
          for (int mj = mpc1.masters_offsets[colsd[q]]; mj < mpc1.masters_offsets[colsd[q] + 1]; ++mj)
            emit(p, q, lo, hi, mpc1.masters[mj], ci * mpc1.coeffs[mj]);
        }
        else
          emit(p, q, lo, hi, colsd[q], ci); // stripped row (:226-236)
      }
    }
  }
  // column masters (:251-267)
  for (int q = 0; q < N1; ++q)
  {
    if (!((csl >> q) & 1))
      continue;
    for (int mj = mpc1.masters_offsets[colsd[q]]; mj < mpc1.masters_offsets[colsd[q] + 1]; ++mj)
    {
      const int32_t m = mpc1.masters[mj];
      const double cj = mpc1.coeffs[mj];
      for (int p = 0; p < N0; ++p)
      {
        if ((rsl >> p) & 1)
          continue;
        emit(p, q, rowptr[rows[p]], rowptr[rows[p] + 1], m, cj);
      }
    }
  }
  if constexpr (!FILL)
    counts[t] = n;
}

// ---------------------------------------------------------------------------
// Bulk matrix kernel, LDS-privatised row blocks.  One workgroup owns a
// contiguous range of CSR rows whose values live in LDS; every entity touching
// the block is evaluated (redundantly across blocks), only rows inside the
// block are kept, and the finished values are written to HBM once, coalesced.
// No device atomics, no searching:
//  * the position of every (local row, local col) pair inside its CSR row is a
//    precomputed 8-bit offset (plan.ent_offs, ND0*ND1 bytes per entity, read
//    coalesced) -- the same role PETSc's per-row column search plays behind
//    MatSetValuesLocal in the reference, hoisted to set-up;
//  * the Dirichlet/slave mask arrives folded into the dofmap (bit 28+k of the
//    blocked dof = "row/col of component k is masked"), so there are no marker
//    gathers either.
// ---------------------------------------------------------------------------
constexpr int MPCX_MASK_SHIFT = 28;
constexpr int MPCX_DOF_MASK = (1 << MPCX_MASK_SHIFT) - 1;
// launch bound 1024 caps the kernel at 128 VGPRs = 4 waves/SIMD; measured faster
// (2.25 ms) than the spill-free 150-VGPR build at 3 waves/SIMD (2.56 ms).
constexpr int ROWBLOCK_MAX_THREADS = 1024;

// LEAN: square form on one space whose dofmap IS the geometry dofmap (P1 on an affine mesh: the
// host passes one device array for both), integral over all cells (entities == NULL), no facets:
// per entity only the masked dofmap row and the scatter offsets are read, and the thread keeps
// three entities in flight.  Everything else takes the general path.
// USE_LAZY: entries are evaluated where they are scattered (Op::prepare / Op::entry) instead of
// holding the element tensor in registers -- P1 elasticity (144 entries) and P2 stiffness (100)
// do not fit 128 VGPRs: 42.7 -> 3.7 ms for elasticity on 128^3.
template <class Op, bool LEAN, bool USE_LAZY>
__global__ void __launch_bounds__(ROWBLOCK_MAX_THREADS) matrix_rowblock_kernel(mpcx_matrix_args_t a)
{
  constexpr int ND0 = Op::ND0, ND1 = Op::ND1, BS0 = Op::BS0, BS1 = Op::BS1, NV = Op::NV;
  constexpr int NOFF = ND0 * ND1;
  const int NT = blockDim.x;
  extern __shared__ __align__(16) unsigned char smem[];
  // XCD-aware order: workgroup w runs on XCD w % 8 (observed placement, speed
  // only); give each XCD a contiguous run of row blocks so neighbouring blocks
  // (shared halo cells / nodes) share an L2.
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (b >= nb)
    return;
  const int tid = threadIdx.x;
  const int r0 = a.plan.block_row0[b], r1 = a.plan.block_row0[b + 1];
  const int nrow = r1 - r0;
  const int64_t nnz0 = a.rowptr[r0];
  const int nnzb = int(a.rowptr[r1] - nnz0);
  // Component-diagonal forms on blocked spaces (S (x) I: vector stiffness / mass, Stokes a00) only ever touch the
  // (k, k) entry of a BS x BS block: LDS keeps ONE value per column block of every scalar row (CW = BS1 times fewer
  // entries, so a workgroup owns BS1 times more rows and the halo shrinks) and the structural zeros are produced
  // when the block is written out.
  constexpr int CW = Op::DIAG ? BS1 : 1;
  double* s_vals = reinterpret_cast<double*>(smem);                             // [max_nnz / CW]
  int32_t* s_rowlo = reinterpret_cast<int32_t*>(s_vals + a.plan.max_nnz / CW); // [max_rows + 1]

  for (int i = tid; i < nnzb / CW; i += NT)
    s_vals[i] = 0.0;
  for (int rl = tid; rl <= nrow; rl += NT)
    s_rowlo[rl] = int(a.rowptr[r0 + rl] - nnz0) / CW;
  __syncthreads();

  // per-entity index data: everything that is read through the entity index
  struct Ent
  {
    int32_t e;
    int lf;
    int32_t xd[NV];
    int32_t m0[ND0], m1[ND1];
    uint32_t ow[(NOFF + 3) / 4]; // scatter offsets, 4 per word (unpacked at use: v_bfe_u32)
  };
  auto load_ent = [&](int64_t e, Ent& E)
  {
    E.e = e;
    if constexpr (LEAN)
    {
#pragma unroll
      for (int i = 0; i < ND0; ++i)
        E.m0[i] = a.mdofmap0[e * ND0 + i];
    }
    else
    {
      const int64_t l = e * a.estride;
      const int64_t cell = (a.entities ? a.entities[l] : e);
      const int64_t cell0 = (a.entities0 ? a.entities0[l] : e);
      const int64_t cell1 = (a.entities1 ? a.entities1[l] : e);
      E.lf = a.estride == 2 ? a.entities[l + 1] : 0;
#pragma unroll
      for (int i = 0; i < ND0; ++i)
        E.m0[i] = a.mdofmap0[cell0 * ND0 + i];
#pragma unroll
      for (int j = 0; j < ND1; ++j)
        E.m1[j] = a.mdofmap1[cell1 * ND1 + j];
#pragma unroll
      for (int i = 0; i < NV; ++i)
        E.xd[i] = a.x_dofmap[cell * NV + i];
    }
    // scatter offsets of this entity (ND0*ND1 bytes, contiguous): its own row of the
    // table, or -- dictionary-compressed plan -- the shared row its 2-byte pattern id selects
    const uint8_t* po = a.plan.ent_offs + (a.plan.ent_pattern ? int64_t(a.plan.ent_pattern[e]) : e) * NOFF;
    if constexpr (NOFF % 16 == 0)
    {
#pragma unroll
      for (int w = 0; w < NOFF / 16; ++w)
      {
        const uint4 v = reinterpret_cast<const uint4*>(po)[w];
        E.ow[4 * w] = v.x;
        E.ow[4 * w + 1] = v.y;
        E.ow[4 * w + 2] = v.z;
        E.ow[4 * w + 3] = v.w;
      }
    }
    else if constexpr (NOFF % 4 == 0)
    {
#pragma unroll
      for (int w = 0; w < NOFF / 4; ++w)
        E.ow[w] = reinterpret_cast<const uint32_t*>(po)[w];
    }
    else
    {
#pragma unroll
      for (int w = 0; w < (NOFF + 3) / 4; ++w)
      {
        uint32_t u = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (4 * w + q < NOFF)
            u |= uint32_t(po[4 * w + q]) << (8 * q);
        E.ow[w] = u;
      }
    }
  };
  auto load_coords = [&](const Ent& E, double (&cd)[NV * 3])
  {
#pragma unroll
    for (int i = 0; i < NV; ++i)
    {
      int64_t v;
      if constexpr (LEAN)
        v = E.m0[i < ND0 ? i : 0] & MPCX_DOF_MASK;
      else
        v = E.xd[i];
#pragma unroll
      for (int k = 0; k < 3; ++k)
        cd[3 * i + k] = a.x[3 * v + k];
    }
  };
  // element tensor of one entity, rows of this block added into the LDS copy of the block
  auto accumulate = [&](const Ent& E, const double (&cd)[NV * 3])
  {
    // the element tensor in registers, or (LAZY operators) the compact context its entries come from
    constexpr bool LAZY = USE_LAZY;
    double Ae[LAZY ? 1 : Op::SIZE];
    typename Op::Lazy lz;
    if constexpr (LAZY)
      Op::prepare(lz, a.constants, cd);
    else
      Op::tabulate(Ae, a.coeffs ? a.coeffs + int64_t(E.e) * a.cstride : nullptr, a.constants, cd, LEAN ? 0 : E.lf, a.kernel);
#pragma unroll
    for (int i = 0; i < ND0; ++i)
    {
#pragma unroll
      for (int k = 0; k < BS0; ++k)
      {
        const int r = (E.m0[i] & MPCX_DOF_MASK) * BS0 + k;
        if (r < r0 || r >= r1 || ((E.m0[i] >> (MPCX_MASK_SHIFT + k)) & 1))
          continue;
        const int base = s_rowlo[r - r0];
#pragma unroll
        for (int j = 0; j < ND1; ++j)
        {
          const int off = int((E.ow[(i * ND1 + j) >> 2] >> (8 * ((i * ND1 + j) & 3))) & 0xff) * (BS1 / CW);
#pragma unroll
          for (int q = 0; q < BS1; ++q)
          {
            if constexpr (Op::DIAG)
            {
              if (k != q)
                continue; // structurally zero
            }
            if (((LEAN ? E.m0[j < ND0 ? j : 0] : E.m1[j]) >> (MPCX_MASK_SHIFT + q)) & 1)
              continue;
            double v;
            if constexpr (LAZY)
              v = Op::entry(lz, i, k, j, q);
            else
              v = Op::get(Ae, i * BS0 + k, j * BS1 + q);
            __hip_atomic_fetch_add(s_vals + base + off + (Op::DIAG ? 0 : q), v, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
  };

  const int64_t e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  const int32_t* __restrict__ ents = a.plan.block_ents;
  int64_t t = e0 + tid;
  if constexpr (NOFF <= 16)
  {
    // Software pipeline (small elements): while one entity is computed, the index data (masked dofmap row, scatter
    // offsets) of the next one and the entity index of the one after it are in flight, so the only
    // round trip an iteration waits for is its own coordinate gather.  (A third stage that also
    // prefetched the next coordinates needs > 128 VGPRs and spills: 7.5 ms instead of 2.0 ms.)
    Ent cur;
    int32_t i1 = 0;
    if (t < e1)
      load_ent(ents[t], cur);
    if (t + NT < e1)
      i1 = ents[t + NT];
    for (; t < e1; t += NT)
    {
      double cd[NV * 3];
      load_coords(cur, cd);
      Ent nxt = cur;
      if (t + NT < e1)
        load_ent(i1, nxt);
      if (t + 2 * NT < e1)
        i1 = ents[t + 2 * NT];
      accumulate(cur, cd);
      cur = nxt;
    }
  }
  else
  {
    for (; t < e1; t += NT)
    {
      Ent cur;
      double cd[NV * 3];
      load_ent(ents[t], cur);
      load_coords(cur, cd);
      accumulate(cur, cd);
    }
  }
  __syncthreads();
  // one coalesced write of the finished block
  if constexpr (Op::DIAG)
  {
    // expand: scalar row r = (node, k) holds (k, k) of every column block; a wave takes whole rows (their bounds
    // from the LDS copy: a dependent global load per row left the waves waiting on rowptr)
    const int wave = tid >> 6, lane = tid & 63, nwaves = NT >> 6;
    for (int rl = wave; rl < nrow; rl += nwaves)
    {
      const int k = (r0 + rl) % BS0;
      const int lo = s_rowlo[rl];
      const int64_t p0 = nnz0 + int64_t(lo) * CW;
      const int len = (s_rowlo[rl + 1] - lo) * CW;
      const double* src = s_vals + lo;
      for (int e = lane; e < len; e += 64)
      {
        const int q = e % BS1;
        const double v = (q == k) ? src[e / BS1] : 0.0;
        if (a.store_mode)
          a.vals[MPCX_VAL_POS(a, p0 + e)] = v;
        else if (q == k)
          a.vals[MPCX_VAL_POS(a, p0 + e)] += v;
      }
    }
  }
  else
    MPCX_WRITE_OUT(a, nnz0, nnzb, s_vals, tid, NT);
}

// Node-block variant for component-diagonal forms on blocked spaces (S (x) I: vector stiffness / mass, the
// Taylor-Hood velocity block), used when the caller passes slot_mask.  The bs component rows of a node hold the
// same scalar value in their (k, k) entries unless a Dirichlet / slave mask zeroes one of them, so LDS keeps ONE
// value per (row node, column node) slot: ND0 * ND1 scatter-adds per entity instead of bs times that, a workgroup
// owns bs^2 times more nodes for the same LDS than with the scalar layout (bs times more than with the compact
// per-row layout above) and the halo shrinks accordingly.  The bs x bs blocks, their structural zeros and the
// masked entries are produced when the block is written out: slot_mask[slot] bit k = "entry (k, k) of this
// block is zero" (mpcx_diag_slot_mask).  The masked dofmaps are read for their dof ids only.
// LDS of a node-block launch that expands to scalar CSR values: the block's values, its row offsets and -- when it fits --
// one mask byte per slot (host and device agree through this function)
__host__ __device__ inline bool nodeblock_stage_mask(int64_t max_nnz, int64_t max_rows, int bs)
{
  const int64_t slots = max_nnz / (int64_t(bs) * bs);
  return slots * 8 + (max_rows / bs + 1) * 4 + ((slots + 3) & ~int64_t(3)) <= 160 * 1024;
}
// (A/B switch of the node-block write-out, read once: the argument block has no spare field for an experiment)
__device__ __constant__ int g_nodeblock_narrow_stores = 0;
__device__ inline bool nodeblock_narrow_stores() { return g_nodeblock_narrow_stores == 1; }

template <class Op, bool USE_LAZY>
__device__ __forceinline__ void nodeblock_body(const mpcx_matrix_args_t& a)
{
  constexpr int ND0 = Op::ND0, ND1 = Op::ND1, BS = Op::BS0, NV = Op::NV;
  constexpr int NOFF = ND0 * ND1;
  static_assert(Op::DIAG && Op::BS0 == Op::BS1, "component-diagonal operator expected");
  const int NT = blockDim.x;
  extern __shared__ __align__(16) unsigned char smem[];
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (b >= nb)
    return;
  const int tid = threadIdx.x;
  const int r0 = a.plan.block_row0[b], r1 = a.plan.block_row0[b + 1];
  const int n0 = r0 / BS, nn = (r1 - r0) / BS; // node rows of the block
  const int64_t nnz0 = a.rowptr[r0];
  const int slots = int((a.rowptr[r1] - nnz0) / (BS * BS));
  double* s_vals = reinterpret_cast<double*>(smem);                                     // [max_nnz / BS^2]
  int32_t* s_rowlo = reinterpret_cast<int32_t*>(s_vals + a.plan.max_nnz / (BS * BS)); // [max_rows / BS + 1]
  for (int i = tid; i < slots; i += NT)
    s_vals[i] = 0.0;
  for (int nl = tid; nl <= nn; nl += NT)
    s_rowlo[nl] = int((a.rowptr[int64_t(n0 + nl) * BS] - nnz0) / (BS * BS));
  const int64_t gslot0 = nnz0 / (BS * BS);
  int any = 0;
  // The write-out reads the masks from LDS when the copy fits (nodeblock_stage_mask).  Their loads are issued HERE and
  // consumed after the entity loop (MW words per thread stay in registers): with one resident workgroup per CU nothing hides
  // a load that the next barrier waits for, and a block has one trip of the entity loop to hide it behind.
  uint8_t* s_mask = reinterpret_cast<uint8_t*>(s_rowlo + a.plan.max_rows / BS + 1); // [max_nnz / BS^2, rounded up to 4]
  const bool stage_mask = nodeblock_stage_mask(a.plan.max_nnz, a.plan.max_rows, BS);
  constexpr int MW = 4;
  const bool late_mask = !a.block_vals && stage_mask && slots <= 4 * MW * NT;
  uint32_t mw[MW] = {};
  if (late_mask)
  {
    const uint8_t* gm = a.slot_mask + gslot0;
#pragma unroll
    for (int j = 0; j < MW; ++j)
    {
      const int o = 4 * (tid + j * NT);
      if (o + 4 <= slots)
        __builtin_memcpy(&mw[j], gm + o, 4); // (any byte alignment: unaligned access mode)
      else
        for (int q = 0; o + q < slots; ++q)
          mw[j] |= uint32_t(gm[o + q]) << (8 * q);
    }
  }
  else if (!a.block_vals) // (block-scalar storage leaves the masks to the consumers)
    for (int i = tid; i < slots; i += NT)
    {
      const uint8_t m = a.slot_mask[gslot0 + i];
      any |= m;
      if (stage_mask)
        s_mask[i] = m;
    }
  __syncthreads();

  const int64_t e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  const int32_t* __restrict__ ents = a.plan.block_ents;
  for (int64_t t = e0 + tid; t < e1; t += NT)
  {
    const int64_t e = ents[t];
    const int64_t l = e * a.estride;
    const int64_t cell = (a.entities ? a.entities[l] : e);
    const int64_t cell0 = (a.entities0 ? a.entities0[l] : e);
    const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
    int32_t m0[ND0];
#pragma unroll
    for (int i = 0; i < ND0; ++i)
      m0[i] = a.mdofmap0[cell0 * ND0 + i] & MPCX_DOF_MASK;
    uint32_t ow[(NOFF + 3) / 4];
    const uint8_t* po = a.plan.ent_offs + e * NOFF;
    if constexpr (NOFF % 4 == 0)
    {
#pragma unroll
      for (int w = 0; w < NOFF / 4; ++w)
        ow[w] = reinterpret_cast<const uint32_t*>(po)[w];
    }
    else
    {
#pragma unroll
      for (int w = 0; w < (NOFF + 3) / 4; ++w)
      {
        uint32_t u = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (4 * w + q < NOFF)
            u |= uint32_t(po[4 * w + q]) << (8 * q);
        ow[w] = u;
      }
    }
    double cd[NV * 3];
    gather_coords<NV>(a.x, a.x_dofmap, cell, cd);
    double Ae[USE_LAZY ? 1 : Op::SIZE];
    typename Op::Lazy lz;
    if constexpr (USE_LAZY)
      Op::prepare(lz, a.constants, cd);
    else
      Op::tabulate(Ae, a.coeffs ? a.coeffs + e * a.cstride : nullptr, a.constants, cd, lf, a.kernel);
#pragma unroll
    for (int i = 0; i < ND0; ++i)
    {
      const int nl = m0[i] - n0;
      if (nl < 0 || nl >= nn)
        continue;
      const int base = s_rowlo[nl];
#pragma unroll
      for (int j = 0; j < ND1; ++j)
      {
        const int off = int((ow[(i * ND1 + j) >> 2] >> (8 * ((i * ND1 + j) & 3))) & 0xff);
        double v;
        if constexpr (USE_LAZY)
          v = Op::entry(lz, i, 0, j, 0);
        else
          v = Ae[i * ND1 + j];
        __hip_atomic_fetch_add(s_vals + base + off, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  if (late_mask)
  {
#pragma unroll
    for (int j = 0; j < MW; ++j)
    {
      const int o = 4 * (tid + j * NT);
      any |= int(mw[j]);
      if (o < slots)
        *reinterpret_cast<uint32_t*>(s_mask + o) = mw[j];
    }
  }
  // most blocks hold no Dirichlet / slave dof at all: they write their values out without looking at the masks
  const bool masked_block = __syncthreads_or(any) != 0;
  if (a.block_vals)
  {
    // block-scalar storage: the block's values as they are, one per bs x bs block; masks and structural zeros are
    // applied by the consumers (mpcx_block_expand, mpcx_spmv_blockscalar)
    if (a.store_mode)
      for (int i = tid; i < slots; i += NT)
        a.block_vals[gslot0 + i] = s_vals[i];
    else
      for (int i = tid; i < slots; i += NT)
        a.block_vals[gslot0 + i] += s_vals[i];
    return;
  }
  // expand: a wave takes whole scalar rows (node nl, component k); row k of a node starts k * L * BS entries after
  // row 0 (L = column blocks of the node's rows).  (A wave per node -- BS * BS * L contiguous entries, 98 % of the
  // lanes busy instead of 66 % -- measured slower: 12.6 against 10.9 ms on the Taylor-Hood velocity block.)
  // Round 6: RB rows per wave and trip, their LDS reads (row bounds, then values) issued together and waited for once:
  // with the row-at-a-time loop every row paid three dependent LDS round trips behind the scatter-adds of the CU's other
  // workgroup, and the write-out ran at half the rate the HBM takes (Taylor-Hood a00, 35.7 GB of values: 13-15 ms).
  const int wave = tid >> 6, lane = tid & 63, nwaves = NT >> 6;
  constexpr int RB = 4;
  const int nrows = nn * BS;
  const bool mapped = a.val_map != nullptr;
  if (a.store_mode && !mapped && !nodeblock_narrow_stores() && (!masked_block || stage_mask))
  {
    // store mode, the launch's own CSR.  A lane holds TWO neighbouring entries of a row and issues one 16-byte store (rows
    // start on 8-byte boundaries).  Everything about a row is wave-uniform and kept in scalar registers (row bounds through
    // readfirstlane); at most one of a lane's two entries is a (k, k) entry, so a row costs ONE unconditional LDS read, two
    // selects and the store -- no branch per entry.  (The first version of this loop evaluated every entry behind its own
    // exec-mask branch, ~300 instructions per trip of four rows.)  The masks of a block with Dirichlet / slave dofs come
    // from LDS: a global load inside this loop would wait for every store issued before it (one in-order counter on gfx9).
    // Taylor-Hood a00 at 128^3 (34.9 GB of values), 1024 threads, blocks of 9216 slots, taken apart with probe switches:
    // entity loop alone 3.1 ms, write-out alone 8.9 ms, both 10.1 ms; the same bytes in the same row pattern from a
    // kernel without global loads (tools/probes/store_pattern.hip, pattern D): 6.4 ms = 5.5 TB/s, memset 5.8 ms.  What
    // separates 8.9 from 6.4 is the chain of dependent loads at the head of every block (block bounds -> row offsets ->
    // entity list -> dofmaps -> coordinates) behind the other CUs' stores, with ONE resident workgroup per CU (88 VGPRs);
    // two of 512 threads overlap it and lose more than they gain (10.0 against 9.2 ms with blocks of 8192 slots).
    typedef double __attribute__((ext_vector_type(2), aligned(8))) double2_a8;
    static_assert(BS >= 2, "blocked spaces only");
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    double* const out = a.vals + nnz0;
    auto rows = [&](auto masked_c)
    {
      constexpr bool MASKED = decltype(masked_c)::value;
      for (int rb = swave * RB; rb < nrows; rb += nwaves * RB)
      {
        int lo[RB], len[RB], kk[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u)
        {
          const int rl = rb + u < nrows ? rb + u : nrows - 1;
          const int nl = rl / BS;
          kk[u] = rl - nl * BS;
          lo[u] = __builtin_amdgcn_readfirstlane(s_rowlo[nl]);
          const int hi = __builtin_amdgcn_readfirstlane(s_rowlo[nl + 1]);
          len[u] = rb + u < nrows ? (hi - lo[u]) * BS : 0;
        }
        int maxlen = 0;
#pragma unroll
        for (int u = 0; u < RB; ++u)
          maxlen = len[u] > maxlen ? len[u] : maxlen;
        for (int base = 0; base < maxlen; base += 128) // (base > 0: vertex rows of P2 spaces, master rows)
        {
          const int e0 = base + 2 * lane;
          const int c0 = e0 / BS, r0 = e0 - c0 * BS;
          const int r1 = r0 + 1 == BS ? 0 : r0 + 1;
          const int c1 = r0 + 1 == BS ? c0 + 1 : c0;
          double val[RB];
          [[maybe_unused]] int mk[RB];
#pragma unroll
          for (int u = 0; u < RB; ++u)
          {
            int p = lo[u] + (r0 == kk[u] ? c0 : c1);
            p = p < slots ? p : slots - 1;
            val[u] = s_vals[p];
            if constexpr (MASKED)
              mk[u] = s_mask[p];
          }
#pragma unroll
          for (int u = 0; u < RB; ++u)
          {
            if (base >= len[u]) // (uniform)
              continue;
            double w = val[u];
            if constexpr (MASKED)
              w = ((mk[u] >> kk[u]) & 1) ? 0.0 : w;
            double2_a8 v;
            v.x = r0 == kk[u] ? w : 0.0;
            v.y = r1 == kk[u] ? w : 0.0;
            double* row = out + (int64_t(lo[u]) * (BS * BS) + int64_t(kk[u]) * len[u]);
            if (e0 + 1 < len[u])
              *reinterpret_cast<double2_a8*>(row + e0) = v;
            else if (e0 < len[u])
              row[e0] = v.x;
          }
        }
      }
    };
    if (masked_block)
      rows(std::true_type{});
    else
      rows(std::false_type{});
    return;
  }
  for (int rb = wave * RB; rb < nrows; rb += nwaves * RB)
  {
    int lo[RB], len[RB], kk[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u)
    {
      const int rl = rb + u < nrows ? rb + u : nrows - 1;
      const int nl = rl / BS;
      kk[u] = rl - nl * BS;
      lo[u] = s_rowlo[nl];
      len[u] = rb + u < nrows ? (s_rowlo[nl + 1] - lo[u]) * BS : 0;
    }
    // the first two 64-entry trips of every row in one go (interior P2 rows of a Kuhn mesh: 87 entries)
    double v[RB][2];
    bool keep[RB][2];
#pragma unroll
    for (int u = 0; u < RB; ++u)
#pragma unroll
      for (int it = 0; it < 2; ++it)
      {
        const int e = lane + 64 * it;
        const int sl = e / BS, q = e - sl * BS;
        keep[u][it] = e < len[u] && q == kk[u];
        if (keep[u][it] && masked_block)
          keep[u][it] = !((a.slot_mask[gslot0 + lo[u] + sl] >> kk[u]) & 1);
        v[u][it] = keep[u][it] ? s_vals[lo[u] + sl] : 0.0;
      }
#pragma unroll
    for (int u = 0; u < RB; ++u)
    {
      const int64_t p0 = nnz0 + int64_t(lo[u]) * (BS * BS) + int64_t(kk[u]) * len[u];
#pragma unroll
      for (int it = 0; it < 2; ++it)
      {
        const int e = lane + 64 * it;
        if (e >= len[u])
          continue;
        const int64_t pos = mapped ? MPCX_VAL_POS(a, p0 + e) : p0 + e;
        if (a.store_mode)
          a.vals[pos] = v[u][it];
        else if (keep[u][it])
          a.vals[pos] += v[u][it];
      }
      for (int e = lane + 128; e < len[u]; e += 64) // long rows (master rows, vertices of high valence)
      {
        const int sl = e / BS, q = e - sl * BS;
        const bool kp = q == kk[u] && !(masked_block && ((a.slot_mask[gslot0 + lo[u] + sl] >> kk[u]) & 1));
        const double w = kp ? s_vals[lo[u] + sl] : 0.0;
        const int64_t pos = mapped ? MPCX_VAL_POS(a, p0 + e) : p0 + e;
        if (a.store_mode)
          a.vals[pos] = w;
        else if (kp)
          a.vals[pos] += w;
      }
    }
  }
}

template <class Op, bool USE_LAZY>
__global__ void __launch_bounds__(ROWBLOCK_MAX_THREADS) matrix_nodeblock_kernel(mpcx_matrix_args_t a)
{
  nodeblock_body<Op, USE_LAZY>(a);
}
// (Round 6, measured and removed: instances capped at 80 / 64 VGPRs so that two workgroups of 768 / 1024 threads share a CU --
// CSR-valued Taylor-Hood a00: 13.3 / 14.9 ms against 11.3 ms for one 1024-thread workgroup of the 84-register instance.)

// set-up for the node-block kernel: one thread per node row
__global__ void diag_slot_mask_kernel(int32_t n_nodes, const mpcx_nnz_t* __restrict__ rowptr,
                                      const int32_t* __restrict__ cols, int bs, const int8_t* __restrict__ bc0,
                                      const int8_t* __restrict__ slave0, const int8_t* __restrict__ bc1,
                                      const int8_t* __restrict__ slave1, uint8_t* __restrict__ out, int32_t* bad)
{
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_nodes)
    return;
  const int64_t p = rowptr[int64_t(n) * bs];
  const int64_t len = rowptr[int64_t(n) * bs + 1] - p;
  bool ok = len % bs == 0 && p % (int64_t(bs) * bs) == 0;
  for (int k = 1; k < bs; ++k)
    ok &= rowptr[int64_t(n) * bs + k + 1] - rowptr[int64_t(n) * bs + k] == len;
  if (!ok)
  {
    *bad = 1;
    return;
  }
  unsigned rm = 0;
  for (int k = 0; k < bs; ++k)
    rm |= unsigned((bc0 && bc0[int64_t(n) * bs + k]) || (slave0 && slave0[int64_t(n) * bs + k])) << k;
  const int64_t slot0 = p / (int64_t(bs) * bs);
  for (int64_t sl = 0; sl < len / bs; ++sl)
  {
    const int32_t c = cols[p + sl * bs]; // first column of the block
    if (c % bs != 0)
    {
      *bad = 1;
      return;
    }
    unsigned m = rm;
    for (int k = 0; k < bs; ++k)
      m |= unsigned((bc1 && bc1[c + k]) || (slave1 && slave1[c + k])) << k;
    out[slot0 + sl] = uint8_t(m);
  }
}

// Row-pair variant of the row-block kernel (plan.row_pairs != 0): the unit of work is one (entity, local row dof)
// pair whose rows lie inside the block, not one entity.  A thread-per-entity block evaluates every entity that
// touches it and masks the rows outside, so with small blocks -- vector-valued and P2 spaces, where 74 KB of LDS
// hold 70-110 nodes -- most lanes of most scatter instructions are idle (the halo factor R = 2.7-3 is paid in LDS
// issue slots).  Here every lane owns rows it keeps: the entity context is recomputed per pair (cheaper than R
// masked passes once R > ~1.5) and the pairs of a block are ordered by (local row, round-robin over row dofs), so
// a wave runs ONE unrolled row body and its lanes add into different CSR rows (no same-address serialisation).
// Operators with a compact context only (Op::prepare / Op::entry).
template <class Op>
__global__ void __launch_bounds__(ROWBLOCK_MAX_THREADS) matrix_rowpair_kernel(mpcx_matrix_args_t a)
{
  constexpr int ND0 = Op::ND0, ND1 = Op::ND1, BS0 = Op::BS0, BS1 = Op::BS1, NV = Op::NV;
  const int NT = blockDim.x;
  extern __shared__ __align__(16) unsigned char smem[];
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (b >= nb)
    return;
  const int tid = threadIdx.x;
  const int r0 = a.plan.block_row0[b], r1 = a.plan.block_row0[b + 1];
  const int nrow = r1 - r0;
  const int64_t nnz0 = a.rowptr[r0];
  const int nnzb = int(a.rowptr[r1] - nnz0);
  constexpr int CW = Op::DIAG ? BS1 : 1; // compact layout of component-diagonal forms, see matrix_rowblock_kernel
  double* s_vals = reinterpret_cast<double*>(smem);
  int32_t* s_rowlo = reinterpret_cast<int32_t*>(s_vals + a.plan.max_nnz / CW);
  for (int i = tid; i < nnzb / CW; i += NT)
    s_vals[i] = 0.0;
  for (int rl = tid; rl <= nrow; rl += NT)
    s_rowlo[rl] = int(a.rowptr[r0 + rl] - nnz0) / CW;
  __syncthreads();

  const int64_t e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  const uint32_t* __restrict__ pairs = reinterpret_cast<const uint32_t*>(a.plan.block_ents);
  for (int64_t t = e0 + tid; t < e1; t += NT)
  {
    const uint32_t id = pairs[t];
    const int64_t e = id / uint32_t(ND0);
    const int i = int(id - uint32_t(e) * uint32_t(ND0));
    const int64_t l = e * a.estride;
    const int64_t cell = (a.entities ? a.entities[l] : e);
    const int64_t cell0 = (a.entities0 ? a.entities0[l] : e);
    const int64_t cell1 = (a.entities1 ? a.entities1[l] : e);
    const int32_t w0 = a.mdofmap0[cell0 * ND0 + i];
    int32_t m1[ND1];
#pragma unroll
    for (int j = 0; j < ND1; ++j)
      m1[j] = a.mdofmap1[cell1 * ND1 + j];
    uint32_t off[ND1];
    const uint8_t* po = a.plan.ent_offs + e * (ND0 * ND1) + i * ND1;
#pragma unroll
    for (int j = 0; j < ND1; ++j)
      off[j] = po[j];
    double cd[NV * 3];
    gather_coords<NV>(a.x, a.x_dofmap, cell, cd);
    typename Op::Lazy lz;
    Op::prepare(lz, a.constants, cd);
#pragma unroll
    for (int I = 0; I < ND0; ++I)
    {
      if (i != I)
        continue;
#pragma unroll
      for (int k = 0; k < BS0; ++k)
      {
        if ((w0 >> (MPCX_MASK_SHIFT + k)) & 1)
          continue;
        const int base = s_rowlo[(w0 & MPCX_DOF_MASK) * BS0 + k - r0];
#pragma unroll
        for (int j = 0; j < ND1; ++j)
        {
#pragma unroll
          for (int q = 0; q < BS1; ++q)
          {
            if constexpr (Op::DIAG)
            {
              if (k != q)
                continue;
            }
            if ((m1[j] >> (MPCX_MASK_SHIFT + q)) & 1)
              continue;
            __hip_atomic_fetch_add(s_vals + base + int(off[j]) * (BS1 / CW) + (Op::DIAG ? 0 : q),
                                   Op::entry(lz, I, k, j, q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
  }
  __syncthreads();
  if constexpr (Op::DIAG)
  {
    const int wave = tid >> 6, lane = tid & 63, nwaves = NT >> 6;
    for (int rl = wave; rl < nrow; rl += nwaves)
    {
      const int k = (r0 + rl) % BS0;
      const int lo = s_rowlo[rl];
      const int64_t p0 = nnz0 + int64_t(lo) * CW;
      const int len = (s_rowlo[rl + 1] - lo) * CW;
      const double* src = s_vals + lo;
      for (int e = lane; e < len; e += 64)
      {
        const int q = e % BS1;
        const double v = (q == k) ? src[e / BS1] : 0.0;
        if (a.store_mode)
          a.vals[MPCX_VAL_POS(a, p0 + e)] = v;
        else if (q == k)
          a.vals[MPCX_VAL_POS(a, p0 + e)] += v;
      }
    }
  }
  else
    MPCX_WRITE_OUT(a, nnz0, nnzb, s_vals, tid, NT);
}

// Per-cell rotation of the local (vertex) numbering used by the lean row-block path: cell c lists
// its local dofs in the order (i + c mod nd) mod nd.  Neighbouring cells (the lanes of one wave)
// that share a matrix entry then reach it at different instructions, which removes most
// same-address LDS atomic conflicts (e.g. the 6 tets round a cube diagonal: 6-way -> 2-way).
__host__ __device__ inline int rotated_local(int i, int64_t cell, int nd) { return int((i + cell % nd) % nd); }

// scatter-offset table (set-up kernel, one thread per (entity, local row block))
__global__ void scatter_offsets_kernel(const mpcx_nnz_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                       int estride, int64_t n_entities, const int32_t* __restrict__ entities0,
                                       const int32_t* __restrict__ entities1, const int32_t* __restrict__ dofmap0,
                                       int nd0, int bs0, const int32_t* __restrict__ dofmap1, int nd1, int bs1,
                                       int rotate, uint8_t* __restrict__ out, int32_t* overflow)
{
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n_entities * nd0)
    return;
  const int64_t e = t / nd0;
  const int i = int(t - e * nd0);
  const int64_t cell0 = entities0 ? entities0[e * estride] : e, cell1 = entities1 ? entities1[e * estride] : e;
  const int r = dofmap0[cell0 * nd0 + (rotate ? rotated_local(i, cell0, nd0) : i)] * bs0;
  const int64_t lo = rowptr[r], hi = rowptr[r + 1];
  for (int j = 0; j < nd1; ++j)
  {
    const int c = dofmap1[cell1 * nd1 + (rotate ? rotated_local(j, cell1, nd1) : j)] * bs1;
    const int64_t pos = csr_find(cols, lo, hi, c);
    const int o = pos < 0 ? 256 : int((pos - lo) / bs1);
    if (o > 255)
      atomicOr(overflow, 1);
    out[(e * nd0 + i) * nd1 + j] = uint8_t(o);
  }
}

// dofmap with the mask folded in (set-up kernel, one thread per dofmap entry)
__global__ void mask_dofmap_kernel(const int32_t* __restrict__ dofmap, int64_t n, int nd, int bs,
                                   const int8_t* __restrict__ bc, const int8_t* __restrict__ is_slave, int rotate,
                                   int32_t* __restrict__ out)
{
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  int64_t src = i;
  if (rotate)
  {
    const int64_t cell = i / nd;
    src = cell * nd + rotated_local(int(i - cell * nd), cell, nd);
  }
  const int32_t d = dofmap[src];
  int32_t m = d;
  for (int k = 0; k < bs; ++k)
  {
    const int64_t u = int64_t(d) * bs + k;
    if ((bc && bc[u]) || is_slave[u])
      m |= 1 << (MPCX_MASK_SHIFT + k);
  }
  out[i] = m;
}

// ---------------------------------------------------------------------------
// Sparsity pattern on the device (create_sparsity_pattern, cpp/utils.h:381-496; same result as
// the host builder mpcx_pattern_build).  Three steps:
//   1. row-block -> cells adjacency: every cell under each of its row blocks and under the
//      master blocks of its row slaves (count, scan by the caller, fill);
//   2. per row block: union of the column blocks of its cells (+ the master blocks of their
//      column slaves), kept as a sorted duplicate-free list in LDS (one row per thread);
//   3. the same walk again, writing the list expanded to the scalar CSR.
// ---------------------------------------------------------------------------
__global__ void pattern_adj_kernel(int64_t num_cells, const int32_t* __restrict__ dofmap0, int nd0, int bs0,
                                   const int32_t* __restrict__ c2s_off, const int32_t* __restrict__ c2s,
                                   const int32_t* __restrict__ m_off, const int32_t* __restrict__ masters,
                                   const int64_t* __restrict__ adj_off, int32_t* __restrict__ counter,
                                   int32_t* __restrict__ adj)
{
  const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c >= num_cells)
    return;
  auto visit = [&](int32_t blk)
  {
    const int32_t k = atomicAdd(counter + blk, 1);
    if (adj) // second pass: place the cell
      adj[adj_off[blk] + k] = int32_t(c);
  };
  for (int i = 0; i < nd0; ++i)
    visit(dofmap0[c * nd0 + i]);
  for (int32_t p = c2s_off[c]; p < c2s_off[c + 1]; ++p)
  {
    const int32_t s = c2s[p];
    for (int32_t q = m_off[s]; q < m_off[s + 1]; ++q)
      visit(masters[q] / bs0);
  }
}

constexpr int PATTERN_MAX_ROW = 128; // distinct column blocks per row block held in LDS
constexpr int PATTERN_THREADS = 64;

template <bool FILL>
__global__ void __launch_bounds__(PATTERN_THREADS)
pattern_rows_kernel(int32_t num_blocks0, const int64_t* __restrict__ adj_off, const int32_t* __restrict__ adj,
                    const int32_t* __restrict__ dofmap1, int nd1, int bs1, const int32_t* __restrict__ c2s_off,
                    const int32_t* __restrict__ c2s, const int32_t* __restrict__ m_off,
                    const int32_t* __restrict__ masters, int32_t* __restrict__ row_count,
                    const mpcx_nnz_t* __restrict__ rowptr, int bs0, int32_t* __restrict__ cols,
                    int32_t* __restrict__ overflow)
{
  // list of thread t: s_list[k * PATTERN_THREADS + t] (bank-conflict-free across the wave)
  __shared__ int32_t s_list[PATTERN_MAX_ROW * PATTERN_THREADS];
  const int tid = threadIdx.x;
  const int64_t r = int64_t(blockIdx.x) * PATTERN_THREADS + tid;
  if (r >= num_blocks0)
    return;
  int n = 0;
  bool over = false;
  auto insert = [&](int32_t x)
  {
    int lo = 0, hi = n; // first position with list[pos] >= x
    while (lo < hi)
    {
      const int mid = (lo + hi) >> 1;
      if (s_list[mid * PATTERN_THREADS + tid] < x)
        lo = mid + 1;
      else
        hi = mid;
    }
    if (lo < n && s_list[lo * PATTERN_THREADS + tid] == x)
      return;
    if (n == PATTERN_MAX_ROW)
    {
      over = true;
      return;
    }
    for (int k = n; k > lo; --k)
      s_list[k * PATTERN_THREADS + tid] = s_list[(k - 1) * PATTERN_THREADS + tid];
    s_list[lo * PATTERN_THREADS + tid] = x;
    ++n;
  };
  for (int64_t a = adj_off[r]; a < adj_off[r + 1]; ++a)
  {
    const int64_t c = adj[a];
    for (int j = 0; j < nd1; ++j)
      insert(dofmap1[c * nd1 + j]);
    for (int32_t p = c2s_off[c]; p < c2s_off[c + 1]; ++p)
    {
      const int32_t s = c2s[p];
      for (int32_t q = m_off[s]; q < m_off[s + 1]; ++q)
        insert(masters[q] / bs1);
    }
  }
  if (over)
    atomicOr(overflow, 1);
  if constexpr (!FILL)
    row_count[r] = n;
  else
  {
    for (int k = 0; k < bs0; ++k)
    {
      int32_t* dst = cols + rowptr[r * bs0 + k];
      for (int j = 0; j < n; ++j)
      {
        const int32_t cb = s_list[j * PATTERN_THREADS + tid];
        for (int l = 0; l < bs1; ++l)
          dst[j * bs1 + l] = cb * bs1 + l;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// val_map: the write-out permutation of mpcx_matrix_args_t (NULL: none)
__global__ void add_diagonal_kernel(const mpcx_nnz_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                    double* vals, const int32_t* __restrict__ dofs, int64_t n, double diagval,
                                    const void* __restrict__ val_map, int val_map_wide)
{
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const int32_t d = dofs[i];
  int64_t pos = csr_find(cols, rowptr[d], rowptr[d + 1], d);
  if (pos >= 0)
  {
    if (val_map)
      pos = val_map_wide ? static_cast<const int64_t*>(val_map)[pos] : int64_t(static_cast<const uint32_t*>(val_map)[pos]);
    atomic_add_f64(vals + pos, diagval);
  }
}

// ---------------------------------------------------------------------------
// Vector kernel: cpp/assemble_vector.cpp:65-90 + modify_mpc_vec
// (cpp/assemble_vector.h:35-69).
// ---------------------------------------------------------------------------
//
// Device-scope f64 atomics run at the memory side (~31 G/s measured), so the
// workgroup first merges its contributions per destination dof in an LDS hash
// table (consecutive entities share most of their dofs) and issues one device
// atomic per distinct dof.  Slave entries go straight to their masters.
#ifndef MPCX_VECTOR_LOG2H_LARGE
#define MPCX_VECTOR_LOG2H_LARGE 10
#endif
template <int N>
struct VectorCfg
{
  static constexpr int NT = N <= 4 ? 256 : (N <= 16 ? 128 : 64); // threads per workgroup
  // table size: >= 2 * NT * N for small elements (can never fill up); larger elements get the table their
  // occupancy allows (48 KB per 2-wave workgroup left 6 waves per CU) and entries that find no slot
  // within the probe limit go to memory directly
  static constexpr int LOG2H = N <= 4 ? 11 : MPCX_VECTOR_LOG2H_LARGE;
  static constexpr int H = 1 << LOG2H;
  static constexpr int PROBES = N <= 4 ? H : 32;
  // minimum waves per SIMD the register allocation is cut for: the P1 source loop runs faster on 5 waves
  // with 96 VGPRs and 28 spills than on 4 waves with 126 VGPRs (3.63 -> 3.34 ms); 6 waves lose (4.13 ms)
  static constexpr int WAVES = N <= 4 ? 5 : 1;
  static_assert(N > 4 || H >= 2 * NT * N, "hash table too small");
};

template <class Op>
__global__ void __launch_bounds__(VectorCfg<Op::N0>::NT) __attribute__((amdgpu_waves_per_eu(VectorCfg<Op::N0>::WAVES)))
vector_kernel(mpcx_vector_args_t a)
{
  constexpr int N = Op::N0, ND = Op::ND0, BS = Op::BS0, NV = Op::NV;
  constexpr int NT = VectorCfg<N>::NT, H = VectorCfg<N>::H, LOG2H = VectorCfg<N>::LOG2H;
  __shared__ int32_t s_key[H];
  __shared__ double s_val[H];
  for (int i = threadIdx.x; i < H; i += NT)
  {
    s_key[i] = -1;
    s_val[i] = 0.0;
  }
  fastmath_init_lds(); // table of the analytic right-hand sides; ends in the barrier the hash needs too
  const int64_t e = int64_t(blockIdx.x) * NT + threadIdx.x;
  if (e < a.n_entities)
  {
    const int64_t l = e * a.estride;
    const int64_t cell = (a.entities ? a.entities[l] : e);
    const int64_t cell0 = (a.entities0 ? a.entities0[l] : e);
    const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
    double cd[NV * 3];
    gather_coords<NV>(a.x, a.x_dofmap, cell, cd);
    double be[N];
    Op::tabulate(be, a.coeffs ? a.coeffs + e * a.cstride : nullptr, a.constants, cd, lf, a.kernel);
#pragma unroll
    for (int i = 0; i < ND; ++i)
    {
      const int32_t d0 = a.dofmap[cell0 * ND + i];
#pragma unroll
      for (int k = 0; k < BS; ++k)
      {
        const int32_t d = d0 * BS + k;
        double v = be[i * BS + k];
        if (a.mpc.is_slave[d])
        {
          const int m0 = a.mpc.masters_offsets[d], m1 = a.mpc.masters_offsets[d + 1];
          for (int mi = m0; mi < m1; ++mi)
            atomic_add_f64(a.b + MPCX_ROW_POS(a, a.mpc.masters[mi]), a.mpc.coeffs[mi] * v);
          if (m1 > m0)
            v = 0.0; // be[slave] is cleared inside the master loop (assemble_vector.h:65)
        }
        if (v != 0.0)
        {
          // open addressing, linear probing (bounded: see VectorCfg)
          unsigned h = (unsigned(d) * 2654435761u) >> (32 - LOG2H);
          int probe = 0;
#pragma nounroll
          for (; probe < VectorCfg<N>::PROBES; ++probe)
          {
            const int32_t old = atomicCAS(&s_key[h], -1, d);
            if (old == -1 || old == d)
            {
              __hip_atomic_fetch_add(&s_val[h], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              break;
            }
            h = (h + 1) & (H - 1);
          }
          if (probe == VectorCfg<N>::PROBES)
            atomic_add_f64(a.b + MPCX_ROW_POS(a, d), v);
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < H; i += NT)
  {
    const int32_t d = s_key[i];
    if (d >= 0)
      atomic_add_f64(a.b + MPCX_ROW_POS(a, d), s_val[i]);
  }
}

// ---------------------------------------------------------------------------
// Row-block vector kernel: the vector counterpart of matrix_rowblock_kernel.  A workgroup owns
// the rows [block_row0[b], block_row0[b+1]) of b in LDS, walks the entities touching them
// (entities on block borders are evaluated by every block they touch), adds the rows it owns
// with ds_add_f64 and adds the finished range to b once, coalesced -- no device atomics (the
// hash kernel above needs ~0.8 device atomics per cell, ~30 G/s at the memory side: 2.9 ms of
// its 3.7 ms at 256^3 do not depend on the quadrature; its arithmetic hides underneath).  Here
// the arithmetic is exposed and entities on block borders are evaluated once per block
// (x1.375): 2.0 ms + 0.22 ms per quadrature point at 256^3, i.e. faster for cheap integrands
// (<= 4 points), slower for the 14-point benchmark right-hand side -- "auto" picks by that.
// Measured and rejected: listing every entity once (owner block) and sending foreign rows with
// device atomics (2.8 ms + 0.11 ms/point), and longer entity runs per hash table (the loop form
// costs the hash kernel its overlap: 5.2 ms).  Rows of slave dofs are skipped here (flag in
// the masked dofmap) and handled by vector_mpc_kernel.
// ---------------------------------------------------------------------------
// (launch bound 1024 = 128 VGPRs: the spill-free build -- 512, 139-175 VGPRs, 2-3 waves per SIMD -- measured slower:
// P2 source 7.7 -> 8.0 ms, vector P1 0.43 -> 0.57 ms)
template <class Op, bool SPLIT = false>
__global__ void __launch_bounds__(ROWBLOCK_MAX_THREADS) vector_rowblock_kernel(mpcx_vector_args_t a)
{
  constexpr int N = Op::N0, ND = Op::ND0, BS = Op::BS0, NV = Op::NV;
  const int NT = blockDim.x;
  extern __shared__ __align__(16) unsigned char smem[];
  double* s_b = reinterpret_cast<double*>(smem); // [max_rows]
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3); // contiguous runs of blocks per XCD
  const int tid = threadIdx.x;
  const int r0 = b < nb ? a.plan.block_row0[b] : 0, r1 = b < nb ? a.plan.block_row0[b + 1] : 0;
  for (int i = tid; i < r1 - r0; i += NT)
    s_b[i] = 0.0;
  fastmath_init_lds(); // ends in a barrier
  if (b >= nb)
    return;

  struct Ent
  {
    int32_t e;
    int lf;
    int32_t xd[NV];
    int32_t m[ND];
  };
  // P1 on an affine mesh: dofmap and geometry dofmap are one device array -> one read
  bool alias = false;
  if constexpr (NV == ND)
    alias = (a.x_dofmap == a.dofmap) && (a.entities0 == a.entities);
  auto load_ent = [&](int64_t e, Ent& E)
  {
    const int64_t l = e * a.estride;
    const int64_t cell = (a.entities ? a.entities[l] : e);
    const int64_t cell0 = (a.entities0 ? a.entities0[l] : e);
    E.e = int32_t(e);
    E.lf = a.estride == 2 ? a.entities[l + 1] : 0;
    if constexpr (ND < 10) // (P2: the dof row is read after the quadrature loop, see below)
    {
#pragma unroll
      for (int i = 0; i < ND; ++i)
        E.m[i] = a.mdofmap[cell0 * ND + i];
    }
    if (alias)
    {
      if constexpr (NV == ND)
      {
#pragma unroll
        for (int i = 0; i < NV; ++i)
          E.xd[i] = E.m[i] & MPCX_DOF_MASK;
      }
    }
    else
    {
#pragma unroll
      for (int i = 0; i < NV; ++i)
        E.xd[i] = a.x_dofmap[cell * NV + i];
    }
  };

  const int64_t e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  const int32_t* __restrict__ ents = a.plan.block_ents;
  int64_t t = e0 + tid;
  // same software pipeline as the matrix kernel: index data one entity ahead, entity index two -- for small
  // elements.  P2 keeps one entity in flight only: the second set of index registers (14 VGPRs) is what made the
  // 24-point source kernel spill inside its quadrature loop
  constexpr bool PIPE = ND < 10 && N <= 8;
  Ent cur;
  int32_t i1 = 0;
  if constexpr (PIPE)
  {
    if (t < e1)
      load_ent(ents[t], cur);
    if (t + NT < e1)
      i1 = ents[t + NT];
  }
  for (; t < e1; t += NT)
  {
    if constexpr (!PIPE)
      load_ent(ents[t], cur);
    double cd[NV * 3];
#pragma unroll
    for (int i = 0; i < NV; ++i)
    {
      const int64_t v = cur.xd[i];
#pragma unroll
      for (int k = 0; k < 3; ++k)
        cd[3 * i + k] = a.x[3 * v + k];
    }
    Ent nxt = cur;
    if constexpr (PIPE)
    {
      if (t + NT < e1)
        load_ent(i1, nxt);
      if (t + 2 * NT < e1)
        i1 = ents[t + 2 * NT];
    }
    // P2: the masked dof row is read AFTER the quadrature loop (ten registers less across it)
    auto load_row = [&]()
    {
      if constexpr (ND >= 10)
      {
        const int64_t l = int64_t(cur.e) * a.estride;
        const int64_t cell0 = (a.entities0 ? a.entities0[l] : cur.e);
#pragma unroll
        for (int i = 0; i < ND; ++i)
          cur.m[i] = a.mdofmap[cell0 * ND + i];
      }
    };
    // vector-valued P2 source without coefficient: one component at a time through the scalar operator (ND
    // accumulators instead of ND * BS: at 128 VGPRs the 30 of P2^3 spilled 204 bytes per thread, 8.5 GB of scratch
    // traffic per launch on the Taylor-Hood benchmark)
    // (SPLIT: chosen by the launcher when there is no coefficient)
    if constexpr (SPLIT)
    {
      {
        using SOp = ElementOp<Op::TDIM, Op::DEG0, 1, Op::DEG0, 1, MPCX_FORM_SOURCE, Op::FN>;
#pragma unroll 1
        for (int k = 0; k < BS; ++k)
        {
          double bk[ND];
          SOp::tabulate(bk, nullptr, a.constants, cd, cur.lf, a.kernel, k);
          load_row();
#pragma unroll
          for (int i = 0; i < ND; ++i)
          {
            const int r = (cur.m[i] & MPCX_DOF_MASK) * BS + k;
            if ((cur.m[i] >> (MPCX_MASK_SHIFT + k)) & 1)
              continue;
            if (r >= r0 && r < r1)
              __hip_atomic_fetch_add(s_b + (r - r0), bk[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
    else
    {
      double be[N];
      Op::tabulate(be, a.coeffs ? a.coeffs + int64_t(cur.e) * a.cstride : nullptr, a.constants, cd, cur.lf, a.kernel);
      load_row();
#pragma unroll
      for (int i = 0; i < ND; ++i)
      {
#pragma unroll
        for (int k = 0; k < BS; ++k)
        {
          const int r = (cur.m[i] & MPCX_DOF_MASK) * BS + k;
          if ((cur.m[i] >> (MPCX_MASK_SHIFT + k)) & 1)
            continue;
          if (r >= r0 && r < r1)
            __hip_atomic_fetch_add(s_b + (r - r0), be[i * BS + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
    if constexpr (PIPE)
      cur = nxt;
  }
  __syncthreads();
  for (int i = tid; i < r1 - r0; i += NT)
    a.b[MPCX_ROW_POS(a, r0 + i)] += s_b[i];
}

// Slave rows of the entities that have any (modify_mpc_vec, cpp/assemble_vector.h:35-69):
// b[master] += coeff * be[slave]; a slave without masters keeps its own row.
// Owner-computes variant of the row-block vector kernel (mpcx_vector_args_t::own_lmap): every entity is evaluated
// ONCE, by the block of its local dof 0.  The LDS copy of the block holds its own dofs followed by the dofs of
// other blocks its entities touch, and the scatter position of every (entity, local dof) comes from a table that
// takes the place of the masked dofmap -- no range tests, no stores from the quadrature loop's neighbourhood (the
// kernel sits at its register bound).  At the end the own part is added to b and the halo part is written out
// contiguously; vector_spill_reduce_kernel adds it to the rows it belongs to.
template <class Op, bool SPLIT = false, int MAXT = ROWBLOCK_MAX_THREADS>
__global__ void __launch_bounds__(MAXT) vector_ownblock_kernel(mpcx_vector_args_t a)
{
  constexpr int N = Op::N0, ND = Op::ND0, BS = Op::BS0, NV = Op::NV;
  constexpr int LMASK = (1 << MPCX_MASK_SHIFT) - 1;
  const int NT = blockDim.x;
  extern __shared__ __align__(16) unsigned char smem[];
  double* s_b = reinterpret_cast<double*>(smem); // [max_rows]: own rows, then halo rows
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const int tid = threadIdx.x;
  const int r0 = b < nb ? a.plan.block_row0[b] : 0, r1 = b < nb ? a.plan.block_row0[b + 1] : 0;
  const int64_t h0 = b < nb ? a.own_hoff[b] : 0, h1 = b < nb ? a.own_hoff[b + 1] : 0;
  const int nown = r1 - r0, nhalo = int(h1 - h0) * BS;
  for (int i = tid; i < nown + nhalo; i += NT)
    s_b[i] = 0.0;
  fastmath_init_lds(); // ends in a barrier
  if (b >= nb)
    return;
  const int64_t e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  const int32_t* __restrict__ ents = a.plan.block_ents;
  for (int64_t t = e0 + tid; t < e1; t += NT)
  {
    const int64_t e = ents[t];
    const int64_t l = e * a.estride;
    const int64_t cell = (a.entities ? a.entities[l] : e);
    const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
    double cd[NV * 3];
    gather_coords<NV>(a.x, a.x_dofmap, cell, cd);
    if constexpr (SPLIT)
    {
      using SOp = ElementOp<Op::TDIM, Op::DEG0, 1, Op::DEG0, 1, MPCX_FORM_SOURCE, Op::FN>;
#pragma unroll 1
      for (int k = 0; k < BS; ++k)
      {
        double bk[ND];
        SOp::tabulate(bk, nullptr, a.constants, cd, lf, a.kernel, k);
#pragma unroll
        for (int i = 0; i < ND; ++i)
        {
          const int32_t w = a.own_lmap[e * ND + i]; // (read after the quadrature loop)
          if (!((w >> (MPCX_MASK_SHIFT + k)) & 1))
            __hip_atomic_fetch_add(s_b + (w & LMASK) * BS + k, bk[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
    else
    {
      double be[N];
      Op::tabulate(be, a.coeffs ? a.coeffs + e * a.cstride : nullptr, a.constants, cd, lf, a.kernel);
#pragma unroll
      for (int i = 0; i < ND; ++i)
      {
        const int32_t w = a.own_lmap[e * ND + i];
#pragma unroll
        for (int k = 0; k < BS; ++k)
          if (!((w >> (MPCX_MASK_SHIFT + k)) & 1))
            __hip_atomic_fetch_add(s_b + (w & LMASK) * BS + k, be[i * BS + k], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < nown; i += NT)
    a.b[MPCX_ROW_POS(a, r0 + i)] += s_b[i];
  for (int i = tid; i < nhalo; i += NT)
    a.own_spill[h0 * BS + i] = s_b[nown + i];
}

// ---- the benchmark's right-hand side per CELL from per-interval tables (mpcx_vector_args_t::grid_eta / grid_J) --------------
// The cluster kernel of csrc/mpcx_cubes.hip (vector_cube_grid_kernel) covers scalar P1 on clusters with the 14-point rule through
// generated index tables.  This one is data driven -- any rule, P1 or P2, one thread per cell: a simplex of a box mesh has its
// vertices on two values per axis, so the coordinate of quadrature point q along axis d is lo_d + h_d eta with eta the sum of the
// barycentric coordinates of the vertices on the high side, one of a few dozen values for all (q, subset).  The launch fills the
// table of the univariate factors per interval (cell_grid_tables_kernel), a block stages the rows of its cells in LDS, and a
// point costs one word (which eta per axis, by the cell's TYPE = which vertices lie high per axis: a box mesh has a handful),
// two pairs and a value from the rows and ND + 3 fma instead of a sine, an exponential and the affine map.  Row of interval r
// (2 NGP + 2 doubles, NGP = grid_ng rounded up to even): pairs (g(t_j), t_j) on axis x, (g(t_j), sin(5 pi t_j)) on y, (g(t_j), 0)
// on z, j < grid_ng; [2 NGP] = |h|.
__global__ void __launch_bounds__(256) cell_grid_tables_kernel(int n0, int n1, int n2, const double* __restrict__ iv, double* __restrict__ tab,
                                                               const double* __restrict__ eta, int ng)
{
  fastmath_init_lds(); // ends in a barrier
  const int ngp = (ng + 1) & ~1, stride = 2 * ngp + 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = i / (ngp + 1), j = i - r * (ngp + 1);
  if (r >= n0 + n1 + n2)
    return;
  const int d = r < n0 ? 0 : (r < n0 + n1 ? 1 : 2);
  const double lo = iv[2 * r], hi = iv[2 * r + 1], h = hi - lo;
  double* row = tab + int64_t(r) * stride;
  if (j == ngp)
  {
    row[2 * ngp] = fabs(h);
    row[2 * ngp + 1] = 0.0;
    return;
  }
  if (j >= ng)
  {
    row[2 * j] = 0.0, row[2 * j + 1] = 0.0;
    return;
  }
  const FmConsts FK = g_fm_consts;
  const double centre = d == 0 ? 0.9 : (d == 1 ? 0.5 : 0.1);
  const double t = fma(h, eta[j], lo - centre); // relative to the centre of the Gaussian, as eval_fn case 1 does
  row[2 * j] = fast_exp_nonpos_k(-(t * t) * (1.0 / 0.02), FK);
  row[2 * j + 1] = d == 0 ? t + 0.9 : (d == 1 ? fast_sinpi_k(fma(5.0, t, 2.5), FK) : 0.0);
}

template <int ND>
__global__ void __launch_bounds__(1024) vector_cell_grid_kernel(mpcx_vector_args_t a)
{
  constexpr int LMASK = (1 << MPCX_MASK_SHIFT) - 1;
  typedef double __attribute__((ext_vector_type(2))) pair_t;
  const int NT = blockDim.x;
  extern __shared__ __align__(16) unsigned char smem[];
  double* s_b = reinterpret_cast<double*>(smem); // [max_rows]: own rows, then halo rows
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (b >= nb)
    return;
  const int tid = threadIdx.x;
  const int r0 = a.plan.block_row0[b], r1 = a.plan.block_row0[b + 1];
  const int64_t h0 = a.own_hoff[b], h1 = a.own_hoff[b + 1];
  const int nown = r1 - r0, nhalo = int(h1 - h0);
  for (int i = tid; i < nown + nhalo; i += NT)
    s_b[i] = 0.0;
  const int ng = a.grid_ng, ngp = (ng + 1) & ~1, stride = 2 * ngp + 2;
  const int lrow = stride + 2; // (LDS rows two doubles longer: rows of different intervals on different banks)
  const int nq = a.kernel.nq;
  const int nslots = a.grid_block_rows_max;
  double* s_rows = s_b + ((a.plan.max_rows + 1) & ~1);                 // [nslots][lrow]
  uint32_t* s_J = reinterpret_cast<uint32_t*>(s_rows + nslots * lrow); // [grid_ntypes][nq]: eta index per axis, one byte each
  {
    const int32_t* __restrict__ br = a.grid_block_rows + int64_t(b) * MPCX_GRID_BLOCK_ROWS;
    for (int i = tid; i < nslots * stride; i += NT)
    {
      const int slot = i / stride, j = i - slot * stride;
      const int r = br[slot];
      if (r >= 0)
        s_rows[slot * lrow + j] = a.grid_tab[int64_t(r) * stride + j];
    }
    const uint32_t* __restrict__ gj = reinterpret_cast<const uint32_t*>(a.grid_J);
    for (int i = tid; i < a.grid_ntypes * nq; i += NT)
      s_J[i] = gj[i];
  }
  __syncthreads();
  const double c0 = a.constants ? a.constants[0] : 1.0;
  const int64_t e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  const int32_t* __restrict__ ents = a.plan.block_ents;
  for (int64_t t = e0 + tid; t < e1; t += NT)
  {
    const int64_t e = ents[t];
    const int4 rec = reinterpret_cast<const int4*>(a.grid_idx)[e];
    const double* __restrict__ rx = s_rows + rec.x * lrow;
    const double* __restrict__ ry = s_rows + rec.y * lrow;
    const double* __restrict__ rz = s_rows + rec.z * lrow;
    const uint32_t* __restrict__ jt = s_J + (rec.w & 0xffff) * nq;
    const double vol = c0 * double(rec.w >> 16) * (rx[2 * ngp] * ry[2 * ngp] * rz[2 * ngp]);
    double acc[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i)
      acc[i] = 0.0;
    for (int q = 0; q < nq; ++q)
    {
      const uint32_t j = jt[q];
      const pair_t px = *reinterpret_cast<const pair_t*>(rx + 2 * (j & 0xff));
      const pair_t py = *reinterpret_cast<const pair_t*>(ry + 2 * ((j >> 8) & 0xff));
      const double gz = rz[2 * ((j >> 16) & 0xff)];
      const double f = (a.kernel.qwts[q] * vol) * fma(px.y, py.y, px.x * py.x * gz);
      if constexpr (ND == 4)
      {
        const double X0 = a.kernel.qpts[3 * q], X1 = a.kernel.qpts[3 * q + 1], X2 = a.kernel.qpts[3 * q + 2];
        acc[0] = fma(f, 1.0 - X0 - X1 - X2, acc[0]);
        acc[1] = fma(f, X0, acc[1]);
        acc[2] = fma(f, X1, acc[2]);
        acc[3] = fma(f, X2, acc[3]);
      }
      else
      {
#pragma unroll
        for (int i = 0; i < ND; ++i)
          acc[i] = fma(f, a.kernel.qphi[q * ND + i], acc[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < ND; ++i)
    {
      const int32_t w = a.own_lmap[e * ND + i];
      if (!((w >> MPCX_MASK_SHIFT) & 1))
        __hip_atomic_fetch_add(s_b + (w & LMASK), acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  for (int i = tid; i < nown; i += NT)
    a.b[MPCX_ROW_POS(a, r0 + i)] += s_b[i];
  for (int i = tid; i < nhalo; i += NT)
    a.own_spill[h0 + i] = s_b[nown + i];
}

// Owner-computes blocks for source forms whose integrand function is AFFINE in x (mpcx_kernel_t::vphi: constant / linear f on
// affine simplices, no coefficient -- the momentum right-hand side of config 3).  With the rule's vertex moments the element
// vector is NV * ND fma per component and the kernel is a gather / LDS-add problem: no quadrature loop, no 128-register
// instance, so it runs 1024-thread workgroups (8 waves per SIMD from two resident blocks) and every component in one pass.
// Round 6: the general instance spent 1.7 ms on 12.6 M P2^3 cells with a tenth of its arithmetic removed -- it is bound by the
// latency of its per-entity chain (entity -> geometry dofmap -> coordinates -> position table -> LDS), i.e. by resident waves.
template <class Op>
__global__ void __launch_bounds__(1024) vector_ownblock_affine_kernel(mpcx_vector_args_t a)
{
  constexpr int ND = Op::ND0, BS = Op::BS0, NV = Op::NV, TDIM = Op::TDIM;
  constexpr int LMASK = (1 << MPCX_MASK_SHIFT) - 1;
  const int NT = blockDim.x;
  extern __shared__ __align__(16) unsigned char smem[];
  double* s_b = reinterpret_cast<double*>(smem);
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (b >= nb)
    return;
  const int tid = threadIdx.x;
  const int r0 = a.plan.block_row0[b], r1 = a.plan.block_row0[b + 1];
  const int64_t h0 = a.own_hoff[b], h1 = a.own_hoff[b + 1];
  const int nown = r1 - r0, nhalo = int(h1 - h0) * BS;
  for (int i = tid; i < nown + nhalo; i += NT)
    s_b[i] = 0.0;
  __syncthreads();
  const int64_t e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  const int32_t* __restrict__ ents = a.plan.block_ents;
  const double* __restrict__ vphi = a.kernel.vphi;
  const double c0 = a.constants ? a.constants[0] : 1.0;
  const int fn = Op::FN >= 0 ? Op::FN : a.kernel.fn_id;
  for (int64_t t = e0 + tid; t < e1; t += NT)
  {
    const int64_t e = ents[t];
    const int64_t cell = a.entities ? a.entities[e * a.estride] : e;
    double cd[NV * 3];
    gather_coords<NV>(a.x, a.x_dofmap, cell, cd);
    int32_t w[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i)
      w[i] = a.own_lmap[e * ND + i]; // (independent of the geometry: in flight with it)
    double det;
    if constexpr (TDIM == 3)
    {
      const double a0 = cd[3] - cd[0], a1 = cd[4] - cd[1], a2 = cd[5] - cd[2];
      const double b0 = cd[6] - cd[0], b1 = cd[7] - cd[1], b2 = cd[8] - cd[2];
      const double d0 = cd[9] - cd[0], d1 = cd[10] - cd[1], d2 = cd[11] - cd[2];
      det = a0 * (b1 * d2 - b2 * d1) - a1 * (b0 * d2 - b2 * d0) + a2 * (b0 * d1 - b1 * d0);
    }
    else
      det = (cd[3] - cd[0]) * (cd[7] - cd[1]) - (cd[4] - cd[1]) * (cd[6] - cd[0]);
    const double sd = c0 * fabs(det);
    // one component at a time, rolled: NV function values live, not BS * NV (92 -> ~64 VGPRs for P2^3: two 1024-thread
    // workgroups per CU)
#pragma unroll 1
    for (int k = 0; k < BS; ++k)
    {
      double F[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v)
      {
        const double xv[3] = {cd[3 * v], cd[3 * v + 1], cd[3 * v + 2]};
        // (the affine members of eval_fn only: the sin / exp branches of the general switch set the register count)
        F[v] = sd * (fn == 4 ? (k + 1) * (1.0 + xv[0] - 2.0 * xv[1] + 0.5 * xv[2]) : (fn == 5 ? a.constants[1 + k] : 1.0));
      }
#pragma unroll
      for (int i = 0; i < ND; ++i)
      {
        if ((w[i] >> (MPCX_MASK_SHIFT + k)) & 1)
          continue;
        double s = 0.0;
#pragma unroll
        for (int v = 0; v < NV; ++v)
          s = fma(vphi[v * ND + i], F[v], s);
        __hip_atomic_fetch_add(s_b + (w[i] & LMASK) * BS + k, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < nown; i += NT)
    a.b[MPCX_ROW_POS(a, r0 + i)] += s_b[i];
  for (int i = tid; i < nhalo; i += NT)
    a.own_spill[h0 * BS + i] = s_b[nown + i];
}

// second half of the owner-computes vector path: one thread per (distinct target dof, component)
__global__ void vector_spill_reduce_kernel(int64_t n_rows, const int32_t* __restrict__ rows,
                                           const int64_t* __restrict__ seg, const int32_t* __restrict__ src,
                                           const double* __restrict__ spill, int bs, double* __restrict__ b,
                                           const int32_t* __restrict__ row_map)
{
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n_rows * bs)
    return;
  const int64_t u = t / bs;
  const int k = int(t - u * bs);
  double sum = 0.0;
  for (int64_t s = seg[u]; s < seg[u + 1]; ++s)
    sum += spill[int64_t(src[s]) * bs + k];
  const int64_t d = int64_t(rows[u]) * bs + k;
  b[row_map ? int64_t(row_map[d]) : d] += sum;
}

// Rows of slave dofs go to their masters (cpp/assemble_vector.h:35-69).  The contributions of a workgroup's entities are
// merged per target row in an LDS hash table first: neighbouring slave cells share their slaves (12 cells round a node of
// a periodic face), and a device-scope fp64 atomic is served at the memory side (~0.1 us each when a step issues a
// million of them: 128 us for the slave layer of config 2 without the table).
constexpr int VECTOR_MPC_THREADS = 256;
constexpr int VECTOR_MPC_LOG2H = 11;
constexpr int VECTOR_MPC_H = 1 << VECTOR_MPC_LOG2H;
template <class Op>
__global__ void __launch_bounds__(VECTOR_MPC_THREADS) vector_mpc_kernel(mpcx_vector_args_t a)
{
  constexpr int N = Op::N0, ND = Op::ND0, BS = Op::BS0, NV = Op::NV;
  __shared__ int32_t s_key[VECTOR_MPC_H];
  __shared__ double s_val[VECTOR_MPC_H];
  for (int i = threadIdx.x; i < VECTOR_MPC_H; i += VECTOR_MPC_THREADS)
  {
    s_key[i] = -1;
    s_val[i] = 0.0;
  }
  fastmath_init_lds(); // ends in a barrier
  auto add = [&](int32_t row, double v)
  {
    unsigned h = (unsigned(row) * 2654435761u) >> (32 - VECTOR_MPC_LOG2H);
#pragma nounroll
    for (int probe = 0; probe < 32; ++probe)
    {
      const int32_t old = atomicCAS(&s_key[h], -1, row);
      if (old == -1 || old == row)
      {
        __hip_atomic_fetch_add(&s_val[h], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
      }
      h = (h + 1) & (VECTOR_MPC_H - 1);
    }
    atomic_add_f64(a.b + MPCX_ROW_POS(a, row), v); // (a full neighbourhood of the table: straight to memory)
  };
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t < a.n_slave_entities)
  {
    const int64_t e = a.slave_entities[t];
    const int64_t l = e * a.estride;
    const int64_t cell = (a.entities ? a.entities[l] : e);
    const int64_t cell0 = (a.entities0 ? a.entities0[l] : e);
    const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
    double cd[NV * 3];
    gather_coords<NV>(a.x, a.x_dofmap, cell, cd);
    double be[N];
    Op::tabulate(be, a.coeffs ? a.coeffs + e * a.cstride : nullptr, a.constants, cd, lf, a.kernel);
#pragma unroll
    for (int i = 0; i < ND; ++i)
    {
      const int32_t d0 = a.dofmap[cell0 * ND + i];
#pragma unroll
      for (int k = 0; k < BS; ++k)
      {
        const int32_t d = d0 * BS + k;
        if (!a.mpc.is_slave[d])
          continue;
        const double v = be[i * BS + k];
        const int m0 = a.mpc.masters_offsets[d], m1 = a.mpc.masters_offsets[d + 1];
        for (int mi = m0; mi < m1; ++mi)
          add(a.mpc.masters[mi], a.mpc.coeffs[mi] * v);
        if (m1 == m0)
          add(d, v);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < VECTOR_MPC_H; i += VECTOR_MPC_THREADS)
  {
    const int32_t row = s_key[i];
    if (row >= 0)
      atomic_add_f64(a.b + MPCX_ROW_POS(a, row), s_val[i]);
  }
}

// ---------------------------------------------------------------------------
// Lifting kernel: cpp/lifting.h:77-133 over the compact list of entities that
// have a bc-marked column dof; Ae is the raw kernel output (:267-272).
// ---------------------------------------------------------------------------
template <class Op>
__global__ void __launch_bounds__(128) lifting_kernel(mpcx_lifting_args_t a)
{
  constexpr int N0 = Op::N0, ND0 = Op::ND0, ND1 = Op::ND1, BS0 = Op::BS0, BS1 = Op::BS1, NV = Op::NV;
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= a.n_lift_entities)
    return;
  const int64_t e = a.lift_entities[t];
  const int64_t l = e * a.estride;
  const int64_t cell = (a.entities ? a.entities[l] : e);
  const int64_t cell0 = (a.entities0 ? a.entities0[l] : e);
  const int64_t cell1 = (a.entities1 ? a.entities1[l] : e);
  const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
  double cd[NV * 3];
  gather_coords<NV>(a.x, a.x_dofmap, cell, cd);
  double Ae[Op::SIZE];
  Op::tabulate(Ae, a.coeffs ? a.coeffs + e * a.cstride : nullptr, a.constants, cd, lf, a.kernel);
  double be[N0];
#pragma unroll
  for (int m = 0; m < N0; ++m)
    be[m] = 0.0;
#pragma unroll
  for (int j = 0; j < ND1; ++j)
  {
    const int32_t d1 = a.dofmap1[cell1 * ND1 + j];
#pragma unroll
    for (int k = 0; k < BS1; ++k)
    {
      const int32_t jj = d1 * BS1 + k;
      if (a.bc_markers1[jj])
      {
        const double g = a.scale * (a.bc_values1[jj] - (a.x0 ? a.x0[jj] : 0.0));
#pragma unroll
        for (int m = 0; m < N0; ++m)
          be[m] -= Op::get(Ae, m, j * BS1 + k) * g;
      }
    }
  }
  // slaves are applied against the row dofmap (cpp/lifting.h:117-127)
#pragma unroll
  for (int i = 0; i < ND0; ++i)
  {
    const int32_t d0 = a.dofmap0[cell0 * ND0 + i];
#pragma unroll
    for (int k = 0; k < BS0; ++k)
    {
      const int32_t d = d0 * BS0 + k;
      double v = be[i * BS0 + k];
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
}

// cpp/MultiPointConstraint.h:129-145.  Slaves whose masters are themselves
// slaves are not resolved (same as the reference's sequential loop only when
// no master is a slave).
__global__ void backsubstitution_kernel(double* u, const int32_t* __restrict__ slaves, int64_t n, mpcx_mpc_t mpc)
{
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const int32_t s = slaves[i];
  double v = 0.0;
  for (int mi = mpc.masters_offsets[s]; mi < mpc.masters_offsets[s + 1]; ++mi)
    v += mpc.coeffs[mi] * u[mpc.masters[mi]];
  u[s] = v;
}

__global__ void homogenize_kernel(double* u, const int32_t* __restrict__ slaves, int64_t n)
{
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n)
    u[slaves[i]] = 0.0;
}

// ---------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------
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

template <class Op>
int launch_matrix(const mpcx_matrix_args_t& a)
{
  if (a.nd0 != Op::ND0 || a.nd1 != Op::ND1 || a.bs0 != Op::BS0 || a.bs1 != Op::BS1 || a.nv != Op::NV)
  {
    mpcx_set_error("mpcx_assemble_matrix: dofmap shapes do not match the element kernel");
    return -12;
  }
  hipStream_t stream = static_cast<hipStream_t>(a.stream);
  int alg = a.algorithm;
  if (alg == MPCX_ALG_AUTO)
    alg = a.plan.num_blocks > 0 ? MPCX_ALG_ROWBLOCK : MPCX_ALG_ATOMIC;
  if (alg == MPCX_ALG_CUBE)
  {
    if (int rc = launch_matrix_cubes(a))
      return rc;
  }
  else if (a.n_entities > 0)
  {
    if (alg == MPCX_ALG_ROWBLOCK && a.plan.row_pairs == 2)
    {
      // pair records + cached contexts (mpcx_pairs.hip)
      if (int rc = launch_matrix_pairs(a))
        return rc;
    }
    else if (alg == MPCX_ALG_ROWBLOCK)
    {
      if (a.plan.num_blocks <= 0)
      {
        mpcx_set_error("mpcx_assemble_matrix: row-block algorithm needs a plan");
        return -3;
      }
      if (!a.mdofmap0 || !a.mdofmap1)
      {
        mpcx_set_error("mpcx_assemble_matrix: row-block algorithm needs masked dofmaps (mpcx_mask_dofmap)");
        return -5;
      }
      if (!a.plan.ent_offs)
      {
        mpcx_set_error("mpcx_assemble_matrix: row-block plan lacks the scatter-offset table (mpcx_scatter_offsets)");
        return -5;
      }
      // component-diagonal forms keep one value per column block (see the kernel): BS1 times less LDS per row
      size_t lds = a.slot_mask ? size_t(a.plan.max_nnz / (Op::BS0 * Op::BS1)) * 8 + size_t(a.plan.max_rows / Op::BS0 + 1) * 4
                               : size_t(a.plan.max_nnz / (Op::DIAG ? Op::BS1 : 1)) * 8 + size_t(a.plan.max_rows + 1) * 4;
      if (a.slot_mask && !a.block_vals && nodeblock_stage_mask(a.plan.max_nnz, a.plan.max_rows, Op::BS0))
        lds += (size_t(a.plan.max_nnz / (Op::BS0 * Op::BS1)) + 3) & ~size_t(3); // the block's mask bytes (read by the write-out)
      if (lds > 160 * 1024)
      {
        mpcx_set_error("mpcx_assemble_matrix: row-block plan exceeds 160 KiB of LDS");
        return -4;
      }
      if (a.lds_floor > 0 && size_t(a.lds_floor) > lds && a.lds_floor <= 160 * 1024) // per-launch occupancy cap (include/mpcx.h)
        lds = size_t(a.lds_floor);
      const unsigned grid = 8u * unsigned((a.plan.num_blocks + 7) / 8);
      // threads per workgroup (two workgroups per CU by the LDS budget of the plan): chosen per kernel
      // below, MPCX_ROWBLOCK_THREADS overrides
      const int env_threads = []
      {
        const char* e = std::getenv("MPCX_ROWBLOCK_THREADS");
        const int t = e ? std::atoi(e) : 0;
        return (t >= 64 && t <= 1024 && t % 64 == 0) ? t : 0;
      }();
      auto launch = [&](auto kernel, int forced_threads = 0) -> int
      {
        if (int rc = check(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)),
                           "hipFuncSetAttribute"))
          return rc;
        int threads = forced_threads ? forced_threads : env_threads;
        if (threads == 0)
        {
          hipFuncAttributes attr;
          if (int rc = check(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kernel)), "hipFuncGetAttributes"))
            return rc;
          // 512 threads (2 x 8 waves per CU) unless the kernel is light enough for 2 x 12 waves AND runs
          // the pipelined small-element loop on full-size blocks: P1 stiffness 1.96 -> 1.81 ms (P2 and elasticity
          // lose with 768); the host picks half-size blocks for that kernel, four 512-thread workgroups per CU
          threads = (attr.numRegs <= 64 && Op::ND0 * Op::ND1 <= 16 && a.plan.max_rows > 256) ? 768 : 512;
          // light threads, many more of them than entities: contact elasticity 1.02 -> 0.96 ms.  Only kernels of at most 64
          // registers: two 1024-thread workgroups per CU are 8 waves per SIMD, and with ONE resident workgroup nothing
          // computes while a block is written out (Taylor-Hood a00, 84 registers: 1024 threads 2.98 ms, 512 2.16 ms)
          if ((a.plan.row_pairs || a.slot_mask) && attr.numRegs <= 64)
            threads = 1024;
          static const bool narrow_set = []
          {
            const char* e = std::getenv("MPCX_NODEBLOCK_NARROW_STORES");
            const int v = (e && e[0] == '1') ? 1 : 0; // 1: the general loop below (8-byte stores) for every block
            if (v)
              (void)hipMemcpyToSymbol(HIP_SYMBOL(g_nodeblock_narrow_stores), &v, sizeof(int));
            return true;
          }();
          (void)narrow_set;
          // node blocks expanded to scalar CSR values (no block_vals): the write-out of 9 x the LDS block is the longer
          // phase and its rate follows the number of waves that issue stores: MPCX_NODEBLOCK_CSR_THREADS (default below)
          if (a.slot_mask && !a.block_vals)
          {
            // (Taylor-Hood a00 at 128^3, 34.9 GB of values, blocks of 8192 slots: 1024 threads 9.0-9.2 ms, 512 10.0; before the
            // 16-byte stores, the branch-free row loop and the masks in LDS: 1024 11.3 ms, 768 13.9, 512 13.2)
            const char* e = std::getenv("MPCX_NODEBLOCK_CSR_THREADS");
            const int t = e ? std::atoi(e) : 1024;
            threads = (t >= 64 && t <= 1024 && t % 64 == 0) ? t : 1024;
          }
        }
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, stream, a);
        return 0;
      };
      constexpr bool CAN_LEAN = Op::SQUARE && !Op::FACET && Op::NV == Op::ND0;
      const bool lean = a.lean != 0;
      if (a.slot_mask)
      {
        if constexpr (Op::DIAG && Op::BS0 == Op::BS1)
        {
          if (a.plan.row_pairs || a.plan.ent_pattern || a.entities0 != a.entities || a.entities1 != a.entities)
          {
            mpcx_set_error("mpcx_assemble_matrix: slot_mask (node-block kernel) needs a plain entity plan and a "
                           "direct offset table");
            return -8;
          }
          bool lazy = false;
          if constexpr (Op::LAZY)
            lazy = Op::lazy_applies(a.kernel);
          int rc = 0;
          if constexpr (Op::LAZY)
            rc = lazy ? launch(matrix_nodeblock_kernel<Op, true>) : launch(matrix_nodeblock_kernel<Op, false>);
          else
            rc = launch(matrix_nodeblock_kernel<Op, false>);
          if (rc)
            return rc;
        }
        else
        {
          mpcx_set_error("mpcx_assemble_matrix: slot_mask given for an operator that is not component-diagonal");
          return -8;
        }
      }
      else if (a.plan.row_pairs)
      {
        bool ok = false;
        if constexpr (Op::LAZY)
          ok = Op::lazy_applies(a.kernel) && !a.plan.ent_pattern && !a.coeffs;
        if (!ok)
        {
          mpcx_set_error("mpcx_assemble_matrix: a row-pair plan needs an operator with a compact context "
                         "(stiffness without coefficient, elasticity, Taylor-Hood blocks) and a direct offset table");
          return -8;
        }
        if constexpr (Op::LAZY)
        {
          if (int rc = launch(matrix_rowpair_kernel<Op>))
            return rc;
        }
      }
      else
      {
      if (lean)
      {
        bool ok = false;
        if constexpr (CAN_LEAN)
          ok = a.estride == 1 && !a.entities && !a.entities0 && !a.entities1 && a.mdofmap1 == a.mdofmap0
               && a.x_dofmap == a.dofmap0 && !a.coeffs;
        if (!ok)
        {
          mpcx_set_error("mpcx_assemble_matrix: lean row-block path needs a square P1-type form over all cells "
                         "without coefficients, dofmap0 == x_dofmap and mdofmap1 == mdofmap0");
          return -8;
        }
      }
      int rc = 0;
      bool lazy = false;
      if constexpr (Op::LAZY)
        lazy = Op::lazy_applies(a.kernel);
      if constexpr (CAN_LEAN && Op::LAZY)
        rc = lean ? (lazy ? launch(matrix_rowblock_kernel<Op, true, true>) : launch(matrix_rowblock_kernel<Op, true, false>))
                  : (lazy ? launch(matrix_rowblock_kernel<Op, false, true>) : launch(matrix_rowblock_kernel<Op, false, false>));
      else if constexpr (CAN_LEAN)
        rc = lean ? launch(matrix_rowblock_kernel<Op, true, false>) : launch(matrix_rowblock_kernel<Op, false, false>);
      else if constexpr (Op::LAZY)
        rc = lazy ? launch(matrix_rowblock_kernel<Op, false, true>) : launch(matrix_rowblock_kernel<Op, false, false>);
      else
        rc = launch(matrix_rowblock_kernel<Op, false, false>);
      if (rc)
        return rc;
      }
    }
    else
    {
      hipLaunchKernelGGL(matrix_atomic_kernel<Op>, dim3(grid_for(a.n_entities, 256)), dim3(256), 0, stream, a);
    }
    if (int rc = check(hipGetLastError(), "matrix kernel launch"))
      return rc;
  }
  if (a.block_vals && a.val_map)
  {
    mpcx_set_error("mpcx_assemble_matrix: val_map with block_vals (block-scalar storage has no CSR write-out to redirect)");
    return -3;
  }
  if (a.block_vals && (!a.slot_mask || alg != MPCX_ALG_ROWBLOCK || (a.n_slave_entities > 0 && (!a.mpc_plan_off || !a.mpc_plan_out))))
  {
    mpcx_set_error("mpcx_assemble_matrix: block_vals needs the node-block kernel (slot_mask, MPCX_ALG_ROWBLOCK) and, with slave "
                   "entities, a master-contribution plan with mpc_plan_out");
    return -8;
  }
  if (a.n_slave_entities > 0)
  {
    if (a.mpc_plan_off)
    {
      if (a.mpc_plan_targets > 0)
      {
        static const bool force_big = std::getenv("MPCX_PLAN_KERNEL_BIG") != nullptr;
        if (Op::SIZE <= 36 && !force_big)
          hipLaunchKernelGGL(matrix_mpc_plan_small_kernel<Op>, dim3(grid_for(a.mpc_plan_targets, 64)), dim3(64), 0, stream, a);
        else
        {
          bool lazy = false;
          if constexpr (Op::LAZY)
            lazy = Op::lazy_applies(a.kernel);
          const int g = a.mpc_plan_group >= 16 ? 16 : (a.mpc_plan_group >= 4 ? 4 : 1);
          const dim3 grid(grid_for(a.mpc_plan_targets * g, 64));
          auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, dim3(64), 0, stream, a); };
          if constexpr (Op::LAZY)
          {
            if (lazy)
              g == 16 ? go(matrix_mpc_plan_kernel<Op, 16, true>)
                      : (g == 4 ? go(matrix_mpc_plan_kernel<Op, 4, true>) : go(matrix_mpc_plan_kernel<Op, 1, true>));
          }
          if (!lazy)
            g == 16 ? go(matrix_mpc_plan_kernel<Op, 16, false>)
                    : (g == 4 ? go(matrix_mpc_plan_kernel<Op, 4, false>) : go(matrix_mpc_plan_kernel<Op, 1, false>));
        }
      }
    }
    else
      hipLaunchKernelGGL(matrix_mpc_kernel<Op>, dim3(grid_for(a.n_slave_entities, 64)), dim3(64), 0, stream, a);
    if (int rc = check(hipGetLastError(), "matrix mpc kernel launch"))
      return rc;
  }
  return 0;
}

int launch_vector_spill_reduce(const mpcx_vector_args_t& a, int bs)
{
  hipLaunchKernelGGL(vector_spill_reduce_kernel, dim3(grid_for(a.n_own_rows * bs, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(a.stream), a.n_own_rows, a.own_rows, a.own_seg, a.own_src, a.own_spill, bs, a.b,
                     a.row_map);
  return check(hipGetLastError(), "vector spill-reduce kernel launch");
}

template <class Op>
int launch_vector(const mpcx_vector_args_t& a)
{
  if (a.nd != Op::ND0 || a.bs != Op::BS0 || a.nv != Op::NV)
  {
    mpcx_set_error("mpcx_assemble_vector: dofmap shape does not match the element kernel");
    return -12;
  }
  if (a.n_entities == 0)
    return 0;
  hipStream_t stream = static_cast<hipStream_t>(a.stream);
  int alg = a.algorithm;
  if (alg == MPCX_ALG_AUTO)
    alg = a.plan.num_blocks > 0 ? MPCX_ALG_ROWBLOCK : MPCX_ALG_ATOMIC;
  if (alg == MPCX_ALG_ROWBLOCK)
  {
    if (a.plan.num_blocks <= 0 || (!a.mdofmap && !a.own_lmap))
    {
      mpcx_set_error("mpcx_assemble_vector: row-block algorithm needs a plan and the slave-masked dofmap");
      return -3;
    }
    size_t lds = size_t(a.plan.max_rows) * 8;
    if (lds > 96 * 1024)
    {
      mpcx_set_error("mpcx_assemble_vector: row-block plan exceeds the LDS budget");
      return -4;
    }
    if (a.lds_floor > 0 && size_t(a.lds_floor) > lds && a.lds_floor <= 160 * 1024)
      lds = size_t(a.lds_floor);
    const unsigned grid = 8u * unsigned((a.plan.num_blocks + 7) / 8);
    constexpr bool BY_COMPONENT = Op::BS0 > 1 && Op::ND0 >= 6 && Op::FORM == MPCX_FORM_SOURCE;
    bool split = false;
    if constexpr (BY_COMPONENT)
      split = a.coeffs == nullptr;
    auto launch = [&](auto kernel) -> int
    {
      if (int rc = check(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)),
                         "hipFuncSetAttribute"))
        return rc;
      // 512 threads (measured: 1024 threads for the large owner-computes blocks, one workgroup per CU by LDS, lose --
      // P2 source 246^3 5.70 -> 6.26 ms, Stokes b0 1.47 -> 1.60 ms)
      const int env_threads = []
      {
        const char* e = std::getenv("MPCX_VECTOR_THREADS");
        const int t = e ? std::atoi(e) : 0;
        return (t >= 64 && t <= 1024 && t % 64 == 0) ? t : 0;
      }();
      const int threads = env_threads ? env_threads : 512;
      hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, stream, a);
      return 0;
    };
    int lrc = 0;
    const bool owner = a.own_lmap != nullptr;
    if (owner && (!a.own_hoff || !a.own_spill || !a.own_seg || (a.n_own_rows > 0 && (!a.own_rows || !a.own_src))))
    {
      mpcx_set_error("mpcx_assemble_vector: incomplete owner-computes plan");
      return -5;
    }
    bool affine_done = false;
    if constexpr (Op::FORM == MPCX_FORM_SOURCE && (Op::DEG0 == 1 || Op::DEG0 == 2) && Op::BS0 == 1 && Op::TDIM == 3 && Op::FN == 1)
    {
      // the right-hand side from per-interval tables of the mesh's tensor grid, per cell (mpcx_vector_args_t::grid_eta / grid_J)
      if (owner && a.grid_idx && a.grid_J)
      {
        if (!a.grid_eta || !a.grid_iv || !a.grid_tab || !a.grid_block_rows || a.grid_ng <= 0 || a.grid_ng > 255 || a.grid_ntypes <= 0
            || a.grid_ntypes > 65535 || a.grid_block_rows_max <= 0 || a.coeffs
            || a.estride != 1 || a.entities || a.kernel.coeff_degree != 0 || a.kernel.nq <= 0 || (Op::DEG0 == 2 && !a.kernel.qphi))
        {
          mpcx_set_error("mpcx_assemble_vector: grid_J needs grid_eta / grid_iv / grid_tab / grid_block_rows, all cells, no coefficient");
          return -8;
        }
        const int ngp = (a.grid_ng + 1) & ~1, stride = 2 * ngp + 2;
        const int rows = a.grid_n[0] + a.grid_n[1] + a.grid_n[2];
        hipLaunchKernelGGL(cell_grid_tables_kernel, dim3(unsigned((int64_t(rows) * (ngp + 1) + 255) / 256)), dim3(256), 0, stream, a.grid_n[0],
                           a.grid_n[1], a.grid_n[2], a.grid_iv, a.grid_tab, a.grid_eta, a.grid_ng);
        const size_t glds = ((lds + 15) & ~size_t(15)) + size_t(a.grid_block_rows_max) * (stride + 2) * 8
                            + size_t(a.grid_ntypes) * a.kernel.nq * 4;
        if (glds > 160 * 1024)
        {
          mpcx_set_error("mpcx_assemble_vector: the blocks' table rows do not fit LDS beside their rows of b");
          return -4;
        }
        auto kernel = vector_cell_grid_kernel<Op::ND0>;
        if (int rc = check(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(glds)),
                           "hipFuncSetAttribute"))
          return rc;
        const char* e = std::getenv("MPCX_CELL_GRID_THREADS");
        int threads = e ? std::atoi(e) : 1024;
        if (threads < 64 || threads > 1024 || threads % 64)
          threads = 1024;
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), glds, stream, a);
        affine_done = true;
      }
    }
    if constexpr (Op::FORM == MPCX_FORM_SOURCE && (Op::DEG0 == 1 || Op::DEG0 == 2) && Op::FN != 1)
    {
      // integrand function affine in x (mpcx_kernel_t::vphi): the gather / LDS-add instance, 1024 threads
      // (MPCX_AFFINE_OWNBLOCK=0: the general instance; MPCX_AFFINE_THREADS)
      static const bool off = []
      {
        const char* e = std::getenv("MPCX_AFFINE_OWNBLOCK");
        return e && e[0] == '0';
      }();
      if (owner && a.kernel.vphi && !a.coeffs && a.estride == 1 && a.kernel.coeff_degree == 0 && !off)
      {
        auto kernel = vector_ownblock_affine_kernel<Op>;
        // two resident workgroups per CU (the blocks are ~60 KB of LDS) with as many waves as the registers allow
        hipFuncAttributes attr;
        if (int rc = check(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kernel)), "hipFuncGetAttributes"))
          return rc;
        const int alloc = ((attr.numRegs + 7) / 8) * 8;
        const int per_simd = std::max(2, std::min(8, 512 / std::max(alloc, 8)));
        const char* e = std::getenv("MPCX_AFFINE_THREADS");
        int threads = e ? std::atoi(e) : std::min(1024, 128 * per_simd);
        if (threads < 64 || threads > 1024 || threads % 64)
          threads = 1024;
        if (int rc = check(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)),
                           "hipFuncSetAttribute"))
          return rc;
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, stream, a);
        affine_done = true;
      }
    }
    if constexpr (BY_COMPONENT)
    {
      // (round 5, measured and not kept: all components in ONE pass over the points -- a third of the geometry / point work,
      // 3 x ND accumulators, compiled for 512 threads and up to 256 registers: Stokes b0 at 128^3 1.95 ms (512 threads),
      // 1.89 (256), 2.34 (384) against 1.47 ms by component)
      if (split && !affine_done)
        lrc = owner ? launch(vector_ownblock_kernel<Op, true>) : launch(vector_rowblock_kernel<Op, true>);
    }
    if (!split && !affine_done)
      lrc = owner ? launch(vector_ownblock_kernel<Op, false>) : launch(vector_rowblock_kernel<Op, false>);
    if (lrc)
      return lrc;
    if (owner && a.n_own_rows > 0)
      if (int rc = launch_vector_spill_reduce(a, Op::BS0))
        return rc;
    if (int rc = check(hipGetLastError(), "vector row-block kernel launch"))
      return rc;
    if (a.n_slave_entities > 0)
    {
      hipLaunchKernelGGL(vector_mpc_kernel<Op>, dim3(grid_for(a.n_slave_entities, VECTOR_MPC_THREADS)), dim3(VECTOR_MPC_THREADS), 0, stream, a);
      return check(hipGetLastError(), "vector mpc kernel launch");
    }
    return 0;
  }
  constexpr int NT = VectorCfg<Op::N0>::NT;
  hipLaunchKernelGGL(vector_kernel<Op>, dim3(grid_for(a.n_entities, NT)), dim3(NT), 0, stream, a);
  return check(hipGetLastError(), "vector kernel launch");
}

template <class Op>
int launch_slave_rows(const mpcx_vector_args_t& a)
{
  if (a.n_slave_entities <= 0)
    return 0;
  hipLaunchKernelGGL(vector_mpc_kernel<Op>, dim3(grid_for(a.n_slave_entities, VECTOR_MPC_THREADS)), dim3(VECTOR_MPC_THREADS), 0,
                     static_cast<hipStream_t>(a.stream), a);
  return check(hipGetLastError(), "vector mpc kernel launch");
}

// the cluster algorithm covers scalar P1 sources on tetrahedra only
int launch_vector_slave_rows(const mpcx_vector_args_t& a)
{
  if (a.kernel.fn_id == 1)
    return launch_slave_rows<ElementOp<3, 1, 1, 1, 1, MPCX_FORM_SOURCE, 1>>(a);
  return launch_slave_rows<ElementOp<3, 1, 1, 1, 1, MPCX_FORM_SOURCE>>(a);
}

template <class Op>
int launch_lifting(const mpcx_lifting_args_t& a)
{
  if (a.nd0 != Op::ND0 || a.nd1 != Op::ND1 || a.bs0 != Op::BS0 || a.bs1 != Op::BS1 || a.nv != Op::NV)
  {
    mpcx_set_error("mpcx_apply_lifting: dofmap shapes do not match the element kernel");
    return -12;
  }
  if (a.n_lift_entities == 0)
    return 0;
  hipLaunchKernelGGL(lifting_kernel<Op>, dim3(grid_for(a.n_lift_entities, 128)), dim3(128), 0,
                     static_cast<hipStream_t>(a.stream), a);
  return check(hipGetLastError(), "lifting kernel launch");
}

inline bool is_space(const mpcx_kernel_t& k, int cell, int deg, int bs)
{
  return k.celltype == cell && k.degree == deg && k.bs == bs && k.degree1 == deg && k.bs1 == bs;
}

// square (test == trial) scalar and P1-vector spaces compiled in
#define MPCX_FOR_SPACES(X, FORM)                                                                   \
  if (is_space(k, MPCX_CELL_TETRAHEDRON, 1, 1))                                                    \
    return X<ElementOp<3, 1, 1, 1, 1, FORM>>(a);                                                   \
  if (is_space(k, MPCX_CELL_TETRAHEDRON, 2, 1))                                                    \
    return X<ElementOp<3, 2, 1, 2, 1, FORM>>(a);                                                   \
  if (is_space(k, MPCX_CELL_TRIANGLE, 1, 1))                                                       \
    return X<ElementOp<2, 1, 1, 1, 1, FORM>>(a);                                                   \
  if (is_space(k, MPCX_CELL_TRIANGLE, 2, 1))                                                       \
    return X<ElementOp<2, 2, 1, 2, 1, FORM>>(a);                                                   \
  if (is_space(k, MPCX_CELL_TETRAHEDRON, 1, 3))                                                    \
    return X<ElementOp<3, 1, 3, 1, 3, FORM>>(a);                                                   \
  if (is_space(k, MPCX_CELL_TRIANGLE, 1, 2))                                                       \
    return X<ElementOp<2, 1, 2, 1, 2, FORM>>(a);

// P2 vector spaces (Taylor-Hood velocity): component-diagonal forms and sources only
#define MPCX_FOR_P2_VECTOR_SPACES(X, FORM)                                                         \
  if (is_space(k, MPCX_CELL_TETRAHEDRON, 2, 3))                                                    \
    return X<ElementOp<3, 2, 3, 2, 3, FORM>>(a);                                                   \
  if (is_space(k, MPCX_CELL_TRIANGLE, 2, 2))                                                       \
    return X<ElementOp<2, 2, 2, 2, 2, FORM>>(a);

#define MPCX_FOR_VECTOR_SPACES(X, FORM)                                                            \
  if (is_space(k, MPCX_CELL_TETRAHEDRON, 1, 3))                                                    \
    return X<ElementOp<3, 1, 3, 1, 3, FORM>>(a);                                                   \
  if (is_space(k, MPCX_CELL_TRIANGLE, 1, 2))                                                       \
    return X<ElementOp<2, 1, 2, 1, 2, FORM>>(a);

// velocity (P2 vector) x pressure (P1) blocks of Taylor-Hood
#define MPCX_FOR_DIV_TEST(X)                                                                       \
  if (k.celltype == MPCX_CELL_TETRAHEDRON && k.degree == 2 && k.bs == 3 && k.degree1 == 1 && k.bs1 == 1) \
    return X<ElementOp<3, 2, 3, 1, 1, MPCX_FORM_DIV_TEST>>(a);                                     \
  if (k.celltype == MPCX_CELL_TRIANGLE && k.degree == 2 && k.bs == 2 && k.degree1 == 1 && k.bs1 == 1) \
    return X<ElementOp<2, 2, 2, 1, 1, MPCX_FORM_DIV_TEST>>(a);
#define MPCX_FOR_DIV_TRIAL(X)                                                                      \
  if (k.celltype == MPCX_CELL_TETRAHEDRON && k.degree == 1 && k.bs == 1 && k.degree1 == 2 && k.bs1 == 3) \
    return X<ElementOp<3, 1, 1, 2, 3, MPCX_FORM_DIV_TRIAL>>(a);                                    \
  if (k.celltype == MPCX_CELL_TRIANGLE && k.degree == 1 && k.bs == 1 && k.degree1 == 2 && k.bs1 == 2) \
    return X<ElementOp<2, 1, 1, 2, 2, MPCX_FORM_DIV_TRIAL>>(a);

int unsupported(const mpcx_kernel_t& k)
{
  mpcx_set_error("unsupported element kernel: form " + std::to_string(k.form) + " celltype "
                 + std::to_string(k.celltype) + " test (degree " + std::to_string(k.degree) + ", bs "
                 + std::to_string(k.bs) + ") trial (degree " + std::to_string(k.degree1) + ", bs "
                 + std::to_string(k.bs1) + ")");
  return -10;
}

} // namespace mpcx

using namespace mpcx;

extern "C" int mpcx_assemble_matrix(const mpcx_matrix_args_t* args)
{
  const mpcx_matrix_args_t& a = *args;
  const mpcx_kernel_t& k = a.kernel;
  if (k.scalar_type != MPCX_SCALAR_F64)
    return launch_matrix_scalar(a);
  if (k.form == MPCX_FORM_UFCX)
    return launch_matrix_ufcx(a);
  if (k.celltype == MPCX_CELL_HEXAHEDRON)
  {
    // hexahedra exist as cluster kernels only (bulk of the cells; the caller assembles the master contributions with
    // the imported kernel of the same integral)
    if (a.algorithm != MPCX_ALG_CUBE || a.n_slave_entities != 0)
    {
      mpcx_set_error("mpcx_assemble_matrix: built-in hexahedron operators run with MPCX_ALG_CUBE and n_slave_entities = 0");
      return -10;
    }
    return launch_matrix_cubes(a);
  }
  switch (k.form)
  {
  case MPCX_FORM_STIFFNESS:
    MPCX_FOR_SPACES(launch_matrix, MPCX_FORM_STIFFNESS)
    MPCX_FOR_P2_VECTOR_SPACES(launch_matrix, MPCX_FORM_STIFFNESS)
    break;
  case MPCX_FORM_MASS:
    MPCX_FOR_SPACES(launch_matrix, MPCX_FORM_MASS)
    MPCX_FOR_P2_VECTOR_SPACES(launch_matrix, MPCX_FORM_MASS)
    break;
  case MPCX_FORM_FACET_MASS:
    MPCX_FOR_SPACES(launch_matrix, MPCX_FORM_FACET_MASS)
    break;
  case MPCX_FORM_ELASTICITY:
    MPCX_FOR_VECTOR_SPACES(launch_matrix, MPCX_FORM_ELASTICITY)
    MPCX_FOR_P2_VECTOR_SPACES(launch_matrix, MPCX_FORM_ELASTICITY)
    break;
  case MPCX_FORM_DIV_TEST:
    MPCX_FOR_DIV_TEST(launch_matrix)
    break;
  case MPCX_FORM_DIV_TRIAL:
    MPCX_FOR_DIV_TRIAL(launch_matrix)
    break;
  default:
    break;
  }
  return unsupported(k);
}

extern "C" int mpcx_assemble_vector(const mpcx_vector_args_t* args)
{
  const mpcx_vector_args_t& a = *args;
  const mpcx_kernel_t& k = a.kernel;
  if (k.scalar_type != MPCX_SCALAR_F64)
    return launch_vector_scalar(a);
  if (k.form == MPCX_FORM_UFCX)
    return launch_vector_ufcx(a);
  if (a.algorithm == MPCX_ALG_CUBE)
    return launch_vector_cubes(a);
  switch (k.form)
  {
  case MPCX_FORM_SOURCE:
    // the periodic benchmark's right-hand side (bench_periodic.py:85-91) gets its own instantiation
    if (is_space(k, MPCX_CELL_TETRAHEDRON, 1, 1) && k.fn_id == 1)
      return launch_vector<ElementOp<3, 1, 1, 1, 1, MPCX_FORM_SOURCE, 1>>(a);
    if (is_space(k, MPCX_CELL_TETRAHEDRON, 2, 1) && k.fn_id == 1)
      return launch_vector<ElementOp<3, 2, 1, 2, 1, MPCX_FORM_SOURCE, 1>>(a);
    // a constant body force on P2 vector spaces (the Taylor-Hood momentum right-hand side, config 3): with the function id
    // known at compile time the map of the quadrature points to physical space drops out of the loop
    if (is_space(k, MPCX_CELL_TETRAHEDRON, 2, 3) && k.fn_id == 5 && a.coeffs == nullptr)
      return launch_vector<ElementOp<3, 2, 3, 2, 3, MPCX_FORM_SOURCE, 5>>(a);
    MPCX_FOR_SPACES(launch_vector, MPCX_FORM_SOURCE)
    MPCX_FOR_P2_VECTOR_SPACES(launch_vector, MPCX_FORM_SOURCE)
    break;
  case MPCX_FORM_FACET_SOURCE:
    MPCX_FOR_SPACES(launch_vector, MPCX_FORM_FACET_SOURCE)
    break;
  default:
    break;
  }
  return unsupported(k);
}

extern "C" int mpcx_apply_lifting(const mpcx_lifting_args_t* args)
{
  const mpcx_lifting_args_t& a = *args;
  const mpcx_kernel_t& k = a.kernel;
  if (k.scalar_type != MPCX_SCALAR_F64)
    return launch_lifting_scalar(a);
  if (k.form == MPCX_FORM_UFCX)
    return launch_lifting_ufcx(a);
  switch (k.form)
  {
  case MPCX_FORM_STIFFNESS:
    MPCX_FOR_SPACES(launch_lifting, MPCX_FORM_STIFFNESS)
    MPCX_FOR_P2_VECTOR_SPACES(launch_lifting, MPCX_FORM_STIFFNESS)
    break;
  case MPCX_FORM_MASS:
    MPCX_FOR_SPACES(launch_lifting, MPCX_FORM_MASS)
    MPCX_FOR_P2_VECTOR_SPACES(launch_lifting, MPCX_FORM_MASS)
    break;
  case MPCX_FORM_FACET_MASS:
    MPCX_FOR_SPACES(launch_lifting, MPCX_FORM_FACET_MASS)
    break;
  case MPCX_FORM_ELASTICITY:
    MPCX_FOR_VECTOR_SPACES(launch_lifting, MPCX_FORM_ELASTICITY)
    MPCX_FOR_P2_VECTOR_SPACES(launch_lifting, MPCX_FORM_ELASTICITY)
    break;
  case MPCX_FORM_DIV_TEST:
    MPCX_FOR_DIV_TEST(launch_lifting)
    break;
  case MPCX_FORM_DIV_TRIAL:
    MPCX_FOR_DIV_TRIAL(launch_lifting)
    break;
  default:
    break;
  }
  return unsupported(k);
}

extern "C" int mpcx_add_diagonal(int32_t nrows, const mpcx_nnz_t* rowptr, const int32_t* cols,
                                 double* vals, const int32_t* dofs, int64_t n, double diagval,
                                 void* stream)
{
  (void)nrows;
  if (n == 0)
    return 0;
  return mpcx_add_diagonal_mapped(nrows, rowptr, cols, vals, dofs, n, diagval, nullptr, 0, stream);
}

extern "C" int mpcx_add_diagonal_mapped(int32_t nrows, const mpcx_nnz_t* rowptr, const int32_t* cols, double* vals,
                                        const int32_t* dofs, int64_t n, double diagval, const void* val_map, int32_t val_map_wide,
                                        void* stream)
{
  (void)nrows;
  if (n == 0)
    return 0;
  hipLaunchKernelGGL(add_diagonal_kernel, dim3(grid_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), rowptr, cols, vals,
                     dofs, n, diagval, val_map, int(val_map_wide));
  return check(hipGetLastError(), "add_diagonal launch");
}

extern "C" int mpcx_backsubstitution(double* u, const int32_t* slaves, int64_t num_slaves,
                                     const mpcx_mpc_t* mpc, void* stream)
{
  if (num_slaves == 0)
    return 0;
  hipLaunchKernelGGL(backsubstitution_kernel, dim3(grid_for(num_slaves, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), u, slaves, num_slaves, *mpc);
  return check(hipGetLastError(), "backsubstitution launch");
}

extern "C" int mpcx_homogenize(double* u, const int32_t* slaves, int64_t num_slaves, void* stream)
{
  if (num_slaves == 0)
    return 0;
  hipLaunchKernelGGL(homogenize_kernel, dim3(grid_for(num_slaves, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), u, slaves, num_slaves);
  return check(hipGetLastError(), "homogenize launch");
}

extern "C" int mpcx_diag_slot_mask(int32_t n_nodes, const mpcx_nnz_t* rowptr, const int32_t* cols, int32_t bs,
                                   const int8_t* bc0, const int8_t* slave0, const int8_t* bc1, const int8_t* slave1,
                                   uint8_t* out, int32_t* bad, void* stream)
{
  if (n_nodes <= 0)
    return 0;
  hipLaunchKernelGGL(mpcx::diag_slot_mask_kernel, dim3(mpcx::grid_for(n_nodes, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), n_nodes, rowptr, cols, bs, bc0, slave0, bc1, slave1, out, bad);
  return mpcx::check(hipGetLastError(), "diag_slot_mask launch");
}

extern "C" int mpcx_mask_dofmap(const int32_t* dofmap, int64_t num_cells, int32_t nd, int32_t bs,
                                const int8_t* bc, const int8_t* is_slave, int32_t rotate, int32_t* out,
                                void* stream)
{
  const int64_t n = num_cells * nd;
  if (n == 0)
    return 0;
  if (bs > 3)
  {
    mpcx_set_error("mpcx_mask_dofmap: block size > 3");
    return -6;
  }
  hipLaunchKernelGGL(mask_dofmap_kernel, dim3(grid_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     dofmap, n, nd, bs, bc, is_slave, rotate, out);
  return check(hipGetLastError(), "mask_dofmap launch");
}

extern "C" int mpcx_scatter_offsets(const mpcx_nnz_t* rowptr, const int32_t* cols, int32_t estride,
                                    int64_t n_entities, const int32_t* entities0,
                                    const int32_t* entities1, const int32_t* dofmap0, int32_t nd0,
                                    int32_t bs0, const int32_t* dofmap1, int32_t nd1, int32_t bs1,
                                    int32_t rotate, uint8_t* ent_offs, int32_t* overflow, void* stream)
{
  if (n_entities == 0)
    return 0;
  hipLaunchKernelGGL(scatter_offsets_kernel, dim3(grid_for(n_entities * nd0, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), rowptr, cols, estride, n_entities, entities0, entities1,
                     dofmap0, nd0, bs0, dofmap1, nd1, bs1, rotate, ent_offs, overflow);
  return check(hipGetLastError(), "scatter_offsets launch");
}

extern "C" int mpcx_pattern_device_adjacency(int64_t num_cells, const int32_t* dofmap0, int32_t nd0, int32_t bs0,
                                             const int32_t* c2s_offsets0, const int32_t* c2s0,
                                             const int32_t* masters_offsets0, const int32_t* masters0,
                                             const int64_t* adj_off, int32_t* counter, int32_t* adj, void* stream)
{
  if (num_cells == 0)
    return 0;
  hipLaunchKernelGGL(pattern_adj_kernel, dim3(grid_for(num_cells, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), num_cells, dofmap0, nd0, bs0, c2s_offsets0, c2s0,
                     masters_offsets0, masters0, adj_off, counter, adj);
  return check(hipGetLastError(), "pattern_adj launch");
}

extern "C" int mpcx_pattern_device_rows(int32_t num_blocks0, const int64_t* adj_off, const int32_t* adj,
                                        const int32_t* dofmap1, int32_t nd1, int32_t bs1,
                                        const int32_t* c2s_offsets1, const int32_t* c2s1,
                                        const int32_t* masters_offsets1, const int32_t* masters1,
                                        int32_t* row_count, const mpcx_nnz_t* rowptr, int32_t bs0, int32_t* cols,
                                        int32_t* overflow, void* stream)
{
  if (num_blocks0 == 0)
    return 0;
  const dim3 grid(grid_for(num_blocks0, PATTERN_THREADS));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (cols)
    hipLaunchKernelGGL(pattern_rows_kernel<true>, grid, dim3(PATTERN_THREADS), 0, st, num_blocks0, adj_off, adj,
                       dofmap1, nd1, bs1, c2s_offsets1, c2s1, masters_offsets1, masters1, row_count, rowptr, bs0,
                       cols, overflow);
  else
    hipLaunchKernelGGL(pattern_rows_kernel<false>, grid, dim3(PATTERN_THREADS), 0, st, num_blocks0, adj_off, adj,
                       dofmap1, nd1, bs1, c2s_offsets1, c2s1, masters_offsets1, masters1, row_count, rowptr, bs0,
                       cols, overflow);
  return check(hipGetLastError(), "pattern_rows launch");
}

extern "C" int mpcx_mpc_plan_device(int64_t n_slave_entities, const int32_t* slave_entities, int32_t estride,
                                    const int32_t* entities0, const int32_t* entities1, const int32_t* dofmap0,
                                    int32_t nd0, int32_t bs0, const int32_t* dofmap1, int32_t nd1, int32_t bs1,
                                    const int8_t* bc0, const int8_t* bc1, const mpcx_mpc_t* mpc0, const mpcx_mpc_t* mpc1,
                                    const mpcx_nnz_t* rowptr, const int32_t* cols, int32_t diag, int64_t* counts,
                                    const int64_t* offsets, int64_t* pos, int32_t* ent, int32_t* pq, double* coef,
                                    void* stream)
{
  if (n_slave_entities == 0)
    return 0;
  if (nd0 * bs0 > MPC_PLAN_MAXN || nd1 * bs1 > MPC_PLAN_MAXN)
  {
    mpcx_set_error("mpcx_mpc_plan_device: more than 32 unrolled dofs per element side");
    return -7;
  }
  const dim3 grid(grid_for(n_slave_entities, 64));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!offsets)
    hipLaunchKernelGGL(mpc_plan_device_kernel<false>, grid, dim3(64), 0, st, n_slave_entities, slave_entities, estride,
                       entities0, entities1, dofmap0, nd0, bs0, dofmap1, nd1, bs1, bc0, bc1, *mpc0, *mpc1, rowptr, cols,
                       diag, counts, offsets, pos, ent, pq, coef);
  else
    hipLaunchKernelGGL(mpc_plan_device_kernel<true>, grid, dim3(64), 0, st, n_slave_entities, slave_entities, estride,
                       entities0, entities1, dofmap0, nd0, bs0, dofmap1, nd1, bs1, bc0, bc1, *mpc0, *mpc1, rowptr, cols,
                       diag, counts, offsets, pos, ent, pq, coef);
  return check(hipGetLastError(), "mpc_plan_device launch");
}

extern "C" int mpcx_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

// Code objects are loaded at the first launch from each translation unit of the library (tens of MB in all: on a cold box
// the first assembly paid for it).  mpcx_preload launches one empty kernel per unit on `stream`; the Python
// layer calls it from a background thread when the library is first loaded on a machine with a device, so the loads run
// beside the host-side problem set-up.
namespace
{
__global__ void preload_kernels_kernel() {}
} // namespace
extern "C" int mpcx_preload_cubes(void*);
extern "C" int mpcx_preload_pairs(void*);
extern "C" int mpcx_preload_prims(void*);
extern "C" int mpcx_preload_plans(void*);
extern "C" int mpcx_preload_cluster_plan(void*);
extern "C" int mpcx_preload_solver(void*);
extern "C" int mpcx_preload(void* stream)
{
  hipLaunchKernelGGL(preload_kernels_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream));
  int rc = hipGetLastError() == hipSuccess ? 0 : -100;
  rc |= mpcx_preload_cubes(stream) | mpcx_preload_pairs(stream) | mpcx_preload_prims(stream) | mpcx_preload_plans(stream)
        | mpcx_preload_cluster_plan(stream) | mpcx_preload_solver(stream);
  if (rc == 0 && hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess)
    rc = -100;
  return rc;
}
