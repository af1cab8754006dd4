// Scalar types other than fp64-real: float32, complex128, complex64 (the reference instantiates the whole path for
// T in {f32, f64, c64, c128}: cpp/assemble_matrix.cpp:729-812, cpp/assemble_vector.cpp:263-297, cpp/lifting.h:441-670, with
// the Hermitian conjugate of the ROW-side coefficients for complex T, cpp/assemble_matrix.cpp:219-223,
// cpp/assemble_vector.h:59-65).
//
// These are the GENERAL kernels -- one thread per entity, CSR binary search, device atomics (the plan-free algorithm of
// mpcx_kernels.hip restated over a scalar type); the tuned fp64 kernels (LDS row blocks, clusters, pairs) stay fp64-real.
// mpcx_assemble_matrix / mpcx_assemble_vector / mpcx_apply_lifting come here when mpcx_kernel_t::scalar_type != 0; the
// double* fields of the argument structs then point to arrays of that scalar type (vals, b, coeffs, constants, the
// constraint's coefficients, bc_values1, x0).
//
// Element tensors.  The built-in operators are multilinear in their data (scale constant, coefficient function, vector
// constant of FN_CONSTANT_VEC; elasticity: linear in (mu, lambda)) with REAL geometry and basis functions, so a complex
// tensor is a combination of real ones: the fp64 ElementOp::tabulate is evaluated on the real / imaginary parts of the data
// (one to four evaluations per entity) and combined -- no second implementation of the integrals.  float32 / complex64:
// the data is widened, the tensor is computed in fp64 and every scatter-add rounds to the storage type (more accurate than
// fp32 arithmetic throughout; tolerance stated in tests/test_gpu_scalar_types.py).
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <string>

#include "mpcx_elements.hpp"
#include "mpcx_internal.h"

namespace mpcx
{
namespace
{
struct cplx
{
  double re, im;
};
__device__ inline cplx operator+(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ inline cplx operator-(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ inline cplx operator*(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ inline cplx operator*(cplx a, double s) { return {a.re * s, a.im * s}; }
__device__ inline cplx wconj(cplx a) { return {a.re, -a.im}; }
__device__ inline double wconj(double a) { return a; }
__device__ inline bool wnonzero(cplx a) { return a.re != 0.0 || a.im != 0.0; }
__device__ inline bool wnonzero(double a) { return a != 0.0; }

// storage traits: T = element type in memory, W = arithmetic type
// LT = the type of the LDS copy of a row block: always fp64 (pairs).  ds_add_f32 takes ~116 cycles per wave instruction on
// gfx950 against ~15 for ds_add_f64 (round 5, SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS of the float32 and complex64 matrix kernels:
// LDS 86 % busy without bank conflicts), so float32 / complex64 accumulate in fp64 and are rounded once, at the write-out.
struct SF64
{
  using T = double;
  using W = double;
  using LT = double;
  static constexpr bool COMPLEX = false;
  __device__ static W load(const T* p) { return *p; }
  __device__ static void store(T* p, W v) { *p = v; }
  __device__ static void atomic_add(T* p, W v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __device__ static W zero() { return 0.0; }
};
struct SF32
{
  using T = float;
  using W = double;
  using LT = double;
  static constexpr bool COMPLEX = false;
  __device__ static W load(const T* p) { return double(*p); }
  __device__ static void store(T* p, W v) { *p = float(v); }
  __device__ static void atomic_add(T* p, W v) { __hip_atomic_fetch_add(p, float(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __device__ static W zero() { return 0.0; }
};
struct SC128
{
  using T = double2;
  using W = cplx;
  using LT = double2;
  static constexpr bool COMPLEX = true;
  __device__ static W load(const T* p) { return {p->x, p->y}; }
  __device__ static void store(T* p, W v) { *p = make_double2(v.re, v.im); }
  __device__ static void atomic_add(T* p, W v)
  {
    double* q = reinterpret_cast<double*>(p);
    __hip_atomic_fetch_add(q, v.re, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(q + 1, v.im, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ static W zero() { return {0.0, 0.0}; }
};
struct SC64
{
  using T = float2;
  using W = cplx;
  using LT = double2;
  static constexpr bool COMPLEX = true;
  __device__ static W load(const T* p) { return {double(p->x), double(p->y)}; }
  __device__ static void store(T* p, W v) { *p = make_float2(float(v.re), float(v.im)); }
  __device__ static void atomic_add(T* p, W v)
  {
    float* q = reinterpret_cast<float*>(p);
    __hip_atomic_fetch_add(q, float(v.re), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(q + 1, float(v.im), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ static W zero() { return {0.0, 0.0}; }
};

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
__device__ inline void gather_coords(const double* __restrict__ x, const int32_t* __restrict__ x_dofmap, int64_t cell,
                                     double (&cd)[NV * 3])
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

constexpr int MAX_CSTRIDE = 96;
// element tensors up to this many entries are kept in registers: every loop over them is unrolled (a rolled loop indexes
// the tensor at run time and sends it to scratch memory -- round 5: float32 row blocks at 128^3 1.55 -> see DESIGN)
constexpr int SC_UNROLL_MAX = 144; // packed coefficient values per entity the scalar path accepts (three P2^3 fields)

// element tensor of one entity in W from data of type T (see the header)
// HASW = false: the form has no coefficient (w is NULL) -- the staging array of the coefficient values, indexed at run time,
// is what put 800 bytes of scratch memory per lane under every instance (round 5: the row-block kernels are instantiated
// without it for forms without coefficients)
template <class Op, class S, bool HASW = true>
__device__ inline bool tabulate_w(typename S::W* A, const typename S::T* w, int cstride, const typename S::T* c,
                                  const double (&cd)[Op::NV * 3], int lf, const mpcx_kernel_t& k)
{
  using W = typename S::W;
  constexpr int SIZE = Op::SIZE;
  constexpr int UN = SIZE <= SC_UNROLL_MAX ? SIZE : 1; // unroll count of the loops over the tensor
  constexpr bool ELAST = Op::FORM == MPCX_FORM_ELASTICITY;
  const bool vecconst = Op::RANK1 && k.fn_id == 5; // f = constants[1 : 1 + bs]
  if (HASW && cstride > MAX_CSTRIDE)
    return false;
  double Ar[SIZE];
  double wr[HASW ? MAX_CSTRIDE : 1];
  double cr[2 + Op::BS0];
  if constexpr (!HASW)
    w = nullptr;
  if constexpr (!S::COMPLEX)
  {
    if constexpr (HASW)
      for (int i = 0; i < cstride; ++i)
        wr[i] = S::load(w + i);
    const int nc = ELAST ? 2 : (vecconst ? 1 + Op::BS0 : 1);
    // (a constant trip count and an unconditional pointer: cr stays in registers.  No constants = the factor 1, which is what
    // ElementOp::tabulate takes for a NULL pointer; the forms that read further constants are never given NULL)
#pragma unroll
    for (int i = 0; i < 2 + Op::BS0; ++i)
      cr[i] = (c && i < nc) ? double(S::load(c + i)) : (i == 0 ? 1.0 : 0.0);
    Op::tabulate(Ar, w ? wr : nullptr, cr, cd, lf, k);
#pragma unroll UN
    for (int i = 0; i < SIZE; ++i)
      A[i] = Ar[i];
    return true;
  }
  else
  {
#pragma unroll UN
    for (int i = 0; i < SIZE; ++i)
      A[i] = S::zero();
    if constexpr (ELAST)
    {
      // A = mu T(1, 0) + lambda T(0, 1)
      for (int part = 0; part < 2; ++part)
      {
        cr[0] = part == 0 ? 1.0 : 0.0;
        cr[1] = part == 0 ? 0.0 : 1.0;
        Op::tabulate(Ar, nullptr, cr, cd, lf, k);
        const W f = S::load(c + part);
#pragma unroll UN
        for (int i = 0; i < SIZE; ++i)
          A[i] = A[i] + f * Ar[i];
      }
      return true;
    }
    else
    {
      // A = c0 * sum over (real / imaginary part of the coefficient) x (real / imaginary part of the vector constant)
      const W c0 = c ? S::load(c) : W{1.0, 0.0};
      const int nw = w ? 2 : 1, ng = (vecconst && c) ? 2 : 1;
      for (int pw = 0; pw < nw; ++pw)
        for (int pg = 0; pg < ng; ++pg)
        {
          if constexpr (HASW)
            if (w)
              for (int i = 0; i < cstride; ++i)
              {
                const W v = S::load(w + i);
                wr[i] = pw == 0 ? v.re : v.im;
              }
          cr[0] = 1.0;
          if (vecconst && c)
            for (int b = 0; b < Op::BS0; ++b)
            {
              const W v = S::load(c + 1 + b);
              cr[1 + b] = pg == 0 ? v.re : v.im;
            }
          Op::tabulate(Ar, w ? wr : nullptr, cr, cd, lf, k);
          // factor i^(number of imaginary parts taken)
          const int ni = pw + pg;
          const W f = ni == 0 ? W{1.0, 0.0} : (ni == 1 ? W{0.0, 1.0} : W{-1.0, 0.0});
          const W cf = c0 * f;
#pragma unroll UN
          for (int i = 0; i < SIZE; ++i)
            A[i] = A[i] + cf * Ar[i];
        }
      return true;
    }
  }
}

template <class Op, class W>
__device__ inline W get_w(const W* A, int p, int q)
{
  if constexpr (Op::DIAG)
    return (p % Op::BS0) == (q % Op::BS1) ? A[(p / Op::BS0) * Op::ND1 + q / Op::BS1] : W{};
  else
    return A[p * Op::N1 + q];
}

// ---------------------------------------------------------------------------------------------------------------
// matrix: cpp/assemble_matrix.cpp:488-547 + modify_mpc_cell (:99-268) per entity, device atomics
// ---------------------------------------------------------------------------------------------------------------
// PART 0: the whole entity (plan-free algorithm); PART 1: only the master contributions, over a.slave_entities (after the
// LDS row-block kernel below has written the entities' own blocks)
template <class Op, class S, int PART>
__global__ void __launch_bounds__(64) matrix_scalar_kernel(mpcx_matrix_args_t a, int32_t* __restrict__ fail)
{
  using T = typename S::T;
  using W = typename S::W;
  constexpr int N0 = Op::N0, N1 = Op::N1, ND0 = Op::ND0, ND1 = Op::ND1, BS0 = Op::BS0, BS1 = Op::BS1, NV = Op::NV;
  fastmath_init_lds();
  const int64_t t0 = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t0 >= (PART == 0 ? a.n_entities : a.n_slave_entities))
    return;
  const int64_t e = PART == 0 ? t0 : int64_t(a.slave_entities[t0]);
  const int64_t l = e * a.estride;
  const int64_t cell = (a.entities ? a.entities[l] : e);
  const int64_t cell0 = (a.entities0 ? a.entities0[l] : e);
  const int64_t cell1 = (a.entities1 ? a.entities1[l] : e);
  const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
  T* vals = reinterpret_cast<T*>(a.vals);
  const T* coeffs = reinterpret_cast<const T*>(a.coeffs);
  const T* mc0 = reinterpret_cast<const T*>(a.mpc0.coeffs);
  const T* mc1 = reinterpret_cast<const T*>(a.mpc1.coeffs);

  double cd[NV * 3];
  gather_coords<NV>(a.x, a.x_dofmap, cell, cd);
  W Ae[Op::SIZE];
  if (!tabulate_w<Op, S>(Ae, coeffs ? coeffs + e * a.cstride : nullptr, a.cstride, reinterpret_cast<const T*>(a.constants), cd, lf,
                         a.kernel))
  {
    *fail = 1;
    return;
  }
  int32_t rows[N0], colsd[N1];
  bool rbc[N0], cbc[N1], rsl[N0], csl[N1];
  bool any_slave = false;
  for (int i = 0; i < ND0; ++i)
  {
    const int32_t d0 = a.dofmap0[cell0 * ND0 + i];
    for (int k = 0; k < BS0; ++k)
    {
      const int32_t r = d0 * BS0 + k;
      rows[i * BS0 + k] = r;
      rbc[i * BS0 + k] = a.bc0 && a.bc0[r];
      rsl[i * BS0 + k] = a.mpc0.is_slave[r];
      any_slave = any_slave || rsl[i * BS0 + k];
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
      any_slave = any_slave || csl[j * BS1 + k];
    }
  }
  // Dirichlet rows / columns are zeroed before the MPC modification (:510-533)
  auto entry = [&](int p, int q) -> W { return (rbc[p] || cbc[q]) ? S::zero() : get_w<Op, W>(Ae, p, q); };
  auto add = [&](int32_t row, int32_t col, W v)
  {
    const int64_t pos = csr_find(a.cols, a.rowptr[row], a.rowptr[row + 1], col);
    if (pos >= 0)
      S::atomic_add(vals + pos, v);
  };
  // the entity's own block with slave rows / columns zeroed (:165-178, :546)
  if constexpr (PART == 0)
    for (int p = 0; p < N0; ++p)
    {
      if (rbc[p] || rsl[p])
        continue;
      for (int q = 0; q < N1; ++q)
      {
        if (cbc[q] || csl[q])
          continue;
        if constexpr (Op::DIAG)
        {
          if ((p % BS0) != (q % BS1))
            continue;
        }
        add(rows[p], colsd[q], get_w<Op, W>(Ae, p, q));
      }
    }
  if (!any_slave)
    return;
  // row masters (:214-246): Hermitian transpose -- the row-side coefficient is conjugated for complex T
  for (int p = 0; p < N0; ++p)
  {
    if (!rsl[p])
      continue;
    for (int mi = a.mpc0.masters_offsets[rows[p]]; mi < a.mpc0.masters_offsets[rows[p] + 1]; ++mi)
    {
      const int32_t m = a.mpc0.masters[mi];
      const W ci = wconj(S::load(mc0 + mi));
      for (int q = 0; q < N1; ++q)
      {
        const W v = entry(p, q);
        if (csl[q])
        {
          // master-master term from the un-stripped tensor (:239-245)
          for (int mj = a.mpc1.masters_offsets[colsd[q]]; mj < a.mpc1.masters_offsets[colsd[q] + 1]; ++mj)
            add(m, a.mpc1.masters[mj], ci * S::load(mc1 + mj) * v);
        }
        else if (!cbc[q])
          add(m, colsd[q], ci * v); // stripped row (:226-236)
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
      const W cj = S::load(mc1 + mj);
      for (int p = 0; p < N0; ++p)
      {
        if (rsl[p] || rbc[p])
          continue;
        add(rows[p], m, cj * entry(p, q));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// LDS row blocks for the other scalar types (MPCX_ALG_ROWBLOCK with a plain entity plan: plan.row_pairs == 0, masked
// dofmaps, the 8-bit scatter-offset table): one workgroup per contiguous CSR row range held in LDS in the STORAGE type
// (complex: two adds per entry), every entity touching the block evaluated, rows outside masked, one coalesced write --
// the formulation of matrix_rowblock_kernel without its fp64-only shortcuts (lean path, lazy entries, register tensors).
// ---------------------------------------------------------------------------------------------------------------
template <class S>
__device__ inline void lds_add(typename S::LT* p, typename S::W v)
{
  if constexpr (S::COMPLEX)
  {
    double* q = reinterpret_cast<double*>(p);
    __hip_atomic_fetch_add(q, v.re, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(q + 1, v.im, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  else
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <class S>
__device__ inline typename S::W lds_load(const typename S::LT* p)
{
  if constexpr (S::COMPLEX)
    return {p->x, p->y};
  else
    return *p;
}
template <class S>
__device__ inline void lds_zero(typename S::LT* p)
{
  if constexpr (S::COMPLEX)
    *p = make_double2(0.0, 0.0);
  else
    *p = 0.0;
}

constexpr int SC_MASK_SHIFT = 28;
constexpr int SC_DOF_MASK = (1 << SC_MASK_SHIFT) - 1;

constexpr int SC_ROWBLOCK_THREADS = 512; // launch bound; 256 threads are launched (MPCX_SCALAR_THREADS: 512 measured +6 % for
// float32 and -8 % for the complex types at 128^3, 1024 -- registers capped at 128 -- spills for complex128; round 5)
template <class Op, class S, bool HASW>
__global__ void __launch_bounds__(SC_ROWBLOCK_THREADS) matrix_rowblock_scalar_kernel(mpcx_matrix_args_t a, int32_t* __restrict__ fail)
{
  using T = typename S::T;
  using W = typename S::W;
  constexpr int ND0 = Op::ND0, ND1 = Op::ND1, BS0 = Op::BS0, BS1 = Op::BS1, NV = Op::NV;
  extern __shared__ __align__(16) unsigned char smem[];
  fastmath_init_lds();
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (b >= nb)
    return;
  const int tid = threadIdx.x, NT = blockDim.x;
  const int r0 = a.plan.block_row0[b], r1 = a.plan.block_row0[b + 1];
  const int nrow = r1 - r0;
  const int64_t nnz0 = a.rowptr[r0];
  const int nnzb = int(a.rowptr[r1] - nnz0);
  using LT = typename S::LT;
  LT* s_vals = reinterpret_cast<LT*>(smem);
  int32_t* s_rowlo = reinterpret_cast<int32_t*>(s_vals + a.plan.max_nnz);
  for (int i = tid; i < nnzb; i += NT)
    lds_zero<S>(s_vals + i);
  for (int rl = tid; rl <= nrow; rl += NT)
    s_rowlo[rl] = int(a.rowptr[r0 + rl] - nnz0);
  __syncthreads();
  T* vals = reinterpret_cast<T*>(a.vals);
  const T* coeffs = reinterpret_cast<const T*>(a.coeffs);
  const int64_t e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  for (int64_t t = e0 + tid; t < e1; t += NT)
  {
    const int64_t e = a.plan.block_ents[t];
    const int64_t l = e * a.estride;
    const int64_t cell = (a.entities ? a.entities[l] : e);
    const int64_t cell0 = (a.entities0 ? a.entities0[l] : e);
    const int64_t cell1 = (a.entities1 ? a.entities1[l] : e);
    const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
    constexpr bool SMALL = Op::SIZE <= SC_UNROLL_MAX;
    const uint8_t* __restrict__ po = a.plan.ent_offs + e * (ND0 * ND1);
    // the masked dofmap rows and the scatter offsets of the entity, loaded up front (SMALL: the unrolled case): inside the
    // row / column tests below every load sat behind a branch and was waited for on its own -- ~40 serialised memory
    // latencies per P1 entity, 1.3-2.3 ms for the 12.6 M cells of a 128^3 Poisson matrix against 0.27 ms for the float64
    // row-block kernel (round 5)
    int32_t m0v[SMALL ? ND0 : 1], m1v[SMALL ? ND1 : 1];
    uint8_t pov[SMALL ? ND0 * ND1 : 1];
    if constexpr (SMALL)
    {
#pragma unroll
      for (int i = 0; i < ND0; ++i)
        m0v[i] = a.mdofmap0[cell0 * ND0 + i];
#pragma unroll
      for (int j = 0; j < ND1; ++j)
        m1v[j] = a.mdofmap1[cell1 * ND1 + j];
#pragma unroll
      for (int i = 0; i < ND0 * ND1; ++i)
        pov[i] = po[i];
    }
    double cd[NV * 3];
    gather_coords<NV>(a.x, a.x_dofmap, cell, cd);
    W Ae[Op::SIZE];
    if (!tabulate_w<Op, S, HASW>(Ae, coeffs ? coeffs + e * a.cstride : nullptr, a.cstride, reinterpret_cast<const T*>(a.constants), cd,
                                 lf, a.kernel))
    {
      *fail = 1;
      continue;
    }
#pragma unroll(SMALL ? ND0 : 1)
    for (int i = 0; i < ND0; ++i)
    {
      int32_t m0;
      if constexpr (SMALL)
        m0 = m0v[i];
      else
        m0 = a.mdofmap0[cell0 * ND0 + i];
#pragma unroll(SMALL ? BS0 : 1)
      for (int k = 0; k < BS0; ++k)
      {
        const int r = (m0 & SC_DOF_MASK) * BS0 + k;
        if (r < r0 || r >= r1 || ((m0 >> (SC_MASK_SHIFT + k)) & 1))
          continue;
        LT* row = s_vals + s_rowlo[r - r0];
#pragma unroll(SMALL ? ND1 : 1)
        for (int j = 0; j < ND1; ++j)
        {
          int32_t m1;
          int off;
          if constexpr (SMALL)
          {
            m1 = m1v[j];
            off = int(pov[i * ND1 + j]) * BS1;
          }
          else
          {
            m1 = a.mdofmap1[cell1 * ND1 + j];
            off = int(po[i * ND1 + j]) * BS1;
          }
#pragma unroll(SMALL ? BS1 : 1)
          for (int q = 0; q < BS1; ++q)
          {
            if ((m1 >> (SC_MASK_SHIFT + q)) & 1)
              continue;
            if constexpr (Op::DIAG)
            {
              if (k != q)
                continue;
            }
            lds_add<S>(row + off + q, get_w<Op, W>(Ae, i * BS0 + k, j * BS1 + q));
          }
        }
      }
    }
  }
  __syncthreads();
  if (a.store_mode)
    for (int i = tid; i < nnzb; i += NT)
      S::store(vals + nnz0 + i, lds_load<S>(s_vals + i));
  else
    for (int i = tid; i < nnzb; i += NT)
      S::store(vals + nnz0 + i, S::load(vals + nnz0 + i) + lds_load<S>(s_vals + i));
}

// vector row blocks: the rows of b a workgroup owns live in LDS; halo entities are evaluated by every block they touch;
// slave rows are masked here and moved to their masters by vector_scalar_kernel<PART 1> over the slave entities
template <class Op, class S, bool HASW>
__global__ void __launch_bounds__(SC_ROWBLOCK_THREADS) vector_rowblock_scalar_kernel(mpcx_vector_args_t a, int32_t* __restrict__ fail)
{
  using T = typename S::T;
  using W = typename S::W;
  constexpr int ND = Op::ND0, BS = Op::BS0, NV = Op::NV;
  extern __shared__ __align__(16) unsigned char smem[];
  fastmath_init_lds();
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (b >= nb)
    return;
  const int tid = threadIdx.x, NT = blockDim.x;
  const int r0 = a.plan.block_row0[b], r1 = a.plan.block_row0[b + 1];
  typename S::LT* s_b = reinterpret_cast<typename S::LT*>(smem);
  for (int i = tid; i < r1 - r0; i += NT)
    lds_zero<S>(s_b + i);
  __syncthreads();
  T* bg = reinterpret_cast<T*>(a.b);
  const T* coeffs = reinterpret_cast<const T*>(a.coeffs);
  const int64_t e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  for (int64_t t = e0 + tid; t < e1; t += NT)
  {
    const int64_t e = a.plan.block_ents[t];
    const int64_t l = e * a.estride;
    const int64_t cell = (a.entities ? a.entities[l] : e);
    const int64_t cell0 = (a.entities0 ? a.entities0[l] : e);
    const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
    double cd[NV * 3];
    gather_coords<NV>(a.x, a.x_dofmap, cell, cd);
    int32_t m0v[ND]; // (loaded before the element vector is formed: see matrix_rowblock_scalar_kernel)
#pragma unroll
    for (int i = 0; i < ND; ++i)
      m0v[i] = a.mdofmap[cell0 * ND + i];
    W be[Op::N0];
    if (!tabulate_w<Op, S, HASW>(be, coeffs ? coeffs + e * a.cstride : nullptr, a.cstride, reinterpret_cast<const T*>(a.constants), cd,
                                 lf, a.kernel))
    {
      *fail = 1;
      continue;
    }
#pragma unroll
    for (int i = 0; i < ND; ++i)
    {
      const int32_t m0 = m0v[i];
#pragma unroll
      for (int k = 0; k < BS; ++k)
      {
        const int r = (m0 & SC_DOF_MASK) * BS + k;
        if (r < r0 || r >= r1 || ((m0 >> (SC_MASK_SHIFT + k)) & 1))
          continue;
        lds_add<S>(s_b + (r - r0), be[i * BS + k]);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < r1 - r0; i += NT)
    S::store(bg + r0 + i, S::load(bg + r0 + i) + lds_load<S>(s_b + i));
}

// ---------------------------------------------------------------------------------------------------------------
// vector: cpp/assemble_vector.cpp:34-91 + modify_mpc_vec (cpp/assemble_vector.h:35-69)
// ---------------------------------------------------------------------------------------------------------------
// PART 0: the whole entity; PART 1: only the slave rows (to their masters), over a.slave_entities
template <class Op, class S, int PART>
__global__ void __launch_bounds__(64) vector_scalar_kernel(mpcx_vector_args_t a, int32_t* __restrict__ fail)
{
  using T = typename S::T;
  using W = typename S::W;
  constexpr int N = Op::N0, ND = Op::ND0, BS = Op::BS0, NV = Op::NV;
  fastmath_init_lds();
  const int64_t t0 = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t0 >= (PART == 0 ? a.n_entities : a.n_slave_entities))
    return;
  const int64_t e = PART == 0 ? t0 : int64_t(a.slave_entities[t0]);
  const int64_t l = e * a.estride;
  const int64_t cell = (a.entities ? a.entities[l] : e);
  const int64_t cell0 = (a.entities0 ? a.entities0[l] : e);
  const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
  T* b = reinterpret_cast<T*>(a.b);
  const T* coeffs = reinterpret_cast<const T*>(a.coeffs);
  const T* mc = reinterpret_cast<const T*>(a.mpc.coeffs);
  double cd[NV * 3];
  gather_coords<NV>(a.x, a.x_dofmap, cell, cd);
  W be[N];
  if (!tabulate_w<Op, S>(be, coeffs ? coeffs + e * a.cstride : nullptr, a.cstride, reinterpret_cast<const T*>(a.constants), cd, lf,
                         a.kernel))
  {
    *fail = 1;
    return;
  }
  for (int i = 0; i < ND; ++i)
  {
    const int32_t d0 = a.dofmap[cell0 * ND + i];
    for (int k = 0; k < BS; ++k)
    {
      const int32_t d = d0 * BS + k;
      W v = be[i * BS + k];
      if (a.mpc.is_slave[d])
      {
        const int m0 = a.mpc.masters_offsets[d], m1 = a.mpc.masters_offsets[d + 1];
        for (int mi = m0; mi < m1; ++mi)
          S::atomic_add(b + a.mpc.masters[mi], wconj(S::load(mc + mi)) * v);
        if (m1 > m0)
          continue; // be[slave] = 0 (inside the master loop of the reference: a slave without masters keeps its entry)
        if constexpr (PART == 1)
          S::atomic_add(b + d, v); // (masked in the row-block kernel)
      }
      if constexpr (PART == 0)
        S::atomic_add(b + d, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// lifting: cpp/lifting.h:77-133
// ---------------------------------------------------------------------------------------------------------------
template <class Op, class S>
__global__ void __launch_bounds__(64) lifting_scalar_kernel(mpcx_lifting_args_t a, int32_t* __restrict__ fail)
{
  using T = typename S::T;
  using W = typename S::W;
  constexpr int N0 = Op::N0, ND0 = Op::ND0, ND1 = Op::ND1, BS0 = Op::BS0, BS1 = Op::BS1, NV = Op::NV;
  fastmath_init_lds();
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= a.n_lift_entities)
    return;
  const int64_t e = a.lift_entities[t];
  const int64_t l = e * a.estride;
  const int64_t cell = (a.entities ? a.entities[l] : e);
  const int64_t cell0 = (a.entities0 ? a.entities0[l] : e);
  const int64_t cell1 = (a.entities1 ? a.entities1[l] : e);
  const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
  T* b = reinterpret_cast<T*>(a.b);
  const T* coeffs = reinterpret_cast<const T*>(a.coeffs);
  const T* mc = reinterpret_cast<const T*>(a.mpc0.coeffs);
  const T* g1 = reinterpret_cast<const T*>(a.bc_values1);
  const T* x0 = reinterpret_cast<const T*>(a.x0);
  double cd[NV * 3];
  gather_coords<NV>(a.x, a.x_dofmap, cell, cd);
  W Ae[Op::SIZE];
  if (!tabulate_w<Op, S>(Ae, coeffs ? coeffs + e * a.cstride : nullptr, a.cstride, reinterpret_cast<const T*>(a.constants), cd, lf,
                         a.kernel))
  {
    *fail = 1;
    return;
  }
  W be[N0];
  for (int m = 0; m < N0; ++m)
    be[m] = S::zero();
  for (int j = 0; j < ND1; ++j)
  {
    const int32_t d1 = a.dofmap1[cell1 * ND1 + j];
    for (int k = 0; k < BS1; ++k)
    {
      const int32_t jj = d1 * BS1 + k;
      if (a.bc_markers1[jj])
      {
        W g = S::load(g1 + jj);
        if (x0)
          g = g - S::load(x0 + jj);
        g = g * a.scale;
        for (int m = 0; m < N0; ++m)
          be[m] = be[m] - get_w<Op, W>(Ae, m, j * BS1 + k) * g;
      }
    }
  }
  for (int i = 0; i < ND0; ++i)
  {
    const int32_t d0 = a.dofmap0[cell0 * ND0 + i];
    for (int k = 0; k < BS0; ++k)
    {
      const int32_t d = d0 * BS0 + k;
      const W v = be[i * BS0 + k];
      if (a.mpc0.is_slave[d])
      {
        const int m0 = a.mpc0.masters_offsets[d], m1 = a.mpc0.masters_offsets[d + 1];
        for (int mi = m0; mi < m1; ++mi)
          S::atomic_add(b + a.mpc0.masters[mi], wconj(S::load(mc + mi)) * v);
        if (m1 > m0)
          continue;
      }
      if (wnonzero(v))
        S::atomic_add(b + d, v);
    }
  }
}

template <class S>
__global__ void add_diagonal_scalar_kernel(const mpcx_nnz_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                           typename S::T* __restrict__ vals, const int32_t* __restrict__ dofs, int64_t n, double re,
                                           double im)
{
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const int32_t d = dofs[i];
  const int64_t pos = csr_find(cols, rowptr[d], rowptr[d + 1], d);
  if (pos < 0)
    return;
  if constexpr (S::COMPLEX)
    S::atomic_add(vals + pos, typename S::W{re, im});
  else
    S::atomic_add(vals + pos, re);
}

// cpp/MultiPointConstraint.h:129-152 (no conjugation: u_slave = sum c u_master)
template <class S>
__global__ void backsubstitution_scalar_kernel(typename S::T* u, const int32_t* __restrict__ slaves, int64_t n, mpcx_mpc_t mpc)
{
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const int32_t s = slaves[i];
  const typename S::T* mc = reinterpret_cast<const typename S::T*>(mpc.coeffs);
  typename S::W v = S::zero();
  for (int mi = mpc.masters_offsets[s]; mi < mpc.masters_offsets[s + 1]; ++mi)
    v = v + S::load(mc + mi) * S::load(u + mpc.masters[mi]);
  S::store(u + s, v);
}

template <class S>
__global__ void homogenize_scalar_kernel(typename S::T* u, const int32_t* __restrict__ slaves, int64_t n)
{
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n)
    S::store(u + slaves[i], S::zero());
}

// a device flag the kernels raise when they cannot represent a call (too many packed coefficient values)
int32_t* fail_flag(hipStream_t stream)
{
  static int32_t* flag = nullptr;
  if (!flag)
  {
    if (hipMalloc(&flag, sizeof(int32_t)) != hipSuccess)
      return nullptr;
  }
  (void)hipMemsetAsync(flag, 0, sizeof(int32_t), stream);
  return flag;
}

int read_fail(int32_t* flag, hipStream_t stream, const char* what)
{
  int32_t h = 0;
  if (int rc = check(hipMemcpyAsync(&h, flag, sizeof(int32_t), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync"))
    return rc;
  if (int rc = check(hipStreamSynchronize(stream), "hipStreamSynchronize"))
    return rc;
  if (h)
  {
    mpcx_set_error(std::string(what) + ": more than 96 packed coefficient values per entity on the scalar-type path");
    return -9;
  }
  return 0;
}

template <template <class, class> class K, class Op, class Args>
int launch_typed(const Args& a, int64_t n, const char* what)
{
  hipStream_t stream = static_cast<hipStream_t>(a.stream);
  if (n <= 0)
    return 0;
  int32_t* flag = fail_flag(stream);
  if (!flag)
  {
    mpcx_set_error("scalar-type path: hipMalloc failed");
    return -100;
  }
  const dim3 grid(grid_for(n, 64));
  switch (a.kernel.scalar_type)
  {
  case MPCX_SCALAR_F64:
    K<Op, SF64>::launch(grid, stream, a, flag);
    break;
  case MPCX_SCALAR_F32:
    K<Op, SF32>::launch(grid, stream, a, flag);
    break;
  case MPCX_SCALAR_C128:
    K<Op, SC128>::launch(grid, stream, a, flag);
    break;
  case MPCX_SCALAR_C64:
    K<Op, SC64>::launch(grid, stream, a, flag);
    break;
  default:
    mpcx_set_error("unknown scalar type");
    return -11;
  }
  if (int rc = check(hipGetLastError(), what))
    return rc;
  return read_fail(flag, stream, what);
}

inline int sc_rowblock_threads()
{
  static const int t = []
  {
    const char* e = std::getenv("MPCX_SCALAR_THREADS");
    const int v = e ? std::atoi(e) : 0;
    return (v >= 64 && v <= SC_ROWBLOCK_THREADS && v % 64 == 0) ? v : 256;
  }();
  return t;
}

template <class Op, class S>
struct MatrixK
{
  static void launch(dim3 grid, hipStream_t st, const mpcx_matrix_args_t& a, int32_t* flag)
  {
    if (a.algorithm == MPCX_ALG_ROWBLOCK && a.plan.num_blocks > 0)
    {
      const size_t lds = size_t(a.plan.max_nnz) * sizeof(typename S::LT) + size_t(a.plan.max_rows + 1) * 4 + 512;
      auto go = [&](auto kern)
      {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        hipLaunchKernelGGL(kern, dim3(8u * unsigned((a.plan.num_blocks + 7) / 8)), dim3(sc_rowblock_threads()), lds, st, a, flag);
      };
      if (a.coeffs)
        go(matrix_rowblock_scalar_kernel<Op, S, true>);
      else
        go(matrix_rowblock_scalar_kernel<Op, S, false>);
      if (a.n_slave_entities > 0)
        hipLaunchKernelGGL((matrix_scalar_kernel<Op, S, 1>), dim3(grid_for(a.n_slave_entities, 64)), dim3(64), 0, st, a, flag);
    }
    else
      hipLaunchKernelGGL((matrix_scalar_kernel<Op, S, 0>), grid, dim3(64), 0, st, a, flag);
  }
};
template <class Op, class S>
struct VectorK
{
  static void launch(dim3 grid, hipStream_t st, const mpcx_vector_args_t& a, int32_t* flag)
  {
    if (a.algorithm == MPCX_ALG_ROWBLOCK && a.plan.num_blocks > 0)
    {
      const size_t lds = size_t(a.plan.max_rows) * sizeof(typename S::LT) + 512;
      auto go = [&](auto kern)
      {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        hipLaunchKernelGGL(kern, dim3(8u * unsigned((a.plan.num_blocks + 7) / 8)), dim3(sc_rowblock_threads()), lds, st, a, flag);
      };
      if (a.coeffs)
        go(vector_rowblock_scalar_kernel<Op, S, true>);
      else
        go(vector_rowblock_scalar_kernel<Op, S, false>);
      if (a.n_slave_entities > 0)
        hipLaunchKernelGGL((vector_scalar_kernel<Op, S, 1>), dim3(grid_for(a.n_slave_entities, 64)), dim3(64), 0, st, a, flag);
    }
    else
      hipLaunchKernelGGL((vector_scalar_kernel<Op, S, 0>), grid, dim3(64), 0, st, a, flag);
  }
};
template <class Op, class S>
struct LiftingK
{
  static void launch(dim3 grid, hipStream_t st, const mpcx_lifting_args_t& a, int32_t* flag)
  {
    hipLaunchKernelGGL((lifting_scalar_kernel<Op, S>), grid, dim3(64), 0, st, a, flag);
  }
};

template <class Op>
int run_matrix(const mpcx_matrix_args_t& a)
{
  if (a.nd0 != Op::ND0 || a.nd1 != Op::ND1 || a.bs0 != Op::BS0 || a.bs1 != Op::BS1 || a.nv != Op::NV)
  {
    mpcx_set_error("mpcx_assemble_matrix: dofmap shapes do not match the element kernel");
    return -12;
  }
  return launch_typed<MatrixK, Op>(a, a.n_entities, "matrix_scalar_kernel");
}
template <class Op>
int run_vector(const mpcx_vector_args_t& a)
{
  if (a.nd != Op::ND0 || a.bs != Op::BS0 || a.nv != Op::NV)
  {
    mpcx_set_error("mpcx_assemble_vector: dofmap shape does not match the element kernel");
    return -12;
  }
  return launch_typed<VectorK, Op>(a, a.n_entities, "vector_scalar_kernel");
}
template <class Op>
int run_lifting(const mpcx_lifting_args_t& a)
{
  if (a.nd0 != Op::ND0 || a.nd1 != Op::ND1 || a.bs0 != Op::BS0 || a.bs1 != Op::BS1 || a.nv != Op::NV)
  {
    mpcx_set_error("mpcx_apply_lifting: dofmap shapes do not match the element kernel");
    return -12;
  }
  return launch_typed<LiftingK, Op>(a, a.n_lift_entities, "lifting_scalar_kernel");
}

inline bool is_space(const mpcx_kernel_t& k, int cell, int deg, int bs)
{
  return k.celltype == cell && k.degree == deg && k.bs == bs && k.degree1 == deg && k.bs1 == bs;
}

// square spaces of the built-in operators (the same list as mpcx_kernels.hip)
#define SC_SQUARE(X, FORM, a)                                                                                          \
  if (is_space(k, MPCX_CELL_TETRAHEDRON, 1, 1))                                                                        \
    return X<ElementOp<3, 1, 1, 1, 1, FORM>>(a);                                                                       \
  if (is_space(k, MPCX_CELL_TETRAHEDRON, 2, 1))                                                                        \
    return X<ElementOp<3, 2, 1, 2, 1, FORM>>(a);                                                                       \
  if (is_space(k, MPCX_CELL_TRIANGLE, 1, 1))                                                                           \
    return X<ElementOp<2, 1, 1, 1, 1, FORM>>(a);                                                                       \
  if (is_space(k, MPCX_CELL_TRIANGLE, 2, 1))                                                                           \
    return X<ElementOp<2, 2, 1, 2, 1, FORM>>(a);                                                                       \
  if (is_space(k, MPCX_CELL_TETRAHEDRON, 1, 3))                                                                        \
    return X<ElementOp<3, 1, 3, 1, 3, FORM>>(a);                                                                       \
  if (is_space(k, MPCX_CELL_TRIANGLE, 1, 2))                                                                           \
    return X<ElementOp<2, 1, 2, 1, 2, FORM>>(a);
#define SC_P2VEC(X, FORM, a)                                                                                           \
  if (is_space(k, MPCX_CELL_TETRAHEDRON, 2, 3))                                                                        \
    return X<ElementOp<3, 2, 3, 2, 3, FORM>>(a);                                                                       \
  if (is_space(k, MPCX_CELL_TRIANGLE, 2, 2))                                                                           \
    return X<ElementOp<2, 2, 2, 2, 2, FORM>>(a);
#define SC_VEC(X, FORM, a)                                                                                             \
  if (is_space(k, MPCX_CELL_TETRAHEDRON, 1, 3))                                                                        \
    return X<ElementOp<3, 1, 3, 1, 3, FORM>>(a);                                                                       \
  if (is_space(k, MPCX_CELL_TRIANGLE, 1, 2))                                                                           \
    return X<ElementOp<2, 1, 2, 1, 2, FORM>>(a);
#define SC_DIV(X, a)                                                                                                   \
  if (k.form == MPCX_FORM_DIV_TEST && k.celltype == MPCX_CELL_TETRAHEDRON && k.degree == 2 && k.bs == 3 && k.degree1 == 1     \
      && k.bs1 == 1)                                                                                                   \
    return X<ElementOp<3, 2, 3, 1, 1, MPCX_FORM_DIV_TEST>>(a);                                                         \
  if (k.form == MPCX_FORM_DIV_TEST && k.celltype == MPCX_CELL_TRIANGLE && k.degree == 2 && k.bs == 2 && k.degree1 == 1        \
      && k.bs1 == 1)                                                                                                   \
    return X<ElementOp<2, 2, 2, 1, 1, MPCX_FORM_DIV_TEST>>(a);                                                         \
  if (k.form == MPCX_FORM_DIV_TRIAL && k.celltype == MPCX_CELL_TETRAHEDRON && k.degree == 1 && k.bs == 1 && k.degree1 == 2    \
      && k.bs1 == 3)                                                                                                   \
    return X<ElementOp<3, 1, 1, 2, 3, MPCX_FORM_DIV_TRIAL>>(a);                                                        \
  if (k.form == MPCX_FORM_DIV_TRIAL && k.celltype == MPCX_CELL_TRIANGLE && k.degree == 1 && k.bs == 1 && k.degree1 == 2       \
      && k.bs1 == 2)                                                                                                   \
    return X<ElementOp<2, 1, 1, 2, 2, MPCX_FORM_DIV_TRIAL>>(a);

int unsupported(const mpcx_kernel_t& k)
{
  mpcx_set_error("scalar-type path: no built-in operator for form " + std::to_string(k.form) + " on this element (imported UFCx "
                 "kernels and hexahedra are fp64-real only)");
  return -10;
}
} // namespace

int launch_matrix_scalar(const mpcx_matrix_args_t& a)
{
  const mpcx_kernel_t& k = a.kernel;
  if (a.val_map)
  {
    mpcx_set_error("mpcx_assemble_matrix: val_map is honoured by the float64 kernels only");
    return -3;
  }
  switch (k.form)
  {
  case MPCX_FORM_STIFFNESS:
    SC_SQUARE(run_matrix, MPCX_FORM_STIFFNESS, a)
    SC_P2VEC(run_matrix, MPCX_FORM_STIFFNESS, a)
    break;
  case MPCX_FORM_MASS:
    SC_SQUARE(run_matrix, MPCX_FORM_MASS, a)
    SC_P2VEC(run_matrix, MPCX_FORM_MASS, a)
    break;
  case MPCX_FORM_FACET_MASS:
    SC_SQUARE(run_matrix, MPCX_FORM_FACET_MASS, a)
    break;
  case MPCX_FORM_ELASTICITY:
    SC_VEC(run_matrix, MPCX_FORM_ELASTICITY, a)
    SC_P2VEC(run_matrix, MPCX_FORM_ELASTICITY, a)
    break;
  case MPCX_FORM_DIV_TEST:
  case MPCX_FORM_DIV_TRIAL:
    SC_DIV(run_matrix, a)
    break;
  default:
    break;
  }
  return unsupported(k);
}

int launch_vector_scalar(const mpcx_vector_args_t& a)
{
  const mpcx_kernel_t& k = a.kernel;
  if (a.row_map)
  {
    mpcx_set_error("mpcx_assemble_vector: row_map is honoured by the float64 kernels only");
    return -3;
  }
  switch (k.form)
  {
  case MPCX_FORM_SOURCE:
    SC_SQUARE(run_vector, MPCX_FORM_SOURCE, a)
    SC_P2VEC(run_vector, MPCX_FORM_SOURCE, a)
    break;
  case MPCX_FORM_FACET_SOURCE:
    SC_SQUARE(run_vector, MPCX_FORM_FACET_SOURCE, a)
    break;
  default:
    break;
  }
  return unsupported(k);
}

int launch_lifting_scalar(const mpcx_lifting_args_t& a)
{
  const mpcx_kernel_t& k = a.kernel;
  if (a.row_map)
  {
    mpcx_set_error("mpcx_apply_lifting: row_map is honoured by the float64 kernels only");
    return -3;
  }
  switch (k.form)
  {
  case MPCX_FORM_STIFFNESS:
    SC_SQUARE(run_lifting, MPCX_FORM_STIFFNESS, a)
    SC_P2VEC(run_lifting, MPCX_FORM_STIFFNESS, a)
    break;
  case MPCX_FORM_MASS:
    SC_SQUARE(run_lifting, MPCX_FORM_MASS, a)
    SC_P2VEC(run_lifting, MPCX_FORM_MASS, a)
    break;
  case MPCX_FORM_FACET_MASS:
    SC_SQUARE(run_lifting, MPCX_FORM_FACET_MASS, a)
    break;
  case MPCX_FORM_ELASTICITY:
    SC_VEC(run_lifting, MPCX_FORM_ELASTICITY, a)
    SC_P2VEC(run_lifting, MPCX_FORM_ELASTICITY, a)
    break;
  case MPCX_FORM_DIV_TEST:
  case MPCX_FORM_DIV_TRIAL:
    SC_DIV(run_lifting, a)
    break;
  default:
    break;
  }
  return unsupported(k);
}
} // namespace mpcx

using namespace mpcx;

extern "C" int mpcx_add_diagonal_scalar(int32_t scalar_type, const mpcx_nnz_t* rowptr, const int32_t* cols, void* vals,
                                        const int32_t* dofs, int64_t n, double re, double im, void* stream)
{
  if (n <= 0)
    return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(grid_for(n, 256));
  switch (scalar_type)
  {
  case MPCX_SCALAR_F64:
    hipLaunchKernelGGL(add_diagonal_scalar_kernel<SF64>, grid, dim3(256), 0, st, rowptr, cols, static_cast<double*>(vals), dofs, n, re, im);
    break;
  case MPCX_SCALAR_F32:
    hipLaunchKernelGGL(add_diagonal_scalar_kernel<SF32>, grid, dim3(256), 0, st, rowptr, cols, static_cast<float*>(vals), dofs, n, re, im);
    break;
  case MPCX_SCALAR_C128:
    hipLaunchKernelGGL(add_diagonal_scalar_kernel<SC128>, grid, dim3(256), 0, st, rowptr, cols, static_cast<double2*>(vals), dofs, n, re, im);
    break;
  case MPCX_SCALAR_C64:
    hipLaunchKernelGGL(add_diagonal_scalar_kernel<SC64>, grid, dim3(256), 0, st, rowptr, cols, static_cast<float2*>(vals), dofs, n, re, im);
    break;
  default:
    mpcx_set_error("mpcx_add_diagonal_scalar: unknown scalar type");
    return -11;
  }
  return check(hipGetLastError(), "add_diagonal_scalar launch");
}

extern "C" int mpcx_backsubstitution_scalar(int32_t scalar_type, void* u, const int32_t* slaves, int64_t n, const mpcx_mpc_t* mpc,
                                            void* stream)
{
  if (n <= 0)
    return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(grid_for(n, 256));
  switch (scalar_type)
  {
  case MPCX_SCALAR_F64:
    hipLaunchKernelGGL(backsubstitution_scalar_kernel<SF64>, grid, dim3(256), 0, st, static_cast<double*>(u), slaves, n, *mpc);
    break;
  case MPCX_SCALAR_F32:
    hipLaunchKernelGGL(backsubstitution_scalar_kernel<SF32>, grid, dim3(256), 0, st, static_cast<float*>(u), slaves, n, *mpc);
    break;
  case MPCX_SCALAR_C128:
    hipLaunchKernelGGL(backsubstitution_scalar_kernel<SC128>, grid, dim3(256), 0, st, static_cast<double2*>(u), slaves, n, *mpc);
    break;
  case MPCX_SCALAR_C64:
    hipLaunchKernelGGL(backsubstitution_scalar_kernel<SC64>, grid, dim3(256), 0, st, static_cast<float2*>(u), slaves, n, *mpc);
    break;
  default:
    mpcx_set_error("mpcx_backsubstitution_scalar: unknown scalar type");
    return -11;
  }
  return check(hipGetLastError(), "backsubstitution_scalar launch");
}

extern "C" int mpcx_homogenize_scalar(int32_t scalar_type, void* u, const int32_t* slaves, int64_t n, void* stream)
{
  if (n <= 0)
    return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(grid_for(n, 256));
  switch (scalar_type)
  {
  case MPCX_SCALAR_F64:
    hipLaunchKernelGGL(homogenize_scalar_kernel<SF64>, grid, dim3(256), 0, st, static_cast<double*>(u), slaves, n);
    break;
  case MPCX_SCALAR_F32:
    hipLaunchKernelGGL(homogenize_scalar_kernel<SF32>, grid, dim3(256), 0, st, static_cast<float*>(u), slaves, n);
    break;
  case MPCX_SCALAR_C128:
    hipLaunchKernelGGL(homogenize_scalar_kernel<SC128>, grid, dim3(256), 0, st, static_cast<double2*>(u), slaves, n);
    break;
  case MPCX_SCALAR_C64:
    hipLaunchKernelGGL(homogenize_scalar_kernel<SC64>, grid, dim3(256), 0, st, static_cast<float2*>(u), slaves, n);
    break;
  default:
    mpcx_set_error("mpcx_homogenize_scalar: unknown scalar type");
    return -11;
  }
  return check(hipGetLastError(), "homogenize_scalar launch");
}
