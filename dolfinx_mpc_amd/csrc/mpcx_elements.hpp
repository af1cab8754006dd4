// Device element kernels for gfx950: the role FFCx-generated tabulate_tensor
// plays in the reference (called at cpp/assemble_matrix.cpp:505-506,
// cpp/assemble_vector.cpp:180-181, cpp/lifting.h:268-270).  Everything is
// compile-time sized so the element tensor lives in VGPRs.
//
// Conventions (Basix/UFC): reference simplex vertices (0,..),(1,0,..),(0,1,..);
// facet i opposite vertex i; P2 dofs = vertices then edges
// tet edges (2,3)(1,3)(1,2)(0,3)(0,2)(0,1), triangle edges (1,2)(0,2)(0,1).
// coordinate_dofs are [nv][3] (3 components always, assemble_matrix.cpp:499).
#pragma once
#include "mpcx.h"
#include "mpcx_fastmath.hpp"
#include <hip/hip_runtime.h>

namespace mpcx
{

template <int TDIM, int DEG>
struct Lagrange
{
  static constexpr int NV = TDIM + 1;
  static constexpr int NE = TDIM == 3 ? 6 : 3;
  static constexpr int ND = DEG == 1 ? NV : NV + NE;

  __device__ static inline void edge(int e, int& a, int& b)
  {
    if constexpr (TDIM == 3)
    {
      constexpr int E[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
      a = E[e][0];
      b = E[e][1];
    }
    else
    {
      constexpr int E[3][2] = {{1, 2}, {0, 2}, {0, 1}};
      a = E[e][0];
      b = E[e][1];
    }
  }

  // phi[ND], dphi[ND][TDIM] (reference gradients)
  __device__ static inline void eval(const double (&X)[3], double (&phi)[ND], double (&dphi)[ND][TDIM])
  {
    double lam[NV];
    lam[0] = 1.0;
#pragma unroll
    for (int d = 0; d < TDIM; ++d)
    {
      lam[0] -= X[d];
      lam[d + 1] = X[d];
    }
    // dlam[i][d] = (i == d+1) - (i == 0)
    if constexpr (DEG == 1)
    {
#pragma unroll
      for (int i = 0; i < NV; ++i)
      {
        phi[i] = lam[i];
#pragma unroll
        for (int d = 0; d < TDIM; ++d)
          dphi[i][d] = (i == 0) ? -1.0 : (i == d + 1 ? 1.0 : 0.0);
      }
    }
    else
    {
#pragma unroll
      for (int i = 0; i < NV; ++i)
      {
        phi[i] = lam[i] * (2.0 * lam[i] - 1.0);
        const double s = 4.0 * lam[i] - 1.0;
#pragma unroll
        for (int d = 0; d < TDIM; ++d)
          dphi[i][d] = (i == 0) ? -s : (i == d + 1 ? s : 0.0);
      }
#pragma unroll
      for (int e = 0; e < NE; ++e)
      {
        int a, b;
        edge(e, a, b);
        phi[NV + e] = 4.0 * lam[a] * lam[b];
#pragma unroll
        for (int d = 0; d < TDIM; ++d)
        {
          const double da = (a == 0) ? -1.0 : (a == d + 1 ? 1.0 : 0.0);
          const double db = (b == 0) ? -1.0 : (b == d + 1 ? 1.0 : 0.0);
          dphi[NV + e][d] = 4.0 * (lam[a] * db + lam[b] * da);
        }
      }
    }
  }
};

// Affine map: K = J^-1 (K[d][a] = dX_d/dx_a), detJ
template <int TDIM>
__device__ inline void affine_geometry(const double* cd, double (&K)[TDIM][TDIM], double& detJ)
{
  if constexpr (TDIM == 2)
  {
    const double J00 = cd[3] - cd[0], J01 = cd[6] - cd[0];
    const double J10 = cd[4] - cd[1], J11 = cd[7] - cd[1];
    const double det = J00 * J11 - J01 * J10;
    detJ = det;
    const double inv = 1.0 / det;
    K[0][0] = J11 * inv;
    K[0][1] = -J01 * inv;
    K[1][0] = -J10 * inv;
    K[1][1] = J00 * inv;
  }
  else
  {
    const double J00 = cd[3] - cd[0], J01 = cd[6] - cd[0], J02 = cd[9] - cd[0];
    const double J10 = cd[4] - cd[1], J11 = cd[7] - cd[1], J12 = cd[10] - cd[1];
    const double J20 = cd[5] - cd[2], J21 = cd[8] - cd[2], J22 = cd[11] - cd[2];
    const double c00 = J11 * J22 - J12 * J21;
    const double c01 = J12 * J20 - J10 * J22;
    const double c02 = J10 * J21 - J11 * J20;
    const double det = J00 * c00 + J01 * c01 + J02 * c02;
    detJ = det;
    const double inv = 1.0 / det; // one division; the products differ from x/det by <= 1 ulp
    K[0][0] = c00 * inv;
    K[0][1] = (J02 * J21 - J01 * J22) * inv;
    K[0][2] = (J01 * J12 - J02 * J11) * inv;
    K[1][0] = c01 * inv;
    K[1][1] = (J00 * J22 - J02 * J20) * inv;
    K[1][2] = (J02 * J10 - J00 * J12) * inv;
    K[2][0] = c02 * inv;
    K[2][1] = (J01 * J20 - J00 * J21) * inv;
    K[2][2] = (J00 * J11 - J01 * J10) * inv;
  }
}

// det(J) * grad(lambda_i) (cofactor rows; row 0 = minus the sum of the others)
template <int TDIM>
__device__ inline void cofactor_gradients(const double* cd, double (&G)[TDIM + 1][TDIM], double& det)
{
  if constexpr (TDIM == 2)
  {
    const double J00 = cd[3] - cd[0], J01 = cd[6] - cd[0];
    const double J10 = cd[4] - cd[1], J11 = cd[7] - cd[1];
    det = J00 * J11 - J01 * J10;
    G[1][0] = J11;
    G[1][1] = -J01;
    G[2][0] = -J10;
    G[2][1] = J00;
  }
  else
  {
    const double J00 = cd[3] - cd[0], J01 = cd[6] - cd[0], J02 = cd[9] - cd[0];
    const double J10 = cd[4] - cd[1], J11 = cd[7] - cd[1], J12 = cd[10] - cd[1];
    const double J20 = cd[5] - cd[2], J21 = cd[8] - cd[2], J22 = cd[11] - cd[2];
    G[1][0] = J11 * J22 - J12 * J21;
    G[1][1] = J02 * J21 - J01 * J22;
    G[1][2] = J01 * J12 - J02 * J11;
    G[2][0] = J12 * J20 - J10 * J22;
    G[2][1] = J00 * J22 - J02 * J20;
    G[2][2] = J02 * J10 - J00 * J12;
    G[3][0] = J10 * J21 - J11 * J20;
    G[3][1] = J01 * J20 - J00 * J21;
    G[3][2] = J00 * J11 - J01 * J10;
    det = J00 * G[1][0] + J01 * G[2][0] + J02 * G[3][0];
  }
#pragma unroll
  for (int a = 0; a < TDIM; ++a)
  {
    double s = 0.0;
#pragma unroll
    for (int d = 0; d < TDIM; ++d)
      s += G[d + 1][a];
    G[0][a] = -s;
  }
}

template <int TDIM>
__device__ inline void push_forward(const double* cd, const double (&X)[3], double (&x)[3])
{
  double l0 = 1.0;
#pragma unroll
  for (int d = 0; d < TDIM; ++d)
    l0 -= X[d];
#pragma unroll
  for (int i = 0; i < 3; ++i)
  {
    double v = l0 * cd[i];
#pragma unroll
    for (int d = 0; d < TDIM; ++d)
      v += X[d] * cd[3 * (d + 1) + i];
    x[i] = v;
  }
}

// analytic right-hand sides; ids shared with dolfinx_mpc_amd/fem.py
__device__ inline double eval_fn(int fn_id, const double (&x)[3], int comp, const double* c)
{
  switch (fn_id)
  {
  case 0:
    return 1.0;
  case 1:
  {
    // python/benchmarks/bench_periodic.py:85-89
    // sin(5 pi y) = sinpi(5 y): exact range reduction, no pi rounding in the argument
    const double dx = x[0] - 0.9, dy = x[1] - 0.5, dz = x[2] - 0.1;
    // (1/0.02 rounds to 50.0: the product differs from the quotient by <= 1 ulp of the exponent)
    return x[0] * fast_sinpi(5.0 * x[1]) + 1.0 * fast_exp_nonpos(-(dx * dx + dy * dy + dz * dz) * (1.0 / 0.02));
  }
  case 2:
    return fast_sinpi(2.0 * x[0]) * fast_sinpi(x[1]) + 0.3 * (comp + 1);
  case 3:
    return 1.0 + 2.0 * x[0] + 3.0 * x[1] * x[1] - x[2] * x[2] * x[2] + x[0] * x[1] * x[2]
           + 0.5 * comp * x[0];
  case 4:
    return (comp + 1) * (1.0 + x[0] - 2.0 * x[1] + 0.5 * x[2]);
  case 5:
    return c[1 + comp]; // constants = [scale, g_0, g_1, ...]
  default:
    return 0.0;
  }
}

// Reference point + weight of quadrature point q of a cell or facet rule.
template <int TDIM, bool FACET>
struct QuadPoint
{
  double fscale = 0.0;
  double fv[TDIM][TDIM]; // reference coords of the facet vertices

  __device__ inline void init_facet(const double* cd, int lf)
  {
    if constexpr (FACET)
    {
      double pv[TDIM][3];
#pragma unroll
      for (int a = 0; a < TDIM; ++a)
      {
        // facet lf = all vertices except lf, ascending
        const int v = a < lf ? a : a + 1;
#pragma unroll
        for (int d = 0; d < TDIM; ++d)
          fv[a][d] = (v == d + 1) ? 1.0 : 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
          pv[a][i] = cd[3 * v + i];
      }
      if constexpr (TDIM == 3)
      {
        const double e1x = pv[1][0] - pv[0][0], e1y = pv[1][1] - pv[0][1], e1z = pv[1][2] - pv[0][2];
        const double e2x = pv[2][0] - pv[0][0], e2y = pv[2][1] - pv[0][1], e2z = pv[2][2] - pv[0][2];
        const double cx = e1y * e2z - e1z * e2y, cy = e1z * e2x - e1x * e2z, cz = e1x * e2y - e1y * e2x;
        fscale = sqrt(cx * cx + cy * cy + cz * cz);
      }
      else
      {
        const double ex = pv[1][0] - pv[0][0], ey = pv[1][1] - pv[0][1], ez = pv[1][2] - pv[0][2];
        fscale = sqrt(ex * ex + ey * ey + ez * ez);
      }
    }
  }

  __device__ inline double point(const mpcx_kernel_t& k, int q, double adet, double (&X)[3]) const
  {
    X[0] = X[1] = X[2] = 0.0;
    if constexpr (FACET)
    {
      const double* s = k.fqpts + q * (TDIM - 1);
      double l0 = 1.0;
#pragma unroll
      for (int d = 0; d < TDIM - 1; ++d)
        l0 -= s[d];
#pragma unroll
      for (int d = 0; d < TDIM; ++d)
      {
        double v = l0 * fv[0][d];
#pragma unroll
        for (int a = 1; a < TDIM; ++a)
          v += s[a - 1] * fv[a][d];
        X[d] = v;
      }
      return k.fqwts[q] * fscale;
    }
    else
    {
#pragma unroll
      for (int d = 0; d < TDIM; ++d)
        X[d] = k.qpts[q * TDIM + d];
      return k.qwts[q] * adet;
    }
  }
};

template <int TDIM>
__device__ inline double eval_coefficient(int coeff_degree, const double* w, const double (&X)[3])
{
  if (coeff_degree == 1)
  {
    double phi[Lagrange<TDIM, 1>::ND], dphi[Lagrange<TDIM, 1>::ND][TDIM];
    Lagrange<TDIM, 1>::eval(X, phi, dphi);
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < Lagrange<TDIM, 1>::ND; ++i)
      v += w[i] * phi[i];
    return v;
  }
  else
  {
    double phi[Lagrange<TDIM, 2>::ND], dphi[Lagrange<TDIM, 2>::ND][TDIM];
    Lagrange<TDIM, 2>::eval(X, phi, dphi);
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < Lagrange<TDIM, 2>::ND; ++i)
      v += w[i] * phi[i];
    return v;
  }
}

// ---------------------------------------------------------------------------
// Generic element operator for a (test, trial) pair of Lagrange spaces.
//
// Test space: degree DEG0, block size BS0 (rows); trial space: DEG1, BS1 (cols);
// blocked dof index i*BS + k, row-major, like a UFCx tabulate_tensor output.
// Storage of the tensor A:
//   rank 1 (SOURCE, FACET_SOURCE)           : [N0]
//   component-diagonal forms with BS0 > 1
//   (STIFFNESS, MASS, FACET_MASS: entry
//   ((i,a),(j,b)) = delta_ab S_ij)            : the scalar [ND0][ND1] matrix S only
//   everything else                          : [N0][N1]
// get(A, p, q) hides the difference.  FN_ >= 0 fixes the analytic source
// function at compile time; FN_ = -1 reads kernel.fn_id.
// ---------------------------------------------------------------------------
template <int TDIM_, int DEG0_, int BS0_, int DEG1_, int BS1_, int FORM_, int FN_ = -1>
struct ElementOp
{
  static constexpr int TDIM = TDIM_;
  static constexpr int FORM = FORM_;
  static constexpr int DEG0 = DEG0_, FN = FN_;
  using L0 = Lagrange<TDIM, DEG0_>;
  using L1 = Lagrange<TDIM, DEG1_>;
  static constexpr int NV = TDIM + 1;
  static constexpr int ND0 = L0::ND, ND1 = L1::ND;
  static constexpr int BS0 = BS0_, BS1 = BS1_;
  static constexpr int N0 = ND0 * BS0, N1 = ND1 * BS1;
  static constexpr bool FACET = (FORM == MPCX_FORM_FACET_MASS || FORM == MPCX_FORM_FACET_SOURCE);
  static constexpr bool RANK1 = (FORM == MPCX_FORM_SOURCE || FORM == MPCX_FORM_FACET_SOURCE);
  static constexpr bool SQUARE = (DEG0_ == DEG1_ && BS0_ == BS1_);
  static constexpr bool DIAG
      = (FORM == MPCX_FORM_STIFFNESS || FORM == MPCX_FORM_MASS || FORM == MPCX_FORM_FACET_MASS) && BS0 > 1;
  static constexpr int SIZE = RANK1 ? N0 : (DIAG ? ND0 * ND1 : N0 * N1);
  static_assert(!(FORM == MPCX_FORM_STIFFNESS || FORM == MPCX_FORM_MASS || FORM == MPCX_FORM_FACET_MASS
                  || FORM == MPCX_FORM_ELASTICITY)
                    || SQUARE,
                "form needs test space == trial space");
  static_assert(FORM != MPCX_FORM_ELASTICITY || BS0 == TDIM, "elasticity needs bs == tdim");
  static_assert(FORM != MPCX_FORM_DIV_TEST || (BS0 == TDIM && BS1 == 1), "div(v) p: vector test, scalar trial");
  static_assert(FORM != MPCX_FORM_DIV_TRIAL || (BS0 == 1 && BS1 == TDIM), "div(u) q: scalar test, vector trial");

  // LAZY operators: the element tensor does not fit the register budget of the row-block kernel
  // (P1 elasticity: 144 entries = 288 VGPRs), but its entries have a closed form in a compact
  // context -- `prepare` fills the context once per entity, `entry` evaluates one entry where it
  // is scattered.  The thread-per-entity kernels keep using `tabulate`.
  //   P1 elasticity:  gradients + |T| mu, |T| lambda
  //   P2 stiffness (scalar, or component-diagonal on a blocked space), no coefficient:
  //                   G_kl = |T| grad(l_k).grad(l_l) of the barycentric coordinates; with
  //                   int l_a = |T|/4, int l_a l_b = |T| (1 + d_ab)/20 every entry is a fixed
  //                   combination of at most four G_kl
  static constexpr bool LAZY_ELASTICITY = (FORM == MPCX_FORM_ELASTICITY && (DEG0_ == 1 || DEG0_ == 2));
  static constexpr bool LAZY_P2_STIFFNESS = (FORM == MPCX_FORM_STIFFNESS && DEG0_ == 2 && DEG1_ == 2);
  //   Taylor-Hood coupling blocks (P2^d x P1): int psi_j d_a(phi_i) from the gradients and the
  //                   same barycentric integrals
  static constexpr bool LAZY_DIV_TEST = (FORM == MPCX_FORM_DIV_TEST && DEG0_ == 2 && DEG1_ == 1);
  static constexpr bool LAZY_DIV_TRIAL = (FORM == MPCX_FORM_DIV_TRIAL && DEG0_ == 1 && DEG1_ == 2);
  static constexpr bool LAZY_DIV = LAZY_DIV_TEST || LAZY_DIV_TRIAL;
  //   P1 stiffness:   cofactor rows C_k = det grad(l_k) and s = c0 / (d! |det|): A_ij = s C_i.C_j.  The 16
  //                   entries would fit, but without them the kernel needs 64 VGPRs instead of 128
  //                   (no spills, twice the waves per SIMD)
  static constexpr bool LAZY_P1_STIFFNESS = (FORM == MPCX_FORM_STIFFNESS && DEG0_ == 1 && DEG1_ == 1);
  static constexpr bool LAZY = LAZY_ELASTICITY || LAZY_P2_STIFFNESS || LAZY_DIV || LAZY_P1_STIFFNESS;
  struct Lazy
  {
    double g[NV][LAZY_P2_STIFFNESS ? NV : TDIM]; // elasticity: physical gradients; P2: G_kl
    double smu, sla;                             // |T| mu, |T| lambda / P2: scale c0, unused
  };
  // may the row-block kernel take the lazy path for this kernel descriptor?
  __device__ __host__ static inline bool lazy_applies(const mpcx_kernel_t& k)
  {
    return LAZY_ELASTICITY || ((LAZY_P2_STIFFNESS || LAZY_DIV || LAZY_P1_STIFFNESS) && k.coeff_degree == 0);
  }
  __device__ static inline void prepare(Lazy& L, const double* c, const double (&cd)[NV * 3])
  {
    if constexpr (LAZY_ELASTICITY || LAZY_DIV)
    {
      double K[TDIM][TDIM], detJ;
      affine_geometry<TDIM>(cd, K, detJ);
      const double vol = fabs(detJ) * (TDIM == 3 ? 1.0 / 6.0 : 0.5);
      if constexpr (LAZY_DIV)
      {
        L.smu = vol * (c ? c[0] : 1.0); // c0 |T|
        L.sla = 0.0;
      }
      else
      {
        L.smu = vol * c[0];
        L.sla = vol * c[1];
      }
#pragma unroll
      for (int a = 0; a < TDIM; ++a)
      {
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < TDIM; ++d)
        {
          L.g[d + 1][a] = K[d][a];
          s += K[d][a];
        }
        L.g[0][a] = -s;
      }
    }
    else if constexpr (LAZY_P1_STIFFNESS)
    {
      double det;
      cofactor_gradients<TDIM>(cd, L.g, det);
      L.smu = (c ? c[0] : 1.0) / ((TDIM == 3 ? 6.0 : 2.0) * fabs(det));
      L.sla = 0.0;
    }
    else
    {
      // cof_k = det * grad(l_k):  G_kl = |T| grad(l_k).grad(l_l) = cof_k.cof_l / (d! |det|)
      double C[NV][TDIM], det;
      cofactor_gradients<TDIM>(cd, C, det);
      const double s = (c ? c[0] : 1.0) / ((TDIM == 3 ? 6.0 : 2.0) * fabs(det));
#pragma unroll
      for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int l = k; l < NV; ++l)
        {
          double dot = 0.0;
#pragma unroll
          for (int d = 0; d < TDIM; ++d)
            dot += C[k][d] * C[l][d];
          L.g[k][l] = L.g[l][k] = s * dot;
        }
      L.smu = L.sla = 0.0;
    }
  }
  // Constant-free, compact form of the context: what matrix_pairs_kernel keeps per entity in HBM (cached per geometry
  // version) instead of recomputing the context for every (entity, local row) pair.  Row 0 of the gradients / of G is
  // minus the sum of the other rows (the barycentric coordinates sum to one), so only rows 1..TDIM travel:
  //   P2 stiffness:          G_kl / c0 for 1 <= k <= l <= TDIM                      (6 doubles in 3D)
  //   elasticity, div blocks: the inverse Jacobian (= gradients of l_1..l_TDIM) + |T|  (10 doubles)
  //   P1 stiffness:          cofactor rows 1..TDIM + 1 / (d! |det|)                 (10 doubles)
  static constexpr int CTXN = LAZY_P2_STIFFNESS ? TDIM * (TDIM + 1) / 2 : TDIM * TDIM + 1;
  __device__ static inline void ctx_store(double (&out)[CTXN], const double (&cd)[NV * 3])
  {
    if constexpr (LAZY_ELASTICITY || LAZY_DIV)
    {
      double K[TDIM][TDIM], detJ;
      affine_geometry<TDIM>(cd, K, detJ);
#pragma unroll
      for (int d = 0; d < TDIM; ++d)
#pragma unroll
        for (int a = 0; a < TDIM; ++a)
          out[d * TDIM + a] = K[d][a];
      out[TDIM * TDIM] = fabs(detJ) * (TDIM == 3 ? 1.0 / 6.0 : 0.5);
    }
    else
    {
      double C[NV][TDIM], det;
      cofactor_gradients<TDIM>(cd, C, det);
      const double s = 1.0 / ((TDIM == 3 ? 6.0 : 2.0) * fabs(det));
      if constexpr (LAZY_P1_STIFFNESS)
      {
#pragma unroll
        for (int k = 0; k < TDIM; ++k)
#pragma unroll
          for (int d = 0; d < TDIM; ++d)
            out[k * TDIM + d] = C[k + 1][d];
        out[TDIM * TDIM] = s;
      }
      else
      {
        int n = 0;
#pragma unroll
        for (int k = 1; k < NV; ++k)
#pragma unroll
          for (int l = k; l < NV; ++l)
          {
            double dot = 0.0;
#pragma unroll
            for (int d = 0; d < TDIM; ++d)
              dot += C[k][d] * C[l][d];
            out[n++] = s * dot;
          }
      }
    }
  }
  __device__ static inline void ctx_load(Lazy& L, const double* c, const double (&in)[CTXN])
  {
    if constexpr (LAZY_P2_STIFFNESS)
    {
      const double c0 = c ? c[0] : 1.0;
      int n = 0;
#pragma unroll
      for (int k = 1; k < NV; ++k)
#pragma unroll
        for (int l = k; l < NV; ++l)
          L.g[k][l] = L.g[l][k] = c0 * in[n++];
      double s00 = 0.0;
#pragma unroll
      for (int l = 1; l < NV; ++l)
      {
        double s = 0.0;
#pragma unroll
        for (int k = 1; k < NV; ++k)
          s += L.g[k][l];
        L.g[0][l] = L.g[l][0] = -s;
        s00 += s;
      }
      L.g[0][0] = s00;
      L.smu = L.sla = 0.0;
    }
    else
    {
#pragma unroll
      for (int a = 0; a < TDIM; ++a)
      {
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < TDIM; ++d)
        {
          L.g[d + 1][a] = in[d * TDIM + a];
          s += in[d * TDIM + a];
        }
        L.g[0][a] = -s;
      }
      const double w = in[TDIM * TDIM];
      if constexpr (LAZY_ELASTICITY)
      {
        L.smu = w * c[0];
        L.sla = w * c[1];
      }
      else
      {
        L.smu = w * (c ? c[0] : 1.0);
        L.sla = 0.0;
      }
    }
  }
  __device__ static inline double entry(const Lazy& L, int i, int a, int j, int b)
  {
    if constexpr (LAZY_ELASTICITY && DEG0_ == 2)
    {
      // dense P2^d elasticity (the `2 mu eps(u):eps(v)` block of python/demos/demo_stokes.py): with
      // grad(phi) = sum_k (d phi / d l_k) grad(l_k) every entry is  sum_{k,l} c_kl(i, j) H(k, l),
      //   H(k, l) = |T| (mu g_k^b g_l^a + lambda g_k^a g_l^b + delta_ab mu g_k.g_l),
      //   c_kl(i, j) = int (d phi_i / d l_k)(d phi_j / d l_l) / |T|   -- the same barycentric integrals as the
      // P2 stiffness entries above, but H is not symmetric in (k, l): k belongs to the test function i
      using LG = Lagrange<TDIM, 2>;
      constexpr double m1 = 1.0 / (TDIM + 1), m2 = 1.0 / ((TDIM + 1) * (TDIM + 2));
      auto d = [](int x, int y) { return x == y ? 2.0 : 1.0; };
      auto H = [&](int k, int l)
      {
        double h = L.smu * L.g[k][b] * L.g[l][a] + L.sla * L.g[k][a] * L.g[l][b];
        if (a == b)
        {
          double dot = 0.0;
#pragma unroll
          for (int x = 0; x < TDIM; ++x)
            dot += L.g[k][x] * L.g[l][x];
          h += L.smu * dot;
        }
        return h;
      };
      if (i < NV && j < NV) // (4 l_i - 1)(4 l_j - 1)
        return (16.0 * d(i, j) * m2 - 8.0 * m1 + 1.0) * H(i, j);
      if (i < NV) // test: vertex i; trial: edge (p, q) -> 4 (l_q grad l_p + l_p grad l_q)
      {
        int p, q;
        LG::edge(j - NV, p, q);
        return 4.0 * ((4.0 * d(i, q) * m2 - m1) * H(i, p) + (4.0 * d(i, p) * m2 - m1) * H(i, q));
      }
      if (j < NV) // test: edge (p, q); trial: vertex j
      {
        int p, q;
        LG::edge(i - NV, p, q);
        return 4.0 * ((4.0 * d(j, q) * m2 - m1) * H(p, j) + (4.0 * d(j, p) * m2 - m1) * H(q, j));
      }
      int p, q, r, t; // edges (p, q) and (r, t)
      LG::edge(i - NV, p, q);
      LG::edge(j - NV, r, t);
      return 16.0 * m2 * (d(q, t) * H(p, r) + d(q, r) * H(p, t) + d(p, t) * H(q, r) + d(p, r) * H(q, t));
    }
    else if constexpr (LAZY_ELASTICITY)
    {
      // A[(i,a),(j,b)] = |T| (mu g_i^b g_j^a + lambda g_i^a g_j^b + delta_ab mu g_i.g_j)
      double v = L.smu * L.g[i][b] * L.g[j][a] + L.sla * L.g[i][a] * L.g[j][b];
      if (a == b)
      {
        double dot = 0.0;
#pragma unroll
        for (int d = 0; d < TDIM; ++d)
          dot += L.g[i][d] * L.g[j][d];
        v += L.smu * dot;
      }
      return v;
    }
    else if constexpr (LAZY_P1_STIFFNESS)
    {
      if (a != b)
        return 0.0; // component-diagonal on blocked spaces
      double dot = 0.0;
#pragma unroll
      for (int d = 0; d < TDIM; ++d)
        dot += L.g[i][d] * L.g[j][d];
      return L.smu * dot;
    }
    else if constexpr (LAZY_DIV)
    {
      // DIV_TEST: A[(i,a)][j] = c0 int psi_j d_a(phi_i); DIV_TRIAL: A[i][(j,b)] = c0 int psi_i d_b(phi_j)
      // (phi: P2 velocity basis, psi: P1 pressure basis).  With v = the P2 index, s = the P1 index,
      // x = the component: vertex v: g_v^x int l_s (4 l_v - 1); edge (p,q): 4 int l_s (l_p g_q^x + l_q g_p^x)
      using LG = Lagrange<TDIM, 2>;
      constexpr double m1 = 1.0 / (TDIM + 1), m2 = 1.0 / ((TDIM + 1) * (TDIM + 2));
      auto d = [](int x, int y) { return x == y ? 2.0 : 1.0; };
      const int v = LAZY_DIV_TEST ? i : j, sidx = LAZY_DIV_TEST ? j : i, x = LAZY_DIV_TEST ? a : b;
      if (v < NV)
        return L.smu * (4.0 * d(sidx, v) * m2 - m1) * L.g[v][x];
      int p, q;
      LG::edge(v - NV, p, q);
      return L.smu * 4.0 * m2 * (d(sidx, p) * L.g[q][x] + d(sidx, q) * L.g[p][x]);
    }
    else
    {
      if (a != b)
        return 0.0; // component-diagonal
      using LG = Lagrange<TDIM, 2>;
      // int l_a = m1 |T|, int l_a l_b = (1 + d_ab) m2 |T| on a TDIM-simplex
      constexpr double m1 = 1.0 / (TDIM + 1), m2 = 1.0 / ((TDIM + 1) * (TDIM + 2));
      auto d = [](int x, int y) { return x == y ? 2.0 : 1.0; }; // 1 + delta
      if (i < NV && j < NV) // vertex-vertex: int (4 l_i - 1)(4 l_j - 1) G_ij
        return (16.0 * d(i, j) * m2 - 8.0 * m1 + 1.0) * L.g[i][j];
      if (i < NV || j < NV) // vertex v, edge (p, q): int (4 l_v - 1) 4 (l_p G_vq + l_q G_vp)
      {
        const int v = i < NV ? i : j, e = (i < NV ? j : i) - NV;
        int p, q;
        LG::edge(e, p, q);
        return 4.0 * ((4.0 * d(v, p) * m2 - m1) * L.g[v][q] + (4.0 * d(v, q) * m2 - m1) * L.g[v][p]);
      }
      int p, q, r, t; // edge (p, q), edge (r, t): int 16 (l_p grad l_q + l_q grad l_p).(l_r grad l_t + l_t grad l_r)
      LG::edge(i - NV, p, q);
      LG::edge(j - NV, r, t);
      return 16.0 * m2 * (d(p, r) * L.g[q][t] + d(p, t) * L.g[q][r] + d(q, r) * L.g[p][t] + d(q, t) * L.g[p][r]);
    }
  }

  // entry (p, q) of the element matrix, p = i*BS0 + a, q = j*BS1 + b
  __device__ static inline double get(const double (&A)[SIZE], int p, int q)
  {
    if constexpr (DIAG)
      return (p % BS0) == (q % BS1) ? A[(p / BS0) * ND1 + q / BS1] : 0.0;
    else
      return A[p * N1 + q];
  }

  __device__ static inline void tabulate(double (&A)[SIZE], const double* w, const double* c,
                                         const double (&cd)[NV * 3], int lf, const mpcx_kernel_t& k,
                                         int comp0 = 0) // comp0: first component (source forms evaluated one
                                                        // component at a time by the scalar operator)
  {
#pragma unroll
    for (int i = 0; i < SIZE; ++i)
      A[i] = 0.0;

    // P1 simplex Laplacian: constant gradients, one point is exact
    if constexpr (FORM == MPCX_FORM_STIFFNESS && DEG0_ == 1)
    {
      if (k.coeff_degree == 0)
      {
        // A_ij = |T| grad(l_i).grad(l_j) with grad(l_{d+1}) = cof_d / det:
        // A_ij = cof_i.cof_j / (d! |det|) -- one division, symmetric
        double G[NV][TDIM], det;
        cofactor_gradients<TDIM>(cd, G, det);
        const double s = (c ? c[0] : 1.0) / ((TDIM == 3 ? 6.0 : 2.0) * fabs(det));
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
          for (int j = i; j < NV; ++j)
          {
            double dot = 0.0;
#pragma unroll
            for (int a = 0; a < TDIM; ++a)
              dot += G[i][a] * G[j][a];
            A[i * NV + j] = A[j * NV + i] = s * dot;
          }
        return;
      }
    }

    double K[TDIM][TDIM], detJ;
    affine_geometry<TDIM>(cd, K, detJ);
    const double adet = fabs(detJ);
    const double c0 = (c && FORM != MPCX_FORM_ELASTICITY) ? c[0] : 1.0;
    // P1 source term over cells without a coefficient: affine map x = x0 + J X (TDIM fma per
    // coordinate) and moment sums S = sum f w, S_d = sum f w X_d, from which the four basis
    // integrals follow (l_0 = 1 - sum X_d): ~7 fp64 instructions less per quadrature point
    // than the generic loop below
    if constexpr (FORM == MPCX_FORM_SOURCE && (DEG0_ == 1 || DEG0_ == 2))
    {
      if (k.coeff_degree == 0 && (DEG0_ == 1 || k.qphi != nullptr))
      {
        double J[3][TDIM];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int d = 0; d < TDIM; ++d)
            J[r][d] = cd[3 * (d + 1) + r] - cd[r];
        const double sd = c0 * adet;
        if constexpr (FN_ != 1)
        {
          if (k.vphi != nullptr)
          {
            // f affine in x (include/mpcx.h mpcx_kernel_t::vphi): f at the vertices, then the rule's vertex moments --
            // the same sum as the loop below up to rounding, NV * ND0 fma per component instead of nq * (ND0 + ~20)
            double F[BS0][NV];
#pragma unroll
            for (int v = 0; v < NV; ++v)
            {
              const double xv[3] = {cd[3 * v], cd[3 * v + 1], cd[3 * v + 2]};
#pragma unroll
              for (int b = 0; b < BS0; ++b)
                F[b][v] = sd * eval_fn(FN_ >= 0 ? FN_ : k.fn_id, xv, b + comp0, c);
            }
#pragma unroll
            for (int i = 0; i < ND0; ++i)
#pragma unroll
              for (int b = 0; b < BS0; ++b)
              {
                double s = A[i * BS0 + b];
#pragma unroll
                for (int v = 0; v < NV; ++v)
                  s = fma(k.vphi[v * ND0 + i], F[b][v], s);
                A[i * BS0 + b] = s;
              }
            return;
          }
        }
        // origin of the affine map (shifted to the Gaussian's centre for the benchmark function, see below)
        double org[3] = {cd[0], cd[1], cd[2]};
        if constexpr (FN_ == 1 && TDIM == 3)
        {
          org[0] -= 0.9;
          org[1] -= 0.5;
          org[2] -= 0.1;
        }
        [[maybe_unused]] FmConsts FK;
        if constexpr (FN_ == 1 && TDIM == 3)
          FK = g_fm_consts; // uniform loads: the polynomial coefficients live in scalar registers
        double S[BS0], SX[BS0][TDIM];
#pragma unroll
        for (int b = 0; b < BS0; ++b)
        {
          S[b] = 0.0;
#pragma unroll
          for (int d = 0; d < TDIM; ++d)
            SX[b][d] = 0.0;
        }
        // two points per trip: the scalar loads of the rule and the LDS table read of fast_exp
        // of one point overlap with the arithmetic of the other
#pragma unroll 2
        for (int q = 0; q < k.nq; ++q)
        {
          double X[TDIM], x[3];
#pragma unroll
          for (int d = 0; d < TDIM; ++d)
            X[d] = k.qpts[q * TDIM + d];
#pragma unroll
          for (int r = 0; r < 3; ++r)
          {
            double v = org[r];
#pragma unroll
            for (int d = 0; d < TDIM; ++d)
              v = fma(J[r][d], X[d], v);
            x[r] = v;
          }
          const double wq = k.qwts[q] * sd;
#pragma unroll
          for (int b = 0; b < BS0; ++b)
          {
            double f;
            if constexpr (FN_ == 1 && TDIM == 3)
            {
              // the benchmark's right-hand side (eval_fn case 1) on coordinates relative to the centre
              // of its Gaussian: x - 0.9, y - 0.5, z - 0.1 come straight out of the affine map (x was
              // formed from the shifted origin above), 5 y = 5 (y - 0.5) + 2.5 is one fma
              const double t = fma(5.0, x[1], 2.5);
              f = wq * ((x[0] + 0.9) * fast_sinpi_k(t, FK)
                        + fast_exp_nonpos_k(-(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]) * (1.0 / 0.02), FK));
            }
            else
              f = wq * eval_fn(FN_ >= 0 ? FN_ : k.fn_id, x, b + comp0, c);
            if constexpr (DEG0_ == 1)
            {
              S[b] += f;
#pragma unroll
              for (int d = 0; d < TDIM; ++d)
                SX[b][d] = fma(f, X[d], SX[b][d]);
            }
            else
            {
              // P2: the basis values at the point are kernel data (kernel.qphi: wave-uniform scalar loads, so
              // every product below is one fma with an SGPR operand)
#pragma unroll
              for (int i = 0; i < ND0; ++i)
                A[i * BS0 + b] = fma(f, k.qphi[q * ND0 + i], A[i * BS0 + b]);
            }
          }
        }
        if constexpr (DEG0_ == 1)
        {
#pragma unroll
          for (int b = 0; b < BS0; ++b)
          {
            double s0 = S[b];
#pragma unroll
            for (int d = 0; d < TDIM; ++d)
            {
              s0 -= SX[b][d];
              A[(d + 1) * BS0 + b] = SX[b][d];
            }
            A[b] = s0;
          }
        }
        return;
      }
    }

    QuadPoint<TDIM, FACET> qp;
    qp.init_facet(cd, lf);
    const int nq = FACET ? k.nqf : k.nq;
    for (int q = 0; q < nq; ++q)
    {
      double X[3];
      const double wq = qp.point(k, q, adet, X);
      double phi[ND0], dphi[ND0][TDIM];
      L0::eval(X, phi, dphi);
      double s = wq * c0;
      if (k.coeff_degree > 0 && FORM != MPCX_FORM_ELASTICITY)
        s *= eval_coefficient<TDIM>(k.coeff_degree, w, X);

      if constexpr (FORM == MPCX_FORM_MASS || FORM == MPCX_FORM_FACET_MASS)
      {
        // scalar matrix S (expanded per component by get() when BS0 > 1)
#pragma unroll
        for (int i = 0; i < ND0; ++i)
#pragma unroll
          for (int j = 0; j < ND1; ++j)
            A[i * ND1 + j] += s * phi[i] * phi[j];
      }
      else if constexpr (RANK1)
      {
        double x[3];
        push_forward<TDIM>(cd, X, x);
#pragma unroll
        for (int b = 0; b < BS0; ++b)
        {
          const double f = s * eval_fn(FN_ >= 0 ? FN_ : k.fn_id, x, b + comp0, c);
#pragma unroll
          for (int i = 0; i < ND0; ++i)
            A[i * BS0 + b] += f * phi[i];
        }
      }
      else
      {
        // physical gradients of the test basis
        double g[ND0][TDIM];
#pragma unroll
        for (int i = 0; i < ND0; ++i)
#pragma unroll
          for (int a = 0; a < TDIM; ++a)
          {
            double v = 0.0;
#pragma unroll
            for (int d = 0; d < TDIM; ++d)
              v += K[d][a] * dphi[i][d];
            g[i][a] = v;
          }
        if constexpr (FORM == MPCX_FORM_STIFFNESS)
        {
#pragma unroll
          for (int i = 0; i < ND0; ++i)
#pragma unroll
            for (int j = 0; j < ND1; ++j)
            {
              double dot = 0.0;
#pragma unroll
              for (int a = 0; a < TDIM; ++a)
                dot += g[i][a] * g[j][a];
              A[i * ND1 + j] += s * dot;
            }
        }
        else if constexpr (FORM == MPCX_FORM_ELASTICITY)
        {
          const double mu = c[0], lmbda = c[1];
#pragma unroll
          for (int i = 0; i < ND0; ++i)
#pragma unroll
            for (int j = 0; j < ND1; ++j)
            {
              double dot = 0.0;
#pragma unroll
              for (int a = 0; a < TDIM; ++a)
                dot += g[i][a] * g[j][a];
#pragma unroll
              for (int a = 0; a < BS0; ++a)
#pragma unroll
                for (int b = 0; b < BS1; ++b)
                {
                  double v = mu * g[i][b] * g[j][a] + lmbda * g[i][a] * g[j][b];
                  if (a == b)
                    v += mu * dot;
                  A[(i * BS0 + a) * N1 + (j * BS1 + b)] += wq * v;
                }
            }
        }
        else if constexpr (FORM == MPCX_FORM_DIV_TEST)
        {
          // A[(i,a)][j] = c0 * int psi_j d_a(phi_i): test = vector space, trial = scalar
          double psi[ND1], dpsi[ND1][TDIM];
          L1::eval(X, psi, dpsi);
#pragma unroll
          for (int i = 0; i < ND0; ++i)
#pragma unroll
            for (int a = 0; a < BS0; ++a)
#pragma unroll
              for (int j = 0; j < ND1; ++j)
                A[(i * BS0 + a) * N1 + j] += s * g[i][a] * psi[j];
        }
      }
      if constexpr (FORM == MPCX_FORM_DIV_TRIAL)
      {
        // A[i][(j,b)] = c0 * int psi_i d_b(phi_j): test = scalar space, trial = vector
        double vphi[ND1], dvphi[ND1][TDIM];
        L1::eval(X, vphi, dvphi);
#pragma unroll
        for (int j = 0; j < ND1; ++j)
#pragma unroll
          for (int b = 0; b < BS1; ++b)
          {
            double gjb = 0.0;
#pragma unroll
            for (int d = 0; d < TDIM; ++d)
              gjb += K[d][b] * dvphi[j][d];
#pragma unroll
            for (int i = 0; i < ND0; ++i)
              A[i * N1 + j * BS1 + b] += s * phi[i] * gjb;
          }
      }
    }
  }
};

} // namespace mpcx
