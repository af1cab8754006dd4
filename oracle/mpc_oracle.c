/*
 * oracle/mpc_oracle.c -- TEST INFRASTRUCTURE ONLY (see mpc_oracle.h).
 *
 * Plain-C restatement of the reference's serial constrained assembly loops:
 *   cpp/assemble_matrix.cpp:33-77     fill_stripped_matrix
 *   cpp/assemble_matrix.cpp:99-268    modify_mpc_cell
 *   cpp/assemble_matrix.cpp:271-415   assemble_exterior_facets
 *   cpp/assemble_matrix.cpp:417-548   assemble_cells_impl
 *   cpp/assemble_matrix.cpp:662-726   _assemble_matrix (slave diagonal)
 *   cpp/assemble_utils.cpp:10-28      compute_local_slave_index
 *   cpp/assemble_vector.h:35-69       modify_mpc_vec
 *   cpp/assemble_vector.cpp:34-91     _assemble_entities_impl
 *   cpp/lifting.h:45-134, 243-397     lift_bc_entities + lifting lambdas
 *   cpp/MultiPointConstraint.h:129-152 backsubstitution / homogenize
 * with PETSc MatSetValues[Blocked]Local(ADD_VALUES) replaced by a per-row
 * binary search into a sorted CSR (python/src/dolfinx_mpc/mpc.cpp:284-287).
 *
 * The element kernels stand in for FFCx-generated tabulate_tensor functions
 * (third party, absent): PARITY UNPINNED for absolute element-tensor values;
 * they are written FFCx-style (quadrature loops over tabulated bases) and
 * pinned by closed-form answers in tests/test_oracle_kernels.py.  The MPC
 * algebra is pinned by the reference's own identities (A_mpc = K^T A K,
 * b_mpc = K^T b: tests/test_oracle_identities.py) and by the one absolute
 * known answer its tests hold, the Poiseuille flow of
 * python/tests/test_stokes_channelflow.py (tests/test_stokes_poiseuille.py).
 */
#include "mpc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ */
/* Reference-element tables (Basix/UFC conventions)                     */
/* ------------------------------------------------------------------ */

/* tetrahedron edges: local vertex pairs */
static const int TET_EDGES[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
/* triangle edges */
static const int TRI_EDGES[3][2] = {{1, 2}, {0, 2}, {0, 1}};
/* facets (opposite vertex i) */
static const int TET_FACETS[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};
static const int TRI_FACETS[3][2] = {{1, 2}, {0, 2}, {0, 1}};
/* reference vertices */
static const double TET_VERTS[4][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
static const double TRI_VERTS[3][2] = {{0, 0}, {1, 0}, {0, 1}};

static int cell_tdim(int celltype) { return celltype == ORACLE_CELL_TETRAHEDRON ? 3 : 2; }

static int lagrange_ndofs(int celltype, int degree)
{
  if (celltype == ORACLE_CELL_TETRAHEDRON)
    return degree == 1 ? 4 : 10;
  return degree == 1 ? 3 : 6;
}

/* Lagrange basis + reference gradients at reference point X.
 * phi[nd], dphi[nd][tdim]. */
static void lagrange_basis(int celltype, int degree, const double* X, double* phi,
                           double* dphi)
{
  const int tdim = cell_tdim(celltype);
  const int nv = tdim + 1;
  double lam[4];
  double dlam[4][3];
  memset(dlam, 0, sizeof(dlam));
  lam[0] = 1.0;
  for (int d = 0; d < tdim; ++d)
  {
    lam[0] -= X[d];
    lam[d + 1] = X[d];
    dlam[0][d] = -1.0;
    dlam[d + 1][d] = 1.0;
  }
  if (degree == 1)
  {
    for (int i = 0; i < nv; ++i)
    {
      phi[i] = lam[i];
      for (int d = 0; d < tdim; ++d)
        dphi[i * tdim + d] = dlam[i][d];
    }
    return;
  }
  /* degree 2 */
  for (int i = 0; i < nv; ++i)
  {
    phi[i] = lam[i] * (2.0 * lam[i] - 1.0);
    for (int d = 0; d < tdim; ++d)
      dphi[i * tdim + d] = (4.0 * lam[i] - 1.0) * dlam[i][d];
  }
  const int ne = tdim == 3 ? 6 : 3;
  for (int e = 0; e < ne; ++e)
  {
    const int a = tdim == 3 ? TET_EDGES[e][0] : TRI_EDGES[e][0];
    const int b = tdim == 3 ? TET_EDGES[e][1] : TRI_EDGES[e][1];
    phi[nv + e] = 4.0 * lam[a] * lam[b];
    for (int d = 0; d < tdim; ++d)
      dphi[(nv + e) * tdim + d] = 4.0 * (lam[a] * dlam[b][d] + lam[b] * dlam[a][d]);
  }
}

/* ------------------------------------------------------------------ */
/* Analytic functions                                                   */
/* ------------------------------------------------------------------ */
double oracle_eval_fn(int fn_id, const double* x, int comp, const double* c)
{
  switch (fn_id)
  {
  case 0:
    return 1.0;
  case 1:
  {
    /* python/benchmarks/bench_periodic.py:85-89 */
    const double dx = x[0] - 0.9, dy = x[1] - 0.5, dz = x[2] - 0.1;
    return x[0] * sin(5.0 * M_PI * x[1]) + 1.0 * exp(-(dx * dx + dy * dy + dz * dz) / 0.02);
  }
  case 2:
    /* python/tests/test_vector_assembly.py:39 style rhs: sin(2 pi x) sin(pi y)... */
    return sin(2.0 * M_PI * x[0]) * sin(M_PI * x[1]) + 0.3 * (comp + 1);
  case 3:
    /* polynomial of total degree 3 (exactness checks) */
    return 1.0 + 2.0 * x[0] + 3.0 * x[1] * x[1] - x[2] * x[2] * x[2] + x[0] * x[1] * x[2]
           + 0.5 * comp * x[0];
  case 4:
    /* linear, component dependent */
    return (comp + 1) * (1.0 + x[0] - 2.0 * x[1] + 0.5 * x[2]);
  case 5:
    /* constant vector stored after the scale: c = [scale, g_0, g_1, ...]
     * (python/tests/test_surface_integral.py:54 traction constant) */
    return c[1 + comp];
  default:
    return 0.0;
  }
}

/* ------------------------------------------------------------------ */
/* Affine geometry                                                      */
/* ------------------------------------------------------------------ */
typedef struct
{
  double J[3][3];
  double K[3][3];
  double detJ;
} affine_map;

/* coordinate_dofs is [nv][3] (always 3 components, assemble_matrix.cpp:473,499) */
static void affine_geometry(int tdim, const double* cd, affine_map* g)
{
  memset(g, 0, sizeof(*g));
  for (int i = 0; i < tdim; ++i)
    for (int j = 0; j < tdim; ++j)
      g->J[i][j] = cd[3 * (j + 1) + i] - cd[i];
  if (tdim == 2)
  {
    const double det = g->J[0][0] * g->J[1][1] - g->J[0][1] * g->J[1][0];
    g->detJ = det;
    g->K[0][0] = g->J[1][1] / det;
    g->K[0][1] = -g->J[0][1] / det;
    g->K[1][0] = -g->J[1][0] / det;
    g->K[1][1] = g->J[0][0] / det;
  }
  else
  {
    const double(*J)[3] = g->J;
    const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
    const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
    const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
    g->detJ = det;
    g->K[0][0] = c00 / det;
    g->K[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
    g->K[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
    g->K[1][0] = c01 / det;
    g->K[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
    g->K[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
    g->K[2][0] = c02 / det;
    g->K[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
    g->K[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
  }
}

/* physical point from reference point (P1 geometry) */
static void push_forward(int tdim, const double* cd, const double* X, double* x)
{
  double l0 = 1.0;
  for (int d = 0; d < tdim; ++d)
    l0 -= X[d];
  for (int i = 0; i < 3; ++i)
  {
    double v = l0 * cd[i];
    for (int d = 0; d < tdim; ++d)
      v += X[d] * cd[3 * (d + 1) + i];
    x[i] = v;
  }
}

#define MAX_ND 10
#define MAX_BS 3

/* ------------------------------------------------------------------ */
/* Generic FFCx-style kernel                                            */
/* ------------------------------------------------------------------ */
void oracle_tabulate_generic(double* A, const double* w, const double* c,
                             const double* coordinate_dofs,
                             const int* entity_local_index,
                             const uint8_t* quadrature_permutation,
                             void* custom_data)
{
  (void)quadrature_permutation;
  const oracle_kernel_desc* kd = (const oracle_kernel_desc*)custom_data;
  const int tdim = cell_tdim(kd->celltype);
  const int nd = lagrange_ndofs(kd->celltype, kd->degree);
  const int bs = kd->bs;
  const int n = nd * bs;
  const double c0 = c ? c[0] : 1.0;
  affine_map g;
  affine_geometry(tdim, coordinate_dofs, &g);
  const double adet = fabs(g.detJ);
  double phi[MAX_ND], dphi[MAX_ND * 3], gphi[MAX_ND][3], cphi[MAX_ND], cdphi[MAX_ND * 3];

  const int is_facet = (kd->form == ORACLE_FORM_FACET_MASS || kd->form == ORACLE_FORM_FACET_SOURCE);
  const int nq = is_facet ? kd->nqf : kd->nq;

  /* facet geometry */
  double fscale = 0.0;
  double fv[3][3]; /* reference coordinates of the facet vertices */
  if (is_facet)
  {
    const int lf = *entity_local_index;
    double pv[3][3];
    for (int a = 0; a < tdim; ++a)
    {
      const int v = tdim == 3 ? TET_FACETS[lf][a] : TRI_FACETS[lf][a];
      for (int d = 0; d < tdim; ++d)
        fv[a][d] = tdim == 3 ? TET_VERTS[v][d] : TRI_VERTS[v][d];
      for (int i = 0; i < 3; ++i)
        pv[a][i] = coordinate_dofs[3 * v + i];
    }
    if (tdim == 3)
    {
      double e1[3], e2[3];
      for (int i = 0; i < 3; ++i)
      {
        e1[i] = pv[1][i] - pv[0][i];
        e2[i] = pv[2][i] - pv[0][i];
      }
      const double cx = e1[1] * e2[2] - e1[2] * e2[1];
      const double cy = e1[2] * e2[0] - e1[0] * e2[2];
      const double cz = e1[0] * e2[1] - e1[1] * e2[0];
      fscale = sqrt(cx * cx + cy * cy + cz * cz); /* rule weights sum to 1/2 */
    }
    else
    {
      const double ex = pv[1][0] - pv[0][0], ey = pv[1][1] - pv[0][1], ez = pv[1][2] - pv[0][2];
      fscale = sqrt(ex * ex + ey * ey + ez * ez); /* rule weights sum to 1 */
    }
  }

  for (int q = 0; q < nq; ++q)
  {
    double X[3] = {0, 0, 0};
    double wq;
    if (is_facet)
    {
      const double* s = kd->fqpts + (size_t)q * (tdim - 1);
      double l0 = 1.0;
      for (int d = 0; d < tdim - 1; ++d)
        l0 -= s[d];
      for (int d = 0; d < tdim; ++d)
      {
        double v = l0 * fv[0][d];
        for (int a = 1; a < tdim; ++a)
          v += s[a - 1] * fv[a][d];
        X[d] = v;
      }
      wq = kd->fqwts[q] * fscale;
    }
    else
    {
      for (int d = 0; d < tdim; ++d)
        X[d] = kd->qpts[(size_t)q * tdim + d];
      wq = kd->qwts[q] * adet;
    }
    lagrange_basis(kd->celltype, kd->degree, X, phi, dphi);
    /* physical gradients: gphi[i][a] = sum_d K[d][a] dphi[i][d] */
    for (int i = 0; i < nd; ++i)
      for (int a = 0; a < tdim; ++a)
      {
        double v = 0.0;
        for (int d = 0; d < tdim; ++d)
          v += g.K[d][a] * dphi[i * tdim + d];
        gphi[i][a] = v;
      }
    double wc = c0;
    if (kd->coeff_degree > 0)
    {
      const int ncd = lagrange_ndofs(kd->celltype, kd->coeff_degree);
      lagrange_basis(kd->celltype, kd->coeff_degree, X, cphi, cdphi);
      double v = 0.0;
      for (int k = 0; k < ncd; ++k)
        v += w[k] * cphi[k];
      wc *= v;
    }
    const double s = wq * wc;
    switch (kd->form)
    {
    case ORACLE_FORM_STIFFNESS:
      for (int i = 0; i < nd; ++i)
        for (int j = 0; j < nd; ++j)
        {
          double dot = 0.0;
          for (int a = 0; a < tdim; ++a)
            dot += gphi[i][a] * gphi[j][a];
          for (int k = 0; k < bs; ++k)
            A[(i * bs + k) * n + (j * bs + k)] += s * dot;
        }
      break;
    case ORACLE_FORM_MASS:
    case ORACLE_FORM_FACET_MASS:
      for (int i = 0; i < nd; ++i)
        for (int j = 0; j < nd; ++j)
          for (int k = 0; k < bs; ++k)
            A[(i * bs + k) * n + (j * bs + k)] += s * phi[i] * phi[j];
      break;
    case ORACLE_FORM_SOURCE:
    case ORACLE_FORM_FACET_SOURCE:
    {
      double x[3];
      push_forward(tdim, coordinate_dofs, X, x);
      for (int k = 0; k < bs; ++k)
      {
        const double f = oracle_eval_fn(kd->fn_id, x, k, c);
        for (int i = 0; i < nd; ++i)
          A[i * bs + k] += s * f * phi[i];
      }
      break;
    }
    case ORACLE_FORM_ELASTICITY:
    {
      /* c = [mu, lambda]; entry ((i,a),(j,b)) =
       *   mu (delta_ab grad(phi_i).grad(phi_j) + d_b phi_i d_a phi_j) + lambda d_a phi_i d_b phi_j */
      const double mu = c[0], lmbda = c[1];
      const double sw = wq; /* c holds [mu, lambda]; no extra scale */
      for (int i = 0; i < nd; ++i)
        for (int j = 0; j < nd; ++j)
        {
          double dot = 0.0;
          for (int a = 0; a < tdim; ++a)
            dot += gphi[i][a] * gphi[j][a];
          for (int a = 0; a < bs; ++a)
            for (int b = 0; b < bs; ++b)
            {
              double v = mu * gphi[i][b] * gphi[j][a] + lmbda * gphi[i][a] * gphi[j][b];
              if (a == b)
                v += mu * dot;
              A[(i * bs + a) * n + (j * bs + b)] += sw * v;
            }
        }
      break;
    }
    case ORACLE_FORM_DIV_TEST:
    case ORACLE_FORM_DIV_TRIAL:
    {
      /* rectangular Taylor-Hood blocks (python/tests/test_stokes_channelflow.py:77-80):
       * DIV_TEST  a(p, v) = c0 p div(v): rows (i,a) of the vector space, cols j of the scalar space
       * DIV_TRIAL a(u, q) = c0 div(u) q: rows i of the scalar space, cols (j,b) of the vector space */
      const int nd1 = lagrange_ndofs(kd->celltype, kd->degree1);
      const int bs1 = kd->bs1;
      const int n1 = nd1 * bs1;
      double psi[MAX_ND], dpsi[MAX_ND * 3];
      lagrange_basis(kd->celltype, kd->degree1, X, psi, dpsi);
      if (kd->form == ORACLE_FORM_DIV_TEST)
      {
        for (int i = 0; i < nd; ++i)
          for (int a = 0; a < bs; ++a)
            for (int j = 0; j < nd1; ++j)
              A[(i * bs + a) * n1 + j] += s * gphi[i][a] * psi[j];
      }
      else
      {
        for (int j = 0; j < nd1; ++j)
          for (int b = 0; b < bs1; ++b)
          {
            double gjb = 0.0;
            for (int d = 0; d < tdim; ++d)
              gjb += g.K[d][b] * dpsi[j * tdim + d];
            for (int i = 0; i < nd; ++i)
              A[i * n1 + j * bs1 + b] += s * phi[i] * gjb;
          }
      }
      break;
    }
    default:
      break;
    }
  }
}

/* ------------------------------------------------------------------ */
/* FFCx-like specialised kernels (the CPU baseline uses these)          */
/* ------------------------------------------------------------------ */
void oracle_tabulate_laplace_p1_tet(double* A, const double* w, const double* c,
                                    const double* cd, const int* entity_local_index,
                                    const uint8_t* quadrature_permutation,
                                    void* custom_data)
{
  (void)w;
  (void)c;
  (void)entity_local_index;
  (void)quadrature_permutation;
  (void)custom_data;
  const double J00 = cd[3] - cd[0], J01 = cd[6] - cd[0], J02 = cd[9] - cd[0];
  const double J10 = cd[4] - cd[1], J11 = cd[7] - cd[1], J12 = cd[10] - cd[1];
  const double J20 = cd[5] - cd[2], J21 = cd[8] - cd[2], J22 = cd[11] - cd[2];
  const double c00 = J11 * J22 - J12 * J21;
  const double c01 = J12 * J20 - J10 * J22;
  const double c02 = J10 * J21 - J11 * J20;
  const double det = J00 * c00 + J01 * c01 + J02 * c02;
  const double id = 1.0 / det;
  /* K = J^-1 ; rows of K are the physical gradients of lambda_1..3 */
  const double K00 = c00 * id, K01 = (J02 * J21 - J01 * J22) * id, K02 = (J01 * J12 - J02 * J11) * id;
  const double K10 = c01 * id, K11 = (J00 * J22 - J02 * J20) * id, K12 = (J02 * J10 - J00 * J12) * id;
  const double K20 = c02 * id, K21 = (J01 * J20 - J00 * J21) * id, K22 = (J00 * J11 - J01 * J10) * id;
  double G[4][3];
  G[1][0] = K00; G[1][1] = K01; G[1][2] = K02;
  G[2][0] = K10; G[2][1] = K11; G[2][2] = K12;
  G[3][0] = K20; G[3][1] = K21; G[3][2] = K22;
  for (int a = 0; a < 3; ++a)
    G[0][a] = -(G[1][a] + G[2][a] + G[3][a]);
  const double vol = fabs(det) / 6.0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      A[4 * i + j] += vol * (G[i][0] * G[j][0] + G[i][1] * G[j][1] + G[i][2] * G[j][2]);
}

void oracle_tabulate_source_p1_tet(double* A, const double* w, const double* c,
                                   const double* cd, const int* entity_local_index,
                                   const uint8_t* quadrature_permutation,
                                   void* custom_data)
{
  (void)w;
  (void)c;
  (void)entity_local_index;
  (void)quadrature_permutation;
  const oracle_kernel_desc* kd = (const oracle_kernel_desc*)custom_data;
  const double J00 = cd[3] - cd[0], J01 = cd[6] - cd[0], J02 = cd[9] - cd[0];
  const double J10 = cd[4] - cd[1], J11 = cd[7] - cd[1], J12 = cd[10] - cd[1];
  const double J20 = cd[5] - cd[2], J21 = cd[8] - cd[2], J22 = cd[11] - cd[2];
  const double det = J00 * (J11 * J22 - J12 * J21) + J01 * (J12 * J20 - J10 * J22)
                     + J02 * (J10 * J21 - J11 * J20);
  const double adet = fabs(det);
  for (int q = 0; q < kd->nq; ++q)
  {
    const double X = kd->qpts[3 * q], Y = kd->qpts[3 * q + 1], Z = kd->qpts[3 * q + 2];
    const double l0 = 1.0 - X - Y - Z;
    double x[3];
    x[0] = l0 * cd[0] + X * cd[3] + Y * cd[6] + Z * cd[9];
    x[1] = l0 * cd[1] + X * cd[4] + Y * cd[7] + Z * cd[10];
    x[2] = l0 * cd[2] + X * cd[5] + Y * cd[8] + Z * cd[11];
    const double s = kd->qwts[q] * adet * oracle_eval_fn(kd->fn_id, x, 0, c);
    A[0] += s * l0;
    A[1] += s * X;
    A[2] += s * Y;
    A[3] += s * Z;
  }
}

/* an element kernel handed in by the caller (UFCx signature, the reference's own seam:
 * cpp/assemble_matrix.cpp:438-439): which == 100 */
static oracle_tabulate_fn g_user_kernel = 0;
void oracle_set_user_kernel(oracle_tabulate_fn fn) { g_user_kernel = fn; }

/* dof transformations of an imported element, applied to the element tensor right after the kernel call
 * (cpp/assemble_matrix.cpp:432-436, 507-508: apply_dof_transformation(_Ae, cell_info0, cell0, ndim1) and
 * apply_dof_transformation_to_transpose(_Ae, cell_info1, cell1, ndim0); cpp/assemble_vector.cpp:184; cpp/lifting.h): the
 * caller hands in the two functions and the cell permutation words, NULL = none (every Lagrange element) */
typedef void (*oracle_transform_fn)(double*, const uint32_t*, int32_t, int32_t);
static oracle_transform_fn g_t0 = 0, g_t1 = 0;
static const uint32_t *g_info0 = 0, *g_info1 = 0;
void oracle_set_dof_transformations(oracle_transform_fn t0, oracle_transform_fn t1, const uint32_t* info0, const uint32_t* info1)
{
  g_t0 = t0, g_t1 = t1, g_info0 = info0, g_info1 = info1;
}

static oracle_tabulate_fn pick_kernel(int which)
{
  switch (which)
  {
  case 100:
    return g_user_kernel;
  case 1:
    return oracle_tabulate_laplace_p1_tet;
  case 2:
    return oracle_tabulate_source_p1_tet;
  default:
    return oracle_tabulate_generic;
  }
}

void oracle_tabulate_one(int which, double* A, const double* w, const double* c,
                         const double* coordinate_dofs, int local_facet,
                         const oracle_kernel_desc* desc)
{
  pick_kernel(which)(A, w, c, coordinate_dofs, &local_facet, NULL, (void*)desc);
}

/* ------------------------------------------------------------------ */
/* CSR insertion (MatSetValuesLocal ADD_VALUES stand-in)                */
/* ------------------------------------------------------------------ */
static inline void csr_add(oracle_csr* A, int32_t row, int32_t col, double v)
{
  int32_t lo = A->rowptr[row], hi = A->rowptr[row + 1];
  while (lo < hi)
  {
    const int32_t mid = lo + ((hi - lo) >> 1);
    if (A->cols[mid] < col)
      lo = mid + 1;
    else
      hi = mid;
  }
  if (lo < A->rowptr[row + 1] && A->cols[lo] == col)
    A->vals[lo] += v;
  else
    A->missing++;
}

/* MatSetValuesBlockedLocal: dolfinx Matrix::set_block_fn (mpc.cpp:285) */
static void csr_add_block(oracle_csr* A, const int32_t* dofs0, int nd0, int bs0,
                          const int32_t* dofs1, int nd1, int bs1, const double* Ae)
{
  const int n1 = nd1 * bs1;
  for (int i = 0; i < nd0; ++i)
    for (int k = 0; k < bs0; ++k)
    {
      const int32_t row = dofs0[i] * bs0 + k;
      for (int j = 0; j < nd1; ++j)
        for (int l = 0; l < bs1; ++l)
          csr_add(A, row, dofs1[j] * bs1 + l, Ae[(i * bs0 + k) * n1 + (j * bs1 + l)]);
    }
}

/* ------------------------------------------------------------------ */
/* cpp/assemble_utils.cpp:10-28                                         */
/* ------------------------------------------------------------------ */
static void compute_local_slave_index(const int32_t* slaves, int ns, int num_dofs, int bs,
                                      const int32_t* cell_dofs, const int8_t* is_slave,
                                      int32_t* local_index)
{
  for (int s = 0; s < ns; ++s)
    local_index[s] = 0;
  for (int i = 0; i < num_dofs; ++i)
    for (int j = 0; j < bs; ++j)
    {
      const int32_t dof = cell_dofs[i] * bs + j;
      if (is_slave[dof])
      {
        int s = 0;
        while (s < ns && slaves[s] != dof)
          ++s;
        if (s < ns) /* the reference would write out of bounds otherwise */
          local_index[s] = i * bs + j;
      }
    }
}

#define MAX_CELL_SLAVES 64
#define MAX_FLAT 512

/* ------------------------------------------------------------------ */
/* cpp/assemble_matrix.cpp:99-268 (fill_stripped_matrix :33-77 inlined) */
/* ------------------------------------------------------------------ */
static int modify_mpc_cell(oracle_csr* A, const int num_dofs[2], double* Ae,
                           const int32_t* dofs[2], const int bs[2],
                           const int32_t* slaves[2], const int nslaves[2],
                           const oracle_mpc* mpc[2], double* scratch)
{
  int32_t local_index[2][MAX_CELL_SLAVES];
  int num_flat[2] = {0, 0};
  for (int axis = 0; axis < 2; ++axis)
  {
    if (nslaves[axis] > MAX_CELL_SLAVES)
      return -2;
    compute_local_slave_index(slaves[axis], nslaves[axis], num_dofs[axis], bs[axis],
                              dofs[axis], mpc[axis]->is_slave, local_index[axis]);
    for (int i = 0; i < num_dofs[axis]; ++i)
      for (int j = 0; j < bs[axis]; ++j)
      {
        const int32_t dof = dofs[axis][i] * bs[axis] + j;
        if (mpc[axis]->is_slave[dof])
          num_flat[axis] += mpc[axis]->masters_offsets[dof + 1] - mpc[axis]->masters_offsets[dof];
      }
  }
  const int ndim0 = bs[0] * num_dofs[0];
  const int ndim1 = bs[1] * num_dofs[1];
  double* Ae_original = scratch;
  double* Ae_stripped = scratch + ndim0 * ndim1;
  double* Arow = scratch + 2 * ndim0 * ndim1;
  double* Acol = Arow + ndim0;
  memcpy(Ae_original, Ae, sizeof(double) * ndim0 * ndim1);

  /* fill_stripped_matrix, :33-77 */
  for (int i = 0; i < num_dofs[0]; ++i)
    for (int r = 0; r < bs[0]; ++r)
    {
      const int slave_row = mpc[0]->is_slave[dofs[0][i] * bs[0] + r];
      const int l_row = i * bs[0] + r;
      for (int j = 0; j < num_dofs[1]; ++j)
        for (int cc = 0; cc < bs[1]; ++cc)
        {
          const int slave_col = mpc[1]->is_slave[dofs[1][j] * bs[1] + cc];
          const int l_col = j * bs[1] + cc;
          Ae_stripped[l_row * ndim1 + l_col]
              = (slave_row && slave_col) ? 0.0 : Ae[l_row * ndim1 + l_col];
        }
    }

  /* zero slave rows / cols of Ae, :165-178 */
  for (int s = 0; s < nslaves[0]; ++s)
    for (int j = 0; j < ndim1; ++j)
      Ae[local_index[0][s] * ndim1 + j] = 0.0;
  for (int s = 0; s < nslaves[1]; ++s)
    for (int r = 0; r < ndim0; ++r)
      Ae[r * ndim1 + local_index[1][s]] = 0.0;

  /* flatten, :182-201 */
  int32_t fm[2][MAX_FLAT], fs[2][MAX_FLAT];
  double fc[2][MAX_FLAT];
  for (int axis = 0; axis < 2; ++axis)
  {
    int cnt = 0;
    for (int i = 0; i < nslaves[axis]; ++i)
    {
      const int32_t s = slaves[axis][i];
      for (int32_t p = mpc[axis]->masters_offsets[s]; p < mpc[axis]->masters_offsets[s + 1]; ++p)
      {
        if (cnt >= MAX_FLAT)
          return -3;
        fs[axis][cnt] = local_index[axis][i];
        fm[axis][cnt] = mpc[axis]->masters[p];
        fc[axis][cnt] = mpc[axis]->coeffs[p];
        ++cnt;
      }
    }
    num_flat[axis] = cnt;
  }

  /* row masters, :214-246 */
  for (int i = 0; i < num_flat[0]; ++i)
  {
    const double coeff_i = fc[0][i]; /* real T: plain transpose */
    const int32_t row = fm[0][i];
    for (int j = 0; j < num_dofs[1]; ++j)
      for (int k = 0; k < bs[1]; ++k)
      {
        Acol[j * bs[1] + k] = coeff_i * Ae_stripped[fs[0][i] * ndim1 + j * bs[1] + k];
        csr_add(A, row, dofs[1][j] * bs[1] + k, Acol[j * bs[1] + k]);
      }
    for (int j = 0; j < num_flat[1]; ++j)
    {
      const double v = coeff_i * fc[1][j] * Ae_original[fs[0][i] * ndim1 + fs[1][j]];
      csr_add(A, row, fm[1][j], v);
    }
  }
  /* column masters, :251-267 */
  for (int i = 0; i < num_flat[1]; ++i)
  {
    const int32_t col = fm[1][i];
    for (int j = 0; j < num_dofs[0]; ++j)
      for (int k = 0; k < bs[0]; ++k)
      {
        Arow[j * bs[0] + k] = fc[1][i] * Ae_stripped[(j * bs[0] + k) * ndim1 + fs[1][i]];
        csr_add(A, dofs[0][j] * bs[0] + k, col, Arow[j * bs[0] + k]);
      }
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* cpp/assemble_matrix.cpp:417-548 (cells, estride 1) and :271-415      */
/* (exterior facets, estride 2)                                         */
/* ------------------------------------------------------------------ */
int oracle_assemble_matrix(oracle_csr* A, int which, const oracle_kernel_desc* desc,
                           int estride, const int32_t* entities,
                           const int32_t* entities0, const int32_t* entities1,
                           int64_t n_entities, const double* x,
                           const int32_t* x_dofmap, int nv,
                           const int32_t* dofmap0, int nd0, int bs0,
                           const int32_t* dofmap1, int nd1, int bs1,
                           const int8_t* bc0, const int8_t* bc1,
                           const double* coeffs, int cstride,
                           const double* constants, const oracle_mpc* mpc0,
                           const oracle_mpc* mpc1)
{
  oracle_tabulate_fn kernel = pick_kernel(which);
  const int ndim0 = nd0 * bs0, ndim1 = nd1 * bs1;
  double* coordinate_dofs = (double*)malloc(sizeof(double) * 3 * nv);
  double* Aeb = (double*)malloc(sizeof(double) * ndim0 * ndim1);
  double* scratch = (double*)malloc(sizeof(double) * (2 * ndim0 * ndim1 + ndim0 + ndim1));
  int rc = 0;
  for (int64_t index = 0; index < n_entities; ++index)
  {
    const int64_t l = index * estride;
    const int32_t cell = entities[l];
    const int32_t cell0 = entities0[l];
    const int32_t cell1 = entities1[l];
    int local_facet = estride == 2 ? entities[l + 1] : 0;
    for (int i = 0; i < nv; ++i)
      memcpy(coordinate_dofs + 3 * i, x + 3 * (size_t)x_dofmap[(size_t)cell * nv + i],
             3 * sizeof(double));
    memset(Aeb, 0, sizeof(double) * ndim0 * ndim1);
    kernel(Aeb, coeffs ? coeffs + (size_t)index * cstride : NULL, constants, coordinate_dofs,
           estride == 2 ? &local_facet : NULL, NULL, (void*)desc);
    if (which == 100 && g_t0) /* cpp/assemble_matrix.cpp:507 */
      g_t0(Aeb, g_info0, cell0, ndim1);
    if (which == 100 && g_t1) /* :508 */
      g_t1(Aeb, g_info1, cell1, ndim0);

    const int32_t* dofs0 = dofmap0 + (size_t)cell0 * nd0;
    const int32_t* dofs1 = dofmap1 + (size_t)cell1 * nd1;
    if (bc0)
      for (int i = 0; i < nd0; ++i)
        for (int k = 0; k < bs0; ++k)
          if (bc0[bs0 * dofs0[i] + k])
            memset(Aeb + (size_t)ndim1 * (bs0 * i + k), 0, sizeof(double) * ndim1);
    if (bc1)
      for (int j = 0; j < nd1; ++j)
        for (int k = 0; k < bs1; ++k)
          if (bc1[bs1 * dofs1[j] + k])
            for (int r = 0; r < ndim0; ++r)
              Aeb[r * ndim1 + bs1 * j + k] = 0.0;

    const int ns0 = mpc0->c2s_offsets[cell0 + 1] - mpc0->c2s_offsets[cell0];
    const int ns1 = mpc1->c2s_offsets[cell1 + 1] - mpc1->c2s_offsets[cell1];
    if (ns0 > 0 || ns1 > 0)
    {
      const int num_dofs[2] = {nd0, nd1};
      const int32_t* dofs[2] = {dofs0, dofs1};
      const int bs[2] = {bs0, bs1};
      const int32_t* slaves[2] = {mpc0->c2s + mpc0->c2s_offsets[cell0], mpc1->c2s + mpc1->c2s_offsets[cell1]};
      const int nslaves[2] = {ns0, ns1};
      const oracle_mpc* mpc[2] = {mpc0, mpc1};
      rc = modify_mpc_cell(A, num_dofs, Aeb, dofs, bs, slaves, nslaves, mpc, scratch);
      if (rc)
        break;
    }
    csr_add_block(A, dofs0, nd0, bs0, dofs1, nd1, bs1, Aeb);
  }
  free(coordinate_dofs);
  free(Aeb);
  free(scratch);
  if (rc)
    return rc;
  return A->missing ? -1 : 0;
}

/* cpp/assemble_matrix.cpp:711-724 */
int oracle_add_slave_diagonal(oracle_csr* A, const oracle_mpc* mpc, double diagval)
{
  for (int32_t i = 0; i < mpc->num_local_slaves; ++i)
    csr_add(A, mpc->slaves[i], mpc->slaves[i], diagval);
  return A->missing ? -1 : 0;
}

/* dolfinx insert_diagonal called from python/src/dolfinx_mpc/assemble_matrix.py:59-62
 * (third party; ADD diagval on every owned bc dof of every bc) */
int oracle_insert_diagonal(oracle_csr* A, const int32_t* bc_dofs, int64_t n, double diagval)
{
  for (int64_t i = 0; i < n; ++i)
    csr_add(A, bc_dofs[i], bc_dofs[i], diagval);
  return A->missing ? -1 : 0;
}

/* ------------------------------------------------------------------ */
/* cpp/assemble_vector.h:35-69                                          */
/* ------------------------------------------------------------------ */
static void modify_mpc_vec(double* b, double* b_local, const double* b_local_copy,
                           const int32_t* dofs, int num_dofs, int bs,
                           const oracle_mpc* mpc, const int32_t* slaves, int ns)
{
  int32_t local_index[MAX_CELL_SLAVES];
  compute_local_slave_index(slaves, ns, num_dofs, bs, dofs, mpc->is_slave, local_index);
  for (int i = 0; i < ns; ++i)
  {
    const int32_t s = slaves[i];
    for (int32_t p = mpc->masters_offsets[s]; p < mpc->masters_offsets[s + 1]; ++p)
    {
      b[mpc->masters[p]] += mpc->coeffs[p] * b_local_copy[local_index[i]];
      b_local[local_index[i]] = 0.0; /* NB inside the master loop, as in the reference */
    }
  }
}

/* cpp/assemble_vector.cpp:34-91 with the lambdas of :152-240 */
int oracle_assemble_vector(double* b, int which, const oracle_kernel_desc* desc,
                           int estride, const int32_t* entities,
                           const int32_t* entities0, int64_t n_entities,
                           const double* x, const int32_t* x_dofmap, int nv,
                           const int32_t* dofmap, int nd, int bs,
                           const double* coeffs, int cstride,
                           const double* constants, const oracle_mpc* mpc)
{
  oracle_tabulate_fn kernel = pick_kernel(which);
  const int n = nd * bs;
  double* coordinate_dofs = (double*)malloc(sizeof(double) * 3 * nv);
  double* be = (double*)malloc(sizeof(double) * n);
  double* be_copy = (double*)malloc(sizeof(double) * n);
  for (int64_t e = 0; e < n_entities; ++e)
  {
    const int64_t l = e * estride;
    const int32_t cell = entities[l];
    const int32_t cell0 = entities0[l];
    int local_facet = estride == 2 ? entities[l + 1] : 0;
    for (int i = 0; i < nv; ++i)
      memcpy(coordinate_dofs + 3 * i, x + 3 * (size_t)x_dofmap[(size_t)cell * nv + i],
             3 * sizeof(double));
    memset(be, 0, sizeof(double) * n);
    kernel(be, coeffs ? coeffs + (size_t)e * cstride : NULL, constants, coordinate_dofs,
           estride == 2 ? &local_facet : NULL, NULL, (void*)desc);
    if (which == 100 && g_t0) /* cpp/assemble_vector.cpp:184 */
      g_t0(be, g_info0, cell0, 1);
    const int32_t* dofs = dofmap + (size_t)cell0 * nd;
    const int ns = mpc->c2s_offsets[cell0 + 1] - mpc->c2s_offsets[cell0];
    if (ns > 0)
    {
      if (ns > MAX_CELL_SLAVES)
        return -2;
      memcpy(be_copy, be, sizeof(double) * n);
      modify_mpc_vec(b, be, be_copy, dofs, nd, bs, mpc, mpc->c2s + mpc->c2s_offsets[cell0], ns);
    }
    for (int i = 0; i < nd; ++i)
      for (int k = 0; k < bs; ++k)
        b[bs * dofs[i] + k] += be[bs * i + k];
  }
  free(coordinate_dofs);
  free(be);
  free(be_copy);
  return 0;
}

/* ------------------------------------------------------------------ */
/* cpp/lifting.h:45-134 with the lambdas of :250-301 / :330-384         */
/* ------------------------------------------------------------------ */
int oracle_apply_lifting(double* b, int which, const oracle_kernel_desc* desc,
                         int estride, const int32_t* entities,
                         const int32_t* entities0, const int32_t* entities1,
                         int64_t n_entities, const double* x,
                         const int32_t* x_dofmap, int nv, const int32_t* dofmap0,
                         int nd0, int bs0, const int32_t* dofmap1, int nd1,
                         int bs1, const int8_t* bc_markers1,
                         const double* bc_values1, const double* x0, double scale,
                         const double* coeffs, int cstride,
                         const double* constants, const oracle_mpc* mpc0)
{
  oracle_tabulate_fn kernel = pick_kernel(which);
  const int num_rows = nd0 * bs0, num_cols = nd1 * bs1;
  double* coordinate_dofs = (double*)malloc(sizeof(double) * 3 * nv);
  double* Ae = (double*)malloc(sizeof(double) * num_rows * num_cols);
  double* be = (double*)malloc(sizeof(double) * num_rows);
  double* be_copy = (double*)malloc(sizeof(double) * num_rows);
  for (int64_t e = 0; e < n_entities; ++e)
  {
    const int64_t l = e * estride;
    const int32_t cell = entities[l];
    const int32_t cell0 = entities0[l];
    const int32_t cell1 = entities1[l];
    int local_facet = estride == 2 ? entities[l + 1] : 0;
    const int32_t* dmap0 = dofmap0 + (size_t)cell0 * nd0;
    const int32_t* dmap1 = dofmap1 + (size_t)cell1 * nd1;
    int has_bc = 0;
    for (int j = 0; j < nd1; ++j)
      for (int k = 0; k < bs1; ++k)
        if (bc_markers1[bs1 * dmap1[j] + k])
          has_bc = 1;
    if (!has_bc)
      continue;
    for (int i = 0; i < nv; ++i)
      memcpy(coordinate_dofs + 3 * i, x + 3 * (size_t)x_dofmap[(size_t)cell * nv + i],
             3 * sizeof(double));
    memset(Ae, 0, sizeof(double) * num_rows * num_cols);
    kernel(Ae, coeffs ? coeffs + (size_t)e * cstride : NULL, constants, coordinate_dofs,
           estride == 2 ? &local_facet : NULL, NULL, (void*)desc);
    if (which == 100 && g_t0) /* cpp/lifting.h: the same two transformations as in the matrix loop */
      g_t0(Ae, g_info0, cell0, num_cols);
    if (which == 100 && g_t1)
      g_t1(Ae, g_info1, cell1, num_rows);
    memset(be, 0, sizeof(double) * num_rows);
    for (int j = 0; j < nd1; ++j)
      for (int k = 0; k < bs1; ++k)
      {
        const int32_t jj = bs1 * dmap1[j] + k;
        if (bc_markers1[jj])
        {
          const double bc = bc_values1[jj];
          const double _x0 = x0 ? x0[jj] : 0.0;
          for (int m = 0; m < num_rows; ++m)
            be[m] -= Ae[m * num_cols + bs1 * j + k] * scale * (bc - _x0);
        }
      }
    /* slaves looked up with cell1, applied against dmap0 (lifting.h:117-127) */
    const int ns = mpc0->c2s_offsets[cell1 + 1] - mpc0->c2s_offsets[cell1];
    if (ns > 0)
    {
      if (ns > MAX_CELL_SLAVES)
        return -2;
      memcpy(be_copy, be, sizeof(double) * num_rows);
      modify_mpc_vec(b, be, be_copy, dmap0, nd0, bs0, mpc0, mpc0->c2s + mpc0->c2s_offsets[cell1], ns);
    }
    for (int i = 0; i < nd0; ++i)
      for (int k = 0; k < bs0; ++k)
        b[bs0 * dmap0[i] + k] += be[bs0 * i + k];
  }
  free(coordinate_dofs);
  free(Ae);
  free(be);
  free(be_copy);
  return 0;
}

/* cpp/MultiPointConstraint.h:129-145 */
void oracle_backsubstitution(const oracle_mpc* mpc, double* u)
{
  for (int32_t i = 0; i < mpc->num_slaves; ++i)
  {
    const int32_t s = mpc->slaves[i];
    u[s] = 0.0;
    for (int32_t p = mpc->masters_offsets[s]; p < mpc->masters_offsets[s + 1]; ++p)
      u[s] += mpc->coeffs[p] * u[mpc->masters[p]];
  }
}

/* cpp/MultiPointConstraint.h:148-152 */
void oracle_homogenize(const oracle_mpc* mpc, double* u)
{
  for (int32_t i = 0; i < mpc->num_slaves; ++i)
    u[mpc->slaves[i]] = 0.0;
}
