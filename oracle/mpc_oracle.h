/*
 * oracle/mpc_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99) of the dolfinx_mpc constrained-assembly hot
 * path.  Nothing under dolfinx_mpc_amd/ may include, link or call this; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * there only as the checker / the reported CPU baseline.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * it follows.
 *
 * PARITY STATUS
 *   - MPC algebra (modify_mpc_cell / modify_mpc_vec / lifting): pinned by the
 *     reference's own test identity  A_mpc[free,free] == K^T A K  and
 *     b_mpc[free] == K^T b  (python/src/dolfinx_mpc/utils/test.py:202-265),
 *     checked in tests/test_oracle_identities.py.
 *   - Element tensors (FFCx-generated tabulate_tensor in the reference, a
 *     third-party dependency absent from /root/reference, fenics-ffcx matching
 *     fenics-dolfinx>=0.12.0.dev0, python/pyproject.toml:23): PARITY UNPINNED.
 *     The reference cannot be compiled or imported here (needs DOLFINx, Basix,
 *     FFCx, PETSc, MPI).  Element tensors are pinned by closed-form known
 *     answers instead (tests/test_oracle_kernels.py).
 */
#ifndef MPC_ORACLE_H
#define MPC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* UFCx tabulate_tensor signature, cpp/assemble_matrix.cpp:438-439 */
typedef void (*oracle_tabulate_fn)(double* A, const double* w, const double* c,
                                   const double* coordinate_dofs,
                                   const int* entity_local_index,
                                   const uint8_t* quadrature_permutation,
                                   void* custom_data);

/* which == 100 in the entry points below: call this function (a UFCx tabulate_tensor compiled by the caller) */
void oracle_set_user_kernel(oracle_tabulate_fn fn);
void oracle_set_dof_transformations(void (*t0)(double*, const uint32_t*, int32_t, int32_t), void (*t1)(double*, const uint32_t*, int32_t, int32_t),
                                    const uint32_t* info0, const uint32_t* info1);

/* form kinds */
enum {
  ORACLE_FORM_STIFFNESS = 0, /* a = c0 * w * grad(u).grad(v) dx (per component if bs>1) */
  ORACLE_FORM_MASS = 1,      /* a = c0 * w * u.v dx */
  ORACLE_FORM_SOURCE = 2,    /* L = c0 * w * f.v dx, f analytic (fn_id) */
  ORACLE_FORM_ELASTICITY = 3,/* a = inner(sigma(u), grad(v)) dx, c = [mu, lambda] */
  ORACLE_FORM_FACET_MASS = 4,  /* a = c0 * u.v ds */
  ORACLE_FORM_FACET_SOURCE = 5, /* L = c0 * f.v ds */
  ORACLE_FORM_DIV_TEST = 6,     /* a = c0 * p div(v) dx (vector test, scalar trial) */
  ORACLE_FORM_DIV_TRIAL = 7     /* a = c0 * div(u) q dx (scalar test, vector trial) */
};

/* cell types */
enum { ORACLE_CELL_TRIANGLE = 1, ORACLE_CELL_TETRAHEDRON = 2 };

/* Descriptor handed to the generic kernels through UFCx custom_data. The
 * quadrature rule is *data* (the reference gets it baked into FFCx code). */
typedef struct
{
  int32_t form;
  int32_t celltype;
  int32_t degree;       /* Lagrange degree of test (=trial) space, 1 or 2 */
  int32_t bs;           /* block size (components) */
  int32_t degree1;      /* trial space degree / block size (rectangular forms only) */
  int32_t bs1;
  int32_t fn_id;        /* analytic function for SOURCE forms */
  int32_t coeff_degree; /* 0: no coefficient; 1/2: Lagrange coefficient packed in w */
  int32_t nq;           /* cell rule: number of points */
  int32_t nqf;          /* facet rule: number of points */
  const double* qpts;   /* [nq][tdim] */
  const double* qwts;   /* [nq] */
  const double* fqpts;  /* [nqf][tdim-1] on the reference facet */
  const double* fqwts;  /* [nqf] */
} oracle_kernel_desc;

/* Finalized MPC in the reference's own (dense-offset) layout,
 * cpp/MultiPointConstraint.h:36-126 */
typedef struct
{
  int32_t num_dofs;            /* unrolled local dofs */
  int32_t num_slaves;
  int32_t num_local_slaves;
  const int8_t* is_slave;      /* [num_dofs] */
  const int32_t* slaves;       /* [num_slaves] sorted */
  const int32_t* masters_offsets; /* [num_dofs+1] */
  const int32_t* masters;      /* local unrolled dofs */
  const double* coeffs;
  const int32_t* c2s_offsets;  /* [num_cells+1] */
  const int32_t* c2s;
} oracle_mpc;

typedef struct
{
  int32_t nrows;
  const int32_t* rowptr;
  const int32_t* cols; /* sorted within each row */
  double* vals;
  int64_t missing; /* number of insertions that found no slot (should be 0) */
} oracle_csr;

/* generic and specialised element kernels */
void oracle_tabulate_generic(double* A, const double* w, const double* c,
                             const double* coordinate_dofs,
                             const int* entity_local_index,
                             const uint8_t* quadrature_permutation,
                             void* custom_data);
void oracle_tabulate_laplace_p1_tet(double* A, const double* w, const double* c,
                                    const double* coordinate_dofs,
                                    const int* entity_local_index,
                                    const uint8_t* quadrature_permutation,
                                    void* custom_data);
void oracle_tabulate_source_p1_tet(double* A, const double* w, const double* c,
                                   const double* coordinate_dofs,
                                   const int* entity_local_index,
                                   const uint8_t* quadrature_permutation,
                                   void* custom_data);
double oracle_eval_fn(int fn_id, const double* x, int comp, const double* c);

/* call one tabulate on one cell (for the kernel known-answer tests) */
void oracle_tabulate_one(int which, double* A, const double* w, const double* c,
                         const double* coordinate_dofs, int local_facet,
                         const oracle_kernel_desc* desc);

/* which: 0 = generic, 1 = laplace_p1_tet fast path, 2 = source_p1_tet fast path */
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
                           const oracle_mpc* mpc1);

int oracle_add_slave_diagonal(oracle_csr* A, const oracle_mpc* mpc, double diagval);
int oracle_insert_diagonal(oracle_csr* A, const int32_t* bc_dofs, int64_t n,
                           double diagval);

int oracle_assemble_vector(double* b, int which, const oracle_kernel_desc* desc,
                           int estride, const int32_t* entities,
                           const int32_t* entities0, int64_t n_entities,
                           const double* x, const int32_t* x_dofmap, int nv,
                           const int32_t* dofmap, int nd, int bs,
                           const double* coeffs, int cstride,
                           const double* constants, const oracle_mpc* mpc);

int oracle_apply_lifting(double* b, int which, const oracle_kernel_desc* desc,
                         int estride, const int32_t* entities,
                         const int32_t* entities0, const int32_t* entities1,
                         int64_t n_entities, const double* x,
                         const int32_t* x_dofmap, int nv, const int32_t* dofmap0,
                         int nd0, int bs0, const int32_t* dofmap1, int nd1,
                         int bs1, const int8_t* bc_markers1,
                         const double* bc_values1, const double* x0, double scale,
                         const double* coeffs, int cstride,
                         const double* constants, const oracle_mpc* mpc0);

void oracle_backsubstitution(const oracle_mpc* mpc, double* u);
void oracle_homogenize(const oracle_mpc* mpc, double* u);

#ifdef __cplusplus
}
#endif
#endif
