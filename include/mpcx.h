/*
 * mpcx.h -- C ABI of libmpcx.so, the MI355X (gfx950) constrained-assembly
 * backend for dolfinx_mpc's hot path.
 *
 * Plain pointers and sizes only.  Pointers marked DEVICE must be HIP device
 * pointers (hipMalloc / torch CUDA tensors); pointers marked HOST are ordinary
 * host memory.  All kernels are launched on the hipStream_t passed as
 * `stream` (NULL = default stream) and are asynchronous: the caller
 * synchronises.  Every function returns 0 on success, <0 on error; the message
 * is available from mpcx_last_error().
 *
 * Each entry point cites the reference interface (relative to the
 * dolfinx_mpc repository) it replaces.  INTEGRATION.md shows the binding a
 * dolfinx_mpc maintainer would add.
 */
#ifndef MPCX_H
#define MPCX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4: mpcx_matrix_args_t::cube_flags (in the padding after cube_rec_bytes), hexahedron and closed-form cluster entry points
 * 5: pair records, scalar types;  6: mpcx_matrix_args_t::cube_rec_index (one cluster record per cluster, before ``stream``)
 * 7: cube_cells, last field before ``stream`` of the matrix and the vector argument block: MPCX_ALG_CUBE with imported (UFCx) kernels
 * 8: dof transformations of imported kernels: transform0_name / transform1_name of the descriptor, cell_info0 / cell_info1 of
 *    the three argument blocks (before ``stream``)
 * 9: mpcx_matrix_args_t::val_map / val_map_wide / out_map / out_delta (before ``stream``), mpcx_add_diagonal_mapped,
 *    mpcx_invert_permutation, mpcx_write_out_order; mpcx_vector_args_t::row_map,
 *    mpcx_lifting_args_t::row_map
 * 10: lds_floor (before ``stream``) of the matrix and the vector argument block: occupancy cap of one launch, so that a kernel
 *    of another stream finds room on every CU (co-running matrix and vector assembly); mpcx_kernel_t::vphi (last field) */
#define MPCX_VERSION 11

/* Offsets into the CSR value / column arrays (rowptr entries, positions): 64-bit, so that one GPU can
 * hold matrices with more than 2^31 - 1 stored entries (Taylor-Hood a00 on 128^3 cells: 4.4 G) -- PETSc's
 * blocked insertion behind the reference (python/src/dolfinx_mpc/mpc.cpp:284-287) has no such limit either.
 * Row and column INDICES stay 32-bit. */
typedef int64_t mpcx_nnz_t;

/* ---- form kinds / cell types (element kernels replacing the FFCx-generated
 *      tabulate_tensor, cpp/assemble_matrix.cpp:438-439) ------------------- */
enum {
  MPCX_FORM_STIFFNESS = 0,
  MPCX_FORM_MASS = 1,
  MPCX_FORM_SOURCE = 2,
  MPCX_FORM_ELASTICITY = 3,
  MPCX_FORM_FACET_MASS = 4,
  MPCX_FORM_FACET_SOURCE = 5,
  MPCX_FORM_DIV_TEST = 6,  /* a(p, v) = c * p div(v) dx: vector test space, scalar trial space */
  MPCX_FORM_DIV_TRIAL = 7, /* a(u, q) = c * div(u) q dx: scalar test space, vector trial space */
  /* an imported UFCx tabulate_tensor (mpcx_ufcx_compile): mpcx_kernel_t::ufcx holds the handle */
  MPCX_FORM_UFCX = 100
};
/* MPCX_CELL_HEXAHEDRON (Q1, trilinear geometry, DOLFINx tensor-product vertex order: local vertex v sits at
 * (v & 1, v >> 1 & 1, v >> 2 & 1) of the reference cube) is known to the MPCX_ALG_CUBE kernels only: scalar
 * stiffness (nq = 8: 2 x 2 x 2 Gauss points) and source forms (nq = 1, 8 or 27), one thread per hexahedron.  Every other
 * path of a hexahedral mesh runs an imported kernel (MPCX_FORM_UFCX). */
enum { MPCX_CELL_TRIANGLE = 1, MPCX_CELL_TETRAHEDRON = 2, MPCX_CELL_HEXAHEDRON = 3 };

/* algorithms for the matrix scatter */
enum {
  MPCX_ALG_AUTO = 0,
  MPCX_ALG_ATOMIC = 1,  /* thread-per-entity, CSR binary search, device atomics */
  MPCX_ALG_ROWBLOCK = 2, /* LDS-privatised row blocks, each value written once */
  /* row blocks fed by CELL CLUSTERS (six P1 tets round a shared edge, eight vertices: what a structured box
   * generator emits per cube) instead of single cells: one thread sums the six element tensors in registers
   * and scatters 46 entries instead of 96; vectors: 8 contributions instead of 24.  Scalar P1 stiffness /
   * source forms on tetrahedra only; cells outside any cluster go through a second call with one of the
   * per-cell algorithms.  See mpcx_cube_records.
   * On hexahedra (kernel.celltype = MPCX_CELL_HEXAHEDRON) the unit of work is the cell itself: eight vertices, all
   * 64 pairs coupled (records of mpcx_hex_records, 96-byte format only; vectors: cube_verts = the cell dofmap). */
  MPCX_ALG_CUBE = 3
};

/* Built-in element kernel: (form, cell, degree, block size) + quadrature table.
 * The table is data (FFCx bakes it into generated code). */
typedef struct
{
  int32_t form;
  int32_t celltype;
  int32_t degree;  /* test space: Lagrange degree */
  int32_t bs;      /* test space: block size */
  int32_t degree1; /* trial space (= degree for square forms and rank-1 forms) */
  int32_t bs1;
  int32_t fn_id;
  int32_t coeff_degree;
  int32_t nq;
  int32_t nqf;
  const double* qpts;  /* DEVICE [nq][tdim] */
  const double* qwts;  /* DEVICE [nq] */
  const double* fqpts; /* DEVICE [nqf][tdim-1] */
  const double* fqwts; /* DEVICE [nqf] */
  const void* ufcx;    /* form == MPCX_FORM_UFCX: handle from mpcx_ufcx_compile; NULL otherwise */
  const double* qphi;  /* DEVICE [nq][nd] values of the test space's scalar basis at the cell rule's points, or NULL:
                        * lets the P2 source kernel take its basis from scalar loads instead of re-evaluating it */
  /* Scalar type T of the assembly (MPCX_SCALAR_*; 0 = fp64 real).  The reference instantiates the path for float32,
   * float64, complex64, complex128 (cpp/assemble_matrix.cpp:729-812).  For T != fp64 the `double*` fields that carry
   * VALUES -- vals / b, coeffs, constants, the constraints' coeffs, bc_values1, x0 -- point to arrays of T (complex:
   * interleaved re, im), geometry and quadrature tables stay fp64, and the general per-entity kernels run
   * (csrc/mpcx_scalar.hip: device atomics, no plan, row-side coefficients conjugated for complex T); built-in operators
   * on simplices only. */
  int32_t scalar_type;
  /* Source forms (MPCX_FORM_SOURCE) whose integrand function is AFFINE in x (constant, linear: fn_id 0, 4, 5) on affine
   * simplices, without a coefficient, optional: DEVICE [nv][nd], vphi[v][i] = sum_q qwts[q] phi_i(X_q) lambda_v(X_q), the
   * moments of the cell rule against the barycentric coordinates.  f(x(X_q)) = sum_v f(x_v) lambda_v(X_q) holds exactly
   * for such f, so sum_q w_q phi_i(X_q) f(x_q) = sum_v vphi[v][i] f(x_v): the kernels evaluate f at the nv vertices and
   * take nd * nv fma instead of walking the rule -- the SAME number as the quadrature sum up to rounding, whatever the
   * rule's degree (what FFCx does for piecewise-linear data on affine cells: weights folded into one table; the loop of
   * cpp/assemble_vector.cpp:65-90 is unchanged).  NULL: the rule is walked. */
  const double* vphi;
} mpcx_kernel_t;

/* ------------------------------------------------------------------------
 * UFCx import.  The reference's element seam is a host function pointer with the UFCx signature
 *   void tabulate_tensor(double* A, const double* w, const double* c, const double* coordinate_dofs,
 *                        const int* entity_local_index, const uint8_t* quadrature_permutation, void* custom_data)
 * (cpp/assemble_matrix.cpp:291-292, 438-439; python/src/dolfinx_mpc/numba/assemble_matrix.py:282-290).  Here the
 * seam is the C SOURCE of such a function (what FFCx writes to disk): mpcx_ufcx_compile turns it into a
 * gfx950 __device__ function with hipRTC and links it with generic per-entity assembly kernels for the given
 * element shape.  The handle goes into mpcx_kernel_t::ufcx with form = MPCX_FORM_UFCX; mpcx_assemble_matrix,
 * mpcx_assemble_vector and mpcx_apply_lifting then call it -- inside the LDS row-block kernels (MPCX_ALG_ROWBLOCK:
 * the same plan, masked dofmaps and scatter-offset table as for the built-in operators with lean = 0; vectors:
 * row blocks or the owner-computes variant; master contributions from the mpc_plan_* arrays when given) or in
 * thread-per-entity kernels with device atomics (MPCX_ALG_ATOMIC, no plan needed).  A is handed over zeroed and is
 * accumulated into, row-major [nd0*bs0][nd1*bs1] with blocked dof index i*bs + k, like the reference does; w holds
 * the packed coefficients of the entity (dolfinx pack_coefficients layout), c the packed constants.  #include lines
 * of the source are dropped (fixed-width integer types and the math functions are built in).
 * Compilation needs no device; NULL + mpcx_last_error() on failure (the compiler log is in the message). */
typedef struct
{
  const char* source;        /* HOST, NUL-terminated C source defining the function (and whatever it needs) */
  const char* function_name; /* HOST */
  int32_t rank;              /* 2: bilinear form (matrix, lifting), 1: linear form (vector) */
  int32_t nd0; int32_t bs0;  /* test space: dofs per cell, block size */
  int32_t nd1; int32_t bs1;  /* trial space (rank 2) */
  int32_t nv;                /* geometry nodes per cell */
  /* Dof transformations, optional (NULL: none -- every Lagrange element): names of functions DEFINED IN ``source`` with the
   * shape of the std::function the reference applies to the element tensor right after the kernel call
   * (cpp/assemble_matrix.cpp:432-436, 507-508; cpp/assemble_vector.cpp:184; cpp/lifting.h),
   *     void T(double* A, const uint32_t* cell_info, int32_t cell, int32_t n);
   * transform0_name: the test space's transformation, A = [nd0 * bs0][n] row-major (n = nd1 * bs1; 1 for a linear form):
   * acts on the ROWS of cell ``cell`` of the test space's mesh; transform1_name (bilinear forms): the trial space's transposed
   * transformation, A = [n][nd1 * bs1] with n = nd0 * bs0: acts on the COLUMNS.  cell_info is the per-cell permutation word
   * of the mesh (DOLFINx ``Topology::get_cell_permutation_info``), handed in through the argument blocks. */
  const char* transform0_name;
  const char* transform1_name;
} mpcx_ufcx_desc_t;
void* mpcx_ufcx_compile(const mpcx_ufcx_desc_t* desc);
/* ``source`` may be a WHOLE FFCx output file: the functions, then the descriptor objects DOLFINx reads (``ufcx_integral
 * integral_<hash> = { ..., .tabulate_tensor_float64 = <function>, ... };`` with its ``#ifndef __STDC_NO_COMPLEX__`` members,
 * the ``form_integrals_...`` arrays, ``ufcx_form form_<hash> = {...};``, the alias ``ufcx_form* form_<file>_<name> = &form_<hash>;``).
 * The objects are read and left out of the device translation unit; ``function_name`` may then name a function of the text, a
 * ufcx_integral object, a ufcx_form object or its alias (the form's first integral with a float64 kernel) -- the way the
 * reference reaches its kernels (cpp/assemble_matrix.cpp:438-439 through DOLFINx's ``form_integrals[k]->tabulate_tensor_float64``);
 * NULL: the file's only ufcx_integral.  mpcx_ufcx_resolve (HOST, no device, no compilation): the function ``name`` stands for,
 * NUL-terminated into out[out_len]; 0, or < 0 with mpcx_last_error(). */
int mpcx_ufcx_resolve(const char* source, const char* name, char* out, int32_t out_len);
/* 1 if the element tensor (nd0 * bs0 * nd1 * bs1 > 12288 entries, e.g. vector-valued Q3 hexahedra: 192 x 192) does not fit a
 * thread's private memory: such a kernel has the per-entity variants only (MPCX_ALG_ATOMIC, mpc_plan_off == NULL); the tensor
 * of a thread then lives in a slab of a scratch array the library allocates at first launch (256 MiB per kernel) and a
 * bounded grid strides over the entities.  Launches of ONE handle must not overlap on several streams beyond one matrix, one
 * master-contribution and one lifting launch. */
int mpcx_ufcx_big_tensor(void* handle);
/* 1 if the kernel was compiled with ROW-WISE copies of the text (bilinear forms with an element tensor of 37 .. 900 entries on
 * simplices of up to ten nodes per cell, no dof transformations; round 6): the imported function is inlined once per local node
 * row and only that row is kept -- the row-block kernel then runs a copy per row an entity keeps in the block, and
 * MPCX_ALG_ROWBLOCK also takes PAIR RECORDS (plan.row_pairs == 2, pair_recs of mpcx_pair_records, pair_ctx = pair_dict = NULL,
 * mdofmap1; mpcx_pairs_plan_create with x = NULL builds all of it): one (entity, node row) pair per lane, a wave runs one copy. */
int mpcx_ufcx_rowwise(void* handle);
int64_t mpcx_ufcx_code_size(void* handle); /* bytes of the gfx950 code object */
int mpcx_ufcx_code(void* handle, void* out); /* HOST out[mpcx_ufcx_code_size]: the code object (inspection, caching) */
void mpcx_ufcx_free(void* handle);

#define MPCX_SCALAR_F64 0
#define MPCX_SCALAR_F32 1
#define MPCX_SCALAR_C128 2
#define MPCX_SCALAR_C64 3

/* Finalized constraint as the kernels read it: the accessors of
 * cpp/MultiPointConstraint.h:155-199 (is_slave, masters, coefficients) as
 * flat arrays.  masters_offsets spans all local dofs (non-slaves have no
 * links), exactly like the reference adjacency lists. */
typedef struct
{
  const int8_t* is_slave;         /* DEVICE [num_dofs] */
  const int32_t* masters_offsets; /* DEVICE [num_dofs + 1] */
  const int32_t* masters;         /* DEVICE local unrolled dofs */
  const double* coeffs;           /* DEVICE */
} mpcx_mpc_t;

/* Optional row-block plan (MPCX_ALG_ROWBLOCK), built by mpcx_rowblock_plan_*. */
typedef struct
{
  int32_t num_blocks;
  int32_t max_rows;             /* max rows per block */
  int32_t max_nnz;              /* max nnz per block  */
  /* 0: block_ents lists the entities touching the block (thread per entity, rows outside the block masked).
   * 1: block_ents lists (entity, local row dof) pairs as entity * nd0 + i, only those whose rows lie in the
   *    block (thread per pair; operators with a compact context only: matrix_rowpair_kernel)
   * 2: pair RECORDS (mpcx_matrix_args_t::pair_recs, built by mpcx_pair_records from the pair list of mode 1): the
   *    kernel reads one self-contained record per pair and nothing else per entity but its cached context
   *    (matrix_pairs_kernel); block_ents / ent_offs are not read */
  int32_t row_pairs;
  const int32_t* block_row0;    /* DEVICE [num_blocks + 1] first row of block */
  const int64_t* block_ent_off; /* DEVICE [num_blocks + 1] into block_ents */
  const int32_t* block_ents;    /* DEVICE entity indices touching the block */
  /* DEVICE [n_entities][nd0][nd1] uint8: position of block column dofs1[j] inside
   * block row dofs0[i] of the CSR, counted in blocks from the row start
   * (mpcx_scatter_offsets); the scalar entry (i*bs0+k, j*bs1+l) lives at
   * rowptr[dofs0[i]*bs0+k] + off*bs1 + l */
  const uint8_t* ent_offs;
  /* Optional dictionary compression of ent_offs (mpcx_compress_offsets): ent_offs then is
   * the table of DISTINCT offset rows [num_patterns][nd0*nd1] and ent_pattern[e] (uint16)
   * selects the row of entity e; NULL = ent_offs is indexed by the entity directly.
   * Structured / tiled meshes have a few hundred distinct rows: 2 B per entity instead of nd0*nd1
   * (a memory saving; measured ~7% slower than the direct table at 256^3, so opt-in). */
  const uint16_t* ent_pattern;
} mpcx_rowblock_plan_t;

/* ------------------------------------------------------------------------
 * mpcx_assemble_matrix: replaces dolfinx_mpc::assemble_matrix
 * (cpp/assemble_matrix.h:28-43; cpp/assemble_matrix.cpp:417-548 cells,
 * :271-415 exterior facets, :99-268 modify_mpc_cell) for ONE integral of the
 * form, ADDing into the values of a pre-built CSR whose pattern is
 * create_sparsity_pattern's (cpp/utils.h:381-496).  The two PETSc insertion
 * callbacks of python/src/dolfinx_mpc/mpc.cpp:284-287 become direct
 * scatter-adds into `vals`.
 * ---------------------------------------------------------------------- */
typedef struct
{
  /* CSR, scalar (unrolled) rows/cols, columns sorted per row */
  int32_t nrows;
  const mpcx_nnz_t* rowptr; /* DEVICE [nrows+1] */
  const int32_t* cols;   /* DEVICE [nnz] */
  double* vals;          /* DEVICE [nnz], accumulated into */
  mpcx_kernel_t kernel;
  /* geometry: Geometry::x padded to 3 comps, one dofmap (cpp/assemble_matrix.cpp:462-470) */
  const double* x;         /* DEVICE [num_nodes][3] */
  const int32_t* x_dofmap; /* DEVICE [num_cells][nv] */
  int32_t nv;
  /* integration domain: Form::domain / domain_arg (cpp/assemble_matrix.cpp:625-630) */
  int32_t estride;          /* 1 cells, 2 (cell, local_facet) */
  int64_t n_entities;
  const int32_t* entities;  /* DEVICE [n_entities*estride]; NULL (all three, estride 1) = entity i is cell i */
  const int32_t* entities0; /* DEVICE, test-space cells (same layout) */
  const int32_t* entities1; /* DEVICE, trial-space cells */
  const double* coeffs;     /* DEVICE [n_entities][cstride] packed coefficients, or NULL */
  int32_t cstride;
  const double* constants;  /* DEVICE packed constants, or NULL */
  /* dofmaps (blocked) */
  const int32_t* dofmap0; int32_t nd0; int32_t bs0;
  const int32_t* dofmap1; int32_t nd1; int32_t bs1;
  /* Dirichlet markers over unrolled dofs, NULL when no bc (cpp/assemble_matrix.cpp:688-705) */
  const int8_t* bc0;
  const int8_t* bc1;
  mpcx_mpc_t mpc0; /* rows */
  mpcx_mpc_t mpc1; /* cols */
  /* entity indices (into the domain) whose cell holds a slave of mpc0 or mpc1:
   * compact form of cell_to_slaves (cpp/mpc_helpers.h:19-94) */
  const int32_t* slave_entities; /* DEVICE */
  int64_t n_slave_entities;
  int32_t algorithm;
  int32_t store_mode; /* rowblock: 1 = block values overwrite vals (no prior zeroing needed) */
  mpcx_rowblock_plan_t plan;
  /* rowblock: dofmaps with the "masked" flag (Dirichlet or slave, per component k)
   * folded into bit 28+k of each blocked dof: no marker gathers in the kernel */
  const int32_t* mdofmap0; /* DEVICE [num_cells][nd0] */
  const int32_t* mdofmap1; /* DEVICE [num_cells][nd1] */
  /* rowblock, lean path (P1-type square forms over all cells, no coefficients): the caller
   * guarantees dofmap0 == x_dofmap (ONE device array: dofs numbered like the mesh nodes),
   * mdofmap1 == mdofmap0, entities == NULL, and built mdofmap0 and plan.ent_offs with
   * rotate = 1; the kernel then reads only mdofmap0 + ent_offs per entity and takes the
   * geometry nodes from mdofmap0.  Checked; violating calls are rejected. */
  int32_t lean;
  /* Optional plan of the slave entities' master contributions (mpcx_mpc_plan_build), DEVICE, gathered by
   * target: position mpc_plan_tgt[t] of vals receives the sum over k in [off[t], off[t+1]) of
   * coef[k] * Ae(entity ent[k])[pq[k] / N1][pq[k] % N1].  mpc_plan_off == NULL: the kernel walks the
   * slave entities and searches the CSR rows itself (device atomics). */
  /* MPCX_ALG_CUBE: one 96-byte record per (row block, cluster) slot, built by mpcx_cube_records; plan carries
   * num_blocks / max_rows / max_nnz / block_row0 / block_ent_off (slots per block), nothing else; n_entities is
   * ignored (the slots say what is assembled) */
  const void* cube_recs;
  /* record format of THIS launch: 96 (or 0) = the records of mpcx_cube_records; 64 = narrow records (4-bit offsets,
   * mpcx_cube_pack_narrow) for row blocks all of whose slots have offsets < 16.  cube_block_ids (DEVICE [plan.num_blocks],
   * or NULL = blocks 0 .. num_blocks-1): the blocks this launch covers -- block j of the launch is row block
   * cube_block_ids[j] of plan.block_row0, its slots are [plan.block_ent_off[j], plan.block_ent_off[j+1]) of cube_recs.
   * The caller launches the narrow and the wide blocks separately (both with store_mode as for one launch). */
  int32_t cube_rec_bytes;
  /* hexahedra, bit 0: every cell of this launch's slots is a parallelepiped (mpcx_hex_slot_shapes says so for all of
   * them): the closed form of the stiffness integral is taken without looking, by a kernel instance that does not
   * carry the quadrature path (104 instead of 252 registers: twice the waves per SIMD).  The caller launches such row
   * blocks separately from the rest (cube_block_ids), like the two record formats of the tetrahedral clusters. */
  int32_t cube_flags;
  const int32_t* cube_block_ids;
  /* rowblock, component-diagonal forms on blocked spaces (bs0 == bs1 = bs > 1), optional: DEVICE [nnz / bs^2], one
   * byte per bs x bs block of the CSR (block s of node row n = entries rowptr[n*bs] / bs^2 + s), bit k set = entry
   * (k, k) of the block is a Dirichlet / slave row or column and stays zero (mpcx_diag_slot_mask).  Given, the
   * kernel keeps one LDS value per block (matrix_nodeblock_kernel); plan.max_rows / max_nnz still count scalar
   * rows / entries (LDS: max_nnz / bs^2 * 8 + (max_rows / bs + 1) * 4 bytes). */
  const uint8_t* slot_mask;
  int64_t mpc_plan_targets;
  const mpcx_nnz_t* mpc_plan_tgt;
  const int64_t* mpc_plan_off;
  const int32_t* mpc_plan_ent;
  const int32_t* mpc_plan_pq;
  const double* mpc_plan_coef;
  int32_t mpc_plan_group; /* lanes per target position: >= 16, >= 4 or one (pick ~ the average tuples per target) */
  /* Block-scalar value storage for component-diagonal forms on blocked spaces (S (x) I: vector stiffness / mass, the
   * Taylor-Hood velocity block), optional, with slot_mask: DEVICE [nnz / bs^2], ONE value per bs x bs block -- the
   * matrix is S (x) I except where slot_mask zeroes a diagonal entry and where constraints add couplings.  Given, the
   * node-block kernel writes (store_mode) / adds its block values here and never touches vals: 8 bytes per block instead
   * of 8 bs^2 (Taylor-Hood a00 on 128^3: 3.9 GB instead of 35 GB written per assembly).  The master contributions then go
   * to mpc_plan_out[t] (+=, one entry per plan target, zeroed by the caller) instead of vals[mpc_plan_tgt[t]], so a plan
   * is required when there are slave entities.  Consumers: mpcx_block_expand (the scalar CSR values, on demand) and
   * mpcx_spmv_blockscalar (y = A x straight from this layout). */
  double* block_vals;
  double* mpc_plan_out;
  /* Imported (UFCx) element kernels with a master-contribution plan, optional: the element tensor of every slave entity
   * is tabulated ONCE per call -- as the reference does (cpp/assemble_matrix.cpp:504-546: one tabulate_tensor per cell, then
   * modify_mpc_cell) -- into slave_tensors (DEVICE scratch of the caller, [N0 * N1][n_slave_entities] doubles, entry-major),
   * and the plan's tuples read their entry from it: mpc_plan_slot[k] = index of tuple k's entity in slave_entities
   * (DEVICE [tuples]).  NULL: every tuple re-tabulates its entity (a black-box kernel has no cheaper way to one entry). */
  double* slave_tensors;
  const int32_t* mpc_plan_slot;
  /* plan.row_pairs == 2 (matrix_pairs_kernel; cell integrals of operators with a compact per-entity context: stiffness
   * without coefficient on P1 / P2 scalar spaces, elasticity, the Taylor-Hood coupling blocks): pair_recs DEVICE
   * [n_pairs][W] uint32, W = mpcx_pair_words(nd1), pair t of plan.block_ent_off, ordered inside a block by local row:
   *   word 0       bits 0-26 entity index, bits 27-30 local row dof i, bit 31 = a column dof of the entity is masked
   *                (Dirichlet / slave: the kernel then reads the masks from mdofmap1, which may be NULL otherwise)
   *   bytes 4-5    uint16 row slot: bs0 == 1: index of the row's first value inside the block (rowptr[r] - rowptr[r0]),
   *                0xFFFF = masked row, nothing to do; bs0 > 1: bits 0-12 the row's node inside the block
   *                ((r - r0) / bs0), bit 13 + k = row of component k is masked
   *   bytes 6..    uint8 [nd1]: position of column block dofs1[j] inside CSR row dofs0[i] * bs0, counted in blocks
   * pair_ctx: DEVICE [n_entities][mpcx_pair_context_size] doubles, the constant-free context of every entity
   * (mpcx_pair_context; geometry only: rebuild when the mesh moves), or NULL = the kernel computes the context of every
   * pair from the coordinates (x, x_dofmap, entities). */
  const uint32_t* pair_recs;
  const double* pair_ctx;
  /* Optional dictionary of offset patterns (mpcx_pair_compress): pair_dict DEVICE [num_patterns][S] uint32, S =
   * mpcx_pair_dict_stride(nd1) (W - 1 rounded up to 2 or 4), entry = words 1 .. W-1 of the DISTINCT full records with the
   * row-slot bits cleared, zero padded; pair_recs then holds COMPACT records, two words
   * per pair: word 0 as above, word 1 = row slot | pattern id << 16.  NULL: full records. */
  const uint32_t* pair_dict;
  /* MPCX_ALG_CUBE, optional: DEVICE [slots of this launch] int32, cube_rec_index[t] = index into cube_recs of the record of
   * slot t.  A record (vertex ids + mask bits, scatter offsets relative to the row starts) is a property of the CLUSTER,
   * not of the (row block, cluster) slot: with an index the caller builds ONE record per cluster (mpcx_cube_records /
   * mpcx_hex_records with block_ents = 0 .. n_clusters-1) and a cluster touching several row blocks is read through the
   * L2 instead of being stored once per block.  Measured (round 4): 6 % less HBM traffic, but the extra dependent load costs
   * more than it saves (1.01 against 0.94 ms at 256^3 cubes) -- the Python host leaves it NULL unless
   * MPCX_CUBE_CLUSTER_RECORDS=1.  NULL: record t belongs to slot t. */
  const int32_t* cube_rec_index;
  /* MPCX_ALG_CUBE with an imported kernel (kernel.form == MPCX_FORM_UFCX; scalar P1 on tetrahedra, nd0 = nd1 = nv = 4, bs = 1):
   * the imported tabulate_tensor is called six times per cluster -- tet t with the coordinates of the cluster's local
   * vertices (0,1,3,7) (0,1,7,5) (0,5,7,4) (0,3,2,7) (0,6,4,7) (0,2,6,7) -- and the six tensors are summed per vertex pair in
   * registers (no symmetry assumed: 46 values, 46 scatter-adds).  The caller vouches that the mesh lists every cluster cell's
   * vertices in exactly that order (mpcx_cluster_ordered says which clusters do), so the function sees each cell as the
   * reference's loop would hand it over (cpp/assemble_matrix.cpp:495-506).  Forms with coefficients (coeffs != NULL)
   * additionally need cube_cells, DEVICE [n_clusters][6]: the CELL of table row t (index into coeffs), and
   * plan.block_ents = the cluster of every slot of this launch.  NULL otherwise. */
  const int32_t* cube_cells;
  /* imported kernels compiled with dof transformations (mpcx_ufcx_desc_t::transform0_name / transform1_name): DEVICE
   * [num_cells] uint32, the cell permutation words of the test / trial space's mesh (cpp/assemble_matrix.cpp:606-616);
   * NULL otherwise */
  const uint32_t* cell_info0;
  const uint32_t* cell_info1;
  /* Write-out through a permutation (the locality twin, dolfinx_mpc_amd/locality.py): rowptr / cols / every plan describe the
   * CSR of the launch, and the value of its entry k is written (added) to vals[val_map[k]] -- the caller's CSR of the same
   * matrix in another row / column numbering -- instead of vals[k].  DEVICE [nnz] uint32 (val_map_wide = 0) or int64 (1);
   * NULL: vals[k].  Honoured by every float64 matrix kernel (MPCX_VAL_POS); other scalar types reject it. */
  const void* val_map;
  int32_t val_map_wide;
  /* The same permutation in the order the row-block kernels write a block out of LDS (mpcx_write_out_order; optional, NULL:
   * val_map is used there as well): slot k of the launch's CSR writes the value of entry k + out_delta[k] (an entry of the
   * same row) to vals[out_map[k]], so that consecutive lanes write consecutive addresses of the caller's row -- val_map
   * alone scatters the 8-byte values of a row in permuted order (config 2 on the twin: matrix kernel 2.1 ms against
   * 1.0 ms for the unpermuted write-out).  out_map: the index type of val_map; out_delta: int16. */
  const void* out_map;
  const int16_t* out_delta;
  /* Occupancy cap of THIS launch (0: none): the row-block / pair / cluster kernel is launched with at least lds_floor bytes
   * of dynamic LDS per workgroup, so at most floor(160 KiB / lds_floor) of its workgroups share a CU and the rest of the CU
   * (LDS, wave slots, registers) stays free for a kernel of ANOTHER stream -- the VALU-bound vector kernel of the same
   * step beside the HBM-bound matrix kernel (dolfinx_mpc_amd/corun.py: the first part of a matrix launch is capped while
   * a vector assembly is in flight, the rest runs uncapped; sub-ranges of a plan are launched by advancing
   * plan.block_row0 / plan.block_ent_off and lowering plan.num_blocks).  Kernels without dynamic LDS ignore it. */
  int32_t lds_floor;
  void* stream;
} mpcx_matrix_args_t;
#define MPCX_VAL_POS(a, k)                                                                                                        \
  ((a).val_map ? ((a).val_map_wide ? ((const int64_t*)(a).val_map)[k] : (int64_t)((const uint32_t*)(a).val_map)[k]) : (int64_t)(k))
/* write-out of LDS slot i of a row block that starts at CSR position nnz0: vals[MPCX_OUT_POS(a, nnz0 + i)] (=, +=) the LDS
 * value at index MPCX_OUT_SRC(a, nnz0 + i, i) */
#define MPCX_OUT_POS(a, k)                                                                                                        \
  ((a).out_delta ? ((a).val_map_wide ? ((const int64_t*)(a).out_map)[k] : (int64_t)((const uint32_t*)(a).out_map)[k])              \
                 : MPCX_VAL_POS(a, k))
#define MPCX_OUT_SRC(a, k, i) ((a).out_delta ? (i) + (int)(a).out_delta[k] : (i))
/* the write-out of a row block (nnzb values from CSR position nnz0, LDS copy s_vals) by the NT threads of a workgroup; the
 * unpermuted case keeps its plain streaming loops (with the tests inside them the pair-record kernel lost 15 %) */
#define MPCX_WRITE_OUT(a, nnz0, nnzb, s_vals, tid, NT)                                                                            \
  do                                                                                                                               \
  {                                                                                                                                \
    if (!(a).val_map)                                                                                                              \
    {                                                                                                                              \
      if ((a).store_mode)                                                                                                          \
        for (int i_ = (tid); i_ < (nnzb); i_ += (NT))                                                                              \
          (a).vals[(nnz0) + i_] = (s_vals)[i_];                                                                                    \
      else                                                                                                                         \
        for (int i_ = (tid); i_ < (nnzb); i_ += (NT))                                                                              \
          (a).vals[(nnz0) + i_] += (s_vals)[i_];                                                                                   \
    }                                                                                                                              \
    else if ((a).store_mode)                                                                                                       \
      for (int i_ = (tid); i_ < (nnzb); i_ += (NT))                                                                                \
        (a).vals[MPCX_OUT_POS(a, (nnz0) + i_)] = (s_vals)[MPCX_OUT_SRC(a, (nnz0) + i_, i_)];                                       \
    else                                                                                                                           \
      for (int i_ = (tid); i_ < (nnzb); i_ += (NT))                                                                                \
        (a).vals[MPCX_OUT_POS(a, (nnz0) + i_)] += (s_vals)[MPCX_OUT_SRC(a, (nnz0) + i_, i_)];                                      \
  } while (0)

int mpcx_assemble_matrix(const mpcx_matrix_args_t* args);

/* Set-up for mpcx_matrix_args_t::slot_mask (all pointers DEVICE; bc / slave markers may be NULL): out [nnz / bs^2].
 * *bad is set if the CSR does not consist of whole bs x bs blocks (the bs rows of a node with the same block columns). */
int mpcx_diag_slot_mask(int32_t n_nodes, const mpcx_nnz_t* rowptr, const int32_t* cols, int32_t bs, const int8_t* bc0,
                        const int8_t* slave0, const int8_t* bc1, const int8_t* slave1, uint8_t* out, int32_t* bad,
                        void* stream);

/* Set-up for MPCX_ALG_ROWBLOCK: out[c][i] = d | (masked(d, k) << (28 + k)) with d = dofmap[c][i],
 * masked = Dirichlet-marked (bc may be NULL) or slave.  All pointers DEVICE, dof blocks must be
 * < 2^28, bs <= 3.  rotate != 0 (lean path, see mpcx_matrix_args_t::lean): cell c lists its local
 * dofs in the rotated order d = dofmap[c][(i + c mod nd) mod nd]. */
int mpcx_mask_dofmap(const int32_t* dofmap, int64_t num_cells, int32_t nd, int32_t bs, const int8_t* bc,
                     const int8_t* is_slave, int32_t rotate, int32_t* out, void* stream);

/* Set-up for MPCX_ALG_ROWBLOCK: the uint8 scatter-offset table described at
 * mpcx_rowblock_plan_t::ent_offs.  All pointers DEVICE.  *overflow (DEVICE int32,
 * zeroed by the caller) is set non-zero if an offset does not fit in 8 bits or
 * a column is missing from the pattern.  rotate != 0: local rows and columns of the cell c an
 * entity lies in are listed in the rotated order of mpcx_mask_dofmap.  entities0 / entities1 == NULL:
 * entity e is cell e. */
int mpcx_scatter_offsets(const mpcx_nnz_t* rowptr, const int32_t* cols, int32_t estride,
                         int64_t n_entities, const int32_t* entities0,
                         const int32_t* entities1, const int32_t* dofmap0, int32_t nd0,
                         int32_t bs0, const int32_t* dofmap1, int32_t nd1, int32_t bs1,
                         int32_t rotate, uint8_t* ent_offs, int32_t* overflow, void* stream);

/* Set-up for MPCX_ALG_CUBE (all pointers DEVICE): recs[k] (96 bytes: 8 x int32 vertex id with the Dirichlet /
 * slave mask of component c in bit 28 + c, then 64 x uint8 offsets [a][b] of column block v[b] inside the CSR rows
 * of block v[a], counted in blocks, for the 46 coupled vertex pairs; bs = block size of the space, 1..3) for every slot k of the row-block plan built over the clusters (mpcx_rowblock_plan_build with
 * dofmap0 = cube_verts, nd0 = 8): block_ents[k] is the cluster of slot k.  *overflow is set if an offset does
 * not fit 8 bits or a column is missing. */
int mpcx_cube_records(int64_t n_slots, const int32_t* block_ents, const int32_t* cube_verts, int32_t bs, const int8_t* bc,
                      const int8_t* is_slave, const mpcx_nnz_t* rowptr, const int32_t* cols, void* recs,
                      int32_t* overflow, void* stream);

/* The same records for hexahedra (cube_verts = the Q1 cell dofmap [n_cells][8], every one of the 64 vertex pairs
 * coupled): the reference's per-row column search behind MatSetValuesLocal (cpp/assemble_matrix.cpp:546), hoisted
 * to set-up like mpcx_scatter_offsets does for the per-cell kernels. */
int mpcx_hex_records(int64_t n_slots, const int32_t* block_ents, const int32_t* cell_verts, int32_t bs, const int8_t* bc,
                     const int8_t* is_slave, const mpcx_nnz_t* rowptr, const int32_t* cols, void* recs,
                     int32_t* overflow, void* stream);

/* general[k] = 1 if the hexahedron of record k is NOT a parallelepiped: one of the bilinear / trilinear coefficients of
 * its map exceeds 2^-46 of the largest edge-vector component -- the test matrix_hex_kernel applies per wave when
 * mpcx_matrix_args_t::cube_flags does not vouch for the launch (all pointers DEVICE; x [num_nodes][3]). */
int mpcx_hex_slot_shapes(int64_t n_slots, const void* recs, const double* x, uint8_t* general, void* stream);
/* the same test on a plain vertex array verts[n][8] (clusters of mpcx_cluster_build, hexahedron dofmaps) */
int mpcx_cell_shapes(int64_t n, const int32_t* verts, const double* x, uint8_t* general, void* stream);

/* Set-up of the P2 cluster kernel (MPCX_ALG_CUBE with a scalar P2 stiffness kernel; all pointers DEVICE).  A cluster of
 * six P2 tets carries 27 dofs -- its eight vertices and the 19 edges between coupled vertices -- and 393 coupled dof pairs
 * (six element tensors: 600 entries); on a parallelepiped cluster every entry is a fixed combination of six metric
 * entries (csrc/mpcx_cubes.hip, P2FAN).  mpcx_p2_cluster_dofs: dofs27[c][0..7] = the vertex dofs (local vertex b of
 * verts[c]), [8..26] = the edge dofs of the coupled vertex pairs (a < b) in row-major order, read from the P2 dofmaps
 * (vertices then edges, local edge e joins the local vertices {2,3},{1,3},{1,2},{0,3},{0,2},{0,1}[e]) of the cluster's
 * six cells fan_cells[c][0..5]; *bad is set if a dof is missing.  mpcx_p2_cluster_records: recs[c] (640 bytes: double M[6] = C^T C /
 * |det J| of the parallelepiped spanned by the local vertices 0, 1, 2, 4 (C = cofactor matrix; entries 00 01 02 11 12 22);
 * uint32 mask, bit I = dof I is a Dirichlet / slave dof; int32 dof[27]; from byte 160: uint8 position of column dof J in
 * CSR row dof I, the coupled columns of every row in ascending order, rows padded to multiples of four bytes); *overflow
 * is set if an offset does not fit 8 bits or a column is missing.  The records hold geometry: rebuild them when the mesh
 * moves.  The kernel takes a row-pair plan over the clusters (plan.row_pairs = 1, pair id = cluster * 27 + local dof),
 * cube_recs = recs, cube_rec_bytes = 640, cube_flags bit 0.  mpcx_p2_cluster_tables (HOST): the constant tables, for tests. */
int mpcx_p2_cluster_dofs(int64_t n, const int32_t* verts, const int32_t* fan_cells, const int32_t* x_dofmap,
                         const int32_t* dofmap, int32_t* dofs27, int32_t* bad, void* stream);
int mpcx_p2_cluster_records(int64_t n, const int32_t* verts, const int32_t* dofs27, const double* x, const int8_t* bc,
                            const int8_t* is_slave,
                            const mpcx_nnz_t* rowptr, const int32_t* cols, void* recs, int32_t* overflow, void* stream);
int mpcx_p2_cluster_tables(double* k, int32_t* coupled, int32_t* edge_vertices, int32_t* row_start);
/* (HOST, tests) the tables of the P1 closed-form kernels: k6 [6][8][8] (scalar stiffness on a tetrahedral cluster), k9
 * [9][8][8] (the sums of gradient products behind the elasticity kernel), hex6 [6][8][8] (Q1 stiffness on a hexahedron);
 * first index: the metric entry 00 01 02 11 12 22 (k9: d * 3 + e) */
int mpcx_p1_cluster_tables(double* k6, double* k9, double* hex6);

/* The whole set-up of MPCX_ALG_CUBE for a scalar P1 stiffness integral over ALL cells of a tetrahedral mesh in one call,
 * with device memory the library allocates itself -- for callers without torch (a C++ binding inside
 * python/src/dolfinx_mpc/mpc.cpp): cluster detection (mpcx_cluster_keys .. mpcx_cluster_canonical), row blocks
 * (mpcx_block_ranges on rowptr_host, max_rows / max_nnz as for mpcx_rowblock_plan_*; row_hints: HOST, first row of every
 * tile of a tiled numbering, or NULL), the (block, cluster) slots, their records, and the split of the row blocks by
 * record format and cluster shape.  x_dofmap [n_cells][4], x [n_nodes][3], rowptr / cols, bc (or NULL), is_slave: DEVICE;
 * rowptr_host: the same offsets on the HOST.  The records hold Dirichlet / slave masks and depend on the coordinates
 * through the split: rebuild when the constraint, the boundary conditions or the mesh geometry change.
 *   mpcx_cluster_plan_num_parts   launches needed (1 .. 4); for part p, mpcx_cluster_plan_part fills plan, cube_recs,
 *                                 cube_rec_bytes, cube_flags, cube_block_ids and algorithm of *args (everything else --
 *                                 CSR, kernel, geometry, mpc, store_mode, slave entities on the LAST part -- is the caller's)
 *   mpcx_cluster_plan_leftover    cells in no cluster (DEVICE, ascending; owned by the plan): a per-cell call adds them
 *   mpcx_cluster_plan_verts       DEVICE [num_clusters][8]: mpcx_vector_args_t::cube_verts of the vector kernels
 * Returns 0, -21 if a scatter offset does not fit 8 bits (use MPCX_ALG_ROWBLOCK / MPCX_ALG_ATOMIC then). */
typedef struct mpcx_cluster_plan mpcx_cluster_plan_t;
int mpcx_cluster_plan_create(int64_t n_cells, const int32_t* x_dofmap, int64_t n_nodes, const double* x, int32_t nrows,
                             const mpcx_nnz_t* rowptr, const mpcx_nnz_t* rowptr_host, const int32_t* cols, const int8_t* bc,
                             const int8_t* is_slave, int32_t max_rows, int32_t max_nnz, const int32_t* row_hints, int32_t n_hints,
                             void* stream, mpcx_cluster_plan_t** plan);
int32_t mpcx_cluster_plan_num_parts(const mpcx_cluster_plan_t* plan);
int64_t mpcx_cluster_plan_num_clusters(const mpcx_cluster_plan_t* plan);
int64_t mpcx_cluster_plan_num_slots(const mpcx_cluster_plan_t* plan);
const int32_t* mpcx_cluster_plan_verts(const mpcx_cluster_plan_t* plan);
int64_t mpcx_cluster_plan_leftover(const mpcx_cluster_plan_t* plan, const int32_t** cells);
int mpcx_cluster_plan_part(const mpcx_cluster_plan_t* plan, int32_t part, mpcx_matrix_args_t* args);
void mpcx_cluster_plan_destroy(mpcx_cluster_plan_t* plan);

/* Narrow records (all pointers DEVICE): mpcx_cube_slot_width: wide[k] = 1 if a coupled offset of record k exceeds 15;
 * mpcx_cube_pack_narrow: out[j] (64 bytes: the 8 ids, then 46 nibbles in row-major order of the coupled pairs) from the
 * 96-byte record src[j]. */
int mpcx_cube_slot_width(int64_t n_slots, const void* recs, uint8_t* wide, void* stream);
int mpcx_cube_pack_narrow(int64_t n_out, const int64_t* src, const void* recs, void* out, void* stream);

/* Set-up for MPCX_ALG_CUBE (DEVICE): for every group g of six consecutive cells (x_dofmap rows 6g .. 6g+5)
 * verts[g][0..7] = the eight vertices read off the fan pattern, ok[g] = 1 if the six cells really form it. */
int mpcx_cube_detect(const int32_t* cells, int64_t n_groups, int32_t* verts, int8_t* ok, void* stream);

/* Set-up for MPCX_ALG_CUBE on ANY cell order and local vertex order (all pointers DEVICE), three steps:
 *   1. mpcx_cluster_keys: keys[c] = (vmin << 32) | vmax of the LONGEST edge of tet c (ties: the smaller pair) -- the
 *      body diagonal for a cube cut into six tets;
 *   2. the caller sorts the cell indices by key (order[], sorted_keys[]);
 *   3. mpcx_cluster_build: a run of exactly six equal keys whose other vertices close into one ring of six distinct
 *      vertices is a cluster: ok[p] = 1 at the run's first position p, verts[p][0..7] = its vertices in the local
 *      numbering of mpcx_vector_args_t::cube_verts, cell_in_fan[c] = 1 for its six cells (zeroed by the caller).
 * The caller compacts verts by ok; cells with cell_in_fan == 0 go through the per-cell kernels. */
int mpcx_cluster_keys(const double* x, const int32_t* cells, int64_t n_cells, int64_t* keys, void* stream);
int mpcx_cluster_build(int64_t n, const int64_t* sorted_keys, const int32_t* order, const int32_t* cells, int32_t* verts,
                       int8_t* ok, int8_t* cell_in_fan, void* stream);
/*   4. mpcx_cluster_canonical (optional; ok may be NULL = every position): the ring walk of step 3 starts at an arbitrary
 *      ring vertex; a fan that is a parallelepiped once its ring is turned by one position (local vertices 1, 2, 4 the
 *      cube-edge neighbours of vertex 0) is renumbered so -- what the closed-form kernels assume (cube_flags bit 0).  */
int mpcx_cluster_canonical(int64_t n, int32_t* verts, const int8_t* ok, const double* x, void* stream);
/*   5. mpcx_cluster_ordered (imported kernels only): verts [n][8] and fan_cells [n][6] (the six cells of every fan, any order)
 *      are renumbered / reordered IN PLACE so that cell fan_cells[p][t] lists its vertices as row t of the table
 *      (0,1,3,7) (0,1,7,5) (0,5,7,4) (0,3,2,7) (0,6,4,7) (0,2,6,7) in the numbering verts[p]; ok[p] = 0 where no numbering
 *      does that (cells listed in another local order: an imported tabulate_tensor must be called with the mesh's own vertex
 *      order, so those fans stay with the per-cell kernels).  All pointers DEVICE. */
int mpcx_cluster_ordered(int64_t n, int32_t* verts, int32_t* fan_cells, const int32_t* x_dofmap, int8_t* ok, void* stream);

/* The entity lists of a row-block plan on the DEVICE (host version: second half of mpcx_rowblock_plan_build):
 * (block, entity) pairs in entity order; two calls like mpcx_mpc_plan_device (offsets == NULL: counts[e] =
 * number of distinct blocks entity e touches; then with the exclusive scan of counts: the pairs).  The caller
 * sorts the pairs by block (stable) to get block_ents / block_ent_off.  block_row0 [num_blocks + 1] DEVICE.
 * pair_rows (optional): bit i set = local dof i of the entity has its rows inside the pair's block; entities of a
 * block ordered by this word make the lanes of a wave skip the same local rows together (rotate != 0: bits in
 * the rotated local order of the lean path, mpcx_mask_dofmap). */
int mpcx_rowblock_pairs_device(int64_t n_entities, int32_t estride, const int32_t* entities0, const int32_t* dofmap0,
                               int32_t nd0, int32_t bs0, int32_t num_blocks, const int32_t* block_row0, int32_t* counts,
                               const int64_t* offsets, int32_t* pair_block, int32_t* pair_ent, int32_t* pair_rows,
                               int32_t rotate, void* stream);

/* Set-up of matrix_pairs_kernel (plan.row_pairs == 2; all pointers DEVICE).  mpcx_pair_words: 32-bit words per record.
 * mpcx_pair_records: one record (layout: mpcx_matrix_args_t::pair_recs) per pair id = entity * nd0 + i of pair_ids
 * [n_pairs] -- the list mode 1 of the row-block plan uses, ordered by (block, local row, ...) -- from the dofmaps, the
 * Dirichlet / slave markers (may be NULL) and the CSR pattern; this is the per-row column search behind the reference's
 * MatSetValuesLocal (cpp/assemble_matrix.cpp:546) hoisted to set-up.  *overflow (zeroed by the caller) is set when a
 * record cannot be written: an offset beyond 8 bits or a column missing from the pattern (1), a row slot beyond 16 / 13
 * bits (2), an entity index beyond 27 bits or nd0 > 16 (4).
 * mpcx_pair_context_size: doubles per entity of the cached context for this kernel descriptor (0: no such operator);
 * mpcx_pair_context: ctx [n_entities][size] from the coordinates (entities == NULL: entity e is cell e). */
int32_t mpcx_pair_words(int32_t nd1);
int mpcx_pair_records(int64_t n_pairs, const uint32_t* pair_ids, int32_t estride, const int32_t* entities0,
                      const int32_t* entities1, const int32_t* dofmap0, int32_t nd0, int32_t bs0, const int32_t* dofmap1,
                      int32_t nd1, int32_t bs1, const int8_t* bc0, const int8_t* slave0, const int8_t* bc1,
                      const int8_t* slave1, const mpcx_nnz_t* rowptr, const int32_t* cols, int32_t num_blocks,
                      const int32_t* block_row0, uint32_t* recs, int32_t* overflow, void* stream);
/* Dictionary compression of the records (all pointers DEVICE except num_patterns): recs2 [n_pairs][2] compact records,
 * table [65535][mpcx_pair_dict_stride(nd1)] the distinct offset patterns, workspace of mpcx_pair_compress_workspace(nd1)
 * bytes.  Blocking.
 * *num_patterns (HOST) = number of distinct patterns, or -1 when there are more than 65535 (an unstructured mesh): keep
 * the full records then.  Structured / tiled meshes have a few thousand patterns whatever their size. */
int32_t mpcx_pair_dict_stride(int32_t nd1);
int64_t mpcx_pair_compress_workspace(int32_t nd1);
int mpcx_pair_compress(int64_t n_pairs, const uint32_t* recs, int32_t nd1, uint32_t* recs2, uint32_t* table,
                       int32_t* num_patterns, void* workspace, void* stream);
int32_t mpcx_pair_context_size(const mpcx_kernel_t* kernel);
int mpcx_pair_context(const mpcx_kernel_t* kernel, int64_t n_entities, int32_t estride, const int32_t* entities,
                      const double* x, const int32_t* x_dofmap, int32_t nv, double* ctx, void* stream);

/* vals[pos(d,d)] += diagval for d in dofs.  Replaces the slave-diagonal loop
 * of cpp/assemble_matrix.cpp:711-724 and dolfinx insert_diagonal called at
 * python/src/dolfinx_mpc/assemble_matrix.py:59-62. */
int mpcx_add_diagonal(int32_t nrows, const mpcx_nnz_t* rowptr, const int32_t* cols,
                      double* vals, const int32_t* dofs, int64_t n,
                      double diagval, void* stream);
/* the same for a launch that writes through mpcx_matrix_args_t::val_map: vals[val_map[pos(d,d)]] += diagval */
int mpcx_add_diagonal_mapped(int32_t nrows, const mpcx_nnz_t* rowptr, const int32_t* cols, double* vals, const int32_t* dofs,
                             int64_t n, double diagval, const void* val_map, int32_t val_map_wide, void* stream);

/* ------------------------------------------------------------------------
 * mpcx_assemble_vector: replaces dolfinx_mpc::assemble_vector
 * (cpp/assemble_vector.h:77-81, cpp/assemble_vector.cpp:34-91, modify_mpc_vec
 * cpp/assemble_vector.h:35-69) for one integral; accumulates into b (not
 * zeroed, like the reference).
 * ---------------------------------------------------------------------- */
typedef struct
{
  double* b; /* DEVICE [num_dofs] */
  int32_t num_dofs;
  mpcx_kernel_t kernel;
  const double* x; const int32_t* x_dofmap; int32_t nv;
  int32_t estride; int64_t n_entities;
  const int32_t* entities; const int32_t* entities0;
  const double* coeffs; int32_t cstride; const double* constants;
  const int32_t* dofmap; int32_t nd; int32_t bs;
  mpcx_mpc_t mpc;
  /* MPCX_ALG_ATOMIC: LDS hash per workgroup + one device atomic per distinct dof.
   * MPCX_ALG_ROWBLOCK: a workgroup owns a contiguous range of rows of b in LDS (plan: row blocks
   * and entity lists from mpcx_rowblock_plan_build, ent_offs unused) and adds it to b once,
   * without atomics; rows of slave dofs are skipped there and a second kernel over
   * `slave_entities` sends them to their masters (cpp/assemble_vector.h:35-69).
   * MPCX_ALG_AUTO: rowblock if a plan is given (the caller decides: row blocks pay off for
   * cheap integrands, see vector_rowblock_kernel). */
  int32_t algorithm;
  mpcx_rowblock_plan_t plan;
  const int32_t* mdofmap;        /* DEVICE [num_cells][nd]: slave flag in bit 28+k (mpcx_mask_dofmap, bc = NULL) */
  const int32_t* slave_entities; /* DEVICE entity indices whose cell holds a slave */
  int64_t n_slave_entities;
  /* MPCX_ALG_CUBE: vertex (= dof) ids of the clusters, DEVICE [n_cubes][8], local vertex b of a cluster has
   * bit0 = x, bit1 = y, bit2 = z of the cube corner; its six tets are (0,1,3,7) (0,1,7,5) (0,5,7,4) (0,3,2,7)
   * (0,6,4,7) (0,2,6,7).  entities / n_entities are ignored.  With own_lmap != NULL (owner-computes, see below:
   * plan.block_ents = the clusters a block owns, own_lmap [n_cubes][8], bs = 1) there is no hash table and no device
   * atomic; slave rows are then skipped (flag in own_lmap) and sent to their masters from slave_entities (CELL
   * indices of the clusters' cells that hold a slave). */
  const int32_t* cube_verts;
  int64_t n_cubes;
  /* MPCX_ALG_ROWBLOCK, owner-computes variant (own_lmap != NULL): plan.block_ents lists every entity ONCE, in the
   * block that holds the rows of its local dof 0.  The LDS copy of a block holds its own dofs followed by the dofs of
   * other blocks its entities touch (its halo): own_lmap[e][i] = LDS position (in dofs) of local dof i of entity e,
   * slave flags of component k in bit 28 + k as in mdofmap; plan.max_rows counts own + halo rows.  The own part is
   * added to b, the halo part is written to own_spill[(own_hoff[b] + j) * bs + k] (contiguous, no per-thread stores),
   * and a second kernel adds the entries own_spill[own_src[s] * bs + k], s in [own_seg[u], own_seg[u + 1]), to the
   * rows of dof own_rows[u].  No entity is evaluated twice, nothing is added atomically outside LDS. */
  const int32_t* own_lmap;  /* DEVICE [n_entities][nd] */
  const int64_t* own_hoff;  /* DEVICE [num_blocks + 1] halo dofs of the blocks */
  double* own_spill;        /* DEVICE [halo dofs * bs] */
  const int32_t* own_src;   /* DEVICE [halo dofs] halo slots ordered by target dof */
  const int32_t* own_rows;  /* DEVICE [n_own_rows] distinct target dofs (blocked), ascending */
  const int64_t* own_seg;   /* DEVICE [n_own_rows + 1] */
  int64_t n_own_rows;
  /* MPCX_ALG_CUBE with an imported kernel and coefficients: DEVICE [n_cubes][6], the cell of every cluster tet in the order
   * (0,1,3,7) (0,1,7,5) (0,5,7,4) (0,3,2,7) (0,6,4,7) (0,2,6,7) (see mpcx_matrix_args_t::cube_cells); NULL otherwise */
  const int32_t* cube_cells;
  const uint32_t* cell_info0; /* the same for the linear form's space, cpp/assemble_vector.cpp:184, or NULL */
  /* write-out through a permutation (the locality twin, as mpcx_matrix_args_t::val_map): everything the launch adds to
   * entry d of its own numbering goes to b[row_map[d]] -- DEVICE [scalar dofs] int32 -- instead of b[d]; NULL: b[d].
   * float64 kernels only. */
  const int32_t* row_map;
  /* MPCX_ALG_CUBE, owner-computes, kernel.fn_id = 1 (python/benchmarks/bench_periodic.py:85-89) with the 14-point rule of
   * degree 5 as kernel.qpts / qwts (the values of csrc/mpcx_box14.hpp; the launch compares and traps on a mismatch), every
   * cluster an axis-aligned box: the clusters on a TENSOR GRID of intervals (optional; grid_idx = NULL: every cluster from
   * its own vertices).  Interval i of axis d is (grid_iv[2 r], grid_iv[2 r + 1]) = (coordinate of corner 0, of corner 7),
   * r = i + grid_n[0] (d >= 1) + grid_n[1] (d = 2); grid_idx DEVICE [n_cubes][4] = the intervals (x, y, z, unused) of every
   * cluster.  The launch first fills grid_tab (DEVICE scratch, MPCX_GRID_ROW doubles per interval) with the univariate
   * factors of the right-hand side at the 19 coordinates the rule puts into an interval -- f = x sin(5 pi y) + g(x) g(y) g(z)
   * is a sum of products of them -- and the cluster kernel reads them instead of evaluating 84 sines and exponentials per
   * cluster: (n_x + n_y + n_z) * 19 evaluations per factor and LAUNCH (nothing is kept between launches).
   * grid_block_rows (optional): when the clusters of every block of `plan` sit on few intervals (a tile of the numbering
   * does) the blocks keep their rows of the table in LDS: DEVICE [num_blocks][MPCX_GRID_BLOCK_ROWS] lists, per block, the
   * table rows r (as above) it needs, -1 = unused -- and grid_idx then holds, per cluster, the POSITIONS of its three rows
   * in the list of the block that owns it instead of interval numbers; grid_block_rows_max = the longest list (LDS is sized
   * by it; 0: MPCX_GRID_BLOCK_ROWS).  NULL: grid_idx holds interval numbers and every cluster reads the table itself. */
  /* MPCX_ALG_CUBE, owner-computes, kernel.fn_id = 1: nonzero = many clusters are axis-aligned boxes with their vertices in
   * corner order (mpcx_cluster_canonical numbers them so): the launch takes the instance that checks every cluster and
   * evaluates boxes factor by factor (csrc/mpcx_cubes.hip box14_source_fn1); 0: every cluster point by point (the lighter
   * instance: a mesh without boxes runs a third faster on it) */
  int32_t cube_boxes;
  const int32_t* grid_idx;
  const double* grid_iv;
  double* grid_tab;
  int32_t grid_n[3];
  const int32_t* grid_block_rows;
  int32_t grid_block_rows_max;
  /* The same per CELL (MPCX_ALG_ROWBLOCK, owner-computes, scalar P1 / P2 source with kernel.fn_id = 1, any rule, no
   * coefficient): a simplex whose vertices take two values per axis (every cell of a box mesh) has its quadrature points at
   * x_d = lo_d + h_d eta, eta = the sum of the barycentric coordinates of the vertices on the high side -- one of grid_ng
   * (<= 255) values grid_eta (DEVICE [grid_ng]) for every point q and vertex subset.  Cells with the same subsets per axis
   * and the same |det J| / (h_x h_y h_z) are of one TYPE (a box mesh has a handful): grid_J DEVICE [grid_ntypes][nq] words,
   * the index of eta along x | y << 8 | z << 16 for point q of a cell of that type.  grid_idx then holds per cell (row_x, row_y,
   * row_z, type | (|det J| / (h_x h_y h_z)) << 16), rows = positions in its block's list (grid_block_rows is required); rows of
   * the table: 2 * ((grid_ng + 1) & ~1) + 2 doubles.  NULL: not used. */
  const double* grid_eta;
  const uint32_t* grid_J;
  int32_t grid_ng;
  int32_t grid_ntypes;
  int32_t lds_floor; /* as mpcx_matrix_args_t::lds_floor: minimum dynamic LDS per workgroup of the row-block / cluster launch */
  void* stream;
} mpcx_vector_args_t;
#define MPCX_GRID_ROW 40 /* doubles per interval of mpcx_vector_args_t::grid_tab */
#define MPCX_GRID_BLOCK_ROWS 128 /* table rows a block can keep in LDS (mpcx_vector_args_t::grid_block_rows) */
#define MPCX_ROW_POS(a, d) ((a).row_map ? (int64_t)(a).row_map[d] : (int64_t)(d))

int mpcx_assemble_vector(const mpcx_vector_args_t* args);

/* One launch for the matrix AND the vector of the periodic benchmark's forms (scalar P1 stiffness on parallelepiped clusters
 * with narrow records + scalar P1 source with the owner-computes cluster plan, MPCX_ALG_CUBE both): a workgroup keeps the LDS
 * copy of one row block of the CSR and of the same rows of b, so that the memory-bound matrix half and the arithmetic-bound
 * vector half of neighbouring workgroups overlap on a CU (csrc/mpcx_cubes.hip fused_cube_kernel).  The two plans must use the
 * same row blocks: vargs->plan carries all of them; part_index (DEVICE [vargs->plan.num_blocks]) = index of a row block in the
 * matrix launch margs (margs->plan.block_ent_off, cube_recs) or -1 for row blocks the caller launches through
 * mpcx_assemble_matrix on its own (other record format / cluster shape); runs on vargs->stream, followed by the vector's
 * halo reduction and slave rows.  The reference has no counterpart (it assembles A and b in separate calls:
 * python/benchmarks/bench_periodic.py:97-108).  MEASURED AND NOT TAKEN by the Python layer: 3.72 ms at 256^3 against 3.50 ms
 * for the two launches on two streams; tests/test_gpu_fused.py keeps its values checked. */
int mpcx_assemble_fused(const mpcx_matrix_args_t* margs, const mpcx_vector_args_t* vargs, const int32_t* part_index);

/* ------------------------------------------------------------------------
 * mpcx_apply_lifting: replaces impl::apply_lifting for one integral of one
 * form a[j] (cpp/lifting.h:45-134, :243-397):
 *    b <- b - scale * K^T A_j (g_j - x0_j)
 * `lift_entities` is the compact list of entity indices with a bc-marked
 * column dof (the `has_bc` test of cpp/lifting.h:93-109 hoisted to set-up).
 * ---------------------------------------------------------------------- */
typedef struct
{
  double* b; int32_t num_dofs;
  mpcx_kernel_t kernel;
  const double* x; const int32_t* x_dofmap; int32_t nv;
  int32_t estride; int64_t n_entities;
  const int32_t* entities; const int32_t* entities0; const int32_t* entities1;
  const double* coeffs; int32_t cstride; const double* constants;
  const int32_t* dofmap0; int32_t nd0; int32_t bs0;
  const int32_t* dofmap1; int32_t nd1; int32_t bs1;
  const int8_t* bc_markers1; /* DEVICE, cpp/lifting.h:166-180 */
  const double* bc_values1;  /* DEVICE */
  const double* x0;          /* DEVICE or NULL (=> 0, cpp/lifting.h:295) */
  double scale;
  const int32_t* lift_entities; int64_t n_lift_entities; /* DEVICE */
  mpcx_mpc_t mpc0;
  const uint32_t* cell_info0; /* as in mpcx_matrix_args_t, or NULL */
  const uint32_t* cell_info1;
  const int32_t* row_map; /* as in mpcx_vector_args_t, or NULL */
  void* stream;
} mpcx_lifting_args_t;

int mpcx_apply_lifting(const mpcx_lifting_args_t* args);

/* MultiPointConstraint::backsubstitution / homogenize
 * (cpp/MultiPointConstraint.h:129-152) on a device vector. */
int mpcx_backsubstitution(double* u, const int32_t* slaves, int64_t num_slaves,
                          const mpcx_mpc_t* mpc, void* stream);
int mpcx_homogenize(double* u, const int32_t* slaves, int64_t num_slaves, void* stream);

/* ------------------------------------------------------------------------
 * HOST set-up routines (no GPU needed).
 * ---------------------------------------------------------------------- */

/* MultiPointConstraint constructor, cpp/MultiPointConstraint.h:36-126, single
 * process (global master index == local index, owners kept verbatim).
 * Outputs are caller-allocated:
 *   is_slave[num_dofs], sorted_slaves[num_slaves], masters_offsets[num_dofs+1],
 *   masters_out/coeffs_out/owners_out[offsets[num_slaves]]. */
int mpcx_mpc_finalize(int32_t num_dofs, int32_t num_owned_dofs, int32_t num_slaves,
                      const int32_t* slaves, const int64_t* masters,
                      const double* coeffs, const int32_t* owners,
                      const int32_t* offsets, int8_t* is_slave,
                      int32_t* sorted_slaves, int32_t* num_local_slaves,
                      int32_t* masters_offsets, int32_t* masters_out,
                      double* coeffs_out, int32_t* owners_out);

/* The same two set-up steps on the DEVICE (all pointers DEVICE unless said otherwise).
 * mpcx_mpc_finalize_device: outputs as mpcx_mpc_finalize; sorted_slaves must hold num_slaves entries, the number of
 * distinct slaves is masters-independent and comes back through the compaction (read num_local_slaves / count
 * is_slave); work: int32 [2 * num_dofs + 1]; *flag (zeroed by the caller): bit 0 slave index out of range, bit 1
 * master index out of range, bit 2 a dof listed twice as a slave (use the host routine).  temp / temp_bytes: rocPRIM
 * workspace, size query with temp == NULL (HOST pointer temp_bytes).
 * mpcx_cell_to_slaves_device: two calls -- c2s == NULL: counts[c] = slaves of cell c; the caller scans counts into
 * c2s_offsets (mpcx_scan_exclusive_i32) and calls again with c2s: each cell's slaves ascending by dof. */
int mpcx_mpc_finalize_device(int32_t num_dofs, int32_t num_owned_dofs, int32_t num_slaves, const int32_t* slaves,
                             const int64_t* masters, const double* coeffs, const int32_t* owners, const int32_t* offsets,
                             int8_t* is_slave, int32_t* sorted_slaves, int32_t* num_local_slaves, int32_t* masters_offsets,
                             int32_t* masters_out, double* coeffs_out, int32_t* owners_out, int32_t* work, int32_t* flag,
                             void* temp, size_t* temp_bytes, void* stream);
int mpcx_cell_to_slaves_device(int64_t num_cells, int32_t nd, int32_t bs, const int32_t* dofmap, const int8_t* is_slave,
                               int32_t* counts, const int32_t* c2s_offsets, int32_t* c2s, void* stream);

/* Device primitives of the set-up path (rocPRIM behind the C ABI, so that a caller without torch can build plans):
 * exclusive scans (out has n + 1 entries, the last one is the total), stable LSD radix sort of (key, value) pairs on
 * key bits [begin_bit, end_bit), run boundaries of a sorted key array (heads[i] = 1 at the first element of a run;
 * with the exclusive scan of heads: run_keys[r], run_start[r], run_start[num_runs] = n).  Workspace: called with
 * temp == NULL they only write the bytes they need to *temp_bytes (HOST pointer). */
int mpcx_scan_exclusive_i32_i64(const int32_t* in, int64_t n, int64_t* out, void* temp, size_t* temp_bytes, void* stream);
int mpcx_scan_exclusive_i32(const int32_t* in, int64_t n, int32_t* out, void* temp, size_t* temp_bytes, void* stream);
int mpcx_scan_exclusive_i64(const int64_t* in, int64_t n, int64_t* out, void* temp, size_t* temp_bytes, void* stream);
/* out[b] = first position of sorted_keys whose (key >> shift) >= b, for b = 0 .. num_segments (out has num_segments + 1
 * entries): the segment offsets of a sorted (segment id << shift | ...) key array, empty segments included */
int mpcx_segment_offsets(const int64_t* sorted_keys, int64_t n, int32_t shift, int64_t num_segments, int64_t* out, void* stream);
int mpcx_sort_pairs_i64_i32(const int64_t* keys_in, int64_t* keys_out, const int32_t* vals_in, int32_t* vals_out, int64_t n,
                            int32_t begin_bit, int32_t end_bit, void* temp, size_t* temp_bytes, void* stream);
int mpcx_sort_pairs_i64_i64(const int64_t* keys_in, int64_t* keys_out, const int64_t* vals_in, int64_t* vals_out, int64_t n,
                            int32_t begin_bit, int32_t end_bit, void* temp, size_t* temp_bytes, void* stream);
int mpcx_run_heads(const int64_t* sorted_keys, int64_t n, int32_t* heads, void* stream);
int mpcx_run_fill(const int64_t* sorted_keys, const int32_t* heads, const int64_t* heads_scan, int64_t n, int64_t* run_keys,
                  int64_t* run_start, void* stream);

/* create_cell_to_dofs_map, cpp/mpc_helpers.h:19-94.  Call with c2s == NULL to
 * fill c2s_offsets[num_cells+1] and get the total; then again with c2s
 * allocated.  Returns total number of links (>=0) or <0. */
int64_t mpcx_cell_to_slaves(int64_t num_cells, int32_t nd, int32_t bs,
                            const int32_t* dofmap, const int8_t* is_slave,
                            int32_t* c2s_offsets, int32_t* c2s);

/* create_sparsity_pattern, cpp/utils.h:381-496 (+ finalize): block pattern
 *   rows(c) x (cols(c) U masters(col-slaves(c)))  U
 *   masters(row-slaves(c)) x (cols(c) U masters(col-slaves(c)))   for all cells c
 * expanded to scalar CSR with sorted columns.  Returns an opaque handle. */
void* mpcx_pattern_build(int64_t num_cells, const int32_t* dofmap0, int32_t nd0,
                         int32_t bs0, int32_t num_blocks0, const int32_t* dofmap1,
                         int32_t nd1, int32_t bs1, int32_t num_blocks1,
                         const int32_t* c2s_offsets0, const int32_t* c2s0,
                         const int32_t* masters_offsets0, const int32_t* masters0,
                         const int32_t* c2s_offsets1, const int32_t* c2s1,
                         const int32_t* masters_offsets1, const int32_t* masters1,
                         int32_t num_threads);
int64_t mpcx_pattern_nnz(void* pattern);
int32_t mpcx_pattern_nrows(void* pattern);
int mpcx_pattern_copy(void* pattern, mpcx_nnz_t* rowptr, int32_t* cols);
void mpcx_pattern_free(void* pattern);

/* The same pattern built on the DEVICE (all pointers DEVICE).  Protocol:
 *   1. mpcx_pattern_device_adjacency(adj = NULL, counter zeroed [num_blocks0]) counts the cells
 *      under every row block (its cells + the cells whose row slaves have a master in it);
 *   2. the caller scans counter into adj_off (int64 [num_blocks0 + 1]), zeroes counter, allocates
 *      adj (int32 [adj_off[num_blocks0]]) and calls it again to place the cells;
 *   3. mpcx_pattern_device_rows(cols = NULL) writes the number of distinct column blocks of
 *      every row block to row_count; the caller expands it to the scalar rowptr
 *      (row r*bs0+k holds row_count[r]*bs1 entries) and allocates cols;
 *   4. mpcx_pattern_device_rows(cols != NULL) writes the sorted columns.
 * *overflow (zeroed by the caller) is set if a row block has more than 128 distinct column
 * blocks: use the host builder then. */
int mpcx_pattern_device_adjacency(int64_t num_cells, const int32_t* dofmap0, int32_t nd0, int32_t bs0,
                                  const int32_t* c2s_offsets0, const int32_t* c2s0,
                                  const int32_t* masters_offsets0, const int32_t* masters0,
                                  const int64_t* adj_off, int32_t* counter, int32_t* adj, void* stream);
int mpcx_pattern_device_rows(int32_t num_blocks0, const int64_t* adj_off, const int32_t* adj,
                             const int32_t* dofmap1, int32_t nd1, int32_t bs1,
                             const int32_t* c2s_offsets1, const int32_t* c2s1,
                             const int32_t* masters_offsets1, const int32_t* masters1,
                             int32_t* row_count, const mpcx_nnz_t* rowptr, int32_t bs0, int32_t* cols,
                             int32_t* overflow, void* stream);

/* Only the row ranges of a row-block plan (HOST): block_row0[0..nb] into the caller's array of `capacity` entries;
 * returns nb (call with block_row0 == NULL to learn it), -1 on error.  rowptr == NULL: one entry per row (plans of
 * the vector kernels: max_rows rows per block). */
int64_t mpcx_block_ranges(int32_t nrows, const mpcx_nnz_t* rowptr, int32_t max_rows, int32_t max_nnz, int32_t bs,
                          const int32_t* row_hints, int32_t n_hints, int32_t* block_row0, int64_t capacity);

/* Owner-computes plan of the row-block vector kernels (mpcx_vector_args_t::own_*) on the DEVICE (csrc/mpcx_plans.hip).
 * Work item e (an entity, or a cell cluster) has nd dofs mrow[e*nd + i] = dof | flags << 28 and belongs to the block that
 * holds the rows of its dof 0; dofs of other blocks are the block's halo, numbered after its own dofs in ascending dof
 * order.  Steps (scans / sorts / run lengths between them: mpcx_scan_exclusive_*, mpcx_sort_pairs_*, mpcx_run_heads):
 *   1. mpcx_owner_plan_count: owner_key[e] = block (int64: the sort key of the item order), item[e] = e,
 *      fcount[e] = number of its dofs outside the block;
 *   2. foff = exclusive scan of fcount;  (owner_key, item) sorted by key -> item order, block offsets;
 *   3. mpcx_owner_plan_keys: keys[foff[e] + k] = block << 32 | dof, src = e*nd + i of every foreign dof;
 *      lmap[e*nd + i] = (dof - first dof of the block) | flags for the own ones;
 *   4. (keys, src) sorted; heads / heads_scan = mpcx_run_heads + exclusive scan; the DISTINCT keys are the halo
 *      entries, hoff = their offsets per block (mpcx_segment_offsets with shift 32);
 *   5. mpcx_owner_plan_halo: lmap[src] = (own dofs of the block + rank of the key inside the block) | flags;
 *      *max_rows = rows (own + halo) of the largest block;
 *   6. spill order: mpcx_low_word_iota(distinct keys) -> (dof, index) sorted by dof; runs of equal dof are the rows
 *      mpcx_vector_args_t::spill_* reduces. */
int mpcx_owner_plan_count(int64_t n, int32_t nd, const int32_t* mrow, int32_t bs, int32_t nb, const int32_t* block_row0,
                          int64_t* owner_key, int32_t* item, int32_t* fcount, void* stream);
int mpcx_owner_plan_keys(int64_t n, int32_t nd, const int32_t* mrow, int32_t bs, int32_t nb, const int32_t* block_row0,
                         const int64_t* owner_key, const int64_t* foff, int64_t* keys, int32_t* src, int32_t* lmap,
                         void* stream);
int mpcx_owner_plan_halo(int64_t nf, const int64_t* sorted_keys, const int32_t* sorted_src, const int32_t* heads,
                         const int64_t* heads_scan, const int64_t* hoff, int32_t nb, const int32_t* block_row0, int32_t bs,
                         const int32_t* mrow, int32_t* lmap, int32_t* max_rows, void* stream);
int mpcx_low_word_iota(int64_t n, const int64_t* keys, int64_t* low, int32_t* iota, void* stream);
/* The six steps above and everything between them behind ONE call, in device memory the library allocates (for callers
 * without torch).  mrow: DEVICE [n][nd], e.g. mpcx_mask_dofmap(cluster vertices or cell dofmap, ..., bc = NULL, is_slave)
 * -- slave flags in the bits 28 + k; rows_per_block own rows per block (multiples of bs; row_hints: HOST, preferred cuts,
 * or NULL); max_lds_rows > 0: fail with -4 if a block with its halo is larger (96 KiB / 8 = 12288 rows for the kernels here).
 * mpcx_owner_plan_fill sets plan and the own_* fields of *args (MPCX_ALG_ROWBLOCK, or MPCX_ALG_CUBE with cube_verts). */
typedef struct mpcx_owner_plan mpcx_owner_plan_t;
int mpcx_owner_plan_create(int64_t n, int32_t nd, const int32_t* mrow, int32_t bs, int32_t nrows, int32_t rows_per_block,
                           const int32_t* row_hints, int32_t n_hints, int32_t max_lds_rows, void* stream, mpcx_owner_plan_t** plan);
int mpcx_owner_plan_fill(const mpcx_owner_plan_t* plan, mpcx_vector_args_t* args);
void mpcx_owner_plan_destroy(mpcx_owner_plan_t* plan);

/* The per-cell plan of MPCX_ALG_ROWBLOCK behind ONE call, in device memory the library allocates (for callers without
 * torch; the counterpart of mpcx_cluster_plan_create for meshes / forms the cluster kernels do not cover): row ranges
 * (mpcx_block_ranges over rowptr_host), the entities of every row block (mpcx_rowblock_pairs_device -> scan -> stable sort;
 * group_rows != 0: entities of a block ordered by the set of their local rows inside it, nd0 <= 30), the scatter-offset table
 * (mpcx_scatter_offsets) and the masked dofmaps (mpcx_mask_dofmap; bc0 / bc1 may be NULL).  entities: DEVICE [n_entities *
 * estride] or NULL (entity e is cell e); num_cells: rows of the dofmaps.  mpcx_cell_plan_fill sets plan, mdofmap0, mdofmap1,
 * lean = 0 and algorithm of *args; everything else (CSR, kernel, geometry, dofmaps, markers, constraints, slave entities and
 * their optional mpc_plan_*) stays the caller's.  Returns 0; -4: a dof block beyond the block capacity, -21: an offset beyond
 * 8 bits (both: use MPCX_ALG_ATOMIC). */
typedef struct mpcx_cell_plan mpcx_cell_plan_t;
int mpcx_cell_plan_create(int32_t nrows, const mpcx_nnz_t* rowptr, const mpcx_nnz_t* rowptr_host, const int32_t* cols,
                          int64_t n_entities, int32_t estride, const int32_t* entities, int64_t num_cells,
                          const int32_t* dofmap0, int32_t nd0, int32_t bs0, const int8_t* bc0, const int8_t* is_slave0,
                          const int32_t* dofmap1, int32_t nd1, int32_t bs1, const int8_t* bc1, const int8_t* is_slave1,
                          int32_t max_rows, int32_t max_nnz, const int32_t* row_hints, int32_t n_hints, int32_t group_rows,
                          void* stream, mpcx_cell_plan_t** plan);
int mpcx_cell_plan_fill(const mpcx_cell_plan_t* plan, mpcx_matrix_args_t* args);
int64_t mpcx_cell_plan_num_slots(const mpcx_cell_plan_t* plan);
int32_t mpcx_cell_plan_num_blocks(const mpcx_cell_plan_t* plan);
void mpcx_cell_plan_destroy(mpcx_cell_plan_t* plan);

/* The plans of the other workloads behind ONE call each, in device memory the library allocates (round 6; for callers without
 * torch -- examples/mpcx_driver_blocks.cpp assembles BASELINE configs[2] and [4] with them).  All pointers DEVICE unless said
 * otherwise; rowptr_host: the CSR offsets on the HOST; row_hints: HOST, preferred cuts (first row of every numbering tile) or NULL.
 *
 * mpcx_pairs_plan_*: matrix_pairs_kernel (plan.row_pairs == 2: scalar P2 stiffness, elasticity, the Taylor-Hood coupling blocks --
 *   cell integrals of operators with a compact context, no coefficient): row ranges, the (entity, local row) pairs of every block
 *   ordered by (local row, round-robin over the row dofs), ONE record per pair (mpcx_pair_records), the column-masked dofmap and the
 *   per-entity contexts (mpcx_pair_context; x == NULL: none, the kernel then recomputes them per pair).  entities: cells of the
 *   integral or NULL (entity e is cell e), borrowed for the plan's lifetime.  mpcx_pairs_plan_update_geometry recomputes the
 *   contexts after the mesh moved.  _fill sets plan, pair_recs, pair_ctx, mdofmap1, algorithm.  -21: a field of a record overflows
 *   (use mpcx_cell_plan_create), -10: the operator has no compact context.
 * mpcx_nodeblock_plan_*: matrix_nodeblock_kernel (component-diagonal forms on blocked spaces, bs = 2 / 3, test space == trial space:
 *   the Taylor-Hood velocity block): the per-cell plan with max_rows / max_nnz counted per NODE row (the kernel keeps one LDS value
 *   per bs x bs block) + the slot masks (mpcx_diag_slot_mask).  _fill sets plan, mdofmap0 / 1, slot_mask, algorithm; block_vals stays
 *   the caller's choice.  -21: the pattern is not made of whole blocks.
 * mpcx_master_plan_*: the master contributions of the slave entities gathered by target position (mpcx_matrix_args_t::mpc_plan_*),
 *   built on the device: mpcx_mpc_plan_device count -> scan -> fill -> stable sort by position -> run lengths.  diag != 0:
 *   component-diagonal form.  _fill sets the mpc_plan_* fields and mpc_plan_group (slave_entities / n_slave_entities stay the
 *   caller's). */
typedef struct mpcx_pairs_plan mpcx_pairs_plan_t;
int mpcx_pairs_plan_create(int32_t nrows, const mpcx_nnz_t* rowptr, const mpcx_nnz_t* rowptr_host, const int32_t* cols,
                           int64_t n_entities, const int32_t* entities, int64_t num_cells, const int32_t* dofmap0, int32_t nd0,
                           int32_t bs0, const int8_t* bc0, const int8_t* is_slave0, const int32_t* dofmap1, int32_t nd1, int32_t bs1,
                           const int8_t* bc1, const int8_t* is_slave1, int32_t max_rows, int32_t max_nnz, const int32_t* row_hints,
                           int32_t n_hints, const mpcx_kernel_t* kernel, const double* x, const int32_t* x_dofmap, int32_t nv,
                           void* stream, mpcx_pairs_plan_t** plan);
int mpcx_pairs_plan_update_geometry(mpcx_pairs_plan_t* plan, const mpcx_kernel_t* kernel, const double* x, const int32_t* x_dofmap,
                                    int32_t nv, void* stream);
int mpcx_pairs_plan_fill(const mpcx_pairs_plan_t* plan, mpcx_matrix_args_t* args);
int64_t mpcx_pairs_plan_num_pairs(const mpcx_pairs_plan_t* plan);
int32_t mpcx_pairs_plan_num_blocks(const mpcx_pairs_plan_t* plan);
void mpcx_pairs_plan_destroy(mpcx_pairs_plan_t* plan);
typedef struct mpcx_nodeblock_plan mpcx_nodeblock_plan_t;
int mpcx_nodeblock_plan_create(int32_t nrows, const mpcx_nnz_t* rowptr, const mpcx_nnz_t* rowptr_host, const int32_t* cols,
                               int64_t n_entities, int32_t estride, const int32_t* entities, int64_t num_cells, const int32_t* dofmap,
                               int32_t nd, int32_t bs, const int8_t* bc, const int8_t* is_slave, int32_t max_rows, int32_t max_nnz,
                               const int32_t* row_hints, int32_t n_hints, void* stream, mpcx_nodeblock_plan_t** plan);
int mpcx_nodeblock_plan_fill(const mpcx_nodeblock_plan_t* plan, mpcx_matrix_args_t* args);
void mpcx_nodeblock_plan_destroy(mpcx_nodeblock_plan_t* plan);
typedef struct mpcx_master_plan mpcx_master_plan_t;
int mpcx_master_plan_create(int64_t n_slave_entities, const int32_t* slave_entities, int32_t estride, const int32_t* entities0,
                            const int32_t* entities1, const int32_t* dofmap0, int32_t nd0, int32_t bs0, const int32_t* dofmap1,
                            int32_t nd1, int32_t bs1, const int8_t* bc0, const int8_t* bc1, const mpcx_mpc_t* mpc0,
                            const mpcx_mpc_t* mpc1, const mpcx_nnz_t* rowptr, const int32_t* cols, int32_t diag, void* stream,
                            mpcx_master_plan_t** plan);
int mpcx_master_plan_fill(const mpcx_master_plan_t* plan, mpcx_matrix_args_t* args);
int64_t mpcx_master_plan_num_targets(const mpcx_master_plan_t* plan);
int64_t mpcx_master_plan_num_tuples(const mpcx_master_plan_t* plan);
void mpcx_master_plan_destroy(mpcx_master_plan_t* plan);

/* Row-block plan for MPCX_ALG_ROWBLOCK: contiguous row ranges with at most
 * max_rows rows / max_nnz nonzeros, and for each block the entities whose
 * test-space cell has a dof in it.  `row_hints` (sorted row indices, may be
 * NULL) are preferred cut positions, e.g. the first row of each numbering tile.
 * Returns an opaque handle. */
/* n_entities == 0: only the row ranges are computed (the entity lists then come from
 * mpcx_rowblock_pairs_device). */
void* mpcx_rowblock_plan_build(int32_t nrows, const mpcx_nnz_t* rowptr, int32_t max_rows,
                               int32_t max_nnz, int64_t n_entities, int32_t estride,
                               const int32_t* entities0, const int32_t* dofmap0,
                               int32_t nd0, int32_t bs0, const int32_t* row_hints,
                               int32_t n_hints, int32_t num_threads);
int32_t mpcx_rowblock_plan_num_blocks(void* plan);
int64_t mpcx_rowblock_plan_num_ents(void* plan);
int mpcx_rowblock_plan_copy(void* plan, int32_t* block_row0, int64_t* block_ent_off,
                            int32_t* block_ents);
void mpcx_rowblock_plan_free(void* plan);

/* HOST: plan of the master contributions (the index logic of modify_mpc_cell,
 * cpp/assemble_matrix.cpp:182-267, evaluated once), gathered by target position of the CSR values:
 * tuples (entity, element-tensor entry p*N1+q, coefficient).  Dirichlet rows/cols of the element
 * tensor contribute nothing; entities0/1 may be NULL (= identity); bc0/bc1 may be NULL. */
void* mpcx_mpc_plan_build(int64_t n_slave_entities, const int32_t* slave_entities, int32_t estride,
                          const int32_t* entities0, const int32_t* entities1, const int32_t* dofmap0,
                          int32_t nd0, int32_t bs0, const int32_t* dofmap1, int32_t nd1, int32_t bs1,
                          const int8_t* bc0, const int8_t* bc1, const int8_t* is_slave0,
                          const int32_t* masters_offsets0, const int32_t* masters0, const double* coeffs0,
                          const int8_t* is_slave1, const int32_t* masters_offsets1, const int32_t* masters1,
                          const double* coeffs1, const mpcx_nnz_t* rowptr, const int32_t* cols);
int64_t mpcx_mpc_plan_size(void* plan);        /* tuples */
int64_t mpcx_mpc_plan_num_targets(void* plan); /* distinct positions */
int mpcx_mpc_plan_copy(void* plan, mpcx_nnz_t* tgt_pos, int64_t* off, int32_t* ent, int32_t* pq, double* coef);
void mpcx_mpc_plan_free(void* plan);

/* The same plan built on the DEVICE (all pointers DEVICE), two calls:
 *   1. offsets == NULL: counts[t] = number of tuples slave entity t emits;
 *   2. the caller scans counts into offsets (exclusive, int64 [n_slave_entities]), allocates pos / ent / pq /
 *      coef with the total and calls again: tuple k of entity t goes to offsets[t] + k.
 * pos = position in vals (-1: outside the pattern, to be dropped).  The caller then sorts the tuples by
 * pos (stable) and forms tgt / off -- see dolfinx_mpc_amd/assemble_matrix.py.  diag != 0: component-diagonal
 * form on blocked spaces (entries with different components are structural zeros and emit nothing).
 * At most 32 unrolled dofs per element side. */
int mpcx_mpc_plan_device(int64_t n_slave_entities, const int32_t* slave_entities, int32_t estride,
                         const int32_t* entities0, const int32_t* entities1, const int32_t* dofmap0, int32_t nd0,
                         int32_t bs0, const int32_t* dofmap1, int32_t nd1, int32_t bs1, const int8_t* bc0,
                         const int8_t* bc1, const mpcx_mpc_t* mpc0, const mpcx_mpc_t* mpc1, const mpcx_nnz_t* rowptr,
                         const int32_t* cols, int32_t diag, int64_t* counts, const int64_t* offsets, mpcx_nnz_t* pos,
                         int32_t* ent, int32_t* pq, double* coef, void* stream);

/* HOST: dictionary-compress n rows of `noff` bytes (the scatter-offset table copied to the
 * host).  pattern_ids[n] (uint16) and table[max_patterns*noff] are caller-allocated.
 * Returns the number of distinct rows, or -1 if there are more than max_patterns (<= 65536). */
int32_t mpcx_compress_offsets(const uint8_t* rows, int64_t n, int32_t noff, int32_t max_patterns,
                              uint16_t* pattern_ids, uint8_t* table);

/* ------------------------------------------------------------------------
 * The caller of the assembly path on the device: what
 * python/src/dolfinx_mpc/problem.py LinearProblem.solve does with PETSc KSP, for the assembled
 * CSR matrix (symmetric positive definite after the MPC reduction: identity rows for slaves and
 * Dirichlet dofs).  All pointers DEVICE.
 *   mpcx_spmv:             y = A x
 *   mpcx_inverse_diagonal: dinv[r] = 1 / A[r,r] (1 where the diagonal is absent or zero)
 *   mpcx_cg_start:         x = 0, r = b, z = dinv r, p = z; scal[8] (device doubles) holds the
 *                          running scalars: [0,1] r.z, [2,3] p.Ap, [4,5] r.r (ping-pong by
 *                          iteration parity), [6] b.b
 *   mpcx_cg_step(k):       iteration k = 0, 1, ...: Ap = A p (+ p.Ap), x += alpha p,
 *                          r -= alpha Ap, z = dinv r (+ r.z, r.r), p = z + beta p; no host round
 *                          trip; |r|^2 after the step is scal[4 + ((k + 1) & 1)]
 * ---------------------------------------------------------------------- */
int mpcx_spmv(int32_t nrows, const mpcx_nnz_t* rowptr, const int32_t* cols, const double* vals,
              const double* x, double* y, void* stream);
int mpcx_inverse_diagonal(int32_t nrows, const mpcx_nnz_t* rowptr, const int32_t* cols, const double* vals,
                          double* dinv, void* stream);
int mpcx_cg_start(int32_t n, const double* dinv, const double* b, double* x, double* r, double* z,
                  double* p, double* scal, void* stream);
int mpcx_cg_step(int32_t n, const mpcx_nnz_t* rowptr, const int32_t* cols, const double* vals,
                 const double* dinv, double* x, double* r, double* z, double* p, double* Ap,
                 double* scal, int32_t k, void* stream);

/* Block-scalar storage (mpcx_matrix_args_t::block_vals), all pointers DEVICE; rowptr / cols are the scalar CSR pattern
 * (whole bs x bs blocks, as mpcx_diag_slot_mask requires), n_nodes = nrows / bs:
 *   mpcx_block_expand:      vals[entry (k, q) of block s] = (k == q and bit k of slot_mask[s] clear) ? block_vals[s] : 0
 *   mpcx_spmv_blockscalar:  y = (S (x) I, masked) x, the same matrix without materialising it
 *   mpcx_csr_positions:     pos[i] = position of (rows[i], colsq[i]) in the CSR, -1 if absent (overlay set-up)
 *   mpcx_spmv_coo_add:      y[rows[i]] += v[i] * x[colsq[i]]  (the overlay: master contributions, diagonals) */
int mpcx_block_expand(int32_t n_nodes, const mpcx_nnz_t* rowptr, int32_t bs, const double* block_vals,
                      const uint8_t* slot_mask, double* vals, void* stream);
int mpcx_spmv_blockscalar(int32_t n_nodes, const mpcx_nnz_t* rowptr, const int32_t* cols, int32_t bs,
                          const double* block_vals, const uint8_t* slot_mask, const double* x, double* y, void* stream);
int mpcx_csr_positions(const mpcx_nnz_t* rowptr, const int32_t* cols, const int32_t* rows, const int32_t* colsq, int64_t n,
                       int64_t* pos, void* stream);
int mpcx_spmv_coo_add(int64_t n, const int32_t* rows, const int32_t* colsq, const double* v, const double* x, double* y,
                      void* stream);

/* Interface exchange between GPUs (the `A.assemble()` / `ghostUpdate(ADD, REVERSE)` step of the
 * reference, python/src/dolfinx_mpc/assemble_matrix.py:64, python/benchmarks/bench_periodic.py:108):
 * pack out[i] = values[idx[i]] for the send buffer, values[idx[i]] += in[i] for the received partial
 * sums.  All pointers DEVICE, 64-bit indices. */
int mpcx_gather_f64(const double* values, const int64_t* idx, int64_t n, double* out, void* stream);
int mpcx_scatter_add_f64(double* values, const int64_t* idx, int64_t n, const double* in, void* stream);

/* Internal renumbering for locality (dolfinx_mpc_amd/locality.py: a mesh numbered without locality is assembled on a
 * spatially reordered twin and the result is handed back in the caller's numbering; all pointers DEVICE).
 * mpcx_csr_permutation: src[k] (uint32, or int64 when wide != 0: more than 2^32 - 1 entries) = position in the twin's
 * CSR (rowptr2, cols2) of entry k of the caller's CSR (rowptr, cols): caller row r = twin row new_of_old0[r], caller
 * column c = twin column new_of_old1[c]; *bad (zeroed by the caller) is set if an entry has no counterpart.
 * mpcx_permute_values: dst[k] = vals2[src[k]], k < n. */
int mpcx_csr_permutation(int32_t nrows, const mpcx_nnz_t* rowptr, const int32_t* cols, const int32_t* new_of_old0,
                         const int32_t* new_of_old1, const mpcx_nnz_t* rowptr2, const int32_t* cols2, void* src, int32_t wide,
                         int32_t* bad, void* stream);
int mpcx_permute_values(int64_t n, const void* src, int32_t wide, const double* vals2, double* dst, void* stream);
/* out[src[k]] = k: the scatter form of the same permutation (mpcx_matrix_args_t::val_map of the twin's launches) */
int mpcx_invert_permutation(int64_t n, const void* src, int32_t wide, void* out, void* stream);
/* out_map / out_delta of mpcx_matrix_args_t from val_map (all DEVICE): for every row r of the launch's CSR (rowptr) the
 * entries are ranked by their position in the caller's row; *bad is set when a row's entries do not fill one contiguous
 * range of caller positions (not a permutation of whole rows) or a row is longer than 32767 entries. */
int mpcx_write_out_order(int32_t nrows, const mpcx_nnz_t* rowptr, const void* val_map, int32_t wide, void* out_map,
                         int16_t* out_delta, int32_t* bad, void* stream);
/* The twin itself (all pointers DEVICE): mpcx_renumber_mesh: x_out[node_new_of_old[n]] = x[n] (3 coordinates per node),
 * cells_out[c][i] = node_new_of_old[cells[cell_old_of_new[c]][i]], cell_new_of_old[cell_old_of_new[c]] = c.
 * mpcx_dof_permutation: new_of_old[dofmap_old[c][i]] = dofmap_new[cell_new_of_old[c]][i] for every cell c and local dof i
 * (new_of_old preset by the caller, e.g. to -1: a dof no cell refers to keeps the preset). */
int mpcx_renumber_mesh(const double* x, int64_t n_nodes, const int32_t* cells, int64_t n_cells, int32_t nv,
                       const int64_t* node_new_of_old, const int64_t* cell_old_of_new, double* x_out, int32_t* cells_out,
                       int64_t* cell_new_of_old, void* stream);
int mpcx_dof_permutation(const int32_t* dofmap_old, const int32_t* dofmap_new, const int64_t* cell_new_of_old, int64_t n_cells,
                         int32_t nd, int64_t* new_of_old, void* stream);

/* The small kernels of the path for any scalar type (mpcx_kernel_t::scalar_type; pointers DEVICE, values of that type):
 * vals[pos(d, d)] += (re + i im) for d in dofs (cpp/assemble_matrix.cpp:711-724 and dolfinx insert_diagonal),
 * u[s] = sum_j c_j u[m_j] (cpp/MultiPointConstraint.h:129-145, no conjugation), u[s] = 0 (:147-152). */
int mpcx_add_diagonal_scalar(int32_t scalar_type, const mpcx_nnz_t* rowptr, const int32_t* cols, void* vals, const int32_t* dofs,
                             int64_t n, double re, double im, void* stream);
int mpcx_backsubstitution_scalar(int32_t scalar_type, void* u, const int32_t* slaves, int64_t n, const mpcx_mpc_t* mpc, void* stream);
int mpcx_homogenize_scalar(int32_t scalar_type, void* u, const int32_t* slaves, int64_t n, void* stream);

/* HBM bandwidth probe (the denominator bench.py prints next to the 8 TB/s specification): 16 bytes per lane and access,
 * grid-stride over `bytes` of DEVICE memory; mode 0 copy dst = src, 1 read src only, 2 write dst only, 3 copy with four
 * loads in flight per lane, 4 the same with non-temporal loads and stores. */
int mpcx_hbm_probe(const void* src, void* dst, int64_t bytes, int32_t mode, void* stream);

/* misc */
const char* mpcx_last_error(void);
/* The tensor grid under a mesh of axis-aligned box clusters (mpcx_vector_args_t::grid_*), built on the DEVICE in
 * library-owned memory: per axis the distinct intervals (coordinate of corner 0, of corner 7) the clusters sit on -- a cluster
 * list sorted by its low corner, one high corner per low corner --, per cluster its three intervals, the scratch table of the
 * launches, and, when `plan` (the owner-computes plan of the vector call: num_blocks / block_ent_off / block_ents) is given and
 * every block needs at most MPCX_GRID_BLOCK_ROWS rows, the rows per block with the clusters numbered by them.
 * Returns 0 and *out = the plan; 1 and *out = NULL when the mesh has no such grid (a cluster that is not a box with its
 * vertices in corner order, two clusters that start at one coordinate and end at different ones, or more intervals than
 * max(4096, clusters / 8)): the caller then leaves grid_idx NULL; negative: error.  Geometry only -- rebuild when the mesh
 * moves.  cube_verts / x: DEVICE, as mpcx_vector_args_t::cube_verts / x.  The torch twin: assemble_vector._cluster_grid. */
typedef struct mpcx_grid_plan mpcx_grid_plan_t;
int mpcx_grid_plan_create(const int32_t* cube_verts, int64_t n_cubes, const double* x, const mpcx_rowblock_plan_t* plan, void* stream,
                          mpcx_grid_plan_t** out);
int mpcx_grid_plan_fill(const mpcx_grid_plan_t* plan, mpcx_vector_args_t* args); /* grid_idx, grid_iv, grid_tab, grid_n, grid_block_rows(_max) */
int32_t mpcx_grid_plan_num_intervals(const mpcx_grid_plan_t* plan, int32_t axis);
int32_t mpcx_grid_plan_block_rows(const mpcx_grid_plan_t* plan); /* longest block list; 0: the clusters read the table itself */
void mpcx_grid_plan_destroy(mpcx_grid_plan_t* plan);

/* The per-cell twin (mpcx_vector_args_t::grid_eta / grid_J / grid_ntypes, with grid_idx, grid_iv, grid_tab, grid_n and
 * grid_block_rows) for the owner-computes launch over ALL cells of a tetrahedral mesh: cells [n_cells][4] vertex ids and x DEVICE,
 * plan = the owner-computes plan of the vector call (its lists hold cells), qpts_host HOST [nq][3] the rule of the form.
 * Returns 0 and the plan; 1 and *out = NULL when the mesh is not of that kind (a cell with a vertex between the ends of its
 * interval or a flat cell, no tensor grid, a block that needs more than MPCX_GRID_BLOCK_ROWS rows, a rule with more than 255 sums);
 * negative: error.  The torch twin: assemble_vector._cell_grid. */
typedef struct mpcx_cell_grid_plan mpcx_cell_grid_plan_t;
int mpcx_cell_grid_plan_create(const int32_t* cells, int64_t n_cells, const double* x, const mpcx_rowblock_plan_t* plan,
                               const double* qpts_host, int32_t nq, void* stream, mpcx_cell_grid_plan_t** out);
int mpcx_cell_grid_plan_fill(const mpcx_cell_grid_plan_t* plan, mpcx_vector_args_t* args);
void mpcx_cell_grid_plan_destroy(mpcx_cell_grid_plan_t* plan);

int mpcx_version(void);
/* Load the library's code objects now (one empty kernel per translation unit on `stream`, then a stream synchronisation)
 * instead of at the first assembly call; optional, idempotent, thread-safe. */
int mpcx_preload(void* stream);
int mpcx_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* MPCX_H */
