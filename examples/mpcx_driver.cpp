// mpcx_driver -- a consumer of the C ABI (include/mpcx.h) that is neither Python nor torch: what a C++ binding inside
// dolfinx_mpc's python/src/dolfinx_mpc/mpc.cpp (the nanobind functions cpp.mpc.assemble_matrix / assemble_vector /
// apply_lifting, :273-318) would do with a Form, a MultiPointConstraint and a list of Dirichlet conditions.  It reads one
// problem file (P1 tetrahedra: coordinates, cells, the add_constraint arrays, Dirichlet markers and values, the two
// kernel descriptors with their quadrature tables), runs the WHOLE constrained assembly through libmpcx.so with memory
// from hipMalloc --
//     mpcx_mpc_finalize -> mpcx_cell_to_slaves -> mpcx_pattern_build                       (host set-up)
//     mpcx_cluster_plan_create (+ mpcx_cell_plan_create for cells outside the clusters) -> mpcx_assemble_matrix -> mpcx_add_diagonal
//     mpcx_mask_dofmap -> mpcx_owner_plan_create (-> mpcx_grid_plan_create) -> mpcx_assemble_vector (+ leftover cells)
//     mpcx_apply_lifting, set_bc
// -- and writes the CSR matrix and the vector.  tests/test_gpu_driver.py compares them with the Python host layer's
// result on the same problem (same kernels underneath, so to rounding of the summation order).
//
//     mpcx_driver problem.bin result.bin [steps]
//
// File format (both files): "MPCX1\0\0\0", int64 count, then per array: char name[32], int32 dtype (0 int8, 1 int32,
// 2 int64, 3 float64), int64 n, the data.
#include "mpcx_bundle.hpp"

int main(int argc, char** argv)
{
  if (argc < 3)
  {
    std::fprintf(stderr, "usage: %s problem.bin result.bin [steps]\n", argv[0]);
    return 2;
  }
  const int steps = argc > 3 ? std::max(1, std::atoi(argv[3])) : 1;
  try
  {
    if (mpcx_device_count() < 1)
      throw std::runtime_error("no HIP device");
    const Bundle in = read_bundle(argv[1]);
    DeviceArena dev;
    // set-up runs on the null stream; a step issues the matrix call and the vector + lifting calls on two streams (the
    // matrix one at high priority: its short memory-bound launches take the slots the long vector kernel frees)
    hipStream_t stream = nullptr, s_mat = nullptr, s_vec = nullptr;
    int prio_lo = 0, prio_hi = 0;
    hip_check(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi), "hipDeviceGetStreamPriorityRange");
    hip_check(hipStreamCreateWithPriority(&s_mat, hipStreamNonBlocking, prio_hi), "hipStreamCreateWithPriority");
    hip_check(hipStreamCreateWithPriority(&s_vec, hipStreamNonBlocking, prio_lo), "hipStreamCreateWithPriority");
    const Array &X = need(in, "x"), &CELLS = need(in, "cells");
    const int64_t n_nodes = X.n / 3, n_cells = CELLS.n / 4;
    const int32_t ndofs = int32_t(n_nodes); // scalar P1: dofs are the mesh nodes
    const int32_t* cells = CELLS.as<int32_t>();
    const Array& P = need(in, "params"); // max_rows, max_nnz of the matrix row blocks, own rows per vector block
    const int32_t max_rows = P.as<int32_t>()[0], max_nnz = P.as<int32_t>()[1], vrows = P.as<int32_t>()[2];
    const int32_t* hints = nullptr;
    int32_t n_hints = 0;
    if (auto it = in.find("hints"); it != in.end() && it->second.n > 0)
      hints = it->second.as<int32_t>(), n_hints = int32_t(it->second.n);

    // ---- MultiPointConstraint::finalize (cpp/MultiPointConstraint.h:36-126) and the cell -> slaves map
    auto t0 = std::chrono::steady_clock::now();
    const Array &SL = need(in, "slaves"), &MA = need(in, "masters"), &CO = need(in, "coeffs"), &OW = need(in, "owners"),
                &OF = need(in, "offsets");
    const int32_t n_slaves = int32_t(SL.n);
    std::vector<int8_t> is_slave(static_cast<size_t>(ndofs), 0);
    const size_t nm = static_cast<size_t>(std::max<int64_t>(MA.n, 1));
    std::vector<int32_t> sorted_slaves(static_cast<size_t>(std::max(n_slaves, 1))), m_off(static_cast<size_t>(ndofs) + 1), m_idx(nm), m_own(nm);
    std::vector<double> m_coef(nm);
    int32_t n_local_slaves = 0;
    mpcx_check(mpcx_mpc_finalize(ndofs, ndofs, n_slaves, SL.as<int32_t>(), MA.as<int64_t>(), CO.as<double>(), OW.as<int32_t>(),
                                 OF.as<int32_t>(), is_slave.data(), sorted_slaves.data(), &n_local_slaves, m_off.data(), m_idx.data(),
                                 m_coef.data(), m_own.data()),
               "mpcx_mpc_finalize");
    std::vector<int32_t> c2s_off(static_cast<size_t>(n_cells) + 1);
    const int64_t n_links = mpcx_cell_to_slaves(n_cells, 4, 1, cells, is_slave.data(), c2s_off.data(), nullptr);
    if (n_links < 0)
      throw std::runtime_error("mpcx_cell_to_slaves failed");
    std::vector<int32_t> c2s(static_cast<size_t>(std::max<int64_t>(n_links, 1)));
    if (mpcx_cell_to_slaves(n_cells, 4, 1, cells, is_slave.data(), c2s_off.data(), c2s.data()) != n_links)
      throw std::runtime_error("mpcx_cell_to_slaves failed");
    std::vector<int32_t> slave_cells;
    for (int64_t c = 0; c < n_cells; ++c)
      if (c2s_off[c + 1] > c2s_off[c])
        slave_cells.push_back(int32_t(c));
    // ---- sparsity pattern (cpp/utils.h:381-496)
    void* pat = mpcx_pattern_build(n_cells, cells, 4, 1, ndofs, cells, 4, 1, ndofs, c2s_off.data(), c2s.data(), m_off.data(), m_idx.data(),
                                   c2s_off.data(), c2s.data(), m_off.data(), m_idx.data(), 8);
    if (!pat)
      throw std::runtime_error(std::string("mpcx_pattern_build: ") + mpcx_last_error());
    const int64_t nnz = mpcx_pattern_nnz(pat);
    std::vector<mpcx_nnz_t> rowptr(static_cast<size_t>(ndofs) + 1);
    std::vector<int32_t> cols(static_cast<size_t>(nnz));
    mpcx_check(mpcx_pattern_copy(pat, rowptr.data(), cols.data()), "mpcx_pattern_copy");
    mpcx_pattern_free(pat);
    const double t_host = seconds_since(t0);

    // ---- device mirrors
    t0 = std::chrono::steady_clock::now();
    const double* d_x = dev.upload(X.as<double>(), size_t(X.n));
    const int32_t* d_cells = dev.upload(cells, size_t(CELLS.n));
    const mpcx_nnz_t* d_rowptr = dev.upload(rowptr);
    const int32_t* d_cols = dev.upload(cols);
    double* d_vals = dev.alloc<double>(size_t(nnz));
    double* d_b = dev.alloc<double>(size_t(ndofs));
    const Array &BCM = need(in, "bc_markers"), &BCV = need(in, "bc_values");
    const int8_t* d_bc = dev.upload(BCM.as<int8_t>(), size_t(BCM.n));
    const double* d_bcv = dev.upload(BCV.as<double>(), size_t(BCV.n));
    mpcx_mpc_t mpc;
    mpc.is_slave = dev.upload(is_slave);
    mpc.masters_offsets = dev.upload(m_off);
    mpc.masters = dev.upload(m_idx);
    mpc.coeffs = dev.upload(m_coef);
    const int32_t* d_slave_cells = dev.upload(slave_cells);
    const int32_t* d_slaves = dev.upload(sorted_slaves);
    std::vector<int32_t> bc_dofs, lift_cells;
    for (int32_t d = 0; d < ndofs; ++d)
      if (BCM.as<int8_t>()[d])
        bc_dofs.push_back(d);
    for (int64_t c = 0; c < n_cells; ++c) // the has_bc test of cpp/lifting.h:93-109, hoisted
      for (int i = 0; i < 4; ++i)
        if (BCM.as<int8_t>()[cells[4 * c + i]])
        {
          lift_cells.push_back(int32_t(c));
          break;
        }
    const int32_t* d_bc_dofs = dev.upload(bc_dofs);
    const int32_t* d_lift = dev.upload(lift_cells);
    const double *mat_constants = nullptr, *vec_constants = nullptr;
    const mpcx_kernel_t Kmat = make_kernel(in, "mat", dev, &mat_constants);
    const mpcx_kernel_t Kvec = make_kernel(in, "vec", dev, &vec_constants);
    hip_check(hipDeviceSynchronize(), "uploads");
    const double t_upload = seconds_since(t0);

    // ---- plans: cell clusters for the matrix, owner-computes row blocks over the clusters for the vector
    t0 = std::chrono::steady_clock::now();
    mpcx_cluster_plan_t* cplan = nullptr;
    mpcx_check(mpcx_cluster_plan_create(n_cells, d_cells, n_nodes, d_x, ndofs, d_rowptr, rowptr.data(), d_cols, d_bc, mpc.is_slave, max_rows,
                                        max_nnz, hints, n_hints, stream, &cplan),
               "mpcx_cluster_plan_create");
    const int64_t n_clusters = mpcx_cluster_plan_num_clusters(cplan);
    const int32_t* d_left = nullptr;
    const int64_t n_left = mpcx_cluster_plan_leftover(cplan, &d_left);
    // slave cells the cluster call answers for: all but the leftover ones
    std::vector<int32_t> left(static_cast<size_t>(n_left));
    if (n_left)
      hip_check(hipMemcpy(left.data(), d_left, size_t(n_left) * 4, hipMemcpyDeviceToHost), "hipMemcpy D2H");
    std::vector<int8_t> is_left(static_cast<size_t>(n_cells), 0);
    for (int32_t c : left)
      is_left[size_t(c)] = 1;
    std::vector<int32_t> slave_cells_cluster, slave_cells_left;
    for (int32_t c : slave_cells)
      (is_left[size_t(c)] ? slave_cells_left : slave_cells_cluster).push_back(c);
    const int32_t* d_slave_cells_cluster = dev.upload(slave_cells_cluster);
    // cells the cluster kernels do not cover (all cells of a mesh without six-tet fans): the per-cell LDS row-block plan
    mpcx_cell_plan_t* cellplan = nullptr;
    const int64_t n_percell = n_clusters > 0 ? n_left : n_cells;
    if (n_percell > 0)
      mpcx_check(mpcx_cell_plan_create(ndofs, d_rowptr, rowptr.data(), d_cols, n_percell, 1, n_clusters > 0 ? d_left : nullptr, n_cells, d_cells, 4,
                                       1, d_bc, mpc.is_slave, d_cells, 4, 1, d_bc, mpc.is_slave, max_rows, max_nnz, hints, n_hints, 1, stream,
                                       &cellplan),
                 "mpcx_cell_plan_create");
    mpcx_owner_plan_t* oplan = nullptr;
    mpcx_owner_plan_t* oplan_cells = nullptr; // no clusters at all: owner-computes row blocks over the cells
    if (n_clusters == 0)
    {
      int32_t* mrow = dev.alloc<int32_t>(size_t(n_cells) * 4);
      mpcx_check(mpcx_mask_dofmap(d_cells, n_cells, 4, 1, nullptr, mpc.is_slave, 0, mrow, stream), "mpcx_mask_dofmap");
      mpcx_check(mpcx_owner_plan_create(n_cells, 4, mrow, 1, ndofs, vrows, hints, n_hints, 12288, stream, &oplan_cells), "mpcx_owner_plan_create");
    }
    if (n_clusters > 0)
    {
      int32_t* mrow = dev.alloc<int32_t>(size_t(n_clusters) * 8);
      mpcx_check(mpcx_mask_dofmap(mpcx_cluster_plan_verts(cplan), n_clusters, 8, 1, nullptr, mpc.is_slave, 0, mrow, stream), "mpcx_mask_dofmap");
      mpcx_check(mpcx_owner_plan_create(n_clusters, 8, mrow, 1, ndofs, vrows, hints, n_hints, 12288, stream, &oplan), "mpcx_owner_plan_create");
    }
    // the benchmark's right-hand side (kernel.fn_id 1, 14-point rule) on a mesh of box clusters: the tensor grid under the
    // clusters, so that every launch evaluates the univariate factors once per interval (mpcx_vector_args_t::grid_*);
    // return code 1 = the mesh has no such grid (the clusters are then evaluated one by one)
    mpcx_grid_plan_t* gplan = nullptr;
    if (oplan && Kvec.fn_id == 1 && Kvec.nq == 14 && Kvec.coeff_degree == 0 && !std::getenv("MPCX_DRIVER_NO_GRID"))
    {
      mpcx_vector_args_t tmp;
      std::memset(&tmp, 0, sizeof(tmp));
      mpcx_check(mpcx_owner_plan_fill(oplan, &tmp), "mpcx_owner_plan_fill");
      const int rc = mpcx_grid_plan_create(mpcx_cluster_plan_verts(cplan), n_clusters, d_x, &tmp.plan, stream, &gplan);
      if (rc < 0)
        mpcx_check(rc, "mpcx_grid_plan_create");
    }
    // master contributions of the slave cells gathered by target position (mpcx_matrix_args_t::mpc_plan_*): the host
    // builder is enough for a thin slave layer; without it the kernel searches the CSR rows itself (device atomics)
    const mpcx_nnz_t* d_plan_tgt = nullptr;
    const int64_t* d_plan_off = nullptr;
    const int32_t *d_plan_ent = nullptr, *d_plan_pq = nullptr;
    const double* d_plan_coef = nullptr;
    int64_t plan_targets = 0, plan_tuples = 0;
    // (entity = cell for the calls that carry them: the last cluster launch, or the per-cell launch of a mesh without clusters)
    if (!slave_cells.empty())
    {
      void* mp = mpcx_mpc_plan_build(int64_t(slave_cells.size()), slave_cells.data(), 1, nullptr, nullptr, cells, 4, 1, cells, 4, 1,
                                     BCM.as<int8_t>(), BCM.as<int8_t>(), is_slave.data(), m_off.data(), m_idx.data(), m_coef.data(),
                                     is_slave.data(), m_off.data(), m_idx.data(), m_coef.data(), rowptr.data(), cols.data());
      if (!mp)
        throw std::runtime_error(std::string("mpcx_mpc_plan_build: ") + mpcx_last_error());
      plan_tuples = mpcx_mpc_plan_size(mp), plan_targets = mpcx_mpc_plan_num_targets(mp);
      if (plan_targets > 0)
      {
        std::vector<mpcx_nnz_t> tgt(static_cast<size_t>(plan_targets));
        std::vector<int64_t> off(static_cast<size_t>(plan_targets) + 1);
        std::vector<int32_t> ent(static_cast<size_t>(plan_tuples)), pq(static_cast<size_t>(plan_tuples));
        std::vector<double> coef(static_cast<size_t>(plan_tuples));
        mpcx_check(mpcx_mpc_plan_copy(mp, tgt.data(), off.data(), ent.data(), pq.data(), coef.data()), "mpcx_mpc_plan_copy");
        d_plan_tgt = dev.upload(tgt), d_plan_off = dev.upload(off), d_plan_ent = dev.upload(ent), d_plan_pq = dev.upload(pq);
        d_plan_coef = dev.upload(coef);
      }
      mpcx_mpc_plan_free(mp);
    }
    hip_check(hipDeviceSynchronize(), "plans");
    const double t_plans = seconds_since(t0);

    // ---- the three calls of the hot path, `steps` times (the last result is written)
    auto matrix_base = [&]()
    {
      mpcx_matrix_args_t a;
      std::memset(&a, 0, sizeof(a));
      a.nrows = ndofs, a.rowptr = d_rowptr, a.cols = d_cols, a.vals = d_vals;
      a.kernel = Kmat;
      a.x = d_x, a.x_dofmap = d_cells, a.nv = 4;
      a.estride = 1, a.n_entities = n_cells;
      a.constants = mat_constants;
      a.dofmap0 = a.dofmap1 = d_cells, a.nd0 = a.nd1 = 4, a.bs0 = a.bs1 = 1;
      a.bc0 = a.bc1 = d_bc;
      a.mpc0 = a.mpc1 = mpc;
      a.stream = s_mat;
      return a;
    };
    auto vector_base = [&]()
    {
      mpcx_vector_args_t v;
      std::memset(&v, 0, sizeof(v));
      v.b = d_b, v.num_dofs = ndofs;
      v.kernel = Kvec;
      v.x = d_x, v.x_dofmap = d_cells, v.nv = 4;
      v.estride = 1, v.n_entities = n_cells;
      v.constants = vec_constants;
      v.dofmap = d_cells, v.nd = 4, v.bs = 1;
      v.mpc = mpc;
      v.stream = s_vec;
      return v;
    };
    // the master contributions of ALL slave cells ride on one call whose entities are the cells themselves (entities ==
    // NULL): slave_entities are entity indices (cpp/assemble_matrix.cpp:504-546 does them inside the cell loop)
    auto with_master_contributions = [&](mpcx_matrix_args_t& a)
    {
      a.slave_entities = d_slave_cells, a.n_slave_entities = int64_t(slave_cells.size());
      if (plan_targets > 0)
      {
        a.mpc_plan_targets = plan_targets, a.mpc_plan_tgt = d_plan_tgt, a.mpc_plan_off = d_plan_off, a.mpc_plan_ent = d_plan_ent;
        a.mpc_plan_pq = d_plan_pq, a.mpc_plan_coef = d_plan_coef;
        const double mean = double(plan_tuples) / double(plan_targets);
        a.mpc_plan_group = mean > 10 ? 16 : (mean > 2.5 ? 4 : 1);
      }
    };
    double t_steps = 0.0;
    // (one untimed pass first when several are asked for: code objects are loaded at the first launch of a kernel)
    for (int step = (steps > 1 ? -1 : 0); step < steps; ++step)
    {
      hip_check(hipDeviceSynchronize(), "sync");
      t0 = std::chrono::steady_clock::now();
      // assemble_matrix (python/src/dolfinx_mpc/assemble_matrix.py:43-65): zero, cells, slave + Dirichlet diagonals
      const int32_t n_parts = mpcx_cluster_plan_num_parts(cplan);
      if (n_parts == 0) // (with clusters every row block is WRITTEN by its launch, store_mode 1: no zeroing pass)
        hip_check(hipMemsetAsync(d_vals, 0, size_t(nnz) * 8, s_mat), "hipMemsetAsync");
      for (int32_t p = 0; p < n_parts; ++p)
      {
        mpcx_matrix_args_t a = matrix_base();
        mpcx_check(mpcx_cluster_plan_part(cplan, p, &a), "mpcx_cluster_plan_part");
        a.store_mode = 1;
        // the master contributions of the slave cells ride on the last launch (they add to rows the launches write)
        if (p == n_parts - 1)
          with_master_contributions(a);
        mpcx_check(mpcx_assemble_matrix(&a), "mpcx_assemble_matrix (clusters)");
      }
      if (n_left > 0 || n_parts == 0)
      {
        // cells in no cluster (or a mesh without clusters): per-cell LDS row blocks, added to what the launches above wrote
        mpcx_matrix_args_t a = matrix_base();
        mpcx_check(mpcx_cell_plan_fill(cellplan, &a), "mpcx_cell_plan_fill");
        if (n_parts > 0)
          a.entities = a.entities0 = a.entities1 = d_left, a.n_entities = n_left; // (their slave cells went with the cluster call)
        else
          with_master_contributions(a);
        mpcx_check(mpcx_assemble_matrix(&a), "mpcx_assemble_matrix (per cell)");
      }
      mpcx_check(mpcx_add_diagonal(ndofs, d_rowptr, d_cols, d_vals, d_slaves, n_local_slaves, 1.0, s_mat), "mpcx_add_diagonal (slaves)");
      mpcx_check(mpcx_add_diagonal(ndofs, d_rowptr, d_cols, d_vals, d_bc_dofs, int64_t(bc_dofs.size()), 1.0, s_mat),
                 "mpcx_add_diagonal (Dirichlet)");
      // assemble_vector (assemble_vector.py:79-104): zero, cells
      hip_check(hipMemsetAsync(d_b, 0, size_t(ndofs) * 8, s_vec), "hipMemsetAsync");
      if (oplan)
      {
        mpcx_vector_args_t v = vector_base();
        mpcx_check(mpcx_owner_plan_fill(oplan, &v), "mpcx_owner_plan_fill");
        v.algorithm = MPCX_ALG_CUBE;
        v.cube_verts = mpcx_cluster_plan_verts(cplan), v.n_cubes = n_clusters;
        v.slave_entities = d_slave_cells_cluster, v.n_slave_entities = int64_t(slave_cells_cluster.size());
        v.cube_boxes = 1; // (box clusters factor by factor; clusters that are no boxes point by point)
        if (gplan)
          mpcx_check(mpcx_grid_plan_fill(gplan, &v), "mpcx_grid_plan_fill");
        mpcx_check(mpcx_assemble_vector(&v), "mpcx_assemble_vector (clusters)");
      }
      if (n_left > 0 || !oplan)
      {
        mpcx_vector_args_t v = vector_base();
        if (oplan_cells)
        {
          mpcx_check(mpcx_owner_plan_fill(oplan_cells, &v), "mpcx_owner_plan_fill");
          v.algorithm = MPCX_ALG_ROWBLOCK;
          v.slave_entities = d_slave_cells, v.n_slave_entities = int64_t(slave_cells.size());
        }
        else
        {
          v.algorithm = MPCX_ALG_ATOMIC; // the few cells outside the clusters: LDS hash + device atomics, no plan
          v.entities = v.entities0 = d_left, v.n_entities = n_left;
        }
        mpcx_check(mpcx_assemble_vector(&v), "mpcx_assemble_vector (per cell)");
      }
      // apply_lifting (assemble_vector.py:25-76), scale 1, x0 empty
      if (!lift_cells.empty())
      {
        mpcx_lifting_args_t l;
        std::memset(&l, 0, sizeof(l));
        l.b = d_b, l.num_dofs = ndofs;
        l.kernel = Kmat;
        l.x = d_x, l.x_dofmap = d_cells, l.nv = 4;
        l.estride = 1, l.n_entities = n_cells;
        l.constants = mat_constants;
        l.dofmap0 = l.dofmap1 = d_cells, l.nd0 = l.nd1 = 4, l.bs0 = l.bs1 = 1;
        l.bc_markers1 = d_bc, l.bc_values1 = d_bcv;
        l.scale = 1.0;
        l.lift_entities = d_lift, l.n_lift_entities = int64_t(lift_cells.size());
        l.mpc0 = mpc;
        l.stream = s_vec;
        mpcx_check(mpcx_apply_lifting(&l), "mpcx_apply_lifting");
      }
      hip_check(hipDeviceSynchronize(), "step");
      if (step >= 0)
        t_steps += seconds_since(t0);
    }
    // ---- results; set_bc on the host copy (dolfinx set_bc, bench_periodic.py:109)
    std::vector<double> vals(static_cast<size_t>(nnz)), b(static_cast<size_t>(ndofs));
    hip_check(hipMemcpy(vals.data(), d_vals, size_t(nnz) * 8, hipMemcpyDeviceToHost), "hipMemcpy D2H");
    hip_check(hipMemcpy(b.data(), d_b, size_t(ndofs) * 8, hipMemcpyDeviceToHost), "hipMemcpy D2H");
    for (int32_t d : bc_dofs)
      b[size_t(d)] = BCV.as<double>()[d];
    const std::vector<double> timings = {t_host, t_upload, t_plans, t_steps / steps, double(n_clusters), double(n_left),
                                         double(mpcx_cluster_plan_num_parts(cplan))};
    write_bundle(argv[2], {{"rowptr", make_array(2, rowptr)}, {"cols", make_array(1, cols)}, {"vals", make_array(3, vals)},
                           {"b", make_array(3, b)}, {"timings", make_array(3, timings)}});
    std::printf("mpcx_driver: %lld dofs, %lld cells, %lld entries, %d slaves; %lld clusters + %lld cells; host set-up %.3f s, uploads %.3f s, "
                "plans %.3f s, step %.3f ms\n",
                (long long)ndofs, (long long)n_cells, (long long)nnz, n_slaves, (long long)n_clusters, (long long)n_left, t_host, t_upload,
                t_plans, 1e3 * t_steps / steps);
    if (gplan)
      mpcx_grid_plan_destroy(gplan);
    if (oplan)
      mpcx_owner_plan_destroy(oplan);
    if (oplan_cells)
      mpcx_owner_plan_destroy(oplan_cells);
    if (cellplan)
      mpcx_cell_plan_destroy(cellplan);
    mpcx_cluster_plan_destroy(cplan);
    return 0;
  }
  catch (const std::exception& e)
  {
    std::fprintf(stderr, "mpcx_driver: %s\n", e.what());
    return 1;
  }
}
