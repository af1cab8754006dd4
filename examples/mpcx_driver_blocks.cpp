// mpcx_driver_blocks -- the second consumer of the C ABI (include/mpcx.h) that is neither Python nor torch: any list of
// bilinear blocks a_ij over one or two Lagrange spaces on a tetrahedral mesh (square or rectangular, each space with its own
// constraint and Dirichlet markers) and of linear forms, assembled the way dolfinx_mpc.assemble_matrix_nest /
// assemble_vector do (python/src/dolfinx_mpc/assemble_matrix.py:120-146: block (i, j) with (constraints[i], constraints[j]);
// cpp/assemble_matrix.h:28-43 is ONE function for every form) -- BASELINE configs[4] (scalar P2 Poisson, periodic) and the
// Taylor-Hood blocks of configs[2] (P2^3 x P2^3 stiffness, p div(v), div(u) q with a slip constraint) in particular.
// Where examples/mpcx_driver.cpp exercises the cluster plans of config 2, this one exercises the plans of the other
// workloads, each behind one library call with library-owned device memory:
//     mpcx_mpc_finalize -> mpcx_cell_to_slaves -> mpcx_pattern_build                        (host set-up, per space / block)
//     mpcx_pairs_plan_create | mpcx_nodeblock_plan_create | mpcx_cell_plan_create           (row-block plan by block kind)
//     mpcx_master_plan_create                                                               (master contributions, on the device)
//     mpcx_assemble_matrix (+ mpcx_add_diagonal for square blocks)
//     mpcx_mask_dofmap -> mpcx_owner_plan_create (-> mpcx_cell_grid_plan_create) -> mpcx_assemble_vector
// and writes every block's CSR and every vector.  tests/test_gpu_driver.py compares them with the oracle.
//
//     mpcx_driver_blocks problem.bin result.bin [steps]
//
// Problem file (MPCX1 bundle, examples/mpcx_bundle.hpp):
//   x f64[n_nodes*3], cells i32[n_cells*4]
//   per space k = 0, 1, ...: s<k>_shape i32[3] = (nd, bs, unrolled dofs), s<k>_dofmap i32[n_cells*nd], the add_constraint arrays
//       s<k>_slaves i32 / _masters i64 / _coeffs f64 / _owners i32 / _offsets i32, s<k>_bc_markers i8[dofs], s<k>_bc_values f64[dofs]
//   per block k: b<k>_spaces i32[3] = (row space, column space, plan kind: 0 per-cell row blocks, 1 pair records, 2 node blocks),
//       b<k>_params i32[2] = (max rows, max entries of a row block), b<k>_kernel / _qpts / _qwts / _constants
//   per vector k: v<k>_space i32[2] = (space, own rows per block), v<k>_kernel / _qpts / _qwts / _constants, optional v<k>_qphi,
//       v<k>_vphi (mpcx_kernel_t::qphi / vphi)
#include "mpcx_bundle.hpp"

#include <memory>

namespace
{
struct Space
{
  int32_t nd = 0, bs = 1, ndofs = 0;
  const int32_t* dofmap_h = nullptr;
  const int32_t* d_dofmap = nullptr;
  std::vector<int8_t> is_slave;
  std::vector<int32_t> slaves, m_off, m_idx, m_own, c2s_off, c2s, slave_cells, bc_dofs;
  std::vector<double> m_coef;
  int32_t n_local_slaves = 0;
  const int8_t* bc_h = nullptr;
  const int8_t* d_bc = nullptr;
  const int32_t* d_slaves = nullptr;
  const int32_t* d_bc_dofs = nullptr;
  const int32_t* d_slave_cells = nullptr;
  mpcx_mpc_t mpc{};
};

Space make_space(const Bundle& in, int k, int64_t n_cells, DeviceArena& dev)
{
  const std::string p = "s" + std::to_string(k);
  Space S;
  const int32_t* shape = need(in, (p + "_shape").c_str()).as<int32_t>();
  S.nd = shape[0], S.bs = shape[1], S.ndofs = shape[2];
  const Array& DM = need(in, (p + "_dofmap").c_str());
  if (DM.n != n_cells * S.nd)
    throw std::runtime_error(p + "_dofmap: wrong size");
  S.dofmap_h = DM.as<int32_t>();
  S.d_dofmap = dev.upload(S.dofmap_h, size_t(DM.n));
  // MultiPointConstraint::finalize (cpp/MultiPointConstraint.h:36-126)
  const Array &SL = need(in, (p + "_slaves").c_str()), &MA = need(in, (p + "_masters").c_str()), &CO = need(in, (p + "_coeffs").c_str()),
              &OW = need(in, (p + "_owners").c_str()), &OF = need(in, (p + "_offsets").c_str());
  const int32_t n_slaves = int32_t(SL.n);
  const size_t nm = size_t(std::max<int64_t>(MA.n, 1));
  S.is_slave.assign(size_t(S.ndofs), 0);
  S.slaves.resize(size_t(std::max(n_slaves, 1)));
  S.m_off.resize(size_t(S.ndofs) + 1);
  S.m_idx.resize(nm), S.m_own.resize(nm), S.m_coef.resize(nm);
  static const int32_t zero_off[1] = {0};
  mpcx_check(mpcx_mpc_finalize(S.ndofs, S.ndofs, n_slaves, SL.as<int32_t>(), MA.as<int64_t>(), CO.as<double>(), OW.as<int32_t>(),
                               n_slaves ? OF.as<int32_t>() : zero_off, S.is_slave.data(), S.slaves.data(), &S.n_local_slaves, S.m_off.data(),
                               S.m_idx.data(), S.m_coef.data(), S.m_own.data()),
             "mpcx_mpc_finalize");
  // cell -> slaves (cpp/mpc_helpers.h:19-94)
  S.c2s_off.resize(size_t(n_cells) + 1);
  const int64_t n_links = mpcx_cell_to_slaves(n_cells, S.nd, S.bs, S.dofmap_h, S.is_slave.data(), S.c2s_off.data(), nullptr);
  if (n_links < 0)
    throw std::runtime_error("mpcx_cell_to_slaves failed");
  S.c2s.resize(size_t(std::max<int64_t>(n_links, 1)));
  if (mpcx_cell_to_slaves(n_cells, S.nd, S.bs, S.dofmap_h, S.is_slave.data(), S.c2s_off.data(), S.c2s.data()) != n_links)
    throw std::runtime_error("mpcx_cell_to_slaves failed");
  for (int64_t c = 0; c < n_cells; ++c)
    if (S.c2s_off[size_t(c) + 1] > S.c2s_off[size_t(c)])
      S.slave_cells.push_back(int32_t(c));
  const Array& BCM = need(in, (p + "_bc_markers").c_str());
  S.bc_h = BCM.as<int8_t>();
  S.d_bc = dev.upload(S.bc_h, size_t(BCM.n));
  for (int32_t d = 0; d < S.ndofs; ++d)
    if (S.bc_h[d])
      S.bc_dofs.push_back(d);
  S.d_bc_dofs = dev.upload(S.bc_dofs);
  S.d_slaves = dev.upload(S.slaves);
  S.d_slave_cells = dev.upload(S.slave_cells);
  S.mpc.is_slave = dev.upload(S.is_slave);
  S.mpc.masters_offsets = dev.upload(S.m_off);
  S.mpc.masters = dev.upload(S.m_idx);
  S.mpc.coeffs = dev.upload(S.m_coef);
  return S;
}

template <class T, void (*Destroy)(T*)>
struct Handle
{
  T* p = nullptr;
  Handle() = default;
  Handle(const Handle&) = delete;
  Handle& operator=(const Handle&) = delete;
  ~Handle()
  {
    if (p)
      Destroy(p);
  }
};

struct Block
{
  int row = 0, col = 0, kind = 0;
  mpcx_kernel_t K{};
  const double* constants = nullptr;
  std::vector<mpcx_nnz_t> rowptr;
  std::vector<int32_t> cols;
  const mpcx_nnz_t* d_rowptr = nullptr;
  const int32_t* d_cols = nullptr;
  double* d_vals = nullptr;
  int64_t nnz = 0;
  std::vector<int32_t> slave_cells; // cells holding a slave of the row or of the column space
  const int32_t* d_slave_cells = nullptr;
  Handle<mpcx_pairs_plan_t, mpcx_pairs_plan_destroy> pairs;
  Handle<mpcx_nodeblock_plan_t, mpcx_nodeblock_plan_destroy> nodeblock;
  Handle<mpcx_cell_plan_t, mpcx_cell_plan_destroy> cellplan;
  Handle<mpcx_master_plan_t, mpcx_master_plan_destroy> master;
};
} // namespace

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
    hipStream_t stream = nullptr, s_mat = nullptr, s_vec = nullptr;
    int prio_lo = 0, prio_hi = 0;
    hip_check(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi), "hipDeviceGetStreamPriorityRange");
    hip_check(hipStreamCreateWithPriority(&s_mat, hipStreamNonBlocking, prio_hi), "hipStreamCreateWithPriority");
    hip_check(hipStreamCreateWithPriority(&s_vec, hipStreamNonBlocking, prio_lo), "hipStreamCreateWithPriority");
    const Array &X = need(in, "x"), &CELLS = need(in, "cells");
    const int64_t n_nodes = X.n / 3, n_cells = CELLS.n / 4;
    (void)n_nodes;
    const double* d_x = dev.upload(X.as<double>(), size_t(X.n));
    const int32_t* d_cells = dev.upload(CELLS.as<int32_t>(), size_t(CELLS.n));

    auto t0 = std::chrono::steady_clock::now();
    std::vector<Space> spaces;
    for (int k = 0; in.count("s" + std::to_string(k) + "_shape"); ++k)
      spaces.push_back(make_space(in, k, n_cells, dev));
    if (spaces.empty())
      throw std::runtime_error("problem file: no space (s0_shape ...)");

    // ---- blocks: pattern (cpp/utils.h:381-496, rows constrained by the row space, columns by the column space), plans
    std::vector<std::unique_ptr<Block>> blocks;
    for (int k = 0; in.count("b" + std::to_string(k) + "_spaces"); ++k)
    {
      const std::string p = "b" + std::to_string(k);
      auto B = std::make_unique<Block>();
      const int32_t* sp = need(in, (p + "_spaces").c_str()).as<int32_t>();
      B->row = sp[0], B->col = sp[1], B->kind = sp[2];
      if (B->row < 0 || B->col < 0 || size_t(B->row) >= spaces.size() || size_t(B->col) >= spaces.size())
        throw std::runtime_error(p + "_spaces: no such space");
      const Space &R = spaces[size_t(B->row)], &C = spaces[size_t(B->col)];
      const int32_t* prm = need(in, (p + "_params").c_str()).as<int32_t>();
      B->K = make_kernel(in, p, dev, &B->constants);
      void* pat = mpcx_pattern_build(n_cells, R.dofmap_h, R.nd, R.bs, R.ndofs / R.bs, C.dofmap_h, C.nd, C.bs, C.ndofs / C.bs, R.c2s_off.data(), R.c2s.data(),
                                     R.m_off.data(), R.m_idx.data(), C.c2s_off.data(), C.c2s.data(), C.m_off.data(), C.m_idx.data(), 8);
      if (!pat)
        throw std::runtime_error(std::string("mpcx_pattern_build: ") + mpcx_last_error());
      B->nnz = mpcx_pattern_nnz(pat);
      if (mpcx_pattern_nrows(pat) != R.ndofs)
        throw std::runtime_error("mpcx_pattern_build: unexpected number of rows");
      B->rowptr.resize(size_t(R.ndofs) + 1);
      B->cols.resize(size_t(B->nnz));
      mpcx_check(mpcx_pattern_copy(pat, B->rowptr.data(), B->cols.data()), "mpcx_pattern_copy");
      mpcx_pattern_free(pat);
      B->d_rowptr = dev.upload(B->rowptr);
      B->d_cols = dev.upload(B->cols);
      B->d_vals = dev.alloc<double>(size_t(B->nnz));
      // cells with a slave of either space (the compact cell_to_slaves of the block)
      {
        std::vector<int8_t> flag(static_cast<size_t>(n_cells), 0);
        for (int32_t c : R.slave_cells)
          flag[size_t(c)] = 1;
        for (int32_t c : C.slave_cells)
          flag[size_t(c)] = 1;
        for (int64_t c = 0; c < n_cells; ++c)
          if (flag[size_t(c)])
            B->slave_cells.push_back(int32_t(c));
        B->d_slave_cells = dev.upload(B->slave_cells);
      }
      if (B->kind == 1)
        mpcx_check(mpcx_pairs_plan_create(R.ndofs, B->d_rowptr, B->rowptr.data(), B->d_cols, n_cells, nullptr, n_cells, R.d_dofmap, R.nd, R.bs, R.d_bc,
                                          R.mpc.is_slave, C.d_dofmap, C.nd, C.bs, C.d_bc, C.mpc.is_slave, prm[0], prm[1], nullptr, 0, &B->K, d_x,
                                          d_cells, 4, stream, &B->pairs.p),
                   "mpcx_pairs_plan_create");
      else if (B->kind == 2)
        mpcx_check(mpcx_nodeblock_plan_create(R.ndofs, B->d_rowptr, B->rowptr.data(), B->d_cols, n_cells, 1, nullptr, n_cells, R.d_dofmap, R.nd, R.bs,
                                              R.d_bc, R.mpc.is_slave, prm[0], prm[1], nullptr, 0, stream, &B->nodeblock.p),
                   "mpcx_nodeblock_plan_create");
      else
        mpcx_check(mpcx_cell_plan_create(R.ndofs, B->d_rowptr, B->rowptr.data(), B->d_cols, n_cells, 1, nullptr, n_cells, R.d_dofmap, R.nd, R.bs, R.d_bc,
                                         R.mpc.is_slave, C.d_dofmap, C.nd, C.bs, C.d_bc, C.mpc.is_slave, prm[0], prm[1], nullptr, 0, 1, stream,
                                         &B->cellplan.p),
                   "mpcx_cell_plan_create");
      if (!B->slave_cells.empty())
      {
        const int diag = (B->K.form == MPCX_FORM_STIFFNESS || B->K.form == MPCX_FORM_MASS) && R.bs > 1 ? 1 : 0;
        mpcx_check(mpcx_master_plan_create(int64_t(B->slave_cells.size()), B->d_slave_cells, 1, nullptr, nullptr, R.d_dofmap, R.nd, R.bs, C.d_dofmap,
                                           C.nd, C.bs, R.d_bc, C.d_bc, &R.mpc, &C.mpc, B->d_rowptr, B->d_cols, diag, stream, &B->master.p),
                   "mpcx_master_plan_create");
      }
      blocks.push_back(std::move(B));
    }

    // ---- vectors: owner-computes row blocks over the cells
    struct Vec
    {
      int space = 0;
      mpcx_kernel_t K{};
      const double* constants = nullptr;
      double* d_b = nullptr;
      Handle<mpcx_owner_plan_t, mpcx_owner_plan_destroy> plan;
      Handle<mpcx_cell_grid_plan_t, mpcx_cell_grid_plan_destroy> grid; // (the benchmark's right-hand side on cells of a box mesh)
    };
    std::vector<std::unique_ptr<Vec>> vecs;
    for (int k = 0; in.count("v" + std::to_string(k) + "_space"); ++k)
    {
      const std::string p = "v" + std::to_string(k);
      auto V = std::make_unique<Vec>();
      const int32_t* sp = need(in, (p + "_space").c_str()).as<int32_t>();
      V->space = sp[0];
      const Space& S = spaces.at(size_t(V->space));
      V->K = make_kernel(in, p, dev, &V->constants);
      if (auto it = in.find(p + "_qphi"); it != in.end() && it->second.n > 0)
        V->K.qphi = dev.upload(it->second.as<double>(), size_t(it->second.n));
      if (auto it = in.find(p + "_vphi"); it != in.end() && it->second.n > 0)
        V->K.vphi = dev.upload(it->second.as<double>(), size_t(it->second.n));
      V->d_b = dev.alloc<double>(size_t(S.ndofs));
      int32_t* mrow = dev.alloc<int32_t>(size_t(n_cells) * size_t(S.nd));
      mpcx_check(mpcx_mask_dofmap(S.d_dofmap, n_cells, S.nd, S.bs, nullptr, S.mpc.is_slave, 0, mrow, stream), "mpcx_mask_dofmap");
      mpcx_check(mpcx_owner_plan_create(n_cells, S.nd, mrow, S.bs, S.ndofs, sp[1], nullptr, 0, 12288, stream, &V->plan.p), "mpcx_owner_plan_create");
      // kernel.fn_id 1 (python/benchmarks/bench_periodic.py:85-89) on a scalar P1 / P2 space: per-interval tables of its univariate
      // factors instead of a sine and an exponential per quadrature point, when the cells are cells of a box mesh
      // (mpcx_vector_args_t::grid_eta / grid_J; return code 1 = they are not: point by point)
      if (V->K.form == MPCX_FORM_SOURCE && V->K.fn_id == 1 && V->K.celltype == 2 && V->K.coeff_degree == 0 && S.bs == 1 && (S.nd == 4 || S.nd == 10)
          && !std::getenv("MPCX_DRIVER_NO_GRID"))
      {
        const Array& q = need(in, (p + "_qpts").c_str());
        mpcx_vector_args_t tmp;
        std::memset(&tmp, 0, sizeof(tmp));
        mpcx_check(mpcx_owner_plan_fill(V->plan.p, &tmp), "mpcx_owner_plan_fill");
        const int rc = mpcx_cell_grid_plan_create(d_cells, n_cells, d_x, &tmp.plan, q.as<double>(), int32_t(q.n / 3), stream, &V->grid.p);
        if (rc < 0)
          mpcx_check(rc, "mpcx_cell_grid_plan_create");
      }
      vecs.push_back(std::move(V));
    }
    hip_check(hipDeviceSynchronize(), "set-up");
    const double t_setup = seconds_since(t0);

    // ---- the hot path, `steps` times
    double t_steps = 0.0;
    for (int step = (steps > 1 ? -1 : 0); step < steps; ++step)
    {
      hip_check(hipDeviceSynchronize(), "sync");
      t0 = std::chrono::steady_clock::now();
      for (auto& Bp : blocks)
      {
        Block& B = *Bp;
        const Space &R = spaces[size_t(B.row)], &C = spaces[size_t(B.col)];
        mpcx_matrix_args_t a;
        std::memset(&a, 0, sizeof(a));
        a.nrows = R.ndofs, a.rowptr = B.d_rowptr, a.cols = B.d_cols, a.vals = B.d_vals;
        a.kernel = B.K;
        a.x = d_x, a.x_dofmap = d_cells, a.nv = 4;
        a.estride = 1, a.n_entities = n_cells;
        a.constants = B.constants;
        a.dofmap0 = R.d_dofmap, a.nd0 = R.nd, a.bs0 = R.bs;
        a.dofmap1 = C.d_dofmap, a.nd1 = C.nd, a.bs1 = C.bs;
        a.bc0 = R.d_bc, a.bc1 = C.d_bc;
        a.mpc0 = R.mpc, a.mpc1 = C.mpc;
        a.stream = s_mat;
        if (B.kind == 1)
          mpcx_check(mpcx_pairs_plan_fill(B.pairs.p, &a), "mpcx_pairs_plan_fill");
        else if (B.kind == 2)
          mpcx_check(mpcx_nodeblock_plan_fill(B.nodeblock.p, &a), "mpcx_nodeblock_plan_fill");
        else
          mpcx_check(mpcx_cell_plan_fill(B.cellplan.p, &a), "mpcx_cell_plan_fill");
        a.store_mode = 1; // every row block is written by the launch: no zeroing pass (assemble_matrix.py:51 zeroes, then adds)
        a.slave_entities = B.d_slave_cells, a.n_slave_entities = int64_t(B.slave_cells.size());
        if (B.master.p)
          mpcx_check(mpcx_master_plan_fill(B.master.p, &a), "mpcx_master_plan_fill");
        mpcx_check(mpcx_assemble_matrix(&a), "mpcx_assemble_matrix");
        if (B.row == B.col)
        {
          // slave diagonal (cpp/assemble_matrix.cpp:711-724) and Dirichlet diagonal (insert_diagonal, assemble_matrix.py:59-62)
          mpcx_check(mpcx_add_diagonal(R.ndofs, B.d_rowptr, B.d_cols, B.d_vals, R.d_slaves, R.n_local_slaves, 1.0, s_mat), "mpcx_add_diagonal");
          mpcx_check(mpcx_add_diagonal(R.ndofs, B.d_rowptr, B.d_cols, B.d_vals, R.d_bc_dofs, int64_t(R.bc_dofs.size()), 1.0, s_mat),
                     "mpcx_add_diagonal");
        }
      }
      for (auto& Vp : vecs)
      {
        Vec& V = *Vp;
        const Space& S = spaces[size_t(V.space)];
        hip_check(hipMemsetAsync(V.d_b, 0, size_t(S.ndofs) * 8, s_vec), "hipMemsetAsync");
        mpcx_vector_args_t v;
        std::memset(&v, 0, sizeof(v));
        v.b = V.d_b, v.num_dofs = S.ndofs;
        v.kernel = V.K;
        v.x = d_x, v.x_dofmap = d_cells, v.nv = 4;
        v.estride = 1, v.n_entities = n_cells;
        v.constants = V.constants;
        v.dofmap = S.d_dofmap, v.nd = S.nd, v.bs = S.bs;
        v.mpc = S.mpc;
        v.stream = s_vec;
        mpcx_check(mpcx_owner_plan_fill(V.plan.p, &v), "mpcx_owner_plan_fill");
        v.algorithm = MPCX_ALG_ROWBLOCK;
        v.slave_entities = S.d_slave_cells, v.n_slave_entities = int64_t(S.slave_cells.size());
        if (V.grid.p)
          mpcx_check(mpcx_cell_grid_plan_fill(V.grid.p, &v), "mpcx_cell_grid_plan_fill");
        mpcx_check(mpcx_assemble_vector(&v), "mpcx_assemble_vector");
      }
      hip_check(hipDeviceSynchronize(), "step");
      if (step >= 0)
        t_steps += seconds_since(t0);
    }

    std::vector<std::pair<std::string, Array>> out;
    for (size_t k = 0; k < blocks.size(); ++k)
    {
      Block& B = *blocks[k];
      std::vector<double> vals(static_cast<size_t>(B.nnz));
      hip_check(hipMemcpy(vals.data(), B.d_vals, size_t(B.nnz) * 8, hipMemcpyDeviceToHost), "hipMemcpy D2H");
      const std::string p = "A" + std::to_string(k);
      out.push_back({p + "_rowptr", make_array(2, B.rowptr)});
      out.push_back({p + "_cols", make_array(1, B.cols)});
      out.push_back({p + "_vals", make_array(3, vals)});
    }
    for (size_t k = 0; k < vecs.size(); ++k)
    {
      const Space& S = spaces[size_t(vecs[k]->space)];
      std::vector<double> b(static_cast<size_t>(S.ndofs));
      hip_check(hipMemcpy(b.data(), vecs[k]->d_b, size_t(S.ndofs) * 8, hipMemcpyDeviceToHost), "hipMemcpy D2H");
      out.push_back({"b" + std::to_string(k), make_array(3, b)});
    }
    const std::vector<double> timings = {t_setup, t_steps / steps};
    out.push_back({"timings", make_array(3, timings)});
    write_bundle(argv[2], out);
    std::printf("mpcx_driver_blocks: %lld cells, %zu space(s), %zu block(s), %zu vector(s); set-up %.3f s, step %.3f ms\n", (long long)n_cells,
                spaces.size(), blocks.size(), vecs.size(), t_setup, 1e3 * t_steps / steps);
    return 0;
  }
  catch (const std::exception& e)
  {
    std::fprintf(stderr, "mpcx_driver_blocks: %s\n", e.what());
    return 1;
  }
}
