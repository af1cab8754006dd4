// Owner-computes plan of the row-block VECTOR kernels (include/mpcx.h, mpcx_vector_args_t::own_*) built on the device
// through the C ABI: every work item (an entity of the integral, or a cell cluster with its eight vertices) belongs to
// the row block that holds the rows of its local dof 0; the dofs of other blocks its items touch are the block's halo,
// appended to its LDS copy.  Three fused passes over the (item, local dof) table replace a chain of full-size
// gathers / searches / index puts; scans, sorts and run lengths in between are mpcx_prims.hip's (rocPRIM).
//   mpcx_owner_plan_count   owner block of every item (as sort key) + its number of foreign dofs
//   mpcx_owner_plan_keys    (block << 32 | dof, item * nd + i) of every foreign dof; local map of the own dofs
//   mpcx_owner_plan_halo    local map of the foreign dofs from the sorted, run-numbered keys
// The reference has no counterpart: its vector loop adds into the global array (cpp/assemble_vector.cpp:60-110).
#include "mpcx.h"
#include "mpcx_internal.h"

#include <cstdlib>
#include <hip/hip_runtime.h>
#include <string>

namespace
{
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

constexpr int32_t DOF_MASK = (1 << 28) - 1;

// largest b in [0, nb) with row0[b] <= row
__device__ inline int32_t find_block(const int32_t* __restrict__ row0, int32_t nb, int32_t row)
{
  int32_t lo = 0, hi = nb; // row0[lo] <= row < row0[hi]
  while (hi - lo > 1)
  {
    const int32_t mid = (lo + hi) >> 1;
    if (row0[mid] <= row)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

__global__ void owner_count_kernel(int64_t n, int nd, const int32_t* __restrict__ mrow, int bs, int32_t nb,
                                   const int32_t* __restrict__ row0, int64_t* __restrict__ owner_key,
                                   int32_t* __restrict__ item, int32_t* __restrict__ fcount)
{
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n)
    return;
  const int32_t* d = mrow + e * nd;
  const int32_t own = find_block(row0, nb, (d[0] & DOF_MASK) * bs);
  int32_t c = 0;
  for (int i = 1; i < nd; ++i)
  {
    const int32_t row = (d[i] & DOF_MASK) * bs;
    c += (row < row0[own] || row >= row0[own + 1]) ? 1 : 0;
  }
  owner_key[e] = own;
  item[e] = static_cast<int32_t>(e);
  fcount[e] = c;
}

__global__ void owner_keys_kernel(int64_t n, int nd, const int32_t* __restrict__ mrow, int bs, int32_t nb,
                                  const int32_t* __restrict__ row0, const int64_t* __restrict__ owner_key,
                                  const int64_t* __restrict__ foff, int64_t* __restrict__ keys, int32_t* __restrict__ src,
                                  int32_t* __restrict__ lmap)
{
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n)
    return;
  const int32_t* d = mrow + e * nd;
  const int32_t own = static_cast<int32_t>(owner_key[e]);
  const int32_t r0 = row0[own], r1 = row0[own + 1];
  int64_t at = foff[e];
  for (int i = 0; i < nd; ++i)
  {
    const int32_t dof = d[i] & DOF_MASK;
    const int32_t row = dof * bs;
    if (row >= r0 && row < r1)
      lmap[e * nd + i] = (dof - r0 / bs) | (d[i] & ~DOF_MASK);
    else
    {
      keys[at] = (int64_t(own) << 32) | int64_t(dof);
      src[at] = static_cast<int32_t>(e * nd + i);
      ++at;
    }
  }
}

__global__ void owner_halo_kernel(int64_t nf, const int64_t* __restrict__ keys, const int32_t* __restrict__ src,
                                  const int32_t* __restrict__ heads, const int64_t* __restrict__ heads_scan,
                                  const int64_t* __restrict__ hoff, const int32_t* __restrict__ row0, int bs,
                                  const int32_t* __restrict__ mrow, int32_t* __restrict__ lmap)
{
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= nf)
    return;
  const int64_t uid = heads_scan[t] + heads[t] - 1; // index of this element's run among the distinct (block, dof) keys
  const int32_t blk = static_cast<int32_t>(keys[t] >> 32);
  const int32_t nown = (row0[blk + 1] - row0[blk]) / bs;
  const int32_t s = src[t];
  lmap[s] = static_cast<int32_t>(nown + (uid - hoff[blk])) | (mrow[s] & ~DOF_MASK);
}

// rows (own + halo) of the largest block, times bs
__global__ void owner_max_rows_kernel(int32_t nb, const int32_t* __restrict__ row0, const int64_t* __restrict__ hoff, int bs,
                                      int32_t* __restrict__ out)
{
  const int32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb)
    return;
  const int32_t rows = (row0[b + 1] - row0[b]) + static_cast<int32_t>(hoff[b + 1] - hoff[b]) * bs;
  atomicMax(out, rows);
}

// low 32 bits of every key as an int64 sort key + iota payload (the spill order of the halo rows)
__global__ void low_word_iota_kernel(int64_t n, const int64_t* __restrict__ keys, int64_t* __restrict__ low, int32_t* __restrict__ iota)
{
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n)
    return;
  low[t] = keys[t] & 0xffffffffLL;
  iota[t] = static_cast<int32_t>(t);
}
} // namespace

extern "C" int mpcx_owner_plan_count(int64_t n, int32_t nd, const int32_t* mrow, int32_t bs, int32_t nb, const int32_t* row0,
                                     int64_t* owner_key, int32_t* item, int32_t* fcount, void* stream)
{
  if (n < 0 || nd <= 0 || bs <= 0 || nb <= 0 || !mrow || !row0 || !owner_key || !item || !fcount)
  {
    mpcx_set_error("mpcx_owner_plan_count: invalid arguments");
    return -1;
  }
  if (n == 0)
    return 0;
  owner_count_kernel<<<grid_for(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(n, nd, mrow, bs, nb, row0, owner_key, item,
                                                                                       fcount);
  return check(hipGetLastError(), "mpcx_owner_plan_count");
}

extern "C" int mpcx_owner_plan_keys(int64_t n, int32_t nd, const int32_t* mrow, int32_t bs, int32_t nb, const int32_t* row0,
                                    const int64_t* owner_key, const int64_t* foff, int64_t* keys, int32_t* src, int32_t* lmap,
                                    void* stream)
{
  if (n < 0 || nd <= 0 || bs <= 0 || nb <= 0 || !mrow || !row0 || !owner_key || !foff || !lmap)
  {
    mpcx_set_error("mpcx_owner_plan_keys: invalid arguments");
    return -1;
  }
  if (n * nd >= (int64_t(1) << 31))
  {
    mpcx_set_error("mpcx_owner_plan_keys: item * nd + i does not fit 32 bits");
    return -2;
  }
  if (n == 0)
    return 0;
  owner_keys_kernel<<<grid_for(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(n, nd, mrow, bs, nb, row0, owner_key, foff,
                                                                                      keys, src, lmap);
  return check(hipGetLastError(), "mpcx_owner_plan_keys");
}

extern "C" int mpcx_owner_plan_halo(int64_t nf, const int64_t* sorted_keys, const int32_t* sorted_src, const int32_t* heads,
                                    const int64_t* heads_scan, const int64_t* hoff, int32_t nb, const int32_t* row0, int32_t bs,
                                    const int32_t* mrow, int32_t* lmap, int32_t* max_rows, void* stream)
{
  if (nf < 0 || nb <= 0 || !hoff || !row0 || !max_rows || (nf > 0 && (!sorted_keys || !sorted_src || !heads || !heads_scan || !mrow || !lmap)))
  {
    mpcx_set_error("mpcx_owner_plan_halo: invalid arguments");
    return -1;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (nf > 0)
    owner_halo_kernel<<<grid_for(nf, 256), 256, 0, st>>>(nf, sorted_keys, sorted_src, heads, heads_scan, hoff, row0, bs, mrow, lmap);
  if (int rc = check(hipMemsetAsync(max_rows, 0, sizeof(int32_t), st), "mpcx_owner_plan_halo"))
    return rc;
  owner_max_rows_kernel<<<grid_for(nb, 256), 256, 0, st>>>(nb, row0, hoff, bs, max_rows);
  return check(hipGetLastError(), "mpcx_owner_plan_halo");
}

extern "C" int mpcx_low_word_iota(int64_t n, const int64_t* keys, int64_t* low, int32_t* iota, void* stream)
{
  if (n < 0 || (n > 0 && (!keys || !low || !iota)))
  {
    mpcx_set_error("mpcx_low_word_iota: invalid arguments");
    return -1;
  }
  if (n == 0)
    return 0;
  low_word_iota_kernel<<<grid_for(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(n, keys, low, iota);
  return check(hipGetLastError(), "mpcx_low_word_iota");
}


// ---------------------------------------------------------------------------------------------------------------
// Internal renumbering for locality (dolfinx_mpc_amd/locality.py): a mesh whose numbering has no locality is assembled on
// a spatially reordered twin (A2 = P A P^T, rows / columns renumbered), and the values are handed back in the caller's
// numbering.  mpcx_csr_permutation: src[k] = position in the TWIN's CSR of entry k of the caller's CSR (row r -> twin
// row new_of_old0[r], column c -> twin column new_of_old1[c], found by binary search in the twin's sorted row) -- the
// per-row column search of MatSetValuesLocal (cpp/assemble_matrix.cpp:546) once more, hoisted to set-up;
// mpcx_permute_values: dst[k] = vals2[src[k]].  The gather form: the caller's values are written in order (full cache
// lines, no read-modify-write of partially written lines -- the scatter form dst[dest[k]] = src[k] measured 1.6 ms
// for the 254 M entries of config 2), and the entries of a caller row come from ONE twin row, so the 8-byte reads of
// neighbouring lanes share lines.
// ---------------------------------------------------------------------------------------------------------------
namespace
{
template <class IDX>
__global__ void __launch_bounds__(256)
    csr_permutation_kernel(int32_t nrows, const mpcx_nnz_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                           const int32_t* __restrict__ new_of_old0, const int32_t* __restrict__ new_of_old1,
                           const mpcx_nnz_t* __restrict__ rowptr2, const int32_t* __restrict__ cols2, IDX* __restrict__ src,
                           int32_t* __restrict__ bad)
{
  // one wave per caller row
  const int64_t w = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (w >= nrows)
    return;
  const int32_t r2 = new_of_old0[w];
  const int64_t p0 = rowptr2[r2], p1 = rowptr2[r2 + 1];
  for (int64_t k = rowptr[w] + lane; k < rowptr[w + 1]; k += 64)
  {
    const int32_t c = new_of_old1[cols[k]];
    int64_t l = p0, h = p1;
    while (l < h)
    {
      const int64_t m = (l + h) >> 1;
      if (cols2[m] < c)
        l = m + 1;
      else
        h = m;
    }
    if (l >= p1 || cols2[l] != c)
    {
      *bad = 1;
      l = p0;
    }
    src[k] = IDX(l);
  }
}

template <class IDX>
__global__ void __launch_bounds__(256)
    permute_values_kernel(int64_t n, const IDX* __restrict__ src, const double* __restrict__ vals2, double* __restrict__ dst)
{
  // grid-stride; MPCX_PERMUTE_WGS bounds the grid (round 5 experiment: the 2 GiB copy probe streams fastest from ~1024
  // workgroups, and a million short workgroups might keep the vector kernel of the same step out -- but the shuffled config-2
  // step measured 5.34 / 5.26 / 5.26 ms with 1024 / 2048 / 4096 workgroups against 5.16 ms unbounded: the gather through
  // ``src`` is not a streaming read, and the default stays one workgroup per 256 entries)
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; k < n; k += stride)
    dst[k] = vals2[src[k]];
}
template <class IDX>
__global__ void __launch_bounds__(256) invert_permutation_kernel(int64_t n, const IDX* __restrict__ src, IDX* __restrict__ out)
{
  const int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k < n)
    out[src[k]] = IDX(k);
}
// one thread per row of the launch's CSR: rank of every entry inside the caller's row (mpcx_write_out_order)
template <class IDX>
__global__ void __launch_bounds__(256)
    write_out_order_kernel(int32_t nrows, const mpcx_nnz_t* __restrict__ rowptr, const IDX* __restrict__ val_map,
                           IDX* __restrict__ out_map, int16_t* __restrict__ out_delta, int32_t* __restrict__ bad)
{
  const int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= nrows)
    return;
  const int64_t rs = rowptr[r], re = rowptr[r + 1];
  if (re == rs)
    return;
  IDX lo = val_map[rs], hi = lo;
  for (int64_t k = rs + 1; k < re; ++k)
  {
    const IDX c = val_map[k];
    lo = c < lo ? c : lo;
    hi = c > hi ? c : hi;
  }
  if (int64_t(hi - lo) != re - rs - 1 || re - rs > 32767)
  {
    *bad = 1;
    return;
  }
  for (int64_t k = rs; k < re; ++k)
  {
    const IDX c = val_map[k];
    const int64_t slot = rs + int64_t(c - lo); // the slot that writes caller position c reads entry k
    out_map[slot] = c;
    out_delta[slot] = int16_t(k - slot);
  }
}
inline unsigned permute_grid(int64_t n)
{
  static const int wgs = []
  {
    const char* e = std::getenv("MPCX_PERMUTE_WGS");
    const int v = e ? std::atoi(e) : 0;
    return v > 0 ? v : 0x7fffffff;
  }();
  const int64_t g = (n + 255) / 256;
  return unsigned(g < wgs ? (g > 0 ? g : 1) : wgs);
}
} // namespace

extern "C" int mpcx_csr_permutation(int32_t nrows, const mpcx_nnz_t* rowptr, const int32_t* cols, const int32_t* new_of_old0,
                                    const int32_t* new_of_old1, const mpcx_nnz_t* rowptr2, const int32_t* cols2, void* src,
                                    int32_t wide, int32_t* bad, void* stream)
{
  if (nrows <= 0)
    return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(grid_for(int64_t(nrows) * 64, 256));
  if (wide)
    hipLaunchKernelGGL(csr_permutation_kernel<int64_t>, grid, dim3(256), 0, st, nrows, rowptr, cols, new_of_old0, new_of_old1,
                       rowptr2, cols2, static_cast<int64_t*>(src), bad);
  else
    hipLaunchKernelGGL(csr_permutation_kernel<uint32_t>, grid, dim3(256), 0, st, nrows, rowptr, cols, new_of_old0, new_of_old1,
                       rowptr2, cols2, static_cast<uint32_t*>(src), bad);
  return check(hipGetLastError(), "csr_permutation_kernel launch");
}

extern "C" int mpcx_permute_values(int64_t n, const void* src, int32_t wide, const double* vals2, double* dst, void* stream)
{
  if (n <= 0)
    return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (wide)
    hipLaunchKernelGGL(permute_values_kernel<int64_t>, dim3(permute_grid(n)), dim3(256), 0, st, n, static_cast<const int64_t*>(src),
                       vals2, dst);
  else
    hipLaunchKernelGGL(permute_values_kernel<uint32_t>, dim3(permute_grid(n)), dim3(256), 0, st, n,
                       static_cast<const uint32_t*>(src), vals2, dst);
  return check(hipGetLastError(), "permute_values_kernel launch");
}

extern "C" int mpcx_write_out_order(int32_t nrows, const mpcx_nnz_t* rowptr, const void* val_map, int32_t wide, void* out_map,
                                    int16_t* out_delta, int32_t* bad, void* stream)
{
  if (nrows <= 0)
    return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned grid = unsigned((int64_t(nrows) + 255) / 256);
  if (wide)
    hipLaunchKernelGGL(write_out_order_kernel<int64_t>, dim3(grid), dim3(256), 0, st, nrows, rowptr,
                       static_cast<const int64_t*>(val_map), static_cast<int64_t*>(out_map), out_delta, bad);
  else
    hipLaunchKernelGGL(write_out_order_kernel<uint32_t>, dim3(grid), dim3(256), 0, st, nrows, rowptr,
                       static_cast<const uint32_t*>(val_map), static_cast<uint32_t*>(out_map), out_delta, bad);
  return check(hipGetLastError(), "write_out_order_kernel launch");
}

extern "C" int mpcx_invert_permutation(int64_t n, const void* src, int32_t wide, void* out, void* stream)
{
  if (n <= 0)
    return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned grid = unsigned((n + 255) / 256);
  if (wide)
    hipLaunchKernelGGL(invert_permutation_kernel<int64_t>, dim3(grid), dim3(256), 0, st, n, static_cast<const int64_t*>(src),
                       static_cast<int64_t*>(out));
  else
    hipLaunchKernelGGL(invert_permutation_kernel<uint32_t>, dim3(grid), dim3(256), 0, st, n, static_cast<const uint32_t*>(src),
                       static_cast<uint32_t*>(out));
  return check(hipGetLastError(), "invert_permutation_kernel launch");
}

// ---------------------------------------------------------------------------------------------------------------
// The renumbered copy of a mesh and the dof permutation of a space on it (dolfinx_mpc_amd/locality.py: the twin of a mesh
// without locality), as plain gathers on the device: x_out[new_of_old[n]] = x[n], cells_out[c][i] =
// new_of_old[cells[old_of_new[c]][i]], cell_new_of_old[old_of_new[c]] = c -- the host's fancy indexing of 4 x 10^8 node ids
// took 12 of the 23 s a shuffled 256^3 mesh spent in the library before its first assembly.
// ---------------------------------------------------------------------------------------------------------------
namespace
{
__global__ void __launch_bounds__(256)
    renumber_mesh_kernel(const double* __restrict__ x, int64_t n_nodes, const int32_t* __restrict__ cells, int64_t n_cells, int nv,
                         const int64_t* __restrict__ node_new_of_old, const int64_t* __restrict__ cell_old_of_new,
                         double* __restrict__ x_out, int32_t* __restrict__ cells_out, int64_t* __restrict__ cell_new_of_old)
{
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t < n_nodes)
  {
    const int64_t m = node_new_of_old[t];
    x_out[3 * m] = x[3 * t], x_out[3 * m + 1] = x[3 * t + 1], x_out[3 * m + 2] = x[3 * t + 2];
  }
  if (t < n_cells)
  {
    const int64_t old = cell_old_of_new[t];
    cell_new_of_old[old] = t;
    for (int i = 0; i < nv; ++i)
      cells_out[t * nv + i] = int32_t(node_new_of_old[cells[old * nv + i]]);
  }
}
__global__ void __launch_bounds__(256)
    dof_permutation_kernel(const int32_t* __restrict__ dofmap_old, const int32_t* __restrict__ dofmap_new,
                           const int64_t* __restrict__ cell_new_of_old, int64_t n_cells, int nd, int64_t* __restrict__ new_of_old)
{
  const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c >= n_cells)
    return;
  const int64_t c2 = cell_new_of_old[c];
  for (int i = 0; i < nd; ++i) // (a dof is written by every cell that holds it, always with the same value)
    new_of_old[dofmap_old[c * nd + i]] = dofmap_new[c2 * nd + i];
}
} // namespace

extern "C" int mpcx_renumber_mesh(const double* x, int64_t n_nodes, const int32_t* cells, int64_t n_cells, int32_t nv,
                                  const int64_t* node_new_of_old, const int64_t* cell_old_of_new, double* x_out,
                                  int32_t* cells_out, int64_t* cell_new_of_old, void* stream)
{
  const int64_t n = n_nodes > n_cells ? n_nodes : n_cells;
  if (n <= 0)
    return 0;
  hipLaunchKernelGGL(renumber_mesh_kernel, dim3(grid_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, n_nodes, cells,
                     n_cells, int(nv), node_new_of_old, cell_old_of_new, x_out, cells_out, cell_new_of_old);
  return check(hipGetLastError(), "renumber_mesh_kernel launch");
}

extern "C" int mpcx_dof_permutation(const int32_t* dofmap_old, const int32_t* dofmap_new, const int64_t* cell_new_of_old,
                                    int64_t n_cells, int32_t nd, int64_t* new_of_old, void* stream)
{
  if (n_cells <= 0)
    return 0;
  hipLaunchKernelGGL(dof_permutation_kernel, dim3(grid_for(n_cells, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), dofmap_old,
                     dofmap_new, cell_new_of_old, n_cells, int(nd), new_of_old);
  return check(hipGetLastError(), "dof_permutation_kernel launch");
}

// (mpcx_preload, csrc/mpcx_kernels.hip: the first launch from a translation unit loads its code object)
namespace
{
__global__ void preload_plans_kernel() {}
} // namespace
extern "C" int mpcx_preload_plans(void* stream)
{
  hipLaunchKernelGGL(preload_plans_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream));
  return hipGetLastError() == hipSuccess ? 0 : -100;
}
