// Device primitives of libmpcx's set-up path (rocPRIM: ships with ROCm, header-only) behind the C ABI, so that plans,
// patterns and the constraint data can be built by a caller that has nothing but HIP device memory (no torch): scans,
// a stable radix sort of (key, value) pairs, run boundaries of a sorted key array, stream compaction -- plus the
// MultiPointConstraint constructor (cpp/MultiPointConstraint.h:36-126) and create_cell_to_dofs_map
// (cpp/mpc_helpers.h:19-94) as count -> scan -> fill kernels.
// Workspace protocol: every primitive that needs temporary storage takes (temp, temp_bytes); called with temp == NULL
// it writes the size it needs to *temp_bytes and does nothing else.
#include "mpcx.h"
#include "mpcx_internal.h"

#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
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

__global__ void last_plus_kernel(const int64_t* __restrict__ excl, const int32_t* __restrict__ in, int64_t n, int64_t* out_total)
{
  if (blockIdx.x == 0 && threadIdx.x == 0)
    *out_total = n > 0 ? excl[n - 1] + int64_t(in[n - 1]) : 0;
}
__global__ void last_plus_kernel32(const int32_t* __restrict__ excl, const int32_t* __restrict__ in, int64_t n, int32_t* out_total)
{
  if (blockIdx.x == 0 && threadIdx.x == 0)
    *out_total = n > 0 ? excl[n - 1] + in[n - 1] : 0;
}

// heads[i] = 1 where keys[i] starts a run (i == 0 or keys[i] != keys[i-1])
__global__ void run_heads_kernel(const int64_t* __restrict__ keys, int64_t n, int32_t* __restrict__ heads)
{
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n)
    heads[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}
// run r (= excl[i] at a head i): run_keys[r] = keys[i], run_start[r] = i; run_start[num_runs] = n
__global__ void run_fill_kernel(const int64_t* __restrict__ keys, const int32_t* __restrict__ heads,
                                const int64_t* __restrict__ excl, int64_t n, int64_t* __restrict__ run_keys,
                                int64_t* __restrict__ run_start)
{
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n && heads[i])
  {
    const int64_t r = excl[i];
    if (run_keys)
      run_keys[r] = keys[i];
    run_start[r] = i;
  }
  if (i == n - 1)
    run_start[excl[i] + heads[i]] = n;
}

// out[b] = first position whose (key >> shift) >= b, b = 0 .. num_segments (sorted keys): the offsets of the segments
__global__ void segment_offsets_kernel(const int64_t* __restrict__ keys, int64_t n, int shift, int64_t num_segments,
                                       int64_t* __restrict__ out)
{
  const int64_t b = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (b > num_segments)
    return;
  int64_t lo = 0, hi = n;
  while (lo < hi)
  {
    const int64_t mid = (lo + hi) >> 1;
    if ((keys[mid] >> shift) < b)
      lo = mid + 1;
    else
      hi = mid;
  }
  out[b] = lo;
}
__global__ void last_plus_kernel64(const int64_t* __restrict__ excl, const int64_t* __restrict__ in, int64_t n, int64_t* out_total)
{
  if (blockIdx.x == 0 && threadIdx.x == 0)
    *out_total = n > 0 ? excl[n - 1] + in[n - 1] : 0;
}

// ---- MultiPointConstraint constructor -----------------------------------------------------------------------
__global__ void mpc_mark_kernel(int32_t num_dofs, int32_t num_slaves, const int32_t* __restrict__ slaves,
                                const int32_t* __restrict__ offsets, int8_t* __restrict__ is_slave,
                                int32_t* __restrict__ num_masters, int32_t* __restrict__ flag)
{
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num_slaves)
    return;
  const int32_t s = slaves[i];
  if (s < 0 || s >= num_dofs)
  {
    atomicOr(flag, 1);
    return;
  }
  is_slave[s] = 1;
  // a dof listed twice as a slave with masters both times: the host routine's sequential "last one wins" cannot be
  // reproduced in parallel -> flagged, the caller takes the host routine
  const int32_t cnt = offsets[i + 1] - offsets[i];
  if (cnt > 0 && atomicAdd(num_masters + s, cnt) != 0)
    atomicOr(flag, 4);
}
__global__ void mpc_fill_kernel(int32_t num_dofs, int32_t num_slaves, const int32_t* __restrict__ slaves,
                                const int64_t* __restrict__ masters, const double* __restrict__ coeffs,
                                const int32_t* __restrict__ owners, const int32_t* __restrict__ offsets,
                                const int32_t* __restrict__ masters_offsets, int32_t* __restrict__ masters_out,
                                double* __restrict__ coeffs_out, int32_t* __restrict__ owners_out, int32_t* __restrict__ flag)
{
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num_slaves)
    return;
  const int32_t s = slaves[i];
  if (s < 0 || s >= num_dofs)
    return;
  const int32_t base = masters_offsets[s];
  for (int32_t j = offsets[i]; j < offsets[i + 1]; ++j)
  {
    const int64_t m = masters[j];
    if (m < 0 || m >= num_dofs)
    {
      atomicOr(flag, 2);
      return;
    }
    const int32_t pos = base + (j - offsets[i]);
    masters_out[pos] = int32_t(m);
    coeffs_out[pos] = coeffs[j];
    owners_out[pos] = owners[j];
  }
}
__global__ void mark_to_i32_kernel(const int8_t* __restrict__ mark, int64_t n, int32_t* __restrict__ out)
{
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n)
    out[i] = mark[i] ? 1 : 0;
}
__global__ void compact_marked_kernel(const int8_t* __restrict__ mark, const int32_t* __restrict__ pos, int32_t n,
                                      int32_t num_owned, int32_t* __restrict__ out, int32_t* __restrict__ num_local)
{
  const int32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d < n && mark[d])
    out[pos[d]] = d;
  if (d == 0)
    *num_local = pos[num_owned < n ? num_owned : n]; // pos has n + 1 entries, the last one is the total
}

// ---- cell -> slaves -----------------------------------------------------------------------------------------
template <bool FILL>
__global__ void cell_slaves_kernel(int64_t num_cells, int nd, int bs, const int32_t* __restrict__ dofmap,
                                   const int8_t* __restrict__ is_slave, int32_t* __restrict__ counts,
                                   const int32_t* __restrict__ offsets, int32_t* __restrict__ c2s)
{
  const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c >= num_cells)
    return;
  int n = 0;
  int32_t* out = FILL ? c2s + offsets[c] : nullptr;
  for (int i = 0; i < nd; ++i)
    for (int k = 0; k < bs; ++k)
    {
      const int32_t d = dofmap[c * nd + i] * bs + k;
      if (is_slave[d])
      {
        if constexpr (FILL)
        {
          // insertion sort: each cell's slaves ascending by dof (the reference inverts a dof -> cells map whose nodes
          // are visited in ascending dof order, cpp/mpc_helpers.h:19-94)
          int p = n;
          while (p > 0 && out[p - 1] > d)
          {
            out[p] = out[p - 1];
            --p;
          }
          out[p] = d;
        }
        ++n;
      }
    }
  if constexpr (!FILL)
    counts[c] = n;
}
} // namespace

extern "C" int mpcx_scan_exclusive_i32_i64(const int32_t* in, int64_t n, int64_t* out, void* temp, size_t* temp_bytes,
                                           void* stream)
{
  hipStream_t st = static_cast<hipStream_t>(stream);
  size_t need = 0;
  auto tin = rocprim::make_transform_iterator(in, [] __device__(int32_t v) { return int64_t(v); });
  if (int rc = check(rocprim::exclusive_scan(nullptr, need, tin, out, int64_t(0), size_t(n > 0 ? n : 1), rocprim::plus<int64_t>(), st),
                     "rocprim::exclusive_scan (size)"))
    return rc;
  if (!temp)
  {
    *temp_bytes = need;
    return 0;
  }
  if (*temp_bytes < need)
  {
    mpcx_set_error("mpcx_scan_exclusive_i32_i64: workspace too small");
    return -3;
  }
  if (n > 0)
    if (int rc = check(rocprim::exclusive_scan(temp, need, tin, out, int64_t(0), size_t(n), rocprim::plus<int64_t>(), st),
                       "rocprim::exclusive_scan"))
      return rc;
  hipLaunchKernelGGL(last_plus_kernel, dim3(1), dim3(1), 0, st, out, in, n, out + n);
  return check(hipGetLastError(), "scan total");
}

extern "C" int mpcx_scan_exclusive_i32(const int32_t* in, int64_t n, int32_t* out, void* temp, size_t* temp_bytes, void* stream)
{
  hipStream_t st = static_cast<hipStream_t>(stream);
  size_t need = 0;
  if (int rc = check(rocprim::exclusive_scan(nullptr, need, in, out, int32_t(0), size_t(n > 0 ? n : 1), rocprim::plus<int32_t>(), st),
                     "rocprim::exclusive_scan (size)"))
    return rc;
  if (!temp)
  {
    *temp_bytes = need;
    return 0;
  }
  if (*temp_bytes < need)
  {
    mpcx_set_error("mpcx_scan_exclusive_i32: workspace too small");
    return -3;
  }
  if (n > 0)
    if (int rc = check(rocprim::exclusive_scan(temp, need, in, out, int32_t(0), size_t(n), rocprim::plus<int32_t>(), st),
                       "rocprim::exclusive_scan"))
      return rc;
  hipLaunchKernelGGL(last_plus_kernel32, dim3(1), dim3(1), 0, st, out, in, n, out + n);
  return check(hipGetLastError(), "scan total");
}

extern "C" int mpcx_scan_exclusive_i64(const int64_t* in, int64_t n, int64_t* out, void* temp, size_t* temp_bytes, void* stream)
{
  hipStream_t st = static_cast<hipStream_t>(stream);
  size_t need = 0;
  if (int rc = check(rocprim::exclusive_scan(nullptr, need, in, out, int64_t(0), size_t(n > 0 ? n : 1), rocprim::plus<int64_t>(), st),
                     "rocprim::exclusive_scan (size)"))
    return rc;
  if (!temp)
  {
    *temp_bytes = need;
    return 0;
  }
  if (*temp_bytes < need)
  {
    mpcx_set_error("mpcx_scan_exclusive_i64: workspace too small");
    return -3;
  }
  if (n > 0)
    if (int rc = check(rocprim::exclusive_scan(temp, need, in, out, int64_t(0), size_t(n), rocprim::plus<int64_t>(), st),
                       "rocprim::exclusive_scan"))
      return rc;
  hipLaunchKernelGGL(last_plus_kernel64, dim3(1), dim3(1), 0, st, out, in, n, out + n);
  return check(hipGetLastError(), "scan total");
}

extern "C" int mpcx_segment_offsets(const int64_t* sorted_keys, int64_t n, int32_t shift, int64_t num_segments, int64_t* out,
                                    void* stream)
{
  hipLaunchKernelGGL(segment_offsets_kernel, dim3(grid_for(num_segments + 1, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     sorted_keys, n, int(shift), num_segments, out);
  return check(hipGetLastError(), "segment_offsets launch");
}

extern "C" int mpcx_sort_pairs_i64_i32(const int64_t* keys_in, int64_t* keys_out, const int32_t* vals_in, int32_t* vals_out,
                                       int64_t n, int32_t begin_bit, int32_t end_bit, void* temp, size_t* temp_bytes, void* stream)
{
  hipStream_t st = static_cast<hipStream_t>(stream);
  size_t need = 0;
  if (int rc = check(rocprim::radix_sort_pairs(nullptr, need, keys_in, keys_out, vals_in, vals_out, size_t(n > 0 ? n : 1),
                                               unsigned(begin_bit), unsigned(end_bit), st),
                     "rocprim::radix_sort_pairs (size)"))
    return rc;
  if (!temp)
  {
    *temp_bytes = need;
    return 0;
  }
  if (*temp_bytes < need)
  {
    mpcx_set_error("mpcx_sort_pairs_i64_i32: workspace too small");
    return -3;
  }
  if (n == 0)
    return 0;
  return check(rocprim::radix_sort_pairs(temp, need, keys_in, keys_out, vals_in, vals_out, size_t(n), unsigned(begin_bit),
                                         unsigned(end_bit), st),
               "rocprim::radix_sort_pairs");
}

extern "C" int mpcx_sort_pairs_i64_i64(const int64_t* keys_in, int64_t* keys_out, const int64_t* vals_in, int64_t* vals_out,
                                       int64_t n, int32_t begin_bit, int32_t end_bit, void* temp, size_t* temp_bytes, void* stream)
{
  hipStream_t st = static_cast<hipStream_t>(stream);
  size_t need = 0;
  if (int rc = check(rocprim::radix_sort_pairs(nullptr, need, keys_in, keys_out, vals_in, vals_out, size_t(n > 0 ? n : 1),
                                               unsigned(begin_bit), unsigned(end_bit), st),
                     "rocprim::radix_sort_pairs (size)"))
    return rc;
  if (!temp)
  {
    *temp_bytes = need;
    return 0;
  }
  if (*temp_bytes < need)
  {
    mpcx_set_error("mpcx_sort_pairs_i64_i64: workspace too small");
    return -3;
  }
  if (n == 0)
    return 0;
  return check(rocprim::radix_sort_pairs(temp, need, keys_in, keys_out, vals_in, vals_out, size_t(n), unsigned(begin_bit),
                                         unsigned(end_bit), st),
               "rocprim::radix_sort_pairs");
}

extern "C" int mpcx_run_heads(const int64_t* sorted_keys, int64_t n, int32_t* heads, void* stream)
{
  if (n == 0)
    return 0;
  hipLaunchKernelGGL(run_heads_kernel, dim3(grid_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), sorted_keys, n, heads);
  return check(hipGetLastError(), "run_heads launch");
}

extern "C" int mpcx_run_fill(const int64_t* sorted_keys, const int32_t* heads, const int64_t* heads_scan, int64_t n,
                             int64_t* run_keys, int64_t* run_start, void* stream)
{
  if (n == 0)
    return 0;
  hipLaunchKernelGGL(run_fill_kernel, dim3(grid_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), sorted_keys, heads,
                     heads_scan, n, run_keys, run_start);
  return check(hipGetLastError(), "run_fill launch");
}

// ---------------------------------------------------------------------------------------------------------
extern "C" int mpcx_mpc_finalize_device(int32_t num_dofs, int32_t num_owned_dofs, int32_t num_slaves, const int32_t* slaves,
                                        const int64_t* masters, const double* coeffs, const int32_t* owners,
                                        const int32_t* offsets, int8_t* is_slave, int32_t* sorted_slaves,
                                        int32_t* num_local_slaves, int32_t* masters_offsets, int32_t* masters_out,
                                        double* coeffs_out, int32_t* owners_out, int32_t* work, int32_t* flag, void* temp,
                                        size_t* temp_bytes, void* stream)
{
  hipStream_t st = static_cast<hipStream_t>(stream);
  size_t need = 0;
  if (int rc = mpcx_scan_exclusive_i32(nullptr, num_dofs, nullptr, nullptr, &need, stream))
    return rc;
  if (!temp)
  {
    *temp_bytes = need;
    return 0;
  }
  // work [2 * num_dofs + 1]: masters per dof, then the slave marker as int32 and its scan for the compaction
  if (int rc = check(hipMemsetAsync(is_slave, 0, size_t(num_dofs), st), "hipMemsetAsync"))
    return rc;
  if (int rc = check(hipMemsetAsync(work, 0, size_t(num_dofs) * 4, st), "hipMemsetAsync"))
    return rc;
  if (num_slaves > 0)
    hipLaunchKernelGGL(mpc_mark_kernel, dim3(grid_for(num_slaves, 256)), dim3(256), 0, st, num_dofs, num_slaves, slaves, offsets,
                       is_slave, work, flag);
  size_t tb = *temp_bytes;
  if (int rc = mpcx_scan_exclusive_i32(work, num_dofs, masters_offsets, temp, &tb, stream))
    return rc;
  if (num_slaves > 0)
    hipLaunchKernelGGL(mpc_fill_kernel, dim3(grid_for(num_slaves, 256)), dim3(256), 0, st, num_dofs, num_slaves, slaves, masters,
                       coeffs, owners, offsets, masters_offsets, masters_out, coeffs_out, owners_out, flag);
  if (num_dofs > 0)
  {
    hipLaunchKernelGGL(mark_to_i32_kernel, dim3(grid_for(num_dofs, 256)), dim3(256), 0, st, is_slave, int64_t(num_dofs), work);
    // scan in place is not allowed: positions go to a second use of masters-per-dof storage behind `work`
    int32_t* pos = work + num_dofs;
    tb = *temp_bytes;
    if (int rc = mpcx_scan_exclusive_i32(work, num_dofs, pos, temp, &tb, stream))
      return rc;
    hipLaunchKernelGGL(compact_marked_kernel, dim3(grid_for(num_dofs, 256)), dim3(256), 0, st, is_slave, pos, num_dofs,
                       num_owned_dofs, sorted_slaves, num_local_slaves);
  }
  return check(hipGetLastError(), "mpc_finalize_device launch");
}

extern "C" int mpcx_cell_to_slaves_device(int64_t num_cells, int32_t nd, int32_t bs, const int32_t* dofmap,
                                          const int8_t* is_slave, int32_t* counts, const int32_t* c2s_offsets, int32_t* c2s,
                                          void* stream)
{
  if (num_cells == 0)
    return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(grid_for(num_cells, 256));
  if (!c2s)
    hipLaunchKernelGGL(cell_slaves_kernel<false>, grid, dim3(256), 0, st, num_cells, int(nd), int(bs), dofmap, is_slave, counts,
                       c2s_offsets, c2s);
  else
    hipLaunchKernelGGL(cell_slaves_kernel<true>, grid, dim3(256), 0, st, num_cells, int(nd), int(bs), dofmap, is_slave, counts,
                       c2s_offsets, c2s);
  return check(hipGetLastError(), "cell_to_slaves_device launch");
}

// (mpcx_preload, csrc/mpcx_kernels.hip: the first launch from a translation unit loads its code object)
namespace
{
__global__ void preload_prims_kernel() {}
} // namespace
extern "C" int mpcx_preload_prims(void* stream)
{
  hipLaunchKernelGGL(preload_prims_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream));
  return hipGetLastError() == hipSuccess ? 0 : -100;
}
