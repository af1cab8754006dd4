// SpMV and a Jacobi-preconditioned conjugate-gradient iteration on the assembled CSR matrix
// (the caller of the assembly path, python/src/dolfinx_mpc/problem.py
// LinearProblem.solve, so that A, b and u never leave the GPU).  gfx950 only.
//
// All kernels are HBM-bound streams.  Algorithmic bytes per CG iteration on an n x n matrix with
// nnz entries:  SpMV 12 nnz + 20 n  (vals, cols, rowptr, p, Ap; the gather of p hits L2 for a
// locality-preserving numbering), update 56 n (p, Ap, x, r, dinv read; x, r, z written),
// direction 24 n.  The scalars (r.z, p.Ap, r.r) stay in device memory: no host round trip
// inside an iteration.

#include "mpcx.h"
#include "mpcx_internal.h"

#include <cstdlib>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <string>

namespace
{
int check(hipError_t err, const char* what)
{
  if (err != hipSuccess)
  {
    mpcx_set_error(std::string(what) + ": " + hipGetErrorString(err));
    return -1;
  }
  return 0;
}
inline unsigned grid_for(int64_t n, int block) { return static_cast<unsigned>((n + block - 1) / block); }

constexpr int SPMV_THREADS = 256;
constexpr int SPMV_GROUP = 8; // lanes per row: rows of P1 tetrahedral meshes hold ~15 entries

__device__ inline double group_sum(double v)
{
#pragma unroll
  for (int o = SPMV_GROUP / 2; o > 0; o >>= 1)
    v += __shfl_xor(v, o, SPMV_GROUP);
  return v;
}

__device__ inline double wave_sum(double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    v += __shfl_xor(v, o, 64);
  return v;
}

// sum over the workgroup (<= 16 waves), result valid in thread 0; one same-address atomic per
// workgroup instead of one per wave (2.1 M same-address atomics per SpMV cost 25 ms)
__device__ inline double block_sum(double v)
{
  __shared__ double s_part[16];
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
    s_part[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < int(blockDim.x >> 6); ++i)
      t += s_part[i];
  __syncthreads();
  return t;
}

// scal layout (device doubles): [0,1] r.z (ping-pong), [2,3] p.Ap (ping-pong), [4,5] r.r, [6] b.b
enum
{
  S_RZ = 0,
  S_PAP = 2,
  S_RR = 4,
  S_BB = 6
};

// y = A x; optionally dot += x.y (one atomic per workgroup) -- SPMV_GROUP lanes share a row
template <bool DOT>
__global__ void __launch_bounds__(SPMV_THREADS)
spmv_kernel(int32_t nrows, const mpcx_nnz_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
            const double* __restrict__ vals, const double* __restrict__ x, double* __restrict__ y, double* dot,
            double* zero_a, double* zero_b)
{
  if (DOT && blockIdx.x == 0 && threadIdx.x == 0)
  {
    // slots nobody reads during this kernel, cleared for the next accumulation
    *zero_a = 0.0;
    *zero_b = 0.0;
  }
  const int lane = threadIdx.x & (SPMV_GROUP - 1);
  constexpr int ROWS = SPMV_THREADS / SPMV_GROUP; // rows per workgroup and pass
  double part = 0.0;
  // grid-stride over row groups: with DOT the grid is capped so that each workgroup ends in one atomic
  for (int64_t row = int64_t(blockIdx.x) * ROWS + threadIdx.x / SPMV_GROUP; row < nrows;
       row += int64_t(gridDim.x) * ROWS)
  {
    const int64_t lo = rowptr[row], hi = rowptr[row + 1];
    double sum = 0.0;
    for (int64_t k = lo + lane; k < hi; k += SPMV_GROUP)
      sum += vals[k] * x[cols[k]];
    sum = group_sum(sum);
    if (lane == 0)
    {
      y[row] = sum;
      if (DOT)
        part += sum * x[row];
    }
  }
  if (DOT)
  {
    part = block_sum(part);
    if (threadIdx.x == 0)
      atomicAdd(dot, part);
  }
}

// alpha = rz/pAp; x += alpha p; r -= alpha Ap; z = dinv r; rz_new += r.z; rr += r.r
__global__ void __launch_bounds__(256)
cg_update_kernel(int32_t n, const double* __restrict__ dinv, const double* __restrict__ p,
                 const double* __restrict__ Ap, double* __restrict__ x, double* __restrict__ r,
                 double* __restrict__ z, const double* rz_old, const double* pAp, double* rz_new, double* rr)
{
  const double alpha = *rz_old / *pAp;
  double a = 0.0, b = 0.0;
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256)
  {
    x[i] += alpha * p[i];
    const double ri = r[i] - alpha * Ap[i];
    const double zi = dinv[i] * ri;
    r[i] = ri;
    z[i] = zi;
    a += ri * zi;
    b += ri * ri;
  }
  a = block_sum(a);
  b = block_sum(b);
  if (threadIdx.x == 0)
  {
    atomicAdd(rz_new, a);
    atomicAdd(rr, b);
  }
}

// beta = rz_new/rz_old; p = z + beta p
__global__ void __launch_bounds__(256)
cg_direction_kernel(int32_t n, const double* __restrict__ z, double* __restrict__ p, const double* rz_new,
                    const double* rz_old, double* zero_a)
{
  if (blockIdx.x == 0 && threadIdx.x == 0)
    *zero_a = 0.0;
  const double beta = *rz_new / *rz_old;
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256)
    p[i] = z[i] + beta * p[i];
}

// start: x = 0, r = b, z = dinv b, p = z, rz[0] = r.z, rr[0] = bb = b.b
__global__ void __launch_bounds__(256)
cg_start_kernel(int32_t n, const double* __restrict__ dinv, const double* __restrict__ b, double* __restrict__ x,
                double* __restrict__ r, double* __restrict__ z, double* __restrict__ p, double* scal)
{
  double a = 0.0, c = 0.0;
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256)
  {
    const double bi = b[i], zi = dinv[i] * bi;
    x[i] = 0.0;
    r[i] = bi;
    z[i] = zi;
    p[i] = zi;
    a += bi * zi;
    c += bi * bi;
  }
  a = block_sum(a);
  c = block_sum(c);
  if (threadIdx.x == 0)
  {
    atomicAdd(scal + S_RZ, a);
    atomicAdd(scal + S_RR, c);
    atomicAdd(scal + S_BB, c);
  }
}

__global__ void inverse_diagonal_kernel(int32_t nrows, const mpcx_nnz_t* __restrict__ rowptr,
                                        const int32_t* __restrict__ cols, const double* __restrict__ vals,
                                        double* __restrict__ dinv)
{
  const int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= nrows)
    return;
  double d = 0.0;
  int64_t lo = rowptr[r], hi = rowptr[r + 1];
  while (lo < hi) // sorted columns
  {
    const int64_t mid = (lo + hi) >> 1;
    if (cols[mid] < r)
      lo = mid + 1;
    else
      hi = mid;
  }
  if (lo < rowptr[r + 1] && cols[lo] == r)
    d = vals[lo];
  dinv[r] = d != 0.0 ? 1.0 / d : 1.0;
}

// pack / unpack of the interface rows exchanged between z-slabs (dolfinx_mpc_amd/distributed.py)
__global__ void gather_f64_kernel(const double* __restrict__ v, const int64_t* __restrict__ idx, int64_t n,
                                  double* __restrict__ out)
{
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n)
    out[i] = v[idx[i]];
}
__global__ void scatter_add_f64_kernel(double* __restrict__ v, const int64_t* __restrict__ idx, int64_t n,
                                       const double* __restrict__ in)
{
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n)
    atomicAdd(v + idx[i], in[i]); // an index may occur more than once
}

inline unsigned stream_grid(int64_t n) // grid-stride kernels: enough workgroups to fill 256 CUs
{
  const int64_t g = (n + 255) / 256;
  return unsigned(g < 8192 ? (g > 0 ? g : 1) : 8192);
}
} // namespace

// HBM bandwidth probes (bench.py's roofline denominators, tools/probes): 16 bytes per lane and access, grid-stride --
// the access shape /opt/skills/guides/MI355X_MICROARCH.md quotes ~6.3 TB/s for; mode 0: copy dst = src (read + write),
// 1: read only (sum kept in a register, one store per workgroup at the end), 2: write only.
__global__ void __launch_bounds__(256) hbm_probe_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16, int mode)
{
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (mode == 0)
  {
    for (; i < n16; i += stride)
      dst[i] = src[i];
  }
  else if (mode == 1)
  {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (; i < n16; i += stride)
    {
      const uint4 v = src[i];
      acc.x ^= v.x;
      acc.y ^= v.y;
      acc.z ^= v.z;
      acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) // (practically never: keeps the loads alive)
      dst[blockIdx.x] = acc;
  }
  else if (mode == 2)
  {
    const uint4 v = make_uint4(1, 2, 3, 4);
    for (; i < n16; i += stride)
      dst[i] = v;
  }
  else
  {
    // modes 3 / 4: copy with four independent 16-byte loads in flight per lane before the first store (mode 4: non-temporal
    // loads and stores) -- the deepest simple pipeline; if these do not beat mode 0 the box, not the probe, sets the rate
    for (; i + 3 * stride < n16; i += 4 * stride)
    {
      uint4 v0, v1, v2, v3;
      if (mode == 4)
      {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        const v4u* s4 = reinterpret_cast<const v4u*>(src);
        v4u* d4 = reinterpret_cast<v4u*>(dst);
        const v4u w0 = __builtin_nontemporal_load(s4 + i), w1 = __builtin_nontemporal_load(s4 + i + stride);
        const v4u w2 = __builtin_nontemporal_load(s4 + i + 2 * stride), w3 = __builtin_nontemporal_load(s4 + i + 3 * stride);
        __builtin_nontemporal_store(w0, d4 + i), __builtin_nontemporal_store(w1, d4 + i + stride);
        __builtin_nontemporal_store(w2, d4 + i + 2 * stride), __builtin_nontemporal_store(w3, d4 + i + 3 * stride);
      }
      else
      {
        v0 = src[i], v1 = src[i + stride], v2 = src[i + 2 * stride], v3 = src[i + 3 * stride];
        dst[i] = v0, dst[i + stride] = v1, dst[i + 2 * stride] = v2, dst[i + 3 * stride] = v3;
      }
    }
    for (; i < n16; i += stride)
      dst[i] = src[i];
  }
}

extern "C" int mpcx_hbm_probe(const void* src, void* dst, int64_t bytes, int32_t mode, void* stream)
{
  if (bytes < 16 || mode < 0 || mode > 4)
    return 0;
  // 8 workgroups per CU, contiguous 4 KB per workgroup and trip; MPCX_HBM_PROBE_WGS: another grid (read at every call -- the
  // rate depends on it: 2 GiB copy 4.8-5.0 TB/s with 2048 workgroups, 5.8-5.9 with 1024 on the same box, round 5)
  const char* e = std::getenv("MPCX_HBM_PROBE_WGS");
  const int wgs = (e && std::atoi(e) > 0) ? std::atoi(e) : 256 * 8;
  hipLaunchKernelGGL(hbm_probe_kernel, dim3(wgs), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const uint4*>(src), static_cast<uint4*>(dst), bytes / 16, int(mode));
  return check(hipGetLastError(), "hbm_probe launch");
}

extern "C" int mpcx_gather_f64(const double* values, const int64_t* idx, int64_t n, double* out, void* stream)
{
  if (n == 0)
    return 0;
  hipLaunchKernelGGL(gather_f64_kernel, dim3(grid_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), values,
                     idx, n, out);
  return check(hipGetLastError(), "gather_f64 launch");
}

extern "C" int mpcx_scatter_add_f64(double* values, const int64_t* idx, int64_t n, const double* in, void* stream)
{
  if (n == 0)
    return 0;
  hipLaunchKernelGGL(scatter_add_f64_kernel, dim3(grid_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     values, idx, n, in);
  return check(hipGetLastError(), "scatter_add_f64 launch");
}

extern "C" int mpcx_spmv(int32_t nrows, const mpcx_nnz_t* rowptr, const int32_t* cols, const double* vals,
                         const double* x, double* y, void* stream)
{
  if (nrows == 0)
    return 0;
  hipLaunchKernelGGL(spmv_kernel<false>, dim3(grid_for(int64_t(nrows) * SPMV_GROUP, SPMV_THREADS)),
                     dim3(SPMV_THREADS), 0, static_cast<hipStream_t>(stream), nrows, rowptr, cols, vals, x, y, nullptr,
                     nullptr, nullptr);
  return check(hipGetLastError(), "spmv launch");
}

namespace
{
// one wave per scalar row (node n, component k): entry (k, q) of block sl
__global__ void block_expand_kernel(int32_t n_nodes, const mpcx_nnz_t* __restrict__ rowptr, int bs,
                                    const double* __restrict__ block_vals, const uint8_t* __restrict__ slot_mask,
                                    double* __restrict__ vals)
{
  const int64_t row = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= int64_t(n_nodes) * bs)
    return;
  const int64_t n = row / bs;
  const int k = int(row - n * bs);
  const int64_t p0 = rowptr[row];
  const int len = int(rowptr[row + 1] - p0);
  const int64_t slot0 = rowptr[n * bs] / (int64_t(bs) * bs);
  for (int e = lane; e < len; e += 64)
  {
    const int sl = e / bs, q = e - sl * bs;
    const bool keep = q == k && !((slot_mask[slot0 + sl] >> k) & 1);
    vals[p0 + e] = keep ? block_vals[slot0 + sl] : 0.0;
  }
}

// The same for bs = 2, 3 as a stream: a wave takes RB consecutive scalar rows, a lane two neighbouring entries of each and
// one 16-byte store (rows start on 8-byte boundaries); the row bounds are wave-uniform scalar loads, the loads of the RB rows
// are issued together.  MPCX_EXPAND_WIDE=0: the kernel above.
constexpr int EXPAND_RB = 4; // rows per wave (34.9 GB of Taylor-Hood a00 at 128^3: 13.2 ms; 8 rows 14.8; the row-at-a-time kernel 23.9)
template <int BS>
__global__ __launch_bounds__(256) void block_expand_wide_kernel(int64_t nrows, const mpcx_nnz_t* __restrict__ rowptr,
                                                                const double* __restrict__ block_vals,
                                                                const uint8_t* __restrict__ slot_mask, double* __restrict__ vals)
{
  typedef double __attribute__((ext_vector_type(2), aligned(8))) double2_a8;
  constexpr int RB = EXPAND_RB;
  const int64_t wave = int64_t(blockIdx.x) * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int64_t row0 = wave * RB;
  if (row0 >= nrows)
    return;
  int64_t p0[RB], slot0[RB];
  int len[RB], kk[RB];
#pragma unroll
  for (int u = 0; u < RB; ++u)
  {
    const int64_t row = row0 + u < nrows ? row0 + u : nrows - 1;
    const int64_t n = row / BS;
    kk[u] = int(row - n * BS);
    p0[u] = rowptr[row];
    len[u] = row0 + u < nrows ? int(rowptr[row + 1] - p0[u]) : 0;
    slot0[u] = rowptr[n * BS] / (BS * BS);
  }
  // at most one of a lane's two entries is a (k, k) entry: one value and one mask byte per lane and row, both loaded
  // unconditionally (index clamped to the row) so that the loads of the RB rows are in flight together
  int maxlen = 0;
#pragma unroll
  for (int u = 0; u < RB; ++u)
    maxlen = len[u] > maxlen ? len[u] : maxlen;
  for (int base = 0; base < maxlen; base += 128)
  {
    const int e0 = base + 2 * lane;
    const int c0 = e0 / BS, r0 = e0 - c0 * BS;
    const int r1 = r0 + 1 == BS ? 0 : r0 + 1;
    const int c1 = r0 + 1 == BS ? c0 + 1 : c0;
    double val[RB];
    int mk[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u)
    {
      int c = r0 == kk[u] ? c0 : c1;
      const int last = len[u] / BS - 1;
      c = c < last ? c : last;
      val[u] = 0.0, mk[u] = 0;
      if (last >= 0) // (uniform)
      {
        val[u] = block_vals[slot0[u] + c];
        mk[u] = slot_mask[slot0[u] + c];
      }
    }
#pragma unroll
    for (int u = 0; u < RB; ++u)
    {
      if (base >= len[u]) // (uniform)
        continue;
      const double w = ((mk[u] >> kk[u]) & 1) ? 0.0 : val[u];
      double2_a8 v;
      v.x = r0 == kk[u] ? w : 0.0;
      v.y = r1 == kk[u] ? w : 0.0;
      double* row = vals + p0[u];
      if (e0 + 1 < len[u])
        *reinterpret_cast<double2_a8*>(row + e0) = v;
      else if (e0 < len[u])
        row[e0] = v.x;
    }
  }
}

// y(n, k) = sum over the blocks of node row n: (bit k of the mask clear) s * x(col block, k); a group of 8 lanes per row
__global__ void spmv_blockscalar_kernel(int32_t n_nodes, const mpcx_nnz_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                        int bs, const double* __restrict__ block_vals, const uint8_t* __restrict__ slot_mask,
                                        const double* __restrict__ x, double* __restrict__ y)
{
  constexpr int G = 8;
  const int64_t row = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  const int lane = threadIdx.x & (G - 1);
  if (row >= int64_t(n_nodes) * bs)
    return;
  const int64_t n = row / bs;
  const int k = int(row - n * bs);
  const int64_t p0 = rowptr[n * bs]; // first row of the node: its entries list the column blocks
  const int nblk = int(rowptr[n * bs + 1] - p0) / bs;
  const int64_t slot0 = p0 / (int64_t(bs) * bs);
  double sum = 0.0;
  for (int sl = lane; sl < nblk; sl += G)
    if (!((slot_mask[slot0 + sl] >> k) & 1))
      sum += block_vals[slot0 + sl] * x[cols[p0 + int64_t(sl) * bs] + k];
#pragma unroll
  for (int m = G / 2; m > 0; m >>= 1)
    sum += __shfl_xor(sum, m, G);
  if (lane == 0)
    y[row] = sum;
}

__global__ void csr_positions_kernel(const mpcx_nnz_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                     const int32_t* __restrict__ rows, const int32_t* __restrict__ colsq, int64_t n,
                                     int64_t* __restrict__ pos)
{
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  int64_t lo = rowptr[rows[i]], hi = rowptr[rows[i] + 1];
  const int64_t end = hi;
  const int32_t c = colsq[i];
  while (lo < hi)
  {
    const int64_t mid = (lo + hi) >> 1;
    if (cols[mid] < c)
      lo = mid + 1;
    else
      hi = mid;
  }
  pos[i] = (lo < end && cols[lo] == c) ? lo : -1;
}

__global__ void spmv_coo_add_kernel(int64_t n, const int32_t* __restrict__ rows, const int32_t* __restrict__ colsq,
                                    const double* __restrict__ v, const double* __restrict__ x, double* __restrict__ y)
{
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n)
    __hip_atomic_fetch_add(y + rows[i], v[i] * x[colsq[i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
} // namespace

extern "C" int mpcx_block_expand(int32_t n_nodes, const mpcx_nnz_t* rowptr, int32_t bs, const double* block_vals,
                                 const uint8_t* slot_mask, double* vals, void* stream)
{
  if (n_nodes == 0)
    return 0;
  static const bool wide = []
  {
    const char* e = std::getenv("MPCX_EXPAND_WIDE");
    return !(e && e[0] == '0');
  }();
  if (wide && (bs == 2 || bs == 3))
  {
    const int64_t nrows = int64_t(n_nodes) * bs;
    const int64_t waves = (nrows + EXPAND_RB - 1) / EXPAND_RB;
    const dim3 grid(unsigned((waves + 3) / 4));
    if (bs == 2)
      hipLaunchKernelGGL(block_expand_wide_kernel<2>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), nrows, rowptr,
                         block_vals, slot_mask, vals);
    else
      hipLaunchKernelGGL(block_expand_wide_kernel<3>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), nrows, rowptr,
                         block_vals, slot_mask, vals);
    return check(hipGetLastError(), "block_expand launch");
  }
  const int64_t threads = int64_t(n_nodes) * bs * 64;
  hipLaunchKernelGGL(block_expand_kernel, dim3(grid_for(threads, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n_nodes,
                     rowptr, int(bs), block_vals, slot_mask, vals);
  return check(hipGetLastError(), "block_expand launch");
}

extern "C" int mpcx_spmv_blockscalar(int32_t n_nodes, const mpcx_nnz_t* rowptr, const int32_t* cols, int32_t bs,
                                     const double* block_vals, const uint8_t* slot_mask, const double* x, double* y, void* stream)
{
  if (n_nodes == 0)
    return 0;
  const int64_t threads = int64_t(n_nodes) * bs * 8;
  hipLaunchKernelGGL(spmv_blockscalar_kernel, dim3(grid_for(threads, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     n_nodes, rowptr, cols, int(bs), block_vals, slot_mask, x, y);
  return check(hipGetLastError(), "spmv_blockscalar launch");
}

extern "C" int mpcx_csr_positions(const mpcx_nnz_t* rowptr, const int32_t* cols, const int32_t* rows, const int32_t* colsq,
                                  int64_t n, int64_t* pos, void* stream)
{
  if (n == 0)
    return 0;
  hipLaunchKernelGGL(csr_positions_kernel, dim3(grid_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), rowptr, cols,
                     rows, colsq, n, pos);
  return check(hipGetLastError(), "csr_positions launch");
}

extern "C" int mpcx_spmv_coo_add(int64_t n, const int32_t* rows, const int32_t* colsq, const double* v, const double* x,
                                 double* y, void* stream)
{
  if (n == 0)
    return 0;
  hipLaunchKernelGGL(spmv_coo_add_kernel, dim3(grid_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n, rows, colsq,
                     v, x, y);
  return check(hipGetLastError(), "spmv_coo_add launch");
}

extern "C" int mpcx_inverse_diagonal(int32_t nrows, const mpcx_nnz_t* rowptr, const int32_t* cols,
                                     const double* vals, double* dinv, void* stream)
{
  if (nrows == 0)
    return 0;
  hipLaunchKernelGGL(inverse_diagonal_kernel, dim3(grid_for(nrows, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), nrows, rowptr, cols, vals, dinv);
  return check(hipGetLastError(), "inverse_diagonal launch");
}

extern "C" int mpcx_cg_start(int32_t n, const double* dinv, const double* b, double* x, double* r, double* z,
                             double* p, double* scal, void* stream)
{
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (int rc = check(hipMemsetAsync(scal, 0, 8 * sizeof(double), st), "cg_start memset"))
    return rc;
  if (n == 0)
    return 0;
  hipLaunchKernelGGL(cg_start_kernel, dim3(stream_grid(n)), dim3(256), 0, st, n, dinv, b, x, r, z, p, scal);
  return check(hipGetLastError(), "cg_start launch");
}

extern "C" int mpcx_cg_step(int32_t n, const mpcx_nnz_t* rowptr, const int32_t* cols, const double* vals,
                            const double* dinv, double* x, double* r, double* z, double* p, double* Ap,
                            double* scal, int32_t k, void* stream)
{
  if (n == 0)
    return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int cur = k & 1, nxt = cur ^ 1;
  // Ap = A p, p.Ap -> scal[S_PAP + cur]; clears r.z and r.r of the next parity
  const unsigned spmv_grid = std::min(grid_for(int64_t(n) * SPMV_GROUP, SPMV_THREADS), 16384u);
  hipLaunchKernelGGL(spmv_kernel<true>, dim3(spmv_grid), dim3(SPMV_THREADS), 0,
                     st, n, rowptr, cols, vals, p, Ap, scal + S_PAP + cur, scal + S_RZ + nxt, scal + S_RR + nxt);
  hipLaunchKernelGGL(cg_update_kernel, dim3(stream_grid(n)), dim3(256), 0, st, n, dinv, p, Ap, x, r, z,
                     scal + S_RZ + cur, scal + S_PAP + cur, scal + S_RZ + nxt, scal + S_RR + nxt);
  hipLaunchKernelGGL(cg_direction_kernel, dim3(stream_grid(n)), dim3(256), 0, st, n, z, p, scal + S_RZ + nxt,
                     scal + S_RZ + cur, scal + S_PAP + nxt);
  return check(hipGetLastError(), "cg_step launch");
}

// (mpcx_preload, csrc/mpcx_kernels.hip: the first launch from a translation unit loads its code object)
namespace
{
__global__ void preload_solver_kernel() {}
} // namespace
extern "C" int mpcx_preload_solver(void* stream)
{
  hipLaunchKernelGGL(preload_solver_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream));
  return hipGetLastError() == hipSuccess ? 0 : -100;
}
