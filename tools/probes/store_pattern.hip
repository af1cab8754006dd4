// Store-pattern probe for the CSR-valued node-block write-out (round 6): which way of writing 35 GB of rows of ~87 doubles
// reaches the write rate a memset gets on this box?
//   hipcc -O3 --offload-arch=gfx950 tools/probes/store_pattern.hip -o gpurun_out/store_pattern && gpurun_out/store_pattern
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>

typedef double __attribute__((ext_vector_type(2), aligned(8))) double2_a8;
typedef double __attribute__((ext_vector_type(2))) double2_a16;

constexpr int ROW = 87;            // entries of an interior P2 row (29 column blocks of 3)
constexpr int CHUNK_ROWS = 848;    // rows of a node block (~590 KB)
constexpr int64_t CHUNK = int64_t(ROW) * CHUNK_ROWS;

// A: the memset pattern: 16 aligned bytes per lane, a wave 1 KB, grid-stride
__global__ void k_flat(double* out, int64_t n)
{
  const int64_t stride = int64_t(gridDim.x) * blockDim.x * 2;
  for (int64_t i = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) * 2; i + 1 < n; i += stride)
  {
    double2_a16 v;
    v.x = double(i), v.y = 0.0;
    *reinterpret_cast<double2_a16*>(out + i) = v;
  }
}

// rows of ROW doubles, a wave takes RB consecutive rows; MODE 0: two 8-byte stores per lane (lane, lane + 64),
// 1: one 16-byte store of entries 2 lane, 2 lane + 1 (8-byte aligned), 2: the same, nontemporal
template <int MODE>
__device__ __forceinline__ void write_row(double* row, int lane, double x)
{
  if constexpr (MODE == 0)
  {
    row[lane] = x;
    if (lane + 64 < ROW)
      row[lane + 64] = 0.0;
  }
  else
  {
    const int e = 2 * lane;
    if (e + 1 < ROW)
    {
      double2_a8 v;
      v.x = x, v.y = 0.0;
      if constexpr (MODE == 2)
        __builtin_nontemporal_store(v, reinterpret_cast<double2_a8*>(row + e));
      else
        *reinterpret_cast<double2_a8*>(row + e) = v;
    }
    else if (e < ROW)
    {
      if constexpr (MODE == 2)
        __builtin_nontemporal_store(x, row + e);
      else
        row[e] = x;
    }
  }
}

// B: one wave = 4 consecutive rows, waves in row order over the whole array (256-thread workgroups, many per CU)
template <int MODE>
__global__ void k_rows_stream(double* out, int64_t nrows)
{
  const int64_t wave = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int u = 0; u < 4; ++u)
  {
    const int64_t r = wave * 4 + u;
    if (r < nrows)
      write_row<MODE>(out + r * ROW, lane, double(r));
  }
}

// D: the node-block shape: a workgroup of 1024 threads owns a chunk of CHUNK_ROWS rows; LDS keeps one workgroup per CU.
// ORDER 0: waves interleaved (wave w takes rows 4 w + 64 trip ..), 1: every wave a contiguous range of rows
template <int MODE, int ORDER>
__global__ void __launch_bounds__(1024) k_rows_chunk(double* out, int nchunks, int lds_doubles)
{
  extern __shared__ double s[];
  const int per = (nchunks + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (b >= nchunks)
    return;
  for (int i = threadIdx.x; i < lds_doubles; i += blockDim.x)
    s[i] = double(i);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  double* base = out + int64_t(b) * CHUNK;
  if constexpr (ORDER == 0)
  {
    for (int rb = wave * 4; rb < CHUNK_ROWS; rb += nw * 4)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (rb + u < CHUNK_ROWS)
          write_row<MODE>(base + int64_t(rb + u) * ROW, lane, s[(rb + u + lane) % lds_doubles]);
  }
  else
  {
    const int per_w = (CHUNK_ROWS + nw - 1) / nw;
    for (int r = wave * per_w; r < (wave + 1) * per_w && r < CHUNK_ROWS; ++r)
      write_row<MODE>(base + int64_t(r) * ROW, lane, s[(r + lane) % lds_doubles]);
  }
}

// E: the chunk as a flat stream: aligned 16-byte stores, all lanes, wave w takes KB (w + 16 trip) of the chunk
template <int NT_STORE>
__global__ void __launch_bounds__(1024) k_flat_chunk(double* out, int nchunks, int lds_doubles)
{
  extern __shared__ double s[];
  const int per = (nchunks + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (b >= nchunks)
    return;
  for (int i = threadIdx.x; i < lds_doubles; i += blockDim.x)
    s[i] = double(i);
  __syncthreads();
  double* base = out + int64_t(b) * CHUNK; // (CHUNK is even: 16-byte aligned)
  for (int64_t i = int64_t(threadIdx.x) * 2; i + 1 < CHUNK; i += int64_t(blockDim.x) * 2)
  {
    double2_a16 v;
    v.x = s[i % lds_doubles], v.y = 0.0;
    if constexpr (NT_STORE)
      __builtin_nontemporal_store(v, reinterpret_cast<double2_a16*>(base + i));
    else
      *reinterpret_cast<double2_a16*>(base + i) = v;
  }
}

#define CHECK(x)                                                                                                         \
  do                                                                                                                     \
  {                                                                                                                      \
    hipError_t e_ = (x);                                                                                                 \
    if (e_ != hipSuccess)                                                                                                \
    {                                                                                                                    \
      std::printf("%s: %s\n", #x, hipGetErrorString(e_));                                                                \
      std::exit(1);                                                                                                      \
    }                                                                                                                    \
  } while (0)

template <class F>
static void timeit(const char* name, double bytes, F launch)
{
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  launch();
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep)
  {
    CHECK(hipEventRecord(a));
    launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    best = ms < best ? ms : best;
  }
  CHECK(hipGetLastError());
  std::printf("%-58s %8.3f ms  %6.2f TB/s\n", name, best, bytes / best / 1e9);
}

int main()
{
  const int nchunks = 59000;
  const int64_t n = CHUNK * nchunks; // doubles (34.8 GB)
  const int64_t nrows = int64_t(CHUNK_ROWS) * nchunks;
  double* out;
  CHECK(hipMalloc(&out, n * 8));
  const double bytes = double(n) * 8;
  timeit("hipMemsetAsync", bytes, [&] { CHECK(hipMemsetAsync(out, 0, n * 8)); });
  timeit("A flat, 16 B aligned per lane, grid-stride", bytes, [&] { hipLaunchKernelGGL(k_flat, dim3(256 * 32), dim3(256), 0, 0, out, n); });
  const unsigned gs = unsigned((nrows / 4 * 64 + 255) / 256);
  timeit("B rows, wave = 4 rows in order, 8 B stores", bytes, [&] { hipLaunchKernelGGL(k_rows_stream<0>, dim3(gs), dim3(256), 0, 0, out, nrows); });
  timeit("B rows, wave = 4 rows in order, 16 B stores", bytes, [&] { hipLaunchKernelGGL(k_rows_stream<1>, dim3(gs), dim3(256), 0, 0, out, nrows); });
  timeit("B rows, wave = 4 rows in order, 16 B nontemporal", bytes, [&] { hipLaunchKernelGGL(k_rows_stream<2>, dim3(gs), dim3(256), 0, 0, out, nrows); });
  const unsigned gc = 8u * unsigned((nchunks + 7) / 8);
  for (int lds_kb : {72, 36})
  {
    const int ld = lds_kb * 128;
    const size_t lds = size_t(ld) * 8;
    std::printf("-- chunks of %d rows, one workgroup of 1024 threads each, %d KB of LDS (%s per CU)\n", CHUNK_ROWS, lds_kb,
                lds_kb > 80 ? "one" : (lds_kb > 53 ? "two by LDS, one by waves" : "more"));
    auto attr = [&](auto k) { CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds))); };
    attr(k_rows_chunk<0, 0>), attr(k_rows_chunk<1, 0>), attr(k_rows_chunk<2, 0>), attr(k_rows_chunk<1, 1>), attr(k_flat_chunk<0>), attr(k_flat_chunk<1>);
    timeit("D chunk, waves interleaved, 8 B stores", bytes, [&] { hipLaunchKernelGGL((k_rows_chunk<0, 0>), dim3(gc), dim3(1024), lds, 0, out, nchunks, ld); });
    timeit("D chunk, waves interleaved, 16 B stores", bytes, [&] { hipLaunchKernelGGL((k_rows_chunk<1, 0>), dim3(gc), dim3(1024), lds, 0, out, nchunks, ld); });
    timeit("D chunk, waves interleaved, 16 B nontemporal", bytes, [&] { hipLaunchKernelGGL((k_rows_chunk<2, 0>), dim3(gc), dim3(1024), lds, 0, out, nchunks, ld); });
    timeit("D chunk, a contiguous range per wave, 16 B stores", bytes, [&] { hipLaunchKernelGGL((k_rows_chunk<1, 1>), dim3(gc), dim3(1024), lds, 0, out, nchunks, ld); });
    timeit("E chunk as a flat stream, aligned 16 B, all lanes", bytes, [&] { hipLaunchKernelGGL(k_flat_chunk<0>, dim3(gc), dim3(1024), lds, 0, out, nchunks, ld); });
    timeit("E chunk as a flat stream, nontemporal", bytes, [&] { hipLaunchKernelGGL(k_flat_chunk<1>, dim3(gc), dim3(1024), lds, 0, out, nchunks, ld); });
    for (int th : {512, 256})
    {
      char nm[96];
      std::snprintf(nm, sizeof nm, "D chunk, waves interleaved, 16 B stores, %d threads", th);
      timeit(nm, bytes, [&] { hipLaunchKernelGGL((k_rows_chunk<1, 0>), dim3(gc), dim3(th), lds, 0, out, nchunks, ld); });
      std::snprintf(nm, sizeof nm, "E chunk as a flat stream, %d threads", th);
      timeit(nm, bytes, [&] { hipLaunchKernelGGL(k_flat_chunk<0>, dim3(gc), dim3(th), lds, 0, out, nchunks, ld); });
    }
  }
  CHECK(hipFree(out));
  return 0;
}
