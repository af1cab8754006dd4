// LDS ds_add_f64 throughput on gfx950: what bounds the row-block matrix kernel's scatter.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/probes/lds_atomic_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int NENT = 4608; // entries of one half-size row block
constexpr int ITER = 4096;

// mode 0: lane l adds to entry (base + l) (conflict-free, contiguous)
// mode 1: pseudo-random entry per lane and iteration
// mode 2: pseudo-random entry shared by groups of 6 neighbouring lanes (the 6 tets round a cube diagonal)
template <int MODE>
__global__ void __launch_bounds__(512) probe(double* out)
{
  __shared__ double s[NENT];
  for (int i = threadIdx.x; i < NENT; i += blockDim.x)
    s[i] = 0.0;
  __syncthreads();
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  const unsigned g = (threadIdx.x / 6) * 2654435761u + blockIdx.x * 40503u + 999u;
  unsigned y = g;
  for (int it = 0; it < ITER; ++it)
  {
    int idx;
    if (MODE == 0)
      idx = (it * 64 + threadIdx.x) % NENT;
    else if (MODE == 1)
    {
      x = x * 1664525u + 1013904223u;
      idx = (x >> 8) % NENT;
    }
    else
    {
      y = y * 1664525u + 1013904223u;
      idx = (y >> 8) % NENT;
    }
    __hip_atomic_fetch_add(&s[idx], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  if (threadIdx.x == 0)
    out[blockIdx.x] = s[0] + s[NENT - 1];
}

template <int MODE>
double run(double* d_out, int blocks)
{
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  probe<MODE><<<blocks, 512>>>(d_out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  probe<MODE><<<blocks, 512>>>(d_out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return double(blocks) * 512 * ITER / (ms * 1e-3) / 1e9; // G lane-atomics per second
}

int main()
{
  const int blocks = 256 * 4 * 4; // 4 workgroups per CU resident, 4 rounds
  double* d_out;
  hipMalloc(&d_out, blocks * sizeof(double));
  printf("ds_add_f64, 512-thread workgroups, %d-entry array (G lane-atomics/s, whole GPU)\n", NENT);
  printf("  contiguous      %.1f\n", run<0>(d_out, blocks));
  printf("  random          %.1f\n", run<1>(d_out, blocks));
  printf("  random, 6-shared %.1f\n", run<2>(d_out, blocks));
  return 0;
}
