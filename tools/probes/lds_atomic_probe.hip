// LDS scatter-add throughput on gfx950: what bounds the row-block matrix kernels' scatter.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/probes/lds_atomic_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
// Compares ds_add_f64 with a plain read-add-write (legal when a colouring makes the lanes' targets disjoint),
// ds_add_f32, and 16-byte read-add-write of two neighbouring entries.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int NENT = 4608; // entries of one half-size row block
constexpr int ITER = 4096;

// ADDR 0: lane l adds to entry (base + l) (conflict-free, contiguous)
// ADDR 1: pseudo-random entry per lane and iteration
// ADDR 2: pseudo-random entry shared by groups of 6 neighbouring lanes (the 6 tets round a cube diagonal)
// ADDR 3: lane l owns a random row start, iteration walks 12 contiguous entries (one element-tensor row)
// OP 0: ds_add_f64   1: read + add + write f64   2: ds_add_f32   3: 16-byte read + 2 adds + 16-byte write
template <int ADDR, int OP>
__global__ void __launch_bounds__(512) probe(double* out)
{
  __shared__ double s[NENT];
  for (int i = threadIdx.x; i < NENT; i += blockDim.x)
    s[i] = 0.0;
  __syncthreads();
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  const unsigned g = (threadIdx.x / 6) * 2654435761u + blockIdx.x * 40503u + 999u;
  unsigned y = g;
  int row = 0;
  for (int it = 0; it < ITER; ++it)
  {
    int idx;
    if (ADDR == 0)
      idx = (it * 64 + threadIdx.x) % NENT;
    else if (ADDR == 1)
    {
      x = x * 1664525u + 1013904223u;
      idx = (x >> 8) % NENT;
    }
    else if (ADDR == 2)
    {
      y = y * 1664525u + 1013904223u;
      idx = (y >> 8) % NENT;
    }
    else
    {
      if (it % 12 == 0)
      {
        x = x * 1664525u + 1013904223u;
        row = (x >> 8) % (NENT - 12);
      }
      idx = row + it % 12;
    }
    if (OP == 0)
      __hip_atomic_fetch_add(&s[idx], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (OP == 1)
    {
      s[idx] += 1.0;
      asm volatile("" ::: "memory");
    }
    else if (OP == 2)
      __hip_atomic_fetch_add(reinterpret_cast<float*>(s) + idx, 1.0f, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WORKGROUP);
    else
    {
      typedef double d2 __attribute__((ext_vector_type(2)));
      d2* p = reinterpret_cast<d2*>(s) + (idx >> 1);
      d2 v = *p;
      v.x += 1.0;
      v.y += 1.0;
      *p = v;
      asm volatile("" ::: "memory");
    }
  }
  __syncthreads();
  if (threadIdx.x == 0)
    out[blockIdx.x] = s[0] + s[NENT - 1];
}

template <int ADDR, int OP>
double run(double* d_out, int blocks, int threads)
{
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  probe<ADDR, OP><<<blocks, threads>>>(d_out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  probe<ADDR, OP><<<blocks, threads>>>(d_out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return double(blocks) * threads * ITER / (ms * 1e-3) / 1e9; // G lane-updates per second
}

template <int OP>
void row(const char* name, double* d_out, int blocks, int threads)
{
  printf("  %-28s %8.1f %8.1f %8.1f %8.1f\n", name, run<0, OP>(d_out, blocks, threads),
         run<1, OP>(d_out, blocks, threads), run<2, OP>(d_out, blocks, threads), run<3, OP>(d_out, blocks, threads));
}

int main()
{
  const int blocks = 256 * 4 * 4; // 4 workgroups per CU resident, 4 rounds
  double* d_out;
  hipMalloc(&d_out, blocks * sizeof(double));
  for (int threads : {512, 256, 128})
  {
    printf("%d-thread workgroups, %d-entry array, G lane-updates/s on the whole GPU\n", threads, NENT);
    printf("  %-28s %8s %8s %8s %8s\n", "", "contig", "random", "rand/6", "rows12");
    row<0>("ds_add_f64", d_out, blocks, threads);
    row<1>("read+add+write f64", d_out, blocks, threads);
    row<2>("ds_add_f32", d_out, blocks, threads);
    row<3>("16B read+2 adds+16B write", d_out, blocks, threads);
  }
  return 0;
}
