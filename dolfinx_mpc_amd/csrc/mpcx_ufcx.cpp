// UFCx import path: the reference's element seam is a C function pointer with the
// UFCx signature
//     void tabulate_tensor(T* A, const T* w, const T* c, const U* coordinate_dofs,
//                          const int* entity_local_index, const uint8_t* quadrature_permutation, void* custom_data)
// (cpp/assemble_matrix.cpp:291-292, 438-439; numba calls it the same way, numba/assemble_matrix.py:282-290).
// A host pointer cannot be called from a kernel, so the seam here is the SOURCE of such a function: it is
// compiled for gfx950 at run time with hipRTC as a __device__ function, together with generic per-entity
// assembly kernels (one thread per entity, element tensor in private memory, Dirichlet masking, the
// K^T A_e K elimination of cpp/assemble_matrix.cpp:99-268, CSR search + device atomics) specialised for the
// element shape through -D options.  Any form FFCx can generate for float64 can be assembled this way; the
// built-in operators remain the fast path (LDS row blocks, closed-form entries).
#include "mpcx.h"
#include "mpcx_internal.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <mutex>
#include <string>
#include <vector>

namespace
{
// include/mpcx.h as text (the kernels take the very same argument structs by value)
const char* const MPCX_H_TEXT =
#include "mpcx_h_embed.inc"
    ;

const char* const KERNELS_TEXT = R"MPCXK(
#define N0 (ND0 * BS0)
#define N1 (ND1 * BS1)

__device__ inline void atomic_add_f64(double* p, double v)
{
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline long long csr_find(const int* __restrict__ cols, long long lo, long long hi, int col)
{
  const long long end = hi;
  while (lo < hi)
  {
    const long long mid = (lo + hi) >> 1;
    if (cols[mid] < col)
      lo = mid + 1;
    else
      hi = mid;
  }
  return (lo < end && cols[lo] == col) ? lo : -1;
}
__device__ inline void gather(const double* __restrict__ x, const int* __restrict__ xd, long long cell, double* cd)
{
  for (int i = 0; i < NV; ++i)
  {
    const long long v = xd[cell * NV + i];
    for (int k = 0; k < 3; ++k)
      cd[3 * i + k] = x[3 * v + k];
  }
}
// element tensor of entity e through the imported function (caller-zeroed, accumulated into: cpp/assemble_matrix.cpp:504)
__device__ inline void tabulate(double* Ae, int n, const double* coeffs, int cstride, const double* constants,
                                const double* cd, long long e, int lf)
{
  for (int i = 0; i < n; ++i)
    Ae[i] = 0.0;
  const unsigned char perm = 0;
  UFCX_FN(Ae, coeffs ? coeffs + e * cstride : (const double*)0, constants, cd, &lf, &perm, (void*)0);
}

#if UFCX_RANK == 2
extern "C" __global__ void __launch_bounds__(64) ufcx_matrix_kernel(mpcx_matrix_args_t a)
{
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.n_entities)
    return;
  const long long l = e * a.estride;
  const long long cell = a.entities ? a.entities[l] : e;
  const long long cell0 = a.entities0 ? a.entities0[l] : e;
  const long long cell1 = a.entities1 ? a.entities1[l] : e;
  const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
  double cd[NV * 3];
  gather(a.x, a.x_dofmap, cell, cd);
  double Ae[N0 * N1];
  tabulate(Ae, N0 * N1, a.coeffs, a.cstride, a.constants, cd, e, lf);
  // bulk part: Dirichlet and slave rows / columns masked (cpp/assemble_matrix.cpp:510-533, 165-178)
  for (int p = 0; p < N0; ++p)
  {
    const int r = a.dofmap0[cell0 * ND0 + p / BS0] * BS0 + p % BS0;
    if ((a.bc0 && a.bc0[r]) || a.mpc0.is_slave[r])
      continue;
    const long long lo = a.rowptr[r], hi = a.rowptr[r + 1];
    for (int q = 0; q < N1; ++q)
    {
      const int c = a.dofmap1[cell1 * ND1 + q / BS1] * BS1 + q % BS1;
      if ((a.bc1 && a.bc1[c]) || a.mpc1.is_slave[c])
        continue;
      const long long pos = csr_find(a.cols, lo, hi, c);
      if (pos >= 0)
        atomic_add_f64(a.vals + pos, Ae[p * N1 + q]);
    }
  }
}

// master contributions of the slave entities: cpp/assemble_matrix.cpp:182-267
extern "C" __global__ void __launch_bounds__(64) ufcx_matrix_mpc_kernel(mpcx_matrix_args_t a)
{
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.n_slave_entities)
    return;
  const long long e = a.slave_entities[t];
  const long long l = e * a.estride;
  const long long cell = a.entities ? a.entities[l] : e;
  const long long cell0 = a.entities0 ? a.entities0[l] : e;
  const long long cell1 = a.entities1 ? a.entities1[l] : e;
  const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
  double cd[NV * 3];
  gather(a.x, a.x_dofmap, cell, cd);
  double Ae[N0 * N1];
  tabulate(Ae, N0 * N1, a.coeffs, a.cstride, a.constants, cd, e, lf);
  int rows[N0], colsd[N1];
  bool rbc[N0], cbc[N1], rsl[N0], csl[N1];
  for (int p = 0; p < N0; ++p)
  {
    const int r = a.dofmap0[cell0 * ND0 + p / BS0] * BS0 + p % BS0;
    rows[p] = r;
    rbc[p] = a.bc0 && a.bc0[r];
    rsl[p] = a.mpc0.is_slave[r];
  }
  for (int q = 0; q < N1; ++q)
  {
    const int c = a.dofmap1[cell1 * ND1 + q / BS1] * BS1 + q % BS1;
    colsd[q] = c;
    cbc[q] = a.bc1 && a.bc1[c];
    csl[q] = a.mpc1.is_slave[c];
  }
  for (int p = 0; p < N0; ++p)
  {
    if (!rsl[p])
      continue;
    for (int mi = a.mpc0.masters_offsets[rows[p]]; mi < a.mpc0.masters_offsets[rows[p] + 1]; ++mi)
    {
      const int m = a.mpc0.masters[mi];
      const double ci = a.mpc0.coeffs[mi];
      const long long lo = a.rowptr[m], hi = a.rowptr[m + 1];
      for (int q = 0; q < N1; ++q)
      {
        const double v = (rbc[p] || cbc[q]) ? 0.0 : Ae[p * N1 + q];
        if (csl[q])
        {
          for (int mj = a.mpc1.masters_offsets[colsd[q]]; mj < a.mpc1.masters_offsets[colsd[q] + 1]; ++mj)
          {
            const long long pos = csr_find(a.cols, lo, hi, a.mpc1.masters[mj]);
            if (pos >= 0)
              atomic_add_f64(a.vals + pos, ci * a.mpc1.coeffs[mj] * v);
          }
        }
        else if (!cbc[q])
        {
          const long long pos = csr_find(a.cols, lo, hi, colsd[q]);
          if (pos >= 0)
            atomic_add_f64(a.vals + pos, ci * v);
        }
      }
    }
  }
  for (int q = 0; q < N1; ++q)
  {
    if (!csl[q])
      continue;
    for (int mj = a.mpc1.masters_offsets[colsd[q]]; mj < a.mpc1.masters_offsets[colsd[q] + 1]; ++mj)
    {
      const int m = a.mpc1.masters[mj];
      const double cj = a.mpc1.coeffs[mj];
      for (int p = 0; p < N0; ++p)
      {
        if (rsl[p] || rbc[p])
          continue;
        const long long pos = csr_find(a.cols, a.rowptr[rows[p]], a.rowptr[rows[p] + 1], m);
        if (pos >= 0)
          atomic_add_f64(a.vals + pos, cj * (cbc[q] ? 0.0 : Ae[p * N1 + q]));
      }
    }
  }
}

// cpp/lifting.h:77-133: raw element tensor, b -= scale * Ae[:, j] (g_j - x0_j), slaves moved to their masters
extern "C" __global__ void __launch_bounds__(64) ufcx_lifting_kernel(mpcx_lifting_args_t a)
{
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.n_lift_entities)
    return;
  const long long e = a.lift_entities[t];
  const long long l = e * a.estride;
  const long long cell = a.entities ? a.entities[l] : e;
  const long long cell0 = a.entities0 ? a.entities0[l] : e;
  const long long cell1 = a.entities1 ? a.entities1[l] : e;
  const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
  double cd[NV * 3];
  gather(a.x, a.x_dofmap, cell, cd);
  double Ae[N0 * N1];
  tabulate(Ae, N0 * N1, a.coeffs, a.cstride, a.constants, cd, e, lf);
  double be[N0];
  for (int m = 0; m < N0; ++m)
    be[m] = 0.0;
  for (int q = 0; q < N1; ++q)
  {
    const int jj = a.dofmap1[cell1 * ND1 + q / BS1] * BS1 + q % BS1;
    if (a.bc_markers1[jj])
    {
      const double g = a.scale * (a.bc_values1[jj] - (a.x0 ? a.x0[jj] : 0.0));
      for (int m = 0; m < N0; ++m)
        be[m] -= Ae[m * N1 + q] * g;
    }
  }
  for (int p = 0; p < N0; ++p)
  {
    const int d = a.dofmap0[cell0 * ND0 + p / BS0] * BS0 + p % BS0;
    double v = be[p];
    if (a.mpc0.is_slave[d])
    {
      const int m0 = a.mpc0.masters_offsets[d], m1 = a.mpc0.masters_offsets[d + 1];
      for (int mi = m0; mi < m1; ++mi)
        atomic_add_f64(a.b + a.mpc0.masters[mi], a.mpc0.coeffs[mi] * v);
      if (m1 > m0)
        v = 0.0;
    }
    if (v != 0.0)
      atomic_add_f64(a.b + d, v);
  }
}
#else
// cpp/assemble_vector.cpp:65-90 + modify_mpc_vec (cpp/assemble_vector.h:35-69)
extern "C" __global__ void __launch_bounds__(64) ufcx_vector_kernel(mpcx_vector_args_t a)
{
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.n_entities)
    return;
  const long long l = e * a.estride;
  const long long cell = a.entities ? a.entities[l] : e;
  const long long cell0 = a.entities0 ? a.entities0[l] : e;
  const int lf = a.estride == 2 ? a.entities[l + 1] : 0;
  double cd[NV * 3];
  gather(a.x, a.x_dofmap, cell, cd);
  double be[N0];
  tabulate(be, N0, a.coeffs, a.cstride, a.constants, cd, e, lf);
  for (int p = 0; p < N0; ++p)
  {
    const int d = a.dofmap[cell0 * ND0 + p / BS0] * BS0 + p % BS0;
    double v = be[p];
    if (a.mpc.is_slave[d])
    {
      const int m0 = a.mpc.masters_offsets[d], m1 = a.mpc.masters_offsets[d + 1];
      for (int mi = m0; mi < m1; ++mi)
        atomic_add_f64(a.b + a.mpc.masters[mi], a.mpc.coeffs[mi] * v);
      if (m1 > m0)
        v = 0.0;
    }
    if (v != 0.0)
      atomic_add_f64(a.b + d, v);
  }
}
#endif
)MPCXK";

struct Rtc
{
  void* lib = nullptr;
  int (*create)(void**, const char*, const char*, int, const char**, const char**) = nullptr;
  int (*compile)(void*, int, const char**) = nullptr;
  int (*log_size)(void*, size_t*) = nullptr;
  int (*log)(void*, char*) = nullptr;
  int (*code_size)(void*, size_t*) = nullptr;
  int (*code)(void*, char*) = nullptr;
  int (*destroy)(void**) = nullptr;
};

Rtc& rtc()
{
  static Rtc r;
  static std::once_flag once;
  std::call_once(once,
                 []
                 {
                   // the copy that belongs to the HIP runtime already in the process (torch ships one), else ROCm's
                   for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"})
                     if ((r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL)))
                       break;
                   if (!r.lib)
                     return;
                   auto sym = [&](const char* n) { return dlsym(r.lib, n); };
                   r.create = reinterpret_cast<decltype(r.create)>(sym("hiprtcCreateProgram"));
                   r.compile = reinterpret_cast<decltype(r.compile)>(sym("hiprtcCompileProgram"));
                   r.log_size = reinterpret_cast<decltype(r.log_size)>(sym("hiprtcGetProgramLogSize"));
                   r.log = reinterpret_cast<decltype(r.log)>(sym("hiprtcGetProgramLog"));
                   r.code_size = reinterpret_cast<decltype(r.code_size)>(sym("hiprtcGetCodeSize"));
                   r.code = reinterpret_cast<decltype(r.code)>(sym("hiprtcGetCode"));
                   r.destroy = reinterpret_cast<decltype(r.destroy)>(sym("hiprtcDestroyProgram"));
                 });
  return r;
}

struct UfcxKernel
{
  mpcx_ufcx_desc_t desc{};
  std::vector<char> code; // gfx950 code object
  hipModule_t module = nullptr;
  hipFunction_t matrix = nullptr, matrix_mpc = nullptr, lifting = nullptr, vector = nullptr;
};

int hip_check(hipError_t err, const char* what)
{
  if (err != hipSuccess)
  {
    mpcx_set_error(std::string(what) + ": " + hipGetErrorString(err));
    return -100;
  }
  return 0;
}

// the code object is loaded on first launch: compiling needs no device (hipRTC cross-compiles)
int ensure_loaded(UfcxKernel* k)
{
  if (k->module)
    return 0;
  if (int rc = hip_check(hipModuleLoadData(&k->module, k->code.data()), "hipModuleLoadData"))
    return rc;
  if (k->desc.rank == 2)
  {
    if (int rc = hip_check(hipModuleGetFunction(&k->matrix, k->module, "ufcx_matrix_kernel"), "hipModuleGetFunction"))
      return rc;
    if (int rc = hip_check(hipModuleGetFunction(&k->matrix_mpc, k->module, "ufcx_matrix_mpc_kernel"), "hipModuleGetFunction"))
      return rc;
    return hip_check(hipModuleGetFunction(&k->lifting, k->module, "ufcx_lifting_kernel"), "hipModuleGetFunction");
  }
  return hip_check(hipModuleGetFunction(&k->vector, k->module, "ufcx_vector_kernel"), "hipModuleGetFunction");
}

template <class Args>
int launch(hipFunction_t f, int64_t n, const Args& a, void* stream)
{
  if (n == 0)
    return 0;
  Args copy = a;
  void* params[] = {&copy};
  const unsigned grid = static_cast<unsigned>((n + 63) / 64);
  return hip_check(hipModuleLaunchKernel(f, grid, 1, 1, 64, 1, 1, 0, static_cast<hipStream_t>(stream), params, nullptr),
                   "hipModuleLaunchKernel");
}
} // namespace

extern "C" void* mpcx_ufcx_compile(const mpcx_ufcx_desc_t* d)
{
  Rtc& r = rtc();
  if (!r.lib || !r.create || !r.compile || !r.code)
  {
    mpcx_set_error("mpcx_ufcx_compile: libhiprtc.so not found");
    return nullptr;
  }
  if (!d->source || !d->function_name || (d->rank != 1 && d->rank != 2) || d->nd0 <= 0 || d->bs0 <= 0 || d->nv <= 0
      || (d->rank == 2 && (d->nd1 <= 0 || d->bs1 <= 0)))
  {
    mpcx_set_error("mpcx_ufcx_compile: incomplete descriptor");
    return nullptr;
  }
  // translation unit: fixed-width types, the C-ABI structs, the imported C function as a __device__ function
  // (FFCx emits C99: `restrict`, plain functions -- force_cuda_host_device makes them callable from kernels),
  // then the assembly kernels
  std::string hdr(MPCX_H_TEXT);
  const std::string inc = "#include <stdint.h>";
  if (auto p = hdr.find(inc); p != std::string::npos)
    hdr.replace(p, inc.size(), "");
  std::string src = "typedef signed char int8_t;\ntypedef unsigned char uint8_t;\ntypedef unsigned short uint16_t;\n"
                    "typedef int int32_t;\ntypedef unsigned int uint32_t;\ntypedef long long int64_t;\n"
                    "typedef unsigned long long uint64_t;\n#define restrict __restrict__\n";
  src += hdr;
  src += "\n#pragma clang force_cuda_host_device begin\n";
  src += d->source;
  src += "\n#pragma clang force_cuda_host_device end\n";
  src += KERNELS_TEXT;
  void* prog = nullptr;
  if (r.create(&prog, src.c_str(), "mpcx_ufcx.hip", 0, nullptr, nullptr) != 0)
  {
    mpcx_set_error("mpcx_ufcx_compile: hiprtcCreateProgram failed");
    return nullptr;
  }
  std::vector<std::string> opts
      = {"--offload-arch=gfx950", "-O3", "-munsafe-fp-atomics", "-DUFCX_FN=" + std::string(d->function_name),
         "-DUFCX_RANK=" + std::to_string(d->rank), "-DND0=" + std::to_string(d->nd0), "-DBS0=" + std::to_string(d->bs0),
         "-DND1=" + std::to_string(d->rank == 2 ? d->nd1 : 1), "-DBS1=" + std::to_string(d->rank == 2 ? d->bs1 : 1),
         "-DNV=" + std::to_string(d->nv)};
  std::vector<const char*> copts;
  for (auto& o : opts)
    copts.push_back(o.c_str());
  const int rc = r.compile(prog, int(copts.size()), copts.data());
  if (rc != 0)
  {
    size_t n = 0;
    r.log_size(prog, &n);
    std::string log(n + 1, '\0');
    if (n)
      r.log(prog, log.data());
    r.destroy(&prog);
    mpcx_set_error("mpcx_ufcx_compile: hipRTC compilation failed:\n" + log.substr(0, 4000));
    return nullptr;
  }
  auto* k = new UfcxKernel;
  k->desc = *d;
  k->desc.source = nullptr;
  k->desc.function_name = nullptr;
  size_t n = 0;
  r.code_size(prog, &n);
  k->code.resize(n);
  r.code(prog, k->code.data());
  r.destroy(&prog);
  return k;
}

extern "C" int64_t mpcx_ufcx_code_size(void* handle) { return handle ? int64_t(static_cast<UfcxKernel*>(handle)->code.size()) : 0; }

extern "C" void mpcx_ufcx_free(void* handle)
{
  auto* k = static_cast<UfcxKernel*>(handle);
  if (!k)
    return;
  if (k->module)
    (void)hipModuleUnload(k->module);
  delete k;
}

namespace mpcx
{
int launch_matrix_ufcx(const mpcx_matrix_args_t& a)
{
  auto* k = static_cast<UfcxKernel*>(const_cast<void*>(a.kernel.ufcx));
  if (!k || k->desc.rank != 2 || a.nd0 != k->desc.nd0 || a.bs0 != k->desc.bs0 || a.nd1 != k->desc.nd1 || a.bs1 != k->desc.bs1
      || a.nv != k->desc.nv)
  {
    mpcx_set_error("mpcx_assemble_matrix: the imported kernel was compiled for other element shapes (or is not bilinear)");
    return -12;
  }
  if (a.algorithm == MPCX_ALG_ROWBLOCK || a.algorithm == MPCX_ALG_CUBE)
  {
    mpcx_set_error("mpcx_assemble_matrix: imported (UFCx) kernels are assembled with MPCX_ALG_ATOMIC");
    return -3;
  }
  if (int rc = ensure_loaded(k))
    return rc;
  if (int rc = launch(k->matrix, a.n_entities, a, a.stream))
    return rc;
  return launch(k->matrix_mpc, a.n_slave_entities, a, a.stream);
}
int launch_vector_ufcx(const mpcx_vector_args_t& a)
{
  auto* k = static_cast<UfcxKernel*>(const_cast<void*>(a.kernel.ufcx));
  if (!k || k->desc.rank != 1 || a.nd != k->desc.nd0 || a.bs != k->desc.bs0 || a.nv != k->desc.nv)
  {
    mpcx_set_error("mpcx_assemble_vector: the imported kernel was compiled for another element shape (or is not linear)");
    return -12;
  }
  if (int rc = ensure_loaded(k))
    return rc;
  return launch(k->vector, a.n_entities, a, a.stream);
}
int launch_lifting_ufcx(const mpcx_lifting_args_t& a)
{
  auto* k = static_cast<UfcxKernel*>(const_cast<void*>(a.kernel.ufcx));
  if (!k || k->desc.rank != 2 || a.nd0 != k->desc.nd0 || a.bs0 != k->desc.bs0 || a.nd1 != k->desc.nd1 || a.bs1 != k->desc.bs1
      || a.nv != k->desc.nv)
  {
    mpcx_set_error("mpcx_apply_lifting: the imported kernel was compiled for other element shapes (or is not bilinear)");
    return -12;
  }
  if (int rc = ensure_loaded(k))
    return rc;
  return launch(k->lifting, a.n_lift_entities, a, a.stream);
}
} // namespace mpcx
