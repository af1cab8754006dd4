// Shared by the two consumers of the C ABI (examples/mpcx_driver.cpp, examples/mpcx_driver_blocks.cpp): the MPCX1 array-bundle
// files, a device arena over hipMalloc, kernel descriptors from a bundle.  Host C++ only, no Python, no torch.
#pragma once
#include "mpcx.h"

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <hip/hip_runtime.h>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace
{
struct Array
{
  int32_t dtype = 0;
  int64_t n = 0;
  std::vector<char> bytes;
  template <typename T>
  const T* as() const
  {
    return reinterpret_cast<const T*>(bytes.data());
  }
};
using Bundle = std::map<std::string, Array>;
constexpr size_t DTYPE_SIZE[4] = {1, 4, 8, 8};

Bundle read_bundle(const char* path)
{
  FILE* f = std::fopen(path, "rb");
  if (!f)
    throw std::runtime_error(std::string("cannot open ") + path);
  char magic[8];
  int64_t count = 0;
  if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "MPCX1\0\0\0", 8) != 0 || std::fread(&count, 8, 1, f) != 1)
    throw std::runtime_error("not an MPCX1 file");
  Bundle out;
  for (int64_t k = 0; k < count; ++k)
  {
    char name[32];
    Array a;
    if (std::fread(name, 1, 32, f) != 32 || std::fread(&a.dtype, 4, 1, f) != 1 || std::fread(&a.n, 8, 1, f) != 1 || a.dtype < 0
        || a.dtype > 3)
      throw std::runtime_error("truncated header");
    a.bytes.resize(size_t(a.n) * DTYPE_SIZE[a.dtype]);
    if (!a.bytes.empty() && std::fread(a.bytes.data(), 1, a.bytes.size(), f) != a.bytes.size())
      throw std::runtime_error("truncated data");
    name[31] = 0;
    out[name] = std::move(a);
  }
  std::fclose(f);
  return out;
}
void write_bundle(const char* path, const std::vector<std::pair<std::string, Array>>& arrays)
{
  FILE* f = std::fopen(path, "wb");
  if (!f)
    throw std::runtime_error(std::string("cannot write ") + path);
  const int64_t count = int64_t(arrays.size());
  std::fwrite("MPCX1\0\0\0", 1, 8, f);
  std::fwrite(&count, 8, 1, f);
  for (const auto& [name, a] : arrays)
  {
    char nm[32] = {0};
    std::strncpy(nm, name.c_str(), 31);
    std::fwrite(nm, 1, 32, f);
    std::fwrite(&a.dtype, 4, 1, f);
    std::fwrite(&a.n, 8, 1, f);
    if (!a.bytes.empty())
      std::fwrite(a.bytes.data(), 1, a.bytes.size(), f);
  }
  std::fclose(f);
}
template <typename T>
Array make_array(int32_t dtype, const std::vector<T>& v)
{
  Array a;
  a.dtype = dtype, a.n = int64_t(v.size());
  a.bytes.resize(v.size() * sizeof(T));
  if (!v.empty())
    std::memcpy(a.bytes.data(), v.data(), a.bytes.size());
  return a;
}

void hip_check(hipError_t e, const char* what)
{
  if (e != hipSuccess)
    throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void mpcx_check(int rc, const char* what)
{
  if (rc != 0)
    throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + mpcx_last_error());
}
// device memory with the lifetime of the driver
struct DeviceArena
{
  std::vector<void*> blocks;
  ~DeviceArena()
  {
    for (void* p : blocks)
      (void)hipFree(p);
  }
  template <typename T>
  T* alloc(size_t n)
  {
    void* p = nullptr;
    hip_check(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)), "hipMalloc");
    blocks.push_back(p);
    return static_cast<T*>(p);
  }
  template <typename T>
  T* upload(const T* host, size_t n)
  {
    T* p = alloc<T>(n);
    if (n)
      hip_check(hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy H2D");
    return p;
  }
  template <typename T>
  T* upload(const std::vector<T>& v)
  {
    return upload(v.data(), v.size());
  }
};

const Array& need(const Bundle& b, const char* name)
{
  auto it = b.find(name);
  if (it == b.end())
    throw std::runtime_error(std::string("problem file: array '") + name + "' is missing");
  return it->second;
}
// kernel descriptor arrays "<p>_kernel" int32[9] = form, celltype, degree, bs, degree1, bs1, fn_id, coeff_degree, nq;
// "<p>_qpts" / "<p>_qwts"; optional "<p>_constants"
mpcx_kernel_t make_kernel(const Bundle& in, const std::string& p, DeviceArena& dev, const double** constants)
{
  const Array& k = need(in, (p + "_kernel").c_str());
  if (k.n != 9)
    throw std::runtime_error("kernel descriptor: 9 integers expected");
  const int32_t* v = k.as<int32_t>();
  mpcx_kernel_t K;
  std::memset(&K, 0, sizeof(K));
  K.form = v[0], K.celltype = v[1], K.degree = v[2], K.bs = v[3], K.degree1 = v[4], K.bs1 = v[5], K.fn_id = v[6];
  K.coeff_degree = v[7], K.nq = v[8];
  const Array& qp = need(in, (p + "_qpts").c_str());
  const Array& qw = need(in, (p + "_qwts").c_str());
  K.qpts = dev.upload(qp.as<double>(), size_t(qp.n));
  K.qwts = dev.upload(qw.as<double>(), size_t(qw.n));
  *constants = nullptr;
  auto it = in.find(p + "_constants");
  if (it != in.end() && it->second.n > 0)
    *constants = dev.upload(it->second.as<double>(), size_t(it->second.n));
  return K;
}
double seconds_since(std::chrono::steady_clock::time_point t0)
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
} // namespace

