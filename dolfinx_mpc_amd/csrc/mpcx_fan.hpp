// The six-tet cluster ("Kuhn fan") in the local numbering every cluster kernel uses, as compile-time tables, and the
// per-slot record formats.  Shared by csrc/mpcx_cubes.hip (built-in operators) and, as embedded text, by the hipRTC
// translation unit of the imported (UFCx) kernels (csrc/mpcx_ufcx.cpp): no includes, fixed-width types come from the
// includer.  Cube corner b: bit 0 = x, bit 1 = y, bit 2 = z; the shared edge is (0, 7), the ring 1-3-2-6-4-5.
#pragma once

namespace mpcx_fan
{
// local vertices of the six tets of a fan
__host__ __device__ constexpr int fan_vertex(int t, int i)
{
  constexpr int T[6][4] = {{0, 1, 3, 7}, {0, 1, 7, 5}, {0, 5, 7, 4}, {0, 3, 2, 7}, {0, 6, 4, 7}, {0, 2, 6, 7}};
  return T[t][i];
}
// the fan walked round its shared edge: consecutive tets share a face
__host__ __device__ constexpr int fan_order(int step)
{
  constexpr int O[6] = {0, 1, 2, 4, 5, 3};
  return O[step];
}
// do local vertices a and b share a tet? (a == b counts)
__host__ __device__ constexpr bool fan_coupled(int a, int b)
{
  for (int t = 0; t < 6; ++t)
  {
    bool ha = false, hb = false;
    for (int i = 0; i < 4; ++i)
    {
      ha |= fan_vertex(t, i) == a;
      hb |= fan_vertex(t, i) == b;
    }
    if (ha && hb)
      return true;
  }
  return false;
}
// last step (in fan_order) whose tet holds both a and b; -1 if they share none
__host__ __device__ constexpr int fan_last_step(int a, int b)
{
  int last = -1;
  for (int s = 0; s < 6; ++s)
  {
    const int t = fan_order(s);
    bool ha = false, hb = false;
    for (int i = 0; i < 4; ++i)
    {
      ha |= fan_vertex(t, i) == a;
      hb |= fan_vertex(t, i) == b;
    }
    if (ha && hb)
      last = s;
  }
  return last;
}
// position of the coupled pair (a, b) among the 46 coupled pairs in row-major order, -1 if a and b share no tet
__host__ __device__ constexpr int fan_pair_index(int a, int b)
{
  if (!fan_coupled(a, b))
    return -1;
  int n = 0;
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j)
    {
      if (i == a && j == b)
        return n;
      if (fan_coupled(i, j))
        ++n;
    }
  return -1;
}

struct __attribute__((aligned(16))) CubeRec
{
  int32_t v[8];    // vertex (= dof) ids with the Dirichlet / slave mask in bit 28
  uint8_t off[64]; // off[a*8+b]: position of column v[b] inside CSR row v[a] (coupled pairs only)
};
static_assert(sizeof(CubeRec) == 96, "record layout");
// Narrow record: rows of at most 16 entries before any of the cluster's columns (every interior row of a Kuhn mesh has
// 15 entries) need 4 bits per offset: 8 ids + 46 nibbles = 55 bytes -> 64-byte records, four 16-byte loads per slot
// instead of six and a third less plan memory.  Row blocks that hold a fat row (master rows of a constraint) keep the
// 96-byte format; the two kinds are launched separately.
struct __attribute__((aligned(16))) CubeRecNarrow
{
  int32_t v[8];
  uint8_t nib[32]; // nibble p = fan_pair_index(a, b): byte p / 2, low half for even p
};
static_assert(sizeof(CubeRecNarrow) == 64, "record layout");
} // namespace mpcx_fan
