// Cell clusters ("Kuhn fans") for P1 tetrahedral meshes on gfx950.
//
// Six tets that share one edge v0-v7 and together have eight vertices -- what every structured
// box generator (ours, DOLFINx create_box) emits per hexahedral cube -- couple only 8 + 2*19 = 46
// matrix entries, while the six element tensors scattered one by one cost 6*16 = 96 scatter-adds.
// One thread takes the whole cluster: 8 vertex ids and 24 coordinates are loaded once, the six
// element tensors are summed in registers (27 distinct values, the matrix is symmetric) and 46
// ds_add_f64 go to the LDS copy of the row block -- half the LDS atomics and a third of the index
// bytes of the per-cell kernel (cpp/assemble_matrix.cpp:488-547 is the loop being replaced; the
// per-cell semantics -- Dirichlet / slave rows and columns masked, values ADDed -- are unchanged).
// The clusters are found topologically at set-up (dolfinx_mpc_amd/clusters.py: six consecutive cells
// with the fan's vertex pattern); cells outside any cluster keep going through the per-cell kernels.
// The geometry of every tet is computed from its own coordinates: nothing assumes a cube.
#include "mpcx.h"
#include "mpcx_elements.hpp"
#include "mpcx_internal.h"

#include <cstdlib>
#include <hip/hip_runtime.h>
#include <string>
#include <type_traits>

#include "mpcx_box14.hpp"
#include "mpcx_fan.hpp"

namespace mpcx
{
namespace
{
constexpr int MASK_SHIFT = 28;
constexpr int DOF_MASK = (1 << MASK_SHIFT) - 1;

using namespace mpcx_fan; // fan_vertex, fan_order, fan_coupled, fan_last_step, fan_pair_index, CubeRec, CubeRecNarrow

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

__device__ inline int64_t find_col(const int32_t* __restrict__ cols, int64_t lo, int64_t hi, int col)
{
  const int64_t end = hi;
  while (lo < hi)
  {
    const int64_t mid = (lo + hi) >> 1;
    if (cols[mid] < col)
      lo = mid + 1;
    else
      hi = mid;
  }
  return (lo < end && cols[lo] == col) ? lo : -1;
}

// set-up: does the group of six consecutive cells g form a fan?  verts[g] = its eight vertices, ok[g] = 1 / 0
__global__ void cube_detect_kernel(const int32_t* __restrict__ cells, int64_t n_groups, int32_t* __restrict__ verts,
                                   int8_t* __restrict__ ok)
{
  const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= n_groups)
    return;
  int32_t c[6][4];
  const uint4* p = reinterpret_cast<const uint4*>(cells + g * 24);
#pragma unroll
  for (int t = 0; t < 6; ++t)
  {
    const uint4 w = p[t];
    c[t][0] = w.x, c[t][1] = w.y, c[t][2] = w.z, c[t][3] = w.w;
  }
  // vertex b of the fan read from the first tet that introduces it; every other occurrence must agree
  int32_t v[8];
  v[0] = c[0][0], v[1] = c[0][1], v[3] = c[0][2], v[7] = c[0][3];
  v[5] = c[1][3], v[4] = c[2][3], v[2] = c[3][2], v[6] = c[4][1];
  bool good = true;
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      good &= c[t][i] == v[fan_vertex(t, i)];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = a + 1; b < 8; ++b)
      good &= v[a] != v[b];
  uint4* q = reinterpret_cast<uint4*>(verts + g * 8);
  q[0] = make_uint4(v[0], v[1], v[2], v[3]);
  q[1] = make_uint4(v[4], v[5], v[6], v[7]);
  ok[g] = good ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------
// Cluster detection from topology and geometry, independent of the order of the cells and of the local vertex order
// inside a cell (DOLFINx reorders both).  A fan is six tets round one shared edge whose other vertices form a closed
// ring of six.  Which edge: every tet names its LONGEST edge (ties: the smaller (vmin, vmax) pair) -- in a cube cut
// into six tets that is the body diagonal, for any box shape.  Tets are then sorted by that key (by the caller); a run
// of exactly six equal keys whose twelve other vertices close into one ring of six distinct vertices is a fan, and its
// eight vertices are written in the local numbering the cluster kernels use (shared edge = local 0 and 7, the ring
// 1-3-2-6-4-5, see fan_vertex).  Orientation does not matter: the kernels take |det| of every tet.
// ---------------------------------------------------------------------------------------------------------
__global__ void tet_long_edge_kernel(const double* __restrict__ x, const int32_t* __restrict__ cells, int64_t n_cells,
                                     int64_t* __restrict__ keys)
{
  const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c >= n_cells)
    return;
  const uint4 w = reinterpret_cast<const uint4*>(cells)[c];
  const int32_t v[4] = {int32_t(w.x), int32_t(w.y), int32_t(w.z), int32_t(w.w)};
  double X[4][3];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k)
      X[i][k] = x[3 * int64_t(v[i]) + k];
  double best = -1.0;
  int64_t key = -1;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = a + 1; b < 4; ++b)
    {
      const double dx = X[a][0] - X[b][0], dy = X[a][1] - X[b][1], dz = X[a][2] - X[b][2];
      // (no fma contraction: the same edge must get the same length in every tet that holds it, whichever of its
      // two vertices comes first -- (a - b)^2 = (b - a)^2 exactly, product by product)
      const double len = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
      const int32_t lo = v[a] < v[b] ? v[a] : v[b], hi = v[a] < v[b] ? v[b] : v[a];
      const int64_t k2 = (int64_t(lo) << 32) | int64_t(uint32_t(hi));
      if (len > best || (len == best && k2 < key))
      {
        best = len;
        key = k2;
      }
    }
  keys[c] = key;
}

// one thread per position of the key-sorted cell list; a run start with exactly six members is examined
__global__ void fan_build_kernel(int64_t n, const int64_t* __restrict__ keys, const int32_t* __restrict__ order,
                                 const int32_t* __restrict__ cells, int32_t* __restrict__ verts, int8_t* __restrict__ ok,
                                 int8_t* __restrict__ cell_in_fan)
{
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= n)
    return;
  ok[p] = 0;
  const int64_t key = keys[p];
  if ((p > 0 && keys[p - 1] == key) || p + 5 >= n || keys[p + 5] != key || (p + 6 < n && keys[p + 6] == key))
    return;
  const int32_t v0 = int32_t(key >> 32), v7 = int32_t(key & 0xffffffff);
  int32_t ea[6], eb[6]; // the two other vertices of every tet: an edge of the ring graph
  for (int t = 0; t < 6; ++t)
  {
    const int32_t* c = cells + int64_t(order[p + t]) * 4;
    int32_t o[2];
    int no = 0, hit = 0;
    for (int i = 0; i < 4; ++i)
    {
      const int32_t u = c[i];
      if (u == v0 || u == v7)
        ++hit;
      else if (no < 2)
        o[no++] = u;
    }
    if (hit != 2 || no != 2 || o[0] == o[1])
      return;
    ea[t] = o[0], eb[t] = o[1];
  }
  // walk the ring: start with tet 0's pair, then always the unused tet that holds the current end
  int32_t ring[6];
  ring[0] = ea[0], ring[1] = eb[0];
  unsigned used = 1u;
  for (int k = 2; k < 7; ++k)
  {
    const int32_t cur = ring[k - 1];
    int found = -1;
    int32_t nxt = -1;
    for (int t = 1; t < 6; ++t)
      if (!((used >> t) & 1) && (ea[t] == cur || eb[t] == cur))
      {
        found = t;
        nxt = ea[t] == cur ? eb[t] : ea[t];
        break;
      }
    if (found < 0)
      return;
    used |= 1u << found;
    if (k < 6)
      ring[k] = nxt;
    else if (nxt != ring[0])
      return; // the sixth tet must close the ring
  }
  for (int a = 0; a < 6; ++a)
  {
    if (ring[a] == v0 || ring[a] == v7)
      return;
    for (int b = a + 1; b < 6; ++b)
      if (ring[a] == ring[b])
        return;
  }
  // local numbering of the cluster kernels: ring 1-3-2-6-4-5 (tets (0,1,3,7) (0,3,2,7) (0,2,6,7) (0,6,4,7) (0,5,7,4) (0,1,7,5))
  int32_t* q = verts + p * 8;
  q[0] = v0, q[7] = v7;
  q[1] = ring[0], q[3] = ring[1], q[2] = ring[2], q[6] = ring[3], q[4] = ring[4], q[5] = ring[5];
  ok[p] = 1;
  for (int t = 0; t < 6; ++t)
    cell_in_fan[order[p + t]] = 1;
}

// set-up: (row block, entity) pairs of the row-block plan, in entity order (the caller sorts them by block):
// entity e belongs to every block that holds one of its nd dofs.  counts / offsets protocol as mpcx_mpc_plan_device.
template <bool FILL>
__global__ void rowblock_pairs_kernel(int64_t n_entities, int estride, const int32_t* __restrict__ entities0,
                                      const int32_t* __restrict__ dofmap0, int nd0, int bs0, int num_blocks,
                                      const int32_t* __restrict__ block_row0, int32_t* __restrict__ counts,
                                      const int64_t* __restrict__ offsets, int32_t* __restrict__ pair_block,
                                      int32_t* __restrict__ pair_ent, int32_t* __restrict__ pair_rows, int rotate)
{
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n_entities)
    return;
  const int64_t cell = entities0 ? entities0[e * estride] : e;
  int32_t seen[32];
  uint32_t rows[32]; // bit i: local dof i has its rows in block seen[k]
  int ns = 0;
  for (int i = 0; i < nd0 && i < 32; ++i)
  {
    const int32_t row = dofmap0[cell * nd0 + i] * bs0;
    int lo = 0, hi = num_blocks; // last block with block_row0[b] <= row
    while (hi - lo > 1)
    {
      const int mid = (lo + hi) >> 1;
      if (block_row0[mid] <= row)
        lo = mid;
      else
        hi = mid;
    }
    // position of local dof i in the order the kernel lists it (lean path: rotated_local, mpcx_kernels.hip)
    const uint32_t bit = 1u << (rotate ? int((i + nd0 - cell % nd0) % nd0) : i);
    bool dup = false;
    for (int k = 0; k < ns; ++k)
      if (seen[k] == lo)
      {
        dup = true;
        rows[k] |= bit;
      }
    if (!dup)
    {
      rows[ns] = bit;
      seen[ns++] = lo;
    }
  }
  if constexpr (!FILL)
    counts[e] = ns;
  else
  {
    const int64_t base = offsets[e];
    for (int k = 0; k < ns; ++k)
    {
      pair_block[base + k] = seen[k];
      pair_ent[base + k] = int32_t(e);
      if (pair_rows)
        pair_rows[base + k] = int32_t(rows[k]);
    }
  }
}

// set-up: one record per (row block, cluster touching it) slot k (ALL: every vertex pair is coupled -- hexahedra)
template <bool ALL>
__global__ void cube_records_kernel(int64_t n_slots, const int32_t* __restrict__ block_ents,
                                    const int32_t* __restrict__ cube_verts, int bs, const int8_t* __restrict__ bc,
                                    const int8_t* __restrict__ is_slave, const mpcx_nnz_t* __restrict__ rowptr,
                                    const int32_t* __restrict__ cols, CubeRec* __restrict__ recs, int32_t* overflow)
{
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n_slots * 8)
    return;
  const int64_t k = t >> 3;
  const int a = int(t & 7);
  const int32_t* v = cube_verts + int64_t(block_ents[k]) * 8;
  const int32_t blk = v[a];
  CubeRec& R = recs[k];
  int32_t m = blk;
  for (int c = 0; c < bs; ++c) // mask of component c in bit 28 + c
  {
    const int64_t d = int64_t(blk) * bs + c;
    if ((bc && bc[d]) || is_slave[d])
      m |= 1 << (MASK_SHIFT + c);
  }
  R.v[a] = m;
  // offsets are counted in column BLOCKS from the start of the block row (every row of a block has the same columns)
  const int64_t lo = rowptr[int64_t(blk) * bs], hi = rowptr[int64_t(blk) * bs + 1];
  for (int b = 0; b < 8; ++b)
  {
    int o = 255;
    if (ALL || fan_coupled(a, b))
    {
      const int64_t pos = find_col(cols, lo, hi, v[b] * bs);
      o = pos < 0 ? 256 : int((pos - lo) / bs);
      if (o > 255)
        atomicOr(overflow, 1);
    }
    R.off[a * 8 + b] = uint8_t(o);
  }
}

// set-up: does slot k fit the narrow format (every coupled offset < 16)?
__global__ void cube_slot_width_kernel(int64_t n_slots, const CubeRec* __restrict__ recs, uint8_t* __restrict__ wide)
{
  const int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= n_slots)
    return;
  const CubeRec& R = recs[k];
  int mx = 0;
  for (int a = 0; a < 8; ++a)
    for (int b = 0; b < 8; ++b)
      if (fan_coupled(a, b))
        mx = max(mx, int(R.off[a * 8 + b]));
  wide[k] = mx > 15 ? 1 : 0;
}
// set-up: narrow record j from wide record src[j]
__global__ void cube_pack_narrow_kernel(int64_t n_out, const int64_t* __restrict__ src, const CubeRec* __restrict__ recs,
                                        CubeRecNarrow* __restrict__ out)
{
  const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= n_out)
    return;
  const CubeRec& R = recs[src[j]];
  CubeRecNarrow N;
  for (int i = 0; i < 8; ++i)
    N.v[i] = R.v[i];
  for (int i = 0; i < 32; ++i)
    N.nib[i] = 0;
  for (int a = 0; a < 8; ++a)
    for (int b = 0; b < 8; ++b)
    {
      const int p = fan_pair_index(a, b);
      if (p >= 0)
        N.nib[p >> 1] |= uint8_t((R.off[a * 8 + b] & 0xf) << (4 * (p & 1)));
    }
  out[j] = N;
}

// ---------------------------------------------------------------------------
// matrix: P1 scalar stiffness, one thread per (row block, cluster) slot
// ---------------------------------------------------------------------------
constexpr int CUBE_MAX_THREADS = 512;

template <bool NARROW>
__global__ void __launch_bounds__(CUBE_MAX_THREADS) matrix_cube_kernel(mpcx_matrix_args_t a)
{
  const int NT = blockDim.x;
  extern __shared__ __align__(16) unsigned char smem[];
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3); // contiguous runs of row blocks per XCD
  if (b >= nb)
    return;
  const int tid = threadIdx.x;
  // (a launch covers the blocks of one record format: cube_block_ids lists them, NULL = all blocks in order)
  const int bb = a.cube_block_ids ? a.cube_block_ids[b] : b;
  const int r0 = a.plan.block_row0[bb], r1 = a.plan.block_row0[bb + 1];
  const int nrow = r1 - r0;
  const int64_t nnz0 = a.rowptr[r0];
  const int nnzb = int(a.rowptr[r1] - nnz0);
  double* s_vals = reinterpret_cast<double*>(smem);
  int32_t* s_rowlo = reinterpret_cast<int32_t*>(s_vals + a.plan.max_nnz);
  const double c0 = a.constants ? a.constants[0] : 1.0;
  constexpr int NW = NARROW ? 4 : 6; // 16-byte words per record
  const uint4* __restrict__ recs = static_cast<const uint4*>(a.cube_recs);
  const int64_t e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  const int32_t* __restrict__ ridx = a.cube_rec_index; // one record per cluster: slot -> record (include/mpcx.h)
  auto load = [&](int64_t t, uint4 (&w)[NW])
  {
    const uint4* p = recs + (ridx ? int64_t(ridx[t]) : t) * NW;
#pragma unroll
    for (int i = 0; i < NW; ++i)
      w[i] = p[i];
  };
  // offset of column v[j] inside row v[i] (static i, j): a byte of the wide record, a nibble of the narrow one
  auto offset_of = [&](const uint4 (&w)[NW], int i, int j) -> int
  {
    uint32_t ow[4 * (NW - 2)];
#pragma unroll
    for (int q = 0; q < NW - 2; ++q)
    {
      ow[4 * q] = w[2 + q].x, ow[4 * q + 1] = w[2 + q].y, ow[4 * q + 2] = w[2 + q].z, ow[4 * q + 3] = w[2 + q].w;
    }
    if constexpr (NARROW)
    {
      const int p = fan_pair_index(i, j);
      return int((ow[p >> 3] >> (4 * (p & 7))) & 0xf);
    }
    else
      return int((ow[(i * 8 + j) >> 2] >> (8 * ((i * 8 + j) & 3))) & 0xff);
  };
  auto gather = [&](const uint4 (&w)[NW], double (&X)[8][3])
  {
    const int32_t v[8] = {int32_t(w[0].x), int32_t(w[0].y), int32_t(w[0].z), int32_t(w[0].w),
                          int32_t(w[1].x), int32_t(w[1].y), int32_t(w[1].z), int32_t(w[1].w)};
#pragma unroll
    for (int i = 0; i < 8; ++i)
    {
      const int64_t n = v[i] & DOF_MASK;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        X[i][k] = a.x[3 * n + k];
    }
  };
  // Software pipeline, two slots deep: while slot t is computed, the 24 coordinates of slot t + NT and the
  // record of slot t + 2 NT are in flight (the kernel runs two waves per SIMD, so the latency of the
  // dependent chain record -> coordinates has to be covered inside the wave; 256 VGPRs are available)
  uint4 cur[NW], nxt[NW];
  double X[8][3], Xn[8][3];
  int64_t t = e0 + tid;
  // the first records and coordinates travel while the block's LDS copy is cleared
  if (t < e1)
  {
    load(t, cur);
    if (t + NT < e1)
      load(t + NT, nxt);
    gather(cur, X);
  }
  for (int i = tid; i < nnzb; i += NT)
    s_vals[i] = 0.0;
  for (int rl = tid; rl < nrow; rl += NT)
    s_rowlo[rl] = int(a.rowptr[r0 + rl] - nnz0);
  __syncthreads();
  for (; t < e1; t += NT)
  {
    const int32_t v[8] = {int32_t(cur[0].x), int32_t(cur[0].y), int32_t(cur[0].z), int32_t(cur[0].w),
                          int32_t(cur[1].x), int32_t(cur[1].y), int32_t(cur[1].z), int32_t(cur[1].w)};
    const bool has_next = t + NT < e1;
    if (has_next)
      gather(nxt, Xn);
    // LDS address of every row of this block that takes contributions (-1: outside the block or masked)
    int base[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
    {
      const int r = v[i] & DOF_MASK;
      const bool mine = r >= r0 && r < r1 && !(v[i] >> MASK_SHIFT);
      base[i] = mine ? s_rowlo[mine ? r - r0 : 0] : -1;
    }
    // The six element tensors are summed per vertex pair (upper triangle, static indices -> registers) walking
    // round the shared edge, and a pair is scattered as soon as its last tet is done, so that only the pairs
    // of the current face stay live (27 accumulators otherwise).
    double A[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = i; j < 8; ++j)
        A[i][j] = 0.0;
#pragma unroll
    for (int step = 0; step < 6; ++step)
    {
      const int tet = fan_order(step);
      double cd[12];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k)
          cd[3 * i + k] = X[fan_vertex(tet, i)][k];
      double G[4][3], det;
      cofactor_gradients<3>(cd, G, det); // det * grad(lambda_i)
      const double s = c0 / (6.0 * fabs(det));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i; j < 4; ++j)
        {
          const int gi = fan_vertex(tet, i), gj = fan_vertex(tet, j);
          const double e = s * (G[i][0] * G[j][0] + G[i][1] * G[j][1] + G[i][2] * G[j][2]);
          A[gi < gj ? gi : gj][gi < gj ? gj : gi] += e;
        }
      // pairs whose last tet this was: rows of this block, unmasked columns
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = i; j < 8; ++j)
        {
          if (fan_last_step(i, j) != step)
            continue;
          const double val = A[i][j];
          if (base[i] >= 0 && !(v[j] >> MASK_SHIFT))
          {
            const int off = offset_of(cur, i, j);
            __hip_atomic_fetch_add(s_vals + base[i] + off, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          if (i != j && base[j] >= 0 && !(v[i] >> MASK_SHIFT))
          {
            const int off = offset_of(cur, j, i);
            __hip_atomic_fetch_add(s_vals + base[j] + off, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
    }
    if (has_next)
    {
#pragma unroll
      for (int i = 0; i < NW; ++i)
        cur[i] = nxt[i];
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k)
          X[i][k] = Xn[i][k];
      if (t + 2 * NT < e1)
        load(t + 2 * NT, nxt);
    }
  }
  __syncthreads();
  MPCX_WRITE_OUT(a, nnz0, nnzb, s_vals, tid, NT);
}

// ---------------------------------------------------------------------------
// vector: P1 source term, one thread per cluster; contributions merged per destination dof in an
// LDS hash table, one device atomic per distinct dof of the workgroup (8 inserts per 6 cells instead
// of 24, and a third of the device atomics: the workgroup's clusters share most of their vertices)
// ---------------------------------------------------------------------------
constexpr int VCUBE_THREADS = 256;
constexpr int VCUBE_LOG2H = 11;
constexpr int VCUBE_H = 1 << VCUBE_LOG2H;
constexpr int VCUBE_PROBES = 64;

template <int FN>
__global__ void __launch_bounds__(VCUBE_THREADS) vector_cube_kernel(mpcx_vector_args_t a)
{
  using Op = ElementOp<3, 1, 1, 1, 1, MPCX_FORM_SOURCE, FN>;
  __shared__ int32_t s_key[VCUBE_H];
  __shared__ double s_val[VCUBE_H];
  for (int i = threadIdx.x; i < VCUBE_H; i += VCUBE_THREADS)
  {
    s_key[i] = -1;
    s_val[i] = 0.0;
  }
  fastmath_init_lds(); // ends in a barrier
  const int64_t c = int64_t(blockIdx.x) * VCUBE_THREADS + threadIdx.x;
  if (c < a.n_cubes)
  {
    int32_t v[8];
    {
      const uint4* p = reinterpret_cast<const uint4*>(a.cube_verts + c * 8);
      const uint4 w0 = p[0], w1 = p[1];
      v[0] = w0.x, v[1] = w0.y, v[2] = w0.z, v[3] = w0.w, v[4] = w1.x, v[5] = w1.y, v[6] = w1.z, v[7] = w1.w;
    }
    double X[8][3];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k)
        X[i][k] = a.x[3 * int64_t(v[i]) + k];
    double be8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      be8[i] = 0.0;
#pragma unroll
    for (int tet = 0; tet < 6; ++tet)
    {
      double cd[12];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k)
          cd[3 * i + k] = X[fan_vertex(tet, i)][k];
      double be[4];
      Op::tabulate(be, nullptr, a.constants, cd, 0, a.kernel);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        be8[fan_vertex(tet, i)] += be[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
    {
      const int32_t d = v[i];
      double val = be8[i];
      if (a.mpc.is_slave[d])
      {
        const int m0 = a.mpc.masters_offsets[d], m1 = a.mpc.masters_offsets[d + 1];
        for (int mi = m0; mi < m1; ++mi)
          __hip_atomic_fetch_add(a.b + MPCX_ROW_POS(a, a.mpc.masters[mi]), a.mpc.coeffs[mi] * val, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
        if (m1 > m0)
          val = 0.0; // be[slave] is cleared once it has been moved (cpp/assemble_vector.h:65)
      }
      if (val != 0.0)
      {
        unsigned h = (unsigned(d) * 2654435761u) >> (32 - VCUBE_LOG2H);
        int probe = 0;
#pragma nounroll
        for (; probe < VCUBE_PROBES; ++probe)
        {
          const int32_t old = atomicCAS(&s_key[h], -1, d);
          if (old == -1 || old == d)
          {
            __hip_atomic_fetch_add(&s_val[h], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
          }
          h = (h + 1) & (VCUBE_H - 1);
        }
        if (probe == VCUBE_PROBES)
          __hip_atomic_fetch_add(a.b + MPCX_ROW_POS(a, d), val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < VCUBE_H; i += VCUBE_THREADS)
  {
    const int32_t d = s_key[i];
    if (d >= 0)
      __hip_atomic_fetch_add(a.b + MPCX_ROW_POS(a, d), s_val[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Owner-computes variant (mpcx_vector_args_t::own_lmap != NULL with MPCX_ALG_CUBE): every cluster is evaluated by the
// row block that holds its local vertex 0; the LDS copy of the block holds its own rows followed by its halo (vertices
// of its clusters that other blocks own), the position of every (cluster, local vertex) comes from own_lmap (slave
// flag in bit 28: skipped here, vector_mpc_kernel moves those rows to their masters).  No hash table, no device
// atomics: b receives every row once from its owner plus the halo sums through vector_spill_reduce_kernel, in a
// fixed order -- the result is bitwise reproducible up to the order of the LDS adds inside a block.
// ---- axis-aligned boxes and the 14-point rule: the right-hand side on a tensor grid -------------------------------------
// The 84 quadrature points of the six tetrahedra of a box lie on 19 x 19 x 19 coordinates (csrc/mpcx_box14.hpp, generated from
// the rule and the fan).  The benchmark's right-hand side (eval_fn case 1, python/benchmarks/bench_periodic.py:85-89) is
//     f = x sin(5 pi y) + g(x - 0.9) g(y - 0.5) g(z - 0.1),   g(t) = exp(-t^2 / 0.02),
// a sum of products of univariate factors: 19 sines and 3 x 19 exponentials per box instead of 84 + 84, and two
// multiplications and one fma per point.  Same points, same weights, same sum as Op::tabulate -- the factors of the Gaussian
// are rounded separately (a few ulp of its value).  Taken when the kernel data hold exactly this rule, no coefficient, and the
// eight vertices of the cluster are the corners of a box (compared exactly: tensor grids give identical coordinates);
// every other cluster takes the per-tetrahedron loop.  MPCX_BOX_GRID=0 switches it off.
__constant__ mpcx_box14::Tables c_box14 = mpcx_box14::TABLES;
__device__ __constant__ int g_box14_enable = 1;

__device__ inline bool box14_rule(const mpcx_kernel_t& k, bool honour_switch = true) // (wave-uniform)
{
  if ((honour_switch && !g_box14_enable) || k.nq != mpcx_box14::NQ || k.coeff_degree != 0 || !k.qpts || !k.qwts)
    return false;
  bool ok = true;
#pragma unroll
  for (int q = 0; q < mpcx_box14::NQ; ++q)
  {
#pragma unroll
    for (int d = 0; d < 3; ++d)
      ok &= k.qpts[3 * q + d] == mpcx_box14::XQ[q][d];
    ok &= k.qwts[q] == mpcx_box14::WQ[q];
  }
  return ok;
}

__device__ inline bool box14_is_box(const double (&X)[8][3])
{
  bool box = true;
#pragma unroll
  for (int v = 1; v < 7; ++v)
#pragma unroll
    for (int d = 0; d < 3; ++d)
      box &= X[v][d] == (((v >> d) & 1) ? X[7][d] : X[0][d]);
  return box;
}

__device__ inline void box14_source_fn1(const double (&X)[8][3], double c0, double (&be8)[8])
{
  using namespace mpcx_box14;
  const FmConsts FK = g_fm_consts; // polynomial coefficients in scalar registers
  const double H[3] = {X[7][0] - X[0][0], X[7][1] - X[0][1], X[7][2] - X[0][2]};
  const double org[3] = {X[0][0] - 0.9, X[0][1] - 0.5, X[0][2] - 0.1}; // relative to the centre of the Gaussian
  double gx[NG], gy[NG], gz[NG], sy[NG];
#pragma unroll
  for (int j = 0; j < NG; ++j)
  {
    const double e = c_box14.grid[j];
    const double x = fma(H[0], e, org[0]), y = fma(H[1], e, org[1]), z = fma(H[2], e, org[2]);
    gx[j] = fast_exp_nonpos_k(-(x * x) * (1.0 / 0.02), FK);
    gy[j] = fast_exp_nonpos_k(-(y * y) * (1.0 / 0.02), FK);
    gz[j] = fast_exp_nonpos_k(-(z * z) * (1.0 / 0.02), FK);
    sy[j] = fast_sinpi_k(fma(5.0, y, 2.5), FK); // 5 y = 5 (y - 0.5) + 2.5
  }
  const double sd = c0 * fabs(H[0] * H[1] * H[2]);
  const double wsd[3] = {c_box14.wu[0] * sd, c_box14.wu[1] * sd, c_box14.wu[2] * sd};
#pragma unroll
  for (int t = 0; t < 6; ++t)
  {
    double S = 0.0, SX[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int ix = IDX[t][q][0], iy = IDX[t][q][1], iz = IDX[t][q][2];
      const double xq = fma(H[0], c_box14.grid[ix], X[0][0]);
      const double f = wsd[WIDX[q]] * fma(xq, sy[iy], gx[ix] * gy[iy] * gz[iz]);
      S += f;
#pragma unroll
      for (int d = 0; d < 3; ++d)
        SX[d] = fma(f, c_box14.lam[LIDX[q][d]], SX[d]);
    }
    double s0 = S;
#pragma unroll
    for (int d = 0; d < 3; ++d)
    {
      s0 -= SX[d];
      be8[fan_vertex(t, d + 1)] += SX[d];
    }
    be8[fan_vertex(t, 0)] += s0;
  }
}

// ---- ... and clusters on a tensor grid of intervals (mpcx_vector_args_t::grid_*): the factors once per interval ----------
// A mesh of boxes whose corners lie on a tensor grid (every box mesh a generator makes) repeats the same 19 coordinates per
// axis in every cluster of a row / column / layer.  The launch first evaluates the univariate factors for every INTERVAL
// (box_grid_tables_kernel: (n_x + n_y + n_z) * 19 points -- 256^3 clusters: 14.6 k points instead of 1.4 G), the cluster
// kernel then reads three rows of the table and spends two multiplications and five fma per quadrature point; it never
// touches the coordinates.  Row of interval r (MPCX_GRID_ROW = 40 doubles): [0, 19) factor A, [20, 39) factor B, [19] = |h|:
//   x: A = g(x - 0.9) |h| c0,  B = x |h| c0;   y: A = g(y - 0.5) |h|,  B = sin(5 pi y) |h|;   z: A = g(z - 0.1) |h|
// so that  f |det| c0 = B_x B_y |h_z| + A_x A_y A_z  at the point (j_x, j_y, j_z).
__global__ void __launch_bounds__(256) box_grid_tables_kernel(int n0, int n1, int n2, const double* __restrict__ iv,
                                                              double* __restrict__ tab, const double* constants,
                                                              mpcx_kernel_t k)
{
  fastmath_init_lds(); // ends in a barrier
  if (!box14_rule(k, false))
    __builtin_trap(); // the caller promised the rule of csrc/mpcx_box14.hpp (include/mpcx.h): fail loudly
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = i / 20, j = i - r * 20;
  if (r >= n0 + n1 + n2)
    return;
  const int d = r < n0 ? 0 : (r < n0 + n1 ? 1 : 2);
  const double lo = iv[2 * r], hi = iv[2 * r + 1];
  const double h = hi - lo, ah = fabs(h);
  double* row = tab + int64_t(r) * MPCX_GRID_ROW;
  if (j == 19)
  {
    row[19] = ah;
    row[39] = 0.0;
    return;
  }
  const FmConsts FK = g_fm_consts;
  const double e = c_box14.grid[j];
  const double centre = d == 0 ? 0.9 : (d == 1 ? 0.5 : 0.1);
  const double t = fma(h, e, lo - centre); // relative to the centre of the Gaussian, as the per-point evaluation does
  const double g = fast_exp_nonpos_k(-(t * t) * (1.0 / 0.02), FK);
  const double sc = d == 0 ? ah * (constants ? constants[0] : 1.0) : ah;
  row[j] = g * sc;
  row[20 + j] = d == 0 ? (t + 0.9) * sc : (d == 1 ? fast_sinpi_k(fma(5.0, t, 2.5), FK) * sc : 0.0);
}

constexpr int VCUBE_OWN_THREADS = 256;
constexpr int VCUBE_OWN_MAX_THREADS = 256; // launch bound of vector_cube_own_kernel (MPCX_VCUBE_THREADS, default 256)
// MPCX_BOX_GRID=0: point by point on every cluster / cell (read per launch: the parity tests run both in one process)
inline int sync_box_switch()
{
  static int grid_on = 1;
  const char* e = std::getenv("MPCX_BOX_GRID");
  const int want = (e && e[0] == '0') ? 0 : 1;
  if (want != grid_on)
  {
    if (int rc = check(hipMemcpyToSymbol(HIP_SYMBOL(g_box14_enable), &want, sizeof(int)), "hipMemcpyToSymbol"))
      return rc;
    grid_on = want;
  }
  return 0;
}
inline int vcube_own_threads()
{
  static const int n = []
  {
    const char* e = std::getenv("MPCX_VCUBE_THREADS");
    const int v = e ? std::atoi(e) : VCUBE_OWN_THREADS;
    return (v >= 64 && v <= VCUBE_OWN_MAX_THREADS && v % 64 == 0) ? v : VCUBE_OWN_THREADS;
  }();
  return n;
}

template <int FN, bool BOXES = false> // BOXES (FN = 1): the instance that looks for box clusters (247 instead of 151 registers)
__global__ void __launch_bounds__(VCUBE_OWN_MAX_THREADS) vector_cube_own_kernel(mpcx_vector_args_t a)
{
  using Op = ElementOp<3, 1, 1, 1, 1, MPCX_FORM_SOURCE, FN>;
  extern __shared__ __align__(16) unsigned char smem[];
  double* s_b = reinterpret_cast<double*>(smem);
  const int NT = blockDim.x;
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const int tid = threadIdx.x;
  const int r0 = b < nb ? a.plan.block_row0[b] : 0, r1 = b < nb ? a.plan.block_row0[b + 1] : 0;
  const int64_t h0 = b < nb ? a.own_hoff[b] : 0, h1 = b < nb ? a.own_hoff[b + 1] : 0;
  const int nown = r1 - r0, nhalo = int(h1 - h0);
  for (int i = tid; i < nown + nhalo; i += NT)
    s_b[i] = 0.0;
  fastmath_init_lds(); // ends in a barrier
  if (b >= nb)
    return;
  const int64_t e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  const int32_t* __restrict__ ents = a.plan.block_ents;
  [[maybe_unused]] bool grid_rule = false;
  if constexpr (FN == 1 && BOXES)
    grid_rule = box14_rule(a.kernel);
  for (int64_t t = e0 + tid; t < e1; t += NT)
  {
    const int64_t c = ents[t];
    int32_t v[8];
    {
      const uint4* p = reinterpret_cast<const uint4*>(a.cube_verts + c * 8);
      const uint4 w0 = p[0], w1 = p[1];
      v[0] = w0.x, v[1] = w0.y, v[2] = w0.z, v[3] = w0.w, v[4] = w1.x, v[5] = w1.y, v[6] = w1.z, v[7] = w1.w;
    }
    double X[8][3];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k)
        X[i][k] = a.x[3 * int64_t(v[i]) + k];
    double be8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      be8[i] = 0.0;
    bool on_grid = false;
    if constexpr (FN == 1 && BOXES)
      on_grid = grid_rule && box14_is_box(X);
    if (on_grid)
    {
      if constexpr (FN == 1 && BOXES)
        box14_source_fn1(X, a.constants ? a.constants[0] : 1.0, be8);
    }
    else
    {
#pragma unroll
      for (int tet = 0; tet < 6; ++tet)
      {
        double cd[12];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int k = 0; k < 3; ++k)
            cd[3 * i + k] = X[fan_vertex(tet, i)][k];
        double be[4];
        Op::tabulate(be, nullptr, a.constants, cd, 0, a.kernel);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          be8[fan_vertex(tet, i)] += be[i];
      }
    }
    // LDS positions of the eight vertices (read after the quadrature: eight registers less across it)
    int32_t w[8];
    {
      const uint4* p = reinterpret_cast<const uint4*>(a.own_lmap + c * 8);
      const uint4 w0 = p[0], w1 = p[1];
      w[0] = w0.x, w[1] = w0.y, w[2] = w0.z, w[3] = w0.w, w[4] = w1.x, w[5] = w1.y, w[6] = w1.z, w[7] = w1.w;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (!(w[i] >> MASK_SHIFT))
        __hip_atomic_fetch_add(s_b + (w[i] & DOF_MASK), be8[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  for (int i = tid; i < nown; i += NT)
    a.b[MPCX_ROW_POS(a, r0 + i)] += s_b[i];
  for (int i = tid; i < nhalo; i += NT)
    a.own_spill[h0 + i] = s_b[nown + i];
}

// The cluster kernel of the tensor-grid plan: owner-computes row blocks as vector_cube_own_kernel, one thread per cluster,
// no coordinates, no transcendental function (box_grid_tables_kernel ran before it on the same stream).
// Rows of the table a block keeps in LDS (mpcx_vector_args_t::grid_block_rows): the intervals its clusters sit on, per axis
// -- a tile of the numbering spans a few of them.  The plan lists them per block and numbers the clusters' intervals by
// their LDS row, so a block whose clusters sit in two places (the end of one row of tiles and the start of the next) stages
// its two groups of rows like any other.  Rows of 42 doubles in LDS: with the table's 40 (80 dwords, 16 mod 32) the rows of
// a wave's clusters fall on two groups of banks; 84 dwords = 20 mod 32 puts eight rows on eight groups of four banks.
constexpr int GRID_LROW = 42;

template <bool STAGED>
__device__ __forceinline__ void vector_cube_grid_body(const mpcx_vector_args_t& a)
{
  using namespace mpcx_box14;
  extern __shared__ __align__(16) unsigned char smem[];
  double* s_b = reinterpret_cast<double*>(smem);
  const int NT = blockDim.x;
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (b >= nb)
    return;
  const int tid = threadIdx.x;
  const int r0 = a.plan.block_row0[b], r1 = a.plan.block_row0[b + 1];
  const int64_t h0 = a.own_hoff[b], h1 = a.own_hoff[b + 1];
  const int nown = r1 - r0, nhalo = int(h1 - h0);
  for (int i = tid; i < nown + nhalo; i += NT)
    s_b[i] = 0.0;
  const int64_t e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  const int32_t* __restrict__ ents = a.plan.block_ents;
  const double* __restrict__ tabx = a.grid_tab;
  const double* __restrict__ taby = tabx + int64_t(a.grid_n[0]) * MPCX_GRID_ROW;
  const double* __restrict__ tabz = taby + int64_t(a.grid_n[1]) * MPCX_GRID_ROW;
  // LDS behind the block's rows of b: the rows of the table the plan lists for the block
  double* s_rows = s_b + ((a.plan.max_rows + 1) & ~1);
  if constexpr (STAGED)
  {
    const int32_t* __restrict__ br = a.grid_block_rows + int64_t(b) * MPCX_GRID_BLOCK_ROWS;
    const int nslots = (a.grid_block_rows_max > 0 && a.grid_block_rows_max <= MPCX_GRID_BLOCK_ROWS) ? a.grid_block_rows_max
                                                                                                      : MPCX_GRID_BLOCK_ROWS;
    for (int i = tid; i < nslots * MPCX_GRID_ROW; i += NT)
    {
      const int slot = i / MPCX_GRID_ROW, j = i - slot * MPCX_GRID_ROW;
      const int r = br[slot];
      if (r >= 0)
        s_rows[slot * GRID_LROW + j] = tabx[int64_t(r) * MPCX_GRID_ROW + j];
    }
  }
  __syncthreads();
  // the cluster id and the intervals of a thread's NEXT cluster are loaded while it works on the current one, the LDS
  // positions of the current one before its arithmetic: one exposed round trip per cluster instead of three
  int64_t cn = 0;
  int4 idxn = {0, 0, 0, 0};
  if (e0 + tid < e1)
  {
    cn = ents[e0 + tid];
    idxn = reinterpret_cast<const int4*>(a.grid_idx)[cn];
  }
  for (int64_t t = e0 + tid; t < e1; t += NT)
  {
    const int64_t c = cn;
    const int4 idx = idxn;
    int32_t w[8];
    {
      const uint4* p = reinterpret_cast<const uint4*>(a.own_lmap + c * 8);
      const uint4 w0 = p[0], w1 = p[1];
      w[0] = w0.x, w[1] = w0.y, w[2] = w0.z, w[3] = w0.w, w[4] = w1.x, w[5] = w1.y, w[6] = w1.z, w[7] = w1.w;
    }
    if (t + NT < e1)
    {
      cn = ents[t + NT];
      idxn = reinterpret_cast<const int4*>(a.grid_idx)[cn];
    }
    const double* __restrict__ rx = STAGED ? s_rows + idx.x * GRID_LROW : tabx + int64_t(idx.x) * MPCX_GRID_ROW;
    const double* __restrict__ ry = STAGED ? s_rows + idx.y * GRID_LROW : taby + int64_t(idx.y) * MPCX_GRID_ROW;
    const double* __restrict__ rz = STAGED ? s_rows + idx.z * GRID_LROW : tabz + int64_t(idx.z) * MPCX_GRID_ROW;
    double gy[NG + 1], sy[NG + 1], gz[NG + 1]; // (rows are 16-byte aligned: pairs; gz[NG] = |h_z|)
    if constexpr (STAGED)
    {
      typedef double __attribute__((ext_vector_type(2))) pair_t;
      typedef const __attribute__((address_space(3))) pair_t lds_pair_t;
      const unsigned ly = unsigned(reinterpret_cast<uintptr_t>(ry)), lz = unsigned(reinterpret_cast<uintptr_t>(rz));
#pragma unroll
      for (int j = 0; j < NG + 1; j += 2)
      {
        const pair_t u = *reinterpret_cast<lds_pair_t*>(ly + 8 * j);
        const pair_t v = *reinterpret_cast<lds_pair_t*>(ly + 8 * (20 + j));
        const pair_t w = *reinterpret_cast<lds_pair_t*>(lz + 8 * j);
        gy[j] = u.x, gy[j + 1] = u.y, sy[j] = v.x, sy[j + 1] = v.y, gz[j] = w.x, gz[j + 1] = w.y;
      }
    }
    else
    {
#pragma unroll
      for (int j = 0; j < NG + 1; j += 2)
      {
        const double2 u = *reinterpret_cast<const double2*>(ry + j);
        const double2 v = *reinterpret_cast<const double2*>(ry + 20 + j);
        const double2 w = *reinterpret_cast<const double2*>(rz + j);
        gy[j] = u.x, gy[j + 1] = u.y, sy[j] = v.x, sy[j + 1] = v.y, gz[j] = w.x, gz[j + 1] = w.y;
      }
    }
    const double hz = gz[NG];
    double be8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      be8[i] = 0.0;
    // plane by plane in x: the two factors in x of a plane are loaded one plane ahead (the opaque address keeps the compiler
    // from loading all 38 at once, and the product order from forming the 84 products g_y g_z before the first plane:
    // either put the kernel above 256 registers)
    [[maybe_unused]] const double* rxp = rx;
    [[maybe_unused]] unsigned rxl = 0; // (LDS byte address of the row)
    if constexpr (STAGED)
      rxl = unsigned(reinterpret_cast<uintptr_t>(rx));
    auto plane = [&](int j, double& g, double& x)
    {
      if constexpr (STAGED)
      {
        asm volatile("" : "+v"(rxl));
        g = *reinterpret_cast<const __attribute__((address_space(3))) double*>(rxl + 8 * j);
        x = *reinterpret_cast<const __attribute__((address_space(3))) double*>(rxl + 8 * (20 + j));
      }
      else
      {
        asm volatile("" : "+v"(rxp));
        g = rxp[j], x = rxp[20 + j];
      }
    };
    double gxn, xqn;
    plane(0, gxn, xqn);
#pragma unroll
    for (int j = 0; j < NG; ++j)
    {
      const double gxj = gxn;
      const double xq = xqn * hz;
      if (j + 1 < NG)
        plane(j + 1, gxn, xqn);
#pragma unroll
      for (int tet = 0; tet < 6; ++tet)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          if (IDX[tet][q][0] != j) // (compile-time: the loops are unrolled)
            continue;
          const int iy = IDX[tet][q][1], iz = IDX[tet][q][2];
          const double f = fma(xq, sy[iy], (gxj * gy[iy]) * gz[iz]);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            be8[fan_vertex(tet, i)] = fma(f, c_box14.wl[WLIDX[q][i]], be8[fan_vertex(tet, i)]);
        }
    }
    // Neighbouring clusters share vertices, and neighbouring lanes usually hold neighbouring clusters (a tile lists its
    // clusters row by row): a lane hands the sums of the vertices it shares with the lane one / eight to its right (the
    // next cluster in x / in y of a tile eight clusters wide) over to that lane instead of adding them to LDS itself --
    // whenever the LDS positions agree, whatever the order of the list.  Interior clusters are left with two LDS adds out
    // of eight: the fp64 LDS adds (about 90 cycles each with their bank conflicts) kept the LDS pipe of a CU busy half of
    // the kernel's time and the table reads of the other waves behind them.
    if (__ballot(1) == ~0ull) // (full waves only: lane shifts)
    {
      auto fold = [&](auto shr_c, auto shl_c, int dst, int src)
      {
        constexpr int SHR = decltype(shr_c)::value, SHL = decltype(shl_c)::value;
        const int wl_ = __builtin_amdgcn_update_dpp(-1, w[src], SHR, 0xf, 0xf, false); // (no lane to the left in the row: -1)
        const unsigned long long v = __double_as_longlong(be8[src]);
        const int lo = __builtin_amdgcn_update_dpp(0, int(unsigned(v)), SHR, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, int(unsigned(v >> 32)), SHR, 0xf, 0xf, false);
        const bool take = wl_ == w[dst] && !(w[dst] >> MASK_SHIFT);
        if (take)
          be8[dst] += __longlong_as_double((long long)((unsigned long long)unsigned(hi) << 32 | unsigned(lo)));
        const int given = __builtin_amdgcn_update_dpp(0, int(take), SHL, 0xf, 0xf, false);
        if (given)
          w[src] = -1; // (flag bits set: skipped below)
      };
      constexpr std::integral_constant<int, 0x111> X_SHR{}; // row_shr:1 / row_shl:1 / row_shr:8 / row_shl:8
      constexpr std::integral_constant<int, 0x101> X_SHL{};
      constexpr std::integral_constant<int, 0x118> Y_SHR{};
      constexpr std::integral_constant<int, 0x108> Y_SHL{};
      fold(X_SHR, X_SHL, 0, 1), fold(X_SHR, X_SHL, 2, 3), fold(X_SHR, X_SHL, 4, 5), fold(X_SHR, X_SHL, 6, 7);
      fold(Y_SHR, Y_SHL, 0, 2), fold(Y_SHR, Y_SHL, 1, 3), fold(Y_SHR, Y_SHL, 4, 6), fold(Y_SHR, Y_SHL, 5, 7);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (!(w[i] >> MASK_SHIFT))
        __hip_atomic_fetch_add(s_b + (w[i] & DOF_MASK), be8[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  for (int i = tid; i < nown; i += NT)
    a.b[MPCX_ROW_POS(a, r0 + i)] += s_b[i];
  for (int i = tid; i < nhalo; i += NT)
    a.own_spill[h0 + i] = s_b[nown + i];
}
// (a budget of 168 registers -- three waves per SIMD -- spills the tail of the tables: 3.9 against 1.2 ms)
template <bool STAGED>
__global__ void __launch_bounds__(VCUBE_OWN_MAX_THREADS) __attribute__((amdgpu_waves_per_eu(2)))
vector_cube_grid_kernel(mpcx_vector_args_t a)
{
  vector_cube_grid_body<STAGED>(a);
}

// ---------------------------------------------------------------------------------------------------------
// Hexahedra (Q1, trilinear geometry) -- the default cell of python/benchmarks/bench_periodic.py:38,199-200.
// A hexahedron is the cluster: eight vertices, all 64 vertex pairs coupled, so the row-block machinery of the
// six-tet fans carries over (96-byte records = 8 vertex ids + 64 scatter offsets, owner-computes vector plan over
// the cell dofmap).  Local vertex v sits at (v & 1, v >> 1 & 1, v >> 2 & 1) of the reference cube.  The map is
//     x(X, Y, Z) = sum_m c[m] X^(m & 1) Y^(m >> 1 & 1) Z^(m >> 2 & 1),
// its Jacobian columns j0, j1, j2 are bilinear in the other two variables, the cofactor columns are cross products:
//     grad(phi_i) = (j1 x j2, j2 x j0, j0 x j1) dphi_i / det,   det = j0 . (j1 x j2).
// ---------------------------------------------------------------------------------------------------------
template <int N>
struct Gauss01; // Gauss-Legendre rule on [0, 1]
template <>
struct Gauss01<1>
{
  static constexpr double P[1] = {0.5};
  static constexpr double W[1] = {1.0};
};
template <>
struct Gauss01<2>
{
  static constexpr double P[2] = {0.21132486540518711775, 0.78867513459481288225};
  static constexpr double W[2] = {0.5, 0.5};
};
template <>
struct Gauss01<3>
{
  static constexpr double P[3] = {0.11270166537925831148, 0.5, 0.88729833462074168852};
  static constexpr double W[3] = {5.0 / 18.0, 8.0 / 18.0, 5.0 / 18.0};
};

// trilinear coefficients from the eight vertices; differences are paired so that a parallelepiped whose opposite
// edges are the same floating-point differences (any axis-aligned box grid) gets exact zeros in c[3], c[5], c[6], c[7]
__device__ inline void hex_coefficients(const double (&X)[8][3], double (&c)[8][3])
{
#pragma unroll
  for (int r = 0; r < 3; ++r)
  {
    const double e10 = X[1][r] - X[0][r], e32 = X[3][r] - X[2][r], e54 = X[5][r] - X[4][r], e76 = X[7][r] - X[6][r];
    const double e20 = X[2][r] - X[0][r], e64 = X[6][r] - X[4][r];
    c[0][r] = X[0][r];
    c[1][r] = e10;
    c[2][r] = e20;
    c[3][r] = e32 - e10;
    c[4][r] = X[4][r] - X[0][r];
    c[5][r] = e54 - e10;
    c[6][r] = e64 - e20;
    c[7][r] = (e76 - e54) - (e32 - e10);
  }
}
// is the cell a parallelepiped (to rounding)?  The bilinear / trilinear coefficients against the edge vectors.
__device__ inline bool hex_is_parallelepiped(const double (&c)[8][3])
{
  double dev = 0.0, len = 0.0;
#pragma unroll
  for (int r = 0; r < 3; ++r)
  {
    dev = fmax(dev, fmax(fmax(fabs(c[3][r]), fabs(c[5][r])), fmax(fabs(c[6][r]), fabs(c[7][r]))));
    len = fmax(len, fmax(fabs(c[1][r]), fmax(fabs(c[2][r]), fabs(c[4][r]))));
  }
  return dev <= 0x1p-46 * len;
}
__global__ void hex_slot_shapes_kernel(int64_t n_slots, const int32_t* __restrict__ verts, int stride,
                                       const double* __restrict__ x, uint8_t* __restrict__ general)
{
  const int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= n_slots)
    return;
  double X[8][3], c[8][3];
#pragma unroll
  for (int i = 0; i < 8; ++i)
  {
    const int64_t n = verts[k * stride + i] & DOF_MASK;
#pragma unroll
    for (int r = 0; r < 3; ++r)
      X[i][r] = x[3 * n + r];
  }
  hex_coefficients(X, c);
  general[k] = hex_is_parallelepiped(c) ? 0 : 1;
}
// The ring of a fan found from topology starts at an arbitrary ring vertex; topologically all six are alike (each shares
// tet edges with both ends of the shared edge), geometrically they alternate between cube-edge neighbours of vertex 0
// (corners 1, 2, 4) and of vertex 7 (corners 3, 6, 5).  The closed-form kernels need the corner numbering: if the
// fan is a parallelepiped only after the ring is turned by one position, it is renumbered here.
__global__ void fan_canonical_kernel(int64_t n, int32_t* __restrict__ verts, const int8_t* __restrict__ ok,
                                     const double* __restrict__ x)
{
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= n || (ok && !ok[p]))
    return;
  int32_t v[8];
  for (int i = 0; i < 8; ++i)
    v[i] = verts[p * 8 + i];
  auto is_par = [&](const int32_t (&w)[8]) -> bool
  {
    double X[8][3], c[8][3];
    for (int i = 0; i < 8; ++i)
      for (int r = 0; r < 3; ++r)
        X[i][r] = x[3 * int64_t(w[i]) + r];
    hex_coefficients(X, c);
    return hex_is_parallelepiped(c);
  };
  // Round 6: an axis-aligned box gets its corners in the order of the coordinate axes as well (corner 1 = the x neighbour of
  // corner 0, 2 = y, 4 = z): the six tetrahedra of the fan are the six orders of adding the three edge vectors, so any
  // permutation of the axes maps the fan onto itself -- and the box paths of the vector kernels (box14_source_fn1, the
  // tensor-grid plan) tell a box by exactly this numbering.  A mesh with shuffled cells leaves the ring walk at any of them.
  auto box_axes = [&](int32_t (&u)[8])
  {
    double X0[3], E[3][3];
    for (int r = 0; r < 3; ++r)
      X0[r] = x[3 * int64_t(u[0]) + r];
    const int corner[3] = {1, 2, 4};
    int axis[3];
    for (int k = 0; k < 3; ++k)
    {
      int nz = 0;
      axis[k] = 0;
      for (int r = 0; r < 3; ++r)
      {
        E[k][r] = x[3 * int64_t(u[corner[k]]) + r] - X0[r];
        if (E[k][r] != 0.0)
          ++nz, axis[k] = r;
      }
      if (nz != 1)
        return;
    }
    if (axis[0] == axis[1] || axis[0] == axis[2] || axis[1] == axis[2] || (axis[0] == 0 && axis[1] == 1))
      return; // not a box, or already in axis order
    int32_t t[8];
    for (int i = 0; i < 8; ++i)
      t[((i & 1) << axis[0]) | (((i >> 1) & 1) << axis[1]) | (((i >> 2) & 1) << axis[2])] = u[i];
    for (int i = 0; i < 8; ++i)
      u[i] = t[i];
  };
  if (is_par(v))
  {
    box_axes(v);
    for (int i = 0; i < 8; ++i)
      verts[p * 8 + i] = v[i];
    return;
  }
  // ring 1-3-2-6-4-5 turned by one: new 1 = old 3, new 3 = old 2, new 2 = old 6, new 6 = old 4, new 4 = old 5, new 5 = old 1
  int32_t w[8] = {v[0], v[3], v[6], v[2], v[5], v[1], v[4], v[7]};
  if (is_par(w))
  {
    box_axes(w);
    for (int i = 0; i < 8; ++i)
      verts[p * 8 + i] = w[i];
  }
}
// Imported (UFCx) element kernels are called with a cell's vertices in the order the mesh lists them: a quadrature rule
// need not be symmetric, so a permuted call is another approximation of the integral.  The cluster kernels built round an
// imported function hand it the coordinates of tet t as local vertices fan_vertex(t, 0..3) of the cluster; that is the
// mesh's own order exactly when the six cells read, in the cluster's numbering, as the rows of the table
//     (0,1,3,7) (0,1,7,5) (0,5,7,4) (0,3,2,7) (0,6,4,7) (0,2,6,7)
// (what a Kuhn box generator emits).  Given a fan found from topology (any numbering of its ring) and its six cells, this
// kernel finds the one numbering under which they do -- vertex 0 is every cell's first vertex, the two cells with the other
// shared vertex in third place are (0,1,7,5) and (0,5,7,4), the rest follows along the ring -- renumbers verts[p], orders
// cells[p] by table row and sets ok[p]; fans whose cells are listed otherwise get ok[p] = 0 (their cells then go through the
// per-cell kernels).
__global__ void fan_ordered_kernel(int64_t n, int32_t* __restrict__ verts, int32_t* __restrict__ fan_cells,
                                   const int32_t* __restrict__ x_dofmap, int8_t* __restrict__ ok)
{
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= n)
    return;
  ok[p] = 0;
  int32_t cell[6], c[6][4];
  for (int t = 0; t < 6; ++t)
  {
    cell[t] = fan_cells[p * 6 + t];
    for (int i = 0; i < 4; ++i)
      c[t][i] = x_dofmap[int64_t(cell[t]) * 4 + i];
  }
  const int32_t v0 = c[0][0];
  const int32_t a0 = verts[p * 8], a7 = verts[p * 8 + 7];
  if (v0 != a0 && v0 != a7)
    return;
  const int32_t v7 = v0 == a0 ? a7 : a0;
  // the two cells with v7 in third place
  int third[2], n3 = 0;
  for (int t = 0; t < 6; ++t)
  {
    if (c[t][0] != v0)
      return;
    if (c[t][2] == v7)
    {
      if (n3 < 2)
        third[n3] = t;
      ++n3;
    }
    else if (c[t][3] != v7)
      return;
  }
  if (n3 != 2)
    return;
  int t1 = third[0], t2 = third[1]; // (0,1,7,5), (0,5,7,4): the first one's last vertex is the second one's second
  if (c[t1][3] != c[t2][1])
  {
    const int s = t1;
    t1 = t2, t2 = s;
  }
  if (c[t1][3] != c[t2][1])
    return;
  int32_t v[8];
  v[0] = v0, v[7] = v7, v[1] = c[t1][1], v[5] = c[t1][3], v[4] = c[t2][3];
  int row[6] = {-1, t1, t2, -1, -1, -1}; // cell of every table row
  unsigned used = (1u << t1) | (1u << t2);
  // (0,1,3,7) -> (0,3,2,7) -> (0,2,6,7) -> (0,6,4,7): each starts where the one before ended
  const int chain_row[4] = {0, 3, 5, 4};
  const int chain_new[4] = {3, 2, 6, 4};
  int32_t cur = v[1];
  for (int k = 0; k < 4; ++k)
  {
    int found = -1;
    for (int t = 0; t < 6; ++t)
      if (!((used >> t) & 1) && c[t][1] == cur && c[t][3] == v7)
        found = t;
    if (found < 0)
      return;
    used |= 1u << found;
    row[chain_row[k]] = found;
    if (k < 3)
      v[chain_new[k]] = c[found][2];
    else if (c[found][2] != v[4])
      return;
    cur = c[found][2];
  }
  for (int a = 0; a < 8; ++a)
    for (int b = a + 1; b < 8; ++b)
      if (v[a] == v[b])
        return;
  for (int t = 0; t < 6; ++t)
    for (int i = 0; i < 4; ++i)
      if (c[row[t]][i] != v[fan_vertex(t, i)])
        return;
  for (int i = 0; i < 8; ++i)
    verts[p * 8 + i] = v[i];
  for (int t = 0; t < 6; ++t)
    fan_cells[p * 6 + t] = cell[row[t]];
  ok[p] = 1;
}
__device__ inline void cross3(const double (&u)[3], const double (&v)[3], double (&w)[3])
{
  w[0] = u[1] * v[2] - u[2] * v[1];
  w[1] = u[2] * v[0] - u[0] * v[2];
  w[2] = u[0] * v[1] - u[1] * v[0];
}
// index of (i, j), i <= j, in the packed upper triangle of an 8 x 8 matrix
__host__ __device__ constexpr int tri8(int i, int j) { return i * 8 - i * (i - 1) / 2 + (j - i); }

// integrals over [0, 1] of products of the 1D hat functions l_0 = 1 - t, l_1 = t
__host__ __device__ constexpr double hat_mass(int a, int b) { return a == b ? 1.0 / 3.0 : 1.0 / 6.0; }
__host__ __device__ constexpr double hat_sign(int a) { return a ? 1.0 : -1.0; }
// coefficient of M_de in the exact Q1 stiffness entry (i, j) of a parallelepiped:
//   A_ij = sum_{d <= e} M_de K_de(i, j),  M = c (C^T C) / |det|
__host__ __device__ constexpr double hex_affine_coef(int d, int e, int i, int j)
{
  const int bi[3] = {i & 1, (i >> 1) & 1, (i >> 2) & 1}, bj[3] = {j & 1, (j >> 1) & 1, (j >> 2) & 1};
  if (d == e)
  {
    double v = hat_sign(bi[d]) * hat_sign(bj[d]);
    for (int k = 0; k < 3; ++k)
      if (k != d)
        v *= hat_mass(bi[k], bj[k]);
    return v;
  }
  // int d_d phi_i d_e phi_j + int d_e phi_i d_d phi_j: (s/2)(s/2) in the two differentiated directions
  const int k = 3 - d - e;
  return (hat_sign(bi[d]) * hat_sign(bj[e]) + hat_sign(bi[e]) * hat_sign(bj[d])) * 0.25 * hat_mass(bi[k], bj[k]);
}

constexpr int HEX_MAX_THREADS = 512;

// matrix: scalar Q1 stiffness c * inner(grad u, grad v) dx, 2 x 2 x 2 Gauss points, one thread per (row block, cell)
// slot.  The 36 entries of the upper triangle are accumulated over the points in registers and scattered once.
// A wave all of whose cells are parallelepipeds (c[3] = c[5] = c[6] = c[7] = 0 up to 2^-46 of the longest edge
// vector: every cell of a box mesh, sheared or not) takes the closed form of the same integral -- the rule is exact
// there -- from the six entries of M = c C^T C / |det|: ~200 fp64 instructions instead of ~2 100.
// AFFINE_PATH: 1 = both paths, chosen per wave; 2 = closed form only (row blocks all of whose cells the set-up found to be
// parallelepipeds, mpcx_matrix_args_t::cube_flags bit 0: 104 registers instead of 252, four waves per SIMD cover the
// record -> coordinates round trip: 1.38 ms instead of 1.75 at 256^3 cells); 0 = quadrature only (MPCX_HEX_NO_AFFINE, tests).
template <int AFFINE_PATH>
__global__ void __launch_bounds__(HEX_MAX_THREADS) matrix_hex_kernel(mpcx_matrix_args_t a)
{
  const int NT = blockDim.x;
  extern __shared__ __align__(16) unsigned char smem[];
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3); // contiguous runs of row blocks per XCD
  if (b >= nb)
    return;
  const int tid = threadIdx.x;
  const int bb = a.cube_block_ids ? a.cube_block_ids[b] : b;
  const int r0 = a.plan.block_row0[bb], r1 = a.plan.block_row0[bb + 1];
  const int nrow = r1 - r0;
  const int64_t nnz0 = a.rowptr[r0];
  const int nnzb = int(a.rowptr[r1] - nnz0);
  double* s_vals = reinterpret_cast<double*>(smem);
  int32_t* s_rowlo = reinterpret_cast<int32_t*>(s_vals + a.plan.max_nnz);
  const double c0 = a.constants ? a.constants[0] : 1.0;
  const uint4* __restrict__ recs = static_cast<const uint4*>(a.cube_recs);
  const int64_t e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  const int32_t* __restrict__ ridx = a.cube_rec_index; // one record per cell: slot -> record (include/mpcx.h)
  auto load = [&](int64_t t, uint4 (&w)[6])
  {
    const uint4* p = recs + (ridx ? int64_t(ridx[t]) : t) * 6;
#pragma unroll
    for (int i = 0; i < 6; ++i)
      w[i] = p[i];
  };
  uint4 cur[6], nxt[6];
  int64_t t = e0 + tid;
  if (t < e1)
    load(t, cur); // the first record travels while the block's LDS copy is cleared
  for (int i = tid; i < nnzb; i += NT)
    s_vals[i] = 0.0;
  for (int rl = tid; rl < nrow; rl += NT)
    s_rowlo[rl] = int(a.rowptr[r0 + rl] - nnz0);
  __syncthreads();
  for (; t < e1; t += NT)
  {
    const int32_t v[8] = {int32_t(cur[0].x), int32_t(cur[0].y), int32_t(cur[0].z), int32_t(cur[0].w),
                          int32_t(cur[1].x), int32_t(cur[1].y), int32_t(cur[1].z), int32_t(cur[1].w)};
    double c[8][3];
    {
      double X[8][3];
#pragma unroll
      for (int i = 0; i < 8; ++i)
      {
        const int64_t n = v[i] & DOF_MASK;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          X[i][k] = a.x[3 * n + k];
      }
      hex_coefficients(X, c);
    }
    const bool has_next = t + NT < e1;
    if (has_next)
      load(t + NT, nxt); // in flight across the arithmetic below
    double A[36];
    bool affine = false;
    if constexpr (AFFINE_PATH == 2)
      affine = true;
    else if constexpr (AFFINE_PATH == 1)
      affine = __all(hex_is_parallelepiped(c));
    if (affine)
    {
      double C0[3], C1[3], C2[3];
      cross3(c[2], c[4], C0);
      cross3(c[4], c[1], C1);
      cross3(c[1], c[2], C2);
      const double det = c[1][0] * C0[0] + c[1][1] * C0[1] + c[1][2] * C0[2];
      const double s = c0 / fabs(det);
      const double M[3][3] = {{s * (C0[0] * C0[0] + C0[1] * C0[1] + C0[2] * C0[2]), s * (C0[0] * C1[0] + C0[1] * C1[1] + C0[2] * C1[2]),
                               s * (C0[0] * C2[0] + C0[1] * C2[1] + C0[2] * C2[2])},
                              {0.0, s * (C1[0] * C1[0] + C1[1] * C1[1] + C1[2] * C1[2]), s * (C1[0] * C2[0] + C1[1] * C2[1] + C1[2] * C2[2])},
                              {0.0, 0.0, s * (C2[0] * C2[0] + C2[1] * C2[1] + C2[2] * C2[2])}};
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = i; j < 8; ++j)
        {
          double acc = 0.0;
#pragma unroll
          for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int e = d; e < 3; ++e)
            {
              const double k = hex_affine_coef(d, e, i, j);
              if (k != 0.0)
                acc = fma(k, M[d][e], acc);
            }
          A[tri8(i, j)] = acc;
        }
    }
    else if constexpr (AFFINE_PATH != 2)
    {
#pragma unroll
      for (int q = 0; q < 36; ++q)
        A[q] = 0.0;
      using G2 = Gauss01<2>;
      // (a rolled loop over the points: unrolled, the scheduler interleaves several points and the 36 accumulators +
      // 24 coefficients + 24 gradient components no longer fit 256 registers)
#pragma unroll 1
      for (int q = 0; q < 8; ++q)
          {
            const int qx = q & 1, qy = (q >> 1) & 1, qz = q >> 2;
            const double xi = qx ? G2::P[1] : G2::P[0], eta = qy ? G2::P[1] : G2::P[0], zeta = qz ? G2::P[1] : G2::P[0];
            double j0[3], j1[3], j2[3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
            {
              j0[r] = fma(c[7][r], eta * zeta, fma(c[5][r], zeta, fma(c[3][r], eta, c[1][r])));
              j1[r] = fma(c[7][r], xi * zeta, fma(c[6][r], zeta, fma(c[3][r], xi, c[2][r])));
              j2[r] = fma(c[7][r], xi * eta, fma(c[6][r], eta, fma(c[5][r], xi, c[4][r])));
            }
            double C0[3], C1[3], C2[3];
            cross3(j1, j2, C0);
            cross3(j2, j0, C1);
            cross3(j0, j1, C2);
            const double det = j0[0] * C0[0] + j0[1] * C0[1] + j0[2] * C0[2];
            const double s = 0.125 * c0 / fabs(det);
            double G[8][3];
#pragma unroll
            for (int i = 0; i < 8; ++i)
            {
              const double lx = (i & 1) ? xi : 1.0 - xi, ly = (i & 2) ? eta : 1.0 - eta, lz = (i & 4) ? zeta : 1.0 - zeta;
              const double d0 = hat_sign(i & 1) * ly * lz, d1 = hat_sign(i & 2) * lx * lz, d2 = hat_sign(i & 4) * lx * ly;
#pragma unroll
              for (int r = 0; r < 3; ++r)
                G[i][r] = fma(C2[r], d2, fma(C1[r], d1, C0[r] * d0));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
            {
              const double h0 = s * G[i][0], h1 = s * G[i][1], h2 = s * G[i][2];
#pragma unroll
              for (int j = i; j < 8; ++j)
                A[tri8(i, j)] = fma(h2, G[j][2], fma(h1, G[j][1], fma(h0, G[j][0], A[tri8(i, j)])));
            }
          }
    }
    // scatter: rows of this block that are not masked, unmasked columns
    const uint32_t ow[16] = {cur[2].x, cur[2].y, cur[2].z, cur[2].w, cur[3].x, cur[3].y, cur[3].z, cur[3].w,
                             cur[4].x, cur[4].y, cur[4].z, cur[4].w, cur[5].x, cur[5].y, cur[5].z, cur[5].w};
#pragma unroll
    for (int i = 0; i < 8; ++i)
    {
      const int r = v[i] & DOF_MASK;
      const bool mine = r >= r0 && r < r1 && !(v[i] >> MASK_SHIFT);
      if (!mine)
        continue;
      const int base = s_rowlo[r - r0];
#pragma unroll
      for (int j = 0; j < 8; ++j)
      {
        if (v[j] >> MASK_SHIFT)
          continue;
        const int off = int((ow[(i * 8 + j) >> 2] >> (8 * ((i * 8 + j) & 3))) & 0xff);
        __hip_atomic_fetch_add(s_vals + base + off, A[i <= j ? tri8(i, j) : tri8(j, i)], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    if (has_next)
    {
#pragma unroll
      for (int i = 0; i < 6; ++i)
        cur[i] = nxt[i];
    }
  }
  __syncthreads();
  MPCX_WRITE_OUT(a, nnz0, nnzb, s_vals, tid, NT);
}

// vector: scalar Q1 source term c * f v dx over an NQ1^3 Gauss rule, one thread per hexahedron, owner-computes row
// blocks (the plan of vector_cube_own_kernel with cube_verts = the cell dofmap).  The points are walked line by line
// in X: the map and two Jacobian columns are linear along a line (three fma each per point), the basis sums are
// sum-factorised (two fma per point, four per line, eight per plane).
template <int FN, int NQ1>
__global__ void __launch_bounds__(VCUBE_OWN_THREADS) vector_hex_own_kernel(mpcx_vector_args_t a)
{
  extern __shared__ __align__(16) unsigned char smem[];
  double* s_b = reinterpret_cast<double*>(smem);
  const int NT = blockDim.x;
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const int tid = threadIdx.x;
  const int r0 = b < nb ? a.plan.block_row0[b] : 0, r1 = b < nb ? a.plan.block_row0[b + 1] : 0;
  const int64_t h0 = b < nb ? a.own_hoff[b] : 0, h1 = b < nb ? a.own_hoff[b + 1] : 0;
  const int nown = r1 - r0, nhalo = int(h1 - h0);
  for (int i = tid; i < nown + nhalo; i += NT)
    s_b[i] = 0.0;
  fastmath_init_lds(); // ends in a barrier
  if (b >= nb)
    return;
  using GQ = Gauss01<NQ1>;
  const double cst = a.constants ? a.constants[0] : 1.0;
  [[maybe_unused]] FmConsts FK;
  if constexpr (FN == 1)
    FK = g_fm_consts; // uniform loads: the polynomial coefficients live in scalar registers
  const int64_t e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  const int32_t* __restrict__ ents = a.plan.block_ents;
  for (int64_t t = e0 + tid; t < e1; t += NT)
  {
    const int64_t cell = ents[t];
    double c[8][3];
    {
      int32_t v[8];
      const uint4* p = reinterpret_cast<const uint4*>(a.cube_verts + cell * 8);
      const uint4 w0 = p[0], w1 = p[1];
      v[0] = w0.x, v[1] = w0.y, v[2] = w0.z, v[3] = w0.w, v[4] = w1.x, v[5] = w1.y, v[6] = w1.z, v[7] = w1.w;
      double X[8][3];
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k)
          X[i][k] = a.x[3 * int64_t(v[i]) + k];
      hex_coefficients(X, c);
    }
    if constexpr (FN == 1)
    {
      // coordinates relative to the centre of the benchmark function's Gaussian (see ElementOp::tabulate)
      c[0][0] -= 0.9;
      c[0][1] -= 0.5;
      c[0][2] -= 0.1;
    }
    double be[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      be[i] = 0.0;
    // an axis-aligned box (exact zeros in the mixed coefficients, diagonal edge vectors: every box grid): the Gauss points
    // are a tensor grid and the benchmark's right-hand side a sum of products of univariate factors (see box14_source_fn1):
    // NQ1 sines and 3 NQ1 exponentials per cell instead of NQ1^3 of each, a constant determinant
    [[maybe_unused]] bool box = false;
    if constexpr (FN == 1)
      box = g_box14_enable && c[3][0] == 0.0 && c[3][1] == 0.0 && c[3][2] == 0.0 && c[5][0] == 0.0 && c[5][1] == 0.0
            && c[5][2] == 0.0 && c[6][0] == 0.0 && c[6][1] == 0.0 && c[6][2] == 0.0 && c[7][0] == 0.0 && c[7][1] == 0.0
            && c[7][2] == 0.0 && c[1][1] == 0.0 && c[1][2] == 0.0 && c[2][0] == 0.0 && c[2][2] == 0.0 && c[4][0] == 0.0
            && c[4][1] == 0.0;
    if (box)
    {
      if constexpr (FN == 1)
      {
        double gx[NQ1], xq[NQ1], gy[NQ1], sy[NQ1], gz[NQ1];
#pragma unroll
        for (int q = 0; q < NQ1; ++q)
        {
          const double x = fma(c[1][0], GQ::P[q], c[0][0]), y = fma(c[2][1], GQ::P[q], c[0][1]), z = fma(c[4][2], GQ::P[q], c[0][2]);
          gx[q] = fast_exp_nonpos_k(-(x * x) * (1.0 / 0.02), FK);
          gy[q] = fast_exp_nonpos_k(-(y * y) * (1.0 / 0.02), FK);
          gz[q] = fast_exp_nonpos_k(-(z * z) * (1.0 / 0.02), FK);
          sy[q] = fast_sinpi_k(fma(5.0, y, 2.5), FK);
          xq[q] = x + 0.9;
        }
        const double vol = cst * fabs(c[1][0] * c[2][1] * c[4][2]);
#pragma unroll
        for (int qz = 0; qz < NQ1; ++qz)
        {
          const double zeta = GQ::P[qz];
          double V[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
#pragma unroll
          for (int qy = 0; qy < NQ1; ++qy)
          {
            const double eta = GQ::P[qy];
            const double gyz = gy[qy] * gz[qz];
            double T0 = 0.0, T1 = 0.0;
#pragma unroll
            for (int qx = 0; qx < NQ1; ++qx)
            {
              const double xi = GQ::P[qx];
              const double F = fma(xq[qx], sy[qy], gx[qx] * gyz) * ((GQ::W[qx] * GQ::W[qy] * GQ::W[qz]) * vol);
              T0 = fma(F, 1.0 - xi, T0);
              T1 = fma(F, xi, T1);
            }
            V[0][0] = fma(T0, 1.0 - eta, V[0][0]);
            V[0][1] = fma(T1, 1.0 - eta, V[0][1]);
            V[1][0] = fma(T0, eta, V[1][0]);
            V[1][1] = fma(T1, eta, V[1][1]);
          }
#pragma unroll
          for (int by = 0; by < 2; ++by)
#pragma unroll
            for (int bx = 0; bx < 2; ++bx)
            {
              be[by * 2 + bx] = fma(V[by][bx], 1.0 - zeta, be[by * 2 + bx]);
              be[4 + by * 2 + bx] = fma(V[by][bx], zeta, be[4 + by * 2 + bx]);
            }
        }
      }
    }
    else
#pragma unroll
    for (int qz = 0; qz < NQ1; ++qz)
    {
      const double zeta = GQ::P[qz];
      double B0[3], B1[3]; // j1 = B0 + B1 X
#pragma unroll
      for (int r = 0; r < 3; ++r)
      {
        B0[r] = fma(c[6][r], zeta, c[2][r]);
        B1[r] = fma(c[7][r], zeta, c[3][r]);
      }
      double V[2][2] = {{0.0, 0.0}, {0.0, 0.0}}; // V[by][bx]: sums of this plane
#pragma unroll
      for (int qy = 0; qy < NQ1; ++qy)
      {
        const double eta = GQ::P[qy];
        double A0[3], A1[3], D0[3], D1[3]; // x = A0 + A1 X (A1 = j0), j2 = D0 + D1 X
#pragma unroll
        for (int r = 0; r < 3; ++r)
        {
          A0[r] = fma(c[6][r], eta * zeta, fma(c[4][r], zeta, fma(c[2][r], eta, c[0][r])));
          A1[r] = fma(c[7][r], eta * zeta, fma(c[5][r], zeta, fma(c[3][r], eta, c[1][r])));
          D0[r] = fma(c[6][r], eta, c[4][r]);
          D1[r] = fma(c[7][r], eta, c[5][r]);
        }
        double T0 = 0.0, T1 = 0.0;
#pragma unroll
        for (int qx = 0; qx < NQ1; ++qx)
        {
          const double xi = GQ::P[qx];
          double x[3], j1[3], j2[3], cr[3];
#pragma unroll
          for (int r = 0; r < 3; ++r)
          {
            x[r] = fma(A1[r], xi, A0[r]);
            j1[r] = fma(B1[r], xi, B0[r]);
            j2[r] = fma(D1[r], xi, D0[r]);
          }
          cross3(j1, j2, cr);
          const double det = A1[0] * cr[0] + A1[1] * cr[1] + A1[2] * cr[2];
          double f;
          if constexpr (FN == 1)
          {
            const double tt = fma(5.0, x[1], 2.5); // 5 y with y = x[1] + 0.5
            f = (x[0] + 0.9) * fast_sinpi_k(tt, FK)
                + fast_exp_nonpos_k(-(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]) * (1.0 / 0.02), FK);
          }
          else
            f = eval_fn(FN >= 0 ? FN : a.kernel.fn_id, x, 0, a.constants);
          const double F = f * ((GQ::W[qx] * GQ::W[qy] * GQ::W[qz]) * cst) * fabs(det);
          T0 = fma(F, 1.0 - xi, T0);
          T1 = fma(F, xi, T1);
        }
        V[0][0] = fma(T0, 1.0 - eta, V[0][0]);
        V[0][1] = fma(T1, 1.0 - eta, V[0][1]);
        V[1][0] = fma(T0, eta, V[1][0]);
        V[1][1] = fma(T1, eta, V[1][1]);
      }
#pragma unroll
      for (int by = 0; by < 2; ++by)
#pragma unroll
        for (int bx = 0; bx < 2; ++bx)
        {
          be[by * 2 + bx] = fma(V[by][bx], 1.0 - zeta, be[by * 2 + bx]);
          be[4 + by * 2 + bx] = fma(V[by][bx], zeta, be[4 + by * 2 + bx]);
        }
    }
    // LDS positions of the eight vertices (read after the quadrature: eight registers less across it)
    int32_t w[8];
    {
      const uint4* p = reinterpret_cast<const uint4*>(a.own_lmap + cell * 8);
      const uint4 w0 = p[0], w1 = p[1];
      w[0] = w0.x, w[1] = w0.y, w[2] = w0.z, w[3] = w0.w, w[4] = w1.x, w[5] = w1.y, w[6] = w1.z, w[7] = w1.w;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (!(w[i] >> MASK_SHIFT))
        __hip_atomic_fetch_add(s_b + (w[i] & DOF_MASK), be[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  for (int i = tid; i < nown; i += NT)
    a.b[MPCX_ROW_POS(a, r0 + i)] += s_b[i];
  for (int i = tid; i < nhalo; i += NT)
    a.own_spill[h0 + i] = s_b[nown + i];
}

// ---------------------------------------------------------------------------------------------------------
// Tetrahedral clusters that are parallelepipeds (the six tets of a cube of a box mesh, sheared or not): with local
// vertex b at corner (b & 1, b >> 1 & 1, b >> 2 & 1) of the reference cube and x = x0 + J X, the sum of the six element
// tensors is a fixed linear combination of the six entries of M = c C^T C / |det J| (C = cofactor matrix of J):
//     A_ij = sum_{d <= e} M_de K_de(i, j),   K_de(i, j) = sum over the tets holding i and j of
//                                                        (g_i^d g_j^e + [d != e] g_i^e g_j^d) / 6,
// g = the gradients of the barycentric coordinates on the reference Kuhn cube (entries 0, +-1).  Row blocks all of whose
// clusters pass hex_is_parallelepiped at set-up (mpcx_hex_slot_shapes on the records; mpcx_matrix_args_t::cube_flags
// bit 0) are launched with this kernel: ~100 registers instead of 236, four waves per SIMD instead of two cover the
// record -> coordinates round trip that the general kernel hides with its software pipeline.
// ---------------------------------------------------------------------------------------------------------
struct FanAffineTable
{
  double k[6][8][8]; // [d <= e packed: 00 01 02 11 12 22][i][j]
};
__host__ __device__ constexpr int sym6(int d, int e) { return d == 0 ? e : (d == 1 ? 2 + e : 5); }
constexpr FanAffineTable make_fan_affine_table()
{
  FanAffineTable T{};
  for (int t = 0; t < 6; ++t)
  {
    // the tet's vertices along the path 0 -> 7 (one more bit per step); gradient of lambda of the vertex with k bits:
    // k = 0: -e_a1, k = 1: e_a1 - e_a2, k = 2: e_a2 - e_a3, k = 3: e_a3 (a_k = the axis added at step k)
    int path[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; ++i)
    {
      const int v = fan_vertex(t, i);
      const int bits = (v & 1) + ((v >> 1) & 1) + ((v >> 2) & 1);
      path[bits] = v;
    }
    int axis[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k)
    {
      const int diff = path[k + 1] ^ path[k];
      axis[k] = diff == 1 ? 0 : (diff == 2 ? 1 : 2);
    }
    double g[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    g[0][axis[0]] = -1.0;
    g[1][axis[0]] = 1.0;
    g[1][axis[1]] = -1.0;
    g[2][axis[1]] = 1.0;
    g[2][axis[2]] = -1.0;
    g[3][axis[2]] = 1.0;
    for (int p = 0; p < 4; ++p)
      for (int q = 0; q < 4; ++q)
        for (int d = 0; d < 3; ++d)
          for (int e = d; e < 3; ++e)
          {
            double v = g[p][d] * g[q][e];
            if (d != e)
              v += g[p][e] * g[q][d];
            T.k[sym6(d, e)][path[p]][path[q]] += v / 6.0;
          }
  }
  return T;
}
static constexpr FanAffineTable FAN_AFFINE = make_fan_affine_table();

constexpr int CUBE_AFFINE_MAX_THREADS = 1024;
template <bool NARROW>
__global__ void __launch_bounds__(CUBE_AFFINE_MAX_THREADS) matrix_cube_affine_kernel(mpcx_matrix_args_t a)
{
  const int NT = blockDim.x;
  extern __shared__ __align__(16) unsigned char smem[];
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (b >= nb)
    return;
  const int tid = threadIdx.x;
  const int bb = a.cube_block_ids ? a.cube_block_ids[b] : b;
  const int r0 = a.plan.block_row0[bb], r1 = a.plan.block_row0[bb + 1];
  const int nrow = r1 - r0;
  const int64_t nnz0 = a.rowptr[r0];
  const int nnzb = int(a.rowptr[r1] - nnz0);
  double* s_vals = reinterpret_cast<double*>(smem);
  int32_t* s_rowlo = reinterpret_cast<int32_t*>(s_vals + a.plan.max_nnz);
  const double c0 = a.constants ? a.constants[0] : 1.0;
  constexpr int NW = NARROW ? 4 : 6;
  const uint4* __restrict__ recs = static_cast<const uint4*>(a.cube_recs);
  const int64_t e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  const int32_t* __restrict__ ridx = a.cube_rec_index; // one record per cluster: slot -> record (include/mpcx.h)
  uint4 cur[NW];
  int64_t t = e0 + tid;
  if (t < e1)
  {
    const int64_t ri = ridx ? int64_t(ridx[t]) : t;
#pragma unroll
    for (int i = 0; i < NW; ++i)
      cur[i] = recs[ri * NW + i];
  }
  for (int i = tid; i < nnzb; i += NT)
    s_vals[i] = 0.0;
  for (int rl = tid; rl < nrow; rl += NT)
    s_rowlo[rl] = int(a.rowptr[r0 + rl] - nnz0);
  __syncthreads();
  for (; t < e1; t += NT)
  {
    const int32_t v[8] = {int32_t(cur[0].x), int32_t(cur[0].y), int32_t(cur[0].z), int32_t(cur[0].w),
                          int32_t(cur[1].x), int32_t(cur[1].y), int32_t(cur[1].z), int32_t(cur[1].w)};
    // a parallelepiped is spanned from vertex 0 by the vertices 1, 2, 4
    double j0[3], j1[3], j2[3];
    {
      const int64_t n0 = v[0] & DOF_MASK, n1 = v[1] & DOF_MASK, n2 = v[2] & DOF_MASK, n4 = v[4] & DOF_MASK;
#pragma unroll
      for (int r = 0; r < 3; ++r)
      {
        const double x0 = a.x[3 * n0 + r];
        j0[r] = a.x[3 * n1 + r] - x0;
        j1[r] = a.x[3 * n2 + r] - x0;
        j2[r] = a.x[3 * n4 + r] - x0;
      }
    }
    uint32_t ow[4 * (NW - 2)];
#pragma unroll
    for (int q = 0; q < NW - 2; ++q)
    {
      ow[4 * q] = cur[2 + q].x, ow[4 * q + 1] = cur[2 + q].y, ow[4 * q + 2] = cur[2 + q].z, ow[4 * q + 3] = cur[2 + q].w;
    }
    if (t + NT < e1)
    {
      const int64_t ri = ridx ? int64_t(ridx[t + NT]) : t + NT;
#pragma unroll
      for (int i = 0; i < NW; ++i)
        cur[i] = recs[ri * NW + i];
    }
    double C0[3], C1[3], C2[3];
    cross3(j1, j2, C0);
    cross3(j2, j0, C1);
    cross3(j0, j1, C2);
    const double det = j0[0] * C0[0] + j0[1] * C0[1] + j0[2] * C0[2];
    const double s = c0 / fabs(det);
    const double M[6] = {s * (C0[0] * C0[0] + C0[1] * C0[1] + C0[2] * C0[2]), s * (C0[0] * C1[0] + C0[1] * C1[1] + C0[2] * C1[2]),
                         s * (C0[0] * C2[0] + C0[1] * C2[1] + C0[2] * C2[2]), s * (C1[0] * C1[0] + C1[1] * C1[1] + C1[2] * C1[2]),
                         s * (C1[0] * C2[0] + C1[1] * C2[1] + C1[2] * C2[2]), s * (C2[0] * C2[0] + C2[1] * C2[1] + C2[2] * C2[2])};
    int base[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
    {
      const int r = v[i] & DOF_MASK;
      const bool mine = r >= r0 && r < r1 && !(v[i] >> MASK_SHIFT);
      base[i] = mine ? s_rowlo[mine ? r - r0 : 0] : -1;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = i; j < 8; ++j)
      {
        if (!fan_coupled(i, j))
          continue;
        double val = 0.0;
#pragma unroll
        for (int m = 0; m < 6; ++m)
        {
          const double k = FAN_AFFINE.k[m][i][j];
          if (k != 0.0)
            val = fma(k, M[m], val);
        }
        if (base[i] >= 0 && !(v[j] >> MASK_SHIFT))
        {
          int off;
          if constexpr (NARROW)
          {
            const int p = fan_pair_index(i, j);
            off = int((ow[p >> 3] >> (4 * (p & 7))) & 0xf);
          }
          else
            off = int((ow[(i * 8 + j) >> 2] >> (8 * ((i * 8 + j) & 3))) & 0xff);
          __hip_atomic_fetch_add(s_vals + base[i] + off, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (i != j && base[j] >= 0 && !(v[i] >> MASK_SHIFT))
        {
          int off;
          if constexpr (NARROW)
          {
            const int p = fan_pair_index(j, i);
            off = int((ow[p >> 3] >> (4 * (p & 7))) & 0xf);
          }
          else
            off = int((ow[(j * 8 + i) >> 2] >> (8 * ((j * 8 + i) & 3))) & 0xff);
          __hip_atomic_fetch_add(s_vals + base[j] + off, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
  }
  __syncthreads();
  MPCX_WRITE_OUT(a, nnz0, nnzb, s_vals, tid, NT);
}

// ---------------------------------------------------------------------------------------------------------
// matrix: P1 vector elasticity (bs = 3) over parallelepiped clusters, closed form.  The six element tensors of a
// cluster hold 6 * 144 = 864 entries but only 46 coupled vertex pairs * 9 = 414 distinct matrix entries.  Summing the
// tensors tet by tet needs the gradients of every tet and 3 x 3 accumulators for all pairs of the current faces: 256
// registers + scratch, slower than the per-cell row-pair kernel (rounds 2 and 3).  On a parallelepiped the sums
//     Q_ij^{ab} = sum_t |T_t| g_i^a g_j^b = (C Kp(i, j) C^T)^{ab} / |det J|,   Kp_de(i, j) = sum_t ghat_i^d ghat_j^e / 6
// (C = cofactor matrix of J, ghat = barycentric gradients on the reference Kuhn cube: a constant table) need nothing
// per tet, and
//     A[(i,a),(j,b)] = mu Q_ij^{ba} + lambda Q_ij^{ab} + delta_ab mu tr Q_ij,   A[(j,b),(i,a)] = the same value
// is scattered pair by pair: nothing stays live.  Clusters that are not parallelepipeds never get here: the plan leaves
// their cells to the per-cell kernel (cube_flags bit 0 is required).  Same records as the scalar kernel
// (mpcx_cube_records with bs = 3: mask of component c in bit 28 + c of the vertex id, offsets counted in column blocks).
// ---------------------------------------------------------------------------------------------------------
struct FanCoupled
{
  bool c[8][8];
};
constexpr FanCoupled make_fan_coupled()
{
  FanCoupled F{};
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j)
      F.c[i][j] = fan_coupled(i, j);
  return F;
}
static constexpr FanCoupled FAN_COUPLED = make_fan_coupled(); // (a table: look-ups with constant indices always fold)
struct FanAffineTable9
{
  double k[9][8][8]; // [d * 3 + e][i][j] = sum over the tets holding i and j of ghat_i^d ghat_j^e / 6
};
constexpr FanAffineTable9 make_fan_affine_table9()
{
  FanAffineTable9 T{};
  for (int t = 0; t < 6; ++t)
  {
    int path[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; ++i)
    {
      const int v = fan_vertex(t, i);
      path[(v & 1) + ((v >> 1) & 1) + ((v >> 2) & 1)] = v;
    }
    int axis[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k)
    {
      const int diff = path[k + 1] ^ path[k];
      axis[k] = diff == 1 ? 0 : (diff == 2 ? 1 : 2);
    }
    double g[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    g[0][axis[0]] = -1.0;
    g[1][axis[0]] = 1.0;
    g[1][axis[1]] = -1.0;
    g[2][axis[1]] = 1.0;
    g[2][axis[2]] = -1.0;
    g[3][axis[2]] = 1.0;
    for (int p = 0; p < 4; ++p)
      for (int q = 0; q < 4; ++q)
        for (int d = 0; d < 3; ++d)
          for (int e = 0; e < 3; ++e)
            T.k[d * 3 + e][path[p]][path[q]] += g[p][d] * g[q][e] / 6.0;
  }
  return T;
}
static constexpr FanAffineTable9 FAN_AFFINE9 = make_fan_affine_table9();
constexpr int CUBE_EL_THREADS = 1024;

// One thread per (cluster, local row vertex I) pair whose three rows lie in the row block (the row-pair plan of
// matrix_rowpair_kernel over the clusters: plan.row_pairs = 1, pair id = cluster * 8 + I, pairs of a block ordered by I
// so that a wave runs one unrolled row body, round-robin over the row nodes).  Every lane keeps what it computes: a
// thread per (block, cluster) slot masks the rows outside the block, and with ~70 nodes per 74 KB block half the lanes of
// every LDS instruction are idle -- 1.1 ms, no better than the per-cell row-pair kernel (1.0 ms) although it issues half
// the scatter-adds.  Records: ONE per cluster (cube_recs[cluster]: mpcx_cube_records over slots = clusters).
__global__ void __launch_bounds__(CUBE_EL_THREADS) matrix_cube_elasticity_rowpair_kernel(mpcx_matrix_args_t a)
{
  constexpr int BS = 3;
  const int NT = blockDim.x;
  extern __shared__ __align__(16) unsigned char smem[];
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (b >= nb)
    return;
  const int tid = threadIdx.x;
  const int r0 = a.plan.block_row0[b], r1 = a.plan.block_row0[b + 1];
  const int nrow = r1 - r0;
  const int64_t nnz0 = a.rowptr[r0];
  const int nnzb = int(a.rowptr[r1] - nnz0);
  double* s_vals = reinterpret_cast<double*>(smem);
  int32_t* s_rowlo = reinterpret_cast<int32_t*>(s_vals + a.plan.max_nnz);
  const double mu = a.constants[0], lmbda = a.constants[1];
  const CubeRec* __restrict__ recs = static_cast<const CubeRec*>(a.cube_recs);
  for (int i = tid; i < nnzb; i += NT)
    s_vals[i] = 0.0;
  for (int rl = tid; rl < nrow; rl += NT)
    s_rowlo[rl] = int(a.rowptr[r0 + rl] - nnz0);
  __syncthreads();
  const int64_t e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  const uint32_t* __restrict__ pairs = reinterpret_cast<const uint32_t*>(a.plan.block_ents);
  for (int64_t t = e0 + tid; t < e1; t += NT)
  {
    const uint32_t id = pairs[t];
    const int64_t e = id >> 3;
    const int i = int(id & 7u);
    const uint4* p = reinterpret_cast<const uint4*>(recs + e);
    const uint4 q0 = p[0], q1 = p[1];
    const int32_t v[8] = {int32_t(q0.x), int32_t(q0.y), int32_t(q0.z), int32_t(q0.w),
                          int32_t(q1.x), int32_t(q1.y), int32_t(q1.z), int32_t(q1.w)};
    const uint2 orow = reinterpret_cast<const uint2*>(recs + e)[4 + i]; // off[i * 8 .. i * 8 + 7]
    double j0[3], j1[3], j2[3];
    {
      const int64_t n0 = v[0] & DOF_MASK, n1 = v[1] & DOF_MASK, n2 = v[2] & DOF_MASK, n4 = v[4] & DOF_MASK;
#pragma unroll
      for (int r = 0; r < 3; ++r)
      {
        const double x0 = a.x[3 * n0 + r];
        j0[r] = a.x[3 * n1 + r] - x0;
        j1[r] = a.x[3 * n2 + r] - x0;
        j2[r] = a.x[3 * n4 + r] - x0;
      }
    }
    double C[3][3]; // C[d] = d-th cofactor column
    cross3(j1, j2, C[0]);
    cross3(j2, j0, C[1]);
    cross3(j0, j1, C[2]);
    const double det = j0[0] * C[0][0] + j0[1] * C[0][1] + j0[2] * C[0][2];
    const double inv = 1.0 / fabs(det);
    const double smu = mu * inv, sla = lmbda * inv;
#pragma unroll
    for (int I = 0; I < 8; ++I)
    {
      if (i != I)
        continue;
      const int mi = v[I] >> MASK_SHIFT;
      const int ri = (v[I] & DOF_MASK) * BS - r0;
      int base[3];
#pragma unroll
      for (int c = 0; c < 3; ++c)
        base[c] = s_rowlo[ri + c];
#pragma unroll
      for (int J = 0; J < 8; ++J)
      {
        if (!FAN_COUPLED.c[I][J])
          continue;
        // T[e][r] = sum_d Kp_de(I, J) C[d][r], Q[r][s] = sum_e T[e][r] C[e][s]  (= |det| sum_t |T_t| g_I^r g_J^s)
        double T[3][3];
#pragma unroll
        for (int e2 = 0; e2 < 3; ++e2)
#pragma unroll
          for (int r = 0; r < 3; ++r)
          {
            double acc = 0.0;
#pragma unroll
            for (int d = 0; d < 3; ++d)
            {
              const double k = FAN_AFFINE9.k[d * 3 + e2][I][J];
              if (k != 0.0)
                acc = fma(k, C[d][r], acc);
            }
            T[e2][r] = acc;
          }
        double Q[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c)
            Q[r][c] = fma(T[2][r], C[2][c], fma(T[1][r], C[1][c], T[0][r] * C[0][c]));
        const double tr = smu * (Q[0][0] + Q[1][1] + Q[2][2]);
        const int off = int(((J < 4 ? orow.x : orow.y) >> (8 * (J & 3))) & 0xff) * BS;
        const int mj = v[J] >> MASK_SHIFT;
#pragma unroll
        for (int ca = 0; ca < 3; ++ca)
        {
          if ((mi >> ca) & 1)
            continue;
#pragma unroll
          for (int cb = 0; cb < 3; ++cb)
          {
            if ((mj >> cb) & 1)
              continue;
            const double val = fma(smu, Q[cb][ca], sla * Q[ca][cb]) + (ca == cb ? tr : 0.0);
            __hip_atomic_fetch_add(s_vals + base[ca] + off + cb, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
  }
  __syncthreads();
  MPCX_WRITE_OUT(a, nnz0, nnzb, s_vals, tid, NT);
}

// ---------------------------------------------------------------------------------------------------------
// P2 stiffness on parallelepiped clusters, closed form.  A cluster carries 27 dofs: the eight vertices (local 0..7 = the
// cube corners) and the 19 edges between coupled vertices (local 8 + rank of the pair (a < b) among the coupled pairs in
// row-major order).  Two local dofs couple when a tet holds both: 393 of the 729 ordered pairs -- six element tensors
// scattered one by one cost 600 scatter-adds.  With M = c C^T C / |det J| (six entries, as for P1)
//     A_IJ = sum_m M_m K_m(I, J),   K_m(I, J) = sum over the common tets of the reference integrals of
//                                                d_d phi_I d_e phi_J (+ the transposed product for d != e),
// phi = the P2 basis in barycentric coordinates: vertex a: (4 l_a - 1) g_a, edge (a, b): 4 (l_a g_b + l_b g_a), g = the
// constant barycentric gradients on the reference Kuhn cube; int l_a l_b = V (1 + delta_ab) / 20, int l_a = V / 4, V = 1/6.
// ---------------------------------------------------------------------------------------------------------
struct P2FanTables
{
  int ea[19], eb[19];     // vertices of edge dof 8 + k
  int edge_of[8][8];      // local dof of the edge (a, b), -1 if a and b share no tet
  unsigned tets[27];      // bit t: tet t holds the dof
  double k[6][27][27];    // K_m(I, J)
  bool coupled[27][27];
  int row_start[28];      // packed position of the first coupled column of row I (rows in order, columns ascending)
  int rank[27][27];       // position of column J among the coupled columns of row I
};
constexpr P2FanTables make_p2_fan_tables()
{
  P2FanTables T{};
  int ne = 0;
  for (int a = 0; a < 8; ++a)
    for (int b = 0; b < 8; ++b)
      T.edge_of[a][b] = -1;
  for (int a = 0; a < 8; ++a)
    for (int b = a + 1; b < 8; ++b)
      if (fan_coupled(a, b))
      {
        T.ea[ne] = a;
        T.eb[ne] = b;
        T.edge_of[a][b] = T.edge_of[b][a] = 8 + ne;
        ++ne;
      }
  for (int I = 0; I < 27; ++I)
    T.tets[I] = 0u;
  for (int t = 0; t < 6; ++t)
  {
    // vertices along the path 0 -> 7, gradients of their barycentric coordinates (see make_fan_affine_table)
    int path[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; ++i)
    {
      const int v = fan_vertex(t, i);
      path[(v & 1) + ((v >> 1) & 1) + ((v >> 2) & 1)] = v;
    }
    int axis[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k)
    {
      const int diff = path[k + 1] ^ path[k];
      axis[k] = diff == 1 ? 0 : (diff == 2 ? 1 : 2);
    }
    double g[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    g[0][axis[0]] = -1.0;
    g[1][axis[0]] = 1.0;
    g[1][axis[1]] = -1.0;
    g[2][axis[1]] = 1.0;
    g[2][axis[2]] = -1.0;
    g[3][axis[2]] = 1.0;
    // the ten P2 functions of the tet: local dof, and gradient = sum_p (alpha_p + sum_q beta_pq l_q) g_p
    //   vertex p: alpha_p = -1, beta_pp = 4;  edge (p, q): beta_pq = beta_qp = 4
    int dof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    double alpha[10][4] = {};
    double beta[10][4][4] = {};
    int n = 0;
    for (int p = 0; p < 4; ++p, ++n)
    {
      dof[n] = path[p];
      alpha[n][p] = -1.0;
      beta[n][p][p] = 4.0;
    }
    for (int p = 0; p < 4; ++p)
      for (int q = p + 1; q < 4; ++q, ++n)
      {
        dof[n] = T.edge_of[path[p]][path[q]];
        beta[n][p][q] = 4.0; // l_q g_p
        beta[n][q][p] = 4.0; // l_p g_q
      }
    for (int i = 0; i < 10; ++i)
      T.tets[dof[i]] |= 1u << t;
    const double V = 1.0 / 6.0;
    for (int i = 0; i < 10; ++i)
      for (int j = 0; j < 10; ++j)
        for (int p = 0; p < 4; ++p)
          for (int q = 0; q < 4; ++q)
          {
            // int (alpha_ip + sum_r beta_ipr l_r) (alpha_jq + sum_s beta_jqs l_s)
            double w = alpha[i][p] * alpha[j][q] * V;
            for (int r = 0; r < 4; ++r)
            {
              w += beta[i][p][r] * alpha[j][q] * V / 4.0 + alpha[i][p] * beta[j][q][r] * V / 4.0;
              for (int s2 = 0; s2 < 4; ++s2)
                w += beta[i][p][r] * beta[j][q][s2] * V * (r == s2 ? 2.0 : 1.0) / 20.0;
            }
            if (w == 0.0)
              continue;
            for (int d = 0; d < 3; ++d)
              for (int e = d; e < 3; ++e)
              {
                double v = g[p][d] * g[q][e];
                if (d != e)
                  v += g[p][e] * g[q][d];
                T.k[sym6(d, e)][dof[i]][dof[j]] += w * v;
              }
          }
  }
  int pos = 0;
  for (int I = 0; I < 27; ++I)
  {
    T.row_start[I] = pos;
    for (int J = 0; J < 27; ++J)
    {
      T.coupled[I][J] = (T.tets[I] & T.tets[J]) != 0u;
      T.rank[I][J] = pos - T.row_start[I];
      if (T.coupled[I][J])
        ++pos;
    }
  }
  T.row_start[27] = pos;
  return T;
}
static constexpr P2FanTables P2FAN = make_p2_fan_tables();
static_assert(P2FAN.row_start[27] == 393, "393 coupled pairs of P2 dofs per cluster");

// One record per cluster (P2CUBE_REC bytes):
//   [  0,  48)  double M[6]     C^T C / |det J| of the cluster's parallelepiped (00 01 02 11 12 22): the geometry, once per
//                               cluster instead of once per (cluster, row) unit
//   [ 48,  52)  uint32 mask     bit I: local dof I is a Dirichlet or slave dof (its row and column stay empty)
//   [ 52, 160)  int32 dof[27]
//   [160, 636)  uint8 off[..]   position of column dof J inside CSR row dof I; the coupled columns of row I in ascending
//                               order, every row starting at a multiple of four bytes (P2PACK.rowoff4)
constexpr int P2CUBE_REC = 640;
constexpr int P2CUBE_OFFS = 160;

// the tables the kernel reads through scalar loads: a wave works on ONE local row I at a time, so the coefficients of the
// entries of that row are wave-uniform (27 unrolled row bodies with the constants folded in were 120 KB of code: 20.8 ms)
struct P2Packed
{
  double k[393][6];       // K_m(I, J) of the packed entry (row-major over the coupled pairs)
  unsigned char col[396]; // its column J
  short rowq[28];         // first packed entry of row I
  short rowoff4[28];      // byte offset of row I's offsets inside the record's offset area (multiples of 4)
};
constexpr P2Packed make_p2_packed()
{
  P2Packed P{};
  int q = 0, o = 0;
  for (int I = 0; I < 27; ++I)
  {
    P.rowq[I] = short(q);
    P.rowoff4[I] = short(o);
    int deg = 0;
    for (int J = 0; J < 27; ++J)
      if (P2FAN.coupled[I][J])
      {
        for (int m = 0; m < 6; ++m)
          P.k[q][m] = P2FAN.k[m][I][J];
        P.col[q] = (unsigned char)J;
        ++q;
        ++deg;
      }
    o += (deg + 3) & ~3;
  }
  P.rowq[27] = short(q);
  P.rowoff4[27] = short(o);
  return P;
}
static constexpr P2Packed P2PACK_HOST = make_p2_packed();
static_assert(P2PACK_HOST.rowoff4[27] + P2CUBE_OFFS <= P2CUBE_REC, "record too small");
__constant__ P2Packed g_p2pack = make_p2_packed();

// set-up: the 27 dofs of every cluster from the P2 dofmaps of its six cells (any local vertex order inside a cell):
// vertex dofs from the cells' vertex slots, edge dofs from their edge slots (local edge e joins the local vertices
// TET_EDGE[e]: dolfinx_mpc_amd/mesh.py)
__global__ void p2_cluster_dofs_kernel(int64_t n, const int32_t* __restrict__ verts, const int32_t* __restrict__ fan_cells,
                                       const int32_t* __restrict__ x_dofmap, const int32_t* __restrict__ dofmap,
                                       int32_t* __restrict__ dofs27, int32_t* bad)
{
  const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c >= n)
    return;
  constexpr int TE[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
  int32_t v[8];
  for (int i = 0; i < 8; ++i)
    v[i] = verts[c * 8 + i];
  int32_t out[27];
  for (int i = 0; i < 27; ++i)
    out[i] = -1;
  for (int t = 0; t < 6; ++t)
  {
    const int64_t cell = fan_cells[c * 6 + t];
    int corner[4];
    for (int i = 0; i < 4; ++i)
    {
      const int32_t w = x_dofmap[cell * 4 + i];
      int b = -1;
      for (int q = 0; q < 8; ++q)
        if (v[q] == w)
          b = q;
      corner[i] = b;
      if (b >= 0)
        out[b] = dofmap[cell * 10 + i];
    }
    for (int e = 0; e < 6; ++e)
    {
      const int a0 = corner[TE[e][0]], b0 = corner[TE[e][1]];
      if (a0 < 0 || b0 < 0)
        continue;
      const int I = P2FAN.edge_of[a0][b0];
      if (I >= 0)
        out[I] = dofmap[cell * 10 + 4 + e];
    }
  }
  bool ok = true;
  for (int i = 0; i < 27; ++i)
  {
    ok &= out[i] >= 0;
    dofs27[c * 27 + i] = out[i];
  }
  if (!ok)
    atomicOr(bad, 1);
}

// set-up: the records; one thread per (cluster, local dof I)
__global__ void p2_cluster_records_kernel(int64_t n, const int32_t* __restrict__ verts, const int32_t* __restrict__ dofs27,
                                          const double* __restrict__ x, const int8_t* __restrict__ bc,
                                          const int8_t* __restrict__ is_slave, const mpcx_nnz_t* __restrict__ rowptr,
                                          const int32_t* __restrict__ cols, unsigned char* __restrict__ recs, int32_t* overflow)
{
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n * 27)
    return;
  const int64_t c = t / 27;
  const int I = int(t - c * 27);
  unsigned char* rec = recs + c * P2CUBE_REC;
  const int32_t* d = dofs27 + c * 27;
  if (I == 0)
  {
    const int64_t n0 = verts[c * 8 + 0], n1 = verts[c * 8 + 1], n2 = verts[c * 8 + 2], n4 = verts[c * 8 + 4];
    double j0[3], j1[3], j2[3];
    for (int r = 0; r < 3; ++r)
    {
      const double x0 = x[3 * n0 + r];
      j0[r] = x[3 * n1 + r] - x0;
      j1[r] = x[3 * n2 + r] - x0;
      j2[r] = x[3 * n4 + r] - x0;
    }
    double C0[3], C1[3], C2[3];
    cross3(j1, j2, C0);
    cross3(j2, j0, C1);
    cross3(j0, j1, C2);
    const double det = j0[0] * C0[0] + j0[1] * C0[1] + j0[2] * C0[2];
    const double s = 1.0 / fabs(det);
    double* M = reinterpret_cast<double*>(rec);
    M[0] = s * (C0[0] * C0[0] + C0[1] * C0[1] + C0[2] * C0[2]);
    M[1] = s * (C0[0] * C1[0] + C0[1] * C1[1] + C0[2] * C1[2]);
    M[2] = s * (C0[0] * C2[0] + C0[1] * C2[1] + C0[2] * C2[2]);
    M[3] = s * (C1[0] * C1[0] + C1[1] * C1[1] + C1[2] * C1[2]);
    M[4] = s * (C1[0] * C2[0] + C1[1] * C2[1] + C1[2] * C2[2]);
    M[5] = s * (C2[0] * C2[0] + C2[1] * C2[1] + C2[2] * C2[2]);
    uint32_t m = 0;
    for (int J = 0; J < 27; ++J)
      if ((bc && bc[d[J]]) || is_slave[d[J]])
        m |= 1u << J;
    *reinterpret_cast<uint32_t*>(rec + 48) = m;
    for (int q = P2CUBE_OFFS + P2PACK_HOST.rowoff4[27]; q < P2CUBE_REC; ++q)
      rec[q] = 0;
  }
  reinterpret_cast<int32_t*>(rec + 52)[I] = d[I];
  const int64_t lo = rowptr[d[I]], hi = rowptr[d[I] + 1];
  unsigned char* orow = rec + P2CUBE_OFFS + P2PACK_HOST.rowoff4[I];
  int p = 0;
  for (int J = 0; J < 27; ++J)
  {
    if (!P2FAN.coupled[I][J])
      continue;
    const int64_t pos = find_col(cols, lo, hi, d[J]);
    const int64_t o = pos < 0 ? 256 : pos - lo;
    if (o > 255)
      atomicOr(overflow, 1);
    orow[p++] = uint8_t(o);
  }
  while (p & 3)
    orow[p++] = 0;
}

// matrix: scalar P2 stiffness on parallelepiped clusters, closed form; one thread per (cluster, local dof I) pair whose
// row lies in the row block (row-pair plan over the clusters: pair id = cluster * 27 + I, pairs of a block ordered by I so
// that a wave works on few distinct rows).  Every lane keeps what it computes -- the per-cell kernel evaluates a cell in
// every block it touches and keeps 3.8 of its 10 rows on average (3.5e8 LDS wave instructions at 246^3 for 1.4e8 full ones).
// MEASURED AND NOT A DEFAULT: 22.2 ms at 246^3 against 14.1 ms for the per-cell kernel (see dolfinx_mpc_amd/dispatch.py).
constexpr int P2CUBE_MAX_THREADS = 1024;

__global__ void __launch_bounds__(P2CUBE_MAX_THREADS) matrix_p2_cube_kernel(mpcx_matrix_args_t a)
{
  const int NT = blockDim.x;
  extern __shared__ __align__(16) unsigned char smem[];
  const int nb = a.plan.num_blocks;
  const int per = (nb + 7) >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (b >= nb)
    return;
  const int tid = threadIdx.x;
  const int r0 = a.plan.block_row0[b], r1 = a.plan.block_row0[b + 1];
  const int nrow = r1 - r0;
  const int64_t nnz0 = a.rowptr[r0];
  const int nnzb = int(a.rowptr[r1] - nnz0);
  double* s_vals = reinterpret_cast<double*>(smem);
  int32_t* s_rowlo = reinterpret_cast<int32_t*>(s_vals + a.plan.max_nnz);
  const double c0 = a.constants ? a.constants[0] : 1.0;
  const unsigned char* __restrict__ recs = static_cast<const unsigned char*>(a.cube_recs);
  for (int i = tid; i < nnzb; i += NT)
    s_vals[i] = 0.0;
  for (int rl = tid; rl < nrow; rl += NT)
    s_rowlo[rl] = int(a.rowptr[r0 + rl] - nnz0);
  __syncthreads();
  const int64_t e0 = a.plan.block_ent_off[b], e1 = a.plan.block_ent_off[b + 1];
  const uint32_t* __restrict__ pairs = reinterpret_cast<const uint32_t*>(a.plan.block_ents);
  const int64_t span = ((e1 - e0 + NT - 1) / NT) * NT; // every lane of a wave takes part in the wave-level votes below
  for (int64_t t = e0 + tid; t < e0 + span; t += NT)
  {
    const bool live = t < e1;
    const uint32_t id = live ? pairs[t] : 0u;
    const uint32_t e = id / 27u;
    const int i = int(id - e * 27u);
    const unsigned char* rec = recs + int64_t(e) * P2CUBE_REC;
    double M[6];
    uint32_t mask = 0;
    int base = 0;
    bool todo = false;
    if (live)
    {
      const double2* pm = reinterpret_cast<const double2*>(rec);
      const double2 m0 = pm[0], m1 = pm[1], m2 = pm[2];
      M[0] = c0 * m0.x, M[1] = c0 * m0.y, M[2] = c0 * m1.x, M[3] = c0 * m1.y, M[4] = c0 * m2.x, M[5] = c0 * m2.y;
      mask = *reinterpret_cast<const uint32_t*>(rec + 48);
      todo = !((mask >> i) & 1); // a Dirichlet / slave row stays empty (cpp/assemble_matrix.cpp:513-525, :165-178)
      if (todo)
        base = s_rowlo[reinterpret_cast<const int32_t*>(rec + 52)[i] - r0];
    }
    // the lanes of a wave hold at most a few distinct local rows (the plan orders the pairs of a block by row): one
    // pass per distinct row, its coefficients through scalar loads
    for (;;)
    {
      const unsigned long long pending = __ballot(todo);
      if (pending == 0ull)
        break;
      const int lane0 = __ffsll((long long)pending) - 1;
      const int I = __builtin_amdgcn_readlane(i, lane0);
      const int q0 = g_p2pack.rowq[I], deg = g_p2pack.rowq[I + 1] - q0, o4 = g_p2pack.rowoff4[I];
      if (todo && i == I)
      {
        for (int c = 0; 4 * c < deg; ++c)
        {
          const uint32_t ow = *reinterpret_cast<const uint32_t*>(rec + P2CUBE_OFFS + o4 + 4 * c);
#pragma unroll
          for (int u = 0; u < 4; ++u)
          {
            const int q = 4 * c + u;
            if (q >= deg)
              break;
            const double* kk = g_p2pack.k[q0 + q];
            const int J = g_p2pack.col[q0 + q];
            const double val = fma(kk[5], M[5], fma(kk[4], M[4], fma(kk[3], M[3], fma(kk[2], M[2], fma(kk[1], M[1], kk[0] * M[0])))));
            if (!((mask >> J) & 1))
              __hip_atomic_fetch_add(s_vals + base + int((ow >> (8 * u)) & 0xff), val, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
        todo = false;
      }
    }
  }
  __syncthreads();
  MPCX_WRITE_OUT(a, nnz0, nnzb, s_vals, tid, NT);
}

} // namespace

// ---------------------------------------------------------------------------------------------------------
// Config 2's two cluster kernels in ONE launch (mpcx_assemble_fused, include/mpcx.h): the matrix kernel is
// HBM / LDS bound (VALU issue 0.30) and the vector kernel VALU bound (0.80), but launched one after the other they hardly overlap
// -- the matrix workgroups take 148 of a CU's 160 KB of LDS, so the vector workgroups wait.  Here a workgroup owns ONE row block
// for both: the LDS copy of its CSR rows (matrix_cube_affine_kernel<narrow records>: closed form on parallelepiped clusters) AND
// the LDS copy of its rows of b with their halo (vector_cube_own_kernel: owner-computes), 512 threads; the two workgroups a CU
// holds start with different halves (by workgroup parity), so that the memory phase of one runs under the arithmetic of the
// other.  Needs the two plans on the SAME row blocks: plan.block_row0 of the vector arguments is the full list, part_index[bb] =
// index of row block bb in the matrix launch (its slots: a.plan.block_ent_off) or -1 (row blocks of another record format /
// cluster shape: their matrix part is launched on its own).  The vector side does every row block.
// ---------------------------------------------------------------------------------------------------------
constexpr int FUSED_THREADS = 512;
// what the fused kernel reads of the two argument blocks (the full structs cost 200 scalar registers: 105 spilled)
struct FusedMatrix
{
  const mpcx_nnz_t* rowptr;
  double* vals;
  const double* x;
  const double* constants;
  const void* cube_recs;
  const int64_t* block_ent_off;
  int32_t max_nnz, max_rows, store_mode;
};
struct FusedVector
{
  double* b;
  const double* x;
  const double* constants;
  mpcx_kernel_t kernel;
  const int32_t* block_row0;
  const int64_t* block_ent_off;
  const int32_t* block_ents;
  const int32_t* cube_verts;
  const int32_t* own_lmap;
  const int64_t* own_hoff;
  double* own_spill;
  int32_t num_blocks;
};
template <int FN>
__global__ void __launch_bounds__(FUSED_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) fused_cube_kernel(FusedMatrix a, FusedVector v, const int32_t* __restrict__ part_index)
{
  using Op = ElementOp<3, 1, 1, 1, 1, MPCX_FORM_SOURCE, FN>;
  const int NT = blockDim.x;
  extern __shared__ __align__(16) unsigned char smem[];
  const int nb = v.num_blocks;
  const int per = (nb + 7) >> 3;
  const int bb = (blockIdx.x & 7) * per + (blockIdx.x >> 3); // contiguous runs of row blocks per XCD
  const int tid = threadIdx.x;
  const bool live = bb < nb;
  const int r0 = live ? v.block_row0[bb] : 0, r1 = live ? v.block_row0[bb + 1] : 0;
  const int j = live ? part_index[bb] : -1; // this row block in the matrix launch
  const int nrow = r1 - r0;
  const int64_t nnz0 = live ? a.rowptr[r0] : 0;
  const int nnzb = (live && j >= 0) ? int(a.rowptr[r1] - nnz0) : 0;
  double* s_vals = reinterpret_cast<double*>(smem);
  int32_t* s_rowlo = reinterpret_cast<int32_t*>(s_vals + a.max_nnz);
  double* s_b = reinterpret_cast<double*>(s_rowlo + ((a.max_rows + 1) & ~1));
  const int64_t h0 = live ? v.own_hoff[bb] : 0, h1 = live ? v.own_hoff[bb + 1] : 0;
  const int nown = nrow, nhalo = int(h1 - h0);
  for (int i = tid; i < nnzb; i += NT)
    s_vals[i] = 0.0;
  if (j >= 0)
    for (int rl = tid; rl < nrow; rl += NT)
      s_rowlo[rl] = int(a.rowptr[r0 + rl] - nnz0);
  for (int i = tid; i < nown + nhalo; i += NT)
    s_b[i] = 0.0;
  fastmath_init_lds(); // ends in a barrier
  if (!live)
    return;
  const double c0 = a.constants ? a.constants[0] : 1.0;
  const uint4* __restrict__ recs = static_cast<const uint4*>(a.cube_recs);
  auto matrix_half = [&]()
  {
    if (j < 0)
      return;
    const int64_t e0 = a.block_ent_off[j], e1 = a.block_ent_off[j + 1];
    for (int64_t t = e0 + tid; t < e1; t += NT)
    {
      uint4 cur[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        cur[i] = recs[t * 4 + i];
      const int32_t vv[8] = {int32_t(cur[0].x), int32_t(cur[0].y), int32_t(cur[0].z), int32_t(cur[0].w),
                             int32_t(cur[1].x), int32_t(cur[1].y), int32_t(cur[1].z), int32_t(cur[1].w)};
      double j0[3], j1[3], j2[3];
      {
        const int64_t n0 = vv[0] & DOF_MASK, n1 = vv[1] & DOF_MASK, n2 = vv[2] & DOF_MASK, n4 = vv[4] & DOF_MASK;
#pragma unroll
        for (int r = 0; r < 3; ++r)
        {
          const double x0 = a.x[3 * n0 + r];
          j0[r] = a.x[3 * n1 + r] - x0;
          j1[r] = a.x[3 * n2 + r] - x0;
          j2[r] = a.x[3 * n4 + r] - x0;
        }
      }
      const uint32_t ow[8] = {cur[2].x, cur[2].y, cur[2].z, cur[2].w, cur[3].x, cur[3].y, cur[3].z, cur[3].w};
      double C0[3], C1[3], C2[3];
      cross3(j1, j2, C0);
      cross3(j2, j0, C1);
      cross3(j0, j1, C2);
      const double det = j0[0] * C0[0] + j0[1] * C0[1] + j0[2] * C0[2];
      const double s = c0 / fabs(det);
      const double M[6] = {s * (C0[0] * C0[0] + C0[1] * C0[1] + C0[2] * C0[2]), s * (C0[0] * C1[0] + C0[1] * C1[1] + C0[2] * C1[2]),
                           s * (C0[0] * C2[0] + C0[1] * C2[1] + C0[2] * C2[2]), s * (C1[0] * C1[0] + C1[1] * C1[1] + C1[2] * C1[2]),
                           s * (C1[0] * C2[0] + C1[1] * C2[1] + C1[2] * C2[2]), s * (C2[0] * C2[0] + C2[1] * C2[1] + C2[2] * C2[2])};
      int base[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
      {
        const int r = vv[i] & DOF_MASK;
        const bool mine = r >= r0 && r < r1 && !(vv[i] >> MASK_SHIFT);
        base[i] = mine ? s_rowlo[mine ? r - r0 : 0] : -1;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jj = i; jj < 8; ++jj)
        {
          if (!fan_coupled(i, jj))
            continue;
          double val = 0.0;
#pragma unroll
          for (int m = 0; m < 6; ++m)
          {
            const double k = FAN_AFFINE.k[m][i][jj];
            if (k != 0.0)
              val = fma(k, M[m], val);
          }
          if (base[i] >= 0 && !(vv[jj] >> MASK_SHIFT))
          {
            const int p = fan_pair_index(i, jj);
            const int off = int((ow[p >> 3] >> (4 * (p & 7))) & 0xf);
            __hip_atomic_fetch_add(s_vals + base[i] + off, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          if (i != jj && base[jj] >= 0 && !(vv[i] >> MASK_SHIFT))
          {
            const int p = fan_pair_index(jj, i);
            const int off = int((ow[p >> 3] >> (4 * (p & 7))) & 0xf);
            __hip_atomic_fetch_add(s_vals + base[jj] + off, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
    }
  };
  auto vector_half = [&]()
  {
    const int64_t e0 = v.block_ent_off[bb], e1 = v.block_ent_off[bb + 1];
    const int32_t* __restrict__ ents = v.block_ents;
    for (int64_t t = e0 + tid; t < e1; t += NT)
    {
      const int64_t c = ents[t];
      int32_t vv[8];
      {
        const uint4* p = reinterpret_cast<const uint4*>(v.cube_verts + c * 8);
        const uint4 w0 = p[0], w1 = p[1];
        vv[0] = w0.x, vv[1] = w0.y, vv[2] = w0.z, vv[3] = w0.w, vv[4] = w1.x, vv[5] = w1.y, vv[6] = w1.z, vv[7] = w1.w;
      }
      double X[8][3];
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k)
          X[i][k] = v.x[3 * int64_t(vv[i]) + k];
      double be8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        be8[i] = 0.0;
#pragma unroll
      for (int tet = 0; tet < 6; ++tet)
      {
        double cd[12];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int k = 0; k < 3; ++k)
            cd[3 * i + k] = X[fan_vertex(tet, i)][k];
        double be[4];
        Op::tabulate(be, nullptr, v.constants, cd, 0, v.kernel);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          be8[fan_vertex(tet, i)] += be[i];
      }
      int32_t w[8];
      {
        const uint4* p = reinterpret_cast<const uint4*>(v.own_lmap + c * 8);
        const uint4 w0 = p[0], w1 = p[1];
        w[0] = w0.x, w[1] = w0.y, w[2] = w0.z, w[3] = w0.w, w[4] = w1.x, w[5] = w1.y, w[6] = w1.z, w[7] = w1.w;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (!(w[i] >> MASK_SHIFT))
          __hip_atomic_fetch_add(s_b + (w[i] & DOF_MASK), be8[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };
  matrix_half();
  vector_half();
  __syncthreads();
  if (j >= 0)
  {
    if (a.store_mode)
      for (int i = tid; i < nnzb; i += NT)
        a.vals[nnz0 + i] = s_vals[i];
    else
      for (int i = tid; i < nnzb; i += NT)
        a.vals[nnz0 + i] += s_vals[i];
  }
  for (int i = tid; i < nown; i += NT)
    v.b[r0 + i] += s_b[i];
  for (int i = tid; i < nhalo; i += NT)
    v.own_spill[h0 + i] = s_b[nown + i];
}

static int launch_matrix_cubes_elasticity(const mpcx_matrix_args_t& a)
{
  if (!a.cube_recs || a.plan.num_blocks <= 0 || !a.constants || !a.plan.row_pairs || !a.plan.block_ents)
  {
    mpcx_set_error("mpcx_assemble_matrix: the elasticity cluster algorithm needs one record per cluster (mpcx_cube_records, "
                   "bs = 3, slots = clusters), a row-pair plan over the clusters (pair id = cluster * 8 + local vertex) and "
                   "the constants (mu, lambda)");
    return -3;
  }
  if (!(a.cube_flags & 1))
  {
    mpcx_set_error("mpcx_assemble_matrix: the elasticity cluster kernel is the closed form on parallelepiped clusters: "
                   "cube_flags bit 0 must vouch for the clusters of the launch (mpcx_cell_shapes); other cells go "
                   "through the per-cell algorithms");
    return -6;
  }
  const size_t lds = size_t(a.plan.max_nnz) * 8 + size_t(a.plan.max_rows) * 4;
  if (lds > 160 * 1024)
  {
    mpcx_set_error("mpcx_assemble_matrix: row-block plan exceeds 160 KiB of LDS");
    return -4;
  }
  if (int rc = check(hipFuncSetAttribute(reinterpret_cast<const void*>(matrix_cube_elasticity_rowpair_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)),
                     "hipFuncSetAttribute"))
    return rc;
  // 512 threads: two workgroups per CU by LDS (74 KB blocks) AND by registers (104 VGPRs: 4 waves per SIMD), so one block
  // is written out while the other computes; contact elasticity (config 4): 1024 threads 1.22 ms, 768 1.23, 512 0.86,
  // 256 1.07; half-size blocks with 256 threads 0.93
  const char* e = std::getenv("MPCX_CUBE_EL_THREADS");
  int threads = e ? std::atoi(e) : 512;
  if (threads < 64 || threads > CUBE_EL_THREADS || threads % 64)
    threads = 512;
  const unsigned grid = 8u * unsigned((a.plan.num_blocks + 7) / 8);
  hipLaunchKernelGGL(matrix_cube_elasticity_rowpair_kernel, dim3(grid), dim3(threads), lds, static_cast<hipStream_t>(a.stream), a);
  return check(hipGetLastError(), "elasticity cluster kernel launch");
}

static int launch_matrix_hex(const mpcx_matrix_args_t& a)
{
  const mpcx_kernel_t& k = a.kernel;
  if (k.form != MPCX_FORM_STIFFNESS || k.degree != 1 || k.bs != 1 || k.degree1 != 1 || k.bs1 != 1 || k.coeff_degree != 0
      || a.coeffs || a.estride != 1 || a.nv != 8 || k.nq != 8)
  {
    mpcx_set_error("mpcx_assemble_matrix: on hexahedra the cluster algorithm covers the scalar Q1 stiffness form with the "
                   "2 x 2 x 2 Gauss rule, without coefficients");
    return -10;
  }
  if (a.plan.num_blocks <= 0 || !a.plan.block_row0 || !a.plan.block_ent_off)
  {
    mpcx_set_error("mpcx_assemble_matrix: the cluster algorithm needs records (mpcx_hex_records) and a row-block plan");
    return -3;
  }
  if (a.cube_rec_bytes != 0 && a.cube_rec_bytes != 96)
  {
    mpcx_set_error("mpcx_assemble_matrix: hexahedra take 96-byte records");
    return -6;
  }
  const size_t lds = size_t(a.plan.max_nnz) * 8 + size_t(a.plan.max_rows) * 4;
  if (lds > 160 * 1024)
  {
    mpcx_set_error("mpcx_assemble_matrix: row-block plan exceeds 160 KiB of LDS");
    return -4;
  }
  static const bool no_affine = std::getenv("MPCX_HEX_NO_AFFINE") != nullptr;
  const bool only_affine = (a.cube_flags & 1) != 0 && !no_affine; // the caller vouches for the cells of this launch
  const void* kern = no_affine ? reinterpret_cast<const void*>(matrix_hex_kernel<0>)
                               : (only_affine ? reinterpret_cast<const void*>(matrix_hex_kernel<2>)
                                              : reinterpret_cast<const void*>(matrix_hex_kernel<1>));
  if (int rc = check(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)), "hipFuncSetAttribute"))
    return rc;
  // threads per workgroup: the closed-form instance is light (104 registers): 512 threads put every slot of a 256-row
  // block (~400) in flight at once, 1.38 ms against 1.66 with 256 threads; the instance with the quadrature path (252
  // registers) runs two waves per SIMD either way: 256 threads 1.75 ms, 512 2.28 ms
  const int dflt = only_affine ? 512 : 256;
  const char* e = std::getenv("MPCX_HEX_THREADS");
  int threads = e ? std::atoi(e) : dflt;
  if (threads < 64 || threads > HEX_MAX_THREADS || threads % 64)
    threads = dflt;
  const unsigned grid = 8u * unsigned((a.plan.num_blocks + 7) / 8);
  if (no_affine)
    hipLaunchKernelGGL(matrix_hex_kernel<0>, dim3(grid), dim3(threads), lds, static_cast<hipStream_t>(a.stream), a);
  else if (only_affine)
    hipLaunchKernelGGL(matrix_hex_kernel<2>, dim3(grid), dim3(threads), lds, static_cast<hipStream_t>(a.stream), a);
  else
    hipLaunchKernelGGL(matrix_hex_kernel<1>, dim3(grid), dim3(threads), lds, static_cast<hipStream_t>(a.stream), a);
  return check(hipGetLastError(), "hexahedron matrix kernel launch");
}

static int launch_matrix_p2_cubes(const mpcx_matrix_args_t& a)
{
  const mpcx_kernel_t& k = a.kernel;
  if (k.coeff_degree != 0 || a.coeffs || a.estride != 1 || a.nv != 4 || !(a.cube_flags & 1) || a.cube_rec_bytes != P2CUBE_REC
      || !a.cube_recs || a.plan.num_blocks <= 0 || !a.plan.row_pairs || !a.plan.block_ents)
  {
    mpcx_set_error("mpcx_assemble_matrix: the P2 cluster kernel covers the scalar P2 stiffness form without coefficients on "
                   "parallelepiped clusters (cube_flags bit 0), with one record per cluster (mpcx_p2_cluster_records, "
                   "cube_rec_bytes = 640) and a row-pair plan over the clusters (pair id = cluster * 27 + local dof)");
    return -10;
  }
  const size_t lds = size_t(a.plan.max_nnz) * 8 + size_t(a.plan.max_rows) * 4;
  if (lds > 160 * 1024)
  {
    mpcx_set_error("mpcx_assemble_matrix: row-block plan exceeds 160 KiB of LDS");
    return -4;
  }
  if (int rc = check(hipFuncSetAttribute(reinterpret_cast<const void*>(matrix_p2_cube_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)),
                     "hipFuncSetAttribute"))
    return rc;
  const char* e = std::getenv("MPCX_P2CUBE_THREADS");
  int threads = e ? std::atoi(e) : 1024; // (246^3: 256 threads 40.2 ms, 512 30.2, 1024 22.2 -- the per-cell kernel: 14.1)
  if (threads < 64 || threads > P2CUBE_MAX_THREADS || threads % 64)
    threads = 1024;
  const unsigned grid = 8u * unsigned((a.plan.num_blocks + 7) / 8);
  hipLaunchKernelGGL(matrix_p2_cube_kernel, dim3(grid), dim3(threads), lds, static_cast<hipStream_t>(a.stream), a);
  return check(hipGetLastError(), "P2 cluster kernel launch");
}

int launch_matrix_cubes(const mpcx_matrix_args_t& a)
{
  const mpcx_kernel_t& k = a.kernel;
  if (k.celltype == MPCX_CELL_HEXAHEDRON)
    return launch_matrix_hex(a);
  if (k.form == MPCX_FORM_STIFFNESS && k.celltype == MPCX_CELL_TETRAHEDRON && k.degree == 2 && k.bs == 1 && k.degree1 == 2
      && k.bs1 == 1)
    return launch_matrix_p2_cubes(a);
  if (k.form == MPCX_FORM_ELASTICITY && k.celltype == MPCX_CELL_TETRAHEDRON && k.degree == 1 && k.bs == 3 && k.degree1 == 1
      && k.bs1 == 3 && !a.coeffs && a.estride == 1 && a.nv == 4)
    return launch_matrix_cubes_elasticity(a);
  if (k.form != MPCX_FORM_STIFFNESS || k.celltype != MPCX_CELL_TETRAHEDRON || k.degree != 1 || k.bs != 1
      || k.degree1 != 1 || k.bs1 != 1 || k.coeff_degree != 0 || a.coeffs || a.estride != 1 || a.nv != 4)
  {
    mpcx_set_error("mpcx_assemble_matrix: the cluster algorithm covers the scalar P1 stiffness form and P1 vector "
                   "elasticity on tetrahedra, without coefficients");
    return -10;
  }
  // (cube_recs may be NULL when none of this launch's row blocks has a slot -- ghost-row blocks of a slab mesh: the
  // blocks are still written, as zeros)
  if (a.plan.num_blocks <= 0 || !a.plan.block_row0 || !a.plan.block_ent_off)
  {
    mpcx_set_error("mpcx_assemble_matrix: the cluster algorithm needs records (mpcx_cube_records) and a row-block plan");
    return -3;
  }
  size_t lds = size_t(a.plan.max_nnz) * 8 + size_t(a.plan.max_rows) * 4;
  if (lds > 160 * 1024)
  {
    mpcx_set_error("mpcx_assemble_matrix: row-block plan exceeds 160 KiB of LDS");
    return -4;
  }
  // occupancy shaping (experiments / co-scheduling with the vector kernel on a second stream): a workgroup asks for at
  // least this much LDS, e.g. 90000 -> one workgroup per CU
  static const size_t lds_floor = []
  {
    const char* e = std::getenv("MPCX_CUBE_LDS_FLOOR");
    return e ? size_t(std::atol(e)) : size_t(0);
  }();
  if (lds_floor > lds && lds_floor <= 160 * 1024)
    lds = lds_floor;
  if (a.lds_floor > 0 && size_t(a.lds_floor) > lds && a.lds_floor <= 160 * 1024) // per-launch cap (include/mpcx.h)
    lds = size_t(a.lds_floor);
  const bool narrow = a.cube_rec_bytes == 64;
  if (a.cube_rec_bytes != 0 && a.cube_rec_bytes != 64 && a.cube_rec_bytes != 96)
  {
    mpcx_set_error("mpcx_assemble_matrix: cube_rec_bytes must be 96 (or 0) or 64");
    return -6;
  }
  const bool affine = (a.cube_flags & 1) != 0; // the caller vouches: every cluster of this launch is a parallelepiped
  const void* kern = affine ? (narrow ? reinterpret_cast<const void*>(matrix_cube_affine_kernel<true>)
                                      : reinterpret_cast<const void*>(matrix_cube_affine_kernel<false>))
                            : (narrow ? reinterpret_cast<const void*>(matrix_cube_kernel<true>)
                                      : reinterpret_cast<const void*>(matrix_cube_kernel<false>));
  if (int rc = check(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)), "hipFuncSetAttribute"))
    return rc;
  const int dflt = affine ? 512 : 256;
  const char* e = std::getenv(affine ? "MPCX_CUBE_AFFINE_THREADS" : "MPCX_CUBE_THREADS");
  int threads = e ? std::atoi(e) : dflt;
  if (threads < 64 || threads > (affine ? CUBE_AFFINE_MAX_THREADS : CUBE_MAX_THREADS) || threads % 64)
    threads = dflt;
  const unsigned grid = 8u * unsigned((a.plan.num_blocks + 7) / 8);
  if (affine && narrow)
    hipLaunchKernelGGL(matrix_cube_affine_kernel<true>, dim3(grid), dim3(threads), lds, static_cast<hipStream_t>(a.stream), a);
  else if (affine)
    hipLaunchKernelGGL(matrix_cube_affine_kernel<false>, dim3(grid), dim3(threads), lds, static_cast<hipStream_t>(a.stream), a);
  else if (narrow)
    hipLaunchKernelGGL(matrix_cube_kernel<true>, dim3(grid), dim3(threads), lds, static_cast<hipStream_t>(a.stream), a);
  else
    hipLaunchKernelGGL(matrix_cube_kernel<false>, dim3(grid), dim3(threads), lds, static_cast<hipStream_t>(a.stream), a);
  return check(hipGetLastError(), "matrix cluster kernel launch");
}

static int launch_vector_hex(const mpcx_vector_args_t& a)
{
  const mpcx_kernel_t& k = a.kernel;
  if (k.form != MPCX_FORM_SOURCE || k.degree != 1 || k.bs != 1 || k.coeff_degree != 0 || a.coeffs || a.nv != 8 || !a.cube_verts
      || (k.nq != 1 && k.nq != 8 && k.nq != 27) || (k.fn_id != 0 && k.fn_id != 1))
  {
    mpcx_set_error("mpcx_assemble_vector: on hexahedra the cluster algorithm covers the scalar Q1 source form (fn_id 0 or 1, "
                   "tensor Gauss rule of 1, 8 or 27 points) without coefficients");
    return -10;
  }
  if (a.n_cubes == 0)
    return 0;
  if (!a.own_lmap || a.plan.num_blocks <= 0 || !a.own_hoff || !a.own_spill || !a.own_seg
      || (a.n_own_rows > 0 && (!a.own_rows || !a.own_src)))
  {
    mpcx_set_error("mpcx_assemble_vector: hexahedra need the owner-computes plan (own_lmap ...)");
    return -5;
  }
  const size_t lds = size_t(a.plan.max_rows) * 8;
  if (lds > 96 * 1024)
  {
    mpcx_set_error("mpcx_assemble_vector: cluster owner plan exceeds the LDS budget");
    return -4;
  }
  const unsigned g = 8u * unsigned((a.plan.num_blocks + 7) / 8);
  hipStream_t st = static_cast<hipStream_t>(a.stream);
  auto go = [&](auto kernel) -> int
  {
    if (int rc = check(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           int(lds)),
                       "hipFuncSetAttribute"))
      return rc;
    hipLaunchKernelGGL(kernel, dim3(g), dim3(VCUBE_OWN_THREADS), lds, st, a);
    return check(hipGetLastError(), "hexahedron vector kernel launch");
  };
  int rc = sync_box_switch();
  if (rc)
    return rc;
  if (k.fn_id == 1)
    rc = k.nq == 27 ? go(vector_hex_own_kernel<1, 3>) : (k.nq == 8 ? go(vector_hex_own_kernel<1, 2>) : go(vector_hex_own_kernel<1, 1>));
  else
    rc = k.nq == 27 ? go(vector_hex_own_kernel<-1, 3>) : (k.nq == 8 ? go(vector_hex_own_kernel<-1, 2>) : go(vector_hex_own_kernel<-1, 1>));
  if (rc)
    return rc;
  if (a.n_own_rows > 0)
    return launch_vector_spill_reduce(a, 1);
  // (rows of slave dofs are skipped here; the caller moves them to their masters with the per-cell kernel over the
  // slave cells -- on hexahedra an imported kernel, mpcx_assemble_vector with n_entities = 0)
  return 0;
}

int launch_vector_cubes(const mpcx_vector_args_t& a)
{
  const mpcx_kernel_t& k = a.kernel;
  if (k.celltype == MPCX_CELL_HEXAHEDRON)
    return launch_vector_hex(a);
  if (k.form != MPCX_FORM_SOURCE || k.celltype != MPCX_CELL_TETRAHEDRON || k.degree != 1 || k.bs != 1
      || k.coeff_degree != 0 || a.coeffs || a.nv != 4 || !a.cube_verts)
  {
    mpcx_set_error("mpcx_assemble_vector: the cluster algorithm covers the scalar P1 source form on tetrahedra "
                   "without coefficients");
    return -10;
  }
  if (a.n_cubes == 0)
    return 0;
  hipStream_t st = static_cast<hipStream_t>(a.stream);
  if (a.own_lmap)
  {
    if (a.plan.num_blocks <= 0 || !a.own_hoff || !a.own_spill || !a.own_seg || (a.n_own_rows > 0 && (!a.own_rows || !a.own_src)))
    {
      mpcx_set_error("mpcx_assemble_vector: incomplete owner-computes plan for the cluster algorithm");
      return -5;
    }
    size_t lds = size_t(a.plan.max_rows) * 8;
    if (lds > 96 * 1024)
    {
      mpcx_set_error("mpcx_assemble_vector: cluster owner plan exceeds the LDS budget");
      return -4;
    }
    static const size_t lds_floor = []
    {
      const char* e = std::getenv("MPCX_VCUBE_LDS_FLOOR");
      return e ? size_t(std::atol(e)) : size_t(0);
    }();
    if (lds_floor > lds && lds_floor <= 160 * 1024)
      lds = lds_floor;
    if (a.lds_floor > 0 && size_t(a.lds_floor) > lds && a.lds_floor <= 160 * 1024)
      lds = size_t(a.lds_floor);
    const unsigned g = 8u * unsigned((a.plan.num_blocks + 7) / 8);
    bool grid_staged = false;
    if (a.grid_idx != nullptr)
    {
      grid_staged = a.grid_block_rows != nullptr;
      if (grid_staged)
      {
        // the block's rows of the table (vector_cube_grid_body): as many as the longest list needs
        const int rows = (a.grid_block_rows_max > 0 && a.grid_block_rows_max <= MPCX_GRID_BLOCK_ROWS) ? a.grid_block_rows_max
                                                                                                        : MPCX_GRID_BLOCK_ROWS;
        lds = ((lds + 15) & ~size_t(15)) + size_t(rows) * GRID_LROW * 8;
      }
    }
    auto go = [&](auto kernel) -> int
    {
      if (int rc = check(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             int(lds)),
                         "hipFuncSetAttribute"))
        return rc;
      hipLaunchKernelGGL(kernel, dim3(g), dim3(vcube_own_threads()), lds, st, a);
      return check(hipGetLastError(), "vector cluster owner kernel launch");
    };
    if (int rc = sync_box_switch())
      return rc;
    if (a.grid_idx != nullptr)
    {
      if (k.fn_id != 1 || !a.grid_iv || !a.grid_tab || a.grid_n[0] <= 0 || a.grid_n[1] <= 0 || a.grid_n[2] <= 0)
      {
        mpcx_set_error("mpcx_assemble_vector: grid_idx needs kernel.fn_id = 1, grid_iv, grid_tab and three interval counts");
        return -8;
      }
      const int rows = a.grid_n[0] + a.grid_n[1] + a.grid_n[2];
      hipLaunchKernelGGL(box_grid_tables_kernel, dim3(unsigned((int64_t(rows) * 20 + 255) / 256)), dim3(256), 0, st, a.grid_n[0],
                         a.grid_n[1], a.grid_n[2], a.grid_iv, a.grid_tab, a.constants, k);
      if (int rc = grid_staged ? go(vector_cube_grid_kernel<true>) : go(vector_cube_grid_kernel<false>))
        return rc;
    }
    else if (int rc = k.fn_id == 1 ? (a.cube_boxes ? go(vector_cube_own_kernel<1, true>) : go(vector_cube_own_kernel<1, false>))
                                   : go(vector_cube_own_kernel<-1>))
      return rc;
    if (a.n_own_rows > 0)
      if (int rc = launch_vector_spill_reduce(a, 1))
        return rc;
    return launch_vector_slave_rows(a); // rows of slave dofs: to their masters (cpp/assemble_vector.h:35-69)
  }
  const dim3 grid(grid_for(a.n_cubes, VCUBE_THREADS));
  if (k.fn_id == 1)
    hipLaunchKernelGGL(vector_cube_kernel<1>, grid, dim3(VCUBE_THREADS), 0, st, a);
  else
    hipLaunchKernelGGL(vector_cube_kernel<-1>, grid, dim3(VCUBE_THREADS), 0, st, a);
  return check(hipGetLastError(), "vector cluster kernel launch");
}

} // namespace mpcx

extern "C" int mpcx_cube_detect(const int32_t* cells, int64_t n_groups, int32_t* verts, int8_t* ok, void* stream)
{
  if (n_groups == 0)
    return 0;
  hipLaunchKernelGGL(mpcx::cube_detect_kernel, dim3(mpcx::grid_for(n_groups, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), cells, n_groups, verts, ok);
  return mpcx::check(hipGetLastError(), "cube_detect launch");
}

extern "C" int mpcx_hex_slot_shapes(int64_t n_slots, const void* recs, const double* x, uint8_t* general, void* stream)
{
  if (n_slots == 0)
    return 0;
  hipLaunchKernelGGL(mpcx::hex_slot_shapes_kernel, dim3(mpcx::grid_for(n_slots, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), n_slots, static_cast<const int32_t*>(recs), 24, x, general);
  return mpcx::check(hipGetLastError(), "hex_slot_shapes launch");
}

extern "C" int mpcx_cell_shapes(int64_t n, const int32_t* verts, const double* x, uint8_t* general, void* stream)
{
  if (n == 0)
    return 0;
  hipLaunchKernelGGL(mpcx::hex_slot_shapes_kernel, dim3(mpcx::grid_for(n, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), n, verts, 8, x, general);
  return mpcx::check(hipGetLastError(), "cell_shapes launch");
}

extern "C" int mpcx_cube_slot_width(int64_t n_slots, const void* recs, uint8_t* wide, void* stream)
{
  if (n_slots == 0)
    return 0;
  hipLaunchKernelGGL(mpcx::cube_slot_width_kernel, dim3(mpcx::grid_for(n_slots, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), n_slots, static_cast<const mpcx::CubeRec*>(recs), wide);
  return mpcx::check(hipGetLastError(), "cube_slot_width launch");
}

extern "C" int mpcx_cube_pack_narrow(int64_t n_out, const int64_t* src, const void* recs, void* out, void* stream)
{
  if (n_out == 0)
    return 0;
  hipLaunchKernelGGL(mpcx::cube_pack_narrow_kernel, dim3(mpcx::grid_for(n_out, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), n_out, src, static_cast<const mpcx::CubeRec*>(recs),
                     static_cast<mpcx::CubeRecNarrow*>(out));
  return mpcx::check(hipGetLastError(), "cube_pack_narrow launch");
}

extern "C" int mpcx_cluster_keys(const double* x, const int32_t* cells, int64_t n_cells, int64_t* keys, void* stream)
{
  if (n_cells == 0)
    return 0;
  hipLaunchKernelGGL(mpcx::tet_long_edge_kernel, dim3(mpcx::grid_for(n_cells, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, cells, n_cells, keys);
  return mpcx::check(hipGetLastError(), "cluster_keys launch");
}

extern "C" int mpcx_cluster_build(int64_t n, const int64_t* sorted_keys, const int32_t* order, const int32_t* cells,
                                  int32_t* verts, int8_t* ok, int8_t* cell_in_fan, void* stream)
{
  if (n == 0)
    return 0;
  hipLaunchKernelGGL(mpcx::fan_build_kernel, dim3(mpcx::grid_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n,
                     sorted_keys, order, cells, verts, ok, cell_in_fan);
  return mpcx::check(hipGetLastError(), "cluster_build launch");
}

extern "C" int mpcx_rowblock_pairs_device(int64_t n_entities, int32_t estride, const int32_t* entities0,
                                          const int32_t* dofmap0, int32_t nd0, int32_t bs0, int32_t num_blocks,
                                          const int32_t* block_row0, int32_t* counts, const int64_t* offsets,
                                          int32_t* pair_block, int32_t* pair_ent, int32_t* pair_rows, int32_t rotate,
                                          void* stream)
{
  if (n_entities == 0)
    return 0;
  if (nd0 > 32)
  {
    mpcx_set_error("mpcx_rowblock_pairs_device: more than 32 dofs per entity");
    return -7;
  }
  const dim3 grid(mpcx::grid_for(n_entities, 256));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!offsets)
    hipLaunchKernelGGL(mpcx::rowblock_pairs_kernel<false>, grid, dim3(256), 0, st, n_entities, estride, entities0, dofmap0,
                       nd0, bs0, num_blocks, block_row0, counts, offsets, pair_block, pair_ent, pair_rows, rotate);
  else
    hipLaunchKernelGGL(mpcx::rowblock_pairs_kernel<true>, grid, dim3(256), 0, st, n_entities, estride, entities0, dofmap0,
                       nd0, bs0, num_blocks, block_row0, counts, offsets, pair_block, pair_ent, pair_rows, rotate);
  return mpcx::check(hipGetLastError(), "rowblock_pairs launch");
}

extern "C" int mpcx_cube_records(int64_t n_slots, const int32_t* block_ents, const int32_t* cube_verts, int32_t bs,
                                 const int8_t* bc, const int8_t* is_slave, const mpcx_nnz_t* rowptr,
                                 const int32_t* cols, void* recs, int32_t* overflow, void* stream)
{
  if (n_slots == 0)
    return 0;
  if (bs < 1 || bs > 3)
  {
    mpcx_set_error("mpcx_cube_records: block size must be 1..3");
    return -6;
  }
  hipLaunchKernelGGL(mpcx::cube_records_kernel<false>, dim3(mpcx::grid_for(n_slots * 8, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), n_slots, block_ents, cube_verts, bs, bc, is_slave, rowptr, cols,
                     static_cast<mpcx::CubeRec*>(recs), overflow);
  return mpcx::check(hipGetLastError(), "cube_records launch");
}

extern "C" int mpcx_hex_records(int64_t n_slots, const int32_t* block_ents, const int32_t* cell_verts, int32_t bs,
                                const int8_t* bc, const int8_t* is_slave, const mpcx_nnz_t* rowptr, const int32_t* cols,
                                void* recs, int32_t* overflow, void* stream)
{
  if (n_slots == 0)
    return 0;
  if (bs != 1)
  {
    mpcx_set_error("mpcx_hex_records: scalar spaces only");
    return -6;
  }
  hipLaunchKernelGGL(mpcx::cube_records_kernel<true>, dim3(mpcx::grid_for(n_slots * 8, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), n_slots, block_ents, cell_verts, bs, bc, is_slave, rowptr, cols,
                     static_cast<mpcx::CubeRec*>(recs), overflow);
  return mpcx::check(hipGetLastError(), "hex_records launch");
}

// the constant tables of the P1 closed-form kernels (tests): k6 [6][8][8] = K_m(i, j) of matrix_cube_affine_kernel,
// k9 [9][8][8] = Kp_de(i, j) of matrix_cube_elasticity_rowpair_kernel, hex6 [6][8][8] = the coefficients of M_de in the
// Q1 stiffness entry (i, j) of a parallelepiped (matrix_hex_kernel; off-diagonal d < e: both orders together)
extern "C" int mpcx_p1_cluster_tables(double* k6, double* k9, double* hex6)
{
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j)
    {
      for (int m = 0; m < 6; ++m)
        k6[(m * 8 + i) * 8 + j] = mpcx::FAN_AFFINE.k[m][i][j];
      for (int m = 0; m < 9; ++m)
        k9[(m * 8 + i) * 8 + j] = mpcx::FAN_AFFINE9.k[m][i][j];
      for (int d = 0; d < 3; ++d)
        for (int e = d; e < 3; ++e)
          hex6[(mpcx::sym6(d, e) * 8 + i) * 8 + j] = mpcx::hex_affine_coef(d, e, i, j);
    }
  return 0;
}

// the constant tables of the P2 cluster kernel (tests: compared with a numpy restatement and with the oracle's element tensors)
extern "C" int mpcx_p2_cluster_tables(double* k, int32_t* coupled, int32_t* edge_vertices, int32_t* row_start)
{
  for (int m = 0; m < 6; ++m)
    for (int I = 0; I < 27; ++I)
      for (int J = 0; J < 27; ++J)
        k[(m * 27 + I) * 27 + J] = mpcx::P2FAN.k[m][I][J];
  for (int I = 0; I < 27; ++I)
    for (int J = 0; J < 27; ++J)
      coupled[I * 27 + J] = mpcx::P2FAN.coupled[I][J] ? 1 : 0;
  for (int e = 0; e < 19; ++e)
  {
    edge_vertices[2 * e] = mpcx::P2FAN.ea[e];
    edge_vertices[2 * e + 1] = mpcx::P2FAN.eb[e];
  }
  for (int I = 0; I <= 27; ++I)
    row_start[I] = mpcx::P2FAN.row_start[I];
  return 0;
}

extern "C" int mpcx_p2_cluster_dofs(int64_t n, const int32_t* verts, const int32_t* fan_cells, const int32_t* x_dofmap,
                                    const int32_t* dofmap, int32_t* dofs27, int32_t* bad, void* stream)
{
  if (n == 0)
    return 0;
  hipLaunchKernelGGL(mpcx::p2_cluster_dofs_kernel, dim3(mpcx::grid_for(n, 128)), dim3(128), 0, static_cast<hipStream_t>(stream), n,
                     verts, fan_cells, x_dofmap, dofmap, dofs27, bad);
  return mpcx::check(hipGetLastError(), "p2_cluster_dofs launch");
}

extern "C" int mpcx_p2_cluster_records(int64_t n, const int32_t* verts, const int32_t* dofs27, const double* x, const int8_t* bc,
                                       const int8_t* is_slave, const mpcx_nnz_t* rowptr, const int32_t* cols, void* recs,
                                       int32_t* overflow, void* stream)
{
  if (n == 0)
    return 0;
  hipLaunchKernelGGL(mpcx::p2_cluster_records_kernel, dim3(mpcx::grid_for(n * 27, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), n, verts, dofs27, x, bc, is_slave, rowptr, cols,
                     static_cast<unsigned char*>(recs), overflow);
  return mpcx::check(hipGetLastError(), "p2_cluster_records launch");
}

extern "C" int mpcx_cluster_canonical(int64_t n, int32_t* verts, const int8_t* ok, const double* x, void* stream)
{
  if (n == 0)
    return 0;
  hipLaunchKernelGGL(mpcx::fan_canonical_kernel, dim3(mpcx::grid_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n,
                     verts, ok, x);
  return mpcx::check(hipGetLastError(), "cluster_canonical launch");
}

extern "C" int mpcx_cluster_ordered(int64_t n, int32_t* verts, int32_t* fan_cells, const int32_t* x_dofmap, int8_t* ok, void* stream)
{
  if (n == 0)
    return 0;
  hipLaunchKernelGGL(mpcx::fan_ordered_kernel, dim3(mpcx::grid_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n,
                     verts, fan_cells, x_dofmap, ok);
  return mpcx::check(hipGetLastError(), "cluster_ordered launch");
}

// (mpcx_preload, csrc/mpcx_kernels.hip: the first launch from a translation unit loads its code object)
namespace
{
__global__ void preload_cubes_kernel() {}
} // namespace
extern "C" int mpcx_preload_cubes(void* stream)
{
  hipLaunchKernelGGL(preload_cubes_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream));
  return hipGetLastError() == hipSuccess ? 0 : -100;
}

extern "C" int mpcx_assemble_fused(const mpcx_matrix_args_t* pa, const mpcx_vector_args_t* pv, const int32_t* part_index)
{
  using namespace mpcx;
  const mpcx_matrix_args_t& a = *pa;
  const mpcx_vector_args_t& v = *pv;
  const mpcx_kernel_t& km = a.kernel;
  const mpcx_kernel_t& kv = v.kernel;
  if (km.form != MPCX_FORM_STIFFNESS || km.celltype != MPCX_CELL_TETRAHEDRON || km.degree != 1 || km.bs != 1 || km.degree1 != 1
      || km.bs1 != 1 || km.coeff_degree != 0 || a.coeffs || a.estride != 1 || a.nv != 4 || a.cube_rec_bytes != 64
      || !(a.cube_flags & 1) || a.cube_rec_index || a.algorithm != MPCX_ALG_CUBE || kv.form != MPCX_FORM_SOURCE
      || kv.celltype != MPCX_CELL_TETRAHEDRON || kv.degree != 1 || kv.bs != 1 || kv.coeff_degree != 0 || v.coeffs || v.nv != 4
      || !v.cube_verts || !v.own_lmap || v.algorithm != MPCX_ALG_CUBE || !part_index || a.x != v.x || a.val_map || v.row_map)
  {
    mpcx_set_error("mpcx_assemble_fused: scalar P1 stiffness (narrow records, parallelepiped clusters) + scalar P1 source with the "
                   "owner-computes cluster plan on the same mesh");
    return -10;
  }
  if (v.plan.num_blocks <= 0 || !v.plan.block_row0 || !v.plan.block_ent_off || !v.plan.block_ents || !v.own_hoff || !v.own_spill
      || !v.own_seg || (v.n_own_rows > 0 && (!v.own_rows || !v.own_src)) || !a.plan.block_ent_off || !a.cube_recs)
  {
    mpcx_set_error("mpcx_assemble_fused: incomplete plans");
    return -5;
  }
  const size_t lds = size_t(a.plan.max_nnz) * 8 + size_t((a.plan.max_rows + 1) & ~1) * 4 + size_t(v.plan.max_rows) * 8;
  if (lds > 160 * 1024)
  {
    mpcx_set_error("mpcx_assemble_fused: the two row-block copies exceed 160 KiB of LDS");
    return -4;
  }
  hipStream_t st = static_cast<hipStream_t>(v.stream);
  const unsigned grid = 8u * unsigned((v.plan.num_blocks + 7) / 8);
  auto go = [&](auto kernel) -> int
  {
    if (int rc = check(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)),
                       "hipFuncSetAttribute"))
      return rc;
    const FusedMatrix fa{a.rowptr, a.vals, a.x, a.constants, a.cube_recs, a.plan.block_ent_off, a.plan.max_nnz, a.plan.max_rows,
                         a.store_mode};
    const FusedVector fv{v.b,          v.x,           v.constants, v.kernel,  v.plan.block_row0, v.plan.block_ent_off, v.plan.block_ents,
                         v.cube_verts, v.own_lmap,    v.own_hoff,  v.own_spill, v.plan.num_blocks};
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(FUSED_THREADS), lds, st, fa, fv, part_index);
    return check(hipGetLastError(), "fused cluster kernel launch");
  };
  if (int rc = kv.fn_id == 1 ? go(fused_cube_kernel<1>) : go(fused_cube_kernel<-1>))
    return rc;
  if (v.n_own_rows > 0)
    if (int rc = launch_vector_spill_reduce(v, 1))
      return rc;
  return launch_vector_slave_rows(v);
}
