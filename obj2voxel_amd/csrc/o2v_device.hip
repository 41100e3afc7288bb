// o2v_device.hip -- the MI355X (gfx950) voxelization pipeline behind include/o2v_hip.h.
//
// Replaces, for one GPU's z-slab of the grid, the reference's chunk loop (src/obj2voxel.cpp:467-520) and
// Voxelizer::voxelize (src/voxelization.cpp:480-526).  Written for CDNA4: 64-wide wavefronts, LDS-staged leaf
// geometry and work queues, register-resident clip stacks, 32-bit atomics on a dense (bricked) HBM grid of per-cell
// counters.  No MFMA: the path is float32 VALU + HBM/atomic traffic.
//
// Stages (all on one HIP stream, no host round trips between them):
//   K0  k_bounds / k_setup     mesh bounds (obj2voxel.cpp:180-200) and mesh transform (obj2voxel.cpp:370-402)
//   K1  k_expand_roots         transform (obj2voxel.cpp:202-224), alignment test (voxelization.cpp:335-347),
//       k_expand_nodes         exact LIFO subdivision (voxelization.cpp:349-379) done breadth-first with an
//                              order key that reproduces the reference's processing order,
//       k_expand_big           tiles of <= 256 candidate voxels for large leaves
//   K2  k_voxelize<UV>         AABB walk + plane cull (voxelization.cpp:426-472) + six-plane clip by triangle
//                              splitting (voxelization.cpp:175-331,383-424); every hit is appended to the hit
//                              pool and counted in its cell with one atomicAdd on the dense grid (-> rank)
//   K5a k_scan_flags/_bricks   reads the dirty bricks of the dense grid, compacts occupied cells, turns the
//                              per-cell counts into offsets (counting sort); k_scatter places the hits
//   K3  k_resolve              per occupied cell: orders the hits like the reference's sequential loops
//                              (sub-voxel, triangle index, leaf order), replays insertWeighted
//                              (voxelization.cpp:56-63,466-468) and moveUvBufferIntoVoxels (:513-526) with
//                              MAX / BLEND, then packs (x, y, z, argb) (obj2voxel.cpp:279-297)
//
// Compile with -ffp-contract=off (see o2v_math.h).
#include "o2v_math.h"

#include "../../include/o2v_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace o2v;

namespace {

// ---- device-side records ----------------------------------------------------------------------------------

constexpr uint32_t kTileSize = 256;       // candidate voxels per work tile
constexpr uint32_t kTilesPerBatch = 128;  // tiles a workgroup stages at once (at most)
constexpr uint32_t kMinTilesPerBatch = 4;  // ... and at least (one tile per wavefront in phase 1)
constexpr uint32_t kBlock = 256;          // threads per workgroup (4 wavefronts)
constexpr uint32_t kMaxRounds = 16;       // subdivision depth limit (order key holds 15 levels)
constexpr uint32_t kHitChunk = 256;       // hit-pool slots a wavefront reserves per global atomic
constexpr uint32_t kInlineTiles = 4;      // leaves with more tiles are expanded by k_expand_big

struct __attribute__((aligned(16))) Leaf {  // 96 B
    float v[9];        // sample-space vertices
    float n[3];        // normalize(normal): plane of the distance cull
    float t[6];        // uv per vertex
    uint32_t tri;      // input triangle index
    uint32_t pathkey;  // order key of this leaf among the leaves of `tri` (0 = unsplit triangle)
    uint32_t bmin_xy;  // clamped AABB min: x | y << 16
    uint32_t bmin_z_dx;  // z | dx << 16
    uint32_t dy_dz;      // dy | dz << 16
    float area;          // area of the whole input triangle (voxelization.cpp:416)
};
static_assert(sizeof(Leaf) == 96, "Leaf layout");

struct __attribute__((aligned(16))) Node {  // 80 B: a sub-triangle that still has to be subdivided
    float v[9];
    float t[6];
    uint32_t tri;
    uint32_t pathkey;
    uint32_t depth;
    float area;
    uint32_t pad;
};
static_assert(sizeof(Node) == 80, "Node layout");

struct Tile {
    uint32_t leaf;
    uint32_t start;  // first candidate index inside the leaf's clamped AABB
};

struct BigLeaf {
    uint32_t leaf, first_tile, ntiles, pad;
};

struct __attribute__((aligned(16))) HitRec {  // 32 B: one (leaf, voxel) hit as emitted by k_voxelize
    uint32_t brick;       // brick of the cell; kHoleBrick marks a pool slot that holds no hit
    uint32_t local_rank;  // cell inside the brick << 24 | rank of this hit among the hits of its cell
    uint32_t keyhi;       // sub-voxel << 29 | triangle index
    uint32_t keylo;       // leaf order key
    float w, u, v;        // WeightedUv of this (leaf, voxel) pair (voxelization.cpp:414-423)
    uint32_t pad;
};
constexpr uint32_t kHoleBrick = 0xffffffffu;
constexpr uint32_t kMaxRank = 1u << 24;

struct __attribute__((aligned(8))) SortedRec {  // 24 B: the same hit, placed contiguously with its cell's other hits
    uint32_t keyhi, keylo;
    float w, u, v;
    uint32_t pad;
};

// The sorted array is read through a view: 6 dwords per record in general, 4 (keyhi, keylo, w, pad: one 16-byte access)
// when the mesh has no textured triangle, because then u and v are never used and the scatter's cost scales with the
// bytes it writes.
struct SortedView {
    const uint32_t *base;
    uint32_t stride;  // dwords per record: 6 or 4
    __device__ __forceinline__ SortedRec load(uint32_t i) const
    {
        if (stride == 4u) {
            const uint4 q = reinterpret_cast<const uint4 *>(base)[i];
            return SortedRec{q.x, q.y, __uint_as_float(q.z), 0.f, 0.f, 0u};
        }
        return reinterpret_cast<const SortedRec *>(base)[i];
    }
};

struct __attribute__((aligned(16))) Occ {  // 16 B: one occupied cell
    uint32_t cell_lo, cell_hi;  // brick * 256 + cell in brick
    uint32_t offset;            // first SortedRec of the cell
    uint32_t count;             // number of hits
};

struct DevTexture {
    const uint8_t *pixels;
    uint32_t width, height, channels, wrap;
};

struct Counters {
    uint32_t n_leaves, n_tiles, n_big, n_hits_reserved;
    uint32_t n_vox, batch_cursor, err_flags, n_lane16;
    uint32_t n_mid, n_long, n_huge, scratch_used;
    uint32_t n_dirty, n_sorted, n_bigl, cursor_big;
    uint32_t cursor_mid, cursor_long, cursor_huge, n_lane;
    uint32_t n_nodes[kMaxRounds + 1];
    uint32_t n_w64, pad1[2];
    unsigned long long n_candidates, n_hits;
    uint32_t bounds_enc[6];
    uint32_t pad2[2];
    float xform[12];
};

enum : uint32_t {
    kErrLeafTooLarge = 1u,
    kErrDepth = 2u,
    kErrRank = 4u,
};

struct Params {
    uint64_t n_tris;
    uint32_t S;            // sample resolution = resolution * supersampling
    uint32_t G;            // output resolution
    uint32_t NBx, NBy;     // bricks per grid row / per z layer (brick = 16 x 4 x 4 cells, stored contiguously)
    uint32_t ss_shift;     // 0, or 1 for 2x supersampling
    uint32_t zs0, zs1;     // slab in sample space
    uint32_t zo0;          // slab begin in output space
    uint32_t blend;
    uint32_t cap_leaves, cap_tiles, cap_big, cap_nodes, cap_hits, cap_vox;
    uint32_t n_bricks;     // bricks of this slab
    uint32_t bounds_known;
    float bounds[6];
    int32_t unit[9];
    uint32_t has_uv;
};

// ---- small device helpers ---------------------------------------------------------------------------------

// A pass that ran out of hit-pool or cell-list space has cell offsets that point past the sorted records: the
// resolve kernels skip such a pass (the host grows the buffers and runs it again).
// A pass whose subdivision ran out of leaf / tile / queue space is discarded by the host as well; k_voxelize skips it
// (a tile may name a leaf that was never written).
__device__ __forceinline__ bool expand_overflowed(const Counters *c, const Params &p)
{
    bool over = c->n_leaves > p.cap_leaves || c->n_tiles > p.cap_tiles || c->n_big > p.cap_big;
    for (uint32_t r = 0; r <= kMaxRounds; ++r) over |= c->n_nodes[r] > p.cap_nodes;
    return over;
}
__device__ __forceinline__ bool pass_overflowed(const Counters *c, const Params &p)
{
    return c->n_hits_reserved > p.cap_hits || c->n_sorted > p.cap_hits || c->n_vox > p.cap_vox;
}

__device__ __forceinline__ uint32_t f2ord(float f)
{
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o)
{
    uint32_t b = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(b);
}

// Dense grid layout: bricks of 16 (x) x 4 (y) x 4 (z) cells, each brick 256 consecutive u32 (1 KiB), bricks ordered
// x fastest.  A surface marks ~12 cells' worth of brick volume per unit area in this shape (the same as 8^3 bricks)
// while every brick row is a full 64-byte line; one wavefront reads a brick with a single 16-byte load per lane.
constexpr uint32_t kBrickX = 16, kBrickY = 4, kBrickZ = 4, kBrickCells = 256;

__device__ __forceinline__ uint64_t cell_index(uint32_t ox, uint32_t oy, uint32_t oz_rel, const Params &p, uint32_t &brick)
{
    brick = ((oz_rel >> 2) * p.NBy + (oy >> 2)) * p.NBx + (ox >> 4);
    return (uint64_t) brick * kBrickCells + (((oz_rel & 3u) * 4u + (oy & 3u)) * 16u + (ox & 15u));
}

// exclusive scan of one uint32 per thread over a 256-thread block; returns the block total in `total`
__device__ __forceinline__ uint32_t block_exscan(uint32_t v, uint32_t *s_wave /*[4]*/, uint32_t &total)
{
    uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    __syncthreads();
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < kBlock / 64; ++w) {
        uint32_t c = s_wave[w];
        if (w < wave) base += c;
        tot += c;
    }
    total = tot;
    return base + inc - v;
}

// ---- K0: bounds + transform ---------------------------------------------------------------------------------

__global__ void k_init(Counters *c)
{
    uint32_t i = threadIdx.x;
    uint32_t *w = reinterpret_cast<uint32_t *>(c);
    for (uint32_t k = i; k < sizeof(Counters) / 4; k += blockDim.x) w[k] = 0;
    __syncthreads();
    if (i < 3) c->bounds_enc[i] = f2ord(__builtin_inff());
    else if (i < 6) c->bounds_enc[i] = f2ord(-__builtin_inff());
}

// findMeshBounds (obj2voxel.cpp:180-200): min/max are exact and order-free, so one reduce replaces the batches.
// The vertex array is streamed as float4 triples (12 floats = 4 vertices, so the axis of every element is static);
// one set of six atomics per workgroup.
__global__ __launch_bounds__(kBlock) void k_bounds(const float *__restrict__ verts, uint64_t n_floats, Counters *c)
{
    __shared__ float s_red[6][kBlock / 64];
    const float inf = __builtin_inff();
    float mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
    const uint64_t n_groups = n_floats / 12;
    const float4 *v4 = reinterpret_cast<const float4 *>(verts);
    for (uint64_t g = (uint64_t) blockIdx.x * kBlock + threadIdx.x; g < n_groups; g += (uint64_t) gridDim.x * kBlock) {
        const float4 a = v4[g * 3], b = v4[g * 3 + 1], d = v4[g * 3 + 2];
        const float e[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, d.x, d.y, d.z, d.w};
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            mn[k % 3] = fmin2(e[k], mn[k % 3]);
            mx[k % 3] = fmax2(e[k], mx[k % 3]);
        }
    }
    if (blockIdx.x == 0)
        for (uint64_t i = n_groups * 12 + threadIdx.x; i < n_floats; i += kBlock) {
            const float f = verts[i];
            const int a = (int) (i % 3);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k == a) {
                    mn[k] = fmin2(f, mn[k]);
                    mx[k] = fmax2(f, mx[k]);
                }
        }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], d, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], d, 64));
        }
    }
    if ((threadIdx.x & 63u) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            s_red[a][threadIdx.x >> 6] = mn[a];
            s_red[3 + a][threadIdx.x >> 6] = mx[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float r = s_red[threadIdx.x][0];
        for (uint32_t w = 1; w < kBlock / 64; ++w) r = threadIdx.x < 3 ? fminf(r, s_red[threadIdx.x][w]) : fmaxf(r, s_red[threadIdx.x][w]);
        if (threadIdx.x < 3) atomicMin(&c->bounds_enc[threadIdx.x], f2ord(r));
        else atomicMax(&c->bounds_enc[threadIdx.x], f2ord(r));
    }
}

__global__ void k_setup(Counters *c, Params p)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    V3 mn, mx;
    if (p.bounds_known) {
        mn = {p.bounds[0], p.bounds[1], p.bounds[2]};
        mx = {p.bounds[3], p.bounds[4], p.bounds[5]};
    }
    else {
        mn = {ord2f(c->bounds_enc[0]), ord2f(c->bounds_enc[1]), ord2f(c->bounds_enc[2])};
        mx = {ord2f(c->bounds_enc[3]), ord2f(c->bounds_enc[4]), ord2f(c->bounds_enc[5])};
    }
    Affine a = compute_mesh_transform(mn, mx, p.S, p.unit);
    for (int i = 0; i < 3; ++i) {
        c->xform[i * 3 + 0] = a.m[i].x;
        c->xform[i * 3 + 1] = a.m[i].y;
        c->xform[i * 3 + 2] = a.m[i].z;
    }
    c->xform[9] = a.t.x;
    c->xform[10] = a.t.y;
    c->xform[11] = a.t.z;
}

// ---- slab planning: where to cut the grid so that N GPUs get equal work ------------------------------------
//
// Pipeline time is proportional to the number of (triangle, voxel) hits (measured: 0.23 ms per million on every
// slab of the weak-scaling job), and the hits of one triangle are predicted to ~0.1 % per slab by the Steiner-type
// count  A_x + A_y + A_z + (L1 perimeter) / 2 + 1  (projected areas and edge lengths in voxel units).  k_zhist
// spreads that estimate over the triangle's z layers into <= 2048 bins (fixed point, integer atomics: the result
// does not depend on the order of the adds, so every rank derives the same cuts).
constexpr uint32_t kPlanBins = 2048;
__global__ __launch_bounds__(kBlock) void k_zhist(const float *__restrict__ verts, const Counters *__restrict__ c,
                                                   unsigned long long *hist, float2 *zrange, float *zrange_xform,
                                                   Params p, uint32_t bin_h)
{
    __shared__ unsigned long long s_hist[kPlanBins];
    __shared__ float s_v[kBlock * 9];
    __shared__ float s_zr[2][kBlock / 64];
    if (blockIdx.x == 0 && threadIdx.x < 12) zrange_xform[threadIdx.x] = c->xform[threadIdx.x];
    for (uint32_t t = threadIdx.x; t < kPlanBins; t += kBlock) s_hist[t] = 0;
    Affine a;
    for (int i = 0; i < 3; ++i) a.m[i] = {c->xform[i * 3], c->xform[i * 3 + 1], c->xform[i * 3 + 2]};
    a.t = {c->xform[9], c->xform[10], c->xform[11]};
    for (uint64_t base = (uint64_t) blockIdx.x * kBlock; base < p.n_tris; base += (uint64_t) gridDim.x * kBlock) {
        __syncthreads();
        const uint32_t n_here = (uint32_t) (p.n_tris - base < kBlock ? p.n_tris - base : kBlock);
        for (uint32_t k = threadIdx.x; k < n_here * 9; k += kBlock) s_v[k] = verts[base * 9 + k];
        __syncthreads();
        const bool live = threadIdx.x < n_here;
        const float *q = &s_v[(live ? threadIdx.x : 0u) * 9];
        const V3 v0 = affine_apply(a, V3{q[0], q[1], q[2]}), v1 = affine_apply(a, V3{q[3], q[4], q[5]}),
                 v2 = affine_apply(a, V3{q[6], q[7], q[8]});
        {
            // z extent of this block of 256 triangles (the same float operations as k_expand_roots, so it can skip
            // the whole block when the extent misses its slab); a NaN disables the shortcut for the block
            const float inf = __builtin_inff();
            float blo = fmin2(v0.z, fmin2(v1.z, v2.z)), bhi = fmax2(v0.z, fmax2(v1.z, v2.z));
            if (!(v0.z == v0.z) || !(v1.z == v1.z) || !(v2.z == v2.z)) {
                blo = -inf;
                bhi = inf;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                blo = fminf(blo, __shfl_xor(blo, d, 64));
                bhi = fmaxf(bhi, __shfl_xor(bhi, d, 64));
            }
            if ((threadIdx.x & 63u) == 0) {
                s_zr[0][threadIdx.x >> 6] = blo;
                s_zr[1][threadIdx.x >> 6] = bhi;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                for (uint32_t wv = 1; wv < kBlock / 64; ++wv) {
                    blo = fminf(blo, s_zr[0][wv]);
                    bhi = fmaxf(bhi, s_zr[1][wv]);
                }
                zrange[base / kBlock] = make_float2(blo, bhi);
            }
        }
        if (!live) continue;
        const V3 n = tri_normal(v0, v1, v2), e0 = v1 - v0, e1 = v2 - v1, e2 = v0 - v2;
        float est = (abs_f(n.x) + abs_f(n.y) + abs_f(n.z)) * 0.5f +
                    (abs_f(e0.x) + abs_f(e0.y) + abs_f(e0.z) + abs_f(e1.x) + abs_f(e1.y) + abs_f(e1.z) + abs_f(e2.x) +
                     abs_f(e2.y) + abs_f(e2.z)) * 0.5f + 1.0f;
        if (!(est < 1e12f)) est = 1e12f;  // also catches NaN
        const float zlo = fmin2(v0.z, fmin2(v1.z, v2.z)), zhi = fmax2(v0.z, fmax2(v1.z, v2.z));
        if (!(zhi >= 0.f) || !(zlo < (float) p.S)) continue;
        const uint32_t l0 = zlo > 0.f ? (uint32_t) zlo : 0u;
        const uint32_t l1 = zhi < (float) (p.S - 1) ? (uint32_t) zhi : p.S - 1;
        const float per_layer = est * 16.0f / (float) (l1 - l0 + 1);
        for (uint32_t b = l0 / bin_h; b <= l1 / bin_h; ++b) {
            const uint32_t lo = b * bin_h > l0 ? b * bin_h : l0;
            const uint32_t hi = (b + 1) * bin_h - 1 < l1 ? (b + 1) * bin_h - 1 : l1;
            atomicAdd(&s_hist[b], (unsigned long long) (per_layer * (float) (hi - lo + 1) + 0.5f));
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < kPlanBins; t += kBlock)
        if (s_hist[t]) atomicAdd(&hist[t], s_hist[t]);
}

// ---- K1: leaves --------------------------------------------------------------------------------------------

struct Sub {  // a (sub-)triangle in registers
    V3 v0, v1, v2;
    V2 t0, t1, t2;
};

struct LeafPlan {
    uint32_t lo[3], d[3];
    uint32_t ntiles;  // 0 = nothing to do (outside the slab)
    uint64_t count;
};

// Clamp the voxel AABB of a leaf to the grid and the slab: voxelization.cpp:440-444 with min/max = slab bounds.
__device__ __forceinline__ LeafPlan plan_leaf(const Sub &s, const Params &p)
{
    LeafPlan pl;
    V3 mn = tri_min(s.v0, s.v1, s.v2), mx = tri_max(s.v0, s.v1, s.v2);
    uint32_t lo[3] = {floor_u32(mn.x), floor_u32(mn.y), floor_u32(mn.z)};
    uint32_t hi[3] = {floor_u32(mx.x) + 1u, floor_u32(mx.y) + 1u, floor_u32(mx.z) + 1u};
    uint32_t glo[3] = {0u, 0u, p.zs0}, ghi[3] = {p.S, p.S, p.zs1};
    bool empty = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = lo[a] > glo[a] ? lo[a] : glo[a];
        hi[a] = hi[a] < ghi[a] ? hi[a] : ghi[a];
        empty |= lo[a] >= hi[a];
        pl.lo[a] = lo[a];
        pl.d[a] = empty ? 0u : hi[a] - lo[a];
    }
    pl.count = empty ? 0ull : (uint64_t) pl.d[0] * pl.d[1] * pl.d[2];
    pl.ntiles = (uint32_t) ((pl.count + kTileSize - 1) / kTileSize);
    return pl;
}

// u32 voxel AABB volume, wrapping like the reference's Vec3u32 product (voxelization.cpp:357-361)
__device__ __forceinline__ uint32_t voxel_volume(const Sub &s)
{
    V3 mn = tri_min(s.v0, s.v1, s.v2), mx = tri_max(s.v0, s.v1, s.v2);
    uint32_t dx = (floor_u32(mx.x) + 1u) - floor_u32(mn.x);
    uint32_t dy = (floor_u32(mx.y) + 1u) - floor_u32(mn.y);
    uint32_t dz = (floor_u32(mx.z) + 1u) - floor_u32(mn.z);
    return dx * dy * dz;
}

// true if the sub-triangle's voxel AABB misses the slab entirely (then none of its descendants can touch it:
// midpoints stay inside the parent's AABB because rounding is monotonic)
__device__ __forceinline__ bool misses_slab(const Sub &s, const Params &p)
{
    V3 mn = tri_min(s.v0, s.v1, s.v2), mx = tri_max(s.v0, s.v1, s.v2);
    uint32_t zlo = floor_u32(mn.z), zhi = floor_u32(mx.z) + 1u;
    uint32_t xlo = floor_u32(mn.x), ylo = floor_u32(mn.y);
    return zhi <= p.zs0 || zlo >= p.zs1 || xlo >= p.S || ylo >= p.S;
}

__device__ __forceinline__ void write_leaf(Leaf *leaves, uint32_t idx, const Sub &s, uint32_t tri, uint32_t pathkey,
                                           float area, const LeafPlan &pl)
{
    V3 n = normalize(tri_normal(s.v0, s.v1, s.v2));  // voxelization.cpp:438
    Leaf l;
    l.v[0] = s.v0.x; l.v[1] = s.v0.y; l.v[2] = s.v0.z;
    l.v[3] = s.v1.x; l.v[4] = s.v1.y; l.v[5] = s.v1.z;
    l.v[6] = s.v2.x; l.v[7] = s.v2.y; l.v[8] = s.v2.z;
    l.n[0] = n.x; l.n[1] = n.y; l.n[2] = n.z;
    l.t[0] = s.t0.x; l.t[1] = s.t0.y; l.t[2] = s.t1.x; l.t[3] = s.t1.y; l.t[4] = s.t2.x; l.t[5] = s.t2.y;
    l.tri = tri;
    l.pathkey = pathkey;
    l.bmin_xy = pl.lo[0] | (pl.lo[1] << 16);
    l.bmin_z_dx = pl.lo[2] | (pl.d[0] << 16);
    l.dy_dz = pl.d[1] | (pl.d[2] << 16);
    l.area = area;
    leaves[idx] = l;
}

__device__ __forceinline__ void write_tiles(Tile *tiles, BigLeaf *big, uint32_t leaf_idx, uint32_t first_tile,
                                            uint32_t ntiles, uint32_t big_slot, const Params &p)
{
    if (ntiles <= kInlineTiles) {
        for (uint32_t k = 0; k < ntiles; ++k)
            if (first_tile + k < p.cap_tiles) tiles[first_tile + k] = Tile{leaf_idx, k * kTileSize};
    }
    else if (big_slot < p.cap_big) {
        big[big_slot] = BigLeaf{leaf_idx, first_tile, ntiles, 0};
    }
}

struct Emit {  // what one lane wants to append this round
    uint32_t n_leaf, n_tile, n_big, n_node;
};

struct BlockSlots {
    uint32_t leaf, tile, big, node;
};

// One reservation per counter per workgroup (a per-lane atomic on one address would serialise at ~88/us).  The four
// per-lane counts are packed into one 64-bit value (tiles: 31 bits; leaves, big leaves, nodes: 11 bits each, a lane
// emits at most 4 of each, so a block at most 1024) so that a single block scan yields all four offsets.
__device__ __forceinline__ BlockSlots reserve_slots(const Emit &e, Counters *c, uint32_t node_round, uint32_t *s_wave,
                                                    uint32_t *s_base)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long mine = (unsigned long long) e.n_tile | ((unsigned long long) e.n_leaf << 31) |
                                    ((unsigned long long) e.n_big << 42) | ((unsigned long long) e.n_node << 53);
    unsigned long long inc = mine;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    unsigned long long *s_wave64 = reinterpret_cast<unsigned long long *>(s_wave);  // [kBlock / 64], 8-byte aligned
    __syncthreads();
    if (lane == 63) s_wave64[wave] = inc;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < kBlock / 64; ++w) {
        const unsigned long long v = s_wave64[w];
        if (w < wave) base += v;
        tot += v;
    }
    const unsigned long long ex = base + inc - mine;
    const uint32_t tot_tile = (uint32_t) tot & 0x7fffffffu, tot_leaf = (uint32_t) (tot >> 31) & 2047u,
                   tot_big = (uint32_t) (tot >> 42) & 2047u, tot_node = (uint32_t) (tot >> 53) & 2047u;
    __syncthreads();
    if (threadIdx.x == 0) {
        s_base[0] = tot_leaf ? atomicAdd(&c->n_leaves, tot_leaf) : 0u;
        s_base[1] = tot_tile ? atomicAdd(&c->n_tiles, tot_tile) : 0u;
        s_base[2] = tot_big ? atomicAdd(&c->n_big, tot_big) : 0u;
        s_base[3] = tot_node ? atomicAdd(&c->n_nodes[node_round], tot_node) : 0u;
    }
    __syncthreads();
    BlockSlots off;
    off.tile = ((uint32_t) ex & 0x7fffffffu) + s_base[1];
    off.leaf = ((uint32_t) (ex >> 31) & 2047u) + s_base[0];
    off.big = ((uint32_t) (ex >> 42) & 2047u) + s_base[2];
    off.node = ((uint32_t) (ex >> 53) & 2047u) + s_base[3];
    return off;
}

// Roots: one lane per input triangle.  applyMeshTransform (obj2voxel.cpp:202-224) then the head of
// voxelizeTriangleToUvBuffer (voxelization.cpp:488-511).
__global__ __launch_bounds__(kBlock) void k_expand_roots(const float *__restrict__ verts, const float *__restrict__ uvs,
                                                         Counters *c, Leaf *leaves, Tile *tiles, BigLeaf *big,
                                                         Node *nodes_out, const float2 *__restrict__ zrange,
                                                         const float *__restrict__ zrange_xform, Params p)
{
    __shared__ __align__(8) uint32_t s_wave[2 * (kBlock / 64)];
    __shared__ uint32_t s_base[4];
    __shared__ float s_v[kBlock * 9];
    __shared__ float s_t[kBlock * 6];
    __shared__ unsigned long long s_cand;

    Affine xf;
    xf.m[0] = {c->xform[0], c->xform[1], c->xform[2]};
    xf.m[1] = {c->xform[3], c->xform[4], c->xform[5]};
    xf.m[2] = {c->xform[6], c->xform[7], c->xform[8]};
    xf.t = {c->xform[9], c->xform[10], c->xform[11]};

    // z extents per block of 256 triangles from the slab plan (k_zhist), valid if they were made with this transform
    bool use_zrange = zrange != nullptr;
    if (use_zrange)
        for (int i = 0; i < 12; ++i) use_zrange &= __float_as_uint(zrange_xform[i]) == __float_as_uint(c->xform[i]);

    const uint64_t n_blocks = (p.n_tris + kBlock - 1) / kBlock;
    for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        if (use_zrange) {
            // every triangle of the block fails misses_slab()'s z test (floor_u32 is monotonic), so none is read
            const float2 r = zrange[blk];
            if (r.y < 1e9f && (floor_u32(r.y) + 1u <= p.zs0 || floor_u32(r.x) >= p.zs1)) continue;
        }
        const uint64_t base = blk * kBlock;
        const uint32_t n_here = (uint32_t) (p.n_tris - base < kBlock ? p.n_tris - base : kBlock);
        __syncthreads();
        if (threadIdx.x == 0) s_cand = 0;
        // coalesced staging of this block's vertices / uvs through LDS
        for (uint32_t i = threadIdx.x; i < n_here * 9; i += kBlock) s_v[i] = verts[base * 9 + i];
        if (p.has_uv)
            for (uint32_t i = threadIdx.x; i < n_here * 6; i += kBlock) s_t[i] = uvs[base * 6 + i];
        __syncthreads();

        const bool live = threadIdx.x < n_here;
        Sub s{};
        Emit e{0, 0, 0, 0};
        LeafPlan pl{};
        float area = 0;
        bool as_leaf = false, as_node = false;
        if (live) {
            const float *q = &s_v[threadIdx.x * 9];
            s.v0 = affine_apply(xf, V3{q[0], q[1], q[2]});
            s.v1 = affine_apply(xf, V3{q[3], q[4], q[5]});
            s.v2 = affine_apply(xf, V3{q[6], q[7], q[8]});
            if (p.has_uv) {
                const float *r = &s_t[threadIdx.x * 6];
                s.t0 = {r[0], r[1]};
                s.t1 = {r[2], r[3]};
                s.t2 = {r[4], r[5]};
            }
            if (!misses_slab(s, p)) {
                area = tri_area(s.v0, s.v1, s.v2);
                if (roughly_axis_aligned(s.v0, s.v1, s.v2) || voxel_volume(s) < kSubdivisionVolumeLimit) {
                    pl = plan_leaf(s, p);
                    if (pl.count >> 32) {
                        atomicOr(&c->err_flags, kErrLeafTooLarge);
                    }
                    else if (pl.ntiles) {
                        as_leaf = true;
                        e.n_leaf = 1;
                        e.n_tile = pl.ntiles;
                        e.n_big = pl.ntiles > kInlineTiles ? 1u : 0u;
                    }
                }
                else {
                    as_node = true;
                    e.n_node = 1;
                }
            }
        }
        // a block whose triangles all miss this GPU's slab has nothing to reserve (the common case on the other
        // ranks of a multi-GPU run, where every rank filters the whole triangle list)
        if (!__syncthreads_or((int) (as_leaf || as_node))) continue;
        BlockSlots slot = reserve_slots(e, c, 0, s_wave, s_base);
        if (as_leaf) {
            if (slot.leaf < p.cap_leaves) write_leaf(leaves, slot.leaf, s, (uint32_t) (base + threadIdx.x), 0u, area, pl);
            write_tiles(tiles, big, slot.leaf, slot.tile, pl.ntiles, slot.big, p);
            atomicAdd(&s_cand, pl.count);
        }
        if (as_node && slot.node < p.cap_nodes) {
            Node n;
            n.v[0] = s.v0.x; n.v[1] = s.v0.y; n.v[2] = s.v0.z;
            n.v[3] = s.v1.x; n.v[4] = s.v1.y; n.v[5] = s.v1.z;
            n.v[6] = s.v2.x; n.v[7] = s.v2.y; n.v[8] = s.v2.z;
            n.t[0] = s.t0.x; n.t[1] = s.t0.y; n.t[2] = s.t1.x; n.t[3] = s.t1.y; n.t[4] = s.t2.x; n.t[5] = s.t2.y;
            n.tri = (uint32_t) (base + threadIdx.x);
            n.pathkey = 0;
            n.depth = 0;
            n.area = area;
            n.pad = 0;
            nodes_out[slot.node] = n;
        }
        __syncthreads();
        if (threadIdx.x == 0 && s_cand) atomicAdd(&c->n_candidates, s_cand);
    }
}

// One breadth-first round of forEachSubdividedTriangle (voxelization.cpp:349-379).  The reference pops a LIFO
// stack: after subdivide4 the centre piece (index 0) replaces the parent and pieces 1,2,3 are pushed, so the
// processing order of the children is 3, 2, 1, 0 (depth first).  A leaf's position in that order is encoded
// in `pathkey`: two bits (3 - childIndex) per level, most significant first, then a terminating 1 bit, so that
// unsigned comparison of keys of one triangle equals the reference's processing order.
__global__ __launch_bounds__(kBlock) void k_expand_nodes(const Node *__restrict__ nodes_in, uint32_t round, Counters *c,
                                                         Leaf *leaves, Tile *tiles, BigLeaf *big, Node *nodes_out,
                                                         Params p)
{
    __shared__ __align__(8) uint32_t s_wave[2 * (kBlock / 64)];
    __shared__ uint32_t s_base[4];
    __shared__ unsigned long long s_cand;
    const uint32_t n_in = c->n_nodes[round] < p.cap_nodes ? c->n_nodes[round] : p.cap_nodes;
    const uint32_t n_blocks = (n_in + kBlock - 1) / kBlock;
    for (uint32_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const uint32_t i = blk * kBlock + threadIdx.x;
        const bool live = i < n_in;
        __syncthreads();
        if (threadIdx.x == 0) s_cand = 0;
        Sub ch[4];
        LeafPlan pl[4];
        uint32_t kind[4] = {0, 0, 0, 0};  // 0 drop, 1 leaf, 2 node
        uint32_t tri = 0, pathkey = 0, depth = 0;
        float area = 0;
        Emit e{0, 0, 0, 0};
        if (live) {
            const Node n = nodes_in[i];
            tri = n.tri;
            pathkey = n.pathkey;
            depth = n.depth;
            area = n.area;
            V3 v0{n.v[0], n.v[1], n.v[2]}, v1{n.v[3], n.v[4], n.v[5]}, v2{n.v[6], n.v[7], n.v[8]};
            V2 t0{n.t[0], n.t[1]}, t1{n.t[2], n.t[3]}, t2{n.t[4], n.t[5]};
            // subdivide4, triangle.hpp:134-143
            V3 g0 = mix(v0, v1, 0.5f), g1 = mix(v1, v2, 0.5f), g2 = mix(v2, v0, 0.5f);
            V2 x0 = mix(t0, t1, 0.5f), x1 = mix(t1, t2, 0.5f), x2 = mix(t2, t0, 0.5f);
            ch[0] = Sub{g0, g1, g2, x0, x1, x2};
            ch[1] = Sub{v0, g0, g2, t0, x0, x2};
            ch[2] = Sub{v1, g1, g0, t1, x1, x0};
            ch[3] = Sub{v2, g2, g1, t2, x2, x1};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (misses_slab(ch[k], p)) continue;
                if (voxel_volume(ch[k]) < kSubdivisionVolumeLimit) {
                    pl[k] = plan_leaf(ch[k], p);
                    if (pl[k].count >> 32) {
                        atomicOr(&c->err_flags, kErrLeafTooLarge);
                    }
                    else if (pl[k].ntiles) {
                        kind[k] = 1;
                        e.n_leaf += 1;
                        e.n_tile += pl[k].ntiles;
                        e.n_big += pl[k].ntiles > kInlineTiles ? 1u : 0u;
                    }
                }
                else if (depth + 1 >= 15) {
                    atomicOr(&c->err_flags, kErrDepth);
                }
                else {
                    kind[k] = 2;
                    e.n_node += 1;
                }
            }
        }
        BlockSlots slot = reserve_slots(e, c, round + 1, s_wave, s_base);
        if (live) {
            unsigned long long cand = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // child digit at this level, then the terminator bit one position below it
                const uint32_t shift = 30u - 2u * depth;
                const uint32_t digit_key = pathkey | ((3u - (uint32_t) k) << shift);
                if (kind[k] == 1) {
                    const uint32_t key = digit_key | (1u << (shift - 1u));
                    if (slot.leaf < p.cap_leaves) write_leaf(leaves, slot.leaf, ch[k], tri, key, area, pl[k]);
                    write_tiles(tiles, big, slot.leaf, slot.tile, pl[k].ntiles, slot.big, p);
                    cand += pl[k].count;
                    slot.leaf += 1;
                    slot.tile += pl[k].ntiles;
                    slot.big += pl[k].ntiles > kInlineTiles ? 1u : 0u;
                }
                else if (kind[k] == 2) {
                    if (slot.node < p.cap_nodes) {
                        Node o;
                        const Sub &s = ch[k];
                        o.v[0] = s.v0.x; o.v[1] = s.v0.y; o.v[2] = s.v0.z;
                        o.v[3] = s.v1.x; o.v[4] = s.v1.y; o.v[5] = s.v1.z;
                        o.v[6] = s.v2.x; o.v[7] = s.v2.y; o.v[8] = s.v2.z;
                        o.t[0] = s.t0.x; o.t[1] = s.t0.y; o.t[2] = s.t1.x; o.t[3] = s.t1.y; o.t[4] = s.t2.x; o.t[5] = s.t2.y;
                        o.tri = tri;
                        o.pathkey = digit_key;
                        o.depth = depth + 1;
                        o.area = area;
                        o.pad = 0;
                        nodes_out[slot.node] = o;
                    }
                    slot.node += 1;
                }
            }
            if (cand) atomicAdd(&s_cand, cand);
        }
        __syncthreads();
        if (threadIdx.x == 0 && s_cand) atomicAdd(&c->n_candidates, s_cand);
    }
}


__global__ __launch_bounds__(kBlock) void k_expand_big(const BigLeaf *__restrict__ big, const Counters *c, Tile *tiles,
                                                       Params p)
{
    const uint32_t n = c->n_big < p.cap_big ? c->n_big : p.cap_big;
    for (uint32_t b = blockIdx.x; b < n; b += gridDim.x) {
        const BigLeaf bl = big[b];
        for (uint32_t k = threadIdx.x; k < bl.ntiles; k += kBlock)
            if (bl.first_tile + k < p.cap_tiles) tiles[bl.first_tile + k] = Tile{bl.leaf, k * kTileSize};
    }
}

// ---- K2: voxelize -------------------------------------------------------------------------------------------

template <bool UV>
struct Piece {  // TexturedTriangle (triangle.hpp:113-144); the uv members are dead code when !UV
    V3 a, b, c;
    V2 ta, tb, tc;
};

// Classification of one piece against one axis plane: SplittingValues + the case switch of splitTriangle
// (voxelization.cpp:110-153,190-232).  Packed so that it can be carried in one register between the cheap
// classification pass and the expensive split pass of the clip loop.
enum : uint32_t {
    kClsModeMask = 3u,   // 0: whole triangle goes to one side, 1: one-planar split, 2: regular split
    kClsSideLo = 4u,     // mode 0: the side is "lo"
    kClsRotShift = 3u,   // bits 3..4: rotation index r (planar vertex for mode 1, isolated vertex for mode 2)
    kClsFlagLo = 32u,    // mode 1: lo flag of vertex r+1;  mode 2: the isolated vertex is lo
};

__device__ __forceinline__ uint32_t classify_piece(float c0, float c1, float c2, float plane)
{
    // SplittingValues, voxelization.cpp:121-131
    const bool p0 = abs_f(c0 - plane) < kEpsilon, p1 = abs_f(c1 - plane) < kEpsilon, p2 = abs_f(c2 - plane) < kEpsilon;
    const bool l0 = c0 < plane, l1 = c1 < plane, l2 = c2 < plane;
    const uint32_t lo_sum = (uint32_t) l0 + (uint32_t) l1 + (uint32_t) l2;
    const uint32_t pl_sum = (uint32_t) p0 + (uint32_t) p1 + (uint32_t) p2;
    if (lo_sum == 0) return 0u;
    if (lo_sum == 3) return kClsSideLo;
    if (pl_sum == 3) return 0u;  // parallel to the plane: pushed by bias (IS_LO_BIASED = false) = hi
    if (pl_sum == 2) return (!p0 ? l0 : (!p1 ? l1 : l2)) ? kClsSideLo : 0u;
    if (pl_sum == 1) {
        const uint32_t r = p0 ? 0u : (p1 ? 1u : 2u);
        const bool lq = r == 0 ? l1 : (r == 1 ? l2 : l0);
        const bool lr = r == 0 ? l2 : (r == 1 ? l0 : l1);
        if (lq == lr) return lq ? kClsSideLo : 0u;
        return 1u | (r << kClsRotShift) | (lq ? kClsFlagLo : 0u);
    }
    const bool iso_lo = lo_sum == 1;
    const uint32_t r = iso_lo ? (l0 ? 0u : (l1 ? 1u : 2u)) : (!l0 ? 0u : (!l1 ? 1u : 2u));
    return 2u | (r << kClsRotShift) | (iso_lo ? kClsFlagLo : 0u);
}

// The geometric part of splitTriangle<DISCARD_LO|DISCARD_HI> for a piece whose classification says it is cut
// (modes 1 and 2).  keep_lo selects DISCARD_HI.  Returns the number of kept pieces (1 or 2): `cur` becomes the
// first kept piece in emission order, `sec` the second.  Vertex order inside emitted pieces is the reference's,
// because later splits depend on it.
template <bool UV>
__device__ __forceinline__ uint32_t split_cut(Piece<UV> &cur, Piece<UV> &sec, uint32_t cls, uint32_t axis, float plane,
                                              bool keep_lo)
{
    const uint32_t r = (cls >> kClsRotShift) & 3u;
    const bool flag_lo = (cls & kClsFlagLo) != 0;
    // rotate (a, b, c) left by r with two conditional cyclic shifts (18 selects instead of 54)
    const bool s1 = r >= 1u, s2 = r == 2u;
    const V3 P1 = s1 ? cur.b : cur.a, Q1 = s1 ? cur.c : cur.b, R1 = s1 ? cur.a : cur.c;
    const V3 P = s2 ? Q1 : P1, Q = s2 ? R1 : Q1, R = s2 ? P1 : R1;
    V2 tP{}, tQ{}, tR{};
    if (UV) {
        const V2 tP1 = s1 ? cur.tb : cur.ta, tQ1 = s1 ? cur.tc : cur.tb, tR1 = s1 ? cur.ta : cur.tc;
        tP = s2 ? tQ1 : tP1;
        tQ = s2 ? tR1 : tQ1;
        tR = s2 ? tP1 : tR1;
    }
    const float cP = comp(P, axis), cQ = comp(Q, axis), cR = comp(R, axis);
    const bool regular = (cls & kClsModeMask) == 2u;
    // first intersection: regular case P->Q (voxelization.cpp:305-311), one-planar case Q->R (:262-266)
    const V3 A0 = regular ? P : Q, A1 = regular ? Q : R;
    const float cA0 = regular ? cP : cQ, cA1 = regular ? cQ : cR;
    const float d0 = -(cA1 - cA0);
    const float i0 = abs_f(d0) < kEpsilon ? 0.f : (cA0 - plane) / d0;
    const V3 G0 = mix(A0, A1, i0);
    V2 x0{};
    if (UV) x0 = mix(regular ? tP : tQ, regular ? tQ : tR, i0);
    if (!regular) {
        // splitTriangle_onePlanarCase: {P,Q,G} goes to Q's side, {P,G,R} to the other
        if (flag_lo == keep_lo) {
            cur.a = P; cur.b = Q; cur.c = G0;
            if (UV) { cur.ta = tP; cur.tb = tQ; cur.tc = x0; }
        }
        else {
            cur.a = P; cur.b = G0; cur.c = R;
            if (UV) { cur.ta = tP; cur.tb = x0; cur.tc = tR; }
        }
        return 1;
    }
    // splitTriangle_regularCase, voxelization.cpp:279-331: P isolated, second intersection P->R
    const float d1 = -(cR - cP);
    const float i1 = abs_f(d1) < kEpsilon ? 0.f : (cP - plane) / d1;
    const V3 G1 = mix(P, R, i1);
    V2 x1{};
    if (UV) x1 = mix(tP, tR, i1);
    if (flag_lo == keep_lo) {
        cur.a = P; cur.b = G0; cur.c = G1;
        if (UV) { cur.ta = tP; cur.tb = x0; cur.tc = x1; }
        return 1;
    }
    cur.a = G0; cur.b = Q; cur.c = R;
    sec.a = G0; sec.b = G1; sec.c = R;
    if (UV) {
        cur.ta = x0; cur.tb = tQ; cur.tc = tR;
        sec.ta = x0; sec.tb = x1; sec.tc = tR;
    }
    return 2;
}

template <bool UV>
__device__ __forceinline__ void accumulate_piece(const Piece<UV> &pc, float area, float &w, float &u, float &v)
{
    // result = mix(result, {area(inputTriangle), piece.textureCenter()}), voxelization.cpp:414-420, util.hpp:160-165
    const float ws = w + area;
    if (UV) {
        const float uc = ((pc.ta.x + pc.tb.x) + pc.tc.x) / 3;
        const float vc = ((pc.ta.y + pc.tb.y) + pc.tc.y) / 3;
        u = (w * u + area * uc) / ws;
        v = (w * v + area * vc) / ws;
    }
    w = ws;
}

// Pending sibling pieces of the depth-first clip walk, one slot per level 1..5, held in registers: every access
// uses a compile-time slot index (selected by a switch), so the array never leaves the VGPR file.
template <bool UV>
struct PieceStack {
    Piece<UV> s0, s1, s2, s3, s4;
};

// Value-level selects (v_cndmask), not control flow: a branchy form gets folded by the compiler into a select of
// addresses, which forces the stack into scratch memory.
template <bool UV>
__device__ __forceinline__ Piece<UV> sel_piece(bool take_x, const Piece<UV> &x, const Piece<UV> &y)
{
    Piece<UV> r;
    r.a = {take_x ? x.a.x : y.a.x, take_x ? x.a.y : y.a.y, take_x ? x.a.z : y.a.z};
    r.b = {take_x ? x.b.x : y.b.x, take_x ? x.b.y : y.b.y, take_x ? x.b.z : y.b.z};
    r.c = {take_x ? x.c.x : y.c.x, take_x ? x.c.y : y.c.y, take_x ? x.c.z : y.c.z};
    if (UV) {
        r.ta = {take_x ? x.ta.x : y.ta.x, take_x ? x.ta.y : y.ta.y};
        r.tb = {take_x ? x.tb.x : y.tb.x, take_x ? x.tb.y : y.tb.y};
        r.tc = {take_x ? x.tc.x : y.tc.x, take_x ? x.tc.y : y.tc.y};
    }
    return r;
}
template <bool UV>
__device__ __forceinline__ void stack_store(PieceStack<UV> &st, uint32_t slot, const Piece<UV> &pc)
{
    st.s0 = sel_piece<UV>(slot == 0, pc, st.s0);
    st.s1 = sel_piece<UV>(slot == 1, pc, st.s1);
    st.s2 = sel_piece<UV>(slot == 2, pc, st.s2);
    st.s3 = sel_piece<UV>(slot == 3, pc, st.s3);
    st.s4 = sel_piece<UV>(slot == 4, pc, st.s4);
}
template <bool UV>
__device__ __forceinline__ void stack_load(const PieceStack<UV> &st, uint32_t slot, Piece<UV> &pc)
{
    Piece<UV> r = st.s4;
    r = sel_piece<UV>(slot == 3, st.s3, r);
    r = sel_piece<UV>(slot == 2, st.s2, r);
    r = sel_piece<UV>(slot == 1, st.s1, r);
    r = sel_piece<UV>(slot == 0, st.s0, r);
    pc = r;
}

// Conservative triangle / voxel overlap test (separating axes: the triangle's plane and the nine edge x axis
// directions; the three box axes are implied by the AABB walk).  The box is inflated by kSatMargin, far more than
// the float32 rounding of the clip (<= a few ulp of the coordinate, 5e-4 at 4096) and than its planarity epsilon
// (2^-16), so every voxel the exact clip can mark is kept: this only removes work, never results.  The test's own
// rounding is covered too: everything is evaluated in the voxel-centred frame, the plane axis uses the unnormalised
// normal e0 x e1 with an explicit error bound (for a sliver, whose normal direction is numerically meaningless, the
// bound exceeds the radius and the plane axis simply never separates), and all comparisons are written so that a
// NaN rejects nothing.
constexpr float kSatMargin = 0.02f;

__device__ __forceinline__ bool sat_axis_separates(float p0, float p1, float rad)
{
    const float lo = p0 < p1 ? p0 : p1, hi = p0 < p1 ? p1 : p0;
    return lo > rad || hi < -rad;
}

__device__ __forceinline__ bool sat_may_overlap(V3 v0, V3 v1, V3 v2, float cx, float cy, float cz)
{
    const float h = 0.5f + kSatMargin;
    const V3 c{cx, cy, cz};
    const V3 a = v0 - c, b = v1 - c, d = v2 - c;
    const V3 e0 = b - a, e1 = d - b, e2 = a - d;
    {
        // plane axis: |n . a| <= h * |n|_1, n = e0 x e1.  Each component of n carries an absolute rounding error of a
        // few ulp of |e0|_1 |e1|_1 (cancellation), which the bound below over-estimates by more than 10x.
        const V3 n = cross(e0, e1);
        const float dist = n.x * a.x + n.y * a.y + n.z * a.z;
        const float rad = h * (abs_f(n.x) + abs_f(n.y) + abs_f(n.z));
        const float l0 = abs_f(e0.x) + abs_f(e0.y) + abs_f(e0.z), l1 = abs_f(e1.x) + abs_f(e1.y) + abs_f(e1.z);
        const float la = abs_f(a.x) + abs_f(a.y) + abs_f(a.z);
        const float err = 1e-5f * l0 * l1 * (la + 1.0f);
        if (abs_f(dist) > rad + err) return false;
    }
    // axis = X x e: projections use only the vertices not on edge e (the edge's own vertices project equally)
#define O2V_SAT_EDGE(E, U, W)                                                                               \
    if (sat_axis_separates(E.z * U.y - E.y * U.z, E.z * W.y - E.y * W.z, h * (abs_f(E.z) + abs_f(E.y)))) return false; \
    if (sat_axis_separates(E.x * U.z - E.z * U.x, E.x * W.z - E.z * W.x, h * (abs_f(E.x) + abs_f(E.z)))) return false; \
    if (sat_axis_separates(E.y * U.x - E.x * U.y, E.y * W.x - E.x * W.y, h * (abs_f(E.y) + abs_f(E.x)))) return false;
    O2V_SAT_EDGE(e0, a, d)
    O2V_SAT_EDGE(e1, b, a)
    O2V_SAT_EDGE(e2, d, b)
#undef O2V_SAT_EDGE
    return true;
}

constexpr uint32_t kLeafStride = 25;          // dwords per staged leaf in LDS (24 + 1 pad: spreads banks)
constexpr uint32_t kMaxSurvivors = 8192;      // survivor queue entries (= candidate voxels) per sub-batch

// K2.  Persistent workgroups pull batches of tiles.  Per batch:
//   phase 1  every candidate voxel of the tiles: decode, plane-distance cull (voxelization.cpp:451-458), SAT
//            pre-test; survivors are queued in LDS (tile slot + index in tile)
//   phase 2  persistent lanes pop survivors and run computeTrianglesUvInVoxel (voxelization.cpp:383-424) as a
//            depth-first walk of the split tree: the reference clips level by level with two 64-entry buffers;
//            visiting the first emitted piece first reproduces its buffer order, so the running mean of
//            :414-420 accumulates in the identical sequence.  Under DISCARD every split keeps <= 2 pieces, so at
//            most one sibling per level 1..5 is pending (register stack).  Per iteration a lane skips every
//            plane its piece passes whole (AABB check), classifies it against the first plane it does not, and
//            cuts if needed; lanes that run out of pieces pop the next survivor, so the wavefront stays full.
// Register budget: 4 waves per SIMD without uv arithmetic, 3 with it (the allocator spills a handful of cold values;
// measured faster than running one wave fewer on the bench mesh and on the large textured workloads).
template <bool UV>
__global__ __launch_bounds__(kBlock, (UV ? 3 : 4)) void k_voxelize(const Leaf *__restrict__ leaves, const Tile *__restrict__ tiles,
                                                     Counters *c, uint32_t *grid, uint8_t *brick_dirty, HitRec *pool,
                                                     Params p)
{
    __shared__ uint32_t s_leaf[kTilesPerBatch * kLeafStride];
    __shared__ uint32_t s_tleaf[kTilesPerBatch];
    __shared__ uint32_t s_tstart[kTilesPerBatch];
    __shared__ uint32_t s_tcount[kTilesPerBatch];
    __shared__ uint32_t s_tprefix[kTilesPerBatch + 2];
    __shared__ uint32_t s_scan[kBlock / 64];
    __shared__ uint32_t s_tend;
    __shared__ uint32_t s_chunk_tile[kMaxSurvivors / 64 + 1];
    __shared__ float s_inv_dx[kTilesPerBatch], s_inv_dy[kTilesPerBatch];
    __shared__ uint16_t s_surv[kMaxSurvivors];
    __shared__ uint32_t s_batch, s_nsurv, s_next, s_hits;

    if (expand_overflowed(c, p)) return;
    const uint32_t n_tiles = c->n_tiles < p.cap_tiles ? c->n_tiles : p.cap_tiles;
    // Batch size: about six batches per workgroup, between one tile per wavefront and what the LDS staging holds.  Few
    // large batches leave workgroups idle at the end of the kernel (and a 96^3 job, a few thousand tiles, would keep 3 %
    // of the machine busy); many small ones pay the per-batch staging and barriers too often.  Measured on seven
    // workload shapes.
    uint32_t tiles_per_batch = (n_tiles + gridDim.x * 6u - 1u) / (gridDim.x * 6u);
    tiles_per_batch = tiles_per_batch < kMinTilesPerBatch ? kMinTilesPerBatch
                      : (tiles_per_batch > kTilesPerBatch ? kTilesPerBatch : tiles_per_batch);
    const uint32_t n_batches = (n_tiles + tiles_per_batch - 1) / tiles_per_batch;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t chunk_base = 0, chunk_used = kHitChunk;  // wave-uniform; forces a reservation at first use
    if (threadIdx.x == 0) s_hits = 0;

    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_batch = atomicAdd(&c->batch_cursor, 1u);
        __syncthreads();
        const uint32_t batch = s_batch;
        if (batch >= n_batches) break;
        const uint32_t first = batch * tiles_per_batch;
        const uint32_t nt = n_tiles - first < tiles_per_batch ? n_tiles - first : tiles_per_batch;
        if (threadIdx.x < nt) {
            const Tile t = tiles[first + threadIdx.x];
            s_tleaf[threadIdx.x] = t.leaf;
            s_tstart[threadIdx.x] = t.start;
        }
        __syncthreads();
        // stage the leaves of this batch in LDS
        for (uint32_t i = threadIdx.x; i < nt * 24u; i += kBlock) {
            const uint32_t k = i / 24u, j = i - k * 24u;
            s_leaf[k * kLeafStride + j] = reinterpret_cast<const uint32_t *>(leaves + s_tleaf[k])[j];
        }
        __syncthreads();
        uint32_t my_count = 0;
        if (threadIdx.x < nt) {
            const uint32_t *lf = &s_leaf[threadIdx.x * kLeafStride];
            const uint32_t dx = lf[21] >> 16, dy = lf[22] & 0xffffu, dz = lf[22] >> 16;
            const uint32_t rem = dx * dy * dz - s_tstart[threadIdx.x];
            my_count = rem < kTileSize ? rem : kTileSize;
            s_tcount[threadIdx.x] = my_count;
            s_inv_dx[threadIdx.x] = 1.0f / (float) dx;
            s_inv_dy[threadIdx.x] = 1.0f / (float) dy;
        }
        {
            // exclusive prefix of the tile sizes: s_tprefix[k] = candidates before tile k, s_tprefix[nt] = total
            uint32_t total;
            const uint32_t ex = block_exscan(my_count, s_scan, total);
            if (threadIdx.x <= nt) s_tprefix[threadIdx.x] = threadIdx.x < nt ? ex : total;
        }

        // sub-batches of whole tiles with at most kMaxSurvivors candidates
        uint32_t t_begin = 0;
        while (t_begin < nt) {
            __syncthreads();
            const uint32_t base_cand = s_tprefix[t_begin];
            // the last tile whose end still fits decides t_end (found by the thread that owns it)
            if (threadIdx.x >= t_begin && threadIdx.x < nt) {
                const bool fits = s_tprefix[threadIdx.x + 1] - base_cand <= kMaxSurvivors;
                const bool next_fits = threadIdx.x + 1 < nt && s_tprefix[threadIdx.x + 2] - base_cand <= kMaxSurvivors;
                if (fits && !next_fits) s_tend = threadIdx.x + 1;
            }
            if (threadIdx.x == 0) {
                s_nsurv = 0;
                s_next = 0;
            }
            __syncthreads();
            const uint32_t t_end = s_tend;

            // ---- phase 1: the sub-batch's candidates flattened over the lanes ------------------------------
            // Candidate g (in sub-batch order) belongs to the tile k with s_tprefix[k] <= g < s_tprefix[k + 1].  A small
            // table gives every 64-candidate chunk the tile its first candidate falls in; a lane then walks forward a
            // few tiles at most, so the 64 lanes stay busy however small the tiles are.
            const uint32_t n_cand = s_tprefix[t_end] - base_cand;
            if (threadIdx.x >= t_begin && threadIdx.x < t_end) {
                const uint32_t lo = s_tprefix[threadIdx.x] - base_cand, hi = s_tprefix[threadIdx.x + 1] - base_cand;
                if (hi > lo)
                    for (uint32_t ch = (lo + 63u) / 64u; ch * 64u < hi; ++ch) s_chunk_tile[ch] = threadIdx.x;
            }
            __syncthreads();
            for (uint32_t g0 = wave * 64u; g0 < n_cand; g0 += kBlock) {
                const uint32_t g = g0 + lane;
                bool keep = false;
                uint32_t k = s_chunk_tile[g0 / 64u], i = 0;
                if (g < n_cand) {
                    while (s_tprefix[k + 1] - base_cand <= g) ++k;
                    i = g - (s_tprefix[k] - base_cand);
                    const uint32_t *lf = &s_leaf[k * kLeafStride];
                    const uint32_t dx = lf[21] >> 16, dy = lf[22] & 0xffffu;
                    const uint32_t j = s_tstart[k] + i;
                    uint32_t row, lx, lz, ly;
                    if (j < (1u << 24)) {
                        // exact quotient from a float estimate (j < 2^24, divisor < 2^16): off by at most one
                        row = (uint32_t) ((float) j * s_inv_dx[k]);
                        int32_t rx = (int32_t) (j - row * dx);
                        if (rx < 0) { row -= 1; rx += (int32_t) dx; }
                        else if ((uint32_t) rx >= dx) { row += 1; rx -= (int32_t) dx; }
                        lx = (uint32_t) rx;
                        lz = (uint32_t) ((float) row * s_inv_dy[k]);
                        int32_t ry = (int32_t) (row - lz * dy);
                        if (ry < 0) { lz -= 1; ry += (int32_t) dy; }
                        else if ((uint32_t) ry >= dy) { lz += 1; ry -= (int32_t) dy; }
                        ly = (uint32_t) ry;
                    }
                    else {
                        row = j / dx;
                        lx = j - row * dx;
                        lz = row / dy;
                        ly = row - lz * dy;
                    }
                    const V3 v0{__uint_as_float(lf[0]), __uint_as_float(lf[1]), __uint_as_float(lf[2])};
                    const V3 v1{__uint_as_float(lf[3]), __uint_as_float(lf[4]), __uint_as_float(lf[5])};
                    const V3 v2{__uint_as_float(lf[6]), __uint_as_float(lf[7]), __uint_as_float(lf[8])};
                    const V3 nrm{__uint_as_float(lf[9]), __uint_as_float(lf[10]), __uint_as_float(lf[11])};
                    const float cx = (float) ((lf[20] & 0xffffu) + lx) + 0.5f, cy = (float) ((lf[20] >> 16) + ly) + 0.5f,
                                cz = (float) ((lf[21] & 0xffffu) + lz) + 0.5f;
                    // plane distance cull, voxelization.cpp:451-458
                    const float sd = dot(nrm, V3{cx, cy, cz} - v0);
                    keep = !(abs_f(sd) > kPlaneDistanceLimit) && sat_may_overlap(v0, v1, v2, cx, cy, cz);
                }
                const unsigned long long m = __ballot(keep);
                if (m) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&s_nsurv, (uint32_t) __popcll(m));
                    base = __shfl(base, 0, 64);
                    if (keep) s_surv[base + (uint32_t) __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t) (((k - t_begin) << 8) | i);
                }
            }
            __syncthreads();
            const uint32_t n_surv = s_nsurv;

            // ---- phase 2: persistent lanes ------------------------------------------------------------------
            Piece<UV> cur{}, sec{};
            PieceStack<UV> stack{};
            uint32_t level = 0, pending = 0, my_k = 0;
            bool active = false, has_job = false;
            float w = 0.f, u = 0.f, v = 0.f, area = 0.f;
            float fx = 0.f, fy = 0.f, fz = 0.f;  // float(pos): the lower planes; upper planes are +1
            uint32_t px = 0, py = 0, pz = 0;
            bool queue_empty = n_surv == 0;
            // parked result of this lane's last finished hit
            float d_w = 0.f, d_u = 0.f, d_v = 0.f;
            uint32_t d_xy = 0, d_zk = 0;  // voxel x | y << 16, z | tile slot << 16 (all below 2^16)
            bool d_valid = false;
            auto flush_results = [&]() {
                const unsigned long long mask = __ballot(d_valid);
                if (!mask) return;
                const uint32_t cnt = (uint32_t) __popcll(mask);
                const uint32_t leader = (uint32_t) __ffsll((long long) mask) - 1u;
                if (chunk_used + cnt > kHitChunk) {
                    // abandon the rest of the chunk (marked as holes for the scatter pass) and reserve a new one
                    const uint32_t hole = chunk_base + chunk_used + lane;
                    if (chunk_used + lane < kHitChunk && hole < p.cap_hits) pool[hole].brick = kHoleBrick;
                    if (chunk_used + 64u + lane < kHitChunk && hole + 64u < p.cap_hits) pool[hole + 64u].brick = kHoleBrick;
                    if (chunk_used + 128u + lane < kHitChunk && hole + 128u < p.cap_hits) pool[hole + 128u].brick = kHoleBrick;
                    if (chunk_used + 192u + lane < kHitChunk && hole + 192u < p.cap_hits) pool[hole + 192u].brick = kHoleBrick;
                    uint32_t base = 0;
                    if (lane == leader) base = atomicAdd(&c->n_hits_reserved, kHitChunk);
                    chunk_base = __shfl(base, (int) leader, 64);
                    chunk_used = 0;
                }
                const uint32_t mine = chunk_base + chunk_used + (uint32_t) __popcll(mask & ((1ull << lane) - 1ull));
                chunk_used += cnt;
                if (d_valid && mine < p.cap_hits) {
                    const uint32_t d_px = d_xy & 0xffffu, d_py = d_xy >> 16, d_pz = d_zk & 0xffffu;
                    const uint32_t *lf = &s_leaf[(d_zk >> 16) * kLeafStride];
                    const uint32_t ox = d_px >> p.ss_shift, oy = d_py >> p.ss_shift, oz = d_pz >> p.ss_shift;
                    uint32_t brick;
                    const uint64_t cell = cell_index(ox, oy, oz - p.zo0, p, brick);
                    const uint32_t sub = p.ss_shift ? ((d_px & 1u) | ((d_py & 1u) << 1) | ((d_pz & 1u) << 2)) : 0u;
                    // the cell's counter hands out this hit's rank; k_scan_bricks turns the counts into offsets
                    const uint32_t rank = atomicAdd(&grid[cell], 1u);
                    if (rank >= kMaxRank) atomicOr(&c->err_flags, kErrRank);
                    brick_dirty[brick] = 1;  // benign race: every writer stores the same value
                    pool[mine] = HitRec{brick, (((uint32_t) cell & 255u) << 24) | (rank & (kMaxRank - 1u)),
                                        (sub << 29) | lf[18], lf[19], d_w, d_u, d_v, 0u};
                }
                if (lane == leader) atomicAdd(&s_hits, cnt);
                d_valid = false;
            };
            for (;;) {
                // pop a pending sibling, or fetch the next survivor
                if (!active) {
                    if (pending) {
                        level = 31u - (uint32_t) __clz((int) pending);
                        pending ^= 1u << level;
                        stack_load<UV>(stack, level - 1u, cur);
                        active = true;
                    }
                    else if (!queue_empty) {
                        const uint32_t q = atomicAdd(&s_next, 1u);
                        if (q < n_surv) {
                            const uint32_t e = s_surv[q];
                            my_k = t_begin + (e >> 8);
                            const uint32_t *lf = &s_leaf[my_k * kLeafStride];
                            const uint32_t dx = lf[21] >> 16, dy = lf[22] & 0xffffu;
                            const uint32_t j = s_tstart[my_k] + (e & 255u);
                            uint32_t row, lx, ly, lz;
                            if (j < (1u << 24)) {
                                row = (uint32_t) ((float) j * s_inv_dx[my_k]);
                                int32_t rx = (int32_t) (j - row * dx);
                                if (rx < 0) { row -= 1; rx += (int32_t) dx; }
                                else if ((uint32_t) rx >= dx) { row += 1; rx -= (int32_t) dx; }
                                lx = (uint32_t) rx;
                                lz = (uint32_t) ((float) row * s_inv_dy[my_k]);
                                int32_t ry = (int32_t) (row - lz * dy);
                                if (ry < 0) { lz -= 1; ry += (int32_t) dy; }
                                else if ((uint32_t) ry >= dy) { lz += 1; ry -= (int32_t) dy; }
                                ly = (uint32_t) ry;
                            }
                            else {
                                row = j / dx;
                                lx = j - row * dx;
                                lz = row / dy;
                                ly = row - lz * dy;
                            }
                            px = (lf[20] & 0xffffu) + lx;
                            py = (lf[20] >> 16) + ly;
                            pz = (lf[21] & 0xffffu) + lz;
                            fx = (float) px;
                            fy = (float) py;
                            fz = (float) pz;
                            cur.a = {__uint_as_float(lf[0]), __uint_as_float(lf[1]), __uint_as_float(lf[2])};
                            cur.b = {__uint_as_float(lf[3]), __uint_as_float(lf[4]), __uint_as_float(lf[5])};
                            cur.c = {__uint_as_float(lf[6]), __uint_as_float(lf[7]), __uint_as_float(lf[8])};
                            if (UV) {
                                cur.ta = {__uint_as_float(lf[12]), __uint_as_float(lf[13])};
                                cur.tb = {__uint_as_float(lf[14]), __uint_as_float(lf[15])};
                                cur.tc = {__uint_as_float(lf[16]), __uint_as_float(lf[17])};
                            }
                            area = __uint_as_float(lf[23]);
                            level = 0;
                            w = 0.f;
                            u = 0.f;
                            v = 0.f;
                            active = true;
                            has_job = true;
                        }
                        else {
                            queue_empty = true;
                        }
                    }
                }
                if (active) {
                    // Skip ahead: a piece whose vertices all satisfy v >= plane (lower planes) or v < plane (upper
                    // planes) is the loSum == 0 / loSum == 3 case of splitTriangle (voxelization.cpp:194-205) and
                    // passes whole, so every such plane from `level` on is skipped at once.
                    const V3 mn = tri_min(cur.a, cur.b, cur.c), mx = tri_max(cur.a, cur.b, cur.c);
                    uint32_t fail = 0;
                    fail |= (mn.x >= fx) ? 0u : 1u;
                    fail |= (mn.y >= fy) ? 0u : 2u;
                    fail |= (mn.z >= fz) ? 0u : 4u;
                    fail |= (mx.x < fx + 1.0f) ? 0u : 8u;
                    fail |= (mx.y < fy + 1.0f) ? 0u : 16u;
                    fail |= (mx.z < fz + 1.0f) ? 0u : 32u;
                    fail &= ~((1u << level) - 1u);
                    if (fail == 0) {
                        accumulate_piece<UV>(cur, area, w, u, v);  // inside all remaining planes
                        active = false;
                    }
                    else {
                        level = (uint32_t) __ffs((int) fail) - 1u;
                        const bool keep_lo = level >= 3u;
                        const uint32_t axis = keep_lo ? level - 3u : level;
                        const float plane = (axis == 0 ? fx : (axis == 1 ? fy : fz)) + (keep_lo ? 1.0f : 0.0f);
                        const uint32_t cls = classify_piece(comp(cur.a, axis), comp(cur.b, axis), comp(cur.c, axis), plane);
                        if ((cls & kClsModeMask) == 0u) {
                            // whole triangle to one side (all-lo/all-hi or one of the planar special cases)
                            if (((cls & kClsSideLo) != 0) == keep_lo) {
                                level += 1u;
                                if (level == 6u) {
                                    accumulate_piece<UV>(cur, area, w, u, v);
                                    active = false;
                                }
                            }
                            else {
                                active = false;  // discarded
                            }
                        }
                        else {
                            const uint32_t n = split_cut<UV>(cur, sec, cls, axis, plane, keep_lo);
                            if (level == 5u) {
                                accumulate_piece<UV>(cur, area, w, u, v);
                                if (n == 2) accumulate_piece<UV>(sec, area, w, u, v);
                                active = false;
                            }
                            else {
                                if (n == 2) {
                                    stack_store<UV>(stack, level, sec);  // slot of level + 1
                                    pending |= 1u << (level + 1u);
                                }
                                level += 1u;
                            }
                        }
                    }
                }
                // A job is finished when nothing of it is in flight.  `not eqExactly(uv.weight, 0.f)` -> insertWeighted
                // (voxelization.cpp:466-468): the hit is appended to the pool and counted in its cell; the ordered
                // combine happens in the resolve kernels.  A finished hit is parked in the lane's result registers and
                // the append section runs only when half the wavefront holds one (or a lane needs its slot again, or
                // the wavefront leaves), not in every iteration.
                const bool finished = has_job && !active && pending == 0;
                const bool fin_hit = finished && w != 0.f;
                if (finished) has_job = false;
                if (__ballot(fin_hit && d_valid)) flush_results();
                if (fin_hit) {
                    d_w = w; d_u = u; d_v = v;
                    d_xy = px | (py << 16);
                    d_zk = pz | (my_k << 16);
                    d_valid = true;
                }
                const bool leaving = !__ballot(active || pending != 0 || !queue_empty);
                const unsigned long long dm = __ballot(d_valid);
                if (dm && ((uint32_t) __popcll(dm) >= 32u || leaving)) flush_results();
                if (leaving) break;
            }
            t_begin = t_end;
        }
    }
    // the unused tail of this wavefront's last chunk holds no hits
    for (uint32_t k = chunk_used + lane; k < kHitChunk; k += 64)
        if (chunk_used != kHitChunk && chunk_base + k < p.cap_hits) pool[chunk_base + k].brick = kHoleBrick;
    __syncthreads();
    if (threadIdx.x == 0 && s_hits) atomicAdd(&c->n_hits, (unsigned long long) s_hits);
}

// ---- K5a: scan + compact + reset ---------------------------------------------------------------------------
// Two steps.  k_scan_flags streams the one-byte-per-brick dirty map (n_bricks bytes, 4 MiB at 1024^3), lists the
// dirty bricks and clears their flags.  k_scan_bricks then reads only those bricks (1 KiB each, one 16-byte load
// per lane), compacts the occupied cells into `occ` through an LDS staging buffer (one global atomic per flush,
// not per cell) and writes zeros back, so the grid and the flag map are clean for the next voxelization.

__global__ __launch_bounds__(kBlock) void k_scan_flags(uint8_t *brick_dirty, Counters *c, uint32_t *dirty_list, Params p)
{
    __shared__ uint32_t s_list[kBlock * 16];
    __shared__ uint32_t s_n, s_base;
    const uint32_t n_groups = (p.n_bricks + 15u) / 16u;  // the flag map is padded to a multiple of 16 bytes
    uint4 *f4 = reinterpret_cast<uint4 *>(brick_dirty);
    for (uint32_t g0 = blockIdx.x * kBlock; g0 < n_groups; g0 += gridDim.x * kBlock) {
        __syncthreads();
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        const uint32_t g = g0 + threadIdx.x;
        if (g < n_groups) {
            const uint4 f = f4[g];
            if (f.x | f.y | f.z | f.w) {
                const uint32_t w[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                for (uint32_t k = 0; k < 16; ++k)
                    if ((w[k >> 2] >> ((k & 3u) * 8u)) & 0xffu) s_list[atomicAdd(&s_n, 1u)] = g * 16u + k;
                f4[g] = make_uint4(0, 0, 0, 0);
            }
        }
        __syncthreads();
        const uint32_t n = s_n;
        if (n) {
            if (threadIdx.x == 0) s_base = atomicAdd(&c->n_dirty, n);
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < n; i += kBlock) dirty_list[s_base + i] = s_list[i];
        }
    }
}

constexpr uint32_t kScanBricksPerWave = 4;                                   // independent 1 KiB loads in flight per wave
constexpr uint32_t kScanBricksPerRound = (kBlock / 64) * kScanBricksPerWave;  // 16 bricks = 4096 cells per block round
constexpr uint32_t kScanFlushAt = 2048;
constexpr uint32_t kScanCap = kScanFlushAt + kScanBricksPerRound * kBrickCells;

constexpr uint32_t kShortList = 8;     // cells with up to this many hits are sorted in registers by k_resolve
constexpr uint32_t kLane16List = 16;   // up to this: 16 lanes per cell, bitonic sort in registers (k_resolve_wave<16>)
constexpr uint32_t kLaneList = 32;     // up to this: 32 lanes per cell (k_resolve_wave<32>)
constexpr uint32_t kWaveList = 64;     // up to this: one wavefront per cell (k_resolve_wave<64>)
constexpr uint32_t kMidList = 256;     // up to this: one wavefront per cell, LDS bitonic sort
constexpr uint32_t kLongList = 2048;   // up to this: one workgroup per cell, LDS bitonic sort
constexpr uint32_t kBigList = 8192;    // up to this: one workgroup per cell, keys + indices in 96 KiB of dynamic LDS;
                                       // beyond: global-memory sort

struct ResolveLists {  // cells k_resolve defers, by hit count class (indices into occ[])
    uint32_t *lane16, *lane, *w64, *mid, *lng, *big, *huge;
    uint32_t cap;
};

// Cells with more than kShortList hits are resolved by the cooperative tiers; k_scan_bricks files them by hit count
// while it builds occ[] (one global atomic per class and flush), so that every resolve tier can start at once.
constexpr uint32_t kResolveClasses = 7;
__device__ __forceinline__ uint32_t resolve_class(uint32_t cnt)
{
    return cnt <= kLane16List ? 0u
           : cnt <= kLaneList ? 1u
           : cnt <= kWaveList ? 2u
           : cnt <= kMidList  ? 3u
           : cnt <= kLongList ? 4u
           : cnt <= kBigList  ? 5u
                              : 6u;
}
__device__ __forceinline__ uint32_t *class_list(const ResolveLists &l, uint32_t k)
{
    return k == 0 ? l.lane16 : k == 1 ? l.lane : k == 2 ? l.w64 : k == 3 ? l.mid : k == 4 ? l.lng : k == 5 ? l.big : l.huge;
}
__device__ __forceinline__ uint32_t *class_counter(Counters *c, uint32_t k)
{
    return k == 0 ? &c->n_lane16 : k == 1 ? &c->n_lane : k == 2 ? &c->n_w64 : k == 3 ? &c->n_mid : k == 4 ? &c->n_long
           : k == 5 ? &c->n_bigl : &c->n_huge;
}

// Writes the staged occupied cells of one workgroup to `occ`, giving every cell the offset of its hits in the sorted
// record array: one reservation of (cells, hits) per flush, offsets by a block-level prefix sum over the counts.
// The offset is also stored in the cell itself, where k_scatter reads it.
__device__ __forceinline__ void scan_flush(uint32_t n, const uint32_t *s_lo, const uint32_t *s_hi, uint32_t *s_cnt,
                                           uint32_t *s_wave, uint32_t *s_base, uint32_t *s_cls /*[7], zero*/,
                                           uint32_t *s_cls_base /*[7]*/, uint32_t *grid, Counters *c, Occ *occ,
                                           const ResolveLists &lists, const Params &p)
{
    // thread t owns the entries [t * per, (t + 1) * per)
    const uint32_t per = (n + kBlock - 1) / kBlock;
    const uint32_t lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += s_cnt[i];
    uint32_t total;
    uint32_t run = block_exscan(sum, s_wave, total);
    __syncthreads();
    if (threadIdx.x == 0) {
        s_base[0] = atomicAdd(&c->n_vox, n);
        s_base[1] = atomicAdd(&c->n_sorted, total);
    }
    __syncthreads();
    const uint32_t base_vox = s_base[0];
    run += s_base[1];
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t cnt = s_cnt[i];
        const bool listed = base_vox + i < p.cap_vox;
        if (listed) occ[base_vox + i] = Occ{s_lo[i], s_hi[i], run, cnt};
        grid[((uint64_t) s_hi[i] << 32) | s_lo[i]] = run;
        if (cnt > kShortList && listed) {
            // rank within its class among this flush's cells; the count is not needed again, the slot keeps the tag
            const uint32_t cls = resolve_class(cnt);
            s_cnt[i] = 0x80000000u | (cls << 24) | atomicAdd(&s_cls[cls], 1u);
        }
        run += cnt;
    }
    __syncthreads();
    if (threadIdx.x < kResolveClasses) {
        const uint32_t n_cls = s_cls[threadIdx.x];
        s_cls[threadIdx.x] = 0;
        if (n_cls) s_cls_base[threadIdx.x] = atomicAdd(class_counter(c, threadIdx.x), n_cls);
    }
    __syncthreads();
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t tag = s_cnt[i];
        if (tag & 0x80000000u) {
            const uint32_t cls = (tag >> 24) & 7u, slot = s_cls_base[cls] + (tag & 0xffffffu);
            if (slot < lists.cap) class_list(lists, cls)[slot] = base_vox + i;
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_scan_bricks(uint32_t *grid, const uint32_t *__restrict__ dirty_list,
                                                        Counters *c, Occ *occ, ResolveLists lists, Params p)
{
    __shared__ uint32_t s_lo[kScanCap], s_hi[kScanCap], s_cnt[kScanCap];
    __shared__ uint32_t s_n, s_base[2], s_wave[kBlock / 64], s_cls[kResolveClasses], s_cls_base[kResolveClasses];
    if (threadIdx.x == 0) s_n = 0;
    if (threadIdx.x < kResolveClasses) s_cls[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t n_dirty = c->n_dirty;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n_rounds = (n_dirty + kScanBricksPerRound - 1) / kScanBricksPerRound;
    for (uint32_t r = blockIdx.x; r < n_rounds; r += gridDim.x) {
        uint32_t brick[kScanBricksPerWave];
        uint4 h[kScanBricksPerWave];
#pragma unroll
        for (uint32_t k = 0; k < kScanBricksPerWave; ++k) {
            const uint32_t item = r * kScanBricksPerRound + wave * kScanBricksPerWave + k;
            brick[k] = item < n_dirty ? dirty_list[item] : 0xffffffffu;
        }
#pragma unroll
        for (uint32_t k = 0; k < kScanBricksPerWave; ++k)
            h[k] = brick[k] != 0xffffffffu ? reinterpret_cast<const uint4 *>(grid + (uint64_t) brick[k] * kBrickCells)[lane]
                                           : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (uint32_t k = 0; k < kScanBricksPerWave; ++k) {
            if (h[k].x | h[k].y | h[k].z | h[k].w) {
                const uint32_t hv[4] = {h[k].x, h[k].y, h[k].z, h[k].w};
#pragma unroll
                for (uint32_t e = 0; e < 4; ++e) {
                    if (hv[e]) {
                        const uint64_t cell = (uint64_t) brick[k] * kBrickCells + lane * 4u + e;
                        const uint32_t slot = atomicAdd(&s_n, 1u);
                        s_lo[slot] = (uint32_t) cell;
                        s_hi[slot] = (uint32_t) (cell >> 32);
                        s_cnt[slot] = hv[e];
                    }
                }
            }
        }
        __syncthreads();
        const uint32_t n = s_n;
        if (n >= kScanFlushAt) {
            scan_flush(n, s_lo, s_hi, s_cnt, s_wave, s_base, s_cls, s_cls_base, grid, c, occ, lists, p);
            __syncthreads();
            if (threadIdx.x == 0) s_n = 0;
        }
        __syncthreads();
    }
    const uint32_t n = s_n;
    if (n) scan_flush(n, s_lo, s_hi, s_cnt, s_wave, s_base, s_cls, s_cls_base, grid, c, occ, lists, p);
}

// ---- K5b: scatter --------------------------------------------------------------------------------------------
// Streams the hit pool once (coalesced 32-byte records, holes skipped) and places every hit at
// offset(cell) + rank, so that each cell's hits are contiguous for the resolve kernels.
__global__ __launch_bounds__(kBlock) void k_scatter(const HitRec *__restrict__ pool, const uint32_t *__restrict__ grid,
                                                    const Counters *c, uint32_t *sorted, uint32_t stride, Params p)
{
    const uint32_t n = c->n_hits_reserved < p.cap_hits ? c->n_hits_reserved : p.cap_hits;
    // XCD-aware work split (speed only): workgroup b runs on XCD b % 8 and the eight L2s are not coherent, so each
    // XCD takes one contiguous eighth of the pool.  Pool order is emission order, i.e. spatially coherent, so the
    // destination lines of one eighth are written (and write-combined) by a single L2 instead of partially by all.
    constexpr uint32_t kXcds = 8;
    const uint32_t xcd = blockIdx.x % kXcds, local_block = blockIdx.x / kXcds, blocks_per_xcd = gridDim.x / kXcds;
    const uint32_t per = ((n + kXcds - 1) / kXcds + kBlock - 1) / kBlock * kBlock;
    const uint32_t lo = xcd * per, hi = lo + per < n ? lo + per : n;
    for (uint32_t i = lo + local_block * kBlock + threadIdx.x; i < hi; i += blocks_per_xcd * kBlock) {
        const HitRec r = pool[i];
        if (r.brick == kHoleBrick) continue;
        const uint64_t cell = (uint64_t) r.brick * kBrickCells + (r.local_rank >> 24);
        const uint32_t pos = grid[cell] + (r.local_rank & (kMaxRank - 1u));
        if (pos < p.cap_hits) {
            if (stride == 4u) reinterpret_cast<uint4 *>(sorted)[pos] = make_uint4(r.keyhi, r.keylo, __float_as_uint(r.w), 0u);
            else reinterpret_cast<SortedRec *>(sorted)[pos] = SortedRec{r.keyhi, r.keylo, r.w, r.u, r.v, 0u};
        }
    }
}

// Zeroes the dirty bricks (whole 1 KiB bricks, one 16-byte store per lane): leaves the dense grid clean for the next
// voxelization.  Runs after k_scatter has read the per-cell offsets.
__global__ __launch_bounds__(kBlock) void k_reset_bricks(uint32_t *grid, const uint32_t *__restrict__ dirty_list,
                                                         const Counters *c)
{
    const uint32_t n_dirty = c->n_dirty;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t item = blockIdx.x * (kBlock / 64) + wave; item < n_dirty; item += gridDim.x * (kBlock / 64))
        reinterpret_cast<uint4 *>(grid + (uint64_t) dirty_list[item] * kBrickCells)[lane] = make_uint4(0, 0, 0, 0);
}

// ---- K3: resolve ---------------------------------------------------------------------------------------------

struct Materials {
    const uint32_t *types;   // nullable: all MATERIALLESS
    const float *colors;     // nullable
    const int32_t *texids;   // nullable: all 0
    const DevTexture *textures;
    uint32_t n_textures;
};

// colorAt_f, triangle.hpp:181-194 (+ texture get, triangle.hpp:161-166; getPixel semantics: see DESIGN.md)
__device__ __forceinline__ void color_at(const Materials &m, uint32_t tri, float u, float v, float &r, float &g, float &b)
{
    const uint32_t type = m.types ? m.types[tri] : (uint32_t) kTriMaterialless;
    if (type == kTriMaterialless) {
        r = g = b = 1.f;
    }
    else if (type == kTriUntextured) {
        r = m.colors ? m.colors[(size_t) tri * 3 + 0] : 0.f;
        g = m.colors ? m.colors[(size_t) tri * 3 + 1] : 0.f;
        b = m.colors ? m.colors[(size_t) tri * 3 + 2] : 0.f;
    }
    else if (type == kTriTextured && m.n_textures) {
        uint32_t id = m.texids ? (uint32_t) m.texids[tri] : 0u;
        if (id >= m.n_textures) id = 0;
        const DevTexture tx = m.textures[id];
        float tu = u, tv = 1 - v;
        if (tx.wrap) {
            tu = tu - floor_f(tu);
            tv = tv - floor_f(tv);
        }
        else {
            tu = tu < 0.f ? 0.f : (tu > 1.f ? 1.f : tu);
            tv = tv < 0.f ? 0.f : (tv > 1.f ? 1.f : tv);
        }
        uint32_t px = (uint32_t) (tu * (float) tx.width), py = (uint32_t) (tv * (float) tx.height);
        if (px >= tx.width) px = tx.width - 1;
        if (py >= tx.height) py = tx.height - 1;
        const uint8_t *q = tx.pixels + ((size_t) py * tx.width + px) * tx.channels;
        const uint32_t o = tx.channels == 4 ? 1u : 0u;
        r = (float) q[o] / 255.f;
        g = (float) q[o + 1] / 255.f;
        b = (float) q[o + 2] / 255.f;
    }
    else {
        r = 1.f;
        g = 0.f;
        b = 1.f;
    }
}

// ---- ordered replay of one cell's hits --------------------------------------------------------------------------
// The hits of a cell arrive in arbitrary order; the reference's result is a sequential fold, so they are
// replayed in the reference's order, i.e. ascending in the key (sub-voxel, triangle index, leaf order):
//   leaves of one triangle   -> insertWeighted<BLEND>(uvBuffer, ...)  voxelization.cpp:466-468 (new, existing)
//   triangles, ascending     -> moveUvBufferIntoVoxels                voxelization.cpp:513-526 (new, existing)
//   sub-voxels, ascending    -> documented downscale semantics        voxelization.hpp:82-85
struct CellFold {
    bool have_tri = false, have_sub = false, have_cell = false;
    uint32_t cur_group = 0;
    WUv tri_acc{0, 0, 0};
    WCol sub_acc{0, 0, 0, 0}, cell_acc{0, 0, 0, 0};

    __device__ __forceinline__ void close_tri(const Materials &m, uint32_t blend)
    {
        float cr, cg, cb;
        color_at(m, cur_group & 0x1fffffffu, tri_acc.u, tri_acc.v, cr, cg, cb);
        const WCol fresh{tri_acc.w, cr, cg, cb};
        sub_acc = have_sub ? wcombine(blend, fresh, sub_acc) : fresh;
        have_sub = true;
        have_tri = false;
    }
    __device__ __forceinline__ void close_sub(uint32_t blend)
    {
        cell_acc = have_cell ? wcombine(blend, sub_acc, cell_acc) : sub_acc;
        have_cell = true;
        have_sub = false;
    }
    // hits must be added in ascending key order
    __device__ __forceinline__ void add(const Materials &m, uint32_t blend, uint32_t keyhi, float w, float u, float v)
    {
        if (have_tri && keyhi != cur_group) close_tri(m, blend);
        if (have_sub && (keyhi >> 29) != (cur_group >> 29)) close_sub(blend);
        const WUv hit{w, u, v};
        tri_acc = have_tri ? wmix(hit, tri_acc) : hit;
        have_tri = true;
        cur_group = keyhi;
    }
    __device__ __forceinline__ uint32_t finish(const Materials &m, uint32_t blend)
    {
        if (have_tri) close_tri(m, blend);
        if (have_sub) close_sub(blend);
        return pack_argb(cell_acc.r, cell_acc.g, cell_acc.b);
    }
};

__device__ __forceinline__ uint4 cell_record(const Occ &o, uint32_t argb, const Params &p)
{
    const uint64_t cell = ((uint64_t) o.cell_hi << 32) | o.cell_lo;
    const uint32_t brick = (uint32_t) (cell >> 8), local = (uint32_t) cell & 255u;
    const uint32_t row = brick / p.NBx;
    const uint32_t bx = brick - row * p.NBx;
    const uint32_t bz = row / p.NBy;
    const uint32_t by = row - bz * p.NBy;
    const uint32_t x = bx * kBrickX + (local & 15u), y = by * kBrickY + ((local >> 4) & 3u), z = bz * kBrickZ + (local >> 6);
    return make_uint4(x, y, z + p.zo0, argb);
}


// Tier 1: one lane per occupied cell.  Cells with up to 8 hits (the common case) are insertion-sorted in registers
// from their contiguous records; longer ones are deferred, by hit count, to the cooperative kernels below.
template <uint32_t STRIDE>
__global__ __launch_bounds__(kBlock) void k_resolve(const Occ *__restrict__ occ, SortedView sorted_dyn,
                                                    const Counters *c, Materials m, uint4 *out, Params p)
{
    constexpr bool kUv = STRIDE == 6;  // 16-byte records carry no uv: the columns shrink to 24 KiB, 6 workgroups per CU
    __shared__ uint64_t s_key[kShortList][kBlock];
    __shared__ float s_w[kShortList][kBlock], s_u[kUv ? kShortList : 1][kBlock], s_v[kUv ? kShortList : 1][kBlock];
    const SortedView sorted{sorted_dyn.base, STRIDE};  // compile-time stride: the preloads below stay branch-free
    if (pass_overflowed(c, p)) return;
    const uint32_t n = c->n_vox < p.cap_vox ? c->n_vox : p.cap_vox;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const Occ o = occ[i];
        if (o.count > kShortList) continue;  // filed for a cooperative tier by k_scan_bricks
        // all loads are issued before anything is consumed (independent round trips overlap), then the records are
        // insertion-sorted into this lane's private LDS column and folded by a rolled loop
        SortedRec r[kShortList];
#pragma unroll
        for (uint32_t k = 0; k < kShortList; ++k) r[k] = sorted.load(o.offset + (k < o.count ? k : 0u));
#pragma unroll
        for (uint32_t k = 0; k < kShortList; ++k) {
            if (k < o.count) {
                const uint64_t key = ((uint64_t) r[k].keyhi << 32) | r[k].keylo;
                uint32_t j = k;
                while (j > 0 && s_key[j - 1][threadIdx.x] > key) {
                    s_key[j][threadIdx.x] = s_key[j - 1][threadIdx.x];
                    s_w[j][threadIdx.x] = s_w[j - 1][threadIdx.x];
                    if (kUv) {
                        s_u[j][threadIdx.x] = s_u[j - 1][threadIdx.x];
                        s_v[j][threadIdx.x] = s_v[j - 1][threadIdx.x];
                    }
                    --j;
                }
                s_key[j][threadIdx.x] = key;
                s_w[j][threadIdx.x] = r[k].w;
                if (kUv) {
                    s_u[j][threadIdx.x] = r[k].u;
                    s_v[j][threadIdx.x] = r[k].v;
                }
            }
        }
        CellFold f;
        for (uint32_t t = 0; t < o.count; ++t)
            f.add(m, p.blend, (uint32_t) (s_key[t][threadIdx.x] >> 32), s_w[t][threadIdx.x],
                  kUv ? s_u[t][threadIdx.x] : 0.f, kUv ? s_v[t][threadIdx.x] : 0.f);
        out[i] = cell_record(o, f.finish(m, p.blend), p);
    }
}

// Tier 2: cells with 9..64 hits, W = 16, 32 or 64 lanes per cell (64 / W cells per wavefront).  Every lane loads one
// record; the (key, position) pairs are bitonic-sorted across the W lanes with cross-lane moves only (no LDS, no
// barrier); the payload is gathered to its sorted lane and the cell is folded in order, every lane of the group
// running the same fold on broadcast values.
template <uint32_t W>
__global__ __launch_bounds__(kBlock) void k_resolve_wave(const uint32_t *__restrict__ list, const uint32_t *n_list,
                                                         const Counters *c, const Occ *__restrict__ occ,
                                                         SortedView sorted, Materials m, uint4 *out, uint32_t list_cap,
                                                         Params p)
{
    constexpr uint32_t kPerWave = 64u / W;
    if (pass_overflowed(c, p)) return;
    const uint32_t total = *n_list < list_cap ? *n_list : list_cap;
    const uint32_t lane = threadIdx.x & 63u, sub = lane / W, sl = lane % W, base_lane = sub * W;
    const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, n_waves = gridDim.x * (kBlock / 64u);
    for (uint32_t item0 = wave * kPerWave; item0 < total; item0 += n_waves * kPerWave) {  // wave-uniform
        const uint32_t item = item0 + sub;
        const bool valid = item < total;
        uint32_t i = 0;
        Occ o{};
        if (valid) {
            i = list[item];
            o = occ[i];
        }
        const uint32_t n = valid ? (o.count < W ? o.count : W) : 0u;
        uint64_t key = ~0ull;
        uint32_t hi = 0, idx = sl;
        float w = 0.f, u = 0.f, v = 0.f;
        if (sl < n) {
            const SortedRec r = sorted.load(o.offset + sl);
            key = ((uint64_t) r.keyhi << 32) | r.keylo;
            hi = r.keyhi;
            w = r.w;
            u = r.u;
            v = r.v;
        }
#pragma unroll
        for (uint32_t k = 2; k <= W; k <<= 1) {
#pragma unroll
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                const uint64_t okey = __shfl_xor(key, (int) j, 64);
                const uint32_t oidx = __shfl_xor(idx, (int) j, 64);
                const bool keep_min = ((sl & k) == 0) == ((sl & j) == 0);
                if (keep_min ? okey < key : okey > key) {
                    key = okey;
                    idx = oidx;
                }
            }
        }
        const int src = (int) (base_lane + idx);
        hi = __shfl(hi, src, 64);
        w = __shfl(w, src, 64);
        u = __shfl(u, src, 64);
        v = __shfl(v, src, 64);
        CellFold f;
        for (uint32_t t = 0; t < W; ++t) {
            if (!__any(t < n)) break;
            const int from = (int) (base_lane + t);
            const uint32_t hh = __shfl(hi, from, 64);
            const float ww = __shfl(w, from, 64), uu = __shfl(u, from, 64), vv = __shfl(v, from, 64);
            if (t < n) f.add(m, p.blend, hh, ww, uu, vv);
        }
        if (n != 0 && sl == 0) out[i] = cell_record(o, f.finish(m, p.blend), p);
    }
}

template <typename KeyPtr, typename IdxPtr>
__device__ __forceinline__ void bitonic_sort(KeyPtr key, IdxPtr idx, uint32_t n_pow2, uint32_t tid, uint32_t nthreads)
{
    for (uint32_t k = 2; k <= n_pow2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < n_pow2; t += nthreads) {
                const uint32_t partner = t ^ j;
                if (partner > t) {
                    const bool up = (t & k) == 0;
                    const uint64_t a = key[t], b = key[partner];
                    if ((a > b) == up) {
                        key[t] = b;
                        key[partner] = a;
                        const uint32_t ia = idx[t];
                        idx[t] = idx[partner];
                        idx[partner] = ia;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// Tiers 2 and 3: THREADS lanes cooperate on one cell (a wavefront for up to 256 hits, a workgroup for up to 2048).
// The cell's records are contiguous: keys are loaded coalesced, (key, idx) pairs are bitonic-sorted in LDS, the
// payload is gathered in sorted order, and lane 0 replays the fold (which is inherently sequential: the float
// combine is not associative).
template <uint32_t THREADS, uint32_t CAP>
__global__ __launch_bounds__(THREADS) void k_resolve_sorted(const uint32_t *__restrict__ list, const uint32_t *n_list,
                                                            uint32_t *cursor, const Counters *c,
                                                            const Occ *__restrict__ occ, SortedView sorted, Materials m,
                                                            uint4 *out, uint32_t list_cap, Params p)
{
    if (pass_overflowed(c, p)) return;
    __shared__ uint64_t s_key[CAP];
    __shared__ uint32_t s_idx[CAP];
    __shared__ uint32_t s_hi[CAP];
    __shared__ float s_w[CAP], s_u[CAP], s_v[CAP];
    __shared__ uint32_t s_item;
    const uint32_t total = *n_list < list_cap ? *n_list : list_cap;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_item = atomicAdd(cursor, 1u);
        __syncthreads();
        const uint32_t item = s_item;
        if (item >= total) break;
        const uint32_t i = list[item];
        const Occ o = occ[i];
        const uint32_t n = o.count < CAP ? o.count : CAP;
        uint32_t n_pow2 = 1;
        while (n_pow2 < n) n_pow2 <<= 1;
        for (uint32_t t = threadIdx.x; t < n_pow2; t += THREADS) {
            if (t < n) {
                const SortedRec r = sorted.load(o.offset + t);
                s_key[t] = ((uint64_t) r.keyhi << 32) | r.keylo;
                s_idx[t] = t;
            }
            else {
                s_key[t] = ~0ull;
                s_idx[t] = 0;
            }
        }
        __syncthreads();
        bitonic_sort(s_key, s_idx, n_pow2, threadIdx.x, THREADS);
        for (uint32_t t = threadIdx.x; t < n; t += THREADS) {
            const SortedRec r = sorted.load(o.offset + s_idx[t]);
            s_hi[t] = r.keyhi;
            s_w[t] = r.w;
            s_u[t] = r.u;
            s_v[t] = r.v;
        }
        __syncthreads();
        if (p.blend) {
            // BLEND: the weighted mean is folded in the reference's order (float mix is not associative), but only the
            // chain over the triangles is sequential: every triangle's own hits (its leaves in this cell) and its colour
            // lookup are independent of the other triangles, so the lane at a group's first record folds the group and
            // leaves {weight, r, g, b} there; lane 0 then combines the groups in order (CellFold's close_tri /
            // close_sub sequence without the loads).
            for (uint32_t t = threadIdx.x; t < n; t += THREADS) {
                if (t == 0 || s_hi[t] != s_hi[t - 1]) {
                    WUv acc{s_w[t], s_u[t], s_v[t]};
                    for (uint32_t j = t + 1; j < n && s_hi[j] == s_hi[t]; ++j) acc = wmix(WUv{s_w[j], s_u[j], s_v[j]}, acc);
                    float cr, cg, cb;
                    color_at(m, s_hi[t] & 0x1fffffffu, acc.u, acc.v, cr, cg, cb);
                    s_w[t] = acc.w;
                    s_u[t] = cr;
                    s_v[t] = cg;
                    s_idx[t] = __float_as_uint(cb);  // the sort indices are no longer needed
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                bool have_sub = false, have_cell = false;
                WCol sub_acc{0, 0, 0, 0}, cell_acc{0, 0, 0, 0};
                uint32_t cur_sub = 0;
                for (uint32_t t = 0; t < n; ++t) {
                    const uint32_t hi = s_hi[t];
                    if (t != 0 && hi == s_hi[t - 1]) continue;
                    if (have_sub && (hi >> 29) != cur_sub) {
                        cell_acc = have_cell ? wcombine(p.blend, sub_acc, cell_acc) : sub_acc;
                        have_cell = true;
                        have_sub = false;
                    }
                    const WCol fresh{s_w[t], s_u[t], s_v[t], __uint_as_float(s_idx[t])};
                    sub_acc = have_sub ? wcombine(p.blend, fresh, sub_acc) : fresh;
                    have_sub = true;
                    cur_sub = hi >> 29;
                }
                if (have_sub) cell_acc = have_cell ? wcombine(p.blend, sub_acc, cell_acc) : sub_acc;
                out[i] = cell_record(o, pack_argb(cell_acc.r, cell_acc.g, cell_acc.b), p);
            }
        }
        else {
            // MAX: `new.w > existing.w ? new : existing` over ascending (sub-voxel, triangle) groups keeps the first
            // group with the greatest weight, which is a true reduction: every group is folded by the lane at its
            // first record (leaves of one triangle, in order), then the groups are max-reduced with ties to the
            // lower position.
            unsigned long long best = 0;
            for (uint32_t t = threadIdx.x; t < n; t += THREADS) {
                if (t == 0 || s_hi[t] != s_hi[t - 1]) {
                    WUv acc{s_w[t], s_u[t], s_v[t]};
                    uint32_t j = t + 1;
                    for (; j < n && s_hi[j] == s_hi[t]; ++j) acc = wmix(WUv{s_w[j], s_u[j], s_v[j]}, acc);
                    // weights are non-negative, so their bit patterns order like the values
                    const unsigned long long cand = ((unsigned long long) __float_as_uint(acc.w) << 32) | (0xffffffffu - t);
                    best = cand > best ? cand : best;
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const unsigned long long other = __shfl_xor(best, d, 64);
                best = other > best ? other : best;
            }
            if (THREADS > 64) {
                __syncthreads();
                if ((threadIdx.x & 63u) == 0) s_key[threadIdx.x >> 6] = best;  // s_key is free after the sort
                __syncthreads();
                best = s_key[0];
                for (uint32_t wv = 1; wv < THREADS / 64; ++wv) best = s_key[wv] > best ? s_key[wv] : best;
            }
            if (threadIdx.x == 0) {
                const uint32_t t = 0xffffffffu - (uint32_t) best;
                // rebuild the winning group's uv (needed for a textured winner) and emit
                WUv acc{s_w[t], s_u[t], s_v[t]};
                for (uint32_t j = t + 1; j < n && s_hi[j] == s_hi[t]; ++j) acc = wmix(WUv{s_w[j], s_u[j], s_v[j]}, acc);
                float cr, cg, cb;
                color_at(m, s_hi[t] & 0x1fffffffu, acc.u, acc.v, cr, cg, cb);
                out[i] = cell_record(o, pack_argb(cr, cg, cb), p);
            }
        }
    }
}

// Tier 3b: cells with 2049..8192 hits (the poles of a finely tessellated sphere at high resolution).  One workgroup
// per cell; (key, idx) pairs are bitonic-sorted in dynamic LDS (96 KiB), the payload stays in global memory: MAX
// folds the groups in parallel straight from it, BLEND stages it in sorted order, 1024 records at a time, for the
// sequential replay.
constexpr uint32_t kBigStage = 1024, kBigThreads = 1024;
__global__ __launch_bounds__(kBigThreads) void k_resolve_big(const uint32_t *__restrict__ list, Counters *c,
                                                        const Occ *__restrict__ occ, SortedView sorted, Materials m,
                                                        uint4 *out, uint32_t list_cap, Params p)
{
    extern __shared__ __align__(16) unsigned char s_dyn[];
    uint64_t *s_key = reinterpret_cast<uint64_t *>(s_dyn);                                  // [kBigList]
    uint32_t *s_idx = reinterpret_cast<uint32_t *>(s_dyn + (size_t) kBigList * 8);           // [kBigList]
    __shared__ uint32_t s_hi[kBigStage];
    __shared__ float s_w[kBigStage], s_u[kBigStage], s_v[kBigStage];
    __shared__ unsigned long long s_best[kBigThreads / 64];
    __shared__ uint32_t s_item;
    if (pass_overflowed(c, p)) return;
    const uint32_t total = c->n_bigl < list_cap ? c->n_bigl : list_cap;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_item = atomicAdd(&c->cursor_big, 1u);
        __syncthreads();
        const uint32_t item = s_item;
        if (item >= total) break;
        const uint32_t i = list[item];
        const Occ o = occ[i];
        const uint32_t n = o.count < kBigList ? o.count : kBigList;
        uint32_t n_pow2 = 1;
        while (n_pow2 < n) n_pow2 <<= 1;
        for (uint32_t t = threadIdx.x; t < n_pow2; t += kBigThreads) {
            if (t < n) {
                const SortedRec r = sorted.load(o.offset + t);
                s_key[t] = ((uint64_t) r.keyhi << 32) | r.keylo;
                s_idx[t] = t;
            }
            else {
                s_key[t] = ~0ull;
                s_idx[t] = 0;
            }
        }
        __syncthreads();
        bitonic_sort(s_key, s_idx, n_pow2, threadIdx.x, kBigThreads);
        if (p.blend) {
            CellFold f;  // only thread 0's copy is used
            for (uint32_t base = 0; base < n; base += kBigStage) {
                const uint32_t m_here = n - base < kBigStage ? n - base : kBigStage;
                __syncthreads();
                for (uint32_t t = threadIdx.x; t < m_here; t += kBigThreads) {
                    const SortedRec r = sorted.load(o.offset + s_idx[base + t]);
                    s_hi[t] = r.keyhi;
                    s_w[t] = r.w;
                    s_u[t] = r.u;
                    s_v[t] = r.v;
                }
                __syncthreads();
                if (threadIdx.x == 0)
                    for (uint32_t t = 0; t < m_here; ++t) f.add(m, p.blend, s_hi[t], s_w[t], s_u[t], s_v[t]);
            }
            if (threadIdx.x == 0) out[i] = cell_record(o, f.finish(m, p.blend), p);
        }
        else {
            unsigned long long best = 0;
            for (uint32_t t = threadIdx.x; t < n; t += kBigThreads) {
                const uint32_t hi = (uint32_t) (s_key[t] >> 32);
                if (t == 0 || (uint32_t) (s_key[t - 1] >> 32) != hi) {
                    SortedRec r = sorted.load(o.offset + s_idx[t]);
                    WUv acc{r.w, r.u, r.v};
                    for (uint32_t j = t + 1; j < n && (uint32_t) (s_key[j] >> 32) == hi; ++j) {
                        r = sorted.load(o.offset + s_idx[j]);
                        acc = wmix(WUv{r.w, r.u, r.v}, acc);
                    }
                    const unsigned long long cand = ((unsigned long long) __float_as_uint(acc.w) << 32) | (0xffffffffu - t);
                    best = cand > best ? cand : best;
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const unsigned long long other = __shfl_xor(best, d, 64);
                best = other > best ? other : best;
            }
            __syncthreads();
            if ((threadIdx.x & 63u) == 0) s_best[threadIdx.x >> 6] = best;
            __syncthreads();
            if (threadIdx.x == 0) {
                for (uint32_t wv = 1; wv < kBigThreads / 64; ++wv) best = s_best[wv] > best ? s_best[wv] : best;
                const uint32_t t = 0xffffffffu - (uint32_t) best;
                const uint32_t hi = (uint32_t) (s_key[t] >> 32);
                SortedRec r = sorted.load(o.offset + s_idx[t]);
                WUv acc{r.w, r.u, r.v};
                for (uint32_t j = t + 1; j < n && (uint32_t) (s_key[j] >> 32) == hi; ++j) {
                    r = sorted.load(o.offset + s_idx[j]);
                    acc = wmix(WUv{r.w, r.u, r.v}, acc);
                }
                float cr, cg, cb;
                color_at(m, hi & 0x1fffffffu, acc.u, acc.v, cr, cg, cb);
                out[i] = cell_record(o, pack_argb(cr, cg, cb), p);
            }
        }
    }
}

// Tier 4: cells with more than 8192 hits (a whole mesh inside a few voxels).  Same algorithm with the (key, idx)
// pairs in a global scratch area; each cell bump-allocates a power-of-two range (scratch holds 2 * cap_hits pairs).
__global__ __launch_bounds__(kBlock) void k_resolve_huge(const uint32_t *__restrict__ list, Counters *c,
                                                         const Occ *__restrict__ occ, SortedView sorted,
                                                         Materials m, uint4 *out, uint64_t *scratch_key,
                                                         uint32_t *scratch_idx, uint32_t scratch_cap, uint32_t list_cap,
                                                         Params p)
{
    __shared__ uint32_t s_item, s_base, s_ok;
    __shared__ unsigned long long s_best[kBlock / 64];
    if (pass_overflowed(c, p)) return;
    const uint32_t total = c->n_huge < list_cap ? c->n_huge : list_cap;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_item = atomicAdd(&c->cursor_huge, 1u);
        __syncthreads();
        const uint32_t item = s_item;
        if (item >= total) break;
        const uint32_t i = list[item];
        const Occ o = occ[i];
        const uint32_t n = o.count;
        uint32_t n_pow2 = 1;
        while (n_pow2 < n) n_pow2 <<= 1;
        if (threadIdx.x == 0) {
            s_base = atomicAdd(&c->scratch_used, n_pow2);
            // scratch too small: the host sees scratch_used > capacity, grows it and re-runs
            s_ok = (uint64_t) s_base + n_pow2 <= scratch_cap ? 1u : 0u;
        }
        __syncthreads();
        if (!s_ok) continue;
        uint64_t *key = scratch_key + s_base;
        uint32_t *idx = scratch_idx + s_base;
        for (uint32_t t = threadIdx.x; t < n_pow2; t += kBlock) {
            if (t < n) {
                const SortedRec r = sorted.load(o.offset + t);
                key[t] = ((uint64_t) r.keyhi << 32) | r.keylo;
                idx[t] = t;
            }
            else {
                key[t] = ~0ull;
                idx[t] = 0;
            }
        }
        __syncthreads();
        bitonic_sort(key, idx, n_pow2, threadIdx.x, kBlock);
        if (p.blend) {
            // BLEND: sequential by nature (see k_resolve_sorted)
            if (threadIdx.x == 0) {
                CellFold f;
                for (uint32_t t = 0; t < n; ++t) {
                    const SortedRec r = sorted.load(o.offset + idx[t]);
                    f.add(m, p.blend, r.keyhi, r.w, r.u, r.v);
                }
                out[i] = cell_record(o, f.finish(m, p.blend), p);
            }
        }
        else {
            // MAX: fold every (sub-voxel, triangle) group at its first record, max-reduce with ties to the earlier group
            unsigned long long best = 0;
            for (uint32_t t = threadIdx.x; t < n; t += kBlock) {
                const uint32_t hi = (uint32_t) (key[t] >> 32);
                if (t == 0 || (uint32_t) (key[t - 1] >> 32) != hi) {
                    SortedRec r = sorted.load(o.offset + idx[t]);
                    WUv acc{r.w, r.u, r.v};
                    for (uint32_t j = t + 1; j < n && (uint32_t) (key[j] >> 32) == hi; ++j) {
                        r = sorted.load(o.offset + idx[j]);
                        acc = wmix(WUv{r.w, r.u, r.v}, acc);
                    }
                    const unsigned long long cand = ((unsigned long long) __float_as_uint(acc.w) << 32) | (0xffffffffu - t);
                    best = cand > best ? cand : best;
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const unsigned long long other = __shfl_xor(best, d, 64);
                best = other > best ? other : best;
            }
            __syncthreads();
            if ((threadIdx.x & 63u) == 0) s_best[threadIdx.x >> 6] = best;
            __syncthreads();
            if (threadIdx.x == 0) {
                for (uint32_t wv = 1; wv < kBlock / 64; ++wv) best = s_best[wv] > best ? s_best[wv] : best;
                const uint32_t t = 0xffffffffu - (uint32_t) best;
                const uint32_t hi = (uint32_t) (key[t] >> 32);
                SortedRec r = sorted.load(o.offset + idx[t]);
                WUv acc{r.w, r.u, r.v};
                for (uint32_t j = t + 1; j < n && (uint32_t) (key[j] >> 32) == hi; ++j) {
                    r = sorted.load(o.offset + idx[j]);
                    acc = wmix(WUv{r.w, r.u, r.v}, acc);
                }
                float cr, cg, cb;
                color_at(m, hi & 0x1fffffffu, acc.u, acc.v, cr, cg, cb);
                out[i] = cell_record(o, pack_argb(cr, cg, cb), p);
            }
        }
    }
}

}  // namespace

// ---- host side: context, buffers, launch sequence ------------------------------------------------------------

struct o2v_hip_ctx {
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    hipEvent_t ev[6] = {};
    std::string err;

    // inputs
    float *d_verts = nullptr, *d_uvs = nullptr, *d_colors = nullptr;
    uint32_t *d_types = nullptr;
    int32_t *d_texids = nullptr;
    uint64_t n_tris = 0;
    bool any_textured = false;
    DevTexture *d_textures = nullptr;
    std::vector<uint8_t *> d_texpix;
    uint32_t n_textures = 0;

    // work buffers (grown on demand)
    Counters *d_ctr = nullptr;
    Counters *h_ctr = nullptr;  // pinned
    hipStream_t aux[3] = {nullptr, nullptr, nullptr};  // the cooperative resolve tiers run beside tier 1
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    unsigned long long *d_zhist = nullptr, *h_zhist = nullptr;  // kPlanBins each (h_: pinned), o2v_hip_plan_slabs
    float2 *d_zrange = nullptr;      // z extent per 256 triangles, written by the slab plan
    float *d_zrange_xform = nullptr;  // the transform they were computed with (12 floats)
    uint32_t cap_zrange = 0;
    uint64_t tri_generation = 0, zrange_generation = ~0ull;  // the extents belong to the triangles of that upload
    Leaf *d_leaves = nullptr;
    Tile *d_tiles = nullptr;
    BigLeaf *d_big = nullptr;
    Node *d_nodes[2] = {nullptr, nullptr};
    HitRec *d_pool = nullptr;
    SortedRec *d_sorted = nullptr;  // cap_hits records (read through SortedView: 24 or 16 bytes per record)
    uint32_t sorted_stride = 6;
    Occ *d_occ = nullptr;
    uint4 *d_out = nullptr;
    uint32_t *d_list_lane16 = nullptr, *d_list_w64 = nullptr, *d_list_lane = nullptr, *d_list_mid = nullptr, *d_list_long = nullptr, *d_list_big = nullptr,
             *d_list_huge = nullptr;  // cap_vox each
    uint64_t *d_scratch_key = nullptr;  // tier-4 resolve scratch, allocated on first need
    uint32_t *d_scratch_idx = nullptr;
    uint32_t cap_scratch = 0;
    uint32_t cap_leaves = 0, cap_tiles = 0, cap_big = 0, cap_nodes = 0, cap_hits = 0, cap_vox = 0;

    // dense grid of list heads for this context's slab
    uint32_t *d_grid = nullptr;
    uint64_t grid_cells = 0;      // allocated
    uint8_t *d_brick_dirty = nullptr;   // one flag per brick (padded to 16 bytes)
    uint32_t *d_dirty_list = nullptr;   // dirty brick ids of the current run
    uint64_t brick_cap = 0;
    bool grid_dirty = false;

    // results of the last run
    uint64_t n_vox = 0;
    o2v_hip_timings timings = {};
    o2v_hip_stats stats = {};
    float xform[12] = {};
};

namespace {

#define O2V_CHECK(expr)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            ctx->err = std::string(#expr) + ": " + hipGetErrorString(e_);                                 \
            return e_ == hipErrorOutOfMemory ? O2V_HIP_ERR_OUT_OF_MEMORY : O2V_HIP_ERR_HIP;               \
        }                                                                                                 \
    } while (0)

template <typename T>
int grow(o2v_hip_ctx *ctx, T *&ptr, uint32_t &cap, uint64_t want)
{
    if (want <= cap && ptr) return O2V_HIP_OK;
    if (want > 0xfffffff0ull) {
        ctx->err = "device buffer would exceed 2^32 records";
        return O2V_HIP_ERR_LIMIT;
    }
    if (ptr) O2V_CHECK(hipFree(ptr));
    ptr = nullptr;
    cap = 0;
    O2V_CHECK(hipMalloc(reinterpret_cast<void **>(&ptr), want * sizeof(T)));
    cap = (uint32_t) want;
    return O2V_HIP_OK;
}

template <typename T>
int upload(o2v_hip_ctx *ctx, T *&dptr, const T *host, uint64_t count)
{
    if (dptr) O2V_CHECK(hipFree(dptr));
    dptr = nullptr;
    if (!host || !count) return O2V_HIP_OK;
    O2V_CHECK(hipMalloc(reinterpret_cast<void **>(&dptr), count * sizeof(T)));
    O2V_CHECK(hipMemcpy(dptr, host, count * sizeof(T), hipMemcpyHostToDevice));
    return O2V_HIP_OK;
}


// O2V_DEBUG_SYNC=1: synchronise and log after every launch (locates a faulting or hanging kernel); the resolve tiers
// then run on one stream.  O2V_DEBUG_SYNC=2: the same, but the tiers keep their own streams (device-wide sync).
int debug_sync_level()
{
    static const int level = [] {
        const char *e = std::getenv("O2V_DEBUG_SYNC");
        return e && (e[0] == '1' || e[0] == '2') ? e[0] - '0' : 0;
    }();
    return level;
}
bool debug_sync_enabled() { return debug_sync_level() != 0; }
#define O2V_STAGE(name)                                                          \
    do {                                                                         \
        if (debug_sync_enabled()) {                                              \
            std::fprintf(stderr, "[o2v] launched %s ...", name);                 \
            std::fflush(stderr);                                                 \
            hipError_t e_ = debug_sync_level() == 2 ? hipDeviceSynchronize() : hipStreamSynchronize(s); \
            std::fprintf(stderr, " %s\n", hipGetErrorString(e_));                \
        }                                                                        \
    } while (0)

float ord2f_host(uint32_t o)
{
    const uint32_t b = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}

// One pass of the pipeline with the current capacities.  Fills h_ctr; the caller checks for overflow.
int run_pass(o2v_hip_ctx *ctx, const Params &p, bool use_uv, uint32_t n_rounds)
{
    hipStream_t s = ctx->stream;
    const uint32_t persistent = (uint32_t) ctx->num_cus * 8u;
    O2V_CHECK(hipEventRecord(ctx->ev[0], s));
    hipLaunchKernelGGL(k_init, dim3(1), dim3(64), 0, s, ctx->d_ctr);
    O2V_STAGE("k_init");
    if (!p.bounds_known) {
        hipLaunchKernelGGL(k_bounds, dim3((uint32_t) std::min<uint64_t>((uint64_t) ctx->num_cus * 4u, (p.n_tris * 9 / 12 + kBlock) / kBlock)),
                           dim3(kBlock), 0, s, ctx->d_verts, p.n_tris * 9, ctx->d_ctr);
        O2V_STAGE("k_bounds");
    }
    hipLaunchKernelGGL(k_setup, dim3(1), dim3(64), 0, s, ctx->d_ctr, p);
    O2V_STAGE("k_setup");
    O2V_CHECK(hipEventRecord(ctx->ev[1], s));

    hipLaunchKernelGGL(k_expand_roots, dim3(std::min<uint64_t>(persistent, (p.n_tris + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, s, ctx->d_verts, ctx->d_uvs, ctx->d_ctr, ctx->d_leaves, ctx->d_tiles,
                       ctx->d_big, ctx->d_nodes[0], ctx->zrange_generation == ctx->tri_generation ? ctx->d_zrange : nullptr,
                       ctx->d_zrange_xform, p);
    O2V_STAGE("k_expand_roots");
    for (uint32_t round = 0; round < n_rounds; ++round) {
        // most rounds are empty or small: a narrow grid keeps an empty launch short (the kernel strides over its input)
        hipLaunchKernelGGL(k_expand_nodes, dim3((uint32_t) ctx->num_cus * 2u), dim3(kBlock), 0, s, ctx->d_nodes[round & 1], round,
                           ctx->d_ctr, ctx->d_leaves, ctx->d_tiles, ctx->d_big, ctx->d_nodes[(round + 1) & 1], p);
        O2V_STAGE("k_expand_nodes");
    }
    hipLaunchKernelGGL(k_expand_big, dim3((uint32_t) ctx->num_cus * 2u), dim3(kBlock), 0, s, ctx->d_big, ctx->d_ctr, ctx->d_tiles, p);
    O2V_STAGE("k_expand_big");
    O2V_CHECK(hipEventRecord(ctx->ev[2], s));

    {
        // persistent workgroups; residency is VGPR-bound (about 4 waves per SIMD without UVs, 3 with)
        const uint32_t blocks = (uint32_t) ctx->num_cus * (use_uv ? 3u : 4u);
        if (use_uv) {
            hipLaunchKernelGGL(k_voxelize<true>, dim3(blocks), dim3(kBlock), 0, s, ctx->d_leaves, ctx->d_tiles,
                               ctx->d_ctr, ctx->d_grid, ctx->d_brick_dirty, ctx->d_pool, p);
        }
        else {
            hipLaunchKernelGGL(k_voxelize<false>, dim3(blocks), dim3(kBlock), 0, s, ctx->d_leaves, ctx->d_tiles,
                               ctx->d_ctr, ctx->d_grid, ctx->d_brick_dirty, ctx->d_pool, p);
        }
        O2V_STAGE("k_voxelize");
    }
    O2V_CHECK(hipEventRecord(ctx->ev[3], s));

    {
        const uint32_t flag_groups = (p.n_bricks + 15u) / 16u;
        hipLaunchKernelGGL(k_scan_flags, dim3(std::min<uint32_t>((uint32_t) ctx->num_cus * 4u, (flag_groups + kBlock - 1) / kBlock)),
                           dim3(kBlock), 0, s, ctx->d_brick_dirty, ctx->d_ctr, ctx->d_dirty_list, p);
        O2V_STAGE("k_scan_flags");
        const ResolveLists lists{ctx->d_list_lane16, ctx->d_list_lane, ctx->d_list_w64, ctx->d_list_mid, ctx->d_list_long,
                                 ctx->d_list_big, ctx->d_list_huge, p.cap_vox};
        hipLaunchKernelGGL(k_scan_bricks, dim3((uint32_t) ctx->num_cus * 2u), dim3(kBlock), 0, s, ctx->d_grid,
                           ctx->d_dirty_list, ctx->d_ctr, ctx->d_occ, lists, p);
        O2V_STAGE("k_scan_bricks");
        hipLaunchKernelGGL(k_scatter, dim3(persistent), dim3(kBlock), 0, s, ctx->d_pool, ctx->d_grid, ctx->d_ctr,
                           reinterpret_cast<uint32_t *>(ctx->d_sorted), use_uv ? 6u : 4u, p);
        O2V_STAGE("k_scatter");
        hipLaunchKernelGGL(k_reset_bricks, dim3((uint32_t) ctx->num_cus * 4u), dim3(kBlock), 0, s, ctx->d_grid,
                           ctx->d_dirty_list, ctx->d_ctr);
        O2V_STAGE("k_reset_bricks");
    }
    O2V_CHECK(hipEventRecord(ctx->ev[4], s));

    {
        Materials m{ctx->d_types, ctx->d_colors, ctx->d_texids, ctx->d_textures, ctx->n_textures};
        const SortedView sorted_view{reinterpret_cast<const uint32_t *>(ctx->d_sorted), use_uv ? 6u : 4u};
        // The tiers work on disjoint cells and were filed by k_scan_bricks, so they run side by side: tier 1 on the
        // main stream, the cooperative tiers (short, latency-bound launches) on three auxiliary streams.
        const bool fork = debug_sync_level() != 1;
        hipStream_t sw = s, sm = s, sl = s;
        if (fork) {
            sw = ctx->aux[0];
            sm = ctx->aux[1];
            sl = ctx->aux[2];
            O2V_CHECK(hipEventRecord(ctx->ev_fork, s));
            for (hipStream_t a : ctx->aux) O2V_CHECK(hipStreamWaitEvent(a, ctx->ev_fork, 0));
        }
        if (use_uv)
            hipLaunchKernelGGL(k_resolve<6>, dim3(persistent), dim3(kBlock), 0, s, ctx->d_occ, sorted_view, ctx->d_ctr, m,
                               ctx->d_out, p);
        else
            hipLaunchKernelGGL(k_resolve<4>, dim3(persistent), dim3(kBlock), 0, s, ctx->d_occ, sorted_view, ctx->d_ctr, m,
                               ctx->d_out, p);
        O2V_STAGE("k_resolve");
        hipLaunchKernelGGL(k_resolve_wave<16>, dim3((uint32_t) ctx->num_cus * 8u), dim3(kBlock), 0, sw, ctx->d_list_lane16,
                           &ctx->d_ctr->n_lane16, ctx->d_ctr, ctx->d_occ, sorted_view, m, ctx->d_out, p.cap_vox, p);
        O2V_STAGE("k_resolve_wave<16>");
        hipLaunchKernelGGL(k_resolve_wave<32>, dim3((uint32_t) ctx->num_cus * 8u), dim3(kBlock), 0, sw, ctx->d_list_lane,
                           &ctx->d_ctr->n_lane, ctx->d_ctr, ctx->d_occ, sorted_view, m, ctx->d_out, p.cap_vox, p);
        O2V_STAGE("k_resolve_wave<32>");
        hipLaunchKernelGGL(k_resolve_wave<64>, dim3((uint32_t) ctx->num_cus * 8u), dim3(kBlock), 0, sm, ctx->d_list_w64,
                           &ctx->d_ctr->n_w64, ctx->d_ctr, ctx->d_occ, sorted_view, m, ctx->d_out, p.cap_vox, p);
        O2V_STAGE("k_resolve_wave<64>");
        hipLaunchKernelGGL((k_resolve_sorted<64, kMidList>), dim3((uint32_t) ctx->num_cus * 16u), dim3(64), 0, sm,
                           ctx->d_list_mid, &ctx->d_ctr->n_mid, &ctx->d_ctr->cursor_mid, ctx->d_ctr, ctx->d_occ, sorted_view, m,
                           ctx->d_out, p.cap_vox, p);
        O2V_STAGE("k_resolve_sorted");
        hipLaunchKernelGGL((k_resolve_sorted<kBlock, kLongList>), dim3((uint32_t) ctx->num_cus * 2u), dim3(kBlock), 0, sl,
                           ctx->d_list_long, &ctx->d_ctr->n_long, &ctx->d_ctr->cursor_long, ctx->d_ctr, ctx->d_occ, sorted_view, m,
                           ctx->d_out, p.cap_vox, p);
        O2V_STAGE("k_resolve_sorted");
        hipLaunchKernelGGL(k_resolve_big, dim3((uint32_t) ctx->num_cus), dim3(kBigThreads), kBigList * 12u, sl, ctx->d_list_big,
                           ctx->d_ctr, ctx->d_occ, sorted_view, m, ctx->d_out, p.cap_vox, p);
        O2V_STAGE("k_resolve_big");
        if (ctx->d_scratch_key) {
            hipLaunchKernelGGL(k_resolve_huge, dim3((uint32_t) ctx->num_cus), dim3(kBlock), 0, sl, ctx->d_list_huge,
                               ctx->d_ctr, ctx->d_occ, sorted_view, m, ctx->d_out, ctx->d_scratch_key,
                               ctx->d_scratch_idx, ctx->cap_scratch, p.cap_vox, p);
            O2V_STAGE("k_resolve_huge");
        }
        if (fork)
            for (int j = 0; j < 3; ++j) {
                O2V_CHECK(hipEventRecord(ctx->ev_join[j], ctx->aux[j]));
                O2V_CHECK(hipStreamWaitEvent(s, ctx->ev_join[j], 0));
            }
    }
    O2V_CHECK(hipEventRecord(ctx->ev[5], s));
    O2V_CHECK(hipMemcpyAsync(ctx->h_ctr, ctx->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, s));
    O2V_CHECK(hipStreamSynchronize(s));
    O2V_CHECK(hipGetLastError());
    return O2V_HIP_OK;
}

}  // namespace

extern "C" {

int o2v_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int o2v_hip_create(int device, o2v_hip_ctx **out_ctx)
{
    if (!out_ctx) return O2V_HIP_ERR_BAD_ARGUMENT;
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return O2V_HIP_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return O2V_HIP_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return O2V_HIP_ERR_NO_DEVICE;
    o2v_hip_ctx *ctx = new o2v_hip_ctx;
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return O2V_HIP_ERR_HIP;
    }
    for (auto &e : ctx->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            delete ctx;
            return O2V_HIP_ERR_HIP;
        }
    bool ok = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int j = 0; j < 3 && ok; ++j)
        ok = hipStreamCreateWithFlags(&ctx->aux[j], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&ctx->ev_join[j], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        delete ctx;
        return O2V_HIP_ERR_HIP;
    }
    if (hipMalloc(reinterpret_cast<void **>(&ctx->d_ctr), sizeof(Counters)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&ctx->h_ctr), sizeof(Counters), hipHostMallocDefault) != hipSuccess) {
        delete ctx;
        return O2V_HIP_ERR_OUT_OF_MEMORY;
    }
    // k_resolve_big sorts in 96 KiB of dynamic LDS (above the default 64 KiB limit)
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(&k_resolve_big), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int) (kBigList * 12u));
    *out_ctx = ctx;
    return O2V_HIP_OK;
}

void o2v_hip_destroy(o2v_hip_ctx *ctx)
{
    if (!ctx) return;
    (void) hipSetDevice(ctx->device);
    if (ctx->stream) (void) hipStreamSynchronize(ctx->stream);
    void *ptrs[] = {ctx->d_verts, ctx->d_uvs,  ctx->d_colors,   ctx->d_types,    ctx->d_texids, ctx->d_textures,
                    ctx->d_ctr,   ctx->d_leaves, ctx->d_tiles,  ctx->d_big,      ctx->d_nodes[0], ctx->d_nodes[1],
                    ctx->d_pool,  ctx->d_sorted, ctx->d_occ,  ctx->d_out,      ctx->d_grid,
                    ctx->d_list_lane16, ctx->d_list_w64, ctx->d_list_lane, ctx->d_list_mid, ctx->d_list_long, ctx->d_list_big, ctx->d_list_huge, ctx->d_scratch_key, ctx->d_scratch_idx,
                    ctx->d_brick_dirty, ctx->d_dirty_list};
    for (void *q : ptrs)
        if (q) (void) hipFree(q);
    for (uint8_t *q : ctx->d_texpix)
        if (q) (void) hipFree(q);
    if (ctx->h_ctr) (void) hipHostFree(ctx->h_ctr);
    if (ctx->d_zhist) (void) hipFree(ctx->d_zhist);
    if (ctx->d_zrange) (void) hipFree(ctx->d_zrange);
    if (ctx->d_zrange_xform) (void) hipFree(ctx->d_zrange_xform);
    if (ctx->h_zhist) (void) hipHostFree(ctx->h_zhist);
    for (auto &e : ctx->ev)
        if (e) (void) hipEventDestroy(e);
    if (ctx->ev_fork) (void) hipEventDestroy(ctx->ev_fork);
    for (int j = 0; j < 3; ++j) {
        if (ctx->ev_join[j]) (void) hipEventDestroy(ctx->ev_join[j]);
        if (ctx->aux[j]) (void) hipStreamDestroy(ctx->aux[j]);
    }
    if (ctx->stream) (void) hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *o2v_hip_last_error(const o2v_hip_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int o2v_hip_set_triangles(o2v_hip_ctx *ctx, const float *verts, const float *uvs, const uint32_t *types,
                          const float *colors, const int32_t *texids, uint64_t count)
{
    if (!ctx || (count && !verts)) return O2V_HIP_ERR_BAD_ARGUMENT;
    if (count >= (1ull << 29)) {
        ctx->err = "triangle count must be below 2^29";
        return O2V_HIP_ERR_LIMIT;
    }
    O2V_CHECK(hipSetDevice(ctx->device));
    int rc;
    if ((rc = upload(ctx, ctx->d_verts, verts, count * 9))) return rc;
    if ((rc = upload(ctx, ctx->d_uvs, uvs, count * 6))) return rc;
    if ((rc = upload(ctx, ctx->d_types, types, count))) return rc;
    if ((rc = upload(ctx, ctx->d_colors, colors, count * 3))) return rc;
    if ((rc = upload(ctx, ctx->d_texids, texids, count))) return rc;
    ctx->n_tris = count;
    ctx->tri_generation += 1;
    ctx->any_textured = false;
    if (types)
        for (uint64_t i = 0; i < count; ++i)
            if (types[i] == O2V_HIP_TRI_TEXTURED) {
                ctx->any_textured = true;
                break;
            }
    return O2V_HIP_OK;
}

int o2v_hip_set_textures(o2v_hip_ctx *ctx, const o2v_hip_texture *textures, uint32_t count)
{
    if (!ctx || (count && !textures)) return O2V_HIP_ERR_BAD_ARGUMENT;
    O2V_CHECK(hipSetDevice(ctx->device));
    for (uint8_t *q : ctx->d_texpix)
        if (q) O2V_CHECK(hipFree(q));
    ctx->d_texpix.clear();
    if (ctx->d_textures) O2V_CHECK(hipFree(ctx->d_textures));
    ctx->d_textures = nullptr;
    ctx->n_textures = 0;
    if (!count) return O2V_HIP_OK;
    std::vector<DevTexture> host(count);
    for (uint32_t i = 0; i < count; ++i) {
        const o2v_hip_texture &t = textures[i];
        if (!t.pixels || !t.width || !t.height || (t.channels != 3 && t.channels != 4)) {
            ctx->err = "texture must have pixels, a non-zero size and 3 or 4 channels";
            return O2V_HIP_ERR_BAD_ARGUMENT;
        }
        const size_t bytes = (size_t) t.width * t.height * t.channels;
        uint8_t *d = nullptr;
        O2V_CHECK(hipMalloc(reinterpret_cast<void **>(&d), bytes));
        ctx->d_texpix.push_back(d);
        O2V_CHECK(hipMemcpy(d, t.pixels, bytes, hipMemcpyHostToDevice));
        host[i] = DevTexture{d, t.width, t.height, t.channels, t.wrap};
    }
    O2V_CHECK(hipMalloc(reinterpret_cast<void **>(&ctx->d_textures), count * sizeof(DevTexture)));
    O2V_CHECK(hipMemcpy(ctx->d_textures, host.data(), count * sizeof(DevTexture), hipMemcpyHostToDevice));
    ctx->n_textures = count;
    return O2V_HIP_OK;
}

int o2v_hip_voxelize(o2v_hip_ctx *ctx, const o2v_hip_params *params, uint64_t *out_voxel_count)
{
    if (!ctx || !params) return O2V_HIP_ERR_BAD_ARGUMENT;
    if (out_voxel_count) *out_voxel_count = 0;
    const uint32_t ss = params->supersampling ? params->supersampling : 1u;
    if (params->resolution == 0 || ss > 2 || params->strategy > 1) {
        ctx->err = "resolution must be non-zero, supersampling 1 or 2, strategy 0 or 1";
        return O2V_HIP_ERR_BAD_ARGUMENT;
    }
    const uint64_t S64 = (uint64_t) params->resolution * ss;
    if (S64 > 65535u) {
        ctx->err = "sample resolution must be below 65536";
        return O2V_HIP_ERR_LIMIT;
    }
    uint32_t z0 = params->z_begin, z1 = params->z_end;
    if (z0 == 0 && z1 == 0) z1 = params->resolution;
    if (z1 > params->resolution || z0 >= z1) {
        ctx->err = "z slab must satisfy z_begin < z_end <= resolution";
        return O2V_HIP_ERR_BAD_ARGUMENT;
    }
    O2V_CHECK(hipSetDevice(ctx->device));
    ctx->n_vox = 0;
    ctx->timings = {};
    ctx->stats = {};
    ctx->stats.triangles = ctx->n_tris;

    Params p{};
    p.n_tris = ctx->n_tris;
    p.S = (uint32_t) S64;
    p.G = params->resolution;
    p.NBx = (p.G + kBrickX - 1) / kBrickX;
    p.NBy = (p.G + kBrickY - 1) / kBrickY;
    const uint32_t NBz = (z1 - z0 + kBrickZ - 1) / kBrickZ;
    const uint64_t n_bricks = (uint64_t) p.NBx * p.NBy * NBz;
    if (n_bricks >= (1ull << 32) / 2) {
        ctx->err = "slab has too many bricks for 32-bit brick ids; use more z-slabs";
        return O2V_HIP_ERR_LIMIT;
    }
    p.n_bricks = (uint32_t) n_bricks;
    p.ss_shift = ss == 2 ? 1u : 0u;
    p.zs0 = z0 * ss;
    p.zs1 = z1 * ss;
    p.zo0 = z0;
    p.blend = params->strategy;
    p.bounds_known = params->bounds_known;
    for (int i = 0; i < 6; ++i) p.bounds[i] = params->bounds[i];
    for (int i = 0; i < 9; ++i) p.unit[i] = params->unit_transform[i];
    p.has_uv = ctx->d_uvs ? 1u : 0u;
    const bool use_uv = ctx->d_uvs && ctx->any_textured;
    ctx->sorted_stride = use_uv ? 6u : 4u;

    // dense grid for the slab (bricked, see cell_index) + one dirty flag per brick; allocated zeroed, kept clean
    // by k_scan_flags / k_scan_bricks
    const uint64_t cells = n_bricks * kBrickCells;
    ctx->stats.grid_cells = cells;
    ctx->stats.grid_bytes = cells * sizeof(uint32_t) + n_bricks;
    if (cells > ctx->grid_cells || !ctx->d_grid) {
        for (void *q : {(void *) ctx->d_grid, (void *) ctx->d_brick_dirty, (void *) ctx->d_dirty_list})
            if (q) O2V_CHECK(hipFree(q));
        ctx->d_grid = nullptr;
        ctx->d_brick_dirty = nullptr;
        ctx->d_dirty_list = nullptr;
        ctx->grid_cells = 0;
        O2V_CHECK(hipMalloc(reinterpret_cast<void **>(&ctx->d_grid), cells * sizeof(uint32_t)));
        ctx->grid_cells = cells;
        ctx->brick_cap = (n_bricks + 15u) & ~15ull;
        O2V_CHECK(hipMalloc(reinterpret_cast<void **>(&ctx->d_brick_dirty), ctx->brick_cap));
        O2V_CHECK(hipMalloc(reinterpret_cast<void **>(&ctx->d_dirty_list), ctx->brick_cap * sizeof(uint32_t)));
        ctx->grid_dirty = true;
    }
    if (ctx->grid_dirty) {
        O2V_CHECK(hipMemsetAsync(ctx->d_grid, 0, ctx->grid_cells * sizeof(uint32_t), ctx->stream));
        O2V_CHECK(hipMemsetAsync(ctx->d_brick_dirty, 0, ctx->brick_cap, ctx->stream));
        ctx->grid_dirty = false;
    }
    if (ctx->n_tris == 0) return O2V_HIP_OK;  // empty mesh: empty model (obj2voxel.cpp:590-594)

    // initial capacities; every counter keeps counting past its capacity so one re-run sizes it exactly
    uint64_t want_leaves = std::max<uint64_t>(ctx->cap_leaves, ctx->n_tris + ctx->n_tris / 4 + (1u << 16));
    uint64_t want_tiles = std::max<uint64_t>(ctx->cap_tiles, ctx->n_tris + ctx->n_tris / 2 + (1u << 16));
    uint64_t want_big = std::max<uint64_t>(ctx->cap_big, 1u << 16);
    uint64_t want_nodes = std::max<uint64_t>(ctx->cap_nodes, 1u << 18);
    uint64_t want_hits = std::max<uint64_t>(ctx->cap_hits, std::min<uint64_t>(16 * ctx->n_tris + (4u << 20), 1ull << 31));
    if (const char *tiny = std::getenv("O2V_TEST_TINY_BUFFERS"); tiny && tiny[0] == '1') {
        // test hook: start with minimal buffers so that every overflow -> grow -> re-run path is exercised
        want_leaves = std::max<uint64_t>(ctx->cap_leaves, 64);
        want_tiles = std::max<uint64_t>(ctx->cap_tiles, 64);
        want_big = std::max<uint64_t>(ctx->cap_big, 4);
        want_nodes = std::max<uint64_t>(ctx->cap_nodes, 16);
        want_hits = std::max<uint64_t>(ctx->cap_hits, 512);
    }
    uint64_t want_scratch = ctx->cap_scratch;
    uint64_t want_vox = std::max<uint64_t>(ctx->cap_vox, std::min<uint64_t>(8 * ctx->n_tris + (2u << 20), 1ull << 31));
    if (const char *tiny = std::getenv("O2V_TEST_TINY_BUFFERS"); tiny && tiny[0] == '1')
        want_vox = std::max<uint64_t>(ctx->cap_vox, 256);

    // Subdivision rounds to launch: every round halves a node's extents and a node becomes a leaf once its voxel
    // AABB volume is below 512, so ceil(log2(S)) rounds cover the usual case; if a node is still waiting after the
    // last round the pass is repeated with the full kMaxRounds (nothing is lost, only re-run).
    uint32_t n_rounds = 4;
    while ((1u << n_rounds) < p.S && n_rounds < kMaxRounds) ++n_rounds;
    ctx->grid_dirty = true;  // until a pass completes (the scan / reset kernels leave it clean)
    for (uint32_t pass = 1; pass <= 12; ++pass) {
        int rc;
        if (pass > 1) {
            // a pass that overflowed a buffer may have left counters / offsets in cells it could not list
            O2V_CHECK(hipMemsetAsync(ctx->d_grid, 0, ctx->grid_cells * sizeof(uint32_t), ctx->stream));
            O2V_CHECK(hipMemsetAsync(ctx->d_brick_dirty, 0, ctx->brick_cap, ctx->stream));
        }
        if ((rc = grow(ctx, ctx->d_leaves, ctx->cap_leaves, want_leaves))) return rc;
        if ((rc = grow(ctx, ctx->d_tiles, ctx->cap_tiles, want_tiles))) return rc;
        if ((rc = grow(ctx, ctx->d_big, ctx->cap_big, want_big))) return rc;
        uint32_t cap_n0 = ctx->cap_nodes, cap_n1 = ctx->cap_nodes;
        if ((rc = grow(ctx, ctx->d_nodes[0], cap_n0, want_nodes))) return rc;
        if ((rc = grow(ctx, ctx->d_nodes[1], cap_n1, want_nodes))) return rc;
        ctx->cap_nodes = cap_n0;
        {
            uint32_t cap_p = ctx->cap_hits, cap_s = ctx->cap_hits;
            if ((rc = grow(ctx, ctx->d_pool, cap_p, want_hits))) return rc;
            if ((rc = grow(ctx, ctx->d_sorted, cap_s, want_hits))) return rc;
            ctx->cap_hits = cap_p;
        }
        uint32_t cap_v0 = ctx->cap_vox, cap_v1 = ctx->cap_vox;
        if ((rc = grow(ctx, ctx->d_occ, cap_v0, want_vox))) return rc;
        if ((rc = grow(ctx, ctx->d_out, cap_v1, want_vox))) return rc;
        for (uint32_t **lp : {&ctx->d_list_lane16, &ctx->d_list_w64, &ctx->d_list_lane, &ctx->d_list_mid, &ctx->d_list_long, &ctx->d_list_big, &ctx->d_list_huge}) {
            uint32_t cap_l = ctx->cap_vox;
            if ((rc = grow(ctx, *lp, cap_l, want_vox))) return rc;
        }
        ctx->cap_vox = cap_v0;
        if (want_scratch) {
            uint32_t cap_s0 = ctx->cap_scratch, cap_s1 = ctx->cap_scratch;
            if ((rc = grow(ctx, ctx->d_scratch_key, cap_s0, want_scratch))) return rc;
            if ((rc = grow(ctx, ctx->d_scratch_idx, cap_s1, want_scratch))) return rc;
            ctx->cap_scratch = cap_s0;
        }
        p.cap_leaves = ctx->cap_leaves;
        p.cap_tiles = ctx->cap_tiles;
        p.cap_big = ctx->cap_big;
        p.cap_nodes = ctx->cap_nodes;
        p.cap_hits = ctx->cap_hits;
        p.cap_vox = ctx->cap_vox;

        if ((rc = run_pass(ctx, p, use_uv, n_rounds))) return rc;
        const Counters &h = *ctx->h_ctr;
        ctx->timings.passes = pass;
        if (h.err_flags) {
            ctx->grid_dirty = false;
            ctx->err = (h.err_flags & kErrLeafTooLarge) ? "a leaf's voxel AABB has 2^32 or more candidate voxels"
                       : (h.err_flags & kErrDepth)      ? "subdivision deeper than 15 levels"
                                                        : "a voxel received 2^24 or more hits";
            return O2V_HIP_ERR_LIMIT;
        }
        uint32_t max_nodes = 0;
        for (uint32_t r = 0; r <= kMaxRounds; ++r) max_nodes = std::max(max_nodes, h.n_nodes[r]);
        bool again = false;
        auto need = [&](uint64_t used, uint32_t cap, uint64_t &want) {
            if (used > cap) {
                want = used + used / 4 + 1024;
                again = true;
            }
        };
        need(h.n_leaves, ctx->cap_leaves, want_leaves);
        need(h.n_tiles, ctx->cap_tiles, want_tiles);
        need(h.n_big, ctx->cap_big, want_big);
        need(max_nodes, ctx->cap_nodes, want_nodes);
        need(h.n_hits_reserved, ctx->cap_hits, want_hits);
        need(h.n_vox, ctx->cap_vox, want_vox);
        if (n_rounds < kMaxRounds && h.n_nodes[n_rounds] != 0) {
            n_rounds = kMaxRounds;  // unusually deep subdivision
            again = true;
        }
        if (!again && h.n_huge && (!ctx->d_scratch_key || h.scratch_used > ctx->cap_scratch)) {
            // some cell holds more than kLongList hits: the global-memory sort tier needs its scratch area
            want_scratch = std::max<uint64_t>(2ull * ctx->cap_hits, (uint64_t) h.scratch_used + 1024);
            again = true;
        }
        if (!again) {
            ctx->grid_dirty = false;
            ctx->n_vox = h.n_vox;
            ctx->stats.leaves = h.n_leaves;
            ctx->stats.tiles = h.n_tiles;
            ctx->stats.candidates = h.n_candidates;
            ctx->stats.hits = h.n_hits;
            ctx->stats.voxels = h.n_vox;
            ctx->stats.bricks = p.n_bricks;
            ctx->stats.dirty_bricks = h.n_dirty;
            ctx->stats.pool_slots = h.n_hits_reserved;
            std::memcpy(ctx->xform, h.xform, sizeof(ctx->xform));
            float ms[5];
            for (int i = 0; i < 5; ++i) O2V_CHECK(hipEventElapsedTime(&ms[i], ctx->ev[i], ctx->ev[i + 1]));
            ctx->timings.bounds_ms = ms[0];
            ctx->timings.expand_ms = ms[1];
            ctx->timings.voxelize_ms = ms[2];
            ctx->timings.scan_ms = ms[3];
            ctx->timings.resolve_ms = ms[4];
            O2V_CHECK(hipEventElapsedTime(&ctx->timings.total_ms, ctx->ev[0], ctx->ev[5]));
            if (out_voxel_count) *out_voxel_count = ctx->n_vox;
            return O2V_HIP_OK;
        }
    }
    ctx->err = "device buffers did not converge after 12 passes";
    return O2V_HIP_ERR_LIMIT;
}

int o2v_hip_plan_slabs(o2v_hip_ctx *ctx, const o2v_hip_params *params, uint32_t n_slabs, uint32_t *out_z,
                       float *out_bounds)
{
    if (!ctx || !params || !out_z || n_slabs == 0) return O2V_HIP_ERR_BAD_ARGUMENT;
    const uint32_t ss = params->supersampling ? params->supersampling : 1u;
    const uint32_t G = params->resolution;
    if (G == 0 || ss > 2 || (uint64_t) G * ss > 65535u || n_slabs > G) {
        ctx->err = "plan_slabs: resolution must be non-zero and below 65536 samples, 1 <= n_slabs <= resolution";
        return O2V_HIP_ERR_BAD_ARGUMENT;
    }
    for (uint32_t k = 0; k <= n_slabs; ++k) out_z[k] = (uint32_t) ((uint64_t) G * k / n_slabs);  // equal heights
    if (out_bounds)
        for (int i = 0; i < 6; ++i) out_bounds[i] = params->bounds_known ? params->bounds[i] : 0.f;
    if (ctx->n_tris == 0) return O2V_HIP_OK;
    O2V_CHECK(hipSetDevice(ctx->device));
    if (!ctx->d_zhist) {
        O2V_CHECK(hipMalloc(reinterpret_cast<void **>(&ctx->d_zhist), kPlanBins * sizeof(unsigned long long)));
        O2V_CHECK(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_zhist), kPlanBins * sizeof(unsigned long long),
                                hipHostMallocDefault));
    }
    Params p{};
    p.n_tris = ctx->n_tris;
    p.S = G * ss;
    p.G = G;
    p.bounds_known = params->bounds_known;
    for (int i = 0; i < 6; ++i) p.bounds[i] = params->bounds[i];
    for (int i = 0; i < 9; ++i) p.unit[i] = params->unit_transform[i];
    // sample layers per bin: a whole number of output layers, at most kPlanBins bins
    const uint32_t bin_out = (G + kPlanBins - 1) / kPlanBins;
    const uint32_t n_bins = (G + bin_out - 1) / bin_out;

    hipStream_t s = ctx->stream;
    hipLaunchKernelGGL(k_init, dim3(1), dim3(64), 0, s, ctx->d_ctr);
    if (!p.bounds_known)
        hipLaunchKernelGGL(k_bounds, dim3((uint32_t) std::min<uint64_t>((uint64_t) ctx->num_cus * 4u, (p.n_tris * 9 / 12 + kBlock) / kBlock)),
                           dim3(kBlock), 0, s, ctx->d_verts, p.n_tris * 9, ctx->d_ctr);
    hipLaunchKernelGGL(k_setup, dim3(1), dim3(64), 0, s, ctx->d_ctr, p);
    O2V_CHECK(hipMemsetAsync(ctx->d_zhist, 0, kPlanBins * sizeof(unsigned long long), s));
    {
        int rc;
        if ((rc = grow(ctx, ctx->d_zrange, ctx->cap_zrange, (p.n_tris + kBlock - 1) / kBlock))) return rc;
        if (!ctx->d_zrange_xform) O2V_CHECK(hipMalloc(reinterpret_cast<void **>(&ctx->d_zrange_xform), 12 * sizeof(float)));
    }
    ctx->zrange_generation = ~0ull;
    hipLaunchKernelGGL(k_zhist, dim3((uint32_t) std::min<uint64_t>((uint64_t) ctx->num_cus * 6u, (p.n_tris + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, s, ctx->d_verts, ctx->d_ctr, ctx->d_zhist, ctx->d_zrange, ctx->d_zrange_xform, p,
                       bin_out * ss);
    O2V_STAGE("k_zhist");
    O2V_CHECK(hipMemcpyAsync(ctx->h_zhist, ctx->d_zhist, n_bins * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    O2V_CHECK(hipMemcpyAsync(ctx->h_ctr, ctx->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, s));
    O2V_CHECK(hipStreamSynchronize(s));
    O2V_CHECK(hipGetLastError());
    ctx->zrange_generation = ctx->tri_generation;  // k_expand_roots may use the extents (it checks the transform)
    if (out_bounds && !params->bounds_known)
        for (int i = 0; i < 6; ++i) out_bounds[i] = ord2f_host(ctx->h_ctr->bounds_enc[i]);

    unsigned __int128 total = 0;
    for (uint32_t b = 0; b < n_bins; ++b) total += ctx->h_zhist[b];
    if (total == 0) return O2V_HIP_OK;
    unsigned __int128 before = 0;
    uint32_t k = 1;
    for (uint32_t b = 0; b < n_bins && k < n_slabs; ++b) {
        const unsigned __int128 after = before + ctx->h_zhist[b];
        while (k < n_slabs && after * n_slabs >= total * k) {
            // the k-th cut falls inside bin b: take whichever end of the bin is closer to the target
            const unsigned __int128 target_n = total * k;  // compare in units of 1/n_slabs
            const bool take_start = (target_n - before * n_slabs) < (after * n_slabs - target_n);
            out_z[k] = std::min<uint32_t>(G, (take_start ? b : b + 1) * bin_out);
            ++k;
        }
        before = after;
    }
    for (; k < n_slabs; ++k) out_z[k] = G;
    // every slab keeps at least one layer
    for (uint32_t j = 1; j < n_slabs; ++j) out_z[j] = std::max(out_z[j], out_z[j - 1] + 1);
    for (uint32_t j = n_slabs - 1; j >= 1; --j) out_z[j] = std::min(out_z[j], out_z[j + 1] - 1);
    return O2V_HIP_OK;
}

int o2v_hip_read_voxels(o2v_hip_ctx *ctx, uint32_t *out, uint64_t first, uint64_t count)
{
    if (!ctx || (!out && count)) return O2V_HIP_ERR_BAD_ARGUMENT;
    if (first + count > ctx->n_vox) {
        ctx->err = "voxel range out of bounds";
        return O2V_HIP_ERR_BAD_ARGUMENT;
    }
    if (!count) return O2V_HIP_OK;
    O2V_CHECK(hipSetDevice(ctx->device));
    O2V_CHECK(hipMemcpy(out, ctx->d_out + first, count * sizeof(uint4), hipMemcpyDeviceToHost));
    return O2V_HIP_OK;
}

int o2v_hip_voxels_device_ptr(o2v_hip_ctx *ctx, const uint32_t **out_ptr, uint64_t *out_count)
{
    if (!ctx || !out_ptr || !out_count) return O2V_HIP_ERR_BAD_ARGUMENT;
    *out_ptr = reinterpret_cast<const uint32_t *>(ctx->d_out);
    *out_count = ctx->n_vox;
    return O2V_HIP_OK;
}

// Debugging aid: the hit records of one output cell of the last run (the occupied-cell list and the hit pool
// stay valid after a run).  Each record is 6 words: keyhi, keylo, w, u, v (as float bits) and the pool index.
int o2v_hip_debug_cell_hits(o2v_hip_ctx *ctx, uint32_t x, uint32_t y, uint32_t z, uint32_t *out, uint32_t max_records,
                            uint32_t *out_count)
{
    if (!ctx || !out || !out_count) return O2V_HIP_ERR_BAD_ARGUMENT;
    *out_count = 0;
    O2V_CHECK(hipSetDevice(ctx->device));
    std::vector<Occ> occ(ctx->n_vox);
    std::vector<uint4> vox(ctx->n_vox);
    if (!ctx->n_vox) return O2V_HIP_OK;
    O2V_CHECK(hipMemcpy(occ.data(), ctx->d_occ, occ.size() * sizeof(Occ), hipMemcpyDeviceToHost));
    O2V_CHECK(hipMemcpy(vox.data(), ctx->d_out, vox.size() * sizeof(uint4), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < ctx->n_vox; ++i) {
        if (vox[i].x != x || vox[i].y != y || vox[i].z != z) continue;
        const uint32_t n = occ[i].count < max_records ? occ[i].count : max_records;
        std::vector<uint32_t> raw((size_t) n * ctx->sorted_stride);
        if (n)
            O2V_CHECK(hipMemcpy(raw.data(), reinterpret_cast<const uint32_t *>(ctx->d_sorted) + (size_t) occ[i].offset * ctx->sorted_stride,
                                raw.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t *r = &raw[(size_t) k * ctx->sorted_stride];
            uint32_t *o = out + k * 6;
            o[0] = r[0];
            o[1] = r[1];
            o[2] = r[2];
            o[3] = ctx->sorted_stride == 6 ? r[3] : 0u;
            o[4] = ctx->sorted_stride == 6 ? r[4] : 0u;
            o[5] = occ[i].offset + k;
        }
        *out_count = n;
        break;
    }
    return O2V_HIP_OK;
}

// Debugging aid: histogram of hits per occupied cell of the last run; bucket b counts cells with 2^(b-1) < hits <= 2^b
// (bucket 0: exactly one hit), 32 buckets.
int o2v_hip_debug_hits_histogram(o2v_hip_ctx *ctx, uint64_t *out32)
{
    if (!ctx || !out32) return O2V_HIP_ERR_BAD_ARGUMENT;
    for (int i = 0; i < 32; ++i) out32[i] = 0;
    if (!ctx->n_vox) return O2V_HIP_OK;
    O2V_CHECK(hipSetDevice(ctx->device));
    std::vector<Occ> occ(ctx->n_vox);
    O2V_CHECK(hipMemcpy(occ.data(), ctx->d_occ, occ.size() * sizeof(Occ), hipMemcpyDeviceToHost));
    for (const Occ &o : occ) {
        uint32_t b = 0;
        while ((1u << b) < o.count && b < 31) ++b;
        out32[b]++;
    }
    return O2V_HIP_OK;
}

int o2v_hip_get_timings(const o2v_hip_ctx *ctx, o2v_hip_timings *out)
{
    if (!ctx || !out) return O2V_HIP_ERR_BAD_ARGUMENT;
    *out = ctx->timings;
    return O2V_HIP_OK;
}

int o2v_hip_get_stats(const o2v_hip_ctx *ctx, o2v_hip_stats *out)
{
    if (!ctx || !out) return O2V_HIP_ERR_BAD_ARGUMENT;
    *out = ctx->stats;
    return O2V_HIP_OK;
}

int o2v_hip_get_transform(const o2v_hip_ctx *ctx, float out12[12])
{
    if (!ctx || !out12) return O2V_HIP_ERR_BAD_ARGUMENT;
    std::memcpy(out12, ctx->xform, sizeof(ctx->xform));
    return O2V_HIP_OK;
}

}  // extern "C"
