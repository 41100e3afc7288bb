// o2v_device.hip -- the MI355X (gfx950) voxelization pipeline behind include/o2v_hip.h.
//
// Replaces, for one GPU's z-slab of the grid, the reference's chunk loop (src/obj2voxel.cpp:467-520) and
// Voxelizer::voxelize (src/voxelization.cpp:480-526).  Written for CDNA4: 64-wide wavefronts, LDS-staged leaf
// geometry and work queues, register-resident clip stacks, 32- and 64-bit atomics on dense (bricked) HBM grids of per-cell
// counters / maxima.  No MFMA: the path is float32 VALU + HBM/atomic traffic.
//
// Stages (one HIP stream; the replay tiers fork onto three auxiliary streams; kernels in o2v_dev_k*.hpp):
//   K0  k_bounds / k_setup     mesh bounds (obj2voxel.cpp:180-200) and mesh transform (obj2voxel.cpp:370-402)
//   K1  k_expand_roots         transform (obj2voxel.cpp:202-224), alignment test (voxelization.cpp:335-347),
//       k_expand_nodes         exact LIFO subdivision (voxelization.cpp:349-379) done breadth-first with an
//                              order key that reproduces the reference's processing order,
//       k_expand_big           tiles of <= 256 candidate voxels for large leaves
//   K2  k_voxelize<UV>         AABB walk + plane cull (voxelization.cpp:426-472) + six-plane clip by triangle
//                              splitting (voxelization.cpp:175-331,383-424).  A hit either goes straight into the
//                              64-bit max grid (MAX strategy, unsplit triangle: one atomicMax) or is
//                              appended to the hit pool and counted in its cell (atomicAdd on the dense grid -> rank)
//   K5  k_scan_flags/_bricks   reads the dirty bricks of the dense grid, compacts occupied cells, turns the
//                              per-cell counts into offsets (counting sort); k_scatter places the pooled hits
//   K3  k_resolve + tiers      per occupied cell: orders the hits like the reference's sequential loops
//                              (sub-voxel, triangle index, leaf order), replays insertWeighted
//                              (voxelization.cpp:56-63,466-468) and moveUvBufferIntoVoxels (:513-526) with
//                              MAX / BLEND, then packs (x, y, z, argb) (obj2voxel.cpp:279-297);
//       k_pick / k_emit_max    direct MAX path: winner colours of textured meshes; the 64-bit max grid -> records
//   plan k_zhist               o2v_hip_plan_slabs: predicted hits per z layer -> work-balanced slabs for N GPUs
// With the direct MAX path K1's counters reach the host while K2 runs, and only the stages that have work are enqueued
// behind it.  N > 1 GPUs: o2v_hip_voxelize_sharded (bounds / work-histogram passes sharded over the ranks, RCCL).
//
// Compile with -ffp-contract=off (see o2v_math.h).
#include "o2v_math.h"

#include "../../include/o2v_hip.h"
#include "o2v_comm.hpp"
#include "o2v_device_internal.hpp"

#ifndef O2V_BUILD_ID
#define O2V_BUILD_ID "unknown"  // the Makefile passes the hash of the device sources
#endif

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace o2v;

namespace {

#include "o2v_dev_common.hpp"
#include "o2v_dev_arith.hpp"
#include "o2v_dev_k0_bounds_plan.hpp"
#include "o2v_dev_k1_expand.hpp"
#include "o2v_dev_k2_voxelize.hpp"
#include "o2v_dev_k5_scan_scatter.hpp"
#include "o2v_dev_k3_resolve.hpp"

}  // namespace

// ---- host side: context, buffers, launch sequence ------------------------------------------------------------

struct o2v_hip_ctx {
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    hipEvent_t ev[6] = {};
    hipEvent_t ev_coll[2] = {};                 // sharded planning: around the collectives
    unsigned long long *d_counts = nullptr, *h_counts = nullptr;  // per-rank voxel counts (all-gathered), world entries
    uint32_t *d_status = nullptr, *h_status = nullptr;            // sharded runs: "this rank is ready" word, max-reduced over the ranks
    uint32_t cap_counts = 0;
    std::string err;

    // inputs
    float *d_verts = nullptr, *d_uvs = nullptr, *d_colors = nullptr;
    uint32_t *d_types = nullptr;
    int32_t *d_texids = nullptr;
    uint64_t n_tris = 0;
    uint64_t cap_tri_bytes[5] = {0, 0, 0, 0, 0};  // allocated bytes of d_verts, d_uvs, d_types, d_colors, d_texids
    // streamed upload (o2v_hip_begin / commit / end_triangles): two page-locked staging blocks, filled in turn
    o2v_hip_staging stage[2] = {};
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
    int stage_cur = 0;
    uint64_t stream_count = 0;
    uint32_t stream_arrays = 0;
    std::vector<void *> retired;  // device arrays replaced by larger ones while a streamed upload was in flight
    bool any_textured = false;
    DevTexture *d_textures = nullptr;
    std::vector<uint8_t *> d_texpix;
    uint32_t n_textures = 0;

    // work buffers (grown on demand)
    Counters *d_ctr = nullptr;
    Counters *h_ctr = nullptr;  // pinned
    bool stage_events = false;  // this call records an event between the stages of a pass (O2V_HIP_FLAG_STAGE_TIMES)
    uint64_t no_pool_key = 0;   // (key + 1 of) the mesh and settings whose last pass pooled no hits (run_pass: k_mark_bricks left out)
    bool marked_bricks = false, mark_missing = false; // the current pass listed its bricks before k_voxelize
    bool skip_big = false;      // no leaf of the uploaded mesh can have more than four tiles (its largest triangle's extent): k_expand_big left out
    bool poisoned = false;      // a collective of a sharded run is stuck on the stream (time limit passed): o2v_hip_destroy must not wait for it
    bool ctr_clean = false;     // d_ctr was zeroed (k_init) behind the last pass and nothing has touched it since
    hipStream_t aux[3] = {nullptr, nullptr, nullptr};  // the cooperative resolve tiers run beside tier 1
    hipEvent_t ev_fork = nullptr, ev_sorted = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    unsigned long long *d_zhist = nullptr, *h_zhist = nullptr;  // kPlanBins each (h_: pinned), o2v_hip_plan_slabs
    float2 *d_zrange = nullptr;      // z extent per 256 triangles, written by the slab plan
    unsigned long long *d_plan_gather = nullptr;  // sharded runs: one record per rank (k_pack_plan), all-gathered
    uint32_t cap_plan_gather = 0;                 // ... in 8-byte words
    float *d_zrange_xform = nullptr;  // the transform they were computed with (12 floats)
    uint32_t cap_zrange = 0;
    uint32_t *d_block_list = nullptr, *d_block_count = nullptr;  // the blocks of 256 triangles that meet the slab (k_list_blocks)
    uint32_t *d_need_list = nullptr;  // k_count_roots: the blocks k_expand_roots still has to walk
    uint32_t cap_need_list = 0;
    bool lean_roots = false;          // this call: k_count_roots ahead of k_expand_roots (o2v_hip_voxelize)
    bool solo_roots = false;          // this call: no k_expand_roots at all, k_voxelize_occ counts the root leaves itself (o2v_hip_voxelize)
    uint64_t solo_refused_key = 0;    // the mesh and settings for which a solo pass found a triangle that needs k_expand_roots
    uint32_t cap_block_list = 0;
    float mesh_bounds_hint[6] = {0, 0, 0, 0, 0, 0};  // bounds and largest triangle extent of the uploaded mesh: only used to
    float max_tri_extent = -1.f;                     // bound the number of subdivision rounds (-1: unknown)
    uint32_t ext_hist[256] = {};                     // k_tri_extent: the uploaded mesh's triangles by the binary exponent of their extent
                                                     // (grid_modes: is the 64-bit max grid worth its memory?)
    uint64_t tri_generation = 0, zrange_generation = ~0ull;  // the extents belong to the triangles of that upload
    Leaf *d_leaves = nullptr;
    Tile *d_tiles = nullptr;
    BigLeaf *d_big = nullptr;
    Node *d_nodes[2] = {nullptr, nullptr};
    uint2 *d_jobq = nullptr;  // k_voxelize's job queues: VoxShape::queue records per workgroup
    HitRec *d_pool = nullptr;
    SortedRec *d_sorted = nullptr;  // cap_hits records (read through SortedView: 24 or 16 bytes per record)
    uint32_t sorted_stride = 6;
    Occ *d_occ = nullptr;
    uint4 *d_out = nullptr;
    uint32_t *d_list_lane8 = nullptr;
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
    uint32_t *d_brick_slab = nullptr;   // per brick: its place in that list = the number of its hit slab (Params::brick_slab)
    uint32_t *d_slabs = nullptr;        // cap_slabs x kInlineHits x 64 hit records (sorted_stride dwords each)
    uint32_t cap_slabs = 0, slabs_stride = 0;
    uint64_t want_slabs_next = 0;       // the brick list of the last pass (+ 1/8): what the slabs are grown to at the next call
    uint64_t slabs_wanted_at_grant = 0; // what was asked for when the slabs were last allocated (they may have got less: the memory was short)
    uint32_t cap_pick_extra = 0;
    PickRec *d_pick_extra = nullptr;  // textured MAX: {cell, key, argb} of the cells resolved by replay (6 words, cap_vox of them)
    unsigned long long *d_maxgrid = nullptr;  // direct MAX path: one 64-bit cell per output voxel (same bricked layout)
    uint8_t *d_dirty_max = nullptr;           // ... its dirty-brick flags and list
    uint32_t *d_dirty_list_max = nullptr;
    uint64_t maxgrid_bytes = 0, maxgrid_brick_cap = 0, maxgrid_map_bytes = 0;
    hipEvent_t ev_k1 = nullptr;               // after K1: its counters decide which stages follow k_voxelize
    bool maxgrid_dirty = false;
    uint64_t brick_cap = 0;
    bool grid_dirty = false;

    // results of the last run
    uint64_t n_vox = 0;
    bool last_ran_general = true;  // the last pass enqueued the counting sort + replay stages
    bool force_general = false;    // ... must do so whatever K1's counters say (set if the shortcut's premise did not hold)
    bool last_direct = false;  // the last run used the 64-bit max grid: occ[] / sorted[] do not describe every voxel
    o2v_hip_timings timings = {};
    o2v_hip_stats stats = {};
    float xform[12] = {};
    uint64_t dbg[16] = {};  // Counters::dbg of the last run (instrumented builds only)

    // O2V_HIP_FLAG_KERNEL_TIMES: an event pair around every launch of a pass
    struct KernelBracket {
        const char *name = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
    };
    std::vector<KernelBracket> ktimes;
    size_t ktimes_used = 0;
    bool ktimes_on = false;
    std::vector<o2v_hip_kernel_time> kernel_times;  // of the last run: one entry per kernel name
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

// (re)allocates a device array only when it has to grow: repeated uploads of similar meshes reuse the allocation
template <typename T>
int ensure_array(o2v_hip_ctx *ctx, T *&dptr, uint64_t &cap_bytes, uint64_t count, bool wanted)
{
    if (!wanted || !count) {
        // an absent optional array must read as null in the kernels (all MATERIALLESS / zero uvs / texture 0)
        if (dptr) O2V_CHECK(hipFree(dptr));
        dptr = nullptr;
        cap_bytes = 0;
        return O2V_HIP_OK;
    }
    const uint64_t bytes = count * sizeof(T);
    if (dptr && bytes <= cap_bytes) return O2V_HIP_OK;
    if (dptr) O2V_CHECK(hipFree(dptr));
    dptr = nullptr;
    cap_bytes = 0;
    O2V_CHECK(hipMalloc(reinterpret_cast<void **>(&dptr), bytes));
    cap_bytes = bytes;
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

// One kernel launch of the pipeline.  With O2V_HIP_FLAG_KERNEL_TIMES the launch is bracketed by two events on the stream
// it goes to (o2v_hip_get_kernel_times; the brackets cost a few microseconds per launch, so bench.py times its steps
// without the flag and collects the per-kernel times in extra steps).
#define O2V_LAUNCH(name, stream, ...)                                            \
    do {                                                                         \
        const int kt_ = ktime_begin(ctx, name, stream);                          \
        hipLaunchKernelGGL(__VA_ARGS__);                                         \
        if (kt_ >= 0) (void) hipEventRecord(ctx->ktimes[(size_t) kt_].e1, stream); \
        O2V_STAGE(name);                                                         \
    } while (0)

// k_voxelize's launch: without the stage events (the default) its duration is still measured, by two events that ride on the
// kernel's own dispatch (hipExtLaunchKernelGGL: the start and end times of that dispatch).  Measured on the bench headline, per
// step: these two 0.004 - 0.006 ms together, an event recorded on the stream between two kernels ~0.004 ms each - the six of
// O2V_HIP_FLAG_STAGE_TIMES 0.02 - 0.025 ms of a 0.5 ms step (profiles/r05/NOTES.md).
#define O2V_LAUNCH_K2(name, kernel, grid, block, ...)                                                               \
    do {                                                                                                            \
        if (ctx->stage_events || ctx->ktimes_on) O2V_LAUNCH(name, s, kernel, grid, block, 0, s, __VA_ARGS__);       \
        else {                                                                                                      \
            hipExtLaunchKernelGGL(kernel, grid, block, 0, s, ctx->ev[2], ctx->ev[3], 0, __VA_ARGS__);               \
            O2V_STAGE(name);                                                                                        \
        }                                                                                                           \
    } while (0)

// An event that is only ever used to read a time: without the system-scope fence a default event performs when it is recorded
// (cache write-back and invalidation in the middle of the pass; nothing on the host reads device memory on its strength).
// Measured: the six stage events of a pass cost 0.009 ms less this way (bench headline, O2V_HIP_FLAG_STAGE_TIMES).
hipError_t create_timing_event(hipEvent_t *e) { return hipEventCreateWithFlags(e, hipEventDisableSystemFence); }

int ktime_begin(o2v_hip_ctx *ctx, const char *name, hipStream_t stream)
{
    if (!ctx->ktimes_on) return -1;
    if (ctx->ktimes_used == ctx->ktimes.size()) {
        o2v_hip_ctx::KernelBracket b{};
        if (create_timing_event(&b.e0) != hipSuccess || create_timing_event(&b.e1) != hipSuccess) return -1;
        ctx->ktimes.push_back(b);
    }
    o2v_hip_ctx::KernelBracket &b = ctx->ktimes[ctx->ktimes_used];
    b.name = name;
    if (hipEventRecord(b.e0, stream) != hipSuccess) return -1;
    return (int) ctx->ktimes_used++;
}

float ord2f_host(uint32_t o)
{
    const uint32_t b = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}

// Which runs take the occupancy-only mode (Params::occupancy_only) and the direct MAX path: decided in one place, for the
// run itself (o2v_hip_voxelize) and for the memory estimate of o2v_hip_max_slab_layers (1 byte per cell against 4 + 8).
struct GridModes {
    bool use_uv, exact_clip, occupancy_only, direct_max;
};
GridModes grid_modes(const o2v_hip_ctx *ctx, const o2v_hip_params *params)
{
    GridModes g;
    g.use_uv = ctx->d_uvs && ctx->any_textured;
    const char *exact = std::getenv("O2V_EXACT_CLIP");
    g.exact_clip = (params->flags & O2V_HIP_FLAG_EXACT_CLIP) || (exact && exact[0] == '1');
    const char *off = std::getenv("O2V_NO_DIRECT_MAX");
    const bool no_direct = off && off[0] == '1';
    // occupancy-only mode: no triangle has a material, so the result is the set of hit voxels, all white, with either
    // strategy.  Not in exact mode: the fast-vs-exact comparison covers this shortcut too.
    const char *no_occ = std::getenv("O2V_NO_OCCUPANCY_ONLY");
    g.occupancy_only = !ctx->d_types && !g.use_uv && !g.exact_clip && !no_direct && !(no_occ && no_occ[0] == '1');
    // Direct MAX path: MAX strategy; with textured triangles in its "pick" variant
    g.direct_max = (params->strategy == 0u || g.occupancy_only) && !no_direct;
    // ... unless most of the mesh will be subdivided anyway: the direct path then stays unused (direct_active() on the device:
    // at most half of the triangles subdivided) while its 64-bit grid takes two thirds of the grids' memory - 29 GB of 43 for the
    // reference README's 8192^3 showcase (19 k large triangles), allocated and zeroed for nothing.  Estimated from the
    // triangles' extents, known since the upload (k_tri_extent): a triangle 16 voxels or more across is subdivided as a rule
    // (a voxel box of 512 cells and more, voxelization.cpp:357-361, unless it is a sliver or axis-aligned).  Either way the
    // result is the same; only which route computes it, and what it needs of the device, changes.
    if (g.direct_max && !g.occupancy_only && ctx->max_tri_extent >= 0.f && ctx->n_tris) {
        const float *b = params->bounds_known ? params->bounds : ctx->mesh_bounds_hint;
        const float max_axis = std::max(b[3] - b[0], std::max(b[4] - b[1], b[5] - b[2]));
        const uint32_t ss = params->supersampling ? params->supersampling : 1u;
        float unit_norm = 0.f;
        for (int i = 0; i < 3; ++i)
            unit_norm = std::max(unit_norm, std::fabs((float) params->unit_transform[i * 3]) + std::fabs((float) params->unit_transform[i * 3 + 1]) +
                                                std::fabs((float) params->unit_transform[i * 3 + 2]));
        const float voxels_per_unit = unit_norm * (float) (params->resolution * ss) / max_axis;
        if (max_axis > 0.f && std::isfinite(voxels_per_unit) && voxels_per_unit > 0.f) {
            uint64_t large = 0;
            for (uint32_t e = 1; e < 255; ++e)   // bin e: extents in [2^(e-127), 2^(e-126))
                if (std::ldexp(1.0f, (int) e - 127) * voxels_per_unit >= 16.0f) large += ctx->ext_hist[e];
            large += ctx->ext_hist[255];          // (infinite / NaN extents: such triangles are subdivided until they vanish)
            if (large * 4 > ctx->n_tris * 3) g.direct_max = false;
        }
    }
    return g;
}

// The part of the output grid the dense grids are allocated for: the mesh's voxel bounding box, not the G^3 cube - the
// reference's VoxelMap only ever holds the chunks the mesh touches (util.hpp:179-208), and a long thin model at a high
// resolution (the reference README's showcase: r = 8192) fills a small fraction of the cube.  From the bounds of the
// uploaded mesh (known since the upload: ctx->mesh_bounds_hint) and the same transform k_setup computes on the device
// (compute_mesh_transform of the bounds in effect: the caller's, or the mesh's own): the eight corners of the mesh's box,
// transformed, one voxel of margin either side, in output space rounded outwards to whole bricks in x and y and cut to
// the slab in z.  A mesh whose bounds are unknown or not finite gets the whole cube.  O2V_NO_CROP=1: always the cube (A/B).
struct GridBox {
    uint32_t lo[3], hi[3];  // output space, [lo, hi); lo[0], lo[1] multiples of the brick edge
    bool empty;             // the slab does not meet the mesh's box: nothing to voxelize
};
GridBox grid_box(const o2v_hip_ctx *ctx, const o2v_hip_params *params, uint32_t ss, uint32_t z0, uint32_t z1)
{
    const uint32_t G = params->resolution, S = G * ss;
    GridBox b{{0u, 0u, z0}, {G, G, z1}, false};
    // (an x / y tile of the grid, o2v_hip_params::x_begin; 0, 0: the whole axis)
    if (params->x_begin || params->x_end) {
        b.lo[0] = params->x_begin;
        b.hi[0] = std::min(params->x_end, G);
    }
    if (params->y_begin || params->y_end) {
        b.lo[1] = params->y_begin;
        b.hi[1] = std::min(params->y_end, G);
    }
    if (b.lo[0] >= b.hi[0] || b.lo[1] >= b.hi[1]) b.empty = true;
    const char *off = std::getenv("O2V_NO_CROP");
    if ((off && off[0] == '1') || ctx->max_tri_extent < 0.f || ctx->n_tris == 0) return b;
    const float *h = ctx->mesh_bounds_hint;
    for (int i = 0; i < 6; ++i)
        if (!std::isfinite(h[i])) return b;
    const float *e = params->bounds_known ? params->bounds : h;
    const Affine a = compute_mesh_transform(V3{e[0], e[1], e[2]}, V3{e[3], e[4], e[5]}, S, params->unit_transform);
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int corner = 0; corner < 8; ++corner) {
        const V3 t = affine_apply(a, V3{h[(corner & 1) ? 3 : 0], h[(corner & 2) ? 4 : 1], h[(corner & 4) ? 5 : 2]});
        const float c[3] = {t.x, t.y, t.z};
        for (int k = 0; k < 3; ++k) {
            if (!std::isfinite(c[k])) return b;
            mn[k] = std::min<double>(mn[k], c[k]);
            mx[k] = std::max<double>(mx[k], c[k]);
        }
    }
    for (int k = 0; k < 3; ++k) {
        // sample space: [floor(min) - 1, floor(max) + 2), within [0, S) (a negative coordinate is voxel 0 on the device)
        const double lo_s = std::min<double>(std::max<double>(std::floor(mn[k]) - 1.0, 0.0), (double) S);
        const double hi_s = std::min<double>(std::max<double>(std::floor(mx[k]) + 2.0, 0.0), (double) S);
        uint32_t lo_o = (uint32_t) lo_s / ss, hi_o = ((uint32_t) hi_s + ss - 1u) / ss;
        if (k < 2) lo_o &= ~(kBrickX - 1u);  // (kBrickX == kBrickY)
        b.lo[k] = std::max(b.lo[k], lo_o);
        b.hi[k] = std::min(b.hi[k], hi_o);
        if (b.lo[k] >= b.hi[k]) b.empty = true;
    }
    return b;
}
static_assert(kBrickX == kBrickY, "grid_box aligns x and y alike");

// One pass of the pipeline with the current capacities.  Fills h_ctr; the caller checks for overflow.
int run_pass(o2v_hip_ctx *ctx, const Params &p, bool use_uv, uint32_t n_rounds)
{
    hipStream_t s = ctx->stream;
    const uint32_t persistent = (uint32_t) ctx->num_cus * 8u;
    ctx->ktimes_used = 0;
    if (ctx->stage_events) O2V_CHECK(hipEventRecord(ctx->ev[0], s));
    // (the counters were zeroed behind the previous pass, off its critical path, unless something else used them since)
    if (!ctx->ctr_clean) O2V_LAUNCH("k_init", s, k_init, dim3(1), dim3(64), 0, s, ctx->d_ctr, kPassCounterWords);
    ctx->ctr_clean = false;
    if (!p.bounds_known) {
        // one workgroup per CU: every workgroup ends with six atomics on the same six words, which serialise (1024
        // workgroups: 43 us for 31 MB, 256: 24 us)
        O2V_LAUNCH("k_bounds", s, k_bounds, dim3((uint32_t) std::min<uint64_t>((uint64_t) ctx->num_cus, (p.n_tris * 9 / 12 + kBoundsBlock) / kBoundsBlock)),
                           dim3(kBoundsBlock), 0, s, ctx->d_verts, p.n_tris * 9, ctx->d_ctr);
    }
    // (letting the last workgroup of k_bounds compute the transform - one launch less - was measured: the stage 0.021 -> 0.027 ms)
    O2V_LAUNCH("k_setup", s, k_setup, dim3(1), dim3(64), 0, s, ctx->d_ctr, p);
    if (ctx->stage_events) O2V_CHECK(hipEventRecord(ctx->ev[1], s));

    // After a slab plan the z extent of every block of 256 triangles is known: a slab that is not the whole grid visits only
    // the blocks that meet it (on N GPUs ~1/N of the list, compacted by k_list_blocks).
    const bool have_zrange = ctx->zrange_generation == ctx->tri_generation;
    const uint64_t n_tri_blocks = (p.n_tris + kBlock - 1) / kBlock;
    uint32_t *block_list = nullptr;
    const char *list_hook = std::getenv("O2V_TEST_BLOCK_LIST");  // test hook: 1 = use the list for any mesh
    const uint64_t list_above = list_hook && list_hook[0] == '1' ? 0ull : 256ull;
    if (have_zrange && (p.zs0 != 0 || p.zs1 < p.S) && n_tri_blocks > list_above && ctx->d_block_list && ctx->cap_block_list >= n_tri_blocks) {
        block_list = ctx->d_block_list;
        // (the list's counter is a word of the pass's counters: zero since k_init)
        O2V_LAUNCH("k_list_blocks", s, k_list_blocks, dim3((uint32_t) std::min<uint64_t>((uint64_t) ctx->num_cus * 4u, (n_tri_blocks + kBlock - 1) / kBlock)),
                           dim3(kBlock), 0, s, ctx->d_zrange, ctx->d_zrange_xform, ctx->d_ctr, ctx->d_block_list, ctx->d_block_count, p);
    }
    // (root_bypass: most super-blocks of three sub-batches are only read and counted - fewer workgroups with several super-blocks
    // each keep the loads of the next one in flight behind the current one's arithmetic)
    // (a slab visits the listed blocks only - how many is known on the device; the slab's share of z, and a third more, is the
    // estimate here: too many workgroups cost this kernel as much again, 0.033 -> 0.06 - 0.08 ms on a slab of the 8-GPU weak job)
    uint64_t walked_blocks = n_tri_blocks;
    if (block_list && p.S) walked_blocks = std::min<uint64_t>(n_tri_blocks, (uint64_t) ((double) n_tri_blocks * (double) (p.zs1 - p.zs0) / (double) p.S * 1.33) + 64u);
    uint64_t root_wgs = p.root_bypass ? std::max<uint64_t>((uint64_t) ctx->num_cus * 2u, walked_blocks / 6u) : (p.n_tris + kBlock - 1) / kBlock;
    const uint32_t *k1_list = block_list, *k1_count = ctx->d_block_count;
    if (p.solo_roots) {
        // every root triangle is a leaf of one tile or misses the slab (o2v_hip_voxelize): nothing for K1 to write, and what it
        // would count k_voxelize_occ counts
    }
    else if (ctx->lean_roots && p.root_bypass) {
        // a tessellated surface: the one-tile root triangles are counted by a kernel of their own, k_expand_roots only walks the
        // blocks that hold something else (k_count_roots)
        O2V_LAUNCH("k_count_roots", s, k_count_roots, dim3((uint32_t) std::min<uint64_t>((uint64_t) ctx->num_cus * kCountRootsWgsPerCu, std::max<uint64_t>((walked_blocks + 3u) / 4u, 1))),
                           dim3(kBlock), 0, s, ctx->d_verts, ctx->d_ctr, block_list, ctx->d_block_count, ctx->d_need_list, &ctx->d_ctr->n_need_blocks, p);
        k1_list = ctx->d_need_list;
        k1_count = &ctx->d_ctr->n_need_blocks;
        root_wgs = (uint64_t) ctx->num_cus;  // (the list is short or empty)
    }
    if (!p.solo_roots)
        O2V_LAUNCH("k_expand_roots", s, k_expand_roots, dim3(std::min<uint64_t>(persistent, std::max<uint64_t>(root_wgs, 1))),
                           dim3(kBlock), 0, s, ctx->d_verts, ctx->d_uvs, ctx->d_ctr, ctx->d_leaves, ctx->d_tiles,
                           ctx->d_big, ctx->d_nodes[0], have_zrange ? ctx->d_zrange : nullptr,
                           ctx->d_zrange_xform, k1_list, k1_count, p);
    for (uint32_t round = 0; round < (p.solo_roots ? 0u : n_rounds); ++round) {
        // most rounds are empty or small: a narrow grid keeps an empty launch short (the kernel strides over its input)
        O2V_LAUNCH("k_expand_nodes", s, k_expand_nodes, dim3((uint32_t) ctx->num_cus * 2u), dim3(kBlock), 0, s, ctx->d_nodes[round & 1], round,
                           ctx->d_ctr, ctx->d_leaves, ctx->d_tiles, ctx->d_big, ctx->d_nodes[(round + 1) & 1], p);
    }
    // (left out if no leaf of this mesh can have more than four tiles, o2v_hip_voxelize; should one turn up, the pass is repeated)
    if (!ctx->skip_big && !p.solo_roots)
        O2V_LAUNCH("k_expand_big", s, k_expand_big, dim3((uint32_t) ctx->num_cus * 2u), dim3(kBlock), 0, s, ctx->d_big, ctx->d_ctr, ctx->d_tiles, p);
    // (a mesh that pooled no hits in its last pass with these settings - every triangle whole and on the direct MAX path - will
    // not pool any now: the two launches are left out; should K1's counters say otherwise, the pass is repeated with them)
    const uint64_t mark_key = ctx->tri_generation * 1000003ull + p.blend * 7u + p.S * 131ull + p.zs0 * 31ull + p.zs1 + p.exact_clip * 3u;
    const bool skip_mark = !ctx->force_general && ctx->no_pool_key == mark_key + 1u;
    ctx->marked_bricks = !p.occupancy_only && !skip_mark;
    if (ctx->marked_bricks) {
        // The bricks that can receive pooled hits are listed before k_voxelize (every brick a leaf's clamped box touches: a
        // superset of the bricks that do), and every listed brick gets a hit slab: the first kInlineHits hits of a cell go
        // there directly.  Neither kernel has work if the pass pools no hits (decided on the device from K1's counters).
        O2V_LAUNCH("k_mark_bricks", s, k_mark_bricks, dim3((uint32_t) ctx->num_cus * 4u), dim3(kBlock), 0, s, ctx->d_leaves, ctx->d_ctr, ctx->d_brick_dirty,
                           ctx->force_general ? 1u : 0u, p);
        const uint32_t flag_groups = (p.n_bricks + 15u) / 16u;
        O2V_LAUNCH("k_scan_flags", s, k_scan_flags, dim3(std::min<uint32_t>((uint32_t) ctx->num_cus * kScanFlagsWgsPerCu, std::max<uint32_t>(1u, (flag_groups + kBlock * kFlagLoads - 1) / (kBlock * kFlagLoads)))),
                           dim3(kBlock), 0, s, ctx->d_brick_dirty, &ctx->d_ctr->n_dirty, ctx->d_dirty_list, ctx->d_ctr, ctx->d_brick_slab, ctx->force_general ? 1u : 0u, p);
    }
    if (ctx->stage_events) O2V_CHECK(hipEventRecord(ctx->ev[2], s));
    // (occupancy only: every hit takes the direct path whatever K1 counted - nothing to decide)
    const bool decide_from_k1 = p.direct_max && !(p.occupancy_only && !ctx->force_general);
    if (decide_from_k1) {
        // K1's counters go to the host on an auxiliary stream while k_voxelize runs (see below)
        if (!ctx->aux[0]) O2V_CHECK(hipStreamCreateWithFlags(&ctx->aux[0], hipStreamNonBlocking));
        O2V_CHECK(hipEventRecord(ctx->ev_k1, s));
        O2V_CHECK(hipStreamWaitEvent(ctx->aux[0], ctx->ev_k1, 0));
        O2V_CHECK(hipMemcpyAsync(ctx->h_ctr, ctx->d_ctr, kPassCounterWords * 4u, hipMemcpyDeviceToHost, ctx->aux[0]));
    }

    {
        // persistent workgroups: four wavefronts per SIMD, in workgroups of VoxShape<UV>::block threads
        if (use_uv) {
            const uint32_t blocks = (uint32_t) ctx->num_cus * (uint32_t) O2V_K2_WAVES_UV * (kBlock / VoxShape<true>::block);
            O2V_LAUNCH_K2("k_voxelize<true>", k_voxelize<true>, dim3(blocks), dim3(VoxShape<true>::block), ctx->d_leaves, ctx->d_tiles,
                          ctx->d_ctr, ctx->d_grid, ctx->d_brick_dirty, ctx->d_pool, ctx->d_jobq, p);
        }
        else if (p.occupancy_only) {
            const uint32_t blocks = (uint32_t) ctx->num_cus * (uint32_t) O2V_K2_WAVES * (kBlock / VoxShape<false>::block);
            O2V_LAUNCH_K2("k_voxelize_occ", k_voxelize_occ, dim3(blocks), dim3(VoxShape<false>::block), ctx->d_leaves, ctx->d_tiles,
                          ctx->d_ctr, ctx->d_grid, ctx->d_brick_dirty, ctx->d_pool, ctx->d_jobq, ctx->d_verts, block_list, ctx->d_block_count, p);
        }
        else {
            const uint32_t blocks = (uint32_t) ctx->num_cus * (uint32_t) O2V_K2_WAVES * (kBlock / VoxShape<false>::block);
            O2V_LAUNCH_K2("k_voxelize<false>", k_voxelize<false>, dim3(blocks), dim3(VoxShape<false>::block), ctx->d_leaves, ctx->d_tiles,
                          ctx->d_ctr, ctx->d_grid, ctx->d_brick_dirty, ctx->d_pool, ctx->d_jobq, p);
        }
    }
    if (ctx->stage_events) O2V_CHECK(hipEventRecord(ctx->ev[3], s));

    // With the direct MAX path the rest of the pass depends on the mesh: one whose triangles are all voxelized whole needs
    // neither the counting sort nor the replay (a dozen launches that would each find nothing), one whose triangles are
    // mostly subdivided does not use the 64-bit grid at all.  Both follow from K1's counters, which reached the host
    // while k_voxelize was running: the follow-up stages are enqueued behind it without the stream ever draining.
    bool run_general = true, run_emit = false;
    if (p.direct_max) {
        if (decide_from_k1) {
            O2V_CHECK(hipStreamSynchronize(ctx->aux[0]));
            const Counters &h = *ctx->h_ctr;
            run_emit = p.occupancy_only || h.n_nodes[0] <= h.n_root_leaves;  // direct_active() on the device
            // hits are pooled only for leaves of subdivided triangles: without any, every hit goes straight into the 64-bit grid
            // (occupancy-only mode: those too)
            run_general = !run_emit || (h.n_nodes[0] != 0 && !p.occupancy_only) || ctx->force_general;
        }
        else {
            run_emit = true;
            run_general = false;
        }
        if (run_emit) {
            const uint32_t groups = (p.n_bricks + 15u) / 16u;
            O2V_LAUNCH("k_scan_flags", s, k_scan_flags, dim3(std::min<uint32_t>((uint32_t) ctx->num_cus * kScanFlagsWgsPerCu, std::max<uint32_t>(1u, (groups + kBlock * kFlagLoads - 1) / (kBlock * kFlagLoads)))),
                               dim3(kBlock), 0, s, ctx->d_dirty_max, &ctx->d_ctr->n_dirty_max, ctx->d_dirty_list_max, ctx->d_ctr, (uint32_t *) nullptr, 0u, p);
        }
    }

    if (run_general && !ctx->marked_bricks && !p.occupancy_only) {
        // the guess above was wrong (it cannot be for the same triangles and settings): no brick has a slab number, the pass is void
        ctx->no_pool_key = 0;
        ctx->mark_missing = true;
        run_general = false;
    }
    else if (!p.occupancy_only) {
        ctx->no_pool_key = run_general ? 0 : mark_key + 1u;
    }
    const ResolveLists lists{ctx->d_list_lane16, ctx->d_list_lane, ctx->d_list_w64, ctx->d_list_mid, ctx->d_list_long,
                             ctx->d_list_big, ctx->d_list_huge, ctx->d_list_lane8, p.cap_vox};
    if (run_general) {
        // (the brick list was made before k_voxelize: see k_mark_bricks)
        O2V_LAUNCH("k_scan_bricks", s, k_scan_bricks, dim3((uint32_t) ctx->num_cus * 2u), dim3(kBlock), 0, s, ctx->d_grid,
                           ctx->d_dirty_list, ctx->d_ctr, ctx->d_occ, lists, p);
    }
    ctx->last_ran_general = run_general;
    if (ctx->stage_events) O2V_CHECK(hipEventRecord(ctx->ev[4], s));

    Materials m{ctx->d_types, ctx->d_colors, ctx->d_texids, ctx->d_textures, ctx->n_textures};
    if (run_general) {
        const SortedView sorted_view{reinterpret_cast<const uint32_t *>(ctx->d_sorted), use_uv ? 6u : 4u};
        const SortedView slab_view{ctx->d_slabs, use_uv ? 6u : 4u};
        // What follows k_scan_bricks runs side by side (the tiers work on disjoint cells, filed by k_scan_bricks):
        //   main stream   tier 1 on the inline cells - most cells; their hits are in the slabs, so it needs no sorted array -
        //                 then, once that exists, on the short cells of bricks without a slab (none as a rule)
        //   aux 0         the counting sort of what the slabs do not hold (k_scatter), then the 9..16-hit tier
        //   aux 1, 2      (behind the sort) the counter reset and the cooperative tiers
        const bool fork = debug_sync_level() != 1;
        hipStream_t sw = s, sm = s, sl = s;
        if (fork) {
            for (int j = 0; j < 3; ++j)
                if (!ctx->aux[j]) O2V_CHECK(hipStreamCreateWithFlags(&ctx->aux[j], hipStreamNonBlocking));
            sw = ctx->aux[0];
            sm = ctx->aux[1];
            sl = ctx->aux[2];
            O2V_CHECK(hipEventRecord(ctx->ev_fork, s));
            O2V_CHECK(hipStreamWaitEvent(sw, ctx->ev_fork, 0));
        }
        // (the inline cells with 5 .. 8 hits: from the slabs like the main stream's launch, so it starts with it)
        if (fork) O2V_CHECK(hipStreamWaitEvent(sm, ctx->ev_fork, 0));
        if (use_uv)
            O2V_LAUNCH("k_resolve_inline_list<6>", sm, k_resolve_inline_list<6>, dim3((uint32_t) ctx->num_cus * 2u), dim3(kBlock), 0, sm, ctx->d_list_lane8,
                               &ctx->d_ctr->n_lane8, ctx->d_ctr, ctx->d_occ, slab_view, m, ctx->d_out, p.cap_vox, p);
        else
            O2V_LAUNCH("k_resolve_inline_list<4>", sm, k_resolve_inline_list<4>, dim3((uint32_t) ctx->num_cus * 2u), dim3(kBlock), 0, sm, ctx->d_list_lane8,
                               &ctx->d_ctr->n_lane8, ctx->d_ctr, ctx->d_occ, slab_view, m, ctx->d_out, p.cap_vox, p);
        O2V_LAUNCH("k_scatter", sw, k_scatter, dim3(persistent), dim3(kBlock), 0, sw, ctx->d_pool, ctx->d_grid, ctx->d_ctr,
                           reinterpret_cast<uint32_t *>(ctx->d_sorted), use_uv ? 6u : 4u, p);
        if (fork) {
            O2V_CHECK(hipEventRecord(ctx->ev_sorted, sw));
            O2V_CHECK(hipStreamWaitEvent(sm, ctx->ev_sorted, 0));
            O2V_CHECK(hipStreamWaitEvent(sl, ctx->ev_sorted, 0));
        }
        // (tier 1 on the inline cells runs as fast with two workgroups per CU as with eight - it is not bound by the wavefronts in
        // flight - and leaves the counting sort and the cooperative tiers beside it room: bench mesh with BLEND -0.1 ms)
        uint32_t resolve_wgs = (uint32_t) ctx->num_cus * 2u;
        if (const char *e = std::getenv("O2V_RESOLVE_WGS_PER_CU"); e && std::atoi(e) > 0) resolve_wgs = (uint32_t) ctx->num_cus * (uint32_t) std::atoi(e);
        if (use_uv)
            O2V_LAUNCH("k_resolve<6>", s, k_resolve<6>, dim3(resolve_wgs), dim3(kBlock), 0, s, ctx->d_occ, sorted_view, slab_view, ctx->d_ctr, m,
                               ctx->d_out, 0u, p);
        else
            O2V_LAUNCH("k_resolve<4>", s, k_resolve<4>, dim3(resolve_wgs), dim3(kBlock), 0, s, ctx->d_occ, sorted_view, slab_view, ctx->d_ctr, m,
                               ctx->d_out, 0u, p);
        if (fork) O2V_CHECK(hipStreamWaitEvent(s, ctx->ev_sorted, 0));
        if (use_uv)
            O2V_LAUNCH("k_resolve<6>", s, k_resolve<6>, dim3(persistent), dim3(kBlock), 0, s, ctx->d_occ, sorted_view, slab_view, ctx->d_ctr, m,
                               ctx->d_out, 1u, p);
        else
            O2V_LAUNCH("k_resolve<4>", s, k_resolve<4>, dim3(persistent), dim3(kBlock), 0, s, ctx->d_occ, sorted_view, slab_view, ctx->d_ctr, m,
                               ctx->d_out, 1u, p);
        if (use_uv)
            O2V_LAUNCH("k_resolve_list16<6>", sw, k_resolve_list16<6>, dim3((uint32_t) ctx->num_cus * 4u), dim3(kBlock), 0, sw, ctx->d_list_lane16,
                               &ctx->d_ctr->n_lane16, ctx->d_ctr, ctx->d_occ, sorted_view, m, ctx->d_out, p.cap_vox, p);
        else
            O2V_LAUNCH("k_resolve_list16<4>", sw, k_resolve_list16<4>, dim3((uint32_t) ctx->num_cus * 4u), dim3(kBlock), 0, sw, ctx->d_list_lane16,
                               &ctx->d_ctr->n_lane16, ctx->d_ctr, ctx->d_occ, sorted_view, m, ctx->d_out, p.cap_vox, p);
        // (behind the 9..16-hit tier: nothing waits for the counters' reset but the next pass)
        O2V_LAUNCH("k_reset_bricks", sw, k_reset_bricks, dim3((uint32_t) ctx->num_cus * 4u), dim3(kBlock), 0, sw, ctx->d_grid,
                           ctx->d_dirty_list, ctx->d_ctr, p);
        {
            // the cooperative tiers for 17 .. 256 hits: one launch of one-wavefront workgroups (k_resolve_tiers)
            const uint32_t g_mid = (uint32_t) ctx->num_cus * 8u, g_w = (uint32_t) ctx->num_cus * 16u;
            const TierLists tl{ctx->d_list_mid, ctx->d_list_w64, ctx->d_list_lane, &ctx->d_ctr->n_mid, &ctx->d_ctr->n_w64, &ctx->d_ctr->n_lane, &ctx->d_ctr->cursor_mid};
            O2V_LAUNCH("k_resolve_tiers", sm, k_resolve_tiers, dim3(g_mid + 2u * g_w), dim3(64), 0, sm, tl, g_mid, g_w, ctx->d_ctr, ctx->d_occ, sorted_view, m,
                               ctx->d_out, p.cap_vox, p);
        }
        O2V_LAUNCH("k_resolve_sorted<256,2048>", sl, (k_resolve_sorted<kBlock, kLongList>), dim3((uint32_t) ctx->num_cus * 2u), dim3(kBlock), 0, sl,
                           ctx->d_list_long, &ctx->d_ctr->n_long, &ctx->d_ctr->cursor_long, ctx->d_ctr, ctx->d_occ, sorted_view, m,
                           ctx->d_out, p.cap_vox, p);
        O2V_LAUNCH("k_resolve_big", sl, k_resolve_big, dim3((uint32_t) ctx->num_cus / 2u), dim3(kBigThreads), kBigList * 12u, sl, ctx->d_list_big,
                           ctx->d_ctr, ctx->d_occ, sorted_view, m, ctx->d_out, p.cap_vox, p);
        if (ctx->d_scratch_key) {
            O2V_LAUNCH("k_resolve_huge", sl, k_resolve_huge, dim3((uint32_t) ctx->num_cus / 2u), dim3(kBlock), 0, sl, ctx->d_list_huge,
                               ctx->d_ctr, ctx->d_occ, sorted_view, m, ctx->d_out, ctx->d_scratch_key,
                               ctx->d_scratch_idx, ctx->cap_scratch, p.cap_vox, p);
        }
        if (fork)
            for (int j = 0; j < 3; ++j) {
                O2V_CHECK(hipEventRecord(ctx->ev_join[j], ctx->aux[j]));
                O2V_CHECK(hipStreamWaitEvent(s, ctx->ev_join[j], 0));
            }
    }
    {
        if (run_emit && p.pick_max) {
            O2V_LAUNCH("k_pick", s, k_pick, dim3(persistent), dim3(kBlock), 0, s, ctx->d_pool, ctx->d_ctr, p);
        }
        if (run_emit && p.occupancy_only) {
            O2V_LAUNCH("k_emit_occ", s, k_emit_occ, dim3((uint32_t) ctx->num_cus * 3u), dim3(kBlock), 0, s, ctx->d_dirty_list_max, ctx->d_ctr, ctx->d_out, p);
        }
        else if (run_emit) {
            // every voxel's winner is in the 64-bit grid now (k_voxelize: unsplit triangles, resolve: the rest)
            O2V_LAUNCH("k_emit_max", s, k_emit_max, dim3((uint32_t) ctx->num_cus * 3u), dim3(kBlock), 0, s, ctx->d_dirty_list_max, ctx->d_ctr, m,
                               ctx->d_out, p);
        }
    }
    if (ctx->stage_events) O2V_CHECK(hipEventRecord(ctx->ev[5], s));
    // (a kernel that writes the counters into the page-locked copy instead of this copy command was measured: the same step time)
    O2V_CHECK(hipMemcpyAsync(ctx->h_ctr, ctx->d_ctr, kPassCounterWords * 4u, hipMemcpyDeviceToHost, s));
    // (polling the stream with hipStreamQuery instead was measured: the same step time - the runtime's wait spins already)
    O2V_CHECK(hipStreamSynchronize(s));
    O2V_CHECK(hipGetLastError());
    ctx->kernel_times.clear();
    for (size_t i = 0; i < ctx->ktimes_used; ++i) {
        const o2v_hip_ctx::KernelBracket &b = ctx->ktimes[i];
        float ms = 0.f;
        O2V_CHECK(hipEventElapsedTime(&ms, b.e0, b.e1));
        auto it = std::find_if(ctx->kernel_times.begin(), ctx->kernel_times.end(),
                               [&](const o2v_hip_kernel_time &k) { return std::strcmp(k.name, b.name) == 0; });
        if (it == ctx->kernel_times.end()) {
            o2v_hip_kernel_time k{};
            std::snprintf(k.name, sizeof(k.name), "%s", b.name);
            ctx->kernel_times.push_back(k);
            it = ctx->kernel_times.end() - 1;
        }
        it->ms += ms;
        it->launches += 1;
    }
    // the next pass's counters: zeroed now, behind this pass (its results are on the host)
    hipLaunchKernelGGL(k_init, dim3(1), dim3(64), 0, s, ctx->d_ctr, kPassCounterWords);
    ctx->ctr_clean = true;
    return O2V_HIP_OK;
}

}  // namespace

namespace o2v {

hipStream_t ctx_stream(o2v_hip_ctx *ctx) { return ctx->stream; }
int ctx_device(const o2v_hip_ctx *ctx) { return ctx->device; }

int ctx_alloc_triangles(o2v_hip_ctx *ctx, uint64_t count, bool uvs, bool types, bool colors, bool texids)
{
    if (count >= (1ull << 29)) {
        ctx->err = "triangle count must be below 2^29";
        return O2V_HIP_ERR_LIMIT;
    }
    O2V_CHECK(hipSetDevice(ctx->device));
    O2V_CHECK(hipStreamSynchronize(ctx->stream));  // nothing may still read the arrays that are about to be replaced
    int rc;
    if ((rc = ensure_array(ctx, ctx->d_verts, ctx->cap_tri_bytes[0], count * 9, true))) return rc;
    if ((rc = ensure_array(ctx, ctx->d_uvs, ctx->cap_tri_bytes[1], count * 6, uvs))) return rc;
    if ((rc = ensure_array(ctx, ctx->d_types, ctx->cap_tri_bytes[2], count, types))) return rc;
    if ((rc = ensure_array(ctx, ctx->d_colors, ctx->cap_tri_bytes[3], count * 3, colors))) return rc;
    if ((rc = ensure_array(ctx, ctx->d_texids, ctx->cap_tri_bytes[4], count, texids))) return rc;
    ctx->n_tris = count;
    ctx->tri_generation += 1;
    ctx->max_tri_extent = -1.f;
    ctx->any_textured = false;
    return O2V_HIP_OK;
}

TriBuffers ctx_tri_buffers(o2v_hip_ctx *ctx)
{
    return TriBuffers{ctx->d_verts, ctx->d_uvs, ctx->d_types, ctx->d_colors, ctx->d_texids, ctx->n_tris};
}

TriHints ctx_tri_hints(const o2v_hip_ctx *ctx)
{
    TriHints h;
    h.any_textured = ctx->any_textured;
    for (int i = 0; i < 6; ++i) h.bounds[i] = ctx->mesh_bounds_hint[i];
    h.max_tri_extent = ctx->max_tri_extent;
    std::memcpy(h.ext_hist, ctx->ext_hist, sizeof(h.ext_hist));
    return h;
}

// After the arrays are filled (on ctx's stream): records whether any triangle is textured and the launch-configuration
// hints (mesh bounds and largest triangle extent, see k_tri_extent) - taken from `hints` if another rank already
// computed them for the same triangles, else computed here.  Waits for the stream.
int ctx_finish_triangles(o2v_hip_ctx *ctx, bool any_textured, const TriHints *hints)
{
    O2V_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint64_t count = ctx->n_tris;
    ctx->any_textured = any_textured;
    if (hints) {
        for (int i = 0; i < 6; ++i) ctx->mesh_bounds_hint[i] = hints->bounds[i];
        ctx->max_tri_extent = hints->max_tri_extent;
        std::memcpy(ctx->ext_hist, hints->ext_hist, sizeof(ctx->ext_hist));
        ctx->any_textured = hints->any_textured;
    }
    else if (count) {
        ctx->ctr_clean = false;
        hipLaunchKernelGGL(k_init, dim3(1), dim3(64), 0, s, ctx->d_ctr, (uint32_t) (sizeof(Counters) / 4));
        hipLaunchKernelGGL(k_bounds, dim3((uint32_t) std::min<uint64_t>((uint64_t) ctx->num_cus, (count * 9 / 12 + kBoundsBlock) / kBoundsBlock)),
                           dim3(kBoundsBlock), 0, s, ctx->d_verts, count * 9, ctx->d_ctr);
        hipLaunchKernelGGL(k_tri_extent, dim3((uint32_t) std::min<uint64_t>((uint64_t) ctx->num_cus * 4u, (count + kBlock - 1) / kBlock)),
                           dim3(kBlock), 0, s, ctx->d_verts, count, &ctx->d_ctr->pad2, ctx->d_ctr->ext_hist);
        O2V_CHECK(hipMemcpyAsync(ctx->h_ctr, ctx->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, s));
        O2V_CHECK(hipStreamSynchronize(s));
        for (int i = 0; i < 6; ++i) ctx->mesh_bounds_hint[i] = ord2f_host(ctx->h_ctr->bounds_enc[i]);
        ctx->max_tri_extent = ord2f_host(ctx->h_ctr->pad2);
        std::memcpy(ctx->ext_hist, ctx->h_ctr->ext_hist, sizeof(ctx->ext_hist));
        return O2V_HIP_OK;
    }
    O2V_CHECK(hipStreamSynchronize(s));
    return O2V_HIP_OK;
}

}  // namespace o2v

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
    // (O2V_INIT_TIMES=1: where a new process' first session spends its time - the runtime's start is most of a CLI run)
    const char *init_times = std::getenv("O2V_INIT_TIMES");
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!(init_times && init_times[0] == '1')) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[o2v_hip_create] %s: %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return O2V_HIP_ERR_NO_DEVICE;
    lap("hipGetDeviceCount (runtime start)");
    if (hipSetDevice(device) != hipSuccess) return O2V_HIP_ERR_NO_DEVICE;
    lap("hipSetDevice");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return O2V_HIP_ERR_NO_DEVICE;
    lap("hipGetDeviceProperties");
    o2v_hip_ctx *ctx = new o2v_hip_ctx;
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return O2V_HIP_ERR_HIP;
    }
    lap("the stream");
    for (auto &e : ctx->ev)
        if (create_timing_event(&e) != hipSuccess) {
            delete ctx;
            return O2V_HIP_ERR_HIP;
        }
    bool ok = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&ctx->ev_sorted, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&ctx->ev_k1, hipEventDisableTiming) == hipSuccess;
    // (aux[1] and aux[2] - the cooperative resolve tiers - are created by the first pass that forks; a stream costs 0.3 ms
    // in a warm process and several in a new one, and a mesh on the direct route never needs them)
    for (int j = 0; j < 3 && ok; ++j) ok = hipEventCreateWithFlags(&ctx->ev_join[j], hipEventDisableTiming) == hipSuccess;
    lap("events");
    // (the auxiliary streams are made by the first pass that needs one - 7 - 8 ms each in a new process, and the occupancy-only
    // route, every STL, never does)
    if (!ok) {
        delete ctx;
        return O2V_HIP_ERR_HIP;
    }
    if (hipMalloc(reinterpret_cast<void **>(&ctx->d_ctr), sizeof(Counters)) != hipSuccess) {
        delete ctx;
        return O2V_HIP_ERR_OUT_OF_MEMORY;
    }
    lap("first hipMalloc");
    if (hipHostMalloc(reinterpret_cast<void **>(&ctx->h_ctr), sizeof(Counters), hipHostMallocDefault) != hipSuccess) {
        delete ctx;
        return O2V_HIP_ERR_OUT_OF_MEMORY;
    }
    lap("first hipHostMalloc");
    ctx->d_block_count = &ctx->d_ctr->n_listed_blocks;
    // k_resolve_big sorts in 96 KiB of dynamic LDS (above the default 64 KiB limit)
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(&k_resolve_big), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int) (kBigList * 12u));
    lap("hipFuncSetAttribute (code object)");
    // the first pass' counters, zeroed now: the launch also makes the runtime load the device code here - on the thread that
    // creates the session beside the input's parsing (obj2voxel_voxelize) - rather than in front of the first pass
    hipLaunchKernelGGL(k_init, dim3(1), dim3(64), 0, ctx->stream, ctx->d_ctr, kPassCounterWords);
    if (hipStreamSynchronize(ctx->stream) == hipSuccess) ctx->ctr_clean = true;
    else (void) hipGetLastError();
    lap("first launch");
    *out_ctx = ctx;
    return O2V_HIP_OK;
}

void o2v_hip_destroy(o2v_hip_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->poisoned) {
        // a stuck collective is queued on the context's stream (o2v_hip_voxelize_sharded timed out): synchronising or freeing would
        // block for ever.  The device memory goes with the process, which the caller was told to end (include/o2v_hip.h).
        delete ctx;
        return;
    }
    (void) hipSetDevice(ctx->device);
    if (ctx->stream) (void) hipStreamSynchronize(ctx->stream);
    void *ptrs[] = {ctx->d_verts, ctx->d_uvs,  ctx->d_colors,   ctx->d_types,    ctx->d_texids, ctx->d_textures,
                    ctx->d_ctr,   ctx->d_leaves, ctx->d_tiles,  ctx->d_big,      ctx->d_nodes[0], ctx->d_nodes[1],
                    ctx->d_pool,  ctx->d_sorted, ctx->d_occ,  ctx->d_out,      ctx->d_grid, ctx->d_jobq,
                    ctx->d_list_lane8, ctx->d_list_lane16, ctx->d_list_w64, ctx->d_list_lane, ctx->d_list_mid, ctx->d_list_long, ctx->d_list_big, ctx->d_list_huge, ctx->d_scratch_key, ctx->d_scratch_idx,
                    ctx->d_brick_dirty, ctx->d_dirty_list, ctx->d_brick_slab, ctx->d_slabs, ctx->d_maxgrid, ctx->d_dirty_max, ctx->d_dirty_list_max, ctx->d_pick_extra};
    for (void *q : ptrs)
        if (q) (void) hipFree(q);
    for (uint8_t *q : ctx->d_texpix)
        if (q) (void) hipFree(q);
    if (ctx->h_ctr) (void) hipHostFree(ctx->h_ctr);
    if (ctx->d_zhist) (void) hipFree(ctx->d_zhist);
    if (ctx->d_zrange) (void) hipFree(ctx->d_zrange);
    if (ctx->d_need_list) (void) hipFree(ctx->d_need_list);
    if (ctx->d_plan_gather) (void) hipFree(ctx->d_plan_gather);
    if (ctx->d_block_list) (void) hipFree(ctx->d_block_list);
    if (ctx->d_zrange_xform) (void) hipFree(ctx->d_zrange_xform);
    if (ctx->h_zhist) (void) hipHostFree(ctx->h_zhist);
    for (o2v_hip_staging &b : ctx->stage)
        for (void *q : {(void *) b.verts, (void *) b.uvs, (void *) b.types, (void *) b.colors, (void *) b.texids})
            if (q) (void) hipHostFree(q);
    for (auto &e : ctx->ev_stage)
        if (e) (void) hipEventDestroy(e);
    for (void *q : ctx->retired) (void) hipFree(q);
    if (ctx->d_counts) (void) hipFree(ctx->d_counts);
    if (ctx->h_counts) (void) hipHostFree(ctx->h_counts);
    if (ctx->d_status) (void) hipFree(ctx->d_status);
    if (ctx->h_status) (void) hipHostFree(ctx->h_status);
    for (auto &e : ctx->ev_coll)
        if (e) (void) hipEventDestroy(e);
    for (auto &e : ctx->ev)
        if (e) (void) hipEventDestroy(e);
    for (auto &b : ctx->ktimes) {
        if (b.e0) (void) hipEventDestroy(b.e0);
        if (b.e1) (void) hipEventDestroy(b.e1);
    }
    if (ctx->ev_fork) (void) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_sorted) (void) hipEventDestroy(ctx->ev_sorted);
    if (ctx->ev_k1) (void) hipEventDestroy(ctx->ev_k1);
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
    int rc;
    if ((rc = o2v::ctx_alloc_triangles(ctx, count, uvs != nullptr, types != nullptr, colors != nullptr, texids != nullptr))) return rc;
    if (count) {
        hipStream_t s = ctx->stream;
        O2V_CHECK(hipMemcpyAsync(ctx->d_verts, verts, count * 9 * sizeof(float), hipMemcpyHostToDevice, s));
        if (uvs) O2V_CHECK(hipMemcpyAsync(ctx->d_uvs, uvs, count * 6 * sizeof(float), hipMemcpyHostToDevice, s));
        if (types) O2V_CHECK(hipMemcpyAsync(ctx->d_types, types, count * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        if (colors) O2V_CHECK(hipMemcpyAsync(ctx->d_colors, colors, count * 3 * sizeof(float), hipMemcpyHostToDevice, s));
        if (texids) O2V_CHECK(hipMemcpyAsync(ctx->d_texids, texids, count * sizeof(int32_t), hipMemcpyHostToDevice, s));
    }
    // (the host scan overlaps the copies)
    bool any_textured = false;
    if (types)
        for (uint64_t i = 0; i < count; ++i)
            if (types[i] == O2V_HIP_TRI_TEXTURED) {
                any_textured = true;
                break;
            }
    return o2v::ctx_finish_triangles(ctx, any_textured, nullptr);
}

}  // extern "C"

namespace {

constexpr uint64_t kStageTriangles = 1u << 16;  // per staging block: 2.25 MiB of vertices, 5 MiB with every optional array

// Makes room for `need` elements in a device array that already holds `have` valid ones (copied over if it has to move).
template <typename T>
int grow_keep(o2v_hip_ctx *ctx, T *&dptr, uint64_t &cap_bytes, uint64_t have, uint64_t need, uint64_t floor_elems)
{
    if (dptr && need * sizeof(T) <= cap_bytes) return O2V_HIP_OK;
    // (floor_elems: room for 2^20 triangles from the start, so that a streamed mesh does not pay for a chain of allocations
    // and device-to-device moves)
    const uint64_t want = std::max<uint64_t>(std::max<uint64_t>(need, 2 * (cap_bytes / sizeof(T))), floor_elems) * sizeof(T);
    T *bigger = nullptr;
    O2V_CHECK(hipMalloc(reinterpret_cast<void **>(&bigger), want));
    if (dptr) {
        if (have) O2V_CHECK(hipMemcpyAsync(bigger, dptr, have * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
        ctx->retired.push_back(dptr);  // freed once the stream has passed the copy (o2v_hip_end_triangles)
    }
    dptr = bigger;
    cap_bytes = want;
    return O2V_HIP_OK;
}

}  // namespace

extern "C" {

int o2v_hip_begin_triangles(o2v_hip_ctx *ctx, o2v_hip_staging *out_block)
{
    if (!ctx || !out_block) return O2V_HIP_ERR_BAD_ARGUMENT;
    O2V_CHECK(hipSetDevice(ctx->device));
    O2V_CHECK(hipStreamSynchronize(ctx->stream));
    if (!ctx->stage[0].verts) {
        for (int b = 0; b < 2; ++b) {
            o2v_hip_staging &st = ctx->stage[b];
            O2V_CHECK(hipHostMalloc(reinterpret_cast<void **>(&st.verts), kStageTriangles * 9 * sizeof(float), hipHostMallocDefault));
            st.capacity = kStageTriangles;
            O2V_CHECK(hipEventCreateWithFlags(&ctx->ev_stage[b], hipEventDisableTiming));
        }
    }
    ctx->stage_cur = 0;
    ctx->stream_count = 0;
    ctx->stream_arrays = 0;
    ctx->n_tris = 0;
    *out_block = ctx->stage[0];
    return O2V_HIP_OK;
}

int o2v_hip_stage_arrays(o2v_hip_ctx *ctx, uint32_t arrays, o2v_hip_staging *inout_block)
{
    if (!ctx || !inout_block || !ctx->stage[0].verts) return O2V_HIP_ERR_BAD_ARGUMENT;
    O2V_CHECK(hipSetDevice(ctx->device));
    for (o2v_hip_staging &st : ctx->stage) {
        if ((arrays & O2V_HIP_ARRAY_UVS) && !st.uvs)
            O2V_CHECK(hipHostMalloc(reinterpret_cast<void **>(&st.uvs), kStageTriangles * 6 * sizeof(float), hipHostMallocDefault));
        if ((arrays & O2V_HIP_ARRAY_TYPES) && !st.types)
            O2V_CHECK(hipHostMalloc(reinterpret_cast<void **>(&st.types), kStageTriangles * sizeof(uint32_t), hipHostMallocDefault));
        if ((arrays & O2V_HIP_ARRAY_COLORS) && !st.colors)
            O2V_CHECK(hipHostMalloc(reinterpret_cast<void **>(&st.colors), kStageTriangles * 3 * sizeof(float), hipHostMallocDefault));
        if ((arrays & O2V_HIP_ARRAY_TEXIDS) && !st.texids)
            O2V_CHECK(hipHostMalloc(reinterpret_cast<void **>(&st.texids), kStageTriangles * sizeof(int32_t), hipHostMallocDefault));
    }
    *inout_block = ctx->stage[ctx->stage_cur];
    return O2V_HIP_OK;
}

int o2v_hip_commit_triangles(o2v_hip_ctx *ctx, uint64_t count, uint32_t arrays, o2v_hip_staging *out_next_block)
{
    if (!ctx || !out_next_block || !ctx->stage[0].verts || count > kStageTriangles) return O2V_HIP_ERR_BAD_ARGUMENT;
    {
        const o2v_hip_staging &st = ctx->stage[ctx->stage_cur];
        if (((arrays & O2V_HIP_ARRAY_UVS) && !st.uvs) || ((arrays & O2V_HIP_ARRAY_TYPES) && !st.types) ||
            ((arrays & O2V_HIP_ARRAY_COLORS) && !st.colors) || ((arrays & O2V_HIP_ARRAY_TEXIDS) && !st.texids)) {
            ctx->err = "an optional triangle array was committed without o2v_hip_stage_arrays";
            return O2V_HIP_ERR_BAD_ARGUMENT;
        }
    }
    const uint64_t have = ctx->stream_count, need = have + count;
    if (need >= (1ull << 29)) {
        ctx->err = "triangle count must be below 2^29";
        return O2V_HIP_ERR_LIMIT;
    }
    O2V_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const o2v_hip_staging &st = ctx->stage[ctx->stage_cur];
    const uint32_t fresh = arrays & ~ctx->stream_arrays;  // arrays that appear with this block
    ctx->stream_arrays |= arrays;
    int rc;
    if ((rc = grow_keep(ctx, ctx->d_verts, ctx->cap_tri_bytes[0], have * 9, need * 9, 9ull << 20))) return rc;
    if (ctx->stream_arrays & O2V_HIP_ARRAY_UVS) {
        if ((rc = grow_keep(ctx, ctx->d_uvs, ctx->cap_tri_bytes[1], (fresh & O2V_HIP_ARRAY_UVS) ? 0 : have * 6, need * 6, 6ull << 20))) return rc;
        if ((fresh & O2V_HIP_ARRAY_UVS) && have) O2V_CHECK(hipMemsetAsync(ctx->d_uvs, 0, have * 6 * sizeof(float), s));
    }
    if (ctx->stream_arrays & O2V_HIP_ARRAY_TYPES) {
        if ((rc = grow_keep(ctx, ctx->d_types, ctx->cap_tri_bytes[2], (fresh & O2V_HIP_ARRAY_TYPES) ? 0 : have, need, 1ull << 20))) return rc;
        if ((fresh & O2V_HIP_ARRAY_TYPES) && have) O2V_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctx->d_types), (int) O2V_HIP_TRI_MATERIALLESS, have, s));
    }
    if (ctx->stream_arrays & O2V_HIP_ARRAY_COLORS) {
        if ((rc = grow_keep(ctx, ctx->d_colors, ctx->cap_tri_bytes[3], (fresh & O2V_HIP_ARRAY_COLORS) ? 0 : have * 3, need * 3, 3ull << 20))) return rc;
        if ((fresh & O2V_HIP_ARRAY_COLORS) && have) O2V_CHECK(hipMemsetAsync(ctx->d_colors, 0, have * 3 * sizeof(float), s));
    }
    if (ctx->stream_arrays & O2V_HIP_ARRAY_TEXIDS) {
        if ((rc = grow_keep(ctx, ctx->d_texids, ctx->cap_tri_bytes[4], (fresh & O2V_HIP_ARRAY_TEXIDS) ? 0 : have, need, 1ull << 20))) return rc;
        if ((fresh & O2V_HIP_ARRAY_TEXIDS) && have) O2V_CHECK(hipMemsetAsync(ctx->d_texids, 0, have * sizeof(int32_t), s));
    }
    if (count) {
        O2V_CHECK(hipMemcpyAsync(ctx->d_verts + have * 9, st.verts, count * 9 * sizeof(float), hipMemcpyHostToDevice, s));
        if (ctx->stream_arrays & O2V_HIP_ARRAY_UVS)
            O2V_CHECK(hipMemcpyAsync(ctx->d_uvs + have * 6, st.uvs, count * 6 * sizeof(float), hipMemcpyHostToDevice, s));
        if (ctx->stream_arrays & O2V_HIP_ARRAY_TYPES)
            O2V_CHECK(hipMemcpyAsync(ctx->d_types + have, st.types, count * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        if (ctx->stream_arrays & O2V_HIP_ARRAY_COLORS)
            O2V_CHECK(hipMemcpyAsync(ctx->d_colors + have * 3, st.colors, count * 3 * sizeof(float), hipMemcpyHostToDevice, s));
        if (ctx->stream_arrays & O2V_HIP_ARRAY_TEXIDS)
            O2V_CHECK(hipMemcpyAsync(ctx->d_texids + have, st.texids, count * sizeof(int32_t), hipMemcpyHostToDevice, s));
    }
    O2V_CHECK(hipEventRecord(ctx->ev_stage[ctx->stage_cur], s));
    ctx->stream_count = need;
    ctx->stage_cur ^= 1;
    O2V_CHECK(hipEventSynchronize(ctx->ev_stage[ctx->stage_cur]));  // the other block's copy (two commits ago) has landed
    *out_next_block = ctx->stage[ctx->stage_cur];
    return O2V_HIP_OK;
}

int o2v_hip_end_triangles(o2v_hip_ctx *ctx, uint32_t any_textured)
{
    if (!ctx) return O2V_HIP_ERR_BAD_ARGUMENT;
    O2V_CHECK(hipSetDevice(ctx->device));
    // optional arrays the mesh never used read as null in the kernels
    if (!(ctx->stream_arrays & O2V_HIP_ARRAY_UVS) && ctx->d_uvs) { ctx->retired.push_back(ctx->d_uvs); ctx->d_uvs = nullptr; ctx->cap_tri_bytes[1] = 0; }
    if (!(ctx->stream_arrays & O2V_HIP_ARRAY_TYPES) && ctx->d_types) { ctx->retired.push_back(ctx->d_types); ctx->d_types = nullptr; ctx->cap_tri_bytes[2] = 0; }
    if (!(ctx->stream_arrays & O2V_HIP_ARRAY_COLORS) && ctx->d_colors) { ctx->retired.push_back(ctx->d_colors); ctx->d_colors = nullptr; ctx->cap_tri_bytes[3] = 0; }
    if (!(ctx->stream_arrays & O2V_HIP_ARRAY_TEXIDS) && ctx->d_texids) { ctx->retired.push_back(ctx->d_texids); ctx->d_texids = nullptr; ctx->cap_tri_bytes[4] = 0; }
    ctx->n_tris = ctx->stream_count;
    ctx->tri_generation += 1;
    ctx->max_tri_extent = -1.f;
    const int rc = o2v::ctx_finish_triangles(ctx, any_textured != 0, nullptr);  // waits for the stream
    for (void *q : ctx->retired) (void) hipFree(q);
    ctx->retired.clear();
    return rc;
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
        O2V_CHECK(hipMalloc(reinterpret_cast<void **>(&d), bytes + 8));  // (+ 8: a texel is read as aligned 32-bit words, texel_ref)
        ctx->d_texpix.push_back(d);
        O2V_CHECK(hipMemset(d + bytes, 0, 8));
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
    if (S64 > 0x7fffffffull) {
        ctx->err = "sample resolution must be below 2^31";
        return O2V_HIP_ERR_LIMIT;
    }
    uint32_t z0 = params->z_begin, z1 = params->z_end;
    if (z0 == 0 && z1 == 0) z1 = params->resolution;
    if (z1 > params->resolution || z0 >= z1) {
        ctx->err = "z slab must satisfy z_begin < z_end <= resolution";
        return O2V_HIP_ERR_BAD_ARGUMENT;
    }
    // an x / y tile of the grid (0, 0: the whole axis), see o2v_hip_params::x_begin
    uint32_t xy0[2] = {params->x_begin, params->y_begin}, xy1[2] = {params->x_end, params->y_end};
    for (int k = 0; k < 2; ++k) {
        if (xy0[k] == 0 && xy1[k] == 0) xy1[k] = params->resolution;
        if (xy1[k] > params->resolution || xy0[k] >= xy1[k] || (xy0[k] & (kBrickX - 1u))) {
            ctx->err = "x / y tile must satisfy begin < end <= resolution, begin a multiple of 4";
            return O2V_HIP_ERR_BAD_ARGUMENT;
        }
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
    // the dense grids cover the mesh's voxel bounding box within the slab (grid_box)
    const GridBox box = grid_box(ctx, params, ss, z0, z1);   // (within the x / y tile, if the call names one)
    if (box.empty) return O2V_HIP_OK;  // the mesh does not reach this slab: no voxels
    // Voxel coordinates travel in 16-bit fields relative to the grid's origin (Params::so): what is limited is the box of one
    // pass, not the resolution.  (The reference carries u32 coordinates and 64-bit Morton keys, src/util.hpp:185-196; a box wider
    // than this is cut into x / y tiles by the caller - obj2voxel_voxelize() does - as a slab too thick for memory is cut in z.)
    for (int k = 0; k < 3; ++k)
        if ((uint64_t) (box.hi[k] - box.lo[k]) * ss > 65535u) {
            ctx->err = "the pass' box (the mesh's voxel bounding box within the slab / tile) must be at most 65535 samples wide; use x / y tiles (o2v_hip_params::x_begin ..) or z-slabs";
            return O2V_HIP_ERR_LIMIT;
        }
    p.xo0 = box.lo[0];
    p.yo0 = box.lo[1];
    p.NBx = (box.hi[0] - box.lo[0] + kBrickX - 1) / kBrickX;
    p.NBy = (box.hi[1] - box.lo[1] + kBrickY - 1) / kBrickY;
    const uint32_t NBz = (box.hi[2] - box.lo[2] + kBrickZ - 1) / kBrickZ;
    const uint64_t n_bricks = (uint64_t) p.NBx * p.NBy * NBz;
    if (n_bricks >= (1ull << 32) / 2) {
        ctx->err = "slab has too many bricks for 32-bit brick ids; use more z-slabs";
        return O2V_HIP_ERR_LIMIT;
    }
    p.n_bricks = (uint32_t) n_bricks;
    p.cap_dirty = (uint32_t) std::min<uint64_t>((n_bricks + 15u) & ~15ull, kDirtyListMax);
    p.ss_shift = ss == 2 ? 1u : 0u;
    p.zs0 = z0 * ss;
    p.zs1 = z1 * ss;
    p.zo0 = box.lo[2];
    for (int k = 0; k < 3; ++k) {
        // the same box in sample space, z cut to the slab: what a leaf's box is clamped to (plan_leaf)
        p.cs_lo[k] = box.lo[k] * ss;
        p.cs_hi[k] = (uint32_t) std::min<uint64_t>((uint64_t) box.hi[k] * ss, S64);
    }
    p.cs_lo[2] = std::max(p.cs_lo[2], p.zs0);
    p.cs_hi[2] = std::min(p.cs_hi[2], p.zs1);
    p.so[0] = p.xo0 * ss;
    p.so[1] = p.yo0 * ss;
    p.so[2] = p.zo0 * ss;
    p.blend = params->strategy;
    p.bounds_known = params->bounds_known;
    for (int i = 0; i < 6; ++i) p.bounds[i] = params->bounds[i];
    for (int i = 0; i < 9; ++i) p.unit[i] = params->unit_transform[i];
    p.has_uv = ctx->d_uvs ? 1u : 0u;
    const GridModes modes = grid_modes(ctx, params);
    p.exact_clip = modes.exact_clip ? 1u : 0u;
    ctx->ktimes_on = (params->flags & O2V_HIP_FLAG_KERNEL_TIMES) != 0;
    ctx->stage_events = (params->flags & (O2V_HIP_FLAG_STAGE_TIMES | O2V_HIP_FLAG_KERNEL_TIMES)) != 0;
    ctx->kernel_times.clear();
    const bool use_uv = modes.use_uv;
    ctx->sorted_stride = use_uv ? 6u : 4u;

    // Dense grids for the slab (bricked, see cell_index), each with one dirty flag per brick; allocated zeroed, kept clean by
    // the scan / reset / emission kernels.  Which ones a run needs depends on the mesh and the strategy:
    //   occupancy-only (no triangle has a material: every STL, an OBJ without materials)   1 byte per cell
    //   MAX strategy                       64-bit max grid (direct path) + 32-bit counter grid (subdivided triangles)
    //   BLEND strategy                     32-bit counter grid
    const uint64_t cells = n_bricks * kBrickCells;
    ctx->stats.grid_cells = cells;
    ctx->stats.grid_bytes = 0;
    const uint64_t brick_cap_want = (n_bricks + 15u) & ~15ull;
    p.occupancy_only = modes.occupancy_only ? 1u : 0u;
    {
        const char *no_bypass = std::getenv("O2V_NO_ROOT_BYPASS");  // (A/B: every leaf through its Leaf / Tile records)
        p.root_bypass = (modes.occupancy_only && !(no_bypass && no_bypass[0] == '1')) ? 1u : 0u;
    }
    // ... and on a tessellated surface - fewer than one triangle in 512 is 4 voxels or more across (the histogram of the triangles'
    // extents, made at upload): practically every block of 256 holds nothing but one-tile leaves - a z-slab run lets k_count_roots
    // run ahead of k_expand_roots (run_pass).  Measured: the expand stage of a slab of the 8-GPU weak job 0.054 -> 0.034 ms; on the
    // whole grid (the bench headline) the two kernels take what k_expand_roots alone takes (0.028 against 0.026 ms), so not there.
    // Either way the same leaves are made; O2V_NO_COUNT_ROOTS=1: never, O2V_COUNT_ROOTS=1: also on the whole grid (A/B).
    ctx->lean_roots = false;
    const char *lean_always = std::getenv("O2V_COUNT_ROOTS");
    if (p.root_bypass && ctx->max_tri_extent >= 0.f && ctx->n_tris && (z0 != 0 || z1 < params->resolution || (lean_always && lean_always[0] == '1'))) {
        const float *b = params->bounds_known ? params->bounds : ctx->mesh_bounds_hint;
        const float max_axis = std::max(b[3] - b[0], std::max(b[4] - b[1], b[5] - b[2]));
        float unit_norm = 0.f;
        for (int i = 0; i < 3; ++i)
            unit_norm = std::max(unit_norm, std::fabs((float) p.unit[i * 3]) + std::fabs((float) p.unit[i * 3 + 1]) + std::fabs((float) p.unit[i * 3 + 2]));
        const float voxels_per_unit = unit_norm * (float) p.S / max_axis;
        const char *off = std::getenv("O2V_NO_COUNT_ROOTS");
        if (max_axis > 0.f && std::isfinite(voxels_per_unit) && voxels_per_unit > 0.f && !(off && off[0] == '1')) {
            uint64_t large = ctx->ext_hist[255];
            for (uint32_t e = 1; e < 255; ++e)   // bin e: extents in [2^(e-127), 2^(e-126))
                if (std::ldexp(1.0f, (int) e - 127) * voxels_per_unit >= 4.0f) large += ctx->ext_hist[e];
            ctx->lean_roots = large * 512u <= ctx->n_tris;
        }
    }
    p.direct_max = modes.direct_max ? 1u : 0u;
    p.pick_max = (p.direct_max && use_uv) ? 1u : 0u;  // textured: the winner's colour is picked afterwards (k_pick)
    p.mat = Materials{ctx->d_types, ctx->d_colors, ctx->d_texids, ctx->d_textures, ctx->n_textures};
    if (p.direct_max) {
        // the max grid: 8 bytes per cell, or - occupancy only - the same buffer used as 1 byte per cell
        const uint64_t want_bytes = cells * (p.occupancy_only ? 1ull : sizeof(unsigned long long));
        if (want_bytes > ctx->maxgrid_bytes || n_bricks > ctx->maxgrid_brick_cap || !ctx->d_maxgrid) {
            for (void *q : {(void *) ctx->d_maxgrid, (void *) ctx->d_dirty_max, (void *) ctx->d_dirty_list_max})
                if (q) O2V_CHECK(hipFree(q));
            ctx->d_maxgrid = nullptr;
            ctx->d_dirty_max = nullptr;
            ctx->d_dirty_list_max = nullptr;
            ctx->maxgrid_bytes = 0;
            ctx->maxgrid_brick_cap = 0;
            // if it does not fit (8 bytes per cell at 4096^3 on one GPU), every hit takes the sort-and-replay route instead.
            // All three buffers exist before the capacity is published: a context (cached process-wide by the C API) whose later
            // allocation failed must not look ready.
            const bool ok = hipMalloc(reinterpret_cast<void **>(&ctx->d_maxgrid), want_bytes) == hipSuccess &&
                            hipMalloc(reinterpret_cast<void **>(&ctx->d_dirty_max), brick_cap_want) == hipSuccess &&
                            hipMalloc(reinterpret_cast<void **>(&ctx->d_dirty_list_max), std::min<uint64_t>(brick_cap_want, kDirtyListMax) * sizeof(uint32_t)) == hipSuccess;
            if (!ok) {
                (void) hipGetLastError();
                for (void *q : {(void *) ctx->d_maxgrid, (void *) ctx->d_dirty_max, (void *) ctx->d_dirty_list_max})
                    if (q) (void) hipFree(q);
                ctx->d_maxgrid = nullptr;
                ctx->d_dirty_max = nullptr;
                ctx->d_dirty_list_max = nullptr;
                p.direct_max = 0;
                p.pick_max = 0;
                p.occupancy_only = 0;
                p.root_bypass = 0;
            }
            else {
                ctx->maxgrid_bytes = want_bytes;
                ctx->maxgrid_brick_cap = brick_cap_want;
                ctx->maxgrid_map_bytes = brick_cap_want;
                ctx->maxgrid_dirty = true;
            }
        }
    }
    if (p.direct_max) {
        if (ctx->maxgrid_dirty) {
            O2V_CHECK(hipMemsetAsync(ctx->d_maxgrid, 0, ctx->maxgrid_bytes, ctx->stream));
            O2V_CHECK(hipMemsetAsync(ctx->d_dirty_max, 0, ctx->maxgrid_map_bytes, ctx->stream));
            ctx->maxgrid_dirty = false;
        }
        p.maxgrid = ctx->d_maxgrid;
        p.occgrid = reinterpret_cast<uint8_t *>(ctx->d_maxgrid);
        p.dirty_max = ctx->d_dirty_max;
        ctx->stats.grid_bytes += cells * (p.occupancy_only ? 1ull : sizeof(unsigned long long)) + n_bricks;
    }
    if (!p.occupancy_only) {
        // the counter grid (no pooled hits exist in occupancy-only mode)
        ctx->stats.grid_bytes += cells * sizeof(uint32_t) + n_bricks;
        if (cells > ctx->grid_cells || !ctx->d_grid) {
            for (void *q : {(void *) ctx->d_grid, (void *) ctx->d_brick_dirty, (void *) ctx->d_dirty_list})
                if (q) O2V_CHECK(hipFree(q));
            ctx->d_grid = nullptr;
            ctx->d_brick_dirty = nullptr;
            ctx->d_dirty_list = nullptr;
            ctx->grid_cells = 0;
            O2V_CHECK(hipMalloc(reinterpret_cast<void **>(&ctx->d_grid), cells * sizeof(uint32_t)));
            ctx->grid_cells = cells;
            ctx->brick_cap = brick_cap_want;
            O2V_CHECK(hipMalloc(reinterpret_cast<void **>(&ctx->d_brick_dirty), ctx->brick_cap));
            O2V_CHECK(hipMalloc(reinterpret_cast<void **>(&ctx->d_dirty_list), std::min<uint64_t>(ctx->brick_cap, kDirtyListMax) * sizeof(uint32_t)));
            if (ctx->d_brick_slab) O2V_CHECK(hipFree(ctx->d_brick_slab));
            ctx->d_brick_slab = nullptr;
            O2V_CHECK(hipMalloc(reinterpret_cast<void **>(&ctx->d_brick_slab), ctx->brick_cap * sizeof(uint32_t)));
            ctx->grid_dirty = true;
        }
        if (ctx->grid_dirty) {
            O2V_CHECK(hipMemsetAsync(ctx->d_grid, 0, ctx->grid_cells * sizeof(uint32_t), ctx->stream));
            O2V_CHECK(hipMemsetAsync(ctx->d_brick_dirty, 0, ctx->brick_cap, ctx->stream));
            ctx->grid_dirty = false;
        }
    }
    if (ctx->n_tris == 0) return O2V_HIP_OK;  // empty mesh: empty model (obj2voxel.cpp:590-594)
    if (!ctx->d_jobq)
        O2V_CHECK(hipMalloc(reinterpret_cast<void **>(&ctx->d_jobq), (size_t) ctx->num_cus * (size_t) (O2V_K2_WAVES > O2V_K2_WAVES_UV ? O2V_K2_WAVES : O2V_K2_WAVES_UV) * (kBlock / 64u) * (64u * 64u) * sizeof(uint2) * 2u));  // = workgroups x VoxShape::queue for every shape; twice that for k_voxelize_occ

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
    // Hit slabs: one per listed brick (about 1.3 x the bricks that end up holding voxels on a tessellated surface).  They are a
    // budget, not a requirement - a listed brick beyond cap_slabs pools all its hits - so a pass is never repeated for them:
    // the capacity follows the last pass's list (ctx->want_slabs_next) up to a sixth of the device memory.
    const uint32_t slab_stride = use_uv ? 6u : 4u;
    if (ctx->slabs_stride != slab_stride) {
        ctx->cap_slabs = 0;  // (the records' size changed: the allocation is counted in slabs of the new size)
        if (ctx->d_slabs) O2V_CHECK(hipFree(ctx->d_slabs));
        ctx->d_slabs = nullptr;
        ctx->slabs_stride = slab_stride;
        ctx->slabs_wanted_at_grant = 0;
    }
    uint64_t want_slabs = 0;
    if (!p.occupancy_only) {
        size_t free_b = 0, total_b = 0;
        O2V_CHECK(hipMemGetInfo(&free_b, &total_b));
        const uint64_t slab_bytes = (uint64_t) kInlineHits * kBrickCells * slab_stride * sizeof(uint32_t);
        const uint64_t budget = std::max<uint64_t>(total_b / 6 / slab_bytes, 1);
        want_slabs = std::max<uint64_t>(ctx->want_slabs_next, ctx->n_tris / 2 + (1u << 14));
        want_slabs = std::min<uint64_t>(std::min<uint64_t>(want_slabs, n_bricks), budget);
        want_slabs = std::max<uint64_t>(want_slabs, ctx->cap_slabs);
        if (const char *tiny = std::getenv("O2V_TEST_TINY_BUFFERS"); tiny && tiny[0] == '1') want_slabs = std::max<uint64_t>(ctx->cap_slabs, 4);
        if (const char *no = std::getenv("O2V_NO_SLABS"); no && no[0] == '1') want_slabs = ctx->cap_slabs;  // (A/B: every hit pooled)
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
    ctx->skip_big = false;
    bool solo_ok = false;
    if (ctx->max_tri_extent >= 0.f) {
        // tighter: a (sub-)triangle whose extent is at most 5 voxels has a voxel AABB of at most 7^3 < 512 cells and is
        // a leaf; every round halves the extents.  Scale = the mesh transform's (obj2voxel.cpp:370-402).
        const float *b = params->bounds_known ? params->bounds : ctx->mesh_bounds_hint;
        const float max_axis = std::max(b[3] - b[0], std::max(b[4] - b[1], b[5] - b[2]));
        float unit_norm = 0.f;
        for (int i = 0; i < 3; ++i)
            unit_norm = std::max(unit_norm, std::fabs((float) p.unit[i * 3]) + std::fabs((float) p.unit[i * 3 + 1]) + std::fabs((float) p.unit[i * 3 + 2]));
        // (the largest triangle's extent in voxels, rounded up a little; its voxel box has at most extent + 2 cells per axis)
        const float ext_vox = ctx->max_tri_extent * unit_norm * ((float) p.S / max_axis) * 1.0001f + 1e-3f;
        if (max_axis > 0.f && ext_vox == ext_vox && ext_vox < 3.0e9f) {
            // a (sub-)triangle less than 6 voxels across has a box of fewer than 8^3 = 512 cells and is a leaf: a mesh of such
            // triangles needs no subdivision round at all (most tessellated surfaces at their resolution), one `depth` halvings
            // larger needs `depth` rounds
            uint32_t depth = 0;
            for (float e = ext_vox; e > 5.9f; e *= 0.5f) ++depth;
            n_rounds = std::min<uint32_t>(n_rounds, depth);
            // ... and a leaf less than 7.9 voxels across has fewer than 10^3 cells = four tiles: none for k_expand_big (a larger
            // one can only be an axis-aligned triangle, voxelization.cpp:335-347)
            ctx->skip_big = ext_vox < 7.9f;
            solo_ok = ext_vox < 4.99f;
        }
    }
    if (const char *all = std::getenv("O2V_ALL_LAUNCHES"); all && all[0] == '1') {  // (A/B: no launch left out on the strength of the hints)
        n_rounds = std::max<uint32_t>(n_rounds, 1u);
        ctx->skip_big = false;
        solo_ok = false;
    }
    // Occupancy only, every triangle less than 5 voxels across (the largest extent, known since the upload, at this call's scale):
    // its voxel box has at most 6 cells per axis - 216: below the subdivision limit of 512 (voxelization.cpp:488-511) and one tile -
    // so every root triangle is a leaf of one tile or misses the slab, and k_expand_roots (K1) would write nothing: it is not
    // launched, k_voxelize_occ makes the leaves (as with root_bypass) and counts them.  The kernel checks the premise per triangle
    // (kErrSoloRoots); should it ever fail, the pass is repeated with K1 and this mesh keeps it.
    const uint64_t solo_key = ctx->tri_generation * 1000003ull + p.S * 131ull + p.zs0 * 31ull + p.zs1 + 1u;
    if (const char *force = std::getenv("O2V_TEST_FORCE_SOLO_ROOTS"); force && force[0] == '1') solo_ok = true;  // test hook: whatever the hint says
    ctx->solo_roots = p.root_bypass && solo_ok && ctx->solo_refused_key != solo_key;
    if (const char *off = std::getenv("O2V_NO_SOLO_ROOTS"); off && off[0] == '1') ctx->solo_roots = false;
    p.solo_roots = ctx->solo_roots ? 1u : 0u;
    ctx->force_general = false;
    ctx->mark_missing = false;  // (a call that ended early - an error, a failed allocation - must not leave it to the next one)
    if (!p.occupancy_only) ctx->grid_dirty = true;  // until a pass completes (the scan / reset kernels leave it clean)
    if (p.direct_max) ctx->maxgrid_dirty = true;
    for (uint32_t pass = 1; pass <= 12; ++pass) {
        int rc;
        if (pass > 1) {
            // a pass that overflowed a buffer may have left counters / offsets in cells it could not list
            if (!p.occupancy_only) {
                O2V_CHECK(hipMemsetAsync(ctx->d_grid, 0, ctx->grid_cells * sizeof(uint32_t), ctx->stream));
                O2V_CHECK(hipMemsetAsync(ctx->d_brick_dirty, 0, ctx->brick_cap, ctx->stream));
            }
            if (p.direct_max) {
                O2V_CHECK(hipMemsetAsync(ctx->d_maxgrid, 0, ctx->maxgrid_bytes, ctx->stream));
                O2V_CHECK(hipMemsetAsync(ctx->d_dirty_max, 0, ctx->maxgrid_map_bytes, ctx->stream));
            }
        }
        // The buffers a pass cannot do without come first; the hit slabs - a budget, the pass runs without them - take what is
        // left afterwards, and give way (are freed, then the allocation is tried again) if one of the others does not fit.
        auto grow_required = [&](auto *&ptr, uint32_t &cap, uint64_t want) -> int {
            int rc_g = grow(ctx, ptr, cap, want);
            if (rc_g == O2V_HIP_ERR_OUT_OF_MEMORY && ctx->d_slabs) {
                (void) hipGetLastError();
                (void) hipFree(ctx->d_slabs);
                ctx->d_slabs = nullptr;
                ctx->cap_slabs = 0;
                want_slabs = 0;  // (this call goes on without slabs: every hit is pooled)
                rc_g = grow(ctx, ptr, cap, want);
            }
            return rc_g;
        };
        if ((rc = grow_required(ctx->d_leaves, ctx->cap_leaves, want_leaves))) return rc;
        if ((rc = grow_required(ctx->d_tiles, ctx->cap_tiles, want_tiles))) return rc;
        if ((rc = grow_required(ctx->d_big, ctx->cap_big, want_big))) return rc;
        if (ctx->lean_roots && (rc = grow_required(ctx->d_need_list, ctx->cap_need_list, (ctx->n_tris + kBlock - 1) / kBlock))) return rc;
        uint32_t cap_n0 = ctx->cap_nodes, cap_n1 = ctx->cap_nodes;
        if ((rc = grow_required(ctx->d_nodes[0], cap_n0, want_nodes))) return rc;
        if ((rc = grow_required(ctx->d_nodes[1], cap_n1, want_nodes))) return rc;
        ctx->cap_nodes = cap_n0;
        {
            uint32_t cap_p = ctx->cap_hits, cap_s = ctx->cap_hits;
            if ((rc = grow_required(ctx->d_pool, cap_p, want_hits))) return rc;
            if ((rc = grow_required(ctx->d_sorted, cap_s, want_hits))) return rc;
            ctx->cap_hits = cap_p;
        }
        uint32_t cap_v0 = ctx->cap_vox, cap_v1 = ctx->cap_vox;
        if ((rc = grow_required(ctx->d_occ, cap_v0, want_vox))) return rc;
        if ((rc = grow_required(ctx->d_out, cap_v1, want_vox))) return rc;
        if (p.pick_max) {
            uint32_t cap_px = ctx->cap_pick_extra;
            if ((rc = grow_required(ctx->d_pick_extra, cap_px, want_vox))) return rc;
            ctx->cap_pick_extra = cap_px;
            p.pick_extra = reinterpret_cast<uint32_t *>(ctx->d_pick_extra);
        }
        for (uint32_t **lp : {&ctx->d_list_lane8, &ctx->d_list_lane16, &ctx->d_list_w64, &ctx->d_list_lane, &ctx->d_list_mid, &ctx->d_list_long, &ctx->d_list_big, &ctx->d_list_huge}) {
            uint32_t cap_l = ctx->cap_vox;
            if ((rc = grow_required(*lp, cap_l, want_vox))) return rc;
        }
        ctx->cap_vox = cap_v0;
        if (want_scratch) {
            uint32_t cap_s0 = ctx->cap_scratch, cap_s1 = ctx->cap_scratch;
            if ((rc = grow_required(ctx->d_scratch_key, cap_s0, want_scratch))) return rc;
            if ((rc = grow_required(ctx->d_scratch_idx, cap_s1, want_scratch))) return rc;
            ctx->cap_scratch = cap_s0;
        }
        // (a grant below what was asked for - the memory was short - is kept until more is asked for than then: asking again
        // with every call would free and allocate the slabs every time)
        if (want_slabs > ctx->cap_slabs && !(ctx->cap_slabs && want_slabs <= ctx->slabs_wanted_at_grant)) {
            ctx->slabs_wanted_at_grant = want_slabs;
            if (ctx->d_slabs) O2V_CHECK(hipFree(ctx->d_slabs));
            ctx->d_slabs = nullptr;
            ctx->cap_slabs = 0;
            // (what is free now, every required buffer being in place, less 1 GiB for what a later pass may have to grow)
            size_t free_now = 0, total_now = 0;
            O2V_CHECK(hipMemGetInfo(&free_now, &total_now));
            const uint64_t slab_bytes = (uint64_t) kInlineHits * kBrickCells * slab_stride * sizeof(uint32_t);
            const uint64_t room = free_now > (1ull << 30) ? ((uint64_t) free_now - (1ull << 30)) / slab_bytes : 0ull;
            const uint64_t n_slabs_now = std::min<uint64_t>(std::min<uint64_t>(want_slabs, room), 0xfffffff0ull);
            if (n_slabs_now && hipMalloc(reinterpret_cast<void **>(&ctx->d_slabs), n_slabs_now * slab_bytes) == hipSuccess)
                ctx->cap_slabs = (uint32_t) n_slabs_now;
            else
                (void) hipGetLastError();  // (no slabs: every hit is pooled)
            want_slabs = ctx->cap_slabs;
        }
        p.cap_leaves = ctx->cap_leaves;
        p.cap_tiles = ctx->cap_tiles;
        p.cap_big = ctx->cap_big;
        p.cap_nodes = ctx->cap_nodes;
        p.cap_hits = ctx->cap_hits;
        p.cap_vox = ctx->cap_vox;
        p.cap_slabs = ctx->cap_slabs;
        p.slab_stride = slab_stride;
        p.slabs = ctx->d_slabs;
        p.brick_slab = ctx->d_brick_slab;

        if ((rc = run_pass(ctx, p, use_uv, n_rounds))) return rc;
        const Counters &h = *ctx->h_ctr;
        ctx->timings.passes = pass;
        if (p.solo_roots && (h.err_flags & kErrSoloRoots)) {
            // a root triangle that is k_expand_roots' business although the hint ruled that out: the pass again, with K1
            ctx->solo_refused_key = solo_key;
            ctx->solo_roots = false;
            p.solo_roots = 0;
            continue;
        }
        if (h.err_flags) {
            // (a dirty-list overflow leaves bricks behind that no list names: the grids stay marked for a full clear)
            // (nor does a pass that ran without a brick list - mark_missing - clean up behind itself)
            if (!(h.err_flags & kErrDirtyList) && !p.occupancy_only && ctx->last_ran_general && ctx->marked_bricks) ctx->grid_dirty = false;
            ctx->mark_missing = false;
            ctx->err = (h.err_flags & kErrLeafTooLarge) ? "a leaf's voxel AABB has 2^32 or more candidate voxels"
                       : (h.err_flags & kErrDepth)      ? "subdivision deeper than 15 levels"
                       : (h.err_flags & kErrDirtyList)  ? "more than 2^27 bricks of the slab hold voxels; use more z-slabs"
                       : (h.err_flags & kErrCounterWrap) ? "2^32 or more leaves or tiles in one slab; use more z-slabs"
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
        need(h.n_sorted, ctx->cap_hits, want_hits);  // (the sorted array also holds what the slabs held of the crowded cells)
        if (!p.occupancy_only) ctx->want_slabs_next = std::max<uint64_t>(ctx->want_slabs_next, (uint64_t) h.n_dirty + h.n_dirty / 8 + 64);
        need(h.n_vox, ctx->cap_vox, want_vox);
        if (p.direct_max) need(h.n_out, ctx->cap_vox, want_vox);
        if (n_rounds < kMaxRounds && h.n_nodes[n_rounds] != 0) {
            n_rounds = kMaxRounds;  // unusually deep subdivision (or the hint about the largest triangle did not hold)
            again = true;
        }
        if (ctx->skip_big && h.n_big != 0) {
            ctx->skip_big = false;  // a leaf of more than four tiles although the hint ruled that out: with k_expand_big, then
            again = true;
        }
        if (ctx->mark_missing) {
            // (run_pass left k_mark_bricks out on the strength of the last pass and K1's counters then asked for the general route)
            ctx->mark_missing = false;
            again = true;
        }
        if (!again && !ctx->last_ran_general && h.n_hits != h.n_direct) {
            // The stages behind k_voxelize were chosen from K1's counters alone, on the premise that only leaves of subdivided
            // triangles are pooled (order key 0 <=> unsplit triangle, a convention of k_expand_*).  Pooled hits exist although
            // the sort + replay stages were skipped: run the pass again with them.
            ctx->force_general = true;
            again = true;
        }
        if (!again && h.n_huge && (!ctx->d_scratch_key || h.scratch_used > ctx->cap_scratch)) {
            // some cell holds more than kLongList hits: the global-memory sort tier needs its scratch area
            want_scratch = std::max<uint64_t>(2ull * ctx->cap_hits, (uint64_t) h.scratch_used + 1024);
            again = true;
        }
        if (!again && ctx->last_ran_general && (uint32_t) (h.n_hits - h.n_direct) != h.n_listed_hits) {
            // k_voxelize counted a hit into a cell of a brick that k_mark_bricks did not list (the two must agree on the leaf's
            // clamped box, supersampling shift and slab origin): its voxel would be missing and its counter would stay behind
            // for the next run.  Never seen; checked because nothing else would notice.  The grids are cleared before the next call.
            ctx->grid_dirty = true;
            ctx->maxgrid_dirty = true;
            ctx->err = "internal error: hits outside the listed bricks (" + std::to_string(h.n_hits - h.n_direct) + " counted, " +
                       std::to_string(h.n_listed_hits) + " listed)";
            return O2V_HIP_ERR_HIP;
        }
        if (!again) {
            if (!p.occupancy_only) ctx->grid_dirty = false;
            ctx->maxgrid_dirty = false;
            const bool direct = p.direct_max && (p.occupancy_only || h.n_nodes[0] <= h.n_root_leaves);  // direct_active() on the device
            const uint64_t n_final = direct ? h.n_out : h.n_vox;
            ctx->last_direct = direct;
            ctx->n_vox = n_final;
            ctx->stats.leaves = h.n_leaves + h.n_bypass;  // (root_bypass: leaves of one tile that k_voxelize_occ made itself)
            ctx->stats.tiles = h.n_tiles + h.n_bypass;
            ctx->stats.candidates = h.n_candidates;
            ctx->stats.hits = h.n_hits;
            ctx->stats.voxels = n_final;
            ctx->stats.direct_hits = h.n_direct;
            ctx->stats.jobs = h.n_jobs;
            ctx->stats.certain_hits = h.n_certain;
            ctx->stats.skipped_jobs = h.n_jobs_skipped;
            ctx->stats.bypassed_leaves = h.n_bypass;
            ctx->stats.bricks = p.n_bricks;
            ctx->stats.dirty_bricks = direct ? h.n_dirty_max : h.n_dirty;
            ctx->stats.pool_slots = h.n_hits_reserved;
            std::memcpy(ctx->xform, h.xform, sizeof(ctx->xform));
            for (int i = 0; i < 16; ++i) ctx->dbg[i] = h.dbg[i];
            float ms[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            if (ctx->stage_events) {
                for (int i = 0; i < 5; ++i) O2V_CHECK(hipEventElapsedTime(&ms[i], ctx->ev[i], ctx->ev[i + 1]));
                O2V_CHECK(hipEventElapsedTime(&ctx->timings.total_ms, ctx->ev[0], ctx->ev[5]));
            }
            else O2V_CHECK(hipEventElapsedTime(&ms[2], ctx->ev[2], ctx->ev[3]));  // (k_voxelize's own dispatch: O2V_LAUNCH_K2)
            ctx->timings.bounds_ms = ms[0];
            ctx->timings.expand_ms = ms[1];
            ctx->timings.voxelize_ms = ms[2];
            ctx->timings.scan_ms = ms[3];
            ctx->timings.resolve_ms = ms[4];
            if (out_voxel_count) *out_voxel_count = ctx->n_vox;
            return O2V_HIP_OK;
        }
    }
    ctx->err = "device buffers did not converge after 12 passes";
    return O2V_HIP_ERR_LIMIT;
}

}  // extern "C"

namespace {

// The triangle passes of the slab plan over the triangles [tri_begin, tri_end) - this rank's share; a single GPU takes the
// whole list: mesh bounds (unless given: reference findMeshBounds, src/obj2voxel.cpp:180-200), transform, the z
// histogram of predicted work and the z extent of every block of 256 triangles.  With a communicator the partial results
// are combined over the ranks: min / max of the bounds, sum of the histogram, all-gather of the block extents
// (`blocks_per_rank` blocks each).  Afterwards the histogram is in ctx->h_zhist and the counters in ctx->h_ctr.
// `bounds_reduced`: the counters already hold the bounds of the whole mesh (o2v_hip_voxelize_sharded reduces them together with
// the ranks' readiness word); else they are computed - and, with a communicator, reduced - here.
// The collectives are timed (collective_ms, parts_ms: two events and a wait for each) only in a call with
// O2V_HIP_FLAG_STAGE_TIMES: each wait is a round trip to the host that the step otherwise does not make.
int plan_passes(o2v_hip_ctx *ctx, const o2v_hip_params *params, uint64_t tri_begin, uint64_t tri_end, o2v_hip_comm *comm,
                uint64_t blocks_per_rank, uint32_t &n_bins, uint32_t &bin_out, float *collective_ms, float *parts_ms = nullptr,
                bool bounds_reduced = false)
{
    const uint32_t ss = params->supersampling ? params->supersampling : 1u;
    const uint32_t G = params->resolution;
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
    for (int k = 0; k < 3; ++k) p.cs_hi[k] = p.S;   // (the planning passes see the whole grid: no crop, no tile)
    p.bounds_known = params->bounds_known;
    for (int i = 0; i < 6; ++i) p.bounds[i] = params->bounds[i];
    for (int i = 0; i < 9; ++i) p.unit[i] = params->unit_transform[i];
    // (what a leaf costs beside its hits, in hit equivalents: k_zhist)
    p.plan_leaf_cost = grid_modes(ctx, params).occupancy_only ? kPlanLeafCostOccupancy : kPlanLeafCost;
    // sample layers per bin: a whole number of output layers, at most kPlanBins bins
    bin_out = (G + kPlanBins - 1) / kPlanBins;
    n_bins = (G + bin_out - 1) / bin_out;
    const uint64_t n_range = tri_end > tri_begin ? tri_end - tri_begin : 0;

    hipStream_t s = ctx->stream;
    float coll_ms = 0.f;
    const bool measure = (params->flags & (O2V_HIP_FLAG_STAGE_TIMES | O2V_HIP_FLAG_KERNEL_TIMES)) != 0 && ctx->ev_coll[0] && ctx->ev_coll[1];
    auto timed = [&](int part, auto &&collectives) -> int {
        if (!measure) return collectives();
        O2V_CHECK(hipEventRecord(ctx->ev_coll[0], s));
        const int rc = collectives();
        if (rc) return rc;
        O2V_CHECK(hipEventRecord(ctx->ev_coll[1], s));
        O2V_CHECK(hipEventSynchronize(ctx->ev_coll[1]));
        float ms = 0.f;
        O2V_CHECK(hipEventElapsedTime(&ms, ctx->ev_coll[0], ctx->ev_coll[1]));
        coll_ms += ms;
        if (parts_ms) parts_ms[part] += ms;
        return O2V_HIP_OK;
    };
    auto comm_failed = [&](int rc) {
        ctx->err = std::string("collective failed: ") + comm->err;
        return rc;
    };
    ctx->ctr_clean = false;
    if (!bounds_reduced) hipLaunchKernelGGL(k_init, dim3(1), dim3(64), 0, s, ctx->d_ctr, kPassCounterWords);
    if (!p.bounds_known && !bounds_reduced) {
        if (n_range)
            hipLaunchKernelGGL(k_bounds, dim3((uint32_t) std::min<uint64_t>((uint64_t) ctx->num_cus, (n_range * 9 / 12 + kBoundsBlock) / kBoundsBlock)),
                               dim3(kBoundsBlock), 0, s, ctx->d_verts + tri_begin * 9, n_range * 9, ctx->d_ctr);
        O2V_STAGE("k_bounds");
        if (comm) {
            int rc = timed(1, [&]() -> int {
                int r = comm->allreduce_min_u32(ctx->d_ctr->bounds_enc, 3, s);
                if (!r) r = comm->allreduce_max_u32(ctx->d_ctr->bounds_enc + 3, 3, s);
                return r;
            });
            if (rc) return comm_failed(rc);
        }
    }
    hipLaunchKernelGGL(k_setup, dim3(1), dim3(64), 0, s, ctx->d_ctr, p);
    O2V_CHECK(hipMemsetAsync(ctx->d_zhist, 0, kPlanBins * sizeof(unsigned long long), s));
    {
        const uint64_t n_blocks = (p.n_tris + kBlock - 1) / kBlock;
        int rc;
        if ((rc = grow(ctx, ctx->d_zrange, ctx->cap_zrange, std::max<uint64_t>(comm ? blocks_per_rank * (uint64_t) comm->world : n_blocks, 1))))
            return rc;
        if (!ctx->d_zrange_xform) O2V_CHECK(hipMalloc(reinterpret_cast<void **>(&ctx->d_zrange_xform), 12 * sizeof(float)));
        if ((rc = grow(ctx, ctx->d_block_list, ctx->cap_block_list, std::max<uint64_t>(n_blocks, 1)))) return rc;
    }
    ctx->zrange_generation = ~0ull;
    hipLaunchKernelGGL(k_zhist, dim3((uint32_t) std::max<uint64_t>(1, std::min<uint64_t>((uint64_t) ctx->num_cus * 6u, (n_range + kBlock - 1) / kBlock))),
                       dim3(kBlock), 0, s, ctx->d_verts, ctx->d_ctr, ctx->d_zhist, ctx->d_zrange, ctx->d_zrange_xform, p,
                       bin_out * ss, tri_begin, tri_end);
    O2V_STAGE("k_zhist");
    if (comm) {
        // one all-gather for both: every rank's partial histogram and the extents of its blocks (k_pack_plan), summed / put in
        // place by every rank itself (an all-reduce and an all-gather, one after the other, cost a collective's latency more)
        const uint64_t rec_words = (uint64_t) kPlanBins + blocks_per_rank;
        int rc;
        if ((rc = grow(ctx, ctx->d_plan_gather, ctx->cap_plan_gather, rec_words * (uint64_t) comm->world))) return rc;  // (sized by the caller already)
        const uint32_t wgs = (uint32_t) std::min<uint64_t>((uint64_t) ctx->num_cus * 2u, (rec_words * (uint64_t) comm->world + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(k_pack_plan, dim3(std::max(1u, std::min<uint32_t>(wgs, (uint32_t) ((rec_words + kBlock - 1) / kBlock)))), dim3(kBlock), 0, s,
                           ctx->d_zhist, ctx->d_zrange + (uint64_t) comm->rank * blocks_per_rank,
                           ctx->d_plan_gather + (uint64_t) comm->rank * rec_words, (uint32_t) blocks_per_rank);
        rc = timed(2, [&]() -> int { return comm->allgather(ctx->d_plan_gather, rec_words * sizeof(unsigned long long), s); });
        if (rc) return comm_failed(rc);
        hipLaunchKernelGGL(k_unpack_plan, dim3(std::max(1u, wgs)), dim3(kBlock), 0, s, ctx->d_plan_gather, (uint32_t) comm->world,
                           (uint32_t) blocks_per_rank, ctx->d_zhist, ctx->d_zrange);
    }
    O2V_CHECK(hipMemcpyAsync(ctx->h_zhist, ctx->d_zhist, n_bins * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    O2V_CHECK(hipMemcpyAsync(ctx->h_ctr, ctx->d_ctr, kPassCounterWords * 4u, hipMemcpyDeviceToHost, s));
    O2V_CHECK(hipStreamSynchronize(s));
    O2V_CHECK(hipGetLastError());
    ctx->zrange_generation = ctx->tri_generation;  // k_expand_roots may use the extents (it checks the transform)
    if (collective_ms) *collective_ms = coll_ms;
    return O2V_HIP_OK;
}

// Cuts the histogram into n_slabs parts of equal predicted work (whole bins; every slab keeps at least one layer).
void cuts_from_histogram(const unsigned long long *hist, uint32_t n_bins, uint32_t bin_out, uint32_t G, uint32_t n_slabs,
                         uint32_t *out_z)
{
    for (uint32_t k = 0; k <= n_slabs; ++k) out_z[k] = (uint32_t) ((uint64_t) G * k / n_slabs);  // equal heights
    unsigned __int128 total = 0;
    for (uint32_t b = 0; b < n_bins; ++b) total += hist[b];
    if (total == 0) return;
    unsigned __int128 before = 0;
    uint32_t k = 1;
    for (uint32_t b = 0; b < n_bins && k < n_slabs; ++b) {
        const unsigned __int128 after = before + hist[b];
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
}

bool plan_params_ok(o2v_hip_ctx *ctx, const o2v_hip_params *params, uint32_t n_slabs)
{
    const uint32_t ss = params->supersampling ? params->supersampling : 1u;
    const uint32_t G = params->resolution;
    if (G == 0 || ss > 2 || (uint64_t) G * ss > 65535u || n_slabs == 0 || n_slabs > G) {
        ctx->err = "slab plan: resolution must be non-zero and below 65536 samples, 1 <= n_slabs <= resolution";
        return false;
    }
    return true;
}

}  // namespace

extern "C" {

int o2v_hip_plan_slabs(o2v_hip_ctx *ctx, const o2v_hip_params *params, uint32_t n_slabs, uint32_t *out_z,
                       float *out_bounds)
{
    if (!ctx || !params || !out_z || n_slabs == 0) return O2V_HIP_ERR_BAD_ARGUMENT;
    if (!plan_params_ok(ctx, params, n_slabs)) return O2V_HIP_ERR_BAD_ARGUMENT;
    const uint32_t G = params->resolution;
    for (uint32_t k = 0; k <= n_slabs; ++k) out_z[k] = (uint32_t) ((uint64_t) G * k / n_slabs);  // equal heights
    if (out_bounds)
        for (int i = 0; i < 6; ++i) out_bounds[i] = params->bounds_known ? params->bounds[i] : 0.f;
    if (ctx->n_tris == 0) return O2V_HIP_OK;
    uint32_t n_bins = 0, bin_out = 0;
    int rc;
    if ((rc = plan_passes(ctx, params, 0, ctx->n_tris, nullptr, 0, n_bins, bin_out, nullptr))) return rc;
    if (out_bounds && !params->bounds_known)
        for (int i = 0; i < 6; ++i) out_bounds[i] = ord2f_host(ctx->h_ctr->bounds_enc[i]);
    cuts_from_histogram(ctx->h_zhist, n_bins, bin_out, G, n_slabs, out_z);
    return O2V_HIP_OK;
}

void o2v_hip_cuts_from_histogram(const uint64_t *hist, uint32_t n_bins, uint32_t bin_layers, uint32_t resolution,
                                 uint32_t n_slabs, uint32_t *out_z)
{
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "histogram word");
    if (!hist || !out_z || !n_slabs || !bin_layers) return;
    cuts_from_histogram(reinterpret_cast<const unsigned long long *>(hist), n_bins, bin_layers, resolution, n_slabs, out_z);
}

int o2v_hip_voxelize_sharded(o2v_hip_ctx *ctx, o2v_hip_comm *comm, const o2v_hip_params *params, uint64_t *out_count,
                             uint64_t *out_counts_all, uint32_t *out_cuts)
{
    if (!ctx || !params) return O2V_HIP_ERR_BAD_ARGUMENT;
    if (out_count) *out_count = 0;
    const uint32_t world = comm ? (uint32_t) comm->world : 1u, rank = comm ? (uint32_t) comm->rank : 0u;
    // test hook: a world of one normally needs no collective; O2V_TEST_FORCE_COLLECTIVES=1 runs them anyway (this is how
    // the RCCL code path is exercised on a single-GPU machine)
    const char *force = std::getenv("O2V_TEST_FORCE_COLLECTIVES");
    if (world == 1 && !(comm && force && force[0] == '1')) {
        o2v_hip_params whole = *params;
        whole.z_begin = whole.z_end = 0;
        uint64_t n = 0;
        const int rc = o2v_hip_voxelize(ctx, &whole, &n);
        if (rc) return rc;
        if (out_count) *out_count = n;
        if (out_counts_all) out_counts_all[0] = n;
        if (out_cuts) {
            out_cuts[0] = 0;
            out_cuts[1] = params->resolution;
        }
        return O2V_HIP_OK;
    }
    // Everything that can fail on one rank alone - the device, the allocations of the planning passes - happens before the
    // first collective, and the ranks then agree on going ahead (one 4-byte max-reduce): a rank that returned early would
    // leave the others waiting for it in RCCL.  The only word that has to exist for that is allocated first.
    if (hipSetDevice(ctx->device) != hipSuccess) {
        ctx->err = "hipSetDevice failed";
        return O2V_HIP_ERR_HIP;  // (nothing can be communicated from a rank without its device)
    }
    if (!ctx->d_status) {
        if (hipMalloc(reinterpret_cast<void **>(&ctx->d_status), kReadyWords * sizeof(uint32_t)) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void **>(&ctx->h_status), sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
            ctx->err = "allocating the status word failed";
            return O2V_HIP_ERR_OUT_OF_MEMORY;
        }
    }
    const uint64_t T = ctx->n_tris, n_blocks = (T + kBlock - 1) / kBlock;
    const uint64_t bpr = std::max<uint64_t>(1, (n_blocks + world - 1) / world);
    float parts_ms[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const bool measure_collectives = (params->flags & (O2V_HIP_FLAG_STAGE_TIMES | O2V_HIP_FLAG_KERNEL_TIMES)) != 0;
    int rc_prepare = O2V_HIP_OK;
    {
        auto prepare = [&]() -> int {
            if (!ctx->ev_coll[0])
                for (auto &e : ctx->ev_coll) O2V_CHECK(create_timing_event(&e));
            if (ctx->cap_counts < world) {
                if (ctx->d_counts) O2V_CHECK(hipFree(ctx->d_counts));
                if (ctx->h_counts) O2V_CHECK(hipHostFree(ctx->h_counts));
                ctx->d_counts = ctx->h_counts = nullptr;
                ctx->cap_counts = 0;
                O2V_CHECK(hipMalloc(reinterpret_cast<void **>(&ctx->d_counts), world * sizeof(unsigned long long)));
                O2V_CHECK(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_counts), world * sizeof(unsigned long long), hipHostMallocDefault));
                ctx->cap_counts = world;
            }
            if (!ctx->d_zhist) {
                O2V_CHECK(hipMalloc(reinterpret_cast<void **>(&ctx->d_zhist), kPlanBins * sizeof(unsigned long long)));
                O2V_CHECK(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_zhist), kPlanBins * sizeof(unsigned long long), hipHostMallocDefault));
            }
            int rc_grow;
            if ((rc_grow = grow(ctx, ctx->d_zrange, ctx->cap_zrange, std::max<uint64_t>(bpr * (uint64_t) world, 1)))) return rc_grow;
            if ((rc_grow = grow(ctx, ctx->d_plan_gather, ctx->cap_plan_gather, ((uint64_t) kPlanBins + bpr) * world))) return rc_grow;
            if (!ctx->d_zrange_xform) O2V_CHECK(hipMalloc(reinterpret_cast<void **>(&ctx->d_zrange_xform), 12 * sizeof(float)));
            if ((rc_grow = grow(ctx, ctx->d_block_list, ctx->cap_block_list, std::max<uint64_t>(n_blocks, 1)))) return rc_grow;
            return O2V_HIP_OK;
        };
        // (bad parameters are a failure of this rank like any other: reported through the status word, so that the other
        // ranks do not wait in the all-reduce for a rank that has already returned)
        const bool params_ok = plan_params_ok(ctx, params, world);
        rc_prepare = params_ok ? prepare() : O2V_HIP_ERR_BAD_ARGUMENT;
        if (const char *fail = std::getenv("O2V_TEST_FAIL_RANK"); fail && std::atoi(fail) == (int) rank && rc_prepare == O2V_HIP_OK) {
            ctx->err = "O2V_TEST_FAIL_RANK: simulated failure of this rank before the collectives";  // test hook
            rc_prepare = O2V_HIP_ERR_OUT_OF_MEMORY;
        }
        // One max-reduce carries the ranks' "not ready" words and - unless the caller gave the bounds - the bounds of every rank's
        // share of the triangles (k_pack_ready): the first collective of the step, and the only one before the histogram.
        hipStream_t s0 = ctx->stream;
        const uint64_t b0r = std::min<uint64_t>(n_blocks, (uint64_t) rank * bpr), b1r = std::min<uint64_t>(n_blocks, (uint64_t) (rank + 1) * bpr);
        const uint64_t share_begin = b0r * kBlock, share_end = std::min<uint64_t>(T, b1r * kBlock);
        const uint64_t n_share = share_end > share_begin ? share_end - share_begin : 0;
        ctx->ctr_clean = false;
        hipLaunchKernelGGL(k_init, dim3(1), dim3(64), 0, s0, ctx->d_ctr, kPassCounterWords);
        if (!params->bounds_known && !rc_prepare && n_share)
            hipLaunchKernelGGL(k_bounds, dim3((uint32_t) std::min<uint64_t>((uint64_t) ctx->num_cus, (n_share * 9 / 12 + kBoundsBlock) / kBoundsBlock)),
                               dim3(kBoundsBlock), 0, s0, ctx->d_verts + share_begin * 9, n_share * 9, ctx->d_ctr);
        hipLaunchKernelGGL(k_pack_ready, dim3(1), dim3(64), 0, s0, ctx->d_ctr, ctx->d_status, rc_prepare ? 1u : 0u);
        // (no local HIP error may keep this rank out of the collective: its peers would wait for the time limit instead of seeing
        // the error in the status word - a timing event that cannot be recorded only switches the timing off)
        bool time_it = measure_collectives && ctx->ev_coll[0] && ctx->ev_coll[1];
        if (time_it && hipEventRecord(ctx->ev_coll[0], s0) != hipSuccess) {
            (void) hipGetLastError();
            time_it = false;
        }
        const std::string prepare_err = ctx->err;
        if (comm->allreduce_max_u32(ctx->d_status, 7, s0)) {
            ctx->err = std::string("collective failed: ") + comm->err;
            return O2V_HIP_ERR_HIP;
        }
        if (time_it && hipEventRecord(ctx->ev_coll[1], s0) != hipSuccess) {
            (void) hipGetLastError();
            time_it = false;
        }
        hipLaunchKernelGGL(k_unpack_ready, dim3(1), dim3(64), 0, s0, ctx->d_status, ctx->d_ctr);
        O2V_CHECK(hipMemcpyAsync(ctx->h_status, ctx->d_status + 6, sizeof(uint32_t), hipMemcpyDeviceToHost, s0));
        // (the first collective of the run: if a rank of the job never gets here - it died, or the node is set up wrongly - the
        // others say so after o2v::comm_timeout_seconds() instead of waiting for ever)
        if (!o2v::stream_wait_limited(s0, "the readiness all-reduce of the sharded run", ctx->err)) {
            // the collective stays queued on the stream: nothing may wait for this stream or this communicator again (a later
            // hipFree / hipStreamSynchronize / ncclCommDestroy would only move the hang to the teardown)
            ctx->poisoned = true;
            comm->poisoned = true;
            return O2V_HIP_ERR_HIP;
        }
        if (time_it) O2V_CHECK(hipEventElapsedTime(&parts_ms[0], ctx->ev_coll[0], ctx->ev_coll[1]));
        if (rc_prepare) {
            ctx->err = prepare_err;
            return rc_prepare;
        }
        if (*ctx->h_status) {
            ctx->err = "another rank could not prepare its sharded run";
            return O2V_HIP_ERR_HIP;
        }
    }
    const auto t0 = std::chrono::steady_clock::now();
    // this rank's share of the triangle list, in whole blocks of 256 (the unit of the block extents)
    const uint64_t b0 = std::min<uint64_t>(n_blocks, (uint64_t) rank * bpr), b1 = std::min<uint64_t>(n_blocks, (uint64_t) (rank + 1) * bpr);
    const uint64_t tri_begin = b0 * kBlock, tri_end = std::min<uint64_t>(T, b1 * kBlock);
    uint32_t n_bins = 0, bin_out = 0;
    float coll_ms = 0.f;
    int rc = plan_passes(ctx, params, tri_begin, tri_end, comm, bpr, n_bins, bin_out, &coll_ms, parts_ms, /*bounds_reduced=*/true);
    if (rc) return rc;
    std::vector<uint32_t> cuts(world + 1);
    cuts_from_histogram(ctx->h_zhist, n_bins, bin_out, params->resolution, world, cuts.data());
    o2v_hip_params mine = *params;
    if (!params->bounds_known) {
        mine.bounds_known = 1;
        for (int i = 0; i < 6; ++i) mine.bounds[i] = ord2f_host(ctx->h_ctr->bounds_enc[i]);
    }
    mine.z_begin = cuts[rank];
    mine.z_end = cuts[rank + 1];
    const float plan_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();

    uint64_t n = 0;
    const int rc_vox = o2v_hip_voxelize(ctx, &mine, &n);
    // Every rank takes part in the exchange of the counts, also one whose voxelization failed (it reports ~0), so that no
    // rank is left waiting in a collective.
    hipStream_t s = ctx->stream;
    ctx->h_counts[rank] = rc_vox ? ~0ull : n;
    const std::string vox_err = ctx->err;
    O2V_CHECK(hipMemcpyAsync(ctx->d_counts + rank, ctx->h_counts + rank, sizeof(unsigned long long), hipMemcpyHostToDevice, s));
    const bool time_counts = measure_collectives && ctx->ev_coll[0] && ctx->ev_coll[1];
    if (time_counts) O2V_CHECK(hipEventRecord(ctx->ev_coll[0], s));
    rc = comm->allgather(ctx->d_counts, sizeof(unsigned long long), s);
    if (rc) {
        ctx->err = std::string("collective failed: ") + comm->err;
        return rc;
    }
    if (time_counts) O2V_CHECK(hipEventRecord(ctx->ev_coll[1], s));
    O2V_CHECK(hipMemcpyAsync(ctx->h_counts, ctx->d_counts, world * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    O2V_CHECK(hipStreamSynchronize(s));
    float ms = 0.f;
    if (time_counts) O2V_CHECK(hipEventElapsedTime(&ms, ctx->ev_coll[0], ctx->ev_coll[1]));
    parts_ms[4] = ms;
    ctx->timings.plan_ms = plan_ms;
    ctx->timings.collective_ms = coll_ms + ms + parts_ms[0];
    for (int i = 0; i < 5; ++i) ctx->timings.collective_parts_ms[i] = parts_ms[i];
    if (rc_vox) {
        ctx->err = vox_err;
        return rc_vox;
    }
    for (uint32_t r = 0; r < world; ++r)
        if (ctx->h_counts[r] == ~0ull) {
            ctx->err = "the voxelization failed on rank " + std::to_string(r);
            return O2V_HIP_ERR_HIP;
        }
    if (out_count) *out_count = n;
    if (out_counts_all)
        for (uint32_t r = 0; r < world; ++r) out_counts_all[r] = ctx->h_counts[r];
    if (out_cuts)
        for (uint32_t r = 0; r <= world; ++r) out_cuts[r] = cuts[r];
    return O2V_HIP_OK;
}

int o2v_hip_max_slab_layers(o2v_hip_ctx *ctx, const o2v_hip_params *params, uint32_t *out_layers)
{
    if (!ctx || !params || !out_layers || !params->resolution) return O2V_HIP_ERR_BAD_ARGUMENT;
    *out_layers = 0;
    O2V_CHECK(hipSetDevice(ctx->device));
    size_t free_b = 0, total_b = 0;
    O2V_CHECK(hipMemGetInfo(&free_b, &total_b));
    const uint64_t G = params->resolution;
    // (a layer of the grids is as wide as the mesh's voxel bounding box, grid_box)
    const uint32_t ss_l = params->supersampling ? params->supersampling : 1u;
    const GridBox box = grid_box(ctx, params, ss_l, 0u, params->resolution);
    const uint64_t per_layer_bricks = box.empty ? 1ull
                                                : (uint64_t) ((box.hi[0] - box.lo[0] + kBrickX - 1) / kBrickX) * ((box.hi[1] - box.lo[1] + kBrickY - 1) / kBrickY);
    // per brick: occupancy only (no triangle of the uploaded mesh has a material) one byte per cell; else the 32-bit counter
    // grid and, for the MAX strategy, the 64-bit grid; each with a dirty flag and a dirty-list entry
    const GridModes modes = grid_modes(ctx, params);  // (the same decision o2v_hip_voxelize takes, flags and environment included)
    const bool occupancy_only = modes.occupancy_only;
    const bool max_grid = modes.direct_max && !occupancy_only;
    const uint64_t per_brick = occupancy_only ? kBrickCells * 1ull + 1 + 4
                                              : kBrickCells * 4ull + 1 + 4 + (max_grid ? kBrickCells * 8ull + 1 + 4 : 0ull);
    // what the context already holds of these grids is reusable
    const uint64_t held = (occupancy_only ? 0ull : ctx->grid_cells * 4ull + ctx->brick_cap * 5ull) + ctx->maxgrid_bytes + ctx->maxgrid_brick_cap * 5ull;
    // the work buffers (leaves, tiles, hit pool, sorted records, output) scale with the mesh, not with the grid: a quarter of
    // the device, at least 8 GiB, stays free for them
    // (the hit slabs - at most a sixth of the device, o2v_hip_voxelize - are part of that reserve: what the context holds of
    // them already counts towards it, and they give way if a required buffer does not fit)
    const uint64_t reserve = std::max<uint64_t>(8ull << 30, total_b / 4);
    const uint64_t held_slabs = (uint64_t) ctx->cap_slabs * kInlineHits * kBrickCells * ctx->slabs_stride * sizeof(uint32_t);
    const uint64_t avail = (uint64_t) free_b + held + held_slabs > reserve ? (uint64_t) free_b + held + held_slabs - reserve : 0;
    uint64_t brick_layers = avail / (per_layer_bricks * per_brick);
    brick_layers = std::min<uint64_t>(brick_layers, ((1ull << 31) - 1) / per_layer_bricks);  // 32-bit brick ids
    const uint64_t layers = brick_layers * kBrickZ;
    // (the grids only span the mesh's box in z as well: if that many layers fit, the whole resolution is one slab)
    const uint64_t box_layers = box.empty ? 0ull : (uint64_t) (box.hi[2] - box.lo[2]) + kBrickZ;
    *out_layers = (layers >= G || layers >= box_layers) ? (uint32_t) G : (uint32_t) layers;
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

int o2v_hip_read_voxels_async(o2v_hip_ctx *ctx, uint32_t *out, uint64_t first, uint64_t count)
{
    if (!ctx || (!out && count)) return O2V_HIP_ERR_BAD_ARGUMENT;
    if (first + count > ctx->n_vox) {
        ctx->err = "voxel range out of bounds";
        return O2V_HIP_ERR_BAD_ARGUMENT;
    }
    if (!count) return O2V_HIP_OK;
    O2V_CHECK(hipSetDevice(ctx->device));
    O2V_CHECK(hipMemcpyAsync(out, ctx->d_out + first, count * sizeof(uint4), hipMemcpyDeviceToHost, ctx->stream));
    return O2V_HIP_OK;
}

int o2v_hip_read_voxels_wait(o2v_hip_ctx *ctx)
{
    if (!ctx) return O2V_HIP_ERR_BAD_ARGUMENT;
    O2V_CHECK(hipSetDevice(ctx->device));
    O2V_CHECK(hipStreamSynchronize(ctx->stream));
    return O2V_HIP_OK;
}

void *o2v_hip_alloc_pinned(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void o2v_hip_free_pinned(void *p)
{
    if (p) (void) hipHostFree(p);
}

int o2v_hip_voxels_device_ptr(o2v_hip_ctx *ctx, const uint32_t **out_ptr, uint64_t *out_count)
{
    if (!ctx || !out_ptr || !out_count) return O2V_HIP_ERR_BAD_ARGUMENT;
    *out_ptr = reinterpret_cast<const uint32_t *>(ctx->d_out);
    *out_count = ctx->n_vox;
    return O2V_HIP_OK;
}

namespace {
// Where a cell's hit records are after a run of the general route (the host-side twin of CellRecords): the first n_slab in its
// brick's slab, the other n_sorted in the sorted array.
struct HostCellRecords {
    size_t slab_first, sorted_first;
    uint32_t n_slab, n_sorted;
};
HostCellRecords host_cell_records(const Occ &o, uint32_t cap_slabs)
{
    const uint32_t cnt = o.count & ~kOccInline;
    const bool inl = (o.count & kOccInline) != 0u;
    const uint32_t slab = inl ? o.offset : o.slab();
    const bool has_slab = inl || slab < cap_slabs;
    HostCellRecords w{};
    w.slab_first = ((size_t) slab * kBrickCells + (o.cell_lo & (kBrickCells - 1u))) * kInlineHits;
    w.n_slab = !has_slab ? 0u : (cnt < kInlineHits ? cnt : kInlineHits);
    w.n_sorted = cnt - w.n_slab;
    w.sorted_first = inl ? 0u : (size_t) o.offset;
    return w;
}
}  // namespace

// Debugging aid: the hit records of one output cell of the last run (the occupied-cell list and the hit pool
// stay valid after a run).  Each record is 6 words: keyhi, keylo, w, u, v (as float bits) and the pool index.
int o2v_hip_debug_cell_hits(o2v_hip_ctx *ctx, uint32_t x, uint32_t y, uint32_t z, uint32_t *out, uint32_t max_records,
                            uint32_t *out_count)
{
    if (!ctx || !out || !out_count) return O2V_HIP_ERR_BAD_ARGUMENT;
    *out_count = 0;
    if (ctx->last_direct) {
        ctx->err = "hit lists are not kept on the direct MAX path: run with O2V_NO_DIRECT_MAX=1 to inspect them";
        return O2V_HIP_ERR_BAD_ARGUMENT;
    }
    O2V_CHECK(hipSetDevice(ctx->device));
    std::vector<Occ> occ(ctx->n_vox);
    std::vector<uint4> vox(ctx->n_vox);
    if (!ctx->n_vox) return O2V_HIP_OK;
    O2V_CHECK(hipMemcpy(occ.data(), ctx->d_occ, occ.size() * sizeof(Occ), hipMemcpyDeviceToHost));
    O2V_CHECK(hipMemcpy(vox.data(), ctx->d_out, vox.size() * sizeof(uint4), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < ctx->n_vox; ++i) {
        if (vox[i].x != x || vox[i].y != y || vox[i].z != z) continue;
        const HostCellRecords where = host_cell_records(occ[i], ctx->cap_slabs);
        const uint32_t cnt = occ[i].count & ~kOccInline;
        const uint32_t n = cnt < max_records ? cnt : max_records;
        std::vector<uint32_t> raw((size_t) n * ctx->sorted_stride);
        const size_t first = where.n_slab ? where.slab_first : where.sorted_first;
        for (uint32_t k = 0; k < n; ++k) {
            const bool in_slab = k < where.n_slab;
            const size_t at = in_slab ? where.slab_first + k : where.sorted_first + (k - where.n_slab);
            O2V_CHECK(hipMemcpy(raw.data() + (size_t) k * ctx->sorted_stride,
                                (in_slab ? ctx->d_slabs : reinterpret_cast<const uint32_t *>(ctx->d_sorted)) + at * ctx->sorted_stride,
                                ctx->sorted_stride * sizeof(uint32_t), hipMemcpyDeviceToHost));
        }
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t *r = &raw[(size_t) k * ctx->sorted_stride];
            uint32_t *o = out + k * 6;
            o[0] = r[0];
            o[1] = r[1];
            o[2] = r[2];
            o[3] = ctx->sorted_stride == 6 ? r[3] : 0u;
            o[4] = ctx->sorted_stride == 6 ? r[4] : 0u;
            o[5] = (uint32_t) (first + (size_t) k);
        }
        *out_count = n;
        break;
    }
    return O2V_HIP_OK;
}

// Debugging aid for parity work: every hit record of the last run (general route), 8 words each: the cell's x, y, z, keyhi
// (sub-voxel << 29 | triangle), keylo (leaf order key), and the bits of w, u, v - what k_voxelize computed per (leaf, voxel)
// pair, before any fold.  Cells in emission order, a cell's hits in the (arbitrary) order of the sorted array.
int o2v_hip_debug_hits(o2v_hip_ctx *ctx, uint32_t *out8, uint64_t max_hits, uint64_t *n_hits)
{
    if (!ctx || !n_hits || (!out8 && max_hits)) return O2V_HIP_ERR_BAD_ARGUMENT;
    *n_hits = 0;
    if (ctx->last_direct) {
        ctx->err = "hit lists are not kept on the direct MAX path: run with O2V_NO_DIRECT_MAX=1 to inspect them";
        return O2V_HIP_ERR_BAD_ARGUMENT;
    }
    if (!ctx->n_vox) return O2V_HIP_OK;
    O2V_CHECK(hipSetDevice(ctx->device));
    std::vector<Occ> occ(ctx->n_vox);
    std::vector<uint4> vox(ctx->n_vox);
    O2V_CHECK(hipMemcpy(occ.data(), ctx->d_occ, occ.size() * sizeof(Occ), hipMemcpyDeviceToHost));
    O2V_CHECK(hipMemcpy(vox.data(), ctx->d_out, vox.size() * sizeof(uint4), hipMemcpyDeviceToHost));
    uint64_t total = 0, end = 0;
    for (const Occ &o : occ) {
        total += o.count & ~kOccInline;
        const HostCellRecords where = host_cell_records(o, ctx->cap_slabs);
        if (where.n_sorted) end = std::max<uint64_t>(end, (uint64_t) where.sorted_first + where.n_sorted);
    }
    *n_hits = total;
    if (total > max_hits) return O2V_HIP_OK;  // (the caller sizes its buffer from *n_hits and calls again)
    std::vector<uint32_t> raw((size_t) end * ctx->sorted_stride);
    if (end) O2V_CHECK(hipMemcpy(raw.data(), ctx->d_sorted, raw.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    // (cells with up to kInlineHits hits keep them side by side in their bricks' slabs)
    std::vector<uint32_t> slabs((size_t) ctx->cap_slabs * kInlineHits * kBrickCells * ctx->sorted_stride);
    if (!slabs.empty()) O2V_CHECK(hipMemcpy(slabs.data(), ctx->d_slabs, slabs.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    uint64_t k = 0;
    for (uint64_t i = 0; i < ctx->n_vox; ++i) {
        const HostCellRecords where = host_cell_records(occ[i], ctx->cap_slabs);
        const uint32_t cnt = occ[i].count & ~kOccInline;
        for (uint32_t h = 0; h < cnt; ++h, ++k) {
            const uint32_t *r = h < where.n_slab ? &slabs[(where.slab_first + h) * ctx->sorted_stride]
                                                 : &raw[(where.sorted_first + (h - where.n_slab)) * ctx->sorted_stride];
            uint32_t *o = out8 + k * 8;
            o[0] = vox[i].x;
            o[1] = vox[i].y;
            o[2] = vox[i].z;
            o[3] = r[0];
            o[4] = r[1];
            o[5] = r[2];
            o[6] = ctx->sorted_stride == 6 ? r[3] : 0u;
            o[7] = ctx->sorted_stride == 6 ? r[4] : 0u;
        }
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
    if (ctx->last_direct) {
        ctx->err = "hit lists are not kept on the direct MAX path: run with O2V_NO_DIRECT_MAX=1 to inspect them";
        return O2V_HIP_ERR_BAD_ARGUMENT;
    }
    O2V_CHECK(hipSetDevice(ctx->device));
    std::vector<Occ> occ(ctx->n_vox);
    O2V_CHECK(hipMemcpy(occ.data(), ctx->d_occ, occ.size() * sizeof(Occ), hipMemcpyDeviceToHost));
    for (const Occ &o : occ) {
        uint32_t b = 0;
        while ((1u << b) < (o.count & ~kOccInline) && b < 31) ++b;
        out32[b]++;
    }
    return O2V_HIP_OK;
}

int o2v_hip_debug_check_third(o2v_hip_ctx *ctx, uint64_t *out2)
{
    if (!ctx || !out2) return O2V_HIP_ERR_BAD_ARGUMENT;
    O2V_CHECK(hipSetDevice(ctx->device));
    unsigned long long *d = nullptr;
    O2V_CHECK(hipMalloc(&d, 2 * sizeof(unsigned long long)));
    const unsigned long long init[2] = {0ull, ~0ull};
    hipError_t e = hipMemcpy(d, init, sizeof(init), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_check_third, dim3((uint32_t) ctx->num_cus * 8u), dim3(256), 0, ctx->stream, d);
        e = hipStreamSynchronize(ctx->stream);
    }
    unsigned long long got[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpy(got, d, sizeof(got), hipMemcpyDeviceToHost);
    (void) hipFree(d);
    O2V_CHECK(e);
    out2[0] = got[0];
    out2[1] = got[0] ? got[1] : 0;
    return O2V_HIP_OK;
}

int o2v_hip_debug_check_div(o2v_hip_ctx *ctx, uint32_t samples, uint64_t seed, uint32_t *out65536)
{
    if (!ctx || !out65536 || !samples) return O2V_HIP_ERR_BAD_ARGUMENT;
    O2V_CHECK(hipSetDevice(ctx->device));
    uint32_t *d = nullptr;
    O2V_CHECK(hipMalloc(&d, 65536 * sizeof(uint32_t)));
    hipError_t e = hipMemsetAsync(d, 0, 65536 * sizeof(uint32_t), ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_check_div, dim3(65536), dim3(256), 0, ctx->stream, d, samples, seed);
        e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out65536, d, 65536 * sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void) hipFree(d);
    O2V_CHECK(e);
    return O2V_HIP_OK;
}

int o2v_hip_debug_counters(const o2v_hip_ctx *ctx, uint64_t *out16)
{
    if (!ctx || !out16) return O2V_HIP_ERR_BAD_ARGUMENT;
    for (int i = 0; i < 16; ++i) out16[i] = ctx->dbg[i];
    return O2V_HIP_OK;
}

int o2v_hip_get_timings(const o2v_hip_ctx *ctx, o2v_hip_timings *out)
{
    if (!ctx || !out) return O2V_HIP_ERR_BAD_ARGUMENT;
    *out = ctx->timings;
    return O2V_HIP_OK;
}

int o2v_hip_get_kernel_times(const o2v_hip_ctx *ctx, o2v_hip_kernel_time *out, uint32_t max_entries, uint32_t *out_count)
{
    if (!ctx || !out_count || (max_entries && !out)) return O2V_HIP_ERR_BAD_ARGUMENT;
    const uint32_t n = (uint32_t) std::min<size_t>(ctx->kernel_times.size(), max_entries);
    for (uint32_t i = 0; i < n; ++i) out[i] = ctx->kernel_times[i];
    *out_count = (uint32_t) ctx->kernel_times.size();
    return O2V_HIP_OK;
}

const char *o2v_hip_build_id(void) { return O2V_BUILD_ID; }

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
