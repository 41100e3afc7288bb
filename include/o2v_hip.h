/*
 * o2v_hip.h -- thin C-ABI between the C++ host code and the gfx950 HIP pipeline.
 *
 * Plain pointers and sizes only; no C++ or torch types.  This is the layer the host side of
 * obj2voxel_voxelize() (include/obj2voxel.h) calls where the reference runs its chunk loop
 * (reference src/obj2voxel.cpp:467-520) and Voxelizer::voxelize (reference src/voxelization.cpp:480-526).
 * A maintainer of the reference binds these entry points from obj2voxel.cpp; see INTEGRATION.md.
 *
 * Call sequence:  create -> set_triangles[/set_textures] -> voxelize -> read_voxels -> destroy.
 * A context owns one GPU's z-slab of the dense voxel grid and may be reused for any number of voxelize
 * calls (the grid is left clean by every call).  One context per GPU; contexts are not thread-safe.
 */
#ifndef O2V_HIP_H
#define O2V_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct o2v_hip_ctx o2v_hip_ctx;

/* error codes returned by every int-returning function */
enum {
    O2V_HIP_OK = 0,
    O2V_HIP_ERR_NO_DEVICE = 1,     /* no usable gfx950 device / HIP runtime failure at create */
    O2V_HIP_ERR_HIP = 2,           /* a HIP call failed; see o2v_hip_last_error */
    O2V_HIP_ERR_BAD_ARGUMENT = 3,
    O2V_HIP_ERR_OUT_OF_MEMORY = 4,
    O2V_HIP_ERR_LIMIT = 5          /* mesh exceeds an implementation limit (e.g. >= 2^29 triangles) */
};

/* triangle material types: reference src/triangle.hpp:21-29 */
enum { O2V_HIP_TRI_MATERIALLESS = 1, O2V_HIP_TRI_UNTEXTURED = 2, O2V_HIP_TRI_TEXTURED = 3 };

/* Texture description; pixels are copied to the device by o2v_hip_set_textures.
 * channels: 3 = RGB, 4 = ARGB, 8 bits each (reference include/obj2voxel.h:317-320). wrap: 0 clamp, 1 repeat. */
typedef struct {
    const uint8_t *pixels;
    uint32_t width, height, channels, wrap;
} o2v_hip_texture;

/* Parameters of one voxelization: what the reference keeps in obj2voxel_instance (src/obj2voxel.cpp:142-173). */
typedef struct {
    uint32_t resolution;       /* output resolution (obj2voxel_set_resolution) */
    uint32_t supersampling;    /* 1 or 2 (obj2voxel_set_supersampling) */
    uint32_t strategy;         /* 0 = MAX, 1 = BLEND (obj2voxel_set_color_strategy) */
    int32_t unit_transform[9]; /* row-major (obj2voxel_set_unit_transform) */
    uint32_t bounds_known;     /* 1: use bounds[] (obj2voxel_set_mesh_boundaries), 0: reduce them on the device */
    float bounds[6];           /* min xyz, max xyz */
    uint32_t z_begin, z_end;   /* this GPU's slab of output z, [z_begin, z_end); 0,0 = the whole grid */
    uint32_t flags;            /* O2V_HIP_FLAG_*; 0 = the product's defaults */
    /* An x / y tile of the output grid, [x_begin, x_end) x [y_begin, y_end) (0, 0 = the whole axis; begin a multiple of 4): only
     * the voxels inside are produced, as with the z slab.  One pass handles a box - the mesh's voxel bounding box within slab and
     * tile - of at most 65 535 samples (resolution x supersampling) per axis: voxel coordinates travel in 16-bit fields relative
     * to the box (the reference carries u32, src/util.hpp:185-196).  A wider grid is voxelized tile by tile; every output voxel
     * belongs to exactly one tile, so the tiles' records concatenated are the whole grid's (obj2voxel_voxelize() does that). */
    uint32_t x_begin, x_end, y_begin, y_end;
} o2v_hip_params;

/* o2v_hip_params::flags.
 * EXACT_CLIP: switches off every piece of work-removal logic in the clip kernel that is not the reference's own
 * arithmetic (the separating-axis row test, the bounding-box plane masks with their margins, the single-plane rule): every
 * candidate voxel of a leaf's clamped AABB is tested as reference src/voxelization.cpp:446-470 does (plane-distance cull,
 * then all six planes through the classification of splitTriangle, :190-232).  Slower, results must be identical: the
 * tests run both on the device and compare (tests/test_gpu_exact_ab.py).  Also forced by O2V_EXACT_CLIP=1 in the
 * environment.
 * KERNEL_TIMES: brackets every kernel launch of the pipeline with two HIP events on the stream it is launched on;
 * o2v_hip_get_kernel_times then returns the per-kernel device times of the call (summed over the launches of one kernel).
 * Implies STAGE_TIMES.
 * STAGE_TIMES: records a HIP event before, between and behind the stages of a pass, for o2v_hip_timings' stage times and
 * total_ms.  Not the default because an event between two kernels is a command of its own in the queue and costs about 4 us
 * of device time (six of them: 3 % of the bench headline's step, profiles/r05/NOTES.md). */
enum { O2V_HIP_FLAG_EXACT_CLIP = 1u, O2V_HIP_FLAG_KERNEL_TIMES = 2u, O2V_HIP_FLAG_STAGE_TIMES = 4u };

/* Device times of the last o2v_hip_voxelize call.  voxelize_ms is measured in every call (two hipEvents that ride on the clip
 * kernel's own dispatch), passes and plan_ms likewise; the other stage times, total_ms and the collectives' times only in a call
 * made with O2V_HIP_FLAG_STAGE_TIMES (else 0) - then voxelize_ms, too, is the time between two events on the stream, and every
 * timed collective of a sharded run is followed by a wait on the host (three more round trips per call). */
typedef struct {
    float bounds_ms;     /* K0  mesh bounds reduce + transform setup */
    float expand_ms;     /* K1  transform, classify, exact subdivision into leaves and tiles */
    float voxelize_ms;   /* K2  AABB walk + plane cull + SAT pre-test + six-plane clip + hit append (dense-grid atomics) */
    float scan_ms;       /* K5  dirty-brick scan, occupied-cell compaction + offsets, hit scatter, brick reset (on the direct
                            MAX path: the flag scan of the 64-bit grid, plus these kernels only if the mesh has subdivided
                            triangles) */
    float resolve_ms;    /* K3  per-cell ordered replay (MAX / BLEND), colour lookup, ARGB pack; emission of the max grid */
    float total_ms;      /* first event to last event */
    uint32_t passes;     /* 1, or more if a device buffer had to grow and the pipeline was re-run */
    float plan_ms;       /* o2v_hip_voxelize_sharded only: sharded bounds + work histogram passes incl. their collectives
                            (host wall time; not part of total_ms) */
    float collective_ms; /* ... of which inside the collectives (all-reduce of bounds and histogram, all-gather of the
                            block extents and of the slab counts) */
    float collective_parts_ms[5]; /* the same by collective, device time on the context's stream: [0] the ranks' readiness words
                            and the mesh bounds in one all-reduce (max of 7 x u32: the minima travel as their complements),
                            [1] unused (0), [2] every rank's z histogram of predicted work (2048 x u64) and the z extents of
                            its blocks (8 bytes per 256 triangles) in one all-gather - each rank adds the histograms up itself -
                            [3] unused (0), [4] slab voxel counts (all-gather, 8 bytes per rank) */
} o2v_hip_timings;

/* Work counters of the last o2v_hip_voxelize call. */
typedef struct {
    uint64_t triangles;   /* input triangles */
    uint64_t leaves;      /* leaf sub-triangles overlapping the slab */
    uint64_t tiles;       /* work tiles of <= 256 candidate voxels */
    uint64_t candidates;  /* (leaf, voxel) pairs examined */
    uint64_t hits;        /* (leaf, voxel) pairs with non-zero weight */
    uint64_t voxels;      /* occupied output voxels */
    uint64_t grid_cells;  /* dense grid cells owned by this context (bricks of 4x4x4, padded) */
    uint64_t grid_bytes;  /* bytes of the dense grid allocation incl. the per-brick dirty flags */
    uint64_t bricks;      /* bricks of the slab */
    uint64_t dirty_bricks;/* bricks that received at least one hit */
    uint64_t pool_slots;  /* hit-pool slots reserved (hits + chunk slack) */
    uint64_t direct_hits; /* hits that went straight into the 64-bit max grid (MAX strategy, unsplit triangles) */
    uint64_t jobs;        /* candidates that passed the plane cull and the separating-axis pre-test: voxel jobs of the clip loop */
    uint64_t certain_hits;/* occupancy-only mode: hits established without a voxel job (the voxel centre's column meets the leaf
                             well inside both), included in hits and direct_hits */
    uint64_t skipped_jobs;/* occupancy-only mode: those of `jobs` that were not run because their voxel was marked already when
                             phase 2 of their batch began (which ones depends on the order the workgroups happen to run in, so
                             this number and `hits` - only hits that were established are counted - vary from run to run;
                             the voxels do not) */
    uint64_t bypassed_leaves; /* occupancy-only mode: those of `leaves` (and `tiles`) that have no Leaf / Tile record - root triangles
                                 of one tile, which the clip kernel stages from the vertex array itself */
} o2v_hip_stats;

int o2v_hip_device_count(void);
int o2v_hip_create(int device, o2v_hip_ctx **out_ctx);
void o2v_hip_destroy(o2v_hip_ctx *ctx);
const char *o2v_hip_last_error(const o2v_hip_ctx *ctx);

/* Host arrays, copied to the device.  verts: [count][9] model-space xyz of the three vertices.
 * uvs: [count][6] or NULL (zeros).  types: [count] or NULL (all MATERIALLESS).  colors: [count][3] or NULL.
 * texids: [count] indices into the texture table, or NULL (all 0). */
int o2v_hip_set_triangles(o2v_hip_ctx *ctx, const float *verts, const float *uvs, const uint32_t *types,
                          const float *colors, const int32_t *texids, uint64_t count);
int o2v_hip_set_textures(o2v_hip_ctx *ctx, const o2v_hip_texture *textures, uint32_t count);

/* Streamed variant of o2v_hip_set_triangles for a triangle source that is drained one triangle at a time (the reference's
 * cache loop, src/obj2voxel.cpp:578-600): the caller fills a block of page-locked staging memory owned by the context,
 * commits it - the block is copied to the device asynchronously while the caller fills the other block - and finishes.
 * `arrays` says which optional arrays the mesh has so far (bit 0 uvs, 1 types, 2 colors, 3 texids; verts always); an
 * array may appear at any commit, the triangles before it get its default (zero uvs / MATERIALLESS / zero colour /
 * texture 0) on the device, and from then on the caller fills it for every triangle.  Only verts is there from the
 * start: before the caller writes an optional array for the first time it asks for it with o2v_hip_stage_arrays (page
 * locking memory costs ~0.1 ms per MiB, and most meshes are vertices only). */
typedef struct {
    float *verts;      /* [capacity][9] */
    float *uvs;        /* [capacity][6], NULL until asked for */
    uint32_t *types;   /* [capacity]    , NULL until asked for */
    float *colors;     /* [capacity][3], NULL until asked for */
    int32_t *texids;   /* [capacity]    , NULL until asked for */
    uint64_t capacity; /* triangles per block */
} o2v_hip_staging;
enum { O2V_HIP_ARRAY_UVS = 1, O2V_HIP_ARRAY_TYPES = 2, O2V_HIP_ARRAY_COLORS = 4, O2V_HIP_ARRAY_TEXIDS = 8 };
int o2v_hip_begin_triangles(o2v_hip_ctx *ctx, o2v_hip_staging *out_block);
/* Makes the optional arrays named by `arrays` part of both staging blocks; *inout_block (the block being filled) gets their
 * addresses.  What the block holds already stays. */
int o2v_hip_stage_arrays(o2v_hip_ctx *ctx, uint32_t arrays, o2v_hip_staging *inout_block);
int o2v_hip_commit_triangles(o2v_hip_ctx *ctx, uint64_t count, uint32_t arrays, o2v_hip_staging *out_next_block);
int o2v_hip_end_triangles(o2v_hip_ctx *ctx, uint32_t any_textured);

/* Runs the whole device pipeline and waits for it.  out_voxel_count receives the number of occupied voxels. */
int o2v_hip_voxelize(o2v_hip_ctx *ctx, const o2v_hip_params *params, uint64_t *out_voxel_count);

/* Work-balanced z-slabs for a multi-GPU job (one process per GPU, every process holding the same triangles):
 * writes n_slabs+1 ascending output-z cuts to out_z (out_z[0] = 0, out_z[n_slabs] = resolution); slab k is
 * [out_z[k], out_z[k+1]) and goes into o2v_hip_params::z_begin/z_end of rank k.  The cuts equalise the predicted
 * number of (triangle, voxel) hits per slab, which the pipeline's time is proportional to; they are a pure
 * function of the triangles and the parameters, so every rank computes the same ones without communicating.
 * The reference balances its worker pool dynamically over 64^3 chunks (src/obj2voxel.cpp:482-497); slabs are
 * static, hence the up-front plan.  out_bounds (optional, 6 floats: min xyz, max xyz) receives the mesh bounds
 * found on the way (reference findMeshBounds, src/obj2voxel.cpp:180-200): passing them back as
 * o2v_hip_params::bounds with bounds_known = 1 saves the voxelize call its own bounds pass.  z_begin/z_end of
 * `params` are ignored. */
int o2v_hip_plan_slabs(o2v_hip_ctx *ctx, const o2v_hip_params *params, uint32_t n_slabs, uint32_t *out_z,
                       float *out_bounds);

/* The thickest z-slab (in output layers, a multiple of 4 unless it is the whole grid) whose dense grids fit the device memory
 * that is free right now (plus what the context already holds), leaving room for the work buffers: a voxelization of
 * `params` can be run as ceil(resolution / layers) calls with consecutive slabs.  obj2voxel_voxelize() does that by itself
 * (the reference's sparse VoxelMap has no such limit: src/util.hpp:179-208); 0 layers = not even one brick layer fits. */
int o2v_hip_max_slab_layers(o2v_hip_ctx *ctx, const o2v_hip_params *params, uint32_t *out_layers);

/* Copies voxels [first, first+count) of the last result to host memory as (x, y, z, argb) uint32 quadruples,
 * the layout of the reference's voxel callback (include/obj2voxel.h:35,200-209).  Order is unspecified. */
int o2v_hip_read_voxels(o2v_hip_ctx *ctx, uint32_t *out, uint64_t first, uint64_t count);
/* The same copy in two steps, for overlapping it with the consumer of the previous batch: _async starts the copy on the
 * context's stream (`out` should be pinned memory, see o2v_hip_alloc_pinned), _wait returns when it has landed. */
int o2v_hip_read_voxels_async(o2v_hip_ctx *ctx, uint32_t *out, uint64_t first, uint64_t count);
int o2v_hip_read_voxels_wait(o2v_hip_ctx *ctx);
/* Page-locked host memory (hipHostMalloc / hipHostFree): transfers from and to it run asynchronously at link rate. */
void *o2v_hip_alloc_pinned(size_t bytes);
/* The same on a thread that has not selected a device yet: `device` becomes the calling thread's current device first (a
 * thread's default is device 0, which need not be the one the caller works on - or one it may touch at all). */
void *o2v_hip_alloc_pinned_on(int device, size_t bytes);
void o2v_hip_free_pinned(void *p);
/* Releases the device session that obj2voxel_voxelize() keeps between calls (contexts, dense grids, staging memory). */
void o2v_release_cached_device_memory(void);
/* Device pointer to the same records (valid until the next voxelize/destroy). */
int o2v_hip_voxels_device_ptr(o2v_hip_ctx *ctx, const uint32_t **out_ptr, uint64_t *out_count);

int o2v_hip_get_timings(const o2v_hip_ctx *ctx, o2v_hip_timings *out);
/* Per-kernel device times of the last o2v_hip_voxelize call made with O2V_HIP_FLAG_KERNEL_TIMES (else none): up to
 * max_entries entries are written, *out_count receives how many there are. */
typedef struct {
    char name[48];     /* kernel name as rocprofv3 prints it, e.g. "k_voxelize<false>" */
    float ms;          /* summed over the launches of the call's last pass */
    uint32_t launches;
} o2v_hip_kernel_time;
int o2v_hip_get_kernel_times(const o2v_hip_ctx *ctx, o2v_hip_kernel_time *out, uint32_t max_entries, uint32_t *out_count);
/* Hash of the device sources this library was built from: profiles record it, bench.py refuses counters of another build. */
const char *o2v_hip_build_id(void);
int o2v_hip_get_stats(const o2v_hip_ctx *ctx, o2v_hip_stats *out);
/* The mesh transform of the last run: row-major 3x3 then translation (reference obj2voxel.cpp:370-402). */
int o2v_hip_get_transform(const o2v_hip_ctx *ctx, float out12[12]);

/* Debugging aid for parity work: hit records of one output cell of the last run, 6 words each
 * (keyhi = sub-voxel<<29 | triangle, keylo = leaf order key, w, u, v as float bits, pool index).  The two debug calls
 * need the hit lists, which the direct MAX path does not keep: they return O2V_HIP_ERR_BAD_ARGUMENT after such a run
 * (set O2V_NO_DIRECT_MAX=1 in the environment to route every hit through the lists). */
int o2v_hip_debug_cell_hits(o2v_hip_ctx *ctx, uint32_t x, uint32_t y, uint32_t z, uint32_t *out, uint32_t max_records,
                            uint32_t *out_count);

/* Debugging aid for parity work: every hit record of the last run, 8 words each (cell x, y, z, keyhi, keylo, bits of w, u, v):
 * what the clip kernel computed per (leaf, voxel) pair, before any fold.  *n_hits receives the number of hits; records are
 * written only if max_hits >= *n_hits (call with max_hits = 0 first).  Same restriction as o2v_hip_debug_cell_hits. */
int o2v_hip_debug_hits(o2v_hip_ctx *ctx, uint32_t *out8, uint64_t max_hits, uint64_t *n_hits);

/* Debugging aid: log2 histogram of hits per occupied cell of the last run (32 buckets; bucket b: 2^(b-1) < hits <= 2^b). */
int o2v_hip_debug_hits_histogram(o2v_hip_ctx *ctx, uint64_t *out32);

/* ---- multi-GPU: the grid sharded by z-slab over the GPUs of one node --------------------------
 *
 * Every rank (one per GPU) holds the whole triangle list, like every 64^3 chunk of the reference sees every triangle
 * that overlaps it (src/obj2voxel.cpp:226-243), and voxelizes only its own z-slab (walk clamped as in
 * src/voxelization.cpp:440-444).  Voxel data never crosses GPUs: each output voxel is owned by exactly one slab and the
 * union of the slabs is bit-identical to the single-GPU result.  What the ranks exchange is planning data, with RCCL over
 * xGMI: the passes over the triangle list that find the mesh bounds and the z histogram of predicted work are SHARDED
 * (rank r streams triangles [r, r+1) * T / N only).  Three collectives per run: an all-reduce (max of 7 words: the six bounds -
 * the minima as complements - and a "this rank cannot go ahead" word); an all-gather of every rank's partial histogram (2048 u64,
 * added up by every rank itself) together with the z extents of its blocks of 256 triangles (so that each rank can skip the
 * blocks that miss its slab without reading them); an all-gather of the per-slab voxel counts (output offsets for the sink).
 *
 * Two ways to use it:
 *   one process per GPU   o2v_hip_comm_unique_id on rank 0 -> ship the 128 bytes to every rank (MPI, torch.distributed,
 *                         a file) -> o2v_hip_comm_create_rccl on every rank -> o2v_hip_voxelize_sharded, collectively.
 *   one process, N GPUs   o2v_hip_group_*: one context and one host thread per GPU; obj2voxel_voxelize() uses this when
 *                         the environment names more than one device (O2V_DEVICES=0,1,2,3 or O2V_DEVICES=all).
 */
#define O2V_HIP_COMM_ID_BYTES 128
typedef struct o2v_hip_comm o2v_hip_comm;

/* Collectives supplied by the embedding program, on HOST memory, in place; every rank calls them in the same order.
 * Return 0 on success.  Used where RCCL cannot be (tests with a gloo group; two ranks sharing one GPU). */
typedef struct {
    void *user;
    int (*allreduce_min_u32)(void *user, uint32_t *buf, size_t n);
    int (*allreduce_max_u32)(void *user, uint32_t *buf, size_t n);
    int (*allreduce_sum_u64)(void *user, uint64_t *buf, size_t n);
    int (*allgather)(void *user, void *buf, size_t bytes_per_rank); /* rank r's part sits at buf + r * bytes_per_rank */
    int (*broadcast)(void *user, void *buf, size_t bytes, int root);
} o2v_hip_comm_callbacks;

int o2v_hip_comm_unique_id(uint8_t id[O2V_HIP_COMM_ID_BYTES]); /* ncclGetUniqueId; call on one rank */
int o2v_hip_comm_create_rccl(const uint8_t id[O2V_HIP_COMM_ID_BYTES], int rank, int world, int device, o2v_hip_comm **out);
int o2v_hip_comm_create_callbacks(const o2v_hip_comm_callbacks *callbacks, int rank, int world, o2v_hip_comm **out);
void o2v_hip_comm_destroy(o2v_hip_comm *comm);
const char *o2v_hip_comm_kind(const o2v_hip_comm *comm); /* "rccl" or "callbacks" */
const char *o2v_hip_comm_last_error(const o2v_hip_comm *comm);

/* Collective over `comm`: every rank calls it with the same triangles (o2v_hip_set_triangles) and the same params
 * (z_begin / z_end are ignored).  Plans work-balanced slabs from the sharded passes, voxelizes this rank's slab and
 * gathers the slab counts.  out_count: this rank's voxels (read them with o2v_hip_read_voxels); out_counts_all (optional):
 * world entries; out_cuts (optional): world + 1 ascending z cuts, rank r owns [out_cuts[r], out_cuts[r + 1]).
 * With a world of 1 this is o2v_hip_voxelize.  * Time limits: ncclCommInitRank and the run's first collective are given O2V_COMM_TIMEOUT_S seconds (120) for the other ranks to
 * arrive; after that the call fails with a message naming the rank.  A collective that timed out stays queued on the device:
 * the context and the communicator are then unusable (every later call on them fails or would wait for ever), o2v_hip_destroy
 * and o2v_hip_comm_destroy return without waiting for the device (the communicator is aborted, the context's device memory is
 * left to the process' end) - the process should report the error and exit.
 */
int o2v_hip_voxelize_sharded(o2v_hip_ctx *ctx, o2v_hip_comm *comm, const o2v_hip_params *params, uint64_t *out_count,
                             uint64_t *out_counts_all, uint32_t *out_cuts);

/* In-process group: N contexts (one per listed device; a device may be listed more than once, which only makes sense
 * for tests on a single-GPU machine) driven by N host threads.  The ranks talk through RCCL when every device is listed
 * once and librccl can be loaded, otherwise through shared host memory. */
typedef struct o2v_hip_group o2v_hip_group;
enum {
    O2V_HIP_UPLOAD_H2D = 0,       /* every GPU copies the triangles from host memory over its own PCIe link, in parallel */
    O2V_HIP_UPLOAD_BROADCAST = 1, /* one H2D copy to the first GPU, then an RCCL broadcast over xGMI */
    O2V_HIP_UPLOAD_PEER = 2       /* one H2D copy to the first GPU, then hipMemcpyPeerAsync to each of the others */
};
int o2v_hip_group_create(const int *devices, uint32_t n_devices, o2v_hip_group **out);
void o2v_hip_group_destroy(o2v_hip_group *group);
uint32_t o2v_hip_group_size(const o2v_hip_group *group);
o2v_hip_ctx *o2v_hip_group_ctx(o2v_hip_group *group, uint32_t rank); /* for read_voxels / timings / stats of one rank */
const char *o2v_hip_group_comm_kind(const o2v_hip_group *group);      /* "rccl" or "callbacks" */
const char *o2v_hip_group_last_error(const o2v_hip_group *group);
int o2v_hip_group_set_triangles(o2v_hip_group *group, const float *verts, const float *uvs, const uint32_t *types,
                                const float *colors, const int32_t *texids, uint64_t count, int upload_mode);
int o2v_hip_group_set_textures(o2v_hip_group *group, const o2v_hip_texture *textures, uint32_t count);
/* out_counts: n_devices entries; out_cuts (optional): n_devices + 1 entries */
int o2v_hip_group_voxelize(o2v_hip_group *group, const o2v_hip_params *params, uint64_t *out_counts, uint32_t *out_cuts);

/* Host-only pieces of the above, exported so that they can be tested without a GPU:
 * the cuts for n_slabs slabs of equal predicted work from a z histogram of n_bins bins of bin_layers output layers each
 * (out_z: n_slabs + 1 entries); a self-test that drives every callback of a callbacks table with known patterns from
 * this rank and checks what comes back (0 = all collectives behave as specified); and the same for the shared-memory
 * exchange between the threads of an in-process group. */
void o2v_hip_cuts_from_histogram(const uint64_t *hist, uint32_t n_bins, uint32_t bin_layers, uint32_t resolution,
                                 uint32_t n_slabs, uint32_t *out_z);
int o2v_hip_comm_callbacks_selftest(const o2v_hip_comm_callbacks *callbacks, int rank, int world);
int o2v_hip_group_exchange_selftest(uint32_t n_threads);
/* The RCCL code path of an n-rank in-process group (unique id, ncclCommInitRank on one thread per rank, all five collectives,
 * teardown) on host memory, without selecting a device: only meaningful with a librccl that works on host memory (the tests'
 * stand-in, loaded through O2V_RCCL_LIB).  0 = everything behaved as specified. */
int o2v_hip_group_rccl_selftest(uint32_t n_ranks);

/* A triangle file (OBJ with MTL + PNG textures, binary STL; `type` = extension or NULL to take the path's) read by the
 * library's own readers into the flat host arrays o2v_hip_set_triangles takes - what obj2voxel_voxelize() does with
 * obj2voxel_set_input_file (reference src/io.cpp:244-312,395-435), without the voxelization.  bench.py uses it to run the
 * real Spot / Dragon / Sponza assets when $O2V_ASSETS holds them.  Arrays a mesh does not need are NULL; the pointers and
 * the texture pixels stay valid until o2v_mesh_free. */
typedef struct o2v_mesh o2v_mesh;
int o2v_mesh_load_file(const char *path, const char *type, o2v_mesh **out);
uint64_t o2v_mesh_arrays(const o2v_mesh *mesh, const float **verts, const float **uvs, const uint32_t **types,
                         const float **colors, const int32_t **texids, uint32_t *n_textures); /* returns the triangle count */
int o2v_mesh_texture(const o2v_mesh *mesh, uint32_t index, o2v_hip_texture *out);
void o2v_mesh_free(o2v_mesh *mesh);

/* Debugging aid for kernel work: 16 event counters of the clip loop of the last run.  All zero unless the library was
 * built with -DO2V_INSTRUMENT (make INSTR=1, tools/instrument.sh); the meaning of each slot is documented there. */
int o2v_hip_debug_counters(const o2v_hip_ctx *ctx, uint64_t *out16);

/* Self checks of the clip loop's short division forms on the device (obj2voxel_amd/csrc/o2v_dev_arith.hpp).
 * _check_third: x / 3 against its three-instruction form for all 2^32 float32 bit patterns; out2[0] = differing inputs,
 * out2[1] = the first of them + 1 (0 if none).
 * _check_div: n / d against the lean form for `samples` pairs per pair of biased exponents (numerator, divisor);
 * out65536[en * 256 + ed] = differing pairs.  The kernels only use the lean form inside the region that is all zero. */
int o2v_hip_debug_check_third(o2v_hip_ctx *ctx, uint64_t *out2);
int o2v_hip_debug_check_div(o2v_hip_ctx *ctx, uint32_t samples, uint64_t seed, uint32_t *out65536);

#ifdef __cplusplus
}
#endif
#endif
