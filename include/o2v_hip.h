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
} o2v_hip_params;

/* Per-stage device times of the last o2v_hip_voxelize call, measured with hipEvents on the pipeline's stream. */
typedef struct {
    float bounds_ms;     /* K0  mesh bounds reduce + transform setup */
    float expand_ms;     /* K1  transform, classify, exact subdivision into leaves and tiles */
    float voxelize_ms;   /* K2  AABB walk + plane cull + SAT pre-test + six-plane clip + hit append (dense-grid atomics) */
    float scan_ms;       /* K5  dirty-brick scan, occupied-cell compaction + offsets, hit scatter, brick reset (on the direct
                            MAX path: the host's look at the counters, plus these kernels only if any hit was pooled) */
    float resolve_ms;    /* K3  per-cell ordered replay (MAX / BLEND), colour lookup, ARGB pack; emission of the max grid */
    float total_ms;      /* first event to last event */
    uint32_t passes;     /* 1, or more if a device buffer had to grow and the pipeline was re-run */
} o2v_hip_timings;

/* Work counters of the last o2v_hip_voxelize call. */
typedef struct {
    uint64_t triangles;   /* input triangles */
    uint64_t leaves;      /* leaf sub-triangles overlapping the slab */
    uint64_t tiles;       /* work tiles of <= 256 candidate voxels */
    uint64_t candidates;  /* (leaf, voxel) pairs examined */
    uint64_t hits;        /* (leaf, voxel) pairs with non-zero weight */
    uint64_t voxels;      /* occupied output voxels */
    uint64_t grid_cells;  /* dense grid cells owned by this context (bricks of 16x4x4, padded) */
    uint64_t grid_bytes;  /* bytes of the dense grid allocation incl. the per-brick dirty flags */
    uint64_t bricks;      /* bricks of the slab */
    uint64_t dirty_bricks;/* bricks that received at least one hit */
    uint64_t pool_slots;  /* hit-pool slots reserved (hits + chunk slack) */
    uint64_t direct_hits; /* hits that went straight into the 64-bit max grid (MAX strategy, unsplit triangles) */
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

/* Copies voxels [first, first+count) of the last result to host memory as (x, y, z, argb) uint32 quadruples,
 * the layout of the reference's voxel callback (include/obj2voxel.h:35,200-209).  Order is unspecified. */
int o2v_hip_read_voxels(o2v_hip_ctx *ctx, uint32_t *out, uint64_t first, uint64_t count);
/* Device pointer to the same records (valid until the next voxelize/destroy). */
int o2v_hip_voxels_device_ptr(o2v_hip_ctx *ctx, const uint32_t **out_ptr, uint64_t *out_count);

int o2v_hip_get_timings(const o2v_hip_ctx *ctx, o2v_hip_timings *out);
int o2v_hip_get_stats(const o2v_hip_ctx *ctx, o2v_hip_stats *out);
/* The mesh transform of the last run: row-major 3x3 then translation (reference obj2voxel.cpp:370-402). */
int o2v_hip_get_transform(const o2v_hip_ctx *ctx, float out12[12]);

/* Debugging aid for parity work: hit records of one output cell of the last run, 6 words each
 * (keyhi = sub-voxel<<29 | triangle, keylo = leaf order key, w, u, v as float bits, pool index).  The two debug calls
 * need the hit lists, which the direct MAX path does not keep: they return O2V_HIP_ERR_BAD_ARGUMENT after such a run
 * (set O2V_NO_DIRECT_MAX=1 in the environment to route every hit through the lists). */
int o2v_hip_debug_cell_hits(o2v_hip_ctx *ctx, uint32_t x, uint32_t y, uint32_t z, uint32_t *out, uint32_t max_records,
                            uint32_t *out_count);

/* Debugging aid: log2 histogram of hits per occupied cell of the last run (32 buckets; bucket b: 2^(b-1) < hits <= 2^b). */
int o2v_hip_debug_hits_histogram(o2v_hip_ctx *ctx, uint64_t *out32);

/* Debugging aid for kernel work: 16 event counters of the clip loop of the last run.  All zero unless the library was
 * built with -DO2V_INSTRUMENT (make INSTR=1, tools/instrument.sh); the meaning of each slot is documented there. */
int o2v_hip_debug_counters(const o2v_hip_ctx *ctx, uint64_t *out16);

#ifdef __cplusplus
}
#endif
#endif
