/*
 * o2v_oracle.c -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * A sequential, plain-C restatement of obj2voxel's per-triangle voxelization path, written from the
 * behaviour of the reference sources (cited per function as file:line relative to the reference tree).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the timed CPU baseline.  The shipped library (libobj2voxel_amd.so) never links or calls it.
 *
 * Pinning status
 *   - The reference cannot be built in this image: its math/colour/image layer is the third-party module
 *     Eisenwave/voxel-io (.gitmodules:1-3, version UN-PINNED, directory empty).  Writing stand-in headers is
 *     not a reference build, so there is no oracle/_ref.
 *   - This restatement is pinned on every result the reference's own tests hold for the path
 *     (test/main.cpp:120-126,128-156,194-252): unit cube @64 -> 23816, @128 -> 96776,
 *     three planes @32 -> 3072, @128 -> 49152 voxels (tests/test_oracle_golden.py).
 *   - voxel-io semantics used here that NO reference test pins ("parity unpinned"):
 *       dot  = sequential sum from 0; cross = standard; vec/scalar = per-component IEEE division;
 *       Color32(Vec3f) = (u8)(c*255) per channel, alpha 0xFF; Image::getPixel(uv) = nearest texel,
 *       x = (u32)(u'*w) clamped to w-1, REPEAT: u' = u - floor(u), CLAMP: u' = clamp(u,0,1);
 *       Morton bit order (unobservable: output is unordered).
 *   - Supersampling: Voxelizer::downscale (voxelization.cpp:538-554) is broken in the reference snapshot
 *     (always yields an empty map).  Implemented here are the DOCUMENTED semantics
 *     (voxelization.hpp:82-85, README.adoc:153-163): every sample voxel goes to pos/2 and collisions are
 *     combined with the strategy's combine function, in ascending sub-voxel order (x | y<<1 | z<<2).
 *     Parity for supersampling is therefore unpinned.
 *
 * All arithmetic is IEEE binary32 evaluated in source order; build with -ffp-contract=off (no FMA).
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define O2V_CHUNK 64u               /* constants.hpp:10 */
#define O2V_BATCH 1024u             /* constants.hpp:11 */
#define O2V_SUBDIV_LIMIT 512u       /* constants.hpp:13 */
#define O2V_DIAG_LIMIT 0.5f         /* constants.hpp:15 */
static const float O2V_EPS = 1.0f / (1 << 16); /* voxelization.cpp:15 */

enum { TRI_NONE = 0, TRI_MATERIALLESS = 1, TRI_UNTEXTURED = 2, TRI_TEXTURED = 3 }; /* triangle.hpp:21-29 */
enum { STRAT_MAX = 0, STRAT_BLEND = 1 };                                           /* obj2voxel.h:43-45 */

typedef struct { float x, y, z; } v3;
typedef struct { float x, y; } v2;
typedef struct { v3 v[3]; v2 t[3]; } ttri; /* TexturedTriangle, triangle.hpp:39-144 */

typedef struct {
    const uint8_t *pixels; /* row-major, `channels` bytes per texel: 3 = RGB, 4 = ARGB (obj2voxel.h:317-320) */
    uint32_t width, height, channels, wrap; /* wrap: 0 clamp, 1 repeat (obj2voxel.h:47-50) */
} o2v_oracle_texture;

typedef struct {
    uint64_t candidates, culled, clips, hits, pieces, splits, leaves;
} o2v_oracle_stats;

static o2v_oracle_stats g_stats;              /* totals of the last run */
static _Thread_local o2v_oracle_stats t_stats;  /* per worker thread, merged at the end of a run */
static int g_threads = 1;
static double g_phase_seconds[3] = {0, 0, 0};
/* Chunk-parallel execution like the reference's worker pool (one VOXELIZE_CHUNK command per chunk,
 * obj2voxel.cpp:415-424,979): results do not depend on the thread count because chunks are independent. */
void o2v_oracle_set_threads(int n) { g_threads = n < 1 ? 1 : (n > 1024 ? 1024 : n); }

/* optional trace of one sample-space voxel (debugging aid for parity work) */
static int g_trace_on = 0;
static uint32_t g_trace_pos[3];
static _Thread_local uint64_t g_trace_tri = 0;
static _Thread_local uint32_t g_trace_leaf = 0;
#include <stdio.h>
void o2v_oracle_trace_voxel(int on, uint32_t x, uint32_t y, uint32_t z)
{
    g_trace_on = on;
    g_trace_pos[0] = x;
    g_trace_pos[1] = y;
    g_trace_pos[2] = z;
}

/* ---- voxel-io value-type operations (restated; see header for what is unpinned) ------------------------- */
static inline v3 v3sub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static inline v3 v3add(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static inline float v3dot(v3 a, v3 b)
{
    float r = 0;
    r += a.x * b.x;
    r += a.y * b.y;
    r += a.z * b.z;
    return r;
}
static inline v3 v3cross(v3 a, v3 b)
{
    v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    return r;
}
static inline float comp(v3 a, unsigned i) { return i == 0 ? a.x : i == 1 ? a.y : a.z; }
static inline float fmin2(float a, float b) { return (b < a) ? b : a; } /* std::min */
static inline float fmax2(float a, float b) { return (a < b) ? b : a; } /* std::max */

/* util.hpp:122-146 */
static inline float v3length(v3 a) { return sqrtf(v3dot(a, a)); }
static inline v3 v3normalize(v3 a)
{
    float l = v3length(a);
    v3 r = {a.x / l, a.y / l, a.z / l};
    return r;
}
static inline v3 v3mix(v3 a, v3 b, float t)
{
    float s = 1 - t;
    v3 r = {s * a.x + t * b.x, s * a.y + t * b.y, s * a.z + t * b.z};
    return r;
}
static inline v2 v2mix(v2 a, v2 b, float t)
{
    float s = 1 - t;
    v2 r = {s * a.x + t * b.x, s * a.y + t * b.y};
    return r;
}

/* triangle.hpp:59-106 */
static inline v3 tri_normal(const ttri *t) { return v3cross(v3sub(t->v[1], t->v[0]), v3sub(t->v[2], t->v[0])); }
static inline float tri_area(const ttri *t) { return v3length(tri_normal(t)) / 2; }
static inline v3 tri_min(const ttri *t)
{
    /* obj2voxel::min(a, min(b, c)) component-wise, util.hpp:74-84 */
    v3 r;
    r.x = fmin2(t->v[0].x, fmin2(t->v[1].x, t->v[2].x));
    r.y = fmin2(t->v[0].y, fmin2(t->v[1].y, t->v[2].y));
    r.z = fmin2(t->v[0].z, fmin2(t->v[1].z, t->v[2].z));
    return r;
}
static inline v3 tri_max(const ttri *t)
{
    v3 r;
    r.x = fmax2(t->v[0].x, fmax2(t->v[1].x, t->v[2].x));
    r.y = fmax2(t->v[0].y, fmax2(t->v[1].y, t->v[2].y));
    r.z = fmax2(t->v[0].z, fmax2(t->v[1].z, t->v[2].z));
    return r;
}
static inline void tri_voxel_bounds(const ttri *t, uint32_t lo[3], uint32_t hi[3])
{
    /* voxelMin = floor(min).cast<u32>(); voxelMax = floor(max).cast<u32>() + 1  (triangle.hpp:91-100) */
    v3 mn = tri_min(t), mx = tri_max(t);
    lo[0] = (uint32_t) floorf(mn.x);
    lo[1] = (uint32_t) floorf(mn.y);
    lo[2] = (uint32_t) floorf(mn.z);
    hi[0] = (uint32_t) floorf(mx.x) + 1u;
    hi[1] = (uint32_t) floorf(mx.y) + 1u;
    hi[2] = (uint32_t) floorf(mx.z) + 1u;
}

/* ---- affine transform (util.hpp:212-281) ----------------------------------------------------------------- */
typedef struct { v3 m[3]; v3 t; } affine;

static affine affine_scale(float s, v3 t)
{
    affine a = {{{s, 0, 0}, {0, s, 0}, {0, 0, s}}, t};
    return a;
}
static v3 affine_col(const affine *a, unsigned j)
{
    v3 r = {comp(a->m[0], j), comp(a->m[1], j), comp(a->m[2], j)};
    return r;
}
static affine affine_mul(const affine *l, const affine *r)
{
    /* util.hpp:270-281: matrix[i][j] = dot(l.row(i), r.col(j)); t[i] = dot(l.row(i), r.t); t += l.t */
    affine o;
    float mm[3][3], tt[3];
    for (unsigned i = 0; i < 3; ++i) {
        for (unsigned j = 0; j < 3; ++j) mm[i][j] = v3dot(l->m[i], affine_col(r, j));
        tt[i] = v3dot(l->m[i], r->t);
    }
    for (unsigned i = 0; i < 3; ++i) {
        o.m[i].x = mm[i][0];
        o.m[i].y = mm[i][1];
        o.m[i].z = mm[i][2];
    }
    o.t.x = tt[0] + l->t.x;
    o.t.y = tt[1] + l->t.y;
    o.t.z = tt[2] + l->t.z;
    return o;
}
static v3 affine_apply(const affine *a, v3 v)
{
    /* util.hpp:262-268 */
    float x = v3dot(a->m[0], v), y = v3dot(a->m[1], v), z = v3dot(a->m[2], v);
    v3 r = {x + a->t.x, y + a->t.y, z + a->t.z};
    return r;
}

/* computeMeshTransform, obj2voxel.cpp:370-402 */
static affine compute_mesh_transform(v3 mesh_min, v3 mesh_max, uint32_t sample_res, const int unit[9])
{
    const float ANTI_BLEED = 0.5f;
    v3 size = v3sub(mesh_max, mesh_min);
    float max_axis = fmax2(size.x, fmax2(size.y, size.z)); /* std::max(a, std::max(b, c)) util.hpp:94-96 */
    float sample_scale = (float) sample_res - ANTI_BLEED;

    v3 neg_min = {-mesh_min.x, -mesh_min.y, -mesh_min.z};
    v3 neg_one = {-1.f, -1.f, -1.f};
    v3 one = {1.f, 1.f, 1.f};
    v3 half_bleed = {ANTI_BLEED / 2, ANTI_BLEED / 2, ANTI_BLEED / 2};

    affine result = affine_scale(1, neg_min);
    affine a2 = affine_scale(2.f / max_axis, neg_one);
    result = affine_mul(&a2, &result);
    affine u;
    for (unsigned i = 0; i < 3; ++i) {
        u.m[i].x = (float) unit[i * 3 + 0];
        u.m[i].y = (float) unit[i * 3 + 1];
        u.m[i].z = (float) unit[i * 3 + 2];
    }
    u.t = one;
    result = affine_mul(&u, &result);
    affine a4 = affine_scale(sample_scale / 2, half_bleed);
    result = affine_mul(&a4, &result);
    return result;
}

/* exported so that host-logic tests can compare the product's transform float-for-float */
void o2v_oracle_mesh_transform(const float bounds[6], uint32_t sample_res, const int unit[9], float out12[12])
{
    static const int ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    v3 mn = {bounds[0], bounds[1], bounds[2]}, mx = {bounds[3], bounds[4], bounds[5]};
    affine a = compute_mesh_transform(mn, mx, sample_res, unit ? unit : ident);
    for (unsigned i = 0; i < 3; ++i) {
        out12[i * 3 + 0] = a.m[i].x;
        out12[i * 3 + 1] = a.m[i].y;
        out12[i * 3 + 2] = a.m[i].z;
    }
    out12[9] = a.t.x;
    out12[10] = a.t.y;
    out12[11] = a.t.z;
}

/* ---- triangle splitting (voxelization.cpp:110-331) -------------------------------------------------------- */
typedef struct { ttri d[64]; unsigned n; } splitbuf; /* ArrayVector<TexturedTriangle,64>, voxelization.hpp:57 */

static inline int is_zero(float x) { return fabsf(x) < O2V_EPS; } /* voxelization.cpp:17-20 */

/* voxelization.cpp:27-31 */
static inline float intersect_ray_axis_plane(v3 org, v3 dir, unsigned axis, uint32_t plane)
{
    float d = -comp(dir, axis);
    return is_zero(d) ? 0 : (comp(org, axis) - (float) plane) / d;
}

/* keep_lo = 1 means DISCARD_HI (pieces sorted to "lo" are kept), 0 means DISCARD_LO (voxelization.cpp:85-106) */
static inline void push_piece(splitbuf *out, const ttri *t, int lo, int keep_lo)
{
    if ((lo != 0) == (keep_lo != 0)) out->d[out->n++] = *t;
}

static void split_triangle(unsigned axis, uint32_t plane, const ttri *t, splitbuf *out, int keep_lo)
{
    /* SplittingValues, voxelization.cpp:110-153 */
    int lo[3], planar[3];
    unsigned lo_sum = 0, planar_sum = 0;
    const float fplane = (float) plane;
    t_stats.splits++;
    for (unsigned i = 0; i < 3; ++i) {
        const float c = comp(t->v[i], axis);
        planar[i] = is_zero(c - fplane);
        lo[i] = c < fplane; /* IS_LO_BIASED = false, voxelization.cpp:108 */
        planar_sum += (unsigned) planar[i];
        lo_sum += (unsigned) lo[i];
    }
    /* voxelization.cpp:190-232 */
    if (lo_sum == 0) { push_piece(out, t, 0, keep_lo); return; }
    if (lo_sum == 3) { push_piece(out, t, 1, keep_lo); return; }
    if (planar_sum == 3) { push_piece(out, t, 0, keep_lo); return; } /* parallel: pushed by bias = hi */
    if (planar_sum == 2) {
        unsigned np = !planar[0] ? 0u : !planar[1] ? 1u : 2u; /* firstNonplanar */
        push_piece(out, t, lo[np], keep_lo);
        return;
    }
    if (planar_sum == 1) {
        /* splitTriangle_onePlanarCase, voxelization.cpp:240-277 */
        unsigned pi = planar[0] ? 0u : planar[1] ? 1u : 2u;
        unsigned n0 = (pi + 1) % 3, n1 = (pi + 2) % 3;
        unsigned np_lo_sum = (unsigned) lo[n0] + (unsigned) lo[n1];
        if (np_lo_sum != 1) {
            push_piece(out, t, np_lo_sum == 2, keep_lo);
            return;
        }
        v3 pv = t->v[pi];
        v2 pt = t->t[pi];
        v3 nv0 = t->v[n0], nv1 = t->v[n1];
        v2 nt0 = t->t[n0], nt1 = t->t[n1];
        v3 edge = v3sub(nv1, nv0);
        float isect = intersect_ray_axis_plane(nv0, edge, axis, plane);
        v3 geo = v3mix(nv0, nv1, isect);
        v2 tex = v2mix(nt0, nt1, isect);
        ttri a = {{pv, nv0, geo}, {pt, nt0, tex}};
        ttri b = {{pv, geo, nv1}, {pt, tex, nt1}};
        int first_lo = lo[n0];
        push_piece(out, &a, first_lo, keep_lo);
        push_piece(out, &b, !first_lo, keep_lo);
        return;
    }
    {
        /* splitTriangle_regularCase, voxelization.cpp:279-331 */
        int iso_lo = lo_sum == 1;
        unsigned iso = iso_lo ? (lo[0] ? 0u : lo[1] ? 1u : 2u) : (!lo[0] ? 0u : !lo[1] ? 1u : 2u);
        unsigned o0 = (iso + 1) % 3, o1 = (iso + 2) % 3;
        v3 iv = t->v[iso];
        v2 it = t->t[iso];
        v3 ov0 = t->v[o0], ov1 = t->v[o1];
        v2 ot0 = t->t[o0], ot1 = t->t[o1];
        v3 e0 = v3sub(ov0, iv), e1 = v3sub(ov1, iv);
        float i0 = intersect_ray_axis_plane(iv, e0, axis, plane);
        float i1 = intersect_ray_axis_plane(iv, e1, axis, plane);
        v3 g0 = v3mix(iv, ov0, i0), g1 = v3mix(iv, ov1, i1);
        v2 x0 = v2mix(it, ot0, i0), x1 = v2mix(it, ot1, i1);
        ttri isolated = {{iv, g0, g1}, {it, x0, x1}};
        ttri other0 = {{g0, ov0, ov1}, {x0, ot0, ot1}};
        ttri other1 = {{g0, g1, ov1}, {x0, x1, ot1}};
        push_piece(out, &isolated, iso_lo, keep_lo);
        push_piece(out, &other0, !iso_lo, keep_lo);
        push_piece(out, &other1, !iso_lo, keep_lo);
    }
}

/* ---- weighted values (util.hpp:150-172) ------------------------------------------------------------------- */
typedef struct { float w; v2 uv; } wuv;
typedef struct { float w; v3 c; } wcol;

static inline wuv wuv_mix(wuv l, wuv r)
{
    float ws = l.w + r.w;
    wuv o;
    o.w = ws;
    o.uv.x = (l.w * l.uv.x + r.w * r.uv.x) / ws;
    o.uv.y = (l.w * l.uv.y + r.w * r.uv.y) / ws;
    return o;
}
static inline wcol wcol_mix(wcol l, wcol r)
{
    float ws = l.w + r.w;
    wcol o;
    o.w = ws;
    o.c.x = (l.w * l.c.x + r.w * r.c.x) / ws;
    o.c.y = (l.w * l.c.y + r.w * r.c.y) / ws;
    o.c.z = (l.w * l.c.z + r.w * r.c.z) / ws;
    return o;
}
static inline wcol wcol_max(wcol l, wcol r) { return l.w > r.w ? l : r; }
static inline wcol wcol_combine(unsigned strategy, wcol fresh, wcol existing)
{
    /* combineFunction(color, location->second): lhs = new, rhs = existing (voxelization.cpp:56-63,520-523) */
    return strategy == STRAT_BLEND ? wcol_mix(fresh, existing) : wcol_max(fresh, existing);
}

/* computeTrianglesUvInVoxel, voxelization.cpp:383-424 */
static wuv triangles_uv_in_voxel(float input_area, const ttri *sub, const uint32_t pos[3], splitbuf *pre,
                                 splitbuf *post)
{
    wuv zero = {0, {0, 0}};
    pre->n = 0;
    post->n = 0;
    pre->d[pre->n++] = *sub;
    t_stats.clips++;
    for (unsigned hi = 0; hi < 2; ++hi) {
        for (unsigned axis = 0; axis < 3; ++axis) {
            uint32_t plane = pos[axis] + hi;
            for (unsigned i = 0; i < pre->n; ++i) split_triangle(axis, plane, &pre->d[i], post, (int) hi);
            pre->n = 0;
            if (post->n == 0) return zero;
            splitbuf *tmp = pre;
            pre = post;
            post = tmp;
        }
    }
    wuv result = zero;
    for (unsigned i = 0; i < pre->n; ++i) {
        const ttri *t = &pre->d[i];
        wuv piece;
        piece.w = input_area;
        piece.uv.x = ((t->t[0].x + t->t[1].x) + t->t[2].x) / 3; /* textureCenter, triangle.hpp:127-130 */
        piece.uv.y = ((t->t[0].y + t->t[1].y) + t->t[2].y) / 3;
        result = wuv_mix(result, piece);
    }
    t_stats.pieces += pre->n;
    return result;
}

/* ---- per-chunk voxelizer state (Voxelizer, voxelization.hpp:55-108) --------------------------------------- */
/* The reference keys unordered_maps by Morton index; a chunk is 64^3 so dense arrays indexed by the local
 * position are an equivalent container (iteration order is not observable: every key is combined
 * independently, voxelization.cpp:513-526). */
#define CHUNK_CELLS (O2V_CHUNK * O2V_CHUNK * O2V_CHUNK)
typedef struct {
    wcol *voxels;       /* CHUNK_CELLS */
    uint8_t *has_voxel; /* CHUNK_CELLS */
    uint32_t *voxel_list;
    uint32_t voxel_count;
    wuv *uvbuf;         /* CHUNK_CELLS */
    uint32_t *uv_stamp; /* CHUNK_CELLS; == serial of the triangle that touched it */
    uint32_t *uv_list;
    uint32_t uv_count;
    uint32_t serial;
    ttri *stack;
    size_t stack_cap;
    splitbuf *pre, *post;
} voxelizer;

static voxelizer *voxelizer_new(void)
{
    voxelizer *vz = (voxelizer *) calloc(1, sizeof(voxelizer));
    vz->voxels = (wcol *) malloc(sizeof(wcol) * CHUNK_CELLS);
    vz->has_voxel = (uint8_t *) calloc(CHUNK_CELLS, 1);
    vz->voxel_list = (uint32_t *) malloc(sizeof(uint32_t) * CHUNK_CELLS);
    vz->uvbuf = (wuv *) malloc(sizeof(wuv) * CHUNK_CELLS);
    vz->uv_stamp = (uint32_t *) calloc(CHUNK_CELLS, sizeof(uint32_t));
    vz->uv_list = (uint32_t *) malloc(sizeof(uint32_t) * CHUNK_CELLS);
    vz->stack_cap = 64;
    vz->stack = (ttri *) malloc(sizeof(ttri) * vz->stack_cap);
    vz->pre = (splitbuf *) malloc(sizeof(splitbuf));
    vz->post = (splitbuf *) malloc(sizeof(splitbuf));
    return vz;
}
static void voxelizer_free(voxelizer *vz)
{
    free(vz->voxels);
    free(vz->has_voxel);
    free(vz->voxel_list);
    free(vz->uvbuf);
    free(vz->uv_stamp);
    free(vz->uv_list);
    free(vz->stack);
    free(vz->pre);
    free(vz->post);
    free(vz);
}

/* voxelizeSubTriangle, voxelization.cpp:426-472 */
static void voxelize_sub_triangle(voxelizer *vz, float input_area, const ttri *sub, const uint32_t cmin[3],
                                  const uint32_t cmax[3])
{
    const float distance_limit = 2;
    v3 org = sub->v[0];
    v3 nrm = v3normalize(tri_normal(sub));
    uint32_t lo[3], hi[3];
    tri_voxel_bounds(sub, lo, hi);
    for (unsigned i = 0; i < 3; ++i) {
        if (lo[i] < cmin[i]) lo[i] = cmin[i];
        if (hi[i] > cmax[i]) hi[i] = cmax[i];
    }
    t_stats.leaves++;
    g_trace_leaf++;
    for (uint32_t z = lo[2]; z < hi[2]; ++z)
        for (uint32_t y = lo[1]; y < hi[1]; ++y)
            for (uint32_t x = lo[0]; x < hi[0]; ++x) {
                uint32_t pos[3] = {x, y, z};
                t_stats.candidates++;
                {
                    /* ENABLE_PLANE_DISTANCE_TEST, voxelization.cpp:451-458 */
                    v3 center = {(float) x + 0.5f, (float) y + 0.5f, (float) z + 0.5f};
                    float sd = v3dot(nrm, v3sub(center, org));
                    if (fabsf(sd) > distance_limit) {
                        t_stats.culled++;
                        continue;
                    }
                }
                wuv uv = triangles_uv_in_voxel(input_area, sub, pos, vz->pre, vz->post);
                if (uv.w != 0.f) {
                    /* insertWeighted<BLEND>(uvBuffer, pos, uv), voxelization.cpp:56-63,466-468 */
                    uint32_t li = ((z - cmin[2]) * O2V_CHUNK + (y - cmin[1])) * O2V_CHUNK + (x - cmin[0]);
                    t_stats.hits++;
                    if (g_trace_on && x == g_trace_pos[0] && y == g_trace_pos[1] && z == g_trace_pos[2])
                        printf("oracle hit tri=%llu leafseq=%u w=%a (%.9g) u=%a v=%a area=%a\n", (unsigned long long) g_trace_tri,
                               g_trace_leaf, uv.w, uv.w, uv.uv.x, uv.uv.y, input_area);
                    if (vz->uv_stamp[li] != vz->serial) {
                        vz->uv_stamp[li] = vz->serial;
                        vz->uvbuf[li] = uv;
                        vz->uv_list[vz->uv_count++] = li;
                    }
                    else {
                        vz->uvbuf[li] = wuv_mix(uv, vz->uvbuf[li]);
                    }
                }
            }
}

/* isRoughlyAlignedWithAnyAxisPlane, voxelization.cpp:335-347 */
static int roughly_axis_aligned(const ttri *t)
{
    const float s = 0.5773502691896257645091487805019574556476017512701268760186023264f;
    v3 n = tri_normal(t);
    v3 an = {fabsf(n.x), fabsf(n.y), fabsf(n.z)};
    v3 nn = v3normalize(an);
    v3 diag = {s, s, s};
    float d = v3dot(nn, diag);
    float d01 = (d - s) / (1 - s);
    return d01 < O2V_DIAG_LIMIT;
}

/* subdivide4, triangle.hpp:134-143 */
static void subdivide4(const ttri *t, ttri out[4])
{
    v3 g0 = v3mix(t->v[0], t->v[1], 0.5f), g1 = v3mix(t->v[1], t->v[2], 0.5f), g2 = v3mix(t->v[2], t->v[0], 0.5f);
    v2 x0 = v2mix(t->t[0], t->t[1], 0.5f), x1 = v2mix(t->t[1], t->t[2], 0.5f), x2 = v2mix(t->t[2], t->t[0], 0.5f);
    ttri o0 = {{g0, g1, g2}, {x0, x1, x2}};
    ttri o1 = {{t->v[0], g0, g2}, {t->t[0], x0, x2}};
    ttri o2 = {{t->v[1], g1, g0}, {t->t[1], x1, x0}};
    ttri o3 = {{t->v[2], g2, g1}, {t->t[2], x2, x1}};
    out[0] = o0;
    out[1] = o1;
    out[2] = o2;
    out[3] = o3;
}

typedef struct {
    ttri geo;
    uint32_t type;
    v3 color;
    int32_t tex;
    uint32_t chunk_min[3], chunk_max[3];
} cached_tri; /* CachedTriangle, obj2voxel.cpp:122-132 */

/* colorAt_f, triangle.hpp:181-194; texture get, triangle.hpp:161-166 (getPixel semantics: see header) */
static v3 color_at(const cached_tri *tri, v2 uv, const o2v_oracle_texture *textures)
{
    v3 white = {1, 1, 1}, magenta = {1, 0, 1};
    switch (tri->type) {
    case TRI_MATERIALLESS: return white;
    case TRI_UNTEXTURED: return tri->color;
    case TRI_TEXTURED: {
        const o2v_oracle_texture *tex = &textures[tri->tex];
        float u = uv.x, v = 1 - uv.y;
        if (tex->wrap) {
            u = u - floorf(u);
            v = v - floorf(v);
        }
        else {
            u = u < 0.f ? 0.f : (u > 1.f ? 1.f : u);
            v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
        }
        uint32_t px = (uint32_t) (u * (float) tex->width), py = (uint32_t) (v * (float) tex->height);
        if (px >= tex->width) px = tex->width - 1;
        if (py >= tex->height) py = tex->height - 1;
        const uint8_t *p = tex->pixels + ((size_t) py * tex->width + px) * tex->channels;
        unsigned o = tex->channels == 4 ? 1u : 0u; /* ARGB: skip alpha */
        v3 c = {(float) p[o] / 255.f, (float) p[o + 1] / 255.f, (float) p[o + 2] / 255.f};
        return c;
    }
    default: return magenta;
    }
}

/* Voxelizer::voxelize, voxelization.cpp:480-526 */
static void voxelizer_voxelize(voxelizer *vz, const cached_tri *tri, const uint32_t cmin[3], const uint32_t cmax[3],
                               unsigned strategy, const o2v_oracle_texture *textures)
{
    const float input_area = tri_area(&tri->geo); /* voxelization.cpp:416 evaluates this per piece; same value */
    g_trace_leaf = 0;
    if (++vz->serial == 0u) {
        /* (a voxelizer lives as long as its thread's slot in the harness: after 2^32 triangles the stamps start over) */
        memset(vz->uv_stamp, 0, sizeof(uint32_t) * CHUNK_CELLS);
        vz->serial = 1u;
    }
    vz->uv_count = 0;

    if (roughly_axis_aligned(&tri->geo)) {
        voxelize_sub_triangle(vz, input_area, &tri->geo, cmin, cmax);
    }
    else {
        /* forEachSubdividedTriangle, voxelization.cpp:349-379 */
        size_t n = 0;
        vz->stack[n++] = tri->geo;
        do {
            ttri *top = &vz->stack[n - 1];
            uint32_t lo[3], hi[3];
            tri_voxel_bounds(top, lo, hi);
            uint32_t volume = (hi[0] - lo[0]) * (hi[1] - lo[1]) * (hi[2] - lo[2]); /* u32, wraps */
            if (volume < O2V_SUBDIV_LIMIT) {
                ttri leaf = *top;
                --n;
                voxelize_sub_triangle(vz, input_area, &leaf, cmin, cmax);
                continue;
            }
            ttri sub[4];
            subdivide4(top, sub);
            *top = sub[0];
            if (n + 3 > vz->stack_cap) {
                vz->stack_cap *= 2;
                vz->stack = (ttri *) realloc(vz->stack, sizeof(ttri) * vz->stack_cap);
            }
            vz->stack[n++] = sub[1];
            vz->stack[n++] = sub[2];
            vz->stack[n++] = sub[3];
        } while (n != 0);
    }

    /* moveUvBufferIntoVoxels, voxelization.cpp:513-526 */
    for (uint32_t i = 0; i < vz->uv_count; ++i) {
        uint32_t li = vz->uv_list[i];
        wuv w = vz->uvbuf[li];
        wcol c;
        c.w = w.w;
        c.c = color_at(tri, w.uv, textures);
        if (!vz->has_voxel[li]) {
            vz->has_voxel[li] = 1;
            vz->voxels[li] = c;
            vz->voxel_list[vz->voxel_count++] = li;
        }
        else {
            vz->voxels[li] = wcol_combine(strategy, c, vz->voxels[li]);
        }
    }
}

/* Color32{rgb}.argb(), obj2voxel.cpp:294-295 (conversion semantics: see header) */
static uint32_t pack_argb(v3 c)
{
    uint32_t r = (uint8_t) (c.x * 255), g = (uint8_t) (c.y * 255), b = (uint8_t) (c.z * 255);
    return 0xFF000000u | (r << 16) | (g << 8) | b;
}

typedef struct { uint32_t *d; size_t n, cap; } outvec;
static void out_push(outvec *o, uint32_t x, uint32_t y, uint32_t z, uint32_t argb)
{
    if (o->n == o->cap) {
        o->cap = o->cap ? o->cap * 2 : 4096;
        o->d = (uint32_t *) realloc(o->d, o->cap * 4 * sizeof(uint32_t));
    }
    uint32_t *p = o->d + o->n * 4;
    p[0] = x;
    p[1] = y;
    p[2] = z;
    p[3] = argb;
    o->n++;
}

/* voxelizeChunk, obj2voxel.cpp:254-314: all triangles of one 64^3 chunk, optional downscale, pack */
static void voxelize_chunk(voxelizer *vz, outvec *ov, const cached_tri *tris, const uint32_t *chunk_items, uint64_t k0,
                           uint64_t k1, uint32_t cx, uint32_t cy, uint32_t cz, unsigned strategy, uint32_t supersampling,
                           const o2v_oracle_texture *textures, uint32_t zlo, uint32_t zhi)
{
    uint32_t cmin[3] = {cx * O2V_CHUNK, cy * O2V_CHUNK, cz * O2V_CHUNK};
    uint32_t cmax[3] = {cmin[0] + O2V_CHUNK, cmin[1] + O2V_CHUNK, cmin[2] + O2V_CHUNK};
    vz->voxel_count = 0;
    for (uint64_t k = k0; k < k1; ++k) {
        g_trace_tri = chunk_items[k];
        voxelizer_voxelize(vz, &tris[chunk_items[k]], cmin, cmax, strategy, textures);
    }

    if (supersampling > 1) {
        /* documented downscale semantics (see header): 2x2x2 blocks, ascending sub order */
        for (uint32_t bz = 0; bz < O2V_CHUNK; bz += 2)
            for (uint32_t by = 0; by < O2V_CHUNK; by += 2)
                for (uint32_t bx = 0; bx < O2V_CHUNK; bx += 2) {
                    int have = 0;
                    wcol acc;
                    for (unsigned s = 0; s < 8; ++s) {
                        uint32_t lx = bx + (s & 1u), ly = by + ((s >> 1) & 1u), lz = bz + (s >> 2);
                        uint32_t li = (lz * O2V_CHUNK + ly) * O2V_CHUNK + lx;
                        if (!vz->has_voxel[li]) continue;
                        if (!have) {
                            acc = vz->voxels[li];
                            have = 1;
                        }
                        else {
                            acc = wcol_combine(strategy, vz->voxels[li], acc);
                        }
                    }
                    if (have) {
                        uint32_t ox = (cmin[0] + bx) / 2, oy = (cmin[1] + by) / 2,
                                 oz = (cmin[2] + bz) / 2;
                        if (zlo == zhi || (oz >= zlo && oz < zhi))
                            out_push(ov, ox, oy, oz, pack_argb(acc.c));
                    }
                }
    }
    else {
        for (uint32_t k = 0; k < vz->voxel_count; ++k) {
            uint32_t li = vz->voxel_list[k];
            uint32_t lx = li % O2V_CHUNK, ly = (li / O2V_CHUNK) % O2V_CHUNK,
                     lz = li / (O2V_CHUNK * O2V_CHUNK);
            uint32_t oz = cmin[2] + lz;
            if (zlo == zhi || (oz >= zlo && oz < zhi))
                out_push(ov, cmin[0] + lx, cmin[1] + ly, oz, pack_argb(vz->voxels[li].c));
        }
    }
    for (uint32_t k = 0; k < vz->voxel_count; ++k) vz->has_voxel[vz->voxel_list[k]] = 0;
}

/* Per-thread state of the harness, kept between calls like the reference's worker threads keep theirs between chunks
 * (obj2voxel.cpp:415-424): the voxelizer (10 MB of chunk-sized arrays) and the thread's output list.  Allocating and
 * releasing them in every call was most of a many-threaded call's time (256 threads: 2.6 GB of fresh pages per call, the page
 * faults and unmaps contending in the kernel; the chunk loop took 0.65 s where the single-threaded one takes 2.8 s). */
#define O2V_MAX_THREADS 1024
static voxelizer *g_vz_cache[O2V_MAX_THREADS];
static outvec g_out_cache[O2V_MAX_THREADS];

/* sortTriangleIntoChunks + the chunk loop (obj2voxel.cpp:226-243,254-314,503-505) for grids too fine for dense per-chunk
 * tables: every (chunk, triangle) pair is listed, the list sorted by chunk and triangle - ascending triangle order inside a
 * chunk, as the reference's serial loop leaves it (obj2voxel.cpp:489-491) - and the chunks voxelized one after the other by
 * the calling thread.  Test infrastructure for the resolutions above 65 535 only. */
typedef struct { uint64_t chunk; uint32_t tri; } o2v_bin_pair;
static int o2v_bin_pair_cmp(const void *a, const void *b)
{
    const o2v_bin_pair *p = (const o2v_bin_pair *) a, *q = (const o2v_bin_pair *) b;
    if (p->chunk != q->chunk) return p->chunk < q->chunk ? -1 : 1;
    return p->tri < q->tri ? -1 : (p->tri > q->tri ? 1 : 0);
}
static int64_t o2v_voxelize_sparse_bins(const cached_tri *tris, uint64_t T, uint32_t chunks_per_axis, uint32_t strategy,
                                        uint32_t supersampling, const o2v_oracle_texture *textures, uint32_t zlo, uint32_t zhi, outvec *ov)
{
    size_t n_pairs = 0, cap = 1024;
    o2v_bin_pair *pairs = (o2v_bin_pair *) malloc(cap * sizeof(o2v_bin_pair));
    for (uint64_t i = 0; i < T; ++i) {
        const cached_tri *t = &tris[i];
        for (uint32_t z = t->chunk_min[2]; z <= t->chunk_max[2] && z < chunks_per_axis; ++z)
            for (uint32_t y = t->chunk_min[1]; y <= t->chunk_max[1] && y < chunks_per_axis; ++y)
                for (uint32_t x = t->chunk_min[0]; x <= t->chunk_max[0] && x < chunks_per_axis; ++x) {
                    if (n_pairs == cap) {
                        cap *= 2;
                        pairs = (o2v_bin_pair *) realloc(pairs, cap * sizeof(o2v_bin_pair));
                    }
                    pairs[n_pairs].chunk = ((uint64_t) z * chunks_per_axis + y) * chunks_per_axis + x;
                    pairs[n_pairs].tri = (uint32_t) i;
                    ++n_pairs;
                }
    }
    qsort(pairs, n_pairs, sizeof(o2v_bin_pair), o2v_bin_pair_cmp);
    uint32_t *items = (uint32_t *) malloc(sizeof(uint32_t) * (n_pairs ? n_pairs : 1));
    for (size_t k = 0; k < n_pairs; ++k) items[k] = pairs[k].tri;
    if (!g_vz_cache[0]) g_vz_cache[0] = voxelizer_new();
    voxelizer *vz = g_vz_cache[0];
    outvec lov = g_out_cache[0];
    lov.n = 0;
    memset(&t_stats, 0, sizeof(t_stats));
    for (size_t k0 = 0; k0 < n_pairs;) {
        size_t k1 = k0;
        while (k1 < n_pairs && pairs[k1].chunk == pairs[k0].chunk) ++k1;
        const uint64_t c = pairs[k0].chunk;
        const uint32_t cx = (uint32_t) (c % chunks_per_axis), cy = (uint32_t) ((c / chunks_per_axis) % chunks_per_axis),
                       cz = (uint32_t) (c / ((uint64_t) chunks_per_axis * chunks_per_axis));
        int skip = 0;
        if (zlo != zhi) {
            const uint32_t oz0 = cz * O2V_CHUNK / supersampling, oz1 = (cz * O2V_CHUNK + O2V_CHUNK - 1u) / supersampling;
            skip = oz1 < zlo || oz0 >= zhi;
        }
        if (!skip) voxelize_chunk(vz, &lov, tris, items, k0, k1, cx, cy, cz, strategy, supersampling, textures, zlo, zhi);
        k0 = k1;
    }
    g_out_cache[0] = lov;
    {
        uint64_t *dst = (uint64_t *) &g_stats;
        const uint64_t *src = (const uint64_t *) &t_stats;
        for (size_t k = 0; k < sizeof(g_stats) / sizeof(uint64_t); ++k) dst[k] += src[k];
    }
    ov->d = (uint32_t *) malloc(sizeof(uint32_t) * 4 * (lov.n ? lov.n : 1));
    ov->n = ov->cap = lov.n;
    if (lov.n) memcpy(ov->d, lov.d, sizeof(uint32_t) * 4 * lov.n);
    free(items);
    free(pairs);
    return (int64_t) ov->n;
}

/*
 * The whole path: cache -> bounds -> transform -> chunk binning -> per chunk voxelize (+downscale) -> pack.
 * obj2voxel.cpp:467-520 (voxelize_specialized<false>), :180-314.
 *
 *  verts   [T][9]   model-space vertices
 *  uvs     [T][6]   or NULL (zeros, as CachedTriangle triangle{} zero-initialises, obj2voxel.cpp:585)
 *  types   [T]      or NULL (all MATERIALLESS)
 *  colors  [T][3]   or NULL
 *  texids  [T]      or NULL; index into textures[]
 *  unit    int[9]   or NULL (identity)
 *  bounds  float[6] or NULL (computed from the mesh)
 *  zlo,zhi          output-resolution z-slab filter [zlo,zhi); (0,0) = everything.  Used by the multi-GPU
 *                   tests: a slab must equal the matching subset of the full run.
 *  out              malloc'd (x,y,z,argb) quadruples, free with o2v_oracle_free
 * returns the voxel count.
 */
int64_t o2v_oracle_voxelize(const float *verts, const float *uvs, const uint32_t *types, const float *colors,
                            const int32_t *texids, uint64_t T, const o2v_oracle_texture *textures,
                            uint32_t resolution, uint32_t supersampling, uint32_t strategy, const int *unit,
                            const float *bounds, uint32_t zlo, uint32_t zhi, uint32_t **out)
{
    static const int ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    outvec ov = {0, 0, 0};
    memset(&g_stats, 0, sizeof(g_stats));
    *out = NULL;
    if (T == 0 || resolution == 0) return 0;
    if (supersampling == 0) supersampling = 1;
    const uint32_t sample_res = resolution * supersampling; /* obj2voxel.cpp:684-698 */
    const uint32_t chunks_per_axis = (sample_res + O2V_CHUNK - 1) / O2V_CHUNK; /* :580-581 */

    const double t_begin = omp_get_wtime();
    /* the memory-bound steps around the chunk loop use a few threads only: with one thread per core of a large host the page
     * faults of the freshly allocated arrays contend in the kernel and the steps get slower than serial (measured: 0.58 s
     * with 256 threads against 0.035 s with one, 870 k triangles) */
    const int aux_threads = g_threads < 4 ? g_threads : 4;
    cached_tri *tris = (cached_tri *) calloc(T, sizeof(cached_tri));
    /* (the harness around the reference's algorithm is parallel too - copy, bounds, transform, binning, merge - so that the
     * multi-threaded baseline measures the algorithm, not a serial prelude; results do not depend on the thread count) */
#pragma omp parallel for num_threads(aux_threads) schedule(static)
    for (int64_t i = 0; i < (int64_t) T; ++i) {
        const float *p = verts + i * 9;
        for (unsigned k = 0; k < 3; ++k) {
            tris[i].geo.v[k].x = p[k * 3 + 0];
            tris[i].geo.v[k].y = p[k * 3 + 1];
            tris[i].geo.v[k].z = p[k * 3 + 2];
            if (uvs) {
                tris[i].geo.t[k].x = uvs[i * 6 + k * 2 + 0];
                tris[i].geo.t[k].y = uvs[i * 6 + k * 2 + 1];
            }
        }
        tris[i].type = types ? types[i] : TRI_MATERIALLESS;
        if (colors) {
            tris[i].color.x = colors[i * 3 + 0];
            tris[i].color.y = colors[i * 3 + 1];
            tris[i].color.z = colors[i * 3 + 2];
        }
        tris[i].tex = texids ? texids[i] : 0;
    }

    /* findMeshBounds in batches of 1024, obj2voxel.cpp:180-200,475-480 */
    v3 mesh_min = {INFINITY, INFINITY, INFINITY}, mesh_max = {-INFINITY, -INFINITY, -INFINITY};
    if (bounds) {
        mesh_min.x = bounds[0];
        mesh_min.y = bounds[1];
        mesh_min.z = bounds[2];
        mesh_max.x = bounds[3];
        mesh_max.y = bounds[4];
        mesh_max.z = bounds[5];
    }
    else {
        /* batches are merged under a mutex in the reference (boundsMutex); min / max are exact and order-free */
        const int64_t n_batches = (int64_t) ((T + O2V_BATCH - 1) / O2V_BATCH);
#pragma omp parallel for num_threads(aux_threads) schedule(static)
        for (int64_t bi = 0; bi < n_batches; ++bi) {
            const uint64_t b = (uint64_t) bi * O2V_BATCH;
            uint64_t end = b + O2V_BATCH < T ? b + O2V_BATCH : T;
            v3 mn = {INFINITY, INFINITY, INFINITY}, mx = {-INFINITY, -INFINITY, -INFINITY};
            for (uint64_t i = b; i < end; ++i) {
                v3 tmn = tri_min(&tris[i].geo), tmx = tri_max(&tris[i].geo);
                mn.x = fmin2(tmn.x, mn.x);
                mn.y = fmin2(tmn.y, mn.y);
                mn.z = fmin2(tmn.z, mn.z);
                mx.x = fmax2(tmx.x, mx.x);
                mx.y = fmax2(tmx.y, mx.y);
                mx.z = fmax2(tmx.z, mx.z);
            }
#pragma omp critical(o2v_bounds)
            {
                mesh_min.x = fmin2(mesh_min.x, mn.x);
                mesh_min.y = fmin2(mesh_min.y, mn.y);
                mesh_min.z = fmin2(mesh_min.z, mn.z);
                mesh_max.x = fmax2(mesh_max.x, mx.x);
                mesh_max.y = fmax2(mesh_max.y, mx.y);
                mesh_max.z = fmax2(mesh_max.z, mx.z);
            }
        }
    }

    affine xf = compute_mesh_transform(mesh_min, mesh_max, sample_res, unit ? unit : ident);

    /* applyMeshTransform, obj2voxel.cpp:202-224 */
#pragma omp parallel for num_threads(aux_threads) schedule(static)
    for (int64_t i = 0; i < (int64_t) T; ++i) {
        for (unsigned k = 0; k < 3; ++k) tris[i].geo.v[k] = affine_apply(&xf, tris[i].geo.v[k]);
        uint32_t lo[3], hi[3];
        tri_voxel_bounds(&tris[i].geo, lo, hi);
        for (unsigned a = 0; a < 3; ++a) {
            tris[i].chunk_min[a] = lo[a] / O2V_CHUNK;
            tris[i].chunk_max[a] = (hi[a] - 1u) / O2V_CHUNK;
        }
    }

    /* sortTriangleIntoChunks, obj2voxel.cpp:226-243: ascending triangle order inside each chunk (CSR here).  The triangle
     * list is cut into contiguous ranges, one per thread: every range counts its entries per chunk, a prefix over (chunk,
     * range) gives each range its place in each chunk's list, and the ranges fill their places - ascending triangle order
     * inside a chunk follows from the order of the ranges, as in the reference's serial loop (obj2voxel.cpp:489-491). */
    const size_t nchunks = (size_t) chunks_per_axis * chunks_per_axis * chunks_per_axis;
    /* A grid of more than 2^27 chunks (resolutions beyond ~32 000) gets no dense per-chunk tables: the (chunk, triangle) pairs are
     * listed and sorted instead - the same lists in the same ascending triangle order (o2v_sparse_bins below). */
    const char *force_sparse = getenv("O2V_ORACLE_SPARSE_BINS"); /* (tests: the sparse path on a grid the dense path handles too) */
    if (nchunks > ((size_t) 1 << 27) || (force_sparse && force_sparse[0] == '1')) {
        const int64_t n = o2v_voxelize_sparse_bins(tris, T, chunks_per_axis, strategy, supersampling, textures, zlo, zhi, &ov);
        free(tris);
        *out = ov.d;
        return n;
    }
    uint64_t *chunk_start = (uint64_t *) calloc(nchunks + 1, sizeof(uint64_t));
    {
        int n_ranges = aux_threads;
        while (n_ranges > 1 && (size_t) n_ranges * nchunks > ((size_t) 1 << 26)) n_ranges /= 2;
        uint32_t *cnt = (uint32_t *) calloc((size_t) n_ranges * nchunks, sizeof(uint32_t));
        uint32_t *chunk_items = NULL;
        for (int pass = 0; pass < 2; ++pass) {
#pragma omp parallel for num_threads(aux_threads) schedule(static, 1)
            for (int r = 0; r < n_ranges; ++r) {
                const uint64_t i0 = T * (uint64_t) r / (uint64_t) n_ranges, i1 = T * (uint64_t) (r + 1) / (uint64_t) n_ranges;
                uint32_t *row = cnt + (size_t) r * nchunks;
                for (uint64_t i = i0; i < i1; ++i) {
                    const cached_tri *t = &tris[i];
                    for (uint32_t z = t->chunk_min[2]; z <= t->chunk_max[2] && z < chunks_per_axis; ++z)
                        for (uint32_t y = t->chunk_min[1]; y <= t->chunk_max[1] && y < chunks_per_axis; ++y)
                            for (uint32_t x = t->chunk_min[0]; x <= t->chunk_max[0] && x < chunks_per_axis; ++x) {
                                size_t c = ((size_t) z * chunks_per_axis + y) * chunks_per_axis + x;
                                if (pass == 0) row[c]++;
                                else chunk_items[chunk_start[c] + row[c]++] = (uint32_t) i;
                            }
                }
            }
            if (pass == 0) {
                /* counts -> places: chunk by chunk, range by range.  A place is relative to its chunk's start (32 bits bound one
                 * chunk's list at 2^32 entries; the starts, and so the total, are 64-bit) */
                uint64_t acc = 0;
                for (size_t c = 0; c < nchunks; ++c) {
                    chunk_start[c] = acc;
                    uint64_t in_chunk = 0;
                    for (int r = 0; r < n_ranges; ++r) {
                        const uint32_t n = cnt[(size_t) r * nchunks + c];
                        cnt[(size_t) r * nchunks + c] = (uint32_t) in_chunk;
                        in_chunk += n;
                    }
                    acc += in_chunk;
                }
                chunk_start[nchunks] = acc;
                chunk_items = (uint32_t *) malloc(sizeof(uint32_t) * (acc ? acc : 1));
            }
        }
        free(cnt);
        {
            g_phase_seconds[0] = omp_get_wtime() - t_begin;
            const double t_vox = omp_get_wtime();
            /* voxelizeChunk for every chunk, obj2voxel.cpp:254-314,503-505 */
            size_t n_work = 0;
            size_t *work = (size_t *) malloc(sizeof(size_t) * (nchunks ? nchunks : 1));
            for (size_t c = 0; c < nchunks; ++c) {
                if (chunk_start[c] == chunk_start[c + 1]) continue;
                if (zlo != zhi) {
                    /* a chunk whose output layers all lie outside the slab would only produce filtered-out voxels
                     * (chunks are independent, obj2voxel.cpp:254-314), so it is not voxelized at all */
                    const uint32_t cz = (uint32_t) (c / ((size_t) chunks_per_axis * chunks_per_axis));
                    const uint32_t oz0 = cz * O2V_CHUNK / supersampling, oz1 = (cz * O2V_CHUNK + O2V_CHUNK - 1u) / supersampling;
                    if (oz1 < zlo || oz0 >= zhi) continue;
                }
                work[n_work++] = c;
            }
            outvec *per_thread = (outvec *) calloc((size_t) g_threads, sizeof(outvec));
            double t_merge = 0.0;
#pragma omp parallel num_threads(g_threads)
            {
                const int tid = omp_get_thread_num();
                if (!g_vz_cache[tid]) g_vz_cache[tid] = voxelizer_new();
                voxelizer *vz = g_vz_cache[tid];
                outvec lov = g_out_cache[tid];
                lov.n = 0;
                memset(&t_stats, 0, sizeof(t_stats));
#pragma omp for schedule(dynamic, 1)
                for (long wi = 0; wi < (long) n_work; ++wi) {
                    const size_t c = work[wi];
                    const uint32_t cx = (uint32_t) (c % chunks_per_axis), cy = (uint32_t) ((c / chunks_per_axis) % chunks_per_axis),
                                   cz = (uint32_t) (c / ((size_t) chunks_per_axis * chunks_per_axis));
                    voxelize_chunk(vz, &lov, tris, chunk_items, chunk_start[c], chunk_start[c + 1], cx, cy, cz, strategy,
                                   supersampling, textures, zlo, zhi);
                }
                /* every thread keeps its own output; the lists are joined below by parallel copies (the sink of the
                 * reference takes each chunk's voxels under a mutex, obj2voxel.cpp:298-303: the order is unspecified) */
                per_thread[tid] = lov;
                g_out_cache[tid] = lov;
#pragma omp critical
                {
                    uint64_t *dst = (uint64_t *) &g_stats;
                    const uint64_t *src = (const uint64_t *) &t_stats;
                    for (size_t k = 0; k < sizeof(g_stats) / sizeof(uint64_t); ++k) dst[k] += src[k];
                }
#pragma omp barrier
#pragma omp single
                {
                    t_merge = omp_get_wtime();
                    size_t total = 0;
                    for (int k = 0; k < g_threads; ++k) total += per_thread[k].n;
                    ov.d = (uint32_t *) malloc(sizeof(uint32_t) * 4 * (total ? total : 1));
                    ov.n = ov.cap = total;
                }
#pragma omp barrier
                if (omp_get_thread_num() < aux_threads) {
                    /* a few threads copy all lists (see aux_threads) */
                    for (int k = omp_get_thread_num(); k < g_threads; k += aux_threads) {
                        size_t before = 0;
                        for (int j = 0; j < k; ++j) before += per_thread[j].n;
                        if (per_thread[k].n) memcpy(ov.d + before * 4, per_thread[k].d, sizeof(uint32_t) * 4 * per_thread[k].n);
                    }
                }
            }
            g_phase_seconds[1] = t_merge - t_vox;
            g_phase_seconds[2] = omp_get_wtime() - t_merge;
            free(per_thread);
            free(work);
            free(chunk_items);
        }
    }
    free(chunk_start);
    free(tris);
    *out = ov.d;
    return (int64_t) ov.n;
}

void o2v_oracle_free(uint32_t *p) { free(p); }

/* releases the per-thread state the harness keeps between calls */
void o2v_oracle_release(void)
{
    for (int k = 0; k < O2V_MAX_THREADS; ++k) {
        if (g_vz_cache[k]) voxelizer_free(g_vz_cache[k]);
        g_vz_cache[k] = NULL;
        free(g_out_cache[k].d);
        g_out_cache[k].d = NULL;
        g_out_cache[k].n = g_out_cache[k].cap = 0;
    }
}

/* wall seconds of the last o2v_oracle_voxelize call: [0] copy + bounds + transform + chunk binning, [1] the chunk loop
 * (the reference's algorithm), [2] joining the threads' output lists */
void o2v_oracle_get_phase_seconds(double out[3])
{
    for (int k = 0; k < 3; ++k) out[k] = g_phase_seconds[k];
}

void o2v_oracle_get_stats(o2v_oracle_stats *out) { *out = g_stats; }
