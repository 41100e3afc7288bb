// o2v_dev_k0_bounds_plan.hpp -- K0: mesh bounds + transform (k_init, k_bounds, k_setup) and the slab plan (k_zhist).
//
// Part of the device code of o2v_device.hip, which includes this file inside its anonymous namespace (one
// translation unit: the stages share records and launch parameters).  Not a stand-alone header.

// ---- K0: bounds + transform ---------------------------------------------------------------------------------

// (n_words: the pass's counters end where ext_hist begins - that kilobyte is only written and read at upload time)
constexpr uint32_t kPassCounterWords = offsetof(Counters, ext_hist) / 4;
__global__ void k_init(Counters *c, uint32_t n_words)
{
    uint32_t i = threadIdx.x;
    uint32_t *w = reinterpret_cast<uint32_t *>(c);
    for (uint32_t k = i; k < n_words; k += blockDim.x) w[k] = 0;
    __syncthreads();
    if (i < 3) c->bounds_enc[i] = f2ord(__builtin_inff());
    else if (i < 6) c->bounds_enc[i] = f2ord(-__builtin_inff());
}

// Sharded runs: the six bounds of this rank's triangles and its "not ready" word travel in ONE max-reduce over the ranks (the
// minima as their complements: the encoding is order preserving, so min x = ~max ~x).
constexpr uint32_t kReadyWords = 8;  // 3 complemented minima, 3 maxima, the status word, one spare
__global__ void k_pack_ready(const Counters *__restrict__ c, uint32_t *__restrict__ words, uint32_t status)
{
    const uint32_t i = threadIdx.x;
    if (i < 3) words[i] = ~c->bounds_enc[i];
    else if (i < 6) words[i] = c->bounds_enc[i];
    else if (i == 6) words[i] = status;
    else if (i == 7) words[i] = 0u;
}
__global__ void k_unpack_ready(const uint32_t *__restrict__ words, Counters *__restrict__ c)
{
    const uint32_t i = threadIdx.x;
    if (i < 3) c->bounds_enc[i] = ~words[i];
    else if (i < 6) c->bounds_enc[i] = words[i];
}

// Sharded runs: a rank's partial z histogram and the z extents of its blocks travel in ONE all-gather (a record per rank: kPlanBins
// sums, then `bpr` extents); every rank then adds the histograms up itself (integer sums: the same on every rank).
__global__ __launch_bounds__(kBlock) void k_pack_plan(const unsigned long long *__restrict__ hist, const float2 *__restrict__ my_extents,
                                                      unsigned long long *__restrict__ record, uint32_t bpr)
{
    float2 *ext = reinterpret_cast<float2 *>(record + 2048);
    for (uint32_t t = blockIdx.x * kBlock + threadIdx.x; t < 2048u + bpr; t += gridDim.x * kBlock) {
        if (t < 2048u) record[t] = hist[t];
        else ext[t - 2048u] = my_extents[t - 2048u];
    }
}
__global__ __launch_bounds__(kBlock) void k_unpack_plan(const unsigned long long *__restrict__ records, uint32_t world, uint32_t bpr,
                                                        unsigned long long *__restrict__ hist, float2 *__restrict__ extents)
{
    const uint64_t rec_words = 2048u + bpr;  // (a float2 is one 8-byte word)
    const uint64_t n = 2048u + (uint64_t) world * bpr;
    for (uint64_t t = (uint64_t) blockIdx.x * kBlock + threadIdx.x; t < n; t += (uint64_t) gridDim.x * kBlock) {
        if (t < 2048u) {
            unsigned long long sum = 0;
            for (uint32_t r = 0; r < world; ++r) sum += records[r * rec_words + t];
            hist[t] = sum;
        }
        else {
            const uint64_t j = t - 2048u;
            const uint32_t r = (uint32_t) (j / bpr), i = (uint32_t) (j % bpr);
            extents[j] = reinterpret_cast<const float2 *>(records + r * rec_words + 2048u)[i];
        }
    }
}

// findMeshBounds (obj2voxel.cpp:180-200): min/max are exact and order-free, so one reduce replaces the batches.
// The vertex array is streamed as float4 triples (12 floats = 4 vertices, so the axis of every element is static);
// one set of six atomics per workgroup.
// (1024-thread workgroups - four wavefronts per SIMD at one workgroup per CU - were measured: 13.9 us against 14.3 for the bench mesh's
// 31 MB; the kernel is short enough to be its launch, its tail and its 6 x 256 atomics)
constexpr uint32_t kBoundsBlock = kBlock;
__global__ __launch_bounds__(kBoundsBlock) void k_bounds(const float *__restrict__ verts, uint64_t n_floats, Counters *c)
{
    __shared__ float s_red[6][kBoundsBlock / 64];
    const float inf = __builtin_inff();
    float mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
    const uint64_t n_groups = n_floats / 12;
    const float4 *v4 = reinterpret_cast<const float4 *>(verts);
#pragma unroll 4
    for (uint64_t g = (uint64_t) blockIdx.x * kBoundsBlock + threadIdx.x; g < n_groups; g += (uint64_t) gridDim.x * kBoundsBlock) {
        const float4 a = v4[g * 3], b = v4[g * 3 + 1], d = v4[g * 3 + 2];
        const float e[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, d.x, d.y, d.z, d.w};
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            mn[k % 3] = fmin2(e[k], mn[k % 3]);
            mx[k % 3] = fmax2(e[k], mx[k % 3]);
        }
    }
    if (blockIdx.x == 0)
        for (uint64_t i = n_groups * 12 + threadIdx.x; i < n_floats; i += kBoundsBlock) {
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
        for (uint32_t w = 1; w < kBoundsBlock / 64; ++w) r = threadIdx.x < 3 ? fminf(r, s_red[threadIdx.x][w]) : fmaxf(r, s_red[threadIdx.x][w]);
        if (threadIdx.x < 3) atomicMin(&c->bounds_enc[threadIdx.x], f2ord(r));
        else atomicMax(&c->bounds_enc[threadIdx.x], f2ord(r));
    }
}

// Largest triangle extent (L-infinity, mesh space), computed once per upload: with the mesh extent it bounds the
// subdivision depth, i.e. how many k_expand_nodes rounds a voxelization has to launch (each is ~10 us even when empty).
// (also: the triangles counted by the binary exponent of their extent - hist[biased exponent], 256 bins - from which the host
// estimates, once the resolution is known, what share of the mesh will be subdivided: grid_modes)
__global__ __launch_bounds__(kBlock) void k_tri_extent(const float *__restrict__ verts, uint64_t n_tris, uint32_t *out_enc, uint32_t *hist)
{
    __shared__ float s_red[kBlock / 64];
    __shared__ uint32_t s_hist[256];
    s_hist[threadIdx.x] = 0;
    static_assert(kBlock == 256, "one bin per thread");
    __syncthreads();
    float ext = 0.f;
    for (uint64_t t = (uint64_t) blockIdx.x * kBlock + threadIdx.x; t < n_tris; t += (uint64_t) gridDim.x * kBlock) {
        const float *q = verts + t * 9;
        float mine = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float lo = fminf(q[a], fminf(q[3 + a], q[6 + a])), hi = fmaxf(q[a], fmaxf(q[3 + a], q[6 + a]));
            float e = hi - lo;
            if (!(e == e)) e = __builtin_inff();  // NaN: no bound
            mine = fmaxf(mine, e);
        }
        ext = fmaxf(ext, mine);
        atomicAdd(&s_hist[(__float_as_uint(mine) >> 23) & 255u], 1u);
    }
    __syncthreads();
    if (s_hist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_hist[threadIdx.x]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) ext = fmaxf(ext, __shfl_xor(ext, d, 64));
    if ((threadIdx.x & 63u) == 0) s_red[threadIdx.x >> 6] = ext;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t w = 1; w < kBlock / 64; ++w) ext = fmaxf(ext, s_red[w]);
        atomicMax(out_enc, f2ord(ext));
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
static_assert(kPlanBins == 2048, "k_pack_plan / k_unpack_plan are written for 2048 bins");
#ifndef O2V_PLAN_LEAF_COST
#define O2V_PLAN_LEAF_COST 4.0f
#endif
constexpr float kPlanLeafCost = O2V_PLAN_LEAF_COST;  // what a leaf costs beside its hits, in hits (see k_zhist)
// ... in occupancy-only mode, where most hits need no voxel job (and most of the jobs that remain are dropped), a hit is cheaper
// and the leaf's own cost - its transform, its candidate rows - weighs more.  Round 5's kernels on the eight slabs of configs[4]
// (sub-voxel triangles; polar slab 42.3 M hits / 5.63 M leaves 1.94 ms, equatorial 36.0 M / 7.20 M 2.25 ms) fit 19 hits per leaf,
// on those of the weak-scaling job (triangles 3.4 voxels across) 7; slowest slab / mean with 4, 16: configs[4] 1.069, 1.022, weak
// job 1.025, 1.050 (profiles/r05/NOTES.md).  10 serves both.
#ifndef O2V_PLAN_LEAF_COST_OCC
#define O2V_PLAN_LEAF_COST_OCC 10.0f
#endif
constexpr float kPlanLeafCostOccupancy = O2V_PLAN_LEAF_COST_OCC;
__global__ __launch_bounds__(kBlock) void k_zhist(const float *__restrict__ verts, const Counters *__restrict__ c,
                                                   unsigned long long *hist, float2 *zrange, float *zrange_xform,
                                                   Params p, uint32_t bin_h, uint64_t tri_begin, uint64_t tri_end)
{
    // [tri_begin, tri_end): this rank's share of the triangle list (a multiple of 256 at the lower end); a single GPU
    // takes the whole list
    __shared__ unsigned long long s_hist[kPlanBins];
    __shared__ float s_v[kBlock * 9];
    __shared__ float s_zr[2][kBlock / 64];
    if (blockIdx.x == 0 && threadIdx.x < 12) zrange_xform[threadIdx.x] = c->xform[threadIdx.x];
    for (uint32_t t = threadIdx.x; t < kPlanBins; t += kBlock) s_hist[t] = 0;
    Affine a;
    for (int i = 0; i < 3; ++i) a.m[i] = {c->xform[i * 3], c->xform[i * 3 + 1], c->xform[i * 3 + 2]};
    a.t = {c->xform[9], c->xform[10], c->xform[11]};
    // software pipeline: the next batch's vertices are already on their way (nine loads per lane, in registers) while
    // this batch is processed from LDS
    float pre[9];
    auto prefetch = [&](uint64_t base) {
        const uint32_t n_f = (uint32_t) (tri_end - base < kBlock ? tri_end - base : kBlock) * 9u;
#pragma unroll
        for (uint32_t k = 0; k < 9; ++k) {
            const uint32_t idx = threadIdx.x + k * kBlock;
            pre[k] = idx < n_f ? verts[base * 9 + idx] : 0.f;
        }
    };
    const uint64_t first = tri_begin + (uint64_t) blockIdx.x * kBlock, step = (uint64_t) gridDim.x * kBlock;
    if (first < tri_end) prefetch(first);
    for (uint64_t base = first; base < tri_end; base += step) {
        __syncthreads();
        const uint32_t n_here = (uint32_t) (tri_end - base < kBlock ? tri_end - base : kBlock);
#pragma unroll
        for (uint32_t k = 0; k < 9; ++k) s_v[threadIdx.x + k * kBlock] = pre[k];
        if (base + step < tri_end) prefetch(base + step);
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
        // Predicted TIME, in hit equivalents: the hits (Steiner-type estimate above) plus what a leaf costs whatever it hits -
        // its expansion, its tile, its candidate rows - measured on the eight slabs of configs[4] (equal hits, 5.3 M to 7.8 M
        // leaves: +0.127 us per leaf, against 0.031 us per hit: kPlanLeafCost hits per leaf; profiles/r03/predict_scaling_8_config4.jsonl).
        // A triangle whose voxel box has V >= 512 cells is subdivided (voxelization.cpp:349-379) into about (V / 512)^(2/3)
        // leaves (every level quarters the triangle and divides its box by about eight).
        const float bx = fmax2(v0.x, fmax2(v1.x, v2.x)) - fmin2(v0.x, fmin2(v1.x, v2.x)) + 1.0f;
        const float by = fmax2(v0.y, fmax2(v1.y, v2.y)) - fmin2(v0.y, fmin2(v1.y, v2.y)) + 1.0f;
        const float bz = fmax2(v0.z, fmax2(v1.z, v2.z)) - fmin2(v0.z, fmin2(v1.z, v2.z)) + 1.0f;
        const float boxes = bx * by * bz * (1.0f / 512.0f);
        const float leaves_est = boxes > 1.0f ? __builtin_exp2f(__builtin_log2f(boxes) * (2.0f / 3.0f)) : 1.0f;
        float est = (abs_f(n.x) + abs_f(n.y) + abs_f(n.z)) * 0.5f +
                    (abs_f(e0.x) + abs_f(e0.y) + abs_f(e0.z) + abs_f(e1.x) + abs_f(e1.y) + abs_f(e1.z) + abs_f(e2.x) +
                     abs_f(e2.y) + abs_f(e2.z)) * 0.5f + 1.0f + p.plan_leaf_cost * leaves_est;
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
