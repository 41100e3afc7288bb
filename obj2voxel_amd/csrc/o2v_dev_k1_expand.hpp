// o2v_dev_k1_expand.hpp -- K1: leaves - root triangles, exact subdivision rounds, tiles (k_expand_roots / _nodes / _big).
//
// Part of the device code of o2v_device.hip, which includes this file inside its anonymous namespace (one
// translation unit: the stages share records and launch parameters).  Not a stand-alone header.

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
    // (the grid's box in sample space: the [0, S)^2 x slab box of the reference's chunk, cut to the mesh's bounding box, which
    // no leaf leaves - Params::cs_lo)
    const uint32_t glo[3] = {p.cs_lo[0], p.cs_lo[1], p.cs_lo[2]}, ghi[3] = {p.cs_hi[0], p.cs_hi[1], p.cs_hi[2]};
    bool empty = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = lo[a] > glo[a] ? lo[a] : glo[a];
        hi[a] = hi[a] < ghi[a] ? hi[a] : ghi[a];
        empty |= lo[a] >= hi[a];
        pl.lo[a] = lo[a] - p.so[a];  // (relative to the grid's origin: lo >= cs_lo >= so)
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
    if (zhi <= p.zs0 || zlo >= p.zs1 || xlo >= p.S || ylo >= p.S) return true;
    // ... or the pass' box in x / y (the grid's box, Params::cs_lo: the mesh's bounding box - which nothing misses - or an x / y
    // tile of a grid wider than 65 535 samples)
    const uint32_t xhi = floor_u32(mx.x) + 1u, yhi = floor_u32(mx.y) + 1u;
    return xhi <= p.cs_lo[0] || xlo >= p.cs_hi[0] || yhi <= p.cs_lo[1] || ylo >= p.cs_hi[1];
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

// (for Counters::n_candidates_sq: how unequal the leaves are - a leaf of more than 65 535 candidates counts as one of 65 535)
__device__ __forceinline__ unsigned long long leaf_size_squared(uint64_t count)
{
    const unsigned long long n = count < 65535ull ? count : 65535ull;
    return n * n;
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
        // leaves and tiles are reserved with ONE 64-bit atomic on the pair (n_leaves, n_tiles): atomics on one cache line
        // serialise (~88 per us), and with a thousand workgroups reserving at once they were most of k_expand_roots' time
        static_assert(offsetof(Counters, n_leaves) % 8 == 0 && offsetof(Counters, n_tiles) == offsetof(Counters, n_leaves) + 4, "pair");
        s_base[0] = s_base[1] = 0u;
        if (tot_leaf | tot_tile) {
            const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long *>(&c->n_leaves),
                                                     (unsigned long long) tot_leaf | ((unsigned long long) tot_tile << 32));
            s_base[0] = (uint32_t) old;
            s_base[1] = (uint32_t) (old >> 32);
            // (the pair is one 64-bit word: a leaf count that carried out of its 32 bits would corrupt the tile count - the
            // counters keep counting past their capacities - so the wrap is reported instead of passing for a small count)
            if ((uint32_t) old + tot_leaf < (uint32_t) old || (uint32_t) (old >> 32) + tot_tile < (uint32_t) (old >> 32)) atomicOr(&c->err_flags, kErrCounterWrap);
        }
        if (tot_leaf && node_round == 0) atomicAdd(&c->n_root_leaves, tot_leaf);
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

// The blocks of 256 triangles whose z extent (from the slab plan, k_zhist) meets this GPU's slab, compacted into a list:
// the reference's sortTriangleIntoChunks (obj2voxel.cpp:226-243) at the granularity of a block and with one "chunk" per GPU.
// On an N-GPU run every rank holds the whole triangle list but only ~1/N of its blocks matter to it; k_expand_roots then
// walks the list instead of testing every block.  count = ~0: the extents were made with another transform, no list.
__global__ __launch_bounds__(kBlock) void k_list_blocks(const float2 *__restrict__ zrange, const float *__restrict__ zrange_xform,
                                                        const Counters *c, uint32_t *list, uint32_t *count, Params p)
{
    bool valid = true;
    for (int i = 0; i < 12; ++i) valid &= __float_as_uint(zrange_xform[i]) == __float_as_uint(c->xform[i]);
    if (!valid) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *count = 0xffffffffu;
        return;
    }
    const uint32_t n_blocks = (uint32_t) ((p.n_tris + kBlock - 1) / kBlock);
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t b0 = blockIdx.x * kBlock; b0 < n_blocks; b0 += gridDim.x * kBlock) {  // (uniform per wavefront)
        const uint32_t b = b0 + threadIdx.x;
        bool keep = false;
        if (b < n_blocks) {
            const float2 r = zrange[b];
            // (a block whose extent is not finite is kept: its triangles decide for themselves)
            keep = !(r.y < 1e9f && (floor_u32(r.y) + 1u <= p.zs0 || floor_u32(r.x) >= p.zs1));
        }
        const unsigned long long m = __ballot(keep);
        if (m) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(count, (uint32_t) __popcll(m));
            base = __shfl(base, 0, 64);
            if (keep) list[base + __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u))] = b;
        }
    }
}

// Roots: one lane per input triangle.  applyMeshTransform (obj2voxel.cpp:202-224) then the head of
// voxelizeTriangleToUvBuffer (voxelization.cpp:488-511).  With a block list (k_list_blocks) only the listed blocks of 256
// triangles are visited.
__global__ __launch_bounds__(kBlock) void k_expand_roots(const float *__restrict__ verts, const float *__restrict__ uvs,
                                                         Counters *c, Leaf *leaves, Tile *tiles, BigLeaf *big,
                                                         Node *nodes_out, const float2 *__restrict__ zrange,
                                                         const float *__restrict__ zrange_xform,
                                                         const uint32_t *__restrict__ block_list, const uint32_t *block_count, Params p)
{
    __shared__ __align__(8) uint32_t s_wave[2 * (kBlock / 64)];
    __shared__ uint32_t s_base[4];
    __shared__ float s_v[kBlock * 9];
    __shared__ float s_t[kBlock * 6];
    __shared__ unsigned long long s_cand, s_cand_sq;
    __shared__ unsigned long long s_bypass[3];  // Params::root_bypass: this workgroup's bypassed triangles, their candidates, the squares
    unsigned long long my_bypass = 0, my_bypass_cand = 0, my_bypass_sq = 0;
    if (threadIdx.x < 3) s_bypass[threadIdx.x] = 0;

    Affine xf;
    xf.m[0] = {c->xform[0], c->xform[1], c->xform[2]};
    xf.m[1] = {c->xform[3], c->xform[4], c->xform[5]};
    xf.m[2] = {c->xform[6], c->xform[7], c->xform[8]};
    xf.t = {c->xform[9], c->xform[10], c->xform[11]};

    // z extents per block of 256 triangles from the slab plan (k_zhist), valid if they were made with this transform
    bool use_zrange = zrange != nullptr;
    if (use_zrange)
        for (int i = 0; i < 12; ++i) use_zrange &= __float_as_uint(zrange_xform[i]) == __float_as_uint(c->xform[i]);

    // One reservation (a handful of global atomics on the same few addresses, which serialise at ~88 per us) per
    // kRootBatch sub-batches of 256 triangles: the sub-batches are classified twice - first only to count what they
    // emit, then, with the slots known, to write it.
#ifndef O2V_ROOT_BATCH
#define O2V_ROOT_BATCH 3
#endif
    constexpr uint32_t kRootBatch = O2V_ROOT_BATCH;
    // the sequence of blocks this launch visits: the listed ones (their extents were already tested), or all
    const bool listed = block_list != nullptr && *block_count != 0xffffffffu;
    if (listed) use_zrange = false;
    const uint64_t n_seq = listed ? (uint64_t) *block_count : (p.n_tris + kBlock - 1) / kBlock;
    const uint64_t n_super = (n_seq + kRootBatch - 1) / kRootBatch;
    auto block_at = [&](uint64_t i) -> uint64_t { return i >= n_seq ? ~0ull : (listed ? (uint64_t) block_list[i] : i); };
    const uint64_t n_blocks = (p.n_tris + kBlock - 1) / kBlock;

    // Vertex (and uv) staging is software-pipelined: while one sub-batch is classified from LDS, the loads of the next
    // one in the sequence are already in flight (15 registers per lane).
    float pre_v[9], pre_t[6];
    uint64_t pre_blk = ~0ull;  // the sub-batch whose data is in pre_v / pre_t (block-uniform)
    auto skipped = [&](uint64_t blk) -> bool {
        if (blk >= n_blocks) return true;
        if (use_zrange) {
            // every triangle of the block fails misses_slab()'s z test (floor_u32 is monotonic), so none is read
            const float2 r = zrange[blk];
            if (r.y < 1e9f && (floor_u32(r.y) + 1u <= p.zs0 || floor_u32(r.x) >= p.zs1)) return true;
        }
        return false;
    };
    auto prefetch = [&](uint64_t blk) {
        pre_blk = ~0ull;
        if (skipped(blk)) return;
        const uint64_t base = blk * kBlock;
        const uint32_t n_here = (uint32_t) (p.n_tris - base < kBlock ? p.n_tris - base : kBlock);
#pragma unroll
        for (uint32_t k = 0; k < 9; ++k) {
            const uint32_t idx = threadIdx.x + k * kBlock;
            pre_v[k] = idx < n_here * 9u ? verts[base * 9 + idx] : 0.f;
        }
        if (p.has_uv) {
#pragma unroll
            for (uint32_t k = 0; k < 6; ++k) {
                const uint32_t idx = threadIdx.x + k * kBlock;
                pre_t[k] = idx < n_here * 6u ? uvs[base * 6 + idx] : 0.f;
            }
        }
        pre_blk = blk;
    };

    // stages sub-batch `blk` (prefetching `next`) and classifies this lane's triangle of it; false if the whole
    // sub-batch is skipped
    auto classify = [&](uint64_t blk, uint64_t next, Sub &s, Emit &e, LeafPlan &pl, float &area, bool &as_leaf, bool &as_node, bool &bypassed) -> bool {
        e = Emit{0, 0, 0, 0};
        as_leaf = as_node = bypassed = false;
        if (skipped(blk)) return false;
        if (pre_blk != blk) prefetch(blk);
        const uint64_t base = blk * kBlock;
        const uint32_t n_here = (uint32_t) (p.n_tris - base < kBlock ? p.n_tris - base : kBlock);
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < 9; ++k) s_v[threadIdx.x + k * kBlock] = pre_v[k];
        if (p.has_uv) {
#pragma unroll
            for (uint32_t k = 0; k < 6; ++k) s_t[threadIdx.x + k * kBlock] = pre_t[k];
        }
        prefetch(next);
        __syncthreads();
        if (threadIdx.x >= n_here) return true;
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
        if (misses_slab(s, p)) return true;
        area = tri_area(s.v0, s.v1, s.v2);
        if (roughly_axis_aligned(s.v0, s.v1, s.v2) || voxel_volume(s) < kSubdivisionVolumeLimit) {
            pl = plan_leaf(s, p);
            if (pl.count >> 32) {
                atomicOr(&c->err_flags, kErrLeafTooLarge);
            }
            else if (p.root_bypass && pl.ntiles == 1u) {
                bypassed = true;  // k_voxelize_occ makes this leaf itself, by the same rules (root_leaf_of_one_tile)
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
        return true;
    };

    for (uint64_t sblk = blockIdx.x; sblk < n_super; sblk += gridDim.x) {
        Sub s{};
        Emit e{0, 0, 0, 0}, sum{0, 0, 0, 0};
        LeafPlan pl{};
        float area = 0;
        bool as_leaf = false, as_node = false, bypassed = false;
        // pass 1: what this lane's (up to) four triangles emit
        for (uint32_t k = 0; k < kRootBatch; ++k) {
            const uint64_t at = sblk * kRootBatch + k, blk = block_at(at);
            // after the last sub-batch of this pass comes the first one again (pass 2) - or, where most triangles are left to
            // k_voxelize_occ (root_bypass) and pass 2 has nothing to write as a rule, the workgroup's next super-block
            const uint64_t after = p.root_bypass ? (sblk + gridDim.x) * kRootBatch : sblk * kRootBatch;
            classify(blk, block_at(k + 1 < kRootBatch ? at + 1 : after), s, e, pl, area, as_leaf, as_node, bypassed);
            if (bypassed) {
                my_bypass += 1;
                my_bypass_cand += pl.count;
                my_bypass_sq += leaf_size_squared(pl.count);
            }
            sum.n_leaf += e.n_leaf;
            sum.n_tile += e.n_tile;
            sum.n_big += e.n_big;
            sum.n_node += e.n_node;
        }
        // a super-block whose triangles all miss this GPU's slab has nothing to reserve (the common case on the other
        // ranks of a multi-GPU run, where every rank filters the whole triangle list)
        if (!__syncthreads_or((int) (sum.n_leaf | sum.n_node))) continue;
        BlockSlots slot = reserve_slots(sum, c, 0, s_wave, s_base);
        if (threadIdx.x == 0) s_cand = s_cand_sq = 0;
        // pass 2: the same triangles again, now written to their slots
        for (uint32_t k = 0; k < kRootBatch; ++k) {
            const uint64_t at = sblk * kRootBatch + k, blk = block_at(at);
            // ... and after the last one of pass 2 the first sub-batch of this workgroup's next super-block
            const uint64_t next = block_at(k + 1 < kRootBatch ? at + 1 : (sblk + gridDim.x) * kRootBatch);
            if (!classify(blk, next, s, e, pl, area, as_leaf, as_node, bypassed)) continue;
            const uint32_t tri = (uint32_t) (blk * kBlock + threadIdx.x);
            if (as_leaf) {
                if (slot.leaf < p.cap_leaves) write_leaf(leaves, slot.leaf, s, tri, 0u, area, pl);
                write_tiles(tiles, big, slot.leaf, slot.tile, pl.ntiles, slot.big, p);
                atomicAdd(&s_cand, pl.count);
                atomicAdd(&s_cand_sq, leaf_size_squared(pl.count));
                slot.leaf += 1;
                slot.tile += pl.ntiles;
                slot.big += e.n_big;
            }
            if (as_node) {
                if (slot.node < p.cap_nodes) {
                    Node n;
                    n.v[0] = s.v0.x; n.v[1] = s.v0.y; n.v[2] = s.v0.z;
                    n.v[3] = s.v1.x; n.v[4] = s.v1.y; n.v[5] = s.v1.z;
                    n.v[6] = s.v2.x; n.v[7] = s.v2.y; n.v[8] = s.v2.z;
                    n.t[0] = s.t0.x; n.t[1] = s.t0.y; n.t[2] = s.t1.x; n.t[3] = s.t1.y; n.t[4] = s.t2.x; n.t[5] = s.t2.y;
                    n.tri = tri;
                    n.pathkey = 0;
                    n.depth = 0;
                    n.area = area;
                    n.pad = 0;
                    nodes_out[slot.node] = n;
                }
                slot.node += 1;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0 && s_cand) {
            atomicAdd(&c->n_candidates, s_cand);
            atomicAdd(&c->n_candidates_sq, s_cand_sq);
        }
    }
    if (p.root_bypass) {
        // (one pair of global atomics per workgroup, not per super-block: they serialise at ~88 per us)
        __syncthreads();
        if (my_bypass) {
            atomicAdd(&s_bypass[0], my_bypass);
            atomicAdd(&s_bypass[1], my_bypass_cand);
            atomicAdd(&s_bypass[2], my_bypass_sq);
        }
        __syncthreads();
        if (threadIdx.x == 0 && s_bypass[0]) {
            atomicAdd(&c->n_bypass, s_bypass[0]);
            atomicAdd(&c->n_candidates, s_bypass[1]);
            atomicAdd(&c->n_candidates_sq, s_bypass[2]);
        }
    }
}

// Params::root_bypass: the root triangles k_expand_roots leaves to k_voxelize_occ - those that are voxelized as they are
// (voxelization.cpp:488-511: roughly axis-aligned, or a voxel AABB of fewer than 512 cells), touch the slab and fit one tile.
// The same sequence of tests as k_expand_roots' classify(), on the same transformed vertices: both kernels decide alike.
// `other` (Params::solo_roots, where k_expand_roots is not launched at all): the triangle is none of k_voxelize_occ's and not
// nothing either - a node, a leaf of several tiles, a leaf too large: k_expand_roots' business.
__device__ __forceinline__ bool root_leaf_of_one_tile(const Sub &s, const Params &p, LeafPlan &pl, bool &other)
{
    other = false;
    if (misses_slab(s, p)) return false;
    if (!(roughly_axis_aligned(s.v0, s.v1, s.v2) || voxel_volume(s) < kSubdivisionVolumeLimit)) {
        other = true;
        return false;
    }
    pl = plan_leaf(s, p);
    other = (pl.count >> 32) != 0u || pl.ntiles > 1u;
    return !(pl.count >> 32) && pl.ntiles == 1u;
}
__device__ __forceinline__ bool root_leaf_of_one_tile(const Sub &s, const Params &p, LeafPlan &pl)
{
    bool other;
    return root_leaf_of_one_tile(s, p, pl, other);
}

// Params::root_bypass on a tessellated surface (practically every triangle is one leaf of one tile: o2v_hip_voxelize decides from the
// histogram of the triangles' extents): k_expand_roots would only read and count, through its LDS staging and barriers.  This kernel
// does that part alone - a wavefront per block of 256 triangles, its 36 loads per lane in flight together, no barrier - and lists the
// blocks that hold a triangle k_expand_roots has to handle (a node, a leaf of several tiles, a leaf too large): k_expand_roots then
// walks that list (as a rule empty) and counts the bypassed triangles of those blocks itself.  `seq_list` / `seq_count`: the blocks
// that meet the slab (k_list_blocks), or null / ~0 for all.
constexpr uint32_t kCountRootsWgsPerCu = 4;
__global__ __launch_bounds__(kBlock) void k_count_roots(const float *__restrict__ verts, Counters *c, const uint32_t *__restrict__ seq_list,
                                                        const uint32_t *seq_count, uint32_t *need_list, uint32_t *need_count, Params p)
{
    __shared__ unsigned long long s_sum[3];
    if (threadIdx.x < 3) s_sum[threadIdx.x] = 0;
    __syncthreads();
    Affine xf;
    xf.m[0] = {c->xform[0], c->xform[1], c->xform[2]};
    xf.m[1] = {c->xform[3], c->xform[4], c->xform[5]};
    xf.m[2] = {c->xform[6], c->xform[7], c->xform[8]};
    xf.t = {c->xform[9], c->xform[10], c->xform[11]};
    const bool listed = seq_list != nullptr && *seq_count != 0xffffffffu;
    const uint64_t n_seq = listed ? (uint64_t) *seq_count : (p.n_tris + kBlock - 1) / kBlock;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave_id = (uint64_t) blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6), n_waves = (uint64_t) gridDim.x * (kBlock / 64u);
    unsigned long long my_n = 0, my_cand = 0, my_sq = 0;
    for (uint64_t i = wave_id; i < n_seq; i += n_waves) {
        const uint64_t blk = listed ? (uint64_t) seq_list[i] : i;
        float q[kBlock / 64u][9];
#pragma unroll
        for (uint32_t j = 0; j < kBlock / 64u; ++j) {
            const uint64_t tri = blk * kBlock + j * 64u + lane;
#pragma unroll
            for (uint32_t k = 0; k < 9; ++k) q[j][k] = tri < p.n_tris ? verts[tri * 9u + k] : 0.f;
        }
        bool need = false;
        unsigned long long bn = 0, bc = 0, bs = 0;
#pragma unroll
        for (uint32_t j = 0; j < kBlock / 64u; ++j) {
            const uint64_t tri = blk * kBlock + j * 64u + lane;
            if (tri >= p.n_tris) continue;
            Sub s{};
            s.v0 = affine_apply(xf, V3{q[j][0], q[j][1], q[j][2]});
            s.v1 = affine_apply(xf, V3{q[j][3], q[j][4], q[j][5]});
            s.v2 = affine_apply(xf, V3{q[j][6], q[j][7], q[j][8]});
            // (k_expand_roots' classify(), same order)
            if (misses_slab(s, p)) continue;
            if (roughly_axis_aligned(s.v0, s.v1, s.v2) || voxel_volume(s) < kSubdivisionVolumeLimit) {
                const LeafPlan pl = plan_leaf(s, p);
                if (pl.count >> 32) need = true;
                else if (pl.ntiles == 1u) {
                    bn += 1;
                    bc += pl.count;
                    bs += leaf_size_squared(pl.count);
                }
                else if (pl.ntiles) need = true;
            }
            else need = true;
        }
        if (__ballot(need) != 0ull) {
            if (lane == 0) need_list[atomicAdd(need_count, 1u)] = (uint32_t) blk;
        }
        else {
            my_n += bn;
            my_cand += bc;
            my_sq += bs;
        }
    }
    if (my_n) {
        atomicAdd(&s_sum[0], my_n);
        atomicAdd(&s_sum[1], my_cand);
        atomicAdd(&s_sum[2], my_sq);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_sum[0]) {
        atomicAdd(&c->n_bypass, s_sum[0]);
        atomicAdd(&c->n_candidates, s_sum[1]);
        atomicAdd(&c->n_candidates_sq, s_sum[2]);
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
    __shared__ unsigned long long s_cand, s_cand_sq;
    const uint32_t n_in = c->n_nodes[round] < p.cap_nodes ? c->n_nodes[round] : p.cap_nodes;
    const uint32_t n_blocks = (n_in + kBlock - 1) / kBlock;
    for (uint32_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const uint32_t i = blk * kBlock + threadIdx.x;
        const bool live = i < n_in;
        __syncthreads();
        if (threadIdx.x == 0) s_cand = s_cand_sq = 0;
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
            unsigned long long cand = 0, cand_sq = 0;
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
                    cand_sq += leaf_size_squared(pl[k].count);
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
            if (cand) {
                atomicAdd(&s_cand, cand);
                atomicAdd(&s_cand_sq, cand_sq);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0 && s_cand) {
            atomicAdd(&c->n_candidates, s_cand);
            atomicAdd(&c->n_candidates_sq, s_cand_sq);
        }
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

// Lists, before k_voxelize, the bricks of the output grid that can receive pooled hits: every brick the clamped box of a leaf
// touches gets its flag set (the dirty-flag map; k_scan_flags then turns the flags into the brick list and gives every listed
// brick its hit slab).  A superset of the bricks that do receive hits - a hit's voxel lies in its leaf's clamped box
// (voxelization.cpp:440-444) - by about a third on tessellated surfaces.  One lane per leaf; a leaf whose box covers more than
// 64 bricks (large axis-aligned triangles) is marked by its whole wavefront.  No work if the pass pools no hits (every
// triangle whole and on the direct MAX path).
__global__ __launch_bounds__(kBlock) void k_mark_bricks(const Leaf *__restrict__ leaves, const Counters *c, uint8_t *brick_flags, uint32_t force_general, Params p)
{
    if (expand_overflowed(c, p)) return;
    if (pools_no_hits(c, p, force_general)) return;
    const uint32_t n_leaves = c->n_leaves < p.cap_leaves ? c->n_leaves : p.cap_leaves;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i0 = blockIdx.x * kBlock; i0 < n_leaves; i0 += gridDim.x * kBlock) {  // (uniform per wavefront)
        const uint32_t i = i0 + threadIdx.x;
        uint32_t b0[3] = {0, 0, 0}, nb[3] = {0, 0, 0};
        if (i < n_leaves) {
            const Leaf &l = leaves[i];
            const uint32_t lo[3] = {l.bmin_xy & 0xffffu, l.bmin_xy >> 16, l.bmin_z_dx & 0xffffu};
            const uint32_t d[3] = {l.bmin_z_dx >> 16, l.dy_dz & 0xffffu, l.dy_dz >> 16};
            const uint32_t sh[3] = {kBrickXs, kBrickYs, kBrickZs};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if (d[a] == 0u) continue;
                // output cells [lo >> ss, (lo + d - 1) >> ss], relative to the grid's origin (as the leaf's box is, Params::so)
                const uint32_t c0 = lo[a] >> p.ss_shift, c1 = (lo[a] + d[a] - 1u) >> p.ss_shift;
                b0[a] = c0 >> sh[a];
                nb[a] = (c1 >> sh[a]) - b0[a] + 1u;
            }
            if (!d[0] || !d[1] || !d[2]) nb[0] = nb[1] = nb[2] = 0u;
        }
        const uint32_t count = nb[0] * nb[1] * nb[2];  // (a box of < 2^16 cells per axis: < 2^14 bricks per axis; leaves that large are rare and aligned, one axis is thin)
        const bool big = count > 64u || nb[0] > 0xffffu;
        if (count && !big) {
            for (uint32_t z = 0; z < nb[2]; ++z)
                for (uint32_t y = 0; y < nb[1]; ++y) {
                    const uint32_t row = ((b0[2] + z) * p.NBy + (b0[1] + y)) * p.NBx + b0[0];
                    for (uint32_t x = 0; x < nb[0]; ++x) brick_flags[row + x] = 1;
                }
        }
        unsigned long long mb = __ballot(big);
        while (mb) {
            const int src = __ffsll((long long) mb) - 1;
            mb &= mb - 1ull;
            const uint32_t x0 = __shfl(b0[0], src, 64), y0 = __shfl(b0[1], src, 64), z0 = __shfl(b0[2], src, 64);
            const uint32_t nx = __shfl(nb[0], src, 64), ny = __shfl(nb[1], src, 64), nz = __shfl(nb[2], src, 64);
            const uint64_t total = (uint64_t) nx * ny * nz;
            for (uint64_t t = lane; t < total; t += 64u) {
                const uint32_t x = (uint32_t) (t % nx), yz = (uint32_t) (t / nx), y = yz % ny, z = yz / ny;
                brick_flags[((size_t) (z0 + z) * p.NBy + (y0 + y)) * p.NBx + (x0 + x)] = 1;
            }
        }
    }
}
