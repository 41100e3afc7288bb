// o2v_dev_k5_scan_scatter.hpp -- K5: dirty-brick scan and the counting sort of the pooled hits BY BRICK
// (k_scan_flags, k_scan_bcount, k_scatter, k_reset_bcount).
//
// Part of the device code of o2v_device.hip, which includes this file inside its anonymous namespace (one
// translation unit: the stages share records and launch parameters).  Not a stand-alone header.
//
// The unit of the sort is the brick (4 x 4 x 4 output cells), not the cell: k_voxelize ranks a pooled hit among the hits of
// its brick with one atomic per (wavefront flush, brick), so the hits a wavefront emits for a brick keep consecutive ranks
// and the scatter writes them as one run.  The order inside a brick - by cell, then the reference's (sub-voxel, triangle,
// leaf) order - is made by the resolve kernels, which read a brick's hits as one contiguous block.  (Round 2 sorted by cell:
// one atomic on a 4-byte counter per hit, a 4-byte random read and a 16/24-byte random write per hit in the scatter, and
// gathers of short per-cell runs in the replay: the three latency-bound kernels of that design.)

// ---- K5a: dirty bricks ---------------------------------------------------------------------------------------
// k_scan_flags streams the one-byte-per-brick dirty map (n_bricks bytes, 16 MiB at 1024^3), lists the dirty bricks and
// clears their flags.

constexpr uint32_t kFlagLoads = 2;  // 16-byte loads of the flag map per thread and round
__global__ __launch_bounds__(kBlock) void k_scan_flags(uint8_t *brick_dirty, uint32_t *n_dirty, uint32_t *dirty_list, Counters *c, Params p)
{
    // The workgroup collects dirty bricks in LDS over several rounds and reserves their place in the list with one global
    // atomic per few thousand of them (atomics on one address serialise at ~88 per us: one per round and workgroup was most
    // of this kernel's time).  Launched with two workgroups per CU.
    constexpr uint32_t kRoundMax = kBlock * 16 * kFlagLoads;  // bricks one round can add
    constexpr uint32_t kFlushAbove = 4096;
    __shared__ uint32_t s_list[kFlushAbove + kRoundMax];
    __shared__ uint32_t s_n, s_base;
    const uint32_t n_groups = (p.n_bricks + 15u) / 16u;  // the flag map is padded to a multiple of 16 bytes
    uint4 *f4 = reinterpret_cast<uint4 *>(brick_dirty);
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    auto flush = [&](uint32_t n) {
        if (threadIdx.x == 0) s_base = atomicAdd(n_dirty, n);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += kBlock)
            if (s_base + i < p.cap_dirty) dirty_list[s_base + i] = s_list[i];
        if (threadIdx.x == 0 && s_base + n > p.cap_dirty) atomicOr(&c->err_flags, kErrDirtyList);
        __syncthreads();
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
    };
    for (uint32_t g0 = blockIdx.x * kBlock * kFlagLoads; g0 < n_groups; g0 += gridDim.x * kBlock * kFlagLoads) {
        uint4 f[kFlagLoads];
#pragma unroll
        for (uint32_t u = 0; u < kFlagLoads; ++u) {
            const uint32_t g = g0 + u * kBlock + threadIdx.x;
            f[u] = g < n_groups ? f4[g] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (uint32_t u = 0; u < kFlagLoads; ++u) {
            if (f[u].x | f[u].y | f[u].z | f[u].w) {
                const uint32_t g = g0 + u * kBlock + threadIdx.x;
                const uint32_t w[4] = {f[u].x, f[u].y, f[u].z, f[u].w};
#pragma unroll
                for (uint32_t k = 0; k < 16; ++k)
                    if ((w[k >> 2] >> ((k & 3u) * 8u)) & 0xffu) s_list[atomicAdd(&s_n, 1u)] = g * 16u + k;
                f4[g] = make_uint4(0, 0, 0, 0);
            }
        }
        __syncthreads();
        const uint32_t n = s_n;
        __syncthreads();  // (every thread has read n before anyone adds to s_n again: the decision below must be uniform)
        if (n > kFlushAbove) flush(n);  // (the next round may add kRoundMax more)
    }
    const uint32_t n = s_n;
    if (n) flush(n);
}

// ---- K5b: brick counts -> offsets --------------------------------------------------------------------------------
// Resolve tiers by the number of pooled hits of a brick (a brick's records are sorted and folded by one wavefront or
// workgroup): up to kTierWave / kTierWave2 / kTierMid by a single wavefront out of its registers (one, two, four records per
// lane), up to kTierLong / kTierLong2 by a workgroup in LDS (every cell of the brick folded by its own lane: the long chains
// of a sphere's pole cells run side by side), up to kTierBig with the sort keys in 96 KiB of dynamic LDS and a sequential
// replay, beyond that in global memory.
constexpr uint32_t kTierWave = 64, kTierWave2 = 128, kTierMid = 256, kTierLong = 1024, kTierLong2 = 4096, kTierBig = 8192;
constexpr uint32_t kResolveClasses = 7;

struct __attribute__((aligned(16))) BrickOcc {  // 32 B: one brick with pooled hits
    uint32_t brick;
    uint32_t offset;    // first SortedRec of the brick
    uint32_t count;     // number of hits
    uint32_t out_base;  // first output record of the brick: its occupied cells follow in ascending order of the cell
    unsigned long long cells;  // one bit per occupied cell
    unsigned long long pad;
};

struct ResolveLists {  // the bricks with pooled hits, by hit count class (each list: cap entries)
    BrickOcc *tier[kResolveClasses];
    uint32_t cap;
};

__device__ __forceinline__ uint32_t resolve_class(uint32_t cnt)
{
    return cnt <= kTierWave ? 0u : cnt <= kTierWave2 ? 1u : cnt <= kTierMid ? 2u : cnt <= kTierLong ? 3u : cnt <= kTierLong2 ? 4u : cnt <= kTierBig ? 5u : 6u;
}
__device__ __forceinline__ uint32_t *class_counter(Counters *c, uint32_t k)
{
    return k == 0 ? &c->n_w64 : k == 1 ? &c->n_w128 : k == 2 ? &c->n_mid : k == 3 ? &c->n_long : k == 4 ? &c->n_long2 : k == 5 ? &c->n_bigl : &c->n_huge;
}

// One thread per dirty brick: its hit count becomes its offset in the sorted record array and the number of its occupied cells
// its place in the output (block-level prefix sums, one reservation of (bricks, hits, cells) per workgroup round: no kernel
// behind this one needs an atomic to place a record); the offset is written back into the counter, where k_scatter reads
// it, the cell mask is taken over into the list entry and cleared, and the brick is filed into the list of its resolve tier.
__global__ __launch_bounds__(kBlock) void k_scan_bcount(uint32_t *bcount, unsigned long long *bmask, const uint32_t *__restrict__ dirty_list,
                                                        Counters *c, ResolveLists lists, Params p)
{
    __shared__ uint32_t s_wave[kBlock / 64], s_base[3], s_cls[kResolveClasses], s_cls_base[kResolveClasses];
    const uint32_t n_dirty = c->n_dirty < p.cap_dirty ? c->n_dirty : p.cap_dirty;
    if (threadIdx.x < kResolveClasses) s_cls[threadIdx.x] = 0;
    for (uint32_t base = blockIdx.x * kBlock; base < n_dirty; base += gridDim.x * kBlock) {  // (uniform per workgroup)
        __syncthreads();
        const uint32_t item = base + threadIdx.x;
        const bool valid = item < n_dirty;
        const uint32_t brick = valid ? dirty_list[item] : 0u;
        const uint32_t cnt = valid ? bcount[brick] : 0u;
        const unsigned long long cells = valid ? bmask[brick] : 0ull;
        if (valid) bmask[brick] = 0ull;
        uint32_t total, total_cells;
        uint32_t run = block_exscan(cnt, s_wave, total);
        __syncthreads();
        uint32_t run_cells = block_exscan((uint32_t) __popcll(cells), s_wave, total_cells);
        const uint32_t n_here = n_dirty - base < kBlock ? n_dirty - base : kBlock;
        uint32_t cls = 0, cls_rank = 0;
        if (valid && cnt) {
            cls = resolve_class(cnt);
            cls_rank = atomicAdd(&s_cls[cls], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(&c->n_bocc, n_here);  // (an upper bound of every tier's list: the host sizes the lists by it)
            s_base[1] = atomicAdd(&c->n_sorted, total);
            s_base[2] = atomicAdd(&c->n_vox, total_cells);
        }
        if (threadIdx.x < kResolveClasses) {
            const uint32_t n_cls = s_cls[threadIdx.x];
            s_cls[threadIdx.x] = 0;
            if (n_cls) s_cls_base[threadIdx.x] = atomicAdd(class_counter(c, threadIdx.x), n_cls);
        }
        __syncthreads();
        if (valid) {
            run += s_base[1];
            bcount[brick] = run;
            if (cnt) {
                const uint32_t at = s_cls_base[cls] + cls_rank;
                if (at < lists.cap) lists.tier[cls][at] = BrickOcc{brick, run, cnt, s_base[2] + run_cells, cells, 0ull};
            }
        }
    }
}

// ---- K5c: scatter --------------------------------------------------------------------------------------------
// Streams the hit pool once (coalesced 32-byte records, holes skipped) and places every hit at offset(brick) + rank, so
// that each brick's hits are contiguous for the resolve kernels.  The hits one wavefront of k_voxelize emitted for a brick
// have consecutive ranks and sit in one 256-slot chunk of the pool, so a wavefront here writes a few runs, not 64 pieces.
// The record keeps its cell inside the brick in its last word.
__global__ __launch_bounds__(kBlock) void k_scatter(const HitRec *__restrict__ pool, const uint32_t *__restrict__ bcount,
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
        if (r.brick == kHoleBrick || r.pad == kPickRecord) continue;
        const uint32_t pos = bcount[r.brick] + (r.local_rank & (kMaxRank - 1u));
        if (pos < p.cap_hits) {
            const uint32_t local = r.local_rank >> 24;
            if (stride == 4u) reinterpret_cast<uint4 *>(sorted)[pos] = make_uint4(r.keyhi, r.keylo, __float_as_uint(r.w), local);
            else reinterpret_cast<SortedRec *>(sorted)[pos] = SortedRec{r.keyhi, r.keylo, r.w, r.u, r.v, local};
        }
    }
}

// Zeroes the counters of the dirty bricks: leaves them clean for the next voxelization.  Runs after k_scatter has read
// the offsets.
__global__ __launch_bounds__(kBlock) void k_reset_bcount(uint32_t *bcount, const uint32_t *__restrict__ dirty_list, const Counters *c, Params p)
{
    const uint32_t n_dirty = c->n_dirty < p.cap_dirty ? c->n_dirty : p.cap_dirty;
    for (uint32_t item = blockIdx.x * kBlock + threadIdx.x; item < n_dirty; item += gridDim.x * kBlock) bcount[dirty_list[item]] = 0u;
}
