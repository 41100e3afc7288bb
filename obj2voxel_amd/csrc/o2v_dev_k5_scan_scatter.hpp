// o2v_dev_k5_scan_scatter.hpp -- K5: dirty-brick scan, counting sort of the hits (k_scan_flags / _bricks, k_scatter, k_reset_bricks).
//
// Part of the device code of o2v_device.hip, which includes this file inside its anonymous namespace (one
// translation unit: the stages share records and launch parameters).  Not a stand-alone header.

// ---- K5a: scan + compact + reset ---------------------------------------------------------------------------
// Two steps.  k_scan_flags streams the one-byte-per-brick dirty map (n_bricks bytes, 16 MiB at 1024^3), lists the
// dirty bricks and clears their flags.  k_scan_bricks then reads only those bricks (four cells = one 16-byte load
// per lane), compacts the occupied cells into `occ` through an LDS staging buffer (one global atomic per flush,
// not per cell) and writes zeros back, so the grid and the flag map are clean for the next voxelization.

// Every wavefront walks its own share of the map with kFlagLoads 16-byte loads per lane in flight and stages the dirty bricks it
// finds in its own LDS list (positions from a prefix sum over the lanes: no LDS atomics, and no workgroup barrier inside the loop -
// the rounds of the barrier version, four on the bench headline, each waited for the slowest wavefront's loads).  A wavefront's
// list is written out when it may not take another load's worth; what is left at the end goes out per workgroup, with ONE atomic
// on the list's counter (atomics on one address serialise, ~5 ns each: one per wavefront would cost more than the scan).
#ifndef O2V_FLAG_LOADS
#define O2V_FLAG_LOADS 4
#endif
#ifndef O2V_SCAN_WGS_PER_CU
#define O2V_SCAN_WGS_PER_CU 2
#endif
constexpr uint32_t kFlagLoads = O2V_FLAG_LOADS;     // 16-byte loads of the flag map per lane and round
constexpr uint32_t kScanFlagsWgsPerCu = O2V_SCAN_WGS_PER_CU;
constexpr uint32_t kFlagWaveCap = 2048;            // entries of a wavefront's list (a load adds at most 64 x 16)
constexpr uint32_t kFlagGroupsPerBlockRound = kBlock * kFlagLoads;
__global__ __launch_bounds__(kBlock) void k_scan_flags(uint8_t *brick_dirty, uint32_t *n_dirty, uint32_t *dirty_list, Counters *c, uint32_t *brick_slab,
                                                       uint32_t force_general, Params p)
{
    if (brick_slab && pools_no_hits(c, p, force_general)) return;
    __shared__ uint32_t s_list[kBlock / 64][kFlagWaveCap];
    __shared__ uint32_t s_count[kBlock / 64], s_base;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t *list = s_list[wave];
    uint32_t n = 0;  // (wavefront-uniform) entries staged
    const uint32_t n_groups = (p.n_bricks + 15u) / 16u;  // the flag map is padded to a multiple of 16 bytes
    uint4 *f4 = reinterpret_cast<uint4 *>(brick_dirty);
    auto write_out = [&](uint32_t base, uint32_t count) {
        for (uint32_t i = lane; i < count; i += 64u)
            if (base + i < p.cap_dirty) {
                dirty_list[base + i] = list[i];
                if (brick_slab) brick_slab[list[i]] = base + i;
            }
        if (lane == 0 && base + count > p.cap_dirty) atomicOr(&c->err_flags, kErrDirtyList);
    };
    const uint32_t wave_id = blockIdx.x * (kBlock / 64u) + wave, n_waves = gridDim.x * (kBlock / 64u);
    for (uint64_t g0 = (uint64_t) wave_id * 64u * kFlagLoads; g0 < n_groups; g0 += (uint64_t) n_waves * 64u * kFlagLoads) {
        uint4 f[kFlagLoads];
#pragma unroll
        for (uint32_t u = 0; u < kFlagLoads; ++u) {
            const uint64_t g = g0 + u * 64u + lane;
            f[u] = g < n_groups ? f4[g] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (uint32_t u = 0; u < kFlagLoads; ++u) {
            const uint32_t w[4] = {f[u].x, f[u].y, f[u].z, f[u].w};
            const bool any = (w[0] | w[1] | w[2] | w[3]) != 0u;
            if (__ballot(any) == 0ull) continue;  // (wavefront-uniform)
            if (n + 64u * 16u > kFlagWaveCap) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(n_dirty, n);
                write_out(__shfl(base, 0, 64), n);
                n = 0;
            }
            uint32_t mine = 0;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) mine += (uint32_t) __popc((((w[k] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w[k]) & 0x80808080u);  // non-zero bytes
            const uint32_t incl = wave_inclusive_scan(mine);
            uint32_t pos = n + incl - mine;
            if (any) {
                const uint32_t g = (uint32_t) (g0 + u * 64u + lane);
#pragma unroll
                for (uint32_t k = 0; k < 16; ++k)
                    if ((w[k >> 2] >> ((k & 3u) * 8u)) & 0xffu) list[pos++] = g * 16u + k;
                f4[g] = make_uint4(0, 0, 0, 0);
            }
            n += __shfl(incl, 63, 64);
        }
    }
    // what is left: one reservation per workgroup
    if (lane == 0) s_count[wave] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < kBlock / 64u; ++w) total += s_count[w];
        s_base = total ? atomicAdd(n_dirty, total) : 0u;
    }
    __syncthreads();
    uint32_t base = s_base;
    for (uint32_t w = 0; w < wave; ++w) base += s_count[w];
    if (n) write_out(base, n);
}

constexpr uint32_t kScanBricksPerWave = 4;                                   // independent 1 KiB loads (four bricks each) in flight per wave
constexpr uint32_t kScanBricksPerRound = (kBlock / 64) * kScanBricksPerWave * kBricksPerLoad;  // 4096 cells per block round
constexpr uint32_t kScanFlushAt = 2048;
constexpr uint32_t kScanCap = kScanFlushAt + kScanBricksPerRound * kBrickCells;

constexpr uint32_t kShortList = 8;     // cells with up to this many hits are sorted in registers by k_resolve
constexpr uint32_t kLane16List = 16;   // up to this: one lane per cell, sixteen records in registers (k_resolve_list16)
constexpr uint32_t kLaneList = 32;     // up to this: 32 lanes per cell (k_resolve_wave<32>)
constexpr uint32_t kWaveList = 64;     // up to this: one wavefront per cell (k_resolve_wave<64>)
constexpr uint32_t kMidList = 256;     // up to this: one wavefront per cell, LDS bitonic sort
constexpr uint32_t kLongList = 2048;   // up to this: one workgroup per cell, LDS bitonic sort
constexpr uint32_t kBigList = 8192;    // up to this: one workgroup per cell, keys + indices in 96 KiB of dynamic LDS;
                                       // beyond: global-memory sort

struct ResolveLists {  // cells k_resolve defers, by hit count class (indices into occ[])
    uint32_t *lane16, *lane, *w64, *mid, *lng, *big, *huge;
    uint32_t *lane8;  // inline cells with 5 .. 8 hits
    uint32_t cap;
};

// Cells with more than kShortList hits are resolved by the cooperative tiers; k_scan_bricks files them by hit count
// while it builds occ[] (one global atomic per class and flush), so that every resolve tier can start at once.
constexpr uint32_t kResolveClasses = 8;  // seven for the cells with more than kShortList hits, one for the inline cells with 5 .. 8
constexpr uint32_t kLane8Class = 7, kFourList = 4;
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
    return k == 0 ? l.lane16 : k == 1 ? l.lane : k == 2 ? l.w64 : k == 3 ? l.mid : k == 4 ? l.lng : k == 5 ? l.big : k == 6 ? l.huge : l.lane8;
}
__device__ __forceinline__ uint32_t *class_counter(Counters *c, uint32_t k)
{
    return k == 0 ? &c->n_lane16 : k == 1 ? &c->n_lane : k == 2 ? &c->n_w64 : k == 3 ? &c->n_mid : k == 4 ? &c->n_long
           : k == 5 ? &c->n_bigl : k == 6 ? &c->n_huge : &c->n_lane8;
}

// Writes the staged occupied cells of one workgroup to `occ`, giving every cell the offset of its hits in the sorted
// record array: one reservation of (cells, hits) per flush, offsets by a block-level prefix sum over the counts.
// The offset is also stored in the cell itself, where k_scatter reads it.
// Cells with at most kInlineHits hits whose brick has a slab keep them there (written by k_voxelize): their occ entry names
// the slab (kOccInline) and they take no part in the counting sort.  The other cells get a range of the sorted array as
// for the hits their slab does not hold (ranks kInlineHits and up, placed by k_scatter; a brick beyond the slab budget has
// all its hits pooled and placed); the resolve tiers read a cell's records from both places (CellRecords).
__device__ __forceinline__ void scan_flush(uint32_t n, const uint32_t *s_lo, const uint32_t *s_hi /* cell >> 32 | slab << 5 */, uint32_t *s_cnt,
                                           uint32_t *s_wave, uint32_t *s_base, uint32_t *s_cls /*[7], zero*/,
                                           uint32_t *s_cls_base /*[7]*/, uint32_t *grid, Counters *c, Occ *occ,
                                           const ResolveLists &lists, const Params &p, uint32_t &hits_seen)
{
    // thread t owns the entries [t * per, (t + 1) * per)
    const uint32_t per = (n + kBlock - 1) / kBlock;
    const uint32_t lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint32_t sum = 0;
    // (what a cell needs of the sorted array: nothing if it is inline, its hits beyond the slab's eight if its brick has a slab)
    auto sorted_need = [&](uint32_t i) -> uint32_t {
        const bool has_slab = (s_hi[i] >> 5) < p.cap_slabs;
        return !has_slab ? s_cnt[i] : (s_cnt[i] > kInlineHits ? s_cnt[i] - kInlineHits : 0u);
    };
    for (uint32_t i = lo; i < hi; ++i) {
        sum += sorted_need(i);
        hits_seen += s_cnt[i];  // (this thread's share of Counters::n_listed_hits)
    }
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
        const uint32_t cell_hi = s_hi[i] & 31u, slab = s_hi[i] >> 5;
        const bool has_slab = slab < p.cap_slabs;
        if (cnt <= kInlineHits && has_slab) {
            if (listed) occ[base_vox + i] = Occ{s_lo[i], s_hi[i], slab, cnt | kOccInline};
            // (tier 1 takes the inline cells with at most four hits straight from occ[] in its four-slot form; the others are
            // filed, so that one launch's lanes all run the eight-slot form)
            if (listed && cnt > kFourList) s_cnt[i] = 0x80000000u | (kLane8Class << 24) | atomicAdd(&s_cls[kLane8Class], 1u);
            continue;
        }
        const uint32_t need = sorted_need(i);
        if (listed) occ[base_vox + i] = Occ{s_lo[i], s_hi[i], run, cnt};
        // (k_scatter places a pooled hit at this + its rank; with a slab the pooled hits' ranks start at kInlineHits)
        grid[((uint64_t) cell_hi << 32) | s_lo[i]] = has_slab ? run - kInlineHits : run;
        if (listed && cnt > kShortList) {
            // rank within its class among this flush's cells; the count is not needed again, the slot keeps the tag
            const uint32_t cls = resolve_class(cnt);
            s_cnt[i] = 0x80000000u | (cls << 24) | atomicAdd(&s_cls[cls], 1u);
        }
        run += need;
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
    const uint32_t n_dirty = c->n_dirty < p.cap_dirty ? c->n_dirty : p.cap_dirty;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t hits_seen = 0;
    const uint32_t n_rounds = (n_dirty + kScanBricksPerRound - 1) / kScanBricksPerRound;
    for (uint32_t r = blockIdx.x; r < n_rounds; r += gridDim.x) {
        uint32_t brick[kScanBricksPerWave], slab[kScanBricksPerWave];
        uint4 h[kScanBricksPerWave];
#pragma unroll
        for (uint32_t k = 0; k < kScanBricksPerWave; ++k) {
            // (a load covers kBricksPerLoad bricks: kLanesPerBrick lanes each)
            const uint32_t item = r * kScanBricksPerRound + (wave * kScanBricksPerWave + k) * kBricksPerLoad + lane / kLanesPerBrick;
            brick[k] = item < n_dirty ? dirty_list[item] : 0xffffffffu;
            slab[k] = item;  // (a brick's place in the list is the number of its hit slab)
        }
#pragma unroll
        for (uint32_t k = 0; k < kScanBricksPerWave; ++k)
            h[k] = brick[k] != 0xffffffffu ? reinterpret_cast<const uint4 *>(grid + (uint64_t) brick[k] * kBrickCells)[lane % kLanesPerBrick]
                                           : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (uint32_t k = 0; k < kScanBricksPerWave; ++k) {
            // The occupied cells are staged in (brick, cell) order - the lanes of a load hold consecutive groups of four cells
            // of consecutive bricks - so that the offsets scan_flush hands out follow the cells' order inside a brick:
            // neighbouring cells then own neighbouring ranges of the sorted array, and the hits k_scatter places for one row of
            // voxels fall into the same few lines.  One LDS atomic per wavefront and load.
            const uint32_t hv[4] = {h[k].x, h[k].y, h[k].z, h[k].w};
            const uint32_t mine = (hv[0] ? 1u : 0u) + (hv[1] ? 1u : 0u) + (hv[2] ? 1u : 0u) + (hv[3] ? 1u : 0u);
            const uint32_t inc = wave_inclusive_scan(mine);
            const uint32_t total = __shfl(inc, 63, 64);
            if (total) {
                uint32_t base = 0;
                if (lane == 63) base = atomicAdd(&s_n, total);
                base = __shfl(base, 63, 64);
                uint32_t slot = base + inc - mine;
#pragma unroll
                for (uint32_t e = 0; e < 4; ++e) {
                    if (hv[e]) {
                        const uint64_t cell = (uint64_t) brick[k] * kBrickCells + (lane % kLanesPerBrick) * 4u + e;
                        s_lo[slot] = (uint32_t) cell;
                        s_hi[slot] = (uint32_t) (cell >> 32) | (slab[k] << 5);  // (cell < 2^37, slab < 2^27)
                        s_cnt[slot] = hv[e];
                        ++slot;
                    }
                }
            }
        }
        __syncthreads();
        const uint32_t n = s_n;
        if (n >= kScanFlushAt) {
            scan_flush(n, s_lo, s_hi, s_cnt, s_wave, s_base, s_cls, s_cls_base, grid, c, occ, lists, p, hits_seen);
            __syncthreads();
            if (threadIdx.x == 0) s_n = 0;
        }
        __syncthreads();
    }
    const uint32_t n = s_n;
    if (n) scan_flush(n, s_lo, s_hi, s_cnt, s_wave, s_base, s_cls, s_cls_base, grid, c, occ, lists, p, hits_seen);
    // the hits this workgroup saw in the listed bricks: one global atomic per workgroup (Counters::n_listed_hits)
    __syncthreads();
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    if (hits_seen) atomicAdd(&s_n, hits_seen);
    __syncthreads();
    if (threadIdx.x == 0 && s_n) atomicAdd(&c->n_listed_hits, s_n);
}

#ifndef O2V_SCATTER_UNROLL
#define O2V_SCATTER_UNROLL 2
#endif
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
    // kScatterUnroll records per lane and round, all loads of a step issued before any is used (the pool record, then the
    // cell's offset: two dependent round trips per record, which only overlap across records)
    constexpr uint32_t kScatterUnroll = O2V_SCATTER_UNROLL;
    for (uint32_t i0 = lo + local_block * kBlock * kScatterUnroll + threadIdx.x; i0 < hi; i0 += blocks_per_xcd * kBlock * kScatterUnroll) {
        HitRec r[kScatterUnroll];
        bool live[kScatterUnroll];
        uint32_t off[kScatterUnroll];
#pragma unroll
        for (uint32_t k = 0; k < kScatterUnroll; ++k) {
            const uint32_t i = i0 + k * kBlock;
            live[k] = i < hi;
            r[k] = pool[live[k] ? i : lo];
        }
#pragma unroll
        for (uint32_t k = 0; k < kScatterUnroll; ++k) {
            live[k] = live[k] && r[k].brick != kHoleBrick && r[k].pad != kPickRecord;
            const uint64_t cell = (uint64_t) r[k].brick * kBrickCells + (r[k].local_rank >> 24);
            off[k] = live[k] ? grid[cell] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < kScatterUnroll; ++k) {
            const uint32_t pos = off[k] + (r[k].local_rank & (kMaxRank - 1u));
            if (live[k] && pos < p.cap_hits) {
                if (stride == 4u) reinterpret_cast<uint4 *>(sorted)[pos] = make_uint4(r[k].keyhi, r[k].keylo, __float_as_uint(r[k].w), 0u);
                else reinterpret_cast<SortedRec *>(sorted)[pos] = SortedRec{r[k].keyhi, r[k].keylo, r[k].w, r[k].u, r[k].v, 0u};
            }
        }
    }
}

// Zeroes the dirty bricks (whole bricks, one 16-byte store per lane): leaves the dense grid clean for the next
// voxelization.  Runs after k_scatter has read the per-cell offsets.
__global__ __launch_bounds__(kBlock) void k_reset_bricks(uint32_t *grid, const uint32_t *__restrict__ dirty_list,
                                                         const Counters *c, Params p)
{
    const uint32_t n_dirty = c->n_dirty < p.cap_dirty ? c->n_dirty : p.cap_dirty;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t item = (blockIdx.x * (kBlock / 64) + wave) * kBricksPerLoad + lane / kLanesPerBrick; item < n_dirty;
         item += gridDim.x * (kBlock / 64) * kBricksPerLoad)
        reinterpret_cast<uint4 *>(grid + (uint64_t) dirty_list[item] * kBrickCells)[lane % kLanesPerBrick] = make_uint4(0, 0, 0, 0);
}
