// o2v_dev_k3_resolve.hpp -- K3: ordered per-cell replay (k_resolve and the cooperative tiers).
//
// Part of the device code of o2v_device.hip, which includes this file inside its anonymous namespace (one
// translation unit: the stages share records and launch parameters).  Not a stand-alone header.

// ---- K3: resolve ---------------------------------------------------------------------------------------------

// ---- ordered replay of one cell's hits --------------------------------------------------------------------------
// The hits of a cell arrive in arbitrary order; the reference's result is a sequential fold, so they are
// replayed in the reference's order, i.e. ascending in the key (sub-voxel, triangle index, leaf order):
//   leaves of one triangle   -> insertWeighted<BLEND>(uvBuffer, ...)  voxelization.cpp:466-468 (new, existing)
//   triangles, ascending     -> moveUvBufferIntoVoxels                voxelization.cpp:513-526 (new, existing)
//   sub-voxels, ascending    -> documented downscale semantics        voxelization.hpp:82-85
struct CellFold {
    bool have_tri = false, have_sub = false, have_cell = false;
    uint32_t cur_group = 0, sub_key = 0, cell_key = 0;  // MAX: the group (sub-voxel | triangle) that holds sub_acc / cell_acc
    WUv tri_acc{0, 0, 0};
    WCol sub_acc{0, 0, 0, 0}, cell_acc{0, 0, 0, 0};

    __device__ __forceinline__ void close_tri(const Materials &m, uint32_t blend)
    {
        float cr, cg, cb;
        color_at(m, cur_group & 0x1fffffffu, tri_acc.u, tri_acc.v, cr, cg, cb);
        const WCol fresh{tri_acc.w, cr, cg, cb};
        if (!have_sub || (!blend && fresh.w > sub_acc.w)) sub_key = cur_group;  // wmax keeps the existing value on a tie
        sub_acc = have_sub ? wcombine(blend, fresh, sub_acc) : fresh;
        have_sub = true;
        have_tri = false;
    }
    __device__ __forceinline__ void close_sub(uint32_t blend)
    {
        if (!have_cell || (!blend && sub_acc.w > cell_acc.w)) cell_key = sub_key;
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

// Where a resolved cell goes.  Normally its (x, y, z, argb) record is final.  On the direct MAX path (Params::direct_max)
// the cell may also have received hits of unsplit triangles straight from k_voxelize, so the winner of the hits resolved
// here - weight `w`, group `keyhi` = sub-voxel << 29 | triangle - competes in the same 64-bit cell and k_emit_max
// writes the record.  Returns true if the caller has to write the record `rec` to the output list itself.
__device__ __forceinline__ bool emit_cell(uint32_t brick, uint32_t local, uint32_t argb, float w, uint32_t keyhi, uint4 &rec,
                                          const Counters *c, const Params &p)
{
    if (direct_active(c, p)) {
        const uint64_t cell = (uint64_t) brick * kBrickCells + local;
        atomicMax(&p.maxgrid[cell], ((unsigned long long) __float_as_uint(w) << 32) | (0xffffffffu - keyhi));
        if (p.pick_max) {
            // textured mesh: the colour is known here, the winner of the cell only later (k_pick)
            const uint32_t slot = atomicAdd(const_cast<uint32_t *>(&c->pad2), 1u);
            if (slot < p.cap_vox) {
                uint32_t *q = p.pick_extra + (size_t) slot * 6u;
                q[0] = (uint32_t) cell;
                q[1] = (uint32_t) (cell >> 32);
                q[2] = keyhi;
                q[3] = __float_as_uint(w);
                q[4] = argb;
                q[5] = 0u;
            }
        }
        return false;
    }
    uint32_t x, y, z;
    cell_position(brick, local, p, x, y, z);
    rec = make_uint4(x, y, z + p.zo0, argb);
    return true;
}

// The fold over one cell's (sub-voxel, triangle) groups, each already reduced to {weight, colour}: CellFold's close_tri /
// close_sub sequence (moveUvBufferIntoVoxels, voxelization.cpp:513-526; downscale, voxelization.hpp:82-85) without the
// per-hit part.  Groups must be added in ascending key order.
struct GroupFold {
    bool have_sub = false, have_cell = false;
    uint32_t cur = 0, sub_key = 0, cell_key = 0;
    WCol sub_acc{0, 0, 0, 0}, cell_acc{0, 0, 0, 0};
    __device__ __forceinline__ void close_sub(uint32_t blend)
    {
        if (!have_cell || (!blend && sub_acc.w > cell_acc.w)) cell_key = sub_key;
        cell_acc = have_cell ? wcombine(blend, sub_acc, cell_acc) : sub_acc;
        have_cell = true;
        have_sub = false;
    }
    __device__ __forceinline__ void add(uint32_t blend, uint32_t keyhi, const WCol &fresh)
    {
        if (have_sub && (keyhi >> 29) != (cur >> 29)) close_sub(blend);
        if (!have_sub || (!blend && fresh.w > sub_acc.w)) sub_key = keyhi;  // wmax keeps the existing value on a tie
        sub_acc = have_sub ? wcombine(blend, fresh, sub_acc) : fresh;
        have_sub = true;
        cur = keyhi;
    }
    __device__ __forceinline__ uint32_t finish(uint32_t blend)
    {
        if (have_sub) close_sub(blend);
        return pack_argb(cell_acc.r, cell_acc.g, cell_acc.b);
    }
};

// Bitonic sort of (cell, key, idx) triples held in two arrays: `ci` = cell << 16 | idx (idx < 2^16), order by (cell, key).
template <typename KeyPtr, typename CiPtr>
__device__ __forceinline__ void bitonic_sort_ck(KeyPtr key, CiPtr ci, uint32_t n_pow2, uint32_t tid, uint32_t nthreads)
{
    for (uint32_t k = 2; k <= n_pow2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < n_pow2; t += nthreads) {
                const uint32_t partner = t ^ j;
                if (partner > t) {
                    const bool up = (t & k) == 0;
                    const uint64_t a = key[t], b = key[partner];
                    const uint32_t ca = ci[t], cb = ci[partner];
                    const bool a_gt_b = (ca >> 16) != (cb >> 16) ? (ca >> 16) > (cb >> 16) : a > b;
                    if (a_gt_b == up) {
                        key[t] = b;
                        key[partner] = a;
                        ci[t] = cb;
                        ci[partner] = ca;
                    }
                }
            }
            __syncthreads();
        }
    }
}
constexpr uint32_t kPadCell = 0xffffu;  // `cell` of the padding entries of a sort: after every real cell (< 64)

// ---- brick resolve: the bulk tiers, one wavefront per brick, out of registers ------------------------------------------
// Bricks with up to 64 * R pooled hits (R = 1, 2, 4 records per lane; nine bricks in ten have at most 64).  Per brick:
//   load     the brick's records are one contiguous block (k_scatter): one coalesced load per lane and record; right behind it
//            the lane asks for the material of its record's triangle (type, colour, texture index: independent loads that
//            arrive while the sort runs, so the colour lookup later has no chain of dependent loads left);
//   sort     by rank: every lane counts the records that precede its own in (cell, sub-voxel, triangle, leaf) order, the
//            other records' keys coming out of the lanes' registers one by one (v_readlane, the index is wavefront-uniform):
//            64 * R steps of a few instructions, no LDS, no barrier; the records are then written to LDS at their rank;
//   A        the lane at the first record of a (cell, sub-voxel, triangle) group folds the group - the leaves of one
//            triangle in one (sub-)voxel, insertWeighted<BLEND> (voxelization.cpp:466-468) in leaf order - and looks its
//            colour up (colorAt_f, triangle.hpp:181-194);
//   B        the lane at the first record of a cell folds the cell's groups in order (GroupFold) and writes the record to
//            its place in the output (BrickOcc::out_base + the number of occupied cells before it).
// The next brick's list entry is requested before the current one is processed.  Workgroup b takes the bricks b, b + grid,
// ... of its tier's list (an atomic per brick on one address would serialise at ~88 per us).
template <uint32_t R>
__global__ __launch_bounds__(64) void k_resolve_brick_wave(const BrickOcc *__restrict__ list, const uint32_t *n_list, Counters *c,
                                                           SortedView sorted, Materials m, uint4 *out, uint32_t list_cap, Params p)
{
    constexpr uint32_t N = 64u * R;
    if (pass_overflowed(c, p)) return;
    __shared__ uint32_t s_cell[N];  // cell | 0x100 at a group's first record
    __shared__ uint32_t s_hi[N], s_lo[N], s_mat[N];
    __shared__ float s_w[N], s_u[N], s_v[N], s_r[N], s_g[N], s_b[N];
    const uint32_t total = *n_list < list_cap ? *n_list : list_cap;
    const uint32_t lane = threadIdx.x;
    uint32_t item = blockIdx.x;
    if (item >= total) return;
    BrickOcc o = list[item];
#ifdef O2V_INSTRUMENT
    unsigned long long tmr[5] = {0, 0, 0, 0, 0}, t_mark = __builtin_readcyclecounter();
    auto lap = [&](int which) {
        const unsigned long long now = __builtin_readcyclecounter();
        tmr[which] += now - t_mark;
        t_mark = now;
    };
#define O2V_RLAP(i) lap(i)
#else
#define O2V_RLAP(i) do { } while (0)
#endif
    for (;;) {
        const uint32_t item_next = item + gridDim.x;
        const bool more = item_next < total;
        BrickOcc o_next{};
        if (more) o_next = list[item_next];
        const uint32_t n = o.count < N ? o.count : N;
        // ---- load ----
        uint32_t a_hi[R], a_lo[R], kb[R], mat[R], rank[R];
        float w[R], u[R], v[R], cr[R], cg[R], cb[R];
#pragma unroll
        for (uint32_t k = 0; k < R; ++k) {
            const uint32_t t = lane + 64u * k;
            a_hi[k] = 0xffffffffu;  // (not a record: after every real one)
            a_lo[k] = kb[k] = mat[k] = rank[k] = 0;
            w[k] = u[k] = v[k] = cr[k] = cg[k] = cb[k] = 0.f;
            if (t < n) {
                const SortedRec r = sorted.load(o.offset + t);
                a_hi[k] = r.pad;
                a_lo[k] = r.keyhi;
                kb[k] = r.keylo;
                w[k] = r.w;
                u[k] = r.u;
                v[k] = r.v;
                const uint32_t tri = r.keyhi & 0x1fffffffu;
                const uint32_t type = m.types ? m.types[tri] : (uint32_t) kTriMaterialless;
                const uint32_t texid = m.texids ? (uint32_t) m.texids[tri] : 0u;
                if (m.colors) {
                    cr[k] = m.colors[(size_t) tri * 3 + 0];
                    cg[k] = m.colors[(size_t) tri * 3 + 1];
                    cb[k] = m.colors[(size_t) tri * 3 + 2];
                }
                mat[k] = (type & 0xffu) | (texid << 8);
            }
        }
        O2V_RLAP(0);
        // ---- sort by rank ----
        // by (cell, sub-voxel, triangle) and, among equal ones, by position: one unique 64-bit key per record (6 + 32 + 8
        // bits), so a step is two v_readlane, one 64-bit compare and one add per own record.  The leaves of one triangle in one
        // (sub-)voxel - equal keys, few and rare - are put into leaf order by the group's fold (A).
        uint64_t key[R];
#pragma unroll
        for (uint32_t k = 0; k < R; ++k) key[k] = ((((uint64_t) a_hi[k] << 32) | a_lo[k]) << 8) | (lane + 64u * k);
#pragma unroll
        for (uint32_t kk = 0; kk < R; ++kk) {
            const uint32_t here = n > 64u * kk ? (n - 64u * kk < 64u ? n - 64u * kk : 64u) : 0u;  // records in register set kk
            const uint32_t klo = (uint32_t) key[kk], khi = (uint32_t) (key[kk] >> 32);
            for (uint32_t j = 0; j < here; ++j) {
                const uint64_t other = ((uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) khi, (int) j) << 32) |
                                       (uint32_t) __builtin_amdgcn_readlane((int) klo, (int) j);
#pragma unroll
                for (uint32_t k = 0; k < R; ++k) rank[k] += other < key[k] ? 1u : 0u;
            }
        }
        O2V_RLAP(1);
        __syncthreads();  // (one wavefront: orders the LDS accesses of the previous brick before these writes)
#pragma unroll
        for (uint32_t k = 0; k < R; ++k) {
            if (lane + 64u * k < n) {
                const uint32_t at = rank[k];
                s_cell[at] = a_hi[k];
                s_hi[at] = a_lo[k];
                s_lo[at] = kb[k];
                s_mat[at] = mat[k];
                s_w[at] = w[k];
                s_u[at] = u[k];
                s_v[at] = v[k];
                s_r[at] = cr[k];
                s_g[at] = cg[k];
                s_b[at] = cb[k];
            }
        }
        __syncthreads();
        O2V_RLAP(2);
        // ---- A: groups -> {weight, colour} at the group's first record ----
        bool start[R];
        float gw[R], gr[R], gg[R], gb[R];
#pragma unroll
        for (uint32_t k = 0; k < R; ++k) {
            const uint32_t t = lane + 64u * k;
            start[k] = false;
            gw[k] = gr[k] = gg[k] = gb[k] = 0.f;
            if (t < n) {
                const uint32_t cell = s_cell[t], hi = s_hi[t];
                start[k] = t == 0 || s_hi[t - 1] != hi || s_cell[t - 1] != cell;
                if (start[k]) {
                    WUv acc{s_w[t], s_u[t], s_v[t]};
                    uint32_t end = t + 1;
                    while (end < n && s_hi[end] == hi && s_cell[end] == cell) ++end;
                    if (end > t + 1) {
                        // several leaves of the triangle in this (sub-)voxel: fold them in leaf order (selection by the leaf key;
                        // a handful at most: a leaf spans several voxels)
                        uint32_t last = 0;
                        bool first = true;
                        for (uint32_t done = t; done < end; ++done) {
                            uint32_t best = 0xffffffffu, at = t;
                            for (uint32_t j = t; j < end; ++j) {
                                const uint32_t kl = s_lo[j];
                                if ((first || kl > last) && kl <= best) {
                                    best = kl;
                                    at = j;
                                }
                            }
                            const WUv hit{s_w[at], s_u[at], s_v[at]};
                            acc = first ? hit : wmix(hit, acc);
                            last = best;
                            first = false;
                        }
                    }
                    const uint32_t mt = s_mat[t], type = mt & 0xffu;
                    gw[k] = acc.w;
                    if (type == kTriMaterialless) {
                        gr[k] = gg[k] = gb[k] = 1.f;
                    }
                    else if (type == kTriUntextured) {
                        gr[k] = s_r[t];
                        gg[k] = s_g[t];
                        gb[k] = s_b[t];
                    }
                    else if (type == kTriTextured && m.n_textures) {
                        texel_color(m, mt >> 8, acc.u, acc.v, gr[k], gg[k], gb[k]);
                    }
                    else {
                        gr[k] = 1.f;
                        gg[k] = 0.f;
                        gb[k] = 1.f;
                    }
                }
            }
        }
        __syncthreads();  // (every group has read its members before the first records are overwritten)
#pragma unroll
        for (uint32_t k = 0; k < R; ++k) {
            const uint32_t t = lane + 64u * k;
            if (start[k]) {
                s_w[t] = gw[k];
                s_r[t] = gr[k];
                s_g[t] = gg[k];
                s_b[t] = gb[k];
                s_cell[t] |= 0x100u;
            }
        }
        __syncthreads();
        O2V_RLAP(3);
        // ---- B: cells ----
#pragma unroll
        for (uint32_t k = 0; k < R; ++k) {
            const uint32_t t = lane + 64u * k;
            if (t < n) {
                const uint32_t cell = s_cell[t] & 0xffu;
                if (t == 0 || (s_cell[t - 1] & 0xffu) != cell) {
                    GroupFold f;
                    for (uint32_t j = t; j < n; ++j) {
                        const uint32_t cj = s_cell[j];
                        if ((cj & 0xffu) != cell) break;
                        if (cj & 0x100u) f.add(p.blend, s_hi[j], WCol{s_w[j], s_r[j], s_g[j], s_b[j]});
                    }
                    const uint32_t argb = f.finish(p.blend);
                    uint4 rec;
                    if (emit_cell(o.brick, cell, argb, f.cell_acc.w, f.cell_key, rec, c, p)) {
                        const uint32_t slot = o.out_base + (uint32_t) __popcll(o.cells & ((1ull << cell) - 1ull));
                        if (slot < p.cap_vox) out[slot] = rec;
                    }
                }
            }
        }
        O2V_RLAP(4);
        if (!more) break;
        item = item_next;
        o = o_next;
    }
#ifdef O2V_INSTRUMENT
    if (R == 1 && lane == 0)
        for (uint32_t k = 0; k < 5; ++k) atomicAdd(&c->dbg[k], tmr[k]);  // (k_voxelize's counts are overwritten: run a BLEND workload)
#endif
}

// ---- brick resolve: tiers up to kTierLong hits -------------------------------------------------------------------------
// THREADS lanes (a wavefront or a workgroup) take one brick at a time from their tier's list: the brick's records are one
// contiguous block (k_scatter), loaded coalesced; (cell, key, position) is bitonic-sorted in LDS; the payload is gathered in
// sorted order; then, all in parallel,
//   A  the lane at the first record of a (cell, sub-voxel, triangle) group folds the group - the leaves of one triangle in one
//      (sub-)voxel, insertWeighted<BLEND> (voxelization.cpp:466-468) in leaf order - and looks its colour up (colorAt_f),
//   B  the lane at the first record of a cell folds the cell's groups in order (GroupFold) and makes the record;
// every record goes straight to its place in the output (BrickOcc::out_base + the number of occupied cells before it).
// DYNAMIC: bricks are taken from a global cursor (few, uneven items); otherwise workgroup b takes the items b, b + grid, ...
// (an atomic per brick on one address serialises at ~88 per us: 23 ms for the two million bricks of configs[3]).
// PARK: the payload (w, u, v) is parked in LDS at its unsorted position when the records are read; otherwise it is read again
// from global memory in sorted order (the tier of up to 4096 hits: 128 KiB of LDS without the parking area).
template <uint32_t THREADS, uint32_t CAP, bool DYNAMIC, bool PARK>
__global__ __launch_bounds__(THREADS) void k_resolve_brick(const BrickOcc *__restrict__ list, const uint32_t *n_list, uint32_t *cursor,
                                                           Counters *c, SortedView sorted, Materials m, uint4 *out, uint32_t list_cap,
                                                           Params p)
{
    if (pass_overflowed(c, p)) return;
    __shared__ uint64_t s_key[CAP];
    __shared__ uint32_t s_ci[CAP];
    __shared__ uint32_t s_hi[CAP];
    __shared__ float s_w[CAP], s_u[CAP], s_v[CAP], s_b[CAP];
    __shared__ float s_w0[PARK ? CAP : 1], s_u0[PARK ? CAP : 1], s_v0[PARK ? CAP : 1];
    __shared__ uint32_t s_item;
    const uint32_t total = *n_list < list_cap ? *n_list : list_cap;
    for (uint32_t round = 0;; ++round) {
        __syncthreads();
        if (DYNAMIC) {
            if (threadIdx.x == 0) s_item = atomicAdd(cursor, 1u);
            __syncthreads();
        }
        const uint32_t item = DYNAMIC ? s_item : blockIdx.x + round * gridDim.x;
        if (item >= total) break;
        const BrickOcc o = list[item];
        const uint32_t n = o.count < CAP ? o.count : CAP;
        uint32_t n_pow2 = 1;
        while (n_pow2 < n) n_pow2 <<= 1;
        // the records are read once: keys for the sort, the payload parked in LDS at its unsorted position
        for (uint32_t t = threadIdx.x; t < n_pow2; t += THREADS) {
            if (t < n) {
                const SortedRec r = sorted.load(o.offset + t);
                s_key[t] = ((uint64_t) r.keyhi << 32) | r.keylo;
                s_ci[t] = (r.pad << 16) | t;
                if (PARK) {
                    s_w0[t] = r.w;
                    s_u0[t] = r.u;
                    s_v0[t] = r.v;
                }
            }
            else {
                s_key[t] = ~0ull;
                s_ci[t] = kPadCell << 16;
            }
        }
        __syncthreads();
        bitonic_sort_ck(s_key, s_ci, n_pow2, threadIdx.x, THREADS);
        for (uint32_t t = threadIdx.x; t < n; t += THREADS) {
            const uint32_t from = s_ci[t] & 0xffffu;
            s_hi[t] = (uint32_t) (s_key[t] >> 32);
            if (PARK) {
                s_w[t] = s_w0[from];
                s_u[t] = s_u0[from];
                s_v[t] = s_v0[from];
            }
            else {
                const SortedRec r = sorted.load(o.offset + from);
                s_w[t] = r.w;
                s_u[t] = r.u;
                s_v[t] = r.v;
            }
        }
        __syncthreads();
        // A: groups -> {weight, r, g, b} at the group's first record (s_w, s_u, s_v, s_b)
        bool starts[(CAP + THREADS - 1) / THREADS];
#pragma unroll
        for (uint32_t k = 0; k < (CAP + THREADS - 1) / THREADS; ++k) {
            const uint32_t t = threadIdx.x + k * THREADS;
            starts[k] = t < n && (t == 0 || s_hi[t] != s_hi[t - 1] || (s_ci[t] >> 16) != (s_ci[t - 1] >> 16));
        }
        float gw[(CAP + THREADS - 1) / THREADS], gr[(CAP + THREADS - 1) / THREADS], gg[(CAP + THREADS - 1) / THREADS],
            gb[(CAP + THREADS - 1) / THREADS];
#pragma unroll
        for (uint32_t k = 0; k < (CAP + THREADS - 1) / THREADS; ++k) {
            const uint32_t t = threadIdx.x + k * THREADS;
            gw[k] = gr[k] = gg[k] = gb[k] = 0.f;
            if (starts[k]) {
                const uint32_t cell = s_ci[t] >> 16;
                WUv acc{s_w[t], s_u[t], s_v[t]};
                for (uint32_t j = t + 1; j < n && s_hi[j] == s_hi[t] && (s_ci[j] >> 16) == cell; ++j) acc = wmix(WUv{s_w[j], s_u[j], s_v[j]}, acc);
                color_at(m, s_hi[t] & 0x1fffffffu, acc.u, acc.v, gr[k], gg[k], gb[k]);
                gw[k] = acc.w;
            }
        }
        __syncthreads();  // (every group has read its members' w, u, v before the first records are overwritten)
#pragma unroll
        for (uint32_t k = 0; k < (CAP + THREADS - 1) / THREADS; ++k) {
            const uint32_t t = threadIdx.x + k * THREADS;
            if (starts[k]) {
                s_w[t] = gw[k];
                s_u[t] = gr[k];
                s_v[t] = gg[k];
                s_b[t] = gb[k];
                s_ci[t] |= 0x8000u;  // marks a group's first record (idx is no longer needed; CAP <= 2^15)
            }
        }
        __syncthreads();
        // B: cells
#pragma unroll
        for (uint32_t k = 0; k < (CAP + THREADS - 1) / THREADS; ++k) {
            const uint32_t t = threadIdx.x + k * THREADS;
            if (t < n && (t == 0 || (s_ci[t] >> 16) != (s_ci[t - 1] >> 16))) {
                const uint32_t cell = s_ci[t] >> 16;
                GroupFold f;
                for (uint32_t j = t; j < n && (s_ci[j] >> 16) == cell; ++j)
                    if (s_ci[j] & 0x8000u) f.add(p.blend, s_hi[j], WCol{s_w[j], s_u[j], s_v[j], s_b[j]});
                const uint32_t argb = f.finish(p.blend);
                uint4 rec;
                if (emit_cell(o.brick, cell, argb, f.cell_acc.w, f.cell_key, rec, c, p)) {
                    // the brick's records follow each other in ascending order of the cell (k_scan_bcount made room for them)
                    const uint32_t slot = o.out_base + (uint32_t) __popcll(o.cells & ((1ull << cell) - 1ull));
                    if (slot < p.cap_vox) out[slot] = rec;
                }
            }
        }
    }
}

// ---- brick resolve: the long tail ----------------------------------------------------------------------------------------
// Bricks with more than kTierLong hits (the poles of a finely tessellated sphere; a whole mesh inside a few voxels): one
// workgroup per brick sorts (cell, key, position) - up to kTierBig in 96 KiB of dynamic LDS (key 8 B + cell / position 4 B),
// beyond that in a global scratch area - and its first thread replays the records in order (CellFold per cell; sequential
// by nature for BLEND, and rare enough for MAX), the payload staged through LDS a thousand records at a time.
constexpr uint32_t kBigStage = 1024, kBigThreads = 1024;
template <bool GLOBAL_SCRATCH>
__global__ __launch_bounds__(kBigThreads) void k_resolve_brick_big(const BrickOcc *__restrict__ list, const uint32_t *n_list, uint32_t *cursor,
                                                                   Counters *c, SortedView sorted, Materials m, uint4 *out,
                                                                   uint64_t *scratch_key, uint32_t *scratch_idx, uint32_t scratch_cap,
                                                                   uint32_t list_cap, Params p)
{
    extern __shared__ __align__(16) unsigned char s_dyn[];
    __shared__ uint32_t s_hi[kBigStage], s_cell[kBigStage];
    __shared__ float s_w[kBigStage], s_u[kBigStage], s_v[kBigStage];
    __shared__ uint32_t s_item, s_base, s_ok;
    if (pass_overflowed(c, p)) return;
    const uint32_t total = *n_list < list_cap ? *n_list : list_cap;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_item = atomicAdd(cursor, 1u);
        __syncthreads();
        const uint32_t item = s_item;
        if (item >= total) break;
        const BrickOcc o = list[item];
        const uint32_t n = GLOBAL_SCRATCH ? o.count : (o.count < kTierBig ? o.count : kTierBig);
        uint32_t n_pow2 = 1;
        while (n_pow2 < n) n_pow2 <<= 1;
        uint64_t *key = reinterpret_cast<uint64_t *>(s_dyn);
        uint32_t *idx = reinterpret_cast<uint32_t *>(s_dyn + (size_t) kTierBig * 8);
        uint32_t *cellv = nullptr;  // GLOBAL_SCRATCH: the position does not fit beside the cell in 32 bits
        if (GLOBAL_SCRATCH) {
            if (threadIdx.x == 0) {
                s_base = atomicAdd(&c->scratch_used, 2u * n_pow2);
                // scratch too small: the host sees scratch_used > capacity, grows it and re-runs
                s_ok = (uint64_t) s_base + 2ull * n_pow2 <= scratch_cap ? 1u : 0u;
            }
            __syncthreads();
            if (!s_ok) continue;
            key = scratch_key + s_base;
            idx = scratch_idx + s_base;
            cellv = scratch_idx + s_base + n_pow2;
        }
        for (uint32_t t = threadIdx.x; t < n_pow2; t += kBigThreads) {
            if (t < n) {
                const SortedRec r = sorted.load(o.offset + t);
                key[t] = ((uint64_t) r.keyhi << 32) | r.keylo;
                if (GLOBAL_SCRATCH) {
                    idx[t] = t;
                    cellv[t] = r.pad;
                }
                else {
                    idx[t] = (r.pad << 16) | t;
                }
            }
            else {
                key[t] = ~0ull;
                if (GLOBAL_SCRATCH) {
                    idx[t] = 0;
                    cellv[t] = kPadCell;
                }
                else {
                    idx[t] = kPadCell << 16;
                }
            }
        }
        __syncthreads();
        if (GLOBAL_SCRATCH) {
            for (uint32_t k = 2; k <= n_pow2; k <<= 1) {
                for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                    for (uint32_t t = threadIdx.x; t < n_pow2; t += kBigThreads) {
                        const uint32_t partner = t ^ j;
                        if (partner > t) {
                            const bool up = (t & k) == 0;
                            const uint64_t a = key[t], b = key[partner];
                            const uint32_t ca = cellv[t], cb = cellv[partner];
                            const bool a_gt_b = ca != cb ? ca > cb : a > b;
                            if (a_gt_b == up) {
                                key[t] = b;
                                key[partner] = a;
                                cellv[t] = cb;
                                cellv[partner] = ca;
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
        else {
            bitonic_sort_ck(key, idx, n_pow2, threadIdx.x, kBigThreads);
        }
        CellFold f;  // only thread 0's copy is used
        uint32_t cur_cell = kPadCell, n_emitted = 0;
        auto emit = [&]() {
            const uint32_t argb = f.finish(m, p.blend);
            uint4 rec;
            if (emit_cell(o.brick, cur_cell, argb, f.cell_acc.w, f.cell_key, rec, c, p) && o.out_base + n_emitted < p.cap_vox)
                out[o.out_base + n_emitted] = rec;
            n_emitted += 1;
        };
        for (uint32_t base = 0; base < n; base += kBigStage) {
            const uint32_t m_here = n - base < kBigStage ? n - base : kBigStage;
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < m_here; t += kBigThreads) {
                const uint32_t at = GLOBAL_SCRATCH ? idx[base + t] : (idx[base + t] & 0xffffu);
                const SortedRec r = sorted.load(o.offset + at);
                s_hi[t] = r.keyhi;
                s_cell[t] = r.pad;
                s_w[t] = r.w;
                s_u[t] = r.u;
                s_v[t] = r.v;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                for (uint32_t t = 0; t < m_here; ++t) {
                    if (s_cell[t] != cur_cell) {
                        if (cur_cell != kPadCell) emit();
                        f = CellFold{};
                        cur_cell = s_cell[t];
                    }
                    f.add(m, p.blend, s_hi[t], s_w[t], s_u[t], s_v[t]);
                }
            }
        }
        if (threadIdx.x == 0 && cur_cell != kPadCell) emit();
    }
}

// ---- direct MAX path: emission -----------------------------------------------------------------------------------
// MAX strategy without textured triangles: `new.w > existing.w ? new : existing` over ascending (sub-voxel, triangle)
// is a true reduction - the greatest weight wins, ties go to the smaller (sub-voxel, triangle) - and a triangle's colour
// does not depend on where it was hit, so a voxel only has to remember max{weight bits << 32 | ~(sub << 29 | triangle)}
// (weights are positive floats: their bit patterns order like the values).  k_voxelize feeds that cell directly for
// unsplit triangles, the resolve kernels add the summed-up subdivided ones, and this kernel turns every non-zero
// cell of the dirty bricks into its (x, y, z, argb) record and zeroes it again (moveUvBufferIntoVoxels + the pack of
// obj2voxel.cpp:279-297).  Records are staged in LDS so that a workgroup reserves output space once per ~1024 voxels.
// Textured MAX: every record {cell, key, argb} - left by k_voxelize for direct hits (in the hit pool) and by the replay
// tiers for the cells they resolved - is compared with the final maximum of its cell; the one that won replaces it by
// its colour.  Keys are unique per cell, so exactly one record matches and nothing races.
__global__ __launch_bounds__(kBlock) void k_pick(const HitRec *__restrict__ pool, const Counters *c, Params p)
{
    if (!direct_active(c, p)) return;
    const uint32_t n = c->n_hits_reserved < p.cap_hits ? c->n_hits_reserved : p.cap_hits;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const HitRec r = pool[i];
        if (r.brick == kHoleBrick || r.pad != kPickRecord) continue;
        const uint64_t cell = (uint64_t) r.brick * kBrickCells + (r.local_rank >> 24);
        const unsigned long long key = ((unsigned long long) __float_as_uint(r.w) << 32) | (0xffffffffu - r.keyhi);
        if (p.maxgrid[cell] == key) p.maxgrid[cell] = kPickTag | r.keylo;
    }
    const uint32_t n2 = c->pad2 < p.cap_vox ? c->pad2 : p.cap_vox;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n2; i += gridDim.x * kBlock) {
        const uint32_t *q = p.pick_extra + (size_t) i * 6u;
        const uint64_t cell = ((uint64_t) q[1] << 32) | q[0];
        const unsigned long long key = ((unsigned long long) q[3] << 32) | (0xffffffffu - q[2]);
        if (p.maxgrid[cell] == key) p.maxgrid[cell] = kPickTag | q[4];
    }
}

constexpr uint32_t kEmitBricksPerWave = 2;
constexpr uint32_t kEmitBricksPerRound = (kBlock / 64) * kEmitBricksPerWave * kBricksPerLoad;
constexpr uint32_t kEmitFlushAt = 1024;  // 48 KiB of staging: three workgroups per CU keep enough brick loads (2 KiB per wavefront) in flight
                                         // (2048 / two workgroups: 0.15 ms on the bench mesh, this: 0.12; 512 / four: 0.16)
constexpr uint32_t kEmitCap = kEmitFlushAt + kEmitBricksPerRound * kBrickCells;

__global__ __launch_bounds__(kBlock) void k_emit_max(const uint32_t *__restrict__ dirty_list, Counters *c, Materials m, uint4 *out,
                                                     Params p)
{
    __shared__ uint4 s_rec[kEmitCap];
    __shared__ uint32_t s_n, s_base;
    if (!direct_active(c, p)) return;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint32_t n_dirty = c->n_dirty_max < p.cap_dirty ? c->n_dirty_max : p.cap_dirty;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n_rounds = (n_dirty + kEmitBricksPerRound - 1) / kEmitBricksPerRound;
    auto flush = [&](uint32_t n) {
        if (threadIdx.x == 0) s_base = atomicAdd(&c->n_out, n);
        __syncthreads();
        const uint32_t base = s_base;
        for (uint32_t i = threadIdx.x; i < n; i += kBlock)
            if (base + i < p.cap_vox) out[base + i] = s_rec[i];
        __syncthreads();
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
    };
    for (uint32_t r = blockIdx.x; r < n_rounds; r += gridDim.x) {
        uint32_t brick[kEmitBricksPerWave];
        ulonglong2 lo[kEmitBricksPerWave], hi[kEmitBricksPerWave];
#pragma unroll
        for (uint32_t k = 0; k < kEmitBricksPerWave; ++k) {
            const uint32_t item = r * kEmitBricksPerRound + (wave * kEmitBricksPerWave + k) * kBricksPerLoad + lane / kLanesPerBrick;
            brick[k] = item < n_dirty ? dirty_list[item] : 0xffffffffu;
        }
#pragma unroll
        for (uint32_t k = 0; k < kEmitBricksPerWave; ++k) {
            lo[k] = hi[k] = make_ulonglong2(0, 0);
            if (brick[k] != 0xffffffffu) {
                const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(p.maxgrid + (uint64_t) brick[k] * kBrickCells) + (lane % kLanesPerBrick) * 2u;
                lo[k] = q[0];
                hi[k] = q[1];
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < kEmitBricksPerWave; ++k) {
            const unsigned long long v4[4] = {lo[k].x, lo[k].y, hi[k].x, hi[k].y};
            if (v4[0] | v4[1] | v4[2] | v4[3]) {
                const uint32_t row = brick[k] / p.NBx;
                const uint32_t bx = brick[k] - row * p.NBx;
                const uint32_t bz = row / p.NBy;
                const uint32_t by = row - bz * p.NBy;
#pragma unroll
                for (uint32_t e = 0; e < 4; ++e) {
                    if (v4[e]) {
                        const uint32_t local = (lane % kLanesPerBrick) * 4u + e;
                        uint32_t argb;
                        if (v4[e] & kPickTag) {
                            argb = (uint32_t) v4[e];  // textured mesh: k_pick already put the winner's colour here
                        }
                        else {
                            const uint32_t keyhi = 0xffffffffu - (uint32_t) v4[e];
                            float cr, cg, cb;
                            color_at(m, keyhi & 0x1fffffffu, 0.f, 0.f, cr, cg, cb);
                            argb = pack_argb(cr, cg, cb);
                        }
                        const uint32_t slot = atomicAdd(&s_n, 1u);
                        s_rec[slot] = make_uint4((bx << kBrickXs) + (local & (kBrickX - 1u)),
                                                 (by << kBrickYs) + ((local >> kBrickXs) & (kBrickY - 1u)),
                                                 (bz << kBrickZs) + (local >> (kBrickXs + kBrickYs)) + p.zo0, argb);
                    }
                }
                // leave the cells clean for the next run
                ulonglong2 *q = reinterpret_cast<ulonglong2 *>(p.maxgrid + (uint64_t) brick[k] * kBrickCells) + (lane % kLanesPerBrick) * 2u;
                q[0] = make_ulonglong2(0, 0);
                q[1] = make_ulonglong2(0, 0);
            }
        }
        __syncthreads();
        const uint32_t n = s_n;
        __syncthreads();  // (every thread has read n before anyone adds to s_n again: the decision below must be uniform)
        if (n >= kEmitFlushAt) flush(n);
    }
    __syncthreads();
    const uint32_t n = s_n;
    if (n) flush(n);
}
