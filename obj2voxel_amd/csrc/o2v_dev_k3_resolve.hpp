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

__device__ __forceinline__ uint4 cell_record(const Occ &o, uint32_t argb, const Params &p)
{
    const uint64_t cell = o.cell();
    uint32_t x, y, z;
    cell_position((uint32_t) (cell >> kBrickShift), (uint32_t) cell & (kBrickCells - 1u), p, x, y, z);
    return make_uint4(x, y, z, argb);
}

// Where a resolved cell goes.  Normally its (x, y, z, argb) record is final.  On the direct MAX path (Params::direct_max)
// the cell may also have received hits of unsplit triangles straight from k_voxelize, so the winner of the hits resolved
// here - weight `w`, group `keyhi` = sub-voxel << 29 | triangle - competes in the same 64-bit cell and k_emit_max
// writes the record.
__device__ __forceinline__ void emit_cell(const Occ &o, uint32_t argb, float w, uint32_t keyhi, uint4 *out, uint32_t i,
                                          const Counters *c, const Params &p)
{
    if (direct_active(c, p)) {
        const uint64_t cell = o.cell();
        atomicMax(&p.maxgrid[cell], ((unsigned long long) __float_as_uint(w) << 32) | (0xffffffffu - keyhi));
        if (p.pick_max) {
            // textured mesh: the colour is known here, the winner of the cell only later (k_pick)
            const uint32_t slot = atomicAdd(const_cast<uint32_t *>(&c->pad2), 1u);
            if (slot < p.cap_vox) {
                uint32_t *q = p.pick_extra + (size_t) slot * 6u;
                q[0] = o.cell_lo;
                q[1] = o.cell_hi & 31u;
                q[2] = keyhi;
                q[3] = __float_as_uint(w);
                q[4] = argb;
                q[5] = 0u;
            }
        }
    }
    else {
        out[i] = cell_record(o, argb, p);
    }
}


// The fold over one cell's (sub-voxel, triangle) groups, each already reduced to {weight, colour}: CellFold's close_tri /
// close_sub sequence (moveUvBufferIntoVoxels, voxelization.cpp:513-526; downscale, voxelization.hpp:82-85) without the
// per-hit part and without its memory accesses.  Groups must be added in ascending key order.
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

constexpr uint32_t kTexCache = 16;  // texture descriptors k_resolve keeps in LDS (more textures: read from global memory)

// Tier 1: one lane per occupied cell.  Cells with up to 8 hits (the common case) are sorted in the lane's registers (a
// sorting network) from their contiguous records; longer ones are deferred, by hit count, to the cooperative kernels below.
// Batcher's odd-even merge sort as a list of compare-exchange pairs (N a power of two): 19 pairs for 8 keys, 63 for 16.
template <uint32_t N>
struct SortNet {
    uint8_t a[N * 4], b[N * 4];
    uint32_t n = 0;
    constexpr SortNet() : a{}, b{}
    {
        for (uint32_t q = 1; q < N; q *= 2)
            for (uint32_t k = q; k >= 1; k /= 2)
                for (uint32_t j = k % q; j + k < N; j += 2 * k)
                    for (uint32_t i = 0; i < k && i + j + k < N; ++i)
                        if ((i + j) / (2 * q) == (i + j + k) / (2 * q)) {
                            a[n] = (uint8_t) (i + j);
                            b[n] = (uint8_t) (i + j + k);
                            ++n;
                        }
    }
};

// One cell with up to N hits, resolved by one lane out of its registers (compile-time indices only): the records are sorted
// by a sorting network, reduced to their (sub-voxel, triangle) groups in one forward pass, coloured four groups at a time
// and folded.  Returns the cell's ARGB; `f` holds the winner for the direct MAX path.
// load_record(k) fetches the cell's k-th hit (k < count).
template <uint32_t STRIDE, uint32_t N, class Load>
__device__ __forceinline__ uint32_t resolve_cell_in_registers(const Load &load_record, uint32_t count, const Materials &m,
                                                              const DevTexture *s_tex, const Params &p, GroupFold &f)
{
    struct {
        uint32_t count;
    } o{count};
    constexpr bool kUv = STRIDE == 6;  // 16-byte records carry no uv
    // All loads are issued before anything is consumed (independent round trips overlap); slots beyond the cell's count get
    // the greatest key and sort to the end.
    uint64_t key[N];
    float w[N], u[N], v[N];
    {
        SortedRec r[N];
#pragma unroll
        for (uint32_t k = 0; k < N; ++k) {
            r[k] = SortedRec{0u, 0u, 0.f, 0.f, 0.f, 0u};
            if (k < o.count) r[k] = load_record(k);  // (lanes without a k-th hit stay out of the load)
        }
#pragma unroll
        for (uint32_t k = 0; k < N; ++k) {
            key[k] = k < o.count ? (((uint64_t) r[k].keyhi << 32) | r[k].keylo) : ~0ull;
            w[k] = r[k].w;
            u[k] = kUv ? r[k].u : 0.f;
            v[k] = kUv ? r[k].v : 0.f;
        }
    }
    // no memory, no data-dependent loop (round 2 insertion-sorted a private LDS column: 40 KiB per workgroup, three wavefronts
    // per SIMD, a dependent LDS round trip per step)
    constexpr SortNet<N> net{};
#pragma unroll
    for (uint32_t e = 0; e < net.n; ++e) {
        const uint32_t a = net.a[e], b = net.b[e];
        // (selects with the mask named: a run of two-operand v_cndmask is slow on gfx950, see vsel in o2v_dev_arith.hpp)
        const unsigned long long sw = lane_mask(key[a] > key[b]);
        const uint32_t alo = (uint32_t) key[a], ahi = (uint32_t) (key[a] >> 32), blo = (uint32_t) key[b], bhi = (uint32_t) (key[b] >> 32);
        key[a] = ((uint64_t) vsel(sw, bhi, ahi) << 32) | vsel(sw, blo, alo);
        key[b] = ((uint64_t) vsel(sw, ahi, bhi) << 32) | vsel(sw, alo, blo);
        const float wa = w[a], wb = w[b];
        w[a] = vsel(sw, wb, wa);
        w[b] = vsel(sw, wa, wb);
        if (kUv) {
            const float ua = u[a], ub = u[b], va = v[a], vb = v[b];
            u[a] = vsel(sw, ub, ua);
            u[b] = vsel(sw, ua, ub);
            v[a] = vsel(sw, vb, va);
            v[b] = vsel(sw, va, vb);
        }
    }
    // The fold in three steps.  (1) One pass forward reduces every (sub-voxel, triangle) group - the leaves of one triangle
    // in one (sub-)voxel, insertWeighted<BLEND> (voxelization.cpp:466-468) - into the slot of its LAST record.
    // (2) The groups' colours (colorAt_f) are looked up four slots at a time, every step's loads independent of each other:
    // a colour is a chain of dependent loads (type -> colour or texture index -> texel), and folding record by record
    // walked that chain once per group, one after the other.  (3) The cell's chain over the groups (GroupFold).
    uint32_t hi[N];
    bool last[N];
    {
        WUv acc{0.f, 0.f, 0.f};
#pragma unroll
        for (uint32_t t = 0; t < N; ++t) {
            hi[t] = (uint32_t) (key[t] >> 32);
            const bool start = t == 0 || hi[t] != hi[t - 1];
            const WUv hit{w[t], u[t], v[t]};
            acc = start ? hit : wmix(hit, acc);
            last[t] = t < o.count && (t + 1 >= o.count || (uint32_t) (key[t + 1 < N ? t + 1 : t] >> 32) != hi[t]);
            w[t] = acc.w;
            u[t] = acc.u;
            v[t] = acc.v;
        }
    }
#pragma unroll
    for (uint32_t g0 = 0; g0 < N; g0 += 4u) {
        if (g0 >= o.count) continue;
        MatFetch mf[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            mf[j] = MatFetch{kTriMaterialless, 0u, 0.f, 0.f, 0.f};
            if (last[g0 + j]) mf[j] = mat_fetch(m, hi[g0 + j] & 0x1fffffffu);  // (only the slots that end a group load)
        }
        uint8_t q[4][3] = {};
        // (without a uv array - 16-byte records - a textured triangle samples its texture at uv = (0, 0), as colorAt_f does with
        // the default-initialised t of such a triangle and as the other tiers' color_at does; the descriptor cache is only
        // filled by the uv variants)
        if (m.n_textures) {
            // only the slots that end a textured group load (every load instruction of this kernel touches one cache line per
            // active lane, which is what bounds it), one aligned word per texel - two if its bytes straddle a word
            TexelRef tr[4];
            bool want[4];
            uint32_t w0[4] = {}, w1[4] = {};
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t id = mf[j].texid < m.n_textures ? mf[j].texid : 0u;
                want[j] = last[g0 + j] && mf[j].type == kTriTextured;
                tr[j] = TexelRef{nullptr, 0u};
                if (want[j]) {
                    const DevTexture tx = (kUv && id < kTexCache) ? s_tex[id] : m.textures[id];
                    tr[j] = texel_ref(texel_address(tx, u[g0 + j], v[g0 + j]));
                }
            }
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j)
                if (want[j]) w0[j] = tr[j].word[0];
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j)
                if (want[j] && tr[j].straddles()) w1[j] = tr[j].word[1];
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) texel_bytes(tr[j], w0[j], w1[j], q[j][0], q[j][1], q[j][2]);
        }
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            if (last[g0 + j]) {
                float cr, cg, cb;
                mat_color(mf[j], m.n_textures != 0u, q[j][0], q[j][1], q[j][2], cr, cg, cb);
                f.add(p.blend, hi[g0 + j], WCol{w[g0 + j], cr, cg, cb});
            }
        }
    }
    return f.finish(p.blend);
}

// `part` 0: the inline cells (hits in their bricks' slabs: they need nothing but k_scan_bricks' list, so this launch runs
// beside the counting sort); 1: the short cells of bricks without a slab (hits in the sorted array; none if every listed brick
// has a slab, the usual case).
template <uint32_t STRIDE>
__global__ __launch_bounds__(kBlock) void k_resolve(const Occ *__restrict__ occ, SortedView sorted_dyn, SortedView slabs_dyn,
                                                    const Counters *c, Materials m, uint4 *out, uint32_t part, Params p)
{
    if (part == 1u && c->n_dirty <= p.cap_slabs) return;
    __shared__ DevTexture s_tex[kTexCache];  // the first textures' descriptors (the colour lookup reads them per group)
    if (pass_overflowed(c, p)) return;
    if (STRIDE == 6) {
        for (uint32_t t = threadIdx.x; t < kTexCache && t < m.n_textures; t += kBlock) s_tex[t] = m.textures[t];
        __syncthreads();
    }
    const uint32_t n = c->n_vox < p.cap_vox ? c->n_vox : p.cap_vox;
    if (part == 0u) {
        // The inline cells with up to four hits - most cells of most meshes - in a software pipeline.  A cell is a chain of
        // dependent round trips (its entry of the list -> its records in the brick's slab -> its triangles' materials -> a texel),
        // and this launch, with two workgroups per CU so that the tiers beside it have room, was the stage's longest (the bench
        // mesh with BLEND: 334 of its 385 us, 37 cells per lane one after the other).  Now the entry of the cell after next and
        // the records of the next cell are requested before the current cell's arithmetic: one or two round trips a cell are
        // left exposed (materials, texel) instead of three or four.  Same loads, same arithmetic, same results.
        const SortedView slabs{slabs_dyn.base, STRIDE};
        const uint32_t step = gridDim.x * kBlock;
        auto wanted = [](const Occ &o) { return (o.count & kOccInline) != 0u && (o.count & ~kOccInline) <= kFourList; };
        auto fetch = [&](const Occ &o, SortedRec (&r)[kFourList]) {
            const uint32_t cnt = o.count & ~kOccInline;
            const size_t first = ((size_t) o.offset * kBrickCells + (o.cell_lo & (kBrickCells - 1u))) * kInlineHits;
#pragma unroll
            for (uint32_t k = 0; k < kFourList; ++k) {
                r[k] = SortedRec{0u, 0u, 0.f, 0.f, 0.f, 0u};
                if (k < cnt) r[k] = slabs.load(first + k);
            }
        };
        uint32_t i0 = blockIdx.x * kBlock + threadIdx.x, i1 = i0 + step;
        Occ o0{}, o1{};
        bool q0 = false;
        SortedRec r0[kFourList], r1[kFourList];
#pragma unroll
        for (uint32_t k = 0; k < kFourList; ++k) r0[k] = r1[k] = SortedRec{0u, 0u, 0.f, 0.f, 0.f, 0u};
        if (i0 < n) {
            o0 = occ[i0];
            q0 = wanted(o0);
            if (q0) fetch(o0, r0);
        }
        if (i1 < n) o1 = occ[i1];
        while (i0 < n) {
            const bool q1 = i1 < n && wanted(o1);
            if (q1) fetch(o1, r1);                      // the next cell's records ...
            const uint32_t i2 = i1 + step;
            Occ o2{};
            if (i2 < n) o2 = occ[i2];                   // ... and the entry of the one after it
            if (q0) {
                GroupFold f;
                const uint32_t argb = resolve_cell_in_registers<STRIDE, kFourList>([&](uint32_t k) { return r0[k]; }, o0.count & ~kOccInline, m, s_tex, p, f);
                emit_cell(o0, argb, f.cell_acc.w, f.cell_key, out, i0, c, p);
            }
            o0 = o1;
            q0 = q1;
#pragma unroll
            for (uint32_t k = 0; k < kFourList; ++k) r0[k] = r1[k];
            o1 = o2;
            i0 = i1;
            i1 = i2;
        }
        return;
    }
    // part 1: the short cells of bricks without a slab (their hits are in the sorted array; none as a rule)
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const Occ o = occ[i];
        const uint32_t count = o.count & ~kOccInline;
        if (count > kShortList) continue;  // filed for another tier by k_scan_bricks
        if ((o.count & kOccInline) != 0u) continue;  // an inline cell: part 0, or k_resolve_inline_list
        const SortedView from{sorted_dyn.base, STRIDE};  // compile-time stride: the preloads stay branch-free
        const size_t first = (size_t) o.offset;
        GroupFold f;
        const uint32_t argb = resolve_cell_in_registers<STRIDE, kShortList>([&](uint32_t k) { return from.load(first + k); }, count, m, s_tex, p, f);
        emit_cell(o, argb, f.cell_acc.w, f.cell_key, out, i, c, p);
    }
}

// Tier 1 for the inline cells with 5 .. 8 hits, from their list (k_scan_bricks files them): the eight-slot form with every
// lane in it.  Needs nothing but the slabs, so it runs beside the four-slot launch and the counting sort.
template <uint32_t STRIDE>
__global__ __launch_bounds__(kBlock) void k_resolve_inline_list(const uint32_t *__restrict__ list, const uint32_t *n_list, const Counters *c,
                                                                const Occ *__restrict__ occ, SortedView slabs_dyn, Materials m, uint4 *out,
                                                                uint32_t list_cap, Params p)
{
    const SortedView slabs{slabs_dyn.base, STRIDE};
    __shared__ DevTexture s_tex[kTexCache];
    if (pass_overflowed(c, p)) return;
    if (STRIDE == 6) {
        for (uint32_t t = threadIdx.x; t < kTexCache && t < m.n_textures; t += kBlock) s_tex[t] = m.textures[t];
        __syncthreads();
    }
    const uint32_t total = *n_list < list_cap ? *n_list : list_cap;
    // (the list entry of the cell after next and the next cell's entry of the cell list are requested ahead of this cell's
    // records: three dependent round trips a cell - records, materials, texel - instead of five; see k_resolve)
    const uint32_t step = gridDim.x * kBlock;
    uint32_t item = blockIdx.x * kBlock + threadIdx.x;
    uint32_t i0 = 0, i1 = 0;
    Occ o0{};
    if (item < total) {
        i0 = list[item];
        o0 = occ[i0];
    }
    if (item + step < total) i1 = list[item + step];
    for (; item < total; item += step) {
        Occ o1{};
        uint32_t i2 = 0;
        if (item + step < total) o1 = occ[i1];
        if (item + 2u * step < total) i2 = list[item + 2u * step];
        const size_t first = ((size_t) o0.offset * kBrickCells + (o0.cell_lo & (kBrickCells - 1u))) * kInlineHits;
        GroupFold f;
        const uint32_t argb = resolve_cell_in_registers<STRIDE, kShortList>([&](uint32_t k) { return slabs.load(first + k); }, o0.count & ~kOccInline, m,
                                                                            s_tex, p, f);
        emit_cell(o0, argb, f.cell_acc.w, f.cell_key, out, i0, c, p);
        o0 = o1;
        i0 = i1;
        i1 = i2;
    }
}

// Tier 1b: cells with 9..16 hits, from their list: the same, one lane per cell with sixteen records in its registers (round 2
// gave such a cell sixteen lanes that all ran its chain: 4.0 ms on configs[3]).
template <uint32_t STRIDE>
__global__ __launch_bounds__(kBlock) void k_resolve_list16(const uint32_t *__restrict__ list, const uint32_t *n_list, const Counters *c,
                                                           const Occ *__restrict__ occ, SortedView sorted_dyn, Materials m, uint4 *out,
                                                           uint32_t list_cap, Params p)
{
    const SortedView sorted{sorted_dyn.base, STRIDE};
    __shared__ DevTexture s_tex[kTexCache];
    if (pass_overflowed(c, p)) return;
    if (STRIDE == 6) {
        for (uint32_t t = threadIdx.x; t < kTexCache && t < m.n_textures; t += kBlock) s_tex[t] = m.textures[t];
        __syncthreads();
    }
    const uint32_t total = *n_list < list_cap ? *n_list : list_cap;
    // (no loads ahead here, unlike k_resolve_inline_list: five more registers put the uv variant above 128 - three wavefronts per SIMD)
    for (uint32_t item = blockIdx.x * kBlock + threadIdx.x; item < total; item += gridDim.x * kBlock) {
        const uint32_t i = list[item];
        const Occ o = occ[i];
        GroupFold f;
        const CellRecords recs = cell_records(sorted, o, p);
        const uint32_t argb = resolve_cell_in_registers<STRIDE, kLane16List>([&](uint32_t k) { return recs.load(k); }, o.count, m, s_tex, p, f);
        emit_cell(o, argb, f.cell_acc.w, f.cell_key, out, i, c, p);
    }
}

// Tier 2: cells with 9..64 hits, W = 16, 32 or 64 lanes per cell (64 / W cells per wavefront).  Every lane loads one
// record; the (key, position) pairs are bitonic-sorted across the W lanes with cross-lane moves only (no LDS, no
// barrier); the payload is gathered to its sorted lane; the groups are folded and coloured in parallel by the lanes at their
// first records, and the cell's chain over the groups runs on every lane of the cell on broadcast values.
// (the body: wavefront `wave` of `n_waves` that share the list)
template <uint32_t W>
__device__ __forceinline__ void resolve_wave_body(uint32_t wave, uint32_t n_waves, const uint32_t *__restrict__ list, const uint32_t *n_list,
                                                  const Counters *c, const Occ *__restrict__ occ, SortedView sorted, const Materials &m,
                                                  uint4 *out, uint32_t list_cap, const Params &p)
{
    constexpr uint32_t kPerWave = 64u / W;
    if (pass_overflowed(c, p)) return;
    const uint32_t total = *n_list < list_cap ? *n_list : list_cap;
    const uint32_t lane = threadIdx.x & 63u, sub = lane / W, sl = lane % W, base_lane = sub * W;
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
            const SortedRec r = cell_records(sorted, o, p).load(sl);
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
        // Lane sl now holds the cell's sl-th record in the reference's order.  The records of one (sub-voxel, triangle) group -
        // the leaves of one triangle in one (sub-)voxel - are neighbours: the lane at a group's first record folds the group
        // (insertWeighted<BLEND>, voxelization.cpp:466-468) and looks its colour up (colorAt_f), ALL GROUPS AT ONCE: the colour
        // lookup is a chain of dependent loads (type -> colour or texture index -> texture -> texel), and folding the cell
        // record by record would walk that chain once per group, one after the other (it did: 4.0 ms on configs[3]).
        const uint32_t prev_hi = __shfl_up(hi, 1u, 64);
        const bool start = sl < n && (sl == 0u || prev_hi != hi);
        WUv acc{w, u, v};
        {
            bool open = start;
            for (uint32_t d = 1; d < W; ++d) {  // (wavefront-uniform trip count: the longest group)
                const uint32_t nh = __shfl_down(hi, d, 64);
                const float nw = __shfl_down(w, d, 64), nu = __shfl_down(u, d, 64), nv = __shfl_down(v, d, 64);
                open = open && sl + d < n && nh == hi;
                if (!__any(open)) break;
                if (open) acc = wmix(WUv{nw, nu, nv}, acc);
            }
        }
        WCol fresh{0.f, 0.f, 0.f, 0.f};
        if (start) {
            float cr, cg, cb;
            color_at(m, hi & 0x1fffffffu, acc.u, acc.v, cr, cg, cb);
            fresh = WCol{acc.w, cr, cg, cb};
        }
        // ... then the chain over the groups, in order, on values that are already there (every lane of the cell runs it)
        GroupFold f;
        for (uint32_t t = 0; t < W; ++t) {
            if (!__any(t < n)) break;
            const int from = (int) (base_lane + t);
            const bool st = __shfl((int) start, from, 64) != 0;
            if (!__any(st)) continue;
            const uint32_t hh = __shfl(hi, from, 64);
            const WCol g{__shfl(fresh.w, from, 64), __shfl(fresh.r, from, 64), __shfl(fresh.g, from, 64), __shfl(fresh.b, from, 64)};
            if (st) f.add(p.blend, hh, g);
        }
        const uint32_t argb = f.finish(p.blend);
        if (n != 0 && sl == 0) emit_cell(o, argb, f.cell_acc.w, f.cell_key, out, i, c, p);
    }
}
template <uint32_t W>
__global__ __launch_bounds__(kBlock) void k_resolve_wave(const uint32_t *__restrict__ list, const uint32_t *n_list,
                                                         const Counters *c, const Occ *__restrict__ occ,
                                                         SortedView sorted, Materials m, uint4 *out, uint32_t list_cap,
                                                         Params p)
{
    resolve_wave_body<W>((blockIdx.x * kBlock + threadIdx.x) >> 6, gridDim.x * (kBlock / 64u), list, n_list, c, occ, sorted, m, out, list_cap, p);
}

// Between phases in which the lanes of ONE wavefront exchange values through LDS: the hardware runs a wavefront's LDS
// instructions in order, so this only has to keep the compiler from moving them across.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// BLEND over a crowded cell: the chain over its (sub-voxel, triangle) groups - CellFold's close_tri / close_sub sequence
// (moveUvBufferIntoVoxels, voxelization.cpp:513-526; insertWeighted / mix, :56-63, util.hpp:160-172; downscale,
// voxelization.hpp:82-85) - run by ONE wavefront on values in LDS.  The float mix is not associative, so the chain is
// sequential; what can be had is a short step.  On entry entry t < n holds, if it is the first record of its group, the
// group's {key, weight, r, g, b (bits)}; the other entries are skipped.
//   1. the group starts are compacted in place (positions 0 .. G - 1; the key becomes the sub-voxel index);
//   2. every sub-voxel's chain is independent of the others', and so are the three colour channels of one chain (they share
//      only the running weight, one add per step, which every lane keeps for itself): lane 3 s + c walks sub-voxel s, channel c -
//      a step is two LDS reads, w + W, w c + W C and ONE division per lane (the fold on one lane, values through v_readlane,
//      took three divisions and five v_readlane per step: ~45 instructions against ~18; a wavefront alone on its SIMD issues
//      one per ~4.6 cycles whatever their dependences);
//   3. the sub-voxels' results are combined in ascending order (at most eight steps) by lanes 0 .. 2.
// Same operations on the same operands in the same order as CellFold: same bits.  Returns the cell's ARGB (every lane).
__device__ __forceinline__ uint32_t blend_chain_lds(uint32_t *s_hi, float *s_w, float *s_r, float *s_g, uint32_t *s_b_bits, uint32_t n,
                                                    uint32_t *s_seg /*[16]*/, float *s_res /*[32]*/)
{
    const uint32_t lane = threadIdx.x & 63u;
    float *s_b = reinterpret_cast<float *>(s_b_bits);
    uint32_t n_groups = 0;  // wave-uniform
    {
        uint32_t prev_hi = 0;  // wave-uniform: the key of the entry before this chunk
        for (uint32_t base = 0; base < n; base += 64u) {
            const uint32_t t = base + lane;
            const bool in = t < n;
            const uint32_t hi_t = in ? s_hi[t] : 0u;
            const uint32_t before = __shfl_up(hi_t, 1u, 64);
            const bool start = in && (t == 0u || (lane == 0u ? prev_hi : before) != hi_t);
            const float w_t = start ? s_w[t] : 0.f, r_t = start ? s_r[t] : 0.f, g_t = start ? s_g[t] : 0.f, b_t = start ? s_b[t] : 0.f;
            prev_hi = (uint32_t) __builtin_amdgcn_readlane((int) hi_t, 63);
            const unsigned long long m = __ballot(start);
            const uint32_t pos = n_groups + __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
            // (every lane has read its entry; pos <= t, and the entries of later chunks lie above every pos of this one)
            if (start) {
                s_hi[pos] = hi_t >> 29;
                s_w[pos] = w_t;
                s_r[pos] = r_t;
                s_g[pos] = g_t;
                s_b[pos] = b_t;
            }
            n_groups += (uint32_t) __popcll(m);
        }
    }
    if (lane < 16u) s_seg[lane] = 0u;  // [s]: first group of sub-voxel s, [8 + s]: one past its last (both 0: none)
    wave_lds_sync();
    for (uint32_t g = lane; g < n_groups; g += 64u) {
        const uint32_t sub = s_hi[g];
        if (g == 0u || s_hi[g - 1u] != sub) s_seg[sub] = g;
        if (g + 1u == n_groups || s_hi[g + 1u] != sub) s_seg[8u + sub] = g + 1u;
    }
    wave_lds_sync();
    const uint32_t sub = lane / 3u, ch = lane - sub * 3u;
    const bool walker = lane < 24u;
    const uint32_t g0 = walker ? s_seg[sub] : 0u, g1 = walker ? s_seg[8u + sub] : 0u;
    const uint32_t len = g1 - g0;
    uint32_t longest = 0;
#pragma unroll
    for (uint32_t q = 0; q < 8u; ++q) {
        const uint32_t l = s_seg[8u + q] - s_seg[q];
        longest = l > longest ? l : longest;
    }
    const float *col = ch == 0u ? s_r : (ch == 1u ? s_g : s_b);
    float W = 0.f, C = 0.f;
    if (len) {
        W = s_w[g0];
        C = col[g0];
    }
    // (the next step's operands are requested before this step's arithmetic: the LDS round trip hides behind it)
    const uint32_t last = len ? g1 - 1u : 0u;
    uint32_t at = g0 + 1u < g1 ? g0 + 1u : last;
    float w_next = s_w[at], c_next = col[at];
    for (uint32_t k = 1; k < longest; ++k) {
        const float w = w_next, c = c_next;
        at = at + 1u < g1 ? at + 1u : last;
        w_next = s_w[at];
        c_next = col[at];
        if (k < len) {
            // wmix(fresh, acc), util.hpp:160-165: ws = l.w + r.w; (l.w l.c + r.w r.c) / ws
            const float ws = w + W;
            C = (w * c + W * C) / ws;
            W = ws;
        }
    }
    if (walker && ch == 0u) s_res[sub * 4u] = W;
    if (walker) s_res[sub * 4u + 1u + ch] = C;
    wave_lds_sync();
    // the cell's chain over its sub-voxels, ascending (close_sub): lane c < 3 runs channel c
    float cw = 0.f, cc = 0.f;
    bool have_cell = false;
    const uint32_t mych = lane < 3u ? lane : 0u;
#pragma unroll
    for (uint32_t q = 0; q < 8u; ++q) {
        if (s_seg[8u + q] == s_seg[q]) continue;  // (wave-uniform)
        const float sw = s_res[q * 4u], sc = s_res[q * 4u + 1u + mych];
        if (have_cell) {
            const float ws = sw + cw;
            cc = (sw * sc + cw * cc) / ws;
            cw = ws;
        }
        else {
            cw = sw;
            cc = sc;
            have_cell = true;
        }
    }
    const float fr = __shfl(cc, 0, 64), fg = __shfl(cc, 1, 64), fb = __shfl(cc, 2, 64);
    return pack_argb(fr, fg, fb);
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
__device__ __forceinline__ void resolve_sorted_body(const uint32_t *__restrict__ list, const uint32_t *n_list,
                                                    uint32_t *cursor, const Counters *c,
                                                    const Occ *__restrict__ occ, SortedView sorted, const Materials &m,
                                                    uint4 *out, uint32_t list_cap, const Params &p)
{
    if (pass_overflowed(c, p)) return;
    __shared__ uint64_t s_key[CAP];
    __shared__ uint32_t s_idx[CAP];
    __shared__ uint32_t s_hi[CAP];
    __shared__ float s_w[CAP], s_u[CAP], s_v[CAP];
    __shared__ uint32_t s_item;
    __shared__ uint32_t s_seg[16];
    __shared__ float s_res[32];
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
                const SortedRec r = cell_records(sorted, o, p).load(t);
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
            const SortedRec r = cell_records(sorted, o, p).load(s_idx[t]);
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
            if (threadIdx.x < 64u) {
                const uint32_t argb = blend_chain_lds(s_hi, s_w, s_u, s_v, s_idx, n, s_seg, s_res);
                if (threadIdx.x == 0) out[i] = cell_record(o, argb, p);
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
                emit_cell(o, pack_argb(cr, cg, cb), acc.w, s_hi[t], out, i, c, p);
            }
        }
    }
}
template <uint32_t THREADS, uint32_t CAP>
__global__ __launch_bounds__(THREADS) void k_resolve_sorted(const uint32_t *__restrict__ list, const uint32_t *n_list,
                                                            uint32_t *cursor, const Counters *c,
                                                            const Occ *__restrict__ occ, SortedView sorted, Materials m,
                                                            uint4 *out, uint32_t list_cap, Params p)
{
    resolve_sorted_body<THREADS, CAP>(list, n_list, cursor, c, occ, sorted, m, out, list_cap, p);
}

// The cooperative tiers for 17 .. 256 hits in ONE launch of one-wavefront workgroups: the first `g_mid` take cells of 65 .. 256
// hits from their cursor (k_resolve_sorted<64, 256>'s body - the longest cells, so they start first), the next `g_w64` are
// k_resolve_wave<64>'s wavefronts (33 .. 64 hits), the rest k_resolve_wave<32>'s (17 .. 32).  Each of the three is a handful of
// latency chains over few cells; as launches of their own on one stream they ran one after the other (configs[1]: 5 + 22 + 50 us
// for 0.2 MB, the bench mesh with BLEND 47 + 33 + 92 us), together they take as long as the slowest.
struct TierLists {
    const uint32_t *mid, *w64, *w32;
    const uint32_t *n_mid, *n_w64, *n_w32;
    uint32_t *cursor_mid;
};
__global__ __launch_bounds__(64) void k_resolve_tiers(TierLists lists, uint32_t g_mid, uint32_t g_w64, const Counters *c, const Occ *__restrict__ occ,
                                                      SortedView sorted, Materials m, uint4 *out, uint32_t list_cap, Params p)
{
    const uint32_t b = blockIdx.x;
    if (b < g_mid) resolve_sorted_body<64, kMidList>(lists.mid, lists.n_mid, lists.cursor_mid, c, occ, sorted, m, out, list_cap, p);
    else if (b < g_mid + g_w64) resolve_wave_body<64>(b - g_mid, g_w64, lists.w64, lists.n_w64, c, occ, sorted, m, out, list_cap, p);
    else resolve_wave_body<32>(b - g_mid - g_w64, gridDim.x - g_mid - g_w64, lists.w32, lists.n_w32, c, occ, sorted, m, out, list_cap, p);
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
                const SortedRec r = cell_records(sorted, o, p).load(t);
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
                    const SortedRec r = cell_records(sorted, o, p).load(s_idx[base + t]);
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
                    SortedRec r = cell_records(sorted, o, p).load(s_idx[t]);
                    WUv acc{r.w, r.u, r.v};
                    for (uint32_t j = t + 1; j < n && (uint32_t) (s_key[j] >> 32) == hi; ++j) {
                        r = cell_records(sorted, o, p).load(s_idx[j]);
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
                SortedRec r = cell_records(sorted, o, p).load(s_idx[t]);
                WUv acc{r.w, r.u, r.v};
                for (uint32_t j = t + 1; j < n && (uint32_t) (s_key[j] >> 32) == hi; ++j) {
                    r = cell_records(sorted, o, p).load(s_idx[j]);
                    acc = wmix(WUv{r.w, r.u, r.v}, acc);
                }
                float cr, cg, cb;
                color_at(m, hi & 0x1fffffffu, acc.u, acc.v, cr, cg, cb);
                emit_cell(o, pack_argb(cr, cg, cb), acc.w, hi, out, i, c, p);
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
                const SortedRec r = cell_records(sorted, o, p).load(t);
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
                    const SortedRec r = cell_records(sorted, o, p).load(idx[t]);
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
                    SortedRec r = cell_records(sorted, o, p).load(idx[t]);
                    WUv acc{r.w, r.u, r.v};
                    for (uint32_t j = t + 1; j < n && (uint32_t) (key[j] >> 32) == hi; ++j) {
                        r = cell_records(sorted, o, p).load(idx[j]);
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
                SortedRec r = cell_records(sorted, o, p).load(idx[t]);
                WUv acc{r.w, r.u, r.v};
                for (uint32_t j = t + 1; j < n && (uint32_t) (key[j] >> 32) == hi; ++j) {
                    r = cell_records(sorted, o, p).load(idx[j]);
                    acc = wmix(WUv{r.w, r.u, r.v}, acc);
                }
                float cr, cg, cb;
                color_at(m, hi & 0x1fffffffu, acc.u, acc.v, cr, cg, cb);
                emit_cell(o, pack_argb(cr, cg, cb), acc.w, hi, out, i, c, p);
            }
        }
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
        if (p.maxgrid[cell] == key) {
            // the winner's colour (moveUvBufferIntoVoxels, voxelization.cpp:513-526): looked up here, once per voxel, not per hit
            float cr, cg, cb;
            color_at(p.mat, r.keyhi & 0x1fffffffu, r.u, r.v, cr, cg, cb);
            p.maxgrid[cell] = kPickTag | pack_argb(cr, cg, cb);
        }
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
// 48 KiB of staging: three workgroups per CU keep enough brick loads (2 KiB per wavefront) in flight (64 KiB / two workgroups: 0.15 ms
// on the bench mesh, this: 0.12; 32 KiB / four: 0.16).  A staged record is 12 bytes (x | y << 16, z, argb) and becomes the 16-byte
// (x, y, z, argb) when it is written out: a flush - one serialising atomic on Counters::n_out each - then takes 2 048 records
// or more instead of 1 024 (see k_emit_occ).
#ifndef O2V_EMIT_FLUSH
#define O2V_EMIT_FLUSH 2048
#endif
constexpr uint32_t kEmitFlushAt = O2V_EMIT_FLUSH;
constexpr uint32_t kEmitCap = kEmitFlushAt + kEmitBricksPerRound * kBrickCells;

__global__ __launch_bounds__(kBlock) void k_emit_max(const uint32_t *__restrict__ dirty_list, Counters *c, Materials m, uint4 *out,
                                                     Params p)
{
    __shared__ uint2 s_rec[kEmitCap];
    __shared__ uint32_t s_argb[kEmitCap];
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
            if (base + i < p.cap_vox) out[base + i] = make_uint4((s_rec[i].x & 0xffffu) + p.xo0, (s_rec[i].x >> 16) + p.yo0, s_rec[i].y + p.zo0, s_argb[i]);
        __syncthreads();
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
    };
    // The list entries of a round are requested two rounds ahead and its bricks' cells one round ahead (both loads depend on the
    // one before: three round trips per round otherwise, with the colour lookup).
    auto list_entry = [&](uint32_t r, uint32_t k) -> uint32_t {
        const uint32_t item = r * kEmitBricksPerRound + (wave * kEmitBricksPerWave + k) * kBricksPerLoad + lane / kLanesPerBrick;
        return (r < n_rounds && item < n_dirty) ? dirty_list[item] : 0xffffffffu;
    };
    auto cells_of = [&](uint32_t b, ulonglong2 &lo_, ulonglong2 &hi_) {
        lo_ = hi_ = make_ulonglong2(0, 0);
        if (b != 0xffffffffu) {
            const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(p.maxgrid + (uint64_t) b * kBrickCells) + (lane % kLanesPerBrick) * 2u;
            lo_ = q[0];
            hi_ = q[1];
        }
    };
    uint32_t next_brick[kEmitBricksPerWave], after_brick[kEmitBricksPerWave];
    ulonglong2 next_lo[kEmitBricksPerWave], next_hi[kEmitBricksPerWave];
#pragma unroll
    for (uint32_t k = 0; k < kEmitBricksPerWave; ++k) {
        next_brick[k] = list_entry(blockIdx.x, k);
        after_brick[k] = list_entry(blockIdx.x + gridDim.x, k);
    }
#pragma unroll
    for (uint32_t k = 0; k < kEmitBricksPerWave; ++k) cells_of(next_brick[k], next_lo[k], next_hi[k]);
    for (uint32_t r = blockIdx.x; r < n_rounds; r += gridDim.x) {
        uint32_t brick[kEmitBricksPerWave];
        ulonglong2 lo[kEmitBricksPerWave], hi[kEmitBricksPerWave];
#pragma unroll
        for (uint32_t k = 0; k < kEmitBricksPerWave; ++k) {
            brick[k] = next_brick[k];
            lo[k] = next_lo[k];
            hi[k] = next_hi[k];
            next_brick[k] = after_brick[k];
            cells_of(after_brick[k], next_lo[k], next_hi[k]);
            after_brick[k] = list_entry(r + 2u * gridDim.x, k);
        }
#pragma unroll
        for (uint32_t k = 0; k < kEmitBricksPerWave; ++k) {
            const unsigned long long v4[4] = {lo[k].x, lo[k].y, hi[k].x, hi[k].y};
            if (v4[0] | v4[1] | v4[2] | v4[3]) {
                uint32_t x0, y0, z0;
                brick_origin_rel(brick[k], p, x0, y0, z0);   // (staged relative to the grid's origin, which the flush adds)
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
                        s_rec[slot] = make_uint2((x0 + (local & (kBrickX - 1u))) | ((y0 + ((local >> kBrickXs) & (kBrickY - 1u))) << 16),
                                                 z0 + (local >> (kBrickXs + kBrickYs)));  // (a pass' box is at most 65 535 cells wide)
                        s_argb[slot] = argb;
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

// Occupancy-only mode (Params::occupancy_only): the dirty bricks of the one-byte-per-cell grid (64 bytes each: one 4-byte load
// per lane, four bricks per wavefront load) become white (x, y, z, argb) records and are zeroed again.
constexpr uint32_t kOccBricksPerWave = 2;  // (the staging buffer then takes 48 KiB: three workgroups per CU)
constexpr uint32_t kOccBricksPerRound = (kBlock / 64) * kOccBricksPerWave * kBricksPerLoad;
// A flush reserves its records with one atomic on Counters::n_out, and those serialise (~5 ns each: with 1 024 records per
// flush they were a third of this kernel - 512: 0.088 ms, 1 024: 0.062).  The staged record is therefore 8 bytes (x | y << 16,
// z; the colour is white) and becomes the 16-byte (x, y, z, argb) when it is written out: the same 48 KiB hold 6 144 of them, a
// flush takes 4 096 or more.
#ifndef O2V_OCC_FLUSH
#define O2V_OCC_FLUSH 4096
#endif
constexpr uint32_t kOccFlushAt = O2V_OCC_FLUSH;
constexpr uint32_t kOccCap = kOccFlushAt + kOccBricksPerRound * kBrickCells;
__global__ __launch_bounds__(kBlock) void k_emit_occ(const uint32_t *__restrict__ dirty_list, Counters *c, uint4 *out, Params p)
{
    static_assert(kLanesPerBrick * 4u == kBrickCells, "one 4-byte load per lane covers four cells");
    __shared__ uint2 s_rec[kOccCap];
    __shared__ uint32_t s_n, s_base;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint32_t white = pack_argb(1.f, 1.f, 1.f);  // colorAt_f of a material-less triangle, triangle.hpp:181-194
    const uint32_t n_dirty = c->n_dirty_max < p.cap_dirty ? c->n_dirty_max : p.cap_dirty;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n_rounds = (n_dirty + kOccBricksPerRound - 1) / kOccBricksPerRound;
    auto flush = [&](uint32_t n) {
        if (threadIdx.x == 0) s_base = atomicAdd(&c->n_out, n);
        __syncthreads();
        const uint32_t base = s_base;
        for (uint32_t i = threadIdx.x; i < n; i += kBlock)
            if (base + i < p.cap_vox) out[base + i] = make_uint4((s_rec[i].x & 0xffffu) + p.xo0, (s_rec[i].x >> 16) + p.yo0, s_rec[i].y + p.zo0, white);
        __syncthreads();
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
    };
    uint32_t *grid4 = reinterpret_cast<uint32_t *>(p.occgrid);
    // the list entries of a round are requested one round ahead (the brick loads depend on them: two round trips per round
    // otherwise)
    auto list_entry = [&](uint32_t r, uint32_t k) -> uint32_t {
        const uint32_t item = r * kOccBricksPerRound + (wave * kOccBricksPerWave + k) * kBricksPerLoad + lane / kLanesPerBrick;
        return (r < n_rounds && item < n_dirty) ? dirty_list[item] : 0xffffffffu;
    };
    // (... and the bricks' cells one round ahead: a round then waits for neither)
    auto cells_of = [&](uint32_t b) -> uint32_t { return b != 0xffffffffu ? grid4[(uint64_t) b * kLanesPerBrick + lane % kLanesPerBrick] : 0u; };
    uint32_t next_brick[kOccBricksPerWave], next_cells[kOccBricksPerWave], after_brick[kOccBricksPerWave];
#pragma unroll
    for (uint32_t k = 0; k < kOccBricksPerWave; ++k) {
        next_brick[k] = list_entry(blockIdx.x, k);
        after_brick[k] = list_entry(blockIdx.x + gridDim.x, k);
    }
#pragma unroll
    for (uint32_t k = 0; k < kOccBricksPerWave; ++k) next_cells[k] = cells_of(next_brick[k]);
    for (uint32_t r = blockIdx.x; r < n_rounds; r += gridDim.x) {
        uint32_t brick[kOccBricksPerWave], cells4[kOccBricksPerWave];
#pragma unroll
        for (uint32_t k = 0; k < kOccBricksPerWave; ++k) {
            brick[k] = next_brick[k];
            cells4[k] = next_cells[k];
            next_brick[k] = after_brick[k];
            next_cells[k] = cells_of(after_brick[k]);
            after_brick[k] = list_entry(r + 2u * gridDim.x, k);
        }
#pragma unroll
        for (uint32_t k = 0; k < kOccBricksPerWave; ++k) {
            // (wavefront-uniform loop over the four cells of a lane: the staging slots are reserved with one LDS atomic per
            // wavefront and cell position, not one per voxel)
            uint32_t x0, y0, z0;
            brick_origin_rel(brick[k] == 0xffffffffu ? 0u : brick[k], p, x0, y0, z0);   // (relative to the grid's origin, which the flush adds)
#pragma unroll
            for (uint32_t e = 0; e < 4; ++e) {
                const bool set = ((cells4[k] >> (8u * e)) & 0xffu) != 0u;
                const unsigned long long m = __ballot(set);
                if (m) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&s_n, (uint32_t) __popcll(m));
                    base = __shfl(base, 0, 64);
                    if (set) {
                        const uint32_t local = (lane % kLanesPerBrick) * 4u + e;
                        const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
                        s_rec[slot] = make_uint2((x0 + (local & (kBrickX - 1u))) | ((y0 + ((local >> kBrickXs) & (kBrickY - 1u))) << 16),
                                                 z0 + (local >> (kBrickXs + kBrickYs)));  // (a pass' box is at most 65 535 cells wide)
                    }
                }
            }
            if (cells4[k]) grid4[(uint64_t) brick[k] * kLanesPerBrick + lane % kLanesPerBrick] = 0u;  // clean for the next run
        }
        __syncthreads();
        const uint32_t n = s_n;
        __syncthreads();  // (every thread has read n before anyone adds to s_n again: the decision below must be uniform)
        if (n >= kOccFlushAt) flush(n);
    }
    __syncthreads();
    const uint32_t n = s_n;
    if (n) flush(n);
}
