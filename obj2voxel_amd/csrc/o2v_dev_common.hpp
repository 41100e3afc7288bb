// o2v_dev_common.hpp -- records, constants and small helpers shared by the kernels.
//
// Part of the device code of o2v_device.hip, which includes this file inside its anonymous namespace (one
// translation unit: the stages share records and launch parameters).  Not a stand-alone header.

// ---- device-side records ----------------------------------------------------------------------------------

constexpr uint32_t kTileSize = 256;       // candidate voxels per work tile
constexpr uint32_t kTilesPerBatch = 256;  // tiles a workgroup stages at once (at most: one thread per tile)
constexpr uint32_t kMinTilesPerBatch = 4;  // ... and at least (one tile per wavefront in phase 1)
constexpr uint32_t kBlock = 256;          // threads per workgroup (4 wavefronts)
constexpr uint32_t kMaxRounds = 16;       // subdivision depth limit (order key holds 15 levels)
constexpr uint32_t kHitChunk = 256;       // hit-pool slots a wavefront reserves per global atomic
constexpr uint32_t kInlineTiles = 4;      // leaves with more tiles are expanded by k_expand_big

struct __attribute__((aligned(16))) Leaf {  // 96 B
    float v[9];        // sample-space vertices
    float n[3];        // normalize(normal): plane of the distance cull
    float t[6];        // uv per vertex
    uint32_t tri;      // input triangle index
    uint32_t pathkey;  // order key of this leaf among the leaves of `tri` (0 = unsplit triangle)
    uint32_t bmin_xy;  // clamped AABB min, relative to the grid's origin in sample space (Params::so): x | y << 16
    uint32_t bmin_z_dx;  // z | dx << 16
    uint32_t dy_dz;      // dy | dz << 16
    float area;          // area of the whole input triangle (voxelization.cpp:416)
};
static_assert(sizeof(Leaf) == 96, "Leaf layout");

struct __attribute__((aligned(16))) Node {  // 80 B: a sub-triangle that still has to be subdivided
    float v[9];
    float t[6];
    uint32_t tri;
    uint32_t pathkey;
    uint32_t depth;
    float area;
    uint32_t pad;
};
static_assert(sizeof(Node) == 80, "Node layout");

struct Tile {
    uint32_t leaf;
    uint32_t start;  // first candidate index inside the leaf's clamped AABB
};

struct BigLeaf {
    uint32_t leaf, first_tile, ntiles, pad;
};

struct __attribute__((aligned(16))) HitRec {  // 32 B: one (leaf, voxel) hit as emitted by k_voxelize
    uint32_t brick;       // brick of the cell; kHoleBrick marks a pool slot that holds no hit
    uint32_t local_rank;  // cell inside the brick << 24 | rank of this hit among the hits of its cell
    uint32_t keyhi;       // sub-voxel << 29 | triangle index
    uint32_t keylo;       // leaf order key
    float w, u, v;        // WeightedUv of this (leaf, voxel) pair (voxelization.cpp:414-423)
    uint32_t pad;
};
constexpr uint32_t kHoleBrick = 0xffffffffu;
struct PickRec {  // 24 B: {cell lo, cell hi, keyhi, weight bits, argb, 0}
    uint32_t w[6];
};
constexpr uint32_t kPickRecord = 1u;  // HitRec::pad of a direct hit's {cell, key, weight, uv} record (textured MAX): k_pick colours the winner
constexpr unsigned long long kPickTag = 1ull << 63;  // a max-grid cell that already holds its final argb (low word)
constexpr uint32_t kMaxRank = 1u << 24;

struct __attribute__((aligned(8))) SortedRec {  // 24 B: the same hit, placed contiguously with its cell's other hits
    uint32_t keyhi, keylo;
    float w, u, v;
    uint32_t pad;
};

// The sorted array is read through a view: 6 dwords per record in general, 4 (keyhi, keylo, w, pad: one 16-byte access)
// when the mesh has no textured triangle, because then u and v are never used and the scatter's cost scales with the
// bytes it writes.
constexpr uint32_t kOccInline = 0x80000000u;  // Occ::count: the cell's hits are in its brick's slab, not in the sorted array
constexpr uint32_t kInlineHits = 8;  // hits per cell kept in the brick's slab (= kShortList: what k_resolve's lanes fold from registers)
struct SortedView {
    const uint32_t *base;
    uint32_t stride;  // dwords per record: 6 or 4
    __device__ __forceinline__ SortedRec load(size_t i) const
    {
        if (stride == 4u) {
            const uint4 q = reinterpret_cast<const uint4 *>(base)[i];
            return SortedRec{q.x, q.y, __uint_as_float(q.z), 0.f, 0.f, 0u};
        }
        return reinterpret_cast<const SortedRec *>(base)[i];
    }
};

struct __attribute__((aligned(16))) Occ {  // 16 B: one occupied cell
    uint32_t cell_lo;           // brick * kBrickCells + cell in brick, low 32 bits
    uint32_t cell_hi;           // bits 0..4: its bits 32..36; bits 5..31: the number of the brick's slab (>= cap_slabs: none)
    uint32_t offset;            // an inline cell: the number of its brick's slab; else its first record in the sorted array
    uint32_t count;             // number of hits; | kOccInline: the hits (at most kInlineHits) are in the brick's slab
    __device__ __host__ __forceinline__ uint64_t cell() const { return ((uint64_t) (cell_hi & 31u) << 32) | cell_lo; }
    __device__ __host__ __forceinline__ uint32_t slab() const { return cell_hi >> 5; }
};

struct DevTexture {
    const uint8_t *pixels;
    uint32_t width, height, channels, wrap;
};

struct Materials {
    const uint32_t *types;   // nullable: all MATERIALLESS
    const float *colors;     // nullable
    const int32_t *texids;   // nullable: all 0
    const DevTexture *textures;
    uint32_t n_textures;
};

struct Counters {
    uint32_t n_leaves, n_tiles, n_big, n_hits_reserved;
    uint32_t n_vox, batch_cursor, err_flags, n_lane16;
    uint32_t n_mid, n_long, n_huge, scratch_used;
    uint32_t n_dirty, n_sorted, n_bigl, cursor_big;
    uint32_t cursor_mid, cursor_long, cursor_huge, n_lane;
    uint32_t n_nodes[kMaxRounds + 1];
    uint32_t n_w64, n_dirty_max, n_out, n_lane8;  // n_lane8: inline cells with 5 .. 8 hits (their own launch of tier 1)
    unsigned long long n_candidates, n_hits;
    uint32_t bounds_enc[6];
    uint32_t n_root_leaves, pad2;  // root triangles that became leaves as they are (the others are in n_nodes[0]);
                                   // pad2: k_tri_extent's result at upload time, pick records of the replay tiers in a pass
    unsigned long long n_direct;
    float xform[12];
    unsigned long long n_certain;   // occupancy-only mode: hits established without a voxel job (certain_prepare)
    unsigned long long n_jobs;      // candidate voxels that passed phase 1 of k_voxelize (= voxel jobs of phase 2)
    unsigned long long n_jobs_skipped;  // occupancy-only mode: jobs dropped before phase 2 because their voxel was marked already
    uint32_t n_listed_hits, n_listed_blocks;  // k_scan_bricks: the counters of the listed bricks' cells added up (modulo 2^32): must equal
                                        // the hits k_voxelize counted into the grid (n_hits - n_direct), see o2v_hip_voxelize
                                        // n_listed_blocks: k_list_blocks' count of the blocks of 256 triangles that meet the slab (0xffffffff: no list)
    unsigned long long n_candidates_sq; // sum over the leaves of (candidates of the leaf)^2: with n_candidates and the number of leaves, how
                                        // unequal the leaves are (k_voxelize sizes the batches of its last quarter by it)
    uint32_t n_need_blocks, pad4;       // k_count_roots: blocks of 256 triangles with a triangle that k_expand_roots has to handle
    unsigned long long n_bypass;        // Params::root_bypass: root triangles that k_voxelize_occ stages itself (no Leaf, no Tile)
    unsigned long long dbg[16];  // event counts of an instrumented build (-DO2V_INSTRUMENT, tools/instrument.sh); else zero
    uint32_t ext_hist[256];      // k_tri_extent (at upload time only): triangles by the binary exponent of their extent
};

enum : uint32_t {
    kErrLeafTooLarge = 1u,
    kErrDepth = 2u,
    kErrRank = 4u,
    kErrDirtyList = 8u,  // more dirty bricks than the dirty list holds (Params::cap_dirty)
    kErrCounterWrap = 16u,  // 2^32 or more leaves (or tiles) in one pass: the counters of k_expand_* wrapped
    kErrSoloRoots = 32u,    // Params::solo_roots: a root triangle is not a leaf of one tile after all (the pass is repeated with k_expand_roots)
};

struct Params {
    uint64_t n_tris;
    uint32_t S;            // sample resolution = resolution * supersampling
    uint32_t G;            // output resolution
    uint32_t NBx, NBy;     // bricks per grid row / per z layer (brick = 4 x 4 x 4 cells, stored contiguously)
    uint32_t ss_shift;     // 0, or 1 for 2x supersampling
    uint32_t zs0, zs1;     // slab in sample space
    uint32_t zo0;          // slab begin in output space
    // The dense grids cover the mesh's voxel bounding box, not the G^3 cube ("crop", o2v_hip_voxelize): (xo0, yo0, zo0) is the
    // grid's origin in output space (xo0, yo0 multiples of the brick edge), NBx / NBy its extent in bricks, and [cs_lo, cs_hi)
    // the same box in sample space (z: cut to the slab), to which every leaf's box is clamped (plan_leaf) - no leaf reaches
    // beyond it (the crop encloses every triangle), the clamp only keeps a wrong crop from writing outside the allocation.
    uint32_t xo0, yo0;
    uint32_t cs_lo[3], cs_hi[3];
    // The grid's origin in sample space, (xo0, yo0, zo0) << ss_shift.  Coordinates that travel in 16-bit fields - a leaf's box
    // (Leaf::bmin_*), the job records of k_voxelize, the staged records of the emission kernels - are RELATIVE to it, so what
    // is limited to 65 535 is a pass' box, not the resolution (o2v_hip_voxelize refuses a box that is wider; obj2voxel_voxelize
    // cuts such a grid into x / y tiles).  Arithmetic on positions (the voxel planes of the clip) uses origin + relative.
    uint32_t so[3];
    uint32_t blend;
    uint32_t cap_leaves, cap_tiles, cap_big, cap_nodes, cap_hits, cap_vox;
    uint32_t n_bricks;     // bricks of this slab
    uint32_t cap_dirty;    // entries of each dirty-brick list
    uint32_t bounds_known;
    float bounds[6];
    int32_t unit[9];
    uint32_t has_uv;
    // Occupancy-only mode: every triangle is MATERIALLESS, so every voxel's colour is white whatever the weights are (MAX
    // picks one white, BLEND computes (w1 * 1 + w2 * 1) / (w1 + w2) = s / s = 1 exactly) and only the set of voxels with a
    // non-zero weight matters: a voxel job ends at its first surviving piece and every hit takes the direct path.
    uint32_t occupancy_only;
    // Occupancy-only mode: a root triangle that is one leaf of one tile (the usual triangle of a tessellated surface) gets no
    // Leaf and no Tile record; k_voxelize_occ makes its leaf from the vertex array (k_expand_roots only counts it).
    uint32_t root_bypass;
    uint32_t solo_roots;   // root_bypass and no k_expand_roots at all: k_voxelize_occ also counts the root leaves (see o2v_hip_voxelize)
    float plan_leaf_cost;  // slab planning (k_zhist): what a leaf costs beside its hits, in hit equivalents
    uint32_t exact_clip;   // O2V_HIP_FLAG_EXACT_CLIP: no work-removal shortcuts in k_voxelize (every leaf is treated as not `small`)
    // Direct MAX path (MAX strategy, no textured triangle; section 4 of DESIGN.md): one 64-bit cell per output voxel that
    // holds max over {weight bits << 32 | ~(sub-voxel << 29 | triangle)}, and its own dirty-brick map.
    uint32_t direct_max;
    // ... with textured triangles ("pick" variant): k_voxelize also samples the colour of a direct hit and keeps a
    // {cell, key, argb} record; k_pick later gives every cell whose winner it was that colour.
    uint32_t pick_max;
    Materials mat;
    // Inline hit slabs (general route): every brick a leaf's clamped box touches is listed before
    // k_voxelize (k_mark_bricks -> k_scan_flags) and owns a slab of kInlineHits x 64 hit records; brick_slab[brick] is its
    // number.  A cell's first kInlineHits hits (by rank) go straight into its slab - record (slab * 64 + cell in brick) *
    // kInlineHits + rank: a cell's hits lie side by side - and never see the pool or the scatter.
    const uint32_t *brick_slab;
    uint32_t *slabs;       // cap_slabs x kInlineHits x 64 records of slab_stride dwords
    uint32_t cap_slabs, slab_stride;
    uint32_t *pick_extra;  // the same records for the winners of cells resolved by replay: 6 words each, cap_vox of them
    unsigned long long *maxgrid;
    uint8_t *occgrid;   // occupancy-only mode: the same buffer as one byte per cell (non-zero = the voxel is hit)
    uint8_t *dirty_max;
};

// ---- small device helpers ---------------------------------------------------------------------------------

// A pass that ran out of hit-pool or cell-list space has cell offsets that point past the sorted records: the
// resolve kernels skip such a pass (the host grows the buffers and runs it again).
// A pass whose subdivision ran out of leaf / tile / queue space is discarded by the host as well; k_voxelize skips it
// (a tile may name a leaf that was never written).
__device__ __forceinline__ bool expand_overflowed(const Counters *c, const Params &p)
{
    bool over = c->n_leaves > p.cap_leaves || c->n_tiles > p.cap_tiles || c->n_big > p.cap_big;
    for (uint32_t r = 0; r <= kMaxRounds; ++r) over |= c->n_nodes[r] > p.cap_nodes;
    return over;
}
// The direct MAX path pays off when most triangles are voxelized whole (their hits skip the pool / sort / replay);
// for a mesh whose triangles are mostly subdivided it would only add a pass.  Decided on the device from K1's counters,
// identically by every kernel of the pass (and by the host afterwards).
__device__ __forceinline__ bool direct_active(const Counters *c, const Params &p)
{
    return p.direct_max && (p.occupancy_only || c->n_nodes[0] <= c->n_root_leaves);
}
// A pass pools no hits at all if every triangle is voxelized whole and takes the direct MAX path (or the mesh has no
// materials): then no brick needs a hit slab and the stages of the general route have nothing to do.
__device__ __forceinline__ bool pools_no_hits(const Counters *c, const Params &p, uint32_t force_general)
{
    return p.occupancy_only || (direct_active(c, p) && c->n_nodes[0] == 0u && !force_general);
}
__device__ __forceinline__ bool pass_overflowed(const Counters *c, const Params &p)
{
    return c->n_hits_reserved > p.cap_hits || c->n_sorted > p.cap_hits || c->n_vox > p.cap_vox;
}

__device__ __forceinline__ uint32_t f2ord(float f)
{
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o)
{
    uint32_t b = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(b);
}

// Dense grid layout: bricks of 4 x 4 x 4 cells, each brick 64 consecutive u32 (256 B; 512 B in the 64-bit grid), bricks
// ordered x fastest, one dirty byte per brick.  A surface of area A meets about A/2 * (1/(sy sz) + 1/(sx sz) + 1/(sx sy))
// bricks of extents (sx, sy, sz), so small cubic bricks mean the fewest cells of dirty bricks for the scan / emission
// passes to read: measured on the bench mesh, k_emit_max 0.121 ms with 16 x 4 x 4 bricks (round 1), 0.103 with 4 x 8 x 8,
// 0.092 with 4 x 4 x 8, 0.078 with 4 x 4 x 4 (the flag map grows to 1/64 byte per cell: k_scan_flags 0.013 -> 0.020 ms).
#ifndef O2V_BRICK_XS
#define O2V_BRICK_XS 2
#define O2V_BRICK_YS 2
#define O2V_BRICK_ZS 2
#endif
constexpr uint32_t kBrickXs = O2V_BRICK_XS, kBrickYs = O2V_BRICK_YS, kBrickZs = O2V_BRICK_ZS;  // log2 of the brick's extents
constexpr uint32_t kBrickShift = kBrickXs + kBrickYs + kBrickZs;
constexpr uint32_t kBrickX = 1u << kBrickXs, kBrickY = 1u << kBrickYs, kBrickZ = 1u << kBrickZs, kBrickCells = 1u << kBrickShift;

// Where the hits of a cell with more than kInlineHits hits are: the first kInlineHits (ranks 0 .. 7, written by k_voxelize)
// in its brick's slab, the others (placed by k_scatter) in the cell's range of the sorted array; a cell of a brick without a
// slab has all of them in the sorted array.  The cooperative resolve tiers read a cell's k-th record through this.
struct CellRecords {
    const uint32_t *sorted, *slabs;
    uint32_t stride;      // dwords per record: 6 or 4
    size_t slab_first;    // the cell's first record in the slab array
    uint32_t n_slab;      // 0 or kInlineHits
    uint32_t offset;      // the cell's first record in the sorted array
    __device__ __forceinline__ SortedRec load(uint32_t k) const
    {
        const uint32_t *at = k < n_slab ? slabs + (slab_first + k) * stride : sorted + ((size_t) offset + (k - n_slab)) * stride;
        if (stride == 4u) {
            const uint4 q = *reinterpret_cast<const uint4 *>(at);
            return SortedRec{q.x, q.y, __uint_as_float(q.z), 0.f, 0.f, 0u};
        }
        return *reinterpret_cast<const SortedRec *>(at);
    }
};
__device__ __forceinline__ CellRecords cell_records(const SortedView &sorted, const Occ &o, const Params &p)
{
    const uint32_t slab = o.slab();
    return CellRecords{sorted.base, p.slabs, sorted.stride, ((size_t) slab * kBrickCells + (o.cell_lo & (kBrickCells - 1u))) * kInlineHits,
                       slab < p.cap_slabs ? kInlineHits : 0u, o.offset};
}

// The kernels that read whole bricks give every lane four consecutive cells (one 16-byte load in the 32-bit grid, two in
// the 64-bit one), so one wavefront load covers 256 cells = kBricksPerLoad bricks.
constexpr uint32_t kLanesPerBrick = kBrickCells / 4u, kBricksPerLoad = 64u / kLanesPerBrick;
// Entries of a dirty-brick list: one per brick of the slab, but no more than this (0.5 GiB; a 4096^3 grid on one GPU has
// 2^30 bricks, of which a surface dirties a few million - more is reported as a limit, see kErrDirtyList).
constexpr uint64_t kDirtyListMax = 1ull << 27;
static_assert(kBrickShift >= 4 && kBrickShift <= 8, "a brick is 16 .. 256 cells (HitRec keeps the cell in 8 bits)");

// (ox, oy, oz: a voxel of the output grid inside the allocated box, see Params::xo0)
__device__ __forceinline__ uint64_t cell_index(uint32_t ox, uint32_t oy, uint32_t oz, const Params &p, uint32_t &brick)
{
    const uint32_t rx = ox - p.xo0, ry = oy - p.yo0, rz = oz - p.zo0;
    brick = ((rz >> kBrickZs) * p.NBy + (ry >> kBrickYs)) * p.NBx + (rx >> kBrickXs);
    return (uint64_t) brick * kBrickCells +
           ((((rz & (kBrickZ - 1u)) << kBrickYs) + (ry & (kBrickY - 1u))) << kBrickXs) + (rx & (kBrickX - 1u));
}
// the same for a voxel given relative to the grid's origin (output cells; Params::so)
__device__ __forceinline__ uint64_t cell_index_rel(uint32_t rx, uint32_t ry, uint32_t rz, const Params &p, uint32_t &brick)
{
    brick = ((rz >> kBrickZs) * p.NBy + (ry >> kBrickYs)) * p.NBx + (rx >> kBrickXs);
    return (uint64_t) brick * kBrickCells +
           ((((rz & (kBrickZ - 1u)) << kBrickYs) + (ry & (kBrickY - 1u))) << kBrickXs) + (rx & (kBrickX - 1u));
}
// the first voxel of a brick, relative to the grid's origin (output cells)
__device__ __forceinline__ void brick_origin_rel(uint32_t brick, const Params &p, uint32_t &x, uint32_t &y, uint32_t &z)
{
    const uint32_t row = brick / p.NBx;
    const uint32_t bx = brick - row * p.NBx;
    const uint32_t bz = row / p.NBy;
    const uint32_t by = row - bz * p.NBy;
    x = bx << kBrickXs;
    y = by << kBrickYs;
    z = bz << kBrickZs;
}
// the first voxel (output grid) of a brick of a grid whose bricks are numbered x fastest
__device__ __forceinline__ void brick_origin(uint32_t brick, const Params &p, uint32_t &x, uint32_t &y, uint32_t &z)
{
    const uint32_t row = brick / p.NBx;
    const uint32_t bx = brick - row * p.NBx;
    const uint32_t bz = row / p.NBy;
    const uint32_t by = row - bz * p.NBy;
    x = (bx << kBrickXs) + p.xo0;
    y = (by << kBrickYs) + p.yo0;
    z = (bz << kBrickZs) + p.zo0;
}
// position (output grid) of a brick's cell `local`
__device__ __forceinline__ void cell_position(uint32_t brick, uint32_t local, const Params &p, uint32_t &x, uint32_t &y, uint32_t &z)
{
    brick_origin(brick, p, x, y, z);
    x += local & (kBrickX - 1u);
    y += (local >> kBrickXs) & (kBrickY - 1u);
    z += local >> (kBrickXs + kBrickYs);
}

// Inclusive prefix sum of one uint32 per lane over the wavefront, with DPP adds: four row_shr steps inside each row of 16 lanes,
// then row_bcast:15 / row_bcast:31 carry the rows' totals on (gfx9 wave64).  Six dependent VALU instructions; the same scan
// with __shfl_up is six dependent LDS-crossbar round trips (ds_bpermute), and k_voxelize's phase 1 - one scan per 64 candidate
// rows - waits on such chains rather than on instruction issue (profiles/r05/NOTES.md).
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
#ifndef O2V_NO_DPP_SCAN
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, true);   // row_shr:1 (lanes without a source add 0)
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
#else
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
#endif
}

// exclusive scan of one uint32 per thread over a 256-thread block; returns the block total in `total`
__device__ __forceinline__ uint32_t block_exscan(uint32_t v, uint32_t *s_wave /*[4]*/, uint32_t &total)
{
    uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v);
    __syncthreads();
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < kBlock / 64; ++w) {
        uint32_t c = s_wave[w];
        if (w < wave) base += c;
        tot += c;
    }
    total = tot;
    return base + inc - v;
}

// colorAt_f in three steps, for code that looks several colours up at once: the loads of one step are independent of each
// other (and of the other colours' loads), so n lookups cost three memory round trips instead of 3 n.
struct MatFetch {  // step 1: what the triangle's material is (independent loads)
    uint32_t type, texid;
    float r, g, b;
};
__device__ __forceinline__ MatFetch mat_fetch(const Materials &m, uint32_t tri)
{
    MatFetch f;
    f.type = m.types ? m.types[tri] : (uint32_t) kTriMaterialless;
    f.texid = m.texids ? (uint32_t) m.texids[tri] : 0u;
    f.r = m.colors ? m.colors[(size_t) tri * 3 + 0] : 0.f;
    f.g = m.colors ? m.colors[(size_t) tri * 3 + 1] : 0.f;
    f.b = m.colors ? m.colors[(size_t) tri * 3 + 2] : 0.f;
    return f;
}
// step 2: the texel's address (texture get, triangle.hpp:161-166; getPixel semantics: see DESIGN.md)
__device__ __forceinline__ const uint8_t *texel_address(const DevTexture &tx, float u, float v)
{
    float tu = u, tv = 1 - v;
    if (tx.wrap) {
        tu = tu - floor_f(tu);
        tv = tv - floor_f(tv);
    }
    else {
        tu = tu < 0.f ? 0.f : (tu > 1.f ? 1.f : tu);
        tv = tv < 0.f ? 0.f : (tv > 1.f ? 1.f : tv);
    }
    uint32_t px = (uint32_t) (tu * (float) tx.width), py = (uint32_t) (tv * (float) tx.height);
    if (px >= tx.width) px = tx.width - 1;
    if (py >= tx.height) py = tx.height - 1;
    return tx.pixels + ((size_t) py * tx.width + px) * tx.channels + (tx.channels == 4 ? 1u : 0u);
}
// The same texel as one or two aligned 32-bit loads instead of three byte loads (a lookup per lane: every load instruction
// of a wavefront touches 64 different cache lines, and the lane tiers are bound by exactly that).  `word` holds the texel's
// first byte at bit `shift`; a second word is needed when the three bytes straddle it (never for 4-channel textures, whose
// r, g, b are bytes 1..3 of their word).  Texture allocations end with 8 spare bytes (o2v_hip_set_textures).
struct TexelRef {
    const uint32_t *word;
    uint32_t shift;  // 0, 8, 16, 24
    __device__ __forceinline__ bool straddles() const { return shift >= 16u; }
};
__device__ __forceinline__ TexelRef texel_ref(const uint8_t *first_byte)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(first_byte);
    return TexelRef{reinterpret_cast<const uint32_t *>(a & ~(uintptr_t) 3u), (uint32_t) (a & 3u) * 8u};
}
__device__ __forceinline__ void texel_bytes(const TexelRef &t, uint32_t w0, uint32_t w1, uint8_t &q0, uint8_t &q1, uint8_t &q2)
{
    const uint32_t x = (uint32_t) ((((uint64_t) w1 << 32) | w0) >> t.shift);
    q0 = (uint8_t) x;
    q1 = (uint8_t) (x >> 8);
    q2 = (uint8_t) (x >> 16);
}
// step 3: the colour from the material and the three texel bytes
__device__ __forceinline__ void mat_color(const MatFetch &f, bool have_textures, uint8_t q0, uint8_t q1, uint8_t q2, float &r, float &g, float &b)
{
    if (f.type == kTriMaterialless) {
        r = g = b = 1.f;
    }
    else if (f.type == kTriUntextured) {
        r = f.r;
        g = f.g;
        b = f.b;
    }
    else if (f.type == kTriTextured && have_textures) {
        r = (float) q0 / 255.f;
        g = (float) q1 / 255.f;
        b = (float) q2 / 255.f;
    }
    else {
        r = 1.f;
        g = 0.f;
        b = 1.f;
    }
}

// colorAt_f, triangle.hpp:181-194 (+ texture get, triangle.hpp:161-166; getPixel semantics: see DESIGN.md)
__device__ __forceinline__ void color_at(const Materials &m, uint32_t tri, float u, float v, float &r, float &g, float &b)
{
    const uint32_t type = m.types ? m.types[tri] : (uint32_t) kTriMaterialless;
    if (type == kTriMaterialless) {
        r = g = b = 1.f;
    }
    else if (type == kTriUntextured) {
        r = m.colors ? m.colors[(size_t) tri * 3 + 0] : 0.f;
        g = m.colors ? m.colors[(size_t) tri * 3 + 1] : 0.f;
        b = m.colors ? m.colors[(size_t) tri * 3 + 2] : 0.f;
    }
    else if (type == kTriTextured && m.n_textures) {
        uint32_t id = m.texids ? (uint32_t) m.texids[tri] : 0u;
        if (id >= m.n_textures) id = 0;
        const DevTexture tx = m.textures[id];
        float tu = u, tv = 1 - v;
        if (tx.wrap) {
            tu = tu - floor_f(tu);
            tv = tv - floor_f(tv);
        }
        else {
            tu = tu < 0.f ? 0.f : (tu > 1.f ? 1.f : tu);
            tv = tv < 0.f ? 0.f : (tv > 1.f ? 1.f : tv);
        }
        uint32_t px = (uint32_t) (tu * (float) tx.width), py = (uint32_t) (tv * (float) tx.height);
        if (px >= tx.width) px = tx.width - 1;
        if (py >= tx.height) py = tx.height - 1;
        const TexelRef t = texel_ref(tx.pixels + ((size_t) py * tx.width + px) * tx.channels + (tx.channels == 4 ? 1u : 0u));
        const uint32_t w0 = t.word[0], w1 = t.straddles() ? t.word[1] : 0u;
        uint8_t q0, q1, q2;
        texel_bytes(t, w0, w1, q0, q1, q2);
        r = (float) q0 / 255.f;
        g = (float) q1 / 255.f;
        b = (float) q2 / 255.f;
    }
    else {
        r = 1.f;
        g = 0.f;
        b = 1.f;
    }
}

