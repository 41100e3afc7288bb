// o2v_dev_k2_voxelize.hpp -- K2: candidate test and six-plane clip (k_voxelize<UV>).
//
// Part of the device code of o2v_device.hip, which includes this file inside its anonymous namespace (one
// translation unit: the stages share records and launch parameters).  Not a stand-alone header.

// ---- K2: voxelize -------------------------------------------------------------------------------------------

template <bool UV>
struct Piece {  // TexturedTriangle (triangle.hpp:113-144); the uv members are dead code when !UV
    V3 a, b, c;
    V2 ta, tb, tc;
};

// Classification of one piece against one axis plane: SplittingValues + the case switch of splitTriangle
// (voxelization.cpp:110-153,190-232).  Packed so that it can be carried in one register between the cheap
// classification pass and the expensive split pass of the clip loop.
enum : uint32_t {
    kClsModeMask = 3u,   // 0: whole triangle goes to one side, 1: one-planar split, 2: regular split
    kClsSideLo = 4u,     // mode 0: the side is "lo"
    kClsRotShift = 3u,   // bits 3..4: rotation index r (planar vertex for mode 1, isolated vertex for mode 2)
    kClsFlagLo = 32u,    // mode 1: lo flag of vertex r+1;  mode 2: the isolated vertex is lo
};

// The case analysis on the six flags (lo and planar per vertex), voxelization.cpp:190-232; index = l0 | l1 << 1 | l2 << 2 |
// p0 << 3 | p1 << 4 | p2 << 5.  The kernel keeps the 64 results in an LDS table: one byte load replaces the branches.
__host__ __device__ constexpr uint32_t classify_flags(uint32_t idx)
{
    const bool l0 = idx & 1u, l1 = idx & 2u, l2 = idx & 4u, p0 = idx & 8u, p1 = idx & 16u, p2 = idx & 32u;
    const uint32_t lo_sum = (uint32_t) l0 + (uint32_t) l1 + (uint32_t) l2;
    const uint32_t pl_sum = (uint32_t) p0 + (uint32_t) p1 + (uint32_t) p2;
    if (lo_sum == 0) return 0u;
    if (lo_sum == 3) return kClsSideLo;
    if (pl_sum == 3) return 0u;  // parallel to the plane: pushed by bias (IS_LO_BIASED = false) = hi
    if (pl_sum == 2) return (!p0 ? l0 : (!p1 ? l1 : l2)) ? kClsSideLo : 0u;
    if (pl_sum == 1) {
        const uint32_t r = p0 ? 0u : (p1 ? 1u : 2u);
        const bool lq = r == 0 ? l1 : (r == 1 ? l2 : l0);
        const bool lr = r == 0 ? l2 : (r == 1 ? l0 : l1);
        if (lq == lr) return lq ? kClsSideLo : 0u;
        return 1u | (r << kClsRotShift) | (lq ? kClsFlagLo : 0u);
    }
    const bool iso_lo = lo_sum == 1;
    const uint32_t r = iso_lo ? (l0 ? 0u : (l1 ? 1u : 2u)) : (!l0 ? 0u : (!l1 ? 1u : 2u));
    return 2u | (r << kClsRotShift) | (iso_lo ? kClsFlagLo : 0u);
}

// Number of pieces splitTriangle<DISCARD_*> keeps (voxelization.cpp:190-232, 236-331) for the same index; keep_lo selects
// DISCARD_HI: a whole triangle is kept or not, the one-planar case keeps one of its two pieces, the regular case keeps
// the isolated vertex's triangle (1) or the quad on the other side, emitted as two triangles (2).
__host__ __device__ constexpr uint32_t classify_kept(uint32_t idx, bool keep_lo)
{
    const uint32_t cls = classify_flags(idx);
    const uint32_t mode = cls & kClsModeMask;
    if (mode == 0u) return ((cls & kClsSideLo) != 0u) == keep_lo ? 1u : 0u;
    if (mode == 1u) return 1u;
    return ((cls & kClsFlagLo) != 0u) == keep_lo ? 1u : 2u;
}

// SplittingValues, voxelization.cpp:121-131: the six flags of a piece against an axis plane
__device__ __forceinline__ uint32_t classify_index(float c0, float c1, float c2, float plane)
{
    const bool p0 = abs_f(c0 - plane) < kEpsilon, p1 = abs_f(c1 - plane) < kEpsilon, p2 = abs_f(c2 - plane) < kEpsilon;
    const bool l0 = c0 < plane, l1 = c1 < plane, l2 = c2 < plane;
    return (uint32_t) l0 | ((uint32_t) l1 << 1) | ((uint32_t) l2 << 2) | ((uint32_t) p0 << 3) | ((uint32_t) p1 << 4) | ((uint32_t) p2 << 5);
}

// The geometric part of splitTriangle<DISCARD_LO|DISCARD_HI> for a piece whose classification says it is cut
// (modes 1 and 2).  keep_lo selects DISCARD_HI.  Returns the number of kept pieces (1 or 2): `cur` becomes the
// first kept piece in emission order, `sec` the second.  Vertex order inside emitted pieces is the reference's,
// because later splits depend on it.
template <bool UV>
__device__ __forceinline__ uint32_t split_cut(Piece<UV> &cur, Piece<UV> &sec, uint32_t cls, uint32_t axis, float plane,
                                              bool keep_lo, bool lean /* wave-uniform: see lean_ok */)
{
    const uint32_t r = (cls >> kClsRotShift) & 3u;
    const bool flag_lo = (cls & kClsFlagLo) != 0;
    // rotate (a, b, c) left by r with two conditional cyclic shifts (18 selects instead of 54); every select names its lane
    // mask (vsel: o2v_dev_arith.hpp)
    const unsigned long long s1 = lane_mask(r >= 1u), s2 = lane_mask(r == 2u);
    const V3 P1 = vsel(s1, cur.b, cur.a), Q1 = vsel(s1, cur.c, cur.b), R1 = vsel(s1, cur.a, cur.c);
    const V3 P = vsel(s2, Q1, P1), Q = vsel(s2, R1, Q1), R = vsel(s2, P1, R1);
    V2 tP{}, tQ{}, tR{};
    if (UV) {
        const V2 tP1 = vsel(s1, cur.tb, cur.ta), tQ1 = vsel(s1, cur.tc, cur.tb), tR1 = vsel(s1, cur.ta, cur.tc);
        tP = vsel(s2, tQ1, tP1);
        tQ = vsel(s2, tR1, tQ1);
        tR = vsel(s2, tP1, tR1);
    }
    const float cP = comp(P, axis), cQ = comp(Q, axis), cR = comp(R, axis);
    const bool regular = (cls & kClsModeMask) == 2u;
    const unsigned long long m_reg = lane_mask(regular);
    // first intersection: regular case P->Q (voxelization.cpp:305-311), one-planar case Q->R (:262-266)
    const V3 A0 = vsel(m_reg, P, Q), A1 = vsel(m_reg, Q, R);
    const float cA0 = vsel(m_reg, cP, cQ), cA1 = vsel(m_reg, cQ, cR);
    const float d0 = -(cA1 - cA0);
    const float i0 = abs_f(d0) < kEpsilon ? 0.f : (lean ? div_lean(cA0 - plane, d0) : (cA0 - plane) / d0);
    const V3 G0 = mix(A0, A1, i0);
    V2 x0{};
    if (UV) x0 = mix(vsel(m_reg, tP, tQ), vsel(m_reg, tQ, tR), i0);
    // second intersection, regular case only: P isolated, P->R (splitTriangle_regularCase, voxelization.cpp:279-331)
    V3 G1{};
    V2 x1{};
    if (regular) {
        const float d1 = -(cR - cP);
        const float i1 = abs_f(d1) < kEpsilon ? 0.f : (lean ? div_lean(cP - plane, d1) : (cP - plane) / d1);
        G1 = mix(P, R, i1);
        if (UV) x1 = mix(tP, tR, i1);
    }
    // The kept pieces, in emission order:
    //   one-planar (splitTriangle_onePlanarCase): {P, Q, G0} is on Q's side, {P, G0, R} on the other - the one on the kept
    //     side stays;
    //   regular, the isolated vertex is kept: {P, G0, G1};
    //   regular, the other side is kept: the quad as {G0, Q, R} and {G0, G1, R}.
    const bool same = flag_lo == keep_lo;
    const bool quad = regular && !same;
    const unsigned long long m_quad = lane_mask(quad);                      // a = G0, else P
    const unsigned long long m_bq = lane_mask(quad || (!regular && same));  // b = Q, else G0
    const unsigned long long m_cr = lane_mask(!same);                       // c = R, else (regular ? G1 : G0)
    cur.a = vsel(m_quad, G0, P);
    cur.b = vsel(m_bq, Q, G0);
    cur.c = vsel(m_cr, R, vsel(m_reg, G1, G0));
    sec.a = G0;
    sec.b = G1;
    sec.c = R;
    if (UV) {
        cur.ta = vsel(m_quad, x0, tP);
        cur.tb = vsel(m_bq, tQ, x0);
        cur.tc = vsel(m_cr, tR, vsel(m_reg, x1, x0));
        sec.ta = x0;
        sec.tb = x1;
        sec.tc = tR;
    }
    return quad ? 2u : 1u;
}

// The lean divisions (o2v_dev_arith.hpp) are used for a voxel job whose operands are known to lie in the middle of float32's
// range.  The cut parameter (n, d) = (c - plane, -(c' - c)): the leaf is `small` (finite coordinates below 2^17), n belongs to
// a vertex that is not planar (|n| >= 2^-16, classify_index) and |d| >= 2^-16 by the branch around the division, so both
// lie in [2^-16, 2^18].  The uv mean, (w u + area uc) / (w + area): the leaf's area is within [2^-30, 2^30] and its uv
// coordinates are at most 2^20 in magnitude (kLeanArea*, kLeanUv: checked once per staged leaf), so the divisor lies in
// [2^-30, 2^37] (w is a small multiple of area) and the quotient - a mean of uv coordinates - is at most ~2^20.  A numerator
// that cancels to almost nothing (not +0, which is exact) could leave the box at its lower end: it shows as a quotient below
// kLeanMinQuotient = 2^-50, and such a lane repeats the division the long way; any other numerator is at least 2^-81.
constexpr float kLeanAreaMin = 9.313225746154785e-10f, kLeanAreaMax = 1073741824.0f;  // 2^-30, 2^30
constexpr float kLeanUv = 1048576.0f;                                                   // 2^20
constexpr float kLeanMinQuotient = 8.881784197001252e-16f;                              // 2^-50

template <bool UV>
__device__ __forceinline__ void accumulate_piece(const Piece<UV> &pc, float area, float &w, float &u, float &v, bool lean /* wave-uniform */)
{
    // result = mix(result, {area(inputTriangle), piece.textureCenter()}), voxelization.cpp:414-420, util.hpp:160-165
    const float ws = w + area;
    if (UV) {
        const float uc = third((pc.ta.x + pc.tb.x) + pc.tc.x);
        const float vc = third((pc.ta.y + pc.tb.y) + pc.tc.y);
        const float nu = w * u + area * uc, nv = w * v + area * vc;
        if (lean) {
            const LeanRecip rc(ws);
            float qu = rc.divide(nu), qv = rc.divide(nv);
            const bool redo = (abs_f(qu) < kLeanMinQuotient && __float_as_uint(nu) != 0u) || (abs_f(qv) < kLeanMinQuotient && __float_as_uint(nv) != 0u);
            if (redo) {
                qu = nu / ws;
                qv = nv / ws;
            }
            u = qu;
            v = qv;
        }
        else {
            u = nu / ws;
            v = nv / ws;
        }
    }
    w = ws;
}

// Pending sibling pieces of the depth-first clip walk: a stack (last in, first out - the sibling pushed last belongs to
// the deepest level and is next in depth-first order).  Under DISCARD a cut keeps <= 2 pieces, so at most one sibling per
// level 1..5 is pending.  The first entries live in registers and are accessed with value selects on the stack pointer
// (every access uses a compile-time slot, so they never leave the VGPR file): three without uv (deeper entries are 1 % of
// the pushes on the bench mesh), two with uv, whose pieces are 15 floats (9 % of the pushes go deeper; measured faster on
// configs[1] and configs[3] than a third slot and its spills).  Deeper entries go to a small per-lane overflow array
// (scratch memory).
#ifndef O2V_STACK_REGS
#define O2V_STACK_REGS 3
#endif
#ifndef O2V_STACK_REGS_UV
#define O2V_STACK_REGS_UV 2
#endif
template <bool UV>
__device__ __forceinline__ constexpr uint32_t stack_regs() { return UV ? O2V_STACK_REGS_UV : O2V_STACK_REGS; }
constexpr uint32_t kStackLevels = 5;
template <bool UV>
struct PieceStack {
    Piece<UV> s0, s1, s2;
};

// Value-level selects (v_cndmask), not control flow: a branchy form gets folded by the compiler into a select of
// addresses, which forces the stack into scratch memory.
template <bool UV>
__device__ __forceinline__ Piece<UV> sel_piece(bool take_x, const Piece<UV> &x, const Piece<UV> &y)
{
    const unsigned long long m = lane_mask(take_x);
    Piece<UV> r;
    r.a = vsel(m, x.a, y.a);
    r.b = vsel(m, x.b, y.b);
    r.c = vsel(m, x.c, y.c);
    if (UV) {
        r.ta = vsel(m, x.ta, y.ta);
        r.tb = vsel(m, x.tb, y.tb);
        r.tc = vsel(m, x.tc, y.tc);
    }
    return r;
}
template <bool UV>
__device__ __forceinline__ void stack_store(PieceStack<UV> &st, uint32_t slot, const Piece<UV> &pc)
{
    if (stack_regs<UV>() > 0u) st.s0 = sel_piece<UV>(slot == 0, pc, st.s0);
    if (stack_regs<UV>() > 1u) st.s1 = sel_piece<UV>(slot == 1, pc, st.s1);
    if (stack_regs<UV>() > 2u) st.s2 = sel_piece<UV>(slot == 2, pc, st.s2);
}
template <bool UV>
__device__ __forceinline__ Piece<UV> stack_load(const PieceStack<UV> &st, uint32_t slot)
{
    if (stack_regs<UV>() > 2u) return sel_piece<UV>(slot == 0, st.s0, sel_piece<UV>(slot == 1, st.s1, st.s2));
    if (stack_regs<UV>() > 1u) return sel_piece<UV>(slot == 0, st.s0, st.s1);
    return st.s0;
}

// Conservative triangle / voxel overlap test (separating axes: the triangle's plane and the nine edge x axis
// directions; the three box axes are implied by the AABB walk).  The box is inflated by kSatMargin, far more than
// the float32 rounding of the clip (<= a few ulp of the coordinate, 5e-4 at 4096) and than its planarity epsilon
// (2^-16), so every voxel the exact clip can mark is kept: this only removes work, never results.  The test's own
// rounding is covered too (below); the plane axis uses the unnormalised normal e0 x e1 with an explicit error bound (for
// a sliver, whose normal direction is numerically meaningless, the bound exceeds the radius and the plane axis simply
// never separates), and all comparisons are written so that a NaN rejects nothing.  Not reference arithmetic.
// For leaves far from the origin the clip's own rounding grows with the coordinates - a vertex of a sub-piece drifts by up to
// 1.2e-6 x the largest |coordinate| m over the five generations (see piece_masks) - so the inflation does too:
// sat_margin(m) = max(0.02, 2.5e-6 m), twice that drift (0.02 up to m = 8000, 0.16 at m = 65536).
constexpr float kSatMargin = 0.02f;
__device__ __forceinline__ float sat_margin(float m) { return fmaxf(kSatMargin, 2.5e-6f * m); }

// The test is solved for x: for one row of candidate voxels (fixed y and z of the leaf's clamped AABB) every axis is a
// linear function of the voxel centre's x, so it bounds x to an interval; the row's candidates are the voxels whose centre
// lies in the intersection of the intervals.  Evaluated in the leaf's own frame (origin = the AABB's first voxel, so all
// magnitudes are extents, not positions).  The box is inflated by kSatMargin in projection space, which covers the
// rounding of the projections (a few 1e-7 of extent x edge length against 0.02 x edge length); the division by the axis'
// x component adds a relative 1e-7, covered by `slack` voxels on either side.  An axis whose x component (nearly)
// vanishes is not used (the x x edge axes, which do not depend on x at all, reject whole rows).
struct RowClip {
    float lo, hi;  // interval of admissible voxel centres (x, leaf frame)
    bool any;
    __device__ __forceinline__ void bound(float a, float lo_ax, float hi_ax, float slack)
    {
        // a * x within [lo_ax, hi_ax]
        if (abs_f(a) > 1e-20f) {
            const float r = __builtin_amdgcn_rcpf(a);  // (1 ulp: part of the relative error `slack` covers)
            const float x0 = lo_ax * r, x1 = hi_ax * r;
            lo = fmaxf(lo, fminf(x0, x1) - slack);
            hi = fminf(hi, fmaxf(x0, x1) + slack);
        }
    }
};
__device__ __forceinline__ void row_clip_edge(RowClip &rc, V3 E, V3 U, V3 W, float cy, float cz, float slack, float h)
{
    // axis x x E = (0, -E.z, E.y): the same for every voxel of the row
    {
        const float pu = E.z * (U.y - cy) - E.y * (U.z - cz), pw = E.z * (W.y - cy) - E.y * (W.z - cz);
        const float rad = h * (abs_f(E.z) + abs_f(E.y));
        if (fminf(pu, pw) > rad || fmaxf(pu, pw) < -rad) rc.any = false;
    }
    // axis y x E = (E.z, 0, -E.x): q(P) = E.x (P.z - cz) - E.z (P.x - cx) = alpha(P) + E.z cx
    {
        const float au = E.x * (U.z - cz) - E.z * U.x, aw = E.x * (W.z - cz) - E.z * W.x;
        const float rad = h * (abs_f(E.x) + abs_f(E.z));
        rc.bound(E.z, -rad - fmaxf(au, aw), rad - fminf(au, aw), slack);
    }
    // axis z x E = (-E.y, E.x, 0): q(P) = E.y (P.x - cx) - E.x (P.y - cy) = beta(P) - E.y cx
    {
        const float bu = E.y * U.x - E.x * (U.y - cy), bw = E.y * W.x - E.x * (W.y - cy);
        const float rad = h * (abs_f(E.y) + abs_f(E.x));
        rc.bound(-E.y, -rad - fmaxf(bu, bw), rad - fminf(bu, bw), slack);
    }
}
// p0, p1, p2: the leaf's vertices relative to the AABB origin; cy, cz: the row's voxel centre; [xlo, xhi]: the row's
// voxels that belong to the tile.  Returns the first admissible voxel and their number.
__device__ __forceinline__ uint32_t row_span(V3 p0, V3 p1, V3 p2, float cy, float cz, uint32_t xlo, uint32_t xhi, float extent,
                                             float margin, uint32_t &first)
{
    const float h = 0.5f + margin;  // margin = sat_margin(the leaf's largest |coordinate|)
    const float slack = 0.01f + 4e-6f * extent;
    RowClip rc{(float) xlo + 0.5f, (float) xhi + 0.5f, true};
    const V3 e0 = p1 - p0, e1 = p2 - p1, e2 = p0 - p2;
    {
        // plane axis n = e0 x e1: |n . (c - p0)| <= h |n|_1 + err, with an error bound (see above) for the farthest voxel
        const V3 n = cross(e0, e1);
        const float rest = n.y * (cy - p0.y) + n.z * (cz - p0.z) - n.x * p0.x;  // n . (c - p0) = n.x cx + rest
        const float l0 = abs_f(e0.x) + abs_f(e0.y) + abs_f(e0.z), l1 = abs_f(e1.x) + abs_f(e1.y) + abs_f(e1.z);
        const float far = extent + abs_f(p0.x) + abs_f(p0.y) + abs_f(p0.z);  // >= |c - p0|_1 for every voxel of the AABB
        const float rad = h * (abs_f(n.x) + abs_f(n.y) + abs_f(n.z)) + 1e-5f * l0 * l1 * (far + 1.0f);
        rc.bound(n.x, -rad - rest, rad - rest, slack);
    }
    row_clip_edge(rc, e0, p0, p2, cy, cz, slack, h);
    row_clip_edge(rc, e1, p1, p0, cy, cz, slack, h);
    row_clip_edge(rc, e2, p2, p1, cy, cz, slack, h);
    // voxel x has its centre at x + 0.5; the interval is clamped to the row before the conversion
    const float fa = ceilf(rc.lo - 0.5f), fb = floorf(rc.hi - 0.5f);
    if (!rc.any || !(fa <= fb)) {
        first = xlo;
        return 0u;
    }
    first = (uint32_t) fa;
    return (uint32_t) fb - (uint32_t) fa + 1u;
}


// Early decisions about a piece from its bounding box (speed only, results unchanged).  For the planes in `planes` (bit =
// level: lo x, y, z, hi x, y, z) of the voxel at (fx, fy, fz):
//   fail  planes the piece does not pass whole.  A piece whose vertices all satisfy v >= plane (lo planes) or v < plane
//         (hi planes) is the loSum == 0 / loSum == 3 case of splitTriangle (voxelization.cpp:194-205) and is handed on
//         unchanged, so only the `fail` planes need the classification.
//   out   planes the piece lies entirely beyond, by more than the leaf's margin.  Every sub-piece of it then lies beyond
//         that plane too - a vertex of a sub-piece is (1-t)*a + t*b of two vertices of its parent with t in [0, 1]
//         (float subtraction is monotonic, so |a - plane| <= |a - b| survives rounding), off their range by the rounding
//         of s = 1 - t, two products and a sum: < 2 * 2^-23 m per generation, < 1.2e-6 m over the five possible
//         generations, for a leaf whose largest |coordinate| is m - and is discarded there at the latest, contributing
//         nothing before.  The margin (out_margin) is eight times that plus twice the 2^-16 planarity band.  The piece is
//         dropped at once.
//   near  planes the piece does not pass whole by more than the same margin (a superset of fail), see single_plane.
// All are only computed for jobs whose leaf has all |coordinates| < 2^17 (`small`: the margins scale with the coordinates and
// the sample grid has at most 65 535 voxels per axis, so this only excludes meshes reaching far outside the grid, whose
// float arithmetic is too coarse to argue about; false for NaN too, and for every leaf
// in exact mode, Params::exact_clip); other jobs get fail = near = planes, out = 0, i.e. every plane is classified, which
// is always exact (single_plane then only fires at the last plane, hi z, where no later plane exists and the number of
// kept pieces is the classification's by definition).
constexpr float kSmallCoord = 131072.0f;

// (once per staged leaf: q = its nine vertex coordinates; m = the largest |coordinate|)
__device__ __forceinline__ bool leaf_is_small(const uint32_t *q, float &m)
{
    float sum = 0.f;
    m = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const float f = __uint_as_float(q[i]);
        m = fmaxf(m, abs_f(f));
        sum += f;
    }
    // sum == sum is false if any coordinate is NaN (fmaxf ignores NaN operands) or inf - inf occurred
    return sum == sum && m < kSmallCoord;
}
__device__ __forceinline__ float out_margin(float m) { return 3.0517578125e-5f + 1e-5f * m; }

// The eighteen conditions are read off the sign bits of differences: one subtraction (the cheap issue class) and one
// v_alignbit (which shifts a sign bit into the mask) per condition instead of a compare and a select.  With dn = min - f and
// dx = max - f per axis (f: the voxel's integer coordinate):
//   fail lo   min < f            sign(dn)          exact: a difference of two floats is negative iff the first is smaller
//   fail hi   max >= f + 1       !sign(dx - 1)     exact: dx is exact for max in [f/2, 2f] (Sterbenz) and on the right side
//                                                  of 1 outside it (rounding is monotonic); likewise dx - 1
//   out / near                   the same differences against -margin, 1 + margin, margin, 1 - margin: their rounding (a few
//                                1e-8 of the coordinate) is far inside the factor 8 the margin has over the drift it covers
// The one case in which a sign bit and the comparison differ is min = -0 at f = 0 (-0 - 0 = -0): a spurious `fail` bit, which
// only sends the plane through the classification (always exact).
__device__ __forceinline__ uint32_t push_sign(uint32_t mask, float d) { return __builtin_amdgcn_alignbit(mask, __float_as_uint(d), 31); }

template <bool UV>
__device__ __forceinline__ void piece_masks(const Piece<UV> &q, float fx, float fy, float fz, bool small, float margin, uint32_t planes,
                                            uint32_t &fail, uint32_t &out, uint32_t &near)
{
    // all coordinates are finite here (small), so min / max need no NaN rule
    const float nx = fminf(fminf(q.a.x, q.b.x), q.c.x), ny = fminf(fminf(q.a.y, q.b.y), q.c.y), nz = fminf(fminf(q.a.z, q.b.z), q.c.z);
    const float xx = fmaxf(fmaxf(q.a.x, q.b.x), q.c.x), xy = fmaxf(fmaxf(q.a.y, q.b.y), q.c.y), xz = fmaxf(fmaxf(q.a.z, q.b.z), q.c.z);
    const float dnx = nx - fx, dny = ny - fy, dnz = nz - fz, dxx = xx - fx, dxy = xy - fy, dxz = xz - fz;
    const float one_p = 1.0f + margin, one_m = 1.0f - margin;
    // (bit order: the sign pushed last is bit 0 - lo x, y, z, hi x, y, z = bits 0..5)
    uint32_t f = 0u, o = 0u, r = 0u;
    f = push_sign(f, dxz - 1.0f); f = push_sign(f, dxy - 1.0f); f = push_sign(f, dxx - 1.0f);
    f = push_sign(f, dnz); f = push_sign(f, dny); f = push_sign(f, dnx);
    f ^= 0x38u;  // hi planes: fail = NOT (max < f + 1)
    o = push_sign(o, one_p - dnz); o = push_sign(o, one_p - dny); o = push_sign(o, one_p - dnx);
    o = push_sign(o, dxz + margin); o = push_sign(o, dxy + margin); o = push_sign(o, dxx + margin);
    r = push_sign(r, dxz - one_m); r = push_sign(r, dxy - one_m); r = push_sign(r, dxx - one_m);
    r = push_sign(r, dnz - margin); r = push_sign(r, dny - margin); r = push_sign(r, dnx - margin);
    r ^= 0x38u;
    fail = small ? (f & planes) : planes;
    out = small ? (o & planes) : 0u;
    near = small ? (r & planes) : planes;
}

// A piece that fails exactly one of the planes ahead and passes those after it by more than the margin (no `near` bit above
// the one `fail` bit; the planes before it are passed whole, which is all the piece itself needs): the pieces its cut at
// that plane keeps pass every later plane whole (same bound as for `out`), i.e. they are final.
// Without uv only their number matters (each adds the leaf's area, voxelization.cpp:414-420), and the number follows from
// the classification alone - no intersection points, no further iteration.
__device__ __forceinline__ bool single_plane(uint32_t fail, uint32_t near) { return fail != 0u && (fail & (fail - 1u)) == 0u && (near ^ fail) < fail; }
template <bool UV>
__device__ __forceinline__ uint32_t single_plane_kept(const Piece<UV> &q, uint32_t fail, float fx, float fy, float fz, const uint8_t *s_kept)
{
    const uint32_t level = (uint32_t) __ffs((int) fail) - 1u;
    const bool keep_lo = level >= 3u;
    const uint32_t axis = keep_lo ? level - 3u : level;
    const float plane = (axis == 0 ? fx : (axis == 1 ? fy : fz)) + (keep_lo ? 1.0f : 0.0f);
    return s_kept[classify_index(comp(q.a, axis), comp(q.b, axis), comp(q.c, axis), plane) | (keep_lo ? 64u : 0u)];
}

// ---- occupancy-only mode: hits that need no clipping ---------------------------------------------------------------------
// In occupancy-only mode (Params::occupancy_only) a voxel job only has to establish that the clip leaves at least one piece.
// For most hit voxels that is certain beforehand: if a point q of the leaf lies inside the voxel by a margin D and inside the
// leaf (in its plane) by a margin rho, the piece of the clip that contains q cannot vanish - the float clip's pieces cover
// the exact intersection of leaf and voxel except for a band along its boundary no wider than the drift of their vertices
// (< 1.2e-6 x the largest |coordinate| m over the five generations, see piece_masks) plus the 2^-16 planarity band (a vertex
// that close to a plane counts as on it; a piece with two such vertices goes to the third one's side: both move the boundary
// by < 2^-16).  With D = rho = certain_margin(m) = 1.5 sat_margin(m) >= 0.03 that band is thirty times narrower than the
// margins.  The point tried is the one the voxel centre's column along the normal's dominant axis d meets the leaf's plane
// in: q = centre - t e_d, t = n . (centre - v0) / n_d.  It lies in the voxel by 0.5 in the two other axes and by 0.5 - |t| in
// d; its barycentric coordinates - computed in the projection along d, which preserves them - say how far inside the leaf it
// is: lambda_i x (altitude on vertex i) is its distance from the edge opposite i.  So the test is
//     |t| <= 0.5 - D   and   lambda_0, lambda_1, lambda_2 >= tau = rho / (smallest altitude) + 1e-4
// and it is tried not only for the centre's column but for every column through five lines across the voxel (spaced 0.22
// or less in one of the two other axes, any offset within 0.5 - D along the other: along a line all conditions are linear
// inequalities in the offset, which leave an interval), so that q stays inside the voxel by >= D in those axes too.  Some 150
// instructions per candidate, all lanes busy, against a voxel job's ~4000 on partly idle lanes.  A sliver has tau > 1/3 and is never certain; a
// degenerate or non-finite leaf is excluded outright, as is a leaf of a triangle of zero area (its weight would be zero:
// no hit, voxelization.cpp:466).  Not reference arithmetic: like the separating-axis test it only removes work, and the
// fast-vs-exact comparison (tests/test_gpu_exact_ab.py) runs with it switched off on the exact side.
#ifndef O2V_CERTAIN_N
#define O2V_CERTAIN_N 5
#define O2V_CERTAIN_STEP 0.22f
#endif
struct CertainPrep {
    float m00, m01, m10, m11, tau;
    uint32_t axis;  // dominant axis of the normal; tau = +inf: the leaf has no certain hits
};
__device__ __forceinline__ float certain_margin(float m) { return 1.5f * sat_margin(m); }
__device__ __forceinline__ CertainPrep certain_prepare(V3 v0, V3 v1, V3 v2, V3 nrm, float m, bool small, float parent_area)
{
    CertainPrep c{0.f, 0.f, 0.f, 0.f, __builtin_inff(), 0u};
    const float ax = abs_f(nrm.x), ay = abs_f(nrm.y), az = abs_f(nrm.z);
    c.axis = ax >= ay && ax >= az ? 0u : (ay >= az ? 1u : 2u);
    const uint32_t iu = c.axis == 0u ? 1u : 0u, iw = c.axis == 2u ? 1u : 2u;  // the two other axes
    const V3 e1 = v1 - v0, e2 = v2 - v0, e3 = v2 - v1;
    const float e1u = comp(e1, iu), e1w = comp(e1, iw), e2u = comp(e2, iu), e2w = comp(e2, iw);
    const float det = e1u * e2w - e1w * e2u;
    const V3 cr = cross(e1, e2);
    const float twice_area = __builtin_sqrtf(cr.x * cr.x + cr.y * cr.y + cr.z * cr.z);
    const float longest = __builtin_sqrtf(fmaxf(fmaxf(e1.x * e1.x + e1.y * e1.y + e1.z * e1.z, e2.x * e2.x + e2.y * e2.y + e2.z * e2.z),
                                                e3.x * e3.x + e3.y * e3.y + e3.z * e3.z));
    // (all comparisons false for NaN: such a leaf keeps tau = +inf)
    if (small && parent_area > 0.f && twice_area > 1e-12f && abs_f(det) > 1e-12f && longest > 0.f) {
        const float h_min = twice_area / longest;
        const float inv = 1.0f / det;
        c.m00 = e2w * inv;
        c.m01 = -e2u * inv;
        c.m10 = -e1w * inv;
        c.m11 = e1u * inv;
        c.tau = certain_margin(m) / h_min + 1e-4f;
    }
    return c;
}

#ifndef O2V_FLUSH_AT
#define O2V_FLUSH_AT 48
#endif
constexpr uint32_t kFlushAt = O2V_FLUSH_AT;             // parked hits per wavefront that trigger the append section
constexpr uint32_t kLeafStride = 25;          // dwords per staged leaf in LDS (24 + 1 pad: spreads banks)
// Launch shape of k_voxelize.  Without uv: 256-thread workgroups (four wavefronts share a leaf table and a job queue).  With
// uv: one wavefront per workgroup with its own table and queue (no workgroup barriers; measured -8 % on configs[3], -5 % on
// configs[1]; without uv the same change costs 4 %).  A workgroup stages at most one tile per thread and its queue holds
// 64 candidates per thread; it takes its tiles from the cursor in batches - about `batches` per workgroup, `batches_large`
// for large jobs (if a batch then still holds `finer_min` tiles): few large batches leave workgroups idle at the end of the
// kernel, many small ones pay the per-batch staging, barriers and tails too often.
#ifndef O2V_VOX_BLOCK
#define O2V_VOX_BLOCK 256
#endif
#ifndef O2V_VOX_BLOCK_UV
#define O2V_VOX_BLOCK_UV 64
#endif
#ifndef O2V_HEAVY_PLANES
#define O2V_HEAVY_PLANES 5
#endif
template <bool UV>
struct VoxShape {
    static constexpr uint32_t block = UV ? O2V_VOX_BLOCK_UV : O2V_VOX_BLOCK;  // threads per workgroup
    static constexpr uint32_t tiles = block;                                  // tiles staged at once (at most)
    static constexpr uint32_t queue = 64u * block;  // job queue records (= candidate voxels at most) per sub-batch and workgroup
#ifndef O2V_BATCHES
#define O2V_BATCHES 4
#define O2V_BATCHES_LARGE 6
#endif
#ifndef O2V_BATCHES_UV
#define O2V_BATCHES_UV 2
#define O2V_BATCHES_LARGE_UV 3
#endif
    static constexpr uint32_t batches = block >= 256u ? O2V_BATCHES : O2V_BATCHES_UV, batches_large = block >= 256u ? O2V_BATCHES_LARGE : O2V_BATCHES_LARGE_UV;
    static constexpr uint32_t finer_min = 32u * block / 256u;
};
constexpr uint32_t kHeavyPlanes = O2V_HEAVY_PLANES;          // a job whose leaf straddles at least this many voxel planes is queued first

// K2.  Persistent workgroups pull batches of tiles.  Per batch:
//   phase 1  the tiles' candidate rows: the separating-axis test solved for x per row (row_span), the rows' survivors
//            flattened over the lanes, plane-distance cull (voxelization.cpp:451-458); survivors become 8-byte job
//            records (position, tile slot, planes the leaf straddles) in the workgroup's queue in global memory, jobs
//            that straddle many planes (the long ones) first
//   phase 2  persistent lanes fetch their next job one ahead and run computeTrianglesUvInVoxel (voxelization.cpp:383-424) as a
//            depth-first walk of the split tree: the reference clips level by level with two 64-entry buffers;
//            visiting the first emitted piece first reproduces its buffer order, so the running mean of
//            :414-420 accumulates in the identical sequence.  Under DISCARD every split keeps <= 2 pieces, so at
//            most one sibling per level 1..5 is pending (register stack with a scratch overflow).  Every piece carries
//            the set of planes it does not pass whole (piece_masks); per iteration a lane classifies its piece against
//            the first of them and cuts it, and the kept pieces are judged at once from their bounding boxes: final
//            (accumulated), beyond a later plane (dropped with its whole subtree), one plane left (without uv: settled
//            from its classification, see single_plane), or to be cut again.  So lanes spend their iterations only on
//            cuts whose pieces are needed (3.8 per voxel job on the bench mesh; 5.9 before the single-plane rule, 8.4
//            events before the masks); lanes that run out of pieces pop the next survivor, so the wavefront stays full.
// Register budget: 4 waves per SIMD for both variants (125 VGPRs without uv arithmetic; with it the allocator spills two
// dozen cold values - measured faster than 3 waves without spills on the large textured workloads: configs[3] 31.5 ->
// 30.1 ms).  5 waves without uv (96 VGPRs, 16 spilled) measured the same as 4.
#ifndef O2V_K2_WAVES
#define O2V_K2_WAVES 4
#endif
#ifndef O2V_K2_WAVES_UV
#define O2V_K2_WAVES_UV 4
#endif
// exclusive scan of one uint32 per thread over k_voxelize's workgroup; returns the total in `total`
template <uint32_t kVoxBlock>
__device__ __forceinline__ uint32_t vox_exscan(uint32_t v, uint32_t *s_wave /*[kVoxBlock / 64]*/, uint32_t &total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v);
    __syncthreads();
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < kVoxBlock / 64; ++w) {
        const uint32_t cnt = s_wave[w];
        if (w < wave) base += cnt;
        tot += cnt;
    }
    total = tot;
    return base + inc - v;
}

// (UV: the mesh has textured triangles - pieces carry uv coordinates.  OCC: occupancy-only mode, Params::occupancy_only -
// its own kernel, k_voxelize_occ, so that the weighted routes' code is not in it and vice versa.)
template <bool UV, bool OCC>
__device__ __forceinline__ void voxelize_body(const Leaf *__restrict__ leaves, const Tile *__restrict__ tiles,
                                              Counters *c, uint32_t *grid, uint8_t *brick_dirty, HitRec *pool,
                                              uint2 *jobq_all, const Params &p, const float *__restrict__ verts = nullptr,
                                              const uint32_t *__restrict__ block_list = nullptr, const uint32_t *block_count = nullptr)
{
    static_assert(!(UV && OCC), "occupancy-only mode has no uv arithmetic");
    constexpr uint32_t kVoxBlock = VoxShape<UV>::block, kVoxTiles = VoxShape<UV>::tiles, kQueueCap = VoxShape<UV>::queue;
    // (occupancy-only mode: most candidates are settled in phase 1, so a batch leaves few voxel jobs - half as many, larger
    // batches keep phase 2's lanes busier: bench mesh 0.54 -> 0.50 ms)
    constexpr uint32_t kBatchesPerBlock = OCC ? (VoxShape<UV>::batches + 1u) / 2u : VoxShape<UV>::batches;
    constexpr uint32_t kBatchesPerBlockLarge = OCC ? (VoxShape<UV>::batches_large + 1u) / 2u : VoxShape<UV>::batches_large;
    constexpr uint32_t kFinerBatchMinTiles = VoxShape<UV>::finer_min;
    __shared__ uint32_t s_leaf[kVoxTiles * kLeafStride];
    __shared__ uint32_t s_tleaf[kVoxTiles];
    __shared__ uint32_t s_tstart[kVoxTiles];
    __shared__ uint32_t s_tcount[kVoxTiles];
    __shared__ uint32_t s_tprefix[kVoxTiles + 6];  // + total + padding for the four-entry window of phase 1
    __shared__ uint32_t s_scan[kVoxBlock / 64];
    __shared__ uint32_t s_tend;
    __shared__ uint8_t s_chunk_tile[kQueueCap / 64 + 4];  // tile slot (< 256) of each 64-candidate chunk's first candidate
    __shared__ float s_inv_dy[kVoxTiles];
    __shared__ float s_mcoord[kVoxTiles];  // the largest |coordinate| of the tile's leaf: out_margin (piece_masks) and sat_margin (row_span) follow from it
    // Survivors of the candidate rows wait here, per wavefront, until 64 of them are together (phase 1): {x in the leaf's box |
    // tile slot << 16, y | z << 16}
    constexpr uint32_t kRing = 128;
    __shared__ uint2 s_ring[(kVoxBlock / 64) * kRing];
    __shared__ uint32_t s_trow0[kVoxTiles];       // first row (y + dy z of the leaf's AABB) the tile's candidates lie in
    __shared__ uint32_t s_rprefix[kVoxTiles + 6];  // rows before tile k (+ total + padding, as s_tprefix)
    __shared__ uint32_t s_batch, s_nheavy, s_nlight, s_next, s_hits, s_direct, s_certain, s_nlive, s_skipped;
    // The job queue of this workgroup lives in global memory (it stays in L2): one 8-byte record per surviving candidate,
    // {x | y << 16, z | tile slot << 16 | plane mask << 24 | small << 30}.  Jobs whose leaf straddles many planes of their
    // voxel (the long ones) are filed from the front, the others from the back, and the queue is served front to back:
    // longest jobs first keeps the end of a sub-batch, when lanes run out of work, short.
    uint2 *jobq = jobq_all + (size_t) blockIdx.x * (OCC ? 2u * kQueueCap : kQueueCap);
    __shared__ uint32_t s_next_x[kVoxBlock], s_next_y[kVoxBlock];  // every lane's prefetched next job record (take_job)
    __shared__ unsigned long long s_solo[OCC ? 3 : 1];  // Params::solo_roots: this workgroup's root leaves, their candidates, the squares
    __shared__ uint8_t s_cls[64];    // classify_flags
    __shared__ uint8_t s_kept[128];  // classify_kept: index | keep_lo << 6

    if (expand_overflowed(c, p)) return;
    const bool use_direct = OCC || (direct_active(c, p) && (!UV || p.pick_max));
    // occupancy-only mode (Params::occupancy_only): a job is decided by its first surviving piece - splitTriangle only
    // ever adds the leaf's area per surviving piece (voxelization.cpp:414-420), so the weight is non-zero from then on
    constexpr bool occ_only = OCC;
    const uint32_t n_tiles = c->n_tiles < p.cap_tiles ? c->n_tiles : p.cap_tiles;
    // Batch size: about kBatchesPerBlock batches per workgroup (VoxShape), between one tile per wavefront and what the LDS staging holds.  Few
    // large batches leave workgroups idle at the end of the kernel (and a 96^3 job, a few thousand tiles, would keep 3 %
    // of the machine busy); many small ones pay the per-batch staging and barriers too often.  Measured on seven
    // workload shapes.
    // Params::root_bypass (occupancy only): the root triangles that are one leaf of one tile have no Leaf / Tile records;
    // they are staged from the vertex array, a run of consecutive triangles per batch ("root batches", behind the batches of
    // the tile list).  The sequence of blocks of 256 triangles is the one k_expand_roots walks: the listed blocks
    // (k_list_blocks: those whose z extent meets the slab) or all of them.
    const bool root_bypass = OCC && p.root_bypass != 0u;
    const bool blocks_listed = root_bypass && block_list != nullptr && *block_count != 0xffffffffu;
    const uint64_t n_seq_blocks = !root_bypass ? 0ull : (blocks_listed ? (uint64_t) *block_count : (p.n_tris + kTilesPerBatch - 1u) / kTilesPerBatch);
    const uint64_t n_work = (uint64_t) n_tiles + n_seq_blocks * kTilesPerBatch;  // tiles + triangles that may become one
    uint32_t tiles_per_batch = (uint32_t) std::min<uint64_t>((n_work + gridDim.x * kBatchesPerBlock - 1u) / (gridDim.x * kBatchesPerBlock), kVoxTiles);
    {
        // large jobs: half as many tiles again per workgroup's share, as long as a batch still fills the workgroup's lanes
        // twice over (shorter tail at the end of the kernel; measured -2 % on the bench mesh, -7 % on the low-poly sphere)
        const uint64_t finer = (n_work + gridDim.x * kBatchesPerBlockLarge - 1u) / (gridDim.x * kBatchesPerBlockLarge);
        if (finer >= kFinerBatchMinTiles) tiles_per_batch = (uint32_t) std::min<uint64_t>(finer, kVoxTiles);
    }
    tiles_per_batch = tiles_per_batch < kMinTilesPerBatch ? kMinTilesPerBatch
                      : (tiles_per_batch > kVoxTiles ? kVoxTiles : tiles_per_batch);
    // The kernel ends when its last batch does, and a workgroup gets only a few batches: where the batches' costs differ a lot -
    // leaves of very unequal sizes - the last ones decide how long most of the machine waits.  The batches of the last quarter
    // of the work are then a quarter as large (measured on the irregular bench mesh, leaf areas 380 : 1: k_voxelize<false> 1.13 ->
    // 0.92 ms; on a uniform tessellation the smaller batches only cost their fixed part again: 0.94 -> 1.01 ms, occupancy only 0.38
    // -> 0.45 - so it depends on the spread of the leaves' candidate counts, which k_expand_* leave in the counters).  The
    // one-wavefront workgroups of the uv variant always do it (their batches are a quarter as large to begin with; uniform
    // sphere 1.59 -> 1.53 ms, configs[3] 14.1 -> 13.8).
    // A work list of n items (the tile list; the sequence of root triangles) is cut into `full` batches of `size` items and,
    // behind them, batches of `small` items.
    bool taper = UV;
    {
        const double n_l = (double) c->n_leaves + (double) c->n_bypass, sum = (double) c->n_candidates, sq = (double) c->n_candidates_sq;
        // (variance of the leaves' candidate counts against their squared mean: above 1/2 the leaves are "very unequal")
        // (Params::solo_roots: these are counted by this kernel itself, so they are not read here - every workgroup must make the same plan)
        if (!(OCC && p.solo_roots) && n_l > 0.0 && sq * n_l > 1.5 * sum * sum) taper = true;
    }
#ifdef O2V_NO_TAPER
    taper = false;
#endif
    struct BatchPlan {
        uint64_t n;
        uint32_t size, small, full;
        uint64_t count;  // batches in all
        __device__ __forceinline__ void make(uint64_t items, uint32_t batch_size, bool taper)
        {
            n = items;
            size = batch_size;
            small = (taper && batch_size >= 4u * kMinTilesPerBatch) ? batch_size / 4u : batch_size;
            full = taper ? (uint32_t) std::min<uint64_t>((items - items / 4u) / batch_size, 0x7fffffffull) : 0u;
            const uint64_t rest = items - (uint64_t) full * size;
            count = full + (rest + small - 1u) / small;
        }
        // batch b: its first item and the number of its items
        __device__ __forceinline__ uint64_t at(uint64_t b, uint32_t &items) const
        {
            const uint64_t start = b < full ? b * size : (uint64_t) full * size + (b - full) * small;
            const uint32_t want = b < full ? size : small;
            items = start >= n ? 0u : (uint32_t) std::min<uint64_t>(want, n - start);
            return start;
        }
    };
    BatchPlan list_plan, root_plan;
    list_plan.make(n_tiles, tiles_per_batch, taper);
    static_assert(kTilesPerBatch == 256u && kBlock == 256u, "blocks of 256 triangles");
    // (the root triangles: the positions 0 .. 256 n_seq_blocks of the block sequence; the last block may be partly filled)
    root_plan.make(n_seq_blocks * kTilesPerBatch, tiles_per_batch, taper);
    const uint32_t n_list_batches = (uint32_t) std::min<uint64_t>(list_plan.count, 0xffffffffull);
    const uint32_t n_batches = (uint32_t) std::min<uint64_t>(list_plan.count + root_plan.count, 0xffffffffull);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t chunk_base = 0, chunk_used = kHitChunk;  // wave-uniform; forces a reservation at first use
#ifdef O2V_INSTRUMENT
    uint32_t dbgc[16] = {};
    unsigned long long t_drain = 0;  // cycles inside file_survivor (of this wavefront), calls, survivors
    uint32_t n_drain = 0, n_drain_lanes = 0;
    unsigned long long tmr[4] = {0, 0, 0, 0};  // cycles: staging + phase 1 | phase 2 loop | waiting at the barrier after phase 2 | whole kernel
    unsigned long long t_mark = __builtin_readcyclecounter();
    const unsigned long long t_kernel0 = t_mark;
    auto lap = [&](int which) {
        const unsigned long long now = __builtin_readcyclecounter();
        tmr[which] += now - t_mark;
        t_mark = now;
    };
#define O2V_LAP(i) lap(i)
#else
#define O2V_LAP(i) do { } while (0)
#endif
    if (threadIdx.x == 0) {
        s_hits = 0;
        s_direct = 0;
        s_certain = 0;
        s_skipped = 0;
    }
    if (OCC && threadIdx.x < 3u) s_solo[threadIdx.x] = 0ull;
    if (threadIdx.x < 64u) s_cls[threadIdx.x] = (uint8_t) classify_flags(threadIdx.x);
    for (uint32_t i = threadIdx.x; i < 128u; i += kVoxBlock) s_kept[i] = (uint8_t) classify_kept(i & 63u, i >= 64u);

    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_batch = atomicAdd(&c->batch_cursor, 1u);
        __syncthreads();
        const uint32_t batch = s_batch;
        if (batch >= n_batches) break;
        uint32_t nt;
        if (OCC && batch >= n_list_batches) {
            // a root batch: the leaves are made here, from the vertex array, as k_expand_roots makes them (applyMeshTransform,
            // obj2voxel.cpp:202-224; the head of voxelizeTriangleToUvBuffer, voxelization.cpp:488-511; write_leaf); a triangle
            // that is not a leaf of one tile (k_expand_roots handled it, or it misses the slab) leaves its tile slot empty
            // (a run of positions of the block sequence: consecutive triangles, except where the run crosses from one listed
            // block into the next)
            const uint64_t pos0 = root_plan.at(batch - n_list_batches, nt);
            const uint64_t pos = pos0 + threadIdx.x;
            // (a thread past the batch's items reads nothing: the last batch's positions may lie past the list's end)
            const bool in_batch = threadIdx.x < nt;
            const uint64_t tri = !in_batch ? ~0ull : blocks_listed ? (uint64_t) block_list[pos >> 8] * kTilesPerBatch + (pos & 255u) : pos;
            const bool have = in_batch && tri < p.n_tris;
            float q[9];
#pragma unroll
            for (uint32_t j = 0; j < 9; ++j) q[j] = have ? verts[tri * 9u + j] : 0.f;
            bool solo_leaf = false, solo_other = false;
            uint32_t solo_cnt = 0;
            if (threadIdx.x < nt) {
                Affine xf;
                xf.m[0] = {c->xform[0], c->xform[1], c->xform[2]};
                xf.m[1] = {c->xform[3], c->xform[4], c->xform[5]};
                xf.m[2] = {c->xform[6], c->xform[7], c->xform[8]};
                xf.t = {c->xform[9], c->xform[10], c->xform[11]};
                Sub sb{};
                sb.v0 = affine_apply(xf, V3{q[0], q[1], q[2]});
                sb.v1 = affine_apply(xf, V3{q[3], q[4], q[5]});
                sb.v2 = affine_apply(xf, V3{q[6], q[7], q[8]});
                LeafPlan pl{};
                bool other = false;
                const bool is_leaf = have && root_leaf_of_one_tile(sb, p, pl, other);
                solo_leaf = is_leaf;
                solo_other = have && other;
                solo_cnt = is_leaf ? (uint32_t) pl.count : 0u;   // (one tile: at most kTileSize)
                uint32_t *lw = &s_leaf[threadIdx.x * kLeafStride];
                const V3 nrm = normalize(tri_normal(sb.v0, sb.v1, sb.v2));  // voxelization.cpp:438
                const float vals[12] = {sb.v0.x, sb.v0.y, sb.v0.z, sb.v1.x, sb.v1.y, sb.v1.z, sb.v2.x, sb.v2.y, sb.v2.z, nrm.x, nrm.y, nrm.z};
#pragma unroll
                for (uint32_t j = 0; j < 12; ++j) lw[j] = is_leaf ? __float_as_uint(vals[j]) : 0u;
                lw[18] = (uint32_t) tri;
                lw[19] = 0u;  // order key of an unsplit triangle
                lw[20] = is_leaf ? pl.lo[0] | (pl.lo[1] << 16) : 0u;
                lw[21] = is_leaf ? pl.lo[2] | (pl.d[0] << 16) : 0u;   // (an empty slot: a box of no cells)
                lw[22] = is_leaf ? pl.d[1] | (pl.d[2] << 16) : 0u;
                lw[23] = __float_as_uint(is_leaf ? tri_area(sb.v0, sb.v1, sb.v2) : 0.f);
                s_tstart[threadIdx.x] = 0u;
            }
            if (p.solo_roots) {
                // no k_expand_roots ran: its counts of the root leaves (statistics; the batches' sizes) are made here, and a
                // triangle that is its business after all voids the pass (o2v_hip_voxelize repeats it with k_expand_roots)
                const unsigned long long ml = __ballot(solo_leaf);
                if (__ballot(solo_other) && lane == 0) atomicOr(&c->err_flags, kErrSoloRoots);
                if (ml) {
                    const uint32_t cand = __shfl(wave_inclusive_scan(solo_cnt), 63, 64), sq = __shfl(wave_inclusive_scan(solo_cnt * solo_cnt), 63, 64);
                    if (lane == 0) {
                        atomicAdd(&s_solo[0], (unsigned long long) __popcll(ml));
                        atomicAdd(&s_solo[1], (unsigned long long) cand);
                        atomicAdd(&s_solo[2], (unsigned long long) sq);
                    }
                }
            }
            __syncthreads();
        }
        else {
            const uint32_t first = (uint32_t) list_plan.at(batch, nt);
            if (threadIdx.x < nt) {
                const Tile t = tiles[first + threadIdx.x];
                s_tleaf[threadIdx.x] = t.leaf;
                s_tstart[threadIdx.x] = t.start;
            }
            __syncthreads();
            // stage the leaves of this batch in LDS
            for (uint32_t i = threadIdx.x; i < nt * 24u; i += kVoxBlock) {
                const uint32_t k = i / 24u, j = i - k * 24u;
                s_leaf[k * kLeafStride + j] = reinterpret_cast<const uint32_t *>(leaves + s_tleaf[k])[j];
            }
            __syncthreads();
        }
        uint32_t my_count = 0, my_rows = 0;
        if (threadIdx.x < nt) {
            const uint32_t *lf = &s_leaf[threadIdx.x * kLeafStride];
            const uint32_t dx = lf[21] >> 16, dy = lf[22] & 0xffffu, dz = lf[22] >> 16;
            const uint32_t rem = dx * dy * dz - s_tstart[threadIdx.x];
            my_count = rem < kTileSize ? rem : kTileSize;
            // bit 31: every coordinate of the leaf is finite and below kSmallCoord (see piece_masks)
            float m;
            // (exact mode, O2V_HIP_FLAG_EXACT_CLIP: no leaf is `small`, so neither row_span nor the piece masks are used)
            const bool is_small = leaf_is_small(lf, m) && !p.exact_clip;
            // bit 30: the job's divisions may take the lean forms (accumulate_piece); without uv only the cut parameter is
            // divided, for which `small` is all that is needed
            bool lean_ok = is_small;
            if (UV) {
                const float a = __uint_as_float(lf[23]);
                float uvmax = 0.f;
#pragma unroll
                for (int i = 12; i < 18; ++i) uvmax = fmaxf(uvmax, abs_f(__uint_as_float(lf[i])));
                float uvsum = 0.f;
#pragma unroll
                for (int i = 12; i < 18; ++i) uvsum += __uint_as_float(lf[i]);  // NaN if any uv is NaN (fmaxf ignores NaN operands)
                lean_ok = lean_ok && a >= kLeanAreaMin && a <= kLeanAreaMax && uvmax <= kLeanUv && uvsum == uvsum;
            }
            s_tcount[threadIdx.x] = my_count | (is_small ? 0x80000000u : 0u) | (lean_ok ? 0x40000000u : 0u);
            s_mcoord[threadIdx.x] = m;
            if (OCC) {
                // the certain-hit test's per-leaf part goes where the staged leaf keeps its uv coordinates (unused without uv)
                uint32_t *lw = &s_leaf[threadIdx.x * kLeafStride];
                const CertainPrep cp = certain_prepare(V3{__uint_as_float(lw[0]), __uint_as_float(lw[1]), __uint_as_float(lw[2])},
                                                       V3{__uint_as_float(lw[3]), __uint_as_float(lw[4]), __uint_as_float(lw[5])},
                                                       V3{__uint_as_float(lw[6]), __uint_as_float(lw[7]), __uint_as_float(lw[8])},
                                                       V3{__uint_as_float(lw[9]), __uint_as_float(lw[10]), __uint_as_float(lw[11])}, m, is_small,
                                                       __uint_as_float(lw[23]));
                lw[12] = __float_as_uint(cp.m00);
                lw[13] = __float_as_uint(cp.m01);
                lw[14] = __float_as_uint(cp.m10);
                lw[15] = __float_as_uint(cp.m11);
                lw[16] = __float_as_uint(cp.tau);
                lw[17] = cp.axis;
            }
            s_inv_dy[threadIdx.x] = 1.0f / (float) dy;
            if (my_count) {
                const uint32_t start = s_tstart[threadIdx.x];
                const uint32_t r0 = start / dx;
                s_trow0[threadIdx.x] = r0;
                my_rows = (start + my_count - 1u) / dx - r0 + 1u;
            }
        }
        {
            // exclusive prefix of the tile sizes: s_tprefix[k] = candidates before tile k, s_tprefix[nt] = total
            uint32_t total;
            const uint32_t ex = vox_exscan<kVoxBlock>(my_count, s_scan, total);
            if (threadIdx.x < nt) s_tprefix[threadIdx.x] = ex;
            if (threadIdx.x == 0) s_tprefix[nt] = total;  // (nt may equal the number of threads)
            if (threadIdx.x < 5u) s_tprefix[nt + 1u + threadIdx.x] = 0xffffffffu;  // never <= a candidate index
            // the same for the tiles' rows
            __syncthreads();
            const uint32_t exr = vox_exscan<kVoxBlock>(my_rows, s_scan, total);
            if (threadIdx.x < nt) s_rprefix[threadIdx.x] = exr;
            if (threadIdx.x == 0) s_rprefix[nt] = total;
            if (threadIdx.x < 5u) s_rprefix[nt + 1u + threadIdx.x] = 0xffffffffu;
        }

        // sub-batches of whole tiles with at most kQueueCap candidates
        uint32_t t_begin = 0;
        while (t_begin < nt) {
            __syncthreads();
            const uint32_t base_cand = s_tprefix[t_begin];
            // the last tile whose end still fits decides t_end (found by the thread that owns it)
            if (threadIdx.x >= t_begin && threadIdx.x < nt) {
                const bool fits = s_tprefix[threadIdx.x + 1] - base_cand <= kQueueCap;
                const bool next_fits = threadIdx.x + 1 < nt && s_tprefix[threadIdx.x + 2] - base_cand <= kQueueCap;
                if (fits && !next_fits) s_tend = threadIdx.x + 1;
            }
            if (threadIdx.x == 0) {
                s_nheavy = 0;
                s_nlight = 0;
                s_next = 0;
            }
            __syncthreads();
            const uint32_t t_end = s_tend;

            // ---- phase 1: the sub-batch's candidate rows, their survivors packed 64 to a wavefront ------------------
            // A tile's candidates are a run of the leaf's clamped AABB in x-fastest order, i.e. a few rows (fixed y, z) of it.
            // A lane takes one row, solves the separating-axis test for x (row_span) and so names the row's surviving
            // voxels without visiting the others.  The survivors go into the wavefront's ring in LDS and are taken out 64 at a
            // time (file_survivor: the reference's plane cull, the certain-hit test, the job record), so that the expensive
            // per-voxel part always runs with every lane - a 64-row chunk of a tessellated surface leaves ~80 survivors, i.e.
            // one full and one quarter-full pass when each chunk was flattened on its own (round 4).
            // Row g (in sub-batch order) belongs to the tile k with s_rprefix[k] <= g < s_rprefix[k + 1]; a small table
            // gives every 64-row chunk the tile its first row falls in and a lane walks forward a few tiles at most.
            const uint32_t base_row = s_rprefix[t_begin];
            const uint32_t n_rows = s_rprefix[t_end] - base_row;
            if (threadIdx.x >= t_begin && threadIdx.x < t_end) {
                const uint32_t lo = s_rprefix[threadIdx.x] - base_row, hi = s_rprefix[threadIdx.x + 1] - base_row;
                if (hi > lo)
                    for (uint32_t ch = (lo + 63u) / 64u; ch * 64u < hi; ++ch) s_chunk_tile[ch] = (uint8_t) threadIdx.x;
            }
            __syncthreads();
            O2V_LAP(0);

            // One survivor per lane (wave-uniform call; `valid`: the lane has one): voxel (lx, ly, lz) of tile slot kk's box.
            auto file_survivor = [&](bool valid, uint32_t kk, uint32_t lx, uint32_t ly, uint32_t lz) {
                bool keep = false, heavy = false, certain = false;
                uint2 rec = make_uint2(0u, 0u);
                if (valid) {
                    const uint32_t *lf = &s_leaf[kk * kLeafStride];
                    // (qx, qy, qz: the voxel relative to the grid's origin - what the 16-bit fields carry; its position adds Params::so)
                    const uint32_t qx = (lf[20] & 0xffffu) + lx, qy = (lf[20] >> 16) + ly, qz = (lf[21] & 0xffffu) + lz;
                    const float px = (float) (qx + p.so[0]), py = (float) (qy + p.so[1]), pz = (float) (qz + p.so[2]);
                    const V3 v0{__uint_as_float(lf[0]), __uint_as_float(lf[1]), __uint_as_float(lf[2])};
                    const V3 v1{__uint_as_float(lf[3]), __uint_as_float(lf[4]), __uint_as_float(lf[5])};
                    const V3 v2{__uint_as_float(lf[6]), __uint_as_float(lf[7]), __uint_as_float(lf[8])};
                    const V3 nrm{__uint_as_float(lf[9]), __uint_as_float(lf[10]), __uint_as_float(lf[11])};
                    // plane distance cull, voxelization.cpp:451-458
                    const V3 rel = V3{px + 0.5f, py + 0.5f, pz + 0.5f} - v0;
                    const float sd = dot(nrm, rel);
                    keep = !(abs_f(sd) > kPlaneDistanceLimit);
                    if (OCC && keep) {
                        // certain hit (see certain_prepare): no voxel job, the voxel is marked here.  Tried are the columns
                        // through a 5 x 5 lattice of points around the voxel centre (every quantity of the test is linear in
                        // the offset); the lattice stays inside the voxel by the margin D.
                        const uint32_t axis = lf[17];
                        const uint32_t iu = axis == 0u ? 1u : 0u, iw = axis == 2u ? 1u : 2u;
                        const float tau = __uint_as_float(lf[16]);
                        if (tau < 0.34f) {  // (else no point of the leaf is far enough from its edges)
                            const float lim = 0.5f - (1.5f * sat_margin(s_mcoord[kk]) + 1e-3f);  // 0.5 - D, D = certain_margin(m)
                            // kN lines across the voxel (constant offset in the second lattice axis); along a line every
                            // condition is a linear inequality in the offset o: they leave an interval, the point exists if it is
                            // not empty (by a slack that absorbs this code's own rounding)
                            constexpr int kN = O2V_CERTAIN_N;
                            const float half = 0.5f * (float) (kN - 1);
                            const float step = fminf(O2V_CERTAIN_STEP, lim / fmaxf(half, 0.5f));
                            const float m00 = __uint_as_float(lf[12]), m01 = __uint_as_float(lf[13]), m10 = __uint_as_float(lf[14]), m11 = __uint_as_float(lf[15]);
                            const float rnd = __builtin_amdgcn_rcpf(comp(nrm, axis));
                            const float tu = comp(nrm, iu) * rnd, tw = comp(nrm, iw) * rnd * step;  // dt per unit of o / per line
                            const float pu = comp(rel, iu), pw = comp(rel, iw) - half * step;       // the first line's centre
                            float l1 = m00 * pu + m01 * pw, l2 = m10 * pu + m11 * pw, t0 = sd * rnd - half * tw;
                            const float d1w = m01 * step, d2w = m11 * step;
                            // a * o >= b  ->  o >= b / a (a > 0), o <= b / a (a < 0), or b <= 0 (a = 0)
                            const float a1 = m00, a2 = m10, a0 = -(m00 + m10), a3 = tu, a4 = -tu;
                            auto inv = [](float a) { return abs_f(a) > 1e-12f ? __builtin_amdgcn_rcpf(a) : 0.f; };
                            const float r1 = inv(a1), r2 = inv(a2), r0 = inv(a0), r3 = inv(a3), r4 = -r3;
#pragma unroll
                            for (int jw = 0; jw < kN; ++jw) {
                                float lo = -lim, hi = lim;
                                bool ok = true;
                                auto bound = [&](float a, float ra, float bb) {
                                    const float x = bb * ra;
                                    lo = (ra > 0.f) ? fmaxf(lo, x) : lo;
                                    hi = (ra < 0.f) ? fminf(hi, x) : hi;
                                    ok = ok && (ra != 0.f || bb <= 0.f);
                                };
                                bound(a1, r1, tau - l1);
                                bound(a2, r2, tau - l2);
                                bound(a0, r0, tau - ((1.0f - l1) - l2));
                                bound(a3, r3, -lim - t0);
                                bound(a4, r4, t0 - lim);
                                certain |= ok && hi - lo >= 2e-3f;
                                l1 += d1w;
                                l2 += d2w;
                                t0 += tw;
                            }
                        }
                        if (certain) {
                            keep = false;
                            const uint32_t ox = qx >> p.ss_shift, oy = qy >> p.ss_shift, oz = qz >> p.ss_shift;
                            uint32_t brick;
                            const uint64_t cell = cell_index_rel(ox, oy, oz, p, brick);
                            p.occgrid[cell] = 1;      // (plain stores; benign races: every writer stores the same value)
                            p.dirty_max[brick] = 1;
                        }
                    }
                    if (keep) {
                        // the job record: position, tile slot, the planes of this voxel the leaf does not pass whole
                        const bool small = (s_tcount[kk] >> 31) != 0u;
                        Piece<false> leaf;
                        leaf.a = v0;
                        leaf.b = v1;
                        leaf.c = v2;
                        uint32_t cf0, out_unused, near_unused;
                        piece_masks<false>(leaf, px, py, pz, small, 0.f, 63u, cf0, out_unused, near_unused);
                        rec = make_uint2(qx | (qy << 16), qz | (kk << 16) | (cf0 << 24) | (small ? 1u << 30 : 0u) | ((s_tcount[kk] & 0x40000000u) << 1));
                        heavy = (uint32_t) __popc(cf0) >= kHeavyPlanes;
                    }
                }
                if (OCC) {
                    const unsigned long long mc = __ballot(certain);
                    if (mc && lane == 0) atomicAdd(&s_certain, (uint32_t) __popcll(mc));
                }
                const unsigned long long mh = __ballot(keep && heavy), ml = __ballot(keep && !heavy);
                if (mh | ml) {
                    uint32_t base_h = 0, base_l = 0;
                    if (lane == 0) {
                        if (mh) base_h = atomicAdd(&s_nheavy, (uint32_t) __popcll(mh));
                        if (ml) base_l = atomicAdd(&s_nlight, (uint32_t) __popcll(ml));
                    }
                    base_h = __shfl(base_h, 0, 64);
                    base_l = __shfl(base_l, 0, 64);
                    if (keep) {
                        const uint32_t at = heavy ? base_h + __builtin_amdgcn_mbcnt_hi((uint32_t) (mh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mh, 0u))
                                                  : kQueueCap - 1u - (base_l + __builtin_amdgcn_mbcnt_hi((uint32_t) (ml >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) ml, 0u)));
                        jobq[at] = rec;
                    }
                }
            };

            // Few long rows (large axis-aligned leaves): every wavefront looks at the same 64 rows and they share the
            // survivors, 64 at a time; otherwise each wavefront has its own rows and its own ring.
            const bool shared_rows = s_tprefix[t_end] - base_cand > 32u * n_rows;
            uint2 *ring = &s_ring[wave * kRing];
            uint32_t r_head = 0, r_count = 0;  // wave-uniform: the ring's entries are [r_head, r_head + r_count) modulo kRing
            // takes survivors out of the ring, 64 at a time, as long as it holds `at_least` (64; 1 for the last, partial group)
            auto ring_drain = [&](uint32_t at_least) {
                while (r_count >= at_least) {
                    const uint2 e = ring[(r_head + lane) & (kRing - 1u)];
#ifdef O2V_INSTRUMENT
                    const unsigned long long t_d0 = __builtin_readcyclecounter();
#endif
                    file_survivor(lane < r_count, e.x >> 16, e.x & 0xffffu, e.y & 0xffffu, e.y >> 16);
#ifdef O2V_INSTRUMENT
                    t_drain += __builtin_readcyclecounter() - t_d0;
                    n_drain += 1u;
                    n_drain_lanes += r_count < 64u ? r_count : 64u;
#endif
                    const uint32_t n_taken = r_count < 64u ? r_count : 64u;
                    r_head += n_taken;
                    r_count -= n_taken;
                }
            };
            // a row's survivors - voxels x_first .. x_first + n_out - 1 of the row yz of tile slot k (xk = x_first | k << 16) - go
            // into the ring, as many at a time as it has room for; full groups of 64 leave it (wave-uniform call)
            auto ring_push = [&](uint32_t n_out, uint32_t xk, uint32_t yz) {
                const uint32_t inc = wave_inclusive_scan(n_out);
                const uint32_t total = __shfl(inc, 63, 64), exc = inc - n_out;
                uint32_t pushed = 0;  // wave-uniform: survivors of these 64 rows (in prefix order) already in the ring
                while (pushed < total) {
                    const uint32_t room = kRing - r_count;
                    const uint32_t take = total - pushed < room ? total - pushed : room;
                    const uint32_t s0 = exc > pushed ? exc : pushed, s1 = exc + n_out < pushed + take ? exc + n_out : pushed + take;
                    for (uint32_t sq = s0; sq < s1; ++sq) ring[(r_head + r_count + (sq - pushed)) & (kRing - 1u)] = make_uint2(xk + (sq - exc), yz);
                    __builtin_amdgcn_wave_barrier();
                    r_count += take;
                    pushed += take;
                    ring_drain(64u);
                }
            };
            for (uint32_t g0 = shared_rows ? 0u : wave * 64u; g0 < n_rows; g0 += shared_rows ? 64u : kVoxBlock) {
                const uint32_t g = g0 + lane;
                uint32_t k = s_chunk_tile[g0 / 64u];
                uint32_t n_out = 0, x_first = 0, ly = 0, lz = 0;
                if (g < n_rows) {
                    const uint32_t e1 = s_rprefix[k + 1], e2 = s_rprefix[k + 2], e3 = s_rprefix[k + 3], e4 = s_rprefix[k + 4];
                    const uint32_t gg = g + base_row;
                    k += (e1 <= gg ? 1u : 0u) + (e2 <= gg ? 1u : 0u) + (e3 <= gg ? 1u : 0u) + (e4 <= gg ? 1u : 0u);
                    if (e4 <= gg)
                        while (s_rprefix[k + 1] <= gg) ++k;
                    const uint32_t i_row = gg - s_rprefix[k], last_row = s_rprefix[k + 1] - s_rprefix[k] - 1u;
                    const uint32_t *lf = &s_leaf[k * kLeafStride];
                    const uint32_t dx = lf[21] >> 16, dy = lf[22] & 0xffffu, dz = lf[22] >> 16;
                    const uint32_t row = s_trow0[k] + i_row;
                    if (row < (1u << 24)) {
                        // exact quotient from a float estimate (row < 2^24, divisor < 2^16): off by at most one
                        lz = (uint32_t) ((float) row * s_inv_dy[k]);
                        int32_t ry = (int32_t) (row - lz * dy);
                        if (ry < 0) { lz -= 1; ry += (int32_t) dy; }
                        else if ((uint32_t) ry >= dy) { lz += 1; ry -= (int32_t) dy; }
                        ly = (uint32_t) ry;
                    }
                    else {
                        lz = row / dy;
                        ly = row - lz * dy;
                    }
                    // the part of the row that belongs to this tile
                    const uint32_t start = s_tstart[k], count = s_tcount[k] & 0x3fffffffu;
                    const uint32_t xlo = i_row == 0u ? start - s_trow0[k] * dx : 0u;
                    const uint32_t xhi = i_row == last_row ? (start + count - 1u) - row * dx : dx - 1u;
                    x_first = xlo;
                    n_out = xhi - xlo + 1u;
                    if (s_tcount[k] >> 31) {
                        const float ox = (float) ((lf[20] & 0xffffu) + p.so[0]), oy = (float) ((lf[20] >> 16) + p.so[1]), oz = (float) ((lf[21] & 0xffffu) + p.so[2]);
                        const V3 p0{__uint_as_float(lf[0]) - ox, __uint_as_float(lf[1]) - oy, __uint_as_float(lf[2]) - oz};
                        const V3 p1{__uint_as_float(lf[3]) - ox, __uint_as_float(lf[4]) - oy, __uint_as_float(lf[5]) - oz};
                        const V3 p2{__uint_as_float(lf[6]) - ox, __uint_as_float(lf[7]) - oy, __uint_as_float(lf[8]) - oz};
                        n_out = row_span(p0, p1, p2, (float) ly + 0.5f, (float) lz + 0.5f, xlo, xhi, (float) (dx + dy + dz), sat_margin(s_mcoord[k]), x_first);
                    }
                }
                const uint32_t xk = x_first | (k << 16), yz = ly | (lz << 16);
                if (!shared_rows) {
                    ring_push(n_out, xk, yz);
                    continue;
                }
                // inclusive prefix of the rows' survivor counts over the wavefront
                const uint32_t inc = wave_inclusive_scan(n_out);
                const uint32_t total = __shfl(inc, 63, 64);
                const uint32_t exc = inc - n_out;
                // (long rows - large axis-aligned leaves - are followed with a wave-uniform cursor instead of the search:
                // a chunk of 64 survivors then usually comes from one row)
                const bool long_rows = total > 512u;
                uint32_t row_b = 0;  // wave-uniform: the row (lane) the survivor b belongs to
                for (uint32_t b = wave * 64u; b < total; b += kVoxBlock) {
                    const uint32_t o = b + lane;  // this lane's survivor
                    // its row: the first lane whose inclusive prefix exceeds o
                    uint32_t src = 0;
                    bool search = true;
                    if (long_rows) {
                        while (__builtin_amdgcn_readlane(inc, row_b) <= b) ++row_b;
                        src = row_b;
                        search = __builtin_amdgcn_readlane(inc, row_b) < (b + 64u < total ? b + 64u : total);
                    }
                    if (search) {
                        src = 0;
#pragma unroll
                        for (uint32_t step = 32; step >= 1; step >>= 1) {
                            const uint32_t probe = __shfl(inc, (int) (src + step - 1u), 64);
                            if (probe <= o) src += step;
                        }
                        src &= 63u;  // (lanes beyond the total look at lane 63; they are masked below)
                    }
                    const uint32_t r_exc = __shfl(exc, (int) src, 64), r_xk = __shfl(xk, (int) src, 64), r_yz = __shfl(yz, (int) src, 64);
                    file_survivor(o < total, r_xk >> 16, (r_xk & 0xffffu) + (o - r_exc), r_yz & 0xffffu, r_yz >> 16);
                }
            }
            ring_drain(1u);  // the last, partial group
            __syncthreads();  // (workgroup scope: the records written above are visible to every wavefront of the workgroup)
            uint32_t n_heavy = s_nheavy, n_surv = n_heavy + s_nlight;
            if (threadIdx.x == 0 && n_surv) atomicAdd(&c->n_jobs, (unsigned long long) n_surv);
            const uint2 *jobs_front = jobq, *jobs_back = jobq + (kQueueCap - 1u);  // job q: front[q] (q < n_heavy), back[-(q - n_heavy)]
#ifndef O2V_NO_JOB_FILTER
            if (OCC) {
                // Occupancy only: a job whose voxel is marked already cannot change the result (the result is the set of marked
                // voxels), and most jobs are such - a voxel that one triangle merely clips at a corner is, as a rule, a
                // certain hit of its neighbour, and the certain hits of this batch (and of the earlier ones) are all stored
                // by now.  The jobs are read once more, those whose cell still reads zero are compacted into the second half of
                // the workgroup's queue, and phase 2 serves that.  A stale zero (another XCD's L2 holds the byte) only
                // means the job runs although it need not: the set of marked voxels is the same either way.
                uint2 *live = jobq + kQueueCap;
                if (threadIdx.x == 0) s_nlive = 0;
                __syncthreads();
                constexpr uint32_t kPerLane = 4;
                for (uint32_t base = 0; base < n_surv; base += kVoxBlock * kPerLane) {
                    uint2 rec[kPerLane];
                    bool ok[kPerLane];
                    uint8_t seen[kPerLane];
#pragma unroll
                    for (uint32_t j = 0; j < kPerLane; ++j) {
                        const uint32_t q = base + j * kVoxBlock + threadIdx.x;
                        ok[j] = q < n_surv;
                        rec[j] = ok[j] ? (q < n_heavy ? jobs_front[q] : jobs_back[-(int32_t) (q - n_heavy)]) : make_uint2(0u, 0u);
                    }
#pragma unroll
                    for (uint32_t j = 0; j < kPerLane; ++j) {
                        uint32_t brick;
                        const uint64_t cell = cell_index_rel((rec[j].x & 0xffffu) >> p.ss_shift, (rec[j].x >> 16) >> p.ss_shift,
                                                             (rec[j].y & 0xffffu) >> p.ss_shift, p, brick);
                        seen[j] = ok[j] ? __hip_atomic_load(&p.occgrid[cell], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint8_t) 1;
                    }
#pragma unroll
                    for (uint32_t j = 0; j < kPerLane; ++j) {
                        const bool keep = ok[j] && seen[j] == 0;
                        const unsigned long long mk = __ballot(keep);
                        if (mk) {
                            uint32_t at = 0;
                            if (lane == 0) at = atomicAdd(&s_nlive, (uint32_t) __popcll(mk));
                            at = __shfl(at, 0, 64);
                            if (keep) live[at + __builtin_amdgcn_mbcnt_hi((uint32_t) (mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mk, 0u))] = rec[j];
                        }
                    }
                }
                __syncthreads();
                if (threadIdx.x == 0 && n_surv != s_nlive) s_skipped += n_surv - s_nlive;
                n_surv = s_nlive;
                n_heavy = n_surv;  // (the compacted list is served front to back)
                jobs_front = live;
            }
#endif

            O2V_LAP(2);
            // ---- phase 2: persistent lanes ------------------------------------------------------------------
            Piece<UV> cur{}, sec{};
            PieceStack<UV> stack{};
            Piece<UV> overflow[kStackLevels - stack_regs<UV>()];
            uint32_t sp = 0, my_k = 0;  // sp: pending siblings of this lane's job; my_k: its tile slot
            uint32_t cf = 0;     // planes the current piece does not pass whole (bit = level), see piece_masks
            uint32_t pmask = 0;  // the same for the pending siblings: 6 bits per stack entry
            bool active = false, has_job = false, small = false;
            bool lean = false;   // this job's divisions may take the lean forms (lean_ok at staging)
            float w = 0.f, u = 0.f, v = 0.f, area = 0.f;
            float fx = 0.f, fy = 0.f, fz = 0.f;  // float(pos): the lower planes; upper planes are +1
            float margin = 0.f;                  // out_margin of the job's leaf
            uint32_t pos_xy = 0, pos_zk = 0;  // voxel x | y << 16, z | tile slot << 16 (all below 2^16)
            // Every lane holds its next job one ahead: the record is requested from the queue (an LDS ticket, then a load
            // that L2 answers) when the lane starts a job and is only looked at when that job is done.  The load goes
            // straight into the lane's LDS slot (global_load_lds: no destination register, so nothing the register
            // allocator does can touch data in flight).  The wait before the slot is read is explicit: hipcc (ROCm 7.2) does
            // track LDS-DMA for its own s_waitcnt placement, but across the loop's back edge it put the wait in front of
            // the wrong LDS reads (seen in the compiled code: the slot was read first).  A stricter-than-needed vmcnt(0)
            // costs nothing here: the lane's only other loads in flight are those of a rare stack overflow.
            bool next_valid = false;
            auto take_job = [&]() {
                const uint32_t q = atomicAdd(&s_next, 1u);
                next_valid = q < n_surv;
                if (next_valid) {
                    const uint2 *src = q < n_heavy ? jobs_front + q : jobs_back - (q - n_heavy);
                    // (LDS address = the wave-uniform base + 4 x lane)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) &src->x,
                                                     (__attribute__((address_space(3))) void *) &s_next_x[threadIdx.x & ~63u], 4, 0, 0);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) &src->y,
                                                     (__attribute__((address_space(3))) void *) &s_next_y[threadIdx.x & ~63u], 4, 0, 0);
                }
            };
            auto job_record = [&]() {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                return make_uint2(s_next_x[threadIdx.x], s_next_y[threadIdx.x]);
            };
            take_job();
            // parked result of this lane's last finished hit
            float d_w = 0.f, d_u = 0.f, d_v = 0.f;
            uint32_t d_xy = 0, d_zk = 0;  // voxel x | y << 16, z | tile slot << 16 (all below 2^16)
            bool d_valid = false;
            auto flush_results = [&]() {
                const unsigned long long all = __ballot(d_valid);
                if (!all) return;
                if (OCC) {
                    // occupancy only: the voxel is hit, nothing else about it matters (plain stores; benign races: every writer
                    // stores the same value)
                    if (d_valid) {
                        const uint32_t d_px = d_xy & 0xffffu, d_py = d_xy >> 16, d_pz = d_zk & 0xffffu;
                        uint32_t brick;
                        const uint64_t cell = cell_index_rel(d_px >> p.ss_shift, d_py >> p.ss_shift, d_pz >> p.ss_shift, p, brick);
                        p.occgrid[cell] = 1;
                        p.dirty_max[brick] = 1;
                    }
                    if (lane == 0) atomicAdd(&s_hits, (uint32_t) __popcll(all));
                    d_valid = false;
                    return;
                }
                // Direct MAX path: a hit of an unsplit triangle (order key 0: its only leaf) is the triangle's whole weight in
                // this (sub-)voxel, so it competes at once - one 64-bit atomic max on the cell, no hit record.  Hits of
                // subdivided triangles still go through the pool (their leaves' weights must be added up in order first).
                const bool direct = d_valid && use_direct && s_leaf[(d_zk >> 16) * kLeafStride + 19] == 0u;
                // The cell's counter hands out the hit's rank.  The first kInlineHits hits of a cell (by rank) go straight into
                // the slab of the cell's brick (Params::brick_slab: listed before this kernel by k_mark_bricks / k_scan_flags) -
                // they never see the pool, the scatter or a second copy: the random writes of the counting sort happen here,
                // under a kernel that is bound by instruction issue, not by memory; only the later hits of crowded cells are
                // pooled (k_scan_bricks turns the counts into offsets, k_scatter places them behind the slab's eight).
                // (handing the ranks out in k_scatter instead - no wait here - was measured: k_voxelize -2 %, k_scatter +13 % on
                // configs[3])
                uint32_t brick = 0, rank = 0, keyhi = 0, slab = 0;
                uint64_t cell = 0;
                const uint32_t *lf = &s_leaf[(d_zk >> 16) * kLeafStride];
                if (d_valid) {
                    const uint32_t d_px = d_xy & 0xffffu, d_py = d_xy >> 16, d_pz = d_zk & 0xffffu;
                    const uint32_t ox = d_px >> p.ss_shift, oy = d_py >> p.ss_shift, oz = d_pz >> p.ss_shift;
                    cell = cell_index_rel(ox, oy, oz, p, brick);
                    // (the origin is a whole number of output voxels: the relative coordinates have the absolute ones' parity)
                    const uint32_t sub = p.ss_shift ? ((d_px & 1u) | ((d_py & 1u) << 1) | ((d_pz & 1u) << 2)) : 0u;
                    keyhi = (sub << 29) | lf[18];
                    if (!direct) {
                        slab = p.brick_slab[brick];
                        rank = atomicAdd(&grid[cell], 1u);
                        if (rank >= kMaxRank) atomicOr(&c->err_flags, kErrRank);
                    }
                }
                const bool inlined = d_valid && !direct && rank < kInlineHits && slab < p.cap_slabs;
                // with textures a direct hit still leaves a record behind: {cell, key, colour} for k_pick
                const bool pooled = d_valid && !inlined && (UV || !direct);
                const unsigned long long mask = __ballot(pooled);
                const uint32_t cnt = (uint32_t) __popcll(mask);
                const uint32_t leader = mask ? (uint32_t) __ffsll((long long) mask) - 1u : 0u;
                if (cnt && chunk_used + cnt > kHitChunk) {
                    // abandon the rest of the chunk (marked as holes for the scatter pass) and reserve a new one
                    const uint32_t hole = chunk_base + chunk_used + lane;
                    if (chunk_used + lane < kHitChunk && hole < p.cap_hits) pool[hole].brick = kHoleBrick;
                    if (chunk_used + 64u + lane < kHitChunk && hole + 64u < p.cap_hits) pool[hole + 64u].brick = kHoleBrick;
                    if (chunk_used + 128u + lane < kHitChunk && hole + 128u < p.cap_hits) pool[hole + 128u].brick = kHoleBrick;
                    if (chunk_used + 192u + lane < kHitChunk && hole + 192u < p.cap_hits) pool[hole + 192u].brick = kHoleBrick;
                    uint32_t base = 0;
                    if (lane == leader) base = atomicAdd(&c->n_hits_reserved, kHitChunk);
                    chunk_base = __shfl(base, (int) leader, 64);
                    chunk_used = 0;
                }
                const uint32_t mine = chunk_base + chunk_used + __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
                chunk_used += cnt;
                if (d_valid) {
                    if (direct) {
                        atomicMax(&p.maxgrid[cell], ((unsigned long long) __float_as_uint(d_w) << 32) | (0xffffffffu - keyhi));
                        p.dirty_max[brick] = 1;  // benign race: every writer stores the same value
                        if (UV && mine < p.cap_hits) {
                            // the triangle is unsplit, so (d_u, d_v) already is its whole uv mean in this voxel; k_pick looks the
                            // colour up (moveUvBufferIntoVoxels, voxelization.cpp:513-526) for the cell's winner only
                            pool[mine] = HitRec{brick, ((uint32_t) cell & (kBrickCells - 1u)) << 24, keyhi, 0u, d_w, d_u, d_v, kPickRecord};
                        }
                    }
                    else {
                        if (use_direct) p.dirty_max[brick] = 1;  // the resolve kernels will add this cell's result
                        if (inlined) {
                            const size_t at = ((size_t) slab * kBrickCells + ((uint32_t) cell & (kBrickCells - 1u))) * kInlineHits + rank;
                            if (UV) reinterpret_cast<SortedRec *>(p.slabs)[at] = SortedRec{keyhi, lf[19], d_w, d_u, d_v, 0u};
                            else reinterpret_cast<uint4 *>(p.slabs)[at] = make_uint4(keyhi, lf[19], __float_as_uint(d_w), 0u);
                        }
                        else if (mine < p.cap_hits) {
                            pool[mine] = HitRec{brick, (((uint32_t) cell & (kBrickCells - 1u)) << 24) | (rank & (kMaxRank - 1u)), keyhi, lf[19], d_w,
                                                d_u, d_v, 0u};
                        }
                    }
                }
                if (lane == 0) atomicAdd(&s_hits, (uint32_t) __popcll(all));
                if (use_direct) {
                    const unsigned long long dmask = __ballot(direct);
                    if (lane == 0 && dmask) atomicAdd(&s_direct, (uint32_t) __popcll(dmask));
                }
                d_valid = false;
            };
#ifdef O2V_INSTRUMENT
#define O2V_EV(i, cond) do { if (cond) dbgc[i] += 1u; } while (0)
#else
#define O2V_EV(i, cond) do { } while (0)
#endif
            bool leaving = false;
            for (;;) {
                O2V_EV(0, lane == 0);
                {
                    // lanes whose job ended in a hit park it (w, u, v, position) in their result registers; if a lane's
                    // registers are still taken, everything parked is appended first
                    const bool fin_hit = has_job && !active && sp == 0;
                    const unsigned long long dv = __ballot(d_valid);
                    if (__ballot(fin_hit && d_valid) || (uint32_t) __popcll(dv) >= kFlushAt || (leaving && dv)) flush_results();
                    if (leaving) break;
                    if (fin_hit) {
                        d_w = w; d_u = u; d_v = v;
                        d_xy = pos_xy;
                        d_zk = pos_zk;
                        d_valid = true;
                        has_job = false;
                    }
                }
                // pop a pending sibling (with the plane mask it was pushed with), or fetch the next survivor
                if (!active) {
                    if (sp) {
                        sp -= 1u;
                        cur = stack_load<UV>(stack, sp);
                        if (sp >= stack_regs<UV>()) cur = overflow[sp - stack_regs<UV>()];
                        const uint32_t sh = __umul24(sp, 6u);  // 6 bits per entry
                        cf = (pmask >> sh) & 63u;
                        pmask &= ~(63u << sh);
                        active = true;
                    }
                    else if (next_valid) {
                        // the next job was fetched while this lane worked on the last one (take_job below)
                        const uint2 rec = job_record();
                        my_k = (rec.y >> 16) & 255u;
                        const uint32_t *lf = &s_leaf[__umul24(my_k, kLeafStride)];
                        pos_xy = rec.x;
                        pos_zk = rec.y & 0x00ffffffu;  // z | tile slot << 16
                        fx = (float) ((rec.x & 0xffffu) + p.so[0]);
                        fy = (float) ((rec.x >> 16) + p.so[1]);
                        fz = (float) ((rec.y & 0xffffu) + p.so[2]);
                        cf = (rec.y >> 24) & 63u;
                        small = (rec.y >> 30) & 1u;
                        lean = (rec.y >> 31) != 0u;
                        cur.a = {__uint_as_float(lf[0]), __uint_as_float(lf[1]), __uint_as_float(lf[2])};
                        cur.b = {__uint_as_float(lf[3]), __uint_as_float(lf[4]), __uint_as_float(lf[5])};
                        cur.c = {__uint_as_float(lf[6]), __uint_as_float(lf[7]), __uint_as_float(lf[8])};
                        if (UV) {
                            cur.ta = {__uint_as_float(lf[12]), __uint_as_float(lf[13])};
                            cur.tb = {__uint_as_float(lf[14]), __uint_as_float(lf[15])};
                            cur.tc = {__uint_as_float(lf[16]), __uint_as_float(lf[17])};
                        }
                        area = __uint_as_float(lf[23]);
                        margin = out_margin(s_mcoord[my_k]);
                        w = 0.f;
                        u = 0.f;
                        v = 0.f;
                        active = true;
                        has_job = true;
                        take_job();
                    }
                }
                // (wave-uniform: one job that needs the compiler's division sends the whole wavefront that way for the iteration)
                const bool lean_all = __ballot(active && !lean) == 0ull;
                O2V_EV(1, active);
#ifdef O2V_INSTRUMENT
                {
                    const uint32_t na = (uint32_t) __popcll(__ballot(active));
                    O2V_EV(4, lane == 0 && na <= 16u);
                    O2V_EV(6, lane == 0 && na <= 32u);
                }
#endif
                if (active) {
                    // `cf` names the planes (bit = level: lo x, y, z, hi x, y, z) this piece does not pass whole; all others
                    // are the loSum == 0 / loSum == 3 case of splitTriangle (voxelization.cpp:194-205), which hands the
                    // triangle on unchanged, so they are skipped.
                    if (cf == 0u) {
                        O2V_EV(2, true);
                        accumulate_piece<UV>(cur, area, w, u, v, lean_all);  // inside all remaining planes
                        active = false;
                        if (occ_only) {
                            sp = 0;
                            pmask = 0;
                        }
                    }
                    else {
                        const uint32_t level = (uint32_t) __ffs((int) cf) - 1u;
                        const bool keep_lo = level >= 3u;
                        const uint32_t axis = keep_lo ? level - 3u : level;
                        const float plane = (axis == 0 ? fx : (axis == 1 ? fy : fz)) + (keep_lo ? 1.0f : 0.0f);
                        const uint32_t cls = s_cls[classify_index(comp(cur.a, axis), comp(cur.b, axis), comp(cur.c, axis), plane)];
                        if ((cls & kClsModeMask) == 0u) {
                            // whole triangle to one side (one of the planar special cases, or a job whose masks are not
                            // computed: see piece_masks)
                            if (((cls & kClsSideLo) != 0) == keep_lo) {
                                O2V_EV(3, true);
                                cf &= cf - 1u;  // passed this plane; the next iteration goes on (or accumulates if none is left)
                            }
                            else {
                                O2V_EV(5, true);
                                active = false;  // discarded
                            }
                        }
                        else {
                            O2V_EV(7, true);
                            const uint32_t n = split_cut<UV>(cur, sec, cls, axis, plane, keep_lo, lean_all);
                            // What becomes of the kept pieces is decided at once, from their bounding boxes against the
                            // planes still ahead: a piece that passes them all is a final piece (accumulated now), one that
                            // lies beyond one of them (by a margin, see piece_masks) can only be discarded there, taking all
                            // its sub-pieces with it (dropped now), anything else goes on.  So a lane spends its iterations
                            // on cuts only.
                            const uint32_t later = 63u & ~((2u << level) - 1u);
                            uint32_t c_fail, c_out, c_near, s_fail, s_out, s_near;
                            piece_masks<UV>(cur, fx, fy, fz, small, margin, later, c_fail, c_out, c_near);
                            piece_masks<UV>(sec, fx, fy, fz, small, margin, later, s_fail, s_out, s_near);
                            const bool has_sec = n == 2u;
                            bool c_done = c_fail == 0u, s_done = has_sec && s_fail == 0u;
                            const bool c_drop = c_out != 0u, s_drop = has_sec && s_out != 0u;
                            uint32_t c_kept = c_done ? 1u : 0u, s_kept_n = s_done ? 1u : 0u;  // (without uv) final pieces they stand for
                            if (!UV) {
                                // a kept piece with one plane left is settled here and now: see single_plane
                                const bool c_single = !c_drop && single_plane(c_fail, c_near);
                                const bool s_single = has_sec && !s_drop && single_plane(s_fail, s_near);
                                O2V_EV(9, c_single);
                                O2V_EV(11, s_single);
                                if (c_single) {
                                    c_kept = single_plane_kept<UV>(cur, c_fail, fx, fy, fz, s_kept);
                                    c_done = true;
                                }
                                if (s_single) {
                                    s_kept_n = single_plane_kept<UV>(sec, s_fail, fx, fy, fz, s_kept);
                                    s_done = true;
                                }
                            }
                            O2V_EV(8, c_done);
                            O2V_EV(10, s_done);
                            // (value selects, not control flow: see sel_piece)
                            const bool c_over = c_done || c_drop;            // the first piece's subtree is finished
                            const bool s_live = has_sec && !s_done && !s_drop;  // the second piece needs more cuts
                            // without uv only the number of pieces matters, so a final second piece is counted at once; with
                            // uv it must wait for its turn if the first piece goes on (the running mean of
                            // voxelization.cpp:414-420 depends on the order): it is pushed with an empty mask
                            const bool s_acc_now = s_done && (!UV || c_over);
                            const bool s_push = has_sec && !s_drop && !c_over && !s_acc_now;
                            const bool s_takes_over = c_over && s_live;  // next in depth-first order
                            if (UV) {
                                if (c_done) accumulate_piece<UV>(cur, area, w, u, v, lean_all);
                                if (s_acc_now) accumulate_piece<UV>(sec, area, w, u, v, lean_all);
                            }
                            else {
                                // w += area once per final piece (util.hpp:160-165 with equal addends: the order is immaterial)
                                const uint32_t kept = c_kept + s_kept_n;
                                w = kept >= 1u ? w + area : w;
                                w = kept >= 2u ? w + area : w;
                                w = kept >= 3u ? w + area : w;
                                w = kept >= 4u ? w + area : w;
                            }
                            O2V_EV(12, s_push);
                            stack_store<UV>(stack, s_push ? sp : 7u, sec);  // 7: no slot, nothing stored
                            if (s_push && sp >= stack_regs<UV>()) overflow[sp - stack_regs<UV>()] = sec;
                            pmask |= s_push ? s_fail << __umul24(sp, 6u) : 0u;
                            sp += s_push ? 1u : 0u;
                            cur = sel_piece<UV>(s_takes_over, sec, cur);
                            cf = s_takes_over ? s_fail : c_fail;
                            active = s_takes_over || !c_over;
                            if (OCC && (c_kept | s_kept_n) != 0u) {
                                // a piece survived: the voxel is hit, the rest of the job cannot change that
                                active = false;
                                sp = 0;
                                pmask = 0;
                            }
                        }
                    }
                }
                // A job is finished when nothing of it is in flight.  `not eqExactly(uv.weight, 0.f)` -> insertWeighted
                // (voxelization.cpp:466-468): the hit is appended to the pool and counted in its cell; the ordered
                // combine happens in the resolve kernels.  A finished hit is parked in the lane's result registers; the
                // append section (flush_results, at the top of the loop) runs when kFlushAt lanes hold one, when a lane
                // needs its slot again or when the wavefront leaves - not in every iteration.
                const bool finished = has_job && !active && sp == 0;
                if (finished && w == 0.f) has_job = false;  // the voxel was not hit
                leaving = !__ballot(active || sp != 0 || next_valid || has_job);
            }
            O2V_LAP(1);
            t_begin = t_end;
        }
    }
    // the unused tail of this wavefront's last chunk holds no hits
    for (uint32_t k = chunk_used + lane; k < kHitChunk; k += 64)
        if (chunk_used != kHitChunk && chunk_base + k < p.cap_hits) pool[chunk_base + k].brick = kHoleBrick;
#ifdef O2V_INSTRUMENT
    tmr[3] = __builtin_readcyclecounter() - t_kernel0;
    for (uint32_t k = 0; k < 9; ++k) {
        uint32_t t = dbgc[k];
        for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
        if (lane == 0 && t) atomicAdd(&c->dbg[k], (unsigned long long) t);
    }
    if (lane == 0)
        for (uint32_t k = 0; k < 4; ++k) atomicAdd(&c->dbg[12 + k], tmr[k]);  // summed over the wavefronts
    if (lane == 0) {  // (phase 1's inside: reported in the slots of the last three cut events)
        atomicAdd(&c->dbg[9], t_drain);
        atomicAdd(&c->dbg[10], (unsigned long long) n_drain);
        atomicAdd(&c->dbg[11], (unsigned long long) n_drain_lanes);
    }
#endif
    __syncthreads();
    // (a certain hit is a hit, a direct one, and a voxel job that did not have to run)
    if (threadIdx.x == 0 && (s_hits | s_certain)) atomicAdd(&c->n_hits, (unsigned long long) s_hits + s_certain);
    // (occupancy only: every hit is a direct one)
    if (threadIdx.x == 0 && OCC && (s_hits | s_certain)) atomicAdd(&c->n_direct, (unsigned long long) s_hits + s_certain);
    if (threadIdx.x == 0 && !OCC && s_direct) atomicAdd(&c->n_direct, (unsigned long long) s_direct);
    if (threadIdx.x == 0 && s_certain) atomicAdd(&c->n_certain, (unsigned long long) s_certain);
    if (threadIdx.x == 0 && s_skipped) atomicAdd(&c->n_jobs_skipped, (unsigned long long) s_skipped);
    if (OCC && threadIdx.x == 0 && p.solo_roots && s_solo[0]) {
        atomicAdd(&c->n_bypass, s_solo[0]);
        atomicAdd(&c->n_candidates, s_solo[1]);
        atomicAdd(&c->n_candidates_sq, s_solo[2]);
    }
}

// The clip kernel of the weighted routes (UV: the mesh has textured triangles) ...
template <bool UV>
__global__ __launch_bounds__(VoxShape<UV>::block, (UV ? O2V_K2_WAVES_UV : O2V_K2_WAVES)) void k_voxelize(const Leaf *__restrict__ leaves, const Tile *__restrict__ tiles,
                                                     Counters *c, uint32_t *grid, uint8_t *brick_dirty, HitRec *pool,
                                                     uint2 *jobq_all, Params p)
{
    voxelize_body<UV, false>(leaves, tiles, c, grid, brick_dirty, pool, jobq_all, p);
}
// ... and of occupancy-only mode (Params::occupancy_only: no triangle of the mesh has a material).  Its job queues are twice
// as long: the second half holds the jobs that are left when the voxels marked already have been taken out.
__global__ __launch_bounds__(VoxShape<false>::block, O2V_K2_WAVES) void k_voxelize_occ(const Leaf *__restrict__ leaves, const Tile *__restrict__ tiles,
                                                     Counters *c, uint32_t *grid, uint8_t *brick_dirty, HitRec *pool,
                                                     uint2 *jobq_all, const float *__restrict__ verts, const uint32_t *__restrict__ block_list,
                                                     const uint32_t *block_count, Params p)
{
    voxelize_body<false, true>(leaves, tiles, c, grid, brick_dirty, pool, jobq_all, p, verts, block_list, block_count);
}
