"""Deterministic synthetic triangle meshes used by tests and bench.py.

The reference's own fixtures (unit cube, three planes; test/main.cpp:14-61 with the quad->triangle rule of
test/testutil.hpp:84-106) are regenerated here as data; the sphere family stands in for the assets
BASELINE.json names (Spot / Dragon / Sponza are not in the reference tree and there is no network).
All generators return float32 arrays shaped [T, 9] (three xyz vertices per triangle).
"""
import numpy as np

# test/main.cpp:20-38 (vertex and quad-element tables, data only)
_UNIT_CUBE_VERTS = np.array([[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1],
                             [1, 0, 0], [1, 0, 1], [1, 1, 0], [1, 1, 1]], dtype=np.float32)
_UNIT_CUBE_QUADS = np.array([[0, 1, 3, 2], [4, 6, 7, 5], [0, 4, 5, 1],
                             [2, 3, 7, 6], [0, 2, 6, 4], [1, 5, 7, 3]])
# test/main.cpp:40-61
_THREE_PLANES_VERTS = np.array([[x, 0, 0] for x in (0.0, 0.5, 1.0)], dtype=np.float32)


def _quads_to_tris(verts, quads):
    """Quad (0,1,2,3) -> triangles (0,1,2), (2,3,0): test/testutil.hpp:84-106."""
    tris = []
    for q in quads:
        tris.append([verts[q[0]], verts[q[1]], verts[q[2]]])
        tris.append([verts[q[2]], verts[q[3]], verts[q[0]]])
    return np.asarray(tris, dtype=np.float32).reshape(-1, 9)


def unit_cube():
    return _quads_to_tris(_UNIT_CUBE_VERTS, _UNIT_CUBE_QUADS)


def three_planes():
    verts = []
    for x in (0.0, 0.5, 1.0):
        verts += [[x, 0, 0], [x, 0, 1], [x, 1, 1], [x, 1, 0]]
    verts = np.asarray(verts, dtype=np.float32)
    quads = np.arange(12).reshape(3, 4)
    return _quads_to_tris(verts, quads)


def single_triangle():
    """test/main.cpp:14-18."""
    return np.array([[0, 0, 0, 0, 0, 1, 1, 0, 0]], dtype=np.float32)


def uv_sphere(nv, nu=None, radius=1.0, center=(0.0, 0.0, 0.0), with_uv=False):
    """UV sphere with nv latitude bands and nu (=2*nv) longitude segments, poles on the y axis.

    T = 2*nu*(nv-1) triangles (pole bands contribute one triangle per segment).
    Vertices are computed in float64 and rounded once to float32, so the mesh is reproducible bit-for-bit.
    """
    nu = 2 * nv if nu is None else nu
    theta = np.pi * np.arange(nv + 1, dtype=np.float64) / nv          # 0..pi
    phi = 2.0 * np.pi * np.arange(nu + 1, dtype=np.float64) / nu      # 0..2pi
    st, ct = np.sin(theta), np.cos(theta)
    sp, cp = np.sin(phi), np.cos(phi)
    sp[-1], cp[-1] = sp[0], cp[0]  # close the seam exactly
    st[0] = st[-1] = 0.0
    P = np.empty((nv + 1, nu + 1, 3), dtype=np.float64)
    P[..., 0] = radius * st[:, None] * cp[None, :] + center[0]
    P[..., 1] = radius * ct[:, None] * np.ones_like(cp)[None, :] + center[1]
    P[..., 2] = radius * st[:, None] * sp[None, :] + center[2]
    UV = np.empty((nv + 1, nu + 1, 2), dtype=np.float64)
    UV[..., 0] = (np.arange(nu + 1) / nu)[None, :]
    UV[..., 1] = (np.arange(nv + 1) / nv)[:, None]
    i, j = np.meshgrid(np.arange(nv), np.arange(nu), indexing="ij")
    i, j = i.ravel(), j.ravel()
    a, b, c, d = (i, j), (i + 1, j), (i + 1, j + 1), (i, j + 1)
    tri1 = (a, b, c)   # valid unless i == nv-1 (b == c at the south pole)
    tri2 = (a, c, d)   # valid unless i == 0 (a == d at the north pole)
    m1 = i != nv - 1
    m2 = i != 0
    tris, uvs = [], []
    for (p, q, r), m in ((tri1, m1), (tri2, m2)):
        tris.append(np.stack([P[p[0][m], p[1][m]], P[q[0][m], q[1][m]], P[r[0][m], r[1][m]]], axis=1))
        uvs.append(np.stack([UV[p[0][m], p[1][m]], UV[q[0][m], q[1][m]], UV[r[0][m], r[1][m]]], axis=1))
    # interleave so that triangle order follows the (i, j) sweep rather than all-tri1-then-all-tri2
    order = np.argsort(np.concatenate([2 * np.flatnonzero(m1), 2 * np.flatnonzero(m2) + 1]), kind="stable")
    verts = np.concatenate(tris, axis=0)[order].astype(np.float32).reshape(-1, 9)
    if with_uv:
        return verts, np.concatenate(uvs, axis=0)[order].astype(np.float32).reshape(-1, 6)
    return verts


def readme_blade(with_uv=True):
    """Stand-in for the only workload the reference publishes a number for (README.adoc:177-178, img/terminal_screenshot.png:
    a 19 392-triangle textured model - a sword - at resolution 8192: 20.3 M voxels): a prolate ellipsoid, 19 320 triangles
    (uv_sphere(70) with two axes scaled by 0.0806), whose surface at 8192^3 is ~13 M voxel faces wide - some 20 M voxels in a
    long thin box of the grid, which is what makes that workload the worst case of a dense grid and the best of a sparse map."""
    v, uv = uv_sphere(70, with_uv=True)
    v = v.reshape(-1, 3, 3) * np.array([0.0806, 1.0, 0.0806], dtype=np.float32)
    v = np.ascontiguousarray(v.reshape(-1, 9), dtype=np.float32)
    return (v, uv) if with_uv else v


def triangle_colors(T):
    """Per-triangle colours ((37i)%256, (91i)%256, (13i)%256)/255."""
    i = np.arange(T, dtype=np.int64)
    return (np.stack([(37 * i) % 256, (91 * i) % 256, (13 * i) % 256], axis=1) / 255.0).astype(np.float32)


def checker_texture(size=256, tiles=16):
    """Procedural RGB texture: checkerboard modulated by a gradient; uint8 [size, size, 3]."""
    y, x = np.mgrid[0:size, 0:size]
    chk = (((x * tiles) // size + (y * tiles) // size) & 1).astype(np.uint8)
    img = np.empty((size, size, 3), dtype=np.uint8)
    img[..., 0] = (x * 255) // (size - 1)
    img[..., 1] = (y * 255) // (size - 1)
    img[..., 2] = 40 + 200 * chk
    return img


def box_room(n=8):
    """Axis-aligned box interior made of n x n quads per wall (large aligned triangles: exercises the
    aligned fast path of voxelization.cpp:335-347,503 and large leaves)."""
    tris = []
    g = np.linspace(0.0, 1.0, n + 1)
    for axis in range(3):
        for side in (0.0, 1.0):
            for a in range(n):
                for b in range(n):
                    def pt(u, v):
                        p = [0.0, 0.0, 0.0]
                        p[axis] = side
                        p[(axis + 1) % 3] = u
                        p[(axis + 2) % 3] = v
                        return p
                    q = [pt(g[a], g[b]), pt(g[a + 1], g[b]), pt(g[a + 1], g[b + 1]), pt(g[a], g[b + 1])]
                    tris.append([q[0], q[1], q[2]])
                    tris.append([q[2], q[3], q[0]])
    return np.asarray(tris, dtype=np.float32).reshape(-1, 9)


def diagonal_strip(n=400, width=6e-5, thickness=4e-5):
    """A thin ribbon along the diagonal of the unit square in the z ~ 0 plane, slightly wavy in z (so that its triangles are not
    axis-aligned): 2 n triangles whose bounds are (0, 0, 0) .. (1, 1, thickness).  At a resolution of 100 000 it crosses every x
    and y of the grid while the mesh's box stays a few layers thick - the shape the x / y tile tests use."""
    t = np.linspace(0.0, 1.0, n + 1, dtype=np.float64)
    # centre line x = y = t (inside the unit square by the ribbon's half width), z waving between 0 and thickness
    half = 0.5 * width
    cx = half + t * (1.0 - width)
    z = 0.5 * thickness * (1.0 + np.sin(np.arange(n + 1) * 1.7))
    a = np.stack([cx - half, cx + half, z], axis=1)          # one edge of the ribbon
    b = np.stack([cx + half, cx - half, thickness - z], axis=1)   # the other
    tris = []
    for i in range(n):
        tris.append(np.concatenate([a[i], b[i], a[i + 1]]))
        tris.append(np.concatenate([b[i], b[i + 1], a[i + 1]]))
    v = np.asarray(tris, dtype=np.float32)
    # (the bounds are exactly the unit square: the first / last vertices touch 0 and 1)
    v[0, 0] = 0.0; v[0, 4] = 0.0
    return np.ascontiguousarray(v)


def random_soup(T, seed=0, scale=0.2):
    """Random small triangles in the unit cube plus a few large diagonal ones (subdivision-heavy)."""
    rng = np.random.default_rng(seed)
    c = rng.random((T, 1, 3), dtype=np.float64)
    v = c + scale * (rng.random((T, 3, 3), dtype=np.float64) - 0.5)
    return np.clip(v, 0.0, 1.0).astype(np.float32).reshape(-1, 9)


def sorted_voxels(vox):
    """Canonical order for comparing unordered (x,y,z,argb) outputs: sort by z, y, x."""
    vox = np.asarray(vox, dtype=np.uint32).reshape(-1, 4)
    if vox.shape[0] == 0:
        return vox
    key = (vox[:, 2].astype(np.uint64) << np.uint64(42)) | (vox[:, 1].astype(np.uint64) << np.uint64(21)) \
        | vox[:, 0].astype(np.uint64)
    return vox[np.argsort(key, kind="stable")]


def stress_soup(kind, T, S, seed=0, z_range=None):
    """Large random triangle soups in VOXEL coordinates of an S^3 sample grid, for the fast-vs-exact comparison of the clip
    kernel (tests/test_gpu_exact_ab.py) and bench routes.  Use together with `stress_bounds(S)`, under which the mesh
    transform (reference obj2voxel.cpp:370-402) is x -> x + 0.5 up to rounding.  Vectorised; float64 -> float32 once.

    kind: "small"   triangles of 0.3 .. 8 voxels, any orientation
          "sliver"  5 .. 120 voxels long, 0.001 .. 0.5 voxels wide (subdivision, near-degenerate normals)
          "huge"    like "small" plus 300 (at 1024^3; fewer above) polygons of up to S/2 voxels, a third of them axis-aligned
          "planar"  vertices on integer voxel planes or a few 2^-16 beside them (the splitter's planar cases and epsilon)
          "mixed"   a quarter of each
    z_range: (z0, z1) in voxels to confine the triangle centres to (for z-slab runs), default the whole grid.
    """
    rng = np.random.default_rng(seed)
    if kind == "mixed":
        parts = [stress_soup(k, T // 4, S, seed * 4 + i + 1, z_range) for i, k in enumerate(("small", "sliver", "huge", "planar"))]
        v = np.concatenate(parts)
        return v[np.random.default_rng(seed).permutation(len(v))]
    lo = np.array([2.0, 2.0, 2.0 if z_range is None else z_range[0]])
    hi = np.array([S - 2.0, S - 2.0, S - 2.0 if z_range is None else z_range[1]])
    c = lo + (hi - lo) * rng.random((T, 1, 3))
    if kind in ("small", "huge"):
        size = np.exp(rng.uniform(np.log(0.3), np.log(8.0), size=(T, 1, 1)))
        v = c + size * (rng.random((T, 3, 3)) - 0.5)
        if kind == "huge":
            n = max(20, int(300 * (1024.0 / S) ** 2))  # about the same number of voxels at every resolution
            big = c[:n] + (S / 2.0) * (rng.random((n, 3, 3)) - 0.5)
            for i in range(0, n, 3):
                big[i][:, rng.integers(0, 3)] = np.round(c[i, 0, 0])  # lies in a voxel plane
            v[:n] = big
    elif kind == "sliver":
        d = rng.normal(size=(T, 1, 3))
        d /= np.linalg.norm(d, axis=2, keepdims=True)
        e = rng.normal(size=(T, 1, 3))
        e -= d * np.sum(d * e, axis=2, keepdims=True)
        e /= np.linalg.norm(e, axis=2, keepdims=True)
        length = np.exp(rng.uniform(np.log(5.0), np.log(120.0), size=(T, 1, 1)))
        width = np.exp(rng.uniform(np.log(1e-3), np.log(0.5), size=(T, 1, 1)))
        t = np.stack([np.full(T, -0.5), np.full(T, 0.5), rng.uniform(-0.5, 0.5, size=T)], axis=1)[:, :, None]
        s = np.stack([np.zeros(T), np.zeros(T), np.ones(T)], axis=1)[:, :, None]
        v = c + d * length * t + e * width * s
    elif kind == "planar":
        k = np.round(c) + rng.integers(-3, 4, size=(T, 3, 3))
        noise = rng.choice([0.0, 0.0, 1e-6, -1e-6, 1.4e-5, -1.4e-5, 1.7e-5, -1.7e-5, 3e-4, 0.25, 0.5], size=(T, 3, 3))
        v = k - 0.5 + noise
        flat = np.arange(0, T, 5)
        ax = rng.integers(0, 3, size=len(flat))
        v[flat, :, ax] = v[flat, 0, ax][:, None]  # triangles lying in one voxel plane
    else:
        raise ValueError(kind)
    return np.clip(v, 0.0, S - 1.0).astype(np.float32).reshape(-1, 9)


def stress_bounds(S):
    """User bounds that make the mesh transform x -> x + 0.5 (up to rounding) on an S^3 sample grid."""
    return [-0.25] * 3 + [S - 0.75] * 3


def scan_like(target=870_000):
    """A deterministic "scanned object": an icosphere refined adaptively (five refinement levels side by side: triangle areas
    spread over more than 100 : 1), its surface displaced radially by a few octaves of smooth pseudo-noise (no RNG: products
    of sines with incommensurate frequencies), every twentieth triangle collapsed into a sliver along one of its edges (5 %
    of the list, aspect ratios around 50 : 1; the holes they leave are what scans have too).  About `target` triangles in
    submission order of the refinement (spatially coherent, as a scanner's strips are).  float64 construction, rounded once
    to float32: reproducible bit for bit.  Returns verts [T, 9]."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    V = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    V /= np.linalg.norm(V, axis=1, keepdims=True)
    F = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                  [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7],
                  [9, 8, 1]])
    tris = V[F]  # [20, 3, 3]

    def split(tr):
        a, b, c = tr[:, 0], tr[:, 1], tr[:, 2]
        ab, bc, ca = a + b, b + c, c + a
        ab /= np.linalg.norm(ab, axis=1, keepdims=True)
        bc /= np.linalg.norm(bc, axis=1, keepdims=True)
        ca /= np.linalg.norm(ca, axis=1, keepdims=True)
        # children in a fixed order, kept next to each other (spatial coherence of the list)
        return np.stack([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1), np.stack([ca, bc, c], 1), np.stack([ab, bc, ca], 1)], 1).reshape(-1, 3, 3)

    base_levels = 5
    for _ in range(base_levels):
        tris = split(tris)  # 20 * 4^5 = 20 480
    # extra refinement 0..4 per base triangle from a smooth field of its centroid; the thresholds put ~ 15 / 25 / 30 / 20 / 10 %
    # of the base triangles on the levels 0..4, which gives about 870 k triangles
    cen = tris.mean(axis=1)
    field = (np.sin(3.1 * cen[:, 0] + 0.4) * np.cos(2.3 * cen[:, 1] - 1.1) + 0.6 * np.sin(4.7 * cen[:, 2] + 2.0 * cen[:, 0]) +
             0.35 * np.cos(7.9 * cen[:, 1] + 1.7 * cen[:, 2]))
    qs = np.quantile(field, [0.15, 0.40, 0.70, 0.90])
    level = np.searchsorted(qs, field)  # 0..4
    # scale the split so that the total is close to `target`
    total = int((4.0 ** level).sum())
    if total > target * 1.15 or total < target * 0.85:
        # (other targets: shift every level by the same amount where possible)
        shift = int(np.round(np.log(target / total) / np.log(4.0)))
        level = np.clip(level + shift, 0, 6)
    # refine level by level, keeping each base triangle's descendants contiguous: process base triangles in order, in
    # groups of equal level (stable: concatenate per base index afterwards)
    order_keys, pieces = [], []
    for lv in range(int(level.max()) + 1):
        idx = np.flatnonzero(level == lv)
        if len(idx) == 0:
            continue
        tr = tris[idx]
        for _ in range(lv):
            tr = split(tr)
        n_per = 4 ** lv
        pieces.append(tr)
        order_keys.append(np.repeat(idx, n_per).astype(np.int64) * 4096 + np.tile(np.arange(n_per), len(idx)))
    tr = np.concatenate(pieces, axis=0)
    tr = tr[np.argsort(np.concatenate(order_keys), kind="stable")]
    # radial displacement: smooth pseudo-noise, four octaves
    def noise(p):
        x, y, z = p[..., 0], p[..., 1], p[..., 2]
        n = 0.060 * np.sin(2.1 * x + 1.3) * np.sin(1.7 * y - 0.6) * np.cos(2.9 * z + 0.2)
        n += 0.030 * np.sin(5.3 * x - 2.2 * z) * np.cos(4.1 * y + 0.9)
        n += 0.012 * np.sin(11.7 * y + 3.1 * x) * np.sin(9.3 * z - 1.4)
        n += 0.004 * np.cos(23.9 * x + 17.1 * y - 19.7 * z)
        return n
    tr = tr * (1.0 + noise(tr))[..., None]
    # slivers: every twentieth triangle is collapsed onto its first edge (c -> 2 % of the way from the edge's midpoint)
    sl = np.arange(7, len(tr), 20)
    mid = 0.5 * (tr[sl, 0] + tr[sl, 1])
    tr[sl, 2] = mid + 0.02 * (tr[sl, 2] - mid)
    return tr.astype(np.float32).reshape(-1, 9)
