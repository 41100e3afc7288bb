"""Developer tool: compare device vs oracle on a case and dump the hit lists of the first mismatching voxel."""
import os
os.environ.setdefault("O2V_NO_DIRECT_MAX", "1")  # the hit lists are only kept on the sort-and-replay route
import ctypes as C
import struct
import sys

sys.path.insert(0, '.')
import numpy as np
from obj2voxel_amd import hip, meshes
from oracle import oracle

dv = hip.DeviceVoxelizer(0)
L = dv._L
L.o2v_hip_debug_cell_hits.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
oracle.lib().o2v_oracle_trace_voxel.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32]


def cmp(name, verts, res, **kw):
    dv.set_triangles(verts, uvs=kw.get('uvs'), types=kw.get('types'), colors=kw.get('colors'), texids=kw.get('texids'))
    run = {k: kw[k] for k in ('strategy', 'supersampling') if k in kw}
    g = meshes.sorted_voxels(dv.voxelize(res, **run))
    w = meshes.sorted_voxels(oracle.voxelize(verts, res, **kw))
    print(name, 'counts', len(g), len(w), dv.stats(), oracle.stats())
    if len(g) != len(w):
        return
    bad = np.flatnonzero(g[:, 3] != w[:, 3])
    print('  pos equal', np.array_equal(g[:, :3], w[:, :3]), 'color mismatches', len(bad))
    for b in bad[:2]:
        x, y, z = (int(t) for t in g[b, :3])
        print('   voxel', x, y, z, hex(g[b, 3]), hex(w[b, 3]))
        buf = np.zeros(64 * 6, np.uint32)
        n = C.c_uint32()
        L.o2v_hip_debug_cell_hits(dv._ctx, x, y, z, buf.ctypes.data, 64, C.byref(n))
        recs = buf[:n.value * 6].reshape(-1, 6)
        for r in sorted(recs.tolist()):
            wf = struct.unpack('f', struct.pack('I', r[2]))[0]
            print('   dev hit tri=%d key=%08x w=%s (%.9g)' % (r[0] & 0x1fffffff, r[1], float(wf).hex(), wf))
        oracle.lib().o2v_oracle_trace_voxel(1, x, y, z)
        oracle.voxelize(verts, res, **kw)
        oracle.lib().o2v_oracle_trace_voxel(0, 0, 0, 0)


v = meshes.uv_sphere(10)
T = len(v)
col = meshes.triangle_colors(T)
ty = np.full(T, 2, np.uint32)
cmp('col blend', v, 96, types=ty, colors=col, strategy=1)
