import sys, json
sys.path.insert(0, '/root/repo')
from obj2voxel_amd import hip, workloads
for name in sys.argv[1:]:
    verts, mat, textures, res, kw, text = workloads.load(name)
    dv = hip.DeviceVoxelizer(0)
    dv.set_textures(textures or [])
    dv.set_triangles(verts, **mat)
    dv.voxelize(res, read=False, **kw)
    dv.voxelize(res, read=False, **kw)
    c = dv.debug_counters()
    print(name, [int(x) for x in c[:16]])
    dv.close()
