"""obj2voxel_amd -- MI355X-native voxelizer behind obj2voxel's C API.

The product is the shared library libobj2voxel_amd.so (hand-written HIP kernels for gfx950 + a C++ host layer),
built in-tree from obj2voxel_amd/csrc.  This package only binds it:

  obj2voxel_amd.hip    DeviceVoxelizer over the device C-ABI (include/o2v_hip.h)
  obj2voxel_amd.capi   the drop-in public C API (include/obj2voxel.h) through ctypes
  obj2voxel_amd.meshes deterministic synthetic meshes for tests and bench.py

There is no CPU implementation in this package; everything fails loudly if the library or the GPU is missing.
"""
from ._lib import LIB_PATH, build  # noqa: F401
