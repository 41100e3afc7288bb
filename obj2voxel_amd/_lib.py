"""Loader for libobj2voxel_amd.so (HIP kernels + C++ host layer). No CPU fallback exists: if the library is
missing, importing the bindings fails loudly."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libobj2voxel_amd.so")
CSRC = os.path.join(_HERE, "csrc")


def build(force=False):
    """Compile the HIP kernels for gfx950 and the host layer in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-s"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("build did not produce " + LIB_PATH)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("O2V_LIB") or LIB_PATH  # O2V_LIB: a developer build (e.g. the instrumented library)
        if not os.path.exists(path):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(obj2voxel_amd has no CPU fallback)")
        _lib = C.CDLL(path)
    return _lib
