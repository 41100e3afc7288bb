"""ctypes binding of the device C-ABI (include/o2v_hip.h): one DeviceVoxelizer per GPU / z-slab."""
import ctypes as C

import numpy as np

from ._lib import lib

TRI_MATERIALLESS, TRI_UNTEXTURED, TRI_TEXTURED = 1, 2, 3
STRATEGY_MAX, STRATEGY_BLEND = 0, 1


class _Params(C.Structure):
    _fields_ = [("resolution", C.c_uint32), ("supersampling", C.c_uint32), ("strategy", C.c_uint32),
                ("unit_transform", C.c_int32 * 9), ("bounds_known", C.c_uint32), ("bounds", C.c_float * 6),
                ("z_begin", C.c_uint32), ("z_end", C.c_uint32), ("flags", C.c_uint32),
                ("x_begin", C.c_uint32), ("x_end", C.c_uint32), ("y_begin", C.c_uint32), ("y_end", C.c_uint32)]


FLAG_KERNEL_TIMES = 2  # ... every launch bracketed by events: DeviceVoxelizer.kernel_times()
FLAG_STAGE_TIMES = 4   # ... an event between the stages of a pass: the stage times and total_ms of DeviceVoxelizer.timings()
FLAG_EXACT_CLIP = 1  # o2v_hip_params::flags: the clip kernel without its work-removal shortcuts (include/o2v_hip.h)


class _Texture(C.Structure):
    _fields_ = [("pixels", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32),
                ("channels", C.c_uint32), ("wrap", C.c_uint32)]


class Timings(C.Structure):
    _fields_ = [("bounds_ms", C.c_float), ("expand_ms", C.c_float), ("voxelize_ms", C.c_float),
                ("scan_ms", C.c_float), ("resolve_ms", C.c_float), ("total_ms", C.c_float), ("passes", C.c_uint32),
                ("plan_ms", C.c_float), ("collective_ms", C.c_float), ("collective_parts_ms", C.c_float * 5)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_}
        d["collective_parts_ms"] = [float(x) for x in self.collective_parts_ms]   # status + bounds (one reduce), -, histograms + block extents (one gather), -, counts
        return d


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("triangles", "leaves", "tiles", "candidates", "hits", "voxels",
                                          "grid_cells", "grid_bytes", "bricks", "dirty_bricks", "pool_slots", "direct_hits", "jobs",
                                          "certain_hits", "skipped_jobs", "bypassed_leaves")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("ms", C.c_float), ("launches", C.c_uint32)]


class DeviceError(RuntimeError):
    pass


def _bind():
    L = lib()
    L.o2v_hip_device_count.restype = C.c_int
    L.o2v_hip_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.o2v_hip_destroy.argtypes = [C.c_void_p]
    L.o2v_hip_last_error.argtypes = [C.c_void_p]
    L.o2v_hip_last_error.restype = C.c_char_p
    L.o2v_hip_set_triangles.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_uint64]
    L.o2v_hip_set_textures.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.o2v_hip_voxelize.argtypes = [C.c_void_p, C.POINTER(_Params), C.POINTER(C.c_uint64)]
    L.o2v_hip_read_voxels.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
    L.o2v_hip_get_timings.argtypes = [C.c_void_p, C.POINTER(Timings)]
    L.o2v_hip_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.o2v_hip_get_transform.argtypes = [C.c_void_p, C.c_void_p]
    L.o2v_hip_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
    L.o2v_hip_debug_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.o2v_hip_debug_check_third.argtypes = [C.c_void_p, C.c_void_p]
    L.o2v_hip_debug_check_div.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]
    L.o2v_hip_comm_unique_id.argtypes = [C.c_void_p]
    L.o2v_hip_comm_create_rccl.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.o2v_hip_comm_create_callbacks.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.o2v_hip_comm_destroy.argtypes = [C.c_void_p]
    L.o2v_hip_comm_kind.argtypes = [C.c_void_p]
    L.o2v_hip_comm_kind.restype = C.c_char_p
    L.o2v_hip_comm_last_error.argtypes = [C.c_void_p]
    L.o2v_hip_comm_last_error.restype = C.c_char_p
    L.o2v_hip_voxelize_sharded.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_Params), C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p]
    L.o2v_hip_group_create.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.o2v_hip_group_destroy.argtypes = [C.c_void_p]
    L.o2v_hip_group_size.argtypes = [C.c_void_p]
    L.o2v_hip_group_size.restype = C.c_uint32
    L.o2v_hip_group_ctx.argtypes = [C.c_void_p, C.c_uint32]
    L.o2v_hip_group_ctx.restype = C.c_void_p
    L.o2v_hip_group_comm_kind.argtypes = [C.c_void_p]
    L.o2v_hip_group_comm_kind.restype = C.c_char_p
    L.o2v_hip_group_last_error.argtypes = [C.c_void_p]
    L.o2v_hip_group_last_error.restype = C.c_char_p
    L.o2v_hip_group_set_triangles.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_uint64, C.c_int]
    L.o2v_hip_group_set_textures.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.o2v_hip_group_voxelize.argtypes = [C.c_void_p, C.POINTER(_Params), C.c_void_p, C.c_void_p]
    L.o2v_hip_plan_slabs.argtypes = [C.c_void_p, C.POINTER(_Params), C.c_uint32, C.c_void_p, C.c_void_p]
    L.o2v_hip_get_kernel_times.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    L.o2v_hip_build_id.restype = C.c_char_p
    L.o2v_mesh_load_file.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
    L.o2v_mesh_arrays.argtypes = [C.c_void_p] + [C.POINTER(C.c_void_p)] * 5 + [C.POINTER(C.c_uint32)]
    L.o2v_mesh_arrays.restype = C.c_uint64
    L.o2v_mesh_texture.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(_Texture)]
    L.o2v_mesh_free.argtypes = [C.c_void_p]
    return L


def build_id():
    """Hash of the device sources the loaded library was built from (o2v_hip_build_id)."""
    return _bind().o2v_hip_build_id().decode()


def load_mesh_file(path):
    """o2v_mesh_load_file: (verts [T, 9], materials dict for set_triangles, textures list for set_textures) of an OBJ / STL
    file, read by the library's own readers."""
    L = _bind()
    h = C.c_void_p()
    if L.o2v_mesh_load_file(str(path).encode(), None, C.byref(h)) != 0:
        raise DeviceError(f"o2v_mesh_load_file({path}) failed: unknown type or unreadable file")
    try:
        ptrs = [C.c_void_p() for _ in range(5)]
        ntex = C.c_uint32(0)
        T = int(L.o2v_mesh_arrays(h, *[C.byref(q) for q in ptrs], C.byref(ntex)))

        def arr(q, ctype, width, dtype):
            if not q.value or not T:
                return None
            return np.ctypeslib.as_array(C.cast(q, C.POINTER(ctype)), shape=(T * width,)).astype(dtype).reshape(T, width).copy()
        verts = arr(ptrs[0], C.c_float, 9, np.float32)
        mat = {}
        for key, q, ctype, width, dtype in (("uvs", ptrs[1], C.c_float, 6, np.float32), ("types", ptrs[2], C.c_uint32, 1, np.uint32),
                                            ("colors", ptrs[3], C.c_float, 3, np.float32), ("texids", ptrs[4], C.c_int32, 1, np.int32)):
            a = arr(q, ctype, width, dtype)
            if a is not None:
                mat[key] = a.reshape(T) if width == 1 else a
        textures = []
        for i in range(ntex.value):
            t = _Texture()
            L.o2v_mesh_texture(h, i, C.byref(t))
            pix = np.ctypeslib.as_array(C.cast(t.pixels, C.POINTER(C.c_uint8)), shape=(t.height, t.width, t.channels)).copy()
            textures.append((pix, int(t.wrap)))
        return (np.zeros((0, 9), np.float32) if verts is None else verts), mat, textures
    finally:
        L.o2v_mesh_free(h)


def device_count():
    return _bind().o2v_hip_device_count()


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class DeviceVoxelizer:
    """Owns one GPU's dense grid slab and work buffers; reusable across voxelize() calls."""

    def __init__(self, device=0, _borrowed_ctx=None):
        self._L = _bind()
        self._owned = _borrowed_ctx is None
        self._n_out = C.c_uint64(0)
        if _borrowed_ctx is not None:       # a rank of a DeviceGroup: the group owns the context
            self._ctx = C.c_void_p(_borrowed_ctx)
            return
        self._ctx = C.c_void_p()
        rc = self._L.o2v_hip_create(device, C.byref(self._ctx))
        if rc != 0:
            raise DeviceError(f"o2v_hip_create(device={device}) failed with code {rc}: no usable MI355X / HIP runtime")
        self._keep = []

    def close(self):
        if self._ctx and self._owned:
            self._L.o2v_hip_destroy(self._ctx)
        self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise DeviceError(f"{what} failed with code {rc}: {self._L.o2v_hip_last_error(self._ctx).decode()}")

    def set_triangles(self, verts, uvs=None, types=None, colors=None, texids=None):
        verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 9)
        T = verts.shape[0]
        uvs = None if uvs is None else np.ascontiguousarray(uvs, dtype=np.float32).reshape(T, 6)
        types = None if types is None else np.ascontiguousarray(types, dtype=np.uint32).reshape(T)
        colors = None if colors is None else np.ascontiguousarray(colors, dtype=np.float32).reshape(T, 3)
        texids = None if texids is None else np.ascontiguousarray(texids, dtype=np.int32).reshape(T)
        self._check(self._L.o2v_hip_set_triangles(self._ctx, _ptr(verts), _ptr(uvs), _ptr(types), _ptr(colors),
                                                  _ptr(texids), T), "o2v_hip_set_triangles")
        self.n_tris = T

    def set_textures(self, textures):
        """textures: sequence of (uint8 [h, w, c] pixels, wrap) with c in (3, 4)."""
        arr = (_Texture * max(1, len(textures)))()
        keep = []
        for i, (pix, wrap) in enumerate(textures):
            pix = np.ascontiguousarray(pix, dtype=np.uint8)
            keep.append(pix)
            h, w, c = pix.shape
            arr[i] = _Texture(pix.ctypes.data, w, h, c, int(wrap))
        self._check(self._L.o2v_hip_set_textures(self._ctx, C.cast(arr, C.c_void_p), len(textures)),
                    "o2v_hip_set_textures")

    @staticmethod
    def _params(resolution, supersampling, strategy, unit_transform, bounds, zslab, flags=0, xtile=(0, 0), ytile=(0, 0)):
        p = _Params()
        p.x_begin, p.x_end = xtile
        p.y_begin, p.y_end = ytile
        p.flags = flags
        p.resolution, p.supersampling, p.strategy = resolution, supersampling, strategy
        ut = (1, 0, 0, 0, 1, 0, 0, 0, 1) if unit_transform is None else tuple(int(x) for x in np.ravel(unit_transform))
        p.unit_transform = (C.c_int32 * 9)(*ut)
        if bounds is not None:
            p.bounds_known = 1
            p.bounds = (C.c_float * 6)(*[float(x) for x in np.ravel(bounds)])
        p.z_begin, p.z_end = zslab
        return p

    def plan_slabs(self, resolution, n_slabs, *, supersampling=1, unit_transform=None, bounds=None):
        """o2v_hip_plan_slabs: (cuts, bounds) -- n_slabs+1 ascending z cuts that equalise the predicted work per slab,
        and the mesh bounds (float32 [6]) to hand back to voxelize(bounds=...)."""
        p = self._params(resolution, supersampling, 0, unit_transform, bounds, (0, 0))
        cuts = np.zeros(n_slabs + 1, dtype=np.uint32)
        bnd = np.zeros(6, dtype=np.float32)
        self._check(self._L.o2v_hip_plan_slabs(self._ctx, C.byref(p), n_slabs, _ptr(cuts), _ptr(bnd)), "o2v_hip_plan_slabs")
        return [int(z) for z in cuts], bnd

    def voxelize(self, resolution, *, supersampling=1, strategy=STRATEGY_MAX, unit_transform=None, bounds=None,
                 zslab=(0, 0), read=True, exact_clip=False, kernel_times=False, stage_times=False, xtile=(0, 0), ytile=(0, 0)):
        """xtile / ytile: an x / y range of the output grid (o2v_hip_params::x_begin ..; begin a multiple of 4), like zslab."""
        flags = (FLAG_EXACT_CLIP if exact_clip else 0) | (FLAG_KERNEL_TIMES if kernel_times else 0) | (FLAG_STAGE_TIMES if stage_times else 0)
        if tuple(xtile) != (0, 0) or tuple(ytile) != (0, 0):
            p = self._params(resolution, supersampling, strategy, unit_transform, bounds, zslab, flags, tuple(xtile), tuple(ytile))
        elif unit_transform is None and bounds is None:
            # (a loop of identical calls - bench.py's timed steps - does not build the parameter block again every time)
            key = (resolution, supersampling, strategy, zslab, flags)
            if getattr(self, "_plain_key", None) != key:
                self._plain_key, self._plain_params = key, self._params(resolution, supersampling, strategy, None, None, zslab, flags)
            p = self._plain_params
        else:
            p = self._params(resolution, supersampling, strategy, unit_transform, bounds, zslab, flags)
        n = self._n_out
        self._check(self._L.o2v_hip_voxelize(self._ctx, C.byref(p), C.byref(n)), "o2v_hip_voxelize")
        self.count = n.value
        if not read:
            return self.count
        return self.read_voxels()

    def voxelize_sharded(self, comm, resolution, *, supersampling=1, strategy=STRATEGY_MAX, unit_transform=None, bounds=None,
                         read=True, stage_times=False):
        """o2v_hip_voxelize_sharded: collective over `comm` (a Comm); this rank voxelizes its planned z-slab.
        Returns (voxels or count of this rank, counts of all ranks, z cuts)."""
        flags = FLAG_STAGE_TIMES if stage_times else 0
        # (a loop of identical calls - bench.py's timed steps - reuses the parameter block and the two small result arrays)
        key = (resolution, supersampling, strategy, flags, comm.world) if unit_transform is None and bounds is None else None
        if key is None or getattr(self, "_sharded_key", None) != key:
            self._sharded_key = key
            self._sharded_params = self._params(resolution, supersampling, strategy, unit_transform, bounds, (0, 0), flags)
            self._sharded_counts = np.zeros(comm.world, dtype=np.uint64)
            self._sharded_cuts = np.zeros(comm.world + 1, dtype=np.uint32)
            self._sharded_ptrs = (_ptr(self._sharded_counts), _ptr(self._sharded_cuts))
        p, n, counts, cuts = self._sharded_params, self._n_out, self._sharded_counts, self._sharded_cuts
        self._check(self._L.o2v_hip_voxelize_sharded(self._ctx, comm.handle, C.byref(p), C.byref(n), *self._sharded_ptrs),
                    "o2v_hip_voxelize_sharded")
        self.count = n.value
        return (self.read_voxels() if read else self.count), counts.tolist(), cuts.tolist()

    def read_voxels(self):
        out = np.empty((self.count, 4), dtype=np.uint32)
        if self.count:
            self._check(self._L.o2v_hip_read_voxels(self._ctx, _ptr(out), 0, self.count), "o2v_hip_read_voxels")
        return out

    def timings(self):
        t = Timings()
        self._L.o2v_hip_get_timings(self._ctx, C.byref(t))
        return t.as_dict()

    def kernel_times(self):
        """{kernel name: (ms, launches)} of the last voxelize(kernel_times=True) call."""
        buf = (KernelTime * 64)()
        n = C.c_uint32(0)
        self._L.o2v_hip_get_kernel_times(self._ctx, buf, 64, C.byref(n))
        return {buf[i].name.decode(): (float(buf[i].ms), int(buf[i].launches)) for i in range(min(n.value, 64))}

    def stats(self):
        s = Stats()
        self._L.o2v_hip_get_stats(self._ctx, C.byref(s))
        return s.as_dict()

    def debug_counters(self):
        out = np.zeros(16, dtype=np.uint64)
        self._L.o2v_hip_debug_counters(self._ctx, _ptr(out))
        return out

    def hits(self):
        """Every hit record of the last run (general route): uint32 array [n, 8] = cell x, y, z, keyhi, keylo, bits of w, u, v."""
        n = C.c_uint64(0)
        self._check(self._L.o2v_hip_debug_hits(self._ctx, None, 0, C.byref(n)), "o2v_hip_debug_hits")
        out = np.zeros((n.value, 8), dtype=np.uint32)
        if n.value:
            self._check(self._L.o2v_hip_debug_hits(self._ctx, _ptr(out), n.value, C.byref(n)), "o2v_hip_debug_hits")
        return out

    def check_third(self):
        """(differing inputs, first of them or None) of x / 3 against the clip loop's short form, all 2^32 float32 patterns."""
        out = np.zeros(2, dtype=np.uint64)
        self._check(self._L.o2v_hip_debug_check_third(self._ctx, _ptr(out)), "o2v_hip_debug_check_third")
        return int(out[0]), (int(out[1]) - 1 if out[0] else None)

    def check_div(self, samples=1024, seed=1):
        """256 x 256 table [numerator's biased exponent, divisor's]: pairs out of `samples` whose lean quotient differs from n / d."""
        out = np.zeros(65536, dtype=np.uint32)
        self._check(self._L.o2v_hip_debug_check_div(self._ctx, samples, seed, _ptr(out)), "o2v_hip_debug_check_div")
        return out.reshape(256, 256)

    def transform(self):
        out = np.zeros(12, dtype=np.float32)
        self._L.o2v_hip_get_transform(self._ctx, _ptr(out))
        return out


class _Callbacks(C.Structure):
    _fields_ = [("user", C.c_void_p),
                ("allreduce_min_u32", C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.c_size_t)),
                ("allreduce_max_u32", C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.c_size_t)),
                ("allreduce_sum_u64", C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t)),
                ("allgather", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)),
                ("broadcast", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int))]


class Comm:
    """The collectives of the sharded voxelization (include/o2v_hip.h, multi-GPU section) for one rank."""

    def __init__(self, handle, rank, world, keep=None):
        self._L = _bind()
        self.handle, self.rank, self.world, self._keep = handle, rank, world, keep

    @property
    def kind(self):
        return self._L.o2v_hip_comm_kind(self.handle).decode()

    def close(self):
        if self.handle:
            self._L.o2v_hip_comm_destroy(self.handle)
            self.handle = None

    @staticmethod
    def unique_id():
        """ncclGetUniqueId through the library: 128 bytes that rank 0 ships to every rank."""
        buf = (C.c_uint8 * 128)()
        if _bind().o2v_hip_comm_unique_id(buf) != 0:
            raise DeviceError("o2v_hip_comm_unique_id failed: librccl is not available")
        return bytes(buf)

    @classmethod
    def rccl(cls, unique_id, rank, world, device):
        """RCCL over xGMI: every rank calls this with rank 0's unique id (blocks until all have joined)."""
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        rc = _bind().o2v_hip_comm_create_rccl(buf, rank, world, device, C.byref(h))
        if rc != 0:
            raise DeviceError(f"o2v_hip_comm_create_rccl failed with code {rc}")
        return cls(h, rank, world)

    @classmethod
    def torch_distributed(cls, dist):
        """Host-memory collectives over an initialised torch.distributed group (gloo): for the CPU-side tests of the
        N > 1 path and for ranks that share one GPU, where RCCL cannot be used.  With the nccl backend the host
        buffers travel through device tensors (bench.py's fallback if the library's own communicator cannot be made)."""
        import torch
        rank, world = dist.get_rank(), dist.get_world_size()
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"

        def array(ptr, n, ctype):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,))

        def guarded(fn):
            # ctypes prints and swallows an exception raised inside a callback and the C side would see 0 = success: the
            # library would then plan its slabs from un-reduced data.  Any failure is reported as a failed collective.
            def wrapper(*args):
                try:
                    fn(*args)
                    return 0
                except BaseException as e:  # noqa: BLE001 - must not propagate into the C caller
                    import sys
                    print(f"obj2voxel_amd: collective callback failed: {type(e).__name__}: {e}", file=sys.stderr)
                    return 1
            return wrapper

        def allreduce(op, ctype):
            def fn(user, buf, n):
                a = array(buf, n, ctype)
                t = torch.from_numpy(a.astype(np.int64)).to(dev)  # gloo has no unsigned reductions; the values fit int64
                dist.all_reduce(t, op=op)
                a[:] = t.cpu().numpy().astype(a.dtype)
            return guarded(fn)

        @guarded
        def allgather(user, buf, bytes_per_rank):
            a = array(buf, bytes_per_rank * world, C.c_uint8)
            parts = [torch.empty(bytes_per_rank, dtype=torch.uint8, device=dev) for _ in range(world)]
            dist.all_gather(parts, torch.from_numpy(a[rank * bytes_per_rank:(rank + 1) * bytes_per_rank].copy()).to(dev))
            a[:] = torch.cat(parts).cpu().numpy()

        @guarded
        def broadcast(user, buf, n, root):
            a = array(buf, n, C.c_uint8)
            t = torch.from_numpy(a.copy()).to(dev)
            dist.broadcast(t, src=root)
            a[:] = t.cpu().numpy()

        F = dict(_Callbacks._fields_)
        cb = _Callbacks(None,
                        F["allreduce_min_u32"](allreduce(dist.ReduceOp.MIN, C.c_uint32)),
                        F["allreduce_max_u32"](allreduce(dist.ReduceOp.MAX, C.c_uint32)),
                        F["allreduce_sum_u64"](allreduce(dist.ReduceOp.SUM, C.c_uint64)),
                        F["allgather"](allgather), F["broadcast"](broadcast))
        h = C.c_void_p()
        rc = _bind().o2v_hip_comm_create_callbacks(C.byref(cb), rank, world, C.byref(h))
        if rc != 0:
            raise DeviceError(f"o2v_hip_comm_create_callbacks failed with code {rc}")
        return cls(h, rank, world, keep=cb)


class DeviceGroup:
    """o2v_hip_group: one process, one context and host thread per listed GPU, grid sharded by z-slab."""

    UPLOAD_H2D, UPLOAD_BROADCAST, UPLOAD_PEER = 0, 1, 2

    def __init__(self, devices):
        self._L = _bind()
        self._g = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        rc = self._L.o2v_hip_group_create(arr, len(devices), C.byref(self._g))
        if rc != 0:
            raise DeviceError(f"o2v_hip_group_create({list(devices)}) failed with code {rc}")
        self.size = len(devices)
        self.ranks = [DeviceVoxelizer(_borrowed_ctx=self._L.o2v_hip_group_ctx(self._g, r)) for r in range(self.size)]

    @property
    def comm_kind(self):
        return self._L.o2v_hip_group_comm_kind(self._g).decode()

    def close(self):
        if self._g:
            self._L.o2v_hip_group_destroy(self._g)
            self._g = C.c_void_p()

    def _check(self, rc, what):
        if rc != 0:
            raise DeviceError(f"{what} failed with code {rc}: {self._L.o2v_hip_group_last_error(self._g).decode()}")

    def set_triangles(self, verts, uvs=None, types=None, colors=None, texids=None, upload=0):
        verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 9)
        T = verts.shape[0]
        uvs = None if uvs is None else np.ascontiguousarray(uvs, dtype=np.float32).reshape(T, 6)
        types = None if types is None else np.ascontiguousarray(types, dtype=np.uint32).reshape(T)
        colors = None if colors is None else np.ascontiguousarray(colors, dtype=np.float32).reshape(T, 3)
        texids = None if texids is None else np.ascontiguousarray(texids, dtype=np.int32).reshape(T)
        self._check(self._L.o2v_hip_group_set_triangles(self._g, _ptr(verts), _ptr(uvs), _ptr(types), _ptr(colors),
                                                        _ptr(texids), T, upload), "o2v_hip_group_set_triangles")

    def set_textures(self, textures):
        arr = (_Texture * max(1, len(textures)))()
        keep = []
        for i, (pix, wrap) in enumerate(textures):
            pix = np.ascontiguousarray(pix, dtype=np.uint8)
            keep.append(pix)
            h, w, c = pix.shape
            arr[i] = _Texture(pix.ctypes.data, w, h, c, int(wrap))
        self._check(self._L.o2v_hip_group_set_textures(self._g, C.cast(arr, C.c_void_p), len(textures)), "o2v_hip_group_set_textures")

    def voxelize(self, resolution, *, supersampling=1, strategy=STRATEGY_MAX, unit_transform=None, bounds=None, read=True,
                 stage_times=False):
        """Returns (list of per-rank voxel arrays, or the per-rank counts if read=False; z cuts)."""
        p = DeviceVoxelizer._params(resolution, supersampling, strategy, unit_transform, bounds, (0, 0), FLAG_STAGE_TIMES if stage_times else 0)
        counts = np.zeros(self.size, dtype=np.uint64)
        cuts = np.zeros(self.size + 1, dtype=np.uint32)
        self._check(self._L.o2v_hip_group_voxelize(self._g, C.byref(p), _ptr(counts), _ptr(cuts)), "o2v_hip_group_voxelize")
        for r, d in enumerate(self.ranks):
            d.count = int(counts[r])
        cuts = [int(z) for z in cuts]
        if not read:
            return [int(c) for c in counts], cuts
        return [d.read_voxels() for d in self.ranks], cuts
