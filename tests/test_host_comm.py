"""Host-only checks of the communicator layer (obj2voxel_amd/csrc/o2v_comm.cpp): no GPU needed."""
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, **env):
    e = dict(os.environ, **env)
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=e, capture_output=True, text=True, timeout=120)


def test_rccl_load_failure_is_reported_not_fatal():
    """librccl missing: o2v_hip_comm_unique_id must fail with a message (it used to crash: dlerror() was called twice and
    the second call returns NULL).  O2V_RCCL_LIB names the library to load, here one that does not exist."""
    code = (
        "import ctypes as C\n"
        "from obj2voxel_amd import hip\n"
        "L = hip._bind()\n"
        "buf = (C.c_uint8 * 128)()\n"
        "rc = L.o2v_hip_comm_unique_id(buf)\n"
        "assert rc != 0, rc\n"
        "try:\n"
        "    hip.Comm.unique_id()\n"
        "except hip.DeviceError as e:\n"
        "    print('raised:', e)\n"
        "else:\n"
        "    raise SystemExit('no error raised')\n"
        "h = C.c_void_p()\n"
        "rc = L.o2v_hip_comm_create_rccl(buf, 0, 1, 0, C.byref(h))\n"
        "assert rc != 0 and not h.value, (rc, h.value)\n"
        "print('ok')\n"
    )
    r = _run(code, O2V_RCCL_LIB="/nonexistent/librccl-missing.so")
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "raised:" in r.stdout and r.stdout.strip().endswith("ok")


def _mock_rccl(tmp_path):
    """tests/mock/mock_rccl.c: a librccl stand-in whose collectives run between threads on host memory (gcc is here)."""
    so = tmp_path / "libmock_rccl.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", os.path.join(ROOT, "tests", "mock", "mock_rccl.c"), "-o", str(so), "-lpthread"],
                   check=True, capture_output=True)
    return str(so)


def test_rccl_path_of_an_eight_rank_group_runs_against_a_mock_librccl(tmp_path):
    """The in-process group's RCCL code path for N = 8 - dlopen + symbol lookup, ncclGetUniqueId, ncclCommInitRank on one
    thread per rank (it blocks until all eight have joined), the five collectives the sharded run issues (min / max of the
    bounds, the 2048-bin histogram sum, the two all-gathers, the broadcast upload) with every argument as the device code
    passes it, and ncclCommDestroy - executed on a machine without a GPU, so that an 8-GPU node is not the first place this
    code runs (the GPUs of rounds 1 - 4 were single).  What stays unexecuted: RCCL itself over xGMI."""
    code = (
        "import ctypes as C\n"
        "from obj2voxel_amd import _lib\n"
        "L = _lib.lib()\n"
        "L.o2v_hip_group_rccl_selftest.argtypes = [C.c_uint32]\n"
        "for n in (1, 2, 3, 8):\n"
        "    rc = L.o2v_hip_group_rccl_selftest(n)\n"
        "    assert rc == 0, (n, rc)\n"
        "import os\n"
        "m = C.CDLL(os.environ['O2V_RCCL_LIB'])\n"
        "calls = [m.mock_rccl_calls(i) for i in range(6)]\n"
        "# 14 ranks in all: one init and one destroy each; ten rounds of 3 all-reduces, 1 all-gather, 1 broadcast per rank\n"
        "assert calls == [14, 14, 14 * 30, 14 * 10, 14 * 10, 4], calls\n"
        "print('ok')\n"
    )
    r = _run(code, O2V_RCCL_LIB=_mock_rccl(tmp_path))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout, r.stderr)


def test_rccl_selftest_reports_a_missing_library():
    code = (
        "import ctypes as C\n"
        "from obj2voxel_amd import _lib\n"
        "L = _lib.lib()\n"
        "L.o2v_hip_group_rccl_selftest.argtypes = [C.c_uint32]\n"
        "print(L.o2v_hip_group_rccl_selftest(8))\n"
    )
    r = _run(code, O2V_RCCL_LIB="/nonexistent/librccl-missing.so")
    assert r.returncode == 0 and r.stdout.strip() == "100", (r.returncode, r.stdout, r.stderr)


def test_a_rank_that_never_arrives_fails_the_others_with_a_message(tmp_path):
    """ncclCommInitRank blocks until every rank of the job has called it; on a node that is set up wrongly (a process that
    died, two ranks on one GPU, another unique id) the ranks that did arrive used to wait for ever.  With the mock librccl:
    rank 0 of a world of two is alone, O2V_COMM_TIMEOUT_S = 1 - the call returns an error, with the reason, after a second."""
    code = (
        "import ctypes as C, time\n"
        "from obj2voxel_amd import hip\n"
        "L = hip._bind()\n"
        "buf = (C.c_uint8 * 128)()\n"
        "assert L.o2v_hip_comm_unique_id(buf) == 0\n"
        "h = C.c_void_p()\n"
        "t0 = time.time()\n"
        "rc = L.o2v_hip_comm_create_rccl(buf, 0, 2, -1, C.byref(h))\n"
        "dt = time.time() - t0\n"
        "assert rc != 0 and not h.value and 0.9 < dt < 20, (rc, h.value, dt)\n"
        "print('ok')\n"
    )
    r = _run(code, O2V_RCCL_LIB=_mock_rccl(tmp_path), O2V_COMM_TIMEOUT_S="1")
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout, r.stderr)
    assert "ncclCommInitRank did not return within 1 s (rank 0 of 2)" in r.stderr
