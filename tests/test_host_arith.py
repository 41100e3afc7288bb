"""x / 3 in float32 equals the double product with 1/3 rounded back to float32 (third() of
obj2voxel_amd/csrc/o2v_dev_arith.hpp) - here on the host, in numpy, for every exponent and sign with 2^16 mantissas each
plus the special values; the device checks all 2^32 patterns itself (tests/test_gpu_arith.py)."""
import numpy as np


def test_third_by_double_product_matches_float_division():
    rng = np.random.default_rng(5)
    mant = np.concatenate([rng.integers(0, 1 << 23, size=(1 << 16) - 4, dtype=np.uint32), np.array([0, 1, (1 << 23) - 1, 1 << 22], np.uint32)])
    bad = 0
    with np.errstate(all="ignore"):
        for sign in (0, 1):
            for e in range(256):
                bits = (np.uint32(sign) << np.uint32(31)) | (np.uint32(e) << np.uint32(23)) | mant
                x = bits.view(np.float32)
                want = x / np.float32(3.0)
                got = (x.astype(np.float64) * (1.0 / 3.0)).astype(np.float32)
                same = (want.view(np.uint32) == got.view(np.uint32)) | (np.isnan(want) & np.isnan(got))
                bad += int((~same).sum())
    assert bad == 0
