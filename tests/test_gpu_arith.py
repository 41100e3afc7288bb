"""The clip loop's short division forms on the device (obj2voxel_amd/csrc/o2v_dev_arith.hpp) against the compiler's
IEEE 754 division: x / 3 for every float32, and n / d over every pair of exponents - the kernels use the lean form only in
a region the map shows to be free of differences (with room to spare on every side)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dv():
    from obj2voxel_amd import hip
    d = hip.DeviceVoxelizer(0)
    yield d
    d.close()


def test_third_is_exact_for_every_float32(dv):
    bad, first = dv.check_third()
    assert bad == 0, (bad, hex(first))


def test_lean_division_agrees_inside_its_box(dv):
    m = dv.check_div(samples=16384, seed=7)
    # rows: the numerator's biased exponent, columns: the divisor's (0 = zero / subnormal, 255 = inf / NaN)
    # What k_voxelize feeds it (o2v_dev_k2_voxelize.hpp, "lean divisions"):
    #   cut parameter: n, d in [2^-16, 2^18]                                            -> biased 111 .. 145 both
    #   uv mean:       d in [2^-30, 2^37]; n = q d with 2^-50 <= |q| <= 2^21 (or n = +0)  -> d 97 .. 164, n 46 .. 185
    # asserted with 10 binades of margin on every side of the union, quotients 2^-70 .. 2^70 (the form starts to differ
    # where the numerator drops below ~2^-103, the divisor leaves 2^+-~100 or the quotient approaches the subnormals)
    en, ed = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    box = (en >= 36) & (en <= 195) & (ed >= 87) & (ed <= 174) & (np.abs(en - ed) <= 70)
    assert int(m[box].sum()) == 0, np.argwhere(box & (m > 0))[:5]
    # the form is NOT generally exact: outside the middle of the range it must differ somewhere, or this test tests nothing
    assert int(m.sum()) > 0
    # +0 numerators (row 0 holds zeros and subnormals; the all-zero mantissa with sign + is the exact zero): covered by the
    # workloads' untextured triangles (uv = 0), see tests/test_gpu_exact_ab.py


def test_lean_division_map_extent(dv):
    """For the record (printed with -s): the largest centred square of exponent pairs without a difference."""
    m = dv.check_div(samples=256, seed=3)
    k = 0
    while k < 127 and int(m[127 - k - 1:127 + k + 2, 127 - k - 1:127 + k + 2].sum()) == 0:
        k += 1
    print(f"lean division: no difference for biased exponents within {k} of 127 (both operands)")
    assert k >= 60
