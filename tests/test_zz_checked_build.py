"""Runs LAST (file name): when the suite was run against the diagnosis build (M3DREG_LIB=.../libm3dreg_checked.so, `make -C mandala_mapping_amd/csrc checked`), no kernel of
any test before it may have met an index outside its bound (m3d_device.h: M3D_CHK; m3dreg_debug_checks). With the shipped library the test only checks that the
report is refused — the product has no checks compiled in."""
import pytest

from mandala_mapping_amd import abi

pytestmark = pytest.mark.gpu


def test_no_index_left_its_bounds_in_this_process(reg):
    R = reg.Registrar(abi.Params.make(leaf=0.25, iterations=2, max_corr_dist=0.5, metric=abi.POINT_TO_POINT))
    try:
        c = R.checks()
    except abi.M3dregError as e:
        assert e.code == abi.ERR_INVALID_ARG     # the shipped library
        return
    assert c["icp"][0] == 0 and c["bucket"][0] == 0, c
