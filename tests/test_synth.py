"""The synthetic workloads' input contract: clouds as m3d_aggregator publishes them."""
import numpy as np

from mandala_mapping_amd import synth


def test_config4_clouds_carry_the_aggregators_self_filter_box():
    """m3d_aggregator keeps a point only when at least one coordinate lies outside its +-1 m box (m3d_aggregator.cpp:65-73,
    defaults :164-171): no config-4 cloud may hold a point inside it, and filtering is all that differs from the raw sweep."""
    k = 17                                   # this pose stands half a metre from a wall: a third of the raw sweep is inside the box
    src, tgt, T = synth.config4_pair(k, 600)
    raw_src, raw_tgt, T_raw = synth.config4_pair(k, 600, self_filter=None)
    assert np.array_equal(T, T_raw)
    for c, raw in ((src, raw_src), (tgt, raw_tgt)):
        assert (np.abs(c) > 1.0).any(axis=1).all()
        keep = (np.abs(raw) > np.float32(1.0)).any(axis=1)
        assert 0 < keep.sum() < len(raw) and np.array_equal(c, raw[keep])      # order preserved, nothing else touched
    a, b, _ = synth.config4_pair(0, 600)                                       # nothing within a metre here: the sweep is complete
    assert len(a) == len(b) == 600 * 32
