"""Property tests (hypothesis) of the registration path: statements that hold for every input, checked on generated ones.

CPU (oracle — the checker must have these properties before anything is compared with it) and GPU (the HIP path through the C ABI:
the same properties, and bit-identity with the oracle on every generated case).

* order invariance: the pose does not depend on the order the points arrive in — every sum of the spec is an exact integer, the
  NN tie-break is on the input index only between EQUAL distances (no exact ties in noisy float data): bit-identical poses;
* inverse consistency: registering B on A gives the inverse of registering A on B (two different linearisations: tolerance);
* frame equivariance: moving both clouds by one rigid motion G conjugates the result, T' = G T G^-1 (another grid: tolerance);
* degenerate clouds (collinear, coplanar with point-to-point, a handful of points, duplicates) end in a status code, never in a
  crash, a NaN pose or a hang.
"""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from mandala_mapping_amd import abi, synth

SETTINGS = dict(max_examples=6, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))


def _params(metric, leaf=0.25, iterations=8):
    return abi.Params.make(leaf=leaf, iterations=iterations, max_corr_dist=2.0 * leaf, metric=metric, normal_leaf=2.0 * leaf,
                           eps_rot=0.0, eps_trans=0.0)


def _pair(seed, n=4000, deg=1.5, trans=0.08):
    rng = np.random.default_rng(seed)
    tgt = synth.planes_cloud(n, seed, sigma=0.003, size=4.0).astype(np.float32)
    T = synth.random_T(rng, deg, trans)
    src = synth.apply_T(synth.inv_T(T), synth.planes_cloud(n, seed + 1000, sigma=0.003, size=4.0)).astype(np.float32)
    return src, tgt, T


class _Oracle:
    def __init__(self, orc):
        self.orc = orc

    def align(self, p, src, tgt, init=None):
        T, s, _ = self.orc.align(p, self.orc.Cloud(p, src, source_only=True), self.orc.Cloud(p, tgt), init_T=init)
        return T, s


class _Hip:
    def __init__(self, reg):
        self.reg = reg

    def align(self, p, src, tgt, init=None):
        R = self.reg.Registrar(p)
        T, s = R.align(R.cloud(src), R.cloud(tgt), init)
        return T, s


def _order_invariance(impl, seed, metric):
    src, tgt, _ = _pair(seed)
    p = _params(metric)
    rng = np.random.default_rng(seed + 7)
    T0, s0 = impl.align(p, src, tgt)
    T1, s1 = impl.align(p, src[rng.permutation(len(src))], tgt[rng.permutation(len(tgt))])
    assert np.array_equal(T0, T1)
    assert (s0.status, s0.iterations, s0.n_corr, s0.rms) == (s1.status, s1.iterations, s1.n_corr, s1.rms)


def _inverse_consistency(impl, seed, metric):
    src, tgt, Tgt = _pair(seed, deg=1.0, trans=0.05)
    p = _params(metric, iterations=15)
    Tf, _ = impl.align(p, src, tgt)
    Tb, _ = impl.align(p, tgt, src)
    # (the two clouds are independent samples of the same three planes, ~10 cm apart on average: the tolerance is the sampling's, not
    # the arithmetic's)
    r, t = synth.pose_error(Tb, synth.inv_T(Tf))
    assert r < 0.3 and t < 0.03, (r, t)
    r, t = synth.pose_error(Tf, Tgt)
    assert r < 0.3 and t < 0.03, (r, t)


def _frame_equivariance(impl, seed, metric):
    src, tgt, _ = _pair(seed, deg=1.0, trans=0.05)
    p = _params(metric, iterations=15)
    G = synth.random_T(np.random.default_rng(seed + 99), 40.0, 3.0)
    T, _ = impl.align(p, src, tgt)
    Tg, _ = impl.align(p, synth.apply_T(G, src).astype(np.float32), synth.apply_T(G, tgt).astype(np.float32))
    r, t = synth.pose_error(Tg, G @ np.asarray(T, np.float64) @ synth.inv_T(G))
    assert r < 0.3 and t < 0.04, (r, t)


def _degenerate(impl, seed, metric):
    rng = np.random.default_rng(seed)
    line = np.stack([np.linspace(-3, 3, 400), np.zeros(400), np.zeros(400)], 1).astype(np.float32)
    plane = np.concatenate([rng.uniform(-3, 3, (800, 2)), np.zeros((800, 1))], 1).astype(np.float32)
    few = rng.uniform(-1, 1, (int(rng.integers(1, 6)), 3)).astype(np.float32)
    dup = np.repeat(rng.uniform(-1, 1, (3, 3)).astype(np.float32), 50, axis=0)
    p = _params(metric, leaf=0.2, iterations=4)
    for cloud in (line, plane, few, dup):
        T, s = impl.align(p, cloud + np.float32(0.01), cloud)
        assert s.status in (abi.CONVERGED, abi.MAX_ITERATIONS, abi.TOO_FEW_CORR, abi.RANK_DEFICIENT, abi.DIVERGED)
        assert np.isfinite(np.asarray(T)).all()
        assert 0 <= s.iterations <= 4


# ---- CPU: the oracle ---------------------------------------------------------------------------------
@settings(**SETTINGS)
@given(seed=st.integers(0, 10_000), metric=st.sampled_from([abi.POINT_TO_POINT, abi.POINT_TO_PLANE]))
def test_oracle_pose_does_not_depend_on_point_order(orc, seed, metric):
    _order_invariance(_Oracle(orc), seed, metric)


@settings(**SETTINGS)
@given(seed=st.integers(0, 10_000), metric=st.sampled_from([abi.POINT_TO_POINT, abi.POINT_TO_PLANE]))
def test_oracle_inverse_consistency(orc, seed, metric):
    _inverse_consistency(_Oracle(orc), seed, metric)


@settings(**SETTINGS)
@given(seed=st.integers(0, 10_000), metric=st.sampled_from([abi.POINT_TO_POINT, abi.POINT_TO_PLANE]))
def test_oracle_frame_equivariance(orc, seed, metric):
    _frame_equivariance(_Oracle(orc), seed, metric)


@settings(**SETTINGS)
@given(seed=st.integers(0, 10_000), metric=st.sampled_from([abi.POINT_TO_POINT, abi.POINT_TO_PLANE]))
def test_oracle_degenerate_clouds_end_in_a_status(orc, seed, metric):
    _degenerate(_Oracle(orc), seed, metric)


# ---- GPU: the HIP path, same properties + bit-identity with the oracle on every generated case -----------
@pytest.mark.gpu
@settings(**SETTINGS)
@given(seed=st.integers(0, 10_000), metric=st.sampled_from([abi.POINT_TO_POINT, abi.POINT_TO_PLANE]))
def test_hip_pose_does_not_depend_on_point_order_and_equals_oracle(reg, orc, seed, metric):
    _order_invariance(_Hip(reg), seed, metric)
    src, tgt, _ = _pair(seed)
    p = _params(metric)
    Th, sh = _Hip(reg).align(p, src, tgt)
    To, so = _Oracle(orc).align(p, src, tgt)
    assert np.array_equal(Th, To) and (sh.status, sh.iterations, sh.n_corr) == (so.status, so.iterations, so.n_corr)


@pytest.mark.gpu
@settings(**SETTINGS)
@given(seed=st.integers(0, 10_000), metric=st.sampled_from([abi.POINT_TO_POINT, abi.POINT_TO_PLANE]))
def test_hip_inverse_consistency_and_frame_equivariance(reg, seed, metric):
    _inverse_consistency(_Hip(reg), seed, metric)
    _frame_equivariance(_Hip(reg), seed, metric)


@pytest.mark.gpu
@settings(**SETTINGS)
@given(seed=st.integers(0, 10_000), metric=st.sampled_from([abi.POINT_TO_POINT, abi.POINT_TO_PLANE]))
def test_hip_degenerate_clouds_end_in_a_status_like_the_oracle(reg, orc, seed, metric):
    _degenerate(_Hip(reg), seed, metric)
    rng = np.random.default_rng(seed)
    plane = np.concatenate([rng.uniform(-3, 3, (800, 2)), np.zeros((800, 1))], 1).astype(np.float32)
    p = _params(metric, leaf=0.2, iterations=4)
    Th, sh = _Hip(reg).align(p, plane + np.float32(0.01), plane)
    To, so = _Oracle(orc).align(p, plane + np.float32(0.01), plane)
    assert np.array_equal(Th, To) and sh.status == so.status and sh.n_corr == so.n_corr


@pytest.mark.gpu
@pytest.mark.parametrize("batch", ["0", "1"])
def test_random_scenes_and_parameters_equal_the_oracle(batch):
    """scripts/r6_fuzz.py for a few seconds: random scenes (walls, floors, blobs denser than a voxel, lines, boxes, exact duplicates, NaN / inf points, 1 ... 60 000 points, extents
    1 ... 80 m) under random parameters (one to three levels, both metrics, thresholds on / off, random initial poses) — error codes, poses, statistics, every level's exported
    sort and every tile image, bit for bit against the CPU oracle; batch = "1": heterogeneous batches through the asynchronous creation + m3dreg_align_batch."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FUZZ_BATCH=batch)
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "r6_fuzz.py"), "25" if batch == "0" else "30", "424242"], capture_output=True, text=True, timeout=400, env=env, cwd=root)
    assert r.returncode == 0 and ", 0 with differences" in r.stdout, (r.stdout[-1500:], r.stderr[-800:])
