"""Both sides against fixtures made by implementations that share no code with them (tests/golden/make_indep_fixtures.py, run in the
build container only): scipy.spatial.cKDTree nearest neighbours and a float64 numpy Gauss-Newton ICP (own NN, own normals, own
linearisation, numpy.linalg.solve). SURVEY.md §8c items 1-2: with no reference source for the path (parity unpinned), these are what
keeps the oracle and the HIP path from being wrong together. The CPU tests pin the ORACLE, the gpu tests the HIP path — on the GPU box
the fixtures are the only independent thing there is (no scipy needed: the files are data).

Tolerances (stated here, used nowhere else): NN index exact wherever the true NN is inside one voxel edge and unique by 1e-5 (relative);
NN distance rtol 1e-5 / atol 1e-6 (float32 fma chain against float64); final pose ||T - T_indep||_F <= 1e-4."""
import json
import os

import numpy as np
import pytest

from mandala_mapping_amd import abi, synth

HERE = os.path.dirname(os.path.abspath(__file__))
NN = np.load(os.path.join(HERE, "golden", "nn_ckdtree_v1.npz"))
ICP = json.load(open(os.path.join(HERE, "golden", "indep_icp_v1.json")))["cases"]
POSE_TOL = 1e-4


def _check_nn(nn_of_cloud, name):
    tgt, q = NN[name + "_target"], NN[name + "_queries"]
    ik, dk, d2nd = NN[name + "_idx"], NN[name + "_d"], NN[name + "_d2nd"]
    leaf, dmax = (float(x) for x in NN[name + "_leaf_dmax"])
    idx, d2 = nn_of_cloud(tgt, q, leaf, dmax)
    d = np.sqrt(d2.astype(np.float64))
    # (1) the true NN closer than one voxel edge (and than d_max) lies inside the 27 voxels: the spec's answer IS the true NN
    near = (dk < 0.999 * leaf) & (dk < dmax * (1 - 1e-6))
    assert near.sum() > len(q) // 4, near.sum()
    assert (idx[near] >= 0).all()
    assert np.allclose(d[near], dk[near], rtol=1e-5, atol=1e-6)
    unique = near & (d2nd > dk * (1 + 1e-5) + 1e-6)
    assert unique.sum() > 0.99 * near.sum()
    assert np.array_equal(idx[unique], ik[unique])
    # (2) everywhere else a match can only be farther than the true NN, never nearer, and never beyond d_max
    m = idx >= 0
    assert (d[m] >= dk[m] * (1 - 1e-5) - 1e-6).all()
    assert (d[m] <= dmax * (1 + 1e-6)).all()
    # (3) no match only where the true NN is farther than one voxel edge or than d_max
    assert not (near & ~m).any()
    return int(unique.sum())


def _params(c):
    return abi.Params.make(leaf=c["leaf"], iterations=c["iters"], max_corr_dist=c["dmax"], metric=abi.POINT_TO_PLANE if c["plane"] else abi.POINT_TO_POINT,
                           normal_leaf=c["normal_leaf"], eps_rot=0.0, eps_trans=0.0)


@pytest.mark.parametrize("name", ["planes", "hdl32"])
def test_oracle_nn_against_ckdtree_fixture(orc, name):
    def nn(tgt, q, leaf, dmax):
        return orc.Cloud(abi.Params.make(leaf=leaf, metric=abi.POINT_TO_POINT), tgt).nn(q, dmax)
    assert _check_nn(nn, name) > 1000


@pytest.mark.parametrize("name", sorted(ICP))
def test_oracle_pose_against_independent_icp(orc, name):
    c = ICP[name]
    assert c["residual_step"] < 1e-9     # the independent ICP had converged: a fixed point is being compared, not a trajectory
    src, tgt, _ = eval("synth." + c["gen"])
    p = _params(c)
    T, st, _ = orc.align(p, orc.Cloud(p, src, omp=True), orc.Cloud(p, tgt, omp=True))
    assert st.n_corr == c["n_corr"]
    assert np.linalg.norm(np.asarray(T, np.float64) - np.array(c["T"])) <= POSE_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["planes", "hdl32"])
def test_hip_nn_against_ckdtree_fixture(reg, name):
    def nn(tgt, q, leaf, dmax):
        R = reg.Registrar(abi.Params.make(leaf=leaf, metric=abi.POINT_TO_POINT))
        return R.cloud(tgt).nn(q, dmax)
    assert _check_nn(nn, name) > 1000


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(ICP))
def test_hip_pose_against_independent_icp(reg, name):
    c = ICP[name]
    src, tgt, _ = eval("synth." + c["gen"])
    R = reg.Registrar(_params(c))
    T, st = R.align(R.cloud(src), R.cloud(tgt))
    assert st.n_corr == c["n_corr"]
    assert np.linalg.norm(np.asarray(T, np.float64) - np.array(c["T"])) <= POSE_TOL
