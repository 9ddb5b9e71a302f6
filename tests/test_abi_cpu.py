"""CPU-side checks of the drop-in boundary: libvoxelba.so builds/loads, exports every symbol declared in
include/voxelba.h, fails loudly without a GPU, and the host-side (non-device) entry points agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    import voxel_slam_amd  # noqa: F401
    from voxel_slam_amd import capi as m
    if not os.path.exists(m.LIB_PATH):
        m.build()
    return m


def test_exports_match_header(capi):
    hdr = open(os.path.join(ROOT, "include", "voxelba.h")).read()
    declared = set(re.findall(r"\b(vba_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"vba_allreduce_fn"}
    lib = capi.load()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)


def test_no_device_is_a_loud_error(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    o = capi.default_options()
    with pytest.raises(capi.VbaError) as e:
        capi.Context(o)
    assert e.value.status == capi.ERR_NO_DEVICE


def test_default_options_are_avia_yaml(capi):
    o = capi.default_options()
    assert (o.win_size, o.max_layer, o.max_points, o.thread_num) == (10, 2, 100, 5)
    assert o.min_eigen_value == 0.0025 and o.imu_coef == 1e-4 and list(o.plane_eigen_value_thre) == [0.25] * 4


def test_shard_owner_is_a_partition(capi):
    rng = np.random.default_rng(0)
    keys = rng.integers(-200, 200, (4000, 3))
    for n in (1, 2, 4, 8):
        own = np.array([capi.shard_owner(k, n) for k in keys])
        assert own.min() >= 0 and own.max() < n
        cnt = np.bincount(own, minlength=n)
        assert cnt.min() > 0.7 * len(keys) / n      # roughly balanced contiguous bucket ranges


def _imu_inputs(seed=0):
    rng = np.random.default_rng(seed)
    n = 21
    t = np.arange(n) / 200.0
    gyr = rng.normal(0, 0.2, (n, 3)); acc = rng.normal(0, 0.5, (n, 3)) + np.array([0, 0, 9.8])
    bg = rng.normal(0, 0.01, 3); ba = rng.normal(0, 0.05, 3)
    nm = np.array([0.01] * 3 + [1.0] * 3); nw = np.array([1e-4] * 6)
    return t, gyr, acc, bg, ba, nm, nw


def test_host_imu_preintegration_matches_oracle(capi, oracle):
    args = _imu_inputs()
    a = capi.imu_preintegrate(*args)
    b = oracle.imu_preintegrate(*args)
    assert np.allclose(a, b, rtol=1e-12, atol=1e-15)


def test_host_imu_give_evaluate_matches_oracle(capi, oracle):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(3)
    imu = capi.imu_preintegrate(*_imu_inputs(1))
    imu[67:73] = rng.normal(0, 1e-3, 6)      # dbg, dba
    def state(k):
        s = np.zeros(25)
        s[1:10] = Rotation.from_rotvec(rng.normal(0, 0.3, 3)).as_matrix().ravel()
        s[10:13] = rng.normal(0, 1, 3); s[13:16] = rng.normal(0, 1, 3)
        s[16:19] = rng.normal(0, 0.01, 3); s[19:22] = rng.normal(0, 0.05, 3); s[22:25] = [0, 0, -9.8]
        return s
    s1, s2 = state(0), state(1)
    for with_g in (False, True):
        ra, Ja, ga = capi.imu_give_evaluate(imu, s1, s2, with_g, True)
        rb, Jb, gb = oracle.imu_give_evaluate(imu, s1, s2, with_g, True)
        assert ra == pytest.approx(rb, rel=1e-10)
        assert np.allclose(Ja, Jb, rtol=1e-9, atol=1e-9 * np.abs(Jb).max())
        assert np.allclose(ga, gb, rtol=1e-9, atol=1e-9 * np.abs(gb).max())
