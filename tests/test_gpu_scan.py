"""Scan pre-processing on the device against the oracle: down_sampling_voxel (tools.hpp:201-238) and the
undistortion loop of motion_blur (ekf_imu.hpp:137-163)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx():
    import voxel_slam_amd  # noqa: F401
    from voxel_slam_amd import capi, synth
    return capi.Context(capi.options_from_workload(synth.CONFIGS["room20k_w4"]))


def _imu_poses(m, rng, t_end):
    from scipy.spatial.transform import Rotation
    ts = np.sort(rng.uniform(0.0, t_end, m)); ts[0] = 0.004
    out = np.zeros((m, 22))
    for j in range(m):
        out[j, 0] = ts[j]
        out[j, 1:10] = Rotation.from_rotvec(rng.normal(0, 0.05, 3)).as_matrix().ravel()
        out[j, 10:13] = rng.normal(0, 0.2, 3); out[j, 13:16] = rng.normal(0, 1.0, 3)
        out[j, 16:19] = rng.normal(0, 0.5, 3); out[j, 19:22] = rng.normal(0, 2.0, 3)
    return out


@pytest.mark.parametrize("n,voxel", [(20000, 0.1), (200000, 0.25), (5000, 0.02), (1, 0.1)])
def test_down_sampling_voxel_parity(oracle, n, voxel):
    rng = np.random.default_rng(7 + n)
    pts = rng.uniform(-20, 20, (n, 3)).astype(np.float32).astype(np.float64)
    pts[: n // 3] = (pts[: n // 3] * 0.05).astype(np.float32)          # a dense clump: many points per voxel, both signs
    ctx = _ctx()
    out, cnt, first = ctx.down_sampling_voxel(pts, voxel)
    o_out, o_cnt, o_first = oracle.down_sampling_voxel(pts, voxel)
    assert len(out) == len(o_out)
    np.testing.assert_array_equal(first, o_first)                     # same voxels, same first-occurrence order
    np.testing.assert_array_equal(cnt, o_cnt)
    assert cnt.sum() == n
    # the reference's running mean is evaluated in float (order dependent); the device forms the f64 centroid and rounds once
    np.testing.assert_allclose(out, o_out, rtol=0, atol=2e-6 * max(1.0, float(cnt.max()) ** 0.5) * 20)
    single = cnt == 1
    np.testing.assert_array_equal(out[single], o_out[single])         # untouched points are bit-identical
    ctx.close()


def test_down_sampling_voxel_small_leaf_is_identity(oracle):
    rng = np.random.default_rng(3)
    pts = rng.normal(0, 5, (1000, 3)).astype(np.float32).astype(np.float64)
    ctx = _ctx()
    out, cnt, first = ctx.down_sampling_voxel(pts, 0.0005)            # TL:203
    np.testing.assert_array_equal(out, pts)
    assert (cnt == 0).all() and (first == np.arange(1000)).all()
    ctx.close()


@pytest.mark.parametrize("n,m", [(30000, 21), (1000, 3), (64, 40)])
def test_undistort_parity(oracle, n, m):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(11 + n)
    pts = rng.uniform(-30, 30, (n, 3)).astype(np.float32).astype(np.float64)
    curv = np.sort(rng.uniform(0.0, 0.1, n)).astype(np.float32).astype(np.float64)
    curv[0] = 0.05 if n == 64 else curv[0]                            # n=64: the first point sits late -> compensated once per pose
    curv = np.sort(curv)
    ip = _imu_poses(m, rng, 0.1)
    end = np.concatenate([Rotation.from_rotvec(rng.normal(0, 0.05, 3)).as_matrix().ravel(), rng.normal(0, 0.2, 3)])
    ext = np.concatenate([Rotation.from_rotvec(rng.normal(0, 0.3, 3)).as_matrix().ravel(), rng.normal(0, 0.1, 3)])
    ctx = _ctx()
    got = ctx.undistort(pts, curv, ip, end, ext)
    want = oracle.undistort(pts, curv, ip, end, ext)
    untouched = curv <= ip[0, 0]
    np.testing.assert_array_equal(got[untouched], pts[untouched])
    assert (got != pts).any()
    # float storage: agreement to one float ulp of the coordinate magnitude
    np.testing.assert_allclose(got, want, rtol=0, atol=np.spacing(np.float32(np.abs(want).max())) * 1.01)
    assert np.mean(got == want) > 0.99
    ctx.close()


def test_down_sampling_pvec_parity(oracle):
    """voxel_map.hpp:39-83 (keyframe cloud of VS:2385): mean point and mean covariance diagonal per voxel."""
    rng = np.random.default_rng(21)
    n = 60000
    pnt = rng.uniform(-15, 15, (n, 3)); pnt[: n // 2] *= 0.1
    A = rng.normal(0, 0.02, (n, 3, 3)); var = (A @ A.transpose(0, 2, 1)).reshape(n, 9)
    ctx = _ctx()
    out, vd, cnt = ctx.down_sampling_pvec(pnt, var, 0.2)
    o_out, o_vd, o_cnt = oracle.down_sampling_pvec(pnt, var, 0.2)
    assert len(out) == len(o_out) and cnt.sum() == n
    np.testing.assert_array_equal(cnt, o_cnt)                          # same voxels in the same (first-occurrence) order
    np.testing.assert_allclose(out, o_out, rtol=0, atol=4e-6)          # running mean in double vs sum / n, then float
    np.testing.assert_allclose(vd, o_vd, rtol=2e-6, atol=1e-12)
    ctx.close()


@pytest.mark.parametrize("n,voxel", [(50000, 0.3), (3000, 0.05)])
def test_down_sampling_close_parity(oracle, n, voxel):
    """tools.hpp:240-298: the input point closest to its voxel's centroid is kept."""
    rng = np.random.default_rng(31 + n)
    pts = rng.uniform(-12, 12, (n, 3)).astype(np.float32).astype(np.float64)
    pts[: n // 3] = (pts[: n // 3] * 0.05).astype(np.float32)
    ctx = _ctx()
    keep = ctx.down_sampling_close(pts, voxel)
    o_keep = oracle.down_sampling_close(pts, voxel)
    assert len(keep) == len(o_keep)
    # the reference sums the centroid in float in input order, the device in f64: near-ties between two candidates may flip
    assert np.mean(keep == o_keep) > 0.998
    assert len(set(keep.tolist())) == len(keep) and keep.min() >= 0 and keep.max() < n
    ctx.close()
