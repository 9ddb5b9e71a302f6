"""Edge cases of the C ABI on the device: empty and tiny inputs, status codes where the reference prints and exits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(name="room20k_w4"):
    import voxel_slam_amd  # noqa: F401
    from voxel_slam_amd import capi, synth
    wl = synth.CONFIGS[name]
    return capi, synth, wl, capi.Context(capi.options_from_workload(wl))


def test_empty_scan_and_empty_map():
    capi, synth, wl, ctx = _ctx()
    W = wl.win_size
    poses = np.tile(np.concatenate([np.eye(3).ravel(), np.zeros(3)]), (W, 1))
    ctx.recut(W, poses, multi=False)                        # nothing inserted yet
    assert ctx.size() == 0 and ctx.num_roots() == 0
    ctx.cut_voxel(0, np.zeros((0, 3)), poses[0])            # empty scan
    ctx.recut(W, poses, multi=False)
    assert ctx.size() == 0
    ctx.margi(W, poses, jour=0.0); ctx.slide(1); ctx.prune(0.0, 700)
    out, cnt, first = ctx.down_sampling_voxel(np.zeros((0, 3)), 0.1)
    assert len(out) == 0
    assert len(ctx.down_sampling_close(np.zeros((0, 3)), 0.1)) == 0
    ctx.close()


def test_too_few_voxels_is_a_status_not_an_exit():
    capi, synth, wl, ctx = _ctx()
    W = wl.win_size
    poses = np.tile(np.concatenate([np.eye(3).ravel(), np.zeros(3)]), (W, 1))
    rng = np.random.default_rng(0)
    pts = np.c_[rng.uniform(-0.2, 0.2, (200, 2)), np.full(200, 1.0) + rng.normal(0, 1e-3, 200)]     # one small planar patch
    for i in range(W):
        ctx.cut_voxel(i, pts, poses[i])
    ctx.recut(W, poses, multi=False)
    assert 0 < ctx.size() < 64
    out = ctx.lidar_ba_damping_iter(poses, max_iter=3, thd_num=64)       # VM:399-403 "Too Less Voxel": g_size < thd_num
    assert out["status"] == -1
    ctx.close()


def test_single_point_and_duplicate_points():
    capi, synth, wl, ctx = _ctx()
    W = wl.win_size
    poses = np.tile(np.concatenate([np.eye(3).ravel(), np.zeros(3)]), (W, 1))
    ctx.cut_voxel(0, np.array([[1.0, 2.0, 3.0]]), poses[0])
    ctx.cut_voxel(1, np.repeat(np.array([[1.0, 2.0, 3.0]]), 50, axis=0), poses[1])       # 50 identical points: zero covariance
    ctx.recut(W, poses, multi=False)
    assert ctx.num_roots() == 1
    leaves = ctx.dump_leaves()
    assert leaves.shape[1] == 39 and np.isfinite(leaves).all()
    assert ctx.size() == 0                                                 # 51 coincident points are no plane (zero covariance, N > min_point but eig test fails or passes harmlessly)
    ctx.close()


def test_hba_window_with_an_empty_keyframe(oracle):
    import dataclasses
    capi, synth, wl, ctx = _ctx()
    s = synth.make_scans(wl)
    clouds = [p.astype(np.float32).astype(np.float64) for p in s["points"]]
    clouds[2] = np.zeros((0, 3))                                              # a keyframe that lost all its points
    poses = synth.poses_flat(s["R0"], s["p0"])
    o = ctx.opt
    cfg = oracle.gba_cfg13(2.0, 0.1, [0.25] * 4, o.voxel_size, o.min_eigen_value, list(o.plane_eigen_value_thre), o.max_layer)
    got = ctx.hba_add_edge(clouds, poses, 2.0, 0.1, [0.25] * 4, 2, 2)
    want = oracle.hba_add_edge(clouds, poses, cfg, 2, 2)
    assert want["status"] == 0
    np.testing.assert_allclose(got["poses"], want["poses"], rtol=0, atol=1e-6)
    assert len(got["edges"]) == len(want["edges"])
    np.testing.assert_array_equal(got["edges"][:, :2], want["edges"][:, :2])
    ctx.close()
