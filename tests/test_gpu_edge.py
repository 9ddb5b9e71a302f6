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


def test_window_with_an_empty_frame_and_a_sparse_frame(oracle):
    """Ragged window: frame 1 has no points at all, frame 2 a twentieth of a scan — the frame's 6x6 diagonal block of H is then
    zero / weak, which the LDL^T handles as Eigen does (zero pivot: the component of the solution is 0, VM:458)."""
    capi, synth, wl, ctx = _ctx()
    s = synth.make_scans(wl)
    W = wl.win_size
    poses = synth.poses_flat(s["R0"], s["p0"])
    pts = [p.copy() for p in s["points"]]
    pts[1] = np.zeros((0, 3)); pts[2] = pts[2][::20]
    om = oracle.VoxelMap(W, wl.voxel_size, wl.max_layer, wl.min_eigen_value, wl.plane_thre, wl.min_point, wl.max_points, 5)
    for i in range(W):
        ctx.cut_voxel(i, pts[i], poses[i]); om.cut_voxel(i, pts[i], poses[i])
    of = oracle.Factor(W)
    ctx.recut(W, poses, multi=False); om.recut(W, poses, of, multi=False)
    assert ctx.size() == of.size() > 50
    H, g, r = ctx.acc_evaluate2(poses); H2, g2, r2 = of.acc_evaluate2(poses)
    assert not H[6:12, 6:12].any() and not H2[6:12, 6:12].any()               # the empty frame's block
    assert np.abs(H - H2).max() < 1e-9 * np.abs(H2).max() and abs(r - r2) < 1e-11 * abs(r2)
    a = ctx.lidar_ba_damping_iter(poses, max_iter=3, thd_num=2)
    b = of.lidar_ba_damping_iter(poses, max_iter=3, thd_num=2)
    assert a["trace"].shape == b["trace"].shape
    assert np.abs(a["poses"] - b["poses"]).max() < 1e-6
    assert np.array_equal(a["poses"][1], poses[1]) and np.array_equal(b["poses"][1], poses[1])   # nothing constrains the empty frame: it stays put
    ctx.close()


def test_points_outside_the_key_range_are_dropped_not_misfiled():
    """The packed root key holds 21 bits per axis: a point 3e6 voxels away has no slot; it must be counted out, not land in
    another voxel (the reference's int64 keys have no such limit; no real map reaches it)."""
    capi, synth, wl, ctx = _ctx()
    W = wl.win_size
    pose = np.concatenate([np.eye(3).ravel(), np.zeros(3)])
    rng = np.random.default_rng(1)
    good = np.c_[rng.uniform(0.05, 0.45, (100, 2)), np.full(100, 0.2)]
    far = good + np.array([3e6 * wl.voxel_size, 0.0, 0.0])
    ctx.cut_voxel(0, np.concatenate([good, far]), pose)
    assert ctx.num_roots() == 1
    leaves = ctx.dump_leaves()
    assert len(leaves) == 1 and leaves[0, 5] == 100
    ctx.close()


def test_maximum_window_size_through_the_map_path(oracle):
    """W = 16 (VBA_MAX_WIN): insert, recut, extraction, lidar LM and marginalisation against the oracle."""
    import dataclasses
    import voxel_slam_amd  # noqa: F401
    from voxel_slam_amd import capi, synth
    wl = dataclasses.replace(synth.CONFIGS["room20k_w4"], win_size=16, n_pts=8000)
    s = synth.make_scans(wl)
    W = 16
    poses = synth.poses_flat(s["R0"], s["p0"])
    ctx = capi.Context(capi.options_from_workload(wl))
    om = oracle.VoxelMap(W, wl.voxel_size, wl.max_layer, wl.min_eigen_value, wl.plane_thre, wl.min_point, wl.max_points, 5)
    for i in range(W):
        ctx.cut_voxel(i, s["points"][i], poses[i], multi=True); om.cut_voxel(i, s["points"][i], poses[i], multi=True)
    of = oracle.Factor(W)
    ctx.recut(W, poses, multi=True); om.recut(W, poses, of, multi=True)
    assert ctx.size() == of.size() > 50
    a = ctx.lidar_ba_damping_iter(poses, max_iter=3, thd_num=2)
    b = of.lidar_ba_damping_iter(poses, max_iter=3, thd_num=2)
    assert a["trace"].shape == b["trace"].shape and np.abs(a["poses"] - b["poses"]).max() < 1e-6
    ctx.evaluate_only_residual(b["poses"]); of.evaluate_only_residual(b["poses"])
    ctx.margi(W, b["poses"], jour=1.0); om.margi(W, b["poses"], of, jour=1.0)
    g = {tuple(int(v) for v in r[:5]): r for r in ctx.dump_leaves()}
    o = {tuple(int(v) for v in r[:5]): r for r in om.dump_leaves()}
    assert set(g) == set(o)
    assert all(g[k][5] == o[k][5] and g[k][6] == o[k][6] and g[k][7] == o[k][7] and g[k][8] == o[k][8] for k in o)
    ctx.close()
