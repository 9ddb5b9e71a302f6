"""GPU parity test for the odometry scan-to-map update: VOXEL_SLAM::lio_state_estimation
(voxelslam.cpp:962-1098) with match() / OctoTree::match (voxel_map.hpp:2167-2205, 1649-1721) on the device map vs the CPU
oracle, after the map's planes were refreshed by a few local-mapping steps (plane_update runs inside margi)."""
import dataclasses

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rand_var(n, seed, scale=0.01):
    rng = np.random.default_rng(seed)
    A = rng.normal(0, scale, (n, 3, 3))
    return np.ascontiguousarray((A @ A.transpose(0, 2, 1) + 1e-6 * np.eye(3)).reshape(n, 9))


def test_lio_state_estimation_parity(oracle):
    import voxel_slam_amd  # noqa: F401
    from voxel_slam_amd import capi, synth
    wl = dataclasses.replace(synth.CONFIGS["room20k_w4"], win_size=4)
    W = wl.win_size
    nscan = 8
    s = synth.make_scans(dataclasses.replace(wl, win_size=nscan))
    ctx = capi.Context(capi.options_from_workload(wl))
    om = oracle.VoxelMap(W, wl.voxel_size, wl.max_layer, wl.min_eigen_value, wl.plane_thre, wl.min_point, wl.max_points, 5)
    of = oracle.Factor(W)
    x_g, x_o = [], []
    win_count = 0
    for k in range(nscan - 1):                       # local mapping on GT poses: builds + refreshes the planes
        pose = synth.poses_flat(s["R_gt"][k:k + 1], s["p_gt"][k:k + 1])[0]
        var = _rand_var(len(s["points"][k]), 100 + k)
        x_g.append(pose.copy()); x_o.append(pose.copy())
        win_count += 1
        ctx.cut_voxel(win_count - 1, s["points"][k], x_g[-1], var=var, multi=True)
        om.cut_voxel(win_count - 1, s["points"][k], x_o[-1], var=var, multi=True)
        ctx.recut(win_count, np.array(x_g), multi=True)
        om.recut(win_count, np.array(x_o), of, multi=True)
        if win_count >= W:
            ctx.margi(win_count, np.array(x_g), jour=float(k))
            om.margi(win_count, np.array(x_o), of)
            ctx.slide(1); om.slide(1)
            x_g = x_g[1:]; x_o = x_o[1:]
            win_count -= 1
    # odometry of the next scan from a perturbed prediction
    k = nscan - 1
    rng = np.random.default_rng(5)
    state = np.zeros(25)
    state[1:10] = (s["R_gt"][k] @ synth.so3_exp(rng.normal(0, np.radians(0.2), 3))).ravel()
    state[10:13] = s["p_gt"][k] + rng.normal(0, 0.02, 3)
    state[13:16] = [1.0, 0.5, 0.0]; state[22:25] = [0, 0, -9.8]
    cov = np.eye(15) * 1e-4
    cov[9:, 9:] = np.eye(6) * 1e-5                    # IMUST::setZero (tools.hpp:188-197)
    pts = s["points"][k]
    var_b = _rand_var(len(pts), 999, scale=0.005)
    ok_g, st_g, cov_g = ctx.lio_state_estimation(pts, var_b, state, cov)
    ok_o, st_o, cov_o, tr = om.lio_state_estimation(pts, var_b, state, cov)
    assert tr[0, 0] > 2000, "too few matches for a meaningful test: %s" % tr
    assert ok_g == ok_o
    assert np.abs(st_g - st_o).max() < 1e-7, np.abs(st_g - st_o).max()
    assert np.abs(cov_g - cov_o).max() < 1e-9 * np.abs(cov_o).max()
    # the update pulled the prediction towards the ground truth
    assert np.abs(st_g[10:13] - s["p_gt"][k]).max() < np.abs(state[10:13] - s["p_gt"][k]).max()


def test_lio_state_estimation_kdtree_parity(oracle):
    """Initialisation odometry (voxelslam.cpp:1102-1252): exact 5-NN plane fit against the point-cloud map + iterated EKF, then
    the map update (append + 0.5 m re-sampling), over a short sequence: the first scan only seeds the map."""
    import voxel_slam_amd  # noqa: F401
    from voxel_slam_amd import capi, synth
    wl = dataclasses.replace(synth.CONFIGS["room20k_w4"], win_size=5)
    s = synth.make_scans(wl)
    ctx = capi.Context(capi.options_from_workload(wl))
    ko = oracle.KdOdom()
    rng = np.random.default_rng(17)
    cov_g = np.eye(15) * 1e-4
    cov_g[9:, 9:] = np.eye(6) * 1e-5
    cov_o = cov_g.copy()
    for k in range(wl.win_size):
        pts = s["points"][k].astype(np.float32).astype(np.float64)          # PCL scan points are float
        state = np.zeros(25)
        dR = np.eye(3) if k == 0 else synth.so3_exp(rng.normal(0, np.radians(0.3), 3))
        state[1:10] = (s["R_gt"][k] @ dR).ravel()
        state[10:13] = s["p_gt"][k] + (0 if k == 0 else rng.normal(0, 0.03, 3))
        state[13:16] = [0.3, 0.1, 0.0]; state[22:25] = [0, 0, -9.8]
        it_g, st_g, cov_g2 = ctx.lio_state_estimation_kdtree(pts, state, cov_g)
        it_o, st_o, cov_o2 = ko.lio_state_estimation(pts, state, cov_o)
        assert it_g == it_o, (k, it_g, it_o)
        if k == 0:
            assert it_g == 0 and ctx.kdtree_size() == len(pts) == len(ko.tree())
            np.testing.assert_array_equal(st_g, state)
            continue
        # the map points differ in the last float digit (the reference's running mean is evaluated in float, the device rounds
        # the f64 centroid once), which moves the fitted planes by ~1e-7
        assert np.abs(st_g - st_o).max() < 1e-5, (k, np.abs(st_g - st_o).max())
        assert np.abs(cov_g2 - cov_o2).max() < 1e-4 * np.abs(cov_o2).max()
        assert np.linalg.norm(st_g[10:13] - s["p_gt"][k]) < np.linalg.norm(state[10:13] - s["p_gt"][k])
        tg, to = ctx.kdtree_points(), ko.tree()
        assert abs(len(tg) - len(to)) <= max(2, len(to) // 500), (len(tg), len(to))   # a point within 1e-5 of a 0.5 m face may change cell
        if len(tg) == len(to):
            assert np.abs(tg - to).max() < 1e-4
        cov_g, cov_o = cov_g2, cov_o2
    ctx.close()


def test_kdtree_odometry_edge_cases():
    """Empty scan, a map below the 100-point threshold (the scan only seeds it, VS:1105-1118) and pl_tree->clear()."""
    import voxel_slam_amd  # noqa: F401
    from voxel_slam_amd import capi, synth
    ctx = capi.Context(capi.options_from_workload(synth.CONFIGS["room20k_w4"]))
    state = np.zeros(25); state[1:10] = np.eye(3).ravel(); state[22:25] = [0, 0, -9.8]
    cov = np.eye(15) * 1e-4
    rng = np.random.default_rng(0)
    it, st, cv = ctx.lio_state_estimation_kdtree(np.zeros((0, 3)), state, cov)
    assert it == 0 and ctx.kdtree_size() == 0
    pts = rng.uniform(-5, 5, (60, 3)).astype(np.float32).astype(np.float64)
    it, st, cv = ctx.lio_state_estimation_kdtree(pts, state, cov)          # 0 < 100: seeds
    assert it == 0 and ctx.kdtree_size() == 60
    np.testing.assert_array_equal(ctx.kdtree_points(), pts)                # identity pose: the float points themselves
    it, st, cv = ctx.lio_state_estimation_kdtree(pts, state, cov)          # 60 < 100: appended, still no estimation
    assert it == 0 and ctx.kdtree_size() == 120
    np.testing.assert_array_equal(st, state); np.testing.assert_array_equal(cv, cov)
    ctx._chk(ctx.lib.vba_odom_kdtree_reset(ctx.h))
    assert ctx.kdtree_size() == 0
    ctx.close()
