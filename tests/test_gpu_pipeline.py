"""Per-scan chain of thd_odometry_localmapping (voxelslam.cpp:1860-2043) through the C ABI, stage by stage against the
oracle on an evolving map: down_sampling_voxel -> var_init -> lio_state_estimation (once the planes exist) -> pvec_update +
cut_voxel_multi -> multi_recut -> (window full) LI_BA_Optimizer::damping_iter -> multi_margi -> ring rotation.
After every stage both sides continue from the DEVICE result, so each comparison sees the same inputs and rounding does not
accumulate into a different trajectory."""
import dataclasses

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_scan_pipeline_stage_parity(oracle):
    import voxel_slam_amd  # noqa: F401
    from voxel_slam_amd import capi, synth
    wl = dataclasses.replace(synth.CONFIGS["room20k_w4"], win_size=4)
    W, nscan = wl.win_size, 8
    big = dataclasses.replace(wl, win_size=nscan)
    s = synth.make_scans(big)
    imu_samples, vel, g = synth.make_imu(big, gyr_sigma=1e-3, acc_sigma=1e-2)
    nm = np.array([0.01] * 3 + [1.0] * 3); nw = np.array([1e-4] * 6)
    imu_all = [capi.imu_preintegrate(t, gy, ac, np.zeros(3), np.zeros(3), nm, nw) for (t, gy, ac) in imu_samples]
    ctx = capi.Context(capi.options_from_workload(wl))
    om = oracle.VoxelMap(W, wl.voxel_size, wl.max_layer, wl.min_eigen_value, wl.plane_thre, wl.min_point, wl.max_points, 5)
    of = oracle.Factor(W)
    ext = np.concatenate([np.eye(3).ravel(), np.zeros(3)])
    states, imus = [], []            # window buffers (x_buf, imu_pre_buf)
    win_count, n_ba, n_odom = 0, 0, 0
    rng = np.random.default_rng(3)
    for k in range(nscan):
        raw = s["points"][k].astype(np.float32).astype(np.float64)
        # down-sampling + per-point covariance model
        pd, cnt, first = ctx.down_sampling_voxel(raw, 0.05)
        po, ocnt, ofirst = oracle.down_sampling_voxel(raw, 0.05)
        np.testing.assert_array_equal(first, ofirst)
        np.testing.assert_allclose(pd, po, rtol=0, atol=5e-6)
        pts, var_b = ctx.var_init(pd, ext, 0.02, 0.05)
        opts, ovar = oracle.var_init(pd, ext, 0.02, 0.05)
        np.testing.assert_allclose(var_b, ovar, rtol=1e-9, atol=1e-15)
        # predicted state: ground truth + a small error (stands in for the IMU propagation)
        st = np.zeros(25)
        st[0] = 0.1 * k
        st[1:10] = (s["R_gt"][k] @ synth.so3_exp(rng.normal(0, np.radians(0.1), 3))).ravel()
        st[10:13] = s["p_gt"][k] + rng.normal(0, 0.01, 3); st[13:16] = vel[k]; st[22:25] = g
        cov = np.eye(15) * 1e-4
        cov[9:, 9:] = np.eye(6) * 1e-5
        if n_ba >= 1:                                  # planes have been refreshed by a margi: scan-to-map update
            ok_g, st_g, cov_g = ctx.lio_state_estimation(pts, var_b, st, cov)
            ok_o, st_o, cov_o, tr = om.lio_state_estimation(pts, var_b, st, cov)
            # (bar 1e-4 m / 1e-4 rad; a point that sits on a 3-sigma gate can be matched on one side only, since the
            #  plane parameters of the two maps agree to ~1e-9, which moves the update by ~1e-6 on this sparse scan)
            assert ok_g == ok_o and np.abs(st_g - st_o).max() < 2e-5
            assert np.abs(cov_g - cov_o).max() < 1e-3 * np.abs(cov_o).max()
            st, cov = st_g, cov_g
            n_odom += 1
        states.append(st.copy())
        if k > 0:
            imus.append(imu_all[k - 1].copy())
        win_count += 1
        pose = np.concatenate([st[1:10], st[10:13]])
        ctx.pvec_update_cut_voxel(win_count - 1, pts, var_b, pose, cov.ravel(), multi=True)
        wvar, pw = oracle.pvec_update(pts, var_b, st, cov.ravel())
        om.cut_voxel(win_count - 1, pts, pose, var=wvar, multi=True)
        x_buf = np.array([np.concatenate([q[1:10], q[10:13]]) for q in states])
        ctx.recut(win_count, x_buf, multi=True)
        om.recut(win_count, x_buf, of, multi=True)
        assert ctx.size() == of.size() and ctx.num_slide_roots() == om.num_slide_roots()
        if win_count >= W:
            S = np.array(states); I = np.array(imus[-(W - 1):])
            a = ctx.li_ba_damping_iter(S, I, gravity=False, max_iter=3)
            b = of.li_ba_damping_iter(S, I, gravity=False, imu_coef=wl.imu_coef, max_iter=3)
            assert a["trace"].shape == b["trace"].shape
            assert np.allclose(a["trace"], b["trace"], rtol=1e-6, atol=1e-10)
            assert np.abs(a["states"] - b["states"]).max() < 1e-6
            states = [q for q in a["states"]]
            imus = imus[:-(W - 1)] + [q for q in a["imus"]]
            n_ba += 1
            x_buf = np.array([np.concatenate([q[1:10], q[10:13]]) for q in states])
            ctx.margi(win_count, x_buf, jour=float(k))
            om.margi(win_count, x_buf, of, jour=float(k))
            assert ctx.num_slide_roots() == om.num_slide_roots() and ctx.num_roots() == om.num_roots()
            ctx.slide(1); om.slide(1)
            states = states[1:]
            win_count -= 1
    assert n_ba == nscan - W + 1 and n_odom >= 3
    ctx.close()
