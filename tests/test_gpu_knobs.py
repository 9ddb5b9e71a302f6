"""GPU parity on the parameter sets the reference SHIPS (its config/*.yaml — the only fixtures /root/reference holds for this
path): every LocalBA knob set and every GBA knob set, run through the C ABI against the CPU oracle on the same seeded scans.

    set        file:line                      voxel  min_eig  max_layer  plane thre (inverted)  imu_coef  dept/beam
    avia       config/avia.yaml:27-47         1      0.0025   2          1/4                    1e-4      0.02/0.05
    avia_fly   config/avia_fly.yaml:26-44     4      0.01     2          1/4                    1e-4      0.01/0.01
    hesai      config/hesai.yaml:32-55        1      0.0025   2          1/1                    2.5e-5    0.01/0.01
    mid360     config/mid360.yaml:24-42       1      0.0025   2          1/4                    2e-4      0.02/0.05
    ouster     config/ouster.yaml:22-40       2      0.01     1          1/1 (*)                2e-4      0.01/0.01
    velodyne   config/velodyne.yaml:22-39     2      0.01     2          1/4                    1e-4      0.01/0.01
    motioninit voxelslam.cpp:624-630          1      0.02     2          1/4                    1e-4      0.02/0.05
(*) ouster.yaml spells the key `eigen_value_array`, which LocalBA never reads: `plane_eigen_value_thre` keeps its default
    {1, 1, 1, 1} (voxelslam.cpp:926) and is inverted to 1.0 (voxelslam.cpp:930-931).
`max_layer 1` changes which leaves keep raw points (voxel_map.hpp:1131-1132) and where fix_divide / subdivide stop.

Chain per set: var_init -> pvec_update + cut_voxel_multi x W -> multi_recut (+ tras_opt) -> acc_evaluate2 -> Lidar_BA_Optimizer ->
LI_BA_Optimizer and LI_BA_OptimizerGravity -> multi_margi -> refined planes (centre, normal, radius, plane_var, cov_add).
Bars: structure exact; pcr_add / cov_add bit-identical; H / g / r / planes at the bars of tests/test_gpu_fullsize.py.
"""
import dataclasses

import numpy as np
import pytest

from test_gpu_fullsize import _assert_structure_equal, _pose_err, _sorted

pytestmark = pytest.mark.gpu

Q = (0.25,) * 4
ONE = (1.0,) * 4
LOCAL_SETS = {
    #            voxel  min_eig  max_layer thre  imu_coef dept  beam
    "avia":       (1.0, 0.0025, 2, Q,   1e-4,   0.02, 0.05),
    "avia_fly":   (4.0, 0.01,   2, Q,   1e-4,   0.01, 0.01),
    "hesai":      (1.0, 0.0025, 2, ONE, 2.5e-5, 0.01, 0.01),
    "mid360":     (1.0, 0.0025, 2, Q,   2e-4,   0.02, 0.05),
    "ouster":     (2.0, 0.01,   1, ONE, 2e-4,   0.01, 0.01),
    "velodyne":   (2.0, 0.01,   2, Q,   1e-4,   0.01, 0.01),
    "motioninit": (1.0, 0.02,   2, Q,   1e-4,   0.02, 0.05),
}
# GBA/voxel_size, GBA/min_eigen_value, GBA/eigen_value_array (inverted, voxelslam.cpp:3020-3024)
GBA_SETS = {
    "avia":     (2.0, 0.1, 1 / 4.0),      # config/avia.yaml:62-64
    "avia_fly": (15.0, 10.0, 1 / 2.0),    # config/avia_fly.yaml:57-59
    "hesai":    (1.0, 0.01, 1 / 2.0),     # config/hesai.yaml:76-78
    "mid360":   (2.0, 0.01, 1 / 4.0),     # config/mid360.yaml:50-52, config/ouster.yaml:53-55
    "velodyne": (2.0, 0.01, 1 / 9.0),     # config/velodyne.yaml:52-54
}


def _workload(synth, name, n_pts=40000):
    vs, mev, ml, thre, coef, dept, beam = LOCAL_SETS[name]
    return dataclasses.replace(synth.CONFIGS["hesai200k_w10"], name="knob_" + name, n_pts=n_pts, voxel_size=vs, min_eigen_value=mev,
                               max_layer=ml, plane_thre=thre, imu_coef=coef, dept_err=dept, beam_err=beam)


def _li_inputs(capi, synth, wl, s):
    W = wl.win_size
    imu_samples, vel, grav = synth.make_imu(wl, gyr_sigma=1e-3, acc_sigma=1e-2)
    nm = np.array([0.01] * 3 + [1.0] * 3); nw = np.array([1e-4] * 6)
    imus = np.stack([capi.imu_preintegrate(t, gy, ac, np.zeros(3), np.zeros(3), nm, nw) for (t, gy, ac) in imu_samples])
    states = np.zeros((W, 25))
    for i in range(W):
        states[i, 0] = 0.1 * i
        states[i, 1:10] = s["R0"][i].ravel(); states[i, 10:13] = s["p0"][i]; states[i, 13:16] = vel[i]; states[i, 22:25] = grav
    return states, imus


@pytest.mark.parametrize("name", list(LOCAL_SETS))
def test_local_mapping_chain_on_shipped_knob_set(oracle, name):
    import voxel_slam_amd  # noqa: F401
    from voxel_slam_amd import capi, synth
    wl = _workload(synth, name)
    s = synth.make_scans(wl)
    W = wl.win_size
    ext = np.concatenate([np.eye(3).ravel(), np.zeros(3)])
    rng = np.random.default_rng(11)
    A = rng.normal(0, 0.003, (15, 15)); cov = A @ A.T + np.eye(15) * 1e-6
    ctx = capi.Context(capi.options_from_workload(wl))
    om = oracle.VoxelMap(W, wl.voxel_size, wl.max_layer, wl.min_eigen_value, wl.plane_thre, wl.min_point, wl.max_points, 5)
    poses = synth.poses_flat(s["R0"], s["p0"])
    for i in range(W):
        state = np.zeros(25); state[1:10] = poses[i, :9]; state[10:13] = poses[i, 9:]
        p_i, v_i = oracle.var_init(s["points"][i], ext, wl.dept_err, wl.beam_err)
        gp, gv = ctx.var_init(s["points"][i], ext, wl.dept_err, wl.beam_err)
        assert np.array_equal(gp, p_i) and np.allclose(gv, v_i, rtol=1e-12, atol=1e-18)
        v_w, _ = oracle.pvec_update(p_i, v_i, state, cov)
        om.cut_voxel(i, p_i, poses[i], var=v_w, multi=True)
        ctx.pvec_update_cut_voxel(i, p_i, v_i, poses[i], cov, multi=True)
    assert ctx.num_roots() == om.num_roots() and ctx.num_slide_roots() == om.num_slide_roots()
    of = oracle.Factor(W)
    ctx.recut(W, poses, multi=True); om.recut(W, poses, of, multi=True)
    V = ctx.size()
    assert V == of.size() and V > 50, (V, of.size())
    g, _, _ = _sorted(ctx.dump_leaves()); gpv, _, _ = _sorted(ctx.dump_plane_var())
    od = om.dump_leaves()
    o, oca, _ = _sorted(od, om.dump_cov_add())
    nplane = _assert_structure_equal(g, o)
    assert nplane >= V and int(o[:, 3].max()) <= wl.max_layer
    assert np.array_equal(g[:, 9] >= 0, o[:, 9] >= 0), "tras_opt selects different leaves"
    assert np.array_equal(gpv[:, 41:], oca), "cov_add after insert / recut is not bit-identical"

    # a9 / a10 at the start poses
    H, gr, r = ctx.acc_evaluate2(poses)
    H2, gr2, r2 = of.acc_evaluate2(poses)
    assert abs(r - r2) < 1e-10 * abs(r2)
    assert np.abs(H - H2).max() < 1e-9 * np.abs(H2).max() and np.abs(gr - gr2).max() < 1e-9 * np.abs(gr2).max()
    assert abs(ctx.evaluate_only_residual(poses) - of.evaluate_only_residual(poses)) < 1e-10 * abs(r2)

    # a13: lidar-only LM
    a = ctx.lidar_ba_damping_iter(poses, max_iter=3, thd_num=2)
    b = of.lidar_ba_damping_iter(poses, max_iter=3, thd_num=2, parallel=True)
    assert a["trace"].shape == b["trace"].shape and a["converge"] == b["converge"]
    for ra, rb in zip(a["trace"], b["trace"]):
        assert np.allclose(ra, rb, rtol=1e-7 if rb[1] < rb[0] else 1e-4, atol=1e-12), (ra, rb)
    ang, tr = _pose_err(a["poses"], b["poses"], W)
    assert ang < 1e-6 and tr < 1e-6, (ang, tr)

    # a11 / a12: LI-BA with the set's imu_coef
    states, imus = _li_inputs(capi, synth, wl, s)
    last = None
    for gravity in (False, True):
        ctx.evaluate_only_residual(poses); of.evaluate_only_residual(poses)
        a = ctx.li_ba_damping_iter(states, imus, gravity=gravity, max_iter=3)
        b = of.li_ba_damping_iter(states, imus, gravity=gravity, imu_coef=wl.imu_coef, max_iter=3, parallel=True)
        assert a["trace"].shape == b["trace"].shape
        assert np.allclose(a["trace"], b["trace"], rtol=1e-6, atol=1e-10), (gravity, a["trace"], b["trace"])
        assert np.abs(a["states"] - b["states"]).max() < 1e-6
        assert np.abs(a["hess"] - b["hess"]).max() < 1e-7 * np.abs(b["hess"]).max()
        last = b
    refined = np.concatenate([last["states"][:, 1:10], last["states"][:, 10:13]], 1)

    # a14: marginalisation with the refined window at identical poses on both sides (see tests/test_gpu_fullsize.py)
    ctx.evaluate_only_residual(refined); of.evaluate_only_residual(refined)
    ctx.margi(W, refined, jour=3.0); om.margi(W, refined, of, jour=3.0)
    assert ctx.num_slide_roots() == om.num_slide_roots() and ctx.num_roots() == om.num_roots()
    g, _, _ = _sorted(ctx.dump_leaves()); gpv, _, _ = _sorted(ctx.dump_plane_var())
    od = om.dump_leaves()
    o, opv, order = _sorted(od, om.dump_plane_var())
    oca = om.dump_cov_add()[order]
    _assert_structure_equal(g, o)
    upd = (o[:, 7] != 0) & (np.abs(o[:, 35:38]).max(1) > 0)
    assert upd.sum() > 50
    assert np.abs(g[upd, 32:35] - o[upd, 32:35]).max() < 1e-9, "plane.center"
    assert np.abs(np.abs((g[upd, 35:38] * o[upd, 35:38]).sum(1)) - 1).max() < 1e-9, "plane.normal"
    assert (np.abs(g[upd, 38] - o[upd, 38]) <= 1e-6 * np.maximum(1e-3, np.abs(o[upd, 38]))).all(), "plane.radius"
    sgn = np.sign((g[upd, 35:38] * o[upd, 35:38]).sum(1))
    G = gpv[upd, 5:41].reshape(-1, 6, 6).copy(); O = opv[upd].reshape(-1, 6, 6)
    G[:, :3, 3:] *= sgn[:, None, None]; G[:, 3:, :3] *= sgn[:, None, None]
    sc = np.abs(O).reshape(len(O), -1).max(1)
    err = np.abs(G - O).reshape(len(O), -1).max(1)
    assert (err <= 1e-5 * sc + 1e-18).all(), ("plane_var", float((err / np.maximum(sc, 1e-300)).max()))
    assert np.quantile(err / np.maximum(sc, 1e-300), 0.99) < 1e-8
    assert np.array_equal(gpv[:, 41:], oca), "cov_add after margi is not bit-identical"
    # the next scan of the session lands on the marginalised map: ring rotation + one more insert / recut (VS:2014-2019, 1916-1927)
    ctx.slide(1); om.slide(1)
    x2 = np.concatenate([refined[1:], refined[-1:]])
    p_i, v_i = oracle.var_init(s["points"][0], ext, wl.dept_err, wl.beam_err)
    state = np.zeros(25); state[1:10] = x2[W - 1, :9]; state[10:13] = x2[W - 1, 9:]
    v_w, _ = oracle.pvec_update(p_i, v_i, state, cov)
    om.cut_voxel(W - 1, p_i, x2[W - 1], var=v_w, multi=True)
    ctx.pvec_update_cut_voxel(W - 1, p_i, v_i, x2[W - 1], cov, multi=True)
    of2 = oracle.Factor(W)
    ctx.recut(W, x2, multi=True); om.recut(W, x2, of2, multi=True)
    assert ctx.size() == of2.size()
    g, _, _ = _sorted(ctx.dump_leaves()); o, _, _ = _sorted(om.dump_leaves())
    _assert_structure_equal(g, o)
    ctx.close()


@pytest.mark.parametrize("name", list(GBA_SETS))
def test_gba_on_shipped_knob_set(oracle, name):
    """OctreeGBA build + HBA_add_edge (loop_refine.hpp:273-537, voxelslam.cpp:2822-3015) with every shipped GBA section."""
    import voxel_slam_amd  # noqa: F401
    from voxel_slam_amd import capi, synth
    gvs, gmev, garr = GBA_SETS[name]
    geig = [garr] * 4
    local = "avia" if name not in LOCAL_SETS else name
    wl = _workload(synth, local)
    s = synth.make_scans(wl)
    clouds = [p.astype(np.float32).astype(np.float64) for p in s["points"]]
    poses = synth.poses_flat(s["R0"], s["p0"])
    ctx = capi.Context(capi.options_from_workload(wl))
    o = ctx.opt
    cfg = oracle.gba_cfg13(gvs, gmev, geig, o.voxel_size, o.min_eigen_value, list(o.plane_eigen_value_thre), o.max_layer)
    n_dev = ctx.gba_build(clouds, poses, gvs, gmev, geig)
    f = oracle.gba_build(clouds, poses, cfg)
    assert n_dev == f.size() and n_dev > 5, (n_dev, f.size())
    ev, evec, pa = ctx.read_back(); oev, oevec, opa = f.read_back()
    ka = np.lexsort((pa[:, 6], pa[:, 9])); kb = np.lexsort((opa[:, 6], opa[:, 9]))
    np.testing.assert_array_equal(pa[ka, 9], opa[kb, 9])
    np.testing.assert_allclose(pa[ka], opa[kb], rtol=0, atol=1e-12 * np.abs(opa[kb, :9]).max())
    H, g, r = ctx.acc_evaluate2(poses); oH, og, orr = f.acc_evaluate2(poses)
    np.testing.assert_allclose(r, orr, rtol=1e-10)
    np.testing.assert_allclose(H, oH, rtol=0, atol=1e-9 * np.abs(oH).max())
    got = ctx.hba_add_edge(clouds, poses, gvs, gmev, geig, 2, 2)
    want = oracle.hba_add_edge(clouds, poses, cfg, 2, 2)
    assert want["status"] == 0 and len(got["resis"]) == len(want["resis"])
    np.testing.assert_allclose(got["resis"], want["resis"], rtol=1e-6)
    np.testing.assert_allclose(got["poses"], want["poses"], rtol=0, atol=1e-6)
    ge, we = got["edges"], want["edges"]
    assert len(ge) == len(we) and len(ge) > 0
    np.testing.assert_array_equal(ge[:, :2], we[:, :2])
    np.testing.assert_allclose(ge[:, 2:14], we[:, 2:14], rtol=0, atol=1e-6)
    np.testing.assert_allclose(ge[:, 14:], we[:, 14:], rtol=1e-3)
    ctx.close()
