"""GPU parity tests (MI355X) at the FULL size of the headline configuration (BASELINE.json configs[2]: Hesai-32 pattern,
200k points per scan, W = 10, 0.3 m voxels): the whole chain K1 x 10 -> K2 -> K3/K4/LM -> K5 through the C ABI against the
CPU oracle on the same seeded scans.

Bars: integer / structural state (leaf set, layers, octant paths, point counts, plane flags, isexist, factor count) EXACT;
the f64 cluster sums pcr_add and cov_add BIT-IDENTICAL to the oracle's (the insert, the recut re-push, the fixed-point paths and
PointCluster::transform run the reference's operation order: csrc/vba_kernels_map.hpp "order-preserving accumulation"), and
bit-identical from run to run; H / g / residual 1e-9 (the 3x3 eigen-solvers differ: direct on the device, Jacobi in the oracle);
poses after 3 LM iterations <= 1e-6 (north-star bar: 1e-4 m / 1e-4 rad); plane centre / normal 1e-9, plane_var 1e-8 (q99).
The comparisons are vectorised (sorted leaf tables): ~1e5 leaves.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NAME = "hesai200k_w10"


@pytest.fixture(scope="module")
def capi():
    import voxel_slam_amd  # noqa: F401
    from voxel_slam_amd import capi as m
    return m


@pytest.fixture(scope="module")
def synth():
    from voxel_slam_amd import synth as s
    return s


@pytest.fixture(scope="module")
def scans(synth):
    wl = synth.CONFIGS[NAME]
    return wl, synth.make_scans(wl)


def _sorted(d, extra=None):
    """Sort a leaf dump by (kx, ky, kz, layer, path); returns (sorted dump, sorted extra, order)."""
    order = np.lexsort((d[:, 4], d[:, 3], d[:, 2], d[:, 1], d[:, 0]))
    return d[order], (extra[order] if extra is not None else None), order


def _assert_structure_equal(g, o, check_plane=True, sum_tol=0.0, eig_tol=1e-12):
    assert g.shape == o.shape, "leaf counts differ: %d vs %d" % (len(g), len(o))
    assert np.array_equal(g[:, :5], o[:, :5]), "leaf key sets differ"
    assert np.array_equal(g[:, 5:7], o[:, 5:7]), "N_add / N_fix differ on %d leaves" % int((g[:, 5:7] != o[:, 5:7]).any(1).sum())
    assert np.array_equal(g[:, 8], o[:, 8]), "isexist differs"
    if sum_tol == 0.0:
        bad = (g[:, 22:32] != o[:, 22:32]).any(1)
        assert not bad.any(), "pcr_add is not bit-identical on %d of %d leaves (max rel %.3g)" % (
            int(bad.sum()), len(o), float(np.abs(g[:, 22:32] - o[:, 22:32]).max() / max(1.0, np.abs(o[:, 22:32]).max())))
    else:
        scale = np.maximum(1.0, np.abs(o[:, 22:31]).max(1))
        assert (np.abs(g[:, 22:32] - o[:, 22:32]).max(1) <= sum_tol * scale).all(), "pcr_add"
    if check_plane:
        bad = g[:, 7] != o[:, 7]
        assert not bad.any(), "is_plane differs on %d leaves, eig %s vs %s" % (int(bad.sum()), g[bad][:3, 10:13], o[bad][:3, 10:13])
        pl = o[:, 7] != 0
        m2 = np.maximum(np.abs(o[pl, 22:28]).max(1) / o[pl, 31], 1.0)
        assert (np.abs(g[pl, 10:13] - o[pl, 10:13]).max(1) < eig_tol * m2).all(), "plane eigenvalues"
    return int((o[:, 7] != 0).sum())


def _opts(capi, wl, **kw):
    o = capi.options_from_workload(wl)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _omap(oracle, wl):
    return oracle.VoxelMap(wl.win_size, wl.voxel_size, wl.max_layer, wl.min_eigen_value, wl.plane_thre, wl.min_point, wl.max_points, 5)


def _pose_err(a, b, W):
    dR = np.einsum("wij,wkj->wik", a[:, :9].reshape(W, 3, 3), b[:, :9].reshape(W, 3, 3))
    ang = np.linalg.norm(dR - np.eye(3)[None], axis=(1, 2)) / np.sqrt(2.0)
    return ang.max(), np.abs(a[:, 9:] - b[:, 9:]).max()


def test_fullsize_rebuild_hessian_lm(capi, oracle, synth, scans):
    """motion_init flow (voxelslam.cpp:664-713) at 2 M points: cut_voxel x 10, recut + tras_opt, acc_evaluate2,
    Lidar_BA_Optimizer::damping_iter and LI_BA_Optimizer(Gravity)::damping_iter."""
    wl, s = scans
    W = wl.win_size
    poses = synth.poses_flat(s["R0"], s["p0"])
    ctx = capi.Context(_opts(capi, wl))
    om = _omap(oracle, wl)
    for i in range(W):
        assert len(s["points"][i]) > 150000
        ctx.cut_voxel(i, s["points"][i], poses[i])
        om.cut_voxel(i, s["points"][i], poses[i])
    assert ctx.num_roots() == om.num_roots() and ctx.num_slide_roots() == om.num_slide_roots()
    g, _, _ = _sorted(ctx.dump_leaves()); o, _, _ = _sorted(om.dump_leaves())
    _assert_structure_equal(g, o, check_plane=False)

    of = oracle.Factor(W)
    om.recut(W, poses, of, multi=False)
    ctx.recut(W, poses, multi=False)
    g, _, _ = _sorted(ctx.dump_leaves()); o, _, _ = _sorted(om.dump_leaves())
    nplane = _assert_structure_equal(g, o)
    V = ctx.size()
    assert V == of.size() and V > 10000 and nplane >= V
    # the opt_state of both sides selects the same leaves (factor order differs: compare as sets)
    assert np.array_equal(g[:, 9] >= 0, o[:, 9] >= 0)

    H, gr, r = ctx.acc_evaluate2(poses)
    H2, gr2, r2 = of.acc_evaluate2(poses)
    assert abs(r - r2) < 1e-11 * abs(r2)
    assert np.abs(H - H2).max() < 1e-9 * np.abs(H2).max() and np.abs(gr - gr2).max() < 1e-9 * np.abs(gr2).max()
    assert np.array_equal(H, H.T)

    a = ctx.lidar_ba_damping_iter(poses, max_iter=3, thd_num=2)
    b = of.lidar_ba_damping_iter(poses, max_iter=3, thd_num=2, parallel=True)
    assert a["trace"].shape == b["trace"].shape and a["converge"] == b["converge"]
    for ra, rb in zip(a["trace"], b["trace"]):
        assert np.allclose(ra, rb, rtol=1e-7 if rb[1] < rb[0] else 1e-4, atol=1e-12)
    ang, tr = _pose_err(a["poses"], b["poses"], W)
    assert ang < 1e-6 and tr < 1e-6, (ang, tr)
    gt = synth.poses_flat(s["R_gt"], s["p_gt"])
    assert np.abs(a["poses"][:, 9:] - gt[:, 9:]).max() < np.abs(poses[:, 9:] - gt[:, 9:]).max()
    # refined plane parameters left by the last residual pass (what margi copies, voxel_map.hpp:1495-1501); factor ORDER differs
    # between the two stores, so they are matched through pcr_add's point count + centroid
    ev, evec, pa = ctx.read_back(); ev2, evec2, pa2 = of.read_back()
    ka = np.lexsort((pa[:, 8], pa[:, 7], pa[:, 6], pa[:, 9])); kb = np.lexsort((pa2[:, 8], pa2[:, 7], pa2[:, 6], pa2[:, 9]))
    # (they were evaluated at the LAST trial poses, which agree to the pose tolerance above — a rejected last step comes out of
    #  an ill-conditioned solve that amplifies the 1e-10 summation-order differences of H — so the sums agree to |dpose| * |p| * N)
    dpa = np.abs(pa[ka] - pa2[kb]).max() / np.abs(pa2).max()
    assert np.array_equal(pa[ka][:, 9], pa2[kb][:, 9]) and dpa < 1e-6, dpa
    assert np.abs(ev[ka] - ev2[kb]).max() < 1e-7, np.abs(ev[ka] - ev2[kb]).max()
    n1 = evec[ka].reshape(V, 3, 3)[:, :, 0]; n2 = evec2[kb].reshape(V, 3, 3)[:, :, 0]
    assert np.abs(np.abs((n1 * n2).sum(1)) - 1).max() < 1e-7

    # LiDAR-inertial BA (the optimiser the node runs per scan, voxelslam.cpp:1969) on the same full-size store
    imu_samples, vel, grav = synth.make_imu(wl, gyr_sigma=1e-3, acc_sigma=1e-2)
    nm = np.array([0.01] * 3 + [1.0] * 3); nw = np.array([1e-4] * 6)
    imus = np.stack([capi.imu_preintegrate(t, gy, ac, np.zeros(3), np.zeros(3), nm, nw) for (t, gy, ac) in imu_samples])
    states = np.zeros((W, 25))
    for i in range(W):
        states[i, 0] = 0.1 * i
        states[i, 1:10] = s["R0"][i].ravel(); states[i, 10:13] = s["p0"][i]; states[i, 13:16] = vel[i]; states[i, 22:25] = grav
    for gravity in (False, True):
        ctx.evaluate_only_residual(poses); of.evaluate_only_residual(poses)      # eigen state at the start poses on both sides
        a = ctx.li_ba_damping_iter(states, imus, gravity=gravity, max_iter=3)
        b = of.li_ba_damping_iter(states, imus, gravity=gravity, imu_coef=wl.imu_coef, max_iter=3, parallel=True)
        assert a["trace"].shape == b["trace"].shape
        assert np.allclose(a["trace"], b["trace"], rtol=1e-6, atol=1e-10)
        pa_ = np.concatenate([a["states"][:, 1:10], a["states"][:, 10:13]], 1); pb_ = np.concatenate([b["states"][:, 1:10], b["states"][:, 10:13]], 1)
        ang, tr = _pose_err(pa_, pb_, W)
        assert ang < 1e-6 and tr < 1e-6, (gravity, ang, tr)
        assert np.abs(a["states"] - b["states"]).max() < 1e-6
        assert np.abs(a["hess"] - b["hess"]).max() < 1e-7 * np.abs(b["hess"]).max()


def test_fullsize_local_mapping_step_planes(capi, oracle, synth, scans):
    """Steady-state step (voxelslam.cpp:1916-2043) at full size with per-point covariances: pvec_update + cut_voxel_multi x 10,
    multi_recut, damping_iter, multi_margi -> refined planes incl. the 6x6 plane_var and the 9x9 cov_add (rows a6 / a14)."""
    wl, s = scans
    W = wl.win_size
    ext = np.concatenate([np.eye(3).ravel(), np.zeros(3)])
    rng = np.random.default_rng(5)
    A = rng.normal(0, 0.003, (15, 15)); cov = A @ A.T + np.eye(15) * 1e-6
    ctx = capi.Context(_opts(capi, wl))
    om = _omap(oracle, wl)
    poses = synth.poses_flat(s["R0"], s["p0"])
    for i in range(W):
        state = np.zeros(25); state[1:10] = poses[i, :9]; state[10:13] = poses[i, 9:]
        p_i, v_i = oracle.var_init(s["points"][i], ext, wl.dept_err, wl.beam_err)
        v_w, _ = oracle.pvec_update(p_i, v_i, state, cov)
        om.cut_voxel(i, p_i, poses[i], var=v_w, multi=True)
        ctx.pvec_update_cut_voxel(i, p_i, v_i, poses[i], cov, multi=True)
    of = oracle.Factor(W)
    ctx.recut(W, poses, multi=True); om.recut(W, poses, of, multi=True)
    assert ctx.size() == of.size() > 10000
    # cov_add right after insertion + subdivision (Bf_var sums, VM:106-121 / 1138-1140)
    gd = ctx.dump_leaves(); gp = ctx.dump_plane_var()
    g, _, _ = _sorted(gd); gpv, _, _ = _sorted(gp)
    od = om.dump_leaves(); oca = om.dump_cov_add()
    o, oca_s, _ = _sorted(od, oca)
    _assert_structure_equal(g, o)
    assert np.array_equal(gpv[:, :5], o[:, :5])
    assert np.array_equal(gpv[:, 41:], oca_s), "cov_add after insert/recut is not bit-identical"
    assert (np.abs(oca_s).max(1) > 0).sum() > 10000

    a = ctx.lidar_ba_damping_iter(poses, max_iter=3, thd_num=2)
    b = of.lidar_ba_damping_iter(poses, max_iter=3, thd_num=2, parallel=True)
    ang, tr = _pose_err(a["poses"], b["poses"], W)
    assert ang < 1e-6 and tr < 1e-6
    # margi copies the sums and eigen-pairs of the LAST residual pass into the map (VM:1495-1501).  The two optimisers' last trial
    # poses agree to the bar above, not to rounding, so to compare the marginalisation itself at rounding level both sides get
    # one more residual pass at IDENTICAL poses (the oracle's refined ones) and marginalise with those
    ctx.evaluate_only_residual(b["poses"]); of.evaluate_only_residual(b["poses"])
    ctx.margi(W, b["poses"], jour=3.0); om.margi(W, b["poses"], of, jour=3.0)
    assert ctx.num_slide_roots() == om.num_slide_roots()
    g, _, _ = _sorted(ctx.dump_leaves()); gpv, _, _ = _sorted(ctx.dump_plane_var())
    od = om.dump_leaves()
    o, opv, order = _sorted(od, om.dump_plane_var())
    oca_s = om.dump_cov_add()[order]
    _assert_structure_equal(g, o)
    upd = (o[:, 7] != 0) & (np.abs(o[:, 35:38]).max(1) > 0)          # planes that plane_update has written
    assert upd.sum() > 10000
    assert np.abs(g[upd, 32:35] - o[upd, 32:35]).max() < 1e-9, "plane.center"
    assert np.abs(np.abs((g[upd, 35:38] * o[upd, 35:38]).sum(1)) - 1).max() < 1e-9, "plane.normal"
    assert (np.abs(g[upd, 38] - o[upd, 38]) <= 1e-6 * np.maximum(1e-3, np.abs(o[upd, 38]))).all(), "plane.radius"
    # plane_var (VM:1356-1383): the normal's sign is free, which flips the sign of the normal rows/columns consistently:
    # block (0:3,0:3) and (3:6,3:6) are sign-free, the cross blocks change sign with the normal
    sgn = np.sign((g[upd, 35:38] * o[upd, 35:38]).sum(1))
    G = gpv[upd, 5:41].reshape(-1, 6, 6).copy(); O = opv[upd].reshape(-1, 6, 6)
    G[:, :3, 3:] *= sgn[:, None, None]; G[:, 3:, :3] *= sgn[:, None, None]
    sc = np.abs(O).reshape(len(O), -1).max(1)
    err = np.abs(G - O).reshape(len(O), -1).max(1)
    # plane_var divides by (lambda0 - lambda_k): the agreement inherits the eigenvalue agreement (1e-12 * second moments) over the gap
    # (over ~2.5e4 planes a few have lambda_1 within ~1e-6 of lambda_0's scale of the gap: their agreement is the eigenvalues' / gap)
    assert (err <= 1e-5 * sc + 1e-18).all(), ("plane_var", float((err / np.maximum(sc, 1e-300)).max()))
    assert np.quantile(err / np.maximum(sc, 1e-300), 0.99) < 1e-8 and np.median(err / np.maximum(sc, 1e-300)) < 1e-10
    assert np.array_equal(gpv[:, 41:], oca_s), "cov_add after margi is not bit-identical"


def test_fullsize_session_sums_stay_bit_identical(capi, oracle, synth):
    """Five steady-state steps at full size (200k points per scan with covariances): multi_margi, slide, pvec_update + cut_voxel_multi,
    multi_recut — both sides on identical poses, so that everything the map accumulates (pcr_add, pcr_fix counts, cov_add, through
    fix_divide over the growing fixed-point chains and subdivide) can be compared bit for bit after every step."""
    import dataclasses
    wl = synth.CONFIGS["hesai200k_w10"]
    W = wl.win_size
    nstep = 5
    traj = dataclasses.replace(wl, name=wl.name + "_traj%d" % (W + nstep), win_size=W + nstep)
    s = synth.make_scans(traj)
    x0 = synth.poses_flat(s["R0"], s["p0"])
    ext = np.concatenate([np.eye(3).ravel(), np.zeros(3)])
    rng = np.random.default_rng(11)
    A = rng.normal(0, 0.003, (15, 15)); cov = A @ A.T + np.eye(15) * 1e-6
    ctx = capi.Context(_opts(capi, wl))
    om = _omap(oracle, wl)
    of = oracle.Factor(W)

    def insert(slot, k):
        state = np.zeros(25); state[1:10] = x0[k, :9]; state[10:13] = x0[k, 9:]
        p_k, v_k = oracle.var_init(s["points"][k], ext, wl.dept_err, wl.beam_err)
        v_w, _ = oracle.pvec_update(p_k, v_k, state, cov)
        om.cut_voxel(slot, p_k, x0[k], var=v_w, multi=True)
        ctx.pvec_update_cut_voxel(slot, p_k, v_k, x0[k], cov, multi=True)

    def compare(tag):
        g, _, _ = _sorted(ctx.dump_leaves()); gpv, _, _ = _sorted(ctx.dump_plane_var())     # (two dumps, each in its own order)
        od = om.dump_leaves()
        o, oca, _ = _sorted(od, om.dump_cov_add())
        _assert_structure_equal(g, o)
        assert np.array_equal(gpv[:, :5], o[:, :5])
        assert np.array_equal(gpv[:, 41:], oca), tag + ": cov_add is not bit-identical"
        return len(o)

    for i in range(W):
        insert(i, i)
    window = x0[:W].copy()
    ctx.recut(W, window, multi=True); om.recut(W, window, of, multi=True)
    n0 = compare("window build")
    last = W - 1
    for step in range(nstep):
        ctx.evaluate_only_residual(window); of.evaluate_only_residual(window)        # the eigen state margi copies (VM:1495-1501)
        ctx.margi(W, window, jour=float(last)); om.margi(W, window, of, jour=float(last))
        ctx.slide(1); om.slide(1)
        last += 1
        insert(W - 1, last)
        window = x0[last - W + 1:last + 1].copy()
        ctx.recut(W, window, multi=True); om.recut(W, window, of, multi=True)
        assert ctx.size() == of.size()
        n = compare("step %d" % step)
    assert n > n0 > 20000


def test_fullsize_rebuild_is_deterministic(capi, synth, scans):
    """No f64 atomics are left on the map path: two runs give the same bits for the structure, the cluster sums, the covariance sums,
    the eigen-pairs and the planes of every leaf.  The ORDER of the factors in the store still depends on the run (node ids and
    the extraction's positions come from integer atomics), so H / g / residual — sums over the store in store order — agree to
    summation-order rounding, not to the bit."""
    wl, s = scans
    W = wl.win_size
    poses = synth.poses_flat(s["R0"], s["p0"])
    dumps, sizes, hs, lm = [], [], [], []
    for rep in range(2):
        ctx = capi.Context(_opts(capi, wl))
        for i in range(W):
            ctx.cut_voxel(i, s["points"][i], poses[i])
        ctx.recut(W, poses, multi=False)
        d, _, _ = _sorted(ctx.dump_leaves())
        dumps.append(d); sizes.append(ctx.size())
        hs.append(ctx.acc_evaluate2(poses))
        lm.append(ctx.lidar_ba_damping_iter(poses, max_iter=3, thd_num=2))
        ctx.close()
    a, b = dumps
    assert sizes[0] == sizes[1]
    assert np.array_equal(a[:, :9], b[:, :9]), "keys / layers / paths / counts / plane flags / isexist differ between two runs"
    assert np.array_equal(a[:, 9] >= 0, b[:, 9] >= 0), "factor leaf sets differ between two runs"
    # (column 9, opt_state, is a position in the extraction order)
    assert np.array_equal(a[:, 10:], b[:, 10:]), "eigen-pairs / sums / planes differ between two runs"
    assert np.abs(hs[0][0] - hs[1][0]).max() < 1e-11 * np.abs(hs[0][0]).max()
    assert abs(hs[0][2] - hs[1][2]) < 1e-12 * abs(hs[0][2])
    assert np.abs(lm[0]["poses"] - lm[1]["poses"]).max() < 1e-10 and np.allclose(lm[0]["trace"], lm[1]["trace"], rtol=1e-7, atol=1e-13)
