"""How close are the map's sums to the CPU oracle's, bit for bit?  Prints the number of leaves whose pcr_add / cov_add differ at all
and the largest relative difference, after the inserts, after the recut and after a marginalisation + next scan.
TEST INFRASTRUCTURE (a parity checker too long for the suite): like tests/, it uses the CPU oracle as the CHECKER of the device results, never as a
part of the path it measures."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import voxel_slam_amd  # noqa
from voxel_slam_amd import capi, synth
import oracle_api as oracle
import dataclasses


def srt(d, extra=None):
    o = np.lexsort((d[:, 4], d[:, 3], d[:, 2], d[:, 1], d[:, 0]))
    return d[o], (extra[o] if extra is not None else None)


def report(tag, ctx, om):
    g, _ = srt(ctx.dump_leaves()); gpv, _ = srt(ctx.dump_plane_var())
    o, oca = srt(om.dump_leaves(), om.dump_cov_add())
    if g.shape != o.shape or not np.array_equal(g[:, :5], o[:, :5]):
        print(tag, "STRUCTURE DIFFERS", g.shape, o.shape); return
    da = (g[:, 22:32] != o[:, 22:32]).any(1); df = (g[:, 12 + 0:12] != 0).any() if False else None
    rel = np.abs(g[:, 22:32] - o[:, 22:32]).max() / max(1.0, np.abs(o[:, 22:32]).max())
    dc = (gpv[:, 41:] != oca).any(1)
    relc = (np.abs(gpv[:, 41:] - oca).max(1) / np.maximum(np.abs(oca).max(1), 1e-300)).max()
    print("%-28s leaves %6d | pcr_add differs on %6d (max rel %.2e) | cov_add differs on %6d (max rel %.2e) | layers %s"
          % (tag, len(g), int(da.sum()), rel, int(dc.sum()), relc, np.bincount(g[:, 3].astype(int)).tolist()))
    if da.sum():
        bad = np.where(da)[0][:3]
        for b in bad: print("   leaf", g[b, :5].tolist(), "N", g[b, 31], o[b, 31], "dP", (g[b, 22:31] - o[b, 22:31]).tolist()[:4])


name = sys.argv[1] if len(sys.argv) > 1 else "hesai200k_w10"
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 0
wl = synth.CONFIGS[name]
if npts: wl = dataclasses.replace(wl, n_pts=npts)
s = synth.make_scans(wl); W = wl.win_size
poses = synth.poses_flat(s["R0"], s["p0"])
ext = np.concatenate([np.eye(3).ravel(), np.zeros(3)])
rng = np.random.default_rng(5); A = rng.normal(0, 0.003, (15, 15)); cov = A @ A.T + np.eye(15) * 1e-6
ctx = capi.Context(capi.options_from_workload(wl))
om = oracle.VoxelMap(W, wl.voxel_size, wl.max_layer, wl.min_eigen_value, wl.plane_thre, wl.min_point, wl.max_points, 5)
for i in range(W):
    state = np.zeros(25); state[1:10] = poses[i, :9]; state[10:13] = poses[i, 9:]
    p_i, v_i = oracle.var_init(s["points"][i], ext, wl.dept_err, wl.beam_err)
    v_w, _ = oracle.pvec_update(p_i, v_i, state, cov)
    om.cut_voxel(i, p_i, poses[i], var=v_w, multi=True)
    ctx.pvec_update_cut_voxel(i, p_i, v_i, poses[i], cov, multi=True)
report("after %d inserts" % W, ctx, om)
of = oracle.Factor(W)
ctx.recut(W, poses, multi=True); om.recut(W, poses, of, multi=True)
report("after recut", ctx, om)
print("factors", ctx.size(), of.size())
f = ctx  # factor store: body clusters must be exact too
H, g_, r = ctx.acc_evaluate2(poses); H2, g2, r2 = of.acc_evaluate2(poses)
print("H rel", np.abs(H - H2).max() / np.abs(H2).max(), "r rel", abs(r - r2) / abs(r2))
b = of.lidar_ba_damping_iter(poses, max_iter=3, thd_num=2, parallel=True)
a = ctx.lidar_ba_damping_iter(poses, max_iter=3, thd_num=2)
print("LM pose diff", np.abs(a["poses"] - b["poses"]).max())
ctx.evaluate_only_residual(b["poses"]); of.evaluate_only_residual(b["poses"])
ctx.margi(W, b["poses"], jour=1.0); om.margi(W, b["poses"], of, jour=1.0)
report("after margi", ctx, om)
ctx.slide(1); om.slide(1)
x2 = np.concatenate([b["poses"][1:], b["poses"][-1:]])
state = np.zeros(25); state[1:10] = x2[W - 1, :9]; state[10:13] = x2[W - 1, 9:]
p_i, v_i = oracle.var_init(s["points"][0], ext, wl.dept_err, wl.beam_err)
v_w, _ = oracle.pvec_update(p_i, v_i, state, cov)
om.cut_voxel(W - 1, p_i, x2[W - 1], var=v_w, multi=True)
ctx.pvec_update_cut_voxel(W - 1, p_i, v_i, x2[W - 1], cov, multi=True)
report("next scan inserted", ctx, om)
of2 = oracle.Factor(W)
ctx.recut(W, x2, multi=True); om.recut(W, x2, of2, multi=True)
report("next recut", ctx, om)
