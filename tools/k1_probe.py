"""Per-scan insertion (K1) under rocprofv3 --kernel-trace: a steady-state window with per-point covariances (the node's path:
pvec_update + cut_voxel_multi, multi_recut, multi_margi), and the same without covariances.  tools/k1_trace_summary.py prints
the kernel sequence of one scan from the trace."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import voxel_slam_amd  # noqa
from voxel_slam_amd import capi, synth

wl = synth.CONFIGS["hesai200k_w10"]
s = synth.make_scans(wl); W = wl.win_size
poses = synth.poses_flat(s["R0"], s["p0"])
ext = np.concatenate([np.eye(3).ravel(), np.zeros(3)])
cov = np.eye(15) * 1e-6
_c0 = capi.Context(capi.options_from_workload(wl))
pv = [_c0.var_init(s["points"][i], ext, wl.dept_err, wl.beam_err) for i in range(W)]      # body-frame covariances on the device (k_var_init)
_c0.close()
for with_var in (True, False):
    ctx = capi.Context(capi.options_from_workload(wl))
    for i in range(W):
        if with_var: ctx.pvec_update_cut_voxel(i, pv[i][0], pv[i][1], poses[i], cov, multi=True)
        else: ctx.cut_voxel(i, s["points"][i], poses[i], multi=True)
    ctx.recut(W, poses, multi=True)
    for k in range(6):
        ctx.margi(W, poses, jour=float(k)); ctx.slide(1)
        if with_var: ctx.pvec_update_cut_voxel(W - 1, pv[k][0], pv[k][1], poses[W - 1], cov, multi=True)
        else: ctx.cut_voxel(W - 1, s["points"][k], poses[W - 1], multi=True)
        ctx.recut(W, poses, multi=True)
    print("with_var", with_var, "factors", ctx.size())
    ctx.close()
