"""Timing of the odometry scan-to-map update (vba_odom_lio_state_estimation) on the bench-size map: 200k-point scan."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voxel_slam_amd
from voxel_slam_amd import capi, synth
wl = synth.CONFIGS["hesai200k_w10"]
s = synth.make_scans(wl)
W = wl.win_size
poses = synth.poses_flat(s["R_gt"], s["p_gt"])
ctx = capi.Context(capi.options_from_workload(wl))
rng = np.random.default_rng(1)
def rand_var(n, scale=0.01):
    A = rng.normal(0, scale, (n, 3, 3))
    return np.ascontiguousarray((A @ A.transpose(0, 2, 1) + 1e-6 * np.eye(3)).reshape(n, 9))
for i in range(W):
    ctx.cut_voxel(i, s["points"][i], poses[i], var=rand_var(len(s["points"][i])), multi=True)
ctx.recut(W, poses, multi=True)
ctx.margi(W, poses, jour=0.0)          # refreshes the planes (plane_update)
k = W - 1
state = np.zeros(25); state[1:10] = s["R_gt"][k].ravel(); state[10:13] = s["p_gt"][k] + 0.01; state[22:25] = [0, 0, -9.8]
cov = np.eye(15) * 1e-4
pts = s["points"][k]; var_b = rand_var(len(pts), 0.005)
ok, st, cv = ctx.lio_state_estimation(pts, var_b, state, cov)
t0 = time.perf_counter()
for _ in range(10): ok, st, cv = ctx.lio_state_estimation(pts, var_b, state, cov)
print("lio_state_estimation: %.2f ms per call (%d points, ok=%s)" % (1e2 * (time.perf_counter() - t0), len(pts), ok))
