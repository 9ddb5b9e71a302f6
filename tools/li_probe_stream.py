import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import voxel_slam_amd
from voxel_slam_amd import capi, synth
wl = synth.CONFIGS["hesai200k_w10"]
s = synth.make_scans(wl)
W = wl.win_size
poses = synth.poses_flat(s["R0"], s["p0"])
for mode in ("null", "side"):
    st = torch.cuda.current_stream().cuda_stream if mode == "null" else torch.cuda.Stream().cuda_stream
    ctx = capi.Context(capi.options_from_workload(wl, stream=st))
    for i in range(W): ctx.cut_voxel(i, s["points"][i], poses[i])
    ctx.recut(W, poses, multi=False)
    imu_samples, vel, g = synth.make_imu(wl, gyr_sigma=1e-3, acc_sigma=1e-2)
    nm = np.array([0.01] * 3 + [1.0] * 3); nw = np.array([1e-4] * 6)
    imus = np.stack([capi.imu_preintegrate(t, gy, ac, np.zeros(3), np.zeros(3), nm, nw) for (t, gy, ac) in imu_samples])
    states = np.zeros((W, 25))
    for i in range(W):
        states[i, 0] = 0.1 * i; states[i, 1:10] = s["R0"][i].ravel(); states[i, 10:13] = s["p0"][i]; states[i, 13:16] = vel[i]; states[i, 22:25] = g
    out = ctx.li_ba_damping_iter(states, imus, gravity=False, max_iter=3)
    t0 = time.perf_counter(); n = 0
    for _ in range(20):
        ctx.evaluate_only_residual(poses)
        out = ctx.li_ba_damping_iter(states, imus, gravity=False, max_iter=3); n += len(out["trace"])
    print(mode, "stream:", 1e6 * (time.perf_counter() - t0) / n, "us per iteration")
    ctx.close()
