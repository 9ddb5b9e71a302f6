"""Top-level HBA_add_edge at W = 60 submaps (sparse path) vs the CPU port.
TEST INFRASTRUCTURE (a timing comparison too long for the suite): like tests/, it uses the CPU oracle as the CHECKER of the device results, never as a
part of the path it measures."""
import sys, os, time, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import voxel_slam_amd
from voxel_slam_amd import capi, synth
import oracle_api
W = int(sys.argv[1]) if len(sys.argv) > 1 else 60
wl = dataclasses.replace(synth.CONFIGS["room20k_w4"], name="room_w%d" % W, win_size=W, n_pts=10000)
s = synth.make_scans(wl)
clouds = [p.astype(np.float32).astype(np.float64) for p in s["points"]]
poses = synth.poses_flat(s["R0"], s["p0"])
ctx = capi.Context(capi.options_from_workload(synth.CONFIGS["hesai200k_w10"]))
gba = (2.0, 0.1, [0.25] * 4)
out = ctx.hba_add_edge(clouds, poses, *gba, 2, 5)
t0 = time.perf_counter(); out = ctx.hba_add_edge(clouds, poses, *gba, 2, 5); t_gpu = time.perf_counter() - t0
o = ctx.opt
cfg = oracle_api.gba_cfg13(gba[0], gba[1], gba[2], o.voxel_size, o.min_eigen_value, list(o.plane_eigen_value_thre), o.max_layer)
t0 = time.perf_counter(); ref = oracle_api.hba_add_edge(clouds, poses, cfg, 2, 5); t_cpu = time.perf_counter() - t0
print("W", W, "points", sum(len(c) for c in clouds), "edges", len(out["edges"]), len(ref["edges"]), "pose diff", np.abs(out["poses"] - ref["poses"]).max(),
      "gpu %.1f ms  cpu port %.1f ms" % (1e3 * t_gpu, 1e3 * t_cpu), "resis", out["resis"].tolist())
