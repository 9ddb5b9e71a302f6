"""top-level window probe: W submaps x 5000 pts, device only"""
import sys, os, time, dataclasses
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import voxel_slam_amd
from voxel_slam_amd import capi, synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 200
wl = dataclasses.replace(synth.CONFIGS["hesai200k_w10"], name="top_w%d" % W, win_size=W, n_pts=5000)
s = synth.make_scans(wl)
clouds = [p.astype(np.float32).astype(np.float64) for p in s["points"]]
poses = synth.poses_flat(s["R0"], s["p0"])
ctx = capi.Context(capi.options_from_workload(synth.CONFIGS["hesai200k_w10"]))
gba = (2.0, 0.1, [0.25] * 4)
out = ctx.hba_add_edge(clouds, poses, *gba, 2, 5, want_cloud=False)
t0 = time.perf_counter(); out = ctx.hba_add_edge(clouds, poses, *gba, 2, 5, want_cloud=False); t_gpu = time.perf_counter() - t0
print("W", W, "edges", len(out["edges"]), "gpu %.1f ms" % (1e3 * t_gpu), "factors", ctx.size())
