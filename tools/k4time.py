"""Device time of the residual pass (K4) on the bench window, from hipEvents around single isolated launches minus the span of an
empty event pair.  RELATIVE comparisons only (kernel variants, sizes): the subtraction under-reports the launch duration by 1-2 us
("A measurement correction"); bench.py times a graph-replayed batch instead.
    python tools/k4time.py [copies]        # copies > 1: the scene tiled `copies` times (V x copies)
    VBA_LIB=build/libvoxelba_tv64.so python tools/k4time.py     # another build of the library (make -C voxel-slam_amd/csrc variants)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voxel_slam_amd  # noqa: F401
from voxel_slam_amd import capi, synth

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 1
wl = synth.CONFIGS["hesai200k_w10"]
s = synth.make_scans(wl)
poses = synth.poses_flat(s["R0"], s["p0"])
W = wl.win_size
ctx = capi.Context(capi.options_from_workload(wl))
R = poses[:, :9].reshape(W, 3, 3)
for i in range(W):
    pts = s["points"][i]
    if copies > 1:
        pts = np.concatenate([pts + (R[i].T @ np.array([100.0 * (c % 8), 100.0 * (c // 8), 0.0]))[None, :] for c in range(copies)])
    ctx.cut_voxel(i, pts, poses[i])
ctx.recut(W, poses, multi=False)
V, occ = ctx.size(), ctx.factor_occupancy()
for _ in range(5):
    ctx.evaluate_only_residual(poses)
ctx.timing_enable(True); ctx.timing_select("residual"); ctx.timing_reset()
for _ in range(200 if copies == 1 else 30):
    ctx.evaluate_only_residual(poses)
t, n = ctx.timing_get("residual")
null = ctx.timing_null_spans(64)
us = t / n - null
by = V * ((occ + 1) * 80 + 8 + 176)
print("lib %s: V=%d occ=%.2f  K4 %.2f us (raw %.2f, empty span %.2f) -> %.0f GB/s algorithmic (%.3f of 8 TB/s)"
      % (os.path.basename(capi.LIB_PATH), V, occ, us, t / n, null, by / us * 1e-3, by / us * 1e-3 / 8000))
