import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voxel_slam_amd
from voxel_slam_amd import capi, synth
wl = synth.CONFIGS["hesai200k_w10"]
s = synth.make_scans(wl)
poses = synth.poses_flat(s["R0"], s["p0"])
ctx = capi.Context(capi.options_from_workload(wl))
for i in range(wl.win_size): ctx.cut_voxel(i, s["points"][i], poses[i])
ctx.recut(wl.win_size, poses, multi=False)
for k in range(4):
    ctx.evaluate_only_residual(poses)
