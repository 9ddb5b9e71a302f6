"""Prints LM traces (r1, r2, u, v, q1) of long damping_iter calls: how many consecutive rejections occur (speculative damping)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voxel_slam_amd  # noqa: F401
from voxel_slam_amd import capi, synth
for name in sys.argv[1:] or ("spin40k_w10", "avia100k_w10"):
    import dataclasses
    wl = dataclasses.replace(synth.CONFIGS["hesai200k_w10"], name=name, n_pts=40000) if name == "spin40k_w10" else synth.CONFIGS[name]
    s = synth.make_scans(wl)
    fac = synth.root_factors(s["points"], s["R0"], s["p0"], wl)
    poses = synth.poses_flat(s["R0"], s["p0"])
    ctx = capi.Context(capi.options_from_workload(wl)); ctx.push_dict(fac)
    a = ctx.lidar_ba_damping_iter(poses, max_iter=14, thd_num=2)
    print(name, a["trace"].shape)
    for r in a["trace"]:
        print("  %.10e %.10e u=%.3e v=%g %s" % (r[0], r[1], r[2], r[3], "rej" if r[1] >= r[0] else "acc"))
    ctx.close()
