"""vba_hba_global on a small hierarchy (60 keyframes x 20k points: 11 windows + a top level of 11 submaps), timed."""
import sys, os, time, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voxel_slam_amd  # noqa
from voxel_slam_amd import capi, synth
nk = int(sys.argv[1]) if len(sys.argv) > 1 else 60
wk = dataclasses.replace(synth.CONFIGS["room20k_w4"], name="hba_kf%d" % nk, win_size=nk, n_pts=20000)
sk = synth.make_scans(wk)
clouds = [p.astype(np.float32).astype(np.float64) for p in sk["points"]]
x0 = synth.poses_flat(sk["R0"], sk["p0"])
ctx = capi.Context(capi.options_from_workload(synth.CONFIGS["hesai200k_w10"]))
gba = (2.0, 0.1, [0.25] * 4)
rag = ctx._ragged(clouds)
for rep in range(4):
    t0 = time.perf_counter(); e1, e2 = ctx.hba_global(rag, x0, x0, *gba, 2); dt = time.perf_counter() - t0
    print("hba_global %d keyframes: %.2f ms, edges %d + %d" % (nk, 1e3 * dt, len(e1), len(e2)), flush=True)
