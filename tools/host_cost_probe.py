"""Host-side cost of the LM entry points (time to ENQUEUE, not to run): bench window, no synchronisation inside the loops."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import voxel_slam_amd  # noqa: F401
from voxel_slam_amd import capi, synth
wl = synth.CONFIGS["hesai200k_w10"]
s = synth.make_scans(wl)
poses = synth.poses_flat(s["R0"], s["p0"])
W = wl.win_size
ctx = capi.Context(capi.options_from_workload(wl))
for i in range(W):
    ctx.cut_voxel(i, s["points"][i], poses[i])
ctx.recut(W, poses, multi=False)
def t(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = time.perf_counter() - t0; torch.cuda.synchronize()
    return 1e6 * dt / n, 1e6 * (time.perf_counter() - t0) / n
ctx.lm_begin(poses, thd_num=2); ctx.lm_refresh_eigen()
for _ in range(10): ctx.lm_iterate(sync=False)
print("lm_iterate        enqueue %.1f us, with drain %.1f us per call" % t(lambda: ctx.lm_iterate(sync=False), 60))
print("lm_refresh_eigen  enqueue %.1f us, with drain %.1f us per call" % t(lambda: ctx.lm_refresh_eigen(), 200))
ctx.lm_end(fetch=False)
def call():
    ctx.lm_begin(poses, thd_num=2); ctx.lm_refresh_eigen()
    for _ in range(3): ctx.lm_iterate(sync=False)
    ctx.lm_end(fetch=False)
print("begin+refresh+3 iterate+end  enqueue %.1f us, with drain %.1f us per call" % t(call, 100))
print("lm_begin alone    enqueue %.1f us" % t(lambda: (ctx.lm_begin(poses, thd_num=2), ctx.lm_end(fetch=False)), 100)[0])
