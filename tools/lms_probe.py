"""Wall-clock split of one local-mapping step (marginalise + slide, insert, recut, 3 LM iterations, fetch)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import voxel_slam_amd
from voxel_slam_amd import capi, synth
from collections import deque
wl = synth.CONFIGS["hesai200k_w10"]
s = synth.make_scans(wl)
W = wl.win_size
poses0 = synth.poses_flat(s["R0"], s["p0"])
ctx = capi.Context(capi.options_from_workload(wl))
for i in range(W): ctx.cut_voxel(i, s["points"][i], poses0[i], multi=True)
ctx.recut(W, poses0, multi=True)
win = deque(range(W)); nxt = 0
T = {k: 0.0 for k in ("margi", "slide", "insert", "recut", "lm_enqueue", "lm_end")}
def tick(name, t0):
    ctx.synchronize(); T[name] += time.perf_counter() - t0
N = 20
for it in range(N + 3):
    if it == 3: T = {k: 0.0 for k in T}
    pw = np.ascontiguousarray(poses0[list(win)])
    t0 = time.perf_counter(); ctx.margi(W, pw, jour=0.0); tick("margi", t0)
    t0 = time.perf_counter(); ctx.slide(1); tick("slide", t0)
    win.popleft(); win.append(nxt)
    t0 = time.perf_counter(); ctx.cut_voxel(W - 1, s["points"][nxt], poses0[nxt], multi=True); tick("insert", t0)
    nxt = (nxt + 1) % W
    pw = np.ascontiguousarray(poses0[list(win)])
    t0 = time.perf_counter(); ctx.recut(W, pw, multi=True); tick("recut", t0)
    t0 = time.perf_counter(); ctx.lm_begin(pw, thd_num=2)
    for _ in range(3): ctx.lm_iterate(sync=False)
    tick("lm_enqueue", t0)
    t0 = time.perf_counter(); ctx.lm_end(fetch=True); tick("lm_end", t0)
print({k: round(1e6 * v / N, 1) for k, v in T.items()}, "us per step; total", round(1e6 * sum(T.values()) / N, 1))
