"""How much of an LM call's device time is launch gaps?  The bench call (begin, restart pass, 3 iterations, end) enqueued directly
vs the same launches replayed from a HIP graph (begin's upload stays outside the graph)."""
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
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = capi.Context(capi.options_from_workload(wl, stream=st.cuda_stream))
for i in range(W):
    ctx.cut_voxel(i, s["points"][i], poses[i])
ctx.recut(W, poses, multi=False)
def body():
    ctx.lm_refresh_eigen()
    for _ in range(3):
        ctx.lm_iterate(sync=False)
def call_direct():
    ctx.lm_begin(poses, thd_num=2); body(); ctx.lm_end(fetch=False)
for _ in range(5): call_direct()
torch.cuda.synchronize()
N = 100
t0 = time.perf_counter()
for _ in range(N): call_direct()
torch.cuda.synchronize()
print("direct: %.1f us per call (3 iterations)" % (1e6 * (time.perf_counter() - t0) / N))
ctx.lm_begin(poses, thd_num=2)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=st):
    body()
ctx.lm_end(fetch=False)
def call_graph():
    ctx.lm_begin(poses, thd_num=2); g.replay(); ctx.lm_end(fetch=False)
for _ in range(5): call_graph()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N): call_graph()
torch.cuda.synchronize()
print("graph : %.1f us per call (3 iterations)" % (1e6 * (time.perf_counter() - t0) / N))
tr = None
ctx.lm_begin(poses, thd_num=2); g.replay(); ctx.lm_end(fetch=True); a = ctx.last_trace().copy()
ctx.lm_begin(poses, thd_num=2); body(); ctx.lm_end(fetch=True); b = ctx.last_trace().copy()
print("traces equal:", np.array_equal(a, b), a.shape)
