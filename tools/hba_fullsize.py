"""BASELINE.json configs[4] at FULL length on one GPU: `nkf` keyframes x 50k points through vba_hba_global (bottom-layer windows of
10 every 5 + the top-level BA over the submaps), with the bottom layer checked against the CPU oracle on a sample of windows.
    python tools/hba_fullsize.py [nkf=2000] [oracle_windows=12]
(too long for the test-suite: the 200-keyframe case is tests/test_gpu_gba.py::test_hba_global_at_scale)
TEST INFRASTRUCTURE (a parity checker too long for the suite): like tests/, it uses the CPU oracle as the CHECKER of the device results, never as a
part of the path it measures."""
import dataclasses, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import voxel_slam_amd  # noqa: F401
from voxel_slam_amd import capi, synth
import oracle_api as oracle

nkf = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nsample = int(sys.argv[2]) if len(sys.argv) > 2 else 12
wd, mg = 10, 5
GBA = dict(gba_voxel_size=2.0, gba_min_eigen_value=0.1, gba_eig=[0.25, 0.25, 0.25, 0.25])
t0 = time.time()
# the synthetic trajectory is a straight line that leaves the room after ~250 keyframes: the session is made of blocks of 200
# keyframes over the same path with independent noise (windows that straddle a block boundary see a pose jump, which BA does not mind)
BLK = 200
clouds, x0s = [], []
for b in range((nkf + BLK - 1) // BLK):
    n_b = min(BLK, nkf - b * BLK)
    wl = dataclasses.replace(synth.CONFIGS["hesai200k_w10"], name="hba_full_%d" % b, win_size=n_b, n_pts=50000, seed=synth.CONFIGS["hesai200k_w10"].seed + 17 * b)
    s = synth.make_scans(wl)
    clouds += [p.astype(np.float32).astype(np.float64) for p in s["points"]]
    x0s.append(synth.poses_flat(s["R0"], s["p0"]))
x0 = np.concatenate(x0s)
npts = sum(len(c) for c in clouds)
print("scene: %d keyframes, %.1f M points (%.1f s to generate)" % (nkf, npts / 1e6, time.time() - t0), flush=True)
ctx = capi.Context(capi.options_from_workload(dataclasses.replace(wl, win_size=wd)))
o = ctx.opt
cfg = oracle.gba_cfg13(GBA["gba_voxel_size"], GBA["gba_min_eigen_value"], GBA["gba_eig"], o.voxel_size, o.min_eigen_value,
                       list(o.plane_eigen_value_thre), o.max_layer)
runs = []
t1 = time.time()
ragged = ctx._ragged(clouds)          # offsets + one [N][3] array: what the C entry point takes (the node keeps its clouds that way)
print("python harness: clouds concatenated in %.2f s (outside the timed call)" % (time.time() - t1), flush=True)
for rep in range(3):
    t1 = time.time()
    e1, e2 = ctx.hba_global(ragged, x0, x0, GBA["gba_voxel_size"], GBA["gba_min_eigen_value"], GBA["gba_eig"], 2, wd, mg)
    dt = time.time() - t1
    runs.append(dt)
    print("vba_hba_global run %d: %.2f s wall (host clouds uploaded inside), %d bottom edges, %d top edges" % (rep, dt, len(e1), len(e2)), flush=True)
nwin = (nkf - wd) // mg + 1
assert len(e1) == nwin * 45 and np.isfinite(e1).all() and np.isfinite(e2).all() and len(e2) > nwin
# bottom layer against the oracle on a sample of windows
starts = sorted(set(np.linspace(0, nwin - 1, nsample).astype(int).tolist()))
worst_p = worst_w = 0.0
t2 = time.time()
for wi in starts:
    st = wi * mg
    r = oracle.hba_add_edge(clouds[st:st + wd], x0[st:st + wd], cfg, 1, 2)
    assert r["status"] == 0
    ee = r["edges"].copy(); ee[:, :2] += st
    g = e1[wi * 45:(wi + 1) * 45]
    assert np.array_equal(g[:, :2], ee[:, :2])
    worst_p = max(worst_p, np.abs(g[:, 2:14] - ee[:, 2:14]).max())
    worst_w = max(worst_w, np.quantile(np.abs(g[:, 14:] / ee[:, 14:] - 1), 0.99))
print("oracle on %d sampled windows (%.1f s): relative-pose entries differ by <= %.2e, edge weights (q99) by <= %.2e relative"
      % (len(starts), time.time() - t2, worst_p, worst_w))
assert worst_p < 1e-6 and worst_w < 1e-4
if len(sys.argv) > 3:        # record for bench.py (profiles/rNN_hba_fullsize.json)
    import json
    json.dump({"workload": "BASELINE configs[4]: %d keyframes x 50000 pts (%.1f M points), windows of %d every %d + the top-level window over %d submaps, ONE MI355X"
                           % (nkf, npts / 1e6, wd, mg, nwin),
               "seconds_per_call": float(min(runs[1:])), "runs_s": runs, "bottom_edges": int(len(e1)), "top_edges": int(len(e2)),
               "keyframes_per_s": nkf / float(min(runs[1:])),
               "oracle_check": {"windows": len(starts), "relative_pose_max_abs": float(worst_p), "edge_weight_q99_rel": float(worst_w)},
               "what": "vba_hba_global wall time, host clouds (one concatenated array) uploaded inside the call; measured by tools/hba_fullsize.py on the GPU box"},
              open(sys.argv[3], "w"), indent=1)
print("OK")
