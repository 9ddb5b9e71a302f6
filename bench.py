#!/usr/bin/env python3
"""bench.py — local-BA iterations/sec on the BASELINE.json workload (Hesai-32 synthetic scans, 200k pts, W=10, 0.3 m).

One "step" = one Levenberg-Marquardt iteration of the sliding-window BA on the full window, i.e. one trip through
the loop body voxel_map.hpp:441-494: Hessian pass K3 (acc_evaluate2) -> cross-rank sum -> gauge/damp/LDLT solve ->
retraction -> residual pass K4 (evaluate_only_residual) -> accept/reject.  Every 3 steps (= one damping_iter call)
the window restarts from the perturbed initial poses; the restart's residual pass (which re-creates the per-voxel
eigen state that recut/tras_opt hand to damping_iter, voxel_map.hpp:1628) runs inside the timed region but is not
counted as a step.  Inputs (the factor store) are resident in HBM before the timed region.

N > 1: one rank per GPU.  Under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment) this process is one
rank; called plainly with --gpus N it starts the N rank processes itself (before it touches the GPU) and fails if the node has
fewer than N devices.  Voxels are sharded by root-voxel hash bucket (vba_shard_owner), each rank evaluates its shard and the packed
[H|g|r] buffer is summed by the library's own RCCL communicator (vba_rccl_init: ncclAllReduce on the context's stream) — total work
is fixed, so scaling is "strong".
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# the headline timed region brackets every 8th K4 launch with a hipEvent pair (cross-check only): a pair costs ~5 us of stream time,
# more than the kernel, so bracketing every launch would put the measurement itself into `value`
K4_SAMPLE_EVERY = 8
K4_BATCH = 100
FP64_PEAK_TFLOPS = 78.6  # MI355X vector/matrix FP64


def _tensor_from_ptr(torch, ptr, n, cache={}):
    key = (ptr, n)
    if key not in cache:
        class _Ext:
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2, "strides": None}
        cache[key] = torch.as_tensor(_Ext(), device="cuda")
    return cache[key]


def build_problem(wl_name):
    """Synthetic scans + perturbed initial poses (workload synthesis, untimed)."""
    from voxel_slam_amd import synth
    wl = synth.CONFIGS[wl_name]
    s = synth.make_scans(wl)
    poses0 = synth.poses_flat(s["R0"], s["p0"])
    return wl, s, poses0


def cpu_baseline(wl, scans, poses0, budget_s=12.0):
    """The CPU oracle (faithful restatement of the reference path) timed on this host: the window's factors are built
    by the oracle's own octree (cut_voxel x W, recut, tras_opt), then damping_iter(max_iter=3) with the reference's
    5 std::thread workers (voxel_map.hpp:521) is repeated on them; iterations/s over a bounded sample.  Also timed once
    each: the per-stage K1 / K2 / K5 work and the LiDAR-inertial optimiser the node runs per scan."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api
    oracle_api.build()
    from voxel_slam_amd import synth, capi
    W = wl.win_size
    om = oracle_api.VoxelMap(W, wl.voxel_size, wl.max_layer, wl.min_eigen_value, wl.plane_thre, wl.min_point, wl.max_points, 5)
    t0 = time.perf_counter()
    t_scan = []
    for i in range(W):
        t1 = time.perf_counter()
        om.cut_voxel(i, scans["points"][i], poses0[i], multi=True)
        t_scan.append(time.perf_counter() - t1)
    t_ins = time.perf_counter() - t0
    f0 = oracle_api.Factor(W)
    t1 = time.perf_counter()
    om.recut(W, poses0, f0, multi=True)
    t_recut = time.perf_counter() - t1
    t_build = time.perf_counter() - t0
    fac = f0.as_dict()
    iters = 0
    t_tot = 0.0
    reps = 0
    n_rej = 0
    while t_tot < budget_s and reps < 50:
        f = oracle_api.Factor(W)
        f.push_dict(fac)
        t0 = time.perf_counter()
        out = f.lidar_ba_damping_iter(poses0, max_iter=3, thd_num=5, parallel=True)
        t_tot += time.perf_counter() - t0
        iters += len(out["trace"])
        n_rej += int((out["trace"][:, 1] >= out["trace"][:, 0]).sum())
        reps += 1
    # courtesy number: the same loop with one worker per core of this box's CPU share
    nthr = max(1, min(len(os.sched_getaffinity(0)), 64))
    it_all, t_all, reps_all = 0, 0.0, 0
    while t_all < budget_s / 3 and reps_all < 20:
        f = oracle_api.Factor(W)
        f.push_dict(fac)
        t0 = time.perf_counter()
        out = f.lidar_ba_damping_iter(poses0, max_iter=3, thd_num=nthr, parallel=True)
        t_all += time.perf_counter() - t0
        it_all += len(out["trace"])
        reps_all += 1
    # LI_BA_Optimizer::damping_iter (voxel_map.hpp:624-713, the call at voxelslam.cpp:1969) on the same factors, 5 threads
    imu_samples, vel, g = synth.make_imu(wl, gyr_sigma=1e-3, acc_sigma=1e-2)
    nm = np.array([0.01] * 3 + [1.0] * 3); nw = np.array([1e-4] * 6)
    imus = np.stack([oracle_api.imu_preintegrate(t, gy, ac, np.zeros(3), np.zeros(3), nm, nw) for (t, gy, ac) in imu_samples])
    states = np.zeros((W, 25))
    for i in range(W):
        states[i, 0] = 0.1 * i; states[i, 1:10] = poses0[i, :9]; states[i, 10:13] = poses0[i, 9:12]; states[i, 13:16] = vel[i]; states[i, 22:25] = g
    li_it, li_t, li_reps = 0, 0.0, 0
    while li_t < budget_s / 3 and li_reps < 20:
        f = oracle_api.Factor(W)
        f.push_dict(fac)
        t0 = time.perf_counter()
        out = f.li_ba_damping_iter(states, imus, gravity=False, imu_coef=wl.imu_coef, max_iter=3, parallel=True)
        li_t += time.perf_counter() - t0
        li_it += len(out["trace"])
        li_reps += 1
    # K5: multi_margi of the oldest scan on the full map (one call; it consumes the factor store's refined state)
    t1 = time.perf_counter()
    om.margi(W, poses0, f0, jour=0.0)
    t_margi = time.perf_counter() - t1
    return dict(value=iters / t_tot, unit="iterations/s", cores=5, kind="port",
                all_cores={"value": it_all / t_all, "threads": nthr, "calls": reps_all},
                li_ba={"value": li_it / li_t, "unit": "iterations/s", "cores": 5, "calls": li_reps,
                       "what": "LI_BA_Optimizer::damping_iter (voxel_map.hpp:624-713) on the same window, 9 IMU factors"},
                stages_ms={"K1_insert_per_200k_scan": 1e3 * float(np.median(t_scan)), "K1_insert_full_window": 1e3 * t_ins,
                           "K2_recut_extract_full_window": 1e3 * t_recut, "K5_margi_full_map": 1e3 * t_margi,
                           "threads": "cut_voxel_multi phase 1 serial + 5 threads (VM:1964-2096), multi_recut / multi_margi 5 threads"},
                rejected_steps=n_rej, iterations=iters,
                sample="%d damping_iter calls (%d LM iterations, %d of them rejected steps that skip the Hessian pass as VM:443 does) on the full "
                       "%d-voxel window, 5 worker threads of %d host cores; full-window rebuild (insert+recut, %d pts) took %.2f s"
                       % (reps, iters, n_rej, len(fac["coe"]), os.cpu_count(), sum(len(p) for p in scans["points"]), t_build))


def alg_bytes_residual(V, occ):
    """the residual pass reads (W_occ + 1) * 80 + 8 bytes and writes 176 bytes per planar voxel."""
    return V * ((occ + 1) * 80 + 8 + 176)


def scaled_residual_pass(capi, torch, wl, scans, poses0, copies=56):
    """The residual pass K4 on `copies` disjoint translated copies of the scene inserted through K1/K2 (V x copies): with 48
    copies one pass touches > 2 x the 256 MiB Infinity Cache, so consecutive launches cannot be served from it."""
    W = wl.win_size
    ctx = capi.Context(capi.options_from_workload(wl, stream=torch.cuda.current_stream().cuda_stream))
    R = poses0[:, :9].reshape(W, 3, 3)
    for i in range(W):
        pts = scans["points"][i]
        tiles = [pts + (R[i].T @ np.array([100.0 * (c % 8), 100.0 * (c // 8), 0.0]))[None, :] for c in range(copies)]
        ctx.cut_voxel(i, np.concatenate(tiles), poses0[i])
        del tiles
    ctx.recut(W, poses0, multi=False)
    V = ctx.size()
    occ = ctx.factor_occupancy()
    for _ in range(3):
        ctx.evaluate_only_residual(poses0)
    ctx.timing_enable(True); ctx.timing_select("residual"); ctx.timing_reset()
    for _ in range(20):
        ctx.evaluate_only_residual(poses0)
    t, n = ctx.timing_get("residual")
    us = max(t / max(n, 1) - ctx.timing_null_spans(32), 1e-3)
    by = alg_bytes_residual(V, occ)
    gbs = by / (us * 1e-6) / 1e9
    ctx.close()
    return {"copies": copies, "voxels": V, "occupied_frames_per_voxel": occ, "avg_launch_us": us, "launches": n, "algorithmic_bytes_per_launch": by,
            "layout_extra_bytes_per_launch": V * 4, "layout_extra_note": "4-byte occupancy mask per voxel", "kernel": "k_residual_v<%d> (one lane per voxel; stores beyond 45k voxels)" % W,
            "exceeds_2x_infinity_cache": bool(by >= 2 * 256 * 2 ** 20),
            "achieved": gbs, "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS}


def local_mapping_step(capi, torch, wl, steps=12):
    """One local-mapping step as the node runs it (voxelslam.cpp:1916-2043): pvec_update + cut_voxel_multi of the newest scan WITH
    per-point covariances, multi_recut, LI_BA_Optimizer::damping_iter (3 iterations, 9 IMU factors, *hess fetched: VS:1969), multi_margi
    with the refined poses, ring rotation.  `lidar_only_ms` is the same step with Lidar_BA_Optimizer instead (round 1-2's definition)."""
    import ctypes as C
    import dataclasses
    from voxel_slam_amd import synth
    W = wl.win_size
    turnover = 6
    nscan = W + turnover + steps
    # a trajectory long enough for every step to see a NEW scan with IMU factors that match it (cycling through the W scans of the
    # headline window would hand the optimiser IMU factors that contradict the poses)
    wll = dataclasses.replace(wl, name=wl.name + "_traj%d" % nscan, win_size=nscan)
    sl = synth.make_scans(wll)
    x0 = synth.poses_flat(sl["R0"], sl["p0"])
    ext = np.concatenate([np.eye(3).ravel(), np.zeros(3)])
    cov = np.eye(15) * 1e-6
    imu_samples, vel, g = synth.make_imu(wll, gyr_sigma=1e-3, acc_sigma=1e-2)
    nm = np.array([0.01] * 3 + [1.0] * 3); nw = np.array([1e-4] * 6)
    imu_all = np.stack([capi.imu_preintegrate(t, gy, ac, np.zeros(3), np.zeros(3), nm, nw) for (t, gy, ac) in imu_samples])
    out = {}
    for name, li in (("li_ba", True), ("lidar_only", False)):
        o = capi.options_from_workload(wl, stream=torch.cuda.current_stream().cuda_stream)
        # capacity hints: a deployment sizes the map once (vba_options), so that no step stalls on a re-allocation
        o.max_points_per_scan, o.max_map_nodes, o.max_fix_points, o.max_voxels = 1 << 18, 1 << 21, 1 << 23, 1 << 17
        ctx = capi.Context(o)
        # body-frame covariances (calcBodyVar, VH:180-234) once per scan on the device, kept in HBM with the points
        pv = [ctx.var_init(p, ext, wl.dept_err, wl.beam_err) for p in sl["points"]]
        dev_p = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a, _ in pv]
        dev_v = [torch.from_numpy(np.ascontiguousarray(b)).cuda() for _, b in pv]

        def insert_dev(slot, idx, pose):
            ctx._chk(ctx.lib.vba_map_pvec_update_cut_voxel(ctx.h, C.c_int(slot), C.c_int(dev_p[idx].shape[0]), C.c_void_p(dev_p[idx].data_ptr()),
                                                          C.c_void_p(dev_v[idx].data_ptr()), pose.ctypes.data_as(C.POINTER(C.c_double)),
                                                          cov.ctypes.data_as(C.POINTER(C.c_double)), C.c_int(1)))
        for i in range(W):
            insert_dev(i, i, x0[i])
        ctx.recut(W, x0[:W], multi=True)
        window = np.ascontiguousarray(x0[:W])          # poses of the scans in the window (refined by the last optimisation)
        last = W - 1                                   # index of the newest scan
        states = np.zeros((W, 25))

        def step():
            nonlocal window, last
            ctx.margi(W, window, jour=float(last))                # multi_margi with the window the last optimisation refined
            ctx.slide(1)
            last += 1
            insert_dev(W - 1, last, x0[last])
            pw = np.ascontiguousarray(np.concatenate([window[1:], x0[last:last + 1]]))
            ctx.recut(W, pw, multi=True)
            if li:
                for i in range(W):
                    k = last - W + 1 + i
                    states[i, 0] = 0.1 * k; states[i, 1:10] = pw[i, :9]; states[i, 10:13] = pw[i, 9:12]; states[i, 13:16] = vel[k]; states[i, 22:25] = g
                r = ctx.li_ba_damping_iter(states, imu_all[last - W + 1:last], gravity=False, max_iter=3)
                window = np.ascontiguousarray(np.concatenate([r["states"][:, 1:10], r["states"][:, 10:13]], 1))
            else:
                ctx.lm_begin(pw, thd_num=2)
                for _ in range(3):
                    ctx.lm_iterate(sync=False)
                window = np.ascontiguousarray(ctx.lm_end(fetch=True)[0])     # the node reads the poses back

        for _ in range(turnover):
            step()
        torch.cuda.synchronize()
        ts = []
        for _ in range(steps):
            t0 = time.perf_counter()
            step()               # (ends with the state fetch: the step is complete when it returns)
            ts.append(1e3 * (time.perf_counter() - t0))
        torch.cuda.synchronize()
        # median: a step that doubles the capacity of a map array is reported separately as the maximum instead of being smeared over the mean
        out[name + "_ms"] = float(np.median(ts)); out[name + "_mean_ms"] = float(np.mean(ts)); out[name + "_max_ms"] = float(np.max(ts))
        out[name + "_planar_voxels"] = ctx.size()
        gt = synth.poses_flat(sl["R_gt"], sl["p_gt"])[last - W + 1:last + 1]
        out[name + "_window_translation_error_m"] = float(np.abs(window[:, 9:] - gt[:, 9:]).max())
        ctx.close()
    out["steps"] = steps
    out["what"] = ("scan (%d pts + 3x3 covariances) resident in HBM; per step: multi_margi + slide, pvec_update + cut_voxel_multi, multi_recut + "
                   "factor extraction, 3 LM iterations (li_ba: LI_BA_Optimizer with 9 IMU factors and *hess read back, voxelslam.cpp:1969; "
                   "lidar_only: Lidar_BA_Optimizer), refined states fetched" % wl.n_pts)
    return out


def cold_residual_pass(ctx, torch, poses0, V, occ, W, reps=12):
    """K4 with the caches flushed between repetitions (a 1 GiB scratch buffer is rewritten before every launch)."""
    scratch = torch.empty(1 << 27, dtype=torch.float64, device="cuda")
    ctx.timing_enable(True); ctx.timing_select("residual"); ctx.timing_reset()
    for k in range(reps):
        scratch.fill_(float(k))
        ctx.evaluate_only_residual(poses0)
    t, n = ctx.timing_get("residual")
    us = max(t / max(n, 1) - ctx.timing_null_spans(32), 1e-3)
    ctx.timing_enable(False); ctx.timing_select(None)
    by = alg_bytes_residual(V, occ)
    del scratch
    return {"avg_launch_us": us, "launches": n, "achieved": by / (us * 1e-6) / 1e9, "unit": "GB/s",
            "frac": by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "flush": "1 GiB device buffer rewritten before each launch"}


def li_variant(ctx, capi, wl, scans, poses0, reps=15):
    """LiDAR-inertial optimiser (LI_BA_Optimizer, voxel_map.hpp:504-714) on the same factor store, device resident: K3/K4, the IMU
    factor kernel, the 150x150 blocked LDL^T and the accept/reject kernel; one call = upload, 3 LM iterations, states + Hessian back."""
    from voxel_slam_amd import synth
    W = wl.win_size
    imu_samples, vel, g = synth.make_imu(wl, gyr_sigma=1e-3, acc_sigma=1e-2)
    nm = np.array([0.01] * 3 + [1.0] * 3); nw = np.array([1e-4] * 6)
    imus = np.stack([capi.imu_preintegrate(t, gy, ac, np.zeros(3), np.zeros(3), nm, nw) for (t, gy, ac) in imu_samples])
    states = np.zeros((W, 25))
    for i in range(W):
        states[i, 0] = 0.1 * i; states[i, 1:10] = poses0[i, :9]; states[i, 10:13] = poses0[i, 9:12]; states[i, 13:16] = vel[i]; states[i, 22:25] = g
    ctx.evaluate_only_residual(poses0)
    ctx.li_ba_damping_iter(states, imus, gravity=False, max_iter=3)
    t0 = time.perf_counter(); n = 0
    for _ in range(reps):
        ctx.evaluate_only_residual(poses0)
        out = ctx.li_ba_damping_iter(states, imus, gravity=False, max_iter=3)
        n += len(out["trace"])
    dt = time.perf_counter() - t0
    return {"iterations_per_s": n / dt, "us_per_iteration": 1e6 * dt / n, "iterations": n,
            "what": "LI_BA_Optimizer::damping_iter, %d IMU factors, 150x150 system; per-call upload/download amortised over its 3 iterations" % (W - 1)}


def hba_window(capi, torch, reps=10, cpu=True):
    """One bottom-layer window of the hierarchical global BA (BASELINE cfg5 shape: 10 keyframes x 50k points, stride-5
    windows are independent): HBA_add_edge(xs, smp_local, gba_edges1, mps, 1, 2, plptr) as thd_globalmapping calls it
    (voxelslam.cpp:3086) = octree build + 4 LM iterations + edges + down-sampled submap cloud."""
    import dataclasses
    from voxel_slam_amd import synth
    wl = dataclasses.replace(synth.CONFIGS["hesai200k_w10"], name="hba50k_w10", n_pts=50000)
    s = synth.make_scans(wl)
    clouds = [p.astype(np.float32).astype(np.float64) for p in s["points"]]
    poses = synth.poses_flat(s["R0"], s["p0"])
    gba = (2.0, 0.1, [0.25] * 4)                       # config/avia.yaml:59-65, ratios inverted
    ctx = capi.Context(capi.options_from_workload(wl, stream=torch.cuda.current_stream().cuda_stream))
    out = ctx.hba_add_edge(clouds, poses, *gba, 1, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = ctx.hba_add_edge(clouds, poses, *gba, 1, 2)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    res = {"workload": "10 keyframes x %d pts, GBA voxel 2 m, max_iter 1, thread_num 2" % wl.n_pts, "window_ms": 1e3 * dt, "windows_per_s": 1.0 / dt,
           "planar_voxels": ctx.size(), "edges": int(len(out["edges"])), "submap_cloud_points": int(len(out["cloud"])),
           "what": "host clouds uploaded per call; octree build, 4 LM iterations, edges, submap cloud fetched"}
    if cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_api
        o = capi.options_from_workload(wl)
        cfg = oracle_api.gba_cfg13(gba[0], gba[1], gba[2], o.voxel_size, o.min_eigen_value, list(o.plane_eigen_value_thre), o.max_layer)
        t0 = time.perf_counter()
        oracle_api.hba_add_edge(clouds, poses, cfg, 1, 2)
        res["cpu_port_window_ms"] = 1e3 * (time.perf_counter() - t0)
    # top layer: all submaps in one window (voxelslam.cpp:3103-3113) -> the sparse any-window path with the dense LDL^T in HBM
    Wt = 60
    wt = dataclasses.replace(synth.CONFIGS["room20k_w4"], name="top_w%d" % Wt, win_size=Wt, n_pts=10000)
    st_ = synth.make_scans(wt)
    clouds_t = [p.astype(np.float32).astype(np.float64) for p in st_["points"]]
    poses_t = synth.poses_flat(st_["R0"], st_["p0"])
    ctx.hba_add_edge(clouds_t, poses_t, *gba, 2, 5, want_cloud=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out_t = ctx.hba_add_edge(clouds_t, poses_t, *gba, 2, 5, want_cloud=False)
    torch.cuda.synchronize()
    res["top_level"] = {"workload": "%d submaps x %d pts in ONE window (360 x 360 system), 2 rounds, thread_num 5" % (Wt, wt.n_pts),
                        "ms": 1e3 * (time.perf_counter() - t0), "edges": int(len(out_t["edges"]))}
    if cpu:
        o = ctx.opt
        cfg = oracle_api.gba_cfg13(gba[0], gba[1], gba[2], o.voxel_size, o.min_eigen_value, list(o.plane_eigen_value_thre), o.max_layer)
        t0 = time.perf_counter()
        oracle_api.hba_add_edge(clouds_t, poses_t, cfg, 2, 5, want_cloud=False)
        res["top_level"]["cpu_port_ms"] = 1e3 * (time.perf_counter() - t0)
    # the whole hierarchy on one map (BASELINE configs[4] shape, reduced): 60 keyframes -> 11 windows of 10 -> 11 submaps -> top BA
    nk = 60
    wk = dataclasses.replace(synth.CONFIGS["room20k_w4"], name="hba_kf%d" % nk, win_size=nk, n_pts=20000)
    sk = synth.make_scans(wk)
    clouds_k = [p.astype(np.float32).astype(np.float64) for p in sk["points"]]
    x0k = synth.poses_flat(sk["R0"], sk["p0"])
    rag_k = ctx._ragged(clouds_k)                    # (the concatenation of the clouds is the harness's, not the call's)
    ctx.hba_global(rag_k, x0k, x0k, *gba, 2)
    torch.cuda.synchronize()
    th = []
    for _ in range(3):                               # median of three: the first calls after another window size still re-size the arena
        t0 = time.perf_counter()
        e1, e2 = ctx.hba_global(rag_k, x0k, x0k, *gba, 2)
        torch.cuda.synchronize()
        th.append(time.perf_counter() - t0)
    res["hierarchy"] = {"workload": "%d keyframes x %d pts: windows of 10 every 5, then the top-level BA over the submaps" % (nk, wk.n_pts),
                        "ms": 1e3 * float(np.median(th)), "edges_bottom": int(len(e1)), "edges_top": int(len(e2))}
    if cpu:
        t0 = time.perf_counter()
        subs, firsts = [], []
        for start in range(0, nk - 10 + 1, 5):
            r = oracle_api.hba_add_edge(clouds_k[start:start + 10], x0k[start:start + 10], cfg, 1, 2)
            subs.append(r["cloud"]); firsts.append(start)
        oracle_api.hba_add_edge(subs, x0k[firsts], cfg, 2, 5, want_cloud=False)
        res["hierarchy"]["cpu_port_ms"] = 1e3 * (time.perf_counter() - t0)
    ctx.close()
    return res


def odometry_update(capi, torch, wl, scans, reps=10, cpu=True):
    """lio_state_estimation (voxelslam.cpp:962-1098) of one 200k-point scan against the full-window map: device point loop
    (world covariance, hash lookup + octant descent, 3-sigma gate, 34 weighted sums) x <= 4 EKF iterations, host 15x15 algebra."""
    from voxel_slam_amd import synth
    W = wl.win_size
    poses = synth.poses_flat(scans["R_gt"], scans["p_gt"])
    ctx = capi.Context(capi.options_from_workload(wl, stream=torch.cuda.current_stream().cuda_stream))
    rng = np.random.default_rng(1)

    def rand_var(n, scale):
        A = rng.normal(0, scale, (n, 3, 3))
        return np.ascontiguousarray((A @ A.transpose(0, 2, 1) + 1e-6 * np.eye(3)).reshape(n, 9))
    for i in range(W):
        ctx.cut_voxel(i, scans["points"][i], poses[i], var=rand_var(len(scans["points"][i]), 0.01), multi=True)
    ctx.recut(W, poses, multi=True)
    ctx.margi(W, poses, jour=0.0)                  # plane_update runs inside margi
    k = W - 1
    state = np.zeros(25); state[1:10] = scans["R_gt"][k].ravel(); state[10:13] = scans["p_gt"][k] + 0.01; state[22:25] = [0, 0, -9.8]
    cov = np.eye(15) * 1e-4
    pts = scans["points"][k]; var_b = rand_var(len(pts), 0.005)
    ctx.lio_state_estimation(pts, var_b, state, cov)
    t0 = time.perf_counter()
    for _ in range(reps):
        ok, st, cv = ctx.lio_state_estimation(pts, var_b, state, cov)
    dt = (time.perf_counter() - t0) / reps
    ctx.close()
    res = {"ms_per_scan": 1e3 * dt, "points": int(len(pts)), "converged": bool(ok), "what": "points + covariances uploaded per call"}
    # the initialisation odometry (lio_state_estimation_kdtree, voxelslam.cpp:1102-1252): exact 5-NN plane fit against the
    # point-cloud map, EKF, map update (append + 0.5 m re-sampling); a short sequence of scans, the first one seeds the map
    ctx = capi.Context(capi.options_from_workload(wl, stream=torch.cuda.current_stream().cuda_stream))
    seq = [p[::10].astype(np.float32).astype(np.float64) for p in scans["points"][:4]]     # 20k-point scans (the CPU port is a brute-force search too)
    def run_kd(step):
        cv = np.eye(15) * 1e-4
        tt, its, tree = 0.0, 0, 0
        for i, p in enumerate(seq):
            s0 = np.zeros(25); s0[1:10] = scans["R_gt"][i].ravel(); s0[10:13] = scans["p_gt"][i] + (0.01 if i else 0.0); s0[22:25] = [0, 0, -9.8]
            t1 = time.perf_counter()
            it, _, cv2 = step(p, s0, cv)
            if i:
                tt += time.perf_counter() - t1; its += it
        return tt / (len(seq) - 1), its
    run_kd(ctx.lio_state_estimation_kdtree); ctx.lib.vba_odom_kdtree_reset(ctx.h)
    t_kd, its_kd = run_kd(ctx.lio_state_estimation_kdtree)
    res["kdtree_variant"] = {"ms_per_scan": 1e3 * t_kd, "points": int(len(seq[0])), "map_points": int(ctx.kdtree_size()), "ekf_iterations": its_kd,
                             "what": "brute-force exact 5-NN on the device + EKF + map re-sampling, scan uploaded per call"}
    ctx.close()
    if cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_api
        ko = oracle_api.KdOdom()
        t_cpu, _ = run_kd(ko.lio_state_estimation)
        res["kdtree_variant"]["cpu_port_ms_per_scan"] = 1e3 * t_cpu     # the oracle searches by brute force too (the reference uses a FLANN kd-tree): not a speed-up claim
    return res


def load_profile(tag_glob="r02"):
    """The committed rocprofv3 summary of this command (tools/prof_summary.py).  It is used ONLY when it was measured on the sources
    this run executes (source hash of bench.py + library sources): a stale profile prices nothing."""
    import glob
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from prof_summary import source_hash
        cur = source_hash()
    except Exception:
        return None, "tools/prof_summary.py missing"
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_k4_profile.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("source_hash") == cur:
            best = (f, d)
    if best is None:
        return None, "no profile under profiles/ matches the running sources (hash %s): rocprofv3 fields are null, the live hipEvent figure prices the line" % cur[:12]
    return best, None


def load_hba_full():
    """BASELINE configs[4] at full length (2000 keyframes x 50k points on one GPU) takes minutes of scan synthesis on the host, so it
    is measured by tools/hba_fullsize.py on the GPU box and committed under profiles/; this reads the newest record."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hba_fullsize.json"))):
        try:
            best = dict(json.load(open(f)), record=os.path.relpath(f, ROOT))
        except Exception:
            pass
    return best


def launcher(args):
    """`python bench.py --gpus N` without a torchrun environment: start the N rank processes (fresh children, spawned before this
    process touches the GPU), one per device, and exit with their status."""
    import glob
    import socket
    import subprocess
    if any(k.startswith("ROCPROFILER_") or k.startswith("ROCP_") for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        sys.stderr.write("bench.py: the multi-rank launcher must not run under rocprofv3 (the preloaded profiler has initialised the GPU; every rank "
                         "spawn would be an exec hop behind it): profile the single-process form, tools/profile_bench.sh\n")
        return 2
    # GPU agents of this node from the KFD topology (sysfs): no HIP / HSA call in the parent before the ranks are spawned
    nd = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(l.split() for l in open(f).read().splitlines() if len(l.split()) == 2)
            nd += 1 if int(props.get("simd_count", "0")) > 0 else 0
        except OSError:
            pass
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
    if vis:
        nd = min(nd, len([v for v in vis.split(",") if v.strip() != ""])) if nd else len([v for v in vis.split(",") if v.strip() != ""])
    if nd < args.gpus:
        sys.stderr.write("bench.py: --gpus %d requested but this node exposes %d device(s); refusing to run fewer ranks than asked\n" % (args.gpus, nd))
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for pr in procs:
        pr.wait()
        rc = rc or pr.returncode
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="hesai200k_w10")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scaled", action="store_true")
    ap.add_argument("--scene-copies", type=int, default=56)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launcher(args))

    import torch
    import torch.distributed as dist
    import voxel_slam_amd  # noqa: F401
    from voxel_slam_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not os.environ.get("VBA_BENCH_ALLOW_MISMATCH"):
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: the line would report a GPU count that was not asked for" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if os.environ.get("VBA_BENCH_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks but only %d device(s) visible" % (world, torch.cuda.device_count()))
    local_rank = local_rank % torch.cuda.device_count()     # (gloo rehearsals put several ranks on one card)
    torch.cuda.set_device(local_rank)
    # VBA_BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL all-reduce per LM iteration) with a single rank, to measure
    # what the exchange step adds on a one-GPU box
    dist_on = world > 1 or os.environ.get("VBA_BENCH_FORCE_DIST", "") not in ("", "0")
    backend = os.environ.get("VBA_BENCH_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm; "gloo" only to rehearse several ranks on one card
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    wl, scans, poses0 = build_problem(args.workload)
    W = wl.win_size

    # everything runs on ONE explicit (non-default) stream: the library launches on it, the torch events of this file record on it,
    # and a HIP graph can be captured from it (the legacy default stream cannot be captured)
    main_stream = torch.cuda.Stream()
    torch.cuda.set_stream(main_stream)
    stream = torch.cuda.current_stream().cuda_stream
    opt = capi.options_from_workload(wl, stream=stream)
    opt.device = local_rank
    if dist_on and world == 1:
        opt.force_collective = 1          # VBA_BENCH_FORCE_DIST: the exchange step with one rank
    ctx = capi.Context(opt)
    if dist_on:
        if backend == "nccl":
            ctx.rccl_init(dist, rank, world)   # the library's own RCCL communicator: ncclAllReduce on the context's stream, no Python in the loop
        else:
            ctx.set_shard(rank, world)         # K1 keeps only the points whose root voxel falls in this rank's bucket range
            ctx.set_torch_allreduce(torch, dist)

    # full-window rebuild on the device (voxelslam.cpp:664-703): K1 insert of W scans, K2 recut + factor extraction.
    # The scans are uploaded to HBM once; the timed rebuild passes consume device pointers.
    dev_scans = [torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in scans["points"]]
    import ctypes as C

    def rebuild():
        ctx.map_reset()
        for i in range(W):
            st = ctx.lib.vba_map_cut_voxel(ctx.h, C.c_int(i), C.c_int(dev_scans[i].shape[0]), C.c_void_p(dev_scans[i].data_ptr()), None,
                                           poses0[i].ctypes.data_as(C.POINTER(C.c_double)), C.c_int(0))
            ctx._chk(st)
        ctx.recut(W, poses0, multi=False)

    rebuild()
    ctx.timing_enable(True)
    ctx.timing_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_rebuild = 3
    for _ in range(n_rebuild):
        rebuild()
    torch.cuda.synchronize()
    t_rebuild = (time.perf_counter() - t0) / n_rebuild
    t_ins, n_ins = ctx.timing_get("insert")
    t_rec, n_rec = ctx.timing_get("recut")
    ctx.timing_enable(False)
    V_local = ctx.size()
    vt = torch.tensor([V_local], device="cuda", dtype=torch.int64)
    if dist_on:
        dist.all_reduce(vt)
    V_total = int(vt.item())
    n_points = sum(len(p) for p in scans["points"])

    def run_steps(k, cx=None):
        cx = cx or ctx
        done = 0
        while done < k:
            cx.lm_begin(poses0, thd_num=2)
            cx.lm_refresh_eigen()                    # re-create the eigen state for the restart (not counted as a step)
            for _ in range(min(3, k - done)):
                cx.lm_iterate(sync=False)
                done += 1
            cx.lm_end(fetch=False)                   # the whole timed region is enqueued without host synchronisation

    # clock conditioning: the timed region of a short run (the driver's --steps 20 lasts ~1 ms) would otherwise be measured on a chip
    # that is still ramping its clocks; 240 untimed steps (~12 ms) come first, whatever --warmup says
    run_steps(240)
    run_steps(args.warmup)
    # timed region: only the residual pass K4 (the kernel whose roofline is reported) is bracketed by hipEvents — every
    # event pair costs host time per launch; the other kernels are timed in a second, untimed pass below
    ctx.timing_enable(True)
    ctx.timing_select("residual")
    ctx.timing_sample_every(K4_SAMPLE_EVERY)
    ctx.timing_reset()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    t_enqueue = time.perf_counter() - t0          # host time to enqueue the timed region (the device may still be running)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    t_res, n_res = ctx.timing_get("residual")
    ctx.timing_sample_every(1)
    null_us = ctx.timing_null_spans(64)      # what an event pair costs when it brackets nothing (subtracted below)
    # second pass (not part of `value`): events around every kernel family; the trace tells how many steps were rejected ones
    ctx.timing_select(None)
    ctx.timing_reset()
    run_steps(min(args.steps, 90))
    torch.cuda.synchronize()
    ctx.lm_begin(poses0, thd_num=2); ctx.lm_refresh_eigen()
    for _ in range(3):
        ctx.lm_iterate(sync=False)
    ctx.lm_end(fetch=True)
    tr = ctx.last_trace()
    rejected_per_call = int((tr[:, 1] >= tr[:, 0]).sum()) if len(tr) else 0
    # one full Hessian pass on its own (inside the loop a third of the launches are gated off, which would blur its duration)
    t_hes_loop, n_hes_loop = ctx.timing_get("hessian")
    t_sol, n_sol = ctx.timing_get("solve")
    t_red, n_red = ctx.timing_get("reduce")
    # (timed like K4 below: a batch of launches replayed from a HIP graph between one event pair — a per-launch span minus an empty span
    #  under-reported it by ~20 %)
    ctx.timing_select(None)
    ctx.timing_enable(False)             # no hipEvent spans inside the captured batches below: the replayed graphs hold kernels only
    ctx.evaluate_only_residual(poses0)
    ctx.lm_begin(poses0, thd_num=2)
    for _ in range(3):
        ctx.timing_launch_hessian()
    torch.cuda.synchronize()
    K3_BATCH = 40
    k3_how = "hipGraph replay"
    g3 = None
    try:
        g3 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g3, stream=main_stream, capture_error_mode="thread_local"):
            for _ in range(K3_BATCH):
                ctx.timing_launch_hessian()
    except Exception as e:      # noqa: BLE001
        g3 = None
        k3_how = "direct enqueue (graph capture refused: %s)" % str(e)[:80]
    e3a, e3b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    k3_batches = []
    for _ in range(4):
        torch.cuda.synchronize()
        e3a.record()
        if g3 is not None:
            g3.replay()
        else:
            for _ in range(K3_BATCH):
                ctx.timing_launch_hessian()
        e3b.record()
        torch.cuda.synchronize()
        k3_batches.append(1e3 * e3a.elapsed_time(e3b) / K3_BATCH)
    ctx.lm_end(fetch=False)
    k3_full_us = float(np.mean(k3_batches[1:]))

    # per-kernel device time from hipEvents recorded on the launch stream
    t_hes, n_hes = t_hes_loop, n_hes_loop
    V_local = ctx.size()
    occ = ctx.factor_occupancy()     # occupied (voxel, frame) slots per voxel
    # algorithmic bytes: residual pass reads (W_occ+1)*80 + 8, writes 176 per voxel; the SoA layout additionally
    # reads a 4-byte occupancy mask per voxel to find the occupied slots: reported separately as layout overhead.
    # Hessian pass reads W_occ*80 + 80 + 96 + 8 per voxel (fixed cluster row is not read: it only needs pcr_add's N, v).
    bytes_res = alg_bytes_residual(V_local, occ)
    bytes_res_layout = V_local * 4            # the 4-byte occupancy mask per voxel that tells the pass which slots to read
    bytes_hes = V_local * (occ * 80 + 80 + 96 + 8)
    flops_hes = V_local * (occ * 600 + 2 * 3 * (6 * W) * (6 * W + 1) / 2.0)      # slot preparation + the G^T C G contraction
    res_us_raw = t_res / max(n_res, 1)           # in-loop spans: upper bound, a span carries the processing of its closing event
    # K4 launch duration: K4_BATCH consecutive launches on the launch stream between ONE hipEvent pair (an event pair per launch costs
    # more stream time than this kernel runs, and how much of it hides under the neighbouring kernels depends on the neighbours:
    # "span minus empty span" gave 2.5-4.5 us for a kernel rocprofv3 times at 6.4 us)
    ctx.lm_begin(poses0, thd_num=2)
    for _ in range(5):
        ctx.lm_refresh_eigen()
    torch.cuda.synchronize()
    # the launches are replayed from a HIP graph so that the host's enqueue rate (several microseconds per launch through
    # ctypes + hipLaunchKernel) cannot be what is measured; if the capture is refused the batch is enqueued directly
    batch_how = "hipGraph replay"
    graph = None
    try:
        graph = torch.cuda.CUDAGraph()
        # (thread-local capture mode: RCCL's helper threads of a multi-rank run may touch the runtime while this thread captures)
        with torch.cuda.graph(graph, stream=main_stream, capture_error_mode="thread_local"):
            for _ in range(K4_BATCH):
                ctx.lm_refresh_eigen()
    except Exception as e:      # noqa: BLE001
        graph = None
        batch_how = "direct enqueue (graph capture refused: %s)" % str(e)[:80]
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    batches = []
    for _ in range(6):
        torch.cuda.synchronize()
        ev_a.record()
        if graph is not None:
            graph.replay()
        else:
            for _ in range(K4_BATCH):
                ctx.lm_refresh_eigen()
        ev_b.record()
        torch.cuda.synchronize()
        batches.append(1e3 * ev_a.elapsed_time(ev_b) / K4_BATCH)
    ctx.lm_end(fetch=False)
    res_us = float(np.mean(batches[1:]))            # mean of 5 batches after one warm-up batch
    # one ACCEPTED iteration as SURVEY.md 8(d) defines it (K3 + partial sum + solve + K4 + accept/reject bookkeeping): calls of
    # [lm_begin, eigen refresh (one K4), ONE lm_iterate, lm_end] from the perturbed start poses — that first step is always accepted
    # on this workload — enqueued back to back between one event pair; the refresh pass (res_us) is subtracted.  The call's 4 KB
    # state upload stays inside the figure.
    FI = 60
    def accepted_calls(k):
        for _ in range(k):
            ctx.lm_begin(poses0, thd_num=2); ctx.lm_refresh_eigen(); ctx.lm_iterate(sync=False); ctx.lm_end(fetch=False)
    accepted_calls(10)
    fi_batches = []
    for _ in range(4):
        torch.cuda.synchronize()
        ev_a.record(); accepted_calls(FI); ev_b.record()
        torch.cuda.synchronize()
        fi_batches.append(1e3 * ev_a.elapsed_time(ev_b) / FI)
    ctx.lm_begin(poses0, thd_num=2); ctx.lm_refresh_eigen(); ctx.lm_iterate(sync=False); ctx.lm_end(fetch=True)
    tr1 = ctx.last_trace()
    fi_accepted = bool(len(tr1) and tr1[0, 1] < tr1[0, 0])
    full_iter_us = float(np.mean(fi_batches[1:])) - res_us
    hes_us = max(t_hes / max(n_hes, 1) - null_us, 1e-3)
    sol_us = max(t_sol / max(n_sol, 1) - null_us, 1e-3)
    red_us = max(t_red / max(n_red, 1) - null_us, 1e-3)
    per_iter = {"hessian": t_hes / max(n_sol, 1), "reduce": t_red / max(n_sol, 1), "solve": t_sol / max(n_sol, 1), "residual": res_us}
    dominant = max(per_iter, key=per_iter.get)
    res_gbs = bytes_res / (res_us * 1e-6) / 1e9

    # rocprofv3 evidence of this same command, when (and only when) it was taken on the running sources
    prof, prof_note = load_profile()
    rocprof_us = traffic = None
    prof_file = None
    if prof is not None and args.workload == "hesai200k_w10" and world == 1:
        prof_file, pd = prof
        k4 = [k for k in pd["kernels"] if k["name"].startswith("vba::k_residual_s<%d" % W)]
        if k4:
            main_grid = max(k4, key=lambda k: k["launches"])           # the grid the LM loop launches (the side legs use other sizes)
            rocprof_us = main_grid.get("avg_ns_without_outliers", main_grid["avg_ns"]) / 1000.0   # (the 12 cache-flushed launches of the `cold` leg share this grid)
            fetch_factor = pd.get("calibration", {}).get("fetch_bytes_per_counted_byte")
            if "FETCH_SIZE_KB_median" in main_grid and "WRITE_SIZE_KB_median" in main_grid and fetch_factor:
                # gfx950: FETCH_SIZE under-reports reads by an access-shape dependent factor (x2 for 16-B-per-lane streams,
                # MI355X_MICROARCH.md "HBM"); for this library's 8-B-per-lane SoA reads the factor is CALIBRATED in the same PMC pass
                # on a launch of known size (k_calib_read8).  The bytes counted are fabric requests: Infinity-Cache hits included.
                traffic = (fetch_factor * main_grid["FETCH_SIZE_KB_median"] + main_grid["WRITE_SIZE_KB_median"]) * 1024.0
    roof = {"bound": "hbm", "kernel": "k_residual_s<%d> (K4, evaluate_only_residual)" % W,
            "achieved": res_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": res_gbs / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_note": "fabric bytes of one launch from the committed rocprofv3 PMC passes (FETCH_SIZE x calibrated factor + WRITE_SIZE); null unless "
                            "profiles/ holds a profile taken on the running sources",
            "duration_basis": "one hipEvent pair on the launch stream around %d consecutive K4 launches (the restart pass of the LM loop, same grid and "
                              "store; %s), divided by %d, mean of 5 batches after a warm-up batch; measured live right after the timed region" % (K4_BATCH, batch_how, K4_BATCH),
            "avg_launch_us": res_us, "launches": 5 * K4_BATCH, "batch_us": batches,
            "in_loop_span_us": res_us_raw, "in_loop_spans": n_res, "empty_span_us": null_us,
            "in_loop_note": "hipEvent pairs around every %d-th K4 launch INSIDE the timed region: an upper bound (the span includes the closing event's "
                            "own processing; an event pair around nothing costs empty_span_us)" % K4_SAMPLE_EVERY,
            "algorithmic_bytes_per_launch": bytes_res, "algorithmic_bytes_per_voxel": bytes_res / max(V_local, 1),
            "layout_extra_bytes_per_launch": bytes_res_layout,
            "rocprofv3_avg_launch_us": rocprof_us,
            "frac_at_rocprofv3_duration": (bytes_res / (rocprof_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if rocprof_us else None,
            "rocprofv3_profile": os.path.relpath(prof_file, ROOT) if prof_file else None, "rocprofv3_note": prof_note,
            "note": "V = %d voxels = %.1f MB per pass: the working set sits in the 256 MiB Infinity Cache between launches and one launch lasts a few "
                    "microseconds, so this line is latency- rather than bandwidth-limited; roofline_residual_pass_big_scene is the same kernel on a "
                    "pass that cannot be cached" % (V_local, bytes_res / 1e6),
            "other_kernels": {
                "k_hessian2<%d> (K3, acc_evaluate2)" % W: {"avg_launch_us": hes_us, "launches": n_hes, "note": "average incl. the launches gated off after a rejected step",
                                                           "algorithmic_GBps": bytes_hes / (hes_us * 1e-6) / 1e9, "algorithmic_bytes_per_launch": bytes_hes,
                                                           "flops_per_full_pass": flops_hes, "full_pass_us": k3_full_us, "full_pass_basis": "%d consecutive full passes (%s) between one hipEvent pair, mean of 3 batches" % (K3_BATCH, k3_how),
                                                           "fraction_of_fp64_peak": flops_hes / (k3_full_us * 1e-6) / 1e12 / FP64_PEAK_TFLOPS},
                "k_reduce_partials": {"avg_launch_us": red_us, "launches": n_red},
                "k_lm_solve_m (gauge + LDLT + retraction, one workgroup)": {"avg_launch_us": sol_us, "launches": n_sol}},
            "per_iteration_us": per_iter, "dominant_by_time": dominant}

    # K4 on the same scene tiled so that one pass touches more than twice the Infinity Cache
    scaled = None
    if world == 1 and not args.no_scaled:
        scaled = scaled_residual_pass(capi, torch, wl, scans, poses0, copies=args.scene_copies)
        if prof is not None:
            k4 = [k for k in prof[1]["kernels"] if k["name"].startswith("vba::k_residual_v<%d" % W)]   # the large-store kernel (one lane per voxel)
            if k4:
                big = max(k4, key=lambda k: k["grid_threads"])
                if big["grid_threads"] >= scaled["voxels"]:
                    scaled["rocprofv3_avg_launch_us"] = big["avg_ns"] / 1000.0
                    ff = prof[1].get("calibration", {}).get("fetch_bytes_per_counted_byte")
                    if "FETCH_SIZE_KB_median" in big and "WRITE_SIZE_KB_median" in big and ff:
                        scaled["traffic"] = (ff * big["FETCH_SIZE_KB_median"] + big["WRITE_SIZE_KB_median"]) * 1024.0

    cold = lms = liv = hba = odo = None
    if world == 1 and not args.no_scaled:
        cold = cold_residual_pass(ctx, torch, poses0, V_local, occ, W)
        liv = li_variant(ctx, capi, wl, scans, poses0)
        lms = local_mapping_step(capi, torch, wl)
        hba = hba_window(capi, torch, cpu=not args.no_cpu_baseline)
        odo = odometry_update(capi, torch, wl, scans, cpu=not args.no_cpu_baseline)
    roof["cold"] = cold
    if world == 1:
        ctx.timing_calibration_read(1 << 30)     # one launch of known size (1 GiB): calibrates FETCH_SIZE in the rocprofv3 PMC pass

    # the same loop with the damping candidates switched off (vba_options::lm_spec = 1): every rejected step then runs its own
    # solve, as the reference's loop does.  Traces are identical bit for bit (tests/test_gpu_spec.py).
    seq = None
    if world == 1:
        opt.lm_spec = 1
        ctx2 = capi.Context(opt)
        opt.lm_spec = 0
        for i in range(W):
            ctx2._chk(ctx2.lib.vba_map_cut_voxel(ctx2.h, C.c_int(i), C.c_int(dev_scans[i].shape[0]), C.c_void_p(dev_scans[i].data_ptr()), None,
                                                 poses0[i].ctypes.data_as(C.POINTER(C.c_double)), C.c_int(0)))
        ctx2.recut(W, poses0, multi=False)
        run_steps(args.warmup, ctx2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(args.steps, ctx2)
        torch.cuda.synchronize()
        seq = {"value": args.steps / (time.perf_counter() - t0), "unit": "iterations/s",
               "what": "the timed loop with one damping value per solve launch (vba_options::lm_spec = 1): a rejected step runs its own solve"}
        ctx2.close()

    if rank == 0:
        out = {
            "metric": "local-BA iterations/sec (200k pts, W=10)", "value": args.steps / dt, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %s synthetic scans, %d pts/scan, W=%d, voxel %.2f m; %d planar voxels "
                                   "(%.1f occupied frames/voxel), lidar-only LM (Lidar_BA_Optimizer)"
                                   % (wl.name, "Hesai-32-like" if wl.pattern == "spin32" else "Livox-Avia-like", wl.n_pts, W, wl.voxel_size, V_total, occ),
                       "parallelism": ("voxel-bucket shard x%d + RCCL all-reduce of [H|g|r] inside the library" % world) if (dist_on and backend == "nccl")
                                      else ("voxel-bucket shard x%d + all-reduce hook (%s rehearsal)" % (world, backend)) if dist_on else "single GPU",
                       "steps_note": "a step is one trip through the LM loop body VM:441-494; %d of the 3 steps of every damping_iter call are rejected "
                                     "steps, which skip the Hessian pass exactly as VM:443 does (the CPU baseline runs the same sequence); every solve launch "
                                     "also solves the damping values of the next 3 rejections on otherwise idle CUs, and a rejected step whose damping was "
                                     "among them installs that solution instead of solving again (same trace bit for bit; `sequential_damping` = the loop "
                                     "without it)" % rejected_per_call},
            "host_enqueue_ms_per_step": 1e3 * t_enqueue / args.steps,
            "sequential_damping": seq,
            "roofline": roof,
            "roofline_residual_pass_big_scene": scaled,
            "local_mapping_step": lms,
            "li_ba_variant": liv,
            "hba_window": hba,
            "odometry_update": odo,
            "full_window_rebuild": {"points": n_points, "wall_ms": 1e3 * t_rebuild, "insert_device_ms": 1e-3 * t_ins / max(n_rebuild, 1),
                                    "recut_extract_device_ms": 1e-3 * t_rec / max(n_rebuild, 1),
                                    "insert_algorithmic_GBps": n_points * 24 / (t_ins / max(n_rebuild, 1) * 1e-6) / 1e9 if t_ins > 0 else None},
            "k3_fraction_of_fp64_peak": flops_hes / (k3_full_us * 1e-6) / 1e12 / FP64_PEAK_TFLOPS,
            "full_iteration": {"us": full_iter_us, "iterations_per_s": 1e6 / full_iter_us, "step_was_accepted": fi_accepted,
                               "what": "one ACCEPTED LM iteration = K3 (Hessian pass) + partial sum + solve + K4 (residual pass) + accept/reject "
                                       "bookkeeping, SURVEY.md 8(d); %d calls of [lm_begin, eigen refresh, one lm_iterate, lm_end] enqueued back to back "
                                       "between one hipEvent pair, the refresh pass (roofline.avg_launch_us) subtracted; `value` counts the loop the "
                                       "reference runs, in which 2 of 3 steps are rejected and skip K3" % FI,
                               "call_us": float(np.mean(fi_batches[1:]))},
            "hba_full_length": load_hba_full(),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(wl, scans, poses0)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            if liv:
                out["li_ba_gpu_over_cpu"] = liv["iterations_per_s"] / out["cpu_baseline"]["li_ba"]["value"]
        print(json.dumps(out))
        sys.stdout.flush()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
