"""Summarises the rocprofv3 runs of ONE bench command into profiles/<tag>_k4_profile.json, which bench.py reads back:

    python tools/prof_summary.py <tag> <kernel_trace_dir> [<pmc_fetch_dir> <pmc_write_dir>]

  * <kernel_trace_dir>: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py ...
  * <pmc_*_dir>       : rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, MI355X_MICROARCH.md "HBM")

Per kernel NAME and GRID SIZE (the residual pass is launched at the bench size inside the LM loop and at other sizes by the side
legs of bench.py: only equal grids are averaged together): launches, average / median duration, median FETCH_SIZE / WRITE_SIZE.
The file carries the hash of the sources the numbers were measured on (source_hash()); bench.py prices nothing from a profile
whose hash differs from the tree it runs in.
"""
import csv
import glob
import hashlib
import json
import os
import statistics
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash():
    """sha256 over bench.py, the C ABI header and every source of the library (the things a kernel's duration depends on)."""
    h = hashlib.sha256()
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "include", "voxelba.h")]
    files += sorted(glob.glob(os.path.join(ROOT, "voxel-slam_amd", "csrc", "*.h*"))) + [os.path.join(ROOT, "voxel-slam_amd", "csrc", "Makefile")]
    files += [os.path.join(ROOT, "voxel-slam_amd", "capi.py"), os.path.join(ROOT, "voxel-slam_amd", "synth.py")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def short(name):
    return name.split("(")[0].replace("void ", "")


def main():
    tag, trace_dir = sys.argv[1], sys.argv[2]
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            dur[(short(r["Kernel_Name"]), grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    pmc = {}
    for which, idx in (("FETCH_SIZE", 3), ("WRITE_SIZE", 4)):
        if len(sys.argv) <= idx:
            continue
        per = defaultdict(list)
        for f in glob.glob(os.path.join(sys.argv[idx], "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == which:
                    per[(short(r["Kernel_Name"]), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
        pmc[which] = per
    out = {"tag": tag, "source_hash": source_hash(), "kernels": []}
    for (name, grid), v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        if not name.startswith("vba::"):
            continue
        med = statistics.median(v)
        core = [x for x in v if x <= 3 * med]          # without the cache-flushed launches of bench.py's `roofline.cold` leg (same grid, ~30x longer)
        rec = {"name": name, "grid_threads": grid, "launches": len(v), "avg_ns": sum(v) / len(v), "median_ns": med,
               "avg_ns_without_outliers": sum(core) / len(core), "outliers_gt_3x_median": len(v) - len(core),
               "min_ns": min(v), "max_ns": max(v), "total_ns": sum(v)}
        for which in pmc:
            vals = pmc[which].get((name, grid))
            if vals:
                rec[which + "_KB_median"] = statistics.median(vals)
                rec[which + "_launches"] = len(vals)
        out["kernels"].append(rec)
    # read-side calibration: k_calib_read8 reads exactly grid_threads * 128 bytes (vba_timing_calibration_read)
    for k in out["kernels"]:
        if k["name"].startswith("vba::k_calib_read8") and k.get("FETCH_SIZE_KB_median"):
            out["calibration"] = {"known_read_bytes": k["grid_threads"] * 128, "FETCH_SIZE_KB": k["FETCH_SIZE_KB_median"],
                                  "fetch_bytes_per_counted_byte": k["grid_threads"] * 128 / (k["FETCH_SIZE_KB_median"] * 1024.0)}
    path = os.path.join(ROOT, "profiles", "%s_k4_profile.json" % tag)
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, "with", len(out["kernels"]), "(kernel, grid) groups; source hash", out["source_hash"][:12])


if __name__ == "__main__":
    main()
