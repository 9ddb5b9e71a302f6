"""Timeline of the last lidar_only local-mapping step of a rocprofv3 kernel-trace CSV (tools/step_probe.py): start offset, duration and
the idle gap before every kernel — shows where the step waits on the host (syncs, launches) rather than on kernels."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ml = [i for i, r in enumerate(rows) if "k_margi_leaf" in r["Kernel_Name"]]
i0, i1 = ml[-2], ml[-1]
t0 = int(rows[i0]["Start_Timestamp"]); prev_end = t0
gaps = 0.0
for r in rows[i0:i1]:
    n = r["Kernel_Name"].replace("void ", "").replace("vba::", "").split("(")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3
    gaps += max(gap, 0.0)
    print("%9.1f us  +%6.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, n[:60]))
    prev_end = max(prev_end, e)
print("idle gaps %.1f us of %.1f us" % (gaps, (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
