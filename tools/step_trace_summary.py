"""Adds up the kernel time of the last complete local-mapping step (from one k_margi_leaf to the next) of the FIRST context (li_ba) and
of the last context (lidar_only) in a rocprofv3 kernel-trace CSV."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ml = [i for i, r in enumerate(rows) if "k_margi_leaf" in r["Kernel_Name"]]
def summarize(i0, i1, tag):
    agg = collections.OrderedDict()
    t0, t1 = int(rows[i0]["Start_Timestamp"]), int(rows[i1]["Start_Timestamp"])
    for r in rows[i0:i1]:
        n = r["Kernel_Name"].replace("void ", "").replace("vba::", "").split("(")[0]
        if "rocprim" in n: n = "rocprim"
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += d
    busy = sum(v[1] for v in agg.values())
    print("---- %s: step wall %.1f us (start to start), kernels busy %.1f us, %d launches" % (tag, (t1 - t0) / 1e3, busy, i1 - i0))
    for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("  %-44s x%-3d %8.1f us" % (n[:44], c, d))
half = [i for i in range(len(ml) - 1) if int(rows[ml[i + 1]]["Start_Timestamp"]) - int(rows[ml[i]]["Start_Timestamp"]) > 50e6]   # gap between the two contexts
split = half[0] if half else len(ml) // 2
summarize(ml[split - 1], ml[split], "li_ba step")
summarize(ml[-2], ml[-1], "lidar_only step")
