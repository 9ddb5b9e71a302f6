"""Prints the kernel sequence of the LAST scan insertion (+ recut, margi) found in a rocprofv3 kernel-trace CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
seq = [i for i, r in enumerate(rows) if "k_scan_to_soa" in r["Kernel_Name"]]
for idx in ([seq[len(seq) // 2 - 1], seq[-1]] if which < 0 else [seq[which]]):
    i0 = idx
    while i0 > 0 and "k_margi_leaf" not in rows[i0]["Kernel_Name"] and idx - i0 < 14: i0 -= 1
    t0 = int(rows[i0]["Start_Timestamp"])
    tot = 0.0
    print("---- scan at trace row", idx)
    for r in rows[i0:i0 + 60]:
        n = r["Kernel_Name"].replace("void ", "").replace("vba::", "")
        if "rocprim" in n: n = "rocprim::" + n.split("wrapped_")[1].split("<")[0] if "wrapped_" in n else "rocprim"
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print("%-58s grid %8s  dur %7.1f us  start +%8.1f" % (n[:58], r["Grid_Size_X"], d, (int(r["Start_Timestamp"]) - t0) / 1e3))
        if "k_extract_write" in n: break
