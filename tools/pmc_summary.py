"""Summarises two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes) into
per-kernel p75 KB per launch:  python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, sys
from collections import defaultdict


def load(d, counter):
    per = defaultdict(list)
    for f in glob.glob(d + "/*counter_collection.csv"):          # (top level only: older runs may sit in sub-directories)
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return per


def p75(v):
    v = sorted(v)
    return v[min(len(v) - 1, int(0.75 * len(v)))]


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    out[k] = {"launches": len(fetch.get(k, write.get(k, []))), "FETCH_SIZE_KB_p75": p75(fetch[k]) if k in fetch else None,
              "WRITE_SIZE_KB_p75": p75(write[k]) if k in write else None}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("kernels:", len(out))
