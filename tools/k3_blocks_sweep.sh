#!/bin/bash
# sweep of the persistent-workgroup count of the Hessian pass (VBA_K3_BLOCKS)
for nb in 256 384 512 768; do
  VBA_K3_BLOCKS=$nb timeout -k 10 200 python bench.py --no-cpu-baseline --no-scaled 2>/dev/null > /tmp/k3_$nb.json
  python - <<PY
import json
d=json.load(open("/tmp/k3_$nb.json"))
print("blocks", $nb, round(d["value"],1), "it/s", round(1e3*d["ms_per_step"],2), "us; K3", round(list(d["roofline"]["other_kernels"].values())[0]["avg_launch_us"],2))
PY
done
