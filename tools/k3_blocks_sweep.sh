#!/bin/bash
# sweep of the persistent-workgroup count of the Hessian pass (vba_options::hessian_workgroups, through the binding's VBA_PY_OPTIONS hook)
for nb in 64 128 192 256; do
  VBA_PY_OPTIONS=hessian_workgroups=$nb timeout -k 10 200 python bench.py --no-cpu-baseline --no-scaled 2>/dev/null > /tmp/k3_$nb.json
  python - <<PY
import json
d=json.loads(open("/tmp/k3_$nb.json").read().strip().splitlines()[-1])
print("workgroups", $nb, round(d["value"],1), "it/s", round(1e3*d["ms_per_step"],2), "us; K3", round(list(d["roofline"]["other_kernels"].values())[0]["avg_launch_us"],2))
PY
done
