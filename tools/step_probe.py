"""One local-mapping step (bench.local_mapping_step: pvec_update + cut_voxel_multi, multi_recut, LI-BA, multi_margi) under
rocprofv3 --kernel-trace; tools/step_trace_summary.py adds up the kernels of the LAST step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import voxel_slam_amd  # noqa
from voxel_slam_amd import capi, synth
import bench
torch.cuda.set_stream(torch.cuda.Stream())
out = bench.local_mapping_step(capi, torch, synth.CONFIGS["hesai200k_w10"], steps=int(sys.argv[1]) if len(sys.argv) > 1 else 3)
print({k: v for k, v in out.items() if k != "what"})
