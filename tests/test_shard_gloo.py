"""Multi-rank path on CPU (gloo, world_size 2): voxels are sharded by vba_shard_owner (root-voxel hash bucket ranges),
each rank evaluates only its shard and the packed [H | g | r] buffer is summed with an all-reduce — the replacement of
the reference's serial thread-sum (voxel_map.hpp:571-581).  Without a GPU the per-shard evaluation is done by the CPU
oracle; what is under test is the product's partition function and the reduction plumbing."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import voxel_slam_amd  # noqa: F401
    from voxel_slam_amd import synth, capi
    import oracle_api
    wl = synth.CONFIGS["room20k_w4"]
    s = synth.make_scans(wl)
    fac = synth.root_factors(s["points"], s["R0"], s["p0"], wl, with_keys=True)
    poses = synth.poses_flat(s["R0"], s["p0"])
    own = np.array([capi.shard_owner(k, world) for k in fac["keys"]])
    mine = {k: v[own == rank] for k, v in fac.items()}
    f = oracle_api.Factor(wl.win_size)
    f.push_dict(mine)
    H, g, r = f.acc_evaluate2(poses)
    n = 6 * wl.win_size
    buf = torch.from_numpy(np.concatenate([H.ravel(), g, [r]]))          # the packed [H | g | r] message
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    cnt = torch.tensor([len(mine["coe"])]); dist.all_reduce(cnt)
    res = torch.tensor([f.evaluate_only_residual(poses)], dtype=torch.float64); dist.all_reduce(res)
    if rank == 0:
        full = oracle_api.Factor(wl.win_size); full.push_dict(fac)
        H0, g0, r0 = full.acc_evaluate2(poses)
        b = buf.numpy()
        ok = (int(cnt.item()) == len(fac["coe"]) and np.allclose(b[:n * n].reshape(n, n), H0, rtol=1e-11, atol=1e-9)
              and np.allclose(b[n * n:n * n + n], g0, rtol=1e-11, atol=1e-12) and abs(b[-1] - r0) < 1e-12
              and abs(res.item() - full.evaluate_only_residual(poses)) < 1e-12)
        open(os.path.join(out_dir, "ok" if ok else "fail"), "w").write("%d %d" % (int(cnt.item()), len(fac["coe"])))
    dist.destroy_process_group()


def test_two_rank_shard_and_allreduce(tmp_path, oracle):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok"), list(tmp_path.iterdir())
