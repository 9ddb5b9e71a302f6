"""The large-store form of the residual pass (beyond 45 000 voxels the voxel-per-lane kernel k_residual_v replaces the slot-parallel
k_residual_s): the same factors pushed several times over must give that multiple of the residual and, voxel for voxel, the
eigen-pairs and BIT-IDENTICAL sums of the small-store kernel."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_large_store_residual_pass_equals_small_store_kernel(oracle):
    import voxel_slam_amd  # noqa: F401
    from voxel_slam_amd import capi, synth
    wl = synth.CONFIGS["hesai200k_w10"]
    s = synth.make_scans(wl)
    fac = synth.root_factors(s["points"], s["R0"], s["p0"], wl)
    V = len(fac["coe"])
    poses = synth.poses_flat(s["R0"], s["p0"])
    o = capi.default_options(); o.win_size = wl.win_size
    small = capi.Context(o); small.push_dict(fac)
    r1 = small.evaluate_only_residual(poses)
    ev1, evec1, pa1 = small.read_back()
    reps = (1 << 16) // V + 2
    big = capi.Context(o)
    for _ in range(reps):
        big.push_dict(fac)
    assert big.size() == reps * V > (1 << 16)
    rb = big.evaluate_only_residual(poses)
    assert abs(rb - reps * r1) < 1e-11 * abs(rb)
    ev, evec, pa = big.read_back()
    for k in (0, reps // 2, reps - 1):
        sl = slice(k * V, (k + 1) * V)
        assert np.array_equal(pa[sl], pa1)                       # same sums, same order of additions
        assert np.abs(ev[sl] - ev1).max() < 1e-13 * max(1.0, np.abs(pa1[:, :6]).max() / 5)
        n1 = evec[sl].reshape(V, 3, 3)[:, :, 0]; n2 = evec1.reshape(V, 3, 3)[:, :, 0]
        assert np.abs(np.abs((n1 * n2).sum(1)) - 1).max() < 1e-12
    # sub-ranges still work on the large store (acc_evaluate2-style head / end arguments)
    f = oracle.Factor(wl.win_size); f.push_dict(fac)
    ra = big.evaluate_only_residual(poses, 5, V - 3); rc = f.evaluate_only_residual(poses, 5, V - 3)
    assert abs(ra - rc) < 1e-9 * abs(rc)      # (18k eigenvalues of ~1e-4, each good to eps x second moments ~1e-13: random-walk sum)


def test_voxel_per_lane_kernel_for_every_window_size():
    """k_residual_v<W> normally runs only beyond 45 000 voxels; vba_options::residual_vpl_from = 1 (injected into every options struct of a child
    pytest through the ctypes binding's VBA_PY_OPTIONS hook) forces it on the small stores of the LM parity tests, for every window size 2..16 and for stores that are NOT in
    occupancy-mask order (pushed by the host)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VBA_PY_OPTIONS="residual_vpl_from=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(root, "tests", "test_gpu_factor.py"), "-k", "all_window_sizes or lidar_ba_damping_iter_parity or li_ba_damping_iter_parity"],
                       capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-1000:]
