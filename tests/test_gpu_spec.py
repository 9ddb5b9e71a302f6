"""Speculative damping (k_lm_solve_m / k_li_solve solve the damping values of the next rejections on idle CUs; a rejected step then
installs a parked candidate instead of solving again): the result must be BIT-IDENTICAL to the plain sequential loop
(vba_options::lm_spec = 1), including runs of rejections longer than the number of candidates.  The Avia workload (a +-35 degree cone that
mostly sees one wall: lidar-only BA is ill-posed) produces such runs: 3 and 6 consecutive rejections within 14 iterations."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import voxel_slam_amd  # noqa: F401
    from voxel_slam_amd import capi, synth
    return capi, synth


def _ctx(capi, wl, spec):
    o = capi.options_from_workload(wl)
    if spec is not None:
        o.lm_spec = spec                                          # vba_options::lm_spec (1 = the plain sequential solve)
    return capi.Context(o)


def _longest_reject_run(trace):
    best = run = 0
    for r in trace:
        run = run + 1 if r[1] >= r[0] else 0
        best = max(best, run)
    return best


def test_lidar_lm_speculative_equals_sequential(mods, oracle):
    capi, synth = mods
    wl = synth.CONFIGS["avia100k_w10"]
    s = synth.make_scans(wl)
    fac = synth.root_factors(s["points"], s["R0"], s["p0"], wl)
    poses = synth.poses_flat(s["R0"], s["p0"])
    out = {}
    for spec in (None, 1, 2):
        ctx = _ctx(capi, wl, spec); ctx.push_dict(fac)
        out[spec] = ctx.lidar_ba_damping_iter(poses, max_iter=14, thd_num=2)
        ev = ctx.read_back()
        out[spec]["eig"] = ev[0]
        ctx.close()
    a, b, c2 = out[None], out[1], out[2]
    assert _longest_reject_run(b["trace"]) >= 5          # longer than LM_SPEC - 1 = 3 parked candidates: the refill path runs too
    for o in (a, c2):
        assert o["trace"].shape == b["trace"].shape
        assert np.array_equal(o["trace"], b["trace"]), np.abs(o["trace"] - b["trace"]).max()
        assert np.array_equal(o["poses"], b["poses"])
        assert np.array_equal(o["eig"], b["eig"])        # the residual passes ran at the same trial poses
    # and the sequence is the reference's: same accept / reject pattern and damping schedule as the CPU oracle
    f = oracle.Factor(wl.win_size); f.push_dict(fac)
    r = f.lidar_ba_damping_iter(poses, max_iter=14, thd_num=2, parallel=False)
    n = min(len(r["trace"]), len(a["trace"]))
    acc_g = a["trace"][:n, 1] < a["trace"][:n, 0]; acc_o = r["trace"][:n, 1] < r["trace"][:n, 0]
    k = int(np.argmax(acc_g != acc_o)) if (acc_g != acc_o).any() else n     # (an ill-posed solve may flip a marginal decision late)
    assert k >= 6, (k, a["trace"][:n], r["trace"][:n])
    assert np.allclose(a["trace"][:k, 2:4], r["trace"][:k, 2:4], rtol=1e-6)


@pytest.mark.parametrize("gravity", [False, True])
def test_li_ba_speculative_equals_sequential(mods, gravity):
    capi, synth = mods
    wl = synth.CONFIGS["avia100k_w10"]
    s = synth.make_scans(wl)
    fac = synth.root_factors(s["points"], s["R0"], s["p0"], wl)
    W = wl.win_size
    imu_samples, vel, g = synth.make_imu(wl, gyr_sigma=1e-3, acc_sigma=1e-2)
    nm = np.array([0.01] * 3 + [1.0] * 3); nw = np.array([1e-4] * 6)
    imus = np.stack([capi.imu_preintegrate(t, gy, ac, np.zeros(3), np.zeros(3), nm, nw) for (t, gy, ac) in imu_samples])
    states = np.zeros((W, 25))
    for i in range(W):
        states[i, 0] = 0.1 * i
        states[i, 1:10] = s["R0"][i].ravel(); states[i, 10:13] = s["p0"][i] + 0.02 * np.sin(np.arange(3) + i); states[i, 13:16] = vel[i]; states[i, 22:25] = g
    out = {}
    for spec in (None, 1):
        ctx = _ctx(capi, wl, spec); ctx.push_dict(fac)
        poses = np.concatenate([states[:, 1:10], states[:, 10:13]], axis=1)
        ctx.evaluate_only_residual(poses)
        out[spec] = ctx.li_ba_damping_iter(states.copy(), imus.copy(), gravity=gravity, max_iter=12)
        ctx.close()
    a, b = out[None], out[1]
    assert a["trace"].shape == b["trace"].shape
    assert np.array_equal(a["trace"], b["trace"]), np.abs(a["trace"] - b["trace"]).max()
    assert np.array_equal(a["states"], b["states"]) and np.array_equal(a["imus"], b["imus"])
    if gravity:       # (the other variant runs 3 iterations, VM:643)
        assert (b["trace"][:, 1] >= b["trace"][:, 0]).any(), "the scenario should contain rejected steps"
