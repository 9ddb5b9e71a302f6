"""The occupancy-compact form of the Hessian pass (k_hessian3, vba_options::hessian_compact_tiles; csrc/vba_kernels_h3.hpp) against the
dense-tile form (k_hessian2, the default) and the oracle: on a store extracted by the map (popcount / mask order -> greedy tiles), on a
store pushed by the host in the oracle's order (one class), for several window sizes, and through a whole LM call."""
import dataclasses

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(capi, wl, compact):
    o = capi.options_from_workload(wl)
    o.hessian_compact_tiles = 1 if compact else 0
    return capi.Context(o)


@pytest.mark.parametrize("name,npts", [("hesai200k_w10", 40000), ("avia100k_w10", 100000), ("room20k_w4", 20000)])
def test_compact_tiles_equal_dense_tiles(oracle, name, npts):
    import voxel_slam_amd  # noqa: F401
    from voxel_slam_amd import capi, synth
    wl = dataclasses.replace(synth.CONFIGS[name], n_pts=npts)
    s = synth.make_scans(wl)
    W = wl.win_size
    poses = synth.poses_flat(s["R0"], s["p0"])
    res = []
    for compact in (False, True):
        ctx = _ctx(capi, wl, compact)
        for i in range(W):
            ctx.cut_voxel(i, s["points"][i], poses[i])
        ctx.recut(W, poses, multi=False)
        H, g, r = ctx.acc_evaluate2(poses)
        lm = ctx.lidar_ba_damping_iter(poses, max_iter=3, thd_num=2)
        res.append((H, g, r, lm, ctx.size()))
        ctx.close()
    (H0, g0, r0, lm0, n0), (H1, g1, r1, lm1, n1) = res
    assert n0 == n1 > 50
    # the two forms add the same per-voxel terms in different orders: a rounding-level bar (f64, ~1e5 terms)
    eh, eg, er = np.abs(H1 - H0).max() / np.abs(H0).max(), np.abs(g1 - g0).max() / np.abs(g0).max(), abs(r1 - r0) / abs(r0)
    assert eh < 1e-11 and eg < 1e-10 and er < 1e-12, (eh, eg, er)
    assert np.array_equal(H1, H1.T)
    assert lm0["trace"].shape == lm1["trace"].shape and np.abs(lm0["poses"] - lm1["poses"]).max() < 1e-6   # rounding through three solves (avia: 1e-8)


@pytest.mark.parametrize("W", [2, 5, 8, 10])
def test_compact_tiles_on_a_host_pushed_store(oracle, W):
    """A store pushed through vba_factor_push_voxels keeps the caller's order: one tile class, unions of up to W frames."""
    import voxel_slam_amd  # noqa: F401
    from voxel_slam_amd import capi, synth
    wl = dataclasses.replace(synth.CONFIGS["room20k_w4"], name="h3_w%d" % W, win_size=W, n_pts=12000)
    s = synth.make_scans(wl)
    fac = synth.root_factors(s["points"], s["R0"], s["p0"], wl)
    poses = synth.poses_flat(s["R0"], s["p0"])
    f = oracle.Factor(W); f.push_dict(fac)
    H2, g2, r2 = f.acc_evaluate2(poses)
    ctx = _ctx(capi, wl, True); ctx.push_dict(fac)
    ctx.evaluate_only_residual(poses); f.evaluate_only_residual(poses)
    H, g, r = ctx.acc_evaluate2(poses)
    H2, g2, r2 = f.acc_evaluate2(poses)
    assert np.abs(H - H2).max() < 1e-9 * np.abs(H2).max() and np.abs(g - g2).max() < 1e-9 * np.abs(g2).max() and abs(r - r2) < 1e-11 * abs(r2)
    ctx.close()
