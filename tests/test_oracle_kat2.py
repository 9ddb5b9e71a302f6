"""Independent known-answer tests for the parts of the CPU oracle that tests/test_oracle_kat.py does not pin (map_oracle.hpp
and the IMU Jacobians had none): a hand-built three-level octree whose leaf table is known by construction, and the IMU factor's
gradient against central finite differences of its own residual.  CPU only."""
import numpy as np
import pytest


def test_hand_built_three_level_octree(oracle):
    """Root voxel (0,0,0) of size 1 (centre 0.5, voxel_map.hpp:1935-1946): 20 coplanar points in octant 0 and a 3x3x3 lattice in
    octant 7.  Expected (voxel_map.hpp:1396-1456): the root (47 points, not a plane) splits; octant 0 is a planar leaf of layer 1;
    octant 7 (a cube: not a plane) splits again at centre 0.75 into the 8 layer-2 leaves with 8 / 4,4,4 / 2,2,2 / 1 points, of
    which only the 8-point one passes N > min_point and it is not a plane; exactly one factor (the layer-1 plane) is extracted."""
    gx, gy = np.meshgrid(np.linspace(0.05, 0.45, 5), np.linspace(0.05, 0.45, 4))
    A = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, 0.2)], 1)                      # 20 points, plane z = 0.2
    l = np.array([0.6, 0.75, 0.9])
    B = np.stack(np.meshgrid(l, l, l, indexing="ij"), -1).reshape(-1, 3)                  # 27 lattice points in octant 7
    pts = np.concatenate([A, B])
    W = 2
    om = oracle.VoxelMap(W, 1.0, 2, 0.0025, (0.25,) * 4, (5,) * 4, 100, 1)
    pose = np.concatenate([np.eye(3).ravel(), np.zeros(3)])
    om.cut_voxel(0, pts, pose)
    d = om.dump_leaves()
    assert len(d) == 1 and d[0, 3] == 0 and d[0, 5] == 47 and om.num_roots() == 1        # one root leaf before recut
    of = oracle.Factor(W)
    om.recut(1, np.stack([pose, pose]), of, multi=False)
    d = om.dump_leaves()
    tab = {(int(r[3]), int(r[4])): r for r in d}
    assert all(tuple(r[:3]) == (0, 0, 0) for r in d)
    # layer 1: octant index 4 x + 2 y + z with "coordinate > centre" (voxel_map.hpp:1214-1219): A -> 0, B -> 7
    assert (1, 0) in tab and tab[(1, 0)][5] == 20 and tab[(1, 0)][7] == 1
    # layer 2 under octant 7: lattice coordinates {0.6, 0.75} are NOT > 0.75 -> bit 0, 0.9 -> bit 1
    want = {}
    for p in B:
        o = 4 * (p[0] > 0.75) + 2 * (p[1] > 0.75) + (p[2] > 0.75)
        want[int(7 * 8 + o)] = want.get(int(7 * 8 + o), 0) + 1
    assert sorted(want.values()) == [1, 2, 2, 2, 4, 4, 4, 8]
    got = {k[1]: int(r[5]) for k, r in tab.items() if k[0] == 2}
    assert got == want
    assert all(tab[(2, k)][7] == 0 for k in got)                                         # no planes below the cube
    assert set(tab) == {(1, 0)} | {(2, k) for k in want}                                  # octants 1..6 were never touched: no leaves
    # the planar leaf: eigen-pairs of cov = P/N - c c^T (tools.hpp:333-337) against numpy
    c = A.mean(0); cov = A.T @ A / len(A) - np.outer(c, c)
    w, V = np.linalg.eigh(cov)
    r = tab[(1, 0)]
    assert np.abs(r[10:13] - w).max() < 1e-15 and abs(abs(r[13:22].reshape(3, 3)[:, 0] @ V[:, 0]) - 1) < 1e-12
    assert np.allclose(r[22:28], [(A[:, 0] ** 2).sum(), (A[:, 1] * A[:, 0]).sum(), (A[:, 2] * A[:, 0]).sum(), (A[:, 1] ** 2).sum(),
                                  (A[:, 2] * A[:, 1]).sum(), (A[:, 2] ** 2).sum()], rtol=1e-14)
    # tras_opt (voxel_map.hpp:1605-1638): lambda0 / lambda1 <= 0.12 -> exactly one factor, its frame-0 cluster = the 20 body points
    assert of.size() == 1
    cl, fix, coe = of.read_inputs()
    assert cl[0, 0, 9] == 20 and cl[0, 1, 9] == 0 and fix[0, 9] == 0 and coe[0] == 1.0
    assert np.allclose(cl[0, 0, 6:9], A.sum(0), rtol=1e-14)


def _retract(synth, st, d):
    st = st.copy()
    st[1:10] = (st[1:10].reshape(3, 3) @ synth.so3_exp(d[0:3])).ravel()
    st[10:13] += d[3:6]; st[13:16] += d[6:9]; st[16:19] += d[9:12]; st[19:22] += d[12:15]
    return st


@pytest.mark.parametrize("with_g", [False, True])
def test_imu_factor_gradient_vs_finite_differences(oracle, with_g):
    """IMU_PRE::give_evaluate(_g) (preintegration.hpp:137-294) returns rho = r^T cov^-1 r and gg = J^T cov^-1 r, so the central
    finite difference of rho along the reference's retraction (R <- R Exp(d), the rest additive) must be 2 gg.  The reference's
    Jacobian is first order in the rotation residual (jr_inv of the residual is evaluated, its derivative is not), which shows
    as a few 1e-3 of the largest gradient entry at this residual size: asserted at 5e-3, the dominant entries at 1e-6."""
    import voxel_slam_amd  # noqa: F401
    from voxel_slam_amd import synth
    wl = synth.CONFIGS["room20k_w4"]
    s = synth.make_scans(wl)
    imu_samples, vel, g = synth.make_imu(wl, gyr_sigma=1e-3, acc_sigma=1e-2)
    nm = np.array([0.01] * 3 + [1.0] * 3); nw = np.array([1e-4] * 6)
    t, gy, ac = imu_samples[0]
    imu = oracle.imu_preintegrate(t, gy, ac, np.zeros(3), np.zeros(3), nm, nw)

    def state(i):
        st = np.zeros(25)
        st[0] = 0.1 * i; st[1:10] = s["R0"][i].ravel(); st[10:13] = s["p0"][i]; st[13:16] = vel[i] + 0.01 * i
        st[16:19] = [1e-3, -2e-3, 5e-4]; st[19:22] = [0.01, -0.02, 0.005]; st[22:25] = g
        return st
    s1, s2 = state(0), state(1)
    nb = 33 if with_g else 30
    r0, jtj, gg = oracle.imu_give_evaluate(imu, s1, s2, with_g=with_g, jac=True)
    assert np.abs(jtj - jtj.T).max() <= 1e-14 * np.abs(jtj).max() and r0 > 0

    def rho(d):
        a = _retract(synth, s1, d[:15]); b = _retract(synth, s2, d[15:30])
        if with_g:
            a[22:25] += d[30:33]; b[22:25] += d[30:33]
        return oracle.imu_give_evaluate(imu, a, b, with_g=with_g, jac=False)[0]
    h = 1e-6
    fd = np.array([(rho(np.eye(nb)[k] * h) - rho(-np.eye(nb)[k] * h)) / (2 * h) for k in range(nb)])
    big = np.abs(gg) > 0.05 * np.abs(gg).max()
    assert np.abs(fd[big] / gg[big] - 2.0).max() < 2e-2
    assert abs(np.median(fd[np.abs(gg) > 1e-6] / gg[np.abs(gg) > 1e-6]) - 2.0) < 1e-6
    assert np.abs(fd - 2 * gg).max() < 5e-3 * np.abs(gg).max()
    # Gauss-Newton matrix: positive semi-definite, and jtj d approximates the change of gg along d for small d
    assert np.linalg.eigvalsh(jtj).min() > -1e-9 * np.abs(jtj).max()
