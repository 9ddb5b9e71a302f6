#!/usr/bin/env python3
"""Generates the committed known-answer fixtures in tests/golden/ (SURVEY.md §4 KAT-1..5, §8c).

The reference ships no tests, golden files or recorded outputs and cannot be built here (no Eigen /
ROS), so these vectors are produced with numpy/scipy ONLY (no oracle, no product code): they pin the
oracle independently of its own arithmetic.

  kat_eig3.npz      symmetric 3x3 matrices -> numpy.linalg.eigh (ascending)               [KAT-3]
  kat_lambda.npz    voxel factors + poses -> lambda_min, central-FD gradient and FD Hessian
                    of sum(coe*lambda_min) w.r.t. the retraction R<-R Exp(dth), p<-p+dp  [KAT-1, KAT-2]
  kat_cluster.npz   points, pose -> cluster(points), cluster(R points + t)                [KAT-4]
  kat_key.npz       world coordinates, voxel sizes -> voxel keys (float-narrowing quirk)   [KAT-5]
  kat_linalg.npz    SPD/indefinite systems -> numpy solve; 15x15 -> inv; so3 vectors -> scipy Rotation

Run:  python tests/golden/make_golden.py     (deterministic; seeds fixed)
"""
import os

import numpy as np
from scipy.spatial.transform import Rotation

HERE = os.path.dirname(os.path.abspath(__file__))


def so3_exp(w):
    return Rotation.from_rotvec(w).as_matrix()


def cluster(pts):
    P = pts.T @ pts
    v = pts.sum(0)
    return np.array([P[0, 0], P[1, 0], P[2, 0], P[1, 1], P[2, 1], P[2, 2], v[0], v[1], v[2], len(pts)])


def unpack(c):
    P = np.array([[c[0], c[1], c[2]], [c[1], c[3], c[4]], [c[2], c[4], c[5]]])
    return P, c[6:9].copy(), c[9]


def lam_min_sum(clusters, fix, coe, R, p):
    """sum_a coe_a * lambda_min( cov( fix_a + sum_i transform(cluster_ai, pose_i) ) )   (tools.hpp:357-363, voxel_map.hpp:297-323)"""
    tot = 0.0
    lams = []
    for a in range(clusters.shape[0]):
        P, v, N = unpack(fix[a])
        for i in range(clusters.shape[1]):
            Pi, vi, ni = unpack(clusters[a, i])
            if ni == 0:
                continue
            Rv = R[i] @ vi
            rp = np.outer(Rv, p[i])
            P = P + R[i] @ Pi @ R[i].T + rp + rp.T + ni * np.outer(p[i], p[i])
            v = v + Rv + ni * p[i]
            N = N + ni
        c = v / N
        lam = np.linalg.eigvalsh(P / N - np.outer(c, c))
        lams.append(lam)
        tot += coe[a] * lam[0]
    return tot, np.array(lams)


def gen_eig():
    rng = np.random.default_rng(1)
    mats = []
    for k in range(300):
        A = rng.normal(size=(3, 3))
        kind = k % 6
        if kind == 0:
            S = A @ A.T
        elif kind == 1:      # thin plane: one tiny eigenvalue
            Q, _ = np.linalg.qr(A)
            S = Q @ np.diag([10.0 ** rng.uniform(-8, -3), rng.uniform(0.01, 1), rng.uniform(0.01, 1)]) @ Q.T
        elif kind == 2:      # two (nearly) equal eigenvalues
            Q, _ = np.linalg.qr(A)
            e = rng.uniform(0.01, 1)
            S = Q @ np.diag([1e-4, e, e * (1 + 10.0 ** rng.uniform(-14, -6))]) @ Q.T
        elif kind == 3:      # already diagonal / axis aligned
            S = np.diag(rng.uniform(0, 1, 3))
        elif kind == 4:      # large offsets (world-coordinate second moments) with cancellation
            pts = rng.normal(size=(40, 3)) * np.array([1.0, 0.5, 0.01]) + rng.uniform(-50, 50, 3)
            c = pts.mean(0)
            S = pts.T @ pts / len(pts) - np.outer(c, c)
        else:                # indefinite symmetric
            S = A + A.T
        S = 0.5 * (S + S.T)
        mats.append(S)
    mats = np.array(mats)
    w = np.array([np.linalg.eigh(m)[0] for m in mats])
    np.savez_compressed(os.path.join(HERE, "kat_eig3.npz"), A=mats, w=w)


def gen_lambda():
    rng = np.random.default_rng(2)
    W, V = 4, 6
    R = np.array([so3_exp(rng.normal(0, 0.2, 3)) for _ in range(W)])
    p = rng.normal(0, 1.0, (W, 3))
    clusters = np.zeros((V, W, 10))
    fix = np.zeros((V, 10))
    for a in range(V):
        # a noisy plane patch in the world, observed from each frame (some frames empty, some voxels with fixed points)
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        c0 = rng.uniform(-3, 3, 3)
        B = np.linalg.svd(n[None, :])[2][1:]    # two in-plane directions
        for i in range(W):
            if rng.uniform() < 0.25 and i != a % W:
                continue
            m = int(rng.integers(6, 30))
            q = c0 + rng.uniform(-0.4, 0.4, (m, 2)) @ B + rng.normal(0, 0.02, (m, 1)) * n
            body = (q - p[i]) @ R[i]           # R^T (q - p)
            clusters[a, i] = cluster(body)
        if a % 2 == 0:
            m = int(rng.integers(5, 20))
            q = c0 + rng.uniform(-0.4, 0.4, (m, 2)) @ B + rng.normal(0, 0.02, (m, 1)) * n
            fix[a] = cluster(q)
    coe = rng.uniform(0.5, 1.5, V)
    # evaluate at slightly perturbed poses so the gradient is non-trivial
    R = np.array([R[i] @ so3_exp(rng.normal(0, 0.01, 3)) for i in range(W)])
    p = p + rng.normal(0, 0.01, (W, 3))

    f0, lams = lam_min_sum(clusters, fix, coe, R, p)
    n = 6 * W

    def f(d):
        Rn = np.array([R[i] @ so3_exp(d[6 * i:6 * i + 3]) for i in range(W)])
        pn = p + d.reshape(W, 6)[:, 3:]
        return lam_min_sum(clusters, fix, coe, Rn, pn)[0]

    h = 1e-5
    g = np.zeros(n)
    for i in range(n):
        e = np.zeros(n); e[i] = h
        g[i] = (f(e) - f(-e)) / (2 * h)
    h2 = 2e-4
    H = np.zeros((n, n))
    for i in range(n):
        for j in range(i, n):
            ei = np.zeros(n); ei[i] = h2
            ej = np.zeros(n); ej[j] = h2
            H[i, j] = H[j, i] = (f(ei + ej) - f(ei - ej) - f(-ei + ej) + f(-ei - ej)) / (4 * h2 * h2)
    np.savez_compressed(os.path.join(HERE, "kat_lambda.npz"), clusters=clusters, fix=fix, coe=coe, R=R, p=p,
                        f0=f0, lams=lams, grad_fd=g, hess_fd=H)


def gen_cluster():
    rng = np.random.default_rng(3)
    pts = rng.normal(0, 2.0, (57, 3)) + np.array([10.0, -4.0, 1.5])
    R = so3_exp(np.array([0.3, -0.2, 0.9]))
    t = np.array([4.0, -7.0, 0.25])
    np.savez_compressed(os.path.join(HERE, "kat_cluster.npz"), pts=pts, R=R, t=t, c_body=cluster(pts), c_world=cluster(pts @ R.T + t))


def gen_key():
    rng = np.random.default_rng(4)
    special = np.array([0.0, -0.0, 0.3, -0.3, 0.6, -0.6, 0.29999999, -0.29999999, 1.0, -1.0, 0.5, -0.5, 2.5, -2.5,
                        1e-9, -1e-9, 123.456, -123.456, 16777216.5 * 0.5, -16777217.0 * 0.5, 0.3 * 7, -0.3 * 7, 0.5 * 1e6, -0.5 * 1e6])
    coords = np.concatenate([special, rng.uniform(-60, 60, 400), np.round(rng.uniform(-40, 40, 100)) * 0.5,
                             np.round(rng.uniform(-40, 40, 100)) * 0.3])
    out = {}
    for vs in (0.3, 0.5, 1.0, 2.0):
        loc = (coords / vs).astype(np.float32)
        loc = np.where(loc < 0, loc - np.float32(1.0), loc).astype(np.float32)
        out["key_%g" % vs] = loc.astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "kat_key.npz"), coords=coords, **out)


def gen_linalg():
    rng = np.random.default_rng(5)
    As, bs, xs = [], [], []
    for k in range(6):
        n = 60
        M = rng.normal(size=(n, n))
        A = M @ M.T + np.eye(n) * 1e-3
        if k >= 3:   # symmetric indefinite (second-order Hessians can be)
            d = rng.normal(size=n)
            Q, _ = np.linalg.qr(M)
            A = Q @ np.diag(d * 10) @ Q.T
            A = 0.5 * (A + A.T)
        b = rng.normal(size=n)
        As.append(A); bs.append(b); xs.append(np.linalg.solve(A, b))
    C = rng.normal(size=(15, 15)); C = C @ C.T + np.eye(15) * 0.1
    rv = rng.normal(0, 1.0, (40, 3))
    rv[:5] *= 1e-6
    rv[5:8] = 0.0
    Rm = Rotation.from_rotvec(rv).as_matrix()
    np.savez_compressed(os.path.join(HERE, "kat_linalg.npz"), A=np.array(As), b=np.array(bs), x=np.array(xs),
                        C=C, Cinv=np.linalg.inv(C), rotvec=rv, rotmat=Rm)


if __name__ == "__main__":
    gen_eig(); gen_lambda(); gen_cluster(); gen_key(); gen_linalg()
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))
