"""Deterministic synthetic workloads for the local-mapping hot path.

Workload synthesis only: scenes, scan patterns, ray casting, ground-truth / perturbed poses and
IMU samples.  Nothing here evaluates the BA; the shipped compute path is the HIP library.

Scenes are axis-aligned rooms (6 walls) plus optional interior rectangular partitions; scans are
ray-cast from the sensor pose with Gaussian range noise along the ray.
"""
from __future__ import annotations

import dataclasses
import math
from typing import List, Tuple

import numpy as np

SEED_BASE = 20241008  # seed = 20241008 + config index


@dataclasses.dataclass
class Workload:
    name: str
    win_size: int
    voxel_size: float
    n_pts: int
    pattern: str            # "spin32" | "avia"
    room: Tuple[float, float, float]   # extents (x, y, z); room spans [-x/2,x/2]x[-y/2,y/2]x[0,z]
    partitions: int
    seed: int
    max_layer: int = 2
    min_eigen_value: float = 0.0025          # config/avia.yaml:32 (LocalBA/min_eigen_value)
    plane_thre: Tuple[float, ...] = (0.25, 0.25, 0.25, 0.25)   # 1/4, stored inverted (VS:930-931)
    min_point: Tuple[float, ...] = (5, 5, 5, 5)                 # VS:917
    max_points: int = 100                     # VM:101
    imu_coef: float = 1e-4                    # VM:500
    range_noise: float = 0.01
    dept_err: float = 0.02                    # config/avia.yaml:27
    beam_err: float = 0.05                    # config/avia.yaml:28


# BASELINE.json configs[0..3]
CONFIGS = {
    "room20k_w4": Workload("room20k_w4", 4, 0.5, 20000, "spin32", (10.0, 8.0, 3.0), 0, SEED_BASE + 0),
    "avia100k_w10": Workload("avia100k_w10", 10, 0.5, 100000, "avia", (40.0, 30.0, 6.0), 6, SEED_BASE + 1),
    "hesai200k_w10": Workload("hesai200k_w10", 10, 0.3, 200000, "spin32", (40.0, 30.0, 6.0), 6, SEED_BASE + 2),
}


def rot_z(yaw: float) -> np.ndarray:
    c, s = math.cos(yaw), math.sin(yaw)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def so3_exp(w: np.ndarray) -> np.ndarray:
    """Rodrigues; same formula as the reference's Exp (tools.hpp:51-66)."""
    n = float(np.linalg.norm(w))
    if n < 1e-11:
        return np.eye(3)
    a = w / n
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + math.sin(n) * K + (1 - math.cos(n)) * (K @ K)


def scene_planes(wl: Workload):
    """Returns a list of (axis, coord, lo(3), hi(3)) axis-aligned rectangles."""
    X, Y, Z = wl.room
    x0, x1, y0, y1, z0, z1 = -X / 2, X / 2, -Y / 2, Y / 2, 0.0, Z
    lo = np.array([x0, y0, z0])
    hi = np.array([x1, y1, z1])
    planes = []
    for ax in range(3):
        for c in (lo[ax], hi[ax]):
            planes.append((ax, float(c), lo.copy(), hi.copy()))
    # interior partitions: thin walls parallel to x or y, leaving the sensor corridor free
    part_specs = [
        (0, -X * 0.30, (y0, y0 + Y * 0.55)), (0, X * 0.25, (y1 - Y * 0.55, y1)),
        (1, -Y * 0.28, (x0 + X * 0.10, x0 + X * 0.40)), (1, Y * 0.30, (x1 - X * 0.45, x1 - X * 0.10)),
        (0, X * 0.05, (y0, y0 + Y * 0.35)), (1, Y * 0.12, (x0, x0 + X * 0.22)),
    ]
    for k in range(min(wl.partitions, len(part_specs))):
        ax, c, (a, b) = part_specs[k]
        plo, phi = lo.copy(), hi.copy()
        other = 1 - ax
        plo[other], phi[other] = a, b
        phi[2] = Z * 0.8
        planes.append((ax, float(c), plo, phi))
    return planes


def scan_dirs(wl: Workload, rng: np.random.Generator) -> np.ndarray:
    """Unit ray directions in the body frame, shape (n_pts, 3)."""
    n = wl.n_pts
    if wl.pattern == "spin32":
        n_el = 32
        n_az = n // n_el
        el = np.deg2rad(np.linspace(-16.0, 15.0, n_el))
        az = np.linspace(-math.pi, math.pi, n_az, endpoint=False)
        A, E = np.meshgrid(az, el, indexing="ij")   # azimuth-major like a spinning sensor
        A, E = A.ravel(), E.ravel()
    elif wl.pattern == "avia":
        A = rng.uniform(np.deg2rad(-35.2), np.deg2rad(35.2), n)
        E = rng.uniform(np.deg2rad(-38.6), np.deg2rad(38.6), n)
    else:
        raise ValueError(wl.pattern)
    d = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], axis=1)
    return d


def ray_cast(origin: np.ndarray, dirs_w: np.ndarray, planes) -> np.ndarray:
    """Distance along each world-frame ray to the nearest rectangle (inf if none)."""
    t_best = np.full(dirs_w.shape[0], np.inf)
    eps = 1e-9
    for ax, c, lo, hi in planes:
        dn = dirs_w[:, ax]
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (c - origin[ax]) / dn
        ok = np.isfinite(t) & (t > 1e-3)
        t = np.where(ok, t, 0.0)
        hit = origin[None, :] + t[:, None] * dirs_w
        for k in range(3):
            if k != ax:
                ok &= (hit[:, k] >= lo[k] - eps) & (hit[:, k] <= hi[k] + eps)
        t_best = np.where(ok & (t < t_best), t, t_best)
    return t_best


def gt_poses(wl: Workload) -> Tuple[np.ndarray, np.ndarray]:
    """Ground truth: p_i = (-2+0.1 i, 0.05 i, 1.5), yaw = 1 deg * i."""
    W = wl.win_size
    R = np.stack([rot_z(math.radians(1.0) * i) for i in range(W)])
    p = np.stack([np.array([-2.0 + 0.1 * i, 0.05 * i, 1.5]) for i in range(W)])
    if wl.room[0] < 20:   # small room: keep the sensor inside
        p[:, 0] += 2.0
    return R, p


def perturbed_poses(R: np.ndarray, p: np.ndarray, rng: np.random.Generator,
                    rot_sigma_deg: float = 0.3, pos_sigma: float = 0.02) -> Tuple[np.ndarray, np.ndarray]:
    """Initial estimates = GT (+) N(0,(0.3 deg)^2) rot, N(0,(2 cm)^2) trans for i>=1; pose 0 exact (gauge)."""
    R2, p2 = R.copy(), p.copy()
    for i in range(1, R.shape[0]):
        R2[i] = R[i] @ so3_exp(rng.normal(0.0, math.radians(rot_sigma_deg), 3))
        p2[i] = p[i] + rng.normal(0.0, pos_sigma, 3)
    return R2, p2


def make_scans(wl: Workload) -> dict:
    """Returns dict(points=[W arrays (n_i,3) body frame], R_gt, p_gt, R0, p0)."""
    rng = np.random.default_rng(wl.seed)
    planes = scene_planes(wl)
    R_gt, p_gt = gt_poses(wl)
    scans = []
    for i in range(wl.win_size):
        d_b = scan_dirs(wl, rng)
        d_w = d_b @ R_gt[i].T
        t = ray_cast(p_gt[i], d_w, planes)
        ok = np.isfinite(t) & (t > 0.3) & (t < 80.0)
        t = t + rng.normal(0.0, wl.range_noise, t.shape[0])
        pts = d_b[ok] * t[ok, None]
        scans.append(np.ascontiguousarray(pts))
    R0, p0 = perturbed_poses(R_gt, p_gt, rng)
    return dict(points=scans, R_gt=R_gt, p_gt=p_gt, R0=R0, p0=p0)


def poses_flat(R: np.ndarray, p: np.ndarray) -> np.ndarray:
    """[W][12] = R row-major (9) + p (3): the pose layout of include/voxelba.h."""
    W = R.shape[0]
    out = np.empty((W, 12))
    out[:, :9] = R.reshape(W, 9)
    out[:, 9:] = p
    return out


def calc_body_var(pb: np.ndarray, range_inc: float, degree_inc: float) -> np.ndarray:
    """Per-point 3x3 body-frame covariance, the measurement model of calcBodyVar (voxelslam.hpp:180-200),
    vectorised; float32 narrowing of range / range_var as in the reference."""
    pb = pb.copy()
    pb[pb[:, 2] == 0, 2] = 0.0001
    rng_ = np.sqrt((pb * pb).sum(1)).astype(np.float32).astype(np.float64)
    range_var = np.float64(np.float32(range_inc) * np.float32(range_inc))
    dv = math.sin(math.radians(np.float32(degree_inc))) ** 2
    d = pb / np.linalg.norm(pb, axis=1, keepdims=True)
    b1 = np.stack([np.ones(len(d)), np.ones(len(d)), -(d[:, 0] + d[:, 1]) / d[:, 2]], 1)
    b1 /= np.linalg.norm(b1, axis=1, keepdims=True)
    b2 = np.cross(b1, d)
    b2 /= np.linalg.norm(b2, axis=1, keepdims=True)
    # A = range * hat(d) @ [b1 b2]
    a1 = rng_[:, None] * np.cross(d, b1)
    a2 = rng_[:, None] * np.cross(d, b2)
    var = range_var * d[:, :, None] * d[:, None, :] + dv * (a1[:, :, None] * a1[:, None, :] + a2[:, :, None] * a2[:, None, :])
    return var


def make_imu(wl: Workload, rate_hz: float = 200.0, scan_dt: float = 0.1, gyr_sigma: float = 0.0, acc_sigma: float = 0.0):
    """IMU samples between consecutive scans for the constant-velocity / constant-yaw-rate GT trajectory.
    Returns list of (t[n], gyr[n,3], acc[n,3]) per interval, velocities v[W,3], gravity g."""
    rng = np.random.default_rng(wl.seed + 77)
    W = wl.win_size
    g = np.array([0.0, 0.0, -9.8])
    yaw_rate = math.radians(1.0) / scan_dt
    vel = np.array([0.1 / scan_dt, 0.05 / scan_dt, 0.0])
    n = int(round(rate_hz * scan_dt)) + 1
    out = []
    for i in range(W - 1):
        t = i * scan_dt + np.arange(n) / rate_hz
        gyr = np.tile(np.array([0.0, 0.0, yaw_rate]), (n, 1)) + rng.normal(0, 1, (n, 3)) * gyr_sigma
        acc = np.empty((n, 3))
        for k in range(n):
            Rk = rot_z(yaw_rate * t[k])
            acc[k] = Rk.T @ (-g)
        acc += rng.normal(0, 1, (n, 3)) * acc_sigma
        out.append((t, gyr, acc))
    v = np.tile(vel, (W, 1))
    return out, v, g


# ----------------------------------------------------------------------------------------------
# Root-voxel factor synthesis (numpy).  Used to produce LidarFactor-shaped inputs for factor-level
# tests before/without the device voxel map: one factor per planar ROOT voxel (no octree levels).

def voxel_key(pw: np.ndarray, voxel_size: float) -> np.ndarray:
    """The reference's key function (voxel_map.hpp:1907-1918): float narrowing, -1 if negative, truncation."""
    loc = (pw / voxel_size).astype(np.float32)
    loc = np.where(loc < 0, loc - np.float32(1.0), loc).astype(np.float32)
    return loc.astype(np.int64)


def pack_clusters(P: np.ndarray, v: np.ndarray, N: np.ndarray) -> np.ndarray:
    out = np.empty(P.shape[:-2] + (10,))
    out[..., 0] = P[..., 0, 0]; out[..., 1] = P[..., 1, 0]; out[..., 2] = P[..., 2, 0]
    out[..., 3] = P[..., 1, 1]; out[..., 4] = P[..., 2, 1]; out[..., 5] = P[..., 2, 2]
    out[..., 6:9] = v
    out[..., 9] = N
    return out


def root_factors(points: List[np.ndarray], R: np.ndarray, p: np.ndarray, wl: Workload, with_keys: bool = False) -> dict:
    W = len(points)
    keys, frames, pw_all, pb_all = [], [], [], []
    for i in range(W):
        pw = points[i] @ R[i].T + p[i]
        keys.append(voxel_key(pw, wl.voxel_size))
        frames.append(np.full(len(pw), i))
        pw_all.append(pw)
        pb_all.append(points[i])
    keys = np.concatenate(keys); frames = np.concatenate(frames)
    pw_all = np.concatenate(pw_all); pb_all = np.concatenate(pb_all)
    ukeys, vid = np.unique(keys, axis=0, return_inverse=True)
    vid = vid.ravel()
    V = int(vid.max()) + 1
    Pw = np.zeros((V, 3, 3)); vw = np.zeros((V, 3)); Nw = np.zeros(V)
    np.add.at(Pw, vid, pw_all[:, :, None] * pw_all[:, None, :])
    np.add.at(vw, vid, pw_all)
    np.add.at(Nw, vid, 1.0)
    Pb = np.zeros((V, W, 3, 3)); vb = np.zeros((V, W, 3)); Nb = np.zeros((V, W))
    np.add.at(Pb, (vid, frames), pb_all[:, :, None] * pb_all[:, None, :])
    np.add.at(vb, (vid, frames), pb_all)
    np.add.at(Nb, (vid, frames), 1.0)
    ok = Nw > wl.min_point[0]
    c = vw / np.maximum(Nw, 1)[:, None]
    cov = Pw / np.maximum(Nw, 1)[:, None, None] - c[:, :, None] * c[:, None, :]
    lam, U = np.linalg.eigh(cov)
    with np.errstate(divide="ignore", invalid="ignore"):
        plane = ok & (lam[:, 0] < wl.min_eigen_value) & (lam[:, 0] / lam[:, 2] < wl.plane_thre[0]) & (lam[:, 0] / lam[:, 1] <= 0.12)
    sel = np.nonzero(plane)[0]
    out = dict(
        clusters=np.ascontiguousarray(pack_clusters(Pb[sel], vb[sel], Nb[sel])),      # [V][W][10]
        fix=np.zeros((len(sel), 10)),
        coe=np.ones(len(sel)),
        eig_val=np.ascontiguousarray(lam[sel]),
        eig_vec=np.ascontiguousarray(U[sel].reshape(len(sel), 9)),                    # row-major, columns = eigenvectors
        pcr_add=np.ascontiguousarray(pack_clusters(Pw[sel], vw[sel], Nw[sel])),
    )
    if with_keys:
        out["keys"] = np.ascontiguousarray(ukeys[sel])
    return out
