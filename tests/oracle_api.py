"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(_ROOT, "oracle", "liboracle.so")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build():
    odir = os.path.join(_ROOT, "oracle")
    src = [os.path.join(odir, f) for f in os.listdir(odir) if f.endswith((".cpp", ".hpp"))]
    if not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle"), "-s"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.vso_factor_create.restype = C.c_void_p
        _lib.vso_map_create.restype = C.c_void_p
        _lib.vso_imu_give_evaluate.restype = C.c_double
        _lib.vso_now.restype = C.c_double
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def eig3(A):
    A = _c(A); w = np.empty(3); V = np.empty((3, 3))
    lib().vso_eig3(_p(A), _p(w), _p(V))
    return w, V


def so3_exp(w):
    w = _c(w); R = np.empty((3, 3)); lib().vso_exp(_p(w), _p(R)); return R


def so3_log(R):
    R = _c(R); w = np.empty(3); lib().vso_log(_p(R), _p(w)); return w


def jr(w):
    w = _c(w); J = np.empty((3, 3)); lib().vso_jr(_p(w), _p(J)); return J


def jr_inv(R):
    R = _c(R); J = np.empty((3, 3)); lib().vso_jr_inv(_p(R), _p(J)); return J


def cluster_from_points(pts):
    pts = _c(pts); cl = np.empty(10); lib().vso_cluster_from_points(_p(pts), C.c_int(len(pts)), _p(cl)); return cl


def cluster_transform(cl, pose12):
    cl = _c(cl); pose12 = _c(pose12); out = np.empty(10)
    lib().vso_cluster_transform(_p(cl), _p(pose12), _p(out)); return out


def cluster_cov(cl):
    cl = _c(cl); out = np.empty((3, 3)); lib().vso_cluster_cov(_p(cl), _p(out)); return out


def ldlt_solve(A, b):
    A = _c(A); b = _c(b); x = np.empty(len(b))
    lib().vso_ldlt_solve(_p(A), _p(b), C.c_int(len(b)), _p(x)); return x


def inverse15(A):
    A = _c(A); out = np.empty((15, 15)); lib().vso_inverse15(_p(A), _p(out)); return out


class Factor:
    """Oracle LidarFactor (voxel_map.hpp:124-339)."""

    def __init__(self, win_size):
        self.W = win_size
        self.h = C.c_void_p(lib().vso_factor_create(C.c_int(win_size)))

    def __del__(self):
        try:
            lib().vso_factor_destroy(self.h)
        except Exception:
            pass

    def clear(self):
        lib().vso_factor_clear(self.h)

    def size(self):
        return lib().vso_factor_size(self.h)

    def push(self, clusters, fix, coe, eig_val, eig_vec, pcr_add):
        n = len(coe)
        a = [_c(x) for x in (clusters, fix, coe, eig_val, eig_vec, pcr_add)]
        lib().vso_factor_push(self.h, C.c_int(n), *[_p(x) for x in a])

    def push_dict(self, f):
        self.push(f["clusters"], f["fix"], f["coe"], f["eig_val"], f["eig_vec"], f["pcr_add"])

    def acc_evaluate2(self, poses, head=0, end=None):
        end = self.size() if end is None else end
        poses = _c(poses); n = 6 * self.W
        H = np.empty((n, n)); g = np.empty(n); r = C.c_double()
        lib().vso_factor_acc_evaluate2(self.h, _p(poses), C.c_int(head), C.c_int(end), _p(H), _p(g), C.byref(r))
        return H, g, r.value

    def evaluate_only_residual(self, poses, head=0, end=None):
        end = self.size() if end is None else end
        poses = _c(poses); r = C.c_double()
        lib().vso_factor_evaluate_only_residual(self.h, _p(poses), C.c_int(head), C.c_int(end), C.byref(r))
        return r.value

    def read_back(self):
        n = self.size()
        ev = np.empty((n, 3)); evec = np.empty((n, 9)); pa = np.empty((n, 10))
        lib().vso_factor_read_back(self.h, _p(ev), _p(evec), _p(pa))
        return ev, evec, pa

    def read_inputs(self):
        n = self.size()
        cl = np.empty((n, self.W, 10)); fix = np.empty((n, 10)); coe = np.empty(n)
        lib().vso_factor_read_inputs(self.h, _p(cl), _p(fix), _p(coe))
        return cl, fix, coe

    def as_dict(self):
        cl, fix, coe = self.read_inputs()
        ev, evec, pa = self.read_back()
        return dict(clusters=cl, fix=fix, coe=coe, eig_val=ev, eig_vec=evec, pcr_add=pa)

    def lidar_ba_damping_iter(self, poses, max_iter=3, thd_num=2, parallel=False):
        poses = _c(poses).copy(); n = 6 * self.W
        H = np.empty((n, n)); resis = np.zeros(2); status = C.c_int(0)
        trace = np.zeros(5 * max(max_iter, 1)); ntr = C.c_int(0)
        conv = lib().vso_lidar_ba_damping_iter(self.h, _p(poses), _p(H), _p(resis), C.c_int(max_iter), C.c_int(thd_num),
                                               C.c_int(int(parallel)), C.byref(status), _p(trace), C.byref(ntr))
        return dict(poses=poses, hess=H, resis=resis, converge=bool(conv), status=status.value,
                    trace=trace[:ntr.value].reshape(-1, 5))

    def li_ba_damping_iter(self, states, imus, gravity=False, imu_coef=1e-4, max_iter=3, parallel=False):
        states = _c(states).copy(); imus = _c(imus).copy()
        n = 15 * self.W + (3 if gravity else 0)
        H = np.empty((n, n)); resis = np.zeros(2)
        trace = np.zeros(5 * max(max_iter, 3)); ntr = C.c_int(0)
        lib().vso_li_ba_damping_iter(self.h, _p(states), _p(imus), C.c_int(int(gravity)), C.c_double(imu_coef), C.c_int(max_iter),
                                     C.c_int(int(parallel)), _p(H), _p(resis), _p(trace), C.byref(ntr))
        return dict(states=states, imus=imus, hess=H, resis=resis, trace=trace[:ntr.value].reshape(-1, 5))


def imu_preintegrate(t, gyr, acc, bg, ba, noise_meas, noise_walk, scale_gravity=1.0):
    t = _c(t); gyr = _c(gyr); acc = _c(acc); bg = _c(bg); ba = _c(ba); nm = _c(noise_meas); nw = _c(noise_walk)
    out = np.empty(304)
    lib().vso_imu_preintegrate(C.c_int(len(t)), _p(t), _p(gyr), _p(acc), _p(bg), _p(ba), _p(nm), _p(nw), C.c_double(scale_gravity), _p(out))
    return out


def imu_give_evaluate(imu, st1, st2, with_g=False, jac=True):
    imu = _c(imu); st1 = _c(st1); st2 = _c(st2)
    nb = 33 if with_g else 30
    jtj = np.zeros((nb, nb)); gg = np.zeros(nb)
    r = lib().vso_imu_give_evaluate(_p(imu), _p(st1), _p(st2), C.c_int(int(with_g)), C.c_int(int(jac)), _p(jtj), _p(gg))
    return r, jtj, gg


def var_init(pnt, ext_pose12, dept_err, beam_err):
    pnt = _c(pnt).copy(); ext = _c(ext_pose12); var = np.empty((len(pnt), 9))
    lib().vso_var_init(C.c_int(len(pnt)), _p(pnt), _p(ext), C.c_double(dept_err), C.c_double(beam_err), _p(var))
    return pnt, var


def pvec_update(pnt, var, state25, cov225):
    pnt = _c(pnt); var = _c(var).copy(); st = _c(state25); cov = _c(cov225); pw = np.empty((len(pnt), 3))
    lib().vso_pvec_update(C.c_int(len(pnt)), _p(pnt), _p(var), _p(st), _p(cov), _p(pw))
    return var, pw


def down_sampling_voxel(pnt, voxel_size):
    pnt = _c(pnt); n = len(pnt)
    out = np.empty((max(n, 1), 3)); cnt = np.zeros(max(n, 1), dtype=np.int32); first = np.zeros(max(n, 1), dtype=np.int32)
    lib().vso_down_sampling_voxel.restype = C.c_int
    m = lib().vso_down_sampling_voxel(C.c_int(n), _p(pnt), C.c_double(voxel_size), _p(out), cnt.ctypes.data_as(C.POINTER(C.c_int)),
                                      first.ctypes.data_as(C.POINTER(C.c_int)))
    return out[:m].copy(), cnt[:m].copy(), first[:m].copy()


def down_sampling_pvec(pnt, var, voxel_size):
    pnt = _c(pnt); var = _c(var); n = len(pnt)
    out = np.empty((max(n, 1), 3)); vd = np.empty((max(n, 1), 3)); cnt = np.zeros(max(n, 1), dtype=np.int32)
    lib().vso_down_sampling_pvec.restype = C.c_int
    m = lib().vso_down_sampling_pvec(C.c_int(n), _p(pnt), _p(var), C.c_double(voxel_size), _p(out), _p(vd), cnt.ctypes.data_as(C.POINTER(C.c_int)))
    return out[:m].copy(), vd[:m].copy(), cnt[:m].copy()


def down_sampling_close(pnt, voxel_size):
    pnt = _c(pnt); n = len(pnt)
    keep = np.zeros(max(n, 1), dtype=np.int32)
    lib().vso_down_sampling_close.restype = C.c_int
    m = lib().vso_down_sampling_close(C.c_int(n), _p(pnt), C.c_double(voxel_size), keep.ctypes.data_as(C.POINTER(C.c_int)))
    return keep[:m].copy()


def undistort(pnt, curv, imu_poses22, end_pose12, ext_pose12):
    pnt = _c(pnt).copy(); curv = _c(curv); ip = _c(imu_poses22)
    lib().vso_undistort(C.c_int(len(pnt)), _p(pnt), _p(curv), C.c_int(len(ip)), _p(ip), _p(_c(end_pose12)), _p(_c(ext_pose12)))
    return pnt


def gba_cfg13(gba_voxel_size, gba_min_eigen_value, gba_eig, voxel_size, min_eigen_value, plane_eig, max_layer):
    return np.array([gba_voxel_size, gba_min_eigen_value, *gba_eig, voxel_size, min_eigen_value, *plane_eig, max_layer], dtype=np.float64)


def _ragged(clouds):
    off = np.zeros(len(clouds) + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(c) for c in clouds])
    return off, _c(np.concatenate(clouds))


def gba_build(clouds, poses, cfg13):
    """OctreeGBA::cut_voxel over all keyframes + OctreeGBA_multi_recut (loop_refine.hpp:439-537) -> Factor."""
    W = len(clouds)
    off, pnt = _ragged(clouds)
    f = Factor(W)
    lib().vso_gba_build(f.h, C.c_int(W), off.ctypes.data_as(C.POINTER(C.c_int)), _p(pnt), _p(_c(poses)), _p(_c(cfg13)))
    return f


def hba_add_edge(clouds, poses, cfg13, max_iter, thread_num, want_cloud=True):
    """HBA_add_edge (voxelslam.cpp:2822-3015)."""
    W = len(clouds)
    off, pnt = _ragged(clouds)
    poses = _c(poses).copy()
    edges = np.zeros((W * (W - 1) // 2 + 1, 20)); ne = C.c_int(0)
    cloud = np.zeros((len(pnt), 3)); ccnt = np.zeros(len(pnt), dtype=np.int32); nc = C.c_int(0)
    rl = np.zeros(2 * max_iter + 2); nl = C.c_int(0)
    lib().vso_hba_add_edge.restype = C.c_int
    st = lib().vso_hba_add_edge(C.c_int(W), off.ctypes.data_as(C.POINTER(C.c_int)), _p(pnt), _p(poses), _p(_c(cfg13)), C.c_int(max_iter),
                                C.c_int(thread_num), _p(edges), C.byref(ne), _p(cloud) if want_cloud else None,
                                ccnt.ctypes.data_as(C.POINTER(C.c_int)), C.byref(nc), _p(rl), C.byref(nl))
    return dict(status=st, poses=poses, edges=edges[:ne.value].copy(), cloud=cloud[:nc.value].copy(), cloud_count=ccnt[:nc.value].copy(),
                resis=rl[:nl.value].reshape(-1, 2).copy())


class KdOdom:
    """Oracle of lio_state_estimation_kdtree (voxelslam.cpp:1102-1252)."""

    def __init__(self):
        lib().vso_kd_create.restype = C.c_void_p
        self.h = C.c_void_p(lib().vso_kd_create())

    def __del__(self):
        try:
            lib().vso_kd_destroy(self.h)
        except Exception:
            pass

    def tree(self):
        n = lib().vso_kd_tree_size(self.h)
        out = np.zeros((max(n, 1), 3))
        if n:
            lib().vso_kd_tree_points(self.h, _p(out))
        return out[:n]

    def lio_state_estimation(self, pnt_body, state25, cov225):
        pnt = _c(pnt_body); st = _c(state25).copy(); cov = _c(cov225).copy()
        lib().vso_kd_lio_state_estimation.restype = C.c_int
        it = lib().vso_kd_lio_state_estimation(self.h, C.c_int(len(pnt)), _p(pnt), _p(st), _p(cov))
        return it, st, cov


def map_key(voxel_size, pw):
    pw = _c(pw); k = (C.c_longlong * 3)()
    lib().vso_map_key(C.c_double(voxel_size), _p(pw), k)
    return np.array([k[0], k[1], k[2]], dtype=np.int64)


class VoxelMap:
    """Oracle voxel hash map + octree (map_oracle.hpp)."""
    LEAF_REC = 39

    def __init__(self, win_size, voxel_size, max_layer=2, min_eigen_value=0.0025, plane_thre=(0.25,) * 4,
                 min_point=(5,) * 4, max_points=100, thread_num=5):
        pt = _c(plane_thre); mpnt = _c(min_point)
        self.W = win_size
        self.h = C.c_void_p(lib().vso_map_create(C.c_int(win_size), C.c_double(voxel_size), C.c_int(max_layer), C.c_double(min_eigen_value),
                                                 _p(pt), _p(mpnt), C.c_int(max_points), C.c_int(thread_num)))

    def __del__(self):
        try:
            lib().vso_map_destroy(self.h)
        except Exception:
            pass

    def cut_voxel(self, win_count, pnt_body, pose12, var=None, multi=False):
        pnt_body = _c(pnt_body); pose12 = _c(pose12)
        vp = _p(_c(var)) if var is not None else None
        lib().vso_map_cut_voxel(self.h, C.c_int(win_count), C.c_int(len(pnt_body)), _p(pnt_body), vp, _p(pose12), C.c_int(int(multi)))

    def cut_voxel_fix(self, pnt_world, jour=0.0):
        pnt_world = _c(pnt_world)
        lib().vso_map_cut_voxel_fix(self.h, C.c_int(len(pnt_world)), _p(pnt_world), C.c_double(jour))

    def recut(self, win_count, poses, factor, multi=False):
        poses = _c(poses)
        lib().vso_map_recut(self.h, C.c_int(win_count), _p(poses), factor.h, C.c_int(int(multi)))

    def margi(self, win_count, poses, factor, jour=0.0):
        poses = _c(poses)
        lib().vso_map_margi(self.h, C.c_int(win_count), _p(poses), factor.h, C.c_double(jour))

    def slide(self, mgsize=1):
        lib().vso_map_slide(self.h, C.c_int(mgsize))

    def prune(self, jour, dist=700):
        lib().vso_map_prune(self.h, C.c_double(jour), C.c_int(dist))

    def lio_state_estimation(self, pnt_body, var_body, state25, cov225):
        pnt_body = _c(pnt_body); var_body = _c(var_body)
        state = _c(state25).copy(); cov = _c(cov225).copy()
        trace = np.zeros(64); ntr = C.c_int(0)
        ok = lib().vso_map_lio_state_estimation(self.h, C.c_int(len(pnt_body)), _p(pnt_body), _p(var_body), _p(state), _p(cov), _p(trace), C.byref(ntr))
        return bool(ok), state, cov, trace[:ntr.value].reshape(-1, 3)

    def num_roots(self):
        return lib().vso_map_num_roots(self.h)

    def num_slide_roots(self):
        return lib().vso_map_num_slide_roots(self.h)

    def dump_leaves(self):
        n = lib().vso_map_dump_leaves(self.h, None, C.c_int(0))
        out = np.zeros((n, self.LEAF_REC))
        if n:
            lib().vso_map_dump_leaves(self.h, _p(out), C.c_int(n))
        return out

    def dump_plane_var(self):
        n = lib().vso_map_dump_plane_var(self.h, None, C.c_int(0))
        out = np.zeros((n, 36))
        if n:
            lib().vso_map_dump_plane_var(self.h, _p(out), C.c_int(n))
        return out

    def dump_cov_add(self):
        n = lib().vso_map_dump_cov_add(self.h, None, C.c_int(0))
        out = np.zeros((n, 45))
        if n:
            lib().vso_map_dump_cov_add(self.h, _p(out), C.c_int(n))
        return out
