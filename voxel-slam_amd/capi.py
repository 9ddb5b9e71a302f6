"""ctypes binding of libvoxelba.so (include/voxelba.h) plus thin Python mirrors of the reference classes
``LidarFactor`` / ``Lidar_BA_Optimizer`` / ``LI_BA_Optimizer`` / ``LI_BA_OptimizerGravity`` (voxel_map.hpp:124-976)
and of the voxel-map entry points (``cut_voxel`` / ``multi_recut`` / ``multi_margi``).

The product has no CPU path: ``load()`` raises if the shared library is missing, ``Context()`` raises if no
HIP device is present.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VBA_LIB") or os.path.join(_PKG, "libvoxelba.so")   # VBA_LIB: an alternative build (tools/ A/B runs)
_dp = C.POINTER(C.c_double)

OK = 0
ERR_NO_DEVICE, ERR_BAD_ARG, ERR_UNSUPPORTED_WINDOW, ERR_TOO_FEW_VOXELS, ERR_OPT_STATE, ERR_HIP, ERR_CAPACITY, ERR_IO, ERR_UNSUPPORTED = range(1, 10)

EXPORTS = [
    "vba_default_options", "vba_create", "vba_destroy", "vba_status_string", "vba_last_error", "vba_synchronize",
    "vba_factor_clear", "vba_factor_push_voxels", "vba_factor_size", "vba_factor_acc_evaluate2",
    "vba_factor_evaluate_only_residual", "vba_factor_read_back", "vba_factor_occupied_slots", "vba_factor_occupancy_masks",
    "vba_lidar_ba_damping_iter", "vba_li_ba_damping_iter", "vba_last_lm_trace",
    "vba_imu_preintegrate", "vba_imu_give_evaluate",
    "vba_map_cut_voxel", "vba_map_pvec_update_cut_voxel", "vba_scan_var_init", "vba_scan_down_sampling_voxel", "vba_scan_down_sampling_pvec", "vba_scan_down_sampling_close", "vba_scan_undistort", "vba_odom_lio_state_estimation_kdtree", "vba_odom_kdtree_reset", "vba_odom_kdtree_size", "vba_odom_kdtree_points", "vba_gba_build", "vba_hba_add_edge", "vba_hba_global", "vba_map_cut_voxel_fix", "vba_map_recut", "vba_map_margi", "vba_map_slide", "vba_map_prune", "vba_map_reset",
    "vba_map_num_roots", "vba_map_num_slide_roots", "vba_map_stats", "vba_map_dump_leaves", "vba_map_dump_plane_var", "vba_odom_lio_state_estimation",
    "vba_set_allreduce", "vba_rccl_get_unique_id", "vba_rccl_init", "vba_set_rccl_comm", "vba_shard_owner", "vba_set_shard",
    "vba_timing_enable", "vba_timing_calibration_read", "vba_timing_select", "vba_timing_sample_every", "vba_timing_launch_hessian", "vba_timing_null_span", "vba_timing_reset", "vba_timing_get",
    "vba_lm_begin", "vba_lm_refresh_eigen", "vba_lm_iterate", "vba_lm_end",
    "vba_io_save_pcd", "vba_io_load_pcd", "vba_io_save_pose", "vba_io_read_lidarstate",
]


class Options(C.Structure):
    _fields_ = [
        ("win_size", C.c_int), ("voxel_size", C.c_double), ("max_layer", C.c_int), ("max_points", C.c_int),
        ("min_eigen_value", C.c_double), ("plane_eigen_value_thre", C.c_double * 4), ("min_point", C.c_double * 4),
        ("imu_coef", C.c_double), ("thread_num", C.c_int), ("device", C.c_int), ("stream", C.c_void_p),
        ("max_voxels", C.c_size_t), ("max_points_per_scan", C.c_size_t),
        ("lm_spec", C.c_int), ("force_collective", C.c_int), ("hessian_workgroups", C.c_int), ("residual_vpl_from", C.c_int), ("hessian_compact_tiles", C.c_int),
        ("max_map_nodes", C.c_size_t), ("max_fix_points", C.c_size_t), ("hba_workers", C.c_int),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class VbaError(RuntimeError):
    def __init__(self, status, msg=""):
        super().__init__("libvoxelba status %d: %s" % (status, msg))
        self.status = status


def build(force: bool = False) -> str:
    """Compile libvoxelba.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    src = os.path.join(_PKG, "csrc")
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["make", "-C", src, "-s"])
    return LIB_PATH


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libvoxelba.so is missing (run __graft_entry__.build()); there is no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        lib.vba_status_string.restype = C.c_char_p
        lib.vba_last_error.restype = C.c_char_p
        lib.vba_shard_owner.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int]
        _lib = lib
    return _lib


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def default_options() -> Options:
    o = Options()
    load().vba_default_options(C.byref(o))
    # test-harness hook of THIS binding (the library itself reads no environment): "field=value,field=value" applied to every
    # options struct this process builds, e.g. VBA_PY_OPTIONS="residual_vpl_from=1" in the child pytest of tests/test_gpu_bigstore.py
    for kv in filter(None, os.environ.get("VBA_PY_OPTIONS", "").split(",")):
        k, v = kv.split("=")
        setattr(o, k.strip(), type(getattr(o, k.strip()))(float(v)))
    return o


def options_from_workload(wl, stream=None) -> Options:
    o = default_options()
    o.win_size = wl.win_size
    o.voxel_size = wl.voxel_size
    o.max_layer = wl.max_layer
    o.max_points = wl.max_points
    o.min_eigen_value = wl.min_eigen_value
    for i in range(4):
        o.plane_eigen_value_thre[i] = wl.plane_thre[i]
        o.min_point[i] = wl.min_point[i]
    o.imu_coef = wl.imu_coef
    if stream is not None:
        o.stream = stream
    return o


def shard_owner(key3, n_ranks: int) -> int:
    return load().vba_shard_owner(int(key3[0]), int(key3[1]), int(key3[2]), int(n_ranks))


def imu_preintegrate(t, gyr, acc, bg, ba, noise_meas, noise_walk, scale_gravity=1.0):
    t, gyr, acc, bg, ba, nm, nw = map(_c, (t, gyr, acc, bg, ba, noise_meas, noise_walk))
    out = np.empty(304)
    st = load().vba_imu_preintegrate(C.c_int(len(t)), _p(t), _p(gyr), _p(acc), _p(bg), _p(ba), _p(nm), _p(nw), C.c_double(scale_gravity), _p(out))
    if st:
        raise VbaError(st)
    return out


def imu_give_evaluate(imu, st1, st2, with_g=False, jac=True):
    imu, st1, st2 = map(_c, (imu, st1, st2))
    nb = 33 if with_g else 30
    jtj = np.zeros((nb, nb)); gg = np.zeros(nb); r = C.c_double()
    st = load().vba_imu_give_evaluate(_p(imu), _p(st1), _p(st2), C.c_int(int(with_g)), C.c_int(int(jac)), _p(jtj), _p(gg), C.byref(r))
    if st:
        raise VbaError(st)
    return r.value, jtj, gg


def _io_chk(st):
    if st:
        raise VbaError(st, load().vba_status_string(st).decode())


def save_pcd(path, xyz):
    """FileReaderWriter::save_pcd (voxelslam.cpp:166-179): binary PCD of PointXYZI, intensity 0."""
    xyz = _c(xyz).reshape(-1, 3)
    _io_chk(load().vba_io_save_pcd(os.fsencode(path), C.c_int(len(xyz)), _p(xyz)))


def load_pcd(path):
    """pcl::io::loadPCDFile as used by previous_map_read (voxelslam.cpp:337-340) -> (xyz [n,3], intensity [n])."""
    n = C.c_int(0)
    st = load().vba_io_load_pcd(os.fsencode(path), C.c_int(0), None, None, C.byref(n))
    if st not in (0, 7):
        _io_chk(st)
    xyz = np.zeros((max(n.value, 1), 3)); inten = np.zeros(max(n.value, 1))
    _io_chk(load().vba_io_load_pcd(os.fsencode(path), C.c_int(n.value), _p(xyz), _p(inten), C.byref(n)))
    return xyz[:n.value], inten[:n.value]


def save_pose(path, states, v6):
    """FileReaderWriter::save_pose (voxelslam.cpp:181-204); writes nothing for fewer than 100 scans."""
    states = _c(states).reshape(-1, 25); v6 = _c(v6).reshape(-1, 6)
    _io_chk(load().vba_io_save_pose(os.fsencode(path), C.c_int(len(states)), _p(states), _p(v6)))


def read_lidarstate(path):
    """read_lidarstate (voxelslam.hpp:268-307) -> (states [n,25], v6 [n,6])."""
    n = C.c_int(0)
    st = load().vba_io_read_lidarstate(os.fsencode(path), C.c_int(0), None, None, C.byref(n))
    if st not in (0, 7):
        _io_chk(st)
    states = np.zeros((max(n.value, 1), 25)); v6 = np.zeros((max(n.value, 1), 6))
    _io_chk(load().vba_io_read_lidarstate(os.fsencode(path), C.c_int(n.value), _p(states), _p(v6), C.byref(n)))
    return states[:n.value], v6[:n.value]


class Context:
    """One vba_ctx: a HIP stream, the HBM factor store (``LidarFactor``) and the device voxel map."""

    def __init__(self, opt: Options):
        self.lib = load()
        self.opt = opt
        self.W = opt.win_size
        h = C.c_void_p()
        st = self.lib.vba_create(C.byref(opt), C.byref(h))
        if st:
            raise VbaError(st, self.lib.vba_status_string(st).decode())
        self.h = h
        self._cb = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.vba_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st):
        if st:
            raise VbaError(st, self.lib.vba_status_string(st).decode() + " | " + self.lib.vba_last_error(self.h).decode())

    def synchronize(self):
        self._chk(self.lib.vba_synchronize(self.h))

    # ---- LidarFactor (voxel_map.hpp:124-339)
    def clear(self):
        self._chk(self.lib.vba_factor_clear(self.h))

    def size(self) -> int:
        return self.lib.vba_factor_size(self.h)

    def push_voxels(self, clusters, fix, coe, eig_val, eig_vec, pcr_add):
        a = [_c(x) for x in (clusters, fix, coe, eig_val, eig_vec, pcr_add)]
        self._chk(self.lib.vba_factor_push_voxels(self.h, C.c_int(len(a[2])), *[_p(x) for x in a]))

    def push_dict(self, f):
        self.push_voxels(f["clusters"], f["fix"], f["coe"], f["eig_val"], f["eig_vec"], f["pcr_add"])

    def acc_evaluate2(self, poses, head=0, end=None):
        end = self.size() if end is None else end
        poses = _c(poses); n = 6 * self.W
        H = np.empty((n, n)); g = np.empty(n); r = C.c_double()
        self._chk(self.lib.vba_factor_acc_evaluate2(self.h, _p(poses), C.c_int(head), C.c_int(end), _p(H), _p(g), C.byref(r)))
        return H, g, r.value

    def evaluate_only_residual(self, poses, head=0, end=None):
        end = self.size() if end is None else end
        poses = _c(poses); r = C.c_double()
        self._chk(self.lib.vba_factor_evaluate_only_residual(self.h, _p(poses), C.c_int(head), C.c_int(end), C.byref(r)))
        return r.value

    def read_back(self):
        n = self.size()
        ev = np.empty((n, 3)); evec = np.empty((n, 9)); pa = np.empty((n, 10))
        self._chk(self.lib.vba_factor_read_back(self.h, _p(ev), _p(evec), _p(pa)))
        return ev, evec, pa

    def factor_occupancy(self) -> float:
        """Occupied (voxel, frame) slots per voxel in the factor store."""
        n = C.c_longlong(0)
        self._chk(self.lib.vba_factor_occupied_slots(self.h, C.byref(n)))
        return n.value / max(self.size(), 1)

    def factor_occupancy_masks(self):
        m = np.zeros(self.size(), dtype=np.uint32)
        self._chk(self.lib.vba_factor_occupancy_masks(self.h, m.ctypes.data_as(C.POINTER(C.c_uint))))
        return m

    # ---- optimizers
    def last_trace(self):
        rows = np.zeros((64, 5))
        n = self.lib.vba_last_lm_trace(self.h, _p(rows), C.c_int(64))
        return rows[:n].copy()

    def lidar_ba_damping_iter(self, poses, max_iter=3, thd_num=2):
        """Lidar_BA_Optimizer::damping_iter (voxel_map.hpp:422-497)."""
        poses = _c(poses).copy(); n = 6 * self.W
        H = np.empty((n, n)); resis = np.zeros(2); conv = C.c_int(0)
        st = self.lib.vba_lidar_ba_damping_iter(self.h, _p(poses), _p(H), _p(resis), C.c_int(max_iter), C.c_int(thd_num), C.byref(conv))
        if st == ERR_TOO_FEW_VOXELS:
            return dict(poses=poses, hess=H, resis=resis, converge=False, status=-1, trace=self.last_trace())
        self._chk(st)
        return dict(poses=poses, hess=H, resis=resis, converge=bool(conv.value), status=0, trace=self.last_trace())

    def li_ba_damping_iter(self, states, imus, gravity=False, max_iter=3):
        """LI_BA_Optimizer::damping_iter (voxel_map.hpp:624-713) / LI_BA_OptimizerGravity::damping_iter (:878-975)."""
        states = _c(states).copy(); imus = _c(imus).copy()
        n = 15 * self.W + (3 if gravity else 0)
        H = np.empty((n, n)); resis = np.zeros(2)
        self._chk(self.lib.vba_li_ba_damping_iter(self.h, _p(states), _p(imus), C.c_int(int(gravity)), C.c_int(max_iter), _p(H), _p(resis)))
        return dict(states=states, imus=imus, hess=H, resis=resis, trace=self.last_trace())

    def lm_begin(self, poses, thd_num=2):
        poses = _c(poses)
        self._chk(self.lib.vba_lm_begin(self.h, _p(poses), C.c_int(thd_num)))

    def lm_iterate(self, sync=True):
        """One LM iteration (loop body voxel_map.hpp:441-494).  sync=False only enqueues the launches."""
        if not sync:
            self._chk(self.lib.vba_lm_iterate(self.h, None, None))
            return None
        acc = C.c_int(0); stop = C.c_int(0)
        self._chk(self.lib.vba_lm_iterate(self.h, C.byref(acc), C.byref(stop)))
        return bool(acc.value), bool(stop.value)

    def lm_refresh_eigen(self):
        self._chk(self.lib.vba_lm_refresh_eigen(self.h))

    def lm_end(self, fetch=True):
        if not fetch:
            self._chk(self.lib.vba_lm_end(self.h, None, None, None))
            return None
        n = 6 * self.W
        poses = np.empty((self.W, 12)); H = np.empty((n, n)); resis = np.zeros(2)
        self._chk(self.lib.vba_lm_end(self.h, _p(poses), _p(H), _p(resis)))
        return poses, H, resis

    # ---- voxel map
    def cut_voxel(self, win_count, pnt_body, pose12, var=None, multi=False):
        pnt_body = _c(pnt_body); pose12 = _c(pose12)
        v = _c(var) if var is not None else None
        self._chk(self.lib.vba_map_cut_voxel(self.h, C.c_int(win_count), C.c_int(len(pnt_body)), _p(pnt_body), _p(v), _p(pose12), C.c_int(int(multi))))

    def pvec_update_cut_voxel(self, win_count, pnt_body, var_body, pose12, cov225, multi=False):
        """pvec_update (voxelslam.hpp:242-265) + cut_voxel[_multi], fused on the device."""
        pnt_body = _c(pnt_body); var_body = _c(var_body); pose12 = _c(pose12); cov = _c(cov225)
        self._chk(self.lib.vba_map_pvec_update_cut_voxel(self.h, C.c_int(win_count), C.c_int(len(pnt_body)), _p(pnt_body), _p(var_body),
                                                         _p(pose12), _p(cov), C.c_int(int(multi))))

    def var_init(self, pnt, ext_pose12, dept_err, beam_err):
        """var_init (voxelslam.hpp:210-234): returns (pnt_out, var_out)."""
        pnt = _c(pnt); ext = _c(ext_pose12)
        po = np.empty_like(pnt); var = np.empty((len(pnt), 9))
        self._chk(self.lib.vba_scan_var_init(self.h, C.c_int(len(pnt)), _p(pnt), _p(ext), C.c_double(dept_err), C.c_double(beam_err), _p(po), _p(var)))
        return po, var

    def down_sampling_voxel(self, pnt, voxel_size):
        pnt = _c(pnt); n = len(pnt)
        out = np.empty((max(n, 1), 3)); cnt = np.zeros(max(n, 1), dtype=np.int32); first = np.zeros(max(n, 1), dtype=np.int32)
        m = C.c_int(0)
        self._chk(self.lib.vba_scan_down_sampling_voxel(self.h, C.c_int(n), _p(pnt), C.c_double(voxel_size), _p(out),
                                                        cnt.ctypes.data_as(C.POINTER(C.c_int)), first.ctypes.data_as(C.POINTER(C.c_int)), C.byref(m)))
        return out[:m.value].copy(), cnt[:m.value].copy(), first[:m.value].copy()

    def down_sampling_pvec(self, pnt, var, voxel_size):
        pnt = _c(pnt); var = _c(var); n = len(pnt)
        out = np.empty((max(n, 1), 3)); vd = np.empty((max(n, 1), 3)); cnt = np.zeros(max(n, 1), dtype=np.int32); m = C.c_int(0)
        self._chk(self.lib.vba_scan_down_sampling_pvec(self.h, C.c_int(n), _p(pnt), _p(var), C.c_double(voxel_size), _p(out), _p(vd),
                                                       cnt.ctypes.data_as(C.POINTER(C.c_int)), C.byref(m)))
        return out[:m.value].copy(), vd[:m.value].copy(), cnt[:m.value].copy()

    def down_sampling_close(self, pnt, voxel_size):
        pnt = _c(pnt); n = len(pnt)
        idx = np.zeros(max(n, 1), dtype=np.int32); m = C.c_int(0)
        self._chk(self.lib.vba_scan_down_sampling_close(self.h, C.c_int(n), _p(pnt), C.c_double(voxel_size), idx.ctypes.data_as(C.POINTER(C.c_int)), C.byref(m)))
        return idx[:m.value].copy()

    def undistort(self, pnt, curv, imu_poses22, end_pose12, ext_pose12):
        pnt = _c(pnt).copy(); curv = _c(curv); ip = _c(imu_poses22)
        self._chk(self.lib.vba_scan_undistort(self.h, C.c_int(len(pnt)), _p(pnt), _p(curv), C.c_int(len(ip)), _p(ip), _p(_c(end_pose12)),
                                              _p(_c(ext_pose12))))
        return pnt

    def cut_voxel_fix(self, pnt_world, jour=0.0):
        pnt_world = _c(pnt_world)
        self._chk(self.lib.vba_map_cut_voxel_fix(self.h, C.c_int(len(pnt_world)), _p(pnt_world), C.c_double(jour)))

    def recut(self, win_count, poses, multi=False):
        poses = _c(poses)
        self._chk(self.lib.vba_map_recut(self.h, C.c_int(win_count), _p(poses), C.c_int(int(multi))))

    def margi(self, win_count, poses, jour=0.0):
        poses = _c(poses)
        self._chk(self.lib.vba_map_margi(self.h, C.c_int(win_count), _p(poses), C.c_double(jour)))

    def slide(self, mgsize=1):
        self._chk(self.lib.vba_map_slide(self.h, C.c_int(mgsize)))

    def prune(self, jour, dist=700):
        self._chk(self.lib.vba_map_prune(self.h, C.c_double(jour), C.c_int(dist)))

    def map_reset(self):
        self._chk(self.lib.vba_map_reset(self.h))

    def num_roots(self):
        return self.lib.vba_map_num_roots(self.h)

    def num_slide_roots(self):
        return self.lib.vba_map_num_slide_roots(self.h)

    def map_stats(self):
        out = (C.c_longlong * 8)()
        self._chk(self.lib.vba_map_stats(self.h, out))
        keys = ("nodes_high_water", "free_roots", "free_blocks", "hash_capacity", "hash_used", "roots", "slide_roots", "fixed_points")
        return dict(zip(keys, [int(x) for x in out]))

    def dump_leaves(self):
        n = self.lib.vba_map_dump_leaves(self.h, None, C.c_int(0))
        out = np.zeros((max(n, 0), 39))
        if n > 0:
            self.lib.vba_map_dump_leaves(self.h, _p(out), C.c_int(n))
        return out

    def dump_plane_var(self):
        """[kx,ky,kz,layer,path, plane_var(36), cov_add upper triangle (45)] per leaf."""
        n = self.lib.vba_map_dump_plane_var(self.h, None, C.c_int(0))
        out = np.zeros((max(n, 0), 86))
        if n > 0:
            self.lib.vba_map_dump_plane_var(self.h, _p(out), C.c_int(n))
        return out

    # ---- odometry
    def lio_state_estimation(self, pnt_body, var_body, state25, cov225):
        """VOXEL_SLAM::lio_state_estimation (voxelslam.cpp:962-1098).  Returns (ok, state, cov)."""
        pnt_body = _c(pnt_body); var_body = _c(var_body)
        state = _c(state25).copy(); cov = _c(cov225).copy(); ok = C.c_int(0)
        self._chk(self.lib.vba_odom_lio_state_estimation(self.h, C.c_int(len(pnt_body)), _p(pnt_body), _p(var_body), _p(state), _p(cov), C.byref(ok)))
        return bool(ok.value), state, cov

    # ---- multi-GPU / timing
    def lio_state_estimation_kdtree(self, pnt_body, state25, cov225):
        pnt = _c(pnt_body); st = _c(state25).copy(); cov = _c(cov225).copy(); it = C.c_int(0)
        self._chk(self.lib.vba_odom_lio_state_estimation_kdtree(self.h, C.c_int(len(pnt)), _p(pnt), _p(st), _p(cov), C.byref(it)))
        return it.value, st, cov

    def kdtree_size(self):
        return self.lib.vba_odom_kdtree_size(self.h)

    def kdtree_points(self):
        n = self.kdtree_size(); out = np.zeros((max(n, 1), 3))
        if n:
            self._chk(self.lib.vba_odom_kdtree_points(self.h, _p(out)))
        return out[:n]

    # ---- hierarchical global BA
    @staticmethod
    def _ragged(clouds):
        off = np.zeros(len(clouds) + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(c) for c in clouds])
        return off, _c(np.concatenate(clouds))

    def gba_build(self, clouds, poses, gba_voxel_size, gba_min_eigen_value, gba_eig):
        off, pnt = self._ragged(clouds)
        self._chk(self.lib.vba_gba_build(self.h, C.c_int(len(clouds)), off.ctypes.data_as(C.POINTER(C.c_int)), _p(pnt), _p(_c(poses)),
                                         C.c_double(gba_voxel_size), C.c_double(gba_min_eigen_value), _p(_c(gba_eig))))
        return self.size()

    def hba_add_edge(self, clouds, poses, gba_voxel_size, gba_min_eigen_value, gba_eig, max_iter, thread_num, want_cloud=True):
        W = len(clouds)
        off, pnt = self._ragged(clouds)
        poses = _c(poses).copy()
        edges = np.zeros((W * (W - 1) // 2 + 1, 20)); ne = C.c_int(0)
        cloud = np.zeros((max(len(pnt), 1), 3)); ccnt = np.zeros(max(len(pnt), 1), dtype=np.int32); nc = C.c_int(0)
        rl = np.zeros((max_iter + 1, 2)); nl = C.c_int(0)
        self._chk(self.lib.vba_hba_add_edge(self.h, C.c_int(W), off.ctypes.data_as(C.POINTER(C.c_int)), _p(pnt), _p(poses),
                                            C.c_double(gba_voxel_size), C.c_double(gba_min_eigen_value), _p(_c(gba_eig)), C.c_int(max_iter),
                                            C.c_int(thread_num), _p(edges), C.byref(ne), _p(cloud) if want_cloud else None,
                                            ccnt.ctypes.data_as(C.POINTER(C.c_int)), C.byref(nc), _p(rl), C.byref(nl)))
        return dict(poses=poses, edges=edges[:ne.value].copy(), cloud=cloud[:nc.value].copy(), cloud_count=ccnt[:nc.value].copy(),
                    resis=rl[:nl.value].copy())

    def hba_global(self, clouds, poses_x0, poses_now, gba_voxel_size, gba_min_eigen_value, gba_eig, total_max_iter, wdsize=10, mgsize=5):
        if isinstance(clouds, tuple):                # (offsets[n + 1], points[N][3]) already concatenated by the caller
            off, pnt = clouds
            n = len(off) - 1
        else:
            n = len(clouds)
            off, pnt = self._ragged(clouds)
        nwin = max(0, (n - wdsize) // mgsize + 1) if n >= wdsize else 0
        cap1 = nwin * (wdsize * (wdsize - 1) // 2) + 1; cap2 = nwin * (nwin - 1) // 2 + 1
        e1 = np.zeros((cap1, 20)); e2 = np.zeros((cap2, 20)); n1 = C.c_int(0); n2 = C.c_int(0)
        self._chk(self.lib.vba_hba_global(self.h, C.c_int(n), off.ctypes.data_as(C.POINTER(C.c_int)), _p(pnt), _p(_c(poses_x0)), _p(_c(poses_now)),
                                          C.c_double(gba_voxel_size), C.c_double(gba_min_eigen_value), _p(_c(gba_eig)), C.c_int(total_max_iter),
                                          C.c_int(wdsize), C.c_int(mgsize), _p(e1), C.c_int(cap1), C.byref(n1), _p(e2), C.c_int(cap2), C.byref(n2)))
        return e1[:n1.value].copy(), e2[:n2.value].copy()

    def set_shard(self, rank, n_ranks):
        self._chk(self.lib.vba_set_shard(self.h, C.c_int(rank), C.c_int(n_ranks)))

    def set_allreduce(self, pyfunc):
        """pyfunc(dev_ptr:int, n_doubles:int, stream:int) -> int; kept alive on the context."""
        def tramp(user, buf, n, stream):
            try:
                return int(pyfunc(buf, n, stream) or 0)
            except Exception:   # noqa: BLE001 - must not unwind through C
                import traceback
                traceback.print_exc()
                return 1
        self._cb = ALLREDUCE_FN(tramp)
        self._chk(self.lib.vba_set_allreduce(self.h, self._cb, None))

    def rccl_init(self, dist, rank, world):
        """Native exchange step: rank 0 makes an ncclUniqueId, torch.distributed (any backend) carries its 128 bytes to the other
        ranks, every rank builds the library's own RCCL communicator (vba_rccl_init also applies the voxel-bucket shard)."""
        buf = C.create_string_buffer(128)
        if rank == 0:
            self._chk(self.lib.vba_rccl_get_unique_id(buf))
        box = [bytes(buf.raw)]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        uid = C.create_string_buffer(box[0], 128)
        self._chk(self.lib.vba_rccl_init(self.h, uid, C.c_int(rank), C.c_int(world)))

    def set_torch_allreduce(self, torch, dist):
        """torch.distributed SUM all-reduce (RCCL with backend "nccl", gloo in rehearsals) of the context's device buffers,
        ORDERED ON THE CONTEXT'S STREAM: the hook is handed the stream the kernels run on, and the collective is issued with
        that stream current.  (Issuing it on torch's default stream instead leaves it unordered against the context's own
        non-blocking stream: the sum then races with k_reduce_partials.)"""
        cache = {}
        streams = {}
        self.collective_calls = 0          # how many exchange steps the library asked for, and how many doubles they carried
        self.collective_doubles = 0

        def hook(ptr, n, stream):
            self.collective_calls += 1
            self.collective_doubles += int(n)
            key = (ptr, n)
            if key not in cache:
                class _Ext:
                    __cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2, "strides": None}
                cache[key] = torch.as_tensor(_Ext(), device="cuda")
            skey = int(stream) if stream else 0
            if skey not in streams:                         # (the hook runs once per LM iteration: keep its host cost down)
                streams[skey] = torch.cuda.ExternalStream(skey) if skey else torch.cuda.default_stream()
            ext = streams[skey]
            if torch.cuda.current_stream() == ext:
                dist.all_reduce(cache[key], op=dist.ReduceOp.SUM)
            else:
                with torch.cuda.stream(ext):
                    dist.all_reduce(cache[key], op=dist.ReduceOp.SUM)
            return 0
        self.set_allreduce(hook)

    def timing_enable(self, on=True):
        self.lib.vba_timing_enable(self.h, C.c_int(int(on)))

    def timing_calibration_read(self, n_bytes):
        self._chk(self.lib.vba_timing_calibration_read(self.h, C.c_size_t(n_bytes)))

    def timing_select(self, name=None):
        self.lib.vba_timing_select(self.h, name.encode() if name else None)

    def timing_sample_every(self, n=1):
        self.lib.vba_timing_sample_every(self.h, C.c_int(int(n)))

    def timing_launch_hessian(self):
        self._chk(self.lib.vba_timing_launch_hessian(self.h))

    def timing_null_spans(self, n=64):
        """Average duration in microseconds of an event pair that brackets nothing."""
        for _ in range(n):
            self.lib.vba_timing_null_span(self.h)
        t, k = self.timing_get("null")
        return t / max(k, 1)

    def timing_reset(self):
        self.lib.vba_timing_reset(self.h)

    def timing_get(self, name):
        tot = C.c_double(); cnt = C.c_int()
        self.lib.vba_timing_get(self.h, name.encode(), C.byref(tot), C.byref(cnt))
        return tot.value, cnt.value
