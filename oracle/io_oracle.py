"""TEST INFRASTRUCTURE — numpy restatement of the session-store formats either side of the path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.

  save_pcd          voxelslam.cpp:166-179  pcl::io::savePCDFileBinary(PointCloud<PointXYZI>)
  save_pose         voxelslam.cpp:181-204  `fixed << setprecision(6) << t`, then setprecision(7)
  read_lidarstate   voxelslam.hpp:268-307  Quaterniond(w, x, y, z).matrix()

Parity unpinned against PCL / Eigen themselves (neither is in this image): the PCD v0.7 header PCL's PCDWriter emits for
PointXYZI and Eigen's matrix<->quaternion conversions are restated from their published definitions; the quaternion
conversion is cross-checked against scipy in tests/test_io_cpu.py.
"""
import numpy as np


def pcd_bytes(xyz):
    xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
    n = len(xyz)
    head = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\n"
            "COUNT 1 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary\n" % (n, n))
    rec = np.zeros((n, 4), dtype="<f4")
    rec[:, :3] = xyz.astype(np.float32)          # ap.x = pw.pnt[0] ... ; intensity keeps the PointXYZI default 0
    return head.encode("ascii") + rec.tobytes()


def quat_from_rot(R):
    """Eigen::Quaterniond(Matrix3d) -> (x, y, z, w)."""
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    q = np.zeros(4)
    t = np.trace(R)
    if t > 0:
        t = np.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0] = (R[2, 1] - R[1, 2]) * t; q[1] = (R[0, 2] - R[2, 0]) * t; q[2] = (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j = (i + 1) % 3; k = (j + 1) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (R[k, j] - R[j, k]) * t
        q[j] = (R[j, i] + R[i, j]) * t
        q[k] = (R[k, i] + R[i, k]) * t
    return q


def rot_from_quat(q):
    """Quaterniond(w, x, y, z).matrix() with q = (x, y, z, w); the coefficients are used as stored."""
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def pose_text(states, v6):
    states = np.asarray(states, dtype=np.float64).reshape(-1, 25); v6 = np.asarray(v6, dtype=np.float64).reshape(-1, 6)
    if len(states) < 100:                          # VS:183-184
        return None
    out = []
    for s, w in zip(states, v6):
        q = quat_from_rot(s[1:10])
        nums = list(s[10:13]) + list(q) + list(s[13:25]) + list(w)
        out.append("%.6f " % s[0] + " ".join("%.7f" % v for v in nums[:3]) + " " + " ".join("%.7f" % v for v in nums[3:]) + "\n")
    return "".join(out)


def read_lidarstate(text):
    states, v6 = [], []
    for line in text.splitlines():
        nums = [float(t) for t in line.split(" ") if t != ""]
        if not nums:
            continue
        s = np.zeros(25); w = np.zeros(6)
        s[24] = -9.8
        s[0] = nums[0]; s[10:13] = nums[1:4]
        s[1:10] = rot_from_quat(nums[4:8]).ravel()
        if len(nums) >= 20:
            s[13:25] = nums[8:20]
        if len(nums) >= 26:
            w[:] = nums[20:26]
        states.append(s); v6.append(w)
    return np.array(states).reshape(-1, 25), np.array(v6).reshape(-1, 6)
