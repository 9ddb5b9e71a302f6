"""Session-store formats: <session>/<i>.pcd (voxelslam.cpp:166-179, 337-340) and alidarState.txt
(voxelslam.cpp:181-204, voxelslam.hpp:268-307).  Host-only entry points of libvoxelba.so against oracle/io_oracle.py and the
hand-written fixtures under tests/golden/.  No GPU."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLD = os.path.join(ROOT, "tests", "golden")


def _capi():
    import voxel_slam_amd  # noqa: F401
    from voxel_slam_amd import capi
    capi.build()
    return capi


def _states(n, seed):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(seed)
    s = np.zeros((n, 25))
    s[:, 0] = 1.7e9 + np.arange(n) * 0.1 + rng.uniform(0, 0.01, n)
    rv = rng.normal(0, 1.5, (n, 3))
    rv[1] = [np.pi - 1e-3, 0, 0]; rv[2] = [0, np.pi - 1e-3, 0]; rv[3] = [0, 0, np.pi - 1e-3]; rv[4] = 0      # every branch of the conversion
    s[:, 1:10] = Rotation.from_rotvec(rv).as_matrix().reshape(n, 9)
    s[:, 10:13] = rng.normal(0, 30, (n, 3)); s[:, 13:16] = rng.normal(0, 2, (n, 3))
    s[:, 16:22] = rng.normal(0, 0.01, (n, 6)); s[:, 22:25] = [0, 0, -9.8]
    v6 = np.abs(rng.normal(0, 1e-4, (n, 6))) + 1e-6
    return s, v6


def test_pcd_written_bytes_and_round_trip(tmp_path):
    import io_oracle
    capi = _capi()
    rng = np.random.default_rng(1)
    for n in (0, 1, 4097):
        xyz = rng.normal(0, 40, (n, 3))
        path = str(tmp_path / ("%d.pcd" % n))
        capi.save_pcd(path, xyz)
        assert open(path, "rb").read() == io_oracle.pcd_bytes(xyz)                  # byte-identical file
        got, inten = capi.load_pcd(path)
        np.testing.assert_array_equal(got, xyz.astype(np.float32).astype(np.float64))
        assert got.shape == (n, 3) and (inten == 0).all()


def test_pcd_reader_other_layouts(tmp_path):
    """loadPCDFile accepts any field set: ascii data, extra fields, f64 coordinates, intensity before xyz."""
    capi = _capi()
    got, inten = capi.load_pcd(os.path.join(GOLD, "scan_ascii.pcd"))
    np.testing.assert_array_equal(got, [[1.5, -2.25, 0.125], [100.0, 0.0, -7.5], [0.001, 0.002, 0.003]])
    np.testing.assert_array_equal(inten, [10, 20, 255])
    # binary, fields: intensity(F4) x y z (F8) ring(U2) normal(F4 x3)
    rec = np.zeros(3, dtype=[("i", "<f4"), ("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("ring", "<u2"), ("nrm", "<f4", 3)])
    rec["i"] = [1, 2, 3]; rec["x"] = [0.1, 0.2, 0.3]; rec["y"] = [-1, -2, -3]; rec["z"] = [7, 8, 9]; rec["ring"] = [5, 6, 7]
    head = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS intensity x y z ring normal\nSIZE 4 8 8 8 2 4\nTYPE F F F F U F\n"
            "COUNT 1 1 1 1 1 3\nWIDTH 3\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 3\nDATA binary\n")
    p = tmp_path / "mixed.pcd"
    p.write_bytes(head.encode() + rec.tobytes())
    got, inten = capi.load_pcd(str(p))
    np.testing.assert_array_equal(got, np.stack([rec["x"], rec["y"], rec["z"]], 1))
    np.testing.assert_array_equal(inten, [1, 2, 3])


def test_pcd_errors(tmp_path):
    capi = _capi()
    with pytest.raises(capi.VbaError) as e:
        capi.load_pcd(str(tmp_path / "absent.pcd"))
    assert e.value.status == 8
    p = tmp_path / "trunc.pcd"
    import io_oracle
    p.write_bytes(io_oracle.pcd_bytes(np.ones((10, 3)))[:-5])                      # truncated payload
    with pytest.raises(capi.VbaError):
        capi.load_pcd(str(p))
    p = tmp_path / "comp.pcd"
    p.write_bytes(io_oracle.pcd_bytes(np.ones((2, 3))).replace(b"DATA binary", b"DATA binary_compressed"))
    with pytest.raises(capi.VbaError):
        capi.load_pcd(str(p))


def test_quaternion_restatement_against_scipy():
    import io_oracle
    from scipy.spatial.transform import Rotation
    s, _ = _states(200, 3)
    for r in s[:, 1:10]:
        q = io_oracle.quat_from_rot(r)
        want = Rotation.from_matrix(r.reshape(3, 3)).as_quat()
        assert min(np.abs(q - want).max(), np.abs(q + want).max()) < 1e-12
        np.testing.assert_allclose(io_oracle.rot_from_quat(q), r.reshape(3, 3), atol=1e-12)


def test_pose_file_text_and_round_trip(tmp_path):
    import io_oracle
    capi = _capi()
    s, v6 = _states(120, 5)
    path = str(tmp_path / "alidarState.txt")
    capi.save_pose(path, s, v6)
    text = open(path).read()
    assert text == io_oracle.pose_text(s, v6)                                       # character-identical file
    assert len(text.splitlines()) == 120 and all(len(l.split(" ")) == 26 for l in text.splitlines())
    g_s, g_v = capi.read_lidarstate(path)
    o_s, o_v = io_oracle.read_lidarstate(text)
    np.testing.assert_array_equal(g_s, o_s)
    np.testing.assert_array_equal(g_v, o_v)
    assert np.abs(g_s[:, 10:25] - s[:, 10:25]).max() <= 0.5e-7 + 1e-12              # 7 decimals
    assert np.abs(g_s[:, 1:10] - s[:, 1:10]).max() < 1e-6
    # fewer than 100 scans: nothing is written (VS:183-184)
    short = str(tmp_path / "short.txt")
    capi.save_pose(short, s[:99], v6[:99])
    assert not os.path.exists(short) and io_oracle.pose_text(s[:99], v6[:99]) is None


def test_read_lidarstate_golden_and_short_lines():
    """tests/golden/alidarState_small.txt: hand-written lines with 8, 20 and 26 columns (VH:287-304)."""
    import io_oracle
    capi = _capi()
    path = os.path.join(GOLD, "alidarState_small.txt")
    g_s, g_v = capi.read_lidarstate(path)
    o_s, o_v = io_oracle.read_lidarstate(open(path).read())
    np.testing.assert_array_equal(g_s, o_s)
    np.testing.assert_array_equal(g_v, o_v)
    assert g_s.shape == (3, 25)
    np.testing.assert_array_equal(g_s[0, 1:10], np.eye(3).ravel())                  # identity quaternion
    np.testing.assert_allclose(g_s[1, 1:10].reshape(3, 3), [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-7)   # 90 deg yaw
    np.testing.assert_array_equal(g_s[1, 13:16], [1.0, 2.0, 3.0])
    np.testing.assert_array_equal(g_v[2], [1e-6] * 6)
    np.testing.assert_array_equal(g_v[0], 0)
    with pytest.raises(capi.VbaError) as e:
        capi.read_lidarstate(os.path.join(GOLD, "no_such_file.txt"))
    assert e.value.status == 8


def test_adapter_header_compiles_and_io_wrappers_run(tmp_path):
    """include/voxelba_adapter.hpp (the reference-side binding of INTEGRATION.md) compiles as plain C++17 against
    libvoxelba.so; its host-only session-store wrappers round-trip a scan and a pose file (no device call is made)."""
    import subprocess
    capi = _capi()
    src = tmp_path / "adapter_check.cpp"
    src.write_text(r'''
#include "voxelba_adapter.hpp"
#include <cmath>
#include <cstdio>
int main(int argc, char **argv) {
  const std::string dir = argv[1];
  std::vector<vba::pointVar> pv(1000);
  for (size_t i = 0; i < pv.size(); i++) { pv[i].pnt[0] = 0.25 * i; pv[i].pnt[1] = -1.5 * i; pv[i].pnt[2] = 3.0; }
  vba::save_pcd(pv, 7, dir);
  std::vector<vba::XYZ> back = vba::load_pcd(dir + "/7.pcd");
  if (back.size() != pv.size() || back[999].x != (float)(0.25 * 999) || back[999].y != (float)(-1.5 * 999)) return 2;
  std::vector<vba::ScanPoseRec> bb(120);
  for (size_t i = 0; i < bb.size(); i++) { bb[i].x.t = 100.0 + i; bb[i].x.p[0] = 0.5 * i; bb[i].x.g[2] = -9.8; for (int k = 0; k < 6; k++) bb[i].v6[k] = 1e-6; }
  vba::save_pose(bb, dir + "/alidarState.txt");
  std::vector<vba::ScanPoseRec> rd = vba::read_lidarstate(dir + "/alidarState.txt");
  if (rd.size() != 120 || std::fabs(rd[119].x.p[0] - 59.5) > 1e-7 || rd[5].x.R[0] != 1.0 || rd[3].v6[5] != 1e-6 || rd[0].x.cov[0] != 1e-4) return 3;
  bool threw = false;
  try { vba::read_lidarstate(dir + "/absent.txt"); } catch (const std::runtime_error &) { threw = true; }
  if (!threw) return 4;
  // the device-side wrappers only have to compile here
  int (*fn)(vba::Context &, const std::vector<vba::pointVar> &, vba::IMUST &) = &vba::lio_state_estimation_kdtree;
  std::printf("ok %p\n", (void *)fn);
  return 0;
}
''')
    exe = tmp_path / "adapter_check"
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lvoxelba", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
