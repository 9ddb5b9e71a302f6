"""Host-side check of the product's direct symmetric 3x3 eigen-solver (csrc/vba_eig3.hpp, the plane fit of K2/K4/K5) against
numpy.linalg.eigh: the KAT-3 fixture, plane-like covariances with world-sized second moments, and the cases that must be
handed to the iterative solver.  The same header is compiled for the device by hipcc; here g++ builds it for the CPU."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def eig():
    out = os.path.join(tempfile.mkdtemp(prefix="vba_eig3_"), "libeig3host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-o", out, os.path.join(HERE, "host", "eig3_host.cpp")])
    lib = C.CDLL(out)
    dp = C.POINTER(C.c_double)

    def f(A):
        a = np.array([A[0, 0], A[1, 0], A[2, 0], A[1, 1], A[2, 1], A[2, 2]], dtype=np.float64)
        w = np.zeros(3); V = np.zeros(9)
        ok = lib.eig3_direct_host(a.ctypes.data_as(dp), w.ctypes.data_as(dp), V.ctypes.data_as(dp))
        return bool(ok), w, V.reshape(3, 3)
    return f


def _check(A, ok, w, V, tol=4e-15):
    L = np.tril(A) + np.tril(A, -1).T
    w_ref = np.linalg.eigvalsh(L)
    scale = max(np.abs(w_ref).max(), 1e-300)
    assert np.all(np.diff(w) >= 0)
    assert np.abs(w - w_ref).max() <= tol * scale + 1e-18, (w, w_ref)
    assert np.abs(V.T @ V - np.eye(3)).max() < 1e-14
    assert np.abs(V @ np.diag(w) @ V.T - L).max() <= 10 * tol * scale


def test_kat3_fixture(eig):
    d = np.load(os.path.join(HERE, "golden", "kat_eig3.npz"))
    n_direct = 0
    for A in d["A"]:
        ok, w, V = eig(A)
        if ok:
            n_direct += 1
            # (the fixture holds near-equal eigenvalue pairs down to the fallback threshold: eigenvalues there are good to
            #  eps * scale / (relative gap), which the 1e-5 threshold bounds by ~1e-11)
            _check(A, ok, w, V, tol=2e-11)
    assert n_direct > 150


def test_plane_covariances(eig):
    """cov = P/N - c c^T of noisy planar patches tens of metres from the origin (the matrices K4 sees)."""
    rng = np.random.default_rng(11)
    worst = 0.0
    for k in range(3000):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        t1 = np.cross(n, rng.normal(size=3)); t1 /= np.linalg.norm(t1); t2 = np.cross(n, t1)
        m = int(rng.integers(6, 400))
        ext = rng.uniform(0.02, 0.3, 2)
        pts = rng.uniform(-40, 40, 3) + np.outer(rng.uniform(-1, 1, m) * ext[0], t1) + np.outer(rng.uniform(-1, 1, m) * ext[1], t2) + np.outer(rng.normal(0, 0.01, m), n)
        c = pts.mean(0)
        A = pts.T @ pts / m - np.outer(c, c)
        ok, w, V = eig(A)
        if not ok:
            continue
        w_ref, V_ref = np.linalg.eigh(A)
        scale = np.abs(w_ref).max()
        worst = max(worst, np.abs(w - w_ref).max() / scale)
        assert np.abs(w - w_ref).max() <= 1e-12 * scale
        assert np.abs(V.T @ V - np.eye(3)).max() < 1e-14
        assert np.abs(A @ V - V * w).max() <= 1e-12 * scale
        if (w_ref[1] - w_ref[0]) > 0.05 * scale:
            assert abs(abs(V[:, 0] @ V_ref[:, 0]) - 1) < 1e-10       # plane normal
    assert worst < 1e-12


def test_general_and_scaled(eig):
    rng = np.random.default_rng(12)
    for k in range(2000):
        Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        w = np.sort(rng.uniform(-1, 1, 3)) * 10.0 ** rng.integers(-200, 200)
        A = Q @ np.diag(w) @ Q.T
        A = 0.5 * (A + A.T)
        ok, w2, V = eig(A)
        if ok:
            _check(A, ok, w2, V, tol=1e-10 if min(w[1] - w[0], w[2] - w[1]) < 1e-3 * np.abs(w).max() else 1e-13)


def test_degenerate_inputs_are_refused(eig):
    """Zero, multiples of the identity and double eigenvalues go to the iterative solver (return value false)."""
    assert not eig(np.zeros((3, 3)))[0]
    assert not eig(2.5 * np.eye(3))[0]
    Q, _ = np.linalg.qr(np.random.default_rng(3).normal(size=(3, 3)))
    assert not eig(Q @ np.diag([1.0, 1.0, 3.0]) @ Q.T)[0]
    assert not eig(Q @ np.diag([1.0, 3.0, 3.0 + 1e-9]) @ Q.T)[0]
    assert not eig(np.full((3, 3), np.nan))[0]
    # exactly diagonal with distinct entries is fine
    ok, w, V = eig(np.diag([3.0, 1.0, 2.0]))
    assert ok and np.allclose(w, [1, 2, 3]) and np.allclose(np.abs(V), np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0]]), atol=1e-15)
