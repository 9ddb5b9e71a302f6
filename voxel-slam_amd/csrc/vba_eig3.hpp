// Symmetric 3x3 eigen-decomposition of the plane fit (replaces Eigen::SelfAdjointEigenSolver<Matrix3d> at
// voxel_map.hpp:312 / :1416 / :1525 / :1771, loop_refine.hpp:363): ascending eigenvalues, orthonormal eigenvectors in the
// columns of V (row-major), result equal to the iterative solver up to rounding and eigenvector sign.
//
// Why not Jacobi: in the residual pass (K4) every lane owns one voxel and the eigen-solve is ONE dependent chain per lane; the
// cyclic-Jacobi version (f32 pre-pass + two f64 sweeps, vba_kernels_factor.hpp) measured 8.5k of the pass's 22k cycles on
// MI355X.  This solver is direct (no sweeps), ~330 f64 operations with short chains:
//   1. exact power-of-two scaling, B = A - (tr A / 3) I, characteristic cubic x^3 - c1 x - c0 (c1 = |B|_F^2 / 2, c0 = det B);
//   2. the ISOLATED root (largest if c0 >= 0, else smallest: its distance to the other two is >= sqrt(3 c1 / 3)...) from an f32
//      trigonometric seed + three f64 Newton steps (two with the raw reciprocal, one exact); the other two roots from the deflated quadratic
//      x^2 + x_a x + (x_a^2 - c1) = 0, whose discriminant is (x_b - x_c)^2;
//   3. eigenvectors without iteration (Eberly, "A Robust Eigensolver for 3x3 Symmetric Matrices"): v_a = largest cross product
//      of two rows of B - x_a I; {U, V} an orthonormal basis of its complement; the eigenvector of the OTHER EXTREME eigenvalue
//      as the null vector of the 2x2 matrix [U V]^T (B - x_b I) [U V]; the middle one as a cross product.
// When the two non-isolated eigenvalues are closer than ~1e-5 of the matrix scale the deflated discriminant loses digits
// (the eigenvalues would still be good to ~1e-11 of the scale) and the caller falls back to the Jacobi solver; for planar
// voxels this concerns ~1e-5 of the matrices.  Compiles for the host too (tests/test_eig3_cpu.py checks it against
// numpy.linalg.eigh on the KAT-3 fixture without a GPU).
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VBE_HD __host__ __device__ __forceinline__
#else
#define VBE_HD inline
#endif

namespace vba {

// reciprocal square root to full f64 precision: v_rsq_f64 (~26 bits) + two Newton steps on the device
VBE_HD double eig_rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double c = __builtin_amdgcn_rsq(x);
  c = c * (1.5 - 0.5 * x * c * c);
  c = c * (1.5 - 0.5 * x * c * c);
  return c;
#else
  return 1.0 / std::sqrt(x);
#endif
}
// reciprocal good to ~26 bits: enough inside a Newton iteration that corrects itself
VBE_HD double eig_rcp_approx(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcp(x);
#else
  return 1.0 / x;
#endif
}
VBE_HD float eig_cos_f32(float x) {   // cos(x), |x| <= pi/3
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_cosf(x * 0.15915494309189535f);   // v_cos_f32 takes revolutions
#else
  return std::cos(x);
#endif
}
VBE_HD float eig_sqrt_f32(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sqrtf(x);
#else
  return std::sqrt(x);
#endif
}
VBE_HD float eig_rcp_f32(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}

// in: a00 a01 a02 a11 a12 a22 (the lower triangle, as Eigen reads it).  out: w0 <= w1 <= w2, V row-major with the eigenvectors in
// its columns.  Returns false when the matrix needs the iterative solver (zero / non-finite / A = qI / a near-double pair).
struct Eig3 { double w0, w1, w2, v00, v01, v02, v10, v11, v12, v20, v21, v22; };   // V row-major: column c = eigenvector c

// (results in a struct of scalars, not through a pointer to the caller's array: merged with the fallback's results they stay in
//  registers, whereas stores through the shared pointer were turned into a run-time indexed private array, i.e. scratch memory)
VBE_HD bool eig3_direct(double a00, double a01, double a02, double a11, double a12, double a22, Eig3 &o) {
  const double s = fmax(fmax(fmax(fabs(a00), fabs(a11)), fabs(a22)), fmax(fmax(fabs(a01), fabs(a02)), fabs(a12)));
  if (!(s > 1e-290 && s < 1e290)) return false;
  int e;
  (void)frexp(s, &e);
  const double sc = ldexp(1.0, -e);                          // exact: the largest entry lands in [0.5, 1)
  a00 *= sc; a01 *= sc; a02 *= sc; a11 *= sc; a12 *= sc; a22 *= sc;
  const double q = (a00 + a11 + a22) * (1.0 / 3.0);
  const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
  const double c1 = 0.5 * (b00 * b00 + b11 * b11 + b22 * b22) + (a01 * a01 + a02 * a02 + a12 * a12);
  if (!(c1 > 1e-20)) return false;                           // A = q I to ~1e-10 of its scale
  const double c0 = b00 * (b11 * b22 - a12 * a12) - a01 * (a01 * b22 - a12 * a02) + a02 * (a01 * a12 - b11 * a02);
  // f32 seed of the isolated root: x = 2 m cos(acos(r) / 3), m = sqrt(c1 / 3), r = |c0| / (2 m^3)  (the largest root of the cubic
  // with c0 replaced by |c0|; for c0 < 0 the smallest root is its negative)
  const float c1f = (float)c1, c0f = fabsf((float)c0);
  const float m = eig_sqrt_f32(c1f * (1.0f / 3.0f));
  float r = c0f * eig_rcp_f32(2.0f * m * m * m);
  r = fminf(r, 1.0f);
  // acos(r), 0 <= r <= 1: sqrt(1 - r) * P(r)   (Abramowitz & Stegun 4.4.45, |error| <= 7e-5)
  const float ac = eig_sqrt_f32(1.0f - r) * (1.5707288f + r * (-0.2121144f + r * (0.0742610f - 0.0187293f * r)));
  double xa = (double)(2.0f * m * eig_cos_f32(ac * (1.0f / 3.0f)));
  const double c0a = fabs(c0);
  // Newton on f(x) = (x^2 - c1) x - |c0|, f' = 3 x^2 - c1 >= 2 c1 at the isolated root (x_a >= sqrt(c1))
#pragma unroll
  for (int it = 0; it < 2; it++) {      // seed good to ~1e-4 x_a: 1e-4 -> ~1e-8 -> ~1e-15 (each step also inherits 2^-26 of the previous error)
    const double x2 = xa * xa;
    const double f = (x2 - c1) * xa - c0a;
    const double fp = 3.0 * x2 - c1;
    xa -= f * eig_rcp_approx(fp);
  }
  {   // one last step with an accurate quotient: f / fp by one Newton refinement of the reciprocal
    const double x2 = xa * xa;
    const double f = (x2 - c1) * xa - c0a;
    const double fp = 3.0 * x2 - c1;
    double ri = eig_rcp_approx(fp);
    ri = ri * (2.0 - fp * ri);
    xa -= f * ri;
  }
  // deflation: the other two roots of the |c0| cubic are (-x_a -+ sqrt(D)) / 2 with D = 4 c1 - 3 x_a^2 = (x_b - x_c)^2
  const double D = 4.0 * c1 - 3.0 * xa * xa;
  if (!(D > 1e-10 * c1)) return false;                       // near-double pair (or NaN): iterative solver
  const double sD = D * eig_rsqrt(D);
  const bool neg = c0 < 0.0;
  // roots of the actual cubic, ascending
  double x0, x1, x2r;
  if (!neg) { x2r = xa; x0 = 0.5 * (-xa - sD); x1 = 0.5 * (-xa + sD); }
  else { x0 = -xa; x1 = 0.5 * (xa - sD); x2r = 0.5 * (xa + sD); }
  const double xiso = neg ? x0 : x2r;                        // isolated root: eigenvector from cross products
  const double xoth = neg ? x2r : x0;                        // the other extreme: eigenvector from the 2x2 complement problem
  // v_a: best cross product of two rows of B - x_iso I
  const double r00 = b00 - xiso, r11 = b11 - xiso, r22 = b22 - xiso;
  const double p0x = a01 * a12 - a02 * r11, p0y = a02 * a01 - r00 * a12, p0z = r00 * r11 - a01 * a01;   // row0 x row1
  const double p1x = a01 * r22 - a02 * a12, p1y = a02 * a02 - r00 * r22, p1z = r00 * a12 - a01 * a02;   // row0 x row2
  const double p2x = r11 * r22 - a12 * a12, p2y = a12 * a02 - a01 * r22, p2z = a01 * a12 - r11 * a02;   // row1 x row2
  const double n0 = p0x * p0x + p0y * p0y + p0z * p0z, n1 = p1x * p1x + p1y * p1y + p1z * p1z, n2 = p2x * p2x + p2y * p2y + p2z * p2z;
  double ax = p0x, ay = p0y, az = p0z, an = n0;
  if (n1 > an) { ax = p1x; ay = p1y; az = p1z; an = n1; }
  if (n2 > an) { ax = p2x; ay = p2y; az = p2z; an = n2; }
  if (!(an > 0.0)) return false;
  { const double ri = eig_rsqrt(an); ax *= ri; ay *= ri; az *= ri; }
  // orthonormal basis {U, W} of the complement of v_a
  double ux, uy, uz;
  if (fabs(ax) > fabs(ay)) { const double ri = eig_rsqrt(ax * ax + az * az); ux = -az * ri; uy = 0.0; uz = ax * ri; }
  else { const double ri = eig_rsqrt(ay * ay + az * az); ux = 0.0; uy = az * ri; uz = -ay * ri; }
  const double wx = ay * uz - az * uy, wy = az * ux - ax * uz, wz = ax * uy - ay * ux;
  // M = [U W]^T (B - x_oth I) [U W]
  const double bux = b00 * ux + a01 * uy + a02 * uz, buy = a01 * ux + b11 * uy + a12 * uz, buz = a02 * ux + a12 * uy + b22 * uz;
  const double bwx = b00 * wx + a01 * wy + a02 * wz, bwy = a01 * wx + b11 * wy + a12 * wz, bwz = a02 * wx + a12 * wy + b22 * wz;
  const double m00 = (ux * bux + uy * buy + uz * buz) - xoth;
  const double m01 = ux * bwx + uy * bwy + uz * bwz;
  const double m11 = (wx * bwx + wy * bwy + wz * bwz) - xoth;
  // null vector of M from its larger row (p, q): (q, -p)
  const bool first = fabs(m00) + fabs(m01) >= fabs(m01) + fabs(m11);
  const double pp = first ? m00 : m01, qq = first ? m01 : m11;
  const double nn = pp * pp + qq * qq;
  double cu = 1.0, cw = 0.0;
  if (nn > 0.0) { const double ri = eig_rsqrt(nn); cu = qq * ri; cw = -pp * ri; }
  const double bx = cu * ux + cw * wx, by = cu * uy + cw * wy, bz = cu * uz + cw * wz;
  // middle eigenvector
  const double mx = ay * bz - az * by, my = az * bx - ax * bz, mz = ax * by - ay * bx;
  const double un = ldexp(1.0, e);
  o.w0 = (q + x0) * un; o.w1 = (q + x1) * un; o.w2 = (q + x2r) * un;
  o.v00 = neg ? ax : bx; o.v10 = neg ? ay : by; o.v20 = neg ? az : bz;
  o.v02 = neg ? bx : ax; o.v12 = neg ? by : ay; o.v22 = neg ? bz : az;
  o.v01 = mx; o.v11 = my; o.v21 = mz;
  return true;
}

VBE_HD bool eig3_direct(double a00, double a01, double a02, double a11, double a12, double a22,
                        double &w0, double &w1, double &w2, double *V) {
  Eig3 o;
  if (!eig3_direct(a00, a01, a02, a11, a12, a22, o)) return false;
  w0 = o.w0; w1 = o.w1; w2 = o.w2;
  V[0] = o.v00; V[1] = o.v01; V[2] = o.v02; V[3] = o.v10; V[4] = o.v11; V[5] = o.v12; V[6] = o.v20; V[7] = o.v21; V[8] = o.v22;
  return true;
}

}  // namespace vba
