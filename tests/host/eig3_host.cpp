// Host build of the product's direct 3x3 eigen-solver (voxel-slam_amd/csrc/vba_eig3.hpp) for CPU-side checks against
// numpy.linalg.eigh (tests/test_eig3_cpu.py).  Test harness only: the product runs this code on the device.
#include "../../voxel-slam_amd/csrc/vba_eig3.hpp"
extern "C" int eig3_direct_host(const double *A6, double *w, double *V) {
  return vba::eig3_direct(A6[0], A6[1], A6[2], A6[3], A6[4], A6[5], w[0], w[1], w[2], V) ? 1 : 0;
}
