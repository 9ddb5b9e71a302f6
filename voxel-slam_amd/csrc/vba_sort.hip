// Stable device-wide radix sort of (leaf id, point index) pairs — the grouping step of the order-preserving insertion
// (vba_kernels_map.hpp, "order-preserving accumulation").  rocPRIM's radix sort is stable and deterministic; it lives in a
// translation unit of its own so that the kernels of voxelba.hip do not pay its template instantiation time.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

namespace vba {

// tmp == nullptr: only tmp_bytes is written (size query).  Keys are compared on bits [0, end_bit).
hipError_t sort_pairs_u32(void *tmp, size_t &tmp_bytes, const unsigned int *keys_in, unsigned int *keys_out, const int *vals_in, int *vals_out,
                          size_t n, unsigned int end_bit, hipStream_t stream) {
  // rocPRIM's default switches to a merge sort (block sort + 8 merge launches at 2e5 pairs: 84 us on MI355X) below 2^20 items; the
  // Onesweep radix path (histogram + scan + one launch per 8-bit digit) is the faster one for a 200k-point scan
  using config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 8192>;
  return rocprim::radix_sort_pairs<config>(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, end_bit, stream, false);
}

}  // namespace vba
