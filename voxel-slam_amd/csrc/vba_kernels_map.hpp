// Device voxel hash map + octree (K1 insert, K2 recut/plane fit, K5 marginalise).  PLACEHOLDER interface:
// the full implementation follows in this round; until then every map entry reports "not implemented".
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include "../../include/voxelba.h"
#include "vba_kernels_factor.hpp"

namespace vba {

// 16-bit bucket of a root voxel key; ranks own contiguous bucket ranges
__host__ __device__ inline uint64_t shard_bucket(int64_t kx, int64_t ky, int64_t kz) {
  uint64_t h = (uint64_t)kx * 0x9E3779B97F4A7C15ull;
  h ^= (uint64_t)ky * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
  h ^= (uint64_t)kz * 0x165667B19E3779F9ull + (h << 6) + (h >> 2);
  h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
  return h & 0xFFFFull;
}

struct MapStore {
  vba_options opt;
  int rank = 0, n_ranks = 1;
};

inline void map_init(MapStore &m, const vba_options &o) { m.opt = o; }
inline void map_free(MapStore &) {}
inline int map_ni(std::string &err) { err = "voxel map level not implemented yet"; return VBA_ERR_BAD_ARG; }
inline int map_cut_voxel(MapStore &, hipStream_t, int, int, const double *, const double *, const double *, bool, std::string &e) { return map_ni(e); }
inline int map_cut_voxel_fix(MapStore &, hipStream_t, int, const double *, double, std::string &e) { return map_ni(e); }
inline int map_recut(MapStore &, hipStream_t, int, const double *, bool, std::string &e, int *) { return map_ni(e); }
inline int map_extract_factors(MapStore &, hipStream_t, FactorView, std::string &e, int *) { return map_ni(e); }
inline int map_margi(MapStore &, hipStream_t, int, const double *, FactorView, int, std::string &e) { return map_ni(e); }
inline int map_slide(MapStore &, int) { return VBA_ERR_BAD_ARG; }
inline int map_reset(MapStore &, hipStream_t, std::string &e) { return map_ni(e); }
inline int map_num_roots(MapStore &, hipStream_t, bool) { return 0; }
inline int map_dump_leaves(MapStore &, hipStream_t, double *, int, std::string &) { return 0; }

}  // namespace vba
