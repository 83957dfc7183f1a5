// Shared helpers of the C-ABI translation units (libvoltrix_hip.so).  See include/voltrix_capi.h.
#pragma once

#include <hip/hip_runtime.h>

#include "voltrix/spmm_kernels.hpp"
#include "voltrix_capi.h"

namespace voltrix_capi {

// Tile space instantiated ahead of time.  A combination is valid when the workgroup's LDS fits the CU (160 KiB) and
// the counted vmcnt wait fits its 6-bit immediate (traits.hpp static_asserts).
template <int FS, int D, int W, int EB>
constexpr bool tile_ok() {
  const int stage = 32 * FS * EB;
  const int wave_lds = D * stage + (2 * D + 1) * 256;
  return (long long)W * wave_lds <= 160 * 1024 && (1 + stage / 1024) * (D - 1) <= 63 && !(EB == 4 && FS == 256);
}

struct TileId {
  int fs, depth, waves;
};

// X(FS, DEPTH, WAVES) over the ahead-of-time space
#define VOLTRIX_TILE_SPACE(X)                                                                          \
  X(32, 2, 1) X(32, 2, 2) X(32, 2, 4) X(32, 3, 1) X(32, 3, 2) X(32, 3, 4) X(32, 4, 1) X(32, 4, 2) X(32, 4, 4)       \
  X(64, 2, 1) X(64, 2, 2) X(64, 2, 4) X(64, 3, 1) X(64, 3, 2) X(64, 3, 4) X(64, 4, 1) X(64, 4, 2) X(64, 4, 4)       \
  X(128, 2, 1) X(128, 2, 2) X(128, 2, 4) X(128, 3, 1) X(128, 3, 2) X(128, 3, 4) X(128, 4, 1) X(128, 4, 2) X(128, 4, 4) \
  X(256, 2, 1) X(256, 2, 2) X(256, 2, 4) X(256, 3, 1) X(256, 3, 2) X(256, 3, 4) X(256, 4, 1) X(256, 4, 2) X(256, 4, 4) \
  X(32, 2, 8) X(32, 3, 8) X(32, 4, 8) X(64, 2, 8) X(64, 3, 8) X(64, 4, 8) X(128, 2, 8) X(128, 3, 8) X(128, 4, 8)

template <int EB>
inline int num_tiles() {
  int n = 0;
#define X(FS, D, W) n += tile_ok<FS, D, W, EB>() ? 1 : 0;
  VOLTRIX_TILE_SPACE(X)
#undef X
  return n;
}

template <int EB>
inline bool tile_at(int index, TileId* out) {
  int n = 0;
#define X(FS, D, W)                      \
  if (tile_ok<FS, D, W, EB>()) {         \
    if (n == index) {                    \
      *out = TileId{FS, D, W};           \
      return true;                       \
    }                                    \
    ++n;                                 \
  }
  VOLTRIX_TILE_SPACE(X)
#undef X
  return false;
}

template <int FS, int D, int W, int EB, class In, bool BF16 = false>
inline int launch_if_ok(const int* blk_offsets, const uint32_t* hspa_packed, const int* hind, int num_nodes,
                        int embedding_dim, const In* input, float* output, hipStream_t stream, const int* order,
                        const float* out_scale, int atomic_out, const int* units, const int* unit_ptr,
                        int max_units_per_xcd, float* partials, const int* row_map, int units_per_wave) {
  if constexpr (tile_ok<FS, D, W, EB>()) {
    return voltrix::launch_spmm_tc16<voltrix::SpmmTile<FS, D, W, EB, BF16>>(blk_offsets, hspa_packed, hind, num_nodes,
                                                                       embedding_dim, input, output, stream, order,
                                                                       out_scale, atomic_out, units, unit_ptr,
                                                                       max_units_per_xcd, partials, row_map, nullptr,
                                                                       units_per_wave);
  } else {
    return voltrix::kErrBadConfig;
  }
}

template <int EB, class In, bool BF16 = false>
inline int dispatch_spmm(int fs, int depth, int waves, const int* blk_offsets, const uint32_t* hspa_packed,
                         const int* hind, int num_nodes, int embedding_dim, const In* input, float* output,
                         hipStream_t stream, const int* order, const float* out_scale = nullptr,
                         int atomic_out = 0, const int* units = nullptr, const int* unit_ptr = nullptr,
                         int max_units_per_xcd = 0, float* partials = nullptr, const int* row_map = nullptr,
                         int units_per_wave = 1) {
#define X(FS, D, W)                                  \
  if (fs == FS && depth == D && waves == W)          \
    return launch_if_ok<FS, D, W, EB, In, BF16>(blk_offsets, hspa_packed, hind, num_nodes, embedding_dim, input, output, stream, order, out_scale, atomic_out, units, unit_ptr, max_units_per_xcd, partials, row_map, units_per_wave);
  VOLTRIX_TILE_SPACE(X)
#undef X
  return voltrix::kErrBadConfig;
}

// Default tile (measured on MI355X, profiles/HISTORY.md section 5; profiles/r02/experiment_units_reddit.log): the widest slab
// the feature width fills (up to 128 columns), a 3-deep ring, 4 waves per workgroup for the 16-bit operands -- (128, 3, 4)
// is the fastest tile on every graph measured, alone (reddit-like F=128: 2.20 ms vs 2.40 ms for 64-column slabs) and
// beside a panel-kernel workgroup (103 KB of LDS + 136 registers leave it room).  The exact-fp32 path is MFMA-heavier
// and prefers one wave per workgroup and 64-column slabs.
inline TileId default_tile(int embedding_dim, bool is_f16) {
  TileId t;
  if (is_f16) {
    t.fs = embedding_dim <= 32 ? 32 : (embedding_dim <= 64 ? 64 : 128);
    t.depth = t.fs == 32 ? 4 : 3;
    t.waves = 4;
  } else {
    t.fs = embedding_dim <= 32 ? 32 : 64;
    t.depth = 3;
    t.waves = 1;
  }
  return t;
}

}  // namespace voltrix_capi
