// Voltrix-SpMM for MI355X (gfx950 / CDNA4) -- compile-time tile geometry.
//
// Counterpart of the reference's voltrix/include/voltrix/traits.h:6-11,13-60 (BLK_H/BLK_W
// macros, PersistenKernelTraits).  Nothing is shared with it beyond the 16x8 "TC block"
// format constants: the gfx950 kernels are wave64, wave-private LDS rings, no thread-block
// wide barriers, and one v_mfma_f32_16x16x32_f16 consumes four 16x8 TC blocks (K = 32).
#pragma once

#include <cstdint>

// Diagnostic / experiment builds (harness/experiments/*) compile with -DVOLTRIX_EXPERIMENTAL plus one of the bit masks
// VOLTRIX_DIAG (spmm_kernels.hpp), VOLTRIX_PANEL_DIAG (spmm_panel_kernels.hpp), VOLTRIX_FUSED_DIAG (spmm_fused_kernels.hpp):
// they drop parts of a kernel (results wrong by design) or add time stamps.  A product build defines none of them; a mask
// without the gate is a build error, so no diagnostic path can slip into libvoltrix_hip.so or a JIT kernel.
#if !defined(VOLTRIX_EXPERIMENTAL) && (defined(VOLTRIX_DIAG) || defined(VOLTRIX_PANEL_DIAG) || defined(VOLTRIX_FUSED_DIAG) || defined(VOLTRIX_STREAM_DIAG))
#error "VOLTRIX_DIAG / VOLTRIX_PANEL_DIAG / VOLTRIX_FUSED_DIAG / VOLTRIX_STREAM_DIAG need -DVOLTRIX_EXPERIMENTAL (diagnostic builds only)"
#endif

#define VOLTRIX_BLK_H 16  // rows per row window          (reference traits.h:6)
#define VOLTRIX_BLK_W 8   // condensed columns per TC block (reference traits.h:7)

namespace voltrix {

constexpr int kBlkH = VOLTRIX_BLK_H;
constexpr int kBlkW = VOLTRIX_BLK_W;
constexpr int kWave = 64;                 // gfx950 wavefront
constexpr int kTcbPerStage = 4;           // 4 TC blocks = 32 condensed columns = one MFMA K step
constexpr int kStageK = kTcbPerStage * kBlkW;
constexpr int kNumXcd = 8;                // MI355X: 8 XCDs, blocks b and b+8 share an L2

// Error codes written to `__return_code` by every C-ABI entry point (the reference plumbs the
// variable but never sets it: template.py:114, runtime.py:50-52).
enum ReturnCode : int {
  kOk = 0,
  kErrBadShape = 1,       // unsupported num_nodes / embedding_dim / alignment
  kErrLaunch = 2,         // hipGetLastError() != hipSuccess after the launch
  kErrBadConfig = 3,      // tile configuration not instantiated / LDS budget exceeded
  kErrOverflow = 4,       // a 32-bit quantity of the reference format would overflow
  kErrDuplicate = 5,      // duplicate (row, col) entries where the contract forbids them
};

// Tile parameters of the SpMM kernel template (the autotune space; replaces the reference's
// `model` 0/1/2 switch, spmm_kernels.cuh:2014-2108).
//   FS     feature slab handled by one wave (columns of B / C): 32, 64, 128 or 256
//   DEPTH  stages of 32 gathered rows kept in flight per wave (LDS ring slots)
//   WAVES  waves per workgroup (each wave owns its (row window, slab) unit; no barriers)
//   EB     bytes per element of the dense operand in LDS: 2 (fp16, v_mfma_f32_16x16x32_f16) or
//          4 (fp32, exact v_mfma_f32_16x16x4_f32)
//   BF16   EB == 2 only: the 16-bit operand is bfloat16 (v_mfma_f32_16x16x32_bf16) instead of fp16
//   WEIGHTED  EB == 2 only: the A operand is a value plane (16 x 8 values of the operand's 16-bit type per TC block,
//          row-major, zeros where there is no edge) instead of the expansion of the bitmaps -- weighted SpMM, no reference
//          counterpart (the reference multiplies a binary A only, bmat_kernels.cuh:100-103)
template <int FS_, int DEPTH_, int WAVES_, int EB_ = 2, bool BF16_ = false, bool WEIGHTED_ = false>
struct SpmmTile {
  static constexpr int FS = FS_;
  static constexpr int DEPTH = DEPTH_;
  static constexpr int WAVES = WAVES_;
  static constexpr int EB = EB_;
  static constexpr bool BF16 = BF16_;
  static constexpr bool WEIGHTED = WEIGHTED_;
  static_assert(!BF16 || EB == 2, "bfloat16 is a 2-byte operand");
  static_assert(!WEIGHTED || EB == 2, "the value plane holds 16-bit values");
  static_assert(FS == 32 || FS == 64 || FS == 128 || FS == 256, "feature slab");
  static_assert(EB == 2 || EB == 4, "element bytes");
  static_assert(DEPTH >= 2 && DEPTH <= 6, "ring depth");
  static constexpr int ROW_BYTES = FS * EB;                      // slab of one gathered row
  static constexpr int STAGE_BYTES = kStageK * ROW_BYTES;         // 32 rows
  static constexpr int DMA_PER_STAGE = STAGE_BYTES / 1024;        // 1 KiB per global_load_lds_dwordx4
  static constexpr int ROWS_PER_DMA = 1024 / ROW_BYTES;
  static constexpr int LANES_PER_ROW = ROW_BYTES / 16;
  static constexpr int SLOTS = FS / 16;                           // 16-column slots (one MFMA N) per row
  static constexpr int META_SLOTS = 2 * DEPTH + 1;                // metadata ring (hind + bitmaps)
  // 32 hind + 16 bitmap words + pad; weighted: + the stage's 4 x 128 values (1 KiB, lane L's row of 8 at 256 + 16 L)
  static constexpr int META_BYTES = WEIGHTED ? 256 + 1024 : 256;
  static constexpr int META_DMAS = WEIGHTED ? 2 : 1;              // LDS-DMAs per metadata slot
  static constexpr int WAVE_LDS = DEPTH * STAGE_BYTES + META_SLOTS * META_BYTES;
  static constexpr int BLOCK_LDS = WAVES * WAVE_LDS;
  static constexpr int THREADS = WAVES * kWave;
  // ops a wave issues per pipeline step: META_DMAS metadata DMAs + DMA_PER_STAGE row-gather DMAs
  static constexpr int VM_PER_STEP = META_DMAS + DMA_PER_STAGE;
  static_assert(VM_PER_STEP * (DEPTH - 1) <= 63, "vmcnt is a 6-bit counter on gfx9");
  // s_waitcnt vmcnt immediate that retires the row stage being consumed while `k` younger row stages (k <= DEPTH-1)
  // and the DEPTH-1 metadata DMAs issued between them stay in flight
  static constexpr int vm_behind(int k) {
    return (DEPTH - 1) * META_DMAS + (k < DEPTH - 1 ? k : DEPTH - 1) * DMA_PER_STAGE;
  }
  static_assert(BLOCK_LDS <= 160 * 1024, "LDS per CU");
};

}  // namespace voltrix
