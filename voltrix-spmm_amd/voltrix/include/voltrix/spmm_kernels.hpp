// Voltrix-SpMM for MI355X (gfx950 / CDNA4) -- the tiled SpMM accumulate.
//
//   C[16w + r, :] = sum_{TC block b of window w} sum_{c < 8} bit(b, r, c) * B[hind[8b + c], :]
//
// Same math and the same (blk_offsets, hspa_packed, hind) handle as the reference's live kernels
// spmm_mma161616_spa_swizzle_d/_dd (voltrix/include/voltrix/spmm_kernels.cuh:1458-1727, :1729-2001)
// and host dispatcher voltrix_spmm_forward_cuda (:2003-2113).  The design is gfx950-native, not a
// translation of that Hopper code:
//
//   reference (sm_90a)                            here (gfx950)
//   ------------------------------------------   -----------------------------------------------------
//   CTA = 1 producer + 4 consumer warps,          work unit = ONE wave64 owning (row window, FS-column
//   mbarrier full/empty ring (:1496-1505)         slab); wave-private LDS ring; NO workgroup barrier at
//                                                 all -- the only ordering an LDS-DMA has is the issuing
//                                                 wave's vmcnt, so a wave-private pipeline needs nothing else
//   cp.async.bulk row gathers + cp.async          global_load_lds_dwordx4 (1 KiB / instruction, per-lane source
//   bitmaps (:1548-1570)                          = row gather) for B rows, global_load_lds_dword for the
//                                                 stage's hind + bitmap words; counted s_waitcnt vmcnt(N)
//   2 x mma.m16n8k8.tf32 per 16 columns per       1 x v_mfma_f32_16x16x32_f16 per 16 columns per FOUR TC blocks
//   TC block, B via 4 scalar LDS loads (:1654)    (K = 32); B operand via ds_read_b64_tr_b16 (hardware transpose)
//   +8 float row padding against bank conflicts   XOR slot swizzle applied on the DMA *source* address and on the
//   (traits.h:116-118)                            transposed read (LDS-DMA writes are lane-linear)
//   bit(lane) of 4 words -> tf32 1.0 (:1632-1644) 8 bits (two nibbles of two words) -> four packed fp16 {1,0} regs
//   hind padding (=0) gathers B[0] (quirk)        padded / out-of-window columns gather the window's first real
//                                                 column instead (A bits are 0), so B[0,:] = NaN cannot leak
//   grid = floor(N/16) (tail rows never written)  ceil(N/16) windows, tail rows masked at the store
//   int32 offsets (:1568,:1688)                   64-bit row offsets
//
// Bound: the kernel is gather-bound (B rows from L2 / Infinity Cache / HBM); MFMA utilisation is a few percent
// by construction.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

#include "voltrix/traits.hpp"

// Diagnostic builds only (-DVOLTRIX_EXPERIMENTAL, traits.hpp; harness/experiments/diag.py): bit 0 skips the consume side (LDS reads + MFMA), bit 1 folds every gathered
// row into the first 1024 rows of B (all L2 hits), bit 2 skips the output stores.  Results are wrong by design; the
// shipped kernels use 0.  Bit 3 (harness/experiments/exp_tail_histogram.py) keeps the results RIGHT and makes every wave
// leave {XCC id, HW id, start, end} (100 MHz ticks) in the int32 buffer passed as `row_map` (which is then not a row map).
#ifndef VOLTRIX_DIAG
#define VOLTRIX_DIAG 0
#endif


namespace voltrix {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef uint16_t bfloat16_bits;  // bfloat16 storage type of the dense operand (pointer arithmetic only)
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint2_t __attribute__((ext_vector_type(2)));
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
typedef int int4_t __attribute__((ext_vector_type(4)));

using gas_ptr = const void __attribute__((address_space(1)))*;
using lds_ptr = void __attribute__((address_space(3)))*;

// ---- LDS / wait helpers.  All LDS *reads* of DMA-written data go through inline asm so that hipcc does not
// ---- serialise them behind in-flight LDS-DMAs with a vmcnt(0) (cdna_hip_programming.md section 5, "Pipelining").
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// s_waitcnt vmcnt(min(m, 63)) for a wave-uniform run-time m.  The count is an immediate of the instruction, so the 64
// forms sit in a table of 8-byte entries (s_waitcnt ; s_branch end) entered through a computed s_setpc_b64.  (A binary
// search written as nested ifs was structurised by hipcc into a chain of ~60 flag tests and taken branches per call: about
// a microsecond per wait, most of the kernel's time.)  PC after s_getpc_b64 = the s_add_u32 below; table = + 12 bytes.
__device__ __forceinline__ void wait_vm(const int m) {
  int t;
  asm volatile("s_min_u32 %[t], %[m], 63\n"
               "s_lshl_b32 %[t], %[t], 3\n"
               "s_add_u32 %[t], %[t], 12\n"
               "s_getpc_b64 vcc\n"
               "s_add_u32 vcc_lo, vcc_lo, %[t]\n"
               "s_addc_u32 vcc_hi, vcc_hi, 0\n"
               "s_setpc_b64 vcc\n"
               "s_waitcnt vmcnt(0)\n s_branch 1f\n"
               "s_waitcnt vmcnt(1)\n s_branch 1f\n"
               "s_waitcnt vmcnt(2)\n s_branch 1f\n"
               "s_waitcnt vmcnt(3)\n s_branch 1f\n"
               "s_waitcnt vmcnt(4)\n s_branch 1f\n"
               "s_waitcnt vmcnt(5)\n s_branch 1f\n"
               "s_waitcnt vmcnt(6)\n s_branch 1f\n"
               "s_waitcnt vmcnt(7)\n s_branch 1f\n"
               "s_waitcnt vmcnt(8)\n s_branch 1f\n"
               "s_waitcnt vmcnt(9)\n s_branch 1f\n"
               "s_waitcnt vmcnt(10)\n s_branch 1f\n"
               "s_waitcnt vmcnt(11)\n s_branch 1f\n"
               "s_waitcnt vmcnt(12)\n s_branch 1f\n"
               "s_waitcnt vmcnt(13)\n s_branch 1f\n"
               "s_waitcnt vmcnt(14)\n s_branch 1f\n"
               "s_waitcnt vmcnt(15)\n s_branch 1f\n"
               "s_waitcnt vmcnt(16)\n s_branch 1f\n"
               "s_waitcnt vmcnt(17)\n s_branch 1f\n"
               "s_waitcnt vmcnt(18)\n s_branch 1f\n"
               "s_waitcnt vmcnt(19)\n s_branch 1f\n"
               "s_waitcnt vmcnt(20)\n s_branch 1f\n"
               "s_waitcnt vmcnt(21)\n s_branch 1f\n"
               "s_waitcnt vmcnt(22)\n s_branch 1f\n"
               "s_waitcnt vmcnt(23)\n s_branch 1f\n"
               "s_waitcnt vmcnt(24)\n s_branch 1f\n"
               "s_waitcnt vmcnt(25)\n s_branch 1f\n"
               "s_waitcnt vmcnt(26)\n s_branch 1f\n"
               "s_waitcnt vmcnt(27)\n s_branch 1f\n"
               "s_waitcnt vmcnt(28)\n s_branch 1f\n"
               "s_waitcnt vmcnt(29)\n s_branch 1f\n"
               "s_waitcnt vmcnt(30)\n s_branch 1f\n"
               "s_waitcnt vmcnt(31)\n s_branch 1f\n"
               "s_waitcnt vmcnt(32)\n s_branch 1f\n"
               "s_waitcnt vmcnt(33)\n s_branch 1f\n"
               "s_waitcnt vmcnt(34)\n s_branch 1f\n"
               "s_waitcnt vmcnt(35)\n s_branch 1f\n"
               "s_waitcnt vmcnt(36)\n s_branch 1f\n"
               "s_waitcnt vmcnt(37)\n s_branch 1f\n"
               "s_waitcnt vmcnt(38)\n s_branch 1f\n"
               "s_waitcnt vmcnt(39)\n s_branch 1f\n"
               "s_waitcnt vmcnt(40)\n s_branch 1f\n"
               "s_waitcnt vmcnt(41)\n s_branch 1f\n"
               "s_waitcnt vmcnt(42)\n s_branch 1f\n"
               "s_waitcnt vmcnt(43)\n s_branch 1f\n"
               "s_waitcnt vmcnt(44)\n s_branch 1f\n"
               "s_waitcnt vmcnt(45)\n s_branch 1f\n"
               "s_waitcnt vmcnt(46)\n s_branch 1f\n"
               "s_waitcnt vmcnt(47)\n s_branch 1f\n"
               "s_waitcnt vmcnt(48)\n s_branch 1f\n"
               "s_waitcnt vmcnt(49)\n s_branch 1f\n"
               "s_waitcnt vmcnt(50)\n s_branch 1f\n"
               "s_waitcnt vmcnt(51)\n s_branch 1f\n"
               "s_waitcnt vmcnt(52)\n s_branch 1f\n"
               "s_waitcnt vmcnt(53)\n s_branch 1f\n"
               "s_waitcnt vmcnt(54)\n s_branch 1f\n"
               "s_waitcnt vmcnt(55)\n s_branch 1f\n"
               "s_waitcnt vmcnt(56)\n s_branch 1f\n"
               "s_waitcnt vmcnt(57)\n s_branch 1f\n"
               "s_waitcnt vmcnt(58)\n s_branch 1f\n"
               "s_waitcnt vmcnt(59)\n s_branch 1f\n"
               "s_waitcnt vmcnt(60)\n s_branch 1f\n"
               "s_waitcnt vmcnt(61)\n s_branch 1f\n"
               "s_waitcnt vmcnt(62)\n s_branch 1f\n"
               "s_waitcnt vmcnt(63)\n"
               "1:\n"
               : [t] "=&s"(t)
               : [m] "s"(m)
               : "vcc", "scc", "memory");
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void wait_lgkmcnt0() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);  // keep MFMAs / users below the wait (guide rule 18)
}
__device__ __forceinline__ unsigned lds_read_b32(unsigned addr) {
  unsigned v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
template <int OFF>
__device__ __forceinline__ unsigned lds_read_b32_off(unsigned addr) {
  unsigned v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int STRIDE, int N, int... I>  // out[i] = LDS word at addr + i * STRIDE (immediate offsets: one address VGPR)
__device__ __forceinline__ void lds_read_b32_strided(unsigned addr, int (&out)[N], std::integer_sequence<int, I...>) {
  ((out[I] = (int)lds_read_b32_off<STRIDE * I>(addr)), ...);
}
__device__ __forceinline__ uint2_t lds_read_b64(unsigned addr) {
  uint2_t v;
  asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ uint4_t lds_read_b128(unsigned addr) {
  uint4_t v;  // every component is consumed by the caller (a dead component of an asynchronous asm read is a hazard)
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
template <int OFF>
__device__ __forceinline__ uint2_t lds_read_tr16_b64(unsigned addr) {
  uint2_t v;  // EXEC must be all ones here (T10): every caller is wave-uniform control flow
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ void dma_b128(const void* src, unsigned lds_byte_addr) {
  __builtin_amdgcn_global_load_lds((gas_ptr)src, (lds_ptr)(uintptr_t)lds_byte_addr, 16, 0, 0);
}
__device__ __forceinline__ void dma_b32(const void* src, unsigned lds_byte_addr) {
  __builtin_amdgcn_global_load_lds((gas_ptr)src, (lds_ptr)(uintptr_t)lds_byte_addr, 4, 0, 0);
}
// non-temporal form (aux = 2) for streams that are read once (metadata): they should not push gathered rows of B out of L2
__device__ __forceinline__ void dma_b32_nt(const void* src, unsigned lds_byte_addr) {
  __builtin_amdgcn_global_load_lds((gas_ptr)src, (lds_ptr)(uintptr_t)lds_byte_addr, 4, 0, 2);
}
__device__ __forceinline__ void dma_b128_nt(const void* src, unsigned lds_byte_addr) {
  __builtin_amdgcn_global_load_lds((gas_ptr)src, (lds_ptr)(uintptr_t)lds_byte_addr, 16, 0, 2);
}

// Slot swizzle: LDS row r keeps logical 32-byte slot s at physical slot s ^ slot_swizzle(r).  One transposed read
// touches, per 32-lane half, rows {8g+q, 8g'+q : q<4} (+4 for the second read); this makes their 8 x 32 B land on
// 8 distinct slots of the 256-B bank row for every FS (derivation in profiles/HISTORY.md section 3.1 "LDS image").
template <int SLOTS>
__device__ __forceinline__ constexpr int slot_swizzle(int r) {
  const int ident = (r & 3) | (((r >> 3) & 1) << 2);
  return SLOTS >= 8 ? ident : (SLOTS == 4 ? (ident >> 1) : (ident >> 2));
}

// Two adjacency nibbles (columns 0-3 and 4-7 of one row of a TC block) -> four packed fp16x2 registers holding 2.0 / 0.0.
// 2.0 = 0x4000 is a single bit, so the expansion is shift + mask only; the accumulators are scaled by 0.5 once at the
// store (exact: scaling by a power of two commutes with every fp32 rounding of the sum).
constexpr float kAScaleInv = 0.5f;
__device__ __forceinline__ half8_t nibbles_to_half8_x2(unsigned nl, unsigned nh) {
  // bit i and bit 15+i = column i.  Written as v_lshl_or_b32: left to itself hipcc turns n | n << 15 into a
  // quarter-rate v_mul_lo_u32 by 0x8001 << k.
  unsigned zl, zh;
  asm("v_lshl_or_b32 %0, %1, 15, %1" : "=v"(zl) : "v"(nl));
  asm("v_lshl_or_b32 %0, %1, 15, %1" : "=v"(zh) : "v"(nh));
  uint4_t r;
  r[0] = (zl << 14) & 0x40004000u;  // columns 0, 1
  r[1] = (zl << 12) & 0x40004000u;  // columns 2, 3
  r[2] = (zh << 14) & 0x40004000u;  // columns 4, 5
  r[3] = (zh << 12) & 0x40004000u;  // columns 6, 7
  return __builtin_bit_cast(half8_t, r);
}

// LDS-DMA with an immediate offset: the instruction's offset field is added to BOTH the global address and the LDS
// address (M0 + offset + lane * 16), so up to four 1-KiB pieces share one M0 value when the global pointer is
// pre-decremented by the same constant.
template <int OFF, int AUX = 0>
__device__ __forceinline__ void dma_b128_off(const void* src, unsigned lds_byte_addr) {
  __builtin_amdgcn_global_load_lds((gas_ptr)((const char*)src - OFF), (lds_ptr)(uintptr_t)lds_byte_addr, 16, OFF, AUX);
}

__device__ __forceinline__ float lds_read_f32(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: remember per (kernel instantiation,
// device) that it has been raised (one process may drive several GPUs; a function-local flag would cover the first only).
template <class Tag = void>
inline int ensure_dynamic_lds(const void* kernel, int bytes) {
  if (bytes <= 64 * 1024) return kOk;
  struct Seen { const void* k; int dev; };
  static std::mutex mu;
  static std::vector<Seen> seen;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return kErrLaunch;
  // fast path without the lock: the last (kernel, device) this thread raised -- the common case of one kernel launched
  // over and over from one host thread
  static thread_local Seen last = {nullptr, -1};
  if (last.k == kernel && last.dev == dev) return kOk;
  std::lock_guard<std::mutex> lock(mu);
  for (const Seen& s : seen)
    if (s.k == kernel && s.dev == dev) {
      last = s;
      return kOk;
    }
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return kErrBadConfig;
  seen.push_back(Seen{kernel, dev});
  last = seen.back();
  return kOk;
}

template <class T>
struct SpmmArgs {
  using in_t = typename std::conditional<T::EB == 4, float,
                                         typename std::conditional<T::BF16, bfloat16_bits, _Float16>::type>::type;
  const int* blk_offsets;        // [W+1]  (reference Pointer1)
  const uint32_t* hspa_packed;   // [4T]   swizzled bitmaps
  const int* hind;               // [8T]   condensed column -> row of B
  const in_t* input;             // [num_input_rows, F] row-major
  float* output;                 // [N, F] fp32 row-major
  int num_nodes;
  int num_windows;
  int F;
  int num_slabs;
  int windows_per_xcd;           // ceil(num_windows / 8)
  const int* window_order;       // optional schedule: position -> window id (launch_window_order); nullptr = natural
  const float* out_scale;        // optional device scalar multiplied into every output (cast_f32_to_f16_scaled); nullptr = 1
  const int4* units;             // optional unit table {window, phase, stride, slot}: the unit runs stages phase, phase +
                                 // stride, ... of the window (a long window cut into `stride` interleaved units of bounded
                                 // length, each sweeping the whole column range); replaces window_order.  slot < 0: the
                                 // result goes to C; slot >= 0: to tile `slot` of `partials` (combine_partials_kernel)
  const int* unit_ptr;           // [9]: XCD x owns units [unit_ptr[x], unit_ptr[x + 1])
  float* partials;               // [num partial tiles][16][F] fp32 (units with slot >= 0)
  const in_t* values;            // WEIGHTED tiles: [T][16][8] values of the operand's type (zeros where no edge); else unused
  const int* row_map;            // optional [16 W]: row i of the handle is row row_map[i] of C (-1: padding).  A handle built
                                 // from a row-permuted CSR (locality reorder, voltrix/reorder.py) writes C through it: no
                                 // un-permute pass.  Column ids are never relabelled, so B is gathered as it is.
  int slab_major;                // unit order inside an XCD's range when F spans several slabs (launcher, see slab_major_order)
  int slab_first;                // this launch covers the column slabs [slab_first, slab_first + num_slabs) of the F-wide operand
  int meta_nt;                   // 1: bitmap / hind DMAs are non-temporal (set by the launcher when one slab covers F, i.e.
                                 // every metadata byte is read exactly once and should not displace rows of B in L2)
  int atomic_out;                // 1: C += result by float atomics (C pre-zeroed; two-level format: no join pass)
};

// One wave64 = one (row window, FS-column slab) unit.  EB == 2: fp16 (or bfloat16) operand, v_mfma_f32_16x16x32_f16 (_bf16).
// EB == 4: fp32 operand, exact products on v_mfma_f32_16x16x4_f32 (k-step m of a stage uses LDS rows 4m + lane/16).
template <int I, int N, class Fn>
__device__ __forceinline__ void static_for(Fn&& fn) {
  if constexpr (I < N) {
    fn(std::integral_constant<int, I>{});
    static_for<I + 1, N>(fn);
  }
}

// NU = units per wave (1, or 2 = "paired" launch of a unit table with F <= FS): a wave that runs two units takes their
// stages alternately through ONE ring (stage t of the wave = stage t / 2 of unit t % 2) into two accumulator sets, so twice
// as many windows sweep their sorted columns side by side per CU at the same LDS and the same bytes in flight.
template <class T, int NU>
__device__ __forceinline__ void spmm_tc16_body(const SpmmArgs<T>& a) {
  static_assert(NU == 1 || NU == 2, "units per wave");
  constexpr int FS = T::FS, D = T::DEPTH, MS = T::META_SLOTS, EB = T::EB;
  constexpr int ROW_BYTES = T::ROW_BYTES, STAGE_BYTES = T::STAGE_BYTES, NDMA = T::DMA_PER_STAGE;
  constexpr int RPD = T::ROWS_PER_DMA, LPR = T::LANES_PER_ROW, SLOTS = T::SLOTS;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
  const unsigned long long diag_t0 = (VOLTRIX_DIAG & 8) ? __builtin_amdgcn_s_memrealtime() : 0ull;

  // XCD x = blockIdx.x % 8 owns the contiguous window range [x * windows_per_xcd, ...) (blocks b, b+8, ... share an
  // XCD: speed only, any placement is correct); its workgroups walk that range window by window (the slabs of a window
  // side by side, or slab by slab: slab_major_order below), so co-resident waves of one L2 gather overlapping row
  // neighbourhoods.  (Measured alternative that lost: non-temporal loads for far rows -- profiles/HISTORY.md section 5.)
  const int xcd = blockIdx.x % kNumXcd;
  int w_begin = xcd * a.windows_per_xcd;
  int w_count = (a.num_windows - w_begin) < a.windows_per_xcd ? (a.num_windows - w_begin) : a.windows_per_xcd;
  if (a.units) {  // unit table: the XCD's range of units instead of its range of windows
    w_begin = a.unit_ptr[xcd];
    w_count = a.unit_ptr[xcd + 1] - w_begin;
  }
  // Issue priority over whatever else shares the SIMD.  Beside a panel-kernel workgroup (two-level format) this wave is
  // the YOUNGER one -- panel workgroups live 4-5 times longer -- and instruction issue is arbitrated by priority, then
  // age: the panel's two issue-dense waves per SIMD took nearly every slot and this gather-bound wave, which only needs a
  // few slots at the right time to keep its DMAs in flight, ran at half speed.  Measured on the reddit-like pair: 1.48 ->
  // 1.34 ms, both kernels finishing together (profiles/r02/experiment_corun_diag_setprio.log).  Alone it changes nothing.
  __builtin_amdgcn_s_setprio(3);
  // Unit(s) of this wave inside the XCD's range.  One unit per wave: position lu of w_count x num_slabs, window-major (the
  // slabs of a window side by side: shared metadata, overlapping row neighbourhoods) or slab-major (the whole range for slab
  // 0, then slab 1, ...: one FS-wide column slab of B is the cache working set at a time).  Two units per wave: slab-major
  // only (launcher), pairs never straddle a slab: pair q of ceil(w_count / 2) per slab = units 2 q, 2 q + 1 of that slab.
  const int wave_pos = (int)(blockIdx.x / kNumXcd) * T::WAVES + wave;
  const int pairs_per_slab = (w_count + 1) / 2;
  if (w_count <= 0 || (NU == 1 ? wave_pos >= w_count * a.num_slabs : wave_pos >= pairs_per_slab * a.num_slabs)) return;  // wave-uniform; no barriers
  const int lu = NU == 1 ? wave_pos : (wave_pos / pairs_per_slab) * w_count + 2 * (wave_pos % pairs_per_slab);
  int w[NU], st0[NU], st_step[NU], slot[NU];   // unit p runs stages st0, st0 + st_step, ... of window w
  int kb0[NU], kb1[NU], nst_u[NU], h_safe[NU];
  bool present[NU];
  int nst_max = 0;
#pragma unroll
  for (int p = 0; p < NU; ++p) {
    present[p] = NU == 1 || (lu % w_count) + p < w_count;   // NU == 2: the second unit of a slab's last pair may not exist
    const int lup = present[p] ? lu + p : lu;   // an odd unit out: the wave's second unit is a copy that does nothing
    const int wpos = w_begin + ((a.slab_major || NU == 2) ? lup % w_count : lup / a.num_slabs);
    st0[p] = 0;
    st_step[p] = 1;
    slot[p] = -1;
    if (a.units) {
      const int4 u = a.units[wpos];
      w[p] = u.x;
      st0[p] = u.y;
      st_step[p] = u.z;
      slot[p] = u.w;
    } else {
      w[p] = a.window_order ? a.window_order[wpos] : wpos;
    }
    kb0[p] = a.blk_offsets[w[p]];
    kb1[p] = a.blk_offsets[w[p] + 1];
    const int nblk = kb1[p] - kb0[p];
    const int nst_window = (nblk + kTcbPerStage - 1) / kTcbPerStage;
    nst_u[p] = st0[p] < nst_window ? (nst_window - st0[p] + st_step[p] - 1) / st_step[p] : 0;
    // a window without edges owns one all-zero TC block (reference quirk, bmat_kernels.cuh:252): nothing to gather
    if (nblk == 1) {
      const uint4 w4 = *reinterpret_cast<const uint4*>(a.hspa_packed + 4ll * kb0[p]);
      if ((w4.x | w4.y | w4.z | w4.w) == 0u) nst_u[p] = 0;
    }
    if (!present[p]) nst_u[p] = 0;
    h_safe[p] = a.hind[8ll * kb0[p]];  // first real column of the window: finite data, gathered anyway
    nst_max = nst_u[p] > nst_max ? nst_u[p] : nst_max;
  }
  const int nst = NU * nst_max;   // stages of the wave: stage t = stage t / NU of unit t % NU (a unit past its end idles)
  const int fs0 = (a.slab_first + (int)((a.slab_major || NU == 2) ? lu / w_count : lu % a.num_slabs)) * FS;
  const int F = a.F;
  auto stage_of = [&](int t) -> int { return st0[t % NU] + st_step[t % NU] * (t / NU); };        // stage of its window
  auto stage_block = [&](int t) -> int { return kb0[t % NU] + kTcbPerStage * stage_of(t); };     // its first TC block
  auto stage_end = [&](int t) -> int { return kb1[t % NU]; };                                     // end of its window

  float4_t acc[NU][SLOTS];
#pragma unroll
  for (int p = 0; p < NU; ++p)
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) acc[p][s] = float4_t{0.f, 0.f, 0.f, 0.f};

  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem + (unsigned)wave * T::WAVE_LDS;
  const unsigned meta0 = lds0 + D * STAGE_BYTES;

  if (nst > 0) {

    // ---- lane constants -------------------------------------------------------------------------------------
    const int k32 = lane & 31;             // condensed column of the stage this lane holds metadata for
    const int kblk = k32 >> 3, kcol = k32 & 7;
    const unsigned colmask = 0x11111111u << (kcol & 3);
    const unsigned vword_off = 128 + 4 * (4 * kblk + 2 * (kcol >> 2));  // words t = 2(c>>2), +1 of block kblk
    const int g = lane >> 4, R = lane & 15;  // MFMA lane group / row (A) or column (B, D)
    const unsigned a_shift = 4 * (R & 7);
    const int mj = (lane - 32) & 15;         // metadata DMA: lanes 32-63 fetch bitmap words (48-63 duplicate)
    // transposed B reads (EB == 2): lane 16g+4q+p supplies LDS row 8g+q (+4), bytes 8p.. of logical slot s
    unsigned tr_base = 0;
    int tr_delta[4] = {0, 0, 0, 0};
    if constexpr (EB == 2) {
      const int trow = 8 * g + ((lane >> 2) & 3);
      const int tr_z = slot_swizzle<SLOTS>(trow);
      tr_base = lds0 + trow * ROW_BYTES + 8 * (lane & 3) + (tr_z << 5);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        tr_delta[b] = ((tr_z >> b) & 1) ? -(32 << b) : (32 << b);
        asm volatile("" : "+v"(tr_delta[b]));   // keep the chain: no re-derivation from tr_z in the loop
      }
      asm volatile("" : "+v"(tr_base));
    }

    // metadata of a stage that lies fully inside the window: lane address = base + tau * stride (one 64-bit mad);
    // lanes 0-31 fetch hind[8 * block + k32] (32 ints / stage), lanes 32-63 the 16 bitmap words of the stage
    const char* const meta_base = lane < 32 ? (const char*)(a.hind + k32) : (const char*)(a.hspa_packed + mj);
    const unsigned meta_blk_bytes = lane < 32 ? 4u * 8u : 4u * 4u;   // bytes per TC block in hind / hspa_packed
    auto issue_meta = [&](int tau, int mslot) {
      const void* src;
      const int sb = stage_block(tau);
      const int kb1 = stage_end(tau);
      if (sb + kTcbPerStage <= kb1) {  // wave-uniform
        src = meta_base + (unsigned long long)(unsigned)sb * meta_blk_bytes;
      } else if (lane < 32) {
        int blk = sb + kblk;
        blk = blk < kb1 ? blk : kb1 - 1;  // stay inside the window: stages past its end re-read its last block
        src = a.hind + (8ll * blk + kcol);
      } else {
        int blk = sb + (mj >> 2);
        blk = blk < kb1 ? blk : kb1 - 1;
        src = a.hspa_packed + (4ll * blk + (mj & 3));
      }
      if (a.meta_nt)   // workgroup-uniform
        dma_b32_nt(src, meta0 + mslot * T::META_BYTES);
      else
        dma_b32(src, meta0 + mslot * T::META_BYTES);
      if constexpr (T::WEIGHTED) {
        // values of the stage's four TC blocks: lane 16 g + R fetches row R of block g (8 values = 16 bytes; the wave's
        // 1 KiB is contiguous inside the window); blocks past the window's end re-read its last block and are zeroed below
        int vblk = sb + (lane >> 4);
        vblk = vblk < kb1 ? vblk : kb1 - 1;
        dma_b128((const char*)a.values + ((long long)vblk * 128 + (lane & 15) * 8) * 2,
                 meta0 + mslot * T::META_BYTES + 256);
      }
    };

    // hv: lane L holds the (sanitised) row of B for condensed column L & 31 of the stage
    // hr[i]: row of B this lane gathers in DMA i of the stage (LDS row i * RPD + lane / LPR)
    // Lane constants of the row gathers: DMA i of a stage reads 16 bytes at column col(i, lane) of row hr[i], i.e.
    // address = cbase[i] + hr[i] * row_bytes -- one v_mad_u64_u32 per DMA.  cbase[i] already carries the -K KiB that
    // the instruction's immediate offset (K = i % 4, shared M0) adds back.
    const unsigned row_bytes = (unsigned)F * EB;
    // DMAs i and i + 4 of a stage (FS = 128, 16-bit operand: 4 rows per DMA) gather the same chunk column with the same
    // immediate offset: the swizzle of LDS row 4 i + x repeats with period 16 rows -- four base pointers serve eight DMAs
    constexpr int NBASE = (EB == 2 && NDMA == 8 && RPD == 4) ? 4 : NDMA;
    static_assert(NBASE == NDMA || slot_swizzle<SLOTS>(4 * RPD + 1) == slot_swizzle<SLOTS>(1), "period of the row swizzle");
    const char* cbase[NBASE];
#pragma unroll
    for (int i = 0; i < NBASE; ++i) {
      const int r = i * RPD + lane / LPR;       // LDS row written by this lane
      const int c = lane % LPR;                 // 16-byte chunk inside the row
      int col;                                  // logical column of that chunk (swizzle on the SOURCE)
      if constexpr (EB == 2)
        col = fs0 + (((c >> 1) ^ slot_swizzle<SLOTS>(r)) * 16) + (c & 1) * 8;
      else
        col = fs0 + (((c >> 2) ^ (r & 1)) * 16) + (c & 3) * 4;
      col = col < F ? col : fs0;                // F % FS tail: stay in bounds, results are not stored
      unsigned long long cb = (unsigned long long)((const char*)a.input + ((long long)col * EB - (i & 3) * 1024));
      asm volatile("" : "+v"(cb));              // one VGPR pair per DMA: no base + uniform-delta re-derivation in the loop
      cbase[i] = (const char*)cb;
    }
    auto issue_data = [&](int dslot, const int (&hr)[NDMA]) {
      const unsigned dst = lds0 + dslot * STAGE_BYTES;
      auto piece = [&](auto kc, int ib) {           // DMA number ib + K of the stage, K = 0..3 sharing one M0
        constexpr int K = decltype(kc)::value;
        if constexpr (K < NDMA) {
          const int i = ib + K;
          unsigned hrow = (unsigned)hr[i];
          if (VOLTRIX_DIAG & 2) hrow &= 1023u;
          const char* src = cbase[i % NBASE] + (unsigned long long)hrow * row_bytes;
          __builtin_amdgcn_global_load_lds((gas_ptr)src, (lds_ptr)(uintptr_t)(dst + ib * 1024), 16, K * 1024, 0);
        }
      };
#pragma unroll
      for (int ib = 0; ib < NDMA; ib += 4) {
        piece(std::integral_constant<int, 0>{}, ib);
        piece(std::integral_constant<int, 1>{}, ib);
        piece(std::integral_constant<int, 2>{}, ib);
        piece(std::integral_constant<int, 3>{}, ib);
      }
    };
    // tail stages only: rows come from the sanitised per-column values (lane L holds column L & 31) via the crossbar
    auto rows_from_columns = [&](int hv, int (&hr)[NDMA]) {
#pragma unroll
      for (int i = 0; i < NDMA; ++i) hr[i] = __shfl(hv, i * RPD + lane / LPR, kWave);
    };
    // every other stage: each lane reads the hind words of its own DMA rows straight from the metadata slot
    const unsigned hr_off = 4 * (lane / LPR);
    auto read_rows = [&](unsigned mbase, int (&hr)[NDMA]) {
      const unsigned base = mbase + hr_off;
      lds_read_b32_strided<4 * RPD>(base, hr, std::make_integer_sequence<int, NDMA>{});
    };

    auto sanitise = [&](int tau, unsigned hraw, uint2_t vw) -> int {
      const bool inwin = (stage_block(tau) + kblk) < stage_end(tau);
      const bool valid = inwin && (((vw[0] | vw[1]) & colmask) != 0u);
      return valid ? (int)hraw : h_safe[tau % NU];  // padded hind slots are 0 in the format: never gather B[0] for them
    };

    // ---- prologue: metadata of stages 0..D-1, then (metadata D+j, rows of stage j) for j < D -------------------
#pragma unroll
    for (int j = 0; j < D; ++j) issue_meta(j, j);
    wait_vmcnt<0>();
#pragma unroll
    for (int j = 0; j < D; ++j) {
      issue_meta(D + j, (D + j) % MS);
      if (j < nst) {
        const unsigned m = meta0 + j * T::META_BYTES;
        const unsigned hraw = lds_read_b32(m + 4 * k32);
        const uint2_t vw = lds_read_b64(m + vword_off);
        wait_lgkmcnt0();
        int hr[NDMA];
        rows_from_columns(sanitise(j, hraw, vw), hr);
        issue_data(j, hr);
      }
    }

    // ---- main loop -----------------------------------------------------------------------------------------
    int dslot = 0;                  // t % D
    int mslot = 0;                  // t % MS
    int mslot_d = D;                // (t + D) % MS
    int mslot_2d = (2 * D) % MS;    // (t + 2D) % MS
    auto kb1_of = [&](int p) -> int { return kb1[p]; };
    // One step = consume stage t, refill its ring slot with stage t+D.  MORE (= t + D < nst) is static: the steady
    // loop always refills and always waits with the same count, the last D steps only drain.
    auto step = [&](int t, auto more_c, auto unit_c) {
      constexpr bool more = decltype(more_c)::value;
      constexpr int P = decltype(unit_c)::value;   // = t % NU: the unit whose stage is consumed (accumulator set)
      const int kb1 = kb1_of(P);
      // wave-uniform: a unit past its end re-gathers its last block with a zero A fragment (keeps the DMA count static); the
      // accumulators of a unit that never had a stage (empty window, odd unit out) are not stored at all (epilogue)
      const bool live = NU == 1 || t / NU < nst_u[P];
      const unsigned mt = meta0 + mslot * T::META_BYTES;
      // padded hind slots exist only in a window's last TC block: every earlier stage takes hind as it is
      const bool tail_stage = stage_block(t + D) + kTcbPerStage >= stage_end(t + D);  // wave-uniform
      unsigned hraw = 0;
      uint2_t vw = {0u, 0u};
      int hr[NDMA];
      if (more) {
        // the DMA rows' hind words are read unconditionally (a tail stage overwrites them below): a conditional read makes
        // the compiler zero-fill hr[] every step (8 v_mov in the steady loop)
        const unsigned md = meta0 + mslot_d * T::META_BYTES;
        read_rows(md, hr);
        if (tail_stage) {
          hraw = lds_read_b32(md + 4 * k32);
          vw = lds_read_b64(md + vword_off);
        }
      }

      if (VOLTRIX_DIAG & 1) {
        wait_lgkmcnt0();  // hraw / vw are inline-asm LDS reads: they must have landed before they become addresses
        issue_meta(t + 2 * D, mslot_2d);
        if (more) {
          if (tail_stage) rows_from_columns(sanitise(t + D, hraw, vw), hr);
          issue_data(dslot, hr);
        }
      } else if constexpr (EB == 2) {
        // A: lane -> row R of TC block g; its 8 bits are nibble R&7 of words t = R>>3 (cols 0-3), 2 + R>>3 (cols 4-7)
        unsigned wlo = 0u, whi = 0u;
        uint4_t avals = {0u, 0u, 0u, 0u};
        if constexpr (T::WEIGHTED) {
          avals = lds_read_b128(mt + 256 + 16 * lane);  // A[row R][8 g .. 8 g + 7] as stored: the MFMA A fragment
        } else {
          wlo = lds_read_b32(mt + 128 + 4 * (4 * g + (R >> 3)));
          whi = lds_read_b32(mt + 128 + 4 * (4 * g + 2 + (R >> 3)));
        }
        // B: lane 16g+4q+p supplies LDS row 8g+q (+4), bytes 8p.. of logical slot s; receives column R
        // physical slot of logical slot s = s ^ tr_z: address(s) = address(0) +- 32, +- 64, ... per set bit of s, the sign
        // being the lane's (tr_delta): SLOTS - 1 adds per stage instead of an xor + add per slot
        unsigned taddr[SLOTS];
        taddr[0] = tr_base + dslot * STAGE_BYTES;
#pragma unroll
        for (int b = 0; (1 << b) < SLOTS; ++b)
#pragma unroll
          for (int s = (1 << b); s < (2 << b) && s < SLOTS; ++s) taddr[s] = taddr[s - (1 << b)] + tr_delta[b];
        // A fragment first (its LDS words were requested above); B fragments in one go, or -- two units per wave, where the
        // second accumulator set has taken the registers -- in two halves, the ring slot being refilled after the last read
        auto a_fragment = [&]() -> half8_t {
          if constexpr (T::WEIGHTED) {
            uint4_t av = avals;
            if (stage_block(t) + g >= kb1 || !live) av = uint4_t{0u, 0u, 0u, 0u};  // TC blocks past the window's end contribute zero
            return __builtin_bit_cast(half8_t, av);
          } else {
            unsigned nl = (wlo >> a_shift) & 0xFu, nh = (whi >> a_shift) & 0xFu;
            if (stage_block(t) + g >= kb1 || !live) nl = nh = 0u;  // TC blocks past the window's end contribute zero
            return nibbles_to_half8_x2(nl, nh);
          }
        };
        auto refill = [&]() {  // metadata for stage t+2D (always: keeps the vmcnt arithmetic static), rows for stage t+D
          issue_meta(t + 2 * D, mslot_2d);
          if (more) {
            if (tail_stage) rows_from_columns(sanitise(t + D, hraw, vw), hr);
            issue_data(dslot, hr);
          }
        };
        auto mfma = [&](const half8_t afrag, const uint2_t lo, const uint2_t hi, const int s) {
          const uint4_t bq = {lo[0], lo[1], hi[0], hi[1]};
          // 2.0 is 0x4000 in fp16 AND in bfloat16, so the A fragment is the same bits for both operand types
          if constexpr (T::BF16)
            acc[P][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, afrag),
                                                                __builtin_bit_cast(bf16x8_t, bq), acc[P][s], 0, 0, 0);
          else
            acc[P][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag, __builtin_bit_cast(half8_t, bq), acc[P][s], 0, 0, 0);
        };
        constexpr int HALVES = (NU == 2 && SLOTS >= 8) ? 2 : 1;
        constexpr int SPH = SLOTS / HALVES;
        half8_t afrag;
#pragma unroll
        for (int h = 0; h < HALVES; ++h) {
          uint2_t blo[SPH], bhi[SPH];
#pragma unroll
          for (int s = 0; s < SPH; ++s) {
            blo[s] = lds_read_tr16_b64<0>(taddr[h * SPH + s]);
            bhi[s] = lds_read_tr16_b64<4 * ROW_BYTES>(taddr[h * SPH + s]);
          }
          wait_lgkmcnt0();
          if (h == 0) afrag = a_fragment();
          if (h == HALVES - 1) refill();
#pragma unroll
          for (int s = 0; s < SPH; ++s) mfma(afrag, blo[s], bhi[s], h * SPH + s);
        }
      } else {
        // A: k-step m covers condensed columns 4m+g: TC block m>>1, column 4(m&1)+g -> word 2(m&1) + R>>3, bit
        // 4(R&7) + g.  Per lane: words R>>3 and 2 + R>>3 of each of the stage's 4 blocks.
        unsigned wlo[4], whi[4];
#pragma unroll
        for (int bm = 0; bm < 4; ++bm) {
          wlo[bm] = lds_read_b32(mt + 128 + 4 * (4 * bm + (R >> 3)));
          whi[bm] = lds_read_b32(mt + 128 + 4 * (4 * bm + 2 + (R >> 3)));
        }
        // B: lane reads LDS row 4m+g, column 16s+R; 64-byte slots are XORed with (row & 1) = g & 1, which turns into
        // two lane bases (even / odd s) plus compile-time offsets
        const int z = g & 1;
        const unsigned lbase = lds0 + dslot * STAGE_BYTES + g * ROW_BYTES + 4 * R;
        const unsigned base_e = lbase + 64 * z, base_o = lbase - 64 * z;
        float bv[8][SLOTS];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
#pragma unroll
          for (int s = 0; s < SLOTS; ++s) {
            bv[m][s] = lds_read_f32(((s & 1) ? base_o : base_e) + (4 * m * ROW_BYTES + 64 * s));
          }
        }
        wait_lgkmcnt0();

        issue_meta(t + 2 * D, mslot_2d);
        if (more) {
          if (tail_stage) rows_from_columns(sanitise(t + D, hraw, vw), hr);
          issue_data(dslot, hr);
        }

#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const int bm = m >> 1;
          unsigned word = (m & 1) ? whi[bm] : wlo[bm];
          if (stage_block(t) + bm >= kb1 || !live) word = 0u;
          const float av = ((word >> (a_shift + g)) & 1u) ? 1.0f : 0.0f;
#pragma unroll
          for (int s = 0; s < SLOTS; ++s) acc[P][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[m][s], acc[P][s], 0, 0, 0);
        }
      }

      dslot = dslot + 1 == D ? 0 : dslot + 1;
      mslot = mslot + 1 == MS ? 0 : mslot + 1;
      mslot_d = mslot_d + 1 == MS ? 0 : mslot_d + 1;
      mslot_2d = mslot_2d + 1 == MS ? 0 : mslot_2d + 1;
    };
    // stage t's rows and stage t+D's metadata were issued D steps ago; everything younger may stay in flight:
    // (D-1) metadata DMAs + NDMA per younger row stage (at most D-1 of them)
    int t = 0;
    if constexpr (NU == 2) {
      // nst is even, so the unit of every step is static: the steady loop runs two steps at a time, the drain is unrolled
      // (step nst - D + i consumes unit (D + i) % 2) -- no branch merges two accumulator sets
      for (; t + 1 < nst - D; t += 2) {
        wait_vmcnt<T::vm_behind(D - 1)>();
        step(t, std::true_type{}, std::integral_constant<int, 0>{});
        wait_vmcnt<T::vm_behind(D - 1)>();
        step(t + 1, std::true_type{}, std::integral_constant<int, 1>{});
      }
      if (t < nst - D) {  // t is even
        wait_vmcnt<T::vm_behind(D - 1)>();
        step(t, std::true_type{}, std::integral_constant<int, 0>{});
      }
      static_for<0, D>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        if (nst - D + I >= 0) {  // wave-uniform (nst < D: the first steps do not exist)
          wait_vmcnt<T::vm_behind(D - 1 - I)>();
          step(nst - D + I, std::false_type{}, std::integral_constant<int, (D + I) & 1>{});
        }
      });
    } else {
      for (; t < nst - D; ++t) {
        wait_vmcnt<T::vm_behind(D - 1)>();
        step(t, std::true_type{}, std::integral_constant<int, 0>{});
      }
      for (; t < nst; ++t) {
        switch (nst - 1 - t) {  // younger row stages still in flight: 0 .. D-1
          case 0: wait_vmcnt<T::vm_behind(0)>(); break;
          case 1: wait_vmcnt<T::vm_behind(1)>(); break;
          case 2: wait_vmcnt<T::vm_behind(2)>(); break;
          case 3: wait_vmcnt<T::vm_behind(3)>(); break;
          case 4: wait_vmcnt<T::vm_behind(4)>(); break;
          default: wait_vmcnt<T::vm_behind(D - 1)>(); break;  // k = 5 = DEPTH - 1 at the deepest ring (DEPTH <= 6)
        }
        step(t, std::false_type{}, std::integral_constant<int, 0>{});
      }
    }
    wait_vmcnt<0>();  // the trailing metadata DMAs must have landed before the wave's LDS is released
  }

  // ---- epilogue: D[row = 4*(lane>>4) + j][col = lane & 15] per 16-column slot ----------------------------------
  if (VOLTRIX_DIAG & 4) {
    if (acc[0][0][0] == 12345.678f) a.output[0] = acc[0][0][0];  // keep the accumulators live
    return;
  }
  // powers of two: exact (barring overflow / underflow of the result itself)
  const float oscale = ((EB == 2 && !T::WEIGHTED) ? kAScaleInv : 1.0f) * (a.out_scale ? *a.out_scale : 1.0f);
  const int ocol0 = fs0 + (lane & 15);
#pragma unroll
  for (int p = 0; p < NU; ++p) {
    if (!present[p]) continue;  // wave-uniform
    // NU == 2: a unit without stages may have seen MFMAs of zero A fragments on rows that are not its own (0 x NaN)
    const bool dead = NU > 1 && nst_u[p] == 0;
    if (slot[p] >= 0) {  // a cut window's partial tile: [16][F] fp32, summed in unit order by combine_partials_kernel
      float* const tile = a.partials + (long long)slot[p] * (kBlkH * (long long)F) + (long long)(4 * (lane >> 4)) * F;
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        const int col = ocol0 + 16 * s;
        if (col < F) {
#pragma unroll
          for (int j = 0; j < 4; ++j) tile[(long long)j * F + col] = dead ? 0.f : acc[p][s][j] * oscale;
        }
      }
      continue;
    }
    const int orow0 = w[p] * kBlkH + 4 * (lane >> 4);
    int orow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      orow[j] = orow0 + j;
      if (a.row_map && !(VOLTRIX_DIAG & 8)) orow[j] = a.row_map[orow[j]];
      if (orow[j] >= a.num_nodes) orow[j] = -1;
    }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int col = ocol0 + 16 * s;
      if (col < F) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = orow[j];
          if (row >= 0) {
            float* const dst = a.output + ((long long)row * F + col);
            const float v = dead ? 0.f : acc[p][s][j] * oscale;
            if (a.atomic_out)
              unsafeAtomicAdd(dst, v);  // global_atomic_add_f32, no return; two addends per element
            else
              *dst = v;
          }
        }
      }
    }
  }
  if ((VOLTRIX_DIAG & 8) && a.row_map && lane == 0) {   // when and where this wave ran
    int* const stamp = const_cast<int*>(a.row_map) + 4ll * ((long long)blockIdx.x * T::WAVES + wave);
    stamp[0] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    stamp[1] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID (CU / SIMD / wave slot)
    stamp[2] = (int)(unsigned)diag_t0;
    stamp[3] = (int)(unsigned)__builtin_amdgcn_s_memrealtime();
  }
}

template <class T>
static __global__ __launch_bounds__(T::THREADS) void spmm_tc16_kernel(const SpmmArgs<T> a) {
  spmm_tc16_body<T, 1>(a);
}

// two units per wave: 160 registers at most, so that the wave still fits beside two panel-kernel waves on its SIMD
template <class T>
static __global__ __launch_bounds__(T::THREADS) __attribute__((amdgpu_num_vgpr(160)))
void spmm_tc16_pair_kernel(const SpmmArgs<T> a) {
  spmm_tc16_body<T, 2>(a);
}

// Unit order for F > FS (SpmmArgs::slab_major): slab-major as soon as a slab's piece of a row of B is a whole 128-byte
// cache line.  One FS-wide column slab of B is then the working set of L2 / Infinity Cache at a time instead of all of
// them (reddit-like F=512: 12.4 -> 9.2 ms for the window format, products-like F=512 FS=64: 17.5 -> 14.5 ms, reddit-like
// F=128 FS=64: 2.53 -> 2.06 ms); with 64-byte pieces two neighbouring slabs share every line and window-major wins
// (F=64 FS=32: 1.55 vs 2.30 ms) -- profiles/r02/experiment_slab_order.log.  The panel kernel's grid (slab = blockIdx.y)
// is slab-major by construction, so the two kernels of the two-level format walk the slabs in step.
inline int slab_major_order(int num_slabs, int slab_row_bytes) { return num_slabs > 1 && slab_row_bytes >= 128; }

// Wide operands (F > FS): ONE LAUNCH PER 256-BYTE GROUP OF COLUMN SLABS instead of one grid over all slabs (round 3), when
// such a group of B fits the Infinity Cache.  Inside one grid the XCDs (and, in the two-level format, the two kernels) drift
// apart by whole slabs and the caches hold pieces of several slabs at once; launch boundaries keep every CU on the same
// columns of B.  Measured through the operator, one call (profiles/r03/experiment_slab_calls.log, experiment_slab_groups.log):
//   reddit-like (233 k rows, 60 MB per 128-column slab)   F=256 2.84 -> 2.58 ms, F=512 6.03-6.08 -> 5.15-5.25, F=1024 12.5 -> 10.5
//   reddit-uniform                                        F=512 7.46 -> 6.86 ms
//   products-like (2.45 M rows, 64-column tiles: 627 MB)  F=512 13.48 as one grid; 14.3 one launch per slab, 13.8 per pair
//   power-law 4 M (1 GB per 128-column slab)              F=256 113.3 as one grid, 110.7 per slab
//   reddit-like at 2x / 4x the rows (114 / 228 MiB)     F=512 two-level 14.28 -> 13.56 / 31.97 -> 29.89, window format 23.1 -> 21.4 / 51.8 -> 50.7
// so: a launch covers 256 bytes of every row of B (one slab of 128 fp16 columns, two of 64) when rows x 256 B <= 256 MiB (the
// Infinity Cache; the rows of A stand in for the rows of B the launcher does not know -- square adjacency: equal), and the
// HBM-resident graphs keep the single grid, where the slabs of a window side by side share its metadata and its rows' DRAM
// pages.  (products-like gains 3.5 % from four calls on CONTIGUOUS 128-column copies of B -- a layout effect, not a launch
// effect; a slab-major B would cost the caller a reformat pass.)  Slabs of whole 128-byte lines only (the rule of the
// slab-major order).  `rows` = the rows of B (the dense operand), which is what has to fit the cache: a row shard of a
// multi-GPU run has few rows of A and all the rows of the gathered B.  `policy`: kSlabAuto = the rule, kSlabOneGrid /
// kSlabLaunches force either form (callers that know better, tests) -- an ARGUMENT of the launchers, nothing is read from
// the environment on the launch path.  Returns the slabs per launch, 0 = one grid.
enum SlabPolicy : int { kSlabAuto = -1, kSlabOneGrid = 0, kSlabLaunches = 1 };
inline int slab_launch_group(int num_slabs, int slab_row_bytes, long long rows, int policy = kSlabAuto) {
  if (policy == kSlabOneGrid || slab_row_bytes < 128) return 0;
  if (policy == kSlabAuto && rows * 256 > (256ll << 20)) return 0;
  const int group = slab_row_bytes >= 256 ? 1 : 256 / slab_row_bytes;
  return num_slabs > group ? group : 0;
}

// ----------------------------------------------------------------------------------------------
// Host launcher for one tile configuration.
template <class T>
inline int launch_spmm_tc16(const int* blk_offsets, const uint32_t* hspa_packed, const int* hind, int num_nodes,
                            int embedding_dim, const void* input /* SpmmArgs<T>::in_t[rows][embedding_dim] */,
                            float* output, hipStream_t stream, const int* window_order = nullptr,
                            const float* out_scale = nullptr, int atomic_out = 0,
                            const int* units = nullptr /* int32[U][4] */, const int* unit_ptr = nullptr /* int32[9] */,
                            int max_units_per_xcd = 0, float* partials = nullptr, const int* row_map = nullptr,
                            const void* values = nullptr /* WEIGHTED tiles: in_t[T][16][8] */,
                            int units_per_wave = 1 /* 2: paired units (unit table, 16-bit binary operand, FS <= 128, one slab or slab-major order; else ignored) */,
                            int slab_first = 0, int slab_count = 0 /* > 0: only the column slabs [slab_first, slab_first + slab_count) of the operand, one launch;
                                                                       0: all of them -- one launch per slab_launch_group() slabs, or one grid over all */,
                            long long input_rows = 0 /* rows of the dense operand (0: num_nodes, a square adjacency) */,
                            int slab_policy = kSlabAuto) {
  if (num_nodes < 0 || embedding_dim < 0) return kErrBadShape;
  if (num_nodes == 0 || embedding_dim == 0) return kOk;
  if (embedding_dim % (16 / T::EB) != 0) return kErrBadShape;  // 16-byte row chunks
  if (((uintptr_t)input & 15) || ((uintptr_t)hspa_packed & 15)) return kErrBadShape;
  SpmmArgs<T> a;
  a.blk_offsets = blk_offsets;
  a.hspa_packed = hspa_packed;
  a.hind = hind;
  a.input = static_cast<const typename SpmmArgs<T>::in_t*>(input);
  a.output = output;
  a.num_nodes = num_nodes;
  a.num_windows = (num_nodes + kBlkH - 1) / kBlkH;
  a.F = embedding_dim;
  const int total_slabs = (embedding_dim + T::FS - 1) / T::FS;
  if (slab_first < 0 || slab_count < 0 || slab_first + slab_count > total_slabs) return kErrBadShape;
  if (const int group = slab_count == 0 ? slab_launch_group(total_slabs, T::FS * T::EB,
                                                            input_rows > 0 ? input_rows : (long long)num_nodes, slab_policy)
                                        : 0) {
    for (int s = 0; s < total_slabs; s += group) {
      const int rc = launch_spmm_tc16<T>(blk_offsets, hspa_packed, hind, num_nodes, embedding_dim, input, output, stream,
                                         window_order, out_scale, atomic_out, units, unit_ptr, max_units_per_xcd, partials,
                                         row_map, values, units_per_wave, s, total_slabs - s < group ? total_slabs - s : group);
      if (rc != kOk) return rc;
    }
    return kOk;
  }
  a.slab_first = slab_count > 0 ? slab_first : 0;
  a.num_slabs = slab_count > 0 ? slab_count : total_slabs;
  a.windows_per_xcd = (a.num_windows + kNumXcd - 1) / kNumXcd;
  a.window_order = window_order;
  a.out_scale = out_scale;
  a.atomic_out = atomic_out;
  a.meta_nt = total_slabs == 1;   // several slabs (in one launch or one launch each) read the metadata again: keep it cached
  a.slab_major = slab_major_order(a.num_slabs, T::FS * T::EB);
  a.units = reinterpret_cast<const int4*>(units);
  a.unit_ptr = unit_ptr;
  a.partials = partials;
  a.row_map = row_map;
  a.values = static_cast<const typename SpmmArgs<T>::in_t*>(values);
  if (T::WEIGHTED && (values == nullptr || ((uintptr_t)values & 15))) return kErrBadShape;
  if (units != nullptr) {
    if (unit_ptr == nullptr || max_units_per_xcd < 0 || ((uintptr_t)units & 15)) return kErrBadShape;
    if (max_units_per_xcd == 0) return kOk;
    a.windows_per_xcd = max_units_per_xcd;  // sizes the grid below; the kernel reads its range from unit_ptr
  }
  if ((long long)a.windows_per_xcd * a.num_slabs > 0x3FFFFFFFll) return kErrBadShape;  // int unit counters
  if constexpr (T::EB == 2 && !T::WEIGHTED && T::FS <= 128) {
    if (units_per_wave == 2 && units != nullptr && (a.num_slabs == 1 || a.slab_major)) {
      const long long pairs = ((long long)a.windows_per_xcd + 1) / 2 * a.num_slabs;   // pairs never straddle a slab
      const long long blocks_per_xcd = (pairs + T::WAVES - 1) / T::WAVES;
      const long long grid = blocks_per_xcd * kNumXcd;
      if (grid > 0x7FFFFFFFll) return kErrBadShape;
      const int lds_rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&spmm_tc16_pair_kernel<T>), T::BLOCK_LDS);
      if (lds_rc != kOk) return lds_rc;
      hipLaunchKernelGGL(spmm_tc16_pair_kernel<T>, dim3((unsigned)grid), dim3(T::THREADS), T::BLOCK_LDS, stream, a);
      return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
    }
  }
  const long long blocks_per_xcd = ((long long)a.windows_per_xcd * a.num_slabs + T::WAVES - 1) / T::WAVES;
  const long long grid = blocks_per_xcd * kNumXcd;
  if (grid > 0x7FFFFFFFll) return kErrBadShape;
  const int lds_rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&spmm_tc16_kernel<T>), T::BLOCK_LDS);
  if (lds_rc != kOk) return lds_rc;
  hipLaunchKernelGGL(spmm_tc16_kernel<T>, dim3((unsigned)grid), dim3(T::THREADS), T::BLOCK_LDS, stream, a);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// ----------------------------------------------------------------------------------------------
// Cut windows (unit table, slot >= 0): the units of one window leave their partial [16][F] tiles in consecutive slots
// of `partials`; this pass sums them IN UNIT ORDER (a fixed order: the result does not depend on which unit finished
// first) and stores the sum to C -- or adds it onto C when `accumulate` (two-level format: C then holds the panel
// kernel's part, complete, because this pass runs after the join).  cuts: int32[num_cuts][4] = {window, first slot,
// units, 0}.  One workgroup per cut window; F % 4 == 0.
static __global__ __launch_bounds__(256) void combine_partials_kernel(const int4* __restrict__ cuts,
                                                                     const float* __restrict__ partials,
                                                                     float* __restrict__ output, const int num_nodes,
                                                                     const int F, const int accumulate,
                                                                     const int* __restrict__ row_map) {
  const int4 c = cuts[blockIdx.x];
  const int f4 = F / 4;
  const long long tile4 = (long long)kBlkH * f4;
  const float4* src = reinterpret_cast<const float4*>(partials) + (long long)c.y * tile4;
  for (int i = threadIdx.x; i < (int)tile4; i += 256) {
    int row = c.x * kBlkH + i / f4;
    if (row_map) row = row_map[row];
    if (row < 0 || row >= num_nodes) continue;  // tail window / padding rows
    float4 sum = src[i];
    int u = 1;
    // eight tiles' loads in flight per round trip, added in unit order (the same bits as one load per addition; a hub window
    // of the web-BerkStan-like stand-in has ~85 units: the serial chain was 62 us, 18 % of that graph's step)
    for (; u + 8 <= c.z; u += 8) {
      float4 p[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) p[k] = src[(long long)(u + k) * tile4 + i];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        sum.x += p[k].x;
        sum.y += p[k].y;
        sum.z += p[k].z;
        sum.w += p[k].w;
      }
    }
    for (; u < c.z; ++u) {
      const float4 p = src[(long long)u * tile4 + i];
      sum.x += p.x;
      sum.y += p.y;
      sum.z += p.z;
      sum.w += p.w;
    }
    float4* const dst = reinterpret_cast<float4*>(output + (long long)row * F) + (i % f4);
    if (accumulate) {
      const float4 d = *dst;
      sum.x += d.x;
      sum.y += d.y;
      sum.z += d.z;
      sum.w += d.w;
    }
    *dst = sum;
  }
}

inline int combine_partials(const int* cuts, int num_cuts, const float* partials, float* output, int num_nodes,
                            int embedding_dim, int accumulate, hipStream_t stream, const int* row_map = nullptr) {
  if (num_cuts < 0 || num_nodes < 0 || embedding_dim < 0 || (embedding_dim % 4) != 0) return kErrBadShape;
  if (num_cuts == 0 || num_nodes == 0 || embedding_dim == 0) return kOk;
  if (((uintptr_t)cuts & 15) || ((uintptr_t)partials & 15) || ((uintptr_t)output & 15)) return kErrBadShape;
  hipLaunchKernelGGL(combine_partials_kernel, dim3((unsigned)num_cuts), dim3(256), 0, stream,
                     reinterpret_cast<const int4*>(cuts), partials, output, num_nodes, embedding_dim, accumulate, row_map);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// ----------------------------------------------------------------------------------------------
// "Balance" schedule (no reference counterpart).  Windows that run side by side on an XCD
// share gathered rows through its L2 only while they sweep their (sorted) columns at a similar pace, i.e. while they
// have a similar number of TC blocks.  window_order[pos] lists, inside every XCD's window range and inside chunks of
// `chunk` consecutive windows (row locality is kept at that granularity), the windows by descending block count.
// Measured on MI355X: -7 % (reddit-like) ... -19 % (uniform columns) kernel time; results are bit-identical.
constexpr int kOrderMaxChunk = 4096;

static __global__ __launch_bounds__(256) void window_order_kernel(const int* __restrict__ blk_offsets,
                                                                 const int num_windows, const int windows_per_xcd,
                                                                 const int chunk, const int chunks_per_xcd,
                                                                 int* __restrict__ order) {
  __shared__ unsigned long long keys[kOrderMaxChunk];
  const int xcd = blockIdx.x / chunks_per_xcd, c = blockIdx.x % chunks_per_xcd;
  const int x_begin = xcd * windows_per_xcd;
  const int x_end = (x_begin + windows_per_xcd < num_windows) ? x_begin + windows_per_xcd : num_windows;
  const int begin = x_begin + c * chunk;
  const int end = (begin + chunk < x_end) ? begin + chunk : x_end;
  const int n = end - begin;
  if (n <= 0) return;  // workgroup-uniform
  int P = 1;
  while (P < n) P <<= 1;
  for (int i = threadIdx.x; i < P; i += 256) {
    unsigned long long k = ~0ull;  // padding sorts last
    if (i < n) {
      const unsigned nblk = (unsigned)(blk_offsets[begin + i + 1] - blk_offsets[begin + i]);
      k = ((unsigned long long)(0xFFFFFFFFu - nblk) << 32) | (unsigned)i;  // descending length, then ascending index
    }
    keys[i] = k;
  }
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < P; i += 256) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long x = keys[i], y = keys[l];
          const bool up = (i & k) == 0;
          if ((x > y) == up) {
            keys[i] = y;
            keys[l] = x;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < n; i += 256) order[begin + i] = begin + (int)(keys[i] & 0xFFFFFFFFu);
}

inline int launch_window_order(const int* blk_offsets, int num_nodes, int chunk, int* order, hipStream_t stream) {
  if (num_nodes < 0 || chunk < 1 || chunk > kOrderMaxChunk) return kErrBadShape;
  const int num_windows = (num_nodes + kBlkH - 1) / kBlkH;
  if (num_windows == 0) return kOk;
  const int windows_per_xcd = (num_windows + kNumXcd - 1) / kNumXcd;  // must match launch_spmm_tc16
  const int chunks_per_xcd = (windows_per_xcd + chunk - 1) / chunk;
  hipLaunchKernelGGL(window_order_kernel, dim3(kNumXcd * chunks_per_xcd), dim3(256), 0, stream, blk_offsets, num_windows,
                     windows_per_xcd, chunk, chunks_per_xcd, order);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// fp32 -> fp16 cast of the dense operand (the reference rounds B to TF32 in-kernel, spmm_kernels.cuh:1671; gfx950
// has no TF32, fp16 keeps the same 10-bit mantissa -- SURVEY.md section 8c "tolerance evidence").
static __global__ __launch_bounds__(256) void cast_f32_to_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst,
                                                              const long long n8) {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const float4 x = reinterpret_cast<const float4*>(src)[2 * i];
    const float4 y = reinterpret_cast<const float4*>(src)[2 * i + 1];
    h8 o = {(_Float16)x.x, (_Float16)x.y, (_Float16)x.z, (_Float16)x.w,
            (_Float16)y.x, (_Float16)y.y, (_Float16)y.z, (_Float16)y.w};
    reinterpret_cast<h8*>(dst)[i] = o;
  }
}

inline int cast_f32_to_f16(const float* src, _Float16* dst, long long n, hipStream_t stream) {
  if (n < 0 || (n % 8) != 0 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return kErrBadShape;
  if (n == 0) return kOk;
  const long long n8 = n / 8;
  const int blocks = (int)(n8 / 256 + 1 < 256 * 16 ? n8 / 256 + 1 : 256 * 16);
  hipLaunchKernelGGL(cast_f32_to_f16_kernel, dim3(blocks), dim3(256), 0, stream, src, dst, n8);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// ---- rows of a dense matrix times a per-row factor (round 6: edge values that factor as r_i c_j, voltrix/weighted.py) ----------
// dst[i, :] = T(float(src[i, :]) * scale[i]); in place allowed (dst == src).  16 bytes per lane; rows are 16-byte multiples.
// HBM-bound, one pass: B of the headline graph (60 MB fp16) in ~25 us, C (119 MB fp32) in ~45 us.
template <typename T, int V>   // V elements of T = 16 bytes
static __global__ __launch_bounds__(256) void scale_rows_kernel(const T* __restrict__ src, const float* __restrict__ scale,
                                                                T* __restrict__ dst, const long long chunks, const int chunks_per_row) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += stride) {
    const float f = scale[i / chunks_per_row];
    uint4_t raw = reinterpret_cast<const uint4_t*>(src)[i];
    if constexpr (std::is_same<T, float>::value) {
      float4_t v = __builtin_bit_cast(float4_t, raw);
      v *= f;
      raw = __builtin_bit_cast(uint4_t, v);
    } else if constexpr (std::is_same<T, _Float16>::value) {
      half8_t v = __builtin_bit_cast(half8_t, raw);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = (_Float16)((float)v[k] * f);
      raw = __builtin_bit_cast(uint4_t, v);
    } else {   // bfloat16 as bits: widen, multiply, round to nearest even
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned w = raw[k];
        float lo = __builtin_bit_cast(float, w << 16) * f, hi = __builtin_bit_cast(float, w & 0xffff0000u) * f;
        unsigned bl = __builtin_bit_cast(unsigned, lo), bh = __builtin_bit_cast(unsigned, hi);
        bl = (bl + 0x7fffu + ((bl >> 16) & 1u)) >> 16;
        bh = (bh + 0x7fffu + ((bh >> 16) & 1u)) & 0xffff0000u;
        raw[k] = bh | bl;
      }
    }
    reinterpret_cast<uint4_t*>(dst)[i] = raw;
  }
}

// dtype: 0 fp32, 1 fp16, 2 bfloat16.  num_feats elements per row; a row must be a multiple of 16 bytes.
inline int scale_rows(const void* src, const float* scale, void* dst, long long rows, int num_feats, int dtype, hipStream_t stream) {
  const int per_chunk = dtype == 0 ? 4 : 8;
  if (rows < 0 || num_feats < 0 || dtype < 0 || dtype > 2 || num_feats % per_chunk) return kErrBadShape;
  if (rows == 0 || num_feats == 0) return kOk;
  if (src == nullptr || dst == nullptr || scale == nullptr || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return kErrBadShape;
  const int cpr = num_feats / per_chunk;
  const long long chunks = rows * cpr;
  const int blocks = (int)(chunks / 256 + 1 < 256 * 16 ? chunks / 256 + 1 : 256 * 16);
  if (dtype == 0)
    hipLaunchKernelGGL((scale_rows_kernel<float, 4>), dim3(blocks), dim3(256), 0, stream, static_cast<const float*>(src), scale,
                       static_cast<float*>(dst), chunks, cpr);
  else if (dtype == 1)
    hipLaunchKernelGGL((scale_rows_kernel<_Float16, 8>), dim3(blocks), dim3(256), 0, stream, static_cast<const _Float16*>(src),
                       scale, static_cast<_Float16*>(dst), chunks, cpr);
  else
    hipLaunchKernelGGL((scale_rows_kernel<bfloat16_bits, 8>), dim3(blocks), dim3(256), 0, stream,
                       static_cast<const bfloat16_bits*>(src), scale, static_cast<bfloat16_bits*>(dst), chunks, cpr);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// ---- range-safe fp32 -> fp16 operand: dst = fp16(src * 2^-e), scale[0] = 2^e with e = exponent(max |src|) - 14 --------
// The reference multiplies in TF32 (8-bit exponent); a plain fp16 cast would overflow above 65504 and flush below
// 6e-8.  A binary A makes the product linear in B, so one power-of-two scale per launch (exact, undone in the SpMM
// epilogue through SpmmArgs::out_scale) moves the operand's largest magnitude to [2^14, 2^15) and keeps the 10-bit
// mantissa (= TF32's) for everything within 2^-28 of it.  All on the stream, no host sync.  Inf / NaN operands:
// scale 1 (they propagate as in fp32).  scale[1] is scratch (max |src| bits).
// NO atomics (round 5): every workgroup leaves its maximum in a scratch word and a one-workgroup kernel reduces the words.  The
// scratch is the head of `dst` itself -- the convert pass overwrites it afterwards, in stream order.  History: one atomicMax
// per wave, 16 k atomics onto one address, was 0.18 ms of a 0.23 ms cast (round 4, profiles/r04/experiment_cast.log); one per
// workgroup "only when it can raise the maximum" still serialised the first thousands that finish together while the maximum
// is 0: 47 us of the 58 us cast of a 29 MB operand (ppi-like, rocprofv3 kernel stats, round 5).
static __global__ __launch_bounds__(256) void amax_abs_f32_kernel(const float* __restrict__ src, const long long n4,
                                                           unsigned* __restrict__ block_max) {
  __shared__ unsigned wave_max[256 / kWave];
  const long long stride = (long long)gridDim.x * blockDim.x;
  unsigned m = 0u;  // |x| as bits: non-negative floats order like unsigned integers (NaN > Inf > finite)
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {          // two independent 16-byte loads in flight per thread
    const uint4 x = reinterpret_cast<const uint4*>(src)[i];
    const uint4 y = reinterpret_cast<const uint4*>(src)[i + stride];
    const unsigned a = x.x & 0x7FFFFFFFu, b = x.y & 0x7FFFFFFFu, c = x.z & 0x7FFFFFFFu, d = x.w & 0x7FFFFFFFu;
    const unsigned e = y.x & 0x7FFFFFFFu, f = y.y & 0x7FFFFFFFu, g = y.z & 0x7FFFFFFFu, h = y.w & 0x7FFFFFFFu;
    const unsigned ab = a > b ? a : b, cd = c > d ? c : d, ef = e > f ? e : f, gh = g > h ? g : h;
    const unsigned abcd = ab > cd ? ab : cd, efgh = ef > gh ? ef : gh, all = abcd > efgh ? abcd : efgh;
    m = m > all ? m : all;
  }
  for (; i < n4; i += stride) {
    const uint4 x = reinterpret_cast<const uint4*>(src)[i];
    const unsigned a = x.x & 0x7FFFFFFFu, b = x.y & 0x7FFFFFFFu, c = x.z & 0x7FFFFFFFu, d = x.w & 0x7FFFFFFFu;
    const unsigned ab = a > b ? a : b, cd = c > d ? c : d, abcd = ab > cd ? ab : cd;
    m = m > abcd ? m : abcd;
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)m, off, kWave);
    m = m > o ? m : o;
  }
  if ((threadIdx.x & (kWave - 1)) == 0) wave_max[threadIdx.x / kWave] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < 256 / kWave; ++w) m = m > wave_max[w] ? m : wave_max[w];
    block_max[blockIdx.x] = m;
  }
}

// scale[1] <- max of the `blocks` workgroup maxima (bits of a non-negative float); one workgroup.  blocks == 0: 0.
static __global__ __launch_bounds__(256) void amax_finish_kernel(const unsigned* __restrict__ block_max, const int blocks,
                                                                 float* __restrict__ scale) {
  __shared__ unsigned wave_max[256 / kWave];
  unsigned m = 0u;
  for (int i = threadIdx.x; i < blocks; i += 256) m = m > block_max[i] ? m : block_max[i];
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)m, off, kWave);
    m = m > o ? m : o;
  }
  if ((threadIdx.x & (kWave - 1)) == 0) wave_max[threadIdx.x / kWave] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < 256 / kWave; ++w) m = m > wave_max[w] ? m : wave_max[w];
    reinterpret_cast<unsigned*>(scale)[1] = m;
  }
}

__device__ __forceinline__ int operand_scale_exponent(const unsigned amax_bits) {
  if (amax_bits == 0u || amax_bits >= 0x7F800000u) return 0;   // all zero, or Inf / NaN present: leave as is
  const int e = (int)(amax_bits >> 23) - 127 - 14;              // subnormal maxima: biased exponent 0 -> e = -141
  return e < -100 ? -100 : e;                                   // 2^100 / 2^-113 stay normal floats
}

static __global__ __launch_bounds__(256) void cast_f32_to_f16_scaled_kernel(const float* __restrict__ src,
                                                                      _Float16* __restrict__ dst, const long long n8,
                                                                      float* __restrict__ scale) {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  const int e = operand_scale_exponent(reinterpret_cast<const unsigned*>(scale)[1]);
  const float down = __builtin_ldexpf(1.0f, -e);
  if (blockIdx.x == 0 && threadIdx.x == 0) scale[0] = __builtin_ldexpf(1.0f, e);
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const float4 x = reinterpret_cast<const float4*>(src)[2 * i];
    const float4 y = reinterpret_cast<const float4*>(src)[2 * i + 1];
    h8 o = {(_Float16)(x.x * down), (_Float16)(x.y * down), (_Float16)(x.z * down), (_Float16)(x.w * down),
            (_Float16)(y.x * down), (_Float16)(y.y * down), (_Float16)(y.z * down), (_Float16)(y.w * down)};
    reinterpret_cast<h8*>(dst)[i] = o;
  }
}

// scale: device float[2], 8-byte aligned.  scale[0] <- 2^e (pass it to launch_spmm_tc16 as out_scale).
inline int cast_f32_to_f16_scaled(const float* src, _Float16* dst, long long n, float* scale, hipStream_t stream) {
  if (n < 0 || (n % 8) != 0 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15) || scale == nullptr ||
      ((uintptr_t)scale & 7))
    return kErrBadShape;
  const long long n8 = n / 8;
  const int blocks = (int)(n8 / 256 + 1 < 256 * 16 ? n8 / 256 + 1 : 256 * 16);
  // three kernels of one stream (kernel nodes are strictly ordered, also inside a captured graph): workgroup maxima into the
  // head of dst (blocks x 4 bytes <= 2 n: at most one word per 2048 elements), their maximum, the conversion
  unsigned* const block_max = reinterpret_cast<unsigned*>(dst);
  if (n > 0) hipLaunchKernelGGL(amax_abs_f32_kernel, dim3(blocks), dim3(256), 0, stream, src, n / 4, block_max);
  hipLaunchKernelGGL(amax_finish_kernel, dim3(1), dim3(256), 0, stream, block_max, n > 0 ? blocks : 0, scale);
  hipLaunchKernelGGL(cast_f32_to_f16_scaled_kernel, dim3(blocks), dim3(256), 0, stream, src, dst, n8, scale);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace voltrix
