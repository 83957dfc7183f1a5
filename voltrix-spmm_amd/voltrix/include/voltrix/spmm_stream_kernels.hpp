// Voltrix-SpMM for MI355X (gfx950 / CDNA4) -- the window format as a STREAM OF STAGES: the kernel for short row windows.
//
// Same math, same (blk_offsets, hspa_packed, hind) handle and the same stage (4 TC blocks = 32 gathered rows of B = one
// K = 32 MFMA step per 16 columns) as spmm_tc16_kernel (spmm_kernels.hpp; reference spmm_mma161616_spa_swizzle_d/_dd,
// voltrix/include/voltrix/spmm_kernels.cuh:1458-2001, one CTA per row window).  What changes is who owns what.  There a
// wave owns ONE window: on the graphs the reference is evaluated on (bench/plot.py:8) eight of twelve have mean degree
// 2-12, a window is 1-6 stages long, and a wave's life is a prologue (two dependent round trips: metadata, then rows), one
// or two MFMA steps and a 4-byte-per-lane store epilogue -- the chip idles on latency (28-34 % of the HBM roofline, round-5
// baseline).  Here a wave owns a RUN: up to 64 consecutive units (whole short windows; a long window's interleaved
// pieces) whose stages it walks as one stream through one LDS ring:
//
//   * the ring never drains at a window boundary: while stage t is multiplied, the rows of stage t + D and the metadata of
//     stage t + 2 D are in flight whatever window they belong to (three cursors over the run's unit table, which lives in
//     five VGPRs -- lane i holds unit i -- and is read with v_readlane: no LDS, no scalar loads in the loop);
//   * a window that ends inside the stream is stored from the loop (the accumulators restart at zero) and the stream goes
//     on; stores and loads share the wave's vmcnt counter, so the counted wait takes the stores of the last D steps into
//     account (run-time count, wait_vm); a unit's first MFMAs take the constant 0 as their C operand and the adjacency
//     bits expand to 1.0 (not the window kernel's 2.0), so nothing clears or rescales accumulators between windows;
//   * the MFMA operands are SWAPPED (features as the A operand, adjacency bits as the B operand: D' = (A_adj B)^T): a lane
//     then holds FOUR CONSECUTIVE COLUMNS of one output row and the epilogue is one global_store_dwordx4 per 16-column
//     slot (1 KiB per instruction) instead of four global_store_dword -- C is the dominant stream of these graphs;
//   * every stage may be a window's last one, so the padded-column fix-up (padded hind slots are 0 in the format and must
//     not gather B[0]) is ballot + select on every stage, not a shuffle path for "tail" stages.
//
// Tables (voltrix/schedule.py::stream_tables, built once per handle): units int32 [U][8] = {first TC block, end TC block
// of the window, TC blocks between two stages of the unit (4 x stride), window, partial-tile slot or -1, stages, columns
// of the window's last TC block that carry an edge (0 = a window without edges), 0},
// runs int32 [R][4] = {first unit, units, stages, 0}, run_ptr int32 [9] = the runs of every XCD (contiguous windows, equal
// work), cuts for combine_partials (windows longer than the cut length are split into interleaved units exactly like the
// unit table of spmm_kernels.hpp; their partial tiles are summed in unit order by the same combine pass).
// Binary A, plain stores; B in fp16 / bf16 (v_mfma_f32_16x16x32) or, round 5, exact fp32 (v_mfma_f32_16x16x4_f32).
#pragma once

#include "voltrix/spmm_kernels.hpp"

// Diagnostic builds only (-DVOLTRIX_EXPERIMENTAL; harness/experiments/exp_stream_diag.py): bit 0 skips the B fragment reads and
// the MFMAs, bit 1 the stores, bit 2 folds every gathered row into the first 1024 rows of B, bit 3 skips the row gathers.
// Results are wrong by design; the shipped kernels use 0.
#ifndef VOLTRIX_STREAM_DIAG
#define VOLTRIX_STREAM_DIAG 0
#endif

namespace voltrix {

// Two adjacency nibbles -> four packed fp16x2 registers holding 1.0 / 0.0 (0x3C00 in fp16; bfloat16: 0x3F80).  One AND + one
// full-rate 24-bit multiply per register: bit i and bit 16 + i of z = column pair, (z & (0x00010001 << k)) * (ONE >> k).  The
// window kernel's 2.0 = one bit needs the accumulators scaled by 0.5 at the store; with 1.0 the stream kernel stores them as
// they are.
template <bool BF16>
__device__ __forceinline__ half8_t nibbles_to_ones_x2(unsigned nl, unsigned nh) {
  constexpr unsigned ONE = BF16 ? 0x3F80u : 0x3C00u;
  unsigned zl, zh;
  asm("v_lshl_or_b32 %0, %1, 15, %1" : "=v"(zl) : "v"(nl));   // bit i and bit 15 + i = column i
  asm("v_lshl_or_b32 %0, %1, 15, %1" : "=v"(zh) : "v"(nh));
  uint4_t r;
  r[0] = __umul24(zl & 0x00010001u, ONE);          // columns 0 (bit 0), 1 (bit 16)
  r[1] = __umul24(zl & 0x00040004u, ONE >> 2);     // columns 2 (bit 2), 3 (bit 18)
  r[2] = __umul24(zh & 0x00010001u, ONE);
  r[3] = __umul24(zh & 0x00040004u, ONE >> 2);
  return __builtin_bit_cast(half8_t, r);
}

template <class T>
struct StreamArgs {
  using in_t = typename SpmmArgs<T>::in_t;
  const uint32_t* hspa_packed;   // [4T]
  const int* hind;               // [8T]
  const in_t* input;             // [rows of B][F]
  float* output;                 // [N][F]
  float* partials;               // [slots][16][F] (units with slot >= 0)
  const int4* units;             // [U][2]
  const int4* runs;              // [R]
  const int* run_ptr;            // [9]
  const float* out_scale;        // optional device scalar
  int num_nodes;
  int F;
  int num_slabs;                 // column slabs of this launch
  int slab_first;
  int slab_major;
  int meta_nt;
  int max_runs_per_xcd;
};

// LDS image of a stage.  DMA i (1 KiB) of a stage writes the LDS rows i * RPD + q, q = lane / LANES_PER_ROW; here lane group q
// gathers the condensed columns NDMA * q + i, i.e. the NDMA columns of one lane group are CONSECUTIVE words of hind (one or two
// 16-byte LDS reads fetch the row ids of all the lane's DMAs; the window kernel reads one word per DMA).  Column c therefore
// lives in LDS row stream_row(c); the transposed fragment reads follow it (lane 16 g + 4 q' + p supplies the row of column
// 8 g + q', then of column 8 g + 4 + q': K position = condensed column, the A fragment is the window kernel's).
template <int NDMA, int RPD>
__device__ __forceinline__ constexpr int stream_row(int c) { return (c % NDMA) * RPD + c / NDMA; }
// slot swizzle of that image (logical 32-byte slot s of LDS row r at physical slot s ^ stream_swizzle(r)): conflict-free
// transposed reads at FS = 128 (rows {4 q' + g}: 8 distinct slots per 32-lane half), 2-way at FS = 64 / 32 (two lane groups of
// a half land on rows of one parity; 8 / 4 fragment reads per stage there, LDS is not their bound)
template <int FS, int EB = 2>
__device__ __forceinline__ constexpr int stream_swizzle(int r) {
  if (EB == 4)   // fp32 rows (below): lane groups g and g + 1 of a 32-lane half read rows 4 (8) apart -- two slots apart
    return FS >= 64 ? (((r >> 2) & 3) << 1) : (((r >> 3) & 1) << 1);
  return FS >= 128 ? (((r >> 2) & 3) | ((r & 1) << 2)) : (FS == 64 ? ((r >> 3) & 3) : ((r >> 4) & 1));
}
// EXACT fp32 operand (EB = 4; round 5): rows of B gathered as they are (no cast pass, no power-of-two scale -- on the
// low-degree graphs the two passes of the cast moved more bytes than the product), multiplied on v_mfma_f32_16x16x4_f32
// (exact products of {0, 1} x fp32, fp32 accumulation: the reference rounds B to TF32, spmm_kernels.cuh:1671).  K step m of a
// stage covers the condensed columns 4 m + g (lane group g): LDS row stream_row(4 m + g), float column 16 s + R, one
// ds_read_b32 per (m, slot); the adjacency bit of (row R, column 4 m + g) becomes 1.0f / 0.0f with a bit-field extract and a
// convert.  The matrix pipe runs at 1/16 of the fp16 rate there -- 32 MFMAs of 32 cycles per stage -- which is still above what
// the memory path feeds a CU on the graphs this kernel is for.

// ADDR64 = false: B is smaller than 4 GiB and has fewer than 2^24 rows -- a gathered row's address is a scalar base plus ONE
// full-rate v_mad_u32_u24 (row * row bytes + lane constant); true: 64-bit per-lane pointers (v_mad_u64_u32), any size.
// One global_store_dwordx4 exactly where it is written (see store_unit).  byte_offset: an immediate (0 .. 4095).
// The s_nop is the hazard the compiler handles for its OWN stores and cannot see inside an asm: a VMEM store of more than 64 bits
// reads its data registers after issue, and a VALU write of them within two wait states (gfx940 family; LLVM
// GCNHazardRecognizer "VmemStoreHazard") corrupts the stored value -- the scaled path reuses one temporary for every slot
// (measured: tests/test_gpu_stream.py fp16-scaled failed without it).
__device__ __forceinline__ void store_f32x4(float* base, const float4_t v, const int byte_offset) {
  asm volatile("global_store_dwordx4 %0, %1, off offset:%2\n\ts_nop 1" ::"v"(base), "v"(v), "n"(byte_offset) : "memory");
}

template <class T, bool ADDR64>
static __global__ __launch_bounds__(T::THREADS) void spmm_stream_kernel(const StreamArgs<T> a) {
  static_assert((T::EB == 2 || (T::EB == 4 && !T::BF16 && T::FS <= 64)) && !T::WEIGHTED, "stream kernel: binary A, fp16 / bf16 / fp32 B");
  constexpr int FS = T::FS, D = T::DEPTH, MS = T::META_SLOTS, EB = T::EB;
  constexpr int ROW_BYTES = T::ROW_BYTES, STAGE_BYTES = T::STAGE_BYTES, NDMA = T::DMA_PER_STAGE;
  constexpr int RPD = T::ROWS_PER_DMA, LPR = T::LANES_PER_ROW, SLOTS = T::SLOTS;
  constexpr int OPS = 1 + NDMA;   // vector-memory loads per step: one metadata DMA + the row DMAs
  static_assert(NDMA == 8 || NDMA == 4 || NDMA == 2, "FS 128 / 64 / 32");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
  const int xcd = blockIdx.x % kNumXcd;
  const int r_begin = a.run_ptr[xcd];
  const int r_count = a.run_ptr[xcd + 1] - r_begin;
  const int wave_pos = (int)(blockIdx.x / kNumXcd) * T::WAVES + wave;
  if (r_count <= 0 || wave_pos >= r_count * a.num_slabs) return;   // wave-uniform; the kernel has no barrier
  const int rpos = r_begin + (a.slab_major ? wave_pos % r_count : wave_pos / a.num_slabs);
  const int fs0 = (a.slab_first + (a.slab_major ? wave_pos / r_count : wave_pos % a.num_slabs)) * FS;
  const int F = a.F;
  const int4 run = a.runs[rpos];
  const int nu = __builtin_amdgcn_readfirstlane(run.y);
  const int nst = __builtin_amdgcn_readfirstlane(run.z);
  if (nu <= 0 || nst <= 0) return;

  // ---- the run's unit table: lane i holds unit i (lanes past the run: its last unit) -------------------------------
  int u_first, u_end, u_step, u_win, u_slot, u_ncl;
  {
    const int ui = run.x + (lane < nu ? lane : nu - 1);
    const int4 q0 = a.units[2ll * ui];
    const int4 q1 = a.units[2ll * ui + 1];
    u_first = q0.x;
    u_end = q0.y;
    u_step = q0.z;
    u_win = q0.w;
    u_slot = q1.x;
    u_ncl = q1.z;   // columns of the window's LAST TC block that carry an edge (1 .. 8; 0: a window without edges)
  }
  struct Cursor {
    int u, blk, end, step, ncl;
  };
  auto cur_load = [&](Cursor& c, int u) {
    c.u = u;
    c.blk = __builtin_amdgcn_readlane(u_first, u);
    c.end = __builtin_amdgcn_readlane(u_end, u);
    c.step = __builtin_amdgcn_readlane(u_step, u);
    c.ncl = __builtin_amdgcn_readlane(u_ncl, u);
  };
  auto cur_advance = [&](Cursor& c) {   // next stage of the stream; past the run's end the cursor stays on its last stage
    const int nb = c.blk + c.step;
    if (nb < c.end) {
      c.blk = nb;
    } else if (c.u + 1 < nu) {
      cur_load(c, c.u + 1);
    }
  };

  float4_t acc[SLOTS];
  bool fresh = true;   // the stream is at a unit's first stage

  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem + (unsigned)wave * T::WAVE_LDS;
  const unsigned meta0 = lds0 + D * STAGE_BYTES;

  // ---- lane constants ------------------------------------------------------------------------------------------------
  const int k32 = lane & 31;
  const int kblk = k32 >> 3, kcol = k32 & 7;
  const int g = lane >> 4, R = lane & 15;
  const int q = lane / LPR;                  // lane group of the row DMAs: LDS row i * RPD + q, condensed column NDMA * q + i
  const unsigned a_shift = 4 * (R & 7);
  const int mj = (lane - 32) & 15;
  unsigned tr_base = 0;
  int tr_delta[4] = {0, 0, 0, 0};
  constexpr int TR_SECOND = (stream_row<NDMA, RPD>(4) - stream_row<NDMA, RPD>(0)) * ROW_BYTES;   // column c + 4 after column c
  static_assert(stream_row<NDMA, RPD>(13) - stream_row<NDMA, RPD>(9) == stream_row<NDMA, RPD>(4) - stream_row<NDMA, RPD>(0), "lane-invariant");
  // fp32 operand: lane (g, R) reads float column 16 s + R of the LDS row of condensed column 4 m + g.  That row is
  // F32_LANE_ROW * g + f32_row_const(m) in both geometries, its swizzle 2 gx (gx = g, or g & 1 at 128-byte rows), so the byte
  // address is a per-slot lane base + an immediate of m
  constexpr int F32_LANE_ROW = NDMA == 8 ? 4 : 8;
  unsigned f32_base[SLOTS];
  unsigned aword_base = 0, a_bit = 0;
  if constexpr (EB == 4) {
    static_assert(NDMA == 8 || NDMA == 4, "fp32 rows of 256 or 128 bytes");
    const int gx = NDMA == 8 ? g : (g & 1);
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      f32_base[s] = lds0 + (unsigned)(F32_LANE_ROW * g) * ROW_BYTES + 64u * (unsigned)(s ^ gx) + 4u * (unsigned)R;
      asm volatile("" : "+v"(f32_base[s]));
    }
    aword_base = 128 + 4 * (R >> 3);      // words (R >> 3) and 2 + (R >> 3) of every TC block of the stage
    a_bit = 4 * (R & 7) + g;              // bit of (row R, column 4 (m & 1) + g of block m >> 1) in word (R >> 3) + 2 (m & 1)
  }
  if constexpr (EB == 2) {
    const int trow = stream_row<NDMA, RPD>(8 * g + ((lane >> 2) & 3));
    const int tr_z = stream_swizzle<FS, 2>(trow);
    static_assert(EB != 2 || (stream_swizzle<FS, 2>(stream_row<NDMA, RPD>(4)) == stream_swizzle<FS, 2>(stream_row<NDMA, RPD>(0)) &&
                              stream_swizzle<FS, 2>(stream_row<NDMA, RPD>(15)) == stream_swizzle<FS, 2>(stream_row<NDMA, RPD>(11))),
                  "the second fragment read shares the first one's swizzle");
    tr_base = lds0 + trow * ROW_BYTES + 8 * (lane & 3) + (tr_z << 5);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      tr_delta[b] = ((tr_z >> b) & 1) ? -(32 << b) : (32 << b);
      asm volatile("" : "+v"(tr_delta[b]));
    }
    asm volatile("" : "+v"(tr_base));
  }
  const char* const meta_base = lane < 32 ? (const char*)(a.hind + k32) : (const char*)(a.hspa_packed + mj);
  const unsigned meta_blk_bytes = lane < 32 ? 4u * 8u : 4u * 4u;
  auto issue_meta = [&](const Cursor& c, int mslot) {
    const void* src;
    const int sb = c.blk, kb1 = c.end;
    if (sb + kTcbPerStage <= kb1) {  // wave-uniform
      src = meta_base + (unsigned long long)(unsigned)sb * meta_blk_bytes;
    } else if (lane < 32) {
      int blk = sb + kblk;
      blk = blk < kb1 ? blk : kb1 - 1;   // stay inside the window: the stage's missing blocks re-read its last one
      src = a.hind + (8ll * blk + kcol);
    } else {
      int blk = sb + (mj >> 2);
      blk = blk < kb1 ? blk : kb1 - 1;
      src = a.hspa_packed + (4ll * blk + (mj & 3));
    }
    if (a.meta_nt)
      dma_b32_nt(src, meta0 + mslot * T::META_BYTES);
    else
      dma_b32(src, meta0 + mslot * T::META_BYTES);
  };
  // row gathers: DMA i of a stage reads 16 bytes at column col(i, lane) of row hr[i].  The source column carries the slot
  // swizzle (LDS-DMA writes are lane-linear); the swizzle of row i * RPD + q repeats with period 4 in i, so four lane constants
  // serve all DMAs, and the four DMAs of a 4-KiB group share one M0 through the instruction's immediate offset (which is added
  // to the global address as well: the base is pre-decremented by it).
  const unsigned row_bytes = (unsigned)F * (unsigned)EB;
  constexpr int NBASE = NDMA < 4 ? NDMA : 4;
  constexpr int CPS = 32 / EB;   // columns per 32-byte slot
  static_assert(stream_swizzle<FS, EB>(4 * RPD + 1) == stream_swizzle<FS, EB>(1) || NDMA <= 4, "period of the row swizzle");
  unsigned lanecol[NBASE];      // byte offset of the lane's chunk inside the slab (ADDR64 = false)
  const char* cbase[NBASE];     // per-lane 64-bit bases (ADDR64 = true)
#pragma unroll
  for (int i = 0; i < NBASE; ++i) {
    const int r = i * RPD + q;
    const int c = lane % LPR;
    int col = fs0 + (((c >> 1) ^ stream_swizzle<FS, EB>(r)) * CPS) + (c & 1) * (CPS / 2);
    col = col < F ? col : fs0;
    lanecol[i] = (unsigned)(col - fs0) * (unsigned)EB;
    unsigned long long cb = (unsigned long long)((const char*)a.input + ((long long)col * EB - (i & 3) * 1024));
    if constexpr (ADDR64) {
      asm volatile("" : "+v"(cb));
    } else {
      asm volatile("" : "+v"(lanecol[i]));
    }
    cbase[i] = (const char*)cb;
  }
  // scalar bases of the four DMAs of a 4-KiB group (pre-decremented by the instruction's immediate offset): kept as four
  // SGPR pairs so that every row DMA is `global_load_lds_dwordx4 v_offset, s[base]` -- no 64-bit vector arithmetic
  unsigned long long sbase[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    sbase[k] = (unsigned long long)((const char*)a.input + (long long)fs0 * EB - k * 1024);
    asm volatile("" : "+s"(sbase[k]));
  }
  auto issue_data = [&](int dslot, const int (&hr)[NDMA]) {
    const unsigned dst = lds0 + dslot * STAGE_BYTES;
    auto piece = [&](auto kc, int ib) {
      constexpr int K = decltype(kc)::value;
      if constexpr (K < NDMA) {
        const int i = ib + K;
        const char* src;
        unsigned hrow = (unsigned)hr[i];
        if (VOLTRIX_STREAM_DIAG & 4) hrow &= 1023u;
        if constexpr (ADDR64) {
          src = cbase[i % NBASE] + (unsigned long long)hrow * row_bytes;
        } else {
          src = (const char*)sbase[K] + (__umul24(hrow, row_bytes) + lanecol[i % NBASE]);   // scalar base + 32-bit offset
        }
        if (VOLTRIX_STREAM_DIAG & 8) return;
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
  // rows of a stage from its metadata slot: the lane group's NDMA consecutive hind words (asynchronous LDS reads) ...
  const unsigned hr_off = 4 * NDMA * q;
  auto read_rows = [&](unsigned mbase, int (&hr)[NDMA]) {
    if constexpr (NDMA == 8) {
      const uint4_t lo = lds_read_b128(mbase + hr_off), hi = lds_read_b128(mbase + hr_off + 16);
      hr[0] = lo[0], hr[1] = lo[1], hr[2] = lo[2], hr[3] = lo[3], hr[4] = hi[0], hr[5] = hi[1], hr[6] = hi[2], hr[7] = hi[3];
    } else if constexpr (NDMA == 4) {
      const uint4_t lo = lds_read_b128(mbase + hr_off);
      hr[0] = lo[0], hr[1] = lo[1], hr[2] = lo[2], hr[3] = lo[3];
    } else {
      const uint2_t lo = lds_read_b64(mbase + hr_off);
      hr[0] = lo[0], hr[1] = lo[1];
    }
  };
  // ... and, after lgkmcnt(0), the fix-up of the columns that carry no edge in this window (padded hind slots are 0 in the
  // format and must not gather B[0]; blocks past the window's end were clamped to its last block): valid columns are a PREFIX
  // of the stage -- every TC block but a window's last is full, the last holds `ncl` (unit table) -- and the others take the
  // stage's first column, a row the window references anyway.  Stages inside a window skip it.
  const int col0 = NDMA * q;
  auto fix_rows = [&](const Cursor& c, int (&hr)[NDMA]) {
    const int left = c.end - c.blk;              // TC blocks from this stage to the window's end
    if (left <= kTcbPerStage) {                  // wave-uniform
      const int nv = (left - 1) * kBlkW + c.ncl - col0;   // valid columns of this lane group
      const int hsafe = __builtin_amdgcn_readfirstlane(hr[0]);
#pragma unroll
      for (int i = 0; i < NDMA; ++i) hr[i] = i < nv ? hr[i] : hsafe;
    }
  };

  const float oscale = a.out_scale ? *a.out_scale : 1.0f;   // fp32 features cast with a power-of-two scale; else 1
  int scaled_s = __builtin_amdgcn_readfirstlane(oscale != 1.0f ? 1 : 0);   // wave-uniform
  asm volatile("" : "+s"(scaled_s));
  // columns this lane stores: 4 g .. 4 g + 3 of every 16-column slot (swapped MFMA operands); live slots are wave-uniform
  const int ns_live = (F - fs0 + 15) / 16 < SLOTS ? (F - fs0 + 15) / 16 : SLOTS;
  const bool full_slab = F - fs0 >= FS;
  auto store_unit = [&](int u) {
    const int win = __builtin_amdgcn_readlane(u_win, u);
    const int slot = __builtin_amdgcn_readlane(u_slot, u);
    float* dst;
    bool ok;
    if (slot >= 0) {   // a cut window's partial tile [16][F], summed in unit order by combine_partials_kernel
      dst = a.partials + ((long long)slot * kBlkH + R) * (long long)F;
      ok = true;
    } else {
      int row = win * kBlkH + R;
      if (VOLTRIX_STREAM_DIAG & 32) row = (win & 255) * kBlkH + R;   // diagnostic: every store into the first 4096 rows of C
      ok = row < a.num_nodes;
      dst = a.output + (long long)row * F;
    }
    dst += fs0 + 4 * g;
    auto store_slots = [&](auto scaled_c) {
      constexpr bool SC = decltype(scaled_c)::value;
      auto value = [&](int s) -> float4_t {
        float4_t v = acc[s];
        if constexpr (SC) {
          v[0] *= oscale;
          v[1] *= oscale;
          v[2] *= oscale;
          v[3] *= oscale;
        }
        return v;
      };
      // The stores are INLINE ASM (round 6): the run-time vmcnt budget of the ring counts exactly ns_live store instructions per
      // finished unit, issued after the step's LDS-DMA loads.  A compiler-generated store may be merged, split or hoisted above
      // issue_data; a volatile asm with a memory clobber is one instruction where it stands (tests/test_stream_isa.py reads
      // the disassembly).  A predicate that is false in every lane would skip the instruction (s_cbranch_execz); neither
      // predicate can be: row 0 of a window exists, and column 4 g = 0 of a live slot is below F.
      if (full_slab) {   // wave-uniform: SLOTS store instructions under one row predicate
        if (ok) {
#pragma unroll
          for (int s = 0; s < SLOTS; ++s) store_f32x4(dst, value(s), 64 * s);
        }
      } else {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
          if (s < ns_live) {   // wave-uniform: exactly ns_live store instructions per unit (the wait counts them)
            if (ok && fs0 + 16 * s + 4 * g < F) store_f32x4(dst, value(s), 64 * s);
          }
        }
      }
    };
    if (__builtin_expect(scaled_s != 0, 0))   // a scalar register: a real branch, not 32 selects
      store_slots(std::true_type{});
    else
      store_slots(std::false_type{});
  };

  unsigned long long tk0 = 0, tk = 0;
  unsigned ph[7] = {0, 0, 0, 0, 0, 0, 0};
  auto stamp = [&](int i) {
    if (VOLTRIX_STREAM_DIAG & 16) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      ph[i] += (unsigned)(now - tk);
      tk = now;
    }
  };
  if (VOLTRIX_STREAM_DIAG & 16) tk0 = tk = __builtin_amdgcn_s_memtime();
  // ---- prologue: metadata of stages 0 .. D-1, then (metadata D + j, rows of stage j) for j < D ---------------------------
  wait_vmcnt<0>();   // the unit table (global loads above) is in registers
  Cursor cm, cr, cc;
  cur_load(cm, 0);
  cr = cm;
  cc = cm;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    issue_meta(cm, j);
    cur_advance(cm);
  }
  wait_vmcnt<0>();
#pragma unroll
  for (int j = 0; j < D; ++j) {
    issue_meta(cm, (D + j) % MS);
    cur_advance(cm);
    int hr[NDMA];
    read_rows(meta0 + j * T::META_BYTES, hr);
    wait_lgkmcnt0();
    fix_rows(cr, hr);
    issue_data(j, hr);
    cur_advance(cr);
  }

  stamp(0);
  // ---- the stream ------------------------------------------------------------------------------------------------------
  int dslot = 0, mslot = 0, mslot_d = D, mslot_2d = (2 * D) % MS;
  int st_hist[D];   // store instructions issued in each of the last D steps (st_hist[0] = the previous step)
#pragma unroll
  for (int k = 0; k < D; ++k) st_hist[k] = 0;
  for (int t = 0; t < nst; ++t) {
    // rows of stage t and metadata of stage t + D were issued D steps ago; younger: (D - 1) steps of loads and the stores of
    // the last D steps
    int young = OPS * (D - 1);
#pragma unroll
    for (int k = 0; k < D; ++k) young += st_hist[k];
    wait_vm(young);
    stamp(1);

    const unsigned mt = meta0 + mslot * T::META_BYTES;
    const unsigned md = meta0 + mslot_d * T::META_BYTES;
    int hr[NDMA];
    if constexpr (EB == 4) {
      // ---- exact fp32: the stage's adjacency words (two per TC block), row ids of stage t + D, 8 x SLOTS floats of B ---------
      unsigned aw[4][2];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        aw[b][0] = lds_read_b32(mt + aword_base + 16 * b);
        aw[b][1] = lds_read_b32(mt + aword_base + 16 * b + 8);
      }
      read_rows(md, hr);
      float bv[8][SLOTS];
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const unsigned off = (unsigned)((NDMA == 8 ? 16 * (m & 1) + (m >> 1) : m) * ROW_BYTES) + (unsigned)dslot * STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) bv[m][s] = lds_read_f32(f32_base[s] + off);
      }
      wait_lgkmcnt0();
      stamp(2);
      fix_rows(cr, hr);
      float av[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        unsigned word = aw[m >> 1][m & 1];
        if (cc.blk + (m >> 1) >= cc.end) word = 0u;   // TC blocks past the window's end contribute zero
        av[m] = (float)((word >> a_bit) & 1u);
      }
      const bool any_edge = cc.ncl != 0;
      issue_meta(cm, mslot_2d);
      issue_data(dslot, hr);
      stamp(3);
      // operands swapped as in the 16-bit path: D'[column 4 g + j of the slot][row R]
      if (fresh) {   // wave-uniform
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
          acc[s] = any_edge ? __builtin_amdgcn_mfma_f32_16x16x4f32(bv[0][s], av[0], float4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0)
                            : float4_t{0.f, 0.f, 0.f, 0.f};
      } else if (any_edge) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[0][s], av[0], acc[s], 0, 0, 0);
      }
      if (any_edge) {
#pragma unroll
        for (int m = 1; m < 8; ++m)
#pragma unroll
          for (int s = 0; s < SLOTS; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[m][s], av[m], acc[s], 0, 0, 0);
      }
    } else {
    // A words of stage t (requested first), row ids of stage t + D, B fragments of stage t: one LDS round trip
    const unsigned wlo = lds_read_b32(mt + 128 + 4 * (4 * g + (R >> 3)));
    const unsigned whi = lds_read_b32(mt + 128 + 4 * (4 * g + 2 + (R >> 3)));
    read_rows(md, hr);

    unsigned taddr[SLOTS];
    taddr[0] = tr_base + dslot * STAGE_BYTES;
#pragma unroll
    for (int b = 0; (1 << b) < SLOTS; ++b)
#pragma unroll
      for (int s = (1 << b); s < (2 << b) && s < SLOTS; ++s) taddr[s] = taddr[s - (1 << b)] + tr_delta[b];
    uint2_t blo[SLOTS], bhi[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      if (VOLTRIX_STREAM_DIAG & 1) {
        blo[s] = bhi[s] = uint2_t{wlo, whi};
      } else {
        blo[s] = lds_read_tr16_b64<0>(taddr[s]);
        bhi[s] = lds_read_tr16_b64<TR_SECOND>(taddr[s]);
      }
    }
    wait_lgkmcnt0();
    stamp(2);
    fix_rows(cr, hr);
    unsigned nl = (wlo >> a_shift) & 0xFu, nh = (whi >> a_shift) & 0xFu;
    if (cc.blk + g >= cc.end) nl = nh = 0u;   // TC blocks past the window's end contribute zero
    const half8_t afrag = nibbles_to_ones_x2<T::BF16>(nl, nh);
    const bool any_edge = cc.ncl != 0;        // a window without edges owns one all-zero block: no MFMA (0 x NaN in B[0])
    // refill: metadata of stage t + 2 D, rows of stage t + D into the ring slot just read
    issue_meta(cm, mslot_2d);
    issue_data(dslot, hr);
    stamp(3);
    // operands swapped: D'[column 4 g + j of the slot][row R] -- four consecutive columns of one output row per lane.  A unit's
    // first stage multiplies onto the constant 0 (the C operand of the MFMA): the accumulators are never cleared by hand
    auto mfma = [&](const int s, const float4_t c) -> float4_t {
      const uint4_t bq = {blo[s][0], blo[s][1], bhi[s][0], bhi[s][1]};
      if constexpr (T::BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bq), __builtin_bit_cast(bf16x8_t, afrag),
                                                       c, 0, 0, 0);
      else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, bq), afrag, c, 0, 0, 0);
    };
    if (VOLTRIX_STREAM_DIAG & 1) {
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) acc[s] = float4_t{(float)blo[s][0], (float)bhi[s][1], (float)blo[s][1], (float)bhi[s][0]};
    } else if (fresh) {   // wave-uniform
      if (any_edge) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) acc[s] = mfma(s, float4_t{0.f, 0.f, 0.f, 0.f});
      } else {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) acc[s] = float4_t{0.f, 0.f, 0.f, 0.f};
      }
    } else if (any_edge) {
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) acc[s] = mfma(s, acc[s]);
    }
    }
    fresh = false;
    stamp(4);
    int stored = 0;
    if (cc.blk + cc.step >= cc.end) {   // the unit's last stage (wave-uniform): store it; the next unit starts from 0
      if (VOLTRIX_STREAM_DIAG & 2) {
        // keeps EVERY accumulator component live: with bit 0 they are copies of LDS-read outputs, and an asynchronous read
        // whose output is dead may land in a register the compiler has handed out again (the inline-asm note of
        // spmm_panel_kernels.hpp) -- bits 0 | 1 together gathered from garbage row ids that way
        float live = 0.f;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) live += (acc[s][0] + acc[s][1]) + (acc[s][2] + acc[s][3]);
        if (live == 12345.678f) a.output[0] = live;
      } else {
        store_unit(cc.u);
        stored = ns_live;
      }
      fresh = true;
    }
    stamp(5);
#pragma unroll
    for (int k = D - 1; k > 0; --k) st_hist[k] = st_hist[k - 1];
    st_hist[0] = stored;
    cur_advance(cc);
    cur_advance(cr);
    cur_advance(cm);
    dslot = dslot + 1 == D ? 0 : dslot + 1;
    mslot = mslot + 1 == MS ? 0 : mslot + 1;
    mslot_d = mslot_d + 1 == MS ? 0 : mslot_d + 1;
    mslot_2d = mslot_2d + 1 == MS ? 0 : mslot_2d + 1;
    stamp(6);
  }
  wait_vmcnt<0>();   // trailing DMAs must have landed before the wave's LDS is released
  if ((VOLTRIX_STREAM_DIAG & 16) && lane == 0) {
    float* const dbg = a.partials + 8ll * ((long long)blockIdx.x * T::WAVES + wave);
#pragma unroll
    for (int i = 0; i < 7; ++i) dbg[i] = (float)ph[i];
    dbg[7] = (float)(unsigned)(__builtin_amdgcn_s_memtime() - tk0);
  }
}

// Host launcher.  slab handling as launch_spmm_tc16 (one launch per 256-byte group of column slabs when B fits the
// Infinity Cache, slab-major order for slabs of whole 128-byte lines).
template <class T>
inline int launch_spmm_stream(const uint32_t* hspa_packed, const int* hind, int num_nodes, int embedding_dim,
                              const void* input, float* output, hipStream_t stream, const int* units /* int32[U][8] */,
                              const int* runs /* int32[R][4] */, const int* run_ptr /* int32[9] */, int max_runs_per_xcd,
                              float* partials, const float* out_scale = nullptr, int slab_first = 0, int slab_count = 0,
                              long long input_rows = 0, int slab_policy = kSlabAuto) {
  if constexpr (T::WEIGHTED || T::FS > 128 || (T::EB == 4 && (T::FS > 64 || T::BF16))) {
    return kErrBadConfig;   // binary A; fp16 / bf16 slabs up to 128 columns, fp32 slabs up to 64 (the tuner never asks for more)
  } else {
  if (num_nodes < 0 || embedding_dim < 0 || max_runs_per_xcd < 0) return kErrBadShape;
  if (num_nodes == 0 || embedding_dim == 0 || max_runs_per_xcd == 0) return kOk;
  if (embedding_dim % (16 / T::EB) != 0) return kErrBadShape;
  if (((uintptr_t)input & 15) || ((uintptr_t)hspa_packed & 15) || ((uintptr_t)units & 15) || ((uintptr_t)runs & 15) ||
      ((uintptr_t)output & 15) || ((uintptr_t)partials & 15) || run_ptr == nullptr)
    return kErrBadShape;
  const int total_slabs = (embedding_dim + T::FS - 1) / T::FS;
  if (slab_first < 0 || slab_count < 0 || slab_first + slab_count > total_slabs) return kErrBadShape;
  if (const int group = slab_count == 0 ? slab_launch_group(total_slabs, T::FS * T::EB,
                                                            input_rows > 0 ? input_rows : (long long)num_nodes, slab_policy)
                                        : 0) {
    for (int s = 0; s < total_slabs; s += group) {
      const int rc = launch_spmm_stream<T>(hspa_packed, hind, num_nodes, embedding_dim, input, output, stream, units, runs,
                                           run_ptr, max_runs_per_xcd, partials, out_scale, s,
                                           total_slabs - s < group ? total_slabs - s : group, input_rows);
      if (rc != kOk) return rc;
    }
    return kOk;
  }
  StreamArgs<T> a;
  a.hspa_packed = hspa_packed;
  a.hind = hind;
  a.input = static_cast<const typename StreamArgs<T>::in_t*>(input);
  a.output = output;
  a.partials = partials;
  a.units = reinterpret_cast<const int4*>(units);
  a.runs = reinterpret_cast<const int4*>(runs);
  a.run_ptr = run_ptr;
  a.out_scale = out_scale;
  a.num_nodes = num_nodes;
  a.F = embedding_dim;
  a.slab_first = slab_count > 0 ? slab_first : 0;
  a.num_slabs = slab_count > 0 ? slab_count : total_slabs;
  a.slab_major = slab_major_order(a.num_slabs, T::FS * T::EB);
  a.meta_nt = total_slabs == 1;
  a.max_runs_per_xcd = max_runs_per_xcd;
  const long long blocks_per_xcd = ((long long)max_runs_per_xcd * a.num_slabs + T::WAVES - 1) / T::WAVES;
  const long long grid = blocks_per_xcd * kNumXcd;
  if (grid > 0x7FFFFFFFll) return kErrBadShape;
  const long long b_rows = input_rows > 0 ? input_rows : (long long)num_nodes;
  const bool small_b = b_rows < (1ll << 24) && b_rows * embedding_dim * T::EB < (1ll << 32) - 8192 && embedding_dim * T::EB < (1 << 24);
  if (small_b) {
    const int lds_rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&spmm_stream_kernel<T, false>), T::BLOCK_LDS);
    if (lds_rc != kOk) return lds_rc;
    hipLaunchKernelGGL((spmm_stream_kernel<T, false>), dim3((unsigned)grid), dim3(T::THREADS), T::BLOCK_LDS, stream, a);
  } else {
    const int lds_rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&spmm_stream_kernel<T, true>), T::BLOCK_LDS);
    if (lds_rc != kOk) return lds_rc;
    hipLaunchKernelGGL((spmm_stream_kernel<T, true>), dim3((unsigned)grid), dim3(T::THREADS), T::BLOCK_LDS, stream, a);
  }
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
  }
}

}  // namespace voltrix
