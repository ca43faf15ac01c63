// fp32 tile GEMM on LDS-DMA stages (v_mfma_f32_16x16x4_f32): the EmbraceNet backward on PRE-MASKED gradients, the backward of large
// Linear layers, and the convolutions of the stored-activation blocks with >= 128 output channels.
//
//   dX_m[B,d_m] = dD_m   W_m      "dgrad"  A = dD_m row-major [M = B][K = c],  Bm = W_m K-major [c][d_m]
//   dW_m[c,d_m] = dD_m^T X_m      "wgrad"  A = dD_m K-major  [K = B][M = c],   Bm = X_m K-major [B][d_m]   (+ db_m = sum_b dD_m)
// with dD_m = dE * [idx == m] * [pre_m > 0] written ONCE by the kernel that produces dE (the classifier head, head.hip, or
// emb_embrace_premask) instead of being re-derived from dE and the code bytes by every wave for every fragment: the four jobs are
// plain GEMMs.  fp32 is the precision the 1e-5 parity bar is stated in and BASELINE configs 3 and 4 run in; its matrix rate is
// 1/16 of bf16's, so this path is bound by the MATRIX PIPE once the operands arrive on time.
// Replaces autograd through EmbraceNetMultimodal.py:52-60,80-88 (utils/training_models_multimodal.py:156, :115 "model.double()":
// the reference trains in fp64; fp32 is the engine's exact-parity precision).
//
// Structure.  One workgroup per 128 x 128 output tile, tiles in list order = longest first (the hardware dispatcher is the load
// balancer).  A group of eight waves (4 x 2, each 32 rows x 64 columns) streams its operands as 32 KB stages (32 reduction
// indices: m-side image 16 KB + n-side image 16 KB) through a DOUBLE BUFFER in LDS, each stage requested by LDS-DMA (issued from
// inline assembly, see gj_dma) while the previous one is multiplied: one raw s_barrier per stage makes the stage visible and
// frees the other slot.  Accumulators leave straight from registers (buffer stores, 16 bytes per lane; the range check drops rows
// past the matrix).
// What feeds the matrix pipe is FOUR waves per SIMD (measured, tools/kbench: two waves per SIMD leave it idle ~30 % of the loop
// whatever their schedule -- a five-slot ring requested four stages ahead, the next stage's fragments read ahead, the two halves of
// the workgroup running half a stage apart: every variant stayed at ~6300 cycles per stage against 4480 of matrix work):
//   - launches with many tiles: 512-thread workgroups, 64 KB of LDS, TWO per CU;
//   - launches with few tiles (gj_launch): 1024-thread workgroups whose two groups of eight waves multiply the two HALVES of the
//     tile's reduction, each through its own double buffer; the second group's accumulators cross to the first through LDS (fixed
//     order: first half + second half) before the epilogue.  This also halves the critical path of a long tile.
// Images.  Row-major operand (dgrad m side): 128 rows x 128 B (32 k).  K-major operands: four QUARTER images of 32 k rows x
// 128 B (32 columns) each.  All rows are 128 bytes with their eight 16-byte slots XOR-permuted by swz16(row) (split_core.h):
// the permutation is applied to the per-lane source address of the (lane-linear) LDS-DMA and again in the fragment reads.
// MFMA step (h, j), h = 0..1, j = 0..3: lane group g supplies k = 16 h + 4 kq(g) + j on both operands -- the row-major operand
// with ONE ds_read_b128 per (tile, h) (its four floats are the four j), the K-major operands with one ds_read_b32 per step.
#pragma once
#include "conv_tiles.h"
#include "embrace_bwd_split.h"

namespace emb {

constexpr int kGjThreads = 512;
constexpr int kGjKC = 32;                         // reduction indices per stage
constexpr int kGjStage = 32 * 1024;               // m-side image (16 KB) + n-side image (16 KB)
constexpr int kGjDma = 4;                         // LDS-DMA instructions per wave and stage
constexpr int kGjSlots = 2;                       // double buffer: the stage being multiplied and the one in flight
constexpr int kGjLds = kGjSlots * kGjStage;       // 64 KB per group of eight waves

struct GJob {
  const char* A;        // m-side operand (pre-masked gradient)
  const char* Bm;       // n-side operand, K-major [K][N]
  char* C;              // dgrad: dX [M][ldc];  wgrad: dW [M][ldc], or the slabs [S][M][ldc] when S > 1   (all f32)
  float* bias;          // wgrad, S == 1: db [M]
  long slice_stride;    // wgrad, S > 1: bytes between the slabs of two slices
  int M, N, K;          // C is M x N, reduction length K    (dgrad: B, d, c;  wgrad: c, d, B)
  int lda, ldb, ldc;    // row pitches in elements
  int tiles_m, tiles_n, tiles;   // 128-wide tiles; tiles = tiles_m * tiles_n
  int S, kper;          // wgrad: the reduction is cut into S slices of kper rows
  int first, count;     // this job's range of the launch's tile list; count = tiles * S
  int kind;             // 0 dgrad (A row-major, Bm K-major), 1 wgrad (both K-major), 2 convolution forward / input gradient (A = activation
                        // rows shifted by the tap, Bm = packed weights ROW-major [N][K]), 3 convolution weight gradient (A = dy K-major,
                        // Bm = activation rows shifted per 32-column quarter)
  // convolutions (kinds 2, 3; Conv1d of CNN_pre.py:37-38 on channels-last rows r = b * L + l): sequence length, channels of the
  // shifted activation operand, left padding, rows B * L
  int L, cin, pad, R;
  int epi;              // epilogue: 0 store, 1 convolution forward (+ bias, per-tile BatchNorm partial sums), 2 slab + bias column (kind 3)
  const float* bias_in; // epi 1: bias [N]
  float* partial;       // epi 1: [tiles_m][2][N] sums of y and y^2 over the tile's rows
  int m_fast;           // tile order inside a slice: 1 = m tile fastest (tiles sharing an n-side panel are neighbours), 0 = n tile fastest
  uint32_t magic_tiles, magic_inner;   // ceil(2^32 / tiles), ceil(2^32 / (m_fast ? tiles_m : tiles_n))
};

struct GArgs {
  GJob j[4];
  int total;          // tiles of the launch
};

// one tile of the list, decoded (all wave-uniform)
struct GTile {
  int valid, kind;
  int m0, n0, k_begin, k_end, nstages;
  int st0;              // first stage of the reduction this group of waves multiplies (kind 2: stage -> (tap, channel block))
  int M, N, lda, ldb, ldc, S;
  const char *A, *Bm;
  char* C;
  float* bias;
  int L, cin, pad, R, epi, tm;
  const float* bias_in;
  float* partial;
};

// The job table is read through the kernel-argument segment POINTER with a run-time job index (scalar loads with a register
// offset).  (Indexing the by-value argument itself is what hipcc 7.2 turns into scratch copies -- embrace_bwd.hip; a select chain
// over four jobs costs ~100 scalar registers per decode.)
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(4))) const GArgs* gj_args_ptr;
__device__ __forceinline__ GTile gj_decode(int id, gj_args_ptr ka) {
  GTile t;
  // two rounds of scalar loads: the three range starts, then the job's whole descriptor in wide loads
  const int f1 = ka->j[1].first, f2 = ka->j[2].first, f3 = ka->j[3].first;
  t.valid = 1;
  const int k = (id >= f1) + (id >= f2) + (id >= f3);
  const GJob j = ka->j[k];
  const int q = id - j.first;
  const int slice = j.S > 1 ? div_magic(q, j.magic_tiles) : 0;
  const int t2 = q - slice * j.tiles;
  const int inner = j.m_fast ? j.tiles_m : j.tiles_n;
  const int hi = div_magic(t2, j.magic_inner), lo = t2 - hi * inner;
  const int tm = j.m_fast ? lo : hi, tn = j.m_fast ? hi : lo;
  t.kind = j.kind;
  t.m0 = tm * 128;
  t.n0 = tn * 128;
  t.k_begin = slice * j.kper;
  t.k_end = min(j.K, t.k_begin + j.kper);
  t.nstages = (t.k_end - t.k_begin + kGjKC - 1) / kGjKC;
  t.st0 = 0;
  t.M = j.M; t.N = j.N; t.lda = j.lda; t.ldb = j.ldb; t.ldc = j.ldc; t.S = j.S;
  t.A = j.A; t.Bm = j.Bm;
  t.C = j.C + (long)slice * j.slice_stride;
  t.bias = j.bias;
  t.L = j.L; t.cin = j.cin; t.pad = j.pad; t.R = j.R; t.epi = j.epi; t.tm = tm;
  t.bias_in = j.bias_in; t.partial = j.partial;
  return t;
}

// The request side.  Per tile: two buffer resources whose bases are the tile's first stage (their range checks
// zero-fill everything past the operands: rows >= M, reduction indices >= k_end, columns >= N via invalid lane offsets), and
// this wave's four per-lane byte offsets; per stage: four LDS-DMA instructions and four vector adds (offsets of lanes that
// must read zeros start at 2^31 and stay out of range).
// LDS-DMA is issued from inline assembly: the compiler orders every LDS read behind ALL outstanding `buffer_load ... lds` it
// knows of (s_waitcnt vmcnt(0)), which would serialise the stages in flight (measured: 2200 cycles per stage instead of ~600);
// every consumer of a stage waits explicitly (s_waitcnt vmcnt(0) + s_barrier).
typedef int gj_rsrc __attribute__((ext_vector_type(4)));
__device__ __forceinline__ gj_rsrc gj_make_rsrc(const void* origin, long bytes) {
  const uint64_t p = (uint64_t)(uintptr_t)origin;
  gj_rsrc rs;
  rs[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)p);
  rs[1] = __builtin_amdgcn_readfirstlane((int)((p >> 32) & 0xffffu));
  rs[2] = __builtin_amdgcn_readfirstlane((int)(bytes < 0 ? 0 : (bytes < 0x7fffffffL ? bytes : 0x7fffffffL)));
  rs[3] = 0x00020000;
  return rs;
}
__device__ __forceinline__ void gj_dma(const gj_rsrc& rs_in, uint32_t lds, uint32_t voff) {
  const int m = __builtin_amdgcn_readfirstlane((int)lds);
  gj_rsrc rs;                                     // (wave-uniform by construction; the constraint needs to SEE scalar values)
#pragma unroll
  for (int i = 0; i < 4; ++i) rs[i] = __builtin_amdgcn_readfirstlane(rs_in[i]);
#ifndef GJ_DIAG_NO_DMA      // (diagnostic builds of tools/kbench only)
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(m), "v"(voff), "s"(rs) : "memory");
#endif
}

struct GFeed {
  // convolution kinds: the tap and channel block of the stage being requested, channel blocks per tap, and per lane the position
  // inside its sequence of this lane's rows (a shifted row outside [0, L) is the zero padding)
  int tap, cb, cpt, L, pad, cin4;      // cin4 = bytes per activation row
  int lpos[2];                          // kind 2: rows 8 wave + rl and + 64;  kind 3: [0] = the k row of this lane's instructions
  uint32_t cbase[2];                    // kind 2: byte offset of the unshifted row;  kind 3: column part of quarter i (or invalid)
  int qtap[2];                          // kind 3: tap of this wave's two quarters
  int krow0;                            // kind 3: first activation row of the tile's slice + this lane's row in the instruction
  int valid, kind, stage, nstages, slot;
  int tail_rem;                    // dgrad: valid bytes of the m-side rows in the LAST stage (128 = whole window)
  gj_rsrc ra, rb;
  uint32_t off[4];                 // per-lane source offsets of this wave's four instructions: A, A, B, B
  uint32_t dst[4];                 // LDS byte offsets of their 1 KiB destinations inside a slot
  uint32_t step_a, step_b;         // bytes per stage
  int slot16;                      // dgrad: byte offset of this lane's slot inside the 128-byte A window (tail check)
};

__device__ __forceinline__ void gj_feed_tile(GFeed& f, const GTile& t, int lane, int wave) {
  f.valid = t.valid;
  if (!t.valid) return;
  f.kind = t.kind;
  f.stage = 0;
  f.nstages = t.nstages;
  const int klen = t.k_end - t.k_begin;
  const int rl = lane >> 3;                                       // row of this lane inside an instruction's eight rows
  // K-major quarter images (32 k rows x 128 B): this wave issues rows 8 (wave & 3) .. + 8 of quarters wave >> 2 and (wave >> 2) + 2
  const uint32_t krow = (uint32_t)(8 * (wave & 3) + rl);
  const int kslot = 16 * ((lane & 7) ^ swz16((int)krow));
  const long b0 = (long)t.k_begin * t.ldb * 4;
  f.rb = gj_make_rsrc(t.Bm + b0, (long)klen * t.ldb * 4);
  f.step_b = (uint32_t)(kGjKC * t.ldb * 4);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = (wave >> 2) + 2 * i;
    const int cb = t.n0 * 4 + 128 * q + kslot;
    f.off[2 + i] = cb < t.N * 4 ? krow * (uint32_t)(t.ldb * 4) + (uint32_t)cb : kDmaInvalid;
    f.dst[2 + i] = (uint32_t)(16384 + q * 4096 + (wave & 3) * 1024);
  }
  if (t.kind == 2) {
    // forward / input gradient of a convolution: y[r][n] = sum_{tap, ci} x[r + tap - pad][ci] w[n][tap * cin + ci]; stage s = (tap,
    // channel block of 32); A image = rows m0 .. m0 + 127 of x SHIFTED by tap - pad (zero outside the sequence), Bm image = 128
    // rows of the packed weights [N][K], 32 reduction indices each
    const uint32_t row = (uint32_t)(8 * wave + rl);
    const int slot = 16 * ((lane & 7) ^ swz16((int)row));
    f.ra = gj_make_rsrc(t.A, (long)t.R * t.cin * 4);
    f.rb = gj_make_rsrc(t.Bm + (long)t.n0 * t.ldb * 4, (long)(t.N - t.n0) * t.ldb * 4);
    f.cpt = t.cin / 32; f.L = t.L; f.pad = t.pad; f.cin4 = t.cin * 4;
    f.tap = t.st0 / f.cpt; f.cb = t.st0 - f.tap * f.cpt;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = t.m0 + (int)row + 64 * i;
      const int b = r / t.L;
      f.lpos[i] = r < t.R ? r - b * t.L : -(1 << 24);
      f.cbase[i] = (uint32_t)r * (uint32_t)(t.cin * 4) + (uint32_t)slot;
      f.off[2 + i] = (row + 64u * i) * (uint32_t)(t.ldb * 4) + (uint32_t)slot + (uint32_t)t.st0 * 128u;
      f.dst[i] = (uint32_t)(i * 8192 + wave * 1024);
      f.dst[2 + i] = (uint32_t)(16384 + i * 8192 + wave * 1024);
    }
    f.step_a = 0; f.step_b = 128u;
    f.slot16 = 0; f.tail_rem = 128;
    return;
  }
  if (t.kind == 3) {
    // weight gradient of a convolution: dW[co][tap * cin + ci] = sum_r dy[r][co] x[r + tap - pad][ci]; A = dy K-major quarters
    // as in kind 1; every 32-column quarter of the n side is one (tap, channel block): its rows are x shifted by tap - pad
    const long a0 = (long)t.k_begin * t.lda * 4;
    f.ra = gj_make_rsrc(t.A + a0, (long)klen * t.lda * 4);
    f.step_a = (uint32_t)(kGjKC * t.lda * 4);
    f.rb = gj_make_rsrc(t.Bm, (long)t.R * t.cin * 4);
    f.step_b = 0;
    f.L = t.L; f.pad = t.pad; f.cin4 = t.cin * 4;
    f.krow0 = t.k_begin + (int)krow;
    f.lpos[0] = f.krow0 % t.L;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = (wave >> 2) + 2 * i;
      const int cbm = t.m0 * 4 + 128 * q + kslot;
      f.off[i] = cbm < t.M * 4 ? krow * (uint32_t)(t.lda * 4) + (uint32_t)cbm : kDmaInvalid;
      f.dst[i] = (uint32_t)(q * 4096 + (wave & 3) * 1024);
      const int col = t.n0 + 32 * q;                         // first column of the quarter: (tap, channel block)
      const int tq = col / t.cin;
      f.qtap[i] = tq;
      f.cbase[i] = col < t.N ? (uint32_t)((col - tq * t.cin) * 4 + kslot) : kDmaInvalid;
      f.dst[2 + i] = (uint32_t)(16384 + q * 4096 + (wave & 3) * 1024);
      f.off[2 + i] = 0;
    }
    f.slot16 = 0; f.tail_rem = 128;
    return;
  }
  if (t.kind == 0) {                                              // A row-major: 128 m rows x 128 B of k; instructions wave, wave + 8
    const uint32_t row = (uint32_t)(8 * wave + rl);
    const int slot = 16 * ((lane & 7) ^ swz16((int)row));         // (swz16 depends on row bits 1 and 3 only: the same for row + 64)
    const long a0 = ((long)t.m0 * t.lda + t.k_begin) * 4;
    f.ra = gj_make_rsrc(t.A + a0, ((long)(t.M - t.m0) * t.lda - t.k_begin) * 4);
    f.step_a = 128u;
    f.off[0] = row * (uint32_t)(t.lda * 4) + (uint32_t)slot;
    f.off[1] = (row + 64u) * (uint32_t)(t.lda * 4) + (uint32_t)slot;
    f.dst[0] = (uint32_t)(wave * 1024);
    f.dst[1] = (uint32_t)(8192 + wave * 1024);
    f.slot16 = slot;
    f.tail_rem = (klen - (t.nstages - 1) * kGjKC) * 4;
  } else {                                                        // A K-major quarters (32 m columns each)
    const long a0 = (long)t.k_begin * t.lda * 4;
    f.ra = gj_make_rsrc(t.A + a0, (long)klen * t.lda * 4);
    f.step_a = (uint32_t)(kGjKC * t.lda * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = (wave >> 2) + 2 * i;
      const int cb = t.m0 * 4 + 128 * q + kslot;
      f.off[i] = cb < t.M * 4 ? krow * (uint32_t)(t.lda * 4) + (uint32_t)cb : kDmaInvalid;
      f.dst[i] = (uint32_t)(q * 4096 + (wave & 3) * 1024);
    }
    f.slot16 = 0;
    f.tail_rem = 128;
  }
}

// Requesting a stage = the four LDS-DMA instructions gj_part<0..3> (gj_more: is there a stage left to request) -- issued BETWEEN
// the MFMA groups of the stage being multiplied: an LDS-DMA instruction holds the issuing wave for ~150 cycles (the texture path
// takes 64 B per clock, 32 KB per stage = 512 cycles per CU), which hides under the other waves' MFMAs only when the eight
// waves do not all issue at once right after the barrier.
__device__ __forceinline__ bool gj_more(const GFeed& f) { return f.stage < f.nstages; }
template <int I> __device__ __forceinline__ void gj_part(GFeed& f, uint32_t lds0) {
  constexpr int IA = I < 2 ? I : 0, IB = I >= 2 ? I - 2 : 0;   // (in-range indices for the branches I does not take)
  const uint32_t buf = lds0 + (uint32_t)(f.slot * kGjStage) + f.dst[I];
  if (f.kind == 2 && I < 2) {                      // activation rows shifted by the stage's tap
    const int sh = f.tap - f.pad;
    const bool ok = (unsigned)(f.lpos[IA] + sh) < (unsigned)f.L;
    gj_dma(f.ra, buf, ok ? f.cbase[IA] + (uint32_t)(sh * f.cin4 + f.cb * 128) : kDmaInvalid);
    if (I == 1) { if (++f.cb == f.cpt) { f.cb = 0; ++f.tap; } }
    return;
  }
  if (f.kind == 3 && I >= 2) {                     // activation rows shifted by the quarter's tap
    const int sh = f.qtap[IB] - f.pad;
    const bool ok = f.cbase[IB] != kDmaInvalid && (unsigned)(f.lpos[0] + sh) < (unsigned)f.L;
    const int r = f.krow0 + f.stage * kGjKC + sh;
    gj_dma(f.rb, buf, ok ? (uint32_t)r * (uint32_t)f.cin4 + f.cbase[IB] : kDmaInvalid);
    if (I == 3) {
      int l = f.lpos[0] + kGjKC;                   // position of the next stage's row inside its sequence (L >= 16)
      if (l >= f.L) l -= f.L;
      if (l >= f.L) l -= f.L;
      f.lpos[0] = l;
      f.slot ^= 1;
      ++f.stage;
    }
    return;
  }
  if (I < 2) {
    uint32_t o = f.off[I];
    if (f.kind == 0 && f.stage + 1 == f.nstages && f.tail_rem < 128 && f.slot16 >= f.tail_rem) o = kDmaInvalid;   // k beyond K reads zeros
    gj_dma(f.ra, buf, o);
    f.off[I] += f.step_a;
  } else {
    gj_dma(f.rb, buf, f.off[I]);
    f.off[I] += f.step_b;
  }
  if (I == 3) {
    f.slot ^= 1;
    ++f.stage;
  }
}
__device__ __forceinline__ bool gj_request(GFeed& f, uint32_t lds0) {
  if (!gj_more(f)) return false;
  gj_part<0>(f, lds0); gj_part<1>(f, lds0); gj_part<2>(f, lds0); gj_part<3>(f, lds0);
  return true;
}
#endif

struct GHalf {                                    // fragments of one half of a stage (16 reduction indices = 4 MFMA steps)
  float a[4][4];                                  // n side: [column tile][step j]
  float b[2][4];                                  // m side: [row tile][step j]
};

// PAIR: sixteen waves; waves 8-15 are a second group that multiplies the SECOND HALF of the tile's reduction on the same 128 x 128
// tile with its own two-slot ring (the two-slot schedule, 4 waves per SIMD inside ONE workgroup: the matrix pipe stays fed whatever
// the number of tiles of the launch, and the longest tile's critical path halves); the two accumulator sets meet through LDS in
// fixed order (first half + second half) before the epilogue, which the first group runs.
template <bool PAIR>
__global__ __launch_bounds__(PAIR ? 2 * kGjThreads : kGjThreads, 4) void gemm_jobs_kernel(const GArgs args) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const gj_args_ptr ka = (gj_args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  const int lane = threadIdx.x & 63, wave16 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wave = wave16 & 7, sub = wave16 >> 3;    // group of eight waves (PAIR: 0 / 1)
  const int wr = wave >> 1, wc = wave & 1;           // this wave: m rows 32 wr .. +32, n columns 64 wc .. +64
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem + (uint32_t)(sub * kGjLds);
  EMB_STAMP(2);
  // ONE tile per workgroup, in list order = longest first: the hardware's workgroup dispatcher is the load balancer (a tile
  // of the input gradient multiplies c / 32 stages, a weight-gradient slice kper / 32; dealt out statically the workgroups that
  // drew two long tiles ran alone at the end -- measured 142 us against 99 us at the cfg3 shapes)
  GTile t = gj_decode(blockIdx.x, ka);
  int iters = t.nstages;                             // barrier intervals of the main loop (the same for all waves of the workgroup)
  if constexpr (PAIR) {
    const int n0s = (t.nstages + 1) >> 1;            // stages of the first group (the longer half)
    iters = n0s;
    if (sub == 0) {
      t.k_end = min(t.k_end, t.k_begin + n0s * kGjKC);
      t.nstages = n0s;
    } else {
      t.k_begin += n0s * kGjKC;
      t.st0 = n0s;
      t.nstages -= n0s;
    }
  }
  GFeed feed;
  feed.slot = 0;
  gj_feed_tile(feed, t, lane, wave);
  (void)gj_request(feed, lds0);                      // stage 0
  EMB_STAMP(3);

  // per-lane parts of the fragment reads (tile-independent).  MFMA step (h, j): lane group g supplies k = 16 h + 4 kq(g) + j with
  // kq = {0, 2, 1, 3}: the two lane groups of a 32-lane half then read k rows EIGHT apart, whose slot permutations differ in the
  // half-row bit -- the ds_read_b32 of the K-major images are bank-conflict free (with kq(g) = g the groups' rows were four apart,
  // same permutation, same banks: every read two-way conflicted and the loads, not the MFMAs, paced the stage).
  // K-major quarter image, column 16 u + r of the quarter (u = 0 / 1): byte (16 h + j) * 128 + 4 kq * 128
  // + 16 * ((4 u + (r >> 2)) ^ swz16(16 h + 4 kq + j)) + 4 (r & 3), swz16(..) = ((j >> 1) & 1) << 1 | (kq >> 1) << 2 = .. | (g & 1) << 2:
  // a lane part and an XOR of byte-offset bit 5 for j >= 2
  const int r = lane & 15, g = lane >> 4;
  const int kq = 2 * (g & 1) + (g >> 1);
  uint32_t kmo[2][2];                                // [u][j >> 1]
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const uint32_t base = (uint32_t)(4 * kq * 128 + 16 * ((4 * u + (r >> 2)) ^ ((g & 1) << 2)) + 4 * (r & 3));
    kmo[u][0] = base;
    kmo[u][1] = base ^ 32u;
  }
  // row-major image: the four floats of slot 4 h + kq of row r are this lane's k = 16 h + 4 kq + (0..3)
  uint32_t rmo[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) rmo[h] = (uint32_t)(r * 128 + (((4 * h + kq) ^ swz16(r)) << 4));
  const uint32_t n_img = (uint32_t)(16384 + wc * 8192);           // this wave's two n-side quarters
  const uint32_t am_img = (uint32_t)(wr * 4096);                  // wgrad: this wave's m-side quarter
  const uint32_t ar_img = (uint32_t)(wr * 2 * 2048);              // dgrad: this wave's 32 m rows

  typedef __attribute__((address_space(3))) const float lds_f32;
  // fragments of half h of the stage in `buf`.  The slot base is added to the four lane parts ONCE per call; everything else
  // of an address is a compile-time constant and travels in the instruction's offset field (one address add per read otherwise)
  auto load = [&](uint32_t buf, int kind, int h, GHalf& x) {
    if (kind == 2) {                                 // packed weights, row-major image of 128 n rows: one 16-byte read per tile
      const uint32_t nbr = rmo[h] + (buf + (uint32_t)(16384 + wc * 64 * 128));
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const f32x4 v = lds_read16<float>(nbr + ni * 2048);
#pragma unroll
        for (int j = 0; j < 4; ++j) x.a[ni][j] = v[j];
      }
    } else {
      uint32_t nb[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) nb[u][jj] = kmo[u][jj] + (buf + n_img);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          x.a[ni][j] = *(lds_f32*)(uintptr_t)(nb[ni & 1][j >> 1] + (uint32_t)((ni >> 1) * 4096 + (16 * h + j) * 128));
    }
    if (kind == 0 || kind == 2) {
      const uint32_t ab = rmo[h] + (buf + ar_img);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const f32x4 v = lds_read16<float>(ab + mi * 2048);
#pragma unroll
        for (int j = 0; j < 4; ++j) x.b[mi][j] = v[j];
      }
    } else {
      uint32_t mb[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) mb[u][jj] = kmo[u][jj] + (buf + am_img);
#pragma unroll
      for (int ci = 0; ci < 2; ++ci)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          x.b[ci][j] = *(lds_f32*)(uintptr_t)(mb[ci][j >> 1] + (uint32_t)((16 * h + j) * 128));
    }
  };

  int slot = 0;
  GHalf f0, f1;
  EMB_STAMP(4);
#ifdef GJ_DIAG_NO_LOAD
  load(lds0, t.kind, 0, f0);
  load(lds0, t.kind, 1, f1);
#endif
  {
    f32x4 acc[4][2];                                 // [n tile][m tile]
    float sb[2] = {0.0f, 0.0f};                      // wgrad, first n tile: this lane's share of the bias gradient
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[ni][mi][q] = 0.0f;
    const bool with_bias = (t.kind == 1 || t.kind == 3) && t.n0 == 0 && wc == 0;
    EMB_STAMP_KIND(t.kind);
#ifdef GJ_DIAG_NO_MMA
#define GJ_MMA(F, J0) asm volatile("" : : "v"(F.a[0][J0]), "v"(F.a[1][J0]), "v"(F.a[2][J0 + 1]), "v"(F.a[3][J0 + 1]), "v"(F.b[0][J0]), "v"(F.b[1][J0 + 1]))
#else
#define GJ_MMA(F, J0)                                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                                     \
      _Pragma("unroll") for (int j = J0; j < J0 + 2; ++j) {                                                                  \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) {                                                                   \
          _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                                   \
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(F.a[ni][j], F.b[mi][j], acc[ni][mi], 0, 0, 0);                \
          sb[mi] += F.b[mi][j];                                                                                              \
        }                                                                                                                    \
      }                                                                                                                      \
      __builtin_amdgcn_sched_barrier(0)
#endif
    {
      // two workgroups per CU, or two groups of eight waves in one (four waves per SIMD): the other workgroup's MFMAs fill this one's barrier, LDS-latency and
      // DMA-issue phases, so the schedule is the plain double buffer: the stage is visible after the barrier, which also frees
      // the other slot (every wave has multiplied the previous stage) for the next request; its first two DMA instructions are
      // issued under the first half's fragment reads, the other two (and the second half's reads) after the first MFMA group
      // (three quarters of a stage left to land)
#pragma unroll 1
      for (int s = 0; s < iters; ++s) {
        EMB_WAIT_VMCNT(0);
#ifndef GJ_DIAG_NO_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
        if (PAIR && s >= t.nstages) continue;        // (the second group of an odd number of stages sits the last interval out)
#ifdef GJ_DIAG_NO_DMA
        const bool feeding = false;
#else
        const bool feeding = gj_more(feed);
#endif
        const uint32_t buf = lds0 + (uint32_t)(slot * kGjStage);
        slot ^= 1;
#ifndef GJ_DIAG_NO_LOAD
        load(buf, t.kind, 0, f0);
#endif
        if (feeding) { gj_part<0>(feed, lds0); gj_part<1>(feed, lds0); }
        GJ_MMA(f0, 0);
#ifndef GJ_DIAG_NO_LOAD
        load(buf, t.kind, 1, f1);
#endif
        if (feeding) { gj_part<2>(feed, lds0); gj_part<3>(feed, lds0); }
        GJ_MMA(f0, 2);
        GJ_MMA(f1, 0);
        GJ_MMA(f1, 2);
      }
    }
#undef GJ_MMA
    EMB_STAMP(5);
    if constexpr (PAIR) {
      // the second group's accumulators (and bias-gradient shares) cross to the first through LDS: [wave][vector][lane] 16 bytes
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                 // both rings are idle
      f32x4* xch = reinterpret_cast<f32x4*>(smem);
      float* xsb = reinterpret_cast<float*>(smem + 8 * 8 * 64 * 16);
      if (sub == 1) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) xch[(wave * 8 + mi * 4 + ni) * 64 + lane] = acc[ni][mi];
          xsb[(wave * 2 + mi) * 64 + lane] = sb[mi];
        }
      }
      __syncthreads();
      if (sub == 0) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[ni][mi] += xch[(wave * 8 + mi * 4 + ni) * 64 + lane];
          sb[mi] += xsb[(wave * 2 + mi) * 64 + lane];
        }
      }
    }
    const bool storing = !PAIR || sub == 0;            // (the second group still takes part in the epilogue's barriers)
    // the tile leaves from registers.  Accumulator (ni, mi) of lane (r, g): row 32 wr + 16 mi + r, columns 64 wc + 16 ni + 4 g .. + 3.
    {
      const long rem = (((long)t.M - t.m0) * t.ldc - t.n0) * 4;                   // rows >= M are dropped by the range check
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc((void*)(t.C + ((long)t.m0 * t.ldc + t.n0) * 4), 0, dma_nrec(rem), 0x00020000);
      if (t.epi == 1) {
        // convolution forward: + bias, and this tile's BatchNorm partial sums (sum y, sum y^2 per channel over its valid rows):
        // per lane over its two row tiles, over the 16 rows of a lane group by DPP, over the four wave rows through LDS
        float s1[4][4], s2[4][4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int ncol = t.n0 + wc * 64 + ni * 16 + 4 * g;
          f32x4 bv = {0.f, 0.f, 0.f, 0.f};
          if (ncol < t.N) bv = *reinterpret_cast<const f32x4*>(t.bias_in + ncol);
#pragma unroll
          for (int q = 0; q < 4; ++q) { s1[ni][q] = 0.f; s2[ni][q] = 0.f; }
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            const bool rv = t.m0 + wr * 32 + mi * 16 + r < t.M;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float y = acc[ni][mi][q] + bv[q];
              acc[ni][mi][q] = y;
              s1[ni][q] += rv ? y : 0.f;
              s2[ni][q] += rv ? y * y : 0.f;
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) { s1[ni][q] = row16_sum<float>(s1[ni][q]); s2[ni][q] = row16_sum<float>(s2[ni][q]); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                             // every stage has been multiplied: the ring is free
        float* red = reinterpret_cast<float*>(smem);                              // [4 wave rows][2][128 channels]
        if (r == 0 && storing) {
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int ch = wc * 64 + ni * 16 + 4 * g + q;
              red[(wr * 2 + 0) * 128 + ch] = s1[ni][q];
              red[(wr * 2 + 1) * 128 + ch] = s2[ni][q];
            }
        }
        __syncthreads();
        if (threadIdx.x < 256) {
          const int which = threadIdx.x >> 7, ch = threadIdx.x & 127;
          if (t.n0 + ch < t.N)
            t.partial[((long)t.tm * 2 + which) * t.N + t.n0 + ch] =
                ((red[(0 * 2 + which) * 128 + ch] + red[(1 * 2 + which) * 128 + ch]) + red[(2 * 2 + which) * 128 + ch]) + red[(3 * 2 + which) * 128 + ch];
        }
      }
      if (storing) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const uint32_t rowoff = (uint32_t)(wr * 32 + mi * 16 + r) * (uint32_t)t.ldc;
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const int ncol = wc * 64 + ni * 16 + 4 * g;
            const bool inside = t.n0 + ncol < t.N;                                // N % 4 == 0: four columns are inside together
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[ni][mi]), rs,
                                                   inside ? (rowoff + (uint32_t)ncol) * 4u : kDmaInvalid, 0, 0);
          }
        }
      }
      if (with_bias && storing) {                                                            // db[m] = sum over the four lane groups' k shares
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          float sum = sb[mi];
          sum += __shfl_xor(sum, 16, 64);
          sum += __shfl_xor(sum, 32, 64);
          const int crow = t.m0 + wr * 32 + mi * 16 + r;
          if (g == 0 && crow < t.M) {
            if (t.S > 1 || t.epi == 2) reinterpret_cast<float*>(t.C)[(long)crow * t.ldc + t.N] = sum;
            else t.bias[crow] = sum;
          }
        }
      }
    }
  }
  EMB_STAMP(8);
#endif
}

// One tile per workgroup.  The matrix pipe needs FOUR waves per SIMD to stay fed through barriers, fragment reads and DMA issue:
// two waves of one workgroup leave it idle ~30 % of the main loop whatever their schedule (tools/kbench timing variants: every
// ingredient costs its full duration on top of the MFMAs; running the two halves of the workgroup half a stage apart changed
// nothing), two independent workgroups per CU reach ~4450 cycles per stage and CU against 4480 of matrix work.
//   many tiles (more than 1.25 per CU): the two-slot kernel, 64 KB of LDS, two workgroups per CU -- a workgroup's set-up and stores
//     also run under its neighbour's main loop;
//   fewer: the PAIRED kernel, sixteen waves on one tile, each group of eight multiplying one half of the reduction (128 KB of LDS).
static int gj_cus() {
  static const int cus = [] {
    int dev = 0, v = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
    return v > 0 ? v : 256;
  }();
  return cus;
}
static void gj_launch(const GArgs& ga, int n, hipStream_t s) {
  const int cus = gj_cus();
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_jobs_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kGjLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_jobs_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kGjLds);
    attr_set = true;
  }
#ifdef GJ_DIAG_FORCE_MODE
  const bool paired = GJ_DIAG_FORCE_MODE == 16;      // (tools/kbench: 2 = eight waves per tile, 16 = paired)
#else
  const bool paired = n <= cus + cus / 2;
#endif
  if (paired) gemm_jobs_kernel<true><<<n, 2 * kGjThreads, 2 * kGjLds, s>>>(ga);
  else gemm_jobs_kernel<false><<<n, kGjThreads, kGjLds, s>>>(ga);
}

// returns 1 when the shapes do not qualify (the caller keeps its other kernels)
static int gemm_jobs_bwd_impl(const void* dD0, const void* dD1, const void* X0, const void* X1, const void* W0, const void* W1,
                                  void* dX0, void* dX1, void* dW0, void* db0, void* dW1, void* db1, void* ws, int64_t ws_bytes, int B,
                                  int d0, int d1, int c, int force_S, hipStream_t s) {
  if (c % 4 || d0 % 4 || d1 % 4) return 1;
  const void* ptrs[] = {dD0, dD1, X0, X1, W0, W1, dX0, dX1, dW0, dW1, ws};
  for (const void* p : ptrs)
    if (p != nullptr && !aligned16(p)) return 1;
  const int dmax = d1 > d0 ? d1 : d0;
  if ((long)B * (dmax > c ? dmax : c) * 4 >= (1l << 31) || (long)c * (dmax + 4) * 4 >= (1l << 31)) return 1;   // 32-bit buffer offsets
  if ((long)cdiv(B, 128) * cdiv(dmax, 128) >= 65536 || (long)cdiv(c, 128) * cdiv(dmax, 128) * 16 >= 65536) return 1;   // div_magic range
  int n = 0;
  int64_t ws_used = 0;
  struct SlabInfo { float* slab; int pitch; int S; } slabs[2] = {{nullptr, 0, 1}, {nullptr, 0, 1}};
  auto wgrad = [&](const void* dD, const void* X, void* dW, void* db, int d, int m) {
    GJob j{};
    if (dD == nullptr) { j.first = n; return j; }    // no such modality (a single Linear layer's backward, linear.hip)
    j.A = (const char*)dD; j.Bm = (const char*)X; j.C = (char*)dW; j.bias = (float*)db;
    j.M = c; j.N = d; j.K = B; j.lda = c; j.ldb = d; j.ldc = d;
    j.tiles_m = cdiv(c, 128); j.tiles_n = cdiv(d, 128); j.tiles = j.tiles_m * j.tiles_n;
    j.S = 1; j.kper = B; j.kind = 1; j.m_fast = 1;
    // slices of 256 batch rows (eight stages) unless the scratch is too small
    int S = force_S > 0 ? force_S : cdiv(B, 256);
    if (S > 16) S = 16;
    const int pitch = cdiv(d + 1, 4) * 4;
    const int64_t per = (int64_t)c * pitch * 4;
    if (ws == nullptr) S = 1;
    else if ((int64_t)S * per > ws_bytes - ws_used) S = (int)((ws_bytes - ws_used) / per);
    if (S > 1) {
      j.kper = cdiv(cdiv(B, S), kGjKC) * kGjKC;
      j.S = cdiv(B, j.kper);
      if (j.S > 1) {
        j.C = (char*)ws + ws_used;
        j.ldc = pitch;
        j.slice_stride = per;
        ws_used += (int64_t)j.S * per;
        slabs[m] = SlabInfo{(float*)j.C, pitch, j.S};
      } else {
        j.kper = B;
      }
    }
    j.first = n;
    j.count = j.tiles * j.S;
    n += j.count;
    j.magic_tiles = make_magic(j.tiles);
    j.magic_inner = make_magic(j.tiles_m);
    return j;
  };
  auto dgrad = [&](const void* dD, const void* W, void* dX, int d) {
    GJob j{};
    if (dD == nullptr) { j.first = n; return j; }
    j.A = (const char*)dD; j.Bm = (const char*)W; j.C = (char*)dX;
    j.M = B; j.N = d; j.K = c; j.lda = c; j.ldb = d; j.ldc = d;
    j.tiles_m = cdiv(B, 128); j.tiles_n = cdiv(d, 128); j.tiles = j.tiles_m * j.tiles_n;
    j.S = 1; j.kper = c; j.kind = 0;
    j.m_fast = d > B ? 1 : 0;            // neighbours in the list share the LARGER operand (W panel when d > B, else the gradient rows)
    j.first = n;
    j.count = dX != nullptr ? j.tiles : 0;
    n += j.count;
    j.magic_tiles = make_magic(j.tiles);
    j.magic_inner = make_magic(j.m_fast ? j.tiles_m : j.tiles_n);
    return j;
  };
  // longest tiles first (the dispatcher hands workgroups out in list order): a weight-gradient tile multiplies kper / 32 stages,
  // an input-gradient tile c / 32
  GJob wg1, dg1, wg0, dg0;
  {
    const int S_guess = force_S > 0 ? force_S : cdiv(B, 256);
    const bool dgrad_first = c > cdiv(B, S_guess > 0 ? S_guess : 1);
    if (dgrad_first) {
      dg1 = dgrad(dD1, W1, dX1, d1); dg0 = dgrad(dD0, W0, dX0, d0);
      wg1 = wgrad(dD1, X1, dW1, db1, d1, 1); wg0 = wgrad(dD0, X0, dW0, db0, d0, 0);
    } else {
      wg1 = wgrad(dD1, X1, dW1, db1, d1, 1); wg0 = wgrad(dD0, X0, dW0, db0, d0, 0);
      dg1 = dgrad(dD1, W1, dX1, d1); dg0 = dgrad(dD0, W0, dX0, d0);
    }
  }
  GArgs ga{};
  {   // in list order (the range starts `first` ascend)
    GJob all[4] = {wg1, dg1, wg0, dg0};
    for (int a = 0; a < 4; ++a)
      for (int b = a + 1; b < 4; ++b)
        if (all[b].first < all[a].first || (all[b].first == all[a].first && all[b].count < all[a].count)) { const GJob tmp = all[a]; all[a] = all[b]; all[b] = tmp; }
    for (int a = 0; a < 4; ++a) ga.j[a] = all[a];
  }
  ga.total = n;
  gj_launch(ga, n, s);
  EMB_CHECK_LAUNCH();
  for (int m = 1; m >= 0; --m) {
    if (slabs[m].S > 1) {
      const int d = m ? d1 : d0;
      ReduceJob j{};
      j.in = slabs[m].slab; j.out[0] = m ? dW1 : dW0; j.out[1] = m ? db1 : db0;
      j.per = (long)c * slabs[m].pitch; j.S = slabs[m].S; j.kind = RJ_LINEAR; j.iv[0] = d; j.iv[1] = slabs[m].pitch;
      const int rc = reduce_submit(j, false, s);
      if (rc != EMB_OK) return rc;
    }
  }
  return EMB_OK;
}

// ---- fp32 convolutions of the stored-activation blocks as ring GEMM jobs (csrc/convblock.hip dispatches here) ---------------------
static bool gj_conv_ok(int B, int L, int cin, int N, int KK) {
  const long R = (long)B * L;
  // N >= 128: a 128-wide tile on fewer output channels leaves whole waves idle (measured at 64: slower than conv_direct.hip)
  return cin % 32 == 0 && KK % 32 == 0 && N % 4 == 0 && N >= 128 && L >= 16 && R * (cin > N ? cin : N) * 4 < (1l << 31) &&
         (long)N * KK * 4 < (1l << 31) && (R / 128 + 1) * (N / 128 + 1) < 65536;
}
// y[R][N] = conv(x[R][cin], w[N][KK]) (+ bias, + partial[tiles_m][2][N] when fwd); returns 1 when the shapes do not qualify.
// The input gradient of a block is the same job on dy with the tap-flipped packed weights.
static int gemm_jobs_conv_impl(bool fwd, const void* x, const void* w, const void* bias, void* out, void* partial, int* partial_rows, int B,
                          int L, int cin, int KK, int N, int pad, hipStream_t s) {
  if (!gj_conv_ok(B, L, cin, N, KK) || !aligned16(x) || !aligned16(w) || !aligned16(out) || (fwd && !aligned16(bias))) return 1;
  const int R = B * L;
  GJob j{};
  j.A = (const char*)x; j.Bm = (const char*)w; j.C = (char*)out;
  j.M = R; j.N = N; j.K = KK; j.lda = cin; j.ldb = KK; j.ldc = N;
  j.tiles_m = cdiv(R, 128); j.tiles_n = cdiv(N, 128); j.tiles = j.tiles_m * j.tiles_n;
  j.S = 1; j.kper = KK; j.kind = 2; j.m_fast = 0;   // n tile fastest: the tiles of one row block (they share the activation rows) are neighbours
  j.first = 0; j.count = j.tiles;
  j.magic_tiles = make_magic(j.tiles);
  j.magic_inner = make_magic(j.tiles_n);
  j.L = L; j.cin = cin; j.pad = pad; j.R = R;
  j.epi = fwd ? 1 : 0; j.bias_in = (const float*)bias; j.partial = (float*)partial;
  GArgs ga{};
  ga.j[0] = j;
  for (int a = 1; a < 4; ++a) { ga.j[a] = GJob{}; ga.j[a].first = j.count; }
  ga.total = j.count;
  gj_launch(ga, j.count, s);
  EMB_CHECK_LAUNCH();
  if (partial_rows != nullptr) *partial_rows = j.tiles_m;
  return EMB_OK;
}
// slab[S][Cout][KK + 1]: per-slice partial weight gradients (column KK = the bias gradient), S slices over the B * L rows
// (*S_io: at most that many slices -- the slab's capacity; receives the number written)
static int gemm_jobs_conv_wgrad_impl(const void* dy, const void* x, void* slab, int B, int L, int cin, int KK, int Cout, int pad, int* S_io,
                                     hipStream_t s) {
  int S = *S_io;
  if (!gj_conv_ok(B, L, cin, Cout, KK) || KK < 64 || !aligned16(dy) || !aligned16(x) || !aligned16(slab) || S < 1) return 1;
  const int R = B * L;
  while (S > 1 && cdiv(R, cdiv(cdiv(R, S), kGjKC) * kGjKC) != S) --S;   // slices of whole stages that cover the rows exactly S times
  *S_io = S;
  GJob j{};
  j.A = (const char*)dy; j.Bm = (const char*)x; j.C = (char*)slab;
  j.M = Cout; j.N = KK; j.K = R; j.lda = Cout; j.ldb = cin; j.ldc = KK + 1;
  j.tiles_m = cdiv(Cout, 128); j.tiles_n = cdiv(KK, 128); j.tiles = j.tiles_m * j.tiles_n;
  j.kper = cdiv(cdiv(R, S), kGjKC) * kGjKC;
  j.S = S; j.slice_stride = (long)Cout * (KK + 1) * 4;
  j.kind = 3; j.m_fast = 1;
  j.first = 0; j.count = j.tiles * S;
  j.magic_tiles = make_magic(j.tiles);
  j.magic_inner = make_magic(j.tiles_m);
  j.L = L; j.cin = cin; j.pad = pad; j.R = R; j.epi = 2;
  GArgs ga{};
  ga.j[0] = j;
  for (int a = 1; a < 4; ++a) { ga.j[a] = GJob{}; ga.j[a].first = j.count; }
  ga.total = j.count;
  gj_launch(ga, j.count, s);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

}  // namespace emb
