// EmbraceNet backward for bf16, LDS-DMA form (split_core.h).  One launch, four kinds of 128 x 128 tile job:
//   dgrad_m  dX_m[B,d_m]  = dD_m   W_m      128 rows x 128 columns, reduction over c
//   wgrad_m  dW_m[c,d_m]  = dD_m^T X_m      128 (c) x 128 columns, reduction over the batch rows, optionally cut into S slices
//            db_m[c]      = sum_b dD_m      (per-slice slabs summed later: reduce.hip, or by the optimizer launch)
// with dD_m = dE * [idx == m] * [pre_m > 0] never materialised: dE and the forward's code bytes travel to LDS untouched
// (LDS-DMA) and the mask is applied to each MFMA fragment as it is read.  The code byte carries the two per-modality keep
// bits the forward kernel prepared (EMB_CODE_KEEP0/1 = bits 6 / 7), so a fragment mask is v_perm_b32 (byte -> high byte of
// a 16-bit lane), a packed arithmetic shift and an AND per dword.
//
// One workgroup (8 waves as 4 x 2, two per SIMD, the whole CU) owns a 128 x 128 output tile and walks the reduction in
// ROUNDS of 128 indices; a round's operand images and code bytes (80 KB) are requested by LDS-DMA into one of two buffers,
// round r + 1 (r + 2) is in flight while round r is multiplied; the finished tile is staged through LDS and leaves as whole
// row segments (per-lane stores at a row stride are store-issue bound):
//   dgrad: wave (wr, wc) = 32 rows x 64 columns; A = W_m^T fragments (ds_read_b64_tr_b16 of the K-major image), B = dD
//          fragments (ds_read_b128 + 8 code bytes, masked once per fragment) -> C[col][row]
//   wgrad: wave (wm, wn) = 32 (c) x 64 columns; both operands K-major (image rows = batch rows), the code bytes of a
//          transposed fragment come from the matching 8-bit transposing read (ds_read_b64_tr_b8); A = X fragments,
//          B = dD^T fragments -> C[col][c]; bias gradient = a ones-fragment product (column tile 0)
// 8 MFMAs per 12-14 fragment reads and 24 mask instructions; the reads of k-step s + 1 are issued before step s is multiplied,
// and the two waves of a SIMD overlap one's vector instructions with the other's MFMAs.
// Replaces autograd through EmbraceNetMultimodal.py:52-60,80-88 (utils/training_models_multimodal.py:156) like embrace_bwd.hip.
#pragma once
#include "reduce.h"
#include "split_core.h"

namespace emb {

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

struct SplitJob {
  const __bf16* A;      // the gradient operand [B][c]: dE (masked from the code bytes per fragment), or the pre-masked dD_m
  const __bf16* Bptr;   // dgrad: W_m [c][d];  wgrad: X_m [B][d]
  void* C;              // dgrad: dX_m [B][d] bf16;  wgrad: dW_m [c][d] f32, or the slabs [S][c][pitch] when S > 1
  float* bias;          // wgrad, S == 1: db_m [c]
  int d, tiles_n, tiles;   // columns; 128-wide column tiles; tiles per slice
  int S, kper;          // wgrad: the batch is cut into S slices of kper rows
  int pitch;            // wgrad: row pitch of C in floats (d when S == 1)
  int end;              // exclusive end of this job's block range
  int mod;              // modality
  uint32_t magic_a, magic_b;   // ceil(2^32 / x) for x = tiles (wgrad: slice of a block), tiles_n (dgrad) / tiles_m (wgrad)
};
// floor(x / n) for x, n < 2^16 with magic = ceil(2^32 / n); n == 1 (whose magic does not fit 32 bits) is encoded as 0
__device__ __forceinline__ int div_magic(int x, uint32_t magic) { return magic ? (int)__umulhi((uint32_t)x, magic) : x; }
static inline uint32_t make_magic(int n) { return n <= 1 ? 0u : (uint32_t)(((1ull << 32) + (uint64_t)n - 1) / (uint64_t)n); }

// keep-mask of one dword (two bf16) from two code bytes of `cw` (sel = v_perm selector placing them in the high bytes of
// the 16-bit lanes): the keep bit must sit in bit 7 of its byte
__device__ __forceinline__ uint32_t mask_pair(uint32_t data, uint32_t cw, uint32_t sel) {
  const uint32_t m = __builtin_amdgcn_perm(0u, cw, sel);        // [code_hi, 0, code_lo, 0]
  const s16x2 s = __builtin_bit_cast(s16x2, m) >> 15;           // 0xFFFF where the keep bit is set
  return data & __builtin_bit_cast(uint32_t, s);
}
// sh = 0 for modality 1 (KEEP1 = bit 7), 1 for modality 0 (KEEP0 = bit 6: one shift of the whole code dword brings it to bit 7
// of every byte; what spills into the neighbouring byte's bit 0 is never looked at)
__device__ __forceinline__ bf16x8 mask_frag(bf16x8 v, uint32_t c_lo, uint32_t c_hi, int sh) {
  u32x4 d = __builtin_bit_cast(u32x4, v);
  c_lo <<= sh;
  c_hi <<= sh;
  d[0] = mask_pair(d[0], c_lo, 0x010c000cu);
  d[1] = mask_pair(d[1], c_lo, 0x030c020cu);
  d[2] = mask_pair(d[2], c_hi, 0x010c000cu);
  d[3] = mask_pair(d[3], c_hi, 0x030c020cu);
  return __builtin_bit_cast(bf16x8, d);
}

// per-lane address parts of the transposing fragment reads of a K-major chunk image (rows = k, 128 bytes = 64 columns):
// column tile ni (16 columns), k-step h (32 rows), second half (rows +4): kb + off[ni] + h * 4096 + second * 512
struct KmLane {
  uint32_t kb;                                    // row and 8-byte half part
  int ps, sw;                                     // p >> 1 and the slot permutation of this lane's rows
};
__device__ __forceinline__ KmLane km_lane(int lane) {
  const int g = lane >> 4, w = lane & 15, q = w >> 2, p = w & 3;
  KmLane k;
  k.kb = (uint32_t)((8 * g + q) * 128 + (p & 1) * 8);
  k.ps = p >> 1;
  k.sw = (((q >> 1) & 1) << 1) | ((g & 1) << 2);  // swz16 of rows 8g + q (+4, +32)
  return k;
}
// byte offset of column tile ni inside the image for this lane (computed, not tabulated: ni may be a run-time value and a
// run-time-indexed register array would live in scratch)
__device__ __forceinline__ uint32_t km_off(const KmLane& k, int ni) { return k.kb + (uint32_t)(((2 * ni + k.ps) ^ k.sw) << 4); }
__device__ __forceinline__ bf16x8 km_frag_at(uint32_t a) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)a);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(a + 512u));
  union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
  u.s.lo = lo;
  u.s.hi = hi;
  return u.v;
}

constexpr int kBwdThreads = 512;                  // 8 waves, two per SIMD: one wave's mask / address instructions overlap the other's MFMAs
// one round = the operand images (+ code bytes when the fragments are masked in the kernel) of 128 reduction indices.  NBUF = 2:
// round r + 1 lands while round r is multiplied, one workgroup per CU (the latency-bound single-wave-of-workgroups shapes:
// B <= 1024).  NBUF = 1 (pre-masked operands only): 68 KB per workgroup, TWO workgroups per CU -- with several waves of workgroups
// per CU (B = 4096) the second workgroup's loads, multiplies and stores fill the first one's set-up / wait / store phases.
template <bool MASKED> constexpr int kBwdBuf = MASKED ? 80 * 1024 : 64 * 1024;
constexpr int kBwdStageMin = 128 * 528 + 2048;    // the f32 weight-gradient tile staged for its row-segment stores
template <bool MASKED, int NBUF> constexpr int kBwdLds = NBUF * kBwdBuf<MASKED> > kBwdStageMin ? NBUF * kBwdBuf<MASKED> : kBwdStageMin;
template <bool MASKED> constexpr int kBwdDmaPerRound = MASKED ? 10 : 8;   // LDS-DMA instructions a wave issues per round (both job kinds); no code images for pre-masked gradients

template <bool MASKED> __device__ __forceinline__ void bwd_wait_round(bool next_in_flight) {
  if (next_in_flight) EMB_WAIT_VMCNT(kBwdDmaPerRound<MASKED>);   // the younger round stays in flight
  else EMB_WAIT_VMCNT(0);
  __builtin_amdgcn_s_barrier();                                  // raw barrier: a __syncthreads() would drain the younger round
  asm volatile("" ::: "memory");
}

// ------------------------------------------------------------------------------------------------ dgrad tile
// buffer: [dE k-chunk 0 | dE k-chunk 1] (128 rows x 128 B each) [code chunk 0 | 1] (128 rows x 64 B)
//         [W chunk 0 cols 0-63 | chunk 0 cols 64-127 | chunk 1 cols 0-63 | chunk 1 cols 64-127] (64 k-rows x 128 B each)
// wave (wr = wave >> 1, wc = wave & 1): rows 32 wr .. +32, columns 64 wc .. +64
struct DgradStep {                                // fragments of one k-step (32 k): 12 LDS reads
  bf16x8 a[4], braw[2];
  u32x2 cw[2];
};
template <bool MASKED, int NBUF>
__device__ __forceinline__ void dgrad_tile(const uint8_t* __restrict__ code, int B, int c, const SplitJob& job, int tile, char* smem) {
  const __bf16* __restrict__ dE = job.A;
  constexpr int DE_IMG = 128 * 128, CD_IMG = MASKED ? 128 * 64 : 0, W_IMG = 64 * 128;
  constexpr int CD_OFF = 2 * DE_IMG, W_OFF = CD_OFF + 2 * CD_IMG;
  static_assert(W_OFF + 4 * W_IMG == kBwdBuf<MASKED>, "dgrad buffer layout");
  EMB_STAMP(2);
  EMB_STAMP_KIND(2 + job.mod);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int tm = div_magic(tile, job.magic_b), tn = tile - tm * job.tiles_n;
  const int row0 = tm * 128, n0 = tn * 128;
  const int d = job.d;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const char* dEo = reinterpret_cast<const char*>(dE) + (long)row0 * c * 2;       // rows >= B read zeros (range check)
  const char* cdo = reinterpret_cast<const char*>(code) + (long)row0 * c;
  const char* Wo = reinterpret_cast<const char*>(job.Bptr);
  const long de_bytes = (long)(B - row0) * c * 2, cd_bytes = (long)(B - row0) * c, w_bytes = (long)c * d * 2;
  DmaImage<128, 8, 8> de;                          // every wave issues an eighth of each image's instructions
  DmaImage<128, 16, 8> dc;
  DmaImage<64, 8, 8> dw[2];
  de.init((uint32_t)c * 2, 0, 128, lane, wave);
  dc.init((uint32_t)c, 0, 64, lane, wave);
  dw[0].init((uint32_t)d * 2, n0 * 2, d * 2, lane, wave);        // columns >= d read zeros
  dw[1].init((uint32_t)d * 2, n0 * 2 + 128, d * 2, lane, wave);
  auto request = [&](int rd) {                                   // round rd: k range [128 rd, 128 rd + 128)
    const uint32_t buf = lds0 + (uint32_t)((rd % NBUF) * kBwdBuf<MASKED>);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int k0 = rd * 128 + cc * 64;
      if (k0 + 64 <= c) {
        de.issue(dEo + k0 * 2, dma_nrec(de_bytes - k0 * 2), buf + cc * DE_IMG);
        if constexpr (MASKED) dc.issue(cdo + k0, dma_nrec(cd_bytes - k0), buf + CD_OFF + cc * CD_IMG);
      } else {                                                   // k beyond c reads zeros (not the next row)
        de.issue_tail(dEo + k0 * 2, dma_nrec(de_bytes - k0 * 2), (c - k0) * 2, buf + cc * DE_IMG);
        if constexpr (MASKED) dc.issue_tail(cdo + k0, dma_nrec(cd_bytes - k0), c - k0, buf + CD_OFF + cc * CD_IMG);
      }
      const uint32_t wrec = dma_nrec(w_bytes - (long)k0 * d * 2);
      dw[0].issue(Wo + (long)k0 * d * 2, wrec, buf + W_OFF + (2 * cc) * W_IMG);
      dw[1].issue(Wo + (long)k0 * d * 2, wrec, buf + W_OFF + (2 * cc + 1) * W_IMG);
    }
  };
  const int rounds = (c + 127) / 128;
  request(0);
  if (NBUF > 1 && rounds > 1) request(1);
  EMB_STAMP(3);

  const RmLane rl = rm_lane(lane);
  const KmLane kl = km_lane(lane);
  uint32_t woff[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) woff[ni] = km_off(kl, ni);
  const int r = lane & 15, g = lane >> 4;
  uint32_t coff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) coff[h] = (uint32_t)(r * 64 + (((2 * h + (g >> 1)) ^ ((r >> 2) & 3)) << 4) + (g & 1) * 8);
  typedef __attribute__((address_space(3))) u32x2 lds_u2;
  const int sh = job.mod ? 0 : 1;
  auto load = [&](uint32_t buf, int step, DgradStep& f) {        // step = 2 * chunk + h
    const int cc = step >> 1, h = step & 1;
    const uint32_t wimg = buf + W_OFF + (uint32_t)((2 * cc + wc) * W_IMG) + (uint32_t)(h * 4096);
    const uint32_t dimg = buf + (uint32_t)(cc * DE_IMG + wr * 2 * 2048) + rl.off[h];
    const uint32_t cimg = buf + CD_OFF + (uint32_t)(cc * CD_IMG + wr * 2 * 1024) + coff[h];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) f.a[ni] = km_frag_at(wimg + woff[ni]);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      f.braw[mi] = lds_read16<__bf16>(dimg + mi * 2048);
      if constexpr (MASKED) f.cw[mi] = *(const lds_u2*)(uintptr_t)(cimg + mi * 1024);
    }
  };

  f32x4 acc[4][2];                                               // [column tile][row tile]
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[ni][mi][q] = 0.0f;

  for (int rd = 0; rd < rounds; ++rd) {
    bwd_wait_round<MASKED>(NBUF > 1 && rd + 1 < rounds);
    if (rd == 0) EMB_STAMP(4);
    const uint32_t buf = lds0 + (uint32_t)((rd % NBUF) * kBwdBuf<MASKED>);
    const int nsteps = min(4, (c - rd * 128 + 31) / 32);         // k-steps of this round that hold data
    DgradStep f[2];
    load(buf, 0, f[0]);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < nsteps) {
        if (s + 1 < nsteps) load(buf, s + 1, f[(s + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          bf16x8 b = f[s & 1].braw[mi];
          if constexpr (MASKED) b = mask_frag(b, f[s & 1].cw[mi][0], f[s & 1].cw[mi][1], sh);
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[s & 1].a[ni], b, acc[ni][mi], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (rd + NBUF < rounds) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();                              // every wave has read this buffer
      request(rd + NBUF);
    }
  }
  EMB_STAMP(5);
  // C[col][row] (lane & 15 = row inside a 16-row tile, registers = 4 consecutive columns) -> LDS tile [128 rows][128 cols]
  // bf16 -> whole 256-byte row segments to memory (per-lane 8-byte stores at a row stride are store-issue bound)
  constexpr int TP = 272;                                        // row pitch of the staged tile (bytes)
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();                                  // the last round's images are dead
  asm volatile("" ::: "memory");
  typedef __attribute__((address_space(3))) bf16x4 lds_bf4;
  typedef __attribute__((address_space(3))) bf16x8 lds_bf8;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      bf16x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = (__bf16)acc[ni][mi][q];
      *(lds_bf4*)(uintptr_t)(lds0 + (uint32_t)((wr * 32 + mi * 16 + r) * TP + (wc * 64 + ni * 16 + 4 * g) * 2)) = o;
    }
  __syncthreads();
  __bf16* out = reinterpret_cast<__bf16*>(job.C);
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int trow = pass * 32 + (int)(threadIdx.x >> 4), col8 = (int)(threadIdx.x & 15) * 8;
    const int row = row0 + trow, n = n0 + col8;
    if (row < B && n < d) {                                      // d % 8 == 0: eight columns are inside together
      const bf16x8 v = *(const lds_bf8*)(uintptr_t)(lds0 + (uint32_t)(trow * TP + col8 * 2));
      *reinterpret_cast<bf16x8*>(out + (long)row * d + n) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------ wgrad tile
// buffer: [dE chunk cc half h: 32 rows x 128 B] x 8 (cc major) [code chunk cc half h: 32 rows x 64 B] x 8 [X chunk cc half h] x 8;
// chunk cc = batch rows [k_begin + 128 round + 32 cc, +32), half h = columns [64 h, 64 h + 64) of the 128-wide window.
// Wave w requests (chunk w >> 1, half w & 1) of every round; computes c rows 32 (w >> 1) .. +32, columns 64 (w & 1) .. +64.
struct WgradStep {                                // fragments of one chunk (32 batch rows): 14 LDS reads
  bf16x8 a[4], braw[2];
  i32x2 cw[2];
};
template <bool MASKED, int NBUF>
__device__ __forceinline__ void wgrad_tile(const uint8_t* __restrict__ code, int B, int c, const SplitJob& job, int tile, int slice,
                                           char* smem) {
  const __bf16* __restrict__ dE = job.A;
  constexpr int KB = 32;
  constexpr int DE_IMG = KB * 128, CD_IMG = MASKED ? KB * 64 : 0, X_IMG = KB * 128;
  constexpr int CD_OFF = 8 * DE_IMG, X_OFF = CD_OFF + 8 * CD_IMG;
  static_assert(X_OFF + 8 * X_IMG == kBwdBuf<MASKED>, "wgrad buffer layout");
  EMB_STAMP(2);
  EMB_STAMP_KIND(job.mod);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;                        // c rows 32 wm.., columns 64 wn..
  const int tiles_m = (c + 127) >> 7;
  const int tn = div_magic(tile, job.magic_b), tmc = tile - tn * tiles_m;   // c tile fastest: tiles sharing an X panel are neighbours
  const int c0 = tmc * 128, n0 = tn * 128;
  const int k_begin = slice * job.kper, k_end = min(B, k_begin + job.kper);
  const int d = job.d;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const char* dEo = reinterpret_cast<const char*>(dE);
  const char* cdo = reinterpret_cast<const char*>(code);
  const char* Xo = reinterpret_cast<const char*>(job.Bptr);
  const int qc = wave >> 1, qh = wave & 1;                        // the (chunk, half) this wave requests
  DmaImage<KB, 8> de, dx;
  DmaImage<KB, 16> dc;
  de.init((uint32_t)c * 2, (c0 + 64 * qh) * 2, c * 2, lane);
  dc.init((uint32_t)c, c0 + 64 * qh, c, lane);
  dx.init((uint32_t)d * 2, (n0 + 64 * qh) * 2, d * 2, lane);
  auto request = [&](int rd) {
    const uint32_t buf = lds0 + (uint32_t)((rd % NBUF) * kBwdBuf<MASKED>);
    const long r0 = k_begin + rd * 128 + qc * KB;                // rows >= k_end read zeros (range check on the slice end)
    de.issue(dEo + r0 * c * 2, dma_nrec(((long)k_end - r0) * c * 2), buf + (uint32_t)((2 * qc + qh) * DE_IMG));
    if constexpr (MASKED) dc.issue(cdo + r0 * c, dma_nrec(((long)k_end - r0) * c), buf + CD_OFF + (uint32_t)((2 * qc + qh) * CD_IMG));
    dx.issue(Xo + r0 * d * 2, dma_nrec(((long)k_end - r0) * d * 2), buf + X_OFF + (uint32_t)((2 * qc + qh) * X_IMG));
  };
  const int rounds = (k_end - k_begin + 127) / 128;
  request(0);
  if (NBUF > 1 && rounds > 1) request(1);
  EMB_STAMP(3);
  const bool with_bias = (n0 == 0) && (wn == 0);

  const KmLane kl = km_lane(lane);
  uint32_t koff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) koff[i] = km_off(kl, i);
  const int chalf = wm >> 1, ct0 = (wm & 1) * 2;                  // this wave's c rows live in image half chalf, column tiles ct0, ct0 + 1
  uint32_t doff[2];
#pragma unroll
  for (int ci = 0; ci < 2; ++ci) doff[ci] = km_off(kl, ct0 + ci);
  // 8-bit transposing read: within a 16-lane group lane 2q'+p' supplies the address of row q', columns 8p' .. 8p'+7 of an
  // 8-row x 16-column byte block; lane i of the group receives column i, rows 0..7 (= the k order of the bf16 fragment)
  const int g = lane >> 4, w = lane & 15;
  uint32_t ctr[2];
#pragma unroll
  for (int ci = 0; ci < 2; ++ci) {
    const int row = 8 * g + (w >> 1);
    ctr[ci] = (uint32_t)(row * 64 + (((ct0 + ci) ^ ((row >> 2) & 3)) << 4) + (w & 1) * 8);
  }
  typedef __attribute__((address_space(3))) i32x2 lds_i32x2;
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  const int sh = job.mod ? 0 : 1;
  auto load = [&](uint32_t buf, int cc, WgradStep& f) {
    const uint32_t dimg = buf + (uint32_t)((2 * cc + chalf) * DE_IMG);
    const uint32_t cimg = buf + CD_OFF + (uint32_t)((2 * cc + chalf) * CD_IMG);
    const uint32_t ximg = buf + X_OFF + (uint32_t)((2 * cc + wn) * X_IMG);
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
      f.braw[ci] = km_frag_at(dimg + doff[ci]);
      if constexpr (MASKED) f.cw[ci] = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)(uintptr_t)(cimg + ctr[ci]));
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) f.a[ni] = km_frag_at(ximg + koff[ni]);
  };

  f32x4 acc[2][4], accb[2];                                      // [c tile][column tile]
#pragma unroll
  for (int ci = 0; ci < 2; ++ci) {
#pragma unroll
    for (int q = 0; q < 4; ++q) accb[ci][q] = 0.0f;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[ci][ni][q] = 0.0f;
  }

  for (int rd = 0; rd < rounds; ++rd) {
    bwd_wait_round<MASKED>(NBUF > 1 && rd + 1 < rounds);
    if (rd == 0) EMB_STAMP(4);
    const uint32_t buf = lds0 + (uint32_t)((rd % NBUF) * kBwdBuf<MASKED>);
    const int nsteps = min(4, (k_end - k_begin - rd * 128 + KB - 1) / KB);   // chunks of this round that hold rows
    WgradStep f[2];
    load(buf, 0, f[0]);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < nsteps) {
        if (s + 1 < nsteps) load(buf, s + 1, f[(s + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) {
          bf16x8 b = f[s & 1].braw[ci];
          if constexpr (MASKED) b = mask_frag(b, (uint32_t)f[s & 1].cw[ci][0], (uint32_t)f[s & 1].cw[ci][1], sh);
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[ci][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[s & 1].a[ni], b, acc[ci][ni], 0, 0, 0);
          if (with_bias) accb[ci] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, b, accb[ci], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (rd + NBUF < rounds) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      request(rd + NBUF);
    }
  }
  EMB_STAMP(5);
  // C[col][c] (lane & 15 = c row inside a 16-row tile, registers = 4 consecutive columns) -> LDS tile [128 c][128 cols] f32
  // -> whole 512-byte row segments to memory
  constexpr int TP = 528;                                        // row pitch of the staged tile (bytes)
  float* out = reinterpret_cast<float*>(job.C) + (job.S > 1 ? (long)slice * c * job.pitch : 0);
  const int pitch = job.pitch;
  if (with_bias && g == 0) {                                     // every register of accb holds the column sum
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
      const int crow = c0 + wm * 32 + ci * 16 + w;
      if (crow < c) {
        if (job.S > 1) out[(long)crow * pitch + d] = accb[ci][0];
        else job.bias[crow] = accb[ci][0];
      }
    }
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();                                  // the last round's images are dead
  asm volatile("" ::: "memory");
  typedef __attribute__((address_space(3))) f32x4 lds_f4;
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
      *(lds_f4*)(uintptr_t)(lds0 + (uint32_t)((wm * 32 + ci * 16 + w) * TP + (wn * 64 + ni * 16 + 4 * g) * 4)) = acc[ci][ni];
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 8; ++pass) {
    const int trow = pass * 16 + (int)(threadIdx.x >> 5), col4 = (int)(threadIdx.x & 31) * 4;
    const int crow = c0 + trow, n = n0 + col4;
    if (crow < c && n < d) {
      const f32x4 v = *(const lds_f4*)(uintptr_t)(lds0 + (uint32_t)(trow * TP + col4 * 4));
      *reinterpret_cast<f32x4*>(out + (long)crow * pitch + n) = v;
    }
  }
}

// field-wise selection of one of the four by-value job descriptors (all kernel arguments are fetched in one go; a run-time
// INDEX into a by-value argument is what hipcc 7.2 miscompiles, see embrace_bwd.hip)
__device__ __forceinline__ SplitJob pick_job(int k, const SplitJob& a, const SplitJob& b, const SplitJob& c, const SplitJob& d) {
  SplitJob j;
#define EMB_PICK(f) j.f = k == 0 ? a.f : (k == 1 ? b.f : (k == 2 ? c.f : d.f))
  EMB_PICK(A); EMB_PICK(Bptr); EMB_PICK(C); EMB_PICK(bias); EMB_PICK(d); EMB_PICK(tiles_n); EMB_PICK(tiles); EMB_PICK(S); EMB_PICK(kper);
  EMB_PICK(pitch); EMB_PICK(end); EMB_PICK(mod); EMB_PICK(magic_a); EMB_PICK(magic_b);
#undef EMB_PICK
  return j;
}

// MASKED: the gradient operand is dE and every fragment is masked from the forward's code bytes; !MASKED: the jobs' operands are
// the pre-masked dD_m the producer of dE wrote (emb_head_ce_masked / emb_embrace_premask): no code images (a fifth of the staged
// bytes, 8 instead of 10 LDS-DMA instructions per wave and round), no mask arithmetic (28 vector instructions per 8 MFMAs)
template <bool MASKED, int NBUF>
__global__ __launch_bounds__(kBwdThreads, NBUF == 1 ? 4 : 2) void embrace_bwd_split_kernel(const uint8_t* __restrict__ code, int B, int c,
                                                                      const SplitJob wg1, const SplitJob dg1,
                                                                      const SplitJob wg0, const SplitJob dg0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = blockIdx.x;
  const int kind = (bid >= wg1.end) + (bid >= dg1.end) + (bid >= wg0.end);     // 0 wgrad1, 1 dgrad1, 2 wgrad0, 3 dgrad0
  const SplitJob job = pick_job(kind, wg1, dg1, wg0, dg0);
  const int first = kind == 0 ? 0 : (kind == 1 ? wg1.end : (kind == 2 ? dg1.end : wg0.end));
  const int q = bid - first;
  if (kind & 1) {
    dgrad_tile<MASKED, NBUF>(code, B, c, job, kind == 1 ? xcd_remap(q, job.tiles) : q, smem);
  } else {
    const int slice = div_magic(q, job.magic_a), t = q - slice * job.tiles;
    wgrad_tile<MASKED, NBUF>(code, B, c, job, kind == 0 ? xcd_remap(t, job.tiles) : t, slice, smem);
  }
  EMB_STAMP(8);
}

// returns 1 when the shapes do not qualify (caller uses the tiled kernel of embrace_bwd.hip)
// dD0 / dD1 != nullptr: the pre-masked gradients (dE and code are then not read)
static int bwd_split_dispatch(const void* dE, const uint8_t* code, const void* dD0, const void* dD1, const void* X0, const void* X1, const void* W0, const void* W1,
                              void* dX0, void* dX1, void* dW0, void* db0, void* dW1, void* db1, void* ws, int64_t ws_bytes, int B,
                              int d0, int d1, int c, int force_S, hipStream_t s) {
  if (c % 16 || d0 % 8 || d1 % 8) return 1;
  const bool premasked = dD0 != nullptr && dD1 != nullptr;
  const void* ptrs[] = {dE, dD0, dD1, X0, X1, W0, W1, dX0, dX1, dW0, dW1};
  for (const void* p : ptrs)
    if (p != nullptr && !aligned16(p)) return 1;
  if (!premasked && (reinterpret_cast<uintptr_t>(code) & 15u) != 0) return 1;
  const long big = (long)B * (d1 > c ? d1 : c) * 2;
  if (big >= (1l << 31) || (long)c * d1 * 2 >= (1l << 31)) return 1;     // 32-bit buffer offsets (split_core.h)
  if ((long)cdiv(B, 128) * cdiv(d1, 128) >= 65536 || (long)cdiv(c, 128) * cdiv(d1, 128) * 16 >= 65536) return 1;   // div_magic range
  int n = 0;
  int64_t ws_used = 0;
  struct SlabInfo { float* slab; int pitch; int S; } slabs[2] = {{nullptr, 0, 1}, {nullptr, 0, 1}};
  auto wgrad = [&](const void* X, void* dW, void* db, int d, int m) {
    SplitJob j{};
    j.A = (const __bf16*)(premasked ? (m ? dD1 : dD0) : dE);
    j.Bptr = (const __bf16*)X; j.C = dW; j.bias = (float*)db; j.d = d;
    j.tiles_n = cdiv(d, 128);
    j.tiles = cdiv(c, 128) * j.tiles_n;
    j.S = 1; j.kper = B; j.pitch = d;
    // a workgroup takes 128 batch rows per round and keeps two rounds in flight: slices of 256 rows are requested whole
    int S = force_S > 0 ? force_S : cdiv(B, 256);
    if (S > 16) S = 16;
    const int pitch = cdiv(d + 1, 4) * 4;
    const int64_t per = (int64_t)c * pitch * 4;
    if (ws == nullptr) S = 1;
    else if ((int64_t)S * per > ws_bytes - ws_used) S = (int)((ws_bytes - ws_used) / per);
    if (S > 1) {
      j.kper = cdiv(cdiv(B, S), 32) * 32;
      j.S = cdiv(B, j.kper);
      if (j.S > 1) {
        j.C = (char*)ws + ws_used;
        j.pitch = pitch;
        ws_used += (int64_t)j.S * per;
        slabs[m] = SlabInfo{(float*)j.C, pitch, j.S};
      } else {
        j.kper = B;
      }
    }
    n += j.tiles * j.S;
    j.end = n;
    j.mod = m;
    j.magic_a = make_magic(j.tiles);
    j.magic_b = make_magic(cdiv(c, 128));
    return j;
  };
  auto dgrad = [&](const void* W, void* dX, int d, int m) {
    SplitJob j{};
    j.A = (const __bf16*)(premasked ? (m ? dD1 : dD0) : dE);
    j.Bptr = (const __bf16*)W; j.C = dX; j.d = d;
    j.tiles_n = cdiv(d, 128);
    j.tiles = cdiv(B, 128) * j.tiles_n;
    if (dX != nullptr) n += j.tiles;
    j.end = n;
    j.mod = m;
    j.magic_a = make_magic(j.tiles);
    j.magic_b = make_magic(j.tiles_n);
    return j;
  };
  const SplitJob wg1 = wgrad(X1, dW1, db1, d1, 1);
  const SplitJob dg1 = dgrad(W1, dX1, d1, 1);
  const SplitJob wg0 = wgrad(X0, dW0, db0, d0, 0);
  const SplitJob dg0 = dgrad(W0, dX0, d0, 0);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&embrace_bwd_split_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, kBwdLds<true, 2>);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&embrace_bwd_split_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, kBwdLds<false, 2>);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&embrace_bwd_split_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, kBwdLds<false, 1>);
    attr_set = true;
  }
  static const int cus = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
    return v;
  }();
  if (premasked && n >= cus + cus / 2) embrace_bwd_split_kernel<false, 1><<<n, kBwdThreads, kBwdLds<false, 1>, s>>>(code, B, c, wg1, dg1, wg0, dg0);
  else if (premasked) embrace_bwd_split_kernel<false, 2><<<n, kBwdThreads, kBwdLds<false, 2>, s>>>(code, B, c, wg1, dg1, wg0, dg0);
  else embrace_bwd_split_kernel<true, 2><<<n, kBwdThreads, kBwdLds<true, 2>, s>>>(code, B, c, wg1, dg1, wg0, dg0);
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

}  // namespace emb
