// EmbraceNet backward, "K split over waves" form (split_core.h), bf16.  One launch, four kinds of tile job:
//   dgrad_m  dX_m[B,d_m]  = dD_m   W_m      tile 64 rows x 64 cols, reduction over c
//   wgrad_m  dW_m[c,d_m]  = dD_m^T X_m      tile 64 (c) x 64 cols, reduction over the batch rows (optionally sliced
//            db_m[c]      = sum_b dD_m                             over S workgroups -> per-slice slabs, reduce.hip)
// with dD_m = dE * [idx == m] * [pre_m > 0] never materialised: dE and the forward's code bytes travel to LDS untouched
// (LDS-DMA) and the mask is applied to each MFMA fragment as it is read -- with the reduction split over waves every
// element is fragment-read exactly once, so this is the minimum mask work, and it needs no register staging.
// The code byte carries the two per-modality keep bits the forward kernel prepared (EMB_CODE_KEEP0/1 = bits 6 / 7), so
// a fragment mask is v_perm_b32 (byte -> high byte of a 16-bit lane), a packed arithmetic shift and an AND per dword.
//
// dgrad: dE tile rows are row-major in k (ds_read_b128), W_m rows are k (K-major: ds_read_b64_tr_b16).
// wgrad: both operands are K-major (image rows = batch rows); the code bytes of a transposed fragment come from the
//        matching 8-bit transposing read (ds_read_b64_tr_b8).
// Replaces autograd through EmbraceNetMultimodal.py:52-60,80-88 (utils/training_models_multimodal.py:156) like embrace_bwd.hip.
#pragma once
#include "reduce.h"
#include "split_core.h"

namespace emb {

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

struct SplitJob {
  const __bf16* Bptr;   // dgrad: W_m [c][d];  wgrad: X_m [B][d]
  void* C;              // dgrad: dX_m [B][d] bf16;  wgrad: dW_m [c][d] f32, or the slabs [S][c][pitch] when S > 1
  float* bias;          // wgrad, S == 1: db_m [c]
  int d, tiles_n, tiles;   // columns; 64-wide column tiles; tiles per slice
  int S, kper;          // wgrad: the batch is cut into S slices of kper rows
  int pitch;            // wgrad: row pitch of C in floats (d when S == 1)
  int end;              // exclusive end of this job's block range
};

// keep-mask of one dword (two bf16) from the code bytes `lo`, `lo + 1` of `cw` (sel = v_perm selector)
template <int MOD> __device__ __forceinline__ uint32_t mask_pair(uint32_t data, uint32_t cw, uint32_t sel) {
  const uint32_t m = __builtin_amdgcn_perm(0u, cw, sel);        // [code_hi, 0, code_lo, 0]
  s16x2 s = __builtin_bit_cast(s16x2, m);
  if (MOD == 0) s = s << 1;                                     // KEEP0 is bit 6, KEEP1 bit 7
  s = s >> 15;                                                  // 0xFFFF where the keep bit is set
  return data & __builtin_bit_cast(uint32_t, s);
}
template <int MOD> __device__ __forceinline__ bf16x8 mask_frag(bf16x8 v, uint32_t c_lo, uint32_t c_hi) {
  u32x4 d = __builtin_bit_cast(u32x4, v);
  d[0] = mask_pair<MOD>(d[0], c_lo, 0x010c000cu);
  d[1] = mask_pair<MOD>(d[1], c_lo, 0x030c020cu);
  d[2] = mask_pair<MOD>(d[2], c_hi, 0x010c000cu);
  d[3] = mask_pair<MOD>(d[3], c_hi, 0x030c020cu);
  return __builtin_bit_cast(bf16x8, d);
}

// Code images: ROWS batch rows x 64 bytes (dgrad: bytes = k; wgrad: bytes = c columns), 16 rows per LDS-DMA instruction
// (DmaRows*<ROWS, 16>); the four 16-byte slots of a row are permuted by (row >> 2) & 3, which makes both the ds_read_b64 of
// a row-major fragment's 8 code bytes and the ds_read_b64_tr_b8 of a transposed fragment's bank-conflict free.

// per-lane address parts of the transposing fragment reads of a K-major chunk image (rows = k, 128 bytes = 64 columns):
// tile ni (16 columns), k-step h (32 rows), second half (rows +4): kb + off[ni] + h * 4096 + second * 512
struct KmLane {
  uint32_t kb;
  uint32_t off[4];
};
__device__ __forceinline__ KmLane km_lane(int lane) {
  const int g = lane >> 4, w = lane & 15, q = w >> 2, p = w & 3;
  const int s = (((q >> 1) & 1) << 1) | ((g & 1) << 2);          // swz16 of rows 8g + q (+4, +32)
  KmLane k;
  k.kb = (uint32_t)((8 * g + q) * 128 + (p & 1) * 8);
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) k.off[ni] = (uint32_t)(((2 * ni + (p >> 1)) ^ s) << 4);
  return k;
}
__device__ __forceinline__ bf16x8 km_frag(uint32_t img, const KmLane& k, int ni, int h) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const uint32_t a = img + k.kb + k.off[ni] + (uint32_t)(h * 4096);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)a);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(a + 512u));
  union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
  u.s.lo = lo;
  u.s.hi = hi;
  return u.v;
}

constexpr int kBwdCS = 68;                        // pitch (floats) of a partial tile in LDS: 64 columns + bias + pad
constexpr int kBwdSlab = 64 * kBwdCS;

// partial 64x64 tiles of the four waves -> LDS; on return (after the barrier) part[w * kBwdSlab + row * kBwdCS + col]
__device__ __forceinline__ void park_partials(const f32x4 (&acc)[4][4], float* part, int wave, int lane) {
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        part[wave * kBwdSlab + (mi * 16 + 4 * (lane >> 4) + q) * kBwdCS + ni * 16 + (lane & 15)] = acc[mi][ni][q];
}

// ------------------------------------------------------------------------------------------------ dgrad tile
template <int MOD, int NSTAGE>
__device__ __forceinline__ void dgrad_tile(const __bf16* __restrict__ dE, const uint8_t* __restrict__ code, int B, int c,
                                           const SplitJob& job, int tile, char* smem) {
  constexpr int A_BYTES = 64 * 128, C_BYTES = 64 * 64, STAGE = A_BYTES + C_BYTES + 64 * 128;
  constexpr int G = 8 + 4 + 8;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row0 = (tile / job.tiles_n) * 64, n0 = (tile % job.tiles_n) * 64;
  const uint32_t ring = (uint32_t)(uintptr_t)smem + (uint32_t)(wave * NSTAGE * STAGE);
  const char* dEb = reinterpret_cast<const char*>(dE);
  const char* Wb = reinterpret_cast<const char*>(job.Bptr);
  const int d = job.d;
  const int nch = (c + 63) / 64;
  const int n_my = nch > wave ? (nch - wave + 3) / 4 : 0;
  DmaRowsK<64, 8> de;
  DmaRowsK<64, 16> dc;
  DmaRowsR<64, 8> dw;
  de.init(dEb, (long)c * 2, row0, B, lane);
  dc.init(reinterpret_cast<const char*>(code), (long)c, row0, B, lane);
  dw.init(Wb, (long)d * 2, n0 * 2, d * 2, lane);
  auto issue = [&](int i, uint32_t st) {         // i-th chunk of this wave: k range [64 ch, 64 ch + 64)
    const int ch = wave + 4 * i;
    de.issue(ch * 128, 128, c * 2, st);
    dc.issue(ch * 64, 64, c, st + A_BYTES);
    dw.issue(ch * 64, (long)ch * 64 * d * 2, c, st + A_BYTES + C_BYTES);
  };
#pragma unroll
  for (int s = 0; s < NSTAGE; ++s)
    if (s < n_my) issue(s, ring + s * STAGE);

  f32x4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[mi][ni][q] = 0.0f;

  const RmLane rl = rm_lane(lane);
  const KmLane kl = km_lane(lane);
  const int r = lane & 15, g = lane >> 4;
  uint32_t coff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) coff[h] = (uint32_t)(r * 64 + (((2 * h + (g >> 1)) ^ ((r >> 2) & 3)) << 4) + (g & 1) * 8);
  typedef __attribute__((address_space(3))) u32x2 lds_u2;

  for (int it = 0; it < n_my; ++it) {
    wait_chunks_in_flight<G>(min(n_my - it - 1, NSTAGE - 1));
    const uint32_t st = ring + (uint32_t)((it % NSTAGE) * STAGE);
    bf16x8 a[2][4], b[2][4];
    u32x2 cw[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        a[h][mi] = lds_read16<__bf16>(st + mi * 2048 + rl.off[h]);
        cw[h][mi] = *(const lds_u2*)(uintptr_t)(st + A_BYTES + mi * 1024 + coff[h]);
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) b[h][ni] = km_frag(st + A_BYTES + C_BYTES, kl, ni, h);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (it + NSTAGE < n_my) issue(it + NSTAGE, st);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) a[h][mi] = mask_frag<MOD>(a[h][mi], cw[h][mi][0], cw[h][mi][1]);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[h][mi], b[h][ni], acc[mi][ni], 0, 0, 0);
    }
  }
  __syncthreads();                               // every wave is done with its ring
  float* part = reinterpret_cast<float*>(smem);
  park_partials(acc, part, wave, lane);
  __syncthreads();
  __bf16* out = reinterpret_cast<__bf16*>(job.C);
#pragma unroll
  for (int i2 = 0; i2 < 2; ++i2) {
    const int item = threadIdx.x + 256 * i2, row = item >> 3, n8 = (item & 7) * 8;
    f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      s0 += *reinterpret_cast<const f32x4*>(part + w * kBwdSlab + row * kBwdCS + n8);
      s1 += *reinterpret_cast<const f32x4*>(part + w * kBwdSlab + row * kBwdCS + n8 + 4);
    }
    if (row0 + row < B && n0 + n8 < d) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = (__bf16)s0[e]; o[4 + e] = (__bf16)s1[e]; }
      *reinterpret_cast<bf16x8*>(out + (long)(row0 + row) * d + n0 + n8) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------ wgrad tile
template <int MOD, int NSTAGE>
__device__ __forceinline__ void wgrad_tile(const __bf16* __restrict__ dE, const uint8_t* __restrict__ code, int B, int c,
                                           const SplitJob& job, int tile, int slice, char* smem) {
  constexpr int KB = 32;                                        // batch rows per chunk = one MFMA k-step
  constexpr int A_BYTES = KB * 128, C_BYTES = KB * 64, STAGE = A_BYTES + C_BYTES + KB * 128;
  constexpr int G = KB / 8 + KB / 16 + KB / 8;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles_m = job.tiles / job.tiles_n;
  const int c0 = (tile % tiles_m) * 64, n0 = (tile / tiles_m) * 64;   // c tile fastest: tiles sharing an X panel are neighbours
  const int k_begin = slice * job.kper, k_end = min(B, k_begin + job.kper);
  const uint32_t ring = (uint32_t)(uintptr_t)smem + (uint32_t)(wave * NSTAGE * STAGE);
  const char* dEb = reinterpret_cast<const char*>(dE);
  const char* Xb = reinterpret_cast<const char*>(job.Bptr);
  const int d = job.d;
  const int nch = (k_end - k_begin + KB - 1) / KB;
  const int n_my = nch > wave ? (nch - wave + 3) / 4 : 0;
  const bool with_bias = n0 == 0;
  DmaRowsR<KB, 8> de, dx;
  DmaRowsR<KB, 16> dc;
  de.init(dEb, (long)c * 2, c0 * 2, c * 2, lane);
  dc.init(reinterpret_cast<const char*>(code), (long)c, c0, c, lane);
  dx.init(Xb, (long)d * 2, n0 * 2, d * 2, lane);
  auto issue = [&](int i, uint32_t st) {         // i-th chunk of this wave: batch rows [r0, r0 + KB), rows >= k_end read zeros
    const int r0 = k_begin + (wave + 4 * i) * KB;
    de.issue(r0, (long)r0 * c * 2, k_end, st);
    dc.issue(r0, (long)r0 * c, k_end, st + A_BYTES);
    dx.issue(r0, (long)r0 * d * 2, k_end, st + A_BYTES + C_BYTES);
  };
#pragma unroll
  for (int s = 0; s < NSTAGE; ++s)
    if (s < n_my) issue(s, ring + s * STAGE);

  f32x4 acc[4][4], accb[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
    for (int q = 0; q < 4; ++q) accb[mi][q] = 0.0f;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[mi][ni][q] = 0.0f;
  }
  const KmLane kl = km_lane(lane);
  // 8-bit transposing read: within a 16-lane group lane 2q'+p' supplies the address of row q', columns 8p' .. 8p'+7 of an
  // 8-row x 16-column byte block; lane i of the group receives column i, rows 0..7 (= the k order of the bf16 fragment)
  const int g = lane >> 4, w = lane & 15;
  uint32_t ctr[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int row = 8 * g + (w >> 1);
    ctr[mi] = (uint32_t)(row * 64 + ((mi ^ ((row >> 2) & 3)) << 4) + (w & 1) * 8);
  }
  typedef __attribute__((address_space(3))) i32x2 lds_i32x2;
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

  for (int it = 0; it < n_my; ++it) {
    wait_chunks_in_flight<G>(min(n_my - it - 1, NSTAGE - 1));
    const uint32_t st = ring + (uint32_t)((it % NSTAGE) * STAGE);
    bf16x8 a[4], b[4];
    i32x2 cw[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      a[mi] = km_frag(st, kl, mi, 0);
      cw[mi] = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)(uintptr_t)(st + A_BYTES + ctr[mi]));
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) b[ni] = km_frag(st + A_BYTES + C_BYTES, kl, ni, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (it + NSTAGE < n_my) issue(it + NSTAGE, st);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) a[mi] = mask_frag<MOD>(a[mi], (uint32_t)cw[mi][0], (uint32_t)cw[mi][1]);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    if (with_bias) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) accb[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], ones, accb[mi], 0, 0, 0);
    }
  }
  __syncthreads();                               // every wave is done with its ring
  float* part = reinterpret_cast<float*>(smem);
  park_partials(acc, part, wave, lane);
  if (with_bias && (lane & 15) == 0) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int q = 0; q < 4; ++q) part[wave * kBwdSlab + (mi * 16 + 4 * (lane >> 4) + q) * kBwdCS + 64] = accb[mi][q];
  }
  __syncthreads();
  float* out = reinterpret_cast<float*>(job.C) + (job.S > 1 ? (long)slice * c * job.pitch : 0);
  const int pitch = job.pitch;
#pragma unroll
  for (int i2 = 0; i2 < 2; ++i2) {
    const int item = threadIdx.x + 256 * i2, row = item >> 3, n8 = (item & 7) * 8;
    f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) {
      s0 += *reinterpret_cast<const f32x4*>(part + wv * kBwdSlab + row * kBwdCS + n8);
      s1 += *reinterpret_cast<const f32x4*>(part + wv * kBwdSlab + row * kBwdCS + n8 + 4);
    }
    if (c0 + row < c && n0 + n8 < d) {
      float* p = out + (long)(c0 + row) * pitch + n0 + n8;
      *reinterpret_cast<f32x4*>(p) = s0;
      *reinterpret_cast<f32x4*>(p + 4) = s1;
    }
  }
  if (with_bias && threadIdx.x < 64 && c0 + threadIdx.x < c) {
    const int row = threadIdx.x;
    const float sb = ((part[row * kBwdCS + 64] + part[kBwdSlab + row * kBwdCS + 64]) + part[2 * kBwdSlab + row * kBwdCS + 64]) +
                     part[3 * kBwdSlab + row * kBwdCS + 64];
    if (job.S > 1) out[(long)(c0 + row) * pitch + d] = sb;
    else job.bias[c0 + row] = sb;
  }
}

template <int NSTAGE_D, int NSTAGE_W>
__global__ __launch_bounds__(kThreads, 2) void embrace_bwd_split_kernel(const __bf16* __restrict__ dE, const uint8_t* __restrict__ code,
                                                                      int B, int c, const SplitJob wg1, const SplitJob dg1,
                                                                      const SplitJob wg0, const SplitJob dg0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = blockIdx.x;
  if (bid < wg1.end) {
    wgrad_tile<1, NSTAGE_W>(dE, code, B, c, wg1, xcd_remap(bid % wg1.tiles, wg1.tiles), bid / wg1.tiles, smem);
  } else if (bid < dg1.end) {
    dgrad_tile<1, NSTAGE_D>(dE, code, B, c, dg1, xcd_remap(bid - wg1.end, dg1.end - wg1.end), smem);
  } else if (bid < wg0.end) {
    const int q = bid - dg1.end;
    wgrad_tile<0, NSTAGE_W>(dE, code, B, c, wg0, q % wg0.tiles, q / wg0.tiles, smem);
  } else {
    dgrad_tile<0, NSTAGE_D>(dE, code, B, c, dg0, bid - wg0.end, smem);
  }
}

// returns 1 when the shapes do not qualify (caller uses the tiled kernel of embrace_bwd.hip)
static int bwd_split_dispatch(const void* dE, const uint8_t* code, const void* X0, const void* X1, const void* W0, const void* W1,
                              void* dX0, void* dX1, void* dW0, void* db0, void* dW1, void* db1, void* ws, int64_t ws_bytes, int B,
                              int d0, int d1, int c, int force_S, hipStream_t s) {
  if (c % 16 || d0 % 8 || d1 % 8) return 1;
  const void* ptrs[] = {dE, X0, X1, W0, W1, dX0, dX1, dW0, dW1};
  for (const void* p : ptrs)
    if (p != nullptr && !aligned16(p)) return 1;
  if ((reinterpret_cast<uintptr_t>(code) & 15u) != 0) return 1;
  constexpr int NSTAGE_D = 1, NSTAGE_W = 2;
  constexpr int lds_d = 4 * NSTAGE_D * (64 * 128 + 64 * 64 + 64 * 128), lds_w = 4 * NSTAGE_W * (32 * 320);
  constexpr int lds_p = 4 * kBwdSlab * 4;
  constexpr int lds = lds_d > lds_w ? (lds_d > lds_p ? lds_d : lds_p) : (lds_w > lds_p ? lds_w : lds_p);
  int n = 0;
  int64_t ws_used = 0;
  struct SlabInfo { float* slab; int pitch; int S; } slabs[2] = {{nullptr, 0, 1}, {nullptr, 0, 1}};
  auto wgrad = [&](const void* X, void* dW, void* db, int d, int m) {
    SplitJob j{};
    j.Bptr = (const __bf16*)X; j.C = dW; j.bias = (float*)db; j.d = d;
    j.tiles_n = cdiv(d, 64);
    j.tiles = cdiv(c, 64) * j.tiles_n;
    j.S = 1; j.kper = B; j.pitch = d;
    // slice the batch when a tile would stream more than ~1k rows (a workgroup pulls ~0.3 KB per row)
    int S = force_S > 0 ? force_S : cdiv(B, 1024);
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
    return j;
  };
  auto dgrad = [&](const void* W, void* dX, int d) {
    SplitJob j{};
    j.Bptr = (const __bf16*)W; j.C = dX; j.d = d;
    j.tiles_n = cdiv(d, 64);
    j.tiles = cdiv(B, 64) * j.tiles_n;
    if (dX != nullptr) n += j.tiles;
    j.end = n;
    return j;
  };
  const SplitJob wg1 = wgrad(X1, dW1, db1, d1, 1);
  const SplitJob dg1 = dgrad(W1, dX1, d1);
  const SplitJob wg0 = wgrad(X0, dW0, db0, d0, 0);
  const SplitJob dg0 = dgrad(W0, dX0, d0);
  auto kern = &embrace_bwd_split_kernel<NSTAGE_D, NSTAGE_W>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  kern<<<n, kThreads, lds, s>>>((const __bf16*)dE, code, B, c, wg1, dg1, wg0, dg0);
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
