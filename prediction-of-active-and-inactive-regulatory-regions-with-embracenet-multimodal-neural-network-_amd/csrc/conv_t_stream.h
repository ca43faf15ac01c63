// Forward / input-gradient convolution of a stored-activation block (CNN_pre.py:37-50, blocks 2..4; the input gradient is the same
// convolution on dy with tap-flipped weights), bf16, 32 * CPT input channels, whole sequences per 512-row tile.  Same MFMA mapping
// and epilogue as conv_t_kernel (channels on the M axis, 16-byte row-major output stores, BatchNorm partial sums in the forward);
// what changes is how the operands reach LDS and how the inner loop addresses them.  Included by conv_direct.hip.
//
// conv_t_kernel spends half of its time (9.7 of 19 us at the A549 block-2 shape) before the first MFMA: 61 KB of weights and
// 70 KB of activations go global -> registers -> LDS with index arithmetic per element.  Here
//   * the weights arrive by LDS-DMA as FRAGMENT-READY 1 KiB blocks, block (k-step ks, channel tile mt) = what the 64 lanes of
//     a wave read for that MFMA step (lane l at byte 16 l): the permutation is in the per-lane SOURCE offset, formed once; an A
//     fragment read is ds_read_b128 at "running block pointer + immediate";
//   * the activation tile arrives by LDS-DMA in the padded-pitch layout the B reads want (cin * 2 + 32 bytes per row:
//     conflict-free 16-byte row reads): LDS slot s = 64 j + lane of instruction j is (row s / (RS + 2), column s % (RS + 2)), the
//     pad column and the halo rows are lanes whose offset is out of range (they write zeros);
//   * a tap is one pitch further down: the loop over taps adds the pitch to NT address registers, everything inside a tap is
//     an immediate.  No other address arithmetic in the loop.
#pragma once
#include "conv_tiles.h"
#include "split_core.h"

namespace emb {

constexpr int kCtsWaves = 8, kCtsNT = 4, kCtsBT = kCtsWaves * 16 * kCtsNT;   // 512 output rows per tile (NT = 4 row tiles per wave);
                                                                             // NT = 2 (256 rows) when the 512-row image does not fit in LDS
constexpr int kCtsMaxXI = 12;                                                 // activation LDS-DMA instructions per wave

struct CtsArgs {
  const __bf16* x;        // [B][L][cin]
  const __bf16* w;        // [N][KK] packed weights (tap-major)
  const float* bias;      // FWD
  __bf16* out;            // [B][L][N]
  float* partial;         // FWD: [nblk_m][2][N]
  int B, L, cin, KK, N, pad, SB, slot, tiles_m, tpb, nblk_m, taps;
};

template <int MT, int CPT, bool FWD, int NT = kCtsNT>   // channel tiles per workgroup; k-steps per tap (cin = 32 * CPT); row tiles per wave
__device__ __forceinline__ void conv_t_stream_body(const CtsArgs& a, const int block) {
  using T = __bf16;
  constexpr int WAVES = kCtsWaves, BN = 16 * MT, CPL = 4 * MT;
  constexpr int RS = 4 * CPT, PS = RS + 2, PITCH = PS * 16;              // 16-byte slots per activation row: real, with pad; bytes
  // (two pad slots: ds_read_b128 is served in 16-lane groups on 64 banks; with a pitch of an odd number of slots seven of eight
  // slots of a group collide two-way -- SQ_LDS_BANK_CONFLICT 2.3 per LDS instruction cycle --, a pitch of 2 mod 4 slots is free)
  extern __shared__ __attribute__((aligned(16))) char arena[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), g = lane >> 4, r16 = lane & 15;
  const int tn = block / a.nblk_m, bm = block % a.nblk_m, col0 = tn * BN;
  const int tm_begin = bm * a.tpb, tm_end = min(a.tiles_m, tm_begin + a.tpb);
  const int L = a.L, SB = a.SB, slot = a.slot, N = a.N, KK = a.KK;
  const int nks = KK / 32, xrows = SB * slot + kXExtra;
  const int nxi = (xrows * PS + 63) / 64;                                // activation LDS-DMA instructions per tile
  const uint32_t lds0 = (uint32_t)(uintptr_t)arena;
  const uint32_t wbase = lds0 + (uint32_t)(nxi * 1024);                  // weights after the activation image
  float* red = reinterpret_cast<float*>(arena + (size_t)nxi * 1024 + (size_t)nks * MT * 1024);   // [WAVES][2][BN]

#if defined(__HIP_DEVICE_COMPILE__)
  // ---- weights: blocks b = ks * MT + mt, this wave issues b = wave, wave + 8, ... (mt = wave % MT for all of them)
  {
    const int mt = wave % MT, ch = col0 + chan_of<T, MT>(mt, r16);
    const uint32_t wv = ch < N ? (uint32_t)(ch * KK * 2 + g * 16) : kDmaInvalid;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, N * KK * 2, 0x00020000);
    for (int b = wave; b < nks * MT; b += 8)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(uintptr_t)(wbase + (uint32_t)b * 1024u), 16, wv, (b / MT) * 64, 0, 0);
  }
#endif
  // ---- activation tile plan (tile independent): instruction j = wave + 8 i covers LDS slots 64 j .. 64 j + 63
  uint32_t xv[kCtsMaxXI];
#pragma unroll
  for (int i = 0; i < kCtsMaxXI; ++i) {
    const int s = 64 * (wave + 8 * i) + lane, row = s / PS, col = s - row * PS;
    const int sq = row / slot, tt = row - sq * slot - a.pad;
    const bool ok = col < RS && sq < SB && tt >= 0 && tt < L;
    xv[i] = ok ? (uint32_t)((sq * L + tt) * (RS * 16) + col * 16) : kDmaInvalid;
  }
  auto issue_x = [&](int tm) {
#if defined(__HIP_DEVICE_COMPILE__)
    const long b0 = (long)tm * SB, left = ((long)a.B - b0) * L * (RS * 16);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + b0 * L * a.cin), 0,
                                                                        (int)(left < 0x7fffffffL ? left : 0x7fffffffL), 0x00020000);
#pragma unroll
    for (int i = 0; i < kCtsMaxXI; ++i)
      if (wave + 8 * i < nxi)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(uintptr_t)(lds0 + (uint32_t)(wave + 8 * i) * 1024u), 16, xv[i], 0, 0, 0);
#endif
  };
  if (tm_begin < tm_end) issue_x(tm_begin);

  // ---- this lane's output rows (tile independent)
  uint32_t xaddr0[NT];                                 // LDS byte address of (row's tap-0 activation row, k group g)
  int row_pk[NT];                                      // (sequence slot << 16) | position; -1: row past the tile's sequences
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int row = wave * (NT * 16) + nt * 16 + r16;
    const int rr = min(row, SB * L - 1), sq = rr / L;
    xaddr0[nt] = lds0 + (uint32_t)((sq * slot + (rr - sq * L)) * PITCH + g * 16);
    row_pk[nt] = row >= SB * L ? -1 : ((sq << 16) | (rr - sq * L));
  }
  float bv[MT][4], s1[MT][4], s2[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bv[mt][r] = FWD ? a.bias[min(col0 + g * CPL + mt * 4 + r, N - 1)] : 0.0f;
      s1[mt][r] = 0.0f;
      s2[mt][r] = 0.0f;
    }
  const uint32_t wlane = wbase + (uint32_t)lane * 16u;

  for (int tm = tm_begin; tm < tm_end; ++tm) {
    const int b0 = tm * SB;
    __syncthreads();                                   // (drains this wave's LDS-DMA) weights and tile are in LDS
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mt][nt][r] = 0.0f;
    uint32_t xa[NT], wa = wlane;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) xa[nt] = xaddr0[nt];
#pragma unroll 2
    for (int tap = 0; tap < a.taps; ++tap) {
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        bf16x8 af[MT], bf[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = lds_read16<T>(wa + (uint32_t)((c * MT + mt) * 1024));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = lds_read16<T>(xa[nt] + (uint32_t)(c * 64));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
      }
      wa += (uint32_t)(CPT * MT * 1024);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) xa[nt] += (uint32_t)PITCH;
    }
    if (tm + 1 < tm_end) {
      __syncthreads();                                 // every wave is done reading this tile
      issue_x(tm + 1);                                 // in flight during the epilogue
    }
    // ---- epilogue: straight from the accumulators (16-byte row-major stores)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int pk = row_pk[nt];
      if (pk < 0 || b0 + (pk >> 16) >= a.B) continue;
      T ov[CPL];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[mt][nt][r] + bv[mt][r];
          ov[mt * 4 + r] = (T)v;
          if (FWD) { s1[mt][r] += v; s2[mt][r] += v * v; }
        }
      const int col = col0 + g * CPL;
      T* dst = a.out + ((long)(b0 + (pk >> 16)) * L + (pk & 0xffff)) * N + col;
      if (CPL % 8 == 0 && col + CPL <= N) {
#pragma unroll
        for (int qv = 0; qv < CPL / 8; ++qv) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = ov[qv * 8 + e];
          *reinterpret_cast<bf16x8*>(dst + qv * 8) = o;
        }
      } else if (CPL == 4 && col + 4 <= N) {
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
        bf16x4_t o = {ov[0], ov[1], ov[2], ov[3]};
        *reinterpret_cast<bf16x4_t*>(dst) = o;
      } else {
#pragma unroll
        for (int j = 0; j < CPL; ++j)
          if (col + j < N) dst[j] = ov[j];
      }
    }
  }

  if (FWD) {   // one partial row per workgroup: the 16 row lanes of a channel set meet, then the waves
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sa = row16_sum<float>(s1[mt][r]), sb = row16_sum<float>(s2[mt][r]);
        if (r16 == 0) {
          red[(wave * 2 + 0) * BN + g * CPL + mt * 4 + r] = sa;
          red[(wave * 2 + 1) * BN + g * CPL + mt * 4 + r] = sb;
        }
      }
    __syncthreads();
    if (threadIdx.x < 2 * BN) {
      const int c = threadIdx.x % BN, which = threadIdx.x / BN;
      if (col0 + c < N) {
        float t = 0.0f;
#pragma unroll
        for (int wv = 0; wv < WAVES; ++wv) t += red[(wv * 2 + which) * BN + c];
        a.partial[((long)bm * 2 + which) * N + col0 + c] = t;
      }
    }
  }
}

template <int MT, int CPT, bool FWD, int NT = kCtsNT>
__global__ __launch_bounds__(kCtsWaves * 64, 2) void conv_t_stream_kernel(const CtsArgs a) {
  conv_t_stream_body<MT, CPT, FWD, NT>(a, (int)blockIdx.x);
}

// LDS bytes of the streaming kernel, 0 when the shape does not qualify
template <int MT> static size_t conv_t_stream_lds(int B, int L, int cin, int KK, int N, int pad, int BT = kCtsBT) {
  if (cin % 32 != 0 || cin > 128 || cin == 96 || KK % cin != 0 || L > BT || L < 1) return 0;
  const ConvTiling t = conv_tiling_bt(B, L, pad, BT);
  if (t.tiles_t != 1) return 0;
  const int PS = cin / 8 + 2, xrows = t.SB * t.slot + kXExtra, nxi = (xrows * PS + 63) / 64, nks = KK / 32;
  if (nxi > 8 * kCtsMaxXI) return 0;                                     // plan registers
  if ((long)N * KK * 2 >= 0x7fffffffL) return 0;
  const size_t lds = (size_t)nxi * 1024 + (size_t)nks * MT * 1024 + (size_t)kCtsWaves * 2 * 16 * MT * sizeof(float);
  return lds <= 160 * 1024 ? lds : 0;
}

template <int MT, int CPT, bool FWD, int NT>
static int launch_conv_t_stream_cfg(const CtsArgs& a0, size_t lds, int tiles_n, hipStream_t s) {
  CtsArgs a = a0;
  const int target = 256;
  const int tpb = cdiv(a.tiles_m * tiles_n, target) < 1 ? 1 : cdiv(a.tiles_m * tiles_n, target);
  a.tpb = tpb;
  a.nblk_m = cdiv(a.tiles_m, tpb);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_t_stream_kernel<MT, CPT, FWD, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  conv_t_stream_kernel<MT, CPT, FWD, NT><<<a.nblk_m * tiles_n, kCtsWaves * 64, lds, s>>>(a);
  EMB_CHECK_LAUNCH();
  return a.nblk_m;
}

// returns the number of partial rows (> 0) when the streaming kernel ran, 0 when the shape does not qualify, < 0 on error
template <int MT, bool FWD, int NT>
static int launch_conv_t_stream_nt(const void* x, const void* w, const void* bias, void* out, void* partial, int B, int L, int cin, int KK, int N,
                                   int pad, hipStream_t s) {
  constexpr int BT = kCtsWaves * 16 * NT;
  const size_t lds = conv_t_stream_lds<MT>(B, L, cin, KK, N, pad, BT);
  if (lds == 0 || !aligned16(x) || !aligned16(w) || !aligned16(out)) return 0;
  const ConvTiling t = conv_tiling_bt(B, L, pad, BT);
  CtsArgs a{};
  a.x = (const __bf16*)x; a.w = (const __bf16*)w; a.bias = (const float*)bias; a.out = (__bf16*)out; a.partial = (float*)partial;
  a.B = B; a.L = L; a.cin = cin; a.KK = KK; a.N = N; a.pad = pad; a.SB = t.SB; a.slot = t.slot; a.tiles_m = t.tiles_m; a.taps = KK / cin;
  const int tiles_n = cdiv(N, 16 * MT);
  switch (cin / 32) {
    case 1: return launch_conv_t_stream_cfg<MT, 1, FWD, NT>(a, lds, tiles_n, s);
    case 2: return launch_conv_t_stream_cfg<MT, 2, FWD, NT>(a, lds, tiles_n, s);
    case 4: return launch_conv_t_stream_cfg<MT, 4, FWD, NT>(a, lds, tiles_n, s);
  }
  return 0;
}

// returns the number of partial rows (> 0) when the streaming kernel ran, 0 when the shape does not qualify, < 0 on error
template <int MT, bool FWD>
static int launch_conv_t_stream(const void* x, const void* w, const void* bias, void* out, void* partial, int B, int L, int cin, int KK, int N,
                                int pad, hipStream_t s) {
  const int rc = launch_conv_t_stream_nt<MT, FWD, 4>(x, w, bias, out, partial, B, L, cin, KK, N, pad, s);
  if (rc != 0) return rc;
  return launch_conv_t_stream_nt<MT, FWD, 2>(x, w, bias, out, partial, B, L, cin, KK, N, pad, s);   // 256-row tiles: half the image
}

}  // namespace emb
