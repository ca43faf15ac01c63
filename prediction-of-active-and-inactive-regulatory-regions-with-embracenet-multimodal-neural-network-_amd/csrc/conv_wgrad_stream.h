// Weight gradient of a stored-activation convolution block (CNN_pre.py:37-50, blocks 2..4), bf16, 64 input channels,
// Cout in {32, 64}, sequences of at most 128 positions:  slab[slice][o][n] = sum over the slice's rows of dy[r][o] * x[r + tap(n)][ci(n)]
// (n = tap * 64 + ci; column KK = sum of dy = bias gradient).  Included by conv_direct.hip.
//
// conv_wgrad_direct_kernel spends 2.9 us per 128-row tile on 0.3 us of matrix work: the tile travels global -> registers -> LDS
// behind two barriers, and with four waves per workgroup nothing overlaps it.  Here the tiles STREAM:
//   * a workgroup = 8 waves owns one 256-column block of the k*64 weight-gradient columns and 1/S of the row tiles; it keeps
//     four tile buffers in LDS and every wave issues its share of a tile as LDS-DMA (buffer_load ... lds, split_core.h) three
//     tiles ahead: no VGPR staging, no address arithmetic per tile (the per-lane source offsets are tile independent, the
//     buffer resource base moves), one raw barrier per tile;
//   * images are unpadded 128-byte rows (x: 64 channels; dy: 64 channels, or 32 channels in 64-byte rows) with the 16-byte
//     slots permuted by the row (swz16 / bit 3 of the row) so that the transposing fragment reads of a 32-lane half touch every
//     bank once -- the padded pitches of the old kernel conflicted (PMC: more conflict cycles than LDS instruction cycles);
//   * the 8 waves are 2 row halves x 4 column groups: waves 0-3 take rows 0-63 of every tile, waves 4-7 rows 64-127, each into
//     its own accumulators; the two halves meet in LDS once, at the end;
//   * 256 workgroups (one per CU) instead of 512: half the slices, so half the slab bytes written here and read by the
//     optimizer launch.
#pragma once
#include "conv_tiles.h"
#include "split_core.h"

namespace emb {

constexpr int kWsThreads = 512, kWsBufs = 4, kWsXRows = 192;     // x image: 24 LDS-DMA instructions of 8 rows (3 per wave)
constexpr int kWsXBytes = kWsXRows * 128, kWsDyBytesMax = 128 * 128;

struct WsArgs {
  const __bf16* dy;     // [B][L][Cout]
  const __bf16* x;      // [B][L][64]
  float* slab;          // [S][Cout][KK + 1]
  int B, L, KK, Cout, pad, SB, slot, tiles_m, n_tiles, S;
};

// fragment of a K-major image at two independent addresses (rows r and r + 4 need not be 512 bytes apart: sequence slots)
__device__ __forceinline__ bf16x8 ws_frag2(uint32_t lo_addr, uint32_t hi_addr) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
  u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)lo_addr);
  u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)hi_addr);
  return u.v;
}

// one LDS-DMA instruction (64 lanes x 16 bytes -> 1 KiB at LDS byte address `lds`, which must be wave-uniform) through a buffer
// resource {origin, valid bytes}: lanes whose `voff` is outside [0, bytes) write zeros
__device__ __forceinline__ void dma16_opaque(const void* origin, long bytes, uint32_t lds, uint32_t voff) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef int i32x4_t __attribute__((ext_vector_type(4)));
  const uint64_t p = (uint64_t)(uintptr_t)origin;
  i32x4_t rs;
  rs[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)p);
  rs[1] = __builtin_amdgcn_readfirstlane((int)((p >> 32) & 0xffffu));
  rs[2] = __builtin_amdgcn_readfirstlane((int)(bytes < 0x7fffffffL ? bytes : 0x7fffffffL));
  rs[3] = 0x00020000;
  const int m = __builtin_amdgcn_readfirstlane((int)lds);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(m), "v"(voff), "s"(rs) : "memory");   // (m0 is reserved: the compiler sets it before each of its own uses)
#endif
}

template <int MIW>   // Cout = 16 * MIW: 2 (dy rows of 64 bytes) or 4 (128 bytes)
__device__ __forceinline__ void conv_wgrad_stream_body(const WsArgs& a, const int block) {
  constexpr int DYROW = 32 * MIW;                          // bytes per dy row
  constexpr int DYB = 128 * DYROW;                         // dy image: 128 rows
  constexpr int NDY = DYB / 1024 / 8;                      // dy LDS-DMA instructions per wave and tile (1 or 2)
  constexpr int G = 3 + NDY;                               // LDS-DMA instructions per wave and tile
  constexpr int BUF = kWsXBytes + DYB;
  extern __shared__ __attribute__((aligned(16))) char arena[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)arena;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), kg = wave >> 2, wn = wave & 3;
  const int g = lane >> 4, r16 = lane & 15, q = r16 >> 2, p = r16 & 3;
  // workgroups are dealt round-robin to the 8 XCDs: the n_tiles workgroups of one slice read the same tiles, so they sit on ONE
  // XCD (its L2 fetches the tile once) when the slices divide evenly
  int nt, slice;
  if (a.S % 8 == 0) {
    const int xcd = block & 7, j = block >> 3;
    nt = j % a.n_tiles;
    slice = (j / a.n_tiles) * 8 + xcd;
  } else {
    nt = block % a.n_tiles;
    slice = block / a.n_tiles;
  }
  const int L = a.L, SB = a.SB, slot = a.slot, n0 = nt * 256;
  const int per = (a.tiles_m + a.S - 1) / a.S, tm_begin = slice * per, tm_end = min(a.tiles_m, tm_begin + per);
  const int ntile = tm_end - tm_begin;

  // ---- LDS-DMA plan (tile independent).  x image row i = (sequence slot sq, position dtp) holds x[b0 + sq][dtp - pad] or zeros
  uint32_t xv[3], dv[NDY];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int row = 8 * (wave + 8 * i) + (lane >> 3), ps = lane & 7;
    const int sq = row / slot, tt = row - sq * slot - a.pad;
    const bool ok = sq < SB && tt >= 0 && tt < L;
    xv[i] = ok ? (uint32_t)((sq * L + tt) * 128 + 16 * (ps ^ swz16(row))) : kDmaInvalid;
  }
#pragma unroll
  for (int i = 0; i < NDY; ++i) {
    if (MIW == 2) {                                        // 64-byte rows, 16 per instruction; slot ^ 2 on rows with bit 3
      const int row = 16 * wave + (lane >> 2), ps = lane & 3;
      dv[i] = row < SB * L ? (uint32_t)(row * 64 + 16 * (ps ^ (((row >> 3) & 1) << 1))) : kDmaInvalid;
    } else {
      const int row = 8 * (wave + 8 * i) + (lane >> 3), ps = lane & 7;
      dv[i] = row < SB * L ? (uint32_t)(row * 128 + 16 * (ps ^ swz16(row))) : kDmaInvalid;
    }
  }
  // The LDS-DMA instructions are issued from inline assembly: the compiler orders every LDS read behind ALL outstanding
  // "buffer_load ... lds" it knows of (s_waitcnt vmcnt(0)), which would serialise the tiles in flight; the ordering that is
  // needed is the counted wait + barrier below.
  auto issue = [&](int tm, int buf) {
    const long b0 = (long)tm * SB;
    const long xleft = ((long)a.B - b0) * L * 128, dleft = ((long)a.B - b0) * L * DYROW;
    const uint32_t base = lds0 + (uint32_t)buf * BUF + (uint32_t)wave * 1024u;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      dma16_opaque(a.x + b0 * L * 64, xleft, base + i * 8192u, xv[i]);
#pragma unroll
    for (int i = 0; i < NDY; ++i)
      dma16_opaque(a.dy + b0 * L * (16 * MIW), dleft, base + kWsXBytes + i * 8192u, dv[i]);
  };
  if (ntile > 0) issue(tm_begin, 0);
  if (ntile > 1) issue(tm_begin + 1, 1);
  if (ntile > 2) issue(tm_begin + 2, 2);

  // ---- fragment addresses (tile independent).  This wave's k-steps: rows 64 kg + 32 ks2 + 8 g + q (+ 4)
  uint32_t aoff[2][2][MIW], boff[2][4][2];
#pragma unroll
  for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = 64 * kg + 32 * ks2 + 8 * g + q + 4 * h;           // tile row = dy image row
#pragma unroll
      for (int mi = 0; mi < MIW; ++mi) {
        const int sl = 2 * mi + (p >> 1);                              // 16-byte slot of the lane's four channels
        aoff[ks2][h][mi] = (uint32_t)(kWsXBytes + r * DYROW + ((MIW == 2 ? (sl ^ (((r >> 3) & 1) << 1)) : (sl ^ swz16(r))) << 4) + (p & 1) * 8);
      }
      const int rr = min(r, SB * L - 1), sq = rr / L, xr0 = sq * slot + (rr - sq * L);   // x image row of tap 0
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int n = n0 + (ni * 4 + wn) * 16 + 4 * p, tap = n >> 6, ci = n & 63;
        const int xr = min(xr0 + tap, kWsXRows - 1);                   // (columns past KK: any row, the products are never stored)
        boff[ks2][ni][h] = (uint32_t)(xr * 128 + (((ci >> 3) ^ swz16(xr)) << 4) + (ci & 7) * 2);
      }
    }

  f32x4 acc[MIW][4], bias_acc[MIW];
#pragma unroll
  for (int mi = 0; mi < MIW; ++mi) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bias_acc[mi][r] = 0.0f;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.0f;
  }
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  const bool do_bias = nt == 0 && wn == 0;                 // column KK: dy^T x ones on the matrix cores

  // Software pipeline: the fragments of the next k-step (or of the next tile's first k-step) are read while the matrix cores
  // work on the current one; the tile barrier sits between the two k-steps of a tile.
  struct Step { bf16x8 af[MIW], bf[4]; };
  auto load0 = [&](uint32_t base, Step& f) {
#pragma unroll
    for (int mi = 0; mi < MIW; ++mi) f.af[mi] = ws_frag2(base + aoff[0][0][mi], base + aoff[0][1][mi]);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) f.bf[ni] = ws_frag2(base + boff[0][ni][0], base + boff[0][ni][1]);
  };
  auto load1 = [&](uint32_t base, Step& f) {
#pragma unroll
    for (int mi = 0; mi < MIW; ++mi) f.af[mi] = ws_frag2(base + aoff[1][0][mi], base + aoff[1][1][mi]);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) f.bf[ni] = ws_frag2(base + boff[1][ni][0], base + boff[1][ni][1]);
  };
  auto mma = [&](const Step& f) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)                         // (column blocks past KK multiply clamped rows; never stored)
#pragma unroll
      for (int mi = 0; mi < MIW; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.af[mi], f.bf[ni], acc[mi][ni], 0, 0, 0);
    if (do_bias) {
#pragma unroll
      for (int mi = 0; mi < MIW; ++mi) bias_acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.af[mi], ones, bias_acc[mi], 0, 0, 0);
    }
  };
  Step f0, f1;
  if (ntile > 0) {
    wait_chunks_in_flight<G>(min(2, ntile - 1));           // this wave's part of tile 0 has landed
    __builtin_amdgcn_s_barrier();                          // ... everyone's
    asm volatile("" ::: "memory");
    load0(lds0, f0);
    if (3 < ntile) issue(tm_begin + 3, 3);
  }
  auto pin_f0 = [&]() {
    if (MIW == 2) asm volatile("" : : "v"(f0.af[0]), "v"(f0.af[1]), "v"(f0.bf[0]), "v"(f0.bf[1]), "v"(f0.bf[2]), "v"(f0.bf[3]));
    else asm volatile("" : : "v"(f0.af[0]), "v"(f0.af[1]), "v"(f0.af[MIW - 2]), "v"(f0.af[MIW - 1]), "v"(f0.bf[0]), "v"(f0.bf[1]), "v"(f0.bf[2]), "v"(f0.bf[3]));
  };
  auto pin_f1 = [&]() {
    if (MIW == 2) asm volatile("" : : "v"(f1.af[0]), "v"(f1.af[1]), "v"(f1.bf[0]), "v"(f1.bf[1]), "v"(f1.bf[2]), "v"(f1.bf[3]));
    else asm volatile("" : : "v"(f1.af[0]), "v"(f1.af[1]), "v"(f1.af[MIW - 2]), "v"(f1.af[MIW - 1]), "v"(f1.bf[0]), "v"(f1.bf[1]), "v"(f1.bf[2]), "v"(f1.bf[3]));
  };
  pin_f0();
  for (int i = 0; i < ntile; ++i) {
    load1(lds0 + (uint32_t)(i & 3) * BUF, f1);
    __builtin_amdgcn_sched_barrier(0);
    mma(f0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's reads of tile i are complete (f1 has arrived) ...
    pin_f1();                                              // ... which the compiler's wait-count model has to know on BOTH paths below
    if (i + 1 < ntile) {
      wait_chunks_in_flight<G>(min(2, ntile - 2 - i));     // this wave's part of tile i + 1 has landed
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      load0(lds0 + (uint32_t)((i + 1) & 3) * BUF, f0);
      if (i + 4 < ntile) issue(tm_begin + i + 4, i & 3);   // tile i's buffer is free
    }
    __builtin_amdgcn_sched_barrier(0);
    mma(f1);
    __builtin_amdgcn_sched_barrier(0);
    // the compiler's wait-count model gives up on LDS reads that are pending across the loop edge (it would wait for
    // EVERYTHING before the first MFMA of the next iteration, the reads just issued included): pin the arrival of f0 here,
    // ten MFMAs after its reads were issued
    pin_f0();
  }

  // ---- the two row halves meet: waves 4-7 park their accumulators in LDS, waves 0-3 add and store
  __syncthreads();                                         // all tiles consumed (and no LDS-DMA outstanding)
  float* park = reinterpret_cast<float*>(arena) + (size_t)(wn * 64 + lane) * (MIW * 20 + 1);   // odd pitch: conflict-free
  if (kg == 1) {
#pragma unroll
    for (int mi = 0; mi < MIW; ++mi) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) park[(mi * 5 + ni) * 4 + r] = acc[mi][ni][r];
#pragma unroll
      for (int r = 0; r < 4; ++r) park[(mi * 5 + 4) * 4 + r] = bias_acc[mi][r];
    }
  }
  __syncthreads();
  if (kg == 0) {
    float* dst = a.slab + (long)slice * a.Cout * (a.KK + 1);
#pragma unroll
    for (int mi = 0; mi < MIW; ++mi) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = mi * 16 + 4 * g + r, n = n0 + (ni * 4 + wn) * 16 + r16;
          if (n < a.KK) dst[(long)o * (a.KK + 1) + n] = acc[mi][ni][r] + park[(mi * 5 + ni) * 4 + r];
        }
      if (do_bias && r16 == 0) {                           // every column of dy^T x ones holds the sum; lane column 0 writes it
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(long)(mi * 16 + 4 * g + r) * (a.KK + 1) + a.KK] = bias_acc[mi][r] + park[(mi * 5 + 4) * 4 + r];
      }
    }
  }
}

template <int MIW>
__global__ __launch_bounds__(kWsThreads, 2) void conv_wgrad_stream_kernel(const WsArgs a) {
  conv_wgrad_stream_body<MIW>(a, (int)blockIdx.x);
}

// shapes the streaming kernel takes (pointer alignment is checked at launch)
inline bool conv_wgrad_stream_shape_ok(int B, int L, int cin, int KK, int Cout, int pad) {
  if (cin != 64 || (Cout != 32 && Cout != 64) || L > kConvBT || L < 1 || KK % 64 != 0 || KK < 64) return false;
  const ConvTiling t = conv_tiling(B, L, pad);
  return t.tiles_t == 1 && t.SB * t.slot <= kWsXRows;
}
inline int conv_wgrad_stream_slices(int B, int L, int KK, int pad) {
  const ConvTiling t = conv_tiling(B, L, pad);
  const int n_tiles = cdiv(KK, 256);
  int S = 256 / n_tiles;                                   // one 8-wave workgroup per CU
  if (S > t.tiles_m) S = t.tiles_m;
  return S < 1 ? 1 : S;
}

static size_t wgrad_stream_fill(WsArgs& a, const void* dy, const void* x, void* slab, int B, int L, int KK, int Cout, int pad, int S) {
  const ConvTiling t = conv_tiling(B, L, pad);
  a.dy = (const __bf16*)dy; a.x = (const __bf16*)x; a.slab = (float*)slab;
  a.B = B; a.L = L; a.KK = KK; a.Cout = Cout; a.pad = pad; a.SB = t.SB; a.slot = t.slot; a.tiles_m = t.tiles_m;
  a.n_tiles = cdiv(KK, 256); a.S = S;
  return (size_t)kWsBufs * (kWsXBytes + (Cout == 32 ? 128 * 64 : 128 * 128));   // dynamic LDS bytes
}

static int launch_wgrad_stream(const void* dy, const void* x, void* slab, int B, int L, int KK, int Cout, int pad, int S, hipStream_t s) {
  WsArgs a{};
  const size_t lds = wgrad_stream_fill(a, dy, x, slab, B, L, KK, Cout, pad, S);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_stream_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_stream_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  if (Cout == 32) conv_wgrad_stream_kernel<2><<<a.n_tiles * S, kWsThreads, lds, s>>>(a);
  else conv_wgrad_stream_kernel<4><<<a.n_tiles * S, kWsThreads, lds, s>>>(a);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

}  // namespace emb
