// Device-side building blocks shared by the direct convolution kernels (conv_direct.hip, conv_first.hip):
// halo-padded activation tiles in LDS, their register-staged prefetch plan, the channel <-> MFMA-row map of the
// transposed (channels on M) kernels.
#pragma once
#include "conv_direct.h"

namespace emb {

template <typename T> struct DCfg;
template <> struct DCfg<__bf16> { static constexpr int WCH = 128, XPAD = 8, WPAD = 8, BNW = 256; };
template <> struct DCfg<float> { static constexpr int WCH = 64, XPAD = 2, WPAD = 2, BNW = 128; };
template <> struct DCfg<double> { static constexpr int WCH = 32, XPAD = 2, WPAD = 2, BNW = 128; };

constexpr int kXExtra = 8;   // zero rows after the last slot (taps of the zero-padded k*cin tail land there)

template <typename T> __device__ __forceinline__ void lds_store_vec(T* dst, const typename Vec16<T>::type& v) {
  constexpr int VEC = Elem<T>::VEC;
  if (sizeof(T) == 2) {
    *reinterpret_cast<typename Vec16<T>::type*>(dst) = v;   // pitch is a multiple of 8 elements: 16-byte aligned
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e) dst[e] = v[e];           // pitch cin+2: element stores, conflict-free
  }
}

// activation rows of `SB` sequences starting at b0, times [t0 - pad, t0 - pad + slot) each, zero outside [0, L)
template <typename T, int NTHR = kThreads>
__device__ __forceinline__ void stage_x_tile(const T* __restrict__ x, T* xs, int XS, int xrows, int SB, int slot, int b0, int t0,
                                             int B, int L, int cin, int pad, int first = 0) {
  constexpr int VEC = Elem<T>::VEC;
  using V = typename Vec16<T>::type;
  const int cvn = cin / VEC;
  for (int i = first + threadIdx.x; i < xrows * cvn; i += NTHR) {
    const int row = i / cvn, cv = (i - row * cvn) * VEC;
    const int s = row / slot, tt = t0 - pad + (row - s * slot);
    V v;
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = (T)0.0f;
    if (s < SB && b0 + s < B && tt >= 0 && tt < L) v = *reinterpret_cast<const V*>(x + ((long)(b0 + s) * L + tt) * cin + cv);
    lds_store_vec<T>(xs + (long)row * XS + cv, v);
  }
}

// LDS row (relative to the tile) that output row r reads for tap 0
__device__ __forceinline__ int tile_xrow(int r, int L, int SB, int slot) {
  if (L >= kConvBT) return r;
  const int rr = min(r, SB * L - 1);
  const int s = rr / L;
  return s * slot + (rr - s * L);
}

// ---- register-staged tiles --------------------------------------------------------------------------------------
// A thread's share of an activation tile as a tile-independent plan (LDS slot, global offset relative to the
// tile origin, validity inputs) so the loads of tile i+1 can be issued -- into registers -- before the MFMA loop
// of tile i and unpacked into LDS after it: one exposed memory round trip per workgroup instead of one per tile.
constexpr int kXV = 8;   // activation-tile vectors per thread held in registers (the rest is staged directly)

template <typename T> struct XPlan {
  int lds[kXV];    // LDS element offset
  int glb[kXV];    // global element offset relative to x + (b0*L + t0)*cin
  int pk[kXV];     // (sequence slot << 16) | (row in slot);  -1: not this thread's
  int live;        // slots any thread of the workgroup uses (uniform)
};

template <typename T, int NTHR = kThreads>
__device__ __forceinline__ void xplan_init(XPlan<T>& p, int XS, int xrows, int SB, int slot, int L, int cin, int pad) {
  constexpr int VEC = Elem<T>::VEC;
  const int cvn = cin / VEC, nxv = xrows * cvn;
  p.live = min(kXV, (nxv + NTHR - 1) / NTHR);
#pragma unroll
  for (int i = 0; i < kXV; ++i) {
    const int idx = threadIdx.x + i * NTHR;
    const int row = idx / cvn, cv = (idx - row * cvn) * VEC, s = row / slot, dtp = row - s * slot;
    p.lds[i] = row * XS + cv;
    p.glb[i] = (s * L + dtp - pad) * cin + cv;
    p.pk[i] = idx < nxv ? (((s < SB ? s : 0x7fff) << 16) | dtp) : -1;
  }
}

template <typename T>
__device__ __forceinline__ void xplan_issue(const XPlan<T>& p, typename Vec16<T>::type (&v)[kXV], const T* __restrict__ x, int b0, int t0,
                                            int B, int L, int cin, int pad) {
  using V = typename Vec16<T>::type;
  const T* xb = x + ((long)b0 * L + t0) * cin;
#pragma unroll
  for (int i = 0; i < kXV; ++i) {
    if (i >= p.live) break;
    V val;
#pragma unroll
    for (int e = 0; e < Elem<T>::VEC; ++e) val[e] = (T)0.0f;
    if (p.pk[i] >= 0) {
      const int s = p.pk[i] >> 16, tt = t0 - pad + (p.pk[i] & 0xffff);
      if (b0 + s < B && tt >= 0 && tt < L) val = *reinterpret_cast<const V*>(xb + p.glb[i]);
    }
    v[i] = val;
  }
}

template <typename T>
__device__ __forceinline__ void xplan_commit(const XPlan<T>& p, const typename Vec16<T>::type (&v)[kXV], T* xs) {
#pragma unroll
  for (int i = 0; i < kXV; ++i)
    if (i < p.live && p.pk[i] >= 0) lds_store_vec<T>(xs + p.lds[i], v[i]);
}

template <typename T> struct AccMap {   // accumulator register r of lane-group g  <->  MFMA row index m
  __device__ static int g_of(int m) { return sizeof(T) == 8 ? (m & 3) : (m >> 2); }
  __device__ static int r_of(int m) { return sizeof(T) == 8 ? (m >> 2) : (m & 3); }
};
template <typename T, int MT> __device__ __forceinline__ int chan_of(int mt, int m) {
  return AccMap<T>::g_of(m) * (4 * MT) + mt * 4 + AccMap<T>::r_of(m);
}
constexpr int kWRegSteps = 4;   // weight k-steps a wave can keep in registers

__host__ __device__ inline int conv_t_xpitch(int cin, int elem_bytes) {   // conflict-free 16-byte row reads
  return elem_bytes == 2 ? ((cin % 16 == 0) ? cin + 8 : cin) : cin + 16 / elem_bytes;
}

// sum over the 16 lanes of a DPP row (every lane of the row receives it); f64 goes through the LDS crossbar
template <typename A> __device__ __forceinline__ A row16_sum(A v) {
  if constexpr (sizeof(A) == 4) {
    int x = __builtin_bit_cast(int, v);
#define EMB_DPP_ADD(ctrl) v += __builtin_bit_cast(A, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
    (void)x;
    EMB_DPP_ADD(0xB1);    // quad_perm [1,0,3,2]
    EMB_DPP_ADD(0x4E);    // quad_perm [2,3,0,1]
    EMB_DPP_ADD(0x124);   // row_ror:4
    EMB_DPP_ADD(0x128);   // row_ror:8
#undef EMB_DPP_ADD
    return v;
  } else {
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m);
    return v;
  }
}


inline ConvTiling conv_tiling_bt(int B, int L, int pad, int BT) {
  ConvTiling t;
  if (L >= BT) {
    t.SB = 1;
    t.tiles_t = (L + BT - 1) / BT;
    t.slot = BT + 2 * pad;
  } else {
    t.SB = BT / L;
    t.tiles_t = 1;
    t.slot = L + 2 * pad;
  }
  t.tiles_m = ((B + t.SB - 1) / t.SB) * t.tiles_t;
  return t;
}

}  // namespace emb
