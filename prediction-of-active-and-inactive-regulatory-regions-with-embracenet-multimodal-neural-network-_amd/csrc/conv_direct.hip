// Direct 1-D convolution kernels (see conv_direct.h).  Reference op: nn.Conv1d of CNN_pre.py:37-38.
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
template <typename T>
__device__ __forceinline__ void stage_x_tile(const T* __restrict__ x, T* xs, int XS, int xrows, int SB, int slot, int b0, int t0,
                                             int B, int L, int cin, int pad) {
  constexpr int VEC = Elem<T>::VEC;
  using V = typename Vec16<T>::type;
  const int cvn = cin / VEC;
  for (int i = threadIdx.x; i < xrows * cvn; i += kThreads) {
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

template <typename T, int BN, bool FWD>
__global__ __launch_bounds__(kThreads) void conv_direct_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                               const typename AccOf<T>::type* __restrict__ bias, T* __restrict__ out,
                                                               typename AccOf<T>::type* __restrict__ partial, int B, int L, int cin,
                                                               int KK, int N, int pad, int SB, int tiles_t, int slot, int tiles_n) {
  using Mm = Mma<T>;
  using Acc = typename Mm::Acc;
  using V = typename Vec16<T>::type;
  constexpr int VEC = Elem<T>::VEC, KSTEP = Mm::KSTEP, WCH = DCfg<T>::WCH, BT = kConvBT, MI = 2, NI = BN / 16;
  constexpr int WS = WCH + DCfg<T>::WPAD, CS = BN + 4;
  constexpr int WV = BN * WCH / VEC / kThreads;   // weight vectors per thread and chunk
  constexpr bool BF = sizeof(T) == 2;
  extern __shared__ __attribute__((aligned(16))) char arena[];
  const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
  const int b0 = (tm / tiles_t) * SB, t0 = (tm % tiles_t) * BT, col0 = tn * BN;
  const int XS = cin + DCfg<T>::XPAD, xrows = SB * slot + kXExtra;
  T* xs = reinterpret_cast<T*>(arena);
  T* ws0 = xs + (((long)xrows * XS + 7) & ~7L);
  T* ws1 = ws0 + BN * WS;
  const int KKp = (KK + KSTEP - 1) / KSTEP * KSTEP, nch = (KKp + WCH - 1) / WCH;

  stage_x_tile<T>(x, xs, XS, xrows, SB, slot, b0, t0, B, L, cin, pad);
  V wreg[WV];
  auto loadw = [&](int ch) {
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int v = threadIdx.x + i * kThreads;
      const int n = v / (WCH / VEC), kc = (v % (WCH / VEC)) * VEC, kk = ch * WCH + kc;
      V val;
#pragma unroll
      for (int e = 0; e < VEC; ++e) val[e] = (T)0.0f;
      if (col0 + n < N && kk < KK) val = *reinterpret_cast<const V*>(w + (long)(col0 + n) * KK + kk);   // KK % VEC == 0
      wreg[i] = val;
    }
  };
  auto storew = [&](T* dst) {
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int v = threadIdx.x + i * kThreads;
      const int n = v / (WCH / VEC), kc = (v % (WCH / VEC)) * VEC;
      lds_store_vec<T>(dst + n * WS + kc, wreg[i]);
    }
  };
  loadw(0);
  storew(ws0);
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r16 = lane & 15;
  int xrow[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) xrow[mi] = tile_xrow(wave * 32 + mi * 16 + r16, L, SB, slot);
  typename Mm::AccV acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0;

  for (int ch = 0; ch < nch; ++ch) {
    const bool more = ch + 1 < nch;
    if (more) loadw(ch + 1);
    const T* wt = (ch & 1) ? ws1 : ws0;
#pragma unroll
    for (int ks = 0; ks < WCH / KSTEP; ++ks) {
      const int kk0 = ch * WCH + ks * KSTEP;
      if (kk0 < KKp) {
        const int kl = BF ? 8 * g : g;                 // this lane's k offset inside the MFMA k-step
        const int tap = (kk0 + kl) / cin, ci = (kk0 + kl) - tap * cin;
        typename Mm::Frag af[MI], bf[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          af[mi] = *reinterpret_cast<const typename Mm::Frag*>(xs + (long)(xrow[mi] + tap) * XS + ci);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          bf[ni] = *reinterpret_cast<const typename Mm::Frag*>(wt + (ni * 16 + r16) * WS + ks * KSTEP + kl);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = Mm::mma(af[mi], bf[ni], acc[mi][ni]);
      }
    }
    if (more) storew((ch & 1) ? ws0 : ws1);
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS slab -> coalesced row-major stores (+ BatchNorm partial sums)
  Acc* cs = reinterpret_cast<Acc*>(arena);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        cs[(wave * 32 + mi * 16 + Mm::acc_row(lane, r)) * CS + ni * 16 + r16] = acc[mi][ni][r];
  __syncthreads();
  const bool multi = L < BT;
  for (int gidx = threadIdx.x; gidx < BT * BN / 4; gidx += kThreads) {
    const int r = gidx / (BN / 4), cq = (gidx % (BN / 4)) * 4;
    const int s = multi ? r / L : 0, tl = multi ? r - s * L : r;
    const bool rv = (multi ? r < SB * L : true) && (b0 + s < B) && (t0 + tl < L);
    const int col = col0 + cq;
    Acc v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = cs[r * CS + cq + j];
      if (FWD) {
        v[j] += bias[min(col + j, N - 1)];
        cs[r * CS + cq + j] = (rv && col + j < N) ? v[j] : (Acc)0;
      }
    }
    if (!rv || col >= N) continue;
    T* dst = out + ((long)(b0 + s) * L + t0 + tl) * N + col;
    if (col + 4 <= N && (N & 3) == 0) {
      typedef T TV4 __attribute__((ext_vector_type(4)));
      TV4 o = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
      *reinterpret_cast<TV4*>(dst) = o;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (col + j < N) dst[j] = (T)v[j];
    }
  }
  if (FWD) {
    __syncthreads();
    constexpr int PARTS = kThreads / BN;
    Acc* red = cs + BT * CS;
    const int colr = threadIdx.x % BN, part = threadIdx.x / BN;
    Acc s1 = 0, s2 = 0;
    for (int r = part; r < BT; r += PARTS) {
      const Acc v = cs[r * CS + colr];
      s1 += v;
      s2 += v * v;
    }
    red[part * BN + colr] = s1;
    red[(PARTS + part) * BN + colr] = s2;
    __syncthreads();
    if (part == 0 && col0 + colr < N) {
      Acc a = 0, b = 0;
      for (int p = 0; p < PARTS; ++p) {
        a += red[p * BN + colr];
        b += red[(PARTS + p) * BN + colr];
      }
      partial[((long)tm * 2 + 0) * N + col0 + colr] = a;
      partial[((long)tm * 2 + 1) * N + col0 + colr] = b;
    }
  }
}

template <typename T, int BN> static size_t conv_direct_lds(int cin, int SB, int slot) {
  using Acc = typename AccOf<T>::type;
  const size_t xs = ((size_t)(SB * slot + kXExtra) * (cin + DCfg<T>::XPAD) + 7) & ~(size_t)7;
  const size_t op = (xs + 2 * (size_t)BN * (DCfg<T>::WCH + DCfg<T>::WPAD)) * sizeof(T);
  const size_t ep = ((size_t)kConvBT * (BN + 4) + 2 * kThreads) * sizeof(Acc);
  return ((op > ep ? op : ep) + 15) & ~(size_t)15;
}

constexpr size_t kMaxDirectLds = 150 * 1024;

template <typename T, int BN, bool FWD>
static int launch_direct(const void* x, const void* w, const void* bias, void* out, void* partial, int B, int L, int cin, int KK, int N,
                         int pad, hipStream_t s) {
  using Acc = typename AccOf<T>::type;
  const ConvTiling t = conv_tiling(B, L, pad);
  const size_t lds = conv_direct_lds<T, BN>(cin, t.SB, t.slot);
  if (lds > kMaxDirectLds) return 1;   // caller falls back to the generic GEMM view
  static size_t attr = 0;
  if (lds > attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_direct_kernel<T, BN, FWD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDirectLds);
    attr = kMaxDirectLds;
  }
  const int tiles_n = cdiv(N, BN);
  conv_direct_kernel<T, BN, FWD><<<t.tiles_m * tiles_n, kThreads, lds, s>>>((const T*)x, (const T*)w, (const Acc*)bias, (T*)out,
                                                                        (Acc*)partial, B, L, cin, KK, N, pad, t.SB, t.tiles_t, t.slot, tiles_n);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

template <typename T> static int launch_direct_t(bool fwd, const void* x, const void* w, const void* bias, void* out, void* partial,
                                                 int B, int L, int cin, int KK, int N, int pad, hipStream_t s) {
  if (N >= 64) return fwd ? launch_direct<T, 64, true>(x, w, bias, out, partial, B, L, cin, KK, N, pad, s)
                          : launch_direct<T, 64, false>(x, w, bias, out, partial, B, L, cin, KK, N, pad, s);
  return fwd ? launch_direct<T, 32, true>(x, w, bias, out, partial, B, L, cin, KK, N, pad, s)
             : launch_direct<T, 32, false>(x, w, bias, out, partial, B, L, cin, KK, N, pad, s);
}

int launch_conv_direct(int dtype, bool fwd, const void* x, const void* w, const void* bias, void* out, void* partial, int B, int L,
                       int cin, int KK, int N, int pad, hipStream_t s) {
  switch (dtype) {
    case EMB_F32: return launch_direct_t<float>(fwd, x, w, bias, out, partial, B, L, cin, KK, N, pad, s);
    case EMB_BF16: return launch_direct_t<__bf16>(fwd, x, w, bias, out, partial, B, L, cin, KK, N, pad, s);
    case EMB_F64: return launch_direct_t<double>(fwd, x, w, bias, out, partial, B, L, cin, KK, N, pad, s);
  }
  return EMB_ERR_DTYPE;
}

// ------------------------------------------------------------------------------------ weight gradient
// slab[slice][o][n] += sum over the slice's row tiles of dy[r][o] * xview[r][n]  (n = tap*cin + ci);  column KK = sum dy
template <typename T, int BMW>
__global__ __launch_bounds__(kThreads) void conv_wgrad_direct_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                     typename AccOf<T>::type* __restrict__ slab, int B, int L, int cin,
                                                                     int KK, int Cout, int pad, int SB, int tiles_t, int slot,
                                                                     int tiles_m, int n_tiles, int m_tiles, int S) {
  using Mm = Mma<T>;
  using Acc = typename Mm::Acc;
  using V = typename Vec16<T>::type;
  constexpr int VEC = Elem<T>::VEC, KSTEP = Mm::KSTEP, BT = kConvBT, BNW = DCfg<T>::BNW;
  constexpr int MIW = BMW / 16, NIW = BNW / 64;        // n-blocks are dealt round-robin to the 4 waves
  constexpr int DS = BMW + (sizeof(T) == 2 ? 8 : 16);  // dy tile pitch (K-major image: [row][o])
  constexpr bool BF = sizeof(T) == 2;
  extern __shared__ __attribute__((aligned(16))) char arena[];
  const int nt = blockIdx.x % n_tiles, mt = (blockIdx.x / n_tiles) % m_tiles, slice = blockIdx.x / (n_tiles * m_tiles);
  const int o0 = mt * BMW, n0 = nt * BNW;
  const int XS = cin + DCfg<T>::XPAD, xrows = SB * slot + kXExtra;
  T* xs = reinterpret_cast<T*>(arena);
  T* dys = xs + (((long)xrows * XS + 7) & ~7L);
  int* rowmap = reinterpret_cast<int*>(dys + BT * DS);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r16 = lane & 15, q = r16 >> 2, p = r16 & 3;
  if (threadIdx.x < BT) rowmap[threadIdx.x] = tile_xrow(threadIdx.x, L, SB, slot);

  // per n-block LDS offset of (tap, ci): the B operand element (n, row r) is xs[(rowmap[r] + tap(n)) * XS + ci(n)]
  int xoff[NIW];
#pragma unroll
  for (int ni = 0; ni < NIW; ++ni) {
    const int n = n0 + (ni * 4 + wave) * 16 + (BF ? 4 * p : r16);
    const int tap = n / cin;
    xoff[ni] = tap * XS + (n - tap * cin);
  }
  typename Mm::AccV acc[MIW][NIW];
#pragma unroll
  for (int mi = 0; mi < MIW; ++mi)
#pragma unroll
    for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0;
  Acc bias_acc = 0;

  const int per = (tiles_m + S - 1) / S, tm_begin = slice * per, tm_end = min(tiles_m, tm_begin + per);
  for (int tm = tm_begin; tm < tm_end; ++tm) {
    const int b0 = (tm / tiles_t) * SB, t0 = (tm % tiles_t) * BT;
    __syncthreads();   // previous tile fully consumed
    stage_x_tile<T>(x, xs, XS, xrows, SB, slot, b0, t0, B, L, cin, pad);
    for (int i = threadIdx.x; i < BT * (BMW / VEC); i += kThreads) {   // dy rows of this tile, zero where invalid
      const int r = i / (BMW / VEC), cv = (i % (BMW / VEC)) * VEC;
      const bool multi = L < BT;
      const int s = multi ? r / L : 0, tl = multi ? r - s * L : r;
      const bool rv = (multi ? r < SB * L : true) && (b0 + s < B) && (t0 + tl < L);
      V v;
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] = (T)0.0f;
      if (rv && o0 + cv < Cout) v = *reinterpret_cast<const V*>(dy + ((long)(b0 + s) * L + t0 + tl) * Cout + o0 + cv);
      *reinterpret_cast<V*>(dys + r * DS + cv) = v;   // DS * sizeof(T) is a multiple of 16
    }
    __syncthreads();
    if (nt == 0 && threadIdx.x < BMW) {   // bias gradient: column sums of dy
      Acc a = 0;
      for (int r = 0; r < BT; ++r) a += (Acc)dys[r * DS + threadIdx.x];
      bias_acc += a;
    }
#pragma unroll 2
    for (int ks = 0; ks < BT / KSTEP; ++ks) {
      typename Mm::Frag af[MIW], bf[NIW];
      if (BF) {
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const int ra = ks * 32 + 8 * g + q;
#pragma unroll
        for (int mi = 0; mi < MIW; ++mi) {
          const T* a0 = dys + ra * DS + mi * 16 + 4 * p;
          union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
          u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
          u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * DS));
          af[mi] = *reinterpret_cast<typename Mm::Frag*>(&u.v);
        }
        const int x0 = rowmap[ra] * XS, x1 = rowmap[ra + 4] * XS;
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) {
          union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
          u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xs + x0 + xoff[ni]));
          u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xs + x1 + xoff[ni]));
          bf[ni] = *reinterpret_cast<typename Mm::Frag*>(&u.v);
        }
      } else {
        const int ra = ks * KSTEP + g;
#pragma unroll
        for (int mi = 0; mi < MIW; ++mi) af[mi] = *reinterpret_cast<const typename Mm::Frag*>(dys + ra * DS + mi * 16 + r16);
        const int x0 = rowmap[ra] * XS;
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) bf[ni] = *reinterpret_cast<const typename Mm::Frag*>(xs + x0 + xoff[ni]);
      }
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni)
        if (n0 + (ni * 4 + wave) * 16 < KK) {   // wave-uniform: n-blocks past the real columns do nothing
#pragma unroll
          for (int mi = 0; mi < MIW; ++mi) acc[mi][ni] = Mm::mma(af[mi], bf[ni], acc[mi][ni]);
        }
    }
  }
  Acc* dst = slab + (long)slice * Cout * (KK + 1);
#pragma unroll
  for (int mi = 0; mi < MIW; ++mi)
#pragma unroll
    for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = o0 + mi * 16 + Mm::acc_row(lane, r), n = n0 + (ni * 4 + wave) * 16 + r16;
        if (o < Cout && n < KK) dst[(long)o * (KK + 1) + n] = acc[mi][ni][r];
      }
  if (nt == 0 && threadIdx.x < BMW && o0 + threadIdx.x < Cout) dst[(long)(o0 + threadIdx.x) * (KK + 1) + KK] = bias_acc;
}

template <typename T, int BMW> static size_t wgrad_direct_lds(int cin, int SB, int slot) {
  constexpr int DS = BMW + (sizeof(T) == 2 ? 8 : 16);
  const size_t xs = ((size_t)(SB * slot + kXExtra) * (cin + DCfg<T>::XPAD) + 7) & ~(size_t)7;
  return (((xs + (size_t)kConvBT * DS) * sizeof(T) + kConvBT * sizeof(int)) + 15) & ~(size_t)15;
}

template <typename T> static int wgrad_slices_t(int B, int L, int pad, int KK, int Cout) {
  const ConvTiling t = conv_tiling(B, L, pad);
  const int bmw = Cout > 32 ? 64 : 32;
  const int wg = cdiv(KK, DCfg<T>::BNW) * cdiv(Cout, bmw);
  int S = 512 / (wg > 0 ? wg : 1);
  if (S > t.tiles_m) S = t.tiles_m;
  if (S < 1) S = 1;
  return S;
}

int conv_wgrad_slices(int B, int L, int pad, int KK, int Cout, int dtype) {
  switch (dtype) {
    case EMB_F32: return wgrad_slices_t<float>(B, L, pad, KK, Cout);
    case EMB_BF16: return wgrad_slices_t<__bf16>(B, L, pad, KK, Cout);
    default: return wgrad_slices_t<double>(B, L, pad, KK, Cout);
  }
}

template <typename T, int BMW>
static int launch_wgrad(const void* dy, const void* x, void* slab, int B, int L, int cin, int KK, int Cout, int pad, int S, hipStream_t s) {
  using Acc = typename AccOf<T>::type;
  const ConvTiling t = conv_tiling(B, L, pad);
  const size_t lds = wgrad_direct_lds<T, BMW>(cin, t.SB, t.slot);
  if (lds > kMaxDirectLds) return 1;
  static size_t attr = 0;
  if (lds > attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_direct_kernel<T, BMW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDirectLds);
    attr = kMaxDirectLds;
  }
  const int n_tiles = cdiv(KK, DCfg<T>::BNW), m_tiles = cdiv(Cout, BMW);
  conv_wgrad_direct_kernel<T, BMW><<<n_tiles * m_tiles * S, kThreads, lds, s>>>((const T*)dy, (const T*)x, (Acc*)slab, B, L, cin, KK, Cout,
                                                                               pad, t.SB, t.tiles_t, t.slot, t.tiles_m, n_tiles, m_tiles, S);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

template <typename T> static int launch_wgrad_t(const void* dy, const void* x, void* slab, int B, int L, int cin, int KK, int Cout,
                                                int pad, int S, hipStream_t s) {
  if (Cout > 32) return launch_wgrad<T, 64>(dy, x, slab, B, L, cin, KK, Cout, pad, S, s);
  return launch_wgrad<T, 32>(dy, x, slab, B, L, cin, KK, Cout, pad, S, s);
}

int launch_conv_wgrad_direct(int dtype, const void* dy, const void* x, void* slab, int B, int L, int cin, int KK, int Cout, int pad,
                             int S, hipStream_t s) {
  switch (dtype) {
    case EMB_F32: return launch_wgrad_t<float>(dy, x, slab, B, L, cin, KK, Cout, pad, S, s);
    case EMB_BF16: return launch_wgrad_t<__bf16>(dy, x, slab, B, L, cin, KK, Cout, pad, S, s);
    case EMB_F64: return launch_wgrad_t<double>(dy, x, slab, B, L, cin, KK, Cout, pad, S, s);
  }
  return EMB_ERR_DTYPE;
}

}  // namespace emb
