// Direct 1-D convolution kernels (see conv_direct.h).  Reference op: nn.Conv1d of CNN_pre.py:37-38.
#include "conv_tiles.h"
#include "conv_wgrad_stream.h"
#include <cstdlib>

namespace emb {

#ifdef EMB_CONV_PROF
__device__ unsigned long long g_conv_prof[64];
__device__ int g_conv_prof_sel;   // which kernel family records: 0 fwd/dgrad resident, 1 wgrad
#define CONV_T(fam, i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && g_conv_prof_sel == fam && (i) < 64) g_conv_prof[i] = wall_clock64(); } while (0)
__device__ int g_conv_dbg;        // experiment switches (bit set = skip that phase)
#define CONV_DBG(bit) ((g_conv_dbg >> (bit)) & 1)
#else
#define CONV_T(fam, i) do {} while (0)
#define CONV_DBG(bit) 0
#endif

// ---- transposed streaming variant -----------------------------------------------------------------------------------
// out^T = W . im2col(x)^T : the MFMA M dimension carries the output CHANNELS, N the output rows.  With the weight
// rows dealt to the MFMA row index so that lane-group g owns channels [g*4MT, (g+1)*4MT) of the tile, every lane
// ends up holding 4*MT CONSECUTIVE channels of one output row: the result leaves the accumulators as 16-byte
// row-major stores (a wave writes 16 whole rows per instruction pair) -- no LDS transpose, no barrier in the
// epilogue.  Each wave owns NT*16 rows; the weights sit in registers (k*cin <= 4 MFMA steps: the one-hot layer) or
// permuted in LDS (one ds_read_b128 per channel tile and k-step, reused by the wave's NT row tiles); the next
// tile's activations are in flight during the MFMA loop.  BatchNorm partial sums are kept per lane across ALL of
// the workgroup's tiles and meet once at the end (4 shuffle steps + 2 KiB of LDS): one partial row per workgroup.
template <typename T, int MT, int NT, int WAVES, bool FWD, bool WREG>
__global__ __launch_bounds__(WAVES * 64) void conv_t_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                          const typename AccOf<T>::type* __restrict__ bias, T* __restrict__ out,
                                                          typename AccOf<T>::type* __restrict__ partial, int B, int L, int cin, int KK,
                                                          int N, int pad, int SB, int tiles_t, int slot, int tiles_m, int tpb, int nblk_m) {
  using Mm = Mma<T>;
  using Acc = typename Mm::Acc;
  using V = typename Vec16<T>::type;
  constexpr int VEC = Elem<T>::VEC, KSTEP = Mm::KSTEP, BN = 16 * MT, BT = WAVES * 16 * NT, CPL = 4 * MT, NTHR = WAVES * 64;   // CPL: channels per lane
  constexpr bool BF = sizeof(T) == 2;
  extern __shared__ __attribute__((aligned(16))) char arena[];
  const int tn = blockIdx.x / nblk_m, bm = blockIdx.x % nblk_m, col0 = tn * BN;
  const int tm_begin = bm * tpb, tm_end = min(tiles_m, tm_begin + tpb);
  const int XS = conv_t_xpitch(cin, (int)sizeof(T)), xrows = SB * slot + kXExtra;
  const int KKp = (KK + KSTEP - 1) / KSTEP * KSTEP, WSR = KKp + DCfg<T>::WPAD, nks = KKp / KSTEP;
  T* xs = reinterpret_cast<T*>(arena);
  T* wsr = xs + (((long)xrows * XS + 7) & ~7L);
  Acc* red = reinterpret_cast<Acc*>(wsr + (WREG ? 0 : (((long)BN * WSR + 7) & ~7L)));
  const bool multi = L < BT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r16 = lane & 15;
  const int kl = BF ? 8 * g : g;                       // this lane's k offset inside an MFMA k-step

  CONV_T(0, 0);
  XPlan<T> xp;
  V xr[kXV];
  xplan_init<T, NTHR>(xp, XS, xrows, SB, slot, L, cin, pad);
  CONV_T(0, 1);
  if (tm_begin < tm_end) xplan_issue<T>(xp, xr, x, (tm_begin / tiles_t) * SB, (tm_begin % tiles_t) * BT, B, L, cin, pad);
  CONV_T(0, 2);

  typename Mm::Frag wf[WREG ? MT : 1][WREG ? kWRegSteps : 1];
  if constexpr (WREG) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int ks = 0; ks < kWRegSteps; ++ks) {
        const int ch = col0 + chan_of<T, MT>(mt, r16), kk = ks * KSTEP + kl;
        typename Mm::Frag f;
        if constexpr (BF) {
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = (T)0.0f;
        } else {
          f = (T)0.0f;
        }
        if (ch < N && kk < KK) f = *reinterpret_cast<const typename Mm::Frag*>(w + (long)ch * KK + kk);   // KK % 8 == 0 (bf16)
        wf[mt][ks] = f;
      }
  } else {   // LDS row mt*16 + m holds the weights of channel chan_of(mt, m)
    const int kvn = KKp / VEC;
    for (int lr = wave; lr < BN; lr += WAVES) {        // one weight row per wave and pass: no index arithmetic
      const int ch = col0 + chan_of<T, MT>(lr >> 4, lr & 15);
      const T* wrow = w + (long)min(ch, N - 1) * KK;
#pragma unroll 2
      for (int kv = lane; kv < kvn; kv += 64) {
        V val;
#pragma unroll
        for (int e = 0; e < VEC; ++e) val[e] = (T)0.0f;
        if (ch < N && kv * VEC < KK) val = *reinterpret_cast<const V*>(wrow + kv * VEC);   // KK % VEC == 0
        lds_store_vec<T>(wsr + lr * WSR + kv * VEC, val);
      }
    }
  }

  int xrow[NT], row_pk[NT];                            // this lane's output rows: LDS row of tap 0; (slot << 16) | time, -1 = none
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int row = wave * (NT * 16) + nt * 16 + r16;
    const int rr = multi ? min(row, SB * L - 1) : row;
    const int sq = multi ? rr / L : 0;
    xrow[nt] = multi ? sq * slot + (rr - sq * L) : row;
    row_pk[nt] = (multi && row >= SB * L) ? -1 : ((sq << 16) | (rr - sq * L));
  }
  Acc bv[MT][4], s1[MT][4], s2[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bv[mt][r] = FWD ? bias[min(col0 + g * CPL + mt * 4 + r, N - 1)] : (Acc)0;
      s1[mt][r] = 0;
      s2[mt][r] = 0;
    }
  const int tap0 = kl / cin, ci0 = kl - tap0 * cin, xo0 = tap0 * XS + ci0;
  const bool vec_out = (N % VEC) == 0 && (CPL % VEC) == 0;
  CONV_T(0, 3);

  for (int tm = tm_begin; tm < tm_end; ++tm) {
    const int b0 = (tm / tiles_t) * SB, t0 = (tm % tiles_t) * BT;
    __syncthreads();                                   // previous tile's LDS reads are done
    CONV_T(0, 4 + (tm - tm_begin) * 5 + 0);
    xplan_commit<T>(xp, xr, xs);
    if (xrows * (cin / VEC) > kXV * NTHR) stage_x_tile<T, NTHR>(x, xs, XS, xrows, SB, slot, b0, t0, B, L, cin, pad, kXV * NTHR);
    __syncthreads();
    CONV_T(0, 4 + (tm - tm_begin) * 5 + 1);
    if (tm + 1 < tm_end) xplan_issue<T>(xp, xr, x, ((tm + 1) / tiles_t) * SB, ((tm + 1) % tiles_t) * BT, B, L, cin, pad);
    CONV_T(0, 4 + (tm - tm_begin) * 5 + 2);

    typename Mm::AccV acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mt][nt][r] = 0;
    int xo = xo0, ci = ci0;
    if constexpr (WREG) {
#pragma unroll
      for (int ks = 0; ks < kWRegSteps; ++ks) {
        if (ks < nks) {
          typename Mm::Frag bf[NT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bf[nt] = *reinterpret_cast<const typename Mm::Frag*>(xs + xrow[nt] * XS + xo);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = Mm::mma(wf[mt][ks], bf[nt], acc[mt][nt]);
          ci += KSTEP;
          xo += KSTEP;
          while (ci >= cin) { ci -= cin; xo += XS - cin; }   // next tap: one LDS row down
        }
      }
    } else {
      const T* wl = wsr + r16 * WSR + kl;
#pragma unroll 2
      for (int ks = 0; ks < nks; ++ks) {
        typename Mm::Frag af[MT], bf[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const typename Mm::Frag*>(wl + mt * 16 * WSR + ks * KSTEP);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = *reinterpret_cast<const typename Mm::Frag*>(xs + xrow[nt] * XS + xo);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = Mm::mma(af[mt], bf[nt], acc[mt][nt]);
        ci += KSTEP;
        xo += KSTEP;
        while (ci >= cin) { ci -= cin; xo += XS - cin; }
      }
    }

    CONV_T(0, 4 + (tm - tm_begin) * 5 + 3);
    // ---- epilogue: straight from the accumulators
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int pk = row_pk[nt];
      if (pk < 0 || b0 + (pk >> 16) >= B || t0 + (pk & 0xffff) >= L) continue;
      T ov[CPL];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const Acc v = acc[mt][nt][r] + bv[mt][r];
          ov[mt * 4 + r] = (T)v;
          if (FWD) { s1[mt][r] += v; s2[mt][r] += v * v; }
        }
      const int col = col0 + g * CPL;
      T* dst = out + ((long)(b0 + (pk >> 16)) * L + t0 + (pk & 0xffff)) * N + col;
      if (vec_out && col + CPL <= N) {
#pragma unroll
        for (int q = 0; q < CPL / VEC; ++q) {
          V o;
#pragma unroll
          for (int e = 0; e < VEC; ++e) o[e] = ov[q * VEC + e];
          *reinterpret_cast<V*>(dst + q * VEC) = o;
        }
      } else {
#pragma unroll
        for (int j = 0; j < CPL; ++j)
          if (col + j < N) dst[j] = ov[j];
      }
    }
    CONV_T(0, 4 + (tm - tm_begin) * 5 + 4);
  }

  if (FWD) {   // one partial row per workgroup: lanes of equal channel set meet (16 row lanes), then the four waves
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const Acc a = row16_sum<Acc>(s1[mt][r]), b = row16_sum<Acc>(s2[mt][r]);
        if (r16 == 0) {
          red[(wave * 2 + 0) * BN + g * CPL + mt * 4 + r] = a;
          red[(wave * 2 + 1) * BN + g * CPL + mt * 4 + r] = b;
        }
      }
    __syncthreads();
    if (threadIdx.x < 2 * BN) {
      const int c = threadIdx.x % BN, which = threadIdx.x / BN;
      if (col0 + c < N) {
        Acc t = 0;
#pragma unroll
        for (int wv = 0; wv < WAVES; ++wv) t += red[(wv * 2 + which) * BN + c];
        partial[((long)bm * 2 + which) * N + col0 + c] = t;
      }
    }
  }
  CONV_T(0, 60);
}

template <typename T, int MT, bool WREG> static size_t conv_t_lds(int cin, int KK, int xrows, int waves) {
  using Acc = typename AccOf<T>::type;
  constexpr int KSTEP = Mma<T>::KSTEP;
  const size_t KKp = (size_t)(KK + KSTEP - 1) / KSTEP * KSTEP;
  const size_t xs = ((size_t)xrows * conv_t_xpitch(cin, (int)sizeof(T)) + 7) & ~(size_t)7;
  const size_t ws = WREG ? 0 : (((size_t)16 * MT * (KKp + DCfg<T>::WPAD) + 7) & ~(size_t)7);
  return (((xs + ws) * sizeof(T) + (size_t)waves * 2 * 16 * MT * sizeof(Acc)) + 15) & ~(size_t)15;
}


// NTHR = 512 (fp32): waves 4-7 are a second group on the same 128 x BN tile that multiplies the ODD k-steps of every weight chunk
// (the first group the even ones); the two accumulator sets meet in the LDS output slab (first + second).  Two workgroups fit a
// CU either way (the activation tile sets the LDS), so this is four waves per SIMD instead of two -- what the fp32 matrix pipe
// needs to stay fed through LDS reads and barriers (gemm_jobs.h).
template <typename T, int BN, bool FWD, int NTHR>
__global__ __launch_bounds__(NTHR, NTHR == 512 ? 4 : 2) void conv_direct_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                               const typename AccOf<T>::type* __restrict__ bias, T* __restrict__ out,
                                                               typename AccOf<T>::type* __restrict__ partial, int B, int L, int cin,
                                                               int KK, int N, int pad, int SB, int tiles_t, int slot, int tiles_n) {
  using Mm = Mma<T>;
  using Acc = typename Mm::Acc;
  using V = typename Vec16<T>::type;
  constexpr int VEC = Elem<T>::VEC, KSTEP = Mm::KSTEP, WCH = DCfg<T>::WCH, BT = kConvBT, MI = 2, NI = BN / 16;
  constexpr int WS = WCH + DCfg<T>::WPAD, CS = BN + 4;
  constexpr int WV = BN * WCH / VEC / NTHR;       // weight vectors per thread and chunk
  constexpr int KSPLIT = NTHR / 256;              // groups of four waves sharing the reduction
  static_assert(WV >= 1 && BN * WCH / VEC % NTHR == 0, "weight chunk / thread count");
  constexpr bool BF = sizeof(T) == 2;
  extern __shared__ __attribute__((aligned(16))) char arena[];
  const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
  const int b0 = (tm / tiles_t) * SB, t0 = (tm % tiles_t) * BT, col0 = tn * BN;
  const int XS = cin + DCfg<T>::XPAD, xrows = SB * slot + kXExtra;
  T* xs = reinterpret_cast<T*>(arena);
  T* ws0 = xs + (((long)xrows * XS + 7) & ~7L);
  T* ws1 = ws0 + BN * WS;
  const int KKp = (KK + KSTEP - 1) / KSTEP * KSTEP, nch = (KKp + WCH - 1) / WCH;

  stage_x_tile<T, NTHR>(x, xs, XS, xrows, SB, slot, b0, t0, B, L, cin, pad);
  V wreg[WV];
  auto loadw = [&](int ch) {
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int v = threadIdx.x + i * NTHR;
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
      const int v = threadIdx.x + i * NTHR;
      const int n = v / (WCH / VEC), kc = (v % (WCH / VEC)) * VEC;
      lds_store_vec<T>(dst + n * WS + kc, wreg[i]);
    }
  };
  loadw(0);
  storew(ws0);
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, sub = threadIdx.x >> 8, g = lane >> 4, r16 = lane & 15;
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
    for (int k2 = 0; k2 < WCH / KSTEP / KSPLIT; ++k2) {
      const int ks = k2 * KSPLIT + (KSPLIT > 1 ? sub : 0);
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
  for (int q = 0; q < KSPLIT; ++q) {               // first group stores, the second adds
    if (sub == q) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            Acc* dst = &cs[(wave * 32 + mi * 16 + Mm::acc_row(lane, r)) * CS + ni * 16 + r16];
            *dst = q == 0 ? acc[mi][ni][r] : *dst + acc[mi][ni][r];
          }
    }
    __syncthreads();
  }
  const bool multi = L < BT;
  for (int gidx = threadIdx.x; gidx < BT * BN / 4; gidx += NTHR) {
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
    constexpr int PARTS = NTHR / BN;
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

template <typename T> constexpr int kDirectThreads = sizeof(T) == 4 ? 512 : 256;   // fp32: two groups of four waves (conv_direct_kernel)
template <typename T, int BN> static size_t conv_direct_lds(int cin, int SB, int slot) {
  using Acc = typename AccOf<T>::type;
  const size_t xs = ((size_t)(SB * slot + kXExtra) * (cin + DCfg<T>::XPAD) + 7) & ~(size_t)7;
  const size_t op = (xs + 2 * (size_t)BN * (DCfg<T>::WCH + DCfg<T>::WPAD)) * sizeof(T);
  const size_t ep = ((size_t)kConvBT * (BN + 4) + 2 * kDirectThreads<T>) * sizeof(Acc);
  return ((op > ep ? op : ep) + 15) & ~(size_t)15;
}

constexpr size_t kMaxDirectLds = 150 * 1024;


}  // namespace emb
#include "conv_t_stream.h"
namespace emb {

// transposed streaming kernel; returns 1 when the shapes do not qualify
template <typename T, int MT, int WAVES, bool FWD, bool WREG>
static int launch_conv_t_cfg(const void* x, const void* w, const void* bias, void* out, void* partial, int* partial_rows, int B, int L,
                             int cin, int KK, int N, int pad, hipStream_t s) {
  using Acc = typename AccOf<T>::type;
  constexpr int NT = 4, BT = WAVES * 16 * NT;
  const ConvTiling t = conv_tiling_bt(B, L, pad, BT);
  const int xrows = t.SB * t.slot + kXExtra, tiles_n = cdiv(N, 16 * MT);
  const size_t lds = conv_t_lds<T, MT, WREG>(cin, KK, xrows, WAVES);
  if (lds > kMaxDirectLds) return 1;
  int per_cu = (int)((160 * 1024) / lds);               // workgroups resident per CU: LDS, then ~2 waves per SIMD of registers
  const int cap = WAVES == 4 ? 2 : 1;                   // (eight-wave workgroups two per CU: measured slower, fp32 first blocks +14 %)
  per_cu = per_cu < 1 ? 1 : (per_cu > cap ? cap : per_cu);
  const int target = 256 * per_cu;
  const int tpb = cdiv(t.tiles_m * tiles_n, target) < 1 ? 1 : cdiv(t.tiles_m * tiles_n, target);
  const int nblk_m = cdiv(t.tiles_m, tpb);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_t_kernel<T, MT, NT, WAVES, FWD, WREG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kMaxDirectLds);
    attr = true;
  }
  conv_t_kernel<T, MT, NT, WAVES, FWD, WREG><<<nblk_m * tiles_n, WAVES * 64, lds, s>>>((const T*)x, (const T*)w, (const Acc*)bias, (T*)out,
                                                                                  (Acc*)partial, B, L, cin, KK, N, pad, t.SB, t.tiles_t,
                                                                                  t.slot, t.tiles_m, tpb, nblk_m);
  EMB_CHECK_LAUNCH();
  if (partial_rows) *partial_rows = nblk_m;
  return EMB_OK;
}

template <typename T, int MT, bool FWD>
static int launch_conv_t(const void* x, const void* w, const void* bias, void* out, void* partial, int* partial_rows, int B, int L, int cin,
                         int KK, int N, int pad, hipStream_t s) {
  constexpr int KSTEP = Mma<T>::KSTEP, VEC = Elem<T>::VEC;
  if (!aligned16(x) || !aligned16(w) || !aligned16(out) || KK % VEC || cin % VEC || (sizeof(T) == 2 && KK % 8)) return 1;
  const bool wreg = cdiv(KK, KSTEP) <= kWRegSteps && MT * kWRegSteps * (int)(sizeof(typename Mma<T>::Frag) / 4) <= 64;
  if constexpr (sizeof(T) == 2) {   // bf16: operands streamed by LDS-DMA (conv_t_stream.h)
    constexpr bool stream_on = true;
    if (!wreg && stream_on) {
      const int rows = launch_conv_t_stream<MT, FWD>(x, w, bias, out, partial, B, L, cin, KK, N, pad, s);
      if (rows < 0) return rows;
      if (rows > 0) {
        if (partial_rows) *partial_rows = rows;
        return EMB_OK;
      }
    }
  }
  if (wreg) return launch_conv_t_cfg<T, MT, 4, FWD, true>(x, w, bias, out, partial, partial_rows, B, L, cin, KK, N, pad, s);
  // weights in LDS: eight waves share them (two per SIMD to cover the LDS latency); four when that does not fit
  const int rc = launch_conv_t_cfg<T, MT, 8, FWD, false>(x, w, bias, out, partial, partial_rows, B, L, cin, KK, N, pad, s);
  if (rc != 1) return rc;
  return launch_conv_t_cfg<T, MT, 4, FWD, false>(x, w, bias, out, partial, partial_rows, B, L, cin, KK, N, pad, s);
}

template <typename T, int BN, bool FWD>
static int launch_direct(const void* x, const void* w, const void* bias, void* out, void* partial, int* partial_rows, int B, int L, int cin,
                         int KK, int N, int pad, hipStream_t s) {
  using Acc = typename AccOf<T>::type;
  {
    const int rc = N <= 16 ? launch_conv_t<T, 1, FWD>(x, w, bias, out, partial, partial_rows, B, L, cin, KK, N, pad, s)
                 : N <= 32 ? launch_conv_t<T, 2, FWD>(x, w, bias, out, partial, partial_rows, B, L, cin, KK, N, pad, s)
                           : launch_conv_t<T, 4, FWD>(x, w, bias, out, partial, partial_rows, B, L, cin, KK, N, pad, s);
    if (rc != 1) return rc;
  }
  const ConvTiling t = conv_tiling(B, L, pad);
  const int tiles_n = cdiv(N, BN);
  const size_t lds = conv_direct_lds<T, BN>(cin, t.SB, t.slot);
  if (lds > kMaxDirectLds) return 1;   // caller falls back to the generic GEMM view
  static size_t attr = 0;
  if (lds > attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_direct_kernel<T, BN, FWD, kDirectThreads<T>>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDirectLds);
    attr = kMaxDirectLds;
  }
  conv_direct_kernel<T, BN, FWD, kDirectThreads<T>><<<t.tiles_m * tiles_n, kDirectThreads<T>, lds, s>>>((const T*)x, (const T*)w, (const Acc*)bias, (T*)out,
                                                                        (Acc*)partial, B, L, cin, KK, N, pad, t.SB, t.tiles_t, t.slot, tiles_n);
  EMB_CHECK_LAUNCH();
  if (partial_rows) *partial_rows = t.tiles_m;
  return EMB_OK;
}

template <typename T> static int launch_direct_t(bool fwd, const void* x, const void* w, const void* bias, void* out, void* partial,
                                                 int* partial_rows, int B, int L, int cin, int KK, int N, int pad, hipStream_t s) {
  if (N >= 64) return fwd ? launch_direct<T, 64, true>(x, w, bias, out, partial, partial_rows, B, L, cin, KK, N, pad, s)
                          : launch_direct<T, 64, false>(x, w, bias, out, partial, partial_rows, B, L, cin, KK, N, pad, s);
  return fwd ? launch_direct<T, 32, true>(x, w, bias, out, partial, partial_rows, B, L, cin, KK, N, pad, s)
             : launch_direct<T, 32, false>(x, w, bias, out, partial, partial_rows, B, L, cin, KK, N, pad, s);
}

int launch_conv_direct(int dtype, bool fwd, const void* x, const void* w, const void* bias, void* out, void* partial, int* partial_rows,
                       int B, int L, int cin, int KK, int N, int pad, hipStream_t s) {
  switch (dtype) {
    case EMB_F32: return launch_direct_t<float>(fwd, x, w, bias, out, partial, partial_rows, B, L, cin, KK, N, pad, s);
    case EMB_BF16: return launch_direct_t<__bf16>(fwd, x, w, bias, out, partial, partial_rows, B, L, cin, KK, N, pad, s);
    case EMB_F64: return launch_direct_t<double>(fwd, x, w, bias, out, partial, partial_rows, B, L, cin, KK, N, pad, s);
  }
  return EMB_ERR_DTYPE;
}

// ------------------------------------------------------------------------------------ weight gradient
// slab[slice][o][n] += sum over the slice's row tiles of dy[r][o] * xview[r][n]  (n = tap*cin + ci);  column KK = sum dy
// NTHR = 512 (fp32): waves 4-7 multiply the odd k-steps of every row tile, waves 0-3 the even ones (conv_direct_kernel's reason:
// four waves per SIMD); the second group's sums cross to the first through LDS after the last tile.
template <typename T, int BMW, int NTHR>
__global__ __launch_bounds__(NTHR, NTHR == 512 ? 4 : 2) void conv_wgrad_direct_kernel(const T* __restrict__ dy, const T* __restrict__ x,
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
  constexpr int KSPLIT = NTHR / 256;
  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, sub = threadIdx.x >> 8, g = lane >> 4, r16 = lane & 15, q = r16 >> 2, p = r16 & 3;
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
  // bias gradient (column sums of dy) rides on the matrix cores: dy^T x ones, accumulated by wave 0 of the nt == 0 workgroups
  typename Mm::AccV bias_acc[MIW];
#pragma unroll
  for (int mi = 0; mi < MIW; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) bias_acc[mi][r] = 0;
  typename Mm::Frag ones;
  if constexpr (BF) {
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (T)1.0f;
  } else {
    ones = (T)1.0f;
  }
  const bool do_bias = nt == 0 && wave == 0;

  const int per = (tiles_m + S - 1) / S, tm_begin = slice * per, tm_end = min(tiles_m, tm_begin + per);
  CONV_T(1, 0);
  constexpr bool PF = sizeof(T) < 8;                       // register-staged prefetch (f64: direct staging)
  constexpr int DV = BT * (BMW / VEC) / NTHR;              // dy-tile vectors per thread
  static_assert(DV >= 1 && BT * (BMW / VEC) % NTHR == 0, "dy tile / thread count");
  const bool multi = L < BT;
  XPlan<T> xp;
  V xr[kXV], dr[DV];
  int d_glb[DV], d_pk[DV];
  if constexpr (PF) {
    xplan_init<T, NTHR>(xp, XS, xrows, SB, slot, L, cin, pad);
#pragma unroll
    for (int i = 0; i < DV; ++i) {
      const int idx = threadIdx.x + i * NTHR;
      const int r = idx / (BMW / VEC), cv = (idx % (BMW / VEC)) * VEC;
      const int sq = multi ? r / L : 0, tl = multi ? r - sq * L : r;
      d_glb[i] = (sq * L + tl) * Cout + o0 + cv;
      d_pk[i] = ((multi ? r < SB * L : true) && o0 + cv < Cout) ? ((sq << 16) | tl) : -1;
    }
  }
  auto issue = [&](int tm) {
    const int b0 = (tm / tiles_t) * SB, t0 = (tm % tiles_t) * BT;
    xplan_issue<T>(xp, xr, x, b0, t0, B, L, cin, pad);
    const T* db = dy + ((long)b0 * L + t0) * Cout;
#pragma unroll
    for (int i = 0; i < DV; ++i) {
      V val;
#pragma unroll
      for (int e = 0; e < VEC; ++e) val[e] = (T)0.0f;
      if (d_pk[i] >= 0 && b0 + (d_pk[i] >> 16) < B && t0 + (d_pk[i] & 0xffff) < L) val = *reinterpret_cast<const V*>(db + d_glb[i]);
      dr[i] = val;
    }
  };
  if constexpr (PF) {
    if (tm_begin < tm_end) issue(tm_begin);
  }
  for (int tm = tm_begin; tm < tm_end; ++tm) {
    const int b0 = (tm / tiles_t) * SB, t0 = (tm % tiles_t) * BT;
    __syncthreads();   // previous tile fully consumed
    CONV_T(1, 1 + (tm - tm_begin) * 4 + 0);
    if constexpr (PF) {
      if (!CONV_DBG(3)) xplan_commit<T>(xp, xr, xs);
      if (xrows * (cin / VEC) > kXV * NTHR) stage_x_tile<T, NTHR>(x, xs, XS, xrows, SB, slot, b0, t0, B, L, cin, pad, kXV * NTHR);
#pragma unroll
      for (int i = 0; i < (CONV_DBG(3) ? 0 : DV); ++i) {
        const int idx = threadIdx.x + i * NTHR;
        *reinterpret_cast<V*>(dys + (idx / (BMW / VEC)) * DS + (idx % (BMW / VEC)) * VEC) = dr[i];   // DS * sizeof(T) % 16 == 0
      }
    } else {
      stage_x_tile<T, NTHR>(x, xs, XS, xrows, SB, slot, b0, t0, B, L, cin, pad);
      for (int i = threadIdx.x; i < BT * (BMW / VEC); i += NTHR) {   // dy rows of this tile, zero where invalid
        const int r = i / (BMW / VEC), cv = (i % (BMW / VEC)) * VEC;
        const int sq = multi ? r / L : 0, tl = multi ? r - sq * L : r;
        const bool rv = (multi ? r < SB * L : true) && (b0 + sq < B) && (t0 + tl < L);
        V v;
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = (T)0.0f;
        if (rv && o0 + cv < Cout) v = *reinterpret_cast<const V*>(dy + ((long)(b0 + sq) * L + t0 + tl) * Cout + o0 + cv);
        *reinterpret_cast<V*>(dys + r * DS + cv) = v;
      }
    }
    __syncthreads();
    CONV_T(1, 1 + (tm - tm_begin) * 4 + 1);
    if constexpr (PF) {
      if (tm + 1 < tm_end && !CONV_DBG(2)) issue(tm + 1);   // in flight during this tile's MFMA loop
    }
    CONV_T(1, 1 + (tm - tm_begin) * 4 + 2);
#pragma unroll 4
    for (int k2 = 0; k2 < (CONV_DBG(0) ? 0 : BT / KSTEP / KSPLIT); ++k2) {
      const int ks = k2 * KSPLIT + (KSPLIT > 1 ? sub : 0);
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
      if (do_bias && !CONV_DBG(1)) {
#pragma unroll
        for (int mi = 0; mi < MIW; ++mi) bias_acc[mi] = Mm::mma(af[mi], ones, bias_acc[mi]);
      }
    }
    CONV_T(1, 1 + (tm - tm_begin) * 4 + 3);
  }
  if constexpr (KSPLIT > 1) {   // second group's sums -> first group: [wave][register][lane] floats, lane-linear
    constexpr int NREG = MIW * NIW * 4 + MIW * 4;
    Acc* xch = reinterpret_cast<Acc*>(arena);
    __syncthreads();            // the last tile's operands are dead
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      if (sub == 1 - pass) {
#pragma unroll
        for (int mi = 0; mi < MIW; ++mi) {
#pragma unroll
          for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              Acc* e = &xch[(wave * NREG + (mi * NIW + ni) * 4 + r) * 64 + lane];
              if (pass == 0) *e = acc[mi][ni][r]; else acc[mi][ni][r] += *e;
            }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            Acc* e = &xch[(wave * NREG + MIW * NIW * 4 + mi * 4 + r) * 64 + lane];
            if (pass == 0) *e = bias_acc[mi][r]; else bias_acc[mi][r] += *e;
          }
        }
      }
      if (pass == 0) __syncthreads();
    }
    if (sub != 0) return;
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
  if (do_bias && r16 == 0) {   // every column of dy^T x ones holds the sum; lane column 0 writes it
#pragma unroll
    for (int mi = 0; mi < MIW; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = o0 + mi * 16 + Mm::acc_row(lane, r);
        if (o < Cout) dst[(long)o * (KK + 1) + KK] = bias_acc[mi][r];
      }
  }
  CONV_T(1, 60);
}

template <typename T, int BMW> static size_t wgrad_direct_lds(int cin, int SB, int slot) {
  constexpr int DS = BMW + (sizeof(T) == 2 ? 8 : 16);
  const size_t xs = ((size_t)(SB * slot + kXExtra) * (cin + DCfg<T>::XPAD) + 7) & ~(size_t)7;
  const size_t op = (((xs + (size_t)kConvBT * DS) * sizeof(T) + kConvBT * sizeof(int)) + 15) & ~(size_t)15;
  // two groups of waves (fp32): the exchange of the second group's sums, [4 waves][registers][64 lanes]
  const size_t xch = kDirectThreads<T> > 256 ? (size_t)4 * ((BMW / 16) * (DCfg<T>::BNW / 64) * 4 + (BMW / 16) * 4) * 64 * sizeof(typename AccOf<T>::type) : 0;
  return op > xch ? op : xch;
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

static bool wgrad_stream_enabled() {
  constexpr bool on = true;
  return on;
}

int conv_wgrad_slices(int B, int L, int cin, int pad, int KK, int Cout, int dtype) {
  if (dtype == EMB_BF16 && wgrad_stream_enabled() && conv_wgrad_stream_shape_ok(B, L, cin, KK, Cout, pad))
    return conv_wgrad_stream_slices(B, L, KK, pad);
  switch (dtype) {
    case EMB_F32: return wgrad_slices_t<float>(B, L, pad, KK, Cout);
    case EMB_BF16: return wgrad_slices_t<__bf16>(B, L, pad, KK, Cout);
    default: return wgrad_slices_t<double>(B, L, pad, KK, Cout);
  }
}

template <typename T, int BMW, int NTHR>
static int launch_wgrad_n(const void* dy, const void* x, void* slab, int B, int L, int cin, int KK, int Cout, int pad, int S, hipStream_t s) {
  using Acc = typename AccOf<T>::type;
  const ConvTiling t = conv_tiling(B, L, pad);
  const size_t lds = wgrad_direct_lds<T, BMW>(cin, t.SB, t.slot);
  if (lds > kMaxDirectLds) return 1;
  static size_t attr = 0;
  if (lds > attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_direct_kernel<T, BMW, NTHR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDirectLds);
    attr = kMaxDirectLds;
  }
  const int n_tiles = cdiv(KK, DCfg<T>::BNW), m_tiles = cdiv(Cout, BMW);
  conv_wgrad_direct_kernel<T, BMW, NTHR><<<n_tiles * m_tiles * S, NTHR, lds, s>>>((const T*)dy, (const T*)x, (Acc*)slab, B, L, cin, KK, Cout, pad, t.SB,
                                                                                 t.tiles_t, t.slot, t.tiles_m, n_tiles, m_tiles, S);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}
template <typename T, int BMW>
static int launch_wgrad(const void* dy, const void* x, void* slab, int B, int L, int cin, int KK, int Cout, int pad, int S, hipStream_t s) {
  // the second group of waves pays where the matrix work dominates the tile (>= 32 input channels); the first block's 4-channel
  // tiles are staging-bound and ran 30 % slower with it
  if constexpr (kDirectThreads<T> > 256)
    if (cin >= 32) return launch_wgrad_n<T, BMW, kDirectThreads<T>>(dy, x, slab, B, L, cin, KK, Cout, pad, S, s);
  return launch_wgrad_n<T, BMW, 256>(dy, x, slab, B, L, cin, KK, Cout, pad, S, s);
}

template <typename T> static int launch_wgrad_t(const void* dy, const void* x, void* slab, int B, int L, int cin, int KK, int Cout,
                                                int pad, int S, hipStream_t s) {
  if (Cout > 32) return launch_wgrad<T, 64>(dy, x, slab, B, L, cin, KK, Cout, pad, S, s);
  return launch_wgrad<T, 32>(dy, x, slab, B, L, cin, KK, Cout, pad, S, s);
}

// ---- weight gradient and input gradient of a stored-activation block in ONE launch (bf16 streaming kernels): both consume dy and
// neither reads the other's output.  The first workgroups run the weight-gradient body, the rest the input-gradient body; as
// the weight-gradient workgroups retire at different times the input-gradient workgroups fill the CUs behind them, so the two
// tails overlap and one launch boundary disappears.
template <int MIW, int CPT>
__global__ __launch_bounds__(512, 2) void conv_bwd_dual_kernel(const WsArgs wa, const CtsArgs ca, const int nws) {
  if ((int)blockIdx.x < nws) conv_wgrad_stream_body<MIW>(wa, (int)blockIdx.x);
  else conv_t_stream_body<4, CPT, false>(ca, (int)blockIdx.x - nws);
}

// returns EMB_OK when the dual launch ran, 1 when the shapes / pointers do not qualify (the caller launches the two kernels)
int launch_conv_bwd_dual(const void* dy, const void* x, void* slab, const void* wflip, void* dx, int B, int L, int cin, int k, int Cout,
                         int pad, int S, hipStream_t s) {
  constexpr bool on = true;
  const int KK = k * cin, KKd = k * Cout;
  if (!on || !wgrad_stream_enabled() || cin != 64 || !conv_wgrad_stream_shape_ok(B, L, cin, KK, Cout, pad) || !aligned16(dy) || !aligned16(x) ||
      !aligned16(wflip) || !aligned16(dx) || S > conv_tiling(B, L, pad).tiles_m)
    return 1;
  if (cdiv(KKd, 32) <= kWRegSteps) return 1;                       // (tiny reductions keep the weights-in-registers kernel)
  const size_t lds_d = conv_t_stream_lds<4>(B, L, Cout, KKd, cin, pad);
  if (lds_d == 0) return 1;
  WsArgs wa{};
  const size_t lds_w = wgrad_stream_fill(wa, dy, x, slab, B, L, KK, Cout, pad, S);
  const ConvTiling t = conv_tiling_bt(B, L, pad, kCtsBT);
  CtsArgs ca{};
  ca.x = (const __bf16*)dy; ca.w = (const __bf16*)wflip; ca.bias = nullptr; ca.out = (__bf16*)dx; ca.partial = nullptr;
  ca.B = B; ca.L = L; ca.cin = Cout; ca.KK = KKd; ca.N = cin; ca.pad = pad; ca.SB = t.SB; ca.slot = t.slot; ca.tiles_m = t.tiles_m;
  ca.taps = k;
  ca.tpb = cdiv(t.tiles_m, 256) < 1 ? 1 : cdiv(t.tiles_m, 256);
  ca.nblk_m = cdiv(t.tiles_m, ca.tpb);
  const int nws = wa.n_tiles * S, nct = ca.nblk_m;               // (cin = 64 = 16 * MT: one column tile)
  const size_t lds = lds_w > lds_d ? lds_w : lds_d;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bwd_dual_kernel<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bwd_dual_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  if (Cout == 32) conv_bwd_dual_kernel<2, 1><<<nws + nct, 512, lds, s>>>(wa, ca, nws);
  else conv_bwd_dual_kernel<4, 2><<<nws + nct, 512, lds, s>>>(wa, ca, nws);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

int launch_conv_wgrad_direct(int dtype, const void* dy, const void* x, void* slab, int B, int L, int cin, int KK, int Cout, int pad,
                             int S, hipStream_t s) {
  if (dtype == EMB_BF16 && wgrad_stream_enabled() && conv_wgrad_stream_shape_ok(B, L, cin, KK, Cout, pad) && aligned16(dy) && aligned16(x) &&
      S <= conv_tiling(B, L, pad).tiles_m)
    return launch_wgrad_stream(dy, x, slab, B, L, KK, Cout, pad, S, s);
  switch (dtype) {
    case EMB_F32: return launch_wgrad_t<float>(dy, x, slab, B, L, cin, KK, Cout, pad, S, s);
    case EMB_BF16: return launch_wgrad_t<__bf16>(dy, x, slab, B, L, cin, KK, Cout, pad, S, s);
    case EMB_F64: return launch_wgrad_t<double>(dy, x, slab, B, L, cin, KK, Cout, pad, S, s);
  }
  return EMB_ERR_DTYPE;
}

#ifdef EMB_CONV_PROF
extern "C" int emb_debug_conv_prof(unsigned long long* out, int select) {
  int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv_prof), sizeof(unsigned long long) * 64);
  unsigned long long z[64] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_prof), z, sizeof(z));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_prof_sel), &select, sizeof(int));
  return rc;
}
extern "C" int emb_debug_conv_dbg(int bits) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_dbg), &bits, sizeof(int)); }
#endif

}  // namespace emb
