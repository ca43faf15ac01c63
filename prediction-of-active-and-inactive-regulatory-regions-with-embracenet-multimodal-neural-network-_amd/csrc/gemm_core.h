// LDS-tiled MFMA GEMM building blocks for gfx950, shared by every dense contraction of the path
// (docking forward, dgrad, wgrad, post-stack linears).
//
//   C[M,N] (+)= sum_k A(m,k) * B(n,k)
//
// Each operand is either "row-major" in global memory (element (r,k) at g[r*ld + k]: k contiguous) or
// "K-major" (element (r,k) at g[k*ld + r]: r contiguous).  The LDS image keeps the global orientation,
// so every global->LDS copy is a straight 16-byte-per-lane coalesced copy; the transposition needed by
// K-major operands happens in the fragment read (scalar reads for f32/f64 whose MFMA operand is one
// element per lane; ds_read_b64_tr_b16 for bf16).
//
// MFMA shapes: 16x16xK for all three storage types (K = 32 bf16, 4 f32, 4 f64), f32 accumulate
// (f64 for f64).  A workgroup is 4 waves arranged WM x WN x WK; WK > 1 splits the K range of a tile
// over waves (small M*N, long K), the partial tiles are summed through LDS in a fixed order.
#pragma once
#include "common.h"

namespace emb {

// ------------------------------------------------------------------------------------------ MFMA
template <typename T> struct Mma;

template <> struct Mma<float> {
  using Acc = float;
  using AccV = f32x4;
  using Frag = float;
  static constexpr int KSTEP = 4;
  __device__ static AccV mma(Frag a, Frag b, AccV c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  __device__ static int acc_row(int lane, int r) { return 4 * (lane >> 4) + r; }
};
template <> struct Mma<double> {
  using Acc = double;
  using AccV = f64x4;
  using Frag = double;
  static constexpr int KSTEP = 4;
  __device__ static AccV mma(Frag a, Frag b, AccV c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  __device__ static int acc_row(int lane, int r) { return (lane >> 4) + 4 * r; }  // f64 C/D map differs
};
template <> struct Mma<__bf16> {
  using Acc = float;
  using AccV = f32x4;
  using Frag = bf16x8;
  static constexpr int KSTEP = 32;
  __device__ static AccV mma(Frag a, Frag b, AccV c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  __device__ static int acc_row(int lane, int r) { return 4 * (lane >> 4) + r; }
};

// --------------------------------------------------------------------------------- LDS pitches
// Row-major image [ROWS][BK + pad]; K-major image [BK][ROWS + pad].  Pads chosen so the fragment
// reads of a 32-lane half hit distinct banks and 16-byte vector stores stay aligned.
template <typename T, bool KMAJOR, int ROWS, int BK> struct Pitch;
template <int ROWS, int BK> struct Pitch<float, false, ROWS, BK> { static constexpr int S = BK + 2; };
template <int ROWS, int BK> struct Pitch<double, false, ROWS, BK> { static constexpr int S = BK + 2; };
template <int ROWS, int BK> struct Pitch<__bf16, false, ROWS, BK> { static constexpr int S = BK + 8; };
template <int ROWS, int BK> struct Pitch<float, true, ROWS, BK> { static constexpr int S = ROWS + 16; };
template <int ROWS, int BK> struct Pitch<double, true, ROWS, BK> { static constexpr int S = ROWS + 16; };
template <int ROWS, int BK> struct Pitch<__bf16, true, ROWS, BK> { static constexpr int S = ROWS + 8; };

// ------------------------------------------------------------------------------ operand transforms
// Applied to an operand element while it is staged (global -> registers -> LDS); `code` is the byte
// at the same [row][k] position of a side array.
struct XfNone {
  static constexpr bool kUsesCode = false;
  template <typename T> __device__ T operator()(T v, uint8_t) const { return v; }
};
// embrace backward: keep dE where the element selected modality `m` and its ReLU was active
struct XfEmbraceMask {
  static constexpr bool kUsesCode = true;
  uint8_t want;  // (m ? EMB_CODE_IDX : 0) | EMB_CODE_ACTIVE
  template <typename T> __device__ T operator()(T v, uint8_t code) const {
    return ((code & (EMB_CODE_IDX | EMB_CODE_ACTIVE)) == want) ? v : (T)0.0f;
  }
};
// linear backward: bit0 = pre-activation > 0 (checked when relu), bit1 = kept by dropout
struct XfLinearMask {
  static constexpr bool kUsesCode = true;
  uint8_t need;   // bits that must be set
  float scale;    // 1 / (1 - p)
  template <typename T> __device__ T operator()(T v, uint8_t code) const {
    using A = typename AccOf<T>::type;
    return ((code & need) == need) ? (T)((A)v * (A)scale) : (T)0.0f;
  }
};

// ------------------------------------------------------------------------------- operand stager
template <typename T, bool KMAJOR, int ROWS, int BK, typename Xf> struct Stager {
  static constexpr int VEC = Elem<T>::VEC;
  static constexpr int S = Pitch<T, KMAJOR, ROWS, BK>::S;
  static constexpr int LDS_ELEMS = KMAJOR ? BK * S : ROWS * S;
  static constexpr int NVEC = ROWS * BK / VEC;           // 16-byte vectors per tile
  static constexpr int NV = (NVEC + kThreads - 1) / kThreads;
  static constexpr int INNER = (KMAJOR ? ROWS : BK) / VEC;  // vectors along the contiguous axis
  using V = typename Vec16<T>::type;

  const T* g;
  const uint8_t* code;
  int ld;       // global leading dimension (elements)
  int row0;     // first tile row
  int nrows;    // operand rows (bounds)
  int K;        // reduction length (bounds)
  bool vec_ok;  // 16-byte loads legal (alignment of base and ld)
  Xf xf;
  int ones_row;  // >= 0: operand row that reads as 1.0 for every k < K (bias-gradient column), else -1
  // Convolution view (cv_L > 0): the operand is the im2col matrix of a channels-last activation
  // x[B, L, cin]: rows R = b*L + t, columns KK = tap*cin + ci, element = x[b, t - pad + tap, ci] or 0 outside
  // [0, L).  Rows overlap in memory, so the address is simply (R - pad)*cin + KK; only validity depends on
  // (t, tap).  Row-major use: (r, k) = (R, KK); K-major use: (r, k) = (KK, R).  Requires cin % VEC == 0.
  int cv_L, cv_cin, cv_pad;
  V regs[NV];

  __device__ void load(int k0) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + i * kThreads;
      V val;
#pragma unroll
      for (int e = 0; e < VEC; ++e) val[e] = (T)0.0f;
      if (NVEC % kThreads == 0 || v < NVEC) {
        const int outer = v / INNER, inner = (v % INNER) * VEC;
        const int r = row0 + (KMAJOR ? inner : outer);
        const int k = k0 + (KMAJOR ? outer : inner);
        int nvalid;
        long off;
        if (cv_L > 0) {
          const int R = KMAJOR ? k : r, KK = KMAJOR ? r : k;
          const int nR = KMAJOR ? K : nrows, nKK = KMAJOR ? nrows : K;
          const int tt = R % cv_L - cv_pad + KK / cv_cin;     // time index the tap reads
          nvalid = (R < nR && tt >= 0 && tt < cv_L) ? nKK - KK : 0;
          off = ((long)R - cv_pad) * cv_cin + KK;
        } else {
          const int rlim = KMAJOR ? nrows - r : (r < nrows ? VEC : 0);   // valid elements along the vector...
          const int klim = KMAJOR ? (k < K ? VEC : 0) : K - k;           // ...and across it
          nvalid = KMAJOR ? (klim > 0 ? rlim : 0) : (rlim > 0 ? klim : 0);
          off = KMAJOR ? (long)k * ld + r : (long)r * ld + k;
        }
        if (nvalid >= VEC && vec_ok) {
          val = *reinterpret_cast<const V*>(g + off);
          if (Xf::kUsesCode) {
            uint64_t cw = 0;   // VEC code bytes, kept in registers (byte e = bits 8e..8e+7)
            if (code != nullptr) {
              if (VEC == 8) cw = *reinterpret_cast<const uint64_t*>(code + off);
              else if (VEC == 4) cw = *reinterpret_cast<const uint32_t*>(code + off);
              else cw = *reinterpret_cast<const uint16_t*>(code + off);
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) val[e] = xf(val[e], (uint8_t)(cw >> (8 * e)));
          }
        } else if (nvalid > 0) {
#pragma unroll
          for (int e = 0; e < VEC; ++e)
            if (e < nvalid) val[e] = xf(g[off + e], (Xf::kUsesCode && code != nullptr) ? code[off + e] : (uint8_t)0);
        }
        if (KMAJOR && ones_row >= 0 && k < K && ones_row >= r && ones_row < r + VEC) {
#pragma unroll
          for (int e = 0; e < VEC; ++e)
            if (r + e == ones_row) val[e] = (T)1.0f;
        }
      }
      regs[i] = val;
    }
  }

  __device__ void store(T* lds) const {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + i * kThreads;
      if (NVEC % kThreads == 0 || v < NVEC) {
        const int outer = v / INNER, inner = (v % INNER) * VEC;
        T* dst = lds + outer * S + inner;
        if (KMAJOR || sizeof(T) == 2) {
          *reinterpret_cast<V*>(dst) = regs[i];                // pitch keeps 16-byte alignment
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) dst[e] = regs[i][e];   // pitch BK+2: element stores, conflict-free
        }
      }
    }
  }
};

// ------------------------------------------------------------------------------- fragment reads
template <typename T, bool KMAJOR, int S> struct FragRead;
template <typename T, int S> struct FragRead<T, false, S> {  // f32 / f64, row-major image
  __device__ static T get(const T* tile, int row_base, int k_base, int lane) {
    return tile[(row_base + (lane & 15)) * S + k_base + (lane >> 4)];
  }
};
template <typename T, int S> struct FragRead<T, true, S> {  // f32 / f64, K-major image
  __device__ static T get(const T* tile, int row_base, int k_base, int lane) {
    return tile[(k_base + (lane >> 4)) * S + row_base + (lane & 15)];
  }
};
template <int S> struct FragRead<__bf16, false, S> {
  __device__ static bf16x8 get(const __bf16* tile, int row_base, int k_base, int lane) {
    return *reinterpret_cast<const bf16x8*>(tile + (row_base + (lane & 15)) * S + k_base + 8 * (lane >> 4));
  }
};
template <int S> struct FragRead<__bf16, true, S> {
  // image [k][row]: two transposing reads, each a 4(k) x 16(row) block per 16-lane group
  __device__ static bf16x8 get(const __bf16* tile, int row_base, int k_base, int lane) {
    const int g = lane >> 4, w = lane & 15, q = w >> 2, p = w & 3;
    // `tile` reaches here as a generic pointer (selected from the double-buffer table at run time), and a generic ->
    // LDS pointer cast costs a 64-bit add, a null compare and a select per read.  The low 32 bits of a generic LDS
    // address ARE the LDS offset (the aperture base lives in the high half), so form the 32-bit address directly.
    const uint32_t a0 = (uint32_t)(uintptr_t)tile + 2u * (uint32_t)((k_base + 8 * g + q) * S + row_base + 4 * p);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(a0 + 2u * 4u * S));
    union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
    u.s.lo = lo;
    u.s.hi = hi;
    return u.v;
  }
};

// ----------------------------------------------------------------------------------- tile config
template <typename T_, int BM_, int BN_, int BK_, int WM_, int WN_, int WK_, bool AKM_, bool BKM_> struct TileCfg {
  using T = T_;
  using M = Mma<T>;
  static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_, WK = WK_;
  static constexpr bool AKM = AKM_, BKM = BKM_;
  static_assert(WM * WN * WK == kWaves, "4 waves per workgroup");
  static constexpr int MI = BM / WM / 16, NI = BN / WN / 16;
  static constexpr int KW = BK / WK;                    // k range of one wave inside a chunk
  static_assert(KW % M::KSTEP == 0 && MI >= 1 && NI >= 1, "tile shape");
  static constexpr int SA = Pitch<T, AKM, BM, BK>::S, SB = Pitch<T, BKM, BN, BK>::S;
  static constexpr int A_ELEMS = AKM ? BK * SA : BM * SA;
  static constexpr int B_ELEMS = BKM ? BK * SB : BN * SB;
  static constexpr int OPERAND_BYTES = 2 * (A_ELEMS + B_ELEMS) * (int)sizeof(T);   // double-buffered
  static constexpr int CS = BN + 4;                                              // C slab pitch
  static constexpr int SLAB = BM * CS;                                           // elements per slab
};

// One pass over K for one output tile: acc += A_tile . B_tile^T.  Caller provides the LDS arena
// (>= Cfg::OPERAND_BYTES, 16-byte aligned).  Ends with all waves past the last barrier, so the arena
// may be reused immediately.
template <class Cfg, class StA, class StB>
__device__ __forceinline__ void gemm_mainloop(StA& sa, StB& sb, int K, char* arena,
                                              typename Cfg::M::AccV (&acc)[Cfg::MI][Cfg::NI], int k_begin = 0) {
  using T = typename Cfg::T;
  using M = typename Cfg::M;
  T* As[2] = {reinterpret_cast<T*>(arena), reinterpret_cast<T*>(arena) + Cfg::A_ELEMS};
  T* Bs[2] = {reinterpret_cast<T*>(arena) + 2 * Cfg::A_ELEMS, reinterpret_cast<T*>(arena) + 2 * Cfg::A_ELEMS + Cfg::B_ELEMS};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wk = wave % Cfg::WK, wn = (wave / Cfg::WK) % Cfg::WN, wm = wave / (Cfg::WK * Cfg::WN);
  const int nchunks = (K - k_begin + Cfg::BK - 1) / Cfg::BK;   // K is the exclusive end of the k range

  sa.load(k_begin);
  sb.load(k_begin);
  sa.store(As[0]);
  sb.store(Bs[0]);
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    const int cur = ch & 1;
    const bool more = ch + 1 < nchunks;
    if (more) {
      sa.load(k_begin + (ch + 1) * Cfg::BK);   // in flight while this chunk is multiplied
      sb.load(k_begin + (ch + 1) * Cfg::BK);
    }
    const T* At = As[cur];
    const T* Bt = Bs[cur];
#pragma unroll
    for (int ks = 0; ks < Cfg::KW / M::KSTEP; ++ks) {
      const int kb = wk * Cfg::KW + ks * M::KSTEP;
      if (k_begin + ch * Cfg::BK + kb < K) {   // wave-uniform: skip zero padding past K
        typename M::Frag af[Cfg::MI], bf[Cfg::NI];
#pragma unroll
        for (int mi = 0; mi < Cfg::MI; ++mi)
          af[mi] = FragRead<T, Cfg::AKM, Cfg::SA>::get(At, (wm * Cfg::MI + mi) * 16, kb, lane);
#pragma unroll
        for (int ni = 0; ni < Cfg::NI; ++ni)
          bf[ni] = FragRead<T, Cfg::BKM, Cfg::SB>::get(Bt, (wn * Cfg::NI + ni) * 16, kb, lane);
#pragma unroll
        for (int mi = 0; mi < Cfg::MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < Cfg::NI; ++ni) acc[mi][ni] = M::mma(af[mi], bf[ni], acc[mi][ni]);
      }
    }
    if (more) {
      sa.store(As[cur ^ 1]);
      sb.store(Bs[cur ^ 1]);
    }
    __syncthreads();
  }
}

// Sum the WK partial tiles into slab 0 of `cs` ([BM][CS] of Acc) in wave order (deterministic).
// On return (after the trailing barrier) slab 0 holds the full tile.
template <class Cfg>
__device__ __forceinline__ void reduce_to_slab(typename Cfg::M::AccV (&acc)[Cfg::MI][Cfg::NI],
                                               typename Cfg::M::Acc* cs) {
  using M = typename Cfg::M;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wk = wave % Cfg::WK, wn = (wave / Cfg::WK) % Cfg::WN, wm = wave / (Cfg::WK * Cfg::WN);
#pragma unroll
  for (int turn = 0; turn < Cfg::WK; ++turn) {
    if (wk == turn) {
#pragma unroll
      for (int mi = 0; mi < Cfg::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < Cfg::NI; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = (wm * Cfg::MI + mi) * 16 + M::acc_row(lane, r);
            const int col = (wn * Cfg::NI + ni) * 16 + (lane & 15);
            typename M::Acc* p = cs + row * Cfg::CS + col;
            *p = (turn == 0) ? acc[mi][ni][r] : (*p + acc[mi][ni][r]);
          }
    }
    __syncthreads();
  }
}

template <class Cfg> __device__ __forceinline__ void zero_acc(typename Cfg::M::AccV (&acc)[Cfg::MI][Cfg::NI]) {
#pragma unroll
  for (int mi = 0; mi < Cfg::MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < Cfg::NI; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0;
}

}  // namespace emb
