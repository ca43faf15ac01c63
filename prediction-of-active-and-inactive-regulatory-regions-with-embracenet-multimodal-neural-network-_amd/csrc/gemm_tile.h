// One output tile of a generic GEMM built on gemm_core.h, with a cooperative row-major epilogue.
// Used by the backward contractions (dgrad / wgrad of the docking and post layers) and by the
// post-stack forward.
#pragma once
#include "gemm_core.h"
#include "philox.h"

namespace emb {

template <typename T> struct GemmOperand {
  const T* ptr;
  const uint8_t* code;  // side array for the operand transform (same indexing as ptr) or nullptr
  int ld;
  bool vec_ok;
};

// Epilogues receive (acc value, global row, global col) for in-range elements, 4 columns at a time.
// --- plain store: C[row*ldc + col] = (OutT)v ; optional "extra column" at col == N -> vec[row]
template <typename OutT> struct EpiStore {
  OutT* C;
  long ldc;
  OutT* extra;      // receives the virtual column `ncols` (bias gradient), may be nullptr
  int ncols;        // real columns
  bool vec_ok;      // 4-wide stores legal
  template <typename Acc> __device__ void operator()(const Acc (&v)[4], int row, int col, int) const {
    const int real = min(4, ncols - col);   // real columns in this group of 4
    if (real <= 0) return;
    OutT* dst = C + (long)row * ldc + col;
    if (real == 4 && vec_ok) {
      typedef OutT OV4 __attribute__((ext_vector_type(4)));
      OV4 o = {(OutT)v[0], (OutT)v[1], (OutT)v[2], (OutT)v[3]};
      *reinterpret_cast<OV4*>(dst) = o;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)   // static indices only: a runtime-indexed v[] would live in scratch
        if (j < real) dst[j] = (OutT)v[j];
    }
  }
  template <typename Acc> __device__ void extra_col(Acc v, int row) const {
    if (extra != nullptr) extra[row] = (OutT)v;
  }
};

// --- linear forward: Y = dropout(relu(v + b[col])), mask byte bit0 = pre > 0, bit1 = kept
template <typename T, typename P> struct EpiLinear {
  T* Y;
  uint8_t* mask;   // nullable
  const P* bias;
  long ldc;
  int ncols;
  bool relu;
  float keep_scale;   // 1/(1-p) or 1
  float drop_p;       // 0 -> no dropout
  uint64_t seed, stream;
  int64_t grow0;
  bool vec_ok;
  template <typename Acc> __device__ void operator()(const Acc (&v)[4], int row, int col, int nval) const {
    T out[4];
    uint8_t mk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cc = min(col + j, ncols - 1);
      Acc pre = v[j] + (Acc)bias[cc];
      bool act = true, keep = true;
      if (relu) {
        act = pre > (Acc)0;
        pre = act ? pre : (Acc)0;
      }
      if (drop_p > 0.0f) {
        const Philox4 ph = philox4x32_10(seed, stream, (uint64_t)(grow0 + row) * (uint64_t)ncols + (uint64_t)(col + j));
        keep = uniform24(ph.x) >= drop_p;
        pre = keep ? pre * (Acc)keep_scale : (Acc)0;
      }
      out[j] = (T)pre;
      mk[j] = (uint8_t)((act ? 1 : 0) | (keep ? 2 : 0));
    }
    const long base = (long)row * ldc + col;
    if (nval == 4 && vec_ok) {
      typedef T TV4 __attribute__((ext_vector_type(4)));
      TV4 o = {out[0], out[1], out[2], out[3]};
      *reinterpret_cast<TV4*>(Y + base) = o;
      if (mask) *reinterpret_cast<uint32_t*>(mask + base) = (uint32_t)mk[0] | ((uint32_t)mk[1] << 8) | ((uint32_t)mk[2] << 16) | ((uint32_t)mk[3] << 24);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < nval) {
          Y[base + j] = out[j];
          if (mask) mask[base + j] = mk[j];
        }
    }
  }
  template <typename Acc> __device__ void extra_col(Acc, int) const {}
};

// Compute tile (tm, tn) of  C[M, Ncols] = A . B^T  and hand it to `epi`.
//   ones_row_b >= 0 appends a virtual all-ones row to operand B (=> one extra output column).
template <class Cfg, class XfA, class Epi>
__device__ __forceinline__ void gemm_tile(const GemmOperand<typename Cfg::T>& A, const GemmOperand<typename Cfg::T>& Bm,
                                          int M, int N, int K, int tm, int tn, XfA xfa, int ones_row_b, const Epi& epi,
                                          char* arena, int k_begin = 0) {
  using T = typename Cfg::T;
  using Mm = typename Cfg::M;
  using Acc = typename Mm::Acc;
  const int row0 = tm * Cfg::BM, col0 = tn * Cfg::BN;
  typename Mm::AccV acc[Cfg::MI][Cfg::NI];
  zero_acc<Cfg>(acc);
  Stager<T, Cfg::AKM, Cfg::BM, Cfg::BK, XfA> sa{A.ptr, A.code, A.ld, row0, M, K, A.vec_ok, xfa, -1};
  Stager<T, Cfg::BKM, Cfg::BN, Cfg::BK, XfNone> sb{Bm.ptr, nullptr, Bm.ld, col0, N, K, Bm.vec_ok, XfNone{}, ones_row_b};
  gemm_mainloop<Cfg>(sa, sb, K, arena, acc, k_begin);   // reduction range [k_begin, K)
  Acc* cs = reinterpret_cast<Acc*>(arena);
  reduce_to_slab<Cfg>(acc, cs);
  const int ncols_total = N + (ones_row_b >= 0 ? 1 : 0);
  constexpr int GROUPS = Cfg::BM * Cfg::BN / 4;
  for (int gidx = threadIdx.x; gidx < GROUPS; gidx += kThreads) {
    const int r = gidx / (Cfg::BN / 4), cq = (gidx % (Cfg::BN / 4)) * 4;
    const int row = row0 + r, col = col0 + cq;
    if (row >= M || col >= ncols_total) continue;
    Acc v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = cs[r * Cfg::CS + cq + j];
    epi(v, row, col, min(4, ncols_total - col));
  }
  // the virtual ones-column (global column N) carries the bias gradient: read it straight from the slab
  if (ones_row_b >= 0 && N >= col0 && N < col0 + Cfg::BN) {
    for (int r = threadIdx.x; r < Cfg::BM; r += kThreads)
      if (row0 + r < M) epi.extra_col(cs[r * Cfg::CS + (N - col0)], row0 + r);
  }
  __syncthreads();   // arena is reused by the next tile / GEMM of this block
}

template <class Cfg> constexpr int gemm_tile_lds() {
  constexpr int slab = Cfg::SLAB * (int)sizeof(typename Cfg::M::Acc);
  return Cfg::OPERAND_BYTES > slab ? Cfg::OPERAND_BYTES : slab;
}

}  // namespace emb
