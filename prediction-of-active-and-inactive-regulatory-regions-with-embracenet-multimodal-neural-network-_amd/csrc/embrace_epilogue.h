// Shared epilogue of the fused EmbraceNet forward kernels: per element, pick the modality
// (idx = (double)cdf0[row] < u, u injected or Philox), add its bias, ReLU, write E and the code byte.
#pragma once
#include "gemm_core.h"
#include "philox.h"

namespace emb {

template <class Cfg>
__device__ __forceinline__ void embrace_epilogue(const typename Cfg::M::Acc* cs0, const typename Cfg::M::Acc* cs1,
                                                 const typename Cfg::M::Acc* __restrict__ b0,
                                                 const typename Cfg::M::Acc* __restrict__ b1, const float* __restrict__ cdf0,
                                                 const double* __restrict__ u, uint64_t seed, uint64_t step_val,
                                                 const uint64_t* __restrict__ step_dev, int64_t grow0,
                                                 typename Cfg::T* __restrict__ E, uint8_t* __restrict__ code, int B, int c,
                                                 int row0, int col0, bool vec_c) {
  using T = typename Cfg::T;
  using Acc = typename Cfg::M::Acc;
  const uint64_t stream = rng_stream(step_val + (step_dev ? *step_dev : 0), EMB_RNG_SELECT);
  constexpr int GROUPS = Cfg::BM * Cfg::BN / 4;
  for (int gidx = threadIdx.x; gidx < GROUPS; gidx += kThreads) {
    const int r = gidx / (Cfg::BN / 4), cq = (gidx % (Cfg::BN / 4)) * 4;
    const int row = row0 + r, col = col0 + cq;
    if (row >= B || col >= c) continue;
    const double thr = (double)cdf0[row];
    const long base = (long)row * c + col;
    const int nval = min(4, c - col);
    double uu[4];
    if (u != nullptr) {
      if (nval == 4 && vec_c) {
        const f64x2 a = *reinterpret_cast<const f64x2*>(u + base);
        const f64x2 b = *reinterpret_cast<const f64x2*>(u + base + 2);
        uu[0] = a[0]; uu[1] = a[1]; uu[2] = b[0]; uu[3] = b[1];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) uu[j] = j < nval ? u[base + j] : 0.0;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const Philox4 ph = philox4x32_10(seed, stream, (uint64_t)(grow0 + row) * (uint64_t)c + (uint64_t)(col + j));
        uu[j] = uniform53(ph.x, ph.y);
      }
    }
    T ev[4];
    uint8_t cv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool sel1 = thr < uu[j];   // first slot with cdf >= u (ATen binary search, M = 2)
      const int cc = min(col + j, c - 1);
      const Acc pre = sel1 ? cs1[r * Cfg::CS + cq + j] + b1[cc] : cs0[r * Cfg::CS + cq + j] + b0[cc];
      const bool act = pre > (Acc)0;
      ev[j] = (T)(act ? pre : (Acc)0);
      cv[j] = (uint8_t)((sel1 ? EMB_CODE_IDX : 0) | (act ? EMB_CODE_ACTIVE : 0));
    }
    if (nval == 4 && vec_c) {
      typedef T TV4 __attribute__((ext_vector_type(4)));
      TV4 o = {ev[0], ev[1], ev[2], ev[3]};
      *reinterpret_cast<TV4*>(E + base) = o;
      *reinterpret_cast<uint32_t*>(code + base) = (uint32_t)cv[0] | ((uint32_t)cv[1] << 8) | ((uint32_t)cv[2] << 16) | ((uint32_t)cv[3] << 24);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < nval) {
          E[base + j] = ev[j];
          code[base + j] = cv[j];
        }
    }
  }
}

}  // namespace emb
