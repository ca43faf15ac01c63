// Shared epilogue of the fused EmbraceNet forward kernels: per element, pick the modality
// (idx = (double)cdf0[row] < u, u injected or Philox), add its bias, ReLU, write E and the code byte.
#pragma once
#include "gemm_core.h"
#include "philox.h"

namespace emb {

// Where the selection threshold of a row comes from: a cdf0[B] vector made earlier (emb_select_prep; the host-replay parity
// mode needs it on its own), or -- cdf0 == NULL -- computed here from the selection probabilities, so that the forward pass
// is ONE launch (emb_embrace_fwd_select).  Same arithmetic either way (select_cdf).
struct SelArgs {
  const float* cdf0;
  const float* p;        // [p_rows][2]
  const float* avail;    // [B][2] or NULL
  int32_t* status;
  int p_rows, device_dropout;
};

// EmbraceNetMultimodal.py:63-76, :178-184 and the cdf torch.multinomial builds from the row.
__device__ __forceinline__ float select_cdf(const SelArgs& s, int row, uint64_t seed, uint64_t step, int64_t grow0, bool* ok) {
  const float* pr = s.p + (s.p_rows == 1 ? 0 : 2 * (long)row);
  float a0 = 1.0f, a1 = 1.0f;
  if (s.device_dropout) {
    const float gate = uniform24(philox4x32_10(seed, rng_stream(step, EMB_RNG_GATE), 0).x);
    if (gate >= 0.5f) {   // EmbraceNetMultimodal.py:180
      const float t = uniform24(philox4x32_10(seed, rng_stream(step, EMB_RNG_ROWMOD), (uint64_t)(grow0 + row)).x);
      const bool m1 = t > 0.5f;   // torch.round: half-to-even, 0.5 -> 0   (:181)
      a0 = m1 ? 0.0f : 1.0f;
      a1 = m1 ? 1.0f : 0.0f;
    }
  } else if (s.avail != nullptr) {
    a0 = s.avail[2 * (long)row];
    a1 = s.avail[2 * (long)row + 1];
  }
  // every operation below is a separately rounded fp32 op, as in the reference's ATen calls
  const float q0 = __fmul_rn(pr[0], a0), q1 = __fmul_rn(pr[1], a1);   // :73
  const float sm = __fadd_rn(q0, q1);                                  // :75
  const float n0 = __fdiv_rn(q0, sm), n1 = __fdiv_rn(q1, sm);          // :76
  // torch.multinomial (ATen CPU kernel): running sum, then cum /= sum
  const float tot = __fadd_rn(n0, n1);
  const float cdf = __fdiv_rn(n0, tot);
  *ok = (n0 >= 0.0f) && (n1 >= 0.0f) && isfinite(n0) && isfinite(n1) && (tot > 0.0f);
  return *ok ? cdf : __builtin_nanf("");
}

// Perf-mode selection uniforms (RNG kind 0): element e = global_row * c + j takes word (e & 3) of Philox(counter = e >> 2),
// u = word * 2^-32.  One Philox call serves four consecutive elements (it was one call per element, 20 64-bit multiplies
// each: the dominant cost of the epilogue); (double)cdf0 < u is evaluated exactly as the integer compare T < word with
// T = floor(cdf0 * 2^32) (select_threshold32).
__device__ __forceinline__ uint32_t philox_word(const Philox4& p, unsigned k) {
  return k == 0 ? p.x : (k == 1 ? p.y : (k == 2 ? p.z : p.w));
}
__device__ __forceinline__ void select_words4(uint64_t seed, uint64_t stream, uint64_t e0, uint32_t (&w)[4]) {
  const Philox4 p0 = philox4x32_10(seed, stream, e0 >> 2);
  const unsigned k0 = (unsigned)(e0 & 3);
  if (k0 == 0) {
    w[0] = p0.x; w[1] = p0.y; w[2] = p0.z; w[3] = p0.w;
  } else {            // four elements straddle two counters (c not a multiple of 4)
    const Philox4 p1 = philox4x32_10(seed, stream, (e0 >> 2) + 1);
#pragma unroll
    for (unsigned j = 0; j < 4; ++j) w[j] = (k0 + j < 4) ? philox_word(p0, k0 + j) : philox_word(p1, k0 + j - 4);
  }
}
__device__ __forceinline__ uint64_t select_threshold32(float cdf0) {
  if (!(cdf0 == cdf0)) return 1ull << 32;       // invalid row: "thr < u" is false for every u
  const double t = (double)cdf0 * 4294967296.0;
  return t >= 4294967296.0 ? (1ull << 32) : (t <= 0.0 ? 0ull : (uint64_t)t);
}

template <class Cfg>
__device__ __forceinline__ void embrace_epilogue(const typename Cfg::M::Acc* cs0, const typename Cfg::M::Acc* cs1,
                                                 const typename Cfg::M::Acc* __restrict__ b0,
                                                 const typename Cfg::M::Acc* __restrict__ b1, const SelArgs sel,
                                                 const double* __restrict__ u, uint64_t seed, uint64_t step_val,
                                                 const uint64_t* __restrict__ step_dev, int64_t grow0,
                                                 typename Cfg::T* __restrict__ E, uint8_t* __restrict__ code, int B, int c,
                                                 int row0, int col0, bool vec_c) {
  using T = typename Cfg::T;
  using Acc = typename Cfg::M::Acc;
  const uint64_t step = step_val + (step_dev ? *step_dev : 0);
  const uint64_t stream = rng_stream(step, EMB_RNG_SELECT);
  constexpr int GROUPS = Cfg::BM * Cfg::BN / 4;
  for (int gidx = threadIdx.x; gidx < GROUPS; gidx += kThreads) {
    const int r = gidx / (Cfg::BN / 4), cq = (gidx % (Cfg::BN / 4)) * 4;
    const int row = row0 + r, col = col0 + cq;
    if (row >= B || col >= c) continue;
    double thr;
    if (sel.cdf0 != nullptr) {
      thr = (double)sel.cdf0[row];
    } else {
      bool ok;
      thr = (double)select_cdf(sel, row, seed, step, grow0, &ok);
      if (!ok && col == 0) atomicOr(sel.status, EMB_STATUS_INVALID_DISTRIBUTION);   // (one column group per row reports)
    }
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
      uint32_t w4[4];
      select_words4(seed, stream, (uint64_t)(grow0 + row) * (uint64_t)c + (uint64_t)col, w4);
#pragma unroll
      for (int j = 0; j < 4; ++j) uu[j] = (double)w4[j] * (1.0 / 4294967296.0);   // exact: thr < uu  <=>  floor(thr * 2^32) < word
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
      cv[j] = (uint8_t)((sel1 ? EMB_CODE_IDX : 0) | (act ? (EMB_CODE_ACTIVE | (sel1 ? EMB_CODE_KEEP1 : EMB_CODE_KEEP0)) : 0));
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
