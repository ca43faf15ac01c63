// EmbraceNet with bypass_docking=True (EmbraceNetMultimodal.py:54-55): the inputs ARE the docking outputs, so the layer is
// the modality selection alone (:63-88) -- an elementwise pass bound by HBM: 2 reads + 1 write of T and one code byte per
// element forward, 1 read of T + one code byte and up to 2 writes of T backward.  Selection arithmetic, RNG contract and
// code bits are those of the fused kernels (embrace_epilogue.h); there is no ReLU on this path, so ACTIVE is always set.
#include "embrace_epilogue.h"

namespace emb {

// one thread = G groups of four consecutive elements of one row (G = 2 for 2-byte types when c % 8 == 0, so that every load
// is 16 bytes per lane): per group one Philox call (or four injected uniforms), one 4-wide load per modality -- all loads are
// issued before the threshold arithmetic --, one 4-wide store of E and one 32-bit store of the code bytes
template <typename T, bool VEC4, int G>
__global__ __launch_bounds__(kThreads) void embrace_bypass_fwd_kernel(const T* __restrict__ X0, const T* __restrict__ X1,
                                                                      const SelArgs sel, const double* __restrict__ u,
                                                                      uint64_t seed, uint64_t step_val,
                                                                      const uint64_t* __restrict__ step_dev, int64_t grow0,
                                                                      T* __restrict__ E, uint8_t* __restrict__ code, int B, int c,
                                                                      int threads_per_row, long nthreads) {
  static_assert(G == 1 || VEC4, "several groups per thread only on the vector path");
  typedef T TV4 __attribute__((ext_vector_type(4)));
  const long t = (long)blockIdx.x * kThreads + threadIdx.x;
  if (t >= nthreads) return;
  const int row = (int)(t / threads_per_row), col0 = (int)(t % threads_per_row) * (4 * G);
  const long base0 = (long)row * c + col0;

  T a[G][4], b[G][4];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const long base = base0 + 4 * g;
    if (VEC4) {
      const TV4 va = *reinterpret_cast<const TV4*>(X0 + base), vb = *reinterpret_cast<const TV4*>(X1 + base);
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[g][j] = va[j]; b[g][j] = vb[j]; }
    } else {
      const int nval = min(4, c - col0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a[g][j] = j < nval ? X0[base + j] : (T)0;
        b[g][j] = j < nval ? X1[base + j] : (T)0;
      }
    }
  }

  const uint64_t step = step_val + (step_dev ? *step_dev : 0);
  double thr;
  if (sel.cdf0 != nullptr) {
    thr = (double)sel.cdf0[row];
  } else {
    bool ok;
    thr = (double)select_cdf(sel, row, seed, step, grow0, &ok);
    if (!ok && col0 == 0) atomicOr(sel.status, EMB_STATUS_INVALID_DISTRIBUTION);
  }
  const uint64_t t32 = select_threshold32((float)thr);
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int col = col0 + 4 * g;
    const long base = base0 + 4 * g;
    const int nval = VEC4 ? 4 : min(4, c - col);
    bool s1[4];
    if (u != nullptr) {                      // parity mode: the host generator's doubles (torch.multinomial, :84)
#pragma unroll
      for (int j = 0; j < 4; ++j) s1[j] = thr < (j < nval ? u[base + j] : 0.0);
    } else {
      uint32_t w4[4];
      select_words4(seed, rng_stream(step, EMB_RNG_SELECT), (uint64_t)(grow0 + row) * (uint64_t)c + (uint64_t)col, w4);
#pragma unroll
      for (int j = 0; j < 4; ++j) s1[j] = t32 < (uint64_t)w4[j];
    }
    T ev[4];
    uint8_t cv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ev[j] = s1[j] ? b[g][j] : a[g][j];     // x_m * 1 + x_other * 0 (:87-88)
      cv[j] = (uint8_t)(EMB_CODE_ACTIVE | (s1[j] ? (EMB_CODE_IDX | EMB_CODE_KEEP1) : EMB_CODE_KEEP0));
    }
    if (VEC4) {
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

// dX_m = dE * [idx == m]  (autograd of :87-88 with the docking step bypassed)
template <typename T, bool VEC4>
__global__ __launch_bounds__(kThreads) void embrace_bypass_bwd_kernel(const T* __restrict__ dE, const uint8_t* __restrict__ code,
                                                                      T* __restrict__ dX0, T* __restrict__ dX1, long n) {
  typedef T TV4 __attribute__((ext_vector_type(4)));
  const long e = ((long)blockIdx.x * kThreads + threadIdx.x) * 4;
  if (e >= n) return;
  if (VEC4) {
    const TV4 g = *reinterpret_cast<const TV4*>(dE + e);
    const uint32_t cw = *reinterpret_cast<const uint32_t*>(code + e);
    TV4 g0, g1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool s1 = (cw >> (8 * j)) & EMB_CODE_IDX;
      g0[j] = s1 ? (T)0 : g[j];
      g1[j] = s1 ? g[j] : (T)0;
    }
    if (dX0) *reinterpret_cast<TV4*>(dX0 + e) = g0;
    if (dX1) *reinterpret_cast<TV4*>(dX1 + e) = g1;
  } else {
    for (int j = 0; j < 4 && e + j < n; ++j) {
      const bool s1 = code[e + j] & EMB_CODE_IDX;
      const T g = dE[e + j];
      if (dX0) dX0[e + j] = s1 ? (T)0 : g;
      if (dX1) dX1[e + j] = s1 ? g : (T)0;
    }
  }
}

template <typename T> static bool aligned_v4(const void* p) { return (reinterpret_cast<uintptr_t>(p) % (4 * sizeof(T))) == 0; }

template <typename T> static int bypass_fwd(const void* X0, const void* X1, const SelArgs& sel, const double* u, uint64_t seed,
                                            uint64_t step_val, const uint64_t* step_dev, int64_t row0, void* E, uint8_t* code,
                                            int B, int c, hipStream_t s) {
  const bool vec = (c % 4 == 0) && aligned_v4<T>(X0) && aligned_v4<T>(X1) && aligned_v4<T>(E) &&
                   ((reinterpret_cast<uintptr_t>(code) & 3u) == 0);
  const bool two = vec && sizeof(T) == 2 && (c % 8 == 0) && aligned16(X0) && aligned16(X1) && aligned16(E);
  const int tpr = two ? c / 8 : cdiv(c, 4);
  const long nthreads = (long)B * tpr;
  const long nblk = (nthreads + kThreads - 1) / kThreads;
  EMB_CHECK_ARG(nblk <= 0x7fffffffL, "emb_embrace_bypass_fwd: B*c too large for one launch");
#define EMB_BYPASS_LAUNCH(V, GG)                                                                                              \
  embrace_bypass_fwd_kernel<T, V, GG><<<(unsigned)nblk, kThreads, 0, s>>>((const T*)X0, (const T*)X1, sel, u, seed, step_val, \
                                                                          step_dev, row0, (T*)E, code, B, c, tpr, nthreads)
  if (two) EMB_BYPASS_LAUNCH(true, 2);
  else if (vec) EMB_BYPASS_LAUNCH(true, 1);
  else EMB_BYPASS_LAUNCH(false, 1);
#undef EMB_BYPASS_LAUNCH
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

template <typename T> static int bypass_bwd(const void* dE, const uint8_t* code, void* dX0, void* dX1, long n, hipStream_t s) {
  const long nblk = ((n + 3) / 4 + kThreads - 1) / kThreads;
  EMB_CHECK_ARG(nblk <= 0x7fffffffL, "emb_embrace_bypass_bwd: B*c too large for one launch");
  const bool vec = (n % 4 == 0) && aligned_v4<T>(dE) && (!dX0 || aligned_v4<T>(dX0)) && (!dX1 || aligned_v4<T>(dX1)) &&
                   ((reinterpret_cast<uintptr_t>(code) & 3u) == 0);
  if (vec)
    embrace_bypass_bwd_kernel<T, true><<<(unsigned)nblk, kThreads, 0, s>>>((const T*)dE, code, (T*)dX0, (T*)dX1, n);
  else
    embrace_bypass_bwd_kernel<T, false><<<(unsigned)nblk, kThreads, 0, s>>>((const T*)dE, code, (T*)dX0, (T*)dX1, n);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

// ------------------------------------------------------------------------------ any number of modalities (M <= 8)
// EmbraceNet.forward with len(input_list) != 2 (EmbraceNetMultimodal.py:47-48): the docking layers run as M calls of
// emb_linear_fwd (Linear + ReLU), the selection over their outputs is the pass below.  code = idx (0 .. M-1).
constexpr int kMaxModalities = 8;
struct ModPtrs {
  void* d[kMaxModalities];
};

// :63-76 and the cdf torch.multinomial builds from the row (ATen: running fp32 sum, then cum /= sum), any M
__global__ __launch_bounds__(kThreads) void select_prep_m_kernel(const float* __restrict__ p, int p_rows,
                                                                 const float* __restrict__ avail, float* __restrict__ cdf,
                                                                 int32_t* __restrict__ status, int B, int M) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= B) return;
  const float* pr = p + (p_rows == 1 ? 0 : (long)M * row);
  const float* ar = avail ? avail + (long)M * row : nullptr;
  float* out = cdf + (long)M * row;
  float sm = 0.0f;
  for (int m = 0; m < M; ++m) sm = __fadd_rn(sm, __fmul_rn(pr[m], ar ? ar[m] : 1.0f));   // :73, :75
  bool ok = true;
  float run = 0.0f;
  for (int m = 0; m < M; ++m) {
    const float n = __fdiv_rn(__fmul_rn(pr[m], ar ? ar[m] : 1.0f), sm);                    // :76
    ok = ok && (n >= 0.0f) && isfinite(n);
    run = __fadd_rn(run, n);
    out[m] = run;
  }
  ok = ok && (run > 0.0f);
  for (int m = 0; m < M; ++m) out[m] = ok ? __fdiv_rn(out[m], run) : __builtin_nanf("");
  if (!ok) atomicOr(status, EMB_STATUS_INVALID_DISTRIBUTION);
}

template <typename T, int M, bool VEC4>
__global__ __launch_bounds__(kThreads) void embrace_select_fwd_kernel(const ModPtrs D, const float* __restrict__ cdf,
                                                                      const double* __restrict__ u, uint64_t seed,
                                                                      uint64_t step_val, const uint64_t* __restrict__ step_dev,
                                                                      int64_t grow0, T* __restrict__ E, uint8_t* __restrict__ code,
                                                                      int c, int groups_per_row, long ngroups) {
  typedef T TV4 __attribute__((ext_vector_type(4)));
  const long g = (long)blockIdx.x * kThreads + threadIdx.x;
  if (g >= ngroups) return;
  const int row = (int)(g / groups_per_row), col = (int)(g % groups_per_row) * 4;
  const long base = (long)row * c + col;
  const int nval = VEC4 ? 4 : min(4, c - col);
  double uu[4];
  if (u != nullptr) {
#pragma unroll
    for (int j = 0; j < 4; ++j) uu[j] = j < nval ? u[base + j] : 0.0;
  } else {
    const uint64_t step = step_val + (step_dev ? *step_dev : 0);
    uint32_t w4[4];
    select_words4(seed, rng_stream(step, EMB_RNG_SELECT), (uint64_t)(grow0 + row) * (uint64_t)c + (uint64_t)col, w4);
#pragma unroll
    for (int j = 0; j < 4; ++j) uu[j] = (double)w4[j] * (1.0 / 4294967296.0);
  }
  // first slot with cdf >= u (ATen's binary search) = number of entries strictly below u; the last bin takes the rest
  int idx[4] = {0, 0, 0, 0};
#pragma unroll
  for (int m = 0; m < M - 1; ++m) {
    const double t = (double)cdf[(long)M * row + m];
#pragma unroll
    for (int j = 0; j < 4; ++j) idx[j] += (t < uu[j]) ? 1 : 0;
  }
  T ev[4] = {(T)0, (T)0, (T)0, (T)0};
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const T* X = (const T*)D.d[m];
    if (VEC4) {
      const TV4 v = *reinterpret_cast<const TV4*>(X + base);
#pragma unroll
      for (int j = 0; j < 4; ++j) ev[j] = idx[j] == m ? v[j] : ev[j];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < nval && idx[j] == m) ev[j] = X[base + j];
    }
  }
  if (VEC4) {
    TV4 o = {ev[0], ev[1], ev[2], ev[3]};
    *reinterpret_cast<TV4*>(E + base) = o;
    *reinterpret_cast<uint32_t*>(code + base) = (uint32_t)idx[0] | ((uint32_t)idx[1] << 8) | ((uint32_t)idx[2] << 16) | ((uint32_t)idx[3] << 24);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < nval) {
        E[base + j] = ev[j];
        code[base + j] = (uint8_t)idx[j];
      }
  }
}

template <typename T, int M, bool VEC4>
__global__ __launch_bounds__(kThreads) void embrace_select_bwd_kernel(const T* __restrict__ dE, const uint8_t* __restrict__ code,
                                                                      const ModPtrs dD, long n) {
  typedef T TV4 __attribute__((ext_vector_type(4)));
  const long e = ((long)blockIdx.x * kThreads + threadIdx.x) * 4;
  if (e >= n) return;
  if (VEC4) {
    const TV4 g = *reinterpret_cast<const TV4*>(dE + e);
    const uint32_t cw = *reinterpret_cast<const uint32_t*>(code + e);
#pragma unroll
    for (int m = 0; m < M; ++m) {
      T* o = (T*)dD.d[m];
      if (o == nullptr) continue;
      TV4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (int)((cw >> (8 * j)) & 0xff) == m ? g[j] : (T)0;
      *reinterpret_cast<TV4*>(o + e) = v;
    }
  } else {
    for (int j = 0; j < 4 && e + j < n; ++j) {
      const int id = code[e + j];
      const T g = dE[e + j];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        T* o = (T*)dD.d[m];
        if (o != nullptr) o[e + j] = id == m ? g : (T)0;
      }
    }
  }
}

template <typename T, int M> static int select_fwd_m(const ModPtrs& D, const float* cdf, const double* u, uint64_t seed,
                                                     uint64_t step_val, const uint64_t* step_dev, int64_t row0, void* E,
                                                     uint8_t* code, int B, int c, hipStream_t s) {
  const int gpr = cdiv(c, 4);
  const long ngroups = (long)B * gpr;
  const long nblk = (ngroups + kThreads - 1) / kThreads;
  EMB_CHECK_ARG(nblk <= 0x7fffffffL, "emb_embrace_select_fwd: B*c too large for one launch");
  bool vec = (c % 4 == 0) && aligned_v4<T>(E) && ((reinterpret_cast<uintptr_t>(code) & 3u) == 0);
  for (int m = 0; m < M; ++m) vec = vec && aligned_v4<T>(D.d[m]);
  if (vec)
    embrace_select_fwd_kernel<T, M, true><<<(unsigned)nblk, kThreads, 0, s>>>(D, cdf, u, seed, step_val, step_dev, row0, (T*)E, code, c, gpr, ngroups);
  else
    embrace_select_fwd_kernel<T, M, false><<<(unsigned)nblk, kThreads, 0, s>>>(D, cdf, u, seed, step_val, step_dev, row0, (T*)E, code, c, gpr, ngroups);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

template <typename T, int M> static int select_bwd_m(const void* dE, const uint8_t* code, const ModPtrs& dD, long n, hipStream_t s) {
  const long nblk = ((n + 3) / 4 + kThreads - 1) / kThreads;
  EMB_CHECK_ARG(nblk <= 0x7fffffffL, "emb_embrace_select_bwd: B*c too large for one launch");
  bool vec = (n % 4 == 0) && aligned_v4<T>(dE) && ((reinterpret_cast<uintptr_t>(code) & 3u) == 0);
  for (int m = 0; m < M; ++m) vec = vec && (!dD.d[m] || aligned_v4<T>(dD.d[m]));
  if (vec)
    embrace_select_bwd_kernel<T, M, true><<<(unsigned)nblk, kThreads, 0, s>>>((const T*)dE, code, dD, n);
  else
    embrace_select_bwd_kernel<T, M, false><<<(unsigned)nblk, kThreads, 0, s>>>((const T*)dE, code, dD, n);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

#define EMB_FOR_M(CALL)                       \
  switch (M) {                                \
    case 1: return CALL(1);                   \
    case 2: return CALL(2);                   \
    case 3: return CALL(3);                   \
    case 4: return CALL(4);                   \
    case 5: return CALL(5);                   \
    case 6: return CALL(6);                   \
    case 7: return CALL(7);                   \
    default: return CALL(8);                  \
  }

template <typename T> static int select_fwd(int M, const ModPtrs& D, const float* cdf, const double* u, uint64_t seed,
                                            uint64_t step_val, const uint64_t* step_dev, int64_t row0, void* E, uint8_t* code,
                                            int B, int c, hipStream_t s) {
#define EMB_CALL(MM) select_fwd_m<T, MM>(D, cdf, u, seed, step_val, step_dev, row0, E, code, B, c, s)
  EMB_FOR_M(EMB_CALL)
#undef EMB_CALL
}
template <typename T> static int select_bwd(int M, const void* dE, const uint8_t* code, const ModPtrs& dD, long n, hipStream_t s) {
#define EMB_CALL(MM) select_bwd_m<T, MM>(dE, code, dD, n, s)
  EMB_FOR_M(EMB_CALL)
#undef EMB_CALL
}

}  // namespace emb

extern "C" int emb_embrace_bypass_fwd(const void* X0, const void* X1, const float* cdf0, const float* p, int p_rows,
                                      const float* avail, int device_dropout, int32_t* status, const double* u, uint64_t seed,
                                      uint64_t step_val, const uint64_t* step_dev, int64_t row0, void* E, uint8_t* code, int B,
                                      int c, int dtype, emb_stream_t stream) {
  EMB_CHECK_ARG(X0 && X1 && E && code, "emb_embrace_bypass_fwd: null pointer");
  EMB_CHECK_ARG(cdf0 || (p && status), "emb_embrace_bypass_fwd: either cdf0 or (p, status) must be given");
  EMB_CHECK_ARG(B >= 0 && c > 0, "emb_embrace_bypass_fwd: bad dims B=%d c=%d", B, c);
  EMB_CHECK_ARG(cdf0 || p_rows == 1 || p_rows == B, "emb_embrace_bypass_fwd: p_rows must be 1 or B (got %d, B=%d)", p_rows, B);
  if (B == 0) return EMB_OK;
  const emb::SelArgs sel = cdf0 ? emb::SelArgs{cdf0, nullptr, nullptr, nullptr, 0, 0}
                                : emb::SelArgs{nullptr, p, avail, status, p_rows, device_dropout};
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case EMB_F32: return emb::bypass_fwd<float>(X0, X1, sel, u, seed, step_val, step_dev, row0, E, code, B, c, s);
    case EMB_BF16: return emb::bypass_fwd<__bf16>(X0, X1, sel, u, seed, step_val, step_dev, row0, E, code, B, c, s);
    case EMB_F64: return emb::bypass_fwd<double>(X0, X1, sel, u, seed, step_val, step_dev, row0, E, code, B, c, s);
  }
  emb::set_error("emb_embrace_bypass_fwd: unsupported dtype %d", dtype);
  return EMB_ERR_DTYPE;
}

extern "C" int emb_embrace_bypass_bwd(const void* dE, const uint8_t* code, void* dX0, void* dX1, int B, int c, int dtype,
                                      emb_stream_t stream) {
  EMB_CHECK_ARG(dE && code, "emb_embrace_bypass_bwd: null pointer");
  EMB_CHECK_ARG(B >= 0 && c > 0, "emb_embrace_bypass_bwd: bad dims B=%d c=%d", B, c);
  if (B == 0 || (!dX0 && !dX1)) return EMB_OK;
  const long n = (long)B * c;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case EMB_F32: return emb::bypass_bwd<float>(dE, code, dX0, dX1, n, s);
    case EMB_BF16: return emb::bypass_bwd<__bf16>(dE, code, dX0, dX1, n, s);
    case EMB_F64: return emb::bypass_bwd<double>(dE, code, dX0, dX1, n, s);
  }
  emb::set_error("emb_embrace_bypass_bwd: unsupported dtype %d", dtype);
  return EMB_ERR_DTYPE;
}

extern "C" int emb_select_prep_m(const float* p, int p_rows, const float* avail, float* cdf, int32_t* status, int B, int M,
                                 emb_stream_t stream) {
  EMB_CHECK_ARG(p && cdf && status, "emb_select_prep_m: null pointer");
  EMB_CHECK_ARG(B >= 0 && M >= 1 && M <= emb::kMaxModalities, "emb_select_prep_m: bad dims B=%d M=%d (M <= 8)", B, M);
  EMB_CHECK_ARG(p_rows == 1 || p_rows == B, "emb_select_prep_m: p_rows must be 1 or B (got %d, B=%d)", p_rows, B);
  if (B == 0) return EMB_OK;
  emb::select_prep_m_kernel<<<emb::cdiv(B, emb::kThreads), emb::kThreads, 0, (hipStream_t)stream>>>(p, p_rows, avail, cdf, status, B, M);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

extern "C" int emb_embrace_select_fwd(const void* const* D, int M, const float* cdf, const double* u, uint64_t seed,
                                      uint64_t step_val, const uint64_t* step_dev, int64_t row0, void* E, uint8_t* code, int B,
                                      int c, int dtype, emb_stream_t stream) {
  EMB_CHECK_ARG(D && cdf && E && code, "emb_embrace_select_fwd: null pointer");
  EMB_CHECK_ARG(B >= 0 && c > 0 && M >= 1 && M <= emb::kMaxModalities, "emb_embrace_select_fwd: bad dims B=%d c=%d M=%d (M <= 8)", B, c, M);
  emb::ModPtrs P{};
  for (int m = 0; m < M; ++m) {
    EMB_CHECK_ARG(D[m], "emb_embrace_select_fwd: null modality pointer %d", m);
    P.d[m] = const_cast<void*>(D[m]);
  }
  if (B == 0) return EMB_OK;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case EMB_F32: return emb::select_fwd<float>(M, P, cdf, u, seed, step_val, step_dev, row0, E, code, B, c, s);
    case EMB_BF16: return emb::select_fwd<__bf16>(M, P, cdf, u, seed, step_val, step_dev, row0, E, code, B, c, s);
    case EMB_F64: return emb::select_fwd<double>(M, P, cdf, u, seed, step_val, step_dev, row0, E, code, B, c, s);
  }
  emb::set_error("emb_embrace_select_fwd: unsupported dtype %d", dtype);
  return EMB_ERR_DTYPE;
}

extern "C" int emb_embrace_select_bwd(const void* dE, const uint8_t* code, void* const* dD, int M, int B, int c, int dtype,
                                      emb_stream_t stream) {
  EMB_CHECK_ARG(dE && code && dD, "emb_embrace_select_bwd: null pointer");
  EMB_CHECK_ARG(B >= 0 && c > 0 && M >= 1 && M <= emb::kMaxModalities, "emb_embrace_select_bwd: bad dims B=%d c=%d M=%d (M <= 8)", B, c, M);
  emb::ModPtrs P{};
  bool any = false;
  for (int m = 0; m < M; ++m) {
    P.d[m] = dD[m];
    any = any || dD[m] != nullptr;
  }
  if (B == 0 || !any) return EMB_OK;
  const long n = (long)B * c;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case EMB_F32: return emb::select_bwd<float>(M, dE, code, P, n, s);
    case EMB_BF16: return emb::select_bwd<__bf16>(M, dE, code, P, n, s);
    case EMB_F64: return emb::select_bwd<double>(M, dE, code, P, n, s);
  }
  emb::set_error("emb_embrace_select_bwd: unsupported dtype %d", dtype);
  return EMB_ERR_DTYPE;
}
