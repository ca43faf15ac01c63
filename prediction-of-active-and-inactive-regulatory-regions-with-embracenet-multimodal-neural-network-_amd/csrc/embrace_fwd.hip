// Fused EmbraceNet forward: docking GEMMs (both modalities) -> bias -> ReLU -> per-element modality
// selection -> fused output.  Replaces EmbraceNetMultimodal.py:52-60 and :80-88 of the reference; the
// [B,c,M] stack / one-hot / product temporaries of the reference never exist here.
//
// One workgroup owns a BM x BN tile of E.  It runs the K loop twice (K = d0 with X0/W0, K = d1 with
// X1/W1), keeps both accumulator tiles, parks them in LDS and finishes with a row-major cooperative
// epilogue: idx = (double)cdf0[row] < u[row][col] (u injected, or Philox keyed on the global element
// index), pre = acc_idx + b_idx[col], E = max(pre, 0), code = idx | (pre > 0) << 1.
#include "embrace_epilogue.h"
#include "embrace_stream.h"
#include "embrace_split.h"
#include <cstdlib>
#include <cstring>

namespace emb {

template <class Cfg>
__global__ __launch_bounds__(kThreads, 2) void embrace_fwd_kernel(
    const typename Cfg::T* __restrict__ X0, const typename Cfg::T* __restrict__ X1,
    const typename Cfg::T* __restrict__ W0, const typename Cfg::T* __restrict__ W1,
    const typename Cfg::M::Acc* __restrict__ b0, const typename Cfg::M::Acc* __restrict__ b1,
    const SelArgs sel, const double* __restrict__ u, uint64_t seed, uint64_t step_val,
    const uint64_t* __restrict__ step_dev, int64_t grow0, typename Cfg::T* __restrict__ E,
    uint8_t* __restrict__ code, int B, int d0, int d1, int c, int tiles_n, int ntiles, bool vec0, bool vec1,
    bool vec_c) {
  using T = typename Cfg::T;
  using M = typename Cfg::M;
  using Acc = typename M::Acc;
  extern __shared__ __attribute__((aligned(16))) char arena[];

  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int row0 = tm * Cfg::BM, col0 = tn * Cfg::BN;

  typename M::AccV acc0[Cfg::MI][Cfg::NI], acc1[Cfg::MI][Cfg::NI];
  zero_acc<Cfg>(acc0);
  zero_acc<Cfg>(acc1);
  {
    Stager<T, false, Cfg::BM, Cfg::BK, XfNone> sa{X0, nullptr, d0, row0, B, d0, vec0, XfNone{}, -1};
    Stager<T, false, Cfg::BN, Cfg::BK, XfNone> sb{W0, nullptr, d0, col0, c, d0, vec0, XfNone{}, -1};
    gemm_mainloop<Cfg>(sa, sb, d0, arena, acc0);
  }
  {
    Stager<T, false, Cfg::BM, Cfg::BK, XfNone> sa{X1, nullptr, d1, row0, B, d1, vec1, XfNone{}, -1};
    Stager<T, false, Cfg::BN, Cfg::BK, XfNone> sb{W1, nullptr, d1, col0, c, d1, vec1, XfNone{}, -1};
    gemm_mainloop<Cfg>(sa, sb, d1, arena, acc1);
  }
  Acc* cs0 = reinterpret_cast<Acc*>(arena);
  Acc* cs1 = cs0 + Cfg::SLAB;
  reduce_to_slab<Cfg>(acc0, cs0);
  reduce_to_slab<Cfg>(acc1, cs1);

  embrace_epilogue<Cfg>(cs0, cs1, b0, b1, sel, u, seed, step_val, step_dev, grow0, E, code, B, c, row0, col0, vec_c);
}

// the selection cdf as a vector of its own (embrace_epilogue.h: select_cdf)
__global__ __launch_bounds__(kThreads) void select_prep_kernel(const float* __restrict__ p, int p_rows,
                                                               const float* __restrict__ avail, int device_dropout,
                                                               uint64_t seed, uint64_t step_val,
                                                               const uint64_t* __restrict__ step_dev, int64_t grow0,
                                                               float* __restrict__ cdf0, int32_t* __restrict__ status,
                                                               int B) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= B) return;
  const SelArgs sel{nullptr, p, avail, status, p_rows, device_dropout};
  bool ok;
  const float cdf = select_cdf(sel, row, seed, step_val + (step_dev ? *step_dev : 0), grow0, &ok);
  if (!ok) atomicOr(status, EMB_STATUS_INVALID_DISTRIBUTION);
  cdf0[row] = cdf;
}

// ----------------------------------------------------------------------------------- dispatch
template <class Cfg> static int launch_fwd(const void* X0, const void* X1, const void* W0, const void* b0, const void* W1,
                                           const void* b1, const SelArgs& sel, const double* u, uint64_t seed,
                                           uint64_t step_val, const uint64_t* step_dev, int64_t row0, void* E, uint8_t* code,
                                           int B, int d0, int d1, int c, hipStream_t s) {
  using T = typename Cfg::T;
  using Acc = typename Cfg::M::Acc;
  constexpr int VEC = Elem<T>::VEC;
  const int tiles_m = cdiv(B, Cfg::BM), tiles_n = cdiv(c, Cfg::BN), ntiles = tiles_m * tiles_n;
  const bool vec0 = (d0 % VEC == 0) && aligned16(X0) && aligned16(W0);
  const bool vec1 = (d1 % VEC == 0) && aligned16(X1) && aligned16(W1);
  const bool vec_c = (c % 4 == 0) && aligned16(E) && aligned16(u) && ((reinterpret_cast<uintptr_t>(code) & 3u) == 0);
  constexpr int slab_bytes = 2 * Cfg::SLAB * (int)sizeof(Acc);
  constexpr int lds = Cfg::OPERAND_BYTES > slab_bytes ? Cfg::OPERAND_BYTES : slab_bytes;
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&embrace_fwd_kernel<Cfg>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  embrace_fwd_kernel<Cfg><<<ntiles, kThreads, lds, s>>>(
      (const T*)X0, (const T*)X1, (const T*)W0, (const T*)W1, (const Acc*)b0, (const Acc*)b1, sel, u, seed, step_val,
      step_dev, row0, (T*)E, code, B, d0, d1, c, tiles_n, ntiles, vec0, vec1, vec_c);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

// tile shapes: "L" 64x64 tile, 2x2 waves, for tile counts that fill the chip; "S" 32x32 tile with
// the K range split over the 4 waves, for small B*c with long K (the A549 shapes: B=1024, c=256..512)
template <typename T> struct FwdCfg;
template <> struct FwdCfg<float> {
  using L = TileCfg<float, 64, 64, 32, 2, 2, 1, false, false>;
  using S = TileCfg<float, 32, 32, 128, 1, 1, 4, false, false>;
};
template <> struct FwdCfg<double> {
  using L = TileCfg<double, 64, 64, 16, 2, 2, 1, false, false>;
  using S = TileCfg<double, 32, 32, 64, 1, 1, 4, false, false>;
};
template <> struct FwdCfg<__bf16> {
  using L = TileCfg<__bf16, 64, 64, 128, 2, 2, 1, false, false>;
  using S = TileCfg<__bf16, 32, 32, 256, 1, 1, 4, false, false>;
};

template <typename T> static int fwd_dispatch(const void* X0, const void* X1, const void* W0, const void* b0, const void* W1,
                                              const void* b1, const SelArgs& sel, const double* u, uint64_t seed,
                                              uint64_t step_val, const uint64_t* step_dev, int64_t row0, void* E,
                                              uint8_t* code, int B, int d0, int d1, int c, hipStream_t s) {
  // EMB_FWD_IMPL=tiled keeps the round-1 kernels (A/B runs); default: K split over waves with LDS-DMA rings
  constexpr bool use_split = true;
  if (use_split) {
    const int rc = fwd_split_dispatch<T>(X0, X1, W0, b0, W1, b1, sel, u, seed, step_val, step_dev, row0, E, code, B, d0, d1, c, s);
    if (rc != 1) return rc;
  }
  const long tiles_L = (long)cdiv(B, 64) * cdiv(c, 64);
  if (tiles_L < 192) {   // small B*c, long K: operands streamed straight into MFMA fragments, K split over the 4 waves
    const int rc = launch_embrace_fwd_stream<T>(X0, X1, W0, b0, W1, b1, sel, u, seed, step_val, step_dev, row0, E, code, B, d0, d1, c, s);
    if (rc != 1) return rc;
  }
  if (tiles_L >= 192)
    return launch_fwd<typename FwdCfg<T>::L>(X0, X1, W0, b0, W1, b1, sel, u, seed, step_val, step_dev, row0, E, code, B, d0, d1, c, s);
  return launch_fwd<typename FwdCfg<T>::S>(X0, X1, W0, b0, W1, b1, sel, u, seed, step_val, step_dev, row0, E, code, B, d0, d1, c, s);
}

}  // namespace emb

extern "C" int emb_select_prep(const float* p, int p_rows, const float* avail, int device_dropout, uint64_t seed,
                               uint64_t step_val, const uint64_t* step_dev, int64_t row0, float* cdf0, int32_t* status,
                               int B, emb_stream_t stream) {
  EMB_CHECK_ARG(p && cdf0 && status, "emb_select_prep: null pointer");
  EMB_CHECK_ARG(B >= 0 && (p_rows == 1 || p_rows == B), "emb_select_prep: p_rows must be 1 or B (got %d, B=%d)", p_rows, B);
  if (B == 0) return EMB_OK;
  emb::select_prep_kernel<<<emb::cdiv(B, emb::kThreads), emb::kThreads, 0, (hipStream_t)stream>>>(
      p, p_rows, avail, device_dropout, seed, step_val, step_dev, row0, cdf0, status, B);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

static int embrace_fwd_entry(const char* who, const void* X0, const void* X1, const void* W0, const void* b0, const void* W1,
                             const void* b1, const emb::SelArgs& sel, const double* u, uint64_t seed, uint64_t step_val,
                             const uint64_t* step_dev, int64_t row0, void* E, uint8_t* code, int B, int d0, int d1, int c, int dtype,
                             emb_stream_t stream) {
  EMB_CHECK_ARG(X0 && X1 && W0 && W1 && b0 && b1 && E && code, "%s: null pointer", who);
  EMB_CHECK_ARG(B >= 0 && d0 > 0 && d1 > 0 && c > 0, "%s: bad dims B=%d d0=%d d1=%d c=%d", who, B, d0, d1, c);
  if (B == 0) return EMB_OK;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case EMB_F32: return emb::fwd_dispatch<float>(X0, X1, W0, b0, W1, b1, sel, u, seed, step_val, step_dev, row0, E, code, B, d0, d1, c, s);
    case EMB_BF16: return emb::fwd_dispatch<__bf16>(X0, X1, W0, b0, W1, b1, sel, u, seed, step_val, step_dev, row0, E, code, B, d0, d1, c, s);
    case EMB_F64: return emb::fwd_dispatch<double>(X0, X1, W0, b0, W1, b1, sel, u, seed, step_val, step_dev, row0, E, code, B, d0, d1, c, s);
  }
  emb::set_error("%s: unsupported dtype %d", who, dtype);
  return EMB_ERR_DTYPE;
}

extern "C" int emb_embrace_fwd(const void* X0, const void* X1, const void* W0, const void* b0, const void* W1,
                               const void* b1, const float* cdf0, const double* u, uint64_t seed, uint64_t step_val,
                               const uint64_t* step_dev, int64_t row0, void* E, uint8_t* code, int B, int d0, int d1, int c,
                               int dtype, emb_stream_t stream) {
  EMB_CHECK_ARG(cdf0, "emb_embrace_fwd: null pointer");
  const emb::SelArgs sel{cdf0, nullptr, nullptr, nullptr, 0, 0};
  return embrace_fwd_entry("emb_embrace_fwd", X0, X1, W0, b0, W1, b1, sel, u, seed, step_val, step_dev, row0, E, code, B, d0, d1, c,
                           dtype, stream);
}

extern "C" int emb_embrace_fwd_select(const void* X0, const void* X1, const void* W0, const void* b0, const void* W1,
                                      const void* b1, const float* p, int p_rows, const float* avail, int device_dropout,
                                      int32_t* status, const double* u, uint64_t seed, uint64_t step_val,
                                      const uint64_t* step_dev, int64_t row0, void* E, uint8_t* code, int B, int d0, int d1,
                                      int c, int dtype, emb_stream_t stream) {
  EMB_CHECK_ARG(p && status, "emb_embrace_fwd_select: null pointer");
  EMB_CHECK_ARG(p_rows == 1 || p_rows == B, "emb_embrace_fwd_select: p_rows must be 1 or B (got %d, B=%d)", p_rows, B);
  const emb::SelArgs sel{nullptr, p, avail, status, p_rows, device_dropout};
  return embrace_fwd_entry("emb_embrace_fwd_select", X0, X1, W0, b0, W1, b1, sel, u, seed, step_val, step_dev, row0, E, code, B, d0,
                           d1, c, dtype, stream);
}
