// Linear (+ReLU, +Dropout) layers of the post stack, forward and backward.
// Reference: EmbraceNetMultimodal.py:143-147 (hidden post layers), :151 (final Linear -> n_classes),
// same shape as FFNN_pre.py:25-33.  Forward fuses bias, ReLU and inverted dropout into the GEMM
// epilogue; backward applies the saved mask while dY is staged and folds the bias gradient into the
// weight-gradient GEMM (virtual ones row), all in one launch.
#include "gemm_tile.h"
#include "gemm_jobs_api.h"
#include "reduce.h"

namespace emb {

// dZ = dY * [mask bits set] * scale: the gradient behind ReLU / inverted dropout, written once for the fp32 ring GEMM
__global__ __launch_bounds__(256) void linear_premask_kernel(const float* __restrict__ dY, const uint8_t* __restrict__ mask,
                                                             float* __restrict__ dZ, long n4, uint8_t need, float scale) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = reinterpret_cast<const f32x4*>(dY)[i];
  const uint32_t m = reinterpret_cast<const uint32_t*>(mask)[i];
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (((m >> (8 * e)) & need) == need) ? v[e] * scale : 0.0f;
  reinterpret_cast<f32x4*>(dZ)[i] = o;
}

template <typename T> struct LinCfg;
template <> struct LinCfg<float> {
  using F = TileCfg<float, 64, 64, 32, 2, 2, 1, false, false>;
  using FS = TileCfg<float, 32, 32, 128, 1, 1, 4, false, false>;
  using FS16 = TileCfg<float, 16, 32, 128, 1, 1, 4, false, false>;   // very few 32 x 32 tiles (B = 1024 rows into <= 64 units)
  using D = TileCfg<float, 64, 64, 32, 2, 2, 1, false, true>;
  using W = TileCfg<float, 64, 64, 32, 2, 2, 1, true, true>;
};
template <> struct LinCfg<double> {
  using F = TileCfg<double, 64, 64, 16, 2, 2, 1, false, false>;
  using FS = TileCfg<double, 32, 32, 64, 1, 1, 4, false, false>;
  using D = TileCfg<double, 64, 64, 16, 2, 2, 1, false, true>;
  using W = TileCfg<double, 64, 64, 16, 2, 2, 1, true, true>;
};
template <> struct LinCfg<__bf16> {
  using F = TileCfg<__bf16, 64, 64, 128, 2, 2, 1, false, false>;
  using FS = TileCfg<__bf16, 32, 32, 256, 1, 1, 4, false, false>;
  using D = TileCfg<__bf16, 64, 64, 64, 2, 2, 1, false, true>;
  using W = TileCfg<__bf16, 64, 64, 64, 2, 2, 1, true, true>;
};

template <class Cfg>
__global__ __launch_bounds__(kThreads, 2) void linear_fwd_kernel(GemmOperand<typename Cfg::T> X, GemmOperand<typename Cfg::T> W,
                                                              EpiLinear<typename Cfg::T, typename Cfg::M::Acc> epi,
                                                              const uint64_t* step_dev, uint64_t step_val, int layer_id,
                                                              int B, int K, int N, int tiles_n, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char arena[];
  const int tile = xcd_remap(blockIdx.x, ntiles);
  epi.stream = rng_stream(step_val + (step_dev ? *step_dev : 0), EMB_RNG_DROPOUT0 + layer_id);
  gemm_tile<Cfg>(X, W, B, N, K, tile / tiles_n, tile % tiles_n, XfNone{}, -1, epi, arena);
}

template <class Cfg>
static int launch_linear_fwd(const void* X, const void* W, const void* b, void* Y, uint8_t* mask, int relu, float dropout_p,
                             int layer_id, uint64_t seed, uint64_t step_val, const uint64_t* step_dev, int64_t row0, int B,
                             int K, int N, hipStream_t s) {
  using T = typename Cfg::T;
  using P = typename Cfg::M::Acc;
  constexpr int VEC = Elem<T>::VEC;
  const bool vk = (K % VEC == 0) && aligned16(X) && aligned16(W);
  GemmOperand<T> Xo{(const T*)X, nullptr, K, vk}, Wo{(const T*)W, nullptr, K, vk};
  EpiLinear<T, P> epi{(T*)Y, mask, (const P*)b, (long)N, N, relu != 0,
                      dropout_p > 0.f ? 1.0f / (1.0f - dropout_p) : 1.0f, dropout_p, seed, 0, row0,
                      (N % 4 == 0) && aligned16(Y) && ((reinterpret_cast<uintptr_t>(mask) & 3u) == 0)};
  const int tiles_n = cdiv(N, Cfg::BN), ntiles = cdiv(B, Cfg::BM) * tiles_n;
  constexpr int lds = gemm_tile_lds<Cfg>();
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_fwd_kernel<Cfg>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  linear_fwd_kernel<Cfg><<<ntiles, kThreads, lds, s>>>(Xo, Wo, epi, step_dev, step_val, layer_id, B, K, N, tiles_n, ntiles);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

template <typename T> struct LinBwdArgs {
  using P = typename AccOf<T>::type;
  GemmOperand<T> dY_rm, dY_km, Wk, Xk;
  T* dX;
  P* dW;
  P* db;
  XfLinearMask xf;
  int B, K, N;
  int end_dgrad, tiles_n_d, tiles_n_w, nblocks;
  int S, kper, tiles_w;      // wgrad split over the batch: S slices of kper rows; S == 1 -> direct store
  P* slab;                   // [S][N][K+1] partial sums (S > 1)
  bool vec_dx, vec_dw;
};

template <typename T> __global__ __launch_bounds__(kThreads, 2) void linear_bwd_kernel(const LinBwdArgs<T> a) {
  using P = typename AccOf<T>::type;
  extern __shared__ __attribute__((aligned(16))) char arena[];
  const int bid = blockIdx.x;
  if (bid < a.end_dgrad) {   // dX[B,K] = dZ[B,N] . W[N,K]
    const int tile = xcd_remap(bid, a.end_dgrad);
    EpiStore<T> epi{a.dX, (long)a.K, nullptr, a.K, a.vec_dx};
    gemm_tile<typename LinCfg<T>::D>(a.dY_rm, a.Wk, a.B, a.K, a.N, tile / a.tiles_n_d, tile % a.tiles_n_d, a.xf, -1, epi, arena);
  } else if (a.S == 1) {     // dW[N,K] = dZ^T[N,B] . X[B,K], db = dZ^T . 1
    const int tile = xcd_remap(bid - a.end_dgrad, a.nblocks - a.end_dgrad);
    EpiStore<P> epi{a.dW, (long)a.K, a.db, a.K, a.vec_dw};
    gemm_tile<typename LinCfg<T>::W>(a.dY_km, a.Xk, a.N, a.K, a.B, tile / a.tiles_n_w, tile % a.tiles_n_w, a.xf, a.K, epi, arena);
  } else {                   // few output tiles, long batch: slice the batch, partial sums to the slab
    const int w = bid - a.end_dgrad, s = w / a.tiles_w, tile = w % a.tiles_w;
    const int k_begin = s * a.kper, k_end = min(a.B, k_begin + a.kper);
    EpiStore<P> epi{a.slab + (long)s * a.N * (a.K + 1), (long)(a.K + 1), nullptr, a.K + 1, false};
    gemm_tile<typename LinCfg<T>::W>(a.dY_km, a.Xk, a.N, a.K, k_end, tile / a.tiles_n_w, tile % a.tiles_n_w, a.xf, a.K, epi, arena, k_begin);
  }
}

template <typename T>
static int linear_bwd_dispatch(const void* dY, const uint8_t* mask, const void* X, const void* W, void* dX, void* dW, void* db,
                               int relu, float dropout_p, void* ws, int64_t ws_bytes, int B, int K, int N, hipStream_t s) {
  using P = typename AccOf<T>::type;
  using CD = typename LinCfg<T>::D;
  using CW = typename LinCfg<T>::W;
  constexpr int VEC = Elem<T>::VEC;
  if constexpr (sizeof(T) == 4) {
    // large fp32 layers (the first post layer of a wide fusion: 1024 -> 256 at B = 1024): the ring GEMM of gemm_jobs.h on the
    // pre-masked gradient -- a Linear layer's backward is one modality of the docking backward (c = N out, d = K in).
    // The 64 x 64 x 32 tiles below run such a layer at ~12 TFLOP/s (89 us at cfg4).
    const int64_t dz_bytes = ((int64_t)B * N * 4 + 255) & ~(int64_t)255;
    const bool masked = mask != nullptr && (relu || dropout_p > 0.f);
    if ((long)B * K * N >= (1l << 27) && K % 4 == 0 && N % 4 == 0 && ((long)B * N) % 4 == 0 && ws != nullptr && aligned16(ws) && aligned16(dY) &&
        (!masked || ((reinterpret_cast<uintptr_t>(mask) & 3u) == 0 && ws_bytes >= dz_bytes))) {
      const void* dZ = dY;
      int64_t used = 0;
      if (masked) {
        const long n4 = (long)B * N / 4;
        linear_premask_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, s>>>((const float*)dY, mask, (float*)ws, n4,
                                                                           (uint8_t)((relu ? 1 : 0) | (dropout_p > 0.f ? 2 : 0)),
                                                                           dropout_p > 0.f ? 1.0f / (1.0f - dropout_p) : 1.0f);
        EMB_CHECK_LAUNCH();
        dZ = ws;
        used = dz_bytes;
      }
      const int rc = gemm_jobs_bwd(dZ, nullptr, X, nullptr, W, nullptr, dX, nullptr, dW, db, nullptr, nullptr, (char*)ws + used,
                                   ws_bytes - used, B, K, 0, N, 0, s);
      if (rc != 1) return rc;      // 1: the shapes do not qualify -- the tile kernel below takes the layer
    }
  }
  LinBwdArgs<T> a;
  const bool vn = (N % VEC == 0) && aligned16(dY) && (mask == nullptr || (reinterpret_cast<uintptr_t>(mask) & 7u) == 0);
  const bool vk = (K % VEC == 0) && aligned16(X) && aligned16(W);
  a.dY_rm = GemmOperand<T>{(const T*)dY, mask, N, vn};
  a.dY_km = a.dY_rm;
  a.Wk = GemmOperand<T>{(const T*)W, nullptr, K, vk};
  a.Xk = GemmOperand<T>{(const T*)X, nullptr, K, vk};
  a.dX = (T*)dX; a.dW = (P*)dW; a.db = (P*)db;
  a.xf = XfLinearMask{(uint8_t)((relu ? 1 : 0) | (dropout_p > 0.f ? 2 : 0)), dropout_p > 0.f ? 1.0f / (1.0f - dropout_p) : 1.0f};
  a.B = B; a.K = K; a.N = N;
  a.tiles_n_d = cdiv(K, CD::BN);
  a.end_dgrad = dX ? cdiv(B, CD::BM) * a.tiles_n_d : 0;
  a.tiles_n_w = cdiv(K + 1, CW::BN);
  a.tiles_w = cdiv(N, CW::BM) * a.tiles_n_w;
  // split the batch when the output is only a few tiles (FFNN / post / head layers) and scratch is available
  a.S = 1; a.kper = B; a.slab = (P*)ws;
  if (ws != nullptr && a.tiles_w < 64 && B >= 4 * CW::BK) {
    int S = 256 / a.tiles_w;
    const int smax = B / (2 * CW::BK);
    if (S > smax) S = smax;
    const int64_t per = (int64_t)N * (K + 1) * (int64_t)sizeof(P);
    if ((int64_t)S * per > ws_bytes) S = (int)(ws_bytes / per);
    if (S > 1) {
      a.kper = cdiv(cdiv(B, S), CW::BK) * CW::BK;
      a.S = cdiv(B, a.kper);
    }
  }
  a.nblocks = a.end_dgrad + a.tiles_w * a.S;
  a.vec_dx = (K % 4 == 0) && aligned16(dX);
  a.vec_dw = (K % 4 == 0) && aligned16(dW);
  constexpr int lds = gemm_tile_lds<CD>() > gemm_tile_lds<CW>() ? gemm_tile_lds<CD>() : gemm_tile_lds<CW>();
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_bwd_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  linear_bwd_kernel<T><<<a.nblocks, kThreads, lds, s>>>(a);
  EMB_CHECK_LAUNCH();
  if (a.S > 1) {
    ReduceJob j{};   // dW[n][k], db[n] from the slab, slices summed in fixed order (reduce.hip)
    j.in = a.slab; j.out[0] = dW; j.out[1] = db; j.per = (long)N * (K + 1); j.S = a.S; j.kind = RJ_LINEAR; j.iv[0] = K;
    const int rc = reduce_submit(j, sizeof(P) == 8, s);
    if (rc != EMB_OK) return rc;
  }
  return EMB_OK;
}

template <typename T>
static int linear_fwd_dispatch(const void* X, const void* W, const void* b, void* Y, uint8_t* mask, int relu, float dropout_p,
                               int layer_id, uint64_t seed, uint64_t step_val, const uint64_t* step_dev, int64_t row0, int B,
                               int K, int N, hipStream_t s) {
  const long tiles_L = (long)cdiv(B, 64) * cdiv(N, 64);
  if (tiles_L >= 192 || K < 256)
    return launch_linear_fwd<typename LinCfg<T>::F>(X, W, b, Y, mask, relu, dropout_p, layer_id, seed, step_val, step_dev, row0, B, K, N, s);
  if constexpr (sizeof(T) == 4) {
    if ((long)cdiv(B, 32) * cdiv(N, 32) <= 64)   // a first FFNN layer (562 -> 32 at B = 1024: 32 tiles of 32 x 32 ran 30 us)
      return launch_linear_fwd<typename LinCfg<T>::FS16>(X, W, b, Y, mask, relu, dropout_p, layer_id, seed, step_val, step_dev, row0, B, K, N, s);
  }
  return launch_linear_fwd<typename LinCfg<T>::FS>(X, W, b, Y, mask, relu, dropout_p, layer_id, seed, step_val, step_dev, row0, B, K, N, s);
}

}  // namespace emb

extern "C" int emb_linear_fwd(const void* X, const void* W, const void* b, void* Y, uint8_t* mask, int relu, float dropout_p,
                              int layer_id, uint64_t seed, uint64_t step_val, const uint64_t* step_dev, int64_t row0, int B,
                              int K, int N, int dtype, emb_stream_t stream) {
  EMB_CHECK_ARG(X && W && b && Y, "emb_linear_fwd: null pointer");
  EMB_CHECK_ARG(B >= 0 && K > 0 && N > 0, "emb_linear_fwd: bad dims B=%d K=%d N=%d", B, K, N);
  EMB_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "emb_linear_fwd: dropout_p must be in [0,1)");
  if (B == 0) return EMB_OK;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case EMB_F32: return emb::linear_fwd_dispatch<float>(X, W, b, Y, mask, relu, dropout_p, layer_id, seed, step_val, step_dev, row0, B, K, N, s);
    case EMB_BF16: return emb::linear_fwd_dispatch<__bf16>(X, W, b, Y, mask, relu, dropout_p, layer_id, seed, step_val, step_dev, row0, B, K, N, s);
    case EMB_F64: return emb::linear_fwd_dispatch<double>(X, W, b, Y, mask, relu, dropout_p, layer_id, seed, step_val, step_dev, row0, B, K, N, s);
  }
  emb::set_error("emb_linear_fwd: unsupported dtype %d", dtype);
  return EMB_ERR_DTYPE;
}

extern "C" int emb_linear_bwd(const void* dY, const uint8_t* mask, const void* X, const void* W, void* dX, void* dW, void* db,
                              int relu, float dropout_p, void* workspace, int64_t workspace_bytes, int B, int K, int N,
                              int dtype, emb_stream_t stream) {
  EMB_CHECK_ARG(dY && X && W && dW && db, "emb_linear_bwd: null pointer");
  EMB_CHECK_ARG(B > 0 && K > 0 && N > 0, "emb_linear_bwd: bad dims B=%d K=%d N=%d", B, K, N);
  EMB_CHECK_ARG(mask || (!relu && dropout_p == 0.f), "emb_linear_bwd: mask required when relu or dropout is on");
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case EMB_F32: return emb::linear_bwd_dispatch<float>(dY, mask, X, W, dX, dW, db, relu, dropout_p, workspace, workspace_bytes, B, K, N, s);
    case EMB_BF16: return emb::linear_bwd_dispatch<__bf16>(dY, mask, X, W, dX, dW, db, relu, dropout_p, workspace, workspace_bytes, B, K, N, s);
    case EMB_F64: return emb::linear_bwd_dispatch<double>(dY, mask, X, W, dX, dW, db, relu, dropout_p, workspace, workspace_bytes, B, K, N, s);
  }
  emb::set_error("emb_linear_bwd: unsupported dtype %d", dtype);
  return EMB_ERR_DTYPE;
}
