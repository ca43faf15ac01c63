// EmbraceNet backward in ONE launch: for each modality m
//   dD_m = dE * [idx == m] * [pre_m > 0]          (never materialised: applied while dE is staged)
//   dX_m = dD_m   W_m          "dgrad"  M=B, N=d_m, K=c   A = dE row-major,  B = W_m K-major
//   dW_m = dD_m^T X_m          "wgrad"  M=c, N=d_m, K=B   A = dE K-major,    B = X_m K-major
//   db_m = sum_b dD_m          folded into wgrad as one extra output column (B gets a virtual ones row)
// Replaces autograd through EmbraceNetMultimodal.py:52-60,80-88 (loss.backward(),
// utils/training_models_multimodal.py:156).  The grid is the concatenation of the four tile lists,
// heaviest (modality 1) first.
//
// Each of the four jobs is its own by-value kernel argument read at constant offsets.  (A single
// argument struct indexed by a run-time modality made hipcc 7.2 re-associate the kernarg address into a
// byte-granular scalar base, which s_load does not honour: the loaded pointers were garbage.)
#include "gemm_tile.h"
#include "reduce.h"
#include "embrace_bwd_split.h"
#include "gemm_jobs_api.h"
#include <cstdlib>
#include <cstring>

namespace emb {

template <typename T> struct BwdCfg;
template <> struct BwdCfg<float> {
  using D = TileCfg<float, 64, 64, 32, 2, 2, 1, false, true>;
  using W = TileCfg<float, 64, 64, 32, 2, 2, 1, true, true>;
};
template <> struct BwdCfg<double> {
  using D = TileCfg<double, 64, 64, 16, 2, 2, 1, false, true>;
  using W = TileCfg<double, 64, 64, 16, 2, 2, 1, true, true>;
};
template <> struct BwdCfg<__bf16> {
  using D = TileCfg<__bf16, 64, 64, 64, 2, 2, 1, false, true>;
  using W = TileCfg<__bf16, 64, 64, 64, 2, 2, 1, true, true>;
};

// one GEMM of the backward pass: C[M,N] = mask(dE) . B^T
template <typename T, typename OutT> struct BwdJob {
  const T* Bptr;     // W_m (dgrad) or X_m (wgrad), K-major
  OutT* C;           // dX_m or dW_m
  OutT* extra;       // db_m (wgrad) or nullptr
  int M, N, K;
  int ldb;           // = d_m
  int tiles_n;
  int end;           // exclusive end of this job's block range
  int want;          // XfEmbraceMask::want
  int vec_b, vec_c;  // 16-byte loads of B / 4-wide stores of C legal
  int S, kper, tiles;  // wgrad only: reduction (batch) split into S slices of kper rows; S == 1 -> direct store
  OutT* slab;          // [S][M][N+1] partial sums when S > 1
};

// (second launch bound = 2 waves per SIMD: caps the kernel at 256 registers, which makes hipcc keep the MFMA accumulators in
// VGPRs; with the default 512-register budget it put them in AGPRs and copied all 16 in and out around every MFMA group)
template <typename T>
__global__ __launch_bounds__(kThreads, 2) void embrace_bwd_kernel(const T* __restrict__ dE, const uint8_t* __restrict__ code,
                                                               int c, int vec_e,
                                                               const BwdJob<T, T> dg1,
                                                               const BwdJob<T, typename AccOf<T>::type> wg1,
                                                               const BwdJob<T, T> dg0,
                                                               const BwdJob<T, typename AccOf<T>::type> wg0) {
  using P = typename AccOf<T>::type;
  using CD = typename BwdCfg<T>::D;
  using CW = typename BwdCfg<T>::W;
  extern __shared__ __attribute__((aligned(16))) char arena[];
  const int bid = blockIdx.x;
  const GemmOperand<T> A{dE, code, c, vec_e != 0};
  if (bid < dg1.end) {
    const int tile = xcd_remap(bid, dg1.end);
    gemm_tile<CD>(A, GemmOperand<T>{dg1.Bptr, nullptr, dg1.ldb, dg1.vec_b != 0}, dg1.M, dg1.N, dg1.K, tile / dg1.tiles_n,
                  tile % dg1.tiles_n, XfEmbraceMask{(uint8_t)dg1.want}, -1,
                  EpiStore<T>{dg1.C, (long)dg1.N, nullptr, dg1.N, dg1.vec_c != 0}, arena);
  } else if (bid < wg1.end) {
    const int wq = bid - dg1.end, sl = wq / wg1.tiles, tile = xcd_remap(wq % wg1.tiles, wg1.tiles);
    const GemmOperand<T> Bop{wg1.Bptr, nullptr, wg1.ldb, wg1.vec_b != 0};
    if (wg1.S == 1) {
      gemm_tile<CW>(A, Bop, wg1.M, wg1.N, wg1.K, tile % (wg1.tiles / wg1.tiles_n), tile / (wg1.tiles / wg1.tiles_n), XfEmbraceMask{(uint8_t)wg1.want},
                    wg1.N, EpiStore<P>{wg1.C, (long)wg1.N, wg1.extra, wg1.N, wg1.vec_c != 0}, arena);
    } else {   // batch slice sl: partial sums (bias column included) to this slice's slab
      const int k_begin = sl * wg1.kper, k_end = min(wg1.K, k_begin + wg1.kper);
      gemm_tile<CW>(A, Bop, wg1.M, wg1.N, k_end, tile % (wg1.tiles / wg1.tiles_n), tile / (wg1.tiles / wg1.tiles_n), XfEmbraceMask{(uint8_t)wg1.want},
                    wg1.N, EpiStore<P>{wg1.slab + (long)sl * wg1.M * (wg1.N + 1), (long)(wg1.N + 1), nullptr, wg1.N + 1, false},
                    arena, k_begin);
    }
  } else if (bid < dg0.end) {
    const int tile = xcd_remap(bid - wg1.end, dg0.end - wg1.end);
    gemm_tile<CD>(A, GemmOperand<T>{dg0.Bptr, nullptr, dg0.ldb, dg0.vec_b != 0}, dg0.M, dg0.N, dg0.K, tile / dg0.tiles_n,
                  tile % dg0.tiles_n, XfEmbraceMask{(uint8_t)dg0.want}, -1,
                  EpiStore<T>{dg0.C, (long)dg0.N, nullptr, dg0.N, dg0.vec_c != 0}, arena);
  } else {
    const int wq = bid - dg0.end, sl = wq / wg0.tiles, tile = xcd_remap(wq % wg0.tiles, wg0.tiles);
    const GemmOperand<T> Bop{wg0.Bptr, nullptr, wg0.ldb, wg0.vec_b != 0};
    if (wg0.S == 1) {
      gemm_tile<CW>(A, Bop, wg0.M, wg0.N, wg0.K, tile % (wg0.tiles / wg0.tiles_n), tile / (wg0.tiles / wg0.tiles_n), XfEmbraceMask{(uint8_t)wg0.want},
                    wg0.N, EpiStore<P>{wg0.C, (long)wg0.N, wg0.extra, wg0.N, wg0.vec_c != 0}, arena);
    } else {   // batch slice sl: partial sums (bias column included) to this slice's slab
      const int k_begin = sl * wg0.kper, k_end = min(wg0.K, k_begin + wg0.kper);
      gemm_tile<CW>(A, Bop, wg0.M, wg0.N, k_end, tile % (wg0.tiles / wg0.tiles_n), tile / (wg0.tiles / wg0.tiles_n), XfEmbraceMask{(uint8_t)wg0.want},
                    wg0.N, EpiStore<P>{wg0.slab + (long)sl * wg0.M * (wg0.N + 1), (long)(wg0.N + 1), nullptr, wg0.N + 1, false},
                    arena, k_begin);
    }
  }
}

template <typename T>
static int bwd_dispatch(const void* dE, const uint8_t* code, const void* X0, const void* X1, const void* W0, const void* W1,
                        void* dX0, void* dX1, void* dW0, void* db0, void* dW1, void* db1, void* ws, int64_t ws_bytes, int B, int d0,
                        int d1, int c, hipStream_t s) {
  using P = typename AccOf<T>::type;
  using CD = typename BwdCfg<T>::D;
  using CW = typename BwdCfg<T>::W;
  constexpr int VEC = Elem<T>::VEC;
  const int vec_e = (c % VEC == 0) && aligned16(dE) && ((reinterpret_cast<uintptr_t>(code) & 7u) == 0);
  int n = 0;
  int64_t ws_used = 0;
  auto dgrad = [&](const void* W, void* dX, int d, int m) {
    BwdJob<T, T> j{};
    j.Bptr = (const T*)W; j.C = (T*)dX; j.extra = nullptr;
    j.M = B; j.N = d; j.K = c; j.ldb = d;
    j.tiles_n = cdiv(d, CD::BN);
    if (dX != nullptr) n += cdiv(B, CD::BM) * j.tiles_n;
    j.end = n;
    j.want = (m ? EMB_CODE_IDX : 0) | EMB_CODE_ACTIVE;
    j.vec_b = (d % VEC == 0) && aligned16(W);
    j.vec_c = (d % 4 == 0) && aligned16(dX);
    return j;
  };
  auto wgrad = [&](const void* X, void* dW, void* db, int d, int m) {
    BwdJob<T, P> j{};
    j.Bptr = (const T*)X; j.C = (P*)dW; j.extra = (P*)db;
    j.M = c; j.N = d; j.K = B; j.ldb = d;
    j.tiles_n = cdiv(d + 1, CW::BN);
    j.tiles = cdiv(c, CW::BM) * j.tiles_n;
    j.S = 1; j.kper = B; j.slab = nullptr;
    if (ws != nullptr && j.tiles < 256 && B >= 4 * CW::BK) {   // few output tiles, long batch: slice the batch
      int S = 512 / j.tiles;
      if (S > 8) S = 8;
      if (S > B / (2 * CW::BK)) S = B / (2 * CW::BK);
      const int64_t per = (int64_t)c * (d + 1) * (int64_t)sizeof(P);
      if ((int64_t)S * per > ws_bytes - ws_used) S = (int)((ws_bytes - ws_used) / per);
      if (S > 1) {
        j.kper = cdiv(cdiv(B, S), CW::BK) * CW::BK;
        j.S = cdiv(B, j.kper);
        j.slab = (P*)((char*)ws + ws_used);
        ws_used += (int64_t)j.S * per;
      }
    }
    n += j.tiles * j.S;
    j.end = n;
    j.want = (m ? EMB_CODE_IDX : 0) | EMB_CODE_ACTIVE;
    j.vec_b = (d % VEC == 0) && aligned16(X);
    j.vec_c = (d % 4 == 0) && aligned16(dW);
    return j;
  };
  const BwdJob<T, T> dg1 = dgrad(W1, dX1, d1, 1);
  const BwdJob<T, P> wg1 = wgrad(X1, dW1, db1, d1, 1);
  const BwdJob<T, T> dg0 = dgrad(W0, dX0, d0, 0);
  const BwdJob<T, P> wg0 = wgrad(X0, dW0, db0, d0, 0);
  constexpr int lds = gemm_tile_lds<CD>() > gemm_tile_lds<CW>() ? gemm_tile_lds<CD>() : gemm_tile_lds<CW>();
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&embrace_bwd_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  embrace_bwd_kernel<T><<<n, kThreads, lds, s>>>((const T*)dE, code, c, vec_e, dg1, wg1, dg0, wg0);
  EMB_CHECK_LAUNCH();
  // per-slice slabs -> dW_m / db_m, slices summed in fixed order (reduce.hip)
  if (wg1.S > 1) {
    ReduceJob j{};
    j.in = wg1.slab; j.out[0] = dW1; j.out[1] = db1; j.per = (long)c * (d1 + 1); j.S = wg1.S; j.kind = RJ_LINEAR; j.iv[0] = d1;
    const int rc = reduce_submit(j, sizeof(P) == 8, s);
    if (rc != EMB_OK) return rc;
  }
  if (wg0.S > 1) {
    ReduceJob j{};
    j.in = wg0.slab; j.out[0] = dW0; j.out[1] = db0; j.per = (long)c * (d0 + 1); j.S = wg0.S; j.kind = RJ_LINEAR; j.iv[0] = d0;
    const int rc = reduce_submit(j, sizeof(P) == 8, s);
    if (rc != EMB_OK) return rc;
  }
  return EMB_OK;
}

// dD_m = dE * keep_m for both modalities in one elementwise pass (8 elements per thread: 16-byte loads of bf16 / two of f32)
template <typename T> __global__ __launch_bounds__(256) void premask_kernel(const T* __restrict__ dE, const uint8_t* __restrict__ code,
                                                                           T* __restrict__ dD0, T* __restrict__ dD1, long n8) {
  constexpr int VEC = Elem<T>::VEC, NV = 8 / VEC;   // 16-byte vectors per thread
  using V = typename Vec16<T>::type;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const u32x2 cw = *reinterpret_cast<const u32x2*>(code + i * 8);
    V v[NV], o0[NV], o1[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = *reinterpret_cast<const V*>(dE + i * 8 + k * VEC);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t cb = cw[e >> 2] >> (8 * (e & 3));
      o0[e / VEC][e % VEC] = (cb & EMB_CODE_KEEP0) ? v[e / VEC][e % VEC] : (T)0.0f;
      o1[e / VEC][e % VEC] = (cb & EMB_CODE_KEEP1) ? v[e / VEC][e % VEC] : (T)0.0f;
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      *reinterpret_cast<V*>(dD0 + i * 8 + k * VEC) = o0[k];
      *reinterpret_cast<V*>(dD1 + i * 8 + k * VEC) = o1[k];
    }
  }
}

}  // namespace emb

extern "C" int emb_embrace_premask(const void* dE, const uint8_t* code, void* dD0, void* dD1, int B, int c, int dtype,
                                   emb_stream_t stream) {
  EMB_CHECK_ARG(dE && code && dD0 && dD1, "emb_embrace_premask: null pointer");
  EMB_CHECK_ARG(B > 0 && c > 0 && ((long)B * c) % 8 == 0, "emb_embrace_premask: B * c must be a multiple of 8 (B=%d c=%d)", B, c);
  EMB_CHECK_ARG(emb::aligned16(dE) && emb::aligned16(dD0) && emb::aligned16(dD1) && (reinterpret_cast<uintptr_t>(code) & 7u) == 0,
                "emb_embrace_premask: dE, dD0, dD1 must be 16-byte, code 8-byte aligned");
  const long n8 = (long)B * c / 8;
  const int blocks = (int)((n8 + 255) / 256 < 2048 ? (n8 + 255) / 256 : 2048);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EMB_BF16) emb::premask_kernel<__bf16><<<blocks, 256, 0, s>>>((const __bf16*)dE, code, (__bf16*)dD0, (__bf16*)dD1, n8);
  else if (dtype == EMB_F32) emb::premask_kernel<float><<<blocks, 256, 0, s>>>((const float*)dE, code, (float*)dD0, (float*)dD1, n8);
  else {
    emb::set_error("emb_embrace_premask: unsupported dtype %d", dtype);
    return EMB_ERR_DTYPE;
  }
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

extern "C" int emb_embrace_bwd_masked_supported(int B, int d0, int d1, int c, int dtype) {
  if (B <= 0 || c <= 0 || d0 <= 0 || d1 <= 0) return 0;
  if (dtype == EMB_F32) return c % 4 == 0 && d0 % 4 == 0 && d1 % 4 == 0;      // ring GEMM (gemm_jobs.h)
  if (dtype == EMB_BF16) return c % 16 == 0 && d0 % 8 == 0 && d1 % 8 == 0;   // split kernel without its mask stage (embrace_bwd_split.h)
  return 0;
}

extern "C" int emb_embrace_bwd_masked(const void* dD0, const void* dD1, const void* X0, const void* X1, const void* W0,
                                      const void* W1, void* dX0, void* dX1, void* dW0, void* db0, void* dW1, void* db1,
                                      void* workspace, int64_t workspace_bytes, int B, int d0, int d1, int c, int dtype,
                                      emb_stream_t stream) {
  EMB_CHECK_ARG(dD0 && dD1 && X0 && X1 && W0 && W1 && dW0 && db0 && dW1 && db1, "emb_embrace_bwd_masked: null pointer");
  EMB_CHECK_ARG(emb_embrace_bwd_masked_supported(B, d0, d1, c, dtype),
                "emb_embrace_bwd_masked: unsupported shape / dtype (see emb_embrace_bwd_masked_supported)");
  constexpr int force_S = 0;
  const int rc = dtype == EMB_BF16
                     ? emb::bwd_split_dispatch(nullptr, nullptr, dD0, dD1, X0, X1, W0, W1, dX0, dX1, dW0, db0, dW1, db1, workspace,
                                               workspace_bytes, B, d0, d1, c, force_S, (hipStream_t)stream)
                     : emb::gemm_jobs_bwd(dD0, dD1, X0, X1, W0, W1, dX0, dX1, dW0, db0, dW1, db1, workspace, workspace_bytes, B,
                                             d0, d1, c, force_S, (hipStream_t)stream);
  if (rc == 1) {
    emb::set_error("emb_embrace_bwd_masked: operands must be 16-byte aligned and smaller than 2 GiB");
    return EMB_ERR_ARG;
  }
  return rc;
}

extern "C" int emb_embrace_bwd(const void* dE, const uint8_t* code, const void* X0, const void* X1, const void* W0,
                               const void* W1, void* dX0, void* dX1, void* dW0, void* db0, void* dW1, void* db1,
                               void* workspace, int64_t workspace_bytes, int B, int d0, int d1, int c, int dtype,
                               emb_stream_t stream) {
  EMB_CHECK_ARG(dE && code && X0 && X1 && W0 && W1 && dW0 && db0 && dW1 && db1, "emb_embrace_bwd: null pointer");
  EMB_CHECK_ARG(B > 0 && d0 > 0 && d1 > 0 && c > 0, "emb_embrace_bwd: bad dims B=%d d0=%d d1=%d c=%d", B, d0, d1, c);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EMB_BF16) {   // K split over waves, LDS-DMA rings (embrace_bwd_split.h); shapes it refuses take the tiled kernel below
    constexpr int force_S = 0;
    {
      const int rc = emb::bwd_split_dispatch(dE, code, nullptr, nullptr, X0, X1, W0, W1, dX0, dX1, dW0, db0, dW1, db1, workspace, workspace_bytes, B, d0, d1, c, force_S, s);
      if (rc != 1) return rc;
    }
  }
  switch (dtype) {
    case EMB_F32: return emb::bwd_dispatch<float>(dE, code, X0, X1, W0, W1, dX0, dX1, dW0, db0, dW1, db1, workspace, workspace_bytes, B, d0, d1, c, s);
    case EMB_BF16: return emb::bwd_dispatch<__bf16>(dE, code, X0, X1, W0, W1, dX0, dX1, dW0, db0, dW1, db1, workspace, workspace_bytes, B, d0, d1, c, s);
    case EMB_F64: return emb::bwd_dispatch<double>(dE, code, X0, X1, W0, W1, dX0, dX1, dW0, db0, dW1, db1, workspace, workspace_bytes, B, d0, d1, c, s);
  }
  emb::set_error("emb_embrace_bwd: unsupported dtype %d", dtype);
  return EMB_ERR_DTYPE;
}
