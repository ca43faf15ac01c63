// Small multi-layer perceptron stacks in ONE launch (forward) / TWO launches (backward):
//   epigenomic pre-network  FFNN_pre.py:18-49   1-4 x (Linear -> ReLU -> Dropout), widths <= 256
//   post stack / head       EmbraceNetMultimodal.py:134-154   0-2 x (Linear -> ReLU -> Dropout) + Linear(-> 2)
// These layers are a few thousand MACs per row: as separate GEMM launches they cost ~5 us each of pure launch
// and fill/drain latency (12 launches per step for a 3-layer FFNN).  Here one workgroup carries 32 rows through
// the whole stack with activations and the current layer's weights in LDS (plain fp32 FMAs -- the matrices are
// far too small for MFMA tiles to pay), stores every layer's output + mask byte for the backward pass, and the
// backward walks the stack in reverse producing dX, per-workgroup partial dW/db, which a second tiny launch sums
// in fixed order (deterministic).  Eligibility (every layer's [N][K+1] fp32 image <= 48 KiB) is checked by
// emb_mlp_supported(); larger layers use the tiled GEMM kernels of linear.hip.
#include "common.h"
#include "philox.h"

namespace emb {

constexpr int kMlpMaxL = 4, kMlpWBudget = 12288;   // LDS elements for one layer's weights
template <typename T> struct MlpRB { static constexpr int value = sizeof(T) == 8 ? 16 : 32; };   // rows per workgroup

template <typename T> struct MlpArgs {
  using P = typename AccOf<T>::type;
  const T* x;            // [B][F]
  const T* W[kMlpMaxL];  // [N_l][K_l] in compute dtype
  const P* b[kMlpMaxL];
  T* h[kMlpMaxL];        // outputs of every layer [B][N_l] (the last one is the result)
  uint8_t* mask[kMlpMaxL];   // bit0 pre-activation > 0, bit1 kept by dropout (nullable when the layer has neither)
  int N[kMlpMaxL], relu[kMlpMaxL], layer_id[kMlpMaxL];
  float drop[kMlpMaxL];
  int B, F, L;
  uint64_t seed, step_val;
  const uint64_t* step_dev;
  int64_t row0;
};

template <typename T> struct MlpBwdArgs {
  using P = typename AccOf<T>::type;
  const T* x;
  const T* W[kMlpMaxL];
  const T* h[kMlpMaxL];
  const uint8_t* mask[kMlpMaxL];
  const T* dy;           // [B][N_{L-1}]
  T* dx;                 // [B][F] or nullptr
  P* part;               // [nblk][total] partial sums; layout per layer: dW [N][K] then db [N]
  int N[kMlpMaxL], relu[kMlpMaxL];
  float drop[kMlpMaxL];
  int B, F, L, total;
};

struct MlpReduceArgs {
  void* dW[kMlpMaxL];
  void* db[kMlpMaxL];
  int N[kMlpMaxL], K[kMlpMaxL];
  int L, total, nblk;
};

// sum_k a[k*sa] * b[k*sb], 4 independent partial sums (LDS reads of consecutive iterations overlap)
template <typename A>
__device__ __forceinline__ A lds_dot(const A* __restrict__ a, int sa, const A* __restrict__ b, int sb, int n, A init) {
  A s0 = init, s1 = 0, s2 = 0, s3 = 0;
  int k = 0;
  for (; k + 4 <= n; k += 4) {
    const A a0 = a[(k + 0) * sa], a1 = a[(k + 1) * sa], a2 = a[(k + 2) * sa], a3 = a[(k + 3) * sa];
    const A b0 = b[(k + 0) * sb], b1 = b[(k + 1) * sb], b2 = b[(k + 2) * sb], b3 = b[(k + 3) * sb];
    s0 += a0 * b0; s1 += a1 * b1; s2 += a2 * b2; s3 += a3 * b3;
  }
  for (; k < n; ++k) s0 += a[k * sa] * b[k * sb];
  return (s0 + s1) + (s2 + s3);
}

// weights of layer l as an [N][K+1] image in LDS (pitch K+1: conflict-free across n)
template <typename T>
__device__ __forceinline__ void mlp_stage_w(const T* __restrict__ W, typename AccOf<T>::type* Ws, int N, int K) {
  using A = typename AccOf<T>::type;
#pragma unroll 8
  for (int i = threadIdx.x; i < N * K; i += blockDim.x) {
    const int n = i / K;
    Ws[n * (K + 1) + (i - n * K)] = (A)W[i];
  }
}

template <typename T, int l>
__device__ __forceinline__ void mlp_fwd_layer(const MlpArgs<T>& a, typename AccOf<T>::type*& in, typename AccOf<T>::type*& out,
                                              const typename AccOf<T>::type* Ws, int pin, int pout, int K, int row_base, uint64_t step) {
  using A = typename AccOf<T>::type;
  constexpr int kMlpRB = MlpRB<T>::value;
  const int N = a.N[l];
  const float p = a.drop[l];
  const float keep_scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  const uint64_t stream = rng_stream(step, EMB_RNG_DROPOUT0 + a.layer_id[l]);
  for (int i = threadIdx.x; i < kMlpRB * N; i += blockDim.x) {
    const int r = i / N, n = i - r * N, row = row_base + r;
    if (row >= a.B) continue;
    A acc = lds_dot<A>(in + r * pin, 1, Ws + n * (K + 1), 1, K, (A)a.b[l][n]);
    bool act = true, keep = true;
    if (a.relu[l]) {
      act = acc > (A)0;
      acc = act ? acc : (A)0;
    }
    if (p > 0.f) {
      keep = uniform24(philox4x32_10(a.seed, stream, (uint64_t)(a.row0 + row) * (uint64_t)N + n).x) >= p;
      acc = keep ? acc * (A)keep_scale : (A)0;
    }
    const T hv = (T)acc;
    a.h[l][(long)row * N + n] = hv;
    if (a.mask[l]) a.mask[l][(long)row * N + n] = (uint8_t)((act ? 1 : 0) | (keep ? 2 : 0));
    out[r * pout + n] = (A)hv;          // the next layer sees what the stored activation holds
  }
  __syncthreads();
  A* t = in; in = out; out = t;
}

// Everything the block needs (all weights, its input rows) is requested up front: one memory round trip, then the
// layers run out of LDS.
template <typename T>
__global__ __launch_bounds__(256) void mlp_fwd_kernel(const MlpArgs<T> a, int pitch) {
  using A = typename AccOf<T>::type;
  constexpr int kMlpRB = MlpRB<T>::value;
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  A* in = reinterpret_cast<A*>(smraw);
  A* out = in + kMlpRB * pitch;
  A* Ws = out + kMlpRB * pitch;
  const int row_base = blockIdx.x * kMlpRB;
  const uint64_t step = a.step_val + (a.step_dev ? *a.step_dev : 0);
  const int K0 = a.F, K1 = a.N[0], K2 = a.N[1], K3 = a.N[2];
  const int w1 = a.N[0] * (K0 + 1), w2 = w1 + (a.L > 1 ? a.N[1] * (K1 + 1) : 0), w3 = w2 + (a.L > 2 ? a.N[2] * (K2 + 1) : 0);
#pragma unroll 8
  for (int i = threadIdx.x; i < kMlpRB * a.F; i += blockDim.x) {
    const int r = i / a.F, k = i - r * a.F;
    in[r * pitch + k] = row_base + r < a.B ? (A)a.x[(long)(row_base + r) * a.F + k] : (A)0;
  }
  mlp_stage_w<T>(a.W[0], Ws, a.N[0], K0);
  if (a.L > 1) mlp_stage_w<T>(a.W[1], Ws + w1, a.N[1], K1);
  if (a.L > 2) mlp_stage_w<T>(a.W[2], Ws + w2, a.N[2], K2);
  if (a.L > 3) mlp_stage_w<T>(a.W[3], Ws + w3, a.N[3], K3);
  __syncthreads();
  mlp_fwd_layer<T, 0>(a, in, out, Ws, pitch, pitch, K0, row_base, step);
  if (a.L > 1) mlp_fwd_layer<T, 1>(a, in, out, Ws + w1, pitch, pitch, K1, row_base, step);
  if (a.L > 2) mlp_fwd_layer<T, 2>(a, in, out, Ws + w2, pitch, pitch, K2, row_base, step);
  if (a.L > 3) mlp_fwd_layer<T, 3>(a, in, out, Ws + w3, pitch, pitch, K3, row_base, step);
}

// one layer of the backward walk; `off` = start of this layer's block in the partial vector.
// hin = this block's rows of the layer's INPUT activation, mk = its mask bytes (both already in LDS).
template <typename T, int l>
__device__ __forceinline__ void mlp_bwd_layer(const MlpBwdArgs<T>& a, typename AccOf<T>::type*& dcur, typename AccOf<T>::type*& dnext,
                                              const typename AccOf<T>::type* hin, int ph, const uint8_t* mk,
                                              const typename AccOf<T>::type* Ws, int pitch, int K, int off, bool need_dprev) {
  using P = typename AccOf<T>::type;
  using A = P;
  constexpr int kMlpRB = MlpRB<T>::value;
  const int N = a.N[l];
  const float scale = a.drop[l] > 0.f ? 1.0f / (1.0f - a.drop[l]) : 1.0f;
  const uint8_t need = (uint8_t)((a.relu[l] ? 1 : 0) | (a.drop[l] > 0.f ? 2 : 0));
  for (int i = threadIdx.x; i < kMlpRB * N; i += blockDim.x) {       // dz = dh * mask factor (in place)
    const int r = i / N, n = i - r * N;
    const uint8_t m = mk[r * N + n];
    dcur[r * pitch + n] = ((m & need) == need) ? dcur[r * pitch + n] * (A)scale : (A)0;   // rows past B hold zeros already
  }
  __syncthreads();
  P* part = a.part + (long)blockIdx.x * a.total + off;
  for (int i = threadIdx.x; i < N * K; i += blockDim.x) {            // partial dW[n][k] = sum_r dz[r][n] h[r][k]
    const int n = i / K, k = i - n * K;
    part[i] = (P)lds_dot<A>(dcur + n, pitch, hin + k, ph, kMlpRB, (A)0);
  }
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    A s = 0;
    for (int r = 0; r < kMlpRB; ++r) s += dcur[r * pitch + n];
    part[N * K + n] = (P)s;
  }
  if (need_dprev) {
    for (int i = threadIdx.x; i < kMlpRB * K; i += blockDim.x) {     // dprev[r][k] = sum_n dz[r][n] W[n][k]
      const int r = i / K, k = i - r * K;
      const A s = lds_dot<A>(dcur + r * pitch, 1, Ws + k, K + 1, N, (A)0);
      dnext[r * pitch + k] = l == 0 ? s : (A)(T)s;                  // what the lower layer receives is stored in T
    }
  }
  __syncthreads();
  A* t = dcur; dcur = dnext; dnext = t;
}

template <typename T>
__global__ __launch_bounds__(256) void mlp_bwd_kernel(const MlpBwdArgs<T> a, int pitch) {
  using A = typename AccOf<T>::type;
  constexpr int kMlpRB = MlpRB<T>::value;
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  const int row_base = blockIdx.x * kMlpRB;
  const int K0 = a.F, K1 = a.N[0], K2 = a.N[1], K3 = a.N[2];
  const int L = a.L, NL = a.N[L - 1];
  // LDS carve: dcur | dnext | input activations of every layer (x, h0, h1, h2) | all weights | mask bytes
  A* dcur = reinterpret_cast<A*>(smraw);
  A* dnext = dcur + kMlpRB * pitch;
  A* act0 = dnext + kMlpRB * pitch;                 // x rows, pitch K0
  A* act1 = act0 + kMlpRB * K0;                     // h0 rows, pitch K1
  A* act2 = act1 + (L > 1 ? kMlpRB * K1 : 0);
  A* act3 = act2 + (L > 2 ? kMlpRB * K2 : 0);
  A* Ws = act3 + (L > 3 ? kMlpRB * K3 : 0);
  const int w1 = a.N[0] * (K0 + 1), w2 = w1 + (L > 1 ? a.N[1] * (K1 + 1) : 0), w3 = w2 + (L > 2 ? a.N[2] * (K2 + 1) : 0);
  const int wend = w3 + (L > 3 ? a.N[3] * (K3 + 1) : 0);
  uint8_t* mk0 = reinterpret_cast<uint8_t*>(Ws + wend);
  uint8_t* mk1 = mk0 + kMlpRB * a.N[0];
  uint8_t* mk2 = mk1 + (L > 1 ? kMlpRB * a.N[1] : 0);
  uint8_t* mk3 = mk2 + (L > 2 ? kMlpRB * a.N[2] : 0);

  auto stage_rows = [&](const T* src, A* dst, int W_) {
#pragma unroll 8
    for (int i = threadIdx.x; i < kMlpRB * W_; i += blockDim.x) {
      const int r = i / W_;
      dst[i] = row_base + r < a.B ? (A)src[(long)row_base * W_ + i] : (A)0;
    }
  };
  auto stage_mask = [&](const uint8_t* src, uint8_t* dst, int W_) {
#pragma unroll 8
    for (int i = threadIdx.x; i < kMlpRB * W_; i += blockDim.x) {
      const int r = i / W_;
      dst[i] = (src != nullptr && row_base + r < a.B) ? src[(long)row_base * W_ + i] : (uint8_t)3;
    }
  };
#pragma unroll 8
  for (int i = threadIdx.x; i < kMlpRB * NL; i += blockDim.x) {
    const int r = i / NL, n = i - r * NL;
    dcur[r * pitch + n] = row_base + r < a.B ? (A)a.dy[(long)(row_base + r) * NL + n] : (A)0;
  }
  stage_rows(a.x, act0, K0);
  mlp_stage_w<T>(a.W[0], Ws, a.N[0], K0);
  stage_mask(a.mask[0], mk0, a.N[0]);
  if (L > 1) { stage_rows(a.h[0], act1, K1); mlp_stage_w<T>(a.W[1], Ws + w1, a.N[1], K1); stage_mask(a.mask[1], mk1, a.N[1]); }
  if (L > 2) { stage_rows(a.h[1], act2, K2); mlp_stage_w<T>(a.W[2], Ws + w2, a.N[2], K2); stage_mask(a.mask[2], mk2, a.N[2]); }
  if (L > 3) { stage_rows(a.h[2], act3, K3); mlp_stage_w<T>(a.W[3], Ws + w3, a.N[3], K3); stage_mask(a.mask[3], mk3, a.N[3]); }
  __syncthreads();
  const int o0 = 0, o1 = o0 + a.N[0] * (K0 + 1), o2 = o1 + (L > 1 ? a.N[1] * (K1 + 1) : 0), o3 = o2 + (L > 2 ? a.N[2] * (K2 + 1) : 0);
  const bool dxn = a.dx != nullptr;
  if (L > 3) mlp_bwd_layer<T, 3>(a, dcur, dnext, act3, K3, mk3, Ws + w3, pitch, K3, o3, true);
  if (L > 2) mlp_bwd_layer<T, 2>(a, dcur, dnext, act2, K2, mk2, Ws + w2, pitch, K2, o2, true);
  if (L > 1) mlp_bwd_layer<T, 1>(a, dcur, dnext, act1, K1, mk1, Ws + w1, pitch, K1, o1, true);
  mlp_bwd_layer<T, 0>(a, dcur, dnext, act0, K0, mk0, Ws, pitch, K0, o0, dxn);
  if (dxn) {
    for (int i = threadIdx.x; i < kMlpRB * a.F; i += blockDim.x) {
      const int r = i / a.F, k = i - r * a.F;
      if (row_base + r < a.B) a.dx[(long)(row_base + r) * a.F + k] = (T)dcur[r * pitch + k];
    }
  }
}

template <typename P> __global__ void mlp_reduce_kernel(const P* __restrict__ part, const MlpReduceArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.total) return;
  P s = 0;
#pragma unroll 8
  for (int b = 0; b < a.nblk; ++b) s += part[(long)b * a.total + i];
  int off = i;
#define EMB_MLP_OUT(l)                                                         \
  if (a.L > l) {                                                               \
    const int nw = a.N[l] * a.K[l];                                            \
    if (off < nw) { ((P*)a.dW[l])[off] = s; return; }                          \
    off -= nw;                                                                 \
    if (off < a.N[l]) { ((P*)a.db[l])[off] = s; return; }                      \
    off -= a.N[l];                                                             \
  }
  EMB_MLP_OUT(0) EMB_MLP_OUT(1) EMB_MLP_OUT(2) EMB_MLP_OUT(3)
#undef EMB_MLP_OUT
}

static int mlp_pitch(int F, const int* N, int L) {
  int m = F;
  for (int l = 0; l < L; ++l) m = N[l] > m ? N[l] : m;
  return m + 1;
}

static long mlp_w_elems(int F, const int* N, int L) {
  long w = 0;
  int K = F;
  for (int l = 0; l < L; ++l) { w += (long)N[l] * (K + 1); K = N[l]; }
  return w;
}
static size_t mlp_fwd_lds(int F, const int* N, int L, size_t a, int rb) {
  return ((size_t)2 * rb * mlp_pitch(F, N, L) + mlp_w_elems(F, N, L)) * a;
}
static size_t mlp_bwd_lds(int F, const int* N, int L, size_t a, int rb) {
  size_t acts = (size_t)rb * F, masks = 0;
  for (int l = 0; l < L; ++l) { if (l + 1 < L) acts += (size_t)rb * N[l]; masks += (size_t)rb * N[l]; }
  return ((size_t)2 * rb * mlp_pitch(F, N, L) + acts + mlp_w_elems(F, N, L)) * a + ((masks + 15) & ~(size_t)15);
}
static bool mlp_ok(int F, const int* N, int L, int dtype) {
  if (L < 1 || L > kMlpMaxL || F < 1 || F > 1024) return false;
  for (int l = 0; l < L; ++l)
    if (N[l] < 1 || N[l] > 256) return false;
  if (mlp_w_elems(F, N, L) > kMlpWBudget) return false;
  const size_t a = dtype == EMB_F64 ? 8 : 4;
  const int rb = dtype == EMB_F64 ? 16 : 32;
  return mlp_fwd_lds(F, N, L, a, rb) <= 150 * 1024 && mlp_bwd_lds(F, N, L, a, rb) <= 150 * 1024;
}

template <typename T>
static int mlp_fwd_t(const void* x, const void* const* W, const void* const* b, void* const* h, uint8_t* const* mask, const int* N,
                     const int* relu, const float* drop, const int* layer_id, int L, int B, int F, uint64_t seed, uint64_t step_val,
                     const uint64_t* step_dev, int64_t row0, hipStream_t s) {
  using P = typename AccOf<T>::type;
  MlpArgs<T> a{};
  a.x = (const T*)x;
  for (int l = 0; l < L; ++l) {
    a.W[l] = (const T*)W[l]; a.b[l] = (const P*)b[l]; a.h[l] = (T*)h[l]; a.mask[l] = mask[l];
    a.N[l] = N[l]; a.relu[l] = relu[l]; a.drop[l] = drop[l]; a.layer_id[l] = layer_id[l];
  }
  a.B = B; a.F = F; a.L = L; a.seed = seed; a.step_val = step_val; a.step_dev = step_dev; a.row0 = row0;
  const int pitch = mlp_pitch(F, N, L);
  constexpr int kMlpRB = MlpRB<T>::value;
  const size_t lds = (mlp_fwd_lds(F, N, L, sizeof(P), kMlpRB) + 15) & ~(size_t)15;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fwd_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  mlp_fwd_kernel<T><<<cdiv(B, kMlpRB), 256, lds, s>>>(a, pitch);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

template <typename T>
static int mlp_bwd_t(const void* x, const void* const* W, const void* const* h, const uint8_t* const* mask, const void* dy, void* dx,
                     void* const* dW, void* const* db, const int* N, const int* relu, const float* drop, int L, int B, int F, void* ws,
                     int64_t ws_bytes, hipStream_t s) {
  using P = typename AccOf<T>::type;
  MlpBwdArgs<T> a{};
  MlpReduceArgs ra{};
  a.x = (const T*)x; a.dy = (const T*)dy; a.dx = (T*)dx;
  int total = 0, K = F;
  for (int l = 0; l < L; ++l) {
    a.W[l] = (const T*)W[l]; a.h[l] = (const T*)h[l]; a.mask[l] = mask[l];
    a.N[l] = N[l]; a.relu[l] = relu[l]; a.drop[l] = drop[l];
    ra.dW[l] = dW[l]; ra.db[l] = db[l]; ra.N[l] = N[l]; ra.K[l] = K;
    total += N[l] * (K + 1);
    K = N[l];
  }
  constexpr int kMlpRB = MlpRB<T>::value;
  const int nblk = cdiv(B, kMlpRB);
  EMB_CHECK_ARG((int64_t)nblk * total * (int64_t)sizeof(P) <= ws_bytes, "emb_mlp_bwd: workspace too small");
  a.B = B; a.F = F; a.L = L; a.total = total; a.part = (P*)ws;
  ra.L = L; ra.total = total; ra.nblk = nblk;
  const int pitch = mlp_pitch(F, N, L);
  const size_t lds = (mlp_bwd_lds(F, N, L, sizeof(P), kMlpRB) + 15) & ~(size_t)15;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  mlp_bwd_kernel<T><<<nblk, 256, lds, s>>>(a, pitch);
  EMB_CHECK_LAUNCH();
  mlp_reduce_kernel<P><<<cdiv(total, 256), 256, 0, s>>>((const P*)ws, ra);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

}  // namespace emb

using namespace emb;

extern "C" int emb_mlp_supported(int F, const int* N, int L, int dtype) { return mlp_ok(F, N, L, dtype) ? 1 : 0; }

extern "C" int emb_mlp_fwd(const void* x, const void* const* W, const void* const* b, void* const* h, uint8_t* const* mask,
                           const int* N, const int* relu, const float* dropout_p, const int* layer_id, int L, int B, int F,
                           uint64_t seed, uint64_t step_val, const uint64_t* step_dev, int64_t row0, int dtype, emb_stream_t stream) {
  EMB_CHECK_ARG(x && W && b && h && mask && N && relu && dropout_p && layer_id && B > 0, "emb_mlp_fwd: bad argument");
  EMB_CHECK_ARG(mlp_ok(F, N, L, dtype), "emb_mlp_fwd: stack not eligible for the fused kernel (see emb_mlp_supported)");
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case EMB_F32: return mlp_fwd_t<float>(x, W, b, h, mask, N, relu, dropout_p, layer_id, L, B, F, seed, step_val, step_dev, row0, s);
    case EMB_BF16: return mlp_fwd_t<__bf16>(x, W, b, h, mask, N, relu, dropout_p, layer_id, L, B, F, seed, step_val, step_dev, row0, s);
    case EMB_F64: return mlp_fwd_t<double>(x, W, b, h, mask, N, relu, dropout_p, layer_id, L, B, F, seed, step_val, step_dev, row0, s);
  }
  set_error("emb_mlp_fwd: unsupported dtype %d", dtype);
  return EMB_ERR_DTYPE;
}

extern "C" int emb_mlp_bwd(const void* x, const void* const* W, const void* const* h, const uint8_t* const* mask, const void* dy,
                           void* dx, void* const* dW, void* const* db, const int* N, const int* relu, const float* dropout_p, int L,
                           int B, int F, void* workspace, int64_t workspace_bytes, int dtype, emb_stream_t stream) {
  EMB_CHECK_ARG(x && W && h && mask && dy && dW && db && N && relu && dropout_p && workspace && B > 0, "emb_mlp_bwd: bad argument");
  EMB_CHECK_ARG(mlp_ok(F, N, L, dtype), "emb_mlp_bwd: stack not eligible for the fused kernel");
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case EMB_F32: return mlp_bwd_t<float>(x, W, h, mask, dy, dx, dW, db, N, relu, dropout_p, L, B, F, workspace, workspace_bytes, s);
    case EMB_BF16: return mlp_bwd_t<__bf16>(x, W, h, mask, dy, dx, dW, db, N, relu, dropout_p, L, B, F, workspace, workspace_bytes, s);
    case EMB_F64: return mlp_bwd_t<double>(x, W, h, mask, dy, dx, dW, db, N, relu, dropout_p, L, B, F, workspace, workspace_bytes, s);
  }
  set_error("emb_mlp_bwd: unsupported dtype %d", dtype);
  return EMB_ERR_DTYPE;
}
