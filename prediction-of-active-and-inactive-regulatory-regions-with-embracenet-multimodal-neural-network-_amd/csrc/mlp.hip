// Small multi-layer perceptron stacks in ONE launch (forward) / TWO launches (backward):
//   epigenomic pre-network  FFNN_pre.py:18-49   1-4 x (Linear -> ReLU -> Dropout), widths <= 256
//   post stack / head       EmbraceNetMultimodal.py:134-154   0-2 x (Linear -> ReLU -> Dropout) + Linear(-> 2)
// These layers are a few thousand MACs per row: as separate GEMM launches they cost ~5 us each of pure launch
// and fill/drain latency (12 launches per step for a 3-layer FFNN).  Here one 1024-thread workgroup carries 16
// rows through the whole stack.  Everything it needs (its rows, ALL weights, for the backward also every stored
// activation and mask byte) is requested from global memory up front as 16-byte vectors held in registers --
// one memory round trip -- and then unpacked into LDS; the layers run out of LDS with plain FMAs (the matrices
// are far too small for MFMA tiles to pay; 16 waves per CU hide the LDS latency).  Narrow layers (the 2-class
// head) split each dot product over a group of lanes and meet with wave shuffles.  The backward walks the stack
// in reverse producing dX and per-workgroup partial dW/db, which a second tiny launch sums in fixed order
// (deterministic).  Eligibility is checked by emb_mlp_supported(); larger layers use the GEMM kernels of linear.hip.
#include "common.h"
#include "philox.h"
#include "reduce.h"
#include "mlp_args.h"
#include "rider.h"

namespace emb {

struct MlpReduceArgs {
  void* dW[kMlpMaxL];
  void* db[kMlpMaxL];
  int N[kMlpMaxL], K[kMlpMaxL];
  int L, total, nblk;
};

#ifdef EMB_MLP_PROF
__device__ unsigned long long g_mlp_prof[32];
#define MLP_T(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_mlp_prof[i] = wall_clock64(); } while (0)
#else
#define MLP_T(i) do {} while (0)
#endif

__host__ __device__ __forceinline__ int mlp_wpitch(int K) { return K | 1; }   // odd: conflict-free across rows of W

// sum_j a[j*sa] * b[j*sb], 4 independent partial sums (the LDS reads of consecutive iterations overlap)
template <typename A>
__device__ __forceinline__ A lds_dot(const A* __restrict__ a, int sa, const A* __restrict__ b, int sb, int n) {
  A s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  int k = 0;
  for (; k + 4 <= n; k += 4) {
    const A a0 = a[(k + 0) * sa], a1 = a[(k + 1) * sa], a2 = a[(k + 2) * sa], a3 = a[(k + 3) * sa];
    const A b0 = b[(k + 0) * sb], b1 = b[(k + 1) * sb], b2 = b[(k + 2) * sb], b3 = b[(k + 3) * sb];
    s0 += a0 * b0; s1 += a1 * b1; s2 += a2 * b2; s3 += a3 * b3;
  }
  for (; k < n; ++k) s0 += a[k * sa] * b[k * sb];
  return (s0 + s1) + (s2 + s3);
}

__device__ __forceinline__ bool dev_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- two-phase staging: request (registers) ... unpack (LDS) -------------------------------------------------
// request up to R 16-byte vectors per thread of src[0..n)
template <typename T, int R>
__device__ __forceinline__ void stage_issue(typename Vec16<T>::type (&v)[R], const T* __restrict__ src, int n) {
  using V = typename Vec16<T>::type;
  const int nv = dev_aligned16(src) ? n / Elem<T>::VEC : 0;
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int i = j * kMlpThreads + (int)threadIdx.x;
    if (i < nv) v[j] = reinterpret_cast<const V*>(src)[i];
  }
}
// unpack into LDS rows of `width` elements (row pitch `pitch`), converted to the accumulation type; whatever
// the vectors did not cover (unaligned source, tail, more than R*1024 vectors) is read by the scalar loop, and
// elements n..total-1 (rows past the end of the batch) are zero-filled
template <typename T, typename A, int R>
__device__ __forceinline__ void stage_commit(const typename Vec16<T>::type (&v)[R], const T* __restrict__ src, int n, A* dst,
                                             int width, int pitch, int total) {
  constexpr int VEC = Elem<T>::VEC;
  const int nv = dev_aligned16(src) ? min(n / VEC, R * kMlpThreads) : 0;
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int i = j * kMlpThreads + (int)threadIdx.x;
    if (i < nv) {
      int row = (i * VEC) / width, col = i * VEC - row * width;
#pragma unroll
      for (int q = 0; q < VEC; ++q) {
        dst[row * pitch + col] = (A)v[j][q];
        if (++col == width) { col = 0; ++row; }
      }
    }
  }
  for (int e = nv * VEC + (int)threadIdx.x; e < total; e += kMlpThreads) {
    const int row = e / width, col = e - row * width;
    dst[row * pitch + col] = e < n ? (A)src[e] : (A)0;
  }
}
// mask bytes: one 16-byte vector per thread covers 16 rows x 1024 bytes; absent mask / rows past the end read as 3
__device__ __forceinline__ void mask_issue(uint4& v, const uint8_t* __restrict__ src, int n) {
  const int nv = (src != nullptr && dev_aligned16(src)) ? n / 16 : 0;
  if ((int)threadIdx.x < nv) v = reinterpret_cast<const uint4*>(src)[threadIdx.x];
}
__device__ __forceinline__ void mask_commit(const uint4& v, const uint8_t* __restrict__ src, int n, uint8_t* dst, int total) {
  const int nv = (src != nullptr && dev_aligned16(src)) ? min(n / 16, kMlpThreads) : 0;
  if ((int)threadIdx.x < nv) reinterpret_cast<uint4*>(dst)[threadIdx.x] = v;   // dst is 16-byte aligned (carved so)
  for (int e = nv * 16 + (int)threadIdx.x; e < total; e += kMlpThreads) dst[e] = (src != nullptr && e < n) ? src[e] : (uint8_t)3;
}

template <typename T> struct MlpStageR {   // vectors per thread held for the input rows / for one weight matrix
  static constexpr int X = 2, W = sizeof(T) == 2 ? 2 : 3;
};

// ---- forward -----------------------------------------------------------------------------------------------------
template <typename T, int l>
__device__ __forceinline__ void mlp_fwd_layer(const MlpArgs<T>& a, typename AccOf<T>::type*& in, typename AccOf<T>::type*& out,
                                              const typename AccOf<T>::type* Ws, int pitch, int K, int row_base, uint64_t step) {
  using A = typename AccOf<T>::type;
  const int N = a.N[l], Kp = mlp_wpitch(K), items = kMlpRB * N;
  const float p = a.drop[l];
  const float keep_scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  const uint64_t stream = rng_stream(step, EMB_RNG_DROPOUT0 + a.layer_id[l]);
  // S lanes share one output when the layer has fewer outputs than the workgroup has threads
  int S = 1, lgS = 0;
  while (S < 64 && items * S * 2 <= kMlpThreads && S * 2 <= K) { S *= 2; ++lgS; }
  for (int it = threadIdx.x; it < items * S; it += kMlpThreads) {
    const int s = it & (S - 1), o = it >> lgS;
    const int r = o / N, n = o - r * N, row = row_base + r;
    A acc = lds_dot<A>(in + r * pitch + s, S, Ws + n * Kp + s, S, (K - s + S - 1) >> lgS);
    for (int m = S >> 1; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
    if (s == 0) {
      acc += (A)a.b[l][n];
      bool act = true, keep = true;
      if (a.relu[l]) {
        act = acc > (A)0;
        acc = act ? acc : (A)0;
      }
      if (p > 0.f && row < a.B) {
        keep = uniform24(philox4x32_10(a.seed, stream, (uint64_t)(a.row0 + row) * (uint64_t)N + n).x) >= p;
        acc = keep ? acc * (A)keep_scale : (A)0;
      }
      const T hv = (T)acc;
      if (row < a.B) {
        a.h[l][(long)row * N + n] = hv;
        if (a.mask[l]) a.mask[l][(long)row * N + n] = (uint8_t)((act ? 1 : 0) | (keep ? 2 : 0));
      }
      out[r * pitch + n] = (A)hv;          // the next layer sees what the stored activation holds
    }
  }
  __syncthreads();
  A* t = in; in = out; out = t;
}

template <typename T>
__global__ __launch_bounds__(kMlpThreads) void mlp_fwd_kernel(const MlpArgs<T> a, int pitch) {
  using A = typename AccOf<T>::type;
  using V = typename Vec16<T>::type;
  constexpr int XR = MlpStageR<T>::X, WR = MlpStageR<T>::W;
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  A* in = reinterpret_cast<A*>(smraw);
  A* out = in + kMlpRB * pitch;
  A* Ws = out + kMlpRB * pitch;
  const int row_base = blockIdx.x * kMlpRB, nrows = min(kMlpRB, a.B - row_base);
  MLP_T(0);
  const int L = a.L, K0 = a.F, K1 = a.N[0], K2 = a.N[1], K3 = a.N[2];
  const int w1 = a.N[0] * mlp_wpitch(K0), w2 = w1 + (L > 1 ? a.N[1] * mlp_wpitch(K1) : 0), w3 = w2 + (L > 2 ? a.N[2] * mlp_wpitch(K2) : 0);
  const T* xs = a.x + (long)row_base * a.F;
  V vx[XR], vw0[WR], vw1[WR], vw2[WR], vw3[WR];
  stage_issue<T, XR>(vx, xs, nrows * a.F);
  stage_issue<T, WR>(vw0, a.W[0], a.N[0] * K0);
  if (L > 1) stage_issue<T, WR>(vw1, a.W[1], a.N[1] * K1);
  if (L > 2) stage_issue<T, WR>(vw2, a.W[2], a.N[2] * K2);
  if (L > 3) stage_issue<T, WR>(vw3, a.W[3], a.N[3] * K3);
  const uint64_t step = a.step_val + (a.step_dev ? *a.step_dev : 0);
  stage_commit<T, A, XR>(vx, xs, nrows * a.F, in, a.F, pitch, kMlpRB * a.F);
  stage_commit<T, A, WR>(vw0, a.W[0], a.N[0] * K0, Ws, K0, mlp_wpitch(K0), a.N[0] * K0);
  if (L > 1) stage_commit<T, A, WR>(vw1, a.W[1], a.N[1] * K1, Ws + w1, K1, mlp_wpitch(K1), a.N[1] * K1);
  if (L > 2) stage_commit<T, A, WR>(vw2, a.W[2], a.N[2] * K2, Ws + w2, K2, mlp_wpitch(K2), a.N[2] * K2);
  if (L > 3) stage_commit<T, A, WR>(vw3, a.W[3], a.N[3] * K3, Ws + w3, K3, mlp_wpitch(K3), a.N[3] * K3);
  __syncthreads();
  MLP_T(1);
  mlp_fwd_layer<T, 0>(a, in, out, Ws, pitch, K0, row_base, step);
  MLP_T(2);
  if (L > 1) mlp_fwd_layer<T, 1>(a, in, out, Ws + w1, pitch, K1, row_base, step);
  MLP_T(3);
  if (L > 2) mlp_fwd_layer<T, 2>(a, in, out, Ws + w2, pitch, K2, row_base, step);
  MLP_T(4);
  if (L > 3) mlp_fwd_layer<T, 3>(a, in, out, Ws + w3, pitch, K3, row_base, step);
  MLP_T(5);
}

// ---- backward ----------------------------------------------------------------------------------------------------
// one layer of the backward walk; `off` = start of this layer's block in the partial vector.
// hin = this block's rows of the layer's INPUT activation (pitch K), mk = its mask bytes (both already in LDS).
template <typename T, int l>
__device__ __forceinline__ void mlp_bwd_layer(const MlpBwdArgs<T>& a, typename AccOf<T>::type*& dcur, typename AccOf<T>::type*& dnext,
                                              const typename AccOf<T>::type* hin, const uint8_t* mk,
                                              const typename AccOf<T>::type* Ws, int pitch, int K, int off, bool need_dprev) {
  using A = typename AccOf<T>::type;
  const int N = a.N[l], Kp = mlp_wpitch(K);
  const float scale = a.drop[l] > 0.f ? 1.0f / (1.0f - a.drop[l]) : 1.0f;
  const uint8_t need = (uint8_t)((a.relu[l] ? 1 : 0) | (a.drop[l] > 0.f ? 2 : 0));
  for (int i = threadIdx.x; i < kMlpRB * N; i += kMlpThreads) {       // dz = dh * mask factor (in place)
    const int r = i / N, n = i - r * N;
    const uint8_t m = mk[i];
    dcur[r * pitch + n] = ((m & need) == need) ? dcur[r * pitch + n] * (A)scale : (A)0;   // rows past B hold zeros already
  }
  __syncthreads();
  A* part = a.part + (long)blockIdx.x * a.total + off;
  for (int i = threadIdx.x; i < N * K; i += kMlpThreads) {            // partial dW[n][k] = sum_r dz[r][n] h[r][k]
    const int n = i / K, k = i - n * K;
    part[i] = lds_dot<A>(dcur + n, pitch, hin + k, K, kMlpRB);
  }
  for (int n = threadIdx.x; n < N; n += kMlpThreads) {
    A s = 0;
#pragma unroll
    for (int r = 0; r < kMlpRB; ++r) s += dcur[r * pitch + n];
    part[N * K + n] = s;
  }
  if (need_dprev) {
    for (int i = threadIdx.x; i < kMlpRB * K; i += kMlpThreads) {     // dprev[r][k] = sum_n dz[r][n] W[n][k]
      const int r = i / K, k = i - r * K;
      const A s = lds_dot<A>(dcur + r * pitch, 1, Ws + k, Kp, N);
      dnext[r * pitch + k] = l == 0 ? s : (A)(T)s;                   // what the lower layer receives is stored in T
    }
  }
  __syncthreads();
  A* t = dcur; dcur = dnext; dnext = t;
}

template <typename T>
__global__ __launch_bounds__(kMlpThreads) void mlp_bwd_kernel(const MlpBwdArgs<T> a, int pitch) {
  using A = typename AccOf<T>::type;
  using V = typename Vec16<T>::type;
  constexpr int XR = MlpStageR<T>::X, WR = MlpStageR<T>::W;
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  const int row_base = blockIdx.x * kMlpRB, nrows = min(kMlpRB, a.B - row_base);
  const int K0 = a.F, K1 = a.N[0], K2 = a.N[1], K3 = a.N[2];
  const int L = a.L, NL = a.N[L - 1];
  // LDS carve: mask bytes (16-byte aligned slots) | dcur | dnext | input activations of every layer | all weights
  uint8_t* mk0 = reinterpret_cast<uint8_t*>(smraw);
  uint8_t* mk1 = mk0 + kMlpRB * a.N[0];                      // kMlpRB = 16: every slot is a multiple of 16 bytes
  uint8_t* mk2 = mk1 + (L > 1 ? kMlpRB * a.N[1] : 0);
  uint8_t* mk3 = mk2 + (L > 2 ? kMlpRB * a.N[2] : 0);
  A* dcur = reinterpret_cast<A*>(mk3 + (L > 3 ? kMlpRB * a.N[3] : 0));
  A* dnext = dcur + kMlpRB * pitch;
  A* act0 = dnext + kMlpRB * pitch;                 // x rows, pitch K0
  A* act1 = act0 + kMlpRB * K0;                     // h0 rows, pitch K1
  A* act2 = act1 + (L > 1 ? kMlpRB * K1 : 0);
  A* act3 = act2 + (L > 2 ? kMlpRB * K2 : 0);
  A* Ws = act3 + (L > 3 ? kMlpRB * K3 : 0);
  const int w1 = a.N[0] * mlp_wpitch(K0), w2 = w1 + (L > 1 ? a.N[1] * mlp_wpitch(K1) : 0), w3 = w2 + (L > 2 ? a.N[2] * mlp_wpitch(K2) : 0);
  MLP_T(8);
  const long rb = row_base;
  const T* dys = a.dy + rb * NL;
  const T* xs = a.x + rb * K0;
  const T* h0s = L > 1 ? a.h[0] + rb * K1 : nullptr;
  const T* h1s = L > 2 ? a.h[1] + rb * K2 : nullptr;
  const T* h2s = L > 3 ? a.h[2] + rb * K3 : nullptr;
  const uint8_t* m0s = a.mask[0] ? a.mask[0] + rb * a.N[0] : nullptr;
  const uint8_t* m1s = (L > 1 && a.mask[1]) ? a.mask[1] + rb * a.N[1] : nullptr;
  const uint8_t* m2s = (L > 2 && a.mask[2]) ? a.mask[2] + rb * a.N[2] : nullptr;
  const uint8_t* m3s = (L > 3 && a.mask[3]) ? a.mask[3] + rb * a.N[3] : nullptr;
  V vdy[1], vx[XR], vh0[1], vh1[1], vh2[1], vw0[WR], vw1[WR], vw2[WR], vw3[WR];
  uint4 vm0, vm1, vm2, vm3;
  stage_issue<T, 1>(vdy, dys, nrows * NL);
  stage_issue<T, XR>(vx, xs, nrows * K0);
  stage_issue<T, WR>(vw0, a.W[0], a.N[0] * K0);
  mask_issue(vm0, m0s, nrows * a.N[0]);
  if (L > 1) { stage_issue<T, 1>(vh0, h0s, nrows * K1); stage_issue<T, WR>(vw1, a.W[1], a.N[1] * K1); mask_issue(vm1, m1s, nrows * a.N[1]); }
  if (L > 2) { stage_issue<T, 1>(vh1, h1s, nrows * K2); stage_issue<T, WR>(vw2, a.W[2], a.N[2] * K2); mask_issue(vm2, m2s, nrows * a.N[2]); }
  if (L > 3) { stage_issue<T, 1>(vh2, h2s, nrows * K3); stage_issue<T, WR>(vw3, a.W[3], a.N[3] * K3); mask_issue(vm3, m3s, nrows * a.N[3]); }
  stage_commit<T, A, 1>(vdy, dys, nrows * NL, dcur, NL, pitch, kMlpRB * NL);
  stage_commit<T, A, XR>(vx, xs, nrows * K0, act0, K0, K0, kMlpRB * K0);
  stage_commit<T, A, WR>(vw0, a.W[0], a.N[0] * K0, Ws, K0, mlp_wpitch(K0), a.N[0] * K0);
  mask_commit(vm0, m0s, nrows * a.N[0], mk0, kMlpRB * a.N[0]);
  if (L > 1) {
    stage_commit<T, A, 1>(vh0, h0s, nrows * K1, act1, K1, K1, kMlpRB * K1);
    stage_commit<T, A, WR>(vw1, a.W[1], a.N[1] * K1, Ws + w1, K1, mlp_wpitch(K1), a.N[1] * K1);
    mask_commit(vm1, m1s, nrows * a.N[1], mk1, kMlpRB * a.N[1]);
  }
  if (L > 2) {
    stage_commit<T, A, 1>(vh1, h1s, nrows * K2, act2, K2, K2, kMlpRB * K2);
    stage_commit<T, A, WR>(vw2, a.W[2], a.N[2] * K2, Ws + w2, K2, mlp_wpitch(K2), a.N[2] * K2);
    mask_commit(vm2, m2s, nrows * a.N[2], mk2, kMlpRB * a.N[2]);
  }
  if (L > 3) {
    stage_commit<T, A, 1>(vh2, h2s, nrows * K3, act3, K3, K3, kMlpRB * K3);
    stage_commit<T, A, WR>(vw3, a.W[3], a.N[3] * K3, Ws + w3, K3, mlp_wpitch(K3), a.N[3] * K3);
    mask_commit(vm3, m3s, nrows * a.N[3], mk3, kMlpRB * a.N[3]);
  }
  __syncthreads();
  MLP_T(9);
  const int o0 = 0, o1 = o0 + a.N[0] * (K0 + 1), o2 = o1 + (L > 1 ? a.N[1] * (K1 + 1) : 0), o3 = o2 + (L > 2 ? a.N[2] * (K2 + 1) : 0);
  const bool dxn = a.dx != nullptr;
  if (L > 3) mlp_bwd_layer<T, 3>(a, dcur, dnext, act3, mk3, Ws + w3, pitch, K3, o3, true);
  MLP_T(10);
  if (L > 2) mlp_bwd_layer<T, 2>(a, dcur, dnext, act2, mk2, Ws + w2, pitch, K2, o2, true);
  MLP_T(11);
  if (L > 1) mlp_bwd_layer<T, 1>(a, dcur, dnext, act1, mk1, Ws + w1, pitch, K1, o1, true);
  MLP_T(12);
  mlp_bwd_layer<T, 0>(a, dcur, dnext, act0, mk0, Ws, pitch, K0, o0, dxn);
  MLP_T(13);
  if (dxn) {
    for (int i = threadIdx.x; i < nrows * a.F; i += kMlpThreads) {
      const int r = i / a.F, k = i - r * a.F;
      a.dx[rb * a.F + i] = (T)dcur[r * pitch + k];
    }
  }
  MLP_T(14);
}

}  // namespace emb
#include "mlp_mfma.h"
namespace emb {

static bool mlp_mfma_enabled() {
  constexpr bool on = true;
  return on;
}

static int mlp_pitch(int F, const int* N, int L) {
  int m = F;
  for (int l = 0; l < L; ++l) m = N[l] > m ? N[l] : m;
  return m + 1;
}

static long mlp_w_elems(int F, const int* N, int L) {
  long w = 0;
  int K = F;
  for (int l = 0; l < L; ++l) { w += (long)N[l] * mlp_wpitch(K); K = N[l]; }
  return w;
}
static long mlp_part_elems(int F, const int* N, int L) {   // one workgroup's partial dW/db vector
  long t = 0;
  int K = F;
  for (int l = 0; l < L; ++l) { t += (long)N[l] * (K + 1); K = N[l]; }
  return t;
}
static size_t mlp_fwd_lds(int F, const int* N, int L, size_t a, int rb = kMlpRB) {
  return ((size_t)2 * rb * mlp_pitch(F, N, L) + mlp_w_elems(F, N, L)) * a;
}
static size_t mlp_bwd_lds(int F, const int* N, int L, size_t a, int rb = kMlpRB) {
  size_t acts = (size_t)rb * F, masks = 0;
  for (int l = 0; l < L; ++l) { if (l + 1 < L) acts += (size_t)rb * N[l]; masks += (size_t)rb * N[l]; }
  return ((size_t)2 * rb * mlp_pitch(F, N, L) + acts + mlp_w_elems(F, N, L)) * a + ((masks + 15) & ~(size_t)15);
}
static bool mlp_ok(int F, const int* N, int L, int dtype) {
  if (L < 1 || L > kMlpMaxL || F < 1 || F > 1024) return false;
  for (int l = 0; l < L; ++l)
    if (N[l] < 1 || N[l] > 256) return false;
  if (mlp_w_elems(F, N, L) > kMlpWBudget) return false;
  if (dtype != EMB_F32 && dtype != EMB_BF16 && dtype != EMB_F64) return false;
  const size_t a = dtype == EMB_F64 ? 8 : 4;
  return mlp_fwd_lds(F, N, L, a) <= 150 * 1024 && mlp_bwd_lds(F, N, L, a) <= 150 * 1024;
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
  if constexpr (sizeof(T) == 2) {   // bf16, MFMA-shaped widths: one wave per 16 rows on the matrix cores (mlp_mfma.h)
    bool al = aligned16(x);
    for (int l = 0; l < L; ++l) al = al && aligned16(W[l]) && aligned16(h[l]) && (mask[l] == nullptr || aligned16(mask[l]));
    MmFwdLayout lay{};
    if (mlp_mfma_enabled() && al && mlp_mfma_ok(F, N, L)) {
      const size_t lds = mlp_mfma_fwd_layout(F, N, L, &lay);
      if (lds <= 150 * 1024) {
        Rider r{};
        r.kind = RIDER_MLP_FWD; r.nwg = cdiv(B, 16); r.lds = lds; r.stream = s; r.fa = a; r.fl = lay;
        if (rider_deferring(s)) {   // carried by the next suitable launch of this stream (rider.h)
          rider_park(r);
          return EMB_OK;
        }
        const int rc = rider_launch(r);
        if (rc != EMB_OK) return rc;
        return EMB_OK;
      }
    }
  }
  const int pitch = mlp_pitch(F, N, L);
  const size_t lds = (mlp_fwd_lds(F, N, L, sizeof(P)) + 15) & ~(size_t)15;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fwd_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  mlp_fwd_kernel<T><<<cdiv(B, kMlpRB), kMlpThreads, lds, s>>>(a, pitch);
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
  const int nblk = cdiv(B, kMlpRB);
  EMB_CHECK_ARG((int64_t)nblk * total * (int64_t)sizeof(P) <= ws_bytes, "emb_mlp_bwd: workspace too small");
  a.B = B; a.F = F; a.L = L; a.total = total; a.part = (P*)ws;
  ra.L = L; ra.total = total; ra.nblk = nblk;
  bool launched = false;
  if constexpr (sizeof(T) == 2) {   // (kMlpRB == 16: the partial-sum layout is the scalar kernel's)
    bool al = aligned16(x) && aligned16(dy) && (dx == nullptr || aligned16(dx));
    for (int l = 0; l < L; ++l) al = al && aligned16(W[l]) && aligned16(h[l]) && (mask[l] == nullptr || aligned16(mask[l]));
    MmBwdLayout lay{};
    if (mlp_mfma_enabled() && al && mlp_mfma_ok(F, N, L)) {
      const size_t lds = mlp_mfma_bwd_layout(F, N, L, &lay);
      if (lds <= 150 * 1024) {
        Rider r{};
        r.kind = RIDER_MLP_BWD; r.nwg = nblk; r.lds = lds; r.stream = s; r.ba = a; r.bl = lay;
        if (rider_deferring(s) && reduce_deferring(s)) {   // (an immediate reduction launch below would overtake a parked producer)
          rider_park(r);
        } else {
          const int rc = rider_launch(r);
          if (rc != EMB_OK) return rc;
        }
        launched = true;
      }
    }
  }
  const int pitch = mlp_pitch(F, N, L);
  const size_t lds = (mlp_bwd_lds(F, N, L, sizeof(P)) + 15) & ~(size_t)15;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  if (!launched) {
    mlp_bwd_kernel<T><<<nblk, kMlpThreads, lds, s>>>(a, pitch);
    EMB_CHECK_LAUNCH();
  }
  ReduceJob j{};   // per-workgroup partials -> dW_l / db_l, workgroups summed in fixed order (reduce.hip)
  j.in = ws; j.per = total; j.S = nblk; j.kind = RJ_MLP; j.iv[0] = L;
  for (int l = 0; l < L; ++l) { j.out[l] = ra.dW[l]; j.out[4 + l] = ra.db[l]; j.iv[1 + l] = ra.N[l]; j.iv[5 + l] = ra.K[l]; }
  return reduce_submit(j, sizeof(P) == 8, s);
}

}  // namespace emb

using namespace emb;

extern "C" int emb_mlp_supported(int F, const int* N, int L, int dtype) { return (N && mlp_ok(F, N, L, dtype)) ? 1 : 0; }

extern "C" int64_t emb_mlp_workspace_bytes(int F, const int* N, int L, int B, int dtype) {
  if (!N || B < 1 || !mlp_ok(F, N, L, dtype)) return 0;
  return (int64_t)cdiv(B, kMlpRB) * mlp_part_elems(F, N, L) * (dtype == EMB_F64 ? 8 : 4);
}

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

#ifdef EMB_MLP_PROF
extern "C" int emb_debug_mlp_prof(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(emb::g_mlp_prof), sizeof(unsigned long long) * 32);
}
#endif
