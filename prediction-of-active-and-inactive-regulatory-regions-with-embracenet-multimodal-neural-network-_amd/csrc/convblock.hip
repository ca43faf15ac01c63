// One block of the sequence pre-network: Conv1d(same padding) -> BatchNorm1d -> ReLU -> MaxPool1d(10, 2)
// [-> Dropout], forward and backward.  Reference: BIOINF_tesi/models/CNN_pre.py:37-50 (83 % of the
// model's FLOPs).
//
// Activations are channels-last ("NLC", x[b][t][c]).  With that layout the im2col matrix of a same-padded
// 1-D convolution is a plain strided VIEW of x: row R = b*L + t, column KK = tap*cin + ci lives at
// x + (R - pad)*cin + KK, and only validity (0 <= t - pad + tap < L) has to be checked.  So the three
// contractions of a block are ordinary MFMA GEMMs on the shared tile core (gemm_core.h), no im2col buffer:
//   forward   y[R, Cout]   = view(x)[R, k*cin]      . wpack[Cout, k*cin]^T       (+ bias, + BN partial sums)
//   dgrad     dx[R, cin]   = view(dy)[R, k*Cout]    . wflip[cin, k*Cout]^T       (taps flipped)
//   wgrad     dW[Cout, k*cin] = dy^T[Cout, R]       . view(x)^T[k*cin, R]        (split over R, fixed-order reduce)
// BatchNorm batch statistics come from per-tile partial sums written by the forward GEMM's epilogue and are
// finalised in double; BN + ReLU + max-pool (+ dropout) is one elementwise pass; the backward recomputes the
// pooled gradient by a deterministic gather (no atomics anywhere: results are bitwise reproducible).
#include "conv_direct.h"
#include "gemm_jobs_api.h"
#include "conv_first.h"
#include "first_fin.h"
#include "rider.h"
#include "reduce.h"
#include "gemm_tile.h"

namespace emb {

template <typename T> constexpr int dtype_code() { return sizeof(T) == 2 ? EMB_BF16 : (sizeof(T) == 4 ? EMB_F32 : EMB_F64); }

template <typename T> struct ConvCfg;
template <> struct ConvCfg<__bf16> {
  using F64 = TileCfg<__bf16, 128, 64, 64, 4, 1, 1, false, false>;
  using F32 = TileCfg<__bf16, 128, 32, 64, 4, 1, 1, false, false>;
  using W = TileCfg<__bf16, 64, 64, 64, 2, 2, 1, true, true>;
};
template <> struct ConvCfg<float> {
  using F64 = TileCfg<float, 128, 64, 32, 4, 1, 1, false, false>;
  using F32 = TileCfg<float, 128, 32, 32, 4, 1, 1, false, false>;
  using W = TileCfg<float, 64, 64, 32, 2, 2, 1, true, true>;
};
template <> struct ConvCfg<double> {
  using F64 = TileCfg<double, 128, 64, 16, 4, 1, 1, false, false>;
  using F32 = TileCfg<double, 128, 32, 16, 4, 1, 1, false, false>;
  using W = TileCfg<double, 64, 64, 16, 2, 2, 1, true, true>;
};

constexpr int kPoolK = 10, kPoolS = 2;   // CNN_pre.py:17,19

// ------------------------------------------------------------------------------------ conv GEMM
// FWD: C = view(x).wpack^T + bias, per-tile column sums of C and C^2 -> partial[tm][2][N].   !FWD: plain store.
template <class Cfg, bool FWD>
__global__ __launch_bounds__(kThreads, 2) void conv_gemm_kernel(const typename Cfg::T* __restrict__ x,
                                                             const typename Cfg::T* __restrict__ w,
                                                             const typename Cfg::M::Acc* __restrict__ bias,
                                                             typename Cfg::T* __restrict__ out,
                                                             typename Cfg::M::Acc* __restrict__ partial, int R, int L, int cin,
                                                             int KK, int N, int pad, int tiles_n, int ntiles, int vec_x,
                                                             int vec_w, int vec_o) {
  using T = typename Cfg::T;
  using Mm = typename Cfg::M;
  using Acc = typename Mm::Acc;
  extern __shared__ __attribute__((aligned(16))) char arena[];
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int row0 = tm * Cfg::BM, col0 = tn * Cfg::BN;
  typename Mm::AccV acc[Cfg::MI][Cfg::NI];
  zero_acc<Cfg>(acc);
  Stager<T, false, Cfg::BM, Cfg::BK, XfNone> sa{x, nullptr, cin, row0, R, KK, vec_x != 0, XfNone{}, -1, L, cin, pad};
  Stager<T, false, Cfg::BN, Cfg::BK, XfNone> sb{w, nullptr, KK, col0, N, KK, vec_w != 0, XfNone{}, -1, 0, 0, 0};
  gemm_mainloop<Cfg>(sa, sb, KK, arena, acc);
  Acc* cs = reinterpret_cast<Acc*>(arena);
  reduce_to_slab<Cfg>(acc, cs);
  constexpr int GROUPS = Cfg::BM * Cfg::BN / 4;
  for (int gidx = threadIdx.x; gidx < GROUPS; gidx += kThreads) {
    const int r = gidx / (Cfg::BN / 4), cq = (gidx % (Cfg::BN / 4)) * 4;
    const int row = row0 + r, col = col0 + cq;
    Acc v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = cs[r * Cfg::CS + cq + j];
      if (FWD) {
        v[j] += bias[min(col + j, N - 1)];
        cs[r * Cfg::CS + cq + j] = (row < R && col + j < N) ? v[j] : (Acc)0;   // what the statistics pass sums
      }
    }
    if (row >= R || col >= N) continue;
    T* dst = out + (long)row * N + col;
    if (col + 4 <= N && vec_o) {
      typedef T TV4 __attribute__((ext_vector_type(4)));
      TV4 o = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
      *reinterpret_cast<TV4*>(dst) = o;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (col + j < N) dst[j] = (T)v[j];
    }
  }
  if (FWD) {
    __syncthreads();
    // column sums over the tile's rows: thread (part, col) sums rows part, part+PARTS, ...; parts combined in order
    constexpr int PARTS = kThreads / Cfg::BN;
    Acc* red = cs + Cfg::SLAB;   // [2][PARTS][BN]
    const int colr = threadIdx.x % Cfg::BN, part = threadIdx.x / Cfg::BN;
    Acc s = 0, s2 = 0;
    for (int r = part; r < Cfg::BM; r += PARTS) {
      const Acc v = cs[r * Cfg::CS + colr];
      s += v;
      s2 += v * v;
    }
    red[part * Cfg::BN + colr] = s;
    red[(PARTS + part) * Cfg::BN + colr] = s2;
    __syncthreads();
    if (part == 0 && col0 + colr < N) {
      Acc a = 0, b = 0;
      for (int p = 0; p < PARTS; ++p) {
        a += red[p * Cfg::BN + colr];
        b += red[(PARTS + p) * Cfg::BN + colr];
      }
      partial[((long)tm * 2 + 0) * N + col0 + colr] = a;
      partial[((long)tm * 2 + 1) * N + col0 + colr] = b;
    }
  }
}

template <class Cfg> constexpr int conv_gemm_lds() {
  constexpr int slab = (Cfg::SLAB + 2 * kThreads) * (int)sizeof(typename Cfg::M::Acc);
  return Cfg::OPERAND_BYTES > slab ? Cfg::OPERAND_BYTES : slab;
}

template <class Cfg, bool FWD>
static int launch_conv_gemm(const void* x, const void* w, const void* bias, void* out, void* partial, int R, int L, int cin,
                            int KK, int N, int pad, hipStream_t s) {
  using T = typename Cfg::T;
  using Acc = typename Cfg::M::Acc;
  constexpr int VEC = Elem<T>::VEC;
  const int tiles_n = cdiv(N, Cfg::BN), ntiles = cdiv(R, Cfg::BM) * tiles_n;
  const int vec_x = (cin % VEC == 0) && aligned16(x), vec_w = (KK % VEC == 0) && aligned16(w);
  const int vec_o = (N % 4 == 0) && aligned16(out);
  constexpr int lds = conv_gemm_lds<Cfg>();
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<Cfg, FWD>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  conv_gemm_kernel<Cfg, FWD><<<ntiles, kThreads, lds, s>>>((const T*)x, (const T*)w, (const Acc*)bias, (T*)out, (Acc*)partial, R,
                                                         L, cin, KK, N, pad, tiles_n, ntiles, vec_x, vec_w, vec_o);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

// sum of (a, b) over the 256 threads of a finalize workgroup, result in sa[0] / sb[0]: butterfly inside each wave (no
// barrier), the four wave sums meet in LDS in wave order -- one barrier instead of the eight of a shared-memory tree
__device__ __forceinline__ void block256_sum2(double a, double b, double* sa, double* sb) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    a += __shfl_xor(a, m, 64);
    b += __shfl_xor(b, m, 64);
  }
  const int tid = threadIdx.x;
  if ((tid & 63) == 0) {
    sa[1 + (tid >> 6)] = a;
    sb[1 + (tid >> 6)] = b;
  }
  __syncthreads();
  if (tid == 0) {
    sa[0] = ((sa[1] + sa[2]) + sa[3]) + sa[4];
    sb[0] = ((sb[1] + sb[2]) + sb[3]) + sb[4];
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------- BN statistics
// stats[0..3][C] = mean, invstd, scale = gamma*invstd, shift = beta - mean*scale   (P-typed)
// `count_dev` (global-batch statistics, emb_convblock_fwd bn_phase 2): the row count behind the sums, on the device.
template <typename P, typename PP>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const PP* __restrict__ partial, int tiles_m, int C, double count,
                                                          const double* __restrict__ count_dev,
                                                          const P* __restrict__ gamma, const P* __restrict__ beta,
                                                          P* __restrict__ running_mean, P* __restrict__ running_var,
                                                          int training, double momentum, double eps, P* __restrict__ stats,
                                                          long long* __restrict__ num_batches_tracked) {
  __shared__ double sa[256], sb[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  double mean, var;
  // everything the finalising thread needs is requested BEFORE the reduction: as dependent loads after it they added a
  // second memory round trip to a kernel that is nothing but latency
  const double g_c = (double)gamma[c], b_c = (double)beta[c], rm_c = (double)running_mean[c], rv_c = (double)running_var[c];
  if (count_dev != nullptr) count = *count_dev;
  if (training) {
    double a = 0, b = 0;
#pragma unroll 4
    for (int t = tid; t < tiles_m; t += 256) {
      a += (double)partial[((long)t * 2 + 0) * C + c];
      b += (double)partial[((long)t * 2 + 1) * C + c];
    }
    block256_sum2(a, b, sa, sb);
    mean = sa[0] / count;
    var = sb[0] / count - mean * mean;   // biased variance (normalisation); double keeps the cancellation benign
    if (var < 0) var = 0;
  } else {
    mean = rm_c;
    var = rv_c;
  }
  if (tid == 0) {
    const double invstd = 1.0 / sqrt(var + eps);
    const double scale = g_c * invstd;
    stats[c] = (P)mean;
    stats[C + c] = (P)invstd;
    stats[2 * C + c] = (P)scale;
    stats[3 * C + c] = (P)(b_c - mean * scale);
    if (training) {   // nn.BatchNorm1d: running_var uses the unbiased estimate
      const double unbiased = count > 1 ? var * count / (count - 1.0) : var;
      running_mean[c] = (P)((1.0 - momentum) * rm_c + momentum * mean);
      running_var[c] = (P)((1.0 - momentum) * rv_c + momentum * unbiased);
      if (c == 0 && num_batches_tracked != nullptr) *num_batches_tracked += 1;   // nn.BatchNorm1d bookkeeping
    }
  }
}

// ---------------------------------------------------------------------- BN + ReLU + MaxPool (+ Dropout)
// argmax byte: bits 0-3 = offset of the maximum inside the window (first maximum wins, as torch), bit 7 = dropped.
template <typename T, bool NCL_OUT>
__device__ __forceinline__ void bn_relu_pool_item(const T* __restrict__ y, const typename AccOf<T>::type* __restrict__ stats,
                                                  T* __restrict__ out, uint8_t* __restrict__ argmax, int B, int L,
                                                  int Lp, int C, float drop_p, uint64_t seed, uint64_t step_val,
                                                  const uint64_t* __restrict__ step_dev, int64_t grow0, int layer_id, const long i) {
  using Acc = typename AccOf<T>::type;
  constexpr int VEC = Elem<T>::VEC;
  using V = typename Vec16<T>::type;
  const int cv = C / VEC;
  // (32-bit divisions: the launch code refuses B * Lp * C / VEC >= 2^31; three 64-bit divisions per item are ~300 instructions)
  const unsigned iu = (unsigned)i, q1 = iu / (unsigned)cv;
  const int c0 = (int)(iu - q1 * (unsigned)cv) * VEC;
  const int b = (int)(q1 / (unsigned)Lp);
  const int p = (int)(q1 - (unsigned)b * (unsigned)Lp);
  Acc sc[VEC], sh[VEC], best[VEC];
  int arg[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    sc[e] = stats[2 * C + c0 + e];
    sh[e] = stats[3 * C + c0 + e];
    best[e] = (Acc)-1;   // post-ReLU values are >= 0
    arg[e] = 0;
  }
  const T* base = y + ((long)b * L + (long)p * kPoolS) * C + c0;
#pragma unroll
  for (int w = 0; w < kPoolK; ++w) {
    const V v = *reinterpret_cast<const V*>(base + (long)w * C);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      Acc z = (Acc)v[e] * sc[e] + sh[e];
      z = z > (Acc)0 ? z : (Acc)0;
      if (z > best[e]) {
        best[e] = z;
        arg[e] = w;
      }
    }
  }
  float keep_scale = 1.0f;
  uint64_t stream = 0;
  if (drop_p > 0.0f) {
    keep_scale = 1.0f / (1.0f - drop_p);
    stream = rng_stream(step_val + (step_dev ? *step_dev : 0), EMB_RNG_DROPOUT0 + layer_id);
  }
  uint64_t am = 0;
  V o;
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    Acc z = best[e];
    int a = arg[e];
    if (drop_p > 0.0f) {
      const uint64_t idx = ((uint64_t)(grow0 + b) * Lp + p) * C + c0 + e;
      const bool keep = uniform24(philox4x32_10(seed, stream, idx).x) >= drop_p;
      z = keep ? z * (Acc)keep_scale : (Acc)0;
      a |= keep ? 0 : 0x80;
    }
    o[e] = (T)z;
    am |= (uint64_t)(uint8_t)a << (8 * e);
  }
  const long nlc = ((long)b * Lp + p) * C + c0;
  if (VEC == 8) *reinterpret_cast<uint64_t*>(argmax + nlc) = am;
  else if (VEC == 4) *reinterpret_cast<uint32_t*>(argmax + nlc) = (uint32_t)am;
  else *reinterpret_cast<uint16_t*>(argmax + nlc) = (uint16_t)am;
  if (NCL_OUT) {   // last block: the reference flattens [B, C, Lp] (CNN_pre.py:74), channel-major
#pragma unroll
    for (int e = 0; e < VEC; ++e) out[((long)b * C + c0 + e) * Lp + p] = o[e];
  } else {
    *reinterpret_cast<V*>(out + nlc) = o;
  }
}

template <typename T, bool NCL_OUT>
__global__ __launch_bounds__(256) void bn_relu_pool_kernel(const T* __restrict__ y, const typename AccOf<T>::type* __restrict__ stats,
                                                           T* __restrict__ out, uint8_t* __restrict__ argmax, int B, int L,
                                                           int Lp, int C, float drop_p, uint64_t seed, uint64_t step_val,
                                                           const uint64_t* __restrict__ step_dev, int64_t grow0, int layer_id) {
  const long total = (long)B * Lp * (C / Elem<T>::VEC);
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  bn_relu_pool_item<T, NCL_OUT>(y, stats, out, argmax, B, L, Lp, C, drop_p, seed, step_val, step_dev, grow0, layer_id, i);
}

// the same pass finalising the BatchNorm statistics itself from the convolution's partial rows (bn_inline.h): a fixed number of
// workgroups (each pays one pass over the partial rows) walk the pooled elements grid-stride
template <bool NCL_OUT>
__global__ __launch_bounds__(256) void bn_relu_pool_fin_kernel(const __bf16* __restrict__ y, const BnFinFwd fin, __bf16* __restrict__ out,
                                                               uint8_t* __restrict__ argmax, int B, int L, int Lp, int C, float drop_p,
                                                               uint64_t seed, uint64_t step_val, const uint64_t* __restrict__ step_dev,
                                                               int64_t grow0, int layer_id) {
  __shared__ double scratch[256 * 4 + 2 * kBnInlineMaxC];
  __shared__ float fs[4 * kBnInlineMaxC];
  bn_fin_fwd<256>(fin, C, scratch, fs, blockIdx.x == 0);
  const long total = (long)B * Lp * (C / 8);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256)
    bn_relu_pool_item<__bf16, NCL_OUT>(y, fs, out, argmax, B, L, Lp, C, drop_p, seed, step_val, step_dev, grow0, layer_id, i);
}

// BatchNorm / pool / ReLU backward, pass 1.  One block = TT consecutive positions of one sequence, all channels.
// The pooled gradient rows (and their argmax bytes) that can reach those positions are staged in LDS once;
// dz[b,t,c] = sum over the <= 5 windows containing t whose argmax is t (deterministic gather, no atomics),
// masked by ReLU.  dz is written to `dz_out` (the dy buffer) and its per-channel sums (dz, dz*xhat) to
// bpart[blk][2][C].
__host__ __device__ constexpr int pool_plo(int t) { return t >= kPoolK - 1 ? (t - (kPoolK - 1) + 1) / 2 : 0; }

template <typename T> static int bn_bwd_tile(int C) {
  int TT = 64;
  while (TT > 8 && (size_t)(TT / 2 + 5) * C * (sizeof(T) + 1) > 40 * 1024) TT >>= 1;
  return TT;
}

template <typename T, bool NCL_IN>
__device__ __forceinline__ void bn_bwd_dz_body(const T* __restrict__ dout, const uint8_t* __restrict__ argmax,
                                               const T* __restrict__ y, const typename AccOf<T>::type* __restrict__ stats,
                                               T* __restrict__ dz_out, typename AccOf<T>::type* __restrict__ bpart, int L,
                                               int Lp, int C, float keep_scale, int TT, int tiles_per_seq, const int bid,
                                               const int nwg, const int nitems) {
  // workgroup `bid` of `nwg` handles items bid, bid + nwg, ... (nwg == nitems: one item each) and writes ONE partial row
  using Acc = typename AccOf<T>::type;
  constexpr int VEC = Elem<T>::VEC;
  using V = typename Vec16<T>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int TX = C / VEC, TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int c0 = tx * VEC;
  Acc s1[VEC], s2[VEC], mean[VEC], inv[VEC], sc[VEC], sh[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    s1[e] = 0; s2[e] = 0;
    mean[e] = stats[c0 + e]; inv[e] = stats[C + c0 + e]; sc[e] = stats[2 * C + c0 + e]; sh[e] = stats[3 * C + c0 + e];
  }
  for (int item = bid; item < nitems; item += nwg) {
  if (item != bid) __syncthreads();                    // the previous item's LDS tiles are consumed
  const int b = item / tiles_per_seq, t0 = (item % tiles_per_seq) * TT;
  const int t_end = min(L, t0 + TT);
  const int P0 = pool_plo(t0), P1 = min(Lp - 1, (t_end - 1) / 2), NP = P1 - P0 + 1;   // NP <= TT/2 + 5
  const int NPmax = TT / 2 + 5;
  T* dsm = reinterpret_cast<T*>(smem);                                   // [NPmax][C]
  uint8_t* asm_ = reinterpret_cast<uint8_t*>(smem) + (size_t)NPmax * C * sizeof(T);   // [NPmax][C]
  // this thread's first two activation rows are requested NOW: they arrive while the pooled gradient is staged (one memory
  // round trip per item instead of two)
  V yq0, yq1;
  {
    const int ta = t0 + ty, tb = t0 + ty + TY;
    if (ty < TY && ta < t_end) yq0 = *reinterpret_cast<const V*>(y + ((long)b * L + ta) * C + c0);
    if (ty < TY && tb < t_end) yq1 = *reinterpret_cast<const V*>(y + ((long)b * L + tb) * C + c0);
  }
  if (NP > 0) {
    const long nlc0 = ((long)b * Lp + P0) * C;
    const int nvec = NP * C / VEC;
    for (int i = threadIdx.x; i < nvec; i += 256) {
      if (!NCL_IN) *reinterpret_cast<V*>(dsm + (long)i * VEC) = *reinterpret_cast<const V*>(dout + nlc0 + (long)i * VEC);
      if (VEC == 8) *reinterpret_cast<uint64_t*>(asm_ + (long)i * VEC) = *reinterpret_cast<const uint64_t*>(argmax + nlc0 + (long)i * VEC);
      else if (VEC == 4) *reinterpret_cast<uint32_t*>(asm_ + (long)i * VEC) = *reinterpret_cast<const uint32_t*>(argmax + nlc0 + (long)i * VEC);
      else *reinterpret_cast<uint16_t*>(asm_ + (long)i * VEC) = *reinterpret_cast<const uint16_t*>(argmax + nlc0 + (long)i * VEC);
    }
    if (NCL_IN) {   // dout[b][c][p], p contiguous: lanes walk p (coalesced loads), a thread gathers VEC channels of its p and
      // stores them as ONE vector (element stores at a pitch of C elements put a whole store group on two banks)
      for (int i = threadIdx.x; i < NP * TX; i += 256) {
        const int cv = i / NP, pp = i % NP;
        V v;
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = dout[((long)b * C + cv * VEC + e) * Lp + P0 + pp];
        *reinterpret_cast<V*>(dsm + (long)pp * C + cv * VEC) = v;
      }
    }
  }
  __syncthreads();
  if (ty < TY) {
    int jj = 0;
    for (int t = t0 + ty; t < t_end; t += TY, ++jj) {
      Acc dz[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) dz[e] = 0;
      const int p_lo = pool_plo(t), p_hi = min(Lp - 1, t / 2);
      for (int p = p_lo; p <= p_hi; ++p) {
        const int off = t - kPoolS * p;
        const long li = (long)(p - P0) * C + c0;
        const V g = *reinterpret_cast<const V*>(dsm + li);
        uint64_t am;
        if (VEC == 8) am = *reinterpret_cast<const uint64_t*>(asm_ + li);
        else if (VEC == 4) am = *reinterpret_cast<const uint32_t*>(asm_ + li);
        else am = *reinterpret_cast<const uint16_t*>(asm_ + li);
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if ((int)((am >> (8 * e)) & 0xFF) == off) dz[e] += (Acc)g[e] * (Acc)keep_scale;   // dropped: bit 7 set, never equal
      }
      const long r = (long)b * L + t;
      V yv;
      if (jj == 0) yv = yq0;
      else if (jj == 1) yv = yq1;
      else yv = *reinterpret_cast<const V*>(y + r * C + c0);
      V o;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const Acc yy = (Acc)yv[e];
        const Acc g = (yy * sc[e] + sh[e]) > (Acc)0 ? dz[e] : (Acc)0;   // ReLU mask
        o[e] = (T)g;
        s1[e] += g;
        s2[e] += g * ((yy - mean[e]) * inv[e]);
      }
      *reinterpret_cast<V*>(dz_out + r * C + c0) = o;
    }
  }
  }   // items
  __syncthreads();   // LDS is reused for the block reduction
  // [TY][2C + 1]: the element-wise stores of a 32-lane group go to (ty, tx) = 8 x 4 rows/columns; with a pitch of 2C words
  // (a multiple of the 32 store banks) all eight ty collided, the odd pitch spreads them over the banks (PMC: conflict
  // cycles were 3/4 of this kernel's LDS cycles)
  Acc* red = reinterpret_cast<Acc*>(smem);
  const int RP = 2 * C + 1;
  if (ty < TY) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      red[(long)ty * RP + c0 + e] = s1[e];
      red[(long)ty * RP + C + c0 + e] = s2[e];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    Acc a = 0;
    for (int q = 0; q < TY; ++q) a += red[(long)q * RP + i];
    bpart[(long)bid * 2 * C + i] = a;
  }
}

template <typename T, bool NCL_IN>
__global__ __launch_bounds__(256) void bn_bwd_dz_kernel(const T* __restrict__ dout, const uint8_t* __restrict__ argmax,
                                                        const T* __restrict__ y, const typename AccOf<T>::type* __restrict__ stats,
                                                        T* __restrict__ dz_out, typename AccOf<T>::type* __restrict__ bpart, int L,
                                                        int Lp, int C, float keep_scale, int TT, int tiles_per_seq, int nwg, int nitems) {
  bn_bwd_dz_body<T, NCL_IN>(dout, argmax, y, stats, dz_out, bpart, L, Lp, C, keep_scale, TT, tiles_per_seq, (int)blockIdx.x, nwg, nitems);
}

// the gather pass carrying the backward of the epigenomic MLP stack as its first `nr` workgroups (rider.h)
template <bool NCL_IN>
__global__ __launch_bounds__(256) void bn_bwd_dz_rider_kernel(const __bf16* __restrict__ dout, const uint8_t* __restrict__ argmax,
                                                              const __bf16* __restrict__ y, const float* __restrict__ stats,
                                                              __bf16* __restrict__ dz_out, float* __restrict__ bpart, int L, int Lp, int C,
                                                              float keep_scale, int TT, int tiles_per_seq, int nwg, int nitems,
                                                              const MlpBwdArgs<__bf16> ba, const MmBwdLayout bl, const int nr) {
  if ((int)blockIdx.x < nr) {
    extern __shared__ __attribute__((aligned(16))) char rider_arena[];
    if (threadIdx.x < 64) mlp_bwd_mfma_body(ba, bl, (int)blockIdx.x, rider_arena);
    return;
  }
  bn_bwd_dz_body<__bf16, NCL_IN>(dout, argmax, y, stats, dz_out, bpart, L, Lp, C, keep_scale, TT, tiles_per_seq, (int)blockIdx.x - nr, nwg, nitems);
}

// finalise dgamma / dbeta and the two per-channel means the apply pass needs: coef[0][c] = mean(dz), coef[1][c] = mean(dz*xhat)
// Global-batch statistics (bn_phase): phase 1 passes `sums` (raw double sums out, coef untouched), phase 2 passes the
// all-reduced sums as `bpart` (PP = double, nblk = 1), `count_dev` and no dgamma / dbeta (those stay LOCAL sums: the
// data-parallel gradient reduction adds them over ranks like every other parameter gradient).
template <typename P, typename PP>
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const PP* __restrict__ bpart, int nblk, int C, double count,
                                                              const double* __restrict__ count_dev, P* __restrict__ dgamma,
                                                              P* __restrict__ dbeta, P* __restrict__ coef, double* __restrict__ sums) {
  __shared__ double sa[256], sb[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  if (count_dev != nullptr) count = *count_dev;
  double a = 0, b = 0;
#pragma unroll 4
  for (int t = tid; t < nblk; t += 256) {
    a += (double)bpart[((long)t * 2 + 0) * C + c];
    b += (double)bpart[((long)t * 2 + 1) * C + c];
  }
  block256_sum2(a, b, sa, sb);
  if (tid == 0) {
    if (dbeta != nullptr) {
      dbeta[c] = (P)sa[0];
      dgamma[c] = (P)sb[0];
    }
    if (sums != nullptr) {
      sums[c] = sa[0];
      sums[C + c] = sb[0];
      if (c == 0) sums[2 * C] = count;
    } else {
      coef[c] = (P)(sa[0] / count);
      coef[C + c] = (P)(sb[0] / count);
    }
  }
}

// phase 1 of the forward with global-batch statistics: per-tile partial sums -> sums[0..C) = sum x, [C..2C) = sum x^2,
// sums[2C] = rows behind them (all double; the caller all-reduces the 2C+1 values over the ranks)
template <typename P>
__global__ __launch_bounds__(256) void bn_sums_kernel(const P* __restrict__ partial, int tiles_m, int C, double count,
                                                      double* __restrict__ sums) {
  __shared__ double sa[256], sb[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  double a = 0, b = 0;
#pragma unroll 4
  for (int t = tid; t < tiles_m; t += 256) {
    a += (double)partial[((long)t * 2 + 0) * C + c];
    b += (double)partial[((long)t * 2 + 1) * C + c];
  }
  block256_sum2(a, b, sa, sb);
  if (tid == 0) {
    sums[c] = sa[0];
    sums[C + c] = sb[0];
    if (c == 0) sums[2 * C] = count;
  }
}

// pass 2 (in place on the dz buffer): dy = scale * (dz - mean(dz) - xhat * mean(dz*xhat))  (training) or scale * dz (eval),
// i.e. per channel dy = A*dz + Bc*y + D.  Thread (tx, ty): fixed channel vector, rows ty, ty+TY, ... of its block.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_affine_kernel(const T* __restrict__ y, const typename AccOf<T>::type* __restrict__ stats,
                                                            const typename AccOf<T>::type* __restrict__ coef, T* __restrict__ dy,
                                                            int R, int C, int training, int rows_per_block) {
  using Acc = typename AccOf<T>::type;
  constexpr int VEC = Elem<T>::VEC;
  using V = typename Vec16<T>::type;
  const int TX = C / VEC, TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  if (ty >= TY) return;
  const int c0 = tx * VEC;
  Acc A[VEC], Bc[VEC], D[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const Acc mean = stats[c0 + e], inv = stats[C + c0 + e], sc = stats[2 * C + c0 + e];
    A[e] = sc;
    Bc[e] = training ? -sc * coef[C + c0 + e] * inv : (Acc)0;
    D[e] = training ? sc * (coef[C + c0 + e] * inv * mean - coef[c0 + e]) : (Acc)0;
  }
  const int r_begin = blockIdx.x * rows_per_block, r_end = min(R, r_begin + rows_per_block);
  for (int r = r_begin + ty; r < r_end; r += TY) {
    const long off = (long)r * C + c0;
    const V dzv = *reinterpret_cast<const V*>(dy + off);
    const V yv = *reinterpret_cast<const V*>(y + off);
    V o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = (T)(A[e] * (Acc)dzv[e] + Bc[e] * (Acc)yv[e] + D[e]);
    *reinterpret_cast<V*>(dy + off) = o;
  }
}

// the same pass finalising dgamma / dbeta and the two means itself from the gather pass's partial rows (bn_inline.h); a fixed
// number of workgroups walk the row blocks grid-stride
__global__ __launch_bounds__(256) void bn_bwd_affine_fin_kernel(const __bf16* __restrict__ y, const float* __restrict__ stats, const BnFinBwd fin,
                                                                __bf16* __restrict__ dy, int R, int C, int training, int rows_per_block) {
  __shared__ double scratch[256 * 4 + 2 * kBnInlineMaxC];
  __shared__ float fc[2 * kBnInlineMaxC];
  const int TX = C / 8, TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int c0 = tx * 8;
  // the rows of the workgroup's first chunk do not depend on the finalised sums: they are requested before the prologue, so
  // that their HBM round trip runs under the prologue's L2 round trip, sums and barriers
  constexpr int NP = 8;                              // rows_per_block == 8 * TY (launch code): the whole first chunk
  bf16x8 pz[NP], py[NP];
  const int rb = blockIdx.x * rows_per_block, rb_end = min(R, rb + rows_per_block);
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int r = rb + ty + j * TY;
    if (ty < TY && r < rb_end) {
      const long off = (long)r * C + c0;
      pz[j] = *reinterpret_cast<const bf16x8*>(dy + off);
      py[j] = *reinterpret_cast<const bf16x8*>(y + off);
    }
  }
  bn_fin_bwd<256>(fin, C, scratch, fc, blockIdx.x == 0);
  if (ty >= TY) return;
  float A[8], Bc[8], D[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float mean = stats[c0 + e], inv = stats[C + c0 + e], sc = stats[2 * C + c0 + e];
    A[e] = sc;
    Bc[e] = training ? -sc * fc[C + c0 + e] * inv : 0.0f;
    D[e] = training ? sc * (fc[C + c0 + e] * inv * mean - fc[c0 + e]) : 0.0f;
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int r = rb + ty + j * TY;
    if (r < rb_end) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (__bf16)(A[e] * (float)pz[j][e] + Bc[e] * (float)py[j][e] + D[e]);
      *reinterpret_cast<bf16x8*>(dy + (long)r * C + c0) = o;
    }
  }
  for (int r0 = rb; r0 < R; r0 += gridDim.x * rows_per_block) {
    const int r_end = min(R, r0 + rows_per_block);
    for (int r = r0 + ty + (r0 == rb ? NP * TY : 0); r < r_end; r += TY) {
      const long off = (long)r * C + c0;
      const bf16x8 dzv = *reinterpret_cast<const bf16x8*>(dy + off);
      const bf16x8 yv = *reinterpret_cast<const bf16x8*>(y + off);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (__bf16)(A[e] * (float)dzv[e] + Bc[e] * (float)yv[e] + D[e]);
      *reinterpret_cast<bf16x8*>(dy + off) = o;
    }
  }
}

// ------------------------------------------------------------------------------------- wgrad
// slab[s][Cout][KK+1] = dy^T . [view(x) | 1] over rows [s*kper, (s+1)*kper)
template <class Cfg>
__global__ __launch_bounds__(kThreads, 2) void conv_wgrad_kernel(const typename Cfg::T* __restrict__ dy,
                                                              const typename Cfg::T* __restrict__ x,
                                                              typename Cfg::M::Acc* __restrict__ slab, int R, int L, int cin,
                                                              int KK, int Cout, int pad, int kper, int tiles_n, int tiles_per_slice,
                                                              int vec_dy, int vec_x) {
  using T = typename Cfg::T;
  using Acc = typename Cfg::M::Acc;
  extern __shared__ __attribute__((aligned(16))) char arena[];
  const int s = blockIdx.x / tiles_per_slice;
  const int tile = xcd_remap(blockIdx.x % tiles_per_slice, tiles_per_slice);
  const int k_begin = s * kper, k_end = min(R, k_begin + kper);
  GemmOperand<T> A{dy, nullptr, Cout, vec_dy != 0};
  const int row0 = (tile / tiles_n) * Cfg::BM, col0 = (tile % tiles_n) * Cfg::BN;
  typename Cfg::M::AccV acc[Cfg::MI][Cfg::NI];
  zero_acc<Cfg>(acc);
  Stager<T, true, Cfg::BM, Cfg::BK, XfNone> sa{A.ptr, nullptr, Cout, row0, Cout, k_end, A.vec_ok, XfNone{}, -1, 0, 0, 0};
  Stager<T, true, Cfg::BN, Cfg::BK, XfNone> sb{x, nullptr, cin, col0, KK, k_end, vec_x != 0, XfNone{}, KK, L, cin, pad};
  gemm_mainloop<Cfg>(sa, sb, k_end, arena, acc, k_begin);
  Acc* cs = reinterpret_cast<Acc*>(arena);
  reduce_to_slab<Cfg>(acc, cs);
  Acc* dst = slab + (long)s * Cout * (KK + 1);
  for (int i = threadIdx.x; i < Cfg::BM * Cfg::BN; i += kThreads) {
    const int r = i / Cfg::BN, c = i % Cfg::BN;
    if (row0 + r < Cout && col0 + c <= KK) dst[(long)(row0 + r) * (KK + 1) + col0 + c] = cs[r * Cfg::CS + c];
  }
}

// ------------------------------------------------------------------------------ layout helpers
// W[Cout][Cin][k] (P) -> wpack[Cout][k*cin_pad] (T, tap-major, zero padded channels)
//                     -> wflip[cin_pad][k*Cout]  (T, wflip[ci][j*Cout + o] = W[o][ci][k-1-j])
template <typename P, typename T>
__global__ void conv_pack_weight_kernel(const P* __restrict__ W, T* __restrict__ wpack, T* __restrict__ wflip, int Cout, int Cin,
                                        int cin_pad, int k) {
  const long n1 = (long)Cout * k * cin_pad, n2 = (long)cin_pad * k * Cout;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n1) {
    const int ci = (int)(i % cin_pad), j = (int)((i / cin_pad) % k), o = (int)(i / ((long)cin_pad * k));
    wpack[i] = ci < Cin ? (T)(typename AccOf<T>::type)W[((long)o * Cin + ci) * k + j] : (T)0.0f;
  } else if (i < n1 + n2 && wflip != nullptr) {
    const long q = i - n1;
    const int o = (int)(q % Cout), j = (int)((q / Cout) % k), ci = (int)(q / ((long)Cout * k));
    wflip[q] = ci < Cin ? (T)(typename AccOf<T>::type)W[((long)o * Cin + ci) * k + (k - 1 - j)] : (T)0.0f;
  }
}

// x[B][C][L] (any float type) -> out[B][L][Cpad] (T), zero padded channels
template <typename S, typename T>
__global__ void ncl_to_nlc_kernel(const S* __restrict__ x, T* __restrict__ out, int B, int C, int L, int Cpad) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over B*L
  if (i >= (long)B * L) return;
  const int t = (int)(i % L);
  const long b = i / L;
  for (int c = 0; c < Cpad; ++c)
    out[i * Cpad + c] = c < C ? (T)(typename AccOf<T>::type)(typename AccOf<S>::type)x[(b * C + c) * L + t] : (T)0.0f;
}

// ------------------------------------------------------------------------------- host plumbing
struct ConvWs {
  size_t stat_partial, gram, bwd_partial, coef, slab, total;
  int tiles_m, nblk_bwd, rows_per_block, S, kper;
};

template <typename T> static ConvWs conv_workspace(int B, int L, int cin_pad, int Cout, int k) {
  using P = typename AccOf<T>::type;
  using CF = typename ConvCfg<T>::F64;
  using CW = typename ConvCfg<T>::W;
  ConvWs w;
  const long R = (long)B * L;
  const int KK = k * cin_pad;
  w.tiles_m = cdiv((int)R, CF::BM);
  w.rows_per_block = bn_bwd_tile<T>(Cout);                 // TT positions of one sequence per block
  w.nblk_bwd = B * cdiv(L, w.rows_per_block);
  const int tiles = cdiv(Cout, CW::BM) * cdiv(KK + 1, CW::BN);
  int S = 1024 / (tiles > 0 ? tiles : 1);
  const int max_s = (int)(R / (4 * CW::BK));
  if (S > max_s) S = max_s;
  if (S < 1) S = 1;
  int kper = cdiv((int)R, S);
  kper = cdiv(kper, CW::BK) * CW::BK;
  S = cdiv((int)R, kper);
  w.S = S;
  w.kper = kper;
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const int pad = (k - 1) / 2;
  const int tiles_direct = conv_tiling(B, L, pad).tiles_m;          // direct kernels tile per sequence
  const int S_direct = conv_wgrad_slices(B, L, cin_pad, pad, KK, Cout, dtype_code<T>());
  const int tiles_stat = w.tiles_m > tiles_direct ? w.tiles_m : tiles_direct;
  w.stat_partial = al((size_t)tiles_stat * 2 * Cout * sizeof(P));
  if (S_direct > S) S = S_direct;
  w.bwd_partial = al((size_t)w.nblk_bwd * 2 * Cout * sizeof(P));
  w.coef = al((size_t)2 * Cout * sizeof(P));
  w.slab = al((size_t)S * Cout * (KK + 1) * sizeof(P));
  if (sizeof(T) == 2 && conv_first_supported(dtype_code<T>(), B, L, cin_pad, Cout, k)) {   // accumulate pass of the fused first block: 64-float rows
    const size_t acc = al((size_t)conv_first_blocks(B, L, cin_pad, Cout, k) * Cout * 64 * sizeof(float));
    if (acc > w.slab) w.slab = acc;
  }
  // fused first block in bf16: per-workgroup partial lag statistics of the input (first_gram.h), after the statistics partials
  w.gram = (sizeof(T) == 2 && conv_first_supported(dtype_code<T>(), B, L, cin_pad, Cout, k))
               ? al(conv_first_gram_part_bytes(B, L, cin_pad, Cout, k)) : 0;
  const size_t fwd = w.stat_partial + w.gram, bwd = w.bwd_partial + w.coef + w.slab;
  w.total = fwd > bwd ? fwd : bwd;
  return w;
}

// the recompute-free backward of the fused first block (first_gram.h) needs the forward to leave the lag statistics behind `stats`
static int g_first_linear = 1;    // emb_convblock_first_linear() switches it (the tests compare the two backward forms)
static bool first_linear_enabled() { return g_first_linear != 0; }
template <typename T> static bool first_linear(int training, int bn_phase) {
  return sizeof(T) == 2 && first_linear_enabled() && training && bn_phase == 0;
}

template <typename T>
static int convblock_fwd(const void* x, const void* wpack, const void* bias, const void* gamma, const void* beta, void* rmean,
                         void* rvar, int training, double momentum, double eps, float drop_p, uint64_t seed, uint64_t step_val,
                         const uint64_t* step_dev, int64_t row0, int layer_id, void* y, void* stats, void* out, uint8_t* argmax,
                         int out_ncl, void* ws, int64_t ws_bytes, void* nbt, int x_codes, int bn_phase, double* bn_sums, int B, int L,
                         int cin_pad, int Cout, int k, hipStream_t s) {
  using P = typename AccOf<T>::type;
  constexpr int VEC = Elem<T>::VEC;
  const ConvWs w = conv_workspace<T>(B, L, cin_pad, Cout, k);
  EMB_CHECK_ARG((long)B * L * Cout < (1l << 31), "emb_convblock_fwd: B * L * Cout must stay below 2^31 (32-bit element indices)");
  EMB_CHECK_ARG((size_t)ws_bytes >= w.total, "emb_convblock_fwd: workspace too small (%lld < %zu)", (long long)ws_bytes, w.total);
  EMB_CHECK_ARG(cin_pad % VEC == 0 && Cout % VEC == 0, "emb_convblock_fwd: channels must be multiples of %d", VEC);
  const int R = B * L, KK = k * cin_pad, pad = (k - 1) / 2, Lp = (L - kPoolK) / kPoolS + 1;
  EMB_CHECK_ARG(Lp >= 1, "emb_convblock_fwd: sequence too short for the pooling window");
  EMB_CHECK_ARG(x_codes == 2 || !x_codes || y == nullptr, "emb_convblock_fwd: base-code input (x_codes = 1) needs the fused first block (y == NULL)");
  EMB_CHECK_ARG(x_codes != 2 || (y != nullptr && training && bn_phase != 2 && cin_pad == 8 && sizeof(T) == 2),
                "emb_convblock_fwd: x_codes = 2 (loader layout in, channels-last image out through y) is a training-mode bf16 call");
  const void* x_img = x;   // what the passes after the statistics pass read
  if (y == nullptr || x_codes == 2) {   // first block, nothing stored: statistics pass + fused apply pass, both recompute the convolution
    EMB_CHECK_ARG(conv_first_supported(dtype_code<T>(), B, L, cin_pad, Cout, k),
                  "emb_convblock_fwd: y may only be NULL when emb_convblock_needs_y() returns 0");
    int rows = 0;
    if (training && bn_phase != 2) {
      { const int rcj = gram_jobs_flush(s); if (rcj != EMB_OK) return rcj; }   // a set parked by the previous forward reads this workspace
      const int rc0 = conv_first_stats(x, x_codes, x_codes == 2 ? y : nullptr, wpack, bias, ws, &rows,
                                       first_linear<T>(training, bn_phase) ? (float*)((char*)ws + w.stat_partial) : nullptr, B, L, Cout, k, s);
      if (rc0 != EMB_OK) return rc0 == 1 ? EMB_ERR_ARG : rc0;
    }
    if (x_codes == 2) {
      x_img = y;
      x_codes = 0;
    }
    if (bn_phase == 1) {
      bn_sums_kernel<P><<<Cout, 256, 0, s>>>((const P*)ws, rows, Cout, (double)R, bn_sums);
      EMB_CHECK_LAUNCH();
      return EMB_OK;
    }
    // local statistics in training: the apply pass finalises them itself from the statistics pass's partial rows (bn_inline.h)
    constexpr bool inline_fin = true;
    BnFinFwd fin{};
    bool inl = false;
    if constexpr (sizeof(P) == 4) {
      if (inline_fin && training && bn_phase == 0 && rows > 0 && rows <= 512) {
        fin.partial = (const float*)ws; fin.rows = rows; fin.gamma = (const float*)gamma; fin.beta = (const float*)beta;
        fin.running_mean = (float*)rmean; fin.running_var = (float*)rvar; fin.stats = (float*)stats; fin.num_batches_tracked = (long long*)nbt;
        fin.momentum = momentum; fin.eps = eps; fin.count = (double)R;
        inl = true;
      }
    }
    if (bn_phase == 2)
      bn_finalize_kernel<P, double><<<Cout, 256, 0, s>>>(bn_sums, 1, Cout, 0.0, bn_sums + 2 * Cout, (const P*)gamma, (const P*)beta,
                                                        (P*)rmean, (P*)rvar, training, momentum, eps, (P*)stats, (long long*)nbt);
    else if (!inl)
      bn_finalize_kernel<P, P><<<Cout, 256, 0, s>>>((const P*)ws, rows, Cout, (double)R, nullptr, (const P*)gamma, (const P*)beta,
                                                   (P*)rmean, (P*)rvar, training, momentum, eps, (P*)stats, (long long*)nbt);
    EMB_CHECK_LAUNCH();
    const bool lin = first_linear<T>(training, bn_phase);   // lag statistics totals: behind the four BatchNorm vectors (emb_convblock_stats_elems)
    const int rc1 = conv_first_apply(x_img, x_codes, wpack, bias, stats, inl ? &fin : nullptr, out, argmax, out_ncl,
                                     lin ? (const float*)((char*)ws + w.stat_partial) : nullptr, lin ? (float*)stats + 4 * Cout : nullptr, drop_p, seed, step_val, step_dev,
                                     row0, layer_id, B, L, Cout, k, s);
    return rc1 == 1 ? EMB_ERR_ARG : rc1;
  }
  if (bn_phase != 2) {   // the convolution (stored) and its per-tile channel sums
    int tiles_m = conv_tiling(B, L, pad).tiles_m;
    int rc = 1;
    if constexpr (sizeof(T) == 4)   // fp32: ring GEMM on the shifted activation rows (gemm_jobs.h); its row tiles are never more than the direct kernel's
      rc = gemm_jobs_conv(true, x, wpack, bias, y, ws, &tiles_m, B, L, cin_pad, KK, Cout, pad, s);
    if (rc == 1) {
      tiles_m = conv_tiling(B, L, pad).tiles_m;
      rc = launch_conv_direct(dtype_code<T>(), true, x, wpack, bias, y, ws, &tiles_m, B, L, cin_pad, KK, Cout, pad, s);
    }
    if (rc == 1) {   // activation tile does not fit in LDS: generic GEMM on the im2col view
      if (Cout >= 64) rc = launch_conv_gemm<typename ConvCfg<T>::F64, true>(x, wpack, bias, y, ws, R, L, cin_pad, KK, Cout, pad, s);
      else rc = launch_conv_gemm<typename ConvCfg<T>::F32, true>(x, wpack, bias, y, ws, R, L, cin_pad, KK, Cout, pad, s);
      tiles_m = cdiv(R, Cout >= 64 ? ConvCfg<T>::F64::BM : ConvCfg<T>::F32::BM);
    }
    if (rc != EMB_OK) return rc;
    if (bn_phase == 1) {
      bn_sums_kernel<P><<<Cout, 256, 0, s>>>((const P*)ws, tiles_m, Cout, (double)R, bn_sums);
      EMB_CHECK_LAUNCH();
      return EMB_OK;
    }
    if constexpr (sizeof(T) == 2) {   // bf16, local statistics: the pooling pass finalises them itself (bn_inline.h)
      constexpr bool inline_fin = true;
      if (inline_fin && training && tiles_m > 0 && tiles_m <= 512 && Cout <= kBnInlineMaxC && Cout % 8 == 0) {
        BnFinFwd fin{};
        fin.partial = (const float*)ws; fin.rows = tiles_m; fin.gamma = (const float*)gamma; fin.beta = (const float*)beta;
        fin.running_mean = (float*)rmean; fin.running_var = (float*)rvar; fin.stats = (float*)stats; fin.num_batches_tracked = (long long*)nbt;
        fin.momentum = momentum; fin.eps = eps; fin.count = (double)R;
        const long total = (long)B * Lp * (Cout / 8);
        const int want = (int)((total + 255) / 256), grid = want < 768 ? want : 768;   // three workgroups per CU (swept 512 .. 2048)
        if (out_ncl)
          bn_relu_pool_fin_kernel<true><<<grid, 256, 0, s>>>((const __bf16*)y, fin, (__bf16*)out, argmax, B, L, Lp, Cout, drop_p, seed, step_val, step_dev, row0, layer_id);
        else
          bn_relu_pool_fin_kernel<false><<<grid, 256, 0, s>>>((const __bf16*)y, fin, (__bf16*)out, argmax, B, L, Lp, Cout, drop_p, seed, step_val, step_dev, row0, layer_id);
        EMB_CHECK_LAUNCH();
        return EMB_OK;
      }
    }
    bn_finalize_kernel<P, P><<<Cout, 256, 0, s>>>((const P*)ws, tiles_m, Cout, (double)R, nullptr, (const P*)gamma, (const P*)beta,
                                                 (P*)rmean, (P*)rvar, training, momentum, eps, (P*)stats, (long long*)nbt);
  } else {
    bn_finalize_kernel<P, double><<<Cout, 256, 0, s>>>(bn_sums, 1, Cout, 0.0, bn_sums + 2 * Cout, (const P*)gamma, (const P*)beta,
                                                      (P*)rmean, (P*)rvar, training, momentum, eps, (P*)stats, (long long*)nbt);
  }
  EMB_CHECK_LAUNCH();
  const long total = (long)B * Lp * (Cout / VEC);
  const int grid = (int)((total + 255) / 256);
  if (out_ncl)
    bn_relu_pool_kernel<T, true><<<grid, 256, 0, s>>>((const T*)y, (const P*)stats, (T*)out, argmax, B, L, Lp, Cout, drop_p, seed, step_val, step_dev, row0, layer_id);
  else
    bn_relu_pool_kernel<T, false><<<grid, 256, 0, s>>>((const T*)y, (const P*)stats, (T*)out, argmax, B, L, Lp, Cout, drop_p, seed, step_val, step_dev, row0, layer_id);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

template <typename T>
static int convblock_bwd(const void* dout, int dout_ncl, const uint8_t* argmax, const void* y, const void* stats, const void* x,
                         const void* wflip, const void* wpack, const void* bias, float drop_p, int training, void* dx, void* dW,
                         void* dbias, void* dgamma, void* dbeta, void* dy, void* ws, int64_t ws_bytes, int x_codes, int bn_phase,
                         double* bn_sums, int B, int L, int Cin, int cin_pad, int Cout, int k, hipStream_t s) {
  using P = typename AccOf<T>::type;
  using CW = typename ConvCfg<T>::W;
  constexpr int VEC = Elem<T>::VEC;
  const ConvWs w = conv_workspace<T>(B, L, cin_pad, Cout, k);
  EMB_CHECK_ARG((size_t)ws_bytes >= w.total, "emb_convblock_bwd: workspace too small (%lld < %zu)", (long long)ws_bytes, w.total);
  const int R = B * L, KK = k * cin_pad, pad = (k - 1) / 2, Lp = (L - kPoolK) / kPoolS + 1;
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  char* base = (char*)ws;
  P* bpart = (P*)base;
  P* coef = (P*)(base + w.bwd_partial);
  P* slab = (P*)(base + w.bwd_partial + w.coef);
  EMB_CHECK_ARG(!x_codes || y == nullptr, "emb_convblock_bwd: base-code input (x_codes) needs the fused first block (y == NULL)");
  if (y == nullptr) {   // first block without stored activations (conv_first.hip)
    EMB_CHECK_ARG(conv_first_supported(dtype_code<T>(), B, L, cin_pad, Cout, k) && dx == nullptr && wpack && bias,
                  "emb_convblock_bwd: y == NULL needs the fused first block (emb_convblock_needs_y() == 0), wpack, bias and no dx");
    const int nb = conv_first_blocks(B, L, cin_pad, Cout, k);
    EMB_CHECK_ARG((size_t)nb * 2 * Cout * sizeof(P) <= w.bwd_partial && (size_t)nb * Cout * (KK + 1) * sizeof(P) <= w.slab &&
                      (size_t)nb * Cout * 64 * sizeof(float) <= w.slab,
                  "emb_convblock_bwd: workspace layout too small for the fused first block");
    int rows = 0, S = 0, rc = EMB_OK;
    rc = gram_jobs_flush(s);   // parked totals jobs nobody carried: they read the workspace the backward is about to reuse
    if (rc != EMB_OK) return rc;
    if (first_linear<T>(training, bn_phase)) {   // one pass A = g^T xview + a per-channel finish: no convolution, no sums pass (first_gram.h)
      rc = conv_first_bwd_acc(dout, dout_ncl, argmax, x, x_codes, keep_scale, slab, &S, B, L, Cout, k, s);
      if (rc != EMB_OK) return rc == 1 ? EMB_ERR_ARG : rc;
      return conv_first_bwd_finish(slab, S, (const float*)stats + 4 * Cout, wpack, bias, stats, dW, dbias, dgamma, dbeta, training, B, L, Cin, Cout, k, s);
    }
    constexpr bool inline_fin = true;
    BnFinBwd fin{};
    bool inl = false;
    if (bn_phase != 2) {
      rc = conv_first_bwd_sums(dout, dout_ncl, argmax, x, x_codes, wpack, bias, stats, keep_scale, bpart, &rows, B, L, Cout, k, s);
      if (rc != EMB_OK) return rc == 1 ? EMB_ERR_ARG : rc;
      if constexpr (sizeof(P) == 4) {   // local statistics: the weight-gradient pass finalises the two means itself (bn_inline.h)
        if (inline_fin && bn_phase == 0 && rows > 0 && rows <= 512) {
          fin.partial = (const float*)bpart; fin.rows = rows; fin.dgamma = (float*)dgamma; fin.dbeta = (float*)dbeta; fin.coef = (float*)coef;
          fin.count = (double)R;
          inl = true;
        }
      }
      if (!inl) {
        bn_bwd_finalize_kernel<P, P><<<Cout, 256, 0, s>>>(bpart, rows, Cout, (double)R, nullptr, (P*)dgamma, (P*)dbeta, coef,
                                                         bn_phase == 1 ? bn_sums : nullptr);
        EMB_CHECK_LAUNCH();
      }
      if (bn_phase == 1) return EMB_OK;
    } else {
      bn_bwd_finalize_kernel<P, double><<<Cout, 256, 0, s>>>(bn_sums, 1, Cout, 0.0, bn_sums + 2 * Cout, (P*)nullptr, (P*)nullptr, coef,
                                                            nullptr);
      EMB_CHECK_LAUNCH();
    }
    rc = conv_first_bwd_wgrad(dout, dout_ncl, argmax, x, x_codes, wpack, bias, stats, coef, inl ? &fin : nullptr, keep_scale, training, slab, &S, B,
                              L, Cout, k, s);
    if (rc != EMB_OK) return rc == 1 ? EMB_ERR_ARG : rc;
    ReduceJob j{};   // slabs -> dW (torch layout, real channels) / dbias, slices summed in fixed order (reduce.hip)
    j.in = slab; j.out[0] = dW; j.out[1] = dbias; j.per = (long)Cout * (KK + 1); j.S = S; j.kind = RJ_CONV;
    j.iv[0] = Cin; j.iv[1] = cin_pad; j.iv[2] = k;
    return reduce_submit(j, sizeof(P) == 8, s);
  }
  {   // dgamma / dbeta are defined in eval mode too (x-hat then uses the running statistics)
    const int TT = w.rows_per_block, tiles_per_seq = cdiv(L, TT);
    const int TY = 256 / (Cout / VEC);
    bool affine_done = false;
    if (bn_phase != 2) {   // dz into the dy buffer + the per-block channel sums
      size_t sm = (size_t)(TT / 2 + 5) * Cout * (sizeof(T) + 1);
      const size_t sm_red = (size_t)TY * (2 * Cout + 1) * sizeof(P);
      if (sm_red > sm) sm = sm_red;
      sm = (sm + 15) & ~(size_t)15;
      // bf16, local statistics: the gather pass runs as at most 512 workgroups (one partial row each) and the elementwise
      // pass below finalises dgamma / dbeta / the two means itself from those rows (bn_inline.h): no finalize launch
      constexpr bool inline_fin = true;
      const bool inl = sizeof(T) == 2 && inline_fin && bn_phase == 0 && Cout <= kBnInlineMaxC && Cout % 8 == 0;
      const int nitems = w.nblk_bwd, nwg = inl && nitems > 512 ? 512 : nitems;
      Rider rd;
      bool carried = false;
      if constexpr (sizeof(T) == 2) {   // a parked MLP backward of this stream rides along (rider.h)
        if (rider_take(s, RIDER_MLP_BWD, &rd)) {
          const size_t sm2 = sm > rd.lds ? sm : rd.lds;
          static bool attr = false;
          if (!attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_bwd_dz_rider_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_bwd_dz_rider_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            attr = true;
          }
          if (dout_ncl)
            bn_bwd_dz_rider_kernel<true><<<nwg + rd.nwg, 256, sm2, s>>>((const T*)dout, argmax, (const T*)y, (const P*)stats, (T*)dy, bpart, L, Lp, Cout, keep_scale, TT, tiles_per_seq, nwg, nitems, rd.ba, rd.bl, rd.nwg);
          else
            bn_bwd_dz_rider_kernel<false><<<nwg + rd.nwg, 256, sm2, s>>>((const T*)dout, argmax, (const T*)y, (const P*)stats, (T*)dy, bpart, L, Lp, Cout, keep_scale, TT, tiles_per_seq, nwg, nitems, rd.ba, rd.bl, rd.nwg);
          carried = true;
        }
      }
      if (carried) {
      } else if (dout_ncl)
        bn_bwd_dz_kernel<T, true><<<nwg, 256, sm, s>>>((const T*)dout, argmax, (const T*)y, (const P*)stats, (T*)dy, bpart, L, Lp, Cout, keep_scale, TT, tiles_per_seq, nwg, nitems);
      else
        bn_bwd_dz_kernel<T, false><<<nwg, 256, sm, s>>>((const T*)dout, argmax, (const T*)y, (const P*)stats, (T*)dy, bpart, L, Lp, Cout, keep_scale, TT, tiles_per_seq, nwg, nitems);
      EMB_CHECK_LAUNCH();
      if constexpr (sizeof(T) == 2) {
        if (inl) {
          BnFinBwd fin{};
          fin.partial = (const float*)bpart; fin.rows = nwg; fin.dgamma = (float*)dgamma; fin.dbeta = (float*)dbeta; fin.coef = (float*)coef;
          fin.count = (double)R;
          const int rpb = TY * 8, nblk = cdiv(R, rpb);
          bn_bwd_affine_fin_kernel<<<nblk < 512 ? nblk : 512, 256, 0, s>>>((const __bf16*)y, (const float*)stats, fin, (__bf16*)dy, R, Cout, training, rpb);
          EMB_CHECK_LAUNCH();
          affine_done = true;
        }
      }
      if (!affine_done) {
        bn_bwd_finalize_kernel<P, P><<<Cout, 256, 0, s>>>(bpart, nwg, Cout, (double)R, nullptr, (P*)dgamma, (P*)dbeta, coef,
                                                         bn_phase == 1 ? bn_sums : nullptr);
        EMB_CHECK_LAUNCH();
      }
      if (bn_phase == 1) return EMB_OK;
    } else {
      bn_bwd_finalize_kernel<P, double><<<Cout, 256, 0, s>>>(bn_sums, 1, Cout, 0.0, bn_sums + 2 * Cout, (P*)nullptr, (P*)nullptr, coef,
                                                            nullptr);
      EMB_CHECK_LAUNCH();
    }
    if (!affine_done) {
      const int rpb = TY * 8;   // rows per block of the elementwise pass
      bn_bwd_affine_kernel<T><<<cdiv(R, rpb), 256, 0, s>>>((const T*)y, (const P*)stats, coef, (T*)dy, R, Cout, training, rpb);
      EMB_CHECK_LAUNCH();
    }
  }
  // wgrad: reduction over all B*L rows, split into slices whose partial slabs are reduced in order
  bool dual = false;   // weight and input gradient in one launch (bf16 streaming kernels, conv_direct.hip)
  {
    int S = conv_wgrad_slices(B, L, cin_pad, pad, KK, Cout, dtype_code<T>());
    int rc = 1;
    if constexpr (sizeof(T) == 2) {
      if (dx != nullptr && wflip != nullptr) {
        rc = launch_conv_bwd_dual(dy, x, slab, wflip, dx, B, L, cin_pad, k, Cout, pad, S, s);
        dual = rc == EMB_OK;
      }
    }
    if constexpr (sizeof(T) == 4) {
      int S_ring = S;
      rc = gemm_jobs_conv_wgrad(dy, x, slab, B, L, cin_pad, KK, Cout, pad, &S_ring, s);
      if (rc == EMB_OK) S = S_ring;
    }
    if (rc == 1) rc = launch_conv_wgrad_direct(dtype_code<T>(), dy, x, slab, B, L, cin_pad, KK, Cout, pad, S, s);
    if (rc == 1) {
      const int tiles_n = cdiv(KK + 1, CW::BN), tiles = cdiv(Cout, CW::BM) * tiles_n;
      const int vec_dy = (Cout % VEC == 0) && aligned16(dy), vec_x = (cin_pad % VEC == 0) && aligned16(x);
      constexpr int lds = gemm_tile_lds<CW>();
      static bool attr_set = false;
      if (!attr_set && lds > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<CW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
      }
      S = w.S;
      conv_wgrad_kernel<CW><<<tiles * S, kThreads, lds, s>>>((const T*)dy, (const T*)x, slab, R, L, cin_pad, KK, Cout, pad, w.kper,
                                                            tiles_n, tiles, vec_dy, vec_x);
      EMB_CHECK_LAUNCH();
    } else if (rc != EMB_OK) {
      return rc;
    }
    ReduceJob j{};   // slabs -> dW (torch layout, real channels) / dbias, slices summed in fixed order (reduce.hip)
    j.in = slab; j.out[0] = dW; j.out[1] = dbias; j.per = (long)Cout * (KK + 1); j.S = S; j.kind = RJ_CONV;
    j.iv[0] = Cin; j.iv[1] = cin_pad; j.iv[2] = k;
    const int rcr = reduce_submit(j, sizeof(P) == 8, s);
    if (rcr != EMB_OK) return rcr;
  }
  if (dx != nullptr && !dual) {   // dgrad: the same conv-view GEMM on dy with flipped taps
    int rc = 1;
    if constexpr (sizeof(T) == 4) rc = gemm_jobs_conv(false, dy, wflip, nullptr, dx, nullptr, nullptr, B, L, Cout, k * Cout, cin_pad, pad, s);
    if (rc == 1) rc = launch_conv_direct(dtype_code<T>(), false, dy, wflip, nullptr, dx, nullptr, nullptr, B, L, Cout, k * Cout, cin_pad, pad, s);
    if (rc == 1) {
      if (cin_pad >= 64) rc = launch_conv_gemm<typename ConvCfg<T>::F64, false>(dy, wflip, nullptr, dx, nullptr, R, L, Cout, k * Cout, cin_pad, pad, s);
      else rc = launch_conv_gemm<typename ConvCfg<T>::F32, false>(dy, wflip, nullptr, dx, nullptr, R, L, Cout, k * Cout, cin_pad, pad, s);
    }
    if (rc != EMB_OK) return rc;
  }
  return EMB_OK;
}

template <typename S> static int ncl_to_nlc_from(const void* x, void* out, int dd, int B, int C, int L, int Cpad, hipStream_t s) {
  const long n = (long)B * L;
  const int grid = (int)((n + 255) / 256);
  switch (dd) {
    case EMB_F32: ncl_to_nlc_kernel<S, float><<<grid, 256, 0, s>>>((const S*)x, (float*)out, B, C, L, Cpad); break;
    case EMB_BF16: ncl_to_nlc_kernel<S, __bf16><<<grid, 256, 0, s>>>((const S*)x, (__bf16*)out, B, C, L, Cpad); break;
    case EMB_F64: ncl_to_nlc_kernel<S, double><<<grid, 256, 0, s>>>((const S*)x, (double*)out, B, C, L, Cpad); break;
    default: set_error("emb_ncl_to_nlc: unsupported dst dtype %d", dd); return EMB_ERR_DTYPE;
  }
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

}  // namespace emb

using namespace emb;

extern "C" int64_t emb_convblock_workspace_bytes(int B, int L, int cin_pad, int Cout, int k, int dtype) {
  if (B <= 0 || L <= 0 || cin_pad <= 0 || Cout <= 0 || k <= 0) return -1;
  switch (dtype) {
    case EMB_F32: return (int64_t)conv_workspace<float>(B, L, cin_pad, Cout, k).total;
    case EMB_BF16: return (int64_t)conv_workspace<__bf16>(B, L, cin_pad, Cout, k).total;
    case EMB_F64: return (int64_t)conv_workspace<double>(B, L, cin_pad, Cout, k).total;
  }
  return -1;
}

extern "C" int emb_ncl_to_nlc(const void* x, int src_dtype, void* out, int dst_dtype, int B, int C, int L, int Cpad,
                              emb_stream_t stream) {
  EMB_CHECK_ARG(x && out && B > 0 && C > 0 && L > 0 && Cpad >= C, "emb_ncl_to_nlc: bad argument");
  hipStream_t s = (hipStream_t)stream;
  switch (src_dtype) {
    case EMB_F32: return ncl_to_nlc_from<float>(x, out, dst_dtype, B, C, L, Cpad, s);
    case EMB_BF16: return ncl_to_nlc_from<__bf16>(x, out, dst_dtype, B, C, L, Cpad, s);
    case EMB_F64: return ncl_to_nlc_from<double>(x, out, dst_dtype, B, C, L, Cpad, s);
  }
  set_error("emb_ncl_to_nlc: unsupported src dtype %d", src_dtype);
  return EMB_ERR_DTYPE;
}

extern "C" int emb_conv_pack_weight(const void* W, void* wpack, void* wflip, int Cout, int Cin, int cin_pad, int k, int dtype,
                                    emb_stream_t stream) {
  EMB_CHECK_ARG(W && wpack && Cout > 0 && Cin > 0 && cin_pad >= Cin && k > 0, "emb_conv_pack_weight: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const long n = (long)Cout * k * cin_pad * 2;
  const int grid = (int)((n + 255) / 256);
  switch (dtype) {
    case EMB_F32: conv_pack_weight_kernel<float, float><<<grid, 256, 0, s>>>((const float*)W, (float*)wpack, (float*)wflip, Cout, Cin, cin_pad, k); break;
    case EMB_BF16: conv_pack_weight_kernel<float, __bf16><<<grid, 256, 0, s>>>((const float*)W, (__bf16*)wpack, (__bf16*)wflip, Cout, Cin, cin_pad, k); break;
    case EMB_F64: conv_pack_weight_kernel<double, double><<<grid, 256, 0, s>>>((const double*)W, (double*)wpack, (double*)wflip, Cout, Cin, cin_pad, k); break;
    default: set_error("emb_conv_pack_weight: unsupported dtype %d", dtype); return EMB_ERR_DTYPE;
  }
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

extern "C" int emb_convblock_fwd(const void* x, const void* wpack, const void* bias, const void* gamma, const void* beta,
                                 void* running_mean, void* running_var, int training, double momentum, double eps,
                                 float dropout_p, uint64_t seed, uint64_t step_val, const uint64_t* step_dev, int64_t row0,
                                 int layer_id, void* y, void* stats, void* out, uint8_t* argmax, int out_ncl, void* workspace,
                                 int64_t workspace_bytes, int64_t* num_batches_tracked, int x_codes, int bn_phase, double* bn_sums,
                                 int B, int L, int cin_pad, int Cout, int k, int dtype, emb_stream_t stream) {
  EMB_CHECK_ARG(x && wpack && bias && gamma && beta && running_mean && running_var && stats && out && argmax && workspace,
                "emb_convblock_fwd: null pointer");
  EMB_CHECK_ARG(B > 0 && L > 0 && cin_pad > 0 && Cout > 0 && k > 0 && (k & 1), "emb_convblock_fwd: bad dims");
  EMB_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "emb_convblock_fwd: dropout_p must be in [0,1)");
  EMB_CHECK_ARG(bn_phase == 0 || ((bn_phase == 1 || bn_phase == 2) && bn_sums != nullptr && training),
                "emb_convblock_fwd: bn_phase is 0, or 1 / 2 with bn_sums in training mode");
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case EMB_F32: return convblock_fwd<float>(x, wpack, bias, gamma, beta, running_mean, running_var, training, momentum, eps, dropout_p, seed, step_val, step_dev, row0, layer_id, y, stats, out, argmax, out_ncl, workspace, workspace_bytes, num_batches_tracked, x_codes, bn_phase, bn_sums, B, L, cin_pad, Cout, k, s);
    case EMB_BF16: return convblock_fwd<__bf16>(x, wpack, bias, gamma, beta, running_mean, running_var, training, momentum, eps, dropout_p, seed, step_val, step_dev, row0, layer_id, y, stats, out, argmax, out_ncl, workspace, workspace_bytes, num_batches_tracked, x_codes, bn_phase, bn_sums, B, L, cin_pad, Cout, k, s);
    case EMB_F64: return convblock_fwd<double>(x, wpack, bias, gamma, beta, running_mean, running_var, training, momentum, eps, dropout_p, seed, step_val, step_dev, row0, layer_id, y, stats, out, argmax, out_ncl, workspace, workspace_bytes, num_batches_tracked, x_codes, bn_phase, bn_sums, B, L, cin_pad, Cout, k, s);
  }
  set_error("emb_convblock_fwd: unsupported dtype %d", dtype);
  return EMB_ERR_DTYPE;
}

extern "C" int64_t emb_convblock_stats_elems(int B, int L, int cin_pad, int Cout, int k, int dtype) {
  return 4 * (int64_t)Cout + ((dtype == EMB_BF16 && conv_first_supported(dtype, B, L, cin_pad, Cout, k)) ? conv_first_gram_floats() : 0);
}

extern "C" int emb_convblock_first_linear(int on) {
  const int was = first_linear_enabled() ? 1 : 0;
  if (on >= 0) g_first_linear = on != 0;
  return was;
}

extern "C" int emb_convblock_needs_y(int B, int L, int cin_pad, int Cout, int k, int dtype) {
  return conv_first_supported(dtype, B, L, cin_pad, Cout, k) ? 0 : 1;
}

extern "C" int emb_convblock_bwd(const void* dout, int dout_ncl, const uint8_t* argmax, const void* y, const void* stats,
                                 const void* x, const void* wflip, const void* wpack, const void* bias, float dropout_p,
                                 int training, void* dx, void* dW, void* dbias, void* dgamma, void* dbeta, void* dy, void* workspace,
                                 int64_t workspace_bytes, int x_codes, int bn_phase, double* bn_sums, int B, int L, int Cin, int cin_pad,
                                 int Cout, int k, int dtype, emb_stream_t stream) {
  EMB_CHECK_ARG(dout && argmax && stats && x && dW && dbias && dgamma && dbeta && workspace && (y == nullptr || dy != nullptr),
                "emb_convblock_bwd: null pointer");
  EMB_CHECK_ARG(dx == nullptr || wflip != nullptr, "emb_convblock_bwd: wflip is required when dx is requested");
  EMB_CHECK_ARG(bn_phase == 0 || ((bn_phase == 1 || bn_phase == 2) && bn_sums != nullptr && training),
                "emb_convblock_bwd: bn_phase is 0, or 1 / 2 with bn_sums in training mode");
  EMB_CHECK_ARG(B > 0 && L > 0 && Cin > 0 && cin_pad >= Cin && Cout > 0 && k > 0 && (k & 1), "emb_convblock_bwd: bad dims");
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case EMB_F32: return convblock_bwd<float>(dout, dout_ncl, argmax, y, stats, x, wflip, wpack, bias, dropout_p, training, dx, dW, dbias, dgamma, dbeta, dy, workspace, workspace_bytes, x_codes, bn_phase, bn_sums, B, L, Cin, cin_pad, Cout, k, s);
    case EMB_BF16: return convblock_bwd<__bf16>(dout, dout_ncl, argmax, y, stats, x, wflip, wpack, bias, dropout_p, training, dx, dW, dbias, dgamma, dbeta, dy, workspace, workspace_bytes, x_codes, bn_phase, bn_sums, B, L, Cin, cin_pad, Cout, k, s);
    case EMB_F64: return convblock_bwd<double>(dout, dout_ncl, argmax, y, stats, x, wflip, wpack, bias, dropout_p, training, dx, dW, dbias, dgamma, dbeta, dy, workspace, workspace_bytes, x_codes, bn_phase, bn_sums, B, L, Cin, cin_pad, Cout, k, s);
  }
  set_error("emb_convblock_bwd: unsupported dtype %d", dtype);
  return EMB_ERR_DTYPE;
}
