// Step epilogue kernels: per-batch class weights + weighted 2-class cross-entropy + confusion counts
// (utils/utils.py:121-140, :80-94; utils/training_models_multimodal.py:140-141,151-154), the optimizers
// the reference can pick (utils/training_models_multimodal.py:318-325: Adam, RMSprop from torch, Nadam
// from timm), dtype casts and the device step counter.  All bandwidth-trivial; written so that no value
// ever has to visit the host during a step (hipGraph-capturable).
#include "common.h"

namespace emb {

template <typename V> __device__ __forceinline__ V block_sum(V v, V* scratch) {
  // fixed-order tree reduction: deterministic for a given block size
  const int tid = threadIdx.x;
  scratch[tid] = v;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if (tid < s) scratch[tid] += scratch[tid + s];
    __syncthreads();
  }
  const V r = scratch[0];
  __syncthreads();
  return r;
}

constexpr int kCeThreads = 1024;

template <typename T>
__global__ __launch_bounds__(kCeThreads) void weighted_ce_kernel(const T* __restrict__ logits, const int64_t* __restrict__ target,
                                                                 int64_t* __restrict__ class_counts, int global_counts,
                                                                 float* __restrict__ loss, T* __restrict__ dlogits,
                                                                 int64_t* __restrict__ confusion, uint64_t* __restrict__ tick_a,
                                                                 uint64_t* __restrict__ tick_b, int B) {
  __shared__ double sd[kCeThreads];
  __shared__ long long sl[kCeThreads];
  const int tid = threadIdx.x;
  long long pos_l = 0;
  for (int i = tid; i < B; i += kCeThreads) pos_l += (target[i] == 1);
  const long long pos_local = block_sum<long long>(pos_l, sl);
  long long pos = pos_local, n = B;
  if (global_counts) {
    pos = class_counts[0];
    n = class_counts[1];
  } else if (tid == 0) {
    class_counts[0] = pos;
    class_counts[1] = n;
  }
  const long long neg = n - pos;
  // utils/utils.py:134-140 in double, then torch.tensor([w_neg, w_pos]) -> fp32
  const double pos_inv = pos != 0 ? 1.0 / (double)pos : 0.0;
  const double neg_inv = neg != 0 ? 1.0 / (double)neg : 0.0;
  const float w1 = (float)(pos_inv / (neg_inv + pos_inv));   // weight of class 1
  const float w0 = (float)(neg_inv / (neg_inv + pos_inv));   // weight of class 0
  const double den = (double)w1 * (double)pos + (double)w0 * (double)neg;   // sum_i w[y_i] over the global batch
  double num_l = 0.0;
  long long tp_l = 0, pp_l = 0;
  for (int i = tid; i < B; i += kCeThreads) {
    const float z0 = (float)logits[2 * (long)i], z1 = (float)logits[2 * (long)i + 1];   // output.float()
    const int y = target[i] == 1 ? 1 : 0;
    const float zm = fmaxf(z0, z1);
    const float e0 = expf(z0 - zm), e1 = expf(z1 - zm);
    const float lse = zm + logf(e0 + e1);
    const float wy = y ? w1 : w0;
    num_l += (double)wy * (double)(lse - (y ? z1 : z0));
    const int pred = z1 > z0 ? 1 : 0;   // torch.argmax: first maximum wins ties
    tp_l += (pred & y);
    pp_l += pred;
    if (dlogits != nullptr) {
      const float inv = 1.0f / (e0 + e1);
      const float g = (float)((double)wy / den);
      dlogits[2 * (long)i] = (T)(g * (e0 * inv - (y ? 0.0f : 1.0f)));
      dlogits[2 * (long)i + 1] = (T)(g * (e1 * inv - (y ? 1.0f : 0.0f)));
    }
  }
  const double num = block_sum<double>(num_l, sd);
  const long long tp = block_sum<long long>(tp_l, sl);
  const long long pp = block_sum<long long>(pp_l, sl);
  if (tid == 0) {
    loss[0] = (float)(num / den);
    if (confusion != nullptr) {
      confusion[0] = tp;
      confusion[1] = pp;
      confusion[2] = pos_local;
      confusion[3] = B;
    }
    // step counters that advance once per loss evaluation (the model's RNG step after its forward, the optimizer's
    // step before its update): folded in here so that a training step needs no counter launches of its own
    if (tick_a != nullptr) tick_a[0] += 1;
    if (tick_b != nullptr) tick_b[0] += 1;
  }
}

__global__ __launch_bounds__(kCeThreads) void count_labels_kernel(const int64_t* __restrict__ target, int64_t* __restrict__ cc, int B) {
  __shared__ long long sl[kCeThreads];
  long long p = 0;
  for (int i = threadIdx.x; i < B; i += kCeThreads) p += (target[i] == 1);
  const long long pos = block_sum<long long>(p, sl);
  if (threadIdx.x == 0) {
    cc[0] = pos;
    cc[1] = B;
  }
}

// ------------------------------------------------------------------------------------ optimizers
template <typename P>
__global__ void adam_kernel(P* __restrict__ p, const P* __restrict__ g, P* __restrict__ m, P* __restrict__ v,
                            __bf16* __restrict__ shadow, int64_t n, double lr, double b1, double b2, double eps, double wd,
                            uint64_t step_val, const uint64_t* __restrict__ step_dev) {
  const double step = (double)(step_val + (step_dev ? *step_dev : 0));
  const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
  const P step_size = (P)(lr / bc1), bc2s = (P)sqrt(bc2);
  const P pb2 = (P)b2, pwd = (P)wd, peps = (P)eps, omb1 = (P)(1.0 - b1), omb2 = (P)(1.0 - b2);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    P pi = p[i];
    const P gi = g[i] + pwd * pi;                 // coupled L2 (torch.optim.Adam weight_decay)
    const P mi = m[i] + (gi - m[i]) * omb1;       // lerp
    const P vi = v[i] * pb2 + gi * gi * omb2;
    const P denom = sqrt(vi) / bc2s + peps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (shadow) shadow[i] = (__bf16)(float)pi;
  }
}

template <typename P>
__global__ void rmsprop_kernel(P* __restrict__ p, const P* __restrict__ g, P* __restrict__ sq, __bf16* __restrict__ shadow,
                               int64_t n, double lr, double alpha, double eps, double wd) {
  const P pa = (P)alpha, oma = (P)(1.0 - alpha), pwd = (P)wd, peps = (P)eps, plr = (P)lr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    P pi = p[i];
    const P gi = g[i] + pwd * pi;
    const P si = sq[i] * pa + gi * gi * oma;
    pi -= plr * (gi / (sqrt(si) + peps));
    p[i] = pi; sq[i] = si;
    if (shadow) shadow[i] = (__bf16)(float)pi;
  }
}

// timm.optim.Nadam (not installed in this image: restated from its published algorithm -- Dozat 2016
// with the warm momentum schedule mu_t = beta1 (1 - 0.5 * 0.96^(t * schedule_decay))).  m_schedule is a
// two-slot ping-pong (read slot (t+1)&1, write slot t&1) so that no thread reads what another writes.
template <typename P>
__global__ void nadam_kernel(P* __restrict__ p, const P* __restrict__ g, P* __restrict__ m, P* __restrict__ v,
                             double* __restrict__ m_schedule, __bf16* __restrict__ shadow, int64_t n, double lr, double b1,
                             double b2, double eps, double wd, double sdecay, uint64_t step_val,
                             const uint64_t* __restrict__ step_dev) {
  const uint64_t t = step_val + (step_dev ? *step_dev : 0);
  const double td = (double)t;
  const double mu_t = b1 * (1.0 - 0.5 * pow(0.96, td * sdecay));
  const double mu_n = b1 * (1.0 - 0.5 * pow(0.96, (td + 1.0) * sdecay));
  const double ms_old = m_schedule[(t + 1) & 1];
  const double ms_new = ms_old * mu_t, ms_next = ms_new * mu_n;
  const double bc2 = 1.0 - pow(b2, td);
  const P c_g = (P)(lr * (1.0 - mu_t) / (1.0 - ms_new)), c_m = (P)(lr * mu_n / (1.0 - ms_next));
  const P bc2s = (P)sqrt(bc2), pb1 = (P)b1, pb2 = (P)b2, omb1 = (P)(1.0 - b1), omb2 = (P)(1.0 - b2), pwd = (P)wd, peps = (P)eps;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    P pi = p[i];
    const P gi = g[i] + pwd * pi;
    const P mi = m[i] * pb1 + gi * omb1;
    const P vi = v[i] * pb2 + gi * gi * omb2;
    const P denom = sqrt(vi) / bc2s + peps;
    pi -= c_g * (gi / denom);
    pi -= c_m * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (shadow) shadow[i] = (__bf16)(float)pi;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) m_schedule[t & 1] = ms_new;
}

template <typename S, typename D> __global__ void cast_kernel(const S* __restrict__ s, D* __restrict__ d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if constexpr (sizeof(D) == 2) d[i] = (D)(float)s[i];
    else d[i] = (D)(typename AccOf<S>::type)s[i];
}

__global__ void counter_add_kernel(uint64_t* c, uint64_t inc) { *c += inc; }

static inline int ew_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

template <typename S> static int cast_from(const void* src, void* dst, int dd, int64_t n, hipStream_t s) {
  const int g = ew_grid(n);
  switch (dd) {
    case EMB_F32: cast_kernel<S, float><<<g, 256, 0, s>>>((const S*)src, (float*)dst, n); break;
    case EMB_BF16: cast_kernel<S, __bf16><<<g, 256, 0, s>>>((const S*)src, (__bf16*)dst, n); break;
    case EMB_F64: cast_kernel<S, double><<<g, 256, 0, s>>>((const S*)src, (double*)dst, n); break;
    default: set_error("emb_cast: unsupported dst dtype %d", dd); return EMB_ERR_DTYPE;
  }
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

}  // namespace emb

using namespace emb;

extern "C" int emb_weighted_ce(const void* logits, const int64_t* target, int64_t* class_counts, int global_counts, float* loss,
                               void* dlogits, int64_t* confusion, uint64_t* tick_a, uint64_t* tick_b, int B, int dtype,
                               emb_stream_t stream) {
  EMB_CHECK_ARG(logits && target && class_counts && loss, "emb_weighted_ce: null pointer");
  EMB_CHECK_ARG(B > 0, "emb_weighted_ce: B must be positive (got %d)", B);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case EMB_F32: weighted_ce_kernel<float><<<1, kCeThreads, 0, s>>>((const float*)logits, target, class_counts, global_counts, loss, (float*)dlogits, confusion, tick_a, tick_b, B); break;
    case EMB_BF16: weighted_ce_kernel<__bf16><<<1, kCeThreads, 0, s>>>((const __bf16*)logits, target, class_counts, global_counts, loss, (__bf16*)dlogits, confusion, tick_a, tick_b, B); break;
    case EMB_F64: weighted_ce_kernel<double><<<1, kCeThreads, 0, s>>>((const double*)logits, target, class_counts, global_counts, loss, (double*)dlogits, confusion, tick_a, tick_b, B); break;
    default: set_error("emb_weighted_ce: unsupported dtype %d", dtype); return EMB_ERR_DTYPE;
  }
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

extern "C" int emb_count_labels(const int64_t* target, int64_t* class_counts, int B, emb_stream_t stream) {
  EMB_CHECK_ARG(target && class_counts && B >= 0, "emb_count_labels: bad argument");
  count_labels_kernel<<<1, kCeThreads, 0, (hipStream_t)stream>>>(target, class_counts, B);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

extern "C" int emb_adam_step(void* param, const void* grad, void* exp_avg, void* exp_avg_sq, void* bf16_shadow, int64_t n,
                             double lr, double beta1, double beta2, double eps, double weight_decay, uint64_t step_val,
                             const uint64_t* step_dev, int dtype, emb_stream_t stream) {
  EMB_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n >= 0, "emb_adam_step: bad argument");
  if (n == 0) return EMB_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EMB_F64)
    adam_kernel<double><<<ew_grid(n), 256, 0, s>>>((double*)param, (const double*)grad, (double*)exp_avg, (double*)exp_avg_sq, (__bf16*)bf16_shadow, n, lr, beta1, beta2, eps, weight_decay, step_val, step_dev);
  else if (dtype == EMB_F32 || dtype == EMB_BF16)
    adam_kernel<float><<<ew_grid(n), 256, 0, s>>>((float*)param, (const float*)grad, (float*)exp_avg, (float*)exp_avg_sq, (__bf16*)bf16_shadow, n, lr, beta1, beta2, eps, weight_decay, step_val, step_dev);
  else { set_error("emb_adam_step: unsupported dtype %d", dtype); return EMB_ERR_DTYPE; }
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

extern "C" int emb_rmsprop_step(void* param, const void* grad, void* square_avg, void* bf16_shadow, int64_t n, double lr,
                                double alpha, double eps, double weight_decay, int dtype, emb_stream_t stream) {
  EMB_CHECK_ARG(param && grad && square_avg && n >= 0, "emb_rmsprop_step: bad argument");
  if (n == 0) return EMB_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EMB_F64)
    rmsprop_kernel<double><<<ew_grid(n), 256, 0, s>>>((double*)param, (const double*)grad, (double*)square_avg, (__bf16*)bf16_shadow, n, lr, alpha, eps, weight_decay);
  else if (dtype == EMB_F32 || dtype == EMB_BF16)
    rmsprop_kernel<float><<<ew_grid(n), 256, 0, s>>>((float*)param, (const float*)grad, (float*)square_avg, (__bf16*)bf16_shadow, n, lr, alpha, eps, weight_decay);
  else { set_error("emb_rmsprop_step: unsupported dtype %d", dtype); return EMB_ERR_DTYPE; }
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

extern "C" int emb_nadam_step(void* param, const void* grad, void* exp_avg, void* exp_avg_sq, double* m_schedule,
                              void* bf16_shadow, int64_t n, double lr, double beta1, double beta2, double eps,
                              double weight_decay, double schedule_decay, uint64_t step_val, const uint64_t* step_dev,
                              int dtype, emb_stream_t stream) {
  EMB_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && m_schedule && n >= 0, "emb_nadam_step: bad argument");
  if (n == 0) return EMB_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EMB_F64)
    nadam_kernel<double><<<ew_grid(n), 256, 0, s>>>((double*)param, (const double*)grad, (double*)exp_avg, (double*)exp_avg_sq, m_schedule, (__bf16*)bf16_shadow, n, lr, beta1, beta2, eps, weight_decay, schedule_decay, step_val, step_dev);
  else if (dtype == EMB_F32 || dtype == EMB_BF16)
    nadam_kernel<float><<<ew_grid(n), 256, 0, s>>>((float*)param, (const float*)grad, (float*)exp_avg, (float*)exp_avg_sq, m_schedule, (__bf16*)bf16_shadow, n, lr, beta1, beta2, eps, weight_decay, schedule_decay, step_val, step_dev);
  else { set_error("emb_nadam_step: unsupported dtype %d", dtype); return EMB_ERR_DTYPE; }
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

extern "C" int emb_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, emb_stream_t stream) {
  EMB_CHECK_ARG(src && dst && n >= 0, "emb_cast: bad argument");
  if (n == 0) return EMB_OK;
  hipStream_t s = (hipStream_t)stream;
  switch (src_dtype) {
    case EMB_F32: return cast_from<float>(src, dst, dst_dtype, n, s);
    case EMB_BF16: return cast_from<__bf16>(src, dst, dst_dtype, n, s);
    case EMB_F64: return cast_from<double>(src, dst, dst_dtype, n, s);
  }
  set_error("emb_cast: unsupported src dtype %d", src_dtype);
  return EMB_ERR_DTYPE;
}

extern "C" int emb_counter_add(uint64_t* counter, uint64_t inc, emb_stream_t stream) {
  EMB_CHECK_ARG(counter, "emb_counter_add: null pointer");
  counter_add_kernel<<<1, 1, 0, (hipStream_t)stream>>>(counter, inc);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

// ------------------------------------------------------------------- multi-tensor optimizer launch
// One launch updates up to kMaxTensors parameter tensors (the reference's models have <= 34).  The pointer
// table travels BY VALUE in the kernel argument (baked into a hipGraph node at capture time, rebuilt for free
// on every eager call) and is first copied to LDS, so that it is never indexed dynamically in the kernarg
// segment (see the hipcc note in embrace_bwd.hip).
#include <cstring>
#include <mutex>
#include <unordered_map>
#include "reduce.h"
#include "rider.h"
#include "first_gram.h"
namespace emb {
int launch_jobs_f32(const ReduceJob* jobs, int n, hipStream_t s);   // reduce.hip
int launch_jobs_f64(const ReduceJob* jobs, int n, hipStream_t s);
constexpr int kMaxTensors = 40;
constexpr int kChunk = 1024;   // elements per block (of a tensor whose gradient is a plain tensor)
constexpr int kMaxSrc = 12;    // queued slab reductions one launch can take over (reduce.h)
// Gradients that arrive as S slices of a slab (the backward kernels' partial sums, reduce.h): blocks walk the SLAB in its
// own element order (coalesced 16-byte reads of every slice, as reduce.hip does), sum the slices in fixed order and update
// the parameter element the slab element belongs to.
struct SlabSrc {
  const void* in;
  long long per;
  int S, kind, lanes, vec;     // lanes (power of two <= 16) threads share one element's slices; vec = elements per thread (4 or 1)
  int blk_end;                 // exclusive end of this job's block range (after the plain tensors' blocks)
  int iv[9];
  signed char tensor[8];       // job output -> tensor index of this launch (-1: not ours)
};
template <typename P> struct MultiArgs {
  P* p[kMaxTensors];
  P* g[kMaxTensors];            // gradient tensor: read (plain) or WRITTEN (slab-sourced: the summed gradient is kept there)
  P* m[kMaxTensors];
  P* v[kMaxTensors];
  __bf16* sh[kMaxTensors];
  __bf16* flip[kMaxTensors];     // conv weights: tap-flipped packed copy (nullable)
  short pk_k[kMaxTensors], pk_cin[kMaxTensors], pk_cinpad[kMaxTensors], pk_cout[kMaxTensors];   // pk_k == 0: plain shadow
  signed char pk_f32[kMaxTensors];   // the packed images hold float (fp32 compute) instead of bf16
  int n[kMaxTensors];
  int blk_end[kMaxTensors];     // plain-gradient tensors only (a slab-sourced tensor has no blocks of its own)
  SlabSrc slab[kMaxSrc];
  int nsrc;
  // queued statistics sums (head.hip: loss, confusion counts) ride along as the last blocks of the launch
  SlabSrc stats[2];
  float* stats_loss[2];
  long long* stats_conf[2];
  int nstats;
  int count;
  // the parked finish of the first conv block's recompute-free backward (first_fin.h): its C channels run as the FIRST nff
  // workgroups of the launch and update the block's four tensors themselves (ff_t: weight, bias, gamma, beta; -1: not ours)
  FirstFinArgs ff;
  int nff;
  signed char ff_t[4];
  SmallCopy copy;               // a parked copy of a few floats (reduce.h): workgroup 0 does it
};
static_assert(sizeof(MultiArgs<float>) + 128 <= 4096, "kernel argument block");
enum { OPT_ADAM = 0, OPT_RMSPROP = 1, OPT_NADAM = 2 };
struct Hyper {
  double lr, b1, b2, eps, wd, alpha, sdecay;
  unsigned long long step_val;
  const uint64_t* step_dev;
  double* m_schedule;
};

template <typename P> struct OptConst {
  P c1, c2, bc2s, pb1, pb2, omb1, omb2, pwd, peps, pa, oma, plr;
};

// update element i of tensor t with the (raw, weight-decay-free) gradient g.  Two halves, so that a thread with several elements
// can issue all its loads before the first store (parameter, state and gradient pointers may alias as far as the compiler knows:
// element by element, every update would wait for the previous one's stores)
template <typename P> struct OptElem { P p, m, v; };
template <typename P, int OPT>
__device__ __forceinline__ OptElem<P> opt_load(const MultiArgs<P>& a, int t, long i) {
  OptElem<P> e;
  e.p = a.p[t][i];
  e.m = OPT == OPT_RMSPROP ? (P)0 : a.m[t][i];
  e.v = a.v[t][i];
  return e;
}
// the update itself: element (p, m, v) and its gradient (raw, weight-decay-free) -> the new element
template <typename P, int OPT>
__device__ __forceinline__ OptElem<P> opt_math(const OptConst<P>& k, P graw, const OptElem<P>& e) {
  OptElem<P> o;
  P pi = e.p;
  const P gi = graw + k.pwd * pi;
  if (OPT == OPT_ADAM) {
    const P mi = e.m + (gi - e.m) * k.omb1;
    const P vi = e.v * k.pb2 + gi * gi * k.omb2;
    pi -= k.c1 * (mi / (sqrt(vi) / k.bc2s + k.peps));
    o.m = mi; o.v = vi;
  } else if (OPT == OPT_RMSPROP) {
    const P si = e.v * k.pa + gi * gi * k.oma;
    pi -= k.plr * (gi / (sqrt(si) + k.peps));
    o.m = (P)0; o.v = si;
  } else {
    const P mi = e.m * k.pb1 + gi * k.omb1;
    const P vi = e.v * k.pb2 + gi * gi * k.omb2;
    const P denom = sqrt(vi) / k.bc2s + k.peps;
    pi -= k.c1 * (gi / denom);
    pi -= k.c2 * (mi / denom);
    o.m = mi; o.v = vi;
  }
  o.p = pi;
  return o;
}
template <typename P, int OPT>
__device__ __forceinline__ void opt_apply(const MultiArgs<P>& a, const OptConst<P>& k, int t, long i, P graw, const OptElem<P>& e) {
  P* p = a.p[t];
  P* m = a.m[t];
  P* v = a.v[t];
  const OptElem<P> o = opt_math<P, OPT>(k, graw, e);
  const P pi = o.p;
  if (OPT != OPT_RMSPROP) m[i] = o.m;
  v[i] = o.v;
  p[i] = pi;
  __bf16* sh = a.sh[t];
  if (sh) {
    const int kk = a.pk_k[t];
    if (kk == 0) {
      sh[i] = (__bf16)(float)pi;
    } else {   // registered conv weight W[o][ci][j]: keep its packed images current (see emb_conv_pack_register)
      const int cin = a.pk_cin[t], ii = (int)i, o = ii / (cin * kk), rem = ii - o * cin * kk, ci = rem / kk, j = rem - ci * kk;
      const long ip = ((long)o * kk + j) * a.pk_cinpad[t] + ci, ifl = ((long)ci * kk + (kk - 1 - j)) * a.pk_cout[t] + o;
      if (a.pk_f32[t]) {
        reinterpret_cast<float*>(sh)[ip] = (float)pi;
        if (a.flip[t]) reinterpret_cast<float*>(a.flip[t])[ifl] = (float)pi;
      } else {
        sh[ip] = (__bf16)(float)pi;
        if (a.flip[t]) a.flip[t][ifl] = (__bf16)(float)pi;
      }
    }
  }
}
template <typename P, int OPT>
__device__ __forceinline__ void opt_update(const MultiArgs<P>& a, const OptConst<P>& k, int t, long i, P graw) {
  opt_apply<P, OPT>(a, k, t, i, graw, opt_load<P, OPT>(a, t, i));
}
// four consecutive elements i0 .. i0 + 3 of tensor t with whole-vector accesses (their summed gradients g4 are also written to
// the tensor's .grad); false: not applicable (alignment, or a registered conv weight whose images need the per-element path)
template <typename P, int OPT>
__device__ __forceinline__ bool opt_update4(const MultiArgs<P>& a, const OptConst<P>& k, int t, long i0, const P (&g4)[4]) {
  typedef P P4 __attribute__((ext_vector_type(4)));
  P* p = a.p[t] + i0;
  P* m = OPT == OPT_RMSPROP ? p : a.m[t] + i0;
  P* v = a.v[t] + i0;
  P* g = a.g[t] + i0;
  __bf16* sh = a.sh[t];
  if (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(g)) &
       (sizeof(P4) - 1)) != 0 || (sh != nullptr && (a.pk_k[t] != 0 || (reinterpret_cast<uintptr_t>(sh + i0) & 7) != 0)))
    return false;
  const P4 pv = *reinterpret_cast<const P4*>(p), vv = *reinterpret_cast<const P4*>(v);
  P4 mv = {0, 0, 0, 0};
  if (OPT != OPT_RMSPROP) mv = *reinterpret_cast<const P4*>(m);
  P4 po, mo, vo, go;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const OptElem<P> o = opt_math<P, OPT>(k, g4[e], OptElem<P>{pv[e], mv[e], vv[e]});
    po[e] = o.p; mo[e] = o.m; vo[e] = o.v; go[e] = g4[e];
  }
  *reinterpret_cast<P4*>(g) = go;
  if (OPT != OPT_RMSPROP) *reinterpret_cast<P4*>(m) = mo;
  *reinterpret_cast<P4*>(v) = vo;
  *reinterpret_cast<P4*>(p) = po;
  if (sh != nullptr) {
    bf16x4 b;
#pragma unroll
    for (int e = 0; e < 4; ++e) b[e] = (__bf16)(float)po[e];
    *reinterpret_cast<bf16x4*>(sh + i0) = b;
  }
  return true;
}

// slab element q of a job -> (output of the job, element of that output); false: padding
__device__ __forceinline__ bool slab_target(const SlabSrc& j, long q, int* which, long* idx) {
  // (32-bit divisions: a slab has fewer than 2^31 elements, checked where the job is claimed; 64-bit ones are ~100 instructions each)
  if (j.kind == RJ_LINEAR) {
    const int N = j.iv[0], pitch = j.iv[1] > 0 ? j.iv[1] : N + 1;
    const unsigned qu = (unsigned)q, mu = qu / (unsigned)pitch;
    const long m = (long)mu;
    const int n = (int)(qu - mu * (unsigned)pitch);
    if (n > N) return false;
    *which = n == N ? 1 : 0;
    *idx = n == N ? m : m * N + n;
    return true;
  }
  if (j.kind == RJ_CONV) {
    const int Cin = j.iv[0], cin_pad = j.iv[1], k = j.iv[2], KK = k * cin_pad;
    const int o = (int)((unsigned)q / (unsigned)(KK + 1)), col = (int)((unsigned)q - (unsigned)o * (unsigned)(KK + 1));
    if (col == KK) { *which = 1; *idx = o; return true; }
    const int tap = col / cin_pad, ci = col - tap * cin_pad;
    if (ci >= Cin) return false;
    *which = 0;
    *idx = ((long)o * Cin + ci) * k + tap;
    return true;
  }
  long off = q;   // RJ_MLP: per layer [N*K weights | N biases]
  const int L = j.iv[0];
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    if (l < L) {
      const long nw = (long)j.iv[1 + l] * j.iv[5 + l];
      if (off < nw) { *which = l; *idx = off; return true; }
      off -= nw;
      if (off < j.iv[1 + l]) { *which = 4 + l; *idx = off; return true; }
      off -= j.iv[1 + l];
    }
  }
  return false;
}

// NT threads per workgroup: 256, or 1024 when the first conv block's finish rides along -- its per-channel workgroups are the
// launch's critical path and run the 1024-thread body of the standalone kernel (first_gram.h: 8.9 us against ~19 us with 256
// threads; the same summation order as the standalone finish).  Every other role is indifferent to NT: a workgroup still owns
// kChunk elements of a plain tensor / (NT / lanes) * vec slab elements, slices meet in lane order, the statistics sums keep
// their 32 x 8 shape.
// b^n for a step count n by squaring (<= 2 log2 n dependent multiplies; libm's pow(double, double) is several hundred
// instructions on the critical path of EVERY workgroup of the launch).  Relative error <= ~log2(n) ulp: far inside the bias
// corrections' use (torch computes them in Python floats; parity tests G7 / G9 hold at their bars).
__device__ __forceinline__ double powi_f64(double b, uint64_t n) {
  double r = 1.0;
  while (n) {
    if (n & 1) r *= b;
    b *= b;
    n >>= 1;
  }
  return r;
}

template <typename P, int OPT, int NT>
__global__ __launch_bounds__(NT) void multi_opt_kernel(const MultiArgs<P> args, const Hyper h) {
  __shared__ MultiArgs<P> a;
  __shared__ P red[4][NT];           // lane partial sums of slab-sourced gradients
  // (requested first: this round trip runs under the argument copy's instead of after it)
  const uint64_t step = h.step_val + (h.step_dev ? __builtin_nontemporal_load(h.step_dev) : 0);
  {
    const unsigned* src = reinterpret_cast<const unsigned*>(&args);
    unsigned* dst = reinterpret_cast<unsigned*>(&a);
    for (int i = threadIdx.x; i < (int)(sizeof(MultiArgs<P>) / 4); i += NT) dst[i] = src[i];
  }
  __syncthreads();
  if (blockIdx.x == 0 && (int)threadIdx.x < a.copy.n) a.copy.dst[threadIdx.x] = a.copy.src[threadIdx.x];
  const int bid = (int)blockIdx.x - a.nff;
  const int plain_end = a.blk_end[a.count - 1];
  const int slab_end = a.nsrc ? a.slab[a.nsrc - 1].blk_end : plain_end;
  if (bid >= 0 && bid >= slab_end) {     // statistics sums: one block per job, slices split over the 256 threads
    const int k = bid - slab_end;
    const SlabSrc& sj = a.stats[k];
    const P* in = (const P*)sj.in;
    const int q = threadIdx.x & 7, sl = threadIdx.x >> 3;       // per == 8 values per slice (head.hip)
    P acc = 0;
    if (threadIdx.x < 256) {                                    // (32 slice groups x 8 values whatever NT: one summation order)
      for (int s = sl; s < sj.S; s += 32) acc += in[(long)s * sj.per + q];
      red[0][threadIdx.x] = acc;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
      P sum = 0;
      for (int i = 0; i < 32; ++i) sum += red[0][i * 8 + threadIdx.x];
      if (threadIdx.x == 0) a.stats_loss[k][0] = (float)sum;
      else if (threadIdx.x <= 4 && a.stats_conf[k] != nullptr) a.stats_conf[k][threadIdx.x - 1] = (long long)(sum + (P)0.5);
    }
    return;
  }
  const double td = (double)step;
  OptConst<P> kc;
  kc.c1 = 0; kc.c2 = 0; kc.bc2s = 1;
  if (OPT == OPT_ADAM) {
    kc.c1 = (P)(h.lr / (1.0 - powi_f64(h.b1, step)));
    kc.bc2s = (P)sqrt(1.0 - powi_f64(h.b2, step));
  } else if (OPT == OPT_NADAM) {
    const double mu_t = h.b1 * (1.0 - 0.5 * pow(0.96, td * h.sdecay));
    const double mu_n = h.b1 * (1.0 - 0.5 * pow(0.96, (td + 1.0) * h.sdecay));
    const double ms_new = h.m_schedule[(step + 1) & 1] * mu_t, ms_next = ms_new * mu_n;
    kc.c1 = (P)(h.lr * (1.0 - mu_t) / (1.0 - ms_new));
    kc.c2 = (P)(h.lr * mu_n / (1.0 - ms_next));
    kc.bc2s = (P)sqrt(1.0 - powi_f64(h.b2, step));
    if (blockIdx.x == 0 && threadIdx.x == 0) h.m_schedule[step & 1] = ms_new;
  }
  kc.pb1 = (P)h.b1; kc.pb2 = (P)h.b2; kc.omb1 = (P)(1.0 - h.b1); kc.omb2 = (P)(1.0 - h.b2); kc.pwd = (P)h.wd; kc.peps = (P)h.eps;
  kc.pa = (P)h.alpha; kc.oma = (P)(1.0 - h.alpha); kc.plr = (P)h.lr;

  if constexpr (sizeof(P) == 4) {
    if (bid < 0) {            // finish of the first conv block's backward for channel blockIdx.x, parameters updated on the spot
      __shared__ __attribute__((aligned(16))) float fin_lds[first_finish_lds_floats<NT>()];
      struct Sink {
        const MultiArgs<P>& a;
        const OptConst<P>& kc;
        int c;
        __device__ __forceinline__ void one(int which, long idx, float v, float* fallback) const {
          const int t = a.ff_t[which];
          if (t < 0) { if (fallback) fallback[idx] = v; return; }
          a.g[t][idx] = v;       // the parameter's .grad holds the gradient, as after the standalone finish
          opt_update<P, OPT>(a, kc, t, idx, v);
        }
        __device__ __forceinline__ void scalars(float dgamma, float dbeta, float dbias) const {
          one(2, c, dgamma, a.ff.dgamma); one(3, c, dbeta, a.ff.dbeta); one(1, c, dbias, a.ff.dbias);
        }
        __device__ __forceinline__ void dw(long idx, float v) const { one(0, idx, v, a.ff.dW); }
      };
      first_finish_body<NT>(a.ff, (int)blockIdx.x, fin_lds, Sink{a, kc, (int)blockIdx.x});
      return;
    }
  }
  if (bid < plain_end) {      // a tensor whose gradient is a plain tensor
    int t = 0;
    while (t < a.count - 1 && bid >= a.blk_end[t]) ++t;
    const long base = (long)(bid - (t == 0 ? 0 : a.blk_end[t - 1])) * kChunk;
    const P* g = a.g[t];
    const int n = a.n[t];
    OptElem<P> el[kChunk / NT];
    P gr[kChunk / NT];
#pragma unroll
    for (int u = 0; u < kChunk / NT; ++u) {                // all loads first (see opt_load)
      const long i = base + u * NT + threadIdx.x;
      if (i < n) { el[u] = opt_load<P, OPT>(a, t, i); gr[u] = g[i]; }
    }
#pragma unroll
    for (int u = 0; u < kChunk / NT; ++u) {
      const long i = base + u * NT + threadIdx.x;
      if (i < n) opt_apply<P, OPT>(a, kc, t, i, gr[u], el[u]);
    }
    return;
  }
  // a queued slab reduction: lane sl of an element sums slices sl, sl + lanes, ...; the lane sums meet in lane order
  int k = 0;
  while (k < a.nsrc - 1 && bid >= a.slab[k].blk_end) ++k;
  const SlabSrc& sj = a.slab[k];
  const int jb = bid - (k == 0 ? plain_end : a.slab[k - 1].blk_end);
  const int lanes = sj.lanes, qpb = NT / lanes, qi = threadIdx.x % qpb, sl = threadIdx.x / qpb;
  const long per = sj.per;
  const P* in = (const P*)sj.in;
  P acc[4] = {0, 0, 0, 0};
  const long q0 = ((long)jb * qpb + qi) * sj.vec;
  if (lanes == 1 && sj.vec == 4) {
    // large slabs (the docking / post-layer weight gradients): every thread sums all slices of ITS four elements (independent
    // 16-byte loads) and updates them with whole-vector accesses of p, m, v, g -- no exchange, one memory round trip.  (With
    // several lanes per element the update below touches 4 bytes per thread at a 16-byte stride: a quarter of every sector.)
    if (q0 >= per) return;
    typedef P P4 __attribute__((ext_vector_type(4)));
    P4 s4 = {0, 0, 0, 0};
#pragma unroll 8
    for (int s = 0; s < sj.S; ++s) s4 += *reinterpret_cast<const P4*>(in + (long)s * per + q0);
    const P g4[4] = {s4[0], s4[1], s4[2], s4[3]};
    int w0 = 0, w3 = 0;
    long i0 = 0, i3 = 0;
    const bool ok0 = slab_target(sj, q0, &w0, &i0), ok3 = slab_target(sj, q0 + 3, &w3, &i3);
    if (ok0 && ok3 && w0 == w3 && i3 == i0 + 3 && sj.tensor[w0] >= 0 && opt_update4<P, OPT>(a, kc, sj.tensor[w0], i0, g4)) return;
    int tt[4];
    long ii[4];
    OptElem<P> el[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {                       // all loads first (see opt_load)
      int which;
      tt[e] = slab_target(sj, q0 + e, &which, &ii[e]) ? sj.tensor[which] : -1;
      if (tt[e] >= 0) el[e] = opt_load<P, OPT>(a, tt[e], ii[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (tt[e] < 0) continue;
      a.g[tt[e]][ii[e]] = g4[e];
      opt_apply<P, OPT>(a, kc, tt[e], ii[e], g4[e], el[e]);
    }
    return;
  }
  if (q0 < per) {
    if (sj.vec == 4) {
      typedef P P4 __attribute__((ext_vector_type(4)));
      P4 s4 = {0, 0, 0, 0};
#pragma unroll 8
      for (int s = sl; s < sj.S; s += lanes) s4 += *reinterpret_cast<const P4*>(in + (long)s * per + q0);
      acc[0] = s4[0]; acc[1] = s4[1]; acc[2] = s4[2]; acc[3] = s4[3];
    } else {
#pragma unroll 8
      for (int s = sl; s < sj.S; s += lanes) acc[0] += in[(long)s * per + q0];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[e][sl * qpb + qi] = acc[e];
  __syncthreads();
  // lane sl of the group updates element sl, sl + lanes, .. of the group's `vec` elements: with four or more lanes the elements
  // of a group are updated side by side (one memory round trip, not `vec`)
  if (q0 < per) {
    for (int e = sl; e < sj.vec; e += lanes) {
      P sum = 0;
      for (int i = 0; i < lanes; ++i) sum += red[e][i * qpb + qi];
      int which;
      long idx;
      if (!slab_target(sj, q0 + e, &which, &idx)) continue;
      const int t = sj.tensor[which];
      if (t < 0) continue;                   // that output of the job is not a tensor of this launch (it stays queued)
      a.g[t][idx] = sum;                     // the parameter's .grad holds the summed gradient, as after a reduction launch
      opt_update<P, OPT>(a, kc, t, idx, sum);
    }
  }
}

// ---- packed conv-weight registry: parameters whose packed bf16 images the optimizer launch maintains
struct PackDesc {
  void* wpack;
  void* wflip;
  int Cout, Cin, cin_pad, k, f32;
};
static std::unordered_map<const void*, PackDesc>& pack_table() {
  static std::unordered_map<const void*, PackDesc> t;
  return t;
}
static std::mutex& pack_mutex() {
  static std::mutex m;
  return m;
}

template <typename P, int OPT>
static int multi_launch(void* const* params, const void* const* grads, void* const* s1, void* const* s2, void* const* shadows,
                        const int64_t* sizes, int ntensors, const Hyper& h, hipStream_t s) {
  {   // a parked rider launch (rider.h) may be the producer of a slab this launch is about to consume
    const int rcr = rider_flush(s);
    if (rcr != EMB_OK) return rcr;
  }
  for (int off = 0; off < ntensors; off += kMaxTensors) {
    const int cnt = ntensors - off < kMaxTensors ? ntensors - off : kMaxTensors;
    MultiArgs<P> a;
    memset(&a, 0, sizeof(a));
    long long blocks = 0;
    int nsrc = 0;
    // a parked first-block finish (first_fin.h) rides in this launch when its weight, gamma and beta gradients are tensors of it
    int fft[4] = {-1, -1, -1, -1};
    FirstFinArgs ffj{};
    bool have_ff = false;
    if (sizeof(P) == 4 && first_fin_peek(s, &ffj)) {
      for (int i = 0; i < cnt; ++i) {
        const void* gp = grads[off + i];
        if (gp == nullptr) continue;
        if (gp == ffj.dW) fft[0] = i; else if (gp == ffj.dbias) fft[1] = i; else if (gp == ffj.dgamma) fft[2] = i; else if (gp == ffj.dbeta) fft[3] = i;
      }
      have_ff = fft[0] >= 0 && fft[2] >= 0 && fft[3] >= 0 && ffj.C <= 1024;
      if (have_ff) {
        first_fin_drop(s);
      } else {
        const int rcf = first_fin_flush(s);   // not ours: the classic launch
        if (rcf != EMB_OK) return rcf;
        fft[0] = fft[1] = fft[2] = fft[3] = -1;
      }
    }
    for (int i = 0; i < kMaxTensors; ++i) {
      const int j = off + (i < cnt ? i : 0);
      a.p[i] = (P*)params[j];
      a.g[i] = (P*)const_cast<void*>(grads[j]);
      a.m[i] = s1 ? (P*)s1[j] : nullptr;
      a.v[i] = s2 ? (P*)s2[j] : nullptr;
      a.sh[i] = shadows ? (__bf16*)shadows[j] : nullptr;
      a.flip[i] = nullptr;
      a.pk_k[i] = a.pk_cin[i] = a.pk_cinpad[i] = a.pk_cout[i] = 0;
      a.pk_f32[i] = 0;
      bool plain = true;
      if (have_ff && i < cnt && (i == fft[0] || i == fft[1] || i == fft[2] || i == fft[3])) plain = false;   // updated by the finish workgroups
      if (i < cnt) {
        if (sizes[j] >= (1ll << 31)) { set_error("optimizer step: tensor of %lld elements", (long long)sizes[j]); return EMB_ERR_ARG; }
        if (sizeof(P) == 4) {
          std::lock_guard<std::mutex> lk(pack_mutex());
          auto it = pack_table().find(params[j]);
          if (it != pack_table().end() && (int64_t)it->second.Cout * it->second.Cin * it->second.k == sizes[j] &&
              it->second.cin_pad < 32768 && it->second.Cout < 32768) {
            a.sh[i] = (__bf16*)it->second.wpack;
            a.flip[i] = (__bf16*)it->second.wflip;
            a.pk_k[i] = (short)it->second.k; a.pk_cin[i] = (short)it->second.Cin; a.pk_cinpad[i] = (short)it->second.cin_pad;
            a.pk_cout[i] = (short)it->second.Cout;
            a.pk_f32[i] = (signed char)it->second.f32;
          }
        }
        // a queued slab reduction that would have produced this gradient: its slices are summed in this launch (reduce.h)
        ReduceClaim cl;
        if (plain && reduce_claim(s, grads[j], sizeof(P) == 8, &cl)) {
          if (cl.job.per >= (1ll << 31)) { set_error("optimizer step: gradient slab of %lld elements", (long long)cl.job.per); return EMB_ERR_ARG; }
          int k = -1;
          for (int q = 0; q < nsrc; ++q)
            if (a.slab[q].in == cl.job.in && a.slab[q].per == cl.job.per) k = q;
          if (k < 0 && nsrc < kMaxSrc) {
            k = nsrc++;
            SlabSrc& d = a.slab[k];
            d.in = cl.job.in; d.per = cl.job.per; d.S = cl.job.S; d.kind = cl.job.kind;
            for (int q = 0; q < 9; ++q) d.iv[q] = cl.job.iv[q];
            for (int q = 0; q < 8; ++q) d.tensor[q] = -1;
            int lanes = 1;
            while (lanes < 16 && lanes < cl.job.S) lanes *= 2;
            d.vec = (cl.job.per % 4 == 0 && aligned16(cl.job.in) && (sizeof(P) == 4 || (reinterpret_cast<uintptr_t>(cl.job.in) & 31) == 0)) ? 4 : 1;
            // large row-major slabs: one thread per four elements, whole-vector updates (multi_opt_kernel)
            if (d.vec == 4 && cl.job.per >= 32768 && (cl.job.kind == RJ_LINEAR || cl.job.kind == RJ_MLP)) lanes = 1;
            d.lanes = lanes;
          }
          if (k >= 0) {
            a.slab[k].tensor[cl.which] = (signed char)i;
            plain = false;
          } else {                       // table full: run that one reduction the classic way, the gradient is then plain
            ReduceJob one = cl.job;
            for (int q = 0; q < 8; ++q) if (q != cl.which) one.out[q] = nullptr;
            const int rc = sizeof(P) == 8 ? launch_jobs_f64(&one, 1, s) : launch_jobs_f32(&one, 1, s);
            if (rc != EMB_OK) return rc;
          }
        }
      }
      a.n[i] = i < cnt ? (int)sizes[j] : 0;
      if (i < cnt && plain) blocks += (sizes[j] + kChunk - 1) / kChunk;
      a.blk_end[i] = (int)blocks;
    }
    a.count = cnt;
    a.nsrc = nsrc;
    // threads per workgroup (multi_opt_kernel): 1024 when a first-block finish rides along AND the rest of the launch is small
    // enough for the finish to be its critical path (cfg2: ~1100 workgroups of 256; with cfg5's 2.9 M docking weights the
    // 1024-thread launch was 29 us slower)
    int nt = 256;
    if (have_ff && sizeof(P) == 4) {
      long b256 = blocks;
      for (int k = 0; k < nsrc; ++k) {
        const long epb = (long)(256 / a.slab[k].lanes) * a.slab[k].vec;
        b256 += (a.slab[k].per + epb - 1) / epb;
      }
      if (b256 <= 2048) nt = 1024;
    }
    for (int k = 0; k < nsrc; ++k) {
      const long epb = (long)(nt / a.slab[k].lanes) * a.slab[k].vec;
      blocks += (a.slab[k].per + epb - 1) / epb;
      a.slab[k].blk_end = (int)blocks;
    }
    a.nstats = 0;
    ReduceJob sj;
    while (a.nstats < 2 && off + cnt >= ntensors && reduce_claim_stats(s, sizeof(P) == 8, &sj)) {   // (last chunk of the call only)
      SlabSrc& d = a.stats[a.nstats];
      d.in = sj.in; d.per = sj.per; d.S = sj.S; d.kind = sj.kind; d.lanes = 1;
      a.stats_loss[a.nstats] = (float*)sj.out[0];
      a.stats_conf[a.nstats] = (long long*)sj.out[1];
      ++a.nstats;
    }
    a.nff = 0;
    if (have_ff) {
      a.ff = ffj;
      a.nff = ffj.C;
      for (int q = 0; q < 4; ++q) a.ff_t[q] = (signed char)fft[q];
    }
    if (blocks == 0 && a.nstats == 0 && a.nff == 0) continue;
    a.copy = SmallCopy{};
    if (off + cnt >= ntensors) (void)small_copy_take(s, &a.copy);   // (last chunk of the call: after every gradient was consumed)
    if (nt == 1024) multi_opt_kernel<P, OPT, 1024><<<(int)blocks + a.nstats + a.nff, 1024, 0, s>>>(a, h);
    else multi_opt_kernel<P, OPT, 256><<<(int)blocks + a.nstats + a.nff, 256, 0, s>>>(a, h);
    EMB_CHECK_LAUNCH();
  }
  return EMB_OK;
}
}  // namespace emb

extern "C" int emb_conv_pack_register(const void* W, void* wpack, void* wflip, int Cout, int Cin, int cin_pad, int k, int dtype) {
  EMB_CHECK_ARG(W && wpack && Cout > 0 && Cin > 0 && cin_pad >= Cin && k > 0, "emb_conv_pack_register: bad argument");
  EMB_CHECK_ARG(dtype == EMB_BF16 || dtype == EMB_F32, "emb_conv_pack_register: the images are bf16 or f32");
  std::lock_guard<std::mutex> lk(emb::pack_mutex());
  emb::pack_table()[W] = emb::PackDesc{wpack, wflip, Cout, Cin, cin_pad, k, dtype == EMB_F32};
  return EMB_OK;
}

extern "C" int emb_conv_pack_unregister(const void* W) {
  std::lock_guard<std::mutex> lk(emb::pack_mutex());
  emb::pack_table().erase(W);
  return EMB_OK;
}

extern "C" int emb_adam_step_multi(void* const* params, const void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                                   void* const* bf16_shadows, const int64_t* sizes, int ntensors, double lr, double beta1,
                                   double beta2, double eps, double weight_decay, uint64_t step_val, const uint64_t* step_dev,
                                   int dtype, emb_stream_t stream) {
  EMB_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && sizes && ntensors >= 0, "emb_adam_step_multi: bad argument");
  const Hyper h{lr, beta1, beta2, eps, weight_decay, 0.0, 0.0, step_val, step_dev, nullptr};
  if (dtype == EMB_F64) return multi_launch<double, OPT_ADAM>(params, grads, exp_avg, exp_avg_sq, bf16_shadows, sizes, ntensors, h, (hipStream_t)stream);
  if (dtype == EMB_F32 || dtype == EMB_BF16) return multi_launch<float, OPT_ADAM>(params, grads, exp_avg, exp_avg_sq, bf16_shadows, sizes, ntensors, h, (hipStream_t)stream);
  set_error("emb_adam_step_multi: unsupported dtype %d", dtype);
  return EMB_ERR_DTYPE;
}

extern "C" int emb_rmsprop_step_multi(void* const* params, const void* const* grads, void* const* square_avg,
                                      void* const* bf16_shadows, const int64_t* sizes, int ntensors, double lr, double alpha,
                                      double eps, double weight_decay, int dtype, emb_stream_t stream) {
  EMB_CHECK_ARG(params && grads && square_avg && sizes && ntensors >= 0, "emb_rmsprop_step_multi: bad argument");
  const Hyper h{lr, 0.0, 0.0, eps, weight_decay, alpha, 0.0, 0, nullptr, nullptr};
  if (dtype == EMB_F64) return multi_launch<double, OPT_RMSPROP>(params, grads, nullptr, square_avg, bf16_shadows, sizes, ntensors, h, (hipStream_t)stream);
  if (dtype == EMB_F32 || dtype == EMB_BF16) return multi_launch<float, OPT_RMSPROP>(params, grads, nullptr, square_avg, bf16_shadows, sizes, ntensors, h, (hipStream_t)stream);
  set_error("emb_rmsprop_step_multi: unsupported dtype %d", dtype);
  return EMB_ERR_DTYPE;
}

extern "C" int emb_nadam_step_multi(void* const* params, const void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                                    double* m_schedule, void* const* bf16_shadows, const int64_t* sizes, int ntensors, double lr,
                                    double beta1, double beta2, double eps, double weight_decay, double schedule_decay,
                                    uint64_t step_val, const uint64_t* step_dev, int dtype, emb_stream_t stream) {
  EMB_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && m_schedule && sizes && ntensors >= 0, "emb_nadam_step_multi: bad argument");
  EMB_CHECK_ARG(ntensors <= emb::kMaxTensors, "emb_nadam_step_multi: at most %d tensors per call (shared m_schedule)", emb::kMaxTensors);
  const Hyper h{lr, beta1, beta2, eps, weight_decay, 0.0, schedule_decay, step_val, step_dev, m_schedule};
  if (dtype == EMB_F64) return multi_launch<double, OPT_NADAM>(params, grads, exp_avg, exp_avg_sq, bf16_shadows, sizes, ntensors, h, (hipStream_t)stream);
  if (dtype == EMB_F32 || dtype == EMB_BF16) return multi_launch<float, OPT_NADAM>(params, grads, exp_avg, exp_avg_sq, bf16_shadows, sizes, ntensors, h, (hipStream_t)stream);
  set_error("emb_nadam_step_multi: unsupported dtype %d", dtype);
  return EMB_ERR_DTYPE;
}

// ------------------------------------------------------------------------------ library plumbing
#include <cstdarg>
#include <cstdio>
namespace emb {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace emb
extern "C" int emb_abi_version(void) { return EMB_ABI_VERSION; }
extern "C" const char* emb_last_error(void) { return emb::g_err; }
