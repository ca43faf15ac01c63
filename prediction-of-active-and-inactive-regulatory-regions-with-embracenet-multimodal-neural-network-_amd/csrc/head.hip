// Classifier head + loss + head backward in ONE launch.
//   logits = E . W^T + b                              the final nn.Linear(width, 2) of the post stack (EmbraceNetMultimodal.py:151-154,190)
//   loss   = per-batch class-weighted 2-class cross-entropy (utils/utils.py:121-140; training_models_multimodal.py:140-141,151-154)
//   dE     = dlogits . W,  dW = dlogits^T . E,  db = sum dlogits          (autograd through the same lines, :156)
// plus the confusion counts of the step table (utils/utils.py:80-94) and the two step counters, i.e. what
// emb_mlp_fwd(last layer) + emb_weighted_ce + emb_mlp_bwd(last layer) do in three dependent launches (~28 us of a
// 300 us step for a 256 -> 2 layer).  The loss needs nothing from other rows except the class counts, which every
// workgroup recounts from the labels (B int64 values out of L2), so rows are independent:
//   wave = one row at a time: lanes cover the K inputs four at a time, two dot products by butterfly reduction, the
//   softmax / loss terms redundantly in every lane, dE written straight from registers, dW accumulated in registers over
//   the wave's rows; the four waves meet in LDS and the workgroup writes one slab row [2][K+1] (bias column last) and one
//   row of loss / count partials.  Slabs are summed in fixed order by reduce.hip (deterministic, deferrable).
// Arithmetic in fp32 on fp32 master weights; logits are rounded to the activation type T before the loss so that the loss
// belongs to the logits the caller sees.
#include "reduce.h"
#include "conv_tiles.h"
#include "first_gram.h"

namespace emb {

constexpr int kHeadRows = 8;        // rows per workgroup (2 per wave)
constexpr int kHeadStats = 8;       // floats per workgroup in the statistics slab: loss share, tp, pp, positives, rows

struct HeadArgs {
  const void* E;
  const float* W;
  const float* bias;
  const int64_t* target;
  int64_t* class_counts;
  int global_counts;
  void* logits;
  void* dE;
  // the fusion layer's code bytes [B][K] (EMB_CODE_KEEP0 / KEEP1) and the two PRE-MASKED gradients dD_m = dE * keep_m the head
  // writes for emb_embrace_bwd_masked (csrc/gemm_jobs.h) when the head sits directly on the fusion layer (all three or none)
  const uint8_t* code;
  void* dD0;
  void* dD1;
  float* slab;        // [nblk][2][K+1]
  float* stats;       // [nblk][kHeadStats]
  uint64_t* tick_a;
  uint64_t* tick_b;
  int B, K;
  // parked totals jobs of the first conv block's lag statistics (first_fin.h): workgroups nblk .. nblk + njobs - 1 of the launch
  GramJobsArgs jobs;
  int nblk, njobs;
};

template <typename T> __device__ __forceinline__ void load4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
  const f32x4 t = *reinterpret_cast<const f32x4*>(p);
  v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <> __device__ __forceinline__ void load4<__bf16>(const __bf16* p, float (&v)[4]) {
  typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
  const bf4 t = *reinterpret_cast<const bf4*>(p);
  v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) {
  f32x4 t = {v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p) = t;
}
template <> __device__ __forceinline__ void store4<__bf16>(__bf16* p, const float (&v)[4]) {
  typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
  bf4 t = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  *reinterpret_cast<bf4*>(p) = t;
}

__device__ __forceinline__ float wave_sum(float v) {   // every lane ends with the same fixed-order sum
  v = row16_sum<float>(v);            // within the rows of 16 lanes by DPP (no LDS-crossbar round trips), then across the four rows
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// KS = ceil(K / 256): lane l owns inputs step*256 + 4*l .. +3
template <typename T, int KS> __global__ __launch_bounds__(256) void head_ce_kernel(const HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [4 waves][2][KS*256] weight-gradient partials
  __shared__ long long scount[256];
  __shared__ float sstat[4][8];
  if ((int)blockIdx.x >= a.nblk) {   // a carried job: this launch leaves most CUs idle
    GramPre pre;
    pre.have = false;
    gram_job<256>((int)blockIdx.x - a.nblk, pre, a.jobs.edge, a.jobs.B, a.jobs.L, a.jobs.part, a.jobs.rows, a.jobs.parts, a.jobs.tot,
                  reinterpret_cast<float*>(scount));
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = a.B, K = a.K;

  // class counts of the whole batch (utils/utils.py:121-133): recounted per workgroup, fixed-order tree
  long long pl = 0;
  for (int i = tid; i < B; i += 256) pl += (a.target[i] == 1);
  {   // integer sums: any order gives the same count; one barrier instead of a nine-barrier tree
    int lo = (int)(pl & 0x7fffffff);             // (a thread sees at most B / 256 + 1 rows)
    lo = row16_sum<int>(lo);
    lo += __shfl_xor(lo, 16, 64);
    lo += __shfl_xor(lo, 32, 64);
    if (lane == 0) scount[wave] = lo;
  }
  __syncthreads();
  long long pos = (scount[0] + scount[1]) + (scount[2] + scount[3]), n = B;
  if (a.global_counts == 2) {
    // data-parallel exchange block of FOUR floats: [0..1] = (positives, rows) of the GLOBAL batch, reduced one step ahead inside
    // the gradient all-reduce; [2..3] receive this shard's own counts for the next reduction (bench.py: the slots ride in the
    // late gradient bucket) -- no count kernel, no conversion launches around the collective
    float* cf = reinterpret_cast<float*>(a.class_counts);
    const long long lpos = pos;
    pos = (long long)llrintf(cf[0]);
    n = (long long)llrintf(cf[1]);
    if (blockIdx.x == 0 && tid == 0) {
      cf[2] = (float)lpos;
      cf[3] = (float)B;
    }
  } else if (a.global_counts) {
    pos = a.class_counts[0];
    n = a.class_counts[1];
  } else if (blockIdx.x == 0 && tid == 0) {
    a.class_counts[0] = pos;
    a.class_counts[1] = n;
  }
  const long long neg = n - pos;
  const double pos_inv = pos != 0 ? 1.0 / (double)pos : 0.0;
  const double neg_inv = neg != 0 ? 1.0 / (double)neg : 0.0;
  const float w1 = (float)(pos_inv / (neg_inv + pos_inv));
  const float w0 = (float)(neg_inv / (neg_inv + pos_inv));
  const double den = (double)w1 * (double)pos + (double)w0 * (double)neg;   // sum_i w[y_i] over the global batch
  const float g_of[2] = {(float)((double)w0 / den), (float)((double)w1 / den)};   // d loss / d (row's CE term): two values, not one f64 division per row

  float wa[KS][4], wb[KS][4];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int j = s * 256 + 4 * lane;
    if (j < K) {
      load4<float>(a.W + j, wa[s]);
      load4<float>(a.W + K + j, wb[s]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) wa[s][e] = wb[s][e] = 0.f;
    }
  }
  const float b0 = a.bias[0], b1 = a.bias[1];

  float ga[KS][4], gb[KS][4];   // dW rows 0 / 1 accumulated over this wave's rows
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int e = 0; e < 4; ++e) ga[s][e] = gb[s][e] = 0.f;
  float db0 = 0.f, db1 = 0.f, tp = 0.f, pp = 0.f, np_ = 0.f, rows = 0.f;
  double num = 0.0;

  const T* E = (const T*)a.E;
  T* dE = (T*)a.dE;
  T* logits = (T*)a.logits;
  const bool train = a.dE != nullptr || a.code != nullptr;   // the head's own backward is wanted
  constexpr int RPW = kHeadRows / 4;
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const long row = (long)blockIdx.x * kHeadRows + wave * RPW + r;
    if (row >= B) break;   // wave-uniform
    float e[KS][4];
    float d0 = 0.f, d1 = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int j = s * 256 + 4 * lane;
      if (j < K) {
        load4<T>(E + row * K + j, e[s]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) e[s][q] = 0.f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        d0 += e[s][q] * wa[s][q];
        d1 += e[s][q] * wb[s][q];
      }
    }
    const float z0 = (float)(T)(wave_sum(d0) + b0), z1 = (float)(T)(wave_sum(d1) + b1);   // the logits the caller sees
    if (lane == 0) {
      logits[2 * row] = (T)z0;
      logits[2 * row + 1] = (T)z1;
    }
    const int y = a.target[row] == 1 ? 1 : 0;
    const float zm = fmaxf(z0, z1);
    const float e0 = expf(z0 - zm), e1 = expf(z1 - zm);
    const float lse = zm + logf(e0 + e1);
    const float wy = y ? w1 : w0;
    num += (double)wy * (double)(lse - (y ? z1 : z0));
    const int pred = z1 > z0 ? 1 : 0;   // torch.argmax: first maximum wins ties
    tp += (float)(pred & y);
    pp += (float)pred;
    np_ += (float)y;
    rows += 1.f;
    if (train) {
      const float inv = 1.0f / (e0 + e1);
      const float g = y ? g_of[1] : g_of[0];
      const float l0 = g * (e0 * inv - (y ? 0.0f : 1.0f)), l1 = g * (e1 * inv - (y ? 1.0f : 0.0f));
      db0 += l0;
      db1 += l1;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int j = s * 256 + 4 * lane;
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          o[q] = l0 * wa[s][q] + l1 * wb[s][q];
          ga[s][q] += l0 * e[s][q];
          gb[s][q] += l1 * e[s][q];
        }
        if (j < K) {
          if (a.code != nullptr) {              // one dword = the code bytes of this lane's four elements
            const uint32_t cw = *reinterpret_cast<const uint32_t*>(a.code + row * K + j);
            float o0[4], o1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const uint32_t cb = cw >> (8 * q);
              o0[q] = (cb & EMB_CODE_KEEP0) ? o[q] : 0.0f;
              o1[q] = (cb & EMB_CODE_KEEP1) ? o[q] : 0.0f;
            }
            store4<T>((T*)a.dD0 + row * K + j, o0);
            store4<T>((T*)a.dD1 + row * K + j, o1);
          }
          if (dE != nullptr) store4<T>(dE + row * K + j, o);
        }
      }
    }
  }

  // the four waves meet in LDS (wave order fixed); one slab row and one statistics row per workgroup
  constexpr int KP = KS * 256;
  if (train) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      f32x4 va = {ga[s][0], ga[s][1], ga[s][2], ga[s][3]}, vb = {gb[s][0], gb[s][1], gb[s][2], gb[s][3]};
      *reinterpret_cast<f32x4*>(red + (wave * 2 + 0) * KP + s * 256 + 4 * lane) = va;
      *reinterpret_cast<f32x4*>(red + (wave * 2 + 1) * KP + s * 256 + 4 * lane) = vb;
    }
  }
  if (lane == 0) {
    sstat[wave][0] = (float)(num / den);
    sstat[wave][1] = tp;
    sstat[wave][2] = pp;
    sstat[wave][3] = np_;
    sstat[wave][4] = rows;
    sstat[wave][5] = db0;
    sstat[wave][6] = db1;
  }
  __syncthreads();
  if (train) {
    float* out = a.slab + (long)blockIdx.x * 2 * (K + 1);
    for (int q = tid; q < 2 * K; q += 256) {
      const int m = q >= K ? 1 : 0, j = q - m * K;
      out[m * (K + 1) + j] = ((red[(0 * 2 + m) * KP + j] + red[(1 * 2 + m) * KP + j]) + red[(2 * 2 + m) * KP + j]) + red[(3 * 2 + m) * KP + j];
    }
    if (tid < 2) out[tid * (K + 1) + K] = ((sstat[0][5 + tid] + sstat[1][5 + tid]) + sstat[2][5 + tid]) + sstat[3][5 + tid];
  }
  if (tid < kHeadStats)
    a.stats[(long)blockIdx.x * kHeadStats + tid] = tid < 5 ? ((sstat[0][tid] + sstat[1][tid]) + sstat[2][tid]) + sstat[3][tid] : 0.f;
  if (blockIdx.x == 0 && tid == 0) {   // counters that advance once per loss evaluation (see weighted_ce_kernel)
    if (a.tick_a != nullptr) a.tick_a[0] += 1;
    if (a.tick_b != nullptr) a.tick_b[0] += 1;
  }
}

static int head_blocks(int B) { return cdiv(B, kHeadRows); }

}  // namespace emb

using namespace emb;

extern "C" int emb_head_ce_supported(int B, int K, int dtype) {
  return (dtype == EMB_F32 || dtype == EMB_BF16) && B > 0 && B < (1 << 24) && K >= 4 && K <= 1024 && K % 4 == 0;
}

extern "C" int64_t emb_head_ce_workspace_bytes(int B, int K) {
  if (B <= 0 || K <= 0) return -1;
  return (int64_t)head_blocks(B) * (2 * (K + 1) + kHeadStats) * (int64_t)sizeof(float);
}

extern "C" int emb_head_ce_masked(const void* E, const void* W, const void* bias, const int64_t* target, int64_t* class_counts,
                                  int global_counts, void* logits, void* dE, const uint8_t* code, void* dD0, void* dD1,
                                  void* workspace, int64_t workspace_bytes, uint64_t* tick_a, uint64_t* tick_b, int B, int K,
                                  int dtype, emb_stream_t stream) {
  EMB_CHECK_ARG(E && W && bias && target && class_counts && logits && workspace, "emb_head_ce: null pointer");
  EMB_CHECK_ARG(emb_head_ce_supported(B, K, dtype), "emb_head_ce: unsupported shape / dtype (see emb_head_ce_supported)");
  EMB_CHECK_ARG(workspace_bytes >= emb_head_ce_workspace_bytes(B, K), "emb_head_ce: workspace too small");
  EMB_CHECK_ARG(aligned16(E) && aligned16(W) && (dE == nullptr || aligned16(dE)), "emb_head_ce: E, W and dE must be 16-byte aligned");
  EMB_CHECK_ARG((code == nullptr) == (dD0 == nullptr) && (code == nullptr) == (dD1 == nullptr),
                "emb_head_ce_masked: code, dD0 and dD1 go together");
  EMB_CHECK_ARG(code == nullptr || (aligned16(dD0) && aligned16(dD1) && (reinterpret_cast<uintptr_t>(code) & 3u) == 0),
                "emb_head_ce_masked: dD0 / dD1 must be 16-byte, code 4-byte aligned");
  const int nblk = head_blocks(B), KS = cdiv(K, 256);
  HeadArgs a{};
  a.E = E; a.W = (const float*)W; a.bias = (const float*)bias; a.target = target; a.class_counts = class_counts;
  a.global_counts = global_counts; a.logits = logits; a.dE = dE;
  a.code = code; a.dD0 = dD0; a.dD1 = dD1;
  a.slab = (float*)workspace;
  a.stats = a.slab + (long)nblk * 2 * (K + 1);
  a.tick_a = tick_a; a.tick_b = tick_b; a.B = B; a.K = K;
  a.nblk = nblk;
  a.njobs = gram_jobs_take((hipStream_t)stream, &a.jobs) ? gram_jobs_count() : 0;
  const size_t lds = (size_t)4 * 2 * KS * 256 * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
#define EMB_HEAD_LAUNCH(TT, KK) head_ce_kernel<TT, KK><<<nblk + a.njobs, 256, lds, s>>>(a)
  if (dtype == EMB_BF16) {
    switch (KS) {
      case 1: EMB_HEAD_LAUNCH(__bf16, 1); break;
      case 2: EMB_HEAD_LAUNCH(__bf16, 2); break;
      case 3: EMB_HEAD_LAUNCH(__bf16, 3); break;
      default: EMB_HEAD_LAUNCH(__bf16, 4); break;
    }
  } else {
    switch (KS) {
      case 1: EMB_HEAD_LAUNCH(float, 1); break;
      case 2: EMB_HEAD_LAUNCH(float, 2); break;
      case 3: EMB_HEAD_LAUNCH(float, 3); break;
      default: EMB_HEAD_LAUNCH(float, 4); break;
    }
  }
#undef EMB_HEAD_LAUNCH
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

extern "C" int emb_head_ce(const void* E, const void* W, const void* bias, const int64_t* target, int64_t* class_counts,
                           int global_counts, void* logits, void* dE, void* workspace, int64_t workspace_bytes, uint64_t* tick_a,
                           uint64_t* tick_b, int B, int K, int dtype, emb_stream_t stream) {
  return emb_head_ce_masked(E, W, bias, target, class_counts, global_counts, logits, dE, nullptr, nullptr, nullptr, workspace,
                            workspace_bytes, tick_a, tick_b, B, K, dtype, stream);
}

extern "C" int emb_head_ce_finish(const void* workspace, void* dW, void* db, float* loss, int64_t* confusion, int B, int K,
                                  emb_stream_t stream) {
  EMB_CHECK_ARG(workspace && loss && (dW == nullptr) == (db == nullptr), "emb_head_ce_finish: null pointer");
  EMB_CHECK_ARG(B > 0 && K > 0, "emb_head_ce_finish: bad dims");
  const int nblk = head_blocks(B);
  const float* slab = (const float*)workspace;
  hipStream_t s = (hipStream_t)stream;
  if (dW != nullptr) {
    ReduceJob j{};
    j.in = slab; j.out[0] = dW; j.out[1] = db; j.per = 2L * (K + 1); j.S = nblk; j.kind = RJ_LINEAR; j.iv[0] = K;
    const int rc = reduce_submit(j, false, s);
    if (rc != EMB_OK) return rc;
  }
  ReduceJob j{};
  j.in = slab + (long)nblk * 2 * (K + 1); j.out[0] = loss; j.out[1] = confusion; j.per = kHeadStats; j.S = nblk;
  j.kind = RJ_HEAD_STATS;
  return reduce_submit(j, false, s);
}
