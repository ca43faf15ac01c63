// One kernel for all slab reductions (see reduce.h).
#include <vector>

#include "reduce.h"
#include "first_fin.h"
#include "rider.h"

namespace emb {

constexpr int kMaxJobs = 8;
struct DevJob {
  ReduceJob j;
  int lanes, blk_end;   // slice lanes per element (power of two <= 16); exclusive prefix end of this job's blocks
  int vec;              // elements per thread: 4 (16-byte loads; per % 4 == 0, aligned) or 1
};
struct ReduceTable {
  DevJob d[kMaxJobs];
  int n;
};

template <typename P> __device__ __forceinline__ void reduce_write(const ReduceJob& j, long q, P t) {
  if (j.kind == RJ_LINEAR) {
    const int N = j.iv[0], pitch = j.iv[1] > 0 ? j.iv[1] : N + 1;   // slab row = [N values | bias | padding up to pitch]
    const long m = q / pitch;
    const int n = (int)(q - m * pitch);
    if (n == N) { if (j.out[1]) ((P*)j.out[1])[m] = t; }
    else if (n < N && j.out[0]) ((P*)j.out[0])[m * N + n] = t;
  } else if (j.kind == RJ_CONV) {   // slab row = [k*cin_pad tap-major columns | bias]; dW in torch layout [Cout][Cin][k]
    const int Cin = j.iv[0], cin_pad = j.iv[1], k = j.iv[2], KK = k * cin_pad;
    const int o = (int)(q / (KK + 1)), col = (int)(q - (long)o * (KK + 1));
    const int tap = col / cin_pad, ci = col - tap * cin_pad;
    if (col == KK) { if (j.out[1]) ((P*)j.out[1])[o] = t; }
    else if (ci < Cin && j.out[0]) ((P*)j.out[0])[((long)o * Cin + ci) * k + tap] = t;
  } else if (j.kind == RJ_HEAD_STATS) {   // head.hip: [loss share, tp, pp, positives, rows, -, -, -]; counts are small exact integers
    if (q == 0) ((float*)j.out[0])[0] = (float)t;
    else if (q <= 4 && j.out[1] != nullptr) ((long long*)j.out[1])[q - 1] = (long long)(t + (P)0.5);
  } else {                          // RJ_MLP: per layer [N*K weights | N biases]
    long off = q;
    const int L = j.iv[0];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      if (l < L) {
        const long nw = (long)j.iv[1 + l] * j.iv[5 + l];
        if (off < nw) { if (j.out[l]) ((P*)j.out[l])[off] = t; return; }
        off -= nw;
        if (off < j.iv[1 + l]) { if (j.out[4 + l]) ((P*)j.out[4 + l])[off] = t; return; }
        off -= j.iv[1 + l];
      }
    }
  }
}

// block = (256 / lanes) elements x `lanes` slice lanes: lane sl sums slices sl, sl + lanes, ...; the lane sums meet in lane order
template <typename P> __global__ __launch_bounds__(256) void multi_reduce_kernel(const ReduceTable tab) {
  __shared__ ReduceTable t;
  __shared__ P red[4][256];
  {
    const unsigned* src = reinterpret_cast<const unsigned*>(&tab);
    unsigned* dst = reinterpret_cast<unsigned*>(&t);
    for (int i = threadIdx.x; i < (int)(sizeof(ReduceTable) / 4); i += 256) dst[i] = src[i];
  }
  __syncthreads();
  int ji = 0;
  while (ji < t.n - 1 && (int)blockIdx.x >= t.d[ji].blk_end) ++ji;
  const DevJob& d = t.d[ji];
  const int bid = (int)blockIdx.x - (ji == 0 ? 0 : t.d[ji - 1].blk_end);
  const int lanes = d.lanes, qpb = 256 / lanes, qi = threadIdx.x % qpb, sl = threadIdx.x / qpb;
  const long per = d.j.per;
  const P* in = (const P*)d.j.in;
  if (d.vec == 4) {   // four consecutive elements per thread, 16-byte (P = float) loads
    typedef P P4 __attribute__((ext_vector_type(4)));
    const long q = ((long)bid * qpb + qi) * 4;
    P4 a = {0, 0, 0, 0};
    if (q < per) {
#pragma unroll 8
      for (int s = sl; s < d.j.S; s += lanes) a += *reinterpret_cast<const P4*>(in + (long)s * per + q);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[e][sl * qpb + qi] = a[e];
    __syncthreads();
    if (sl == 0 && q < per) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        P sum = 0;
        for (int i = 0; i < lanes; ++i) sum += red[e][i * qpb + qi];
        reduce_write<P>(d.j, q + e, sum);
      }
    }
    return;
  }
  const long q = (long)bid * qpb + qi;
  P a = 0;
  if (q < per) {
#pragma unroll 8
    for (int s = sl; s < d.j.S; s += lanes) a += in[(long)s * per + q];
  }
  red[0][sl * qpb + qi] = a;
  __syncthreads();
  if (sl == 0 && q < per) {
    P sum = 0;
    for (int i = 0; i < lanes; ++i) sum += red[0][i * qpb + qi];
    reduce_write<P>(d.j, q, sum);
  }
}

struct Pending {
  std::vector<ReduceJob> f32, f64;
};
static Pending& pending() {
  static Pending p;
  return p;
}
static bool g_defer = false;

template <typename P> static int launch_jobs(const ReduceJob* jobs, int n, hipStream_t s) {
  for (int off = 0; off < n; off += kMaxJobs) {
    ReduceTable t{};
    const int cnt = n - off < kMaxJobs ? n - off : kMaxJobs;
    int blocks = 0;
    for (int i = 0; i < cnt; ++i) {
      t.d[i].j = jobs[off + i];
      int lanes = 1;
      while (lanes < 16 && lanes < t.d[i].j.S) lanes *= 2;
      t.d[i].lanes = lanes;
      t.d[i].vec = (t.d[i].j.per % 4 == 0 && aligned16(t.d[i].j.in)) ? 4 : 1;
      const long epb = (long)(256 / lanes) * t.d[i].vec;   // elements per block
      blocks += (int)((t.d[i].j.per + epb - 1) / epb);
      t.d[i].blk_end = blocks;
    }
    t.n = cnt;
    if (blocks == 0) continue;
    multi_reduce_kernel<P><<<blocks, 256, 0, s>>>(t);
    EMB_CHECK_LAUNCH();
  }
  return EMB_OK;
}

static int job_outputs(const ReduceJob& j) { return j.kind == RJ_MLP ? 8 : 2; }

bool reduce_claim(const void* grad, bool is_double, ReduceClaim* out) {
  if (grad == nullptr) return false;
  std::vector<ReduceJob>& v = is_double ? pending().f64 : pending().f32;
  for (size_t i = 0; i < v.size(); ++i) {
    if (v[i].kind == RJ_HEAD_STATS) continue;
    const int no = job_outputs(v[i]);
    for (int k = 0; k < no; ++k) {
      if (v[i].out[k] != grad) continue;
      out->job = v[i];
      out->which = k;
      v[i].out[k] = nullptr;
      bool any = false;
      for (int t = 0; t < no; ++t) any = any || v[i].out[t] != nullptr;
      if (!any) v.erase(v.begin() + (long)i);
      return true;
    }
  }
  return false;
}

bool reduce_claim_stats(bool is_double, ReduceJob* out) {
  std::vector<ReduceJob>& v = is_double ? pending().f64 : pending().f32;
  for (size_t i = 0; i < v.size(); ++i)
    if (v[i].kind == RJ_HEAD_STATS) {
      *out = v[i];
      v.erase(v.begin() + (long)i);
      return true;
    }
  return false;
}

int launch_jobs_f32(const ReduceJob* jobs, int n, hipStream_t s) { return launch_jobs<float>(jobs, n, s); }
int launch_jobs_f64(const ReduceJob* jobs, int n, hipStream_t s) { return launch_jobs<double>(jobs, n, s); }

bool reduce_deferring() { return g_defer; }

// ---- the parked finish of the first conv block's recompute-free backward (first_fin.h)
static FirstFinArgs g_fin;
static bool g_fin_valid = false;
int first_fin_flush(hipStream_t s) {
  if (!g_fin_valid) return EMB_OK;
  g_fin_valid = false;
  return first_fin_launch(g_fin, s);
}
int first_fin_submit(const FirstFinArgs& f, hipStream_t s) {
  if (!g_defer) return first_fin_launch(f, s);
  const int rc = first_fin_flush(s);   // (one slot)
  if (rc != EMB_OK) return rc;
  g_fin = f;
  g_fin_valid = true;
  return EMB_OK;
}
bool first_fin_peek(FirstFinArgs* out) {
  if (g_fin_valid) *out = g_fin;
  return g_fin_valid;
}
void first_fin_drop() { g_fin_valid = false; }

// ---- the parked totals jobs of the first conv block's lag statistics (first_fin.h)
static GramJobsArgs g_jobs;
static bool g_jobs_valid = false;
int gram_jobs_flush(hipStream_t s) {
  if (!g_jobs_valid) return EMB_OK;
  g_jobs_valid = false;
  return gram_jobs_launch(g_jobs, s);
}
void gram_jobs_park(const GramJobsArgs& a, hipStream_t s) {
  (void)gram_jobs_flush(s);
  g_jobs = a;
  g_jobs_valid = true;
}
bool gram_jobs_take(GramJobsArgs* out) {
  if (!g_jobs_valid) return false;
  *out = g_jobs;
  g_jobs_valid = false;
  return true;
}

int reduce_submit(const ReduceJob& job, bool is_double, hipStream_t s) {
  if (job.per <= 0 || job.S <= 0) return EMB_OK;
  if (g_defer) {
    std::vector<ReduceJob>& v = is_double ? pending().f64 : pending().f32;
    v.push_back(job);
    return EMB_OK;
  }
  return is_double ? launch_jobs<double>(&job, 1, s) : launch_jobs<float>(&job, 1, s);
}

}  // namespace emb

extern "C" int emb_reduce_defer(int on) {
  emb::g_defer = on != 0;
  return EMB_OK;
}

extern "C" int emb_reduce_flush(emb_stream_t stream) {
  emb::Pending& p = emb::pending();
  int rc = emb::rider_flush();   // a parked launch may be the producer of a queued slab
  if (rc != EMB_OK) return rc;
  rc = emb::gram_jobs_flush((hipStream_t)stream);
  if (rc != EMB_OK) return rc;
  rc = emb::first_fin_flush((hipStream_t)stream);
  if (rc != EMB_OK) return rc;
  if (!p.f32.empty()) rc = emb::launch_jobs<float>(p.f32.data(), (int)p.f32.size(), (hipStream_t)stream);
  if (rc == EMB_OK && !p.f64.empty()) rc = emb::launch_jobs<double>(p.f64.data(), (int)p.f64.size(), (hipStream_t)stream);
  p.f32.clear();
  p.f64.clear();
  return rc;
}
