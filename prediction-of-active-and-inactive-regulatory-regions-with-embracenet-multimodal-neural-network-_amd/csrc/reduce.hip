// One kernel for all slab reductions (see reduce.h).
#include <map>
#include <mutex>
#include <vector>

#include "reduce.h"
#include "first_fin.h"
#include "first_gram.h"
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
    const unsigned qu = (unsigned)q, mu = qu / (unsigned)pitch;     // (slabs have fewer than 2^31 elements: reduce_submit)
    const long m = (long)mu;
    const int n = (int)(qu - mu * (unsigned)pitch);
    if (n == N) { if (j.out[1]) ((P*)j.out[1])[m] = t; }
    else if (n < N && j.out[0]) ((P*)j.out[0])[m * N + n] = t;
  } else if (j.kind == RJ_CONV) {   // slab row = [k*cin_pad tap-major columns | bias]; dW in torch layout [Cout][Cin][k]
    const int Cin = j.iv[0], cin_pad = j.iv[1], k = j.iv[2], KK = k * cin_pad;
    const int o = (int)((unsigned)q / (unsigned)(KK + 1)), col = (int)((unsigned)q - (unsigned)o * (unsigned)(KK + 1));
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
// nfin > 0: the first `nfin` workgroups run the parked per-channel finish of the first conv block's recompute-free backward
// (first_gram.h) -- in a data-parallel step the gradients must be complete before the all-reduce, so the finish cannot wait
// for the optimizer launch, but it need not be a launch of its own either (9.4 us): it rides here
struct FinStoreR {
  const FirstFinArgs& a;
  int c;
  __device__ __forceinline__ void scalars(float dgamma, float dbeta, float dbias) const {
    a.dgamma[c] = dgamma; a.dbeta[c] = dbeta; a.dbias[c] = dbias;
  }
  __device__ __forceinline__ void dw(long idx, float v) const { a.dW[idx] = v; }
};
// NT threads per workgroup: 256, or 1024 when the finish rides in a small launch (its workgroups are the critical path and then run
// the 1024-thread body of the standalone kernel; multi_opt_kernel, loss_optim.hip, does the same).
template <typename P, int NT> __global__ __launch_bounds__(NT) void multi_reduce_kernel(const ReduceTable tab, const FirstFinArgs fin, const int nfin) {
  if constexpr (sizeof(P) == 4) {
    if ((int)blockIdx.x < nfin) {
      __shared__ __attribute__((aligned(16))) float fin_lds[first_finish_lds_floats<NT>()];
      first_finish_body<NT>(fin, (int)blockIdx.x, fin_lds, FinStoreR{fin, (int)blockIdx.x});
      return;
    }
  }
  __shared__ ReduceTable t;
  __shared__ P red[4][NT];
  {
    const unsigned* src = reinterpret_cast<const unsigned*>(&tab);
    unsigned* dst = reinterpret_cast<unsigned*>(&t);
    for (int i = threadIdx.x; i < (int)(sizeof(ReduceTable) / 4); i += NT) dst[i] = src[i];
  }
  __syncthreads();
  int ji = 0;
  const int blk = (int)blockIdx.x - nfin;
  while (ji < t.n - 1 && blk >= t.d[ji].blk_end) ++ji;
  const DevJob& d = t.d[ji];
  const int bid = blk - (ji == 0 ? 0 : t.d[ji - 1].blk_end);
  const int lanes = d.lanes, qpb = NT / lanes, qi = threadIdx.x % qpb, sl = threadIdx.x / qpb;
  const long per = d.j.per;
  const P* in = (const P*)d.j.in;
  if (d.vec == 4 && lanes == 1) {
    // large row-major slabs (the docking weight gradients): every thread sums all slices of ITS four elements (independent 16-byte
    // loads, slice order) and stores them as one vector when they are four weights of one row -- no exchange through LDS, no
    // 4-byte stores at a 16-byte stride
    typedef P P4 __attribute__((ext_vector_type(4)));
    const long q = ((long)bid * qpb + qi) * 4;
    if (q >= per) return;
    P4 a = {0, 0, 0, 0};
#pragma unroll 8
    for (int s = 0; s < d.j.S; ++s) a += *reinterpret_cast<const P4*>(in + (long)s * per + q);
    if (d.j.kind == RJ_LINEAR && d.j.out[0] != nullptr) {
      const int N = d.j.iv[0], pitch = d.j.iv[1] > 0 ? d.j.iv[1] : N + 1;
      const unsigned mu = (unsigned)q / (unsigned)pitch;
      const long m = (long)mu;
      const int n = (int)((unsigned)q - mu * (unsigned)pitch);
      P* dst = (P*)d.j.out[0] + m * N + n;
      if (n + 3 < N && (reinterpret_cast<uintptr_t>(dst) & (sizeof(P4) - 1)) == 0) {
        *reinterpret_cast<P4*>(dst) = a;
        return;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) reduce_write<P>(d.j, q + e, a[e]);
    return;
  }
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

// ---- everything a stream has parked (reduce.h, first_fin.h, rider.h): process state keyed by the stream, guarded by one mutex.
// Two trainers in one process (each on its own stream, from any threads -- autograd runs backward nodes on its own device
// thread) never see each other's jobs; the same stream used from several threads is the caller's ordering problem, as with any
// enqueue.  Nothing here owns device memory: a lot only holds launch descriptors (raw pointers the caller keeps alive).
struct Lot {
  bool reduce_defer = false, rider_defer = false;
  std::vector<ReduceJob> f32, f64;
  FirstFinArgs fin{};
  bool fin_valid = false;
  GramJobsArgs jobs{};
  bool jobs_valid = false;
  Rider rider{};
  SmallCopy copy{};            // a few floats the next optimizer launch copies (emb_copy_park)
  bool idle() const { return f32.empty() && f64.empty() && !fin_valid && !jobs_valid && rider.kind == RIDER_NONE && copy.n == 0; }
  int parked() const {
    return (int)f32.size() + (int)f64.size() + (fin_valid ? 1 : 0) + (jobs_valid ? 1 : 0) + (rider.kind != RIDER_NONE ? 1 : 0) + (copy.n ? 1 : 0);
  }
};
static std::recursive_mutex& lot_mutex() {
  static std::recursive_mutex m;
  return m;
}
static std::map<hipStream_t, Lot>& lots() {
  static std::map<hipStream_t, Lot> m;
  return m;
}
#define EMB_LOT(var, stream) std::lock_guard<std::recursive_mutex> lot_guard__(emb::lot_mutex()); emb::Lot& var = emb::lots()[(stream)]

template <typename P> static int launch_jobs(const ReduceJob* jobs, int n, hipStream_t s, const FirstFinArgs* fin = nullptr) {
  for (int off = 0; off < n; off += kMaxJobs) {
    ReduceTable t{};
    const int cnt = n - off < kMaxJobs ? n - off : kMaxJobs;
    for (int i = 0; i < cnt; ++i) {
      t.d[i].j = jobs[off + i];
      int lanes = 1;
      while (lanes < 16 && lanes < t.d[i].j.S) lanes *= 2;
      t.d[i].vec = (t.d[i].j.per % 4 == 0 && aligned16(t.d[i].j.in) && (sizeof(P) == 4 || (reinterpret_cast<uintptr_t>(t.d[i].j.in) & 31) == 0)) ? 4 : 1;
      if (t.d[i].vec == 4 && t.d[i].j.per >= 32768 && (t.d[i].j.kind == RJ_LINEAR || t.d[i].j.kind == RJ_MLP)) lanes = 1;   // (the wide path)
      t.d[i].lanes = lanes;
    }
    t.n = cnt;
    const int nfin = (fin != nullptr && off == 0) ? fin->C : 0;   // (rides on the first launch)
    auto count = [&](int nt) {
      int b = 0;
      for (int i = 0; i < cnt; ++i) {
        const long epb = (long)(nt / t.d[i].lanes) * t.d[i].vec;   // elements per block
        b += (int)((t.d[i].j.per + epb - 1) / epb);
        t.d[i].blk_end = b;
      }
      return b;
    };
    int blocks = count(256);
    const bool big = nfin > 0 && sizeof(P) == 4 && blocks <= 2048;
    if (big) blocks = count(1024);
    if (blocks + nfin == 0) continue;
    if (big) multi_reduce_kernel<P, 1024><<<blocks + nfin, 1024, 0, s>>>(t, *fin, nfin);
    else multi_reduce_kernel<P, 256><<<blocks + nfin, 256, 0, s>>>(t, nfin ? *fin : FirstFinArgs{}, nfin);
    EMB_CHECK_LAUNCH();
  }
  return EMB_OK;
}

static int job_outputs(const ReduceJob& j) { return j.kind == RJ_MLP ? 8 : 2; }

bool reduce_claim(hipStream_t st, const void* grad, bool is_double, ReduceClaim* out) {
  if (grad == nullptr) return false;
  EMB_LOT(lot, st);
  std::vector<ReduceJob>& v = is_double ? lot.f64 : lot.f32;
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

bool reduce_claim_stats(hipStream_t st, bool is_double, ReduceJob* out) {
  EMB_LOT(lot, st);
  std::vector<ReduceJob>& v = is_double ? lot.f64 : lot.f32;
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

bool reduce_deferring(hipStream_t s) {
  EMB_LOT(lot, s);
  return lot.reduce_defer;
}

// ---- the parked finish of the first conv block's recompute-free backward (first_fin.h)
int first_fin_flush(hipStream_t s) {
  EMB_LOT(lot, s);
  if (!lot.fin_valid) return EMB_OK;
  lot.fin_valid = false;
  return first_fin_launch(lot.fin, s);
}
int first_fin_submit(const FirstFinArgs& f, hipStream_t s) {
  EMB_LOT(lot, s);
  if (!lot.reduce_defer) return first_fin_launch(f, s);
  const int rc = first_fin_flush(s);   // (one slot per stream)
  if (rc != EMB_OK) return rc;
  lot.fin = f;
  lot.fin_valid = true;
  return EMB_OK;
}
// ---- a parked copy of a few floats (emb_copy_park): taken over by the next optimizer launch of the stream, else run at the flush
__global__ void small_copy_kernel(SmallCopy c) {
  if ((int)threadIdx.x < c.n) c.dst[threadIdx.x] = c.src[threadIdx.x];
}
bool small_copy_take(hipStream_t s, SmallCopy* out) {
  EMB_LOT(lot, s);
  if (lot.copy.n == 0) return false;
  *out = lot.copy;
  lot.copy = SmallCopy{};
  return true;
}
int small_copy_flush(hipStream_t s) {
  SmallCopy c;
  if (!small_copy_take(s, &c)) return EMB_OK;
  small_copy_kernel<<<1, 64, 0, s>>>(c);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}
bool first_fin_peek(hipStream_t s, FirstFinArgs* out) {
  EMB_LOT(lot, s);
  if (lot.fin_valid) *out = lot.fin;
  return lot.fin_valid;
}
void first_fin_drop(hipStream_t s) {
  EMB_LOT(lot, s);
  lot.fin_valid = false;
}

// ---- the parked totals jobs of the first conv block's lag statistics (first_fin.h)
int gram_jobs_flush(hipStream_t s) {
  EMB_LOT(lot, s);
  if (!lot.jobs_valid) return EMB_OK;
  lot.jobs_valid = false;
  return gram_jobs_launch(lot.jobs, s);
}
void gram_jobs_park(const GramJobsArgs& a, hipStream_t s) {
  EMB_LOT(lot, s);
  (void)gram_jobs_flush(s);
  lot.jobs = a;
  lot.jobs_valid = true;
}
bool gram_jobs_take(hipStream_t s, GramJobsArgs* out) {
  EMB_LOT(lot, s);
  if (!lot.jobs_valid) return false;
  *out = lot.jobs;
  lot.jobs_valid = false;
  return true;
}

// ---- the parked rider launch (rider.h)
bool rider_deferring(hipStream_t s) {
  EMB_LOT(lot, s);
  return lot.rider_defer;
}
int rider_flush(hipStream_t s) {
  EMB_LOT(lot, s);
  if (lot.rider.kind == RIDER_NONE) return EMB_OK;
  const Rider r = lot.rider;
  lot.rider.kind = RIDER_NONE;
  return rider_launch(r);
}
void rider_park(const Rider& r) {
  EMB_LOT(lot, r.stream);
  (void)rider_flush(r.stream);
  lot.rider = r;
}
bool rider_take(hipStream_t s, int kind, Rider* out) {
  EMB_LOT(lot, s);
  if (lot.rider.kind != kind) return false;
  *out = lot.rider;
  lot.rider.kind = RIDER_NONE;
  return true;
}

int reduce_submit(const ReduceJob& job, bool is_double, hipStream_t s) {
  if (job.per <= 0 || job.S <= 0) return EMB_OK;
  if (job.per >= (1ll << 31)) { set_error("slab reduction: %lld elements per slice (32-bit element indices)", (long long)job.per); return EMB_ERR_ARG; }
  {
    EMB_LOT(lot, s);
    if (lot.reduce_defer) {
      (is_double ? lot.f64 : lot.f32).push_back(job);
      return EMB_OK;
    }
  }
  return is_double ? launch_jobs<double>(&job, 1, s) : launch_jobs<float>(&job, 1, s);
}

}  // namespace emb

extern "C" int emb_reduce_defer(emb_stream_t stream, int on) {
  EMB_LOT(lot, (hipStream_t)stream);
  lot.reduce_defer = on != 0;
  return EMB_OK;
}

extern "C" int emb_rider_defer(emb_stream_t stream, int on) {
  EMB_LOT(lot, (hipStream_t)stream);
  lot.rider_defer = on != 0;
  return EMB_OK;
}
extern "C" int emb_rider_flush(emb_stream_t stream) { return emb::rider_flush((hipStream_t)stream); }

extern "C" int emb_reduce_flush(emb_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  EMB_LOT(lot, s);
  int rc = emb::rider_flush(s);   // a parked launch may be the producer of a queued slab
  if (rc != EMB_OK) return rc;
  rc = emb::gram_jobs_flush(s);
  if (rc != EMB_OK) return rc;
  // a parked first-block finish rides on the f32 reduction launch when there is one (and it fits one launch's blocks)
  const bool ride = lot.fin_valid && !lot.f32.empty() && lot.fin.C <= 1024;
  if (!ride) {
    rc = emb::first_fin_flush(s);
    if (rc != EMB_OK) return rc;
  }
  if (!lot.f32.empty()) {
    rc = emb::launch_jobs<float>(lot.f32.data(), (int)lot.f32.size(), s, ride ? &lot.fin : nullptr);
    if (ride) lot.fin_valid = false;
  }
  if (rc == EMB_OK && !lot.f64.empty()) rc = emb::launch_jobs<double>(lot.f64.data(), (int)lot.f64.size(), s);
  lot.f32.clear();
  lot.f64.clear();
  return rc;
}

extern "C" int emb_copy_park(emb_stream_t stream, const float* src, float* dst, int n) {
  EMB_CHECK_ARG(src && dst && n >= 1 && n <= 64, "emb_copy_park: 1 .. 64 floats");
  hipStream_t s = (hipStream_t)stream;
  const int rc = emb::small_copy_flush(s);   // (one slot per stream)
  if (rc != EMB_OK) return rc;
  EMB_LOT(lot, s);
  lot.copy = emb::SmallCopy{src, dst, n};
  return EMB_OK;
}
extern "C" int emb_copy_flush(emb_stream_t stream) { return emb::small_copy_flush((hipStream_t)stream); }

extern "C" int emb_parked_count(emb_stream_t stream, int all_streams) {
  std::lock_guard<std::recursive_mutex> g(emb::lot_mutex());
  int n = 0;
  for (auto& kv : emb::lots())
    if (all_streams || kv.first == (hipStream_t)stream) n += kv.second.parked();
  return n;
}

extern "C" int emb_reset(void) {
  std::lock_guard<std::recursive_mutex> g(emb::lot_mutex());
  int n = 0;
  for (auto& kv : emb::lots()) n += kv.second.parked();
  emb::lots().clear();   // descriptors only: nothing is launched, nothing is freed; defer flags return to "off"
  return n;
}

extern "C" int emb_reset_stream(emb_stream_t stream) {
  std::lock_guard<std::recursive_mutex> g(emb::lot_mutex());
  auto it = emb::lots().find((hipStream_t)stream);
  if (it == emb::lots().end()) return 0;
  const int n = it->second.parked();
  emb::lots().erase(it);
  return n;
}
