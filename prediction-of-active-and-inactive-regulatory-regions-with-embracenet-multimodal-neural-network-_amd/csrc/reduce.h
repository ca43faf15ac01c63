// Slab reductions of the backward kernels (reduce.hip).  Every weight-gradient kernel of this library writes per-slice /
// per-workgroup partial results ("slabs") that are then summed in a fixed order -- deterministic, no float atomics.  All
// of those sums have one shape: S slices of `per` elements, element q of the sum scattered to a parameter-gradient layout.
// They run through ONE kernel that takes a table of such jobs, so a trainer can defer them (emb_reduce_defer) and pay a
// single launch per backward pass (emb_reduce_flush) instead of one per layer.
#pragma once
#include "common.h"

namespace emb {

enum { RJ_LINEAR = 0, RJ_CONV = 1, RJ_MLP = 2, RJ_HEAD_STATS = 3 };

struct ReduceJob {
  const void* in;    // [S][per] partial sums (P-typed)
  void* out[8];      // RJ_LINEAR: dW, db;  RJ_CONV: dW, dbias;  RJ_MLP: dW_0..3, db_0..3;  RJ_HEAD_STATS: loss (float), confusion (int64[4])
  long per;
  int S, kind;
  int iv[9];         // RJ_LINEAR: N, pitch (row = [N values | bias | pad], pitch 0 = N + 1);  RJ_CONV: Cin, cin_pad, k;  RJ_MLP: L, N_0..3, K_0..3
};

// immediate mode: launches on `s`; deferred mode (per stream): queued on `s` until emb_reduce_flush(s).  is_double selects P.
int reduce_submit(const ReduceJob& job, bool is_double, hipStream_t s);
bool reduce_deferring(hipStream_t s);   // jobs submitted on this stream are queued (emb_reduce_defer) rather than launched at once

// Queued jobs as seen by the multi-tensor optimizer launch (loss_optim.hip), which sums the slices of a gradient itself
// instead of reading the reduced tensor: `reduce_claim` looks for a queued job with an output == `grad`, returns its
// descriptor and which output it is, and clears that output in the queue (a job whose outputs are all claimed leaves it).
// `reduce_claim_stats` hands out queued RJ_HEAD_STATS jobs (loss / confusion sums: run as extra blocks of the same launch).
struct ReduceClaim {
  ReduceJob job;
  int which;         // index into job.out
};
bool reduce_claim(hipStream_t s, const void* grad, bool is_double, ReduceClaim* out);
bool reduce_claim_stats(hipStream_t s, bool is_double, ReduceJob* out);

// where element i of output `which` of a job lives inside one slice (the inverse of the scatter in reduce.hip)
__host__ __device__ inline long reduce_slab_index(int kind, const int* iv, int which, long i) {
  if (kind == RJ_LINEAR) {
    const int N = iv[0], pitch = iv[1] > 0 ? iv[1] : N + 1;
    return which == 0 ? (i / N) * pitch + i % N : i * pitch + N;
  }
  if (kind == RJ_CONV) {      // slab row = [k*cin_pad tap-major columns | bias]; dW in torch layout [Cout][Cin][k]
    const int Cin = iv[0], cin_pad = iv[1], k = iv[2], KK = k * cin_pad;
    if (which == 1) return i * (KK + 1) + KK;
    const long o = i / ((long)Cin * k);
    const int rem = (int)(i - o * Cin * k), ci = rem / k, tap = rem - ci * k;
    return o * (KK + 1) + (long)tap * cin_pad + ci;
  }
  // RJ_MLP: per layer [N*K weights | N biases]; outputs 0..3 = dW_l, 4..7 = db_l
  const int l = which & 3;
  long off = 0;
  for (int t = 0; t < l; ++t) off += (long)iv[1 + t] * iv[5 + t] + iv[1 + t];
  return which < 4 ? off + i : off + (long)iv[1 + l] * iv[5 + l] + i;
}

// a few floats copied by the next optimizer launch of the stream (emb_copy_park; loss_optim.hip takes it, emb_copy_flush runs what
// nobody took).  Data-parallel steps move the all-reduced class counts of the exchange block into the slot the next step's
// classifier head reads -- 8 bytes that cost a 4.6 us launch as a kernel of their own.
struct SmallCopy {              // (plain aggregate: it travels inside a kernel argument block; SmallCopy{} is the empty one)
  const float* src;
  float* dst;
  int n;
};
bool small_copy_take(hipStream_t s, SmallCopy* out);
int small_copy_flush(hipStream_t s);

}  // namespace emb
