// The per-channel finish of the first conv block's recompute-free backward (first_gram.h): its argument block, and the slot in which
// a training step parks it so that the optimizer launch can run it as extra workgroups (loss_optim.hip) instead of a launch
// of its own.
#pragma once
#include "common.h"

namespace emb {

struct FirstFinArgs {
  const float* slab;        // [S][C][64]: A partials, compact columns tap * 4 + ci (4 real input channels), column 4 k = sum of g
  const float* gram;        // [kGramRow] totals
  const __bf16* w;          // [C][KK] packed weights (tap-major, 8 input channels per tap, what the forward multiplied with)
  const float* bias;        // [C]
  const float* stats;       // [4][C] mean, invstd, scale, shift
  float* dW;                // [C][Cin][k]  (torch layout)
  float* dbias;             // [C]
  float* dgamma;
  float* dbeta;
  int S, C, k, Cin, pad, training;
  double count;             // rows behind the batch statistics (B * L)
};

// deferred mode (emb_reduce_defer): parks the job (one slot per stream; a second submit launches the first); else launches the finish kernel
int first_fin_submit(const FirstFinArgs& f, hipStream_t s);
bool first_fin_peek(hipStream_t s, FirstFinArgs* out);   // the job parked on this stream, if any (stays parked)
void first_fin_drop(hipStream_t s);                      // the optimizer launch took it over
int first_fin_flush(hipStream_t s);          // launches a parked job the classic way (emb_reduce_flush, or an optimizer call that cannot take it)
int first_fin_launch(const FirstFinArgs& f, hipStream_t s);   // conv_first.hip

// The totals of the lag statistics (first_gram.h, gram_job): kGramJobs independent jobs that only have to be done before the block's
// backward.  The apply pass parks them instead of running them in its prologue; the head / loss launch of the same forward
// (head.hip) carries them as extra workgroups on CUs it leaves idle.  Anything that would overwrite their inputs or needs their
// output flushes a parked set as a launch of its own: the block's next forward, its backward, emb_reduce_flush.
struct GramJobsArgs {
  const __bf16* edge;       // edge image [112][B]
  const float* part;        // [rows][512] partial G0 rows
  float* tot;               // [4096] totals
  int B, L, rows, parts;
};
void gram_jobs_park(const GramJobsArgs& a, hipStream_t s);    // (one slot per stream: a set parked earlier is launched first)
bool gram_jobs_take(hipStream_t s, GramJobsArgs* out);        // a carrier launch of this stream takes the parked set over
int gram_jobs_flush(hipStream_t s);
int gram_jobs_launch(const GramJobsArgs& a, hipStream_t s);   // conv_first.hip
int gram_jobs_count();                                        // kGramJobs

}  // namespace emb
