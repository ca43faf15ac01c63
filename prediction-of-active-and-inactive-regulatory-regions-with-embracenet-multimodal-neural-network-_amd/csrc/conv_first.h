// First convolution block without a stored convolution output (conv_first.hip): every pass recomputes the
// convolution of the few-channel input on the matrix cores.  bf16, cin_pad == 8, odd k <= 15, L <= 256, Cout in {16, 32, 64}.
#pragma once
#include "common.h"
#include "bn_inline.h"

namespace emb {

int conv_first_supported(int dtype, int B, int L, int cin_pad, int Cout, int k);   // 1 when the fused kernels apply
int conv_first_blocks(int B, int L, int cin_pad, int Cout, int k);                 // workgroups = partial rows = wgrad slices
// x_codes == 0: x is [B][L][8] bf16 channels-last; else x is [B][L] uint8 base codes (0-3), one-hot expanded while staging.
// each returns EMB_OK, a negative error, or 1 when the shapes do not qualify
// conv_first_stats: x_codes == 2 -> x is the loader's [B][4][L] bf16 tensor and nlc_out receives the [B][L][8] image.
// gram_part (nullable): conv_first_gram_part_bytes() of workspace for the partial lag statistics of the input (first_gram.h)
int conv_first_stats(const void* x, int x_codes, void* nlc_out, const void* w, const void* bias, void* partial, int* rows, float* gram_part,
                     int B, int L, int Cout, int k, hipStream_t s);
// fin (apply / weight-gradient pass): non-null with fin->partial set -> the BatchNorm vectors are finalised in the launch's prologue
// from the partial rows of the preceding statistics / sums pass (bn_inline.h); no finalize launch in between.
// gram_tot (nullable): receives the totals of the lag statistics (conv_first_gram_floats() floats), built in the launch's prologue
int conv_first_apply(const void* x, int x_codes, const void* w, const void* bias, const void* stats, const BnFinFwd* fin, void* out, uint8_t* argmax, int out_ncl,
                     const float* gram_part, float* gram_tot, float drop_p, uint64_t seed, uint64_t step_val, const uint64_t* step_dev, int64_t row0, int layer_id, int B, int L,
                     int Cout, int k, hipStream_t s);
int conv_first_bwd_sums(const void* dout, int dout_ncl, const uint8_t* argmax, const void* x, int x_codes, const void* w, const void* bias,
                        const void* stats, float keep_scale, void* bpart, int* rows, int B, int L, int Cout, int k, hipStream_t s);
int conv_first_bwd_wgrad(const void* dout, int dout_ncl, const uint8_t* argmax, const void* x, int x_codes, const void* w, const void* bias,
                         const void* stats, const void* coef, const BnFinBwd* fin, float keep_scale, int training, void* slab, int* slices,
                         int B, int L, int Cout, int k, hipStream_t s);

// recompute-free backward of the first block (first_gram.h): A = g^T xview into slabs [slices][Cout][k * 8 + 1], then the per-channel
// finish from the slabs, the lag statistics of the forward (gram_tot), the packed weights and the BatchNorm vectors
int conv_first_bwd_acc(const void* dout, int dout_ncl, const uint8_t* argmax, const void* x, int x_codes, float keep_scale, void* slab, int* slices, int B, int L,
                       int Cout, int k, hipStream_t s);
int conv_first_bwd_finish(const void* slab, int slices, const float* gram_tot, const void* w, const void* bias, const void* stats, void* dW,
                          void* dbias, void* dgamma, void* dbeta, int training, int B, int L, int Cin, int Cout, int k, hipStream_t s);
int conv_first_gram_floats();
size_t conv_first_gram_part_bytes(int B, int L, int cin_pad, int Cout, int k);

}  // namespace emb
