/* embrace_hip.h -- C ABI of libembrace_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for ONE path of the reference
 * (nikiiny/Prediction-of-Active-and-Inactive-Regulatory-Regions-with-Embracenet-Multimodal-Neural-Network-):
 * EmbraceNet docking -> ReLU -> multinomial modality selection -> masked-sum fusion -> post MLP ->
 * 2-class weighted cross-entropy, forward and backward, plus the optimizer step.
 * The reference has no FFI of its own (it is pure Python on ATen); each entry point below names the
 * reference lines (relative to /root/reference/BIOINF_tesi/models/) it replaces.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory unless marked "host";
 *  - the caller owns every buffer; the library allocates no device memory.  Host-side state it keeps: the thread-local
 *    error string, the registry of packed conv-weight images (emb_conv_pack_register) and -- only between an
 *    emb_reduce_defer / emb_rider_defer call and the matching flush -- launch descriptors parked PER STREAM (queued slab
 *    reductions, one rider launch, the first conv block's finish / totals jobs).  That state is keyed by the stream and
 *    mutex-guarded: independent trainers on different streams (from any host threads) do not interact, and emb_reset()
 *    drops all of it;
 *  - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*), never synchronises,
 *    never reads device memory from the host -> safe under hipGraph stream capture;
 *  - return value: 0 on success, negative EMB_ERR_* otherwise (text via emb_last_error());
 *  - dtype: EMB_F32 | EMB_BF16 | EMB_F64 is the storage type of activations and weights ("T").
 *    "P" is the parameter/gradient type: float for EMB_F32 and EMB_BF16 (bf16 weights are shadows
 *    of fp32 masters), double for EMB_F64.  Accumulation is fp32 (fp64 for EMB_F64).
 *  - matrices are dense row-major; weights use the nn.Linear layout [out_features, in_features];
 *  - 16-byte alignment of every base pointer is required (torch allocations satisfy it).
 *
 * RNG contract (perf mode; parity mode injects the host generator's uniforms instead):
 *  Philox4x32-10, key = seed, counter = (index.lo, index.hi, stream.lo, stream.hi) with
 *  stream = (step << 8) | kind and step = step_val + (step_dev ? *step_dev : 0).
 *  kinds: 0 selection uniform of element e = global_row * c + j -> word (e & 3) of the counter e >> 2, u = word * 2^-32
 *           (one Philox call per four consecutive elements; idx = cdf0 < u  <=>  floor(cdf0 * 2^32) < word);
 *         1 modality-dropout gate (index 0) -> 24-bit float from word 0;
 *         2 per-row dropped-modality draw (index global_row) -> 24-bit float from word 0;
 *         16+L activation-dropout mask of layer L (index global_row * N + j) -> 24-bit float.
 *  Keying on GLOBAL rows makes every draw invariant to how the batch is sharded over GPUs.
 */
#ifndef EMBRACE_HIP_H_
#define EMBRACE_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMB_ABI_VERSION 2

enum { EMB_F32 = 0, EMB_BF16 = 1, EMB_F64 = 2 };
enum { EMB_OK = 0, EMB_ERR_ARG = -1, EMB_ERR_DTYPE = -2, EMB_ERR_ALIGN = -3, EMB_ERR_LAUNCH = -4 };
enum { EMB_RNG_SELECT = 0, EMB_RNG_GATE = 1, EMB_RNG_ROWMOD = 2, EMB_RNG_DROPOUT0 = 16 };
/* bits of the per-element code byte written by emb_embrace_fwd: the selected modality, whether its ReLU was active, and
 * (derived, for the backward kernel's fragment masks) "gradient flows to docking_0 / docking_1 here" = ACTIVE && IDX == 0 / 1 */
enum { EMB_CODE_IDX = 1, EMB_CODE_ACTIVE = 2, EMB_CODE_KEEP0 = 64, EMB_CODE_KEEP1 = 128 };
/* bits of the status word */
enum { EMB_STATUS_INVALID_DISTRIBUTION = 1 };

typedef void* emb_stream_t; /* hipStream_t */

int emb_abi_version(void);
const char* emb_last_error(void);

/* EmbraceNetMultimodal.py:63-76 (availability * probability, renormalise), :178-184 (modality dropout,
 * selection-probability broadcast) and the cdf construction inside torch.multinomial (:84).
 *   p        [p_rows, 2] fp32, p_rows == 1 (broadcast, :184) or B
 *   avail    [B, 2] fp32 or NULL (all available, :64-65); ignored when device_dropout != 0
 *   device_dropout != 0: draw the gate r and the per-row dropped modality in-kernel (RNG kinds 1, 2)
 *            with the reference's semantics: r >= 0.5 -> avail = one_hot(round(rand[row]))
 *   cdf0     [B] fp32 out: the threshold torch.multinomial compares its uniforms with;
 *            NaN marks an invalid row and sets EMB_STATUS_INVALID_DISTRIBUTION in *status (atomic or)
 *   row0     global index of local row 0 (data-parallel shard offset) */
int emb_select_prep(const float* p, int p_rows, const float* avail, int device_dropout,
                    uint64_t seed, uint64_t step_val, const uint64_t* step_dev, int64_t row0,
                    float* cdf0, int32_t* status, int B, emb_stream_t stream);

/* EmbraceNetMultimodal.py:52-60 (docking Linear + ReLU, both modalities), :80 (stack), :84 (multinomial),
 * :85 (one_hot), :87-88 (mul, sum) -- one fused kernel, nothing [B,c,M]-shaped is materialised.
 *   X0 [B,d0] T, X1 [B,d1] T, W0 [c,d0] T, W1 [c,d1] T, b0,b1 [c] P
 *   cdf0 [B] fp32 from emb_select_prep
 *   u    [B,c] fp64 uniforms of the host generator (parity mode) or NULL -> Philox kind 0
 *   E    [B,c] T out;  code [B,c] u8 out (EMB_CODE_* bits; idx is bit 0) */
int emb_embrace_fwd(const void* X0, const void* X1, const void* W0, const void* b0, const void* W1,
                    const void* b1, const float* cdf0, const double* u, uint64_t seed, uint64_t step_val,
                    const uint64_t* step_dev, int64_t row0, void* E, uint8_t* code, int B, int d0, int d1,
                    int c, int dtype, emb_stream_t stream);
/* The same with the selection threshold of every row computed inside the launch from (p, avail, device_dropout) -- the
 * arguments and arithmetic of emb_select_prep, *status as there -- so that EmbraceNet.forward (:34-90) is ONE kernel. */
int emb_embrace_fwd_select(const void* X0, const void* X1, const void* W0, const void* b0, const void* W1,
                           const void* b1, const float* p, int p_rows, const float* avail, int device_dropout,
                           int32_t* status, const double* u, uint64_t seed, uint64_t step_val, const uint64_t* step_dev,
                           int64_t row0, void* E, uint8_t* code, int B, int d0, int d1, int c, int dtype,
                           emb_stream_t stream);

/* autograd of the above (loss.backward(), utils/training_models_multimodal.py:156):
 *   dD_m = dE * [idx == m] * [pre_m > 0];  dW_m = dD_m^T X_m;  db_m = sum_b dD_m;  dX_m = dD_m W_m
 *   dE [B,c] T, code [B,c] u8;  dX0 [B,d0] T, dX1 [B,d1] T (either may be NULL: not needed);
 *   dW0 [c,d0] P, db0 [c] P, dW1 [c,d1] P, db1 [c] P  (overwritten, not accumulated)
 *   workspace (nullable): transient scratch; when given and c*d is only a few tiles, the weight gradients are split
 *   over the batch into per-slice partial sums and reduced in a fixed order (second small launch). */
int emb_embrace_bwd(const void* dE, const uint8_t* code, const void* X0, const void* X1, const void* W0,
                    const void* W1, void* dX0, void* dX1, void* dW0, void* db0, void* dW1, void* db1,
                    void* workspace, int64_t workspace_bytes, int B, int d0, int d1, int c, int dtype,
                    emb_stream_t stream);

/* EmbraceNet with bypass_docking=True (EmbraceNetMultimodal.py:20 "connect the input data directly to the embracement
 * layer", :54-55): the inputs are the docking outputs, the layer is the selection alone -- :63-76 (availability *
 * probability, renormalise), :80 (stack), :84 (multinomial), :85 (one_hot), :87-88 (mul, sum).  One elementwise launch.
 *   X0, X1 [B,c] T;  E [B,c] T out;  code [B,c] u8 out (IDX bit 0; ACTIVE always set: there is no ReLU on this path)
 *   thresholds: cdf0 [B] fp32 from emb_select_prep, or cdf0 == NULL and (p, p_rows, avail, device_dropout, status) with
 *   the meaning and arithmetic of emb_embrace_fwd_select;  u [B,c] fp64 or NULL -> Philox kind 0 (same RNG contract). */
int emb_embrace_bypass_fwd(const void* X0, const void* X1, const float* cdf0, const float* p, int p_rows,
                           const float* avail, int device_dropout, int32_t* status, const double* u, uint64_t seed,
                           uint64_t step_val, const uint64_t* step_dev, int64_t row0, void* E, uint8_t* code, int B, int c,
                           int dtype, emb_stream_t stream);
/* autograd of the above: dX_m = dE * [idx == m];  dX0, dX1 [B,c] T, either may be NULL (not needed). */
int emb_embrace_bypass_bwd(const void* dE, const uint8_t* code, void* dX0, void* dX1, int B, int c, int dtype,
                           emb_stream_t stream);

/* EmbraceNet.forward for any number of modalities M (1 <= M <= 8; EmbraceNetMultimodal.py:46-48 takes len(input_list)):
 * the docking layers (:52-60) are M calls of emb_linear_fwd with relu != 0, the rest of the layer is the three entries below.
 * (The reference's own call sites use M == 2, which emb_embrace_fwd fuses into one launch.)
 *
 * emb_select_prep_m -- :63-76 and the cdf torch.multinomial builds (:84), as emb_select_prep for M columns:
 *   p [p_rows, M] fp32 (p_rows == 1 or B), avail [B, M] fp32 or NULL, cdf [B, M] fp32 out (NaN row + status bit when invalid)
 * emb_embrace_select_fwd -- :80 (stack), :84 (multinomial), :85 (one_hot), :87-88 (mul, sum):
 *   D     host array of M device pointers, D[m] = docking output of modality m, [B,c] T
 *   idx   = number of cdf entries m < M-1 with (double)cdf[row][m] < u  (ATen's binary search);  u, seed, step, row0: as
 *           emb_embrace_fwd (u [B,c] fp64 or NULL -> Philox kind 0)
 *   E [B,c] T out;  code [B,c] u8 out = idx (NOT the EMB_CODE_* bits of the two-modality kernels)
 * emb_embrace_select_bwd -- autograd of the above: dD[m] = dE * [idx == m];  dD host array of M device pointers (NULL: skip) */
int emb_select_prep_m(const float* p, int p_rows, const float* avail, float* cdf, int32_t* status, int B, int M,
                      emb_stream_t stream);
int emb_embrace_select_fwd(const void* const* D, int M, const float* cdf, const double* u, uint64_t seed,
                           uint64_t step_val, const uint64_t* step_dev, int64_t row0, void* E, uint8_t* code, int B, int c,
                           int dtype, emb_stream_t stream);
int emb_embrace_select_bwd(const void* dE, const uint8_t* code, void* const* dD, int M, int B, int c, int dtype,
                           emb_stream_t stream);

/* One layer of the post stack, EmbraceNetMultimodal.py:143-147 / :151 (also FFNN_pre.py:25-33):
 * Y = dropout(relu(X W^T + b)).   X [B,K] T, W [N,K] T, b [N] P, Y [B,N] T.
 *   relu != 0 applies ReLU; dropout_p > 0 applies an inverted-dropout mask from RNG kind 16+layer_id.
 *   mask [B,N] u8 out (nullable when relu == 0 and dropout_p == 0): bit0 = pre-activation > 0,
 *   bit1 = kept by dropout. */
int emb_linear_fwd(const void* X, const void* W, const void* b, void* Y, uint8_t* mask, int relu,
                   float dropout_p, int layer_id, uint64_t seed, uint64_t step_val, const uint64_t* step_dev,
                   int64_t row0, int B, int K, int N, int dtype, emb_stream_t stream);

/* backward of emb_linear_fwd: dZ = dY * mask-derived factor; dX = dZ W; dW = dZ^T X; db = sum_b dZ.
 *   dX [B,K] T (nullable), dW [N,K] P, db [N] P
 *   workspace (nullable): transient scratch; when given and the layer is small (few output tiles, long batch)
 *   the weight gradient is split over the batch and reduced in a fixed order. */
int emb_linear_bwd(const void* dY, const uint8_t* mask, const void* X, const void* W, void* dX, void* dW,
                   void* db, int relu, float dropout_p, void* workspace, int64_t workspace_bytes, int B, int K,
                   int N, int dtype, emb_stream_t stream);

/* utils/utils.py:121-140 (per-batch class weights) + nn.CrossEntropyLoss(weight) on output.float()
 * (utils/training_models_multimodal.py:140-141,151-154) + argmax confusion counts for the per-batch
 * AUPRC / F1 (utils/utils.py:80-94), device-resident.
 *   logits [B,2] T, target [B] int64
 *   class_counts [2] int64: IN -- (positives, rows) of the GLOBAL batch when global_counts != 0
 *                (filled by an all-reduce under data parallelism), else computed in-kernel and written
 *   loss  [1] fp32 out: sum_i w[y_i] nll_i / sum_i w[y_i]  (local numerator / global denominator under DP)
 *   dlogits [B,2] T out (nullable): d loss / d logits
 *   confusion [4] int64 out (nullable): TP, predicted-positive, positive, n of THIS call's rows (written;
 *                the caller keeps one slot per step and evaluates AP / P / R / F1 once per epoch)
 *   tick_a, tick_b uint64 device counters (nullable) incremented by one by this launch: a trainer passes the model's
 *                RNG step counter (advances after every forward) and the optimizer's step counter (advances
 *                before every update) so that a step needs no emb_counter_add launches */
int emb_weighted_ce(const void* logits, const int64_t* target, int64_t* class_counts, int global_counts,
                    float* loss, void* dlogits, int64_t* confusion, uint64_t* tick_a, uint64_t* tick_b, int B,
                    int dtype, emb_stream_t stream);

/* counts positives of a label shard into class_counts[0..1] = (pos, n) (for the DP all-reduce) */
int emb_count_labels(const int64_t* target, int64_t* class_counts, int B, emb_stream_t stream);

/* Optimizers (utils/training_models_multimodal.py:318-325, :158): one launch updates a flat
 * parameter buffer.  param/grad/state are P-typed [n]; shadow (nullable) receives the bf16 copy used by
 * the EMB_BF16 kernels.  step = step_val + *step_dev (1-based). Weight decay is coupled L2 as in torch. */
int emb_adam_step(void* param, const void* grad, void* exp_avg, void* exp_avg_sq, void* bf16_shadow,
                  int64_t n, double lr, double beta1, double beta2, double eps, double weight_decay,
                  uint64_t step_val, const uint64_t* step_dev, int dtype, emb_stream_t stream);
int emb_rmsprop_step(void* param, const void* grad, void* square_avg, void* bf16_shadow, int64_t n, double lr,
                     double alpha, double eps, double weight_decay, int dtype, emb_stream_t stream);
int emb_nadam_step(void* param, const void* grad, void* exp_avg, void* exp_avg_sq, double* m_schedule,
                   void* bf16_shadow, int64_t n, double lr, double beta1, double beta2, double eps,
                   double weight_decay, double schedule_decay, uint64_t step_val, const uint64_t* step_dev,
                   int dtype, emb_stream_t stream);

/* Riders (csrc/rider.h).  The epigenomic MLP stack (FFNN_pre.py) and the sequence CNN (CNN_pre.py) are independent until the
 * EmbraceNet layer joins them; the MLP launches are tiny and latency-bound.  With deferral armed on a stream
 * (emb_rider_defer(stream, 1)), an emb_mlp_fwd / emb_mlp_bwd call that takes the bf16 matrix-core kernels (widths % 16 == 0, F % 8 == 0)
 * does NOT launch: it is parked and carried as the first workgroups of the next carrier launch on the same stream --
 * emb_convblock_fwd's statistics pass of the fused first block for a forward, emb_convblock_bwd's BatchNorm gather pass of a
 * stored-activation block for a backward -- so the two chains overlap on the CUs without a second stream.  emb_rider_defer(stream, 0)
 * disarms (a parked launch stays parked); emb_rider_flush(stream) launches a parked rider on its own.  The optimizer and reduction
 * entry points flush first.  The caller keeps every tensor of a parked launch alive, and does not read its outputs, until a
 * carrier has run or the flush.  At most one rider is parked per stream (parking a second one flushes the first); the slot belongs to the
 * stream, not to the calling thread (autograd runs backward nodes on its own thread). */
int emb_rider_defer(emb_stream_t stream, int on);
int emb_rider_flush(emb_stream_t stream);

/* Small MLP stacks fused into one forward launch and two backward launches (FFNN_pre.py:18-49; the post stack
 * and the Linear(->2) head of EmbraceNetMultimodal.py:134-154).  L <= 4 layers, Y_l = dropout(relu(Y_{l-1} W_l^T +
 * b_l)) with per-layer relu flag / dropout_p / RNG layer id.  Pointer and int arrays are HOST arrays of length L
 * (they travel in the kernel argument); h[l] [B][N_l] T receives every layer's output (h[L-1] is the result),
 * mask[l] [B][N_l] u8 its mask byte (entry may be NULL for a layer with neither ReLU nor dropout).
 * emb_mlp_supported() tells whether a stack fits the kernel's LDS budget; otherwise call emb_linear_* per layer.
 * emb_mlp_bwd: dy [B][N_{L-1}] T -> dx [B][F] T (nullable), dW[l] [N_l][K_l] P, db[l] [N_l] P; workspace holds
 * ceil(B/16) partial sums of all weight gradients, reduced in fixed order. */
int emb_mlp_supported(int F, const int* N, int L, int dtype);
/* bytes of `workspace` emb_mlp_bwd needs for a batch of B rows (0 when the stack is not eligible) */
int64_t emb_mlp_workspace_bytes(int F, const int* N, int L, int B, int dtype);
int emb_mlp_fwd(const void* x, const void* const* W, const void* const* b, void* const* h, uint8_t* const* mask,
                const int* N, const int* relu, const float* dropout_p, const int* layer_id, int L, int B, int F,
                uint64_t seed, uint64_t step_val, const uint64_t* step_dev, int64_t row0, int dtype,
                emb_stream_t stream);
int emb_mlp_bwd(const void* x, const void* const* W, const void* const* h, const uint8_t* const* mask, const void* dy,
                void* dx, void* const* dW, void* const* db, const int* N, const int* relu, const float* dropout_p,
                int L, int B, int F, void* workspace, int64_t workspace_bytes, int dtype, emb_stream_t stream);

/* Multi-tensor forms: ONE launch updates `ntensors` parameter tensors (<= 40 per launch, chunked inside).
 * The pointer / size arrays are HOST arrays read at call time (they travel in the kernel argument, so a
 * captured call replays with the captured addresses); the tensors they point to are device memory.
 * Nadam's m_schedule is one shared device double[2] (all parameters share the momentum schedule). */
int emb_adam_step_multi(void* const* params, const void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                        void* const* bf16_shadows, const int64_t* sizes, int ntensors, double lr, double beta1,
                        double beta2, double eps, double weight_decay, uint64_t step_val, const uint64_t* step_dev,
                        int dtype, emb_stream_t stream);
int emb_rmsprop_step_multi(void* const* params, const void* const* grads, void* const* square_avg,
                           void* const* bf16_shadows, const int64_t* sizes, int ntensors, double lr, double alpha,
                           double eps, double weight_decay, int dtype, emb_stream_t stream);
int emb_nadam_step_multi(void* const* params, const void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                         double* m_schedule, void* const* bf16_shadows, const int64_t* sizes, int ntensors, double lr,
                         double beta1, double beta2, double eps, double weight_decay, double schedule_decay,
                         uint64_t step_val, const uint64_t* step_dev, int dtype, emb_stream_t stream);

/* ---- sequence pre-network; activations channels-last x[B][L][C] --------------------------
 * One block of CNN_pre.py:37-50: Conv1d(k odd, stride 1, same padding) -> BatchNorm1d -> ReLU ->
 * MaxPool1d(10, 2) [-> Dropout].  cin_pad = input channels rounded up to the 16-byte vector (zero channels).
 *   emb_ncl_to_nlc        x[B][C][L] (reference loader layout, dataprepare.py:398-412) -> out[B][L][Cpad] T
 *   emb_conv_pack_weight  nn.Conv1d weight W[Cout][Cin][k] P -> wpack[Cout][k*cin_pad] T (tap-major) and,
 *                         when wflip != NULL, wflip[cin_pad][k*Cout] T (flipped taps, for the input gradient)
 *   emb_convblock_workspace_bytes  scratch the two calls below need (caller allocates; contents are transient)
 *   emb_convblock_fwd     y[B][L][Cout] T = conv + bias (kept for backward); stats[4][Cout] P = mean, invstd,
 *                         scale, shift (batch statistics when training != 0, running statistics otherwise;
 *                         running_mean/var updated with `momentum` as nn.BatchNorm1d does, *num_batches_tracked
 *                         (nullable, device int64) incremented when training);
 *                         out = dropout(maxpool(relu(bn(y)))) as [B][Lp][Cout] or, out_ncl != 0, [B][Cout][Lp]
 *                         (the flatten order of CNN_pre.py:74); argmax[B][Lp][Cout] u8 (bits 0-3 window
 *                         offset of the maximum, bit 7 dropped).  Dropout mask: RNG kind 16+layer_id.
 *   emb_convblock_bwd     dout (layout as `out`) -> dx[B][L][cin_pad] T (nullable), dW[Cout][Cin][k] P (torch
 *                         layout), dbias, dgamma, dbeta [Cout] P; dy[B][L][Cout] T is caller-provided scratch.
 *   emb_convblock_needs_y 0 for the FIRST block of the stack when its input has so few channels (cin_pad == 8, bf16,
 *                         odd k <= 15, L <= 256, Cout in {16, 32, 64}) that the convolution is cheaper to recompute
 *                         than to store: the caller then passes y = NULL to emb_convblock_fwd and y = dy = dx = NULL
 *                         plus wpack and bias to emb_convblock_bwd, and the [B][L][Cout] tensors never exist
 *                         (forward: statistics pass + fused conv/BN/ReLU/pool pass; backward: reduction pass + fused
 *                         dz/weight-gradient pass).  The argmax byte then also carries bit 6 = pooled output is 0.
 *                         In that mode the input may be staged as BASE CODES: x_codes != 0 -> x is
 *                         uint8[B][L] with 0-3 = the hot channel of the one-hot column (dataprepare.py:398-412 order) and
 *                         any other value = an all-zero column; the one-hot row is formed in LDS, the [B][4][L] float
 *                         tensor and its layout conversion never exist (8x less input traffic).  x_codes = 0 otherwise.
 *                         x_codes = 2 (emb_convblock_fwd, training, bf16, no bn_phase 2): x is the loader's [B][4][L] tensor
 *                         itself; the statistics pass stages from it and writes the channels-last image [B][L][8] to `y`
 *                         (here an OUTPUT the caller keeps for emb_convblock_bwd, which takes it as x with x_codes = 0) --
 *                         the emb_ncl_to_nlc launch disappears from the training step.
 *   bn_phase / bn_sums    BatchNorm statistics of the GLOBAL batch when the batch rows are sharded over processes
 *                         (nn.BatchNorm1d at CNN_pre.py:41 normalises over the whole batch).  bn_phase = 0:
 *                         one call, statistics of this call's rows (bn_sums ignored).  Otherwise the block is two calls
 *                         with the same arguments around one SUM all-reduce of bn_sums (2*Cout+1 doubles, caller-owned):
 *                           forward   phase 1: convolution + sums -> bn_sums = {sum y[c]}, {sum y[c]^2}, rows;
 *                                     phase 2: mean / invstd / running statistics from bn_sums, then BN/ReLU/pool/dropout;
 *                           backward  phase 1: dz + sums -> bn_sums = {sum dz[c]}, {sum dz[c]*xhat[c]}, rows; dgamma / dbeta
 *                                     written (LOCAL sums: the gradient all-reduce adds them like any other gradient);
 *                                     phase 2: dy from the all-reduced means, weight / bias / input gradients.
 *                         Training mode only (eval uses the running statistics, nothing to exchange). */
int64_t emb_convblock_workspace_bytes(int B, int L, int cin_pad, int Cout, int k, int dtype);
int emb_ncl_to_nlc(const void* x, int src_dtype, void* out, int dst_dtype, int B, int C, int L, int Cpad,
                   emb_stream_t stream);
int emb_conv_pack_weight(const void* W, void* wpack, void* wflip, int Cout, int Cin, int cin_pad, int k, int dtype,
                         emb_stream_t stream);
/* Keep packed images current without a pack launch per step: after emb_conv_pack_register(W, wpack, wflip, ...)
 * every emb_*_step_multi call that updates the fp32 parameter at address W also writes the new values into wpack
 * (and wflip when not NULL), in the layouts of emb_conv_pack_weight; `dtype` is the images' element type
 * (EMB_BF16 for bf16 compute on fp32 masters, EMB_F32 for fp32 compute).  The caller keeps the buffers alive and
 * unregisters before freeing W or them.  Host-side table only; no device work. */
int emb_conv_pack_register(const void* W, void* wpack, void* wflip, int Cout, int Cin, int cin_pad, int k, int dtype);
int emb_conv_pack_unregister(const void* W);
int emb_convblock_fwd(const void* x, const void* wpack, const void* bias, const void* gamma, const void* beta,
                      void* running_mean, void* running_var, int training, double momentum, double eps,
                      float dropout_p, uint64_t seed, uint64_t step_val, const uint64_t* step_dev, int64_t row0,
                      int layer_id, void* y, void* stats, void* out, uint8_t* argmax, int out_ncl, void* workspace,
                      int64_t workspace_bytes, int64_t* num_batches_tracked, int x_codes, int bn_phase, double* bn_sums,
                      int B, int L, int cin_pad, int Cout, int k, int dtype, emb_stream_t stream);
int emb_convblock_bwd(const void* dout, int dout_ncl, const uint8_t* argmax, const void* y, const void* stats,
                      const void* x, const void* wflip, const void* wpack, const void* bias, float dropout_p,
                      int training, void* dx, void* dW, void* dbias, void* dgamma, void* dbeta, void* dy,
                      void* workspace, int64_t workspace_bytes, int x_codes, int bn_phase, double* bn_sums, int B, int L,
                      int Cin, int cin_pad, int Cout, int k, int dtype, emb_stream_t stream);
int emb_convblock_needs_y(int B, int L, int cin_pad, int Cout, int k, int dtype);
/* elements (of P) the `stats` buffer of emb_convblock_fwd / emb_convblock_bwd must have: 4 * Cout, plus -- for the fused first
 * block in bf16 -- the lag statistics of the input that the forward leaves there for the recompute-free backward
 * (csrc/first_gram.h: the first block's convolution is linear in its weights, so the BatchNorm backward is assembled from
 * A = g^T xview, the input's lag statistics and the weights instead of re-running the convolution twice). */
int64_t emb_convblock_stats_elems(int B, int L, int cin_pad, int Cout, int k, int dtype);
/* The recompute-free backward of the fused first block (on by default): on != 0 /
 * on == 0 switches it for the following emb_convblock_fwd + emb_convblock_bwd PAIRS (the backward needs what its forward left in
 * `stats`), on < 0 only asks.  Returns the previous setting.  In a deferring step (emb_reduce_defer) that backward parks its
 * per-channel finish until the optimizer launch (emb_*_step_multi holding the block's weight, gamma and beta gradients) or
 * emb_reduce_flush: `stats`, the workspace, `wpack` and `bias` of the call must stay valid and untouched until then.
 * The forward likewise parks the jobs that write the lag-statistics totals into `stats`: the next emb_head_ce on the stream carries
 * them; otherwise the block's emb_convblock_bwd, its next emb_convblock_fwd or emb_reduce_flush launch them.  `stats` and the
 * workspace of the forward must therefore stay valid until one of those calls (they do in any forward -> backward sequence). */
int emb_convblock_first_linear(int on);

/* Slab reductions.  The weight-gradient kernels of emb_embrace_bwd / emb_linear_bwd / emb_mlp_bwd / emb_convblock_bwd write
 * per-slice partial sums into the caller's workspace and finish with a (deterministic, fixed-order) reduction launch.
 * After emb_reduce_defer(stream, 1) the reductions of calls on THAT stream are only queued; emb_reduce_flush(stream) runs all of
 * the stream's queued ones in one launch.  Until the flush the parameter gradients are incomplete and every workspace handed to
 * a queued call must stay untouched (give each call site its own workspace).  Per-stream switch, default off (immediate).
 * The multi-tensor optimizer entry points (emb_adam_step_multi / emb_rmsprop_step_multi / emb_nadam_step_multi) CONSUME queued
 * jobs: a job whose outputs are gradient tensors of the launch is taken off the queue and its slices are summed inside the
 * optimizer launch (fixed order: same result as the reduction launch, which then is not needed at all); the summed gradient is
 * also stored to the gradient tensor.  Jobs that are not claimed stay queued for emb_reduce_flush.  Both entry points first
 * launch a parked rider (emb_rider_flush), which may be the producer of a queued slab. */
int emb_reduce_defer(emb_stream_t stream, int on);
int emb_reduce_flush(emb_stream_t stream);
/* Parked launch descriptors (queued reductions, rider, first-block finish / totals jobs) of `stream`, or of every stream when
 * all_streams != 0.  0 after a completed step. */
int emb_parked_count(emb_stream_t stream, int all_streams);
/* Parks a copy of n (1 .. 64) floats src -> dst on `stream`: the next emb_*_step_multi launch of the stream performs it (its first
 * workgroup; the launch reads none of these floats otherwise), emb_copy_flush runs it as a launch of its own if none did.  One slot
 * per stream (parking a second copy first runs the parked one).  Data-parallel steps use it for the 8 bytes of all-reduced class
 * counts the next step's emb_head_ce reads (global_counts == 2). */
int emb_copy_park(emb_stream_t stream, const float* src, float* dst, int n);
int emb_copy_flush(emb_stream_t stream);
/* Drops everything parked on every stream WITHOUT launching it and returns every defer switch to off: the recovery call after an
 * exception between a deferring call and its flush (the parked descriptors point at tensors that may be gone).  Returns the
 * number of descriptors dropped.  Touches no device memory. */
int emb_reset(void);
/* The same for ONE stream (what a trainer calls when its step raised between a deferring call and the flush). */
int emb_reset_stream(emb_stream_t stream);

/* ---- classifier head + loss in one launch (rows a11-a13 of SURVEY 8 fused) ------------------------------------
 * emb_head_ce      logits[B][2] T = E[B][K] . W[2][K]^T + bias (the final nn.Linear(width, 2), EmbraceNetMultimodal.py:151-154,
 *                  :190; W / bias fp32), the per-batch class-weighted cross-entropy on them (semantics of emb_weighted_ce) and,
 *                  when dE != NULL, the head's backward: dE[B][K] T = dlogits . W, with dW / db left as per-workgroup partial
 *                  sums in `workspace`.  tick_a / tick_b as in emb_weighted_ce.  T in {f32, bf16}, K % 4 == 0, K <= 1024
 *                  (emb_head_ce_supported).  Replaces three dependent launches (last emb_mlp_fwd layer, emb_weighted_ce, last
 *                  emb_mlp_bwd layer).
 * emb_head_ce_finish  sums the partials (reduce.hip; queued under emb_reduce_defer): dW[2][K], db[2] fp32 (both NULL in
 *                  evaluation), *loss (fp32) and confusion[4] int64 = {tp, predicted positives, positives, rows} of this call's
 *                  rows.  `workspace` must stay untouched between the two calls (and until emb_reduce_flush when deferred). */
int emb_head_ce_supported(int B, int K, int dtype);
int64_t emb_head_ce_workspace_bytes(int B, int K);
int emb_head_ce(const void* E, const void* W, const void* bias, const int64_t* target, int64_t* class_counts, int global_counts,
                void* logits, void* dE, void* workspace, int64_t workspace_bytes, uint64_t* tick_a, uint64_t* tick_b, int B,
                int K, int dtype, emb_stream_t stream);
int emb_head_ce_finish(const void* workspace, void* dW, void* db, float* loss, int64_t* confusion, int B, int K,
                       emb_stream_t stream);
/* emb_head_ce with the fusion layer's backward prepared inside it: when the head sits directly on the EmbraceNet output
 * (n_post_layers = 0, EmbraceNetMultimodal.py:134-154), `code` = the forward's code bytes [B][K] and dD0 / dD1 [B][K] T receive
 * the PRE-MASKED gradients dD_m = dE * keep_m (keep_m = EMB_CODE_KEEP0 / KEEP1: selected modality and active ReLU) that
 * emb_embrace_bwd_masked multiplies -- the mask is applied once, by the kernel that produces dE, instead of per MFMA fragment
 * in the backward GEMMs.  code, dD0, dD1: all three or none (then identical to emb_head_ce).  dE may be NULL when only the
 * masked gradients are wanted.
 * global_counts (both entry points): 0 = count this batch's labels (class_counts int64[2] is written), 1 = class_counts holds
 * the (positives, rows) of the GLOBAL batch, 2 = class_counts is a float[4] exchange block: [0..1] the global counts as floats
 * (read), [2..3] receive this shard's own (positives, rows) -- a data-parallel trainer lets [2..3] ride in its gradient
 * all-reduce and copies the sums to [0..1] for the next step (bench.py). */
int emb_head_ce_masked(const void* E, const void* W, const void* bias, const int64_t* target, int64_t* class_counts,
                       int global_counts, void* logits, void* dE, const uint8_t* code, void* dD0, void* dD1, void* workspace,
                       int64_t workspace_bytes, uint64_t* tick_a, uint64_t* tick_b, int B, int K, int dtype, emb_stream_t stream);

/* ---- EmbraceNet backward on pre-masked gradients (csrc/gemm_jobs.h; autograd through EmbraceNetMultimodal.py:52-60,80-88) ----
 * emb_embrace_premask      dD0 = dE * keep0, dD1 = dE * keep1 (elementwise; for producers of dE other than emb_head_ce_masked);
 *                          B * c % 8 == 0, EMB_F32 / EMB_BF16.
 * emb_embrace_bwd_masked   the four GEMMs of emb_embrace_bwd (dX_m = dD_m W_m, dW_m = dD_m^T X_m, db_m = sum_b dD_m) as one
 *                          persistent launch streaming 32 KB operand stages through a five-slot LDS ring (LDS-DMA, four stages
 *                          ahead, across tile boundaries).  Same outputs, workspace and slab / reduction contract as
 *                          emb_embrace_bwd.  EMB_F32 (v_mfma_f32_16x16x4_f32), c % 4 == 0, d0 % 4 == 0, d1 % 4 == 0
 *                          (emb_embrace_bwd_masked_supported); bf16 keeps emb_embrace_bwd (csrc/embrace_bwd_split.h), which
 *                          measured faster there. */
int emb_embrace_premask(const void* dE, const uint8_t* code, void* dD0, void* dD1, int B, int c, int dtype, emb_stream_t stream);
int emb_embrace_bwd_masked_supported(int B, int d0, int d1, int c, int dtype);
int emb_embrace_bwd_masked(const void* dD0, const void* dD1, const void* X0, const void* X1, const void* W0, const void* W1,
                           void* dX0, void* dX1, void* dW0, void* db0, void* dW1, void* db1, void* workspace,
                           int64_t workspace_bytes, int B, int d0, int d1, int c, int dtype, emb_stream_t stream);


/* ---- input staging ---------------------------------------------------------------------------
 * The split (features, sequence codes / one-hot windows, labels) stays resident in HBM; a batch is a row gather:
 *   dst[t][i][0..row_bytes[t]) = src[t][idx[i]][0..row_bytes[t])     t < n_tables (1..4), i < n
 * for all tables in one launch (they share the index list idx[n], int64 on the device; n_rows = rows of every source
 * table).  src / dst / row_bytes are HOST arrays of n_tables entries (device pointers and sizes in bytes).  Replaces the
 * per-sample fetch + .to(device) of Dataset_Wrap.__getitem__ and the DataLoader collate (data_pipe/dataprepare.py:399-412,
 * :589-594).  An index outside [0, n_rows) yields a zero row (never an out-of-bounds read). */
int emb_gather_rows(const void* const* src, void* const* dst, const int64_t* row_bytes, int n_tables, const int64_t* idx,
                    int64_t n, int64_t n_rows, emb_stream_t stream);
/* HOST helper for the balanced batch lists (BalancePos_BatchSampler.__iter__, data_pipe/dataprepare.py:431-447, shuffles
 * with Python's `random`): random.shuffle(items) restated natively -- MT19937 state[624] + position as returned by
 * random.Random(seed).getstate(), advanced in place, so consecutive shuffles continue one stream exactly as the module
 * generator does.  items[n] int64 in host memory, n < 2^31.  No device work. */
int emb_mt19937_shuffle(uint32_t* state, int* pos, int64_t* items, int64_t n);

/* helpers: dtype conversion (fp32/fp64 master -> bf16 shadow etc.) and a device step counter */
int emb_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, emb_stream_t stream);
int emb_counter_add(uint64_t* counter, uint64_t inc, emb_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EMBRACE_HIP_H_ */
