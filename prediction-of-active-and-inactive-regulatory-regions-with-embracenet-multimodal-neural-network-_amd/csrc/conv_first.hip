// The FIRST convolution block of the sequence pre-network (CNN_pre.py:37-44: Conv1d(4 -> C, k) -> BatchNorm1d -> ReLU ->
// MaxPool1d(10, 2) -> Dropout on the one-hot DNA window) without ever materialising the convolution output.
//
// With 4 (padded: 8) input channels the convolution is ~60 MACs per output element, while its output -- [B, 256, C],
// 33.5 MB in bf16 for the A549 shapes -- is the largest tensor of the whole training step and the stored-activation
// path touches it seven times (conv write; BN/pool read; backward: read + dz write, read + read + dy write, wgrad read).
// Here every pass RECOMPUTES the convolution on the matrix cores from the 4 MB input instead:
//   forward   F_STATS   conv -> per-channel sums of z and z^2                       (no activation traffic)
//             (bn_finalize_kernel)
//             F_APPLY   conv -> BN -> ReLU -> LDS tile -> MaxPool/argmax/Dropout -> pooled output (16 MB) + argmax bytes
//   backward  F_BSUMS   conv -> xhat tile in LDS; the two batch sums are taken in WINDOW space -- sum dy = sum of the pooled
//                       gradients that have a target, sum dy*xhat = sum of g[p][c] * xhat[2p + argmax(p,c)][c] -- one LDS
//                       read per pooled element, no dense dy
//             (bn_bwd_finalize_kernel)
//             F_BWGRAD  conv; gather; dz = A*dy + Bc*z + D -> LDS tile -> weight-gradient MFMAs -> per-workgroup slabs
//             (conv_wgrad_reduce_kernel)
// The input gradient does not exist (the one-hot input needs none), which is what makes the block fusable.
// One workgroup tile = SB whole sequences (SB*L <= 256 rows), so pooling windows never cross a tile.  Layout and
// MFMA mapping are those of the transposed streaming kernel (conv_direct.hip): channels on the MFMA M axis, weights in
// registers, lane-group g owns 4*MT consecutive channels of its rows.  bf16 only (the fp32/fp64 paths keep the
// stored-activation kernels).
#include "conv_first.h"
#include "conv_tiles.h"
#include "rider.h"
#include "bn_inline.h"
#include "first_gram.h"
#include "philox.h"

namespace emb {

#ifdef EMB_CONV_PROF
__device__ unsigned long long g_first_prof[64];
__device__ int g_first_sel;
#define FIRST_T(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && g_first_sel == MODE && (i) < 64) g_first_prof[i] = wall_clock64(); } while (0)
#else
#define FIRST_T(i) do {} while (0)
#endif

// n / d for 0 <= n < 2^16, d >= 1 with rd = 1.0f / d: a handful of instructions instead of the ~35 of an integer division by a
// runtime value (the row-offset tables below need dozens of these per thread)
__device__ __forceinline__ int small_div(int n, int d, float rd) {
  int q = (int)((float)n * rd);
  q -= (q * d > n);
  q += ((q + 1) * d <= n);
  return q;
}

constexpr int kFBT = 256, kFXV = 2;            // rows per workgroup tile; activation vectors per thread (tile <= 512 rows)
enum { F_STATS = 0, F_APPLY = 1, F_BSUMS = 2, F_BWGRAD = 3, F_BACC = 4 };   // F_BACC: A = g^T xview only (first_gram.h), no convolution

struct FirstArgs {
  const void* x;            // x_codes == 0: [B][L][8] bf16 channels-last, zero-padded channels
                            // x_codes != 0: [B][L] uint8 base codes (0-3 = A,C,G,T channel; anything else = all-zero column),
                            //               expanded to the one-hot row while staging
                            // x_codes == 2 (F_STATS only): [B][4][L] bf16, the loader's layout; the statistics pass stages
                            //               from it and writes the channels-last image to `nlc_out` for the later passes
  int x_codes;
  __bf16* nlc_out;          // x_codes == 2: [B][L][8]
  const __bf16* w;          // [C][KK] packed weights, KK = k*8
  const float* bias;        // [C]
  const float* stats;       // [4][C] mean, invstd, scale, shift (not read by F_STATS)
  float* partial;           // F_STATS / F_BSUMS: [nblk][2][C]
  __bf16* out;              // F_APPLY: pooled output, [B][Lp][C] or [B][C][Lp]
  uint8_t* argmax;          // F_APPLY out / backward in: [B][Lp][C]
  const __bf16* dout;       // backward: gradient of the pooled output (layout as `out`)
  const float* coef;        // F_BWGRAD: [2][C] mean(dy), mean(dy*xhat)
  float* slab;              // F_BWGRAD: [nblk][C][KK+1]
  const uint64_t* step_dev;
  uint64_t seed, step_val;
  int64_t grow0;
  float drop_p, keep_scale;
  int ncl, training, layer_id;
  int B, L, Lp, KK, C, pad, SB, slot, tiles_m, tpb;
  // Lag statistics of the INPUT (first_gram.h): F_STATS writes one partial G0 row (kGramPart floats) per workgroup to `gram_part`
  // (nullptr: not wanted) and the edge image; the totals jobs (gram_job: parked for the head launch, or -- gram_tot != nullptr --
  // run in F_APPLY's prologue, one per workgroup) turn them into `gram_tot`, which the recompute-free backward's finish reads.
  float* gram_part;
  float* gram_tot;
  __bf16* gram_edge;        // the edge image (first_gram.h): F_STATS writes it, F_APPLY's jobs read it
  int gram_rows;
  BnFinFwd fin_f;           // F_APPLY: partial != nullptr -> the statistics are finalised in this launch's prologue (bn_inline.h)
  BnFinBwd fin_b;           // F_BWGRAD: likewise for coef / dgamma / dbeta
};

__host__ __device__ constexpr int first_plo(int t) { return t >= 9 ? (t - 8) / 2 : 0; }   // first pooling window containing t

// MT channel tiles (16 channels each) per wave, CH channel groups per workgroup: C = 16*MT*CH; 4*CH waves, wave = (row
// quarter, channel group).  Splitting the channels over wave pairs halves the per-wave register state (weights, constants,
// accumulators) so that 8 waves fit on a CU.
template <int MT, int CH, int MODE>
__device__ __forceinline__ void first_body(const FirstArgs& a, const int bm) {
  using Mm = Mma<__bf16>;
  using T = __bf16;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  constexpr int CPL = 4 * MT, GW = 16 * MT, BN = GW * CH, XS = 8, NW = 4 * CH, NTHR = 256 * CH;
  constexpr bool BW = MODE == F_BWGRAD || MODE == F_BACC;   // passes that build a dense gradient tile and run the weight-gradient MFMAs
  constexpr int ZP = BN + 8;                            // pitch of the row-major bf16 tile (BN+ReLU output / dz)
  extern __shared__ __attribute__((aligned(16))) char arena[];
  const int L = a.L, Lp = a.Lp, C = a.C, SB = a.SB, slot = a.slot, KK = a.KK;
  const int xrows = SB * slot + kXExtra;
  // LDS carve
  constexpr int DPP = BN + 8, AMP = BN + 8;             // pitches of the pooled-gradient rows (bf16) / argmax rows (bytes)
  T* xs = reinterpret_cast<T*>(arena);                                 // [xrows][8]
  T* zt = xs + ((xrows * XS + 7) & ~7);                                // F_APPLY: BN/ReLU output, F_BWGRAD: dz; [256][ZP] bf16
  constexpr int XP = BN + 4;                                           // F_BSUMS: fp32 xhat tile [256][XP] in the zt region
  float* xt = reinterpret_cast<float*>(zt);
  constexpr int FP = BN + 4;                                           // F_BACC: fp32 scatter tile [256][FP] behind the bf16 tile
  float* ft = reinterpret_cast<float*>(zt + kFBT * ZP);
  T* dp = zt + ((MODE == F_APPLY || BW) ? kFBT * ZP : (MODE == F_BSUMS ? 2 * kFBT * XP : 0)) + (MODE == F_BACC ? 2 * kFBT * FP : 0);   // backward: pooled gradient [SB*Lp][DPP]
  uint8_t* am = reinterpret_cast<uint8_t*>(dp + (MODE >= F_BSUMS ? SB * Lp * DPP : 0));   // backward: argmax bytes [SB*Lp][AMP]
  float* red = reinterpret_cast<float*>(am + (MODE >= F_BSUMS ? ((SB * Lp * AMP + 15) & ~15) : 0));   // [4][2][BN]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r16 = lane & 15;
  const int grp = wave % CH, rq = wave / CH, gcol = grp * GW;   // channel group, row quarter, first channel of the group
  const int tm_begin = bm * a.tpb, tm_end = min(a.tiles_m, tm_begin + a.tpb);

  FIRST_T(0);
  // ---- this thread's share of the activation tile (tile independent)
  int x_lds[kFXV], x_glb[kFXV], x_pk[kFXV];
#pragma unroll
  for (int i = 0; i < kFXV; ++i) {
    const int row = threadIdx.x + i * NTHR, s = row / slot, dtp = row - s * slot;
    x_lds[i] = row * XS;
    x_glb[i] = (s * L + dtp - a.pad) * XS;
    x_pk[i] = row < xrows ? (((s < SB ? s : 0x7fff) << 16) | dtp) : -1;
  }
  bf16x8 xr[kFXV];
  auto issue_x = [&](int tm) {
    const int b0 = tm * SB;
    if (MODE == F_STATS && a.x_codes == 2) {   // loader layout [b][ch][t]: four 2-byte loads per row, lanes walk t (coalesced)
      const T* xb = reinterpret_cast<const T*>(a.x) + (long)b0 * 4 * L;
#pragma unroll
      for (int i = 0; i < kFXV; ++i) {
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (T)0.0f;
        if (x_pk[i] >= 0) {
          const int s = x_pk[i] >> 16, tt = (x_pk[i] & 0xffff) - a.pad;
          if (b0 + s < a.B && tt >= 0 && tt < L) {
            const T* xp = xb + (long)s * 4 * L + tt;
            v[0] = xp[0]; v[1] = xp[L]; v[2] = xp[2 * L]; v[3] = xp[3 * L];
          }
        }
        xr[i] = v;
      }
    } else if (!a.x_codes) {
      const T* xb = reinterpret_cast<const T*>(a.x) + (long)b0 * L * XS;
#pragma unroll
      for (int i = 0; i < kFXV; ++i) {
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (T)0.0f;
        if (x_pk[i] >= 0) {
          const int s = x_pk[i] >> 16, tt = (x_pk[i] & 0xffff) - a.pad;
          if (b0 + s < a.B && tt >= 0 && tt < L) v = *reinterpret_cast<const bf16x8*>(xb + x_glb[i]);
        }
        xr[i] = v;
      }
    } else {   // one byte per position; the one-hot row is formed when the tile is written to LDS
      const uint8_t* cb = reinterpret_cast<const uint8_t*>(a.x) + (long)b0 * L;
#pragma unroll
      for (int i = 0; i < kFXV; ++i) {
        uint32_t code = 0xFFu;
        if (x_pk[i] >= 0) {
          const int s = x_pk[i] >> 16, tt = (x_pk[i] & 0xffff) - a.pad;
          if (b0 + s < a.B && tt >= 0 && tt < L) code = cb[(long)s * L + tt];
        }
        uint4 w = make_uint4(code, 0u, 0u, 0u);
        xr[i] = __builtin_bit_cast(bf16x8, w);
      }
    }
  };
  if (tm_begin < tm_end) issue_x(tm_begin);

  // ---- weights (A operand, channel-permuted rows) and per-channel constants of this lane's CPL channels
  // The tile keeps 8 channels per position (4 real + 4 zero: 16-byte rows) but the convolution contracts only the real ones: a
  // k-step of 32 = EIGHT taps x 4 channels (lane group g: taps 8*ks + 2g and + 1, one 8-byte read each), half the MFMAs and LDS
  // bytes of four-taps-times-eight.  The weight fragments are gathered to match from the packed [C][k][8] rows.
  constexpr int NKS = 2;                                 // k <= 15 -> 4k <= 64
  const int nks = (4 * a.KK / 8 + 31) / 32;
  bf16x8 wf[MT][NKS];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int ch = gcol + chan_of<T, MT>(mt, r16), t0 = ks * 8 + 2 * g;
      bf16x4 lo = {(T)0.0f, (T)0.0f, (T)0.0f, (T)0.0f}, hi = lo;
      if (ch < C && t0 * 8 < KK) lo = *reinterpret_cast<const bf16x4*>(a.w + (long)ch * KK + t0 * 8);
      if (ch < C && (t0 + 1) * 8 < KK) hi = *reinterpret_cast<const bf16x4*>(a.w + (long)ch * KK + (t0 + 1) * 8);
      bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      wf[mt][ks] = f;
    }
  // ---- BatchNorm vectors: from global memory, or finalised here from the producer's partial rows (bn_inline.h)
  const float* st_src = a.stats;
  const float* cf_src = a.coef;
  if (MODE == F_APPLY || MODE == F_BWGRAD) {
    double* fin_scratch = reinterpret_cast<double*>(arena);                  // the arena is not in use before the first tile
    float* fin_out = reinterpret_cast<float*>(fin_scratch + NTHR * 4 + 2 * BN);   // [4][BN] floats
    GramPre gpre;
    gpre.have = false;
    if (MODE == F_APPLY && a.gram_tot != nullptr && bm < kGramJobs) gpre = gram_job_preload<NTHR>(bm, a.gram_edge, a.B, L);   // (loads in flight below)
    if (MODE == F_APPLY && a.fin_f.partial != nullptr) {
      bn_fin_fwd<NTHR>(a.fin_f, C, fin_scratch, fin_out, bm == 0);
      st_src = fin_out;
    }
    if (MODE == F_APPLY && a.gram_tot != nullptr) {   // totals of the lag statistics, one job per workgroup (first_gram.h)
      for (int job = bm; job < kGramJobs; job += (int)gridDim.x) {
        gram_job<NTHR>(job, gpre, a.gram_edge, a.B, L, a.gram_part, a.gram_rows, NW / 4, a.gram_tot, reinterpret_cast<float*>(fin_scratch));
        gpre.have = false;
      }
    }
    if (MODE == F_BWGRAD && a.fin_b.partial != nullptr) {
      bn_fin_bwd<NTHR>(a.fin_b, C, fin_scratch, fin_out, bm == 0);
      cf_src = fin_out;
    }
  }
  // per-channel constants with the conv bias folded in (acc = convolution without bias):
  //   F_STATS   z = acc + k0                                  k0 = bias
  //   F_APPLY   bn(z) = acc*k0 + k1                           k0 = scale, k1 = bias*scale + shift
  //   F_BSUMS   xhat = (acc - k0)*k1                          k0 = mean - bias, k1 = invstd
  //   F_BWGRAD  dz = k0*dy + k1*acc + k2                      bn_bwd_affine_kernel's A, Bc, D with z = acc + bias
  float k0[MT][4], k1[MT][4], k2[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = min(gcol + g * CPL + mt * 4 + r, C - 1);
      const float bias = a.bias[ch];
      k0[mt][r] = 0; k1[mt][r] = 0; k2[mt][r] = 0;
      if (MODE == F_STATS) {
        k0[mt][r] = bias;
      } else if (MODE == F_APPLY) {
        k0[mt][r] = st_src[2 * C + ch];
        k1[mt][r] = bias * k0[mt][r] + st_src[3 * C + ch];
      } else if (MODE == F_BSUMS) {
        k0[mt][r] = a.stats[ch] - bias;
        k1[mt][r] = a.stats[C + ch];
      } else if (MODE == F_BACC) {
      } else {
        const float mean = a.stats[ch], inv = a.stats[C + ch], sc = a.stats[2 * C + ch];
        const float bc = a.training ? -sc * cf_src[C + ch] * inv : 0.0f;
        const float d = a.training ? sc * (cf_src[C + ch] * inv * mean - cf_src[ch]) : 0.0f;
        k0[mt][r] = sc;
        k1[mt][r] = bc;
        k2[mt][r] = bc * bias + d;
      }
    }
  float s1[MT][4], s2[MT][4];   // F_STATS: sum z, z^2;  F_BSUMS: sum dy, dy*xhat
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[mt][r] = 0; s2[mt][r] = 0; }

  // ---- this lane's output rows
  int xrow[4], row_sq[4], row_t[4];      // LDS row of tap 0; sequence slot; time (-1: row past the tile's sequences)
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int row = rq * 64 + nt * 16 + r16;
    const int rr = min(row, SB * L - 1), sq = rr / L;
    xrow[nt] = sq * slot + (rr - sq * L);
    row_sq[nt] = sq;
    row_t[nt] = row < SB * L ? rr - sq * L : -1;
  }
  // LDS element offset of tile row `row` (k-step rows of the weight-gradient / lag-statistics MFMAs): computed in the loops, cheap
  // with small_div -- a table in registers costs occupancy, a table in LDS serialises map read -> fragment read -> MFMA
  const float rL = 1.0f / (float)L;
  auto row_off = [&](int row) {
    const int rr = min(row, SB * L - 1), sq = small_div(rr, L, rL);
    return (sq * slot + (rr - sq * L)) * XS;
  };
  // lag statistics (first_gram.h): MFMA accumulator of G0 = x~[r - pad]^T . xview (this wave's column blocks), accumulated over the
  // workgroup's tiles
  typename Mm::AccV accg = {0, 0, 0, 0};
  const bool gram = MODE == F_STATS && a.gram_part != nullptr;
  // weight-gradient accumulators: all MIW channel tiles x this wave's n-blocks (16 columns of k*8) nb = ni*NW + wave
  constexpr int MIW = MT * CH, NIW = 8 / NW;
  constexpr int kBaccNB = 4, MIWB = MIW / (NW / kBaccNB);   // F_BACC: four compact column blocks, the waves of a block share the channel tiles
  typename Mm::AccV accw[BW ? MIW : 1][NIW];
  if (BW) {
#pragma unroll
    for (int mi = 0; mi < MIW; ++mi)
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) accw[mi][ni][r] = 0;
  }
  const int q4 = r16 >> 2, p4 = r16 & 3;
  // ---- this thread's pooled-tensor items (window p of sequence slot sq, 8 channels from c0): tile independent, so the
  // index arithmetic (runtime divisions) happens once per kernel, not per tile and phase
  constexpr int NIT = 1024 / NTHR, CV = BN / 8;        // SB * Lp * C / 8 <= 1024 items per tile
  int it_row[NIT], it_pc[NIT];                         // tile row of the window start | (c0 << 16);  p | sq << 8; -1 none
  if (MODE != F_STATS) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int it = threadIdx.x + i * NTHR, c0 = (it % CV) * 8;
      int sp = it / CV, sq = sp / Lp, p = sp - sq * Lp;
      if (MODE == F_BACC) {
        // the scatter below works in five rounds by p % 5: the windows are dealt to the threads SORTED by that phase (then sequence,
        // then p), so that the 64 / CV windows of a wave share a phase and a wave runs the scatter code once, not once per round
        int w = sp, ph = 0;
        for (; ph < 5; ++ph) {
          const int n = SB * ((Lp - ph + 4) / 5);
          if (w < n) break;
          w -= n;
        }
        const int nq = ph < 5 ? (Lp - ph + 4) / 5 : 1;
        sq = ph < 5 ? w / nq : SB;
        p = ph < 5 ? 5 * (w - sq * nq) + ph : 0;
        sp = sq * Lp + p;
      }
      it_row[i] = MODE == F_APPLY ? ((sq * L + 2 * p) | (c0 << 16)) : (sp | (c0 << 16));   // backward: LDS row sp of dp / am
      it_pc[i] = sq < SB ? (p | (sq << 8)) : -1;
    }
  }
  // F_BACC: the pooled gradient is SCATTERED into a dense tile: window p of a sequence gives its gradient to row 2p + argmax.
  // Windows p and p + 5 share no row (10 rows, stride 2), so the items with p % 5 == ph touch distinct (row, channel) cells: five
  // rounds of plain read-modify-write, no atomics, a fixed order of additions.
  int it_ph[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) it_ph[i] = MODE == F_BACC && it_pc[i] >= 0 ? (it_pc[i] & 0xFF) % 5 : -1;
  // backward: the tile's pooled gradient and argmax bytes travel global -> registers -> LDS like the activations
  bf16x8 gv[MODE >= F_BSUMS ? NIT : 1];
  uint64_t av[MODE >= F_BSUMS ? NIT : 1];
  auto issue_g = [&](int tm) {
    const int b0 = tm * SB, nseq = min(SB, a.B - b0);
    const T* gsrc = a.dout + (long)b0 * Lp * C;
    const uint8_t* asrc = a.argmax + (long)b0 * Lp * C;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      gv[i] = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));   // (whole-vector write: element writes would push the array to scratch)
      av[i] = 0x8080808080808080ull;                   // "dropped": matches no window offset
      if (it_pc[i] >= 0 && (it_pc[i] >> 8) < nseq) {
        const long off = MODE == F_BACC ? (long)(it_row[i] & 0xFFFF) * C + (it_row[i] >> 16)   // (phase-sorted items)
                                        : (long)(threadIdx.x + i * NTHR) * 8;                   // items are laid out exactly as [sq][p][c]
        gv[i] = *reinterpret_cast<const bf16x8*>(gsrc + off);
        av[i] = *reinterpret_cast<const uint64_t*>(asrc + off);
      }
    }
  };
  if (MODE >= F_BSUMS) {
    if (!a.ncl && tm_begin < tm_end) issue_g(tm_begin);
  }

  float q1[8], q2[8];   // F_BSUMS: this thread's sums of g and g*xhat for its 8 channels (items of a thread share c0)
#pragma unroll
  for (int e = 0; e < 8; ++e) { q1[e] = 0.0f; q2[e] = 0.0f; }

  FIRST_T(1);
  for (int tm = tm_begin; tm < tm_end; ++tm) {
    const int b0 = tm * SB, nseq = min(SB, a.B - b0);
    __syncthreads();                                   // previous tile's LDS contents are consumed
    FIRST_T(2 + (tm - tm_begin) * 8 + 0);
#pragma unroll
    for (int i = 0; i < kFXV; ++i)
      if (x_pk[i] >= 0) {
        bf16x8 v = xr[i];
        if (MODE == F_STATS && a.x_codes == 2) {   // every real row is staged by exactly one tile: write the channels-last image
          const int s = x_pk[i] >> 16, tt = (x_pk[i] & 0xffff) - a.pad;
          if (b0 + s < a.B && tt >= 0 && tt < L) *reinterpret_cast<bf16x8*>(a.nlc_out + ((long)(b0 + s) * L + tt) * 8) = v;
        } else if (a.x_codes) {   // bf16 1.0 = 0x3F80 in the selected channel (two channels per 32-bit word)
          const uint32_t code = __builtin_bit_cast(uint4, xr[i]).x;
          const uint4 w = make_uint4(code == 0 ? 0x00003F80u : code == 1 ? 0x3F800000u : 0u,
                                     code == 2 ? 0x00003F80u : code == 3 ? 0x3F800000u : 0u, 0u, 0u);
          v = __builtin_bit_cast(bf16x8, w);
        }
        *reinterpret_cast<bf16x8*>(xs + x_lds[i]) = v;
      }
    if (MODE >= F_BSUMS) {
      if (!a.ncl) {
#pragma unroll
        for (int i = 0; i < NIT; ++i)
          if (MODE == F_BWGRAD && it_pc[i] >= 0) {   // F_BSUMS / F_BACC consume the registers directly
            const int sp = it_row[i] & 0xFFFF, c0 = it_row[i] >> 16;
            *reinterpret_cast<bf16x8*>(dp + sp * DPP + c0) = gv[i];
            *reinterpret_cast<uint64_t*>(am + sp * AMP + c0) = av[i];
          }
      } else {                                          // dout[b][c][p] (single-block stack): element-wise transpose
        for (int i = threadIdx.x; i < SB * C * Lp; i += NTHR) {
          const int sq = i / (C * Lp), rem = i - sq * C * Lp, c = rem / Lp, p = rem - c * Lp;
          const bool v = sq < nseq;
          dp[(sq * Lp + p) * DPP + c] = v ? a.dout[((long)(b0 + sq) * C + c) * Lp + p] : (T)0.0f;
          am[(sq * Lp + p) * AMP + c] = v ? a.argmax[((long)(b0 + sq) * Lp + p) * C + c] : (uint8_t)0x80;
        }
      }
    }
    if (MODE == F_BACC) {                                // zero the scatter tile: 256 * FP floats, 16 bytes per store
      const float4 z4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      for (int i = threadIdx.x; i < kFBT * FP / 4; i += NTHR) reinterpret_cast<float4*>(ft)[i] = z4;
    }
    __syncthreads();
    FIRST_T(2 + (tm - tm_begin) * 8 + 2);
    if (MODE == F_BACC) {
      // this tile's items move to a second register set and the next tile's loads start now: a whole tile of time to land
      bf16x8 gcur[NIT];
      uint64_t acur[NIT];
#pragma unroll
      for (int i = 0; i < NIT; ++i) { gcur[i] = gv[i]; acur[i] = av[i]; }
      if (tm + 1 < tm_end) {
        issue_x(tm + 1);
        if (!a.ncl) issue_g(tm + 1);
      }
      FIRST_T(2 + (tm - tm_begin) * 8 + 6);
#pragma unroll 1
      for (int ph = 0; ph < 5; ++ph) {
#pragma unroll
        for (int i = 0; i < NIT; ++i)
          if (it_ph[i] == ph && (it_pc[i] >> 8) < nseq) {
            const int sp = it_row[i] & 0xFFFF, c0 = it_row[i] >> 16, sq = it_pc[i] >> 8, p = it_pc[i] & 0xFF;
            bf16x8 gq;
            uint64_t aq;
            if (!a.ncl) {
              gq = gcur[i];
              aq = acur[i];
            } else {
              gq = *reinterpret_cast<const bf16x8*>(dp + sp * DPP + c0);
              aq = *reinterpret_cast<const uint64_t*>(am + sp * AMP + c0);
            }
            // the eight channels are eight different columns: all reads first, then the writes (one LDS round trip, not eight).
            // Channels without a gradient (bit 6 / 7 of the code: ReLU zero / dropped) add into the row's pad columns, which
            // nobody reads: no branches
            // Inside its group of eight, channel e of tile row r sits at position (e + r) & 7: with the natural order the 64 lanes
            // of one of these accesses (rows 2p + code, groups c0, ONE e) all fall on the 16 banks = e (mod 4) -- the row pitch
            // and c0 are multiples of 4 -- and the scatter ran 4-way conflicted; rotated by the row they spread over all banks
            const int r0 = sq * L + 2 * p;
            float* cell = ft + r0 * FP + c0;
            float cur[8];
            int off[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int code = (int)((aq >> (8 * e)) & 0xFF);
              off[e] = code < 0x40 ? code * FP + ((e + r0 + code) & 7) : BN - c0 + (e & 3);
              cur[e] = cell[off[e]];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) cell[off[e]] = cur[e] + (float)gq[e];
          }
        __syncthreads();
      }
      FIRST_T(2 + (tm - tm_begin) * 8 + 5);
      // fp32 -> the bf16 A-operand tile: thread (row, 32-channel half)
      for (int i = threadIdx.x; i < kFBT * (BN / 8); i += NTHR) {
        const int row = i / (BN / 8), c8 = (i - row * (BN / 8)) * 8;
        const float* grp = ft + row * FP + c8;             // (the group's channels are rotated by the row, see the scatter)
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (T)(grp[(e + row) & 7] * a.keep_scale);
        *reinterpret_cast<bf16x8*>(zt + row * ZP + c8) = o;
      }
    }
    if (MODE != F_BACC && tm + 1 < tm_end) {
      issue_x(tm + 1);
      if (BW) {
        if (!a.ncl) issue_g(tm + 1);
      }
    }
    if (MODE == F_STATS) {   // (after the next tile's loads are in flight: this phase hides their latency)
      if (gram) {
        gram_edge_store(xs, a.gram_edge, a.B, L, slot, a.pad, b0, nseq);
        if (SB == 1) gram_tile<NW, 4>(xs, [&](int row) { return min(row, L - 1) * XS; }, accg, L, SB, slot, lane, wave);   // one sequence per tile: no division
        else gram_tile<NW, 2>(xs, row_off, accg, L, SB, slot, lane, wave);
      }
    }
    FIRST_T(2 + (tm - tm_begin) * 8 + 1);

#pragma unroll 1
    for (int h = 0; h < (MODE == F_BACC ? 0 : 2); ++h) {   // two row tiles at a time (accumulator registers); a real loop: the
                                                       // unrolled form interleaves the halves and doubles the live registers
      const int xrow_h[2] = {h ? xrow[2] : xrow[0], h ? xrow[3] : xrow[1]};
      const int sq_h[2] = {h ? row_sq[2] : row_sq[0], h ? row_sq[3] : row_sq[1]};
      const int t_h[2] = {h ? row_t[2] : row_t[0], h ? row_t[3] : row_t[1]};
      typename Mm::AccV acc[MT][2];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mt][j][r] = 0;
#pragma unroll
      for (int ks = 0; ks < (MODE == F_BACC ? 0 : NKS); ++ks) {   // (F_BACC needs no convolution)
        if (ks < nks) {
          bf16x8 bf[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {                // k-step ks, lane group g: the four real channels of taps 8*ks + 2g and + 1
            const T* xp = xs + (xrow_h[j] + 8 * ks + 2 * g) * XS;
            const bf16x4 lo = *reinterpret_cast<const bf16x4*>(xp), hi = *reinterpret_cast<const bf16x4*>(xp + XS);
            bf16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            bf[j] = v;
          }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[mt][j] = Mm::mma(wf[mt][ks], bf[j], acc[mt][j]);
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int nt = 2 * h + j, sq = sq_h[j], t = t_h[j];
        const bool rv = t >= 0 && sq < nseq;
        const int row = rq * 64 + nt * 16 + r16;
        if (MODE == F_STATS) {
          if (rv) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float z = acc[mt][j][r] + k0[mt][r];
                s1[mt][r] += z;
                s2[mt][r] += z * z;
              }
          }
        } else if (MODE == F_APPLY) {
          T ov[CPL];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float z = acc[mt][j][r] * k0[mt][r] + k1[mt][r];
              ov[mt * 4 + r] = (T)(z > 0.0f ? z : 0.0f);
            }
          T* dst = zt + row * ZP + gcol + g * CPL;
          if (CPL >= 8) {
#pragma unroll
            for (int qv = 0; qv < CPL / 8; ++qv) {
              bf16x8 o;
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = ov[qv * 8 + e];
              *reinterpret_cast<bf16x8*>(dst + qv * 8) = o;
            }
          } else {
            bf16x4 o = {ov[0], ov[1], ov[2], ov[3]};
            *reinterpret_cast<bf16x4*>(dst) = o;
          }
        } else if (MODE == F_BSUMS) {
          float* dst = xt + row * XP + gcol + g * CPL;   // xhat of this lane's row (0 outside the batch)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            float4 o;
            o.x = rv ? (acc[mt][j][0] - k0[mt][0]) * k1[mt][0] : 0.0f;
            o.y = rv ? (acc[mt][j][1] - k0[mt][1]) * k1[mt][1] : 0.0f;
            o.z = rv ? (acc[mt][j][2] - k0[mt][2]) * k1[mt][2] : 0.0f;
            o.w = rv ? (acc[mt][j][3] - k0[mt][3]) * k1[mt][3] : 0.0f;
            *reinterpret_cast<float4*>(dst + mt * 4) = o;
          }
        } else {
          // dy[ch] of this lane's row t: the windows containing t are p = t/2 - k, k = 0..4, and window p selected row t iff
          // its argmax byte equals t - 2p = (t & 1) + 2k (ReLU zeros / dropped elements carry bit 6 / 7: never equal)
          float dy[CPL];
#pragma unroll
          for (int e = 0; e < CPL; ++e) dy[e] = 0.0f;
          {
            const int tt = rv ? t : 0, par = tt & 1, pb = tt >> 1;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
              const int pw = pb - k;
              const bool pv = rv && pw >= 0 && pw < Lp;
              const uint32_t want = pv ? (uint32_t)(par + 2 * k) : 0xFFu;
              const int sp = sq * Lp + (pv ? pw : 0);
              const T* gp = dp + sp * DPP + gcol + g * CPL;
              const uint8_t* ap = am + sp * AMP + gcol + g * CPL;
#pragma unroll
              for (int qv = 0; qv < CPL / 4; ++qv) {
                const bf16x4 gq = *reinterpret_cast<const bf16x4*>(gp + qv * 4);
                const uint32_t aq = *reinterpret_cast<const uint32_t*>(ap + qv * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  dy[qv * 4 + e] += ((aq >> (8 * e)) & 0xFFu) == want ? (float)gq[e] : 0.0f;
              }
            }
#pragma unroll
            for (int e = 0; e < CPL; ++e) dy[e] *= a.keep_scale;
          }
          {   // F_BWGRAD
            T ov[CPL];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float dz = MODE == F_BACC ? dy[mt * 4 + r] : k0[mt][r] * dy[mt * 4 + r] + k1[mt][r] * acc[mt][j][r] + k2[mt][r];
                ov[mt * 4 + r] = (T)(rv ? dz : 0.0f);
              }
            T* dst = zt + row * ZP + gcol + g * CPL;
            if (CPL >= 8) {
#pragma unroll
              for (int qv = 0; qv < CPL / 8; ++qv) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = ov[qv * 8 + e];
                *reinterpret_cast<bf16x8*>(dst + qv * 8) = o;
              }
            } else {
              bf16x4 o = {ov[0], ov[1], ov[2], ov[3]};
              *reinterpret_cast<bf16x4*>(dst) = o;
            }
          }
        }
      }
    }

    FIRST_T(2 + (tm - tm_begin) * 8 + 3);
    if (MODE == F_BSUMS) {
      __syncthreads();                                 // xhat tile complete
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        if (it_pc[i] >= 0 && (it_pc[i] >> 8) < nseq) {
          const int sp = it_row[i] & 0xFFFF, c0 = it_row[i] >> 16, sq = it_pc[i] >> 8, p = it_pc[i] & 0xFF;
          bf16x8 gq;
          uint64_t aq;
          if (!a.ncl) {
            gq = gv[i];
            aq = av[i];
          } else {
            gq = *reinterpret_cast<const bf16x8*>(dp + sp * DPP + c0);
            aq = *reinterpret_cast<const uint64_t*>(am + sp * AMP + c0);
          }
          const float* xrow0 = xt + (sq * L + 2 * p) * XP + c0;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int code = (int)((aq >> (8 * e)) & 0xFF);
            const bool ok = code < 0x40;                 // bit 6 / 7: ReLU zero / dropped -> no gradient
            const float gval = ok ? (float)gq[e] * a.keep_scale : 0.0f;
            const float xh = xrow0[(ok ? code : 0) * XP + e];
            q1[e] += gval;
            q2[e] += gval * xh;
          }
        }
      }
      if (tm + 1 < tm_end) {
        if (!a.ncl) issue_g(tm + 1);                   // the registers are free again
      }
    }
    if (MODE == F_APPLY) {
      __syncthreads();
      // MaxPool(10, 2) + argmax + Dropout over the tile's sequences; thread = (sequence, window, 8 channels)
      float keep_scale = 1.0f;
      uint64_t stream = 0;
      if (a.drop_p > 0.0f) {
        keep_scale = 1.0f / (1.0f - a.drop_p);
        stream = rng_stream(a.step_val + (a.step_dev ? *a.step_dev : 0), EMB_RNG_DROPOUT0 + a.layer_id);
      }
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        if (it_pc[i] < 0 || ((it_pc[i] >> 8) & 0xFF) >= nseq) continue;
        const int c0 = it_row[i] >> 16, p = it_pc[i] & 0xFF, b = b0 + ((it_pc[i] >> 8) & 0xFF);
        const T* base = zt + (it_row[i] & 0xFFFF) * ZP + c0;
        // max and argmax in ONE unsigned max per element and row: post-ReLU bf16 values are >= 0, so their bit patterns
        // order like the values; widened to the high half of a word, the low bits carry 9 - row (ties keep the first row,
        // as torch does)
        uint32_t key[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) key[e] = 0;
#pragma unroll
        for (int wdw = 0; wdw < 10; ++wdw) {
          const uint4 v = *reinterpret_cast<const uint4*>(base + wdw * ZP);
          const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            key[2 * q] = max(key[2 * q], (wd[q] << 16) | (uint32_t)(9 - wdw));
            key[2 * q + 1] = max(key[2 * q + 1], (wd[q] & 0xffff0000u) | (uint32_t)(9 - wdw));
          }
        }
        float best[8];
        int arg[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          best[e] = __uint_as_float(key[e] & 0xffff0000u);
          arg[e] = 9 - (int)(key[e] & 0xFu);
        }
        uint64_t amv = 0;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float z = best[e];
          int code = arg[e] | (z > 0.0f ? 0 : 0x40);          // bit 6: ReLU output is zero -> no gradient
          if (a.drop_p > 0.0f) {
            const uint64_t idx = ((uint64_t)(a.grow0 + b) * Lp + p) * C + c0 + e;
            const bool keep = uniform24(philox4x32_10(a.seed, stream, idx).x) >= a.drop_p;
            z = keep ? z * keep_scale : 0.0f;
            code |= keep ? 0 : 0x80;
          }
          o[e] = (T)z;
          amv |= (uint64_t)(uint8_t)code << (8 * e);
        }
        const long nlc = ((long)b * Lp + p) * C + c0;
        *reinterpret_cast<uint64_t*>(a.argmax + nlc) = amv;
        if (a.ncl) {
#pragma unroll
          for (int e = 0; e < 8; ++e) a.out[((long)b * C + c0 + e) * Lp + p] = o[e];
        } else {
          *reinterpret_cast<bf16x8*>(a.out + nlc) = o;
        }
      }
    }

    FIRST_T(2 + (tm - tm_begin) * 8 + 4);
    if (MODE == F_BACC) {
      __syncthreads();
      // A[o][n'] += sum_r g[r][o] * xview[r][n'] over the REAL input channels only: column n' = tap * 4 + ci (the transposing read
      // of a lane takes the four real channels of tap nb * 4 + p4), 4 k <= 64 columns = four 16-column blocks instead of eight.
      // Wave = (block nb, half mh of the channel tiles).  Column 4 k is forced to ones: sum_r g = dbeta.
      const int nbb = (4 * a.KK / 8) >> 4, lane_b = (4 * a.KK / 8) & 15;
      const int nb = wave % kBaccNB, mh = wave / kBaccNB, xoff = (nb * 4 + p4) * XS;
#pragma unroll 2
      for (int ks = 0; ks < kFBT / 32; ++ks) {
        const int ra = ks * 32 + 8 * g + q4;
        bf16x8 af[MIWB];
#pragma unroll
        for (int mi = 0; mi < MIWB; ++mi) {
          const T* a0 = zt + ra * ZP + (mh * MIWB + mi) * 16 + 4 * p4;
          union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
          u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
          u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * ZP));
          af[mi] = u.v;
        }
        const int x0 = row_off(ra), x1 = row_off(ra + 4);
        union { struct { s16x4 lo, hi; } s; bf16x8 v; } ub;
        ub.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xs + x0 + xoff));
        ub.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xs + x1 + xoff));
        if (nb == nbb && r16 == lane_b) {
#pragma unroll
          for (int e = 0; e < 8; ++e) ub.v[e] = (T)1.0f;
        }
#pragma unroll
        for (int mi = 0; mi < MIWB; ++mi) accw[mi][0] = Mm::mma(af[mi], ub.v, accw[mi][0]);
      }
    }
    if (MODE == F_BWGRAD) {
      __syncthreads();
      // dW[o][n] += sum_r dz[r][o] * xview[r][n]:  A = dz^T (tr16 reads of the row-major tile), B = x view, K = 256 rows.
      // column KK of the B operand is forced to ones: slab column KK = sum_r dz = bias gradient
      const int nb_bias = KK >> 4, lane_bias = KK & 15;
#pragma unroll 2
      for (int ks = 0; ks < kFBT / 32; ++ks) {
        const int ra = ks * 32 + 8 * g + q4;
        bf16x8 af[MIW], bfw[NIW];
#pragma unroll
        for (int mi = 0; mi < MIW; ++mi) {
          const T* a0 = zt + ra * ZP + mi * 16 + 4 * p4;
          union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
          u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
          u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * ZP));
          af[mi] = u.v;
        }
        const int x0 = row_off(ra), x1 = row_off(ra + 4);
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) {
          const int nb = ni * NW + wave, xoff = nb * 16 + 4 * p4;   // column n = tap*8 + ci = LDS offset (pitch 8)
          union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
          u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xs + x0 + xoff));
          u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xs + x1 + xoff));
          bfw[ni] = u.v;
          if (nb == nb_bias && r16 == lane_bias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) bfw[ni][e] = (T)1.0f;
          }
        }
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
          for (int mi = 0; mi < MIW; ++mi) accw[mi][ni] = Mm::mma(af[mi], bfw[ni], accw[mi][ni]);
      }
    }
  }

  FIRST_T(40);
  if (MODE == F_BSUMS) {   // one partial row per workgroup: threads tid = c8 + CV*m share the channels 8*c8 .. 8*c8+7
    __syncthreads();
    // [16][NTHR + 8]: value k (q1[0..7], q2[0..7]) of every thread in one row -- consecutive lanes store consecutive words,
    // and the 16 * CV readers (k, c8) walk their row with lanes spread over the banks by the +8 pitch.  (The former
    // [NTHR][16] layout put the 32 lanes of a store on two banks: 16-way conflicts, 16 stores per thread.)
    float* red2 = xt;
    constexpr int RP2 = NTHR + 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red2[e * RP2 + threadIdx.x] = q1[e];
      red2[(8 + e) * RP2 + threadIdx.x] = q2[e];
    }
    __syncthreads();
    if (threadIdx.x < 16 * CV) {
      const int k = threadIdx.x / CV, c8 = threadIdx.x % CV, which = k >> 3, c = c8 * 8 + (k & 7);
      float t = 0.0f;
      for (int m = 0; m < NTHR / CV; ++m) t += red2[k * RP2 + c8 + CV * m];
      if (c < C) a.partial[((long)bm * 2 + which) * C + c] = t;
    }
  }
  if (MODE == F_STATS) {
    if (gram) gram_store(a.gram_part + (long)bm * kGramPart, accg, lane, wave);
  }
  FIRST_T(42);
  if (MODE == F_STATS) {   // one partial row per workgroup
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sa = row16_sum<float>(s1[mt][r]), sb = row16_sum<float>(s2[mt][r]);
        if (r16 == 0) {
          red[(rq * 2 + 0) * BN + gcol + g * CPL + mt * 4 + r] = sa;
          red[(rq * 2 + 1) * BN + gcol + g * CPL + mt * 4 + r] = sb;
        }
      }
    __syncthreads();
    if (threadIdx.x < 2 * BN) {   // the four row quarters, in order
      const int c = threadIdx.x % BN, which = threadIdx.x / BN;
      if (c < C)
        a.partial[((long)bm * 2 + which) * C + c] =
            ((red[(0 * 2 + which) * BN + c] + red[(1 * 2 + which) * BN + c]) + red[(2 * 2 + which) * BN + c]) + red[(3 * 2 + which) * BN + c];
    }
  }
  if (MODE == F_BACC) {
    float* dst = a.slab + (long)bm * C * kFinCols;      // [C][64]: the compact columns as they are (first_fin.h)
    const int nb = wave % kBaccNB, mh = wave / kBaccNB, np = nb * 16 + r16;
#pragma unroll
    for (int mi = 0; mi < MIWB; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = (mh * MIWB + mi) * 16 + Mm::acc_row(lane, r);
        if (o < C) dst[(long)o * kFinCols + np] = accw[mi][0][r];
      }
  }
  if (MODE == F_BWGRAD) {
    float* dst = a.slab + (long)bm * C * (KK + 1);
#pragma unroll
    for (int mi = 0; mi < MIW; ++mi)
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = mi * 16 + Mm::acc_row(lane, r), n = (ni * NW + wave) * 16 + r16;
          if (o < C && n <= KK) dst[(long)o * (KK + 1) + n] = accw[mi][ni][r];
        }
  }
  FIRST_T(41);
}

template <int MT, int CH, int MODE>
__global__ __launch_bounds__(256 * CH) void first_kernel(const FirstArgs a) {
  first_body<MT, CH, MODE>(a, (int)blockIdx.x);
}

// the statistics pass carrying the forward of the epigenomic MLP stack as its first `nr` workgroups (rider.h): one wave of each
// runs 16 rows of the stack, the pass's own workgroups follow
template <int MT, int CH>
__global__ __launch_bounds__(256 * CH, 3) void first_stats_rider_kernel(const FirstArgs a, const MlpArgs<__bf16> fa, const MmFwdLayout fl, const int nr) {
  if ((int)blockIdx.x < nr) {
    extern __shared__ __attribute__((aligned(16))) char rider_arena[];
    if (threadIdx.x < 64) mlp_fwd_mfma_body(fa, fl, (int)blockIdx.x, rider_arena);
    return;
  }
  first_body<MT, CH, F_STATS>(a, (int)blockIdx.x - nr);
}

#ifdef EMB_CONV_PROF
extern "C" int emb_debug_first_prof(unsigned long long* out, int select) {
  int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_first_prof), sizeof(unsigned long long) * 64);
  unsigned long long z[64] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_first_prof), z, sizeof(z));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_first_sel), &select, sizeof(int));
  return rc;
}
#endif

// ------------------------------------------------------------------------------------------------- host side
struct FirstGeom {
  int SB, slot, tiles_m, tpb, nblk, xrows;
};

static bool first_geom(int B, int L, int cin_pad, int Cout, int k, FirstGeom* gm) {
  const int pad = (k - 1) / 2, KK = k * 8, Lp = (L - 10) / 2 + 1;
  if (cin_pad != 8 || (k & 1) == 0 || KK >= 128 || L < 10 || L > kFBT || Lp < 1) return false;
  if (Cout != 16 && Cout != 32 && Cout != 64) return false;
  gm->SB = kFBT / L;
  gm->slot = L + 2 * pad;
  gm->xrows = gm->SB * gm->slot + kXExtra;
  if (gm->xrows > 256 * kFXV) return false;   // (512-thread variants stage more per slot; 256 is the common bound)
  if ((long)gm->SB * ((L - 10) / 2 + 1) * Cout / 8 > 1024) return false;   // pooled-gradient items per thread (NIT)
  gm->tiles_m = cdiv(B, gm->SB);
  const int target = Cout == 64 ? 256 : 512;                            // 8 waves per CU: one 8-wave or two 4-wave workgroups
  gm->tpb = cdiv(gm->tiles_m, target) < 1 ? 1 : cdiv(gm->tiles_m, target);
  gm->nblk = cdiv(gm->tiles_m, gm->tpb);
  return true;
}

static size_t first_lds(int mode, const FirstGeom& gm, int Lp, int C) {
  const int BN = C, ZP = BN + 8, DPP = BN + 8, AMP = BN + 8;
  size_t bytes = (((size_t)gm.xrows * 8 + 7) & ~(size_t)7) * 2;
  if (mode == F_APPLY || mode == F_BWGRAD || mode == F_BACC) bytes += (size_t)kFBT * ZP * 2;
  if (mode == F_BACC) bytes += (size_t)kFBT * (BN + 4) * 4;   // fp32 scatter tile
  if (mode == F_BSUMS) bytes += (size_t)kFBT * (BN + 4) * 4;
  if (mode >= F_BSUMS) bytes += (size_t)gm.SB * Lp * DPP * 2 + (((size_t)gm.SB * Lp * AMP + 15) & ~(size_t)15);
  bytes += (size_t)4 * 2 * BN * sizeof(float);
  return (bytes + 15) & ~(size_t)15;
}

int conv_first_supported(int dtype, int B, int L, int cin_pad, int Cout, int k) {
  FirstGeom gm;
  if (dtype != EMB_BF16 || !first_geom(B, L, cin_pad, Cout, k, &gm)) return 0;
  const int Lp = (L - 10) / 2 + 1;
  return first_lds(F_BWGRAD, gm, Lp, Cout) <= 150 * 1024 ? 1 : 0;
}

int conv_first_blocks(int B, int L, int cin_pad, int Cout, int k) {
  FirstGeom gm;
  return first_geom(B, L, cin_pad, Cout, k, &gm) ? gm.nblk : 0;
}

template <int MODE> static int first_launch(FirstArgs& a, const FirstGeom& gm, hipStream_t s) {
  const size_t lds = first_lds(MODE, gm, a.Lp, a.C);
  if (MODE == F_STATS) {   // a parked MLP forward of this stream rides along (rider.h)
    Rider r;
    if (rider_take(s, RIDER_MLP_FWD, &r)) {
      const size_t lds2 = lds > r.lds ? lds : r.lds;
      auto go2 = [&](auto kern, int threads) {
        static bool attr = false;
        if (!attr) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
          attr = true;
        }
        kern<<<gm.nblk + r.nwg, threads, lds2, s>>>(a, r.fa, r.fl, r.nwg);
      };
      switch (a.C) {
        case 16: go2(&first_stats_rider_kernel<1, 1>, 256); break;
        case 32: go2(&first_stats_rider_kernel<2, 1>, 256); break;
        default: go2(&first_stats_rider_kernel<2, 2>, 512); break;
      }
      EMB_CHECK_LAUNCH();
      return EMB_OK;
    }
  }
  auto go = [&](auto kern, int threads) {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      attr = true;
    }
    kern<<<gm.nblk, threads, lds, s>>>(a);
  };
  switch (a.C) {
    case 16: go(&first_kernel<1, 1, MODE>, 256); break;
    case 32: go(&first_kernel<2, 1, MODE>, 256); break;
    default: go(&first_kernel<2, 2, MODE>, 512); break;
  }
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

static void first_fill(FirstArgs& a, const FirstGeom& gm, int B, int L, int Cout, int k) {
  a.B = B; a.L = L; a.Lp = (L - 10) / 2 + 1; a.KK = k * 8; a.C = Cout; a.pad = (k - 1) / 2;
  a.SB = gm.SB; a.slot = gm.slot; a.tiles_m = gm.tiles_m; a.tpb = gm.tpb;
}

int conv_first_stats(const void* x, int x_codes, void* nlc_out, const void* w, const void* bias, void* partial, int* rows, float* gram_part,
                     int B, int L, int Cout, int k, hipStream_t s) {
  FirstGeom gm;
  if (!first_geom(B, L, 8, Cout, k, &gm)) return 1;
  FirstArgs a{};
  first_fill(a, gm, B, L, Cout, k);
  a.x = x; a.x_codes = x_codes; a.nlc_out = (__bf16*)nlc_out; a.w = (const __bf16*)w; a.bias = (const float*)bias; a.partial = (float*)partial;
  a.gram_part = gram_part;
  a.gram_edge = gram_part ? reinterpret_cast<__bf16*>(gram_part + (size_t)gm.nblk * kGramPart) : nullptr;
  *rows = gm.nblk;
  return first_launch<F_STATS>(a, gm, s);
}

int conv_first_apply(const void* x, int x_codes, const void* w, const void* bias, const void* stats, const BnFinFwd* fin, void* out,
                     uint8_t* argmax, int out_ncl, const float* gram_part, float* gram_tot,
                     float drop_p, uint64_t seed, uint64_t step_val, const uint64_t* step_dev, int64_t row0, int layer_id, int B, int L,
                     int Cout, int k, hipStream_t s) {
  FirstGeom gm;
  if (!first_geom(B, L, 8, Cout, k, &gm)) return 1;
  FirstArgs a{};
  first_fill(a, gm, B, L, Cout, k);
  a.x = x; a.x_codes = x_codes; a.w = (const __bf16*)w; a.bias = (const float*)bias; a.stats = (const float*)stats;
  a.out = (__bf16*)out; a.argmax = argmax; a.ncl = out_ncl; a.drop_p = drop_p; a.seed = seed; a.step_val = step_val;
  a.step_dev = step_dev; a.grow0 = row0; a.layer_id = layer_id;
  if (fin != nullptr) a.fin_f = *fin;
  a.gram_part = const_cast<float*>(gram_part); a.gram_tot = gram_tot; a.gram_rows = gm.nblk;
  a.gram_edge = gram_part ? reinterpret_cast<__bf16*>(const_cast<float*>(gram_part) + (size_t)gm.nblk * kGramPart) : nullptr;
  constexpr bool park = true;
  if (gram_tot != nullptr && gram_part != nullptr && park) {   // the totals jobs leave this launch's prologue (first_fin.h)
    GramJobsArgs j{};
    j.edge = a.gram_edge; j.part = gram_part; j.tot = gram_tot; j.B = B; j.L = L; j.rows = gm.nblk; j.parts = Cout == 64 ? 2 : 1;
    gram_jobs_park(j, s);
    a.gram_tot = nullptr;
  }
  return first_launch<F_APPLY>(a, gm, s);
}

int conv_first_bwd_sums(const void* dout, int dout_ncl, const uint8_t* argmax, const void* x, int x_codes, const void* w,
                        const void* bias,
                        const void* stats, float keep_scale, void* bpart, int* rows, int B, int L, int Cout, int k, hipStream_t s) {
  FirstGeom gm;
  if (!first_geom(B, L, 8, Cout, k, &gm)) return 1;
  FirstArgs a{};
  first_fill(a, gm, B, L, Cout, k);
  a.x = x; a.x_codes = x_codes; a.w = (const __bf16*)w; a.bias = (const float*)bias; a.stats = (const float*)stats;
  a.dout = (const __bf16*)dout; a.ncl = dout_ncl; a.argmax = const_cast<uint8_t*>(argmax); a.keep_scale = keep_scale;
  a.partial = (float*)bpart;
  *rows = gm.nblk;
  return first_launch<F_BSUMS>(a, gm, s);
}

int conv_first_bwd_wgrad(const void* dout, int dout_ncl, const uint8_t* argmax, const void* x, int x_codes, const void* w,
                         const void* bias,
                         const void* stats, const void* coef, const BnFinBwd* fin, float keep_scale, int training, void* slab, int* slices,
                         int B, int L, int Cout, int k, hipStream_t s) {
  FirstGeom gm;
  if (!first_geom(B, L, 8, Cout, k, &gm)) return 1;
  FirstArgs a{};
  first_fill(a, gm, B, L, Cout, k);
  a.x = x; a.x_codes = x_codes; a.w = (const __bf16*)w; a.bias = (const float*)bias; a.stats = (const float*)stats;
  a.dout = (const __bf16*)dout; a.ncl = dout_ncl; a.argmax = const_cast<uint8_t*>(argmax); a.keep_scale = keep_scale;
  a.coef = (const float*)coef; a.training = training; a.slab = (float*)slab;
  if (fin != nullptr) a.fin_b = *fin;
  *slices = gm.nblk;
  return first_launch<F_BWGRAD>(a, gm, s);
}

// recompute-free backward (first_gram.h): one pass A = g^T xview into per-workgroup slabs ...
int conv_first_bwd_acc(const void* dout, int dout_ncl, const uint8_t* argmax, const void* x, int x_codes, float keep_scale, void* slab, int* slices, int B, int L,
                       int Cout, int k, hipStream_t s) {
  FirstGeom gm;
  if (!first_geom(B, L, 8, Cout, k, &gm)) return 1;
  FirstArgs a{};
  first_fill(a, gm, B, L, Cout, k);
  a.x = x; a.x_codes = x_codes; a.dout = (const __bf16*)dout; a.ncl = dout_ncl; a.argmax = const_cast<uint8_t*>(argmax); a.keep_scale = keep_scale;
  a.training = 1; a.slab = (float*)slab;
  *slices = gm.nblk;
  return first_launch<F_BACC>(a, gm, s);
}

// ... and the per-channel finish: slab sums, lag statistics, weights -> dW (torch layout), dbias, dgamma, dbeta.  In a deferring
// training step (emb_reduce_defer) the job is parked: the optimizer launch runs it as its first workgroups (first_fin.h)
int conv_first_bwd_finish(const void* slab, int slices, const float* gram_tot, const void* w, const void* bias, const void* stats, void* dW,
                          void* dbias, void* dgamma, void* dbeta, int training, int B, int L, int Cin, int Cout, int k, hipStream_t s) {
  FirstFinArgs f{};
  f.slab = (const float*)slab; f.gram = gram_tot; f.w = (const __bf16*)w; f.bias = (const float*)bias; f.stats = (const float*)stats;
  f.dW = (float*)dW; f.dbias = (float*)dbias; f.dgamma = (float*)dgamma; f.dbeta = (float*)dbeta;
  f.S = slices; f.C = Cout; f.k = k; f.Cin = Cin; f.pad = (k - 1) / 2; f.training = training; f.count = (double)B * L;
  return first_fin_submit(f, s);
}

struct FirstFinStore {
  const FirstFinArgs& a;
  int c;
  __device__ __forceinline__ void scalars(float dgamma, float dbeta, float dbias) const {
    a.dgamma[c] = dgamma; a.dbeta[c] = dbeta; a.dbias[c] = dbias;
  }
  __device__ __forceinline__ void dw(long idx, float v) const { a.dW[idx] = v; }
};

// one workgroup per output channel
__global__ __launch_bounds__(1024) void first_bwd_finish_kernel(const FirstFinArgs a) {
  __shared__ float lds[first_finish_lds_floats<1024>()];
  first_finish_body<1024>(a, (int)blockIdx.x, lds, FirstFinStore{a, (int)blockIdx.x});
}

int first_fin_launch(const FirstFinArgs& f, hipStream_t s) {
  first_bwd_finish_kernel<<<f.C, 1024, 0, s>>>(f);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

// the totals jobs as a launch of their own (a parked set nobody carried) ...
__global__ __launch_bounds__(256) void gram_jobs_kernel(const GramJobsArgs a) {
  __shared__ float scratch[16 * 4];
  GramPre pre;
  pre.have = false;
  gram_job<256>((int)blockIdx.x, pre, a.edge, a.B, a.L, a.part, a.rows, a.parts, a.tot, scratch);
}
int gram_jobs_launch(const GramJobsArgs& a, hipStream_t s) {
  gram_jobs_kernel<<<kGramJobs, 256, 0, s>>>(a);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}
int gram_jobs_count() { return kGramJobs; }

int conv_first_gram_floats() { return kGramRow; }
size_t conv_first_gram_part_bytes(int B, int L, int cin_pad, int Cout, int k) {   // partial G0 rows, then the edge image
  FirstGeom gm;
  if (!first_geom(B, L, cin_pad, Cout, k, &gm)) return 0;
  return (size_t)gm.nblk * kGramPart * sizeof(float) + (size_t)kGramEdgeRows * B * sizeof(__bf16);
}

}  // namespace emb
