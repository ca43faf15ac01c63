// Lag statistics of the first conv block's INPUT and the recompute-free backward built on them.
//
// The first block is Conv1d(4 -> C, k) -> BatchNorm1d -> ReLU -> MaxPool (CNN_pre.py:37-44) on a few-channel input x.  Its
// convolution z[r][c] = sum_m W[c][m] xv[r][m] + b[c] (xv = the zero-padded k*cin view of x) is LINEAR in W, so every sum the
// BatchNorm backward needs can be written with three small objects instead of re-running the convolution:
//   A[c][m]  = sum_r g[r][c] xv[r][m]        (g = dL/dy after pool / ReLU backward; one pass over the pooled gradient: F_BACC)
//   M[m'][m] = sum_r xv[r][m'] xv[r][m]      (depends on the input only)
//   sx[m]    = sum_r xv[r][m]
// with  sum g z = sum_m W A + b sum g,  sum z xv = W M + b sx,  xhat = (z - mean) inv:
//   dgamma = inv (sum g z - mean sum g),  dbeta = sum g,  m1 = dbeta / R,  m2 = dgamma / R,
//   dW[c][m] = scale_c (A - m1 sx - m2 inv (W M + (b - mean) sx))[c][m],   db = 0.
// M is block-Toeplitz up to the zero padding at the sequence ends: with P(u, d) = sum_b x_b[u] (x) x_b[u + d] (4 x 4, zero
// past the sequence),  M(t1, t2) = Tot[d] - Head[max(0, t1 - pad)][d] - Tail[max(0, pad - t1)][d]  for d = t2 - t1 >= 0
// (transposed for d < 0), Tot[d] = sum_u P(u, d), Head[j] = sum_{u < j} P(u, .), Tail[j] = sum_{v < j} P(L - 1 - v, .).
// The statistics pass (F_STATS) already has every input tile in LDS: it adds
//   G0[ci][(tap, c2)] = sum_r x~[r - pad][ci] x~[r - pad + tap][c2] = Tot[tap] - Tail[pad][tap]      one MFMA per wave and k-step
// and writes it as one partial row per workgroup.  The apply pass's prologue (gram_job) sums those rows column-wise and computes
// the edge products P(u, d), P(L - 1 - u, d) for u < 7 straight from x (they touch 21 positions at either end of a sequence),
// one job per workgroup, so the totals are in global memory when the backward needs them.
// Numerically this is the same computation in a different association (verified against autograd in fp64 to 1e-15 in a numpy
// prototype); on the GPU both paths round the dense gradient tile to bf16 once and agree to 2e-3 .. 2e-2 of the largest gradient
// (tests/test_gpu_convblock.py::test_first_block_linear_backward_equals_the_recomputing_backward).
#pragma once
#include "gemm_core.h"
#include "conv_tiles.h"
#include "first_fin.h"

namespace emb {

// layout of the TOTALS row (floats)
constexpr int kGramG0 = 0;                       // [4][128]        G0[ci][tap * 8 + c2]
constexpr int kGramP = 512;                      // [2][7][15][16]  P(u, d) at the head (which = 0: position u) / tail (position L - 1 - u)
constexpr int kGramColS = kGramP + 2 * 7 * 15 * 16;   // (64 floats not in use)
constexpr int kGramEdgeS = kGramColS + 64;       // [2][7][4]       x at the head / tail positions, summed over the batch
constexpr int kGramRow = 4096;                   // floats per totals row (3992 used)
constexpr int kGramMaxK = 15, kGramEdge = 7;     // taps, edge positions (k <= 15)
constexpr int kGramOnesCol = 120;                // column of G0 that holds sum_r x~[r - pad][ci] = colTot - TailS[pad] (k * 8 <= 120)

constexpr int kGramEdgeHead = 21, kGramEdgeTail = 7, kGramEdgeRows = (kGramEdgeHead + kGramEdgeTail) * 4;   // edge image rows (position, channel)
constexpr int kGramPart = 512;                    // floats per PARTIAL row (one per statistics workgroup): the G0 block only
constexpr int kGramJobs = 2 * kGramEdge * kGramMaxK + 2 * kGramEdge + kGramPart / 16;   // 210 edge products, 14 edge sums, 32 G0 column blocks

// one tile of the statistics pass.  xs: the staged tile [rows][8] bf16 (zero halos, channels 4..7 zero), row_off(row): LDS element offset of tile row `row`.
// G0 on the matrix cores over the REAL channels: A = x~[r - pad] (the tap-0 block of the view), B = compact 16-column block nb of
// the view (column n' = tap * 4 + c2: a lane's transposing read takes the four real channels of tap nb * 4 + p4), 64 columns =
// four blocks.  Wave = (block nb, part kh of the k-steps): NW / 4 partial accumulators per block, stored side by side.
constexpr int kGramOnesC = 60;                    // compact column forced to ones (4 k <= 60): G0[ci][60] = sum_r x~[r - pad][ci]
template <int NW, int UNR, typename RowOff>
__device__ __forceinline__ void gram_tile(const __bf16* xs, RowOff row_off, f32x4& accg, int L, int SB, int slot, int lane, int wave) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  constexpr int KSN = 8 / (NW / 4);                          // k-steps per wave
  const int g = lane >> 4, r16 = lane & 15, q4 = r16 >> 2, p4 = r16 & 3;
  const int nb = wave & 3, ks0 = (wave >> 2) * KSN, xoff = (nb * 4 + p4) * 8;
#pragma unroll UNR
  for (int ks = ks0; ks < ks0 + KSN; ++ks) {
    const int ra = ks * 32 + 8 * g + q4;
    const int x0 = row_off(ra), x1 = row_off(ra + 4);
    const int zrow = SB * slot * 8;                          // A operand: rows past the tile's sequences read a zero row
    union { struct { s16x4 lo, hi; } s; bf16x8 v; } ua, ub;
    ua.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xs + (ra < SB * L ? x0 : zrow) + 4 * p4));
    ua.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xs + (ra + 4 < SB * L ? x1 : zrow) + 4 * p4));
    ub.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xs + x0 + xoff));
    ub.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xs + x1 + xoff));
    if (nb == kGramOnesC / 16 && r16 == kGramOnesC % 16) {
#pragma unroll
      for (int e = 0; e < 8; ++e) ub.v[e] = (__bf16)1.0f;
    }
    accg = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, accg, 0, 0, 0);
  }
}

// partial row: [NW / 4 parts][4 ci][64 n']
__device__ __forceinline__ void gram_store(float* row, const f32x4& accg, int lane, int wave) {
  const int g = lane >> 4, r16 = lane & 15;
  if (g == 0) {                                             // accumulator rows 0..3 = input channels 0..3
#pragma unroll
    for (int r = 0; r < 4; ++r) row[(wave >> 2) * 256 + r * 64 + (wave & 3) * 16 + r16] = accg[r];
  }
}

// The edge image E[row][b] (bf16, b contiguous): row = j * 4 + ch for the head positions j = 0..20 and (21 + j) * 4 + ch for the tail
// positions L - 7 + j, j = 0..6; zero where the position is outside the sequence.  The statistics pass writes it from its staged
// tiles (112 two-byte stores per sequence); the edge jobs below then read whole lines of it instead of one position per line of x.
__device__ __forceinline__ void gram_edge_store(const __bf16* xs, __bf16* edge, int B, int L, int slot, int pad, int b0, int nseq) {
  const int t = threadIdx.x;
  if (t >= kGramEdgeRows) return;
  const int j = t >> 2, ch = t & 3, pos = j < kGramEdgeHead ? j : L - kGramEdgeTail + (j - kGramEdgeHead);
  for (int sq = 0; sq < nseq; ++sq)
    edge[(long)t * B + b0 + sq] = (pos >= 0 && pos < L) ? xs[(sq * slot + pos + pad) * 8 + ch] : (__bf16)0.0f;
}

// Totals of the lag statistics, written by the apply pass's prologue: job j of kGramJobs (workgroups take jobs j = wg, wg + nwg, ..)
//   j < 210        P(which, u, d) = sum_b x_b[pos] (x) x_b[pos + d], pos = u (head) or L - 1 - u (tail): eight rows of the edge image,
//                  the threads walk the batch, sixteen sums meet wave-wise and then across the waves in wave order
//   210 <= j < 224 x at the edge positions, summed over the batch
//   224 <= j       eight entries of G0: column sums of the statistics pass's partial rows (and of their `parts` k-step parts)
struct GramEdgeJob {          // decoded edge job: image rows of x[pos] and x[pos + d]
  int ja, jb;                 // jb < 0: no partner (edge sums), or the partner lies past the sequence (the products are zero)
  bool prod, live;
};
__device__ __forceinline__ GramEdgeJob gram_edge_job(int job, int L) {
  constexpr int NP = 2 * kGramEdge * kGramMaxK;
  GramEdgeJob j;
  j.prod = job < NP;
  int which, u, d = 0;
  if (j.prod) {
    which = job / (kGramEdge * kGramMaxK);
    const int rem = job - which * kGramEdge * kGramMaxK;
    u = rem / kGramMaxK; d = rem - u * kGramMaxK;
  } else {
    which = (job - NP) / kGramEdge; u = (job - NP) - which * kGramEdge;
  }
  // head: u + d <= 20 always inside the image; tail: pos + d <= L - 1 iff d <= u
  j.ja = which ? kGramEdgeHead + (kGramEdgeTail - 1 - u) : u;
  j.jb = !j.prod ? -1 : which ? (d <= u ? j.ja + d : -1) : u + d;
  j.live = u < L && (!j.prod || j.jb >= 0);
  return j;
}
// two sequences per thread and load (B even): the eight loads of a job's first chunk are issued early (before the BatchNorm
// finalisation of the same prologue) and consumed by gram_job
struct GramPre { uint32_t va[4], vb[4]; bool have; };
template <int NTHR>
__device__ __forceinline__ GramPre gram_job_preload(int job, const __bf16* __restrict__ edge, int B, int L) {
  GramPre r;
  r.have = false;
#pragma unroll
  for (int c = 0; c < 4; ++c) { r.va[c] = 0u; r.vb[c] = 0u; }
  if (job >= 2 * kGramEdge * kGramMaxK + 2 * kGramEdge || (B & 1)) return r;
  const GramEdgeJob j = gram_edge_job(job, L);
  r.have = true;
  const int b2 = threadIdx.x;
  if (!j.live || 2 * b2 >= B) return r;
  const uint32_t* ea = reinterpret_cast<const uint32_t*>(edge + (long)j.ja * 4 * B);
  const uint32_t* eb = reinterpret_cast<const uint32_t*>(edge + (long)(j.jb < 0 ? j.ja : j.jb) * 4 * B);
#pragma unroll
  for (int c = 0; c < 4; ++c) { r.va[c] = ea[(long)c * (B >> 1) + b2]; r.vb[c] = eb[(long)c * (B >> 1) + b2]; }
  return r;
}

template <int NTHR>
__device__ __forceinline__ void gram_job(int job, const GramPre& pre, const __bf16* __restrict__ edge, int B, int L, const float* __restrict__ part,
                                         int rows, int parts, float* __restrict__ tot, float* scratch) {
  constexpr int NP = 2 * kGramEdge * kGramMaxK, NE = 2 * kGramEdge, NWV = NTHR / 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (job < NP + NE) {
    const GramEdgeJob j = gram_edge_job(job, L);
    const bool prod = j.prod;
    float p[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) p[e] = 0.0f;
    auto fma16 = [&](const float (&xa)[4], const float (&xb)[4]) {
#pragma unroll
      for (int c1 = 0; c1 < 4; ++c1)
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) p[c1 * 4 + c2] += prod ? xa[c1] * xb[c2] : (c2 == 0 ? xa[c1] : 0.0f);
    };
    auto pair16 = [&](const uint32_t (&va)[4], const uint32_t (&vb)[4]) {   // sequences 2 * b2 (low halves) and 2 * b2 + 1, in that order
      float xa[4], xb[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) { xa[c] = __uint_as_float(va[c] << 16); xb[c] = __uint_as_float(vb[c] << 16); }
      fma16(xa, xb);
#pragma unroll
      for (int c = 0; c < 4; ++c) { xa[c] = __uint_as_float(va[c] & 0xFFFF0000u); xb[c] = __uint_as_float(vb[c] & 0xFFFF0000u); }
      fma16(xa, xb);
    };
    if (j.live) {
      const __bf16* ea = edge + (long)j.ja * 4 * B;
      const __bf16* eb = edge + (long)(j.jb < 0 ? j.ja : j.jb) * 4 * B;
      if (!(B & 1)) {
        int b2 = t;
        if (pre.have) { pair16(pre.va, pre.vb); b2 += NTHR; }   // (lanes past the batch preloaded zeros)
        for (; 2 * b2 < B; b2 += NTHR) {
          uint32_t va[4], vb[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            va[c] = reinterpret_cast<const uint32_t*>(ea)[(long)c * (B >> 1) + b2];
            vb[c] = reinterpret_cast<const uint32_t*>(eb)[(long)c * (B >> 1) + b2];
          }
          pair16(va, vb);
        }
      } else {
        for (int b = t; b < B; b += NTHR) {
          float xa[4], xb[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) { xa[c] = (float)ea[(long)c * B + b]; xb[c] = (float)eb[(long)c * B + b]; }
          fma16(xa, xb);
        }
      }
    }
    // sixteen sums: within each row of 16 lanes by DPP (no LDS round trips), across the four rows with two shuffles (the sixteen
    // chains are independent), then the NWV wave sums in wave order
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float r = row16_sum<float>(p[e]);
      r += __shfl_xor(r, 16, 64);
      r += __shfl_xor(r, 32, 64);
      if (lane == 0) scratch[e * NWV + wave] = r;
    }
    __syncthreads();
    if (t < 16) {
      float s = 0.0f;
#pragma unroll
      for (int w = 0; w < NWV; ++w) s += scratch[t * NWV + w];
      if (prod) tot[kGramP + job * 16 + t] = s;
      else if ((t & 3) == 0) tot[kGramEdgeS + (job - NP) * 4 + (t >> 2)] = s;   // (sum slot c1 * 4: the edge value of channel c1)
    }
    __syncthreads();
    return;
  }
  // G0: eight of the 4 x 64 compact entries; thread (row group rg, entry): rows rg, rg + G, .. of every part; the eight row groups
  // of a wave meet by DPP / shuffles (lanes 8 apart hold the same entry), the waves in wave order; the totals keep the padded
  // column index the finish uses (tap * 8 + c2; the ones column -> kGramOnesCol)
  constexpr int G = NTHR / 8;
  const int o = (job - NP - NE) * 8 + (t & 7), rg = t >> 3;
  float acc = 0.0f;
  for (int r = rg; r < rows; r += G)
    for (int h = 0; h < parts; ++h) acc += part[(long)r * kGramPart + h * 256 + o];
  acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x128, 0xf, 0xf, true));   // row_ror:8
  acc += __shfl_xor(acc, 16, 64);
  acc += __shfl_xor(acc, 32, 64);
  if (lane < 8) scratch[wave * 8 + lane] = acc;
  __syncthreads();
  if (t < 8) {
    float s = 0.0f;
#pragma unroll
    for (int q = 0; q < NWV; ++q) s += scratch[q * 8 + t];
    const int ci = o >> 6, np = o & 63;
    tot[kGramG0 + ci * 128 + (np == kGramOnesC ? kGramOnesCol : (np >> 2) * 8 + (np & 3))] = s;
  }
  __syncthreads();
}

}  // namespace emb

namespace emb {

constexpr int kGramUsed = kGramEdgeS + 2 * kGramEdge * 4;   // floats of a totals row in use
constexpr int kFinFixed = kGramUsed + 4 + 128 + 128 + 4;   // floats of LDS before As
static_assert(kFinFixed % 4 == 0, "As must start on a 16-byte boundary (vector stores of the slab sum)");
constexpr int kFinCols = 64;                       // floats per slab row of the accumulate pass: 4 k compact columns + sum of g, k <= 15
// slice groups of the finish's slab sum: NTHR / 64 threads per column (1024 threads: 16), or NTHR / 16 per four columns (fewer)
template <int NTHR> constexpr int first_finish_groups() { return NTHR >= 1024 ? NTHR / kFinCols : NTHR / 16; }
template <int NTHR> constexpr int first_finish_lds_floats() { return kFinFixed + first_finish_groups<NTHR>() * kFinCols; }

// The per-channel finish of the recompute-free backward (top of this file): channel c, NTHR threads (a multiple of 128), `lds`
// = first_finish_lds_floats<NTHR>() floats.  Results go to `sink`: sink.scalars(dgamma, dbeta, dbias) once (thread 0) and
// sink.dw(index into the torch-layout weight [C][Cin][k], value) for every real weight of the channel -- the standalone kernel
// stores them, the optimizer launch (loss_optim.hip) updates the parameters with them on the spot.
template <int NTHR, typename Sink>
__device__ __forceinline__ void first_finish_body(const FirstFinArgs& a, const int c, float* lds, Sink&& sink) {
  constexpr int NG = first_finish_groups<NTHR>(), NPd = kGramMaxK * 16;   // slice groups of the slab sum; (d, c1, c2) entries per edge position
  float* G0 = lds + kGramG0;                              // the totals row, verbatim: G0 [4][128],
  float* PP = lds + kGramP;                               // [2][7][15][16] P totals -> inclusive prefix sums over the edge position,
  float* ES = lds + kGramEdgeS;                           // [2][7][4] edge sums -> inclusive prefix sums
  float* colT = lds + kGramUsed;                          // [4]
  float* Ar = colT + 4;                                   // [128]
  float* Wr = Ar + 128;                                   // [128]
  float* red = Wr + 128;                                  // [4]
  float* As = red + 4;                                    // [NG][kFinCols]
  const int KK = a.k * 8, k = a.k, pad = a.pad, tid = threadIdx.x;
  // wide variant (the standalone kernel): the totals travel through registers, in flight during the slab sum; narrow variant
  // (inside the optimizer launch, whose other workgroups want few registers): a plain copy first, eight slab loads in flight
  constexpr bool WIDE = NTHR >= 1024;
  constexpr int NGV = WIDE ? (kGramUsed + NTHR - 1) / NTHR : 1, NIF = WIDE ? 16 : 8;
  float gtot[NGV];
  if (WIDE) {
#pragma unroll
    for (int q = 0; q < NGV; ++q) gtot[q] = tid + q * NTHR < kGramUsed ? a.gram[tid + q * NTHR] : 0.0f;
  } else {
    for (int i = tid; i < kGramUsed; i += NTHR) lds[i] = a.gram[i];
  }
  if constexpr (WIDE) {
      // A[c][n'] = sum over the slices (compact columns n' = tap * 4 + ci, 4 k = sum of g; kFinCols per slab row): NG slice groups
      // x 64 columns, NIF loads in flight, the groups meet in group order
    const int m = tid & (kFinCols - 1), sgp = tid / kFinCols;
    float s = 0.0f;
    if (m <= 4 * k) {
      const float* src = a.slab + (long)c * kFinCols + m;
      const long stride = (long)a.C * kFinCols;
      int i = sgp;
      for (; i + (NIF - 1) * NG < a.S; i += NIF * NG) {
        float v[NIF];
#pragma unroll
        for (int j = 0; j < NIF; ++j) v[j] = src[(long)(i + j * NG) * stride];
#pragma unroll
        for (int j = 0; j < NIF; ++j) s += v[j];
      }
      for (; i < a.S; i += NG) s += src[(long)i * stride];
    }
    As[sgp * kFinCols + m] = s;
  } else {
    // the same sum with 16-byte loads: thread (slice group, four columns), NG = NTHR / 16 groups -- at S = 256 slices and 256
    // threads every thread holds 16 slices, two rounds of eight loads (one column per thread: 64 slices, eight dependent rounds,
    // the critical path of the optimizer launch this finish rides in)
    const int cg = tid & 15, sgp = tid >> 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (4 * cg <= 4 * k) {
      const float* src = a.slab + (long)c * kFinCols + 4 * cg;
      const long stride = (long)a.C * kFinCols;
      int i = sgp;
      for (; i + (NIF - 1) * NG < a.S; i += NIF * NG) {
        f32x4 v[NIF];
#pragma unroll
        for (int j = 0; j < NIF; ++j) v[j] = *reinterpret_cast<const f32x4*>(src + (long)(i + j * NG) * stride);
#pragma unroll
        for (int j = 0; j < NIF; ++j) s += v[j];
      }
      for (; i < a.S; i += NG) s += *reinterpret_cast<const f32x4*>(src + (long)i * stride);
    }
    *reinterpret_cast<f32x4*>(As + sgp * kFinCols + 4 * cg) = s;
  }
  if (WIDE) {
#pragma unroll
    for (int q = 0; q < NGV; ++q)
      if (tid + q * NTHR < kGramUsed) lds[tid + q * NTHR] = gtot[q];
  }
  __syncthreads();
  // prefix sums over the edge positions, in place: PP[which][u] := sum_{u' <= u} P(which, u'), likewise the edge sums
  for (int i = tid; i < 2 * NPd; i += NTHR) {
    const int which = i / NPd, e = i - which * NPd;
    float run = 0.0f;
#pragma unroll
    for (int u = 0; u < kGramEdge; ++u) {
      run += PP[(which * kGramEdge + u) * NPd + e];
      PP[(which * kGramEdge + u) * NPd + e] = run;
    }
  }
  if (tid >= NTHR - 8) {
    const int i = tid - (NTHR - 8), which = i >> 2, ch = i & 3;
    float run = 0.0f;
    for (int u = 0; u < kGramEdge; ++u) {
      run += ES[(which * kGramEdge + u) * 4 + ch];
      ES[(which * kGramEdge + u) * 4 + ch] = run;
    }
  }
  if (tid < 128) {   // Ar: the padded column index the weights use (tap * 8 + ci; KK = sum of g), zero in the pad columns
    const int tap = tid >> 3, ci = tid & 7, src = tid == KK ? 4 * k : (tid < KK && ci < 4 ? tap * 4 + ci : -1);
    float s = 0.0f;
    if (src >= 0) {
#pragma unroll
      for (int q = 0; q < NG; ++q) s += As[q * kFinCols + src];
    }
    Ar[tid] = s;
    Wr[tid] = tid < KK ? (float)a.w[(long)c * KK + tid] : 0.0f;
  }
  __syncthreads();
  // Head[j] = sum_{u < j} P_head(u), Tail[j] = sum_{v < j} P_tail(v): j = 0 is the empty sum
  auto Hd = [&](int j, int e) { return j ? PP[((0 * kGramEdge + j - 1)) * NPd + e] : 0.0f; };
  auto Tl = [&](int j, int e) { return j ? PP[((1 * kGramEdge + j - 1)) * NPd + e] : 0.0f; };
  auto HS = [&](int j, int ch) { return j ? ES[(0 * kGramEdge + j - 1) * 4 + ch] : 0.0f; };
  auto TS = [&](int j, int ch) { return j ? ES[(1 * kGramEdge + j - 1) * 4 + ch] : 0.0f; };
  if (tid >= 128) return;
  const int t = tid;
  // sum_m W[c][m] A[c][m]
  float part = t < KK ? Wr[t] * Ar[t] : 0.0f;
  part = row16_sum<float>(part);
  part += __shfl_xor(part, 16, 64);
  part += __shfl_xor(part, 32, 64);
  if ((t & 63) == 0) red[t >> 6] = part;
  // (read before the barrier: the sink of thread 0 may UPDATE the bias parameter in place)
  const float mean = a.stats[c], inv = a.stats[a.C + c], sc = a.stats[2 * a.C + c], b = a.bias[c];
  __syncthreads();   // (all 128 remaining threads: two whole waves)
  const float dotWA = red[0] + red[1], sg = Ar[KK];
  const float gz = dotWA + b * sg;                 // sum g z
  const float gx = inv * (gz - mean * sg);         // sum g xhat = dgamma
  const float m1 = (float)((double)sg / a.count), m2 = (float)((double)gx / a.count);
  if (t == 0) sink.scalars(gx, sg, a.training ? 0.0f : sc * sg);   // (behind training-mode BatchNorm the bias gradient is exactly zero)
  if (t < KK) {
    const int t2 = t >> 3, c2 = t & 7;
    if (c2 < a.Cin) {
      float dw = sc * Ar[t];
      if (a.training) {
        const int h2 = max(0, t2 - pad), l2 = max(0, pad - t2);
        // the ones column of G0 sums positions 0 .. L - 1 - pad
        const float sx = (G0[c2 * 128 + kGramOnesCol] + TS(pad, c2)) - HS(h2, c2) - TS(l2, c2);
        float wm = 0.0f;                           // sum_{m'} W[c][m'] M[m'][m]
        for (int t1 = 0; t1 < k; ++t1) {
          const bool up = t1 <= t2;                // d >= 0: block (t1, t2) as stored; else its transpose
          const int d = up ? t2 - t1 : t1 - t2, te = up ? t1 : t2;
          const int hj = max(0, te - pad), lj = max(0, pad - te);
#pragma unroll
          for (int c1 = 0; c1 < 4; ++c1) {
            const int ab = up ? c1 * 4 + c2 : c2 * 4 + c1;                 // (c1, c2) of the stored block
            const int ga = up ? c1 : c2, gb = up ? c2 : c1;
            const float tot = G0[ga * 128 + d * 8 + gb] + Tl(pad, d * 16 + ab);
            const float mel = tot - Hd(hj, d * 16 + ab) - Tl(lj, d * 16 + ab);
            wm += Wr[t1 * 8 + c1] * mel;
          }
        }
        dw = sc * (Ar[t] - m1 * sx - m2 * inv * (wm + (b - mean) * sx));
      }
      sink.dw(((long)c * a.Cin + c2) * k + t2, dw);
    }
  }
}

}  // namespace emb
