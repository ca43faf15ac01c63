// BatchNorm statistics finalised by the CONSUMER of the per-workgroup partial sums, in its prologue: every workgroup of the
// consuming launch sums the (few hundred) partial rows itself -- same loads, same order, so all workgroups hold identical
// values -- and workgroup 0 also writes the finalised vectors / running statistics to global memory for later launches.
// This removes the one-workgroup-per-channel finalize launches (≈5 us each, nothing but dispatch + two memory round trips)
// from the critical path where the partial-row count is small (<= 512 rows: 128 KiB per consuming workgroup, L2 resident).
// The global-batch statistics path (bn_phase 1 / 2: the sums travel through an all-reduce) keeps the separate kernels.
#pragma once
#include "common.h"

namespace emb {

constexpr int kBnInlineMaxC = 128;   // channels of the kernels that keep the finalisation scratch in static LDS

struct BnFinFwd {                // forward: sum z, sum z^2 -> mean, invstd, scale, shift (+ running statistics)
  const float* partial;          // [rows][2][C]; nullptr: the statistics are already final (read `stats`)
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  float* stats;                  // [4][C], written by workgroup 0
  long long* num_batches_tracked;
  double momentum, eps, count;
  int rows;
};
struct BnFinBwd {                // backward: sum dz, sum dz*xhat -> dbeta, dgamma, coef = sums / count
  const float* partial;
  float* dgamma;
  float* dbeta;
  float* coef;                   // [2][C], written by workgroup 0
  double count;
  int rows;
};

// column sums of partial[rows][cols] into tot[cols] (LDS, double); sm = NT * 4 doubles of LDS scratch.  cols % 4 == 0.
// Thread (group, 16-byte column) sums rows group, group + G, ... in row order; the groups meet in group order.
template <int NT>
__device__ __forceinline__ void bn_colsums_f32(const float* __restrict__ partial, int rows, int cols, double* sm, double* tot) {
  const int ncv = cols >> 2, G = NT / ncv;
  const int cv = threadIdx.x % ncv, grp = threadIdx.x / ncv;
  if (grp < G) {
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    const float* src = partial + cv * 4;
    int r = grp;
    for (; r + 15 * G < rows; r += 16 * G) {           // sixteen rows in flight
      float4 v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<const float4*>(src + (long)(r + j * G) * cols);
      // the sixteen rows meet in f32 (a fixed binary tree: four additions deep), the sixteen-row sums in f64: 1/16 of the f64
      // conversions and additions, which sit in the prologue of EVERY workgroup of the consuming launch (the partial rows are f32
      // tile sums already; the tree loses ~2e-7 of a sixteen-row sum, the f64 part keeps E[x^2] - mean^2 benign; bf16 compute only)
#pragma unroll
      for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
        for (int j = 0; j < w; ++j) { v[j].x += v[j + w].x; v[j].y += v[j + w].y; v[j].z += v[j + w].z; v[j].w += v[j + w].w; }
      a0 += (double)v[0].x; a1 += (double)v[0].y; a2 += (double)v[0].z; a3 += (double)v[0].w;
    }
    for (; r < rows; r += G) {
      const float4 v = *reinterpret_cast<const float4*>(src + (long)r * cols);
      a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
    }
    double* d = sm + grp * cols + cv * 4;
    d[0] = a0; d[1] = a1; d[2] = a2; d[3] = a3;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < cols; i += NT) {
    double t = 0.0;
    for (int g2 = 0; g2 < G; ++g2) t += sm[g2 * cols + i];
    tot[i] = t;
  }
  __syncthreads();
}

// forward finalisation into LDS fs[4][C] (all workgroups) and to global memory (workgroup `first` only).  scratch: NT*4 + 2C doubles.
template <int NT>
__device__ __forceinline__ void bn_fin_fwd(const BnFinFwd& f, int C, double* scratch, float* fs, bool first) {
  double* tot = scratch + NT * 4;
  bn_colsums_f32<NT>(f.partial, f.rows, 2 * C, scratch, tot);
  for (int c = threadIdx.x; c < C; c += NT) {
    const double count = f.count;
    const double mean = tot[c] / count;
    double var = tot[C + c] / count - mean * mean;       // biased variance (normalisation); double keeps the cancellation benign
    if (var < 0) var = 0;
    const double invstd = 1.0 / sqrt(var + f.eps);
    const double scale = (double)f.gamma[c] * invstd;
    const float m = (float)mean, is = (float)invstd, sc = (float)scale, sh = (float)((double)f.beta[c] - mean * scale);
    fs[c] = m; fs[C + c] = is; fs[2 * C + c] = sc; fs[3 * C + c] = sh;
    if (first) {
      f.stats[c] = m; f.stats[C + c] = is; f.stats[2 * C + c] = sc; f.stats[3 * C + c] = sh;
      const double unbiased = count > 1 ? var * count / (count - 1.0) : var;   // nn.BatchNorm1d: running_var is unbiased
      f.running_mean[c] = (float)((1.0 - f.momentum) * (double)f.running_mean[c] + f.momentum * mean);
      f.running_var[c] = (float)((1.0 - f.momentum) * (double)f.running_var[c] + f.momentum * unbiased);
      if (c == 0 && f.num_batches_tracked != nullptr) *f.num_batches_tracked += 1;
    }
  }
  __syncthreads();
}

// backward finalisation into LDS fc[2][C] = mean(dz), mean(dz*xhat) and (workgroup `first`) dbeta / dgamma / coef in global memory
template <int NT>
__device__ __forceinline__ void bn_fin_bwd(const BnFinBwd& f, int C, double* scratch, float* fc, bool first) {
  double* tot = scratch + NT * 4;
  bn_colsums_f32<NT>(f.partial, f.rows, 2 * C, scratch, tot);
  for (int c = threadIdx.x; c < C; c += NT) {
    const float c0 = (float)(tot[c] / f.count), c1 = (float)(tot[C + c] / f.count);
    fc[c] = c0; fc[C + c] = c1;
    if (first) {
      f.dbeta[c] = (float)tot[c];
      f.dgamma[c] = (float)tot[C + c];
      f.coef[c] = c0; f.coef[C + c] = c1;
    }
  }
  __syncthreads();
}

}  // namespace emb
