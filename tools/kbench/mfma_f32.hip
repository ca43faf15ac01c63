// Developer micro-benchmark: v_mfma_f32_16x16x4_f32 cadence with one and with two waves per SIMD (is the matrix pipe shared without loss?)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC> __global__ void k(float* out, unsigned long long* cyc, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 4096 * 8);
  const int iters = 2000;
  for (int threads : {64, 256, 512, 1024}) {
    for (int rep = 0; rep < 2; ++rep) {
      k<8><<<256, threads>>>(out, cyc, iters);
      hipDeviceSynchronize();
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<8><<<2048, threads>>>(out, cyc, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("threads %4d: wall %.3f ms for 2048 workgroups -> %.1f TFLOP/s\n", threads, ms, 2048.0 * (threads / 64) * iters * 8.0 * 2048.0 / ms * 1e-9);
    k<8><<<256, threads>>>(out, cyc, iters); hipDeviceSynchronize();
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
    const double waves_per_simd = threads / 256.0 < 1 ? 1 : threads / 256.0;
    printf("threads %4d: %.1f cycles per MFMA per wave, %.1f per MFMA per SIMD\n", threads, avg / (iters * 8.0), avg / (iters * 8.0 * waves_per_simd));
  }
  return 0;
}
