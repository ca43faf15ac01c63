// Philox4x32-10 counter RNG (Salmon et al., SC'11) -- the library's perf-mode RNG contract
// (include/embrace_hip.h, "RNG contract").  Restated for the oracle in oracle/embrace_oracle.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace emb {

struct Philox4 {
  uint32_t x, y, z, w;
};

__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint64_t seed, uint64_t stream, uint64_t index) {
  uint32_t c0 = (uint32_t)index, c1 = (uint32_t)(index >> 32);
  uint32_t c2 = (uint32_t)stream, c3 = (uint32_t)(stream >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}

// same construction as at::uniform_real_distribution<double>: (x & (2^53-1)) * 2^-53
__host__ __device__ __forceinline__ double uniform53(uint32_t lo, uint32_t hi) {
  const uint64_t raw = ((uint64_t)hi << 32) | lo;
  return (double)(raw & ((1ull << 53) - 1)) * (1.0 / 9007199254740992.0);
}
// same construction as at::uniform_real_distribution<float>: (x & (2^24-1)) * 2^-24
__host__ __device__ __forceinline__ float uniform24(uint32_t x) {
  return (float)(x & ((1u << 24) - 1)) * (1.0f / 16777216.0f);
}

__host__ __device__ __forceinline__ uint64_t rng_stream(uint64_t step, uint32_t kind) {
  return (step << 8) | (uint64_t)kind;
}

}  // namespace emb
