// Probe: lane/byte permutation performed by ds_read_b64_tr_b16 and ds_read_b64_tr_b8 (gfx950).
// Each lane l supplies the LDS address 8*l; LDS byte at offset o holds (o & 255) in pass 0 and (o >> 8) in pass 1, so the
// printed table says, for every destination (lane, byte), which source (lane, byte) it received.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) i32x2 lds_i32x2;

__global__ void probe(unsigned long long* out16, unsigned long long* out8, int pass) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[512];
  for (int o = threadIdx.x; o < 512; o += 64) sm[o] = pass == 0 ? (o & 255) : (o >> 8);
  __syncthreads();
  const unsigned a = (unsigned)(size_t)sm + 8u * threadIdx.x;
  s16x4 v16 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(size_t)a);
  i32x2 v8 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)(size_t)a);
  union { s16x4 v; unsigned long long u; } c16; c16.v = v16;
  union { i32x2 v; unsigned long long u; } c8; c8.v = v8;
  out16[threadIdx.x] = c16.u;
  out8[threadIdx.x] = c8.u;
}

int main() {
  unsigned long long *d16, *d8, h16[2][64], h8[2][64];
  hipMalloc(&d16, 512); hipMalloc(&d8, 512);
  for (int pass = 0; pass < 2; ++pass) {
    probe<<<1, 64>>>(d16, d8, pass);
    hipMemcpy(h16[pass], d16, 512, hipMemcpyDeviceToHost);
    hipMemcpy(h8[pass], d8, 512, hipMemcpyDeviceToHost);
  }
  for (int which = 0; which < 2; ++which) {
    printf("%s: dst lane: [dst byte -> src lane.byte]\n", which == 0 ? "tr16" : "tr8");
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d:", l);
      for (int b = 0; b < 8; ++b) {
        unsigned long long lo = which == 0 ? h16[0][l] : h8[0][l], hi = which == 0 ? h16[1][l] : h8[1][l];
        int off = (int)((lo >> (8 * b)) & 255) | (int)(((hi >> (8 * b)) & 255) << 8);
        printf(" %2d.%d", off / 8, off % 8);
      }
      printf("\n");
    }
  }
  return 0;
}
