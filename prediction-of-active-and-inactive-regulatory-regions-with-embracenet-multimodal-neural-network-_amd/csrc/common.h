// Common definitions for the gfx950 (MI355X / CDNA4) EmbraceNet kernels.
// Wave = 64 lanes, workgroup = 256 threads (4 waves, one per SIMD) everywhere in this library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/embrace_hip.h"

namespace emb {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) double f64x2;
typedef __attribute__((ext_vector_type(4))) double f64x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int kThreads = 256;
constexpr int kWaves = 4;

// thread-local error text behind emb_last_error()
void set_error(const char* fmt, ...);

#define EMB_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      emb::set_error(__VA_ARGS__);               \
      return EMB_ERR_ARG;                        \
    }                                            \
  } while (0)

#define EMB_CHECK_LAUNCH()                                             \
  do {                                                                 \
    hipError_t e__ = hipGetLastError();                                \
    if (e__ != hipSuccess) {                                           \
      emb::set_error("launch failed: %s", hipGetErrorString(e__));     \
      return EMB_ERR_LAUNCH;                                           \
    }                                                                  \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Storage type <-> arithmetic helpers ------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int VEC = 4;  // elements per 16-byte vector
  __device__ static float to_f(float v) { return v; }
  __device__ static float from_f(float v) { return v; }
};
template <> struct Elem<double> {
  static constexpr int VEC = 2;
  __device__ static double to_f(double v) { return v; }
  __device__ static double from_f(double v) { return v; }
};
template <> struct Elem<__bf16> {
  static constexpr int VEC = 8;
  __device__ static float to_f(__bf16 v) { return (float)v; }
  __device__ static __bf16 from_f(float v) { return (__bf16)v; }  // v_cvt_pk_bf16_f32, RNE, NaN-safe
};

// accumulator / "math" type of a storage type: bf16 and f32 accumulate in f32, f64 in f64
template <typename T> struct AccOf { using type = float; };
template <> struct AccOf<double> { using type = double; };

// 16-byte register vector of T
template <typename T> struct Vec16 {
  typedef T type __attribute__((ext_vector_type(16 / sizeof(T))));
};

// XCD-aware, bijective block remap (8 XCDs, round-robin dispatch): blocks that end up on one XCD
// (same id % 8) get a contiguous range of tile ids, so tiles that share operand panels share an L2.
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n >> 3, r = n & 7, x = id & 7, o = id >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
}

}  // namespace emb
