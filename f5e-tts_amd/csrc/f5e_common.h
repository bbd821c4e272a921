// Shared device/host helpers for the F5E hot-path kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#define F5E_STREAM_T hipStream_t
#include "../../include/f5e_abi.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define F5E_WAVE 64

// thread-local last error (f5e_last_error)
void f5e_set_error(const char* fmt, ...);

#define F5E_REQUIRE(cond, ...)                         \
  do {                                                 \
    if (!(cond)) {                                     \
      f5e_set_error(__VA_ARGS__);                      \
      return F5E_ERR_BAD_SHAPE;                        \
    }                                                  \
  } while (0)

// No synchronisation here: only picks up launch-configuration errors, so every op stays graph-capturable.
#define F5E_LAUNCH_CHECK(name)                                                   \
  do {                                                                           \
    hipError_t e_ = hipGetLastError();                                           \
    if (e_ != hipSuccess) {                                                      \
      f5e_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
      return F5E_ERR_HIP;                                                        \
    }                                                                            \
  } while (0)

// A device-side scalar that a PREVIOUS launch wrote (the ODE evaluation counter): read it through the constant address
// space so it becomes an s_load (scalar cache, invalidated at kernel start) instead of a vector load, whose
// s_waitcnt vmcnt(0) would drain every LDS-DMA already in flight and serialise the dependent table loads behind it.
__device__ __forceinline__ int load_uniform_i32(const int* p) {
  return *(const __attribute__((address_space(4))) int*)(uintptr_t)p;
}

__device__ __forceinline__ bf16x4 f2bf4(float a, float b, float c, float d) {
  f32x4 v = {a, b, c, d};
  return __builtin_convertvector(v, bf16x4);
}
__device__ __forceinline__ bf16x2 f2bf2(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_convertvector(v, bf16x2);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float gelu_tanh_f(float x) {
  // 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))); tanh(u) = 1 - 2/(1+exp(2u))
  float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  float t = 1.0f - 2.0f / (1.0f + __expf(2.0f * u));
  return 0.5f * x * (1.0f + t);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float mish_f(float x) {
  // x * tanh(softplus(x)); softplus with torch's threshold 20
  float sp = x > 20.0f ? x : log1pf(__expf(x));
  return x * tanhf(sp);
}
__device__ __forceinline__ float apply_act(float x, int act) {
  switch (act) {
    case F5E_ACT_SILU: return silu_f(x);
    case F5E_ACT_GELU_ERF: return gelu_erf_f(x);
    case F5E_ACT_GELU_TANH: return gelu_tanh_f(x);
    case F5E_ACT_RELU: return fmaxf(x, 0.0f);
    case F5E_ACT_MISH: return mish_f(x);
    default: return x;
  }
}
