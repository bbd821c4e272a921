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

// Per-device host-side caches (a process may drive several GPUs, one per thread): the > 64 KiB dynamic-LDS opt-in is a
// per-device function attribute, and the CU count differs per device.  Lock-free and idempotent: two threads racing on
// the same device both set the attribute, which is harmless.
#include <atomic>
inline int f5e_device_index() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev < 0 ? 0 : (dev > 63 ? 63 : dev);
}
struct F5eDeviceOnce {
  std::atomic<unsigned long long> done{0};
};
// returns hipSuccess, or the error of hipFuncSetAttribute (the launch would fail with "invalid value" otherwise)
inline hipError_t f5e_opt_in_lds(F5eDeviceOnce& once, const void* kernel, int lds_bytes) {
  const unsigned long long bit = 1ull << f5e_device_index();
  if (once.done.load(std::memory_order_acquire) & bit) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e == hipSuccess) once.done.fetch_or(bit, std::memory_order_release);
  return e;
}
#define F5E_OPT_IN_LDS(once, kernel, lds)                                                    \
  do {                                                                                       \
    hipError_t e_ = f5e_opt_in_lds(once, (const void*)(kernel), lds);                        \
    if (e_ != hipSuccess) {                                                                  \
      f5e_set_error("hipFuncSetAttribute(%d B of LDS): %s", (int)(lds), hipGetErrorString(e_)); \
      return F5E_ERR_HIP;                                                                    \
    }                                                                                        \
  } while (0)
inline int f5e_cu_count() {
  static std::atomic<int> cache[64];
  const int dev = f5e_device_index();
  int n = cache[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cache[dev].store(v, std::memory_order_relaxed);
    n = v;
  }
  return n;
}

// Infinity-Cache (MALL) prefetch of the NEXT kernels' weights by a few extra workgroups appended to a launch's grid.
// At batch 1 the 646 MB of block weights cycle through the 256 MB cache, so every GEMM streams its weights from HBM
// while the HBM is >90 % idle; tools/mall_probe.py puts cache-resident weights at -9.7 % per C2 pass.  The extra
// workgroups (blockIdx >= n_main) issue plain 16-byte loads over [ptr, ptr + bytes) and exit: nothing waits for the data
// (s_endpgm drains the counter), the lines stay in the memory-side cache for the kernel that needs them 10-30 us later.
// Results never depend on it.  Internal to the library: the C ABI entry points launch without prefetch.
struct F5ePrefetch {
  const void* ptr[2];
  unsigned bytes[2];
};
// Granule (launches whose workgroups share a CU: attention, the classic GEMM kernels): 32 KiB per workgroup (256 threads x 8
// loads x 16 B); the role-split GEMMs, one workgroup per CU, pack theirs instead (f5e_prefetch_run_packed below).  The prefetch workgroups share their CUs' address path with
// the host launch's own LDS-DMAs, so FEWER, LONGER ones hurt the host (sweep at C2, ms per pass on one box: 256 KiB 48.3,
// 128 KiB 45.4, 64 KiB -- rounds 2-3 -- 43.4, 32 KiB 41.4, 16 KiB 41.6, 8 KiB 41.7, 4 KiB 42.0; prefetch off 45.7).
constexpr int F5E_PF_BYTES_PER_WG = 32 * 1024;
inline int f5e_prefetch_wgs(const F5ePrefetch* pf) {
  if (!pf) return 0;
  unsigned long long b = (pf->ptr[0] ? pf->bytes[0] : 0) + (unsigned long long)(pf->ptr[1] ? pf->bytes[1] : 0);
  return (int)((b + F5E_PF_BYTES_PER_WG - 1) / F5E_PF_BYTES_PER_WG);
}
// workgroup `wg` (0-based among the prefetch workgroups) of an NT-thread launch.  The loads are LDS-DMAs into 4 KiB of
// the workgroup's (otherwise unused) LDS: no destination VGPR, so no register of this wave can be overwritten by a load
// that lands late, and nothing ever reads what arrives.
template <int NT = 256>   // threads of the hosting launch (256, or 128 for the 2-way split attention kernel)
__device__ __forceinline__ void f5e_prefetch_run(const F5ePrefetch& pf, int wg, int tid, void* lds) {
  unsigned long long off = (unsigned long long)wg * F5E_PF_BYTES_PER_WG + (unsigned)tid * 16u;
  const unsigned long long b0 = pf.ptr[0] ? pf.bytes[0] : 0, b1 = pf.ptr[1] ? pf.bytes[1] : 0;
  char* dst = (char*)lds + (tid >> 6) * 1024;   // wave-uniform base; the DMA adds lane * 16
#pragma unroll
  for (int i = 0; i < F5E_PF_BYTES_PER_WG / (NT * 16); ++i, off += NT * 16) {
    const char* p = nullptr;
    if (off + 16 <= b0) p = (const char*)pf.ptr[0] + off;
    else if (off >= b0 && off - b0 + 16 <= b1) p = (const char*)pf.ptr[1] + (off - b0);
    if (p)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  }
}

// The same with a RUN-TIME granule (a multiple of NT * 16 bytes): the role-split GEMMs of a one-round grid occupy one CU per
// workgroup (their LDS ring excludes a second one), so prefetch workgroups cannot start beside them -- they queue for the
// CUs the main grid leaves idle.  Those launches therefore pack the whole prefetch into exactly as many workgroups as there
// are idle CUs (gemm_bf16.hip launch()), each walking a longer slice.
__device__ __forceinline__ void f5e_prefetch_run_packed(const F5ePrefetch& pf, int wg, int tid, int nt, unsigned per_wg,
                                                        void* lds) {
  unsigned long long off = (unsigned long long)wg * per_wg + (unsigned)tid * 16u;
  const unsigned long long end = (unsigned long long)(wg + 1) * per_wg;
  const unsigned long long b0 = pf.ptr[0] ? pf.bytes[0] : 0, b1 = pf.ptr[1] ? pf.bytes[1] : 0;
  char* dst = (char*)lds + (tid >> 6) * 1024;
  for (; off < end; off += (unsigned)nt * 16u) {
    const char* p = nullptr;
    if (off + 16 <= b0) p = (const char*)pf.ptr[0] + off;
    else if (off >= b0 && off - b0 + 16 <= b1) p = (const char*)pf.ptr[1] + (off - b0);
    if (p)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  }
}

// internal entry points with a prefetch hint (gemm_bf16.hip, attention.hip), used by f5e_dit_forward
int f5e_gemm_bf16_bias_pf(hipStream_t st, const void* A, int lda, const void* W, int ldw, const float* bias, void* out,
                          int ldo, int M, int N, int K, int act, int out_f32, int tile_hint, const f5e_ln_fuse* ln,
                          const F5ePrefetch* pf);
int f5e_gemm_bf16_gate_residual_pf(hipStream_t st, const void* A, int lda, const void* W, int ldw, const float* bias,
                                   float* resid, int ldr, const float* gate, int gate_stride, int gate_rows,
                                   const int* eval_ptr, int eval_stride, int rows_per_seq, const int* seq_len, int M,
                                   int N, int K, int tile_hint, const f5e_ln_fuse* ln, const F5ePrefetch* pf);
int f5e_gemm_bf16_qkv_rope_pf(hipStream_t st, const void* A, int lda, const void* W, int ldw, const float* bias, void* q,
                              void* k, void* vt, int n_pad, int heads, int rope_heads, const float* cos_sin,
                              const float* q_norm_w, const float* k_norm_w, int rows_per_seq, int M, int K, int tile_hint,
                              const f5e_ln_fuse* ln, const F5ePrefetch* pf);
int f5e_flash_attn_pf(hipStream_t st, const void* q, const void* k, const void* v, void* o, int ldo, const int* kv_len,
                      int S, int H, int rows_per_seq, int n_pad, int splits, const F5ePrefetch* pf);

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
// Cross-lane adds without the LDS crossbar (__shfl_xor is a ds_bpermute, ~100+ cycles of latency per step on a
// dependent chain): DPP inside a 16-lane row, v_permlane16/32_swap (gfx950) across rows.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float add_xor1(float v) { return v + dpp_f32<0xB1>(v); }    // quad_perm [1,0,3,2]
__device__ __forceinline__ float add_xor2(float v) { return v + dpp_f32<0x4E>(v); }    // quad_perm [2,3,0,1]
// v_permlane16/32_swap exchange halves between two registers (gfx950): lanes 16-31 / 48-63 (resp. 32-63) of vdst swap
// with lanes 0-15 / 32-47 (resp. 0-31) of src.  Started from two copies of v, vdst + src is v + v[lane ^ 16] (resp.
// ^ 32) in every lane.  Inline asm on purpose: with `__builtin_amdgcn_permlane32_swap(u, u, ...)` hipcc 7.2 folded the
// two results into one (it emitted r0 + r0; tools/xl_test.hip catches that on the GPU).  The s_nop covers the
// "VALU write -> v_permlane read" hazard (2 wait states) that the compiler does not see inside asm.
__device__ __forceinline__ void swap16(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap32(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float add_xor16(float v) {  // v + v[lane ^ 16]
  float a = v, b = v;
  swap16(a, b);
  return a + b;
}
__device__ __forceinline__ float add_xor32(float v) {  // v + v[lane ^ 32]
  float a = v, b = v;
  swap32(a, b);
  return a + b;
}
__device__ __forceinline__ float max_xor32(float v) {  // max(v, v[lane ^ 32])
  float a = v, b = v;
  swap32(a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float wave_sum(float v) {
  v = add_xor1(v);
  v = add_xor2(v);
  v += dpp_f32<0x124>(v);  // row_ror:4  -> two quads
  v += dpp_f32<0x128>(v);  // row_ror:8  -> the 16-lane row
  v = add_xor16(v);
  return add_xor32(v);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float gelu_tanh_f(float x) {
  // 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3)  ==  x / (1 + exp(-2u))  ==  x * rcp(1 + exp2(x * p(x^2))),
  // p(t) = -2 sqrt(2/pi) log2(e) (1 + 0.044715 t): 3 mul + 1 fma + 1 add and two quarter-rate ops (v_exp_f32, v_rcp_f32).
  // The epilogue of a 256 x 256 GEMM tile applies this to 128 values per lane with nothing to hide behind, so the IEEE
  // division of the textbook form (~10 instructions) matters; v_rcp_f32 is good to 1 ulp, the result is rounded to bf16.
  const float t = x * x;
  const float p = __builtin_fmaf(t, -2.0f * 0.7978845608028654f * 1.4426950408889634f * 0.044715f,
                                 -2.0f * 0.7978845608028654f * 1.4426950408889634f);
  const float e = __builtin_amdgcn_exp2f(x * p);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float mish_f(float x) {
  // x * tanh(softplus(x)) (softplus with torch's threshold 20).  With e = exp(x): tanh(log(1 + e)) = n / (n + 2),
  // n = e (e + 2) -- all terms positive, no cancellation; 1 exp + 1 rcp instead of libm's log1p and tanh (which cost
  // ~7 us of VALU per conv position embedding launch at batch 1).
  const float e = __builtin_amdgcn_exp2f(fminf(x, 20.0f) * 1.4426950408889634f);
  const float n = e * (e + 2.0f);
  const float r = x * (n * __builtin_amdgcn_rcpf(n + 2.0f));
  return x > 20.0f ? x : r;
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
