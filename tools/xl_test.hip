// GPU unit test of the cross-lane helpers in f5e_common.h (DPP / v_permlane16|32_swap) against a host loop.
#include "../f5e-tts_amd/csrc/f5e_common.h"
#include <cstdio>
#include <cmath>
void f5e_set_error(const char*, ...) {}
__global__ void k(const float* in, float* out) {
  const int l = threadIdx.x;
  float v = in[l];
  out[l] = add_xor32(v);
  out[64 + l] = max_xor32(v);
  out[128 + l] = add_xor16(v);
  out[192 + l] = add_xor1(v);
  out[256 + l] = add_xor2(v);
  out[320 + l] = wave_sum(v);
  float m = v * 2.0f;            // VALU write right before the swap
  out[384 + l] = max_xor32(m);
}
int main() {
  float h[64], o[448]; for (int i = 0; i < 64; ++i) h[i] = (i % 7 == 3) ? -INFINITY : (float)(i * i % 23) - 5.f;
  float *d, *e; hipMalloc(&d, 256); hipMalloc(&e, sizeof(o)); hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e); hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) {
    auto eq = [](float a, float b) { return a == b || (std::isinf(a) && std::isinf(b)); };
    if (!eq(o[i], h[i] + h[i ^ 32])) { bad++; printf("add32 lane %d got %f want %f\n", i, o[i], h[i] + h[i ^ 32]); }
    if (!eq(o[64 + i], fmaxf(h[i], h[i ^ 32]))) { bad++; printf("max32 lane %d got %f want %f\n", i, o[64 + i], fmaxf(h[i], h[i ^ 32])); }
    if (!eq(o[128 + i], h[i] + h[i ^ 16])) { bad++; printf("add16 lane %d\n", i); }
    if (!eq(o[192 + i], h[i] + h[i ^ 1])) { bad++; printf("add1 lane %d\n", i); }
    if (!eq(o[256 + i], h[i] + h[i ^ 2])) { bad++; printf("add2 lane %d\n", i); }
    if (!eq(o[384 + i], fmaxf(2 * h[i], 2 * h[i ^ 32]))) { bad++; printf("max32b lane %d got %f\n", i, o[384 + i]); }
  }
  printf("bad=%d wave_sum=%f\n", bad, o[320]);
  return bad != 0;
}
