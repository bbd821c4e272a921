/* abi_client.c -- a plain-C host of libf5e_hip.so: no Python, no torch, no C++.
 *
 * What a reference-side maintainer binding the C ABI from another language has to do, in the smallest form: own the
 * device buffers (hipMalloc), own the stream, call the entry points of include/f5e_abi.h, check return codes through
 * f5e_last_error().  The program checks three things against loops on the host and prints one line per check:
 *   1. f5e_gemm_f32          C = A . W^T + bias                              (the fp32 linears, model/modules.py:728-730)
 *   2. f5e_gemm_bf16_bias    the same on bf16 operands, fp32 accumulation     (ff project_in, model/modules.py:348)
 *   3. f5e_ode_update        CFG combine + Euler step, captured ONCE with f5e_graph_begin / _end and replayed over a
 *                            4-step grid by f5e_graph_launch (model/cfm.py:447 + torchdiffeq euler; the per-step dt comes
 *                            from a device table indexed by the device-side evaluation counter)
 *
 * Build (tests/test_c_host.py does exactly this):
 *   gcc -std=c11 -O1 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c_host/abi_client.c \
 *       -L f5e-tts_amd -lf5e_hip -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/f5e-tts_amd -Wl,-rpath,/opt/rocm/lib
 * Exit code 0 = every check passed. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "f5e_abi.h"

#define HIP_OK(call)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                          \
      return 2;                                                                           \
    }                                                                                     \
  } while (0)
#define F5E_TRY(call)                                                                     \
  do {                                                                                    \
    int rc_ = (call);                                                                     \
    if (rc_ != F5E_OK) {                                                                  \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, f5e_last_error());              \
      return 3;                                                                           \
    }                                                                                     \
  } while (0)

static uint32_t lcg_state = 12345u;
static float frand(void) { /* uniform in [-1, 1) */
  lcg_state = lcg_state * 1664525u + 1013904223u;
  return (float)(lcg_state >> 8) * (1.0f / 8388608.0f) - 1.0f;
}
static uint16_t f32_to_bf16(float f) { /* round to nearest even */
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf16_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main(void) {
  if (f5e_abi_version() != F5E_ABI_VERSION) {
    fprintf(stderr, "header says ABI %d, library says %d\n", F5E_ABI_VERSION, f5e_abi_version());
    return 1;
  }
  HIP_OK(hipSetDevice(0));
  F5E_TRY(f5e_check_device());
  hipStream_t st;
  HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  int failed = 0;

  /* ---- 1 + 2: C[M][N] = A[M][K] . W[N][K]^T + bias, fp32 and bf16 operands ---- */
  enum { M = 70, N = 192, K = 128 };
  static float A[M * K], W[N * K], bias[N], C[M * N];
  static uint16_t Ah[M * K], Wh[N * K], Ch[M * N];
  for (int i = 0; i < M * K; ++i) { A[i] = frand(); Ah[i] = f32_to_bf16(A[i]); }
  for (int i = 0; i < N * K; ++i) { W[i] = frand() * 0.1f; Wh[i] = f32_to_bf16(W[i]); }
  for (int i = 0; i < N; ++i) bias[i] = frand();
  float *dA, *dW, *dB, *dC;
  void *dAh, *dWh, *dCh;
  HIP_OK(hipMalloc((void**)&dA, sizeof A)); HIP_OK(hipMalloc((void**)&dW, sizeof W));
  HIP_OK(hipMalloc((void**)&dB, sizeof bias)); HIP_OK(hipMalloc((void**)&dC, sizeof C));
  HIP_OK(hipMalloc(&dAh, sizeof Ah)); HIP_OK(hipMalloc(&dWh, sizeof Wh)); HIP_OK(hipMalloc(&dCh, sizeof Ch));
  HIP_OK(hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(dW, W, sizeof W, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dB, bias, sizeof bias, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dAh, Ah, sizeof Ah, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(dWh, Wh, sizeof Wh, hipMemcpyHostToDevice));

  F5E_TRY(f5e_gemm_f32(st, dA, K, M, F5E_ACT_NONE, dW, K, dB, F5E_ACT_NONE, NULL, NULL, 0, 0, NULL, dC, N, NULL, 0, M, N, K));
  F5E_TRY(f5e_gemm_bf16_bias(st, dAh, K, dWh, K, dB, dCh, N, M, N, K, F5E_ACT_NONE, 0, 0));
  HIP_OK(hipStreamSynchronize(st));
  HIP_OK(hipMemcpy(C, dC, sizeof C, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(Ch, dCh, sizeof Ch, hipMemcpyDeviceToHost));
  double err32 = 0.0, err16 = 0.0;
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double s32 = bias[n], s16 = bias[n];
      for (int k = 0; k < K; ++k) {
        s32 += (double)A[m * K + k] * W[n * K + k];
        s16 += (double)bf16_to_f32(Ah[m * K + k]) * bf16_to_f32(Wh[n * K + k]);
      }
      const double e32 = fabs(C[m * N + n] - s32), e16 = fabs(bf16_to_f32(Ch[m * N + n]) - s16) / (1.0 + fabs(s16));
      if (e32 > err32) err32 = e32;
      if (e16 > err16) err16 = e16;
    }
  printf("f5e_gemm_f32        max abs err %.3e (tolerance 1e-5)\n", err32);
  printf("f5e_gemm_bf16_bias  max rel err %.3e (tolerance 2^-8: one bf16 rounding of the output)\n", err16);
  failed += !(err32 < 1e-5) + !(err16 < 1.0 / 256);

  /* ---- 3: four Euler steps with classifier-free guidance, one captured step replayed ---- */
  enum { NE = 4096, STEPS = 4 };
  static float y[NE], pred[2 * NE], yref[NE];
  const float dt[STEPS] = {0.1f, 0.2f, 0.3f, 0.4f}, cfg = 2.0f;
  for (int i = 0; i < NE; ++i) { y[i] = frand(); yref[i] = y[i]; }
  for (int i = 0; i < 2 * NE; ++i) pred[i] = frand();
  float *dY, *dP, *dDt;
  int* dEval;
  unsigned* dDone;
  HIP_OK(hipMalloc((void**)&dY, sizeof y)); HIP_OK(hipMalloc((void**)&dP, sizeof pred)); HIP_OK(hipMalloc((void**)&dDt, sizeof dt));
  HIP_OK(hipMalloc((void**)&dEval, 4)); HIP_OK(hipMalloc((void**)&dDone, 4));
  HIP_OK(hipMemcpy(dY, y, sizeof y, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(dP, pred, sizeof pred, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dDt, dt, sizeof dt, hipMemcpyHostToDevice));
  HIP_OK(hipMemset(dEval, 0, 4)); HIP_OK(hipMemset(dDone, 0, 4));
  void* graph = NULL;
  F5E_TRY(f5e_graph_begin(st));
  /* mode 1: v = p0 + (p0 - p1) * w0   (cond + (cond - uncond) * cfg_strength, model/cfm.py:447);  y += dt[*eval] * v */
  int rc = f5e_ode_update(st, dP, NE, 1, cfg, 0.0f, dY, dY, NULL, dDt, dEval, dDone, NE);
  F5E_TRY(f5e_graph_end(st, &graph));   /* end the capture even when the op refused its arguments */
  if (rc != F5E_OK) { fprintf(stderr, "f5e_ode_update failed (%d): %s\n", rc, f5e_last_error()); return 3; }
  for (int s = 0; s < STEPS; ++s) F5E_TRY(f5e_graph_launch(graph, st));
  HIP_OK(hipStreamSynchronize(st));
  HIP_OK(hipMemcpy(y, dY, sizeof y, hipMemcpyDeviceToHost));
  int evals = -1;
  HIP_OK(hipMemcpy(&evals, dEval, 4, hipMemcpyDeviceToHost));
  double erry = 0.0;
  for (int i = 0; i < NE; ++i) {
    const float v = pred[i] + (pred[i] - pred[NE + i]) * cfg;
    for (int s = 0; s < STEPS; ++s) yref[i] = yref[i] + dt[s] * v;
    const double e = fabs(y[i] - yref[i]);
    if (e > erry) erry = e;
  }
  printf("f5e_ode_update x %d (one captured step, replayed)  max abs err %.3e (tolerance 1e-5), evaluation counter %d\n",
         STEPS, erry, evals);
  failed += !(erry < 1e-5) + (evals != STEPS);
  F5E_TRY(f5e_graph_destroy(graph));

  /* a refused call leaves a message and no launch */
  if (f5e_gemm_bf16_bias(st, dAh, K, dWh, K, dB, dCh, N, M, N, 100 /* K % 64 != 0 */, F5E_ACT_NONE, 0, 0) == F5E_OK) {
    fprintf(stderr, "a K that is not a multiple of 64 was accepted\n");
    ++failed;
  } else {
    printf("refused as documented: %s\n", f5e_last_error());
  }
  HIP_OK(hipStreamDestroy(st));
  printf(failed ? "FAILED (%d)\n" : "OK\n", failed);
  return failed ? 1 : 0;
}
