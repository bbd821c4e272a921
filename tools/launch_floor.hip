// Measures the cost of N dependent trivial kernels replayed from a hipGraph (one stream) on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void k_empty(float* p) { if (p && threadIdx.x == 1024) p[0] = 1.f; }
__global__ void k_touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
int main() {
  float* d; hipMalloc(&d, 64 << 20);
  hipStream_t st; hipStreamCreate(&st);
  for (int variant = 0; variant < 4; ++variant) {
    int grid = (variant == 0) ? 1 : (variant == 1 ? 256 : (variant == 2 ? 1024 : 1024));
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
    for (int i = 0; i < 160; ++i) {
      if (variant < 3) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, st, d);
      else hipLaunchKernelGGL(k_touch, dim3(4096), dim3(256), 0, st, d, 1 << 20);  // 4 MB RMW
    }
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int w = 0; w < 3; ++w) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    auto t0 = std::chrono::high_resolution_clock::now();
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
    printf("variant %d (grid %d%s): %.2f us per kernel\n", variant, variant < 3 ? grid : 4096, variant == 3 ? ", 4MB RMW" : "", us / reps / 160);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
