// How fast can ONE CU pull L2-resident bytes, by access type?  (gfx950; diagnostic, not part of the library.)
//   mode 0: LDS-DMA global_load_lds_dwordx4 (what the GEMM rings use)     mode 1: global_load_dwordx4 -> VGPR (8 in flight per lane)
//   mode 2: mode 1 + ds_write_b128 of every vector (register-staged ring fill)
// Every workgroup (256 threads) sweeps its own 256 KiB region `passes` times; regions repeat with period `nreg` workgroups so
// that an XCD's working set (nreg / 8 regions x 256 KiB) stays inside its 4 MiB L2 while a CU's 32 KiB L1 never holds a region.
// usage: l2_probe.bin <mode> <wgs_per_cu> <nreg> [passes]     prints B/clk/CU from in-kernel s_memtime and GB/s per CU from wall time
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/l2_probe.hip -o tools/l2_probe.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr int REGION = 256 * 1024;

template <int MODE>
__global__ __launch_bounds__(256) void probe(const char* base, int nreg, int passes, unsigned long long* out, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6;
  const char* reg = base + (size_t)(blockIdx.x % nreg) * REGION;
  u32x4 acc = {0, 0, 0, 0};
  u32x4 nxt[8];
  if (MODE != 0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) nxt[u] = *(const u32x4*)(reg + (size_t)u * 4096 + tid * 16);
  }
  unsigned long long t0 = 0;
  for (int p = 0; p < passes; ++p) {
    if (p == 1) { __syncthreads(); t0 = __builtin_amdgcn_s_memtime(); }
    // 256 KiB = 64 steps of 4 KiB per workgroup (16 B per thread per step)
    for (int s = 0; s < 64; s += 8) {
      if (MODE == 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reg + (size_t)(s + u) * 4096 + tid * 16),
                                           (__attribute__((address_space(3))) void*)(smem + ((s + u) & 15) * 4096 + wave * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // one group of 8 stays in flight while the next is issued
      } else {
        // two groups of 8 vectors: group g + 1 is requested before group g is consumed (16 loads per lane in flight)
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = nxt[u];
        const int sn = (s + 8) & 63;
#pragma unroll
        for (int u = 0; u < 8; ++u) nxt[u] = *(const u32x4*)(reg + (size_t)(sn + u) * 4096 + tid * 16);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (MODE == 2) *(u32x4*)(smem + ((s + u) & 15) * 4096 + tid * 16) = v[u];
          else acc ^= v[u];
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) out[blockIdx.x] = t1 - t0;
  if (MODE == 2) acc = *(u32x4*)(smem + tid * 16);
  if (acc[0] == 0x12345678u && acc[1] == 0x9abcdefu) sink[0] = acc[2] ^ acc[3];   // keeps the loads alive
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0, per_cu = argc > 2 ? atoi(argv[2]) : 1, nreg = argc > 3 ? atoi(argv[3]) : 64;
  const int passes = argc > 4 ? atoi(argv[4]) : 9;
  const int grid = 256 * per_cu;
  char* base; unsigned long long* out; unsigned* sink;
  hipMalloc(&base, (size_t)nreg * REGION); hipMemset(base, 1, (size_t)nreg * REGION);
  hipMalloc(&out, grid * 8); hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 65536, 0, base, nreg, passes, out, sink);
    else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 65536, 0, base, nreg, passes, out, sink);
    else hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(256), 65536, 0, base, nreg, passes, out, sink);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<unsigned long long> h(grid);
  hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double bytes = (double)(passes - 1) * REGION;
  printf("mode %d wgs/cu %d nreg %d (%.1f MiB per XCD): cycles per warm pass median %.0f -> %.1f B/clk per workgroup, %.1f B/clk per CU | wall %.1f us, all passes %.1f GB/s per CU\n",
         mode, per_cu, nreg, nreg / 8.0 * REGION / 1048576.0, (double)h[grid / 2] / (passes - 1), bytes / h[grid / 2], per_cu * bytes / h[grid / 2],
         ms * 1e3, (double)passes * REGION * per_cu / (ms * 1e-3) / 1e9);
  return 0;
}
