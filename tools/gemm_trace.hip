// Per-workgroup timeline of the bf16 GEMM kernel (diagnostic build, DBG >= 3 stamps) inside a chain of dependent
// launches replayed from a hipGraph.  Answers: where do the microseconds of a small-M launch go (launch gap, start
// ramp, first tile, K loop, epilogue)?    usage: gemm_trace.bin <variant> <M> <N> <K> <epi: 0 bf16, 1 gelu, 2 gate_res, 3 qkv+rope (2 sequences)>
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/gemm_trace.hip f5e-tts_amd/csrc/gemm_bf16_pp.hip -o tools/gemm_trace.bin
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../f5e-tts_amd/csrc/gemm_bf16.hip"

void f5e_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}

template <int EPI>
static int run_variant(int v, GemmArgs& a, hipStream_t st, int* bm, int* bn) {
  switch (v) {
    case 0: *bm = 64; *bn = 64; return launch<64, 64, EPI, 3, 2, 2, 3>(a, st);
    case 1: *bm = 128; *bn = 64; return launch<128, 64, EPI, 3, 2, 2, 3>(a, st);
    case 2: *bm = 128; *bn = 128; return launch<128, 128, EPI, 2, 4, 2, 3>(a, st);
    case 3: *bm = 64; *bn = 64; return launch<64, 64, EPI, 4, 2, 2, 3>(a, st);
    case 4: *bm = 128; *bn = 64; return launch<128, 64, EPI, 4, 2, 2, 3>(a, st);
    case 5: *bm = 64; *bn = 64; return launch<64, 64, EPI, 2, 2, 2, 3>(a, st);
    case 6: *bm = 64; *bn = 64; return launch<64, 64, EPI, 4, 2, 4, 3>(a, st);   // 8 waves (2 x 4), 4-stage ring
    case 7: *bm = 64; *bn = 64; return launch<64, 64, EPI, 3, 2, 4, 3>(a, st);   // 8 waves, 3-stage ring
    case 8: *bm = 64; *bn = 64; return launch<64, 64, EPI, 4, 4, 2, 3>(a, st);   // 8 waves (4 x 2), 4-stage ring
    // role split (NLOAD loader waves + 4 consumer waves): template order BM, BN, EPI, NSTAGE, WGM, WGN, DBG, FUSE, NLOAD
    case 30: *bm = 64; *bn = 64; return launch<64, 64, EPI, 3, 2, 2, 3, 0, 1>(a, st);
    case 31: *bm = 64; *bn = 64; return launch<64, 64, EPI, 3, 2, 2, 3, 0, 2>(a, st);
    case 32: *bm = 64; *bn = 64; return launch<64, 64, EPI, 4, 2, 2, 3, 0, 1>(a, st);
    case 33: *bm = 64; *bn = 64; return launch<64, 64, EPI, 4, 2, 2, 3, 0, 2>(a, st);
    case 34: *bm = 64; *bn = 64; return launch<64, 64, EPI, 5, 2, 2, 3, 0, 1>(a, st);
    case 35: *bm = 128; *bn = 64; return launch<128, 64, EPI, 3, 2, 2, 3, 0, 2>(a, st);
    case 36: *bm = 128; *bn = 128; return launch<128, 128, EPI, 3, 4, 2, 3, 0, 2>(a, st);
    case 37: *bm = 128; *bn = 128; return launch<128, 128, EPI, 3, 2, 2, 3, 0, 2>(a, st);   // 4 consumers, 64 x 64 each
    case 38: *bm = 64; *bn = 64; return launch<64, 64, EPI, 3, 2, 2, 3, 0, 4>(a, st);      // 4 loaders + 4 consumers
    case 39: *bm = 64; *bn = 64; return launch<64, 64, EPI, 4, 2, 2, 3, 0, 4>(a, st);
    case 50: *bm = 128; *bn = 128; return launch<128, 128, EPI, 3, 4, 2, 3, 0, 4>(a, st);  // 8 consumers (32 x 64 each)
    case 51: *bm = 128; *bn = 64; return launch<128, 64, EPI, 3, 2, 2, 3, 0, 4>(a, st);    // 4 consumers (64 x 32 each)
    case 52: *bm = 64; *bn = 128; return launch<64, 128, EPI, 3, 2, 2, 3, 0, 4>(a, st);    // 4 consumers (32 x 64 each)
    case 53: *bm = 128; *bn = 128; return launch<128, 128, EPI, 3, 2, 2, 3, 0, 4>(a, st);  // 4 consumers (64 x 64 each)
    case 54: *bm = 128; *bn = 64; return launch<128, 64, EPI, 4, 2, 2, 3, 0, 4>(a, st);
    // 8 consumer waves (32 x 32 sub-tiles, the per-wave epilogue work of the classic 64 x 64 tile) + 4 loaders on bigger tiles
    case 60: *bm = 128; *bn = 64; return launch<128, 64, EPI, 3, 4, 2, 3, 0, 4>(a, st);
    case 61: *bm = 64; *bn = 128; return launch<64, 128, EPI, 3, 2, 4, 3, 0, 4>(a, st);
    case 62: *bm = 128; *bn = 64; return launch<128, 64, EPI, 4, 4, 2, 3, 0, 4>(a, st);
    case 63: *bm = 96; *bn = 128; return launch<96, 128, EPI, 3, 2, 4, 3, 0, 4>(a, st);    // 48 x 32 sub-tiles
    case 65: *bm = 96; *bn = 128; return launch<96, 128, EPI, 4, 2, 4, 3, 0, 4>(a, st);
    case 70: *bm = 64; *bn = 192; return launch<64, 192, EPI, 3, 2, 6, 3, 0, 4>(a, st);   // 12 consumers + 4 loaders
    case 71: *bm = 64; *bn = 192; return launch<64, 192, EPI, 4, 2, 6, 3, 0, 4>(a, st);
    case 72: *bm = 64; *bn = 128; return launch<64, 128, EPI, 4, 2, 4, 3, 0, 4>(a, st);   // 8 consumers + 4 loaders, 4 stages
    case 73:  // fused AdaLN consumers on the wide role-split tiles (bf16 / gelu / qkv epilogues only)
      if constexpr (EPI != EPI_GATE_RES) { *bm = 64; *bn = 192; return launch<64, 192, EPI, 3, 2, 6, 3, 1, 4>(a, st); }
      return -1;
    case 74:
      if constexpr (EPI != EPI_GATE_RES) { *bm = 64; *bn = 192; return launch<64, 192, EPI, 4, 2, 6, 3, 1, 4>(a, st); }
      return -1;
    case 75:
      if constexpr (EPI != EPI_GATE_RES) { *bm = 64; *bn = 128; return launch<64, 128, EPI, 3, 2, 4, 3, 1, 4>(a, st); }
      return -1;
    case 76:
      if constexpr (EPI != EPI_GATE_RES) { *bm = 64; *bn = 128; return launch<64, 128, EPI, 4, 2, 4, 3, 1, 4>(a, st); }
      return -1;
    // two tiles per hand-over barrier (DSTEP 2), ring of 6 tiles
    case 80: *bm = 64; *bn = 64; return launch<64, 64, EPI, 6, 2, 2, 3, 0, 4, 2>(a, st);
    case 81:
      *bm = 64; *bn = 64;
      if constexpr (EPI == EPI_GATE_RES) return launch<64, 64, EPI, 6, 2, 2, 3, 2, 4, 2>(a, st);
      else return launch<64, 64, EPI, 6, 2, 2, 3, 1, 4, 2>(a, st);
    case 82: *bm = 64; *bn = 128; return launch<64, 128, EPI, 6, 2, 4, 3, 0, 4, 2>(a, st);
    case 83:
      if constexpr (EPI != EPI_GATE_RES) { *bm = 64; *bn = 128; return launch<64, 128, EPI, 6, 2, 4, 3, 1, 4, 2>(a, st); }
      return -1;
    // 128-row role-split tiles (M in (1024, 2048]): 90 the fused producer / consumers as dispatched, 93 the producer without
    // the AdaLN fusion, 94 the 128 x 128 consumer with two full fragment sets
    case 90:
      if constexpr (EPI == EPI_GATE_RES) { *bm = 128; *bn = 64; return launch<128, 64, EPI, 6, 4, 2, 3, 2, 4, 2>(a, st); }
      else if constexpr (EPI == EPI_QKV_ROPE) { *bm = 128; *bn = 192; return launch<128, 192, EPI, 3, 2, 6, 3, 1, 4, 1, 0>(a, st); }
      else { *bm = 128; *bn = 128; return launch<128, 128, EPI, 4, 2, 4, 3, 1, 4, 1, 0>(a, st); }
    // 128-row tiles without the AdaLN fusion: fat consumer waves (64 x 64) vs 64 x 32 ones
    case 96: *bm = 128; *bn = 128; return launch<128, 128, EPI, 4, 2, 2, 3, 0, 4, 1, 1>(a, st);   // 4 consumers, two fragment sets
    case 97: *bm = 128; *bn = 128; return launch<128, 128, EPI, 4, 2, 2, 3, 0, 4, 1, 0>(a, st);   // 4 consumers, half-tile pipelined
    case 98: *bm = 128; *bn = 192; return launch<128, 192, EPI, 4, 2, 3, 3, 0, 4, 1, 0>(a, st);   // 6 consumers of 64 x 64
    case 100: *bm = 128; *bn = 128; return launch<128, 128, EPI, 4, 2, 4, 3, 0, 4, 1, 0>(a, st);  // 8 consumers of 64 x 32
    case 101: *bm = 128; *bn = 192; return launch<128, 192, EPI, 4, 2, 6, 3, 0, 4, 1, 0>(a, st);  // 12 consumers of 64 x 32
    case 102: *bm = 128; *bn = 128; return launch<128, 128, EPI, 4, 2, 4, 4, 0, 4, 1, 0>(a, st);  // 100 without MFMAs
    case 103: *bm = 128; *bn = 128; return launch<128, 128, EPI, 4, 2, 4, 5, 0, 4, 1, 0>(a, st);  // 100 without reads and MFMAs
    case 93: *bm = 128; *bn = 64; return launch<128, 64, EPI, 6, 4, 2, 3, 0, 4, 2>(a, st);
    case 94:
      if constexpr (EPI == EPI_BF16 || EPI == EPI_BF16_GELU) { *bm = 128; *bn = 128; return launch<128, 128, EPI, 4, 2, 4, 3, 1, 4, 1, 1>(a, st); }
      return -1;
    case 42:  // fused AdaLN, 4 loaders, 4 stages (what the dispatcher launches for the one-round producers)
      *bm = 64; *bn = 64;
      if constexpr (EPI == EPI_GATE_RES) return launch<64, 64, EPI, 4, 2, 2, 3, 2, 4>(a, st);
      else return launch<64, 64, EPI, 4, 2, 2, 3, 1, 4>(a, st);
    case 41:  // fused AdaLN, 4 loaders
      *bm = 64; *bn = 64;
      if constexpr (EPI == EPI_GATE_RES) return launch<64, 64, EPI, 3, 2, 2, 3, 2, 4>(a, st);
      else return launch<64, 64, EPI, 3, 2, 2, 3, 1, 4>(a, st);
    case 40:  // fused AdaLN, role split
      *bm = 64; *bn = 64;
      if constexpr (EPI == EPI_GATE_RES) return launch<64, 64, EPI, 3, 2, 2, 3, 2, 1>(a, st);
      else return launch<64, 64, EPI, 3, 2, 2, 3, 1, 1>(a, st);
    case 10:  // fused AdaLN: consumer for the bf16 / gelu epilogues, producer for gate + residual
      *bm = 64; *bn = 64;
      if constexpr (EPI == EPI_GATE_RES) return launch<64, 64, EPI, 3, 2, 2, 3, 2>(a, st);
      else return launch<64, 64, EPI, 3, 2, 2, 3, 1>(a, st);
  }
  return -1;
}

static bool is_fused(int variant) {
  return variant == 10 || (variant >= 40 && variant <= 42) || (variant >= 73 && variant <= 76) || variant == 81 || variant == 83 ||
         variant == 90 || variant == 94;
}

static double med(std::vector<double> v) {
  if (v.empty()) return 0;
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main(int argc, char** argv) {
  const int variant = argc > 1 ? atoi(argv[1]) : 0;
  const int M = argc > 2 ? atoi(argv[2]) : 938, N = argc > 3 ? atoi(argv[3]) : 1024, K = argc > 4 ? atoi(argv[4]) : 1024;
  const int epi = argc > 5 ? atoi(argv[5]) : 2;
  const int NW = 22, L = 44;
  bf16 *A, *W, *out;
  float *bias, *resid, *gate;
  unsigned long long* trace;
  const size_t max_grid = 4096;
  hipMalloc(&A, (size_t)M * K * 2);
  hipMalloc(&W, (size_t)NW * N * K * 2);
  hipMalloc(&out, (size_t)M * N * 2);
  hipMalloc(&bias, N * 4);
  hipMalloc(&gate, N * 4);
  hipMalloc(&resid, (size_t)M * N * 4);
  hipMalloc(&trace, (size_t)L * max_grid * 48 * 8);
  float *stats, *cd, *row_mean;
  hipMalloc(&row_mean, (size_t)M * 4);
  hipMemset(row_mean, 0, (size_t)M * 4);
  hipMalloc(&stats, (size_t)M * 32 * 2 * 4);
  hipMalloc(&cd, (size_t)2 * N * 4);
  hipMemset(stats, 0, (size_t)M * 32 * 2 * 4);
  hipMemset(cd, 0, (size_t)2 * N * 4);
  // qkv + rope: two sequences of M / 2 rows, heads = N / 192
  const int rps = epi == 3 ? M / 2 : M, heads = N / 192, n_pad = (rps + 63) / 64 * 64;
  bf16 *qb = nullptr, *kb = nullptr, *vtb = nullptr;
  float* cs = nullptr;
  if (epi == 3) {
    const size_t hb = (size_t)2 * heads * n_pad * 64 * 2;
    hipMalloc(&qb, hb); hipMalloc(&kb, hb); hipMalloc(&vtb, hb);
    hipMalloc(&cs, (size_t)rps * 64 * 4);
    hipMemset(cs, 0, (size_t)rps * 64 * 4);
  }
  {  // small random-ish fill (values do not matter for timing, but keep them finite)
    std::vector<unsigned short> h((size_t)NW * N * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (unsigned short)((i * 2654435761u) >> 25);  // ~[0.0078, 0.0156]
    hipMemcpy(W, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice);
    hipMemset(bias, 0, N * 4);
    hipMemset(gate, 0, N * 4);
    hipMemset(resid, 0, (size_t)M * N * 4);
  }
  hipStream_t st;
  hipStreamCreate(&st);
  int bm = 0, bn = 0, grid = 0;
  auto one = [&](int l) {
    GemmArgs a{};
    a.A = A; a.lda = K; a.W = W + (size_t)(l % NW) * N * K; a.ldw = K; a.bias = bias;
    a.M = M; a.N = N; a.K = K; a.out = out; a.ldo = N;
    a.resid = resid; a.ldr = N; a.gate = gate; a.gate_stride = 0; a.gate_rows = 1; a.rows_per_seq = rps;
    if (epi == 3) { a.q = qb; a.k = kb; a.vt = vtb; a.n_pad = n_pad; a.heads = heads; a.rope_heads = heads; a.cos_sin = cs; }
    a.trace = trace + (size_t)l * max_grid * 48;
    if (is_fused(variant)) {
      a.row_mean = row_mean;
      if (epi == 2) { a.xs_out = out; a.ld_xs = N; a.next_scale = gate; a.stats_out = stats; }
      else { a.ln_stats = stats; a.ln_parts = K / 64; a.ln_c = cd; a.ln_d = cd + N; a.cd_stride = 2 * N; a.cd_rows = 1;
             a.ln_eps = 1e-6f; a.bias = nullptr; }
    }
    int rc = epi == 3 ? run_variant<EPI_QKV_ROPE>(variant, a, st, &bm, &bn) : epi == 2 ? run_variant<EPI_GATE_RES>(variant, a, st, &bm, &bn)
                      : (epi == 1 ? run_variant<EPI_BF16_GELU>(variant, a, st, &bm, &bn) : run_variant<EPI_BF16>(variant, a, st, &bm, &bn));
    grid = a.tiles_m * a.tiles_n;
    return rc;
  };
  if (one(0) != 0 || (size_t)grid > max_grid) { fprintf(stderr, "bad variant / grid\n"); return 1; }
  hipStreamSynchronize(st);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
  for (int l = 0; l < L; ++l) one(l);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int r = 0; r < 4; ++r) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  std::vector<unsigned long long> h((size_t)L * max_grid * 48);
  hipMemcpy(h.data(), trace, h.size() * 8, hipMemcpyDeviceToHost);
  const int KT = K / 64;
  std::vector<double> gap, ramp, span, first, iter, loop, epil, wg_total, clk, issue, dmaissue;
  for (int l = 12; l < L; ++l) {
    auto T = [&](int wg, int slot) { return h[((size_t)l * max_grid + wg) * 48 + slot]; };
    auto P = [&](int wg, int slot) { return h[((size_t)(l - 1) * max_grid + wg) * 48 + slot]; };
    unsigned long long s_min = ~0ull, s_max = 0, e_max = 0, pe_max = 0;
    for (int w = 0; w < grid; ++w) {
      s_min = std::min(s_min, T(w, 0)); s_max = std::max(s_max, T(w, 0)); e_max = std::max(e_max, T(w, 41));
      pe_max = std::max(pe_max, P(w, 41));
      first.push_back((double)(T(w, 4) - T(w, 1)));
      issue.push_back((double)(T(w, 2) - T(w, 1)));
      dmaissue.push_back((double)(T(w, 46) - T(w, 1)));
      iter.push_back((double)(T(w, 4 + std::min(KT, 36) - 1) - T(w, 4)) / (std::min(KT, 36) - 1));
      loop.push_back((double)(T(w, 3) - T(w, 4)));
      epil.push_back((double)(T(w, 40) - T(w, 3)));
      wg_total.push_back((double)(T(w, 40) - T(w, 1)));
      const double rt = (double)(T(w, 41) - T(w, 0));
      if (rt > 0) clk.push_back((double)(T(w, 40) - T(w, 1)) / rt * 100.0);  // MHz
    }
    gap.push_back(((double)s_min - (double)pe_max) * 10.0);
    ramp.push_back((double)(s_max - s_min) * 10.0);
    span.push_back((double)(e_max - s_min) * 10.0);
  }
  const double mhz = med(clk);
  printf("variant %d tile %dx%d M=%d N=%d K=%d epi=%d grid=%d  shader clock ~%.0f MHz\n", variant, bm, bn, M, N, K, epi, grid, mhz);
  printf("  per launch [ns]: gap after previous kernel's last end -> first start %.0f | start ramp (first->last workgroup) %.0f | span first start -> last end %.0f\n",
         med(gap), med(ramp), med(span));
  printf("  per workgroup [cycles, median]: entry->prologue DMAs issued %.0f | +prefetch issued %.0f | entry->tile0 landed %.0f | K-loop iteration %.0f (x%d) | loop %.0f | epilogue %.0f | total %.0f (= %.2f us)\n",
         med(dmaissue), med(issue), med(first), med(iter), KT, med(loop), med(epil), med(wg_total), med(wg_total) / mhz);
  if (variant >= 30) {  // role split: when the loader's prologue DMAs were all issued (slot 47, loader thread 0)
    std::vector<double> li;
    for (int l = 12; l < L; ++l)
      for (int w = 0; w < grid; ++w) {
        const unsigned long long* t = &h[((size_t)l * max_grid + w) * 48];
        li.push_back((double)(t[47] - t[1]));
      }
    printf("  loader: entry -> prologue DMAs issued %.0f cycles\n", med(li));
  }
  {  // per-K-step profile (median over workgroups of the last launch): cycles from step kt's barrier to step kt+1's
    const int l = L - 1, nk = std::min(KT, 36);
    printf("  K-step profile [cycles]:");
    for (int kt = 0; kt + 1 < nk; ++kt) {
      std::vector<double> d;
      for (int w = 0; w < grid; ++w) {
        const unsigned long long* t = &h[((size_t)l * max_grid + w) * 48];
        d.push_back((double)(t[4 + kt + 1] - t[4 + kt]));
      }
      printf(" %.0f", med(d));
    }
    printf("\n");
  }
  if (is_fused(variant)) {  // fused AdaLN epilogue split (cycles after the K loop): slots 42..45, see gemm_bf16.hip
    std::vector<double> a42, a43, a44, a45, a40;
    const int l = L - 1;
    for (int w = 0; w < grid; ++w) {
      const unsigned long long* t = &h[((size_t)l * max_grid + w) * 48];
      a42.push_back((double)(t[42] - t[3])); a43.push_back((double)(t[43] - t[3])); a44.push_back((double)(t[44] - t[3]));
      a45.push_back((double)(t[45] - t[3])); a40.push_back((double)(t[40] - t[3]));
    }
    if (epi == 2)
      printf("  producer epilogue: stores issued at +%.0f | statistics in LDS +%.0f | barrier passed +%.0f | all stores acknowledged +%.0f\n",
             med(a43), med(a44), med(a45), med(a40));
    else
      printf("  consumer epilogue: statistics reduced +%.0f | stores issued +%.0f | all stores acknowledged +%.0f\n",
             med(a42), med(a43), med(a40));
  }
  {  // distribution of per-workgroup start offsets and totals for the last launch
    const int l = L - 1;
    std::vector<double> so, tot;
    unsigned long long s_min = ~0ull;
    for (int w = 0; w < grid; ++w) s_min = std::min(s_min, h[((size_t)l * max_grid + w) * 48]);
    for (int w = 0; w < grid; ++w) {
      so.push_back((double)(h[((size_t)l * max_grid + w) * 48] - s_min) * 10.0);
      tot.push_back((double)(h[((size_t)l * max_grid + w) * 48 + 41] - h[((size_t)l * max_grid + w) * 48]) * 10.0);
    }
    std::sort(so.begin(), so.end()); std::sort(tot.begin(), tot.end());
    printf("  last launch: start offset ns p10 %.0f p50 %.0f p90 %.0f max %.0f | workgroup lifetime ns p10 %.0f p50 %.0f p90 %.0f max %.0f\n",
           so[grid / 10], so[grid / 2], so[grid * 9 / 10], so[grid - 1], tot[grid / 10], tot[grid / 2], tot[grid * 9 / 10], tot[grid - 1]);
  }
  return 0;
}
