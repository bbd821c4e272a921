// Small HBM-bound elementwise kernels of the sampler (K2, K3 gather, K6, K14, K15 of SURVEY 2.3).
#include "f5e_common.h"

namespace {

inline int grid_for(size_t total) {
  size_t g = (total + 255) / 256;
  return (int)(g < 2048 ? (g ? g : 1) : 2048);
}

// SinusPositionEmbedding (modules.py:149-161): out[e] = cat(sin(a), cos(a)), a = (scale * t[e]) * freqs[k],
// freqs[k] = exp(-k * ln(1e4)/(half-1)) is a host-built constant table.
__global__ void sinus_embed_kernel(const float* t, const float* freqs, float* out, int E, int dim, float scale) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E * half) return;
  const int e = i / half, k = i - e * half;
  const float arg = (scale * t[e]) * freqs[k];
  float s, c;
  sincosf(arg, &s, &c);
  out[(size_t)e * dim + k] = s;
  out[(size_t)e * dim + half + k] = c;
}

// Rotary table (x_transformers RotaryEmbedding.forward_from_seq_len): out[n][i] = (cos, sin)(n * inv_freq[i])
__global__ void rope_table_kernel(const float* inv_freq, float* out, int N, int half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * half) return;
  const int n = i / half, k = i - n * half;
  const float ang = (float)n * inv_freq[k];
  float s, c;
  sincosf(ang, &s, &c);
  out[(size_t)i * 2] = c;
  out[(size_t)i * 2 + 1] = s;
}

// TextEmbedding front half (backbones/dit.py:68-80): out[b][n] = (table[ids[b][n]] + pos[min(n, max_pos-1)]) * keep[b][n]
__global__ void text_gather_kernel(const int* ids, const float* table, const float* pos, const float* keep, float* out,
                                   int B, int N, int TD, int max_pos, int table_rows) {
  const int td4 = TD / 4;
  const size_t total = (size_t)B * N * td4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % td4) * 4;
    const size_t bn = i / td4;
    const int n = (int)(bn % N);
    // ids are validated on the host where they start there (CFM._prepare raises IndexError like nn.Embedding); the clamp
    // keeps a bad id handed over as a device tensor from reading outside the table
    const int id = min(max(ids[bn], 0), table_rows - 1);
    f32x4 v = *(const f32x4*)(table + (size_t)id * TD + c);
    if (pos) v += *(const f32x4*)(pos + (size_t)min(n, max_pos - 1) * TD + c);
    if (keep) v *= keep[bn];
    *(f32x4*)(out + bn * TD + c) = v;
  }
}

// CFG combination + fixed-grid ODE update (cfm.py:447, :187, :310; torchdiffeq euler/midpoint stage):
//   v = mode 0: p0 | mode 1: p0 + (p0 - p1) * w0 | mode 2: w0 (p2 - p1) + w1 (p1 - p0) + p0
//   dst = base + coef[eval] * v     (+ optional trajectory row)
__global__ void ode_update_kernel(const float* pred, size_t branch_stride, int mode, float w0, float w1,
                                  const float* base, float* dst, float* traj, size_t traj_stride, int traj_div,
                                  const float* coef, int* eval_ptr, unsigned* done_ctr, size_t n) {
  // counter and step size through the scalar cache (both were written by earlier launches): one s_load chain instead of
  // two dependent vector round trips in front of a 4 us kernel
  const int e = eval_ptr ? load_uniform_i32(eval_ptr) : 0;
  const float h = __builtin_bit_cast(float, load_uniform_i32((const int*)(coef + e)));
  // trajectory row chosen on the device: row (e + 1) / traj_div of [rows][traj_stride] (one captured step serves the
  // whole grid without a copy launch per step); traj_stride 0 = traj is the row itself
  if (traj && traj_stride) traj += (size_t)((e + 1) / traj_div) * traj_stride;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float p0 = pred[i];
    float v = p0;
    if (mode == 1) {
      v = p0 + (p0 - pred[i + branch_stride]) * w0;
    } else if (mode == 2) {
      const float p1 = pred[i + branch_stride], p2 = pred[i + 2 * branch_stride];
      v = w0 * (p2 - p1) + w1 * (p1 - p0) + p0;
    }
    const float y = base[i] + h * v;
    dst[i] = y;
    if (traj) traj[i] = y;
  }
  // Advance the evaluation counter once every block has read it: the last block to finish bumps it and re-arms the
  // ticket.  Visibility to the NEXT kernel is given by the kernel boundary; within this kernel only the ticket is
  // shared, through a device-scope atomic (placement independent).
  if (done_ctr) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned t = atomicAdd(done_ctr, 1u);
      if (t == gridDim.x - 1) {
        *eval_ptr = e + 1;
        *done_ctr = 0u;
      }
    }
  }
}

__global__ void advance_eval_kernel(int* eval_ptr) { *eval_ptr += 1; }

// out = where(mask[b][n], cond, y)   (cfm.py:476)
__global__ void stitch_kernel(const float* cond, const float* y, const unsigned char* mask, float* out, size_t rows,
                              int C) {
  const size_t total = rows * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256)
    out[i] = mask[i / C] ? cond[i] : y[i];
}

__global__ void cast_bf16_kernel(const float* x, bf16* y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = (bf16)x[i];
}

__global__ void cast_f32_kernel(const bf16* x, float* y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = (float)x[i];
}

// out = a x + b y (+ c): the duplicate_test start state (1 - t) y0 + t cond (cfm.py:463) and "1 + scale" rows
__global__ void axpby_kernel(const float* x, const float* y, float* out, float a, float b, float c, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    out[i] = a * x[i] + (y ? b * y[i] : 0.0f) + c;
}

// GumbelVectorQuantizer.forward in eval mode (reference model/modules.py:881-950): per (row, group) argmax of the
// logits -> one-hot -> codebook row; targets = the argmax indices.  One wave per (row, group).
__global__ __launch_bounds__(256) void vq_lookup_kernel(const float* logits, int ld, const float* vars, int combine,
                                                         float* out, int* targets, int rows, int G, int V, int vd) {
  const int lane = threadIdx.x & 63;
  const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (long long)rows * G) return;
  const int r = (int)(item / G), g = (int)(item % G);
  const float* lp = logits + (size_t)r * ld + (size_t)g * V;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int v = lane; v < V; v += 64) {
    const float x = lp[v];
    if (x > best) { best = x; bi = v; }  // strict: the smallest index wins within a lane
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }  // first maximal index, as torch.max
  }
  if (lane == 0 && targets) targets[(size_t)r * G + g] = bi;
  const float* src = vars + ((size_t)(combine ? 0 : g) * V + bi) * vd;
  for (int d = lane; d < vd; d += 64) out[(size_t)r * G * vd + (size_t)g * vd + d] = src[d];
}

// code / prob perplexity of the same forward: hard_probs = mean one-hot, avg_probs = mean softmax, each
// exp(-sum p log(p + 1e-7)) summed over groups.  One block, deterministic (no atomics).
__global__ __launch_bounds__(256) void vq_stats_kernel(const float* logits, int ld, const int* targets, float* stats,
                                                        int rows, int G, int V) {
  extern __shared__ float sm[];  // [4 waves][V] softmax sums, then [V] counts
  float* accw = sm;
  float* cnt = sm + 4 * V;
  __shared__ float red[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float code_ppl = 0.f, prob_ppl = 0.f;
  for (int g = 0; g < G; ++g) {
    for (int i = tid; i < 5 * V; i += 256) sm[i] = 0.f;
    __syncthreads();
    for (int r = wave; r < rows; r += 4) {
      const float* lp = logits + (size_t)r * ld + (size_t)g * V;
      float mx = -INFINITY;
      for (int v = lane; v < V; v += 64) mx = fmaxf(mx, lp[v]);
      mx = wave_max(mx);
      float s = 0.f;
      for (int v = lane; v < V; v += 64) s += expf(lp[v] - mx);
      s = wave_sum(s);
      const float inv = 1.0f / s;
      for (int v = lane; v < V; v += 64) accw[wave * V + v] += expf(lp[v] - mx) * inv;
    }
    __syncthreads();
    if (tid == 0)  // counts: serial over rows keeps it exact and ordered (rows is a few thousand at most)
      for (int r = 0; r < rows; ++r) cnt[targets[(size_t)r * G + g]] += 1.0f;
    __syncthreads();
    float hc = 0.f, hp = 0.f;
    for (int v = tid; v < V; v += 256) {
      const float p_hard = cnt[v] / (float)rows;
      const float p_avg = ((accw[v] + accw[V + v]) + (accw[2 * V + v] + accw[3 * V + v])) / (float)rows;
      hc += p_hard * logf(p_hard + 1e-7f);
      hp += p_avg * logf(p_avg + 1e-7f);
    }
    hc = wave_sum(hc);
    hp = wave_sum(hp);
    if (tid == 0) { red[0] = 0.f; red[1] = 0.f; }
    __syncthreads();
    if (lane == 0) { atomicAdd(&red[0], hc); atomicAdd(&red[1], hp); }  // 4 LDS adds per group
    __syncthreads();
    code_ppl += expf(-red[0]);
    prob_ppl += expf(-red[1]);
    __syncthreads();
  }
  if (tid == 0) { stats[0] = code_ppl; stats[1] = prob_ppl; }
}

}  // namespace

extern "C" {

int f5e_sinus_embed(hipStream_t st, const float* t, const float* freqs, float* out, int E, int dim, float scale) {
  F5E_REQUIRE(t && freqs && out && E > 0 && dim >= 4 && dim % 2 == 0, "sinus_embed: bad arguments");
  const int total = E * dim / 2;
  hipLaunchKernelGGL(sinus_embed_kernel, dim3((total + 255) / 256), dim3(256), 0, st, t, freqs, out, E, dim, scale);
  F5E_LAUNCH_CHECK("sinus_embed");
  return F5E_OK;
}

int f5e_rope_table(hipStream_t st, const float* inv_freq, float* out, int N, int half) {
  F5E_REQUIRE(inv_freq && out && N > 0 && half > 0, "rope_table: bad arguments");
  const int total = N * half;
  hipLaunchKernelGGL(rope_table_kernel, dim3((total + 255) / 256), dim3(256), 0, st, inv_freq, out, N, half);
  F5E_LAUNCH_CHECK("rope_table");
  return F5E_OK;
}

int f5e_text_gather(hipStream_t st, const int* ids, const float* table, const float* pos, const float* keep,
                    float* out, int B, int N, int TD, int max_pos, int table_rows) {
  F5E_REQUIRE(ids && table && out && B > 0 && N > 0 && TD > 0 && TD % 4 == 0 && table_rows > 0, "text_gather: bad arguments");
  hipLaunchKernelGGL(text_gather_kernel, dim3(grid_for((size_t)B * N * TD / 4)), dim3(256), 0, st, ids, table, pos,
                     keep, out, B, N, TD, max_pos, table_rows);
  F5E_LAUNCH_CHECK("text_gather");
  return F5E_OK;
}

int f5e_ode_update(hipStream_t st, const float* pred, long long branch_stride, int mode, float w0, float w1,
                   const float* base, float* dst, float* traj, const float* coef, int* eval_ptr, unsigned* done_ctr,
                   long long n) {
  return f5e_ode_update_traj(st, pred, branch_stride, mode, w0, w1, base, dst, traj, 0, 1, coef, eval_ptr, done_ctr, n);
}

int f5e_ode_update_traj(hipStream_t st, const float* pred, long long branch_stride, int mode, float w0, float w1,
                        const float* base, float* dst, float* traj, long long traj_stride, int traj_div,
                        const float* coef, int* eval_ptr, unsigned* done_ctr, long long n) {
  F5E_REQUIRE(pred && base && dst && coef && n > 0, "ode_update: bad arguments");
  F5E_REQUIRE(mode >= 0 && mode <= 2, "ode_update: mode must be 0 (plain), 1 (cfg) or 2 (three-branch)");
  F5E_REQUIRE(!done_ctr || eval_ptr, "ode_update: done_ctr (auto-advance) needs eval_ptr");
  F5E_REQUIRE(traj_stride >= 0 && traj_div >= 1 && (traj_stride == 0 || eval_ptr), "ode_update: bad trajectory indexing");
  hipLaunchKernelGGL(ode_update_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, st, pred, (size_t)branch_stride, mode,
                     w0, w1, base, dst, traj, (size_t)traj_stride, traj_div, coef, eval_ptr, done_ctr, (size_t)n);
  F5E_LAUNCH_CHECK("ode_update");
  return F5E_OK;
}

int f5e_advance_eval(hipStream_t st, int* eval_ptr) {
  F5E_REQUIRE(eval_ptr, "advance_eval: null");
  hipLaunchKernelGGL(advance_eval_kernel, dim3(1), dim3(1), 0, st, eval_ptr);
  F5E_LAUNCH_CHECK("advance_eval");
  return F5E_OK;
}

int f5e_stitch(hipStream_t st, const float* cond, const float* y, const unsigned char* mask, float* out,
               long long rows, int C) {
  F5E_REQUIRE(cond && y && mask && out && rows > 0 && C > 0, "stitch: bad arguments");
  hipLaunchKernelGGL(stitch_kernel, dim3(grid_for((size_t)rows * C)), dim3(256), 0, st, cond, y, mask, out,
                     (size_t)rows, C);
  F5E_LAUNCH_CHECK("stitch");
  return F5E_OK;
}

int f5e_vq_eval(hipStream_t st, const float* logits, int ld, const float* vars, int combine_groups, float* out,
                int* targets, float* stats, int rows, int groups, int num_vars, int var_dim) {
  F5E_REQUIRE(logits && vars && out && targets, "vq_eval: null operand");
  F5E_REQUIRE(rows > 0 && groups > 0 && num_vars > 0 && var_dim > 0 && ld >= groups * num_vars, "vq_eval: bad shape");
  const long long items = (long long)rows * groups;
  hipLaunchKernelGGL(vq_lookup_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, logits, ld, vars,
                     combine_groups, out, targets, rows, groups, num_vars, var_dim);
  F5E_LAUNCH_CHECK("vq_lookup");
  if (stats) {
    F5E_REQUIRE(num_vars <= 2048, "vq_eval: statistics kernel supports num_vars <= 2048");
    hipLaunchKernelGGL(vq_stats_kernel, dim3(1), dim3(256), 5 * num_vars * sizeof(float), st, logits, ld, targets, stats,
                       rows, groups, num_vars);
    F5E_LAUNCH_CHECK("vq_stats");
  }
  return F5E_OK;
}

int f5e_cast_bf16(hipStream_t st, const float* x, void* y, long long n) {
  F5E_REQUIRE(x && y && n > 0, "cast_bf16: bad arguments");
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, st, x, (bf16*)y, (size_t)n);
  F5E_LAUNCH_CHECK("cast_bf16");
  return F5E_OK;
}

int f5e_cast_f32(hipStream_t st, const void* x, float* y, long long n) {
  F5E_REQUIRE(x && y && n > 0, "cast_f32: bad arguments");
  hipLaunchKernelGGL(cast_f32_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, st, (const bf16*)x, y, (size_t)n);
  F5E_LAUNCH_CHECK("cast_f32");
  return F5E_OK;
}

int f5e_axpby(hipStream_t st, const float* x, const float* y, float* out, float a, float b, float c, long long n) {
  F5E_REQUIRE(x && out && n > 0, "axpby: bad arguments");
  hipLaunchKernelGGL(axpby_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, st, x, y, out, a, b, c, (size_t)n);
  F5E_LAUNCH_CHECK("axpby");
  return F5E_OK;
}

}  // extern "C"
