// ABI plumbing (version, thread-local errors, device check), hipGraph capture helpers and the fused DiT evaluation
// that chains the per-op kernels for one network call (reference backbones/dit.py:452-470, model/modules.py:627-641).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <cstdlib>

#include "f5e_common.h"
#include "gemm_bf16_args.h"

static thread_local char g_err[512] = "";

void f5e_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

#define HIP_TRY(call, what)                                                        \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess) {                                                        \
      f5e_set_error("%s: %s", what, hipGetErrorString(e_));                        \
      return F5E_ERR_HIP;                                                          \
    }                                                                              \
  } while (0)

#define F5E_TRY(call)        \
  do {                       \
    int r_ = (call);         \
    if (r_ != F5E_OK) return r_; \
  } while (0)

extern "C" {

int f5e_abi_version(void) { return F5E_ABI_VERSION; }
const char* f5e_last_error(void) { return g_err; }

int f5e_check_device(void) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev), "hipGetDevice");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    f5e_set_error("libf5e_hip is built for gfx950 only; device %d is %s", dev, prop.gcnArchName);
    return F5E_ERR_UNSUPPORTED;
  }
  return F5E_OK;
}

int f5e_graph_begin(hipStream_t st) {
  HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed), "hipStreamBeginCapture");
  return F5E_OK;
}

int f5e_graph_end(hipStream_t st, void** graph_exec_out) {
  F5E_REQUIRE(graph_exec_out, "graph_end: null output");
  hipGraph_t g = nullptr;
  HIP_TRY(hipStreamEndCapture(st, &g), "hipStreamEndCapture");
  hipGraphExec_t ex = nullptr;
  hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);  // the executable graph keeps what it needs; a failure here would only leak the template
  if (e != hipSuccess) {
    f5e_set_error("hipGraphInstantiate: %s", hipGetErrorString(e));
    return F5E_ERR_HIP;
  }
  *graph_exec_out = (void*)ex;
  return F5E_OK;
}

int f5e_graph_launch(void* graph_exec, hipStream_t st) {
  F5E_REQUIRE(graph_exec, "graph_launch: null graph");
  HIP_TRY(hipGraphLaunch((hipGraphExec_t)graph_exec, st), "hipGraphLaunch");
  return F5E_OK;
}

int f5e_graph_destroy(void* graph_exec) {
  if (graph_exec) HIP_TRY(hipGraphExecDestroy((hipGraphExec_t)graph_exec), "hipGraphExecDestroy");
  return F5E_OK;
}

int f5e_workspace_bytes(const f5e_dit_plan* p, f5e_dit_workspace* out) {
  F5E_REQUIRE(p && out, "workspace_bytes: null argument");
  F5E_REQUIRE(p->S > 0 && p->N > 0 && p->D > 0 && p->H > 0 && p->FF > 0 && p->mel > 0, "workspace_bytes: bad shape");
  const unsigned long long M = (unsigned long long)p->S * p->N, D = p->D, inner = (unsigned long long)p->H * 64;
  const unsigned long long n_pad = ((unsigned long long)p->N + 63) / 64 * 64;
  const unsigned long long qk = (unsigned long long)p->S * p->H * n_pad * 64 * 2;
  memset(out, 0, sizeof(*out));
  out->n_pad = (int)n_pad;
  unsigned long long* b = out->bytes;
  b[F5E_WS_H0] = M * D * 4;       b[F5E_WS_H0_BF16] = M * D * 2;  b[F5E_WS_C1] = M * D * 2;
  b[F5E_WS_X] = M * D * 4;        b[F5E_WS_HN] = M * D * 2;
  b[F5E_WS_Q] = qk;               b[F5E_WS_K] = qk;               b[F5E_WS_VT] = qk;
  b[F5E_WS_AO] = M * inner * 2;   b[F5E_WS_FF] = M * (unsigned long long)p->FF * 2;
  b[F5E_WS_PRED] = M * (unsigned long long)p->mel * 4;
  b[F5E_WS_LN_STATS] = p->fuse_ln ? M * (D / 64) * 2 * 4 : 0;
  b[F5E_WS_LN_ROWMEAN] = p->fuse_ln ? M * 4 : 0;
  b[F5E_WS_SKIP_RES] = p->w_skip ? M * D * 4 : 0;
  b[F5E_WS_SKIP_TMP] = p->w_skip ? M * D * 4 : 0;
  unsigned long long off = 0;
  for (int i = 0; i < F5E_WS_COUNT; ++i) {
    out->offset[i] = off;
    off += (b[i] + 255) / 256 * 256;
  }
  out->total = off;
  return F5E_OK;
}

struct F5eTimer {
  int capacity, count;
  hipEvent_t* start;
  hipEvent_t* stop;
  int* op;  // op class of each recorded pair
};

int f5e_timer_create(int capacity, void** timer_out) {
  F5E_REQUIRE(capacity > 0 && timer_out, "timer_create: bad arguments");
  F5eTimer* t = new F5eTimer{capacity, 0, new hipEvent_t[capacity], new hipEvent_t[capacity], new int[capacity]};
  for (int i = 0; i < capacity; ++i) {
    HIP_TRY(hipEventCreate(&t->start[i]), "hipEventCreate");
    HIP_TRY(hipEventCreate(&t->stop[i]), "hipEventCreate");
  }
  *timer_out = t;
  return F5E_OK;
}

int f5e_timer_destroy(void* timer) {
  if (!timer) return F5E_OK;
  F5eTimer* t = (F5eTimer*)timer;
  for (int i = 0; i < t->capacity; ++i) {
    (void)hipEventDestroy(t->start[i]);
    (void)hipEventDestroy(t->stop[i]);
  }
  delete[] t->start;
  delete[] t->stop;
  delete[] t->op;
  delete t;
  return F5E_OK;
}

int f5e_timer_reset(void* timer) {
  F5E_REQUIRE(timer, "timer_reset: null");
  ((F5eTimer*)timer)->count = 0;
  return F5E_OK;
}

int f5e_timer_read(void* timer, float* ms_out_host, int max_out, int* count_out_host) {
  F5E_REQUIRE(timer && ms_out_host && count_out_host, "timer_read: null");
  F5eTimer* t = (F5eTimer*)timer;
  const int n = t->count < max_out ? t->count : max_out;
  for (int i = 0; i < n; ++i) {
    HIP_TRY(hipEventSynchronize(t->stop[i]), "hipEventSynchronize");
    HIP_TRY(hipEventElapsedTime(&ms_out_host[i], t->start[i], t->stop[i]), "hipEventElapsedTime");
  }
  *count_out_host = n;
  return F5E_OK;
}

int f5e_timer_read_ops(void* timer, int* ops_out_host, int max_out, int* count_out_host) {
  F5E_REQUIRE(timer && ops_out_host && count_out_host, "timer_read_ops: null");
  F5eTimer* t = (F5eTimer*)timer;
  const int n = t->count < max_out ? t->count : max_out;
  for (int i = 0; i < n; ++i) ops_out_host[i] = t->op[i];
  *count_out_host = n;
  return F5E_OK;
}

// records a start/stop event pair around `call` when the plan's timer selects this op class (eager launches only)
#define F5E_TIMED(opclass, call)                                                  \
  do {                                                                            \
    const bool sel_ = p->timer_op >= 0 ? p->timer_op == (opclass) : (((-p->timer_op) >> (opclass)) & 1) != 0; \
    F5eTimer* tm_ = (p->timer && sel_) ? (F5eTimer*)p->timer : nullptr;          \
    const bool on_ = tm_ && tm_->count < tm_->capacity;                           \
    if (on_) HIP_TRY(hipEventRecord(tm_->start[tm_->count], st), "hipEventRecord"); \
    F5E_TRY(call);                                                                \
    if (on_) {                                                                    \
      HIP_TRY(hipEventRecord(tm_->stop[tm_->count], st), "hipEventRecord");       \
      tm_->op[tm_->count] = (opclass);                                            \
      tm_->count++;                                                               \
    }                                                                             \
  } while (0)

int f5e_dit_forward(hipStream_t st, const f5e_dit_plan* p) {
  F5E_REQUIRE(p, "dit_forward: null plan");
  F5E_REQUIRE(p->S > 0 && p->B > 0 && p->S % p->B == 0 && p->N > 0 && p->L > 0, "dit_forward: bad S/B/N/L");
  F5E_REQUIRE(p->D % 256 == 0 && p->H > 0 && p->FF % 64 == 0 && p->mel % 4 == 0, "dit_forward: unsupported dims");
  F5E_REQUIRE(p->y && p->w_x && p->in_const && p->rope_cs && p->mod && p->blocks && p->w_proj, "dit_forward: null input");
  F5E_REQUIRE(p->h0 && p->h0_bf16 && p->c1 && p->x && p->hn && p->q && p->k && p->vt && p->ao && p->ff && p->pred,
              "dit_forward: null workspace");
  const int M = p->S * p->N, D = p->D, inner = p->H * 64;
  const int row_stride = p->L * 6 * D + 2 * D;
  const int eval_stride = p->mod_rows * row_stride;

  // K4: input projection, x-part per step + hoisted cond/text/ppg part (backbones/dit.py:173-175)
  F5E_TIMED(F5E_OP_INPROJ, f5e_gemm_f32(st, p->y, p->mel, p->B * p->N, F5E_ACT_NONE, p->w_x, p->ldw_x, nullptr, F5E_ACT_NONE, nullptr,
                       p->in_const, D, M, nullptr, p->h0, D, p->h0_bf16, D, M, D, p->mel));
  // K5: conv position embedding + residual (dit.py:176)
  F5E_TIMED(F5E_OP_CONVPOS, f5e_convpos(st, p->h0_bf16, D, p->convpos_w1, p->convpos_b1, 0, p->c1, D, nullptr, 0, nullptr, 0, p->S, p->N, D, p->convpos_groups));
  F5E_TIMED(F5E_OP_CONVPOS, f5e_convpos(st, p->c1, D, p->convpos_w2, p->convpos_b2, 1, nullptr, 0, p->x, D, p->h0, D, p->S, p->N, D, p->convpos_groups));

  if (p->w_skip) {  // long skip connection keeps the embedded input (backbones/dit.py:456-457)
    F5E_REQUIRE(p->skip_res && p->skip_tmp, "dit_forward: long skip needs skip_res / skip_tmp");
    HIP_TRY(hipMemcpyAsync(p->skip_res, p->x, (size_t)M * D * sizeof(float), hipMemcpyDeviceToDevice, st),
            "hipMemcpyAsync(skip_res)");
  }
  if (p->fuse_ln) {
    // Fused AdaLN chain (f5e_ln_fuse): hn holds xs = bf16((x - row_mean) (1 + scale)) for the next linear, ln_stats the tile
    // statistics of x relative to row_mean; every gate+residual GEMM refreshes both for the norm that follows it.
    F5E_REQUIRE(!p->w_skip, "dit_forward: fused AdaLN does not cover the long skip connection");
    F5E_REQUIRE(p->ln_stats && p->ln_rowmean && p->cd && D % 256 == 0 && D <= 1024 && p->mod_rows == 1 && inner == D && p->cd_stride % 4 == 0,
                "dit_forward: fused AdaLN needs ln_stats, ln_rowmean, cd tables, D %% 256 == 0, D <= 1024, one modulation row and heads * 64 == D");
    const int parts = D / 64, ls = 6 * inner + 2 * p->FF, cd_eval_stride = p->mod_rows * p->cd_stride;
    F5E_REQUIRE(p->cd_stride >= p->L * ls + 2 * p->mel, "dit_forward: cd_stride too small");
    f5e_ln_fuse cons{}, prod{};
    cons.stats = p->ln_stats; cons.parts = parts; cons.cd_stride = p->cd_stride; cons.cd_rows = p->mod_rows;
    cons.cd_eval_stride = cd_eval_stride; cons.eval_ptr = p->eval_ptr; cons.rows_per_seq = p->N; cons.eps = 1e-6f;
    prod.xs_out = p->hn; prod.ld_xs = D; prod.stats_out = p->ln_stats;
    cons.row_mean = prod.row_mean = p->ln_rowmean;
    // head of the chain: the embedded input's exact row means become the first centring offsets.  (Round 2 folded this
    // into the second position-embedding conv; a conv workgroup sees 64 of a row's channels, not its mean.)
    F5E_TIMED(F5E_OP_LN, f5e_adaln_pre(st, p->x, D, p->hn, D, p->mod + D, row_stride, p->mod_rows, p->N, p->eval_ptr, eval_stride,
                                       p->ln_stats, parts, p->ln_rowmean, M, D));
    // Infinity-Cache prefetch (f5e_common.h): every GEMM / attention launch of a block drags the weights of the launch after
    // next into the memory-side cache with a few grid-tail workgroups, so the batch-1 GEMMs stop waiting for HBM.
    const bool pfon = p->mall_prefetch != 0;
    // Who hosts what: QKV -> w_out, attention (four-wave workgroups without an LDS ring: the extra ones start beside them) ->
    // w_ff1, OUT -> w_ff2, FF2 -> the next block's w_qkv.  Measured at C2, ms per pass, round 2: off 48.35; FF1 hosting the
    // next w_qkv instead of FF2 47.45; this scheme 46.41; attention hosting both FF weights 46.95.  Round 4: the role-split
    // GEMMs pack their share onto the CUs their one-round grid leaves idle (gemm_bf16.hip launch()), after which five other
    // assignments of the four weights to the five launches measure within 0.9 % of this one (the diagnostics build
    // keeps them behind F5E_PF_SCHEME).
    const unsigned b_out = (unsigned)((size_t)D * inner * 2), b_ff = (unsigned)((size_t)p->FF * D * 2);
    const unsigned b_qkv = (unsigned)((size_t)3 * inner * D * 2);
    for (int l = 0; l < p->L; ++l) {
      const f5e_dit_block_weights& w = p->blocks[l];
      F5E_REQUIRE(!w.q_norm_w, "dit_forward: fused AdaLN and qk_norm are exclusive");
      const float* mb = p->mod + (size_t)l * 6 * D;
      const float* cdl = p->cd + (size_t)l * ls;
      const F5ePrefetch pf_next{{l + 1 < p->L ? p->blocks[l + 1].w_qkv : p->w_proj, nullptr},
                                {l + 1 < p->L ? b_qkv : (unsigned)((size_t)p->mel * D * 2), 0}};
      F5ePrefetch pf_qkv{{w.w_out, nullptr}, {b_out, 0}};
      F5ePrefetch pf_attn{{w.w_ff1, nullptr}, {b_ff, 0}};
      F5ePrefetch pf_out{{w.w_ff2, nullptr}, {b_ff, 0}};
      F5ePrefetch pf_nx = pf_next, pf_ff1{};
      {
        // One assignment at every M.  (For a while in round 4 launches past one round of 64-row tiles switched to scheme 3 --
        // attention hosting both FF weights, FF2 the next w_qkv and w_out: C4 -0.8 % -- until the attention kernel stopped
        // splitting K/V at those sizes: unsplit it pays 2.4 us for hosting 8.4 MB, and scheme 0 is ahead again)
        int scheme = 0;
#ifdef F5E_TOOLS
        static const int scheme_env = getenv("F5E_PF_SCHEME") ? atoi(getenv("F5E_PF_SCHEME")) : -1;   // diagnostics build: A/B
        if (scheme_env >= 0) scheme = scheme_env;
#endif
        const void* wout_next = l + 1 < p->L ? (const void*)p->blocks[l + 1].w_out : nullptr;
        if (scheme == 1 || scheme == 3) {   // attention hosts both FF weights, the out-projection nothing
          pf_attn = F5ePrefetch{{w.w_ff1, w.w_ff2}, {b_ff, b_ff}};
          pf_out = F5ePrefetch{};
        }
        if (scheme == 2 || scheme == 3) {   // FF2 hosts the next block's w_qkv AND w_out, QKV nothing (block 0: its own w_out)
          pf_nx = F5ePrefetch{{pf_next.ptr[0], wout_next}, {pf_next.bytes[0], wout_next ? b_out : 0}};
          if (l > 0) pf_qkv = F5ePrefetch{};
        }
#ifdef F5E_TOOLS
        if (scheme == 5 || scheme == 6) {   // the GEMMs host everything, attention nothing
          pf_qkv = F5ePrefetch{{w.w_out, w.w_ff1}, {b_out, b_ff}};
          pf_attn = F5ePrefetch{};
          if (scheme == 6) { pf_out = F5ePrefetch{}; pf_ff1 = F5ePrefetch{{w.w_ff2, nullptr}, {b_ff, 0}}; }
        }
        if (scheme == 7) {                  // just in time: every GEMM hosts the weights of the next GEMM
          pf_attn = F5ePrefetch{};
          pf_out = F5ePrefetch{{w.w_ff1, nullptr}, {b_ff, 0}};
          pf_ff1 = F5ePrefetch{{w.w_ff2, nullptr}, {b_ff, 0}};
        }
        if (scheme == 4) {                  // attention hosts w_ff1 and the next block's w_qkv, FF2 nothing
          pf_attn = F5ePrefetch{{w.w_ff1, pf_next.ptr[0]}, {b_ff, pf_next.bytes[0]}};
          pf_nx = F5ePrefetch{};
        }
#endif
      }
      cons.c = cdl; cons.d = cdl + 3 * inner;
      F5E_TIMED(F5E_OP_QKV, f5e_gemm_bf16_qkv_rope_pf(st, p->hn, D, w.w_qkv, D, nullptr, p->q, p->k, p->vt, p->n_pad, p->H,
                                        p->rope_heads, p->rope_cs, nullptr, nullptr, p->N, M, D, 0, &cons, pfon ? &pf_qkv : nullptr));
      F5E_TIMED(F5E_OP_ATTN, f5e_flash_attn_pf(st, p->q, p->k, p->vt, p->ao, inner, p->seq_len, p->S, p->H, p->N, p->n_pad, 0,
                                               pfon ? &pf_attn : nullptr));
      prod.next_scale = mb + 4 * D;  // scale_mlp
      F5E_TIMED(F5E_OP_OUT, f5e_gemm_bf16_gate_residual_pf(st, p->ao, inner, w.w_out, inner, w.b_out, p->x, D, mb + 2 * D,
                                          row_stride, p->mod_rows, p->eval_ptr, eval_stride, p->N, p->seq_len, M, D,
                                          inner, 0, &prod, pfon ? &pf_out : nullptr));
      cons.c = cdl + 6 * inner; cons.d = cdl + 6 * inner + p->FF;
      F5E_TIMED(F5E_OP_FF1, f5e_gemm_bf16_bias_pf(st, p->hn, D, w.w_ff1, D, nullptr, p->ff, p->FF, M, p->FF, D,
                                    F5E_ACT_GELU_TANH, 0, 0, &cons, pfon && pf_ff1.ptr[0] ? &pf_ff1 : nullptr));
      // next norm: attn_norm of block l+1 (scale_msa at +D) or the final AdaLN (scale first: modules.py:333)
      prod.next_scale = (l + 1 < p->L) ? mb + 6 * D + D : p->mod + (size_t)p->L * 6 * D;
      F5E_TIMED(F5E_OP_FF2, f5e_gemm_bf16_gate_residual_pf(st, p->ff, p->FF, w.w_ff2, p->FF, w.b_ff2, p->x, D, mb + 5 * D,
                                          row_stride, p->mod_rows, p->eval_ptr, eval_stride, p->N, nullptr, M, D, p->FF,
                                          0, &prod, pfon ? &pf_nx : nullptr));
    }
    cons.c = p->cd + (size_t)p->L * ls; cons.d = cons.c + p->mel;
    F5E_TIMED(F5E_OP_FINAL, f5e_gemm_bf16_bias_ln(st, p->hn, D, p->w_proj, D, nullptr, p->pred, p->mel, M, p->mel, D,
                                    F5E_ACT_NONE, 1, 0, &cons));
    return F5E_OK;
  }
  for (int l = 0; l < p->L; ++l) {
    const f5e_dit_block_weights& w = p->blocks[l];
    const float* mb = p->mod + (size_t)l * 6 * D;  // shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
    F5E_TIMED(F5E_OP_LN, f5e_layernorm(st, p->x, D, p->hn, D, 1, nullptr, nullptr, mb + D, mb, row_stride, p->mod_rows, p->N,
                          p->eval_ptr, eval_stride, M, D, 1e-6f));
    F5E_TIMED(F5E_OP_QKV, f5e_gemm_bf16_qkv_rope(st, p->hn, D, w.w_qkv, D, w.b_qkv, p->q, p->k, p->vt, p->n_pad, p->H, p->rope_heads,
                                   p->rope_cs, w.q_norm_w, w.k_norm_w, p->N, M, D, 0));
    F5E_TIMED(F5E_OP_ATTN, f5e_flash_attn(st, p->q, p->k, p->vt, p->ao, inner, p->seq_len, p->S, p->H, p->N, p->n_pad, 0));
    F5E_TIMED(F5E_OP_OUT, f5e_gemm_bf16_gate_residual(st, p->ao, inner, w.w_out, inner, w.b_out, p->x, D, mb + 2 * D, row_stride,
                                        p->mod_rows, p->eval_ptr, eval_stride, p->N, p->seq_len, M, D, inner, 0));
    F5E_TIMED(F5E_OP_LN, f5e_layernorm(st, p->x, D, p->hn, D, 1, nullptr, nullptr, mb + 4 * D, mb + 3 * D, row_stride, p->mod_rows,
                          p->N, p->eval_ptr, eval_stride, M, D, 1e-6f));
    F5E_TIMED(F5E_OP_FF1, f5e_gemm_bf16_bias(st, p->hn, D, w.w_ff1, D, w.b_ff1, p->ff, p->FF, M, p->FF, D, F5E_ACT_GELU_TANH, 0, 0));
    F5E_TIMED(F5E_OP_FF2, f5e_gemm_bf16_gate_residual(st, p->ff, p->FF, w.w_ff2, p->FF, w.b_ff2, p->x, D, mb + 5 * D, row_stride,
                                        p->mod_rows, p->eval_ptr, eval_stride, p->N, nullptr, M, D, p->FF, 0));
  }
  if (p->w_skip) {  // x = Linear_{2D->D, no bias}(cat(x, residual)) as two fp32 GEMMs (dit.py:466-467)
    F5E_TRY(f5e_gemm_f32(st, p->x, D, M, F5E_ACT_NONE, p->w_skip, 2 * D, nullptr, F5E_ACT_NONE, nullptr, nullptr, 0, 0,
                         nullptr, p->skip_tmp, D, nullptr, 0, M, D, D));
    F5E_TRY(f5e_gemm_f32(st, p->skip_res, D, M, F5E_ACT_NONE, p->w_skip + D, 2 * D, nullptr, F5E_ACT_NONE, nullptr,
                         p->skip_tmp, D, M, nullptr, p->x, D, nullptr, 0, M, D, D));
  }
  // K13: final AdaLN (scale, shift order: modules.py:333) + proj_out
  const float* mf = p->mod + (size_t)p->L * 6 * D;
  F5E_TIMED(F5E_OP_LN, f5e_layernorm(st, p->x, D, p->hn, D, 1, nullptr, nullptr, mf, mf + D, row_stride, p->mod_rows, p->N,
                        p->eval_ptr, eval_stride, M, D, 1e-6f));
  F5E_TIMED(F5E_OP_FINAL, f5e_gemm_bf16_bias(st, p->hn, D, p->w_proj, D, p->b_proj, p->pred, p->mel, M, p->mel, D, F5E_ACT_NONE, 1, 0));
  return F5E_OK;
}

int f5e_sample_loop(hipStream_t st, const f5e_loop_plan* lp) {
  F5E_REQUIRE(lp && lp->eval_a, "sample_loop: null plan");
  F5E_REQUIRE(lp->steps > 0 && lp->n > 0 && lp->y && lp->pred && lp->coef && lp->eval_ptr && lp->done_ctr,
              "sample_loop: steps, n, y, pred, coef, eval_ptr and done_ctr are required");
  F5E_REQUIRE(lp->eval_a->eval_ptr == lp->eval_ptr && lp->eval_a->y == lp->y,
              "sample_loop: eval_a must read the loop's state y and evaluation counter");
  const bool midpoint = lp->eval_b != nullptr;
  if (midpoint)
    F5E_REQUIRE(lp->y_mid && lp->eval_b->y == lp->y_mid && lp->eval_b->eval_ptr == lp->eval_ptr,
                "sample_loop: eval_b must read y_mid and the loop's evaluation counter");
  const long long ts = lp->traj ? lp->n : 0;   // trajectory row picked on the device from the evaluation counter
  for (int i = 0; i < lp->steps; ++i) {
    F5E_TRY(f5e_dit_forward(st, lp->eval_a));
    if (!midpoint) {
      F5E_TRY(f5e_ode_update_traj(st, lp->pred, lp->n, lp->mode, lp->w0, lp->w1, lp->y, lp->y, lp->traj, ts, 1, lp->coef,
                                  lp->eval_ptr, lp->done_ctr, lp->n));
    } else {
      F5E_TRY(f5e_ode_update_traj(st, lp->pred, lp->n, lp->mode, lp->w0, lp->w1, lp->y, lp->y_mid, nullptr, 0, 1, lp->coef,
                                  lp->eval_ptr, lp->done_ctr, lp->n));
      F5E_TRY(f5e_dit_forward(st, lp->eval_b));
      F5E_TRY(f5e_ode_update_traj(st, lp->pred, lp->n, lp->mode, lp->w0, lp->w1, lp->y, lp->y, lp->traj, ts, 2, lp->coef,
                                  lp->eval_ptr, lp->done_ctr, lp->n));
    }
  }
  return F5E_OK;
}

}  // extern "C"
