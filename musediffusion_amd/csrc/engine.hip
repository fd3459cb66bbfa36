// Library plumbing (errors, version, hipGraph wrappers) and the whole-denoiser forward:
// TransformerNetModel.forward (models/network.py:131-158) as one stream-ordered sequence of the
// kernels in this library.  No allocation, no synchronisation: safe to capture into a hipGraph.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void mh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------- per-launch timing
// The recorder is the library's only process-wide mutable state (measurement, off by default; include/musehip.h says so): one list of
// launch records for the whole process, every access under g_prof_mu, so that launches from several host threads interleave safely while it
// is on.  While it is off (g_mh_prof_on == 0, the product's state) no launch touches it.
#include <mutex>
#include <string>
#include <vector>
int g_mh_prof_on = 0;
namespace {
std::mutex g_prof_mu;
struct ProfRec { std::string name, note; unsigned grid, block; hipEvent_t e0, e1; hipStream_t s; };
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_prof_pool;
std::string g_prof_note;
hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace
void mh_prof_note(const char* fmt, ...) {
  if (!g_mh_prof_on) return;
  char buf[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  std::lock_guard<std::mutex> lock(g_prof_mu);
  g_prof_note = buf;
}
void mh_prof_begin(const char* kernel, unsigned grid_x, unsigned block_x, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  ProfRec r{kernel, g_prof_note, grid_x, block_x, prof_event(), prof_event(), stream};
  g_prof_note.clear();
  (void)hipEventRecord(r.e0, stream);
  g_prof.push_back(r);
}
void mh_prof_end(hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  if (!g_prof.empty()) (void)hipEventRecord(g_prof.back().e1, stream);
}
extern "C" int mh_profile_start(void) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  for (auto& r : g_prof) { g_prof_pool.push_back(r.e0); g_prof_pool.push_back(r.e1); }
  g_prof.clear();
  g_mh_prof_on = 1;
  return MH_OK;
}
// Stops recording, waits for the device, and writes one line per launch: "kernel\tnote\tgrid\tblock\tstream\tms\n".
// Returns the number of bytes the full report needs (call again with a larger buffer if it exceeds `cap`), or a negative status.
extern "C" int64_t mh_profile_stop(char* out, size_t cap) {
  g_mh_prof_on = 0;
  if (hipDeviceSynchronize() != hipSuccess) { mh_set_error("profile_stop: device synchronize failed"); return MH_ERR_HIP; }
  std::lock_guard<std::mutex> lock(g_prof_mu);
  std::string rep;
  char line[768];
  for (auto& r : g_prof) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) ms = -1.f;
    snprintf(line, sizeof(line), "%s\t%s\t%u\t%u\t%p\t%.6f\n", r.name.c_str(), r.note.c_str(), r.grid, r.block, (void*)r.s, ms);
    rep += line;
  }
  if (out && cap > 0) {
    const size_t n = rep.size() < cap - 1 ? rep.size() : cap - 1;
    memcpy(out, rep.data(), n);
    out[n] = 0;
  }
  return (int64_t)rep.size() + 1;
}

namespace { MH_KNOB(int, g_fuse_ln, 1); MH_KNOB(int, g_defer_ln, 1); MH_KNOB(int, g_skip, 0); MH_KNOB(int, g_prescale_q, 0); }
// 1: the bf16 panel forward stores the queries multiplied by softmax scale x log2(e) (QKV epilogue) and runs the streaming attention
// on them without a per-score multiply-subtract (mh_attention_stream_fwd_prescaled: a quarter fewer vector instructions per tile).
// Measured (round 3): 3.782 vs 3.785 ms per step at config 2, 7.329 vs 7.339 at c2-bertbase, 29.3 vs 29.5 us per half-batch launch
// alone, 18.6 vs 19.4 us for ONE (batch, head) alone on the chip - the kernel is not bound by its vector-instruction count.
// Default 0 (the queries keep the reference's scaling and one rounding less)
#ifdef MH_ABLATE
extern "C" int mh_denoiser_set_prescale_q(int on) {
  g_prescale_q = on != 0;
  return MH_OK;
}
#endif
// timing-only A/B (tools/ab_step.py skip): leave launches of one kind out of the bf16 panel forward to read their marginal cost inside
// the captured step (outputs are then garbage).  bit 0 QKV, 1 attention, 2 attention-output dense + LN, 3 FFN1, 4 FFN2 + LN,
// 5 up-projection chain (pack, two GEMMs, embedding LayerNorm), 6 down-projection
#ifdef MH_ABLATE
extern "C" int mh_denoiser_set_skip(int mask) {
  g_skip = mask;
  return MH_OK;
}
#endif
extern "C" int mh_denoiser_get_defer_ln(void) { return g_defer_ln; }
// 0 = never, 1 (default) = where no full-row LayerNorm epilogue exists for the width (d_model 768), 2 = always (A/B)
#ifdef MH_ABLATE
extern "C" int mh_denoiser_set_defer_ln(int mode) {
  g_defer_ln = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
  return MH_OK;
}
#endif
extern "C" int mh_denoiser_get_fuse_ln(void) { return g_fuse_ln; }
#ifdef MH_ABLATE
extern "C" int mh_denoiser_set_fuse_ln(int on) {
  g_fuse_ln = on != 0;
  return MH_OK;
}
#endif

extern "C" const char* mh_last_error(void) { return g_err; }
extern "C" int mh_abi_version(void) { return 1; }

extern "C" int mh_device_name(int ordinal, char* buf, int buflen) {
  MH_CHECK_ARG(buf && buflen > 0, "device_name: bad buffer");
  hipDeviceProp_t p;
  MH_HIP(hipGetDeviceProperties(&p, ordinal));
  snprintf(buf, buflen, "%s|%s|cus=%d", p.name, p.gcnArchName, p.multiProcessorCount);
  return MH_OK;
}

// ---------------------------------------------------------------- hipGraph wrappers
extern "C" int mh_graph_begin_capture(mh_stream_t stream) {
  MH_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return MH_OK;
}
extern "C" int mh_graph_end_capture(mh_stream_t stream, void** graph_exec_out) {
  MH_CHECK_ARG(graph_exec_out, "graph_end_capture: null out pointer");
  hipGraph_t graph = nullptr;
  MH_HIP(hipStreamEndCapture((hipStream_t)stream, &graph));
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);   // the executable graph keeps what it needs
  if (e != hipSuccess) {
    mh_set_error("hipGraphInstantiate: %s", hipGetErrorString(e));
    return MH_ERR_HIP;
  }
  *graph_exec_out = (void*)exec;
  return MH_OK;
}
extern "C" int mh_graph_launch(void* graph_exec, mh_stream_t stream) {
  MH_CHECK_ARG(graph_exec, "graph_launch: null graph");
  MH_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return MH_OK;
}
extern "C" int mh_graph_destroy(void* graph_exec) {
  if (graph_exec) MH_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return MH_OK;
}

// ---------------------------------------------------------------- denoiser forward
namespace {

inline size_t esize(int dtype) { return dtype == MH_BF16 ? 2 : 4; }   // (split precision: two 16-bit parts per element)
inline bool is_split(int dtype) { return dtype == MH_BF16X3 || dtype == MH_F16X3; }
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Workspace {
  char *xin, *buf0, *buf1, *bufX, *bufX1, *q, *k, *vt, *ffn;
  float *stats1, *stats2;   // deferred LayerNorm: partial (sum, sumsq) per row and 128-column tile of the two raw row buffers
  size_t total;
};

Workspace carve(const mh_denoiser* m, int B, int L, char* base) {
  const size_t N = (size_t)B * L, es = esize(m->dtype);
  Workspace w{};
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
  w.xin = take(N * (size_t)m->E_pad * es);
  w.buf1 = take(N * m->H * es);
  w.bufX = take(N * m->H * es);
  // One buffer for the residual stream: a sub-layer's output rows overwrite the rows they were computed from (every dense + residual
  // [+ LayerNorm] launch reads a tile's / a row block's residual before it stores that same tile / block, and no later launch reads
  // the old rows) - 17 MB less per batch slice in the caches at config 2 (MH_WS_INPLACE=0: two buffers, A/B)
#ifndef MH_WS_INPLACE
#define MH_WS_INPLACE 1
#endif
  w.bufX1 = MH_WS_INPLACE ? w.bufX : take(N * m->H * es);
  const size_t qkv0 = off;
  w.q = take(N * m->H * es);
  w.k = take(N * m->H * es);
  w.vt = take(N * m->H * es + 256);  // slack: the last V^T row may be over-read by one 16-B chunk
  w.buf0 = take(N * m->H * es);      // (attention output; up / down projection intermediates)
  // The FFN intermediate lives where q | k | V^T | attention output lived: all four are dead when FFN1 writes it and it is dead when the
  // next layer's projections write them (one stream orders the launches).  A batch slice then touches 121 MB instead of 188 MB at
  // config 2 - two slices' buffers fit the 256 MB Infinity Cache (MH_WS_ALIAS=0: separate buffers, A/B)
#ifndef MH_WS_ALIAS
#define MH_WS_ALIAS 1
#endif
  if (MH_WS_ALIAS && N * (size_t)m->F * es <= off - qkv0) w.ffn = base ? base + qkv0 : nullptr;   // (every path runs its launches in this order; F <= 4 H)
  else w.ffn = take(N * (size_t)m->F * es);
  const size_t slots = (size_t)(m->H + 127) / 128;
  w.stats1 = (float*)take(N * slots * 2 * sizeof(float));
  w.stats2 = (float*)take(N * slots * 2 * sizeof(float));
  w.total = off;
  return w;
}

int check_model(const mh_denoiser* m) {
  MH_CHECK_ARG(m, "denoiser: null descriptor");
  MH_CHECK_ARG(m->dtype == MH_F32 || m->dtype == MH_BF16 || is_split(m->dtype), "denoiser: unknown dtype %d", m->dtype);
  if (is_split(m->dtype)) {
    const int dh = m->nh > 0 ? m->H / m->nh : 0;
    MH_CHECK_ARG(!m->panel && (dh == 16 || dh == 32 || dh == 64) && m->H <= 2048, "denoiser(split precision): panel must be 0, head dim in {16, 32, 64}, hidden size <= 2048");
  }
  MH_CHECK_ARG(m->H > 0 && m->H % 64 == 0 && m->H <= 2048, "denoiser: hidden size %d must be a multiple of 64, <= 2048", m->H);
  MH_CHECK_ARG(m->F > 0 && m->F % 64 == 0, "denoiser: ffn size %d must be a multiple of 64", m->F);
  MH_CHECK_ARG(m->nh > 0 && m->H % m->nh == 0, "denoiser: heads %d must divide hidden %d", m->nh, m->H);
  MH_CHECK_ARG(m->E_pad % 64 == 0 && m->E_pad >= m->E && m->Tt_pad % 64 == 0 && m->Tt_pad >= m->Tt &&
                   m->T4_pad % 64 == 0 && m->T4_pad >= 4 * m->Tt,
               "denoiser: bad padded sizes");
  MH_CHECK_ARG(m->has_proj == (m->E != m->H), "denoiser: has_proj must equal (E != H)");
  MH_CHECK_ARG(!m->panel || (m->dtype == MH_BF16 && m->E % 4 == 0 && (m->H / m->nh) % 32 == 0), "denoiser: panel layout needs bf16, E %% 4 == 0 and head dim %% 32 == 0");
  MH_CHECK_ARG(m->layers || m->nL == 0, "denoiser: null layer table");
  return MH_OK;
}

}  // namespace

extern "C" size_t mh_denoiser_workspace_bytes(const mh_denoiser* m, int B, int L) {
  if (!m || B <= 0 || L <= 0) return 0;
  const size_t fwd = carve(m, B, L, nullptr).total;
  const size_t te = align256((size_t)B * m->Tt_pad * 4) + align256((size_t)B * m->T4_pad * 4);
  return fwd > te ? fwd : te;
}

extern "C" int mh_time_embed(const mh_denoiser* m, const float* t, float* emb_t_out, int B, void* workspace,
                             size_t workspace_bytes, mh_stream_t stream) {
  int rc = check_model(m);
  if (rc) return rc;
  MH_CHECK_ARG(t && emb_t_out && B > 0 && workspace, "time_embed: bad arguments");
  const int tdt = is_split(m->dtype) ? MH_F32 : m->dtype;   // (split precision: the time MLP runs once per table build, in fp32)
  const size_t es = esize(tdt);
  const size_t need = align256((size_t)B * m->Tt_pad * es) + align256((size_t)B * m->T4_pad * es);
  MH_CHECK_ARG(workspace_bytes >= need, "time_embed: workspace too small (%zu < %zu)", workspace_bytes, need);
  char* sin_buf = (char*)workspace;
  char* hid = sin_buf + align256((size_t)B * m->Tt_pad * es);
  if ((rc = mh_timestep_embedding(t, sin_buf, B, m->Tt, m->Tt_pad, 10000.0f, tdt, stream))) return rc;
  if (m->T4_pad != 4 * m->Tt) MH_HIP(hipMemsetAsync(hid, 0, (size_t)B * m->T4_pad * es, (hipStream_t)stream));
  if ((rc = mh_gemm_bias_act(sin_buf, m->Tt_pad, m->w_t0, m->Tt_pad, m->b_t0, nullptr, 0, hid, m->T4_pad, 0, B,
                             4 * m->Tt, m->Tt_pad, MH_ACT_SILU, tdt, stream)))
    return rc;
  return mh_gemm_bias_act(hid, m->T4_pad, m->w_t2, m->T4_pad, m->b_t2, nullptr, 0, emb_t_out, m->H, 1, B, m->H,
                          m->T4_pad, MH_ACT_NONE, tdt, stream);
}

namespace {
// The forward in three phases over K32-panel activation buffers the CALLER may own (bf16 panel path only): phase 1 "head" (latent ->
// up-projection -> + position / time -> LayerNorm) writes its rows to xh (ld = ldh rows per panel), phase 2 "layers" reads layer 0's
// input from xi (ldi) and writes the last layer's output to xo (ldo), phase 4 "tail" (down-projection) reads xt (ldt).  A null pointer
// means the workspace's own buffer.  With row windows of ONE full-batch buffer (pointer + first_row * 32 elements, ld = the full
// batch's rows) the sampler runs head and tail once for the whole batch and only the encoder layers per batch slice: the head / tail
// launches are latency-bound at a half batch (16 K-steps, a quarter to a half of the chip's block slots), so two half-size launches
// cost twice one full-size launch.
struct Phases { int mask; void* xh; int64_t ldh; const void* xi; int64_t ldi; void* xo; int64_t ldo; const void* xt; int64_t ldt;
                float* sq;      // sq: optional [B L] |output row|^2, written by the fused down-projection (mh_denoiser_gives_sqnorm)
                const void* tsplit; int V; int32_t* idx; const mh_step_update* upd; };   // rounding (+ update) inside the down-projection kernel (mh_denoiser_forward_round)
int denoiser_run(const mh_denoiser* m, const float* x, const float* emb_t, const int32_t* emb_row, float* out, int B, int L,
                 void* workspace, size_t workspace_bytes, mh_stream_t stream, const Phases& ph);
}  // namespace

extern "C" int mh_denoiser_forward(const mh_denoiser* m, const float* x, const float* emb_t, const int32_t* emb_row,
                                   float* out, int B, int L, void* workspace, size_t workspace_bytes,
                                   mh_stream_t stream) {
  MH_CHECK_ARG(x && emb_t && out, "denoiser_forward: null pointer");
  return denoiser_run(m, x, emb_t, emb_row, out, B, L, workspace, workspace_bytes, stream, Phases{7, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0});
}

// The forward that also leaves |out row|^2 per token (what the rounding scores add to |W_v|^2, models/rounding.py:23) - only on the
// path whose last kernel is the fused down-projection; ask mh_denoiser_gives_sqnorm first
extern "C" int mh_denoiser_gives_sqnorm(const mh_denoiser* m) {
  return m && m->panel && m->has_proj && mh_down_proj_fused_supported(m->E, m->H);
}
extern "C" int mh_denoiser_forward_sqnorm(const mh_denoiser* m, const float* x, const float* emb_t, const int32_t* emb_row, float* out,
                                          float* out_sqnorm, int B, int L, void* workspace, size_t workspace_bytes, mh_stream_t stream) {
  MH_CHECK_ARG(x && emb_t && out && out_sqnorm, "denoiser_forward_sqnorm: null pointer");
  MH_CHECK_ARG(mh_denoiser_gives_sqnorm(m), "denoiser_forward_sqnorm: this model's forward does not end in the fused down-projection");
  return denoiser_run(m, x, emb_t, emb_row, out, B, L, workspace, workspace_bytes, stream, Phases{7, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, out_sqnorm});
}

// The forward whose last kernel also rounds its rows to the nearest embedding (models/rounding.py:21-28 on the split-bf16 matrix pipe,
// csrc/headtail.hip): idx_out [B L] receives the nearest table row per token.  table_split: mh_round_split_table's buffer.
extern "C" int mh_denoiser_rounds_in_forward(const mh_denoiser* m, int V) {
  return m && m->panel && m->has_proj && mh_down_proj_round_supported(m->E, m->H, V);
}
extern "C" int mh_denoiser_forward_round(const mh_denoiser* m, const float* x, const float* emb_t, const int32_t* emb_row, float* out,
                                         const void* table_split, int V, int32_t* idx_out, const mh_step_update* upd, int B, int L,
                                         void* workspace, size_t workspace_bytes, mh_stream_t stream) {
  MH_CHECK_ARG(x && emb_t && out && table_split && idx_out, "denoiser_forward_round: null pointer");
  MH_CHECK_ARG(mh_denoiser_rounds_in_forward(m, V), "denoiser_forward_round: this model / vocabulary is not served by the fused rounding");
  return denoiser_run(m, x, emb_t, emb_row, out, B, L, workspace, workspace_bytes, stream,
                      Phases{7, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, table_split, V, idx_out, upd});
}

// Phased entry points (bf16 K32-panel models with up / down projections only; see Phases).  X buffers: bf16 [H / 32][ld rows][32].
extern "C" int mh_denoiser_phases_supported(const mh_denoiser* m) { return m && m->panel && m->has_proj && m->nL > 0; }
extern "C" int mh_denoiser_head(const mh_denoiser* m, const float* x, const float* emb_t, const int32_t* emb_row, void* x_out, int64_t ld_out,
                                int B, int L, void* workspace, size_t workspace_bytes, mh_stream_t stream) {
  MH_CHECK_ARG(mh_denoiser_phases_supported(m) && x && emb_t && x_out && ld_out >= (int64_t)B * L, "denoiser_head: bad arguments");
  return denoiser_run(m, x, emb_t, emb_row, nullptr, B, L, workspace, workspace_bytes, stream, Phases{1, x_out, ld_out, nullptr, 0, nullptr, 0, nullptr, 0});
}
extern "C" int mh_denoiser_layers(const mh_denoiser* m, const void* x_in, int64_t ld_in, void* x_out, int64_t ld_out, int B, int L,
                                  void* workspace, size_t workspace_bytes, mh_stream_t stream) {
  MH_CHECK_ARG(mh_denoiser_phases_supported(m) && x_in && x_out && ld_in >= (int64_t)B * L && ld_out >= (int64_t)B * L, "denoiser_layers: bad arguments");
  return denoiser_run(m, nullptr, nullptr, nullptr, nullptr, B, L, workspace, workspace_bytes, stream, Phases{2, nullptr, 0, x_in, ld_in, x_out, ld_out, nullptr, 0});
}
extern "C" int mh_denoiser_tail(const mh_denoiser* m, const void* x_in, int64_t ld_in, float* out, int B, int L, void* workspace,
                                size_t workspace_bytes, mh_stream_t stream) {
  MH_CHECK_ARG(mh_denoiser_phases_supported(m) && x_in && out && ld_in >= (int64_t)B * L, "denoiser_tail: bad arguments");
  return denoiser_run(m, nullptr, nullptr, nullptr, out, B, L, workspace, workspace_bytes, stream, Phases{4, nullptr, 0, nullptr, 0, nullptr, 0, x_in, ld_in});
}

namespace {
int denoiser_run(const mh_denoiser* m, const float* x, const float* emb_t, const int32_t* emb_row, float* out, int B, int L,
                 void* workspace, size_t workspace_bytes, mh_stream_t stream, const Phases& ph) {
  int rc = check_model(m);
  if (rc) return rc;
  MH_CHECK_ARG(workspace, "denoiser_forward: null workspace");
  MH_CHECK_ARG(B > 0 && L > 0 && L <= m->L_max, "denoiser_forward: seq_len %d exceeds position table %d", L, m->L_max);
  MH_CHECK_ARG(L % 8 == 0, "denoiser_forward: seq_len %d must be a multiple of 8", L);
  const Workspace w = carve(m, B, L, (char*)workspace);
  MH_CHECK_ARG(workspace_bytes >= w.total, "denoiser_forward: workspace too small (%zu < %zu)", workspace_bytes, w.total);
  const int64_t N = (int64_t)B * L;
  const int H = m->H, F = m->F, dt = m->dtype, dh = m->H / m->nh;
  const float scale = 1.0f / sqrtf((float)dh);

  if (m->panel) {
    // ---- bf16 throughput path: every activation and weight in the K32-panel layout (ld = rows per panel)
    const int P = 1;
    // where the phases hand their rows over (caller-owned row windows or the workspace's own bufX)
    void* const XH = ph.xh ? ph.xh : (void*)w.bufX;            const int64_t ldH = ph.xh ? ph.ldh : N;
    const void* const XI = ph.xi ? ph.xi : (const void*)w.bufX; const int64_t ldI = ph.xi ? ph.ldi : N;
    void* const XO = ph.xo ? ph.xo : (void*)w.bufX;            const int64_t ldO = ph.xo ? ph.ldo : N;
    const void* const XT = ph.xt ? ph.xt : (const void*)w.bufX; const int64_t ldT = ph.xt ? ph.ldt : N;
    auto gemm = [&](const void* A, int64_t lda, const void* W, int64_t w_rows, const float* bias, const void* res, int64_t ldr, void* o,
                    int of32, int64_t ldo, int Nout, int K, int act) {
      return mh_gemm_bias_act_ex(A, lda, P, W, w_rows, P, bias, res, ldr, P, o, ldo, of32 ? 0 : P, of32, N, Nout, K, act, dt, stream);
    };
    auto gemm_ln = [&](const void* A, const void* W, const float* bias, const void* res, int64_t ldr, const float* gamma, const float* beta,
                       void* o, int64_t ldo, int K) {
      return mh_gemm_bias_res_ln(A, N, P, W, H, P, bias, res, ldr, P, gamma, beta, m->ln_eps, o, ldo, P, N, H, K, stream);
    };
    const bool fuse_ln = g_fuse_ln && mh_gemm_bias_res_ln_supported(H);
    const bool stream_attn = mh_attention_stream_enabled() && mh_attention_stream_supported(L, dh) && H % 64 == 0;
    const bool pre_q = stream_attn && g_prescale_q && mh_attention_stream_prescaled_supported(L, dh);
    const float q_scale = scale * 1.4426950408889634f;
    auto attention = [&]() {
      return pre_q ? mh_attention_stream_fwd_prescaled(w.q, w.k, w.vt, w.buf0, N, P, B, L, m->nh, dh, stream)
                   : mh_attention_stream_fwd(w.q, w.k, w.vt, w.buf0, N, P, B, L, m->nh, dh, scale, stream);
    };
    if (!(ph.mask & 1) || (m->has_proj && (g_skip & 32))) {
    } else if (m->has_proj && mh_up_proj_ln_fused_supported(m->E, m->E_pad, H)) {
      // one kernel: up-projection (both dense layers), + position / time, embedding LayerNorm (csrc/headtail.hip)
      if ((rc = mh_up_proj_ln_fused(x, m->E, m->E_pad, m->w_up0, m->b_up0, m->w_up2, m->b_up2, m->pos, emb_t, emb_row, m->ln0_g, m->ln0_b,
                                    m->ln_eps, XH, ldH, B, L, H, stream)))
        return rc;
    } else if (m->has_proj) {
      if ((rc = mh_pack_panel(x, m->E, w.xin, N, N, m->E, m->E_pad, stream))) return rc;
      if ((rc = gemm(w.xin, N, m->w_up0, H, m->b_up0, nullptr, 0, w.buf0, 0, N, H, m->E_pad, MH_ACT_TANH))) return rc;
      if ((rc = gemm(w.buf0, N, m->w_up2, H, m->b_up2, nullptr, 0, w.buf1, 0, N, H, H, MH_ACT_NONE))) return rc;
      if ((rc = mh_add_pos_time_layernorm_panel(w.buf1, N, 0, m->pos, emb_t, emb_row, m->ln0_g, m->ln0_b, XH, ldH, B, L, H,
                                                m->ln_eps, stream)))
        return rc;
    } else {
      if ((rc = mh_add_pos_time_layernorm_panel(x, H, 1, m->pos, emb_t, emb_row, m->ln0_g, m->ln0_b, XH, ldH, B, L, H,
                                                m->ln_eps, stream)))
        return rc;
    }
    const int nLrun = (ph.mask & 2) ? m->nL : 0;     // (the layer loops below run only in phase 2)
    // Deferred LayerNorm (DeferArgs, gemm.hip): the attention-output and FFN-output GEMMs store RAW rows + partial row
    // statistics and their consumers normalise on the fly, so every GEMM runs on the 256x128 tile (no full-row tile, no
    // LayerNorm kernel - d_model 768 included).  The last layer's output is normalised by the panel LayerNorm kernel.
    // Measured (same box, tools: bench.py --defer-ln 0/2): +2% steps/s at d_model 768 (two LayerNorm kernels per layer
    // saved), -7% at d_model 512, where the full-row tile's LayerNorm epilogue is cheaper than the consumers' statistics staging
    // (a global round trip + two more barriers per tile, and no DMA prefetch across the epilogue): default = only where there is
    // no fused epilogue.
    bool defer = (g_defer_ln == 2 || (g_defer_ln == 1 && !fuse_ln)) && stream_attn && m->nL > 0;
    for (int l = 0; l < m->nL && defer; ++l) defer = m->layers[l].w_ff1_f && (l == 0 || m->layers[l].w_qkv_f);
    if (defer) {
      const int S = (H + 127) / 128;
      bool prev_raw = false;   // bufX holds raw rows of the previous layer's output (statistics in stats2)
      for (int l = 0; l < nLrun; ++l) {
        const mh_layer_weights& lw = m->layers[l];
        const mh_layer_weights* pl = l ? &m->layers[l - 1] : nullptr;
        const void* Xin = l == 0 ? XI : (const void*)w.bufX;      // this layer's input rows (layer 0: the head's)
        const int64_t ldin = l == 0 ? ldI : N;
        mh_ln_defer d{};
        d.h_norm = H; d.eps = m->ln_eps;
        if (g_skip & 1) {
          rc = MH_OK;
        } else if (prev_raw) {
          d.a_stats = w.stats2; d.a_slots = S; d.c1 = lw.c1_qkv;
          if (pre_q) rc = mh_gemm_qkv_vtperm_qs(Xin, ldin, lw.w_qkv_f, 3 * H, lw.c2_qkv, w.q, w.k, w.vt, B, L, H, m->nh, q_scale, &d, stream);
          else rc = mh_gemm_qkv_vtperm_defer(Xin, ldin, lw.w_qkv_f, 3 * H, lw.c2_qkv, w.q, w.k, w.vt, B, L, H, m->nh, &d, stream);
        } else {
          if (pre_q) rc = mh_gemm_qkv_vtperm_qs(Xin, ldin, lw.w_qkv, 3 * H, lw.b_qkv, w.q, w.k, w.vt, B, L, H, m->nh, q_scale, nullptr, stream);
          else rc = mh_gemm_qkv_vtperm(Xin, ldin, P, lw.w_qkv, 3 * H, P, lw.b_qkv, w.q, w.k, w.vt, B, L, H, m->nh, stream);
        }
        if (rc) return rc;
        if (!(g_skip & 2) && (rc = attention())) return rc;
        // y1 = ctx W_ao^T + b_ao + X  (raw) -> bufX1, statistics -> stats1
        d = mh_ln_defer{};
        d.h_norm = H; d.eps = m->ln_eps;
        if (prev_raw) { d.r_stats = w.stats2; d.r_slots = S; d.r_gamma = pl->ln2_g; d.r_beta = pl->ln2_b; }
        d.o_stats = w.stats1; d.o_slots = S;
        if (!(g_skip & 4) && (rc = mh_gemm_bias_act_defer(w.buf0, N, lw.w_ao, H, lw.b_ao, Xin, ldin, w.bufX1, N, N, H, H, MH_ACT_NONE, &d, stream))) return rc;
        // f = gelu(LN1(y1) W1^T + b1)
        d = mh_ln_defer{};
        d.h_norm = H; d.eps = m->ln_eps;
        d.a_stats = w.stats1; d.a_slots = S; d.c1 = lw.c1_ff1;
        if (!(g_skip & 8) && (rc = mh_gemm_bias_act_defer(w.bufX1, N, lw.w_ff1_f, F, lw.c2_ff1, nullptr, 0, w.ffn, N, N, F, H, MH_ACT_GELU_ERF, &d, stream))) return rc;
        // y2 = f W2^T + b2 + LN1(y1)
        const bool last = l == m->nL - 1;
        d = mh_ln_defer{};
        d.h_norm = H; d.eps = m->ln_eps;
        d.r_stats = w.stats1; d.r_slots = S; d.r_gamma = lw.ln1_g; d.r_beta = lw.ln1_b;
        if (!last) { d.o_stats = w.stats2; d.o_slots = S; }
        if (!(g_skip & 16) && (rc = mh_gemm_bias_act_defer(w.ffn, N, lw.w_ff2, H, lw.b_ff2, w.bufX1, N, last ? w.buf1 : w.bufX, N, N, H, F, MH_ACT_NONE, &d, stream))) return rc;
        if (last) {
          if ((rc = mh_layernorm_panel(w.buf1, N, lw.ln2_g, lw.ln2_b, XO, ldO, N, H, m->ln_eps, stream))) return rc;
        }
        prev_raw = !last;
      }
    }
    for (int l = 0; l < nLrun && !defer; ++l) {
      const mh_layer_weights& lw = m->layers[l];
      const void* Xin = l == 0 ? XI : (const void*)w.bufX;      // this layer's input rows (layer 0: the head's) ...
      const int64_t ldin = l == 0 ? ldI : N;
      void* Xout = l == m->nL - 1 ? XO : (void*)w.bufX;         // ... and where its output rows go (last layer: the tail's input)
      const int64_t ldout = l == m->nL - 1 ? ldO : N;
      if (stream_attn) {   // V^T written in the streaming kernel's key order: its stages are straight LDS-DMA copies
        if (!(g_skip & 1)) {
          if (pre_q) rc = mh_gemm_qkv_vtperm_qs(Xin, ldin, lw.w_qkv, 3 * H, lw.b_qkv, w.q, w.k, w.vt, B, L, H, m->nh, q_scale, nullptr, stream);
          else rc = mh_gemm_qkv_vtperm(Xin, ldin, P, lw.w_qkv, 3 * H, P, lw.b_qkv, w.q, w.k, w.vt, B, L, H, m->nh, stream);
          if (rc) return rc;
        }
        if (!(g_skip & 2) && (rc = attention())) return rc;
      } else {
        if ((rc = mh_gemm_qkv_ex(Xin, ldin, P, lw.w_qkv, 3 * H, P, lw.b_qkv, w.q, w.k, w.vt, B, L, H, m->nh, dt, stream))) return rc;
        if ((rc = mh_attention_fwd_ex(w.q, w.k, w.vt, w.buf0, N, P, B, L, m->nh, dh, scale, dt, stream))) return rc;
      }
      if (g_skip & 4) {
      } else if (fuse_ln) {   // dense + residual + LayerNorm in one kernel: the block owns complete rows
        if ((rc = gemm_ln(w.buf0, lw.w_ao, lw.b_ao, Xin, ldin, lw.ln1_g, lw.ln1_b, w.bufX1, N, H))) return rc;
      } else {
        if ((rc = gemm(w.buf0, N, lw.w_ao, H, lw.b_ao, Xin, ldin, w.buf1, 0, N, H, H, MH_ACT_NONE))) return rc;
        if ((rc = mh_layernorm_panel(w.buf1, N, lw.ln1_g, lw.ln1_b, w.bufX1, N, N, H, m->ln_eps, stream))) return rc;
      }
      if (!(g_skip & 8) && (rc = gemm(w.bufX1, N, lw.w_ff1, F, lw.b_ff1, nullptr, 0, w.ffn, 0, N, F, H, MH_ACT_GELU_ERF))) return rc;
      if (g_skip & 16) {
      } else if (fuse_ln) {
        if ((rc = gemm_ln(w.ffn, lw.w_ff2, lw.b_ff2, w.bufX1, N, lw.ln2_g, lw.ln2_b, Xout, ldout, F))) return rc;
      } else {
        if ((rc = gemm(w.ffn, N, lw.w_ff2, H, lw.b_ff2, w.bufX1, N, w.buf1, 0, N, H, F, MH_ACT_NONE))) return rc;
        if ((rc = mh_layernorm_panel(w.buf1, N, lw.ln2_g, lw.ln2_b, Xout, ldout, N, H, m->ln_eps, stream))) return rc;
      }
    }
    if (!(ph.mask & 4)) return MH_OK;
    if (m->has_proj) {
      if (g_skip & 64) return MH_OK;
      if (ph.idx)   // ... and the nearest-embedding rounding of its rows
        return mh_down_proj_round_fused(XT, ldT, m->w_dn0, m->b_dn0, m->w_dn2, m->b_dn2, out, ph.sq, ph.tsplit, ph.V, ph.idx, ph.upd, N, m->E, H, stream);
      if (mh_down_proj_fused_supported(m->E, H))   // one kernel for both dense layers of the down-projection (csrc/headtail.hip)
        return mh_down_proj_fused(XT, ldT, m->w_dn0, m->b_dn0, m->w_dn2, m->b_dn2, out, ph.sq, N, m->E, H, stream);
      if ((rc = gemm(XT, ldT, m->w_dn0, H, m->b_dn0, nullptr, 0, w.buf0, 0, N, H, H, MH_ACT_TANH))) return rc;
      return gemm(w.buf0, N, m->w_dn2, m->E, m->b_dn2, nullptr, 0, out, 1, m->E, m->E, H, MH_ACT_NONE);
    }
    return mh_unpack_panel_f32(XT, ldT, out, m->E, N, m->E, stream);
  }
  MH_CHECK_ARG(ph.mask == 7, "denoiser: the phased entry points serve the bf16 panel path only");
  if (is_split(dt)) {
    // ---- split precision (csrc/split.hip): activations as split panels (hi + lo 16-bit parts), three matrix-pipe products per
    // reference product, LayerNorm on fp32 rows.  Buffers (4 bytes per element each): xin / buf0 / bufX / bufX1 / ffn split panels,
    // buf1 fp32 rows, q..k = the packed [N, 2H] split row-major q | k projection, vt = the [H, N] split row-major V^T projection.
    const int64_t part_qk = N * 2 * H, part_vt = (int64_t)H * N;
    auto sgemm = [&](const void* A, const void* W, int64_t w_rows, const float* bias, const void* res, void* o, int64_t ldo, int mode, int64_t M2, int Nout,
                     int K, int act) {
      return mh_split_gemm(A, N, W, w_rows, bias, 0, res, N, o, ldo, mode, 0, M2, Nout, K, act, dt, stream);
    };
    if (m->has_proj) {   // network.py:141-149
      if ((rc = mh_split_pack(x, m->E, w.xin, N, N, m->E, m->E_pad, dt, stream))) return rc;
      if ((rc = sgemm(w.xin, m->w_up0, H, m->b_up0, nullptr, w.buf0, N, 0, N, H, m->E_pad, MH_ACT_TANH))) return rc;
      if ((rc = sgemm(w.buf0, m->w_up2, H, m->b_up2, nullptr, w.buf1, H, 2, N, H, H, MH_ACT_NONE))) return rc;
      if ((rc = mh_split_layernorm((const float*)w.buf1, H, m->pos, emb_t, emb_row, m->ln0_g, m->ln0_b, w.bufX, N, N, L, H, m->ln_eps, dt, stream))) return rc;
    } else {
      if ((rc = mh_split_layernorm(x, H, m->pos, emb_t, emb_row, m->ln0_g, m->ln0_b, w.bufX, N, N, L, H, m->ln_eps, dt, stream))) return rc;
    }
    for (int l = 0; l < m->nL; ++l) {   // HF BertLayer, post-LN (network.py:151)
      const mh_layer_weights& lw = m->layers[l];
      // q | k = X [W_q; W_k]^T + b  ->  [N, 2H];   V^T = W_v X^T + b_v (rows)  ->  [H, N]: the attention's operand images, no transposes
      if ((rc = mh_split_gemm(w.bufX, N, lw.w_qkv, 3 * H, lw.b_qkv, 0, nullptr, 0, w.q, 2 * H, 1, part_qk, N, 2 * H, H, MH_ACT_NONE, dt, stream))) return rc;
      if ((rc = mh_split_gemm((const char*)lw.w_qkv + (size_t)2 * H * 32 * 2, 3 * H, w.bufX, N, lw.b_qkv + 2 * H, 1, nullptr, 0, w.vt, N, 1, part_vt, H, (int)N, H,
                              MH_ACT_NONE, dt, stream)))
        return rc;
      if ((rc = mh_split_attention(w.q, 2 * H, H, part_qk, w.vt, N, part_vt, w.buf0, N, B, L, m->nh, dh, scale, dt, stream))) return rc;
      const bool fuse = mh_split_gemm_res_ln_supported(H);   // dense + residual + LayerNorm in one kernel (complete rows per block)
      if (fuse) {
        if ((rc = mh_split_gemm_res_ln(w.buf0, N, lw.w_ao, H, lw.b_ao, w.bufX, N, lw.ln1_g, lw.ln1_b, m->ln_eps, w.bufX1, N, N, H, H, dt, stream))) return rc;
      } else {
        if ((rc = sgemm(w.buf0, lw.w_ao, H, lw.b_ao, w.bufX, w.buf1, H, 2, N, H, H, MH_ACT_NONE))) return rc;
        if ((rc = mh_split_layernorm((const float*)w.buf1, H, nullptr, nullptr, nullptr, lw.ln1_g, lw.ln1_b, w.bufX1, N, N, L, H, m->ln_eps, dt, stream))) return rc;
      }
      if ((rc = sgemm(w.bufX1, lw.w_ff1, F, lw.b_ff1, nullptr, w.ffn, N, 0, N, F, H, MH_ACT_GELU_ERF))) return rc;
      if (fuse) {   // (the output rows overwrite the layer's input rows: X is not read by this launch)
        if ((rc = mh_split_gemm_res_ln(w.ffn, N, lw.w_ff2, H, lw.b_ff2, w.bufX1, N, lw.ln2_g, lw.ln2_b, m->ln_eps, w.bufX, N, N, H, F, dt, stream))) return rc;
      } else {
        if ((rc = sgemm(w.ffn, lw.w_ff2, H, lw.b_ff2, w.bufX1, w.buf1, H, 2, N, H, F, MH_ACT_NONE))) return rc;
        if ((rc = mh_split_layernorm((const float*)w.buf1, H, nullptr, nullptr, nullptr, lw.ln2_g, lw.ln2_b, w.bufX, N, N, L, H, m->ln_eps, dt, stream))) return rc;
      }
    }
    if (m->has_proj) {   // network.py:153-157
      if ((rc = sgemm(w.bufX, m->w_dn0, H, m->b_dn0, nullptr, w.buf0, N, 0, N, H, H, MH_ACT_TANH))) return rc;
      return sgemm(w.buf0, m->w_dn2, m->E, m->b_dn2, nullptr, out, m->E, 2, N, m->E, H, MH_ACT_NONE);
    }
    return mh_split_join(w.bufX, N, out, m->E, N, m->E, H, dt, stream);
  }
  // ---- embeddings: (up-projection) + position + time, LayerNorm          network.py:141-149
  if (m->has_proj) {
    if ((rc = mh_cast_pad(x, m->E, w.xin, m->E_pad, N, m->E, N, dt, stream))) return rc;
    if ((rc = mh_gemm_bias_act(w.xin, m->E_pad, m->w_up0, m->E_pad, m->b_up0, nullptr, 0, w.buf0, H, 0, N, H, m->E_pad,
                               MH_ACT_TANH, dt, stream)))
      return rc;
    if ((rc = mh_gemm_bias_act(w.buf0, H, m->w_up2, H, m->b_up2, nullptr, 0, w.buf1, H, 0, N, H, H, MH_ACT_NONE, dt,
                               stream)))
      return rc;
    if ((rc = mh_add_pos_time_layernorm(w.buf1, H, 0, m->pos, emb_t, emb_row, m->ln0_g, m->ln0_b, w.bufX, B, L, H,
                                        m->ln_eps, dt, stream)))
      return rc;
  } else {
    if ((rc = mh_add_pos_time_layernorm(x, H, 1, m->pos, emb_t, emb_row, m->ln0_g, m->ln0_b, w.bufX, B, L, H, m->ln_eps,
                                        dt, stream)))
      return rc;
  }
  // ---- encoder layers (HF BertLayer, post-LN)                            network.py:151
  for (int l = 0; l < m->nL; ++l) {
    const mh_layer_weights& lw = m->layers[l];
    if ((rc = mh_gemm_qkv(w.bufX, H, lw.w_qkv, H, lw.b_qkv, w.q, w.k, w.vt, B, L, H, m->nh, dt, stream))) return rc;
    if ((rc = mh_attention_fwd(w.q, w.k, w.vt, w.buf0, H, B, L, m->nh, dh, scale, dt, stream))) return rc;
    if ((rc = mh_gemm_bias_act(w.buf0, H, lw.w_ao, H, lw.b_ao, w.bufX, H, w.buf1, H, 0, N, H, H, MH_ACT_NONE, dt, stream)))
      return rc;
    if ((rc = mh_layernorm(w.buf1, lw.ln1_g, lw.ln1_b, w.bufX1, N, H, m->ln_eps, dt, stream))) return rc;
    if ((rc = mh_gemm_bias_act(w.bufX1, H, lw.w_ff1, H, lw.b_ff1, nullptr, 0, w.ffn, F, 0, N, F, H, MH_ACT_GELU_ERF, dt,
                               stream)))
      return rc;
    if ((rc = mh_gemm_bias_act(w.ffn, F, lw.w_ff2, F, lw.b_ff2, w.bufX1, H, w.buf1, H, 0, N, H, F, MH_ACT_NONE, dt, stream)))
      return rc;
    if ((rc = mh_layernorm(w.buf1, lw.ln2_g, lw.ln2_b, w.bufX, N, H, m->ln_eps, dt, stream))) return rc;
  }
  // ---- output down-projection                                            network.py:153-157
  if (m->has_proj) {
    if ((rc = mh_gemm_bias_act(w.bufX, H, m->w_dn0, H, m->b_dn0, nullptr, 0, w.buf0, H, 0, N, H, H, MH_ACT_TANH, dt, stream)))
      return rc;
    return mh_gemm_bias_act(w.buf0, H, m->w_dn2, H, m->b_dn2, nullptr, 0, out, m->E, 1, N, m->E, H, MH_ACT_NONE, dt, stream);
  }
  return mh_cast_to_f32(w.bufX, H, out, m->E, N, m->E, dt, stream);
}
}  // namespace
