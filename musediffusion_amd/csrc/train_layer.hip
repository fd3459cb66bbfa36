// One encoder layer of training_losses (models/diffusion.py:594-699 -> models/network.py:151 -> HF BertLayer: BertSelfAttention,
// BertSelfOutput, BertIntermediate, BertOutput, and their autograd backward) as two host calls that launch the layer's kernels with every
// GEMM operand in the K32-panel layout of the sampler's tiles (round 6).  Rounds 1 - 5 ran these GEMMs on row-major activations: a K step of
// a 256-row tile is then sixteen 64-byte row segments per LDS-DMA instruction instead of one contiguous KiB, and the same kernels ran 6 - 30 %
// slower than on panels (tools/train_gemm_ab.py: the K = 2048 product with the LayerNorm epilogue 97.7 -> 67.3 us, the QKV projection
// 69.8 -> 53.7, the act-grad input gradient 99.4 -> 82.6).  What is row-major here is what only row kernels read: the q | k | v
// projection (the attention kernels stream its heads), the pre-LayerNorm rows and the gradients entering a LayerNorm backward.
//
// Forward:   qkv = x Wqkv^T + b            [N][3H] rows          gemm_big 256x128 panels
//            vt  = V^T per head (key order of the streaming kernel)                       head_permute mode 4
//            ctx = softmax(q k^T / sqrt(dh)) [o keep / (1 - p)] v  -> panels, lse          attn_stream_bf16_kernel
//            pre1 = drop(ctx Wao^T + b) + x ; x1 = LN1(pre1)       rows / panels           gemm_big 128x512pp EPI 3 (training form)
//            g = gelu(x1 W1^T + b), dact = gelu'(.)                panels                  gemm_big 256x128 + second output
//            pre2 = drop(g W2^T + b) + x1 ; y = LN2(pre2)          rows / panels           gemm_big 128x512pp EPI 3
// Backward:  LN2' -> r2 rows (residual branch), m2 panels (o keep / (1 - p): dense branch), dln2
//            dW2 | db2 = m2^T g                                    gemm_tn panels, folded in fixed order
//            d1 = (m2 W2) o dact                                   panels   gemm_big + act-grad epilogue
//            dW1 | db1 = d1^T x1
//            t = d1 W1 + r2                                        rows     (the gradient entering LN1)
//            LN1' -> r1 rows, m1 panels, dln1
//            dWao | dbao = m1^T ctx
//            dctx = m1 Wao                                         rows
//            attention backward -> dqkv panels [3H / 32][N][32]
//            dWqkv | dbqkv = dqkv^T x
//            dx = dqkv Wqkv + r1                                   rows
#include <mutex>

#include "common.h"

int mh_ln_bwd_rows(const void* x, const void* dy, const float* gamma, void* dx, void* dx_dropped, int64_t ldm, int m_panel, int always,
                   const mh_dropout* drop, float* partial, int n_partial, int64_t rows, int H, float eps, int dtype, mh_stream_t stream);
int mh_ln_bwd_fold(const float* partial, int n_partial, int H, float* dgamma, float* dbeta, int accumulate, mh_stream_t stream);

namespace {

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

struct Scratch {
  char *r, *m, *d1, *t, *dqkv, *D, *lnp[2], *part[4];   // (one partial buffer per fold: the folds run behind the kernels that follow them)
  size_t total;
};

int ln_partials(int64_t N) { const int64_t nb = (N + 3) / 4; return (int)(nb < 1024 ? nb : 1024); }

size_t tn_partial_floats(int64_t N, int M, int Ncols) {
  return (size_t)mh_gemm_dw_splits(N, M, Ncols) * ((size_t)M * Ncols + M);
}

Scratch carve(char* base, int B, int L, int H, int F, int nh, int64_t ld) {
  const int64_t N = (int64_t)B * L;
  Scratch s{};
  size_t o = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + o : nullptr; o += align256(bytes); return p; };
  s.r = take((size_t)N * H * 2);          // residual-branch gradient of the LayerNorm backward in flight (rows)
  s.m = take((size_t)ld * H * 2);         // its dense-branch twin (panels)
  s.d1 = take((size_t)ld * F * 2);        // gradient of the FFN pre-activation (panels)
  s.t = take((size_t)N * H * 2);          // rows: the gradient entering LN1, then d(ctx)
  s.dqkv = take((size_t)ld * 3 * H * 2);  // panels
  s.D = take((size_t)B * nh * L * 4);
  for (int i = 0; i < 2; ++i) s.lnp[i] = take((size_t)2 * ln_partials(N) * H * 4);
  s.part[0] = take(tn_partial_floats(N, H, F) * 4);        // dW2
  s.part[1] = take(tn_partial_floats(N, F, H) * 4);        // dW1
  s.part[2] = take(tn_partial_floats(N, H, H) * 4);        // dWao
  s.part[3] = take(tn_partial_floats(N, 3 * H, H) * 4);    // dWqkv
  s.total = o;
  return s;
}

const mh_dropout* site(const mh_dropout& d) { return (d.p > 0.f || d.mask) ? &d : nullptr; }

// Gradient finalisation off the critical path: the folds of the split-K partials (mh_sum_slices: 300 MB of fp32 partials per layer) and of
// the LayerNorm backward's column partials produce parameter gradients only - nothing in the layer's backward reads them - so they run on
// a side stream under the GEMM that follows the kernel whose partials they fold, and join the caller's stream at the end of the call.
// The events are per device and re-recorded by every call (a wait captures the record that precedes it in program order).
struct Side {
  hipStream_t main, side;     // side == nullptr: everything on `main`, in order
  hipEvent_t ev[2];           // [0] main -> side hand-over, [1] side -> main join
};
hipEvent_t* side_events() {
  static hipEvent_t ev[MH_MAX_DEVICES][2];
  static bool made[MH_MAX_DEVICES] = {};
  static std::mutex mu;      // (creation only; the events themselves belong to whoever drives this device's layers - see musehip.h)
  const int dev = mh_current_device();
  std::lock_guard<std::mutex> lock(mu);
  if (!made[dev]) {
    for (int i = 0; i < 2; ++i) if (hipEventCreateWithFlags(&ev[dev][i], hipEventDisableTiming) != hipSuccess) return nullptr;
    made[dev] = true;
  }
  return ev[dev];
}
// the stream a fold of what `main` has just produced runs on
int hand_over(const Side& sd, hipStream_t* out) {
  *out = sd.main;
  if (!sd.side) return MH_OK;
  MH_HIP(hipEventRecord(sd.ev[0], sd.main));
  MH_HIP(hipStreamWaitEvent(sd.side, sd.ev[0], 0));
  *out = sd.side;
  return MH_OK;
}
int join(const Side& sd) {
  if (!sd.side) return MH_OK;
  MH_HIP(hipEventRecord(sd.ev[1], sd.side));
  MH_HIP(hipStreamWaitEvent(sd.main, sd.ev[1], 0));
  return MH_OK;
}

// dW [M][N] | db [M] = A^T B over the N tokens (panels), folded in fixed order into `out`
int dw(const void* A, const void* Bm, int64_t N, int64_t ld, int M, int Ncols, float* part, float* out, const Side& sd) {
  const int S = mh_gemm_dw_splits(N, M, Ncols);
  const int64_t n = (int64_t)M * Ncols + M;
  int rc = mh_gemm_dw_bias_ex(A, ld, Bm, ld, 1, S == 1 ? out : part, S, N, M, Ncols, 1, sd.main);
  if (rc || S == 1) return rc;
  hipStream_t fs;
  if ((rc = hand_over(sd, &fs))) return rc;
  return mh_sum_slices(part, S, n, out, fs);
}

int ln_bwd(const void* pre, const void* dy, const float* gamma, void* r, void* m, int64_t ld, const mh_dropout* drop, float* lnp, int nb, float* dg,
           int64_t N, int H, float eps, const Side& sd) {
  int rc = mh_ln_bwd_rows(pre, dy, gamma, r, m, ld, 1, 1, drop, lnp, nb, N, H, eps, MH_BF16, sd.main);
  if (rc) return rc;
  hipStream_t fs;
  if ((rc = hand_over(sd, &fs))) return rc;
  return mh_ln_bwd_fold(lnp, nb, H, dg, dg + H, 0, fs);
}

}  // namespace

extern "C" int mh_train_layer_supported(int B, int L, int H, int F, int nh) {
  if (B <= 0 || L <= 0 || nh <= 0 || H % nh != 0) return 0;
  const int dh = H / nh;
  const int64_t N = (int64_t)B * L;
  return H == 512 && F > 0 && F % 256 == 0 && L % 64 == 0 && (dh == 32 || dh == 64) && mh_attention_stream_bwd_supported(L, dh) && N % 32 == 0 &&
         N * (int64_t)(F > 3 * H ? F : 3 * H) * 2 < (1ll << 31);
}

extern "C" size_t mh_train_layer_scratch_bytes(int B, int L, int H, int F, int nh, int64_t ld) {
  if (!mh_train_layer_supported(B, L, H, F, nh) || ld < (int64_t)B * L) return 0;
  return carve(nullptr, B, L, H, F, nh, ld).total;
}

extern "C" int64_t mh_train_layer_grad_floats(int H, int F) {
  return (int64_t)3 * H * H + 3 * H + (int64_t)H * H + H + (int64_t)F * H + F + (int64_t)H * F + H + 4 * H;
}

extern "C" int mh_train_layer_fwd(const mh_train_layer* t, mh_stream_t stream) {
  MH_CHECK_ARG(t && mh_train_layer_supported(t->B, t->L, t->H, t->F, t->nh), "train_layer_fwd: shape not served (H = 512, F %% 256 == 0, L %% 64 == 0, L >= 512, head dim 32 / 64)");
  MH_CHECK_ARG(t->wqkv && t->wao && t->w1 && t->w2 && t->bqkv && t->bao && t->b1 && t->b2 && t->ln1_g && t->ln1_b && t->ln2_g && t->ln2_b, "train_layer_fwd: null weight");
  MH_CHECK_ARG(t->x && t->qkv && t->vt && t->ctx && t->lse && t->pre1 && t->x1 && t->g && t->dact && t->pre2 && t->y, "train_layer_fwd: null activation buffer");
  const int B = t->B, L = t->L, H = t->H, F = t->F, nh = t->nh, dh = H / nh;
  const int64_t N = (int64_t)B * L, ld = t->ld;
  MH_CHECK_ARG(ld >= N && ld * (int64_t)(F > 3 * H ? F : 3 * H) * 2 < (1ll << 31), "train_layer: ld must cover the B L rows");
  const float scale = 1.0f / sqrtf((float)dh);
  bf16* qkv = reinterpret_cast<bf16*>(t->qkv);
  int rc;
  {   // q | k | v projection -> rows [N][3H]
    mh_gemm_desc d{};
    d.A = t->x; d.lda = ld; d.a_panel = 1; d.W = t->wqkv; d.ldw = 3 * H; d.w_panel = 1; d.bias = t->bqkv;
    d.out = qkv; d.ldo = 3 * H; d.M = N; d.N = 3 * H; d.K = H;
    if ((rc = mh_gemm_desc_launch(&d, stream))) return rc;
  }
  if ((rc = mh_head_permute(qkv + 2 * H, t->vt, 3 * H, B, L, nh, dh, 4, MH_BF16, stream))) return rc;
  const mh_dropout* da = site(t->drop_attn);
  MH_CHECK_ARG(!da || t->keep_bits, "train_layer_fwd: attention dropout needs keep_bits");
  if (da) rc = mh_attention_stream_fwd_drop(qkv, qkv + H, t->vt, t->ctx, ld, 1, B, L, nh, dh, scale, t->lse, (int64_t)L * 3 * H, dh, 3 * H, da, t->keep_bits,
                                            t->bits_in, stream);
  else rc = mh_attention_stream_fwd_ex(qkv, qkv + H, t->vt, t->ctx, ld, 1, B, L, nh, dh, scale, t->lse, (int64_t)L * 3 * H, dh, 3 * H, stream);
  if (rc) return rc;
  {   // BertSelfOutput: dense -> dropout -> LayerNorm(. + x)
    mh_gemm_desc d{};
    d.A = t->ctx; d.lda = ld; d.a_panel = 1; d.W = t->wao; d.ldw = H; d.w_panel = 1; d.bias = t->bao;
    d.residual = t->x; d.ldr = ld; d.r_panel = 1; d.out = t->x1; d.ldo = ld; d.o_panel = 1;
    d.pre_out = t->pre1; d.ldp = H; d.p_panel = 0; d.ln_gamma = t->ln1_g; d.ln_beta = t->ln1_b; d.ln_eps = t->ln_eps;
    d.drop = site(t->drop_ao); d.M = N; d.N = H; d.K = H;
    if ((rc = mh_gemm_desc_launch(&d, stream))) return rc;
  }
  {   // BertIntermediate: dense + GELU; gelu' kept for the backward
    mh_gemm_desc d{};
    d.A = t->x1; d.lda = ld; d.a_panel = 1; d.W = t->w1; d.ldw = F; d.w_panel = 1; d.bias = t->b1;
    d.out = t->g; d.ldo = ld; d.o_panel = 1; d.pre_out = t->dact; d.pre_kind = 1; d.act = MH_ACT_GELU_ERF; d.M = N; d.N = F; d.K = H;
    if ((rc = mh_gemm_desc_launch(&d, stream))) return rc;
  }
  {   // BertOutput: dense -> dropout -> LayerNorm(. + x1)
    mh_gemm_desc d{};
    d.A = t->g; d.lda = ld; d.a_panel = 1; d.W = t->w2; d.ldw = H; d.w_panel = 1; d.bias = t->b2;
    d.residual = t->x1; d.ldr = ld; d.r_panel = 1; d.out = t->y; d.ldo = t->y_panel ? ld : H; d.o_panel = t->y_panel;
    d.pre_out = t->pre2; d.ldp = H; d.p_panel = 0; d.ln_gamma = t->ln2_g; d.ln_beta = t->ln2_b; d.ln_eps = t->ln_eps;
    d.drop = site(t->drop_ffn); d.M = N; d.N = H; d.K = F;
    if ((rc = mh_gemm_desc_launch(&d, stream))) return rc;
  }
  return MH_OK;
}

extern "C" int mh_train_layer_bwd(const mh_train_layer* t, mh_stream_t stream) {
  MH_CHECK_ARG(t && mh_train_layer_supported(t->B, t->L, t->H, t->F, t->nh), "train_layer_bwd: shape not served");
  MH_CHECK_ARG(t->wqkv_t && t->wao_t && t->w1_t && t->w2_t && t->ln1_g && t->ln2_g, "train_layer_bwd: null weight");
  MH_CHECK_ARG(t->x && t->qkv && t->ctx && t->lse && t->pre1 && t->x1 && t->g && t->dact && t->pre2 && t->dy && t->dx && t->grads && t->scratch,
               "train_layer_bwd: null buffer");
  const int B = t->B, L = t->L, H = t->H, F = t->F, nh = t->nh, dh = H / nh;
  const int64_t N = (int64_t)B * L, ld = t->ld;
  MH_CHECK_ARG(ld >= N, "train_layer: ld must cover the B L rows");
  MH_CHECK_ARG(t->scratch_bytes >= mh_train_layer_scratch_bytes(B, L, H, F, nh, ld), "train_layer_bwd: scratch too small");
  const Scratch s = carve(reinterpret_cast<char*>(t->scratch), B, L, H, F, nh, ld);
  hipStream_t st = (hipStream_t)stream;
  const float scale = 1.0f / sqrtf((float)dh);
  float* g = t->grads;
  float* gWqkv = g;                 g += (int64_t)3 * H * H + 3 * H;
  float* gWao = g;                  g += (int64_t)H * H + H;
  float* gW1 = g;                   g += (int64_t)F * H + F;
  float* gW2 = g;                   g += (int64_t)H * F + H;
  float* gln1 = g;                  g += 2 * H;
  float* gln2 = g;
  float* part[4];
  for (int i = 0; i < 4; ++i) part[i] = reinterpret_cast<float*>(s.part[i]);
  float* lnp[2] = {reinterpret_cast<float*>(s.lnp[0]), reinterpret_cast<float*>(s.lnp[1])};
  const int nb = ln_partials(N);
  Side sd{st, (hipStream_t)t->side_stream, {nullptr, nullptr}};
  if (sd.side == sd.main) sd.side = nullptr;
  if (sd.side) {
    hipEvent_t* ev = side_events();
    if (!ev) { mh_set_error("train_layer_bwd: hipEventCreate failed"); return MH_ERR_HIP; }
    sd.ev[0] = ev[0]; sd.ev[1] = ev[1];
  }
  const bf16* qkv = reinterpret_cast<const bf16*>(t->qkv);
  int rc;
  // LN2 backward: r = d(pre2) rows (also the residual branch's gradient), m = r o keep / (1 - p) panels
  if ((rc = ln_bwd(t->pre2, t->dy, t->ln2_g, s.r, s.m, ld, site(t->drop_ffn), lnp[0], nb, gln2, N, H, t->ln_eps, sd))) return rc;
  if ((rc = dw(s.m, t->g, N, ld, H, F, part[0], gW2, sd))) return rc;
  {   // d1 = (m W2) o gelu'
    mh_gemm_desc d{};
    d.A = s.m; d.lda = ld; d.a_panel = 1; d.W = t->w2_t; d.ldw = F; d.w_panel = 1;
    d.residual = t->dact; d.ldr = ld; d.r_panel = 1; d.act_grad = MH_ACT_DERIV; d.out = s.d1; d.ldo = ld; d.o_panel = 1; d.M = N; d.N = F; d.K = H;
    if ((rc = mh_gemm_desc_launch(&d, stream))) return rc;
  }
  if ((rc = dw(s.d1, t->x1, N, ld, F, H, part[1], gW1, sd))) return rc;
  {   // gradient entering LN1 = d1 W1 + r (rows)
    mh_gemm_desc d{};
    d.A = s.d1; d.lda = ld; d.a_panel = 1; d.W = t->w1_t; d.ldw = H; d.w_panel = 1;
    d.residual = s.r; d.ldr = H; d.out = s.t; d.ldo = H; d.M = N; d.N = H; d.K = F;
    if ((rc = mh_gemm_desc_launch(&d, stream))) return rc;
  }
  if ((rc = ln_bwd(t->pre1, s.t, t->ln1_g, s.r, s.m, ld, site(t->drop_ao), lnp[1], nb, gln1, N, H, t->ln_eps, sd))) return rc;
  if ((rc = dw(s.m, t->ctx, N, ld, H, H, part[2], gWao, sd))) return rc;
  {   // d(ctx) = m Wao (rows: the attention backward streams its heads)
    mh_gemm_desc d{};
    d.A = s.m; d.lda = ld; d.a_panel = 1; d.W = t->wao_t; d.ldw = H; d.w_panel = 1; d.out = s.t; d.ldo = H; d.M = N; d.N = H; d.K = H;
    if ((rc = mh_gemm_desc_launch(&d, stream))) return rc;
  }
  {
    bf16* dq = reinterpret_cast<bf16*>(s.dqkv);
    const int64_t blk = (int64_t)(H / 32) * ld * 32;      // elements of one of the three column blocks in panel form
    const mh_dropout* da = site(t->drop_attn);
    if ((rc = mh_attention_stream_bwd_layout(qkv, qkv + H, qkv + 2 * H, s.t, t->ctx, 1, ld, t->lse, reinterpret_cast<float*>(s.D), dq, dq + blk, dq + 2 * blk, ld, 1,
                                             B, L, nh, dh, scale, (int64_t)L * 3 * H, dh, 3 * H, (int64_t)L * H, dh, H, da ? t->keep_bits : nullptr,
                                             da ? t->drop_attn.p : 0.f, stream))) return rc;
  }
  if ((rc = dw(s.dqkv, t->x, N, ld, 3 * H, H, part[3], gWqkv, sd))) return rc;
  {   // dx = dqkv Wqkv + r (rows)
    mh_gemm_desc d{};
    d.A = s.dqkv; d.lda = ld; d.a_panel = 1; d.W = t->wqkv_t; d.ldw = H; d.w_panel = 1;
    d.residual = s.r; d.ldr = H; d.out = t->dx; d.ldo = H; d.M = N; d.N = H; d.K = 3 * H;
    if ((rc = mh_gemm_desc_launch(&d, stream))) return rc;
  }
  return join(sd);   // the caller's stream sees every gradient; the next layer's kernels may reuse the partial buffers
}
