// Kernels of the training path (training_losses forward + backward, models/diffusion.py:594-699 driven by
// utils/train_util.py:188-232): layout permutes and transposes that turn every backward product into the
// same "A W^T" GEMM the forward uses, activation / LayerNorm / softmax / cross-entropy / squared-error
// backward, column sums for bias and LayerNorm-parameter gradients, embedding scatter-add.
// All HBM-bound streaming kernels; element type T in {float, bf16}, reductions always in fp32.
#include "common.h"

namespace {

constexpr int TB = 256;
inline int tgrid(int64_t n) {
  int64_t b = (n + TB - 1) / TB;
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

// ------------------------------------------------------------------ transposes / head permutes
// out[b][c][r] = in[b][r][c]  (32 x 32 tiles through LDS, +1 padding)
template <typename T>
__global__ void transpose_kernel(const T* __restrict__ in, int64_t ld_in, int64_t s_in, T* __restrict__ out, int64_t ld_out,
                                 int64_t s_out, int rows, int cols) {
  __shared__ T tile[32][33];
  const T* src = in + (int64_t)blockIdx.z * s_in;
  T* dst = out + (int64_t)blockIdx.z * s_out;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[(int64_t)r * ld_in + c] : from_f32<T>(0.f);
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) dst[(int64_t)c * ld_out + r] = tile[tx][i];
  }
}

// bf16 fast path: 64 x 64 tiles, 16-byte loads along the input rows and 16-byte stores along the output rows
// (rows % 64 == 0, cols % 64 == 0, ld_in / ld_out / strides multiples of 8)
__global__ __launch_bounds__(256) void transpose64_kernel(const bf16* __restrict__ in, int64_t ld_in, int64_t s_in, bf16* __restrict__ out,
                                                          int64_t ld_out, int64_t s_out) {
  __shared__ bf16 tile[64][64 + 8];
  const bf16* src = in + (int64_t)blockIdx.z * s_in;
  bf16* dst = out + (int64_t)blockIdx.z * s_out;
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < 64 * 8; i += 256) {
    const int r = i >> 3, c = i & 7;
    *reinterpret_cast<f32x4*>(&tile[r][c * 8]) = *reinterpret_cast<const f32x4*>(src + (int64_t)(r0 + r) * ld_in + c0 + c * 8);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 8; i += 256) {
    const int c = i >> 3, rc = i & 7;         // output row c0 + c, output columns r0 + 8 rc .. + 7
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = tile[8 * rc + j][c];
    *reinterpret_cast<bf16x8*>(dst + (int64_t)(c0 + c) * ld_out + r0 + 8 * rc) = v;
  }
}

// mode 0: tokens [B*L, ld] (head h at cols [h*dh,(h+1)*dh)) -> heads [B, nh, L, dh]
// mode 1: heads -> tokens;  mode 2: tokens -> heads transposed [B, nh, dh, L];  mode 3: as mode 2 with the positions of
// every group of 16 stored as 0-3, 8-11, 4-7, 12-15 (the streaming attention kernels' "P-operand" order)
template <typename T>
__global__ void head_permute_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t ld_tok, int B, int L, int nh,
                                    int dh, int mode) {
  const int64_t total = (int64_t)B * L * nh * dh;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    // i enumerates the OUTPUT in its natural order (coalesced writes)
    if (mode == 0) {
      const int d = (int)(i % dh); int64_t r = i / dh;
      const int l = (int)(r % L); r /= L;
      const int h = (int)(r % nh); const int64_t b = r / nh;
      out[i] = in[(b * L + l) * ld_tok + h * dh + d];
    } else if (mode == 1) {
      const int c = (int)(i % (nh * dh)); const int64_t tok = i / (nh * dh);
      const int h = c / dh, d = c % dh;
      const int64_t b = tok / L, l = tok % L;
      out[tok * ld_tok + c] = in[((b * nh + h) * L + l) * dh + d];
    } else {
      int l = (int)(i % L); int64_t r = i / L;
      if (mode == 3) l = (l & ~15) | ((((l >> 3) & 1) | ((l >> 1) & 2)) << 2) | (l & 3);   // the permutation is an involution
      const int d = (int)(r % dh); r /= dh;
      const int h = (int)(r % nh); const int64_t b = r / nh;
      out[i] = in[(b * L + l) * ld_tok + h * dh + d];
    }
  }
}

// ------------------------------------------------------------------ column sums (deterministic two-stage)
// out[b, l, :] = pos[l, :] + x[b, l, :] + emb[b, :]   (network.py:146-148, kept as a tensor for the backward)
template <typename T>
__global__ void add_pos_time_kernel(const T* __restrict__ x, int64_t ldx, const float* __restrict__ pos, const float* __restrict__ emb,
                                    T* __restrict__ out, int B, int L, int H) {
  const int64_t total = (int64_t)B * L * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % H); const int64_t tok = i / H;
    const int64_t b = tok / L, l = tok % L;
    out[i] = from_f32<T>((pos[l * H + c] + to_f32(x[tok * ldx + c])) + emb[b * H + c]);
  }
}

// the same, 8 columns per thread (bf16, H % 8 == 0, ldx % 8 == 0): 16-byte loads and stores instead of one element, two 64-bit divisions each
__global__ void add_pos_time8_kernel(const bf16* __restrict__ x, int64_t ldx, const float* __restrict__ pos, const float* __restrict__ emb,
                                     bf16* __restrict__ out, int B, int L, int H) {
  const int h8 = H >> 3;
  const int64_t total = (int64_t)B * L * h8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % h8) << 3; const int64_t tok = i / h8;
    const int64_t b = tok / L, l = tok % L;
    float xv[8], pv[8], ev[8], o[8];
    load8(x + tok * ldx + c, xv);
    load8(pos + l * H + c, pv);
    load8(emb + b * H + c, ev);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (pv[e] + xv[e]) + ev[e];
    store8(out + tok * H + c, o);
  }
}

// bf16 fast paths of head_permute.  Modes 0 / 1: 16 bytes (8 head-dim elements) per thread.
__global__ void head_permute8_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, int64_t ld_tok, int B, int L, int nh, int dh8,
                                     int mode) {
  const int64_t total = (int64_t)B * L * nh * dh8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % dh8); int64_t r = i / dh8;
    if (mode == 0) {        // i enumerates [b][h][l][d/8]
      const int l = (int)(r % L); r /= L;
      const int h = (int)(r % nh); const int64_t b = r / nh;
      *reinterpret_cast<f32x4*>(out + i * 8) = *reinterpret_cast<const f32x4*>(in + (b * L + l) * ld_tok + (h * dh8 + c) * 8);
    } else {                // i enumerates [tok][h][d/8]
      const int h = (int)(r % nh); const int64_t tok = r / nh;
      const int64_t b = tok / L, l = tok % L;
      *reinterpret_cast<f32x4*>(out + tok * ld_tok + (h * dh8 + c) * 8) = *reinterpret_cast<const f32x4*>(in + (((b * nh + h) * L + l) * dh8 + c) * 8);
    }
  }
}
// Modes 2 / 3 (tokens -> [B, nh, dh, L], optionally with the P-operand position order): 64-token tiles through LDS,
// 16-byte loads along the head dim, 16-byte stores along the sequence.  grid (L / 64, B * nh), 256 threads, DH <= 128.
template <int DH>
__global__ __launch_bounds__(256) void head_transpose_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, int64_t ld_tok, int L,
                                                             int nh, int perm) {
  __shared__ bf16 tile[64][DH + 8];
  const int bh = blockIdx.y, b = bh / nh, h = bh % nh, l0 = blockIdx.x * 64;
  if (perm == 2 && blockIdx.x == 0 && blockIdx.y == 0)      // mode 4: zero the 256-element slack behind the last row
    out[(int64_t)gridDim.y * DH * L + threadIdx.x] = (bf16)0.0f;
  perm = perm != 0;
  constexpr int CPR = DH / 8;
  for (int i = threadIdx.x; i < 64 * CPR; i += 256) {
    const int l = i / CPR, c = i % CPR;
    *reinterpret_cast<f32x4*>(&tile[l][c * 8]) = *reinterpret_cast<const f32x4*>(in + ((int64_t)b * L + l0 + l) * ld_tok + h * DH + c * 8);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < DH * 8; i += 256) {
    const int d = i >> 3, c = i & 7;          // output chunk c: positions 8c .. 8c+7 of this tile
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int l = 8 * c + j;
      if (perm) l = (l & ~15) | ((((l >> 3) & 1) | ((l >> 1) & 2)) << 2) | (l & 3);
      v[j] = tile[l][d];
    }
    *reinterpret_cast<bf16x8*>(out + ((int64_t)bh * DH + d) * L + l0 + 8 * c) = v;
  }
}

template <typename T>
__global__ void colsum_partial_kernel(const T* __restrict__ in_all, int64_t ld, int64_t rows, int cols, float* __restrict__ part_all,
                                      int64_t s_in) {
  // grid (ceil(cols/64), P, batch): block handles 64 columns over rows p, p+P, ... ; 4 row-lanes x 64 cols
  const T* in = in_all + (int64_t)blockIdx.z * s_in;
  float* part = part_all + (int64_t)blockIdx.z * gridDim.y * cols;
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6, P = gridDim.y;
  __shared__ float red[4][64];
  float s = 0.f;
  if (c < cols)
    for (int64_t r = (int64_t)blockIdx.y * 4 + rl; r < rows; r += (int64_t)P * 4) s += to_f32(in[r * ld + c]);
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < cols) part[(int64_t)blockIdx.y * cols + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ part_all, int P, int cols, float* __restrict__ out_all, int accumulate) {
  // block = 64 columns x 16 partial-lanes: every lane issues all of its loads (P / 16 <= 64) before the first add, so the fold
  // costs one memory round trip instead of a chain of them (it is pure latency: P x cols floats); fixed summation order ->
  // reproducible: lane pl adds partials pl, pl + 16, ... in order, lanes are folded 0..15 in order
  const float* part = part_all + (int64_t)blockIdx.y * P * cols;
  float* out = out_all + (int64_t)blockIdx.y * cols;
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
  __shared__ float red[16][64];
  float s = 0.f;
  if (c < cols) {
    int p = pl;
    for (; p + 48 < P; p += 64) {
      const float a0 = part[(int64_t)p * cols + c], a1 = part[(int64_t)(p + 16) * cols + c];
      const float a2 = part[(int64_t)(p + 32) * cols + c], a3 = part[(int64_t)(p + 48) * cols + c];
      s = (((s + a0) + a1) + a2) + a3;
    }
    for (; p < P; p += 16) s += part[(int64_t)p * cols + c];
  }
  red[pl][threadIdx.x & 63] = s;
  __syncthreads();
  if (pl == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w][threadIdx.x];
    out[c] = accumulate ? out[c] + t : t;
  }
}

// 16-byte loads: a wave covers 8 x 64 = 512 consecutive columns of one row per instruction (cols % 8 == 0, ld % 8 == 0)
template <typename T>
__global__ void colsum8_partial_kernel(const T* __restrict__ in_all, int64_t ld, int64_t rows, int cols, float* __restrict__ part_all,
                                       int64_t s_in) {
  constexpr int V = 16 / sizeof(T);
  const T* in = in_all + (int64_t)blockIdx.z * s_in;
  float* part = part_all + (int64_t)blockIdx.z * gridDim.y * cols;
  const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6, P = gridDim.y;
  const int c = (blockIdx.x * 64 + lane) * V;
  __shared__ float red[4][64 * V];
  float s[V];
#pragma unroll
  for (int j = 0; j < V; ++j) s[j] = 0.f;
  if (c < cols)
    for (int64_t r = (int64_t)blockIdx.y * 4 + rl; r < rows; r += (int64_t)P * 4) {
      float v[V];
      if constexpr (sizeof(T) == 2) load8(reinterpret_cast<const bf16*>(in) + r * ld + c, v);
      else { const f32x4 t = *reinterpret_cast<const f32x4*>(in + r * ld + c); v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
#pragma unroll
      for (int j = 0; j < V; ++j) s[j] += v[j];
    }
#pragma unroll
  for (int j = 0; j < V; ++j) red[rl][lane * V + j] = s[j];
  __syncthreads();
  if (rl == 0 && c < cols) {
#pragma unroll
    for (int j = 0; j < V; ++j)
      part[(int64_t)blockIdx.y * cols + c + j] = (red[0][lane * V + j] + red[1][lane * V + j]) + (red[2][lane * V + j] + red[3][lane * V + j]);
  }
}

// ------------------------------------------------------------------ activations
template <typename T>
__global__ void act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, int act) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = to_f32(x[i]);
    float r = v;
    if (act == MH_ACT_TANH) r = tanhf(v);
    else if (act == MH_ACT_GELU_ERF) r = gelu_erf(v);
    else if (act == MH_ACT_SILU) r = silu(v);
    y[i] = from_f32<T>(r);
  }
}
// dx = dy * act'(x)   (x = saved PRE-activation)
template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ dx, int64_t n, int act) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = to_f32(x[i]), g = to_f32(dy[i]);
    float d = 1.f;
    if (act == MH_ACT_TANH) { const float t = tanhf(v); d = 1.f - t * t; }
    else if (act == MH_ACT_GELU_ERF) d = 0.5f * (1.f + erff(v * 0.70710678118654752440f)) + v * 0.3989422804014327f * expf(-0.5f * v * v);
    else if (act == MH_ACT_SILU) { const float sg = 1.f / (1.f + expf(-v)); d = sg * (1.f + v * (1.f - sg)); }
    dx[i] = from_f32<T>(g * d);
  }
}

// ------------------------------------------------------------------ LayerNorm backward
// one wave per row (H <= 2048); block = 4 waves walking rows blockIdx.x*4 + w, + gridDim.x*4, ...;
// dgamma / dbeta partials accumulate in registers over the block's rows and are written per block.
// LNCH = 16-byte chunks per lane (ceil(H / 512)): a compile-time bound keeps the per-row and per-column registers of a narrow model
// at a quarter of the H = 2048 build's, i.e. four times the waves per SIMD to hide the row's dependent reductions behind
// DROPM: also writes dxm = dx o keep / (1 - p) for the dropout site of the dense layer in FRONT of this LayerNorm (the dense branch of
// its backward reads dxm, the residual branch dx) - the mask re-created from the site's descriptor exactly as mh_dropout_fwd does on
// the stored bf16 dx, so the separate re-application pass (a read and a write of the tensor, one launch) disappears.
template <typename T, int LNCH, bool DROPM = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ gamma,
                                                     T* __restrict__ dx, float* __restrict__ pg, float* __restrict__ pb,
                                                     int64_t rows, int H, float eps, T* __restrict__ dxm = nullptr, const DropArgs drop = DropArgs{},
                                                     int64_t ldm = 0, int m_panel = 0) {
  // (no FP contraction: the instantiation that also writes the dropped copy must round dx exactly as the one that does not)
#pragma clang fp contract(off)
  __shared__ float red[2][4][512 * LNCH];  // [dgamma|dbeta][wave][col]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nch = H >> 3;
  float ag[LNCH][8], ab[LNCH][8];
#pragma unroll
  for (int i = 0; i < LNCH; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[i][e] = 0.f; ab[i][e] = 0.f; }
  // the NEXT row's x / dy are fetched while this row's three dependent wave reductions run (round 6: a wave walks ~8 rows one after the
  // other, and with the loads at the head of each row's body their latency stood in front of every row)
  float xn[LNCH][8], gn[LNCH][8];
  auto fetch = [&](int64_t r) {
#pragma unroll
    for (int i = 0; i < LNCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) { load8(x + r * H + c * 8, xn[i]); load8(dy + r * H + c * 8, gn[i]); }
    }
  };
  const int64_t row_step = (int64_t)gridDim.x * 4;
  if ((int64_t)blockIdx.x * 4 + w < rows) fetch((int64_t)blockIdx.x * 4 + w);
  for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < rows; row += row_step) {
    float xv[LNCH][8], gv[LNCH][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LNCH; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) { xv[i][e] = xn[i][e]; gv[i][e] = gn[i][e]; }
    if (row + row_step < rows) fetch(row + row_step);
#pragma unroll
    for (int i = 0; i < LNCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += xv[i][e];
      }
    }
    const float mean = wave_sum(sum) / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < LNCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = xv[i][e] - mean; sq += d * d; }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)H + eps);
    float s1 = 0.f, s2 = 0.f;   // sum(dy*g), sum(dy*g*xhat)
#pragma unroll
    for (int i = 0; i < LNCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        float g[8];
        load8(gamma + c * 8, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (xv[i][e] - mean) * rstd;
          const float dg = gv[i][e] * g[e];
          s1 += dg; s2 += dg * xh;
          ag[i][e] += gv[i][e] * xh;
          ab[i][e] += gv[i][e];
          xv[i][e] = xh; gv[i][e] = dg;
        }
      }
    }
    s1 = wave_sum(s1) / (float)H; s2 = wave_sum(s2) / (float)H;
#pragma unroll
    for (int i = 0; i < LNCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rstd * (gv[i][e] - s1 - xv[i][e] * s2);
        store8(dx + row * H + c * 8, o);
        if constexpr (DROPM) {
          if (drop.thr || drop.mask) {   // (block-uniform; a site without dropout still gets its second copy: the panel-layout GEMM operand)
            const uint32_t km = drop_keep8_at(drop, (uint64_t)row * H + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (km >> e) & 1u ? to_f32(from_f32<T>(o[e])) * drop.rscale : 0.f;   // (of the ROUNDED dx, as the separate pass did)
          }
          // row-major [rows][ldm] or K32 panels [H / 32][ldm rows][32] (what the input-gradient and weight-gradient GEMMs read: csrc/train_layer.hip)
          store8(m_panel ? dxm + ((int64_t)(c >> 2) * ldm + row) * 32 + (c & 3) * 8 : dxm + row * ldm + c * 8, o);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < LNCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nch)
#pragma unroll
      for (int e = 0; e < 8; ++e) { red[0][w][c * 8 + e] = ag[i][e]; red[1][w][c * 8 + e] = ab[i][e]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 256) {
    pg[(int64_t)blockIdx.x * H + c] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
    pb[(int64_t)blockIdx.x * H + c] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
  }
}

// ------------------------------------------------------------------ row softmax forward / backward (attention probabilities)
template <typename T>
__global__ void softmax_rows_kernel(T* __restrict__ s, int64_t rows, int L, int64_t ld, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  T* p = s + row * ld;
  float mx = -INFINITY;
  for (int c = lane; c < L; c += 64) mx = fmaxf(mx, to_f32(p[c]) * scale);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < L; c += 64) sum += expf(to_f32(p[c]) * scale - mx);
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int c = lane; c < L; c += 64) p[c] = from_f32<T>(expf(to_f32(p[c]) * scale - mx) * inv);
}
// ds = p * (dp - sum(dp * p)) * scale   (in place on dp)
template <typename T>
__global__ void softmax_bwd_rows_kernel(const T* __restrict__ p, T* __restrict__ dp, int64_t rows, int L, int64_t ld, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* pr = p + row * ld;
  T* dr = dp + row * ld;
  float dot = 0.f;
  for (int c = lane; c < L; c += 64) dot += to_f32(pr[c]) * to_f32(dr[c]);
  dot = wave_sum(dot);
  for (int c = lane; c < L; c += 64) dr[c] = from_f32<T>(to_f32(pr[c]) * (to_f32(dr[c]) - dot) * scale);
}

// ------------------------------------------------------------------ token cross-entropy over logits [N, V] fp32
__global__ void ce_fwd_kernel(const float* __restrict__ logits, int64_t ld, const int32_t* __restrict__ target,
                              float* __restrict__ loss, float* __restrict__ lse, int64_t n, int V) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float* p = logits + row * ld;
  float mx = -INFINITY;
  for (int c = lane; c < V; c += 64) mx = fmaxf(mx, p[c]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < V; c += 64) sum += expf(p[c] - mx);
  sum = wave_sum(sum);
  const float l = mx + logf(sum);
  if (lane == 0) {
    int t = target[row];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    lse[row] = l;
    loss[row] = l - p[t];
  }
}
// dlogits[n, v] = (softmax - onehot) * g[n]   (written in the compute dtype for the following GEMMs)
template <typename T>
__global__ void ce_bwd_kernel(const float* __restrict__ logits, int64_t ld, const int32_t* __restrict__ target,
                              const float* __restrict__ lse, const float* __restrict__ g, T* __restrict__ dl, int64_t ldd,
                              int64_t n, int V, int Vpad) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float* p = logits + row * ld;
  const float l = lse[row], gs = g[row];
  int t = target[row];
  t = t < 0 ? 0 : (t >= V ? V - 1 : t);
  for (int c = lane; c < Vpad; c += 64) {
    float v = 0.f;
    if (c < V) v = (expf(p[c] - l) - (c == t ? 1.f : 0.f)) * gs;
    dl[row * ldd + c] = from_f32<T>(v);
  }
}

// ------------------------------------------------------------------ squared error: per-batch mean and gradient
// out[b] = mean_i (scale_a * a[b,i] - b[b,i])^2        (b may be NULL = 0)
__global__ __launch_bounds__(1024) void sqdiff_mean_kernel(const float* __restrict__ a, const float* __restrict__ bb, float scale_a,
                                                           float* __restrict__ out, int64_t per_batch) {
  __shared__ float red[16];
  const int b = blockIdx.x;
  const float* pa = a + (int64_t)b * per_batch;
  const float* pb = bb ? bb + (int64_t)b * per_batch : nullptr;
  float s4[4] = {0.f, 0.f, 0.f, 0.f};
  if (per_batch % 4 == 0 && (reinterpret_cast<uintptr_t>(pa) & 15) == 0 && (!pb || (reinterpret_cast<uintptr_t>(pb) & 15) == 0)) {
    for (int64_t i = (int64_t)threadIdx.x * 4; i < per_batch; i += (int64_t)blockDim.x * 4) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(pa + i);
      f32x4 y = {0.f, 0.f, 0.f, 0.f};
      if (pb) y = *reinterpret_cast<const f32x4*>(pb + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = scale_a * x[e] - y[e]; s4[e] += d * d; }
    }
  } else {
    for (int64_t i = threadIdx.x; i < per_batch; i += blockDim.x) {
      const float d = scale_a * pa[i] - (pb ? pb[i] : 0.f);
      s4[0] += d * d;
    }
  }
  float s = wave_sum((s4[0] + s4[1]) + (s4[2] + s4[3]));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    out[b] = t / (float)per_batch;
  }
}
// da[b,i] (+)= g[b] * 2 * scale_a * (scale_a*a - b) / per_batch ;  db = -da / scale_a (optional)
__global__ void sqdiff_bwd_kernel(const float* __restrict__ a, const float* __restrict__ bb, float scale_a, const float* __restrict__ g,
                                  float* __restrict__ da, float* __restrict__ db, int accumulate, int B, int64_t per_batch) {
  const int64_t total = (int64_t)B * per_batch;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / per_batch);
    const float d = scale_a * a[i] - (bb ? bb[i] : 0.f);
    const float base = g[b] * 2.0f * d / (float)per_batch;
    if (da) da[i] = (accumulate ? da[i] : 0.f) + base * scale_a;
    if (db) db[i] = (accumulate ? db[i] : 0.f) - base;
  }
}

// ------------------------------------------------------------------ misc
template <typename T>
__global__ void add_inplace_kernel(T* __restrict__ dst, const T* __restrict__ src, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = from_f32<T>(to_f32(dst[i]) + to_f32(src[i]));
}
// table[v, :] += sum of the rows src[t, :] with ids[t] == v  (embedding-weight gradient), without atomics: block
// (v, chunk) scans its chunk of the ids 64 at a time (ballot) and adds the matching rows in index order into
// partial[chunk][v][:]; a second kernel folds the chunks in order.  Frequent tokens (padding is most of a batch) are
// spread over the chunks, and the summation order does not change from run to run.
// RV vocabulary rows per block (round 6): a block scans its token chunk ONCE for RV rows - with one row per block (rounds 1 - 5) the 729
// blocks of a chunk each re-scanned the same ids and the scan, not the gather, was the kernel's time (120 us for a 16 MB input)
constexpr int SCATTER_RV = 8;
__global__ __launch_bounds__(256) void scatter_rows_partial_kernel(const float* __restrict__ src, const int32_t* __restrict__ ids,
                                                                   float* __restrict__ partial, int64_t n, int E, int V, int64_t chunk) {
  extern __shared__ float red[];   // [4 waves][RV][256]
  const int v0 = blockIdx.x * SCATTER_RV, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tb = (int64_t)blockIdx.y * chunk, te = tb + chunk < n ? tb + chunk : n;
  for (int e0 = 0; e0 < E; e0 += 256) {        // 4 columns per lane per pass
    float acc[SCATTER_RV][4];
#pragma unroll
    for (int k = 0; k < SCATTER_RV; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[k][c] = 0.f;
    for (int64_t t0 = tb + (int64_t)wave * 64; t0 < te; t0 += 256) {
      int id = -1;
      if (t0 + lane < te) { id = ids[t0 + lane]; id = id < 0 ? 0 : (id >= V ? V - 1 : id); }
#pragma unroll
      for (int k = 0; k < SCATTER_RV; ++k) {
        unsigned long long m = __builtin_amdgcn_ballot_w64(id == v0 + k);
        // the matches of a group four at a time: their row loads are independent and go out together, the adds keep the token order (a
        // frequent token - the padding id is a third of a ComMU batch - made its block walk hundreds of dependent loads one by one)
        while (m) {
          int j[4];
          float rv[4][4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            j[u] = m ? __builtin_ctzll(m) : -1;
            m &= m - 1;                          // (0 & anything = 0: stays empty)
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float* row = src + (t0 + (j[u] < 0 ? 0 : j[u])) * E + e0;
#pragma unroll
            for (int c = 0; c < 4; ++c) rv[u][c] = (j[u] >= 0 && e0 + lane + 64 * c < E) ? row[lane + 64 * c] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (j[u] >= 0) {
#pragma unroll
              for (int c = 0; c < 4; ++c) acc[k][c] += rv[u][c];
            }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < SCATTER_RV; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c) red[(wave * SCATTER_RV + k) * 256 + lane + 64 * c] = acc[k][c];
    __syncthreads();
    for (int o = threadIdx.x; o < SCATTER_RV * 256; o += 256) {
      const int k = o >> 8, col = o & 255;
      if (v0 + k < V && e0 + col < E) {
        float* dst = partial + ((int64_t)blockIdx.y * V + v0 + k) * E;
        dst[e0 + col] = (red[o] + red[SCATTER_RV * 256 + o]) + (red[2 * SCATTER_RV * 256 + o] + red[3 * SCATTER_RV * 256 + o]);
      }
    }
    __syncthreads();
  }
}
__global__ void scatter_rows_final_kernel(const float* __restrict__ partial, float* __restrict__ table, int nchunks, int64_t VE) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < VE; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < nchunks; ++c) s += partial[(int64_t)c * VE + i];
    table[i] += s;
  }
}
// q_sample backward (diffusion.py:245-255): dst[b,i] (+)= src[b,i] * (mask[token] == 0 ? 1 : scale[b])
__global__ void scale_rows_kernel(const float* __restrict__ src, const float* __restrict__ scale, const int32_t* __restrict__ mask,
                                  float* __restrict__ dst, int accumulate, int B, int64_t per_batch, int E) {
  const int64_t total = (int64_t)B * per_batch;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / per_batch);
    const bool anchored = mask && mask[i / E] == 0;
    const float v = src[i] * (anchored ? 1.0f : (scale ? scale[b] : 1.0f));
    dst[i] = (accumulate ? dst[i] : 0.f) + v;
  }
}

// ------------------------------------------------------------------ fused AdamW + EMA + grad-norm (multi-tensor)
// One launch updates EVERY parameter tensor: block b handles chunk b = (tensor, offset) from a device table.
// Replaces the reference's per-tensor foreach: AdamW.step + 3 x update_ema + grad-norm = ~600 launches and
// ~200 .item() syncs per optimizer step (utils/train_util.py:246-280, :21-31).
#pragma clang fp contract(off)
// One AdamW + EMA element update (torch.optim.AdamW, decoupled weight decay, single-tensor order of operations; every host-side
// scalar - 1-beta, 1-lr*wd, lr/bias1, 1-rate - is evaluated in double by the host like torch does)
__device__ __forceinline__ float adamw_elem(float p, float g, float& m, float& v, const mh_opt_hparams& hp) {
  p = p * hp.decay_mul;
  m = m * hp.beta1 + g * hp.one_minus_beta1;
  v = v * hp.beta2 + (g * g) * hp.one_minus_beta2;
  const float denom = sqrtf(v) / hp.bias2_sqrt + hp.eps;
  return p - hp.step_size * (m / denom);
}
// 16-byte accesses (a chunk starts on a multiple of 65536 elements of a 256-B aligned tensor); the parameter stream and the EMA
// copies are read and written exactly once per step: 13 streams of 4 B per parameter, HBM-bound
template <int NEMA>
__device__ __forceinline__ void adamw_ema_chunk(const mh_opt_tensor& t, const mh_opt_chunk& ck, const mh_opt_hparams& hp) {
  const int64_t n4 = ck.count >> 2;
  if (!t.grad) {   // no gradient (frozen / unused parameter): torch.optim.AdamW skips it, update_ema (train_util.py:21-31) does not
    if constexpr (NEMA > 0) {
      for (int64_t i = ck.offset + threadIdx.x; i < ck.offset + ck.count; i += blockDim.x) {
        const float p = t.param[i];
#pragma unroll
        for (int k = 0; k < NEMA; ++k) t.ema[k][i] = t.ema[k][i] * hp.ema_rate[k] + p * hp.ema_one_minus[k];
      }
    }
    return;
  }
  f32x4* __restrict__ P = reinterpret_cast<f32x4*>(t.param + ck.offset);
  const f32x4* __restrict__ G = reinterpret_cast<const f32x4*>(t.grad + ck.offset);
  f32x4* __restrict__ M = reinterpret_cast<f32x4*>(t.exp_avg + ck.offset);
  f32x4* __restrict__ V = reinterpret_cast<f32x4*>(t.exp_avg_sq + ck.offset);
  for (int64_t i = threadIdx.x; i < n4; i += blockDim.x) {
    f32x4 p = P[i], m = M[i], v = V[i];
    const f32x4 g = __builtin_nontemporal_load(G + i);
    f32x4 e[NEMA > 0 ? NEMA : 1];
#pragma unroll
    for (int k = 0; k < NEMA; ++k) e[k] = reinterpret_cast<const f32x4*>(t.ema[k] + ck.offset)[i];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float mm = m[c], vv = v[c];
      p[c] = adamw_elem(p[c], g[c], mm, vv, hp);
      m[c] = mm; v[c] = vv;
#pragma unroll
      for (int k = 0; k < NEMA; ++k) e[k][c] = e[k][c] * hp.ema_rate[k] + p[c] * hp.ema_one_minus[k];   // update_ema, train_util.py:21-31
    }
    P[i] = p;
    __builtin_nontemporal_store(m, M + i);
    __builtin_nontemporal_store(v, V + i);
#pragma unroll
    for (int k = 0; k < NEMA; ++k) __builtin_nontemporal_store(e[k], reinterpret_cast<f32x4*>(t.ema[k] + ck.offset) + i);
  }
  for (int64_t i = ck.offset + (n4 << 2) + threadIdx.x; i < ck.offset + ck.count; i += blockDim.x) {   // count % 4 tail
    float m = t.exp_avg[i], v = t.exp_avg_sq[i];
    const float p = adamw_elem(t.param[i], t.grad[i], m, v, hp);
    t.exp_avg[i] = m; t.exp_avg_sq[i] = v; t.param[i] = p;
#pragma unroll
    for (int k = 0; k < NEMA; ++k) t.ema[k][i] = t.ema[k][i] * hp.ema_rate[k] + p * hp.ema_one_minus[k];
  }
}
__global__ __launch_bounds__(256) void adamw_ema_kernel(const mh_opt_tensor* __restrict__ tensors, const mh_opt_chunk* __restrict__ chunks,
                                                       mh_opt_hparams hp) {
  const mh_opt_chunk ck = chunks[blockIdx.x];
  const mh_opt_tensor t = tensors[ck.tensor];
  switch (hp.n_ema) {
    case 0: adamw_ema_chunk<0>(t, ck, hp); break;
    case 1: adamw_ema_chunk<1>(t, ck, hp); break;
    case 2: adamw_ema_chunk<2>(t, ck, hp); break;
    case 3: adamw_ema_chunk<3>(t, ck, hp); break;
    default: adamw_ema_chunk<4>(t, ck, hp); break;
  }
}
__global__ __launch_bounds__(256) void sumsq_chunks_kernel(const mh_opt_tensor* __restrict__ tensors, const mh_opt_chunk* __restrict__ chunks,
                                                          float* __restrict__ partial) {
  __shared__ float red[4];
  const mh_opt_chunk ck = chunks[blockIdx.x];
  if (!tensors[ck.tensor].grad) {   // a parameter without a gradient adds nothing (train_util.py:277 `if p.grad is not None`)
    if (threadIdx.x == 0) partial[blockIdx.x] = 0.f;
    return;
  }
  const float* g = tensors[ck.tensor].grad + ck.offset;
  // fixed per-thread order: 16-byte pieces strided by the block, then the count % 4 tail
  float s = 0.f;
  const int64_t n4 = ck.count >> 2;
  for (int64_t i = threadIdx.x; i < n4; i += blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
    s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  for (int64_t i = (n4 << 2) + threadIdx.x; i < ck.count; i += blockDim.x) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// torch.nn.utils.clip_grad_norm_: g *= min(1, max_norm / (total_norm + 1e-6)) with the norm read from the device
__global__ __launch_bounds__(256) void clip_grads_kernel(const mh_opt_tensor* __restrict__ tensors, const mh_opt_chunk* __restrict__ chunks,
                                                        const float* __restrict__ norm, float max_norm) {
  const float coef = fminf(max_norm / (norm[0] + 1e-6f), 1.0f);
  if (coef >= 1.0f) return;
  const mh_opt_chunk ck = chunks[blockIdx.x];
  if (!tensors[ck.tensor].grad) return;
  float* g = const_cast<float*>(tensors[ck.tensor].grad) + ck.offset;
  const int64_t n4 = ck.count >> 2;
  for (int64_t i = threadIdx.x; i < n4; i += blockDim.x) {
    f32x4 v = reinterpret_cast<f32x4*>(g)[i];
    v[0] *= coef; v[1] *= coef; v[2] *= coef; v[3] *= coef;
    reinterpret_cast<f32x4*>(g)[i] = v;
  }
  for (int64_t i = (n4 << 2) + threadIdx.x; i < ck.count; i += blockDim.x) g[i] *= coef;
}
__global__ void sum_partials_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
}

}  // namespace

#define MH_DTYPE_SWITCH(dtype, CALL_BF16, CALL_F32, what)                                 \
  if ((dtype) == MH_BF16) { CALL_BF16; } else if ((dtype) == MH_F32) { CALL_F32; } else { \
    mh_set_error(what ": unknown dtype %d", (dtype)); return MH_ERR_INVALID; }

extern "C" int mh_transpose(const void* in, int64_t ld_in, int64_t stride_in, void* out, int64_t ld_out, int64_t stride_out,
                            int rows, int cols, int batch, int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(in && out && rows > 0 && cols > 0 && batch > 0 && batch <= 65535, "transpose: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MH_BF16 && rows % 64 == 0 && cols % 64 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0 && stride_in % 8 == 0 && stride_out % 8 == 0 &&
      (reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    MH_LAUNCH(transpose64_kernel, dim3(cols / 64, rows / 64, batch), dim3(256), 0, s, (const bf16*)in, ld_in, stride_in, (bf16*)out, ld_out,
              stride_out);
    MH_CHECK_LAUNCH();
    return MH_OK;
  }
  dim3 grid((cols + 31) / 32, (rows + 31) / 32, batch);
  MH_DTYPE_SWITCH(dtype,
                  MH_LAUNCH((transpose_kernel<bf16>), grid, dim3(256), 0, s, (const bf16*)in, ld_in, stride_in, (bf16*)out, ld_out, stride_out, rows, cols),
                  MH_LAUNCH((transpose_kernel<float>), grid, dim3(256), 0, s, (const float*)in, ld_in, stride_in, (float*)out, ld_out, stride_out, rows, cols),
                  "transpose");
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_head_permute(const void* in, void* out, int64_t ld_tok, int B, int L, int nh, int dh, int mode, int dtype,
                               mh_stream_t stream) {
  MH_CHECK_ARG(in && out && B > 0 && L > 0 && nh > 0 && dh > 0 && mode >= 0 && mode <= 4 && (mode < 3 || L % 16 == 0), "head_permute: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const bool vec_ok = dtype == MH_BF16 && dh % 8 == 0 && ld_tok % 8 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  if (vec_ok && mode <= 1) {
    MH_LAUNCH(head_permute8_kernel, dim3(tgrid((int64_t)B * L * nh * (dh / 8))), dim3(TB), 0, s, (const bf16*)in, (bf16*)out, ld_tok, B, L,
              nh, dh / 8, mode);
    MH_CHECK_LAUNCH();
    return MH_OK;
  }
  if (vec_ok && mode >= 2 && L % 64 == 0 && (dh == 32 || dh == 64 || dh == 128)) {
    const dim3 grid(L / 64, B * nh);
    const int perm = mode == 4 ? 2 : (mode == 3 ? 1 : 0);
    if (dh == 32) MH_LAUNCH((head_transpose_kernel<32>), grid, dim3(256), 0, s, (const bf16*)in, (bf16*)out, ld_tok, L, nh, perm);
    else if (dh == 64) MH_LAUNCH((head_transpose_kernel<64>), grid, dim3(256), 0, s, (const bf16*)in, (bf16*)out, ld_tok, L, nh, perm);
    else MH_LAUNCH((head_transpose_kernel<128>), grid, dim3(256), 0, s, (const bf16*)in, (bf16*)out, ld_tok, L, nh, perm);
    MH_CHECK_LAUNCH();
    return MH_OK;
  }
  MH_CHECK_ARG(mode != 4, "head_permute: mode 4 needs bf16, seq_len %% 64 == 0, head dim 32 / 64 / 128 and 16-byte aligned rows");
  const int grid = tgrid((int64_t)B * L * nh * dh);
  MH_DTYPE_SWITCH(dtype,
                  MH_LAUNCH((head_permute_kernel<bf16>), dim3(grid), dim3(TB), 0, s, (const bf16*)in, (bf16*)out, ld_tok, B, L, nh, dh, mode),
                  MH_LAUNCH((head_permute_kernel<float>), dim3(grid), dim3(TB), 0, s, (const float*)in, (float*)out, ld_tok, B, L, nh, dh, mode),
                  "head_permute");
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_col_sum(const void* in, int64_t ld, int64_t rows, int cols, int batch, int64_t stride_in, float* partial,
                          int n_partial, float* out, int accumulate, int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(in && partial && out && rows > 0 && cols > 0 && n_partial > 0 && n_partial <= 1024 && batch > 0 && batch <= 65535,
               "col_sum: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int vec = dtype == MH_BF16 ? 8 : 4;
  if (cols % vec == 0 && ld % vec == 0 && stride_in % vec == 0) {
    dim3 grid((cols / vec + 63) / 64, n_partial, batch);
    MH_DTYPE_SWITCH(dtype, MH_LAUNCH((colsum8_partial_kernel<bf16>), grid, dim3(256), 0, s, (const bf16*)in, ld, rows, cols, partial, stride_in),
                    MH_LAUNCH((colsum8_partial_kernel<float>), grid, dim3(256), 0, s, (const float*)in, ld, rows, cols, partial, stride_in), "col_sum");
  } else {
    dim3 grid((cols + 63) / 64, n_partial, batch);
    MH_DTYPE_SWITCH(dtype, MH_LAUNCH((colsum_partial_kernel<bf16>), grid, dim3(256), 0, s, (const bf16*)in, ld, rows, cols, partial, stride_in),
                    MH_LAUNCH((colsum_partial_kernel<float>), grid, dim3(256), 0, s, (const float*)in, ld, rows, cols, partial, stride_in), "col_sum");
  }
  MH_CHECK_LAUNCH();
  MH_LAUNCH(colsum_final_kernel, dim3((cols + 63) / 64, batch), dim3(1024), 0, s, partial, n_partial, cols, out, accumulate);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_add_pos_time(const void* x, int64_t ldx, const float* pos, const float* emb, void* out, int B, int L, int H,
                               int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(x && pos && emb && out && B > 0 && L > 0 && H > 0, "add_pos_time: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MH_BF16 && H % 8 == 0 && ldx % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(pos) & 15) == 0 && (reinterpret_cast<uintptr_t>(emb) & 15) == 0) {
    MH_LAUNCH(add_pos_time8_kernel, dim3(tgrid((int64_t)B * L * (H >> 3))), dim3(TB), 0, s, (const bf16*)x, ldx, pos, emb, (bf16*)out, B, L, H);
    MH_CHECK_LAUNCH();
    return MH_OK;
  }
  const int grid = tgrid((int64_t)B * L * H);
  MH_DTYPE_SWITCH(dtype, MH_LAUNCH((add_pos_time_kernel<bf16>), dim3(grid), dim3(TB), 0, s, (const bf16*)x, ldx, pos, emb, (bf16*)out, B, L, H),
                  MH_LAUNCH((add_pos_time_kernel<float>), dim3(grid), dim3(TB), 0, s, (const float*)x, ldx, pos, emb, (float*)out, B, L, H), "add_pos_time");
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_act_fwd(const void* x, void* y, int64_t n, int act, int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(x && y && n > 0, "act_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  MH_DTYPE_SWITCH(dtype, MH_LAUNCH((act_fwd_kernel<bf16>), dim3(tgrid(n)), dim3(TB), 0, s, (const bf16*)x, (bf16*)y, n, act),
                  MH_LAUNCH((act_fwd_kernel<float>), dim3(tgrid(n)), dim3(TB), 0, s, (const float*)x, (float*)y, n, act), "act_fwd");
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_act_bwd(const void* dy, const void* x, void* dx, int64_t n, int act, int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(dy && x && dx && n > 0, "act_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  MH_DTYPE_SWITCH(dtype, MH_LAUNCH((act_bwd_kernel<bf16>), dim3(tgrid(n)), dim3(TB), 0, s, (const bf16*)dy, (const bf16*)x, (bf16*)dx, n, act),
                  MH_LAUNCH((act_bwd_kernel<float>), dim3(tgrid(n)), dim3(TB), 0, s, (const float*)dy, (const float*)x, (float*)dx, n, act), "act_bwd");
  MH_CHECK_LAUNCH();
  return MH_OK;
}

int mh_drop_args(const mh_dropout* d, DropArgs* out);
extern "C" int mh_layernorm_bwd_drop(const void* x, const void* dy, const float* gamma, void* dx, void* dx_dropped, const mh_dropout* drop,
                                     float* partial, int n_partial, float* dgamma, float* dbeta, int accumulate, int64_t rows, int H,
                                     float eps, int dtype, mh_stream_t stream);
extern "C" int mh_layernorm_bwd_ex(const void* x, const void* dy, const float* gamma, void* dx, void* dx_dropped, int64_t ldm, int m_panel,
                                   int always, const mh_dropout* drop, float* partial, int n_partial, float* dgamma, float* dbeta, int accumulate,
                                   int64_t rows, int H, float eps, int dtype, mh_stream_t stream);
extern "C" int mh_layernorm_bwd(const void* x, const void* dy, const float* gamma, void* dx, float* partial, int n_partial,
                                float* dgamma, float* dbeta, int accumulate, int64_t rows, int H, float eps, int dtype,
                                mh_stream_t stream) {
  return mh_layernorm_bwd_drop(x, dy, gamma, dx, nullptr, nullptr, partial, n_partial, dgamma, dbeta, accumulate, rows, H, eps, dtype, stream);
}
// mh_layernorm_bwd that also writes dx_dropped = dx o keep / (1 - p) for the dropout site `drop` (element index = row * H + col, the
// dense sites' convention): what mh_dropout_fwd would make of dx in a second pass.  drop == null or p == 0: plain mh_layernorm_bwd.
extern "C" int mh_layernorm_bwd_drop(const void* x, const void* dy, const float* gamma, void* dx, void* dx_dropped, const mh_dropout* drop,
                                     float* partial, int n_partial, float* dgamma, float* dbeta, int accumulate, int64_t rows, int H,
                                     float eps, int dtype, mh_stream_t stream) {
  return mh_layernorm_bwd_ex(x, dy, gamma, dx, dx_dropped, H, 0, 0, drop, partial, n_partial, dgamma, dbeta, accumulate, rows, H, eps, dtype, stream);
}
// mh_layernorm_bwd_drop with the second output's layout chosen by the caller: row-major with pitch ldm, or (m_panel) K32 panels
// [H / 32][ldm rows][32]; always != 0: the second output is written even when the site drops nothing (then a copy of dx in that layout).
int mh_ln_bwd_rows(const void* x, const void* dy, const float* gamma, void* dx, void* dx_dropped, int64_t ldm, int m_panel, int always,
                   const mh_dropout* drop, float* partial, int n_partial, int64_t rows, int H, float eps, int dtype, mh_stream_t stream);
int mh_ln_bwd_fold(const float* partial, int n_partial, int H, float* dgamma, float* dbeta, int accumulate, mh_stream_t stream);
extern "C" int mh_layernorm_bwd_ex(const void* x, const void* dy, const float* gamma, void* dx, void* dx_dropped, int64_t ldm, int m_panel,
                                   int always, const mh_dropout* drop, float* partial, int n_partial, float* dgamma, float* dbeta, int accumulate,
                                   int64_t rows, int H, float eps, int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(dgamma && dbeta, "layernorm_bwd: null pointer");
  int rc = mh_ln_bwd_rows(x, dy, gamma, dx, dx_dropped, ldm, m_panel, always, drop, partial, n_partial, rows, H, eps, dtype, stream);
  if (rc) return rc;
  return mh_ln_bwd_fold(partial, n_partial, H, dgamma, dbeta, accumulate, stream);
}
// the two halves of mh_layernorm_bwd_ex (library-internal: csrc/train_layer.hip folds the partials on a side stream, under the next GEMM):
// the row kernel (dx, dx_dropped, per-block column partials) ...
int mh_ln_bwd_rows(const void* x, const void* dy, const float* gamma, void* dx, void* dx_dropped, int64_t ldm, int m_panel, int always,
                   const mh_dropout* drop, float* partial, int n_partial, int64_t rows, int H, float eps, int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(x && dy && gamma && dx && partial, "layernorm_bwd: null pointer");
  MH_CHECK_ARG(rows > 0 && H % 8 == 0 && H <= 2048 && n_partial > 0 && n_partial <= 1024, "layernorm_bwd: bad shape");
  MH_CHECK_ARG(!m_panel || (H % 32 == 0 && ldm >= rows), "layernorm_bwd: a panel output needs H %% 32 == 0 and ldm >= rows");
  hipStream_t s = (hipStream_t)stream;
  float* pg = partial;
  float* pb = partial + (int64_t)n_partial * H;
  DropArgs da;
  int rcd = mh_drop_args(drop, &da);
  if (rcd) return rcd;
  if (da.thr != 0 || da.mask || (always && dx_dropped)) {
    MH_CHECK_ARG(dx_dropped, "layernorm_bwd_drop: dropout needs the second output");
#define MH_LNBD(T, N) MH_LAUNCH((ln_bwd_kernel<T, N, true>), dim3(n_partial), dim3(256), 0, s, (const T*)x, (const T*)dy, gamma, (T*)dx, pg, pb, rows, H, eps, (T*)dx_dropped, da, ldm, m_panel)
    if (H <= 512) { MH_DTYPE_SWITCH(dtype, MH_LNBD(bf16, 1), MH_LNBD(float, 1), "layernorm_bwd_drop"); }
    else if (H <= 1024) { MH_DTYPE_SWITCH(dtype, MH_LNBD(bf16, 2), MH_LNBD(float, 2), "layernorm_bwd_drop"); }
    else { MH_DTYPE_SWITCH(dtype, MH_LNBD(bf16, 4), MH_LNBD(float, 4), "layernorm_bwd_drop"); }
#undef MH_LNBD
  } else {
#define MH_LNB(T, N) MH_LAUNCH((ln_bwd_kernel<T, N>), dim3(n_partial), dim3(256), 0, s, (const T*)x, (const T*)dy, gamma, (T*)dx, pg, pb, rows, H, eps)
    if (H <= 512) { MH_DTYPE_SWITCH(dtype, MH_LNB(bf16, 1), MH_LNB(float, 1), "layernorm_bwd"); }
    else if (H <= 1024) { MH_DTYPE_SWITCH(dtype, MH_LNB(bf16, 2), MH_LNB(float, 2), "layernorm_bwd"); }
    else { MH_DTYPE_SWITCH(dtype, MH_LNB(bf16, 4), MH_LNB(float, 4), "layernorm_bwd"); }
#undef MH_LNB
  }
  MH_CHECK_LAUNCH();
  return MH_OK;
}
// ... and the fold of the partials into dgamma / dbeta (fixed order)
int mh_ln_bwd_fold(const float* partial, int n_partial, int H, float* dgamma, float* dbeta, int accumulate, mh_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const float* pg = partial;
  const float* pb = partial + (int64_t)n_partial * H;
  // (a colsum_final block folds 64 columns: 4 partial-lanes x 64)
  if (dbeta == dgamma + H) {   // the two gradients side by side (as the partials are): one launch folds both
    MH_LAUNCH(colsum_final_kernel, dim3((H + 63) / 64, 2), dim3(1024), 0, s, pg, n_partial, H, dgamma, accumulate);
    MH_CHECK_LAUNCH();
    return MH_OK;
  }
  MH_LAUNCH(colsum_final_kernel, dim3((H + 63) / 64, 1), dim3(1024), 0, s, pg, n_partial, H, dgamma, accumulate);
  MH_CHECK_LAUNCH();
  MH_LAUNCH(colsum_final_kernel, dim3((H + 63) / 64, 1), dim3(1024), 0, s, pb, n_partial, H, dbeta, accumulate);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_softmax_rows(void* s_inout, int64_t rows, int L, int64_t ld, float scale, int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(s_inout && rows > 0 && L > 0, "softmax_rows: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((rows + 3) / 4));
  MH_DTYPE_SWITCH(dtype, MH_LAUNCH((softmax_rows_kernel<bf16>), grid, dim3(256), 0, s, (bf16*)s_inout, rows, L, ld, scale),
                  MH_LAUNCH((softmax_rows_kernel<float>), grid, dim3(256), 0, s, (float*)s_inout, rows, L, ld, scale), "softmax_rows");
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_softmax_bwd_rows(const void* p, void* dp_inout, int64_t rows, int L, int64_t ld, float scale, int dtype,
                                   mh_stream_t stream) {
  MH_CHECK_ARG(p && dp_inout && rows > 0 && L > 0, "softmax_bwd_rows: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((rows + 3) / 4));
  MH_DTYPE_SWITCH(dtype, MH_LAUNCH((softmax_bwd_rows_kernel<bf16>), grid, dim3(256), 0, s, (const bf16*)p, (bf16*)dp_inout, rows, L, ld, scale),
                  MH_LAUNCH((softmax_bwd_rows_kernel<float>), grid, dim3(256), 0, s, (const float*)p, (float*)dp_inout, rows, L, ld, scale), "softmax_bwd_rows");
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_cross_entropy_fwd(const float* logits, int64_t ld, const int32_t* target, float* loss, float* lse, int64_t n,
                                    int V, mh_stream_t stream) {
  MH_CHECK_ARG(logits && target && loss && lse && n > 0 && V > 0, "cross_entropy_fwd: bad arguments");
  MH_LAUNCH(ce_fwd_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, ld, target, loss, lse, n, V);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_cross_entropy_bwd(const float* logits, int64_t ld, const int32_t* target, const float* lse, const float* grad,
                                    void* dlogits, int64_t ldd, int64_t n, int V, int Vpad, int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(logits && target && lse && grad && dlogits && n > 0 && V > 0 && Vpad >= V && ldd >= Vpad, "cross_entropy_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((n + 3) / 4));
  MH_DTYPE_SWITCH(dtype, MH_LAUNCH((ce_bwd_kernel<bf16>), grid, dim3(256), 0, s, logits, ld, target, lse, grad, (bf16*)dlogits, ldd, n, V, Vpad),
                  MH_LAUNCH((ce_bwd_kernel<float>), grid, dim3(256), 0, s, logits, ld, target, lse, grad, (float*)dlogits, ldd, n, V, Vpad), "cross_entropy_bwd");
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_sqdiff_mean(const float* a, const float* b, float scale_a, float* out, int B, int64_t per_batch, mh_stream_t stream) {
  MH_CHECK_ARG(a && out && B > 0 && per_batch > 0, "sqdiff_mean: bad arguments");
  MH_LAUNCH(sqdiff_mean_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, a, b, scale_a, out, per_batch);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_sqdiff_bwd(const float* a, const float* b, float scale_a, const float* grad, float* da, float* db, int accumulate,
                             int B, int64_t per_batch, mh_stream_t stream) {
  MH_CHECK_ARG(a && grad && (da || db) && B > 0 && per_batch > 0, "sqdiff_bwd: bad arguments");
  MH_LAUNCH(sqdiff_bwd_kernel, dim3(tgrid((int64_t)B * per_batch)), dim3(TB), 0, (hipStream_t)stream, a, b, scale_a, grad, da, db, accumulate, B, per_batch);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_add_inplace(void* dst, const void* src, int64_t n, int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(dst && src && n > 0, "add_inplace: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  MH_DTYPE_SWITCH(dtype, MH_LAUNCH((add_inplace_kernel<bf16>), dim3(tgrid(n)), dim3(TB), 0, s, (bf16*)dst, (const bf16*)src, n),
                  MH_LAUNCH((add_inplace_kernel<float>), dim3(tgrid(n)), dim3(TB), 0, s, (float*)dst, (const float*)src, n), "add_inplace");
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" size_t mh_scatter_add_rows_workspace_bytes(int E, int V) { return (size_t)32 * V * E * sizeof(float); }

extern "C" int mh_scatter_add_rows(const float* src, const int32_t* ids, float* table, int64_t n, int E, int V, void* workspace,
                                   size_t workspace_bytes, mh_stream_t stream) {
  MH_CHECK_ARG(src && ids && table && workspace && n > 0 && E > 0 && V > 0, "scatter_add_rows: bad arguments");
  MH_CHECK_ARG(workspace_bytes >= mh_scatter_add_rows_workspace_bytes(E, V), "scatter_add_rows: workspace too small");
  const int nchunks = n >= 32 * 256 ? 32 : (int)((n + 255) / 256);
  const int64_t chunk = ((n + nchunks - 1) / nchunks + 63) / 64 * 64;
  hipStream_t s = (hipStream_t)stream;
  MH_LAUNCH(scatter_rows_partial_kernel, dim3((V + SCATTER_RV - 1) / SCATTER_RV, nchunks), dim3(256), 4 * SCATTER_RV * 256 * sizeof(float), s, src, ids,
            (float*)workspace, n, E, V, chunk);
  MH_CHECK_LAUNCH();
  MH_LAUNCH(scatter_rows_final_kernel, dim3(tgrid((int64_t)V * E)), dim3(TB), 0, s, (const float*)workspace, table, nchunks, (int64_t)V * E);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_scale_rows(const float* src, const float* scale, const int32_t* mask, float* dst, int accumulate, int B,
                             int64_t per_batch, int E, mh_stream_t stream) {
  MH_CHECK_ARG(src && dst && B > 0 && per_batch > 0 && E > 0, "scale_rows: bad arguments");
  MH_LAUNCH(scale_rows_kernel, dim3(tgrid((int64_t)B * per_batch)), dim3(TB), 0, (hipStream_t)stream, src, scale, mask, dst,
            accumulate, B, per_batch, E);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// out[i] = sum over s of in[s * n + i]   (fp32; the split-K partials of a weight-gradient GEMM), fixed order
__global__ void sum_slices_kernel(const float* __restrict__ in, int slices, int64_t n, float* __restrict__ out) {
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
    f32x4 acc = *reinterpret_cast<const f32x4*>(in + i);
    for (int s = 1; s < slices; ++s) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(in + (int64_t)s * n + i);
      acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    *reinterpret_cast<f32x4*>(out + i) = acc;
  }
}

extern "C" int mh_sum_slices(const float* in, int slices, int64_t n, float* out, mh_stream_t stream) {
  MH_CHECK_ARG(in && out && slices > 0 && n > 0 && n % 4 == 0, "sum_slices: bad arguments (n must be a multiple of 4)");
  mh_prof_note("slices=%d n=%lld", slices, (long long)n);
  MH_LAUNCH(sum_slices_kernel, dim3(tgrid(n / 4)), dim3(TB), 0, (hipStream_t)stream, in, slices, n, out);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// ------------------------------------------------------------------ bf16 row-major <-> K32 panels (the encoder stack's entry)
namespace {
__global__ void repack_panel_kernel(const bf16* __restrict__ in, int64_t ld_in, bf16* __restrict__ out, int64_t ld_out, int64_t rows, int cols,
                                    int to_panel) {
  const int cpr = cols >> 3;   // 16-byte chunks per row
  const int64_t total = rows * cpr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    // four consecutive threads move one 64-byte panel row, the next four the next token's: contiguous on the panel side, 64-byte row
    // segments on the row-major side
    const int64_t per_panel = rows * 4;
    const int cp = (int)(i / per_panel);
    const int64_t rem = i - (int64_t)cp * per_panel;
    const int64_t r = rem >> 2;
    const int c = cp * 4 + (int)(rem & 3);
    const int64_t pan = ((int64_t)(c >> 2) * (to_panel ? ld_out : ld_in) + r) * 32 + (c & 3) * 8;
    const int64_t rm = r * (to_panel ? ld_in : ld_out) + c * 8;
    *reinterpret_cast<f32x4*>(out + (to_panel ? pan : rm)) = *reinterpret_cast<const f32x4*>(in + (to_panel ? rm : pan));
  }
}
}  // namespace

// bf16 [rows, cols] row-major (pitch ld) <-> K32 panels [cols / 32][ld rows][32]; cols % 32 == 0.  to_panel != 0: in row-major -> out panels.
extern "C" int mh_repack_panel(const void* in, int64_t ld_in, void* out, int64_t ld_out, int64_t rows, int cols, int to_panel, mh_stream_t stream) {
  MH_CHECK_ARG(in && out && rows > 0 && cols > 0 && cols % 32 == 0, "repack_panel: bad arguments (cols must be a multiple of 32)");
  MH_CHECK_ARG(to_panel ? (ld_in % 8 == 0 && ld_in >= cols && ld_out >= rows) : (ld_out % 8 == 0 && ld_out >= cols && ld_in >= rows), "repack_panel: bad leading dimensions");
  MH_LAUNCH(repack_panel_kernel, dim3(tgrid(rows * (cols >> 3))), dim3(TB), 0, (hipStream_t)stream, (const bf16*)in, ld_in, (bf16*)out, ld_out, rows, cols, to_panel);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// ------------------------------------------------------------------ bf16 working copies of the master weights, one launch
namespace {
__global__ __launch_bounds__(256) void weight_prep_kernel(const mh_wprep_item* __restrict__ items, int n_items) {
  __shared__ bf16 tile[64][66];
  int it = 0;
  for (int i = 1; i < n_items; ++i) if ((int)blockIdx.x >= items[i].tile_start) it = i;     // (block-uniform; tables are short)
  const mh_wprep_item w = items[it];
  const int local = (int)blockIdx.x - w.tile_start, tc = w.cols >> 6;
  const int r0 = (local / tc) * 64, c0 = (local % tc) * 64;
  const int t = threadIdx.x, tr = t >> 4, c4 = (t & 15) * 4;
  bf16* dst = reinterpret_cast<bf16*>(w.dst);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = tr + 16 * r;
    const f32x4 v = *reinterpret_cast<const f32x4*>(w.src + (int64_t)(r0 + row) * w.cols + c0 + c4);
    bf16x4 b;
#pragma unroll
    for (int e = 0; e < 4; ++e) { b[e] = (bf16)v[e]; tile[row][c4 + e] = b[e]; }
    // pad_ bit 0: dst as K32 panels [cols / 32][ld_dst rows][32]
    if (w.pad_ & 1) *reinterpret_cast<bf16x4*>(dst + ((int64_t)((c0 + c4) >> 5) * w.ld_dst + r0 + row) * 32 + ((c0 + c4) & 31)) = b;
    else *reinterpret_cast<bf16x4*>(dst + (int64_t)(r0 + row) * w.ld_dst + c0 + c4) = b;
  }
  if (!w.dst_t) return;
  __syncthreads();
  bf16* dstT = reinterpret_cast<bf16*>(w.dst_t);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int col = tr + 16 * r;                      // column of the source tile = row of the transposed one
    bf16x4 b;
#pragma unroll
    for (int e = 0; e < 4; ++e) b[e] = tile[c4 + e][col];
    // pad_ bit 1: dst_t (the [cols][rows] transpose) as K32 panels [rows / 32][ld_t rows][32]
    if (w.pad_ & 2) *reinterpret_cast<bf16x4*>(dstT + ((int64_t)((r0 + c4) >> 5) * w.ld_t + c0 + col) * 32 + ((r0 + c4) & 31)) = b;
    else *reinterpret_cast<bf16x4*>(dstT + (int64_t)(c0 + col) * w.ld_t + r0 + c4) = b;
  }
}
}  // namespace

extern "C" int mh_weight_prep(const mh_wprep_item* items, int n_items, int total_tiles, mh_stream_t stream) {
  MH_CHECK_ARG(items && n_items > 0 && total_tiles > 0, "weight_prep: empty table");
  MH_LAUNCH(weight_prep_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, items, n_items);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_adamw_ema_step(const mh_opt_tensor* tensors, const mh_opt_chunk* chunks, int n_chunks,
                                 const mh_opt_hparams* hp, mh_stream_t stream) {
  MH_CHECK_ARG(tensors && chunks && hp && n_chunks > 0, "adamw_ema_step: bad arguments");
  MH_CHECK_ARG(hp->n_ema >= 0 && hp->n_ema <= 4, "adamw_ema_step: at most 4 EMA copies");
  mh_prof_note("chunks=%d n_ema=%d", n_chunks, hp->n_ema);
  MH_LAUNCH(adamw_ema_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, tensors, chunks, *hp);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_clip_grads(const mh_opt_tensor* tensors, const mh_opt_chunk* chunks, int n_chunks, const float* norm, float max_norm,
                             mh_stream_t stream) {
  MH_CHECK_ARG(tensors && chunks && norm && n_chunks > 0 && max_norm > 0.f, "clip_grads: bad arguments");
  MH_LAUNCH(clip_grads_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, tensors, chunks, norm, max_norm);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_grad_norm(const mh_opt_tensor* tensors, const mh_opt_chunk* chunks, int n_chunks, float* partial,
                            float* out, mh_stream_t stream) {
  MH_CHECK_ARG(tensors && chunks && partial && out && n_chunks > 0, "grad_norm: bad arguments");
  MH_LAUNCH(sumsq_chunks_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, tensors, chunks, partial);
  MH_CHECK_LAUNCH();
  MH_LAUNCH(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n_chunks, out);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
