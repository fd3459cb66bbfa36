// LayerNorm kernels (HBM-bound): one 64-lane wavefront per row, 16-B vector loads, the row is
// held in registers, mean / variance by two in-register passes and xor-shuffle wave reductions
// (no LDS).  eps 1e-12 and the biased variance follow torch.nn.LayerNorm as used at
// models/network.py:79,:149 and in HF BertSelfOutput / BertOutput.
#include "common.h"

namespace {

constexpr int MAXCH = 4;  // 8-element chunks per lane: H <= 64*8*4 = 2048

template <typename T, bool ADD>
__global__ __launch_bounds__(256) void ln_kernel(const T* __restrict__ x, int64_t ldx, const float* __restrict__ xf32,
                                                 const float* __restrict__ pos, const float* __restrict__ emb_t,
                                                 const int32_t* __restrict__ emb_row, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, T* __restrict__ out, int64_t rows,
                                                 int L, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = H >> 3;
  float v[MAXCH][8];
  float sum = 0.f;
  const float* prow = nullptr;
  const float* trow = nullptr;
  if constexpr (ADD) {
    const int64_t b = row / L, l = row % L;
    prow = pos + l * H;
    trow = emb_t + (int64_t)(emb_row ? emb_row[b] : (int)b) * H;
  }
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      if constexpr (ADD) {
        if (xf32) load8(xf32 + row * ldx + c * 8, v[i]);
        else load8(x + row * ldx + c * 8, v[i]);
        float p[8], t[8];
        load8(prow + c * 8, p);
        load8(trow + c * 8, t);
        // same association as the reference: (pos + x) + t   (network.py:148)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = (p[e] + v[i][e]) + t[e];
      } else {
        load8(x + row * ldx + c * 8, v[i]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += v[i][e];
    }
  }
  const float mean = wave_sum(sum) / (float)H;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[i][e] - mean;
        sq += d * d;
      }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      float g[8], bt[8], y[8];
      load8(gamma + c * 8, g);
      load8(beta + c * 8, bt);
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = (v[i][e] - mean) * rstd * g[e] + bt[e];
      store8(out + row * H + c * 8, y);
    }
  }
}

template <typename T>
int launch_ln(const void* x, int64_t ldx, const float* xf32, const float* pos, const float* emb_t,
              const int32_t* emb_row, const float* gamma, const float* beta, void* out, int64_t rows, int L, int H,
              float eps, bool add, hipStream_t s) {
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (add)
    MH_LAUNCH((ln_kernel<T, true>), grid, block, 0, s, (const T*)x, ldx, xf32, pos, emb_t, emb_row, gamma,
                       beta, (T*)out, rows, L, H, eps);
  else
    MH_LAUNCH((ln_kernel<T, false>), grid, block, 0, s, (const T*)x, ldx, xf32, pos, emb_t, emb_row, gamma,
                       beta, (T*)out, rows, L, H, eps);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

}  // namespace

extern "C" int mh_layernorm(const void* x, const float* gamma, const float* beta, void* out, int64_t rows, int H,
                            float eps, int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(x && gamma && beta && out, "layernorm: null pointer");
  MH_CHECK_ARG(rows > 0 && H > 0 && H % 8 == 0 && H <= 2048, "layernorm: H=%d must be a multiple of 8, <= 2048", H);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MH_BF16) return launch_ln<bf16>(x, H, nullptr, nullptr, nullptr, nullptr, gamma, beta, out, rows, 1, H, eps, false, s);
  if (dtype == MH_F32) return launch_ln<float>(x, H, nullptr, nullptr, nullptr, nullptr, gamma, beta, out, rows, 1, H, eps, false, s);
  mh_set_error("layernorm: unknown dtype %d", dtype);
  return MH_ERR_INVALID;
}

extern "C" int mh_add_pos_time_layernorm(const void* x, int64_t ldx, int x_is_f32, const float* pos,
                                         const float* emb_t, const int32_t* emb_row, const float* gamma,
                                         const float* beta, void* out, int B, int L, int H, float eps, int dtype,
                                         mh_stream_t stream) {
  MH_CHECK_ARG(x && pos && emb_t && gamma && beta && out, "add_pos_time_layernorm: null pointer");
  MH_CHECK_ARG(B > 0 && L > 0 && H % 8 == 0 && H <= 2048, "add_pos_time_layernorm: H=%d must be a multiple of 8, <= 2048", H);
  MH_CHECK_ARG(ldx % 8 == 0, "add_pos_time_layernorm: ldx must be a multiple of 8");
  hipStream_t s = (hipStream_t)stream;
  const int64_t rows = (int64_t)B * L;
  const float* xf = x_is_f32 ? (const float*)x : nullptr;
  if (dtype == MH_BF16) return launch_ln<bf16>(x_is_f32 ? nullptr : x, ldx, xf, pos, emb_t, emb_row, gamma, beta, out, rows, L, H, eps, true, s);
  if (dtype == MH_F32) return launch_ln<float>(x_is_f32 ? nullptr : x, ldx, xf, pos, emb_t, emb_row, gamma, beta, out, rows, L, H, eps, true, s);
  mh_set_error("add_pos_time_layernorm: unknown dtype %d", dtype);
  return MH_ERR_INVALID;
}
