// LayerNorm kernels (HBM-bound): one 64-lane wavefront per row, 16-B vector loads, the row is
// held in registers, mean / variance by two in-register passes and xor-shuffle wave reductions
// (no LDS).  eps 1e-12 and the biased variance follow torch.nn.LayerNorm as used at
// models/network.py:79,:149 and in HF BertSelfOutput / BertOutput.
#include "common.h"

namespace {

constexpr int MAXCH = 4;  // 8-element chunks per lane: H <= 64*8*4 = 2048
MH_KNOB(int, g_ln_rows4, 1);       // panel LayerNorm: 4 rows per wave (default) or 16 (A/B: mh_layernorm_set_rows4)

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

// ---- K32-panel layout ([H/32][ld rows][32], bf16): one wave = 16 rows; lane (r = lane>>2, c = lane&3) owns
// 8 elements of every panel, so each wave-instruction moves 16 rows x 64 B = 1 KiB of contiguous memory.
template <int NKB, bool ADD>
__global__ __launch_bounds__(256) void ln_panel_kernel(const bf16* __restrict__ x, int64_t ldx, const float* __restrict__ xf32,
                                                       const float* __restrict__ pos, const float* __restrict__ emb_t,
                                                       const int32_t* __restrict__ emb_row, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, bf16* __restrict__ out, int64_t ldo,
                                                       int64_t rows, int L, float eps) {
  constexpr int H = NKB * 32;
  const int lane = threadIdx.x & 63, r = lane >> 2, c = lane & 3;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
  if (row0 >= rows) return;
  int64_t row = row0 + r;
  const bool valid = row < rows;
  if (!valid) row = rows - 1;
  float v[NKB][8];
  float sum = 0.f;
  const float* prow = nullptr;
  const float* trow = nullptr;
  if constexpr (ADD) {
    const int64_t b = row / L, l = row % L;
    prow = pos + l * H;
    trow = emb_t + (int64_t)(emb_row ? emb_row[b] : (int)b) * H;
  }
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    if constexpr (ADD) {
      if (xf32) load8(xf32 + row * ldx + kb * 32 + c * 8, v[kb]);
      else load8(x + ((int64_t)kb * ldx + row) * 32 + c * 8, v[kb]);
      float p[8], t[8];
      load8(prow + kb * 32 + c * 8, p);
      load8(trow + kb * 32 + c * 8, t);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[kb][e] = (p[e] + v[kb][e]) + t[e];
    } else {
      load8(x + ((int64_t)kb * ldx + row) * 32 + c * 8, v[kb]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += v[kb][e];
  }
  sum += __shfl_xor(sum, 1, 64);
  sum += __shfl_xor(sum, 2, 64);
  const float mean = sum / (float)H;
  float sq = 0.f;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float d = v[kb][e] - mean;
      sq += d * d;
    }
  sq += __shfl_xor(sq, 1, 64);
  sq += __shfl_xor(sq, 2, 64);
  const float rstd = 1.0f / sqrtf(sq / (float)H + eps);
  if (!valid) return;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    float g[8], bt[8], y[8];
    load8(gamma + kb * 32 + c * 8, g);
    load8(beta + kb * 32 + c * 8, bt);
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = (v[kb][e] - mean) * rstd * g[e] + bt[e];
    store8(out + ((int64_t)kb * ldo + row) * 32 + c * 8, y);
  }
}

// The same with 4 rows per wave (lane = (row r = lane >> 4, c = lane & 15); c covers chunk c & 3 of panel 4 j + (c >> 2) in step j): 16 rows
// per 256-thread block instead of 64, i.e. four times the waves for the same tensor - the 16-row form leaves a CU 4 waves with 128
// registers each at the sampler's half batch (256 blocks for 16384 rows: latency-bound, 1.8 TB/s) - and 32 value registers per lane
// instead of 128.  A wave-instruction still moves whole lines: 4 rows x 64 B = 256 contiguous bytes in each of 4 panels.
template <int NKB, bool ADD>
__global__ __launch_bounds__(256) void ln_panel4_kernel(const bf16* __restrict__ x, int64_t ldx, const float* __restrict__ xf32,
                                                        const float* __restrict__ pos, const float* __restrict__ emb_t,
                                                        const int32_t* __restrict__ emb_row, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, bf16* __restrict__ out, int64_t ldo,
                                                        int64_t rows, int L, float eps) {
  static_assert(NKB % 4 == 0, "four panels per step");
  constexpr int H = NKB * 32, NJ = NKB / 4;
  const int lane = threadIdx.x & 63, r = lane >> 4, c = lane & 15;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
  if (row0 >= rows) return;
  int64_t row = row0 + r;
  const bool valid = row < rows;
  if (!valid) row = rows - 1;
  const int col0 = (c >> 2) * 32 + (c & 3) * 8;     // first column of this lane's chunk in step 0 (step j: + 128 j)
  float v[NJ][8];
  float sum = 0.f;
  const float* prow = nullptr;
  const float* trow = nullptr;
  if constexpr (ADD) {
    const int64_t b = row / L, l = row % L;
    prow = pos + l * H;
    trow = emb_t + (int64_t)(emb_row ? emb_row[b] : (int)b) * H;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = col0 + 128 * j;
    if constexpr (ADD) {
      if (xf32) load8(xf32 + row * ldx + col, v[j]);
      else load8(x + ((int64_t)(col >> 5) * ldx + row) * 32 + (col & 31), v[j]);
      float p[8], t[8];
      load8(prow + col, p);
      load8(trow + col, t);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[j][e] = (p[e] + v[j][e]) + t[e];      // same association as the reference: (pos + x) + t
    } else {
      load8(x + ((int64_t)(col >> 5) * ldx + row) * 32 + (col & 31), v[j]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += v[j][e];
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum / (float)H;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float d = v[j][e] - mean;
      sq += d * d;
    }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) sq += __shfl_xor(sq, o, 64);
  const float rstd = 1.0f / sqrtf(sq / (float)H + eps);
  if (!valid) return;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = col0 + 128 * j;
    float g[8], bt[8], y[8];
    load8(gamma + col, g);
    load8(beta + col, bt);
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = (v[j][e] - mean) * rstd * g[e] + bt[e];
    store8(out + ((int64_t)(col >> 5) * ldo + row) * 32 + (col & 31), y);
  }
}

template <bool ADD>
int launch_ln_panel(const void* x, int64_t ldx, const float* xf32, const float* pos, const float* emb_t,
                    const int32_t* emb_row, const float* gamma, const float* beta, void* out, int64_t ldo, int64_t rows,
                    int L, int H, float eps, hipStream_t s) {
  if (g_ln_rows4 && (H / 32) % 4 == 0) {
    dim3 grid4((unsigned)((rows + 15) / 16)), block4(256);
#define MH_LNP4(N)                                                                                                      \
  case N:                                                                                                              \
    MH_LAUNCH((ln_panel4_kernel<N, ADD>), grid4, block4, 0, s, (const bf16*)x, ldx, xf32, pos, emb_t, emb_row, gamma, beta, \
              (bf16*)out, ldo, rows, L, eps);                                                                          \
    break;
    switch (H / 32) {
      MH_LNP4(4) MH_LNP4(8) MH_LNP4(12) MH_LNP4(16) MH_LNP4(24)
      default:
        mh_set_error("layernorm(panel): hidden size %d not in {128,256,384,512,768}", H);
        return MH_ERR_UNSUPPORTED;
    }
#undef MH_LNP4
    MH_CHECK_LAUNCH();
    return MH_OK;
  }
  dim3 grid((unsigned)((rows + 63) / 64)), block(256);
#define MH_LNP(N)                                                                                                      \
  case N:                                                                                                              \
    MH_LAUNCH((ln_panel_kernel<N, ADD>), grid, block, 0, s, (const bf16*)x, ldx, xf32, pos, emb_t, emb_row, gamma, beta, \
              (bf16*)out, ldo, rows, L, eps);                                                                          \
    break;
  switch (H / 32) {
    MH_LNP(2) MH_LNP(4) MH_LNP(8) MH_LNP(12) MH_LNP(16) MH_LNP(24)
    default:
      mh_set_error("layernorm(panel): hidden size %d not in {64,128,256,384,512,768}", H);
      return MH_ERR_UNSUPPORTED;
  }
#undef MH_LNP
  MH_CHECK_LAUNCH();
  return MH_OK;
}

}  // namespace

#ifdef MH_ABLATE
extern "C" int mh_layernorm_set_rows4(int on) {
  g_ln_rows4 = on != 0;
  return MH_OK;
}
#endif

extern "C" int mh_layernorm_panel(const void* x, int64_t ldx, const float* gamma, const float* beta, void* out,
                                  int64_t ldo, int64_t rows, int H, float eps, mh_stream_t stream) {
  MH_CHECK_ARG(x && gamma && beta && out && rows > 0, "layernorm_panel: bad arguments");
  return launch_ln_panel<false>(x, ldx, nullptr, nullptr, nullptr, nullptr, gamma, beta, out, ldo, rows, 1, H, eps,
                                (hipStream_t)stream);
}

extern "C" int mh_add_pos_time_layernorm_panel(const void* x, int64_t ldx, int x_is_f32, const float* pos,
                                               const float* emb_t, const int32_t* emb_row, const float* gamma,
                                               const float* beta, void* out, int64_t ldo, int B, int L, int H, float eps,
                                               mh_stream_t stream) {
  MH_CHECK_ARG(x && pos && emb_t && gamma && beta && out && B > 0 && L > 0, "add_pos_time_layernorm_panel: bad arguments");
  return launch_ln_panel<true>(x_is_f32 ? nullptr : x, ldx, x_is_f32 ? (const float*)x : nullptr, pos, emb_t, emb_row,
                               gamma, beta, out, ldo, (int64_t)B * L, L, H, eps, (hipStream_t)stream);
}

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
