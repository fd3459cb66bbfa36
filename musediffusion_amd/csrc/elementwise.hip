// HBM-bound streaming kernels: packing / casts, embedding gather, timestep embedding, the fp32
// diffusion arithmetic (q_sample, fused p_sample / ddim tails), the counter-based truncated-normal
// generator and the device-side loop state of a captured reverse step.
//
// The diffusion arithmetic is written to be BIT-EXACT against the reference's fp32 torch ops given
// identical inputs: every product and sum is rounded separately (FP contraction is switched off for
// this file; torch never fuses a*b+c on CPU) and all per-step scalars arrive pre-computed by the
// host with the same fp32 torch expressions the reference evaluates (models/diffusion.py:904-917).
#include "common.h"
#include "step_update.h"

#pragma clang fp contract(off)

namespace {

constexpr int EW_BLOCK = 256;
inline int ew_grid(int64_t n_items) {
  int64_t b = (n_items + EW_BLOCK - 1) / EW_BLOCK;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));  // grid-stride beyond 8192 blocks (32 per CU)
}

// ---------------------------------------------------------------- packing
template <typename T>
__global__ void cast_pad_kernel(const float* __restrict__ in, int64_t ld_in, T* __restrict__ out, int64_t ld_out,
                                int64_t rows, int64_t cols, int64_t rows_out) {
  const int64_t total = rows_out * ld_out;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / ld_out, c = i % ld_out;
    const float v = (r < rows && c < cols) ? in[r * ld_in + c] : 0.f;
    out[i] = from_f32<T>(v);
  }
}

template <typename T>
__global__ void cast_to_f32_kernel(const T* __restrict__ in, int64_t ld_in, float* __restrict__ out, int64_t ld_out,
                                   int64_t rows, int64_t cols) {
  const int64_t total = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols, c = i % cols;
    out[r * ld_out + c] = to_f32(in[r * ld_in + c]);
  }
}

// row-major fp32 [rows, cols] -> K32-panel bf16 [cols_pad/32][ld rows][32] (zero fill outside rows x cols)
__global__ void pack_panel_kernel(const float* __restrict__ in, int64_t ld_in, bf16* __restrict__ out, int64_t ld_rows,
                                  int64_t rows, int64_t cols, int64_t cols_pad) {
  const int64_t total = (cols_pad / 32) * ld_rows * 32;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t kb = i / (ld_rows * 32), rem = i % (ld_rows * 32), r = rem / 32, c = kb * 32 + rem % 32;
    out[i] = (bf16)((r < rows && c < cols) ? in[r * ld_in + c] : 0.f);
  }
}
__global__ void unpack_panel_kernel(const bf16* __restrict__ in, int64_t ld_rows, float* __restrict__ out, int64_t ld_out,
                                    int64_t rows, int64_t cols) {
  const int64_t total = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols, c = i % cols;
    out[r * ld_out + c] = (float)in[((c >> 5) * ld_rows + r) * 32 + (c & 31)];
  }
}

__global__ void row_sqnorm_kernel(const float* __restrict__ table, float* __restrict__ out, int V, int E) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= V) return;
  // sequential-in-lane then wave tree; rounding differs from torch's pairwise sum by O(1e-7) relative
  float s = 0.f;
  for (int c = lane; c < E; c += 64) {
    const float v = table[(int64_t)row * E + c];
    s += v * v;
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}

__global__ void embed_gather_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids,
                                    float* __restrict__ out, int64_t n_tokens, int E, int V) {
  const int64_t total = n_tokens * E;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / E;
    const int e = (int)(i % E);
    int id = ids[n];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    out[i] = table[(int64_t)id * E + e];
  }
}

template <typename T>
__global__ void timestep_embedding_kernel(const float* __restrict__ t, T* __restrict__ out, int B, int dim,
                                          int64_t ld_out, float neg_log_period) {
  const int half = dim / 2;
  const int64_t total = (int64_t)B * ld_out;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / ld_out), c = (int)(i % ld_out);
    float v = 0.f;
    if (c < 2 * half) {
      const int k = c < half ? c : c - half;
      // freqs = exp(-ln(max_period) * k / half), args = t * freqs   (network.py:119-125)
      const float f = expf(neg_log_period * (float)k / (float)half);
      const float a = t[b] * f;
      v = c < half ? cosf(a) : sinf(a);
    }
    out[i] = from_f32<T>(v);
  }
}

// ---------------------------------------------------------------- diffusion arithmetic
__device__ __forceinline__ bool anchored(const int32_t* mask, int mask_per_elem, int64_t i, int E) {
  if (!mask) return false;
  return (mask_per_elem ? mask[i] : mask[i / E]) == 0;
}

__global__ void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                const float* __restrict__ a, const float* __restrict__ s,
                                const int32_t* __restrict__ mask, int mask_per_elem, float* __restrict__ out, int B,
                                int64_t per_batch, int E) {
  const int64_t total = (int64_t)B * per_batch;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / per_batch);
    const float x = x0[i];
    const float xt = a[b] * x + s[b] * noise[i];
    out[i] = anchored(mask, mask_per_elem, i, E) ? x : xt;
  }
}

template <bool DDIM>
__global__ void step_epilogue_kernel(const float* __restrict__ model_out, const float* __restrict__ x_t,
                                     const float* __restrict__ noise, const int32_t* __restrict__ round_idx,
                                     const float* __restrict__ table, const mh_step_coef* __restrict__ coef,
                                     int coef_per_batch, int clip, const int32_t* __restrict__ mask, int mask_per_elem,
                                     const float* __restrict__ x_start, float* __restrict__ out,
                                     float* __restrict__ pred_xstart, float* __restrict__ mean_out, int B,
                                     int64_t per_batch, int E) {
  const int64_t total = (int64_t)B * per_batch;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / per_batch);
    const mh_step_coef c = coef[coef_per_batch ? b : 0];
    float x0;
    if (round_idx) x0 = table[(int64_t)round_idx[i / E] * E + (i % E)];
    else x0 = model_out[i];
    if (clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
    const float xt = x_t[i];
    const float nz = noise ? noise[i] : 0.f;
    float mean, sample;
    if constexpr (DDIM) {
      const float eps = (c.recip * xt - x0) / c.recipm1;
      mean = x0 * c.sqrt_abp + c.dir * eps;
      sample = mean + c.sigma * nz;
    } else {
      mean = c.coef1 * x0 + c.coef2 * xt;
      sample = mean + c.sigma * nz;
    }
    if (anchored(mask, mask_per_elem, i, E)) sample = x_start[i];
    if (pred_xstart) pred_xstart[i] = x0;
    if (mean_out) mean_out[i] = mean;
    out[i] = sample;
  }
}

// The same arithmetic on 4 consecutive elements per thread (16-byte loads / stores, 32-bit index math, one coefficient / index / mask
// lookup per group): E % 4 == 0 and 16-byte aligned tensors - the shapes of the sampling loops.  The per-element statements are the
// scalar kernel's, so the results are bit-identical.
// SLOTS: the nearest-embedding index of a row is still spread over `nslots` partial (score, index) pairs (the rounding GEMM's
// per-column-slot winners, mh_round_scores): the group folds them itself - same rule as argbest_reduce_kernel: the larger score, on a
// tie the smaller index - so the separate reduce launch and its [rows] index round trip go; the thread of a row's first group also
// writes the index to round_idx_out (may be NULL).
// RNG (with SLOTS): the group draws its own four truncated normals - the values mh_trunc_normal_at writes for these elements
// (same Philox counters: global group number, loop step, stream) - instead of reading `noise`: the generator launch at the head of
// a step and the noise tensor's write + read are gone, and the batch slices of a step start level.
template <bool DDIM, bool SLOTS = false, bool RNG = false>
__global__ __launch_bounds__(256) void step_epilogue4_kernel(const float* __restrict__ model_out, const float* __restrict__ x_t,
                                                             const float* __restrict__ noise, const int32_t* __restrict__ round_idx,
                                                             const float* __restrict__ table, const mh_step_coef* __restrict__ coef,
                                                             int coef_per_batch, int clip, const int32_t* __restrict__ mask, int mask_per_elem,
                                                             const float* __restrict__ x_start, float* __restrict__ out,
                                                             float* __restrict__ pred_xstart, float* __restrict__ mean_out, int64_t ngroups,
                                                             uint32_t groups_per_batch, uint32_t groups_per_row, int E,
                                                             const float* __restrict__ pbest = nullptr, const int32_t* __restrict__ pidx = nullptr,
                                                             int nslots = 0, int32_t* __restrict__ round_idx_out = nullptr, const StepRng rng = StepRng{}) {
  uint32_t rng_step = 0;
  if constexpr (RNG) rng_step = rng.step ? *rng.step : 0u;
  for (int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < ngroups; gi += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t g = (uint32_t)gi;                      // (ngroups < 2^32: checked by the launcher)
    const uint32_t b = g / groups_per_batch, row = g / groups_per_row, cg = g - row * groups_per_row;
    const mh_step_coef c = coef[coef_per_batch ? b : 0];
    const int64_t i = (int64_t)g * 4;
    f32x4 x0;
    if constexpr (SLOTS) {
      float best = -INFINITY;
      int bi = 0x7fffffff;
      const int64_t s0 = (int64_t)row * nslots;
      if (nslots == 0) bi = pidx[row];                    // (already folded: the index itself)
      for (int sl = 0; sl < nslots; ++sl) {               // (the lanes of a row read the same addresses: one broadcast line per step)
        const float v = pbest[s0 + sl];
        const int k = pidx[s0 + sl];
        if (v > best || (v == best && k < bi)) { best = v; bi = k; }
      }
      if (bi == 0x7fffffff) bi = 0;
      if (round_idx_out && cg == 0) round_idx_out[row] = bi;
      x0 = *reinterpret_cast<const f32x4*>(table + (int64_t)bi * E + cg * 4);
    } else
    if (round_idx) x0 = *reinterpret_cast<const f32x4*>(table + (int64_t)round_idx[row] * E + cg * 4);
    else x0 = *reinterpret_cast<const f32x4*>(model_out + i);
    const f32x4 xt = *reinterpret_cast<const f32x4*>(x_t + i);
    f32x4 nz = {0.f, 0.f, 0.f, 0.f};
    if constexpr (RNG) {
      float z[4];
      trunc_normal4(rng.first_group + g, rng_step, rng.bound, rng.seed_lo, rng.seed_hi, rng.stream_id, z);
      nz = f32x4{z[0], z[1], z[2], z[3]};
    } else
    if (noise) nz = *reinterpret_cast<const f32x4*>(noise + i);
    f32x4 mean, sample;
    step_update4<DDIM>(x0, xt, nz, c, clip, mean, sample);
    if (mask) {
      if (mask_per_elem) {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (mask[i + e] == 0) sample[e] = x_start[i + e];
      } else if (mask[row] == 0) sample = *reinterpret_cast<const f32x4*>(x_start + i);
    }
    if (pred_xstart) *reinterpret_cast<f32x4*>(pred_xstart + i) = x0;
    if (mean_out) *reinterpret_cast<f32x4*>(mean_out + i) = mean;
    *reinterpret_cast<f32x4*>(out + i) = sample;
  }
}

namespace {
// the 4-wide kernel's preconditions: whole groups per row, 16-byte aligned tensors, 32-bit group numbers
inline bool step_epilogue_vec_ok(const float* model_out, const float* x_t, const float* noise, const float* table, const float* x_start,
                                 const float* out, const float* pred, const float* mean, int B, int64_t per_batch, int E) {
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return E % 4 == 0 && per_batch % E == 0 && (int64_t)B * per_batch / 4 < (1ll << 32) && al(model_out) && al(x_t) && al(noise) && al(table) &&
         al(x_start) && al(out) && al(pred) && al(mean);
}
}  // namespace

__global__ __launch_bounds__(256) void trunc_normal_kernel(float* __restrict__ out, int64_t n, int64_t first, float bound, uint32_t seed_lo,
                                                          uint32_t seed_hi, uint32_t stream_id, const uint32_t* __restrict__ step_counter) {
  const uint32_t step = step_counter ? *step_counter : 0u;
  const int64_t ngroups = (n + 3) >> 2;
  for (int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < ngroups; gi += (int64_t)gridDim.x * blockDim.x) {
    float z[4];
    trunc_normal4((uint64_t)((first >> 2) + gi), step, bound, seed_lo, seed_hi, stream_id, z);
    const int64_t i = gi << 2;
    if (i + 4 <= n) *reinterpret_cast<f32x4*>(out + i) = f32x4{z[0], z[1], z[2], z[3]};
    else for (int k = 0; k < 4 && i + k < n; ++k) out[i + k] = z[k];
  }
}

// ---------------------------------------------------------------- captured-step loop state
__global__ void step_begin_kernel(mh_loop_state* state, const int32_t* __restrict__ steps,
                                  const mh_step_coef* __restrict__ coef_table, mh_step_coef* cur_coef,
                                  int32_t* emb_row, int B) {
  uint32_t pos = state->pos;
  if (pos >= state->n_steps) pos = state->n_steps - 1;
  const int32_t t = steps[pos];
  for (int b = threadIdx.x; b < B; b += blockDim.x) emb_row[b] = t;
  if (threadIdx.x == 0) {
    state->cur_t = t;
    *cur_coef = coef_table[t];
  }
}
__global__ void step_end_kernel(mh_loop_state* state) { state->pos += 1; }
// step_begin and step_end as ONE node at the head of a captured step (the two single-thread launches and the dependent kernel
// boundary between the step's last kernel and step_end come off the serial chain at every step boundary): the step in flight keeps
// its iteration number in state->rng_step (what the in-graph noise generator reads), state->pos already counts the next one
__global__ void step_advance_kernel(mh_loop_state* state, const int32_t* __restrict__ steps, const mh_step_coef* __restrict__ coef_table,
                                    mh_step_coef* cur_coef, int32_t* emb_row, int B) {
  const uint32_t raw = state->pos;
  const uint32_t pos = raw >= state->n_steps ? state->n_steps - 1 : raw;
  const int32_t t = steps[pos];
  for (int b = threadIdx.x; b < B; b += blockDim.x) emb_row[b] = t;
  __syncthreads();                                    // every thread has read state->pos before it moves
  if (threadIdx.x == 0) {
    state->cur_t = t;
    state->rng_step = raw;
    state->pos = raw + 1;
    *cur_coef = coef_table[t];
  }
}

}  // namespace

extern "C" int mh_cast_pad(const float* in, int64_t ld_in, void* out, int64_t ld_out, int64_t rows, int64_t cols,
                           int64_t rows_out, int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(in && out, "cast_pad: null pointer");
  MH_CHECK_ARG(rows >= 0 && cols >= 0 && rows_out >= rows && ld_out >= cols && ld_in >= cols, "cast_pad: bad shape");
  if (rows_out * ld_out == 0) return MH_OK;
  hipStream_t s = (hipStream_t)stream;
  const int grid = ew_grid(rows_out * ld_out);
  if (dtype == MH_BF16) MH_LAUNCH((cast_pad_kernel<bf16>), dim3(grid), dim3(EW_BLOCK), 0, s, in, ld_in, (bf16*)out, ld_out, rows, cols, rows_out);
  else if (dtype == MH_F32) MH_LAUNCH((cast_pad_kernel<float>), dim3(grid), dim3(EW_BLOCK), 0, s, in, ld_in, (float*)out, ld_out, rows, cols, rows_out);
  else { mh_set_error("cast_pad: unknown dtype %d", dtype); return MH_ERR_INVALID; }
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_cast_to_f32(const void* in, int64_t ld_in, float* out, int64_t ld_out, int64_t rows, int64_t cols,
                              int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(in && out && rows >= 0 && cols >= 0 && ld_in >= cols && ld_out >= cols, "cast_to_f32: bad arguments");
  if (rows * cols == 0) return MH_OK;
  hipStream_t s = (hipStream_t)stream;
  const int grid = ew_grid(rows * cols);
  if (dtype == MH_BF16) MH_LAUNCH((cast_to_f32_kernel<bf16>), dim3(grid), dim3(EW_BLOCK), 0, s, (const bf16*)in, ld_in, out, ld_out, rows, cols);
  else if (dtype == MH_F32) MH_LAUNCH((cast_to_f32_kernel<float>), dim3(grid), dim3(EW_BLOCK), 0, s, (const float*)in, ld_in, out, ld_out, rows, cols);
  else { mh_set_error("cast_to_f32: unknown dtype %d", dtype); return MH_ERR_INVALID; }
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_pack_panel(const float* in, int64_t ld_in, void* out, int64_t ld_rows, int64_t rows, int64_t cols,
                             int64_t cols_pad, mh_stream_t stream) {
  MH_CHECK_ARG(in && out && rows >= 0 && cols >= 0 && ld_rows >= rows && cols_pad >= cols && cols_pad % 32 == 0 && ld_in >= cols,
               "pack_panel: bad arguments");
  if (cols_pad * ld_rows == 0) return MH_OK;
  MH_LAUNCH(pack_panel_kernel, dim3(ew_grid(cols_pad * ld_rows)), dim3(EW_BLOCK), 0, (hipStream_t)stream, in, ld_in,
            (bf16*)out, ld_rows, rows, cols, cols_pad);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_unpack_panel_f32(const void* in, int64_t ld_rows, float* out, int64_t ld_out, int64_t rows, int64_t cols,
                                   mh_stream_t stream) {
  MH_CHECK_ARG(in && out && rows >= 0 && cols >= 0 && ld_rows >= rows && ld_out >= cols, "unpack_panel_f32: bad arguments");
  if (rows * cols == 0) return MH_OK;
  MH_LAUNCH(unpack_panel_kernel, dim3(ew_grid(rows * cols)), dim3(EW_BLOCK), 0, (hipStream_t)stream, (const bf16*)in,
            ld_rows, out, ld_out, rows, cols);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_row_sqnorm(const float* table, float* out, int V, int E, mh_stream_t stream) {
  MH_CHECK_ARG(table && out && V > 0 && E > 0, "row_sqnorm: bad arguments");
  MH_LAUNCH(row_sqnorm_kernel, dim3((V + 3) / 4), dim3(256), 0, (hipStream_t)stream, table, out, V, E);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// get_logits with logits_mode 2 (models/network.py:94-104): scores[n][v] = -sqrt(clamp((|W_v|^2 + |x_n|^2) - 2 W_v.x_n, 0, inf)), from the
// fp32 product W x^T the caller ran as a GEMM; the reference's association of the three terms (this file compiles without FP contraction)
namespace {
__global__ void distance_scores_kernel(const float* __restrict__ dots, int64_t ld, const float* __restrict__ wn, const float* __restrict__ xn,
                                       float* __restrict__ out, int64_t ldo, int64_t n, int V) {
  const int64_t total = n * V;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / V;
    const int v = (int)(i - row * V);
    const float d = (wn[v] + xn[row]) - 2.0f * dots[row * ld + v];
    out[row * ldo + v] = -sqrtf(fmaxf(d, 0.0f));
  }
}
}  // namespace
extern "C" int mh_distance_scores(const float* dots, int64_t ld, const float* w_sqnorm, const float* x_sqnorm, float* out, int64_t ldo, int64_t n,
                                  int V, mh_stream_t stream) {
  MH_CHECK_ARG(dots && w_sqnorm && x_sqnorm && out && n > 0 && V > 0 && ld >= V && ldo >= V, "distance_scores: bad arguments");
  MH_LAUNCH(distance_scores_kernel, dim3(ew_grid(n * V)), dim3(EW_BLOCK), 0, (hipStream_t)stream, dots, ld, w_sqnorm, x_sqnorm, out, ldo, n, V);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_embed_gather(const float* table, const int32_t* ids, float* out, int64_t n_tokens, int E, int V,
                               mh_stream_t stream) {
  MH_CHECK_ARG(table && ids && out, "embed_gather: null pointer");
  MH_CHECK_ARG(n_tokens >= 0 && E > 0 && V > 0, "embed_gather: bad shape");
  if (n_tokens == 0) return MH_OK;
  MH_LAUNCH(embed_gather_kernel, dim3(ew_grid(n_tokens * E)), dim3(EW_BLOCK), 0, (hipStream_t)stream, table,
                     ids, out, n_tokens, E, V);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_timestep_embedding(const float* t, void* out, int B, int dim, int64_t ld_out, float max_period,
                                     int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(t && out && B > 0 && dim > 0 && ld_out >= dim, "timestep_embedding: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const float nlp = -logf(max_period);
  const int grid = ew_grid((int64_t)B * ld_out);
  if (dtype == MH_BF16) MH_LAUNCH((timestep_embedding_kernel<bf16>), dim3(grid), dim3(EW_BLOCK), 0, s, t, (bf16*)out, B, dim, ld_out, nlp);
  else if (dtype == MH_F32) MH_LAUNCH((timestep_embedding_kernel<float>), dim3(grid), dim3(EW_BLOCK), 0, s, t, (float*)out, B, dim, ld_out, nlp);
  else { mh_set_error("timestep_embedding: unknown dtype %d", dtype); return MH_ERR_INVALID; }
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_q_sample(const float* x0, const float* noise, const float* a, const float* s, const int32_t* mask,
                           int mask_per_elem, float* out, int B, int64_t per_batch, int E, mh_stream_t stream) {
  MH_CHECK_ARG(x0 && noise && a && s && out, "q_sample: null pointer");
  MH_CHECK_ARG(B > 0 && per_batch > 0 && E > 0 && per_batch % E == 0, "q_sample: bad shape");
  MH_LAUNCH(q_sample_kernel, dim3(ew_grid((int64_t)B * per_batch)), dim3(EW_BLOCK), 0, (hipStream_t)stream, x0,
                     noise, a, s, mask, mask_per_elem, out, B, per_batch, E);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_p_sample_epilogue(const float* model_out, const float* x_t, const float* noise,
                                    const int32_t* round_idx, const float* table, const mh_step_coef* coef,
                                    int coef_per_batch, int clip, const int32_t* mask, int mask_per_elem,
                                    const float* x_start, float* out, float* pred_xstart, float* mean_out, int B,
                                    int64_t per_batch, int E, mh_stream_t stream) {
  MH_CHECK_ARG(x_t && coef && out, "p_sample_epilogue: null pointer");
  MH_CHECK_ARG(model_out || round_idx, "p_sample_epilogue: need model_out or round_idx");
  MH_CHECK_ARG(!round_idx || table, "p_sample_epilogue: round_idx needs the embedding table");
  MH_CHECK_ARG(!mask || x_start, "p_sample_epilogue: mask needs x_start");
  MH_CHECK_ARG(B > 0 && per_batch > 0 && E > 0 && per_batch % E == 0, "p_sample_epilogue: bad shape");
  if (step_epilogue_vec_ok(model_out, x_t, noise, table, x_start, out, pred_xstart, mean_out, B, per_batch, E)) {
    const int64_t ng = (int64_t)B * per_batch / 4;
    MH_LAUNCH((step_epilogue4_kernel<false>), dim3(ew_grid(ng)), dim3(EW_BLOCK), 0, (hipStream_t)stream, model_out, x_t, noise, round_idx, table, coef,
              coef_per_batch, clip, mask, mask_per_elem, x_start, out, pred_xstart, mean_out, ng, (uint32_t)(per_batch / 4), (uint32_t)(E / 4), E);
  } else
  MH_LAUNCH((step_epilogue_kernel<false>), dim3(ew_grid((int64_t)B * per_batch)), dim3(EW_BLOCK), 0,
                     (hipStream_t)stream, model_out, x_t, noise, round_idx, table, coef, coef_per_batch, clip, mask,
                     mask_per_elem, x_start, out, pred_xstart, mean_out, B, per_batch, E);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_ddim_epilogue(const float* model_out, const float* x_t, const float* noise, const int32_t* round_idx,
                                const float* table, const mh_step_coef* coef, int coef_per_batch, int clip,
                                const int32_t* mask, int mask_per_elem, const float* x_start, float* out,
                                float* pred_xstart, int B, int64_t per_batch, int E, mh_stream_t stream) {
  MH_CHECK_ARG(x_t && coef && out, "ddim_epilogue: null pointer");
  MH_CHECK_ARG(model_out || round_idx, "ddim_epilogue: need model_out or round_idx");
  MH_CHECK_ARG(!round_idx || table, "ddim_epilogue: round_idx needs the embedding table");
  MH_CHECK_ARG(!mask || x_start, "ddim_epilogue: mask needs x_start");
  MH_CHECK_ARG(B > 0 && per_batch > 0 && E > 0 && per_batch % E == 0, "ddim_epilogue: bad shape");
  if (step_epilogue_vec_ok(model_out, x_t, noise, table, x_start, out, pred_xstart, nullptr, B, per_batch, E)) {
    const int64_t ng = (int64_t)B * per_batch / 4;
    MH_LAUNCH((step_epilogue4_kernel<true>), dim3(ew_grid(ng)), dim3(EW_BLOCK), 0, (hipStream_t)stream, model_out, x_t, noise, round_idx, table, coef,
              coef_per_batch, clip, mask, mask_per_elem, x_start, out, pred_xstart, (float*)nullptr, ng, (uint32_t)(per_batch / 4), (uint32_t)(E / 4), E);
  } else
  MH_LAUNCH((step_epilogue_kernel<true>), dim3(ew_grid((int64_t)B * per_batch)), dim3(EW_BLOCK), 0,
                     (hipStream_t)stream, model_out, x_t, noise, round_idx, table, coef, coef_per_batch, clip, mask,
                     mask_per_elem, x_start, out, pred_xstart, (float*)nullptr, B, per_batch, E);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_step_epilogue_slots(int ddim, const float* x_t, const float* noise, const float* pbest, const int32_t* pidx, int nslots,
                                      const float* table, const mh_step_coef* coef, int coef_per_batch, int clip, const int32_t* mask,
                                      int mask_per_elem, const float* x_start, float* out, float* pred_xstart, float* mean_out,
                                      int32_t* round_idx_out, const mh_step_rng* rng, int B, int64_t per_batch, int E, mh_stream_t stream) {
  MH_CHECK_ARG(x_t && coef && out && pidx && table && nslots >= 0 && (pbest || nslots == 0), "step_epilogue_slots: null pointer");
  MH_CHECK_ARG(!rng || (rng->first_elem % 4 == 0 && (rng->bound <= 0.f || rng->bound >= 0.1f)), "step_epilogue_slots: bad rng descriptor");
  MH_CHECK_ARG(!mask || x_start, "step_epilogue_slots: mask needs x_start");
  MH_CHECK_ARG(B > 0 && per_batch > 0 && E > 0 && per_batch % E == 0, "step_epilogue_slots: bad shape");
  MH_CHECK_ARG(step_epilogue_vec_ok(nullptr, x_t, noise, table, x_start, out, pred_xstart, ddim ? nullptr : mean_out, B, per_batch, E),
               "step_epilogue_slots: needs E %% 4 == 0 and 16-byte aligned tensors");
  const int64_t ng = (int64_t)B * per_batch / 4;
  const float* none = nullptr;
  const int32_t* nidx = nullptr;
  if (rng) {
    const StepRng r{(uint32_t)rng->seed, (uint32_t)(rng->seed >> 32), rng->stream_id, rng->bound, rng->step_counter, (uint64_t)(rng->first_elem >> 2)};
    if (ddim)
      MH_LAUNCH((step_epilogue4_kernel<true, true, true>), dim3(ew_grid(ng)), dim3(EW_BLOCK), 0, (hipStream_t)stream, none, x_t, none, nidx, table, coef,
                coef_per_batch, clip, mask, mask_per_elem, x_start, out, pred_xstart, (float*)nullptr, ng, (uint32_t)(per_batch / 4), (uint32_t)(E / 4), E,
                pbest, pidx, nslots, round_idx_out, r);
    else
      MH_LAUNCH((step_epilogue4_kernel<false, true, true>), dim3(ew_grid(ng)), dim3(EW_BLOCK), 0, (hipStream_t)stream, none, x_t, none, nidx, table, coef,
                coef_per_batch, clip, mask, mask_per_elem, x_start, out, pred_xstart, mean_out, ng, (uint32_t)(per_batch / 4), (uint32_t)(E / 4), E,
                pbest, pidx, nslots, round_idx_out, r);
    MH_CHECK_LAUNCH();
    return MH_OK;
  }
  if (ddim)
    MH_LAUNCH((step_epilogue4_kernel<true, true>), dim3(ew_grid(ng)), dim3(EW_BLOCK), 0, (hipStream_t)stream, none, x_t, noise, nidx, table, coef,
              coef_per_batch, clip, mask, mask_per_elem, x_start, out, pred_xstart, (float*)nullptr, ng, (uint32_t)(per_batch / 4), (uint32_t)(E / 4), E,
              pbest, pidx, nslots, round_idx_out);
  else
    MH_LAUNCH((step_epilogue4_kernel<false, true>), dim3(ew_grid(ng)), dim3(EW_BLOCK), 0, (hipStream_t)stream, none, x_t, noise, nidx, table, coef,
              coef_per_batch, clip, mask, mask_per_elem, x_start, out, pred_xstart, mean_out, ng, (uint32_t)(per_batch / 4), (uint32_t)(E / 4), E,
              pbest, pidx, nslots, round_idx_out);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_trunc_normal_at(float* out, int64_t n, int64_t first, float bound, uint64_t seed, uint32_t stream_id,
                                  const uint32_t* step_counter, mh_stream_t stream) {
  MH_CHECK_ARG(out && n >= 0 && first >= 0 && first % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
               "trunc_normal: bad arguments (the first element index must be a multiple of 4, the output 16-byte aligned)");
  MH_CHECK_ARG(bound <= 0.f || bound >= 0.1f, "trunc_normal: bound %g too tight for rejection sampling", (double)bound);
  if (n == 0) return MH_OK;
  mh_prof_note("n=%lld first=%lld", (long long)n, (long long)first);
  MH_LAUNCH(trunc_normal_kernel, dim3(ew_grid((n + 3) / 4)), dim3(EW_BLOCK), 0, (hipStream_t)stream, out, n, first, bound,
                     (uint32_t)seed, (uint32_t)(seed >> 32), stream_id, step_counter);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
extern "C" int mh_trunc_normal(float* out, int64_t n, float bound, uint64_t seed, uint32_t stream_id,
                               const uint32_t* step_counter, mh_stream_t stream) {
  return mh_trunc_normal_at(out, n, 0, bound, seed, stream_id, step_counter, stream);
}

namespace {
// idles one wave for about `us` microseconds (s_memrealtime ticks at 100 MHz): puts a phase lag between concurrent streams
__global__ void spin_kernel(unsigned us) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), d = 100ull * us;
  while (__builtin_amdgcn_s_memrealtime() - t0 < d) __builtin_amdgcn_s_sleep(32);
}
}  // namespace
extern "C" int mh_stream_delay(unsigned microseconds, mh_stream_t stream) {
  MH_CHECK_ARG(microseconds <= 100000u, "stream_delay: at most 100 ms");
  if (microseconds == 0) return MH_OK;
  MH_LAUNCH(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, microseconds);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_step_begin(mh_loop_state* state, const int32_t* steps, const mh_step_coef* coef_table,
                             mh_step_coef* cur_coef, int32_t* emb_row, int B, mh_stream_t stream) {
  MH_CHECK_ARG(state && steps && coef_table && cur_coef && emb_row && B > 0, "step_begin: bad arguments");
  MH_LAUNCH(step_begin_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, state, steps, coef_table, cur_coef,
                     emb_row, B);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_step_advance(mh_loop_state* state, const int32_t* steps, const mh_step_coef* coef_table, mh_step_coef* cur_coef,
                               int32_t* emb_row, int B, mh_stream_t stream) {
  MH_CHECK_ARG(state && steps && coef_table && cur_coef && emb_row && B > 0, "step_advance: bad arguments");
  MH_LAUNCH(step_advance_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, state, steps, coef_table, cur_coef, emb_row, B);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_step_end(mh_loop_state* state, mh_stream_t stream) {
  MH_CHECK_ARG(state, "step_end: null state");
  MH_LAUNCH(step_end_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
