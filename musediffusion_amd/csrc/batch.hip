// Batch producers and token validators on the device: the integer work either side of
// the hot path.  Reference: MuseDiffusion/data/corruption.py:100-195 (four corruptions), data/wrapper.py:90-127
// (collate_batches), utils/decode_util.py:221-230 (meta_to_batch), :73-84 / :142-183 (remove_padding, validate_once,
// validate_rigidly).  Sequences are ragged: `values` = all rows back to back (int32), `offsets[B + 1]` (int64).
// One 256-thread block per row; rows of up to MH_BATCH_MAX_ROW tokens (the row lives in LDS).  All kernels are
// HBM / latency bound byte movers - no MFMA shape here on purpose.  Random draws are inputs (the reference's
// random.Random stream cannot be reproduced on a GPU; see oracle/batch.py), indexed the way the reference consumes them.
#include "common.h"

namespace {

constexpr int TB = 256;
constexpr int MAX_ROW = 4096;   // tokens per row the LDS-resident kernels accept (the reference's seq_len tops out at 2096)

// exclusive prefix sum of one flag per thread over the block (256 threads = 4 waves); returns the block total in `total`
__device__ __forceinline__ int block_excl_scan(int flag, int* wsum, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m = __builtin_amdgcn_ballot_w64(flag != 0);
  const int before = __builtin_popcountll(m & ((1ull << lane) - 1ull));
  if (lane == 0) wsum[wave] = __builtin_popcountll(m);
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  return base + before;
}

__global__ void ragged_to_padded_kernel(const int32_t* __restrict__ values, const int64_t* __restrict__ offsets, int32_t* __restrict__ out,
                                        int32_t* __restrict__ length, int L, int32_t pad) {
  const int b = blockIdx.x;
  const int64_t off = offsets[b];
  const int n = (int)(offsets[b + 1] - off);
  for (int j = threadIdx.x; j < L; j += TB) out[(int64_t)b * L + j] = j < n ? values[off + j] : pad;
  if (length && threadIdx.x == 0) length[b] = n;
}

__global__ void meta_to_batch_kernel(const int32_t* __restrict__ meta, int len_meta, int32_t* __restrict__ ids, int32_t* __restrict__ mask,
                                     int64_t total, int L) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % L);
    ids[i] = j < len_meta ? meta[j] : 0;
    mask[i] = j < len_meta + 1 ? 0 : 1;
  }
}

// masking_token (corruption.py:100-114)
__global__ void corrupt_mt_kernel(const int32_t* __restrict__ values, const int64_t* __restrict__ offsets, const float* __restrict__ u,
                                  float p, int32_t* __restrict__ out) {
  __shared__ int eos_s;
  const int b = blockIdx.x;
  const int64_t off = offsets[b];
  const int n = (int)(offsets[b + 1] - off);
  if (threadIdx.x == 0) eos_s = n;
  __syncthreads();
  int first = n;
  for (int j = 12 + threadIdx.x; j < n; j += TB)
    if (values[off + j] == 1 && j < first) first = j;
  if (first < n) atomicMin(&eos_s, first);
  __syncthreads();
  const int eos = eos_s;
  for (int j = threadIdx.x; j < n; j += TB) {
    const int32_t v = values[off + j];
    out[off + j] = (j >= 12 && j < eos && u[off + j - 12] < p) ? 0 : v;
  }
}

// masking_note / randomize_note (corruption.py:117-162).  MODE 0: zero seq[idx-1 : idx+3]; MODE 1: (velocity, pitch, duration) := new[rank]
template <int MODE>
__global__ void corrupt_note_kernel(const int32_t* __restrict__ values, const int64_t* __restrict__ offsets, const float* __restrict__ u,
                                    const int32_t* __restrict__ newv, float p, int32_t* __restrict__ out) {
  __shared__ int rank_of[MAX_ROW];     // rank of the velocity token at j if it fires, else -1
  __shared__ int wsum[4];
  const int b = blockIdx.x;
  const int64_t off = offsets[b];
  const int n = (int)(offsets[b + 1] - off);
  if (n > MAX_ROW) {   // host wrappers reject such rows; never index LDS out of range
    for (int j = threadIdx.x; j < n; j += TB) out[off + j] = values[off + j];
    return;
  }
  int carry = 0;
  for (int j0 = 0; j0 < n; j0 += TB) {
    const int j = j0 + threadIdx.x;
    int elig = 0;
    if (j < n) {
      const int32_t v = values[off + j];
      elig = (v >= 131 && v <= 194 && j + 3 <= n) ? 1 : 0;
    }
    int total;
    const int r = carry + block_excl_scan(elig, wsum, total);
    if (j < n) rank_of[j] = (elig && u[off + r] < p) ? r : -1;
    carry += total;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < n; j += TB) {
    int32_t v = values[off + j];
    if constexpr (MODE == 0) {
      // zeroed when a firing velocity token sits at idx in [j-2, j+1] (it clears idx-1 .. idx+2); idx = 0 clears seq[-1:3] = nothing
      bool z = false;
#pragma unroll
      for (int d = -2; d <= 1; ++d) {
        const int idx = j + d;
        if (idx >= 1 && idx < n && rank_of[idx] >= 0) z = true;
      }
      if (z) v = 0;
    } else {
      // sequential semantics: the firing token with the LARGEST idx <= j wins (it is processed last)
      if (rank_of[j] >= 0) v = newv[(off + rank_of[j]) * 3 + 0];
      else if (j >= 1 && rank_of[j - 1] >= 0) v = newv[(off + rank_of[j - 1]) * 3 + 1];
      else if (j >= 2 && rank_of[j - 2] >= 0) v = newv[(off + rank_of[j - 2]) * 3 + 2];
    }
    out[off + j] = v;
  }
}

// random_rotating (corruption.py:165-195): `count` bar swaps; bar starts and the last EOS come from the INPUT row once
__global__ void corrupt_rr_kernel(const int32_t* __restrict__ values, const int64_t* __restrict__ offsets, const int32_t* __restrict__ pairs,
                                  int count, int32_t* __restrict__ out, int32_t* __restrict__ status) {
  __shared__ int32_t buf[2][MAX_ROW];
  __shared__ int bars[MAX_ROW / 2];
  __shared__ int wsum[4];
  __shared__ int eos_s, nbar_s;
  const int b = blockIdx.x;
  const int64_t off = offsets[b];
  const int n = (int)(offsets[b + 1] - off);
  if (n > MAX_ROW) {
    for (int j = threadIdx.x; j < n; j += TB) out[off + j] = values[off + j];
    if (status && threadIdx.x == 0) status[b] = 2;
    return;
  }
  if (threadIdx.x == 0) eos_s = -1;
  __syncthreads();
  int carry = 0, last_eos = -1;
  for (int j0 = 0; j0 < n; j0 += TB) {
    const int j = j0 + threadIdx.x;
    int isbar = 0;
    if (j < n) {
      const int32_t v = values[off + j];
      buf[0][j] = v;
      isbar = v == 2;
      if (v == 1) last_eos = j;
    }
    int total;
    const int r = carry + block_excl_scan(isbar, wsum, total);
    if (isbar && r < MAX_ROW / 2) bars[r] = j;
    carry += total;
  }
  if (last_eos >= 0) atomicMax(&eos_s, last_eos);
  if (threadIdx.x == 0) nbar_s = carry;
  __syncthreads();
  const int nbar = nbar_s, eos = eos_s;
  int cur = 0;
  bool ok = nbar > 1 && eos >= 0;
  for (int s = 0; s < count && ok; ++s) {
    const int first = pairs[((int64_t)b * count + s) * 2], second = pairs[((int64_t)b * count + s) * 2 + 1];
    if (first < 0 || second <= first || second >= nbar) { ok = false; break; }
    const int b1s = bars[first], b2s = bars[second], b1e = bars[first + 1];
    const int b2e = second < nbar - 1 ? bars[second + 1] : eos;
    if (b2e < b2s) { ok = false; break; }
    const int lenB2 = b2e - b2s, lenM = b2s - b1e;
    for (int j = threadIdx.x; j < n; j += TB) {
      int src = j;
      if (j >= b1s && j < b2e) {
        const int q = j - b1s;
        if (q < lenB2) src = b2s + q;
        else if (q < lenB2 + lenM) src = b1e + (q - lenB2);
        else src = b1s + (q - lenB2 - lenM);
      }
      buf[cur ^ 1][j] = buf[cur][src];
    }
    __syncthreads();
    cur ^= 1;
  }
  for (int j = threadIdx.x; j < n; j += TB) out[off + j] = buf[cur][j];
  if (status && threadIdx.x == 0) status[b] = ok ? 0 : 1;
}

// remove_padding / validate_once / validate_rigidly (decode_util.py:73-84, :142-183) of row b of tokens [B, L]:
// result[b] = (index of the first EOS or -1, once ok, rigid ok | -2 where the reference indexes past the end)
__global__ void validate_tokens_kernel(const int32_t* __restrict__ tokens, const int32_t* __restrict__ lens, int32_t* __restrict__ result, int L) {
  __shared__ int eos_s, once_s;
  const int b = blockIdx.x, lane = threadIdx.x;
  const int32_t* seq = tokens + (int64_t)b * L;
  const int len = lens ? lens[b] : L;
  if (lane == 0) { eos_s = len; once_s = 0; }
  __syncthreads();
  int first = len;
  for (int j = lane; j < len; j += 64)
    if (seq[j] == 1 && j < first) first = j;
  if (first < len) atomicMin(&eos_s, first);
  __syncthreads();
  if (eos_s >= len) {
    if (lane == 0) { result[b * 3 + 0] = -1; result[b * 3 + 1] = 0; result[b * 3 + 2] = 0; }
    return;
  }
  const int n = eos_s + 1;   // the cut sequence: everything up to and including the first EOS
  // validate_once: some idx <= n - 3 holds a velocity token between a position token and (pitch, duration); seq[-1] wraps
  bool hit = false;
  for (int i = lane; i + 2 <= n - 1; i += 64) {
    const int32_t t = seq[i], prev = seq[i == 0 ? n - 1 : i - 1];
    if (t >= 131 && t < 195 && prev >= 432 && prev < 560 && seq[i + 1] >= 3 && seq[i + 1] < 131 && seq[i + 2] >= 304 && seq[i + 2] < 432) hit = true;
  }
  if (hit) once_s = 1;
  __syncthreads();
  if (lane == 0) {
    int rigid = 0, i = 0;
    while (true) {
      if (i >= n) break;
      const int32_t t = seq[i];
      if (t == 1) { rigid = 1; break; }
      if (t == 2) { ++i; continue; }
      if (!(t >= 432 && t < 560)) break;
      if (i + 1 >= n) { rigid = -2; break; }
      const int32_t t1 = seq[i + 1];
      if (t1 >= 131 && t1 < 195) {
        if (i + 3 >= n) { rigid = -2; break; }
        if (seq[i + 2] >= 3 && seq[i + 2] < 131 && seq[i + 3] >= 304 && seq[i + 3] < 432) { i += 4; continue; }
        break;
      }
      if (t1 >= 195 && t1 < 304) { i += 2; continue; }
      break;
    }
    result[b * 3 + 0] = eos_s;
    result[b * 3 + 1] = once_s;
    result[b * 3 + 2] = rigid;
  }
}

// MSIM feature vectors (metric.py:4-71 get_vectors): one lane per sequence walks its note tokens - the scan is a data-dependent
// state machine (steps of 1, 2 or 4 tokens), so sequences, not tokens, are the parallel dimension.
// out[b] = [32 rhythm | 12 melody | 12 harmony], each L2-normalised; status[b]: 0 ok, 1 no BAR, 2 malformed token / ran off the end
__global__ void msim_vectors_kernel(const int32_t* __restrict__ tokens, const int32_t* __restrict__ lens, float* __restrict__ out,
                                    int32_t* __restrict__ status, int B, int L, float note_len) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int32_t* midi = tokens + (int64_t)b * L;
  const int n = lens ? lens[b] : L;
  float rhythm[32], tmp[32], melody[12], harmony[12];
#pragma unroll
  for (int k = 0; k < 32; ++k) { rhythm[k] = 1e-8f; tmp[k] = 1e-8f; }
#pragma unroll
  for (int k = 0; k < 12; ++k) { melody[k] = 1e-8f; harmony[k] = 0.f; }
  int st = 0, i = 0;
  while (i < n && midi[i] != 2) ++i;
  if (i >= n) st = 1;
  ++i;
  int cur_hi = -1, prev_hi = -1, prev_startp = -1, startp = -2;
  auto norm32 = [](const float* v, int cnt) {
    float ss = 0.f;
    for (int k = 0; k < cnt; ++k) ss += v[k] * v[k];
    return sqrtf(ss);
  };
  while (st == 0) {
    if (i >= n) { st = 2; break; }
    const int32_t tok = midi[i];
    if (tok <= 2) {
      const float nt = norm32(tmp, 32);
      for (int k = 0; k < 32; ++k) { rhythm[k] += tmp[k] / nt; tmp[k] = 1e-8f; }
      ++i;
      if (tok == 2) { prev_startp = -1; continue; }
      if (prev_startp != startp && prev_hi >= 0) melody[(((cur_hi - prev_hi) % 12) + 12) % 12] += 1.f;
      break;
    }
    if (tok < 432 || tok > 559 || i + 1 >= n) { st = 2; break; }
    startp = tok - 432;
    const int32_t t1 = midi[i + 1];
    if (t1 >= 195 && t1 <= 303) { i += 2; continue; }
    if (i + 3 >= n || t1 < 131 || t1 > 194 || midi[i + 2] < 3 || midi[i + 2] > 130 || midi[i + 3] < 304 || midi[i + 3] > 431) { st = 2; break; }
    const int pitch = midi[i + 2];
    const int endp = startp + midi[i + 3] - 303;
    harmony[pitch % 12] += 1.f;
    const double amp0 = 0.00542676376 * (double)(t1 - 130) * 2.0 + 0.310801;
    const double amp = amp0 * amp0;
    const int tend = endp < 128 ? endp : 128;
    for (int t = 0; t < tend; t += 4) {
      if (t < startp) continue;
      double w = 1.0 - (double)(t - startp) / (double)note_len;
      if (w < 0.0) w = 0.0;
      const double v = amp * w;
      if (v > (double)tmp[t >> 2]) tmp[t >> 2] = (float)v;
    }
    if (cur_hi >= 0 && prev_startp != startp) {
      if (prev_hi >= 0) melody[(((cur_hi - prev_hi) % 12) + 12) % 12] += 1.f;
      prev_hi = cur_hi;
      cur_hi = pitch;
    }
    cur_hi = pitch > cur_hi ? pitch : cur_hi;
    prev_startp = startp;
    i += 4;
  }
  float* o = out + (int64_t)b * 56;
  const float nr = norm32(rhythm, 32), nm = norm32(melody, 12), nh = norm32(harmony, 12);
  for (int k = 0; k < 32; ++k) o[k] = rhythm[k] / nr;
  for (int k = 0; k < 12; ++k) o[32 + k] = melody[k] / nm;
  for (int k = 0; k < 12; ++k) o[44 + k] = harmony[k] / nh;
  if (status) status[b] = st;
}

// Controllability counters (metric.py:120-168): per row, (sum and count of pitch tokens 3..130) and (count of velocity tokens
// 131..194, count of those outside the meta's [min, max]).  out[b] = (pitch_sum, pitch_count, vel_total, vel_wrong) int32;
// meta[b] = the 11 meta tokens.  One wave per row.
__global__ void controllability_kernel(const int32_t* __restrict__ tokens, const int32_t* __restrict__ lens, const int32_t* __restrict__ metas,
                                       int32_t* __restrict__ out, int L, int meta_ld) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int32_t* midi = tokens + (int64_t)b * L;
  const int n = lens ? lens[b] : L;
  const int lo_v = metas[(int64_t)b * meta_ld + 7] - 524, hi_v = metas[(int64_t)b * meta_ld + 8] - 524;
  int psum = 0, pcnt = 0, vtot = 0, vbad = 0;
  for (int j = lane; j < n; j += 64) {
    const int32_t t = midi[j];
    if (t >= 3 && t <= 130) { psum += t; ++pcnt; }
    if (t >= 131 && t <= 194) {
      ++vtot;
      if (!((lo_v == 130 || lo_v <= t) && (hi_v == 195 || t <= hi_v))) ++vbad;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    psum += __shfl_xor(psum, o, 64); pcnt += __shfl_xor(pcnt, o, 64);
    vtot += __shfl_xor(vtot, o, 64); vbad += __shfl_xor(vbad, o, 64);
  }
  if (lane == 0) { out[b * 4 + 0] = psum; out[b * 4 + 1] = pcnt; out[b * 4 + 2] = vtot; out[b * 4 + 3] = vbad; }
}

}  // namespace

extern "C" int mh_batch_max_row(void) { return MAX_ROW; }

extern "C" int mh_ragged_to_padded(const int32_t* values, const int64_t* offsets, int32_t* out, int32_t* length, int B, int L,
                                   int32_t pad, mh_stream_t stream) {
  MH_CHECK_ARG(values && offsets && out && B > 0 && L > 0, "ragged_to_padded: bad arguments");
  MH_LAUNCH(ragged_to_padded_kernel, dim3(B), dim3(TB), 0, (hipStream_t)stream, values, offsets, out, length, L, pad);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_meta_to_batch(const int32_t* meta, int len_meta, int32_t* ids, int32_t* mask, int B, int L, mh_stream_t stream) {
  MH_CHECK_ARG(meta && ids && mask && B > 0 && L > 0 && len_meta >= 0 && len_meta <= L, "meta_to_batch: bad arguments");
  const int64_t total = (int64_t)B * L;
  const int grid = (int)((total + TB - 1) / TB < 4096 ? (total + TB - 1) / TB : 4096);
  MH_LAUNCH(meta_to_batch_kernel, dim3(grid), dim3(TB), 0, (hipStream_t)stream, meta, len_meta, ids, mask, total, L);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_corrupt_masking_token(const int32_t* values, const int64_t* offsets, const float* u, float p, int32_t* out, int B,
                                        mh_stream_t stream) {
  MH_CHECK_ARG(values && offsets && u && out && B > 0, "corrupt_masking_token: bad arguments");
  MH_LAUNCH(corrupt_mt_kernel, dim3(B), dim3(TB), 0, (hipStream_t)stream, values, offsets, u, p, out);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_corrupt_masking_note(const int32_t* values, const int64_t* offsets, const float* u, float p, int32_t* out, int B,
                                       mh_stream_t stream) {
  MH_CHECK_ARG(values && offsets && u && out && B > 0, "corrupt_masking_note: bad arguments");
  MH_LAUNCH((corrupt_note_kernel<0>), dim3(B), dim3(TB), 0, (hipStream_t)stream, values, offsets, u, (const int32_t*)nullptr, p, out);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_corrupt_randomize_note(const int32_t* values, const int64_t* offsets, const float* u, const int32_t* new_tokens,
                                         float p, int32_t* out, int B, mh_stream_t stream) {
  MH_CHECK_ARG(values && offsets && u && new_tokens && out && B > 0, "corrupt_randomize_note: bad arguments");
  MH_LAUNCH((corrupt_note_kernel<1>), dim3(B), dim3(TB), 0, (hipStream_t)stream, values, offsets, u, new_tokens, p, out);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_corrupt_random_rotating(const int32_t* values, const int64_t* offsets, const int32_t* pairs, int count, int32_t* out,
                                          int32_t* status, int B, mh_stream_t stream) {
  MH_CHECK_ARG(values && offsets && pairs && out && B > 0 && count >= 0, "corrupt_random_rotating: bad arguments");
  MH_LAUNCH(corrupt_rr_kernel, dim3(B), dim3(TB), 0, (hipStream_t)stream, values, offsets, pairs, count, out, status);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_validate_tokens(const int32_t* tokens, const int32_t* lens, int32_t* result, int B, int L, mh_stream_t stream) {
  MH_CHECK_ARG(tokens && result && B > 0 && L > 0, "validate_tokens: bad arguments");
  MH_LAUNCH(validate_tokens_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, tokens, lens, result, L);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_msim_vectors(const int32_t* tokens, const int32_t* lens, float* out, int32_t* status, int B, int L, float note_len,
                               mh_stream_t stream) {
  MH_CHECK_ARG(tokens && out && B > 0 && L > 0 && note_len > 0.f, "msim_vectors: bad arguments");
  MH_LAUNCH(msim_vectors_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, tokens, lens, out, status, B, L, note_len);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_controllability_counts(const int32_t* tokens, const int32_t* lens, const int32_t* metas, int meta_ld, int32_t* out, int B,
                                         int L, mh_stream_t stream) {
  MH_CHECK_ARG(tokens && metas && out && B > 0 && L > 0 && meta_ld >= 9, "controllability_counts: bad arguments");
  MH_LAUNCH(controllability_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, tokens, lens, metas, out, L, meta_ld);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
