// Nearest-embedding rounding (the "clamp" denoised_fn, models/rounding.py:21-47) and the final
// logits argmax (run/sample.py:218-220 over models/network.py:91-93).  Always fp32: the index that
// comes out must equal the reference's, so the contraction runs on the fp32 VALU (same peak rate as
// the fp32 MFMA on gfx950) with fmaf chains, the reference's expression order for the distance
// (emb_norm + arr_norm) - 2.0 * dot, clamp at 0, and first-index tie-breaking like torch.max /
// torch.argmax.  One workgroup = 64 tokens x the whole vocabulary, vocabulary and embedding dim walked
// in 64 x 64 LDS tiles; the [N, V] score matrix never exists in memory.
#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int TLD = 68;  // padded LDS row (floats): rows tx+16b / 4ty+a read as float4 are conflict-free

template <int MODE>  // 0: rounding (argmin distance)   1: logits (argmax x.W + b)
__global__ __launch_bounds__(256) void vocab_argmax_kernel(const float* __restrict__ x, const float* __restrict__ table,
                                                           const float* __restrict__ aux, int32_t* __restrict__ idx_out,
                                                           int64_t n_tokens, int E, int V) {
  __shared__ __attribute__((aligned(16))) float Xs[64 * TLD];
  __shared__ __attribute__((aligned(16))) float Ws[64 * TLD];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int64_t n0 = (int64_t)blockIdx.x * 64;
  const int ne = (E + 63) / 64;

  float best[4];
  int bidx[4];
  float xn[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 4; ++a) { best[a] = -INFINITY; bidx[a] = 0; }

  for (int v0 = 0; v0 < V; v0 += 64) {
    float dot[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) dot[a][b] = 0.f;
    for (int ec = 0; ec < ne; ++ec) {
      const int e0 = ec * 64;
      __syncthreads();
      for (int i = tid; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        int64_t tok = n0 + r; if (tok >= n_tokens) tok = n_tokens - 1;
        int vr = v0 + r; if (vr >= V) vr = V - 1;
        const bool in = (e0 + c) < E;
        Xs[r * TLD + c] = in ? x[tok * E + e0 + c] : 0.f;
        Ws[r * TLD + c] = in ? table[(int64_t)vr * E + e0 + c] : 0.f;
      }
      __syncthreads();
      if (MODE == 0 && v0 == 0) {
        // |x_n|^2, accumulated once (first vocabulary tile): lanes split the chunk, xor-reduce over tx
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          float s = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float v = Xs[(4 * ty + a) * TLD + tx + 16 * k];
            s += v * v;
          }
#pragma unroll
          for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
          xn[a] += s;
        }
      }
#pragma unroll 2
      for (int e = 0; e < 64; e += 4) {
        f32x4 xv[4], wv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) xv[a] = *reinterpret_cast<const f32x4*>(Xs + (4 * ty + a) * TLD + e);
#pragma unroll
        for (int b = 0; b < 4; ++b) wv[b] = *reinterpret_cast<const f32x4*>(Ws + (tx + 16 * b) * TLD + e);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int k = 0; k < 4; ++k) dot[a][b] = fmaf(xv[a][k], wv[b][k], dot[a][b]);
      }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int v = v0 + tx + 16 * b;
      if (v < V) {
        const float av = aux[v];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          float score;
          if (MODE == 0) {
            float dist = (av + xn[a]) - 2.0f * dot[a][b];
            dist = fmaxf(dist, 0.0f);
            score = -dist;
          } else {
            score = dot[a][b] + av;
          }
          if (score > best[a]) { best[a] = score; bidx[a] = v; }  // increasing v per lane: keeps the first
        }
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    float s = best[a];
    int i = bidx[a];
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      const float so = __shfl_xor(s, off, 64);
      const int io = __shfl_xor(i, off, 64);
      if (so > s || (so == s && io < i)) { s = so; i = io; }
    }
    const int64_t tok = n0 + 4 * ty + a;
    if (tx == 0 && tok < n_tokens) idx_out[tok] = i;
  }
}

}  // namespace

extern "C" int mh_round_to_embedding(const float* x, const float* table, const float* table_norm, int32_t* idx,
                                     int64_t n_tokens, int E, int V, mh_stream_t stream) {
  MH_CHECK_ARG(x && table && table_norm && idx, "round_to_embedding: null pointer");
  MH_CHECK_ARG(n_tokens > 0 && E > 0 && V > 0, "round_to_embedding: bad shape");
  MH_LAUNCH((vocab_argmax_kernel<0>), dim3((unsigned)((n_tokens + 63) / 64)), dim3(256), 0,
                     (hipStream_t)stream, x, table, table_norm, idx, n_tokens, E, V);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_logits_argmax(const float* x, const float* table, const float* bias, int32_t* idx, int64_t n_tokens,
                                int E, int V, mh_stream_t stream) {
  MH_CHECK_ARG(x && table && bias && idx, "logits_argmax: null pointer");
  MH_CHECK_ARG(n_tokens > 0 && E > 0 && V > 0, "logits_argmax: bad shape");
  MH_LAUNCH((vocab_argmax_kernel<1>), dim3((unsigned)((n_tokens + 63) / 64)), dim3(256), 0,
                     (hipStream_t)stream, x, table, bias, idx, n_tokens, E, V);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
