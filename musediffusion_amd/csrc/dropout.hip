// Train-mode dropout outside the fused kernels (reference: nn.Dropout at models/network.py:76, :149 and the HF BertEncoder's
// hidden / attention-probability dropouts, both 0.1 from the bert-base config, network.py:44-46).
//   mh_dropout_fwd         y = x o keep / (1 - p) over a dense [rows, cols] activation: the embedding-LayerNorm site forward,
//                          and the BACKWARD of every dense site (the mask is re-created from (seed, offset), never stored)
//   mh_dropout_bits        the attention-probability keep mask as a bit tensor (lane-native layout, common.h drop_word_index),
//                          exactly the words the fused forward (attention.hip) writes
//   mh_dropout_bits_apply  P o keep / (1 - p) on a materialised [B nh, L, ldp] probability / gradient tensor (fp32 parity mode
//                          and shapes the streaming kernels do not serve)
// The Philox counter conventions live in common.h (drop_keep8 / drop_keep_attn).
#include "common.h"

int mh_drop_args(const mh_dropout* d, DropArgs* out) {
  DropArgs a{};
  if (d && d->p > 0.f) {
    MH_CHECK_ARG(d->p < 1.f, "dropout: p=%g must be in [0, 1)", (double)d->p);
    a.thr = (uint32_t)(d->p * 65536.0f + 0.5f);
    const float p256 = d->p * 256.0f;
    a.thr8 = (uint32_t)p256;                                            // floor
    a.thr_tie = (uint32_t)((p256 - (float)a.thr8) * 256.0f + 0.5f);     // drop on the tie when the next byte is below this
    a.rscale = 1.0f / (1.0f - d->p);
    a.seed_lo = (uint32_t)d->seed; a.seed_hi = (uint32_t)(d->seed >> 32);
    a.off_lo = (uint32_t)d->offset; a.off_hi = (uint32_t)(d->offset >> 32);
    a.mask = d->mask;
  } else {
    a.rscale = 1.0f;
  }
  *out = a;
  return MH_OK;
}

namespace {

template <typename T>
__global__ void dropout_fwd_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ out, int64_t ldo, int64_t rows, int cols,
                                   const DropArgs d) {
  const int gpr = cols >> 3;                                  // 8-element groups per row
  const int64_t total = rows * gpr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / gpr;
    const int col = (int)(i - row * gpr) << 3;
    float v[8];
    load8(x + row * ldx + col, v);
    const uint32_t m = drop_keep8_at(d, (uint64_t)row * cols + col);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (m >> e) & 1u ? v[e] * d.rscale : 0.f;
    store8(out + row * ldo + col, v);
  }
}

// one wave per (bh, 32-query block, run of `kchunk` pairs of 32-key blocks - the 64-key S^T tiles of the fused kernels), lane-native
// words.  No division in the inner loop, and the counter advances by additions.
__global__ __launch_bounds__(256) void dropout_bits_kernel(uint32_t* __restrict__ bits, int nitems, int nchunks, int kchunk, int L, int nb, int nkp,
                                                           const DropArgs d) {
  const int lane = threadIdx.x & 63, lq = lane & 31, h = lane >> 5;
  const uint32_t thr16 = (d.thr8 << 8) + d.thr_tie;
  for (int item = blockIdx.x * 4 + (threadIdx.x >> 6); item < nitems; item += gridDim.x * 4) {
    const int rowblk = item / nchunks, kp0 = (item - rowblk * nchunks) * kchunk;
    const int bh = rowblk / nb, qb = rowblk - bh * nb;
    const int kp1 = kp0 + kchunk < nkp ? kp0 + kchunk : nkp;
    int q = qb * 32 + lq; if (q >= L) q = L - 1;
    const uint64_t idx0 = ((((uint64_t)bh * L + q) * nb) << 1) | (uint64_t)h;     // drop_keep_attn's counter for key block 0
    uint32_t* dst = bits + drop_word_index(bh, nb, qb, 0, lane);
    for (int kp = kp0; kp < kp1; ++kp) {
      const uint64_t i0 = idx0 + 4 * (uint64_t)kp;
      uint32_t c0[4] = {(uint32_t)i0, (uint32_t)(i0 >> 32), d.off_lo, d.off_hi};
      mh_philox<7>(c0, d.seed_lo, d.seed_hi);
      uint32_t kw = drop_keep_attn_from(c0, thr16);
      if (2 * kp + 1 < nb) {
        const uint64_t i1 = i0 + 2;
        uint32_t c1[4] = {(uint32_t)i1, (uint32_t)(i1 >> 32), d.off_lo, d.off_hi};
        mh_philox<7>(c1, d.seed_lo, d.seed_hi);
        kw |= drop_keep_attn_from(c1, thr16) << 16;
      }
      dst[(int64_t)kp << 6] = kw;
    }
  }
}

template <typename T>
__global__ void dropout_bits_apply_kernel(T* __restrict__ P, int64_t ldp, const uint32_t* __restrict__ bits, int64_t BH, int L, int nb,
                                          float rscale) {
  const int64_t total = BH * L * L;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % L);
    const int q = (int)((i / L) % L);
    const int64_t bh = i / ((int64_t)L * L);
    T* p = P + (bh * L + q) * ldp + k;
    *p = drop_flag(bits, bh, nb, q, k) ? from_f32<T>(to_f32(*p) * rscale) : from_f32<T>(0.f);
  }
}

inline int ew_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 65535 * 4 ? 65535 * 4 : b));
}

}  // namespace

extern "C" size_t mh_dropout_bits_words(int BH, int L) {
  const size_t nb = (size_t)((L + 31) / 32);
  return (size_t)BH * nb * ((nb + 1) / 2) * 64;
}

extern "C" int mh_dropout_fwd(const void* x, int64_t ldx, void* out, int64_t ldo, int64_t rows, int cols, int dtype,
                              const mh_dropout* drop, mh_stream_t stream) {
  MH_CHECK_ARG(x && out && rows > 0 && cols > 0 && cols % 8 == 0 && ldx >= cols && ldo >= cols && ldx % 8 == 0 && ldo % 8 == 0,
               "dropout_fwd: bad arguments (cols and leading dimensions must be multiples of 8)");
  DropArgs d;
  int rc = mh_drop_args(drop, &d);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int grid = ew_grid(rows * (cols / 8));
  if (dtype == MH_BF16) MH_LAUNCH((dropout_fwd_kernel<bf16>), dim3(grid), dim3(256), 0, s, (const bf16*)x, ldx, (bf16*)out, ldo, rows, cols, d);
  else if (dtype == MH_F32) MH_LAUNCH((dropout_fwd_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)x, ldx, (float*)out, ldo, rows, cols, d);
  else { mh_set_error("dropout_fwd: unknown dtype %d", dtype); return MH_ERR_INVALID; }
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_dropout_bits(uint32_t* keep_bits, int BH, int L, const mh_dropout* drop, mh_stream_t stream) {
  MH_CHECK_ARG(keep_bits && BH > 0 && L > 0 && drop && drop->p > 0.f && !drop->mask, "dropout_bits: bad arguments");
  DropArgs d;
  int rc = mh_drop_args(drop, &d);
  if (rc) return rc;
  const int nb = (L + 31) / 32, nkp = (nb + 1) / 2;
  const int64_t nrows = (int64_t)BH * nb;
  // a wave walks a run of key-block pairs of one row block: whole rows when there are enough of them to fill the chip, shorter runs otherwise
  int kchunk = nkp;
  while (kchunk > 1 && nrows * ((nkp + kchunk - 1) / kchunk) < 8192) kchunk = (kchunk + 1) / 2;
  const int nchunks = (nkp + kchunk - 1) / kchunk;
  const int64_t nitems = nrows * nchunks;
  MH_CHECK_ARG(nitems < (1ll << 31), "dropout_bits: %lld work items do not fit 31 bits", (long long)nitems);
  const int grid = (int)((nitems + 3) / 4 < 262140 ? (nitems + 3) / 4 : 262140);
  MH_LAUNCH(dropout_bits_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, keep_bits, (int)nitems, nchunks, kchunk, L, nb, nkp, d);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_dropout_bits_apply(void* P, int64_t ldp, const uint32_t* keep_bits, int BH, int L, float p, int dtype,
                                     mh_stream_t stream) {
  MH_CHECK_ARG(P && keep_bits && BH > 0 && L > 0 && ldp >= L && p >= 0.f && p < 1.f, "dropout_bits_apply: bad arguments");
  const int nb = (L + 31) / 32;
  const float rs = 1.0f / (1.0f - p);
  hipStream_t s = (hipStream_t)stream;
  const int grid = ew_grid((int64_t)BH * L * L);
  if (dtype == MH_BF16) MH_LAUNCH((dropout_bits_apply_kernel<bf16>), dim3(grid), dim3(256), 0, s, (bf16*)P, ldp, keep_bits, (int64_t)BH, L, nb, rs);
  else if (dtype == MH_F32) MH_LAUNCH((dropout_bits_apply_kernel<float>), dim3(grid), dim3(256), 0, s, (float*)P, ldp, keep_bits, (int64_t)BH, L, nb, rs);
  else { mh_set_error("dropout_bits_apply: unknown dtype %d", dtype); return MH_ERR_INVALID; }
  MH_CHECK_LAUNCH();
  return MH_OK;
}
