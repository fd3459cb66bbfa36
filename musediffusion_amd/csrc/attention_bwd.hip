// Fused attention backward (bf16 MFMA, gfx950) for training_losses (models/diffusion.py:594-699 -> HF
// BertSelfAttention backward).  Two streaming kernels in the mould of attn_stream_bf16_kernel (attention.hip):
// persistent 8-wave blocks, operands copied into LDS by LDS-DMA in 128-row stages, double-buffered, the score
// matrix never leaves registers.  P is re-created from the forward's log-sum-exp: P[q][k] = exp2(s c - lse2[q]).
//
//   attn_bwd_dq_kernel   (one 32-query tile per wave, keys streamed):
//       S^T = K Q^T,  dP^T = V dO^T,  dS^T = P^T o (dP^T - D[q]),  dQ^T += K^T dS^T
//   attn_bwd_dkv_kernel  (one 32-key tile per wave, queries streamed):
//       S = Q K^T,  dP = dO V^T,  dS = P o (dP - D[q]),  dV^T += dO^T P,  dK^T += Q^T dS
//
// with D[q] = sum_d dO[q][d] O[q][d].  Every product is a v_mfma_f32_32x32x16_bf16 whose B operand is either a
// register-resident fragment of the wave's own tile or the previous product's accumulator converted to bf16 in
// place (its register order is the "P-operand" key order, which is why the transposed operands K^T, Q^T, dO^T
// arrive with the keys of every group of 16 stored as 0-3, 8-11, 4-7, 12-15: mh_head_permute mode 3).
// Recomputing S in both kernels costs 2 extra products but needs neither atomics nor an [L, L] tensor in HBM.
#include "common.h"

namespace {

constexpr int SKB = 128;   // rows (keys or queries) per LDS stage

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// piece p (1 KiB) of a row-layout operand stage: rows [row][DH] bf16, 16-B chunk c of row r stored at c ^ ((r / RPB) & (CH-1))
template <int DH>
__device__ __forceinline__ void dma_rows(const bf16* src, int64_t ld, char* dst, int p, int lane, int valid_rows) {
  constexpr int CH = DH / 8, RPB = 128 / DH, KROWB = DH * 2, KRP = 1024 / KROWB;
  const int row = p * KRP + lane / CH, pc = lane % CH;
  const int lc = pc ^ ((row / RPB) & (CH - 1));
  const int rsrc = row < valid_rows ? row : valid_rows - 1;     // rows past the sequence end: a valid row (their terms are masked)
  glds16(src + (int64_t)rsrc * ld + lc * 8, dst + p * 1024);
}
// where the [L, dh] rows of one (batch, head) live: [B, nh, L, dh] tensors or column blocks of token-major [B L, ld] ones
struct RowLayout {
  int64_t sB, sH, ld;   // batch stride, head stride, row pitch (elements)
  __device__ __forceinline__ int64_t at(int bh, int nh, int64_t row) const { return (int64_t)(bh / nh) * sB + (int64_t)(bh % nh) * sH + row * ld; }
};
// piece p of a transposed operand stage: global [DH][L] (key order permuted per 16), LDS = 64-column tiles of [DH][128 B],
// chunk c of row d stored at c ^ ((d >> 1) & 7)
template <int DH>
__device__ __forceinline__ void dma_cols(const bf16* src, int64_t L, char* dst, int p, int lane, int valid_cols) {
  const int t = p / (DH / 8), d = (p % (DH / 8)) * 8 + (lane >> 3), pc = lane & 7;
  const int lc = pc ^ ((d >> 1) & 7);
  int c = t * 64 + lc * 8;
  if (c >= valid_cols) c = valid_cols - 8;                      // columns past the end: a valid chunk (finite values x 0)
  glds16(src + (int64_t)d * L + c, dst + p * 1024);
}
// A-operand fragment (32 rows x 16 k) of a row-layout stage: stage-relative row R, k-step ks, lane half h
template <int DH>
__device__ __forceinline__ bf16x8 frag_rows(const char* stage, int R, int ks, int h) {
  constexpr int CH = DH / 8, RPB = 128 / DH, KROWB = DH * 2;
  return *reinterpret_cast<const bf16x8*>(stage + R * KROWB + (((2 * ks + h) ^ ((R / RPB) & (CH - 1))) << 4));
}
// A-operand fragment (32 d rows x 16 columns) of a transposed stage: 64-column tile t, 16-column slab sl (0..3), row d
template <int DH>
__device__ __forceinline__ bf16x8 frag_cols(const char* stage, int t, int sl, int d, int h) {
  return *reinterpret_cast<const bf16x8*>(stage + t * (DH * 128) + d * 128 + (((2 * sl + h) ^ ((d >> 1) & 7)) << 4));
}

__device__ __forceinline__ bf16x8 cvt8(const f32x16& v, int off) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (bf16)v[off + j];
  return r;
}

// ---------------------------------------------------------------------------------------------------------
// FULL (both kernels): seq_len % 256 == 0 - no partial block, stage or tile, the per-score bound compares are compiled out
template <int DH, bool DROP, bool FULL>
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                          const bf16* __restrict__ V, const bf16* __restrict__ KT,
                                                          const bf16* __restrict__ dO, const bf16* __restrict__ O,
                                                          const float* __restrict__ lse2, float* __restrict__ Dv,
                                                          bf16* __restrict__ dQ, int64_t ld_dq,
                                                          int L, int nh, int nbh, float scale, float scale_log2e, RowLayout lq_,
                                                          RowLayout lo_, const uint32_t* __restrict__ keep_bits, float rscale) {
  constexpr int NW = 8;
  constexpr int KS = DH / 16, DT = DH / 32;
  constexpr int ST = SKB * DH * 2;                 // bytes of one operand stage (row layout and transposed alike)
  constexpr int PK = ST / 1024 / NW;               // DMA pieces per wave per operand per stage
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, lq = lane & 31;
  const int nqb = (L + 255) / 256, nst = (L + SKB - 1) / SKB;     // last query block / key stage may be partial (L % 16 == 0)
  const int nitems = nbh * nqb;
  const int my_items = (nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_items * nst;

  auto issue = [&](int g) {
    const int item = blockIdx.x + (g / nst) * gridDim.x, st = g % nst;
    const int bh = item / nqb;
    const int64_t roff = lq_.at(bh, nh, (int64_t)st * SKB);
    const bf16* Tb = KT + (int64_t)bh * DH * L + (int64_t)st * SKB;
    char* base = smem_dyn + (g & 1) * (3 * ST);
    const int valid = L - st * SKB;                       // rows of this stage that exist (the rest: clamped sources, masked scores)
#pragma unroll
    for (int j = 0; j < PK; ++j) dma_rows<DH>(K + roff, lq_.ld, base, wave + NW * j, lane, valid);
#pragma unroll
    for (int j = 0; j < PK; ++j) dma_rows<DH>(V + roff, lq_.ld, base + ST, wave + NW * j, lane, valid);
#pragma unroll
    for (int j = 0; j < PK; ++j) dma_cols<DH>(Tb, L, base + 2 * ST, wave + NW * j, lane, valid);
    if constexpr (DROP && FULL) {   // this wave's keep words of the stage's two 64-key tiles (2 x 64 words, contiguous): 512 B, lanes 0..31
      const int qbw = (item % nqb) * 8 + wave;
      if (lane < 32)
        glds16(keep_bits + drop_word_index(bh, (L + 31) >> 5, qbw, st * (SKB / 64), 0) + 4 * lane, smem_dyn + 2 * 3 * ST + (g & 1) * 4096 + wave * 512);
    }
  };

  bf16x8 qf[KS], dof[KS];
  f32x16 dq[DT];
  float lse_q = 0.f, D_q = 0.f;
  int q0 = 0;
  bool active = false;
  if (total > 0) issue(0);
  for (int g = 0; g < total; ++g) {
    const int item = blockIdx.x + (g / nst) * gridDim.x, st = g % nst;
    const int bh = item / nqb, qb = item % nqb;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (g + 1 < total) issue(g + 1);
    if (st == 0) {
      q0 = qb * 256 + wave * 32;
      active = q0 < L;
      const int qc = q0 + lq < L ? q0 + lq : L - 1;                 // clamped row of a partial / absent query tile
      const int64_t qrow = (int64_t)bh * L + qc;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = *reinterpret_cast<const bf16x8*>(Q + lq_.at(bh, nh, qc) + 16 * ks + 8 * h);
        dof[ks] = *reinterpret_cast<const bf16x8*>(dO + lo_.at(bh, nh, qc) + 16 * ks + 8 * h);
      }
      lse_q = lse2[qrow];
      // D[q] = sum_d dO[q][d] O[q][d]: each lane half holds half of the head dim of its query; kept for the dK/dV kernel
      D_q = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 of = *reinterpret_cast<const bf16x8*>(O + lo_.at(bh, nh, qc) + 16 * ks + 8 * h);
#pragma unroll
        for (int j = 0; j < 8; ++j) D_q += (float)dof[ks][j] * (float)of[j];
      }
      D_q += __shfl_xor(D_q, 32, 64);
      if (h == 0 && q0 + lq < L) Dv[qrow] = D_q;
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;
    }
    const char* kst = smem_dyn + (g & 1) * (3 * ST);
    const char* vst = kst + ST;
    const char* tst = kst + 2 * ST;
    const int st_keys = L - st * SKB;
    for (int t = 0; active && t < SKB / 64 && (FULL || t * 64 < st_keys); ++t) {
      const int tile_keys = FULL ? 64 : st_keys - t * 64;
      f32x16 s[2], dp[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const int R = t * 64 + 32 * kt + lq;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[kt][r] = 0.f; dp[kt][r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<DH>(kst, R, ks, h), qf[ks], s[kt], 0, 0, 0);
          dp[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<DH>(vst, R, ks, h), dof[ks], dp[kt], 0, 0, 0);
        }
      }
      uint32_t kw = 0xffffffffu;
      if constexpr (DROP && FULL) {   // ... staged in LDS with the operands (a global load here would wait for the next stage's DMA)
        kw = *reinterpret_cast<const uint32_t*>(smem_dyn + 2 * 3 * ST + (g & 1) * 4096 + wave * 512 + t * 256 + lane * 4);
      } else if constexpr (DROP) {   // dP = dP_drop o keep / (1 - p): this lane's word of the 64-key tile, as the forward stored it
        const int nb32 = (L + 31) >> 5;
        kw = keep_bits[drop_word_index(bh, nb32, q0 >> 5, (st * SKB + t * 64) >> 6, lane)];
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const uint32_t km = kw >> (16 * kt);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float p = __builtin_amdgcn_exp2f(s[kt][r] * scale_log2e - lse_q);
          if (!FULL && tile_keys < 64 && kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h >= tile_keys) p = 0.f;   // key past the sequence end
          float dpv = dp[kt][r];
          if constexpr (DROP) dpv = (km >> r) & 1u ? dpv * rscale : 0.f;
          s[kt][r] = p * (dpv - D_q);                            // dS^T (the 1/sqrt(dh) factor is applied to dQ at the end)
        }
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 dsf = cvt8(s[kt], 8 * s2);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
            dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<DH>(tst, t, 2 * kt + s2, dt * 32 + lq, h), dsf, dq[dt], 0, 0, 0);
        }
    }
    if (st == nst - 1 && q0 + lq < L) {
      const int b = bh / nh, head = bh % nh;
      const int64_t tok = (int64_t)b * L + q0 + lq;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        bf16* dst = dQ + tok * ld_dq + head * DH + dt * 32;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          bf16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (bf16)(dq[dt][rg * 4 + e] * scale);
          *reinterpret_cast<bf16x4*>(dst + 8 * rg + 4 * h) = v;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
}

// ---------------------------------------------------------------------------------------------------------
template <int DH, bool DROP, bool FULL>
__global__ __launch_bounds__(512) void attn_bwd_dkv_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                           const bf16* __restrict__ V, const bf16* __restrict__ QT,
                                                           const bf16* __restrict__ dO, const bf16* __restrict__ dOT,
                                                           const float* __restrict__ lse2, const float* __restrict__ Dv,
                                                           bf16* __restrict__ dK, bf16* __restrict__ dV, int64_t ld_d, int L,
                                                           int nh, int nbh, float scale, float scale_log2e, RowLayout lq_,
                                                           RowLayout lo_, const uint32_t* __restrict__ keep_bits, float rscale) {
  constexpr int NW = 8;
  constexpr int KS = DH / 16, DT = DH / 32;
  constexpr int ST = SKB * DH * 2;
  constexpr int PK = ST / 1024 / NW;
  constexpr int BUF = 4 * ST + 1024 + (DROP && FULL ? 4096 : 0);   // Q rows, dO rows, Q^T, dO^T, (lse2 | D) of the stage's 128 queries, keep words
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, lq = lane & 31;
  const int nkb = (L + 255) / 256, nst = (L + SKB - 1) / SKB;
  const int nitems = nbh * nkb;
  const int my_items = (nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_items * nst;

  auto issue = [&](int g) {
    const int item = blockIdx.x + (g / nst) * gridDim.x, st = g % nst;
    const int bh = item / nkb;
    const int64_t r0 = (int64_t)bh * L + (int64_t)st * SKB;
    const int64_t c0 = (int64_t)bh * DH * L + (int64_t)st * SKB;
    char* base = smem_dyn + (g & 1) * BUF;
    const int valid = L - st * SKB;
#pragma unroll
    for (int j = 0; j < PK; ++j) dma_rows<DH>(Q + lq_.at(bh, nh, (int64_t)st * SKB), lq_.ld, base, wave + NW * j, lane, valid);
#pragma unroll
    for (int j = 0; j < PK; ++j) dma_rows<DH>(dO + lo_.at(bh, nh, (int64_t)st * SKB), lo_.ld, base + ST, wave + NW * j, lane, valid);
#pragma unroll
    for (int j = 0; j < PK; ++j) dma_cols<DH>(QT + c0, L, base + 2 * ST, wave + NW * j, lane, valid);
#pragma unroll
    for (int j = 0; j < PK; ++j) dma_cols<DH>(dOT + c0, L, base + 3 * ST, wave + NW * j, lane, valid);
    if (wave == 0) {   // lanes 0-31: lse2 of the 128 queries, lanes 32-63: D (groups of 4; past the end: the last valid group)
      int qo = 4 * (lane & 31);
      if (qo >= valid) qo = valid - 4;
      glds16((lane < 32 ? lse2 : Dv) + r0 + qo, base + 4 * ST);
    }
    if constexpr (DROP && FULL) {   // keep words of (4 query blocks of the stage) x (the block's 4 pairs of key blocks): 1 KiB per query block
      if (wave >= 4)
        glds16(keep_bits + drop_word_index(bh, (L + 31) >> 5, st * (SKB / 32) + wave - 4, (item % nkb) * 4, 0) + 4 * lane,
               base + 4 * ST + 1024 + (wave - 4) * 1024);
    }
  };

  bf16x8 kf[KS], vf[KS];
  f32x16 dk[DT], dv[DT];
  int k0 = 0;
  bool active = false;
  if (total > 0) issue(0);
  for (int g = 0; g < total; ++g) {
    const int item = blockIdx.x + (g / nst) * gridDim.x, st = g % nst;
    const int bh = item / nkb, kb = item % nkb;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (g + 1 < total) issue(g + 1);
    if (st == 0) {
      k0 = kb * 256 + wave * 32;
      active = k0 < L;
      const int kc = k0 + lq < L ? k0 + lq : L - 1;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = *reinterpret_cast<const bf16x8*>(K + lq_.at(bh, nh, kc) + 16 * ks + 8 * h);
        vf[ks] = *reinterpret_cast<const bf16x8*>(V + lq_.at(bh, nh, kc) + 16 * ks + 8 * h);
      }
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }
    }
    const char* qst = smem_dyn + (g & 1) * BUF;
    const char* ost = qst + ST;
    const char* qtst = qst + 2 * ST;
    const char* otst = qst + 3 * ST;
    const float* lst = reinterpret_cast<const float*>(qst + 4 * ST);   // [0..127] lse2, [128..255] D
    const int st_q = L - st * SKB;                                      // queries of this stage that exist
    for (int t = 0; active && t < SKB / 64 && (FULL || t * 64 < st_q); ++t) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const int R = t * 64 + 32 * qt + lq;
        f32x16 s, dp;
        // keep flags of this lane's key for its 16 queries of the tile: query 8 rg + 4 h + e lives in the word of forward lane
        // (query % 32) + 32 hk, at bit rk (+ 16 for an odd key block): four consecutive words per rg
        uint4 kwv[4];
        int kshift = 0;
        if constexpr (DROP) {
          const int nb32 = (L + 31) >> 5;
          int qb = (st * SKB + t * 64 + 32 * qt) >> 5; if (qb >= nb32) qb = nb32 - 1;
          const int kbk = k0 >> 5, hk = (lq >> 2) & 1;
          kshift = (lq & 3) + 4 * (lq >> 3) + 16 * (kbk & 1);
          if constexpr (FULL) {   // staged in LDS with the operands
            const uint32_t* wl = reinterpret_cast<const uint32_t*>(qst + 4 * ST + 1024) + (2 * t + qt) * 256 + ((wave >> 1) & 3) * 64 + 32 * hk + 4 * h;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) kwv[rg] = *reinterpret_cast<const uint4*>(wl + 8 * rg);
          } else {
            const uint32_t* wp = keep_bits + drop_word_index(bh, nb32, qb, kbk >> 1, 32 * hk + 4 * h);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) kwv[rg] = *reinterpret_cast<const uint4*>(wp + 8 * rg);
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<DH>(qst, R, ks, h), kf[ks], s, 0, 0, 0);      // S[query][key]
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<DH>(ost, R, ks, h), vf[ks], dp, 0, 0, 0);    // dP[query][key]
        }
        // register r <-> query (r & 3) + 8 (r >> 2) + 4 h of this 32-query tile: four runs of four consecutive queries
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int qi = t * 64 + 32 * qt + 8 * rg + 4 * h;
          const f32x4 ls = *reinterpret_cast<const f32x4*>(lst + qi);
          const f32x4 dd = *reinterpret_cast<const f32x4*>(lst + 128 + qi);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float p = __builtin_amdgcn_exp2f(s[rg * 4 + e] * scale_log2e - ls[e]);
            if (!FULL && qi + e >= st_q) p = 0.f;                                  // query past the sequence end
            float dpv = dp[rg * 4 + e], pd = p;
            if constexpr (DROP) {                                         // P_drop feeds dV; dP = dP_drop o keep / (1 - p)
              const uint32_t kwe = e == 0 ? kwv[rg].x : e == 1 ? kwv[rg].y : e == 2 ? kwv[rg].z : kwv[rg].w;
              const bool keep = (kwe >> kshift) & 1u;
              pd = keep ? p * rscale : 0.f;
              dpv = keep ? dpv * rscale : 0.f;
            }
            s[rg * 4 + e] = pd;
            dp[rg * 4 + e] = p * (dpv - dd[e]);
          }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pf = cvt8(s, 8 * s2), dsf = cvt8(dp, 8 * s2);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<DH>(otst, t, 2 * qt + s2, dt * 32 + lq, h), pf, dv[dt], 0, 0, 0);
            dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<DH>(qtst, t, 2 * qt + s2, dt * 32 + lq, h), dsf, dk[dt], 0, 0, 0);
          }
        }
      }
    }
    if (st == nst - 1 && k0 + lq < L) {
      const int b = bh / nh, head = bh % nh;
      const int64_t tok = (int64_t)b * L + k0 + lq;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        bf16* dstk = dK + tok * ld_d + head * DH + dt * 32;
        bf16* dstv = dV + tok * ld_d + head * DH + dt * 32;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          bf16x4 a, c;
#pragma unroll
          for (int e = 0; e < 4; ++e) { a[e] = (bf16)(dk[dt][rg * 4 + e] * scale); c[e] = (bf16)dv[dt][rg * 4 + e]; }
          *reinterpret_cast<bf16x4*>(dstk + 8 * rg + 4 * h) = a;
          *reinterpret_cast<bf16x4*>(dstv + 8 * rg + 4 * h) = c;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
}

// D[b, h, l] = sum_d dctx[tok][h dh + d] * ctx[tok][h dh + d]   (token-major inputs, one wave per 64 / dh ... simple: thread per (tok, head))
__global__ void attn_bwd_rowdot_kernel(const bf16* __restrict__ dctx, const bf16* __restrict__ ctx, int64_t ld, float* __restrict__ Dv,
                                       int B, int L, int nh, int dh) {
  const int64_t total = (int64_t)B * L * nh;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int head = (int)(i % nh);
    const int64_t tok = i / nh;
    const int64_t b = tok / L, l = tok % L;
    const bf16* a = dctx + tok * ld + head * dh;
    const bf16* c = ctx + tok * ld + head * dh;
    float s = 0.f;
    for (int d = 0; d < dh; d += 8) {
      float x[8], y[8];
      load8(a + d, x);
      load8(c + d, y);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += x[e] * y[e];
    }
    Dv[(b * nh + head) * L + l] = s;
  }
}

template <int DH, bool DROP, bool FULL>
int launch_bwd(const bf16* q, const bf16* k, const bf16* v, const bf16* qT, const bf16* kT, const bf16* dO, const bf16* dOT, const bf16* o,
               const float* lse2, float* Dv, bf16* dq, bf16* dk, bf16* dv, int64_t ld, int B, int L, int nh, float scale,
               RowLayout lqkv, RowLayout ldo, const uint32_t* keep_bits, float rscale, hipStream_t s) {
  constexpr int ST = SKB * DH * 2;
  constexpr int bytes_dq = 2 * 3 * ST + (DROP && FULL ? 2 * 4096 : 0), bytes_dkv = 2 * (4 * ST + 1024 + (DROP && FULL ? 4096 : 0));
  static bool attr_set = false;
  if (!attr_set) {
    MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<DH, DROP, FULL>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes_dq));
    MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<DH, DROP, FULL>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes_dkv));
    attr_set = true;
  }
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  const int nbh = B * nh, nitems = nbh * ((L + 255) / 256);
  const dim3 grid((unsigned)(nitems < cus ? nitems : cus)), block(512);
  const float sl2 = scale * 1.4426950408889634f;
  mh_prof_note("attn_bwd B*nh=%d L=%d dh=%d", nbh, L, DH);
  MH_LAUNCH((attn_bwd_dq_kernel<DH, DROP, FULL>), grid, block, bytes_dq, s, q, k, v, kT, dO, o, lse2, Dv, dq, ld, L, nh, nbh, scale, sl2, lqkv, ldo,
            keep_bits, rscale);
  MH_CHECK_LAUNCH();
  mh_prof_note("attn_bwd B*nh=%d L=%d dh=%d", nbh, L, DH);
  MH_LAUNCH((attn_bwd_dkv_kernel<DH, DROP, FULL>), grid, block, bytes_dkv, s, q, k, v, qT, dO, dOT, lse2, Dv, dk, dv, ld, L, nh, nbh, scale, sl2, lqkv, ldo,
            keep_bits, rscale);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

}  // namespace

extern "C" int mh_attention_stream_bwd_supported(int L, int dh) { return L >= 512 && L % 16 == 0 && (dh == 32 || dh == 64); }

extern "C" int mh_attention_bwd_rowdot(const void* dctx, const void* ctx, int64_t ld, float* D, int B, int L, int nh, int dh,
                                       mh_stream_t stream) {
  MH_CHECK_ARG(dctx && ctx && D && B > 0 && L > 0 && nh > 0 && dh > 0 && dh % 8 == 0 && ld % 8 == 0, "attention_bwd_rowdot: bad arguments");
  const int64_t total = (int64_t)B * L * nh;
  const int grid = (int)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
  MH_LAUNCH(attn_bwd_rowdot_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)dctx, (const bf16*)ctx, ld, D, B, L, nh, dh);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_attention_stream_bwd_ex(const void* q, const void* k, const void* v, const void* qT_perm, const void* kT_perm,
                                          const void* dO, const void* dOT_perm, const void* o, const float* lse2, float* D, void* dq, void* dk,
                                          void* dv, int64_t ld_d, int B, int L, int nh, int dh, float scale, int64_t qkv_batch_stride,
                                          int64_t qkv_head_stride, int64_t qkv_row_stride, int64_t do_batch_stride,
                                          int64_t do_head_stride, int64_t do_row_stride, mh_stream_t stream);

extern "C" int mh_attention_stream_bwd(const void* q, const void* k, const void* v, const void* qT_perm, const void* kT_perm,
                                       const void* dO, const void* dOT_perm, const void* o, const float* lse2, float* D, void* dq, void* dk,
                                       void* dv, int64_t ld_d, int B, int L, int nh, int dh, float scale, mh_stream_t stream) {
  const int64_t sB = (int64_t)nh * L * dh, sH = (int64_t)L * dh;
  return mh_attention_stream_bwd_ex(q, k, v, qT_perm, kT_perm, dO, dOT_perm, o, lse2, D, dq, dk, dv, ld_d, B, L, nh, dh, scale, sB, sH, dh,
                                    sB, sH, dh, stream);
}

extern "C" int mh_attention_stream_bwd_drop(const void* q, const void* k, const void* v, const void* qT_perm, const void* kT_perm,
                                            const void* dO, const void* dOT_perm, const void* o, const float* lse2, float* D, void* dq, void* dk,
                                            void* dv, int64_t ld_d, int B, int L, int nh, int dh, float scale, int64_t qsB, int64_t qsH,
                                            int64_t qld, int64_t osB, int64_t osH, int64_t old_, const uint32_t* keep_bits, float drop_p,
                                            mh_stream_t stream);

extern "C" int mh_attention_stream_bwd_ex(const void* q, const void* k, const void* v, const void* qT_perm, const void* kT_perm,
                                          const void* dO, const void* dOT_perm, const void* o, const float* lse2, float* D, void* dq, void* dk,
                                          void* dv, int64_t ld_d, int B, int L, int nh, int dh, float scale, int64_t qsB, int64_t qsH,
                                          int64_t qld, int64_t osB, int64_t osH, int64_t old_, mh_stream_t stream) {
  return mh_attention_stream_bwd_drop(q, k, v, qT_perm, kT_perm, dO, dOT_perm, o, lse2, D, dq, dk, dv, ld_d, B, L, nh, dh, scale, qsB, qsH, qld,
                                      osB, osH, old_, nullptr, 0.f, stream);
}

// Backward of the streaming attention whose forward dropped probabilities with rate drop_p: `keep_bits` is the bit tensor that
// forward wrote (or was given); null / drop_p == 0: no dropout.
extern "C" int mh_attention_stream_bwd_drop(const void* q, const void* k, const void* v, const void* qT_perm, const void* kT_perm,
                                            const void* dO, const void* dOT_perm, const void* o, const float* lse2, float* D, void* dq, void* dk,
                                            void* dv, int64_t ld_d, int B, int L, int nh, int dh, float scale, int64_t qsB, int64_t qsH,
                                            int64_t qld, int64_t osB, int64_t osH, int64_t old_, const uint32_t* keep_bits, float drop_p,
                                            mh_stream_t stream) {
  MH_CHECK_ARG(qsB % 8 == 0 && qsH % 8 == 0 && qld % 8 == 0 && osB % 8 == 0 && osH % 8 == 0 && old_ % 8 == 0 && qld >= dh && old_ >= dh,
               "attention_stream_bwd: row strides must be multiples of 8 elements");
  const RowLayout lqkv{qsB, qsH, qld}, ldo{osB, osH, old_};
  MH_CHECK_ARG(q && k && v && qT_perm && kT_perm && dO && dOT_perm && o && lse2 && D && dq && dk && dv, "attention_stream_bwd: null pointer");
  MH_CHECK_ARG(B > 0 && nh > 0 && mh_attention_stream_bwd_supported(L, dh),
               "attention_stream_bwd: needs seq_len %% 16 == 0, seq_len >= 512 and head dim 32 or 64 (got L=%d dh=%d)", L, dh);
  MH_CHECK_ARG(ld_d % 4 == 0, "attention_stream_bwd: ld_d must be a multiple of 4");
  MH_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || keep_bits), "attention_stream_bwd: dropout needs keep_bits and p in [0, 1)");
  hipStream_t s = (hipStream_t)stream;
  const float rs = 1.0f / (1.0f - drop_p);
#define MH_BWD_ARGS (const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)qT_perm, (const bf16*)kT_perm, (const bf16*)dO, \
                    (const bf16*)dOT_perm, (const bf16*)o, lse2, D, (bf16*)dq, (bf16*)dk, (bf16*)dv, ld_d, B, L, nh, scale, lqkv, ldo, keep_bits, rs, s
  if (L % 256 == 0 && mh_attention_stream_enabled() != 4) {   // (mode 4 = A/B: the bound-checking build on every length)
    if (drop_p > 0.f) return dh == 64 ? launch_bwd<64, true, true>(MH_BWD_ARGS) : launch_bwd<32, true, true>(MH_BWD_ARGS);
    return dh == 64 ? launch_bwd<64, false, true>(MH_BWD_ARGS) : launch_bwd<32, false, true>(MH_BWD_ARGS);
  }
  if (drop_p > 0.f) return dh == 64 ? launch_bwd<64, true, false>(MH_BWD_ARGS) : launch_bwd<32, true, false>(MH_BWD_ARGS);
  return dh == 64 ? launch_bwd<64, false, false>(MH_BWD_ARGS) : launch_bwd<32, false, false>(MH_BWD_ARGS);
#undef MH_BWD_ARGS
}
