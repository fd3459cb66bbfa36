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
// place; its register order is the "P-operand" row order (element j of lane half h <-> row 8 (j >> 2) + 4 h + (j & 3) of
// a 16-row slab).  The transposed operands K^T, Q^T, dO^T are NOT separate tensors: the row-layout LDS stages that feed
// S and dP are read a second time with the transposing ds_read_b64_tr_b16 (cdna_hip_programming.md T10), whose four
// rows per 16-lane group are addressed in exactly that order - no transposed copies in HBM, half the LDS-DMA traffic.
// One swizzle serves both kinds of read (row_swz).  The fragments of a 32-row sub-tile are read ahead of the MFMAs that
// use them: the next sub-tile's row fragments and this one's transposed fragments land under the exp / dS vector work.
// Recomputing S in both kernels costs 2 extra products but needs neither atomics nor an [L, L] tensor in HBM.
#include "common.h"

namespace {

#ifndef MH_BWD_SKB
#define MH_BWD_SKB 128
#endif
constexpr int SKB = MH_BWD_SKB;   // rows (keys or queries) per LDS stage: 128 or 256 (build-time A/B: -DMH_BWD_SKB=256)
static_assert(SKB == 128 || SKB == 256, "stage of 128 or 256 rows");

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// 16-byte chunk c of row r of a row-layout stage ([row][DH] bf16) is stored at chunk c ^ row_swz(r).
// dh 64 (128-byte rows, two per 256 bytes of banks): a rotation of (r >> 1) & 7 - any bijection of those three bits keeps the
// ds_read_b128 row reads conflict-free (the 16 rows of a lane group then fall on 16 different 16-byte bank slots); the transposing
// read of a 32-lane half takes 4 consecutive rows x 4 chunks, rows r and r + 2 share their 32 banks, so row bit 1 must move the
// chunk by 4: bit 0 of (r >> 1) goes to bit 2.  dh 32 (64-byte rows, four per 256 bytes): 4 rows x 4 chunks are 16 slots anyway.
template <int DH>
__device__ __forceinline__ int row_swz(int row) {
  if constexpr (DH == 64) { const int v = (row >> 1) & 7; return ((v & 1) << 2) | (v >> 1); }
  else return (row / (128 / DH)) & (DH / 8 - 1);
}
// piece p (1 KiB) of a row-layout operand stage
template <int DH>
__device__ __forceinline__ void dma_rows(const bf16* src, int64_t ld, char* dst, int p, int lane, int valid_rows) {
  constexpr int CH = DH / 8, KROWB = DH * 2, KRP = 1024 / KROWB;
  const int row = p * KRP + lane / CH, pc = lane % CH;
  const int lc = pc ^ row_swz<DH>(row);
  const int rsrc = row < valid_rows ? row : valid_rows - 1;     // rows past the sequence end: a valid row (their terms are masked)
  glds16(src + (int64_t)rsrc * ld + lc * 8, dst + p * 1024);
}
// where the [L, dh] rows of one (batch, head) live: [B, nh, L, dh] tensors or column blocks of token-major [B L, ld] ones
struct RowLayout {
  int64_t sB, sH, ld;   // batch stride, head stride, row pitch (elements)
  __device__ __forceinline__ int64_t at(int bh, int nh, int64_t row) const { return (int64_t)(bh / nh) * sB + (int64_t)(bh % nh) * sH + row * ld; }
};
// A-operand fragment (32 rows x 16 k) of a row-layout stage: stage-relative row R, k-step ks, lane half h
template <int DH>
__device__ __forceinline__ bf16x8 frag_rows(const char* stage, int R, int ks, int h) {
  return *reinterpret_cast<const bf16x8*>(stage + R * (DH * 2) + (((2 * ks + h) ^ row_swz<DH>(R)) << 4));
}
// A-operand fragment (32 d x 16 k) of the TRANSPOSE of a row-layout stage: k = the 16 rows from Rb (a multiple of 16) in P-operand
// order, d = 32 dt + lane % 32.  Two transposing reads; lane 4 q + p of a 16-lane group (group = lane half h x column half c16)
// supplies the address of row 4 h + q (+ 8 for the second read), columns 32 dt + 16 c16 + 4 p ... + 3, and receives column
// 32 dt + 16 c16 + (its index in the group) of the group's four rows.  The per-lane offsets do not depend on Rb (row_swz has the
// period 16): tr_offsets computes them once per kernel.
typedef __attribute__((ext_vector_type(4))) short s16x4;
template <int DH> struct TrOff { int o[DH / 32][2]; };
template <int DH>
__device__ __forceinline__ TrOff<DH> tr_offsets(int lane) {
  TrOff<DH> t;
  const int i = lane & 15, q = i >> 2, p = i & 3, c16 = (lane >> 4) & 1, h = lane >> 5;
#pragma unroll
  for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int r = 4 * h + q + 8 * half, chunk = 4 * dt + 2 * c16 + (p >> 1);
      t.o[dt][half] = r * (DH * 2) + ((chunk ^ row_swz<DH>(r)) << 4) + 8 * (p & 1);
    }
  return t;
}
template <int DH>
__device__ __forceinline__ bf16x8 frag_tr(const char* stage, int Rb, const TrOff<DH>& to, int dt) {
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  const char* b = stage + Rb * (DH * 2);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(b + to.o[dt][0]));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(b + to.o[dt][1]));
  bf16x8 r;
  __builtin_memcpy(&r, &lo, 8);
  __builtin_memcpy(reinterpret_cast<char*>(&r) + 8, &hi, 8);
  return r;
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
                                                          const bf16* __restrict__ V,
                                                          const bf16* __restrict__ dO, const bf16* __restrict__ O,
                                                          const float* __restrict__ lse2, float* __restrict__ Dv,
                                                          bf16* __restrict__ dQ, int64_t ld_dq,
                                                          int L, int nh, int nbh, float scale, float scale_log2e, RowLayout lq_,
                                                          RowLayout lo_, const uint32_t* __restrict__ keep_bits, float rscale,
                                                          int o_panel, int64_t o_ld, int d_panel) {
  // o_panel: O (the forward's context rows) as K32 panels [H / 32][o_ld rows][32] of the token-major [B L, H] tensor; d_panel: dQ likewise
  // ([.][ld_dq rows][32]) - the layouts the training step's GEMMs read (csrc/train_layer.hip); dO stays in lo_'s row layout
  constexpr int NW = 8;
  constexpr int KS = DH / 16, DT = DH / 32;
  constexpr int ST = SKB * DH * 2;                 // bytes of one operand stage
  constexpr int PK = ST / 1024 / NW;               // DMA pieces per wave per operand per stage
  constexpr int NSUB = SKB / 32;                   // 32-key sub-tiles per stage
  constexpr int KWW = (SKB / 64) * 256, KWB = NW * KWW;   // keep words: bytes per wave and stage (one 256-byte row of words per 64-key tile), per buffer
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, lq = lane & 31;
  const int nqb = (L + 255) / 256, nst = (L + SKB - 1) / SKB;     // last query block / key stage may be partial (L % 16 == 0)
  const int nitems = nbh * nqb;
  // items (batch-head, query block) in XCD-aware order (round 6): the query blocks of one (batch, head) stream the same K / V rows; dealt
  // out by blockIdx they sat on different XCDs and every L2 fetched those rows again (FETCH_SIZE 394 MB per launch, 134 MB of K / V / dO / O)
  const int bx = mh_xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int my_items = (nitems - bx + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_items * nst;
  const TrOff<DH> tro = tr_offsets<DH>(lane);

  auto issue = [&](int g) {
    const int item = bx + (g / nst) * gridDim.x, st = g % nst;
    const int bh = item / nqb;
    const int64_t roff = lq_.at(bh, nh, (int64_t)st * SKB);
    char* base = smem_dyn + (g & 1) * (2 * ST);
    const int valid = L - st * SKB;                       // rows of this stage that exist (the rest: clamped sources, masked scores)
#pragma unroll
    for (int j = 0; j < PK; ++j) dma_rows<DH>(K + roff, lq_.ld, base, wave + NW * j, lane, valid);
#pragma unroll
    for (int j = 0; j < PK; ++j) dma_rows<DH>(V + roff, lq_.ld, base + ST, wave + NW * j, lane, valid);
    if constexpr (DROP && FULL) {   // this wave's keep words of the stage's two 64-key tiles (2 x 64 words, contiguous): 512 B, lanes 0..31
      const int qbw = (item % nqb) * 8 + wave;
      if (lane < KWW / 16)
        glds16(keep_bits + drop_word_index(bh, (L + 31) >> 5, qbw, st * (SKB / 64), 0) + 4 * lane, smem_dyn + 2 * 2 * ST + (g & 1) * KWB + wave * KWW);
    }
  };

#ifdef MH_BWD_PROF
  unsigned long long prof_acc[4] = {0, 0, 0, 0};
#endif
  bf16x8 qf[KS], dof[KS];
  f32x16 dq[DT];
  float lse_q = 0.f, D_q = 0.f;
  int q0 = 0;
  bool active = false;
  if (total > 0) issue(0);
  for (int g = 0; g < total; ++g) {
    const int item = bx + (g / nst) * gridDim.x, st = g % nst;
    const int bh = item / nqb, qb = item % nqb;
#ifdef MH_BWD_PROF
    const unsigned long long tp0 = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef MH_BWD_PROF
    const unsigned long long tp1 = __builtin_amdgcn_s_memtime();
#endif
    __builtin_amdgcn_s_barrier();
#ifdef MH_BWD_PROF
    const unsigned long long tp2 = __builtin_amdgcn_s_memtime();
#endif
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
        const bf16* op = o_panel ? O + ((int64_t)((bh % nh) * DT + (ks >> 1)) * o_ld + (int64_t)(bh / nh) * L + qc) * 32 + 16 * (ks & 1) + 8 * h
                                 : O + lo_.at(bh, nh, qc) + 16 * ks + 8 * h;
        const bf16x8 of = *reinterpret_cast<const bf16x8*>(op);
#pragma unroll
        for (int j = 0; j < 8; ++j) D_q += (float)dof[ks][j] * (float)of[j];
      }
      D_q += __shfl_xor(D_q, 32, 64);
      if constexpr (DROP) {   // dS = p (dP keep / (1 - p_drop) - D) = [p / (1 - p_drop)] (dP keep - D (1 - p_drop)): the factor goes into the exponent,
        lse_q -= __builtin_amdgcn_logf(rscale);   // D is kept pre-multiplied (for the dK/dV kernel too) - nothing per element
        D_q *= 1.0f / rscale;
      }
      if (h == 0 && q0 + lq < L) Dv[qrow] = D_q;
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;
    }
    const char* kst = smem_dyn + (g & 1) * (2 * ST);
    const char* vst = kst + ST;
    const int st_keys = L - st * SKB;
    if (active) {
      bf16x8 fk[KS], fv[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) { fk[ks] = frag_rows<DH>(kst, lq, ks, h); fv[ks] = frag_rows<DH>(vst, lq, ks, h); }
      uint32_t kw = 0xffffffffu;
#pragma unroll
      for (int i = 0; i < NSUB; ++i) {
        if (!FULL && i * 32 >= st_keys) break;                     // (wave-uniform)
        const int sub_keys = FULL ? 32 : st_keys - i * 32;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[ks], qf[ks], s, 0, 0, 0);       // S^T[key][query]
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv[ks], dof[ks], dp, 0, 0, 0);    // dP^T[key][query]
        }
        __builtin_amdgcn_sched_barrier(0);
        // K^T fragments of this sub-tile (for dQ^T += K^T dS^T), then the next sub-tile's row fragments: both land under the exp / dS work
        bf16x8 ft[2][DT];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) ft[s2][dt] = frag_tr<DH>(kst, 32 * i + 16 * s2, tro, dt);
        if (i + 1 < NSUB && (FULL || (i + 1) * 32 < st_keys)) {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) { fk[ks] = frag_rows<DH>(kst, 32 * (i + 1) + lq, ks, h); fv[ks] = frag_rows<DH>(vst, 32 * (i + 1) + lq, ks, h); }
        }
        __builtin_amdgcn_sched_barrier(0);
        if ((i & 1) == 0) {
          if constexpr (DROP && FULL) {   // ... staged in LDS with the operands (a global load here would wait for the next stage's DMA)
            kw = *reinterpret_cast<const uint32_t*>(smem_dyn + 2 * 2 * ST + (g & 1) * KWB + wave * KWW + (i >> 1) * 256 + lane * 4);
          } else if constexpr (DROP) {   // dP = dP_drop o keep / (1 - p): this lane's word of the 64-key tile, as the forward stored it
            const int nb32 = (L + 31) >> 5;
            kw = keep_bits[drop_word_index(bh, nb32, q0 >> 5, (st * SKB + i * 32) >> 6, lane)];
          }
        }
        const uint32_t km = kw >> (16 * (i & 1));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float p = __builtin_amdgcn_exp2f(s[r] * scale_log2e - lse_q);
          if (!FULL && sub_keys < 32 && (r & 3) + 8 * (r >> 2) + 4 * h >= sub_keys) p = 0.f;   // key past the sequence end
          float dpv = dp[r];
          if constexpr (DROP) dpv = and_bits(dpv, keep_mask(km, r));
          s[r] = p * (dpv - D_q);                                  // dS^T (the 1/sqrt(dh) factor is applied to dQ at the end)
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 dsf = cvt8(s, 8 * s2);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ft[s2][dt], dsf, dq[dt], 0, 0, 0);
        }
      }
    }
    if (st == nst - 1 && q0 + lq < L) {
      const int b = bh / nh, head = bh % nh;
      const int64_t tok = (int64_t)b * L + q0 + lq;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        bf16* dst = d_panel ? dQ + ((int64_t)(head * DT + dt) * ld_dq + tok) * 32 : dQ + tok * ld_dq + head * DH + dt * 32;
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
#ifdef MH_BWD_PROF
    {   // debug build (tools/debug/bwd_prof.py): per (block, wave) clocks spent waiting for the DMA, at the top barrier, in the stage, at the end barrier
      const unsigned long long tp3 = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();
      const unsigned long long tp4 = __builtin_amdgcn_s_memtime();
      prof_acc[0] += tp1 - tp0; prof_acc[1] += tp2 - tp1; prof_acc[2] += tp3 - tp2; prof_acc[3] += tp4 - tp3;
      if (g == total - 1 && lane == 0 && keep_bits) {
        unsigned long long* dstp = reinterpret_cast<unsigned long long*>(const_cast<uint32_t*>(keep_bits)) + ((size_t)blockIdx.x * NW + wave) * 4;
        dstp[0] = prof_acc[0]; dstp[1] = prof_acc[1]; dstp[2] = prof_acc[2]; dstp[3] = prof_acc[3];
      }
    }
#else
    __builtin_amdgcn_s_barrier();
#endif
  }
}

// ---------------------------------------------------------------------------------------------------------
template <int DH, bool DROP, bool FULL>
__global__ __launch_bounds__(512) void attn_bwd_dkv_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                           const bf16* __restrict__ V,
                                                           const bf16* __restrict__ dO,
                                                           const float* __restrict__ lse2, const float* __restrict__ Dv,
                                                           bf16* __restrict__ dK, bf16* __restrict__ dV, int64_t ld_d, int L,
                                                           int nh, int nbh, float scale, float scale_log2e, RowLayout lq_,
                                                           RowLayout lo_, const uint32_t* __restrict__ keep_bits, float rscale, int d_panel) {
  constexpr int NW = 8;
  constexpr int KS = DH / 16, DT = DH / 32;
  constexpr int ST = SKB * DH * 2;
  constexpr int PK = ST / 1024 / NW;
  constexpr int NSUB = SKB / 32;                   // 32-query sub-tiles per stage
  constexpr int LSTB = 2 * SKB * 4, KWQ = SKB / 32;                  // (lse2 | D) bytes per stage; 32-query blocks (1 KiB of keep words each) per stage
  constexpr int BUF = 2 * ST + LSTB + (DROP && FULL ? KWQ * 1024 : 0);   // Q rows, dO rows, (lse2 | D) of the stage's queries, keep words
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, lq = lane & 31;
  const int nkb = (L + 255) / 256, nst = (L + SKB - 1) / SKB;
  const int nitems = nbh * nkb;
  const int bx = mh_xcd_remap((int)blockIdx.x, (int)gridDim.x);     // (XCD-aware item order: see attn_bwd_dq_kernel)
  const int my_items = (nitems - bx + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_items * nst;
  // DROP: the S accumulators start at log2(1 / (1 - p_drop)) / (scale log2 e), so that exp2(s scale log2 e - lse2) is P / (1 - p_drop)
  const float s_init = DROP ? __builtin_amdgcn_logf(rscale) / scale_log2e : 0.f;
  const TrOff<DH> tro = tr_offsets<DH>(lane);

  auto issue = [&](int g) {
    const int item = bx + (g / nst) * gridDim.x, st = g % nst;
    const int bh = item / nkb;
    const int64_t r0 = (int64_t)bh * L + (int64_t)st * SKB;
    char* base = smem_dyn + (g & 1) * BUF;
    const int valid = L - st * SKB;
#pragma unroll
    for (int j = 0; j < PK; ++j) dma_rows<DH>(Q + lq_.at(bh, nh, (int64_t)st * SKB), lq_.ld, base, wave + NW * j, lane, valid);
#pragma unroll
    for (int j = 0; j < PK; ++j) dma_rows<DH>(dO + lo_.at(bh, nh, (int64_t)st * SKB), lo_.ld, base + ST, wave + NW * j, lane, valid);
    if constexpr (SKB == 128) {
      if (wave == 0) {   // lanes 0-31: lse2 of the 128 queries, lanes 32-63: D (groups of 4; past the end: the last valid group)
        int qo = 4 * (lane & 31);
        if (qo >= valid) qo = valid - 4;
        glds16((lane < 32 ? lse2 : Dv) + r0 + qo, base + 2 * ST);
      }
    } else {
      if (wave < 2) {    // wave 0: lse2 of the 256 queries, wave 1: D
        int qo = 4 * lane;
        if (qo >= valid) qo = valid - 4;
        glds16((wave == 0 ? lse2 : Dv) + r0 + qo, base + 2 * ST + wave * 1024);
      }
    }
    if constexpr (DROP && FULL) {   // keep words of (the stage's 32-query blocks) x (the block's 4 pairs of key blocks): 1 KiB per query block
      if (wave >= NW - KWQ)
        glds16(keep_bits + drop_word_index(bh, (L + 31) >> 5, st * KWQ + wave - (NW - KWQ), (item % nkb) * 4, 0) + 4 * lane,
               base + 2 * ST + LSTB + (wave - (NW - KWQ)) * 1024);
    }
  };

  bf16x8 kf[KS], vf[KS];
  f32x16 dk[DT], dv[DT];
  int k0 = 0;
  bool active = false;
  if (total > 0) issue(0);
  for (int g = 0; g < total; ++g) {
    const int item = bx + (g / nst) * gridDim.x, st = g % nst;
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
    const float* lst = reinterpret_cast<const float*>(qst + 2 * ST);   // [0..SKB) lse2, [SKB..2 SKB) D
    const int st_q = L - st * SKB;                                      // queries of this stage that exist
    if (active) {
      // (the bound-checking builds read a sub-tile's row fragments when they need them: the read-ahead does not fit their registers)
      bf16x8 fq[KS], fo[KS];
      if constexpr (FULL) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { fq[ks] = frag_rows<DH>(qst, lq, ks, h); fo[ks] = frag_rows<DH>(ost, lq, ks, h); }
      }
#pragma unroll
      for (int i = 0; i < NSUB; ++i) {
        if (!FULL && i * 32 >= st_q) break;                          // (wave-uniform)
        if constexpr (!FULL) {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) { fq[ks] = frag_rows<DH>(qst, 32 * i + lq, ks, h); fo[ks] = frag_rows<DH>(ost, 32 * i + lq, ks, h); }
        }
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = s_init; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[ks], kf[ks], s, 0, 0, 0);      // S[query][key]
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fo[ks], vf[ks], dp, 0, 0, 0);    // dP[query][key]
        }
        __builtin_amdgcn_sched_barrier(0);
        // dO^T and Q^T fragments of this sub-tile (transposing reads of the same stages): they land under the exp / dS work
        bf16x8 fot[2][DT], fqt[2][DT];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            fot[s2][dt] = frag_tr<DH>(ost, 32 * i + 16 * s2, tro, dt);
            fqt[s2][dt] = frag_tr<DH>(qst, 32 * i + 16 * s2, tro, dt);
          }
        // keep flags of this lane's key for its 16 queries of the tile: query 8 rg + 4 h + e lives in the word of forward lane
        // (query % 32) + 32 hk, at bit rk (+ 16 for an odd key block): four consecutive words per rg (fetched per rg: 16 live words
        // on top of the prefetched fragments do not fit 256 registers)
        const uint32_t* kwp = nullptr;
        int kshift = 0;
        if constexpr (DROP) {
          const int nb32 = (L + 31) >> 5;
          int qb = (st * SKB + 32 * i) >> 5; if (qb >= nb32) qb = nb32 - 1;
          const int kbk = k0 >> 5, hk = (lq >> 2) & 1;
          kshift = (lq & 3) + 4 * (lq >> 3) + 16 * (kbk & 1);
          if constexpr (FULL) kwp = reinterpret_cast<const uint32_t*>(qst + 2 * ST + LSTB) + i * 256 + ((wave >> 1) & 3) * 64 + 32 * hk + 4 * h;   // staged in LDS
          else kwp = keep_bits + drop_word_index(bh, nb32, qb, kbk >> 1, 32 * hk + 4 * h);
        }
        __builtin_amdgcn_sched_barrier(0);
        // register r <-> query (r & 3) + 8 (r >> 2) + 4 h of this 32-query tile: four runs of four consecutive queries
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int qi = 32 * i + 8 * rg + 4 * h;
          const f32x4 ls = *reinterpret_cast<const f32x4*>(lst + qi);
          const f32x4 dd = *reinterpret_cast<const f32x4*>(lst + SKB + qi);
          uint4 kwv = {0u, 0u, 0u, 0u};
          if constexpr (DROP) kwv = *reinterpret_cast<const uint4*>(kwp + 8 * rg);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float p = __builtin_amdgcn_exp2f(s[rg * 4 + e] * scale_log2e - ls[e]);
            if (!FULL && qi + e >= st_q) p = 0.f;                                  // query past the sequence end
            float dpv = dp[rg * 4 + e], pd = p;
            if constexpr (DROP) {                                         // P_drop feeds dV; dP = dP_drop o keep / (1 - p_drop)
              // p here is P / (1 - p_drop) (s_init) and D arrives multiplied by (1 - p_drop) (the dQ kernel): the factor costs nothing
              const uint32_t kwe = e == 0 ? kwv.x : e == 1 ? kwv.y : e == 2 ? kwv.z : kwv.w;
              const uint32_t km = keep_mask(kwe, kshift);
              pd = and_bits(p, km);
              dpv = and_bits(dpv, km);
            }
            s[rg * 4 + e] = pd;
            dp[rg * 4 + e] = p * (dpv - dd[e]);
          }
        }
        bf16x8 pf[2], dsf[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) { pf[s2] = cvt8(s, 8 * s2); dsf[s2] = cvt8(dp, 8 * s2); }
        __builtin_amdgcn_sched_barrier(0);
        // the next sub-tile's row fragments take the registers S and dP have just left; they land under the eight products below
        if constexpr (FULL) {
          if (i + 1 < NSUB) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { fq[ks] = frag_rows<DH>(qst, 32 * (i + 1) + lq, ks, h); fo[ks] = frag_rows<DH>(ost, 32 * (i + 1) + lq, ks, h); }
          }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fot[s2][dt], pf[s2], dv[dt], 0, 0, 0);
            dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fqt[s2][dt], dsf[s2], dk[dt], 0, 0, 0);
          }
      }
    }
    if (st == nst - 1 && k0 + lq < L) {
      const int b = bh / nh, head = bh % nh;
      const int64_t tok = (int64_t)b * L + k0 + lq;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int64_t doff = d_panel ? ((int64_t)(head * DT + dt) * ld_d + tok) * 32 : tok * ld_d + head * DH + dt * 32;
        bf16* dstk = dK + doff;
        bf16* dstv = dV + doff;
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
               RowLayout lqkv, RowLayout ldo, const uint32_t* keep_bits, float rscale, hipStream_t s, int o_panel = 0, int64_t o_ld = 0, int d_panel = 0) {
  constexpr int ST = SKB * DH * 2;
  constexpr int bytes_dq = 2 * 2 * ST + (DROP && FULL ? 2 * 8 * (SKB / 64) * 256 : 0), bytes_dkv = 2 * (2 * ST + 2 * SKB * 4 + (DROP && FULL ? (SKB / 32) * 1024 : 0));
  (void)qT; (void)kT; (void)dOT;   // (the transposed copies of the round-2 interface: no longer read)
  static bool attr_set[MH_MAX_DEVICES] = {};   // per device: hipFuncSetAttribute acts on the current device's copy of the kernel
  const int adev = mh_current_device();
  if (!attr_set[adev]) {
    MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<DH, DROP, FULL>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes_dq));
    MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<DH, DROP, FULL>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes_dkv));
    attr_set[adev] = true;
  }
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  const int nbh = B * nh, nitems = nbh * ((L + 255) / 256);
  const dim3 grid((unsigned)(nitems < cus ? nitems : cus)), block(512);
  const float sl2 = scale * 1.4426950408889634f;
  mh_prof_note("attn_bwd B*nh=%d L=%d dh=%d", nbh, L, DH);
  MH_LAUNCH((attn_bwd_dq_kernel<DH, DROP, FULL>), grid, block, bytes_dq, s, q, k, v, dO, o, lse2, Dv, dq, ld, L, nh, nbh, scale, sl2, lqkv, ldo,
            keep_bits, rscale, o_panel, o_ld, d_panel);
  MH_CHECK_LAUNCH();
  mh_prof_note("attn_bwd B*nh=%d L=%d dh=%d", nbh, L, DH);
  MH_LAUNCH((attn_bwd_dkv_kernel<DH, DROP, FULL>), grid, block, bytes_dkv, s, q, k, v, dO, lse2, Dv, dk, dv, ld, L, nh, nbh, scale, sl2, lqkv, ldo,
            keep_bits, rscale, d_panel);
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

extern "C" int mh_attention_stream_bwd_layout(const void* q, const void* k, const void* v, const void* dO, const void* o, int o_panel, int64_t o_ld,
                                              const float* lse2, float* D, void* dq, void* dk, void* dv, int64_t ld_d, int d_panel, int B, int L,
                                              int nh, int dh, float scale, int64_t qsB, int64_t qsH, int64_t qld, int64_t osB, int64_t osH,
                                              int64_t old_, const uint32_t* keep_bits, float drop_p, mh_stream_t stream);
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
  (void)qT_perm; (void)kT_perm; (void)dOT_perm;
  return mh_attention_stream_bwd_layout(q, k, v, dO, o, 0, 0, lse2, D, dq, dk, dv, ld_d, 0, B, L, nh, dh, scale, qsB, qsH, qld, osB, osH, old_, keep_bits,
                                        drop_p, stream);
}

// The same with the layouts of O and of the three outputs chosen by the caller: o_panel / d_panel != 0: K32 panels [H / 32][o_ld | ld_d rows][32]
// of the token-major tensors (dq / dk / dv then point at the first panel of their column block); dO (and, row-major, O) in the os* row layout.
extern "C" int mh_attention_stream_bwd_layout(const void* q, const void* k, const void* v, const void* dO, const void* o, int o_panel, int64_t o_ld,
                                              const float* lse2, float* D, void* dq, void* dk, void* dv, int64_t ld_d, int d_panel, int B, int L,
                                              int nh, int dh, float scale, int64_t qsB, int64_t qsH, int64_t qld, int64_t osB, int64_t osH,
                                              int64_t old_, const uint32_t* keep_bits, float drop_p, mh_stream_t stream) {
  const void* qT_perm = nullptr; const void* kT_perm = nullptr; const void* dOT_perm = nullptr;
  MH_CHECK_ARG(!o_panel || o_ld >= (int64_t)B * L, "attention_stream_bwd: o_ld must cover the B L token rows");
  MH_CHECK_ARG(!d_panel || ld_d >= (int64_t)B * L, "attention_stream_bwd: ld_d must cover the B L token rows");
  MH_CHECK_ARG(qsB % 8 == 0 && qsH % 8 == 0 && qld % 8 == 0 && osB % 8 == 0 && osH % 8 == 0 && old_ % 8 == 0 && qld >= dh && old_ >= dh,
               "attention_stream_bwd: row strides must be multiples of 8 elements");
  const RowLayout lqkv{qsB, qsH, qld}, ldo{osB, osH, old_};
  MH_CHECK_ARG(q && k && v && dO && o && lse2 && D && dq && dk && dv, "attention_stream_bwd: null pointer");   // (qT_perm, kT_perm, dOT_perm: unused, may be null)
  MH_CHECK_ARG(B > 0 && nh > 0 && mh_attention_stream_bwd_supported(L, dh),
               "attention_stream_bwd: needs seq_len %% 16 == 0, seq_len >= 512 and head dim 32 or 64 (got L=%d dh=%d)", L, dh);
  MH_CHECK_ARG(ld_d % 4 == 0, "attention_stream_bwd: ld_d must be a multiple of 4");
  MH_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || keep_bits), "attention_stream_bwd: dropout needs keep_bits and p in [0, 1)");   // (p == 0 with keep_bits: the MH_BWD_PROF debug build's output buffer)
  hipStream_t s = (hipStream_t)stream;
  const float rs = 1.0f / (1.0f - drop_p);
#define MH_BWD_ARGS (const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)qT_perm, (const bf16*)kT_perm, (const bf16*)dO, \
                    (const bf16*)dOT_perm, (const bf16*)o, lse2, D, (bf16*)dq, (bf16*)dk, (bf16*)dv, ld_d, B, L, nh, scale, lqkv, ldo, keep_bits, rs, s, \
                    o_panel, o_ld, d_panel
  if (L % 256 == 0 && mh_attention_stream_enabled() != 4) {   // (mode 4 = A/B: the bound-checking build on every length)
    if (drop_p > 0.f) return dh == 64 ? launch_bwd<64, true, true>(MH_BWD_ARGS) : launch_bwd<32, true, true>(MH_BWD_ARGS);
    return dh == 64 ? launch_bwd<64, false, true>(MH_BWD_ARGS) : launch_bwd<32, false, true>(MH_BWD_ARGS);
  }
  if (drop_p > 0.f) return dh == 64 ? launch_bwd<64, true, false>(MH_BWD_ARGS) : launch_bwd<32, true, false>(MH_BWD_ARGS);
  return dh == 64 ? launch_bwd<64, false, false>(MH_BWD_ARGS) : launch_bwd<32, false, false>(MH_BWD_ARGS);
#undef MH_BWD_ARGS
}
