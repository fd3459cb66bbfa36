// GEMM family: out = act(A W^T + bias) [+ residual]  and the fused QKV projection with head scatter.
//
// CDNA4 mapping (one design for both element types, 128x128 block tile, 4 waves as 2(M) x 2(N),
// each wave a 64x64 sub-tile = 4x4 MFMA tiles of 16x16):
//   bf16: v_mfma_f32_16x16x32_bf16, K-tile 64.  LDS rows are 128 B; the 16-B chunk c of row r lives
//         at chunk (c ^ (r & 7)) so that the ds_read_b128 fragment reads (16 rows x 2 chunks per
//         16-lane group) hit 16 distinct 16-B slots of the 256-B bank row: conflict-free.
//         Staging is register-staged (global_load_dwordx4 -> ds_write_b128).
//   f32:  v_mfma_f32_16x16x4_f32 (exact fp32 fma chain), K-tile 16.  The k index served by lane
//         group g in instruction s is 4g+s for BOTH operands, so one ds_read_b128 per tile row
//         feeds four MFMAs.  LDS rows are padded to 96 B (slot = 6*row + g: conflict-free).
//   Both: double-buffered LDS, one barrier per K-tile; the epilogue stages each wave's
//         accumulators through LDS so that bias / activation / residual / stores work on 8
//         contiguous columns per lane (16-B bf16 stores).
// nn.Linear convention: W is [N, K] row-major ("B^T"), which is exactly the k-contiguous layout
// the MFMA B operand wants, so no weight transposition happens anywhere.
#include <type_traits>

#include "common.h"

namespace {

// Deferred LayerNorm (bf16 throughput path): a dense + residual GEMM writes its RAW pre-LayerNorm rows plus per-row partial
// statistics (one (sum, sum of squares) pair per 128-column tile), and the consumers apply the normalisation themselves:
//   * as A operand:  LN(y) W^T + b = rstd_r ((y W'^T)_rc - mean_r c1_c) + c2_c  with W' = gamma o W (folded once, engine arena),
//     c1_c = sum_k W'_ck, c2_c = sum_k beta_k W_ck + b_c - the row scale / shift runs in the epilogue, on the accumulators;
//   * as residual:   (y - mean_r) rstd_r gamma_c + beta_c, element by element in the epilogue.
// No tile then needs to own complete rows: every GEMM of a layer runs on the 256x128 tile at two blocks per CU (d_model 768
// included, which has no full-row tile), and the LayerNorm kernels / epilogues disappear.
struct DeferArgs {
  const float* a_stats; int a_slots;   // A rows are raw: [M][a_slots][2] partial (sum, sumsq); g.bias then holds c2
  const float* c1;                     // [N] sum_k W'[c][k]
  const float* r_stats; int r_slots;   // residual rows are raw
  const float* r_gamma; const float* r_beta;
  float* o_stats; int o_slots;         // write the output rows' partial statistics, slot = column tile (n0 / BN)
  float inv_h, eps;                    // 1 / (normalised width), LayerNorm eps
};

struct GemmArgs {
  const void* A; int64_t lda;
  const void* W; int64_t ldw;
  const float* bias;
  const void* residual; int64_t ldr;
  void* out; int64_t ldo;
  int out_f32;
  int64_t M; int N; int K;
  int act;
  // QKV scatter
  void* q; void* k; void* vt;
  void* pre_out;   // EPI 0 with an activation (big tile): also store the pre-activation (bias added) here, same layout as out
  int64_t ldp; int p_panel;   // EPI 3 (training form): pre_out's own layout (row pitch / panel rows; 0 = as `out`: ldo, row-major)
  int act_grad;    // EPI 0 (big tile): `residual` holds a PRE-activation and the result is multiplied by act'(it) instead of added to
  int L, H, nh, dh;
  // EPI 2 (nearest-embedding scores): aux[col] = |W_col|^2, rown[row] = |x_row|^2, partial best per (row, slot)
  const float* aux; const float* rown; float* pbest; int32_t* pidx; int nslots;
  int a_panel, w_panel, o_panel, r_panel;  // operand stored as K32 panels: [cols/32][ld rows][32]
  int64_t sA, sW, sO, sR;  // batch strides in elements (grid.y = batch index)
  const float* ln_gamma; const float* ln_beta; float ln_eps;   // EPI 3
  int dbg;  // timing-only ablation bits (mh_gemm_set_debug): 1 no DMA, 2 no MFMA, 4 no stores
  int ntiles;    // persistent big-tile launch: ntiles output tiles walked by gridDim.x blocks
  int vt_perm;   // QKV scatter: V^T keys in the P-operand order of mh_attention_stream_fwd (middle groups of 4 swapped per 16)
  int pre_kind;  // what pre_out receives: 0 the pre-activation, 1 act'(pre) (GELU: gelu_erf_fast8_dgelu)
  float q_scale; // QKV scatter (big tile): the query columns are stored multiplied by this (0 = unscaled): softmax scale x log2(e) for mh_attention_stream_fwd_prescaled
  DeferArgs d;   // DBG bit 128 kernels only
  DropArgs drop; // EPI 0: train-mode dropout of (A W^T + bias) before the residual is added (thr == 0: off)
};

template <typename T> struct Tile;
template <> struct Tile<bf16> { static constexpr int BK = 64, ROWB = 128, CHUNKS = 8; };
template <> struct Tile<float> { static constexpr int BK = 16, ROWB = 80, CHUNKS = 4; };

constexpr int BM = 128, BN = 128, CS_LD = 68;

template <typename T>
__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case MH_ACT_TANH: return sizeof(T) == 2 ? tanh_fast(v) : tanhf(v);
    case MH_ACT_GELU_ERF: return sizeof(T) == 2 ? gelu_erf_fast(v) : gelu_erf(v);
    case MH_ACT_SILU: return silu(v);
    default: return v;
  }
}

// XCD-aware block remap (8 XCDs, blocks dealt round-robin): give each XCD a contiguous run of
// tiles so the A row-panel a run shares stays in that XCD's L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

template <typename T, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs g) {
  using TT = Tile<T>;
  constexpr int BK = TT::BK, ROWB = TT::ROWB, CHUNKS = TT::CHUNKS;
  constexpr int TILE_BYTES = 128 * ROWB;
  constexpr int NCH = (128 * CHUNKS) / 256;  // 16-B chunks per thread per operand tile
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (g.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t m0 = (int64_t)(bid / tiles_n) * BM;
  const int n0 = (bid % tiles_n) * BN;
  const T* __restrict__ A = reinterpret_cast<const T*>(g.A) + (int64_t)blockIdx.y * g.sA;
  const T* __restrict__ W = reinterpret_cast<const T*>(g.W) + (int64_t)blockIdx.y * g.sW;
  const int nk = g.K / BK;
  constexpr int EPC = 16 / sizeof(T);  // elements per 16-B chunk

  // ---- per-thread staging coordinates (fixed over the K loop)
  const T* srcA[NCH];
  const T* srcW[NCH];
  int ldsoff[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    int row, c, off;
    const int qd = tid + 256 * j;
    row = qd / CHUNKS;
    c = qd % CHUNKS;
    if constexpr (sizeof(T) == 2) off = row * ROWB + ((c ^ (row & 7)) << 4);
    else off = row * ROWB + (c << 4);
    int64_t ra = m0 + row; if (ra >= g.M) ra = g.M - 1;
    int rw = n0 + row; if (rw >= g.N) rw = g.N - 1;
    srcA[j] = A + ra * g.lda + c * EPC;
    srcW[j] = W + (int64_t)rw * g.ldw + c * EPC;
    ldsoff[j] = off;
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 stA[NCH], stW[NCH];  // staging registers

  auto issue = [&](int kt, int) {
    const int koff = kt * BK;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      stA[j] = *reinterpret_cast<const f32x4*>(srcA[j] + koff);
      stW[j] = *reinterpret_cast<const f32x4*>(srcW[j] + koff);
    }
  };
  auto commit = [&](int buf) {
    char* base = smem + buf * 2 * TILE_BYTES;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      *reinterpret_cast<f32x4*>(base + ldsoff[j]) = stA[j];
      *reinterpret_cast<f32x4*>(base + TILE_BYTES + ldsoff[j]) = stW[j];
    }
  };

  issue(0, 0);
  commit(0);
  __syncthreads();

  const int fr = lane & 15, fg = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
    const char* As = smem + cur * 2 * TILE_BYTES;
    const char* Ws = As + TILE_BYTES;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 a[4], b[4];
        const int chunk = kk * 4 + fg;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ra = wm * 64 + i * 16 + fr;
          a[i] = *reinterpret_cast<const bf16x8*>(As + ra * ROWB + ((chunk ^ (ra & 7)) << 4));
          const int rb = wn * 64 + i * 16 + fr;
          b[i] = *reinterpret_cast<const bf16x8*>(Ws + rb * ROWB + ((chunk ^ (rb & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    } else {
      f32x4 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const f32x4*>(As + (wm * 64 + i * 16 + fr) * ROWB + (fg << 4));
        b[i] = *reinterpret_cast<const f32x4*>(Ws + (wn * 64 + i * 16 + fr) * ROWB + (fg << 4));
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) commit(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: each wave stages 32 rows x 64 cols of fp32 at a time in its own LDS region
  float* Cs = reinterpret_cast<float*>(smem) + wave * (32 * CS_LD);
  const bool vec_ok = (g.ldo % 8 == 0) && (g.ldr % 8 == 0);
  T* outT = reinterpret_cast<T*>(g.out) + (int64_t)blockIdx.y * g.sO;
  float* outF = reinterpret_cast<float*>(g.out) + (int64_t)blockIdx.y * g.sO;
  const T* res = g.residual ? reinterpret_cast<const T*>(g.residual) + (int64_t)blockIdx.y * g.sR : nullptr;
  const int wcol0 = n0 + wn * 64;                      // first column of this wave's region
#pragma unroll
  for (int p = 0; p < 2; ++p) {
#pragma unroll
    for (int il = 0; il < 2; ++il)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Cs[(il * 16 + fg * 4 + r) * CS_LD + j * 16 + fr] = acc[2 * p + il][j][r];
    __syncthreads();
    const int64_t wrow0 = m0 + wm * 64 + p * 32;
    if constexpr (EPI == 1) {
      // QKV scatter.  A wave's 64 columns lie inside one of Q | K | V and inside one head block
      // boundary multiple (H % 64 == 0), so `which` is wave-uniform.
      const int which = wcol0 / g.H;
      if (which < 2) {
        T* dst = reinterpret_cast<T*>(which == 0 ? g.q : g.k);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int idx = it * 64 + lane, rl = idx >> 3, c8 = (idx & 7) * 8;
          const int64_t row = wrow0 + rl;
          const int col = wcol0 + c8;
          if (row < g.M && col < g.N) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = Cs[rl * CS_LD + c8 + e] + g.bias[col + e];
            const int c = col - which * g.H, head = c / g.dh, d = c % g.dh;
            const int64_t b = row / g.L, l = row % g.L;
            store8(dst + ((b * g.nh + head) * g.L + l) * g.dh + d, v);
          }
        }
      } else {
        T* dst = reinterpret_cast<T*>(g.vt);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int idx = it * 64 + lane, cl = idx & 63, rg = idx >> 6;
          const int64_t row = wrow0 + rg * 8;
          const int col = wcol0 + cl;
          if (row < g.M && col < g.N) {
            float v[8];
            const float bv = g.bias[col];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = Cs[(rg * 8 + e) * CS_LD + cl] + bv;
            const int c = col - 2 * g.H, head = c / g.dh, d = c % g.dh;
            const int64_t b = row / g.L, l = row % g.L;
            store8(dst + ((b * g.nh + head) * g.dh + d) * g.L + l, v);
          }
        }
      }
    } else if constexpr (EPI == 2) {
      // rounding scores (models/rounding.py:21-28): -(clamp((|W_v|^2 + |x_n|^2) - 2 x.W_v, 0)); every row keeps
      // the best (score, first index) of this wave's 64 columns -> one partial per (row, column-slot)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = it * 64 + lane, rl = idx >> 3, c8 = (idx & 7) * 8;
        const int64_t row = wrow0 + rl;
        const int col = wcol0 + c8;
        float best = -INFINITY;
        int bi = 0x7fffffff;
        if (row < g.M) {
          const float xn = g.rown[row];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (col + e < g.N) {
              float dist = (g.aux[col + e] + xn) - 2.0f * Cs[rl * CS_LD + c8 + e];
              dist = fmaxf(dist, 0.0f);
              const float sc = -dist;
              if (sc > best) { best = sc; bi = col + e; }
            }
          }
        }
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
          const float so = __shfl_xor(best, off, 64);
          const int io = __shfl_xor(bi, off, 64);
          if (so > best || (so == best && io < bi)) { best = so; bi = io; }
        }
        if ((lane & 7) == 0 && row < g.M) {
          const int slot = (n0 / BN) * 2 + wn;
          g.pbest[row * g.nslots + slot] = best;
          g.pidx[row * g.nslots + slot] = bi;
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = it * 64 + lane, rl = idx >> 3, c8 = (idx & 7) * 8;
        const int64_t row = wrow0 + rl;
        const int col = wcol0 + c8;
        if (row < g.M && col < g.N) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = Cs[rl * CS_LD + c8 + e];
          const int nv = (g.N - col) < 8 ? (g.N - col) : 8;
          if (g.bias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e < nv) v[e] += g.bias[col + e];
          }
          if (g.act != MH_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = apply_act<T>(v[e], g.act);
          }
          if (g.drop.thr) {   // HF BertSelfOutput / BertOutput: dense -> dropout -> (+ input) (N % 8 == 0: checked by the entry point)
            const uint32_t km = drop_keep8_at(g.drop, (uint64_t)row * g.N + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (km >> e) & 1u ? v[e] * g.drop.rscale : 0.f;
          }
          if (nv == 8 && vec_ok) {
            if (res) {
              float rv[8];
              load8(res + row * g.ldr + col, rv);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += rv[e];
            }
            if (g.out_f32) store8(outF + row * g.ldo + col, v);
            else store8(outT + row * g.ldo + col, v);
          } else {
            for (int e = 0; e < nv; ++e) {
              float x = v[e];
              if (res) x += to_f32(res[row * g.ldr + col + e]);
              if (g.out_f32) outF[row * g.ldo + col + e] = x;
              else outT[row * g.ldo + col + e] = from_f32<T>(x);
            }
          }
        }
      }
    }
    __syncthreads();
  }
}


// =====================================================================================================
// bf16 "big tile" kernels (the throughput path).  The kernel above is latency-bound at the denoiser's
// shapes (K = 512: eight K-steps, one tile of prefetch): rocprof showed ~14 us per 128x128 tile against
// 1.7 us of MFMA time.  These are built around keeping loads in flight:
//   * block tile BM x BN, WM x WN waves, each wave (BM/WM) x (BN/WN) = TI x TJ MFMA tiles of 16x16x32;
//     K-step 32; NST-stage LDS ring filled by global_load_lds_dwordx4 (one 1-KiB DMA piece = 16 rows x 64 B),
//     counted `s_waitcnt vmcnt(pieces x stages-in-flight)` + raw s_barrier so that younger stages stay in
//     flight ACROSS the barrier (a __syncthreads() would drain them).
//   * 64-B LDS rows, chunk c of row r stored at c ^ G[(r>>2)&3], G = {0,2,3,1}: every 16-lane group of a
//     ds_read_b128 fragment read hits 16 distinct 16-B slots (SQ_LDS_BANK_CONFLICT = 0); the DMA writes LDS
//     linearly, so the swizzle is applied to the per-lane global SOURCE address.
//   * the MFMA is issued with the operands SWAPPED (D = W_tile . A_tile^T) and the W rows of each 64-column
//     group are dealt to the MFMA input rows as 32(jj>>1) + 8(p>>2) + 4(jj&1) + (p&3): a lane then owns 8
//     CONSECUTIVE output columns of one row per (group, half), so bias / activation / residual / LayerNorm /
//     stores work straight from the accumulators with 16-byte accesses - no LDS round trip.
//     (V^T of the QKV projection wants consecutive TOKENS per lane instead: those waves issue the MFMA
//     un-swapped with the plain row order.)
//   * operands row-major or K32-panel (runtime strides only).
// Configurations:  Std 256x128 / 4 waves / 3 stages (72 KiB: two blocks per CU);  Wide 256x256 / 8 waves /
// 4 stages (128 KiB, 1.5x fewer DMA bytes per flop);  Row 128x512 / 8 waves / 3 stages (120 KiB): one block
// owns complete rows of an N = 512 output, which lets bias + residual + LayerNorm run in the epilogue.
template <int BM_, int BN_, int WM_, int WN_, int NST_, bool PP_ = false>
struct BigCfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, NST = NST_;
  static constexpr bool PP = PP_;   // ping-pong main loop (two wave groups half a K-step apart)
  static constexpr int PRO = PP_ ? NST_ - 1 : NST_;   // stages a main loop has in flight before its first K-step
  static constexpr int NW = WM * WN, THREADS = NW * 64;
  static constexpr int TI = BM / WM / 16, TJ = BN / WN / 16;
  static constexpr int STAGE = (BM + BN) * 64;
  // DMA pieces (16 rows x 64 B) per wave and stage.  A tile with fewer A pieces than waves (BM 64 on 8 waves) still gives every
  // wave one: the upper waves re-load the lower waves' pieces (identical bytes to the same LDS address), so that every wave's
  // vmcnt arithmetic stays the same
  static constexpr int APIECES = BM / 16;
  static constexpr int PA = (APIECES + NW - 1) / NW, PW = BN / 16 / NW, PIECES = PA + PW;
  static_assert(PA >= 1 && PW >= 1 && TJ % 4 == 0 && NST >= 3 && (APIECES % NW == 0 || NW % APIECES == 0), "unsupported big-tile configuration");
};
using CfgStd = BigCfg<256, 128, 2, 2, 3>;
using CfgWide = BigCfg<256, 256, 2, 4, 4>;
using CfgRow = BigCfg<128, 512, 2, 4, 3>;
using CfgWidePP = BigCfg<256, 256, 2, 4, 4, true>;
using CfgRowPP = BigCfg<128, 512, 2, 4, 3, true>;
using CfgRow64 = BigCfg<64, 512, 1, 8, 3>;   // full-row tile over 64 rows: twice the blocks of CfgRow (short K: the epilogue dominates)
constexpr int B2K = 32;

// one DMA stage (K-step kt) of a tile into ring slot kt % NST: PA + PW 1-KiB pieces per wave
// the LayerNorm epilogue's residual rows are read once, by this block: non-temporal loads (same-box A/B of two builds: -0.6 % step time)
#ifndef MH_EPI3_RES_NT
#define MH_EPI3_RES_NT 1
#endif
// cache policy of the full-row (ping-pong) tile's A-operand DMA: nt (aux 2) - its rows are read by exactly one block, once, and
// should not displace the weight matrix every block re-reads from L2 (same-box A/B of two builds: -1.6 % step time; 0 = default policy)
#ifndef MH_PP_A_AUX
#define MH_PP_A_AUX 2
#endif
// DBG bit 16384 kernels (K32-panel operands only): the stage DMA as `buffer_load_dwordx4 ... lds` - ONE per-lane byte offset per operand for the
// whole kernel (a lane's chunk of its piece), the tile's base in a buffer descriptor (scalar registers), the K step as the instruction's
// scalar offset and a wave's consecutive pieces (16 rows x 64 B apart) as its immediate offset: no vector instruction per piece, where the
// `global_load_lds` form spends two 64-bit vector adds on every piece of every K step.  Rows beyond M / N are not clamped: inside the buffer they
// read other rows (their outputs are never stored), beyond it the descriptor's bound makes them zeros.
struct BufDma {
  __amdgpu_buffer_rsrc_t ra, rw;   // tile bases: A rows tm0.., W rows tn0.. of panel 0
  int va, vw;                      // this lane's byte offset inside its first piece's rows
  int ka, kw;                      // bytes per K32 panel
};
template <int J, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (J < N) { f(std::integral_constant<int, J>{}); static_for<J + 1, N>(f); }
}
template <class C>
__device__ __forceinline__ void issue_stage_buf(const char* smem, const BufDma& b, const int (&ldsA)[C::PA], const int (&ldsW)[C::PW], int kt) {
  char* base = const_cast<char*>(smem) + (kt % C::NST) * C::STAGE;
  constexpr int AUX_A = C::PP ? MH_PP_A_AUX : 0;     // (the full-row tile's A rows: nt, as in issue_stage)
  static_for<0, C::PA>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    // (the instruction's immediate offset advances the LDS address as well as the buffer address: every piece names the FIRST piece's slot)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(b.ra, (__attribute__((address_space(3))) void*)(base + ldsA[0]), 16, b.va, kt * b.ka, j * 1024, AUX_A);
  });
  static_for<0, C::PW>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(b.rw, (__attribute__((address_space(3))) void*)(base + C::BM * 64 + ldsW[0]), 16, b.vw, kt * b.kw, j * 1024, 0);
  });
}

template <class C, int DBG>
__device__ __forceinline__ void issue_stage(const char* smem, const char* const (&srcA)[C::PA], const char* const (&srcW)[C::PW],
                                            const int (&ldsA)[C::PA], const int (&ldsW)[C::PW], int kt, int64_t kstepA,
                                            int64_t kstepW) {
  if constexpr ((DBG & 1) != 0) return;
  char* base = const_cast<char*>(smem) + (kt % C::NST) * C::STAGE;
#pragma unroll
  for (int j = 0; j < C::PA; ++j) {
    if constexpr (C::PP)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[j] + kt * kstepA),
                                       (__attribute__((address_space(3))) void*)(base + ldsA[j]), 16, 0, MH_PP_A_AUX);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[j] + kt * kstepA),
                                       (__attribute__((address_space(3))) void*)(base + ldsA[j]), 16, 0, 0);
  }
  if constexpr ((DBG & 4096) != 0) return;   // timing-only ablation: the W pieces are not issued (what their issue costs the loop)
#pragma unroll
  for (int j = 0; j < C::PW; ++j)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW[j] + kt * kstepW),
                                     (__attribute__((address_space(3))) void*)(base + C::BM * 64 + ldsW[j]), 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until all but the youngest `stages` DMA stages (PIECES loads each) of this wave have landed; EXTRA = vector-memory
// operations of another kind (the previous tile's epilogue stores) issued after the awaited stage (vmcnt counts in issue order)
template <int PIECES, int EXTRA = 0> __device__ __forceinline__ void wait_stages(int stages) {
  static_assert(3 * PIECES + EXTRA <= 63, "vmcnt is a 6-bit counter");
  if (stages >= 3) wait_vmcnt<3 * PIECES + EXTRA>();
  else if (stages == 2) wait_vmcnt<2 * PIECES + EXTRA>();
  else if (stages == 1) wait_vmcnt<PIECES + EXTRA>();
  else wait_vmcnt<EXTRA>();
}

// Main loop.  Fragments of K-step kt live in registers while its MFMAs run; the fragments of kt+1 are read
// from LDS underneath them (W into a second register set up front, the A row-fragment i into its own
// registers right after the last MFMA that uses it), so neither the LDS latency nor its bandwidth
// (12 KiB per wave per K-step) sits between two MFMA phases.  All NST ring slots hold DMA stages: slot
// kt % NST is refilled with stage kt + NST as soon as the barrier says every wave has read stage kt out of it.
// `pre`: 0 = issue the first stages here; 1 = they were issued before the previous tile's epilogue: drain everything (stores
// included); 2 = the same, and that epilogue issued exactly NSTORE stores per wave (a full tile): the waits for the prefetched
// stages count the stores as younger operations instead of waiting for them, so the stores drain under this tile's first K-steps.
template <class C, bool SWAP, int DBG, int NSTORE = 0>
__device__ __forceinline__ void big_mainloop(f32x4 (&acc)[C::TI][C::TJ], const char* smem, const char* const (&srcA)[C::PA],
                                             const char* const (&srcW)[C::PW], const int (&ldsA)[C::PA], const int (&ldsW)[C::PW],
                                             int nk, int a_off, const int (&b_offs)[C::TJ], int64_t kstepA, int64_t kstepW,
                                             int pre, unsigned* prof = nullptr, const BufDma* bd = nullptr) {
  constexpr int TI = C::TI, TJ = C::TJ;
  constexpr int NPIECES = (DBG & 4096) != 0 ? C::PA : C::PIECES;   // (ablation 4096: only the A pieces are issued)
  // DBG bit 4 (tools/gemm_bench.py --dbg 28): per-wave shader-clock totals of the three phases of a K-step
  unsigned long long pt_wait = 0, pt_bar = 0, pt_work = 0, pt0 = 0, pt1 = 0;
  auto tick = [&]() -> unsigned long long { if constexpr ((DBG & 16) != 0) return __builtin_amdgcn_s_memtime(); else return 0ull; };
  auto issue = [&](int kt) {
    if constexpr ((DBG & 16384) != 0) issue_stage_buf<C>(smem, *bd, ldsA, ldsW, kt);
    else issue_stage<C, DBG>(smem, srcA, srcW, ldsA, ldsW, kt, kstepA, kstepW);
  };
  auto read_frag = [&](const char* p) -> bf16x8 {
    if constexpr ((DBG & 8) != 0) { bf16x8 v; asm volatile("" : "=v"(v)); return v; }   // ablation: no LDS reads
    else return *reinterpret_cast<const bf16x8*>(p);
  };
  unsigned long long pc0 = 0, pr0 = 0;
  if constexpr ((DBG & 16) != 0) { pc0 = __builtin_amdgcn_s_memtime(); pr0 = __builtin_amdgcn_s_memrealtime(); }
  const int npro = nk < C::NST ? nk : C::NST;
  constexpr bool COUNTED = NSTORE > 0 && (C::NST - 1) * NPIECES + NSTORE <= 63;
  if (pre == 2 && COUNTED) {
    wait_stages<NPIECES, COUNTED ? NSTORE : 0>(npro - 1);   // stage 0 landed; stages 1.. and the stores stay in flight
  } else if (pre) {   // the stages were issued before the previous tile's epilogue, whose stores share the counter: drain all
    wait_vmcnt<0>();
  } else {
    for (int st = 0; st < npro; ++st) issue(st);
    wait_stages<NPIECES>(npro - 1);
  }
  __builtin_amdgcn_s_barrier();
  bf16x8 a[TI], b[TJ];
#pragma unroll
  for (int j = 0; j < TJ; ++j) b[j] = read_frag(smem + C::BM * 64 + b_offs[j]);
#pragma unroll
  for (int i = 0; i < TI; ++i) a[i] = read_frag(smem + a_off + i * (16 * 64));
  auto mfma_row = [&](int i) {
    if constexpr ((DBG & 2) != 0) {
      asm volatile("" ::"v"(a[i]));
#pragma unroll
      for (int j = 0; j < TJ; ++j) asm volatile("" ::"v"(b[j]));
    } else {
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        if constexpr (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
  };
  // One K step.  (Two forms of it were built and measured out in round 5 - profiles/r05_ab_nulls.txt: the loop unrolled by the ring depth so
  // that every fragment read is `base + immediate`: +20 % per step, four copies of a K step do not fit the instruction cache; the next step's
  // W fragments into a second register set at the head of the step: +0.9 %, 16 registers and 8 moves per step.)
  auto kstep = [&](int kt) {
    const char* As = smem + ((kt + 1) % C::NST) * C::STAGE;
    const char* Ws = As + C::BM * 64;
    // stage kt+1 must have landed (stages kt+2 .. kt+NST-1 stay in flight across the barrier); this wave's
    // reads of stage kt were issued a whole MFMA phase ago, so the lgkmcnt wait is free
    const int younger = nk - 2 - kt < C::NST - 2 ? nk - 2 - kt : C::NST - 2;
    pt0 = tick();
    if (kt > 0) pt_work += pt0 - pt1;
    // stage kt+1 was prefetched before the stores for kt + 1 < NST: the stores are younger than it
    if (pre == 2 && COUNTED && kt + 1 < C::NST) wait_stages<NPIECES, COUNTED ? NSTORE : 0>(younger);
    else wait_stages<NPIECES>(younger);
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
    pt1 = tick();
    pt_wait += pt1 - pt0;
    __builtin_amdgcn_s_barrier();
    pt0 = tick();
    pt_bar += pt0 - pt1;
    pt1 = pt0;
    if (kt + C::NST < nk) issue(kt + C::NST);   // slot kt % NST: every wave has read stage kt out of it
    // the W fragments of stage kt + 1 are read straight into b[] behind the LAST MFMA row of this step (their latency falls under the next
    // step's wait + barrier): no second register set, no copy
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      mfma_row(i);
      a[i] = read_frag(As + a_off + i * (16 * 64));
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j) b[j] = read_frag(Ws + b_offs[j]);
  };
  for (int kt = 0; kt + 1 < nk; ++kt) kstep(kt);
#pragma unroll
  for (int i = 0; i < TI; ++i) mfma_row(i);
  if constexpr ((DBG & 16) != 0) {
    const unsigned long long pc1 = __builtin_amdgcn_s_memtime(), pr1 = __builtin_amdgcn_s_memrealtime();
    if (prof && (threadIdx.x & 63) == 0) {
      prof[0] = (unsigned)pt_wait; prof[1] = (unsigned)pt_bar; prof[2] = (unsigned)pt_work; prof[3] = (unsigned)nk;
      prof[4] = (unsigned)(pc1 - pc0); prof[5] = (unsigned)(pr1 - pr0);
    }
  }
}

// Ping-pong main loop (one block of 8 waves per CU).  The two wave rows (group = wm) run half a K-step
// apart: while one group issues its 32 MFMAs at raised priority, the other - its partner wave on every SIMD -
// reads its 12 fragments of the next K-step from LDS and issues its share of the DMA (an LDS-DMA piece
// costs the ISSUING wave ~100 cycles; put behind the partner's MFMAs it costs the matrix pipe nothing).
// Slots are separated by block-wide raw barriers; group 1 enters one barrier late and group 0 leaves one late.
//   group 0: slot 2kt = LOAD(kt), slot 2kt+1 = MFMA(kt);   group 1: slot 2kt+1 = LOAD(kt), slot 2kt+2 = MFMA(kt)
//   LOAD(kt) reads stage kt and issues stage kt+NST-1 into the ring slot of stage kt-1, whose last readers
//   (group 1, slot 2kt-1) drained lgkmcnt before the barrier that opens slot 2kt;
//   every wave retires its pieces of stage kt+1 at the end of slot 2kt+1, before the barrier that opens the
//   slot in which group 0 reads it - two younger stages stay in flight across that barrier.
template <class C, bool SWAP, int DBG>
__device__ __forceinline__ void pp_mainloop(f32x4 (&acc)[C::TI][C::TJ], const char* smem, const char* const (&srcA)[C::PA],
                                            const char* const (&srcW)[C::PW], const int (&ldsA)[C::PA], const int (&ldsW)[C::PW],
                                            int nk, int a_off, const int (&b_offs)[C::TJ], int64_t kstepA, int64_t kstepW,
                                            int group, bool pre, const BufDma* bd = nullptr) {
  constexpr int TI = C::TI, TJ = C::TJ, D = C::NST - 1;
  static_assert(C::WM == 2, "ping-pong needs exactly two wave rows");
  auto issue = [&](int kt) {
    if constexpr ((DBG & 16384) != 0) issue_stage_buf<C>(smem, *bd, ldsA, ldsW, kt);
    else issue_stage<C, DBG>(smem, srcA, srcW, ldsA, ldsW, kt, kstepA, kstepW);
  };
  auto read_frag = [&](const char* p) -> bf16x8 {
    if constexpr ((DBG & 8) != 0) { bf16x8 v; asm volatile("" : "=v"(v)); return v; }
    else return *reinterpret_cast<const bf16x8*>(p);
  };
  const int npro = nk < D ? nk : D;
  if (pre) {
    wait_vmcnt<0>();
  } else {
    for (int st = 0; st < npro; ++st) issue(st);
    wait_stages<C::PIECES>(npro - 1);
  }
  __builtin_amdgcn_s_barrier();
  if (group == 1) __builtin_amdgcn_s_barrier();
  bf16x8 a[TI], b[TJ];
  for (int kt = 0; kt < nk; ++kt) {
    const char* As = smem + (kt % C::NST) * C::STAGE;
    const char* Ws = As + C::BM * 64;
    const int younger = nk - 2 - kt < D - 1 ? nk - 2 - kt : D - 1;   // stages issued after kt+1 by the end of slot 2kt+1
    // ---- LOAD(kt)
#pragma unroll
    for (int j = 0; j < TJ; ++j) b[j] = read_frag(Ws + b_offs[j]);
#pragma unroll
    for (int i = 0; i < TI; ++i) a[i] = read_frag(As + a_off + i * (16 * 64));
    if (kt + D < nk) issue(kt + D);
    if (group == 1 && kt + 1 < nk) wait_stages<C::PIECES>(younger);
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): fragments in registers, ring slot released
    __builtin_amdgcn_s_barrier();
    // ---- MFMA(kt)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      if constexpr ((DBG & 2) != 0) {
        asm volatile("" ::"v"(a[i]));
#pragma unroll
        for (int j = 0; j < TJ; ++j) asm volatile("" ::"v"(b[j]));
      } else {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          if constexpr (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    if (group == 0 && kt + 1 < nk) wait_stages<C::PIECES>(younger);
    __builtin_amdgcn_s_barrier();
  }
  if (group == 0) __builtin_amdgcn_s_barrier();
}

template <class C, bool SWAP, int DBG, int NSTORE = 0>
__device__ __forceinline__ void run_mainloop(f32x4 (&acc)[C::TI][C::TJ], const char* smem, const char* const (&srcA)[C::PA],
                                             const char* const (&srcW)[C::PW], const int (&ldsA)[C::PA], const int (&ldsW)[C::PW],
                                             int nk, int a_off, const int (&b_offs)[C::TJ], int64_t kstepA, int64_t kstepW,
                                             int group, int pre, unsigned* prof = nullptr, const BufDma* bd = nullptr) {
  if constexpr (C::PP) pp_mainloop<C, SWAP, DBG>(acc, smem, srcA, srcW, ldsA, ldsW, nk, a_off, b_offs, kstepA, kstepW, group, pre != 0, bd);
  else big_mainloop<C, SWAP, DBG, NSTORE>(acc, smem, srcA, srcW, ldsA, ldsW, nk, a_off, b_offs, kstepA, kstepW, pre, prof, bd);
}

// EPI: 0 generic (bias / act / residual), 1 QKV head scatter, 3 bias + residual + LayerNorm over complete rows
template <class C, int EPI, int ACT, int DBG = 0>
__global__ __launch_bounds__(C::THREADS, C::STAGE * C::NST <= 80 * 1024 ? 2 : 1) void gemm_big_kernel(const GemmArgs g) {
  // LDS: the DMA ring, then (EPI 3) the row-statistics exchange - kept apart so that the next tile's first
  // stages can already be landing in the ring while this tile's epilogue runs
  // (EPI 3 also keeps bias / LayerNorm gain / shift of the block's BN = N columns in LDS: read back with ds_read in the epilogue,
  // they cost neither vector registers across the main loop nor vmcnt waits between the stores)
  // deferred LayerNorm (DeferArgs), one compiled variant per operand combination so that unused vectors cost no registers:
  // DA = A rows raw, DR = residual rows raw, DO = write the output rows' partial statistics
  constexpr bool DA = (DBG & 128) != 0, DR = (DBG & 256) != 0, DO = (DBG & 512) != 0, DEFER = DA || DR || DO;
  constexpr int TI_ = C::TI, TJ_ = C::TJ;
  // (EPI 3 stages the tile's residual rows through LDS after the main loop: each wave's TI x TJ/2 KiB go where the ring was;
  // a last wave that does not fit - the 128x512 tile: 8 x 16 KiB against a 120 KiB ring - gets its own area at the end)
  constexpr int RES_W = TI_ * (TJ_ / 2) * 1024, RING = C::NST * C::STAGE;
  constexpr bool RES_EXTRA = EPI == 3 && C::NW * RES_W > RING;
  static_assert(EPI != 3 || (C::NW - 1) * RES_W <= RING, "residual staging: at most the last wave may overflow the ring");
  // EPI 0, DBG bits 16..19 (round 6): the FORM of the epilogue fixed at compile time - 1 plain (no residual), 2 residual add, 4 the residual
  // tensor holds act'(pre) and multiplies (the backward's act-grad GEMM), 8 ACT = GELU with gelu'(pre) as a second output (the training
  // forward's FFN1); 0 = generic: `residual`, `act_grad`, `out_f32`, `pre_out` are looked at per 8-value group.  They are wave-uniform and
  // loop-invariant, but hipcc unswitches them into 20 - 27 thousand instructions per kernel (the straight-line forms have 4 - 5 thousand: the
  // difference is the instruction cache); launch_big picks the form
  constexpr int FORM = (DBG >> 16) & 15;
  constexpr bool GEN = FORM == 0;
  static_assert(GEN || (EPI == 0 && !DEFER && (DBG & 64) == 0 && (FORM == 1 || FORM == 2 || FORM == 4 || (FORM == 8 && ACT == MH_ACT_GELU_ERF))), "fixed epilogue forms: EPI 0, no deferred LayerNorm, no dropout");
  __shared__ __attribute__((aligned(16))) char smem[C::NST * C::STAGE + (EPI == 3 ? C::BM * C::WN * 4 + 3 * C::BN * 4 : 0) +
                                                    (DEFER ? C::BM * 8 * (2 + C::WN) : 0) + (RES_EXTRA ? RES_W : 0)];
  constexpr int TI = C::TI, TJ = C::TJ;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WN, wn = wave % C::WN;
  const int tiles_n = (g.N + C::BN - 1) / C::BN;
  const int nk = g.K / B2K;
  const int fr = lane & 15, fg = lane >> 4;
  constexpr int GSW[4] = {0, 2, 3, 1};
  // DMA coordinates: one piece covers 16 rows x 64 B; lane i lands at row i/4, physical chunk i%4.
  // row-major operand: rows ld elements apart, a K-step advances 32 elements; K32-panel operand
  // ([K/32][ld rows][32]): rows 32 elements apart, a K-step advances one whole panel (ld * 32)
  const char* srcA[C::PA];
  const char* srcW[C::PW];
  int ldsA[C::PA], ldsW[C::PW];
  const int64_t a_row = g.a_panel ? 32 : g.lda, w_row = g.w_panel ? 32 : g.ldw;
  const int64_t kstepA = g.a_panel ? g.lda * 64 : 64, kstepW = g.w_panel ? g.ldw * 64 : 64;
#pragma unroll
  for (int j = 0; j < C::PA; ++j) ldsA[j] = ((wave * C::PA + j) % C::APIECES) * 16 * 64;
#pragma unroll
  for (int j = 0; j < C::PW; ++j) ldsW[j] = (wave * C::PW + j) * 16 * 64;
  constexpr bool BUFDMA = (DBG & 16384) != 0;
  static_assert(!BUFDMA || ((C::APIECES % C::NW == 0 || C::PA == 1) && C::PA * 1024 <= 4096 && C::PW * 1024 <= 4096),
                "buffer DMA: a wave's pieces of a stage must be consecutive (immediate offsets of 1 KiB, 12 bits)");
  BufDma bd;
  if constexpr (BUFDMA) {
    const int rl = lane >> 2, lc = (lane & 3) ^ GSW[(rl >> 2) & 3];
    bd.va = (((wave * C::PA) % C::APIECES) * 16 + rl) * 64 + lc * 16;
    bd.vw = ((wave * C::PW) * 16 + rl) * 64 + lc * 16;
    bd.ka = (int)(g.lda * 64);
    bd.kw = (int)(g.ldw * 64);
  }
  auto set_sources = [&](int tile) {
    const int b2 = xcd_remap(tile, g.ntiles);
    const int64_t tm0 = (int64_t)(b2 / tiles_n) * C::BM;
    const int tn0 = (b2 % tiles_n) * C::BN;
    if constexpr (BUFDMA) {   // (panel operands: row r of panel 0 at byte 64 r; everything here is wave-uniform)
      const int64_t offA = ((int64_t)blockIdx.y * g.sA) * 2 + tm0 * 64, offW = ((int64_t)blockIdx.y * g.sW) * 2 + (int64_t)tn0 * 64;
      // bound = the rows this operand OWNS in its last panel (g.A may be a row window of a larger panel buffer: lda > M, first row > 0): rows
      // beyond M / N of the last panel are zero-filled by the descriptor instead of being fetched from behind the allocation
      const int64_t bytesA = (int64_t)(g.K / 32 - 1) * g.lda * 64 + (g.M - tm0) * 64, bytesW = (int64_t)(g.K / 32 - 1) * g.ldw * 64 + ((int64_t)g.N - tn0) * 64;
      bd.ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(g.A)) + offA, 0, (int)bytesA, 0x00020000);
      bd.rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(g.W)) + offW, 0, (int)bytesW, 0x00020000);
      return;
    }
    const int rl = lane >> 2, pc = lane & 3;
    const int lc = pc ^ GSW[(rl >> 2) & 3];          // logical chunk stored at this physical slot
#pragma unroll
    for (int j = 0; j < C::PA; ++j) {
      int64_t ra = tm0 + ((wave * C::PA + j) % C::APIECES) * 16 + rl; if (ra >= g.M) ra = g.M - 1;
      srcA[j] = reinterpret_cast<const char*>(g.A) + ((int64_t)blockIdx.y * g.sA + ra * a_row + lc * 8) * 2;
    }
#pragma unroll
    for (int j = 0; j < C::PW; ++j) {
      int rw = tn0 + (wave * C::PW + j) * 16 + rl; if (rw >= g.N) rw = g.N - 1;
      srcW[j] = reinterpret_cast<const char*>(g.W) + ((int64_t)blockIdx.y * g.sW + (int64_t)rw * w_row + lc * 8) * 2;
    }
  };
  if constexpr (EPI == 3) {
    float* vecs = reinterpret_cast<float*>(smem + C::NST * C::STAGE + C::BM * C::WN * 4);
    for (int c = tid; c < C::BN; c += C::THREADS) {
      vecs[c] = g.bias[c];
      vecs[C::BN + c] = g.ln_gamma[c];
      vecs[2 * C::BN + c] = g.ln_beta[c];
    }
    __syncthreads();
  }
  int pre = 0;   // 1 / 2: this tile's first stages were issued before the previous tile's epilogue (2: a full tile's, see big_mainloop)
  // persistent: after a tile's main loop the ring is idle, so the next tile's first stages are put in flight
  // BEFORE the epilogue: their latency (an HBM miss for the A rows) hides behind the stores
  auto prefetch_next = [&](int vt, bool full_tile) {
    const int vn = vt + (int)gridDim.x;
    pre = 0;
    if constexpr (EPI == 3) return;   // the row-statistics epilogue has no registers to spare for the carried pointers
    if constexpr (DEFER) return;      // (its epilogue holds the row statistics: no registers to spare either)
    if (g.dbg & 32) return;           // A/B: no prefetch across the epilogue
    if (vn < g.ntiles) {
      set_sources(vn);
      if constexpr (!C::PP) {   // (the ping-pong loop ends on a barrier that every fragment read precedes)
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
      }
      const int npro = nk < C::PRO ? nk : C::PRO;
      for (int st = 0; st < npro; ++st) {
        if constexpr (BUFDMA) issue_stage_buf<C>(smem, bd, ldsA, ldsW, st);
        else issue_stage<C, DBG>(smem, srcA, srcW, ldsA, ldsW, st, kstepA, kstepW);
      }
      pre = (full_tile && nk >= C::NST && !(g.dbg & 64)) ? 2 : 1;   // (dbg bit 64: A/B, always drain)
    }
  };
  // DEFER: (mean, rstd) of the tile's A rows / residual rows from the producers' partial sums, staged in LDS for every wave
  float2* lds_a = reinterpret_cast<float2*>(smem + C::NST * C::STAGE);
  float2* lds_r = lds_a + C::BM;
  float2* lds_o = lds_r + C::BM;      // [BM][WN]
  auto stage_row_stats = [&](int64_t m0) {
    if constexpr (DA || DR) {
      for (int t = tid; t < C::BM; t += C::THREADS) {
        int64_t row = m0 + t; if (row >= g.M) row = g.M - 1;
        if constexpr (DA) {
          float s1 = 0.f, s2 = 0.f;
          for (int sl = 0; sl < g.d.a_slots; ++sl) { const float2 p = reinterpret_cast<const float2*>(g.d.a_stats)[row * g.d.a_slots + sl]; s1 += p.x; s2 += p.y; }
          const float mean = s1 * g.d.inv_h, var = fmaxf(s2 * g.d.inv_h - mean * mean, 0.f);
          lds_a[t] = float2{mean, 1.0f / sqrtf(var + g.d.eps)};
        }
        if constexpr (DR) {
          float s1 = 0.f, s2 = 0.f;
          for (int sl = 0; sl < g.d.r_slots; ++sl) { const float2 p = reinterpret_cast<const float2*>(g.d.r_stats)[row * g.d.r_slots + sl]; s1 += p.x; s2 += p.y; }
          const float mean = s1 * g.d.inv_h, var = fmaxf(s2 * g.d.inv_h - mean * mean, 0.f);
          lds_r[t] = float2{mean, 1.0f / sqrtf(var + g.d.eps)};
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  };
  for (int vt = blockIdx.x; vt < g.ntiles; vt += gridDim.x) {
  if (!pre) {
    if (vt != (int)blockIdx.x) {   // the ring is reused: every wave must be done reading the previous tile's last stage
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
    }
    set_sources(vt);
  }
  const int bid = xcd_remap(vt, g.ntiles);
  const int64_t m0 = (int64_t)(bid / tiles_n) * C::BM;
  const int n0 = (bid % tiles_n) * C::BN;
  const int frag_off = fr * 64 + ((fg ^ GSW[(fr >> 2) & 3]) << 4);
  const int a_off = wm * (TI * 16 * 64) + frag_off;
  const int wcol0 = n0 + wn * (TJ * 16);
  const int64_t wrow0 = m0 + wm * (TI * 16);
  const bool v_wave = (EPI == 1) && (wcol0 / g.H == 2);
  const bool full_tile = m0 + C::BM <= g.M && n0 + C::BN <= g.N;   // every lane stores every element: the store count per wave is known
  int b_offs[TJ];
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int jj = j & 3;
    const int row = (j >> 2) * 64 + (v_wave ? 16 * jj + fr : 32 * (jj >> 1) + 8 * (fr >> 2) + 4 * (jj & 1) + (fr & 3));
    b_offs[j] = wn * (TJ * 16 * 64) + row * 64 + ((fg ^ GSW[(row >> 2) & 3]) << 4);
  }

  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if constexpr (DA || DR) {
    // deferred LayerNorm: the tile's first DMA stages go out first and the row statistics (a global round trip + a barrier) are
    // fetched underneath them, instead of standing between the main loop and the epilogue
    if (!pre) {
      const int npro = nk < C::PRO ? nk : C::PRO;
      for (int st = 0; st < npro; ++st) {
        if constexpr (BUFDMA) issue_stage_buf<C>(smem, bd, ldsA, ldsW, st);
        else issue_stage<C, DBG>(smem, srcA, srcW, ldsA, ldsW, st, kstepA, kstepW);
      }
      pre = 1;
    }
    stage_row_stats(m0);
  }
  // lane's 8 consecutive output columns for (64-column group q, half h): wcol0 + 64q + 32h + 8fg, values
  // acc[i][4q + 2h + (e>>2)][e&3], e = 0..7
  if constexpr (EPI == 1) {
    const int which = wcol0 / g.H;   // wave-uniform: 0 q, 1 k, 2 v
    const int M32 = (int)g.M, r0 = (int)wrow0;
    if (which == 2) {
      run_mainloop<C, false, DBG, TI * TJ>(acc, smem, srcA, srcW, ldsA, ldsW, nk, a_off, b_offs, kstepA, kstepW, wm, pre, nullptr, &bd);
      prefetch_next(vt, full_tile);
      // acc[i][j][r] = D[m = 16i + 4fg + r][n = 16j + fr]: 4 consecutive tokens per lane -> V^T rows
      // FULL (interior tile, wave-uniform): no per-lane guards, so the epilogue is straight-line code.  With divergent guards
      // hipcc cannot prove the bias loads complete on every path and puts `s_waitcnt vmcnt(0)` in front of EVERY store block,
      // which also waits for the previous store: the tile's stores then leave one round trip at a time.
      auto epi_v = [&](auto fullc) {
        constexpr bool FULL = decltype(fullc)::value;
        bf16* dst = reinterpret_cast<bf16*>(g.vt);
        float bv[TJ], c1v[TJ];
        int64_t coloff[TJ];
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          const int col = wcol0 + 16 * j + fr;
          const int cc = (FULL || col < g.N) ? col : g.N - 1;
          bv[j] = g.bias[cc];
          if constexpr (DA) c1v[j] = g.d.c1[cc];
          const int c = cc - 2 * g.H, head = c / g.dh, d = c % g.dh;
          coloff[j] = (FULL || col < g.N) ? ((int64_t)head * g.dh + d) * g.L : -1;
        }
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          const int row = r0 + 16 * i + 4 * fg;
          if (FULL || row < M32) {
            const int b = row / g.L;
            int l = row - b * g.L;
            if (g.vt_perm) l = (l & ~15) | ((((l >> 3) & 1) | ((l >> 1) & 2)) << 2);   // 4-token group 0,1,2,3 -> 0,2,1,3
            bf16* base = dst + (int64_t)b * g.H * g.L + l;
            float mu[4] = {0.f, 0.f, 0.f, 0.f}, rsd[4] = {1.f, 1.f, 1.f, 1.f};
            if constexpr (DA) {   // rows 16 i + 4 fg + r of the tile: four consecutive (mean, rstd) pairs
              const f32x4* sp = reinterpret_cast<const f32x4*>(lds_a + wm * (TI * 16) + 16 * i + 4 * fg);
              const f32x4 s01 = sp[0], s23 = sp[1];
              mu[0] = s01[0]; rsd[0] = s01[1]; mu[1] = s01[2]; rsd[1] = s01[3];
              mu[2] = s23[0]; rsd[2] = s23[1]; mu[3] = s23[2]; rsd[3] = s23[3];
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
              if (FULL || coloff[j] >= 0) {
                bf16x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  if constexpr (DA) v[r] = (bf16)fmaf(rsd[r], fmaf(-mu[r], c1v[j], acc[i][j][r]), bv[j]);
                  else v[r] = (bf16)(acc[i][j][r] + bv[j]);
                }
                *reinterpret_cast<bf16x4*>(base + coloff[j]) = v;
              }
            }
          }
        }
      };
      if (full_tile) epi_v(std::true_type{}); else epi_v(std::false_type{});
    } else {
      run_mainloop<C, true, DBG, TI * (TJ / 2)>(acc, smem, srcA, srcW, ldsA, ldsW, nk, a_off, b_offs, kstepA, kstepW, wm, pre, nullptr, &bd);
      prefetch_next(vt, full_tile);
      auto epi_qk = [&](auto fullc) {
        constexpr bool FULL = decltype(fullc)::value;
        bf16* dst = reinterpret_cast<bf16*>(which == 0 ? g.q : g.k);
        float bv[TJ / 2][8];          // every bias load before the first store: a load behind a store would wait for it
        float c1v[TJ / 2][8];
        float mu[TI], rsd[TI];
        const bool scale_q = which == 0 && g.q_scale != 0.f;   // (wave-uniform)
#pragma unroll
        for (int qh = 0; qh < TJ / 2; ++qh) {
          const int col = wcol0 + 32 * qh + 8 * fg;
          load8(g.bias + ((FULL || col < g.N) ? col : 0), bv[qh]);
          if constexpr (DA) load8(g.d.c1 + ((FULL || col < g.N) ? col : 0), c1v[qh]);
        }
        if constexpr (DA) {
#pragma unroll
          for (int i = 0; i < TI; ++i) {
            const float2 st = lds_a[wm * (TI * 16) + 16 * i + fr];
            mu[i] = st.x; rsd[i] = st.y;
          }
        }
#pragma unroll
        for (int qh = 0; qh < TJ / 2; ++qh) {
          const int col = wcol0 + 32 * qh + 8 * fg;
          if (FULL || col < g.N) {
            const int c = col - which * g.H, head = c / g.dh, d = c % g.dh;
            const int64_t coloff = (int64_t)head * g.L * g.dh + d;
#pragma unroll
            for (int i = 0; i < TI; ++i) {
              const int row = r0 + 16 * i + fr;
              if (FULL || row < M32) {
                const int b = row / g.L, l = row - b * g.L;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  if constexpr (DA) v[e] = fmaf(rsd[i], fmaf(-mu[i], c1v[qh][e], acc[i][2 * qh + (e >> 2)][e & 3]), bv[qh][e]);
                  else v[e] = acc[i][2 * qh + (e >> 2)][e & 3] + bv[qh][e];
                }
                if (scale_q) {
#pragma unroll
                  for (int e = 0; e < 8; ++e) v[e] *= g.q_scale;
                }
                if constexpr ((DBG & 32) != 0) store8(dst + ((int64_t)b * g.nh * g.L + l) * g.dh + coloff, v);
                else store8_nt(dst + ((int64_t)b * g.nh * g.L + l) * g.dh + coloff, v);
              }
            }
          }
        }
      };
      if (full_tile) epi_qk(std::true_type{}); else epi_qk(std::false_type{});
    }
  } else {
    // stores per wave of a full tile: one 16-byte store per (row tile, 32-column half); a second one with pre_out
    run_mainloop<C, true, DBG, TI * (TJ / 2)>(acc, smem, srcA, srcW, ldsA, ldsW, nk, a_off, b_offs, kstepA, kstepW, wm, pre,
                                              reinterpret_cast<unsigned*>(g.out) + 64 + ((int64_t)bid * C::NW + wave) * 8, &bd);
    prefetch_next(vt, full_tile && !g.pre_out && !g.out_f32);
    bf16* outT = reinterpret_cast<bf16*>(g.out) + (int64_t)blockIdx.y * g.sO;
    float* outF = reinterpret_cast<float*>(g.out) + (int64_t)blockIdx.y * g.sO;
    const bf16* res = g.residual ? reinterpret_cast<const bf16*>(g.residual) + (int64_t)blockIdx.y * g.sR : nullptr;
    if constexpr ((DBG & 4) != 0) {
      float sacc = 0.f;
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (sacc == 12345.678f) outF[0] = sacc;
      continue;
    }
    if constexpr (EPI == 3) {
      // ---- bias + residual, then LayerNorm over the complete row (the block owns all N columns): two-pass
      // statistics, in-lane -> across the 4 lanes of a row (xor 16, 32) -> across the WN waves through LDS.
      // Every global load of the epilogue (the residual rows) is consumed before the first store, and bias / gain / shift come
      // from LDS: with a global load pending behind divergent row guards hipcc puts `s_waitcnt vmcnt(0)` in front of every
      // store, which also waits for the previous store - the tile's stores then leave one round trip at a time.
      {
        constexpr bool FULL = false;
        float* red = reinterpret_cast<float*>(smem + C::NST * C::STAGE);   // [BM][WN] floats, reused for both passes
        const float* vecs = red + C::BM * C::WN;                            // bias | gamma | beta of the BN columns
        // The residual rows take ONE memory round trip: every wave DMAs its own TI x TJ/2 pieces (lane-linear: a lane's 16 bytes of
        // piece (qh, i) are exactly the 8 columns it holds of row 16 i + fr) into the ring the main loop has just left - all waves
        // are past its last barrier - adds the bias while they fly, and reads them back with ds_read_b128 (no other wave touches them).
        // (Loading them group by group into registers cost four dependent round trips: there are no registers for more at once.)
        char* rbase = smem + ((wave + 1) * RES_W <= RING ? wave * RES_W : RING + C::BM * C::WN * 4 + 3 * C::BN * 4);
        if constexpr (!C::PP) {   // (the ping-pong loop ends on a barrier that every fragment read precedes; the plain loop does not)
          __builtin_amdgcn_s_waitcnt(0xC07F);
          __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int qh = 0; qh < TJ / 2; ++qh) {
          const int col = wcol0 + 32 * qh + 8 * fg;
#pragma unroll
          for (int i = 0; i < TI; ++i) {
            int64_t row = wrow0 + 16 * i + fr; if (!FULL && row >= g.M) row = g.M - 1;
            const int64_t ro = g.r_panel ? ((int64_t)(col >> 5) * g.ldr + row) * 32 + (col & 31) : row * g.ldr + col;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(res + ro),
                                             (__attribute__((address_space(3))) void*)(rbase + (qh * TI + i) * 1024), 16, 0, MH_EPI3_RES_NT ? 2 : 0);
          }
        }
#pragma unroll
        for (int qh = 0; qh < TJ / 2; ++qh) {
          float bv[8];
          load8(vecs + wcol0 + 32 * qh + 8 * fg, bv);
#pragma unroll
          for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[i][2 * qh + (e >> 2)][e & 3] += bv[e];
        }
        if constexpr ((DBG & 64) != 0) {   // train-mode dropout of the dense output, before the residual (same element numbering as EPI 0)
          if (g.drop.thr)                    // (the training build also serves p = 0: it is the one that writes pre_out)
#pragma unroll
          for (int qh = 0; qh < TJ / 2; ++qh) {
            const int col = wcol0 + 32 * qh + 8 * fg;
#pragma unroll
            for (int i = 0; i < TI; ++i) {
              int64_t row = wrow0 + 16 * i + fr; if (row >= g.M) row = g.M - 1;
              const uint32_t km = drop_keep8_at(g.drop, (uint64_t)row * g.N + col);
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float a = acc[i][2 * qh + (e >> 2)][e & 3];
                acc[i][2 * qh + (e >> 2)][e & 3] = (km >> e) & 1u ? a * g.drop.rscale : 0.f;
              }
            }
          }
        }
        wait_vmcnt<0>();
        float rs[TI];
#pragma unroll
        for (int i = 0; i < TI; ++i) rs[i] = 0.f;
        // pre_out (training): the un-normalised rows are kept for the LayerNorm backward, rounded to bf16, and the statistics are taken
        // from the ROUNDED values - what a separate LayerNorm kernel reading that tensor would see
        bf16* preT = (DBG & 64) != 0 ? reinterpret_cast<bf16*>(g.pre_out) : nullptr;   // (compile-time off in the sampling build)
#pragma unroll
        for (int qh = 0; qh < TJ / 2; ++qh) {
          const int col = wcol0 + 32 * qh + 8 * fg;
#pragma unroll
          for (int i = 0; i < TI; ++i) {
            const bf16x8 rraw = *reinterpret_cast<const bf16x8*>(rbase + (qh * TI + i) * 1024 + lane * 16);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[i][2 * qh + (e >> 2)][e & 3] + (float)rraw[e];
            if (preT) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (float)(bf16)v[e];
              const int64_t row = wrow0 + 16 * i + fr;
              const int64_t ldp = g.ldp ? g.ldp : g.ldo;
              if (row < g.M) store8_nt(preT + (g.p_panel ? ((int64_t)(col >> 5) * ldp + row) * 32 + (col & 31) : row * ldp + col), v);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              acc[i][2 * qh + (e >> 2)][e & 3] = v[e];
              rs[i] += v[e];
            }
          }
        }
        const float invN = 1.0f / (float)g.N;
        float mean[TI], rstd[TI];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
          for (int i = 0; i < TI; ++i) {
            float v = rs[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (fg == 0) red[(wm * (TI * 16) + 16 * i + fr) * C::WN + wn] = v;
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
#pragma unroll
          for (int i = 0; i < TI; ++i) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < C::WN; ++w) t += red[(wm * (TI * 16) + 16 * i + fr) * C::WN + w];
            if (pass == 0) {
              mean[i] = t * invN;
              float sq = 0.f;
#pragma unroll
              for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = acc[i][j][r] - mean[i]; sq += d * d; }
              rs[i] = sq;
            } else {
              rstd[i] = 1.0f / sqrtf(t * invN + g.ln_eps);
            }
          }
          __builtin_amdgcn_s_barrier();                    // reads done before the second pass overwrites `red`
        }
#pragma unroll
        for (int qh = 0; qh < TJ / 2; ++qh) {
          const int col = wcol0 + 32 * qh + 8 * fg;
          float gv[8], bt[8];
          load8(vecs + C::BN + col, gv);
          load8(vecs + 2 * C::BN + col, bt);
#pragma unroll
          for (int i = 0; i < TI; ++i) {
            const int64_t row = wrow0 + 16 * i + fr;
            if (FULL || row < g.M) {
              float v[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (acc[i][2 * qh + (e >> 2)][e & 3] - mean[i]) * rstd[i] * gv[e] + bt[e];
              const int64_t oo = g.o_panel ? ((int64_t)(col >> 5) * g.ldo + row) * 32 + (col & 31) : row * g.ldo + col;
              store8(outT + oo, v);   // ordinary store: the next GEMM re-reads these rows (A operand and residual) from L2 / MALL
            }
          }
        }
      }
    } else {
      auto epi_gen = [&](auto fullc) {
        constexpr bool FULL = decltype(fullc)::value;
        float bv[TJ / 2][8];          // every bias load before the first store
        // element offsets as (column part) + (row part), the row part advanced by additions: a 64-bit multiply per store is three
        // quarter-rate instructions, and the epilogue is vector-bound
        const int64_t o_rs = g.o_panel ? 32 : g.ldo, r_rs = g.r_panel ? 32 : g.ldr;     // elements between consecutive rows
        const int64_t o_row0 = (wrow0 + fr) * o_rs, r_row0 = (wrow0 + fr) * r_rs;
        float os1[TI], os2[TI];       // DO: running (sum, sum of squares) of this lane's part of each row
        if constexpr (DO) {
#pragma unroll
          for (int i = 0; i < TI; ++i) { os1[i] = 0.f; os2[i] = 0.f; }
        }
#pragma unroll
        for (int qh = 0; qh < TJ / 2; ++qh) {
          const int col = wcol0 + 32 * qh + 8 * fg;
#pragma unroll
          for (int e = 0; e < 8; ++e) bv[qh][e] = 0.f;
          if (g.bias && (FULL || col < g.N)) {
            if (FULL || col + 8 <= g.N) load8(g.bias + col, bv[qh]);
            else { const f32x4 b4 = *reinterpret_cast<const f32x4*>(g.bias + col); bv[qh][0] = b4[0]; bv[qh][1] = b4[1]; bv[qh][2] = b4[2]; bv[qh][3] = b4[3]; }
          }
        }
        // DR (round 6): the tile's residual rows take ONE round trip through the idle ring (LDS-DMA, lane-linear pieces: a lane's 16 bytes of
        // piece (qh, i) are the 8 columns it holds of row 16 i + fr - the EPI 3 epilogue's scheme) instead of TI x 4 registers per column
        // group: with them the raw-residual + output-statistics kernel (the attention-output and FFN-output dense of every d_model 768 layer)
        // needed 287 registers, spilled 31, and ran 40.2 us where either feature alone runs 29.3 - 30.6 (tools/debug/defer_epilogue_bench.py)
        constexpr bool RSTAGE = DR && !C::PP && C::NW * (TI * (TJ / 2) * 1024) <= C::NST * C::STAGE;
        const char* rstage = smem + wave * (TI * (TJ / 2) * 1024);
        const char* gbstage = smem + C::NW * (TI * (TJ / 2) * 1024) + wave * 1024;
        if constexpr (RSTAGE) {
          __builtin_amdgcn_s_waitcnt(0xC07F);      // (the plain main loop does not end on a barrier: every wave's fragment reads first)
          __builtin_amdgcn_s_barrier();
#pragma unroll
          for (int qh = 0; qh < TJ / 2; ++qh) {
            const int col = wcol0 + 32 * qh + 8 * fg;
            const int cc = (FULL || col + 8 <= g.N) ? col : 0;
            const int64_t r_col = g.r_panel ? (int64_t)(cc >> 5) * g.ldr * 32 + (cc & 31) : cc;
#pragma unroll
            for (int i = 0; i < TI; ++i) {
              int64_t row = wrow0 + 16 * i + fr; if (!FULL && row >= g.M) row = g.M - 1;
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(res + r_col + row * r_rs),
                                               (__attribute__((address_space(3))) void*)const_cast<char*>(rstage + (qh * TI + i) * 1024), 16, 0, 0);
            }
          }
          // ... and the wave's 64 residual gains | shifts ride along as one more piece (lanes 0 - 15 gamma, 16 - 31 beta, the rest repeat):
          // read back per row tile, they are not live across the column group's rows (16 registers fewer)
          static_assert(!RSTAGE || C::NW * (TI * (TJ / 2) * 1024) + C::NW * 1024 <= C::NST * C::STAGE, "no room for the gain / shift pieces");
          {
            const int l32 = lane & 31, gc = wcol0 + 4 * (l32 & 15);
            const float* gsrc = (l32 < 16 ? g.d.r_gamma : g.d.r_beta) + ((FULL || gc + 4 <= g.N) ? gc : 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                             (__attribute__((address_space(3))) void*)const_cast<char*>(gbstage), 16, 0, 0);
          }
          wait_vmcnt<0>();     // (the bias loads above included: nothing the epilogue loads is pending behind its first store)
        }
#pragma unroll
        for (int qh = 0; qh < TJ / 2; ++qh) {
          const int col = wcol0 + 32 * qh + 8 * fg;
          if (FULL || col < g.N) {   // N % 8 == 0, or an fp32 output with N % 8 == 4 (big_tile_ok): at least the first 4 columns are valid
            float c1v[8], rgv[8], rbv[8];   // deferred-LayerNorm column vectors of this group (loaded with its residual rows)
            if constexpr (DA) load8(g.d.c1 + ((FULL || col + 8 <= g.N) ? col : 0), c1v);
            if constexpr (DR && !RSTAGE) { load8(g.d.r_gamma + ((FULL || col + 8 <= g.N) ? col : 0), rgv); load8(g.d.r_beta + ((FULL || col + 8 <= g.N) ? col : 0), rbv); }
            bf16x8 rraw[RSTAGE ? 1 : TI];         // the group's residual rows: all loads in flight together, behind the previous group's stores
            if constexpr (!RSTAGE) if (GEN ? res != nullptr : (FORM & 6) != 0) {
              const int64_t r_col = g.r_panel ? (int64_t)(col >> 5) * g.ldr * 32 + (col & 31) : col;
              int64_t ro = r_col + r_row0;
#pragma unroll
              for (int i = 0; i < TI; ++i) {
                if (FULL || wrow0 + 16 * i + fr < g.M) rraw[i] = *reinterpret_cast<const bf16x8*>(res + ro);
                else rraw[i] = *reinterpret_cast<const bf16x8*>(res + r_col + (g.M - 1) * r_rs);
                ro += 16 * r_rs;
              }
            }
            const int64_t o_col = g.o_panel ? (int64_t)(col >> 5) * g.ldo * 32 + (col & 31) : col;
            int64_t oo = o_col + o_row0 - 16 * o_rs;
#pragma unroll
            for (int i = 0; i < TI; ++i) {
              const int64_t row = wrow0 + 16 * i + fr;
              oo += 16 * o_rs;
              if (FULL || row < g.M) {
                float v[8];
                const int rt = wm * (TI * 16) + 16 * i + fr;     // row of the tile: (mean, rstd) pairs staged in LDS
                if constexpr (DA) {
                  const float2 sa = lds_a[rt];
#pragma unroll
                  for (int e = 0; e < 8; ++e) v[e] = fmaf(sa.y, fmaf(-sa.x, c1v[e], acc[i][2 * qh + (e >> 2)][e & 3]), bv[qh][e]);
                } else {
#pragma unroll
                  for (int e = 0; e < 8; ++e) v[e] = acc[i][2 * qh + (e >> 2)][e & 3] + bv[qh][e];
                }
                if constexpr (ACT != MH_ACT_NONE) {
                  bool done = false;
                  if constexpr (ACT == MH_ACT_GELU_ERF) {
                    if (FORM == 8 || (GEN && !DEFER && g.pre_out && g.pre_kind == 1)) {   // training: the backward gets gelu'(pre), from the same exp / rcp as gelu(pre)
                      float gp[8];
                      gelu_erf_fast8_dgelu(v, gp);
                      store8_nt(reinterpret_cast<bf16*>(g.pre_out) + (int64_t)blockIdx.y * g.sO + oo, gp);
                      done = true;
                    }
                  }
                  if (!done) {
                    if (GEN && !DEFER && g.pre_out) {   // training: the backward needs the pre-activation
                      store8_nt(reinterpret_cast<bf16*>(g.pre_out) + (int64_t)blockIdx.y * g.sO + oo, v);
                    }
                    if constexpr (ACT == MH_ACT_GELU_ERF) gelu_erf_fast8(v);
                    else {
#pragma unroll
                      for (int e = 0; e < 8; ++e) v[e] = apply_act<bf16>(v[e], ACT);
                    }
                  }
                }
                if constexpr ((DBG & 64) != 0) {   // train-mode dropout of the dense output, before the residual
                  const uint32_t km = drop_keep8_at(g.drop, (uint64_t)row * g.N + col);
#pragma unroll
                  for (int e = 0; e < 8; ++e) v[e] = (km >> e) & 1u ? v[e] * g.drop.rscale : 0.f;
                }
                // (the deferred-LayerNorm kernels - launch_big sees to it - have no activation gradient, no fp32 output, and with DR always a
                // residual: compile-time there, so that a group is straight-line code instead of a dozen uniform branches)
                if (DR || (GEN ? res != nullptr : (FORM & 6) != 0)) {
                  bf16x8 rv;
                  if constexpr (RSTAGE) rv = *reinterpret_cast<const bf16x8*>(rstage + (qh * TI + i) * 1024 + lane * 16);
                  else rv = rraw[RSTAGE ? 0 : i];
                  if (GEN && !DEFER && g.act_grad == MH_ACT_GELU_ERF) {          // backward of dense + GELU: dpre = (dY W) o gelu'(pre)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= gelu_erf_grad((float)rv[e]);
                  } else if (FORM == 4 || (GEN && !DEFER && g.act_grad == MH_ACT_DERIV)) {        // the tensor holds act'(pre) already (mh_gemm_bias_act_dact)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= (float)rv[e];
                  } else if (GEN && !DEFER && g.act_grad == MH_ACT_TANH) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float th = tanhf((float)rv[e]); v[e] *= 1.0f - th * th; }
                  } else if constexpr (DR) {
                    const float2 sr = lds_r[rt];
                    if constexpr (RSTAGE) {
                      load8(reinterpret_cast<const float*>(gbstage) + 32 * qh + 8 * fg, rgv);
                      load8(reinterpret_cast<const float*>(gbstage + 256) + 32 * qh + 8 * fg, rbv);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += fmaf(((float)rv[e] - sr.x) * sr.y, rgv[e], rbv[e]);
                  } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
                  }
                }
                if constexpr (DO) {   // statistics of the row as the consumers will read it: from the bf16-rounded values
#pragma unroll
                  for (int e = 0; e < 8; ++e) { const float r = (float)(bf16)v[e]; os1[i] += r; os2[i] += r * r; }
                }
                if (GEN && !DEFER && g.out_f32) {
                  if (FULL || col + 8 <= g.N) store8(outF + oo, v);
                  else *reinterpret_cast<f32x4*>(outF + oo) = f32x4{v[0], v[1], v[2], v[3]};   // N % 8 == 4 tail
                } else {
                  // streaming stores for outputs read once, much later or by a streaming reader; the deferred-LayerNorm producers' raw
                  // rows are re-read at once as A operand and residual: ordinary stores (c2-bertbase -2.0 % step time, A/B of two builds)
                  if constexpr ((DBG & 2048) != 0) {   // timing-only ablation: the epilogue's arithmetic without its stores (values kept live)
                    float keep = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) keep += v[e];
                    if (keep == 12345.678f) store8(outT + oo, v);
                  }
                  else if constexpr ((DBG & 32) != 0 || DO) store8(outT + oo, v); else store8_nt(outT + oo, v);
                }
              }
            }
          }
        }
        if constexpr (DO) {
          {   // fold the lane partials over the 4 lanes of a row, then over the WN column waves (fixed order)
#pragma unroll
            for (int i = 0; i < TI; ++i) {
              float a = os1[i], b = os2[i];
              a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
              b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
              if (fg == 0) lds_o[(wm * (TI * 16) + 16 * i + fr) * C::WN + wn] = float2{a, b};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            for (int t = tid; t < C::BM; t += C::THREADS) {
              float a = 0.f, b = 0.f;
#pragma unroll
              for (int w = 0; w < C::WN; ++w) { const float2 p = lds_o[t * C::WN + w]; a += p.x; b += p.y; }
              if (m0 + t < g.M) reinterpret_cast<float2*>(g.d.o_stats)[(m0 + t) * g.d.o_slots + n0 / C::BN] = float2{a, b};
            }
          }
        }
      };
      if (full_tile) epi_gen(std::true_type{}); else epi_gen(std::false_type{});
    }
  }
  }   // persistent tile loop
}

MH_KNOB(int, g_dbg, 0);
MH_KNOB(int, g_variant, 2);  // bf16 kernel choice: 0 the 128x128 register-staged tile everywhere (the fallback of shapes the big tiles do not serve), 2 big tiles (256x128; 256x256 for row-major launches that fill the chip with it), 4 big 256x256

// compute units of the CURRENT device (cached per device: the library may serve several devices from one process)
int device_cus() {
  static int cus[MH_MAX_DEVICES] = {};
  const int dev = mh_current_device();
  if (!cus[dev]) {
    int n = 0;
    cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cus[dev];
}

#ifndef MH_PLAIN_STORES_DEFAULT
#define MH_PLAIN_STORES_DEFAULT 0
#endif
// round 5: K32-panel launches (the engine's) issue their stage DMA as buffer loads (BufDma above): bit-identical results, no vector address
// arithmetic per piece.  A/B: mh_gemm_set_buf_dma(0) = global_load_lds with per-piece 64-bit addresses (rounds 1 - 4)
MH_KNOB(int, g_buf_dma, 1);
MH_KNOB(int, g_plain_stores, MH_PLAIN_STORES_DEFAULT);   // A/B: bit 0 QKV streaming instead of ordinary stores, bit 1 dense+GELU ordinary instead of streaming stores; bit 2: full-row tile without ping-pong; bits 3 / 4: 64-row full-row tile

template <class C, int EPI>
int launch_big(const GemmArgs& g0, hipStream_t s, int batch) {
  GemmArgs g = g0;
  const int64_t t2 = (int64_t)ceil_div(g.M, C::BM) * ceil_div(g.N, C::BN);
  MH_CHECK_ARG(t2 > 0 && t2 < (1ll << 31), "gemm: bad grid (M=%lld N=%d)", (long long)g.M, g.N);
  // persistent: one block per CU slot walks the tiles (no re-launch, the ring stays allocated)
  const int cus = device_cus();
  const int per_cu = (C::STAGE * C::NST <= 80 * 1024 && C::NW <= 4) ? 2 : 1;
  const int64_t slots = (int64_t)cus * per_cu;
  g.ntiles = (int)t2;
  const dim3 grid((unsigned)(t2 < slots ? t2 : slots), (unsigned)batch), block(C::THREADS);
  mh_prof_note("tile=%dx%d%s epi=%d act=%d M=%lld N=%d K=%d batch=%d", C::BM, C::BN, C::PP ? "pp" : "", EPI, g.act, (long long)g.M, g.N, g.K, batch);
  const bool defer = g.d.a_stats || g.d.r_stats || g.d.o_stats;
  // the stage DMA as buffer loads (BufDma): K32-panel operands (the engine's launches) whose byte extents fit a 32-bit descriptor
  const bool buf_dma = g_buf_dma && g.a_panel && g.w_panel && !(g.dbg & 31) &&
                       (int64_t)(g.K / 32) * g.lda * 64 < (1ll << 31) && (int64_t)(g.K / 32) * g.ldw * 64 < (1ll << 31);
#define MH_LAUNCH_BIG(EPI_, ACT_, BITS_)                                                                             \
  do {                                                                                                                \
    if (buf_dma) MH_LAUNCH((gemm_big_kernel<C, EPI_, ACT_, (BITS_) | 16384>), grid, block, 0, s, g);                  \
    else MH_LAUNCH((gemm_big_kernel<C, EPI_, ACT_, (BITS_)>), grid, block, 0, s, g);                                  \
  } while (0)
  if constexpr (EPI == 1) {
    if (defer) {
      if constexpr (C::NW == 4) {
        MH_CHECK_ARG(g.d.a_stats && !g.d.r_stats && !g.d.o_stats, "gemm_qkv: deferred LayerNorm applies to the A operand only");
        MH_LAUNCH_BIG(1, MH_ACT_NONE, 128);
      } else { mh_set_error("gemm: deferred LayerNorm needs the 256x128 tile"); return MH_ERR_UNSUPPORTED; }
    }
    // q / k leave with ordinary stores: the attention kernel reads them back at once (round 2, after the epilogue restructuring:
    // +0.9 % steps/s over streaming stores, tools/ab_step.py plain_stores 0 1; round 1 had measured the opposite); bit 0 = streaming
    else if (g_plain_stores & 1) MH_LAUNCH((gemm_big_kernel<C, 1, MH_ACT_NONE>), grid, block, 0, s, g);
    else MH_LAUNCH_BIG(1, MH_ACT_NONE, 32);
  } else if constexpr (EPI == 3) {
    if (g.drop.thr || g.pre_out) {   // the training build: dropout (p may be 0) + the un-normalised rows kept for the backward
      if constexpr (C::BN == 512 && C::PP) MH_LAUNCH_BIG(3, MH_ACT_NONE, 64);
      else { mh_set_error("gemm: the dropout + LayerNorm epilogue is built for the 128x512 tile only"); return MH_ERR_UNSUPPORTED; }
    } else MH_LAUNCH_BIG(3, MH_ACT_NONE, 0);
  } else {
    if (g.dbg & 31) {   // timing-only ablations (tools/gemm_bench.py): 1 no DMA, 2 no MFMA, 4 no epilogue, 8 no LDS reads
      switch (g.dbg & 31) {
        case 1: MH_LAUNCH((gemm_big_kernel<C, 0, MH_ACT_NONE, 1>), grid, block, 0, s, g); break;
        case 2: MH_LAUNCH((gemm_big_kernel<C, 0, MH_ACT_NONE, 2>), grid, block, 0, s, g); break;
        case 4: MH_LAUNCH((gemm_big_kernel<C, 0, MH_ACT_NONE, 4>), grid, block, 0, s, g); break;
        case 5: MH_LAUNCH((gemm_big_kernel<C, 0, MH_ACT_NONE, 5>), grid, block, 0, s, g); break;
        case 6: MH_LAUNCH((gemm_big_kernel<C, 0, MH_ACT_NONE, 6>), grid, block, 0, s, g); break;
        case 12: MH_LAUNCH((gemm_big_kernel<C, 0, MH_ACT_NONE, 12>), grid, block, 0, s, g); break;
        case 13: MH_LAUNCH((gemm_big_kernel<C, 0, MH_ACT_NONE, 13>), grid, block, 0, s, g); break;
        case 20: MH_LAUNCH((gemm_big_kernel<C, 0, MH_ACT_NONE, 20>), grid, block, 0, s, g); break;
        case 28: MH_LAUNCH((gemm_big_kernel<C, 0, MH_ACT_NONE, 28>), grid, block, 0, s, g); break;
        case 3: MH_LAUNCH((gemm_big_kernel<C, 0, MH_ACT_GELU_ERF, 2048>), grid, block, 0, s, g); break;      // bias + GELU, no stores
        case 7: MH_LAUNCH((gemm_big_kernel<C, 0, MH_ACT_GELU_ERF, 2049>), grid, block, 0, s, g); break;      // ... and no stage DMA
        case 9: MH_LAUNCH((gemm_big_kernel<C, 0, MH_ACT_NONE, 2048>), grid, block, 0, s, g); break;          // bias only, no stores
        case 10: MH_LAUNCH((gemm_big_kernel<C, 0, MH_ACT_NONE, 4096 + 4>), grid, block, 0, s, g); break;     // no epilogue, no W pieces
        default: MH_LAUNCH((gemm_big_kernel<C, 0, MH_ACT_NONE, 14>), grid, block, 0, s, g); break;
      }
    } else if (defer) {
      if constexpr (C::NW == 4) {   // the operand combinations a post-LN encoder layer needs (engine.hip)
        const int da = g.d.a_stats ? 1 : 0, dr = g.d.r_stats ? 1 : 0, dd = g.d.o_stats ? 1 : 0;
        MH_CHECK_ARG(!g.act_grad && !g.out_f32 && !g.pre_out && (!dr || g.residual), "gemm: a deferred-LayerNorm launch has no activation gradient, fp32 or second output, and a raw residual needs the residual");
        if (da && !dr && !dd && g.act == MH_ACT_GELU_ERF) MH_LAUNCH_BIG(0, MH_ACT_GELU_ERF, 128);   // FFN1
        else if (da && !dr && !dd && g.act == MH_ACT_TANH) MH_LAUNCH_BIG(0, MH_ACT_TANH, 128);      // (down-projection)
        else if (!da && !dr && dd && g.act == MH_ACT_NONE) MH_LAUNCH_BIG(0, MH_ACT_NONE, 512);      // first attention-output dense
        else if (!da && dr && dd && g.act == MH_ACT_NONE) MH_LAUNCH_BIG(0, MH_ACT_NONE, 768);       // dense + raw residual -> raw rows
        else if (!da && dr && !dd && g.act == MH_ACT_NONE) MH_LAUNCH_BIG(0, MH_ACT_NONE, 256);      // last FFN output dense
        else { mh_set_error("gemm: unsupported deferred-LayerNorm operand combination (a=%d r=%d o=%d act=%d)", da, dr, dd, g.act); return MH_ERR_UNSUPPORTED; }
      } else { mh_set_error("gemm: deferred LayerNorm needs the 256x128 tile"); return MH_ERR_UNSUPPORTED; }
    } else {
      // the epilogue's form, where it is one of the fixed ones (see gemm_big_kernel: FORM); everything else takes the generic epilogue
      int form = 0;
      if (!g.out_f32 && !g.drop.thr) {
        if (!g.residual && !g.act_grad && !g.pre_out) form = 1;
        else if (g.residual && !g.act_grad && !g.pre_out) form = 2;
        else if (g.residual && g.act_grad == MH_ACT_DERIV && !g.pre_out) form = 4;
        else if (!g.residual && !g.act_grad && g.pre_out && g.pre_kind == 1 && g.act == MH_ACT_GELU_ERF) form = 8;
      }
      switch (g.act) {
      case MH_ACT_TANH:
        if (form == 1) MH_LAUNCH_BIG(0, MH_ACT_TANH, 1 << 16);
        else MH_LAUNCH_BIG(0, MH_ACT_TANH, 0);
        break;
      case MH_ACT_GELU_ERF:
        if (g_plain_stores & 2) MH_LAUNCH((gemm_big_kernel<C, 0, MH_ACT_GELU_ERF, 32>), grid, block, 0, s, g);
        else if (form == 1) MH_LAUNCH_BIG(0, MH_ACT_GELU_ERF, 1 << 16);
        else if (form == 8) MH_LAUNCH_BIG(0, MH_ACT_GELU_ERF, 8 << 16);
        else MH_LAUNCH_BIG(0, MH_ACT_GELU_ERF, 0);
        break;
      case MH_ACT_SILU: MH_LAUNCH((gemm_big_kernel<C, 0, MH_ACT_SILU>), grid, block, 0, s, g); break;
      default:
        if (g.drop.thr) MH_LAUNCH((gemm_big_kernel<C, 0, MH_ACT_NONE, 64>), grid, block, 0, s, g);
        else if (form == 1) MH_LAUNCH_BIG(0, MH_ACT_NONE, 1 << 16);
        else if (form == 2) MH_LAUNCH_BIG(0, MH_ACT_NONE, 2 << 16);
        else if (form == 4) MH_LAUNCH_BIG(0, MH_ACT_NONE, 4 << 16);
        else MH_LAUNCH_BIG(0, MH_ACT_NONE, 0);
        break;
      }
    }
  }
#undef MH_LAUNCH_BIG
  MH_CHECK_LAUNCH();
  return MH_OK;
}

bool big_tile_ok(const GemmArgs& g) {
  // N % 8 == 0: a lane's 8 output columns are all valid; an fp32 row-major output may end on a half group (N % 4 == 0, e.g. the
  // released checkpoints' E = 500 down-projection): the epilogue stores only the first four of the last lane's columns
  const bool n_ok = g.N % 8 == 0 || (g.out_f32 && g.N % 4 == 0 && g.act_grad == 0 && !g.residual && !g.pre_out && !g.q && !g.ln_gamma && !g.drop.thr);
  return n_ok && g.K % B2K == 0 && (g.o_panel || g.ldo % 8 == 0 || (g.out_f32 && g.ldo % 4 == 0)) &&
         (g.r_panel || g.ldr % 8 == 0);
}

// The wide (256x256, one block per CU) tile moves 1.5x fewer operand bytes per flop, but measured inside the captured step
// (tools/ab_step.py, two graph branches) the 256x128 tile is 1.2% faster: at two blocks per CU the blocks of two
// concurrently running kernels share a CU, which is what the branches are for.  Wide / ping-pong stay selectable (4 / 5).
// Round 3: launches with ROW-MAJOR operands (the training tape, direct callers: one GEMM on the chip at a time) take the wide tile
// when it still gives every CU a block - bench.py --workload train -2.1 % (tools/ab_train.py gemm_variant 2 4); the engine's
// K32-panel launches (two concurrent branches) keep the 256x128 tile.  mh_gemm_set_auto_wide(0) switches the rule off.
MH_KNOB(int, g_auto_wide, 1);
MH_KNOB(int, g_wide_roles, 0);   // A/B (mh_gemm_set_wide_roles): panel launches on the 256x256 tile by role: bit 0 dense + GELU, bit 1 QKV scatter, bit 2 the rest
bool want_wide(const GemmArgs& g, int batch) {
  if (g.N % 256 != 0) return false;
  if (g_variant >= 4) return true;
  if (g_wide_roles && (g.a_panel || g.w_panel)) {
    const int role = g.act == MH_ACT_GELU_ERF ? 1 : (g.q ? 2 : 4);
    if (g_wide_roles & role) return true;
  }
  if (g_variant != 2 || !g_auto_wide || batch != 1) return false;
  if (g.a_panel || g.w_panel || g.o_panel || g.r_panel || g.q) return false;
  return (int64_t)ceil_div(g.M, 256) * (g.N / 256) >= device_cus();
}

#include "gemm_strip.h"
// round 6: dense + bias + GELU of K32 panels (the sampler's FFN1) on the column-strip kernel; A/B: mh_gemm_set_strip(0) = gemm_big_kernel
MH_KNOB(int, g_strip, 1);

template <int EPI>
int launch(const GemmArgs& g, int dtype, hipStream_t s, int batch = 1) {
  const int64_t tiles = (int64_t)ceil_div(g.M, BM) * ceil_div(g.N, BN);
  MH_CHECK_ARG(tiles > 0 && tiles < (1ll << 31), "gemm: bad grid (M=%lld N=%d)", (long long)g.M, g.N);
  MH_CHECK_ARG(batch >= 1 && batch <= 65535, "gemm: batch %d out of range", batch);
  dim3 grid((unsigned)tiles, (unsigned)batch), block(256);
  mh_prof_note("tile=128x128 epi=%d act=%d M=%lld N=%d K=%d batch=%d dtype=%d", EPI, g.act, (long long)g.M, g.N, g.K, batch, dtype);
  if (dtype == MH_BF16) {
    MH_CHECK_ARG((g.a_panel || g.lda % 8 == 0) && (g.w_panel || g.ldw % 8 == 0), "gemm(bf16): lda/ldw must be multiples of 8");
    const bool any_panel = g.a_panel || g.w_panel || g.o_panel || g.r_panel;
    const bool big_ok = big_tile_ok(g);
    MH_CHECK_ARG(!any_panel || (big_ok && g_variant >= 2), "gemm: panel layouts need the big-tile bf16 kernel");
    MH_CHECK_ARG(g.K > 0 && (g.K % 64 == 0 || (g.K % B2K == 0 && big_ok && g_variant >= 2)),
                 "gemm(bf16): K=%d must be a positive multiple of 64 (32 with the big-tile kernel)", g.K);
    if (EPI != 2 && g_variant >= 2 && big_ok) {
      if constexpr (EPI == 2) return MH_OK;
      // QKV scatter: a wave's columns must not straddle the q/k/v boundary (H % 64 == 0 for the wide tile)
      else {
        if constexpr (EPI == 0) {
          if (g_strip && batch == 1 && g_variant == 2 && strip_ok(g)) return launch_strip(g, s);
        }
        if (g.d.a_stats || g.d.r_stats || g.d.o_stats) return launch_big<CfgStd, EPI>(g, s, batch);   // deferred LayerNorm: 256x128 only
        if (want_wide(g, batch) && (EPI != 1 || g.H % 64 == 0)) return launch_big<CfgWide, EPI>(g, s, batch);
        return launch_big<CfgStd, EPI>(g, s, batch);
      }
    } else {
      MH_LAUNCH((gemm_kernel<bf16, EPI>), grid, block, 0, s, g);
    }
  } else if (dtype == MH_F32) {
    MH_CHECK_ARG(!(g.a_panel || g.w_panel || g.o_panel || g.r_panel), "gemm(f32): panel layouts are bf16 only");
    MH_CHECK_ARG(g.K % 16 == 0 && g.K > 0, "gemm(f32): K=%d must be a positive multiple of 16", g.K);
    MH_CHECK_ARG(g.lda % 4 == 0 && g.ldw % 4 == 0, "gemm(f32): lda/ldw must be multiples of 4");
    MH_LAUNCH((gemm_kernel<float, EPI>), grid, block, 0, s, g);
  } else {
    MH_CHECK_ARG(false, "gemm: unknown dtype %d", dtype);
  }
  MH_CHECK_LAUNCH();
  return MH_OK;
}

#ifdef MH_ABLATE
#ifndef MH_CARRY_VALU
#define MH_CARRY_VALU 6
#endif
#include "gemm_carry.h"
#endif

}  // namespace

int mh_drop_args(const mh_dropout* d, DropArgs* out);
extern "C" int mh_gemm_bias_res_ln_supported(int N) { return N == 128 || N == 256 || N == 512; }

#ifdef MH_ABLATE
extern "C" int mh_gemm_set_plain_stores(int mask) {
  g_plain_stores = mask;
  return MH_OK;
}
#endif

#ifdef MH_ABLATE
extern "C" int mh_gemm_set_strip(int on) {
  g_strip = on != 0;
  return MH_OK;
}
#endif

#ifdef MH_ABLATE
extern "C" int mh_gemm_set_buf_dma(int on) {
  g_buf_dma = on != 0;
  return MH_OK;
}
#endif

#ifdef MH_ABLATE
extern "C" int mh_gemm_set_debug(int bits) {
  g_dbg = bits & 127;
  return MH_OK;
}
#endif

#ifdef MH_ABLATE
extern "C" int mh_gemm_set_wide_roles(int mask) {
  g_wide_roles = mask;
  return MH_OK;
}
#endif
#ifdef MH_ABLATE
extern "C" int mh_gemm_set_auto_wide(int on) {
  g_auto_wide = on ? 1 : 0;
  return MH_OK;
}
#endif

#ifdef MH_ABLATE
extern "C" int mh_gemm_set_variant(int variant) {
  MH_CHECK_ARG(variant == 0 || variant == 2 || variant == 4, "gemm_set_variant: variant must be 0, 2 or 4");
  g_variant = variant;
  return MH_OK;
}
#endif

#ifdef MH_ABLATE
// experiment (gemm_carry.h): dense + bias + GELU of K32-panel operands into a K32-panel output with the previous tile's epilogue carried
extern "C" int mh_gemm_ffn1_carry(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out, int64_t ldo,
                                  int64_t M, int N, int K, int variant, mh_stream_t stream) {
  MH_CHECK_ARG(A && W && bias && out && M > 0 && N > 0, "gemm_ffn1_carry: null pointer / empty problem");
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.out = out; g.ldo = ldo; g.ldr = 8;
  g.M = M; g.N = N; g.K = K; g.act = MH_ACT_GELU_ERF; g.a_panel = g.w_panel = g.o_panel = 1;
  return launch_carry(g, variant, (hipStream_t)stream);
}
#endif

extern "C" int mh_gemm_bias_act(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                                const void* residual, int64_t ldr, void* out, int64_t ldo, int out_f32,
                                int64_t M, int N, int K, int act, int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(A && W && out, "gemm: null pointer");
  MH_CHECK_ARG(M > 0 && N > 0, "gemm: empty problem M=%lld N=%d", (long long)M, N);
  MH_CHECK_ARG(act >= MH_ACT_NONE && act <= MH_ACT_SILU, "gemm: unknown activation %d", act);
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias;
  g.residual = residual; g.ldr = residual ? ldr : 8; g.out = out; g.ldo = ldo; g.out_f32 = out_f32;
  g.M = M; g.N = N; g.K = K; g.act = act; g.dbg = g_dbg & 127;
  return launch<0>(g, dtype, (hipStream_t)stream);
}

extern "C" int mh_gemm_bias_act_ex(const void* A, int64_t lda, int a_panel, const void* W, int64_t ldw, int w_panel,
                                   const float* bias, const void* residual, int64_t ldr, int r_panel, void* out,
                                   int64_t ldo, int o_panel, int out_f32, int64_t M, int N, int K, int act, int dtype,
                                   mh_stream_t stream) {
  MH_CHECK_ARG(A && W && out, "gemm: null pointer");
  MH_CHECK_ARG(M > 0 && N > 0, "gemm: empty problem M=%lld N=%d", (long long)M, N);
  MH_CHECK_ARG(act >= MH_ACT_NONE && act <= MH_ACT_SILU, "gemm: unknown activation %d", act);
  MH_CHECK_ARG(!(o_panel && out_f32), "gemm: fp32 output is row-major only");
  MH_CHECK_ARG(!o_panel || N % 32 == 0 || true, "gemm: bad panel output");
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias;
  g.residual = residual; g.ldr = residual ? ldr : 8; g.out = out; g.ldo = ldo; g.out_f32 = out_f32;
  g.M = M; g.N = N; g.K = K; g.act = act; g.dbg = g_dbg & 127;
  g.a_panel = a_panel; g.w_panel = w_panel; g.o_panel = o_panel; g.r_panel = residual ? r_panel : 0;
  return launch<0>(g, dtype, (hipStream_t)stream);
}

namespace {
int fill_defer(const mh_ln_defer* d, GemmArgs& g, const char* who) {
  if (!d) return MH_OK;
  MH_CHECK_ARG(d->h_norm > 0 && d->eps >= 0.f, "%s: deferred LayerNorm needs the normalised width and eps", who);
  MH_CHECK_ARG(!d->a_stats || (d->c1 && d->a_slots > 0), "%s: a_stats needs c1 and a_slots", who);
  MH_CHECK_ARG(!d->r_stats || (d->r_gamma && d->r_beta && d->r_slots > 0 && g.residual), "%s: r_stats needs gamma, beta, r_slots and a residual", who);
  MH_CHECK_ARG(!d->o_stats || d->o_slots >= ceil_div(g.N, 128), "%s: o_slots must cover the %d column tiles", who, ceil_div(g.N, 128));
  g.d.a_stats = d->a_stats; g.d.a_slots = d->a_slots; g.d.c1 = d->c1;
  g.d.r_stats = d->r_stats; g.d.r_slots = d->r_slots; g.d.r_gamma = d->r_gamma; g.d.r_beta = d->r_beta;
  g.d.o_stats = d->o_stats; g.d.o_slots = d->o_slots;
  g.d.inv_h = 1.0f / (float)d->h_norm; g.d.eps = d->eps;
  return MH_OK;
}
}  // namespace

// mh_gemm_bias_act_ex in the K32-panel layout (all four operands) with deferred-LayerNorm operands (struct mh_ln_defer):
//   out = act(LN?(A) W^T + bias) [+ LN?(residual)], optionally writing the output rows' partial statistics.
extern "C" int mh_gemm_bias_act_defer(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const void* residual,
                                      int64_t ldr, void* out, int64_t ldo, int64_t M, int N, int K, int act, const mh_ln_defer* defer,
                                      mh_stream_t stream) {
  MH_CHECK_ARG(A && W && out && bias, "gemm_bias_act_defer: null pointer");
  MH_CHECK_ARG(M > 0 && N > 0 && N % 8 == 0 && K % B2K == 0, "gemm_bias_act_defer: bad problem M=%lld N=%d K=%d", (long long)M, N, K);
  MH_CHECK_ARG(act == MH_ACT_NONE || act == MH_ACT_TANH || act == MH_ACT_GELU_ERF, "gemm_bias_act_defer: activation %d", act);
  MH_CHECK_ARG(g_variant >= 2, "gemm_bias_act_defer: needs the big-tile bf16 kernel");
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias;
  g.residual = residual; g.ldr = residual ? ldr : 8; g.out = out; g.ldo = ldo;
  g.M = M; g.N = N; g.K = K; g.act = act;
  g.a_panel = 1; g.w_panel = 1; g.o_panel = 1; g.r_panel = residual ? 1 : 0;
  int rc = fill_defer(defer, g, "gemm_bias_act_defer");
  if (rc) return rc;
  return launch<0>(g, MH_BF16, (hipStream_t)stream);
}

// out = LayerNorm(A W^T + bias + residual) * gamma + beta over complete rows: the block owns all N columns
// (N = 128, 256 or 512), so the normalisation runs on the accumulators (models/network.py:150 ->
// BertSelfOutput / BertOutput: dense -> dropout(eval: identity) -> LayerNorm(hidden + input))
extern "C" int mh_gemm_bias_res_ln(const void* A, int64_t lda, int a_panel, const void* W, int64_t ldw, int w_panel,
                                   const float* bias, const void* residual, int64_t ldr, int r_panel, const float* gamma,
                                   const float* beta, float eps, void* out, int64_t ldo, int o_panel, int64_t M, int N,
                                   int K, mh_stream_t stream) {
  MH_CHECK_ARG(A && W && out && bias && residual && gamma && beta, "gemm_bias_res_ln: null pointer");
  MH_CHECK_ARG(M > 0 && K > 0 && K % B2K == 0, "gemm_bias_res_ln: bad problem M=%lld K=%d", (long long)M, K);
  MH_CHECK_ARG(mh_gemm_bias_res_ln_supported(N), "gemm_bias_res_ln: N=%d must be 128, 256 or 512", N);
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias;
  g.residual = residual; g.ldr = ldr; g.out = out; g.ldo = ldo;
  g.M = M; g.N = N; g.K = K; g.act = MH_ACT_NONE;
  g.a_panel = a_panel; g.w_panel = w_panel; g.o_panel = o_panel; g.r_panel = r_panel;
  g.ln_gamma = gamma; g.ln_beta = beta; g.ln_eps = eps;
  MH_CHECK_ARG((a_panel || lda % 8 == 0) && (w_panel || ldw % 8 == 0) && big_tile_ok(g), "gemm_bias_res_ln: leading dimensions must be multiples of 8");
  hipStream_t s = (hipStream_t)stream;
  if (N == 128) return launch_big<CfgStd, 3>(g, s, 1);
  if (N == 256) return launch_big<CfgWidePP, 3>(g, s, 1);
  // one block per CU: the ping-pong main loop pays here (-4.5% step time, tools/ab_step.py); bit 2 of the A/B mask = plain loop
  if (g_plain_stores & 4) return launch_big<CfgRow, 3>(g, s, 1);
  if ((g_plain_stores & 16) || ((g_plain_stores & 8) && K <= 512)) return launch_big<CfgRow64, 3>(g, s, 1);   // A/B: 64-row full-row tile
  return launch_big<CfgRowPP, 3>(g, s, 1);
}

// out = dropout(A W^T + bias) + residual: the dense half of HF BertSelfOutput / BertOutput in train mode (the LayerNorm that
// follows stays a separate kernel on the training path).  Row-major operands; N % 8 == 0.  The keep mask of element (row, col)
// comes from Philox4x32-7 keyed by drop->seed at counter ((row N + col) / 8, drop->offset) - mh_dropout_fwd with the same
// descriptor re-creates it for the backward - or from drop->mask (tests).
int mh_drop_args(const mh_dropout* d, DropArgs* out);
extern "C" int mh_gemm_bias_dropout_res(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const void* residual,
                                        int64_t ldr, void* out, int64_t ldo, int64_t M, int N, int K, int dtype, const mh_dropout* drop,
                                        mh_stream_t stream) {
  MH_CHECK_ARG(A && W && out, "gemm_bias_dropout_res: null pointer");
  MH_CHECK_ARG(M > 0 && N > 0 && N % 8 == 0, "gemm_bias_dropout_res: bad problem M=%lld N=%d (N must be a multiple of 8)", (long long)M, N);
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias;
  g.residual = residual; g.ldr = residual ? ldr : 8; g.out = out; g.ldo = ldo;
  g.M = M; g.N = N; g.K = K; g.act = MH_ACT_NONE; g.dbg = 0;
  int rc = mh_drop_args(drop, &g.drop);
  if (rc) return rc;
  MH_CHECK_ARG(ldo % 8 == 0 && (!residual || ldr % 8 == 0), "gemm_bias_dropout_res: ldo / ldr must be multiples of 8");
  return launch<0>(g, dtype, (hipStream_t)stream);
}

// out = LayerNorm(pre) with pre = dropout(A W^T + bias) + residual, and pre itself (bf16-rounded, what the LayerNorm backward reads):
// BertSelfOutput / BertOutput in train mode as ONE kernel (N = 512, row-major bf16; drop may be null or p = 0)
extern "C" int mh_gemm_bias_dropout_res_ln(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const void* residual,
                                           int64_t ldr, const float* gamma, const float* beta, float eps, void* pre_out, void* out,
                                           int64_t ldo, int64_t M, int N, int K, const mh_dropout* drop, mh_stream_t stream) {
  MH_CHECK_ARG(A && W && out && pre_out && bias && residual && gamma && beta, "gemm_bias_dropout_res_ln: null pointer");
  MH_CHECK_ARG(M > 0 && K > 0 && K % B2K == 0 && N == 512, "gemm_bias_dropout_res_ln: needs N = 512 and K %% 32 == 0 (got N=%d K=%d)", N, K);
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias;
  g.residual = residual; g.ldr = ldr; g.out = out; g.ldo = ldo; g.pre_out = pre_out;
  g.M = M; g.N = N; g.K = K; g.act = MH_ACT_NONE;
  g.ln_gamma = gamma; g.ln_beta = beta; g.ln_eps = eps;
  int rc = mh_drop_args(drop, &g.drop);
  if (rc) return rc;
  MH_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && ldo % 8 == 0 && ldr % 8 == 0, "gemm_bias_dropout_res_ln: leading dimensions must be multiples of 8");
  return launch_big<CfgRowPP, 3>(g, (hipStream_t)stream, 1);
}

// act(A W^T + bias) -> out AND A W^T + bias -> pre_out in one pass (bf16, row-major, big-tile shapes only): the forward of
// a dense + activation layer under autograd, whose backward needs the pre-activation (training.py:_Linear)
extern "C" int mh_gemm_bias_act_pre(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* pre_out,
                                    void* out, int64_t ldo, int64_t M, int N, int K, int act, mh_stream_t stream) {
  MH_CHECK_ARG(A && W && out && pre_out, "gemm_bias_act_pre: null pointer");
  MH_CHECK_ARG(M > 0 && N > 0 && act > MH_ACT_NONE && act <= MH_ACT_SILU, "gemm_bias_act_pre: bad problem / activation");
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.ldr = 8; g.out = out; g.ldo = ldo;
  g.M = M; g.N = N; g.K = K; g.act = act; g.pre_out = pre_out;
  MH_CHECK_ARG(g_variant >= 2 && lda % 8 == 0 && ldw % 8 == 0 && big_tile_ok(g), "gemm_bias_act_pre: shape not served by the big-tile kernel");
  return launch<0>(g, MH_BF16, (hipStream_t)stream);
}

// mh_gemm_bias_act_pre for GELU with the DERIVATIVE in place of the pre-activation: dact_out = gelu'(A W^T + bias) (bf16), out = gelu(...).
// The backward multiplies by it (mh_gemm_act_grad with act = MH_ACT_DERIV) instead of evaluating gelu' again from the pre-activation.
extern "C" int mh_gemm_bias_act_dact(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* dact_out,
                                     void* out, int64_t ldo, int64_t M, int N, int K, int act, mh_stream_t stream) {
  MH_CHECK_ARG(A && W && out && dact_out, "gemm_bias_act_dact: null pointer");
  MH_CHECK_ARG(M > 0 && N > 0 && act == MH_ACT_GELU_ERF, "gemm_bias_act_dact: bad problem / activation (GELU only)");
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.ldr = 8; g.out = out; g.ldo = ldo;
  g.M = M; g.N = N; g.K = K; g.act = act; g.pre_out = dact_out; g.pre_kind = 1;
  MH_CHECK_ARG(g_variant >= 2 && lda % 8 == 0 && ldw % 8 == 0 && big_tile_ok(g), "gemm_bias_act_dact: shape not served by the big-tile kernel");
  return launch<0>(g, MH_BF16, (hipStream_t)stream);
}

// out = (A W^T) o act'(pre): the input-gradient GEMM of the layer AFTER an activation with the activation's own backward
// folded into its epilogue (bf16 row-major, big-tile shapes; act = tanh or erf-GELU)
extern "C" int mh_gemm_act_grad(const void* A, int64_t lda, const void* W, int64_t ldw, const void* pre, int64_t ld_pre, void* out,
                                int64_t ldo, int64_t M, int N, int K, int act, mh_stream_t stream) {
  MH_CHECK_ARG(A && W && pre && out, "gemm_act_grad: null pointer");
  MH_CHECK_ARG(M > 0 && N > 0 && (act == MH_ACT_TANH || act == MH_ACT_GELU_ERF || act == MH_ACT_DERIV), "gemm_act_grad: bad problem / activation");
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.residual = pre; g.ldr = ld_pre; g.out = out; g.ldo = ldo;
  g.M = M; g.N = N; g.K = K; g.act = MH_ACT_NONE; g.act_grad = act;
  MH_CHECK_ARG(g_variant >= 2 && lda % 8 == 0 && ldw % 8 == 0 && big_tile_ok(g), "gemm_act_grad: shape not served by the big-tile kernel");
  return launch<0>(g, MH_BF16, (hipStream_t)stream);
}

// One descriptor for every bf16 dense launch of the training step (csrc/train_layer.hip): each operand row-major or K32-panel on its own,
// optional second output (pre-activation / act'(pre) / the pre-LayerNorm rows), activation-gradient epilogue, dropout, full-row LayerNorm.
extern "C" int mh_gemm_desc_launch(const mh_gemm_desc* d, mh_stream_t stream) {
  MH_CHECK_ARG(d && d->A && d->W && d->out, "gemm_desc: null pointer");
  MH_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0 && d->K % B2K == 0, "gemm_desc: bad problem M=%lld N=%d K=%d", (long long)d->M, d->N, d->K);
  MH_CHECK_ARG(d->act >= MH_ACT_NONE && d->act <= MH_ACT_SILU, "gemm_desc: unknown activation %d", d->act);
  MH_CHECK_ARG(!(d->o_panel && d->out_f32), "gemm_desc: fp32 output is row-major only");
  MH_CHECK_ARG(g_variant >= 2, "gemm_desc: needs the big-tile bf16 kernel");
  GemmArgs g{};
  g.A = d->A; g.lda = d->lda; g.a_panel = d->a_panel; g.W = d->W; g.ldw = d->ldw; g.w_panel = d->w_panel; g.bias = d->bias;
  g.residual = d->residual; g.ldr = d->residual ? d->ldr : 8; g.r_panel = d->residual ? d->r_panel : 0;
  g.out = d->out; g.ldo = d->ldo; g.o_panel = d->o_panel; g.out_f32 = d->out_f32;
  g.pre_out = d->pre_out; g.ldp = d->ldp; g.p_panel = d->p_panel; g.pre_kind = d->pre_kind;
  g.M = d->M; g.N = d->N; g.K = d->K; g.act = d->act; g.act_grad = d->act_grad;
  int rc = mh_drop_args(d->drop, &g.drop);
  if (rc) return rc;
  MH_CHECK_ARG((g.a_panel || g.lda % 8 == 0) && (g.w_panel || g.ldw % 8 == 0) && big_tile_ok(g), "gemm_desc: leading dimensions must be multiples of 8");
  if (d->ln_gamma) {
    MH_CHECK_ARG(d->ln_beta && d->bias && d->residual && d->pre_out, "gemm_desc: the LayerNorm epilogue needs bias, residual, beta and pre_out");
    MH_CHECK_ARG(d->N == 512 && d->act == MH_ACT_NONE && !d->act_grad && !d->out_f32, "gemm_desc: the dropout + LayerNorm epilogue is built for N = 512");
    MH_CHECK_ARG(d->p_panel || (d->ldp ? d->ldp : d->ldo) % 8 == 0, "gemm_desc: ldp must be a multiple of 8");
    g.ln_gamma = d->ln_gamma; g.ln_beta = d->ln_beta; g.ln_eps = d->ln_eps;
    return launch_big<CfgRowPP, 3>(g, (hipStream_t)stream, 1);
  }
  MH_CHECK_ARG(!d->pre_out || d->act != MH_ACT_NONE, "gemm_desc: pre_out without a LayerNorm needs an activation");
  MH_CHECK_ARG(!d->act_grad || d->residual, "gemm_desc: act_grad reads the stored pre-activation / derivative through `residual`");
  return launch<0>(g, MH_BF16, (hipStream_t)stream);
}

extern "C" int mh_gemm_batched(const void* A, int64_t lda, int64_t strideA, const void* W, int64_t ldw, int64_t strideW,
                               const float* bias, void* out, int64_t ldo, int64_t strideO, int out_f32, int batch, int64_t M,
                               int N, int K, int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(A && W && out, "gemm_batched: null pointer");
  MH_CHECK_ARG(M > 0 && N > 0 && batch > 0, "gemm_batched: empty problem");
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.ldr = 8; g.out = out; g.ldo = ldo; g.out_f32 = out_f32;
  g.M = M; g.N = N; g.K = K; g.act = MH_ACT_NONE; g.sA = strideA; g.sW = strideW; g.sO = strideO;
  return launch<0>(g, dtype, (hipStream_t)stream, batch);
}

extern "C" int mh_gemm_qkv_ex(const void* A, int64_t lda, int a_panel, const void* Wqkv, int64_t ldw, int w_panel,
                              const float* bqkv, void* q, void* k, void* vt, int B, int L, int H, int nh, int dtype,
                              mh_stream_t stream);

extern "C" int mh_gemm_qkv(const void* A, int64_t lda, const void* Wqkv, int64_t ldw, const float* bqkv, void* q,
                           void* k, void* vt, int B, int L, int H, int nh, int dtype, mh_stream_t stream) {
  return mh_gemm_qkv_ex(A, lda, 0, Wqkv, ldw, 0, bqkv, q, k, vt, B, L, H, nh, dtype, stream);
}

namespace { int qkv_impl(const void* A, int64_t lda, int a_panel, const void* Wqkv, int64_t ldw, int w_panel, const float* bqkv,
                         void* q, void* k, void* vt, int B, int L, int H, int nh, int dtype, int vt_perm, mh_stream_t stream,
                         const mh_ln_defer* defer = nullptr, float q_scale = 0.f); }

// mh_gemm_qkv_vtperm (panel operands) whose A rows are raw pre-LayerNorm values (defer->a_stats / c1; bqkv = c2)
extern "C" int mh_gemm_qkv_vtperm_defer(const void* A, int64_t lda, const void* Wqkv, int64_t ldw, const float* c2, void* q, void* k,
                                        void* vt_perm, int B, int L, int H, int nh, const mh_ln_defer* defer, mh_stream_t stream) {
  MH_CHECK_ARG(L % 16 == 0, "gemm_qkv_vtperm_defer: seq_len %d must be a multiple of 16", L);
  MH_CHECK_ARG(g_variant >= 2 && defer && defer->a_stats, "gemm_qkv_vtperm_defer: needs the big-tile bf16 kernel and a_stats");
  return qkv_impl(A, lda, 1, Wqkv, ldw, 1, c2, q, k, vt_perm, B, L, H, nh, MH_BF16, 1, stream, defer);
}

extern "C" int mh_gemm_qkv_ex(const void* A, int64_t lda, int a_panel, const void* Wqkv, int64_t ldw, int w_panel,
                              const float* bqkv, void* q, void* k, void* vt, int B, int L, int H, int nh, int dtype,
                              mh_stream_t stream) {
  return qkv_impl(A, lda, a_panel, Wqkv, ldw, w_panel, bqkv, q, k, vt, B, L, H, nh, dtype, 0, stream);
}

extern "C" int mh_gemm_qkv_vtperm(const void* A, int64_t lda, int a_panel, const void* Wqkv, int64_t ldw, int w_panel,
                                  const float* bqkv, void* q, void* k, void* vt_perm, int B, int L, int H, int nh,
                                  mh_stream_t stream) {
  MH_CHECK_ARG(L % 16 == 0, "gemm_qkv_vtperm: seq_len %d must be a multiple of 16", L);
  MH_CHECK_ARG(g_variant >= 2, "gemm_qkv_vtperm: needs the big-tile bf16 kernel");
  return qkv_impl(A, lda, a_panel, Wqkv, ldw, w_panel, bqkv, q, k, vt_perm, B, L, H, nh, MH_BF16, 1, stream);
}
// the same with the queries stored as (x Wq^T + bq) * q_scale (one rounding, from the fp32 accumulator): q_scale = softmax scale x log2(e)
// is what mh_attention_stream_fwd_prescaled expects; defer may be NULL
extern "C" int mh_gemm_qkv_vtperm_qs(const void* A, int64_t lda, const void* Wqkv, int64_t ldw, const float* bqkv, void* q, void* k,
                                     void* vt_perm, int B, int L, int H, int nh, float q_scale, const mh_ln_defer* defer, mh_stream_t stream) {
  MH_CHECK_ARG(L % 16 == 0, "gemm_qkv_vtperm_qs: seq_len %d must be a multiple of 16", L);
  MH_CHECK_ARG(g_variant >= 2 && q_scale > 0.f, "gemm_qkv_vtperm_qs: needs the big-tile bf16 kernel and a positive scale");
  MH_CHECK_ARG(!defer || defer->a_stats, "gemm_qkv_vtperm_qs: a deferred LayerNorm descriptor needs a_stats");
  return qkv_impl(A, lda, 1, Wqkv, ldw, 1, bqkv, q, k, vt_perm, B, L, H, nh, MH_BF16, 1, stream, defer, q_scale);
}

namespace {
int qkv_impl(const void* A, int64_t lda, int a_panel, const void* Wqkv, int64_t ldw, int w_panel, const float* bqkv,
             void* q, void* k, void* vt, int B, int L, int H, int nh, int dtype, int vt_perm, mh_stream_t stream,
             const mh_ln_defer* defer, float q_scale) {
  MH_CHECK_ARG(A && Wqkv && bqkv && q && k && vt, "gemm_qkv: null pointer");
  MH_CHECK_ARG(H % 64 == 0, "gemm_qkv: hidden size %d must be a multiple of 64", H);
  MH_CHECK_ARG(nh > 0 && H % nh == 0 && (H / nh) % 8 == 0, "gemm_qkv: head dim must be a multiple of 8");
  MH_CHECK_ARG(L % 8 == 0, "gemm_qkv: seq_len %d must be a multiple of 8", L);
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = Wqkv; g.ldw = ldw; g.bias = bqkv; g.ldr = 8; g.ldo = 8;
  g.M = (int64_t)B * L; g.N = 3 * H; g.K = H;
  g.a_panel = a_panel; g.w_panel = w_panel;
  g.q = q; g.k = k; g.vt = vt; g.L = L; g.H = H; g.nh = nh; g.dh = H / nh;
  g.vt_perm = vt_perm;
  g.q_scale = q_scale;
  int rcd = fill_defer(defer, g, "gemm_qkv");
  if (rcd) return rcd;
  MH_CHECK_ARG(!vt_perm || (big_tile_ok(g) && H % 64 == 0), "gemm_qkv_vtperm: shape not served by the big-tile kernel");
  return launch<1>(g, dtype, (hipStream_t)stream);
}
}  // namespace


// ---------------------------------------------------------------- nearest-embedding rounding on the fp32 MFMA
namespace {

#pragma clang fp contract(off)
__global__ void row_sqnorm_f32_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ out, int64_t rows, int E) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < E; c += 64) {
    const float v = x[row * ldx + c];
    s += v * v;
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}

__global__ void argbest_reduce_kernel(const float* __restrict__ pbest, const int32_t* __restrict__ pidx, int nslots,
                                      int32_t* __restrict__ idx_out, int64_t rows) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int s = 0; s < nslots; ++s) {   // slots are in increasing column order: strict > keeps the first index
    const float v = pbest[row * nslots + s];
    const int i = pidx[row * nslots + s];
    if (v > best || (v == best && i < bi)) { best = v; bi = i; }
  }
  idx_out[row] = bi == 0x7fffffff ? 0 : bi;
}

}  // namespace

extern "C" size_t mh_round_workspace_bytes(int64_t n_tokens, int E, int V) {
  const int64_t nslots = 2 * ((V + BN - 1) / BN);
  const int64_t Ep = (E + 15) / 16 * 16;
  size_t b = (size_t)n_tokens * 4;                         // |x_n|^2
  b += (size_t)n_tokens * nslots * 8;                      // partial (score, index)
  if (Ep != E) b += (size_t)n_tokens * Ep * 4;             // zero-padded copy of x
  return b + 1024;
}

// table_pad: [V, E_pad16] fp32 (rows zero-padded to a multiple of 16 columns; == table when E % 16 == 0)
extern "C" int mh_round_to_embedding_mfma(const float* x, const float* table_pad, const float* table_norm, int32_t* idx,
                                          int64_t n_tokens, int E, int V, void* workspace, size_t workspace_bytes,
                                          mh_stream_t stream) {
  MH_CHECK_ARG(x && table_pad && table_norm && idx && workspace, "round_to_embedding_mfma: null pointer");
  MH_CHECK_ARG(n_tokens > 0 && E > 0 && V > 0, "round_to_embedding_mfma: bad shape");
  MH_CHECK_ARG(workspace_bytes >= mh_round_workspace_bytes(n_tokens, E, V), "round_to_embedding_mfma: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int nslots = 2 * ceil_div(V, BN);
  const int Ep = (E + 15) / 16 * 16;
  char* ws = (char*)workspace;
  float* rown = (float*)ws; ws += ((size_t)n_tokens * 4 + 255) & ~(size_t)255;
  float* pbest = (float*)ws; ws += ((size_t)n_tokens * nslots * 4 + 255) & ~(size_t)255;
  int32_t* pidx = (int32_t*)ws; ws += ((size_t)n_tokens * nslots * 4 + 255) & ~(size_t)255;
  const float* xa = x;
  int64_t lda = E;
  if (Ep != E) {
    float* xp = (float*)ws;
    int rc = mh_cast_pad(x, E, xp, Ep, n_tokens, E, n_tokens, MH_F32, stream);
    if (rc) return rc;
    xa = xp; lda = Ep;
  }
  MH_LAUNCH(row_sqnorm_f32_kernel, dim3((unsigned)((n_tokens + 3) / 4)), dim3(256), 0, s, x, (int64_t)E, rown, n_tokens, E);
  MH_CHECK_LAUNCH();
  GemmArgs g{};
  g.A = xa; g.lda = lda; g.W = table_pad; g.ldw = Ep; g.ldr = 8; g.ldo = 8;
  g.M = n_tokens; g.N = V; g.K = Ep;
  g.aux = table_norm; g.rown = rown; g.pbest = pbest; g.pidx = pidx; g.nslots = nslots;
  int rc = launch<2>(g, MH_F32, s);
  if (rc) return rc;
  MH_LAUNCH(argbest_reduce_kernel, dim3((unsigned)((n_tokens + 255) / 256)), dim3(256), 0, s, pbest, pidx, nslots, idx, n_tokens);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// The score GEMM of mh_round_to_embedding_mfma alone: |x_n|^2 comes from the caller (the fused down-projection writes it beside its
// rows, csrc/headtail.hip) and the per-slot winners stay in pbest / pidx [n_tokens][mh_round_slots(V)] for mh_step_epilogue_slots to
// fold - two launches less per batch slice and step.  E % 16 == 0 (no padded copy here).
extern "C" int mh_round_slots(int V) { return 2 * ceil_div(V, BN); }
extern "C" int mh_round_scores(const float* x, const float* x_sqnorm, const float* table_pad, const float* table_norm, float* pbest,
                               int32_t* pidx, int64_t n_tokens, int E, int V, mh_stream_t stream) {
  MH_CHECK_ARG(x && x_sqnorm && table_pad && table_norm && pbest && pidx && n_tokens > 0 && V > 0, "round_scores: bad arguments");
  MH_CHECK_ARG(E > 0 && E % 16 == 0, "round_scores: E = %d must be a multiple of 16", E);
  GemmArgs g{};
  g.A = x; g.lda = E; g.W = table_pad; g.ldw = E; g.ldr = 8; g.ldo = 8;
  g.M = n_tokens; g.N = V; g.K = E;
  g.aux = table_norm; g.rown = x_sqnorm; g.pbest = pbest; g.pidx = pidx; g.nslots = mh_round_slots(V);
  return launch<2>(g, MH_F32, (hipStream_t)stream);
}

// =====================================================================================================
// Weight-gradient GEMM ("TN"): dW[m][n] = sum_k A[k][m] * B[k][n] with BOTH operands stored k-major (A = dY [tokens, out
// features], B = X [tokens, in features], exactly as the forward wrote them) - no transposed copies.  The reduction runs over
// the tokens, the output is tiny, so the token range is cut into `splits` slices (grid.y) that write fp32 partials
// [splits][M][N] for mh_sum_slices.  Tile 256 (m) x 128 (n), 4 waves of 128 x 64, K-step 32 tokens, 3-stage LDS-DMA ring
// as in gemm_big_kernel.  The LDS image keeps the k-major rows ([32 k][256 m] and [32 k][128 n]); MFMA fragments
// (8 consecutive k of one m / n per lane) come out of it through the transposing read ds_read_b64_tr_b16, two per fragment.
// 16-byte chunk c of row k is stored at c ^ f(k), f(k) = 2 ((k & 3) | ((k >> 3 & 1) << 2)) (applied on the DMA source
// address): the 8 rows a 32-lane half reads in one instruction then fall on 8 different 32-byte bank slots.
namespace {

struct TnArgs {
  const bf16* A; int64_t lda;   // [K, lda], columns = m
  const bf16* B; int64_t ldb;   // [K, ldb], columns = n
  float* out;                   // [splits][slice]: M x N products, then (CS) the M column sums of A
  int M, N;
  int64_t Ksteps;               // K steps (32 tokens each) in all; slice s of `splits` runs steps [s Ksteps / splits, (s + 1) Ksteps / splits)
  int64_t slice;                // floats per split slice: M N (+ M)
  int splits, tiles;            // grid = tiles x splits blocks, one-dimensional (see the block mapping in the kernel)
};

typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ int tn_f(int row) { return ((row & 3) | (((row >> 3) & 1) << 2)) << 1; }

// CS > 0: the kernel also writes the column sums of A over its token slice (the bias gradient of the same linear: A = dY) behind
// the M x N products of the slice.  They come out of the matrix pipe - one more MFMA against an all-ones operand gives sum_k A[k][m]
// in every row of the product - and the work is dealt out over the blocks and waves that share a 256-column panel of A: CS = number
// of (n-tile, wave column) workers taking part (2, 4 or 8), worker w sums the 16-column tiles i with i % CS == w.
// WN = wave columns of 64 output columns each: 2 -> the 256 x 128 tile on 4 waves, two blocks per CU; 4 -> a 256 x 256 tile on 8 waves,
// one block per CU - the same waves per SIMD and the same wave tile, but HALF the blocks for the same chip occupancy: the fp32
// partials of a launch (one tile per block: blocks x 128 KiB, whatever the shape) and their fold by mh_sum_slices halve, and both
// operand panels are read once per 256 x 256 outputs.
// PANEL (round 6): both operands as K32 panels [cols / 32][ld rows][32] - the layout every GEMM operand of the training step has since the
// forward and input-gradient GEMMs moved onto the sampler's panel tiles.  A stage keeps the panels apart ([panel][32 k][64 B]: one DMA piece =
// 16 consecutive tokens of one panel = 1 KiB of contiguous memory), the transposing reads address 64-byte rows, and the 32-byte halves of a row
// swap for k rows 8 - 15 / 24 - 31 (on the DMA source address) so that the 8 rows of a 32-lane half fall on 8 different 32-byte bank slots.
template <int CS, int WN = 2, bool PANEL = false>
__global__ __launch_bounds__(128 * WN, WN == 2 ? 2 : 1) void gemm_tn_kernel(const TnArgs g) {
  constexpr int BMt = 256, BNt = 64 * WN, NWt = 2 * WN, NSTt = 3, ASTAGE = 32 * BMt * 2, BSTAGE = 32 * BNt * 2, STAGEt = ASTAGE + BSTAGE;
  constexpr int TIt = 8, TJt = 4;
  constexpr int PAt = 16 / NWt, PBt = (BSTAGE / 1024) / NWt;     // 1-KiB DMA pieces per wave and stage: A 4 / 2, B 2
  constexpr int CPRB = BNt / 8, RPPB = 64 / CPRB;                // B: 16-byte chunks per k-row, k-rows per piece
  __shared__ __attribute__((aligned(16))) char smem[NSTt * STAGEt];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (g.N + BNt - 1) / BNt;
  // XCD-aware block mapping (round 6): the blocks of ONE token slice share its operand panels (every m-tile's A panels are read by all
  // n-tiles and vice versa), and workgroups go to the 8 XCDs round-robin - dealt out as (tile, slice) = (blockIdx.x, blockIdx.y) the 16 tiles
  // of a slice landed on all 8 L2s and every panel was fetched from HBM up to 8 times (rocprofv3 FETCH_SIZE: 313 MB per launch against
  // 167 MB of operands).  xcd_remap gives each XCD a contiguous run of (slice, tile) pairs, i.e. whole slices.
  const int vb = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int slice_i = vb / g.tiles, tile_i = vb % g.tiles;
  const int m0 = (tile_i / tiles_n) * BMt, n0 = (tile_i % tiles_n) * BNt;
  // (uneven slices: the split count is chosen to fill the chip - 21 slices of a 12-tile output on 256 CUs - not to divide the K steps)
  const int64_t ks0 = (int64_t)slice_i * g.Ksteps / g.splits, ks1 = (int64_t)(slice_i + 1) * g.Ksteps / g.splits;
  const int64_t k_begin = ks0 * 32;
  const int nk = (int)(ks1 - ks0);
  const int fr = lane & 15, fg = lane >> 4;

  // DMA: A stage = 16 pieces of (2 k-rows x 512 B); B stage = pieces of (RPPB k-rows x BNt * 2 B)
  const bf16* srcA[PAt];
  const bf16* srcB[PBt];
  if constexpr (PANEL) {
    // piece = (panel of the tile, half of the stage's 32 tokens); lane i lands at token i / 4, physical chunk i % 4
    const int rl = lane >> 2, pc = lane & 3;
#pragma unroll
    for (int j = 0; j < PAt; ++j) {
      const int piece = wave * PAt + j, row = (piece & 1) * 16 + rl, lc = pc ^ (((row >> 3) & 1) << 1);
      int pn = (m0 >> 5) + (piece >> 1);
      if (pn > (g.M >> 5) - 1) pn = (g.M >> 5) - 1;         // M % 32 == 0: clamp whole panels (results beyond M are not stored)
      srcA[j] = g.A + ((int64_t)pn * g.lda + k_begin + row) * 32 + lc * 8;
    }
#pragma unroll
    for (int j = 0; j < PBt; ++j) {
      const int piece = wave * PBt + j, row = (piece & 1) * 16 + rl, lc = pc ^ (((row >> 3) & 1) << 1);
      int pn = (n0 >> 5) + (piece >> 1);
      if (pn > (g.N >> 5) - 1) pn = (g.N >> 5) - 1;
      srcB[j] = g.B + ((int64_t)pn * g.ldb + k_begin + row) * 32 + lc * 8;
    }
  } else {
#pragma unroll
  for (int j = 0; j < PAt; ++j) {
    const int piece = wave * PAt + j, row = piece * 2 + (lane >> 5), pc = lane & 31;
    int col = m0 + ((pc ^ tn_f(row)) << 3);
    if (col > g.M - 8) col = g.M - 8;                       // M % 8 == 0: clamp whole chunks (results beyond M are not stored)
    srcA[j] = g.A + (k_begin + row) * g.lda + col;
  }
#pragma unroll
  for (int j = 0; j < PBt; ++j) {
    const int piece = wave * PBt + j, row = piece * RPPB + lane / CPRB, pc = lane % CPRB;
    int col = n0 + ((pc ^ tn_f(row)) << 3);
    if (col > g.N - 8) col = g.N - 8;
    srcB[j] = g.B + (k_begin + row) * g.ldb + col;
  }
  }
  const int64_t kadvA = PANEL ? 32 * 32 : 32 * g.lda, kadvB = PANEL ? 32 * 32 : 32 * g.ldb;   // elements per K step (32 tokens)
  auto issue = [&](int kt) {
    char* base = smem + (kt % NSTt) * STAGEt;
#pragma unroll
    for (int j = 0; j < PAt; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[j] + (int64_t)kt * kadvA),
                                       (__attribute__((address_space(3))) void*)(base + (wave * PAt + j) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < PBt; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[j] + (int64_t)kt * kadvB),
                                       (__attribute__((address_space(3))) void*)(base + ASTAGE + (wave * PBt + j) * 1024), 16, 0, 0);
  };
  // transposing reads: lane 4q + p of a 16-lane group addresses row q, columns 4p .. 4p+3 of a (4 k) x (16 columns) block and
  // receives column (lane & 15) of the 4 rows.  Group = k-group fg: rows 8 fg + 4 half + q.
  const int q = (lane & 15) >> 2, p = lane & 3;
  int offA[2], offB[2];                                    // byte offsets inside a stage for half = 0 / 1, column tile 0
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int row = 8 * fg + 4 * half + q;
    const int f = tn_f(row);
    if constexpr (PANEL) {   // (row start + the lane's 8 bytes inside a 32-byte half; frag_half adds the panel and the half)
      offA[half] = row * 64 + ((p >> 1) << 4) + ((p & 1) << 3);
      offB[half] = ASTAGE + offA[half];
      continue;
    }
    // column 4p of a 16-column tile starting at a multiple of 16: chunk (tile*2 + (p >> 1)) ^ f, byte (p & 1) * 8
    offA[half] = row * (BMt * 2) + ((((p >> 1)) ^ f) << 4) + ((p & 1) << 3);
    offB[half] = ASTAGE + row * (BNt * 2) + ((((p >> 1)) ^ f) << 4) + ((p & 1) << 3);
  }
  // The transposing reads are issued as inline asm: behind the intrinsic form hipcc puts `s_waitcnt vmcnt(0)` in front of the first
  // read of every K-step (it cannot tell that the read does not alias the LDS-DMA stage it has just queued), which turns the
  // three-stage ring into a synchronous copy.  The asm form hides the reads from that analysis; their results are only touched
  // after the explicit lgkmcnt(0) below, which names them as operands so that nothing that uses them can move above it.
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto frag_half = [&](unsigned stage_off, const int (&off)[2], int tile16, int half) -> s16x4 {
    // tile16 = index of the 16-column tile: its two chunks are 2*tile16, 2*tile16 + 1, XOR-ed with the row's swizzle f
    const int row = 8 * fg + 4 * half + q;
    unsigned addr;
    if constexpr (PANEL) {   // panel tile16 / 2 of the operand's stage, 64-byte rows, 32-byte half (tile16 & 1) ^ (row bit 3)
      addr = lds0 + stage_off + off[half] + (tile16 >> 1) * 2048 + (((tile16 & 1) ^ ((row >> 3) & 1)) << 5);
    } else {
    const int f = tn_f(row);
    const int base = off[half] - (((p >> 1) ^ f) << 4);          // row start (+ byte-in-chunk)
    addr = lds0 + stage_off + base + ((((tile16 << 1) + (p >> 1)) ^ f) << 4);
    }
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
  };
  auto join = [](const s16x4& lo, const s16x4& hi) -> bf16x8 {
    bf16x8 r;
    __builtin_memcpy(&r, &lo, 8);
    __builtin_memcpy(reinterpret_cast<char*>(&r) + 8, &hi, 8);
    return r;
  };

  f32x4 acc[TIt][TJt];
#pragma unroll
  for (int i = 0; i < TIt; ++i)
#pragma unroll
    for (int j = 0; j < TJt; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int NCS = TIt / (CS > 0 ? CS : TIt);            // column-sum tiles per worker
  f32x4 accs[NCS];
#pragma unroll
  for (int c = 0; c < NCS; ++c) accs[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int worker = (tile_i % tiles_n) * WN + wn;  // (wave-uniform)
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

  const int npro = nk < NSTt - 1 ? nk : NSTt - 1;
  for (int st = 0; st < npro; ++st) issue(st);
  for (int kt = 0; kt < nk; ++kt) {
    const int younger = nk - 1 - kt < NSTt - 2 ? nk - 1 - kt : NSTt - 2;
    wait_stages<PAt + PBt>(younger);
    __builtin_amdgcn_s_barrier();
    if (kt + NSTt - 1 < nk) issue(kt + NSTt - 1);
    const unsigned stage = (unsigned)((kt % NSTt) * STAGEt);
    s16x4 ah[TIt][2], bh[TJt][2];
#pragma unroll
    for (int j = 0; j < TJt; ++j) { bh[j][0] = frag_half(stage, offB, wn * 4 + j, 0); bh[j][1] = frag_half(stage, offB, wn * 4 + j, 1); }
#pragma unroll
    for (int i = 0; i < TIt; ++i) { ah[i][0] = frag_half(stage, offA, wm * 8 + i, 0); ah[i][1] = frag_half(stage, offA, wm * 8 + i, 1); }
    static_assert(TIt == 8 && TJt == 4, "the wait below names the 24 fragment halves");
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(ah[0][0]), "+v"(ah[0][1]), "+v"(ah[1][0]), "+v"(ah[1][1]), "+v"(ah[2][0]), "+v"(ah[2][1]), "+v"(ah[3][0]), "+v"(ah[3][1]),
                   "+v"(ah[4][0]), "+v"(ah[4][1]), "+v"(ah[5][0]), "+v"(ah[5][1]), "+v"(ah[6][0]), "+v"(ah[6][1]), "+v"(ah[7][0]), "+v"(ah[7][1]),
                   "+v"(bh[0][0]), "+v"(bh[0][1]), "+v"(bh[1][0]), "+v"(bh[1][1]), "+v"(bh[2][0]), "+v"(bh[2][1]), "+v"(bh[3][0]), "+v"(bh[3][1]));
    bf16x8 a[TIt], b[TJt];
#pragma unroll
    for (int j = 0; j < TJt; ++j) b[j] = join(bh[j][0], bh[j][1]);
#pragma unroll
    for (int i = 0; i < TIt; ++i) a[i] = join(ah[i][0], ah[i][1]);
#pragma unroll
    for (int i = 0; i < TIt; ++i)
#pragma unroll
      for (int j = 0; j < TJt; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);   // D'[n][m]
    if constexpr (CS > 0) {
      if (worker < CS) {
#pragma unroll
        for (int c = 0; c < NCS; ++c) {
          bf16x8 ac = a[c * CS];
#pragma unroll
          for (int w = 1; w < CS; ++w) if (worker == w) ac = a[c * CS + w];
          accs[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, ac, accs[c], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }
  // D' tile (rows n, cols m): lane holds m = fr, n = 4 fg + r -> 4 consecutive n of one m: one 16-byte store
  float* outp = g.out + (int64_t)slice_i * g.slice;
  if constexpr (CS > 0) {
    if (worker < CS && fg == 0) {                           // every row of the ones-product holds the sums: take row 0 (lanes 0..15)
#pragma unroll
      for (int c = 0; c < NCS; ++c) {
        const int m = m0 + wm * 128 + 16 * (c * CS + worker) + fr;
        if (m < g.M) outp[(int64_t)g.M * g.N + m] = accs[c][0];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < TIt; ++i) {
    const int m = m0 + wm * 128 + 16 * i + fr;
    if (m < g.M) {
#pragma unroll
      for (int j = 0; j < TJt; ++j) {
        const int n = n0 + wn * 64 + 16 * j + 4 * fg;
        if (n < g.N) *reinterpret_cast<f32x4*>(outp + (int64_t)m * g.N + n) = acc[i][j];
      }
    }
  }
}

}  // namespace

namespace { MH_KNOB(int, g_dw_blocks, 512); MH_KNOB(int, g_dw_wide, 1); }
// blocks a weight-gradient launch aims for when it cuts the token range (A/B knob; 512 = two 4-wave blocks per CU; the 256 x 256
// tile's 8-wave blocks count double)
#ifdef MH_ABLATE
extern "C" int mh_gemm_dw_set_blocks(int blocks) {
  g_dw_blocks = blocks < 1 ? 1 : blocks;
  return MH_OK;
}
#endif
// A/B: 0 = always the 256 x 128 tile (round 2), 1 = the 256 x 256 tile where N is a multiple of 256
#ifdef MH_ABLATE
extern "C" int mh_gemm_dw_set_wide(int on) {
  g_dw_wide = on ? 1 : 0;
  return MH_OK;
}
#endif
namespace { bool dw_wide(int N) { return g_dw_wide && N % 256 == 0; } }

extern "C" int mh_gemm_dw_splits(int64_t K, int M, int N) {
  const bool wide = dw_wide(N);
  const int tiles = ceil_div(M, 256) * ceil_div(N, wide ? 256 : 128);
  const int target = wide ? (g_dw_blocks + 1) / 2 : g_dw_blocks;
  // as many slices as keep every block slot of the chip busy ONCE (round 6: any count - the slices may differ by one K step; rounds 2 - 5
  // doubled the count until it reached the target, which ran the [1536 x 512] gradient's 12 tiles as 384 blocks = 1.5 rounds on 256 CUs),
  // at most 64 and at least 16 K steps (512 tokens) per slice
  int64_t S = target / tiles;
  const int64_t ksteps = K / 32;
  if (S > ksteps / 16) S = ksteps / 16;
  if (S > 64) S = 64;
  if (S < 1) S = 1;
  return (int)S;
}

// dW = A^T B for k-major bf16 operands: out_partials [splits][M][N] fp32 (splits = mh_gemm_dw_splits(K, M, N); fold with
// mh_sum_slices).  M, N multiples of 8, lda / ldb multiples of 8, K a multiple of 32 * splits.
extern "C" int mh_gemm_dw_bias(const void* A, int64_t lda, const void* B, int64_t ldb, float* out_partials, int splits, int64_t K, int M,
                               int N, int with_colsum, mh_stream_t stream);
extern "C" int mh_gemm_dw_bias_ex(const void* A, int64_t lda, const void* B, int64_t ldb, int panel, float* out_partials, int splits, int64_t K,
                                  int M, int N, int with_colsum, mh_stream_t stream);

extern "C" int mh_gemm_dw(const void* A, int64_t lda, const void* B, int64_t ldb, float* out_partials, int splits, int64_t K, int M,
                          int N, mh_stream_t stream) {
  return mh_gemm_dw_bias(A, lda, B, ldb, out_partials, splits, K, M, N, 0, stream);
}

// with_colsum != 0: every split slice is M N + M floats - the products, then the column sums of A over the slice's tokens (A = dY:
// the bias gradient of the linear whose weight gradient this is); one mh_sum_slices over M N + M elements folds both.
extern "C" int mh_gemm_dw_bias(const void* A, int64_t lda, const void* B, int64_t ldb, float* out_partials, int splits, int64_t K, int M,
                               int N, int with_colsum, mh_stream_t stream) {
  return mh_gemm_dw_bias_ex(A, lda, B, ldb, 0, out_partials, splits, K, M, N, with_colsum, stream);
}

// panel != 0: both operands as K32 panels [cols / 32][ld rows][32] (lda / ldb = rows of the panel buffers, M and N multiples of 32)
extern "C" int mh_gemm_dw_bias_ex(const void* A, int64_t lda, const void* B, int64_t ldb, int panel, float* out_partials, int splits, int64_t K,
                                  int M, int N, int with_colsum, mh_stream_t stream) {
  MH_CHECK_ARG(A && B && out_partials, "gemm_dw: null pointer");
  MH_CHECK_ARG(!panel || (M % 32 == 0 && N % 32 == 0 && lda >= K && ldb >= K), "gemm_dw: panel operands need M, N multiples of 32 and ld >= K rows");
  MH_CHECK_ARG(M > 0 && N > 0 && M % 8 == 0 && N % 4 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "gemm_dw: M, N, lda, ldb must be multiples of 8");
  MH_CHECK_ARG(splits >= 1 && splits <= 65535 && K > 0 && K % 32 == 0 && splits <= K / 32, "gemm_dw: K=%lld must be a multiple of 32 with at least one K step per split", (long long)K);
  const bool wide = dw_wide(N);
  const int tiles_n = ceil_div(N, wide ? 256 : 128);
  const int tiles = ceil_div(M, 256) * tiles_n;
  MH_CHECK_ARG((int64_t)tiles * splits < (1ll << 31), "gemm_dw: grid too large");
  TnArgs g{(const bf16*)A, lda, (const bf16*)B, ldb, out_partials, M, N, K / 32, (int64_t)M * N + (with_colsum ? M : 0), splits, tiles};
  const dim3 grid((unsigned)(tiles * splits));
  mh_prof_note("gemm_dw M=%d N=%d K=%lld splits=%d colsum=%d tile=256x%d", M, N, (long long)K, splits, with_colsum != 0, wide ? 256 : 128);
  hipStream_t st = (hipStream_t)stream;
  if (panel && wide) {
    if (!with_colsum) MH_LAUNCH((gemm_tn_kernel<0, 4, true>), grid, dim3(512), 0, st, g);
    else if (tiles_n >= 2) MH_LAUNCH((gemm_tn_kernel<8, 4, true>), grid, dim3(512), 0, st, g);
    else MH_LAUNCH((gemm_tn_kernel<4, 4, true>), grid, dim3(512), 0, st, g);
  }
  else if (panel) {
    if (!with_colsum) MH_LAUNCH((gemm_tn_kernel<0, 2, true>), grid, dim3(256), 0, st, g);
    else if (tiles_n >= 4) MH_LAUNCH((gemm_tn_kernel<8, 2, true>), grid, dim3(256), 0, st, g);
    else if (tiles_n >= 2) MH_LAUNCH((gemm_tn_kernel<4, 2, true>), grid, dim3(256), 0, st, g);
    else MH_LAUNCH((gemm_tn_kernel<2, 2, true>), grid, dim3(256), 0, st, g);
  }
  else if (wide) {   // column-sum workers = n-tiles x 4 wave columns
    if (!with_colsum) MH_LAUNCH((gemm_tn_kernel<0, 4>), grid, dim3(512), 0, st, g);
    else if (tiles_n >= 2) MH_LAUNCH((gemm_tn_kernel<8, 4>), grid, dim3(512), 0, st, g);
    else MH_LAUNCH((gemm_tn_kernel<4, 4>), grid, dim3(512), 0, st, g);
  }
  else if (!with_colsum) MH_LAUNCH(gemm_tn_kernel<0>, grid, dim3(256), 0, st, g);
  else if (tiles_n >= 4) MH_LAUNCH(gemm_tn_kernel<8>, grid, dim3(256), 0, st, g);
  else if (tiles_n >= 2) MH_LAUNCH(gemm_tn_kernel<4>, grid, dim3(256), 0, st, g);
  else MH_LAUNCH(gemm_tn_kernel<2>, grid, dim3(256), 0, st, g);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
