// Head and tail of the denoiser as ONE kernel each (bf16 K32-panel path, d_model 256 / 512, E_pad <= 128):
//
//   head (network.py:141-149):  x [rows, E] fp32 -> tanh(x W0^T + b0) -> . W2^T + b2 -> (+ pos[l]) + emb_t[b] -> LayerNorm -> bf16 panel rows
//   tail (network.py:153-157):  X bf16 panel rows -> tanh(X W0^T + b0) -> . W2^T + b2 -> y [rows, E] fp32
//
// Before: pack_panel + 2 GEMMs + add-pos-time-LayerNorm (4 launches, the [rows, 512] intermediate written and re-read twice) and
// 2 GEMMs on tiles whose K or N is 128 (150 - 400 TFLOP/s).  Here a block owns 64 complete rows: the [64, H] intermediate of the
// first dense layer stays in LDS as the second one's A operand, only the weights stream (L2 -> LDS by LDS-DMA, 3-stage ring of
// whole K32 steps: H rows x 64 B), and the LayerNorm runs on the fp32 accumulators.
//
// Layouts are gemm.hip's: LDS rows of 64 B (one K32 step), 16-byte chunk c of row r at c ^ G[(r >> 2) & 3], G = {0, 2, 3, 1}
// (conflict-free ds_read_b128 fragments); DMA pieces of 16 rows x 64 B with the swizzle on the SOURCE address; MFMA 16x16x32 bf16
// issued as D = W_tile . A_tile^T with the W rows of a wave's 64 columns dealt to the MFMA input rows as
// 32 (jj >> 1) + 8 (p >> 2) + 4 (jj & 1) + (p & 3), so that a lane owns 8 CONSECUTIVE output columns of one row.
#include "common.h"
#include "step_update.h"

namespace {

// token rows per block (template parameter ROWS): 64 at d_model 256 / 512; 32 at d_model 768 (bert-base, the reference-true width:
// network.py:44-46), where 64 rows of a 768-wide intermediate plus three weight stages would need 240 KB of LDS - there the ring has
// TWO slots, refilled between a K step's fragment reads and its MFMAs (the same two stages in flight).  One K32 step of the A operand
// = ROWS x 64 B (a "slab").
// G = {0, 2, 3, 1} as a packed table (no memory lookup)
__device__ __forceinline__ int ht_g(int x) { return (0x78 >> (2 * x)) & 3; }

struct HeadArgs {
  const float* x; int64_t rows; int E, E_pad, L;
  const bf16* w0; const float* b0;      // [E_pad / 32][H][32]
  const bf16* w2; const float* b2;      // [H / 32][H][32]
  const float* pos; const float* emb_t; const int32_t* emb_row;
  const float* gamma; const float* beta; float eps;
  bf16* out; int64_t ldo;               // [H / 32][ldo][32]
};

struct TailArgs {
  const bf16* X; int64_t ldx; int64_t rows; int E;
  const bf16* w0; const float* b0;      // [H / 32][H][32]
  const bf16* w2; const float* b2;      // [H / 32][E][32]  (E rows: the down-projection's outputs)
  float* out;                           // [rows, E] fp32
  float* sqnorm;                        // optional [rows]: |out row|^2 (the rounding scores' |x_n|^2, models/rounding.py:23)
  // ROUND (TJ3 > 0): nearest-embedding rounding of the block's rows inside the same kernel (models/rounding.py:21-28)
  const bf16* tsplit;                   // [3 E / 32][Vp][32]: the table as bf16 hi | hi | lo parts (mh_round_split_table)
  const float* tnorm;                   // [Vp]: |T_v|^2 fp32, +inf for rows >= V
  int V;
  int32_t* idx;                         // [rows]: nearest row per token
  // UPDATE (with ROUND): the posterior / DDIM step of the block's rows in the same kernel (models/diffusion.py:319-347, :390-397, :729-757)
  float* x;                             // [rows, E] x_t in, x_{t-1} out (in place); NULL: no update here
  const float* x_start; const int32_t* mask; int mask_per_elem;
  const float* table;                   // [V, E] fp32 embedding rows (pred_xstart = table[idx])
  const mh_step_coef* coef; int clip, ddim;
  float* pred; float* mean;             // optional outputs
  const float* noise;                   // read when has_rng == 0
  StepRng rng; int has_rng;
};

template <int N> __device__ __forceinline__ void ht_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void ht_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// one W stage (K32 step kt of a [K / 32][WROWS][32] panel matrix) into an LDS slot: WROWS / 16 pieces dealt over NW waves
template <int WROWS, int NW>
__device__ __forceinline__ void ht_issue_w(const bf16* __restrict__ w, int kt, char* slot, int wave, int lane) {
  constexpr int PIECES = WROWS / 16, PER = (PIECES + NW - 1) / NW;
  const int rl = lane >> 2, pc = lane & 3, lc = pc ^ ht_g((rl >> 2) & 3);
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int p = (wave * PER + j) % PIECES;    // (fewer pieces than waves: the upper waves repeat a piece - same bytes, same address - so every wave's count is the same)
    const bf16* src = w + ((int64_t)kt * WROWS + p * 16 + rl) * 32 + lc * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(slot + p * 1024), 16, 0, 0);
  }
}

// 4 RT MFMAs of a wave's (16 RT) x 64 sub-tile for one K32 step: A rows from a slab, W rows from a ring slot
template <int RT>
__device__ __forceinline__ void ht_kstep(f32x4 (&acc)[RT][4], const char* a_slab, const char* w_slot, int a_off, const int (&b_offs)[4]) {
  bf16x8 a[RT], b[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(w_slot + b_offs[j]);
#pragma unroll
  for (int i = 0; i < RT; ++i) a[i] = *reinterpret_cast<const bf16x8*>(a_slab + a_off + i * 1024);
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
}

// K loop over nk K32 steps: A slabs resident in LDS (a_base + kt * slab bytes), W streamed through an NSLOT-slot ring.  `issued`: stages
// the caller already put in flight (0 .. 2, issued as the wave's LAST vector-memory operations).  PER = pieces per wave and stage.
// NO block barrier inside: the pieces a wave DMAs (rows 64 wave .. 64 wave + 63 of the stage) are exactly the W rows its own 64 output
// columns read, so a stage is private to the wave that loaded it - its own counted vmcnt says when the bytes have landed, its own
// lgkmcnt when the slot may be refilled - and the waves of a block drift apart instead of meeting 16 MFMAs apart.  The caller
// publishes the (shared) A slabs with one barrier before the loop.
// NSLOT 3: stage kt + 2 goes into the slot stage kt - 1 was read from, before stage kt's reads.  NSLOT 2: stage kt's fragments go to
// registers first, then its own slot takes stage kt + 2 - two stages in flight either way.
template <int WROWS, int NW, int RT, int NSLOT>
__device__ __forceinline__ void ht_loop(f32x4 (&acc)[RT][4], const bf16* __restrict__ w, int nk, const char* a_base, char* ring, int slot_bytes,
                                        int a_off, const int (&b_offs)[4], int wave, int lane, int issued) {
  constexpr int PER = (WROWS / 16 + NW - 1) / NW, SLAB = RT * 16 * 64;
  static_assert(WROWS / 16 == PER * NW, "a wave must load exactly its own rows");
  static_assert(NSLOT == 2 || NSLOT == 3, "ring of two or three slots");
  for (int st = issued; st < 2 && st < nk; ++st) ht_issue_w<WROWS, NW>(w, st, ring + (st % NSLOT) * slot_bytes, wave, lane);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) ht_wait_vmcnt<PER>(); else ht_wait_vmcnt<0>();    // this wave's stage kt has landed (stage kt + 1 may still fly)
    if constexpr (NSLOT == 3) {
      ht_lgkm0();                                                        // ... and its reads of stage kt - 1 are done: that slot is free
      if (kt + 2 < nk) ht_issue_w<WROWS, NW>(w, kt + 2, ring + ((kt + 2) % 3) * slot_bytes, wave, lane);
      ht_kstep<RT>(acc, a_base + kt * SLAB, ring + (kt % 3) * slot_bytes, a_off, b_offs);
    } else {
      const char* a_slab = a_base + kt * SLAB;
      const char* w_slot = ring + (kt & 1) * slot_bytes;
      bf16x8 a[RT], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(w_slot + b_offs[j]);
#pragma unroll
      for (int i = 0; i < RT; ++i) a[i] = *reinterpret_cast<const bf16x8*>(a_slab + a_off + i * 1024);
      ht_lgkm0();                                                        // stage kt is in registers: its slot takes stage kt + 2
      if (kt + 2 < nk) ht_issue_w<WROWS, NW>(w, kt + 2, ring + (kt & 1) * slot_bytes, wave, lane);
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
  }
}

template <int H, int ROWS = 64>
__global__ __launch_bounds__(H / 64 * 64, H / 256) void head_fused_kernel(const HeadArgs g) {
  constexpr int NW = H / 64, THREADS = NW * 64, NK2 = H / 32, SLOT = H * 64, RT = ROWS / 16, SLAB = ROWS * 64;
  constexpr int NSLOT = NK2 * SLAB + 3 * SLOT <= 160 * 1024 ? 3 : 2;
  constexpr int H1_BYTES = NK2 * SLAB, LDS_BYTES = H1_BYTES + NSLOT * SLOT;
  static_assert(LDS_BYTES <= 160 * 1024, "head: LDS");
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  char* const h1 = smem;                 // phase 1: the x tile (E_pad / 32 slabs); phase 2: tanh(up0) as A operand (H / 32 slabs)
  char* const ring = smem + H1_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int nk1 = g.E_pad / 32;
  // the first two weight stages fly while the x tile is converted
  for (int st = 0; st < 2 && st < nk1; ++st) ht_issue_w<H, NW>(g.w0, st, ring + (st % NSLOT) * SLOT, wave, lane);
  // x tile: fp32 [64, E] -> bf16 slabs; a thread owns 16-byte chunks (row, 8 columns)
  for (int c = tid; c < ROWS * (g.E_pad / 8); c += THREADS) {
    const int row = c / (g.E_pad / 8), ch = c % (g.E_pad / 8), col = ch * 8;
    int64_t r = row0 + row; if (r >= g.rows) r = g.rows - 1;
    float v[8];
    if (col + 8 <= g.E) load8(g.x + r * g.E + col, v);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = col + e < g.E ? g.x[r * g.E + col + e] : 0.f;
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
    *reinterpret_cast<bf16x8*>(h1 + (ch >> 2) * SLAB + row * 64 + (((ch & 3) ^ ht_g((row >> 2) & 3)) << 4)) = o;
  }
  const int a_off = fr * 64 + ((fg ^ ht_g((fr >> 2) & 3)) << 4);
  int b_offs[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int row = wave * 64 + 32 * (jj >> 1) + 8 * (fr >> 2) + 4 * (jj & 1) + (fr & 3);
    b_offs[jj] = row * 64 + ((fg ^ ht_g((row >> 2) & 3)) << 4);
  }
  f32x4 acc[RT][4];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // ---- phase 1: [64, E_pad] . W0^T
  ht_lgkm0();
  __builtin_amdgcn_s_barrier();           // the x tile is complete (the K loop itself has no barrier: its W stages are wave-private)
  ht_loop<H, NW, RT, NSLOT>(acc, g.w0, nk1, h1, ring, SLOT, a_off, b_offs, wave, lane, nk1 < 2 ? nk1 : 2);
  ht_lgkm0();
  __builtin_amdgcn_s_barrier();           // every wave is done with the x tile and the ring
  for (int st = 0; st < 2; ++st) ht_issue_w<H, NW>(g.w2, st, ring + (st % NSLOT) * SLOT, wave, lane);   // phase 2's first stages fly under the epilogue
  // epilogue 1: tanh(acc + b0) -> bf16 -> the A slabs of phase 2.  Lane: row 16 i + fr, columns 64 wave + 32 h + 8 fg + e
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float bv[8];
    load8(g.b0 + wave * 64 + 32 * h + 8 * fg, bv);
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int row = 16 * i + fr;
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16)tanh_fast(acc[i][2 * h + (e >> 2)][e & 3] + bv[e]);
      *reinterpret_cast<bf16x8*>(h1 + (2 * wave + h) * SLAB + row * 64 + ((fg ^ ht_g((row >> 2) & 3)) << 4)) = o;
    }
  }
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the position rows of the second epilogue (64 distinct table rows per block, 128 KB from L2) are requested here and arrive under
  // phase 2 instead of standing between its last MFMA and the LayerNorm: 64 registers held across the loop
  float posv[2][RT][8];
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    int64_t r = row0 + 16 * i + fr; if (r >= g.rows) r = g.rows - 1;
    const float* prow = g.pos + (r % g.L) * H;
#pragma unroll
    for (int h = 0; h < 2; ++h) load8(prow + wave * 64 + 32 * h + 8 * fg, posv[h][i]);
  }
  // ---- phase 2: [64, H] . W2^T
  ht_lgkm0();
  __builtin_amdgcn_s_barrier();           // the tanh slabs are complete
  ht_loop<H, NW, RT, NSLOT>(acc, g.w2, NK2, h1, ring, SLOT, a_off, b_offs, wave, lane, 2);
  ht_lgkm0();
  __builtin_amdgcn_s_barrier();           // the ring is idle: its first bytes take the row statistics
  // ---- epilogue 2: (pos + (acc + b2)) + emb_t, LayerNorm over the row (two passes: in-lane -> 4 lanes -> NW waves through LDS)
  float* red = reinterpret_cast<float*>(ring);          // [64][NW]
  const float* trow[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    int64_t r = row0 + 16 * i + fr; if (r >= g.rows) r = g.rows - 1;
    const int64_t b = r / g.L;
    trow[i] = g.emb_t + (int64_t)(g.emb_row ? g.emb_row[b] : (int)b) * H;
  }
  float rs[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) rs[i] = 0.f;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int col = wave * 64 + 32 * h + 8 * fg;
    float bv[8];
    load8(g.b2 + col, bv);
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      float t[8];
      load8(trow[i] + col, t);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = (posv[h][i][e] + (acc[i][2 * h + (e >> 2)][e & 3] + bv[e])) + t[e];     // the reference's association: (pos + x) + emb
        acc[i][2 * h + (e >> 2)][e & 3] = v;
        rs[i] += v;
      }
    }
  }
  const float invH = 1.0f / (float)H;
  float mean[RT], rstd[RT];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      float v = rs[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (fg == 0) red[(16 * i + fr) * NW + wave] = v;
    }
    ht_lgkm0();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red[(16 * i + fr) * NW + w];
      if (pass == 0) {
        mean[i] = t * invH;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float d = acc[i][j][r] - mean[i]; sq += d * d; }
        rs[i] = sq;
      } else {
        rstd[i] = 1.0f / sqrtf(t * invH + g.eps);
      }
    }
    ht_lgkm0();
    __builtin_amdgcn_s_barrier();
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int col = wave * 64 + 32 * h + 8 * fg;
    float gv[8], bt[8];
    load8(g.gamma + col, gv);
    load8(g.beta + col, bt);
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int64_t r = row0 + 16 * i + fr;
      if (r < g.rows) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (acc[i][2 * h + (e >> 2)][e & 3] - mean[i]) * rstd[i] * gv[e] + bt[e];
        store8(g.out + ((int64_t)(col >> 5) * g.ldo + r) * 32 + (col & 31), v);
      }
    }
  }
}

// tail: the block's 64 x H input rows are DMA'd whole into the slab area (they are the A operand of the first dense layer), its
// tanh output replaces them there, and the E outputs of the second layer are spread over the waves 16 columns each
// (E = 128 on 8 waves, E = 64 on 4).
// TJ3 > 0: phase 3 - the rounding scores of the block's 64 rows against the whole embedding table on the bf16 matrix pipe at fp32
// grade: x = x_hi + x_lo, T = T_hi + T_lo (bf16 parts), x . T ~ x_hi T_hi + x_lo T_hi + x_hi T_lo (three products per element, the
// dropped x_lo T_lo term is 2^-16 of the product), as ONE K = 3 E contraction [x_hi | x_lo | x_hi] . [T_hi | T_hi | T_lo]^T with fp32
// accumulation.  A wave owns 16 TJ3 table rows; -(clamp((|T_v|^2 + |x_n|^2) - 2 x.T_v, 0)) and the first-index argmax exactly as the
// score GEMM's epilogue + argbest_reduce do them, folded lane -> 4 lanes -> waves through LDS.  Replaces the exact-fp32 score GEMM
// launch (30 us per half batch) in the bf16 throughput mode; the fp32 parity mode keeps the exact-fp32 MFMA path.
template <int H, int TJ3 = 0, int ROWS = 64>
__global__ __launch_bounds__(H / 64 * 64, H / 256) void tail_fused_kernel(const TailArgs g) {
  constexpr int NW = H / 64, NK = H / 32, SLOT = H * 64, RT = ROWS / 16, SLAB = ROWS * 64;
  constexpr int NSLOT = NK * SLAB + 3 * SLOT <= 160 * 1024 ? 3 : 2;
  constexpr int H1_BYTES = NK * SLAB, LDS_BYTES = H1_BYTES + NSLOT * SLOT;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  char* const h1 = smem;
  char* const ring = smem + H1_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  // A rows: 4 pieces (16 rows x 64 B) per K32 step, NK steps, dealt over the waves; then the first two weight stages
  {
    const int rl = lane >> 2, pc = lane & 3, lc = pc ^ ht_g((rl >> 2) & 3);
    constexpr int PIECES = NK * RT, PER = PIECES / NW;
    static_assert(PIECES == PER * NW, "tail: the A pieces must deal evenly over the waves");
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int p = wave * PER + j, kt = p / RT, rb = (p % RT) * 16;
      int64_t r = row0 + rb + rl; if (r >= g.rows) r = g.rows - 1;
      const bf16* src = g.X + ((int64_t)kt * g.ldx + r) * 32 + lc * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(h1 + kt * SLAB + rb * 64), 16, 0, 2);   // nt: read once
    }
  }
  for (int st = 0; st < 2; ++st) ht_issue_w<H, NW>(g.w0, st, ring + (st % NSLOT) * SLOT, wave, lane);
  const int a_off = fr * 64 + ((fg ^ ht_g((fr >> 2) & 3)) << 4);
  int b_offs[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int row = wave * 64 + 32 * (jj >> 1) + 8 * (fr >> 2) + 4 * (jj & 1) + (fr & 3);
    b_offs[jj] = row * 64 + ((fg ^ ht_g((row >> 2) & 3)) << 4);
  }
  f32x4 acc[RT][4];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  ht_wait_vmcnt<2 * (H / 16 / NW)>();     // this wave's pieces of the A rows have landed (the two weight stages behind them may still fly)
  __builtin_amdgcn_s_barrier();           // ... and everybody else's: the A slabs are shared, the weight stages are wave-private
  ht_loop<H, NW, RT, NSLOT>(acc, g.w0, NK, h1, ring, SLOT, a_off, b_offs, wave, lane, 2);
  ht_lgkm0();
  __builtin_amdgcn_s_barrier();
  // second layer's weight stages: E rows x 64 B each
  const int E16 = g.E / 16;                       // 16-column groups of the output (8 for E = 128)
  // (d_model 768: 12 waves for E / 16 = 8 column groups - the upper waves sit this layer out; a wave that repeated a lower wave's piece
  // would refill a slot that wave may still be reading)
  const bool w2_active = wave < E16;
  auto issue2 = [&](int kt, int slot) {
    const int rl = lane >> 2, pc = lane & 3, lc = pc ^ ht_g((rl >> 2) & 3);
    const int p = wave % E16;
    const bf16* src = g.w2 + ((int64_t)kt * g.E + p * 16 + rl) * 32 + lc * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(ring + slot * SLOT + p * 1024), 16, 0, 0);
  };
  if (w2_active) {
    issue2(0, 0);
    issue2(1, 1);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float bv[8];
    load8(g.b0 + wave * 64 + 32 * h + 8 * fg, bv);
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int row = 16 * i + fr;
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16)tanh_fast(acc[i][2 * h + (e >> 2)][e & 3] + bv[e]);
      *reinterpret_cast<bf16x8*>(h1 + (2 * wave + h) * SLAB + row * 64 + ((fg ^ ht_g((row >> 2) & 3)) << 4)) = o;
    }
  }
  // ---- second layer: wave -> output columns 16 p .. 16 p + 15, p = wave % (E / 16); D[m = 4 fg + r][n = fr]: 4 consecutive columns of row fr
  const int p2 = wave % E16;
  const int b2_off = (p2 * 16 + fr) * 64 + ((fg ^ ht_g((fr >> 2) & 3)) << 4);
  f32x4 y[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) y[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  ht_lgkm0();
  __builtin_amdgcn_s_barrier();                   // the tanh slabs are complete (the loop's weight pieces are wave-private: no barrier inside)
  if (w2_active)
  for (int kt = 0; kt < NK; ++kt) {
    if (kt + 1 < NK) ht_wait_vmcnt<1>(); else ht_wait_vmcnt<0>();
    bf16x8 b, a[RT];
    if constexpr (NSLOT == 3) {
      ht_lgkm0();
      if (kt + 2 < NK) issue2(kt + 2, (kt + 2) % 3);
      b = *reinterpret_cast<const bf16x8*>(ring + (kt % 3) * SLOT + b2_off);
#pragma unroll
      for (int i = 0; i < RT; ++i) a[i] = *reinterpret_cast<const bf16x8*>(h1 + kt * SLAB + a_off + i * 1024);
    } else {   // two slots: the stage's fragments go to registers first, then its own slot takes stage kt + 2
      b = *reinterpret_cast<const bf16x8*>(ring + (kt & 1) * SLOT + b2_off);
#pragma unroll
      for (int i = 0; i < RT; ++i) a[i] = *reinterpret_cast<const bf16x8*>(h1 + kt * SLAB + a_off + i * 1024);
      ht_lgkm0();
      if (kt + 2 < NK) issue2(kt + 2, kt & 1);
    }
#pragma unroll
    for (int i = 0; i < RT; ++i) y[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a[i], y[i], 0, 0, 0);
  }
  float ss[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) ss[i] = 0.f;
  if (w2_active) {
    const int col = p2 * 16 + 4 * fg;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(g.b2 + col);
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int64_t r = row0 + 16 * i + fr;
      const f32x4 v = y[i] + bv;
      ss[i] = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
      if (r < g.rows) *reinterpret_cast<f32x4*>(g.out + r * g.E + col) = v;
    }
  }
  if constexpr (TJ3 > 0) {
    constexpr int VP = 16 * TJ3 * NW, SLOT3 = VP * 64, NK3 = 3 * 4;      // (E = 128: 4 slabs per part, 12 K32 steps)
    static_assert(2 * SLOT3 <= NSLOT * SLOT && NK3 * SLAB + (2 * ROWS * NW + ROWS) * 4 <= H1_BYTES, "rounding phase: LDS");
    ht_lgkm0();
    __builtin_amdgcn_s_barrier();           // every wave is past its last fragment read of phase 2: slabs and ring are free
    auto issue3 = [&](int kt) {
      const int rl = lane >> 2, pc = lane & 3, lc = pc ^ ht_g((rl >> 2) & 3);
#pragma unroll
      for (int j = 0; j < TJ3; ++j) {
        const int p = wave * TJ3 + j;
        const bf16* src = g.tsplit + ((int64_t)kt * VP + p * 16 + rl) * 32 + lc * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(ring + (kt & 1) * SLOT3 + p * 1024), 16, 0, 0);
      }
    };
    issue3(0);
    // x_hi | x_lo | x_hi slabs: this lane's 4 columns of row 16 i + fr -> 8 bytes in slab col / 32 (+ 4 for the low part, + 8 again the high part)
    float* red = reinterpret_cast<float*>(h1 + NK3 * SLAB);            // [64][NW] |row|^2 partials, then best scores
    int* redi = reinterpret_cast<int*>(red + ROWS * NW);               // [64][NW] best indices
    float* xn = reinterpret_cast<float*>(redi + ROWS * NW);            // [64] |x_n|^2
    {
      const int col = p2 * 16 + 4 * fg;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(g.b2 + col);
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const int row = 16 * i + fr;
        const f32x4 v = y[i] + bv;
        bf16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) { hi[e] = (bf16)v[e]; lo[e] = (bf16)(v[e] - (float)hi[e]); }
        char* dst = h1 + (col >> 5) * SLAB + row * 64 + (((((col & 31) >> 3)) ^ ht_g((row >> 2) & 3)) << 4) + (col & 4) * 2;
        if (w2_active) {          // (the waves that hold y; the others add a zero partial below)
          *reinterpret_cast<bf16x4*>(dst) = hi;
          *reinterpret_cast<bf16x4*>(dst + 4 * SLAB) = lo;
          *reinterpret_cast<bf16x4*>(dst + 8 * SLAB) = hi;
        }
        float v2 = ss[i];
        v2 += __shfl_xor(v2, 16, 64);
        v2 += __shfl_xor(v2, 32, 64);
        if (fg == 0) red[row * NW + wave] = v2;
      }
    }
    ht_lgkm0();
    __builtin_amdgcn_s_barrier();
    if (tid < ROWS) {
      float t = 0.f;
      for (int w = 0; w < NW; ++w) t += red[tid * NW + w];
      xn[tid] = t;
      if (g.sqnorm && row0 + tid < g.rows) g.sqnorm[row0 + tid] = t;
    }
    f32x4 sc[RT][TJ3];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < TJ3; ++j) sc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int b3_offs[TJ3];
#pragma unroll
    for (int j = 0; j < TJ3; ++j) b3_offs[j] = ((wave * TJ3 + j) * 16 + fr) * 64 + ((fg ^ ht_g((fr >> 2) & 3)) << 4);
    for (int kt = 0; kt < NK3; ++kt) {            // (a wave's table rows are its own pieces: no barrier inside)
      ht_wait_vmcnt<0>();                         // this wave's stage kt has landed
      ht_lgkm0();                                 // ... and its reads of stage kt - 1 are done: that slot takes stage kt + 1
      if (kt + 1 < NK3) issue3(kt + 1);
      bf16x8 a[RT], b[TJ3];
#pragma unroll
      for (int j = 0; j < TJ3; ++j) b[j] = *reinterpret_cast<const bf16x8*>(ring + (kt & 1) * SLOT3 + b3_offs[j]);
#pragma unroll
      for (int i = 0; i < RT; ++i) a[i] = *reinterpret_cast<const bf16x8*>(h1 + kt * SLAB + a_off + i * 1024);
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < TJ3; ++j) sc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], sc[i][j], 0, 0, 0);
    }
    // scores -> the best (score, first index) of every row: D[m = 4 fg + r][n = fr] = table row 16 (wave TJ3 + j) + 4 fg + r, token row 16 i + fr
    float tn[TJ3][4];
#pragma unroll
    for (int j = 0; j < TJ3; ++j) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(g.tnorm + (wave * TJ3 + j) * 16 + 4 * fg);
      tn[j][0] = t4[0]; tn[j][1] = t4[1]; tn[j][2] = t4[2]; tn[j][3] = t4[3];
    }
    ht_lgkm0();
    __builtin_amdgcn_s_barrier();               // (xn was written before the K loop's first barrier; `red` is reused below)
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const float x2 = xn[16 * i + fr];
      float best = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < TJ3; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = (wave * TJ3 + j) * 16 + 4 * fg + r;
          float dist = (tn[j][r] + x2) - 2.0f * sc[i][j][r];
          dist = fmaxf(dist, 0.0f);
          const float s1 = -dist;
          if (col < g.V && s1 > best) { best = s1; bi = col; }
        }
#pragma unroll
      for (int off = 16; off < 64; off <<= 1) {
        const float so = __shfl_xor(best, off, 64);
        const int io = __shfl_xor(bi, off, 64);
        if (so > best || (so == best && io < bi)) { best = so; bi = io; }
      }
      if (fg == 0) { red[(16 * i + fr) * NW + wave] = best; redi[(16 * i + fr) * NW + wave] = bi; }
    }
    ht_lgkm0();
    __builtin_amdgcn_s_barrier();
    if (tid < ROWS && row0 + tid < g.rows) {
      float best = -INFINITY;
      int bi = 0x7fffffff;
      for (int w = 0; w < NW; ++w) {              // waves hold increasing column ranges: strict > keeps the first index
        const float v = red[tid * NW + w];
        const int k = redi[tid * NW + w];
        if (v > best || (v == best && k < bi)) { best = v; bi = k; }
      }
      bi = bi == 0x7fffffff ? 0 : bi;
      g.idx[row0 + tid] = bi;
      redi[tid * NW] = bi;                        // (its own row's slot 0: only this thread read the row's slots)
    }
    if (!g.x) return;
    // ---- the update of the block's rows: the arithmetic of step_epilogue4_kernel (step_update.h), 4 consecutive elements per thread
    ht_lgkm0();
    __builtin_amdgcn_s_barrier();
    const mh_step_coef c = *g.coef;
    uint32_t rng_step = 0;
    if (g.has_rng) rng_step = g.rng.step ? *g.rng.step : 0u;
    const int gpr = g.E / 4;                      // groups per row
    for (int q = tid; q < ROWS * gpr; q += NW * 64) {
      const int row = q / gpr, cg = q - row * gpr;
      const int64_t r = row0 + row;
      if (r >= g.rows) continue;
      const int64_t i = r * g.E + cg * 4;
      f32x4 x0 = *reinterpret_cast<const f32x4*>(g.table + (int64_t)redi[row * NW] * g.E + cg * 4);
      const f32x4 xt = *reinterpret_cast<const f32x4*>(g.x + i);
      f32x4 nz = {0.f, 0.f, 0.f, 0.f};
      if (g.has_rng) {
        float z[4];
        trunc_normal4(g.rng.first_group + (uint64_t)(i >> 2), rng_step, g.rng.bound, g.rng.seed_lo, g.rng.seed_hi, g.rng.stream_id, z);
        nz = f32x4{z[0], z[1], z[2], z[3]};
      } else if (g.noise) nz = *reinterpret_cast<const f32x4*>(g.noise + i);
      f32x4 mean, sample;
      if (g.ddim) step_update4<true>(x0, xt, nz, c, g.clip, mean, sample);
      else step_update4<false>(x0, xt, nz, c, g.clip, mean, sample);
      if (g.mask) {
        if (g.mask_per_elem) {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (g.mask[i + e] == 0) sample[e] = g.x_start[i + e];
        } else if (g.mask[r] == 0) sample = *reinterpret_cast<const f32x4*>(g.x_start + i);
      }
      if (g.pred) *reinterpret_cast<f32x4*>(g.pred + i) = x0;
      if (g.mean && !g.ddim) *reinterpret_cast<f32x4*>(g.mean + i) = mean;
      *reinterpret_cast<f32x4*>(g.x + i) = sample;
    }
    return;
  }
  if (g.sqnorm) {   // |row|^2: 4 columns in the lane -> the 4 lanes of a row (xor 16, 32) -> the E / 16 waves through LDS, fixed order
    ht_lgkm0();
    __builtin_amdgcn_s_barrier();           // every wave is past its last fragment read: the ring's first bytes are free
    float* red = reinterpret_cast<float*>(ring);          // [64][NW]
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      float v = ss[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (fg == 0) red[(16 * i + fr) * NW + wave] = v;
    }
    ht_lgkm0();
    __builtin_amdgcn_s_barrier();
    if (tid < ROWS) {
      float t = 0.f;
      for (int w = 0; w < E16 && w < NW; ++w) t += red[tid * NW + w];
      if (row0 + tid < g.rows) g.sqnorm[row0 + tid] = t;
    }
  }
}

MH_KNOB(int, g_fuse_headtail, 1);

}  // namespace

#ifdef MH_ABLATE
extern "C" int mh_denoiser_set_fuse_headtail(int on) {
  g_fuse_headtail = on != 0;
  return MH_OK;
}
#endif

extern "C" int mh_up_proj_ln_fused_supported(int E, int E_pad, int H) {
  return g_fuse_headtail && (H == 256 || H == 512 || H == 768) && E_pad % 32 == 0 && E_pad <= 128 && E <= E_pad && E % 4 == 0;
}
// (E / 16 == H / 64: every wave owns exactly one 16-column group of the second layer, so its weight pieces are private to it and the K
// loop needs no barrier; E = 128 at d_model 512, E = 64 at d_model 256)
// (d_model 768 - bert-base, the width network.py:44-46 builds - has 12 waves: E = 128 occupies eight of them in the second layer)
extern "C" int mh_down_proj_fused_supported(int E, int H) {
  return g_fuse_headtail && E % 16 == 0 && (((H == 256 || H == 512) && E / 16 == H / 64) || (H == 768 && E == 128));
}

extern "C" int mh_up_proj_ln_fused(const float* x, int E, int E_pad, const void* w0, const float* b0, const void* w2, const float* b2,
                                   const float* pos, const float* emb_t, const int32_t* emb_row, const float* gamma, const float* beta,
                                   float eps, void* out, int64_t ldo, int B, int L, int H, mh_stream_t stream) {
  MH_CHECK_ARG(x && w0 && b0 && w2 && b2 && pos && emb_t && gamma && beta && out && B > 0 && L > 0, "up_proj_ln_fused: bad arguments");
  MH_CHECK_ARG(mh_up_proj_ln_fused_supported(E, E_pad, H), "up_proj_ln_fused: shape E=%d E_pad=%d H=%d not served", E, E_pad, H);
  HeadArgs g{x, (int64_t)B * L, E, E_pad, L, (const bf16*)w0, b0, (const bf16*)w2, b2, pos, emb_t, emb_row, gamma, beta, eps, (bf16*)out, ldo};
  const int rpb = H == 768 ? 32 : 64;      // token rows per block
  const dim3 grid((unsigned)((g.rows + rpb - 1) / rpb));
  mh_prof_note("head rows=%lld E=%d H=%d", (long long)g.rows, E, H);
  if (H == 768) MH_LAUNCH((head_fused_kernel<768, 32>), grid, dim3(768), 0, (hipStream_t)stream, g);
  else if (H == 512) MH_LAUNCH((head_fused_kernel<512>), grid, dim3(512), 0, (hipStream_t)stream, g);
  else MH_LAUNCH((head_fused_kernel<256>), grid, dim3(256), 0, (hipStream_t)stream, g);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// the rounding phase is built for the ComMU vocabulary on the config-2 width: E = 128 (8 waves x 16 output columns, 12 K32 steps of
// hi | lo | hi parts), d_model 512, 640 < V <= 768 (six 16-row table tiles per wave)
extern "C" int mh_down_proj_round_supported(int E, int H, int V) {
  return mh_down_proj_fused_supported(E, H) && E == 128 && (H == 512 || H == 768) && V > 640 && V <= 768;
}
extern "C" size_t mh_round_split_bytes(int E, int V) { return (size_t)(3 * E / 32) * 768 * 32 * 2 + 768 * 4; }
namespace {
__global__ void round_split_kernel(const float* __restrict__ table, int V, int E, bf16* __restrict__ tsplit, float* __restrict__ tnorm, int Vp) {
  // tsplit [3 E / 32][Vp][32]: parts hi | hi | lo of every table row (zero rows beyond V); tnorm [Vp] = |T_v|^2 (exact fp32, the sum
  // order of row_sqnorm_f32_kernel is not needed here: the value only has to be the same for every query) or +inf beyond V
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= Vp) return;
  float n2 = 0.f;
  for (int c = 0; c < E; ++c) {
    const float t = v < V ? table[(int64_t)v * E + c] : 0.f;
    const bf16 hi = (bf16)t;
    const bf16 lo = (bf16)(t - (float)hi);
    n2 += t * t;
    const int64_t o = ((int64_t)(c >> 5) * Vp + v) * 32 + (c & 31);
    const int64_t part = (int64_t)(E / 32) * Vp * 32;
    tsplit[o] = hi;
    tsplit[o + part] = hi;
    tsplit[o + 2 * part] = lo;
  }
  tnorm[v] = v < V ? n2 : INFINITY;
}
}  // namespace
// one-time preparation of the table operands of the rounding phase: `buf` (mh_round_split_bytes) <- the split panels, then |T_v|^2.
// table_norm (optional): the caller's own |T_v|^2 [V] (mh_row_sqnorm of the table: what the separate score GEMM adds), copied instead
extern "C" int mh_round_split_table(const float* table, const float* table_norm, int V, int E, void* buf, mh_stream_t stream) {
  MH_CHECK_ARG(table && buf && E == 128 && V > 0 && V <= 768, "round_split_table: bad arguments");
  bf16* ts = (bf16*)buf;
  float* tn = (float*)((char*)buf + (size_t)(3 * E / 32) * 768 * 32 * 2);
  MH_LAUNCH(round_split_kernel, dim3(3), dim3(256), 0, (hipStream_t)stream, table, V, E, ts, tn, 768);
  MH_CHECK_LAUNCH();
  if (table_norm) MH_HIP(hipMemcpyAsync(tn, table_norm, (size_t)V * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MH_OK;
}

extern "C" int mh_down_proj_round_fused(const void* X, int64_t ldx, const void* w0, const float* b0, const void* w2, const float* b2, float* out,
                                        float* out_sqnorm, const void* table_split, int V, int32_t* idx_out, const mh_step_update* upd,
                                        int64_t rows, int E, int H, mh_stream_t stream) {
  MH_CHECK_ARG(X && w0 && b0 && w2 && b2 && out && table_split && idx_out && rows > 0 && ldx >= rows, "down_proj_round_fused: bad arguments");
  MH_CHECK_ARG(mh_down_proj_round_supported(E, H, V), "down_proj_round_fused: shape E=%d H=%d V=%d not served", E, H, V);
  const float* tn = (const float*)((const char*)table_split + (size_t)(3 * E / 32) * 768 * 32 * 2);
  TailArgs g{(const bf16*)X, ldx, rows, E, (const bf16*)w0, b0, (const bf16*)w2, b2, out, out_sqnorm, (const bf16*)table_split, tn, V, idx_out};
  g.x = nullptr;
  if (upd) {
    MH_CHECK_ARG(upd->x && upd->table && upd->coef && (!upd->mask || upd->x_start), "down_proj_round_fused: bad update descriptor");
    MH_CHECK_ARG(!upd->rng || (upd->rng->first_elem % 4 == 0 && (upd->rng->bound <= 0.f || upd->rng->bound >= 0.1f)),
                 "down_proj_round_fused: bad rng descriptor");
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    MH_CHECK_ARG(al(upd->x) && al(upd->x_start) && al(upd->table) && al(upd->pred_xstart) && al(upd->mean_out) && al(upd->noise),
                 "down_proj_round_fused: the update's tensors must be 16-byte aligned");
    g.x = upd->x; g.x_start = upd->x_start; g.mask = upd->mask; g.mask_per_elem = upd->mask_per_elem; g.table = upd->table;
    g.coef = upd->coef; g.clip = upd->clip; g.ddim = upd->ddim; g.pred = upd->pred_xstart; g.mean = upd->mean_out; g.noise = upd->noise;
    g.has_rng = upd->rng != nullptr;
    if (upd->rng)
      g.rng = StepRng{(uint32_t)upd->rng->seed, (uint32_t)(upd->rng->seed >> 32), upd->rng->stream_id, upd->rng->bound, upd->rng->step_counter,
                      (uint64_t)(upd->rng->first_elem >> 2)};
  }
  const int rpb = H == 768 ? 32 : 64;
  const dim3 grid((unsigned)((rows + rpb - 1) / rpb));
  mh_prof_note("tail+round%s rows=%lld E=%d H=%d V=%d", upd ? "+update" : "", (long long)rows, E, H, V);
  if (H == 768) MH_LAUNCH((tail_fused_kernel<768, 4, 32>), grid, dim3(768), 0, (hipStream_t)stream, g);   // 12 waves x 4 table tiles of 16 rows = 768
  else MH_LAUNCH((tail_fused_kernel<512, 6>), grid, dim3(512), 0, (hipStream_t)stream, g);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_down_proj_fused(const void* X, int64_t ldx, const void* w0, const float* b0, const void* w2, const float* b2, float* out,
                                  float* out_sqnorm, int64_t rows, int E, int H, mh_stream_t stream) {
  MH_CHECK_ARG(X && w0 && b0 && w2 && b2 && out && rows > 0 && ldx >= rows, "down_proj_fused: bad arguments");
  MH_CHECK_ARG(mh_down_proj_fused_supported(E, H), "down_proj_fused: shape E=%d H=%d not served", E, H);
  TailArgs g{(const bf16*)X, ldx, rows, E, (const bf16*)w0, b0, (const bf16*)w2, b2, out, out_sqnorm, nullptr, nullptr, 0, nullptr};
  g.x = nullptr;
  const int rpb = H == 768 ? 32 : 64;
  const dim3 grid((unsigned)((rows + rpb - 1) / rpb));
  mh_prof_note("tail rows=%lld E=%d H=%d", (long long)rows, E, H);
  if (H == 768) MH_LAUNCH((tail_fused_kernel<768, 0, 32>), grid, dim3(768), 0, (hipStream_t)stream, g);
  else if (H == 512) MH_LAUNCH((tail_fused_kernel<512>), grid, dim3(512), 0, (hipStream_t)stream, g);
  else MH_LAUNCH((tail_fused_kernel<256>), grid, dim3(256), 0, (hipStream_t)stream, g);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
