// Split-precision kernels: the mode between "fast" (bf16) and "exact" (fp32) - compute_dtype "bf16x3" / "f16x3".
//
// Every operand value v is held as TWO 16-bit floats, hi = rn16(v) and lo = rn16(v - hi), and every product of the reference's fp32
// arithmetic (models/network.py:131-158 run in fp32, models/diffusion.py:914) becomes three matrix-pipe products accumulated in fp32,
//     x w  ~  x_lo w_hi + x_hi w_lo + x_hi w_hi          (x_lo w_lo, relative 2^-16 (bf16) / 2^-22 (f16), is dropped)
// in ONE K loop: per K-step of 32 two LDS-DMA stages, {A lo, W hi} and {A hi, W lo}, and three sets of MFMAs - the third product runs on the
// A hi fragments of the second stage and the W hi fragments kept in registers from the first.  Measured on gfx950 (tools/micro/split_mfma.hip, K = 512,
// max |error| / rms of the fp64 result): bf16 7.9e-3, f16 6.9e-4, bf16x3 1.2e-5, f16x3 2.5e-6, an fp32 fma chain 1.7e-6;
// v_mfma_f32_*_f16 keeps subnormal inputs, so an f16 lo part below 2^-14 keeps an absolute precision of 2^-25.
// f16 parts saturate at +-65504 (bf16 parts have fp32's range).
//
// Layout: a "split panel" matrix [rows, C] is [2][C / 32][ld rows][32] 16-bit - the K32-panel layout of the bf16 path (DESIGN.md
// section 3) once for hi and once for lo, so every LDS-DMA piece of the GEMM is still 16 rows x 64 contiguous bytes.
// LayerNorm / residual sums / softmax statistics stay fp32; GELU is the exact erf form, tanh is tanhf (as in the fp32 mode).
#include <type_traits>

#include "common.h"

namespace {

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

template <typename T> struct Sp;
template <> struct Sp<bf16> {
  typedef bf16x8 x8;
  typedef bf16x4 x4;
  static __device__ __forceinline__ f32x4 mma16(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x16 mma32(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ float sat(float v) { return v; }
  // two values -> packed hi parts and packed lo parts, by TRUNCATION (3 vector instructions per value): hi = the top 16 bits of v (its float
  // value needs no conversion back), lo = the top 16 bits of the exact remainder; |v - hi - lo| <= 2^-15 |v|.  For values whose split is
  // made per use (the attention probabilities); stored operands use the round-to-nearest split().
  static __device__ __forceinline__ void split2(float p0, float p1, uint32_t& hi, uint32_t& lo) {
    const uint32_t u0 = __builtin_bit_cast(uint32_t, p0), u1 = __builtin_bit_cast(uint32_t, p1);
    hi = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float d0 = p0 - __builtin_bit_cast(float, u0 & 0xFFFF0000u), d1 = p1 - __builtin_bit_cast(float, u1 & 0xFFFF0000u);
    lo = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, d1), __builtin_bit_cast(uint32_t, d0), 0x07060302u);
  }
};
template <> struct Sp<f16> {
  typedef f16x8 x8;
  typedef f16x4 x4;
  static __device__ __forceinline__ f32x4 mma16(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x16 mma32(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ float sat(float v) { return fminf(fmaxf(v, -65504.0f), 65504.0f); }
  // (v_cvt_pkrtz_f16_f32: two values per instruction, rounded toward zero; the remainder is exact, so hi + lo keeps 21 bits)
  static __device__ __forceinline__ void split2(float p0, float p1, uint32_t& hi, uint32_t& lo) {
    const auto h = __builtin_amdgcn_cvt_pkrtz(p0, p1);
    hi = __builtin_bit_cast(uint32_t, h);
    const auto l = __builtin_amdgcn_cvt_pkrtz(p0 - (float)h[0], p1 - (float)h[1]);
    lo = __builtin_bit_cast(uint32_t, l);
  }
};

template <typename T, int N>
__device__ __forceinline__ void split(const float (&v)[N], T (&hi)[N], T (&lo)[N]) {
#pragma unroll
  for (int e = 0; e < N; ++e) {
    const float s = Sp<T>::sat(v[e]);
    hi[e] = (T)s;
    lo[e] = (T)(s - (float)hi[e]);
  }
}
template <typename T>
__device__ __forceinline__ void store_split8(T* hi_p, T* lo_p, const float (&v)[8]) {
  T h[8], l[8];
  split<T, 8>(v, h, l);
  typename Sp<T>::x8 hv, lv;
#pragma unroll
  for (int e = 0; e < 8; ++e) { hv[e] = h[e]; lv[e] = l[e]; }
  *reinterpret_cast<typename Sp<T>::x8*>(hi_p) = hv;
  *reinterpret_cast<typename Sp<T>::x8*>(lo_p) = lv;
}
template <typename T>
__device__ __forceinline__ void store_split8_nt(T* hi_p, T* lo_p, const float (&v)[8]) {
  T h[8], l[8];
  split<T, 8>(v, h, l);
  typename Sp<T>::x8 hv, lv;
#pragma unroll
  for (int e = 0; e < 8; ++e) { hv[e] = h[e]; lv[e] = l[e]; }
  f32x4 hr, lr;
  __builtin_memcpy(&hr, &hv, 16);
  __builtin_memcpy(&lr, &lv, 16);
  __builtin_nontemporal_store(hr, reinterpret_cast<f32x4*>(hi_p));
  __builtin_nontemporal_store(lr, reinterpret_cast<f32x4*>(lo_p));
}
template <typename T>
__device__ __forceinline__ void load_split8(const T* hi_p, const T* lo_p, float (&v)[8]) {
  const typename Sp<T>::x8 hv = *reinterpret_cast<const typename Sp<T>::x8*>(hi_p), lv = *reinterpret_cast<const typename Sp<T>::x8*>(lo_p);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)hv[e] + (float)lv[e];
}

// ------------------------------------------------------------------------------------------ fp32 rows <-> split panels
// grid (ceil(rows / 64), kpad / 32): thread (r = tid / 4, c = tid % 4) moves 8 elements of one row of one panel
template <typename T>
__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ x, int64_t ldx, T* __restrict__ out, int64_t ld, int64_t rows,
                                                         int cols, int kpad) {
  const int r = threadIdx.x >> 2, c = threadIdx.x & 3, panel = blockIdx.y;
  const int64_t row = (int64_t)blockIdx.x * 64 + r;
  if (row >= rows) return;
  const int col = panel * 32 + c * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = col + e < cols ? x[row * ldx + col + e] : 0.f;
  T* hi = out + ((int64_t)panel * ld + row) * 32 + c * 8;
  store_split8<T>(hi, hi + (int64_t)(kpad / 32) * ld * 32, v);
}

template <typename T>
__global__ __launch_bounds__(256) void split_join_kernel(const T* __restrict__ in, int64_t ld, int npanels, float* __restrict__ out, int64_t ldo,
                                                         int64_t rows, int cols) {
  const int r = threadIdx.x >> 2, c = threadIdx.x & 3, panel = blockIdx.y;
  const int64_t row = (int64_t)blockIdx.x * 64 + r;
  if (row >= rows) return;
  const int col = panel * 32 + c * 8;
  const T* hi = in + ((int64_t)panel * ld + row) * 32 + c * 8;
  float v[8];
  load_split8<T>(hi, hi + (int64_t)npanels * ld * 32, v);
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (col + e < cols) out[row * ldo + col + e] = v[e];
}

// ------------------------------------------------------------------------------------------ LayerNorm: fp32 rows -> split panels
// One wave per row, the row in registers, two in-register passes (as csrc/norm.hip).  ADD: (pos + x) + emb_t first (network.py:148).
constexpr int LN_MAXCH = 4;   // 8-element chunks per lane: H <= 2048
template <typename T, bool ADD>
__global__ __launch_bounds__(256) void split_ln_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ pos,
                                                       const float* __restrict__ emb_t, const int32_t* __restrict__ emb_row,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ out,
                                                       int64_t ld, int64_t rows, int L, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = H >> 3;
  float v[LN_MAXCH][8];
  float sum = 0.f;
  const float* prow = nullptr;
  const float* trow = nullptr;
  if constexpr (ADD) {
    const int64_t b = row / L, l = row % L;
    prow = pos + l * H;
    trow = emb_t + (int64_t)(emb_row ? emb_row[b] : (int)b) * H;
  }
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      load8(x + row * ldx + c * 8, v[i]);
      if constexpr (ADD) {
        float p[8], t[8];
        load8(prow + c * 8, p);
        load8(trow + c * 8, t);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = (p[e] + v[i][e]) + t[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += v[i][e];
    }
  }
  const float mean = wave_sum(sum) / (float)H;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; sq += d * d; }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)H + eps);
  const int64_t part = (int64_t)(H / 32) * ld * 32;
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      float g[8], bt[8], y[8];
      load8(gamma + c * 8, g);
      load8(beta + c * 8, bt);
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = (v[i][e] - mean) * rstd * g[e] + bt[e];
      T* hi = out + ((int64_t)(c >> 2) * ld + row) * 32 + (c & 3) * 8;
      store_split8<T>(hi, hi + part, y);
    }
  }
}

// ------------------------------------------------------------------------------------------ GEMM
// out = act(A W^T + bias) [+ residual]: A split panels [2][K/32][lda][32] (M rows), W split panels [2][K/32][ldw][32] (N rows).
// 256 x 128 tile, 4 waves (2 x 2, 128 x 64 each), 3-stage LDS-DMA ring of K-steps of 32, two blocks per CU: the bf16 path's
// geometry (csrc/gemm.hip, BigCfg<256,128,2,2,3>), LDS rows of 64 B with the chunk swizzle c ^ G[(r >> 2) & 3], the W-row remap that
// leaves a lane 8 consecutive output columns.
struct SpGemmArgs {
  const void* A; int64_t lda;
  const void* W; int64_t ldw;
  const float* bias; int bias_rows;       // bias_rows: bias[row] instead of bias[col] (the transposed V projection)
  const void* res; int64_t ldr;           // residual: split panels [2][N/32][ldr][32]
  int stream_out;                         // split-panel output larger than the Infinity Cache, read back by a later launch: non-temporal stores
  void* out; int64_t ldo; int out_mode;   // 0 split panels [2][N/32][ldo][32], 1 split row-major (lo part o_part elements after hi), 2 fp32 row-major
  int64_t o_part;
  int64_t M; int N, K;
  int ntiles;                             // persistent launch: output tiles walked by gridDim.x blocks
};

constexpr int GBM = 256, GBN = 128, GNST = 3, GSTAGE = (GBM + GBN) * 64, GTI = 8, GTJ = 4, GPA = 4, GPW = 2, GPIECES = GPA + GPW;

template <int N> __device__ __forceinline__ void sp_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void sp_wait_stages(int stages) {
  if (stages >= 2) sp_wait_vmcnt<2 * GPIECES>();
  else if (stages == 1) sp_wait_vmcnt<GPIECES>();
  else sp_wait_vmcnt<0>();
}
__device__ __forceinline__ int sp_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

template <typename T, int ACT>
__global__ __launch_bounds__(256, 2) void split_gemm_kernel(const SpGemmArgs g) {
  typedef typename Sp<T>::x8 x8;
  __shared__ __attribute__((aligned(16))) char smem[GNST * GSTAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (g.N + GBN - 1) / GBN;
  const int nk0 = g.K / 32, nk = 2 * nk0;   // two DMA stages per K-step of 32 (three products: see `issue`)
  const int fr = lane & 15, fg = lane >> 4;
  constexpr int GSW[4] = {0, 2, 3, 1};
  // DMA coordinates: a piece = 16 rows x 64 B; lane i lands at row i / 4, physical chunk i % 4, which holds logical chunk pc ^ G[..]
  const int rl = lane >> 2, pc = lane & 3, lc = pc ^ GSW[(rl >> 2) & 3];
  // The stage DMA as `buffer_load_dwordx4 ... lds` (round 5, as csrc/gemm.hip BufDma): the tile's rows of part 0, panel 0 as the base of a buffer
  // descriptor (scalar registers), the (part, panel) of the stage as the instruction's scalar offset, a wave's consecutive pieces as its
  // immediate offset (which advances the LDS address with the buffer address) and ONE per-lane offset: no vector address arithmetic per piece.
  // Rows beyond M / N are not clamped: inside the buffer they read other rows (never stored); the descriptor ends with the rows this operand OWNS in
  // its last (part, panel) - A / W may be row windows of a larger buffer (the V^T projection: rows 2H.. of the stacked q | k | v weight) - so rows
  // beyond them read as zeros instead of bytes behind the allocation.
  const int kbA = (int)(g.lda * 64), kbW = (int)(g.ldw * 64);
  const int va = ((wave * GPA) * 16 + rl) * 64 + lc * 16, vw = ((wave * GPW) * 16 + rl) * 64 + lc * 16;
  __amdgpu_buffer_rsrc_t rA, rW;
  auto set_sources = [&](int64_t m0, int n0) {
    rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(g.A)) + m0 * 64, 0, (int)((2ll * nk0 - 1) * g.lda * 64 + (g.M - m0) * 64), 0x00020000);
    rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(g.W)) + (int64_t)n0 * 64, 0, (int)((2ll * nk0 - 1) * g.ldw * 64 + ((int64_t)g.N - n0) * 64), 0x00020000);
  };
  // Two DMA stages per K-step kk serve its three products: stage 2kk = {A lo, W hi}, stage 2kk+1 = {A hi, W lo}; the third product,
  // A hi x W hi, runs in the odd stage on the A fragments it holds and the W hi fragments KEPT IN REGISTERS from the even stage - no
  // third load of either tile, no third round of LDS reads, no third barrier (a third less LDS-DMA, the path that paces these tiles).
  auto issue = [&](int kt) {
    const int kk = kt >> 1, odd = kt & 1;
    const int offA = ((odd ? 0 : nk0) + kk) * kbA, offW = ((odd ? nk0 : 0) + kk) * kbW;
    char* base = smem + (kt % GNST) * GSTAGE;
    auto la = (__attribute__((address_space(3))) void*)(base + (wave * GPA) * 1024);
    auto lw = (__attribute__((address_space(3))) void*)(base + GBM * 64 + (wave * GPW) * 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, la, 16, va, offA, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, la, 16, va, offA, 1024, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, la, 16, va, offA, 2048, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, la, 16, va, offA, 3072, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, lw, 16, vw, offW, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, lw, 16, vw, offW, 1024, 0);
    static_assert(GPA == 4 && GPW == 2, "piece counts are written out (immediate offsets must be literals)");
  };
  const int frag_off = fr * 64 + ((fg ^ GSW[(fr >> 2) & 3]) << 4);
  const int a_off = wm * (GTI * 16 * 64) + frag_off;
  int b_offs[GTJ];
#pragma unroll
  for (int j = 0; j < GTJ; ++j) {
    const int row = 32 * (j >> 1) + 8 * (fr >> 2) + 4 * (j & 1) + (fr & 3);
    b_offs[j] = wn * (GTJ * 16 * 64) + row * 64 + ((fg ^ GSW[(row >> 2) & 3]) << 4);
  }
  // Persistent: gridDim.x blocks (two per CU) walk the g.ntiles output tiles; a tile's first stages are put in flight BEFORE the previous
  // tile's epilogue (the ring is idle then), so the pipeline fill of a 16-K-step tile hides behind the stores.
  const int npro = nk < GNST ? nk : GNST;
  bool pre = false;
  for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
  const int bid = sp_xcd_remap(tile, g.ntiles);
  const int64_t m0 = (int64_t)(bid / tiles_n) * GBM;
  const int n0 = (bid % tiles_n) * GBN;
  set_sources(m0, n0);   // (recomputed when prefetched: the pointers do not live across the epilogue)
  f32x4 acc[GTI][GTJ];
#pragma unroll
  for (int i = 0; i < GTI; ++i)
#pragma unroll
    for (int j = 0; j < GTJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (pre) {
    sp_wait_vmcnt<0>();   // the prefetched stages and the previous epilogue's stores (one counter)
  } else {
    for (int st = 0; st < npro; ++st) issue(st);
    sp_wait_stages(npro - 1);
  }
  __builtin_amdgcn_s_barrier();
  // W fragments by PART, not by stage: bA = W hi (the even stage's operand, used again by the odd stage's hi x hi product), bB = W lo (odd stages
  // only).  bB is idle during an even stage, so the odd stage's fragments are read straight into it; bA is busy until the odd stage's last MFMA
  // row and is re-read behind it.  No second register set and no copies (round 5; the sums keep their order: bit-identical).
  x8 a[GTI], bA[GTJ], bB[GTJ];
#pragma unroll
  for (int j = 0; j < GTJ; ++j) bA[j] = *reinterpret_cast<const x8*>(smem + GBM * 64 + b_offs[j]);
#pragma unroll
  for (int i = 0; i < GTI; ++i) a[i] = *reinterpret_cast<const x8*>(smem + a_off + i * (16 * 64));
  // one stage: the MFMAs of stage kt (odd stages: both W parts) with the fragments of stage kt + 1 read underneath them
  auto stage = [&](int kt, auto oddc, auto nextc) {
    constexpr bool ODD = decltype(oddc)::value;
    constexpr bool next = decltype(nextc)::value;   // (compile-time since round 5: the tile's last stage is peeled; as a run-time flag every fragment read sat behind a scalar branch)
    const char* As = smem + ((kt + 1) % GNST) * GSTAGE;
    const char* Ws = As + GBM * 64;
    if constexpr (next) {
      const int younger = nk - 2 - kt < GNST - 2 ? nk - 2 - kt : GNST - 2;
      sp_wait_stages(younger);
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's reads of stage kt are done
      __builtin_amdgcn_s_barrier();
      if (kt + GNST < nk) issue(kt + GNST);   // slot kt % NST: every wave has read stage kt out of it
      if constexpr (!ODD) {
#pragma unroll
        for (int j = 0; j < GTJ; ++j) bB[j] = *reinterpret_cast<const x8*>(Ws + b_offs[j]);
      }
    }
#pragma unroll
    for (int i = 0; i < GTI; ++i) {
      if constexpr (ODD) {
#pragma unroll
        for (int j = 0; j < GTJ; ++j) acc[i][j] = Sp<T>::mma16(bB[j], a[i], acc[i][j]);
      }
#pragma unroll
      for (int j = 0; j < GTJ; ++j) acc[i][j] = Sp<T>::mma16(bA[j], a[i], acc[i][j]);
      if constexpr (next) a[i] = *reinterpret_cast<const x8*>(As + a_off + i * (16 * 64));
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (ODD && next) {
      {
#pragma unroll
        for (int j = 0; j < GTJ; ++j) bA[j] = *reinterpret_cast<const x8*>(Ws + b_offs[j]);
      }
    }
  };
  for (int kk = 0; kk < nk0 - 1; ++kk) {
    stage(2 * kk, std::false_type{}, std::true_type{});
    stage(2 * kk + 1, std::true_type{}, std::true_type{});
  }
  stage(nk - 2, std::false_type{}, std::true_type{});
  stage(nk - 1, std::true_type{}, std::false_type{});

  // ---- epilogue: lane's 8 consecutive columns of half qh: wcol0 + 32 qh + 8 fg, values acc[i][2 qh + (e >> 2)][e & 3]; row wrow0 + 16 i + fr
  const int wcol0 = n0 + wn * 64;
  const int64_t wrow0 = m0 + wm * 128;
  pre = false;
  if (tile + (int)gridDim.x < g.ntiles) {
    const int nb = sp_xcd_remap(tile + (int)gridDim.x, g.ntiles);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();       // every wave has read the last stages out of the ring
    set_sources((int64_t)(nb / tiles_n) * GBM, (nb % tiles_n) * GBN);
    for (int st = 0; st < npro; ++st) issue(st);
    pre = true;
  }
  T* outT = reinterpret_cast<T*>(g.out);
  float* outF = reinterpret_cast<float*>(g.out);
  const T* res = reinterpret_cast<const T*>(g.res);
  const int64_t r_part = (int64_t)(g.N / 32) * g.ldr * 32;
  const int64_t o_part = g.out_mode == 0 ? (int64_t)(g.N / 32) * g.ldo * 32 : g.o_part;
  // FAST (round 6): an interior tile of the common launch - column bias (or none), split-panel output - without per-lane guards (with
  // divergent guards hipcc waits vmcnt(0) before every store: the tile's stores leave one round trip at a time)
  auto epi = [&](auto fastc) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fastc)::value;
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
      const int col = wcol0 + 32 * qh + 8 * fg;
      if (!FAST && col >= g.N) continue;
      const int nval = (FAST || g.N - col >= 8) ? 8 : g.N - col;   // (a last group may be partial: fp32 row-major outputs only)
      float bv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[e] = (g.bias && (FAST || !g.bias_rows) && e < nval) ? g.bias[col + e] : 0.f;
#pragma unroll
      for (int i = 0; i < GTI; ++i) {
        const int64_t row = wrow0 + 16 * i + fr;
        if (!FAST && row >= g.M) continue;
        float v[8];
        const float br = (!FAST && g.bias && g.bias_rows) ? g.bias[row] : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[i][2 * qh + (e >> 2)][e & 3] + bv[e] + br;
        if constexpr (ACT == MH_ACT_TANH) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
        } else if constexpr (ACT == MH_ACT_GELU_ERF) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
        }
        if (res) {
          const T* rp = res + ((int64_t)(col >> 5) * g.ldr + row) * 32 + (col & 31);
          float rv[8];
          load_split8<T>(rp, rp + r_part, rv);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rv[e];
        }
        if (!FAST && g.out_mode == 2) {
          if (nval == 8) store8(outF + row * g.ldo + col, v);
          else
            for (int e = 0; e < nval; ++e) outF[row * g.ldo + col + e] = v[e];
        } else {
          T* hp = (FAST || g.out_mode == 0) ? outT + ((int64_t)(col >> 5) * g.ldo + row) * 32 + (col & 31) : outT + row * g.ldo + col;
          if (g.stream_out) store_split8_nt<T>(hp, hp + o_part, v);
          else store_split8<T>(hp, hp + o_part, v);
        }
      }
    }
  };
  const bool fast = m0 + GBM <= g.M && n0 + GBN <= g.N && g.out_mode == 0 && !(g.bias && g.bias_rows);   // (block-uniform)
  if (fast) epi(std::true_type{}); else epi(std::false_type{});
  }   // persistent tile loop
}

// ---- dense + residual + LayerNorm in one kernel (hidden size 512): out = LN(A W^T + bias + residual) as split panels.
// A block owns 128 COMPLETE rows (128 x 512 tile, 8 waves as 2 x 4, 64 x 128 each, 3 stages of 40 KB, one block per CU - the bf16 path's
// full-row tile); stages per K-step: {A hi, W lo} then {A lo, W hi}, the hi x hi product on the A hi fragments kept from the first.
// Epilogue as csrc/gemm.hip EPI 3: two-pass statistics, in-lane -> the 4 lanes of a row -> the 4 column waves through LDS.
constexpr int RBM = 128, RBN = 512, RNST = 3, RSTAGE = (RBM + RBN) * 64, RTI = 4, RTJ = 8, RPW = 4, RPIECES = 1 + RPW;
__device__ __forceinline__ void sp_wait_stages_r(int stages) {
  if (stages >= 2) sp_wait_vmcnt<2 * RPIECES>();
  else if (stages == 1) sp_wait_vmcnt<RPIECES>();
  else sp_wait_vmcnt<0>();
}
template <typename T>
__global__ __launch_bounds__(512, 1) void split_gemm_ln_kernel(const SpGemmArgs g, const float* __restrict__ gamma, const float* __restrict__ beta, float eps) {
  typedef typename Sp<T>::x8 x8;
  __shared__ __attribute__((aligned(16))) char smem[RNST * RSTAGE + RBM * 4 * 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nk0 = g.K / 32, nk = 2 * nk0;
  const int fr = lane & 15, fg = lane >> 4;
  constexpr int GSW[4] = {0, 2, 3, 1};
  const int64_t m0 = (int64_t)blockIdx.x * RBM;
  const int rl = lane >> 2, pc = lane & 3, lc = pc ^ GSW[(rl >> 2) & 3];
  // (stage DMA as buffer loads: see split_gemm_kernel)
  const int kbA = (int)(g.lda * 64), kbW = (int)(g.ldw * 64);
  const int va = (wave * 16 + rl) * 64 + lc * 16, vw = ((wave * RPW) * 16 + rl) * 64 + lc * 16;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(g.A)) + m0 * 64, 0,
                                                                      (int)((2ll * nk0 - 1) * g.lda * 64 + (g.M - m0) * 64), 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(g.W)), 0, (int)((2ll * nk0 - 1) * g.ldw * 64 + (int64_t)g.N * 64), 0x00020000);
  auto issue = [&](int kt) {
    const int kk = kt >> 1, odd = kt & 1;
    const int offA = ((odd ? nk0 : 0) + kk) * kbA, offW = ((odd ? 0 : nk0) + kk) * kbW;
    char* base = smem + (kt % RNST) * RSTAGE;
    auto lw = (__attribute__((address_space(3))) void*)(base + RBM * 64 + (wave * RPW) * 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(base + wave * 1024), 16, va, offA, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, lw, 16, vw, offW, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, lw, 16, vw, offW, 1024, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, lw, 16, vw, offW, 2048, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, lw, 16, vw, offW, 3072, 0);
    static_assert(RPW == 4, "piece counts are written out (immediate offsets must be literals)");
  };
  const int frag_off = fr * 64 + ((fg ^ GSW[(fr >> 2) & 3]) << 4);
  const int a_off = wm * (RTI * 16 * 64) + frag_off;
  // W fragment j reads row (j>>2) 64 + 32 ((j&3)>>1) + 8 (fr>>2) + 4 (j&1) + (fr&3) of the wave's 128: the lane's part of the address (and the
  // swizzle, which sees the row only through (2 (fr>>2) + (j&1)) & 3) takes two registers, the rest is a compile-time offset per j
  int b_base[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int row = 8 * (fr >> 2) + 4 * p + (fr & 3);
    b_base[p] = wn * (RTJ * 16 * 64) + row * 64 + ((fg ^ GSW[(row >> 2) & 3]) << 4);
  }
  auto b_off = [&](int j) { return b_base[j & 1] + ((j >> 2) * 64 + 32 * ((j & 3) >> 1)) * 64; };
  f32x4 acc[RTI][RTJ];
#pragma unroll
  for (int i = 0; i < RTI; ++i)
#pragma unroll
    for (int j = 0; j < RTJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int npro = nk < RNST ? nk : RNST;
  for (int st = 0; st < npro; ++st) issue(st);
  sp_wait_stages_r(npro - 1);
  __builtin_amdgcn_s_barrier();
  // (column-group-major MFMA order: a W fragment is re-read in place right after its last use, so only the four A fragments need a second
  // register set for the next stage - the tile's 128 accumulators leave no room for more at two waves per SIMD)
  // A fragments by PART (as the W fragments of split_gemm_kernel): aX = A hi (even stages, and the odd stage's hi x hi product), aY = A lo (odd
  // stages only, read during the even stage); aX is re-read behind the odd stage's last column group.
  x8 aX[RTI], aY[RTI], b[RTJ];
#pragma unroll
  for (int j = 0; j < RTJ; ++j) b[j] = *reinterpret_cast<const x8*>(smem + RBM * 64 + b_off(j));
#pragma unroll
  for (int i = 0; i < RTI; ++i) aX[i] = *reinterpret_cast<const x8*>(smem + a_off + i * (16 * 64));
  auto stage = [&](int kt, auto oddc, auto nextc) {
    constexpr bool ODD = decltype(oddc)::value;
    constexpr bool next = decltype(nextc)::value;
    const char* As = smem + ((kt + 1) % RNST) * RSTAGE;
    const char* Ws = As + RBM * 64;
    if constexpr (next) {
      const int younger = nk - 2 - kt < RNST - 2 ? nk - 2 - kt : RNST - 2;
      sp_wait_stages_r(younger);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
      if (kt + RNST < nk) issue(kt + RNST);
      if constexpr (!ODD) {
#pragma unroll
        for (int i = 0; i < RTI; ++i) aY[i] = *reinterpret_cast<const x8*>(As + a_off + i * (16 * 64));
      }
    }
#pragma unroll
    for (int j = 0; j < RTJ; ++j) {
      if constexpr (ODD) {   // b = W hi: the lo part of A first, then the hi x hi product on the A hi fragments of the even stage
#pragma unroll
        for (int i = 0; i < RTI; ++i) acc[i][j] = Sp<T>::mma16(b[j], aY[i], acc[i][j]);
      }
#pragma unroll
      for (int i = 0; i < RTI; ++i) acc[i][j] = Sp<T>::mma16(b[j], aX[i], acc[i][j]);
      if constexpr (next) b[j] = *reinterpret_cast<const x8*>(Ws + b_off(j));
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (ODD && next) {
      {
#pragma unroll
        for (int i = 0; i < RTI; ++i) aX[i] = *reinterpret_cast<const x8*>(As + a_off + i * (16 * 64));
      }
    }
  };
  for (int kk = 0; kk < nk0 - 1; ++kk) {
    stage(2 * kk, std::false_type{}, std::true_type{});
    stage(2 * kk + 1, std::true_type{}, std::true_type{});
  }
  stage(nk - 2, std::false_type{}, std::true_type{});
  stage(nk - 1, std::true_type{}, std::false_type{});
  // ---- epilogue: v = acc + bias + residual; LayerNorm over the row; split-panel store
  float* red = reinterpret_cast<float*>(smem + RNST * RSTAGE);   // [RBM][4 column waves]
  const T* res = reinterpret_cast<const T*>(g.res);
  const int64_t r_part = (int64_t)(RBN / 32) * g.ldr * 32, o_part = (int64_t)(RBN / 32) * g.ldo * 32;
  const int wcol0 = wn * 128;
  const int64_t wrow0 = m0 + wm * 64;
  float rs[RTI];
#pragma unroll
  for (int i = 0; i < RTI; ++i) rs[i] = 0.f;
#pragma unroll
  for (int qh = 0; qh < RTJ / 2; ++qh) {
    const int col = wcol0 + 32 * qh + 8 * fg;
    float bv[8];
    load8(g.bias + col, bv);
#pragma unroll
    for (int i = 0; i < RTI; ++i) {
      int64_t row = wrow0 + 16 * i + fr; if (row >= g.M) row = g.M - 1;
      const T* rp = res + ((int64_t)(col >> 5) * g.ldr + row) * 32 + (col & 31);
      float rv[8];
      load_split8<T>(rp, rp + r_part, rv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = (acc[i][2 * qh + (e >> 2)][e & 3] + bv[e]) + rv[e];
        acc[i][2 * qh + (e >> 2)][e & 3] = v;
        rs[i] += v;
      }
    }
  }
  const float invN = 1.0f / (float)RBN;
  float mean[RTI], rstd[RTI];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int i = 0; i < RTI; ++i) {
      float v = rs[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (fg == 0) red[(wm * 64 + 16 * i + fr) * 4 + wn] = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < RTI; ++i) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(red + (wm * 64 + 16 * i + fr) * 4);
      const float t = ((t4[0] + t4[1]) + t4[2]) + t4[3];
      if (pass == 0) {
        mean[i] = t * invN;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < RTJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float d = acc[i][j][r] - mean[i]; sq += d * d; }
        rs[i] = sq;
      } else {
        rstd[i] = 1.0f / sqrtf(t * invN + eps);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // reads done before the second pass overwrites `red`
  }
  T* outT = reinterpret_cast<T*>(g.out);
#pragma unroll
  for (int qh = 0; qh < RTJ / 2; ++qh) {
    const int col = wcol0 + 32 * qh + 8 * fg;
    float gv[8], bt[8];
    load8(gamma + col, gv);
    load8(beta + col, bt);
#pragma unroll
    for (int i = 0; i < RTI; ++i) {
      const int64_t row = wrow0 + 16 * i + fr;
      if (row < g.M) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (acc[i][2 * qh + (e >> 2)][e & 3] - mean[i]) * rstd[i] * gv[e] + bt[e];
        T* hp = outT + ((int64_t)(col >> 5) * g.ldo + row) * 32 + (col & 31);
        store_split8<T>(hp, hp + o_part, v);
      }
    }
  }
}

template <typename T>
int launch_gemm_ln(const SpGemmArgs& g, const float* gamma, const float* beta, float eps, hipStream_t s) {
  MH_CHECK_ARG((int64_t)(g.K / 16) * g.lda * 64 < (1ll << 31) && (int64_t)(g.K / 16) * g.ldw * 64 < (1ll << 31),
               "split_gemm_res_ln: an operand beyond 2 GiB (K=%d lda=%lld ldw=%lld): buffer-descriptor addressing is 32-bit", g.K, (long long)g.lda, (long long)g.ldw);
  const dim3 grid((unsigned)((g.M + RBM - 1) / RBM)), block(512);
  mh_prof_note("split tile=128x512 +LN M=%lld N=%d K=3x%d", (long long)g.M, g.N, g.K);
  MH_LAUNCH((split_gemm_ln_kernel<T>), grid, block, 0, s, g, gamma, beta, eps);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

inline int sp_device_cus() {
  static int cus[MH_MAX_DEVICES] = {0};
  const int dev = mh_current_device();
  if (!cus[dev]) {
    int n = 0;
    cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cus[dev];
}
template <typename T>
int launch_gemm(const SpGemmArgs& g0, int act, hipStream_t s) {
  const int64_t tiles = (int64_t)((g0.M + GBM - 1) / GBM) * ((g0.N + GBN - 1) / GBN);
  MH_CHECK_ARG(tiles > 0 && tiles < (1ll << 31), "split_gemm: bad grid (M=%lld N=%d)", (long long)g0.M, g0.N);
  MH_CHECK_ARG((int64_t)(g0.K / 16) * g0.lda * 64 < (1ll << 31) && (int64_t)(g0.K / 16) * g0.ldw * 64 < (1ll << 31),
               "split_gemm: an operand of [2][K/32][ld][32] beyond 2 GiB (K=%d lda=%lld ldw=%lld): buffer-descriptor addressing is 32-bit", g0.K,
               (long long)g0.lda, (long long)g0.ldw);
  SpGemmArgs g = g0;
  g.ntiles = (int)tiles;
  const int64_t slots = 2 * (int64_t)sp_device_cus();
  const dim3 grid((unsigned)(tiles < slots ? tiles : slots)), block(256);
  mh_prof_note("split tile=256x128 act=%d M=%lld N=%d K=3x%d out=%d", act, (long long)g.M, g.N, g.K, g.out_mode);
  switch (act) {
    case MH_ACT_NONE: MH_LAUNCH((split_gemm_kernel<T, MH_ACT_NONE>), grid, block, 0, s, g); break;
    case MH_ACT_TANH: MH_LAUNCH((split_gemm_kernel<T, MH_ACT_TANH>), grid, block, 0, s, g); break;
    case MH_ACT_GELU_ERF: MH_LAUNCH((split_gemm_kernel<T, MH_ACT_GELU_ERF>), grid, block, 0, s, g); break;
    default: mh_set_error("split_gemm: unsupported activation %d", act); return MH_ERR_UNSUPPORTED;
  }
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// ------------------------------------------------------------------------------------------ attention
// Unmasked self-attention (HF BertSelfAttention from network.py:151) on split operands, flash style: one workgroup = 4 waves = 128
// queries of one (batch, head), each wave 32 queries, keys in tiles of 64 - the geometry and LDS images of attn_bf16_kernel
// (csrc/attention.hip), every fragment once for hi and once for lo:
//   S^T  = K_hi Q_lo^T + K_lo Q_hi^T + K_hi Q_hi^T        (32x32x16, the query on the lane)
//   P    = exp2((S - max) scale log2 e) in fp32, split into P_hi + P_lo in registers (no LDS round trip: the accumulator IS the B operand)
//   O^T += V^T_hi P_lo^T + V^T_lo P_hi^T + V^T_hi P_hi^T
// q / k: split row-major [2][tokens][ldq] (q at column head dh, k at column koff + head dh; lo part qk_part elements after hi);
// vt: split row-major [2][H][ldv] (the transposed V projection: row head dh + d, column token); ctx: split panels [2][H/32][ld_ctx][32].
// NW waves of 32 queries per block: 4 (128 queries), or 8 when the sequence has 256 queries or more - every block streams ALL keys and
// values of its (batch, head) through LDS, so twice the queries per block halve the global loads and LDS writes per query.
template <typename T, int DH, int NW = 4>
__global__ __launch_bounds__(64 * NW) void split_attn_kernel(const T* __restrict__ QK, int64_t ldq, int koff, int64_t qk_part, const T* __restrict__ VT,
                                                         int64_t ldv, int64_t vt_part, T* __restrict__ ctx, int64_t ld_ctx, int H, int L, int nh,
                                                         float scale_log2e) {
  typedef typename Sp<T>::x8 x8;
  typedef typename Sp<T>::x4 x4;
  constexpr int CH = DH / 8;            // 16-B chunks per K row
  constexpr int RPB = 128 / DH;         // K rows per 256-B bank row
  constexpr int KROWB = DH * 2;
  constexpr int KT_BYTES = 64 * KROWB, VT_BYTES = DH * 128, PART = KT_BYTES + VT_BYTES, BUF = 2 * PART;
  constexpr int KS = DH / 16;           // k-steps of QK^T
  constexpr int DT = (DH + 31) / 32;    // 32-row d tiles of O^T (DH 16: half a tile, the upper rows are zero)
  constexpr int THREADS = 64 * NW;
  constexpr int KCH = (64 * CH + THREADS - 1) / THREADS;   // K chunks per thread
  constexpr int VCH = (DH * 8 + THREADS - 1) / THREADS;    // V^T 16-B chunks per thread
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, lq = lane & 31;
  const int bh = blockIdx.y, b = bh / nh, head = bh % nh;
  const int q0 = blockIdx.x * (32 * NW) + wave * 32;
  const T* Qb[2] = {QK + (int64_t)b * L * ldq + head * DH, QK + qk_part + (int64_t)b * L * ldq + head * DH};
  const T* Kb[2] = {Qb[0] + koff, Qb[1] + koff};
  const T* Vb[2] = {VT + (int64_t)head * DH * ldv + (int64_t)b * L, VT + vt_part + (int64_t)head * DH * ldv + (int64_t)b * L};

  x8 qf[2][KS];   // [part][k-step]: lane holds Q[q0 + lq][16 ks + 8 h .. + 8]
  {
    int qr = q0 + lq; if (qr >= L) qr = L - 1;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) qf[p][ks] = *reinterpret_cast<const x8*>(Qb[p] + (int64_t)qr * ldq + 16 * ks + 8 * h);
  }
  f32x16 o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  f32x4 stK[2][KCH], stV[2][VCH];
  const int ntiles = (L + 63) / 64;
  auto issue = [&](int t) {
    const int k0 = t * 64;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int j = 0; j < KCH; ++j) {
        const int qd = tid + THREADS * j;
        if (qd < 64 * CH) {
          const int row = qd / CH, c = qd % CH;
          int kr = k0 + row; if (kr >= L) kr = L - 1;
          stK[p][j] = *reinterpret_cast<const f32x4*>(Kb[p] + (int64_t)kr * ldq + c * 8);
        }
      }
#pragma unroll
      for (int j = 0; j < VCH; ++j) {
        const int qd = tid + THREADS * j, d = qd >> 3, c = qd & 7;
        const int key = k0 + c * 8;
        if (d < DH && key < L) stV[p][j] = *reinterpret_cast<const f32x4*>(Vb[p] + (int64_t)d * ldv + key);   // (L % 8 == 0)
        else stV[p][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      char* kb = smem + buf * BUF + p * PART;
      char* vb = kb + KT_BYTES;
#pragma unroll
      for (int j = 0; j < KCH; ++j) {
        const int qd = tid + THREADS * j;
        if (qd < 64 * CH) {
          const int row = qd / CH, c = qd % CH;
          *reinterpret_cast<f32x4*>(kb + row * KROWB + ((c ^ ((row / RPB) & (CH - 1))) << 4)) = stK[p][j];
        }
      }
#pragma unroll
      for (int j = 0; j < VCH; ++j) {
        const int qd = tid + THREADS * j, d = qd >> 3, c = qd & 7;
        if (d < DH) {
          const int sblk = c >> 1, sw = ((d >> 1) ^ d) & 7, half = (c & 1) * 8;   // (^ d: rows d, d + 1 of one 16-lane ds_write_b64 group take chunks of opposite parity - with (d >> 1) alone they met on the same banks: 2 M conflict cycles per launch)
          // keys 8c .. 8c+3 -> chunk 2 sblk, keys 8c+4 .. 8c+7 -> chunk 2 sblk + 1: the key order of the P fragment's registers
          f32x2 lo = {stV[p][j][0], stV[p][j][1]}, hi = {stV[p][j][2], stV[p][j][3]};
          *reinterpret_cast<f32x2*>(vb + d * 128 + (((2 * sblk) ^ sw) << 4) + half) = lo;
          *reinterpret_cast<f32x2*>(vb + d * 128 + (((2 * sblk + 1) ^ sw) << 4) + half) = hi;
        }
      }
    }
  };

  issue(0);
  commit(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntiles) issue(t + 1);
    const char* kb0 = smem + cur * BUF;           // hi part; lo part PART bytes further
    const char* vb0 = kb0 + KT_BYTES;
    // ---- S^T tiles: [2 x 32 keys][32 queries]
    f32x16 s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
      const int row = kt * 32 + lq;
      const int sw = (row / RPB) & (CH - 1);
      x8 kh[KS], kl[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        kh[ks] = *reinterpret_cast<const x8*>(kb0 + row * KROWB + (((2 * ks + h) ^ sw) << 4));
        kl[ks] = *reinterpret_cast<const x8*>(kb0 + PART + row * KROWB + (((2 * ks + h) ^ sw) << 4));
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) s[kt] = Sp<T>::mma32(kh[ks], qf[1][ks], s[kt]);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) s[kt] = Sp<T>::mma32(kl[ks], qf[0][ks], s[kt]);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) s[kt] = Sp<T>::mma32(kh[ks], qf[0][ks], s[kt]);
    }
    // ---- online softmax; register r of tile kt is key t*64 + kt*32 + (r&3) + 8(r>>2) + 4h
    const int k0 = t * 64;
    if (k0 + 64 > L) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h >= L) s[kt][r] = -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, s[kt][r]), s[kt][r + 1]);   // (v_max3_f32)
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
    const float mb = m_new * scale_log2e;
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][r], scale_log2e, -mb));   // one v_fma + one v_exp per score (2^-inf = 0: masked keys)
        s[kt][r] = p;
        psum += p;
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    // ---- O^T += V^T P^T : k-step sp = 2kt + s2 takes registers 8*s2 .. 8*s2+7 of s[kt], split pairwise into packed hi / lo parts
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        uint32_t hw[4], lw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) Sp<T>::split2(s[kt][8 * s2 + 2 * j], s[kt][8 * s2 + 2 * j + 1], hw[j], lw[j]);
        x8 ph, pl;
        __builtin_memcpy(&ph, hw, 16);
        __builtin_memcpy(&pl, lw, 16);
        const int sp = 2 * kt + s2;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int d = dt * 32 + lq;
          x8 vh, vl;
          if (DH % 32 == 0 || d < DH) {
            vh = *reinterpret_cast<const x8*>(vb0 + d * 128 + (((2 * sp + h) ^ (((d >> 1) ^ d) & 7)) << 4));
            vl = *reinterpret_cast<const x8*>(vb0 + PART + d * 128 + (((2 * sp + h) ^ (((d >> 1) ^ d) & 7)) << 4));
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { vh[j] = (T)0.f; vl[j] = (T)0.f; }
          }
          o[dt] = Sp<T>::mma32(vh, pl, o[dt]);
          o[dt] = Sp<T>::mma32(vl, ph, o[dt]);
          o[dt] = Sp<T>::mma32(vh, ph, o[dt]);
        }
      }
    if (t + 1 < ntiles) commit(cur ^ 1);
    __syncthreads();
  }
  // ---- normalise and store: lane holds query q0+lq, d = dt*32 + 8*(r>>2) + 4h + (r&3)
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int qr = q0 + lq;
  if (qr < L) {
    const int64_t tok = (int64_t)b * L + qr;
    const int64_t c_part = (int64_t)(H / 32) * ld_ctx * 32;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d = dt * 32 + 8 * rg + 4 * h;
        if (DH % 32 != 0 && d >= DH) continue;
        const int col = head * DH + d;
        float v[4];
        T hv[4], lv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = o[dt][rg * 4 + e] * inv;
        split<T, 4>(v, hv, lv);
        x4 h4, l4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { h4[e] = hv[e]; l4[e] = lv[e]; }
        T* dst = ctx + ((int64_t)(col >> 5) * ld_ctx + tok) * 32 + (col & 31);
        *reinterpret_cast<x4*>(dst) = h4;
        *reinterpret_cast<x4*>(dst + c_part) = l4;
      }
  }
}

template <typename T>
int launch_attn(const void* qk, int64_t ldq, int koff, int64_t qk_part, const void* vt, int64_t ldv, int64_t vt_part, void* ctx, int64_t ld_ctx, int B, int L,
                int nh, int dh, float scale, hipStream_t s) {
  const float sl2 = scale * 1.4426950408889634f;
  const int H = nh * dh;
  mh_prof_note("split attention B=%d L=%d nh=%d dh=%d", B, L, nh, dh);
  if (dh == 64 && L >= 256) {   // eight waves: 256 queries per block
    const dim3 grid8((unsigned)((L + 255) / 256), (unsigned)(B * nh)), block8(512);
    MH_LAUNCH((split_attn_kernel<T, 64, 8>), grid8, block8, 0, s, (const T*)qk, ldq, koff, qk_part, (const T*)vt, ldv, vt_part, (T*)ctx, ld_ctx, H, L, nh, sl2);
    MH_CHECK_LAUNCH();
    return MH_OK;
  }
  const dim3 grid((unsigned)((L + 127) / 128), (unsigned)(B * nh)), block(256);
  switch (dh) {
    case 16: MH_LAUNCH((split_attn_kernel<T, 16>), grid, block, 0, s, (const T*)qk, ldq, koff, qk_part, (const T*)vt, ldv, vt_part, (T*)ctx, ld_ctx, H, L, nh, sl2); break;
    case 32: MH_LAUNCH((split_attn_kernel<T, 32>), grid, block, 0, s, (const T*)qk, ldq, koff, qk_part, (const T*)vt, ldv, vt_part, (T*)ctx, ld_ctx, H, L, nh, sl2); break;
    case 64: MH_LAUNCH((split_attn_kernel<T, 64>), grid, block, 0, s, (const T*)qk, ldq, koff, qk_part, (const T*)vt, ldv, vt_part, (T*)ctx, ld_ctx, H, L, nh, sl2); break;
    default: mh_set_error("split_attention: head dim %d not in {16, 32, 64}", dh); return MH_ERR_UNSUPPORTED;
  }
  MH_CHECK_LAUNCH();
  return MH_OK;
}

}  // namespace

#define SP_DISPATCH(dtype, CALL_BF16, CALL_F16)                                                   \
  do {                                                                                            \
    if ((dtype) == MH_BF16X3) return CALL_BF16;                                                   \
    if ((dtype) == MH_F16X3) return CALL_F16;                                                     \
    mh_set_error("split kernels: dtype %d is neither MH_BF16X3 nor MH_F16X3", (int)(dtype));      \
    return MH_ERR_INVALID;                                                                        \
  } while (0)

namespace {
template <typename T>
int pack_impl(const float* x, int64_t ldx, void* out, int64_t ld, int64_t rows, int cols, int kpad, hipStream_t s) {
  const dim3 grid((unsigned)((rows + 63) / 64), (unsigned)(kpad / 32)), block(256);
  MH_LAUNCH((split_pack_kernel<T>), grid, block, 0, s, x, ldx, (T*)out, ld, rows, cols, kpad);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
template <typename T>
int join_impl(const void* in, int64_t ld, float* out, int64_t ldo, int64_t rows, int cols, int cpad, hipStream_t s) {
  const dim3 grid((unsigned)((rows + 63) / 64), (unsigned)(cpad / 32)), block(256);
  MH_LAUNCH((split_join_kernel<T>), grid, block, 0, s, (const T*)in, ld, cpad / 32, out, ldo, rows, cols);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
template <typename T>
int ln_impl(const float* x, int64_t ldx, const float* pos, const float* emb_t, const int32_t* emb_row, const float* gamma, const float* beta, void* out,
            int64_t ld, int64_t rows, int L, int H, float eps, hipStream_t s) {
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (pos) MH_LAUNCH((split_ln_kernel<T, true>), grid, block, 0, s, x, ldx, pos, emb_t, emb_row, gamma, beta, (T*)out, ld, rows, L, H, eps);
  else MH_LAUNCH((split_ln_kernel<T, false>), grid, block, 0, s, x, ldx, pos, emb_t, emb_row, gamma, beta, (T*)out, ld, rows, L, H, eps);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
}  // namespace

extern "C" int mh_split_supported(int dtype) { return dtype == MH_BF16X3 || dtype == MH_F16X3; }

extern "C" int mh_split_pack(const float* x, int64_t ldx, void* out, int64_t ld_rows, int64_t rows, int cols, int kpad, int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(x && out && rows > 0 && cols > 0 && kpad >= cols && kpad % 32 == 0 && ld_rows >= rows && ldx >= cols, "split_pack: bad arguments");
  SP_DISPATCH(dtype, pack_impl<bf16>(x, ldx, out, ld_rows, rows, cols, kpad, (hipStream_t)stream),
              pack_impl<f16>(x, ldx, out, ld_rows, rows, cols, kpad, (hipStream_t)stream));
}

extern "C" int mh_split_join(const void* in, int64_t ld_rows, float* out, int64_t ldo, int64_t rows, int cols, int cpad, int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(in && out && rows > 0 && cols > 0 && cpad >= cols && cpad % 32 == 0 && ld_rows >= rows && ldo >= cols, "split_join: bad arguments");
  SP_DISPATCH(dtype, join_impl<bf16>(in, ld_rows, out, ldo, rows, cols, cpad, (hipStream_t)stream),
              join_impl<f16>(in, ld_rows, out, ldo, rows, cols, cpad, (hipStream_t)stream));
}

extern "C" int mh_split_layernorm(const float* x, int64_t ldx, const float* pos, const float* emb_t, const int32_t* emb_row, const float* gamma,
                                  const float* beta, void* out, int64_t ld_rows, int64_t rows, int L, int H, float eps, int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(x && gamma && beta && out && rows > 0 && H % 32 == 0 && H <= 64 * 8 * LN_MAXCH && ldx >= H && ldx % 4 == 0 && ld_rows >= rows,
               "split_layernorm: bad arguments (H %% 32 == 0, H <= 2048)");
  MH_CHECK_ARG(!pos || (emb_t && L > 0 && rows % L == 0), "split_layernorm: position / time rows need emb_t and L | rows");
  SP_DISPATCH(dtype, ln_impl<bf16>(x, ldx, pos, emb_t, emb_row, gamma, beta, out, ld_rows, rows, L, H, eps, (hipStream_t)stream),
              ln_impl<f16>(x, ldx, pos, emb_t, emb_row, gamma, beta, out, ld_rows, rows, L, H, eps, (hipStream_t)stream));
}

extern "C" int mh_split_gemm(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, int bias_rows, const void* residual, int64_t ldr,
                             void* out, int64_t ldo, int out_mode, int64_t out_part, int64_t M, int N, int K, int act, int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(A && W && out && M > 0 && N > 0 && K > 0 && K % 32 == 0 && lda >= M && ldw >= N, "split_gemm: bad operands (K %% 32 == 0, ld >= rows)");
  MH_CHECK_ARG(out_mode >= 0 && out_mode <= 2, "split_gemm: out_mode %d not in 0..2", out_mode);
  MH_CHECK_ARG(out_mode == 2 ? (ldo >= N && ldo % 4 == 0) : (N % 8 == 0 && (out_mode == 0 ? (N % 32 == 0 && ldo >= M) : (ldo >= N && ldo % 8 == 0 && out_part % 8 == 0))),
               "split_gemm: output shape / stride not supported by mode %d (N=%d ldo=%lld)", out_mode, N, (long long)ldo);
  MH_CHECK_ARG(!residual || (N % 32 == 0 && ldr >= M), "split_gemm: the residual is a split panel matrix [2][N/32][ldr][32]");
  const int stream_out = out_mode == 0 && (int64_t)M * N * 4 > (192ll << 20);   // (config 2: FFN1's 268 MB intermediate)
  SpGemmArgs g{A, lda, W, ldw, bias, bias_rows, residual, ldr, stream_out, out, ldo, out_mode, out_part, M, N, K};
  SP_DISPATCH(dtype, launch_gemm<bf16>(g, act, (hipStream_t)stream), launch_gemm<f16>(g, act, (hipStream_t)stream));
}

extern "C" int mh_split_gemm_res_ln_supported(int N) { return N == RBN; }
extern "C" int mh_split_gemm_res_ln(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const void* residual, int64_t ldr,
                                    const float* gamma, const float* beta, float eps, void* out, int64_t ldo, int64_t M, int N, int K, int dtype,
                                    mh_stream_t stream) {
  MH_CHECK_ARG(A && W && bias && residual && gamma && beta && out && M > 0 && K > 0 && K % 32 == 0 && lda >= M && ldw >= N && ldr >= M && ldo >= M,
               "split_gemm_res_ln: bad operands");
  MH_CHECK_ARG(mh_split_gemm_res_ln_supported(N), "split_gemm_res_ln: the full-row tile is built for N = %d (got %d)", RBN, N);
  SpGemmArgs g{A, lda, W, ldw, bias, 0, residual, ldr, 0, out, ldo, 0, 0, M, N, K};
  SP_DISPATCH(dtype, launch_gemm_ln<bf16>(g, gamma, beta, eps, (hipStream_t)stream), launch_gemm_ln<f16>(g, gamma, beta, eps, (hipStream_t)stream));
}

extern "C" int mh_split_attention(const void* qk, int64_t ld_qk, int k_offset, int64_t qk_part, const void* vt, int64_t ld_vt, int64_t vt_part, void* ctx,
                                  int64_t ld_ctx, int B, int L, int nh, int dh, float scale, int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(qk && vt && ctx && B > 0 && L > 0 && L % 8 == 0 && nh > 0 && ld_qk % 8 == 0 && k_offset % 8 == 0 && qk_part % 8 == 0 && ld_vt % 8 == 0 &&
                   vt_part % 8 == 0 && (nh * dh) % 32 == 0 && ld_ctx >= (int64_t)B * L && (int64_t)B * nh <= 65535,
               "split_attention: bad arguments (L %% 8 == 0, 16-byte aligned strides)");
  SP_DISPATCH(dtype, launch_attn<bf16>(qk, ld_qk, k_offset, qk_part, vt, ld_vt, vt_part, ctx, ld_ctx, B, L, nh, dh, scale, (hipStream_t)stream),
              launch_attn<f16>(qk, ld_qk, k_offset, qk_part, vt, ld_vt, vt_part, ctx, ld_ctx, B, L, nh, dh, scale, (hipStream_t)stream));
}
