// GEMM family: out = act(A W^T + bias) [+ residual]  and the fused QKV projection with head scatter.
//
// CDNA4 mapping (one design for both element types, 128x128 block tile, 4 waves as 2(M) x 2(N),
// each wave a 64x64 sub-tile = 4x4 MFMA tiles of 16x16):
//   bf16: v_mfma_f32_16x16x32_bf16, K-tile 64.  LDS rows are 128 B; the 16-B chunk c of row r lives
//         at chunk (c ^ (r & 7)) so that the ds_read_b128 fragment reads (16 rows x 2 chunks per
//         16-lane group) hit 16 distinct 16-B slots of the 256-B bank row: conflict-free.
//         Staging is either register-staged (global_load_dwordx4 -> ds_write_b128) or direct
//         global_load_lds_dwordx4 (LDS image is lane-linear, so the swizzle is applied to the
//         per-lane SOURCE address).
//   f32:  v_mfma_f32_16x16x4_f32 (exact fp32 fma chain), K-tile 16.  The k index served by lane
//         group g in instruction s is 4g+s for BOTH operands, so one ds_read_b128 per tile row
//         feeds four MFMAs.  LDS rows are padded to 96 B (slot = 6*row + g: conflict-free).
//   Both: double-buffered LDS, one barrier per K-tile; the epilogue stages each wave's
//         accumulators through LDS so that bias / activation / residual / stores work on 8
//         contiguous columns per lane (16-B bf16 stores).
// nn.Linear convention: W is [N, K] row-major ("B^T"), which is exactly the k-contiguous layout
// the MFMA B operand wants, so no weight transposition happens anywhere.
#include "common.h"

namespace {

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
  int L, H, nh, dh;
  int dbg;  // timing-only ablation bits (mh_gemm_set_debug): 1 no DMA, 2 no MFMA, 4 no stores
};

template <typename T> struct Tile;
template <> struct Tile<bf16> { static constexpr int BK = 64, ROWB = 128, CHUNKS = 8; };
template <> struct Tile<float> { static constexpr int BK = 16, ROWB = 96, CHUNKS = 4; };

constexpr int BM = 128, BN = 128, CS_LD = 68;

template <typename T>
__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case MH_ACT_TANH: return tanhf(v);
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

template <typename T, int EPI, int GLDS>
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
  const T* __restrict__ A = reinterpret_cast<const T*>(g.A);
  const T* __restrict__ W = reinterpret_cast<const T*>(g.W);
  const int nk = g.K / BK;
  constexpr int EPC = 16 / sizeof(T);  // elements per 16-B chunk

  // ---- per-thread staging coordinates (fixed over the K loop)
  const T* srcA[NCH];
  const T* srcW[NCH];
  int ldsoff[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    int row, c, off;
    if constexpr (GLDS) {
      // wave-instruction (wave*NCH + j) fills 1 KiB = 8 rows; lane i lands at row i/8, phys chunk i%8
      const int r8 = (wave * NCH + j) * 8;
      row = r8 + (lane >> 3);
      c = (lane & 7) ^ (row & 7);                   // logical chunk that belongs at this LDS slot
      off = r8 * ROWB;                              // wave-uniform LDS base of the instruction
    } else {
      const int qd = tid + 256 * j;
      row = qd / CHUNKS;
      c = qd % CHUNKS;
      if constexpr (sizeof(T) == 2) off = row * ROWB + ((c ^ (row & 7)) << 4);
      else off = row * ROWB + (c << 4);
    }
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

  f32x4 stA[NCH], stW[NCH];  // staging registers (register-staged mode)

  auto issue = [&](int kt, int buf) {
    const int koff = kt * BK;
    char* base = smem + buf * 2 * TILE_BYTES;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      if constexpr (GLDS) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[j] + koff),
                                         (__attribute__((address_space(3))) void*)(base + ldsoff[j]), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW[j] + koff),
                                         (__attribute__((address_space(3))) void*)(base + TILE_BYTES + ldsoff[j]), 16, 0, 0);
      } else {
        stA[j] = *reinterpret_cast<const f32x4*>(srcA[j] + koff);
        stW[j] = *reinterpret_cast<const f32x4*>(srcW[j] + koff);
      }
    }
  };
  auto commit = [&](int buf) {
    if constexpr (GLDS) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      char* base = smem + buf * 2 * TILE_BYTES;
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        *reinterpret_cast<f32x4*>(base + ldsoff[j]) = stA[j];
        *reinterpret_cast<f32x4*>(base + TILE_BYTES + ldsoff[j]) = stW[j];
      }
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
  T* outT = reinterpret_cast<T*>(g.out);
  float* outF = reinterpret_cast<float*>(g.out);
  const T* res = reinterpret_cast<const T*>(g.residual);
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
// bf16 "big tile" kernel (the throughput path).  The first kernel above is latency-bound at the
// denoiser's shapes (K = 512: eight K-steps, one tile of prefetch): rocprof showed ~14 us per 128x128
// tile against 1.7 us of MFMA time.  This one is built around keeping loads in flight:
//   * 256(M) x 128(N) block tile, 4 waves as 2 x 2, each wave 128 x 64 (8 x 4 MFMA tiles of
//     16x16x32 = 128 accumulator registers): per K-step 12 ds_read_b128 feed 32 MFMAs.
//   * K-step 32, THREE-stage LDS ring filled by global_load_lds_dwordx4 (24 KiB per stage, 6 DMA
//     instructions per wave), counted `s_waitcnt vmcnt(6)` + raw s_barrier so that the next stage
//     stays in flight ACROSS the barrier (a __syncthreads() would drain it); 72 KiB per block ->
//     two blocks per CU, whose prologues / epilogues overlap each other's main loops.
//   * 64-B LDS rows, chunk c of row r stored at c ^ G[(r>>2)&3], G = {0,2,3,1}: every 16-lane group
//     of a ds_read_b128 fragment read hits 16 distinct 16-B slots (conflict-free); the DMA writes LDS
//     linearly, so the swizzle is applied to the per-lane global SOURCE address.
//   * the MFMA is issued with the operands SWAPPED (D = W_tile . A_tile^T): a lane then owns 4
//     consecutive output columns of one row, so bias / activation / residual / store work straight
//     from the accumulators with 8-byte accesses - no LDS round trip, no barrier in the epilogue.
//     (V^T of the QKV projection wants 4 consecutive TOKENS per lane instead: those waves issue the
//     MFMA un-swapped.)
constexpr int B2M = 256, B2N = 128, B2K = 32, B2STAGES = 3;
constexpr int B2_STAGE_BYTES = (B2M + B2N) * 64;

template <bool SWAP, int DBG>
__device__ __forceinline__ void big_mainloop(f32x4 (&acc)[8][4], const char* smem, const char* const (&srcA)[4],
                                             const char* const (&srcW)[2], const int (&ldsA)[4], const int (&ldsW)[2],
                                             int nk, int a_off, int b_off) {
  auto issue = [&](int kt) {
    if constexpr ((DBG & 1) != 0) return;
    char* base = const_cast<char*>(smem) + (kt % B2STAGES) * B2_STAGE_BYTES;
    const int koff = kt * (B2K * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[j] + koff),
                                       (__attribute__((address_space(3))) void*)(base + ldsA[j]), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW[j] + koff),
                                       (__attribute__((address_space(3))) void*)(base + B2M * 64 + ldsW[j]), 16, 0, 0);
  };
  issue(0);
  if (nk > 1) issue(1);
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt must have landed; stage kt+1 (6 DMA ops of this wave) may stay in flight
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // everyone's share of stage kt is in LDS; stage kt-1 is fully consumed
    if (kt + 2 < nk) issue(kt + 2); // refill the slot stage kt-1 occupied
    const char* As = smem + (kt % B2STAGES) * B2_STAGE_BYTES;
    const char* Ws = As + B2M * 64;
    bf16x8 a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const bf16x8*>(As + a_off + i * (16 * 64));
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(Ws + b_off + j * (16 * 64));
    if constexpr ((DBG & 2) != 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(a[i]));
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(b[j]));
      continue;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
  }
}

template <int EPI, int ACT, int DBG = 0>
__global__ __launch_bounds__(256, 2) void gemm_big_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) char smem[B2STAGES * B2_STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (g.N + B2N - 1) / B2N;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t m0 = (int64_t)(bid / tiles_n) * B2M;
  const int n0 = (bid % tiles_n) * B2N;
  const int nk = g.K / B2K;
  const int fr = lane & 15, fg = lane >> 4;
  constexpr int GSW[4] = {0, 2, 3, 1};

  // DMA coordinates: one instruction covers 16 rows x 64 B; lane i lands at row i/4, physical chunk i%4
  const char* srcA[4];
  const char* srcW[2];
  int ldsA[4], ldsW[2];
  {
    const int rl = lane >> 2, pc = lane & 3;
    const int lc = pc ^ GSW[(rl >> 2) & 3];          // logical chunk stored at this physical slot
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r16 = (wave * 4 + j) * 16;
      int64_t ra = m0 + r16 + rl; if (ra >= g.M) ra = g.M - 1;
      srcA[j] = reinterpret_cast<const char*>(g.A) + (ra * g.lda + lc * 8) * 2;
      ldsA[j] = r16 * 64;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r16 = (wave * 2 + j) * 16;
      int rw = n0 + r16 + rl; if (rw >= g.N) rw = g.N - 1;
      srcW[j] = reinterpret_cast<const char*>(g.W) + ((int64_t)rw * g.ldw + lc * 8) * 2;
      ldsW[j] = r16 * 64;
    }
  }
  const int frag_off = fr * 64 + ((fg ^ GSW[(fr >> 2) & 3]) << 4);
  const int a_off = wm * (128 * 64) + frag_off, b_off = wn * (64 * 64) + frag_off;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wcol0 = n0 + wn * 64;
  const int64_t wrow0 = m0 + wm * 128;
  if constexpr (EPI == 1) {
    const int which = wcol0 / g.H;   // wave-uniform: 0 q, 1 k, 2 v
    const int M32 = (int)g.M, r0 = (int)wrow0;
    if (which == 2) {
      big_mainloop<false, DBG>(acc, smem, srcA, srcW, ldsA, ldsW, nk, a_off, b_off);
      // acc[i][j][r] = D[m = 16i + 4fg + r][n = 16j + fr]: 4 consecutive tokens per lane -> V^T rows
      bf16* dst = reinterpret_cast<bf16*>(g.vt);
      float bv[4];
      int64_t coloff[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = wcol0 + 16 * j + fr;
        const int cc = col < g.N ? col : g.N - 1;
        bv[j] = g.bias[cc];
        const int c = cc - 2 * g.H, head = c / g.dh, d = c % g.dh;
        coloff[j] = col < g.N ? ((int64_t)head * g.dh + d) * g.L : -1;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = r0 + 16 * i + 4 * fg;
        if (row < M32) {
          const int b = row / g.L, l = row - b * g.L;
          bf16* base = dst + (int64_t)b * g.H * g.L + l;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (coloff[j] >= 0) {
              bf16x4 v;
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = (bf16)(acc[i][j][r] + bv[j]);
              *reinterpret_cast<bf16x4*>(base + coloff[j]) = v;
            }
          }
        }
      }
    } else {
      big_mainloop<true, DBG>(acc, smem, srcA, srcW, ldsA, ldsW, nk, a_off, b_off);
      // acc[i][j][r] = D[n = 16j + 4fg + r][m = 16i + fr]: 4 consecutive head dims per lane
      bf16* dst = reinterpret_cast<bf16*>(which == 0 ? g.q : g.k);
      f32x4 bv[4];
      int64_t coloff[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = wcol0 + 16 * j + 4 * fg;
        const int cc = col < g.N ? col : g.N - 4;
        bv[j] = *reinterpret_cast<const f32x4*>(g.bias + cc);
        const int c = cc - which * g.H, head = c / g.dh, d = c % g.dh;
        coloff[j] = col < g.N ? (int64_t)head * g.L * g.dh + d : -1;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = r0 + 16 * i + fr;
        if (row < M32) {
          const int b = row / g.L, l = row - b * g.L;
          bf16* base = dst + ((int64_t)b * g.nh * g.L + l) * g.dh;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (coloff[j] >= 0) {
              bf16x4 v;
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = (bf16)(acc[i][j][r] + bv[j][r]);
              *reinterpret_cast<bf16x4*>(base + coloff[j]) = v;
            }
          }
        }
      }
    }
  } else {
    big_mainloop<true, DBG>(acc, smem, srcA, srcW, ldsA, ldsW, nk, a_off, b_off);
    bf16* outT = reinterpret_cast<bf16*>(g.out);
    float* outF = reinterpret_cast<float*>(g.out);
    const bf16* res = reinterpret_cast<const bf16*>(g.residual);
    if constexpr ((DBG & 4) != 0) {
      float sacc = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (sacc == 12345.678f) outF[0] = sacc;
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = wcol0 + 16 * j + 4 * fg;
      if (col < g.N) {   // N % 4 == 0 (checked by the launcher): a lane's 4 columns are all valid
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) bv = *reinterpret_cast<const f32x4*>(g.bias + col);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int64_t row = wrow0 + 16 * i + fr;
          if (row < g.M) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bv[r];
            if constexpr (ACT != MH_ACT_NONE) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = apply_act<bf16>(v[r], ACT);
            }
            if (res) {
              const bf16x4 rv = *reinterpret_cast<const bf16x4*>(res + row * g.ldr + col);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
            }
            if (g.out_f32) {
              *reinterpret_cast<f32x4*>(outF + row * g.ldo + col) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
              bf16x4 o;
#pragma unroll
              for (int r = 0; r < 4; ++r) o[r] = (bf16)v[r];
              *reinterpret_cast<bf16x4*>(outT + row * g.ldo + col) = o;
            }
          }
        }
      }
    }
  }
}

int g_dbg = 0;
int g_variant = 2;  // bf16 kernel choice: 0 small-tile register-staged, 1 small-tile global_load_lds, 2 big tile  // bf16 staging mode, switchable for A/B runs (mh_gemm_set_glds)

template <int EPI>
int launch(const GemmArgs& g, int dtype, hipStream_t s) {
  const int64_t tiles = (int64_t)ceil_div(g.M, BM) * ceil_div(g.N, BN);
  MH_CHECK_ARG(tiles > 0 && tiles < (1ll << 31), "gemm: bad grid (M=%lld N=%d)", (long long)g.M, g.N);
  dim3 grid((unsigned)tiles), block(256);
  if (dtype == MH_BF16) {
    MH_CHECK_ARG(g.K % 64 == 0 && g.K > 0, "gemm(bf16): K=%d must be a positive multiple of 64", g.K);
    MH_CHECK_ARG(g.lda % 8 == 0 && g.ldw % 8 == 0, "gemm(bf16): lda/ldw must be multiples of 8");
    const bool big_ok = g.N % 4 == 0 && g.K % B2K == 0 && g.ldo % 4 == 0 && g.ldr % 4 == 0;
    if (g_variant == 2 && big_ok) {
      const int64_t t2 = (int64_t)ceil_div(g.M, B2M) * ceil_div(g.N, B2N);
      const dim3 grid2((unsigned)t2);
      if constexpr (EPI == 1) {
        MH_LAUNCH((gemm_big_kernel<1, MH_ACT_NONE>), grid2, block, 0, s, g);
      } else {
        if (g.dbg) {   // timing-only ablations (tools/gemm_bench.py)
          switch (g.dbg & 7) {
            case 1: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_NONE, 1>), grid2, block, 0, s, g); break;
            case 2: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_NONE, 2>), grid2, block, 0, s, g); break;
            case 3: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_NONE, 3>), grid2, block, 0, s, g); break;
            case 4: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_NONE, 4>), grid2, block, 0, s, g); break;
            case 5: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_NONE, 5>), grid2, block, 0, s, g); break;
            case 6: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_NONE, 6>), grid2, block, 0, s, g); break;
            default: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_NONE, 7>), grid2, block, 0, s, g); break;
          }
        } else switch (g.act) {
          case MH_ACT_TANH: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_TANH>), grid2, block, 0, s, g); break;
          case MH_ACT_GELU_ERF: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_GELU_ERF>), grid2, block, 0, s, g); break;
          case MH_ACT_SILU: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_SILU>), grid2, block, 0, s, g); break;
          default: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_NONE>), grid2, block, 0, s, g); break;
        }
      }
    } else if (g_variant == 1) {
      MH_LAUNCH((gemm_kernel<bf16, EPI, 1>), grid, block, 0, s, g);
    } else {
      MH_LAUNCH((gemm_kernel<bf16, EPI, 0>), grid, block, 0, s, g);
    }
  } else if (dtype == MH_F32) {
    MH_CHECK_ARG(g.K % 16 == 0 && g.K > 0, "gemm(f32): K=%d must be a positive multiple of 16", g.K);
    MH_CHECK_ARG(g.lda % 4 == 0 && g.ldw % 4 == 0, "gemm(f32): lda/ldw must be multiples of 4");
    MH_LAUNCH((gemm_kernel<float, EPI, 0>), grid, block, 0, s, g);
  } else {
    MH_CHECK_ARG(false, "gemm: unknown dtype %d", dtype);
  }
  MH_CHECK_LAUNCH();
  return MH_OK;
}

}  // namespace

extern "C" int mh_gemm_set_debug(int bits) {
  g_dbg = bits;
  return MH_OK;
}

extern "C" int mh_gemm_set_variant(int variant) {
  MH_CHECK_ARG(variant >= 0 && variant <= 2, "gemm_set_variant: variant must be 0, 1 or 2");
  g_variant = variant;
  return MH_OK;
}

extern "C" int mh_gemm_bias_act(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                                const void* residual, int64_t ldr, void* out, int64_t ldo, int out_f32,
                                int64_t M, int N, int K, int act, int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(A && W && out, "gemm: null pointer");
  MH_CHECK_ARG(M > 0 && N > 0, "gemm: empty problem M=%lld N=%d", (long long)M, N);
  MH_CHECK_ARG(act >= MH_ACT_NONE && act <= MH_ACT_SILU, "gemm: unknown activation %d", act);
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias;
  g.residual = residual; g.ldr = residual ? ldr : 8; g.out = out; g.ldo = ldo; g.out_f32 = out_f32;
  g.M = M; g.N = N; g.K = K; g.act = act; g.dbg = g_dbg;
  return launch<0>(g, dtype, (hipStream_t)stream);
}

extern "C" int mh_gemm_qkv(const void* A, int64_t lda, const void* Wqkv, int64_t ldw, const float* bqkv, void* q,
                           void* k, void* vt, int B, int L, int H, int nh, int dtype, mh_stream_t stream) {
  MH_CHECK_ARG(A && Wqkv && bqkv && q && k && vt, "gemm_qkv: null pointer");
  MH_CHECK_ARG(H % 64 == 0, "gemm_qkv: hidden size %d must be a multiple of 64", H);
  MH_CHECK_ARG(nh > 0 && H % nh == 0 && (H / nh) % 8 == 0, "gemm_qkv: head dim must be a multiple of 8");
  MH_CHECK_ARG(L % 8 == 0, "gemm_qkv: seq_len %d must be a multiple of 8", L);
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = Wqkv; g.ldw = ldw; g.bias = bqkv; g.ldr = 8; g.ldo = 8;
  g.M = (int64_t)B * L; g.N = 3 * H; g.K = H;
  g.q = q; g.k = k; g.vt = vt; g.L = L; g.H = H; g.nh = nh; g.dh = H / nh;
  return launch<1>(g, dtype, (hipStream_t)stream);
}
