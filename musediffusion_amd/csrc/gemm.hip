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
  // EPI 2 (nearest-embedding scores): aux[col] = |W_col|^2, rown[row] = |x_row|^2, partial best per (row, slot)
  const float* aux; const float* rown; float* pbest; int32_t* pidx; int nslots;
  int a_panel, w_panel, o_panel, r_panel;  // operand stored as K32 panels: [cols/32][ld rows][32]
  int64_t sA, sW, sO, sR;  // batch strides in elements (grid.y = batch index)
  int stagger;
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

template <bool SWAP, int DBG, bool PP, bool INTERLEAVE = false>
__device__ __forceinline__ void big_mainloop(f32x4 (&acc)[8][4], const char* smem, const char* const (&srcA)[4],
                                             const char* const (&srcW)[2], const int (&ldsA)[4], const int (&ldsW)[2],
                                             int nk, int a_off, const int (&b_offs)[4], int64_t kstepA, int64_t kstepW, int group) {
  auto issue_one = [&](int kt, int idx) {   // idx 0..3: A pieces, 4..5: W pieces (one 1-KiB DMA each)
    if constexpr ((DBG & 1) != 0) return;
    char* base = const_cast<char*>(smem) + (kt % B2STAGES) * B2_STAGE_BYTES;
    if (idx < 4)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[idx] + kt * kstepA),
                                       (__attribute__((address_space(3))) void*)(base + ldsA[idx]), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW[idx - 4] + kt * kstepW),
                                       (__attribute__((address_space(3))) void*)(base + B2M * 64 + ldsW[idx - 4]), 16, 0, 0);
  };
  auto issue = [&](int kt) {
#pragma unroll
    for (int idx = 0; idx < 6; ++idx) issue_one(kt, idx);
  };
  // PP (ping-pong): two 4-wave groups share one workgroup, each with its own tile and LDS ring.  Every
  // K-step has TWO workgroup barriers (after the DMA wait, after the fragment reads) and group 1 runs one
  // barrier behind group 0, so on every SIMD one wave is in its MFMA segment while its partner is in its
  // load segment (DMA issue + ds_reads): the matrix pipe never waits for LDS / DMA latency.
  if constexpr (PP) { if (group == 1) __builtin_amdgcn_s_barrier(); }
  issue(0);
  if (nk > 1) issue(1);
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt must have landed; stage kt+1 (6 DMA ops of this wave) may stay in flight
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // everyone's share of stage kt is in LDS; stage kt-1 is fully consumed
    const bool refill = kt + 2 < nk;   // stage kt+2 goes into the slot stage kt-1 occupied
    if constexpr (!INTERLEAVE) { if (refill) issue(kt + 2); }
    const char* As = smem + (kt % B2STAGES) * B2_STAGE_BYTES;
    const char* Ws = As + B2M * 64;
    // all 12 fragment reads are issued up front (W first: every MFMA row needs all four of them), then the
    // scheduler is fenced so the MFMAs drain them behind counted lgkmcnt waits instead of four full stalls
    bf16x8 a[8], b[4];
    if constexpr ((DBG & 8) != 0) {   // ablation: no LDS fragment reads (operands are whatever the registers hold)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" : "=v"(b[j]));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "=v"(a[i]));
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(Ws + b_offs[j]);
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const bf16x8*>(As + a_off + i * (16 * 64));
    }
    if constexpr (PP) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr ((DBG & 2) != 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(a[i]));
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(b[j]));
      continue;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      if constexpr (INTERLEAVE) {
        // one DMA piece in the shadow of every 4 MFMAs: its ~100-cycle issue cost hides behind the matrix pipe
        if (i >= 1 && i <= 6) {
          __builtin_amdgcn_sched_barrier(0);
          if (refill) issue_one(kt + 2, i - 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  if constexpr (PP) { if (group == 0) __builtin_amdgcn_s_barrier(); }
}

template <int EPI, int ACT, int DBG = 0, bool PP = false>
__global__ __launch_bounds__(PP ? 512 : 256, PP ? 1 : 2) void gemm_big_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) char smem_all[(PP ? 2 : 1) * B2STAGES * B2_STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int group = PP ? (wave_all >> 2) : 0, wave = wave_all & 3;
  const char* smem = smem_all + group * (B2STAGES * B2_STAGE_BYTES);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (g.N + B2N - 1) / B2N;
  const int tiles_total = tiles_n * (int)((g.M + B2M - 1) / B2M);
  int bid;
  bool tile_valid = true;
  if constexpr (PP) {
    bid = 2 * xcd_remap(blockIdx.x, gridDim.x) + group;   // the two groups take adjacent column tiles: same A panel
    if (bid >= tiles_total) { bid = tiles_total - 1; tile_valid = false; }
  } else {
    bid = xcd_remap(blockIdx.x, gridDim.x);
  }
  const int64_t m0 = (int64_t)(bid / tiles_n) * B2M;
  const int n0 = (bid % tiles_n) * B2N;
  const int nk = g.K / B2K;
  const int fr = lane & 15, fg = lane >> 4;
  constexpr int GSW[4] = {0, 2, 3, 1};
  if (g.stagger > 0 && (int)blockIdx.x < 512 && ((g.stagger & 1) ? (((int)blockIdx.x >> 3) & 1) : ((int)blockIdx.x >= 256))) {
    // experiment: phase-shift the second resident block of every CU so that its epilogue (VALU + stores)
    // overlaps the first block's main loop (MFMA + DMA) instead of running in lockstep with it
    const uint64_t t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (uint64_t)g.stagger) __builtin_amdgcn_s_sleep(8);
  }

  // DMA coordinates: one instruction covers 16 rows x 64 B; lane i lands at row i/4, physical chunk i%4
  const char* srcA[4];
  const char* srcW[2];
  int ldsA[4], ldsW[2];
  // row-major operand: rows lda elements apart, a K-step advances 32 elements; K32-panel operand
  // ([K/32][ld rows][32]): rows 32 elements apart, a K-step advances one whole panel (ld * 32)
  const int64_t a_row = g.a_panel ? 32 : g.lda, w_row = g.w_panel ? 32 : g.ldw;
  const int64_t kstepA = g.a_panel ? g.lda * 64 : 64, kstepW = g.w_panel ? g.ldw * 64 : 64;
  {
    const int rl = lane >> 2, pc = lane & 3;
    const int lc = pc ^ GSW[(rl >> 2) & 3];          // logical chunk stored at this physical slot
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r16 = (wave * 4 + j) * 16;
      int64_t ra = m0 + r16 + rl; if (ra >= g.M) ra = g.M - 1;
      srcA[j] = reinterpret_cast<const char*>(g.A) + ((int64_t)blockIdx.y * g.sA + ra * a_row + lc * 8) * 2;
      ldsA[j] = r16 * 64;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r16 = (wave * 2 + j) * 16;
      int rw = n0 + r16 + rl; if (rw >= g.N) rw = g.N - 1;
      srcW[j] = reinterpret_cast<const char*>(g.W) + ((int64_t)blockIdx.y * g.sW + (int64_t)rw * w_row + lc * 8) * 2;
      ldsW[j] = r16 * 64;
    }
  }
  const int frag_off = fr * 64 + ((fg ^ GSW[(fr >> 2) & 3]) << 4);
  const int a_off = wm * (128 * 64) + frag_off;
  // W fragment rows.  Un-swapped MFMA (V^T waves): tile j takes rows 16j + fr.  Swapped MFMA: tile j, input
  // row p = fr takes W row 32(j>>1) + 8(p>>2) + 4(j&1) + (p&3), so that a lane's accumulators (output rows
  // 4fg + r of tiles 2h, 2h+1) are the 8 CONSECUTIVE output columns 32h + 8fg .. +7: 16-byte epilogue
  // accesses, four lanes covering 64 contiguous bytes of a row.  Still conflict-free under the same swizzle.
  int b_offs[4];
  const bool v_wave = (EPI == 1) && ((n0 + wn * 64) / g.H == 2);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = v_wave ? 16 * j + fr : 32 * (j >> 1) + 8 * (fr >> 2) + 4 * (j & 1) + (fr & 3);
    b_offs[j] = wn * (64 * 64) + row * 64 + ((fg ^ GSW[(row >> 2) & 3]) << 4);
  }

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
      big_mainloop<false, DBG, PP>(acc, smem, srcA, srcW, ldsA, ldsW, nk, a_off, b_offs, kstepA, kstepW, group);
      if (!tile_valid) return;
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
      big_mainloop<true, DBG, PP>(acc, smem, srcA, srcW, ldsA, ldsW, nk, a_off, b_offs, kstepA, kstepW, group);
      if (!tile_valid) return;
      // acc[i][2h + q][r] = D[n = 32h + 8fg + 4q + r][m = 16i + fr]: 8 consecutive head dims per lane
      bf16* dst = reinterpret_cast<bf16*>(which == 0 ? g.q : g.k);
      float bv[2][8];
      int64_t coloff[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int col = wcol0 + 32 * h + 8 * fg;
        const int cc = col < g.N ? col : g.N - 8;
        load8(g.bias + cc, bv[h]);
        const int c = cc - which * g.H, head = c / g.dh, d = c % g.dh;
        coloff[h] = col < g.N ? (int64_t)head * g.L * g.dh + d : -1;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = r0 + 16 * i + fr;
        if (row < M32) {
          const int b = row / g.L, l = row - b * g.L;
          bf16* base = dst + ((int64_t)b * g.nh * g.L + l) * g.dh;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (coloff[h] >= 0) {
              float v[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = acc[i][2 * h + (e >> 2)][e & 3] + bv[h][e];
              store8(base + coloff[h], v);
            }
          }
        }
      }
    }
  } else {
    big_mainloop<true, DBG, PP>(acc, smem, srcA, srcW, ldsA, ldsW, nk, a_off, b_offs, kstepA, kstepW, group);
    if (!tile_valid) return;
    bf16* outT = reinterpret_cast<bf16*>(g.out) + (int64_t)blockIdx.y * g.sO;
    float* outF = reinterpret_cast<float*>(g.out) + (int64_t)blockIdx.y * g.sO;
    const bf16* res = g.residual ? reinterpret_cast<const bf16*>(g.residual) + (int64_t)blockIdx.y * g.sR : nullptr;
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
    for (int h = 0; h < 2; ++h) {
      const int col = wcol0 + 32 * h + 8 * fg;
      if (col < g.N) {   // N % 8 == 0 (checked by the launcher): a lane's 8 columns are all valid
        float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (g.bias) load8(g.bias + col, bv);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int64_t row = wrow0 + 16 * i + fr;
          if (row < g.M) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[i][2 * h + (e >> 2)][e & 3] + bv[e];
            if constexpr (ACT != MH_ACT_NONE) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = apply_act<bf16>(v[e], ACT);
            }
            if (res) {
              const int64_t ro = g.r_panel ? ((int64_t)(col >> 5) * g.ldr + row) * 32 + (col & 31) : row * g.ldr + col;
              float rv[8];
              load8(res + ro, rv);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += rv[e];
            }
            if (g.out_f32) {
              store8(outF + row * g.ldo + col, v);
            } else {
              const int64_t oo = g.o_panel ? ((int64_t)(col >> 5) * g.ldo + row) * 32 + (col & 31) : row * g.ldo + col;
              store8(outT + oo, v);
            }
          }
        }
      }
    }
  }
}

int g_dbg = 0;
int g_stagger = 0;
int g_variant = 2;  // bf16 kernel choice: 0 small-tile register-staged, 1 small-tile global_load_lds, 2 big tile  // bf16 staging mode, switchable for A/B runs (mh_gemm_set_glds)

template <int EPI>
int launch(const GemmArgs& g, int dtype, hipStream_t s, int batch = 1) {
  const int64_t tiles = (int64_t)ceil_div(g.M, BM) * ceil_div(g.N, BN);
  MH_CHECK_ARG(tiles > 0 && tiles < (1ll << 31), "gemm: bad grid (M=%lld N=%d)", (long long)g.M, g.N);
  MH_CHECK_ARG(batch >= 1 && batch <= 65535, "gemm: batch %d out of range", batch);
  dim3 grid((unsigned)tiles, (unsigned)batch), block(256);
  if (dtype == MH_BF16) {
    MH_CHECK_ARG((g.a_panel || g.lda % 8 == 0) && (g.w_panel || g.ldw % 8 == 0), "gemm(bf16): lda/ldw must be multiples of 8");
    const bool any_panel = g.a_panel || g.w_panel || g.o_panel || g.r_panel;
    const bool big_ok = g.N % 8 == 0 && g.K % B2K == 0 && (g.o_panel || g.ldo % 8 == 0 || (g.out_f32 && g.ldo % 4 == 0)) &&
                        (g.r_panel || g.ldr % 8 == 0);
    MH_CHECK_ARG(!any_panel || (big_ok && g_variant >= 2), "gemm: panel layouts need the big-tile bf16 kernel");
    MH_CHECK_ARG(g.K > 0 && (g.K % 64 == 0 || (g.K % B2K == 0 && big_ok && g_variant >= 2)),
                 "gemm(bf16): K=%d must be a positive multiple of 64 (32 with the big-tile kernel)", g.K);
    if (g_variant == 3 && big_ok && !g.dbg) {
      const int64_t t2 = (int64_t)ceil_div(g.M, B2M) * ceil_div(g.N, B2N);
      const dim3 grid3((unsigned)((t2 + 1) / 2), (unsigned)batch), block3(512);
      if constexpr (EPI == 1) {
        MH_LAUNCH((gemm_big_kernel<1, MH_ACT_NONE, 0, true>), grid3, block3, 0, s, g);
      } else {
        switch (g.act) {
          case MH_ACT_TANH: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_TANH, 0, true>), grid3, block3, 0, s, g); break;
          case MH_ACT_GELU_ERF: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_GELU_ERF, 0, true>), grid3, block3, 0, s, g); break;
          case MH_ACT_SILU: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_SILU, 0, true>), grid3, block3, 0, s, g); break;
          default: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_NONE, 0, true>), grid3, block3, 0, s, g); break;
        }
      }
    } else if (g_variant == 3 && big_ok) {
      const int64_t t2 = (int64_t)ceil_div(g.M, B2M) * ceil_div(g.N, B2N);
      const dim3 grid3((unsigned)((t2 + 1) / 2), (unsigned)batch), block3(512);
      if ((g.dbg & 15) == 4) MH_LAUNCH((gemm_big_kernel<0, MH_ACT_NONE, 4, true>), grid3, block3, 0, s, g);
      else MH_LAUNCH((gemm_big_kernel<0, MH_ACT_NONE, 5, true>), grid3, block3, 0, s, g);
    } else if (g_variant >= 2 && big_ok) {
      const int64_t t2 = (int64_t)ceil_div(g.M, B2M) * ceil_div(g.N, B2N);
      const dim3 grid2((unsigned)t2, (unsigned)batch);
      if constexpr (EPI == 1) {
        MH_LAUNCH((gemm_big_kernel<1, MH_ACT_NONE>), grid2, block, 0, s, g);
      } else {
        if (g.dbg) {   // timing-only ablations (tools/gemm_bench.py)
          switch (g.dbg & 15) {
            case 1: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_NONE, 1>), grid2, block, 0, s, g); break;
            case 2: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_NONE, 2>), grid2, block, 0, s, g); break;
            case 3: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_NONE, 3>), grid2, block, 0, s, g); break;
            case 4: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_NONE, 4>), grid2, block, 0, s, g); break;
            case 5: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_NONE, 5>), grid2, block, 0, s, g); break;
            case 6: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_NONE, 6>), grid2, block, 0, s, g); break;
            case 7: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_NONE, 7>), grid2, block, 0, s, g); break;
            case 12: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_NONE, 12>), grid2, block, 0, s, g); break;
            case 13: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_NONE, 13>), grid2, block, 0, s, g); break;
            default: MH_LAUNCH((gemm_big_kernel<0, MH_ACT_NONE, 14>), grid2, block, 0, s, g); break;
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
    MH_CHECK_ARG(!(g.a_panel || g.w_panel || g.o_panel || g.r_panel), "gemm(f32): panel layouts are bf16 only");
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
  g_dbg = bits & 15;
  g_stagger = bits >> 4;   // shader cycles
  return MH_OK;
}

extern "C" int mh_gemm_set_variant(int variant) {
  MH_CHECK_ARG(variant >= 0 && variant <= 3, "gemm_set_variant: variant must be 0..3");
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
  g.M = M; g.N = N; g.K = K; g.act = act; g.dbg = g_dbg & 15; g.stagger = g_stagger;
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
  g.M = M; g.N = N; g.K = K; g.act = act; g.dbg = g_dbg & 15; g.stagger = g_stagger;
  g.a_panel = a_panel; g.w_panel = w_panel; g.o_panel = o_panel; g.r_panel = residual ? r_panel : 0;
  return launch<0>(g, dtype, (hipStream_t)stream);
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

extern "C" int mh_gemm_qkv_ex(const void* A, int64_t lda, int a_panel, const void* Wqkv, int64_t ldw, int w_panel,
                              const float* bqkv, void* q, void* k, void* vt, int B, int L, int H, int nh, int dtype,
                              mh_stream_t stream) {
  MH_CHECK_ARG(A && Wqkv && bqkv && q && k && vt, "gemm_qkv: null pointer");
  MH_CHECK_ARG(H % 64 == 0, "gemm_qkv: hidden size %d must be a multiple of 64", H);
  MH_CHECK_ARG(nh > 0 && H % nh == 0 && (H / nh) % 8 == 0, "gemm_qkv: head dim must be a multiple of 8");
  MH_CHECK_ARG(L % 8 == 0, "gemm_qkv: seq_len %d must be a multiple of 8", L);
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = Wqkv; g.ldw = ldw; g.bias = bqkv; g.ldr = 8; g.ldo = 8;
  g.M = (int64_t)B * L; g.N = 3 * H; g.K = H; g.stagger = g_stagger;
  g.a_panel = a_panel; g.w_panel = w_panel;
  g.q = q; g.k = k; g.vt = vt; g.L = L; g.H = H; g.nh = nh; g.dh = H / nh;
  return launch<1>(g, dtype, (hipStream_t)stream);
}


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
