// Micro-benchmark for the FFN1 question: can the GELU epilogue of tile t hide in the issue slots of tile t+1's MFMAs when a wave
// has its SIMD to itself (512 registers: two accumulator sets)?  One "tile" = 16 K-steps of a 128x64 wave tile on mfma_16x16x32
// (32 MFMAs + 12 ds_read_b128 [+ 6 LDS-DMA pieces] per K-step) + 16 epilogue groups (8 x (bias + GELU) + 4 cvt_pk + one 16-byte store).
//   mode 0: MFMA K-steps only        mode 1: epilogue groups only
//   mode 2: 16 K-steps, then 16 groups (today's structure)         mode 3: one epilogue group interleaved into every K-step
// threads 256 = one wave per SIMD, 512 = two (today's two co-resident blocks).  DMA = 1: the K-steps also issue their 6 pieces.
// Build: hipcc --offload-arch=gfx950 -O3 ffn1_overlap.hip -o ffn1_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ float gelu_fast(float x) {
  constexpr float kL2E = 1.4426950408889634f;
  const float u = fminf(x * x, 64.0f);
  float p = fmaf(0.0007030335771326705f * kL2E, u, -0.07401129204508086f * kL2E);
  p = fmaf(p, u, -1.5950157685701116f * kL2E);
  const float e = __builtin_amdgcn_exp2f(x * p);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// the same wave tile on mfma_32x32x16 (16 per K-step, 16 ds_read_b64... here: 6 ds_read_b128 halves): does the larger shape leave
// the issue port to the partner wave's vector instructions?  modes 0 / 1 / 2 / 4 as below
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k32(const bf16* __restrict__ src, bf16* __restrict__ dst, const float* __restrict__ bias, int tiles,
                                               unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* mine = lds + (wave & 3) * 24576;
  for (int i = tid; i < 4 * 24576 / 16; i += blockDim.x) reinterpret_cast<f32x4*>(lds)[i] = reinterpret_cast<const f32x4*>(src)[(i + blockIdx.x * 97) & 65535];
  __syncthreads();
  constexpr bool TWO = MODE == 3 || MODE == 5;      // a second accumulator set: the previous tile's epilogue rides in this tile's K-loop (5: without its stores)
  f32x16 acc[4][2], prev[TWO ? 4 : 1][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.1f * i + 0.01f * r + 0.05f * lane; if (TWO) prev[TWO ? i : 0][j][r] = acc[i][j][r]; }
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = bias[(lane * 8 + e) & 1023];
  bf16* gdst = dst + ((size_t)blockIdx.x * blockDim.x + tid) * 8;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < tiles; ++t) {
    auto kstep = [&](int kt) {
      const char* st = mine + (kt % 3) * 8192;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {     // two k = 16 halves of a 32-deep K-step
        bf16x8 a[4], b[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(st + ((kk * 2048 + j * 1024 + lane * 16) & 8191));
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(st + ((4096 + kk * 2048 + i * 512 + lane * 16) & 8191));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
      }
    };
    auto egroup = [&](int g) {
      const int i = g >> 2, j = (g >> 1) & 1, h = g & 1;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gelu_fast((TWO ? prev[TWO ? i : 0][j][8 * h + e] : acc[i][j][8 * h + e]) + bv[e]);
      bf16x8 r;
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = (bf16)v[e];
      f32x4 raw;
      __builtin_memcpy(&raw, &r, 16);
      if (MODE == 5) { asm volatile("" :: "v"(raw)); if (raw[0] == 12345.678f) __builtin_nontemporal_store(raw, reinterpret_cast<f32x4*>(gdst)); }
      else __builtin_nontemporal_store(raw, reinterpret_cast<f32x4*>(gdst + ((size_t)g + 16 * (size_t)(t & 3)) * 8 * 131072));
    };
    auto zero = [&]() {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    if ((MODE == 4 || MODE == 7) && wave >= 4) {
#pragma unroll
      for (int g = 0; g < 16; ++g) egroup(g);
      zero();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 7) __builtin_amdgcn_s_setprio(2);      // the matrix phase outranks the partner's vector phase: its MFMAs issue on time, the partner fills the gaps
    if (MODE == 0 || MODE == 2 || MODE == 4 || MODE == 7) {
#pragma unroll
      for (int kt = 0; kt < 16; ++kt) { kstep(kt); __builtin_amdgcn_sched_barrier(0); }
    }
    if (MODE == 7) __builtin_amdgcn_s_setprio(0);
    if (MODE == 1 || MODE == 2 || ((MODE == 4 || MODE == 7) && wave < 4)) {
#pragma unroll
      for (int g = 0; g < 16; ++g) egroup(g);
      if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(acc[i][0]), "+v"(acc[i][1]));
      } else zero();
    }
    if (MODE == 3 || MODE == 5) {
#pragma unroll
      for (int kt = 0; kt < 16; ++kt) {
        kstep(kt);
        egroup(kt);
        // one epilogue group (~90 vector instructions, 16 of them transcendental) spread over the K-step's 16 MFMAs: the 24 issue cycles
        // a 32x32x16 leaves free take 5 - 6 of them
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);   // 6 VALU
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { prev[TWO ? i : 0][j] = acc[i][j]; }
      zero();
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0][0] + acc[i][1][3] + (TWO ? prev[TWO ? i : 0][0][5] : 0.f);
  if (s == 12345.678f) dst[0] = (bf16)s;
  if (tid == 0) cyc[blockIdx.x] = (t1 - t0) / tiles;
}

// K-steps only on a 128x128 wave tile (64 mfma_16x16x32 + 16 ds_read_b128 per K-step, 256 accumulator registers: one wave per SIMD
// with the 512-register budget): does the larger tile keep the matrix pipe as busy as two waves of 128x64 do, with 2/3 of the LDS bytes?
template <int THREADS>
__global__ __launch_bounds__(THREADS) void kbig(const bf16* __restrict__ src, bf16* __restrict__ dst, int tiles) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* mine = lds + (wave & 3) * 24576;
  for (int i = tid; i < 4 * 24576 / 16; i += blockDim.x) reinterpret_cast<f32x4*>(lds)[i] = reinterpret_cast<const f32x4*>(src)[(i + blockIdx.x * 97) & 65535];
  __syncthreads();
  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.1f * i, 0.2f * j, -0.3f, 0.05f * lane};
  for (int t = 0; t < tiles; ++t) {
#pragma unroll 4
    for (int kt = 0; kt < 16; ++kt) {
      bf16x8 a[8], b[8];
      const char* st = mine + (kt % 3) * 8192;
#pragma unroll
      for (int j = 0; j < 8; ++j) b[j] = *reinterpret_cast<const bf16x8*>(st + ((j * 512 + lane * 16) & 8191));
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const bf16x8*>(st + ((4096 + i * 512 + lane * 16) & 8191));
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
  if (s == 12345.678f) dst[0] = (bf16)s;
}

template <int MODE, int DMA, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const bf16* __restrict__ src, bf16* __restrict__ dst, const float* __restrict__ bias, int tiles,
                                         unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* mine = lds + (wave & 3) * 24576 + (wave >> 2) * 3 * 24576 * 0;   // the waves of a SIMD share the staging area (timing only)
  // fill the staging area once so the fragments are not zeros
  for (int i = tid; i < 4 * 24576 / 16; i += blockDim.x) reinterpret_cast<f32x4*>(lds)[i] = reinterpret_cast<const f32x4*>(src)[(i + blockIdx.x * 97) & 65535];
  __syncthreads();
  constexpr bool TWO = MODE == 3;      // a second accumulator set only where the epilogue of the previous tile rides in the K-loop
  f32x4 acc[8][4], prev[TWO ? 8 : 1][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[i][j] = f32x4{0.1f * i, 0.2f * j, -0.3f, 0.05f * lane}; if (TWO) prev[i][j] = acc[i][j]; }
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = bias[(lane * 8 + e) & 1023];
  const bf16* gsrc = src + ((size_t)blockIdx.x * 64 + wave) * 8192 + lane * 8;
  bf16* gdst = dst + ((size_t)blockIdx.x * blockDim.x + tid) * 8;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < tiles; ++t) {
    auto kstep = [&](int kt) {
      bf16x8 a[8], b[4];
      const char* st = mine + (kt % 3) * 8192;
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(st + ((j * 1024 + lane * 16) & 8191));
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const bf16x8*>(st + ((4096 + i * 512 + lane * 16) & 8191));
      if (DMA) {
#pragma unroll
        for (int p = 0; p < 6; ++p)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + (size_t)((kt * 6 + p) & 127) * 512),
                                           (__attribute__((address_space(3))) void*)(mine + 16384 + (p & 7) * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
      if (DMA) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    };
    auto egroup = [&](int g) {   // group g = (row tile i = g >> 1, column half qh = g & 1): 8 values of the PREVIOUS tile's accumulators
      const int i = g >> 1, qh = g & 1;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gelu_fast((TWO ? prev[TWO ? i : 0][2 * qh + (e >> 2)][e & 3] : acc[i][2 * qh + (e >> 2)][e & 3]) + bv[e]);
      bf16x8 r;
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = (bf16)v[e];
      f32x4 raw;
      __builtin_memcpy(&raw, &r, 16);
      __builtin_nontemporal_store(raw, reinterpret_cast<f32x4*>(gdst + ((size_t)g + 16 * (size_t)(t & 3)) * 8 * 131072));   // (a different row block per tile)
    };
    if (MODE == 4 && wave >= 4) {   // the second wave of every SIMD runs half a tile out of phase: its epilogue first
#pragma unroll
      for (int g = 0; g < 16; ++g) egroup(g);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 0 || MODE == 2 || MODE == 4) {
#pragma unroll
      for (int kt = 0; kt < 16; ++kt) { kstep(kt); __builtin_amdgcn_sched_barrier(0); }
    }
    if (MODE == 5) {   // the same 512 MFMAs with the fragments of ONE K-step kept in registers: what the LDS reads cost (time and clock)
      bf16x8 a[8], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(mine + ((j * 1024 + lane * 16) & 8191));
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const bf16x8*>(mine + ((4096 + i * 512 + lane * 16) & 8191));
#pragma unroll
      for (int kt = 0; kt < 16; ++kt) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (MODE == 4 && wave < 4) {
#pragma unroll
      for (int g = 0; g < 16; ++g) egroup(g);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int g = 0; g < 16; ++g) egroup(g);
      if (MODE == 1) {      // (keep the inputs loop-variant)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(acc[i][0]), "+v"(acc[i][1]), "+v"(acc[i][2]), "+v"(acc[i][3]));
      }
    }
    if (MODE == 3) {
#pragma unroll
      for (int kt = 0; kt < 16; ++kt) {
        kstep(kt);
        egroup(kt);
        // spread the group's ~100 vector instructions between the 32 MFMAs: 1 MFMA, then 3 VALU / transcendental slots
#pragma unroll
        for (int m = 0; m < 32; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // 3 VALU
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the finished tile becomes the previous tile (register renaming, no moves in a real kernel: here a cheap dependency)
    if (TWO) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { prev[i][j] = acc[i][j]; acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + (TWO ? prev[TWO ? i : 0][j][1] : 0.f);
  if (s == 12345.678f) dst[0] = (bf16)s;
  if (tid == 0) cyc[blockIdx.x] = (t1 - t0) / tiles;
}

template <int MODE, int DMA, int THREADS>
void run(const char* name, const bf16* src, bf16* dst, const float* bias, unsigned long long* cyc) {
  const int tiles = 200;
  const int threads = THREADS;
  auto kern = k<MODE, DMA, THREADS>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 24576);
  kern<<<256, threads, 4 * 24576>>>(src, dst, bias, tiles, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  kern<<<256, threads, 4 * 24576>>>(src, dst, bias, tiles, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const int wps = threads / 256;
  printf("%-44s dma %d  %d wave(s)/SIMD: %7llu cycles per tile per wave, %7.3f us per tile-round, %7.3f us per tile per SIMD\n", name, DMA, wps, c,
         ms * 1e3 / tiles, ms * 1e3 / tiles / wps);
}

template <int MODE, int THREADS>
void run32(const char* name, const bf16* src, bf16* dst, const float* bias, unsigned long long* cyc) {
  const int tiles = 200, threads = THREADS;
  auto kern = k32<MODE, THREADS>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 24576);
  kern<<<256, threads, 4 * 24576>>>(src, dst, bias, tiles, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  kern<<<256, threads, 4 * 24576>>>(src, dst, bias, tiles, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int wps = threads / 256;
  printf("32x32x16: %-38s %d wave(s)/SIMD: %7.3f us per tile-round, %7.3f us per tile per SIMD\n", name, wps, ms * 1e3 / tiles, ms * 1e3 / tiles / wps);
}



// The structure DESIGN section 5 names as the next step, compiler-scheduled as far as hipcc can be led: one wave per SIMD, 128x64 wave
// tile on mfma_32x32x16, the finished tile COPIED to a second register set, every K-step's fragments read one K-step ahead (no read is
// waited for right before its MFMA), and one epilogue group of the previous tile behind the K-step's 16 MFMAs.
template <int THREADS, bool STORES>
__global__ __launch_bounds__(THREADS) void kpipe(const bf16* __restrict__ src, bf16* __restrict__ dst, const float* __restrict__ bias, int tiles) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* mine = lds + (wave & 3) * 24576;
  for (int i = tid; i < 4 * 24576 / 16; i += blockDim.x) reinterpret_cast<f32x4*>(lds)[i] = reinterpret_cast<const f32x4*>(src)[(i + blockIdx.x * 97) & 65535];
  __syncthreads();
  f32x16 acc[4][2], prev[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; prev[i][j][r] = 0.1f * i + 0.01f * r + 0.05f * lane; }
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = bias[(lane * 8 + e) & 1023];
  bf16* gdst = dst + ((size_t)blockIdx.x * blockDim.x + tid) * 8;
  bf16x8 fa[2][2][4], fb[2][2][2];     // [buffer][k half][tile]
  auto load = [&](int buf, int kt) {
    const char* st = mine + (kt % 3) * 8192;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[buf][kk][j] = *reinterpret_cast<const bf16x8*>(st + ((kk * 2048 + j * 1024 + lane * 16) & 8191));
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[buf][kk][i] = *reinterpret_cast<const bf16x8*>(st + ((4096 + kk * 2048 + i * 512 + lane * 16) & 8191));
    }
  };
  load(0, 0);
  for (int t = 0; t < tiles; ++t) {
#pragma unroll
    for (int kt = 0; kt < 16; ++kt) {
      const int cur = kt & 1;
      load(cur ^ 1, kt + 1);                       // next K-step's fragments (the last one reads ahead into the next tile's first stage)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][kk][j], fa[cur][kk][i], acc[i][j], 0, 0, 0);
      {   // epilogue group kt of the previous tile
        const int i = kt >> 2, j = (kt >> 1) & 1, h = kt & 1;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_fast(prev[i][j][8 * h + e] + bv[e]);
        bf16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (bf16)v[e];
        f32x4 raw;
        __builtin_memcpy(&raw, &r, 16);
        if (STORES) __builtin_nontemporal_store(raw, reinterpret_cast<f32x4*>(gdst + ((size_t)kt + 16 * (size_t)(t & 3)) * 8 * 131072));
        else { asm volatile("" :: "v"(raw)); }
      }
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read (12 per K-step)
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);   // 6 VALU
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // the finished tile moves to the second register set (128 register moves: 0.25 us), the accumulators restart
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { prev[i][j][r] = acc[i][j][r]; acc[i][j][r] = 0.f; }
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(prev[i][0]), "+v"(prev[i][1]));     // (in vector registers, not accumulation registers)
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0][0] + acc[i][1][3] + prev[i][0][5] + prev[i][1][7];
  if (s == 12345.678f) dst[0] = (bf16)s;
}

// Power / clock test: the same kernel on a fraction of the chip.  If a wave's time per tile falls and its clock (shader cycles per
// microsecond) rises when fewer CUs are busy, the full-chip rate is set by the power budget, not by a pipe.
template <int MODE, int THREADS>
void run_grid(const char* name, const bf16* src, bf16* dst, const float* bias, unsigned long long* cyc) {
  const int tiles = 400;
  auto kern = k<MODE, 0, THREADS>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 24576);
  for (int grid = 256; grid >= 16; grid /= 2) {
    kern<<<grid, THREADS, 4 * 24576>>>(src, dst, bias, tiles, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    kern<<<grid, THREADS, 4 * 24576>>>(src, dst, bias, tiles, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s %3d blocks of %d waves/SIMD: %7.3f us per tile-round, %6llu shader cycles -> %.2f GHz\n", name, grid, THREADS / 256, ms * 1e3 / tiles, c,
           (double)c / (ms * 1e3 / tiles) * 1e-3);
  }
}

int main() {
  bf16 *src, *dst; float* bias; unsigned long long* cyc;
  hipMalloc(&src, (size_t)256 * 64 * 8192 * 2 + (1 << 20)); hipMalloc(&dst, (size_t)64 * 8 * 131072 * 2 + (1 << 24)); hipMalloc(&bias, 4096); hipMalloc(&cyc, 256 * 8);
  {  // random-ish operands
    const size_t n = (size_t)256 * 64 * 8192 + (1 << 19);
    unsigned short* h = (unsigned short*)malloc(n * 2);
    unsigned x = 12345;
    for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = (unsigned short)(0x3c00 + ((x >> 16) & 0x3ff) + ((x >> 31) << 15)); }
    hipMemcpy(src, h, n * 2, hipMemcpyHostToDevice);
    free(h);
    float hb[1024]; for (int i = 0; i < 1024; ++i) hb[i] = 0.01f * (i % 37) - 0.2f;
    hipMemcpy(bias, hb, 4096, hipMemcpyHostToDevice);
  }
#define ALL(T) \
  run<0, 0, T>("K-steps only (32 MFMA + 12 ds_read each)", src, dst, bias, cyc); \
  run<1, 0, T>("epilogue groups only", src, dst, bias, cyc); \
  run<2, 0, T>("16 K-steps then 16 groups", src, dst, bias, cyc); \
  run<3, 0, T>("a group interleaved into every K-step", src, dst, bias, cyc); \
  run<4, 0, T>("waves 4-7 half a tile out of phase", src, dst, bias, cyc); \
  run<0, 1, T>("K-steps only", src, dst, bias, cyc); \
  run<2, 1, T>("16 K-steps then 16 groups", src, dst, bias, cyc); \
  run<3, 1, T>("a group interleaved into every K-step", src, dst, bias, cyc);
  ALL(256)
  ALL(512)
#define ALL32(T) \
  run32<0, T>("K-steps only (16 MFMA 32x32x16)", src, dst, bias, cyc); \
  run32<1, T>("epilogue groups only", src, dst, bias, cyc); \
  run32<2, T>("16 K-steps then 16 groups", src, dst, bias, cyc); \
  run32<3, T>("a group interleaved into every K-step", src, dst, bias, cyc); \
  run32<5, T>("... the same without the stores", src, dst, bias, cyc); \
  run32<4, T>("waves 4-7 half a tile out of phase", src, dst, bias, cyc); \
  run32<7, T>("... with the matrix phase at s_setprio 2", src, dst, bias, cyc);
  ALL32(256)
  ALL32(512)
  {
    const int tiles = 200;
    auto kern = kbig<256>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 24576);
    kern<<<256, 256, 4 * 24576>>>(src, dst, tiles);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    kern<<<256, 256, 4 * 24576>>>(src, dst, tiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("128x128 wave tile, K-steps only, 1 wave/SIMD: %7.3f us per (double-size) tile = %7.3f us per 512 MFMAs per SIMD\n", ms * 1e3 / tiles, ms * 1e3 / tiles / 2);
  }
  {
    const int tiles = 200;
    auto k1 = kpipe<256, true>; auto k0 = kpipe<256, false>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 24576);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 24576);
    for (int st = 1; st >= 0; --st) {
      if (st) k1<<<256, 256, 4 * 24576>>>(src, dst, bias, tiles); else k0<<<256, 256, 4 * 24576>>>(src, dst, bias, tiles);
      hipDeviceSynchronize();
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (st) k1<<<256, 256, 4 * 24576>>>(src, dst, bias, tiles); else k0<<<256, 256, 4 * 24576>>>(src, dst, bias, tiles);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("32x32x16 pipelined (fragments one K-step ahead, previous tile's group behind every K-step%s), 1 wave/SIMD: %7.3f us per tile per SIMD\n",
             st ? "" : ", no stores", ms * 1e3 / tiles);
    }
  }
  run_grid<0, 512>("K-steps only", src, dst, bias, cyc);
  run_grid<5, 512>("K-steps, no LDS reads", src, dst, bias, cyc);
  run_grid<5, 256>("K-steps, no LDS reads", src, dst, bias, cyc);
  run_grid<0, 256>("K-steps only", src, dst, bias, cyc);
  run_grid<1, 512>("epilogue groups only", src, dst, bias, cyc);
  run_grid<2, 512>("K-steps then groups", src, dst, bias, cyc);
  return 0;
}
