// Micro-benchmark: do the MFMA phase of one wave and the VALU phase of another wave on the same SIMD overlap?
// Each wave loops { PHASE M: 16 x v_mfma_f32_32x32x16_bf16 (two accumulator chains) ; PHASE V: NV fma + NE exp }.
// mode 1 = M only, 2 = V only, 3 = both (sequential inside every wave); waves per SIMD = threads / 256.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE, int IL>
__global__ void k(float* out, int iters, float seed) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + threadIdx.x * 0.001f + j); b[j] = (__bf16)(seed * 0.5f + j * 0.01f); }
  f32x16 c0 = {0}, c1 = {0};
  float v[32];
  for (int i = 0; i < 32; ++i) v[i] = seed + i + threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (IL == 0) {
      if (MODE & 1) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
        }
      }
      if (MODE & 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < 32; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = __builtin_amdgcn_exp2f(v[i] * 1e-6f);
      }
    } else {   // interleaved inside the wave: one MFMA per 10 VALU
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        if (MODE & 1) { if (m & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0); else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); }
        if (MODE & 2) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[(m * 8 + i) & 31] = __builtin_fmaf(v[(m * 8 + i) & 31], 1.0001f, 0.5f);
#pragma unroll
          for (int i = 0; i < 2; ++i) v[(m * 2 + i) & 31] = __builtin_amdgcn_exp2f(v[(m * 2 + i) & 31] * 1e-6f);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  for (int i = 0; i < 32; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / iters;
}

template <int MODE, int IL>
void run(int threads, const char* name) {
  float* out;
  hipMalloc(&out, 256 * 1024 * 4);
  const int iters = 2000;
  k<MODE, IL><<<256, threads>>>(out, iters, 1.0f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<MODE, IL><<<256, threads>>>(out, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float cyc; hipMemcpy(&cyc, out, 4, hipMemcpyDeviceToHost);
  printf("%-34s threads %4d (%d waves/SIMD): %8.1f cycles/iter/wave (wave 0), %7.3f us/iter wall\n", name, threads, threads / 256, cyc, ms * 1e3 / iters);
  hipFree(out);
}

int main() {
  for (int threads : {256, 512, 1024}) {
    run<1, 0>(threads, "MFMA only (16 per iter)");
    run<2, 0>(threads, "VALU only (128 fma + 32 exp)");
    run<3, 0>(threads, "MFMA phase then VALU phase");
    run<3, 1>(threads, "MFMA/VALU interleaved in-wave");
  }
  return 0;
}
