// How many vector instructions hide behind a v_mfma_f32_32x32x16_bf16 (8 passes: the matrix pipe is busy 32 cycles, the vector issue port 8
// of them)?  One wave per SIMD, two independent accumulator chains, N fillers of one kind behind every MFMA, all inline asm (nothing for
// the compiler to move); the same with two waves per SIMD (the fillers then also compete with the partner's MFMAs).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_fillers.hip -o mfma_fillers
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NF, int KIND, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, int iters, float seed) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + threadIdx.x * 0.001f + j); b[j] = (__bf16)(seed * 0.5f + j * 0.01f); }
  f32x16 c0 = {0}, c1 = {0};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x * 1e-3f;
  const float cc = 1.0001f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (m & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(b), "v"(a));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[f & 7]) : "v"(cc));
        if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[f & 7]));
        if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[f & 7]) : "v"(cc));
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NF, int KIND, int THREADS>
double run() {
  float* out;
  hipMalloc(&out, 256 * 1024 * 4);
  const int iters = 4000;
  k<NF, KIND, THREADS><<<256, THREADS>>>(out, 10, 1.0f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<NF, KIND, THREADS><<<256, THREADS>>>(out, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(out);
  return ms * 1e6 / (iters * 16.0) / (THREADS / 256);   // ns per MFMA and SIMD
}

template <int KIND, int THREADS>
void row(const char* name) {
  printf("%-28s %d wave(s)/SIMD, ns per MFMA with 0..8 fillers: %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f\n", name, THREADS / 256,
         run<0, KIND, THREADS>(), run<1, KIND, THREADS>(), run<2, KIND, THREADS>(), run<3, KIND, THREADS>(), run<4, KIND, THREADS>(), run<5, KIND, THREADS>(),
         run<6, KIND, THREADS>(), run<7, KIND, THREADS>(), run<8, KIND, THREADS>());
}

int main() {
  row<0, 256>("v_fma_f32 fillers");
  row<1, 256>("v_exp_f32 fillers");
  row<2, 256>("v_cvt_pk_bf16_f32 fillers");
  row<0, 512>("v_fma_f32 fillers");
  row<1, 512>("v_exp_f32 fillers");
  return 0;
}
