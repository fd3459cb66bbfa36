// Probe for the split-precision modes: (1) does v_mfma_f32_16x16x32_f16 keep subnormal f16 inputs, (2) error of a K = 512 dot product
// against an fp64 reference for bf16, bf16 hi/lo (3 products), f16, f16 hi/lo (3 products), fp32 fma.
// Build: hipcc --offload-arch=gfx950 -O3 split_mfma.hip -o split_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// D[16x16] = A[16xK] B[16xK]^T, A / B fp32 row-major [16][K]; mode 0 bf16, 1 bf16x3, 2 f16, 3 f16x3
template <int MODE>
__global__ void k(const float* A, const float* B, float* D, int K) {
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 32) {
    float a[8], b[8];
    for (int e = 0; e < 8; ++e) { a[e] = A[r * K + k0 + 8 * g + e]; b[e] = B[r * K + k0 + 8 * g + e]; }
    if constexpr (MODE < 2) {
      bf16x8 ah, al, bh, bl;
      for (int e = 0; e < 8; ++e) { ah[e] = (__bf16)a[e]; al[e] = (__bf16)(a[e] - (float)ah[e]); bh[e] = (__bf16)b[e]; bl[e] = (__bf16)(b[e] - (float)bh[e]); }
      if (MODE == 1) { acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0); }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
    } else {
      f16x8 ah, al, bh, bl;
      for (int e = 0; e < 8; ++e) { ah[e] = (_Float16)a[e]; al[e] = (_Float16)(a[e] - (float)ah[e]); bh[e] = (_Float16)b[e]; bl[e] = (_Float16)(b[e] - (float)bh[e]); }
      if (MODE == 3) { acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0); }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
    }
  }
  // D[m = 4g + i][n = r]  (A rows on m: A operand first)
  for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = acc[i];
}

int main() {
  const int K = 512;
  float *A, *B, *D;
  hipMallocManaged(&A, 16 * K * 4); hipMallocManaged(&B, 16 * K * 4); hipMallocManaged(&D, 256 * 4);
  // 1: subnormal probe: A = 2^-20 (f16 subnormal), B = 2^10: exact product 2^-10 per element, K of them
  for (int i = 0; i < 16 * K; ++i) { A[i] = ldexpf(1.f, -20); B[i] = 1024.f; }
  k<2><<<1, 64>>>(A, B, D, K); hipDeviceSynchronize();
  printf("f16 subnormal inputs: D = %g (kept: %g, flushed: 0)\n", D[0], K * ldexp(1.0, -10));
  srand(7);
  for (int scale_i = 0; scale_i < 3; ++scale_i) {
    const float sc = scale_i == 0 ? 1.f : (scale_i == 1 ? 1e-3f : 30.f);
    for (int i = 0; i < 16 * K; ++i) { A[i] = sc * ((rand() / (float)RAND_MAX) * 2 - 1); B[i] = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.05f; }
    const char* names[4] = {"bf16", "bf16x3", "f16", "f16x3"};
    for (int mode = 0; mode < 5; ++mode) {
      if (mode == 0) k<0><<<1, 64>>>(A, B, D, K);
      if (mode == 1) k<1><<<1, 64>>>(A, B, D, K);
      if (mode == 2) k<2><<<1, 64>>>(A, B, D, K);
      if (mode == 3) k<3><<<1, 64>>>(A, B, D, K);
      hipDeviceSynchronize();
      double maxe = 0, ref_rms = 0;
      for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        double ref = 0; float f = 0.f;
        for (int kk = 0; kk < K; ++kk) { ref += (double)A[m * K + kk] * B[n * K + kk]; f = fmaf(A[m * K + kk], B[n * K + kk], f); }
        const double e = fabs((mode == 4 ? f : D[m * 16 + n]) - ref);
        if (e > maxe) maxe = e;
        ref_rms += ref * ref;
      }
      ref_rms = sqrt(ref_rms / 256);
      printf("scale %g  %-7s max |err| / rms(ref) = %.3e\n", sc, mode == 4 ? "fp32fma" : names[mode], maxe / ref_rms);
    }
  }
  return 0;
}
