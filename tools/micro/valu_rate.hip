// Calibration: issue cost of single vector instructions on gfx950, in shader clocks per wave64 instruction per SIMD, at 1 / 2 / 4
// waves per SIMD.  32 independent registers per kind, REP passes per loop iteration, inline asm so that nothing is folded.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define CHAIN32(STMT) _Pragma("unroll") for (int r = 0; r < 4; ++r) _Pragma("unroll") for (int i = 0; i < 32; ++i) { STMT; }

template <int KIND>
__global__ void k(float* out, unsigned long long* clk, int iters, float seed) {
  float v[32];
  f32x2 p[16];
  unsigned u[32];
  for (int i = 0; i < 32; ++i) { v[i] = seed + i + threadIdx.x * 1e-3f; u[i] = (unsigned)(seed * 977) + i * 7919u + threadIdx.x; }
  for (int i = 0; i < 16; ++i) p[i] = f32x2{v[2 * i], v[2 * i + 1]};
  const float c = 1.0001f;
  const f32x2 c2 = {1.0001f, 0.9999f};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) CHAIN32(asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c)))
    if (KIND == 1) { _Pragma("unroll") for (int r = 0; r < 8; ++r) _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(c2)); }
    if (KIND == 2) CHAIN32(asm volatile("v_exp_f32 %0, %0" : "+v"(v[i])))
    if (KIND == 3) CHAIN32(asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i])))
    if (KIND == 4) CHAIN32(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 31])))
    if (KIND == 5) CHAIN32(asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 31])))
    if (KIND == 6) CHAIN32(asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c)))
    if (KIND == 7) CHAIN32(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 31])))
    if (KIND == 8) CHAIN32(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c)))
    if (KIND == 9) CHAIN32(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(unsigned long long*)&u[(2 * i) & 31]) : "v"(u[(i + 5) & 31]), "v"(u[(i + 9) & 31]) : "vcc"))
    if (KIND == 10) { _Pragma("unroll") for (int r = 0; r < 8; ++r) _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2)); }
    if (KIND == 11) CHAIN32(asm volatile("v_log_f32 %0, %0" : "+v"(v[i])))
    if (KIND == 12) CHAIN32(asm volatile("v_sqrt_f32 %0, %0" : "+v"(v[i])))
    if (KIND == 13) CHAIN32(asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 31])))
    if (KIND == 14) CHAIN32(asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(v[(i + 1) & 31])))
    if (KIND == 15) CHAIN32(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 31]), "v"(u[(i + 2) & 31])))
    if (KIND == 16) CHAIN32(asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c)))
    if (KIND == 17) CHAIN32(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c)))
    if (KIND == 18) CHAIN32(asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 31])))
    if (KIND == 19) CHAIN32(asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(u[i])))
    if (KIND == 20) CHAIN32(asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 31])))
    if (KIND == 21) CHAIN32(asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "+v"(u[i]) : "v"(u[(i + 1) & 31])))
    if (KIND == 22) CHAIN32(asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[i]) : "v"(u[(i + 1) & 31])))
    if (KIND == 23) CHAIN32(asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(u[i])))
    if (KIND == 24) CHAIN32(asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 31])))
    if (KIND == 25) CHAIN32(asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 31]) : "vcc"))
    if (KIND == 27) CHAIN32(asm volatile("v_mov_b32 %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 31])))
    if (KIND == 28) CHAIN32(asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c)))
    if (KIND == 29) CHAIN32(asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(v[i]) : "v"(c)))
    if (KIND == 30) CHAIN32(asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(v[i])))
    if (KIND == 31) CHAIN32(asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 31]), "v"(u[(i + 2) & 31])))
    if (KIND == 32) CHAIN32(asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 31]), "v"(u[(i + 2) & 31])))
    if (KIND == 33) CHAIN32(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i & 15]) : "v"(c2)))
    if (KIND == 34) CHAIN32(asm volatile("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(u[i]) : "v"(u[(i + 1) & 31])))
    if (KIND == 35) CHAIN32(asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 31]), "v"(u[(i + 2) & 31])))
    if (KIND == 36) CHAIN32(asm volatile("v_mul_f32 %0, 0x3fb8aa3b, %0" : "+v"(v[i])))
    if (KIND == 37) CHAIN32(asm volatile("v_fmaak_f32 %0, %0, %1, 0x3fb8aa3b" : "+v"(v[i]) : "v"(c)))
    // round 4: the fp16 forms a packed-fp16 GELU would be made of (two results per v_pk_* instruction)
    if (KIND == 40) CHAIN32(asm volatile("v_pk_fma_f16 %0, %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 31])))
    if (KIND == 41) CHAIN32(asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 31])))
    if (KIND == 42) CHAIN32(asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 31])))
    if (KIND == 43) CHAIN32(asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 31])))
    if (KIND == 44) CHAIN32(asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c)))
    if (KIND == 45) CHAIN32(asm volatile("v_exp_f16 %0, %0" : "+v"(u[i])))
    if (KIND == 46) CHAIN32(asm volatile("v_rcp_f16 %0, %0" : "+v"(u[i])))
    if (KIND == 47) CHAIN32(asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(u[i])))
    if (KIND == 48) CHAIN32(asm volatile("v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(u[i])))
    if (KIND == 49) CHAIN32(asm volatile("v_exp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(u[i])))
    if (KIND == 50) CHAIN32(asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(v[(i + 1) & 31])))
    if (KIND == 51) CHAIN32(asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(v[i]) : "v"(u[(i + 1) & 31]), "v"(u[(i + 2) & 31])))
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 32; ++i) s += v[i] + (float)u[i];
  for (int i = 0; i < 16; ++i) s += p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name) {
  float* out;
  unsigned long long* clk;
  hipMalloc(&out, 1024 * 1024 * 4);
  hipMalloc(&clk, 4096 * 8);
  const int iters = 2000;
  printf("%-16s", name);
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int threads = 256 * wps;   // 4 SIMDs x wps waves
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<256, threads>>>(out, clk, 10, 1.0f);
    hipEventRecord(e0);
    k<KIND><<<256, threads>>>(out, clk, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int b = 0; b < 256; ++b) cyc += (double)h[b];
    cyc /= 256;
    // per SIMD: wps waves x iters x 128 instructions
    const double n = (double)wps * iters * 128;
    printf("  %dw: %6.2f ns/instr (%5.2f counter ticks)", wps, ms * 1e6 / n, cyc / n);
  }
  printf("\n");
  hipFree(out); hipFree(clk);
}

int main() {
  run<0>("v_fma_f32");
  run<1>("v_pk_fma_f32");
  run<10>("v_pk_mul_f32");
  run<6>("v_min_f32");
  run<14>("v_max3_f32");
  run<7>("v_xor_b32");
  run<15>("v_perm_b32");
  run<8>("v_cvt_pk_bf16");
  run<2>("v_exp_f32");
  run<3>("v_rcp_f32");
  run<11>("v_log_f32");
  run<12>("v_sqrt_f32");
  run<4>("v_mul_lo_u32");
  run<5>("v_mul_hi_u32");
  run<13>("v_mul_u32_u24");
  run<9>("v_mad_u64_u32");
  run<16>("v_add_f32"); run<28>("v_sub_f32"); run<17>("v_mul_f32"); run<36>("v_mul_f32 lit"); run<29>("v_fmac_f32"); run<37>("v_fmaak_f32"); run<33>("v_pk_add_f32");
  run<30>("v_cvt_f32_u32"); run<27>("v_mov_b32"); run<18>("v_and_b32"); run<19>("v_lshlrev_b32"); run<20>("v_add_u32"); run<35>("v_add3_u32");
 run<31>("v_or3_b32"); run<32>("v_and_or_b32"); run<24>("v_lshl_or_b32"); run<23>("v_bfe_u32"); run<22>("v_alignbit_b32");
  if (getenv("VALU_F16")) {
    run<40>("v_pk_fma_f16"); run<41>("v_pk_mul_f16"); run<42>("v_pk_add_f16"); run<43>("v_pk_min_f16"); run<44>("v_cvt_pkrtz_f16"); run<45>("v_exp_f16");
    run<46>("v_rcp_f16"); run<47>("v_cvt_f32_f16"); run<48>("v_cvt_f32_f16 hi"); run<49>("v_exp_f16 sdwa hi"); run<50>("v_med3_f32"); run<51>("v_dot2_f32_bf16");
    return 0;
  }
  run<21>("v_sub_u32_sdwa"); run<34>("v_and_b32_sdwa"); run<25>("v_cmp+cndmask x2");
  return 0;
}
