// Micro-benchmark: L2 -> LDS feed rate per CU for global_load_lds_dwordx4 under different
// per-instruction row shapes, and for register-staged loads.  Each block streams a private-ish
// panel (rows x K bytes) that stays L2 resident.  Build: hipcc --offload-arch=gfx950 -O3 dma_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;

// MODE 0: glds, ROWB bytes contiguous per row per instruction (64 or 128 or 256...), 16B per lane
// MODE 1: global_load_dwordx4 to registers + ds_write_b128
static __device__ int NPANELS_dummy;
template <int ROWB, int MODE, int DEPTH>
__global__ __launch_bounds__(256) void k(const char* __restrict__ src, int64_t row_stride, int rows_per_block, int ksteps,
                                         float* out, int NPANELS, int kwrap) {
  __shared__ __attribute__((aligned(16))) char smem[64 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int LPR = ROWB / 16;            // lanes per row
  constexpr int RPI = 64 / LPR;             // rows per instruction
  // one "stage" = rows_per_block rows x ROWB bytes; each wave issues rows_per_block/RPI/4 instructions
  const int ipw = rows_per_block / RPI / 4;
  const char* base = src + (int64_t)(blockIdx.x % NPANELS) * rows_per_block * row_stride;
  float acc = 0.f;
  int inflight = 0;
  for (int ks = 0; ks < ksteps; ++ks) {
    char* stage = smem + (ks % DEPTH) * (rows_per_block * ROWB);
    for (int j = 0; j < ipw; ++j) {
      const int r0 = (wave * ipw + j) * RPI;
      const int row = r0 + lane / LPR;
      const char* g = base + row * row_stride + (int64_t)(ks % kwrap) * ROWB + (lane % LPR) * 16;
      if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(stage + r0 * ROWB), 16, 0, 0);
      } else {
        f32x4 v = *reinterpret_cast<const f32x4*>(g);
        *reinterpret_cast<f32x4*>(stage + r0 * ROWB + lane * 16) = v;
      }
    }
    inflight++;
    if (inflight >= DEPTH) {
      // wait for the oldest stage (approximate: drain all but DEPTH-1 stages)
      if (MODE == 0) {
        if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(0) : "memory");
      }
      __builtin_amdgcn_s_barrier();
      acc += reinterpret_cast<float*>(smem)[(tid * 4 + ks) & 16383];
      inflight = DEPTH - 1;
    }
  }
  if (acc == 1234.5f) out[0] = acc;
}

template <int ROWB, int MODE, int DEPTH>
void run(const char* name, const char* d, int64_t row_stride, int rows, int ksteps, int blocks, float* out, int npanels = 8, int passes = 8) {
  const int kwrap = ksteps; ksteps *= passes;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<ROWB, MODE, DEPTH>), dim3(blocks), dim3(256), 0, 0, d, row_stride, rows, ksteps, out, npanels, kwrap);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL((k<ROWB, MODE, DEPTH>), dim3(blocks), dim3(256), 0, 0, d, row_stride, rows, ksteps, out, npanels, kwrap);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double bytes = (double)blocks * rows * ROWB * ksteps;
  printf("%-34s rowB=%3d rows=%3d blocks=%4d: %7.1f us  %7.2f TB/s  %6.1f GB/s per CU\n", name, ROWB, rows, blocks,
         ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}

int main() {
  const int64_t row_stride = 1024 + 128;  // bytes
  const int blocks = 512, rows = 256;
  const size_t bytes = (size_t)1024 * rows * row_stride + (1 << 20);
  char* d; float* out;
  hipMalloc(&d, bytes); hipMalloc(&out, 64);
  hipMemset(d, 1, bytes);
  // K extent = 1024 bytes per row
  run<64, 0, 1>("glds 64B rows, depth1", d, row_stride, rows, 16, blocks, out);
  run<128, 0, 1>("glds 128B rows, depth1", d, row_stride, rows, 8, blocks, out);
  run<256, 0, 1>("glds 256B rows, depth1", d, row_stride, rows, 4, blocks, out);
  run<64, 1, 1>("regs 64B rows, depth1", d, row_stride, rows, 16, blocks, out);
  run<128, 1, 1>("regs 128B rows, depth1", d, row_stride, rows, 8, blocks, out);
  run<64, 0, 2>("glds 64B rows, drain-each", d, row_stride, 128, 16, blocks, out);
  run<128, 0, 2>("glds 128B rows, drain-each", d, row_stride, 128, 8, blocks, out);
  // 1 block per CU
  run<64, 0, 1>("glds 64B rows, 256 blocks", d, row_stride, rows, 16, 256, out);
  run<128, 0, 1>("glds 128B rows, 256 blocks", d, row_stride, rows, 8, 256, out);
  run<128, 0, 1>("glds 128B rows, 1024 blocks", d, row_stride, rows, 8, 1024, out);
  return 0;
}
