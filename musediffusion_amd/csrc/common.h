// Shared device helpers for libmusehip (gfx950 / CDNA4 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/musehip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MH_WAVE 64

// ---- error plumbing (thread-local last error string, returned by mh_last_error()).
void mh_set_error(const char* fmt, ...);
#define MH_CHECK_ARG(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      mh_set_error(__VA_ARGS__);       \
      return MH_ERR_INVALID;           \
    }                                  \
  } while (0)
// A launch site clears any stale (sticky) runtime error first so that the check after it reports
// only this launch's status.
#define MH_LAUNCH(...)                      \
  do {                                      \
    (void)hipGetLastError();                \
    hipLaunchKernelGGL(__VA_ARGS__);        \
  } while (0)
#define MH_CHECK_LAUNCH()                                        \
  do {                                                           \
    hipError_t e__ = hipGetLastError();                          \
    if (e__ != hipSuccess) {                                     \
      mh_set_error("%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return MH_ERR_HIP;                                         \
    }                                                            \
  } while (0)
#define MH_HIP(call)                                             \
  do {                                                           \
    hipError_t e__ = (call);                                     \
    if (e__ != hipSuccess) {                                     \
      mh_set_error("%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return MH_ERR_HIP;                                         \
    }                                                            \
  } while (0)

// ---- scalar conversions.  Plain casts: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving).
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// 8 contiguous elements <-> 8 floats (16 B for bf16, 32 B for f32).
__device__ __forceinline__ void load8(const bf16* p, float (&v)[8]) {
  bf16x8 r = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
}
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
__device__ __forceinline__ void store8(bf16* p, const float (&v)[8]) {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (bf16)v[i];
  *reinterpret_cast<bf16x8*>(p) = r;
}
// streaming store: bytes that no block of this kernel reads again must not evict the operand panels from L2
__device__ __forceinline__ void store8_nt(bf16* p, const float (&v)[8]) {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (bf16)v[i];
  f32x4 raw;
  __builtin_memcpy(&raw, &r, 16);
  __builtin_nontemporal_store(raw, reinterpret_cast<f32x4*>(p));
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}

// ---- wave64 reductions through cross-lane shuffles (no LDS).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact-erf GELU (HF hidden_act="gelu"), tanh, SiLU in fp32
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }
// GELU for the bf16 path: x * Phi(x) with Phi(x) ~ sigmoid(x (a + b x^2 + c x^4)), a minimax fit of the erf form
// (max |error| 2.5e-5 over all x - two orders below the bf16 rounding of the result; the fp32 parity path
// uses erff).  One v_exp, one v_rcp, 7 cheap VALU ops; x^2 is clamped at 64 where the quartic is still
// increasing (sigmoid is 0 / 1 to fp32 precision beyond |x| = 8).
__device__ __forceinline__ float gelu_erf_fast(float x) {
  constexpr float kL2E = 1.4426950408889634f;
  const float u = fminf(x * x, 64.0f);
  float p = fmaf(0.0007030335771326705f * kL2E, u, -0.07401129204508086f * kL2E);
  p = fmaf(p, u, -1.5950157685701116f * kL2E);
  const float e = __builtin_amdgcn_exp2f(x * p);              // exp(-x (a + b u + c u^2))
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// d/dx of the erf-form GELU, Phi(x) + x phi(x), for the fused backward epilogue: erf by Abramowitz-Stegun 7.1.26
// (|err| <= 1.5e-7); exp(-x^2/2) is shared between the erf tail and the density
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);   // exp(-x^2 / 2)
  const float erf_abs = 1.0f - p * t * e;
  return 0.5f * (1.0f + copysignf(erf_abs, x)) + x * 0.3989422804014327f * e;
}

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
