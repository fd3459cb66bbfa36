// Shared device helpers for libmusehip (gfx950 / CDNA4 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/musehip.h"

// A/B switches, ablation knobs and diagnostics (include/musehip_dbg.h) exist only in the debug library (-DMH_ABLATE ->
// libmusehip_dbg.so).  The production library has no process-global mutable configuration: there every MH_KNOB is a compile-time
// constant holding the default, the setters are not compiled, and the kernel variants only a knob can reach are dead code.
// per-device bookkeeping of hipFuncSetAttribute (it acts on the CURRENT device's copy of a kernel): the library may serve several
// devices from one process
constexpr int MH_MAX_DEVICES = 64;
static inline int mh_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MH_MAX_DEVICES) dev = 0;
  return dev;
}
#ifdef MH_ABLATE
#define MH_KNOB(type, name, value) type name = value
#else
#define MH_KNOB(type, name, value) constexpr type name = value
#endif

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
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
// Per-launch timing (mh_profile_start / mh_profile_stop, engine.hip): when switched on - never inside a hipGraph capture - every
// launch is bracketed by two HIP events on ITS stream, so bench.py can report each kernel's share of a step and its rate against
// the roofline live, the way rocprofv3 --kernel-trace lists them.  `kernel` is a single macro argument (templates in parentheses).
extern int g_mh_prof_on;
void mh_prof_begin(const char* kernel, unsigned grid_x, unsigned block_x, hipStream_t stream);
void mh_prof_end(hipStream_t stream);
void mh_prof_note(const char* fmt, ...);     // free-form detail attached to the NEXT launch record (shape, role)
#define MH_LAUNCH(kernel, grid, block, shmem, stream, ...)                      \
  do {                                                                          \
    (void)hipGetLastError();                                                    \
    if (g_mh_prof_on) mh_prof_begin(#kernel, dim3(grid).x, dim3(block).x, stream); \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);        \
    if (g_mh_prof_on) mh_prof_end(stream);                                      \
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

// the same function on 8 values as 4 register pairs: v_pk_mul / v_pk_fma / v_pk_add (two fp32 lanes per instruction) - 5 full-rate +
// 2 transcendental instructions per value instead of 7 + 2; the operations and their order are gelu_erf_fast's (bit-identical)
__device__ __forceinline__ void gelu_erf_fast8(float (&v)[8]) {
  constexpr float kL2E = 1.4426950408889634f;
  const f32x2 c2 = {0.0007030335771326705f * kL2E, 0.0007030335771326705f * kL2E};
  const f32x2 c1 = {-0.07401129204508086f * kL2E, -0.07401129204508086f * kL2E};
  const f32x2 c0 = {-1.5950157685701116f * kL2E, -1.5950157685701116f * kL2E};
  const f32x2 one = {1.0f, 1.0f};
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    f32x2 x = {v[e], v[e + 1]};
    f32x2 u = x * x;
    u.x = fminf(u.x, 64.0f); u.y = fminf(u.y, 64.0f);
    f32x2 p = __builtin_elementwise_fma(c2, u, c1);
    p = __builtin_elementwise_fma(p, u, c0);
    const f32x2 t = x * p;
    const f32x2 ex = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    const f32x2 d = one + ex;
    const f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    x = x * r;
    v[e] = x.x; v[e + 1] = x.y;
  }
}

// gelu_erf_fast and ITS derivative from the same exp / rcp pair (training forward: the derivative is stored instead of the
// pre-activation, so the backward's epilogue is one multiply).  With e = exp(x A(u)), s = 1 / (1 + e), f = x s and
// B(u) = d/dx [x A(x^2)] = A0 + 3 A1 u + 5 A2 u^2:   f' = s - x s^2 e B = s - f (1 - s) B.   |f' - gelu'| <= 1.1e-4.
__device__ __forceinline__ void gelu_erf_fast8_dgelu(float (&v)[8], float (&gp)[8]) {
  constexpr float kL2E = 1.4426950408889634f;
  constexpr float A0 = -1.5950157685701116f, A1 = -0.07401129204508086f, A2 = 0.0007030335771326705f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = v[e];
    const float u = fminf(x * x, 64.0f);
    float p = fmaf(A2 * kL2E, u, A1 * kL2E);
    p = fmaf(p, u, A0 * kL2E);
    const float ex = __builtin_amdgcn_exp2f(x * p);
    const float s = __builtin_amdgcn_rcpf(1.0f + ex);
    const float f = x * s;
    float b = fmaf(5.0f * A2, u, 3.0f * A1);
    b = fmaf(b, u, A0);
    v[e] = f;
    gp[e] = fmaf(-(f * (1.0f - s)), b, s);
  }
}

// tanh for the bf16 path: 1 - 2 / (1 + 2^(2 x log2 e)) with one v_exp and one v_rcp (|error| < 3e-7 relative to fp32 tanh for
// |x| < 10, exact saturation beyond: two orders below the bf16 rounding of the result); the fp32 parity path uses tanhf
__device__ __forceinline__ float tanh_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(fminf(x * 2.885390081777927f, 126.0f));
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
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

// ---- Philox4x32 (counter-based RNG): 10 rounds for the sampling noise (mh_trunc_normal), 7 rounds (the fewest that
// pass BigCrush, Salmon et al. SC'11) for train-mode dropout masks, where one call pays for 8 elements.
template <int ROUNDS>
__device__ __forceinline__ void mh_philox(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// Train-mode dropout (models/network.py:149; HF BertSelfOutput / BertOutput / BertSelfAttention dropouts).  A keep
// decision takes 16 random bits: keep <=> u16 >= thr, thr = round(p * 65536), so the drop rate is p to 2^-17.
// Device-side view of mh_dropout (include/musehip.h): all by value, usable inside kernel argument structs.
struct DropArgs {
  uint32_t thr8, thr_tie;  // attention probabilities: 8-bit threshold floor(256 p) and the 8-bit tie-break threshold (below)
  uint32_t thr;            // 0: dropout off
  float rscale;            // 1 / (1 - p)
  uint32_t seed_lo, seed_hi, off_lo, off_hi;
  const uint8_t* mask;     // test-only: explicit keep flags per element (row-major, dense sites), overrides Philox
};
__device__ __forceinline__ uint32_t drop_thr(float p) { return (uint32_t)(p * 65536.0f + 0.5f); }
// dense sites: element e = row * cols + col; one call serves the 8 consecutive elements of group e >> 3 (cols % 8 == 0)
// returns bit i set <=> element 8 g + i is KEPT
__device__ __forceinline__ uint32_t drop_keep8(const DropArgs& d, uint64_t group) {
  uint32_t c[4] = {(uint32_t)group, (uint32_t)(group >> 32), d.off_lo, d.off_hi};
  mh_philox<7>(c, d.seed_lo, d.seed_hi);
  uint32_t m = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    m |= ((c[w] & 0xffffu) >= d.thr ? 1u : 0u) << (2 * w);
    m |= ((c[w] >> 16) >= d.thr ? 1u : 0u) << (2 * w + 1);
  }
  return m;
}
// same decision for 8 consecutive elements starting at element index e0 (e0 % 8 == 0), from the explicit mask or Philox
__device__ __forceinline__ uint32_t drop_keep8_at(const DropArgs& d, uint64_t e0) {
  if (d.mask) {
    const uint64_t raw = *reinterpret_cast<const uint64_t*>(d.mask + e0);
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) m |= (((raw >> (8 * i)) & 0xff) != 0 ? 1u : 0u) << i;
    return m;
  }
  return drop_keep8(d, e0 >> 3);
}
// attention probabilities: the keep flags of one lane of an S^T tile (query q, lane half h) for the 32-key block kb.  ONE
// Philox call gives the lane's 16 elements a random BYTE each (register r of the 32x32x16 accumulator <-> byte r of the 128
// bits <-> key (r & 3) + 8 (r >> 2) + 4 h): the element is dropped when its byte is below thr8 = floor(256 p), kept when above,
// and on the tie (probability 2^-8) decided by the NEXT byte against thr_tie = round(256 frac(256 p)) - so the drop rate is p to
// 2^-16, as at the dense sites, at half the generator work (the forward is VALU-bound).  The tie-break byte is another
// element's primary byte: a dependence that touches 0.4% of the elements.  nkb = ceil(L / 32).
// Returns bit r set <=> P[q][32 kb + key(r)] is KEPT.
// (Instruction form: byte r over byte r + 1 is ONE 16-bit number, and the rule above is `number >= 256 thr8 + thr_tie`.  A v_perm
// puts two neighbours' numbers into the halves of a dword, a subtraction leaves "dropped" in the sign bit and v_alignbit shifts it
// into the mask: 2.5 instructions per element instead of the 9 of three byte compares - on gfx950 compares, selects, shifts and
// byte extracts all issue at half the rate of and / add / xor, tools/micro/valu_rate.hip.)
__device__ __forceinline__ uint32_t drop_keep_attn_from(const uint32_t (&c)[4], uint32_t thr16) {
  uint32_t m = 0;   // bit r <=> element r is DROPPED; elements enter from the top one down
#pragma unroll
  for (int w = 3; w >= 0; --w) {
    // bytes (B[r + 1], B[r], B[r + 2], B[r + 1]), low to high, for r = 4 w + 2 and r = 4 w
    const uint32_t d23 = __builtin_amdgcn_perm(c[(w + 1) & 3], c[w], 0x03040203u);
    const uint32_t d01 = __builtin_amdgcn_perm(c[(w + 1) & 3], c[w], 0x01020001u);
    m = __builtin_amdgcn_alignbit(m, (d23 >> 16) - thr16, 31);
    m = __builtin_amdgcn_alignbit(m, (d23 & 0xffffu) - thr16, 31);
    m = __builtin_amdgcn_alignbit(m, (d01 >> 16) - thr16, 31);
    m = __builtin_amdgcn_alignbit(m, (d01 & 0xffffu) - thr16, 31);
  }
  return ~m & 0xffffu;
}
__device__ __forceinline__ uint32_t drop_keep_attn(const DropArgs& d, int64_t bh, int L, int nkb, int q, int kb, int h) {
  const uint64_t idx = ((((uint64_t)bh * L + q) * nkb + kb) << 1) | (uint64_t)h;
  uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), d.off_lo, d.off_hi};
  mh_philox<7>(c, d.seed_lo, d.seed_hi);
  return drop_keep_attn_from(c, (d.thr8 << 8) + d.thr_tie);
}
// Applying a keep flag without a compare + select pair (half-rate on gfx950): v_bfe_i32 turns bit `bit` of the word into 0 / ~0 and
// a v_and clears the dropped value (+0.0f).
__device__ __forceinline__ uint32_t keep_mask(uint32_t word, uint32_t bit) { return (uint32_t)__builtin_amdgcn_sbfe((int)word, bit, 1u); }
__device__ __forceinline__ float and_bits(float x, uint32_t m) { return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x) & m); }
// The keep-bit tensor is LANE-NATIVE: uint32 [B nh][nb = ceil(L / 32) query blocks][ceil(nb / 2) pairs of 32-key blocks][64 lanes].
// The word of lane (lq, h) holds that lane's 16 flags of the pair's even key block in its low half and of the odd block in its high
// half - bit r <-> key (r & 3) + 8 (r >> 2) + 4 h of the block, query lq of the query block: exactly the registers the lane holds
// of an S^T tile, so the forward stores its flags as they are (no ballots) and the dQ kernel loads one word per 64-key tile.
__device__ __forceinline__ int64_t drop_word_index(int64_t bh, int nb, int qb, int kp, int lane) {
  return ((((int64_t)bh * nb + qb) * ((nb + 1) >> 1) + kp) << 6) + lane;
}
// flag of probability (q, k) of (batch, head) bh - for the kernels that do not hold S^T tiles (materialised path)
__device__ __forceinline__ uint32_t drop_flag(const uint32_t* bits, int64_t bh, int nb, int q, int k) {
  const int kk = k & 31, h = (kk >> 2) & 1, r = (kk & 3) + 4 * (kk >> 3);
  return (bits[drop_word_index(bh, nb, q >> 5, k >> 6, (q & 31) + 32 * h)] >> (r + 16 * ((k >> 5) & 1))) & 1u;
}

// XCD-aware block remap (8 XCDs, workgroups dealt round-robin): block b -> a virtual index such that each XCD owns a contiguous run of
// indices, so neighbours in index space (tiles of one row panel, query blocks of one (batch, head)) share that XCD's L2.  Bijective for
// any grid size.
__device__ __forceinline__ int mh_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}
static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
