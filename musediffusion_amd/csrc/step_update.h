// Arithmetic of one reverse-diffusion update shared by the stand-alone update kernels (elementwise.hip) and the fused tail of the
// denoiser (headtail.hip): the in-graph truncated normal and the posterior / DDIM step on a group of 4 consecutive elements.  One
// definition, FP contraction off inside it, so both callers produce the same bits.
#pragma once
#include "common.h"

// ---- Philox4x32-10 truncated normal
__device__ __forceinline__ float u01(uint32_t r) {
#pragma clang fp contract(off)
  return ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

// One thread owns the 4 consecutive elements of a group; Philox call (group, step, stream, attempt) yields four Box-Muller
// normals, candidate k going to element k of the group if that element has not accepted one yet (rejection against the bound,
// resolved in registers: no host sync, diffusion.py:378-388).  Elements are numbered GLOBALLY (first + local index), so a batch
// slice drawn on its own (the decoupled graph branches of _ReverseLoop) gets exactly the values the whole-batch launch gives it.
// Hardware transcendentals (v_log / v_sqrt / v_sin / v_cos: sin and cos take revolutions, so 2 pi u needs no range reduction).
// the four truncated normals of element group g (elements 4 g .. 4 g + 3, numbered globally) at loop step `step`
__device__ __forceinline__ void trunc_normal4(uint64_t g, uint32_t step, float bound, uint32_t seed_lo, uint32_t seed_hi, uint32_t stream_id,
                                              float (&z)[4]) {
#pragma clang fp contract(off)
  z[0] = z[1] = z[2] = z[3] = 0.f;
  uint32_t pending = 0xfu;
  // up to 1024 calls per group (the loop ends with the group's last acceptance: the usual cost is 1 - 2 calls); attempts beyond
  // 255 continue in the top byte of the second counter word, so every value an earlier build drew is unchanged
  for (uint32_t attempt = 0; attempt < 1024 && pending; ++attempt) {
    uint32_t c[4] = {(uint32_t)g, (uint32_t)(g >> 32) | ((attempt >> 8) << 24), step, (stream_id << 8) | (attempt & 0xffu)};
    mh_philox<10>(c, seed_lo, seed_hi);
    // sqrt(-2 ln u) = sqrt(-2 ln 2 log2 u)
    const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(c[0])));
    const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(c[2])));
    const float a0 = u01(c[1]), a1 = u01(c[3]);
    const float cand[4] = {r0 * __builtin_amdgcn_cosf(a0), r0 * __builtin_amdgcn_sinf(a0), r1 * __builtin_amdgcn_cosf(a1),
                           r1 * __builtin_amdgcn_sinf(a1)};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (((pending >> k) & 1u) && (bound <= 0.f || fabsf(cand[k]) <= bound)) {
        z[k] = cand[k];
        pending &= ~(1u << k);
      }
    }
  }
  // an element keeps z = 0 only after 1024 rejected candidates in a row: (1 - 0.0797)^1024 = 1e-37 at the tightest bound the entry
  // point accepts (0.1: acceptance probability 0.0797), 3e-172 at bound 0.2; the reference loops until every element is accepted
  // (diffusion.py:378-388)
}


// what a kernel needs to draw its own noise: Philox key, stream, top-p bound, the device step counter, the global number of its first group
struct StepRng { uint32_t seed_lo, seed_hi, stream_id; float bound; const uint32_t* step; uint64_t first_group; };

// models/diffusion.py:319-347, :390-397 (p_sample) and :729-757 (ddim_sample) on 4 elements: x0 comes in as the (rounded) prediction
// and goes out clipped; out = mean + sigma * noise (anchoring is the caller's)
template <bool DDIM>
__device__ __forceinline__ void step_update4(f32x4& x0, const f32x4& xt, const f32x4& nz, const mh_step_coef& c, int clip, f32x4& mean, f32x4& sample) {
#pragma clang fp contract(off)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float v = x0[e];
    if (clip) v = fminf(fmaxf(v, -1.0f), 1.0f);
    x0[e] = v;
    if constexpr (DDIM) {
      const float eps = (c.recip * xt[e] - v) / c.recipm1;
      mean[e] = v * c.sqrt_abp + c.dir * eps;
      sample[e] = mean[e] + c.sigma * nz[e];
    } else {
      mean[e] = c.coef1 * v + c.coef2 * xt[e];
      sample[e] = mean[e] + c.sigma * nz[e];
    }
  }
}
