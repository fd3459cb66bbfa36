// gemm_rowln.h - included by gemm.hip inside its anonymous namespace (both builds).
//
// Round 6: out = LayerNorm(A W^T + bias + residual) over complete rows of N = 512 (HF BertSelfOutput / BertOutput: the sampler's attention-
// output and FFN-output dense, 40 % of its step) on the column-strip kernel's main loop (gemm_strip.h): gemm_big_kernel<128x512pp, EPI 3>'s
// block tile - 128 rows x all 512 columns, eight waves, one block per CU, three-stage buffer-DMA ring - with
//   * mfma_f32_32x32x16_bf16 on a 64 x 128 wave tile (2 x 4 MFMA tiles, 128 accumulators): half the MFMA instructions per flop;
//   * two fragment register sets: the next K-step's 12 ds_read_b128 are issued at the head of this K-step, under its MFMAs;
//   * the K loop in pairs of K-steps (fragment set parity), ring slots rotated in scalar registers, the last four K-steps peeled: every wait
//     count is a compile-time constant;
//   * W rows dealt to the MFMA rows so that accumulator registers 8 s .. 8 s + 7 of lane half h hold output columns 16 h + 8 s .. of a
//     32-column group: exactly the columns lanes fg = 2 h (s = 0) and fg = 2 h + 1 (s = 1) of the 16x16x32 layout hold.  The row statistics
//     keep the old kernel's ORDER of additions - a lane's two partial sums are the old lanes' sums, their sum is the old first shuffle level,
//     the exchange with lane + 32 the second, the four column waves fold through LDS as before - so the result is bit-identical;
//   * the residual rows come as 16 register loads per lane right after the K loop (the fragment registers are free then), the output leaves
//     through the v_permlane16_swap exchange: 1 KiB of contiguous panel rows per store instruction.
// K32-panel operands, residual and output; M % 128 == 0; no dropout / pre-LayerNorm output (the training form stays on gemm_big_kernel).

__global__ __launch_bounds__(512, 1) void gemm_rowln_kernel(const GemmArgs g, int nk) {
  constexpr int BM = 128, BN = 512, NST = 3, STAGE = (BM + BN) * 64, PW = 4, PIECES = 1 + PW, WN = 4;
  // the ring, the row-statistics exchange [BM][WN], bias | gamma | beta of the 512 columns
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE + BM * WN * 4 + 3 * BN * 4];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  constexpr int GSW[4] = {0, 2, 3, 1};
  float* const red = reinterpret_cast<float*>(smem + NST * STAGE);
  float* const vecs = red + BM * WN;
  for (int c = tid; c < BN; c += 512) {
    vecs[c] = g.bias[c];
    vecs[BN + c] = g.ln_gamma[c];
    vecs[2 * BN + c] = g.ln_beta[c];
  }
  __syncthreads();

  // stage DMA: one A piece and four consecutive W pieces per wave (16 rows x 64 B each)
  int va, vw;
  {
    const int rl = lane >> 2, lc = (lane & 3) ^ GSW[(rl >> 2) & 3];
    va = (wave * 16 + rl) * 64 + lc * 16;
    vw = ((wave * PW) * 16 + rl) * 64 + lc * 16;
  }
  const int ka = (int)(g.lda * 64), kw = (int)(g.ldw * 64);
  const int ldsA0 = wave * 1024, ldsW0 = BM * 64 + wave * PW * 1024;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(g.W)), 0,
                                                                      (int)((int64_t)(g.K / 32 - 1) * g.ldw * 64 + (int64_t)g.N * 64), 0x00020000);
  // fragments: lane = (row l31 of a 32-row group, 16-byte chunk 2 kk + h); MFMA row l31 of a column group reads W row wc
  const int wc = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
  const int a_base = (wm * 64 + l31) * 64;
  const int b_base = BM * 64 + (wn * 128 + wc) * 64;
  int swa[2], swb[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    swa[kk] = ((2 * kk + h) ^ GSW[(l31 >> 2) & 3]) << 4;
    swb[kk] = ((2 * kk + h) ^ GSW[(wc >> 2) & 3]) << 4;
  }
  bf16x8 fa[2][2][2], fb[2][2][4];
  auto load = [&](auto bufc, int slot) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value;
    const char* st = smem + slot;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) fb[buf][kk][jj] = *reinterpret_cast<const bf16x8*>(st + b_base + jj * 2048 + swb[kk]);
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[buf][kk][i] = *reinterpret_cast<const bf16x8*>(st + a_base + i * 2048 + swa[kk]);
    }
  };
  const bf16* const res = reinterpret_cast<const bf16*>(g.residual);
  bf16* const out = reinterpret_cast<bf16*>(g.out);
  const int tiles = (int)(g.M / BM);

  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t m0 = (int64_t)tile * BM;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(g.A)) + m0 * 64, 0,
                                                                        (int)((int64_t)(g.K / 32 - 1) * g.lda * 64 + (g.M - m0) * 64), 0x00020000);
    auto issue = [&](int slot, int k) __attribute__((always_inline)) {
      char* base = smem + slot;
      // (the full-row tile's A rows are read by exactly one block, once: nt, as gemm_big_kernel's MH_PP_A_AUX)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(base + ldsA0), 16, va, k * ka, 0, MH_PP_A_AUX);
      static_for<0, PW>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(base + ldsW0), 16, vw, k * kw, j * 1024, 0);
      });
    };
    int s0 = 0, s1 = STAGE, s2 = 2 * STAGE;      // ring slots of stages kt, kt + 1, kt + 2
    f32x16 acc[2][4];
    // one K-step: FIRST zeroes the accumulators; DMA: stage kt + 3 exists; YOUNG: stages in flight behind stage kt + 1; NEXT: stage kt + 1 exists
    auto kstep = [&](auto bufc, auto firstc, auto dmac, auto youngc, auto nextc, int kt) __attribute__((always_inline)) {
      constexpr int buf = decltype(bufc)::value, YOUNG = decltype(youngc)::value;
      constexpr bool FIRST = decltype(firstc)::value, DMA = decltype(dmac)::value, NEXT = decltype(nextc)::value;
      if constexpr (NEXT) {
        wait_vmcnt<YOUNG * PIECES>();
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's reads of stage kt are done: its slot may be refilled
        __builtin_amdgcn_s_barrier();
        if constexpr (DMA) issue(s0, kt + 3);
        load(std::integral_constant<int, buf ^ 1>{}, s1);
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[buf][kk][jj], fa[buf][kk][i], (FIRST && kk == 0) ? z : acc[i][jj], 0, 0, 0);
          }
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (m < PIECES) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        if (m < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int t = s0; s0 = s1; s1 = s2; s2 = t;
    };
    using T = std::true_type;
    using F = std::false_type;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    if (tile != (int)blockIdx.x) {   // the ring is reused: every wave must be done with the previous tile's last stage and its epilogue
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
    }
    issue(s0, 0); issue(s1, 1); issue(s2, 2);
    wait_vmcnt<2 * PIECES>();
    __builtin_amdgcn_s_barrier();
    load(I0{}, s0);
    // nk even, >= 6 (rowln_ok): K-steps 0 .. nk - 5 in pairs, then nk - 4 (the last with a stage to issue), and the three tail steps
    kstep(I0{}, T{}, T{}, I1{}, T{}, 0);
    kstep(I1{}, F{}, T{}, I1{}, T{}, 1);
    int kt = 2;
    for (; kt + 4 < nk; kt += 2) {
      kstep(I0{}, F{}, T{}, I1{}, T{}, kt);
      kstep(I1{}, F{}, T{}, I1{}, T{}, kt + 1);
    }
    kstep(I0{}, F{}, T{}, I1{}, T{}, kt);             // nk - 4: issues the last stage (nk - 1)
    kstep(I1{}, F{}, F{}, I1{}, T{}, kt + 1);         // nk - 3: stage nk - 1 in flight behind stage nk - 2
    kt += 2;
    kstep(I0{}, F{}, F{}, I0{}, T{}, kt);             // nk - 2: stage nk - 1 must have landed, nothing behind it
    kstep(I1{}, F{}, F{}, I0{}, F{}, kt + 1);         // nk - 1

    // ---- epilogue: bias + residual, LayerNorm over the row (two-pass statistics in gemm_big_kernel's order of additions)
    const int64_t rowb = m0 + wm * 64 + l31;          // + 32 i
    bf16x8 rr[2][4][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          rr[i][jj][s] = *reinterpret_cast<const bf16x8*>(res + ((int64_t)(wn * 4 + jj) * g.ldr + rowb + 32 * i) * 32 + 16 * h + 8 * s);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float bv[8];
        load8(vecs + wn * 128 + 32 * jj + 16 * h + 8 * s, bv);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[i][jj][8 * s + e] += bv[e];
      }
    float sa[2][2];    // [i][s]: the partial row sums of the old layout's lanes fg = 2 h + s
#pragma unroll
    for (int i = 0; i < 2; ++i) { sa[i][0] = 0.f; sa[i][1] = 0.f; }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float v = acc[i][jj][8 * s + e] + (float)rr[i][jj][s][e];
            acc[i][jj][8 * s + e] = v;
            sa[i][s] += v;
          }
    const float invN = 1.0f / (float)g.N;
    float mean[2], rstd[2];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float v = sa[i][0] + sa[i][1];          // = the old lanes' first shuffle level (fg ^ 1)
        v += __shfl_xor(v, 32, 64);             // ... and the second (fg ^ 2)
        if (h == 0) red[(wm * 64 + 32 * i + l31) * WN + wn] = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WN; ++w) t += red[(wm * 64 + 32 * i + l31) * WN + w];
        if (pass == 0) {
          mean[i] = t * invN;
          float q0 = 0.f, q1 = 0.f;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = acc[i][jj][e] - mean[i]; q0 += d * d; }
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = acc[i][jj][8 + e] - mean[i]; q1 += d * d; }
          }
          sa[i][0] = q0; sa[i][1] = q1;
        } else {
          rstd[i] = 1.0f / sqrtf(t * invN + g.ln_eps);
        }
      }
      __builtin_amdgcn_s_barrier();            // reads done before the second pass overwrites `red`
    }
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      bf16* const po = out + ((int64_t)(wn * 4 + jj) * g.ldo + m0 + wm * 64 + (lane & 15)) * 32 + (lane >> 4) * 8;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        u32x4 xy[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          float gv[8], bt[8];
          load8(vecs + BN + wn * 128 + 32 * jj + 16 * h + 8 * s, gv);
          load8(vecs + 2 * BN + wn * 128 + 32 * jj + 16 * h + 8 * s, bt);
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16)((acc[i][jj][8 * s + e] - mean[i]) * rstd[i] * gv[e] + bt[e]);
          __builtin_memcpy(&xy[s], &o, 16);
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const auto r = __builtin_amdgcn_permlane16_swap(xy[0][d], xy[1][d], false, false);
          xy[0][d] = r[0]; xy[1][d] = r[1];
        }
        // ordinary stores: the next GEMM re-reads these rows (A operand and residual) from L2 / MALL
        *reinterpret_cast<u32x4*>(po + i * 1024) = xy[0];          // tokens 32 i + 0..15
        *reinterpret_cast<u32x4*>(po + i * 1024 + 512) = xy[1];    // tokens 32 i + 16..31
      }
    }
  }
}

bool rowln_ok(const GemmArgs& g) {
  return g.a_panel && g.w_panel && g.o_panel && g.r_panel && g.residual && g.bias && g.ln_gamma && g.ln_beta && !g.out_f32 && !g.pre_out &&
         !g.drop.thr && !g.act_grad && g.act == MH_ACT_NONE && !(g.dbg & 127) && g.N == 512 && g.K % 64 == 0 && g.K >= 192 && g.M % 128 == 0 &&
         g.M > 0 && g.sA == 0 && g.sW == 0 && g.sO == 0 && g.sR == 0 &&
         (int64_t)(g.K / 32) * g.lda * 64 < (1ll << 31) && (int64_t)(g.K / 32) * g.ldw * 64 < (1ll << 31);
}

int launch_rowln(const GemmArgs& g, hipStream_t s) {
  const int tiles = (int)(g.M / 128), cus = device_cus();
  mh_prof_note("rowln tile=128x512 M=%lld N=%d K=%d grid=%d", (long long)g.M, g.N, g.K, tiles < cus ? tiles : cus);
  MH_LAUNCH(gemm_rowln_kernel, dim3(tiles < cus ? tiles : cus), dim3(512), 0, s, g, g.K / 32);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
