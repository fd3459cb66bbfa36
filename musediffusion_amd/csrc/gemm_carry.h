// gemm_carry.h - included by gemm.hip inside its anonymous namespace; libmusehip_dbg.so (-DMH_ABLATE) only.
//
// Round-6 experiment (round-5 verdict, item 5): FFN1's dense + bias + GELU with the PREVIOUS tile's epilogue carried under the next tile's
// K loop.  The product kernel (gemm_big_kernel<CfgStd, 0, GELU>) runs a tile's 16 K-steps and then its 16 epilogue groups, two blocks per CU:
// 20 - 24 of its 89 us per full batch are the GELU's vector instructions.  This kernel is the geometry the
// verdict named:
//   * the same 256 x 128 block tile and K32-panel buffer-DMA ring, but ONE block of four waves per CU - one wave per SIMD, 512 registers:
//     two accumulator sets of 128 (the tile being multiplied, the tile being finished), two fragment sets (the next K-step's fragments are
//     read from LDS under this K-step's MFMAs);
//   * mfma_f32_32x32x16_bf16 (8 passes: the vector issue port is free for 24 of its 32 cycles; the product's 16x16x32 leaves 8 of 16:
//     profiles/r03_mfma_fillers.txt), W rows dealt to the MFMA rows so that a lane owns 16 consecutive output columns of one token row;
//   * the K loop fully unrolled (K = 512: 16 K-steps), epilogue group kt of the previous tile (8 values per lane: bias, GELU, convert, one
//     16-byte store) placed into K-step kt by `sched_group_barrier`;
//   * a block walks m-tiles of ONE column strip: the bias stays in registers, the W tile in L2, and the stage ring runs across tiles without
//     draining (the next tile's first stages are issued from this tile's last K-steps);
//   * the stage waits count the carried stores as younger operations (one per K-step).
// Result and ISA: profiles/r06_ffn1_carry.txt.  FULL tiles, panel operands and panel output only.

// NST: stages of the ring (one block per CU: up to six fit the LDS); VAR bit 0: timing-only, no carried epilogue at all (the main loop of
// this geometry alone; outputs are not written)
template <int NK, int NST, int VAR>
__global__ __launch_bounds__(256, (VAR & 4) ? 2 : 1) void gemm_carry_kernel(const GemmArgs g, int tiles_n, int tpb) {
  using C = CfgStd;
  constexpr int STAGE = C::STAGE;
  constexpr bool NOEPI = (VAR & 1) != 0;
  constexpr bool SEQ = (VAR & 2) != 0;     // the tile's epilogue after its own K loop (nothing carried): this geometry's main loop with the product's order
  static_assert(NK % 2 == 0 && NK >= NST && NST >= 3 && (NST - 2) * C::PIECES + NST - 1 <= 63, "the fragment buffers alternate per K-step across tiles; vmcnt is 6 bits");
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE + C::BN * 4];   // the ring, then the strip's bias
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  constexpr int GSW[4] = {0, 2, 3, 1};
  // block -> (column strip, run of m-tiles): the 32 blocks of an XCD (blockIdx % 8) cover a band of m-tiles x every column strip
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int groups = ((int)gridDim.x >> 3) / tiles_n;
  const int n_tile = jb % tiles_n, grp = jb / tiles_n;
  const int m_tile0 = xcd * (tpb * groups) + grp * tpb;
  const int n0 = n_tile * C::BN;

  // stage DMA (gemm_big_kernel's BufDma form)
  int va, vw;
  {
    const int rl = lane >> 2, lc = (lane & 3) ^ GSW[(rl >> 2) & 3];
    va = ((wave * C::PA) * 16 + rl) * 64 + lc * 16;
    vw = ((wave * C::PW) * 16 + rl) * 64 + lc * 16;
  }
  const int ka = (int)(g.lda * 64), kw = (int)(g.ldw * 64);
  const int ldsA0 = wave * C::PA * 1024, ldsW0 = C::BM * 64 + wave * C::PW * 1024;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(g.W)) + (int64_t)n0 * 64, 0,
      (int)((int64_t)(g.K / 32 - 1) * g.ldw * 64 + ((int64_t)g.N - n0) * 64), 0x00020000);
  auto a_rsrc = [&](int m_tile) {
    const int64_t tm0 = (int64_t)m_tile * C::BM;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(g.A)) + tm0 * 64, 0,
                                             (int)((int64_t)(g.K / 32 - 1) * g.lda * 64 + (g.M - tm0) * 64), 0x00020000);
  };
  auto issue = [&](const __amdgpu_buffer_rsrc_t& ra, int slot, int k) {
    char* base = smem + slot;
    static_for<0, C::PA>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(base + ldsA0), 16, va, k * ka, j * 1024, 0);
    });
    static_for<0, C::PW>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(base + ldsW0), 16, vw, k * kw, j * 1024, 0);
    });
  };

  // fragments of a K-step: 2 k-halves x (4 token tiles + 2 column tiles) of 32 rows; lane = (row l31, 16-byte chunk 2 kk + h)
  // (MFMA row l31 of a column tile holds output column wc: see the epilogue)
  constexpr bool PAIR = (VAR & 32) != 0;
  static_assert(!PAIR || NST == 3, "PAIR: the counted waits assume one K-step of every two carries the pair's two stores");
  const int wc = PAIR ? 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3)
                      : 16 * (l31 >> 4) + 8 * ((l31 >> 2) & 1) + 4 * ((l31 >> 3) & 1) + (l31 & 3);
  const int a_base = (wm * 128 + l31) * 64;
  const int b_base = C::BM * 64 + (wn * 64 + wc) * 64;
  int swa[2], swb[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    swa[kk] = ((2 * kk + h) ^ GSW[(l31 >> 2) & 3]) << 4;
    swb[kk] = ((2 * kk + h) ^ GSW[(wc >> 2) & 3]) << 4;
  }
  bf16x8 fa[2][2][4], fb[2][2][2];
  auto load = [&](auto bufc, int slot) {
    constexpr int buf = decltype(bufc)::value;
    const char* st = smem + slot;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) fb[buf][kk][jj] = *reinterpret_cast<const bf16x8*>(st + b_base + jj * 2048 + swb[kk]);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[buf][kk][i] = *reinterpret_cast<const bf16x8*>(st + a_base + i * 2048 + swa[kk]);
    }
  };

  // a lane's outputs: token row m0 + 128 wm + 32 i + l31, columns n0 + 64 wn + 32 jj + (accumulator register 8 s + e ->) 16 s + 8 h + e,
  // or, PAIR (VAR bit 5), 16 h + 8 s + e.  Without PAIR a store instruction writes 32 bytes of each of 32 rows (two lanes hold a token row
  // of a 32-column tile): half-written 64-byte panel rows, which cost the streaming stores half their bandwidth (55 us for FFN1's 134 MB).
  // PAIR: the two halves (s = 0, 1) of a (token tile, column tile) are exchanged between lanes r and r + 16 by v_permlane16_swap - one
  // register then holds all four 16-byte pieces of tokens 0 - 15 (lane L: token L & 15, piece L >> 4 = s + 2 h), the other those of
  // tokens 16 - 31 - and every store instruction writes 1 KiB of contiguous memory, as the product's epilogue does.
  // the strip's bias lives in LDS (read back per group: no registers across the K loops, no vector-memory operation among the counted ones)
  float* const sbias = reinterpret_cast<float*>(smem + NST * STAGE);
  if (tid < C::BN) sbias[tid] = g.bias[n0 + tid];
  __syncthreads();
  bf16* const out = reinterpret_cast<bf16*>(g.out);
  auto out_ptr = [&](int m_tile, int jj) -> bf16* {
    bf16* strip = out + ((int64_t)(((n0 + wn * 64) >> 5) + jj) * g.ldo + (int64_t)m_tile * C::BM + wm * 128) * 32;
    return PAIR ? strip + (lane & 15) * 32 + (lane >> 4) * 8 : strip + l31 * 32 + 8 * h;
  };
  bf16* pprev[2] = {nullptr, nullptr};   // the carried tile's two column-tile bases
  // bias + GELU + convert of group gi = 4 i + 2 jj + s (8 values per lane)
  auto half = [&](auto gc, const f32x16 (&prv)[4][2]) -> bf16x8 {
    constexpr int gi = decltype(gc)::value, i = gi >> 2, jj = (gi >> 1) & 1, s = gi & 1;
    float v[8], bv[8];
    load8(sbias + wn * 64 + 32 * jj + (PAIR ? 16 * h + 8 * s : 16 * s + 8 * h), bv);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = prv[i][jj][8 * s + e] + bv[e];
    if constexpr ((VAR & 8) == 0) gelu_erf_fast8(v);   // (bit 3: timing-only, bias + convert + store without the activation)
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (bf16)v[e];
    return r;
  };
  auto put = [&](bf16* p, const bf16x8& r) {
    f32x4 raw;
    __builtin_memcpy(&raw, &r, 16);
    if constexpr ((VAR & 16) != 0) *reinterpret_cast<f32x4*>(p) = raw;   // (bit 4: ordinary instead of streaming stores)
    else __builtin_nontemporal_store(raw, reinterpret_cast<f32x4*>(p));
  };
  bf16x8 xkeep;   // PAIR: the s = 0 half of the pair in flight
  auto group = [&](auto gc, const f32x16 (&prv)[4][2]) {
    constexpr int gi = decltype(gc)::value, i = gi >> 2, jj = (gi >> 1) & 1, s = gi & 1;
    if constexpr (!PAIR) {
      put(pprev[jj] + i * 1024 + 16 * s, half(gc, prv));
    } else if constexpr (s == 0) {
      xkeep = half(gc, prv);
    } else {
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      const bf16x8 y = half(gc, prv);
      u32x4 xa, ya;
      __builtin_memcpy(&xa, &xkeep, 16);
      __builtin_memcpy(&ya, &y, 16);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const auto r = __builtin_amdgcn_permlane16_swap(xa[d], ya[d], false, false);
        xa[d] = r[0]; ya[d] = r[1];
      }
      bf16x8 x2, y2;
      __builtin_memcpy(&x2, &xa, 16);
      __builtin_memcpy(&y2, &ya, 16);
      put(pprev[jj] + i * 1024, x2);            // tokens 32 i + 0..15, all 64 bytes of each
      put(pprev[jj] + i * 1024 + 512, y2);      // tokens 32 i + 16..31
    }
  };

  int so[NST];                                 // ring slot of stage (kt % NST) of the current tile
#pragma unroll
  for (int k = 0; k < NST; ++k) so[k] = k * STAGE;
  __amdgpu_buffer_rsrc_t ra_cur = a_rsrc(m_tile0), ra_next = ra_cur;

  // one K-step.  MODE 0: the block's first tile (nothing carried), 1: a middle tile, 2: the block's last tile (no stages beyond it)
  auto kstep = [&](auto ktc, auto modec, f32x16 (&cur)[4][2], const f32x16 (&prv)[4][2]) {
    constexpr int kt = decltype(ktc)::value, MODE = decltype(modec)::value, buf = kt & 1;
    constexpr bool need_next = kt + 1 < NK || MODE != 2;    // a stage kt + 1 exists
    // stages kt + 2 .. kt + NST - 1 are in flight behind it (the block's last tile: those that exist)
    constexpr int young = MODE != 2 ? NST - 2 : (NK - kt - 2 < 0 ? 0 : (NK - kt - 2 < NST - 2 ? NK - kt - 2 : NST - 2));
    constexpr bool dman = kt + NST < NK || MODE != 2;       // stage kt + NST is issued here
    // operations younger than stage kt + 1 (issued NST - 1 K-steps ago): the stages above and the carried stores of the K-steps since.  The
    // first NST - 1 K-steps of a tile do not count the stores (the previous tile may have had none): they wait for them - never wrong, rarely late
    // (SEQ: a tile's 16 stores leave between its last K-step and the next tile's first: younger than the stages issued before them)
    constexpr int nst = NOEPI ? 0 : SEQ ? ((MODE != 0 && kt + 1 < NST) ? 16 : 0) : ((MODE != 0 && kt >= NST - 1) ? NST - 1 : 0);
    if constexpr (need_next) {
      wait_vmcnt<young * C::PIECES + nst>();
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's reads of stage kt (issued a K-step ago) are done: its slot may be refilled
      __builtin_amdgcn_s_barrier();
      if constexpr (dman) {
        if constexpr (kt + NST < NK) issue(ra_cur, so[kt % NST], kt + NST);
        else issue(ra_next, so[kt % NST], kt + NST - NK);
      }
      load(std::integral_constant<int, buf ^ 1>{}, so[(kt + 1) % NST]);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          if (kt == 0 && kk == 0) {
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            cur[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[buf][kk][jj], fa[buf][kk][i], z, 0, 0, 0);
          } else {
            cur[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[buf][kk][jj], fa[buf][kk][i], cur[i][jj], 0, 0, 0);
          }
        }
    if constexpr (!NOEPI && !SEQ && MODE != 0 && kt < 16) group(std::integral_constant<int, kt>{}, prv);
    // the K-step's issue order: behind every MFMA one stage piece (the first six), one fragment read (the first twelve) and a share of
    // the carried group's vector instructions; the group's store last
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (m < C::PIECES) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      if (m < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, MH_CARRY_VALU, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto tile = [&](auto modec, f32x16 (&cur)[4][2], const f32x16 (&prv)[4][2], int t) {
    constexpr int MODE = decltype(modec)::value;
    if constexpr (MODE != 2) ra_next = a_rsrc(m_tile0 + t + 1);
    static_for<0, NK>([&](auto ktc) { kstep(ktc, modec, cur, prv); });
    // the next tile: its stage k sits where this tile's stage k + NK went
    int sn[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) sn[k] = so[(NK + k) % NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) so[k] = sn[k];
    ra_cur = ra_next;
    pprev[0] = out_ptr(m_tile0 + t, 0);
    pprev[1] = out_ptr(m_tile0 + t, 1);
    if constexpr (SEQ && !NOEPI) {
      static_for<0, 16>([&](auto gc) { group(gc, cur); });
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  f32x16 accA[4][2], accB[4][2];
  static_for<0, NST>([&](auto kc) { issue(ra_cur, so[decltype(kc)::value], decltype(kc)::value); });
  wait_vmcnt<(NST - 1) * C::PIECES>();     // stage 0 landed
  __builtin_amdgcn_s_barrier();
  load(std::integral_constant<int, 0>{}, so[0]);
  int t = 1;
  if constexpr (SEQ) {
    tile(std::integral_constant<int, 0>{}, accA, accA, 0);
    for (; t + 1 < tpb; ++t) tile(std::integral_constant<int, 1>{}, accA, accA, t);
    tile(std::integral_constant<int, 2>{}, accA, accA, t);
    if constexpr (NOEPI) {   // (keep the accumulators alive)
      float sacc = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc += accA[i][jj][r];
      if (sacc == 12345.678f) out[0] = (bf16)sacc;
    }
    return;
  }
  tile(std::integral_constant<int, 0>{}, accA, accB, 0);
  for (; t + 1 < tpb; t += 2) {
    tile(std::integral_constant<int, 1>{}, accB, accA, t);
    tile(std::integral_constant<int, 1>{}, accA, accB, t + 1);
  }
  tile(std::integral_constant<int, 2>{}, accB, accA, t);
  if constexpr (!NOEPI) static_for<0, 16>([&](auto gc) { group(gc, accB); });
  else {   // (keep the accumulators alive)
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc += accA[i][jj][r] + accB[i][jj][r];
    if (sacc == 12345.678f) out[0] = (bf16)sacc;
  }
}

int launch_carry(const GemmArgs& g0, int variant, hipStream_t s) {
  GemmArgs g = g0;
  using C = CfgStd;
  MH_CHECK_ARG(g.K == 512, "gemm_ffn1_carry: K = %d (the unrolled K loop is built for 512)", g.K);
  MH_CHECK_ARG(g.M % C::BM == 0 && g.N % C::BN == 0, "gemm_ffn1_carry: full tiles only (M %% 256, N %% 128)");
  MH_CHECK_ARG(g.a_panel && g.w_panel && g.o_panel && g.bias && !g.residual && !g.out_f32, "gemm_ffn1_carry: panel operands, panel bf16 output, bias");
  MH_CHECK_ARG((int64_t)(g.K / 32) * g.lda * 64 < (1ll << 31) && (int64_t)(g.K / 32) * g.ldw * 64 < (1ll << 31), "gemm_ffn1_carry: operand too large for a 32-bit descriptor");
  const int tiles_m = (int)(g.M / C::BM), tiles_n = g.N / C::BN;
  // the largest grid of (8 XCDs) x (column strips) x (groups of m-tile runs) that fits the chip with an even run of >= 2 tiles per block
  const int cus = device_cus();
  MH_CHECK_ARG(tiles_m % 8 == 0, "gemm_ffn1_carry: M / 256 must be a multiple of 8");
  const int mx = tiles_m / 8;
  int groups = 0;
  for (int gq = mx / 2; gq >= 1; --gq)
    if (mx % gq == 0 && (mx / gq) % 2 == 0 && 8 * tiles_n * gq <= cus * (variant == 7 || variant == 8 || variant == 16 ? 2 : 1)) { groups = gq; break; }
  MH_CHECK_ARG(groups > 0, "gemm_ffn1_carry: no grid for M = %lld N = %d on %d CUs", (long long)g.M, g.N, cus);
  const int tpb = mx / groups;
  mh_prof_note("carry tile=256x128 M=%lld N=%d K=%d grid=%d tpb=%d", (long long)g.M, g.N, g.K, 8 * tiles_n * groups, tpb);
  const dim3 grid(8 * tiles_n * groups), block(256);
  switch (variant) {
    case 0: MH_LAUNCH((gemm_carry_kernel<16, 3, 0>), grid, block, 0, s, g, tiles_n, tpb); break;
    case 1: MH_LAUNCH((gemm_carry_kernel<16, 3, 1>), grid, block, 0, s, g, tiles_n, tpb); break;
    case 2: MH_LAUNCH((gemm_carry_kernel<16, 6, 0>), grid, block, 0, s, g, tiles_n, tpb); break;
    case 5: MH_LAUNCH((gemm_carry_kernel<16, 3, 2>), grid, block, 0, s, g, tiles_n, tpb); break;
    case 7: MH_LAUNCH((gemm_carry_kernel<16, 3, 6>), grid, block, 0, s, g, tiles_n, tpb); break;   // two blocks per CU, epilogue after its tile
    case 8: MH_LAUNCH((gemm_carry_kernel<16, 3, 7>), grid, block, 0, s, g, tiles_n, tpb); break;   // ... its main loop alone
    case 11: MH_LAUNCH((gemm_carry_kernel<16, 3, 8>), grid, block, 0, s, g, tiles_n, tpb); break;  // carried, no GELU
    case 12: MH_LAUNCH((gemm_carry_kernel<16, 3, 16>), grid, block, 0, s, g, tiles_n, tpb); break; // carried, ordinary stores
    case 14: MH_LAUNCH((gemm_carry_kernel<16, 3, 32>), grid, block, 0, s, g, tiles_n, tpb); break; // carried, paired full-row streaming stores
    case 16: MH_LAUNCH((gemm_carry_kernel<16, 3, 38>), grid, block, 0, s, g, tiles_n, tpb); break; // ... two blocks per CU
    case 17: MH_LAUNCH((gemm_carry_kernel<16, 3, 40>), grid, block, 0, s, g, tiles_n, tpb); break; // carried, paired stores, no GELU
    default: mh_set_error("gemm_ffn1_carry: variant %d is not built (0 1 2 5 7 8 11 12 14 16 17; the others are in the history of this file)", variant); return MH_ERR_INVALID;
  }
  MH_CHECK_LAUNCH();
  return MH_OK;
}
