// gemm_strip.h - included by gemm.hip inside its anonymous namespace (both builds).
//
// Round 6: dense + bias + GELU of K32-panel operands into a K32-panel bf16 output - the sampler's FFN1 - as a "column strip" kernel.  It came
// out of the carried-epilogue experiment (gemm_carry.h, profiles/r06_ffn1_carry.txt): carrying a tile's epilogue under the next tile's K loop
// lost, but the MAIN LOOP built for it runs FFN1's K loops in 34 us per full batch where gemm_big_kernel's needs 62, and with the product's
// order of work (a tile's epilogue after its own K loop, two blocks per CU) the whole launch takes 72 - 76 us against 82 - 85 - bit-identical
// outputs.  (Later the same round the 256 x 128 tile's epilogue got its form fixed at compile time - it had been 25.6 thousand instructions -
// and runs this launch in 77.7 - 78.7 us: the strip kernel's own margin is 3 % of the launch, 0.3 % of the sampler step.)
// What differs from gemm_big_kernel<CfgStd, 0, GELU>:
//   * mfma_f32_32x32x16_bf16 on a 128 x 64 wave tile (4 x 2 MFMA tiles, 128 accumulators): half the MFMA instructions per flop, each holding
//     the wave's issue port for 8 of its 32 cycles (16x16x32: 8 of 16) - fragment reads, stage DMA and waits issue in the other 24;
//   * two fragment register sets: the next K-step's 12 ds_read_b128 are issued at the head of this K-step, under its MFMAs;
//   * the K loop is unrolled over its K-steps (K = 512: 16), every wait count and ring slot a compile-time constant;
//   * a block walks a run of m-tiles down a 128-column strip (and may cross into the next strip once: all blocks get equal runs): the bias of
//     its two strips sits in LDS, a strip's W tile stays in L2, and the three-stage ring never drains - the next tile's first stages are issued
//     from this tile's last K-steps and land under its epilogue, whose 16 stores the next tile's first waits count as younger operations;
//   * W rows are dealt to the MFMA rows so that accumulator registers 8 s .. 8 s + 7 of lane half h hold output columns 16 h + 8 s ..; the
//     two 8-column halves (s = 0, 1) of a (32-token, 32-column) tile are then exchanged between lanes r and r + 16 by v_permlane16_swap_b32:
//     one register set ends up with all four 16-byte pieces of tokens 0 - 15 (lane L: token L & 15, piece L >> 4), the other with tokens
//     16 - 31, and every store instruction writes 1 KiB of contiguous memory.  (Without the exchange a store writes 32 bytes of each of 32
//     panel rows: half-written 64-byte rows cost the streaming stores half their bandwidth - 55 us for FFN1's 134 MB.)
// FULL tiles only (M % 256 == 0, N % 128 == 0, K == 32 NK): everything else stays on gemm_big_kernel (launch<0> decides).

template <int NK, int ACT>
__global__ __launch_bounds__(256, 2) void gemm_strip_kernel(const GemmArgs g, int tiles_m, int tiles_n, int nbands) {
  using C = CfgStd;
  constexpr int NST = 3, STAGE = C::STAGE;
  static_assert(NK % 2 == 0 && NK >= NST, "the fragment buffers alternate per K-step across tiles");
  static_assert((NST - 2) * C::PIECES + 16 <= 63, "vmcnt is a 6-bit counter");
  // the ring, then the bias of the block's two strips
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE + 2 * C::BN * 4];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  constexpr int GSW[4] = {0, 2, 3, 1};
  // block -> a run of tiles.  The m-tiles are cut into nbands bands (8 = one per XCD: blockIdx % 8 under round-robin dispatch, so that an A
  // tile is fetched into ONE L2 and the W strips into all eight); a band's tiles are numbered strip-major (T = strip x band_m + m) and dealt
  // to the band's blocks in equal runs (they differ by at most one tile).  A run is at least two tiles and at most band_m: it walks down a
  // strip and may cross into the NEXT one once (launch_strip sees to both) - the block keeps the vectors of two strips in LDS
  const int band = nbands > 1 ? (int)(blockIdx.x % nbands) : 0, jb = nbands > 1 ? (int)(blockIdx.x / nbands) : (int)blockIdx.x;
  const int band_m = tiles_m / nbands, band_blocks = (int)gridDim.x / nbands, band_tiles = band_m * tiles_n;
  const int T0 = (int)((int64_t)jb * band_tiles / band_blocks), T1 = (int)((int64_t)(jb + 1) * band_tiles / band_blocks);
  const int strip0 = T0 / band_m;
  auto tile_m = [&](int T) __attribute__((always_inline)) { return band * band_m + T % band_m; };

  // stage DMA (gemm_big_kernel's BufDma form: tile base in a descriptor, K step as the scalar offset, a wave's pieces as immediates)
  int va, vw;
  {
    const int rl = lane >> 2, lc = (lane & 3) ^ GSW[(rl >> 2) & 3];
    va = ((wave * C::PA) * 16 + rl) * 64 + lc * 16;
    vw = ((wave * C::PW) * 16 + rl) * 64 + lc * 16;
  }
  const int ka = (int)(g.lda * 64), kw = (int)(g.ldw * 64);
  const int ldsA0 = wave * C::PA * 1024, ldsW0 = C::BM * 64 + wave * C::PW * 1024;
  auto w_rsrc = [&](int strip) __attribute__((always_inline)) {
    const int64_t tn0 = (int64_t)strip * C::BN;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(g.W)) + tn0 * 64, 0,
                                             (int)((int64_t)(g.K / 32 - 1) * g.ldw * 64 + ((int64_t)g.N - tn0) * 64), 0x00020000);
  };
  auto a_rsrc = [&](int m_tile) __attribute__((always_inline)) {
    const int64_t tm0 = (int64_t)m_tile * C::BM;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(g.A)) + tm0 * 64, 0,
                                             (int)((int64_t)(g.K / 32 - 1) * g.lda * 64 + (g.M - tm0) * 64), 0x00020000);
  };
  auto issue = [&](const __amdgpu_buffer_rsrc_t& ra, const __amdgpu_buffer_rsrc_t& rw, int slot, int k) __attribute__((always_inline)) {
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

  // fragments of a K-step: 2 k-halves x (4 token tiles + 2 column tiles) of 32 rows; lane = (row l31, 16-byte chunk 2 kk + h).  MFMA row l31
  // of a column tile reads W row wc (the dealing described above): accumulator register 8 s + e of lane half h = column 16 h + 8 s + e
  const int wc = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
  const int a_base = (wm * 128 + l31) * 64;
  const int b_base = C::BM * 64 + (wn * 64 + wc) * 64;
  int swa[2], swb[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    swa[kk] = ((2 * kk + h) ^ GSW[(l31 >> 2) & 3]) << 4;
    swb[kk] = ((2 * kk + h) ^ GSW[(wc >> 2) & 3]) << 4;
  }
  bf16x8 fa[2][2][4], fb[2][2][2];
  auto load = [&](auto bufc, int slot) __attribute__((always_inline)) {
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

  // the strip's bias lives in LDS (read back per group: no registers across the K loops, no vector-memory operation among the counted ones)
  float* const sbias = reinterpret_cast<float*>(smem + NST * STAGE);
  {
    const int c = strip0 * C::BN + tid;      // the block's first strip and the one after it: 2 x 128 columns
    if (c < g.N) sbias[tid] = g.bias[c];
  }
  __syncthreads();
  int sb = 0;                                  // offset of the current tile's strip in those vectors (0 or 128)
  int n0 = strip0 * C::BN;                     // ... and its first column
  bf16* const out = reinterpret_cast<bf16*>(g.out);
  // bias + activation + convert of group gi = 4 i + 2 jj + s: the lane's 8 values (i, jj, s)
  auto half = [&](auto gc, const f32x16 (&acc)[4][2]) __attribute__((always_inline)) -> bf16x8 {
    constexpr int gi = decltype(gc)::value, i = gi >> 2, jj = (gi >> 1) & 1, s = gi & 1;
    float v[8], bv[8];
    load8(sbias + sb + wn * 64 + 32 * jj + 16 * h + 8 * s, bv);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = acc[i][jj][8 * s + e] + bv[e];
    if constexpr (ACT == MH_ACT_GELU_ERF) gelu_erf_fast8(v);
    else if constexpr (ACT != MH_ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = apply_act<bf16>(v[e], ACT);
    }
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (bf16)v[e];
    return r;
  };
  auto put = [&](bf16* p, const bf16x8& r) __attribute__((always_inline)) {   // streaming store: the output is large and read once, by the next launch
    f32x4 raw;
    __builtin_memcpy(&raw, &r, 16);
    __builtin_nontemporal_store(raw, reinterpret_cast<f32x4*>(p));
  };
  bf16* pout[2];    // lane L after the exchange: token (L & 15) [+ 16], 16-byte piece L >> 4 of the 64-byte panel row, column tile jj
  bf16x8 xkeep;     // the s = 0 half of the pair in flight
  auto group = [&](auto gc, const f32x16 (&acc)[4][2]) __attribute__((always_inline)) {
    constexpr int gi = decltype(gc)::value, i = gi >> 2, jj = (gi >> 1) & 1, s = gi & 1;
    if constexpr (s == 0) {
      xkeep = half(gc, acc);
    } else {
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      const bf16x8 y = half(gc, acc);
      u32x4 xa, ya;
      __builtin_memcpy(&xa, &xkeep, 16);
      __builtin_memcpy(&ya, &y, 16);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const auto r = __builtin_amdgcn_permlane16_swap(xa[d], ya[d], false, false);   // xa rows 1, 3 <-> ya rows 0, 2 (rows of 16 lanes)
        xa[d] = r[0]; ya[d] = r[1];
      }
      bf16x8 x2, y2;
      __builtin_memcpy(&x2, &xa, 16);
      __builtin_memcpy(&y2, &ya, 16);
      put(pout[jj] + i * 1024, x2);            // tokens 32 i + 0..15, all 64 bytes of each
      put(pout[jj] + i * 1024 + 512, y2);      // tokens 32 i + 16..31
    }
  };
  auto epilogue = [&](int m_tile, const f32x16 (&acc)[4][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
      pout[jj] = out + ((int64_t)(((n0 + wn * 64) >> 5) + jj) * g.ldo + (int64_t)m_tile * C::BM + wm * 128 + (lane & 15)) * 32 + (lane >> 4) * 8;
    static_for<0, 16>([&](auto gc) { group(gc, acc); });
  };

  int so[NST];                                 // ring slot of stage (kt % NST) of the current tile
#pragma unroll
  for (int k = 0; k < NST; ++k) so[k] = k * STAGE;
  __amdgpu_buffer_rsrc_t ra_cur = a_rsrc(tile_m(T0)), ra_next = ra_cur;
  __amdgpu_buffer_rsrc_t rw_cur = w_rsrc(strip0), rw_next = rw_cur;
  f32x16 acc[4][2];

  // one K-step.  PREV: a tile of this block ran before this one (its 16 stores are in flight); NEXT: another follows (its stages are issued)
  auto kstep = [&](auto ktc, auto prevc, auto nextc) __attribute__((always_inline)) {
    constexpr int kt = decltype(ktc)::value, buf = kt & 1;
    constexpr bool PREV = decltype(prevc)::value, NEXT = decltype(nextc)::value;
    constexpr bool need_next = kt + 1 < NK || NEXT;    // a stage kt + 1 exists
    // stages kt + 2 .. kt + NST - 1 are in flight behind it (the block's last tile: those that exist) ...
    constexpr int young = NEXT ? NST - 2 : (NK - kt - 2 < 0 ? 0 : (NK - kt - 2 < NST - 2 ? NK - kt - 2 : NST - 2));
    constexpr bool dman = kt + NST < NK || NEXT;       // stage kt + NST is issued here
    // ... and, where stage kt + 1 was issued before the previous tile's epilogue, that tile's 16 stores
    constexpr int nst = (PREV && kt + 1 < NST) ? 16 : 0;
    if constexpr (need_next) {
      wait_vmcnt<young * C::PIECES + nst>();
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's reads of stage kt (issued a K-step ago) are done: its slot may be refilled
      __builtin_amdgcn_s_barrier();
      if constexpr (dman) {
        if constexpr (kt + NST < NK) issue(ra_cur, rw_cur, so[kt % NST], kt + NST);
        else issue(ra_next, rw_next, so[kt % NST], kt + NST - NK);
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
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[buf][kk][jj], fa[buf][kk][i], z, 0, 0, 0);
          } else {
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[buf][kk][jj], fa[buf][kk][i], acc[i][jj], 0, 0, 0);
          }
        }
    // the K-step's issue order: behind each of the first MFMAs one stage piece and one fragment read
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (m < C::PIECES) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      if (m < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto tile = [&](auto prevc, auto nextc, int T) __attribute__((always_inline)) {
    const int m_tile = tile_m(T);
    sb = (T / band_m - strip0) * C::BN;
    n0 = (T / band_m) * C::BN;
    if constexpr (decltype(nextc)::value) { ra_next = a_rsrc(tile_m(T + 1)); rw_next = w_rsrc((T + 1) / band_m); }
    static_for<0, NK>([&](auto ktc) { kstep(ktc, prevc, nextc); });
    // the next tile: its stage k sits where this tile's stage k + NK went
    int sn[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) sn[k] = so[(NK + k) % NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) so[k] = sn[k];
    ra_cur = ra_next;
    rw_cur = rw_next;
    epilogue(m_tile, acc);
    __builtin_amdgcn_sched_barrier(0);
  };

  static_for<0, NST>([&](auto kc) { issue(ra_cur, rw_cur, so[decltype(kc)::value], decltype(kc)::value); });
  wait_vmcnt<(NST - 1) * C::PIECES>();     // stage 0 landed
  __builtin_amdgcn_s_barrier();
  load(std::integral_constant<int, 0>{}, so[0]);
  using T = std::true_type;
  using F = std::false_type;
  // (launch_strip gives every block at least two tiles)
  tile(F{}, T{}, T0);
  int t = T0 + 1;
  for (; t + 1 < T1; ++t) tile(T{}, T{}, t);
  tile(T{}, F{}, t);
}

// the shapes the strip kernel serves (launch<0> asks): dense + bias + GELU of K32 panels into a K32-panel bf16 output, K = 512, full tiles,
// at least two m-tiles
bool strip_ok(const GemmArgs& g) {
  using C = CfgStd;
  return g.a_panel && g.w_panel && g.o_panel && !g.out_f32 && !g.residual && !g.pre_out && !g.q && !g.ln_gamma && !g.drop.thr && !g.act_grad &&
         !g.d.a_stats && !g.d.r_stats && !g.d.o_stats && !(g.dbg & 127) && g.act == MH_ACT_GELU_ERF && g.bias && g.K == 512 &&
         g.M >= 2 * C::BM && g.M % C::BM == 0 && g.N % C::BN == 0 && g.sA == 0 && g.sW == 0 && g.sO == 0 &&
         (int64_t)(g.K / 32) * g.lda * 64 < (1ll << 31) && (int64_t)(g.K / 32) * g.ldw * 64 < (1ll << 31);
}

int launch_strip(const GemmArgs& g, hipStream_t s) {
  using C = CfgStd;
  const int tiles_m = (int)(g.M / C::BM), tiles_n = g.N / C::BN;
  // two blocks per CU.  Bands: one per XCD where the m-tiles divide by 8, else one; a band's blocks get equal runs of its tiles, a run
  // between 2 tiles (the kernel's first / last tile forms) and band_m (it crosses at most one strip boundary)
  const int slots = 2 * device_cus();
  int nbands = (tiles_m % 8 == 0 && slots % 8 == 0) ? 8 : 1;
  int band_m = tiles_m / nbands;
  if (band_m < 2) { nbands = 1; band_m = tiles_m; }
  const int band_tiles = band_m * tiles_n;
  int blocks = slots / nbands;                                  // per band
  if (blocks > band_tiles / 2) blocks = band_tiles / 2;
  const int least = (band_tiles + band_m - 1) / band_m;        // (a run of at most band_m tiles)
  if (blocks < least) blocks = least;
  mh_prof_note("strip tile=256x128 act=%d M=%lld N=%d K=%d grid=%d bands=%d", g.act, (long long)g.M, g.N, g.K, blocks * nbands, nbands);
  MH_LAUNCH((gemm_strip_kernel<16, MH_ACT_GELU_ERF>), dim3(blocks * nbands), dim3(256), 0, s, g, tiles_m, tiles_n, nbands);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
