// Register-staged MFMA GEMM kernels (128x128 / 128x64 / 64x64 tiles, single + grouped launch, intra-block k split) and their launchers.
// Header: the kernel templates are instantiated by four small translation units (gemm_s128.hip, gemm_s64.hip, gemm_g128.hip, gemm_g64.hip)
// so that the GEMM family compiles in parallel; the dispatch rules and the C entry points live in gemm.hip.
#pragma once
#include "gemm_common.h"

namespace phg {
namespace {



// One output tile over the k-tiles [kt_begin, kt_end).  XCD_REMAP: block_id is a hardware block index of a one-tile-per-block launch
// (re-mapped so that each XCD owns a contiguous run of tiles); otherwise block_id already is the tile index.  splitk: the tile has
// other contributors (raw partial sums: workspace slice split_id, or fp32 atomics into C when there is no workspace).
// KS = 2: intra-block split of the k loop (512 threads): thread group g = threadIdx.x >> 8 multiplies the k-tiles kt_begin + 2i + g
// in its own pair of LDS stage buffers, the write-out sums the two partial tiles.  For the decoder's launches (M = B*T = 960 rows:
// fewer tiles than CUs, so one block per CU and one wave per SIMD) the k loop is a latency chain -- LDS write, barrier, LDS read,
// 4 MFMAs per wave -- and a second wave per SIMD working on the other half of K hides half of it.
template <int BM, int BN, bool TA, bool TB, int PF, int CONV = 0, bool XCD_REMAP = true, int KS = 1>
__device__ __forceinline__ void gemm_tile(const GemmParams& p, const int block_id, const int kt_begin, const int kt_end, const bool splitk,
                                          const int split_id = 0) {
  static_assert(KS == 1 || (KS == 2 && CONV == 0 && PF > 1), "intra-block k split: plain ring kernels only");   // (KS = 4, 1024 threads:
  // built and measured in round 4 -- every M = 960 launch 1 us SLOWER than KS = 2, profiles/r4_probe_ks4.txt: sixteen waves share one
  // block-wide barrier per k-tile and the LDS write/read passes of four groups; the code below stays generic in KS)
  static_assert(CONV == 0 || (CONV == 1 && !TA && !TB) || (CONV == 2 && TA && TB), "conv gather: A of an NN problem or B of a TT problem");
  constexpr int WM = BM / 2, WN = BN / 2;   // wave tile
  constexpr int TM = WM / 32, TN = WN / 32; // 32x32 MFMA tiles per wave
  constexpr int A_BYTES = TA ? TileBytes<BM>::ks : TileBytes<BM>::kc;
  constexpr int B_BYTES = TB ? TileBytes<BN>::ks : TileBytes<BN>::kc;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = A_BYTES + B_BYTES;   // stage s: A image at smem + s*STAGE, B image right behind it

  // XCD-aware tile mapping: hardware places block b on XCD b % 8; give each XCD a contiguous run of tiles.
  int nt = p.tiles_m * p.tiles_n;
  int bid = block_id;
  if constexpr (XCD_REMAP) {
    int q = nt / 8, r = nt % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // L2-aware rasterisation: tiles are walked in groups of GM row-panels, rows fastest.  The ~64 tiles an XCD runs at once
  // then cover ~8 row-panels x 8 column-panels (16 operand panels through its 4 MB L2) instead of 1-3 row-panels x ALL
  // column-panels: PMC showed 3-4x the algorithmic HBM bytes on the wide-N GEMMs (qkv / fc / LM head) with row-major order.
  constexpr int GM = 8;
  const int group_sz = GM * p.tiles_n;
  const int first_m = (bid / group_sz) * GM;
  const int gm = min(GM, p.tiles_m - first_m);
  const int rin = bid % group_sz;
  int tm = first_m + rin % gm, tn = rin / gm;
  int m0 = tm * BM, n0 = tn * BN;
  if (kt_begin >= kt_end) return;

  const int tid = KS > 1 ? (int)(threadIdx.x & 255) : (int)threadIdx.x;
  const int grp = KS > 1 ? (int)(threadIdx.x >> 8) : 0;
  int lane = tid & 63, wave = tid >> 6;
  int wm = wave >> 1, wn = wave & 1;
  char* const smem_g = smem + grp * 2 * STAGE;          // this thread group's two stage buffers
  PH_TL_DECL;
  PH_TL(0);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // Register prefetch ring of depth D: the global loads of k-tiles t+1 .. t+D are in flight while tile t is multiplied
  // (HBM/L2 latency is ~1-2 us, one 128x128x64 tile of MFMA work is ~0.2-0.4 us: depth 1 stalls every iteration).
  // Ring slot s holds tile (t0 + s); slots are indexed with compile-time constants only (static_for) so they stay
  // in VGPRs.  LDS stays double-buffered: one barrier per k-tile.
  constexpr int D = PF;
  u32x4 ra[D][BM * 8 / 256], rb[D][BN * 8 / 256];
  PixRow px[BM * 8 / 256];
  ColTap ct[BN * 8 / 256];
  if constexpr (CONV == 1) {
#pragma unroll
    for (int i = 0; i < BM * 8 / 256; ++i) px[i] = pix_of(p.cv, min(m0 + ((int)(threadIdx.x + 256 * i) >> 3), p.M - 1));
  }
  if constexpr (CONV == 2) coltaps_of<BN>(p.cv, n0, p.N, ct);
  auto gload = [&](int kt, u32x4 (&xa)[BM * 8 / 256], u32x4 (&xb)[BN * 8 / 256]) {
    int k0 = kt * BK;
    // plain ring kernels are only launched when K % BK == 0 (no k predicate); with a gathered operand K = taps * C is any multiple
    // of 8 and the OTHER operand keeps its predicate -- the gather zero-fills k >= K, but 0 x (whatever lies behind the weight row,
    // possibly NaN bit patterns at the end of an allocation) is not 0
    constexpr bool KF = PF > 1 && CONV == 0;
    if constexpr (CONV == 1) load_kc_conv<BM>(p.cv, p.A, px, k0, xa);
    else if (TA) load_ks<BM, KF>(p.A, p.lda, m0, p.M, k0, p.K, xa, tid); else load_kc<BM, KF>(p.A, p.lda, m0, p.M, k0, p.K, xa, tid);
    if constexpr (CONV == 2) load_ks_conv<BN>(p.cv, p.B, ct, k0, p.K, xb);
    else if (TB) load_ks<BN, KF>(p.B, p.ldb, n0, p.N, k0, p.K, xb, tid); else load_kc<BN, KF>(p.B, p.ldb, n0, p.N, k0, p.K, xb, tid);
  };
  auto lstore = [&](int buf, const u32x4 (&xa)[BM * 8 / 256], const u32x4 (&xb)[BN * 8 / 256]) {
    char* sa = smem_g + buf * STAGE;
    char* sb = sa + A_BYTES;
    if (TA) store_ks<BM>(sa, xa, tid); else store_kc<BM>(sa, xa, tid);
    if constexpr (CONV == 2) store_ks_conv<BN>(sb, xb);
    else if (TB) store_ks<BN>(sb, xb, tid); else store_kc<BN>(sb, xb, tid);
  };
  auto compute = [&](int buf) {
    const char* la = smem_g + buf * STAGE;
    const char* lb = la + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 fx[TM], fw[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fx[i] = TA ? frag_ks<BM>(la, wm * WM + i * 32, kk, lane) : frag_kc(la, wm * WM + i * 32, kk, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fw[j] = TB ? frag_ks<BN>(lb, wn * WN + j * 32, kk, lane) : frag_kc(lb, wn * WN + j * 32, kk, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[j], fx[i], acc[i][j], 0, 0, 0);
    }
  };

  // Schedule (per k-tile t, one barrier):   [LDS(cur) = tile t, registers = tile t+1 landed or landing]
  //     write registers -> LDS(cur^1)      (tile t+1; its loads were issued a whole iteration ago)
  //     re-issue the SAME registers <- global tile t+2      (in flight across the barrier)
  //     MFMAs on LDS(cur)
  //     barrier
  // i.e. two tiles of look-ahead with one register set and two LDS buffers; the LDS write pass sits BEFORE this wave's
  // MFMAs, where it overlaps the co-resident block's matrix work instead of trailing its own.
  if constexpr (D == 1) {
    gload(kt_begin, ra[0], rb[0]);
    lstore(0, ra[0], rb[0]);
    if (kt_begin + 1 < kt_end) gload(kt_begin + 1, ra[0], rb[0]);
    __syncthreads();
    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      if (kt + 1 < kt_end) lstore(cur ^ 1, ra[0], rb[0]);
      if (kt + 2 < kt_end) gload(kt + 2, ra[0], rb[0]);
      compute(cur);
      __syncthreads();
      cur ^= 1;
    }
  } else {
    // Ring of D register sets: the loads of tile i+1+D are issued in iteration i and written to LDS in iteration i+D, so a
    // load has D iterations to land (one is not enough when a CU holds a single block, i.e. every GEMM with few tiles).
    // Loads are unconditional (tile index clamped to the last one) and slots are compile-time constants.
    // (KS = 2: group g owns tiles kt_begin + 2i + g; nk = iterations of the block, nkg = tiles of this group -- one less for group 1
    //  when the count is odd: it then runs its last iteration on a clamped reload without multiplying)
    const int nkt = kt_end - kt_begin;
    const int nk = (nkt + KS - 1) / KS, nkg = (nkt - grp + KS - 1) / KS;
    auto gl = [&](int i, u32x4 (&xa)[BM * 8 / 256], u32x4 (&xb)[BN * 8 / 256]) { gload(kt_begin + KS * min(i, nkg - 1) + grp, xa, xb); };
    static_for(std::make_integer_sequence<int, D>{}, [&](auto dd) { gl(decltype(dd)::value, ra[decltype(dd)::value], rb[decltype(dd)::value]); });
    PH_TL(1);
    lstore(0, ra[0], rb[0]);
    gl(D, ra[0], rb[0]);
    __syncthreads();
    PH_TL(2);
    int cur = 0;
    int i0 = 0;
    for (; i0 + D <= nk; i0 += D) {             // full groups: no predicate anywhere (the store after the last tile lands in the
      static_for(std::make_integer_sequence<int, D>{}, [&](auto dd) {      // idle buffer and is never read)
        constexpr int d = decltype(dd)::value, slot = (d + 1) % D;
        lstore(cur ^ 1, ra[slot], rb[slot]);
        gl(i0 + d + 1 + D, ra[slot], rb[slot]);
        if (KS == 1 || i0 + d < nkg) compute(cur);
        __syncthreads();
        cur ^= 1;
      });
    }
    static_for(std::make_integer_sequence<int, D - 1>{}, [&](auto dd) {    // remainder: nk % D tiles
      constexpr int d = decltype(dd)::value, slot = (d + 1) % D;
      if (i0 + d < nk) {
        lstore(cur ^ 1, ra[slot], rb[slot]);
        if (KS == 1 || i0 + d < nkg) compute(cur);
        __syncthreads();
        cur ^= 1;
      }
    });
  }

#ifdef PH_GEMM_DIAG_NOEPI   // diagnostics build (tools/build_variant.py): main loop only, nothing written unless a sentinel hits
  {
    float chk = 0.f;
    static_for(std::make_integer_sequence<int, TM * TN>{}, [&](auto idx) {
      constexpr int i = decltype(idx)::value / TN, j = decltype(idx)::value % TN;
#pragma unroll
      for (int e = 0; e < 16; ++e) chk += acc[i][j][e];
    });
    if (chk == 123456.789f) reinterpret_cast<float*>(p.C)[threadIdx.x] = chk;
    return;
  }
#endif
  // ---- epilogue --------------------------------------------------------------------------------------------------
  // MFMA leaves lane l with C[m = l&31][n = 8g + 4(l>>5) + e]: 32 different rows per store instruction.  Going straight
  // to HBM from that layout costs one cache-line touch per 16 B; instead the fp32 tile is parked in LDS (re-using the
  // stage buffers, 16-B chunks XOR-swizzled by the row so both sides are conflict-free) and re-read row-wise: 32
  // consecutive lanes then own 256 contiguous bytes of one output row for every load/store of the fused epilogue.
  PH_TL(3);
  constexpr int WO_THR = KS > 1 ? 512 : 256;
  const int epi = epi_classify(p, splitk);
  PH_WO_DECL(BM, BN, WO_THR);
  DropCtx dc;
  const bool drop = p.drop_p > 0.0f;
  if (drop) dc = make_drop(p.drop_seed, p.drop_stream, p.drop_p);
  constexpr int CH = BN / 4;                       // 16-B chunks per tile row
  float* cl = reinterpret_cast<float*>(smem) + grp * (BM * BN);     // (KS = 2: one parking area per thread group, summed by the write-out)
  // (the main loop ended with a barrier: nobody reads the stage buffers any more)
  // acc[][] must only ever be indexed with compile-time constants (a runtime index demotes the accumulators to
  // scratch memory and the main loop then spills them every k-tile), so the tile loop is a static_for.
  static_for(std::make_integer_sequence<int, TM * TN * 4>{}, [&](auto idx) {
    constexpr int i = decltype(idx)::value / (TN * 4), j = (decltype(idx)::value / 4) % TN, g = decltype(idx)::value % 4;
    const int ml = wm * WM + i * 32 + (lane & 31);
    const int c = (wn * WN + j * 32 + g * 8 + (lane >> 5) * 4) >> 2;
    f32x4 v = {acc[i][j][g * 4 + 0] * p.alpha, acc[i][j][g * 4 + 1] * p.alpha, acc[i][j][g * 4 + 2] * p.alpha,
               acc[i][j][g * 4 + 3] * p.alpha};
    *reinterpret_cast<f32x4*>(cl + ml * BN + ((c ^ (ml & (CH - 1))) << 2)) = v;
  });
  __syncthreads();
  PH_TL(5);
  // every read of the fused chain for ALL steps of this thread, requested before its first store (epilogue classes, gemm_common.h).  These
  // kernels run two blocks per CU on a 256-register budget: the reads are requested after the accumulators are parked (64 registers
  // freed) and their latency is covered by the co-resident block; the one-block-per-CU 256x128 kernel requests them before parking.
  if constexpr (KS > 1) {
    static_assert(KS * BM * BN * 4 <= 2 * KS * STAGE, "the parked fp32 partial tiles must fit the stage buffers");
    wo_prefetch<BM, BN, WO_THR>(epi, p, m0, n0, PH_WO_ARGS);
    tile_writeout<BM, BN, 512, KS>(PH_TL_ARG epi, p, reinterpret_cast<float*>(smem), m0, n0, splitk, drop, dc, PH_WO_ARGS);
  } else {
    wo_prefetch<BM, BN, WO_THR>(epi, p, m0, n0, PH_WO_ARGS);
    tile_writeout<BM, BN, 256, 1>(PH_TL_ARG epi, p, cl, m0, n0, splitk, drop, dc, PH_WO_ARGS, split_id);
    if (p.col_stats) tile_colstats<BM, BN, 256>(p, cl, m0, n0);
  }
  PH_TL(8);
#ifdef PH_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PH_TL(9);
  PH_TL_FLUSH(block_id, grp, (threadIdx.x & 255) == 0);
#endif
}

template <int BM, int BN, bool TA, bool TB, int PF, int CONV = 0>
__device__ __forceinline__ void gemm_body(const GemmParams& p, const int block_id, const int split_id, const int nsplits) {
  const int kt_begin = split_id * p.k_tiles_per_split;
  const int kt_end = min(kt_begin + p.k_tiles_per_split, (p.K + BK - 1) / BK);
  gemm_tile<BM, BN, TA, TB, PF, CONV, true>(p, block_id, kt_begin, kt_end, nsplits > 1, split_id);
}

// (256, 2): two blocks per CU (64 KB of LDS each) -- the register allocation must stay within 256 per lane
template <int BM, int BN, bool TA, bool TB, int PF, int CONV = 0>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
  gemm_body<BM, BN, TA, TB, PF, CONV>(p, blockIdx.x, blockIdx.z, gridDim.z);
}


// The grid may be SMALLER than the number of tiles (ph_gemm_grouped_bf16's max_blocks): each block then walks tiles
// blockIdx.x, blockIdx.x + gridDim.x, ... -- a background launch that occupies at most max_blocks block slots and leaves the rest
// of the chip to the latency-bound chain on the main stream (deferred weight gradients beside the decoder's backward).
template <int BM, int BN, bool TA, bool TB, int PF, int CONV = 0>
__global__ __launch_bounds__(256, 2) void gemm_grouped_kernel(GroupParams g) {
  const int total = g.tile_start[g.n];
  int i = 0;
  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    while (i + 1 < g.n && t >= g.tile_start[i + 1]) ++i;
    const int local = t - g.tile_start[i], ns = g.nsplit[i];
    if (ns > 1) {                        // grouped AND split over K (long reductions with few output tiles): partial sums -> workspace
      const int ntile = g.p[i].tiles_m * g.p[i].tiles_n;
      gemm_body<BM, BN, TA, TB, PF, CONV>(g.p[i], local % ntile, local / ntile, ns);
    } else {
      gemm_body<BM, BN, TA, TB, PF, CONV>(g.p[i], local, 0, 1);
    }
    __syncthreads();                     // the epilogue's LDS staging area is the next tile's stage buffer
  }
}



// prefetch depth per tile size (VGPR budget: 128x128 tiles hold 32 staging VGPRs per slot, 64x64 tiles 16)
// A/B on MI355X (tools/ab_probe.py, graph replay): depth 2 on 128x128 tiles +2..10 % on the NT / NN layouts (TN: -4 %, kept
// at 1); depth 3 on 64x64 tiles +7..11 % on the one-block-per-CU decoder GEMMs; split-K launches keep depth 1.
#ifndef PH_RING128
#define PH_RING128 2
#endif
#ifndef PH_RING64
#define PH_RING64 3
#endif
#define PF_DEPTH(bm) ((bm) == 128 ? PH_RING128 : PH_RING64)

// 512-thread form with the k loop split between two thread groups (gemm_tile KS = 2); 64x64 tiles, K % 64 == 0, no split-K
template <bool TA, bool TB>
__global__ __launch_bounds__(512) void gemm_ks2_kernel(GemmParams p) {
  gemm_tile<64, 64, TA, TB, PH_RING64, 0, true, 2>(p, blockIdx.x, 0, p.K / BK, false);
}
template <bool TA, bool TB>
int launch_ks2(const GemmParams& p, hipStream_t s) {
  constexpr int smem = 4 * ((TA ? TileBytes<64>::ks : TileBytes<64>::kc) + (TB ? TileBytes<64>::ks : TileBytes<64>::kc));
  PH_SET_SMEM_ONCE((&gemm_ks2_kernel<TA, TB>), smem);
  count_launch(PH_GEMM_CLS_KS2);
  hipLaunchKernelGGL((gemm_ks2_kernel<TA, TB>), dim3(p.tiles_m * p.tiles_n), dim3(512), smem, s, p);
  PH_LAUNCH_CHECK("gemm_ks2_kernel");
  return PH_OK;
}

// (round 5: two further forms of the decoder's M = 960 launches were built, tested and measured neutral-to-negative -- a whole-K-resident 64x64
//  kernel with every operand and write-out read requested in the prologue, and a 64x192-tile kernel for the wide qkv / MLP launches -- the
//  N = 768 launches already sit at the latency of ONE tile, and 32-row wave tiles are LDS-bandwidth bound: profiles/r5_decoder_gemm.txt)

template <int BM, int BN, bool TA, bool TB, int PF, int CONV = 0>
int launch_pf(const GemmParams& p, int splits, hipStream_t s) {
  constexpr int smem_min = 2 * ((TA ? TileBytes<BM>::ks : TileBytes<BM>::kc) + (TB ? TileBytes<BN>::ks : TileBytes<BN>::kc));
  constexpr int smem = smem_min;
  PH_SET_SMEM_ONCE((&gemm_kernel<BM, BN, TA, TB, PF, CONV>), smem);
  count_launch(BM == 64 ? PH_GEMM_CLS_64 : PH_GEMM_CLS_128);
  dim3 grid(p.tiles_m * p.tiles_n, 1, splits);
  hipLaunchKernelGGL((gemm_kernel<BM, BN, TA, TB, PF, CONV>), grid, dim3(256), smem, s, p);
  PH_LAUNCH_CHECK("gemm_kernel");
  return PH_OK;
}

template <int BM, int BN, bool TA, bool TB>
int launch(const GemmParams& p, int splits, hipStream_t s) {
  if constexpr (PF_DEPTH(BM) > 1 && !(BM == 128 && TA)) {
    if (p.K % BK == 0 && splits == 1) return launch_pf<BM, BN, TA, TB, PF_DEPTH(BM)>(p, splits, s);   // the ring needs unpredicated k loads
  }
  return launch_pf<BM, BN, TA, TB, 1>(p, splits, s);
}

template <int BM, int BN>
int dispatch_layout(const GemmParams& p, int ta, int tb, int splits, hipStream_t s) {
  if (p.cv.C > 0) {       // implicit-GEMM weight gradient: B = im2col view (K-strided gather); square tiles only, no prefetch ring
    if constexpr (BM == BN) return launch_pf<BM, BN, true, true, 1, 2>(p, splits, s);
    else return ph_fail(PH_ERR_UNSUPPORTED, "ph_gemm_bf16: conv gather needs square tiles");
  }
  if (!ta && !tb) return launch<BM, BN, false, false>(p, splits, s);
  if (!ta && tb) return launch<BM, BN, false, true>(p, splits, s);
  if (ta && tb) return launch<BM, BN, true, true>(p, splits, s);
  return launch<BM, BN, true, false>(p, splits, s);
}

template <int BM, bool TA, bool TB, int PF, int CONV = 0>
int launch_grouped(const GroupParams& g, int total, int max_blocks, hipStream_t s) {
  constexpr int smem = 2 * ((TA ? TileBytes<BM>::ks : TileBytes<BM>::kc) + (TB ? TileBytes<BM>::ks : TileBytes<BM>::kc));
  PH_SET_SMEM_ONCE((&gemm_grouped_kernel<BM, BM, TA, TB, PF, CONV>), smem);
  count_launch(PH_GEMM_CLS_GROUPED);
  const int grid = (max_blocks > 0 && max_blocks < total) ? max_blocks : total;
  hipLaunchKernelGGL((gemm_grouped_kernel<BM, BM, TA, TB, PF, CONV>), dim3(grid), dim3(256), smem, s, g);
  PH_LAUNCH_CHECK("gemm_grouped_kernel");
  return PH_OK;
}
template <int BM, int PF>
int launch_grouped_layout(const GroupParams& g, int total, int max_blocks, int ta, int tb, hipStream_t s) {
  if (!ta && !tb) return launch_grouped<BM, false, false, PF>(g, total, max_blocks, s);
  if (!ta && tb) return launch_grouped<BM, false, true, PF>(g, total, max_blocks, s);
  if (ta && tb) return launch_grouped<BM, true, true, PF>(g, total, max_blocks, s);
  return launch_grouped<BM, true, false, PF>(g, total, max_blocks, s);
}
// pf: 0 = depth-1 schedule, 1 = prefetch ring; conv: 0 plain, 1 gathered A (NN layout), 2 gathered B (TT layout, depth 1 only)
template <int BM>
int launch_grouped_any(const GroupParams& g, int total, int max_blocks, int ta, int tb, int pf, int conv, hipStream_t s) {
  constexpr int RING = PF_DEPTH(BM);
  if (conv == 2) return launch_grouped<BM, true, true, 1, 2>(g, total, max_blocks, s);
  if (conv == 1) return (pf && RING > 1) ? launch_grouped<BM, false, false, RING, 1>(g, total, max_blocks, s) : launch_grouped<BM, false, false, 1, 1>(g, total, max_blocks, s);
  if (pf && RING > 1) return launch_grouped_layout<BM, RING>(g, total, max_blocks, ta, tb, s);
  return launch_grouped_layout<BM, 1>(g, total, max_blocks, ta, tb, s);
}

}  // namespace
}  // namespace phg
