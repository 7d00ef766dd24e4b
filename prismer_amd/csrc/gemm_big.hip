// 256x128 LDS-DMA GEMM kernels for gfx950 (ping-pong main loop, [N,K] / [K,N] / [K,M] operands, single and grouped persistent launch).
// Split from gemm.hip so that the two halves of the GEMM family compile in parallel; the dispatch rules live in gemm.hip.
#include "gemm_common.h"

namespace phg {

// ---------------------------------------------------------------------------------------------------------------------------
// "Big" kernel for the forward-shaped GEMMs with many rows (both operands K-contiguous, K % 64 == 0):
//   block = 512 threads = 8 waves as 4 (M) x 2 (N), wave tile 64 x 64, block tile 256 x 128, BK = 64;
//   operands go global -> LDS by the LDS-DMA path (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass) into a ring of
//   THREE 48-KB stages: the loads of k-tile t+2 are issued right after the barrier that publishes tile t and stay in flight across
//   the next barrier (counted s_waitcnt vmcnt(6): each thread owns 6 DMA instructions per tile), one raw s_barrier per k-tile.
//   The LDS image is lane-linear per DMA instruction (8 rows x 128 B per wave instruction); the 16-B chunk XOR swizzle the
//   ds_read_b128 fragment reads need is applied on the SOURCE address (lane l fetches chunk (l&7) ^ swz(row) of its row) and
//   again on the read -- the destination stays linear (hardware writes base + lane*16).
//   1 block per CU (144 KB of LDS), 2 waves per SIMD.  Epilogue: the 256x128 fp32 tile is parked in the (drained) ring and
//   written out row-wise by the same fused chain as the 128x128 kernel.
namespace big {
constexpr int NTHR = 512, STAGES = 3;
constexpr int A_BYTES = BM * KC_ROW_BYTES, B_BYTES = BN * KC_ROW_BYTES, STAGE = A_BYTES + B_BYTES;   // 32 KB + 16 KB
constexpr int A_INSTR = BM * 8 / NTHR, B_INSTR = BN * 8 / NTHR;                                     // 4 + 2 DMA instructions / thread / tile
constexpr int SMEM = STAGES * STAGE;                                                                // 147456 B (>= 256*128*4 for the epilogue)

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

// CONV = 1 (round 4, the stems' 3x3 convolutions and their data gradients): the A operand is the im2col VIEW of an NHWC activation
// (gemm_common.h, ConvGather).  LDS-DMA takes a per-lane global address, so the gather is nothing but a different source address per
// 16-B chunk: (tap, channel) of the lane's k chunk once per k-tile, the pixel of each of its 4 rows from loop-invariant per-row state
// (element offset of tap (0,0) + a bit mask of the taps inside the image); a chunk that falls outside the image, or beyond the real K, reads
// the library's ZERO PAGE instead (p.zero16).  K need not be a multiple
// of 64 here (K = 9 * 96 = 864): the chunks of B beyond K read the zero page as well.
template <int VARIANT, bool TA, bool TB, bool XCD_REMAP, int CONV = 0>
__device__ __forceinline__ void big_tile(const GemmParams& p, const int block_id) {
  static_assert(CONV == 0 || (CONV == 1 && !TA && !TB && (VARIANT & 4)), "conv gather: A of a forward-shaped problem, ping-pong loop");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // tile mapping.  Hardware places block b on XCD b % 8, each XCD with its own 4-MB L2.  One-tile-per-block launches (XCD_REMAP) give every
  // XCD WHOLE row panels: XCD x owns tiles_m / 8 (+1) consecutive 256-row panels with ALL their column tiles, so an A panel is fetched
  // by one L2 only (round 4: the round-1 numbering cut the tile list into eight equal runs, which put the 24-tile panel groups of a
  // 198-tile launch astride two XCDs -- FETCH_SIZE 29.6 MB for 14.0 MB of operands, profiles/r4_fetch_write_by_shape_before.txt).
  // The grid is 8 x (largest share); the blocks beyond an XCD's share exit.  Inside a share, tiles run in groups of GM panels, rows
  // fastest, so that the ~32 tiles an XCD holds at once cover GM row panels x 8 column panels of operands.
  constexpr int GM = 4;
  int tm, tn;
  if constexpr (XCD_REMAP) {
    const int xcd = block_id & 7, idx = block_id >> 3;
    if (big_xcd_panels(p.tiles_m, p.tiles_n)) {
      const int q = p.tiles_m >> 3, r = p.tiles_m & 7;
      const int cnt = q + (xcd < r ? 1 : 0), p0 = xcd * q + min(xcd, r);
      if (idx >= cnt * p.tiles_n) return;
      const int group_sz = GM * p.tiles_n;
      const int first = (idx / group_sz) * GM;
      const int gm = min(GM, cnt - first);
      const int rin = idx - (idx / group_sz) * group_sz;
      tm = p0 + first + rin % gm; tn = rin / gm;
    } else {
      // few or badly divisible row panels (see big_xcd_panels): XCD x takes an even run of the GM-grouped tile list instead
      const int nt = p.tiles_m * p.tiles_n, q = nt >> 3, r = nt & 7;
      if (idx >= q + (xcd < r ? 1 : 0)) return;
      const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
      const int group_sz = GM * p.tiles_n;
      const int first_m = (lin / group_sz) * GM;
      const int gm = min(GM, p.tiles_m - first_m);
      const int rin = lin % group_sz;
      tm = first_m + rin % gm; tn = rin / gm;
    }
  } else {
    const int group_sz = GM * p.tiles_n;
    const int first_m = (block_id / group_sz) * GM;
    const int gm = min(GM, p.tiles_m - first_m);
    const int rin = block_id % group_sz;
    tm = first_m + rin % gm; tn = rin / gm;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = CONV ? (p.K + BK - 1) / BK : p.K / BK;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;                   // wave tile: rows wm*64, cols wn*64
  PH_TL_DECL;
  PH_TL(0);

  // ---- DMA addressing: instruction i of this wave covers tile rows (i*8 + wave)*8 .. +8, lane l -> row +(l>>3), LDS slot l&7
  const bf16* a_src[A_INSTR];
  const bf16* b_src[B_INSTR];
  // CONV: per row of the lane (loop invariant) the element offset of tap (0,0) and a 9-bit mask of the taps that fall inside the image
  int row_off[CONV ? A_INSTR : 1], tap_ok[CONV ? A_INSTR : 1];
  // the lane's k chunk inside a k-tile is the same for all of its instructions: ((i*8 + wave)*8 + (lane>>3)) >> 1 & 7 does not depend on i
  const int conv_k = (((lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7))) * 8;
  if constexpr (CONV == 1) {
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
      const PixRow px = pix_of(p.cv, min(m0 + (i * 8 + wave) * 8 + (lane >> 3), p.M - 1));
      row_off[i] = (px.base + px.iy0 * p.cv.W + px.ix0) * p.cv.C;          // (may be negative for a border pixel: only used where the tap is valid)
      int m = 0;
      for (int ty = 0; ty < p.cv.kh; ++ty)
        for (int tx = 0; tx < p.cv.kw; ++tx)
          m |= ((unsigned)(px.iy0 + ty) < (unsigned)p.cv.H && (unsigned)(px.ix0 + tx) < (unsigned)p.cv.W) ? (1 << (ty * p.cv.kw + tx)) : 0;
      tap_ok[i] = m;
    }
  }
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    if constexpr (CONV == 1) {
      a_src[i] = nullptr;
    } else if constexpr (TA) {   // A = [K][M]: one instruction = 2 k-rows x 512 B; lane -> k-row (lane >> 5), LDS slot lane & 31 holds chunk c
      const int kr = (i * 8 + wave) * 2 + (lane >> 5), pos = lane & 31;
      const int c = (((pos >> 2) ^ (kr & 3)) << 2) | (pos & 3);
      a_src[i] = p.A + (size_t)kr * p.lda + min(m0 + c * 8, p.M - 8);
    } else {
      const int r = (i * 8 + wave) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      a_src[i] = p.A + (size_t)min(m0 + r, p.M - 1) * p.lda + c * 8;
    }
  }
#pragma unroll
  for (int i = 0; i < B_INSTR; ++i) {
    if constexpr (TB) {   // B = [K][N]: one instruction = 4 k-rows x 256 B; lane -> k-row (lane >> 4), LDS slot lane & 15 holds chunk c
      const int kr = (i * 8 + wave) * 4 + (lane >> 4), pos = lane & 15;
      const int c = (((pos >> 2) ^ (kr & 3)) << 2) | (pos & 3);
      b_src[i] = p.B + (size_t)kr * p.ldb + min(n0 + c * 8, p.N - 8);
    } else {
      const int r = (i * 8 + wave) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      b_src[i] = p.B + (size_t)min(n0 + r, p.N - 1) * p.ldb + c * 8;
    }
  }
  // CONV: the gathered source addresses of k-tile kt.  ~60 VALU instructions per wave: inside the ping-pong loop they run in the R phase
  // (beside the fragment reads), NOT in front of the MFMAs of the M phase -- the phases are barrier-locked, so VALU work at the head of
  // an M phase idles the matrix pipe of the SIMD for its whole length (first build: 440 instead of 660 TFLOP/s on the 384->768 layer).
  const bf16* a_nxt[CONV ? A_INSTR : 1];
  const bf16* b_nxt[CONV ? B_INSTR : 1];
  // every scalar the per-tile address work needs is pinned in SGPRs up front: left to itself the compiler re-reads them from the kernel
  // arguments inside the loop (s_load + s_waitcnt lgkmcnt(0) in the middle of the fragment reads) and turns the selects into branches
  int cvW = p.cv.W, cvC = p.cv.C, cvKreal = p.cv.Kreal, cvkw = p.cv.kw, Kfull = p.K;
  float cv_inv_c = p.cv.inv_c;
  const bf16* conv_x = p.A;
  const bf16* zero_pg = p.zero16 + lane * 8;          // (a wave's padding lanes read 64 different 16-B pieces of the zero page, not one address)
  if constexpr (CONV == 1) asm volatile("" : "+s"(cvW), "+s"(cvC), "+s"(cvKreal), "+s"(cvkw), "+s"(Kfull), "+s"(cv_inv_c), "+s"(conv_x));
  // conv_prep(kt) is called for kt = 0, 1, 2, ... in order (every wave requests its tiles in order; at the tail the clamped index
  // repeats, see below).  C % 64 == 0 (192, 384, 768: every stem layer but the 96-channel one): a k-tile lies inside ONE tap, so the
  // (tap, channel) bookkeeping is wave-uniform and kept incrementally on the SALU -- no per-lane division; a state that has run past the
  // last tap (the clamped surplus requests of the non-LEAN loop) has no bit in tap_ok and reads the zero page.
  const bool conv_c64 = CONV == 1 && (cvC & 63) == 0 && (cvKreal & 63) == 0;
  int s_ky = 0, s_kx = 0, s_c0 = 0, s_tap = 0;
  auto conv_prep = [&](int kt) {
    if constexpr (CONV == 1) {
      const int koff = kt * BK;
      const int k = koff + conv_k;
      int tap_off, tap_bit;
      if (conv_c64) {
        tap_off = (s_ky * cvW + s_kx) * cvC + s_c0 + conv_k;
        tap_bit = s_tap < 16 ? (1 << s_tap) : 0;
        s_c0 += BK;
        if (s_c0 >= cvC) { s_c0 = 0; ++s_tap; ++s_kx; if (s_kx == cvkw) { s_kx = 0; ++s_ky; } }
      } else {
        const bool kin = k < cvKreal;
        const int kk = kin ? k : 0;
        const int tap = fdiv(kk, cvC, cv_inv_c), c0 = kk - tap * cvC;
        const int ky = cvkw == 3 ? (tap >= 6 ? 2 : (tap >= 3 ? 1 : 0)) : (cvkw == 2 ? (tap >> 1) : tap), kx = tap - ky * cvkw;
        tap_off = (ky * cvW + kx) * cvC + c0;
        tap_bit = kin ? (1 << tap) : 0;
      }
#pragma unroll
      for (int i = 0; i < A_INSTR; ++i) {
        const bf16* src = conv_x + (row_off[i] + tap_off);
        a_nxt[i] = (tap_ok[i] & tap_bit) ? src : zero_pg;
      }
      const bool bin = k < Kfull;
#pragma unroll
      for (int i = 0; i < B_INSTR; ++i) { const bf16* src = b_src[i] + koff; b_nxt[i] = bin ? src : zero_pg; }
#pragma unroll
      for (int i = 0; i < A_INSTR; ++i) asm volatile("" : "+v"(a_nxt[i]));      // computed HERE (volatile asms keep their order: in front of the phase's wait)
#pragma unroll
      for (int i = 0; i < B_INSTR; ++i) asm volatile("" : "+v"(b_nxt[i]));
    }
  };
  auto issue = [&](int kt, int stage) {
    char* sa = smem + stage * STAGE;
    char* sb = sa + A_BYTES;
    const int koff = kt * BK;
    if constexpr (CONV == 1) {          // sources prepared by conv_prep(kt) (ping-pong loop: in the R phase, see there)
#pragma unroll
      for (int i = 0; i < A_INSTR; ++i) __builtin_amdgcn_global_load_lds((gptr_t*)a_nxt[i], (lptr_t*)(sa + (i * 8 + wave) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < B_INSTR; ++i) __builtin_amdgcn_global_load_lds((gptr_t*)b_nxt[i], (lptr_t*)(sb + (i * 8 + wave) * 1024), 16, 0, 0);
      return;
    }
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t*)(a_src[i] + (TA ? (size_t)koff * p.lda : (size_t)koff)), (lptr_t*)(sa + (i * 8 + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t*)(b_src[i] + (TB ? (size_t)koff * p.ldb : (size_t)koff)), (lptr_t*)(sb + (i * 8 + wave) * 1024), 16, 0, 0);
  };

  // one DMA instruction of a tile (round 6, VARIANT & 16: the six requests of a k-tile are spread over the M phase, one behind every
  // second MFMA, instead of wherever the scheduler bunches them -- it put all six among the last four MFMAs, i.e. 24 requests of a wave
  // group reach the CU's one address path inside ~130 cycles and the waves stall on it with the matrix pipe idle)
  auto issue_piece = [&](int kt, int stage, auto idx_c) {
    constexpr int idx = decltype(idx_c)::value;
    char* sa = smem + stage * STAGE;
    char* sb = sa + A_BYTES;
    const int koff = kt * BK;
    if constexpr (CONV == 1) {          // sources prepared by conv_prep(kt) in the R phase
      if constexpr (idx < A_INSTR) __builtin_amdgcn_global_load_lds((gptr_t*)a_nxt[idx], (lptr_t*)(sa + (idx * 8 + wave) * 1024), 16, 0, 0);
      else __builtin_amdgcn_global_load_lds((gptr_t*)b_nxt[idx - A_INSTR], (lptr_t*)(sb + ((idx - A_INSTR) * 8 + wave) * 1024), 16, 0, 0);
    } else if constexpr (idx < A_INSTR)
      __builtin_amdgcn_global_load_lds((gptr_t*)(a_src[idx] + (TA ? (size_t)koff * p.lda : (size_t)koff)), (lptr_t*)(sa + (idx * 8 + wave) * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((gptr_t*)(b_src[idx - A_INSTR] + (TB ? (size_t)koff * p.ldb : (size_t)koff)), (lptr_t*)(sb + ((idx - A_INSTR) * 8 + wave) * 1024), 16, 0, 0);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int stage) {
    const char* la = smem + stage * STAGE;
    const char* lb = la + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 fx[2], fw[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fx[i] = TA ? frag_ks_dma<512>(la, wm * 64 + i * 32, kk, lane) : frag_kc(la, wm * 64 + i * 32, kk, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) fw[j] = TB ? frag_ks_dma<256>(lb, wn * 64 + j * 32, kk, lane) : frag_kc(lb, wn * 64 + j * 32, kk, lane);
      if (VARIANT & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[j], fx[i], acc[i][j], 0, 0, 0);
      if (VARIANT & 1) __builtin_amdgcn_s_setprio(0);
    }
  };

  if constexpr ((VARIANT & 4) == 0) {
    // ---- main loop: tile t lives in stage t % 3.  Every iteration issues exactly one tile (index clamped: the surplus loads of
    // the last two iterations land in a stage nobody reads again), so "all but the newest tile have landed" is always vmcnt(6).
    issue(0, 0);
    issue(min(1, nk - 1), 1);
    int st = 0;                                                // stage of tile t
    for (int t = 0; t < nk; ++t) {
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");         // my DMA writes of tile t are in LDS (tile t+1 may still fly)
      __builtin_amdgcn_s_barrier();                            // everybody's are; everybody finished reading tile t-1
      asm volatile("" ::: "memory");
      int st2 = st + 2; st2 = st2 >= 3 ? st2 - 3 : st2;
      issue(min(t + 2, nk - 1), st2);                          // overwrites the stage tile t-1 was read from
      compute(st);
      st = st + 1 == 3 ? 0 : st + 1;
    }
  } else {
    // ---- ping-pong main loop (VARIANT & 4).  The two waves of a SIMD (wave w of group 0 = waves 0..3, wave w+4 of group 1) take
    // turns on the matrix pipe: every k-tile is two PHASES per group, R(t) = the 16 ds_read_b128 of the wave's whole k-tile (64
    // fragment VGPRs) and M(t) = its 16 MFMAs with the LDS-DMA of a later tile issued between them; group 1 runs one phase behind
    // group 0, every phase ends in one workgroup barrier:
    //     phase 2t   : group 0 R(t)                 | group 1 M(t-1) + DMA(t+2)
    //     phase 2t+1 : group 0 M(t) + DMA(t+2)      | group 1 R(t)
    // so while one wave of a SIMD issues back-to-back MFMAs its partner collects operands, instead of both waves alternating
    // ds_read -> s_waitcnt -> 4 MFMAs in lockstep (what the compiler makes of the plain loop).
    // Ordering (3-stage ring, tile t in stage t % 3; group g's M(t) issues tile t+2+g, so every wave has exactly one tile newer
    // than the one it must have landed and the wait is always vmcnt(6)):
    //   RAW  every wave waits vmcnt(6) before the barrier that ends an ODD phase (group 0: after M(t)'s issue, group 1: in R(t)):
    //        all shares of tile t+1 are then in LDS, the first read of tile t+1 is in phase 2t+2 (one barrier later).
    //   WAR  R phases end with lgkmcnt(0) BEFORE their barrier; stage (t+2)%3 = stage of tile t-1 was last read in phase 2t-1 and
    //        is overwritten from phase 2t+1 (group 0) / 2t (group 1, tile t+2 = (t-1)+3) on.
    // LEAN tail (VARIANT & 8, round 4): no surplus DMA.  The clamped re-loads of the last iterations kept every wait at vmcnt(6) but had
    // to be drained before the ring could hold the C tile (0.45 us per tile, tools/timeline_probe.py), and group 0 idled through group 1's
    // last M phase.  Here a tile that does not exist is not requested: a wave whose newest request is the tile it needs waits vmcnt(0)
    // instead (newer = "tile t+2 exists"), nothing is in flight after the loop, group 1 skips the barrier behind its last M phase (pure
    // register work) and group 0 goes straight from its last barrier to the epilogue -- both groups have then passed 1 + 2*nk barriers.
    constexpr bool LEAN = (VARIANT & 8) != 0, SPREAD = (VARIANT & 16) != 0, RDMA = (VARIANT & 32) != 0;
    constexpr bool DIAG_NOWAIT = (VARIANT & 64) != 0;          // diagnostic builds (tools/big_probe.py modes 9-11): WRONG results by design
    static_assert(!RDMA || (LEAN && CONV == 0), "R-phase requests: LEAN tail, dense operands");
    const int grp = wave >> 2;
    conv_prep(0);
    issue(0, 0);
    conv_prep(min(1, nk - 1));
    issue(min(1, nk - 1), 1);
    if (grp) {
      if (!LEAN || nk > 2) {
        conv_prep(min(2, nk - 1));
        issue(min(2, nk - 1), 2);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      }
    } else {
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    PH_TL(1);
    __builtin_amdgcn_s_barrier();                              // tile 0 is in LDS for everybody
    PH_TL(2);
    if (grp) __builtin_amdgcn_s_barrier();                     // group 1 idles through phase 0
    int st = 0;
    // one k-tile of the schedule; NEWER: tile t+2 exists, ISS0 / ISS1: group 0 / 1 has a tile to request in M(t), LAST: t == nk - 1.
    // The flags are compile-time so that the steady-state loop carries no branch on t (the first LEAN build tested them inside the loop
    // and lost 0.1 us per k-tile to it: k loop 8.0 -> 9.3 us at K = 768, profiles/r4_timeline_after_epilogue_classes.txt); the last
    // three k-tiles of a LEAN launch are peeled copies of the body.
    auto ktile = [&](const int t, auto newer_c, auto iss0_c, auto iss1_c, auto last_c) {
      constexpr bool NEWER = decltype(newer_c)::value, ISS0 = decltype(iss0_c)::value, ISS1 = decltype(iss1_c)::value, LAST = decltype(last_c)::value;
      // R(t)
      const char* la = smem + st * STAGE;
      const char* lb = la + A_BYTES;
      bf16x8 fx[BK / 16][2], fw[BK / 16][2];
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
#pragma unroll
        for (int i = 0; i < 2; ++i) fx[kk][i] = TA ? frag_ks_dma<512>(la, wm * 64 + i * 32, kk, lane) : frag_kc(la, wm * 64 + i * 32, kk, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) fw[kk][j] = TB ? frag_ks_dma<256>(lb, wn * 64 + j * 32, kk, lane) : frag_kc(lb, wn * 64 + j * 32, kk, lane);
      }
      if constexpr (CONV == 1 && ISS0) {
        // the sources of the tile this wave requests in M(t), worked out while the fragments arrive, behind ALL 16 fragment reads (interleaved,
        // the VALU work delays their issue).  Measured on the 384->768 layer, one problem (tools/conv_probe.py; the materialised im2col
        // matrix through the plain kernel: 44 us): in front of the MFMAs of the M phase 64 us (VALU at the head of a barrier-locked phase
        // idles the matrix pipe), woven between the MFMAs 63 us, here 58 us with the per-lane division and less with the SALU bookkeeping.
        __builtin_amdgcn_sched_barrier(0);
        if (ISS1 || !grp) conv_prep(min(t + 2 + grp, nk - 1));
      }
      if (grp) {
        if constexpr (NEWER) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // M(t)
      int sn = st + 2 + grp; sn = sn >= 3 ? sn - 3 : sn;
      if (VARIANT & 1) __builtin_amdgcn_s_setprio(1);
      if constexpr (SPREAD) {
        // MFMA, MFMA, request, MFMA, MFMA, request, ... : requests behind MFMAs 1, 3, 5, 7, 9, 11; the last four MFMAs run bare in front of the wait
        const bool mine = ISS0 && (ISS1 || !grp);
        const int kt_n = min(t + 2 + grp, nk - 1);
        static_for(std::make_integer_sequence<int, 16>{}, [&](auto m_c) {
          constexpr int m = decltype(m_c)::value, kk = m / 4, i = (m / 2) % 2, j = m % 2;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[kk][j], fx[kk][i], acc[i][j], 0, 0, 0);
          if constexpr (ISS0 && (m & 1) && m / 2 < A_INSTR + B_INSTR) {
            __builtin_amdgcn_sched_barrier(0);
            if (mine) issue_piece(kt_n, sn, std::integral_constant<int, m / 2>{});
            __builtin_amdgcn_sched_barrier(0);
          }
        });
      } else {
      if constexpr (ISS0 && ISS1) issue(min(t + 2 + grp, nk - 1), sn);
      else if constexpr (ISS0) { if (!grp) issue(min(t + 2, nk - 1), sn); }
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[kk][j], fx[kk][i], acc[i][j], 0, 0, 0);
      }
      if (VARIANT & 1) __builtin_amdgcn_s_setprio(0);
      if (!grp) {
        if constexpr (NEWER) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (LAST) { if (!grp) __builtin_amdgcn_s_barrier(); }       // (LEAN: group 1 skips the barrier behind its last M phase)
      else __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      st = st + 1 == 3 ? 0 : st + 1;
    };
    using T_ = std::true_type; using F_ = std::false_type;
    if constexpr (!LEAN) {
      for (int t = 0; t < nk; ++t) ktile(t, T_{}, T_{}, T_{}, F_{});          // clamped surplus requests keep every wait at vmcnt(6)
    } else {
      int t = 0;
      for (; t + 3 < nk; ++t) ktile(t, T_{}, T_{}, T_{}, F_{});               // tiles t+2 (group 0) and t+3 (group 1) exist
      if (nk >= 3) { ktile(t, T_{}, T_{}, F_{}, F_{}); ++t; }                 // t = nk-3: only group 0 still has a tile to request
      ktile(t, F_{}, F_{}, F_{}, F_{}); ++t;                                  // t = nk-2: the newest request is the tile the wave needs
      ktile(t, F_{}, F_{}, F_{}, T_{});                                       // t = nk-1
    }
    PH_TL(3);
    if (!LEAN && !grp) __builtin_amdgcn_s_barrier();           // group 0 idles through the last phase (group 1's M(nk-1))
  }
  if constexpr ((VARIANT & 8) == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // drain the surplus DMA before the ring is reused as the C tile
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  PH_TL(4);

  // ---- epilogue (same chain as gemm_body, 512 threads, 256 x 128 tile).  The reads of the fused chain (bias, residual, saved
  // derivative) are requested here, before the accumulators are parked: their latency hides behind the LDS transpose and the store
  // loop never waits on memory (gemm_common.h, "Round 4").  Raw s_barrier + lgkmcnt(0): __syncthreads() would drain those loads.
  const int epi = epi_classify(p, false);
  PH_WO_DECL(BM, BN, NTHR);
  wo_prefetch<BM, BN, NTHR>(epi, p, m0, n0, PH_WO_ARGS);
  DropCtx dc;
  const bool drop = p.drop_p > 0.0f;
  if (drop) dc = make_drop(p.drop_seed, p.drop_stream, p.drop_p);
  constexpr int CH = BN / 4;
  float* cl = reinterpret_cast<float*>(smem);
  static_for(std::make_integer_sequence<int, 2 * 2 * 4>{}, [&](auto idx) {
    constexpr int i = decltype(idx)::value / 8, j = (decltype(idx)::value / 4) % 2, g = decltype(idx)::value % 4;
    const int ml = wm * 64 + i * 32 + (lane & 31);
    const int c = (wn * 64 + j * 32 + g * 8 + (lane >> 5) * 4) >> 2;
    f32x4 v = {acc[i][j][g * 4 + 0] * p.alpha, acc[i][j][g * 4 + 1] * p.alpha, acc[i][j][g * 4 + 2] * p.alpha,
               acc[i][j][g * 4 + 3] * p.alpha};
    *reinterpret_cast<f32x4*>(cl + ml * BN + ((c ^ (ml & (CH - 1))) << 2)) = v;
  });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  PH_TL(5);
  tile_writeout<BM, BN, NTHR>(PH_TL_ARG epi, p, cl, m0, n0, false, drop, dc, PH_WO_ARGS);
  if constexpr (CONV == 1) {
    if (p.col_stats) tile_colstats<BM, BN, NTHR>(p, cl, m0, n0);       // train-mode BatchNorm sums from the parked tile
  }
  PH_TL(8);
#ifdef PH_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PH_TL(9);
  PH_TL_FLUSH(block_id, wave >> 2, (threadIdx.x & 255) == 0);
#endif
}


template <int VARIANT, bool TA, bool TB>
__global__ __launch_bounds__(NTHR) void gemm_big_kernel(GemmParams p) {
  big_tile<VARIANT, TA, TB, true>(p, blockIdx.x);
}

template <int VARIANT, bool TA, bool TB>
int launch(const GemmParams& p, hipStream_t s) {
  PH_SET_SMEM_ONCE((&gemm_big_kernel<VARIANT, TA, TB>), SMEM);
  count_launch(PH_GEMM_CLS_BIG);
  hipLaunchKernelGGL((gemm_big_kernel<VARIANT, TA, TB>), dim3(big_xcd_grid(p.tiles_m, p.tiles_n)), dim3(NTHR), SMEM, s, p);      // (XCD shares, see big_tile)
  PH_LAUNCH_CHECK("gemm_big_kernel");
  return PH_OK;
}
}  // namespace big

// Grouped launch on the 256x128 ping-pong kernel (weight-gradient layout, long reductions): a persistent grid of one block per CU
// walks the tiles round by round; within a round the XCD-contiguous numbering of gemm_tile is kept (tiles that share an operand
// panel run on one XCD at the same time).
namespace big {
template <int VARIANT, bool TA, bool TB>
__global__ __launch_bounds__(NTHR) void gemm_big_grouped_kernel(GroupParams g) {
  const int total = g.tile_start[g.n], grid = gridDim.x;
  int i = 0;
  for (int base = 0; base < total; base += grid) {
    const int cnt = min(grid, total - base);
    if ((int)blockIdx.x >= cnt) break;
    const int q = cnt / 8, r = cnt % 8, xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
    const int t = base + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    while (i + 1 < g.n && t >= g.tile_start[i + 1]) ++i;
    big_tile<VARIANT, TA, TB, false>(g.p[i], t - g.tile_start[i]);
    __syncthreads();                     // the write-out's LDS staging area is the next tile's DMA ring
  }
}
template <int VARIANT, bool TA, bool TB>
int launch_grouped(const GroupParams& g, int total, hipStream_t s) {
  PH_SET_SMEM_ONCE((&gemm_big_grouped_kernel<VARIANT, TA, TB>), SMEM);
  count_launch(PH_GEMM_CLS_BIG_GROUPED);
  hipLaunchKernelGGL((gemm_big_grouped_kernel<VARIANT, TA, TB>), dim3(total < 256 ? total : 256), dim3(NTHR), SMEM, s, g);
  PH_LAUNCH_CHECK("gemm_big_grouped_kernel");
  return PH_OK;
}
// Forward-shaped implicit-GEMM convolutions, grouped (the same layer of several expert stems, or the parity classes of a stride-2
// data gradient): one block per tile -- the problems' k loops differ (1, 2 or 4 taps per parity class), so the hardware's block
// dispatch does the balancing -- numbered per problem like a single launch (XCD x owns whole 256-row panels of the problem: the
// activation rows a panel gathers, 9 taps each, are fetched by one L2).  tile_start[] counts BLOCKS here, a multiple of 8 per problem.
template <int VARIANT>
__global__ __launch_bounds__(NTHR) void gemm_big_conv_kernel(GroupParams g) {
  const int b = blockIdx.x;
  int i = 0;
  while (i + 1 < g.n && b >= g.tile_start[i + 1]) ++i;
  big_tile<VARIANT, false, false, true, 1>(g.p[i], b - g.tile_start[i]);
}
template <int VARIANT>
int launch_conv_t(const GroupParams& g, int blocks, hipStream_t s) {
  PH_SET_SMEM_ONCE((&gemm_big_conv_kernel<VARIANT>), SMEM);
  count_launch(PH_GEMM_CLS_BIG_GROUPED);
  hipLaunchKernelGGL((gemm_big_conv_kernel<VARIANT>), dim3(blocks), dim3(NTHR), SMEM, s, g);
  PH_LAUNCH_CHECK("gemm_big_conv_kernel");
  return PH_OK;
}
int launch_grouped_conv(const GroupParams& g, int blocks, int variant, hipStream_t s) {
  return variant == 7 ? launch_conv_t<28>(g, blocks, s) : launch_conv_t<12>(g, blocks, s);
}
}  // namespace big

namespace big {
// variant: 0 = plain loop, 5 = ping-pong + LEAN tail (round 4), 7 = + SPREAD requests (round 6, default)
int launch_single(const GemmParams& p, int variant, bool ta, bool tb, hipStream_t s) {
  if (ta) return variant == 7 ? launch<28, true, true>(p, s) : launch<12, true, true>(p, s);      // weight-gradient layout: ping-pong only
  if (tb) return variant == 0 ? launch<0, false, true>(p, s) : variant == 7 ? launch<28, false, true>(p, s) : launch<12, false, true>(p, s);
  return variant == 0 ? launch<0, false, false>(p, s) : variant == 7 ? launch<28, false, false>(p, s) : launch<12, false, false>(p, s);
}
int launch_grouped_wgrad(const GroupParams& g, int total, int variant, hipStream_t s) {
  return variant == 7 ? launch_grouped<28, true, true>(g, total, s) : launch_grouped<12, true, true>(g, total, s);
}
}  // namespace big
#ifdef PH_TIMELINE
extern "C" int ph_tl_fetch_big(unsigned long long* host, int n, int reset) {
  hipDeviceSynchronize();
  if (host && n > 0) hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tl), sizeof(unsigned long long) * (size_t)n);
  if (reset) { void* d = nullptr; hipGetSymbolAddress(&d, HIP_SYMBOL(g_tl)); hipMemset(d, 0, sizeof(g_tl)); }
  return 0;
}
#endif

}  // namespace phg
