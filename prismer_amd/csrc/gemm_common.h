// Shared device code of the GEMM translation units (gemm.hip: 128x128 / 64x64 / grouped kernels and the host dispatch;
// gemm_big.hip: the 256x128 LDS-DMA ping-pong kernels): operand loaders, fragment reads, the fused write-out chain, parameter blocks.
// Named namespace: the parameter structs cross the translation-unit boundary in the big-kernel launchers below.
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <math.h>
#include <atomic>
#include <mutex>
#include <utility>

#include "common.h"

namespace phg {


template <int... Is, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}

constexpr int BK = 64;
constexpr int KC_ROW_BYTES = BK * 2;   // K-contiguous image: 128 B per row

template <int R>
struct TileBytes {
  static constexpr int kc = R * KC_ROW_BYTES;          // K-contiguous image
  static constexpr int ks_stride = R * 2 + 64;         // K-strided image: bytes per k-row (64-B pad)
  static constexpr int ks = BK * ks_stride;
  static constexpr int max = ks > kc ? ks : kc;
};

// Implicit-GEMM convolution: one operand is the im2col VIEW of an NHWC activation x[B][H][W][C] (C % 8 == 0), never materialised.
// Logical matrix col[m][k], m = (b, oy, ox), k = (ky*ks + kx)*C + c (zero for out-of-image taps and for k >= ks*ks*C).
struct ConvGather {
  int H, W, C, ks, stride, Ho, Wo, Kreal;
  float inv_howo, inv_wo, inv_c;      // reciprocals for the index decompositions (operands < 2^24: one float multiply + fix-up)
  int kh, kw, offy, offx;             // window: tap (ty, tx) reads pixel (oy * stride + offy + ty, ox * stride + offx + tx); 3x3 pad 1: 3, 3, -1, -1
};
__device__ __forceinline__ int fdiv(int a, int d, float inv) {          // a / d for 0 <= a < 2^24, d > 0
  int q = (int)((float)a * inv);
  q += ((q + 1) * d <= a) ? 1 : 0;
  q -= (q * d > a) ? 1 : 0;
  return q;
}

struct GemmParams {
  const bf16* A; const bf16* B; void* C;
  int M, N, K, lda, ldb, ldc;
  const float* bias;
  const float* bias_or_zero;   // bias, or the library's device-resident zero vector when the call has none (N <= PH_ZERO_BIAS_FLOATS), or null
  bf16* pre_out;
  const bf16* act_in; int ld_act;
  const bf16* residual; int ldr; int res_f32;
  float drop_p; const uint64_t* drop_seed; uint32_t drop_stream;
  int act, out_f32, accumulate, pre_grad;
  float alpha;
  int k_tiles_per_split;   // in units of BK
  int tiles_m, tiles_n;
  float* ws; int ldws;     // split-K partial tiles: ws[split][M][ldws] fp32 (plain stores), folded by splitk_reduce_kernel
  double* col_stats;       // optional fp64 [2][N]: += column sums / sums of squares of the (bf16-rounded) outputs (BatchNorm statistics)
  int rm_wo, rm_mul, rm_sub, rm_add; float rm_inv_wo;     // output row map (rm_wo > 0): C row of result row m = m * mul - (m % wo) * sub + add
  ConvGather cv;           // CONV kernels only: geometry of the gathered operand (A for CONV=1, B for CONV=2)
  const bf16* zero16;      // >= 1 KB of device-resident zeros (64 lanes x 16 B): what the LDS-DMA lanes read for a padding tap / a chunk beyond K (gemm_big.hip, CONV)
};

// ---- XCD shares of a one-tile-per-block launch of the 256x128 kernel (gemm_big.hip big_tile, and its host-side grid sizes) ----------------
// Panel ownership (round 4): XCD x owns whole 256-row panels -> an A panel is fetched by one L2.  It only pays while it costs no extra block
// round: with tiles_m < 8 some XCDs own nothing (M = 512, N = 4096: 2 of 8 XCDs work), with tiles_m = 8k + 1 one XCD owns twice the others'
// share (round-4 advisor finding).  Rule, evaluated identically on host and device: panels iff every XCD gets one AND the largest share needs no
// more rounds of the XCD's 32 CUs than an even split of the tile list would; otherwise even runs of the (GM-grouped) tile list per XCD.
__host__ __device__ inline int big_xcd_even_share(int tiles_m, int tiles_n) { return (tiles_m * tiles_n + 7) / 8; }
__host__ __device__ inline bool big_xcd_panels(int tiles_m, int tiles_n) {
  const int share = ((tiles_m + 7) / 8) * tiles_n;
  return tiles_m >= 8 && (share + 31) / 32 <= (big_xcd_even_share(tiles_m, tiles_n) + 31) / 32;
}
__host__ __device__ inline int big_xcd_grid(int tiles_m, int tiles_n) {
  return 8 * (big_xcd_panels(tiles_m, tiles_n) ? ((tiles_m + 7) / 8) * tiles_n : big_xcd_even_share(tiles_m, tiles_n));
}

// C row of result row m (identity unless an output row map is set: parity-class data gradients of the stride-2 convolutions)
__device__ __forceinline__ size_t crow(const GemmParams& p, int m) {
  if (p.rm_wo == 0) return (size_t)m;
  const int q = fdiv(m, p.rm_wo, p.rm_inv_wo);
  return (size_t)(m * p.rm_mul - (m - q * p.rm_wo) * p.rm_sub + p.rm_add);
}

// ---- global -> register staging ------------------------------------------------------------------------
// K-contiguous operand: tile = R rows x 64 k. chunk id -> (row = id/8, c = id%8), 16 B each.
// KFULL: K is a multiple of BK, so no k predicate -> straight-line loads (the compiler's vmcnt bookkeeping stays exact,
// which the deep prefetch ring depends on).
template <int R, bool KFULL = false>
__device__ __forceinline__ void load_kc(const bf16* __restrict__ base, int ld, int row0, int rows, int k0, int K,
                                        u32x4 (&regs)[R * 8 / 256], const int tid) {
#pragma unroll
  for (int i = 0; i < R * 8 / 256; ++i) {
    int id = tid + 256 * i;
    int r = id >> 3, c = id & 7;
    int row = row0 + r;
    row = row < rows ? row : rows - 1;          // clamp: out-of-range rows only feed out-of-range outputs
    int k = k0 + c * 8;
    if (KFULL) {
      regs[i] = *reinterpret_cast<const u32x4*>(base + (size_t)row * ld + k);
    } else {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (k < K) v = *reinterpret_cast<const u32x4*>(base + (size_t)row * ld + k);
      regs[i] = v;
    }
  }
}
template <int R>
__device__ __forceinline__ void store_kc(char* lds, const u32x4 (&regs)[R * 8 / 256], const int tid) {
#pragma unroll
  for (int i = 0; i < R * 8 / 256; ++i) {
    int id = tid + 256 * i;
    int r = id >> 3, c = id & 7;
    int cs = c ^ ((r >> 1) & 7);
    *reinterpret_cast<u32x4*>(lds + r * KC_ROW_BYTES + cs * 16) = regs[i];
  }
}
// K-strided operand: memory [K][rows] (rows contiguous). tile = 64 k-rows x R. chunk id -> (kr = id/(R/8), c = id%(R/8)).
template <int R, bool KFULL = false>
__device__ __forceinline__ void load_ks(const bf16* __restrict__ base, int ld, int row0, int rows, int k0, int K,
                                        u32x4 (&regs)[R * 8 / 256], const int tid) {
  constexpr int CPR = R / 8;
#pragma unroll
  for (int i = 0; i < R * 8 / 256; ++i) {
    int id = tid + 256 * i;
    int kr = id / CPR, c = id % CPR;
    int k = k0 + kr;
    int r = row0 + c * 8;
    if (KFULL) {                                 // chunks beyond `rows` re-read the last chunk (they only feed out-of-range outputs)
      r = r < rows ? r : ((rows - 1) & ~7);
      regs[i] = *reinterpret_cast<const u32x4*>(base + (size_t)k * ld + r);
    } else {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (k < K && r < rows) v = *reinterpret_cast<const u32x4*>(base + (size_t)k * ld + r);
      regs[i] = v;
    }
  }
}
template <int R>
__device__ __forceinline__ void store_ks(char* lds, const u32x4 (&regs)[R * 8 / 256], const int tid) {
  constexpr int CPR = R / 8;
#pragma unroll
  for (int i = 0; i < R * 8 / 256; ++i) {
    int id = tid + 256 * i;
    int kr = id / CPR, c = id % CPR;
    *reinterpret_cast<u32x4*>(lds + kr * TileBytes<R>::ks_stride + c * 16) = regs[i];
  }
}

// CONV = 1: K-contiguous A operand gathered from the activation.  Per thread the tile rows are fixed over the k loop, so the
// pixel decomposition (PixRow) is done once; per k-tile one (tap, channel) decomposition of this thread's 8-wide k chunk.
struct PixRow { int base, iy0, ix0; };     // base = b*H*W (pixels), (iy0, ix0) = input coordinates of tap (0,0)
__device__ __forceinline__ PixRow pix_of(const ConvGather& cv, int m) {
  int b = fdiv(m, cv.Ho * cv.Wo, cv.inv_howo);
  int rem = m - b * cv.Ho * cv.Wo;
  int oy = fdiv(rem, cv.Wo, cv.inv_wo), ox = rem - oy * cv.Wo;
  return PixRow{b * cv.H * cv.W, oy * cv.stride + cv.offy, ox * cv.stride + cv.offx};
}
template <int R>
__device__ __forceinline__ void load_kc_conv(const ConvGather& cv, const bf16* __restrict__ x, const PixRow (&px)[R * 8 / 256], int k0,
                                             u32x4 (&regs)[R * 8 / 256]) {
  const int k = k0 + (threadIdx.x & 7) * 8;
  const bool kin = k < cv.Kreal;
  const int kk = kin ? k : 0;
  const int tap = fdiv(kk, cv.C, cv.inv_c), c0 = kk - tap * cv.C;
  const int ky = cv.kw == 3 ? (tap >= 6 ? 2 : (tap >= 3 ? 1 : 0)) : (cv.kw == 2 ? (tap >> 1) : tap), kx = tap - ky * cv.kw;
#pragma unroll
  for (int i = 0; i < R * 8 / 256; ++i) {
    const int iy = px[i].iy0 + ky, ix = px[i].ix0 + kx;
    const bool ok = kin && (unsigned)iy < (unsigned)cv.H && (unsigned)ix < (unsigned)cv.W;
    const int iyc = min(max(iy, 0), cv.H - 1), ixc = min(max(ix, 0), cv.W - 1);          // always a valid address: the load is unconditional
    u32x4 v = *reinterpret_cast<const u32x4*>(x + ((size_t)(px[i].base + iyc * cv.W + ixc)) * cv.C + c0);
    const u32x4 z = {0u, 0u, 0u, 0u};
    regs[i] = ok ? v : z;
  }
}
// CONV = 2: K-strided B operand (wgrad: reduction index = output pixel m, column = (tap, channel)).  Thread -> ONE k row
// (pixel) per k-tile and R/32 column chunks 4j + (tid & 3): the pixel decomposition (two divisions) is paid once per thread and
// k-tile, the column decompositions are loop invariants (ColTap, computed before the k loop).
struct ColTap { int dy, dx, c0, ok; };
template <int R>
__device__ __forceinline__ void coltaps_of(const ConvGather& cv, int col0, int ncols, ColTap (&ct)[R * 8 / 256]) {
#pragma unroll
  for (int j = 0; j < R * 8 / 256; ++j) {
    const int col = col0 + (4 * j + (threadIdx.x & 3)) * 8;
    const bool cin = col < cv.Kreal && col < ncols;
    const int cc = cin ? col : 0;
    const int tap = fdiv(cc, cv.C, cv.inv_c);
    const int ky = cv.ks == 3 ? (tap >= 6 ? 2 : (tap >= 3 ? 1 : 0)) : 0;
    ct[j] = ColTap{ky, tap - ky * cv.ks, cc - tap * cv.C, cin ? 1 : 0};
  }
}
template <int R>
__device__ __forceinline__ void load_ks_conv(const ConvGather& cv, const bf16* __restrict__ x, const ColTap (&ct)[R * 8 / 256], int k0, int K,
                                             u32x4 (&regs)[R * 8 / 256]) {
  const int m = k0 + ((int)threadIdx.x >> 2);
  const bool min_ = m < K;
  const PixRow p = pix_of(cv, min_ ? m : 0);
#pragma unroll
  for (int j = 0; j < R * 8 / 256; ++j) {
    const int iy = p.iy0 + ct[j].dy, ix = p.ix0 + ct[j].dx;
    const bool ok = ct[j].ok && min_ && (unsigned)iy < (unsigned)cv.H && (unsigned)ix < (unsigned)cv.W;
    const int iyc = min(max(iy, 0), cv.H - 1), ixc = min(max(ix, 0), cv.W - 1);
    u32x4 v = *reinterpret_cast<const u32x4*>(x + ((size_t)(p.base + iyc * cv.W + ixc)) * cv.C + ct[j].c0);
    const u32x4 z = {0u, 0u, 0u, 0u};
    regs[j] = ok ? v : z;
  }
}
template <int R>
__device__ __forceinline__ void store_ks_conv(char* lds, const u32x4 (&regs)[R * 8 / 256]) {
#pragma unroll
  for (int j = 0; j < R * 8 / 256; ++j)
    *reinterpret_cast<u32x4*>(lds + ((int)threadIdx.x >> 2) * TileBytes<R>::ks_stride + (4 * j + (threadIdx.x & 3)) * 16) = regs[j];
}

// ---- LDS -> MFMA fragment ------------------------------------------------------------------------------
// 32x32x16 operand fragment: lane l holds 8 consecutive k for row (l & 31), k-half (l >> 5).
__device__ __forceinline__ bf16x8 frag_kc(const char* lds, int rbase, int kk, int lane) {
  int r = rbase + (lane & 31);
  int c = kk * 2 + (lane >> 5);
  int cs = c ^ ((r >> 1) & 7);
  return *reinterpret_cast<const bf16x8*>(lds + r * KC_ROW_BYTES + cs * 16);
}
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
template <int R>
__device__ __forceinline__ bf16x8 frag_ks(const char* lds, int rbase, int kk, int lane) {
  // ds_read_b64_tr_b16: within a 16-lane group, source lane (4j+q) supplies 4 consecutive row-elements of
  // k-row j; result lane c receives element j = T[k_j][rows 4*(c/4).. + c%4] i.e. column c of the 4x16 block.
  int g = lane >> 4, i = lane & 15, j = i >> 2, q = i & 3;
  int k = kk * 16 + (g >> 1) * 8 + j;
  int r = rbase + (g & 1) * 16 + q * 4;
  const char* p = lds + k * TileBytes<R>::ks_stride + r * 2;
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 4 * TileBytes<R>::ks_stride));
  bf16x8 o;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
  o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
  return o;
}

// K-strided operand image written by the LDS-DMA path (big kernel; B given as [K][N], A given as [K][M]): 64 k-rows of 256 B (128 n)
// or 512 B (256 m), no padding -- the
// DMA destination is lane-linear.  Bank spreading is done by XOR-ing the 64-B block index of a k-row with (k & 3): the four k-rows a
// 16-lane group of ds_read_b64_tr_b16 touches then sit in four different 16-bank quarters (same effect as the 64-B row pad of the
// register-staged image).  The DMA source addressing applies the same permutation (gemm_big_kernel).
template <int RB>      // bytes per k-row: 256 (128-wide B tile) or 512 (256-wide A tile)
__device__ __forceinline__ bf16x8 frag_ks_dma(const char* lds, int rbase, int kk, int lane) {
  const int g = lane >> 4, i = lane & 15, j = i >> 2, q = i & 3;
  const int k = kk * 16 + (g >> 1) * 8 + j;                    // k & 3 == j, for the second read (k + 4) too
  const int byte = (rbase + (g & 1) * 16 + q * 4) * 2;         // offset inside the k-row
  const char* p = lds + k * RB + ((((byte >> 6) ^ j) << 6) | (byte & 63));
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 4 * RB));
  bf16x8 o;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
  o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
  return o;
}

// one lane's 4 consecutive outputs C[m][n..n+3]
__device__ __forceinline__ void epilogue_store(const GemmParams& p, int m, int n, float (&v)[4], bool splitk, bool drop,
                                               const DropCtx& dc, int split_id = 0) {
  const bool full = (n + 4 <= p.N);
  if (splitk && p.ws) {   // split-K with workspace: raw partial sums, the full epilogue runs in splitk_reduce_kernel
    float* c = p.ws + ((size_t)split_id * p.M + m) * p.ldws + n;
    f32x4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(c) = t;      // ldws % 4 == 0 and n % 4 == 0: always a full, aligned vector (pad columns are scratch)
    return;
  }
  if (splitk) {   // no workspace: raw fp32 atomics into C (host guarantees a plain fp32 accumulate epilogue)
    float* c = reinterpret_cast<float*>(p.C) + crow(p, m) * p.ldc + n;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (n + e < p.N) atomicAdd(c + e, v[e]);
    return;
  }
  if (p.bias) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += p.bias[min(n + e, p.N - 1)];
  }
  const bool fused_grad = p.pre_out && p.pre_grad && !p.act_in;
  if (p.pre_out) {
    bf16* q = p.pre_out + (size_t)m * p.ldc + n;
    float w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (fused_grad) act_fwd_grad(p.act, v[e], v[e], w[e]);       // v becomes act(x), w = act'(x)
      else w[e] = v[e];
    }
    if (full) { bf16x4 t = {f2bf(w[0]), f2bf(w[1]), f2bf(w[2]), f2bf(w[3])}; *reinterpret_cast<bf16x4*>(q) = t; }
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (n + e < p.N) q[e] = f2bf(w[e]);
    }
  }
  if (p.act_in) {       // backward through an activation: multiply by act'(saved pre-activation)
    const bf16* q = p.act_in + (size_t)m * p.ld_act + n;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= act_grad(p.act, bf2f(q[min(e, p.N - 1 - n)]));
  } else if (p.act != PH_ACT_NONE && !fused_grad) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = act_fwd(p.act, v[e]);
  }
  if (drop) {           // element index m*N+n ; N % 4 == 0 is required with dropout (checked on host)
    u32x4 r = drop_rand4(dc, ((uint64_t)m * (uint64_t)p.N + (uint64_t)n) >> 2);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = drop_apply(dc, r[e], v[e]);
  }
  if (p.residual && p.res_f32) {    // fp32 residual stream (decoder: LayerNorm outputs stay fp32 like under autocast)
    const float* q = reinterpret_cast<const float*>(p.residual) + (size_t)m * p.ldr + n;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += q[min(e, p.N - 1 - n)];
  } else if (p.residual) {
    const bf16* q = p.residual + (size_t)m * p.ldr + n;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += bf2f(q[min(e, p.N - 1 - n)]);
  }
  if (p.out_f32) {
    float* c = reinterpret_cast<float*>(p.C) + crow(p, m) * p.ldc + n;
    if (p.accumulate) {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (n + e < p.N) v[e] += c[e];
    }
    if (full && ((p.ldc & 3) == 0)) { f32x4 t = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f32x4*>(c) = t; }
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (n + e < p.N) c[e] = v[e];
    }
  } else {
    bf16* c = reinterpret_cast<bf16*>(p.C) + crow(p, m) * p.ldc + n;
    if (p.accumulate) {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (n + e < p.N) v[e] += bf2f(c[e]);
    }
    if (full && ((p.ldc & 3) == 0)) { bf16x4 t = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])}; *reinterpret_cast<bf16x4*>(c) = t; }
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (n + e < p.N) c[e] = f2bf(v[e]);
    }
  }
}

// 8 consecutive outputs C[m][n..n+7] with 16-B vector loads/stores (interior tiles, 16-B aligned leading dimensions), in two
// halves: epi_load8 issues the global READS of the fused chain (saved activation, residual), epi_apply8 does the arithmetic and the
// stores.  The write-out loops issue the loads of several steps before the first apply, so their latency (1-2 us under load) is
// paid once per group instead of once per step (stores to C may alias the residual -- in-place residual adds -- so the compiler
// cannot hoist the loads by itself; every thread reads exactly the elements it later writes, which keeps the reordering exact).
struct EpiIn { bf16x8 a, rb; f32x4 r0, r1; };   // act_in | bf16 residual | fp32 residual (typed fields: no punning through the
                                                 // aggregate, or it is not promoted to registers)
__device__ __forceinline__ void epi_load8(const GemmParams& p, int m, int n, EpiIn& in) {
  if (p.act_in) in.a = *reinterpret_cast<const bf16x8*>(p.act_in + (size_t)m * p.ld_act + n);
  if (p.residual && p.res_f32) {
    const float* q = reinterpret_cast<const float*>(p.residual) + (size_t)m * p.ldr + n;
    in.r0 = *reinterpret_cast<const f32x4*>(q); in.r1 = *reinterpret_cast<const f32x4*>(q + 4);
  } else if (p.residual) {
    in.rb = *reinterpret_cast<const bf16x8*>(p.residual + (size_t)m * p.ldr + n);
  }
}
__device__ __forceinline__ void epi_apply8(const GemmParams& p, int m, int n, float (&v)[8], bool drop, const DropCtx& dc, const EpiIn& in,
                                           const f32x4& b0, const f32x4& b1) {
  if (p.bias) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
  }
  const bool fused_grad = p.pre_out && p.pre_grad && !p.act_in;
  if (p.pre_out) {
    bf16x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float w = v[e];
      if (fused_grad) act_fwd_grad(p.act, v[e], v[e], w);
      t[e] = f2bf(w);
    }
    *reinterpret_cast<bf16x8*>(p.pre_out + (size_t)m * p.ldc + n) = t;
  }
  if (p.act_in) {
    const bf16x8 t = in.a;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= act_grad(p.act, bf2f(t[e]));
  } else if (p.act != PH_ACT_NONE && !fused_grad) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = act_fwd(p.act, v[e]);
  }
  if (drop) {
    uint64_t i4 = ((uint64_t)m * (uint64_t)p.N + (uint64_t)n) >> 2;
    u32x4 r0 = drop_rand4(dc, i4), r1 = drop_rand4(dc, i4 + 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = drop_apply(dc, r0[e], v[e]); v[4 + e] = drop_apply(dc, r1[e], v[4 + e]); }
  }
  if (p.residual && p.res_f32) {
    const f32x4 r0 = in.r0, r1 = in.r1;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
  } else if (p.residual) {
    const bf16x8 t = in.rb;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += bf2f(t[e]);
  }
  if (p.out_f32) {
    float* c = reinterpret_cast<float*>(p.C) + crow(p, m) * p.ldc + n;
    if (p.accumulate) {
      f32x4 c0 = *reinterpret_cast<const f32x4*>(c), c1 = *reinterpret_cast<const f32x4*>(c + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += c0[e]; v[4 + e] += c1[e]; }
    }
    f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
    *reinterpret_cast<f32x4*>(c) = o0;
    *reinterpret_cast<f32x4*>(c + 4) = o1;
  } else {
    bf16* c = reinterpret_cast<bf16*>(p.C) + crow(p, m) * p.ldc + n;
    if (p.accumulate) {
      bf16x8 t = *reinterpret_cast<const bf16x8*>(c);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bf2f(t[e]);
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
    *reinterpret_cast<bf16x8*>(c) = o;
  }
}

// Row-wise write-out of a BM x BN fp32 tile parked in LDS (16-B chunks XOR-swizzled by the row): consecutive lanes own
// consecutive 16/32-B pieces of one output row for every load / store of the fused epilogue chain.
template <int BM, int BN, int NTHR, int NPART = 1, int GCAP = 4>      // NPART: partial tiles to sum (intra-block k split); GCAP: cap on the prefetch group
__device__ __forceinline__ void tile_writeout_generic(PH_TL_PARAM const GemmParams& p, const float* cl, int m0, int n0, bool splitk, bool drop,
                                              const DropCtx& dc, int split_id = 0) {   // split_id: workspace slice of a split-K block; the NPART partial tiles lie back to back at cl
  constexpr int CH = BN / 4;                       // 16-B chunks per tile row
  // fast path: 8 outputs per thread per step (16-B loads/stores) when every leading dimension / pointer allows it
  const bool vec8 = !(splitk) && p.N >= 8 && (p.ldc % 8) == 0 && (!p.act_in || (p.ld_act % 8) == 0) && (!p.residual || (p.ldr % 8) == 0) &&
                    ((reinterpret_cast<uintptr_t>(p.C) | reinterpret_cast<uintptr_t>(p.pre_out) | reinterpret_cast<uintptr_t>(p.act_in) |
                      reinterpret_cast<uintptr_t>(p.residual) | reinterpret_cast<uintptr_t>(p.bias)) & 15) == 0;
  if (vec8) {
    // the bias of a thread's column block is the same for all its steps (NTHR is a multiple of the CH / 2 column blocks of a row): read once,
    // ahead of every store (round 4, see "epilogue classes" below)
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
    {
      const int c = ((int)threadIdx.x % (CH / 2)) * 2;
      const int n = (n0 + c * 4 + 8 <= p.N) ? n0 + c * 4 : 0;
      if (p.bias) { b0 = *reinterpret_cast<const f32x4*>(p.bias + n); b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4); }
    }
    // steps per thread; steps per load group (16 VGPRs of prefetched inputs per step: 4 for the 512-thread kernel, whose register budget
    // is 256/lane at its occupancy; 2 for the 256-thread kernels, which must stay under 192 + 64 accumulators for 2 blocks per CU)
    constexpr int IT = BM * CH / (2 * NTHR), G = (NTHR >= 512 && IT % 4 == 0 && GCAP >= 4) ? 4 : ((IT % 2 == 0 && GCAP >= 2) ? 2 : 1);
    for (int it0 = 0; it0 < IT; it0 += G) {
      EpiIn in[G];                  // (indexed with compile-time constants only and fully initialised: stays in VGPRs)
      static_for(std::make_integer_sequence<int, G>{}, [&](auto uu) {
        constexpr int u = decltype(uu)::value;
        const int id = (it0 + u) * NTHR + threadIdx.x;
        const int ml = id / (CH / 2), c = (id % (CH / 2)) * 2;
        const int m = min(m0 + ml, p.M - 1), n = (n0 + c * 4 + 8 <= p.N) ? n0 + c * 4 : 0;     // always a valid address: the loads are unconditional
        in[u] = EpiIn{};
        epi_load8(p, m, n, in[u]);
      });
      if (it0 == 0) PH_TL(6);
      static_for(std::make_integer_sequence<int, G>{}, [&](auto uu) {
        constexpr int u = decltype(uu)::value;
        const int id = (it0 + u) * NTHR + threadIdx.x;
        const int ml = id / (CH / 2), c = (id % (CH / 2)) * 2;
        const int m = m0 + ml, n = n0 + c * 4;
        const int sw = ml & (CH - 1);
        f32x4 t0 = *reinterpret_cast<const f32x4*>(cl + ml * BN + ((c ^ sw) << 2));
        f32x4 t1 = *reinterpret_cast<const f32x4*>(cl + ml * BN + (((c + 1) ^ sw) << 2));
#pragma unroll
        for (int q = 1; q < NPART; ++q) {
          t0 += *reinterpret_cast<const f32x4*>(cl + q * BM * BN + ml * BN + ((c ^ sw) << 2));
          t1 += *reinterpret_cast<const f32x4*>(cl + q * BM * BN + ml * BN + (((c + 1) ^ sw) << 2));
        }
        if (m < p.M && n + 8 <= p.N) {
          float v[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
          epi_apply8(p, m, n, v, drop, dc, in[u], b0, b1);
        } else if (m < p.M) {
          float v0[4] = {t0[0], t0[1], t0[2], t0[3]}, v1[4] = {t1[0], t1[1], t1[2], t1[3]};
          if (n < p.N) epilogue_store(p, m, n, v0, false, drop, dc);
          if (n + 4 < p.N) epilogue_store(p, m, n + 4, v1, false, drop, dc);
        }
      });
      if (it0 == 0) PH_TL(7);
    }
  } else {
#pragma unroll 4
    for (int it = 0; it < BM * CH / NTHR; ++it) {
      const int id = it * NTHR + threadIdx.x;
      const int ml = id / CH, c = id % CH;
      const int m = m0 + ml, n = n0 + c * 4;
      f32x4 t = *reinterpret_cast<const f32x4*>(cl + ml * BN + ((c ^ (ml & (CH - 1))) << 2));
#pragma unroll
      for (int q = 1; q < NPART; ++q) t += *reinterpret_cast<const f32x4*>(cl + q * BM * BN + ml * BN + ((c ^ (ml & (CH - 1))) << 2));
      float v[4] = {t[0], t[1], t[2], t[3]};
      if (m < p.M && n < p.N) epilogue_store(p, m, n, v, splitk, drop, dc, split_id);
    }
  }
}

// BatchNorm statistics of a conv-as-GEMM output, taken from the tile while it is parked in LDS: per column the sum and the sum of
// ---- epilogue classes (round 4) -----------------------------------------------------------------------------------------------
// gfx950 retires a wave's vector-memory operations on ONE in-order counter: a wait for a load that was issued after a store also
// waits for that store's write acknowledgement (1-2 us under load).  The generic write-out above reads the bias / residual / saved
// derivative between its stores, behind two dozen wave-uniform runtime branches that hipcc answers with s_waitcnt vmcnt(0) at every
// join: a 256x128 tile with bias + residual paid eight to ten dependent store-ack + load round trips -- 11.7 us of write-out against
// an 8 us k loop, 4.3 us for the same tile without bias and residual (tools/timeline_probe.py, profiles/r4_timeline_before.txt).
// The call sites of the training step need only a handful of chains, so each of them is a CLASS with a branch-free write-out:
//   every global read of the tile (bias, residual or saved derivative or old C, for ALL steps of the thread) is requested by
//   wo_prefetch() before the accumulators are parked -- its latency hides behind the LDS transpose -- and wo_apply<EPI>() is straight-line
//   code: LDS read -> arithmetic -> store, no memory wait between the stores.
// Everything else (dropout, fp32 residual streams, row maps, ragged N, split-K partials, ...) takes the generic path.
enum { EPI_GENERIC = 0,
       EPI_PLAIN,        // bias -> bf16                                   (QKV / K,V projections, plain data gradients)
       EPI_RES,          // bias + bf16 residual -> bf16                   (attention out-proj, adaptor up, MLP proj; dgrad + skip gradient)
       EPI_QGELU_GRAD,   // bias, QuickGELU value -> bf16 C, derivative -> bf16 pre_out          (ViT c_fc)
       EPI_RELU2_GRAD,   // bias, squared-ReLU value -> C, derivative -> pre_out                 (adaptor down, resampler fc)
       EPI_SAVED,        // x saved derivative (act_in) -> bf16            (data gradient through an activation)
       EPI_F32 };        // alpha * acc -> fp32                            (weight gradients, single writer; accumulating calls: generic path --
                         //                                                  their old C would cost 64 more prefetch registers per thread)
constexpr int PH_ZERO_BIAS_FLOATS = 1 << 16;          // the library substitutes a device-resident zero vector for a null bias (fill_params)
__device__ __forceinline__ int epi_classify(const GemmParams& p, bool splitk) {
  const bool aligned = p.N >= 8 && (p.N % 8) == 0 && (p.ldc % 8) == 0 && (!p.act_in || (p.ld_act % 8) == 0) && (!p.residual || (p.ldr % 8) == 0) &&
                       ((reinterpret_cast<uintptr_t>(p.C) | reinterpret_cast<uintptr_t>(p.pre_out) | reinterpret_cast<uintptr_t>(p.act_in) |
                         reinterpret_cast<uintptr_t>(p.residual) | reinterpret_cast<uintptr_t>(p.bias_or_zero)) & 15) == 0;
  if (!aligned || splitk || p.drop_p > 0.0f || p.rm_wo != 0 || !p.bias_or_zero) return EPI_GENERIC;
  const bool plain_in = !p.pre_out && !p.act_in && p.act == PH_ACT_NONE;
  if (p.out_f32) return (plain_in && !p.residual && !p.bias && !p.accumulate) ? EPI_F32 : EPI_GENERIC;
  if (p.accumulate) return EPI_GENERIC;
  if (p.act_in) return (p.act == PH_ACT_SAVED_GRAD && !p.pre_out && !p.residual && !p.bias) ? EPI_SAVED : EPI_GENERIC;
  if (p.pre_out) {
    if (!p.pre_grad || p.residual) return EPI_GENERIC;
    return p.act == PH_ACT_QUICKGELU ? EPI_QGELU_GRAD : (p.act == PH_ACT_RELU2 ? EPI_RELU2_GRAD : EPI_GENERIC);
  }
  if (p.act != PH_ACT_NONE) return EPI_GENERIC;
  if (p.residual) return p.res_f32 ? EPI_GENERIC : EPI_RES;
  return EPI_PLAIN;
}

template <int BM, int BN, int NTHR>
struct WoCfg {
  static constexpr int CH = BN / 4, IT = BM * CH / (2 * NTHR);      // 16-B chunks per tile row; steps of 8 outputs per thread
  static_assert(NTHR % (CH / 2) == 0 && IT >= 1, "write-out: a thread must keep its column block over its steps");
};
// the prefetched values are plain locals of the calling kernel (an aggregate holding them ended up in scratch memory)
#define PH_WO_DECL(BM, BN, NTHR) bf16x8 wo_x[WoCfg<BM, BN, NTHR>::IT]; f32x4 wo_b0, wo_b1
#define PH_WO_ARGS wo_x, wo_b0, wo_b1
template <int BM, int BN, int NTHR>
__device__ __forceinline__ void wo_prefetch(const int epi, const GemmParams& p, int m0, int n0, bf16x8 (&x)[WoCfg<BM, BN, NTHR>::IT],
                                            f32x4& b0, f32x4& b1) {
  constexpr int CH = BN / 4, IT = WoCfg<BM, BN, NTHR>::IT;
  if (epi == EPI_GENERIC) return;
  const int c = ((int)threadIdx.x % (CH / 2)) * 2;
  const int n = (n0 + c * 4 + 8 <= p.N) ? n0 + c * 4 : 0;          // always a valid address: the loads are unconditional
  b0 = *reinterpret_cast<const f32x4*>(p.bias_or_zero + n);          // (a zero vector when the call has no bias: no branch, no select)
  b1 = *reinterpret_cast<const f32x4*>(p.bias_or_zero + n + 4);
  if (epi == EPI_RES || epi == EPI_SAVED) {
    const bf16* src = epi == EPI_RES ? p.residual : p.act_in;
    const int ld = epi == EPI_RES ? p.ldr : p.ld_act;
    static_for(std::make_integer_sequence<int, IT>{}, [&](auto uu) {
      constexpr int u = decltype(uu)::value;
      const int ml = (u * NTHR + (int)threadIdx.x) / (CH / 2);
      x[u] = *reinterpret_cast<const bf16x8*>(src + (size_t)min(m0 + ml, p.M - 1) * ld + n);
    });
  }
}
__device__ __forceinline__ void wo_settle(f32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void wo_settle(bf16x8& v) {
  u32x4 t = __builtin_bit_cast(u32x4, v);
  asm volatile("" : "+v"(t));
  v = __builtin_bit_cast(bf16x8, t);
}
template <int EPI, int BM, int BN, int NTHR, int NPART>
__device__ __forceinline__ void wo_apply(PH_TL_PARAM const GemmParams& p, const float* cl, int m0, int n0, bf16x8 (&x)[WoCfg<BM, BN, NTHR>::IT],
                                         f32x4& b0, f32x4& b1) {
  constexpr int CH = BN / 4, IT = WoCfg<BM, BN, NTHR>::IT;
  // ONE wait for everything wo_prefetch requested (no store of this wave is in flight yet, so it waits for those reads only), then the
  // values pass through an empty asm: to hipcc they are plain registers from here on.  Without this its waitcnt pass answers every use
  // behind the conditional stores below with s_waitcnt vmcnt(0) -- the number of stores issued since the read is unknown to it -- and the
  // wave stalls on its own previous store in every step.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wo_settle(b0); wo_settle(b1);
  if constexpr (EPI == EPI_RES || EPI == EPI_SAVED) {
    static_for(std::make_integer_sequence<int, IT>{}, [&](auto uu) { wo_settle(x[decltype(uu)::value]); });
  }
  PH_TL(6);
  static_for(std::make_integer_sequence<int, IT>{}, [&](auto uu) {
    constexpr int u = decltype(uu)::value;
    const int id = u * NTHR + (int)threadIdx.x;
    const int ml = id / (CH / 2), c = (id % (CH / 2)) * 2;
    const int m = m0 + ml, n = n0 + c * 4;
    const int sw = ml & (CH - 1);
    f32x4 t0 = *reinterpret_cast<const f32x4*>(cl + ml * BN + ((c ^ sw) << 2));
    f32x4 t1 = *reinterpret_cast<const f32x4*>(cl + ml * BN + (((c + 1) ^ sw) << 2));
#pragma unroll
    for (int q = 1; q < NPART; ++q) {
      t0 += *reinterpret_cast<const f32x4*>(cl + q * BM * BN + ml * BN + ((c ^ sw) << 2));
      t1 += *reinterpret_cast<const f32x4*>(cl + q * BM * BN + ml * BN + (((c + 1) ^ sw) << 2));
    }
    float v[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
    const bool live = m < p.M && n < p.N;               // (N % 8 == 0 in every class: a live chunk is a whole chunk)
    if constexpr (EPI == EPI_PLAIN || EPI == EPI_RES || EPI == EPI_QGELU_GRAD || EPI == EPI_RELU2_GRAD) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
    }
    if constexpr (EPI == EPI_QGELU_GRAD || EPI == EPI_RELU2_GRAD) {
      bf16x8 gq;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float y, g;
        act_fwd_grad(EPI == EPI_QGELU_GRAD ? PH_ACT_QUICKGELU : PH_ACT_RELU2, v[e], y, g);
        v[e] = y; gq[e] = f2bf(g);
      }
      if (live) *reinterpret_cast<bf16x8*>(p.pre_out + (size_t)m * p.ldc + n) = gq;
    }
    if constexpr (EPI == EPI_SAVED) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= bf2f(x[u][e]);
    }
    if constexpr (EPI == EPI_RES) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bf2f(x[u][e]);
    }
    if constexpr (EPI == EPI_F32) {
      if (live) {
        float* cp = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
        f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
        *reinterpret_cast<f32x4*>(cp) = o0;
        *reinterpret_cast<f32x4*>(cp + 4) = o1;
      }
    } else {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
      if (live) *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.C) + (size_t)m * p.ldc + n) = o;
    }
    if (u == 0) PH_TL(7);
  });
}
// write-out of a parked tile by class (wave-uniform switch: one branch per tile)
template <int BM, int BN, int NTHR, int NPART = 1, int GCAP = 4>
__device__ __forceinline__ void tile_writeout(PH_TL_PARAM const int epi, const GemmParams& p, const float* cl, int m0, int n0, bool splitk, bool drop,
                                              const DropCtx& dc, bf16x8 (&x)[WoCfg<BM, BN, NTHR>::IT], f32x4& b0, f32x4& b1, int split_id = 0) {
  switch (epi) {
    case EPI_PLAIN: wo_apply<EPI_PLAIN, BM, BN, NTHR, NPART>(PH_TL_ARG p, cl, m0, n0, x, b0, b1); break;
    case EPI_RES: wo_apply<EPI_RES, BM, BN, NTHR, NPART>(PH_TL_ARG p, cl, m0, n0, x, b0, b1); break;
    case EPI_QGELU_GRAD: wo_apply<EPI_QGELU_GRAD, BM, BN, NTHR, NPART>(PH_TL_ARG p, cl, m0, n0, x, b0, b1); break;
    case EPI_RELU2_GRAD: wo_apply<EPI_RELU2_GRAD, BM, BN, NTHR, NPART>(PH_TL_ARG p, cl, m0, n0, x, b0, b1); break;
    case EPI_SAVED: wo_apply<EPI_SAVED, BM, BN, NTHR, NPART>(PH_TL_ARG p, cl, m0, n0, x, b0, b1); break;
    case EPI_F32: wo_apply<EPI_F32, BM, BN, NTHR, NPART>(PH_TL_ARG p, cl, m0, n0, x, b0, b1); break;
    default: tile_writeout_generic<BM, BN, NTHR, NPART, GCAP>(PH_TL_ARG p, cl, m0, n0, splitk, drop, dc, split_id); break;
  }
}

// BatchNorm statistics of a conv-as-GEMM output, taken from the tile while it is parked in LDS: per column the sum and the sum of
// squares of the bf16-ROUNDED values (what the next layer reads) over the tile's valid rows, added to col_stats[2][N] with one
// atomic pair per column and row half.  Plain epilogues only (the parked tile is alpha * acc: no bias / activation in a conv).
// The global accumulators are fp64: the variance is later formed as E[x^2] - E[x]^2, and in fp32 that difference (and the order
// of the atomics) is worth 1e-7 * x^2 -- visible against BatchNorm's eps = 1e-5 in channels that are constant over the batch
// (piecewise-constant label maps), where it made the step's loss wander by 2e-4 from run to run.
template <int BM, int BN, int NTHR>
__device__ __forceinline__ void tile_colstats(const GemmParams& p, const float* cl, int m0, int n0) {
  constexpr int CH = BN / 4, PARTS = NTHR / BN, RP = BM / PARTS;
  const int col = threadIdx.x % BN, part = threadIdx.x / BN;
  if (part >= PARTS || n0 + col >= p.N) return;
  const int rows = min(BM, p.M - m0);
  float s = 0.f, ss = 0.f;
  const int c4 = col >> 2, e = col & 3;
  for (int r = part * RP; r < min((part + 1) * RP, rows); ++r) {
    // (round 4 A/B: statistics of the fp32 accumulator values instead -- what the round-3 review suggested -- made the stems' gradients
    //  WORSE against the reference at bs32, profiles/r4_ab_bn_statistics.txt: the next layer normalises the ROUNDED values, and
    //  statistics that do not belong to them leave a per-channel offset; PyTorch's autocast also takes them from the bf16 conv output)
    float v = bf2f(f2bf(cl[r * BN + ((c4 ^ (r & (CH - 1))) << 2) + e]));
    s += v; ss += v * v;
  }
  // PH_COLSTAT_SLABS interleaved copies of the accumulators (slab = block id mod 8): a tall conv output (401408 x 96: 3136 tiles)
  // otherwise queues thousands of atomics on each of its 192 addresses (measured: 49 -> 158 us for that GEMM)
  double* st = p.col_stats + (size_t)(blockIdx.x % PH_COLSTAT_SLABS) * 2 * p.N;
  atomicAdd(st + n0 + col, (double)s);
  atomicAdd(st + p.N + n0 + col, (double)ss);
}

// Grouped launch: up to PH_GEMM_GROUP_MAX independent problems of one layout in ONE grid (block -> (problem, tile) through a
// prefix table in the kernel arguments).  The deferred weight-gradient GEMMs of a layer (outputs of 18..144 tiles each, far
// below the 512 block slots of the chip) are issued this way instead of one under-filled launch + split-K reduce apiece.
struct GroupParams {
  int n;
  int tile_start[PH_GEMM_GROUP_MAX + 1];   // first block of problem i (blocks of a problem = tiles x nsplit)
  int nsplit[PH_GEMM_GROUP_MAX];           // k splits of problem i (1 = none): block b of the problem is tile b % tiles, split b / tiles
  GemmParams p[PH_GEMM_GROUP_MAX];
};

// ---- process-wide state of the GEMM entry points (thread-safe: the C ABI may be entered from the host thread and from autograd's
// backward thread at once): kernel attributes are set through std::call_once; the tuning values ph_gemm_tuning() may change at run
// time are atomics.  (The library reads no environment variables: the round-1..3 A/B switches are gone, their measurements are in
// profiles/ and DESIGN.md.)
#define PH_SET_SMEM_ONCE(kernel_expr, bytes)                                                                                   \
  do {                                                                                                                         \
    static std::once_flag once__;                                                                                              \
    std::call_once(once__, [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_expr), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)); }); \
  } while (0)
// launches per kernel class since the last reset (ph_gemm_dispatch_counts): lets a test assert WHICH kernels a program ran through
enum { PH_GEMM_CLS_128 = 0, PH_GEMM_CLS_64, PH_GEMM_CLS_KS2, PH_GEMM_CLS_BIG, PH_GEMM_CLS_BIG_GROUPED, PH_GEMM_CLS_GROUPED,
       PH_GEMM_CLS_SPLITK_REDUCE, PH_GEMM_CLS_COUNT };
extern std::atomic<long long> g_gemm_counts[PH_GEMM_CLS_COUNT];
inline void count_launch(int cls) { g_gemm_counts[cls].fetch_add(1, std::memory_order_relaxed); if (g_ph_prof_enabled) ::g_ph_prof_last_cls = cls; }

// launchers of the register-staged kernels (gemm_kernels.h, instantiated by gemm_s128.hip / gemm_s64.hip / gemm_g128.hip / gemm_g64.hip)
namespace reg {
int launch_single_128(const GemmParams& p, int bn, int ta, int tb, int splits, hipStream_t s);      // 128x128 (bn = 128) or 128x64 tiles
int launch_single_64(const GemmParams& p, int ta, int tb, int splits, hipStream_t s);               // 64x64 tiles
int launch_ks2(const GemmParams& p, int tb, hipStream_t s);                                         // 64x64 tiles, k loop split inside the block
// pf: 0 = depth-1 schedule, 1 = prefetch ring; conv: 0 plain, 1 gathered A (NN layout), 2 gathered B (TT layout)
int launch_grouped_128(const GroupParams& g, int total, int max_blocks, int ta, int tb, int pf, int conv, hipStream_t s);
int launch_grouped_64(const GroupParams& g, int total, int max_blocks, int ta, int tb, int pf, int conv, hipStream_t s);
}  // namespace reg

// launchers of the 256x128 kernels (gemm_big.hip); variant: 0 plain main loop, 4 ping-pong, 5 ping-pong with the LEAN tail
namespace big {
constexpr int BM = 256, BN = 128;
int launch_single(const GemmParams& p, int variant, bool ta, bool tb, hipStream_t s);
int launch_grouped_wgrad(const GroupParams& g, int total, int variant, hipStream_t s);      // A = [K,M], B = [K,N], ping-pong, persistent grid
int launch_grouped_conv(const GroupParams& g, int blocks, int variant, hipStream_t s);       // forward-shaped implicit-GEMM convolutions, one block per tile
}  // namespace big

}  // namespace phg
