// Fused multi-head attention forward / backward for gfx950 (flash style: online softmax, no S x S matrix).
// Replaces: nn.MultiheadAttention core (vit.py:53, resampler.py:31) and RobertaSelfAttention
// scores -> +mask -> clamp -> softmax -> dropout -> PV (roberta.py:101-126), plus their autograd.
//
// Layout trick (wave64, MFMA 16x16x32 bf16): the score tile is computed TRANSPOSED,
//     S^T[key][q] = K . Q^T        (A = K rows from LDS, B = Q rows held in registers)
// so that every lane owns ONE query column (q = lane & 15) and 4 consecutive keys per 16-key sub-tile.
// Row max / row sum are then in-lane reductions plus two cross-lane shuffles (xor 16, 32), the softmax
// statistics are per-lane scalars, and P^T is already in MFMA B-operand form for
//     O^T[d][q] = V^T . P^T        (A = V^T fetched from the row-major V tile with ds_read_b64_tr_b16)
// with the key order inside a 32-key MFMA step permuted identically on both operands (a reduction index may be
// permuted freely).  No P round trip through LDS, no transposed V copy.
// The dK/dV kernel uses the mirror image (S[q][key] = Q . K^T, one key per lane) for the same reason.
//
// Block = 4 waves; forward / dQ: 64 queries per block (16 per wave), K/V streamed in 64-key tiles through a
// double-buffered LDS image; dK/dV: 64 keys per block, Q/dO streamed.
#include <float.h>
#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <type_traits>
#include <utility>
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

template <int... Is, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}

#ifndef PH_ATTN_PAD
#define PH_ATTN_PAD 8
#endif
#ifndef PH_DQ_WAVES            // waves per SIMD the register allocation of the 16-row backward kernels aims at (dh <= 64)
#define PH_DQ_WAVES 3
#endif
#ifndef PH_DKV_WAVES
#define PH_DKV_WAVES 3
#endif
template <int DH>
struct Cfg {
  static constexpr int KS = DH / 32;       // MFMA k-steps over the head dimension
  static constexpr int DT = DH / 16;       // 16-wide tiles over the head dimension
  static constexpr int RS = DH + PH_ATTN_PAD;   // LDS row stride in elements
  static constexpr int TILE = 64 * RS;     // elements per 64-row tile
  static constexpr int CPR = DH / 8;       // 16-B chunks per row
  static constexpr int NLD = 64 * CPR / 256 > 0 ? 64 * CPR / 256 : 1;   // chunks per thread per tile
};

// ---- tile staging: 64 rows x DH of a strided [token][head*dh] tensor -> registers -> LDS.
// Rows beyond n_rows re-read the last valid row (finite data; every consumer masks them by index): the loads carry no
// predicate, so the prefetch is straight-line code and the compiler's vmcnt bookkeeping stays exact around the tile loop.
// Tile loads (round 3) go through a buffer descriptor: address = descriptor base (the (batch, head) slice, SGPRs) + tile offset
// (SGPR soffset) + per-lane chunk offset (one 32-bit VGPR per chunk, computed once per kernel) -- zero address arithmetic in the
// tile loop, and rows beyond the end of the sequence read as zeros by the hardware bounds check instead of being clamped in
// software.  The 64-bit form this replaces cost 31-45 VALU instructions per tile, 8-12 of them quarter-rate v_mul_lo_u32, in
// kernels that are VALU-issue-bound (ISA count: ~900 VALU cycles against 256 MFMA cycles per wave and 64-key forward tile).
// The launcher checks that a slice spans < 2 GiB.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t tile_rsrc(const bf16* base, int64_t ts, int n_rows, int dh) {
  // readfirstlane makes the wave-uniformity of the descriptor provable (no waterfall loop around every load)
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  const uint32_t bytes = __builtin_amdgcn_readfirstlane((uint32_t)(((int64_t)(n_rows - 1) * ts + dh) * 2));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, bytes, 0x00020000);
}
template <int DH>
__device__ __forceinline__ void tile_offsets(int64_t ts, uint32_t (&off)[Cfg<DH>::NLD]) {
  constexpr int CPR = Cfg<DH>::CPR;
#pragma unroll
  for (int i = 0; i < Cfg<DH>::NLD; ++i) {
    int id = threadIdx.x + 256 * i;
    int r = id / CPR, c = id % CPR;
    off[i] = (uint32_t)((r * ts + c * 8) * 2);
  }
}
template <int DH>
__device__ __forceinline__ void tile_gload(rsrc_t rs, int64_t ts, int row0, const uint32_t (&off)[Cfg<DH>::NLD],
                                           u32x4 (&regs)[Cfg<DH>::NLD]) {
  const uint32_t soff = __builtin_amdgcn_readfirstlane((uint32_t)(row0 * ts * 2));
#pragma unroll
  for (int i = 0; i < Cfg<DH>::NLD; ++i)
    regs[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off[i], soff, 0));
}
template <int DH>
__device__ __forceinline__ void tile_lstore(bf16* lds, const u32x4 (&regs)[Cfg<DH>::NLD]) {
  constexpr int CPR = Cfg<DH>::CPR;
#pragma unroll
  for (int i = 0; i < Cfg<DH>::NLD; ++i) {
    int id = threadIdx.x + 256 * i;
    int r = id / CPR, c = id % CPR;
    *reinterpret_cast<u32x4*>(lds + r * Cfg<DH>::RS + c * 8) = regs[i];
  }
}

// row-major fragment: lane gets X[row0 + (l&15)][ks*32 + (l>>4)*8 .. +8]
template <int DH>
__device__ __forceinline__ bf16x8 frag_rows(const bf16* lds, int row0, int ks, int lane) {
  return *reinterpret_cast<const bf16x8*>(lds + (row0 + (lane & 15)) * Cfg<DH>::RS + ks * 32 + (lane >> 4) * 8);
}
// transposed fragment: lane (c = l&15, g = l>>4) gets X[kappa(g,j)][col0 + c], j = 0..7 with
// kappa(g,j) = kbase + 16*(j>>2) + 4*g + (j&3)   -- the key order produced by two adjacent 16-row C tiles.
template <int DH>
__device__ __forceinline__ bf16x8 frag_tr(const bf16* lds, int kbase, int col0, int lane) {
  int g = lane >> 4, i = lane & 15;
  const bf16* p = lds + (kbase + 4 * g + (i >> 2)) * Cfg<DH>::RS + col0 + (i & 3) * 4;
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 16 * Cfg<DH>::RS));
  bf16x8 o;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
  o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
  return o;
}
__device__ __forceinline__ bf16x8 pack2(const f32x4& a, const f32x4& b) {
  bf16x8 o;
  o[0] = f2bf(a[0]); o[1] = f2bf(a[1]); o[2] = f2bf(a[2]); o[3] = f2bf(a[3]);
  o[4] = f2bf(b[0]); o[5] = f2bf(b[1]); o[6] = f2bf(b[2]); o[7] = f2bf(b[3]);
  return o;
}
__device__ __forceinline__ float xor_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xor_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

#define NEG_MASK (-FLT_MAX)

// Values fetched from global memory BEFORE a tile loop (the resident Q / dO / K / V fragments, lse): passing them through
// an empty asm once makes them plain register values.  Without it the compiler keeps "may still be in flight" state for
// them around the loop back-edge and emits s_waitcnt vmcnt(0..1) in front of the first MFMAs of every iteration, i.e. it
// drains the next tile's prefetch right after issuing it (seen in the ISA: the prefetch never overlapped the math).
__device__ __forceinline__ void settle(bf16x8& v) {
  u32x4 t = __builtin_bit_cast(u32x4, v);
  asm volatile("" : "+v"(t));
  v = __builtin_bit_cast(bf16x8, t);
}
__device__ __forceinline__ void settle(float& v) { asm volatile("" : "+v"(v)); }

// per-key state of a 64-key tile, staged through LDS with the tile (one float per key, handled by threads 0..63):
//   0 = live key, NEG_MASK = excluded by key_mask, -inf = beyond Sk
// The raw mask byte is prefetched with the tile and only interpreted at staging time (no dependent use -> no early wait).
__device__ __forceinline__ uint32_t key_raw(const uint8_t* km, int ki, int Sk) { return km ? (uint32_t)km[min(ki, Sk - 1)] : 1u; }
__device__ __forceinline__ float key_state(uint32_t raw, int ki, int Sk) {
  if (ki >= Sk) return -INFINITY;
  return raw ? 0.f : NEG_MASK;
}
__device__ __forceinline__ float score_of(float s, float scale, float kstate, bool causal_cut) {
  float sc = causal_cut ? NEG_MASK : s * scale;
  return kstate == 0.f ? sc : kstate;
}

// PLAIN kernels (round 3): no causal cut, no key mask, no probability dropout -- the ViT blocks and the resampler, i.e. the bulk of
// the attention time.  The generic kernels spend ~15 VALU instructions per score element (scale, two mask selects with their index
// arithmetic, subtract, multiply by log2(e), exp) against one MFMA per 4 elements: by counter they sat at 6-10 % MFMA utilisation,
// VALU-bound.  The PLAIN path folds scale * log2(e) into one constant and works in base 2 (p = exp2(fma(s, c, -m)): two
// instructions per element); only a tile that crosses the end of the sequence pays a bounds select.
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
#ifdef PH_ATTN_NOEXP           // timing diagnostic (tools/build_variant.py attention.hip:-DPH_ATTN_NOEXP): WRONG results, the exponential replaced by one fma
__device__ __forceinline__ float fast_exp2(float x) { return fmaf(x, 1e-3f, 1.0f); }
#else
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
#endif

// =====================================================================================================
// forward
// =====================================================================================================
// LDS image (all three kernels): [stage 0: tile X, tile Y][stage 1: tile X, tile Y][per-row fp32 side data, 2 x 128]
//   forward / dQ : X = K, Y = V, side = key state (64 floats per stage)
//   dK/dV        : X = Q, Y = dO, side = lse (64) + delta (64) per stage
// XCD-aware block placement (round 3).  Hardware puts block L of a 1-D grid on XCD L % 8, each XCD with its own L2.  With the plain
// (x = tile of the sequence, y = head) grid the 5 query blocks of a ViT head landed on 5 different XCDs, i.e. every head's K / V was
// pulled into five L2s.  The linear id is re-mapped so that an XCD walks whole heads: slot = L / 8, head = (slot / nx) * 8 + L % 8,
// x = slot % nx (heads beyond the last multiple of 8 keep the plain order).
struct BlockXY { int x, y; };
__device__ __forceinline__ BlockXY block_xy(int nx, int ny) {
  const int L = blockIdx.x;
  const int full = (ny / 8) * 8 * nx;                  // blocks of the heads that fill whole groups of 8
  if (L < full) {
    const int xcd = L & 7, slot = L >> 3;
    return BlockXY{slot % nx, (slot / nx) * 8 + xcd};
  }
  const int r = L - full;
  return BlockXY{r % nx, (ny / 8) * 8 + r / nx};
}

// QT (round 3): 16-query sub-tiles per wave.  QT = 2 -> a wave owns 32 queries: every K / V fragment fetched from LDS feeds two
// MFMAs instead of one (the kernels were LDS-bound: ~16 KB of fragment reads per wave and 64-key tile against 256 cycles of MFMA),
// and a block covers 128 queries, so the K / V tiles of a head are staged by half as many blocks.
template <int DH, bool PLAIN, int QT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DH <= 64 ? (QT == 1 ? 4 : 2) : 2))) void attn_fwd_kernel(ph_attn_fwd_args a) {
  using C = Cfg<DH>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16* smem = reinterpret_cast<bf16*>(smem_raw);
  float* side = reinterpret_cast<float*>(smem + 4 * C::TILE);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const BlockXY bxy = block_xy((a.Sq + 64 * QT - 1) / (64 * QT), a.B * a.H);
  const int b = bxy.y / a.H, h = bxy.y % a.H;
  const int q0 = bxy.x * (64 * QT) + wave * (16 * QT);
  const bool wave_live = q0 < a.Sq;                 // a wave whose queries are all padding only helps staging
  PH_TL_DECL;
  PH_TL(0);
  const bf16* Q = reinterpret_cast<const bf16*>(a.q) + b * a.q_bs + (int64_t)h * DH;
  const bf16* K = reinterpret_cast<const bf16*>(a.k) + b * a.k_bs + (int64_t)h * DH;
  const bf16* V = reinterpret_cast<const bf16*>(a.v) + b * a.v_bs + (int64_t)h * DH;
  const uint8_t* km = a.key_mask ? a.key_mask + (int64_t)b * a.Sk : nullptr;

  int qi[QT];
  bf16x8 qf[QT][C::KS];
  f32x4 o[QT][C::DT];
  float m[QT], lsum[QT];
  uint32_t rowid[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    qi[t] = q0 + t * 16 + c;
    const int qr = qi[t] < a.Sq ? qi[t] : a.Sq - 1;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks)
      qf[t][ks] = *reinterpret_cast<const bf16x8*>(Q + (int64_t)qr * a.q_ts + ks * 32 + g * 8);
#pragma unroll
    for (int d = 0; d < C::DT; ++d) o[t][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    m[t] = NEG_MASK; lsum[t] = 0.f;
    rowid[t] = (uint32_t)((b * a.H + h) * a.Sq + qr);
  }
  DropCtx dc;
  const bool drop = !PLAIN && a.drop_p > 0.f;
  if (drop) dc = make_drop(a.drop_seed, a.drop_stream, a.drop_p);

  const int ntiles = (a.Sk + 63) / 64;
  u32x4 rk[C::NLD], rv[C::NLD];
  uint32_t kraw = 1u;
  int kraw_i = threadIdx.x;
  uint32_t koff[C::NLD], voff[C::NLD];
  tile_offsets<DH>(a.k_ts, koff);
  tile_offsets<DH>(a.v_ts, voff);
  const rsrc_t krs = tile_rsrc(K, a.k_ts, a.Sk, DH), vrs = tile_rsrc(V, a.v_ts, a.Sk, DH);
  tile_gload<DH>(krs, a.k_ts, 0, koff, rk);
  tile_gload<DH>(vrs, a.v_ts, 0, voff, rv);
  if (!PLAIN && threadIdx.x < 64) kraw = key_raw(km, kraw_i, a.Sk);
#pragma unroll
  for (int t = 0; t < QT; ++t)
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) settle(qf[t][ks]);
  PH_TL(1);
  tile_lstore<DH>(smem, rk);
  tile_lstore<DH>(smem + C::TILE, rv);
  if (!PLAIN && threadIdx.x < 64) side[threadIdx.x] = key_state(kraw, kraw_i, a.Sk);
  __syncthreads();
  PH_TL(2);
  int cur = 0;
  for (int tl = 0; tl < ntiles; ++tl) {
    const bool more = tl + 1 < ntiles;
    if (more) {
      tile_gload<DH>(krs, a.k_ts, (tl + 1) * 64, koff, rk);
      tile_gload<DH>(vrs, a.v_ts, (tl + 1) * 64, voff, rv);
      kraw_i = (tl + 1) * 64 + threadIdx.x;
      if (!PLAIN && threadIdx.x < 64) kraw = key_raw(km, kraw_i, a.Sk);
    }
    const bf16* kl = smem + cur * 2 * C::TILE;
    const bf16* vl = kl + C::TILE;
    const float* kst = side + cur * 128;
    const int kbase = tl * 64;
    // TAIL = false: a PLAIN tile entirely inside the key sequence -- no bounds selects, no sub-tile guards (they cost more VALU
    // issue slots than the softmax itself: 51 v_cndmask + 18 v_cmp per 16 MFMAs in the first PLAIN build)
    auto tile_body = [&](auto tail_c) {
      constexpr bool TAIL = decltype(tail_c)::value;
      f32x4 s[QT][4];
      float mx[QT];
#pragma unroll
      for (int t = 0; t < QT; ++t) mx[t] = -INFINITY;
      // fragment prefetch (interior tiles): ALL K and V^T fragments of the tile are requested up front -- the 24 LDS round trips then
      // overlap the QK^T MFMAs and the softmax arithmetic instead of each MFMA waiting for its own operand (by counter the waves sat
      // in s_waitcnt 46 % of their cycles)
      bf16x8 kpre[TAIL ? 1 : 4][C::KS], vpre[TAIL ? 1 : 2][C::DT];
      if constexpr (!TAIL) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int ks = 0; ks < C::KS; ++ks) kpre[nt][ks] = frag_rows<DH>(kl, nt * 16, ks, lane);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int d = 0; d < C::DT; ++d) vpre[k2][d] = frag_tr<DH>(vl, k2 * 32, d * 16, lane);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if (!TAIL || kbase + nt * 16 < a.Sk) {        // 16-key sub-tiles entirely beyond Sk cost nothing
          f32x4 acc[QT];
#pragma unroll
          for (int t = 0; t < QT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < C::KS; ++ks) {
            bf16x8 kfr;
            if constexpr (TAIL) kfr = frag_rows<DH>(kl, nt * 16, ks, lane); else kfr = kpre[nt][ks];
#pragma unroll
            for (int t = 0; t < QT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[t][ks], acc[t], 0, 0, 0);
          }
          if constexpr (PLAIN) {                      // raw scores; the scale (> 0) is applied inside the exponent's fma below
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                if (TAIL) acc[t][r] = (kbase + nt * 16 + g * 4 + r < a.Sk) ? acc[t][r] : -INFINITY;
                mx[t] = fmaxf(mx[t], acc[t][r]);
              }
          } else {
            const f32x4 st = *reinterpret_cast<const f32x4*>(kst + nt * 16 + g * 4);
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                int ki = kbase + nt * 16 + g * 4 + r;
                acc[t][r] = score_of(acc[t][r], a.scale, st[r], a.causal && ki > qi[t]);
                mx[t] = fmaxf(mx[t], acc[t][r]);
              }
          }
#pragma unroll
          for (int t = 0; t < QT; ++t) s[t][nt] = acc[t];
        } else {
#pragma unroll
          for (int t = 0; t < QT; ++t) s[t][nt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
      }
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        // PLAIN: m is kept in base-2 units of the scaled score, p = exp2(fma(s_raw, scale * log2(e), -m)); the arithmetic is written on
        // float pairs so that it issues as v_pk_fma_f32 / v_pk_add_f32 (the kernel is VALU-issue-bound, not MFMA- or LDS-bound)
        const float c2 = a.scale * LOG2E;
        const float mxt = PLAIN ? xor_max(mx[t]) * c2 : xor_max(mx[t]);
        const float m_new = fmaxf(m[t], mxt);
        const float alpha = PLAIN ? fast_exp2(m[t] - m_new) : __expf(m[t] - m_new);
        float rs = 0.f;
        if constexpr (PLAIN) {
          const f32x2 c2v = {c2, c2}, nm = {-m_new, -m_new};
          f32x2 rs2 = {0.f, 0.f};
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
              f32x2 x = {s[t][nt][r], s[t][nt][r + 1]};
              x = __builtin_elementwise_fma(x, c2v, nm);
              f32x2 p2 = {fast_exp2(x[0]), fast_exp2(x[1])};
              rs2 += p2;
              s[t][nt][r] = p2[0];
              s[t][nt][r + 1] = p2[1];
            }
          rs = rs2[0] + rs2[1];
        } else {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float p = __expf(s[t][nt][r] - m_new);
              rs += p;
              s[t][nt][r] = p;
            }
        }
        rs = xor_sum(rs);
        lsum[t] = lsum[t] * alpha + rs;
        m[t] = m_new;
#pragma unroll
        for (int d = 0; d < C::DT; ++d) o[t][d] *= alpha;
        if (drop) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            int k4 = (kbase + nt * 16 + g * 4) >> 2;
            u32x4 rnd = philox4x32((uint32_t)k4, rowid[t], dc.stream, 0xa77eu, dc.k0, dc.k1);
#pragma unroll
            for (int r = 0; r < 4; ++r) s[t][nt][r] = drop_apply(dc, rnd[r], s[t][nt][r]);
          }
        }
      }
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        if (!TAIL || kbase + k2 * 32 < a.Sk) {
          bf16x8 pf[QT];
#pragma unroll
          for (int t = 0; t < QT; ++t) pf[t] = pack2(s[t][2 * k2], s[t][2 * k2 + 1]);
#pragma unroll
          for (int d = 0; d < C::DT; ++d) {
            bf16x8 vfr;
            if constexpr (TAIL) vfr = frag_tr<DH>(vl, k2 * 32, d * 16, lane); else vfr = vpre[k2][d];
#pragma unroll
            for (int t = 0; t < QT; ++t) o[t][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr, pf[t], o[t][d], 0, 0, 0);
          }
        }
      }
    };
    if (wave_live) {
      if (!PLAIN || kbase + 64 > a.Sk) tile_body(std::true_type{}); else tile_body(std::false_type{});
    }
    if (more) {
      tile_lstore<DH>(smem + (cur ^ 1) * 2 * C::TILE, rk);
      tile_lstore<DH>(smem + (cur ^ 1) * 2 * C::TILE + C::TILE, rv);
      if (!PLAIN && threadIdx.x < 64) side[(cur ^ 1) * 128 + threadIdx.x] = key_state(kraw, kraw_i, a.Sk);
    }
    __syncthreads();
    cur ^= 1;
  }
  PH_TL(3);
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    if (qi[t] < a.Sq) {
      float inv = 1.0f / lsum[t];
      bf16* O = reinterpret_cast<bf16*>(a.o) + b * a.o_bs + (int64_t)qi[t] * a.o_ts + (int64_t)h * DH;
#pragma unroll
      for (int d = 0; d < C::DT; ++d) {
        bf16x4 tt = {f2bf(o[t][d][0] * inv), f2bf(o[t][d][1] * inv), f2bf(o[t][d][2] * inv), f2bf(o[t][d][3] * inv)};
        *reinterpret_cast<bf16x4*>(O + d * 16 + g * 4) = tt;
      }
      if (g == 0 && a.lse) a.lse[(int64_t)(b * a.H + h) * a.Sq + qi[t]] = (PLAIN ? m[t] * LN2 : m[t]) + __logf(lsum[t]);    // natural-log lse either way
    }
  }
#ifdef PH_TIMELINE
  PH_TL(8);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PH_TL(9);
  PH_TL_FLUSH((int)blockIdx.x, 0, threadIdx.x == 0);
#endif
}

// =====================================================================================================
// backward: dQ  (same streaming structure as forward)
// =====================================================================================================
template <int DH, bool PLAIN, int QT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DH <= 64 ? (QT == 1 ? PH_DQ_WAVES : 2) : 2))) void attn_bwd_dq_kernel(ph_attn_bwd_args a) {
  using C = Cfg<DH>;
  const ph_attn_fwd_args& f = a.f;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16* smem = reinterpret_cast<bf16*>(smem_raw);
  float* side = reinterpret_cast<float*>(smem + 4 * C::TILE);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const BlockXY bxy = block_xy((f.Sq + 64 * QT - 1) / (64 * QT), f.B * f.H);
  const int b = bxy.y / f.H, h = bxy.y % f.H;
  const int q0 = bxy.x * (64 * QT) + wave * (16 * QT);
  const bool wave_live = q0 < f.Sq;
  const bf16* Q = reinterpret_cast<const bf16*>(f.q) + b * f.q_bs + (int64_t)h * DH;
  const bf16* K = reinterpret_cast<const bf16*>(f.k) + b * f.k_bs + (int64_t)h * DH;
  const bf16* V = reinterpret_cast<const bf16*>(f.v) + b * f.v_bs + (int64_t)h * DH;
  const bf16* dO = reinterpret_cast<const bf16*>(a.d_o) + b * a.do_bs + (int64_t)h * DH;
  const uint8_t* km = f.key_mask ? f.key_mask + (int64_t)b * f.Sk : nullptr;

  int qi[QT];
  int64_t ridx[QT];
  bf16x8 qf[QT][C::KS], dof[QT][C::KS];
  float lse[QT], delta[QT], dsum[QT];
  f32x4 dq[QT][C::DT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    qi[t] = q0 + t * 16 + c;
    const int qr = qi[t] < f.Sq ? qi[t] : f.Sq - 1;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      qf[t][ks] = *reinterpret_cast<const bf16x8*>(Q + (int64_t)qr * f.q_ts + ks * 32 + g * 8);
      dof[t][ks] = *reinterpret_cast<const bf16x8*>(dO + (int64_t)qr * a.do_ts + ks * 32 + g * 8);
    }
    ridx[t] = (int64_t)(b * f.H + h) * f.Sq + qr;
    lse[t] = f.lse[ridx[t]];
    delta[t] = 0.f; dsum[t] = 0.f;
#pragma unroll
    for (int d = 0; d < C::DT; ++d) dq[t][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  DropCtx dc;
  const bool drop = !PLAIN && f.drop_p > 0.f;
  if (drop) dc = make_drop(f.drop_seed, f.drop_stream, f.drop_p);

  // Two sweeps over the K/V tiles, run as ONE loop of 2*ntiles steps so that the prefetch never drains in between.
  // Sweep 0 computes delta_i = sum_j P_ij * dP_ij in fp32 from the SAME recomputed P and dP sweep 1 uses (softmax backward
  // needs dP_ij - delta_i, which cancels to ~0 on peaked rows; taking delta from the bf16-rounded forward output O,
  // rowsum(dO*O), leaves a 2^-9 relative inconsistency that shows up as 10 %-level noise in dQ/dK of peaked attention
  // rows).  Sweep 1 forms dS and accumulates dQ.
  const int ntiles = (f.Sk + 63) / 64;
  const int nsteps = 2 * ntiles;
  u32x4 rk[C::NLD], rv[C::NLD];
  uint32_t kraw = 1u;
  int kraw_i = threadIdx.x;
  uint32_t koff[C::NLD], voff[C::NLD];
  tile_offsets<DH>(f.k_ts, koff);
  tile_offsets<DH>(f.v_ts, voff);
  const rsrc_t krs = tile_rsrc(K, f.k_ts, f.Sk, DH), vrs = tile_rsrc(V, f.v_ts, f.Sk, DH);
  tile_gload<DH>(krs, f.k_ts, 0, koff, rk);
  tile_gload<DH>(vrs, f.v_ts, 0, voff, rv);
  if (!PLAIN && threadIdx.x < 64) kraw = key_raw(km, kraw_i, f.Sk);
#pragma unroll
  for (int t = 0; t < QT; ++t) {
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) { settle(qf[t][ks]); settle(dof[t][ks]); }
    settle(lse[t]);
  }
  tile_lstore<DH>(smem, rk);
  tile_lstore<DH>(smem + C::TILE, rv);
  if (!PLAIN && threadIdx.x < 64) side[threadIdx.x] = key_state(kraw, kraw_i, f.Sk);
  __syncthreads();
  int cur = 0;
  for (int u = 0; u < nsteps; ++u) {
    const bool sweep1 = u >= ntiles;
    const int tl = sweep1 ? u - ntiles : u;
    const bool more = u + 1 < nsteps;
    if (more) {
      const int tn = (tl + 1 == ntiles) ? 0 : tl + 1;
      tile_gload<DH>(krs, f.k_ts, tn * 64, koff, rk);
      tile_gload<DH>(vrs, f.v_ts, tn * 64, voff, rv);
      kraw_i = tn * 64 + threadIdx.x;
      if (!PLAIN && threadIdx.x < 64) kraw = key_raw(km, kraw_i, f.Sk);
    }
    const bf16* kl = smem + cur * 2 * C::TILE;
    const bf16* vl = kl + C::TILE;
    const float* kst = side + cur * 128;
    const int kbase = tl * 64;
    auto tile_body = [&](auto tail_c) {                  // TAIL = false: PLAIN tile inside the key sequence (see attn_fwd_kernel)
      constexpr bool TAIL = decltype(tail_c)::value;
      f32x4 ds[QT][4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if (!TAIL || kbase + nt * 16 < f.Sk) {
          f32x4 acc[QT], dp[QT];
#pragma unroll
          for (int t = 0; t < QT; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
          for (int ks = 0; ks < C::KS; ++ks) {
            const bf16x8 kfr = frag_rows<DH>(kl, nt * 16, ks, lane);
            const bf16x8 vfr = frag_rows<DH>(vl, nt * 16, ks, lane);
#pragma unroll
            for (int t = 0; t < QT; ++t) {
              acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[t][ks], acc[t], 0, 0, 0);
              dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr, dof[t][ks], dp[t], 0, 0, 0);
            }
          }
          if constexpr (PLAIN) {
            const float c2 = f.scale * LOG2E;
#pragma unroll
            for (int t = 0; t < QT; ++t) {
              const float l2 = lse[t] * LOG2E;
              const f32x2 c2v = {c2, c2}, nl = {-l2, -l2}, dl2 = {delta[t], delta[t]};
              f32x2 part = {0.f, 0.f};
#pragma unroll
              for (int r = 0; r < 4; r += 2) {                // float pairs: v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32
                f32x2 x = {acc[t][r], acc[t][r + 1]};
                x = __builtin_elementwise_fma(x, c2v, nl);
                f32x2 p2 = {fast_exp2(x[0]), fast_exp2(x[1])};
                if (TAIL) {
                  p2[0] = (kbase + nt * 16 + g * 4 + r < f.Sk) ? p2[0] : 0.f;
                  p2[1] = (kbase + nt * 16 + g * 4 + r + 1 < f.Sk) ? p2[1] : 0.f;
                }
                const f32x2 dd = {dp[t][r], dp[t][r + 1]};
                part = __builtin_elementwise_fma(p2, dd, part);
                const f32x2 dv2 = p2 * (dd - dl2);
                ds[t][nt][r] = dv2[0];
                ds[t][nt][r + 1] = dv2[1];
              }
              dsum[t] += part[0] + part[1];
            }
          } else {
            const f32x4 st = *reinterpret_cast<const f32x4*>(kst + nt * 16 + g * 4);
#pragma unroll
            for (int t = 0; t < QT; ++t) {
              u32x4 rnd;
              if (drop) rnd = philox4x32((uint32_t)((kbase + nt * 16 + g * 4) >> 2), (uint32_t)ridx[t], dc.stream, 0xa77eu, dc.k0, dc.k1);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                int ki = kbase + nt * 16 + g * 4 + r;
                float p = __expf(score_of(acc[t][r], f.scale, st[r], f.causal && ki > qi[t]) - lse[t]);
                float dpe = drop ? drop_apply(dc, rnd[r], dp[t][r]) : dp[t][r];
                dsum[t] += p * dpe;
                ds[t][nt][r] = p * (dpe - delta[t]);
              }
            }
          }
        } else {
#pragma unroll
          for (int t = 0; t < QT; ++t) ds[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      if (sweep1) {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          if (!TAIL || kbase + k2 * 32 < f.Sk) {
            bf16x8 dsf[QT];
#pragma unroll
            for (int t = 0; t < QT; ++t) dsf[t] = pack2(ds[t][2 * k2], ds[t][2 * k2 + 1]);
#pragma unroll
            for (int d = 0; d < C::DT; ++d) {
              const bf16x8 kfr = frag_tr<DH>(kl, k2 * 32, d * 16, lane);
#pragma unroll
              for (int t = 0; t < QT; ++t) dq[t][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, dsf[t], dq[t][d], 0, 0, 0);
            }
          }
        }
      }
      if (u == ntiles - 1) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          delta[t] = xor_sum(dsum[t]);
          if (g == 0 && qi[t] < f.Sq) a.delta[ridx[t]] = delta[t];    // consumed by the dK/dV kernel (launched after this one)
        }
      }
    };
    if (wave_live) {
      if (!PLAIN || kbase + 64 > f.Sk) tile_body(std::true_type{}); else tile_body(std::false_type{});
    }
    if (more) {
      tile_lstore<DH>(smem + (cur ^ 1) * 2 * C::TILE, rk);
      tile_lstore<DH>(smem + (cur ^ 1) * 2 * C::TILE + C::TILE, rv);
      if (!PLAIN && threadIdx.x < 64) side[(cur ^ 1) * 128 + threadIdx.x] = key_state(kraw, kraw_i, f.Sk);
    }
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    if (qi[t] < f.Sq) {
      bf16* dQ = reinterpret_cast<bf16*>(a.dq) + b * a.dq_bs + (int64_t)qi[t] * a.dq_ts + (int64_t)h * DH;
#pragma unroll
      for (int d = 0; d < C::DT; ++d) {
        bf16x4 tt = {f2bf(dq[t][d][0] * f.scale), f2bf(dq[t][d][1] * f.scale), f2bf(dq[t][d][2] * f.scale), f2bf(dq[t][d][3] * f.scale)};
        *reinterpret_cast<bf16x4*>(dQ + d * 16 + g * 4) = tt;
      }
    }
  }
}

// word j of the 4-word value held by lane R of this lane's quad (DPP quad_perm broadcast: no LDS, no wait)
template <int R>
__device__ __forceinline__ uint32_t quad_word(const u32x4& v, int j) {
  constexpr int ctrl = R | (R << 2) | (R << 4) | (R << 6);
  const uint32_t w0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)v[0], ctrl, 0xf, 0xf, true);
  const uint32_t w1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)v[1], ctrl, 0xf, 0xf, true);
  const uint32_t w2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)v[2], ctrl, 0xf, 0xf, true);
  const uint32_t w3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)v[3], ctrl, 0xf, 0xf, true);
  return j == 0 ? w0 : (j == 1 ? w1 : (j == 2 ? w2 : w3));
}

// =====================================================================================================
// backward: dK, dV  (one key per lane and sub-tile; Q / dO streamed in 64-query tiles)
// =====================================================================================================
template <int DH, bool PLAIN, int QT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DH <= 64 ? (QT == 1 ? PH_DKV_WAVES : 2) : 2))) void attn_bwd_dkv_kernel(ph_attn_bwd_args a) {
  using C = Cfg<DH>;
  const ph_attn_fwd_args& f = a.f;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16* smem = reinterpret_cast<bf16*>(smem_raw);
  float* side = reinterpret_cast<float*>(smem + 4 * C::TILE);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const BlockXY bxy = block_xy((f.Sk + 64 * QT - 1) / (64 * QT), f.B * f.H);
  const int b = bxy.y / f.H, h = bxy.y % f.H;
  const int k0 = bxy.x * (64 * QT) + wave * (16 * QT);
  const bool wave_live = k0 < f.Sk;
  const bf16* Q = reinterpret_cast<const bf16*>(f.q) + b * f.q_bs + (int64_t)h * DH;
  const bf16* K = reinterpret_cast<const bf16*>(f.k) + b * f.k_bs + (int64_t)h * DH;
  const bf16* V = reinterpret_cast<const bf16*>(f.v) + b * f.v_bs + (int64_t)h * DH;
  const bf16* dO = reinterpret_cast<const bf16*>(a.d_o) + b * a.do_bs + (int64_t)h * DH;
  const uint8_t* km = f.key_mask ? f.key_mask + (int64_t)b * f.Sk : nullptr;

  int ki[QT];
  bf16x8 kf[QT][C::KS], vf[QT][C::KS];
  float kstate[QT];
  f32x4 dk[QT][C::DT], dv[QT][C::DT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    ki[t] = k0 + t * 16 + c;
    const int kr = ki[t] < f.Sk ? ki[t] : f.Sk - 1;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      kf[t][ks] = *reinterpret_cast<const bf16x8*>(K + (int64_t)kr * f.k_ts + ks * 32 + g * 8);
      vf[t][ks] = *reinterpret_cast<const bf16x8*>(V + (int64_t)kr * f.v_ts + ks * 32 + g * 8);
    }
    kstate[t] = key_raw(km, kr, f.Sk) ? 0.f : 1.f;   // this lane's key: excluded by key_mask?
#pragma unroll
    for (int d = 0; d < C::DT; ++d) { dk[t][d] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[t][d] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }
  DropCtx dc;
  const bool drop = !PLAIN && f.drop_p > 0.f;
  if (drop) dc = make_drop(f.drop_seed, f.drop_stream, f.drop_p);
  const float* lse_base = f.lse + (int64_t)(b * f.H + h) * f.Sq;
  const float* delta_base = a.delta + (int64_t)(b * f.H + h) * f.Sq;
  // per-query softmax statistics of a tile travel with it: threads 0..63 fetch lse, 64..127 delta
  const float* stat_base = threadIdx.x < 64 ? lse_base : delta_base;
  const int stat_row = threadIdx.x & 63;

  const int ntiles = (f.Sq + 63) / 64;
  u32x4 rq[C::NLD], rd[C::NLD];
  float sreg = 0.f;
  uint32_t qoff[C::NLD], dooff[C::NLD];
  tile_offsets<DH>(f.q_ts, qoff);
  tile_offsets<DH>(a.do_ts, dooff);
  const rsrc_t qrs = tile_rsrc(Q, f.q_ts, f.Sq, DH), dors = tile_rsrc(dO, a.do_ts, f.Sq, DH);
  tile_gload<DH>(qrs, f.q_ts, 0, qoff, rq);
  tile_gload<DH>(dors, a.do_ts, 0, dooff, rd);
  if (threadIdx.x < 128) sreg = stat_base[min(stat_row, f.Sq - 1)];
  bool key_masked[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) { settle(kf[t][ks]); settle(vf[t][ks]); }
    settle(kstate[t]);
    key_masked[t] = kstate[t] != 0.f;
  }
  tile_lstore<DH>(smem, rq);
  tile_lstore<DH>(smem + C::TILE, rd);
  // PLAIN: the lse half of the side data is staged as -lse * log2(e), the addend of the exponent's fma (one multiply per row and
  // tile instead of one per score element in every wave)
  const float stat_mul = (PLAIN && threadIdx.x < 64) ? -LOG2E : 1.f;
  if (threadIdx.x < 128) side[threadIdx.x] = sreg * stat_mul;
  __syncthreads();
  int cur = 0;
  for (int tl = 0; tl < ntiles; ++tl) {
    const bool more = tl + 1 < ntiles;
    if (more) {
      tile_gload<DH>(qrs, f.q_ts, (tl + 1) * 64, qoff, rq);
      tile_gload<DH>(dors, a.do_ts, (tl + 1) * 64, dooff, rd);
      if (threadIdx.x < 128) sreg = stat_base[min((tl + 1) * 64 + stat_row, f.Sq - 1)];
    }
    const bf16* ql = smem + cur * 2 * C::TILE;
    const bf16* dl = ql + C::TILE;
    const float* stl = side + cur * 128;
    const int qbase = tl * 64;
    auto tile_body = [&](auto tail_c) {                  // TAIL = false: PLAIN tile pair without padding rows / keys
      constexpr bool TAIL = decltype(tail_c)::value;
      f32x4 pd[QT][4], ds[QT][4];
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        if (!TAIL || qbase + qt * 16 < f.Sq) {
          f32x4 acc[QT], dp[QT];
#pragma unroll
          for (int t = 0; t < QT; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
          for (int ks = 0; ks < C::KS; ++ks) {
            const bf16x8 qfr = frag_rows<DH>(ql, qt * 16, ks, lane);
            const bf16x8 dfr = frag_rows<DH>(dl, qt * 16, ks, lane);
#pragma unroll
            for (int t = 0; t < QT; ++t) {
              acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[t][ks], acc[t], 0, 0, 0);
              dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dfr, vf[t][ks], dp[t], 0, 0, 0);
            }
          }
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(stl + qt * 16 + g * 4);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(stl + 64 + qt * 16 + g * 4);
          if constexpr (PLAIN) {
            const float c2 = f.scale * LOG2E;
            const f32x2 c2v = {c2, c2};
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
              for (int r = 0; r < 4; r += 2) {                // float pairs: v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32
                const f32x2 nl = {l4[r], l4[r + 1]}, dl2 = {d4[r], d4[r + 1]};
                f32x2 x = {acc[t][r], acc[t][r + 1]};
                x = __builtin_elementwise_fma(x, c2v, nl);
                f32x2 p2 = {fast_exp2(x[0]), fast_exp2(x[1])};
                if (TAIL) {
                  p2[0] = (qbase + qt * 16 + g * 4 + r < f.Sq && ki[t] < f.Sk) ? p2[0] : 0.f;
                  p2[1] = (qbase + qt * 16 + g * 4 + r + 1 < f.Sq && ki[t] < f.Sk) ? p2[1] : 0.f;
                }
                f32x2 dd = {dp[t][r], dp[t][r + 1]};
                dd = p2 * (dd - dl2);
                pd[t][qt][r] = p2[0];
                pd[t][qt][r + 1] = p2[1];
                ds[t][qt][r] = dd[0];
                ds[t][qt][r + 1] = dd[1];
              }
          } else {
#pragma unroll
            for (int t = 0; t < QT; ++t) {
              // Dropout words of this lane's 4 elements (one key, queries g*4 .. g*4+3).  The mask is defined per ROW: word (key & 3) of
              // philox(counter = (key >> 2, row id)), which suits the forward / dQ layout (one query, 4 consecutive keys per lane: one call).
              // Here the four lanes of a quad hold keys 4j .. 4j+3 of the same four rows, i.e. they need the SAME four calls and one word of
              // each: lane q of the quad generates the call of row q, the words travel by DPP quad broadcast (round 4: one call per lane and
              // 4 elements instead of four -- the generator was most of this kernel's time in the decoder).
              uint32_t rr4[4] = {0u, 0u, 0u, 0u};
              if (drop) {
                const uint32_t row_me = (uint32_t)((b * f.H + h) * f.Sq + qbase + qt * 16 + g * 4 + (lane & 3));
                const u32x4 mine = philox4x32((uint32_t)(ki[t] >> 2), row_me, dc.stream, 0xa77eu, dc.k0, dc.k1);
                const int j = lane & 3;                                            // = key & 3 (the tile's key base is a multiple of 16)
                rr4[0] = quad_word<0>(mine, j); rr4[1] = quad_word<1>(mine, j); rr4[2] = quad_word<2>(mine, j); rr4[3] = quad_word<3>(mine, j);
              }
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                int qi = qbase + qt * 16 + g * 4 + r;         // C layout here: row = query, col (lane & 15) = key
                float p = 0.f, dpe = dp[t][r];
                if (qi < f.Sq && ki[t] < f.Sk) {
                  float sc = (key_masked[t] || (f.causal && ki[t] > qi)) ? NEG_MASK : acc[t][r] * f.scale;
                  p = __expf(sc - l4[r]);
                  float pdrop = p;
                  if (drop) {
                    const uint32_t rr = rr4[r];
                    pdrop = drop_apply(dc, rr, p);
                    dpe = drop_apply(dc, rr, dp[t][r]);
                  }
                  pd[t][qt][r] = pdrop;
                  ds[t][qt][r] = p * (dpe - d4[r]);
                } else {
                  pd[t][qt][r] = 0.f;
                  ds[t][qt][r] = 0.f;
                }
              }
            }
          }
        } else {
#pragma unroll
          for (int t = 0; t < QT; ++t) { pd[t][qt] = f32x4{0.f, 0.f, 0.f, 0.f}; ds[t][qt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
      }
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        if (!TAIL || qbase + k2 * 32 < f.Sq) {
          bf16x8 pf[QT], dsf[QT];
#pragma unroll
          for (int t = 0; t < QT; ++t) { pf[t] = pack2(pd[t][2 * k2], pd[t][2 * k2 + 1]); dsf[t] = pack2(ds[t][2 * k2], ds[t][2 * k2 + 1]); }
#pragma unroll
          for (int d = 0; d < C::DT; ++d) {
            const bf16x8 dfr = frag_tr<DH>(dl, k2 * 32, d * 16, lane);
            const bf16x8 qfr = frag_tr<DH>(ql, k2 * 32, d * 16, lane);
#pragma unroll
            for (int t = 0; t < QT; ++t) {
              dv[t][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dfr, pf[t], dv[t][d], 0, 0, 0);
              dk[t][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, dsf[t], dk[t][d], 0, 0, 0);
            }
          }
        }
      }
    };
    if (wave_live) {
      if (!PLAIN || (qbase + 64 > f.Sq) || (k0 + 16 * QT > f.Sk)) tile_body(std::true_type{}); else tile_body(std::false_type{});
    }
    if (more) {
      tile_lstore<DH>(smem + (cur ^ 1) * 2 * C::TILE, rq);
      tile_lstore<DH>(smem + (cur ^ 1) * 2 * C::TILE + C::TILE, rd);
      if (threadIdx.x < 128) side[(cur ^ 1) * 128 + threadIdx.x] = sreg * stat_mul;
    }
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    if (ki[t] < f.Sk) {
      bf16* dK = reinterpret_cast<bf16*>(a.dk) + b * a.dk_bs + (int64_t)ki[t] * a.dk_ts + (int64_t)h * DH;
      bf16* dV = reinterpret_cast<bf16*>(a.dv) + b * a.dv_bs + (int64_t)ki[t] * a.dv_ts + (int64_t)h * DH;
#pragma unroll
      for (int d = 0; d < C::DT; ++d) {
        bf16x4 tk = {f2bf(dk[t][d][0] * f.scale), f2bf(dk[t][d][1] * f.scale), f2bf(dk[t][d][2] * f.scale), f2bf(dk[t][d][3] * f.scale)};
        bf16x4 tv = {f2bf(dv[t][d][0]), f2bf(dv[t][d][1]), f2bf(dv[t][d][2]), f2bf(dv[t][d][3])};
        *reinterpret_cast<bf16x4*>(dK + d * 16 + g * 4) = tk;
        *reinterpret_cast<bf16x4*>(dV + d * 16 + g * 4) = tv;
      }
    }
  }
}

// =====================================================================================================
// Head-resident attention BACKWARD (round 6): the ViT blocks at 224^2 (vit.py:52-53; S = 260 with experts, 196 without), head dim 64, PLAIN.
// =====================================================================================================
// The streaming kernels above cut a head into 64-query (64-key) blocks: five blocks per head each stream ALL keys (queries) of the head through a
// double-buffered LDS image, one 64-row tile per memory latency, and the dQ kernel sweeps them twice.  A head at these lengths is SMALL: K and V
// (or Q and dO) of all <= 272 tokens are 2 x 36.9 KB as bf16.  So ONE block owns a (batch, head) pair -- 384 blocks at bs32, all resident at
// once at two blocks per CU -- stages both operands ONCE with every request in flight together (one memory latency per block, 25 instead of
// 127 MB through the L2s per launch), and each wave walks 16-row sub-tiles of the head (sub-tile s -> wave s % NW) entirely out of LDS:
//   dQ      : ONE pass instead of two sweeps: P and dP rows of the 16 queries in registers (2 x 68 VGPRs at 17 key tiles), delta from the same
//             fp32 P / dP as before (the consistency argument in attn_bwd_dq_kernel is untouched), dS -> dQ^T += K^T dS^T from the registers:
//             3 matrix products instead of 5, the exponentials once.  44.6 -> 26.9 us per launch at bs32 (rocprofv3, same box).
//   dK / dV : 16 keys per wave and sub-tile, Q / dO / -lse log2(e) / delta of ALL queries resident; rows beyond Sq carry lse = +inf, i.e. P = 0
//             without a bounds select; 8 waves per block (four per SIMD at 118 VGPRs).  36.0 -> 31.8 us.
// LDS image: 128-B rows (no padding) whose 16-B chunk index is XOR-ed with (row & 7): conflict-free for the ds_read_b128 row fragments (the
// 16 lanes of an LDS cycle cover 16 distinct (row parity, chunk) slots) and for the ds_read_b64_tr_b16 transposing reads (8 rows x 2 chunks per
// 32-lane cycle) -- the padded 144-B rows of the streaming kernels are 2-way conflicted on the row fragments and would not leave room for two
// blocks per CU (2 x 41.5 KB).  `split` > 1 cuts the sub-tiles of a head over several blocks when there are too few heads to fill the chip.
// Measured and NOT kept (profiles/r6_ab_attention_resident.txt): the same structure for the FORWARD (all score tiles of 16 queries in
// registers, exact two-pass softmax: 25.5 vs 24.4 us), explicit one-tile-ahead fragment double buffers in all three kernels (+-0), V row
// fragments straight from global memory (17 dependent latencies: 43.5 us), heads cut over 2 / 3 / 5 blocks at bs32 (+2...+4 us).  One sub-tile
// costs a wave ~6 us of issue time whatever the occupancy: these kernels are bound by the per-wave VALU + LDS + MFMA instruction stream.
constexpr int RES_RS = 64;                                    // elements per LDS row (128 B)
// Per-lane element offsets inside an image, computed ONCE per kernel: with row0 a multiple of 16 (32 for the transposing reads) the swizzle term
// (row & 7) depends on the lane only, so a fragment address is `image + row0 * 64 + offset[ks or d]` -- an immediate on the ds_read when row0 is
// a compile-time constant (the fully unrolled key loops), one SALU add otherwise.
struct ResOff { int rows[2], tr[4]; };
__device__ __forceinline__ ResOff res_offsets(int lane) {
  ResOff o;
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) o.rows[ks] = c * RES_RS + (((ks * 4 + g) ^ (c & 7)) << 3);
  const int r = 4 * g + (c >> 2);
#pragma unroll
  for (int d = 0; d < 4; ++d) o.tr[d] = r * RES_RS + ((((2 * d + ((c & 3) >> 1)) ^ (r & 7)) << 3) | ((c & 1) << 2));
  return o;
}
// row-major fragment: lane gets X[row0 + (l&15)][ks*32 + (l>>4)*8 .. +8]
__device__ __forceinline__ bf16x8 rfrag_rows(const bf16* lds, int row0, int ks, const ResOff& o) {
  return *reinterpret_cast<const bf16x8*>(lds + row0 * RES_RS + o.rows[ks]);
}
// transposed fragment, same element order as frag_tr: lane (c, g) gets X[kappa(g, j)][d*16 + c]
__device__ __forceinline__ bf16x8 rfrag_tr(const bf16* lds, int kbase, int d, const ResOff& o) {
  const bf16* p = lds + kbase * RES_RS + o.tr[d];
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 16 * RES_RS));          // (row + 16) & 7 == row & 7
  bf16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return v;
}
// two images of `rows` rows (a multiple of 32) x 64 columns each, all requests issued before the first LDS store; rows beyond the tensor read zeros
template <int NCH, int NTHR>
__device__ __forceinline__ void res_stage2(bf16* la, bf16* lb, rsrc_t ra, rsrc_t rb, int64_t tsa, int64_t tsb, int rows) {
  u32x4 xa[NCH], xb[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int id = threadIdx.x + NTHR * i, r = id >> 3, cc = id & 7;
    xa[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, (uint32_t)((r * tsa + cc * 8) * 2), 0, 0));
    xb[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (uint32_t)((r * tsb + cc * 8) * 2), 0, 0));
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int id = threadIdx.x + NTHR * i, r = id >> 3, cc = id & 7;
    if (r < rows) {
      *reinterpret_cast<u32x4*>(la + r * RES_RS + ((cc ^ (r & 7)) << 3)) = xa[i];
      *reinterpret_cast<u32x4*>(lb + r * RES_RS + ((cc ^ (r & 7)) << 3)) = xb[i];
    }
  }
}
// sub-tiles [s_lo, s_hi) of the block: block x of `split` takes an even share of the nsub 16-row sub-tiles of the head
__device__ __forceinline__ void res_share(int nsub, int split, int x, int& s_lo, int& s_hi) {
  const int q = nsub / split, r = nsub % split;
  s_lo = x * q + min(x, r);
  s_hi = s_lo + q + (x < r ? 1 : 0);
}

template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_dq_res_kernel(ph_attn_bwd_args a, int split) {
  constexpr int DH = 64, KS = 2, DT = 4, ROWS = (NT + 1) / 2 * 32, NCH = ROWS * 8 / 256;
  const ph_attn_fwd_args& f = a.f;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16* kl = reinterpret_cast<bf16*>(smem_raw);
  bf16* vl = kl + ROWS * RES_RS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const ResOff ro = res_offsets(lane);
  PH_TL_DECL;
  PH_TL(0);
  const BlockXY bxy = block_xy(split, f.B * f.H);
  const int b = bxy.y / f.H, h = bxy.y % f.H;
  const bf16* Q = reinterpret_cast<const bf16*>(f.q) + b * f.q_bs + (int64_t)h * DH;
  const bf16* K = reinterpret_cast<const bf16*>(f.k) + b * f.k_bs + (int64_t)h * DH;
  const bf16* V = reinterpret_cast<const bf16*>(f.v) + b * f.v_bs + (int64_t)h * DH;
  const bf16* dO = reinterpret_cast<const bf16*>(a.d_o) + b * a.do_bs + (int64_t)h * DH;
  const float* lse_base = f.lse + (int64_t)(b * f.H + h) * f.Sq;
  res_stage2<NCH, 256>(kl, vl, tile_rsrc(K, f.k_ts, f.Sk, DH), tile_rsrc(V, f.v_ts, f.Sk, DH), f.k_ts, f.v_ts, ROWS);
  int s_lo, s_hi;
  res_share((f.Sq + 15) / 16, split, bxy.x, s_lo, s_hi);
  int sub = s_lo + wave;
  bf16x8 qf[KS], dof[KS], qn[KS], don[KS];
  float lse = 0.f, lsen = 0.f;
  auto qload = [&](int sb, bf16x8 (&dq_)[KS], bf16x8 (&dd_)[KS], float& l_) {
    const int qr = min(sb * 16 + c, f.Sq - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      dq_[ks] = *reinterpret_cast<const bf16x8*>(Q + (int64_t)qr * f.q_ts + ks * 32 + g * 8);
      dd_[ks] = *reinterpret_cast<const bf16x8*>(dO + (int64_t)qr * a.do_ts + ks * 32 + g * 8);
    }
    l_ = lse_base[qr];
  };
  if (sub < s_hi) qload(sub, qf, dof, lse);
  PH_TL(1);
  __syncthreads();
  PH_TL(2);
  int tl_first__ = 1;
  const float c2 = f.scale * LOG2E;
  for (; sub < s_hi; sub += 4) {
    int sk = f.Sk;
    asm volatile("" : "+s"(sk));                      // opaque per iteration: the tail compares are NOT hoisted out of the loop (68 mask pairs -> SGPR spills)
    f32x4 P[NT], DP[NT];
    f32x2 part = {0.f, 0.f};
    const float l2 = lse * LOG2E;
    const f32x2 c2v = {c2, c2}, nl = {-l2, -l2};
    // Software pipeline over the key tiles, three stages deep, written out by hand (round 6, timeline: a tile cost 265 cycles = LDS latency + two dependent
    // MFMA pairs + the exponentials IN SEQUENCE, and two waves per SIMD cannot hide that): step s requests the fragments of tile s, multiplies tile s - 1
    // (fragments requested a step ago) and runs the exponentials of tile s - 2 (accumulators finished a step ago) -- three independent instruction groups.
    bf16x8 kfr[2][KS], vfr[2][KS];
    f32x4 sacc[2], dacc[2];
    static_for(std::make_integer_sequence<int, NT + 2>{}, [&](auto s_c) {
      constexpr int st = decltype(s_c)::value;
      if constexpr (st >= 1 && st <= NT) {
        constexpr int nt = st - 1;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[nt & 1][ks], qf[ks], acc, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[nt & 1][ks], dof[ks], dp, 0, 0, 0);
        }
        sacc[nt & 1] = acc; dacc[nt & 1] = dp;
      }
      if constexpr (st < NT) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { kfr[st & 1][ks] = rfrag_rows(kl, st * 16, ks, ro); vfr[st & 1][ks] = rfrag_rows(vl, st * 16, ks, ro); }
      }
      if constexpr (st >= 2) {
        constexpr int nt = st - 2;
        const f32x4 acc = sacc[nt & 1], dp = dacc[nt & 1];
        const bool tail = nt * 16 + 16 > sk;
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          f32x2 x = {acc[r], acc[r + 1]};
          x = __builtin_elementwise_fma(x, c2v, nl);
          f32x2 p2 = {fast_exp2(x[0]), fast_exp2(x[1])};
          if (tail) {
            asm volatile("");                            // (a real branch: if-converted, the selects ran on all NT tiles -- 207 VALU instructions per sub-tile)
            p2[0] = (nt * 16 + g * 4 + r < sk) ? p2[0] : 0.f;
            p2[1] = (nt * 16 + g * 4 + r + 1 < sk) ? p2[1] : 0.f;
          }
          const f32x2 dd = {dp[r], dp[r + 1]};
          part = __builtin_elementwise_fma(p2, dd, part);
          P[nt][r] = p2[0]; P[nt][r + 1] = p2[1];
        }
        DP[nt] = dp;
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if (tl_first__) PH_TL(3);
    if (sub + 4 < s_hi) qload(sub + 4, qn, don, lsen);          // (behind the P / dP pass: 17 registers fewer live across it)
    const float delta = xor_sum(part[0] + part[1]);
    if (tl_first__) PH_TL(4);
    const int qi = sub * 16 + c;
    if (g == 0 && qi < f.Sq)
      a.delta[(int64_t)(b * f.H + h) * f.Sq + qi] = delta;      // consumed by the dK/dV kernel (launched after this one)
    f32x4 dq[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    // same three-stage pipeline over the 32-key steps: transposed K fragments of step s, dS of step s - 1 packed, dQ MFMAs of step s - 2
    constexpr int NK2 = (NT + 1) / 2;
    bf16x8 ktr[3][DT], dsf[2];
    static_for(std::make_integer_sequence<int, NK2 + 2>{}, [&](auto s_c) {
      constexpr int st = decltype(s_c)::value;
      if constexpr (st >= 2) {
        constexpr int k2 = st - 2;
        if (k2 * 32 < sk) {
#pragma unroll
          for (int d = 0; d < DT; ++d) dq[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktr[k2 % 3][d], dsf[k2 & 1], dq[d], 0, 0, 0);
        }
      }
      if constexpr (st < NK2) {
#pragma unroll
        for (int d = 0; d < DT; ++d) ktr[st % 3][d] = rfrag_tr(kl, st * 32, d, ro);
      }
      if constexpr (st >= 1 && st <= NK2) {
        constexpr int k2 = st - 1;
        f32x4 ds0, ds1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ds0[r] = P[2 * k2][r] * (DP[2 * k2][r] - delta);
          if constexpr (2 * k2 + 1 < NT) ds1[r] = P[2 * k2 + 1][r] * (DP[2 * k2 + 1][r] - delta);
          else ds1[r] = 0.f;
        }
        dsf[k2 & 1] = pack2(ds0, ds1);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if (tl_first__) PH_TL(5);
    if (qi < f.Sq) {
      bf16* dQ = reinterpret_cast<bf16*>(a.dq) + b * a.dq_bs + (int64_t)qi * a.dq_ts + (int64_t)h * DH;
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        bf16x4 tt = {f2bf(dq[d][0] * f.scale), f2bf(dq[d][1] * f.scale), f2bf(dq[d][2] * f.scale), f2bf(dq[d][3] * f.scale)};
        *reinterpret_cast<bf16x4*>(dQ + d * 16 + g * 4) = tt;
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { qf[ks] = qn[ks]; dof[ks] = don[ks]; }
    lse = lsen;
    if (tl_first__) PH_TL(6);
    tl_first__ = 0;
  }
  PH_TL(8);
#ifdef PH_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PH_TL(9);
  PH_TL_FLUSH((int)blockIdx.x, 0, threadIdx.x == 0);
#endif
}

// dK / dV: Q, dO and the per-query statistics of the whole head resident (rows = Sq rounded up to 32); a wave walks 16-key sub-tiles
template <int R32, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW / 2, NW / 2))) void attn_bwd_dkv_res_kernel(ph_attn_bwd_args a, int split) {
  constexpr int DH = 64, KS = 2, DT = 4, ROWS = R32 * 32, NCH = (ROWS * 8 + NW * 64 - 1) / (NW * 64);
  const ph_attn_fwd_args& f = a.f;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16* ql = reinterpret_cast<bf16*>(smem_raw);
  bf16* dl = ql + ROWS * RES_RS;
  float* stl = reinterpret_cast<float*>(dl + ROWS * RES_RS);        // [ROWS] -lse * log2(e) (rows beyond Sq: -inf -> P = 0), then [ROWS] delta
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const ResOff ro = res_offsets(lane);
  const BlockXY bxy = block_xy(split, f.B * f.H);
  const int b = bxy.y / f.H, h = bxy.y % f.H;
  const bf16* Q = reinterpret_cast<const bf16*>(f.q) + b * f.q_bs + (int64_t)h * DH;
  const bf16* K = reinterpret_cast<const bf16*>(f.k) + b * f.k_bs + (int64_t)h * DH;
  const bf16* V = reinterpret_cast<const bf16*>(f.v) + b * f.v_bs + (int64_t)h * DH;
  const bf16* dO = reinterpret_cast<const bf16*>(a.d_o) + b * a.do_bs + (int64_t)h * DH;
  const float* lse_base = f.lse + (int64_t)(b * f.H + h) * f.Sq;
  const float* delta_base = a.delta + (int64_t)(b * f.H + h) * f.Sq;
  for (int r = threadIdx.x; r < ROWS; r += NW * 64) {
    stl[r] = r < f.Sq ? lse_base[r] * -LOG2E : -INFINITY;
    stl[ROWS + r] = r < f.Sq ? delta_base[r] : 0.f;
  }
  res_stage2<NCH, NW * 64>(ql, dl, tile_rsrc(Q, f.q_ts, f.Sq, DH), tile_rsrc(dO, a.do_ts, f.Sq, DH), f.q_ts, a.do_ts, ROWS);
  int s_lo, s_hi;
  res_share((f.Sk + 15) / 16, split, bxy.x, s_lo, s_hi);
  int sub = s_lo + wave;
  bf16x8 kf[KS], vf[KS], kn[KS], vn[KS];
  auto kload = [&](int sb, bf16x8 (&k_)[KS], bf16x8 (&v_)[KS]) {
    const int kr = min(sb * 16 + c, f.Sk - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      k_[ks] = *reinterpret_cast<const bf16x8*>(K + (int64_t)kr * f.k_ts + ks * 32 + g * 8);
      v_[ks] = *reinterpret_cast<const bf16x8*>(V + (int64_t)kr * f.v_ts + ks * 32 + g * 8);
    }
  };
  if (sub < s_hi) kload(sub, kf, vf);
  __syncthreads();
  const float c2 = f.scale * LOG2E;
  const f32x2 c2v = {c2, c2};
  const int nq32 = (f.Sq + 31) / 32;
  for (; sub < s_hi; sub += NW) {
    if (sub + NW < s_hi) kload(sub + NW, kn, vn);
    f32x4 dk[DT], dv[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) { dk[d] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[d] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int k2 = 0; k2 < nq32; ++k2) {
      f32x4 pd[2], ds[2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int q0 = k2 * 32 + hf * 16;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rfrag_rows(ql, q0, ks, ro), kf[ks], acc, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rfrag_rows(dl, q0, ks, ro), vf[ks], dp, 0, 0, 0);
        }
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(stl + q0 + g * 4);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(stl + ROWS + q0 + g * 4);
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const f32x2 nl = {l4[r], l4[r + 1]}, dl2 = {d4[r], d4[r + 1]};
          f32x2 x = {acc[r], acc[r + 1]};
          x = __builtin_elementwise_fma(x, c2v, nl);
          const f32x2 p2 = {fast_exp2(x[0]), fast_exp2(x[1])};
          f32x2 dd = {dp[r], dp[r + 1]};
          dd = p2 * (dd - dl2);
          pd[hf][r] = p2[0]; pd[hf][r + 1] = p2[1];
          ds[hf][r] = dd[0]; ds[hf][r + 1] = dd[1];
        }
      }
      const bf16x8 pf = pack2(pd[0], pd[1]), dsf = pack2(ds[0], ds[1]);
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rfrag_tr(dl, k2 * 32, d, ro), pf, dv[d], 0, 0, 0);
        dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rfrag_tr(ql, k2 * 32, d, ro), dsf, dk[d], 0, 0, 0);
      }
    }
    const int ki = sub * 16 + c;
    if (ki < f.Sk) {
      bf16* dK = reinterpret_cast<bf16*>(a.dk) + b * a.dk_bs + (int64_t)ki * a.dk_ts + (int64_t)h * DH;
      bf16* dV = reinterpret_cast<bf16*>(a.dv) + b * a.dv_bs + (int64_t)ki * a.dv_ts + (int64_t)h * DH;
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        bf16x4 tk = {f2bf(dk[d][0] * f.scale), f2bf(dk[d][1] * f.scale), f2bf(dk[d][2] * f.scale), f2bf(dk[d][3] * f.scale)};
        bf16x4 tv = {f2bf(dv[d][0]), f2bf(dv[d][1]), f2bf(dv[d][2]), f2bf(dv[d][3])};
        *reinterpret_cast<bf16x4*>(dK + d * 16 + g * 4) = tk;
        *reinterpret_cast<bf16x4*>(dV + d * 16 + g * 4) = tv;
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { kf[ks] = kn[ks]; vf[ks] = vn[ks]; }
  }
}

// =====================================================================================================
// Small-query attention (round 5): the DECODER's launches.  Sq = T <= 32 text tokens per (batch, head) against Sk = T keys (causal cut, key
// mask, probability dropout: roberta.py:101-126 self-attention) or against the Sk = 260 image tokens (cross-attention), head dim 64.  With the
// streaming kernels above such a launch is a latency chain: 64-query blocks of which two waves hold nothing, the 64-key tiles of a head
// walked one after the other (cross-attention: 5 tiles forward, 10 tile steps in the dQ kernel, then a second launch for dK / dV that
// re-reads everything), 12-17 us per launch for a few MFLOP (0.016-0.028 of the MFMA peak, 75 launches = 1.08 ms per step).
// Here ONE block owns a (batch, head) pair and EVERY wave owns ALL queries (QT 16-query sub-tiles): the four waves split the KEYS in 32-key
// units (unit u -> wave u % 4), so the key sequence is walked four units at a time, and the partial results are merged through LDS:
//   forward : per wave running (max, sum, O^T) over its units, merged like split-K flash decoding;
//   backward: ONE launch for dQ, dK and dV.  Sweep 0 = delta (fp32, from the same recomputed P / dP as sweep 1, see attn_bwd_dq_kernel),
//             merged over the waves; sweep 1 per unit = the dQ role (S^T = K Q^T orientation, dQ^T += K^T dS^T, merged at the end) and
//             the dK / dV role (mirror orientation S = Q K^T with one key per lane: every wave sees all queries, so dK / dV of its keys
//             are complete when the unit is done -- no second kernel, no delta round trip through HBM).
// K / V row fragments (MFMA A / B operands) come straight from global memory; only what is read TRANSPOSED (ds_read_b64_tr_b16) is staged
// in LDS: V forward; K, Q and dO backward.  Same arithmetic, same dropout words (pure functions of (seed, stream, row, key)), same finfo.min
// mask semantics as the streaming kernels: tests/test_kernels_gpu.py runs both against the torch reference on the same cases.
// =====================================================================================================
constexpr int SM_MAX_UNITS = 10;                     // Sk <= 320
constexpr int SM_MAXU = 3;                           // units per wave: ceil(SM_MAX_UNITS / 4)
constexpr int SM_TS = 36;                            // bf16 row stride of the backward's transposition scratch (32 queries + 4: conflict-free b16 writes)
constexpr int SM_OS = 68;                            // fp32 row stride of the merge images (64 + 4: conflict-free 16-B lane stores)
__host__ __device__ constexpr int sm_max(int x, int y) { return x > y ? x : y; }
template <int QT> __host__ __device__ constexpr int sm_merge_bytes() { return 4 * QT * 16 * SM_OS * 4 + 2 * 4 * QT * 16 * 4; }
__host__ __device__ inline int sm_fwd_bytes(int rows_pad, int merge) { return sm_max(rows_pad * Cfg<64>::RS * 2 + rows_pad * 4, merge); }
template <int QT> __host__ __device__ inline int sm_bwd_bytes(int rows_pad) {          // region A + Q, dO images + 4 scratch pairs + key states + delta + partials
  return sm_max(rows_pad * Cfg<64>::RS * 2, 4 * QT * 16 * SM_OS * 4) + 2 * QT * 16 * Cfg<64>::RS * 2 + 4 * 2 * 32 * 36 * 2 + rows_pad * 4 + (1 + 4) * QT * 16 * 4;
}

// cooperative staging of `rows` rows x 64 channels (strided [token][head*dh] source) into a [rows][RS] LDS image; rows beyond n_valid
// read as zeros (buffer bounds check), so every product with an excluded probability is 0 x 0
template <int MAXLD>
__device__ __forceinline__ void sm_stage(bf16* img, const bf16* src, int64_t ts, int n_valid, int rows) {
  const rsrc_t rs = tile_rsrc(src, ts, n_valid, 64);
  const int nld = (rows * 8 + 255) >> 8;
  u32x4 r[MAXLD];
#pragma unroll
  for (int i = 0; i < MAXLD; ++i)
    if (i < nld) {
      const int id = threadIdx.x + 256 * i;
      r[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)(((id >> 3) * ts + (id & 7) * 8) * 2), 0, 0));
    }
#pragma unroll
  for (int i = 0; i < MAXLD; ++i)
    if (i < nld) {
      const int id = threadIdx.x + 256 * i;
      if (id < rows * 8) *reinterpret_cast<u32x4*>(img + (id >> 3) * Cfg<64>::RS + (id & 7) * 8) = r[i];
    }
}

template <int QT>
__global__ __launch_bounds__(256) void attn_fwd_small_kernel(ph_attn_fwd_args a) {
  constexpr int DH = 64;
  using C = Cfg<DH>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int nunits = (a.Sk + 31) >> 5, rows_pad = nunits * 32;
  bf16* vimg = reinterpret_cast<bf16*>(smem_raw);
  float* kst = reinterpret_cast<float*>(vimg + rows_pad * C::RS);
  float* obuf = reinterpret_cast<float*>(smem_raw);                         // merge images alias the V image (two barriers apart)
  float* mbuf = obuf + 4 * QT * 16 * SM_OS;
  float* lbuf = mbuf + 4 * QT * 16;
  const bf16* Q = reinterpret_cast<const bf16*>(a.q) + b * a.q_bs + (int64_t)h * DH;
  const bf16* K = reinterpret_cast<const bf16*>(a.k) + b * a.k_bs + (int64_t)h * DH;
  const bf16* V = reinterpret_cast<const bf16*>(a.v) + b * a.v_bs + (int64_t)h * DH;
  const uint8_t* km = a.key_mask ? a.key_mask + (int64_t)b * a.Sk : nullptr;

  int qi[QT];
  bf16x8 qf[QT][C::KS];
  f32x4 o[QT][C::DT];
  float m[QT], lsum[QT];
  uint32_t rowid[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    qi[t] = t * 16 + c;
    const int qr = qi[t] < a.Sq ? qi[t] : a.Sq - 1;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) qf[t][ks] = *reinterpret_cast<const bf16x8*>(Q + (int64_t)qr * a.q_ts + ks * 32 + g * 8);
#pragma unroll
    for (int d = 0; d < C::DT; ++d) o[t][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    m[t] = NEG_MASK; lsum[t] = 0.f;
    rowid[t] = (uint32_t)((b * a.H + h) * a.Sq + qr);
  }
  // K row fragments of ALL this wave's units straight from global memory (rows beyond Sk re-read the last key; masked by index below):
  // one round of global latency for the whole kernel
  bf16x8 kfu[SM_MAXU][2][C::KS];
#pragma unroll
  for (int ui = 0; ui < SM_MAXU; ++ui)
    if (wave + 4 * ui < nunits) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int kr = min((wave + 4 * ui) * 32 + nt * 16 + c, a.Sk - 1);
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) kfu[ui][nt][ks] = *reinterpret_cast<const bf16x8*>(K + (int64_t)kr * a.k_ts + ks * 32 + g * 8);
      }
    }
  sm_stage<SM_MAX_UNITS>(vimg, V, a.v_ts, a.Sk, rows_pad);
  for (int i = threadIdx.x; i < rows_pad; i += 256) kst[i] = key_state(key_raw(km, i, a.Sk), i, a.Sk);
  DropCtx dc;
  const bool drop = a.drop_p > 0.f;
  if (drop) dc = make_drop(a.drop_seed, a.drop_stream, a.drop_p);
#pragma unroll
  for (int t = 0; t < QT; ++t)
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) settle(qf[t][ks]);
  __syncthreads();

#pragma unroll
  for (int ui = 0; ui < SM_MAXU; ++ui) {
    if (wave + 4 * ui >= nunits) break;
    const int kbase = (wave + 4 * ui) * 32;
    f32x4 s[QT][2];
    float mx[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) mx[t] = -INFINITY;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      f32x4 acc[QT];
#pragma unroll
      for (int t = 0; t < QT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
        for (int t = 0; t < QT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfu[ui][nt][ks], qf[t][ks], acc[t], 0, 0, 0);
      const f32x4 st = *reinterpret_cast<const f32x4*>(kst + kbase + nt * 16 + g * 4);
#pragma unroll
      for (int t = 0; t < QT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ki = kbase + nt * 16 + g * 4 + r;
          acc[t][r] = score_of(acc[t][r], a.scale, st[r], a.causal && ki > qi[t]);
          mx[t] = fmaxf(mx[t], acc[t][r]);
        }
        s[t][nt] = acc[t];
      }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      const float m_new = fmaxf(m[t], xor_max(mx[t]));
      const float alpha = __expf(m[t] - m_new);
      float rs = 0.f;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __expf(s[t][nt][r] - m_new);
          rs += p;
          s[t][nt][r] = p;
        }
      rs = xor_sum(rs);
      lsum[t] = lsum[t] * alpha + rs;
      m[t] = m_new;
#pragma unroll
      for (int d = 0; d < C::DT; ++d) o[t][d] *= alpha;
      if (drop) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const u32x4 rnd = philox4x32((uint32_t)((kbase + nt * 16 + g * 4) >> 2), rowid[t], dc.stream, 0xa77eu, dc.k0, dc.k1);
#pragma unroll
          for (int r = 0; r < 4; ++r) s[t][nt][r] = drop_apply(dc, rnd[r], s[t][nt][r]);
        }
      }
    }
    bf16x8 pf[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) pf[t] = pack2(s[t][0], s[t][1]);
#pragma unroll
    for (int d = 0; d < C::DT; ++d) {
      const bf16x8 vfr = frag_tr<DH>(vimg, kbase, d * 16, lane);
#pragma unroll
      for (int t = 0; t < QT; ++t) o[t][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr, pf[t], o[t][d], 0, 0, 0);
    }
  }
  // ---- one unit (self-attention at T <= 32): wave 0 holds the whole result
  if (nunits == 1) {
    if (wave == 0) {
#pragma unroll
      for (int t = 0; t < QT; ++t)
        if (qi[t] < a.Sq) {
          const float inv = 1.0f / lsum[t];
          bf16* O = reinterpret_cast<bf16*>(a.o) + b * a.o_bs + (int64_t)qi[t] * a.o_ts + (int64_t)h * DH;
#pragma unroll
          for (int d = 0; d < C::DT; ++d) {
            const bf16x4 tt = {f2bf(o[t][d][0] * inv), f2bf(o[t][d][1] * inv), f2bf(o[t][d][2] * inv), f2bf(o[t][d][3] * inv)};
            *reinterpret_cast<bf16x4*>(O + d * 16 + g * 4) = tt;
          }
          if (g == 0 && a.lse) a.lse[(int64_t)(b * a.H + h) * a.Sq + qi[t]] = m[t] + __logf(lsum[t]);
        }
    }
    return;
  }
  // ---- merge the (max, sum, O^T) partials of the waves that held units
  __syncthreads();                                          // every wave is done with the V image / key states: the merge images may overwrite them
  if (wave < nunits) {
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      const int q = t * 16 + c;
      if (g == 0) { mbuf[wave * QT * 16 + q] = m[t]; lbuf[wave * QT * 16 + q] = lsum[t]; }
#pragma unroll
      for (int d = 0; d < C::DT; ++d) *reinterpret_cast<f32x4*>(obuf + (wave * QT * 16 + q) * SM_OS + d * 16 + g * 4) = o[t][d];
    }
  }
  __syncthreads();
  const int q = threadIdx.x >> 2, col = (threadIdx.x & 3) * 16;
  if (q < a.Sq && q < QT * 16) {
    const int nw = nunits < 4 ? nunits : 4;
    float M = NEG_MASK;
    for (int w = 0; w < nw; ++w) M = fmaxf(M, mbuf[w * QT * 16 + q]);
    float L = 0.f;
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    for (int w = 0; w < nw; ++w) {
      const float wgt = __expf(mbuf[w * QT * 16 + q] - M);
      L += wgt * lbuf[w * QT * 16 + q];
      const float* src = obuf + (w * QT * 16 + q) * SM_OS + col;
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + j4 * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j4 * 4 + r] += wgt * v[r];
      }
    }
    const float inv = 1.0f / L;
    bf16* O = reinterpret_cast<bf16*>(a.o) + b * a.o_bs + (int64_t)q * a.o_ts + (int64_t)h * DH + col;
    bf16x8 o0, o1;
#pragma unroll
    for (int j = 0; j < 8; ++j) { o0[j] = f2bf(acc[j] * inv); o1[j] = f2bf(acc[8 + j] * inv); }
    *reinterpret_cast<bf16x8*>(O) = o0;
    *reinterpret_cast<bf16x8*>(O + 8) = o1;
    if ((threadIdx.x & 3) == 0 && a.lse) a.lse[(int64_t)(b * a.H + h) * a.Sq + q] = M + __logf(L);
  }
}


// Backward of a (batch, head) pair in ONE launch and ONE pass of the generator / exponentials (second version, round 5: the first one
// recomputed scores and masks three times and fetched fragments from global memory inside the unit loop -- 36 us against 42 us of the two
// streaming kernels on the decoder's cross-attention):
//   prologue : K, Q, dO images -> LDS (K and dO / Q are also read transposed), V row fragments of ALL this wave's units -> registers
//   sweep 0  : per unit S^T = K Q^T and dP^T = V dO^T (lane = query column), P = exp(S - lse), dropout words, dP~ = drop(dP); P, dP~ and the
//              16 keep bits stay in registers; delta_i = sum_j P_ij dP~_ij merged over g, the units and -- through LDS -- the waves
//   sweep 1  : dS = P (dP~ - delta);  dQ^T += K^T dS^T (K^T by transposing LDS reads);  P~ and dS go through a per-wave LDS scratch
//              [key][query] and come back as the B operands of dV^T += dO^T P~ and dK^T += Q^T dS (lane = key column): every wave holds all
//              queries, so dK / dV of its keys are final -- stored at once
//   epilogue : dQ merged over the waves through LDS.
template <int QT>
__global__ __launch_bounds__(256) void attn_bwd_small_kernel(ph_attn_bwd_args a) {
  constexpr int DH = 64;
  using C = Cfg<DH>;
  static_assert(QT == 2, "32 queries: one MFMA k-step of the dK/dV products, one b64 pair per scratch read");
  const ph_attn_fwd_args& f = a.f;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / f.H, h = blockIdx.x % f.H;
  const int nunits = (f.Sk + 31) >> 5, rows_pad = nunits * 32;
  constexpr int NQ = QT * 16;
  // region A: K image, overwritten at the end by the dQ merge images; region B: what lives to the end
  bf16* kimg = reinterpret_cast<bf16*>(smem_raw);
  float* dqbuf = reinterpret_cast<float*>(smem_raw);
  char* regb = smem_raw + sm_max(rows_pad * C::RS * 2, 4 * NQ * SM_OS * 4);
  bf16* qimg = reinterpret_cast<bf16*>(regb);
  bf16* doimg = qimg + NQ * C::RS;
  bf16* tsc = doimg + NQ * C::RS + wave * (2 * 32 * SM_TS);  // this wave's scratch: P~ [32 keys][SM_TS], dS [32 keys][SM_TS]
  float* kst = reinterpret_cast<float*>(doimg + NQ * C::RS + 4 * 2 * 32 * SM_TS);
  float* delta_s = kst + rows_pad;
  float* dpart = delta_s + NQ;                               // [4][NQ]
  const bf16* Q = reinterpret_cast<const bf16*>(f.q) + b * f.q_bs + (int64_t)h * DH;
  const bf16* K = reinterpret_cast<const bf16*>(f.k) + b * f.k_bs + (int64_t)h * DH;
  const bf16* V = reinterpret_cast<const bf16*>(f.v) + b * f.v_bs + (int64_t)h * DH;
  const bf16* dO = reinterpret_cast<const bf16*>(a.d_o) + b * a.do_bs + (int64_t)h * DH;
  const uint8_t* km = f.key_mask ? f.key_mask + (int64_t)b * f.Sk : nullptr;
  const float* lse_base = f.lse + (int64_t)(b * f.H + h) * f.Sq;

  int qi[QT];
  uint32_t ridx[QT];
  bf16x8 qf[QT][C::KS], dof[QT][C::KS];
  float lse[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    qi[t] = t * 16 + c;
    const int qr = qi[t] < f.Sq ? qi[t] : f.Sq - 1;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      qf[t][ks] = *reinterpret_cast<const bf16x8*>(Q + (int64_t)qr * f.q_ts + ks * 32 + g * 8);
      dof[t][ks] = *reinterpret_cast<const bf16x8*>(dO + (int64_t)qr * a.do_ts + ks * 32 + g * 8);
    }
    ridx[t] = (uint32_t)((b * f.H + h) * f.Sq + qr);
    lse[t] = lse_base[qr];
  }
  // V row fragments of all units of this wave (A operands of dP^T = V dO^T): one round of global latency for the whole kernel
  bf16x8 vfu[SM_MAXU][2][C::KS];
#pragma unroll
  for (int ui = 0; ui < SM_MAXU; ++ui)
    if (wave + 4 * ui < nunits) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int kr = min((wave + 4 * ui) * 32 + nt * 16 + c, f.Sk - 1);
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) vfu[ui][nt][ks] = *reinterpret_cast<const bf16x8*>(V + (int64_t)kr * f.v_ts + ks * 32 + g * 8);
      }
    }
  sm_stage<SM_MAX_UNITS>(kimg, K, f.k_ts, f.Sk, rows_pad);
  sm_stage<(QT * 16 * 8 + 255) / 256>(qimg, Q, f.q_ts, f.Sq, NQ);
  sm_stage<(QT * 16 * 8 + 255) / 256>(doimg, dO, a.do_ts, f.Sq, NQ);
  for (int i = threadIdx.x; i < rows_pad; i += 256) kst[i] = key_state(key_raw(km, i, f.Sk), i, f.Sk);
  DropCtx dc;
  const bool drop = f.drop_p > 0.f;
  if (drop) dc = make_drop(f.drop_seed, f.drop_stream, f.drop_p);
#pragma unroll
  for (int t = 0; t < QT; ++t) {
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) { settle(qf[t][ks]); settle(dof[t][ks]); }
    settle(lse[t]);
  }
  __syncthreads();

  // ---- sweep 0
  f32x4 pu[SM_MAXU][QT][2], du[SM_MAXU][QT][2];              // P and dropped dP of this wave's units
  uint32_t keepu[SM_MAXU][QT];                               // bit nt*4 + r: element kept by the dropout
  float dsum[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) dsum[t] = 0.f;
#pragma unroll
  for (int ui = 0; ui < SM_MAXU; ++ui) {
    if (wave + 4 * ui < nunits) {
      const int kbase = (wave + 4 * ui) * 32;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        f32x4 acc[QT], dp[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          const bf16x8 kfr = frag_rows<DH>(kimg, kbase + nt * 16, ks, lane);
#pragma unroll
          for (int t = 0; t < QT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[t][ks], acc[t], 0, 0, 0);
            dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfu[ui][nt][ks], dof[t][ks], dp[t], 0, 0, 0);
          }
        }
        const f32x4 st = *reinterpret_cast<const f32x4*>(kst + kbase + nt * 16 + g * 4);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          u32x4 rnd = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
          if (drop) rnd = philox4x32((uint32_t)((kbase + nt * 16 + g * 4) >> 2), ridx[t], dc.stream, 0xa77eu, dc.k0, dc.k1);
          if (nt == 0) keepu[ui][t] = 0u;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ki = kbase + nt * 16 + g * 4 + r;
            const float p = __expf(score_of(acc[t][r], f.scale, st[r], f.causal && ki > qi[t]) - lse[t]);
            const bool keep = !drop || (rnd[r] >> 8) >= dc.thr;
            const float dpe = drop ? (keep ? dp[t][r] * dc.scale : 0.f) : dp[t][r];
            keepu[ui][t] |= keep ? (1u << (nt * 4 + r)) : 0u;
            pu[ui][t][nt][r] = p;
            du[ui][t][nt][r] = dpe;
            dsum[t] += p * dpe;
          }
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const float v = xor_sum(dsum[t]);
    if (g == 0) dpart[wave * NQ + t * 16 + c] = v;           // (0 from a wave without units)
  }
  __syncthreads();
  if (threadIdx.x < NQ) delta_s[threadIdx.x] = dpart[threadIdx.x] + dpart[NQ + threadIdx.x] + dpart[2 * NQ + threadIdx.x] + dpart[3 * NQ + threadIdx.x];
  __syncthreads();
  float delta[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) delta[t] = delta_s[t * 16 + c];

  // ---- sweep 1
  f32x4 dq[QT][C::DT];
#pragma unroll
  for (int t = 0; t < QT; ++t)
#pragma unroll
    for (int d = 0; d < C::DT; ++d) dq[t][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float pscale = drop ? dc.scale : 1.f;
#pragma unroll
  for (int ui = 0; ui < SM_MAXU; ++ui) {
    if (wave + 4 * ui < nunits) {
      const int kbase = (wave + 4 * ui) * 32;
      bf16x8 dsf[QT];
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        f32x4 ds[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = pu[ui][t][nt][r];
            const float dsv = p * (du[ui][t][nt][r] - delta[t]);
            ds[nt][r] = dsv;
            const float pdrop = ((keepu[ui][t] >> (nt * 4 + r)) & 1u) ? p * pscale : 0.f;
            // scratch[key][query]: this lane owns ONE query column (t*16 + c) and the keys nt*16 + g*4 + r
            const int key = nt * 16 + g * 4 + r;
            tsc[key * SM_TS + t * 16 + c] = f2bf(qi[t] < f.Sq ? pdrop : 0.f);
            tsc[32 * SM_TS + key * SM_TS + t * 16 + c] = f2bf(qi[t] < f.Sq ? dsv : 0.f);
          }
        dsf[t] = pack2(ds[0], ds[1]);
      }
#pragma unroll
      for (int d = 0; d < C::DT; ++d) {
        const bf16x8 kfr = frag_tr<DH>(kimg, kbase, d * 16, lane);
#pragma unroll
        for (int t = 0; t < QT; ++t) dq[t][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, dsf[t], dq[t][d], 0, 0, 0);
      }
      // dK / dV of this unit's keys: lane = key column c, reduction over the 32 queries in the order frag_tr delivers dO^T / Q^T
      // (kappa(g, j) = 16 (j >> 2) + 4 g + (j & 3)): B element j of lane (c, g) = scratch[key c][16 (j >> 2) + 4 g + (j & 3)]
      // (the wave's own LDS writes above are ordered before these reads: same wave, in-order LDS pipe)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const int kk = kbase + kt * 16 + c;
        const bf16* row = tsc + (kt * 16 + c) * SM_TS + 4 * g;
        const bf16x4 p0 = *reinterpret_cast<const bf16x4*>(row), p1 = *reinterpret_cast<const bf16x4*>(row + 16);
        const bf16x4 s0 = *reinterpret_cast<const bf16x4*>(row + 32 * SM_TS), s1 = *reinterpret_cast<const bf16x4*>(row + 32 * SM_TS + 16);
        bf16x8 pf, sf;
#pragma unroll
        for (int j = 0; j < 4; ++j) { pf[j] = p0[j]; pf[4 + j] = p1[j]; sf[j] = s0[j]; sf[4 + j] = s1[j]; }
        f32x4 dk[C::DT], dv[C::DT];
#pragma unroll
        for (int d = 0; d < C::DT; ++d) {
          const bf16x8 dfr = frag_tr<DH>(doimg, 0, d * 16, lane);
          const bf16x8 qfr = frag_tr<DH>(qimg, 0, d * 16, lane);
          dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dfr, pf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, sf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
        if (kk < f.Sk) {
          bf16* dK = reinterpret_cast<bf16*>(a.dk) + b * a.dk_bs + (int64_t)kk * a.dk_ts + (int64_t)h * DH;
          bf16* dV = reinterpret_cast<bf16*>(a.dv) + b * a.dv_bs + (int64_t)kk * a.dv_ts + (int64_t)h * DH;
#pragma unroll
          for (int d = 0; d < C::DT; ++d) {
            const bf16x4 tk = {f2bf(dk[d][0] * f.scale), f2bf(dk[d][1] * f.scale), f2bf(dk[d][2] * f.scale), f2bf(dk[d][3] * f.scale)};
            const bf16x4 tv = {f2bf(dv[d][0]), f2bf(dv[d][1]), f2bf(dv[d][2]), f2bf(dv[d][3])};
            *reinterpret_cast<bf16x4*>(dK + d * 16 + g * 4) = tk;
            *reinterpret_cast<bf16x4*>(dV + d * 16 + g * 4) = tv;
          }
        }
      }
    }
  }
  // ---- dQ: one unit -> wave 0 holds all of it; otherwise merged over the waves through LDS
  if (nunits == 1) {
    if (wave == 0) {
#pragma unroll
      for (int t = 0; t < QT; ++t)
        if (qi[t] < f.Sq) {
          bf16* dQ = reinterpret_cast<bf16*>(a.dq) + b * a.dq_bs + (int64_t)qi[t] * a.dq_ts + (int64_t)h * DH;
#pragma unroll
          for (int d = 0; d < C::DT; ++d) {
            const bf16x4 tt = {f2bf(dq[t][d][0] * f.scale), f2bf(dq[t][d][1] * f.scale), f2bf(dq[t][d][2] * f.scale), f2bf(dq[t][d][3] * f.scale)};
            *reinterpret_cast<bf16x4*>(dQ + d * 16 + g * 4) = tt;
          }
        }
    }
    return;
  }
  __syncthreads();                                           // the K image is dead: the dQ images may overwrite it
  if (wave < nunits) {
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
      for (int d = 0; d < C::DT; ++d) *reinterpret_cast<f32x4*>(dqbuf + (wave * NQ + t * 16 + c) * SM_OS + d * 16 + g * 4) = dq[t][d];
  }
  __syncthreads();
  const int q = threadIdx.x >> 2, col = (threadIdx.x & 3) * 16;
  if (q < f.Sq && q < NQ) {
    const int nw = nunits < 4 ? nunits : 4;
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    for (int w = 0; w < nw; ++w) {
      const float* src = dqbuf + (w * NQ + q) * SM_OS + col;
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + j4 * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j4 * 4 + r] += v[r];
      }
    }
    bf16* dQ = reinterpret_cast<bf16*>(a.dq) + b * a.dq_bs + (int64_t)q * a.dq_ts + (int64_t)h * DH + col;
    bf16x8 o0, o1;
#pragma unroll
    for (int j = 0; j < 8; ++j) { o0[j] = f2bf(acc[j] * f.scale); o1[j] = f2bf(acc[8 + j] * f.scale); }
    *reinterpret_cast<bf16x8*>(dQ) = o0;
    *reinterpret_cast<bf16x8*>(dQ + 8) = o1;
  }
}

// eligibility of the small-query kernels: head dim 64, all queries of a (batch, head) pair in QT = 2 sub-tiles (Sq <= 32), K / V image within the LDS budget
int attn_small_qt(const ph_attn_fwd_args* f) {
  if (f->dh != 64 || f->Sk > 32 * SM_MAX_UNITS) return 0;
  return f->Sq <= 32 ? 2 : 0;
}

// no causal cut, no key mask, no probability dropout -> the PLAIN kernels
bool attn_plain_ok(const ph_attn_fwd_args* f) {
  return !f->causal && !f->key_mask && !(f->drop_p > 0.f) && f->scale > 0.f;     // (the row max is taken on raw scores: needs scale > 0)
}

// head-resident kernels (round 6): plain launches with head dim 64 and at most 272 queries and keys.  Returns the number of blocks per head
// (0 = not eligible): one when the heads alone fill the chip at two blocks per CU, else the 16-row sub-tiles of a head are cut over up to
// ceil(sub-tiles / 4) blocks (at that point a block is what a 64-row block of the streaming kernels was, minus the tile-by-tile staging).
int g_attn_small = 3;      // bit 0: small-query kernels (decoder), bit 1: head-resident kernels (ViT at 224^2); ph_attention_tuning(0) forces the streaming kernels (A/B, tests)
int res_split(const ph_attn_fwd_args* f) {
  if (!(g_attn_small & 2) || !attn_plain_ok(f) || f->dh != 64 || f->Sq > 272 || f->Sk > 272 || f->Sq <= 32) return 0;
  const int heads = f->B * f->H, nsub = (std::max(f->Sq, f->Sk) + 15) / 16;
  if (heads >= 256) return 1;
  return std::max(1, std::min((nsub + 3) / 4, (511 + heads) / heads));
}

// 32 queries (keys, in the dK/dV kernel) per wave for long enough sequences
int attn_qt2_ok() { return 1; }          // 1 = by sequence length (0 = never and 2 = wherever eligible were the round-3 A/B arms)

template <typename KernelT>
int set_smem(KernelT k, int bytes) {
  // the attribute is raised only when a launch needs more than any earlier launch of THAT kernel did (the call costs host time on every
  // eager launch otherwise; round-5 advisor finding).  Keyed by the kernel's address: KernelT is the same type for every kernel of one signature.
  if (bytes <= 48 * 1024) return bytes;
  static std::mutex mu;
  static std::unordered_map<const void*, int> largest;
  const void* key = reinterpret_cast<const void*>(k);
  std::lock_guard<std::mutex> lock(mu);
  int& cur = largest[key];
  if (bytes > cur) {
    hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    cur = bytes;
  }
  return bytes;
}


int check_fwd(const ph_attn_fwd_args* f, const char* who) {
  PH_CHECK_ARG(f && f->q && f->k && f->v && f->o, "%s: null pointer", who);
  PH_CHECK_ARG(f->B > 0 && f->H > 0 && f->Sq > 0 && f->Sk > 0, "%s: bad dims", who);
  PH_CHECK_ARG(f->dh == 32 || f->dh == 64 || f->dh == 96 || f->dh == 128 || f->dh == 160, "%s: head dim %d unsupported (32/64/96/128/160)", who, f->dh);
  // 160 = the Experts Resampler of Prismer-HUGE (ViT-H width 1280 / 8 heads, resampler.py:18-24): plain attention only
  PH_CHECK_ARG(f->dh != 160 || attn_plain_ok(f), "%s: head dim 160 is built for plain attention (no causal cut, key mask or dropout)", who);
  PH_CHECK_ARG(((f->q_ts | f->k_ts | f->v_ts | f->o_ts | f->q_bs | f->k_bs | f->v_bs | f->o_bs) % 8) == 0, "%s: strides must be multiples of 8 elements", who);
  PH_CHECK_ARG((((uintptr_t)f->q | (uintptr_t)f->k | (uintptr_t)f->v | (uintptr_t)f->o) & 15) == 0, "%s: pointers must be 16-B aligned", who);
  PH_CHECK_ARG(((int64_t)f->Sq * f->q_ts | (int64_t)f->Sk * f->k_ts | (int64_t)f->Sk * f->v_ts) < (1ll << 30),
               "%s: a (batch, head) slice must span < 2 GiB (tiles are fetched through 32-bit buffer offsets)", who);
  PH_CHECK_ARG(!(f->drop_p > 0.f) || f->drop_seed, "%s: dropout needs a seed", who);
  return PH_OK;
}

}  // namespace

extern "C" int ph_attention_fwd(const ph_attn_fwd_args* a, hipStream_t stream) {
  int rc = check_fwd(a, "ph_attention_fwd");
  if (rc) return rc;
  ProfScope prof__(PH_FAM_ATTN_FWD, 4.0 * a->B * (double)a->H * a->Sq * (double)a->Sk * a->dh, 0.0, stream);
  // the decoder's cross-attention: one block per (batch, head), keys split over the waves.  (One or two key units -- self-attention at T <= 64 --
  // stay on the streaming kernel in the FORWARD: 4.7 vs 5.2 us, profiles/r5_ab_attention_small_query.txt; the fused backward wins at every size.)
  if (const int sq = ((g_attn_small & 1) && a->Sk > 64) ? attn_small_qt(a) : 0) {
    const int rows_pad = ceil_div(a->Sk, 32) * 32;
    (void)sq;
    const int smem = set_smem(attn_fwd_small_kernel<2>, sm_fwd_bytes(rows_pad, sm_merge_bytes<2>()));
    hipLaunchKernelGGL((attn_fwd_small_kernel<2>), dim3(a->B * a->H), dim3(256), smem, stream, *a);
    PH_LAUNCH_CHECK("attn_fwd_small_kernel");
    return PH_OK;
  }
  const bool plain = attn_plain_ok(a);
  // 32 queries per wave (QT = 2): measured slower than 16 in the forward at every shape once the blocks of a head share an XCD
  // (ViT 26.8 vs 32.6 us, LARGE 58 vs 66 us); PH_ATTN_QT2=2 forces it for experiments
  const bool qt2 = plain && attn_qt2_ok() == 2 && a->dh <= 64 && a->Sq >= 128;
  dim3 grid(ceil_div(a->Sq, qt2 ? 128 : 64) * a->B * a->H);                          // 1-D: block_xy() re-maps it XCD-aware
#define PH_FWD_LAUNCH(DHV, PL, QTV)                                                               \
  {                                                                                               \
    int smem = set_smem(attn_fwd_kernel<DHV, PL, QTV>, 4 * Cfg<DHV>::TILE * 2 + 1024);            \
    hipLaunchKernelGGL((attn_fwd_kernel<DHV, PL, QTV>), grid, dim3(256), smem, stream, *a);       \
  }
#define PH_FWD(DHV)                                                                               \
  case DHV: {                                                                                     \
    if constexpr (DHV <= 64) { if (qt2) { PH_FWD_LAUNCH(DHV, true, 2) break; } }                  \
    if (plain) PH_FWD_LAUNCH(DHV, true, 1) else PH_FWD_LAUNCH(DHV, false, 1)                      \
  } break;
  switch (a->dh) { PH_FWD(32) PH_FWD(64) PH_FWD(96) PH_FWD(128) case 160: PH_FWD_LAUNCH(160, true, 1) break; }
#undef PH_FWD_LAUNCH
#undef PH_FWD
  PH_LAUNCH_CHECK("attn_fwd_kernel");
  return PH_OK;
}

extern "C" int ph_attention_bwd(const ph_attn_bwd_args* a, hipStream_t stream) {
  PH_CHECK_ARG(a, "ph_attention_bwd: null args");
  ProfScope prof__(PH_FAM_ATTN_BWD, 10.0 * a->f.B * (double)a->f.H * a->f.Sq * (double)a->f.Sk * a->f.dh, 0.0, stream);
  int rc = check_fwd(&a->f, "ph_attention_bwd");
  if (rc) return rc;
  PH_CHECK_ARG(a->d_o && a->dq && a->dk && a->dv && a->f.lse, "ph_attention_bwd: null pointer");
  PH_CHECK_ARG(((a->do_ts | a->dq_ts | a->dk_ts | a->dv_ts | a->do_bs | a->dq_bs | a->dk_bs | a->dv_bs) % 8) == 0, "ph_attention_bwd: strides must be multiples of 8");
  PH_CHECK_ARG((int64_t)a->f.Sq * a->do_ts < (1ll << 30), "ph_attention_bwd: a (batch, head) slice of dO must span < 2 GiB");
  const ph_attn_fwd_args& f = a->f;
  if (const int sq = (g_attn_small & 1) ? attn_small_qt(&f) : 0) {                // dQ, dK and dV of the decoder's launches in ONE launch
    const int rows_pad = ceil_div(f.Sk, 32) * 32;
    (void)sq;
    const int smem = set_smem(attn_bwd_small_kernel<2>, sm_bwd_bytes<2>(rows_pad));
    hipLaunchKernelGGL((attn_bwd_small_kernel<2>), dim3(f.B * f.H), dim3(256), smem, stream, *a);
    PH_LAUNCH_CHECK("attn_bwd_small_kernel");
    return PH_OK;
  }
  PH_CHECK_ARG(a->delta, "ph_attention_bwd: the streaming kernels need the delta workspace (only the small-query path leaves it untouched)");
  const bool plain = attn_plain_ok(&f);
  if (const int split = res_split(&f)) {                                      // head-resident kernels: the ViT blocks at 224^2
    const dim3 grid(f.B * f.H * split);
#define PH_RES_DQ(NTV) { const int sm = set_smem(attn_bwd_dq_res_kernel<NTV>, 2 * ((NTV + 1) / 2 * 32) * RES_RS * 2);             \
                         hipLaunchKernelGGL((attn_bwd_dq_res_kernel<NTV>), grid, dim3(256), sm, stream, *a, split); }
#define PH_RES_DKV(NCV) { const int sm = set_smem(attn_bwd_dkv_res_kernel<NCV, 8>, 2 * NCV * 32 * RES_RS * 2 + 2 * NCV * 32 * 4);    \
                          hipLaunchKernelGGL((attn_bwd_dkv_res_kernel<NCV, 8>), grid, dim3(512), sm, stream, *a, split); }
    if (f.Sk <= 208) PH_RES_DQ(13) else PH_RES_DQ(17)
    if (f.Sq <= 224) PH_RES_DKV(7) else PH_RES_DKV(9)
#undef PH_RES_DQ
#undef PH_RES_DKV
    PH_LAUNCH_CHECK("attn_bwd_res kernels");
    return PH_OK;
  }
  // backward: 32 queries / keys per wave pay from ~512 tokens on (LARGE, S = 1220: 184 vs 189 us; ViT S = 260: 91 vs 84 us)
  const int qt_mode = attn_qt2_ok();
  const bool q2 = plain && qt_mode && f.dh <= 64 && f.Sq >= (qt_mode == 2 ? 128 : 512);
  const bool k2 = plain && qt_mode && f.dh <= 64 && f.Sk >= (qt_mode == 2 ? 128 : 512);
  dim3 gq(ceil_div(f.Sq, q2 ? 128 : 64) * f.B * f.H), gk(ceil_div(f.Sk, k2 ? 128 : 64) * f.B * f.H);      // 1-D, see block_xy()
#define PH_DQ_LAUNCH(DHV, PL, QTV)                                                                        \
  {                                                                                                       \
    int smem = set_smem(attn_bwd_dq_kernel<DHV, PL, QTV>, 4 * Cfg<DHV>::TILE * 2 + 1024);                 \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<DHV, PL, QTV>), gq, dim3(256), smem, stream, *a);              \
  }
#define PH_DKV_LAUNCH(DHV, PL, QTV)                                                                       \
  {                                                                                                       \
    int smem = set_smem(attn_bwd_dkv_kernel<DHV, PL, QTV>, 4 * Cfg<DHV>::TILE * 2 + 1024);                \
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DHV, PL, QTV>), gk, dim3(256), smem, stream, *a);             \
  }
#define PH_BWD(DHV)                                                                                       \
  case DHV: {                                                                                             \
    bool dq_done = false, dkv_done = false;                                                               \
    if constexpr (DHV <= 64) {                                                                            \
      if (q2) { PH_DQ_LAUNCH(DHV, true, 2) dq_done = true; }                                              \
    }                                                                                                     \
    if (!dq_done) { if (plain) PH_DQ_LAUNCH(DHV, true, 1) else PH_DQ_LAUNCH(DHV, false, 1) }              \
    if constexpr (DHV <= 64) {                                                                            \
      if (k2) { PH_DKV_LAUNCH(DHV, true, 2) dkv_done = true; }                                            \
    }                                                                                                     \
    if (!dkv_done) { if (plain) PH_DKV_LAUNCH(DHV, true, 1) else PH_DKV_LAUNCH(DHV, false, 1) }           \
  } break;
  switch (f.dh) { PH_BWD(32) PH_BWD(64) PH_BWD(96) PH_BWD(128) case 160: { PH_DQ_LAUNCH(160, true, 1) PH_DKV_LAUNCH(160, true, 1) } break; }
#undef PH_DQ_LAUNCH
#undef PH_DKV_LAUNCH
#undef PH_BWD
  PH_LAUNCH_CHECK("attn_bwd kernels");
  return PH_OK;
}

extern "C" int ph_attention_tuning(int small_query_kernels) {
  // argument / return value: 0 = streaming kernels only, 1 = default (every family), 2 = small-query kernels without the one-pass dQ kernel
  const int old = g_attn_small == 3 ? 1 : (g_attn_small == 1 ? 2 : 0);
  if (small_query_kernels >= 0) g_attn_small = small_query_kernels == 0 ? 0 : (small_query_kernels == 2 ? 1 : 3);
  return old;
}

#ifdef PH_TIMELINE
extern "C" int ph_tl_fetch_attn(unsigned long long* host, int n, int reset) {
  hipDeviceSynchronize();
  if (host && n > 0) hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tl), sizeof(unsigned long long) * (size_t)n);
  if (reset) { void* d = nullptr; hipGetSymbolAddress(&d, HIP_SYMBOL(g_tl)); hipMemset(d, 0, sizeof(g_tl)); }
  return 0;
}
#endif
