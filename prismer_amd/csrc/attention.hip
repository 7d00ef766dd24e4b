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
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

template <int DH>
struct Cfg {
  static constexpr int KS = DH / 32;       // MFMA k-steps over the head dimension
  static constexpr int DT = DH / 16;       // 16-wide tiles over the head dimension
  static constexpr int RS = DH + 8;        // LDS row stride in elements (16-B pad)
  static constexpr int TILE = 64 * RS;     // elements per 64-row tile
  static constexpr int CPR = DH / 8;       // 16-B chunks per row
  static constexpr int NLD = 64 * CPR / 256 > 0 ? 64 * CPR / 256 : 1;   // chunks per thread per tile
};

// ---- tile staging: 64 rows x DH of a strided [token][head*dh] tensor -> registers -> LDS.
// Rows beyond n_rows re-read the last valid row (finite data; every consumer masks them by index): the loads carry no
// predicate, so the prefetch is straight-line code and the compiler's vmcnt bookkeeping stays exact around the tile loop.
template <int DH>
__device__ __forceinline__ void tile_gload(const bf16* __restrict__ base, int64_t ts, int row0, int n_rows,
                                           u32x4 (&regs)[Cfg<DH>::NLD]) {
  constexpr int CPR = Cfg<DH>::CPR;
  static_assert(64 * CPR % 256 == 0, "tile chunks must divide over 256 threads");
#pragma unroll
  for (int i = 0; i < Cfg<DH>::NLD; ++i) {
    int id = threadIdx.x + 256 * i;
    int r = id / CPR, c = id % CPR;
    int row = min(row0 + r, n_rows - 1);
    regs[i] = *reinterpret_cast<const u32x4*>(base + (int64_t)row * ts + c * 8);
  }
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
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// =====================================================================================================
// forward
// =====================================================================================================
// LDS image (all three kernels): [stage 0: tile X, tile Y][stage 1: tile X, tile Y][per-row fp32 side data, 2 x 128]
//   forward / dQ : X = K, Y = V, side = key state (64 floats per stage)
//   dK/dV        : X = Q, Y = dO, side = lse (64) + delta (64) per stage
template <int DH, bool PLAIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DH <= 64 ? 4 : 2))) void attn_fwd_kernel(ph_attn_fwd_args a) {
  using C = Cfg<DH>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16* smem = reinterpret_cast<bf16*>(smem_raw);
  float* side = reinterpret_cast<float*>(smem + 4 * C::TILE);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int b = blockIdx.y / a.H, h = blockIdx.y % a.H;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const int qi = q0 + c;
  const bool wave_live = q0 < a.Sq;                 // a wave whose 16 queries are all padding only helps staging
  const bf16* Q = reinterpret_cast<const bf16*>(a.q) + b * a.q_bs + (int64_t)h * DH;
  const bf16* K = reinterpret_cast<const bf16*>(a.k) + b * a.k_bs + (int64_t)h * DH;
  const bf16* V = reinterpret_cast<const bf16*>(a.v) + b * a.v_bs + (int64_t)h * DH;
  const uint8_t* km = a.key_mask ? a.key_mask + (int64_t)b * a.Sk : nullptr;

  bf16x8 qf[C::KS];
  {
    int qr = qi < a.Sq ? qi : a.Sq - 1;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks)
      qf[ks] = *reinterpret_cast<const bf16x8*>(Q + (int64_t)qr * a.q_ts + ks * 32 + g * 8);
  }
  f32x4 o[C::DT];
#pragma unroll
  for (int d = 0; d < C::DT; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = NEG_MASK, lsum = 0.f;
  DropCtx dc;
  const bool drop = a.drop_p > 0.f;
  if (drop) dc = make_drop(a.drop_seed, a.drop_stream, a.drop_p);
  const uint32_t rowid = (uint32_t)((b * a.H + h) * a.Sq + (qi < a.Sq ? qi : a.Sq - 1));

  const int ntiles = (a.Sk + 63) / 64;
  u32x4 rk[C::NLD], rv[C::NLD];
  uint32_t kraw = 1u;
  int kraw_i = threadIdx.x;
  tile_gload<DH>(K, a.k_ts, 0, a.Sk, rk);
  tile_gload<DH>(V, a.v_ts, 0, a.Sk, rv);
  if (threadIdx.x < 64) kraw = key_raw(km, kraw_i, a.Sk);
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) settle(qf[ks]);
  tile_lstore<DH>(smem, rk);
  tile_lstore<DH>(smem + C::TILE, rv);
  if (threadIdx.x < 64) side[threadIdx.x] = key_state(kraw, kraw_i, a.Sk);
  __syncthreads();
  int cur = 0;
  for (int t = 0; t < ntiles; ++t) {
    const bool more = t + 1 < ntiles;
    if (more) {
      tile_gload<DH>(K, a.k_ts, (t + 1) * 64, a.Sk, rk);
      tile_gload<DH>(V, a.v_ts, (t + 1) * 64, a.Sk, rv);
      kraw_i = (t + 1) * 64 + threadIdx.x;
      if (threadIdx.x < 64) kraw = key_raw(km, kraw_i, a.Sk);
    }
    const bf16* kl = smem + cur * 2 * C::TILE;
    const bf16* vl = kl + C::TILE;
    const float* kst = side + cur * 128;
    const int kbase = t * 64;
    if (wave_live) {
      f32x4 s[4];
      float mx = -INFINITY;
      const bool tail = kbase + 64 > a.Sk;            // (uniform) this tile crosses the end of the key sequence
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if (kbase + nt * 16 < a.Sk) {                 // 16-key sub-tiles entirely beyond Sk cost nothing
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < C::KS; ++ks)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<DH>(kl, nt * 16, ks, lane), qf[ks], acc, 0, 0, 0);
          if constexpr (PLAIN) {                      // base-2 scores: s * scale * log2(e)
            const float c2 = a.scale * LOG2E;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              acc[r] *= c2;
              if (tail) acc[r] = (kbase + nt * 16 + g * 4 + r < a.Sk) ? acc[r] : -INFINITY;
              mx = fmaxf(mx, acc[r]);
            }
          } else {
            const f32x4 st = *reinterpret_cast<const f32x4*>(kst + nt * 16 + g * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              int ki = kbase + nt * 16 + g * 4 + r;
              acc[r] = score_of(acc[r], a.scale, st[r], a.causal && ki > qi);
              mx = fmaxf(mx, acc[r]);
            }
          }
          s[nt] = acc;
        } else {
          s[nt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
      }
      mx = xor_max(mx);
      float m_new = fmaxf(m, mx);
      float alpha = PLAIN ? fast_exp2(m - m_new) : __expf(m - m_new);
      float rs = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = PLAIN ? fast_exp2(s[nt][r] - m_new) : __expf(s[nt][r] - m_new);
          rs += p;
          s[nt][r] = p;
        }
      rs = xor_sum(rs);
      lsum = lsum * alpha + rs;
      m = m_new;
#pragma unroll
      for (int d = 0; d < C::DT; ++d) o[d] *= alpha;
      if (drop) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          int k4 = (kbase + nt * 16 + g * 4) >> 2;
          u32x4 rnd = philox4x32((uint32_t)k4, rowid, dc.stream, 0xa77eu, dc.k0, dc.k1);
#pragma unroll
          for (int r = 0; r < 4; ++r) s[nt][r] = drop_apply(dc, rnd[r], s[nt][r]);
        }
      }
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        if (kbase + k2 * 32 < a.Sk) {
          bf16x8 pf = pack2(s[2 * k2], s[2 * k2 + 1]);
#pragma unroll
          for (int d = 0; d < C::DT; ++d)
            o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DH>(vl, k2 * 32, d * 16, lane), pf, o[d], 0, 0, 0);
        }
      }
    }
    if (more) {
      tile_lstore<DH>(smem + (cur ^ 1) * 2 * C::TILE, rk);
      tile_lstore<DH>(smem + (cur ^ 1) * 2 * C::TILE + C::TILE, rv);
      if (threadIdx.x < 64) side[(cur ^ 1) * 128 + threadIdx.x] = key_state(kraw, kraw_i, a.Sk);
    }
    __syncthreads();
    cur ^= 1;
  }
  if (qi < a.Sq) {
    float inv = 1.0f / lsum;
    bf16* O = reinterpret_cast<bf16*>(a.o) + b * a.o_bs + (int64_t)qi * a.o_ts + (int64_t)h * DH;
#pragma unroll
    for (int d = 0; d < C::DT; ++d) {
      bf16x4 t = {f2bf(o[d][0] * inv), f2bf(o[d][1] * inv), f2bf(o[d][2] * inv), f2bf(o[d][3] * inv)};
      *reinterpret_cast<bf16x4*>(O + d * 16 + g * 4) = t;
    }
    if (g == 0 && a.lse) a.lse[(int64_t)(b * a.H + h) * a.Sq + qi] = (PLAIN ? m * LN2 : m) + __logf(lsum);    // natural-log lse either way
  }
}

// =====================================================================================================
// backward: dQ  (same streaming structure as forward)
// =====================================================================================================
template <int DH, bool PLAIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DH <= 64 ? 4 : 2))) void attn_bwd_dq_kernel(ph_attn_bwd_args a) {
  using C = Cfg<DH>;
  const ph_attn_fwd_args& f = a.f;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16* smem = reinterpret_cast<bf16*>(smem_raw);
  float* side = reinterpret_cast<float*>(smem + 4 * C::TILE);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int b = blockIdx.y / f.H, h = blockIdx.y % f.H;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const int qi = q0 + c;
  const int qr = qi < f.Sq ? qi : f.Sq - 1;
  const bool wave_live = q0 < f.Sq;
  const bf16* Q = reinterpret_cast<const bf16*>(f.q) + b * f.q_bs + (int64_t)h * DH;
  const bf16* K = reinterpret_cast<const bf16*>(f.k) + b * f.k_bs + (int64_t)h * DH;
  const bf16* V = reinterpret_cast<const bf16*>(f.v) + b * f.v_bs + (int64_t)h * DH;
  const bf16* dO = reinterpret_cast<const bf16*>(a.d_o) + b * a.do_bs + (int64_t)h * DH;
  const uint8_t* km = f.key_mask ? f.key_mask + (int64_t)b * f.Sk : nullptr;

  bf16x8 qf[C::KS], dof[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) {
    qf[ks] = *reinterpret_cast<const bf16x8*>(Q + (int64_t)qr * f.q_ts + ks * 32 + g * 8);
    dof[ks] = *reinterpret_cast<const bf16x8*>(dO + (int64_t)qr * a.do_ts + ks * 32 + g * 8);
  }
  const int64_t ridx = (int64_t)(b * f.H + h) * f.Sq + qr;
  float lse = f.lse[ridx];
  DropCtx dc;
  const bool drop = f.drop_p > 0.f;
  if (drop) dc = make_drop(f.drop_seed, f.drop_stream, f.drop_p);
  const uint32_t rowid = (uint32_t)ridx;

  f32x4 dq[C::DT];
#pragma unroll
  for (int d = 0; d < C::DT; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};

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
  tile_gload<DH>(K, f.k_ts, 0, f.Sk, rk);
  tile_gload<DH>(V, f.v_ts, 0, f.Sk, rv);
  if (threadIdx.x < 64) kraw = key_raw(km, kraw_i, f.Sk);
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) { settle(qf[ks]); settle(dof[ks]); }
  settle(lse);
  tile_lstore<DH>(smem, rk);
  tile_lstore<DH>(smem + C::TILE, rv);
  if (threadIdx.x < 64) side[threadIdx.x] = key_state(kraw, kraw_i, f.Sk);
  __syncthreads();
  int cur = 0;
  float delta = 0.f, dsum = 0.f;
  for (int u = 0; u < nsteps; ++u) {
    const bool sweep1 = u >= ntiles;
    const int t = sweep1 ? u - ntiles : u;
    const bool more = u + 1 < nsteps;
    if (more) {
      const int tn = (t + 1 == ntiles) ? 0 : t + 1;
      tile_gload<DH>(K, f.k_ts, tn * 64, f.Sk, rk);
      tile_gload<DH>(V, f.v_ts, tn * 64, f.Sk, rv);
      kraw_i = tn * 64 + threadIdx.x;
      if (threadIdx.x < 64) kraw = key_raw(km, kraw_i, f.Sk);
    }
    const bf16* kl = smem + cur * 2 * C::TILE;
    const bf16* vl = kl + C::TILE;
    const float* kst = side + cur * 128;
    const int kbase = t * 64;
    if (wave_live) {
      f32x4 ds[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if (kbase + nt * 16 < f.Sk) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < C::KS; ++ks) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<DH>(kl, nt * 16, ks, lane), qf[ks], acc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<DH>(vl, nt * 16, ks, lane), dof[ks], dp, 0, 0, 0);
          }
          if constexpr (PLAIN) {
            const float c2 = f.scale * LOG2E, l2 = lse * LOG2E;
            const bool tail = kbase + 64 > f.Sk;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float p = fast_exp2(fmaf(acc[r], c2, -l2));
              if (tail) p = (kbase + nt * 16 + g * 4 + r < f.Sk) ? p : 0.f;
              dsum += p * dp[r];
              ds[nt][r] = p * (dp[r] - delta);
            }
          } else {
            const f32x4 st = *reinterpret_cast<const f32x4*>(kst + nt * 16 + g * 4);
            u32x4 rnd;
            if (drop) rnd = philox4x32((uint32_t)((kbase + nt * 16 + g * 4) >> 2), rowid, dc.stream, 0xa77eu, dc.k0, dc.k1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              int ki = kbase + nt * 16 + g * 4 + r;
              float p = __expf(score_of(acc[r], f.scale, st[r], f.causal && ki > qi) - lse);
              float dpe = drop ? drop_apply(dc, rnd[r], dp[r]) : dp[r];
              dsum += p * dpe;
              ds[nt][r] = p * (dpe - delta);
            }
          }
        } else {
          ds[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      if (sweep1) {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          if (kbase + k2 * 32 < f.Sk) {
            bf16x8 dsf = pack2(ds[2 * k2], ds[2 * k2 + 1]);
#pragma unroll
            for (int d = 0; d < C::DT; ++d)
              dq[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DH>(kl, k2 * 32, d * 16, lane), dsf, dq[d], 0, 0, 0);
          }
        }
      }
      if (u == ntiles - 1) {
        delta = xor_sum(dsum);
        if (g == 0 && qi < f.Sq) a.delta[ridx] = delta;    // consumed by the dK/dV kernel (launched after this one)
      }
    }
    if (more) {
      tile_lstore<DH>(smem + (cur ^ 1) * 2 * C::TILE, rk);
      tile_lstore<DH>(smem + (cur ^ 1) * 2 * C::TILE + C::TILE, rv);
      if (threadIdx.x < 64) side[(cur ^ 1) * 128 + threadIdx.x] = key_state(kraw, kraw_i, f.Sk);
    }
    __syncthreads();
    cur ^= 1;
  }
  if (qi < f.Sq) {
    bf16* dQ = reinterpret_cast<bf16*>(a.dq) + b * a.dq_bs + (int64_t)qi * a.dq_ts + (int64_t)h * DH;
#pragma unroll
    for (int d = 0; d < C::DT; ++d) {
      bf16x4 t = {f2bf(dq[d][0] * f.scale), f2bf(dq[d][1] * f.scale), f2bf(dq[d][2] * f.scale), f2bf(dq[d][3] * f.scale)};
      *reinterpret_cast<bf16x4*>(dQ + d * 16 + g * 4) = t;
    }
  }
}

// =====================================================================================================
// backward: dK, dV  (one key per lane; Q / dO streamed in 64-query tiles)
// =====================================================================================================
template <int DH, bool PLAIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DH <= 64 ? 3 : 2))) void attn_bwd_dkv_kernel(ph_attn_bwd_args a) {
  using C = Cfg<DH>;
  const ph_attn_fwd_args& f = a.f;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16* smem = reinterpret_cast<bf16*>(smem_raw);
  float* side = reinterpret_cast<float*>(smem + 4 * C::TILE);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int b = blockIdx.y / f.H, h = blockIdx.y % f.H;
  const int k0 = blockIdx.x * 64 + wave * 16;
  const int ki = k0 + c;
  const int kr = ki < f.Sk ? ki : f.Sk - 1;
  const bool wave_live = k0 < f.Sk;
  const bf16* Q = reinterpret_cast<const bf16*>(f.q) + b * f.q_bs + (int64_t)h * DH;
  const bf16* K = reinterpret_cast<const bf16*>(f.k) + b * f.k_bs + (int64_t)h * DH;
  const bf16* V = reinterpret_cast<const bf16*>(f.v) + b * f.v_bs + (int64_t)h * DH;
  const bf16* dO = reinterpret_cast<const bf16*>(a.d_o) + b * a.do_bs + (int64_t)h * DH;
  const uint8_t* km = f.key_mask ? f.key_mask + (int64_t)b * f.Sk : nullptr;

  bf16x8 kf[C::KS], vf[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) {
    kf[ks] = *reinterpret_cast<const bf16x8*>(K + (int64_t)kr * f.k_ts + ks * 32 + g * 8);
    vf[ks] = *reinterpret_cast<const bf16x8*>(V + (int64_t)kr * f.v_ts + ks * 32 + g * 8);
  }
  DropCtx dc;
  const bool drop = f.drop_p > 0.f;
  if (drop) dc = make_drop(f.drop_seed, f.drop_stream, f.drop_p);
  const float* lse_base = f.lse + (int64_t)(b * f.H + h) * f.Sq;
  const float* delta_base = a.delta + (int64_t)(b * f.H + h) * f.Sq;
  // per-query softmax statistics of a tile travel with it: threads 0..63 fetch lse, 64..127 delta
  const float* stat_base = threadIdx.x < 64 ? lse_base : delta_base;
  const int stat_row = threadIdx.x & 63;

  f32x4 dk[C::DT], dv[C::DT];
#pragma unroll
  for (int d = 0; d < C::DT; ++d) { dk[d] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[d] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int ntiles = (f.Sq + 63) / 64;
  u32x4 rq[C::NLD], rd[C::NLD];
  float sreg = 0.f;
  tile_gload<DH>(Q, f.q_ts, 0, f.Sq, rq);
  tile_gload<DH>(dO, a.do_ts, 0, f.Sq, rd);
  if (threadIdx.x < 128) sreg = stat_base[min(stat_row, f.Sq - 1)];
  float kstate = key_raw(km, kr, f.Sk) ? 0.f : 1.f;   // this lane's key: excluded by key_mask?
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) { settle(kf[ks]); settle(vf[ks]); }
  settle(kstate);
  const bool key_masked = kstate != 0.f;
  tile_lstore<DH>(smem, rq);
  tile_lstore<DH>(smem + C::TILE, rd);
  if (threadIdx.x < 128) side[threadIdx.x] = sreg;
  __syncthreads();
  int cur = 0;
  for (int t = 0; t < ntiles; ++t) {
    const bool more = t + 1 < ntiles;
    if (more) {
      tile_gload<DH>(Q, f.q_ts, (t + 1) * 64, f.Sq, rq);
      tile_gload<DH>(dO, a.do_ts, (t + 1) * 64, f.Sq, rd);
      if (threadIdx.x < 128) sreg = stat_base[min((t + 1) * 64 + stat_row, f.Sq - 1)];
    }
    const bf16* ql = smem + cur * 2 * C::TILE;
    const bf16* dl = ql + C::TILE;
    const float* stl = side + cur * 128;
    const int qbase = t * 64;
    if (wave_live) {
      f32x4 pd[4], ds[4];
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        if (qbase + qt * 16 < f.Sq) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < C::KS; ++ks) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<DH>(ql, qt * 16, ks, lane), kf[ks], acc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<DH>(dl, qt * 16, ks, lane), vf[ks], dp, 0, 0, 0);
          }
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(stl + qt * 16 + g * 4);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(stl + 64 + qt * 16 + g * 4);
          if constexpr (PLAIN) {
            const float c2 = f.scale * LOG2E;
            const bool tail = (qbase + 64 > f.Sq) || (k0 + 16 > f.Sk);     // (wave-uniform) this tile pair touches padding rows / keys
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float p = fast_exp2(fmaf(acc[r], c2, -l4[r] * LOG2E));
              if (tail) p = (qbase + qt * 16 + g * 4 + r < f.Sq && ki < f.Sk) ? p : 0.f;
              pd[qt][r] = p;
              ds[qt][r] = p * (dp[r] - d4[r]);
            }
          } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int qi = qbase + qt * 16 + g * 4 + r;         // C layout here: row = query, col (lane & 15) = key
            float p = 0.f, dpe = dp[r];
            if (qi < f.Sq && ki < f.Sk) {
              float sc = (key_masked || (f.causal && ki > qi)) ? NEG_MASK : acc[r] * f.scale;
              p = __expf(sc - l4[r]);
              float pdrop = p;
              if (drop) {
                uint32_t rowid = (uint32_t)((b * f.H + h) * f.Sq + qi);
                u32x4 rnd = philox4x32((uint32_t)(ki >> 2), rowid, dc.stream, 0xa77eu, dc.k0, dc.k1);
                uint32_t rr = rnd[ki & 3];
                pdrop = drop_apply(dc, rr, p);
                dpe = drop_apply(dc, rr, dp[r]);
              }
              pd[qt][r] = pdrop;
              ds[qt][r] = p * (dpe - d4[r]);
            } else {
              pd[qt][r] = 0.f;
              ds[qt][r] = 0.f;
            }
          }
          }
        } else {
          pd[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
          ds[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        if (qbase + k2 * 32 < f.Sq) {
          bf16x8 pf = pack2(pd[2 * k2], pd[2 * k2 + 1]);
          bf16x8 dsf = pack2(ds[2 * k2], ds[2 * k2 + 1]);
#pragma unroll
          for (int d = 0; d < C::DT; ++d) {
            dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DH>(dl, k2 * 32, d * 16, lane), pf, dv[d], 0, 0, 0);
            dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DH>(ql, k2 * 32, d * 16, lane), dsf, dk[d], 0, 0, 0);
          }
        }
      }
    }
    if (more) {
      tile_lstore<DH>(smem + (cur ^ 1) * 2 * C::TILE, rq);
      tile_lstore<DH>(smem + (cur ^ 1) * 2 * C::TILE + C::TILE, rd);
      if (threadIdx.x < 128) side[(cur ^ 1) * 128 + threadIdx.x] = sreg;
    }
    __syncthreads();
    cur ^= 1;
  }
  if (ki < f.Sk) {
    bf16* dK = reinterpret_cast<bf16*>(a.dk) + b * a.dk_bs + (int64_t)ki * a.dk_ts + (int64_t)h * DH;
    bf16* dV = reinterpret_cast<bf16*>(a.dv) + b * a.dv_bs + (int64_t)ki * a.dv_ts + (int64_t)h * DH;
#pragma unroll
    for (int d = 0; d < C::DT; ++d) {
      bf16x4 tk = {f2bf(dk[d][0] * f.scale), f2bf(dk[d][1] * f.scale), f2bf(dk[d][2] * f.scale), f2bf(dk[d][3] * f.scale)};
      bf16x4 tv = {f2bf(dv[d][0]), f2bf(dv[d][1]), f2bf(dv[d][2]), f2bf(dv[d][3])};
      *reinterpret_cast<bf16x4*>(dK + d * 16 + g * 4) = tk;
      *reinterpret_cast<bf16x4*>(dV + d * 16 + g * 4) = tv;
    }
  }
}

// no causal cut, no key mask, no probability dropout -> the PLAIN kernels (PH_ATTN_PLAIN=0: A/B against the generic ones)
bool attn_plain_ok(const ph_attn_fwd_args* f) {
  static const bool on = [] { const char* e = getenv("PH_ATTN_PLAIN"); return !e || atoi(e) != 0; }();
  return on && !f->causal && !f->key_mask && !(f->drop_p > 0.f);
}

template <typename KernelT>
int set_smem(KernelT k, int bytes) {
  if (bytes > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  return bytes;
}

int check_fwd(const ph_attn_fwd_args* f, const char* who) {
  PH_CHECK_ARG(f && f->q && f->k && f->v && f->o, "%s: null pointer", who);
  PH_CHECK_ARG(f->B > 0 && f->H > 0 && f->Sq > 0 && f->Sk > 0, "%s: bad dims", who);
  PH_CHECK_ARG(f->dh == 32 || f->dh == 64 || f->dh == 96 || f->dh == 128, "%s: head dim %d unsupported (32/64/96/128)", who, f->dh);
  PH_CHECK_ARG(((f->q_ts | f->k_ts | f->v_ts | f->o_ts | f->q_bs | f->k_bs | f->v_bs | f->o_bs) % 8) == 0, "%s: strides must be multiples of 8 elements", who);
  PH_CHECK_ARG((((uintptr_t)f->q | (uintptr_t)f->k | (uintptr_t)f->v | (uintptr_t)f->o) & 15) == 0, "%s: pointers must be 16-B aligned", who);
  PH_CHECK_ARG(!(f->drop_p > 0.f) || f->drop_seed, "%s: dropout needs a seed", who);
  return PH_OK;
}

}  // namespace

extern "C" int ph_attention_fwd(const ph_attn_fwd_args* a, hipStream_t stream) {
  int rc = check_fwd(a, "ph_attention_fwd");
  if (rc) return rc;
  ProfScope prof__(PH_FAM_ATTN_FWD, 4.0 * a->B * (double)a->H * a->Sq * (double)a->Sk * a->dh, 0.0, stream);
  dim3 grid(ceil_div(a->Sq, 64), a->B * a->H);
  const bool plain = attn_plain_ok(a);
#define PH_FWD(DHV)                                                                               \
  case DHV: {                                                                                     \
    if (plain) {                                                                                  \
      int smem = set_smem(attn_fwd_kernel<DHV, true>, 4 * Cfg<DHV>::TILE * 2 + 1024);             \
      hipLaunchKernelGGL((attn_fwd_kernel<DHV, true>), grid, dim3(256), smem, stream, *a);        \
    } else {                                                                                      \
      int smem = set_smem(attn_fwd_kernel<DHV, false>, 4 * Cfg<DHV>::TILE * 2 + 1024);            \
      hipLaunchKernelGGL((attn_fwd_kernel<DHV, false>), grid, dim3(256), smem, stream, *a);       \
    }                                                                                             \
  } break;
  switch (a->dh) { PH_FWD(32) PH_FWD(64) PH_FWD(96) PH_FWD(128) }
#undef PH_FWD
  PH_LAUNCH_CHECK("attn_fwd_kernel");
  return PH_OK;
}

extern "C" int ph_attention_bwd(const ph_attn_bwd_args* a, hipStream_t stream) {
  PH_CHECK_ARG(a, "ph_attention_bwd: null args");
  ProfScope prof__(PH_FAM_ATTN_BWD, 10.0 * a->f.B * (double)a->f.H * a->f.Sq * (double)a->f.Sk * a->f.dh, 0.0, stream);
  int rc = check_fwd(&a->f, "ph_attention_bwd");
  if (rc) return rc;
  PH_CHECK_ARG(a->d_o && a->dq && a->dk && a->dv && a->delta && a->f.lse, "ph_attention_bwd: null pointer");
  PH_CHECK_ARG(((a->do_ts | a->dq_ts | a->dk_ts | a->dv_ts | a->do_bs | a->dq_bs | a->dk_bs | a->dv_bs) % 8) == 0, "ph_attention_bwd: strides must be multiples of 8");
  const ph_attn_fwd_args& f = a->f;
  dim3 gq(ceil_div(f.Sq, 64), f.B * f.H), gk(ceil_div(f.Sk, 64), f.B * f.H);
  const bool plain = attn_plain_ok(&f);
#define PH_BWD(DHV)                                                                                       \
  case DHV: {                                                                                             \
    if (plain) {                                                                                          \
      int smem = set_smem(attn_bwd_dq_kernel<DHV, true>, 4 * Cfg<DHV>::TILE * 2 + 1024);                  \
      set_smem(attn_bwd_dkv_kernel<DHV, true>, smem);                                                     \
      hipLaunchKernelGGL((attn_bwd_dq_kernel<DHV, true>), gq, dim3(256), smem, stream, *a);               \
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<DHV, true>), gk, dim3(256), smem, stream, *a);              \
    } else {                                                                                              \
      int smem = set_smem(attn_bwd_dq_kernel<DHV, false>, 4 * Cfg<DHV>::TILE * 2 + 1024);                 \
      set_smem(attn_bwd_dkv_kernel<DHV, false>, smem);                                                    \
      hipLaunchKernelGGL((attn_bwd_dq_kernel<DHV, false>), gq, dim3(256), smem, stream, *a);              \
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<DHV, false>), gk, dim3(256), smem, stream, *a);             \
    }                                                                                                     \
  } break;
  switch (f.dh) { PH_BWD(32) PH_BWD(64) PH_BWD(96) PH_BWD(128) }
#undef PH_BWD
  PH_LAUNCH_CHECK("attn_bwd kernels");
  return PH_OK;
}
