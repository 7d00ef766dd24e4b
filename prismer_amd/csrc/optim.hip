// Optimizer + small HBM-bound utilities for gfx950.
// AdamW replaces torch.optim.AdamW (reference: train_caption.py:111-112,133); the rest are the glue kernels of
// the layer programs (bias gradients, casts, strided row copies, conv-weight layout changes).
#include <stdlib.h>

#include "common.h"

namespace {

// torch.optim.AdamW:  p *= 1 - lr*wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
//                     p -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// HBM-bound: 16 B read + 14 B written per parameter.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16* __restrict__ pb, int64_t n, const float* __restrict__ hyper,
                                                    float b1, float b2, float eps, float wd, float gscale, int zero_g, int mode,
                                                    const uint32_t* __restrict__ keep) {
  const float lr = hyper[0], bc1 = hyper[1], bc2 = hyper[2];
  const float step = lr / bc1, rbc2 = rsqrtf(bc2), decay = 1.f - lr * wd;
  int64_t n4 = n >> 2;
#ifndef PH_ADAMW_NT
#define PH_ADAMW_NT 0
#endif
#if PH_ADAMW_NT
#define LD4(ptr, i) __builtin_nontemporal_load(reinterpret_cast<f32x4*>(ptr) + (i))
#define ST4(ptr, i, val) __builtin_nontemporal_store((val), reinterpret_cast<f32x4*>(ptr) + (i))
#else
#define LD4(ptr, i) (reinterpret_cast<f32x4*>(ptr)[i])
#define ST4(ptr, i, val) (reinterpret_cast<f32x4*>(ptr)[i] = (val))
#endif
  // keep (optional): bit c set = the 1024 gradients of chunk c are NOT zeroed -- ranges whose producer overwrites them in the next
  // step (single-writer weight gradients, Trainer): saves their 4 B/parameter of zero stores here and the C read of the producing GEMM
  auto zero_vec = [&](int64_t i) -> bool {            // i: float4 index
    if (!zero_g) return false;
    if (!keep) return true;
    const int64_t c = i >> 8;
    return !((keep[c >> 5] >> (c & 31)) & 1u);
  };
  auto update = [&](f32x4& tp, const f32x4& tg, f32x4& tm, f32x4& tv, bf16x4& ob) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gg = tg[e] * gscale;
      float pp = tp[e] * decay;
      tm[e] = b1 * tm[e] + (1.f - b1) * gg;
      tv[e] = b2 * tv[e] + (1.f - b2) * gg * gg;
      pp -= step * tm[e] / (sqrtf(tv[e]) * rbc2 + eps);
      tp[e] = pp;
      ob[e] = f2bf(pp);
    }
  };
  if (mode == 1) {
    // Every block owns ONE contiguous chunk of each of the nine streams (round 5; rounds 1-4 walked the buffers with a grid stride: at any
    // moment the 2048 blocks then touched 2048 separate 4-KB pieces of every stream) and keeps two 16-B vectors per stream and thread in
    // flight: 4.50 -> 5.15 TB/s on 174 M parameters, 3.90 -> 4.27 TB/s on a slower box (profiles/r5_ab_adamw_contiguous.txt); block count
    // (256 .. 8192) and 4 vectors in flight: within 2 %; non-temporal stores: no gain.
    constexpr int U = 2;
    const int64_t per = ((n4 + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;          // whole 1024-gradient chunks (keep bitmap granularity)
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
    for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += (int64_t)U * 256) {
      f32x4 tp[U], tg[U], tm[U], tv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + u * 256 < hi ? i0 + u * 256 : i0;
        tp[u] = LD4(p, i); tg[u] = LD4(g, i); tm[u] = LD4(m, i); tv[u] = LD4(v, i);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + u * 256;
        if (i < hi) {
          bf16x4 ob;
          update(tp[u], tg[u], tm[u], tv[u], ob);
          ST4(p, i, tp[u]); ST4(m, i, tm[u]); ST4(v, i, tv[u]);
          if (pb) reinterpret_cast<bf16x4*>(pb)[i] = ob;
          if (zero_vec(i)) ST4(g, i, (f32x4{0.f, 0.f, 0.f, 0.f}));
        }
      }
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
      f32x4 tp = LD4(p, i), tg = LD4(g, i);
      f32x4 tm = LD4(m, i), tv = LD4(v, i);
      bf16x4 ob;
      update(tp, tg, tm, tv, ob);
      ST4(p, i, tp);
      ST4(m, i, tm);
      ST4(v, i, tv);
      if (pb) reinterpret_cast<bf16x4*>(pb)[i] = ob;
      if (zero_vec(i)) ST4(g, i, (f32x4{0.f, 0.f, 0.f, 0.f}));      // optimizer.zero_grad() of the NEXT step
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    int64_t i = (n4 << 2) + threadIdx.x;
    float gg = g[i] * gscale, pp = p[i] * decay;
    float mm = b1 * m[i] + (1.f - b1) * gg, vv = b2 * v[i] + (1.f - b2) * gg * gg;
    pp -= step * mm / (sqrtf(vv) * rbc2 + eps);
    p[i] = pp; m[i] = mm; v[i] = vv;
    if (pb) pb[i] = f2bf(pp);
    if (zero_g) g[i] = 0.f;
  }
}

__global__ __launch_bounds__(256) void cast_f2b_kernel(const float* __restrict__ x, bf16* __restrict__ y, int64_t n, float scale) {
  int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 t = reinterpret_cast<const f32x4*>(x)[i];
    bf16x4 o = {f2bf(t[0] * scale), f2bf(t[1] * scale), f2bf(t[2] * scale), f2bf(t[3] * scale)};
    reinterpret_cast<bf16x4*>(y)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { int64_t i = (n4 << 2) + threadIdx.x; y[i] = f2bf(x[i] * scale); }
}
__global__ __launch_bounds__(256) void cast_b2f_kernel(const bf16* __restrict__ x, float* __restrict__ y, int64_t n) {
  int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    bf16x4 t = reinterpret_cast<const bf16x4*>(x)[i];
    f32x4 o = {bf2f(t[0]), bf2f(t[1]), bf2f(t[2]), bf2f(t[3])};
    reinterpret_cast<f32x4*>(y)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { int64_t i = (n4 << 2) + threadIdx.x; y[i] = bf2f(x[i]); }
}
__global__ __launch_bounds__(256) void add_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, bf16* __restrict__ y, int64_t n) {
  int64_t n8 = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    bf16x8 ta = reinterpret_cast<const bf16x8*>(a)[i], tb = reinterpret_cast<const bf16x8*>(b)[i], o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(ta[e]) + bf2f(tb[e]));
    reinterpret_cast<bf16x8*>(y)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) { int64_t i = (n8 << 3) + threadIdx.x; y[i] = f2bf(bf2f(a[i]) + bf2f(b[i])); }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ pre, bf16* __restrict__ dx, int64_t n, int act) {
  int64_t n8 = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    bf16x8 td = reinterpret_cast<const bf16x8*>(dy)[i], tp = reinterpret_cast<const bf16x8*>(pre)[i], o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(td[e]) * act_grad(act, bf2f(tp[e])));
    reinterpret_cast<bf16x8*>(dx)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) { int64_t i = (n8 << 3) + threadIdx.x; dx[i] = f2bf(bf2f(dy[i]) * act_grad(act, bf2f(pre[i]))); }
}

// out[n] += sum_m x[m][n].  Block = 32 column-chunks (8 columns each) x 8 row lanes; grid = (column panels, row strips).
// Rows are folded in registers, the 8 row lanes through LDS, and each block issues ONE fp32 atomic per column
// (device-scope atomics are slow on the multi-XCD part: ~35 / ns chip-wide, so they are kept to strips x N).
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ x, int M, int N, int ld, float* __restrict__ out) {
  __shared__ float red[8][256];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 256 + cl * 8;
  const int rows_per = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (c0 + 8 <= N) {
    for (int r = r0 + rl; r < r1; r += 8) {
      bf16x8 t = *reinterpret_cast<const bf16x8*>(x + (int64_t)r * ld + c0);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += bf2f(t[e]);
    }
  } else if (c0 < N) {
    for (int r = r0 + rl; r < r1; r += 8)
      for (int e = 0; e < 8 && c0 + e < N; ++e) acc[e] += bf2f(x[(int64_t)r * ld + c0 + e]);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cl * 8 + e] = acc[e];
  __syncthreads();
  int col = blockIdx.x * 256 + threadIdx.x;
  if (col < N) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][threadIdx.x];
    atomicAdd(out + col, s);
  }
}

__device__ __forceinline__ int map_row(const ph_rowmap& m, int r) {
  return m.seg_in ? (r / m.seg_in) * m.seg_out + m.seg_off + (r % m.seg_in) : r;
}
__global__ __launch_bounds__(256) void copy_rows_kernel(const bf16* __restrict__ src, int lds_, ph_rowmap smap, bf16* __restrict__ dst, int ldd,
                                                        ph_rowmap dmap, int rows, int cols, int accumulate) {
  const int cpr = cols / 8;
  int64_t total = (int64_t)rows * cpr;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    int c = (int)(id % cpr) * 8, r = (int)(id / cpr);
    bf16x8 t = *reinterpret_cast<const bf16x8*>(src + (int64_t)map_row(smap, r) * lds_ + c);
    bf16* d = dst + (int64_t)map_row(dmap, r) * ldd + c;
    if (accumulate) {
      bf16x8 o = *reinterpret_cast<const bf16x8*>(d);
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = f2bf(bf2f(t[e]) + bf2f(o[e]));
    }
    *reinterpret_cast<bf16x8*>(d) = t;
  }
}

// dst[r] = src[idx[r]]: beam reordering of the decoder's self-attention K/V caches (generate: the hypotheses kept after a beam
// step are a permutation-with-repeats of the previous ones)
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16* __restrict__ src, int lds_, const int32_t* __restrict__ idx,
                                                          bf16* __restrict__ dst, int ldd, int rows, int cols) {
  const int cpr = cols / 8;
  int64_t total = (int64_t)rows * cpr;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    int c = (int)(id % cpr) * 8, r = (int)(id / cpr);
    *reinterpret_cast<bf16x8*>(dst + (int64_t)r * ldd + c) = *reinterpret_cast<const bf16x8*>(src + (int64_t)idx[r] * lds_ + c);
  }
}

// w[Cout][Cin][ks][ks] fp32  ->  shadow[Cout][Kp] bf16 with column (ky*ks+kx)*Cin + c (zero padded)
__global__ void conv_w_shadow_kernel(const float* __restrict__ w, bf16* __restrict__ s, int Cout, int Cin, int ks, int Kp) {
  int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (int64_t)Cout * Kp) return;
  int k = (int)(id % Kp), o = (int)(id / Kp);
  float v = 0.f;
  if (k < ks * ks * Cin) {
    int tap = k / Cin, c = k % Cin;
    v = w[((int64_t)o * Cin + c) * ks * ks + tap];
  }
  s[id] = f2bf(v);
}
__global__ void conv_g_shadow_kernel(const float* __restrict__ ds, float* __restrict__ dw, int Cout, int Cin, int ks, int Kp) {
  int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int K = ks * ks * Cin;
  if (id >= (int64_t)Cout * K) return;
  int k = (int)(id % K), o = (int)(id / K);
  int tap = k / Cin, c = k % Cin;
  dw[((int64_t)o * Cin + c) * ks * ks + tap] += ds[(int64_t)o * Kp + k];
}

// grouped forms: all conv layers of the stems in one grid (block -> (layer, element block) through a table in the kernel args)
struct ConvFoldGroup {
  int n;
  int blk_start[PH_CONV_GROUP_MAX + 1];
  struct Item { const float* src; void* dst; int Cout, Cin, ks, Kp; } it[PH_CONV_GROUP_MAX];
};
// One block per output channel: the row (ks*ks*Cin values) passes through LDS so that BOTH the global read and the global write
// are contiguous (the direct form read or wrote with a stride of ks*ks floats: 250 / 210 us per step for 25 M stem parameters).
__global__ __launch_bounds__(256) void conv_g_shadow_grouped_kernel(ConvFoldGroup g) {
  extern __shared__ float row[];
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.blk_start[i + 1]) ++i;
  const ConvFoldGroup::Item& t = g.it[i];
  const int o = (int)blockIdx.x - g.blk_start[i];
  const int T = t.ks * t.ks, K = T * t.Cin;
  const float* src = t.src + (int64_t)o * t.Kp;
  for (int k = threadIdx.x; k < K; k += 256) row[k] = src[k];
  __syncthreads();
  float* dst = reinterpret_cast<float*>(t.dst) + (int64_t)o * K;
  for (int j = threadIdx.x; j < K; j += 256) {
    const int c = j / T, tap = j - c * T;
    dst[j] += row[tap * t.Cin + c];
  }
}
__global__ __launch_bounds__(256) void conv_w_shadow_grouped_kernel(ConvFoldGroup g) {
  extern __shared__ float row[];
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.blk_start[i + 1]) ++i;
  const ConvFoldGroup::Item& t = g.it[i];
  const int o = (int)blockIdx.x - g.blk_start[i];
  const int T = t.ks * t.ks, K = T * t.Cin;
  const float* src = t.src + (int64_t)o * K;
  for (int j = threadIdx.x; j < K; j += 256) row[j] = src[j];
  __syncthreads();
  bf16* dst = reinterpret_cast<bf16*>(t.dst) + (int64_t)o * t.Kp;
  for (int k = threadIdx.x; k < t.Kp; k += 256) {
    float v = 0.f;
    if (k < K) { const int tap = k / t.Cin, c = k - tap * t.Cin; v = row[c * T + tap]; }
    dst[k] = f2bf(v);
  }
}

__global__ void advance_seed_kernel(uint64_t* seed) {
  uint64_t z = seed[0] + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  seed[0] = z ^ (z >> 31);
}

// ---- layout probe: (1) ds_read_b64_tr_b16 semantics, (2) MFMA 16x16x32 and 32x32x16 C layouts -----------
// in : bf16[64*16] filled by the host with value = index.
// out[0..255]      : lane l, element j of ONE tr-read with lane-linear addresses (lane l -> element 4*l)
// out[256..511]    : C of mfma_16x16x32(A=I-like, B) ... see tests/test_kernels_gpu.py for the expectation
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
__global__ void probe_kernel(const bf16* __restrict__ in, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) bf16 lds[1024];
  int l = threadIdx.x;
  for (int i = l; i < 1024; i += 64) lds[i] = in[i];
  __syncthreads();
  bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lds + 4 * l));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = bf2f(t[j]);
  // MFMA 16x16x32: A[i][k] = (i == k ? 1 : 0) for k < 16 ; B[k][j] = in[k*16 + j] -> C[i][j] = B[i][j] (i < 16)
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    int k = (l >> 4) * 8 + j;
    a[j] = f2bf(((l & 15) == k) ? 1.f : 0.f);
    b[j] = (k < 16) ? in[k * 16 + (l & 15)] : f2bf(0.f);
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[256 + l * 4 + r] = c[r];
  // MFMA 32x32x16: A[i][k] = (i == k) (k < 16), B[k][j] = in[k*32 + j]  -> C[i][j] = in[i*32 + j] for i < 16, 0 otherwise
  bf16x8 a2, b2;
  for (int j = 0; j < 8; ++j) {
    int k = (l >> 5) * 8 + j;
    a2[j] = f2bf(((l & 31) == k) ? 1.f : 0.f);
    b2[j] = in[k * 32 + (l & 31)];
  }
  f32x16 c2;
  for (int r = 0; r < 16; ++r) c2[r] = 0.f;
  c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, c2, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[512 + l * 16 + r] = c2[r];
}

// ---- weight operand of the implicit data gradient (include/prismer_hip.h: ph_conv_dgrad_shadow_grouped) ----------------------------
struct ConvDgradGroup {
  int n;
  int blk_start[PH_CONV_GROUP_MAX + 1];
  ph_conv_dgrad_item it[PH_CONV_GROUP_MAX];
};
// one block per (layer, 32 x 32 tile of [Cout x Cin]): the nine taps of the tile are read row-wise from w[co][ci][3][3] (288
// consecutive floats per co), transposed through LDS and written as 64-B runs of consecutive co (first version: one thread per
// output element with a Cin*36-byte read stride -- 313 us per step for the 18 layers)
__global__ __launch_bounds__(256) void conv_dgrad_shadow_grouped_kernel(ConvDgradGroup g) {
  __shared__ float tile[32][32 * 9 + 1];
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.blk_start[i + 1]) ++i;
  const ph_conv_dgrad_item& a = g.it[i];
  const int tiles_ci = (a.Cin + 31) / 32;
  const int tb = blockIdx.x - g.blk_start[i];
  const int co0 = (tb / tiles_ci) * 32, ci0 = (tb % tiles_ci) * 32;
  const int nci = min(32, a.Cin - ci0), nco = min(32, a.Cout - co0);
  for (int e = threadIdx.x; e < 32 * 288; e += 256) {             // tile[co][(ci, ky, kx)]
    const int co = e / 288, r = e - co * 288;
    tile[co][r] = (co < nco && r < nci * 9) ? a.w[((size_t)(co0 + co) * a.Cin + ci0) * 9 + r] : 0.f;
  }
  __syncthreads();
  bf16* dst = reinterpret_cast<bf16*>(a.dst);
  for (int e = threadIdx.x; e < 32 * 9 * 32; e += 256) {          // (ci, t, co): co fastest
    const int co = e & 31, t = (e >> 5) % 9, ci = e / (32 * 9);
    if (co >= nco || ci >= nci) continue;
    int ky, kx;
    if (a.stride == 1) {
      ky = 2 - t / 3; kx = 2 - t % 3;
    } else {                                                       // parity classes (0,0) | (0,1) | (1,0) | (1,1) back to back
      const int py = t >= 3, px = (t >= 1 && t < 3) || t >= 5;
      const int tt = t - (t >= 5 ? 5 : (t >= 3 ? 3 : (t >= 1 ? 1 : 0)));
      const int ty = px && py ? tt >> 1 : (py ? tt : 0), tx = px && py ? tt & 1 : (px ? tt : 0);
      ky = py ? (ty ? 0 : 2) : 1;
      kx = px ? (tx ? 0 : 2) : 1;
    }
    dst[(size_t)(ci0 + ci) * 9 * a.Cout + (size_t)t * a.Cout + co0 + co] = f2bf(tile[co][ci * 9 + ky * 3 + kx]);
  }
}
inline int grid_for(int64_t work_items) { return (int)std::min<int64_t>(ceil_div64(work_items, 256), 256 * 16); }

}  // namespace

extern "C" int ph_adamw_keep(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, const float* hyper, float beta1,
                             float beta2, float eps, float weight_decay, float grad_scale, int zero_grad, const uint32_t* keep_bitmap,
                             hipStream_t stream) {
  PH_CHECK_ARG(p && g && m && v && hyper && n > 0, "ph_adamw: bad args");
  ProfScope prof__(PH_FAM_OPTIM, 0.0, 30.0 * (double)n, stream);
  PH_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && (((uintptr_t)p_bf16) & 7) == 0, "ph_adamw: misaligned");
  constexpr int mode = 1;             // contiguous chunk per block, two vectors per stream in flight (0: the round-1 one-vector grid-stride loop)
  constexpr int blocks_cap = 2048;    // 8 blocks per CU
  int grid = grid_for(n / 4 + 1);
  if (blocks_cap > 0 && grid > blocks_cap) grid = blocks_cap;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, stream, p, g, m, v, (bf16*)p_bf16, n, hyper, beta1, beta2, eps,
                     weight_decay, grad_scale, zero_grad, mode, keep_bitmap);
  PH_LAUNCH_CHECK("adamw_kernel");
  return PH_OK;
}
extern "C" int ph_adamw(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, const float* hyper, float beta1,
                        float beta2, float eps, float weight_decay, float grad_scale, int zero_grad, hipStream_t stream) {
  return ph_adamw_keep(p, g, m, v, p_bf16, n, hyper, beta1, beta2, eps, weight_decay, grad_scale, zero_grad, nullptr, stream);
}
extern "C" int ph_cast_f32_to_bf16(const float* x, void* y, int64_t n, hipStream_t stream) {
  PH_CHECK_ARG(x && y && n > 0 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)y) & 7) == 0, "ph_cast_f32_to_bf16: bad args");
  ProfScope prof__(PH_FAM_MISC, 0.0, 0.0, stream, "ph_cast_f32_to_bf16");
  hipLaunchKernelGGL(cast_f2b_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, stream, x, (bf16*)y, n, 1.0f);
  PH_LAUNCH_CHECK("cast_f2b_kernel");
  return PH_OK;
}
extern "C" int ph_scale_cast_f32_to_bf16(const float* x, void* y, int64_t n, float scale, hipStream_t stream) {
  PH_CHECK_ARG(x && y && n > 0 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)y) & 7) == 0, "ph_scale_cast_f32_to_bf16: bad args");
  ProfScope prof__(PH_FAM_MISC, 0.0, 0.0, stream, "ph_scale_cast_f32_to_bf16");
  hipLaunchKernelGGL(cast_f2b_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, stream, x, (bf16*)y, n, scale);
  PH_LAUNCH_CHECK("cast_f2b_kernel");
  return PH_OK;
}
extern "C" int ph_cast_bf16_to_f32(const void* x, float* y, int64_t n, hipStream_t stream) {
  PH_CHECK_ARG(x && y && n > 0 && (((uintptr_t)y) & 15) == 0 && (((uintptr_t)x) & 7) == 0, "ph_cast_bf16_to_f32: bad args");
  ProfScope prof__(PH_FAM_MISC, 0.0, 0.0, stream, "ph_cast_bf16_to_f32");
  hipLaunchKernelGGL(cast_b2f_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, stream, (const bf16*)x, y, n);
  PH_LAUNCH_CHECK("cast_b2f_kernel");
  return PH_OK;
}
namespace {
struct ColsumGroup {
  int n;
  int blk_start[PH_GEMM_GROUP_MAX + 1];
  struct Item { const bf16* x; float* out; int M, N, ld, ncb; } it[PH_GEMM_GROUP_MAX];
};
// several bias-gradient column sums in one grid (block -> (problem, column block, row strip))
__global__ __launch_bounds__(256) void colsum_grouped_kernel(ColsumGroup g) {
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.blk_start[i + 1]) ++i;
  const ColsumGroup::Item& t = g.it[i];
  const int lid = (int)blockIdx.x - g.blk_start[i];
  const int strips = (g.blk_start[i + 1] - g.blk_start[i]) / t.ncb;
  const int cb = lid % t.ncb, strip = lid / t.ncb;
  __shared__ float red[8][256];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c0 = cb * 256 + cl * 8;
  const int rows_per = (t.M + strips - 1) / strips;
  const int r0 = strip * rows_per, r1 = min(t.M, r0 + rows_per);
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (c0 + 8 <= t.N) {
#ifndef PH_COLSUM_UNROLL
#define PH_COLSUM_UNROLL 4            // (round 6 A/B under the step: 1 -> 32.9, 4 -> 30.0, 8 -> 33.0 us per launch)
#endif
#pragma unroll PH_COLSUM_UNROLL
    for (int r = r0 + rl; r < r1; r += 8) {
      bf16x8 v = *reinterpret_cast<const bf16x8*>(t.x + (int64_t)r * t.ld + c0);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
    }
  } else if (c0 < t.N) {
    for (int r = r0 + rl; r < r1; r += 8)
      for (int e = 0; e < 8 && c0 + e < t.N; ++e) acc[e] += bf2f(t.x[(int64_t)r * t.ld + c0 + e]);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cl * 8 + e] = acc[e];
  __syncthreads();
  int col = cb * 256 + threadIdx.x;
  if (col < t.N) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][threadIdx.x];
    atomicAdd(t.out + col, s);
  }
}
}  // namespace

extern "C" int ph_colsum_grouped_bf16(const ph_colsum_item* items, int n, hipStream_t stream) {
  PH_CHECK_ARG(items && n >= 1 && n <= PH_GEMM_GROUP_MAX, "ph_colsum_grouped_bf16: need 1..%d items, got %d", PH_GEMM_GROUP_MAX, n);
  ProfScope prof__(PH_FAM_MISC, 0.0, 0.0, stream, "ph_colsum_grouped_bf16");
  ColsumGroup g;
  g.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    const ph_colsum_item& a = items[i];
    PH_CHECK_ARG(a.x && a.out && a.M > 0 && a.N > 0 && a.ld % 8 == 0 && (((uintptr_t)a.x) & 15) == 0, "ph_colsum_grouped_bf16: bad item %d", i);
    int strips = std::max(1, std::min(64, a.M / 64));
    g.it[i].x = (const bf16*)a.x; g.it[i].out = a.out; g.it[i].M = a.M; g.it[i].N = a.N; g.it[i].ld = a.ld; g.it[i].ncb = ceil_div(a.N, 256);
    g.blk_start[i] = total;
    total += g.it[i].ncb * strips;
  }
  g.blk_start[n] = total;
  hipLaunchKernelGGL(colsum_grouped_kernel, dim3(total), dim3(256), 0, stream, g);
  PH_LAUNCH_CHECK("colsum_grouped_kernel");
  return PH_OK;
}

extern "C" int ph_colsum_bf16(const void* x, int M, int N, int ld, float* out, hipStream_t stream) {
  PH_CHECK_ARG(x && out && M > 0 && N > 0 && ld % 8 == 0 && (((uintptr_t)x) & 15) == 0, "ph_colsum_bf16: bad args");
  ProfScope prof__(PH_FAM_MISC, 0.0, 0.0, stream, "ph_colsum_bf16");
  int strips = std::max(1, std::min(64, M / 64));
  hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(N, 256), strips), dim3(256), 0, stream, (const bf16*)x, M, N, ld, out);
  PH_LAUNCH_CHECK("colsum_kernel");
  return PH_OK;
}
extern "C" int ph_add_bf16(const void* a, const void* b, void* y, int64_t n, hipStream_t stream) {
  PH_CHECK_ARG(a && b && y && n > 0 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)y)) & 15) == 0, "ph_add_bf16: bad args");
  ProfScope prof__(PH_FAM_MISC, 0.0, 0.0, stream, "ph_add_bf16");
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 8 + 1)), dim3(256), 0, stream, (const bf16*)a, (const bf16*)b, (bf16*)y, n);
  PH_LAUNCH_CHECK("add_kernel");
  return PH_OK;
}
extern "C" int ph_act_bwd_bf16(const void* dy, const void* pre, void* dx, int64_t n, int act, hipStream_t stream) {
  PH_CHECK_ARG(dy && pre && dx && n > 0 && ((((uintptr_t)dy) | ((uintptr_t)pre) | ((uintptr_t)dx)) & 15) == 0, "ph_act_bwd_bf16: bad args");
  ProfScope prof__(PH_FAM_MISC, 0.0, 0.0, stream, "ph_act_bwd_bf16");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n / 8 + 1)), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)pre, (bf16*)dx, n, act);
  PH_LAUNCH_CHECK("act_bwd_kernel");
  return PH_OK;
}
extern "C" int ph_copy_rows_bf16(const void* src, int lds, ph_rowmap src_map, void* dst, int ldd, ph_rowmap dst_map, int rows,
                                 int cols, int accumulate, hipStream_t stream) {
  PH_CHECK_ARG(src && dst && rows > 0 && cols > 0 && cols % 8 == 0 && lds % 8 == 0 && ldd % 8 == 0, "ph_copy_rows_bf16: bad args");
  ProfScope prof__(PH_FAM_MISC, 0.0, 0.0, stream, "ph_copy_rows_bf16");
  hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for((int64_t)rows * cols / 8)), dim3(256), 0, stream, (const bf16*)src, lds, src_map,
                     (bf16*)dst, ldd, dst_map, rows, cols, accumulate);
  PH_LAUNCH_CHECK("copy_rows_kernel");
  return PH_OK;
}
extern "C" int ph_gather_rows_bf16(const void* src, int64_t lds, const int32_t* idx, void* dst, int64_t ldd, int rows, int cols, hipStream_t stream) {
  PH_CHECK_ARG(src && dst && idx && src != dst && rows > 0 && cols > 0 && cols % 8 == 0 && lds % 8 == 0 && ldd % 8 == 0 && lds < (1ll << 31) &&
               ldd < (1ll << 31), "ph_gather_rows_bf16: bad args");
  ProfScope prof__(PH_FAM_MISC, 0.0, 0.0, stream, "ph_gather_rows_bf16");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((int64_t)rows * cols / 8)), dim3(256), 0, stream, (const bf16*)src, (int)lds, idx,
                     (bf16*)dst, (int)ldd, rows, cols);
  PH_LAUNCH_CHECK("gather_rows_kernel");
  return PH_OK;
}
extern "C" int ph_conv_weight_to_shadow(const float* w, void* shadow, int Cout, int Cin, int ks, int Kp, hipStream_t stream) {
  PH_CHECK_ARG(w && shadow && Kp >= Cin * ks * ks, "ph_conv_weight_to_shadow: bad args");
  ProfScope prof__(PH_FAM_MISC, 0.0, 0.0, stream, "ph_conv_weight_to_shadow");
  hipLaunchKernelGGL(conv_w_shadow_kernel, dim3((unsigned)ceil_div64((int64_t)Cout * Kp, 256)), dim3(256), 0, stream, w, (bf16*)shadow, Cout, Cin, ks, Kp);
  PH_LAUNCH_CHECK("conv_w_shadow_kernel");
  return PH_OK;
}
extern "C" int ph_conv_grad_from_shadow(const float* dshadow, float* dw, int Cout, int Cin, int ks, int Kp, hipStream_t stream) {
  PH_CHECK_ARG(dshadow && dw && Kp >= Cin * ks * ks, "ph_conv_grad_from_shadow: bad args");
  ProfScope prof__(PH_FAM_MISC, 0.0, 0.0, stream, "ph_conv_grad_from_shadow");
  hipLaunchKernelGGL(conv_g_shadow_kernel, dim3((unsigned)ceil_div64((int64_t)Cout * Cin * ks * ks, 256)), dim3(256), 0, stream, dshadow, dw, Cout, Cin, ks, Kp);
  PH_LAUNCH_CHECK("conv_g_shadow_kernel");
  return PH_OK;
}
static int conv_group_launch(const ph_conv_layout_item* items, int n, bool to_shadow, hipStream_t stream, const char* who) {
  PH_CHECK_ARG(items && n >= 1 && n <= PH_CONV_GROUP_MAX, "%s: need 1..%d items, got %d", who, PH_CONV_GROUP_MAX, n);
  ConvFoldGroup g;
  g.n = n;
  int total = 0, max_k = 0;
  for (int i = 0; i < n; ++i) {
    const ph_conv_layout_item& a = items[i];
    PH_CHECK_ARG(a.src && a.dst && a.Kp >= a.Cin * a.ks * a.ks, "%s: bad item %d", who, i);
    g.it[i].src = a.src; g.it[i].dst = a.dst; g.it[i].Cout = a.Cout; g.it[i].Cin = a.Cin; g.it[i].ks = a.ks; g.it[i].Kp = a.Kp;
    g.blk_start[i] = total;
    total += a.Cout;                         // one block per output channel
    max_k = std::max(max_k, a.Cin * a.ks * a.ks);
  }
  g.blk_start[n] = total;
  PH_CHECK_ARG(max_k * 4 <= 64 * 1024, "%s: ks*ks*Cin = %d too large for the LDS row", who, max_k);
  if (to_shadow) hipLaunchKernelGGL(conv_w_shadow_grouped_kernel, dim3(total), dim3(256), sizeof(float) * max_k, stream, g);
  else hipLaunchKernelGGL(conv_g_shadow_grouped_kernel, dim3(total), dim3(256), sizeof(float) * max_k, stream, g);
  PH_LAUNCH_CHECK(who);
  return PH_OK;
}
extern "C" int ph_conv_weight_to_shadow_grouped(const ph_conv_layout_item* items, int n, hipStream_t stream) {
  ProfScope prof__(PH_FAM_MISC, 0.0, 0.0, stream, "ph_conv_weight_to_shadow_grouped");
  return conv_group_launch(items, n, true, stream, "ph_conv_weight_to_shadow_grouped");
}
extern "C" int ph_conv_grad_from_shadow_grouped(const ph_conv_layout_item* items, int n, hipStream_t stream) {
  ProfScope prof__(PH_FAM_MISC, 0.0, 0.0, stream, "ph_conv_grad_from_shadow_grouped");
  return conv_group_launch(items, n, false, stream, "ph_conv_grad_from_shadow_grouped");
}
extern "C" int ph_conv_dgrad_shadow_grouped(const ph_conv_dgrad_item* items, int n, hipStream_t stream) {
  PH_CHECK_ARG(items && n >= 1 && n <= PH_CONV_GROUP_MAX, "ph_conv_dgrad_shadow_grouped: need 1..%d items, got %d", PH_CONV_GROUP_MAX, n);
  ProfScope prof__(PH_FAM_MISC, 0.0, 0.0, stream, "ph_conv_dgrad_shadow_grouped");
  ConvDgradGroup g;
  g.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    PH_CHECK_ARG(items[i].w && items[i].dst && items[i].Cout > 0 && items[i].Cin > 0 && (items[i].stride == 1 || items[i].stride == 2),
                 "ph_conv_dgrad_shadow_grouped: bad item %d (3x3 kernels, stride 1 or 2)", i);
    g.it[i] = items[i];
    g.blk_start[i] = total;
    total += ((items[i].Cin + 31) / 32) * ((items[i].Cout + 31) / 32);
  }
  g.blk_start[n] = total;
  hipLaunchKernelGGL(conv_dgrad_shadow_grouped_kernel, dim3(total), dim3(256), 0, stream, g);
  PH_LAUNCH_CHECK("conv_dgrad_shadow_grouped_kernel");
  return PH_OK;
}
// ---- per-step host scalars without a copy engine (round 6): the learning rate / Adam bias corrections and the instance-embedding draw
// table travel in the KERNEL ARGUMENTS (<= 1.25 KB by value) instead of two pinned hipMemcpyAsync.  Those copies ran on the SDMA queue the
// loader's 56-MB host-to-device prefetch also uses, and the compute stream waited behind it at the start of every step
// (tools/loader_probe.py: +0.45 ms per step with a loader attached).
namespace {
struct StepWords { uint32_t w[PH_STORE_WORDS_MAX]; };
__global__ void store_words_kernel(uint32_t* __restrict__ dst0, int n0, uint32_t* __restrict__ dst1, int n1, StepWords v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n0) dst0[i] = v.w[i];
  else if (i < n0 + n1) dst1[i - n0] = v.w[i];
}
}  // namespace
extern "C" int ph_store_words(void* dst0, int n0, void* dst1, int n1, const uint32_t* host_words, hipStream_t stream) {
  PH_CHECK_ARG(dst0 && host_words && n0 > 0 && n1 >= 0 && (n1 == 0 || dst1) && n0 + n1 <= PH_STORE_WORDS_MAX,
               "ph_store_words: need dst0, host words and 1 <= n0 + n1 <= %d (got %d + %d)", PH_STORE_WORDS_MAX, n0, n1);
  ProfScope prof__(PH_FAM_MISC, 0.0, 0.0, stream, "ph_store_words");
  StepWords v;
  for (int i = 0; i < n0 + n1; ++i) v.w[i] = host_words[i];
  for (int i = n0 + n1; i < PH_STORE_WORDS_MAX; ++i) v.w[i] = 0u;
  hipLaunchKernelGGL(store_words_kernel, dim3((n0 + n1 + 255) / 256), dim3(256), 0, stream, (uint32_t*)dst0, n0, (uint32_t*)dst1, n1, v);
  PH_LAUNCH_CHECK("store_words_kernel");
  return PH_OK;
}
extern "C" int ph_advance_seed(uint64_t* seed, hipStream_t stream) {
  PH_CHECK_ARG(seed, "ph_advance_seed: null");
  ProfScope prof__(PH_FAM_MISC, 0.0, 0.0, stream, "ph_advance_seed");
  hipLaunchKernelGGL(advance_seed_kernel, dim3(1), dim3(1), 0, stream, seed);
  PH_LAUNCH_CHECK("advance_seed_kernel");
  return PH_OK;
}
// ---- step glue that used to run as stock torch kernels inside the captured step (round-3 review): BatchNorm's num_batches_tracked += 1
// (torch._foreach_add_) and the batch loss (stack().sum() / B, (weights * loss).mean())
namespace {
__global__ void add_i64_kernel(int64_t* x, int n, int64_t v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] += v;
}
__global__ __launch_bounds__(256) void weighted_sum_kernel(const float* __restrict__ x, const float* __restrict__ w, int n, float scale, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += w ? x[i] * w[i] : x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) * scale;
}
}  // namespace
namespace {
// zero fill / flat copy as KERNELS (16-B vectors; n16 = bytes / 16): what the captured step uses instead of torch.zeros / Tensor.copy_, so that a
// replayed segment holds nothing but this library's kernel nodes (round 5: a memset NODE of a captured hipGraph does not replay correctly on
// ROCm 7.0 -- tools/graph_memset_probe.py -- and a memcpy node is one bug report away from the same)
__global__ __launch_bounds__(256) void fill_zero_kernel(u32x4* __restrict__ p, int64_t n16) {
  const u32x4 z = {0u, 0u, 0u, 0u};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) p[i] = z;
}
__global__ __launch_bounds__(256) void copy_bytes_kernel(u32x4* __restrict__ d, const u32x4* __restrict__ s, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) d[i] = s[i];
}
}  // namespace
extern "C" int ph_fill_zero(void* p, int64_t bytes, hipStream_t stream) {
  PH_CHECK_ARG(p && bytes > 0 && (bytes % 16) == 0 && (((uintptr_t)p) & 15) == 0, "ph_fill_zero: need a 16-B aligned buffer of a multiple of 16 bytes");
  ProfScope prof__(PH_FAM_MISC, 0.0, 0.0, stream, "ph_fill_zero");
  hipLaunchKernelGGL(fill_zero_kernel, dim3(grid_for(bytes / 16)), dim3(256), 0, stream, (u32x4*)p, bytes / 16);
  PH_LAUNCH_CHECK("fill_zero_kernel");
  return PH_OK;
}
extern "C" int ph_copy_bytes(void* dst, const void* src, int64_t bytes, hipStream_t stream) {
  PH_CHECK_ARG(dst && src && bytes > 0 && (bytes % 16) == 0 && ((((uintptr_t)dst) | ((uintptr_t)src)) & 15) == 0,
               "ph_copy_bytes: need 16-B aligned buffers of a multiple of 16 bytes");
  ProfScope prof__(PH_FAM_MISC, 0.0, 0.0, stream, "ph_copy_bytes");
  hipLaunchKernelGGL(copy_bytes_kernel, dim3(grid_for(bytes / 16)), dim3(256), 0, stream, (u32x4*)dst, (const u32x4*)src, bytes / 16);
  PH_LAUNCH_CHECK("copy_bytes_kernel");
  return PH_OK;
}
extern "C" int ph_add_i64(int64_t* x, int n, int64_t value, hipStream_t stream) {
  PH_CHECK_ARG(x && n > 0, "ph_add_i64: bad args");
  ProfScope prof__(PH_FAM_MISC, 0.0, 0.0, stream, "ph_add_i64");
  hipLaunchKernelGGL(add_i64_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, x, n, value);
  PH_LAUNCH_CHECK("add_i64_kernel");
  return PH_OK;
}
extern "C" int ph_weighted_sum_f32(const float* x, const float* weights, int n, float scale, float* out, hipStream_t stream) {
  PH_CHECK_ARG(x && out && n > 0, "ph_weighted_sum_f32: bad args");
  ProfScope prof__(PH_FAM_MISC, 0.0, 0.0, stream, "ph_weighted_sum_f32");
  hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(256), 0, stream, x, weights, n, scale, out);
  PH_LAUNCH_CHECK("weighted_sum_kernel");
  return PH_OK;
}
extern "C" int ph_probe_layouts(const void* in_bf16, float* out, hipStream_t stream) {
  PH_CHECK_ARG(in_bf16 && out, "ph_probe_layouts: null");
  hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, stream, (const bf16*)in_bf16, out);
  PH_LAUNCH_CHECK("probe_kernel");
  return PH_OK;
}
