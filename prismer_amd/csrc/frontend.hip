// Encoder front end for gfx950: patchify, expert-stem data movement (bilinear resize, im2col / col2im),
// train-mode BatchNorm statistics + backward, token assembly (pos / instance embeddings).
// Replaces the non-GEMM parts of model/modules/vit.py:86-160.  All of these are HBM-bound byte movers:
// every thread moves 16-B vectors along the channel (innermost NHWC) dimension.
#include "common.h"
#include <algorithm>

namespace {

// ------------------------------------------------------------------------------------------------ patchify
// vit.py:86,138: Conv2d(3, D, kernel=p, stride=p, bias=False) == GEMM over non-overlapping patches.
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, bf16* __restrict__ col, int B, int C,
                                                       int R, int p, int Kp) {
  int g = R / p;
  int64_t total = (int64_t)B * g * g * (Kp / 2);
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    int k2 = (int)(id % (Kp / 2));
    int64_t m = id / (Kp / 2);
    int gx = (int)(m % g), gy = (int)((m / g) % g), b = (int)(m / ((int64_t)g * g));
    bf16x2 o;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      int k = k2 * 2 + e;
      float v = 0.f;
      if (k < C * p * p) {
        int c = k % C, tap = k / C, py = tap / p, px = tap % p;   // column order (py, px, c): same as the conv shadows
        v = img[(((int64_t)b * C + c) * R + gy * p + py) * R + gx * p + px];
      }
      o[e] = f2bf(v);
    }
    *reinterpret_cast<bf16x2*>(col + m * Kp + k2 * 2) = o;
  }
}

// ------------------------------------------------------------------------------------------------ resize
// nn.UpsamplingBilinear2d == F.interpolate(mode='bilinear', align_corners=True) (vit.py:89,106), fused with the
// NCHW fp32 -> NHWC bf16 layout change.  One block per (b, out row, group of 8 channels): consecutive lanes walk
// consecutive x of one channel plane (coalesced source rows), the 8-channel pixel vectors are assembled in LDS
// and stored as 16-B pieces.
// per-sample min / max of a dense expert map, first stage: block (p, b) reduces a contiguous 1/nparts share of sample b's n values to one
// (min, max) pair; the consumer (resize_kernel) folds the nparts pairs itself -- no atomics, no second launch
__global__ __launch_bounds__(256) void dense_minmax_partial_kernel(const float* __restrict__ x, float* __restrict__ part, int64_t n, int nparts) {
  __shared__ float red[2][4];
  const int p = blockIdx.x, b = blockIdx.y;
  const int64_t per = (n + nparts - 1) / nparts, lo_i = p * per, hi_i = min(n, lo_i + per);
  const float* xs = x + (int64_t)b * n;
  float lo = INFINITY, hi = -INFINITY;
  for (int64_t i = lo_i + threadIdx.x; i < hi_i; i += 256) { const float v = xs[i]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = lo; red[1][threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[((int64_t)b * nparts + p) * 2] = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
    part[((int64_t)b * nparts + p) * 2 + 1] = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
  }
}

// mm_part != nullptr (round 6): the per-sample min-max remap of the dense experts (dataset/utils.py:120-121, `2 * (x - min) / (max - min + eps) - 1`
// with eps = 1e-6 over the whole [C, H, W] map of a sample) is applied to every tap BEFORE the interpolation, in the reference's expression
// order; the sample's min / max are folded from the `nparts` partial pairs ph_dense_minmax_partial left (32 cached loads per thread).
template <bool REMAP>          // (a template flag, not a run-time one: the plain instantiation must stay bit-identical to inpaint_resize_kernel's tap arithmetic)
__global__ __launch_bounds__(256) void resize_kernel(const float* __restrict__ x, bf16* __restrict__ y, int B, int C, int Hin,
                                                     int Win, int Hout, int Wout, const float* __restrict__ mm_part, int nparts) {
  __shared__ bf16 tile[1024 * 8];
  int cgroups = (C + 7) / 8;
  int cg = blockIdx.x % cgroups;
  int oy = (blockIdx.x / cgroups) % Hout;
  int b = blockIdx.x / (cgroups * Hout);
  int c0 = cg * 8, nc = min(8, C - c0);
  float mn = 0.f, den = 1.f;
  if constexpr (REMAP) {
    float lo = INFINITY, hi = -INFINITY;
    for (int p = 0; p < nparts; ++p) { lo = fminf(lo, mm_part[((int64_t)b * nparts + p) * 2]); hi = fmaxf(hi, mm_part[((int64_t)b * nparts + p) * 2 + 1]); }
    mn = lo; den = (hi - lo) + 1e-6f;
  }
  float sy = Hout > 1 ? (float)(Hin - 1) / (float)(Hout - 1) : 0.f;
  float sx = Wout > 1 ? (float)(Win - 1) / (float)(Wout - 1) : 0.f;
  float fy = oy * sy;
  int y0 = min((int)fy, Hin - 1), y1 = min(y0 + 1, Hin - 1);
  float wy = fy - (float)y0;
  for (int xb = 0; xb < Wout; xb += 1024) {
    int wchunk = min(1024, Wout - xb);
    for (int id = threadIdx.x; id < wchunk * nc; id += 256) {
      int cc = id / wchunk, ox = xb + id % wchunk;
      float fx = ox * sx;
      int x0 = min((int)fx, Win - 1), x1 = min(x0 + 1, Win - 1);
      float wx = fx - (float)x0;
      const float* pl = x + ((int64_t)b * C + c0 + cc) * Hin * Win;
      float v00 = pl[(int64_t)y0 * Win + x0], v01 = pl[(int64_t)y0 * Win + x1];
      float v10 = pl[(int64_t)y1 * Win + x0], v11 = pl[(int64_t)y1 * Win + x1];
      if constexpr (REMAP) {
        v00 = (2.f * (v00 - mn)) / den - 1.f; v01 = (2.f * (v01 - mn)) / den - 1.f;
        v10 = (2.f * (v10 - mn)) / den - 1.f; v11 = (2.f * (v11 - mn)) / den - 1.f;
      }
      float v = (1.f - wy) * ((1.f - wx) * v00 + wx * v01) + wy * ((1.f - wx) * v10 + wx * v11);
      tile[(ox - xb) * 8 + cc] = f2bf(v);
    }
    __syncthreads();
    bf16* orow = y + (((int64_t)b * Hout + oy) * Wout + xb) * C + c0;
    if (nc == 8 && (C % 8) == 0) {
      for (int ox = threadIdx.x; ox < wchunk; ox += 256)
        *reinterpret_cast<u32x4*>(orow + (int64_t)ox * C) = *reinterpret_cast<const u32x4*>(tile + ox * 8);
    } else {
      for (int id = threadIdx.x; id < wchunk * nc; id += 256) {
        int ox = id / nc, cc = id % nc;
        orow[(int64_t)ox * C + cc] = tile[ox * 8 + cc];
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ label in-painting
// post_label_process (dataset/utils.py:117-160) + the stem's UpsamplingBilinear2d (vit.py:88-90) in one pass: the label
// experts arrive as a uint8 label map [B, Hin, Win] plus a [256, C] fp32 table per image (row l = the CLIP feature the reference
// paints over label l, row 255 = background) instead of the 64-channel fp32 image the reference's CPU loader materialises
// (12.8 MB per image and expert: the 1.2 GB/step host-to-device stream of SURVEY 8f #2).  One thread per output pixel and
// 8-channel group: the four bilinear taps are looked up in the table (same tap positions, weights and expression as
// resize_kernel, so the result equals resize(in-painted image) to the last bit), NHWC bf16 out.
__global__ __launch_bounds__(256) void inpaint_resize_kernel(const uint8_t* __restrict__ lab, const float* __restrict__ table,
                                                             int64_t table_bs, bf16* __restrict__ y, int B, int C, int Hin, int Win,
                                                             int Hout, int Wout) {
  const int cgroups = C / 8;
  const int64_t total = (int64_t)B * Hout * Wout * cgroups;
  const float sy = Hout > 1 ? (float)(Hin - 1) / (float)(Hout - 1) : 0.f;
  const float sx = Wout > 1 ? (float)(Win - 1) / (float)(Wout - 1) : 0.f;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int cg = (int)(id % cgroups);
    const int64_t pix = id / cgroups;
    const int ox = (int)(pix % Wout), oy = (int)((pix / Wout) % Hout), b = (int)(pix / ((int64_t)Wout * Hout));
    const float fy = oy * sy, fx = ox * sx;
    const int y0 = min((int)fy, Hin - 1), y1 = min(y0 + 1, Hin - 1);
    const int x0 = min((int)fx, Win - 1), x1 = min(x0 + 1, Win - 1);
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const uint8_t* lb = lab + (int64_t)b * Hin * Win;
    const float* tb = table + (int64_t)b * table_bs + cg * 8;
    const float* r00 = tb + (int)lb[y0 * Win + x0] * C;
    const float* r01 = tb + (int)lb[y0 * Win + x1] * C;
    const float* r10 = tb + (int)lb[y1 * Win + x0] * C;
    const float* r11 = tb + (int)lb[y1 * Win + x1] * C;
    bf16x8 o;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 v00 = *reinterpret_cast<const f32x4*>(r00 + 4 * h), v01 = *reinterpret_cast<const f32x4*>(r01 + 4 * h);
      f32x4 v10 = *reinterpret_cast<const f32x4*>(r10 + 4 * h), v11 = *reinterpret_cast<const f32x4*>(r11 + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = (1.f - wy) * ((1.f - wx) * v00[e] + wx * v01[e]) + wy * ((1.f - wx) * v10[e] + wx * v11[e]);
        o[4 * h + e] = f2bf(v);
      }
    }
    *reinterpret_cast<bf16x8*>(y + pix * C + cg * 8) = o;
  }
}

// ------------------------------------------------------------------------------------------------ im2col
// id -> (id / d, id % d): 64-bit integer division costs ~100 instructions per element on the GPU; every index space of this
// model fits 32 bits, so the (uniform) fast path divides unsigned 32-bit numbers
__device__ __forceinline__ void divmod(int64_t id, int d, bool fits32, int64_t& q, int& r) {
  if (fits32) { unsigned u = (unsigned)id, ud = (unsigned)d; q = u / ud; r = (int)(u % ud); }
  else { q = id / d; r = (int)(id % d); }
}

// col[m][(ky*ks+kx)*C + c] = act(x[b][oy*s - pad + ky][ox*s - pad + kx][c]), zero outside / beyond K.
// act = BN(scale, shift) + ReLU when bn_scale != NULL (the conv of vit.py:92-103 consumes relu(bn(prev))).
template <bool VEC>
__global__ __launch_bounds__(256) void im2col_kernel(const bf16* __restrict__ x, bf16* __restrict__ col, int B, int H, int W, int C,
                                                     int ks, int stride, int Kp, int Ho, int Wo,
                                                     const float* __restrict__ sc, const float* __restrict__ sh) {
  const int pad = ks / 2;
  const int K = ks * ks * C;
  if (VEC) {   // C % 8 == 0: one thread per 16-B chunk
    const int cpr = Kp / 8;
    int64_t total = (int64_t)B * Ho * Wo * cpr;
    const bool f32 = total < (1ll << 31);
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
      int ch, ox, oy;
      int64_t m, t, bb;
      divmod(id, cpr, f32, m, ch);
      divmod(m, Wo, f32, t, ox);
      divmod(t, Ho, f32, bb, oy);
      const int b = (int)bb;
      int k = ch * 8;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (k < K) {
        int tap = k / C, c = k % C, ky = tap / ks, kx = tap % ks;
        int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
          v = *reinterpret_cast<const u32x4*>(x + (((int64_t)b * H + iy) * W + ix) * C + c);
          if (sc) {
            bf16x8 t = *reinterpret_cast<bf16x8*>(&v);
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = f2bf(fmaxf(bf2f(t[e]) * sc[c + e] + sh[c + e], 0.f));
            v = *reinterpret_cast<u32x4*>(&t);
          }
        }
      }
      *reinterpret_cast<u32x4*>(col + m * Kp + k) = v;
    }
  } else {     // small C (1 or 3): one thread per 8 consecutive k of an output pixel (round 6: was one thread per ELEMENT with four 64-bit
               // divisions and a 2-byte store each: 37 us per launch for 19 MB of traffic)
    const int cpr = Kp / 8;
    const int64_t total = (int64_t)B * Ho * Wo * cpr;
    const bool f32 = total < (1ll << 31);
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
      int ch, ox, oy;
      int64_t m, t, bb;
      divmod(id, cpr, f32, m, ch);
      divmod(m, Wo, f32, t, ox);
      divmod(t, Ho, f32, bb, oy);
      const bf16* xb = x + (int64_t)bb * H * W * C;
      const int iy0 = oy * stride - pad, ix0 = ox * stride - pad;
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = ch * 8 + e;
        float v = 0.f;
        if (k < K) {
          const int tap = k / C, c = k - tap * C, ky = tap / ks, kx = tap - ky * ks;
          const int iy = iy0 + ky, ix = ix0 + kx;
          if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            v = bf2f(xb[((int64_t)iy * W + ix) * C + c]);
            if (sc) v = fmaxf(v * sc[c] + sh[c], 0.f);
          }
        }
        o[e] = f2bf(v);
      }
      *reinterpret_cast<bf16x8*>(col + m * Kp + ch * 8) = o;
    }
  }
}

// adjoint (gather form): dx[b][y][x][c] = sum over taps (ky,kx) with (y + pad - ky) % s == 0 of dcol[m][tap*C + c]
__global__ __launch_bounds__(256) void col2im_kernel(const bf16* __restrict__ dcol, bf16* __restrict__ dx, int B, int H, int W, int C,
                                                     int ks, int stride, int Kp, int Ho, int Wo) {
  const int pad = ks / 2;
  const int cpr = C / 8;
  int64_t total = (int64_t)B * H * W * cpr;
  const bool f32 = total < (1ll << 31);
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    int c, ix, iy;
    int64_t pix, t, bb;
    divmod(id, cpr, f32, pix, c);
    c *= 8;
    divmod(pix, W, f32, t, ix);
    divmod(t, H, f32, bb, iy);
    const int b = (int)bb;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int ky = 0; ky < ks; ++ky) {
      int ty = iy + pad - ky;
      if (ty < 0 || (ty % stride) != 0) continue;
      int oy = ty / stride;
      if (oy >= Ho) continue;
      for (int kx = 0; kx < ks; ++kx) {
        int tx = ix + pad - kx;
        if (tx < 0 || (tx % stride) != 0) continue;
        int ox = tx / stride;
        if (ox >= Wo) continue;
        int64_t m = ((int64_t)b * Ho + oy) * Wo + ox;
        bf16x8 t = *reinterpret_cast<const bf16x8*>(dcol + m * Kp + (ky * ks + kx) * C + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += bf2f(t[e]);
      }
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e]);
    *reinterpret_cast<bf16x8*>(dx + pix * C + c) = o;
  }
}

// ------------------------------------------------------------------------------------------------ BatchNorm
// column reduction helper: rows x C (C % 8 == 0), each thread owns one 8-channel chunk of a row strip.
// MODE 0: sums[c] += x, sums[C+c] += x*x.          (forward statistics)
// MODE 1: g = da * [scale*y+shift > 0]; sums[c] += g, sums[C+c] += g * xhat   (backward sums)
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const bf16* __restrict__ y, const bf16* __restrict__ da, int M, int C,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ sums) {
  extern __shared__ float red[];   // [2][C]
  const int tpr = C / 8;
  const int rpp = 256 / tpr;       // rows per pass
  for (int i = threadIdx.x; i < 2 * C; i += 256) red[i] = 0.f;
  __syncthreads();
  float a0[8], a1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
  const int r_in = threadIdx.x / tpr, cc = threadIdx.x % tpr;
  float mu[8], rs[8], gm[8], bt[8];
  if (MODE == 1 && r_in < rpp) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { mu[e] = mean[cc * 8 + e]; rs[e] = rstd[cc * 8 + e]; gm[e] = gamma[cc * 8 + e]; bt[e] = beta[cc * 8 + e]; }
  }
  if (r_in < rpp) {
#pragma unroll 4                      // keep 4 (MODE 1: 8) row loads in flight per thread
    for (int64_t r = (int64_t)blockIdx.x * rpp + r_in; r < M; r += (int64_t)gridDim.x * rpp) {
      bf16x8 t = *reinterpret_cast<const bf16x8*>(y + r * C + cc * 8);
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { float v = bf2f(t[e]); a0[e] += v; a1[e] += v * v; }
      } else {
        bf16x8 d = *reinterpret_cast<const bf16x8*>(da + r * C + cc * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float xh = (bf2f(t[e]) - mu[e]) * rs[e];
          float g = (xh * gm[e] + bt[e] > 0.f) ? bf2f(d[e]) : 0.f;
          a0[e] += g; a1[e] += g * xh;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { atomicAdd(&red[cc * 8 + e], a0[e]); atomicAdd(&red[C + cc * 8 + e], a1[e]); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) atomicAdd(&sums[i], red[i]);
}

__global__ void bn_finalize_kernel(const float* __restrict__ sums, int M, int C, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, float momentum, float eps, int training,
                                   float* mean, float* rstd, float* scale, float* shift) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mu, var;
  if (training) {
    mu = sums[c] / (float)M;
    var = fmaxf(sums[C + c] / (float)M - mu * mu, 0.f);
    float unb = M > 1 ? var * (float)M / (float)(M - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
  } else {
    mu = running_mean[c];
    var = running_var[c];
  }
  float r = rsqrtf(var + eps);
  mean[c] = mu; rstd[c] = r;
  float s = gamma[c] * r;
  scale[c] = s; shift[c] = beta[c] - mu * s;
}

// ---- grouped BatchNorm passes: the same-index layers of several expert stems in ONE launch ---------------------------------
// forward: a = relu(bn(y)).  The per-channel sums come from the conv GEMM's epilogue (ph_gemm_args.col_stats); every block
// re-derives mean / rstd / scale / shift of its item into LDS (2C loads + C rsqrt: nothing next to the streaming pass); block 0
// of an item also writes them out for the backward and updates the running statistics (train mode).
struct BnGroup {
  int n;
  int blk_start[PH_BN_GROUP_MAX + 1];
  ph_bn_item it[PH_BN_GROUP_MAX];
};
__global__ __launch_bounds__(256) void bn_apply_grouped_kernel(BnGroup g, float momentum, float eps, int training) {
  extern __shared__ float lds[];                    // [2][C]: scale, shift
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.blk_start[i + 1]) ++i;
  const ph_bn_item& t = g.it[i];
  const int lb = (int)blockIdx.x - g.blk_start[i], nb = g.blk_start[i + 1] - g.blk_start[i];
  const int C = t.C;
  const double* sums = (const double*)t.sums;
  for (int c = threadIdx.x; c < C; c += 256) {
    float mu, var;
    if (training) {
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int sl = 0; sl < PH_COLSTAT_SLABS; ++sl) { s1 += sums[(size_t)sl * 2 * C + c]; s2 += sums[(size_t)sl * 2 * C + C + c]; }
      const double m1 = s1 / (double)t.M, m2 = s2 / (double)t.M;
      mu = (float)m1;
      var = (float)fmax(m2 - m1 * m1, 0.0);
    } else {
      mu = t.running_mean[c]; var = t.running_var[c];
    }
    const float r = rsqrtf(var + eps), sc = t.gamma[c] * r;
    lds[c] = sc; lds[C + c] = t.beta[c] - mu * sc;
    if (lb == 0) {
      t.stats[c] = mu; t.stats[C + c] = r; t.stats[2 * C + c] = sc; t.stats[3 * C + c] = t.beta[c] - mu * sc;
      if (training) {
        const float unb = t.M > 1 ? var * (float)t.M / (float)(t.M - 1) : var;
        t.running_mean[c] = (1.f - momentum) * t.running_mean[c] + momentum * mu;
        t.running_var[c] = (1.f - momentum) * t.running_var[c] + momentum * unb;
      }
    }
  }
  __syncthreads();
  // streaming pass (round 6): a thread keeps ONE group of 8 channels (its scale / shift live in registers) and walks the rows; the per-element
  // LDS reads of the first version were 16 of the 18 memory instructions of an iteration
  const int tpr = C / 8, rpp = 256 / tpr;
  const int r_in = threadIdx.x / tpr, cc = threadIdx.x % tpr;
  if (r_in >= rpp) return;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = lds[cc * 8 + e]; sh[e] = lds[C + cc * 8 + e]; }
  const bf16* y = (const bf16*)t.y;
  bf16* a = (bf16*)t.a;
#pragma unroll 4
  for (int64_t r = (int64_t)lb * rpp + r_in; r < t.M; r += (int64_t)nb * rpp) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(y + r * C + cc * 8);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(fmaxf(bf2f(v[e]) * sc[e] + sh[e], 0.f));
    *reinterpret_cast<bf16x8*>(a + r * C + cc * 8) = o;
  }
}

// backward, pass 1 (grouped bn_reduce<1>): sums[c] += g, sums[C+c] += g*xhat with g = da * [a > 0]
__global__ __launch_bounds__(256) void bn_bwd_reduce_grouped_kernel(BnGroup g) {
  extern __shared__ float red[];                    // [2][C]
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.blk_start[i + 1]) ++i;
  const ph_bn_item& t = g.it[i];
  const int lb = (int)blockIdx.x - g.blk_start[i], nb = g.blk_start[i + 1] - g.blk_start[i];
  const int C = t.C, tpr = C / 8, rpp = 256 / tpr;
  for (int k = threadIdx.x; k < 2 * C; k += 256) red[k] = 0.f;
  __syncthreads();
  const int r_in = threadIdx.x / tpr, cc = threadIdx.x % tpr;
  if (r_in < rpp) {
    float a0[8], a1[8], mu[8], rs[8], sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a0[e] = 0.f; a1[e] = 0.f;
      mu[e] = t.stats[cc * 8 + e]; rs[e] = t.stats[C + cc * 8 + e]; sc[e] = t.stats[2 * C + cc * 8 + e]; sh[e] = t.stats[3 * C + cc * 8 + e];
    }
    const bf16* y = (const bf16*)t.y;
    const bf16* da = (const bf16*)t.a;              // (the `a` slot of the item carries dA in the backward)
#ifndef PH_BN_REDUCE_UNROLL
#define PH_BN_REDUCE_UNROLL 8          // (round 6 A/B, profiles/r6_ab_batchnorm_passes.txt: 4 -> 8 rows in flight 79.0 -> 74.0 us per launch)
#endif
#pragma unroll PH_BN_REDUCE_UNROLL
    for (int64_t r = (int64_t)lb * rpp + r_in; r < t.M; r += (int64_t)nb * rpp) {
      bf16x8 v = *reinterpret_cast<const bf16x8*>(y + r * C + cc * 8);
      bf16x8 d = *reinterpret_cast<const bf16x8*>(da + r * C + cc * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float yv = bf2f(v[e]);
        const float gg = (yv * sc[e] + sh[e] > 0.f) ? bf2f(d[e]) : 0.f;
        a0[e] += gg; a1[e] += gg * (yv - mu[e]) * rs[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { atomicAdd(&red[cc * 8 + e], a0[e]); atomicAdd(&red[C + cc * 8 + e], a1[e]); }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 2 * C; k += 256) atomicAdd((float*)t.sums + k, red[k]);
}
// backward, pass 2: dy = gamma*rstd*(g - sum_g/M - xhat*sum_gx/M); block 0 of an item adds dgamma / dbeta
__global__ __launch_bounds__(256) void bn_bwd_apply_grouped_kernel(BnGroup g) {
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.blk_start[i + 1]) ++i;
  const ph_bn_item& t = g.it[i];
  const int lb = (int)blockIdx.x - g.blk_start[i], nb = g.blk_start[i + 1] - g.blk_start[i];
  const int C = t.C;
  const float invM = 1.f / (float)t.M;
  const bf16* y = (const bf16*)t.y;
  const bf16* da = (const bf16*)t.a;
  bf16* dy = (bf16*)t.dy;
  const float* bsums = (const float*)t.sums;
  // round 6: a thread keeps ONE group of 8 channels and walks the rows, so the per-channel terms are loop invariant and live in registers
  // (the first version re-loaded 48 statistics per 8 elements).  dy = scale*(g - S1/M - xhat*S2/M) with xhat = (y - mean)*rstd, regrouped as
  // dy = scale*g + A*y + B,  A = -scale*rstd*S2/M,  B = -scale*S1/M - A*mean       (g = dA where relu(bn(y)) > 0, else 0)
  const int tpr = C / 8, rpp = 256 / tpr;
  const int r_in = threadIdx.x / tpr, cc = threadIdx.x % tpr;
  if (r_in < rpp) {
    float sc[8], sh[8], ca[8], cb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = cc * 8 + e;
      const float mu = t.stats[c], rs = t.stats[C + c];
      sc[e] = t.stats[2 * C + c]; sh[e] = t.stats[3 * C + c];
      ca[e] = -sc[e] * rs * bsums[C + c] * invM;
      cb[e] = -sc[e] * bsums[c] * invM - ca[e] * mu;
    }
#pragma unroll 4
    for (int64_t r = (int64_t)lb * rpp + r_in; r < t.M; r += (int64_t)nb * rpp) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(y + r * C + cc * 8);
      const bf16x8 d = *reinterpret_cast<const bf16x8*>(da + r * C + cc * 8);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float yv = bf2f(v[e]);
        const float gg = (yv * sc[e] + sh[e] > 0.f) ? bf2f(d[e]) : 0.f;
        o[e] = f2bf(sc[e] * gg + (ca[e] * yv + cb[e]));
      }
      *reinterpret_cast<bf16x8*>(dy + r * C + cc * 8) = o;
    }
  }
  if (lb == 0) {
    for (int c = threadIdx.x; c < C; c += 256) {
      if (t.dbeta) atomicAdd(t.dbeta + c, bsums[c]);
      if (t.dgamma) atomicAdd(t.dgamma + c, bsums[C + c]);
    }
  }
}

// dy = gamma*rstd*(g - dbeta/M - xhat*dgamma/M); block 0 also accumulates dgamma/dbeta into the parameter grads.
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16* __restrict__ da, const bf16* __restrict__ y, bf16* __restrict__ dy,
                                                           int64_t M, int C, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ sums,
                                                           float* dgamma, float* dbeta) {
  const int cpr = C / 8;
  int64_t total = M * cpr;
  float invM = 1.f / (float)M;
  const bool f32 = total < (1ll << 31);
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    int c;
    int64_t r;
    divmod(id, cpr, f32, r, c);
    c *= 8;
    bf16x8 t = *reinterpret_cast<const bf16x8*>(y + r * C + c);
    bf16x8 d = *reinterpret_cast<const bf16x8*>(da + r * C + c);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float xh = (bf2f(t[e]) - mean[c + e]) * rstd[c + e];
      float g = (xh * gamma[c + e] + beta[c + e] > 0.f) ? bf2f(d[e]) : 0.f;
      o[e] = f2bf(gamma[c + e] * rstd[c + e] * (g - sums[c + e] * invM - xh * sums[C + c + e] * invM));
    }
    *reinterpret_cast<bf16x8*>(dy + r * C + c) = o;
  }
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += 256) {
      if (dbeta) atomicAdd(dbeta + c, sums[c]);
      if (dgamma) atomicAdd(dgamma + c, sums[C + c]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ tokens
__device__ __forceinline__ int nearest_src(int dst, int in, int out) {   // F.interpolate(mode='nearest'), vit.py:142
  float scale = (float)in / (float)out;
  return min((int)floorf((float)dst * scale), in - 1);
}

__global__ __launch_bounds__(256) void tokens_kernel(const bf16* __restrict__ feat, const float* __restrict__ pos, bf16* __restrict__ tokens,
                                                     int B, int G, int D, int tpb, int off, const int64_t* __restrict__ inst, int E,
                                                     int g, const int32_t* __restrict__ table, const float* __restrict__ inst_emb) {
  const int cpr = D / 4;
  int64_t total = (int64_t)B * G * cpr;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    int c = (int)(id % cpr) * 4;
    int64_t bt = id / cpr;
    int t = (int)(bt % G), b = (int)(bt / G);
    bf16x4 f = *reinterpret_cast<const bf16x4*>(feat + bt * D + c);
    f32x4 p = *reinterpret_cast<const f32x4*>(pos + (int64_t)t * D + c);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = bf2f(f[e]) + p[e];
    if (inst) {
      int ty = t / g, tx = t % g;
      int64_t lab = inst[((int64_t)b * E + nearest_src(ty, E, g)) * E + nearest_src(tx, E, g)];
      int row = table[(int)(lab & 255)];
      f32x4 ie = *reinterpret_cast<const f32x4*>(inst_emb + (int64_t)row * D + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += ie[e];
    }
    bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
    *reinterpret_cast<bf16x4*>(tokens + ((int64_t)b * tpb + off + t) * D + c) = o;
  }
}

// one thread per (t, 4 channels) and batch slice (blockIdx.y): the slice is walked 8 images at a time with all 8 loads
// issued before the first store (a load -> store loop over the batch is a 2-3 us latency chain per image: 95 us per call);
// dpos / dinst_emb via fp32 atomics (gridDim.y partial sums per position).
__global__ __launch_bounds__(256) void tokens_bwd_kernel(const bf16* __restrict__ dtok, bf16* __restrict__ dfeat, float* __restrict__ dpos,
                                                         int B, int G, int D, int tpb, int off, const int64_t* __restrict__ inst,
                                                         int E, int g, const int32_t* __restrict__ table, float* __restrict__ dinst) {
  const int cpr = D / 4;
  int total = G * cpr;
  int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= total) return;
  int c = (id % cpr) * 4, t = id / cpr;
  const int per = (B + gridDim.y - 1) / gridDim.y;
  const int b_lo = blockIdx.y * per, b_hi = min(B, b_lo + per);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int src_y = 0, src_x = 0;
  if (inst && dinst) { src_y = nearest_src(t / g, E, g); src_x = nearest_src(t % g, E, g); }
  for (int b0 = b_lo; b0 < b_hi; b0 += 8) {
    bf16x4 d[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int b = min(b0 + j, b_hi - 1);
      d[j] = *reinterpret_cast<const bf16x4*>(dtok + ((int64_t)b * tpb + off + t) * D + c);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int b = b0 + j;
      if (b < b_hi) {
        if (dfeat) *reinterpret_cast<bf16x4*>(dfeat + ((int64_t)b * G + t) * D + c) = d[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += bf2f(d[j][e]);
        if (inst && dinst) {
          int64_t lab = inst[((int64_t)b * E + src_y) * E + src_x];
          int row = table[(int)(lab & 255)];
#pragma unroll
          for (int e = 0; e < 4; ++e) atomicAdd(dinst + (int64_t)row * D + c + e, bf2f(d[j][e]));
        }
      }
    }
  }
  if (dpos) {
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(dpos + (int64_t)t * D + c + e, acc[e]);
  }
}

// dinst[row(b,t)][c] += dtok[b, off+t, c]: a segmented sum over the B*G tokens keyed by <= 128 instance rows (typically ~10
// distinct ones).  One global atomic per (b,t,c) -- 4.8 M atomics on a few hundred addresses -- took 165 us; here a block owns
// 64 channels x a slice of the tokens, accumulates in LDS (ds_add_f32) and flushes only the rows it touched.
__global__ __launch_bounds__(256) void tokens_inst_bwd_kernel(const bf16* __restrict__ dtok, int B, int G, int D, int tpb, int off,
                                                              const int64_t* __restrict__ inst, int E, int g,
                                                              const int32_t* __restrict__ table, float* __restrict__ dinst, int per_block) {
  __shared__ float acc[128 * 64];
  __shared__ int touched[128];
  for (int i = threadIdx.x; i < 128 * 64; i += 256) acc[i] = 0.f;
  if (threadIdx.x < 128) touched[threadIdx.x] = 0;
  __syncthreads();
  const int c4 = threadIdx.x & 15, lane_item = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + c4 * 4;
  const int total = B * G;
  const int i0 = blockIdx.y * per_block, i1 = min(total, i0 + per_block);
  if (c < D) {
#pragma unroll 4
    for (int it = i0 + lane_item; it < i1; it += 16) {
      const int b = it / G, t = it % G;
      const int64_t lab = inst[((int64_t)b * E + nearest_src(t / g, E, g)) * E + nearest_src(t % g, E, g)];
      const int row = table[(int)(lab & 255)] & 127;
      const bf16x4 d = *reinterpret_cast<const bf16x4*>(dtok + ((int64_t)b * tpb + off + t) * D + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(&acc[row * 64 + c4 * 4 + e], bf2f(d[e]));
      touched[row] = 1;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 128 * 64; i += 256) {
    const int row = i >> 6, cc = blockIdx.x * 64 + (i & 63);
    if (touched[row] && cc < D) atomicAdd(dinst + (int64_t)row * D + cc, acc[i]);
  }
}

// ------------------------------------------------------------------------------------------------ taps
__global__ void gather_taps_kernel(const float* __restrict__ in, float* __restrict__ out, const int32_t* __restrict__ idx,
                                   const float* __restrict__ w, int n_out, int taps, int D) {
  int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (int64_t)n_out * D) return;
  int d = (int)(id % D), i = (int)(id / D);
  float s = 0.f;
  for (int t = 0; t < taps; ++t) s += w[i * taps + t] * in[(int64_t)idx[i * taps + t] * D + d];
  out[id] = s;
}
__global__ void scatter_taps_kernel(const float* __restrict__ dout, float* __restrict__ din, const int32_t* __restrict__ idx,
                                    const float* __restrict__ w, int n_out, int taps, int D) {
  int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (int64_t)n_out * D) return;
  int d = (int)(id % D), i = (int)(id / D);
  float g = dout[id];
  for (int t = 0; t < taps; ++t) atomicAdd(din + (int64_t)idx[i * taps + t] * D + d, w[i * taps + t] * g);
}

inline int grid_for(int64_t work_items) { return (int)std::min<int64_t>(ceil_div64(work_items, 256), 256 * 16); }

}  // namespace

extern "C" int ph_patchify(const float* img, void* col, int B, int C, int R, int p, int Kp, hipStream_t stream) {
  PH_CHECK_ARG(img && col && B > 0 && R >= p && Kp >= C * p * p && Kp % 8 == 0, "ph_patchify: bad args");   // g = floor(R/p) like Conv2d(stride=p)
  ProfScope prof__(PH_FAM_FRONTEND, 0.0, 0.0, stream, "ph_patchify");
  int g = R / p;
  hipLaunchKernelGGL(patchify_kernel, dim3(grid_for((int64_t)B * g * g * (Kp / 2))), dim3(256), 0, stream, img, (bf16*)col, B, C, R, p, Kp);
  PH_LAUNCH_CHECK("patchify_kernel");
  return PH_OK;
}

extern "C" int ph_resize_bilinear_nchw_to_nhwc(const float* x, void* y, int B, int C, int Hin, int Win, int Hout, int Wout,
                                               hipStream_t stream) {
  PH_CHECK_ARG(x && y && B > 0 && C > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "ph_resize_bilinear: bad args");
  ProfScope prof__(PH_FAM_FRONTEND, 0.0, 0.0, stream, "ph_resize_bilinear_nchw_to_nhwc");
  int64_t blocks = (int64_t)B * Hout * ((C + 7) / 8);
  PH_CHECK_ARG(blocks < (1ll << 31), "ph_resize_bilinear: grid too large");
  hipLaunchKernelGGL(resize_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, x, (bf16*)y, B, C, Hin, Win, Hout, Wout, (const float*)nullptr, 0);
  PH_LAUNCH_CHECK("resize_kernel");
  return PH_OK;
}

extern "C" int ph_dense_minmax_partial(const float* x, float* part, int B, int64_t n_per_sample, int nparts, hipStream_t stream) {
  PH_CHECK_ARG(x && part && B > 0 && n_per_sample > 0 && nparts > 0 && nparts <= 64, "ph_dense_minmax_partial: bad args");
  ProfScope prof__(PH_FAM_FRONTEND, 0.0, 0.0, stream, "ph_dense_minmax_partial");
  hipLaunchKernelGGL(dense_minmax_partial_kernel, dim3(nparts, B), dim3(256), 0, stream, x, part, n_per_sample, nparts);
  PH_LAUNCH_CHECK("dense_minmax_partial_kernel");
  return PH_OK;
}

extern "C" int ph_resize_remap_nchw_to_nhwc(const float* x, const float* minmax_part, int nparts, void* y, int B, int C, int Hin, int Win, int Hout,
                                            int Wout, hipStream_t stream) {
  PH_CHECK_ARG(x && y && minmax_part && nparts > 0 && nparts <= 64 && B > 0 && C > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "ph_resize_remap: bad args");
  ProfScope prof__(PH_FAM_FRONTEND, 0.0, 0.0, stream, "ph_resize_remap_nchw_to_nhwc");
  int64_t blocks = (int64_t)B * Hout * ((C + 7) / 8);
  PH_CHECK_ARG(blocks < (1ll << 31), "ph_resize_remap: grid too large");
  hipLaunchKernelGGL(resize_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, x, (bf16*)y, B, C, Hin, Win, Hout, Wout, minmax_part, nparts);
  PH_LAUNCH_CHECK("resize_kernel");
  return PH_OK;
}

extern "C" int ph_inpaint_resize_nhwc(const uint8_t* labels, const float* table, int64_t table_batch_stride, void* y, int B, int C, int Hin,
                                      int Win, int Hout, int Wout, hipStream_t stream) {
  PH_CHECK_ARG(labels && table && y && B > 0 && C > 0 && (C % 8) == 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && table_batch_stride >= 0,
               "ph_inpaint_resize_nhwc: bad args");
  PH_CHECK_ARG((((uintptr_t)table | (uintptr_t)y) & 15) == 0 && (table_batch_stride % 4) == 0, "ph_inpaint_resize_nhwc: table / output must be 16-B aligned");
  ProfScope prof__(PH_FAM_FRONTEND, 0.0, 0.0, stream, "ph_inpaint_resize_nhwc");
  hipLaunchKernelGGL(inpaint_resize_kernel, dim3(grid_for((int64_t)B * Hout * Wout * (C / 8))), dim3(256), 0, stream, labels, table,
                     table_batch_stride, (bf16*)y, B, C, Hin, Win, Hout, Wout);
  PH_LAUNCH_CHECK("inpaint_resize_kernel");
  return PH_OK;
}

extern "C" int ph_im2col_nhwc(const void* x, void* col, int B, int H, int W, int C, int ksize, int stride, int Kp,
                              const float* bn_scale, const float* bn_shift, hipStream_t stream) {
  PH_CHECK_ARG(x && col && (ksize == 1 || ksize == 3) && stride >= 1 && Kp >= ksize * ksize * C && Kp % 8 == 0, "ph_im2col_nhwc: bad args");
  ProfScope prof__(PH_FAM_FRONTEND, 0.0, 0.0, stream, "ph_im2col_nhwc");
  PH_CHECK_ARG((bn_scale == nullptr) == (bn_shift == nullptr), "ph_im2col_nhwc: scale/shift must come together");
  int pad = ksize / 2;
  int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
  if (C % 8 == 0) {
    hipLaunchKernelGGL(im2col_kernel<true>, dim3(grid_for((int64_t)B * Ho * Wo * (Kp / 8))), dim3(256), 0, stream, (const bf16*)x,
                       (bf16*)col, B, H, W, C, ksize, stride, Kp, Ho, Wo, bn_scale, bn_shift);
  } else {
    hipLaunchKernelGGL(im2col_kernel<false>, dim3(grid_for((int64_t)B * Ho * Wo * (Kp / 8))), dim3(256), 0, stream, (const bf16*)x,
                       (bf16*)col, B, H, W, C, ksize, stride, Kp, Ho, Wo, bn_scale, bn_shift);
  }
  PH_LAUNCH_CHECK("im2col_kernel");
  return PH_OK;
}

extern "C" int ph_col2im_nhwc(const void* dcol, void* dx, int B, int H, int W, int C, int ksize, int stride, int Kp,
                              hipStream_t stream) {
  PH_CHECK_ARG(dcol && dx && (ksize == 1 || ksize == 3) && C % 8 == 0 && Kp >= ksize * ksize * C, "ph_col2im_nhwc: bad args");
  ProfScope prof__(PH_FAM_FRONTEND, 0.0, 0.0, stream, "ph_col2im_nhwc");
  int pad = ksize / 2;
  int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
  hipLaunchKernelGGL(col2im_kernel, dim3(grid_for((int64_t)B * H * W * (C / 8))), dim3(256), 0, stream, (const bf16*)dcol, (bf16*)dx, B,
                     H, W, C, ksize, stride, Kp, Ho, Wo);
  PH_LAUNCH_CHECK("col2im_kernel");
  return PH_OK;
}

// Every bn_reduce block ends with 2*C fp32 global atomics (~35/ns chip-wide): with 1024 blocks they cost more than the loads
// (A/B: 100352x192 stats 40.8 us at 1024 blocks, 21.8 us at 256).  Budget ~100k atomics per launch.
// (a kernel, not hipMemsetAsync: a small memset NODE of a captured hipGraph does not replay correctly on ROCm 7.0 -- tools/graph_memset_probe.py)
__global__ void zero_floats_kernel(float* __restrict__ p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.f;
}
static void zero_floats(float* p, int n, hipStream_t stream) {
  hipLaunchKernelGGL(zero_floats_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, p, n);
}
#ifndef PH_BN_REDUCE_BUDGET
#define PH_BN_REDUCE_BUDGET 49152      // blocks per item = budget / C: every block ends in 2C global float atomics (~40 G/s chip-wide); 4 x the blocks: 79 -> 134 us per launch
#endif
static int bn_reduce_blocks(int C) { return std::max(64, PH_BN_REDUCE_BUDGET / C); }

extern "C" int ph_bn_stats(const void* y, int M, int C, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float momentum, float eps, int training, float* mean, float* rstd, float* scale,
                           float* shift, int prezeroed, hipStream_t stream) {
  PH_CHECK_ARG(gamma && beta && running_mean && running_var && mean && rstd && scale && shift, "ph_bn_stats: null pointer");
  ProfScope prof__(PH_FAM_FRONTEND, 0.0, 0.0, stream, "ph_bn_stats");
  PH_CHECK_ARG(C % 8 == 0 && C <= 2048 && M > 0, "ph_bn_stats: C=%d unsupported", C);
  float* sums = scale;   // scale/shift double as the [2*C] reduction scratch when they are contiguous; otherwise use mean/rstd
  // we need 2*C contiguous floats: require shift == scale + C (the host allocates the four vectors as one block)
  PH_CHECK_ARG(shift == scale + C, "ph_bn_stats: scale/shift must be one contiguous [2*C] block");
  if (training) {
    PH_CHECK_ARG(y, "ph_bn_stats: null y");
    if (!prezeroed) zero_floats(sums, 2 * C, stream);
    int rpp = 256 / (C / 8);
    int grid = (int)std::min<int64_t>(ceil_div64(M, (int64_t)rpp * 8), bn_reduce_blocks(C));
    hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3(grid), dim3(256), sizeof(float) * 2 * C, stream, (const bf16*)y, (const bf16*)nullptr, M,
                       C, nullptr, nullptr, nullptr, nullptr, sums);
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, stream, sums, M, C, gamma, beta, running_mean, running_var,
                     momentum, eps, training, mean, rstd, scale, shift);
  PH_LAUNCH_CHECK("bn_stats kernels");
  return PH_OK;
}

extern "C" int ph_bn_relu_bwd(const void* da, const void* y, void* dy, int M, int C, const float* gamma, const float* beta,
                              const float* mean, const float* rstd, float* dgamma, float* dbeta, float* sums, int prezeroed,
                              hipStream_t stream) {
  PH_CHECK_ARG(da && y && dy && gamma && beta && mean && rstd && sums, "ph_bn_relu_bwd: null pointer");
  ProfScope prof__(PH_FAM_FRONTEND, 0.0, 0.0, stream, "ph_bn_relu_bwd");
  PH_CHECK_ARG(C % 8 == 0 && C <= 2048 && M > 0, "ph_bn_relu_bwd: C=%d unsupported", C);
  if (!prezeroed) zero_floats(sums, 2 * C, stream);
  int rpp = 256 / (C / 8);
  int grid = (int)std::min<int64_t>(ceil_div64(M, (int64_t)rpp * 8), bn_reduce_blocks(C));
  hipLaunchKernelGGL(bn_reduce_kernel<1>, dim3(grid), dim3(256), sizeof(float) * 2 * C, stream, (const bf16*)y, (const bf16*)da, M, C, mean,
                     rstd, gamma, beta, sums);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for((int64_t)M * (C / 8))), dim3(256), 0, stream, (const bf16*)da, (const bf16*)y,
                     (bf16*)dy, (int64_t)M, C, gamma, beta, mean, rstd, sums, dgamma, dbeta);
  PH_LAUNCH_CHECK("bn_relu_bwd kernels");
  return PH_OK;
}

static int bn_group_fill(const ph_bn_item* items, int n, BnGroup& g, int* max_c, const char* who, bool reduce_grid) {
  PH_CHECK_ARG(items && n >= 1 && n <= PH_BN_GROUP_MAX, "%s: need 1..%d items, got %d", who, PH_BN_GROUP_MAX, n);
  g.n = n;
  int total = 0, mc = 0;
  for (int i = 0; i < n; ++i) {
    const ph_bn_item& t = items[i];
    PH_CHECK_ARG(t.y && t.a && t.stats && t.sums && t.gamma && t.beta && t.M > 0 && t.C > 0 && t.C % 8 == 0 && t.C <= 2048, "%s: bad item %d", who, i);
    g.it[i] = t;
    g.blk_start[i] = total;
    int blocks;
    if (reduce_grid) {
      const int rpp = 256 / (t.C / 8);
      blocks = (int)std::min<int64_t>(ceil_div64(t.M, (int64_t)rpp * 8), bn_reduce_blocks(t.C));
    } else {
      blocks = (int)std::min<int64_t>(ceil_div64(t.M, (int64_t)(256 / (t.C / 8)) * 4), 2048);     // streaming passes: rows strided, 4+ rows per thread
    }
    total += std::max(blocks, 1);
    mc = std::max(mc, t.C);
  }
  g.blk_start[n] = total;
  *max_c = mc;
  return PH_OK;
}

extern "C" int ph_bn_apply_relu_grouped(const ph_bn_item* items, int n, float momentum, float eps, int training, hipStream_t stream) {
  ProfScope prof__(PH_FAM_FRONTEND, 0.0, 0.0, stream, "ph_bn_apply_relu_grouped");
  BnGroup g; int mc;
  int rc = bn_group_fill(items, n, g, &mc, "ph_bn_apply_relu_grouped", false);
  if (rc) return rc;
  for (int i = 0; i < n; ++i) PH_CHECK_ARG(items[i].running_mean && items[i].running_var, "ph_bn_apply_relu_grouped: running stats missing");
  hipLaunchKernelGGL(bn_apply_grouped_kernel, dim3(g.blk_start[n]), dim3(256), sizeof(float) * 2 * mc, stream, g, momentum, eps, training);
  PH_LAUNCH_CHECK("bn_apply_grouped_kernel");
  return PH_OK;
}

extern "C" int ph_bn_relu_bwd_grouped(const ph_bn_item* items, int n, hipStream_t stream) {
  ProfScope prof__(PH_FAM_FRONTEND, 0.0, 0.0, stream, "ph_bn_relu_bwd_grouped");
  BnGroup g; int mc;
  int rc = bn_group_fill(items, n, g, &mc, "ph_bn_relu_bwd_grouped", true);
  if (rc) return rc;
  for (int i = 0; i < n; ++i) PH_CHECK_ARG(items[i].dy, "ph_bn_relu_bwd_grouped: dy missing");
  hipLaunchKernelGGL(bn_bwd_reduce_grouped_kernel, dim3(g.blk_start[n]), dim3(256), sizeof(float) * 2 * mc, stream, g);
  rc = bn_group_fill(items, n, g, &mc, "ph_bn_relu_bwd_grouped", false);
  if (rc) return rc;
  hipLaunchKernelGGL(bn_bwd_apply_grouped_kernel, dim3(g.blk_start[n]), dim3(256), 0, stream, g);
  PH_LAUNCH_CHECK("bn_relu_bwd_grouped kernels");
  return PH_OK;
}

extern "C" int ph_tokens_finalize(const void* feat, const float* pos, void* tokens, int B, int G, int D, int tok_per_batch,
                                  int tok_off, const int64_t* inst, int E, int g, const int32_t* table, const float* inst_emb,
                                  hipStream_t stream) {
  PH_CHECK_ARG(feat && pos && tokens && D % 4 == 0 && (!inst || g * g == G), "ph_tokens_finalize: bad args");
  ProfScope prof__(PH_FAM_FRONTEND, 0.0, 0.0, stream, "ph_tokens_finalize");
  PH_CHECK_ARG(!inst || (table && inst_emb && E > 0), "ph_tokens_finalize: instance inputs incomplete");
  hipLaunchKernelGGL(tokens_kernel, dim3(grid_for((int64_t)B * G * (D / 4))), dim3(256), 0, stream, (const bf16*)feat, pos, (bf16*)tokens, B,
                     G, D, tok_per_batch, tok_off, inst, E, g, table, inst_emb);
  PH_LAUNCH_CHECK("tokens_kernel");
  return PH_OK;
}

extern "C" int ph_tokens_finalize_bwd(const void* dtokens, void* dfeat, float* dpos, int B, int G, int D, int tok_per_batch,
                                      int tok_off, const int64_t* inst, int E, int g, const int32_t* table, float* dinst_emb,
                                      hipStream_t stream) {
  PH_CHECK_ARG(dtokens && D % 4 == 0 && (!inst || g * g == G), "ph_tokens_finalize_bwd: bad args");
  ProfScope prof__(PH_FAM_FRONTEND, 0.0, 0.0, stream, "ph_tokens_finalize_bwd");
  const bool seg = inst && dinst_emb && table;      // instance-embedding gradient: segmented LDS reduction instead of per-element atomics
  hipLaunchKernelGGL(tokens_bwd_kernel, dim3(ceil_div(G * (D / 4), 256), B >= 16 ? 4 : 1), dim3(256), 0, stream, (const bf16*)dtokens, (bf16*)dfeat, dpos, B, G,
                     D, tok_per_batch, tok_off, seg ? nullptr : inst, E, g, table, seg ? nullptr : dinst_emb);
  if (seg) {
    const int total = B * G, slices = std::max(1, std::min(32, total / 256)), per_block = ceil_div(total, slices);
    hipLaunchKernelGGL(tokens_inst_bwd_kernel, dim3(ceil_div(D, 64), slices), dim3(256), 0, stream, (const bf16*)dtokens, B, G, D, tok_per_batch,
                       tok_off, inst, E, g, table, dinst_emb, per_block);
  }
  PH_LAUNCH_CHECK("tokens_bwd_kernel");
  return PH_OK;
}

extern "C" int ph_gather_taps(const float* in, float* out, const int32_t* idx, const float* w, int n_out, int taps, int D,
                              hipStream_t stream) {
  PH_CHECK_ARG(in && out && idx && w, "ph_gather_taps: null pointer");
  ProfScope prof__(PH_FAM_FRONTEND, 0.0, 0.0, stream, "ph_gather_taps");
  hipLaunchKernelGGL(gather_taps_kernel, dim3((unsigned)ceil_div64((int64_t)n_out * D, 256)), dim3(256), 0, stream, in, out, idx, w, n_out, taps, D);
  PH_LAUNCH_CHECK("gather_taps_kernel");
  return PH_OK;
}
extern "C" int ph_scatter_taps(const float* dout, float* din, const int32_t* idx, const float* w, int n_out, int taps, int D,
                               hipStream_t stream) {
  PH_CHECK_ARG(dout && din && idx && w, "ph_scatter_taps: null pointer");
  ProfScope prof__(PH_FAM_FRONTEND, 0.0, 0.0, stream, "ph_scatter_taps");
  hipLaunchKernelGGL(scatter_taps_kernel, dim3((unsigned)ceil_div64((int64_t)n_out * D, 256)), dim3(256), 0, stream, dout, din, idx, w, n_out, taps, D);
  PH_LAUNCH_CHECK("scatter_taps_kernel");
  return PH_OK;
}
