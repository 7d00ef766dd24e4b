// LayerNorm forward / backward (fp32 math on bf16 rows) for gfx950.
// Replaces model/modules/utils.py:14-19 (fp32 F.layer_norm + casts) and its autograd.
// One 64-lane wave owns one row: the row lives in registers (8-B bf16x4 loads, D/4 chunks spread over the
// lanes), mean / variance are two wave-shuffle reductions, nothing goes through LDS.  HBM-bound:
// algorithmic bytes per row = 2*D (read) + 2*D (write) forward.
#include "common.h"

namespace {

constexpr int MAX_CH = 8;   // chunks of 4 elements per lane -> D <= 2048

__device__ __forceinline__ int map_row(const ph_rowmap& m, int r) {
  return m.seg_in ? (r / m.seg_in) * m.seg_out + m.seg_off + (r % m.seg_in) : r;
}

template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(ph_layernorm_fwd_args a) {
  int lane = threadIdx.x & 63;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.M) return;
  PH_TL_DECL;
  PH_TL(0);
  const int nch = a.D >> 2;
  const bf16* x = reinterpret_cast<const bf16*>(a.x) + (size_t)row * a.D;
  const float* xf = reinterpret_cast<const float*>(a.x) + (size_t)row * a.D;
  float v[NCH][4];
  float s = 0.f;
  f32x4 gam[NCH], bet[NCH];            // requested with the row: their latency hides behind the two reductions
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = min(lane + 64 * i, nch - 1);
    gam[i] = *reinterpret_cast<const f32x4*>(a.gamma + c * 4);
    bet[i] = *reinterpret_cast<const f32x4*>(a.beta + c * 4);
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = lane + 64 * i;
    if (c < nch) {
      if (a.x_f32) {
        f32x4 t = *reinterpret_cast<const f32x4*>(xf + c * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[i][e] = t[e]; s += v[i][e]; }
      } else {
        bf16x4 t = *reinterpret_cast<const bf16x4*>(x + c * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[i][e] = bf2f(t[e]); s += v[i][e]; }
      }
    }
  }
  PH_TL(1);
  float mean = wave_sum(s) / (float)a.D;
  PH_TL(2);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = lane + 64 * i;
    if (c < nch) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { float d = v[i][e] - mean; q += d * d; }
    }
  }
  float rstd = rsqrtf(wave_sum(q) / (float)a.D + a.eps);
  PH_TL(3);
  if (lane == 0) {
    if (a.mean) a.mean[row] = mean;
    if (a.rstd) a.rstd[row] = rstd;
  }
  bf16* y = reinterpret_cast<bf16*>(a.y) + (size_t)map_row(a.y_map, row) * a.D;
  bf16* y2 = a.y2 ? reinterpret_cast<bf16*>(a.y2) + (size_t)map_row(a.y2_map, row) * a.D : nullptr;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = lane + 64 * i;
    if (c < nch) {
      const f32x4 g = gam[i], b = bet[i];
      bf16x4 o;
      f32x4 of;
#pragma unroll
      for (int e = 0; e < 4; ++e) { of[e] = (v[i][e] - mean) * rstd * g[e] + b[e]; o[e] = f2bf(of[e]); }
      *reinterpret_cast<bf16x4*>(y + c * 4) = o;
      if (a.y_f32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.y_f32) + (size_t)row * a.D + c * 4) = of;
      if (y2) *reinterpret_cast<bf16x4*>(y2 + c * 4) = o;
    }
  }
#ifdef PH_TIMELINE
  PH_TL(8);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PH_TL(9);
  PH_TL_FLUSH((int)blockIdx.x, 0, threadIdx.x == 0);
#endif
}

// bf16 rows whose length is a multiple of 256 (every trunk LayerNorm: D = 768 / 1024 / 1280): HALF a wave owns a row, so each lane
// moves 16-B vectors (NV = D / 256 of them) -- half the memory instructions of the 8-B form above for the same bytes -- and a wave
// covers two rows; the reductions stay inside the 32-lane half (xor 16 .. 1).  Round 5: the 8320-row launches of the trunk ran at
// 2.1 TB/s on the one-row-per-wave kernel (12.4 us for 25.6 MB).
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd16_kernel(ph_layernorm_fwd_args a) {
  const int lane = threadIdx.x & 31;
  const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
  if (row >= a.M) return;
  const bf16* x = reinterpret_cast<const bf16*>(a.x) + (size_t)row * a.D;
  bf16x8 t[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) t[i] = *reinterpret_cast<const bf16x8*>(x + (lane + 32 * i) * 8);
  f32x4 g0[NV], g1[NV], b0[NV], b1[NV];     // requested with the row: their latency hides behind the two reductions
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 32 * i) * 8;
    g0[i] = *reinterpret_cast<const f32x4*>(a.gamma + c); g1[i] = *reinterpret_cast<const f32x4*>(a.gamma + c + 4);
    b0[i] = *reinterpret_cast<const f32x4*>(a.beta + c); b1[i] = *reinterpret_cast<const f32x4*>(a.beta + c + 4);
  }
  float v[NV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[i][e] = bf2f(t[i][e]); s += v[i][e]; }
  const float mean = half_sum(s) / (float)a.D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
  const float rstd = rsqrtf(half_sum(q) / (float)a.D + a.eps);
  if (lane == 0) {
    if (a.mean) a.mean[row] = mean;
    if (a.rstd) a.rstd[row] = rstd;
  }
  bf16* y = reinterpret_cast<bf16*>(a.y) + (size_t)map_row(a.y_map, row) * a.D;
  bf16* y2 = a.y2 ? reinterpret_cast<bf16*>(a.y2) + (size_t)map_row(a.y2_map, row) * a.D : nullptr;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 32 * i) * 8;
    bf16x8 o;
    f32x4 of0, of1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      of0[e] = (v[i][e] - mean) * rstd * g0[i][e] + b0[i][e];
      of1[e] = (v[i][4 + e] - mean) * rstd * g1[i][e] + b1[i][e];
      o[e] = f2bf(of0[e]); o[4 + e] = f2bf(of1[e]);
    }
    *reinterpret_cast<bf16x8*>(y + c) = o;
    if (a.y_f32) {
      float* yf = reinterpret_cast<float*>(a.y_f32) + (size_t)row * a.D + c;
      *reinterpret_cast<f32x4*>(yf) = of0;
      *reinterpret_cast<f32x4*>(yf + 4) = of1;
    }
    if (y2) *reinterpret_cast<bf16x8*>(y2 + c) = o;
  }
}

// Backward.  Each wave walks rows (grid-stride) keeping its dgamma / dbeta partials in registers; the block
// folds its 4 waves through LDS and issues one fp32 atomic per column.
// One row's HBM operands, requested a whole row ahead of their use (rows beyond M re-read row M-1: no predicate, so
// the prefetch is straight-line code and the waits in front of the math leave the next row's loads in flight).
template <int NCH, bool XF32>
struct LnBwdRow {
  bf16x4 x[XF32 ? 1 : NCH];
  f32x4 xf[XF32 ? NCH : 1];
  bf16x4 dy[NCH], dy2[NCH], dsk[NCH];
  float mean, rstd;
};

template <int NCH, bool XF32>
__device__ __forceinline__ void ln_bwd_fetch(const ph_layernorm_bwd_args& a, int row, int lane, int nch, LnBwdRow<NCH, XF32>& r) {
  row = min(row, a.M - 1);
  const bf16* x = reinterpret_cast<const bf16*>(a.x) + (size_t)row * a.D;
  const float* xf = reinterpret_cast<const float*>(a.x) + (size_t)row * a.D;
  const bf16* dy = reinterpret_cast<const bf16*>(a.dy) + (size_t)map_row(a.dy_map, row) * a.D;
  const bf16* dy2 = a.dy2 ? reinterpret_cast<const bf16*>(a.dy2) + (size_t)map_row(a.dy2_map, row) * a.D : nullptr;
  const bf16* dsk = a.dskip ? reinterpret_cast<const bf16*>(a.dskip) + (size_t)row * a.D : nullptr;
  r.mean = a.mean[row];
  r.rstd = a.rstd[row];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = min(lane + 64 * i, nch - 1);
    if (XF32) r.xf[i] = *reinterpret_cast<const f32x4*>(xf + c * 4);
    else r.x[i] = *reinterpret_cast<const bf16x4*>(x + c * 4);
    r.dy[i] = *reinterpret_cast<const bf16x4*>(dy + c * 4);
    if (dy2) r.dy2[i] = *reinterpret_cast<const bf16x4*>(dy2 + c * 4);
    if (dsk) r.dsk[i] = *reinterpret_cast<const bf16x4*>(dsk + c * 4);
  }
}

template <int NCH, bool XF32>
__global__ __launch_bounds__(256) void ln_bwd_kernel(ph_layernorm_bwd_args a) {
  __shared__ float red_flat[2 * 4 * 512];   // [dgamma|dbeta][wave][512 columns]: 16 KB, D is folded in passes of 512
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = a.D >> 2;
  float dg[NCH][4], db[NCH][4];
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; }
  DropCtx dc;
  const bool drop = a.dx_drop && a.drop_p > 0.f;
  if (drop) dc = make_drop(a.drop_seed, a.drop_stream, a.drop_p);
  const bool has_dy2 = a.dy2 != nullptr, has_dsk = a.dskip != nullptr;

  const int stride = gridDim.x * 4;
  int row = blockIdx.x * 4 + wave;
  LnBwdRow<NCH, XF32> cur, nxt;
  f32x4 gm[NCH];
  if (row < a.M) {
    ln_bwd_fetch<NCH, XF32>(a, row, lane, nch, cur);
#pragma unroll
    for (int i = 0; i < NCH; ++i) gm[i] = *reinterpret_cast<const f32x4*>(a.gamma + min(lane + 64 * i, nch - 1) * 4);
#pragma unroll
    for (int i = 0; i < NCH; ++i) asm volatile("" : "+v"(gm[i]));      // loop-invariant: settle before the row loop
  }
  // Ping-pong over two register sets (no copies): row i+1's operands are requested before row i is reduced and stored,
  // and the wait in front of row i's math leaves them in flight.
  auto process = [&](const LnBwdRow<NCH, XF32>& r, int row) {
    const float mean = r.mean, rstd = r.rstd;
    float xh[NCH][4], g[NCH][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int c = lane + 64 * i;
      if (c < nch) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float xv = XF32 ? r.xf[i][e] : bf2f(r.x[i][e]);
          float d = bf2f(r.dy[i][e]);
          if (has_dy2) d += bf2f(r.dy2[i][e]);
          xh[i][e] = (xv - mean) * rstd;
          dg[i][e] += d * xh[i][e];
          db[i][e] += d;
          g[i][e] = d * gm[i][e];
          s1 += g[i][e];
          s2 += g[i][e] * xh[i][e];
        }
      }
    }
    float m1 = wave_sum(s1) / (float)a.D, m2 = wave_sum(s2) / (float)a.D;
    bf16* dx = reinterpret_cast<bf16*>(a.dx) + (size_t)row * a.D;
    bf16* dxd = drop ? reinterpret_cast<bf16*>(a.dx_drop) + (size_t)row * a.D : nullptr;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int c = lane + 64 * i;
      if (c < nch) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = rstd * (g[i][e] - m1 - xh[i][e] * m2);
          if (has_dsk) o[e] += bf2f(r.dsk[i][e]);
        }
        bf16x4 ob = {f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
        *reinterpret_cast<bf16x4*>(dx + c * 4) = ob;
        if (drop) {
          u32x4 r = drop_rand4(dc, ((uint64_t)row * (uint64_t)a.D + (uint64_t)c * 4) >> 2);
          bf16x4 od = {f2bf(drop_apply(dc, r[0], o[0])), f2bf(drop_apply(dc, r[1], o[1])),
                       f2bf(drop_apply(dc, r[2], o[2])), f2bf(drop_apply(dc, r[3], o[3]))};
          *reinterpret_cast<bf16x4*>(dxd + c * 4) = od;
        }
      }
    }
  };
  for (; row < a.M; row += 2 * stride) {
    ln_bwd_fetch<NCH, XF32>(a, row + stride, lane, nch, nxt);
    process(cur, row);
    if (row + stride >= a.M) break;
    ln_bwd_fetch<NCH, XF32>(a, row + 2 * stride, lane, nch, cur);
    process(nxt, row + stride);
  }
  if (!a.dgamma && !a.dbeta) return;
  // fold the 4 waves: column passes of 512 columns (128 chunks) to keep LDS at 16 KB
  for (int pass = 0; pass < (a.D + 511) / 512; ++pass) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int c = lane + 64 * i;
      int cl = c - pass * 128;
      if (c < nch && cl >= 0 && cl < 128) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          red_flat[(0 * 4 + wave) * 512 + cl * 4 + e] = dg[i][e];
          red_flat[(1 * 4 + wave) * 512 + cl * 4 + e] = db[i][e];
        }
      }
    }
    __syncthreads();
    for (int col = threadIdx.x; col < 512; col += 256) {
      int gc = pass * 512 + col;
      if (gc < a.D) {
        float sg = 0.f, sb = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { sg += red_flat[(0 * 4 + w) * 512 + col]; sb += red_flat[(1 * 4 + w) * 512 + col]; }
        if (a.partial_ws) {          // per-block partials, folded by ln_param_reduce_kernel (no atomics)
          a.partial_ws[((size_t)blockIdx.x * 2 + 0) * a.D + gc] = sg;
          a.partial_ws[((size_t)blockIdx.x * 2 + 1) * a.D + gc] = sb;
        } else {
          if (a.dgamma) atomicAdd(a.dgamma + gc, sg);
          if (a.dbeta) atomicAdd(a.dbeta + gc, sb);
        }
      }
    }
  }
}

// dgamma[c] += sum_b partial[b][0][c], dbeta[c] += sum_b partial[b][1][c]; blockIdx.y strides over the partial rows
__global__ void ln_param_reduce_kernel(const float* __restrict__ ws, int nblk, int D, float* dgamma, float* dbeta) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= 2 * D) return;
  int which = c / D, col = c % D;
  float s = 0.f;
  for (int b = blockIdx.y; b < nblk; b += gridDim.y) s += ws[((size_t)b * 2 + which) * D + col];
  float* dst = which ? dbeta : dgamma;
  if (dst) atomicAdd(dst + col, s);
}

struct LnReduceGroup {
  int n;
  int blk_start[PH_GEMM_GROUP_MAX + 1];
  ph_ln_reduce_item it[PH_GEMM_GROUP_MAX];
};
// grouped form of ln_param_reduce_kernel: block -> (LayerNorm, column block, row strip)
__global__ void ln_param_reduce_grouped_kernel(LnReduceGroup g) {
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.blk_start[i + 1]) ++i;
  const ph_ln_reduce_item& t = g.it[i];
  const int lid = (int)blockIdx.x - g.blk_start[i];
  const int ncb = (2 * t.D + 255) / 256;
  const int strips = (g.blk_start[i + 1] - g.blk_start[i]) / ncb;
  const int c = (lid % ncb) * 256 + threadIdx.x, strip = lid / ncb;
  if (c >= 2 * t.D) return;
  const int which = c / t.D, col = c % t.D;
  float s = 0.f;
  for (int b = strip; b < t.blocks; b += strips) s += t.ws[((size_t)b * 2 + which) * t.D + col];
  float* dst = which ? t.dbeta : t.dgamma;
  if (dst) atomicAdd(dst + col, s);
}

}  // namespace

extern "C" int ph_layernorm_bwd_blocks(int M) { return min(ceil_div(M, 4), 768); }

extern "C" int ph_ln_param_reduce_grouped(const ph_ln_reduce_item* items, int n, hipStream_t stream) {
  PH_CHECK_ARG(items && n >= 1 && n <= PH_GEMM_GROUP_MAX, "ph_ln_param_reduce_grouped: need 1..%d items, got %d", PH_GEMM_GROUP_MAX, n);
  ProfScope prof__(PH_FAM_LAYERNORM, 0.0, 0.0, stream);
  LnReduceGroup g;
  g.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    PH_CHECK_ARG(items[i].ws && items[i].blocks > 0 && items[i].D > 0, "ph_ln_param_reduce_grouped: bad item %d", i);
    g.it[i] = items[i];
    g.blk_start[i] = total;
    total += ceil_div(2 * items[i].D, 256) * min(items[i].blocks, 32);
  }
  g.blk_start[n] = total;
  hipLaunchKernelGGL(ln_param_reduce_grouped_kernel, dim3(total), dim3(256), 0, stream, g);
  PH_LAUNCH_CHECK("ln_param_reduce_grouped_kernel");
  return PH_OK;
}

int g_ln_fwd16 = 1;                  // (diagnostics: ph_layernorm_tuning(0) forces the one-row-per-wave kernel for A/B)
extern "C" int ph_layernorm_tuning(int fwd16) { const int old = g_ln_fwd16; if (fwd16 >= 0) g_ln_fwd16 = fwd16 ? 1 : 0; return old; }
extern "C" int ph_layernorm_fwd(const ph_layernorm_fwd_args* a, hipStream_t stream) {
  PH_CHECK_ARG(a && a->x && a->y && a->gamma && a->beta, "ph_layernorm_fwd: null pointer");
  ProfScope prof__(PH_FAM_LAYERNORM, 0.0, 4.0 * a->M * (double)a->D, stream);
  PH_CHECK_ARG(a->M > 0 && a->D > 0 && (a->D % 4) == 0 && a->D <= MAX_CH * 256, "ph_layernorm_fwd: D=%d unsupported (need D%%4==0, D<=%d)", a->D, MAX_CH * 256);
  // chunks of 4 elements per lane: the row lives in NCH*4 registers per lane (templated so D=768 does not pay for 2048)
  const int nch = ceil_div(a->D, 256);
  // bf16 rows of a multiple of 256 elements, 16-B aligned everywhere: half a wave per row, 16-B vectors (ln_fwd16_kernel)
  if (g_ln_fwd16 && !a->x_f32 && (a->D % 256) == 0 && a->D <= 1280 && a->M >= 1024 &&
      (((uintptr_t)a->x | (uintptr_t)a->y | (uintptr_t)a->y2 | (uintptr_t)a->y_f32 | (uintptr_t)a->gamma | (uintptr_t)a->beta) & 15) == 0) {
    dim3 g16(ceil_div(a->M, 8));
    switch (a->D / 256) {
      case 1: hipLaunchKernelGGL(ln_fwd16_kernel<1>, g16, dim3(256), 0, stream, *a); break;
      case 2: hipLaunchKernelGGL(ln_fwd16_kernel<2>, g16, dim3(256), 0, stream, *a); break;
      case 3: hipLaunchKernelGGL(ln_fwd16_kernel<3>, g16, dim3(256), 0, stream, *a); break;
      case 4: hipLaunchKernelGGL(ln_fwd16_kernel<4>, g16, dim3(256), 0, stream, *a); break;
      default: hipLaunchKernelGGL(ln_fwd16_kernel<5>, g16, dim3(256), 0, stream, *a); break;
    }
    PH_LAUNCH_CHECK("ln_fwd16_kernel");
    return PH_OK;
  }
  dim3 grid(ceil_div(a->M, 4));
  if (nch <= 1) hipLaunchKernelGGL(ln_fwd_kernel<1>, grid, dim3(256), 0, stream, *a);
  else if (nch <= 2) hipLaunchKernelGGL(ln_fwd_kernel<2>, grid, dim3(256), 0, stream, *a);
  else if (nch <= 3) hipLaunchKernelGGL(ln_fwd_kernel<3>, grid, dim3(256), 0, stream, *a);
  else if (nch <= 4) hipLaunchKernelGGL(ln_fwd_kernel<4>, grid, dim3(256), 0, stream, *a);
  else hipLaunchKernelGGL(ln_fwd_kernel<8>, grid, dim3(256), 0, stream, *a);
  PH_LAUNCH_CHECK("ln_fwd_kernel");
  return PH_OK;
}

extern "C" int ph_layernorm_bwd(const ph_layernorm_bwd_args* a, hipStream_t stream) {
  PH_CHECK_ARG(a && a->dy && a->x && a->mean && a->rstd && a->gamma && a->dx, "ph_layernorm_bwd: null pointer");
  ProfScope prof__(PH_FAM_LAYERNORM, 0.0, 6.0 * a->M * (double)a->D, stream);
  PH_CHECK_ARG(a->M > 0 && a->D > 0 && (a->D % 4) == 0 && a->D <= MAX_CH * 256, "ph_layernorm_bwd: D=%d unsupported", a->D);
  PH_CHECK_ARG(!a->dx_drop || !(a->drop_p > 0.f) || a->drop_seed, "ph_layernorm_bwd: dropout needs a seed");
  int grid = ph_layernorm_bwd_blocks(a->M);
  ph_layernorm_bwd_args b = *a;
  const bool need_params = a->dgamma || a->dbeta;
  if (!need_params || (int64_t)grid * 2 * a->D * 4 > a->partial_ws_bytes) b.partial_ws = nullptr;
  const int nch = ceil_div(a->D, 256);
#define PH_LN_BWD(N)                                                                                          \
  do {                                                                                                       \
    if (b.x_f32) hipLaunchKernelGGL((ln_bwd_kernel<N, true>), dim3(grid), dim3(256), 0, stream, b);          \
    else hipLaunchKernelGGL((ln_bwd_kernel<N, false>), dim3(grid), dim3(256), 0, stream, b);                 \
  } while (0)
  if (nch <= 1) PH_LN_BWD(1);
  else if (nch <= 2) PH_LN_BWD(2);
  else if (nch <= 3) PH_LN_BWD(3);
  else if (nch <= 4) PH_LN_BWD(4);
  else PH_LN_BWD(8);
#undef PH_LN_BWD
  PH_CHECK_ARG(!a->defer_reduce || !need_params || b.partial_ws, "ph_layernorm_bwd: defer_reduce needs a partial_ws of >= blocks*2*D floats");
  if (need_params && b.partial_ws && !a->defer_reduce)
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(ceil_div(2 * a->D, 256), min(grid, 32)), dim3(256), 0, stream, b.partial_ws, grid, a->D, a->dgamma, a->dbeta);
  PH_LAUNCH_CHECK("ln_bwd_kernel");
  return PH_OK;
}

#ifdef PH_TIMELINE
extern "C" int ph_tl_fetch_norm(unsigned long long* host, int n, int reset) {
  hipDeviceSynchronize();
  if (host && n > 0) hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tl), sizeof(unsigned long long) * (size_t)n);
  if (reset) { void* d = nullptr; hipGetSymbolAddress(&d, HIP_SYMBOL(g_tl)); hipMemset(d, 0, sizeof(g_tl)); }
  return 0;
}
#endif
