// Decoder embeddings (gather + LayerNorm + dropout) and the shifted label-smoothed cross entropy, gfx950.
// Replaces RobertaEmbeddings.forward (roberta.py:38-45,66-76) and the loss of
// RobertaForCausalLMModified.forward (roberta.py:381-387: CrossEntropyLoss(reduction='none',
// label_smoothing=0.1) on shifted logits, summed per sample), plus their autograd.
#include <float.h>

#include "common.h"

namespace {

constexpr int MAX_CH = 8;   // H <= 2048

// position id of token t in row b: cumsum(ids != pad)[t] * (ids[t] != pad) + pad   (roberta.py:38-45)
__device__ __forceinline__ int position_id(const int64_t* ids_row, int t, int pad, int lane) {
  int cnt = 0;
  for (int j0 = 0; j0 <= t; j0 += 64) {
    int j = j0 + lane;
    cnt += (j <= t && ids_row[j] != pad) ? 1 : 0;
  }
  cnt = (int)wave_sum((float)cnt);
  return ids_row[t] != pad ? cnt + pad : pad;
}

__global__ __launch_bounds__(256) void embed_fwd_kernel(ph_embed_fwd_args a) {
  int lane = threadIdx.x & 63;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.B * a.T) return;
  int b = row / a.T, t = row % a.T;
  const int64_t* ids_row = a.ids + (int64_t)b * a.T;
  int64_t id = ids_row[t];
  int pid = position_id(ids_row, t, a.pad_id, lane);
  const int nch = a.H >> 2;
  const float* w = a.word + id * a.H;
  const float* p = a.pos + (int64_t)pid * a.H;
  float v[MAX_CH][4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAX_CH; ++i) {
    int c = lane + 64 * i;
    if (c < nch) {
      f32x4 tw = *reinterpret_cast<const f32x4*>(w + c * 4);
      f32x4 tp = *reinterpret_cast<const f32x4*>(p + c * 4);
      f32x4 tt = *reinterpret_cast<const f32x4*>(a.type + c * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[i][e] = (tw[e] + tt[e]) + tp[e]; s += v[i][e]; }   // roberta.py:72-73 order
    }
  }
  float mean = wave_sum(s) / (float)a.H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAX_CH; ++i) {
    int c = lane + 64 * i;
    if (c < nch) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { float d = v[i][e] - mean; q += d * d; }
    }
  }
  float rstd = rsqrtf(wave_sum(q) / (float)a.H + a.eps);
  if (lane == 0 && a.rstd) a.rstd[row] = rstd;
  DropCtx dc;
  const bool drop = a.drop_p > 0.f;
  if (drop) dc = make_drop(a.drop_seed, a.drop_stream, a.drop_p);
  bf16* out = reinterpret_cast<bf16*>(a.out) + (int64_t)row * a.H;
  bf16* xh = a.xhat ? reinterpret_cast<bf16*>(a.xhat) + (int64_t)row * a.H : nullptr;
#pragma unroll
  for (int i = 0; i < MAX_CH; ++i) {
    int c = lane + 64 * i;
    if (c < nch) {
      f32x4 g = *reinterpret_cast<const f32x4*>(a.gamma + c * 4);
      f32x4 bt = *reinterpret_cast<const f32x4*>(a.beta + c * 4);
      float o[4], h[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { h[e] = (v[i][e] - mean) * rstd; o[e] = h[e] * g[e] + bt[e]; }
      if (xh) { bf16x4 th = {f2bf(h[0]), f2bf(h[1]), f2bf(h[2]), f2bf(h[3])}; *reinterpret_cast<bf16x4*>(xh + c * 4) = th; }
      if (drop) {
        u32x4 r = drop_rand4(dc, ((uint64_t)row * (uint64_t)a.H + (uint64_t)c * 4) >> 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = drop_apply(dc, r[e], o[e]);
      }
      bf16x4 to = {f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
      *reinterpret_cast<bf16x4*>(out + c * 4) = to;
      if (a.out_f32) { f32x4 tf = {o[0], o[1], o[2], o[3]}; *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out_f32) + (int64_t)row * a.H + c * 4) = tf; }
    }
  }
}

__global__ __launch_bounds__(256) void embed_bwd_kernel(ph_embed_bwd_args a) {
  __shared__ float red[3 * 4 * 512];
  const ph_embed_fwd_args& f = a.f;
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = f.H >> 2;
  float dg[MAX_CH][4], db[MAX_CH][4], dt[MAX_CH][4];
#pragma unroll
  for (int i = 0; i < MAX_CH; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; dt[i][e] = 0.f; }
  DropCtx dc;
  const bool drop = f.drop_p > 0.f;
  if (drop) dc = make_drop(f.drop_seed, f.drop_stream, f.drop_p);
  const int rows = f.B * f.T;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    int b = row / f.T, t = row % f.T;
    const int64_t* ids_row = f.ids + (int64_t)b * f.T;
    int64_t id = ids_row[t];
    int pid = position_id(ids_row, t, f.pad_id, lane);
    const bf16* dout = reinterpret_cast<const bf16*>(a.dout) + (int64_t)row * f.H;
    const bf16* xh = reinterpret_cast<const bf16*>(f.xhat) + (int64_t)row * f.H;
    float rstd = f.rstd[row];
    float g[MAX_CH][4], h[MAX_CH][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_CH; ++i) {
      int c = lane + 64 * i;
      if (c < nch) {
        bf16x4 td = *reinterpret_cast<const bf16x4*>(dout + c * 4);
        bf16x4 th = *reinterpret_cast<const bf16x4*>(xh + c * 4);
        f32x4 gm = *reinterpret_cast<const f32x4*>(f.gamma + c * 4);
        u32x4 r;
        if (drop) r = drop_rand4(dc, ((uint64_t)row * (uint64_t)f.H + (uint64_t)c * 4) >> 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float d = bf2f(td[e]);
          if (drop) d = drop_apply(dc, r[e], d);
          h[i][e] = bf2f(th[e]);
          dg[i][e] += d * h[i][e];
          db[i][e] += d;
          g[i][e] = d * gm[e];
          s1 += g[i][e];
          s2 += g[i][e] * h[i][e];
        }
      }
    }
    float m1 = wave_sum(s1) / (float)f.H, m2 = wave_sum(s2) / (float)f.H;
    float* dw = (a.dword && id != f.pad_id) ? a.dword + id * f.H : nullptr;       // Embedding(padding_idx): no grad
    float* dp = (a.dpos && pid != f.pad_id) ? a.dpos + (int64_t)pid * f.H : nullptr;
#pragma unroll
    for (int i = 0; i < MAX_CH; ++i) {
      int c = lane + 64 * i;
      if (c < nch) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float dx = rstd * (g[i][e] - m1 - h[i][e] * m2);
          dt[i][e] += dx;
          if (dw) atomicAdd(dw + c * 4 + e, dx);
          if (dp) atomicAdd(dp + c * 4 + e, dx);
        }
      }
    }
  }
  for (int pass = 0; pass < (f.H + 511) / 512; ++pass) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAX_CH; ++i) {
      int c = lane + 64 * i;
      int cl = c - pass * 128;
      if (c < nch && cl >= 0 && cl < 128) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          red[(0 * 4 + wave) * 512 + cl * 4 + e] = dg[i][e];
          red[(1 * 4 + wave) * 512 + cl * 4 + e] = db[i][e];
          red[(2 * 4 + wave) * 512 + cl * 4 + e] = dt[i][e];
        }
      }
    }
    __syncthreads();
    for (int col = threadIdx.x; col < 512; col += 256) {
      int gc = pass * 512 + col;
      if (gc < f.H) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { s0 += red[(0 * 4 + w) * 512 + col]; s1 += red[(1 * 4 + w) * 512 + col]; s2 += red[(2 * 4 + w) * 512 + col]; }
        if (a.dgamma) atomicAdd(a.dgamma + gc, s0);
        if (a.dbeta) atomicAdd(a.dbeta + gc, s1);
        if (a.dtype) atomicAdd(a.dtype + gc, s2);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ CE
__device__ __forceinline__ float block_reduce(float v, float* sh, bool is_max) {
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = sh[0];
#pragma unroll
  for (int w = 1; w < 4; ++w) r = is_max ? fmaxf(r, sh[w]) : r + sh[w];
  return r;
}

// out[row][j] = softmax(logits[row, :V])[ids[j]]: the first-token probabilities of the answer candidates of `inference='rank'`
// (prismer_caption.py:70, prismer_vqa.py:51: `F.softmax(logits, dim=1).index_select(dim=1, index=answer_first_token)`).  One block per row:
// the same single-sweep max / sum-of-exponentials as ce_fwd_kernel, then the gathered columns.
__global__ __launch_bounds__(256) void softmax_gather_kernel(const bf16* __restrict__ logits, int64_t ld, int V, const int64_t* __restrict__ ids, int n,
                                                             float* __restrict__ out) {
  __shared__ float sh[4];
  const bf16* x = logits + (int64_t)blockIdx.x * ld;
  float mx = -FLT_MAX, se = 0.f;
  const int nch = (V + 7) / 8;
  for (int c = threadIdx.x; c < nch; c += 256) {
    bf16x8 tv = *reinterpret_cast<const bf16x8*>(x + c * 8);
    float v[8];
    float cm = -FLT_MAX;
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] = (c * 8 + e < V) ? bf2f(tv[e]) : -INFINITY; cm = fmaxf(cm, v[e]); }
    const float nm = fmaxf(mx, cm);
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += __expf(v[e] - nm);
    se = se * __expf(mx - nm) + acc;
    mx = nm;
  }
  const float gm = block_reduce(mx, sh, true);
  const float gs = block_reduce(se * __expf(mx - gm), sh, false);
  const float inv = 1.0f / gs;
  for (int j = threadIdx.x; j < n; j += 256) {
    const int64_t id = ids[j];
    out[(int64_t)blockIdx.x * n + j] = (id >= 0 && id < V) ? __expf(bf2f(x[id]) - gm) * inv : 0.f;
  }
}

// Per-row loss into row_loss[row] (0 for ignored / last positions), summed per sample by ce_sample_sum_kernel: no atomics and -- the round-5
// finding -- NO hipMemsetAsync in front of the kernel: a small memset NODE of a captured hipGraph does not replay correctly on this ROCm
// (tools/graph_memset_probe.py, profiles/r5_graph_memset_order.txt: from the second replay on it leaves junk instead of zeros), which is what
// made the loader leg of round 4 report 1e8..1e18 losses while the training state itself stayed healthy.
__global__ __launch_bounds__(256) void ce_fwd_kernel(const bf16* __restrict__ logits, int ld, const int64_t* __restrict__ labels, int B,
                                                     int T, int V, float eps, float* __restrict__ row_loss, float* __restrict__ row_lse) {
  __shared__ float sh[4];
  int row = blockIdx.x;                 // row = b*T + t
  int b = row / T, t = row % T;
  int64_t lab = (t < T - 1) ? labels[(int64_t)b * T + t + 1] : -100;
  if (lab < 0) {                        // ignore_index = -100 (and the last position, which has no next token)
    if (threadIdx.x == 0) row_loss[row] = 0.f;
    return;
  }
  const bf16* x = logits + (int64_t)row * ld;
  float mx = -FLT_MAX, se = 0.f, sl = 0.f;
  const int nch = (V + 7) / 8;
  for (int c = threadIdx.x; c < nch; c += 256) {
    bf16x8 tv = *reinterpret_cast<const bf16x8*>(x + c * 8);
    float v[8];
    float cm = -FLT_MAX;
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] = (c * 8 + e < V) ? bf2f(tv[e]) : -INFINITY; cm = fmaxf(cm, v[e]); }
    float nm = fmaxf(mx, cm);
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { acc += __expf(v[e] - nm); sl += (c * 8 + e < V) ? v[e] : 0.f; }
    se = se * __expf(mx - nm) + acc;
    mx = nm;
  }
  float gm = block_reduce(mx, sh, true);
  float gs = block_reduce(se * __expf(mx - gm), sh, false);
  float gl = block_reduce(sl, sh, false);
  if (threadIdx.x == 0) {
    float lse = gm + __logf(gs);
    row_lse[row] = lse;
    float nll = lse - bf2f(x[lab]);
    float smooth = lse - gl / (float)V;
    row_loss[row] = (1.f - eps) * nll + eps * smooth;
  }
}

// loss[b] = sum_t row_loss[b, t]: one wave per sample, fixed summation order (run-to-run reproducible, unlike the atomics it replaces)
__global__ __launch_bounds__(256) void ce_sample_sum_kernel(const float* __restrict__ row_loss, int B, int T, float* __restrict__ loss) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  float s = 0.f;
  for (int t = lane; t < T; t += 64) s += row_loss[(int64_t)b * T + t];
  s = wave_sum(s);
  if (lane == 0) loss[b] = s;
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(bf16* __restrict__ logits, int ld, const int64_t* __restrict__ labels, int B, int T, int V,
                                                     int Vpad, float eps, const float* __restrict__ row_lse, const float* __restrict__ dloss) {
  int row = blockIdx.x;
  int b = row / T, t = row % T;
  bf16* x = logits + (int64_t)row * ld;
  int64_t lab = (t < T - 1) ? labels[(int64_t)b * T + t + 1] : -100;
  const int nch = Vpad / 8;
  if (lab < 0) {
    u32x4 z = {0u, 0u, 0u, 0u};
    for (int c = threadIdx.x; c < nch; c += 256) *reinterpret_cast<u32x4*>(x + c * 8) = z;
    return;
  }
  float lse = row_lse[row], g = dloss[b], smooth = eps / (float)V;
  for (int c = threadIdx.x; c < nch; c += 256) {
    bf16x8 tv = *reinterpret_cast<const bf16x8*>(x + c * 8);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int col = c * 8 + e;
      float d = 0.f;
      if (col < V) d = g * (__expf(bf2f(tv[e]) - lse) - (col == lab ? (1.f - eps) : 0.f) - smooth);
      o[e] = f2bf(d);
    }
    *reinterpret_cast<bf16x8*>(x + c * 8) = o;
  }
}

}  // namespace

extern "C" int ph_embed_fwd(const ph_embed_fwd_args* a, hipStream_t stream) {
  PH_CHECK_ARG(a && a->ids && a->word && a->pos && a->type && a->gamma && a->beta && a->out, "ph_embed_fwd: null pointer");
  ProfScope prof__(PH_FAM_EMBED_CE, 0.0, 0.0, stream);
  PH_CHECK_ARG(a->H % 4 == 0 && a->H <= MAX_CH * 256 && a->B > 0 && a->T > 0, "ph_embed_fwd: bad dims");
  PH_CHECK_ARG(!(a->drop_p > 0.f) || a->drop_seed, "ph_embed_fwd: dropout needs a seed");
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(ceil_div(a->B * a->T, 4)), dim3(256), 0, stream, *a);
  PH_LAUNCH_CHECK("embed_fwd_kernel");
  return PH_OK;
}

extern "C" int ph_embed_bwd(const ph_embed_bwd_args* a, hipStream_t stream) {
  PH_CHECK_ARG(a && a->dout && a->f.ids && a->f.xhat && a->f.rstd && a->f.gamma, "ph_embed_bwd: null pointer");
  ProfScope prof__(PH_FAM_EMBED_CE, 0.0, 0.0, stream);
  PH_CHECK_ARG(a->f.H % 4 == 0 && a->f.H <= MAX_CH * 256, "ph_embed_bwd: bad dims");
  int grid = min(ceil_div(a->f.B * a->f.T, 4), 256);
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid), dim3(256), 0, stream, *a);
  PH_LAUNCH_CHECK("embed_bwd_kernel");
  return PH_OK;
}

extern "C" int ph_ce_fwd(const void* logits, int ld, const int64_t* labels, int B, int T, int V, float eps, float* loss,
                         float* row_lse, float* row_loss, hipStream_t stream) {
  PH_CHECK_ARG(logits && labels && loss && row_lse && row_loss && ld % 8 == 0 && ld >= V && B > 0 && T > 1, "ph_ce_fwd: bad args");
  ProfScope prof__(PH_FAM_EMBED_CE, 0.0, 2.0 * B * (double)T * V, stream);
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(B * T), dim3(256), 0, stream, (const bf16*)logits, ld, labels, B, T, V, eps, row_loss, row_lse);
  hipLaunchKernelGGL(ce_sample_sum_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, stream, (const float*)row_loss, B, T, loss);
  PH_LAUNCH_CHECK("ce_fwd_kernel");
  return PH_OK;
}

extern "C" int ph_softmax_gather_bf16(const void* logits, int64_t ld, int rows, int V, const int64_t* ids, int n, float* out, hipStream_t stream) {
  PH_CHECK_ARG(logits && ids && out && rows > 0 && V > 0 && n > 0 && ld >= V && (ld % 8) == 0 && (((uintptr_t)logits) & 15) == 0, "ph_softmax_gather_bf16: bad args");
  ProfScope prof__(PH_FAM_EMBED_CE, 0.0, 2.0 * rows * (double)V, stream);
  hipLaunchKernelGGL(softmax_gather_kernel, dim3(rows), dim3(256), 0, stream, (const bf16*)logits, ld, V, ids, n, out);
  PH_LAUNCH_CHECK("softmax_gather_kernel");
  return PH_OK;
}

extern "C" int ph_ce_bwd(void* logits, int ld, const int64_t* labels, int B, int T, int V, int Vpad, float eps, const float* row_lse,
                         const float* dloss, hipStream_t stream) {
  PH_CHECK_ARG(logits && labels && row_lse && dloss && ld % 8 == 0 && Vpad % 8 == 0 && Vpad <= ld && Vpad >= V, "ph_ce_bwd: bad args");
  ProfScope prof__(PH_FAM_EMBED_CE, 0.0, 4.0 * B * (double)T * V, stream);
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(B * T), dim3(256), 0, stream, (bf16*)logits, ld, labels, B, T, V, Vpad, eps, row_lse, dloss);
  PH_LAUNCH_CHECK("ce_bwd_kernel");
  return PH_OK;
}
