// bf16 MFMA GEMM with fused epilogue for gfx950 (CDNA4).
//
//   C[M,N] = epilogue( alpha * sum_k opA[m,k] * opB[n,k] )
//
// Replaces every nn.Linear / conv-as-GEMM contraction on the Prismer hot path:
//   forward   x[M,K] . W[N,K]^T            (both K-contiguous)               -> trans_a=0, trans_b=0
//   dgrad     dY[M,N'] . W[N',K']          (W stored [red][out])             -> trans_a=0, trans_b=1
//   wgrad     dY[M',N]^T . X[M',K]         (both stored [red][out])          -> trans_a=1, trans_b=1
// reference call sites: vit.py:41-47,53 (ViT), resampler.py:18-24, utils.py:50-56 (Adaptor),
// roberta.py:86-92,134,163,177,415-421 (decoder), vit.py:86-120 (convs lowered to im2col GEMMs).
//
// Design (wave64, MFMA 32x32x16 bf16, fp32 accumulate) -- details and measurements in DESIGN.md section 3:
//   * block = 256 threads = 4 waves in a 2x2 grid; block tile 128x128 or 64x64 (cost model on the host), BK = 64.
//   * operands staged global -> VGPR -> LDS (double buffer, one barrier per k-tile); the registers of tile t+1 are
//     written to LDS right after the barrier and immediately re-issued for tile t+2 (two tiles of look-ahead).
//   * K-contiguous operand: LDS image [rows][64] bf16 (128-B rows), 16-B chunk index XOR-swizzled with (row>>1)&7 so
//     the ds_read_b128 fragment reads of a 16-lane group hit 16 distinct 16-B slots.
//   * K-strided operand (stored [K][rows]): LDS image [64][rows] with a 64-B row pad, fragments fetched with the gfx950
//     transposing LDS read ds_read_b64_tr_b16 (two per 8-element fragment) -- no transposed copies anywhere.
//   * epilogue through LDS (row-contiguous 16-B vectors), fused bias / activation / act' / dropout / residual / accumulate.
//   * split-K (grid.z): partial tiles to an fp32 workspace + splitk_reduce_kernel (which runs the same epilogue).
//   * XCD-aware, grouped tile rasterisation (each XCD walks 8-row-panel groups, rows fastest).
#include "gemm_common.h"
#include <vector>

using namespace phg;

namespace {

// folds the split-K partials and applies the fused epilogue (bias / activation / dropout / residual / accumulate / dtype).
// LANES threads share one output vector: lane l sums the splits l, l + LANES, ... and the group is folded with shuffles -- with up to
// 256 splits of a tiny output (the stems' first-layer weight gradients: 768 output vectors) one thread per vector walked 256
// dependent loads (62 us for 3 blocks); 16 lanes per vector make it 16 loads and 48 blocks.
template <int LANES>
__device__ __forceinline__ void splitk_reduce_body(const GemmParams& p, const int splits, const int block, const int nblocks) {
  const int n4 = (p.N + 3) / 4;
  int64_t total = (int64_t)p.M * n4;
  DropCtx dc;
  const bool drop = p.drop_p > 0.0f;
  if (drop) dc = make_drop(p.drop_seed, p.drop_stream, p.drop_p);
  const int sub = threadIdx.x % LANES;
  // (the LANES lanes of a group share `id`, hence their control flow: the shuffles below always see the whole group)
  for (int64_t id = ((int64_t)block * 256 + threadIdx.x) / LANES; id < total; id += (int64_t)nblocks * (256 / LANES)) {
    int m = (int)(id / n4), n = (int)(id % n4) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = sub; s < splits; s += LANES) acc += *reinterpret_cast<const f32x4*>(p.ws + ((size_t)s * p.M + m) * p.ldws + n);
    if constexpr (LANES > 1) {
#pragma unroll
      for (int o = LANES / 2; o > 0; o >>= 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
      }
    }
    if (sub == 0) {
      float v[4] = {acc[0], acc[1], acc[2], acc[3]};
      epilogue_store(p, m, n, v, false, drop, dc);
    }
  }
}
template <int LANES>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmParams p, int splits) {
  splitk_reduce_body<LANES>(p, splits, blockIdx.x, gridDim.x);
}
// the fold passes of several deferred split-K GEMMs in one grid (ph_gemm_args.defer_reduce / ph_gemm_flush_deferred)
struct ReduceGroup {
  int n;
  int blk_start[PH_GEMM_GROUP_MAX + 1];
  int splits[PH_GEMM_GROUP_MAX];
  GemmParams p[PH_GEMM_GROUP_MAX];
};
template <int LANES>
__global__ __launch_bounds__(256) void splitk_reduce_grouped_kernel(ReduceGroup g) {
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.blk_start[i + 1]) ++i;
  splitk_reduce_body<LANES>(g.p[i], g.splits[i], (int)blockIdx.x - g.blk_start[i], g.blk_start[i + 1] - g.blk_start[i]);
}

}  // namespace


namespace phg { std::atomic<long long> g_gemm_counts[PH_GEMM_CLS_COUNT]; }
// run-time tuning (ph_gemm_tuning); -1 = not set yet: the first reader takes the environment default
static std::atomic<int> g_big_mode{-1}, g_big_min_tiles{-1};
static int big_mode_now() {
  int m = g_big_mode.load(std::memory_order_relaxed);
  if (m < 0) m = 7;
  return m;
}
static int big_min_tiles_now() {
  int m = g_big_min_tiles.load(std::memory_order_relaxed);
  if (m < 0) m = 128;
  return m;
}

// ---- deferred fold passes (ph_gemm_args.defer_reduce): one queue per process, see the header for the calling contract
namespace {
struct DeferredReduce { GemmParams p; int splits; bool wide; };    // wide: the 16-lanes-per-vector form
std::mutex g_defer_mu;
std::vector<DeferredReduce> g_defer;
size_t g_defer_cursor = 0;              // bytes of the caller's workspace taken by the queued partial sums
const void* g_defer_ws = nullptr;

int reduce_blocks(const GemmParams& p, bool wide) {
  const int64_t vecs = (int64_t)p.M * ((p.N + 3) / 4);
  return (int)std::min<int64_t>(2048, ceil_div64(wide ? vecs * 16 : vecs, 256));
}
int flush_deferred_locked(hipStream_t stream) {
  for (int wide = 0; wide < 2; ++wide) {
    size_t i = 0;
    while (true) {
      ReduceGroup g;
      g.n = 0; g.blk_start[0] = 0;
      for (; i < g_defer.size() && g.n < PH_GEMM_GROUP_MAX; ++i) {
        if ((int)g_defer[i].wide != wide) continue;
        g.p[g.n] = g_defer[i].p; g.splits[g.n] = g_defer[i].splits;
        g.blk_start[g.n + 1] = g.blk_start[g.n] + reduce_blocks(g_defer[i].p, wide != 0);
        ++g.n;
      }
      if (g.n == 0) break;
      count_launch(PH_GEMM_CLS_SPLITK_REDUCE);
      if (wide) hipLaunchKernelGGL(splitk_reduce_grouped_kernel<16>, dim3(g.blk_start[g.n]), dim3(256), 0, stream, g);
      else hipLaunchKernelGGL(splitk_reduce_grouped_kernel<1>, dim3(g.blk_start[g.n]), dim3(256), 0, stream, g);
      PH_LAUNCH_CHECK("splitk_reduce_grouped_kernel");
    }
  }
  g_defer.clear();
  g_defer_cursor = 0;
  g_defer_ws = nullptr;
  return PH_OK;
}
}  // namespace


extern "C" int ph_gemm_flush_deferred(hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_defer_mu);
  if (g_defer.empty()) return PH_OK;
  return flush_deferred_locked(stream);
}

extern "C" int ph_gemm_tuning(int big_mode, int big_min_tiles) {
  if (big_mode >= 0) g_big_mode.store(big_mode, std::memory_order_relaxed);
  if (big_min_tiles >= 0) g_big_min_tiles.store(big_min_tiles, std::memory_order_relaxed);
  return PH_OK;
}

extern "C" int ph_gemm_dispatch_counts(int64_t* out, int n, int reset) {
  PH_CHECK_ARG(n >= 0 && (out || n == 0), "ph_gemm_dispatch_counts: null output");
  for (int i = 0; i < PH_GEMM_CLS_COUNT; ++i) {
    const long long v = reset ? g_gemm_counts[i].exchange(0, std::memory_order_relaxed) : g_gemm_counts[i].load(std::memory_order_relaxed);
    if (i < n) out[i] = (int64_t)v;
  }
  return PH_GEMM_CLS_COUNT;
}

// device-resident zero vector standing in for a null bias (epilogue classes, gemm_common.h): part of the code object, no allocation
static __device__ float g_zero_bias[PH_ZERO_BIAS_FLOATS];
// (the address of a __device__ array is PER DEVICE: cached per device id -- a process that launches on a second GPU must not hand it the first
// GPU's address; round-4 advisor finding)
static const float* zero_bias_ptr() {
  constexpr int MAX_DEV = 64;
  static std::atomic<const float*> cache[MAX_DEV];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
  const float* ptr = cache[dev].load(std::memory_order_acquire);
  if (!ptr) {
    void* d = nullptr;
    if (hipGetSymbolAddress(&d, HIP_SYMBOL(g_zero_bias)) != hipSuccess) return nullptr;
    ptr = static_cast<const float*>(d);
    cache[dev].store(ptr, std::memory_order_release);
  }
  return ptr;
}

// argument validation + kernel parameter block shared by the single and the grouped entry point
static int fill_params(const ph_gemm_args* a, GemmParams& p) {
  PH_CHECK_ARG(a && a->A && a->B && a->C, "ph_gemm_bf16: null pointer");
  PH_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0, "ph_gemm_bf16: bad dims M=%d N=%d K=%d", a->M, a->N, a->K);
  PH_CHECK_ARG(((a->lda % 8) == 0 || (a->conv && !a->trans_a)) && ((a->ldb % 8) == 0 || (a->conv && a->trans_b)),
               "ph_gemm_bf16: lda/ldb must be multiples of 8 (16-B rows)");
  PH_CHECK_ARG((((uintptr_t)a->A | (uintptr_t)a->B) & 15) == 0, "ph_gemm_bf16: A/B must be 16-B aligned");
  PH_CHECK_ARG(a->trans_a || a->trans_b || (a->K % 8) == 0, "ph_gemm_bf16: K %% 8 != 0 needs a K-strided operand");
  PH_CHECK_ARG((a->lda >= (a->trans_a ? a->M : a->K) || (a->conv && !a->trans_a)) && (a->ldb >= (a->trans_b ? a->N : a->K) || (a->conv && a->trans_b)) &&
               a->ldc >= a->N, "ph_gemm_bf16: leading dimension too small");
  PH_CHECK_ARG(!(a->drop_p > 0.0f) || ((a->N % 4) == 0 && a->drop_seed), "ph_gemm_bf16: dropout needs N %% 4 == 0 and a seed");
  PH_CHECK_ARG(a->drop_p >= 0.0f && a->drop_p < 1.0f, "ph_gemm_bf16: bad dropout p");
  p.A = (const bf16*)a->A; p.B = (const bf16*)a->B; p.C = a->C;
  p.M = a->M; p.N = a->N; p.K = a->K; p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc;
  p.bias = a->bias; p.bias_or_zero = a->bias ? a->bias : (a->N <= PH_ZERO_BIAS_FLOATS ? zero_bias_ptr() : nullptr);
  p.zero16 = reinterpret_cast<const bf16*>(zero_bias_ptr());
  p.pre_out = (bf16*)a->pre_out; p.act_in = (const bf16*)a->act_in; p.ld_act = a->ld_act;
  p.residual = (const bf16*)a->residual; p.ldr = a->ldr; p.res_f32 = a->residual_f32;
  p.drop_p = a->drop_p; p.drop_seed = a->drop_seed; p.drop_stream = a->drop_stream;
  p.act = a->act; p.out_f32 = a->out_f32; p.accumulate = a->accumulate; p.alpha = a->alpha; p.pre_grad = a->pre_grad;
  p.ws = nullptr; p.ldws = (a->N + 3) / 4 * 4;
  p.col_stats = a->col_stats;
  p.cv = ConvGather{};
  p.rm_wo = a->rowmap_wo; p.rm_mul = a->rowmap_mul; p.rm_sub = a->rowmap_sub; p.rm_add = a->rowmap_add;
  p.rm_inv_wo = a->rowmap_wo > 0 ? 1.0f / (float)a->rowmap_wo : 0.f;
  PH_CHECK_ARG(a->rowmap_wo >= 0 && (a->rowmap_wo == 0 || (!a->pre_out && !a->accumulate && a->split_k <= 1 && a->M < (1 << 24) && a->rowmap_mul >= 1 && a->rowmap_sub >= 0)),
               "ph_gemm_bf16: output row map needs a plain store epilogue (no pre_out / accumulate / split-K) and M < 2^24");
  if (a->conv) {
    const ph_conv_gather& c = *a->conv;
    const bool general = c.kh > 0;
    PH_CHECK_ARG(c.B > 0 && c.H > 0 && c.W > 0 && c.C > 0 && (c.C % 8) == 0 && (general || c.ks == 1 || c.ks == 3) && c.stride >= 1,
                 "ph_gemm_bf16: bad conv gather (C %% 8 == 0, ks 1 or 3)");
    PH_CHECK_ARG(!general || (c.kh >= 1 && c.kh <= 3 && c.kw >= 1 && c.kw <= 3 && c.Ho > 0 && c.Wo > 0 && !a->trans_a && !a->trans_b),
                 "ph_gemm_bf16: generalised conv window: 1..3 x 1..3 taps, forward-shaped gather only");
    const int pad = c.ks / 2;
    const int Ho = general ? c.Ho : (c.H + 2 * pad - c.ks) / c.stride + 1, Wo = general ? c.Wo : (c.W + 2 * pad - c.ks) / c.stride + 1;
    const int64_t rows = (int64_t)c.B * Ho * Wo;
    PH_CHECK_ARG(rows < (1 << 24) && (int64_t)c.B * c.H * c.W < (1ll << 31) / 1, "ph_gemm_bf16: conv gather index space too large");
    const int Kreal = general ? c.kh * c.kw * c.C : c.ks * c.ks * c.C;
    if (!a->trans_a && !a->trans_b) PH_CHECK_ARG(a->M == rows && a->K >= Kreal && (a->K % 8) == 0, "ph_gemm_bf16: conv A: M must be B*Ho*Wo (%lld), K >= ks*ks*C", (long long)rows);
    else if (a->trans_a && a->trans_b) PH_CHECK_ARG(a->K == rows && a->N >= Kreal, "ph_gemm_bf16: conv B: K must be B*Ho*Wo (%lld), N >= ks*ks*C", (long long)rows);
    else return ph_fail(PH_ERR_UNSUPPORTED, "ph_gemm_bf16: conv gather is defined for the NN (forward) and TT (weight gradient) layouts");
    p.cv.H = c.H; p.cv.W = c.W; p.cv.C = c.C; p.cv.ks = c.ks; p.cv.stride = c.stride; p.cv.Ho = Ho; p.cv.Wo = Wo; p.cv.Kreal = Kreal;
    p.cv.inv_howo = 1.0f / (float)(Ho * Wo); p.cv.inv_wo = 1.0f / (float)Wo; p.cv.inv_c = 1.0f / (float)c.C;
    p.cv.kh = general ? c.kh : c.ks; p.cv.kw = general ? c.kw : c.ks; p.cv.offy = general ? c.off_y : -pad; p.cv.offx = general ? c.off_x : -pad;
  }
  PH_CHECK_ARG(!a->col_stats || (!a->bias && a->act == PH_ACT_NONE && !a->act_in && !a->residual && !(a->drop_p > 0.f) && !a->accumulate),
               "ph_gemm_bf16: col_stats needs a plain epilogue");
  return PH_OK;
}

extern "C" int ph_gemm_grouped_bf16(const ph_gemm_args* args, int n, hipStream_t stream) {
  return ph_gemm_grouped_capped_bf16(args, n, 0, stream);
}

extern "C" int ph_gemm_grouped_capped_bf16(const ph_gemm_args* args, int n, int max_blocks, hipStream_t stream) {
  PH_CHECK_ARG(args && n >= 1 && n <= PH_GEMM_GROUP_MAX, "ph_gemm_grouped_bf16: need 1..%d problems, got %d", PH_GEMM_GROUP_MAX, n);
  double flops = 0.0, bytes = 0.0;
  double w128 = 0.0, w64 = 0.0, kt_max = 0.0;      // tile-iterations of work per tile shape
  bool kfull = true, thin = false;
  const bool conv = args[0].conv != nullptr;
  for (int i = 0; i < n; ++i) {
    PH_CHECK_ARG(args[i].trans_a == args[0].trans_a && args[i].trans_b == args[0].trans_b, "ph_gemm_grouped_bf16: mixed layouts in one group");
    PH_CHECK_ARG((args[i].conv != nullptr) == conv, "ph_gemm_grouped_bf16: implicit-GEMM conv problems cannot share a group with plain ones");
    flops += 2.0 * args[i].M * (double)args[i].N * args[i].K;
    bytes += 2.0 * ((double)args[i].M * args[i].K + (double)args[i].N * args[i].K + (double)args[i].M * args[i].N);
    const double kt = ceil_div(args[i].K, BK);
    w128 += (double)ceil_div(args[i].M, 128) * ceil_div(args[i].N, 128) * kt;
    w64 += (double)ceil_div(args[i].M, 64) * ceil_div(args[i].N, 64) * kt;
    kt_max = kt > kt_max ? kt : kt_max;
    kfull = kfull && (args[i].K % BK) == 0;
    thin = thin || args[i].M <= 64 || args[i].N <= 64;
  }
  char desc__[96];
  if (g_ph_prof_enabled) snprintf(desc__, sizeof(desc__), "gemm grouped n=%d ta=%d tb=%d K0=%d", n, args[0].trans_a, args[0].trans_b, args[0].K);
  ProfScope prof__(PH_FAM_GEMM, flops, bytes, stream, desc__);
  // 128x128 tiles (twice the arithmetic intensity per staged byte) unless the group is so small that they would leave most
  // CUs idle; long reductions always take them (measured: 16 adaptor wgrads 768x384xK=8320 ran 440 us on 64x64 tiles)
  double t128 = 0.0;
  for (int i = 0; i < n; ++i) t128 += (double)ceil_div(args[i].M, 128) * ceil_div(args[i].N, 128);
  const int BMsel = (!thin && (t128 >= 192.0 || (kt_max >= 32.0 && t128 >= 64.0))) ? 128 : 64;
  (void)w128; (void)w64;
  GroupParams g;                             // ~2.3 KB: copied into the kernel arguments by the launch
  g.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    int rc = fill_params(&args[i], g.p[i]);
    if (rc) return rc;
    g.p[i].tiles_m = ceil_div(args[i].M, BMsel); g.p[i].tiles_n = ceil_div(args[i].N, BMsel);
    g.p[i].k_tiles_per_split = ceil_div(args[i].K, BK);
    g.tile_start[i] = total;
    g.nsplit[i] = 1;
    total += g.p[i].tiles_m * g.p[i].tiles_n;
  }
  g.tile_start[n] = total;
  // ---- grouped AND split over K (round 3): weight-gradient groups with few output tiles and very long reductions -- the stems' conv
  // weight gradients, 6 experts x (2..42 tiles over up to 6272 k-tiles).  As single launches each of them split K on its own (28-92 us
  // per launch at 5-60 % fill, 28 launches per step); here every problem is cut into pieces of ~W/slots k-tiles so that the whole
  // group fills the chip once, partial sums go to the caller's workspace and ONE grouped pass folds them (plain fp32 epilogues only).
  {
    const ph_gemm_args& a0 = args[0];
    bool ok = max_blocks == 0 && a0.trans_a && a0.trans_b && a0.workspace && a0.workspace_bytes > 0;
    double W = 0.0;
    for (int i = 0; i < n && ok; ++i) {
      const ph_gemm_args& a = args[i];
      ok = a.out_f32 && !a.bias && a.act == PH_ACT_NONE && !a.pre_out && !a.act_in && !a.residual && !(a.drop_p > 0.f) && !a.col_stats &&
           a.rowmap_wo == 0 && a.workspace == a0.workspace;
      W += (double)g.p[i].tiles_m * g.p[i].tiles_n * ceil_div(a.K, BK);
    }
    const double slots = BMsel == 128 ? 512.0 : 1024.0;
    if (ok && total * 2 <= slots) {
      const int L = std::max(8, (int)ceil(W / slots));            // k-tiles per block
      int blocks = 0, nsp[PH_GEMM_GROUP_MAX];
      size_t off = 0, offs[PH_GEMM_GROUP_MAX];
      bool any = false;
      for (int i = 0; i < n; ++i) {
        const int kti = ceil_div(args[i].K, BK);
        int ns = std::min(ceil_div(kti, L), std::max(1, kti / 4));
        const int per = ceil_div(kti, std::max(1, ns));
        ns = ceil_div(kti, per);
        nsp[i] = ns; offs[i] = off;
        if (ns > 1) { any = true; off += ((size_t)ns * args[i].M * ((args[i].N + 3) / 4 * 4) * 4 + 255) / 256 * 256; }
        blocks += g.p[i].tiles_m * g.p[i].tiles_n * ns;
      }
      if (any && off <= (size_t)a0.workspace_bytes) {
        int tot = 0;
        for (int i = 0; i < n; ++i) {
          const int kti = ceil_div(args[i].K, BK);
          g.nsplit[i] = nsp[i];
          g.p[i].k_tiles_per_split = ceil_div(kti, nsp[i]);
          g.p[i].ldws = (args[i].N + 3) / 4 * 4;
          g.p[i].ws = nsp[i] > 1 ? reinterpret_cast<float*>(reinterpret_cast<char*>(a0.workspace) + offs[i]) : nullptr;
          g.tile_start[i] = tot;
          tot += g.p[i].tiles_m * g.p[i].tiles_n * nsp[i];
        }
        g.tile_start[n] = tot;
        int rc;
        rc = BMsel == 128 ? reg::launch_grouped_128(g, tot, 0, 1, 1, 0, conv ? 2 : 0, stream) : reg::launch_grouped_64(g, tot, 0, 1, 1, 0, conv ? 2 : 0, stream);
        if (rc != PH_OK) return rc;
        ReduceGroup r;
        r.n = 0; r.blk_start[0] = 0;
        for (int i = 0; i < n; ++i) {
          if (nsp[i] <= 1) continue;
          r.p[r.n] = g.p[i]; r.splits[r.n] = nsp[i];
          r.blk_start[r.n + 1] = r.blk_start[r.n] + reduce_blocks(g.p[i], false);
          ++r.n;
        }
        count_launch(PH_GEMM_CLS_SPLITK_REDUCE);
        hipLaunchKernelGGL(splitk_reduce_grouped_kernel<1>, dim3(r.blk_start[r.n]), dim3(256), 0, stream, r);
        PH_LAUNCH_CHECK("splitk_reduce_grouped_kernel");
        return PH_OK;
      }
    }
  }
  // ---- weight-gradient groups with long reductions: the 256x128 ping-pong kernel, one persistent block per CU ----
  {
    bool ok = big_mode_now() > 0 && !conv && max_blocks == 0 && args[0].trans_a && args[0].trans_b;
    int tbig = 0, kt_min = 1 << 30, kt_big = 0;
    for (int i = 0; i < n && ok; ++i) {
      const ph_gemm_args& a = args[i];
      ok = (a.K % BK) == 0 && (a.M % 8) == 0 && (a.N % 8) == 0 && a.M >= 8 && a.N >= 8 && !a.col_stats;
      tbig += ceil_div(a.M, big::BM) * ceil_div(a.N, big::BN);
      kt_min = min(kt_min, a.K / BK); kt_big = max(kt_big, a.K / BK);
    }
    if (ok && kt_min >= 32) {
      // rounds x (k loop + fixed part) of either kernel, constants from the per-shape fits: 0.67 us per k-tile for the
      // one-per-CU 256x128 block, 0.84 us per k-tile and pair of co-resident 128x128 blocks (0.5 us for a lone one)
      const double cost_big = ceil(tbig / 256.0) * (kt_big * 0.67 + 14.0);
      const int t128 = total;           // (tiles of the BMsel grid computed above; BMsel is 128 for these groups)
      const double cost_128 = BMsel == 128 ? (t128 <= 256 ? kt_big * 0.5 + 10.0 : ceil(t128 / 512.0) * (kt_big * 0.84 + 10.0)) : 1e30;
      if (cost_big < cost_128) {
        int tot = 0;
        for (int i = 0; i < n; ++i) {
          g.p[i].tiles_m = ceil_div(args[i].M, big::BM); g.p[i].tiles_n = ceil_div(args[i].N, big::BN);
          g.p[i].k_tiles_per_split = args[i].K / BK;
          g.tile_start[i] = tot;
          tot += g.p[i].tiles_m * g.p[i].tiles_n;
        }
        g.tile_start[n] = tot;
        return big::launch_grouped_wgrad(g, tot, big_mode_now() >= 7 ? 7 : 5, stream);
      }
    }
  }
  const int ta = args[0].trans_a, tb = args[0].trans_b;
  // ---- forward-shaped implicit convolutions with many rows and a long reduction (the stems' layers 2..4 and their data gradients): the
  // 256x128 LDS-DMA kernel with the gather in the DMA source address (round 4; the register-staged gather kernel ran them at ~410 TFLOP/s)
  if (conv && !ta && !tb && big_mode_now() > 1 && max_blocks == 0) {
    bool ok = zero_bias_ptr() != nullptr;                    // the gather's padding lanes read the zero page
    for (int i = 0; i < n && ok; ++i) {
      const ph_gemm_args& a = args[i];
      ok = a.M >= big::BM && a.K >= 2 * BK && (a.N % 8) == 0 && a.N >= 8 && (a.ldb % 8) == 0 && a.split_k <= 1 &&
           (int64_t)a.conv->B * a.conv->H * a.conv->W * a.conv->C < (1ll << 31);          // 32-bit element offsets in the gather
    }
    if (ok) {
      int blocks = 0;
      for (int i = 0; i < n; ++i) {
        g.p[i].tiles_m = ceil_div(args[i].M, big::BM); g.p[i].tiles_n = ceil_div(args[i].N, big::BN);
        g.p[i].k_tiles_per_split = ceil_div(args[i].K, BK);
        g.p[i].ws = nullptr; g.p[i].ldws = 0;
        g.tile_start[i] = blocks;
        blocks += big_xcd_grid(g.p[i].tiles_m, g.p[i].tiles_n);          // (XCD shares, see big_tile: the blocks beyond a share exit)
      }
      g.tile_start[n] = blocks;
      return big::launch_grouped_conv(g, blocks, big_mode_now() >= 7 ? 7 : 5, stream);
    }
  }
  if (conv) {               // gathered operand: forward (NN, A = im2col view) or weight gradient (TT, B = im2col view); no prefetch ring
    PH_CHECK_ARG(ta == tb, "ph_gemm_grouped_bf16: conv gather needs the NN or the TT layout");
    if (!ta) {
      // gathered A operand through the register prefetch ring as well (round 3): the gather loads are unconditional (clamped address +
      // select), so the ring's straight-line bookkeeping holds for any K
      return BMsel == 128 ? reg::launch_grouped_128(g, total, max_blocks, 0, 0, 1, 1, stream)
                          : reg::launch_grouped_64(g, total, max_blocks, 0, 0, 1, 1, stream);
    }
    return BMsel == 128 ? reg::launch_grouped_128(g, total, max_blocks, 1, 1, 0, 2, stream) : reg::launch_grouped_64(g, total, max_blocks, 1, 1, 0, 2, stream);
  }
  if (BMsel == 128) {
    // prefetch ring for the [K,M] x [K,N] weight-gradient groups too (round 3: step -0.07 / -0.17 ms in two same-box pairs)
    return reg::launch_grouped_128(g, total, max_blocks, ta, tb, kfull ? 1 : 0, 0, stream);
  }
  return reg::launch_grouped_64(g, total, max_blocks, ta, tb, kfull ? 1 : 0, 0, stream);
}

extern "C" int ph_gemm_bf16(const ph_gemm_args* a, hipStream_t stream) {
  PH_CHECK_ARG(a && a->A && a->B && a->C, "ph_gemm_bf16: null pointer");
  // ---- tail split (round 3): a launch whose tile count is a little above a whole number of block rounds pays a full extra round for
  // a handful of tiles (c_fc 8320 x 3072: 65 x 24 = 1560 tiles of 128x128 on 512 block slots = 3.05 rounds -> 4 tile times, ~95 us).
  // The rows that cause the overshoot (the last, partial row panels) are cut off and computed by a second, small launch:
  // 8192 x 3072 = exactly 3 rounds (+ 128 x 3072 on 64x64 tiles, ~10 us beside an otherwise idle chip).
  {   // (before the profiling scope of this call: the two halves are profiled as two launches)
    const int64_t tn128 = ceil_div(a->N, 128), tm128 = ceil_div(a->M, 128);
    const int64_t tiles = tm128 * tn128, slots = 512;
    if (!a->trans_a && !a->conv && !a->col_stats && !(a->drop_p > 0.0f) && a->split_k <= 0 && a->rowmap_wo == 0 && tiles > slots && (a->K % BK) == 0) {
      const int64_t over = tiles % slots;
      const int64_t panels_main = (tiles / slots) * slots / tn128;              // row panels that fit the whole rounds
      const int64_t m_main = panels_main * 128, m_rem = a->M - m_main;
      // worth it when the overshoot is small (<= 12 % of a round) and the cut-off part is itself a small problem
      if (over > 0 && over * 100 <= slots * 12 && m_rem > 0 && m_rem <= 512 && (m_main % 256) == 0 && m_main >= 2048) {
        ph_gemm_args b = *a, c = *a;
        b.M = (int)m_main;
        c.M = (int)m_rem;
        c.A = (const char*)a->A + (size_t)m_main * a->lda * 2;
        c.C = (char*)a->C + (size_t)m_main * a->ldc * (a->out_f32 ? 4 : 2);
        if (a->pre_out) c.pre_out = (char*)a->pre_out + (size_t)m_main * a->ldc * 2;
        if (a->act_in) c.act_in = (const char*)a->act_in + (size_t)m_main * a->ld_act * 2;
        if (a->residual) c.residual = (const char*)a->residual + (size_t)m_main * a->ldr * (a->residual_f32 ? 4 : 2);
        int rc = ph_gemm_bf16(&b, stream);
        if (rc != PH_OK) return rc;
        return ph_gemm_bf16(&c, stream);
      }
    }
  }
  char desc__[96];
  if (g_ph_prof_enabled) snprintf(desc__, sizeof(desc__), "gemm M=%d N=%d K=%d ta=%d tb=%d f32=%d acc=%d", a->M, a->N, a->K, a->trans_a, a->trans_b, a->out_f32, a->accumulate);
  ProfScope prof__(PH_FAM_GEMM, 2.0 * a->M * (double)a->N * a->K, 2.0 * ((double)a->M * a->K + (double)a->N * a->K + (double)a->M * a->N), stream, desc__);
  if (a->conv && !a->trans_a && !a->trans_b) return ph_gemm_grouped_capped_bf16(a, 1, 0, stream);     // implicit-GEMM forward: grouped kernel
  GemmParams p;
  {
    int rc = fill_params(a, p);
    if (rc) return rc;
  }
  PH_CHECK_ARG(!a->col_stats || a->split_k <= 1, "ph_gemm_bf16: col_stats cannot be combined with split-K");

  // ---- big-tile LDS-DMA kernel: forward-shaped (both operands K-contiguous) GEMMs with enough 256x128 tiles for the chip ----
  {
    // g_big_mode (PH_GEMM_BIG / ph_gemm_tuning): 0 = off, 1 = plain main loop (round-2 first version), 5 = ping-pong main loop (the
    // two waves of a SIMD alternate read and MFMA phases; retired in round 6, now = 6), 6 = ping-pong with the LEAN tail (no surplus DMA, no drain;
    // round 4), 7 = 6 with the DMA requests spread over the M phase (default since round 6)
    const int big_mode = big_mode_now(), big_min_tiles = big_min_tiles_now();
    const int64_t tb = (int64_t)ceil_div(a->M, big::BM) * ceil_div(a->N, big::BN);
    // block rounds of either kernel (constants from the per-shape fits, us): a 256x128 block alone on its CU, a pair of co-resident
    // 128x128 blocks, a lone 128x128 block (what the last, half-empty round of that kernel is made of -- the reason a tile count just
    // above a multiple of 256 favours it: 41 x 8 tiles of 256x128 are two full rounds, 81 x 8 of 128x128 one pair round + one lone round)
    bool big_cheaper = true;
    if (big_min_tiles > 1) {
      const double ktd = (double)(a->K / BK);
      const int64_t t128b = (int64_t)ceil_div(a->M, 128) * ceil_div(a->N, 128), rem = t128b % 512;
      const double cost_big = (double)ceil_div64(tb, 256) * (ktd * 0.67 + 12.0);
      const double cost_128 = (double)(t128b / 512) * (ktd * 0.84 + 12.0) + (rem > 256 ? ktd * 0.84 + 12.0 : (rem > 0 ? ktd * 0.5 + 12.0 : 0.0));
      big_cheaper = cost_big < cost_128 * (a->trans_b ? 1.0 : 1.02);      // ties (two full rounds vs pair + lone round): measured in favour of
                                                                           // the big kernel for [N,K] B (stem 25088x384x1728), against it for [K,N] B (LARGE dgrads)
    }
    if (big_mode > 0 && !a->conv && !a->col_stats && !a->trans_a && (a->K % BK) == 0 && a->K >= 2 * BK && a->split_k <= 0 && a->M >= big::BM &&
        (a->N % 8) == 0 && a->N >= 8 && tb >= big_min_tiles && (tb >= 192 || a->K >= 32 * BK || big_min_tiles <= 1) && big_cheaper) {
      // Forward-shaped (B = [N][K]) and dgrad-shaped (B = [K][N], trans_b) problems alike.  Isolated (tools/big_probe.py,
      // profiles/r2_ab_big_tile_gemm.txt) the ping-pong kernel beats the 128x128 register-staged kernel on every shape of the
      // step; inside the step (rocprofv3 per-grid durations, profiles/r2_gemm_by_grid.txt) only the launches with N <= 1024 or a
      // long k loop keep the gain -- the wide-N, K = 768 launches are write-out bound there and the 2-blocks-per-CU kernel overlaps
      // one tile's write-out with the other's main loop -- so those stay on the 128x128 kernel.
      p.tiles_m = ceil_div(a->M, big::BM); p.tiles_n = ceil_div(a->N, big::BN);
      p.k_tiles_per_split = a->K / BK;
      p.ws = nullptr; p.ldws = 0;
      return big::launch_single(p, big_mode == 1 ? 0 : (big_mode == 6 ? 5 : (big_mode >= 7 && big_mode <= 11 ? big_mode : 4)), false, a->trans_b != 0, stream);
    }
    // weight-gradient layout (A = [K][M], B = [K][N]) with a long reduction: same kernel, both operands through the transposing reads
    if (big_mode > 0 && !a->conv && !a->col_stats && a->trans_a && a->trans_b && (a->K % BK) == 0 && a->K >= 32 * BK && a->split_k <= 0 &&
        a->M >= big::BM && (a->M % 8) == 0 && (a->N % 8) == 0 && a->N >= 8 && tb >= 64 && big_cheaper) {
      p.tiles_m = ceil_div(a->M, big::BM); p.tiles_n = ceil_div(a->N, big::BN);
      p.k_tiles_per_split = a->K / BK;
      p.ws = nullptr; p.ldws = 0;
      return big::launch_single(p, big_mode >= 7 ? 7 : 5, true, true, stream);
    }
  }

  // ---- tile shape and split-K selection -------------------------------------------------------------------------
  // 128x128 tiles when they fill the chip; otherwise split the K loop (partials -> workspace -> reduce+epilogue) so
  // that >= ~2 blocks per CU are in flight; 64x64 tiles for genuinely small outputs.
  const int kt = ceil_div(a->K, BK);
  const int ldws = (a->N + 3) / 4 * 4;
  const bool plain_acc = !a->bias && !a->pre_out && !a->act_in && a->act == PH_ACT_NONE && !a->residual && !(a->drop_p > 0.0f) &&
                         a->out_f32 && a->accumulate;
  auto ws_fits = [&](int sp) { return a->workspace && (int64_t)sp * a->M * ldws * 4 <= a->workspace_bytes; };
  const int64_t t128 = (int64_t)ceil_div(a->M, 128) * ceil_div(a->N, 128), t64 = (int64_t)ceil_div(a->M, 64) * ceil_div(a->N, 64);
  // pick (tile, split) by a small cost model calibrated on the MI355X microbenchmarks (profiles/r1_*): a k-tile
  // iteration costs ~1.5 us for co-resident 128x128 blocks (2 per CU) and ~0.5 us for 64x64 blocks (4 per CU);
  // a split adds the reduce launch (~4 us) plus splits*M*N*4 bytes of partial traffic.
  int BM = 128, BN = 0, splits = 1;      // BN = 0: square tile
  if (a->split_k > 0) {
    splits = a->split_k;
    if (a->M <= 64 || a->N <= 64 || t128 * splits < 128) BM = 64;
  } else {
    double best = 1e30;
    const int cand[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256};   // (the stems' first-layer weight gradients:
                                                                                       //  2 output tiles, 6272 k-tiles -- 48 splits left them at 94 us)
    for (int bm = 64; bm <= 128; bm += 64) {
      if (bm == 128 && (a->M <= 64 || a->N <= 64)) continue;
      // round-1 slope probe on MI355X: 8320x768 NT, K 768 -> 3072: 1.07 us per k-tile for one wave of co-resident
      // 128x128 blocks (2 per CU), 0.83 us per k-tile and wave of 64x64 blocks (4 per CU); ~4 us fixed per block wave.
      const double tiles = (double)(bm == 128 ? t128 : t64), cap = bm == 128 ? 512.0 : 1024.0, t_iter = bm == 128 ? 1.07 : 0.83;
      for (int sp : cand) {
        if (sp > 1 && (sp > kt / 2 || !(ws_fits(sp) || plain_acc))) continue;
        double waves = ceil(tiles * sp / cap);
        double cost = waves * (ceil((double)kt / sp) * t_iter + 4.0);
        if (sp > 1) cost += 8.0 + (double)sp * a->M * ldws * 4.0 / 3.0e6;   // reduce launch: ~7 us measured (rocprof) + partial traffic
        if (cost < best) { best = cost; BM = bm; splits = sp; }
      }
    }
  }
  if (splits > kt) splits = kt;
  // N = 64 mod 128 and narrow (the stems' N = 192 convs): 128x64 tiles cover N exactly instead of wasting half a column tile
  // (100352x192x864: 71 -> 66 us); on every other shape the narrower wave tile (one B fragment per two MFMAs) loses
  if (BN == 0 && BM == 128 && splits == 1 && a->N % 128 == 64 && a->N <= 192) BN = 64;
  if (BN == 0) BN = BM;
  p.tiles_m = ceil_div(a->M, BM); p.tiles_n = ceil_div(a->N, BN);
  p.k_tiles_per_split = ceil_div(kt, splits);
  splits = ceil_div(kt, p.k_tiles_per_split);
  p.ws = nullptr; p.ldws = ldws;
  bool deferred = false;
  if (splits > 1) {
    if (ws_fits(splits)) {
      p.ws = (float*)a->workspace;
      if (a->defer_reduce) {              // partial sums of consecutive deferred calls sit one after the other in the caller's workspace
        std::lock_guard<std::mutex> lock(g_defer_mu);
        const size_t need = ((size_t)splits * a->M * ldws * 4 + 255) / 256 * 256;
        if (g_defer.empty()) g_defer_cursor = 0;
        if (!g_defer.empty() && (g_defer_ws != a->workspace || g_defer_cursor + need > (size_t)a->workspace_bytes)) {
          int rcf = flush_deferred_locked(stream);
          if (rcf != PH_OK) return rcf;
        }
        p.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(a->workspace) + g_defer_cursor);
        g_defer_cursor += need;
        g_defer_ws = a->workspace;
        deferred = true;
      }
    } else PH_CHECK_ARG(plain_acc, "ph_gemm_bf16: split_k > 1 without workspace needs out_f32 + accumulate and no fused epilogue");
  }
  {
    // 64x64-tile launches that leave CUs with a single block (the decoder's M = 960 rows): split the k loop inside the block instead
    // of over gridDim.z + a reduce launch.  also instead of a workspace split-K of a 64x64 launch
    const bool plain64 = BM == 64 && BN == 64 && !a->conv && !a->col_stats && !a->trans_a && (a->K % BK) == 0 && kt >= 8 && a->split_k <= 0;
    if (plain64 && t64 <= 512) {
      p.tiles_m = ceil_div(a->M, 64); p.tiles_n = ceil_div(a->N, 64);
      p.k_tiles_per_split = kt; p.ws = nullptr;
      return reg::launch_ks2(p, a->trans_b, stream);
    }
  }
  int rc = BM == 64 ? reg::launch_single_64(p, a->trans_a, a->trans_b, splits, stream) : reg::launch_single_128(p, BN, a->trans_a, a->trans_b, splits, stream);
  if (rc != PH_OK || !p.ws) return rc;
  const int64_t vecs = (int64_t)a->M * ((a->N + 3) / 4);
  if (deferred) {                                         // the fold pass joins the queue (the partials stay where they are until the flush)
    std::lock_guard<std::mutex> lock(g_defer_mu);
    g_defer.push_back(DeferredReduce{p, splits, splits >= 32 && vecs <= 16384});
    return PH_OK;
  }
  count_launch(PH_GEMM_CLS_SPLITK_REDUCE);
  if (splits >= 32 && vecs <= 16384) {                    // many splits of a small output: 16 lanes per output vector
    int grid = (int)min((int64_t)2048, ceil_div64(vecs * 16, 256));
    hipLaunchKernelGGL(splitk_reduce_kernel<16>, dim3(grid), dim3(256), 0, stream, p, splits);
  } else {
    int grid = (int)min((int64_t)2048, ceil_div64(vecs, 256));
    hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3(grid), dim3(256), 0, stream, p, splits);
  }
  PH_LAUNCH_CHECK("splitk_reduce_kernel");
  return PH_OK;
}
