// Common device/host helpers for the prismer_hip library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/prismer_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define PH_WAVE 64

// ---- error plumbing -----------------------------------------------------------------------------
void ph_set_error(const std::string& msg);
int ph_fail(int code, const char* fmt, ...);
#define PH_CHECK_ARG(cond, ...)                                   \
  do {                                                            \
    if (!(cond)) return ph_fail(PH_ERR_BAD_ARG, __VA_ARGS__);     \
  } while (0)
#define PH_LAUNCH_CHECK(name)                                                          \
  do {                                                                                 \
    hipError_t e__ = hipGetLastError();                                                \
    if (e__ != hipSuccess) return ph_fail(PH_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
  } while (0)

// ---- optional per-family kernel timing (HIP events on the launch stream; off by default, never inside graph capture)
enum { PH_FAM_GEMM = 0, PH_FAM_LAYERNORM, PH_FAM_ATTN_FWD, PH_FAM_ATTN_BWD, PH_FAM_FRONTEND, PH_FAM_EMBED_CE, PH_FAM_OPTIM, PH_FAM_MISC,
       PH_FAM_COUNT };
extern int g_ph_prof_enabled;
extern int g_ph_prof_last_cls;      // GEMM kernel class of the profiled call in flight (core.hip)
void ph_prof_begin(int family, double flops, double bytes, hipStream_t s, const char* desc = nullptr);
void ph_prof_end(hipStream_t s);
struct ProfScope {
  hipStream_t s; bool on;
  ProfScope(int family, double flops, double bytes, hipStream_t st, const char* desc = nullptr) : s(st), on(g_ph_prof_enabled != 0) { if (on) ph_prof_begin(family, flops, bytes, st, desc); }
  ~ProfScope() { if (on) ph_prof_end(s); }
};

// ---- diagnostics build only (-DPH_TIMELINE, tools/timeline_probe.py): s_memtime stamps of one wave per thread group, kept in SGPRs and
// stored once at the end of the block; slot meaning is documented in the probe.  Never defined in the product build.
#ifdef PH_TIMELINE
#define PH_TL_SLOTS 16
#define PH_TL_BLOCKS 4096
static __device__ unsigned long long g_tl[PH_TL_BLOCKS * 2 * PH_TL_SLOTS];
struct TlStamps { unsigned long long t[PH_TL_SLOTS]; };
#define PH_TL_DECL TlStamps tl__ = {}; tl__.t[10] = __builtin_amdgcn_s_memrealtime()
#define PH_TL(s) (tl__.t[(s)] = __builtin_amdgcn_s_memtime())
#define PH_TL_ARG tl__,
#define PH_TL_PARAM TlStamps& tl__,
#define PH_TL_FLUSH(blk, half, who)                                                                              \
  do {                                                                                                           \
    if ((who) && (blk) < PH_TL_BLOCKS) {                                                                         \
      tl__.t[11] = __builtin_amdgcn_s_memrealtime();                                                                \
      tl__.t[PH_TL_SLOTS - 1] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492); \
      _Pragma("unroll") for (int i__ = 0; i__ < PH_TL_SLOTS; ++i__) g_tl[((blk) * 2 + (half)) * PH_TL_SLOTS + i__] = tl__.t[i__]; \
    }                                                                                                            \
  } while (0)
#else
#define PH_TL_DECL
#define PH_TL(s)
#define PH_TL_ARG
#define PH_TL_PARAM
#define PH_TL_FLUSH(blk, half, who)
#endif

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- bf16 <-> f32 -------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }   // v_cvt_pk_bf16_f32: RNE

// ---- wave reductions (64 lanes) -----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- activations --------------------------------------------------------------------------------
// reference: QuickGELU utils.py:23-25, SquaredReLU utils.py:28-30, erf-GELU roberta.py:164,423
// Activation math runs inside GEMM epilogues (64 outputs per thread per tile): exact IEEE division / erff there cost more
// VALU time than the tile's global stores.  v_rcp_f32 / v_exp_f32 are 1-ulp instructions; erf uses Abramowitz-Stegun
// 7.1.26 (|abs err| <= 1.5e-7), far below the bf16 rounding every result of these functions goes through.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sigmoid(float x) { return fast_rcp(1.0f + __expf(-x)); }
// returns erf(|u|) given e = exp(-u*u)
__device__ __forceinline__ float erf_abs_from_exp(float au, float e) {
  float t = fast_rcp(fmaf(0.3275911f, au, 1.0f));
  float poly = fmaf(fmaf(fmaf(fmaf(1.061405429f, t, -1.453152027f), t, 1.421413741f), t, -0.284496736f), t, 0.254829592f) * t;
  return 1.0f - poly * e;
}
__device__ __forceinline__ float act_fwd(int act, float x) {
  switch (act) {
    case PH_ACT_QUICKGELU: return x * fast_sigmoid(1.702f * x);
    case PH_ACT_RELU2: { float r = fmaxf(x, 0.0f); return r * r; }
    case PH_ACT_GELU: {
      float u = x * 0.70710678118654752f;
      float er = erf_abs_from_exp(fabsf(u), __expf(-u * u));
      return 0.5f * x * (1.0f + copysignf(er, x));
    }
    case PH_ACT_RELU: return fmaxf(x, 0.0f);
    default: return x;
  }
}
__device__ __forceinline__ float act_grad(int act, float x) {
  switch (act) {
    case PH_ACT_QUICKGELU: {
      float s = fast_sigmoid(1.702f * x);
      return s * (1.0f + 1.702f * x * (1.0f - s));
    }
    case PH_ACT_RELU2: return 2.0f * fmaxf(x, 0.0f);
    case PH_ACT_GELU: {
      float u = x * 0.70710678118654752f;
      float e = __expf(-u * u);                                   // = exp(-x^2 / 2)
      float c = 0.5f * (1.0f + copysignf(erf_abs_from_exp(fabsf(u), e), x));
      return c + x * 0.3989422804014327f * e;
    }
    case PH_ACT_RELU: return x > 0.0f ? 1.0f : 0.0f;
    case PH_ACT_SAVED_GRAD: return x;                               // the forward stored act'(x) itself (ph_gemm_args.pre_grad)
    default: return 1.0f;
  }
}

// value and derivative together (forward epilogue with ph_gemm_args.pre_grad): the transcendental part is shared
__device__ __forceinline__ void act_fwd_grad(int act, float x, float& y, float& g) {
  switch (act) {
    case PH_ACT_QUICKGELU: {
      float s = fast_sigmoid(1.702f * x);
      y = x * s;
      g = s * (1.0f + 1.702f * x * (1.0f - s));
      return;
    }
    case PH_ACT_RELU2: { float r = fmaxf(x, 0.0f); y = r * r; g = 2.0f * r; return; }
    case PH_ACT_GELU: {
      float u = x * 0.70710678118654752f;
      float e = __expf(-u * u);
      float c = 0.5f * (1.0f + copysignf(erf_abs_from_exp(fabsf(u), e), x));
      y = x * c;
      g = c + x * 0.3989422804014327f * e;
      return;
    }
    case PH_ACT_RELU: y = fmaxf(x, 0.0f); g = x > 0.0f ? 1.0f : 0.0f; return;
    default: y = x; g = 1.0f; return;
  }
}

// ---- Philox4x32-7 (dropout masks are a pure function of (seed, stream, element index)) -------------
// Seven rounds (round 4; ten before): Philox4x32 passes BigCrush from 7 rounds on (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3",
// SC'11, table 2; 10 is Random123's safety margin), and a round is four quarter-rate 32-bit multiplies -- ~90 cycles per wave.  At ten rounds the
// generator was 85 % of the decoder's attention kernels (one call per 4 keys and query: 3500 of ~4000 cycles per 64-key tile; timeline:
// 1.9 us per tile for 30 live queries).  tests/util.py mirrors the function; the masks are never compared with the reference's (its generator
// is torch's, a different stream), only their statistics and their forward / backward consistency are tested.
#ifndef PH_PHILOX_ROUNDS_N
#define PH_PHILOX_ROUNDS_N 7
#endif
constexpr int PH_PHILOX_ROUNDS = PH_PHILOX_ROUNDS_N;      // (-DPH_PHILOX_ROUNDS_N=10: the A/B build of profiles/r4_ab_philox_rounds.txt)
__device__ __forceinline__ u32x4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < PH_PHILOX_ROUNDS; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  u32x4 o = {c0, c1, c2, c3};
  return o;
}
// keep-mask for 4 consecutive elements whose first linear index is idx (idx % 4 == 0).
// keep[i] <=> u_i >= p  with u_i = top 24 bits / 2^24.
struct DropCtx {
  uint32_t k0, k1, stream, thr;   // thr = p * 2^24
  float scale;                    // 1/(1-p)
};
__device__ __forceinline__ DropCtx make_drop(const uint64_t* seed_ptr, uint32_t stream, float p) {
  DropCtx d;
  uint64_t s = seed_ptr ? *seed_ptr : 0ull;
  d.k0 = (uint32_t)s; d.k1 = (uint32_t)(s >> 32); d.stream = stream;
  d.thr = (uint32_t)(p * 16777216.0f);
  d.scale = 1.0f / (1.0f - p);
  return d;
}
__device__ __forceinline__ u32x4 drop_rand4(const DropCtx& d, uint64_t idx4) {   // idx4 = element index / 4
  return philox4x32((uint32_t)idx4, (uint32_t)(idx4 >> 32), d.stream, 0x5eedu, d.k0, d.k1);
}
__device__ __forceinline__ float drop_apply(const DropCtx& d, uint32_t r, float v) {
  return ((r >> 8) >= d.thr) ? v * d.scale : 0.0f;
}
