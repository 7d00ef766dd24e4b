// Error plumbing + version for libprismer_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

static thread_local std::string g_last_error;

void ph_set_error(const std::string& msg) { g_last_error = msg; }

int ph_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

extern "C" int ph_version(void) { return PH_VERSION; }

extern "C" int ph_layernorm_bwd_blocks(int M);
extern "C" int64_t ph_query_workspace(int op, const int64_t* dims, int ndims) {
  if (!dims) return -1;
  switch (op) {
    case PH_WS_GEMM_SPLITK: {
      if (ndims != 3 || dims[0] <= 0 || dims[1] <= 0 || dims[2] <= 0) return -1;
      const int64_t kt = (dims[2] + 63) / 64, splits = kt / 2 < 256 ? (kt / 2 < 1 ? 1 : kt / 2) : 256;     // a split holds >= 2 k-tiles
      return splits <= 1 ? 0 : splits * dims[0] * ((dims[1] + 3) / 4 * 4) * 4;
    }
    case PH_WS_LAYERNORM_BWD:
      if (ndims != 2 || dims[0] <= 0 || dims[1] <= 0) return -1;
      return (int64_t)ph_layernorm_bwd_blocks((int)dims[0]) * 2 * dims[1] * 4;
    case PH_WS_ATTENTION_BWD:
      if (ndims != 3 || dims[0] <= 0 || dims[1] <= 0 || dims[2] <= 0) return -1;
      return dims[0] * dims[1] * dims[2] * 4;
    case PH_WS_CONV_COLSTATS:
      if (ndims != 1 || dims[0] <= 0) return -1;
      return (int64_t)PH_COLSTAT_SLABS * 2 * dims[0] * 8;
    default:
      return -1;
  }
}
extern "C" const char* ph_last_error(void) { return g_last_error.c_str(); }

// ---- kernel-family timing -----------------------------------------------------------------------------------------
#include <vector>
int g_ph_prof_enabled = 0;
namespace {
struct ProfRec { int fam; double flops, bytes; hipEvent_t a, b; std::string desc; int cls; };     // cls: GEMM kernel class of the launch (count_launch), -1 otherwise
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e; (void)hipEventCreate(&e); return e;
}
}  // namespace
int g_ph_prof_last_cls = -1;        // set by the GEMM launchers (gemm_common.h count_launch): which kernel class the call in flight ended up on
void ph_prof_begin(int family, double flops, double bytes, hipStream_t s, const char* desc) {
  ProfRec r{family, flops, bytes, get_event(), get_event(), desc ? desc : "", -1};
  (void)hipEventRecord(r.a, s);
  g_ph_prof_last_cls = -1;
  g_recs.push_back(r);
}
void ph_prof_end(hipStream_t s) { (void)hipEventRecord(g_recs.back().b, s); g_recs.back().cls = g_ph_prof_last_cls; }

extern "C" int ph_prof_enable(int on) { g_ph_prof_enabled = on; return PH_OK; }
/* writes one CSV line per recorded call (family, ms, flops, desc, GEMM kernel class or -1) WITHOUT clearing: call before ph_prof_collect */
extern "C" int ph_prof_dump(const char* path) {
  (void)hipDeviceSynchronize();
  FILE* f = fopen(path, "w");
  if (!f) return ph_fail(PH_ERR_BAD_ARG, "ph_prof_dump: cannot open %s", path);
  for (auto& r : g_recs) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.a, r.b);
    fprintf(f, "%d,%.4f,%.0f,%s,%d\n", r.fam, ms, r.flops, r.desc.c_str(), r.cls);
  }
  fclose(f);
  return PH_OK;
}
/* synchronises the device, sums (ms, flops, bytes, launches) per family into out[PH_FAM_COUNT][4], clears the records */
extern "C" int ph_prof_collect(double* out) {
  (void)hipDeviceSynchronize();
  for (int i = 0; i < PH_FAM_COUNT * 4; ++i) out[i] = 0.0;
  for (auto& r : g_recs) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.a, r.b);
    out[r.fam * 4 + 0] += ms; out[r.fam * 4 + 1] += r.flops; out[r.fam * 4 + 2] += r.bytes; out[r.fam * 4 + 3] += 1.0;
    g_pool.push_back(r.a); g_pool.push_back(r.b);
  }
  g_recs.clear();
  return PH_OK;
}
