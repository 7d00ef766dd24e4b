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
extern "C" const char* ph_last_error(void) { return g_last_error.c_str(); }

// ---- kernel-family timing -----------------------------------------------------------------------------------------
#include <vector>
int g_ph_prof_enabled = 0;
namespace {
struct ProfRec { int fam; double flops, bytes; hipEvent_t a, b; std::string desc; };
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e; (void)hipEventCreate(&e); return e;
}
}  // namespace
void ph_prof_begin(int family, double flops, double bytes, hipStream_t s, const char* desc) {
  ProfRec r{family, flops, bytes, get_event(), get_event(), desc ? desc : ""};
  (void)hipEventRecord(r.a, s);
  g_recs.push_back(r);
}
void ph_prof_end(hipStream_t s) { (void)hipEventRecord(g_recs.back().b, s); }

extern "C" int ph_prof_enable(int on) { g_ph_prof_enabled = on; return PH_OK; }
/* writes one CSV line per recorded call (family, ms, flops, desc) WITHOUT clearing: call before ph_prof_collect */
extern "C" int ph_prof_dump(const char* path) {
  (void)hipDeviceSynchronize();
  FILE* f = fopen(path, "w");
  if (!f) return ph_fail(PH_ERR_BAD_ARG, "ph_prof_dump: cannot open %s", path);
  for (auto& r : g_recs) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.a, r.b);
    fprintf(f, "%d,%.4f,%.0f,%s\n", r.fam, ms, r.flops, r.desc.c_str());
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
