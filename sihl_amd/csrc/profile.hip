#include "profile.h"
#include <vector>
#include <mutex>

namespace {
struct Rec { hipEvent_t a, b; int slot, dtype; double flops, bytes; };
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;  // events are created once and recycled: hipEventCreate per launch cost ~2 ms/step
size_t g_used = 0;
std::mutex g_mu;
Rec g_cur;
bool g_open = false;

bool take_event(hipEvent_t* e) {
  if (g_used == g_pool.size()) {
    hipEvent_t n;
    if (hipEventCreate(&n) != hipSuccess) return false;
    g_pool.push_back(n);
  }
  *e = g_pool[g_used++];
  return true;
}
}  // namespace

void sihl_prof_begin(int slot, int dtype, double flops, double bytes, hipStream_t stream) {
  if (!g_on) return;
  std::lock_guard<std::mutex> lk(g_mu);
  g_cur.slot = slot; g_cur.dtype = dtype; g_cur.flops = flops; g_cur.bytes = bytes;
  if (!take_event(&g_cur.a) || !take_event(&g_cur.b)) return;
  (void)hipEventRecord(g_cur.a, stream);
  g_open = true;
}

void sihl_prof_end(hipStream_t stream) {
  if (!g_on || !g_open) return;
  std::lock_guard<std::mutex> lk(g_mu);
  (void)hipEventRecord(g_cur.b, stream);
  g_recs.push_back(g_cur);
  g_open = false;
}

extern "C" {

// Turn per-launch event timing on/off; turning it on clears earlier records.
int sihl_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (on) {  // earlier records are dropped; their events go back to the pool
    g_recs.clear();
    g_used = 0;
  }
  g_on = on != 0;
  return 0;
}

// Sum the records of (slot, dtype): launches, total milliseconds, total algorithmic flops and bytes.
// Synchronises on the recorded events.
int sihl_profile_collect(int slot, int dtype, long* launches, double* total_ms, double* total_flops,
                         double* total_bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  long n = 0; double ms = 0, fl = 0, by = 0;
  for (auto& r : g_recs) {
    if (r.slot != slot || r.dtype != dtype) continue;
    hipError_t e = hipEventSynchronize(r.b);
    if (e != hipSuccess) return (int)e;
    float t = 0.f;
    e = hipEventElapsedTime(&t, r.a, r.b);
    if (e != hipSuccess) return (int)e;
    ++n; ms += t; fl += r.flops; by += r.bytes;
  }
  if (launches) *launches = n;
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (total_bytes) *total_bytes = by;
  return 0;
}

// Per-launch records of (slot, dtype) in launch order: out[3*i + {0,1,2}] = milliseconds, flops, bytes.
// Returns the number of records (which may exceed cap; only cap are written), negative on error.
long sihl_profile_records(int slot, int dtype, double* out, long cap) {
  std::lock_guard<std::mutex> lk(g_mu);
  long n = 0;
  for (auto& r : g_recs) {
    if (r.slot != slot || r.dtype != dtype) continue;
    if (out && n < cap) {
      if (hipEventSynchronize(r.b) != hipSuccess) return -1;
      float t = 0.f;
      if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return -1;
      out[3 * n] = t; out[3 * n + 1] = r.flops; out[3 * n + 2] = r.bytes;
    }
    ++n;
  }
  return n;
}

}  // extern "C"
