// Optional per-kernel timing with HIP events on the launch stream (diagnostics for bench.py; off by default).
// This is the only process-global state in the library and it never changes results.
#include "common.h"
#include "profile.h"

namespace {
struct Slot {
  hipEvent_t e0, e1;
};
struct Prof {
  int cap = 0;
  unsigned mask = ~0u;
  int n[GEOA3_PROF_TAGS] = {0};
  Slot* slots[GEOA3_PROF_TAGS] = {nullptr};
} g_prof;
}  // namespace

bool geoa3_prof_on() { return g_prof.cap > 0; }
bool geoa3_prof_tag_on(int tag) { return g_prof.cap > 0 && tag >= 0 && tag < GEOA3_PROF_TAGS && ((g_prof.mask >> tag) & 1u); }

void geoa3_prof_begin(int tag, hipStream_t s) {
  if (g_prof.cap <= 0 || tag < 0 || tag >= GEOA3_PROF_TAGS || g_prof.n[tag] >= g_prof.cap) return;
  if (!((g_prof.mask >> tag) & 1u)) return;
  (void)hipEventRecord(g_prof.slots[tag][g_prof.n[tag]].e0, s);
}

void geoa3_prof_end(int tag, hipStream_t s) {
  if (g_prof.cap <= 0 || tag < 0 || tag >= GEOA3_PROF_TAGS || g_prof.n[tag] >= g_prof.cap) return;
  if (!((g_prof.mask >> tag) & 1u)) return;
  (void)hipEventRecord(g_prof.slots[tag][g_prof.n[tag]].e1, s);
  g_prof.n[tag]++;
}

extern "C" int geoa3_profile_enable(int capacity) {
  for (int t = 0; t < GEOA3_PROF_TAGS; ++t) {
    if (g_prof.slots[t]) {
      for (int i = 0; i < g_prof.cap; ++i) {
        (void)hipEventDestroy(g_prof.slots[t][i].e0);
        (void)hipEventDestroy(g_prof.slots[t][i].e1);
      }
      delete[] g_prof.slots[t];
      g_prof.slots[t] = nullptr;
    }
    g_prof.n[t] = 0;
  }
  g_prof.cap = 0;
  if (capacity <= 0) return GEOA3_OK;
  for (int t = 0; t < GEOA3_PROF_TAGS; ++t) {
    g_prof.slots[t] = new Slot[capacity];
    for (int i = 0; i < capacity; ++i) {
      if (hipEventCreate(&g_prof.slots[t][i].e0) != hipSuccess) return GEOA3_ELAUNCH;
      if (hipEventCreate(&g_prof.slots[t][i].e1) != hipSuccess) return GEOA3_ELAUNCH;
    }
  }
  g_prof.cap = capacity;
  return GEOA3_OK;
}

extern "C" int geoa3_profile_select(unsigned mask) {
  g_prof.mask = mask;
  return GEOA3_OK;
}

extern "C" int geoa3_profile_read(int tag, float* ms_host, int cap) {
  if (tag < 0 || tag >= GEOA3_PROF_TAGS || !ms_host) return GEOA3_EINVAL;
  const int n = g_prof.n[tag] < cap ? g_prof.n[tag] : cap;
  for (int i = 0; i < n; ++i) {
    if (hipEventSynchronize(g_prof.slots[tag][i].e1) != hipSuccess) return GEOA3_ELAUNCH;
    if (hipEventElapsedTime(&ms_host[i], g_prof.slots[tag][i].e0, g_prof.slots[tag][i].e1) != hipSuccess)
      return GEOA3_ELAUNCH;
  }
  g_prof.n[tag] = 0;
  return n;
}
