// asvgp_create / asvgp_destroy and the per-handle settings (C-ABI, include/asvgp_hip.h).
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "handle.hpp"

namespace asvgp {

static DebugEnv g_dbg;
static std::once_flag g_dbg_once;            // (handles may be driven from two host threads: the first callers race for the load)
static void debug_env_load() {
  g_dbg.no_assembly = getenv("ASVGP_DEBUG_NO_ASSEMBLY") ? 1 : 0;   // test hook: the helpers never report -> the chains give up waiting
  g_dbg.chain_stamps = getenv("ASVGP_CHAIN_STAMPS") ? 1 : 0;
  g_dbg.spin_limit = getenv("ASVGP_SPIN_LIMIT") ? atol(getenv("ASVGP_SPIN_LIMIT")) : (1L << 25);
  g_dbg.host_times = getenv("ASVGP_HOST_TIMES") ? 1 : 0;
  g_dbg.plan_first = getenv("ASVGP_PLAN_FIRST") ? 1 : 0;
  g_dbg.bcr_stamps = getenv("ASVGP_BCR_STAMPS") ? atoi(getenv("ASVGP_BCR_STAMPS")) : 0;
  g_dbg.no_split = getenv("ASVGP_NO_SPLIT") ? atoi(getenv("ASVGP_NO_SPLIT")) : 0;   // the matrix-core P chain on ONE workgroup (measurement aid)
}
const DebugEnv& debug_env() {
  std::call_once(g_dbg_once, debug_env_load);
  return g_dbg;
}

static Handle* g_default = nullptr;
static std::mutex g_default_mu;

Handle* as_handle(asvgp_handle_t h) {
  if (h) return reinterpret_cast<Handle*>(h);
  std::lock_guard<std::mutex> lk(g_default_mu);
  if (!g_default) {
    g_default = new Handle;
    (void)hipGetDevice(&g_default->device);
  }
  return g_default;
}

// Wait until the GPU has consumed the last factor table handed to it (its kernel stores the table's sequence number into done[slot]
// once the table sits in LDS) and, when the result mirror is armed, until the last launch has written it: after that nothing in
// flight reads the pinned ring or writes the mirror.  Bounded; a launch that gave up waiting never reports, so the fallback is a
// device synchronisation.  This replaces a device-wide synchronisation on every model teardown.
static void handle_quiesce(Handle* h) {
  if (h->ahead.valid) (void)elbo_publish_theta(h, 1.0, 1.0, 1.0, true);   // (a launch still waiting for its theta is told to give up)
  handle_publish_forward(h);
  bool need_sync = false;
  struct timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  auto expired = [&]() { struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1); return (t1.tv_sec - t0.tv_sec) > 2; };
  if (h->done_host && h->seq > 0) {
    volatile unsigned long long* flag = h->done_host + (h->seq % TAB_SLOTS);
    while (*flag < h->seq) { sched_yield(); if (expired()) { need_sync = true; break; } }
  }
  if (!need_sync && h->mirror_host && h->mirror_pending) {
    volatile double* m = h->mirror_host + 10;
    while (*m != (double)h->mirror_pending) { sched_yield(); if (expired()) { need_sync = true; break; } }
  }
  if (need_sync) (void)hipDeviceSynchronize();
}

static void plan_release(Handle* h) {
  handle_quiesce(h);
  if (h->plan) { prior_plan_destroy(h->plan); h->plan = nullptr; }
  handle_prior_dd_release(h);
  if (h->node_rec_dev) { (void)hipFree(h->node_rec_dev); h->node_rec_dev = nullptr; }
  if (h->tab_host) { (void)hipHostFree(h->tab_host); h->tab_host = nullptr; h->tab_dev = nullptr; }
  if (h->done_host) { (void)hipHostFree(h->done_host); h->done_host = nullptr; h->done_dev = nullptr; }
  if (h->ready_host) { (void)hipHostFree(h->ready_host); h->ready_host = nullptr; h->ready_dev = nullptr; }
  h->slot_doubles = 0;
}

static void worker_wait_idle(Handle* h) {
  while (__atomic_load_n(&h->wstate, __ATOMIC_ACQUIRE) == 1) sched_yield();
}
// The worker spins while jobs keep coming (a post is then seen within a pause instruction); after ~2 ms without one it parks on the
// handle's condition variable - a model kept alive after fit() costs no host core - and the next post wakes it (one futex call).
struct WorkerPark {
  std::mutex mu;
  std::condition_variable cv;
  int parked = 0;                            // (written under mu)
};
static WorkerPark* park_of(Handle* h) { return static_cast<WorkerPark*>(h->worker_park); }
static void worker_main(Handle* h) {
  long idle = 0;
  WorkerPark* pk = park_of(h);
  for (;;) {
    const int st = __atomic_load_n(&h->wstate, __ATOMIC_ACQUIRE);
    if (st == 2) return;
    if (st == 1) {
      if (h->plan) (void)prior_plan_eval(h->plan, h->fwd.coef, h->fwd.dcoef, h->fwd.tab);   // a non-positive pivot is reported through `info` by the kernel
      __atomic_store_n(h->ready_host + h->fwd.slot, h->fwd.seq, __ATOMIC_RELEASE);
      __atomic_store_n(&h->wstate, 0, __ATOMIC_RELEASE);
      idle = 0;
    } else if (++idle > 200000) {            // (~2 ms of pauses) a model that has stopped stepping: sleep until the next post
      std::unique_lock<std::mutex> lk(pk->mu);
      pk->parked = 1;
      pk->cv.wait(lk, [&] { return __atomic_load_n(&h->wstate, __ATOMIC_ACQUIRE) != 0; });
      pk->parked = 0;
      idle = 0;
    } else {
      __builtin_ia32_pause();
    }
  }
}
static void worker_signal(Handle* h, int state) {
  WorkerPark* pk = park_of(h);
  __atomic_store_n(&h->wstate, state, __ATOMIC_SEQ_CST);
  bool wake;
  { std::lock_guard<std::mutex> lk(pk->mu); wake = pk->parked != 0; }   // (the worker sets `parked` and re-tests the state under the same lock)
  if (wake) pk->cv.notify_one();
}
void handle_post_forward(Handle* h) {
  if (!h->worker) {
    h->worker_park = new WorkerPark;
    h->worker = new std::thread(worker_main, h);
  }
  worker_signal(h, 1);
}
static void worker_stop(Handle* h) {
  if (!h->worker) return;
  worker_wait_idle(h);
  worker_signal(h, 2);
  std::thread* t = static_cast<std::thread*>(h->worker);
  t->join();
  delete t;
  delete park_of(h);
  h->worker_park = nullptr;
  h->worker = nullptr;
  h->wstate = 0;
}

void handle_publish_forward(Handle* h) {
  if (h->worker) worker_wait_idle(h);        // (worker mode: the pass is on its way; callers need it finished)
  if (!h->fwd.valid) return;
  h->fwd.valid = false;
  if (h->plan) (void)prior_plan_eval(h->plan, h->fwd.coef, h->fwd.dcoef, h->fwd.tab);   // a non-positive pivot is reported through `info` by the kernel
  __atomic_store_n(h->ready_host + h->fwd.slot, h->fwd.seq, __ATOMIC_RELEASE);
}

double* handle_table_acquire(Handle* h, unsigned long long* seq_out, int* slot_out) {
  handle_publish_forward(h);                 // (a launch still waiting for its table goes first)
  const unsigned long long seq = ++h->seq;
  const int slot = (int)(seq % TAB_SLOTS);
  if (seq > TAB_SLOTS) {   // the slot's previous table (sequence seq - TAB_SLOTS) must have been read by its kernel
    const unsigned long long need = seq - TAB_SLOTS;
    volatile unsigned long long* flag = h->done_host + slot;
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    while (*flag < need) {
      sched_yield();
      struct timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((t1.tv_sec - t0.tv_sec) > 2) { (void)hipDeviceSynchronize(); break; }   // (a failed launch never sets the flag)
    }
  }
  *seq_out = seq;
  *slot_out = slot;
  return h->tab_host + (size_t)slot * h->slot_doubles;
}

bool handle_mesh_is_linspace(Handle* h, const double* mesh_dev, long n_mesh, hipStream_t st, double* step_out, double* first_out, double* last_out) {
  for (int i = 0; i < h->n_mesh_seen; ++i)
    if (h->mesh_seen[i].ptr == mesh_dev && h->mesh_seen[i].n == n_mesh) {
      *step_out = h->mesh_seen[i].step;
      if (first_out) *first_out = h->mesh_seen[i].first;
      if (last_out) *last_out = h->mesh_seen[i].last;
      return h->mesh_seen[i].regular != 0;
    }
  std::vector<double> m((size_t)n_mesh);
  bool regular = false;
  double step = 0.0, first = 0.0, last = 0.0;
  if (n_mesh >= 2 && hipMemcpyAsync(m.data(), mesh_dev, sizeof(double) * (size_t)n_mesh, hipMemcpyDeviceToHost, st) == hipSuccess &&
      hipStreamSynchronize(st) == hipSuccess) {
    const double m0 = m[0];
    first = m0; last = m[n_mesh - 1];
    step = (m[n_mesh - 1] - m0) / (double)(n_mesh - 1);
    regular = step > 0.0 && step == step;
    for (long i = 0; i < n_mesh - 1 && regular; ++i) {
      volatile double t = (double)i * step;        // two roundings, no contraction: what numpy.linspace does
      regular = (m[i] == t + m0);
    }
  }
  if (h->n_mesh_seen < 8) h->mesh_seen[h->n_mesh_seen++] = Handle::MeshSeen{mesh_dev, n_mesh, regular ? 1 : 0, step, first, last};
  *step_out = step;
  if (first_out) *first_out = first;
  if (last_out) *last_out = last;
  return regular;
}

}  // namespace asvgp

using namespace asvgp;

// Host-only: Kuu for one theta in closed form (interior diagonal values + boundary-column table), as the matrix-core ELBO launch receives it.
extern "C" int asvgp_prior_interior_kuu_host(const double* static_bands_host, int n_terms, int64_t M, int k, const double* coef_host,
                                             double* kuu_diag8, int64_t* lo, int64_t* hi, double* bnd256) {
  if (!coef_host || !kuu_diag8 || !lo || !hi || !bnd256) { set_error("prior_interior_kuu_host: bad argument"); return ASVGP_ERR_BAD_ARG; }
  char err[256] = "";
  PriorPlan* p = prior_plan_create(static_bands_host, n_terms, (long)M, k, err, sizeof(err));
  if (!p) { set_error("%s", err); return strstr(err, "bad argument") ? ASVGP_ERR_BAD_ARG : ASVGP_ERR_UNSUPPORTED; }
  long l = 0, h = 0;
  for (int i = 0; i < 8; ++i) kuu_diag8[i] = 0.0;
  for (int i = 0; i < 2 * PRIOR_BND_DIAGS * PRIOR_BND; ++i) bnd256[i] = 0.0;
  prior_plan_interior_kuu(p, coef_host, kuu_diag8, &l, &h, bnd256);
  *lo = l; *hi = h;
  prior_plan_destroy(p);
  return ASVGP_OK;
}

extern "C" int asvgp_host_mantissa_bits(void) { return prior_plan_mantissa_bits(); }

extern "C" int asvgp_debug_reload_env(void) {
  asvgp::debug_env_load();
  return ASVGP_OK;
}

extern "C" int asvgp_create(asvgp_handle_t* out) {
  if (!out) { set_error("asvgp_create: bad argument"); return ASVGP_ERR_BAD_ARG; }
  Handle* h = new Handle;
  if (hipGetDevice(&h->device) != hipSuccess) { delete h; set_error("asvgp_create: no HIP device"); return ASVGP_ERR_HIP; }
  *out = reinterpret_cast<asvgp_handle_t>(h);
  return ASVGP_OK;
}

extern "C" int asvgp_destroy(asvgp_handle_t handle) {
  if (!handle) return ASVGP_OK;
  Handle* h = reinterpret_cast<Handle*>(handle);
  if (h->magic != 0x41535647u) { set_error("asvgp_destroy: not a handle"); return ASVGP_ERR_BAD_ARG; }
  worker_stop(h);
  plan_release(h);
  if (h->mirror_host) (void)hipHostFree(h->mirror_host);
  if (h->evK) (void)hipEventDestroy(h->evK);
  if (h->evP) (void)hipEventDestroy(h->evP);
  if (h->prof_made)
    for (int i = 0; i < PROF_RING; ++i) { (void)hipEventDestroy(h->prof_ev[i][0]); (void)hipEventDestroy(h->prof_ev[i][1]); }
  if (h->order_dev) { (void)hipFree(h->order_dev); h->order_dev = nullptr; }
  if (h->box_host) { (void)hipHostFree(h->box_host); h->box_host = nullptr; h->box_dev = nullptr; }
  h->magic = 0;
  delete h;
  return ASVGP_OK;
}

extern "C" int asvgp_set_phi_algorithm(asvgp_handle_t handle, int algo) {
  if (algo != 0 && algo != 1 && algo != 3 && algo != 5 && algo != 6) {
    set_error("set_phi_algorithm: 0 auto (6 where it applies, else 5, else 3), 1 fp64 LDS-atomic band scatter, 3 fixed-point band scatter, 5 fixed-point centred-moment scatter, 6 tile sort + register moments");
    return ASVGP_ERR_BAD_ARG;
  }
  as_handle(handle)->phi_algo = algo;
  return ASVGP_OK;
}

extern "C" int asvgp_set_phi_input_order(asvgp_handle_t handle, int order) {
  if (order < 0 || order > 2) { set_error("set_phi_input_order: 0 probe once per (x, N), 1 unsorted, 2 time series"); return ASVGP_ERR_BAD_ARG; }
  Handle* h = as_handle(handle);
  h->phi_order = order;
  h->n_order_seen = 0;
  return ASVGP_OK;
}

extern "C" int asvgp_phi_last_input_order(asvgp_handle_t handle) { return as_handle(handle)->phi_last_series ? 2 : 1; }

extern "C" int asvgp_set_phi_workgroups(asvgp_handle_t handle, int n) {
  if (n < 0 || n > 256) { set_error("set_phi_workgroups: 0 (default, one per CU) .. 256"); return ASVGP_ERR_BAD_ARG; }
  as_handle(handle)->phi_blocks = n;
  return ASVGP_OK;
}

extern "C" int asvgp_set_phi_deferred_reduce(asvgp_handle_t handle, int on) {
  Handle* h = as_handle(handle);
  h->phi_defer = on != 0;
  if (!h->phi_defer) h->pend.valid = false;
  return ASVGP_OK;
}

extern "C" int asvgp_set_band_algorithm(asvgp_handle_t handle, int algo) {
  if (algo < 0 || algo > 4) {
    set_error("set_band_algorithm: 0 auto, 1 sequential sweeps, 2 block cyclic reduction on the GPU, 3 block cyclic reduction with the planned (host, long double) prior forward pass, 4 the planned chains on the matrix cores");
    return ASVGP_ERR_BAD_ARG;
  }
  as_handle(handle)->band_algo = algo;
  return ASVGP_OK;
}

extern "C" int asvgp_set_deferred_forward_pass(asvgp_handle_t handle, int on) {
  Handle* h = as_handle(handle);
  if (on < 0 || on > 2) { set_error("set_deferred_forward_pass: 0 inline, 1 the caller publishes (asvgp_prior_publish), 2 the handle's worker thread"); return ASVGP_ERR_BAD_ARG; }
  handle_publish_forward(h);
  if (on != 2) worker_stop(h);
  h->defer_forward = on == 1;
  h->fwd_worker = on == 2;
  return ASVGP_OK;
}

extern "C" int asvgp_prior_publish(asvgp_handle_t handle) {
  handle_publish_forward(as_handle(handle));
  return ASVGP_OK;
}

extern "C" int asvgp_result_mirror(asvgp_handle_t handle, int enable, const double** host_ptr) {
  Handle* h = as_handle(handle);
  if (enable && !h->mirror_host) {
    if (hipHostMalloc(reinterpret_cast<void**>(&h->mirror_host), sizeof(double) * 16, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void**>(&h->mirror_dev), h->mirror_host, 0) != hipSuccess) {
      if (h->mirror_host) { (void)hipHostFree(h->mirror_host); h->mirror_host = nullptr; }
      h->mirror_dev = nullptr;
      set_error("result_mirror: pinned allocation failed: %s", hipGetErrorString(hipGetLastError()));
      return ASVGP_ERR_HIP;
    }
    memset(h->mirror_host, 0, sizeof(double) * 16);
  }
  if (!enable && h->mirror_host) {
    handle_quiesce(h);                       // (a launch in flight may still write it)
    (void)hipHostFree(h->mirror_host);
    h->mirror_host = nullptr; h->mirror_dev = nullptr; h->mirror_pending = 0;
  }
  if (host_ptr) *host_ptr = h->mirror_host;
  return ASVGP_OK;
}

extern "C" uint64_t asvgp_result_mirror_pending(asvgp_handle_t handle) { return as_handle(handle)->mirror_pending; }

extern "C" int asvgp_elbo_chain_sync(asvgp_handle_t handle, int enable) {
  Handle* h = as_handle(handle);
  if (enable && !h->evK) {
    if (hipEventCreateWithFlags(&h->evK, hipEventDisableTiming | hipEventReleaseToDevice) != hipSuccess ||
        hipEventCreateWithFlags(&h->evP, hipEventDisableTiming | hipEventReleaseToDevice) != hipSuccess) {
      set_error("elbo_chain_sync: hipEventCreate failed");
      return ASVGP_ERR_HIP;
    }
  }
  h->sync_on = enable != 0;
  return ASVGP_OK;
}

extern "C" int asvgp_profile_enable(asvgp_handle_t handle, int on) {
  Handle* h = as_handle(handle);
  if (on && !h->prof_made) {
    for (int i = 0; i < PROF_RING; ++i)
      for (int j = 0; j < 2; ++j)
        if (hipEventCreateWithFlags(&h->prof_ev[i][j], hipEventReleaseToDevice) != hipSuccess) { set_error("hipEventCreate failed"); return ASVGP_ERR_HIP; }
    h->prof_made = true;
  }
  h->prof_on = on != 0;
  h->prof_every = on > 1 ? on : 1;   // on = n > 1: every n-th launch
  h->prof_calls = 0;
  h->prof_n = 0;
  return ASVGP_OK;
}

extern "C" int asvgp_profile_read(asvgp_handle_t handle, double* phi_kernel_ms_sum, int64_t* launches) {
  Handle* h = as_handle(handle);
  if (!phi_kernel_ms_sum || !launches) { set_error("profile_read: bad argument"); return ASVGP_ERR_BAD_ARG; }
  double tot = 0.0;
  for (long i = 0; i < h->prof_n; ++i) {
    if (hipEventSynchronize(h->prof_ev[i][1]) != hipSuccess) { set_error("hipEventSynchronize failed"); return ASVGP_ERR_HIP; }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, h->prof_ev[i][0], h->prof_ev[i][1]);
    tot += ms;
  }
  *phi_kernel_ms_sum = tot;
  *launches = h->prof_n;
  h->prof_n = 0;
  return ASVGP_OK;
}

// Plan of the prior chain for one (basis, kernel kind): static_bands_host = the (n_terms, k+1, M) HOST copy of the array the
// ELBO entry points receive on the device.  Returns ASVGP_OK with *planned = 1, or *planned = 0 when the bands have no
// Toeplitz structure to exploit (the all-GPU chain is used then).
extern "C" int asvgp_prior_plan_1d(asvgp_handle_t handle, const double* static_bands_host, int n_terms, int64_t M, int k,
                                   int* planned) {
  Handle* h = as_handle(handle);
  if (planned) *planned = 0;
  plan_release(h);
  if (!static_bands_host) return ASVGP_OK;     // NULL: drop the plan
  char err[256] = "";
  PriorPlan* p = prior_plan_create(static_bands_host, n_terms, (long)M, k, err, sizeof(err));
  if (!p) {
    if (strstr(err, "bad argument")) { set_error("%s", err); return ASVGP_ERR_BAD_ARG; }
    return ASVGP_OK;                           // unstructured band: no plan, not an error
  }
  const int nb = prior_plan_nb(p);
  h->slot_doubles = (prior_plan_table_doubles(p) + 63) / 64 * 64;
  bool ok = hipMalloc(reinterpret_cast<void**>(&h->node_rec_dev), sizeof(int) * (size_t)nb) == hipSuccess &&
            hipMemcpy(h->node_rec_dev, prior_plan_node_rec(p), sizeof(int) * (size_t)nb, hipMemcpyHostToDevice) == hipSuccess &&
            hipHostMalloc(reinterpret_cast<void**>(&h->tab_host), sizeof(double) * h->slot_doubles * TAB_SLOTS, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) == hipSuccess &&
            hipHostGetDevicePointer(reinterpret_cast<void**>(&h->tab_dev), h->tab_host, 0) == hipSuccess &&
            hipHostMalloc(reinterpret_cast<void**>(&h->done_host), sizeof(unsigned long long) * TAB_SLOTS, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) == hipSuccess &&
            hipHostGetDevicePointer(reinterpret_cast<void**>(&h->done_dev), h->done_host, 0) == hipSuccess &&
            hipHostMalloc(reinterpret_cast<void**>(&h->ready_host), sizeof(unsigned long long) * TAB_SLOTS, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) == hipSuccess &&
            hipHostGetDevicePointer(reinterpret_cast<void**>(&h->ready_dev), h->ready_host, 0) == hipSuccess;
  if (!ok) {
    prior_plan_destroy(p);
    plan_release(h);
    set_error("prior_plan_1d: device / pinned allocation failed: %s", hipGetErrorString(hipGetLastError()));
    return ASVGP_ERR_HIP;
  }
  // The handle's sequence number is MONOTONIC across re-plans: the message boxes of the split launch in the caller's
  // workspace (separator share, partial sums, done words) are validated by tags derived from it and are never cleared, so a new plan
  // that started again at 1 could accept what the old plan's launch number 1 left there.  The fresh ring counts as consumed up to it.
  for (int i = 0; i < TAB_SLOTS; ++i) { h->done_host[i] = h->seq; h->ready_host[i] = h->seq; }
  h->plan = p;
  h->plan_terms = n_terms;
  if (planned) *planned = 1;
  return ASVGP_OK;
}

// Host-only evaluation of the planner (no device): table of prior_plan_eval and the node -> record map for one theta.
// table_host needs asvgp_prior_table_doubles entries, node_rec_host ceil(M / k) ints.  Used by the CPU tests; the product
// path goes through asvgp_prior_plan_1d + the ELBO entry points.
extern "C" size_t asvgp_prior_table_doubles(const double* static_bands_host, int n_terms, int64_t M, int k) {
  char err[256];
  PriorPlan* p = prior_plan_create(static_bands_host, n_terms, (long)M, k, err, sizeof(err));
  if (!p) return 0;
  const size_t n = prior_plan_table_doubles(p);
  prior_plan_destroy(p);
  return n;
}

extern "C" int asvgp_prior_forward_host(const double* static_bands_host, int n_terms, int64_t M, int k, const double* coef_host,
                                        const double* dcoef_dl_host, double* table_host, size_t table_doubles, int* node_rec_host) {
  if (!coef_host || !dcoef_dl_host || !table_host || !node_rec_host) { set_error("prior_forward_host: bad argument"); return ASVGP_ERR_BAD_ARG; }
  char err[256] = "";
  PriorPlan* p = prior_plan_create(static_bands_host, n_terms, (long)M, k, err, sizeof(err));
  if (!p) { set_error("%s", err); return strstr(err, "bad argument") ? ASVGP_ERR_BAD_ARG : ASVGP_ERR_UNSUPPORTED; }
  if (table_doubles < prior_plan_table_doubles(p)) { prior_plan_destroy(p); set_error("prior_forward_host: table too small"); return ASVGP_ERR_WORKSPACE; }
  (void)prior_plan_eval(p, coef_host, dcoef_dl_host, table_host);
  memcpy(node_rec_host, prior_plan_node_rec(p), sizeof(int) * (size_t)prior_plan_nb(p));
  prior_plan_destroy(p);
  return ASVGP_OK;
}

// Host-only: the device image of the plan (what the GPU forward pass of prior_dd.hpp walks): ints / doubles as laid out in prior_plan.cpp.
// Call with NULL buffers to get the sizes.  For the CPU tests.
extern "C" int asvgp_prior_plan_image_host(const double* static_bands_host, int n_terms, int64_t M, int k, int* ints, size_t* n_ints,
                                           double* doubles, size_t* n_doubles) {
  if (!n_ints || !n_doubles) { set_error("prior_plan_image_host: bad argument"); return ASVGP_ERR_BAD_ARG; }
  char err[256] = "";
  PriorPlan* p = prior_plan_create(static_bands_host, n_terms, (long)M, k, err, sizeof(err));
  if (!p) { set_error("%s", err); return strstr(err, "bad argument") ? ASVGP_ERR_BAD_ARG : ASVGP_ERR_UNSUPPORTED; }
  const size_t ni = prior_plan_image_ints(p), nd = prior_plan_image_doubles(p);
  int rc = ASVGP_OK;
  if (ints && doubles) {
    if (*n_ints < ni || *n_doubles < nd) { set_error("prior_plan_image_host: buffers too small"); rc = ASVGP_ERR_WORKSPACE; }
    else prior_plan_image(p, ints, doubles);
  }
  *n_ints = ni; *n_doubles = nd;
  prior_plan_destroy(p);
  return rc;
}
