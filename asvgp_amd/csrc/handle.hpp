// The opaque library handle (asvgp_create / asvgp_destroy): everything that used to be process-wide state -
// algorithm choices, the Phi-pass workgroup count, the chain-ordering events, the kernel-timing ring, and the host
// planner of the prior chain with its pinned factor-table ring - lives here, one handle per model / stream / host thread.
#pragma once
#include "asvgp_common.hpp"
#include "prior_plan.hpp"

namespace asvgp {

constexpr int PROF_RING = 1024;
constexpr int TAB_SLOTS = 16;

struct Handle {
  unsigned magic = 0x41535647u;   // 'ASVG'
  int device = 0;
  // Phi pass
  int phi_algo = 0, phi_blocks = 0;
  int phi_last = 0;                       // algorithm the last asvgp_phi_accumulate_1d actually ran (1, 3, 5, 6)
  // deferred cross-workgroup reduce (asvgp_set_phi_deferred_reduce): the moment kernel's partials wait here for asvgp_phi_reduce_1d
  bool phi_defer = false;
  struct PendingReduce { const double* partials; int G, M, K; double* stats; bool valid; const int* ranges; } pend = {nullptr, 0, 0, 0, nullptr, false, nullptr};
  // band algebra
  int band_algo = 0;
  bool sync_on = false;
  hipEvent_t evK = nullptr, evP = nullptr;
  // timing ring around the dominant kernel (bench.py's roofline figure)
  bool prof_on = false, prof_made = false;
  int prof_every = 1;
  long prof_calls = 0, prof_n = 0;
  hipEvent_t prof_ev[PROF_RING][2];
  // prior chain planner (prior_plan.cpp): host forward pass, factor tables handed to the GPU through a pinned ring
  PriorPlan* plan = nullptr;
  int* node_rec_dev = nullptr;            // device copy of the node -> record map
  double* tab_host = nullptr;             // TAB_SLOTS x slot_doubles, pinned + mapped
  double* tab_dev = nullptr;              // the same memory as seen from the device
  unsigned long long* done_host = nullptr;   // per slot: sequence number of the last table the GPU has finished reading
  unsigned long long* done_dev = nullptr;
  unsigned long long* ready_host = nullptr;  // per slot: sequence number of the table the HOST has finished writing (the matrix-core launch is
  unsigned long long* ready_dev = nullptr;   // issued BEFORE the host forward pass; its Kuu workgroup waits here - run_chains in elbo.hip)
  size_t slot_doubles = 0;
  unsigned long long seq = 0;
  int plan_terms = 0;
  // deferred forward pass (asvgp_set_deferred_forward_pass): the matrix-core launch returns at once; asvgp_prior_publish runs the host
  // forward pass for it and sets the slot's ready word (the next ELBO call or the handle's teardown does it if the caller did not)
  bool defer_forward = false;
  // ... or runs on the handle's own worker thread (asvgp_set_deferred_forward_pass(h, 2)): posted BEFORE the launch call, so the ~19 us of
  // long-double arithmetic overlap the launch path and the caller's next enqueues; the worker spins while jobs keep coming and parks on a condition variable after ~2 ms without one
  bool fwd_worker = false;
  void* worker = nullptr;                 // std::thread*
  void* worker_park = nullptr;            // condition variable the idle worker sleeps on (handle.hip)
  int wstate = 0;                         // 0 idle, 1 job posted, 2 exit  (accessed with __atomic builtins)
  struct PendingForward { bool valid; double coef[ASVGP_MAX_KUU_TERMS], dcoef[ASVGP_MAX_KUU_TERMS]; double* tab; int slot; unsigned long long seq; } fwd = {false, {0}, {0}, nullptr, 0, 0};
  // forward pass on the GPU in double-double (asvgp_set_prior_forward(h, 1); prior_dd.hip): device image of the plan, device table ring
  bool prior_forward_gpu = false;
  int* dd_img_i = nullptr;
  int dd_n_img = 0;
  double* dd_img_d = nullptr;
  double* dd_tab = nullptr;               // TAB_SLOTS x slot_doubles, device memory (slot numbering shared with the pinned ring)
  hipStream_t dd_stream = nullptr;        // the matrix-core launch runs the pass BESIDE the chains' launch (its Kuu workgroup waits on dd_ready)
  unsigned long long* dd_ready = nullptr; // TAB_SLOTS device words: sequence number of the table the pass has finished writing
  // result mirror (asvgp_result_mirror): 16 pinned doubles the fused launch's last ticket writes [out[0..7], info[0], info[1], sequence]
  double* mirror_host = nullptr;
  double* mirror_dev = nullptr;
  unsigned long long mirror_seq = 0;      // sequence number of the last launch that writes the mirror
  unsigned long long mirror_pending = 0;  // = mirror_seq when the LAST ELBO launch writes the mirror, else 0
  // meshes already inspected (handle_mesh_is_linspace): device pointer, length, first / last knot -> verdict
  struct MeshSeen { const double* ptr; long n; int regular; double step, first, last; };
  MeshSeen mesh_seen[8];
  int n_mesh_seen = 0;
  // input order of the Phi pass (asvgp_set_phi_input_order): 0 = probe once per (x pointer, N) - a sample of 128-point rows, one
  // 4-byte device-to-host copy -, 1 = treat as unsorted, 2 = treat as a time series.  Both kernels are correct for ANY input: the
  // verdict only chooses between the instantiation that is fastest on i.i.d. points and the one with the time-series front loop.
  int phi_order = 0;
  struct OrderSeen { const double* ptr; long n; int series; };
  OrderSeen order_seen[8];
  int n_order_seen = 0;
  int* order_dev = nullptr;               // one device word for the probe's count
  int phi_last_series = 0;                // 1: the last tile-sort launch was the time-series instantiation
  // launch-ahead of the matrix-core ELBO launch (asvgp_elbo_grad_ahead_1d / asvgp_elbo_publish_theta): a pinned ring of theta boxes (one per
  // table slot) the kernel reads once the host has filled it; `ahead`: the launch that is out and still waits for its theta
  bool ahead_req = false;                 // set by the entry point around its call into the launcher
  void* box_host = nullptr;               // TAB_SLOTS x BOX_BYTES, pinned + mapped
  void* box_dev = nullptr;
  struct PendingAhead { bool valid; int slot; unsigned long long seq; double* tab; int kind; long N; } ahead = {false, 0, 0, nullptr, 0, 0};
};
constexpr size_t BOX_BYTES = 4096;

// Debug / measurement switches from the environment, read ONCE (a getenv per launch costs microseconds where the environment is large)
// and again on asvgp_debug_reload_env() - the tests that flip them call that.
struct DebugEnv { int no_assembly; int chain_stamps; long spin_limit; int host_times; int plan_first; int bcr_stamps; int no_split; };
const DebugEnv& debug_env();

Handle* as_handle(asvgp_handle_t h);      // NULL -> the process-wide default handle (created on first use)

// Is the device mesh table the fp64 linspace numpy makes (knot i = i * step + start, last knot = stop)?  Decided once per
// (pointer, length) from a host copy (a 16 KB device-to-host copy and ONE stream synchronisation, at the first Phi pass of a
// model); the kernel instantiation that generates its knots on the VALU re-checks the table and reports a mismatch loudly.
bool handle_mesh_is_linspace(Handle* h, const double* mesh_dev, long n_mesh, hipStream_t st, double* step_out, double* first_out, double* last_out);

// A deferred cross-workgroup reduce (asvgp_set_phi_deferred_reduce) that is still pending is enqueued on `st` (phi_pass.hip).  Called by
// asvgp_phi_reduce_1d, by every consumer of the statistics buffer (so a bound is never built from the zeroed buffer) and by the next
// accumulate call on the handle (so the parked partials are never overwritten).  stats != NULL: only when the pending reduce targets it.
int handle_flush_phi_reduce(Handle* h, const double* stats, hipStream_t st);

// run a pending deferred forward pass (no-op when none): prior_plan_eval into the slot, then the ready word; worker mode: wait for it
void handle_publish_forward(Handle* h);
// worker mode: hand the pass described by h->fwd to the worker thread (started on first use)
void handle_post_forward(Handle* h);

// all-GPU forward pass (prior_dd.hip): enqueue the double-double forward pass for one theta on `st` into slot `slot` of the handle's
// device table ring (*tab_out); the consumer is launched behind it on the same stream and needs no ready word
int handle_prior_dd_prepare(Handle* h);
void handle_prior_dd_release(Handle* h);
// seq != 0: on the handle's own stream, concurrently with whatever `st` holds; the kernel then publishes seq in dd_ready[slot] (*ready_out)
int handle_prior_dd_forward(Handle* h, const double* coef, const double* dcoef, int slot, hipStream_t st, double** tab_out, unsigned long long* stamps = nullptr,
                            unsigned long long seq = 0, const unsigned long long** ready_out = nullptr);

// launch-ahead: fill the pending launch's theta box and publish it, then run the host forward pass for it (elbo.hip, k = 4 unit);
// withdraw = true: tell the waiting kernel to give up instead (teardown)
int elbo_publish_theta(Handle* h, double v, double l, double s, bool withdraw);

// next table slot for writing: waits (bounded) until the GPU has consumed the slot's previous table
double* handle_table_acquire(Handle* h, unsigned long long* seq_out, int* slot_out);

}  // namespace asvgp
