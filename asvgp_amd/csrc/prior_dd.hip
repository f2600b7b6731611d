// Launchers of the all-GPU (double-double) forward pass of the prior chain: device code and rationale in prior_dd.hpp.
// This translation unit is also compiled with -ffp-contract=off (asvgp_amd/build.py); the header states it per function as well, since
// the fused ELBO launch (elbo.hip) includes it into a translation unit that contracts.
#include <hip/hip_runtime.h>

#include <vector>

#include "handle.hpp"
#include "prior_dd.hpp"

namespace asvgp {

namespace {

template <int B>
__global__ __launch_bounds__(64 * pd_waves<B>()) void prior_forward_dd_kernel(const int* __restrict__ img, const double* __restrict__ stat,
                                                                              PdCoefs cf, double* __restrict__ tab, int n_img_lds,
                                                                              unsigned long long* stamps, unsigned long long* ready, unsigned long long seq) {
  extern __shared__ double lds[];
  prior_forward_dd<B>(img, n_img_lds, stat, cf, tab, lds, (int)threadIdx.x, 64 * pd_waves<B>(), stamps);
  // a consumer on ANOTHER stream (the matrix-core launch's Kuu workgroup) waits for this word: every wave's stores are acknowledged (the
  // barrier above), one release writes the L2 back for the other XCDs
  if (ready && threadIdx.x == 0) __hip_atomic_store(ready, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

template <int B>
int pd_launch(const int* img_i, int n_img, int n_rec, const double* img_d, const PdCoefs& cf, double* tab, hipStream_t st, unsigned long long* stamps,
              unsigned long long* ready, unsigned long long seq) {
  int n_img_lds = n_img;                                          // the class maps ride in the LDS when they fit beside the blocks
  if (sizeof(double) * pd_lds_doubles<B>(n_rec, n_img_lds) > 160 * 1024) n_img_lds = 0;
  const size_t lds_bytes = sizeof(double) * pd_lds_doubles<B>(n_rec, n_img_lds);
  if (lds_bytes > 160 * 1024) { set_error("prior forward pass on the GPU: %zu B of LDS", lds_bytes); return ASVGP_ERR_LDS_CAPACITY; }
  auto kern = prior_forward_dd_kernel<B>;
  static size_t granted = 64 * 1024;
  if (lds_bytes > granted) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
    granted = lds_bytes;
  }
  hipLaunchKernelGGL(kern, dim3(1), dim3(64 * pd_waves<B>()), lds_bytes, st, img_i, img_d, cf, tab, n_img_lds, stamps, ready, seq);
  return ASVGP_OK;
}

}  // namespace

// the device image of the handle's plan (uploaded once per plan) and a device table ring of the pinned ring's shape
int handle_prior_dd_prepare(Handle* h) {
  if (!h->plan) { set_error("prior forward pass on the GPU: the handle holds no plan (asvgp_prior_plan_1d)"); return ASVGP_ERR_BAD_ARG; }
  if (h->dd_img_i) return ASVGP_OK;
  const size_t ni = prior_plan_image_ints(h->plan), nd = prior_plan_image_doubles(h->plan);
  std::vector<int> ints(ni);
  std::vector<double> dbls(nd ? nd : 1);
  prior_plan_image(h->plan, ints.data(), dbls.data());
  h->dd_n_img = (int)ni;
  bool ok = hipMalloc(reinterpret_cast<void**>(&h->dd_img_i), sizeof(int) * ni) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&h->dd_img_d), sizeof(double) * (nd ? nd : 1)) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&h->dd_tab), sizeof(double) * h->slot_doubles * TAB_SLOTS) == hipSuccess &&
            hipMemcpy(h->dd_img_i, ints.data(), sizeof(int) * ni, hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(h->dd_img_d, dbls.data(), sizeof(double) * (nd ? nd : 1), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) {
    handle_prior_dd_release(h);
    set_error("prior forward pass on the GPU: device allocation failed: %s", hipGetErrorString(hipGetLastError()));
    return ASVGP_ERR_HIP;
  }
  return ASVGP_OK;
}

void handle_prior_dd_release(Handle* h) {
  if (h->dd_img_i) { (void)hipFree(h->dd_img_i); h->dd_img_i = nullptr; }
  if (h->dd_img_d) { (void)hipFree(h->dd_img_d); h->dd_img_d = nullptr; }
  if (h->dd_tab) { (void)hipFree(h->dd_tab); h->dd_tab = nullptr; }
  if (h->dd_stream) { (void)hipStreamSynchronize(h->dd_stream); (void)hipStreamDestroy(h->dd_stream); h->dd_stream = nullptr; }
  if (h->dd_ready) { (void)hipFree(h->dd_ready); h->dd_ready = nullptr; }
}

// enqueue the forward pass for one theta on `st`; the table lands in slot `slot` of the handle's DEVICE ring (returned)
int handle_prior_dd_forward(Handle* h, const double* coef, const double* dcoef, int slot, hipStream_t st, double** tab_out, unsigned long long* stamps,
                            unsigned long long seq, const unsigned long long** ready_out) {
  int rc = handle_prior_dd_prepare(h);
  if (rc) return rc;
  hipStream_t lst = st;
  unsigned long long* rdy = nullptr;
  if (seq) {
    if (!h->dd_stream) {
      if (hipStreamCreateWithFlags(&h->dd_stream, hipStreamNonBlocking) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&h->dd_ready), sizeof(unsigned long long) * TAB_SLOTS) != hipSuccess ||
          hipMemset(h->dd_ready, 0, sizeof(unsigned long long) * TAB_SLOTS) != hipSuccess) {
        set_error("prior forward pass on the GPU: stream / ready words: %s", hipGetErrorString(hipGetLastError()));
        return ASVGP_ERR_HIP;
      }
    }
    lst = h->dd_stream;
    rdy = h->dd_ready + slot;
    if (ready_out) *ready_out = rdy;
  }
  PdCoefs cf;
  for (int t = 0; t < ASVGP_MAX_KUU_TERMS; ++t) { cf.c[t] = coef[t]; cf.dc[t] = dcoef[t]; }
  double* tab = h->dd_tab + (size_t)slot * h->slot_doubles;
  switch (prior_plan_k(h->plan)) {
    case 1: rc = pd_launch<1>(h->dd_img_i, h->dd_n_img, prior_plan_nrec(h->plan), h->dd_img_d, cf, tab, lst, stamps, rdy, seq); break;
    case 2: rc = pd_launch<2>(h->dd_img_i, h->dd_n_img, prior_plan_nrec(h->plan), h->dd_img_d, cf, tab, lst, stamps, rdy, seq); break;
    case 3: rc = pd_launch<3>(h->dd_img_i, h->dd_n_img, prior_plan_nrec(h->plan), h->dd_img_d, cf, tab, lst, stamps, rdy, seq); break;
    case 4: rc = pd_launch<4>(h->dd_img_i, h->dd_n_img, prior_plan_nrec(h->plan), h->dd_img_d, cf, tab, lst, stamps, rdy, seq); break;
    case 5: rc = pd_launch<5>(h->dd_img_i, h->dd_n_img, prior_plan_nrec(h->plan), h->dd_img_d, cf, tab, lst, stamps, rdy, seq); break;
    case 6: rc = pd_launch<6>(h->dd_img_i, h->dd_n_img, prior_plan_nrec(h->plan), h->dd_img_d, cf, tab, lst, stamps, rdy, seq); break;
    default: set_error("prior forward pass on the GPU: bandwidth 1..6"); return ASVGP_ERR_BAD_ARG;
  }
  if (rc) return rc;
  *tab_out = tab;
  return check_launch("prior forward pass (double-double)");
}

}  // namespace asvgp

using namespace asvgp;

extern "C" int asvgp_set_prior_forward(asvgp_handle_t handle, int mode) {
  if (mode != 0 && mode != 1) { set_error("set_prior_forward: 0 host (long double), 1 GPU (double-double)"); return ASVGP_ERR_BAD_ARG; }
  as_handle(handle)->prior_forward_gpu = mode == 1;
  return ASVGP_OK;
}

// The GPU forward pass alone: the table for one theta copied back to the host (tests; compare with asvgp_prior_forward_host).
extern "C" int asvgp_prior_forward_device(asvgp_handle_t handle, const double* coef_host, const double* dcoef_dl_host, double* table_host,
                                          size_t table_doubles, void* stream) {
  Handle* h = as_handle(handle);
  if (!coef_host || !dcoef_dl_host || !table_host) { set_error("prior_forward_device: bad argument"); return ASVGP_ERR_BAD_ARG; }
  if (!h->plan) { set_error("prior_forward_device: the handle holds no plan (asvgp_prior_plan_1d)"); return ASVGP_ERR_BAD_ARG; }
  const size_t n = prior_plan_table_doubles(h->plan);
  if (table_doubles < n) { set_error("prior_forward_device: table too small"); return ASVGP_ERR_WORKSPACE; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  unsigned long long seq = 0;
  int slot = 0;
  (void)handle_table_acquire(h, &seq, &slot);
  double* tab = nullptr;
  int rc = handle_prior_dd_forward(h, coef_host, dcoef_dl_host, slot, st, &tab);
  if (rc) return rc;
  if (hipMemcpyAsync(table_host, tab, sizeof(double) * n, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
    set_error("prior_forward_device: copy failed: %s", hipGetErrorString(hipGetLastError()));
    return ASVGP_ERR_HIP;
  }
  __atomic_store_n(h->done_host + slot, seq, __ATOMIC_RELEASE);   // (no kernel consumes this slot)
  return ASVGP_OK;
}

// Diagnostics: cycle stamps (s_memtime of thread 0) of one forward pass - out64[63] = how many, out64[0..] = the stamps in kernel order
// (tools/prior_dd_probe.py names them).
extern "C" int asvgp_prior_forward_stamps(asvgp_handle_t handle, const double* coef_host, const double* dcoef_dl_host, uint64_t* out64, void* stream) {
  Handle* h = as_handle(handle);
  if (!coef_host || !dcoef_dl_host || !out64 || !h->plan) { set_error("prior_forward_stamps: bad argument / no plan"); return ASVGP_ERR_BAD_ARG; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  unsigned long long* dev = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&dev), sizeof(unsigned long long) * 64) != hipSuccess) { set_error("prior_forward_stamps: hipMalloc failed"); return ASVGP_ERR_HIP; }
  (void)hipMemsetAsync(dev, 0, sizeof(unsigned long long) * 64, st);
  unsigned long long seq = 0;
  int slot = 0;
  (void)handle_table_acquire(h, &seq, &slot);
  double* tab = nullptr;
  int rc = handle_prior_dd_forward(h, coef_host, dcoef_dl_host, slot, st, &tab, dev);
  if (rc == ASVGP_OK && (hipMemcpyAsync(out64, dev, sizeof(unsigned long long) * 64, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) {
    set_error("prior_forward_stamps: copy failed");
    rc = ASVGP_ERR_HIP;
  }
  (void)hipFree(dev);
  __atomic_store_n(h->done_host + slot, seq, __ATOMIC_RELEASE);
  return rc;
}
