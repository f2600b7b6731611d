// Shared between the host planner (prior_plan.cpp, plain C++) and the device backward pass (bcr_pre.hpp).
// Table layout (doubles): [PRIOR_TAB_HEADER header | value plane: n_rec records of prior_rec_fields(B) | tangent plane: same]
// header: [0] log|Kuu|, [1] d log|Kuu| / d lengthscale, [2] first failing column + 1 (0 = positive definite), [3] n_rec.
// Record of an eliminated node (the fields bcr.hpp keeps per node in its workspace): L (B x B, lower, row-major), 1 / diag(L)
// (B), U_a = L^-1 A[i,a] (B x B), U_b = L^-1 A[i,b] (B x B), then what the matrix-core backward pass (bcr_mfma.hpp) consumes directly:
// G_a^T = U_a^T L^-1, G_b^T = U_b^T L^-1, D^-1 = L^-T L^-1 (B x B each).  The root record carries Sigma_00 in the U_a slot.
#pragma once
#include <stddef.h>

#include "../../include/asvgp_hip.h"

#if defined(__HIPCC__) || defined(__CUDACC__)
#define ASVGP_HD __host__ __device__
#else
#define ASVGP_HD
#endif

namespace asvgp {

constexpr int PRIOR_TAB_HEADER = 8;
ASVGP_HD constexpr int prior_f_L(int) { return 0; }
ASVGP_HD constexpr int prior_f_I(int B) { return B * B; }
ASVGP_HD constexpr int prior_f_UA(int B) { return B * B + B; }
ASVGP_HD constexpr int prior_f_UB(int B) { return 2 * B * B + B; }
ASVGP_HD constexpr int prior_f_GAT(int B) { return 3 * B * B + B; }
ASVGP_HD constexpr int prior_f_GBT(int B) { return 4 * B * B + B; }
ASVGP_HD constexpr int prior_f_DINV(int B) { return 5 * B * B + B; }
ASVGP_HD constexpr int prior_rec_fields(int B) { return 6 * B * B + B; }

int prior_plan_mantissa_bits();                               // mantissa width of the host forward pass's arithmetic (64: x87 extended)
struct PriorPlan;
PriorPlan* prior_plan_create(const double* statics_host, int n_terms, long M, int k, char* err, size_t errlen);
void prior_plan_destroy(PriorPlan* p);
// Kuu for one theta without the band: B + 1 interior diagonal values on columns [*lo, *hi) and the boundary columns' entries
// (bnd: 2 * PRIOR_BND_DIAGS * PRIOR_BND doubles; layout in prior_plan.cpp); *hi <= *lo: not available
constexpr int PRIOR_BND = 16, PRIOR_BND_DIAGS = 8;
void prior_plan_interior_kuu(const PriorPlan* p, const double* coef, double* kuu_diag, long* lo, long* hi, double* bnd);
int prior_plan_nrec(const PriorPlan* p);
int prior_plan_nb(const PriorPlan* p);
long prior_plan_M(const PriorPlan* p);
int prior_plan_k(const PriorPlan* p);
int prior_plan_terms(const PriorPlan* p);
const int* prior_plan_node_rec(const PriorPlan* p);          // nb ints: record index of every block node (root: n_rec - 1)
size_t prior_plan_table_doubles(const PriorPlan* p);
// device image of the plan for the all-GPU (double-double) forward pass of prior_dd.hip; layout in prior_plan.cpp
constexpr int PRIOR_MAX_CLASSES = 24;                         // node / block classes per level a plan may have
size_t prior_plan_image_ints(const PriorPlan* p);
size_t prior_plan_image_doubles(const PriorPlan* p);
void prior_plan_image(const PriorPlan* p, int* ints, double* dbls);
// numeric pass for one theta: coef / dcoef = asvgp_matern_coeffs; fills `tab` (host memory); returns the failing column + 1 or 0
int prior_plan_eval(const PriorPlan* p, const double* coef, const double* dcoef, double* tab);

}  // namespace asvgp
