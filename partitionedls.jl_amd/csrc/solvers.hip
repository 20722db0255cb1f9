// solvers.hip — fit(Alt) and fit(BnB) on the Gram kernels (placeholders until implemented).
#include "common.h"
extern "C" {
partls_status partls_fit_alt(partls_ctx *, const double *, int64_t, int64_t, int64_t, const double *, const int64_t *, int64_t,
                             int64_t, double, double, int64_t, const double *, const double *, double *, double *, double *,
                             double *, int64_t *)
{ partls::set_error("partls_fit_alt: not implemented yet"); return PARTLS_ERR_UNSUPPORTED; }
partls_status partls_fit_bnb(partls_ctx *, const double *, int64_t, int64_t, int64_t, const double *, const int64_t *, int64_t,
                             int64_t, double, double *, double *, double *, double *, int64_t *)
{ partls::set_error("partls_fit_bnb: not implemented yet"); return PARTLS_ERR_UNSUPPORTED; }
}
