// pending.hip -- entry points declared in include/rsdsfm.h whose kernels are not built yet.
// They fail loudly (no CPU fallback exists anywhere in this library).
#include "rsdsfm_internal.hpp"
using namespace rsdsfm;
#define NOT_YET(ctx, what) return fail((ctx) ? &(ctx)->c : nullptr, RSDSFM_ERR_INVALID, what ": HIP kernels not built yet")
extern "C" {
int rsdsfm_refine(rsdsfm_ctx* ctx, const double*, int64_t, int64_t, const double*, const double*, const double*, const int64_t*, const double*, const double*, double, int, int, double*, double*, double*, double*, rsdsfm_lm_summary*) { NOT_YET(ctx, "rsdsfm_refine"); }
int rsdsfm_flatten(rsdsfm_ctx* ctx, const double*, int32_t, int32_t, double, double, double, double, double, double, double*, double*, double*, double*, int64_t*) { NOT_YET(ctx, "rsdsfm_flatten"); }
int rsdsfm_depth_map(rsdsfm_ctx* ctx, double*, int64_t, double*, double, double, double, double, int32_t, int32_t, double*, int32_t*, int32_t*, int*) { NOT_YET(ctx, "rsdsfm_depth_map"); }
}
