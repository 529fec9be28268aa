// Temporary: entry points not implemented yet (removed as they land).
#include "ctx.h"
int finalize_superglue(im_ctx* ctx) { return ctx->fail(-99, "superglue: not implemented yet"); }
extern "C" {
#define NOTYET(ctx, name) return (ctx) ? (ctx)->fail(-99, name ": not implemented yet") : -1
int im_superglue_forward(im_ctx* ctx, const float*, const float*, const float*, const int32_t*, const float*, const im_superglue_conf*, int32_t*, float*, int32_t*, void*) { NOTYET(ctx, "im_superglue_forward"); }
int im_log_optimal_transport(im_ctx* ctx, const float*, int, int, int, float, int, float*, void*) { NOTYET(ctx, "im_log_optimal_transport"); }
}
