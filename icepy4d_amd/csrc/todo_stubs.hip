// Temporary: entry points not implemented yet (removed as they land).
#include "ctx.h"
extern "C" {
#define NOTYET(ctx, name) return (ctx) ? (ctx)->fail(-99, name ": not implemented yet") : -1
int im_ctx_reserve(im_ctx* ctx, int, int, int, int) { NOTYET(ctx, "im_ctx_reserve"); }
int im_set_tensor(im_ctx* ctx, const char*, const char*, const float*, size_t) { NOTYET(ctx, "im_set_tensor"); }
int im_finalize_weights(im_ctx* ctx, const char*) { NOTYET(ctx, "im_finalize_weights"); }
int im_superpoint_forward(im_ctx* ctx, const uint8_t*, int, int, int, int, float, int, int, int, float*, float*, float*, int32_t*, void*) { NOTYET(ctx, "im_superpoint_forward"); }
int im_lightglue_forward(im_ctx* ctx, const float*, const float*, const int32_t*, const float*, const im_lightglue_conf*, int32_t*, float*, int32_t*, int32_t*, void*) { NOTYET(ctx, "im_lightglue_forward"); }
int im_superglue_forward(im_ctx* ctx, const float*, const float*, const float*, const int32_t*, const float*, const im_superglue_conf*, int32_t*, float*, int32_t*, void*) { NOTYET(ctx, "im_superglue_forward"); }
int im_nms(im_ctx* ctx, const float*, float*, int, int, int, int, void*) { NOTYET(ctx, "im_nms"); }
int im_select_topk(im_ctx* ctx, const float*, int, int, int, int, float, int, float*, float*, int32_t*, void*) { NOTYET(ctx, "im_select_topk"); }
int im_sample_descriptors(im_ctx* ctx, const float*, int, int, int, const float*, const int32_t*, float*, void*) { NOTYET(ctx, "im_sample_descriptors"); }
int im_assign_from_sim(im_ctx* ctx, const float*, int, int, int, const float*, const float*, float, int32_t*, int32_t*, float*, float*, void*) { NOTYET(ctx, "im_assign_from_sim"); }
int im_log_optimal_transport(im_ctx* ctx, const float*, int, int, int, float, int, float*, void*) { NOTYET(ctx, "im_log_optimal_transport"); }
}
