// extern "C" boundary of libicematch (see include/icematch.h): context, weights, and the stage entry points.
// The model-level orchestration lives in superpoint.hip / lightglue.hip / superglue.hip.
#include "ctx.h"

#include <cstdio>
#include <cstring>

using namespace im;

extern "C" {

int im_version(void) { return 100; }

int im_ctx_create(int device, im_ctx** out) {
    if (!out) return -1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return -2;
    if (hipSetDevice(device) != hipSuccess) return -3;
    im_ctx* c = new im_ctx();
    c->device = device;
    *out = c;
    return 0;
}

void im_ctx_destroy(im_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    ctx->free_all();
    delete ctx;
}

const char* im_last_error(im_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int im_gemm_nt(im_ctx* ctx, const float* d_a, const float* d_w, const float* d_bias, float* d_c, int m, int n, int k,
               float alpha, int big_tile, void* stream) {
    IM_CHECK_CTX(ctx);
    GemmArgs g;
    g.A = d_a; g.lda = k; g.W = d_w; g.ldw = k; g.bias = d_bias; g.N = n; g.K = k; g.m_max = m;
    g.C = d_c; g.ldc = n; g.alpha = alpha; g.epi = EPI_BIAS; g.big_tile = big_tile;
    IM_HIP(ctx, launch_gemm(g, (hipStream_t)stream));
    return 0;
}

static int conv3x3_entry(im_ctx* ctx, bool wino, const float* d_in, const float* h_weight, const float* h_bias, float* d_out,
                         int b, int h, int w, int cin, int cout, int relu, int pool, void* stream);

int im_conv3x3(im_ctx* ctx, const float* d_in, const float* h_weight, const float* h_bias, float* d_out, int b, int h,
               int w, int cin, int cout, int relu, int pool, void* stream) {
    return conv3x3_entry(ctx, false, d_in, h_weight, h_bias, d_out, b, h, w, cin, cout, relu, pool, stream);
}

int im_conv3x3_winograd(im_ctx* ctx, const float* d_in, const float* h_weight, const float* h_bias, float* d_out, int b,
                        int h, int w, int cin, int cout, int relu, int pool, void* stream) {
    return conv3x3_entry(ctx, true, d_in, h_weight, h_bias, d_out, b, h, w, cin, cout, relu, pool, stream);
}

static int conv3x3_entry(im_ctx* ctx, bool wino, const float* d_in, const float* h_weight, const float* h_bias, float* d_out,
                         int b, int h, int w, int cin, int cout, int relu, int pool, void* stream) {
    IM_CHECK_CTX(ctx);
    if (cin % 16 || cout % 64) return ctx->fail(-10, "im_conv3x3: cin %% 16 and cout %% 64 must be 0");
    std::vector<float> packed = wino ? pack_conv3x3_wino(h_weight, cout, cin) : pack_conv3x3(h_weight, cout, cin);
    float *dw = nullptr, *db = nullptr;
    IM_HIP(ctx, hipMalloc(&dw, packed.size() * sizeof(float)));
    IM_HIP(ctx, hipMalloc(&db, cout * sizeof(float)));
    IM_HIP(ctx, hipMemcpy(dw, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
    IM_HIP(ctx, hipMemcpy(db, h_bias, cout * sizeof(float), hipMemcpyHostToDevice));
    ConvArgs a;
    a.in = d_in; a.w = dw; a.bias = db; a.out = d_out; a.B = b; a.H = h; a.W = w; a.Cin = cin; a.Cout = cout;
    a.relu = relu; a.pool = pool;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    if (ctx->prof_on) {
        im_ctx::ProfEntry pe{wino ? "conv_wino" : "conv_direct", ctx->prof_event(), ctx->prof_event()};
        hipEventRecord(pe.e0, st);
        e = wino ? launch_conv3x3_wino(a, st) : launch_conv3x3(a, st);
        hipEventRecord(pe.e1, st);
        ctx->prof.push_back(pe);
    } else {
        e = wino ? launch_conv3x3_wino(a, st) : launch_conv3x3(a, st);
    }
    hipError_t e2 = hipStreamSynchronize(st);
    hipFree(dw);
    hipFree(db);
    IM_HIP(ctx, e);
    IM_HIP(ctx, e2);
    return 0;
}

int im_flash_attn(im_ctx* ctx, const float* d_q, const float* d_k, const float* d_v, float* d_out, const int32_t* d_n,
                  int n_max, int batch, int heads, int cross, float scale, void* stream) {
    IM_CHECK_CTX(ctx);
    AttnArgs a;
    a.q = d_q; a.k = d_k; a.v = d_v; a.hstride = (long)n_max * 64; a.bstride = a.hstride * heads;
    a.out = d_out; a.ldo = heads * 64; a.out_bstride = (long)n_max * a.ldo;
    a.n_ptr = d_n; a.n_max = n_max; a.batch = batch; a.heads = heads; a.cross = cross; a.scale = scale;
    // split-KV scratch of the stage entry point (the model paths use the reserved workspace): grown on demand
    const size_t nf = attn_part_floats(n_max, batch, heads), ni = attn_counter_ints(n_max, batch, heads);
    if (nf > ctx->stage_attn_floats || ni > ctx->stage_attn_ints) {
        IM_HIP(ctx, hipDeviceSynchronize());
        float* p = ctx->dalloc<float>(nf);
        int* c = ctx->dalloc<int>(ni);
        if (!p || !c) return ctx->fail(-11, "im_flash_attn: out of device memory");
        IM_HIP(ctx, hipMemset(c, 0, ni * sizeof(int)));
        ctx->stage_attn_part = p; ctx->stage_attn_cnt = c; ctx->stage_attn_floats = nf; ctx->stage_attn_ints = ni;
    }
    a.part = ctx->stage_attn_part; a.counters = ctx->stage_attn_cnt;
    IM_HIP(ctx, launch_flash_attn(a, (hipStream_t)stream));
    return 0;
}

}  // extern "C"
