// extern "C" boundary of libicematch (see include/icematch.h): context, weights, and the stage entry points.
// The model-level orchestration lives in superpoint.hip / lightglue.hip / superglue.hip.
#include "ctx.h"

#include <cstdio>
#include <cstdlib>
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
    const char* g = getenv("IM_DEBUG_GUARDS");
    c->guards_on = g && g[0] == '1';
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
    g.C = d_c; g.ldc = n; g.alpha = alpha; g.epi = EPI_BIAS; g.big_tile = big_tile & 1; g.bx = (big_tile >> 1) & 1;
    IM_HIP(ctx, launch_gemm(g, (hipStream_t)stream));
    return 0;
}

int im_ffn_fused(im_ctx* ctx, int act, float* d_x, const float* d_att, const float* h_w0, const float* h_b0, const float* h_ln_g,
                   const float* h_ln_b, const float* h_w3, const float* h_b3, int n_images, int n_rows, const int32_t* d_n, void* stream) {
    IM_CHECK_CTX(ctx);
    if (n_images < 1 || n_rows < 1 || act < 0 || act > 1 || (act == 0 && (!h_ln_g || !h_ln_b))) return ctx->fail(-12, "im_ffn_fused: bad arguments");
    if (act == 1) h_ln_g = h_ln_b = h_b0;   // unused by the ReLU form
    const std::vector<float> p0 = pack_frag_weights(h_w0, 512, 512), p3 = pack_frag_weights(h_w3, 256, 512);
    const size_t sizes[6] = {p0.size(), 512, 512, 512, p3.size(), 256};
    const float* src[6] = {p0.data(), h_b0, h_ln_g, h_ln_b, p3.data(), h_b3};
    float* d[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 6 && e == hipSuccess; ++i) {
        e = hipMalloc(&d[i], sizes[i] * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(d[i], src[i], sizes[i] * sizeof(float), hipMemcpyHostToDevice);
    }
    hipStream_t st = (hipStream_t)stream;
    if (e == hipSuccess) {
        FfnArgs f;
        f.act = act; f.x = d_x; f.x_bstride = (long)n_rows * 256; f.att = d_att; f.att_bstride = (long)n_rows * 256;
        f.w0p = d[0]; f.b0 = d[1]; f.ln_g = d[2]; f.ln_b = d[3]; f.w3p = d[4]; f.b3 = d[5];
        f.m_max = n_rows; f.batch = n_images; f.m_ptr = d_n; f.pstride = 2;
        if (ctx->prof_on) {
            im_ctx::ProfEntry pe{"ffn_fused", ctx->prof_event(), ctx->prof_event()};
            hipEventRecord(pe.e0, st);
            e = launch_ffn_fused(f, st);
            hipEventRecord(pe.e1, st);
            ctx->prof.push_back(pe);
        } else {
            e = launch_ffn_fused(f, st);
        }
    }
    const hipError_t e2 = hipStreamSynchronize(st);
    for (int i = 0; i < 6; ++i) hipFree(d[i]);
    IM_HIP(ctx, e);
    IM_HIP(ctx, e2);
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
    std::vector<float> planes = wino ? pack_conv3x3_wino_bx(h_weight, cout, cin) : std::vector<float>(1);   // the product form of the Winograd layer (IM_CONV_F32=1: the f32-input form)
    float *dw = nullptr, *db = nullptr, *dx = nullptr;
    IM_HIP(ctx, hipMalloc(&dw, packed.size() * sizeof(float)));
    IM_HIP(ctx, hipMalloc(&db, cout * sizeof(float)));
    IM_HIP(ctx, hipMalloc(&dx, planes.size() * sizeof(float)));
    IM_HIP(ctx, hipMemcpy(dw, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
    IM_HIP(ctx, hipMemcpy(db, h_bias, cout * sizeof(float), hipMemcpyHostToDevice));
    IM_HIP(ctx, hipMemcpy(dx, planes.data(), planes.size() * sizeof(float), hipMemcpyHostToDevice));
    ConvArgs a;
    a.in = d_in; a.w = dw; a.bias = db; a.out = d_out; a.B = b; a.H = h; a.W = w; a.Cin = cin; a.Cout = cout;
    a.relu = relu; a.pool = pool;
    if (wino) a.wx = dx;
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
    hipFree(dx);
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
    a.n_ptr = d_n; a.n_max = n_max; a.batch = batch; a.heads = heads; a.cross = cross & 1; a.f32_form = (cross >> 1) & 1; a.scale = scale;
    // split-KV scratch of the stage entry point (the model paths use the reserved workspace): grown on demand
    // each of the three grows on its own, to max(old, new) (growing both when either was short re-allocated on alternating shapes,
    // a failed second allocation leaked the first, and the planes were grown for calls that do not use them)
    const size_t nf = attn_part_floats(n_max, batch, heads), ni = attn_counter_ints(n_max, batch, heads);
    if (nf > ctx->stage_attn_floats) {
        IM_HIP(ctx, hipDeviceSynchronize());
        float* p = ctx->dalloc<float>(nf, "stage_attn_part");
        if (!p) return ctx->fail(-11, "im_flash_attn: out of device memory");
        ctx->dfree(ctx->stage_attn_part);        // the smaller scratch it replaces (the device is idle: synchronised above)
        ctx->stage_attn_part = p; ctx->stage_attn_floats = nf;
    }
    if (ni > ctx->stage_attn_ints) {
        IM_HIP(ctx, hipDeviceSynchronize());
        int* c = ctx->dalloc<int>(ni, "stage_attn_cnt");
        if (!c) return ctx->fail(-11, "im_flash_attn: out of device memory");
        IM_HIP(ctx, hipMemset(c, 0, ni * sizeof(int)));
        ctx->dfree(ctx->stage_attn_cnt);
        ctx->stage_attn_cnt = c; ctx->stage_attn_ints = ni;
    }
    const bool wants_planes = !(cross & 4) && !a.f32_form;      // bit 2: K / V cut inside the attention kernel; the f32 form has no planes
    const size_t nb = attn_planes_bytes(n_max, batch, heads);
    if (wants_planes && nb > ctx->stage_attn_plane_bytes) {
        IM_HIP(ctx, hipDeviceSynchronize());
        unsigned char* p = ctx->dalloc<unsigned char>(nb, "stage_attn_planes");
        if (!p) return ctx->fail(-11, "im_flash_attn: out of device memory");
        ctx->dfree(ctx->stage_attn_planes);
        ctx->stage_attn_planes = p; ctx->stage_attn_plane_bytes = nb;
    }
    a.part = ctx->stage_attn_part; a.counters = ctx->stage_attn_cnt; a.planes = wants_planes ? ctx->stage_attn_planes : nullptr;
    IM_HIP(ctx, launch_attn_planes(a, (hipStream_t)stream));
    IM_HIP(ctx, launch_flash_attn(a, (hipStream_t)stream));
    IM_GUARD_CHECK(ctx, (hipStream_t)stream, "im_flash_attn");
    return 0;
}

}  // extern "C"
