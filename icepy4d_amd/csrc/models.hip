// Model-level orchestration behind the C ABI: workspace, weight packing, SuperPoint and LightGlue forward
// passes as a fixed sequence of kernel launches on one stream (no host synchronisation inside a forward:
// keypoint counts, early-stop and pruning state live in device memory).
#include <cmath>
#include <cstdlib>
#include <cstring>

#include <algorithm>
#include "ctx.h"
#include "lg_misc.h"
#include "sp_post.h"
#include "workspace.h"

using namespace im;

static const char* SP_CONV3[10] = {"conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convPa", "convDa", nullptr};
static const int SP_CIN[9] = {64, 64, 64, 64, 128, 128, 128, 128, 128};
static const int SP_COUT[9] = {64, 64, 64, 128, 128, 128, 128, 256, 256};

static const std::vector<float>* find_w(im_ctx* ctx, const std::string& model, const std::string& key, size_t numel) {
    auto it = ctx->host_w.find(model + "/" + key);
    if (it == ctx->host_w.end()) {
        ctx->fail(-20, "weights: missing tensor %s of model %s", key.c_str(), model.c_str());
        return nullptr;
    }
    if (it->second.size() != numel) {
        ctx->fail(-21, "weights: tensor %s has %zu elements, expected %zu", key.c_str(), it->second.size(), numel);
        return nullptr;
    }
    return &it->second;
}

#define GETW(var, model, key, numel)                           \
    const std::vector<float>* var = find_w(ctx, model, key, numel); \
    if (!var) return -20

static int finalize_superpoint(im_ctx* ctx) {
    SuperPointW& w = ctx->sp;
    {
        GETW(cw, "superpoint", "conv1a.weight", 64 * 9);
        GETW(cb, "superpoint", "conv1a.bias", 64);
        std::vector<float> p(9 * 64);
        for (int co = 0; co < 64; ++co)
            for (int t = 0; t < 9; ++t) p[t * 64 + co] = (*cw)[co * 9 + t];
        w.c1a_w = ctx->upload(p);
        w.c1a_b = ctx->upload(*cb);
        // contraction index k of the matrix form: 0 = bias (times the in-image mask), 1 + t = tap t; lane half hh supplies k = 2 s + hh
        std::vector<float> pq(64 * 2 * 8, 0.f);
        for (int co = 0; co < 64; ++co)
            for (int k = 0; k < 10; ++k) pq[((size_t)co * 2 + (k & 1)) * 8 + (k >> 1)] = k == 0 ? (*cb)[co] : (*cw)[co * 9 + (k - 1)];
        w.c1a_wq = ctx->upload(pq);
    }
    for (int i = 0; SP_CONV3[i]; ++i) {
        const std::string nm = SP_CONV3[i];
        GETW(cw, "superpoint", nm + ".weight", (size_t)SP_COUT[i] * SP_CIN[i] * 9);
        GETW(cb, "superpoint", nm + ".bias", (size_t)SP_COUT[i]);
        w.cw[i] = ctx->upload(pack_conv3x3(cw->data(), SP_COUT[i], SP_CIN[i]));
        w.cww[i] = ctx->upload(pack_conv3x3_wino(cw->data(), SP_COUT[i], SP_CIN[i]));
        w.cwx[i] = ctx->upload(pack_conv3x3_wino_bx(cw->data(), SP_COUT[i], SP_CIN[i]));
        w.cb[i] = ctx->upload(*cb);
        if (!w.cw[i] || !w.cww[i] || !w.cwx[i] || !w.cb[i]) return ctx->fail(-22, "weights: upload failed");
    }
    GETW(pbw, "superpoint", "convPb.weight", 65 * 256);
    GETW(pbb, "superpoint", "convPb.bias", 65);
    GETW(dbw, "superpoint", "convDb.weight", 256 * 256);
    GETW(dbb, "superpoint", "convDb.bias", 256);
    w.pb_w = ctx->upload(*pbw); w.pb_b = ctx->upload(*pbb);
    w.db_w = ctx->upload(*dbw); w.db_b = ctx->upload(*dbb);
    if (!w.pb_w || !w.pb_b || !w.db_w || !w.db_b || !w.c1a_w || !w.c1a_b) return ctx->fail(-22, "weights: upload failed");
    w.ready = true;
    return 0;
}

static int finalize_lightglue(im_ctx* ctx) {
    LightGlueW& w = ctx->lg;
    const int L = 9;
    auto cat = [&](const char* fmt, size_t numel, std::vector<float>& dst, int count) -> int {
        dst.clear();
        for (int i = 0; i < count; ++i) {
            char key[160];
            snprintf(key, sizeof(key), fmt, i);
            const std::vector<float>* t = find_w(ctx, "lightglue", key, numel);
            if (!t) return -20;
            dst.insert(dst.end(), t->begin(), t->end());
        }
        return 0;
    };
    std::vector<float> buf;
#define CAT_UP(dstptr, fmt, numel, count)        \
    if (cat(fmt, numel, buf, count)) return -20; \
    dstptr = ctx->upload(buf);                   \
    if (!dstptr) return ctx->fail(-22, "weights: upload failed")

    {
        GETW(wr, "lightglue", "posenc.Wr.weight", 64);
        w.wr = ctx->upload(*wr);
    }
    // Wqkv rows: original index head*192 + d*3 + which  ->  which*256 + head*64 + d  (`lightglue.py:155`)
    {
        std::vector<float> qw, qb, pw((size_t)L * 768 * 256), pb((size_t)L * 768);
        if (cat("transformers.%d.self_attn.Wqkv.weight", 768 * 256, qw, L)) return -20;
        if (cat("transformers.%d.self_attn.Wqkv.bias", 768, qb, L)) return -20;
        for (int l = 0; l < L; ++l)
            for (int h = 0; h < 4; ++h)
                for (int d = 0; d < 64; ++d)
                    for (int which = 0; which < 3; ++which) {
                        const int src = h * 192 + d * 3 + which, dst = which * 256 + h * 64 + d;
                        memcpy(&pw[((size_t)l * 768 + dst) * 256], &qw[((size_t)l * 768 + src) * 256], 256 * sizeof(float));
                        pb[(size_t)l * 768 + dst] = qb[(size_t)l * 768 + src];
                    }
        w.qkv_w = ctx->upload(pw);
        w.qkv_b = ctx->upload(pb);
        {   // per layer: fragment order for proj_rows_kernel
            std::vector<float> packed;
            packed.reserve(pw.size() * 3 / 2);
            for (int l = 0; l < L; ++l) {
                const std::vector<float> one = pack_frag_weights(&pw[(size_t)l * 768 * 256], 768, 256);
                packed.insert(packed.end(), one.begin(), one.end());
            }
            w.qkv_wp = ctx->upload(packed);
        }
    }
    // out_proj / to_out are folded into the second half of ffn.0 (`lightglue.py:160-162, 212-216`: the message is used
    // only as ffn input): ffn.0([x | Wo a + bo]) = W0a x + (W0b Wo) a + (W0b bo + b0). Products accumulated in double;
    // one 256 -> 256 GEMM launch and the `message` round trip through HBM less per block.
    auto pack_layers = [&](const std::vector<float>& src, int n, int k) -> float* {   // per layer: fragment order for ffn_fused_kernel
        std::vector<float> packed;
        packed.reserve(src.size());
        for (int l = 0; l < L; ++l) {
            const std::vector<float> one = pack_frag_weights(&src[(size_t)l * n * k], n, k);
            packed.insert(packed.end(), one.begin(), one.end());
        }
        return ctx->upload(packed);
    };
    auto fold = [&](const char* ow, const char* ob, const char* fw, const char* fb, float*& dw, float*& db, float*& dwp) -> int {
        std::vector<float> o_w, o_b, f_w, f_b;
        if (cat(ow, 256 * 256, o_w, L) || cat(ob, 256, o_b, L) || cat(fw, 512 * 512, f_w, L) || cat(fb, 512, f_b, L)) return -20;
        std::vector<double> row(256);
        for (int l = 0; l < L; ++l) {
            const float* Wo = &o_w[(size_t)l * 65536];
            const float* bo = &o_b[(size_t)l * 256];
            for (int n = 0; n < 512; ++n) {
                float* w0 = &f_w[((size_t)l * 512 + n) * 512 + 256];
                double bacc = f_b[(size_t)l * 512 + n];
                for (int k = 0; k < 256; ++k) row[k] = 0.0;
                for (int j = 0; j < 256; ++j) {
                    const double wj = w0[j];
                    const float* wor = Wo + (size_t)j * 256;
                    for (int k = 0; k < 256; ++k) row[k] += wj * (double)wor[k];
                    bacc += wj * (double)bo[j];
                }
                for (int k = 0; k < 256; ++k) w0[k] = (float)row[k];
                f_b[(size_t)l * 512 + n] = (float)bacc;
            }
        }
        dw = ctx->upload(f_w);
        db = ctx->upload(f_b);
        dwp = pack_layers(f_w, 512, 512);
        return (dw && db && dwp) ? 0 : -22;
    };
    if (int rc = fold("transformers.%d.self_attn.out_proj.weight", "transformers.%d.self_attn.out_proj.bias",
                      "transformers.%d.self_attn.ffn.0.weight", "transformers.%d.self_attn.ffn.0.bias", w.sf0_w, w.sf0_b, w.sf0_wp))
        return ctx->fail(rc, "weights: self ffn.0 / out_proj");
    CAT_UP(w.sln_g, "transformers.%d.self_attn.ffn.1.weight", 512, L);
    CAT_UP(w.sln_b, "transformers.%d.self_attn.ffn.1.bias", 512, L);
    CAT_UP(w.sf3_w, "transformers.%d.self_attn.ffn.3.weight", 256 * 512, L);
    if (!(w.sf3_wp = pack_layers(buf, 256, 512))) return ctx->fail(-22, "weights: upload failed");
    CAT_UP(w.sf3_b, "transformers.%d.self_attn.ffn.3.bias", 256, L);
    {   // [to_qk ; to_v] as one 256 -> 512 projection
        std::vector<float> qw, qb, vw, vb, pw((size_t)L * 512 * 256), pb((size_t)L * 512);
        if (cat("transformers.%d.cross_attn.to_qk.weight", 256 * 256, qw, L) || cat("transformers.%d.cross_attn.to_qk.bias", 256, qb, L) ||
            cat("transformers.%d.cross_attn.to_v.weight", 256 * 256, vw, L) || cat("transformers.%d.cross_attn.to_v.bias", 256, vb, L))
            return -20;
        for (int l = 0; l < L; ++l) {
            memcpy(&pw[(size_t)l * 512 * 256], &qw[(size_t)l * 65536], 65536 * sizeof(float));
            memcpy(&pw[(size_t)l * 512 * 256 + 65536], &vw[(size_t)l * 65536], 65536 * sizeof(float));
            memcpy(&pb[(size_t)l * 512], &qb[(size_t)l * 256], 256 * sizeof(float));
            memcpy(&pb[(size_t)l * 512 + 256], &vb[(size_t)l * 256], 256 * sizeof(float));
        }
        w.cqv_w = ctx->upload(pw);
        w.cqv_b = ctx->upload(pb);
        {
            std::vector<float> packed;
            packed.reserve(pw.size() * 3 / 2);
            for (int l = 0; l < L; ++l) {
                const std::vector<float> one = pack_frag_weights(&pw[(size_t)l * 512 * 256], 512, 256);
                packed.insert(packed.end(), one.begin(), one.end());
            }
            w.cqv_wp = ctx->upload(packed);
        }
        if (!w.cqv_w || !w.cqv_b || !w.cqv_wp) return ctx->fail(-22, "weights: upload failed");
    }
    if (int rc = fold("transformers.%d.cross_attn.to_out.weight", "transformers.%d.cross_attn.to_out.bias",
                      "transformers.%d.cross_attn.ffn.0.weight", "transformers.%d.cross_attn.ffn.0.bias", w.cf0_w, w.cf0_b, w.cf0_wp))
        return ctx->fail(rc, "weights: cross ffn.0 / to_out");
    CAT_UP(w.cln_g, "transformers.%d.cross_attn.ffn.1.weight", 512, L);
    CAT_UP(w.cln_b, "transformers.%d.cross_attn.ffn.1.bias", 512, L);
    CAT_UP(w.cf3_w, "transformers.%d.cross_attn.ffn.3.weight", 256 * 512, L);
    if (!(w.cf3_wp = pack_layers(buf, 256, 512))) return ctx->fail(-22, "weights: upload failed");
    CAT_UP(w.cf3_b, "transformers.%d.cross_attn.ffn.3.bias", 256, L);
    CAT_UP(w.fp_w, "log_assignment.%d.final_proj.weight", 256 * 256, L);
    CAT_UP(w.fp_b, "log_assignment.%d.final_proj.bias", 256, L);
    CAT_UP(w.ma_w, "log_assignment.%d.matchability.weight", 256, L);
    CAT_UP(w.ma_b, "log_assignment.%d.matchability.bias", 1, L);
    CAT_UP(w.tc_w, "token_confidence.%d.token.0.weight", 256, L - 1);
    CAT_UP(w.tc_b, "token_confidence.%d.token.0.bias", 1, L - 1);
#undef CAT_UP
    if (ctx->host_w.count("lightglue/confidence_thresholds")) {
        GETW(thr, "lightglue", "confidence_thresholds", (size_t)L);
        for (int i = 0; i < L; ++i) w.thr[i] = (*thr)[i];
    } else {
        // a registered buffer the reference computes in __init__ and loads with strict=False (`lightglue.py:371-373, 392,
        // 558-561`): absent from a checkpoint saved without buffers. Same formula, double -> float32.
        for (int i = 0; i < L; ++i) {
            double t = 0.8 + 0.1 * std::exp(-4.0 * i / L);
            w.thr[i] = (float)(t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t));
        }
    }
    if (!w.wr || !w.qkv_w || !w.qkv_b || !w.qkv_wp) return ctx->fail(-22, "weights: upload failed");
    w.ready = true;
    return 0;
}

int finalize_superglue(im_ctx* ctx);  // superglue.hip

extern "C" {

int im_profile_begin(im_ctx* ctx) {
    IM_CHECK_CTX(ctx);
    IM_HIP(ctx, hipDeviceSynchronize());
    for (auto& e : ctx->prof) { ctx->prof_pool.push_back(e.e0); ctx->prof_pool.push_back(e.e1); }
    ctx->prof.clear();
    ctx->prof_on = true;
    return 0;
}

// Writes {"name": {"count": n, "total_ms": t}, ...} (aggregated over the launches since im_profile_begin).
int im_profile_end(im_ctx* ctx, char* buf, size_t cap) {
    IM_CHECK_CTX(ctx);
    ctx->prof_on = false;
    // calibration: an event pair with nothing in between still reads a few microseconds (each record is a packet the
    // command processor has to retire); measured here on the same stream and reported as "_empty_event_pair" so that
    // the caller can subtract it per launch
    constexpr int NCAL = 32;
    hipEvent_t cal[2 * NCAL];
    for (int i = 0; i < NCAL; ++i) {
        cal[2 * i] = ctx->prof_event(); cal[2 * i + 1] = ctx->prof_event();
        hipEventRecord(cal[2 * i], ctx->prof_stream);
        hipEventRecord(cal[2 * i + 1], ctx->prof_stream);
    }
    IM_HIP(ctx, hipDeviceSynchronize());
    std::map<std::string, std::pair<int, double>> agg;
    for (int i = 0; i < NCAL; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, cal[2 * i], cal[2 * i + 1]) == hipSuccess) { agg["_empty_event_pair"].first += 1; agg["_empty_event_pair"].second += ms; }
        ctx->prof_pool.push_back(cal[2 * i]); ctx->prof_pool.push_back(cal[2 * i + 1]);
    }
    for (auto& e : ctx->prof) {
        float ms = 0.f;
        IM_HIP(ctx, hipEventElapsedTime(&ms, e.e0, e.e1));
        auto& a = agg[e.name];
        a.first += 1;
        a.second += ms;
    }
    std::string out = "{";
    bool first = true;
    for (auto& kv : agg) {
        char line[256];
        snprintf(line, sizeof(line), "%s\"%s\": {\"count\": %d, \"total_ms\": %.6f}", first ? "" : ", ", kv.first.c_str(),
                 kv.second.first, kv.second.second);
        out += line;
        first = false;
    }
    out += "}";
    if (!buf || out.size() + 1 > cap) return ctx->fail(-60, "im_profile_end: buffer too small (%zu needed)", out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    return 0;
}

int im_set_tensor(im_ctx* ctx, const char* model, const char* key, const float* h_data, size_t numel) {
    IM_CHECK_CTX(ctx);
    if (!model || !key || !h_data) return ctx->fail(-1, "im_set_tensor: null argument");
    ctx->host_w[std::string(model) + "/" + key].assign(h_data, h_data + numel);
    return 0;
}

int im_finalize_weights(im_ctx* ctx, const char* model) {
    IM_CHECK_CTX(ctx);
    const std::string m = model ? model : "";
    if (m != "superpoint" && m != "lightglue" && m != "superglue")
        return ctx->fail(-23, "im_finalize_weights: unknown model '%s'", m.c_str());
    IM_HIP(ctx, hipDeviceSynchronize());
    std::vector<void*>& mine = ctx->model_allocs[m];
    for (void* p : mine) ctx->gfree(p);   // a reload replaces the previous device copy of this model
    mine.clear();
    ctx->cur_model = &mine;
    int rc;
    if (m == "superpoint") rc = finalize_superpoint(ctx);
    else if (m == "lightglue") rc = finalize_lightglue(ctx);
    else rc = finalize_superglue(ctx);
    ctx->cur_model = nullptr;
    if (rc) return rc;
    IM_HIP(ctx, hipDeviceSynchronize());
    return 0;
}

int im_ctx_reserve(im_ctx* ctx, int max_h, int max_w, int max_images, int max_kpts) {
    IM_CHECK_CTX(ctx);
    if (max_h < 8 || max_w < 8 || max_images < 1 || max_kpts < 1) return ctx->fail(-30, "im_ctx_reserve: bad sizes");
    // the kernels address one image / one score matrix through 32-bit buffer offsets (range-checked descriptors): fail loudly
    // instead of wrapping around. Larger inputs go through the tile modes of the matcher (`matchers.py:304-469`).
    // (16 rows of margin: halo rows below the image and the rows of a last, partial region are addressed too - past the descriptor's
    // range, where loads return zero and stores are dropped - and their offsets must not wrap either)
    if ((long)(max_h / 2 + 16) * (max_w / 2) * 64 * 4 >= (1L << 32))
        return ctx->fail(-33, "im_ctx_reserve: %d x %d exceeds 67 MP per image (32-bit offsets into the half-resolution activations): use a tile mode", max_h, max_w);
    if (max_kpts >= 32768)
        return ctx->fail(-34, "im_ctx_reserve: %d keypoints per image: the K x K score matrix is addressed with 32-bit byte offsets (K < 32768)", max_kpts);
    if (ctx->ws && max_h <= ctx->max_h && max_w <= ctx->max_w && max_images <= ctx->max_images && max_kpts <= ctx->max_kpts) return 0;
    IM_HIP(ctx, hipDeviceSynchronize());
    if (ctx->ws) {
        IM_GUARD_CHECK(ctx, nullptr, "the last use of the workspace that im_ctx_reserve replaces");
        for (void* p : ctx->ws->allocs) ctx->gfree(p);
        delete ctx->ws;
        ctx->ws = nullptr;
    }
    if (ctx->max_h > max_h) max_h = ctx->max_h;
    if (ctx->max_w > max_w) max_w = ctx->max_w;
    if (ctx->max_images > max_images) max_images = ctx->max_images;
    if (ctx->max_kpts > max_kpts) max_kpts = ctx->max_kpts;
    Workspace* ws = new Workspace();
    const long B = max_images;
    const long H8 = (max_h / 8) * 8, W8 = (max_w / 8) * 8, cells = (H8 / 8) * (W8 / 8), K = max_kpts;
    bool ok = true;
    auto alloc_named = [&](auto*& p, size_t n, const char* name) {
        using T = std::remove_reference_t<decltype(*p)>;
        p = reinterpret_cast<T*>(ctx->galloc(n * sizeof(T), name, ws->allocs));
        if (!p) ok = false;
    };
#define A(ptr, n) alloc_named(ptr, n, #ptr)
    A(ws->act0, (size_t)B * (max_h / 2) * (max_w / 2) * 64);  // conv1a is fused: nothing is stored at full resolution
    A(ws->act1, (size_t)B * (max_h / 2) * (max_w / 2) * 64);
    A(ws->logits, (size_t)B * cells * 65);
    A(ws->dense, (size_t)B * cells * 256);
    A(ws->smap, (size_t)B * H8 * W8);
    A(ws->nms, (size_t)B * H8 * W8);
    A(ws->rest, (size_t)B * H8 * W8);
    A(ws->mask, (size_t)B * H8 * W8);
    A(ws->supp, (size_t)B * H8 * W8);
    A(ws->kpsel.n_cand, (sel_state_bytes((int)B) + 3) / 4);
    A(ws->kpsel.keys, (size_t)B * H8 * W8);
    ws->kpsel.ties_cap = (size_t)B * (H8 + 32) * (W8 + 64);      // whole 32 x 64 tiles per image: the candidate staging of the last NMS round
    A(ws->kpsel.ties, ws->kpsel.ties_cap);
    A(ws->kpsel.chosen, (size_t)B * K);
    // matcher buffers: NP pairs = 2 NP images side by side (batch over pairs: one launch serves every pair)
    const size_t NP = (size_t)(B + 1) / 2, NI = 2 * NP;
    for (int i = 0; i < 2; ++i) {
        A(ws->x[i], NI * K * 256);
        A(ws->cs[i], NI * K * 32);
        A(ws->sn[i], NI * K * 32);
        A(ws->ind[i], NI * K);
    }
    A(ws->q, NI * K * 256); A(ws->k, NI * K * 256); A(ws->v, NI * K * 256);
    A(ws->att, NI * K * 256); A(ws->msg, NI * K * 256); A(ws->h, NI * K * 512);
    A(ws->attn_part, attn_part_floats((int)K, (int)NI, 4)); A(ws->attn_cnt, attn_counter_ints((int)K, (int)NI, 4));
    A(ws->attn_planes, attn_planes_bytes((int)K, (int)NI, 4));
    A(ws->conf, NI * K); A(ws->msc, NI * K); A(ws->keep_idx, NI * K); A(ws->prune, NI * K);
    A(ws->md, NI * K * 256); A(ws->z, NI * K); A(ws->lz, NI * K);
    ws->sim_ps = (((size_t)K * K + (size_t)K + 4) + 3) & ~(size_t)3;     // floats between the score matrices of consecutive pairs
    ws->vec_ps = ((size_t)K + 4 + 3) & ~(size_t)3;
    ws->part_ps = (size_t)((K + 15) / 16 + 1) * (K + 4);
    A(ws->sim, NP * ws->sim_ps + (size_t)K + 8);
    A(ws->sim2, (size_t)1);
    A(ws->rmax, NP * ws->vec_ps); A(ws->rlog, NP * ws->vec_ps); A(ws->cmax, NP * ws->vec_ps); A(ws->clog, NP * ws->vec_ps);
    A(ws->part, NP * ws->part_ps);
    A(ws->ridx, NP * ws->vec_ps); A(ws->rval, NP * ws->vec_ps); A(ws->cbest, NP * ws->vec_ps);
    A(ws->st, NP); A(ws->sel, NP + 3);
    A(ws->uv, (size_t)4 * (K + 8));
#undef A
    ws->n_pairs = (int)NP;
    if (!ok) {
        for (void* p : ws->allocs) ctx->gfree(p);
        delete ws;
        ctx->max_h = ctx->max_w = ctx->max_images = ctx->max_kpts = 0;   // no workspace any more: the next call starts from its own sizes
        return ctx->fail(-31, "im_ctx_reserve: out of device memory (%d x %d, %d images, %d keypoints)", max_h, max_w, max_images, max_kpts);
    }
    ctx->ws = ws;
    ctx->max_h = max_h; ctx->max_w = max_w; ctx->max_images = max_images; ctx->max_kpts = max_kpts;
    IM_HIP(ctx, hipMemset(ws->st, 0, sizeof(LGState) * NP));
    IM_HIP(ctx, hipMemset(ws->sel, 0, (NP + 3) * sizeof(int)));
    IM_HIP(ctx, hipMemset(ws->attn_cnt, 0, attn_counter_ints((int)K, (int)NI, 4) * sizeof(int)));
    IM_HIP(ctx, hipDeviceSynchronize());
    return 0;
}

// ------------------------------------------------------------------------------------------------ SuperPoint
int im_superpoint_forward(im_ctx* ctx, const uint8_t* d_img, int n_images, int h, int w, int channels, int nms_radius,
                          float threshold, int border, int max_kpts, int flavour, float* d_kpts, float* d_scores, float* d_desc,
                          int32_t* d_n, void* stream) {
    IM_CHECK_CTX(ctx);
    // flavour: both select the same candidate set (the border test commutes with the threshold for thr >= 0, tested); it
    // only picks the gray conversion of 3-channel input
    if (channels != 1 && channels != 3 && channels != 4)
        return ctx->fail(-44, "im_superpoint_forward: channels must be 1 (uint8 gray), 3 (uint8 RGB) or 4 (float32 gray), got %d", channels);
    if (!ctx->sp.ready) return ctx->fail(-40, "im_superpoint_forward: weights not finalized");
    Workspace* ws = ctx->ws;
    if (!ws || h > ctx->max_h || w > ctx->max_w || n_images > ctx->max_images)
        return ctx->fail(-41, "im_superpoint_forward: %d x %d x %d exceeds the reserved workspace", n_images, h, w);
    if (h < 8 || w < 8) return ctx->fail(-42, "im_superpoint_forward: image smaller than one cell");
    if (threshold < 0.f) return ctx->fail(-43, "im_superpoint_forward: negative detection threshold unsupported");
    hipStream_t s = (hipStream_t)stream;
    const SuperPointW& W = ctx->sp;
    const int B = n_images, K = ctx->max_kpts;
    // 3x3 layers run in Winograd F(2x2,3x3) form with the products on the bf16 matrix cores (conv_wino.hip BX, six bf16 products per fp32
    // product); IM_CONV_F32=1 selects the same form on the f32-input MFMA (rounds 2-5; read by launch_conv3x3_wino), IM_CONV_DIRECT=1 the direct
    // implicit GEMM (read per call, not cached: the parity tests run all forms in one process; a captured graph keeps the form it was captured with)
    const char* const direct_env = getenv("IM_CONV_DIRECT");
    const bool direct = direct_env && direct_env[0] == '1';
    auto conv = [&](ConvArgs& a, int layer) {
        a.w = direct ? W.cw[layer] : W.cww[layer];
        a.wx = W.cwx[layer];
        a.clock = ctx->clock_of(1);
        return direct ? launch_conv3x3(a, s) : launch_conv3x3_wino(a, s);
    };
    // conv1a is fused into conv1b's patch producer: the full-resolution 64-channel activation never touches HBM
    float* src = nullptr;
    float* dst = ws->act1;
    int ch = h, cw_ = w;
    static const int pool_after[7] = {1, 0, 1, 0, 1, 0, 0};  // conv1b, 2a, 2b, 3a, 3b, 4a, 4b
    for (int i = 0; i < 7; ++i) {
        ConvArgs a;
        a.in = src; a.bias = W.cb[i]; a.out = dst; a.B = B; a.H = ch; a.W = cw_;
        a.Cin = SP_CIN[i]; a.Cout = SP_COUT[i]; a.pool = pool_after[i]; a.relu = 1;
        if (i == 0) { a.img = d_img; a.img_channels = channels; a.gray_mode = flavour == 1 ? 1 : 0; a.w1 = W.c1a_w; a.b1 = W.c1a_b; a.w1q = W.c1a_wq; }
        IM_LAUNCH(ctx, SP_CONV3[i], s, conv(a, i));
        if (pool_after[i]) { ch /= 2; cw_ /= 2; }
        src = dst;
        dst = (dst == ws->act1) ? ws->act0 : ws->act1;
    }
    dst = (src == ws->act1) ? ws->act0 : ws->act1;
    // src = feat [B][hc][wc][128] (in act1), dst = act0 free
    const int hc = ch, wc = cw_;
    const long cells = (long)B * hc * wc;
    float* feat = src;
    float* tmp = dst;
    {
        ConvArgs a;
        a.in = feat; a.bias = W.cb[7]; a.out = tmp; a.B = B; a.H = hc; a.W = wc; a.Cin = 128; a.Cout = 256;
        IM_LAUNCH(ctx, "convPa", s, conv(a, 7));
        GemmArgs g;
        g.A = tmp; g.lda = 256; g.W = W.pb_w; g.ldw = 256; g.bias = W.pb_b; g.N = 65; g.K = 256; g.m_max = (int)cells;
        g.C = ws->logits; g.ldc = 65; g.epi = EPI_BIAS;
        IM_LAUNCH(ctx, "convPb_gemm", s, launch_gemm(g, s));
        IM_LAUNCH(ctx, "det_softmax", s, launch_det_softmax(ws->logits, 65, ws->smap, B, hc, wc, s));
    }
    const int H8 = hc * 8, W8 = wc * 8;
    IM_LAUNCH(ctx, "nms_select", s, launch_nms_select(ws->smap, ws->nms, ws->mask, ws->supp, ws->rest, B, H8, W8, nms_radius, border,
                                                    threshold, max_kpts, K, ws->kpsel, d_kpts, d_scores, d_n, s));
    {
        ConvArgs a;
        a.in = feat; a.bias = W.cb[8]; a.out = tmp; a.B = B; a.H = hc; a.W = wc; a.Cin = 128; a.Cout = 256;
        IM_LAUNCH(ctx, "convDa", s, conv(a, 8));
        GemmArgs g;
        g.A = tmp; g.lda = 256; g.W = W.db_w; g.ldw = 256; g.bias = W.db_b; g.N = 256; g.K = 256; g.m_max = (int)cells;
        g.C = ws->dense; g.ldc = 256; g.epi = EPI_BIAS;
        IM_LAUNCH(ctx, "convDb_gemm", s, launch_gemm(g, s));
        IM_LAUNCH(ctx, "sample_desc", s, launch_sample_desc(ws->dense, B, hc, wc, d_kpts, d_n, K, d_desc, s));
    }
    IM_GUARD_CHECK(ctx, s, "im_superpoint_forward");
    return 0;
}

int im_superpoint_candidates(im_ctx* ctx, int n_images, int32_t* h_counts, void* stream) {
    IM_CHECK_CTX(ctx);
    if (!ctx->ws || !h_counts || n_images < 1 || n_images > ctx->max_images) return ctx->fail(-41, "im_superpoint_candidates: bad arguments");
    IM_HIP(ctx, hipMemcpyAsync(h_counts, ctx->ws->kpsel.n_cand, sizeof(int32_t) * n_images, hipMemcpyDeviceToHost, (hipStream_t)stream));
    IM_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

// ------------------------------------------------------------------------------------------------ LightGlue
static constexpr int ST_INTS = (int)(sizeof(LGState) / sizeof(int));   // ints between the states of consecutive pairs

// The two K = 256 projections of a block run as row blocks (proj_rows_kernel, round 6); IM_PROJ_TILED=1 (read per call) puts them back on the tiled GEMM
// (gemm_nt_kernel, rounds 1-5): same bits, the A/B and test switch
static bool proj_tiled() {
    const char* e = getenv("IM_PROJ_TILED");
    return e && e[0] == '1';
}

static int lg_block(im_ctx* ctx, hipStream_t s, int NI, int layer, bool cross, float* x, const float* cs, const float* sn) {
    Workspace* ws = ctx->ws;
    const LightGlueW& W = ctx->lg;
    const int K = ctx->max_kpts;
    const long xb = (long)K * 256;
    const int* n_ptr = ws->st->n;
    const int* active = &ws->st->active;
    GemmArgs base;
    base.m_max = K; base.m_ptr = n_ptr; base.active = active; base.pstride = ST_INTS; base.batch = NI; base.bx = 1;
    AttnArgs at;
    at.q = ws->q; at.k = cross ? ws->q : ws->k; at.v = ws->v; at.hstride = (long)K * 64; at.bstride = (long)K * 256;
    at.out = ws->att; at.out_bstride = xb; at.ldo = 256; at.n_ptr = n_ptr; at.pstride = ST_INTS; at.n_max = K; at.batch = NI; at.heads = 4;
    at.cross = cross ? 1 : 0; at.active = active; at.part = ws->attn_part; at.counters = ws->attn_cnt; at.planes = ws->attn_planes; at.clock = ctx->clock_of(0);
    if (!cross) {
        GemmArgs g = base;
        g.A = x; g.a_bstride = xb; g.lda = 256; g.W = W.qkv_w + (long)layer * 768 * 256; g.ldw = 256;
        g.bias = W.qkv_b + (long)layer * 768; g.N = 768; g.K = 256; g.epi = EPI_QKV_ROPE;
        g.q = ws->q; g.k = ws->k; g.v = ws->v; g.head_bstride = (long)K * 256; g.head_stride = (long)K * 64;
        g.cs = cs; g.sn = sn; g.enc_bstride = (long)K * 32; g.big_tile = NI >= 4;
        if (proj_tiled()) {
            IM_LAUNCH(ctx, "lg_qkv_rope_gemm", s, launch_gemm(g, s));
        } else {
            g.wp = reinterpret_cast<const unsigned char*>(W.qkv_wp) + (size_t)layer * 768 * 256 * 6;
            IM_LAUNCH(ctx, "lg_qkv_rope_gemm", s, launch_proj_rows(g, s));
        }
        at.scale = 0.125f;  // SDPA default 1/sqrt(64) (`lightglue.py:120-123`)
    } else {
        GemmArgs g = base;
        g.A = x; g.a_bstride = xb; g.lda = 256; g.W = W.cqv_w + (long)layer * 512 * 256; g.ldw = 256;
        g.bias = W.cqv_b + (long)layer * 512; g.N = 512; g.K = 256; g.epi = EPI_HEADS_QV;
        g.q = ws->q; g.v = ws->v; g.head_bstride = (long)K * 256; g.head_stride = (long)K * 64;
        g.alpha = (float)0.35355339059327373;  // scale**0.5 = 64**-0.25 on to_qk (`lightglue.py:201`); to_v unscaled
        g.big_tile = NI >= 4;
        if (proj_tiled()) {
            IM_LAUNCH(ctx, "lg_proj_gemm", s, launch_gemm(g, s));
        } else {
            g.wp = reinterpret_cast<const unsigned char*>(W.cqv_wp) + (size_t)layer * 512 * 256 * 6;
            IM_LAUNCH(ctx, "lg_proj_gemm", s, launch_proj_rows(g, s));
        }
        at.scale = 1.f;
    }
    IM_LAUNCH(ctx, "attn_kv_planes", s, launch_attn_planes(at, s));
    IM_LAUNCH(ctx, cross ? "flash_attn_cross" : "flash_attn_self", s, launch_flash_attn(at, s));
    // A/B switch: IM_FFN_UNFUSED=1 keeps the three-launch form (ffn.0 GEMM, LayerNorm + GELU, ffn.3 GEMM + residual)
    static const bool unfused = getenv("IM_FFN_UNFUSED") && getenv("IM_FFN_UNFUSED")[0] == '1';
    if (!unfused) {   // ffn.0 on cat([x, att]) (out_proj folded into the weights), LayerNorm, GELU, ffn.3, residual: one kernel
        FfnArgs f;
        f.x = x; f.x_bstride = xb; f.att = ws->att; f.att_bstride = xb;
        f.w0p = (cross ? W.cf0_wp : W.sf0_wp) + (long)layer * 512 * 512 * 3 / 2; f.b0 = (cross ? W.cf0_b : W.sf0_b) + (long)layer * 512;
        f.ln_g = (cross ? W.cln_g : W.sln_g) + (long)layer * 512; f.ln_b = (cross ? W.cln_b : W.sln_b) + (long)layer * 512;
        f.w3p = (cross ? W.cf3_wp : W.sf3_wp) + (long)layer * 256 * 512 * 3 / 2; f.b3 = (cross ? W.cf3_b : W.sf3_b) + (long)layer * 256;
        f.m_max = K; f.batch = NI; f.m_ptr = n_ptr; f.active = active; f.pstride = ST_INTS;
        IM_LAUNCH(ctx, "lg_ffn_fused", s, launch_ffn_fused(f, s));
        return 0;
    }
    {   // ffn.0 on cat([x, out_proj(att)]) with out_proj folded into the weights: the second source is the attention output
        GemmArgs g = base;
        g.A = x; g.a_bstride = xb; g.lda = 256; g.A1 = ws->att; g.a1_bstride = xb; g.lda1 = 256; g.ksplit = 256;
        g.W = (cross ? W.cf0_w : W.sf0_w) + (long)layer * 512 * 512; g.ldw = 512;
        g.bias = (cross ? W.cf0_b : W.sf0_b) + (long)layer * 512; g.N = 512; g.K = 512;
        g.C = ws->h; g.c_bstride = (long)K * 512; g.ldc = 512; g.epi = EPI_BIAS;
        IM_LAUNCH(ctx, "lg_ffn0_gemm", s, launch_gemm(g, s));
    }
    IM_LAUNCH(ctx, "lg_layernorm_gelu", s, launch_layernorm_gelu(ws->h, (long)K * 512, ws->st, NI, K, (cross ? W.cln_g : W.sln_g) + (long)layer * 512,
                                                               (cross ? W.cln_b : W.sln_b) + (long)layer * 512, s));
    {   // x += ffn.3(h)
        GemmArgs g = base;
        g.A = ws->h; g.a_bstride = (long)K * 512; g.lda = 512;
        g.W = (cross ? W.cf3_w : W.sf3_w) + (long)layer * 256 * 512; g.ldw = 512;
        g.bias = (cross ? W.cf3_b : W.sf3_b) + (long)layer * 256; g.N = 256; g.K = 512;
        g.C = x; g.c_bstride = xb; g.ldc = 256; g.R = x; g.r_bstride = xb; g.ldr = 256; g.epi = EPI_BIAS_RESID;
        IM_LAUNCH(ctx, "lg_ffn3_gemm", s, launch_gemm(g, s));
    }
    return 0;
}

static int lightglue_forward(im_ctx* ctx, int n_pairs, const float* d_kpts, const float* d_desc, const int32_t* d_n, const float* h_size,
                             const im_lightglue_conf* conf, int32_t* d_matches, float* d_mscores, int32_t* d_prune, int32_t* d_info,
                             void* stream) {
    IM_CHECK_CTX(ctx);
    if (!ctx->lg.ready) return ctx->fail(-50, "im_lightglue_forward: weights not finalized");
    Workspace* ws = ctx->ws;
    if (!ws) return ctx->fail(-51, "im_lightglue_forward: call im_ctx_reserve first");
    if (n_pairs < 1 || n_pairs > ws->n_pairs || n_pairs > 64)
        return ctx->fail(-53, "im_lightglue_forward: %d pairs, the workspace was reserved for %d (max_images / 2)", n_pairs, ws->n_pairs);
    hipStream_t s = (hipStream_t)stream;
    const LightGlueW& W = ctx->lg;
    const int K = ctx->max_kpts;
    const int L = conf->n_layers;
    const int NP = n_pairs, NI = 2 * n_pairs;
    if (L < 1 || L > 9) return ctx->fail(-52, "im_lightglue_forward: n_layers must be 1..9");
    const bool do_stop = conf->depth_confidence > 0, do_prune = conf->width_confidence > 0;
    const long xb = (long)K * 256, eb = (long)K * 32;
    LGState* st = ws->st;

    IM_HIP(ctx, launch_lg_init(st, NI, d_n, ws->ind[0], ws->prune, K, K, d_matches, d_mscores, K, s));
    IM_HIP(ctx, hipMemcpyAsync(ws->x[0], d_desc, sizeof(float) * NI * xb, hipMemcpyDeviceToDevice, s));
    IM_HIP(ctx, launch_posenc(d_kpts, (long)K * 2, st, NI, K, W.wr, h_size, ws->cs[0], ws->sn[0], eb, s));
    int cur = 0;
    for (int i = 0; i < L; ++i) {
        int rc = lg_block(ctx, s, NI, i, false, ws->x[cur], ws->cs[cur], ws->sn[cur]);
        if (rc) return rc;
        rc = lg_block(ctx, s, NI, i, true, ws->x[cur], ws->cs[cur], ws->sn[cur]);
        if (rc) return rc;
        if (i == L - 1) break;
        if (!do_stop && !do_prune) continue;
        IM_LAUNCH(ctx, "lg_adapt", s, launch_rowdot(ws->x[cur], xb, st, NI, K, do_stop ? W.tc_w + (long)i * 256 : nullptr, W.tc_b + i, 1,
                                  do_prune ? W.ma_w + (long)i * 256 : nullptr, W.ma_b + i, nullptr, ws->conf, ws->msc, K,
                                  W.thr[i], do_stop ? i : -1, 1, s));
        // keep threshold: `scores > (1 - width_confidence)` evaluated in double, compared in fp32 (`lightglue.py:566`)
        const float keep_thr = (float)(1.0 - (double)conf->width_confidence);
        IM_LAUNCH(ctx, "lg_adapt", s, launch_stop_prune(st, NP, i, do_stop, do_prune, (float)conf->depth_confidence, keep_thr, W.thr[i], ws->conf,
                                                      ws->msc, K, ws->ind[cur], ws->ind[1 - cur], ws->keep_idx, ws->prune, K,
                                                      conf->pruning_min_kpts, s));
        if (do_prune) {
            IM_LAUNCH(ctx, "lg_adapt", s, launch_gather_rows(st, NI, K, ws->keep_idx, K, ws->x[cur], ws->x[1 - cur], xb, ws->cs[cur], ws->cs[1 - cur],
                                                           ws->sn[cur], ws->sn[1 - cur], eb, do_stop ? i : -1, (float)conf->depth_confidence, s));
            cur = 1 - cur;
        }
    }
    ctx->dbg_cur = cur;
    // ---- assignment with log_assignment[last executed layer] (per pair: a device-side layer index)
    IM_HIP(ctx, launch_lg_select_layer(st, NP, L, ws->sel, d_info, s));
    {
        GemmArgs g;
        g.m_max = K; g.m_ptr = st->n; g.pstride = ST_INTS; g.batch = NI; g.bx = 1;
        g.A = ws->x[cur]; g.a_bstride = xb; g.lda = 256; g.W = W.fp_w; g.ldw = 256; g.bias = W.fp_b;
        g.sel = ws->sel; g.w_sel_stride = 65536; g.bias_sel_stride = 256; g.N = 256; g.K = 256;
        g.C = ws->md; g.c_bstride = xb; g.ldc = 256; g.alpha = 0.25f;  // / 256**0.25 (`lightglue.py:279`)
        g.epi = EPI_BIAS;
        IM_LAUNCH(ctx, "lg_proj_gemm", s, launch_gemm(g, s));
    }
    IM_HIP(ctx, launch_rowdot(ws->x[cur], xb, st, NI, K, W.ma_w, W.ma_b, 0, nullptr, nullptr, ws->sel, ws->z, nullptr, K, 0.f, -1, 0, s));
    IM_HIP(ctx, launch_logsig(ws->z, K, st, NI, K, ws->lz, s));
    {   // one score matrix per pair: md of image 2p against md of image 2p + 1
        GemmArgs g;
        g.m_max = K; g.m_ptr = &st->n[0]; g.n_ptr = &st->n[1]; g.pstride = ST_INTS; g.pair_batched = 1; g.batch = NP; g.bx = 1;
        g.A = ws->md; g.a_bstride = 2 * xb; g.lda = 256; g.W = ws->md + xb; g.w_bstride = 2 * xb; g.ldw = 256; g.N = K; g.K = 256;
        g.C = ws->sim; g.c_bstride = (long)ws->sim_ps; g.ldc = K; g.epi = EPI_BIAS; g.big_tile = 1;
        IM_LAUNCH(ctx, "score_gemm", s, launch_gemm(g, s));
    }
    AssignArgs a;
    a.sim = ws->sim; a.ld = K; a.m_ptr = &st->n[0]; a.n_ptr = &st->n[1]; a.m_max = K; a.n_max = K;
    a.lz0 = ws->lz; a.lz1 = ws->lz + K;
    a.rmax = ws->rmax; a.rlog = ws->rlog; a.cmax = ws->cmax; a.clog = ws->clog; a.part = ws->part;
    a.ridx = ws->ridx; a.rval = ws->rval; a.cbest = ws->cbest; a.threshold = (float)conf->filter_threshold;
    a.ind0 = ws->ind[cur]; a.ind1 = ws->ind[cur] + K;
    a.out_m0 = d_matches; a.out_m1 = d_matches + K; a.out_s0 = d_mscores; a.out_s1 = d_mscores + K;
    a.n_pairs = NP; a.sim_ps = (long)ws->sim_ps; a.vec_ps = (long)ws->vec_ps; a.part_ps = (long)ws->part_ps; a.lz_ps = 2L * K;
    a.out_ps = 2L * K; a.state_ps = ST_INTS;
    IM_LAUNCH(ctx, "assign", s, launch_assign(a, s));
    IM_HIP(ctx, hipMemcpyAsync(d_prune, ws->prune, sizeof(int) * NI * K, hipMemcpyDeviceToDevice, s));
    IM_GUARD_CHECK(ctx, s, "im_lightglue_forward");
    return 0;
}

int im_lightglue_forward(im_ctx* ctx, const float* d_kpts, const float* d_desc, const int32_t* d_n, const float* h_size,
                         const im_lightglue_conf* conf, int32_t* d_matches, float* d_mscores, int32_t* d_prune,
                         int32_t* d_info, void* stream) {
    return lightglue_forward(ctx, 1, d_kpts, d_desc, d_n, h_size, conf, d_matches, d_mscores, d_prune, d_info, stream);
}

int im_lightglue_forward_pairs(im_ctx* ctx, int n_pairs, const float* d_kpts, const float* d_desc, const int32_t* d_n, const float* h_size,
                               const im_lightglue_conf* conf, int32_t* d_matches, float* d_mscores, int32_t* d_prune,
                               int32_t* d_info, void* stream) {
    return lightglue_forward(ctx, n_pairs, d_kpts, d_desc, d_n, h_size, conf, d_matches, d_mscores, d_prune, d_info, stream);
}

int im_pack_record(im_ctx* ctx, const int32_t* d_n, const int32_t* d_matches0, const float* d_mscores0, const int32_t* d_info,
                   int epoch, int32_t* d_record, void* stream) {
    IM_CHECK_CTX(ctx);
    if (!ctx->ws) return ctx->fail(-51, "im_pack_record: call im_ctx_reserve first");
    IM_HIP(ctx, launch_pack_record(d_n, d_matches0, d_mscores0, d_info, epoch, ctx->max_kpts, d_record, 1, nullptr, (hipStream_t)stream));
    return 0;
}

int im_pack_records(im_ctx* ctx, int n_pairs, const int32_t* d_n, const int32_t* d_matches, const float* d_mscores, const int32_t* d_info,
                    int first_epoch, int32_t* d_records, const float* d_kpts, void* stream) {
    IM_CHECK_CTX(ctx);
    if (!ctx->ws || n_pairs < 1) return ctx->fail(-51, "im_pack_records: call im_ctx_reserve first");
    IM_HIP(ctx, launch_pack_record(d_n, d_matches, d_mscores, d_info, first_epoch, ctx->max_kpts, d_records, n_pairs, d_kpts, (hipStream_t)stream));
    return 0;
}

// Copies an internal buffer of the last forward to d_dst (stage-level parity tests). Names: "lg_x" (descriptors after the
// last executed layer, [2][K][256]), "lg_cos" / "lg_sin" (rotary tables [2][K][32]), "sim" ([K][K] score matrix),
// "md" (projected matching descriptors [2][K][256]), "sp_smap" / "sp_nms" (SuperPoint score map before / after simple_nms of
// the last im_superpoint_forward, [n_images][H8][W8] densely packed for that call's size). Synchronises the stream.
int im_debug_read(im_ctx* ctx, const char* name, float* d_dst, size_t nfloats, void* stream) {
    IM_CHECK_CTX(ctx);
    Workspace* ws = ctx->ws;
    if (!ws || !name) return ctx->fail(-51, "im_debug_read: no workspace");
    const size_t K = (size_t)ctx->max_kpts;
    const std::string nm = name;
    const float* src = nullptr;
    size_t avail = 0;
    if (nm == "lg_x") { src = ws->x[ctx->dbg_cur]; avail = 2 * K * 256; }
    else if (nm == "lg_cos") { src = ws->cs[ctx->dbg_cur]; avail = 2 * K * 32; }
    else if (nm == "lg_sin") { src = ws->sn[ctx->dbg_cur]; avail = 2 * K * 32; }
    else if (nm == "sim") { src = ws->sim; avail = K * K; }   // pair 0
    else if (nm == "md") { src = ws->md; avail = 2 * K * 256; }
    else if (nm == "sp_smap") { src = ws->smap; avail = (size_t)ctx->max_images * ((ctx->max_h / 8) * 8) * ((ctx->max_w / 8) * 8); }
    else if (nm == "sp_nms") { src = ws->nms; avail = (size_t)ctx->max_images * ((ctx->max_h / 8) * 8) * ((ctx->max_w / 8) * 8); }
    else return ctx->fail(-61, "im_debug_read: unknown buffer '%s'", name);
    if (nfloats > avail) return ctx->fail(-62, "im_debug_read: %zu floats requested, %zu available", nfloats, avail);
    IM_HIP(ctx, hipMemcpyAsync(d_dst, src, nfloats * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    IM_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

// Sustained shader clock of the two matrix-core kernel classes: arm = 1 allocates / clears the probe words and hands them to
// every attention (attention_bx.hip) and Winograd BX (conv_wino.hip) launch of this context until arm = 0; the kernels' first wave stores the
// shader cycles (s_memtime) and the 100 MHz reference ticks (s_memrealtime) spent in its main loop. Reading (arm = 0 or 2): the median over the
// blocks that wrote of cycles / ticks x 100 MHz, per class; h_out = {attention MHz, attention blocks, convolution MHz, convolution blocks}.
int im_debug_clock_probe(im_ctx* ctx, int arm, double* h_out, void* stream) {
    IM_CHECK_CTX(ctx);
    hipStream_t s = (hipStream_t)stream;
    const size_t bytes = (size_t)CLOCK_PROBE_SLOTS * 2 * sizeof(unsigned long long);
    if (arm == 1) {
        if (!ctx->clock_buf[0]) {
            ctx->clock_buf[0] = ctx->dalloc<unsigned long long>((size_t)CLOCK_PROBE_SLOTS * 2, "clock_probe_attention");
            ctx->clock_buf[1] = ctx->dalloc<unsigned long long>((size_t)CLOCK_PROBE_SLOTS * 2, "clock_probe_convolution");
            if (!ctx->clock_buf[0] || !ctx->clock_buf[1]) return ctx->fail(-22, "im_debug_clock_probe: allocation failed");
        }
        IM_HIP(ctx, hipMemsetAsync(ctx->clock_buf[0], 0, bytes, s));
        IM_HIP(ctx, hipMemsetAsync(ctx->clock_buf[1], 0, bytes, s));
        IM_HIP(ctx, hipStreamSynchronize(s));
        ctx->clock_armed = true;
        return 0;
    }
    if (!ctx->clock_buf[0] || !h_out) return ctx->fail(-51, "im_debug_clock_probe: not armed");
    std::vector<unsigned long long> h((size_t)CLOCK_PROBE_SLOTS * 2);
    unsigned long long* src[2] = {ctx->clock_buf[0], ctx->clock_buf[1]};
    for (int c = 0; c < 2; ++c) {
        IM_HIP(ctx, hipMemcpyAsync(h.data(), src[c], bytes, hipMemcpyDeviceToHost, s));
        IM_HIP(ctx, hipStreamSynchronize(s));
        std::vector<double> mhz;
        for (int i = 0; i < CLOCK_PROBE_SLOTS; ++i)
            if (h[2 * i + 1] > 0) mhz.push_back(100.0 * (double)h[2 * i] / (double)h[2 * i + 1]);
        std::sort(mhz.begin(), mhz.end());
        h_out[2 * c] = mhz.empty() ? 0.0 : mhz[mhz.size() / 2];
        h_out[2 * c + 1] = (double)mhz.size();
    }
    if (arm == 0) ctx->clock_armed = false;       // the buffers stay with the context (a graph captured while armed keeps writing into them, harmlessly)
    return 0;
}

// ------------------------------------------------------------------------------------------------ stage entry points
int im_nms(im_ctx* ctx, const float* d_scores, float* d_out, int n_images, int h, int w, int radius, void* stream) {
    IM_CHECK_CTX(ctx);
    Workspace* ws = ctx->ws;
    if (!ws || (long)n_images * h * w > (long)ctx->max_images * ((ctx->max_h / 8) * 8) * ((ctx->max_w / 8) * 8))
        return ctx->fail(-41, "im_nms: exceeds the reserved workspace");
    IM_HIP(ctx, launch_nms(d_scores, d_out, ws->mask, ws->supp, ws->rest, n_images, h, w, radius, (hipStream_t)stream));
    IM_GUARD_CHECK(ctx, (hipStream_t)stream, "im_nms");
    return 0;
}

int im_select_topk(im_ctx* ctx, const float* d_nms, int n_images, int h, int w, int border, float threshold, int max_kpts,
                   float* d_kpts, float* d_scores, int32_t* d_n, void* stream) {
    IM_CHECK_CTX(ctx);
    Workspace* ws = ctx->ws;
    if (!ws || (long)n_images * h * w > (long)ctx->max_images * ((ctx->max_h / 8) * 8) * ((ctx->max_w / 8) * 8))
        return ctx->fail(-41, "im_select_topk: exceeds the reserved workspace");
    IM_HIP(ctx, launch_select_topk(d_nms, n_images, h, w, border, threshold, max_kpts, ctx->max_kpts, ws->kpsel, d_kpts, d_scores, d_n,
                                   (hipStream_t)stream));
    IM_GUARD_CHECK(ctx, (hipStream_t)stream, "im_select_topk");
    return 0;
}

int im_sample_descriptors(im_ctx* ctx, const float* d_dense_raw, int n_images, int hc, int wc, const float* d_kpts,
                          const int32_t* d_n, float* d_desc, void* stream) {
    IM_CHECK_CTX(ctx);
    if (!ctx->ws) return ctx->fail(-51, "im_sample_descriptors: call im_ctx_reserve first");
    IM_HIP(ctx, launch_sample_desc(d_dense_raw, n_images, hc, wc, d_kpts, d_n, ctx->max_kpts, d_desc, (hipStream_t)stream));
    return 0;
}

int im_assign_from_sim(im_ctx* ctx, const float* d_sim, int m, int n, int ld, const float* d_z0, const float* d_z1,
                       float threshold, int32_t* d_m0, int32_t* d_m1, float* d_ms0, float* d_ms1, void* stream) {
    IM_CHECK_CTX(ctx);
    Workspace* ws = ctx->ws;
    const int K = ctx->max_kpts;
    if (!ws || m > K || n > K || m < 1 || n < 1) return ctx->fail(-41, "im_assign_from_sim: m, n must be 1..max_kpts");
    hipStream_t s = (hipStream_t)stream;
    // reuse the matcher state as the (m, n) holder
    const int mn[2] = {m, n};
    IM_HIP(ctx, hipMemcpyAsync(ws->st->n, mn, sizeof(mn), hipMemcpyHostToDevice, s));
    IM_HIP(ctx, hipStreamSynchronize(s));
    IM_HIP(ctx, hipMemcpyAsync(ws->z, d_z0, sizeof(float) * m, hipMemcpyDeviceToDevice, s));
    IM_HIP(ctx, hipMemcpyAsync(ws->z + K, d_z1, sizeof(float) * n, hipMemcpyDeviceToDevice, s));
    IM_HIP(ctx, launch_logsig(ws->z, K, ws->st, 2, K, ws->lz, s));
    AssignArgs a;
    a.sim = d_sim; a.ld = ld; a.m_ptr = &ws->st->n[0]; a.n_ptr = &ws->st->n[1]; a.m_max = m; a.n_max = n;
    a.lz0 = ws->lz; a.lz1 = ws->lz + K;
    a.rmax = ws->rmax; a.rlog = ws->rlog; a.cmax = ws->cmax; a.clog = ws->clog; a.part = ws->part;
    a.ridx = ws->ridx; a.rval = ws->rval; a.cbest = ws->cbest; a.threshold = threshold;
    a.out_m0 = d_m0; a.out_m1 = d_m1; a.out_s0 = d_ms0; a.out_s1 = d_ms1;
    a.n_pairs = 1;
    IM_HIP(ctx, launch_assign(a, s));
    IM_GUARD_CHECK(ctx, s, "im_assign_from_sim");
    return 0;
}

}  // extern "C"
