// Context object behind the C ABI: host copies of the state-dict tensors, packed device weights, workspaces.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/icematch.h"
#include "kernels.h"

namespace im {

struct MergeScratch {                     // im_merge_tile_matches (tile_merge.hip); device memory owned by im_ctx::allocs
    unsigned long long* key = nullptr;    // (ordered bits of x) << 32 | ordered bits of y, of mkpts0
    unsigned long long* skey = nullptr;   // the same for first occurrences, ~0 for later duplicates
    unsigned* seq = nullptr;              // p * K + i: the position in the reference's concatenation order
    int* count = nullptr;                 // [0] entries collected, [1] unique rows
    size_t cap = 0;
};
std::vector<float> pack_conv3x3(const float* w, int cout, int cin);       // [cout][cin][3][3] -> [cin/16][9][cout][16]
std::vector<float> pack_conv3x3_wino(const float* w, int cout, int cin);  // -> G g G^T as [cin/8][16][cout][8]
std::vector<float> pack_conv3x3_wino_bx(const float* w, int cout, int cin);  // -> the same values as three bf16 planes in MFMA-fragment order (bytes in floats)

struct SuperPointW {
    bool ready = false;
    float* c1a_w = nullptr; float* c1a_b = nullptr;          // [9][64], [64]
    float* c1a_wq = nullptr;                                 // [64][2][8]: conv1a weights + bias in the k order of the resident-patch form (conv_wino.hip)
    float* cw[10] = {nullptr}; float* cb[10] = {nullptr};     // conv1b..conv4b, convPa, convDa (packed slabs)
    float* cww[10] = {nullptr};                               // the same layers, Winograd-transformed weights
    float* cwx[10] = {nullptr};                               // the same again as bf16 planes (conv_wino.hip BX)
    float* pb_w = nullptr; float* pb_b = nullptr;             // convPb [65][256], [65]
    float* db_w = nullptr; float* db_b = nullptr;             // convDb [256][256], [256]
};

struct LightGlueW {
    bool ready = false;
    float* wr = nullptr;                    // posenc.Wr [32][2]
    // per layer (9): contiguous blocks so that a device-side layer index can select them
    float* qkv_w = nullptr; float* qkv_b = nullptr;      // [L][768][256] rows permuted to [q|k|v][head][d], [L][768]
    float* sf0_w = nullptr; float* sf0_b = nullptr;      // self ffn.0 [L][512][512] with out_proj folded into columns 256..511
    float* sln_g = nullptr; float* sln_b = nullptr;      // [L][512]
    float* sf3_w = nullptr; float* sf3_b = nullptr;      // [L][256][512]
    float* cqv_w = nullptr; float* cqv_b = nullptr;      // cross [to_qk ; to_v] [L][512][256]
    float* cf0_w = nullptr; float* cf0_b = nullptr;      // cross ffn.0 with to_out folded in
    float* cln_g = nullptr; float* cln_b = nullptr;
    float* cf3_w = nullptr; float* cf3_b = nullptr;
    float* qkv_wp = nullptr; float* cqv_wp = nullptr;    // the two projections again as bf16 planes in MFMA-fragment order, per layer (gemm.hip proj_rows_kernel)
    float* sf0_wp = nullptr; float* sf3_wp = nullptr;    // the four FFN matrices again in MFMA-fragment order (ffn_fused.hip)
    float* cf0_wp = nullptr; float* cf3_wp = nullptr;
    float* fp_w = nullptr;  float* fp_b = nullptr;       // log_assignment.final_proj [L][256][256], [L][256]
    float* ma_w = nullptr;  float* ma_b = nullptr;       // matchability [L][256], [L]
    float* tc_w = nullptr;  float* tc_b = nullptr;       // token_confidence [L-1][256], [L-1]
    float thr[16] = {0};                                 // confidence_thresholds (host)
};

struct SuperGlueW {
    bool ready = false;
    float* kenc_w[5] = {nullptr}; float* kenc_b[5] = {nullptr};  // BN folded; layer 0 K padded 3 -> 32
    float* proj_w = nullptr; float* proj_b = nullptr;    // [18][3][256][256] head-major output rows, [18][3][256]
    float* proj_wp = nullptr;                            // proj_w again as bf16 planes in MFMA-fragment order, per layer (gemm.hip proj_rows_kernel)
    float* mlp0_w = nullptr; float* mlp0_b = nullptr;    // [18][512][512] BN folded, second half of K head-major-agnostic
    float* mlp3_w = nullptr; float* mlp3_b = nullptr;    // [18][256][512]
    float* mlp0_wp = nullptr; float* mlp3_wp = nullptr;  // the same matrices in MFMA-fragment order (ffn_fused.hip)
    float* fp_w = nullptr; float* fp_b = nullptr;        // final_proj [256][256]
    float bin_score = 1.f;
};

}  // namespace im

struct im_ctx {
    int device = 0;
    std::string err;
    std::map<std::string, std::vector<float>> host_w;  // "model/key" -> data
    std::vector<void*> allocs;                          // everything hipMalloc'ed through dalloc (freed by free_all)
    std::map<std::string, std::vector<void*>> model_allocs;  // weight buffers per model (freed when a model is reloaded)
    std::vector<void*>* cur_model = nullptr;
    im::SuperPointW sp;
    im::LightGlueW lg;
    im::SuperGlueW sg;

    // optional per-launch timing (HIP events on the launch stream), see im_profile_begin / im_profile_end
    struct ProfEntry { const char* name; hipEvent_t e0, e1; };
    bool prof_on = false;
    hipStream_t prof_stream = nullptr;   // stream of the last timed launch (event-overhead calibration in im_profile_end)
    std::vector<ProfEntry> prof;
    std::vector<hipEvent_t> prof_pool;
    hipEvent_t prof_event() {
        if (!prof_pool.empty()) { hipEvent_t e = prof_pool.back(); prof_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        hipEventCreate(&e);
        return e;
    }

    // reserved workspace
    int max_h = 0, max_w = 0, max_images = 0, max_kpts = 0;
    struct Workspace* ws = nullptr;
    float* stage_attn_part = nullptr; int* stage_attn_cnt = nullptr; size_t stage_attn_floats = 0, stage_attn_ints = 0;  // im_flash_attn
    unsigned char* stage_attn_planes = nullptr; size_t stage_attn_plane_bytes = 0;                                       // im_flash_attn (attention_bx.hip)
    im::MergeScratch* merge = nullptr;   // scratch of im_merge_tile_matches (tile_merge.hip), grown on demand
    unsigned long long* clock_buf[2] = {nullptr, nullptr};   // im_debug_clock_probe: per-block (cycles, 100 MHz ticks) of the attention / Winograd BX main loops
    bool clock_armed = false;
    unsigned long long* clock_of(int cls) const { return clock_armed ? clock_buf[cls] : nullptr; }
    int dbg_cur = 0;  // which ping-pong descriptor buffer the last LightGlue forward ended in (im_debug_read)

    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
    // ---- IM_DEBUG_GUARDS=1 (read when the context is created; debugging aid, GPU AddressSanitizer is not available on this
    // pool): every device buffer the library allocates gets 256 bytes of guard words in front of it and behind it; a small kernel
    // compares them after every forward (and before a buffer is freed), a changed word fails the call with -90 naming the buffer.
    struct Guard { void* base; unsigned* lo; unsigned* hi; std::string name; };
    bool guards_on = false;
    std::vector<Guard> guards;
    bool guards_dirty = false;            // the device-side table of guard blocks is older than `guards`
    unsigned** d_guard_blocks = nullptr;  // [2 * guards.size()] device pointers: lo, hi of every buffer
    int* d_guard_flag = nullptr;          // 0, or 1 + index of the first changed block seen
    size_t guard_table_cap = 0;
    std::vector<unsigned**> retired_guard_tables;   // older tables stay valid until free_all: a check kernel still in flight may read them
    void* galloc(size_t bytes, const char* name, std::vector<void*>& owner);   // hipMalloc (+ guards); owner gets the base pointer
    void gfree(void* base);                                                    // hipFree (+ forget its guards)
    void dfree(void* user_ptr);   // a buffer dalloc'ed outside a model (scratch that is being replaced by a larger one): out of `allocs`, gfree
    int guards_check(hipStream_t s, const char* where);                        // 0 ok / not enabled; < 0 with `err` set

    template <typename T>
    T* dalloc(size_t n, const char* name = "dalloc") {
        return reinterpret_cast<T*>(galloc(n * sizeof(T), name, cur_model ? *cur_model : allocs));
    }
    float* upload(const std::vector<float>& v) {
        float* p = dalloc<float>(v.size());
        if (p && hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        return p;
    }
    void free_all();
};

// at the end of a forward: compare the guard words (no-op unless IM_DEBUG_GUARDS=1)
#define IM_GUARD_CHECK(ctx, stream, where)                                   \
    do {                                                                     \
        if ((ctx)->guards_on) {                                              \
            const int _g = (ctx)->guards_check((stream), (where));           \
            if (_g) return _g;                                               \
        }                                                                    \
    } while (0)

#define IM_CHECK_CTX(ctx)                                \
    do {                                                 \
        if (!(ctx)) return -1;                           \
        if (hipSetDevice((ctx)->device) != hipSuccess) return (ctx)->fail(-3, "hipSetDevice failed"); \
    } while (0)

// launch wrapper: when profiling is on, brackets the launch with two events on its stream
#define IM_LAUNCH(ctx, name, stream, expr)                                   \
    do {                                                                     \
        if ((ctx)->prof_on) {                                                \
            im_ctx::ProfEntry _pe{name, (ctx)->prof_event(), (ctx)->prof_event()}; \
            (ctx)->prof_stream = (stream);                                   \
            hipEventRecord(_pe.e0, (stream));                                \
            hipError_t _le = (expr);                                         \
            hipEventRecord(_pe.e1, (stream));                                \
            (ctx)->prof.push_back(_pe);                                      \
            IM_HIP(ctx, _le);                                                \
        } else {                                                             \
            IM_HIP(ctx, expr);                                               \
        }                                                                    \
    } while (0)

#define IM_HIP(ctx, expr)                                                                                   \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess)                                                                               \
            return (ctx)->fail(-100 - (int)_e, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
    } while (0)
