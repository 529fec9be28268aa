// Host-side weight handling: tensors arrive under their official state-dict key names (im_set_tensor),
// im_finalize_weights re-packs them for the kernels and uploads.
#include <atomic>
#include "ctx.h"
#include "workspace.h"

#include <cmath>
#include <cstring>

namespace im {

std::vector<float> pack_conv3x3(const float* w, int cout, int cin) {
    std::vector<float> p((size_t)cin * 9 * cout);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int tap = 0; tap < 9; ++tap)
                p[(((size_t)(ci / 16) * 9 + tap) * cout + co) * 16 + (ci % 16)] = w[((size_t)co * cin + ci) * 9 + tap];
    return p;
}

// U = G g G^T per (cout, cin), evaluated in double and rounded once; layout [cin/8][pos = 4 xi + nu][cout][cin % 8]
std::vector<float> pack_conv3x3_wino(const float* w, int cout, int cin) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    std::vector<float> p((size_t)cin * 16 * cout);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float* g = w + ((size_t)co * cin + ci) * 9;
            double t[4][3];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[0 * 3 + j] + G[i][1] * g[1 * 3 + j] + G[i][2] * g[2 * 3 + j];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const double u = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
                    p[(((size_t)(ci / 8) * 16 + (i * 4 + j)) * cout + co) * 8 + (ci % 8)] = (float)u;
                }
        }
    return p;
}

// The same U values (pack_conv3x3_wino's: evaluated in double, rounded once to fp32) cut into three bf16 planes, x = h + m + l with
// h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (round to nearest even each), in the fragment order of v_mfma_f32_32x32x16_bf16's B
// operand: [cin / 16][pos][cout / 32][plane][lane = (cin % 16 / 8) * 32 + cout % 32][cin % 8] bf16 - one plane of one (position, 32 output
// channels, 16 input channels) fragment is 1 KB, lane-linear 16 bytes per lane. Returned as floats holding the bytes (for im_ctx::upload).
static uint16_t wx_bf16(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float wx_f32(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
std::vector<float> pack_conv3x3_wino_bx(const float* w, int cout, int cin) {
    const std::vector<float> u = pack_conv3x3_wino(w, cout, cin);      // [cin/8][16][cout][8]
    const size_t n16 = (size_t)cin * 16 * cout * 3;
    std::vector<float> out((n16 + 1) / 2);
    uint16_t* p = reinterpret_cast<uint16_t*>(out.data());
    const int nct = cout / 32;
    for (int ci = 0; ci < cin; ++ci)
        for (int pos = 0; pos < 16; ++pos)
            for (int co = 0; co < cout; ++co) {
                const float x = u[(((size_t)(ci / 8) * 16 + pos) * cout + co) * 8 + (ci % 8)];
                const uint16_t h = wx_bf16(x);
                const float r1 = x - wx_f32(h);
                const uint16_t m = wx_bf16(r1);
                const uint16_t l = wx_bf16(r1 - wx_f32(m));
                const int lane = ((ci % 16) / 8) * 32 + (co % 32);
                const size_t base = ((((size_t)(ci / 16) * 16 + pos) * nct + co / 32) * 3) * 512 + (size_t)lane * 8 + (ci % 8);
                p[base] = h; p[base + 512] = m; p[base + 1024] = l;
            }
    return out;
}

}  // namespace im

// ------------------------------------------------------------------------------------------------ IM_DEBUG_GUARDS
static constexpr unsigned GUARD_WORDS = 64;                  // 256 bytes on each side
static constexpr unsigned GUARD_PATTERN = 0xA5C3F00Du;
static std::atomic<int> g_guard_failures{0};                  // process-wide tally (im_debug_guard_failures): contexts on several threads

__global__ void guard_fill_kernel(unsigned* lo, unsigned* hi) {
    lo[threadIdx.x] = GUARD_PATTERN ^ threadIdx.x;
    hi[threadIdx.x] = GUARD_PATTERN ^ threadIdx.x;
}

__global__ void guard_check_kernel(unsigned* const* blocks, int* flag) {
    if (blocks[blockIdx.x][threadIdx.x] != (GUARD_PATTERN ^ threadIdx.x)) atomicCAS(flag, 0, (int)blockIdx.x + 1);
}

void* im_ctx::galloc(size_t bytes, const char* name, std::vector<void*>& owner) {
    void* base = nullptr;
    if (!guards_on) {
        if (hipMalloc(&base, bytes + 256) != hipSuccess) return nullptr;
        owner.push_back(base);
        return base;
    }
    const size_t body = (bytes + 15) & ~(size_t)15;          // the guard behind starts at the first 16-byte boundary past the buffer
    if (hipMalloc(&base, 256 + body + 256 + 256) != hipSuccess) return nullptr;   // (+ the slack every allocation of the library has)
    owner.push_back(base);
    Guard g{base, reinterpret_cast<unsigned*>(base), reinterpret_cast<unsigned*>(static_cast<char*>(base) + 256 + body), name};
    guard_fill_kernel<<<1, GUARD_WORDS, 0, nullptr>>>(g.lo, g.hi);
    hipStreamSynchronize(nullptr);
    guards.push_back(g);
    guards_dirty = true;
    return static_cast<char*>(base) + 256;
}

void im_ctx::dfree(void* user_ptr) {
    if (!user_ptr) return;
    void* base = guards_on ? static_cast<char*>(user_ptr) - 256 : user_ptr;
    for (size_t i = 0; i < allocs.size(); ++i)
        if (allocs[i] == base) { allocs.erase(allocs.begin() + i); gfree(base); return; }
}

void im_ctx::gfree(void* base) {
    if (guards_on)
        for (size_t i = 0; i < guards.size(); ++i)
            if (guards[i].base == base) { guards.erase(guards.begin() + i); guards_dirty = true; break; }
    hipFree(base);
}

int im_ctx::guards_check(hipStream_t s, const char* where) {
    if (!guards_on || guards.empty()) return 0;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    hipStreamIsCapturing(s, &cap);
    const bool capturing = cap != hipStreamCaptureStatusNone;
    // Inside a graph capture nothing is recorded: a captured check kernel would keep the table pointer and the block count of the capture
    // for every later replay, while buffers may have come and gone since. The caller checks from the host after the replay
    // (`im_debug_guards_check`; icepy4d_amd/sequence.py does under IM_DEBUG_GUARDS=1), and im_ctx_destroy checks in any case.
    if (capturing) return 0;
    if (guards_dirty) {
        // a FRESH table per change (a few hundred pointers; debugging mode only): no synchronisation with check kernels that may still read
        // the previous one on another stream - it is retired, not freed, until the context goes
        std::vector<unsigned*> h;
        for (const Guard& g : guards) { h.push_back(g.lo); h.push_back(g.hi); }
        if (d_guard_blocks) retired_guard_tables.push_back(d_guard_blocks);
        d_guard_blocks = nullptr;
        guard_table_cap = h.size();
        if (hipMalloc((void**)&d_guard_blocks, guard_table_cap * sizeof(unsigned*)) != hipSuccess) return fail(-91, "IM_DEBUG_GUARDS: table allocation failed");
        if (!d_guard_flag) {
            if (hipMalloc((void**)&d_guard_flag, sizeof(int)) != hipSuccess) return fail(-91, "IM_DEBUG_GUARDS: flag allocation failed");
            hipMemset(d_guard_flag, 0, sizeof(int));
        }
        hipMemcpy(d_guard_blocks, h.data(), h.size() * sizeof(unsigned*), hipMemcpyHostToDevice);
        guards_dirty = false;
    }
    guard_check_kernel<<<(unsigned)(2 * guards.size()), GUARD_WORDS, 0, s>>>(d_guard_blocks, d_guard_flag);
    int flag = 0;
    if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(&flag, d_guard_flag, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
        return fail(-92, "IM_DEBUG_GUARDS: reading the flag failed after %s", where);
    if (!flag) return 0;
    hipMemset(d_guard_flag, 0, sizeof(int));
    ++g_guard_failures;
    const Guard& g = guards[(size_t)(flag - 1) / 2];
    fprintf(stderr, "IM_DEBUG_GUARDS: guard words %s buffer '%s' were overwritten (seen after %s)\n", (flag - 1) % 2 ? "BEHIND" : "IN FRONT OF",
            g.name.c_str(), where);
    return fail(-90, "IM_DEBUG_GUARDS: guard words %s buffer '%s' were overwritten (seen after %s)", (flag - 1) % 2 ? "behind" : "in front of",
                g.name.c_str(), where);
}

extern "C" int im_debug_guard_failures(void) { return g_guard_failures.load(); }

extern "C" int im_debug_guards_check(im_ctx* ctx, void* stream) {
    if (!ctx) return -1;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(-3, "hipSetDevice failed");
    return ctx->guards_check((hipStream_t)stream, "im_debug_guards_check (after a graph replay)");
}

__global__ void guard_poke_kernel(unsigned* word, unsigned value) { *word = value; }

// Self-test of the mechanism: one stray 4-byte store right behind the first workspace buffer (what an off-by-one row of a kernel
// would do) must fail the check with -90; the word is restored and the tally decremented, so a passing self-test leaves no trace.
extern "C" int im_debug_guard_selftest(im_ctx* ctx, void* stream) {
    if (!ctx) return -1;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(-3, "hipSetDevice failed");
    if (!ctx->guards_on) return ctx->fail(-93, "im_debug_guard_selftest: the context was created without IM_DEBUG_GUARDS=1");
    if (ctx->guards.empty()) return ctx->fail(-93, "im_debug_guard_selftest: nothing allocated yet (call im_ctx_reserve first)");
    hipStream_t s = (hipStream_t)stream;
    int rc = ctx->guards_check(s, "im_debug_guard_selftest (before)");
    if (rc) return rc;
    unsigned* w = ctx->guards.back().hi;                      // first word behind the newest buffer
    guard_poke_kernel<<<1, 1, 0, s>>>(w, 0xDEADBEEFu);
    rc = ctx->guards_check(s, "im_debug_guard_selftest (the stray store is deliberate)");
    guard_poke_kernel<<<1, 1, 0, s>>>(w, GUARD_PATTERN ^ 0u);
    hipStreamSynchronize(s);
    if (rc != -90) return ctx->fail(-94, "im_debug_guard_selftest: a stray store behind '%s' went unnoticed", ctx->guards.back().name.c_str());
    --g_guard_failures;
    return ctx->guards_check(s, "im_debug_guard_selftest (after)");
}

void im_ctx::free_all() {
    if (guards_on) {
        hipDeviceSynchronize();
        guards_check(nullptr, "im_ctx_destroy");
    }
    for (void* p : allocs) gfree(p);
    allocs.clear();
    for (auto& kv : model_allocs)
        for (void* p : kv.second) gfree(p);
    model_allocs.clear();
    if (ws) {
        for (void* p : ws->allocs) gfree(p);
        delete ws;
        ws = nullptr;
    }
    if (d_guard_blocks) hipFree(d_guard_blocks);
    for (unsigned** t : retired_guard_tables) hipFree(t);
    retired_guard_tables.clear();
    if (d_guard_flag) hipFree(d_guard_flag);
    d_guard_blocks = nullptr; d_guard_flag = nullptr; guard_table_cap = 0;
    delete merge;
    merge = nullptr;
}
