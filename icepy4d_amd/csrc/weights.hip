// Host-side weight handling: tensors arrive under their official state-dict key names (im_set_tensor),
// im_finalize_weights re-packs them for the kernels and uploads.
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

}  // namespace im

void im_ctx::free_all() {
    for (void* p : allocs) hipFree(p);
    allocs.clear();
    for (auto& kv : model_allocs)
        for (void* p : kv.second) hipFree(p);
    model_allocs.clear();
    if (ws) {
        for (void* p : ws->allocs) hipFree(p);
        delete ws;
        ws = nullptr;
    }
    delete merge;
    merge = nullptr;
}
