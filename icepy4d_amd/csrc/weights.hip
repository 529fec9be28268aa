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
}
