#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>
#include <vector>

#include "lg_misc.h"
#include "sp_post.h"

// device workspace of one context, sized by im_ctx_reserve
struct Workspace {
    std::vector<void*> allocs;
    // SuperPoint
    float* act0 = nullptr; float* act1 = nullptr;       // NHWC ping-pong
    float* logits = nullptr; float* dense = nullptr;    // [cells][65], [cells][256]
    float* smap = nullptr; float* nms = nullptr; float* rest = nullptr;
    uint8_t* mask = nullptr; uint8_t* supp = nullptr;
    im::SelBuffers kpsel;   // keypoint selection: candidate counters + radix state, candidate keys, tie list, chosen keys
    // LightGlue / SuperGlue
    float* x[2] = {nullptr, nullptr}; float* cs[2] = {nullptr, nullptr}; float* sn[2] = {nullptr, nullptr};
    int* ind[2] = {nullptr, nullptr};
    float* q = nullptr; float* k = nullptr; float* v = nullptr;
    float* att = nullptr; float* msg = nullptr; float* h = nullptr;
    float* attn_part = nullptr; int* attn_cnt = nullptr;   // split-KV partials and block counters (attention.hip)
    unsigned char* attn_planes = nullptr;                  // K / V of one attention launch as bf16 triples (attention_bx.hip)
    float* conf = nullptr; float* msc = nullptr; int* keep_idx = nullptr; int* prune = nullptr;
    float* md = nullptr; float* z = nullptr; float* lz = nullptr;
    float* sim = nullptr; float* sim2 = nullptr;
    float* rmax = nullptr; float* rlog = nullptr; float* cmax = nullptr; float* clog = nullptr;
    float2* part = nullptr; int* ridx = nullptr; float* rval = nullptr; unsigned long long* cbest = nullptr;
    im::LGState* st = nullptr; int* sel = nullptr;
    float* uv = nullptr;  // Sinkhorn potentials
    int n_pairs = 1;      // pair capacity of the matcher buffers (reserved max_images / 2)
    size_t sim_ps = 0, vec_ps = 0, part_ps = 0;   // element strides between consecutive pairs (score matrix, per-row vectors, strip partials)
};
