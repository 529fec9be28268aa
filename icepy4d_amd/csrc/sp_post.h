#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
namespace im {
// buffers of the keypoint selection, per context: n_cand = [B] candidate counters followed by the per-image radix state
// (sel_state_bytes(B) bytes in all, zeroed by the launchers), keys / ties = [B][H8 * W8] candidate keys, chosen = [B][kmax]
struct SelBuffers {
    int* n_cand = nullptr;
    unsigned long long* keys = nullptr;
    unsigned long long* ties = nullptr;
    unsigned long long* chosen = nullptr;
    size_t ties_cap = 0;   // entries of `ties`: also the per-tile staging of the candidate keys (32 x 64 entries per NMS tile), while it is free
};
size_t sel_state_bytes(int B);
hipError_t ensure_dyn_lds(const void* fn, size_t bytes, size_t* cache);   // cache: static size_t [IM_MAX_DEVICES] of the call site
hipError_t launch_det_softmax(const float* logits, int ld, float* smap, int B, int hc, int wc, hipStream_t s);
hipError_t launch_nms(const float* s, float* out, uint8_t* mask, uint8_t* supp, float* rest, int B, int H, int W, int r, hipStream_t st);
hipError_t launch_nms_select(const float* s, float* nms_out, uint8_t* mask, uint8_t* supp, float* rest, int B, int H, int W, int r,
                             int border, float thr, int k_req, int kmax, const SelBuffers& sb, float* kpts, float* scores, int* n_out,
                             hipStream_t st);
hipError_t launch_select_topk(const float* nms, int B, int H, int W, int border, float thr, int k_req, int kmax, const SelBuffers& sb,
                              float* kpts, float* scores, int* n_out, hipStream_t st);
hipError_t launch_sample_desc(const float* dense, int B, int hc, int wc, const float* kpts, const int* n_ptr, int kmax,
                              float* desc, hipStream_t st);
}  // namespace im
