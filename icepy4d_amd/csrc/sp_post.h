#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
namespace im {
hipError_t launch_det_softmax(const float* logits, int ld, float* smap, int B, int hc, int wc, hipStream_t s);
hipError_t launch_nms(const float* s, float* out, uint8_t* mask, uint8_t* supp, float* rest, int B, int H, int W, int r, hipStream_t st);
hipError_t launch_select_topk(const float* nms, int B, int H, int W, int border, float thr, int k_req, int kmax,
                              int* counts, int* n_cand, unsigned long long* keys, float* kpts, float* scores,
                              int* n_out, hipStream_t st);
hipError_t launch_sample_desc(const float* dense, int B, int hc, int wc, const float* kpts, const int* n_ptr, int kmax,
                              float* desc, hipStream_t st);
}  // namespace im
