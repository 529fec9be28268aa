#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
namespace im {

// device-resident matcher state: live point counts, early-stop flag, per-layer "unconfident" counters
struct LGState {
    int n[2];
    int n_orig[2];
    int active;
    int stop_layer;
    int cnt[16];
};

struct AssignArgs {
    const float* sim = nullptr; int ld = 0;
    const int* m_ptr = nullptr; const int* n_ptr = nullptr; int m_max = 0, n_max = 0;
    const float* lz0 = nullptr; const float* lz1 = nullptr;   // logsigmoid(z)
    float* rmax = nullptr; float* rlog = nullptr; float* cmax = nullptr; float* clog = nullptr;
    float2* part = nullptr;                                   // [ceil(m_max/16)][n_max] strip partials (8 bytes per entry)
    int* ridx = nullptr; float* rval = nullptr; unsigned long long* cbest = nullptr;
    float threshold = 0.1f;
    int mode = 0;   // 0: LightGlue double log-softmax; 1: SuperGlue OT (rmax = u, cmax = v, rlog[0] = norm)
    // batch over pairs (blockIdx.y = pair): element strides between consecutive pairs of every buffer above / below
    int n_pairs = 1;
    long sim_ps = 0, vec_ps = 0, part_ps = 0, lz_ps = 0, out_ps = 0; int state_ps = 0;
    const int* ind0 = nullptr; const int* ind1 = nullptr;     // compact -> original index (null = identity)
    int* out_m0 = nullptr; int* out_m1 = nullptr; float* out_s0 = nullptr; float* out_s1 = nullptr;
};

// n_images = 2 x pairs: batch element b is image b & 1 of pair b >> 1, whose state is st[b >> 1]
hipError_t launch_posenc(const float* kpts, long kp_bstride, const LGState* st, int n_images, int n_max, const float* wr,
                         const float* h_size, float* cs, float* sn, long enc_bstride, hipStream_t s);
hipError_t launch_layernorm_gelu(float* h, long bstride, const LGState* st, int n_images, int n_max, const float* g, const float* be,
                                 hipStream_t s);
hipError_t launch_rowdot(const float* x, long bstride, LGState* st, int n_images, int n_max, const float* w0, const float* b0, int act0,
                         const float* w1, const float* b1, const int* sel, float* out0, float* out1, long out_bstride,
                         float thr, int count_layer, int check_active, hipStream_t s);
hipError_t launch_stop_prune(LGState* st, int n_pairs, int layer, int do_stop, int do_prune, float depth_conf, float keep_thr,
                             float conf_thr, const float* conf, const float* msc, long vec_bstride, const int* ind_cur,
                             int* ind_next, int* keep_idx, int* prune, long idx_bstride, int prune_min, hipStream_t s);
hipError_t launch_gather_rows(LGState* st, int n_images, int n_max, const int* keep_idx, long idx_bstride, const float* x_src,
                              float* x_dst, long x_bstride, const float* cs_src, float* cs_dst, const float* sn_src,
                              float* sn_dst, long enc_bstride, int commit_layer, float depth_conf, hipStream_t s);
hipError_t launch_lg_init(LGState* st, int n_images, const int* n_in, int* ind, int* prune, long idx_bstride, int n_max, int* out_m,
                          float* out_s, long out_bstride, hipStream_t s);
hipError_t launch_lg_select_layer(LGState* st, int n_pairs, int n_layers, int* sel, int* info, hipStream_t s);
hipError_t launch_assign(const AssignArgs& a, hipStream_t s);
hipError_t launch_zero_words(void* p, long nwords, hipStream_t s);   // use instead of hipMemsetAsync inside forwards (see lg_misc.hip)
hipError_t launch_pack_record(const int* n, const int* matches0, const float* mscores0, const int* info, int epoch, int K,
                              int* rec, int n_pairs, const float* kpts, hipStream_t s);
hipError_t launch_logsig(const float* z, long bstride, const LGState* st, int n_images, int n_max, float* lz, hipStream_t s);

}  // namespace im
