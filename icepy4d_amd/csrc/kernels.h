// Internal launch API of libicematch (C++ side; the public boundary is include/icematch.h).
// All pointers are device pointers unless stated. Dynamic sizes (keypoint counts) live in device
// memory so that no launch needs a host round trip: grids are sized for the maximum and blocks
// beyond the live count exit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace im {

// ------------------------------------------------------------------ gemm.hip
enum GemmEpi {
    EPI_BIAS = 0,       // C = alpha * (acc + bias)
    EPI_BIAS_RESID = 1, // C = R + (acc + bias)
    EPI_HEADS = 2,      // head-major store: q[z][col / 64][row][col % 64] = alpha * (acc + bias)
    EPI_QKV_ROPE = 3,   // N = 768 = [q | k | v] x [head][64]; rotary on q, k; head-major stores
    EPI_BIAS_RELU = 4,  // C = max(acc + bias, 0)
    EPI_HEADS_QV = 5,   // N = 512 = [qk | v] x [head][64]: head-major stores to q (scaled by alpha) and v (unscaled)
};

struct GemmArgs {
    // C_z[m][n] = epi(sum_k A_z[m][k] * W[n][k]),  A row-major (K contiguous), W row-major [N][K]
    const float* A = nullptr;  long a_bstride = 0;  int lda = 0;
    const float* A1 = nullptr; long a1_bstride = 0; int lda1 = 0; int ksplit = 0;  // k >= ksplit reads A1[m][k - ksplit]
    const float* W = nullptr;  int ldw = 0; long w_bstride = 0;
    const float* bias = nullptr;
    const int* sel = nullptr; long w_sel_stride = 0; long bias_sel_stride = 0;     // W += sel[pair] * stride (device-side layer index)
    int N = 0, K = 0;
    int m_max = 0;                 // rows the grid covers
    // Batch element z is image (z & 1) of pair (z >> 1) - or, with pair_batched, pair z itself (the score GEMM). Live sizes and
    // flags of a pair sit `pstride` ints apart (2 = a plain int array [n0, n1, n2, ...]; sizeof(LGState) / 4 = the matcher state):
    const int* m_ptr = nullptr;    // live rows: m_ptr[(z >> 1) * pstride + (z & 1)]  (pair_batched: m_ptr[z * pstride]); null => m_max
    const int* n_ptr = nullptr;    // live columns of the score GEMM: n_ptr[z * pstride]; null => N
    const int* active = nullptr;   // per-pair flag active[pair * pstride], 0 => this batch element is a no-op
    int pstride = 2;
    int pair_batched = 0;
    int batch = 1;
    float alpha = 1.f;
    float* C = nullptr; long c_bstride = 0; int ldc = 0;
    const float* R = nullptr; long r_bstride = 0; int ldr = 0;
    float* q = nullptr; float* k = nullptr; float* v = nullptr; long head_bstride = 0; long head_stride = 0;
    const float* cs = nullptr; const float* sn = nullptr; long enc_bstride = 0;    // rotary tables [rows][32]
    int epi = EPI_BIAS;
    int big_tile = 0;              // 128x128 block tile (score GEMM) instead of 64x64
    int bx = 0;                    // the product on the bf16 matrix cores, six bf16 products per fp32 product (gemm.hip BX; the matchers' GEMMs;
                                   // SuperPoint's stay on the f32-input MFMA: its outputs are bit-identical to round 4's)
    const void* wp = nullptr;      // launch_proj_rows only: the rows of W (after the layer offset) as bf16 planes in MFMA-fragment order (pack_frag_weights(W, N, 256))
};
hipError_t launch_gemm(const GemmArgs& a, hipStream_t s);
hipError_t launch_proj_rows(const GemmArgs& a, hipStream_t s);   // gemm.hip: the K = 256 projections of a LightGlue block as row blocks (same bits as launch_gemm with bx)

// ------------------------------------------------------------------ ffn_fused.hip
struct FfnArgs {
    // x_z[m][0..256) += W3 . act(W0 . [x_z[m] | att_z[m]] + b0) + b3   for the live rows of every image z
    int act = 0;                                       // 0: gelu(layernorm(.)) (LightGlue), 1: relu(.) (SuperGlue, BatchNorm folded)
    float* x = nullptr; long x_bstride = 0;            // [z][row][256], updated in place
    const float* att = nullptr; long att_bstride = 0;  // [z][row][256]
    const float* w0p = nullptr; const float* b0 = nullptr;      // W0 [512][512] as bf16 planes packed by pack_frag_weights (6 bytes per weight), bias [512]
    const float* ln_g = nullptr; const float* ln_b = nullptr;   // LayerNorm(512) weight / bias
    const float* w3p = nullptr; const float* b3 = nullptr;      // W3 [256][512] packed, bias [256]
    int m_max = 0; int batch = 1;
    const int* m_ptr = nullptr;    // live rows of image z: m_ptr[(z >> 1) * pstride + (z & 1)]; null => m_max
    const int* active = nullptr;   // per-pair flag active[(z >> 1) * pstride]
    int pstride = 2;
};
hipError_t launch_ffn_fused(const FfnArgs& a, hipStream_t s);
std::vector<float> pack_frag_weights(const float* w, int n, int k);   // host: row-major W[n][k] -> three bf16 planes in MFMA-fragment order (n k 3 / 2 floats of bytes)

// ------------------------------------------------------------------ attention.hip
struct AttnArgs {
    // out[z][row][head * 64 + d] = softmax_j(scale * q_z[head][row] . k_y[head][j]) v_y[head][j],  y = cross ? z ^ 1 : z
    const float* q = nullptr; const float* k = nullptr; const float* v = nullptr;
    long bstride = 0; long hstride = 0;         // [z][head][row][64]
    float* out = nullptr; long out_bstride = 0; int ldo = 256;
    const int* n_ptr = nullptr;                 // live points of image z: n_ptr[(z >> 1) * pstride + (z & 1)]
    int pstride = 2;                            // ints between the states of consecutive pairs (2 = plain array n[batch])
    int n_max = 0; int batch = 2; int heads = 4; int cross = 0;
    float scale = 1.f;
    const int* active = nullptr;                // per-pair flag active[(z >> 1) * pstride]
    // split-KV workspace (optional): when set, (image, head, query block)s with few queries are cut into up to
    // ATTN_MAX_SPLIT key ranges on separate blocks; the last block to finish merges the partials (fixed order)
    float* part = nullptr;       // [ATTN_MAX_SPLIT][batch][heads][n_max][66]  (64 x O, m, l)
    int* counters = nullptr;     // [batch * heads * ceil(n_max / 128)], zero before the first launch, self-resetting
    // attention_bx.hip (optional): K and V of the launch as bf16 triples, written by its own first kernel: [K | V][batch][heads][n_max][3][64] bf16.
    // Without it the attention kernel cuts the fp32 tiles itself, once per 128-query block
    void* planes = nullptr;
    int f32_form = 0;            // 1: attention.hip's kernel on the f32-input MFMA (rounds 1-5) for this launch; IM_ATTN_F32=1 sets it for all
    // im_debug_clock_probe: when set, wave 0 of every block stores (shader cycles, 100 MHz reference ticks) spent in its main loop at
    // clock[2 * (blockIdx.x % CLOCK_PROBE_SLOTS) ..]; no other code reads these words (the sustained clock of the kernel = their ratio x 100 MHz)
    unsigned long long* clock = nullptr;
};
static constexpr int CLOCK_PROBE_SLOTS = 4096;
static constexpr int ATTN_MAX_SPLIT = 4;
inline size_t attn_part_floats(int n_max, int batch, int heads) { return (size_t)ATTN_MAX_SPLIT * batch * heads * n_max * 66; }
inline size_t attn_planes_bytes(int n_max, int batch, int heads) { return (size_t)2 * batch * heads * n_max * 384; }
inline size_t attn_counter_ints(int n_max, int batch, int heads) { return (size_t)batch * heads * ((n_max + 127) / 128); }
hipError_t launch_flash_attn(const AttnArgs& a, hipStream_t s);      // attention_bx.hip unless IM_ATTN_F32=1 (attention.hip: the f32-input MFMA form)
hipError_t launch_attn_planes(const AttnArgs& a, hipStream_t s);     // a.planes <- K / V as bf16 triples; call before launch_flash_attn when a.planes is set
hipError_t launch_flash_attn_bx(const AttnArgs& a, hipStream_t s);   // fp32 accuracy from six bf16 products per fp32 product on the bf16 matrix cores

// ------------------------------------------------------------------ conv.hip
struct ConvArgs {
    const float* in = nullptr;   // NHWC [B][H][W][Cin]
    const float* w = nullptr;    // packed [Cin / 16][9][Cout][16]
    const float* bias = nullptr; // [Cout]
    float* out = nullptr;        // NHWC [B][Ho][Wo][Cout], Ho = pool ? H / 2 : H
    int B = 1, H = 0, W = 0, Cin = 0, Cout = 0;
    int pool = 0;                // fused 2x2 max-pool (floor) after bias + ReLU
    int relu = 1;
    // fused first layer: when img != nullptr the input is conv1a(img / 255) computed on the fly (Cin must be 64)
    const uint8_t* img = nullptr; // uint8 [B][H][W][img_channels]
    int img_channels = 1;         // 1 = gray, 3 = RGB interleaved (converted per pixel on the fly: common.h image_value)
    int gray_mode = 0;            // 3 channels: 0 = float-image weights (LightGlue flavour), 1 = OpenCV uint8 fixed point (SuperGlue)
    const float* w1 = nullptr;    // conv1a weights [9][64]
    const float* b1 = nullptr;    // conv1a bias [64]
    const float* w1q = nullptr;   // conv1a as MFMA A operand: [64 channels][2 lane halves][8] = {bias, tap 1, 3, 5, 7, -, -, -} / {tap 0, 2, 4, 6, 8, -, -, -}
    // Winograd form on the bf16 matrix cores (conv_wino.hip BX, round 6): the same U = G g G^T cut into three bf16 planes per value in
    // MFMA-fragment order (pack_conv3x3_wino_bx). When set, launch_conv3x3_wino runs the six-bf16-products form; a.w may then be null
    // unless the f32-input form is asked for (IM_CONV_F32=1 / f32_form)
    const void* wx = nullptr;
    int f32_form = 0;             // 1: the f32-input MFMA Winograd kernel (rounds 2-5) although wx is set
    unsigned long long* clock = nullptr;   // im_debug_clock_probe (as AttnArgs::clock): the BX kernel's main loop
};
hipError_t launch_conv3x3(const ConvArgs& a, hipStream_t s);
// Winograd F(2x2, 3x3) variant (conv_wino.hip); a.w = weights packed by pack_conv3x3_wino: [Cin / 8][16][Cout][8] and / or
// a.wx = the same as bf16 planes (pack_conv3x3_wino_bx): [Cin / 16][16][Cout / 32][3][64][8] bf16
hipError_t launch_conv3x3_wino(const ConvArgs& a, hipStream_t s);
// the same layer (plain: no fused first layer), bf16 planes, eight waves per block with the U planes shared by two tile groups (conv_wino_bx2.hip)
hipError_t launch_conv3x3_wino_bx2(const ConvArgs& a, hipStream_t s);
// conv1a: u8 gray [B][H][W] -> (x / 255) * w + b, ReLU -> NHWC [B][H][W][64]; w packed [9][64]
hipError_t launch_conv1a(const uint8_t* img, int channels, int gray_mode, const float* w, const float* bias, float* out, int B, int H,
                         int W, hipStream_t s);

}  // namespace im
