// Inference post-processing of the one-stage detectors and the RCNN head as HIP kernels
// (models/det/retinanet.py:172-209, fcos.py:181-216, layers/head/rcnn.py:84-93, faster_rcnn.py:98-131,
// layers/common/post_processing.py:50-103): scores, candidate decode after the per-level top-k (bd_segment_topk),
// and the final gather + rescale + clip after bd_nms_batched.  Compiled with -ffp-contract=off like the other box ops.
#pragma clang fp contract(off)
#include "box_dev.h"

namespace {

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// scores[r*K + k] = sigmoid(logit)  or  sqrt(sigmoid(logit) * sigmoid(ctr[r]))   (fcos.py:194)
__global__ __launch_bounds__(256) void det_scores_kernel(const bf16_raw* __restrict__ logits, const bf16_raw* __restrict__ ctr,
                                                         int ctr_ld, int ctr_off, long long rows, int K, float* __restrict__ scores) {
    const long long total = rows * K;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        float s = sigmoid_f(bf2f(logits[i]));
        if (ctr) s = sqrtf(s * sigmoid_f(bf2f(ctr[(i / K) * ctr_ld + ctr_off])));
        scores[i] = s;
    }
}

// softmax over K+1 logits, background column dropped (rcnn.py:86): scores [R][K]; boxes [R][K][4] = decode(roi, deltas_k)
__global__ __launch_bounds__(256) void rcnn_predict_kernel(const bf16_raw* __restrict__ raw, int ld, int K, int box_off,
                                                           const float* __restrict__ rois, const int* __restrict__ num_rois,
                                                           int rois_per_img, int R, Coder coder, float* __restrict__ scores,
                                                           float* __restrict__ boxes) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const bool valid = (r % rois_per_img) < num_rois[r / rois_per_img];
    const bf16_raw* rp = raw + (long long)r * ld;
    float mx = -INFINITY;
    for (int c = lane; c <= K; c += 64) mx = fmaxf(mx, bf2f(rp[c]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float se = 0.f;
    for (int c = lane; c <= K; c += 64) se += expf(bf2f(rp[c]) - mx);
    se = wave_sum(se);
    const Box roi = ld_box(rois + r * 4ll);
    for (int c = lane; c < K; c += 64) {
        const float p = expf(bf2f(rp[c + 1]) - mx) / se;
        scores[(long long)r * K + c] = valid ? p : -INFINITY;
        const bf16_raw* dp = rp + box_off + c * 4;
        const f32x4_t d = {bf2f(dp[0]), bf2f(dp[1]), bf2f(dp[2]), bf2f(dp[3])};
        *reinterpret_cast<f32x4_t*>(boxes + ((long long)r * K + c) * 4) = decode_dev(roi, d, coder);
    }
}

struct CandLevels { int row_off[BD_MAX_SEGS]; int L; };

// mode 0: BoxCoder.decode(anchor, offsets) (retinanet.py:195-196); mode 1: PointCoder.decode (fcos.py:206-207);
// mode 2: boxes precomputed per item (RCNN: item = roi*K + class)
__global__ __launch_bounds__(256) void det_candidates_kernel(int mode, const int* __restrict__ topk_idx, const float* __restrict__ topk_score,
                                                             const int* __restrict__ topk_cnt, CandLevels lv, int k, int K,
                                                             const float* __restrict__ anchors, const bf16_raw* __restrict__ offsets,
                                                             int off_ld, int A, Coder coder, const float* __restrict__ item_boxes,
                                                             float* __restrict__ boxes, float* __restrict__ scores,
                                                             int* __restrict__ labels) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= lv.L * k) return;
    const int l = c / k, rnk = c - l * k;
    f32x4_t b = {0.f, 0.f, 0.f, 0.f};
    float sc = -INFINITY;
    int lab = 0;
    if (rnk < topk_cnt[l]) {
        const int idx = topk_idx[c];
        sc = topk_score[c];
        lab = idx % K;
        const long long row = (long long)lv.row_off[l] + idx / K;      // anchor / point / roi index
        if (mode == 2) {
            b = *reinterpret_cast<const f32x4_t*>(item_boxes + (row * K + lab) * 4);
        } else {
            const long long pix = row / A;
            const int a = (int)(row - pix * A);
            const bf16_raw* dp = offsets + pix * off_ld + a * 4;
            const f32x4_t d = {bf2f(dp[0]), bf2f(dp[1]), bf2f(dp[2]), bf2f(dp[3])};
            if (mode == 0) b = decode_dev(ld_box(anchors + row * 4), d, coder);
            else {
                const float px = anchors[row * 2], py = anchors[row * 2 + 1];
                b = (f32x4_t){px - d[0], py - d[1], px + d[2], py + d[3]};        // structures/boxcoder.py:135-141
            }
        }
    }
    *reinterpret_cast<f32x4_t*>(boxes + c * 4ll) = b;
    scores[c] = sc;
    labels[c] = lab;
}

// post_processing.py:93-101: gather the NMS survivors, scale to the original image size, clip
__global__ __launch_bounds__(256) void det_finalize_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                           const int* __restrict__ labels, const int* __restrict__ keep,
                                                           const int* __restrict__ num_keep, int max_out,
                                                           const float* __restrict__ im_info, float* __restrict__ out_boxes,
                                                           float* __restrict__ out_scores, int* __restrict__ out_labels) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= max_out) return;
    f32x4_t b = {0.f, 0.f, 0.f, 0.f};
    float s = 0.f;
    int l = -1;
    if (j < num_keep[0]) {
        const int i = keep[j];
        b = *reinterpret_cast<const f32x4_t*>(boxes + i * 4ll);
        const float sh = im_info[2] / im_info[0], sw = im_info[3] / im_info[1];
        const float h = im_info[2], w = im_info[3];
        b[0] = fminf(fmaxf(b[0] * sw, 0.f), w); b[1] = fminf(fmaxf(b[1] * sh, 0.f), h);
        b[2] = fminf(fmaxf(b[2] * sw, 0.f), w); b[3] = fminf(fmaxf(b[3] * sh, 0.f), h);
        s = scores[i];
        l = labels[i];
    }
    *reinterpret_cast<f32x4_t*>(out_boxes + j * 4ll) = b;
    out_scores[j] = s;
    out_labels[j] = l;
}

}  // namespace

extern "C" int bd_det_scores(const void* logits, const void* ctr, int ctr_ld, int ctr_off, int64_t rows, int K, float* scores,
                             bd_stream_t stream) {
    BD_REQUIRE(logits && scores && K > 0 && rows >= 0, "det_scores: bad arguments");
    if (rows == 0) return BD_OK;
    long long g = cdiv64(rows * K, 256);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(det_scores_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)logits,
                       (const bf16_raw*)ctr, ctr_ld, ctr_off, (long long)rows, K, scores);
    BD_CHECK_LAUNCH("bd_det_scores");
    return BD_OK;
}

extern "C" int bd_rcnn_predict(const void* raw, int ld, int K, int box_off, const float* rois, const int32_t* num_rois,
                               int rois_per_img, int R, const float* mean4_host, const float* std4_host, float* scores, float* boxes,
                               bd_stream_t stream) {
    BD_REQUIRE(raw && rois && num_rois && scores && boxes, "rcnn_predict: null pointer");
    BD_REQUIRE(K > 0 && box_off >= K + 1 && box_off + 4 * K <= ld && rois_per_img > 0, "rcnn_predict: bad channel layout");
    if (R == 0) return BD_OK;
    hipLaunchKernelGGL(rcnn_predict_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)raw, ld, K, box_off,
                       rois, num_rois, rois_per_img, R, make_coder(mean4_host, std4_host), scores, boxes);
    BD_CHECK_LAUNCH("bd_rcnn_predict");
    return BD_OK;
}

extern "C" int bd_det_candidates(int mode, const int32_t* topk_idx, const float* topk_score, const int32_t* topk_cnt, int L, int k,
                                 const int32_t* lvl_row_off_host, int K, const float* anchors, const void* offsets, int off_ld, int A,
                                 const float* mean4_host, const float* std4_host, const float* item_boxes, float* boxes, float* scores,
                                 int32_t* labels, bd_stream_t stream) {
    BD_REQUIRE(topk_idx && topk_score && topk_cnt && lvl_row_off_host && boxes && scores && labels, "det_candidates: null pointer");
    BD_REQUIRE(mode >= 0 && mode <= 2 && L > 0 && L <= BD_MAX_SEGS && k > 0 && K > 0 && A > 0, "det_candidates: bad sizes");
    BD_REQUIRE(mode == 2 ? item_boxes != nullptr : (anchors && offsets), "det_candidates: missing inputs for mode %d", mode);
    CandLevels lv{};
    lv.L = L;
    for (int l = 0; l < L; ++l) lv.row_off[l] = lvl_row_off_host[l];
    hipLaunchKernelGGL(det_candidates_kernel, dim3(cdiv(L * k, 256)), dim3(256), 0, (hipStream_t)stream, mode, topk_idx, topk_score,
                       topk_cnt, lv, k, K, anchors, (const bf16_raw*)offsets, off_ld, A, make_coder(mean4_host, std4_host), item_boxes,
                       boxes, scores, labels);
    BD_CHECK_LAUNCH("bd_det_candidates");
    return BD_OK;
}

extern "C" int bd_det_finalize(const float* boxes, const float* scores, const int32_t* labels, const int32_t* keep,
                               const int32_t* num_keep, int max_out, const float* im_info, float* out_boxes, float* out_scores,
                               int32_t* out_labels, bd_stream_t stream) {
    BD_REQUIRE(boxes && scores && labels && keep && num_keep && im_info && out_boxes && out_scores && out_labels && max_out > 0,
               "det_finalize: bad arguments");
    hipLaunchKernelGGL(det_finalize_kernel, dim3(cdiv(max_out, 256)), dim3(256), 0, (hipStream_t)stream, boxes, scores, labels, keep,
                       num_keep, max_out, im_info, out_boxes, out_scores, out_labels);
    BD_CHECK_LAUNCH("bd_det_finalize");
    return BD_OK;
}
