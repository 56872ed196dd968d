// The reference's operator surface (basedet.layers / basedet.structures) as stand-alone fp32 kernels: the elementwise forms of the
// losses with the reference's signatures (value per element, gradient w.r.t. the prediction given an upstream gradient), Matcher on
// a materialised (G, A) matrix, the max-mode RoI pooling and the FPN level rule.  The training step itself never calls these -- it
// uses the fused, label-driven kernels of losses.hip / boxops.hip / rcnn_ops.hip -- they serve callers written against
// basedet.layers.* (layers/losses/*.py, layers/common/matcher.py:31-51, layers/common/roi_pool.py:12-78).
#include <math.h>

#include "common.h"

namespace {

__device__ __forceinline__ float log_sigmoid(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// layers/losses/sigmoid_focal_loss.py:9-36.  alpha < 0: no class weighting; gamma == 0: no modulation.
__global__ __launch_bounds__(256) void focal_elem_kernel(const float* __restrict__ x_, const float* __restrict__ t_, long long n,
                                                        float alpha, float gamma, const float* __restrict__ gout,
                                                        float* __restrict__ loss, float* __restrict__ dx) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float x = x_[i], t = t_[i];
        const float p = sigmoidf(x);
        const float ce = -(t * log_sigmoid(x) + (1.f - t) * log_sigmoid(-x));
        const float m = t * (1.f - p) + (1.f - t) * p;
        const float mod = gamma != 0.f ? powf(m, gamma) : 1.f;
        const float at = alpha >= 0.f ? t * alpha + (1.f - t) * (1.f - alpha) : 1.f;
        if (loss) loss[i] = ce * mod * at;
        if (dx) {
            float d = mod * (p - t);
            if (gamma != 0.f) d += gamma * powf(m, gamma - 1.f) * (1.f - 2.f * t) * p * (1.f - p) * ce;
            dx[i] = d * at * (gout ? gout[i] : 1.f);
        }
    }
}

// layers/losses/cross_entropy.py:7-29
__global__ __launch_bounds__(256) void bce_elem_kernel(const float* __restrict__ x_, const float* __restrict__ t_, long long n,
                                                      int with_logits, const float* __restrict__ gout, float* __restrict__ loss,
                                                      float* __restrict__ dx) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float x = x_[i], t = t_[i];
        float l, d;
        if (with_logits) {
            l = -(t * log_sigmoid(x) + (1.f - t) * log_sigmoid(-x));
            d = sigmoidf(x) - t;
        } else {
            l = -(t * logf(x) + (1.f - t) * logf(1.f - x));
            d = -t / x + (1.f - t) / (1.f - x);
        }
        if (loss) loss[i] = l;
        if (dx) dx[i] = d * (gout ? gout[i] : 1.f);
    }
}

// layers/losses/smooth_l1_loss.py:7-34
__global__ __launch_bounds__(256) void smooth_l1_elem_kernel(const float* __restrict__ p_, const float* __restrict__ t_, long long n,
                                                            float beta, const float* __restrict__ gout, float* __restrict__ loss,
                                                            float* __restrict__ dp) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float x = p_[i] - t_[i], a = fabsf(x);
        float l, d;
        if (beta < 1e-5f) {
            l = a; d = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
        } else if (a < beta) {
            l = 0.5f * x * x / beta; d = x / beta;
        } else {
            l = a - 0.5f * beta; d = x > 0.f ? 1.f : -1.f;
        }
        if (loss) loss[i] = l;
        if (dp) dp[i] = d * (gout ? gout[i] : 1.f);
    }
}

// loss = f(iou) (layers/losses/iou_loss.py:95-100): type 0 "iou" -log(clip(iou, eps)), 1 "linear_iou" 1 - iou, 2 "giou" 1 - giou,
// 3 "square_iou" 1 - iou^2;  returns dloss/d(iou or giou)
__device__ __forceinline__ float iou_to_loss(float v, int type, float eps, float* dv) {
    if (type == 0) {
        *dv = v > eps ? -1.f / v : 0.f;
        return -logf(fmaxf(v, eps));
    }
    if (type == 3) {
        *dv = -2.f * v;
        return 1.f - v * v;
    }
    *dv = -1.f;
    return 1.f - v;
}

// get_ltrb_boxes_iou (iou_loss.py:9-56) row-wise + the loss map; gradient w.r.t. the four predicted distances
__global__ __launch_bounds__(256) void iou_loss_ltrb_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                           long long n, int type, float eps, const float* __restrict__ gout,
                                                           float* __restrict__ loss, float* __restrict__ ious,
                                                           float* __restrict__ dpred) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const f32x4_t p = *reinterpret_cast<const f32x4_t*>(pred + i * 4);
        const f32x4_t t = *reinterpret_cast<const f32x4_t*>(target + i * 4);
        const float pw = p[0] + p[2], ph = p[1] + p[3];
        const float pwc = fmaxf(pw, 0.f), phc = fmaxf(ph, 0.f);
        const float a1 = pwc * phc;
        const float a2 = fmaxf(t[0] + t[2], 0.f) * fmaxf(t[1] + t[3], 0.f);
        const float wi_raw = fminf(p[2], t[2]) + fminf(p[0], t[0]);
        const float hi_raw = fminf(p[3], t[3]) + fminf(p[1], t[1]);
        const float wi = fmaxf(wi_raw, 0.f), hi = fmaxf(hi_raw, 0.f);
        const float ai = wi * hi;
        const float au = a1 + a2 - ai;
        const float auc = fmaxf(au, eps);
        const float iou = ai / auc;
        const float gw = fmaxf(p[2], t[2]) + fmaxf(p[0], t[0]);
        const float gh = fmaxf(p[3], t[3]) + fmaxf(p[1], t[1]);
        const float ac = gw * gh;
        const float acl = fmaxf(ac, eps);
        const float v = type == 2 ? iou - (ac - au) / acl : iou;
        float dv;
        const float l = iou_to_loss(v, type, eps, &dv);
        if (loss) loss[i] = l;
        if (ious) ious[i] = v;
        if (dpred) {
            const float go = gout ? gout[i] : 1.f;
            f32x4_t g;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool isx = (k & 1) == 0;
                const float da1 = isx ? (pw > 0.f ? phc : 0.f) : (ph > 0.f ? pwc : 0.f);
                const float dmin = p[k] < t[k] ? 1.f : 0.f;
                const float dai = isx ? (wi_raw > 0.f ? dmin * hi : 0.f) : (hi_raw > 0.f ? dmin * wi : 0.f);
                const float dau = da1 - dai;
                const float dauc = au > eps ? dau : 0.f;
                float d = (dai * auc - ai * dauc) / (auc * auc);
                if (type == 2) {
                    const float dmax = p[k] > t[k] ? 1.f : 0.f;
                    const float dac = isx ? dmax * gh : dmax * gw;
                    const float dacl = ac > eps ? dac : 0.f;
                    d -= ((dac - dau) * acl - (ac - au) * dacl) / (acl * acl);
                }
                g[k] = d * dv * go;
            }
            *reinterpret_cast<f32x4_t*>(dpred + i * 4) = g;
        }
    }
}

// the xyxy branch of iou_loss (iou_loss.py:83-91) works on the PAIRWISE (N, M) matrix of Boxes.iou / Boxes.giou: this maps it
__global__ __launch_bounds__(256) void iou_map_kernel(const float* __restrict__ v, long long n, int type, float eps,
                                                     float* __restrict__ loss) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float dv;
        loss[i] = iou_to_loss(v[i], type, eps, &dv);
    }
}

// ---- Matcher.__call__(matrix) (layers/common/matcher.py:31-51) --------------------------------------------------------
__global__ __launch_bounds__(256) void rowmax_kernel(const float* __restrict__ m, int G, long long A, float* __restrict__ rmax) {
    __shared__ float red[4];
    const int g = blockIdx.x;
    float v = -INFINITY;
    for (long long a = threadIdx.x; a < A; a += 256) v = fmaxf(v, m[(long long)g * A + a]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) rmax[g] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

struct MatchCfg {
    float thr[8];     // interior thresholds, ascending (the reference pads them with -inf / +inf)
    int lab[9];
    int nthr;
};

__global__ __launch_bounds__(256) void matcher_kernel(const float* __restrict__ m, int G, long long A, MatchCfg c, int allow_lq,
                                                     const float* __restrict__ rmax, int* __restrict__ idx, int* __restrict__ labels) {
    const long long a = (long long)blockIdx.x * 256 + threadIdx.x;
    if (a >= A) return;
    float best = -INFINITY;
    int bi = 0;
    bool lq = false;
    for (int g = 0; g < G; ++g) {
        const float v = m[(long long)g * A + a];
        if (v > best) { best = v; bi = g; }          // first maximum wins (F.argmax tie rule documented in oracle/box_ops.py)
        lq |= (v == rmax[g]);
    }
    int lab = -1;
    // bands [low, high): -inf, thr..., +inf
    for (int k = 0; k <= c.nthr; ++k) {
        const float lo = k == 0 ? -INFINITY : c.thr[k - 1];
        const float hi = k == c.nthr ? INFINITY : c.thr[k];
        if (best >= lo && best < hi) lab = c.lab[k];
    }
    if (allow_lq && lq) lab = 1;
    idx[a] = bi;
    labels[a] = lab;
}

// ---- assign_rois (roi_pool.py:12-25): level = clamp(floor(4 + log2(sqrt(area) / 224)), min, max) - min ----------------
__global__ __launch_bounds__(256) void roi_levels_kernel(const float* __restrict__ rois, int ld, int R, int min_level, int max_level,
                                                        int* __restrict__ out) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const float* b = rois + (long long)r * ld + (ld == 5 ? 1 : 0);
    const float area = (b[2] - b[0]) * (b[3] - b[1]);
    const float v = 4.f + logf(sqrtf(area) / 224.f) / 0.6931471805599453f;
    int lvl = min_level;
    if (v == INFINITY) lvl = max_level;
    else if (v == v && v != -INFINITY) lvl = (int)floorf(v);
    lvl = min(max(lvl, min_level), max_level) - min_level;
    out[r] = lvl;
}

// ---- F.nn.roi_pooling(mode="max") (roi_pool.py:65): the Caffe ROIPooling rule --------------------------------------------
// rois [R][5] = (batch index, x1, y1, x2, y2); feat fp32 NCHW [N][C][H][W]; out [R][C][PH][PW]
__global__ __launch_bounds__(256) void roi_pool_max_kernel(const float* __restrict__ feat, int N, int C, int H, int W,
                                                          const float* __restrict__ rois, int R, float scale, int PH, int PW,
                                                          float* __restrict__ out) {
    const long long total = (long long)R * C * PH * PW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int pw = (int)(i % PW), ph = (int)((i / PW) % PH), c = (int)((i / ((long long)PW * PH)) % C);
        const int r = (int)(i / ((long long)PW * PH * C));
        const float* b = rois + (long long)r * 5;
        const int n = (int)b[0];
        const int x1 = (int)roundf(b[1] * scale), y1 = (int)roundf(b[2] * scale);
        const int x2 = (int)roundf(b[3] * scale), y2 = (int)roundf(b[4] * scale);
        const int rw = max(x2 - x1 + 1, 1), rh = max(y2 - y1 + 1, 1);
        const float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
        int hs = (int)floorf(ph * bh) + y1, he = (int)ceilf((ph + 1) * bh) + y1;
        int ws = (int)floorf(pw * bw) + x1, we = (int)ceilf((pw + 1) * bw) + x1;
        hs = min(max(hs, 0), H); he = min(max(he, 0), H);
        ws = min(max(ws, 0), W); we = min(max(we, 0), W);
        const bool empty = he <= hs || we <= ws || n < 0 || n >= N;
        float v = empty ? 0.f : -INFINITY;
        if (!empty) {
            const float* f = feat + ((long long)n * C + c) * H * W;
            for (int y = hs; y < he; ++y)
                for (int x = ws; x < we; ++x) v = fmaxf(v, f[(long long)y * W + x]);
        }
        out[i] = v;
    }
}

inline int egrid(long long n) { return (int)std::min<long long>(cdiv64(std::max<long long>(n, 1), 256), 256 * 32); }

}  // namespace

extern "C" {

int bd_sigmoid_focal_loss_elem(const float* logits, const float* targets, int64_t n, float alpha, float gamma, const float* gout,
                               float* loss, float* dlogits, bd_stream_t stream) {
    BD_REQUIRE(logits && targets && (loss || dlogits), "sigmoid_focal_loss_elem: null pointer");
    if (n == 0) return BD_OK;
    hipLaunchKernelGGL(focal_elem_kernel, dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream, logits, targets, (long long)n, alpha,
                       gamma, gout, loss, dlogits);
    BD_CHECK_LAUNCH("bd_sigmoid_focal_loss_elem");
    return BD_OK;
}

int bd_bce_elem(const float* pred, const float* label, int64_t n, int with_logits, const float* gout, float* loss, float* dpred,
                bd_stream_t stream) {
    BD_REQUIRE(pred && label && (loss || dpred), "bce_elem: null pointer");
    if (n == 0) return BD_OK;
    hipLaunchKernelGGL(bce_elem_kernel, dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream, pred, label, (long long)n, with_logits,
                       gout, loss, dpred);
    BD_CHECK_LAUNCH("bd_bce_elem");
    return BD_OK;
}

int bd_smooth_l1_elem(const float* pred, const float* target, int64_t n, float beta, const float* gout, float* loss, float* dpred,
                      bd_stream_t stream) {
    BD_REQUIRE(pred && target && (loss || dpred), "smooth_l1_elem: null pointer");
    if (n == 0) return BD_OK;
    hipLaunchKernelGGL(smooth_l1_elem_kernel, dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream, pred, target, (long long)n, beta,
                       gout, loss, dpred);
    BD_CHECK_LAUNCH("bd_smooth_l1_elem");
    return BD_OK;
}

int bd_iou_loss_ltrb(const float* pred, const float* target, int64_t n, int loss_type, float eps, const float* gout, float* loss,
                     float* ious, float* dpred, bd_stream_t stream) {
    BD_REQUIRE(pred && target && (loss || ious || dpred), "iou_loss_ltrb: null pointer");
    BD_REQUIRE(loss_type >= 0 && loss_type <= 3, "iou_loss_ltrb: loss_type %d", loss_type);
    if (n == 0) return BD_OK;
    hipLaunchKernelGGL(iou_loss_ltrb_kernel, dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream, pred, target, (long long)n, loss_type,
                       eps, gout, loss, ious, dpred);
    BD_CHECK_LAUNCH("bd_iou_loss_ltrb");
    return BD_OK;
}

int bd_iou_to_loss(const float* ious, int64_t n, int loss_type, float eps, float* loss, bd_stream_t stream) {
    BD_REQUIRE(ious && loss, "iou_to_loss: null pointer");
    BD_REQUIRE(loss_type >= 0 && loss_type <= 3, "iou_to_loss: loss_type %d", loss_type);
    if (n == 0) return BD_OK;
    hipLaunchKernelGGL(iou_map_kernel, dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream, ious, (long long)n, loss_type, eps, loss);
    BD_CHECK_LAUNCH("bd_iou_to_loss");
    return BD_OK;
}

int bd_matcher_matrix(const float* matrix, int G, int64_t A, const float* thresholds_host, const int32_t* labels_host, int n_thresholds,
                      int allow_low_quality, int32_t* match_idx, int32_t* labels, float* ws_rowmax, bd_stream_t stream) {
    BD_REQUIRE(matrix && match_idx && labels && ws_rowmax && thresholds_host && labels_host, "matcher_matrix: null pointer");
    BD_REQUIRE(G >= 1 && A >= 0, "matcher_matrix: G=%d (an empty matrix has no argmax; the reference raises too)", G);
    BD_REQUIRE(n_thresholds >= 0 && n_thresholds <= 8, "matcher_matrix: at most 8 thresholds");
    if (A == 0) return BD_OK;
    MatchCfg c{};
    c.nthr = n_thresholds;
    for (int i = 0; i < n_thresholds; ++i) c.thr[i] = thresholds_host[i];
    for (int i = 0; i <= n_thresholds; ++i) c.lab[i] = labels_host[i];
    hipLaunchKernelGGL(rowmax_kernel, dim3(G), dim3(256), 0, (hipStream_t)stream, matrix, G, (long long)A, ws_rowmax);
    hipLaunchKernelGGL(matcher_kernel, dim3((unsigned)cdiv64(A, 256)), dim3(256), 0, (hipStream_t)stream, matrix, G, (long long)A, c,
                       allow_low_quality, ws_rowmax, match_idx, labels);
    BD_CHECK_LAUNCH("bd_matcher_matrix");
    return BD_OK;
}

int bd_assign_roi_levels(const float* rois, int ld, int R, int min_level, int max_level, int32_t* levels, bd_stream_t stream) {
    BD_REQUIRE(rois && levels && (ld == 4 || ld == 5), "assign_roi_levels: rois must be [R][4] or [R][5]");
    if (R == 0) return BD_OK;
    hipLaunchKernelGGL(roi_levels_kernel, dim3(cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, rois, ld, R, min_level, max_level, levels);
    BD_CHECK_LAUNCH("bd_assign_roi_levels");
    return BD_OK;
}

int bd_roi_pool_max_fwd(const float* feat_nchw, int N, int C, int H, int W, const float* rois5, int R, float scale, int PH, int PW,
                        float* out, bd_stream_t stream) {
    BD_REQUIRE(feat_nchw && rois5 && out, "roi_pool_max: null pointer");
    BD_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && PH > 0 && PW > 0, "roi_pool_max: bad sizes");
    if (R == 0) return BD_OK;
    hipLaunchKernelGGL(roi_pool_max_kernel, dim3(egrid((long long)R * C * PH * PW)), dim3(256), 0, (hipStream_t)stream, feat_nchw, N, C,
                       H, W, rois5, R, scale, PH, PW, out);
    BD_CHECK_LAUNCH("bd_roi_pool_max_fwd");
    return BD_OK;
}

}  // extern "C"
