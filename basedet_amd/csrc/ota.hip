// OTA target assignment with the top-k ("simOTA") matcher (basedet/models/det/ota.py:76-181, layers/common/matcher.py:123-161),
// three launches for the whole batch:
//
//   ota_prep_kernel     per point: the background focal cost  S_bg = sum_k focal(x_k, 0)  (ota.py:130-135) and zeroed match counters;
//   ota_gt_kernel       one workgroup per (gt, image): dynamic k = max(1, int(sum of the 10 largest IoUs between the gt and the
//                       predicted boxes)) (matcher.py:139-140), then the k points of smallest cost (:141-143) by radix select, where
//                       cost(g, p) = focal cost of p against class(g) + 1.5 * -log(IoU) + 1e6 * [p outside the gt or its 2.5-stride
//                       centre box] (ota.py:95-151); each selected point gets its counter bumped;
//   ota_resolve_kernel  per point: no match -> background; one match -> that gt; several -> the gt of smallest cost over ALL gts
//                       (matcher.py:147-152); writes label, ltrb target, IoU target and the foreground count.
// The class cost is evaluated as (S_bg - focal(x_c, 0)) + focal(x_c, 1) instead of a sum over the K one-hot columns: G x P x K
// transcendental pairs become G x P x 2 (a re-association of the fp32 sum; assignments can differ from the literal form only
// where two costs agree to ~1e-6 relative).  Ties in the selections go to the lowest point / gt index (F.topk / F.argmin order is
// unpinned in the reference).
#include <float.h>

#include "select_dev.h"

namespace {

constexpr int OTA_MAX_K = 32;
struct OtaLevels { int start[BD_MAX_SEGS + 1]; float radius[BD_MAX_SEGS]; int L; };

// sigmoid_focal_loss value (layers/losses/sigmoid_focal_loss.py:30-35), same arithmetic as losses.hip
__device__ __forceinline__ float focal_value(float x, bool t, float alpha, float gamma) {
    const float e = __expf(-fabsf(x));
    const float l1p = __logf(1.f + e);
    const float ce = t ? -(fminf(x, 0.f) - l1p) : -(fminf(-x, 0.f) - l1p);
    const float inv = __frcp_rn(1.f + e);
    const float p = x >= 0.f ? inv : e * inv;
    const float pt = t ? 1.f - p : p;
    const float a = alpha >= 0.f ? (t ? alpha : 1.f - alpha) : 1.f;
    const float mod = gamma == 2.f ? pt * pt : (gamma == 0.f ? 1.f : powf(pt, gamma));
    return a * ce * mod;
}

// get_ltrb_boxes_iou(iou_type="iou") (layers/losses/iou_loss.py:9-44) between a predicted and a target ltrb
__device__ __forceinline__ float ltrb_iou_dev(const f32x4_t p, const f32x4_t t, float eps) {
    const float a1 = fmaxf(p[0] + p[2], 0.f) * fmaxf(p[1] + p[3], 0.f);
    const float a2 = fmaxf(t[0] + t[2], 0.f) * fmaxf(t[1] + t[3], 0.f);
    const float wi = fmaxf(fminf(p[2], t[2]) + fminf(p[0], t[0]), 0.f);
    const float hi = fmaxf(fminf(p[3], t[3]) + fminf(p[1], t[1]), 0.f);
    const float ai = wi * hi;
    return ai / fmaxf(a1 + a2 - ai, eps);
}

struct OtaIn {
    const float* points; const bf16_raw* logits; const bf16_raw* offsets; const float* sbg;
    int P, K; float alpha, gamma, reg_w;
};

__device__ __forceinline__ f32x4_t ld_pred(const bf16_raw* offsets, long long row) {
    const u32x2_t v = *reinterpret_cast<const u32x2_t*>(offsets + row * 4);
    return (f32x4_t){bf_lo(v[0]), bf_hi(v[0]), bf_lo(v[1]), bf_hi(v[1])};
}

// cost and IoU of point p (row = n*P + p, level radius rad) against one gt
__device__ __forceinline__ void ota_pair(const OtaIn& in, long long row, int p, float rad, const Box& gb, int cls, float& cost, float& iou,
                                         f32x4_t& delta) {
    const float px = in.points[2ll * p], py = in.points[2ll * p + 1];
    delta = (f32x4_t){px - gb.x1, py - gb.y1, gb.x2 - px, gb.y2 - py};                       // PointCoder.encode (boxcoder.py:132-133)
    const bool in_box = fminf(fminf(delta[0], delta[1]), fminf(delta[2], delta[3])) > 0.01f;   // ota.py:96
    const float cx = (gb.x1 + gb.x2) / 2.f, cy = (gb.y1 + gb.y2) / 2.f;
    const float c1x = fmaxf(cx - rad, gb.x1), c1y = fmaxf(cy - rad, gb.y1);
    const float c2x = fminf(cx + rad, gb.x2), c2y = fminf(cy + rad, gb.y2);
    const bool in_ctr = fminf(fminf(px - c1x, py - c1y), fminf(c2x - px, c2y - py)) > 0.f;     // ota.py:98-113
    iou = ltrb_iou_dev(ld_pred(in.offsets, row), delta, FLT_EPSILON);
    const float loss_delta = -logf(fmaxf(iou, FLT_EPSILON));                                    // iou_loss(loss_type="iou") (:97-98)
    const float xc = bf2f(in.logits[row * in.K + cls]);
    const float cls_cost = (in.sbg[row] - focal_value(xc, false, in.alpha, in.gamma)) + focal_value(xc, true, in.alpha, in.gamma);
    cost = (cls_cost + in.reg_w * loss_delta) + ((in_box && in_ctr) ? 0.f : 1e6f);            // ota.py:151
}

__global__ __launch_bounds__(256) void ota_prep_kernel(const bf16_raw* __restrict__ logits, long long rows, int K, float alpha,
                                                       float gamma, float* __restrict__ sbg, int* __restrict__ cnt,
                                                       int* __restrict__ gsel) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    float s = 0.f;
    for (int k = 0; k < K; k += 8) {
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(logits + r * K + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s += focal_value(bf_lo(v[j]), false, alpha, gamma);
            s += focal_value(bf_hi(v[j]), false, alpha, gamma);
        }
    }
    sbg[r] = s; cnt[r] = 0; gsel[r] = 0;
}

__global__ __launch_bounds__(1024) void ota_gt_kernel(OtaIn in, OtaLevels lv, const float* __restrict__ gt, const int* __restrict__ num_gt,
                                                      int Gmax, int cand_k, int* __restrict__ cnt, int* __restrict__ gsel) {
    __shared__ unsigned int hist[256];
    __shared__ int sh[8];
    __shared__ int wcnt[16];
    __shared__ float s_top[OTA_MAX_K];
    __shared__ int s_n, s_k;
    const int tid = threadIdx.x, g = blockIdx.x, n = blockIdx.y;
    if (g >= num_gt[n]) return;
    const float* gp = gt + ((long long)n * Gmax + g) * 5;
    const Box gb = ld_gt(gp);
    const int cls = (int)gp[4] - 1;
    const int P = in.P;
    const long long row0 = (long long)n * P;
    auto level_radius = [&](int p) { int l = 0; for (int k = 1; k < lv.L; ++k) if (p >= lv.start[k]) l = k; return lv.radius[l]; };
    auto pair = [&](int p, float& cost, float& iou) { f32x4_t d; ota_pair(in, row0 + p, p, level_radius(p), gb, cls, cost, iou, d); };

    // ---- dynamic k from the cand_k largest IoUs (matcher.py:139-140)
    auto key_iou = [&](int i, bool& valid) -> unsigned int { valid = true; float c, u; pair(i, c, u); return f32_asc_key(u); };
    const SelResult ri = radix_select_largest(P, cand_k, 4, key_iou, hist, sh);
    if (tid == 0) s_n = 0;
    __syncthreads();
    for (int i = tid; i < P; i += 1024) {
        bool valid; const unsigned int kv = key_iou(i, valid);
        if (ri.take_all || kv > ri.T) { const int q = atomicAdd(&s_n, 1); if (q < OTA_MAX_K) s_top[q] = f32_from_asc_key(kv); }
    }
    __syncthreads();
    if (tid == 0) {
        int m = s_n < OTA_MAX_K ? s_n : OTA_MAX_K;
        if (!ri.take_all) for (int j = 0; j < ri.need_eq && m < OTA_MAX_K; ++j) s_top[m++] = f32_from_asc_key(ri.T);
        for (int a = 1; a < m; ++a) {                       // insertion sort, descending: the order F.topk returns
            const float v = s_top[a]; int b = a - 1;
            while (b >= 0 && s_top[b] < v) { s_top[b + 1] = s_top[b]; --b; }
            s_top[b + 1] = v;
        }
        float s = 0.f;
        for (int a = 0; a < m; ++a) s += s_top[a];
        const int k = (int)s;
        s_k = k < 1 ? 1 : k;
    }
    __syncthreads();
    const int dyn_k = s_k;

    // ---- the dyn_k points of smallest cost (matcher.py:141-143)
    auto key_cost = [&](int i, bool& valid) -> unsigned int { valid = true; float c, u; pair(i, c, u); return ~f32_asc_key(c); };
    const SelResult rc = radix_select_largest(P, dyn_k, 4, key_cost, hist, sh);
    int eq_base = 0;
    for (int c0 = 0; c0 < P; c0 += 1024) {
        const int i = c0 + tid;
        bool valid = false; unsigned int kv = 0;
        if (i < P) kv = key_cost(i, valid);
        bool take = valid && (rc.take_all || kv > rc.T);
        if (!rc.take_all) {
            int tot;
            const bool eq = valid && kv == rc.T;
            const int my = eq_base + block_rank_1024(eq, wcnt, tot);
            eq_base += tot;
            take = take || (eq && my < rc.need_eq);
        }
        if (take) { atomicAdd(&cnt[row0 + i], 1); gsel[row0 + i] = g; }
    }
}

__global__ __launch_bounds__(256) void ota_resolve_kernel(OtaIn in, OtaLevels lv, const float* __restrict__ gt, const int* __restrict__ num_gt,
                                                          int Gmax, const int* __restrict__ cnt, const int* __restrict__ gsel,
                                                          int* __restrict__ labels, float* __restrict__ targets,
                                                          float* __restrict__ gt_ious, float* __restrict__ stats) {
    const int n = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    float fg = 0.f;
    if (p < in.P) {
        const long long row = (long long)n * in.P + p;
        int l = 0;
        for (int k = 1; k < lv.L; ++k) if (p >= lv.start[k]) l = k;
        const float rad = lv.radius[l];
        const int G = num_gt[n] < Gmax ? num_gt[n] : Gmax;
        const int c = cnt[row];
        int lab = 0; float iou_t = 0.f;
        f32x4_t tgt = {0.f, 0.f, 0.f, 0.f};
        if (c > 0) {
            int gsel_ = gsel[row];
            if (c > 1) {                                                     // matcher.py:147-152: argmin of the cost over all gts
                float best = INFINITY;
                for (int g = 0; g < G; ++g) {
                    const float* gp = gt + ((long long)n * Gmax + g) * 5;
                    float cost, iou; f32x4_t d;
                    ota_pair(in, row, p, rad, ld_gt(gp), (int)gp[4] - 1, cost, iou, d);
                    if (cost < best) { best = cost; gsel_ = g; }
                }
            }
            const float* gp = gt + ((long long)n * Gmax + gsel_) * 5;
            float cost;
            ota_pair(in, row, p, rad, ld_gt(gp), (int)gp[4] - 1, cost, iou_t, tgt);
            lab = (int)gp[4];
            fg = lab > 0 ? 1.f : 0.f;
            if (lab <= 0) { iou_t = 0.f; tgt = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
        }
        labels[row] = lab;
        *reinterpret_cast<f32x4_t*>(targets + row * 4) = tgt;
        gt_ious[row] = iou_t;
    }
    fg = wave_sum(fg);
    if ((threadIdx.x & 63) == 0 && fg != 0.f) { atomicAdd(&stats[0], fg); atomicAdd(&stats[1], 2.f * fg); }   // integers: exact
}

}  // namespace

extern "C" size_t bd_ota_assign_workspace_bytes(int N, int P) {
    if (N <= 0 || P <= 0) return 256;
    return (size_t)N * P * 12 + 256;
}

extern "C" int bd_ota_assign(const float* points, int P, const int32_t* lvl_start, const int32_t* strides, int L, const void* logits,
                             int K, const void* pred_ltrb, const float* gt_boxes, const int32_t* num_gt, int N, int Gmax, float alpha,
                             float gamma, float reg_weight, float center_radius, int candidate_k, int32_t* labels, float* targets,
                             float* gt_ious, float* stats, void* ws, size_t ws_bytes, bd_stream_t stream) {
    BD_REQUIRE(points && lvl_start && strides && logits && pred_ltrb && gt_boxes && num_gt && labels && targets && gt_ious && stats && ws,
               "ota_assign: null pointer");
    BD_REQUIRE(L >= 1 && L <= BD_MAX_SEGS && P > 0 && N > 0 && Gmax > 0 && K > 0 && K % 8 == 0, "ota_assign: bad sizes");
    BD_REQUIRE(candidate_k >= 1 && candidate_k <= OTA_MAX_K, "ota_assign: candidate_k=%d must be in 1..%d", candidate_k, OTA_MAX_K);
    if (ws_bytes < bd_ota_assign_workspace_bytes(N, P)) {
        bd_set_error("ota_assign: workspace %zu < %zu bytes", ws_bytes, bd_ota_assign_workspace_bytes(N, P));
        return BD_EWORKSPACE;
    }
    OtaLevels lv{};
    lv.L = L;
    for (int l = 0; l < L; ++l) { lv.start[l] = lvl_start[l]; lv.radius[l] = (float)strides[l] * center_radius; }
    lv.start[L] = lvl_start[L];
    const long long rows = (long long)N * P;
    float* sbg = (float*)ws;
    int* cnt = (int*)((unsigned char*)ws + rows * 4);
    int* gsel = (int*)((unsigned char*)ws + rows * 8);
    OtaIn in{points, (const bf16_raw*)logits, (const bf16_raw*)pred_ltrb, sbg, P, K, alpha, gamma, reg_weight};
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(stats, 0, 2 * sizeof(float), st);
    hipLaunchKernelGGL(ota_prep_kernel, dim3((unsigned)cdiv64(rows, 256)), dim3(256), 0, st, (const bf16_raw*)logits, rows, K, alpha, gamma,
                       sbg, cnt, gsel);
    hipLaunchKernelGGL(ota_gt_kernel, dim3(Gmax, N), dim3(1024), 0, st, in, lv, gt_boxes, num_gt, Gmax, candidate_k, cnt, gsel);
    hipLaunchKernelGGL(ota_resolve_kernel, dim3(cdiv(P, 256), N), dim3(256), 0, st, in, lv, gt_boxes, num_gt, Gmax, cnt, gsel, labels,
                       targets, gt_ious, stats);
    BD_CHECK_LAUNCH("bd_ota_assign");
    return BD_OK;
}
