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
                                         f32x4_t& delta, bool* inside = nullptr) {
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
    if (inside) *inside = in_box && in_ctr;
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

// ------------------------------------------------------------------------------------------------------------
// Sinkhorn matcher (layers/common/matcher.py:106-121, layers/blocks/sinkhorn_distance.py:22-50): cfg MATCHING = "sinkhorn"
// ------------------------------------------------------------------------------------------------------------
constexpr int SK_MAX_ROWS = 512;         // gts + background row held in LDS vectors

// one workgroup per (gt, image): the cost row (ota.py:151) into the workspace and the supply mu_g = max(1, int(sum of the 20 largest
// IoUs, IoU counted only inside the gt / centre box)) (matcher.py:112-113, ota.py:155)
__global__ __launch_bounds__(1024) void ota_sk_rows_kernel(OtaIn in, OtaLevels lv, const float* __restrict__ gt, const int* __restrict__ num_gt,
                                                           int Gmax, int topq, float* __restrict__ cost, float* __restrict__ mu) {
    __shared__ unsigned int hist[256];
    __shared__ int sh[8];
    __shared__ float s_top[OTA_MAX_K];
    __shared__ int s_n;
    const int tid = threadIdx.x, g = blockIdx.x, n = blockIdx.y;
    const int P = in.P;
    if (g >= num_gt[n]) return;
    const float* gp = gt + ((long long)n * Gmax + g) * 5;
    const Box gb = ld_gt(gp);
    const int cls = (int)gp[4] - 1;
    const long long row0 = (long long)n * P;
    float* crow = cost + ((long long)n * (Gmax + 1) + g) * P;
    auto level_radius = [&](int p) { int l = 0; for (int k = 1; k < lv.L; ++k) if (p >= lv.start[k]) l = k; return lv.radius[l]; };
    auto masked_iou = [&](int p, float& c) -> float {
        float u; f32x4_t d; bool inside;
        ota_pair(in, row0 + p, p, level_radius(p), gb, cls, c, u, d, &inside);
        return inside ? u : u * 0.f;
    };
    for (int p = tid; p < P; p += 1024) { float c; masked_iou(p, c); crow[p] = c; }
    auto key_iou = [&](int i, bool& valid) -> unsigned int { valid = true; float c; return f32_asc_key(masked_iou(i, c)); };
    const SelResult ri = radix_select_largest(P, topq, 4, key_iou, hist, sh);
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
        for (int a = 1; a < m; ++a) {
            const float v = s_top[a]; int b = a - 1;
            while (b >= 0 && s_top[b] < v) { s_top[b + 1] = s_top[b]; --b; }
            s_top[b + 1] = v;
        }
        float s = 0.f;
        for (int a = 0; a < m; ++a) s += s_top[a];
        const int k = (int)s;
        mu[n * (Gmax + 1) + g] = (float)(k < 1 ? 1 : k);
    }
}

// one workgroup per image: log-domain Sinkhorn iterations on the (G+1) x P cost (background row = S_bg, ota.py:154), then every point
// goes to the row of largest rescaled plan entry pi_ij / max_j pi_ij (matcher.py:118-121; ties: lowest row)
__global__ __launch_bounds__(1024) void ota_sinkhorn_kernel(OtaIn in, OtaLevels lv, const float* __restrict__ gt, const int* __restrict__ num_gt,
                                                            int Gmax, float eps, int iters, float* __restrict__ cost,
                                                            const float* __restrict__ mu_g, float* __restrict__ vbuf, int* __restrict__ labels,
                                                            float* __restrict__ targets, float* __restrict__ gt_ious,
                                                            float* __restrict__ stats) {
    __shared__ float u[SK_MAX_ROWS], lmu[SK_MAX_ROWS], rmax[SK_MAX_ROWS];
    __shared__ float s_mu_sum;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = blockIdx.x;
    const int P = in.P;
    const int G = num_gt[n] < Gmax ? num_gt[n] : Gmax;
    const long long row0 = (long long)n * P;
    float fgc = 0.f;
    if (G > 0) {
        const int G1 = G + 1;
        float* C = cost + (long long)n * (Gmax + 1) * P;
        float* v = vbuf + row0;
        for (int p = tid; p < P; p += 1024) { C[(long long)G * P + p] = in.sbg[row0 + p]; v[p] = 1.f; }      // background row, v = 1
        if (tid == 0) {
            float s = 0.f;
            for (int i = 0; i < G; ++i) s += mu_g[n * (Gmax + 1) + i];
            s_mu_sum = s;
        }
        __syncthreads();
        if (tid < G1) {
            const float m = tid < G ? mu_g[n * (Gmax + 1) + tid] : (float)P - s_mu_sum;      // matcher.py:114
            lmu[tid] = logf(m + 1e-8f);
            u[tid] = 1.f;
        }
        __syncthreads();
        const float lnu = logf(1.f + 1e-8f);
        const float inv = 1.f / eps;
        for (int it = 0; it < iters; ++it) {
            // v_j += eps (log nu_j - logsumexp_i M_ij),  M_ij = (-c_ij + u_i + v_j) / eps   (sinkhorn_distance.py:28-30)
            for (int p = tid; p < P; p += 1024) {
                const float vj = v[p];
                float m = -INFINITY;
                for (int i = 0; i < G1; ++i) m = fmaxf(m, (-C[(long long)i * P + p] + u[i] + vj) * inv);
                float s = 0.f;
                for (int i = 0; i < G1; ++i) s += expf((-C[(long long)i * P + p] + u[i] + vj) * inv - m);
                v[p] = vj + eps * (lnu - (m + logf(s)));
            }
            __syncthreads();
            // u_i += eps (log mu_i - logsumexp_j M_ij) with the new v (:31-33); rows spread over the 16 waves
            for (int i = wave; i < G1; i += 16) {
                const float ui = u[i];
                float m = -INFINITY, s = 0.f;
                for (int p = lane; p < P; p += 64) {
                    const float x = (-C[(long long)i * P + p] + ui + v[p]) * inv;
                    if (x > m) { s = s * expf(m - x) + 1.f; m = x; } else s += expf(x - m);
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
                    const float mm = fmaxf(m, m2);
                    s = (m == -INFINITY ? 0.f : s * expf(m - mm)) + (m2 == -INFINITY ? 0.f : s2 * expf(m2 - mm));
                    m = mm;
                }
                if (lane == 0) u[i] = ui + eps * (lmu[i] - (m + logf(s)));
            }
            __syncthreads();
        }
        // pi = exp(M); rescale each row by its maximum
        for (int i = wave; i < G1; i += 16) {
            float m = -INFINITY;
            for (int p = lane; p < P; p += 64) m = fmaxf(m, expf((-C[(long long)i * P + p] + u[i] + v[p]) * inv));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            if (lane == 0) rmax[i] = m;
        }
        __syncthreads();
        for (int p = tid; p < P; p += 1024) {
            float best = -INFINITY; int bi = 0;
            for (int i = 0; i < G1; ++i) {
                const float r = expf((-C[(long long)i * P + p] + u[i] + v[p]) * inv) / rmax[i];
                if (r > best) { best = r; bi = i; }
            }
            int lab = 0; float iou_t = 0.f;
            f32x4_t tgt = {0.f, 0.f, 0.f, 0.f};
            if (bi != G) {
                int l = 0;
                for (int k = 1; k < lv.L; ++k) if (p >= lv.start[k]) l = k;
                const float* gp = gt + ((long long)n * Gmax + bi) * 5;
                float c; bool inside;
                ota_pair(in, row0 + p, p, lv.radius[l], ld_gt(gp), (int)gp[4] - 1, c, iou_t, tgt, &inside);
                if (!inside) iou_t = iou_t * 0.f;                              // ious * is_in_boxes (ota.py:155)
                lab = (int)gp[4];
                if (lab > 0) fgc += 1.f; else { iou_t = 0.f; tgt = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
            }
            labels[row0 + p] = lab;
            *reinterpret_cast<f32x4_t*>(targets + (row0 + p) * 4) = tgt;
            gt_ious[row0 + p] = iou_t;
        }
    } else {
        for (int p = tid; p < P; p += 1024) {
            labels[row0 + p] = 0; gt_ious[row0 + p] = 0.f;
            *reinterpret_cast<f32x4_t*>(targets + (row0 + p) * 4) = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
    }
    fgc = wave_sum(fgc);
    if (lane == 0 && fgc != 0.f) { atomicAdd(&stats[0], fgc); atomicAdd(&stats[1], 2.f * fgc); }
}

}  // namespace

extern "C" size_t bd_ota_sinkhorn_workspace_bytes(int N, int P, int Gmax) {
    if (N <= 0 || P <= 0 || Gmax <= 0) return 256;
    return (size_t)N * P * 8 + (size_t)N * (Gmax + 1) * P * 4 + (size_t)N * (Gmax + 1) * 4 + 1024;
}

extern "C" int bd_ota_assign_sinkhorn(const float* points, int P, const int32_t* lvl_start, const int32_t* strides, int L,
                                      const void* logits, int K, const void* pred_ltrb, const float* gt_boxes, const int32_t* num_gt, int N,
                                      int Gmax, float alpha, float gamma, float reg_weight, float center_radius, int topq, float eps,
                                      int iters, int32_t* labels, float* targets, float* gt_ious, float* stats, void* ws, size_t ws_bytes,
                                      bd_stream_t stream) {
    BD_REQUIRE(points && lvl_start && strides && logits && pred_ltrb && gt_boxes && num_gt && labels && targets && gt_ious && stats && ws,
               "ota_assign_sinkhorn: null pointer");
    BD_REQUIRE(L >= 1 && L <= BD_MAX_SEGS && P > 0 && N > 0 && Gmax > 0 && K > 0 && K % 8 == 0, "ota_assign_sinkhorn: bad sizes");
    BD_REQUIRE(Gmax + 1 <= SK_MAX_ROWS, "ota_assign_sinkhorn: %d gt slots exceed %d", Gmax, SK_MAX_ROWS - 1);
    BD_REQUIRE(topq >= 1 && topq <= OTA_MAX_K && eps > 0.f && iters >= 1, "ota_assign_sinkhorn: bad matcher parameters");
    if (ws_bytes < bd_ota_sinkhorn_workspace_bytes(N, P, Gmax)) {
        bd_set_error("ota_assign_sinkhorn: workspace %zu < %zu bytes", ws_bytes, bd_ota_sinkhorn_workspace_bytes(N, P, Gmax));
        return BD_EWORKSPACE;
    }
    OtaLevels lv{};
    lv.L = L;
    for (int l = 0; l < L; ++l) { lv.start[l] = lvl_start[l]; lv.radius[l] = (float)strides[l] * center_radius; }
    lv.start[L] = lvl_start[L];
    const long long rows = (long long)N * P;
    unsigned char* wb = (unsigned char*)ws;
    float* sbg = (float*)wb;
    float* vbuf = (float*)(wb + rows * 4);
    float* cost = (float*)(wb + rows * 8);
    float* mu = (float*)(wb + rows * 8 + (size_t)N * (Gmax + 1) * P * 4);
    OtaIn in{points, (const bf16_raw*)logits, (const bf16_raw*)pred_ltrb, sbg, P, K, alpha, gamma, reg_weight};
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(stats, 0, 2 * sizeof(float), st);
    // the prep kernel also zeroes two int vectors of `rows` entries: point them at the v buffer and the first cost row
    hipLaunchKernelGGL(ota_prep_kernel, dim3((unsigned)cdiv64(rows, 256)), dim3(256), 0, st, (const bf16_raw*)logits, rows, K, alpha, gamma,
                       sbg, (int*)vbuf, (int*)vbuf);
    hipLaunchKernelGGL(ota_sk_rows_kernel, dim3(Gmax, N), dim3(1024), 0, st, in, lv, gt_boxes, num_gt, Gmax, topq, cost, mu);
    hipLaunchKernelGGL(ota_sinkhorn_kernel, dim3(N), dim3(1024), 0, st, in, lv, gt_boxes, num_gt, Gmax, eps, iters, cost, mu, vbuf, labels,
                       targets, gt_ious, stats);
    BD_CHECK_LAUNCH("bd_ota_assign_sinkhorn");
    return BD_OK;
}

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
