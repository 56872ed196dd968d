// FreeAnchor bag losses (basedet/models/det/free_anchor.py:20-142) with their gradients, three launches per step:
//
//   fa_gt_kernel     one workgroup per (gt, image): (a) max over all anchors of IoU(gt, decoded prediction) -> the upper clip
//                    threshold of the box probability (:58-63); (b) the BUCKET_SIZE anchors of largest IoU(gt, anchor) by radix
//                    select (:84-89; ties at the boundary: lowest anchor index first -- the reference's F.topk order is unpinned);
//                    (c) the positive bag: p_j = sigmoid(logit[a_j, class]) * exp(-w * smooth_l1(offsets[a_j] - encode(a_j, gt)))
//                    (:91-121), loss -log(sum_j w_j p_j / sum_j w_j), w_j = 1 / (1 - p_j) (:32-36), and dL/dlogit, dL/doffsets of
//                    its members into the workspace;
//   fa_neg_kernel    one workgroup per (64 anchors, image): box probability of every (anchor, class) -- the clipped, rescaled IoU
//                    of the decoded prediction with each gt, written at the gt's class, the LAST gt winning where two share
//                    anchor and class (:72-73, an indexed assignment; order unpinned) -- kept in LDS as (gt index, value) under
//                    a 64-bit atomicMax, then the negative loss q^gamma * -log(1 - q), q = sigmoid(x)(1 - box_prob) (:38-39,
//                    :128-130) and its gradient for all logits;
//   fa_apply_kernel  one workgroup per image: adds the bag gradients to d_logits / d_offsets gt after gt (fixed order:
//                    two bags may share an anchor), and block 0 reduces the losses in a fixed order.
// safelog = log(max(x, FLT_MIN)) (layers/common/function.py:35-44).  Compiled with -ffp-contract=off (IoU / encode / decode as in
// the oracle).
#pragma clang fp contract(off)
#include <float.h>

#include "select_dev.h"

namespace {

constexpr int FA_MAX_BUCKET = 64;
constexpr int FA_NEG_ANCH = 64;          // anchors per negative-loss workgroup

struct FaWs { size_t thresh2, bag_idx, bag_grad, pos_loss, neg_part, maxiou, total; };
inline FaWs fa_layout(int N, int Gmax, int bucket, int A) {
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    FaWs w{};
    size_t o = 0;
    w.thresh2 = o; o = up(o + sizeof(float) * N * Gmax);
    w.bag_idx = o; o = up(o + sizeof(int) * (size_t)N * Gmax * bucket);
    w.bag_grad = o; o = up(o + sizeof(float) * (size_t)N * Gmax * bucket * 5);
    w.pos_loss = o; o = up(o + sizeof(float) * N * Gmax);
    w.neg_part = o; o = up(o + sizeof(float) * (size_t)N * cdiv(A, FA_NEG_ANCH));
    w.maxiou = o; o = up(o + sizeof(unsigned int) * N * Gmax);
    w.total = o;
    return w;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float safelogf_(float x) { return logf(fmaxf(x, FLT_MIN)); }

__device__ __forceinline__ f32x4_t ld_offsets(const bf16_raw* offsets, long long n_pix0, int a, int apix, int ld) {
    const bf16_raw* p = offsets + (n_pix0 + a / apix) * ld + (a % apix) * 4;
    const u32x2_t v = *reinterpret_cast<const u32x2_t*>(p);
    return (f32x4_t){bf_lo(v[0]), bf_hi(v[0]), bf_lo(v[1]), bf_hi(v[1])};
}

__device__ __forceinline__ int total_fg(const int* num_gt, int N) {
    int s = 0;
    for (int i = 0; i < N; ++i) s += num_gt[i];
    return s;
}

// (a) of fa_gt_kernel's job, round 5: max over all anchors of IoU(gt, decoded prediction) for EVERY gt of the image in one sweep -- a
// prediction is decoded once (two exponentials) and met with all gts, instead of once per gt in that gt's workgroup; the maximum of
// non-negative floats is the maximum of their bit patterns, whatever the order: atomicMax, same bits as the per-gt loop
__global__ __launch_bounds__(256) void fa_maxiou_kernel(const bf16_raw* __restrict__ offsets, int box_ld, int apix, const float* __restrict__ anchors,
                                                        int A, const float* __restrict__ gt, const int* __restrict__ num_gt, int Gmax,
                                                        Coder coder, unsigned int* __restrict__ maxiou) {
    __shared__ f32x4_t s_gt[256];
    const int n = blockIdx.y, a = blockIdx.x * 256 + threadIdx.x;
    const int G = num_gt[n];
    if (G <= 0) return;
    Box p{0.f, 0.f, 0.f, 0.f};
    float parea = 0.f;
    if (a < A) {
        const Box ab = ld_box(anchors + 4ll * a);
        const f32x4_t pb = decode_dev(ab, ld_offsets(offsets, (long long)n * (A / apix), a, apix, box_ld), coder);
        p = Box{pb[0], pb[1], pb[2], pb[3]};
        parea = box_area(p);
    }
    for (int g0 = 0; g0 < G; g0 += 256) {          // the image's gts through LDS, 256 at a time (a scalar load per gt and wave cost 2 us each)
    __syncthreads();
    if (g0 + (int)threadIdx.x < G) {
        const float* gp = gt + ((long long)n * Gmax + g0 + threadIdx.x) * 5;
        s_gt[threadIdx.x] = (f32x4_t){gp[0], gp[1], gp[2], gp[3]};
    }
    __syncthreads();
    const int g1 = G - g0 < 256 ? G : g0 + 256;
    for (int g = g0; g < g1; ++g) {
        const f32x4_t gv = s_gt[g - g0];
        const Box gb{gv[0], gv[1], gv[2], gv[3]};
        // 64 consecutive anchors sit on ~7 neighbouring pixels: for most gts no prediction of the wave has a positive-width intersection
        // (IoU exactly 0, box_inter) -- decided on four min / max and a ballot, before areas, division, reduction and atomic
        const bool hit = a < A && fminf(gb.x2, p.x2) > fmaxf(gb.x1, p.x1) && fminf(gb.y2, p.y2) > fmaxf(gb.y1, p.y1);
        if (__ballot(hit) == 0ull) continue;
        float u = hit ? box_iou_dev(gb, box_area(gb), p, parea) : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) u = fmaxf(u, __shfl_xor(u, o, 64));
        if ((threadIdx.x & 63) == 0 && u > 0.f) atomicMax(&maxiou[n * Gmax + g], __float_as_uint(u));
    }
    }
}

__global__ __launch_bounds__(1024) void fa_gt_kernel(const bf16_raw* __restrict__ logits, const bf16_raw* __restrict__ offsets,
                                                     int box_ld, int apix, const float* __restrict__ anchors, int A, int K,
                                                     const float* __restrict__ gt, const int* __restrict__ num_gt, int N, int Gmax,
                                                     Coder coder, float iou_thresh, int bucket, float beta, float reg_weight,
                                                     float alpha, const unsigned int* __restrict__ maxiou, float* __restrict__ thresh2,
                                                     int* __restrict__ bag_idx, float* __restrict__ bag_grad, float* __restrict__ pos_loss) {
    __shared__ unsigned int hist[256];
    __shared__ int sh[8];
    __shared__ int wcnt[16];
    __shared__ float red[16];
    __shared__ int s_idx[FA_MAX_BUCKET], s_eq[FA_MAX_BUCKET];
    __shared__ float s_p[FA_MAX_BUCKET], s_w[FA_MAX_BUCKET];
    __shared__ float s_bag[2];
    const int tid = threadIdx.x, g = blockIdx.x, n = blockIdx.y;
    const int slot = n * Gmax + g;
    if (g >= num_gt[n]) {
        if (tid == 0) { pos_loss[slot] = 0.f; thresh2[slot] = 1.f; }
        return;
    }
    const float* gp = gt + (long long)slot * 5;
    const Box gb = ld_gt(gp);
    const float garea = box_area(gb);
    const int cls = (int)gp[4] - 1;
    const long long pix0 = (long long)n * (A / apix);

    // (a) max IoU of the gt with the decoded predictions: fa_maxiou_kernel; here the clip (:58-63)
    if (tid == 0) thresh2[slot] = fminf(fmaxf(__uint_as_float(maxiou[slot]), iou_thresh + 1e-7f), 1.f);

    // (b) the bag: `bucket` largest IoU(gt, anchor), ascending anchor index.  Nearly every anchor has IoU exactly 0 with a given gt:
    // counted into the radix histograms they would all hit one LDS bin (same-address atomics serialise), so the selection first runs
    // over the positive IoUs only and falls back to all anchors in the rare case that fewer than `bucket` overlap the gt at all.
    bool pos_only = true;
    auto key = [&](int i, bool& valid) -> unsigned int {
        const Box ab = ld_box(anchors + 4ll * i);
        // round 5: an anchor without a positive-width intersection has IoU exactly 0 (box_inter) and is not a candidate of the first pass --
        // decided on four min / max before the areas and the division; anchors come in (level, y, x) order, so whole waves leave here
        if (pos_only && !(fminf(gb.x2, ab.x2) > fmaxf(gb.x1, ab.x1) && fminf(gb.y2, ab.y2) > fmaxf(gb.y1, ab.y1))) { valid = false; return 0u; }
        const float u = box_iou_dev(gb, garea, ab, box_area(ab));
        valid = !pos_only || u > 0.f;
        return f32_asc_key(u);
    };
    SelResult r = radix_select_largest(A, bucket, 4, key, hist, sh);
    if (r.take_all && r.n_valid < bucket && r.n_valid < A) {          // workgroup-uniform
        pos_only = false;
        r = radix_select_largest(A, bucket, 4, key, hist, sh);
    }
    // members above the threshold: unordered collection; members equal to it: the first need_eq in index order.  Round 5: the same sweep
    // also lists the anchors AT the threshold (IoUs are floats: almost always exactly need_eq = 1 of them) -- with at most 64 of them
    // the ordered walk over all anchors below (two barriers per 1024 anchors, half of A on average) is not needed
    if (tid == 0) { sh[4] = 0; sh[5] = 0; }
    __syncthreads();
    for (int i = tid; i < A; i += 1024) {
        bool valid; const unsigned int kv = key(i, valid);
        if (valid && (r.take_all || kv > r.T)) { const int q = atomicAdd(&sh[4], 1); if (q < FA_MAX_BUCKET) s_idx[q] = i; }
        else if (valid && kv == r.T) { const int q = atomicAdd(&sh[5], 1); if (q < FA_MAX_BUCKET) s_eq[q] = i; }
    }
    __syncthreads();
    int base = sh[4];
    const int n_eq = sh[5];
    if (!r.take_all && r.need_eq > 0 && n_eq <= FA_MAX_BUCKET) {
        if (tid == 0) {
            for (int x = 1; x < n_eq; ++x) {                         // ascending anchor index
                const int v = s_eq[x]; int y = x - 1;
                while (y >= 0 && s_eq[y] > v) { s_eq[y + 1] = s_eq[y]; --y; }
                s_eq[y + 1] = v;
            }
            for (int x = 0; x < r.need_eq && x < n_eq; ++x) if (base + x < FA_MAX_BUCKET) s_idx[base + x] = s_eq[x];
        }
        base += r.need_eq;
    } else if (!r.take_all && r.need_eq > 0) {
        int eq_base = 0;
        for (int c0 = 0; c0 < A && eq_base < r.need_eq; c0 += 1024) {       // eq_base is workgroup-uniform
            const int i = c0 + tid;
            bool valid = false; unsigned int kv = 0;
            if (i < A) kv = key(i, valid);
            int tot;
            const bool eq = valid && kv == r.T;
            const int my = eq_base + block_rank_1024(eq, wcnt, tot);
            if (eq && my < r.need_eq && base + my < FA_MAX_BUCKET) s_idx[base + my] = i;
            eq_base += tot;
        }
        base += r.need_eq;
    }
    __syncthreads();
    if (tid == 0) {             // ascending anchor index: a fixed summation order for the bag (the atomic collection order is not)
        const int mm = base < FA_MAX_BUCKET ? base : FA_MAX_BUCKET;
        for (int x = 1; x < mm; ++x) {
            const int v = s_idx[x]; int y = x - 1;
            while (y >= 0 && s_idx[y] > v) { s_idx[y + 1] = s_idx[y]; --y; }
            s_idx[y + 1] = v;
        }
    }
    __syncthreads();
    const int m = base < bucket ? base : bucket;          // fewer than `bucket` anchors exist only in toy cases

    // (c) positive bag loss and the gradients of its members
    float score = 0.f, p = 0.f, w = 0.f;
    f32x4_t gsl = {0.f, 0.f, 0.f, 0.f};
    int a = 0;
    if (tid < m) {
        a = s_idx[tid];
        score = sigmoidf_(bf2f(logits[((long long)n * A + a) * K + cls]));
        const Box ab = ld_box(anchors + 4ll * a);
        const f32x4_t tgt = encode_dev(ab, gb, coder);
        const f32x4_t pr = ld_offsets(offsets, pix0, a, apix, box_ld);
        float rl = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float x = pr[k] - tgt[k], ax = fabsf(x);
            if (beta < 1e-5f) { rl += ax; gsl[k] = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
            else if (ax < beta) { rl += 0.5f * x * x / beta; gsl[k] = x / beta; }
            else { rl += ax - 0.5f * beta; gsl[k] = x > 0.f ? 1.f : -1.f; }
        }
        rl *= reg_weight;
        p = score * expf(-rl);
        w = 1.f / (1.f - p);
        s_p[tid] = p; s_w[tid] = w;
    }
    __syncthreads();
    if (tid == 0) {
        float W = 0.f, S = 0.f;
        for (int j = 0; j < m; ++j) { W += s_w[j]; S += s_w[j] * s_p[j]; }
        s_bag[0] = W; s_bag[1] = S / W;
        const float nf = fmaxf(1.f, (float)total_fg(num_gt, N));
        pos_loss[slot] = -safelogf_(S / W) * alpha / nf;
    }
    __syncthreads();
    if (tid < bucket) {
        float* gq = bag_grad + ((long long)slot * bucket + tid) * 5;
        int idx = -1;
        float dl = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
        if (tid < m) {
            const float W = s_bag[0], bag = s_bag[1];
            const float nf = fmaxf(1.f, (float)total_fg(num_gt, N));
            // d bag / d p_j = (w_j + w_j^2 (p_j - bag)) / W ;  d(-log(max(bag, tiny))) / d bag = -1 / bag (0 below tiny)
            const float dbag = bag > FLT_MIN ? -1.f / bag : 0.f;
            const float dp = dbag * (w + w * w * (p - bag)) / W * (alpha / nf);
            dl = dp * p * (1.f - score);                       // p = score * e: dp/dlogit = e * score (1 - score) = p (1 - score)
            const float dr = -dp * p * reg_weight;             // dp / d reg_loss = -p
            d0 = dr * gsl[0]; d1 = dr * gsl[1]; d2 = dr * gsl[2]; d3 = dr * gsl[3];
            idx = a;
        }
        bag_idx[(long long)slot * bucket + tid] = idx;
        gq[0] = dl; gq[1] = d0; gq[2] = d1; gq[3] = d2; gq[4] = d3;
    }
}

__global__ __launch_bounds__(256) void fa_neg_kernel(const bf16_raw* __restrict__ logits, const bf16_raw* __restrict__ offsets,
                                                     int box_ld, int apix, const float* __restrict__ anchors, int A, int K,
                                                     const float* __restrict__ gt, const int* __restrict__ num_gt, int N, int Gmax,
                                                     Coder coder, float iou_thresh, int bucket, float alpha, float gamma,
                                                     const float* __restrict__ thresh2, float* __restrict__ neg_part,
                                                     bf16_raw* __restrict__ d_logits) {
    extern __shared__ unsigned long long bp[];            // [FA_NEG_ANCH][K]: (gt index + 1) << 32 | float bits of the box probability
    __shared__ float red[4];
    const int tid = threadIdx.x, n = blockIdx.y;
    const int a0 = blockIdx.x * FA_NEG_ANCH;
    const int G = num_gt[n];
    for (int i = tid; i < FA_NEG_ANCH * K; i += 256) bp[i] = 0ull;
    __syncthreads();
    {
        const int la = tid & 63, a = a0 + la;
        if (a < A && G > 0) {
            const Box ab = ld_box(anchors + 4ll * a);
            const f32x4_t pb = decode_dev(ab, ld_offsets(offsets, (long long)n * (A / apix), a, apix, box_ld), coder);
            const Box p{pb[0], pb[1], pb[2], pb[3]};
            const float parea = box_area(p);
            for (int g = tid >> 6; g < G; g += 4) {
                const float* gp = gt + ((long long)n * Gmax + g) * 5;
                const Box gb = ld_gt(gp);
                const float ov = box_iou_dev(gb, box_area(gb), p, parea);
                const float t2 = thresh2[n * Gmax + g];
                const float prob = fminf(fmaxf((ov - iou_thresh) / (t2 - iou_thresh), 0.f), 1.f);
                if (prob != 0.f)
                    atomicMax(&bp[la * K + ((int)gp[4] - 1)], ((unsigned long long)(g + 1) << 32) | __float_as_uint(prob));
            }
        }
    }
    __syncthreads();
    const float nf = fmaxf(1.f, (float)total_fg(num_gt, N) * (float)bucket);
    const float scale = (1.f - alpha) / nf;
    float acc = 0.f;
    const int nel = (A - a0 < FA_NEG_ANCH ? A - a0 : FA_NEG_ANCH) * K;
    const long long e0 = ((long long)n * A + a0) * K;
    auto elem = [&](float x, float bpv, float& g) -> float {
        if (gamma == 2.f) {
            // round 5: the default focal exponent on the raw transcendental instructions (v_exp_f32 / v_log_f32 / v_rcp_f32, 1 ulp), as
            // focal_g2_kernel (losses.hip): expf / logf / the two divisions compile to ~80 vector instructions per logit, this to ~25
            const float e = __builtin_amdgcn_exp2f(-x * 1.4426950408889634f);
            const float s = __builtin_amdgcn_rcpf(1.f + e);
            const float keep = 1.f - bpv;
            const float q = s * keep;
            const float om = 1.f - q;
            const float nl = -0.6931471805599453f * __builtin_amdgcn_logf(fmaxf(om, FLT_MIN));
            const float qg = q * q;
            const float dq = 2.f * q * nl + (om > FLT_MIN ? qg * __builtin_amdgcn_rcpf(om) : 0.f);
            g = scale * dq * s * (1.f - s) * keep;
            return qg * nl;
        }
        const float s = sigmoidf_(x);
        const float keep = 1.f - bpv;
        const float q = s * keep;
        const float om = 1.f - q;
        const float nl = -safelogf_(om);
        const float qg = gamma == 2.f ? q * q : (q > 0.f ? expf(gamma * logf(q)) : 0.f);
        const float qg1 = gamma == 2.f ? q : (q > 0.f ? expf((gamma - 1.f) * logf(q)) : 0.f);
        const float dq = gamma * qg1 * nl + (om > FLT_MIN ? qg / om : 0.f);
        g = scale * dq * s * (1.f - s) * keep;
        return qg * nl;
    };
    if (K % 8 == 0) {           // 16-byte logit / gradient accesses (e0 is a multiple of 8)
        for (int i = tid; i < nel / 8; i += 256) {
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(logits + e0 + 8ll * i);
            u32x4_t o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float g0, g1;
                acc += elem(bf_lo(v[j]), __uint_as_float((unsigned int)(bp[8 * i + 2 * j] & 0xffffffffull)), g0);
                acc += elem(bf_hi(v[j]), __uint_as_float((unsigned int)(bp[8 * i + 2 * j + 1] & 0xffffffffull)), g1);
                o[j] = pack_bf2(g0, g1);
            }
            *reinterpret_cast<u32x4_t*>(d_logits + e0 + 8ll * i) = o;
        }
    } else {
        for (int i = tid; i < nel; i += 256) {
            float g;
            acc += elem(bf2f(logits[e0 + i]), __uint_as_float((unsigned int)(bp[i] & 0xffffffffull)), g);
            d_logits[e0 + i] = f2bf(g);
        }
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) neg_part[(long long)n * gridDim.x + blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
}

__global__ __launch_bounds__(256) void fa_apply_kernel(const int* __restrict__ num_gt, int N, int Gmax, int A, int K, int apix,
                                                       int box_ld, int bucket, const float* __restrict__ gt,
                                                       const int* __restrict__ bag_idx, const float* __restrict__ bag_grad,
                                                       const float* __restrict__ pos_loss, const float* __restrict__ neg_part,
                                                       int neg_blocks, bf16_raw* __restrict__ d_logits,
                                                       bf16_raw* __restrict__ d_offsets, float* __restrict__ loss_out) {
    __shared__ float red[4];
    const int tid = threadIdx.x, n = blockIdx.x;
    const int G = num_gt[n];
    for (int g = 0; g < G; ++g) {
        const int slot = n * Gmax + g;
        const int cls = (int)gt[(long long)slot * 5 + 4] - 1;
        if (tid < bucket) {
            const int a = bag_idx[(long long)slot * bucket + tid];
            if (a >= 0) {
                const float* gq = bag_grad + ((long long)slot * bucket + tid) * 5;
                bf16_raw* dl = d_logits + ((long long)n * A + a) * K + cls;
                *dl = f2bf(bf2f(*dl) + gq[0]);
                bf16_raw* dp = d_offsets + ((long long)n * (A / apix) + a / apix) * box_ld + (a % apix) * 4;
#pragma unroll
                for (int k = 0; k < 4; ++k) dp[k] = f2bf(bf2f(dp[k]) + gq[1 + k]);
            }
        }
        __syncthreads();
    }
    if (n == 0) {           // fixed-order loss reductions
        float s = 0.f;
        for (int i = tid; i < N * Gmax; i += 256) s += pos_loss[i];
        s = wave_sum(s);
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) loss_out[0] = (red[0] + red[1]) + (red[2] + red[3]);
        __syncthreads();
        s = 0.f;
        for (int i = tid; i < N * neg_blocks; i += 256) s += neg_part[i];
        s = wave_sum(s);
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) loss_out[1] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

}  // namespace

extern "C" size_t bd_freeanchor_workspace_bytes(int N, int Gmax, int bucket, int A) {
    if (N <= 0 || Gmax <= 0 || bucket <= 0 || A <= 0) return 256;
    return fa_layout(N, Gmax, bucket, A).total;
}

extern "C" int bd_freeanchor_loss_fwd_bwd(const void* logits, const void* offsets, int box_ld, int anchors_per_pix,
                                          const float* anchors, int A, int K, const float* gt, const int32_t* num_gt, int N,
                                          int Gmax, const float* mean4, const float* std4, float iou_thresh, int bucket,
                                          float beta, float reg_weight, float alpha, float gamma, float* loss_out,
                                          void* d_logits, void* d_offsets, void* ws, size_t ws_bytes, bd_stream_t stream) {
    BD_REQUIRE(logits && offsets && anchors && gt && num_gt && loss_out && d_logits && d_offsets && ws, "freeanchor: null pointer");
    BD_REQUIRE(N > 0 && Gmax > 0 && A > 0 && K > 0, "freeanchor: empty problem (N=%d Gmax=%d A=%d K=%d)", N, Gmax, A, K);
    BD_REQUIRE(bucket > 0 && bucket <= FA_MAX_BUCKET, "freeanchor: bucket=%d must be in 1..%d", bucket, FA_MAX_BUCKET);
    BD_REQUIRE(anchors_per_pix > 0 && A % anchors_per_pix == 0 && box_ld >= 4 * anchors_per_pix && box_ld % 4 == 0,
               "freeanchor: A=%d / anchors_per_pix=%d / box_ld=%d inconsistent", A, anchors_per_pix, box_ld);
    BD_REQUIRE((size_t)FA_NEG_ANCH * K * 8 <= 64 * 1024, "freeanchor: K=%d classes exceed the LDS tile", K);
    const FaWs w = fa_layout(N, Gmax, bucket, A);
    BD_REQUIRE(ws_bytes >= w.total, "freeanchor: workspace %zu < %zu bytes", ws_bytes, w.total);
    unsigned char* wb = (unsigned char*)ws;
    float* thresh2 = (float*)(wb + w.thresh2);
    int* bag_idx = (int*)(wb + w.bag_idx);
    float* bag_grad = (float*)(wb + w.bag_grad);
    float* pos_loss = (float*)(wb + w.pos_loss);
    float* neg_part = (float*)(wb + w.neg_part);
    const Coder coder = make_coder(mean4, std4);
    hipStream_t st = (hipStream_t)stream;
    const int neg_blocks = cdiv(A, FA_NEG_ANCH);
    (void)hipMemsetAsync(d_offsets, 0, (size_t)N * (A / anchors_per_pix) * box_ld * sizeof(bf16_raw), st);
    unsigned int* maxiou = (unsigned int*)(wb + w.maxiou);
    (void)hipMemsetAsync(maxiou, 0, sizeof(unsigned int) * N * Gmax, st);
    hipLaunchKernelGGL(fa_maxiou_kernel, dim3(cdiv(A, 256), N), dim3(256), 0, st, (const bf16_raw*)offsets, box_ld, anchors_per_pix, anchors, A,
                       gt, num_gt, Gmax, coder, maxiou);
    hipLaunchKernelGGL(fa_gt_kernel, dim3(Gmax, N), dim3(1024), 0, st, (const bf16_raw*)logits, (const bf16_raw*)offsets, box_ld,
                       anchors_per_pix, anchors, A, K, gt, num_gt, N, Gmax, coder, iou_thresh, bucket, beta, reg_weight, alpha,
                       (const unsigned int*)maxiou, thresh2, bag_idx, bag_grad, pos_loss);
    hipLaunchKernelGGL(fa_neg_kernel, dim3(neg_blocks, N), dim3(256), (size_t)FA_NEG_ANCH * K * 8, st, (const bf16_raw*)logits,
                       (const bf16_raw*)offsets, box_ld, anchors_per_pix, anchors, A, K, gt, num_gt, N, Gmax, coder, iou_thresh,
                       bucket, alpha, gamma, thresh2, neg_part, (bf16_raw*)d_logits);
    hipLaunchKernelGGL(fa_apply_kernel, dim3(N), dim3(256), 0, st, num_gt, N, Gmax, A, K, anchors_per_pix, box_ld, bucket, gt, bag_idx,
                       bag_grad, pos_loss, neg_part, neg_blocks, (bf16_raw*)d_logits, (bf16_raw*)d_offsets, loss_out);
    BD_CHECK_LAUNCH("bd_freeanchor_loss_fwd_bwd");
    return BD_OK;
}
