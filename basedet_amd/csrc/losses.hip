// Fused loss forward + gradient kernels (bf16 predictions in, bf16 gradients out, fp32 math).
// They replace the dense one-hot target + boolean-mask gathers of the reference
// (models/det/retinanet.py:144-162, models/det/fcos.py:146-170) by reading the int32 anchor labels directly.
#include "common.h"

namespace {

__device__ __forceinline__ float load_norm(const void* norm, int is_float) {
    float v = is_float ? *reinterpret_cast<const float*>(norm) : (float)*reinterpret_cast<const int*>(norm);
    return fmaxf(v, 1.f);   // F.maximum(1, num_fg)
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// sigmoid_focal_loss (layers/losses/sigmoid_focal_loss.py:30-35) + binary_cross_entropy (cross_entropy.py:26)
__device__ __forceinline__ void focal_elem(float x, bool t, float alpha, float gamma, float& loss, float& grad) {
    // hardware exp/log/rcp (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp): the kernel is HBM-bound only if the
    // per-logit VALU work stays small; log(1 + e) with e in (0, 1] needs no log1p (absolute error < 1e-7)
    const float e = __expf(-fabsf(x));
    const float l1p = __logf(1.f + e);
    const float ls_pos = fminf(x, 0.f) - l1p;     // logsigmoid(x)
    const float ls_neg = fminf(-x, 0.f) - l1p;    // logsigmoid(-x)
    const float inv = __frcp_rn(1.f + e);
    const float p = x >= 0.f ? inv : e * inv;     // sigmoid(x)
    const float ce = t ? -ls_pos : -ls_neg;
    const float pt = t ? 1.f - p : p;             // t*(1-p) + (1-t)*p
    const float a = alpha >= 0.f ? (t ? alpha : 1.f - alpha) : 1.f;
    const float dce = p - (t ? 1.f : 0.f);
    const float dpt = (t ? -1.f : 1.f) * p * (1.f - p);
    float mod, dmod;
    if (gamma == 2.f) { mod = pt * pt; dmod = 2.f * pt; }
    else if (gamma == 0.f) { mod = 1.f; dmod = 0.f; }
    else { mod = powf(pt, gamma); dmod = gamma * powf(pt, gamma - 1.f); }
    loss = a * ce * mod;
    grad = a * (dce * mod + ce * dmod * dpt);
}

// gamma == 2 (the configured value, retinanet_cfg.py:30): with z = x for a negative target and -x for the positive one, q = sigmoid(z)
// is the probability of the WRONG answer and  ce = log(1 + e^-|x|) + max(z, 0),  loss = a * ce * q^2,
// d loss / dx = s * a * q^2 * (q + 2 ce (1 - q))  (s = +1 / -1, a = 1 - alpha / alpha): ~15 vector instructions + exp, log, rcp.
// Hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, 1 ulp): `__expf`, `__logf` and `__frcp_rn` compiled to their denormal-safe and
// correctly rounded forms -- range scaling with compares, selects and ldexp around exp / log, the ten-instruction IEEE division sequence for
// the reciprocal -- ~60 vector instructions per logit for a result that is rounded to bf16 (gradient) or summed over 2.6e8 terms (loss).
// The arguments here are harmless: e^-|x| underflows to 0 for |x| > 87 (q = 0 or 1, ce = max(z, 0): the right limits), 1 <= 1 + e <= 2.
__device__ __forceinline__ void focal_g2(float x, float z, float a_signed, float a_abs, float& loss, float& grad) {
    const float e = __builtin_amdgcn_exp2f(-fabsf(x) * 1.4426950408889634f);
    const float t1 = 1.f + e;
    const float inv = __builtin_amdgcn_rcpf(t1);
    const float q = z >= 0.f ? inv : e * inv;
    const float ce = __builtin_amdgcn_logf(t1) * 0.6931471805599453f + fmaxf(z, 0.f);
    const float q2 = q * q;
    loss = a_abs * ce * q2;
    grad = a_signed * q2 * (q + 2.f * ce * (1.f - q));
}

// two negatives at once (z = x, a_signed = a_abs = a): the arithmetic around the six transcendentals as packed fp32 instructions
typedef __attribute__((ext_vector_type(2))) float f32x2_l;
__device__ __forceinline__ void focal_g2_neg2(float x0, float x1, float a, f32x2_l& loss, f32x2_l& grad) {
    const f32x2_l e = {__builtin_amdgcn_exp2f(-fabsf(x0) * 1.4426950408889634f), __builtin_amdgcn_exp2f(-fabsf(x1) * 1.4426950408889634f)};
    const f32x2_l t1 = e + 1.f;
    const f32x2_l inv = {__builtin_amdgcn_rcpf(t1[0]), __builtin_amdgcn_rcpf(t1[1])};
    const f32x2_l einv = e * inv;
    const f32x2_l q = {x0 >= 0.f ? inv[0] : einv[0], x1 >= 0.f ? inv[1] : einv[1]};
    const f32x2_l l2 = {__builtin_amdgcn_logf(t1[0]), __builtin_amdgcn_logf(t1[1])};
    const f32x2_l mz = {fmaxf(x0, 0.f), fmaxf(x1, 0.f)};
    const f32x2_l ce = l2 * 0.6931471805599453f + mz;
    const f32x2_l aq2 = q * q * a;
    loss = aq2 * ce;
    grad = aq2 * ((ce + ce) * (1.f - q) + q);
}

// The 8 logits of a vector are treated as negatives; the (at most one) positive among them is recomputed in a rare branch.
__global__ __launch_bounds__(256) void focal_g2_kernel(const bf16_raw* __restrict__ logits, const int* __restrict__ labels,
                                                       long long rows, int K, float alpha, const void* norm, int norm_is_float,
                                                       float grad_scale, float* __restrict__ loss_sum, bf16_raw* __restrict__ dlogits) {
    __shared__ float red[4];
    const float inv_norm = 1.f / load_norm(norm, norm_is_float);
    const float gs = grad_scale * inv_norm;
    const float a_neg = alpha >= 0.f ? 1.f - alpha : 1.f, a_pos = alpha >= 0.f ? alpha : 1.f;
    const int kv = K / 8;
    const long long nvec = rows * kv;
    float acc = 0.f;
    const long long stride = (long long)gridDim.x * 256;
    const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
    long long row = i0 / kv;
    int ch = (int)(i0 - row * kv);
    const long long dq = stride / kv;
    const int dr = (int)(stride - dq * kv);
    // four vectors per pass, their label and logit loads issued before any arithmetic (the logits are read whatever the label says:
    // ignored rows are rare, and a load that waits for the label serialises two memory latencies per 8 logits)
    constexpr int U = 4;
    for (long long i = i0; i < nvec; i += U * stride) {
        int lab[U], c0[U];
        u32x4_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = i + u * stride < nvec;
            c0[u] = ch * 8;
            lab[u] = live ? labels[row] : -1;
            v[u] = live ? *reinterpret_cast<const u32x4_t*>(logits + (i + u * stride) * 8) : (u32x4_t){0u, 0u, 0u, 0u};
            row += dq; ch += dr;
            if (ch >= kv) { ch -= kv; ++row; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i + u * stride >= nvec) break;
            u32x4_t o = {0u, 0u, 0u, 0u};
            if (lab[u] >= 0) {
                float g[8];
                f32x2_l lsum = {0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    f32x2_l l2, g2;
                    focal_g2_neg2(bf_lo(v[u][k]), bf_hi(v[u][k]), a_neg, l2, g2);
                    g[2 * k] = g2[0]; g[2 * k + 1] = g2[1];
                    lsum += l2;
                }
                acc += lsum[0] + lsum[1];
                const int pos = lab[u] - 1 - c0[u];
                if (pos >= 0 && pos < 8) {                      // this vector holds the row's positive class: redo that one element
                    const unsigned w = v[u][pos >> 1];
                    const float x = (pos & 1) ? bf_hi(w) : bf_lo(w);
                    float ln, gn, lp, gp;
                    focal_g2(x, x, a_neg, a_neg, ln, gn);
                    focal_g2(x, -x, -a_pos, a_pos, lp, gp);
                    acc += lp - ln;
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (k == pos) g[k] = gp;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = pack_bf2(g[2 * k] * gs, g[2 * k + 1] * gs);
            }
            *reinterpret_cast<u32x4_t*>(dlogits + (i + u * stride) * 8) = o;
        }
    }
    const float s = block_sum_256(acc, red);
    if (threadIdx.x == 0 && s != 0.f) atomicAdd(loss_sum, s * inv_norm);
}

__global__ __launch_bounds__(256) void focal_kernel(const bf16_raw* __restrict__ logits, const int* __restrict__ labels,
                                                    long long rows, int K, float alpha, float gamma, const void* norm,
                                                    int norm_is_float, float grad_scale, float* __restrict__ loss_sum,
                                                    bf16_raw* __restrict__ dlogits) {
    __shared__ float red[4];
    const float inv_norm = 1.f / load_norm(norm, norm_is_float);
    const float gs = grad_scale * inv_norm;
    const int kv = K / 8;
    const long long nvec = rows * kv;
    float acc = 0.f;
    // (row, chunk) of vector i advance by a fixed (quotient, remainder) per grid stride: one 64-bit division per thread instead of
    // one per 8 logits (a software division is ~100 VALU instructions -- more than the loss itself)
    const long long stride = (long long)gridDim.x * 256;
    const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
    long long row = i0 / kv;
    int ch = (int)(i0 - row * kv);
    const long long dq = stride / kv;
    const int dr = (int)(stride - dq * kv);
    for (long long i = i0; i < nvec; i += stride) {
        const int c0 = ch * 8;
        const int lab = labels[row];
        row += dq; ch += dr;
        if (ch >= kv) { ch -= kv; ++row; }
        u32x4_t o = {0u, 0u, 0u, 0u};
        if (lab >= 0) {
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(logits + i * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float l0, g0, l1, g1;
                focal_elem(bf_lo(v[k]), lab - 1 == c0 + 2 * k, alpha, gamma, l0, g0);
                focal_elem(bf_hi(v[k]), lab - 1 == c0 + 2 * k + 1, alpha, gamma, l1, g1);
                acc += l0 + l1;
                o[k] = pack_bf2(g0 * gs, g1 * gs);
            }
        }
        *reinterpret_cast<u32x4_t*>(dlogits + i * 8) = o;
    }
    const float s = block_sum_256(acc, red);
    if (threadIdx.x == 0 && s != 0.f) atomicAdd(loss_sum, s * inv_norm);
}

// smooth_l1_loss (layers/losses/smooth_l1_loss.py:26-33) on fg rows; prediction addressed as
// pred[(row / A) * ld + (row % A) * 4 + k]; pad slots (ld/4 - A per pixel) get zero gradient.
__global__ __launch_bounds__(256) void smooth_l1_kernel(const bf16_raw* __restrict__ pred, const float* __restrict__ target,
                                                        const int* __restrict__ labels, long long pixels, int A, int ld,
                                                        float beta, const void* norm, int norm_is_float, float weight,
                                                        float* __restrict__ loss_sum, bf16_raw* __restrict__ dpred) {
    __shared__ float red[4];
    const float inv_norm = 1.f / load_norm(norm, norm_is_float);
    const float gs = weight * inv_norm;
    const int slots = ld / 4;
    const long long total = pixels * slots;
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pix = i / slots;
        const int a = (int)(i - pix * slots);
        u32x2_t o = {0u, 0u};
        if (a < A) {
            const long long row = pix * A + a;
            if (labels[row] > 0) {
                const u32x2_t pv = *reinterpret_cast<const u32x2_t*>(pred + pix * ld + a * 4);
                const f32x4_t tv = *reinterpret_cast<const f32x4_t*>(target + row * 4);
                const float x[4] = {bf_lo(pv[0]) - tv[0], bf_hi(pv[0]) - tv[1], bf_lo(pv[1]) - tv[2], bf_hi(pv[1]) - tv[3]};
                float g[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float ax = fabsf(x[k]);
                    if (beta < 1e-5f) { acc += ax; g[k] = x[k] > 0.f ? 1.f : (x[k] < 0.f ? -1.f : 0.f); }
                    else if (ax < beta) { acc += 0.5f * x[k] * x[k] / beta; g[k] = x[k] / beta; }
                    else { acc += ax - 0.5f * beta; g[k] = x[k] > 0.f ? 1.f : -1.f; }
                }
                o[0] = pack_bf2(g[0] * gs, g[1] * gs);
                o[1] = pack_bf2(g[2] * gs, g[3] * gs);
            }
        }
        *reinterpret_cast<u32x2_t*>(dpred + pix * ld + a * 4) = o;
    }
    const float s = block_sum_256(acc, red);
    if (threadIdx.x == 0 && s != 0.f) atomicAdd(loss_sum, s * gs);
}

// iou_loss(box_mode="ltrb", loss_type="giou") (layers/losses/iou_loss.py:9-56, 59-105), weighted by centerness
__global__ __launch_bounds__(256) void giou_ltrb_kernel(const bf16_raw* __restrict__ pred, const float* __restrict__ target,
                                                        const float* __restrict__ wgt, const int* __restrict__ labels,
                                                        long long rows, const float* __restrict__ norm, float loss_weight,
                                                        float* __restrict__ loss_sum, bf16_raw* __restrict__ dpred) {
    __shared__ float red[4];
    const float eps = 1e-8f;
    const float gs = loss_weight / fmaxf(*norm, 1.f);
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < rows; i += (long long)gridDim.x * 256) {
        u32x2_t o = {0u, 0u};
        if (labels[i] > 0) {
            const u32x2_t pv = *reinterpret_cast<const u32x2_t*>(pred + i * 4);
            const f32x4_t t = *reinterpret_cast<const f32x4_t*>(target + i * 4);
            const float p[4] = {bf_lo(pv[0]), bf_hi(pv[0]), bf_lo(pv[1]), bf_hi(pv[1])};   // l, t, r, b
            // areas
            const float pw = p[0] + p[2], ph = p[1] + p[3];
            const float pwc = fmaxf(pw, 0.f), phc = fmaxf(ph, 0.f);
            const float a1 = pwc * phc;
            const float a2 = fmaxf(t[0] + t[2], 0.f) * fmaxf(t[1] + t[3], 0.f);
            // intersection
            const float wi_raw = fminf(p[2], t[2]) + fminf(p[0], t[0]);
            const float hi_raw = fminf(p[3], t[3]) + fminf(p[1], t[1]);
            const float wi = fmaxf(wi_raw, 0.f), hi = fmaxf(hi_raw, 0.f);
            const float ai = wi * hi;
            const float au = a1 + a2 - ai;
            const float auc = fmaxf(au, eps);
            const float iou = ai / auc;
            // hull
            const float gw = fmaxf(p[2], t[2]) + fmaxf(p[0], t[0]);
            const float gh = fmaxf(p[3], t[3]) + fmaxf(p[1], t[1]);
            const float ac = gw * gh;
            const float acc_ = fmaxf(ac, eps);
            const float giou = iou - (ac - au) / acc_;
            const float w = wgt ? wgt[i] : 1.f;
            acc += (1.f - giou) * w;
            // gradients w.r.t. p[k]; index k: 0 l, 1 t, 2 r, 3 b   (l,r are "x" sides, t,b are "y" sides)
            float g[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool isx = (k & 1) == 0;
                const float da1 = isx ? (pw > 0.f ? phc : 0.f) : (ph > 0.f ? pwc : 0.f);
                const float dmin = p[k] < t[k] ? 1.f : 0.f;                 // d min(p,t)/dp
                const float dai = isx ? (wi_raw > 0.f ? dmin * hi : 0.f) : (hi_raw > 0.f ? dmin * wi : 0.f);
                const float dau = da1 - dai;
                const float dauc = au > eps ? dau : 0.f;
                const float diou = (dai * auc - ai * dauc) / (auc * auc);
                const float dmax = p[k] > t[k] ? 1.f : 0.f;                 // d max(p,t)/dp
                const float dac = isx ? dmax * gh : dmax * gw;
                const float dacc = ac > eps ? dac : 0.f;
                const float dterm = ((dac - dau) * acc_ - (ac - au) * dacc) / (acc_ * acc_);
                g[k] = -(diou - dterm) * w * gs;
            }
            o[0] = pack_bf2(g[0], g[1]);
            o[1] = pack_bf2(g[2], g[3]);
        }
        *reinterpret_cast<u32x2_t*>(dpred + i * 4) = o;
    }
    const float s = block_sum_256(acc, red);
    if (threadIdx.x == 0 && s != 0.f) atomicAdd(loss_sum, s * gs);
}

// binary_cross_entropy with logits on fg rows (layers/losses/cross_entropy.py:26)
__global__ __launch_bounds__(256) void bce_kernel(const bf16_raw* __restrict__ pred, int ld, int off, const float* __restrict__ target,
                                                  const int* __restrict__ labels, long long rows, const float* __restrict__ norm,
                                                  float* __restrict__ loss_sum, bf16_raw* __restrict__ dpred) {
    __shared__ float red[4];
    const float gs = 1.f / fmaxf(*norm, 1.f);
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < rows; i += (long long)gridDim.x * 256) {
        float g = 0.f;
        if (labels[i] > 0) {
            const float x = bf2f(pred[i * ld + off]), t = target[i];
            const float e = __expf(-fabsf(x));
            const float l1p = log1pf(e);
            const float ls_pos = fminf(x, 0.f) - l1p, ls_neg = fminf(-x, 0.f) - l1p;
            acc += -(t * ls_pos + (1.f - t) * ls_neg);
            const float inv = 1.f / (1.f + e);
            const float p = x >= 0.f ? inv : e * inv;
            g = (p - t) * gs;
        }
        dpred[i] = f2bf(g);
    }
    const float s = block_sum_256(acc, red);
    if (threadIdx.x == 0 && s != 0.f) atomicAdd(loss_sum, s * gs);
}

// RPN losses (models/det/rpn.py:113-131): binary cross entropy with logits, mean over the sampled (label >= 0) anchors,
// and smooth-L1 over the positive anchors divided by max(num_valid, 1).  One thread per (pixel, cell anchor); the fused
// prediction row holds [cls_off + a] logits and [box_off + 4a .. +3] offsets; padding channels are never written.
__global__ __launch_bounds__(256) void rpn_loss_kernel(const bf16_raw* __restrict__ raw, int ldc, int A, int cls_off, int box_off,
                                                       const int* __restrict__ labels, const float* __restrict__ targets,
                                                       long long rows, float beta, const int* __restrict__ num_valid,
                                                       float* __restrict__ loss, bf16_raw* __restrict__ draw) {
    __shared__ float red[4];
    const float gs = 1.f / fmaxf((float)*num_valid, 1.f);
    float acc_c = 0.f, acc_b = 0.f;
    const long long total = rows * A;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pix = i / A;
        const int a = (int)(i - pix * A);
        const int l = labels[i];
        const bf16_raw* rp = raw + pix * ldc;
        bf16_raw* dp = draw + pix * ldc;
        float gc = 0.f;
        float gb[4] = {0.f, 0.f, 0.f, 0.f};
        if (l >= 0) {
            const float x = bf2f(rp[cls_off + a]), t = (float)l;
            const float e = __expf(-fabsf(x));
            const float l1p = log1pf(e);
            const float ls_pos = fminf(x, 0.f) - l1p, ls_neg = fminf(-x, 0.f) - l1p;
            acc_c += -(t * ls_pos + (1.f - t) * ls_neg);
            const float inv = 1.f / (1.f + e);
            const float p = x >= 0.f ? inv : e * inv;
            gc = (p - t) * gs;
        }
        if (l > 0) {
            const f32x4_t tv = *reinterpret_cast<const f32x4_t*>(targets + i * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float x = bf2f(rp[box_off + a * 4 + k]) - tv[k];
                const float ax = fabsf(x);
                float g;
                if (beta < 1e-5f) { acc_b += ax; g = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
                else if (ax < beta) { acc_b += 0.5f * x * x / beta; g = x / beta; }
                else { acc_b += ax - 0.5f * beta; g = x > 0.f ? 1.f : -1.f; }
                gb[k] = g * gs;
            }
        }
        dp[cls_off + a] = f2bf(gc);
#pragma unroll
        for (int k = 0; k < 4; ++k) dp[box_off + a * 4 + k] = f2bf(gb[k]);
    }
    const float sc = block_sum_256(acc_c, red);
    __syncthreads();
    const float sb = block_sum_256(acc_b, red);
    if (threadIdx.x == 0) {
        if (sc != 0.f) atomicAdd(loss, sc * gs);
        if (sb != 0.f) atomicAdd(loss + 1, sb * gs);
    }
}

// RCNN losses (layers/head/rcnn.py:65-83): softmax cross entropy over K+1 classes (mean over the sampled RoIs) and
// smooth-L1 on the deltas of the ground-truth class of the foreground RoIs, divided by the number of samples.
// One wave per RoI row; the fused prediction row holds K+1 logits at [0, K] and K*4 deltas from `box_off`.
// Rows with label < 0 are empty sample slots: zero gradient, no loss.
__global__ __launch_bounds__(256) void rcnn_loss_kernel(const bf16_raw* __restrict__ raw, int ld, int K, int box_off,
                                                        const int* __restrict__ labels, const float* __restrict__ targets,
                                                        int R, float beta, const int* __restrict__ num_samples,
                                                        float* __restrict__ loss, bf16_raw* __restrict__ draw) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float gs = 1.f / fmaxf((float)*num_samples, 1.f);
    const bf16_raw* rp = raw + (long long)r * ld;
    bf16_raw* dp = draw + (long long)r * ld;
    const int l = labels[r];
    if (l < 0) {
        for (int c = lane; c < ld; c += 64) dp[c] = 0;
        return;
    }
    const int nc = K + 1;
    float mx = -INFINITY;
    for (int c = lane; c < nc; c += 64) mx = fmaxf(mx, bf2f(rp[c]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float se = 0.f;
    for (int c = lane; c < nc; c += 64) se += expf(bf2f(rp[c]) - mx);
    se = wave_sum(se);
    const float lse = logf(se) + mx;
    for (int c = lane; c < nc; c += 64) {
        const float p = expf(bf2f(rp[c]) - lse);
        dp[c] = f2bf((p - (c == l ? 1.f : 0.f)) * gs);
    }
    for (int c = nc + lane; c < ld; c += 64) {
        float g = 0.f;
        const int q = c - box_off;
        if (l > 0 && q >= (l - 1) * 4 && q < l * 4) {
            const float x = bf2f(rp[c]) - targets[(long long)r * 4 + (q & 3)];
            if (beta < 1e-5f) g = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
            else if (fabsf(x) < beta) g = x / beta;
            else g = x > 0.f ? 1.f : -1.f;
            g *= gs;
        }
        dp[c] = f2bf(g);
    }
    if (lane == 0) {
        atomicAdd(loss, (lse - bf2f(rp[l])) * gs);
        if (l > 0) {
            float sb = 0.f;
            for (int k = 0; k < 4; ++k) {
                const float x = bf2f(rp[box_off + (l - 1) * 4 + k]) - targets[(long long)r * 4 + k];
                const float ax = fabsf(x);
                if (beta < 1e-5f) sb += ax;
                else if (ax < beta) sb += 0.5f * x * x / beta;
                else sb += ax - 0.5f * beta;
            }
            atomicAdd(loss + 1, sb * gs);
        }
    }
}

inline int loss_grid(long long n) {
    long long g = (n + 255) / 256;
    if (g < 1) g = 1;
    return (int)(g < 4096 ? g : 4096);
}

}  // namespace

static int focal_launch(const void* logits, const int32_t* labels, int64_t rows, int K, float alpha, float gamma, const void* norm,
                        int norm_is_float, float grad_scale, float* loss_sum, void* dlogits, bool general, bd_stream_t stream) {
    BD_REQUIRE(logits && labels && norm && loss_sum && dlogits, "focal_loss: null pointer");
    BD_REQUIRE(K > 0 && K % 8 == 0, "focal_loss: K=%d must be a multiple of 8", K);
    if (rows == 0) return BD_OK;
    if (gamma == 2.f && !general)
        hipLaunchKernelGGL(focal_g2_kernel, dim3(loss_grid(rows * (K / 8))), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_raw*)logits, labels, (long long)rows, K, alpha, norm, norm_is_float, grad_scale, loss_sum,
                           (bf16_raw*)dlogits);
    else
        hipLaunchKernelGGL(focal_kernel, dim3(loss_grid(rows * (K / 8))), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_raw*)logits, labels, (long long)rows, K, alpha, gamma, norm, norm_is_float, grad_scale,
                           loss_sum, (bf16_raw*)dlogits);
    BD_CHECK_LAUNCH("bd_focal_loss_fwd_bwd");
    return BD_OK;
}

extern "C" int bd_focal_loss_fwd_bwd(const void* logits, const int32_t* labels, int64_t rows, int K, float alpha, float gamma,
                                     const void* norm, int norm_is_float, float grad_scale, float* loss_sum, void* dlogits,
                                     bd_stream_t stream) {
    return focal_launch(logits, labels, rows, K, alpha, gamma, norm, norm_is_float, grad_scale, loss_sum, dlogits, false, stream);
}

// the general-gamma kernel whatever gamma is (the gamma == 2 instance of bd_focal_loss_fwd_bwd is checked against it)
extern "C" int bd_focal_loss_fwd_bwd_general(const void* logits, const int32_t* labels, int64_t rows, int K, float alpha, float gamma,
                                             const void* norm, int norm_is_float, float grad_scale, float* loss_sum, void* dlogits,
                                             bd_stream_t stream) {
    return focal_launch(logits, labels, rows, K, alpha, gamma, norm, norm_is_float, grad_scale, loss_sum, dlogits, true, stream);
}

extern "C" int bd_smooth_l1_fwd_bwd(const void* pred, const float* target, const int32_t* labels, int64_t pixels, int A,
                                    int ld, float beta, const void* norm, int norm_is_float, float weight, float* loss_sum,
                                    void* dpred, bd_stream_t stream) {
    BD_REQUIRE(pred && target && labels && norm && loss_sum && dpred, "smooth_l1: null pointer");
    BD_REQUIRE(A > 0 && ld >= 4 * A && ld % 4 == 0, "smooth_l1: ld=%d must be a multiple of 4 and >= 4*A", ld);
    if (pixels == 0) return BD_OK;
    hipLaunchKernelGGL(smooth_l1_kernel, dim3(loss_grid(pixels * (ld / 4))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_raw*)pred, target, labels, (long long)pixels, A, ld, beta, norm, norm_is_float, weight,
                       loss_sum, (bf16_raw*)dpred);
    BD_CHECK_LAUNCH("bd_smooth_l1_fwd_bwd");
    return BD_OK;
}

extern "C" int bd_giou_ltrb_fwd_bwd(const void* pred, const float* target, const float* weight, const int32_t* labels,
                                    int64_t rows, const float* norm, float loss_weight, float* loss_sum, void* dpred,
                                    bd_stream_t stream) {
    BD_REQUIRE(pred && target && labels && norm && loss_sum && dpred, "giou_ltrb: null pointer");
    if (rows == 0) return BD_OK;
    hipLaunchKernelGGL(giou_ltrb_kernel, dim3(loss_grid(rows)), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)pred,
                       target, weight, labels, (long long)rows, norm, loss_weight, loss_sum, (bf16_raw*)dpred);
    BD_CHECK_LAUNCH("bd_giou_ltrb_fwd_bwd");
    return BD_OK;
}

extern "C" int bd_bce_logits_fwd_bwd(const void* pred, int ld, int off, const float* target, const int32_t* labels, int64_t rows,
                                     const float* norm, float* loss_sum, void* dpred, bd_stream_t stream) {
    BD_REQUIRE(pred && target && labels && norm && loss_sum && dpred, "bce_logits: null pointer");
    BD_REQUIRE(ld >= 1 && off >= 0 && off < ld, "bce_logits: bad ld/off");
    if (rows == 0) return BD_OK;
    hipLaunchKernelGGL(bce_kernel, dim3(loss_grid(rows)), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)pred, ld, off, target,
                       labels, (long long)rows, norm, loss_sum, (bf16_raw*)dpred);
    BD_CHECK_LAUNCH("bd_bce_logits_fwd_bwd");
    return BD_OK;
}

extern "C" int bd_rpn_loss_fwd_bwd(const void* raw, int ldc, int A, int cls_off, int box_off, const int32_t* labels,
                                   const float* targets, int64_t rows, float beta, const int32_t* num_valid, float* loss2,
                                   void* draw, bd_stream_t stream) {
    BD_REQUIRE(raw && labels && targets && num_valid && loss2 && draw, "rpn_loss: null pointer");
    BD_REQUIRE(A > 0 && cls_off >= 0 && box_off >= 0 && cls_off + A <= ldc && box_off + 4 * A <= ldc, "rpn_loss: bad channel layout");
    if (rows == 0) return BD_OK;
    hipLaunchKernelGGL(rpn_loss_kernel, dim3(loss_grid(rows * A)), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)raw, ldc, A,
                       cls_off, box_off, labels, targets, (long long)rows, beta, num_valid, loss2, (bf16_raw*)draw);
    BD_CHECK_LAUNCH("bd_rpn_loss_fwd_bwd");
    return BD_OK;
}

extern "C" int bd_rcnn_loss_fwd_bwd(const void* raw, int ld, int K, int box_off, const int32_t* labels, const float* targets, int R,
                                    float beta, const int32_t* num_samples, float* loss2, void* draw, bd_stream_t stream) {
    BD_REQUIRE(raw && labels && targets && num_samples && loss2 && draw, "rcnn_loss: null pointer");
    BD_REQUIRE(K > 0 && box_off >= K + 1 && box_off + 4 * K <= ld, "rcnn_loss: bad channel layout");
    if (R == 0) return BD_OK;
    hipLaunchKernelGGL(rcnn_loss_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)raw, ld, K, box_off,
                       labels, targets, R, beta, num_samples, loss2, (bf16_raw*)draw);
    BD_CHECK_LAUNCH("bd_rcnn_loss_fwd_bwd");
    return BD_OK;
}
