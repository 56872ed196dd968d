// GroupNorm(32, C) + ReLU forward / backward for the FCOS PointHead towers (layers/head/point_head.py:47-58), and
// the per-level "relu(bbox_pred * scale_l) * stride_l" transform (:143) with its gradient.
//
// Activations are pixel-major NHWC bf16 with several pyramid levels per image; one GroupNorm instance (shared weights)
// normalises every (image, level, group) independently over H_l*W_l pixels x C/32 channels.  With C = 256 a group is
// exactly 8 channels = one 16-byte chunk per pixel, so every lane moves one chunk.  All passes are HBM-bound:
//   fwd: stats (read y) -> apply (read y, write z)            bwd: sums (read dz, y) -> apply (read dz, y, write dy)
// Round 5 (rounds 1-4: ten tensor passes per tower layer, 3.1 ms of the FCOS step):
//   * the backward no longer reads z: the ReLU gate z > 0 is recomputed from y with the forward's own arithmetic (gn_affine: pinned
//     fma / mul / sub, bf16 rounding keeps the sign of a normal float) -- seven passes become five;
//   * the partial stage is cut into fixed 128-pixel slots;
//   * measured and left OFF (bd_groupnorm_set_chunks, default 0 / 0 = the whole batch per launch): both directions image chunk by image
//     chunk (stats of chunk c, then apply of chunk c), so that the second kernel re-reads what the first has just pulled through the
//     256 MB Infinity Cache -- FCOS-R50 batch 16, one box: 622.7 img/s unchunked, 589.1 at 8 / 4 images per chunk, 544.0 at 4 / 2, 520.4
//     at 2 / 2 (profiles/r05_gn_chunks.txt): the smaller launches lose more in ramps and tails than the cache returns.
// Reductions are two-stage with a fixed partial order (no float atomics): bitwise reproducible, and independent of the chunking.
#include "common.h"

namespace {

constexpr int GN_SLOT_PX = 128;   // pixels per partial-stage slot
constexpr int MAXL = BD_MAX_SEGS;

// slot0[l] = first slot of level l inside an image's slot list (slot0[L] = slots per image)
struct GnLevels { int L; int off[MAXL]; int cnt[MAXL]; int slot0[MAXL + 1]; };

// z = relu(xhat * gamma + beta) with xhat = (y - mean) * rstd: ONE instruction sequence for the forward's value and the backward's gate
__device__ __forceinline__ float gn_xhat(float y, float mean, float rstd) { return __fmul_rn(__fsub_rn(y, mean), rstd); }
__device__ __forceinline__ float gn_affine(float y, float mean, float rstd, float gamma, float beta) {
    return __fmaf_rn(gn_xhat(y, mean, rstd), gamma, beta);
}

__device__ __forceinline__ int slot_level(const GnLevels& lv, int slot) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < MAXL; ++k)
        if (k < lv.L && slot >= lv.slot0[k]) l = k;
    return l;
}

// block = 256 threads = 32 groups x 8 pixel lanes; grid = (slots per image, images of the chunk)
// partial[(n * S + slot) * 32 + g][2] = (sum, sumsq)
__global__ __launch_bounds__(256) void gn_stats_partial_kernel(const bf16_raw* __restrict__ y, GnLevels lv, int ppi, int C,
                                                               float* __restrict__ partial) {
    __shared__ float red[256][2];
    const int g = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int slot = blockIdx.x, n = blockIdx.y, S = lv.slot0[lv.L];
    const int l = slot_level(lv, slot);
    const int cnt = lv.cnt[l];
    const int p0 = (slot - lv.slot0[l]) * GN_SLOT_PX;
    int p1 = p0 + GN_SLOT_PX;
    if (p1 > cnt) p1 = cnt;
    const bf16_raw* base = y + ((long long)n * ppi + lv.off[l]) * C + g * 8;
    float s = 0.f, ss = 0.f;
    // four pixels per thread and trip: four independent 16-byte loads in flight
    for (int p = p0 + pl; p < p1; p += 32) {
        u32x4_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = p + 8 * u;
            v[u] = q < p1 ? *reinterpret_cast<const u32x4_t*>(base + (long long)q * C) : (u32x4_t){0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a = bf_lo(v[u][k]), b = bf_hi(v[u][k]);
                s += a + b; ss += a * a + b * b;
            }
    }
    red[threadIdx.x][0] = s; red[threadIdx.x][1] = ss;
    __syncthreads();
    if (pl == 0) {
        float ts = 0.f, tss = 0.f;
        for (int k = 0; k < 8; ++k) { ts += red[k * 32 + g][0]; tss += red[k * 32 + g][1]; }
        float* o = partial + (((long long)n * S + slot) * 32 + g) * 2;
        o[0] = ts; o[1] = tss;
    }
}

// stats[(n*L + l)*32 + g] = (mean, rstd).  One workgroup per (image, level): GN_FQ lanes per group walk the level's slots (lane q: slots q,
// q + GN_FQ, ...), then the partial sums are added in lane order -- a fixed order; a single thread per (n, l, g) summing the 132 slots
// of the largest level one after the other took 37 us per launch (rocprofv3, first form of round 5).  Round 6: 32 lanes per group instead of 8 --
// the slots of the fused form are the convolution's 4 x 16-pixel patches (275 on the largest level at 800 x 1344): 16 -> 6 us per launch (rocprofv3).
constexpr int GN_FQ = 32;
__global__ __launch_bounds__(32 * GN_FQ) void gn_stats_final_kernel(const float* __restrict__ partial, GnLevels lv, int cpg, float eps,
                                                                    float* __restrict__ stats) {
    __shared__ float red[GN_FQ][32][2];
    const int g = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int l = blockIdx.x, n = blockIdx.y, S = lv.slot0[lv.L];
    float s = 0.f, ss = 0.f;
    for (int c = lv.slot0[l] + q; c < lv.slot0[l + 1]; c += GN_FQ) {
        const float* o = partial + (((long long)n * S + c) * 32 + g) * 2;
        s += o[0]; ss += o[1];
    }
    red[q][g][0] = s; red[q][g][1] = ss;
    __syncthreads();
    if (q == 0) {
        s = 0.f; ss = 0.f;
#pragma unroll
        for (int k = 0; k < GN_FQ; ++k) { s += red[k][g][0]; ss += red[k][g][1]; }
        const float inv = 1.f / ((float)lv.cnt[l] * (float)cpg);
        const float mean = s * inv;
        const float var = fmaxf(ss * inv - mean * mean, 0.f);
        const int i = (n * lv.L + l) * 32 + g;
        stats[i * 2] = mean;
        stats[i * 2 + 1] = rsqrtf(var + eps);
    }
}

__device__ __forceinline__ int level_of(const GnLevels& lv, int p) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < MAXL; ++k)
        if (k < lv.L && p >= lv.off[k]) l = k;
    return l;
}

// z = relu((y - mean) * rstd * gamma + beta); thread per (pixel, group)
// rev: walk the tensor from its END (the statistics pass before it walked from the start: its last ~100 MB are still in the Infinity Cache)
__global__ void gn_apply_kernel(const bf16_raw* __restrict__ y, const float* __restrict__ stats, const float* __restrict__ gamma,
                                const float* __restrict__ beta, GnLevels lv, int N, int ppi, int relu, int rev, bf16_raw* __restrict__ z) {
    const long long total = (long long)N * ppi * 32;
    for (long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += (long long)gridDim.x * blockDim.x) {
        const long long i = rev ? total - 1 - i0 : i0;
        const int g = (int)(i & 31);
        const long long pix = i >> 5;
        const int n = (int)(pix / ppi), p = (int)(pix - (long long)n * ppi);
        const int l = level_of(lv, p);
        const float mean = stats[((n * lv.L + l) * 32 + g) * 2], rstd = stats[((n * lv.L + l) * 32 + g) * 2 + 1];
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(y + i * 8);
        const f32x4_t g0 = *reinterpret_cast<const f32x4_t*>(gamma + g * 8), g1 = *reinterpret_cast<const f32x4_t*>(gamma + g * 8 + 4);
        const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(beta + g * 8), b1 = *reinterpret_cast<const f32x4_t*>(beta + g * 8 + 4);
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float xv = (k & 1) ? bf_hi(v[k >> 1]) : bf_lo(v[k >> 1]);
            o[k] = gn_affine(xv, mean, rstd, k < 4 ? g0[k & 3] : g1[k & 3], k < 4 ? b0[k & 3] : b1[k & 3]);
        }
        u32x4_t w;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            w[k] = relu ? pack_bf2(fmaxf(o[2 * k], 0.f), fmaxf(o[2 * k + 1], 0.f)) : pack_bf2(o[2 * k], o[2 * k + 1]);
        *reinterpret_cast<u32x4_t*>(z + i * 8) = w;
    }
}

// backward sums per (n, slot, g): A = sum dzm*gamma*xhat, B = sum dzm*gamma ; and per channel dgamma / dbeta partials
// pg[(n*S + slot)*32 + g][2], pc[((n*S + slot)*C + c)][2];  dzm = dz where relu(xhat*gamma + beta) > 0 (relu != 0), else dz
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const bf16_raw* __restrict__ dz, const bf16_raw* __restrict__ y,
                                                             const float* __restrict__ stats, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, GnLevels lv, int ppi, int C,
                                                             int relu, float* __restrict__ pg, float* __restrict__ pc) {
    __shared__ float red[256][18];
    const int g = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int slot = blockIdx.x, n = blockIdx.y, S = lv.slot0[lv.L];
    const int l = slot_level(lv, slot);
    const int cnt = lv.cnt[l];
    const int p0 = (slot - lv.slot0[l]) * GN_SLOT_PX;
    int p1 = p0 + GN_SLOT_PX;
    if (p1 > cnt) p1 = cnt;
    const long long base = ((long long)n * ppi + lv.off[l]) * C + g * 8;
    const float mean = stats[((n * lv.L + l) * 32 + g) * 2], rstd = stats[((n * lv.L + l) * 32 + g) * 2 + 1];
    float gm[8], bt[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { gm[k] = gamma[g * 8 + k]; bt[k] = beta[g * 8 + k]; }
    float A = 0.f, B = 0.f, dg[8], db[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { dg[k] = 0.f; db[k] = 0.f; }
    for (int p = p0 + pl; p < p1; p += 32) {          // four pixels per trip: eight independent loads in flight
        u32x4_t vd[4], vy[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = p + 8 * u;
            const long long o = base + (long long)(q < p1 ? q : p) * C;
            vd[u] = *reinterpret_cast<const u32x4_t*>(dz + o);
            vy[u] = *reinterpret_cast<const u32x4_t*>(y + o);
            if (q >= p1) vd[u] = (u32x4_t){0u, 0u, 0u, 0u};      // a zero gradient contributes nothing
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int w = k >> 1;
                float d = (k & 1) ? bf_hi(vd[u][w]) : bf_lo(vd[u][w]);
                const float yy = (k & 1) ? bf_hi(vy[u][w]) : bf_lo(vy[u][w]);
                const float xh = gn_xhat(yy, mean, rstd);
                if (relu && !(__fmaf_rn(xh, gm[k], bt[k]) > 0.f)) d = 0.f;
                A += d * gm[k] * xh; B += d * gm[k];
                dg[k] += d * xh; db[k] += d;
            }
    }
    red[threadIdx.x][0] = A; red[threadIdx.x][1] = B;
#pragma unroll
    for (int k = 0; k < 8; ++k) { red[threadIdx.x][2 + k] = dg[k]; red[threadIdx.x][10 + k] = db[k]; }
    __syncthreads();
    if (pl == 0) {
        float t[18];
#pragma unroll
        for (int q = 0; q < 18; ++q) t[q] = 0.f;
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int q = 0; q < 18; ++q) t[q] += red[k * 32 + g][q];
        const long long sl = (long long)n * S + slot;
        pg[(sl * 32 + g) * 2] = t[0]; pg[(sl * 32 + g) * 2 + 1] = t[1];
#pragma unroll
        for (int k = 0; k < 8; ++k) { pc[(sl * C + g * 8 + k) * 2] = t[2 + k]; pc[(sl * C + g * 8 + k) * 2 + 1] = t[10 + k]; }
    }
}

// ab[(n*L + l)*32 + g] = (A, B) / count: one workgroup per (image, level), as gn_stats_final_kernel
__device__ __forceinline__ void gn_bwd_final_body(const float* __restrict__ pg, const GnLevels& lv, int cpg, float* __restrict__ ab, int l, int n,
                                                  float (*red)[32][2]) {
    // (every thread of the block reaches the barrier; threads past the first 256 -- gn_bwd_finals_kernel launches 1024 -- carry nothing)
    const int g = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int S = lv.slot0[lv.L];
    float A = 0.f, B = 0.f;
    if (q < 8) {
        for (int c = lv.slot0[l] + q; c < lv.slot0[l + 1]; c += 8) {
            const float* o = pg + (((long long)n * S + c) * 32 + g) * 2;
            A += o[0]; B += o[1];
        }
        red[q][g][0] = A; red[q][g][1] = B;
    }
    __syncthreads();
    if (q == 0) {
        A = 0.f; B = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { A += red[k][g][0]; B += red[k][g][1]; }
        const float inv = 1.f / ((float)lv.cnt[l] * (float)cpg);
        const int i = (n * lv.L + l) * 32 + g;
        ab[i * 2] = A * inv; ab[i * 2 + 1] = B * inv;
    }
}
__global__ __launch_bounds__(256) void gn_bwd_final_kernel(const float* __restrict__ pg, GnLevels lv, int cpg, float* __restrict__ ab) {
    __shared__ float red[8][32][2];
    gn_bwd_final_body(pg, lv, cpg, ab, blockIdx.x, blockIdx.y, red);
}

// dgamma[c], dbeta[c] (+)= sums over every (n, slot): a block owns 8 channels, 128 lanes walk the slots (each reads the 64 contiguous bytes
// of its 8 channels), then a fixed-order LDS reduction in two steps -- reproducible, and no long serial chains (2 848 slots at batch 16)
__device__ __forceinline__ void gn_bwd_final_c_body(const float* __restrict__ pc, int slots, int C, float* __restrict__ dgamma,
                                                    float* __restrict__ dbeta, int accumulate, int block, float (*red)[8][2]) {
    const int cl = threadIdx.x & 7, sl = threadIdx.x >> 3;
    const int c = block * 8 + cl;
    float dg = 0.f, db = 0.f;
    if (c < C)
        for (int s = sl; s < slots; s += 128) { dg += pc[((long long)s * C + c) * 2]; db += pc[((long long)s * C + c) * 2 + 1]; }
    red[sl][cl][0] = dg; red[sl][cl][1] = db;
    __syncthreads();
    float a16 = 0.f, b16 = 0.f;
    if (sl < 8)                        // 16 partials each, in order
        for (int k = 0; k < 16; ++k) { a16 += red[sl * 16 + k][cl][0]; b16 += red[sl * 16 + k][cl][1]; }
    __syncthreads();
    if (sl < 8) { red[sl][cl][0] = a16; red[sl][cl][1] = b16; }
    __syncthreads();
    if (sl == 0 && c < C) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < 8; ++k) { a += red[k][cl][0]; b += red[k][cl][1]; }
        dgamma[c] = accumulate ? dgamma[c] + a : a;
        dbeta[c] = accumulate ? dbeta[c] + b : b;
    }
}
__global__ __launch_bounds__(1024) void gn_bwd_final_c_kernel(const float* __restrict__ pc, int slots, int C, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int accumulate) {
    __shared__ float red[128][8][2];
    gn_bwd_final_c_body(pc, slots, C, dgamma, dbeta, accumulate, blockIdx.x, red);
}
// Round 5: both finals in ONE launch when the batch is not chunked (blocks [0, L * N): the per-(image, level, group) sums the apply pass waits
// for; the rest: dgamma / dbeta) -- two small grids one after the other were 24 us per layer of an otherwise idle GPU on FCOS's main chain
__global__ __launch_bounds__(1024) void gn_bwd_finals_kernel(const float* __restrict__ pg, GnLevels lv, int cpg, float* __restrict__ ab, int N,
                                                             const float* __restrict__ pc, int slots, int C, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int accumulate) {
    __shared__ float red[128][8][2];
    const int nb = lv.L * N;
    if ((int)blockIdx.x < nb) {
        float (*r2)[32][2] = reinterpret_cast<float (*)[32][2]>(&red[0][0][0]);
        gn_bwd_final_body(pg, lv, cpg, ab, (int)blockIdx.x % lv.L, (int)blockIdx.x / lv.L, r2);        // (all 1024 threads: one barrier inside)
        return;
    }
    gn_bwd_final_c_body(pc, slots, C, dgamma, dbeta, accumulate, (int)blockIdx.x - nb, red);
}

// dy = rstd * (dzm*gamma - B - xhat*A)
__global__ void gn_bwd_apply_kernel(const bf16_raw* __restrict__ dz, const bf16_raw* __restrict__ y, const float* __restrict__ stats,
                                    const float* __restrict__ ab, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    GnLevels lv, int N, int ppi, int relu, int rev, bf16_raw* __restrict__ dy) {
    const long long total = (long long)N * ppi * 32;
    for (long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += (long long)gridDim.x * blockDim.x) {
        const long long i = rev ? total - 1 - i0 : i0;
        const int g = (int)(i & 31);
        const long long pix = i >> 5;
        const int n = (int)(pix / ppi), p = (int)(pix - (long long)n * ppi);
        const int l = level_of(lv, p);
        const int si = ((n * lv.L + l) * 32 + g) * 2;
        const float mean = stats[si], rstd = stats[si + 1], A = ab[si], B = ab[si + 1];
        const u32x4_t vd = *reinterpret_cast<const u32x4_t*>(dz + i * 8);
        const u32x4_t vy = *reinterpret_cast<const u32x4_t*>(y + i * 8);
        const f32x4_t g0 = *reinterpret_cast<const f32x4_t*>(gamma + g * 8), g1 = *reinterpret_cast<const f32x4_t*>(gamma + g * 8 + 4);
        const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(beta + g * 8), b1 = *reinterpret_cast<const f32x4_t*>(beta + g * 8 + 4);
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int w = k >> 1;
            float d = (k & 1) ? bf_hi(vd[w]) : bf_lo(vd[w]);
            const float yy = (k & 1) ? bf_hi(vy[w]) : bf_lo(vy[w]);
            const float gk = k < 4 ? g0[k & 3] : g1[k & 3], bk = k < 4 ? b0[k & 3] : b1[k & 3];
            const float xh = gn_xhat(yy, mean, rstd);
            if (relu && !(__fmaf_rn(xh, gk, bk) > 0.f)) d = 0.f;
            o[k] = rstd * (d * gk - B - xh * A);
        }
        u32x4_t w4;
#pragma unroll
        for (int k = 0; k < 4; ++k) w4[k] = pack_bf2(o[2 * k], o[2 * k + 1]);
        *reinterpret_cast<u32x4_t*>(dy + i * 8) = w4;
    }
}

// ---- FCOS offsets: off[p][k] = relu(raw[p][k] * scale_l) * stride_l (k < 4); raw rows have `ld` channels -------------
struct OffLevels { int L; int off[MAXL]; int cnt[MAXL]; float stride[MAXL]; };

__global__ void fcos_offsets_fwd_kernel(const bf16_raw* __restrict__ raw, int ld, const float* __restrict__ scales, OffLevels lv,
                                        int N, int ppi, bf16_raw* __restrict__ out) {
    const long long total = (long long)N * ppi;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(i % ppi);
        int l = 0;
#pragma unroll
        for (int k = 1; k < MAXL; ++k)
            if (k < lv.L && p >= lv.off[k]) l = k;
        const float sc = scales[l], st = lv.stride[l];
        const u32x2_t v = *reinterpret_cast<const u32x2_t*>(raw + i * ld);
        u32x2_t o;
        o[0] = pack_bf2(fmaxf(bf_lo(v[0]) * sc, 0.f) * st, fmaxf(bf_hi(v[0]) * sc, 0.f) * st);
        o[1] = pack_bf2(fmaxf(bf_lo(v[1]) * sc, 0.f) * st, fmaxf(bf_hi(v[1]) * sc, 0.f) * st);
        *reinterpret_cast<u32x2_t*>(out + i * 4) = o;
    }
}

// d_raw[p][k] = d_off[p][k] * stride * scale * (raw*scale > 0) (k < 4), d_raw[p][4] = d_ctr[p], rest 0;
// dscale partial per block: one fixed-order partial per (block, level)
__global__ __launch_bounds__(256) void fcos_offsets_bwd_kernel(const bf16_raw* __restrict__ raw, int ld, const float* __restrict__ scales,
                                                               OffLevels lv, int N, int ppi, const bf16_raw* __restrict__ d_off,
                                                               const bf16_raw* __restrict__ d_ctr, bf16_raw* __restrict__ d_raw,
                                                               float* __restrict__ dscale_partial) {
    __shared__ float red[4][MAXL];
    float ds[MAXL];
#pragma unroll
    for (int k = 0; k < MAXL; ++k) ds[k] = 0.f;
    const long long total = (long long)N * ppi;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(i % ppi);
        int l = 0;
#pragma unroll
        for (int k = 1; k < MAXL; ++k)
            if (k < lv.L && p >= lv.off[k]) l = k;
        const float sc = scales[l], st = lv.stride[l];
        const u32x2_t v = *reinterpret_cast<const u32x2_t*>(raw + i * ld);
        const u32x2_t g = *reinterpret_cast<const u32x2_t*>(d_off + i * 4);
        const float r[4] = {bf_lo(v[0]), bf_hi(v[0]), bf_lo(v[1]), bf_hi(v[1])};
        const float go[4] = {bf_lo(g[0]), bf_hi(g[0]), bf_lo(g[1]), bf_hi(g[1])};
        float dr[4], acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool on = r[k] * sc > 0.f;
            dr[k] = on ? go[k] * st * sc : 0.f;
            acc += on ? go[k] * st * r[k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < MAXL; ++k) if (k == l) ds[k] += acc;
        u32x4_t o = {pack_bf2(dr[0], dr[1]), pack_bf2(dr[2], dr[3]), (unsigned int)d_ctr[i], 0u};
        *reinterpret_cast<u32x4_t*>(d_raw + i * ld) = o;       // ld == 8
    }
#pragma unroll
    for (int k = 0; k < MAXL; ++k) ds[k] = wave_sum(ds[k]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < MAXL; ++k) red[threadIdx.x >> 6][k] = ds[k];
    __syncthreads();
    if (threadIdx.x < MAXL)
        dscale_partial[(long long)blockIdx.x * MAXL + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// one wave per level: lane q adds blocks q, q + 64, ... in order, then the 64 partial sums go through a fixed butterfly (round 5: one thread
// per level walking all 512 blocks was 51 us of an otherwise idle GPU on FCOS's main chain)
__global__ __launch_bounds__(64 * MAXL) void fcos_dscale_final_kernel(const float* __restrict__ partial, int nblocks, int L, float* __restrict__ dscale) {
    const int l = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float s = 0.f;
    if (l < L)
        for (int b = lane; b < nblocks; b += 64) s += partial[(long long)b * MAXL + l];
    s = wave_sum(s);
    if (l < L && lane == 0) dscale[l] = s;
}

inline GnLevels make_levels(int L, const int32_t* off, const int32_t* cnt) {
    GnLevels lv{};
    lv.L = L;
    int s = 0;
    for (int i = 0; i < L; ++i) { lv.off[i] = off[i]; lv.cnt[i] = cnt[i]; lv.slot0[i] = s; s += cdiv(cnt[i], GN_SLOT_PX); }
    for (int i = L; i <= MAXL; ++i) lv.slot0[i] = s;
    return lv;
}
inline int egrid(long long n) { long long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }
constexpr int OFF_BLOCKS = 512;

// (Rounds 4-5 could run the forward / backward in chunks of images -- bd_groupnorm_set_chunks: statistics of a chunk, then its apply pass
// re-reading the chunk from the Infinity Cache -- or walk the apply pass from the END: measured slower at every chunk size
// (profiles/r05_gn_chunks.txt); round 6 removed the switch.  The loops below run once, over the whole batch.)
constexpr int GN_CHUNK = 0;

}  // namespace

// slots per image <= pix_per_img / 128 + L
extern "C" size_t bd_groupnorm_workspace_bytes(int N, int L, int C, int64_t pix_per_img) {
    const size_t S = (size_t)(pix_per_img / GN_SLOT_PX) + (size_t)L + 1;
    return (size_t)N * S * (32 * 2 + (size_t)C * 2) * sizeof(float) + (size_t)N * L * 32 * 2 * sizeof(float);
}

extern "C" int bd_groupnorm_fwd(const void* y, const float* gamma, const float* beta, int N, int L, const int32_t* lvl_off_host,
                                const int32_t* lvl_cnt_host, int64_t pix_per_img, int C, float eps, int relu, float* stats,
                                void* z, void* ws, size_t ws_bytes, bd_stream_t stream) {
    BD_REQUIRE(y && gamma && beta && stats && z && ws && lvl_off_host && lvl_cnt_host, "groupnorm_fwd: null pointer");
    BD_REQUIRE(C == 256, "groupnorm_fwd: C=%d unsupported (32 groups x 8 channels only)", C);
    BD_REQUIRE(L >= 1 && L <= MAXL && N >= 1, "groupnorm_fwd: bad N/L");
    if (ws_bytes < bd_groupnorm_workspace_bytes(N, L, C, pix_per_img)) { bd_set_error("groupnorm_fwd: workspace too small"); return BD_EWORKSPACE; }
    const GnLevels lv = make_levels(L, lvl_off_host, lvl_cnt_host);
    const int S = lv.slot0[L];
    hipStream_t st = (hipStream_t)stream;
    const int step = GN_CHUNK > 0 ? GN_CHUNK : N;
    const int rev = 0;
    for (int n0 = 0; n0 < N; n0 += step) {
        const int nc = n0 + step <= N ? step : N - n0;
        const bf16_raw* yc = (const bf16_raw*)y + (long long)n0 * pix_per_img * C;
        bf16_raw* zc = (bf16_raw*)z + (long long)n0 * pix_per_img * C;
        float* part = (float*)ws + (size_t)n0 * S * 64;
        float* stc = stats + (size_t)n0 * L * 64;
        hipLaunchKernelGGL(gn_stats_partial_kernel, dim3(S, nc), dim3(256), 0, st, yc, lv, (int)pix_per_img, C, part);
        hipLaunchKernelGGL(gn_stats_final_kernel, dim3(L, nc), dim3(32 * GN_FQ), 0, st, (const float*)part, lv, C / 32, eps, stc);
        hipLaunchKernelGGL(gn_apply_kernel, dim3(egrid((long long)nc * pix_per_img * 32)), dim3(256), 0, st, yc, (const float*)stc, gamma, beta,
                           lv, nc, (int)pix_per_img, relu, rev, zc);
    }
    BD_CHECK_LAUNCH("bd_groupnorm_fwd");
    return BD_OK;
}

int bd_conv3x3_pp_patch_starts(const bd_conv_desc* d, int* starts);          // conv3x3_pp.hip

// GroupNorm forward from the per-patch statistics the producing convolution left (bd_conv2d_fwd_gnstats): the finalize kernel sums a level's
// patches exactly as it sums 128-pixel slots (8 lanes per group in patch order, then the eight partial sums in lane order: a fixed order),
// then the unchanged apply pass.  d = that convolution's descriptor (its output levels are the tensor's levels).
extern "C" int bd_groupnorm_fwd_parts(const bd_conv_desc* d, const void* y, const float* part, const float* gamma, const float* beta, float eps,
                                      int relu, float* stats, void* z, bd_stream_t stream) {
    BD_REQUIRE(d && y && part && gamma && beta && stats && z, "groupnorm_fwd_parts: null pointer");
    BD_REQUIRE(d->Cout == 256 && d->nseg >= 1 && d->nseg <= MAXL && d->N >= 1, "groupnorm_fwd_parts: 256 channels, 1 .. %d levels", MAXL);
    GnLevels lv{};
    lv.L = d->nseg;
    int starts[BD_MAX_SEGS + 1];
    const int S = bd_conv3x3_pp_patch_starts(d, starts);
    for (int i = 0; i < d->nseg; ++i) { lv.off[i] = d->out_off[i]; lv.cnt[i] = d->Ho[i] * d->Wo[i]; lv.slot0[i] = starts[i]; }
    for (int i = d->nseg; i <= MAXL; ++i) lv.slot0[i] = S;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_stats_final_kernel, dim3(d->nseg, d->N), dim3(32 * GN_FQ), 0, st, part, lv, 8, eps, stats);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(egrid((long long)d->N * d->out_pix_per_img * 32)), dim3(256), 0, st, (const bf16_raw*)y, (const float*)stats,
                       gamma, beta, lv, d->N, (int)d->out_pix_per_img, relu, 0, (bf16_raw*)z);
    BD_CHECK_LAUNCH("bd_groupnorm_fwd_parts");
    return BD_OK;
}

extern "C" int bd_groupnorm_bwd(const void* dz, const void* y, const float* gamma, const float* beta, const float* stats, int N, int L,
                                const int32_t* lvl_off_host, const int32_t* lvl_cnt_host, int64_t pix_per_img, int C, int relu,
                                void* dy, float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes, bd_stream_t stream) {
    BD_REQUIRE(dz && y && gamma && beta && stats && dy && dgamma && dbeta && ws, "groupnorm_bwd: null pointer");
    BD_REQUIRE(C == 256, "groupnorm_bwd: C=%d unsupported", C);
    BD_REQUIRE(L >= 1 && L <= MAXL && N >= 1, "groupnorm_bwd: bad N/L");
    if (ws_bytes < bd_groupnorm_workspace_bytes(N, L, C, pix_per_img)) { bd_set_error("groupnorm_bwd: workspace too small"); return BD_EWORKSPACE; }
    const GnLevels lv = make_levels(L, lvl_off_host, lvl_cnt_host);
    const int S = lv.slot0[L];
    hipStream_t st = (hipStream_t)stream;
    float* pg = (float*)ws;
    float* pc = pg + (size_t)N * S * 32 * 2;
    float* ab = pc + (size_t)N * S * C * 2;
    const int step = GN_CHUNK > 0 ? GN_CHUNK : N;
    const int rev = 0;
    for (int n0 = 0; n0 < N; n0 += step) {
        const int nc = n0 + step <= N ? step : N - n0;
        const long long eo = (long long)n0 * pix_per_img * C;
        const bf16_raw* dzc = (const bf16_raw*)dz + eo;
        const bf16_raw* yc = (const bf16_raw*)y + eo;
        const float* stc = stats + (size_t)n0 * L * 64;
        float* abc = ab + (size_t)n0 * L * 64;
        hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(S, nc), dim3(256), 0, st, dzc, yc, stc, gamma, beta, lv, (int)pix_per_img, C, relu,
                           pg + (size_t)n0 * S * 64, pc + (size_t)n0 * S * C * 2);
        if (nc == N)
            hipLaunchKernelGGL(gn_bwd_finals_kernel, dim3(L * N + cdiv(C, 8)), dim3(1024), 0, st, (const float*)pg, lv, C / 32, abc, N, (const float*)pc,
                               N * S, C, dgamma, dbeta, accumulate);
        else
            hipLaunchKernelGGL(gn_bwd_final_kernel, dim3(L, nc), dim3(256), 0, st, (const float*)(pg + (size_t)n0 * S * 64), lv, C / 32, abc);
        hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(egrid((long long)nc * pix_per_img * 32)), dim3(256), 0, st, dzc, yc, stc, (const float*)abc,
                           gamma, beta, lv, nc, (int)pix_per_img, relu, rev, (bf16_raw*)dy + eo);
    }
    // dgamma / dbeta over every (image, slot) of the batch, in slot order: independent of the chunking
    if (step < N) hipLaunchKernelGGL(gn_bwd_final_c_kernel, dim3(cdiv(C, 8)), dim3(1024), 0, st, (const float*)pc, N * S, C, dgamma, dbeta, accumulate);
    BD_CHECK_LAUNCH("bd_groupnorm_bwd");
    return BD_OK;
}

extern "C" int bd_fcos_offsets_fwd(const void* raw, int ld, const float* scales, int N, int L, const int32_t* lvl_off_host,
                                   const int32_t* lvl_cnt_host, const int32_t* strides_host, int64_t pix_per_img, void* out,
                                   bd_stream_t stream) {
    BD_REQUIRE(raw && scales && out && lvl_off_host && lvl_cnt_host && strides_host, "fcos_offsets_fwd: null pointer");
    BD_REQUIRE(ld == 8 && L >= 1 && L <= MAXL, "fcos_offsets_fwd: ld must be 8");
    OffLevels lv{};
    lv.L = L;
    for (int i = 0; i < L; ++i) { lv.off[i] = lvl_off_host[i]; lv.cnt[i] = lvl_cnt_host[i]; lv.stride[i] = (float)strides_host[i]; }
    hipLaunchKernelGGL(fcos_offsets_fwd_kernel, dim3(egrid((long long)N * pix_per_img)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_raw*)raw, ld, scales, lv, N, (int)pix_per_img, (bf16_raw*)out);
    BD_CHECK_LAUNCH("bd_fcos_offsets_fwd");
    return BD_OK;
}

extern "C" size_t bd_fcos_offsets_workspace_bytes(void) { return (size_t)OFF_BLOCKS * MAXL * sizeof(float); }

extern "C" int bd_fcos_offsets_bwd(const void* raw, int ld, const float* scales, int N, int L, const int32_t* lvl_off_host,
                                   const int32_t* lvl_cnt_host, const int32_t* strides_host, int64_t pix_per_img, const void* d_off,
                                   const void* d_ctr, void* d_raw, float* dscale, void* ws, size_t ws_bytes, bd_stream_t stream) {
    BD_REQUIRE(raw && scales && d_off && d_ctr && d_raw && dscale && ws, "fcos_offsets_bwd: null pointer");
    BD_REQUIRE(ld == 8 && L >= 1 && L <= MAXL, "fcos_offsets_bwd: ld must be 8");
    if (ws_bytes < bd_fcos_offsets_workspace_bytes()) { bd_set_error("fcos_offsets_bwd: workspace too small"); return BD_EWORKSPACE; }
    OffLevels lv{};
    lv.L = L;
    for (int i = 0; i < L; ++i) { lv.off[i] = lvl_off_host[i]; lv.cnt[i] = lvl_cnt_host[i]; lv.stride[i] = (float)strides_host[i]; }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(fcos_offsets_bwd_kernel, dim3(OFF_BLOCKS), dim3(256), 0, st, (const bf16_raw*)raw, ld, scales, lv, N,
                       (int)pix_per_img, (const bf16_raw*)d_off, (const bf16_raw*)d_ctr, (bf16_raw*)d_raw, (float*)ws);
    hipLaunchKernelGGL(fcos_dscale_final_kernel, dim3(1), dim3(64 * MAXL), 0, st, (const float*)ws, OFF_BLOCKS, L, dscale);
    BD_CHECK_LAUNCH("bd_fcos_offsets_bwd");
    return BD_OK;
}
