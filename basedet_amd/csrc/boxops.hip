// Box operators of basedet.layers / basedet.structures as wavefront-primitive HIP kernels.
// This translation unit is compiled with -ffp-contract=off (no FMA contraction) and HIP's default IEEE-correct
// fp32 division / sqrt, so that every decision taken on an IoU value (thresholds, equality with the per-gt row
// maximum, NMS suppression) is bit-identical to the float32 numpy oracle (oracle/box_ops.py).
#pragma clang fp contract(off)
#include "box_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------------------
// anchors  (layers/common/anchor_generator.py:23-30, 111-122, 152-165)
// ------------------------------------------------------------------------------------------------------------
__global__ void anchors_kernel(int H, int W, int stride, float shift, const float* __restrict__ base, int A,
                               float* __restrict__ out) {
    const long long total = (long long)H * W * A;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int a = (int)(i % A);
    const long long pix = i / A;
    const int x = (int)(pix % W), y = (int)(pix / W);
    const float gx = (float)x * (float)stride + shift;
    const float gy = (float)y * (float)stride + shift;
    f32x4_t o;
    o[0] = gx + base[a * 4 + 0]; o[1] = gy + base[a * 4 + 1];
    o[2] = gx + base[a * 4 + 2]; o[3] = gy + base[a * 4 + 3];
    *reinterpret_cast<f32x4_t*>(out + i * 4) = o;
}

__global__ void points_kernel(int H, int W, int stride, float shift, int A, float* __restrict__ out) {
    const long long total = (long long)H * W * A;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long pix = i / A;
    const int x = (int)(pix % W), y = (int)(pix / W);
    out[i * 2 + 0] = (float)x * (float)stride + shift;
    out[i * 2 + 1] = (float)y * (float)stride + shift;
}

// ------------------------------------------------------------------------------------------------------------
// pairwise ops (structures/op_patch.py:33-97, 170-227; structures/boxes.py:74-95, 114-130)
// ------------------------------------------------------------------------------------------------------------
__global__ void pairwise_kernel(const float* __restrict__ b1, int m, const float* __restrict__ b2, int n, int mode,
                                float* __restrict__ out) {
    const long long total = (long long)m * n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / n), c = (int)(i % n);
        const Box a = ld_box(b1 + r * 4ll), b = ld_box(b2 + c * 4ll);
        const float inter = box_inter(a, b);
        float v;
        if (mode == 2) v = inter;
        else if (mode == 1) v = fmaxf(inter / box_area(b), 0.f);
        else {
            const float uni = (box_area(a) + box_area(b)) - inter;
            if (mode == 0) v = fmaxf(inter / uni, 0.f);
            else {
                const float iou = inter / uni;
                const float w = fmaxf(fmaxf(a.x2, b.x2) - fminf(a.x1, b.x1), 0.f);
                const float h = fmaxf(fmaxf(a.y2, b.y2) - fminf(a.y1, b.y1), 0.f);
                const float hull = w * h;
                v = iou - (hull - uni) / hull;
            }
        }
        out[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------
// BoxCoder (structures/boxcoder.py:61-98)
// ------------------------------------------------------------------------------------------------------------
__global__ void encode_kernel(const float* __restrict__ anchors, const float* __restrict__ gt, long long n, Coder c,
                              float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    *reinterpret_cast<f32x4_t*>(out + i * 4) = encode_dev(ld_box(anchors + i * 4), ld_box(gt + i * 4), c);
}

__global__ void decode_kernel(const float* __restrict__ anchors, const float* __restrict__ deltas, long long n, Coder c,
                              float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    *reinterpret_cast<f32x4_t*>(out + i * 4) = decode_dev(ld_box(anchors + i * 4), *reinterpret_cast<const f32x4_t*>(deltas + i * 4), c);
}

// ------------------------------------------------------------------------------------------------------------
// RetinaNet target assignment (models/det/retinanet.py:211-232 + layers/common/matcher.py:31-51)
// ------------------------------------------------------------------------------------------------------------
constexpr int ASSIGN_APT = 4;   // anchors per thread in the row-max pass

// pass 1: per-gt maximum IoU over all anchors (matcher.py:48 F.max(matrix, axis=1)); IoU >= 0 so the uint
// ordering of the float bits is the float ordering -> atomicMax on bits.
__global__ __launch_bounds__(256) void gt_rowmax_kernel(const float* __restrict__ anchors, int A,
                                                        const float* __restrict__ gt_boxes, const int* __restrict__ num_gt,
                                                        int Gmax, unsigned int* __restrict__ gtmax) {
    __shared__ float red[4];
    const int n = blockIdx.y;
    const int G = min(num_gt[n], Gmax);
    Box ab[ASSIGN_APT]; float aa[ASSIGN_APT]; bool av[ASSIGN_APT];
#pragma unroll
    for (int k = 0; k < ASSIGN_APT; ++k) {
        const int a = (blockIdx.x * ASSIGN_APT + k) * 256 + threadIdx.x;
        av[k] = a < A;
        ab[k] = ld_box(anchors + (av[k] ? a : 0) * 4ll);
        aa[k] = box_area(ab[k]);
    }
    for (int g = 0; g < G; ++g) {
        const float* gpp = gt_boxes + ((long long)n * Gmax + g) * 5;    // 20-byte rows: scalar loads
        const Box gb = Box{gpp[0], gpp[1], gpp[2], gpp[3]};
        float m = 0.f;
        const float ga = box_area(gb);
#pragma unroll
        for (int k = 0; k < ASSIGN_APT; ++k)
            if (av[k]) m = fmaxf(m, box_iou_dev(gb, ga, ab[k], aa[k]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float mm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            atomicMax(gtmax + n * Gmax + g, __float_as_uint(mm));
        }
        __syncthreads();
    }
}

// pass 2: per-anchor max/argmax over gts, threshold bands, low-quality rule, class label, encode
__global__ __launch_bounds__(256) void retina_assign_kernel(const float* __restrict__ anchors, int A,
                                                            const float* __restrict__ gt_boxes, const int* __restrict__ num_gt,
                                                            int Gmax, float thr_lo, float thr_hi, int allow_lq, Coder coder,
                                                            const unsigned int* __restrict__ gtmax, int* __restrict__ labels,
                                                            int* __restrict__ match_idx, float* __restrict__ offsets,
                                                            int* __restrict__ num_fg, int class_agnostic) {
    const int n = blockIdx.y;
    const int a = blockIdx.x * 256 + threadIdx.x;
    const int G = min(num_gt[n], Gmax);
    int fg = 0;
    if (a < A) {
        const Box ab = ld_box(anchors + a * 4ll);
        const float aa = box_area(ab);
        float best = -1.f; int bi = 0; bool lq = false;
        const float* gp = gt_boxes + (long long)n * Gmax * 5;
        for (int g = 0; g < G; ++g) {
            const Box gb = ld_gt(gp + g * 5);
            const float iou = box_iou_dev(gb, box_area(gb), ab, aa);
            if (iou > best) { best = iou; bi = g; }          // first maximum wins (lowest gt index)
            if (iou == __uint_as_float(gtmax[n * Gmax + g])) lq = true;
        }
        int lab;
        f32x4_t off = {0.f, 0.f, 0.f, 0.f};
        if (G == 0) { lab = 0; bi = 0; }
        else {
            // matcher.py:43-45 half-open bands: (-inf, lo) -> 0, [lo, hi) -> -1, [hi, inf) -> 1
            lab = best < thr_lo ? 0 : (best < thr_hi ? -1 : 1);
            if (allow_lq && lq) lab = 1;
            const float* mg = gp + bi * 5;
            if (lab == 1 && !class_agnostic) lab = (int)mg[4];
            off = encode_dev(ab, ld_gt(mg), coder);
        }
        const long long o = (long long)n * A + a;
        labels[o] = lab;
        match_idx[o] = bi;
        *reinterpret_cast<f32x4_t*>(offsets + o * 4) = off;
        fg = lab > 0;
    }
    const unsigned long long bal = __ballot(fg);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(num_fg, __popcll(bal));
}

// ------------------------------------------------------------------------------------------------------------
// FCOS target assignment (models/det/fcos.py:222-293)
// ------------------------------------------------------------------------------------------------------------
struct FcosLevels { int start[BD_MAX_SEGS + 1]; float lo[BD_MAX_SEGS], hi[BD_MAX_SEGS]; float radius[BD_MAX_SEGS]; int L; };

__global__ __launch_bounds__(256) void fcos_assign_kernel(const float* __restrict__ points, int P, FcosLevels lv,
                                                          int use_center, const float* __restrict__ gt_boxes,
                                                          const int* __restrict__ num_gt, int Gmax, int* __restrict__ labels,
                                                          float* __restrict__ offsets, float* __restrict__ ctrness,
                                                          float* __restrict__ stats) {
    const int n = blockIdx.y;
    const int pidx = blockIdx.x * 256 + threadIdx.x;
    const int G = min(num_gt[n], Gmax);
    float fgf = 0.f, ctr_fg = 0.f;
    if (pidx < P) {
        int l = 0;
        for (int k = 1; k < lv.L; ++k) if (pidx >= lv.start[k]) l = k;
        const float px = points[pidx * 2ll], py = points[pidx * 2ll + 1];
        const float lo = lv.lo[l], hi = lv.hi[l], rad = lv.radius[l];
        float best = INFINITY; int bi = 0;
        const float* gp = gt_boxes + (long long)n * Gmax * 5;
        for (int g = 0; g < G; ++g) {
            const Box b = ld_gt(gp + g * 5);
            const float ol = px - b.x1, ot = py - b.y1, orr = b.x2 - px, ob = b.y2 - py;
            const float mx = fmaxf(fmaxf(ol, ot), fmaxf(orr, ob));
            const bool cared = mx >= lo && mx <= hi;                            // fcos.py:244-247
            bool inb;
            if (use_center) {                                                   // fcos.py:249-262
                const float cx = (b.x1 + b.x2) / 2.f, cy = (b.y1 + b.y2) / 2.f;
                const float c1x = fmaxf(cx - rad, b.x1), c1y = fmaxf(cy - rad, b.y1);
                const float c2x = fminf(cx + rad, b.x2), c2y = fminf(cy + rad, b.y2);
                inb = fminf(fminf(px - c1x, py - c1y), fminf(c2x - px, c2y - py)) > 0.f;
            } else {
                inb = fminf(fminf(ol, ot), fminf(orr, ob)) > 0.f;
            }
            const float area = (cared && inb) ? box_area(b) : INFINITY;
            if (area < best) { best = area; bi = g; }                           // first minimum wins
        }
        int lab = 0;
        f32x4_t off = {0.f, 0.f, 0.f, 0.f};
        float ctr = 0.f;
        if (G > 0) {
            const float* mg = gp + bi * 5;
            lab = isinf(best) ? 0 : (int)mg[4];                                 // fcos.py:274-275
            off[0] = px - mg[0]; off[1] = py - mg[1]; off[2] = mg[2] - px; off[3] = mg[3] - py;
            const float lr = fmaxf(fminf(off[0], off[2]) / fmaxf(off[0], off[2]), 0.f);
            const float tb = fmaxf(fminf(off[1], off[3]) / fmaxf(off[1], off[3]), 0.f);
            ctr = sqrtf(lr * tb);                                               // fcos.py:280-283
        }
        const long long o = (long long)n * P + pidx;
        labels[o] = lab;
        *reinterpret_cast<f32x4_t*>(offsets + o * 4) = off;
        ctrness[o] = ctr;
        if (lab > 0) { fgf = 1.f; ctr_fg = ctr; }
    }
    fgf = wave_sum(fgf);
    (void)ctr_fg;
    // num_fg: a sum of ones, exact in fp32 whatever the order of the atomics.  The sum of centre-ness (the regression loss's normaliser,
    // fcos.py:139-144) is left to ctr_sum_kernel below: float atomics would make it -- and with it every gradient of the step -- differ in
    // the last bits from run to run.
    if ((threadIdx.x & 63) == 0 && fgf > 0.f) atomicAdd(stats, fgf);
}

// stats[1] = sum of ctrness over the foreground points, in a FIXED order (one workgroup: strided per-thread partial sums in index order,
// then a tree): bitwise reproducible.  N * P is a few 100 K elements.  Round 5: sixteen elements per thread and trip as four independent
// 16-byte load pairs (the scalar form -- 350 dependent label -> value trips per thread -- took 112 us with nothing else on the GPU: FCOS's
// regression loss waits for this sum); then the remainder one by one.
__global__ __launch_bounds__(1024) void ctr_sum_kernel(const int* __restrict__ labels, const float* __restrict__ ctrness, long long n,
                                                       float* __restrict__ stats) {
    __shared__ float red[1024];
    float s = 0.f;
    typedef __attribute__((ext_vector_type(4))) int i32x4_l;
    const bool vec = (((size_t)labels | (size_t)ctrness) & 15) == 0;
    const long long n16 = vec ? n / 16384 * 16384 : 0;          // whole trips of 1024 threads x 16 elements
    for (long long i0 = 0; i0 < n16; i0 += 16384) {
        i32x4_l lb[4]; f32x4_t cv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long i = i0 + (long long)u * 4096 + threadIdx.x * 4;
            lb[u] = *reinterpret_cast<const i32x4_l*>(labels + i);
            cv[u] = *reinterpret_cast<const f32x4_t*>(ctrness + i);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) s += lb[u][k] > 0 ? cv[u][k] : 0.f;
    }
    for (long long i = n16 + threadIdx.x; i < n; i += 1024)
        if (labels[i] > 0) s += ctrness[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 512; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) stats[1] = red[0];
}

// ------------------------------------------------------------------------------------------------------------
// ATSS target assignment (models/det/atss.py:17-86)
// ------------------------------------------------------------------------------------------------------------
constexpr int ATSS_TOPK_MAX = 16;

// one workgroup per (gt, image): per level the TOPK points closest to the gt centre (ties: lower index), their IoUs with the
// point-centred anchors, threshold = mean + std over all candidates, then an atomicMax per surviving candidate on
// best[n][p] = (iou bits << 32) | ~g  (highest IoU wins, ties -> lowest gt index = argmax over the gt axis)
__global__ __launch_bounds__(256) void atss_candidates_kernel(const float* __restrict__ points, FcosLevels lv, int topk, float half_scale,
                                                              const float* __restrict__ gt_boxes, const int* __restrict__ num_gt,
                                                              int Gmax, int P, unsigned long long* __restrict__ best) {
    __shared__ unsigned long long red[4];
    __shared__ int cand[BD_MAX_SEGS * ATSS_TOPK_MAX];
    __shared__ float ciou[BD_MAX_SEGS * ATSS_TOPK_MAX];
    __shared__ float s_thr;
    const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    if (g >= min(num_gt[n], Gmax)) return;
    const Box gb = ld_gt(gt_boxes + ((long long)n * Gmax + g) * 5);
    const float cx = (gb.x1 + gb.x2) / 2.f, cy = (gb.y1 + gb.y2) / 2.f;     // box_center (op_patch.py:101-112)
    int ncand = 0;
    for (int l = 0; l < lv.L; ++l) {
        const int s0 = lv.start[l], cnt = lv.start[l + 1] - s0;
        const int kk = min(topk, cnt);
        unsigned long long last = 0ull;
        bool have_last = false;
        for (int k = 0; k < kk; ++k) {
            unsigned long long mine = ~0ull;
            for (int i = tid; i < cnt; i += 256) {
                const float dx = cx - points[(s0 + i) * 2ll], dy = cy - points[(s0 + i) * 2ll + 1];
                const float d = sqrtf(dx * dx + dy * dy);                        // atss.py:40-42
                const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)i;
                if ((!have_last || key > last) && key < mine) mine = key;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned long long t = __shfl_xor(mine, o, 64);
                mine = t < mine ? t : mine;
            }
            __syncthreads();
            if ((tid & 63) == 0) red[tid >> 6] = mine;
            __syncthreads();
            unsigned long long m = red[0];
#pragma unroll
            for (int w = 1; w < 4; ++w) m = red[w] < m ? red[w] : m;
            last = m; have_last = true;
            if (tid == 0) cand[ncand + k] = s0 + (int)(m & 0xffffffffull);
        }
        ncand += kk;
    }
    __syncthreads();
    if (tid < ncand) {
        const int p = cand[tid];
        int l = 0;
        for (int q = 1; q < lv.L; ++q) if (p >= lv.start[q]) l = q;
        const float hs = lv.radius[l] * half_scale;                               // stride * SCALE / 2 (atss.py:33-36)
        const float px = points[p * 2ll], py = points[p * 2ll + 1];
        const Box ab = Box{px - hs, py - hs, px + hs, py + hs};
        ciou[tid] = box_iou_dev(gb, box_area(gb), ab, box_area(ab));
    }
    __syncthreads();
    if (tid == 0) {                                                               // mean + std (population), index order
        float sum = 0.f;
        for (int i = 0; i < ncand; ++i) sum += ciou[i];
        const float mean = sum / (float)ncand;
        float var = 0.f;
        for (int i = 0; i < ncand; ++i) { const float d = ciou[i] - mean; var += d * d; }
        s_thr = mean + sqrtf(var / (float)ncand);
    }
    __syncthreads();
    if (tid < ncand) {
        const int p = cand[tid];
        const float px = points[p * 2ll], py = points[p * 2ll + 1];
        const bool inb = fminf(fminf(px - gb.x1, py - gb.y1), fminf(gb.x2 - px, gb.y2 - py)) > 0.f;     // atss.py:56-58
        if (ciou[tid] >= s_thr && inb)
            atomicMax(best + (long long)n * P + p, ((unsigned long long)__float_as_uint(ciou[tid]) << 32) | (0xffffffffu - (unsigned int)g));
    }
}

__global__ __launch_bounds__(256) void atss_finalize_kernel(const float* __restrict__ points, int P, const float* __restrict__ gt_boxes,
                                                            const int* __restrict__ num_gt, int Gmax,
                                                            const unsigned long long* __restrict__ best, int* __restrict__ labels,
                                                            float* __restrict__ offsets, float* __restrict__ ctrness,
                                                            float* __restrict__ stats) {
    const int n = blockIdx.y;
    const int pidx = blockIdx.x * 256 + threadIdx.x;
    const int G = min(num_gt[n], Gmax);
    float fgf = 0.f, ctr_fg = 0.f;
    if (pidx < P) {
        const float px = points[pidx * 2ll], py = points[pidx * 2ll + 1];
        const unsigned long long key = best[(long long)n * P + pidx];
        int lab = 0;
        f32x4_t off = {0.f, 0.f, 0.f, 0.f};
        float ctr = 0.f;
        if (G > 0) {
            const int bi = key ? (int)(0xffffffffu - (unsigned int)(key & 0xffffffffull)) : 0;     // unmatched: argmax of all -1 = 0
            const float* mg = gt_boxes + ((long long)n * Gmax + bi) * 5;
            lab = key ? (int)mg[4] : 0;
            off[0] = px - mg[0]; off[1] = py - mg[1]; off[2] = mg[2] - px; off[3] = mg[3] - py;
            const float lr = fmaxf(fminf(off[0], off[2]) / fmaxf(off[0], off[2]), 0.f);
            const float tb = fmaxf(fminf(off[1], off[3]) / fmaxf(off[1], off[3]), 0.f);
            ctr = sqrtf(lr * tb);                                                 // atss.py:70-75
        }
        const long long o = (long long)n * P + pidx;
        labels[o] = lab;
        *reinterpret_cast<f32x4_t*>(offsets + o * 4) = off;
        ctrness[o] = ctr;
        if (lab > 0) { fgf = 1.f; ctr_fg = ctr; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) fgf += __shfl_xor(fgf, o, 64);
    (void)ctr_fg;
    if ((threadIdx.x & 63) == 0 && fgf > 0.f) atomicAdd(stats, fgf);          // (the centre-ness sum: ctr_sum_kernel, fixed order)
}

// ------------------------------------------------------------------------------------------------------------
// batched NMS (layers/common/post_processing.py:17-47)
// ------------------------------------------------------------------------------------------------------------
constexpr int NMS_MAX = 16384;

// one workgroup: max coordinate, class-offset boxes, bitonic sort of (score desc, index asc)
__global__ __launch_bounds__(1024) void nms_prepare_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                           const int* __restrict__ idxs, int n, int npow2,
                                                           float* __restrict__ sboxes, int* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    __shared__ float red[16];
    const int tid = threadIdx.x;
    float mx = -INFINITY;
    for (int i = tid; i < n * 4; i += 1024) mx = fmaxf(mx, boxes[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = red[0];
    for (int k = 1; k < 16; ++k) mx = fmaxf(mx, red[k]);
    const float step = mx + 1.f;                                   // post_processing.py:44-45
    for (int i = tid; i < npow2; i += 1024) {
        unsigned long long key = ~0ull;
        if (i < n) {
            key = ((unsigned long long)float_desc_key(scores[i]) << 32) | (unsigned int)i;
            const float off = idxs ? (float)idxs[i] * step : 0.f;
            const Box b = ld_box(boxes + i * 4ll);
            f32x4_t o = {b.x1 + off, b.y1 + off, b.x2 + off, b.y2 + off};
            *reinterpret_cast<f32x4_t*>(sboxes + i * 4ll) = o;
        }
        keys[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= npow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npow2; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], b = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < n; i += 1024) order[i] = (int)(keys[i] & 0xffffffffu);
}

// mask[i][w] bit b: sorted box (w*64+b) is suppressed by sorted box i  (iou > thr, j > i)
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ sboxes, const int* __restrict__ order, int n,
                                                      float thr, unsigned long long* __restrict__ mask, int words) {
    const int i = blockIdx.x;          // sorted row
    const int w = blockIdx.y;          // column word
    const int j = w * 64 + threadIdx.x;
    bool sup = false;
    if (j < n && j > i) {
        const Box a = ld_box(sboxes + order[i] * 4ll), b = ld_box(sboxes + order[j] * 4ll);
        const float inter = box_inter(a, b);
        const float uni = (box_area(a) + box_area(b)) - inter;
        sup = (inter / uni) > thr;     // keep iff iou <= thr (py_cpu_nms, post_processing.py:130)
    }
    const unsigned long long bal = __ballot(sup);
    if (threadIdx.x == 0) mask[(long long)i * words + w] = bal;
}

// one wave walks the sorted list; removed[] lives in LDS (words <= 256)
__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* __restrict__ mask, const int* __restrict__ order,
                                                      int n, int words, int max_output, int* __restrict__ keep,
                                                      int* __restrict__ num_keep) {
    __shared__ unsigned long long removed[NMS_MAX / 64];
    const int lane = threadIdx.x;
    for (int w = lane; w < words; w += 64) removed[w] = 0ull;
    __syncthreads();
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        const unsigned long long r = removed[i >> 6];
        if ((r >> (i & 63)) & 1ull) continue;   // wave-uniform
        if (lane == 0) keep[cnt] = order[i];
        ++cnt;
        if (max_output > 0 && cnt >= max_output) break;
        for (int w = lane; w < words; w += 64) removed[w] |= mask[(long long)i * words + w];
        __syncthreads();
    }
    if (lane == 0) *num_keep = cnt;
}

}  // namespace

extern "C" int bd_anchors_generate(int H, int W, int stride, float offset, const float* base, int A, float* out,
                                   bd_stream_t stream) {
    BD_REQUIRE(base && out && H > 0 && W > 0 && A > 0, "anchors_generate: bad arguments");
    const long long total = (long long)H * W * A;
    hipLaunchKernelGGL(anchors_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, H, W, stride,
                       offset * (float)stride, base, A, out);
    BD_CHECK_LAUNCH("bd_anchors_generate");
    return BD_OK;
}

extern "C" int bd_points_generate(int H, int W, int stride, float offset, int A, float* out, bd_stream_t stream) {
    BD_REQUIRE(out && H > 0 && W > 0 && A > 0, "points_generate: bad arguments");
    const long long total = (long long)H * W * A;
    hipLaunchKernelGGL(points_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, H, W, stride,
                       offset * (float)stride, A, out);
    BD_CHECK_LAUNCH("bd_points_generate");
    return BD_OK;
}

extern "C" int bd_box_pairwise(const float* b1, int m, const float* b2, int n, int mode, float* out, bd_stream_t stream) {
    BD_REQUIRE(m >= 0 && n >= 0 && mode >= 0 && mode <= 3, "box_pairwise: bad arguments");
    if (m == 0 || n == 0) return BD_OK;
    BD_REQUIRE(b1 && b2 && out, "box_pairwise: null pointer");
    long long g = cdiv64((long long)m * n, 256);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(pairwise_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, b1, m, b2, n, mode, out);
    BD_CHECK_LAUNCH("bd_box_pairwise");
    return BD_OK;
}

extern "C" int bd_box_encode(const float* anchors, const float* gt, int64_t n, const float* mean4, const float* std4,
                             float* out, bd_stream_t stream) {
    if (n == 0) return BD_OK;
    BD_REQUIRE(anchors && gt && out && n > 0, "box_encode: bad arguments");
    hipLaunchKernelGGL(encode_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, anchors, gt,
                       (long long)n, make_coder(mean4, std4), out);
    BD_CHECK_LAUNCH("bd_box_encode");
    return BD_OK;
}

extern "C" int bd_box_decode(const float* anchors, const float* deltas, int64_t n, const float* mean4, const float* std4,
                             float* out, bd_stream_t stream) {
    if (n == 0) return BD_OK;
    BD_REQUIRE(anchors && deltas && out && n > 0, "box_decode: bad arguments");
    hipLaunchKernelGGL(decode_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, anchors, deltas,
                       (long long)n, make_coder(mean4, std4), out);
    BD_CHECK_LAUNCH("bd_box_decode");
    return BD_OK;
}

static int assign_encode_impl(const char* who, const float* anchors, int A, const float* gt_boxes, const int32_t* num_gt, int N,
                              int Gmax, float thr_lo, float thr_hi, int allow_low_quality, const float* mean4, const float* std4,
                              int32_t* labels, int32_t* match_idx, float* offsets, int32_t* num_fg, void* ws, size_t ws_bytes,
                              int class_agnostic, bd_stream_t stream) {
    BD_REQUIRE(anchors && gt_boxes && num_gt && labels && match_idx && offsets && num_fg && ws, "%s: null pointer", who);
    BD_REQUIRE(A > 0 && N > 0 && Gmax > 0, "%s: bad sizes", who);
    if (ws_bytes < (size_t)N * Gmax * sizeof(float)) {
        bd_set_error("%s: workspace %zu < %zu bytes", who, ws_bytes, (size_t)N * Gmax * sizeof(float));
        return BD_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(ws, 0, (size_t)N * Gmax * sizeof(float), st);
    (void)hipMemsetAsync(num_fg, 0, sizeof(int32_t), st);
    hipLaunchKernelGGL(gt_rowmax_kernel, dim3(cdiv(A, 256 * ASSIGN_APT), N), dim3(256), 0, st, anchors, A, gt_boxes, num_gt,
                       Gmax, (unsigned int*)ws);
    hipLaunchKernelGGL(retina_assign_kernel, dim3(cdiv(A, 256), N), dim3(256), 0, st, anchors, A, gt_boxes, num_gt, Gmax,
                       thr_lo, thr_hi, allow_low_quality, make_coder(mean4, std4), (const unsigned int*)ws, labels,
                       match_idx, offsets, num_fg, class_agnostic);
    BD_CHECK_LAUNCH(who);
    return BD_OK;
}

extern "C" int bd_retina_assign_encode(const float* anchors, int A, const float* gt_boxes, const int32_t* num_gt, int N,
                                       int Gmax, float thr_lo, float thr_hi, int allow_low_quality, const float* mean4,
                                       const float* std4, int32_t* labels, int32_t* match_idx, float* offsets,
                                       int32_t* num_fg, void* ws, size_t ws_bytes, bd_stream_t stream) {
    return assign_encode_impl("bd_retina_assign_encode", anchors, A, gt_boxes, num_gt, N, Gmax, thr_lo, thr_hi, allow_low_quality,
                              mean4, std4, labels, match_idx, offsets, num_fg, ws, ws_bytes, 0, stream);
}

extern "C" int bd_rpn_assign_encode(const float* anchors, int A, const float* gt_boxes, const int32_t* num_gt, int N,
                                    int Gmax, float thr_lo, float thr_hi, int allow_low_quality, const float* mean4,
                                    const float* std4, int32_t* labels, int32_t* match_idx, float* offsets,
                                    int32_t* num_fg, void* ws, size_t ws_bytes, bd_stream_t stream) {
    return assign_encode_impl("bd_rpn_assign_encode", anchors, A, gt_boxes, num_gt, N, Gmax, thr_lo, thr_hi, allow_low_quality,
                              mean4, std4, labels, match_idx, offsets, num_fg, ws, ws_bytes, 1, stream);
}

extern "C" int bd_fcos_assign(const float* points, int P, const int32_t* lvl_start, const float* soi, const int32_t* strides,
                              int L, float radius, const float* gt_boxes, const int32_t* num_gt, int N, int Gmax,
                              int32_t* labels, float* offsets, float* ctrness, float* stats, bd_stream_t stream) {
    BD_REQUIRE(points && lvl_start && soi && strides && gt_boxes && num_gt && labels && offsets && ctrness && stats,
               "fcos_assign: null pointer");
    BD_REQUIRE(L >= 1 && L <= BD_MAX_SEGS && P > 0 && N > 0 && Gmax > 0, "fcos_assign: bad sizes");
    FcosLevels lv{};
    lv.L = L;
    for (int l = 0; l < L; ++l) {
        lv.start[l] = lvl_start[l];
        lv.lo[l] = soi[2 * l]; lv.hi[l] = soi[2 * l + 1];
        lv.radius[l] = (float)strides[l] * radius;
    }
    lv.start[L] = lvl_start[L];
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(stats, 0, 2 * sizeof(float), st);
    hipLaunchKernelGGL(fcos_assign_kernel, dim3(cdiv(P, 256), N), dim3(256), 0, st, points, P, lv, radius > 0.f ? 1 : 0,
                       gt_boxes, num_gt, Gmax, labels, offsets, ctrness, stats);
    hipLaunchKernelGGL(ctr_sum_kernel, dim3(1), dim3(1024), 0, st, (const int*)labels, (const float*)ctrness, (long long)N * P, stats);
    BD_CHECK_LAUNCH("bd_fcos_assign");
    return BD_OK;
}

extern "C" size_t bd_atss_assign_workspace_bytes(int N, int P) { return (size_t)(N > 0 ? N : 0) * (size_t)(P > 0 ? P : 0) * 8 + 64; }

extern "C" int bd_atss_assign(const float* points, int P, const int32_t* lvl_start, const int32_t* strides, int L, int topk,
                              float anchor_scale, const float* gt_boxes, const int32_t* num_gt, int N, int Gmax, int32_t* labels,
                              float* offsets, float* ctrness, float* stats, void* ws, size_t ws_bytes, bd_stream_t stream) {
    BD_REQUIRE(points && lvl_start && strides && gt_boxes && num_gt && labels && offsets && ctrness && stats && ws, "atss_assign: null pointer");
    BD_REQUIRE(L >= 1 && L <= BD_MAX_SEGS && P > 0 && N > 0 && Gmax > 0 && topk >= 1 && topk <= ATSS_TOPK_MAX, "atss_assign: bad sizes");
    if (ws_bytes < bd_atss_assign_workspace_bytes(N, P)) {
        bd_set_error("atss_assign: workspace %zu < %zu bytes", ws_bytes, bd_atss_assign_workspace_bytes(N, P));
        return BD_EWORKSPACE;
    }
    FcosLevels lv{};
    lv.L = L;
    for (int l = 0; l < L; ++l) { lv.start[l] = lvl_start[l]; lv.radius[l] = (float)strides[l]; }
    lv.start[L] = lvl_start[L];
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(ws, 0, (size_t)N * P * 8, st);
    (void)hipMemsetAsync(stats, 0, 2 * sizeof(float), st);
    hipLaunchKernelGGL(atss_candidates_kernel, dim3(Gmax, N), dim3(256), 0, st, points, lv, topk, anchor_scale / 2.f, gt_boxes, num_gt, Gmax, P,
                       (unsigned long long*)ws);
    hipLaunchKernelGGL(atss_finalize_kernel, dim3(cdiv(P, 256), N), dim3(256), 0, st, points, P, gt_boxes, num_gt, Gmax,
                       (const unsigned long long*)ws, labels, offsets, ctrness, stats);
    hipLaunchKernelGGL(ctr_sum_kernel, dim3(1), dim3(1024), 0, st, (const int*)labels, (const float*)ctrness, (long long)N * P, stats);
    BD_CHECK_LAUNCH("bd_atss_assign");
    return BD_OK;
}

static inline int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

extern "C" size_t bd_nms_workspace_bytes(int n) {
    if (n <= 0) return 16;
    const size_t words = (size_t)cdiv(n, 64);
    return (size_t)n * 16 + (size_t)n * 4 + (size_t)n * words * 8 + 64;
}

extern "C" int bd_batched_nms(const float* boxes, const float* scores, const int32_t* idxs, int n, float iou_thresh,
                              int max_output, int32_t* keep, int32_t* num_keep, void* ws, size_t ws_bytes,
                              bd_stream_t stream) {
    BD_REQUIRE(num_keep, "batched_nms: null num_keep");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { (void)hipMemsetAsync(num_keep, 0, sizeof(int32_t), st); return BD_OK; }
    BD_REQUIRE(boxes && scores && keep && ws, "batched_nms: null pointer");
    BD_REQUIRE(n > 0 && n <= NMS_MAX, "batched_nms: n=%d out of range (1..%d)", n, NMS_MAX);
    if (ws_bytes < bd_nms_workspace_bytes(n)) {
        bd_set_error("batched_nms: workspace %zu < %zu bytes", ws_bytes, bd_nms_workspace_bytes(n));
        return BD_EWORKSPACE;
    }
    const int words = cdiv(n, 64);
    unsigned char* p = (unsigned char*)ws;
    float* sboxes = (float*)p;                 p += (size_t)n * 16;
    int* order = (int*)p;                      p += (((size_t)n * 4 + 15) / 16) * 16;
    unsigned long long* mask = (unsigned long long*)p;
    const int npow2 = next_pow2(n);
    const size_t lds = (size_t)npow2 * 8;
    BD_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nms_prepare_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  NMS_MAX * 8));
    hipLaunchKernelGGL(nms_prepare_kernel, dim3(1), dim3(1024), lds, st, boxes, scores, idxs, n, npow2, sboxes, order);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(n, words), dim3(64), 0, st, (const float*)sboxes, (const int*)order, n,
                       iou_thresh, mask, words);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(64), 0, st, (const unsigned long long*)mask, (const int*)order, n, words,
                       max_output, keep, num_keep);
    BD_CHECK_LAUNCH("bd_batched_nms");
    return BD_OK;
}
