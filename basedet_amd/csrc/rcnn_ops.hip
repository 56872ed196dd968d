// Faster R-CNN operators on the HIP path: RPN proposal selection (per-level top-k, decode, clip, size filter, NMS across
// levels), random subsampling of labels, RoI sampling + target encoding, multi-level RoIAlign forward / backward and the
// FPNP6 sub-sampling.  Reference: models/det/rpn.py, layers/head/rcnn.py, layers/common/roi_pool.py,
// layers/common/sampling.py, layers/backbone/fpn_backbone.py:172-183.
//
// Everything that decides an index (top-k membership, NMS survivors, fg/bg sampling) is integer / exact-fp32 work and is
// bit-exact against oracle/rcnn_ops.py; this translation unit is compiled with -ffp-contract=off like boxops.hip.
// All kernels are HBM/LDS-latency bound selection passes: one workgroup per (image, level) or per image, wave ballots
// and LDS histograms instead of global sorts.
#pragma clang fp contract(off)
#include "select_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------------------
// segmented top-k (rpn.py:155 F.topk per level; retinanet.py:188-192 / fcos.py:196-204 at inference)
// ------------------------------------------------------------------------------------------------------------
constexpr int TOPK_MAX = 2048;
struct TopkSegs { int start[BD_MAX_SEGS]; int count[BD_MAX_SEGS]; int nseg; };

// Item i of a segment is element (row = i / A, a = i % A) at scores[batch*batch_stride + (start+row)*ldc + coff + a].
// Order of the output: score descending, then item index ascending (a stable descending sort).
template <bool BF16>
__global__ __launch_bounds__(1024) void segment_topk_kernel(const void* __restrict__ scores, long long batch_stride, int A,
                                                            int ldc, int coff, TopkSegs segs, int k, float min_score,
                                                            int use_min, int* __restrict__ out_idx,
                                                            float* __restrict__ out_score, int* __restrict__ out_cnt) {
    __shared__ unsigned long long keys[TOPK_MAX];
    __shared__ unsigned int hist[256];
    __shared__ int sh[8];
    __shared__ int wcnt[16];
    const int tid = threadIdx.x;
    const int seg = blockIdx.x, n = blockIdx.y;
    const int cnt = segs.count[seg];
    const float invA = 1.f / (float)A;
    const long long base = (long long)n * batch_stride + (long long)segs.start[seg] * ldc + coff;
    auto key = [&](int i, bool& valid) -> unsigned int {
        int row = i, a = 0;
        if (A != 1) {
            row = (int)((float)i * invA);
            a = i - row * A;
            if (a < 0) { --row; a += A; } else if (a >= A) { ++row; a -= A; }
        }
        const long long e = base + (long long)row * ldc + a;
        const float v = BF16 ? bf2f(reinterpret_cast<const bf16_raw*>(scores)[e]) : reinterpret_cast<const float*>(scores)[e];
        valid = !use_min || v > min_score;
        return f32_asc_key(v);
    };
    const SelResult r = radix_select_largest(cnt, k, BF16 ? 2 : 4, key, hist, sh);
    // bf16 scores: two radix passes fix the upper 16 key bits, and r.T carries zeros below them -- but the key of a NEGATIVE bf16 has 0xffff
    // there (f32_asc_key complements negative values).  Rounds 1-4 compared whole keys with r.T: every item of a negative threshold bin then
    // counted as "above" it, none as a tie.  With few ties that only over-filled `keys` by the surplus (still sorted and cut correctly);
    // with many (found by round 5's tie-heavy proposal test: 17 distinct scores) the surplus ran over the slots of the later ties and past
    // the array.  The threshold tests use the selected bits only.
    const unsigned int sel_mask = BF16 ? 0xffff0000u : 0xffffffffu;
    for (int i = tid; i < TOPK_MAX; i += 1024) keys[i] = 0ull;
    if (tid == 0) sh[4] = 0;
    __syncthreads();
    const int n_gt = r.take_all ? r.n_valid : k - r.need_eq;
    const int m = r.take_all ? r.n_valid : k;
    // unordered collection of everything above the threshold
    for (int i = tid; i < cnt; i += 1024) {
        bool valid;
        const unsigned int kv = key(i, valid);
        if (valid && (r.take_all || (kv & sel_mask) > r.T)) {
            const int p = atomicAdd(&sh[4], 1);
            if (p < TOPK_MAX) keys[p] = ((unsigned long long)kv << 32) | (0xffffffffu - (unsigned int)i);
        }
    }
    // ties at the threshold: lowest indices first
    if (!r.take_all) {
        int eq_base = 0;
        for (int c0 = 0; c0 < cnt && eq_base < r.need_eq; c0 += 1024) {
            const int i = c0 + tid;
            bool valid = false;
            unsigned int kv = 0;
            if (i < cnt) kv = key(i, valid);
            const bool eq = valid && (kv & sel_mask) == r.T;
            int tot;
            const int my = eq_base + block_rank_1024(eq, wcnt, tot);
            if (eq && my < r.need_eq) keys[n_gt + my] = ((unsigned long long)kv << 32) | (0xffffffffu - (unsigned int)i);
            eq_base += tot;
        }
    }
    __syncthreads();
    // bitonic sort, descending
    for (int kk = 2; kk <= TOPK_MAX; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < TOPK_MAX; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], b = keys[ixj];
                    const bool up = (i & kk) == 0;
                    if ((a < b) == up) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    const long long ob = ((long long)n * segs.nseg + seg) * k;
    for (int i = tid; i < k; i += 1024) {
        int idx = -1;
        float sc = 0.f;
        if (i < m) {
            const unsigned long long kv = keys[i];
            idx = (int)(0xffffffffu - (unsigned int)(kv & 0xffffffffull));
            sc = f32_from_asc_key((unsigned int)(kv >> 32));
        }
        out_idx[ob + i] = idx;
        out_score[ob + i] = sc;
    }
    if (tid == 0) out_cnt[(long long)n * segs.nseg + seg] = m;
}

// ------------------------------------------------------------------------------------------------------------
// RPN: decode + clip + size filter of the per-level top-k candidates (rpn.py:150-172)
// ------------------------------------------------------------------------------------------------------------
struct RpnLevels { int pix_off[BD_MAX_SEGS]; int cand_off[BD_MAX_SEGS + 1]; int L; };

__global__ __launch_bounds__(256) void rpn_decode_kernel(const bf16_raw* __restrict__ raw, long long ppi, int ldc, int A, int doff,
                                                         const float* __restrict__ anchors, RpnLevels lv, int k,
                                                         const int* __restrict__ topk_idx, const float* __restrict__ topk_score,
                                                         const int* __restrict__ topk_cnt, const float* __restrict__ im_info,
                                                         int info_ld, Coder coder, int C, float* __restrict__ boxes,
                                                         float* __restrict__ scores, int* __restrict__ levels) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    if (c >= C) return;
    int l = 0;
    for (int q = 1; q < lv.L; ++q) if (c >= lv.cand_off[q]) l = q;
    const int rnk = c - lv.cand_off[l];
    f32x4_t box = {0.f, 0.f, 0.f, 0.f};
    float sc = -INFINITY;
    if (rnk < topk_cnt[n * lv.L + l]) {
        const long long t = ((long long)n * lv.L + l) * k + rnk;
        const int idx = topk_idx[t];
        const int pixel = idx / A, a = idx - pixel * A;
        const Box an = ld_box(anchors + ((long long)lv.pix_off[l] * A + idx) * 4);
        const bf16_raw* dp = raw + ((long long)n * ppi + lv.pix_off[l] + pixel) * ldc + doff + a * 4;
        const f32x4_t d = {bf2f(dp[0]), bf2f(dp[1]), bf2f(dp[2]), bf2f(dp[3])};
        f32x4_t b = decode_dev(an, d, coder);
        const float h = im_info[n * info_ld + 0], w = im_info[n * info_ld + 1];
        b[0] = fminf(fmaxf(b[0], 0.f), w); b[1] = fminf(fmaxf(b[1], 0.f), h);     // Boxes.clip (structures/boxes.py:152-176)
        b[2] = fminf(fmaxf(b[2], 0.f), w); b[3] = fminf(fmaxf(b[3], 0.f), h);
        if ((b[2] - b[0]) > 0.f && (b[3] - b[1]) > 0.f) {                           // filter_by_size (boxes.py:132-150)
            box = b;
            sc = topk_score[t];
        }
    }
    const long long o = (long long)n * C + c;
    *reinterpret_cast<f32x4_t*>(boxes + o * 4) = box;
    scores[o] = sc;
    levels[o] = l;
}

// ------------------------------------------------------------------------------------------------------------
// batched NMS over B independent problems of capacity C (post_processing.py:17-47 per problem).
// Items with score == -inf are absent.  Greedy order: score descending, then index ascending.
// ------------------------------------------------------------------------------------------------------------
constexpr int NMSB_MAX = 16384;

__global__ __launch_bounds__(1024) void nmsb_prepare_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                            const int* __restrict__ idxs, int C, int npow2,
                                                            float* __restrict__ sboxes, int* __restrict__ order,
                                                            int* __restrict__ nvalid) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    __shared__ float red[16];
    __shared__ int cnt_sh;
    const int tid = threadIdx.x;
    const long long b0 = (long long)blockIdx.x * C;
    boxes += b0 * 4; scores += b0; sboxes += b0 * 4; order += b0;
    if (idxs) idxs += b0;
    if (tid == 0) cnt_sh = 0;
    float mx = -INFINITY;
    int nv = 0;
    for (int i = tid; i < C; i += 1024) {
        if (scores[i] > -INFINITY) {
            const Box b = ld_box(boxes + i * 4ll);
            mx = fmaxf(mx, fmaxf(fmaxf(b.x1, b.y1), fmaxf(b.x2, b.y2)));
            ++nv;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    if (nv) atomicAdd(&cnt_sh, nv);
    mx = red[0];
    for (int q = 1; q < 16; ++q) mx = fmaxf(mx, red[q]);
    const float step = mx + 1.f;                                   // post_processing.py:44-45
    for (int i = tid; i < npow2; i += 1024) {
        unsigned long long key = ~0ull;
        if (i < C) {
            key = ((unsigned long long)float_desc_key(scores[i]) << 32) | (unsigned int)i;
            const float off = idxs ? (float)idxs[i] * step : 0.f;
            const Box b = ld_box(boxes + i * 4ll);
            f32x4_t o = {b.x1 + off, b.y1 + off, b.x2 + off, b.y2 + off};
            *reinterpret_cast<f32x4_t*>(sboxes + i * 4ll) = o;
        }
        keys[i] = key;
    }
    __syncthreads();
    for (int kk = 2; kk <= npow2; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npow2; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], b = keys[ixj];
                    const bool up = (i & kk) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < C; i += 1024) order[i] = (int)(keys[i] & 0xffffffffu);
    if (tid == 0) nvalid[blockIdx.x] = cnt_sh;
}

// 64 sorted rows x 64 sorted columns per wave; mask[i][w] bit b: sorted box (64w+b) is suppressed by sorted box i.
// Only the upper triangle (w >= i/64) inside the valid range is written -- the scan never reads anything else.
__global__ __launch_bounds__(64) void nmsb_mask_kernel(const float* __restrict__ sboxes, const int* __restrict__ order,
                                                       const int* __restrict__ nvalid, int C, int words, float thr,
                                                       unsigned long long* __restrict__ mask) {
    __shared__ float rows[64 * 4];
    const int rt = blockIdx.x, w = blockIdx.y, b = blockIdx.z;
    const int nv = nvalid[b];
    if (w < rt || rt * 64 >= nv || w * 64 >= nv) return;
    const long long b0 = (long long)b * C;
    sboxes += b0 * 4; order += b0; mask += b0 * words;
    const int lane = threadIdx.x;
    const int j = w * 64 + lane;
    Box cb{0.f, 0.f, 0.f, 0.f};
    if (j < nv) cb = ld_box(sboxes + order[j] * 4ll);
    const int ri = rt * 64 + lane;
    f32x4_t rb = {0.f, 0.f, 0.f, 0.f};
    if (ri < nv) rb = *reinterpret_cast<const f32x4_t*>(sboxes + order[ri] * 4ll);
    *reinterpret_cast<f32x4_t*>(rows + lane * 4) = rb;
    __syncthreads();
    const float ca = box_area(cb);
    const int rmax = min(64, nv - rt * 64);
    unsigned long long mine = 0ull;
    for (int q = 0; q < rmax; ++q) {
        const int i = rt * 64 + q;
        const Box a = Box{rows[q * 4], rows[q * 4 + 1], rows[q * 4 + 2], rows[q * 4 + 3]};
        bool sup = false;
        if (j < nv && j > i) {
            const float inter = box_inter(a, cb);
            const float uni = (box_area(a) + ca) - inter;
            sup = (inter / uni) > thr;     // keep iff iou <= thr (py_cpu_nms, post_processing.py:130)
        }
        const unsigned long long bal = __ballot(sup);
        if (lane == q) mine = bal;
    }
    if (lane < rmax) mask[(long long)(rt * 64 + lane) * words + w] = mine;
}

// one wave per problem walks the sorted list 64 boxes at a time: the diagonal 64x64 block is resolved in registers,
// then the rows of the survivors are OR-ed into the LDS `removed` bitmap with independent (pipelined) loads.
__global__ __launch_bounds__(64) void nmsb_scan_kernel(const unsigned long long* __restrict__ mask, const int* __restrict__ order,
                                                       const int* __restrict__ nvalid, int C, int words, int max_output,
                                                       int keep_ld, int* __restrict__ keep, int* __restrict__ num_keep) {
    __shared__ unsigned long long removed[NMSB_MAX / 64];
    const int b = blockIdx.x, lane = threadIdx.x;
    const long long b0 = (long long)b * C;
    mask += b0 * words; order += b0; keep += (long long)b * keep_ld;
    const int nv = nvalid[b];
    const int nw = (nv + 63) >> 6;
    for (int w = lane; w < nw; w += 64) removed[w] = 0ull;
    __syncthreads();
    int cnt = 0;
    const int cap = max_output > 0 ? max_output : nv;
    for (int c = 0; c < nw && cnt < cap; ++c) {
        const int i0 = c * 64;
        const int i = i0 + lane;
        const unsigned long long diag = i < nv ? mask[(long long)i * words + c] : 0ull;
        const int nin = min(64, nv - i0);
        unsigned long long alive = ~removed[c];
        if (nin < 64) alive &= (1ull << nin) - 1ull;
        unsigned long long kept = 0ull;
        const unsigned int dlo = (unsigned int)diag, dhi = (unsigned int)(diag >> 32);
        for (int q = 0; q < nin; ++q) {
            if ((alive >> q) & 1ull) {                                   // wave-uniform
                kept |= 1ull << q;
                const unsigned long long row = ((unsigned long long)(unsigned int)__shfl((int)dhi, q, 64) << 32) |
                                               (unsigned int)__shfl((int)dlo, q, 64);
                alive &= ~row;
            }
        }
        // honour max_output: keep only the first (cap - cnt) survivors of this chunk
        int nk = __popcll(kept);
        if (cnt + nk > cap) {
            int drop = cnt + nk - cap;
            while (drop > 0) { kept &= ~(1ull << (63 - __builtin_clzll(kept))); --drop; }
            nk = cap - cnt;
        }
        if ((kept >> lane) & 1ull) keep[cnt + __popcll(kept & ((1ull << lane) - 1ull))] = order[i];
        cnt += nk;
        if (cnt >= cap) break;
        for (int w = c + 1 + lane; w < nw; w += 64) {
            unsigned long long acc = 0ull;
            unsigned long long kk = kept;
            while (kk) {
                const int q = __ffsll((long long)kk) - 1;
                kk &= kk - 1ull;
                acc |= mask[(long long)(i0 + q) * words + w];
            }
            removed[w] |= acc;
        }
        __syncthreads();
    }
    if (lane == 0) num_keep[b] = cnt;
}

// ------------------------------------------------------------------------------------------------------------
// RPN proposals, round 5: the batched NMS level by level.  rpn.py:163-172 runs ONE batched_nms over an image's candidates with the
// pyramid level as the class id: boxes of different levels are shifted apart before the greedy pass (post_processing.py:44-45), so that
// pass decomposes into L independent ones -- 16 x 5 problems of <= 2 048 boxes instead of 16 of ~8 900 (a 157-chunk serial scan per
// image: nmsb_scan 0.76 ms, nmsb_mask 0.47 ms, the 16 384-key sort 0.26 ms per step at batch 16) -- and the joint keep list (score
// descending, candidate index ascending, first post_k) is the MERGE of the per-level lists.  The boxes keep the joint form's shift
// (level x (max coordinate + 1), in fp32): the IoUs, and with them every keep decision, are the same bits.
// ------------------------------------------------------------------------------------------------------------
constexpr int NMSL_CAP = 2048;                  // candidates per (image, level): pre_k <= TOPK_MAX
constexpr int NMSL_WORDS = NMSL_CAP / 64;

// grid (L, N): sorts one level's candidates (score descending, index ascending), writes the shifted boxes of that level
__global__ __launch_bounds__(1024) void nmsl_prepare_kernel(const float* __restrict__ boxes, const float* __restrict__ scores, RpnLevels lv,
                                                            int C, float* __restrict__ sboxes, int* __restrict__ order,
                                                            int* __restrict__ nvalid) {
    __shared__ unsigned long long keys[NMSL_CAP];
    __shared__ float red[16];
    __shared__ int cnt_sh;
    const int tid = threadIdx.x, l = blockIdx.x, n = blockIdx.y;
    const long long b0 = (long long)n * C;
    boxes += b0 * 4; scores += b0; sboxes += b0 * 4; order += b0;
    if (tid == 0) cnt_sh = 0;
    // the shift step of the JOINT problem: largest coordinate over ALL of the image's candidates + 1
    float mx = -INFINITY;
    for (int i = tid; i < C; i += 1024) {
        if (scores[i] > -INFINITY) {
            const Box b = ld_box(boxes + i * 4ll);
            mx = fmaxf(mx, fmaxf(fmaxf(b.x1, b.y1), fmaxf(b.x2, b.y2)));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = red[0];
    for (int q = 1; q < 16; ++q) mx = fmaxf(mx, red[q]);
    const float off = (float)l * (mx + 1.f);
    const int c0 = lv.cand_off[l], cap = lv.cand_off[l + 1] - c0;
    int nv = 0;
    for (int i = tid; i < NMSL_CAP; i += 1024) {
        unsigned long long key = ~0ull;
        if (i < cap) {
            const float sc = scores[c0 + i];
            key = ((unsigned long long)float_desc_key(sc) << 32) | (unsigned int)(c0 + i);
            nv += sc > -INFINITY;
            const Box b = ld_box(boxes + (c0 + i) * 4ll);
            f32x4_t o = {b.x1 + off, b.y1 + off, b.x2 + off, b.y2 + off};
            *reinterpret_cast<f32x4_t*>(sboxes + (c0 + i) * 4ll) = o;
        }
        keys[i] = key;
    }
    if (nv) atomicAdd(&cnt_sh, nv);
    __syncthreads();
    for (int kk = 2; kk <= NMSL_CAP; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < NMSL_CAP; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], b = keys[ixj];
                    const bool up = (i & kk) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < cap; i += 1024) order[c0 + i] = (int)(keys[i] & 0xffffffffu);        // candidate index inside the image
    if (tid == 0) nvalid[n * lv.L + l] = cnt_sh;
}

// grid (32 row tiles, 32 words, N * L), one wave: as nmsb_mask_kernel on one level's sorted list
__global__ __launch_bounds__(64) void nmsl_mask_kernel(const float* __restrict__ sboxes, const int* __restrict__ order,
                                                       const int* __restrict__ nvalid, RpnLevels lv, int C, float thr,
                                                       unsigned long long* __restrict__ mask) {
    __shared__ float rows[64 * 4];
    const int rt = blockIdx.x, w = blockIdx.y, b = blockIdx.z;
    const int nv = nvalid[b];
    if (w < rt || rt * 64 >= nv || w * 64 >= nv) return;
    const int n = b / lv.L, l = b - n * lv.L;
    sboxes += (long long)n * C * 4; order += (long long)n * C + lv.cand_off[l]; mask += (long long)b * NMSL_CAP * NMSL_WORDS;
    const int lane = threadIdx.x;
    const int j = w * 64 + lane;
    Box cb{0.f, 0.f, 0.f, 0.f};
    if (j < nv) cb = ld_box(sboxes + order[j] * 4ll);
    const int ri = rt * 64 + lane;
    f32x4_t rb = {0.f, 0.f, 0.f, 0.f};
    if (ri < nv) rb = *reinterpret_cast<const f32x4_t*>(sboxes + order[ri] * 4ll);
    *reinterpret_cast<f32x4_t*>(rows + lane * 4) = rb;
    __syncthreads();
    const float ca = box_area(cb);
    const int rmax = min(64, nv - rt * 64);
    unsigned long long mine = 0ull;
    for (int q = 0; q < rmax; ++q) {
        const int i = rt * 64 + q;
        const Box a = Box{rows[q * 4], rows[q * 4 + 1], rows[q * 4 + 2], rows[q * 4 + 3]};
        bool sup = false;
        if (j < nv && j > i) {
            const float inter = box_inter(a, cb);
            const float uni = (box_area(a) + ca) - inter;
            sup = (inter / uni) > thr;     // keep iff iou <= thr (py_cpu_nms, post_processing.py:130)
        }
        const unsigned long long bal = __ballot(sup);
        if (lane == q) mine = bal;
    }
    if (lane < rmax) mask[(long long)(rt * 64 + lane) * NMSL_WORDS + w] = mine;
}

// grid (N * L), one wave: as nmsb_scan_kernel; a level keeps at most post_k boxes (no more of them can reach the joint list's first post_k)
__global__ __launch_bounds__(64) void nmsl_scan_kernel(const unsigned long long* __restrict__ mask, const int* __restrict__ order,
                                                       const int* __restrict__ nvalid, RpnLevels lv, int C, int post_k,
                                                       int* __restrict__ keep_l, int* __restrict__ num_l) {
    __shared__ unsigned long long removed[NMSL_WORDS];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = b / lv.L, l = b - n * lv.L;
    mask += (long long)b * NMSL_CAP * NMSL_WORDS; order += (long long)n * C + lv.cand_off[l]; keep_l += (long long)b * post_k;
    const int nv = nvalid[b];
    const int nw = (nv + 63) >> 6;
    if (lane < NMSL_WORDS) removed[lane] = 0ull;
    __syncthreads();
    int cnt = 0;
    const int cap = post_k;
    for (int c = 0; c < nw && cnt < cap; ++c) {
        const int i0 = c * 64;
        const int i = i0 + lane;
        const unsigned long long diag = i < nv ? mask[(long long)i * NMSL_WORDS + c] : 0ull;
        const int nin = min(64, nv - i0);
        unsigned long long alive = ~removed[c];
        if (nin < 64) alive &= (1ull << nin) - 1ull;
        unsigned long long kept = 0ull;
        const unsigned int dlo = (unsigned int)diag, dhi = (unsigned int)(diag >> 32);
        for (int q = 0; q < nin; ++q) {
            if ((alive >> q) & 1ull) {                                   // wave-uniform
                kept |= 1ull << q;
                const unsigned long long row = ((unsigned long long)(unsigned int)__shfl((int)dhi, q, 64) << 32) |
                                               (unsigned int)__shfl((int)dlo, q, 64);
                alive &= ~row;
            }
        }
        int nk = __popcll(kept);
        if (cnt + nk > cap) {
            int drop = cnt + nk - cap;
            while (drop > 0) { kept &= ~(1ull << (63 - __builtin_clzll(kept))); --drop; }
            nk = cap - cnt;
        }
        if ((kept >> lane) & 1ull) keep_l[cnt + __popcll(kept & ((1ull << lane) - 1ull))] = order[i];
        cnt += nk;
        if (cnt >= cap) break;
        const int w = c + 1 + lane;
        if (w < nw) {
            unsigned long long acc = 0ull;
            unsigned long long kk = kept;
            while (kk) {
                const int q = __ffsll((long long)kk) - 1;
                kk &= kk - 1ull;
                acc |= mask[(long long)(i0 + q) * NMSL_WORDS + w];
            }
            removed[w] |= acc;
        }
        __syncthreads();
    }
    if (lane == 0) num_l[b] = cnt;
}

// grid (N): joint rank of every kept box = its rank inside its level + the kept boxes of the other levels that precede it in
// (score descending, candidate index ascending) order -- levels are laid out in candidate-index order, so a box of a LOWER level precedes
// on a score tie, one of a higher level does not.  Binary searches over the per-level lists (sorted by construction).
__global__ __launch_bounds__(1024) void nmsl_merge_kernel(const float* __restrict__ scores, const int* __restrict__ keep_l,
                                                          const int* __restrict__ num_l, RpnLevels lv, int C, int post_k,
                                                          int* __restrict__ keep, int* __restrict__ num_keep) {
    const int n = blockIdx.x, tid = threadIdx.x;
    scores += (long long)n * C; keep_l += (long long)n * lv.L * post_k; num_l += n * lv.L; keep += (long long)n * post_k;
    int tot = 0;
    for (int l = 0; l < lv.L; ++l) tot += num_l[l];
    for (int l = 0; l < lv.L; ++l) {
        const int nl = num_l[l];
        for (int r = tid; r < nl; r += 1024) {
            const int c = keep_l[l * post_k + r];
            const unsigned int key = float_desc_key(scores[c]);
            int rank = r;
            for (int o = 0; o < lv.L; ++o) {
                if (o == l) continue;
                const int* lst = keep_l + o * post_k;
                int lo = 0, hi = num_l[o];               // first index whose key is > key (o < l: ties precede) or >= key (o > l)
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    const unsigned int km = float_desc_key(scores[lst[mid]]);
                    const bool before = o < l ? km <= key : km < key;
                    if (before) lo = mid + 1; else hi = mid;
                }
                rank += lo;
            }
            if (rank < post_k) keep[rank] = c;
        }
    }
    if (tid == 0) num_keep[n] = tot < post_k ? tot : post_k;
}

__global__ __launch_bounds__(256) void rpn_gather_kernel(const float* __restrict__ boxes, const int* __restrict__ keep,
                                                         const int* __restrict__ num_keep, int C, int post_k,
                                                         float* __restrict__ rois) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    if (j >= post_k) return;
    f32x4_t v = {0.f, 0.f, 0.f, 0.f};
    if (j < num_keep[n]) v = *reinterpret_cast<const f32x4_t*>(boxes + ((long long)n * C + keep[(long long)n * post_k + j]) * 4);
    *reinterpret_cast<f32x4_t*>(rois + ((long long)n * post_k + j) * 4) = v;
}

// ------------------------------------------------------------------------------------------------------------
// sample_labels (layers/common/sampling.py:7-30) applied as in RPN.get_ground_truth (rpn.py:229-232)
// ------------------------------------------------------------------------------------------------------------
// keeps the `k` smallest random keys among the items with labels == value (ties: lowest index), the rest become -1
__device__ void subsample_value(int* __restrict__ labels, const float* __restrict__ keys, int A, int value, int k,
                                unsigned int* hist, int* sh, int* wcnt) {
    const int tid = threadIdx.x;
    auto key = [&](int i, bool& valid) -> unsigned int {
        valid = labels[i] == value;
        return ~__float_as_uint(keys[i]);          // keys in [0,1): bit order == value order; inverted -> smallest wins
    };
    if (k <= 0) {
        for (int i = tid; i < A; i += 1024) if (labels[i] == value) labels[i] = -1;
        __syncthreads();
        return;
    }
    const SelResult r = radix_select_largest(A, k, 4, key, hist, sh);
    if (r.take_all) return;
    int eq_base = 0;
    for (int c0 = 0; c0 < A; c0 += 1024) {
        const int i = c0 + tid;
        bool valid = false;
        unsigned int kv = 0;
        if (i < A) kv = key(i, valid);
        const bool eq = valid && kv == r.T;
        int tot;
        const int my = eq_base + block_rank_1024(eq, wcnt, tot);
        if (valid && (kv < r.T || (eq && my >= r.need_eq))) labels[i] = -1;
        eq_base += tot;
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void sample_labels_kernel(int* __restrict__ labels, const float* __restrict__ keys_pos,
                                                             const float* __restrict__ keys_neg, int A, int num_pos_max,
                                                             int num_total, int* __restrict__ num_valid) {
    __shared__ unsigned int hist[256];
    __shared__ int sh[8];
    __shared__ int wcnt[16];
    __shared__ int cnts[2];
    const int n = blockIdx.x, tid = threadIdx.x;
    labels += (long long)n * A; keys_pos += (long long)n * A; keys_neg += (long long)n * A;
    if (tid == 0) { cnts[0] = 0; cnts[1] = 0; }
    __syncthreads();
    int p = 0, q = 0;
    for (int i = tid; i < A; i += 1024) { const int l = labels[i]; p += l == 1; q += l == 0; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { p += __shfl_xor(p, o, 64); q += __shfl_xor(q, o, 64); }
    if ((tid & 63) == 0) { atomicAdd(&cnts[0], p); atomicAdd(&cnts[1], q); }
    __syncthreads();
    const int npos = cnts[0], nneg = cnts[1];
    int pos_kept = npos;
    if (npos > num_pos_max) { subsample_value(labels, keys_pos, A, 1, num_pos_max, hist, sh, wcnt); pos_kept = num_pos_max; }
    const int neg_max = num_total - pos_kept;
    int neg_kept = nneg;
    if (nneg > neg_max) { subsample_value(labels, keys_neg, A, 0, neg_max, hist, sh, wcnt); neg_kept = neg_max > 0 ? neg_max : 0; }
    if (tid == 0) atomicAdd(num_valid, pos_kept + neg_kept);
}

// ------------------------------------------------------------------------------------------------------------
// RCNN.get_ground_truth (layers/head/rcnn.py:95-147): one workgroup per image
// ------------------------------------------------------------------------------------------------------------
constexpr int RCNN_CAP = 2048;

__global__ __launch_bounds__(1024) void rcnn_sample_kernel(const float* __restrict__ rois, const int* __restrict__ num_rois, int post_k,
                                                           const float* __restrict__ gt_boxes, const int* __restrict__ num_gt,
                                                           int Gmax, const float* __restrict__ keys_fg,
                                                           const float* __restrict__ keys_bg, int key_ld, int num_samples,
                                                           int num_fg_max, float fg_thr, float bg_hi, float bg_lo, Coder coder,
                                                           float* __restrict__ out_rois, int* __restrict__ out_labels,
                                                           float* __restrict__ out_targets, int* __restrict__ out_count,
                                                           int* __restrict__ total_count) {
    __shared__ float s_key[RCNN_CAP];
    __shared__ unsigned char s_flag[RCNN_CAP];      // bit0 fg candidate, bit1 bg candidate, bit2 kept
    __shared__ int wcnt[16];
    __shared__ int cnts[2];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int nr = min(num_rois[n], post_k);
    const int G = min(num_gt[n], Gmax);
    const int M = nr + G;
    const float* gp = gt_boxes + (long long)n * Gmax * 5;
    if (tid == 0) { cnts[0] = 0; cnts[1] = 0; }
    __syncthreads();
    Box bx[2];
    int am[2], lab[2];
    // thread t owns items 2t and 2t+1 (index order == thread order for the compaction)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = 2 * tid + u;
        bx[u] = Box{0.f, 0.f, 0.f, 0.f};
        am[u] = 0; lab[u] = 0;
        unsigned char fl = 0;
        if (i < M) {
            bx[u] = i < nr ? ld_box(rois + ((long long)n * post_k + i) * 4) : ld_gt(gp + (i - nr) * 5);
            const float ba = box_area(bx[u]);
            float best = -1.f;
            for (int g = 0; g < G; ++g) {
                const Box gb = ld_gt(gp + g * 5);
                const float iou = box_iou_dev(bx[u], ba, gb, box_area(gb));
                if (iou > best) { best = iou; am[u] = g; }           // first maximum (argmax)
            }
            if (G == 0) best = 0.f;
            lab[u] = G > 0 ? (int)gp[am[u] * 5 + 4] : 0;
            const bool fg = best >= fg_thr && lab[u] >= 0 && G > 0;
            const bool bg = best >= bg_lo && best < bg_hi;
            fl = (fg ? 1 : 0) | (bg ? 2 : 0);
            if (fg) atomicAdd(&cnts[0], 1);
            if (bg) atomicAdd(&cnts[1], 1);
        }
        if (i < RCNN_CAP) s_flag[i] = fl;
    }
    __syncthreads();
    const int nfg = cnts[0], nbg = cnts[1];
    const int fg_kept = min(nfg, num_fg_max);
    const int bg_max = num_samples - fg_kept;
    // two rounds: fg with keys_fg, bg with keys_bg.  keep the items with the smallest keys (ties: lowest index)
    for (int round = 0; round < 2; ++round) {
        const unsigned char bit = round == 0 ? 1 : 2;
        const int have = round == 0 ? nfg : nbg;
        const int limit = round == 0 ? num_fg_max : bg_max;
        const float* kp = (round == 0 ? keys_fg : keys_bg) + (long long)n * key_ld;
        if (have <= limit) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = 2 * tid + u;
                if (i < M && (s_flag[i] & bit)) s_flag[i] |= 4;
            }
        } else {
            __syncthreads();
            for (int i = tid; i < M; i += 1024) s_key[i] = kp[i];
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = 2 * tid + u;
                if (i < M && (s_flag[i] & bit)) {
                    const float ki = s_key[i];
                    int rank = 0;
                    for (int j = 0; j < M; ++j)
                        if ((s_flag[j] & bit) && (s_key[j] < ki || (s_key[j] == ki && j < i))) ++rank;
                    if (rank < limit) s_flag[i] |= 4;
                }
            }
        }
        __syncthreads();
    }
    // rcnn.py:128 labels[bg_inds_mask] = 0 (after the fg sampling: an item can be fg-candidate and bg? no, bands are disjoint)
    int mine = 0;
    bool kp2[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = 2 * tid + u;
        kp2[u] = i < M && (s_flag[i] & 4);
        mine += kp2[u];
    }
    // exclusive scan of `mine` (0..2) over the workgroup
    int inc = mine;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) wcnt[wave] = inc;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int c = wcnt[w]; if (w < wave) before += c; total += c; }
    int slot = before + inc - mine;
    float* orr = out_rois + (long long)n * num_samples * 4;
    int* olb = out_labels + (long long)n * num_samples;
    float* otg = out_targets + (long long)n * num_samples * 4;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (!kp2[u]) continue;
        const int i = 2 * tid + u;
        const bool is_bg = (s_flag[i] & 2) != 0;
        f32x4_t t = {0.f, 0.f, 0.f, 0.f};
        if (G > 0) t = encode_dev(bx[u], ld_gt(gp + am[u] * 5), coder);
        if (slot < num_samples) {
            *reinterpret_cast<f32x4_t*>(orr + slot * 4ll) = (f32x4_t){bx[u].x1, bx[u].y1, bx[u].x2, bx[u].y2};
            olb[slot] = is_bg ? 0 : lab[u];
            *reinterpret_cast<f32x4_t*>(otg + slot * 4ll) = t;
        }
        ++slot;
    }
    const int filled = min(total, num_samples);
    for (int s = filled + tid; s < num_samples; s += 1024) {
        *reinterpret_cast<f32x4_t*>(orr + s * 4ll) = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        olb[s] = -1;
        *reinterpret_cast<f32x4_t*>(otg + s * 4ll) = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    if (tid == 0) { out_count[n] = filled; atomicAdd(total_count, filled); }
}

// ------------------------------------------------------------------------------------------------------------
// multi-level RoIAlign (layers/common/roi_pool.py:12-78; MegEngine roi_align: average mode, 2x2 samples, aligned)
// ------------------------------------------------------------------------------------------------------------
struct RoiLevels { int pix_off[BD_MAX_SEGS]; int H[BD_MAX_SEGS]; int W[BD_MAX_SEGS]; float scale[BD_MAX_SEGS]; int L; int min_level, max_level; };

// assign_rois (roi_pool.py:12-25): floor(4 + log2(sqrt(area) / 224)) clamped to the pyramid
__device__ __forceinline__ int roi_level(const Box& b, const RoiLevels& lv) {
    const float area = (b.x2 - b.x1) * (b.y2 - b.y1);
    const float v = 4.f + logf(sqrtf(area) / 224.f) / 0.6931471805599453f;
    int l = lv.min_level;
    if (v == v && v > (float)lv.min_level) l = v >= (float)lv.max_level ? lv.max_level : (int)floorf(v);
    return l - lv.min_level;
}

struct Bilinear { int y0, y1, x0, x1; float w00, w01, w10, w11; bool ok; };
__device__ __forceinline__ Bilinear bilinear_setup(float y, float x, int H, int W) {
    Bilinear r;
    r.ok = !(y < -1.f || y > (float)H || x < -1.f || x > (float)W);
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    r.y0 = (int)y; r.x0 = (int)x;
    if (r.y0 >= H - 1) { r.y0 = r.y1 = H - 1; y = (float)r.y0; } else r.y1 = r.y0 + 1;
    if (r.x0 >= W - 1) { r.x0 = r.x1 = W - 1; x = (float)r.x0; } else r.x1 = r.x0 + 1;
    const float ly = y - (float)r.y0, lx = x - (float)r.x0, hy = 1.f - ly, hx = 1.f - lx;
    r.w00 = hy * hx; r.w01 = hy * lx; r.w10 = ly * hx; r.w11 = ly * lx;
    return r;
}

__global__ __launch_bounds__(256) void roi_align_fwd_kernel(const bf16_raw* __restrict__ feat, long long ppi, int C, RoiLevels lv,
                                                            const float* __restrict__ rois, const int* __restrict__ labels,
                                                            int rois_per_img, int PH, int PW, int S,
                                                            bf16_raw* __restrict__ out) {
    const int r = blockIdx.x;
    const int n = r / rois_per_img;
    const int tid = threadIdx.x;
    const int nb = PH * PW;
    bf16_raw* op = out + (long long)r * nb * C;
    if (labels && labels[r] < 0) {
        for (int e = tid * 8; e < nb * C; e += 256 * 8) *reinterpret_cast<u32x4_t*>(op + e) = (u32x4_t){0u, 0u, 0u, 0u};
        return;
    }
    const Box b = ld_box(rois + r * 4ll);
    const int l = roi_level(b, lv);
    const int H = lv.H[l], W = lv.W[l];
    const float sc = lv.scale[l];
    const float sw = b.x1 * sc - 0.5f, sh_ = b.y1 * sc - 0.5f;
    const float rw = (b.x2 * sc - 0.5f) - sw, rh = (b.y2 * sc - 0.5f) - sh_;
    const float bw = rw / (float)PW, bh = rh / (float)PH;
    const bf16_raw* fp = feat + ((long long)n * ppi + lv.pix_off[l]) * C;
    const int cgs = C / 8;
    const float inv = 1.f / (float)(S * S);
    for (int w = tid; w < nb * cgs; w += 256) {
        const int bin = w / cgs, cg = w - bin * cgs;
        const int ph = bin / PW, pw = bin - ph * PW;
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
        for (int iy = 0; iy < S; ++iy) {
            const float y = sh_ + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)S;
            for (int ix = 0; ix < S; ++ix) {
                const float x = sw + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)S;
                const Bilinear bl = bilinear_setup(y, x, H, W);
                if (!bl.ok) continue;
                const u32x4_t v00 = *reinterpret_cast<const u32x4_t*>(fp + ((long long)bl.y0 * W + bl.x0) * C + cg * 8);
                const u32x4_t v01 = *reinterpret_cast<const u32x4_t*>(fp + ((long long)bl.y0 * W + bl.x1) * C + cg * 8);
                const u32x4_t v10 = *reinterpret_cast<const u32x4_t*>(fp + ((long long)bl.y1 * W + bl.x0) * C + cg * 8);
                const u32x4_t v11 = *reinterpret_cast<const u32x4_t*>(fp + ((long long)bl.y1 * W + bl.x1) * C + cg * 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[2 * q] += bl.w00 * bf_lo(v00[q]) + bl.w01 * bf_lo(v01[q]) + bl.w10 * bf_lo(v10[q]) + bl.w11 * bf_lo(v11[q]);
                    acc[2 * q + 1] += bl.w00 * bf_hi(v00[q]) + bl.w01 * bf_hi(v01[q]) + bl.w10 * bf_hi(v10[q]) + bl.w11 * bf_hi(v11[q]);
                }
            }
        }
        u32x4_t o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = pack_bf2(acc[2 * q] * inv, acc[2 * q + 1] * inv);
        *reinterpret_cast<u32x4_t*>(op + (long long)bin * C + cg * 8) = o;
    }
}

// scatter of dL/d(pooled) into the fp32 feature-gradient pyramid; one thread per channel, wave-contiguous atomics.
// All samples of a bin carry the same gradient g/S^2, so their bilinear weights are first merged per target pixel (a
// workgroup-uniform 4x4 coefficient patch anchored at the top-left sample corner): one atomic per distinct pixel of the bin
// (typically 9) instead of one per sample corner (16).  Bins whose samples spread over more than 4 rows / columns (very large
// RoIs) take the direct path.
// (Rounds 2-5 also carried a packed-bf16 variant -- global_atomic_pk_add_bf16 straight into the bf16 pyramid, running bf16 sums -- and a
// separable 7 x 7 form of this scatter; round 5's tiled fixed-order sum, roi_align_bwd_tile_kernel below, replaced both.  This kernel stays
// as the GENERAL form: any pooled size, any number of RoIs per image.)
__global__ __launch_bounds__(256) void roi_align_bwd_kernel(const bf16_raw* __restrict__ gout, long long ppi, int C, RoiLevels lv,
                                                            const float* __restrict__ rois, const int* __restrict__ labels,
                                                            int rois_per_img, int PH, int PW, int S,
                                                            float* __restrict__ gfeat) {
    const int r = blockIdx.x;
    if (labels && labels[r] < 0) return;
    const int n = r / rois_per_img;
    const int nb = PH * PW;
    const Box b = ld_box(rois + r * 4ll);
    const int l = roi_level(b, lv);
    const int H = lv.H[l], W = lv.W[l];
    const float sc = lv.scale[l];
    const float sw = b.x1 * sc - 0.5f, sh_ = b.y1 * sc - 0.5f;
    const float rw = (b.x2 * sc - 0.5f) - sw, rh = (b.y2 * sc - 0.5f) - sh_;
    const float bw = rw / (float)PW, bh = rh / (float)PH;
    float* gp = gfeat + ((long long)n * ppi + lv.pix_off[l]) * C;
    const bf16_raw* go = gout + (long long)r * nb * C;
    const float inv = 1.f / (float)(S * S);
    for (int bin = 0; bin < nb; ++bin) {
        const int ph = bin / PW, pw = bin - ph * PW;
        bool merged = S == 2;
        float coef[4][4];
        int Y0 = 0, X0 = 0;
        if (merged) {
            Bilinear sm[4];
            int ymin = 1 << 30, xmin = 1 << 30, ymax = -1, xmax = -1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int iy = k >> 1, ix = k & 1;
                const float y = sh_ + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)S;
                const float x = sw + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)S;
                sm[k] = bilinear_setup(y, x, H, W);
                if (sm[k].ok) {
                    ymin = min(ymin, sm[k].y0); xmin = min(xmin, sm[k].x0);
                    ymax = max(ymax, sm[k].y1); xmax = max(xmax, sm[k].x1);
                }
            }
            if (ymax < 0) continue;                                   // every sample is outside the map
            merged = (ymax - ymin) < 4 && (xmax - xmin) < 4;
            if (merged) {
                Y0 = ymin; X0 = xmin;
#pragma unroll
                for (int py = 0; py < 4; ++py)
#pragma unroll
                    for (int px = 0; px < 4; ++px) {
                        float cf = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (!sm[k].ok) continue;
                            const int ya = sm[k].y0 - Y0, yb = sm[k].y1 - Y0, xa = sm[k].x0 - X0, xb = sm[k].x1 - X0;
                            if (ya == py && xa == px) cf += sm[k].w00;
                            if (ya == py && xb == px) cf += sm[k].w01;
                            if (yb == py && xa == px) cf += sm[k].w10;
                            if (yb == py && xb == px) cf += sm[k].w11;
                        }
                        coef[py][px] = cf;
                    }
            }
        }
        if (merged) {
            for (int c = threadIdx.x; c < C; c += 256) {
                const float g = bf2f(go[(long long)bin * C + c]) * inv;
#pragma unroll
                for (int py = 0; py < 4; ++py)
#pragma unroll
                    for (int px = 0; px < 4; ++px)
                        if (coef[py][px] != 0.f)                      // workgroup-uniform
                            unsafeAtomicAdd(gp + ((long long)(Y0 + py) * W + X0 + px) * C + c, coef[py][px] * g);
            }
            continue;
        }
        for (int iy = 0; iy < S; ++iy) {
            const float y = sh_ + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)S;
            for (int ix = 0; ix < S; ++ix) {
                const float x = sw + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)S;
                const Bilinear bl = bilinear_setup(y, x, H, W);
                if (!bl.ok) continue;
                for (int c = threadIdx.x; c < C; c += 256) {
                    const float g = bf2f(go[(long long)bin * C + c]) * inv;
                    unsafeAtomicAdd(gp + ((long long)bl.y0 * W + bl.x0) * C + c, bl.w00 * g);
                    unsafeAtomicAdd(gp + ((long long)bl.y0 * W + bl.x1) * C + c, bl.w01 * g);
                    unsafeAtomicAdd(gp + ((long long)bl.y1 * W + bl.x0) * C + c, bl.w10 * g);
                    unsafeAtomicAdd(gp + ((long long)bl.y1 * W + bl.x1) * C + c, bl.w11 * g);
                }
            }
        }
    }
}

// the row (or column) half of bilinear_setup: sample positions and validity exactly as in roi_align_fwd_kernel
struct Lin1 { int i0, i1; float w0, w1; bool ok; };
__device__ __forceinline__ Lin1 linear_setup(float y, int H) {
    Lin1 r;
    r.ok = !(y < -1.f || y > (float)H);
    if (y <= 0.f) y = 0.f;
    r.i0 = (int)y;
    if (r.i0 >= H - 1) { r.i0 = r.i1 = H - 1; y = (float)r.i0; } else r.i1 = r.i0 + 1;
    r.w1 = y - (float)r.i0; r.w0 = 1.f - r.w1;
    return r;
}


// ------------------------------------------------------------------------------------------------------------
// deterministic RoIAlign backward (round 5: the training step's default).  The gradient pyramid is cut into 8x8-pixel tiles; every tile
// gets the list of the RoIs whose sample footprint touches it, IN SLOT ORDER (one thread per tile walks its image's <= 512 footprints:
// count, scan, fill -- no atomics anywhere), and one wave per (tile, 128-channel slice) sums the tile in registers:
//     dF[y][x] += sum_ph A[y][ph] * (sum_pw B[x][pw] * g[ph][pw])          (a bilinear weight factors into a row and a column part)
// with A (8 rows x 7 bins) and B (8 columns x 7 bins) restricted to the tile, two channels per lane as packed fp32 pairs, the 49 pooled
// gradients of the next RoI requested under the arithmetic of the current one.  The result is written ONCE as bf16 -- optionally added to
// what the buffer holds (the RPN head's dL/dP): no fp32 pyramid to clear, no float atomics, no conversion pass, fixed summation order.
// Tiles no RoI touches cost a list lookup (accumulate) or a zero fill.
// ------------------------------------------------------------------------------------------------------------
constexpr int RT = 8;                    // tile side
constexpr int ROI_LIST_MAX = 512;        // RoI slots per image (the footprints of an image sit in LDS)
// tile-list entries reserved per RoI slot = the tiles of the largest level: no RoI can touch more, the lists cannot overflow (typical: ~11)
struct PyrTiles { int pix_off[BD_MAX_SEGS]; int H[BD_MAX_SEGS]; int W[BD_MAX_SEGS]; int tile_start[BD_MAX_SEGS + 1]; int tiles_x[BD_MAX_SEGS]; int L; };

// conservative tile range of the sample footprint of every RoI slot on its level (samples clamp into the map: the range as well):
// foot[r] = {level or -1 (empty slot), tx0 | tx1 << 16, ty0 | ty1 << 16, 0}
__global__ __launch_bounds__(256) void roi_foot_kernel(const float* __restrict__ rois, const int* __restrict__ labels, int total, RoiLevels lv,
                                                       PyrTiles pt, int4* __restrict__ foot) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= total) return;
    int4 f = make_int4(-1, 0, 0, 0);
    if (!(labels && labels[r] < 0)) {
        const Box b = ld_box(rois + r * 4ll);
        const int l = roi_level(b, lv);
        const float sc = lv.scale[l];
        const float sw = b.x1 * sc - 0.5f, sh_ = b.y1 * sc - 0.5f;
        const float ew = b.x2 * sc - 0.5f, eh = b.y2 * sc - 0.5f;
        float fx0 = fminf(sw, ew) - 1.f, fx1 = fmaxf(sw, ew) + 1.f, fy0 = fminf(sh_, eh) - 1.f, fy1 = fmaxf(sh_, eh) + 1.f;
        const float W = (float)pt.W[l], H = (float)pt.H[l];
        if (!(fx0 == fx0 && fx1 == fx1 && fy0 == fy0 && fy1 == fy1)) { fx0 = 0.f; fx1 = W; fy0 = 0.f; fy1 = H; }
        fx0 = fminf(fmaxf(fx0, 0.f), W - 1.f); fx1 = fminf(fmaxf(fx1, 0.f), W - 1.f);
        fy0 = fminf(fmaxf(fy0, 0.f), H - 1.f); fy1 = fminf(fmaxf(fy1, 0.f), H - 1.f);
        f = make_int4(l, ((int)fx0 / RT) | (((int)fx1 / RT) << 16), ((int)fy0 / RT) | (((int)fy1 / RT) << 16), 0);
    }
    foot[r] = f;
}

__device__ __forceinline__ void tile_decode(const PyrTiles& pt, int t, int& la, int& ty, int& tx) {
    la = 0;
    for (int q = 1; q < pt.L; ++q) if (t >= pt.tile_start[q]) la = q;
    const int tt = t - pt.tile_start[la];
    ty = tt / pt.tiles_x[la];
    tx = tt - ty * pt.tiles_x[la];
}

// count (fill == 0) or write (fill == 1) the RoI list of every tile, slots ascending: one WAVE per tile walks its image's footprints 64 at a
// time (ballot + prefix count: the order within the list is the slot order, whatever the hardware does); blockIdx.y = image
__global__ __launch_bounds__(1024) void roi_tile_list_kernel(const int4* __restrict__ foot, int S, PyrTiles pt, int fill, int* __restrict__ tile_cnt,
                                                             const int* __restrict__ tile_off, unsigned short* __restrict__ entries, int cap) {
    __shared__ int4 s_f[ROI_LIST_MAX];
    const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tpi = pt.tile_start[pt.L];
    for (int i = tid; i < S; i += 1024) s_f[i] = foot[(long long)n * S + i];
    __syncthreads();
    const int t = blockIdx.x * 16 + wave;
    if (t >= tpi) return;
    int la, ty, tx;
    tile_decode(pt, t, la, ty, tx);
    const int base = fill ? tile_off[n * tpi + t] : 0;
    int cnt = 0;
    for (int i0 = 0; i0 < S; i0 += 64) {
        const int i = i0 + lane;
        bool hit = false;
        if (i < S) {
            const int4 f = s_f[i];
            hit = f.x == la && tx >= (f.y & 0xffff) && tx <= (f.y >> 16) && ty >= (f.z & 0xffff) && ty <= (f.z >> 16);
        }
        const unsigned long long bal = __ballot(hit);
        if (fill && hit) {
            const int o = base + cnt + __popcll(bal & ((1ull << lane) - 1ull));
            if (o < cap) entries[o] = (unsigned short)i;
        }
        cnt += __popcll(bal);
    }
    if (!fill && lane == 0) tile_cnt[n * tpi + t] = cnt;
}

// exclusive scan of the tile counts in two small launches: every workgroup scans its own 1024 counts (coalesced) and leaves their total;
// the second launch adds the totals of the workgroups in front (at most a few dozen)
__global__ __launch_bounds__(1024) void roi_tile_scan_local_kernel(const int* __restrict__ tile_cnt, int* __restrict__ tile_off, int ntiles,
                                                                   int* __restrict__ chunk_tot) {
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * 1024 + tid;
    const int v = i < ntiles ? tile_cnt[i] : 0;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (i < ntiles) tile_off[i] = before + inc - v;
    if (tid == 1023) chunk_tot[blockIdx.x] = before + inc;
}

__global__ __launch_bounds__(1024) void roi_tile_scan_add_kernel(int* __restrict__ tile_off, int ntiles, const int* __restrict__ chunk_tot) {
    int before = 0;
    for (int b = 0; b < (int)blockIdx.x; ++b) before += chunk_tot[b];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < ntiles) tile_off[i] += before;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) tile_off[ntiles] = before + chunk_tot[blockIdx.x];
}

__global__ __launch_bounds__(64, 2) void roi_align_bwd_tile_kernel(const bf16_raw* __restrict__ gout, long long ppi, int C, RoiLevels lv, PyrTiles pt,
                                                                   const float* __restrict__ rois, int S, int SP, const int* __restrict__ tile_off,
                                                                   const unsigned short* __restrict__ entries, int cap,
                                                                   bf16_raw* __restrict__ gfeat, int accumulate) {
    constexpr int P = 7;
    __shared__ f32x4_t s_par[2][64];               // sw, sh, bw, bh of the RoIs of the current / next chunk of the list
    __shared__ int s_slot[2][64];
    const int lane = threadIdx.x;
    const int slices = (C + 127) / 128;
    int bid = blockIdx.x;
    {   // neighbouring tiles share RoIs: every XCD (blocks go round-robin over the eight) takes a contiguous range of them
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int slice = bid % slices;
    const int tb = bid / slices;                   // n * tiles_per_img + t
    const int tpi = pt.tile_start[pt.L];
    const int n = tb / tpi;
    int la, tyi, txi;
    tile_decode(pt, tb - n * tpi, la, tyi, txi);
    const int ty0 = tyi * RT, tx0 = txi * RT;
    const int H = pt.H[la], W = pt.W[la];
    const int beg = min(tile_off[tb], cap), cnt = min(tile_off[tb + 1], cap) - beg;
    const int c0 = slice * 128 + lane * 2;
    const bool cok = c0 < C;                        // (C is even)
    unsigned int* gp_out = reinterpret_cast<unsigned int*>(gfeat + ((long long)n * ppi + pt.pix_off[la]) * C + (cok ? c0 : 0));
    const int rowp = C / 2;                         // pixel pitch in 32-bit words
    if (cnt == 0) {
        if (!accumulate && cok)
            for (int y = 0; y < RT && ty0 + y < H; ++y)
                for (int x = 0; x < RT && tx0 + x < W; ++x) gp_out[((long long)(ty0 + y) * W + tx0 + x) * rowp] = 0u;
        return;
    }
    const float sc = lv.scale[la];                  // (tiles with a list are on a RoI level)
    const float inv_sp = 1.f / (float)SP;
    // the tile's list and the geometry of its RoIs, 64 at a time: one coalesced read and one box per lane instead of a chain of
    // dependent loads (list entry -> box -> pooled gradients) in front of every RoI
    auto stage = [&](int q0) {
        const int q = q0 + lane;
        if (q < cnt) {
            const int slot = entries[beg + q];
            const Box b = ld_box(rois + ((long long)n * S + slot) * 4);
            const float sw = b.x1 * sc - 0.5f, sh_ = b.y1 * sc - 0.5f;
            s_par[(q0 >> 6) & 1][lane] = (f32x4_t){sw, sh_, ((b.x2 * sc - 0.5f) - sw) / (float)P, ((b.y2 * sc - 0.5f) - sh_) / (float)P};
            s_slot[(q0 >> 6) & 1][lane] = slot;
        }
    };
    f32x2_t acc[RT][RT];
#pragma unroll
    for (int y = 0; y < RT; ++y)
#pragma unroll
        for (int x = 0; x < RT; ++x) acc[y][x] = (f32x2_t){0.f, 0.f};
    stage(0);
    const int iy = lane / P, pb = lane - iy * P;    // lanes 0..55: (tile row / column, bin)
    const unsigned int* gbase = reinterpret_cast<const unsigned int*>(gout + (long long)n * S * (P * P) * C + (cok ? c0 : 0));
    for (int q = 0; q < cnt; ++q) {
        if ((q & 63) == 0) {                         // (the chunk staged 64 RoIs ago becomes visible; the next one goes into the buffer just left)
            __syncthreads();
            if (q + 64 < cnt) stage(q + 64);
        }
        const f32x4_t par = s_par[(q >> 6) & 1][q & 63];
        const int slot = __builtin_amdgcn_readfirstlane(s_slot[(q >> 6) & 1][q & 63]);
        const float sw = par[0], sh_ = par[1], bw = par[2], bh = par[3];
        // A (row weights of the tile's 8 rows x 7 bins) and B (columns): lane y * 7 + ph holds A[y][ph] and B[y][ph].  Which rows / columns
        // carry any weight comes from two ballots (scalar); the weights themselves are read as SCALARS right where they are used
        // (v_readlane into an SGPR pair, two weights per pair, picked by op_sel): 128 VGPRs of sums and 98 of pooled gradients leave no
        // room for 112 weights in registers at two waves per SIMD, and an LDS round trip per row costs more than seven lane reads
        float av = 0.f, bv = 0.f;
        if (lane < RT * P) {
            for (int iq = 0; iq < SP; ++iq) {        // sample positions and validity exactly as in roi_align_fwd_kernel
                const Lin1 ly = linear_setup(sh_ + (float)pb * bh + ((float)iq + 0.5f) * bh / (float)SP, H);
                const Lin1 lx = linear_setup(sw + (float)pb * bw + ((float)iq + 0.5f) * bw / (float)SP, W);
                if (ly.ok) { if (ly.i0 == ty0 + iy) av += ly.w0; if (ly.i1 == ty0 + iy) av += ly.w1; }
                if (lx.ok) { if (lx.i0 == tx0 + iy) bv += lx.w0; if (lx.i1 == tx0 + iy) bv += lx.w1; }
            }
            av *= inv_sp; bv *= inv_sp;
        }
        const unsigned long long amask = __ballot(av != 0.f), bmask = __ballot(bv != 0.f);
        if (amask == 0ull || bmask == 0ull) continue;            // (the footprints of the lists are conservative)
        f32x2_t g[P * P];
        {
            const unsigned int* go = gbase + (long long)slot * (P * P) * rowp;
            unsigned int gpk[P * P];
#pragma unroll
            for (int k = 0; k < P * P; ++k) gpk[k] = go[k * rowp];
#pragma unroll
            for (int k = 0; k < P * P; ++k) g[k] = (f32x2_t){__uint_as_float(gpk[k] << 16), __uint_as_float(gpk[k] & 0xffff0000u)};
        }
        float a[RT][P];                               // (volatile: a lane read the compiler may not repeat at every use -- it would, 8 times)
#pragma unroll
        for (int y = 0; y < RT; ++y)
#pragma unroll
            for (int ph = 0; ph < P; ++ph) asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(a[y][ph]) : "v"(av), "n"(y * P + ph));
#pragma unroll
        for (int x = 0; x < RT; ++x) {
            if (((bmask >> (x * P)) & 0x7full) == 0ull) continue;                   // a pixel column no sample of this RoI touches
            f32x2_t tt[P];
            {
                float bx[P];
#pragma unroll
                for (int pw = 0; pw < P; ++pw) bx[pw] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bv), x * P + pw));
#pragma unroll
                for (int ph = 0; ph < P; ++ph) tt[ph] = g[ph * P] * bx[0];
#pragma unroll
                for (int pw = 1; pw < P; ++pw)                // (explicit FMAs: this file is compiled with -ffp-contract=off)
#pragma unroll
                    for (int ph = 0; ph < P; ++ph) tt[ph] = __builtin_elementwise_fma(g[ph * P + pw], (f32x2_t){bx[pw], bx[pw]}, tt[ph]);
            }
#pragma unroll
            for (int y = 0; y < RT; ++y) {
                if (((amask >> (y * P)) & 0x7full) == 0ull) continue;
#pragma unroll
                for (int ph = 0; ph < P; ++ph) acc[y][x] = __builtin_elementwise_fma(tt[ph], (f32x2_t){a[y][ph], a[y][ph]}, acc[y][x]);
            }
        }
    }
    if (cok && accumulate) {          // every old value requested before the first store (loads do not move across stores that may alias them)
        unsigned int old[RT][RT];
#pragma unroll
        for (int y = 0; y < RT; ++y)
#pragma unroll
            for (int x = 0; x < RT; ++x)
                old[y][x] = (ty0 + y < H && tx0 + x < W) ? gp_out[(long long)((ty0 + y) * W + tx0 + x) * rowp] : 0u;
#pragma unroll
        for (int y = 0; y < RT; ++y)
#pragma unroll
            for (int x = 0; x < RT; ++x) { acc[y][x][0] += __uint_as_float(old[y][x] << 16); acc[y][x][1] += __uint_as_float(old[y][x] & 0xffff0000u); }
    }
    if (cok)
#pragma unroll
        for (int y = 0; y < RT; ++y)
#pragma unroll
            for (int x = 0; x < RT; ++x)
                if (ty0 + y < H && tx0 + x < W) {
                    unsigned int* o = gp_out + (long long)((ty0 + y) * W + tx0 + x) * rowp;
                    *o = (unsigned int)f2bf(acc[y][x][0]) | ((unsigned int)f2bf(acc[y][x][1]) << 16);
                }
}

// ------------------------------------------------------------------------------------------------------------
// FPNP6 (fpn_backbone.py:172-183): max_pool2d(kernel 1, stride 2) == take every other pixel
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void subsample_fwd_kernel(const bf16_raw* __restrict__ src, long long src_ppi, long long src_off,
                                                            int Ws, bf16_raw* __restrict__ dst, long long dst_ppi,
                                                            long long dst_off, int Hd, int Wd, int C, int N) {
    const int cgs = C / 8;
    const long long total = (long long)N * Hd * Wd * cgs;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cg = (int)(i % cgs);
        long long p = i / cgs;
        const int x = (int)(p % Wd); p /= Wd;
        const int y = (int)(p % Hd);
        const int n = (int)(p / Hd);
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(src + ((long long)n * src_ppi + src_off + (long long)(2 * y) * Ws + 2 * x) * C + cg * 8);
        *reinterpret_cast<u32x4_t*>(dst + ((long long)n * dst_ppi + dst_off + (long long)y * Wd + x) * C + cg * 8) = v;
    }
}

// gsrc[n, 2y, 2x, :] += gdst[n, y, x, :]
__global__ __launch_bounds__(256) void subsample_bwd_kernel(const bf16_raw* __restrict__ gdst, long long dst_ppi, long long dst_off,
                                                            int Hd, int Wd, bf16_raw* __restrict__ gsrc, long long src_ppi,
                                                            long long src_off, int Ws, int C, int N) {
    const int cgs = C / 8;
    const long long total = (long long)N * Hd * Wd * cgs;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cg = (int)(i % cgs);
        long long p = i / cgs;
        const int x = (int)(p % Wd); p /= Wd;
        const int y = (int)(p % Hd);
        const int n = (int)(p / Hd);
        const u32x4_t g = *reinterpret_cast<const u32x4_t*>(gdst + ((long long)n * dst_ppi + dst_off + (long long)y * Wd + x) * C + cg * 8);
        bf16_raw* sp = gsrc + ((long long)n * src_ppi + src_off + (long long)(2 * y) * Ws + 2 * x) * C + cg * 8;
        const u32x4_t s = *reinterpret_cast<const u32x4_t*>(sp);
        u32x4_t o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = pack_bf2(bf_lo(s[q]) + bf_lo(g[q]), bf_hi(s[q]) + bf_hi(g[q]));
        *reinterpret_cast<u32x4_t*>(sp) = o;
    }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, bf16_raw* __restrict__ dst, long long n8) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(src + i * 8), b = *reinterpret_cast<const f32x4_t*>(src + i * 8 + 4);
        u32x4_t o = {pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3])};
        *reinterpret_cast<u32x4_t*>(dst + i * 8) = o;
    }
}

// dst = bf16(float(dst) + src): the fp32 RoIAlign-backward pyramid joins a gradient that is already in dst (the RPN head's dL/dP)
__global__ __launch_bounds__(256) void f32_to_bf16_add_kernel(const float* __restrict__ src, bf16_raw* __restrict__ dst, long long n8) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(src + i * 8), b = *reinterpret_cast<const f32x4_t*>(src + i * 8 + 4);
        const u32x4_t d = *reinterpret_cast<const u32x4_t*>(dst + i * 8);
        u32x4_t o = {pack_bf2(a[0] + bf_lo(d[0]), a[1] + bf_hi(d[0])), pack_bf2(a[2] + bf_lo(d[1]), a[3] + bf_hi(d[1])),
                     pack_bf2(b[0] + bf_lo(d[2]), b[1] + bf_hi(d[2])), pack_bf2(b[2] + bf_lo(d[3]), b[3] + bf_hi(d[3]))};
        *reinterpret_cast<u32x4_t*>(dst + i * 8) = o;
    }
}

inline int next_pow2_i(int n) { int p = 1; while (p < n) p <<= 1; return p; }
inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

struct NmsbLayout { size_t sboxes, order, nvalid, mask, total; int words; };
inline NmsbLayout nmsb_layout(int B, int C) {
    NmsbLayout l;
    l.words = (C + 63) / 64;
    size_t o = 0;
    l.sboxes = o; o += align256((size_t)B * C * 16);
    l.order = o; o += align256((size_t)B * C * 4);
    l.nvalid = o; o += align256((size_t)B * 4);
    l.mask = o; o += align256((size_t)B * C * l.words * 8);
    l.total = o;
    return l;
}

int nmsb_run(const float* boxes, const float* scores, const int32_t* idxs, int B, int C, float thr, int max_output, int keep_ld,
             int32_t* keep, int32_t* num_keep, unsigned char* ws, hipStream_t st) {
    const NmsbLayout l = nmsb_layout(B, C);
    float* sboxes = (float*)(ws + l.sboxes);
    int* order = (int*)(ws + l.order);
    int* nvalid = (int*)(ws + l.nvalid);
    unsigned long long* mask = (unsigned long long*)(ws + l.mask);
    const int npow2 = next_pow2_i(C);
    BD_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute((const void*)nmsb_prepare_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NMSB_MAX * 8));
    hipLaunchKernelGGL(nmsb_prepare_kernel, dim3(B), dim3(1024), (size_t)npow2 * 8, st, boxes, scores, idxs, C, npow2, sboxes,
                       order, nvalid);
    const int tiles = (C + 63) / 64;
    hipLaunchKernelGGL(nmsb_mask_kernel, dim3(tiles, l.words, B), dim3(64), 0, st, sboxes, order, nvalid, C, l.words, thr, mask);
    hipLaunchKernelGGL(nmsb_scan_kernel, dim3(B), dim3(64), 0, st, mask, order, nvalid, C, l.words, max_output, keep_ld, keep,
                       num_keep);
    return BD_OK;
}

struct NmslLayout { size_t sboxes, order, nvalid, mask, keep_l, num_l, total; };
inline NmslLayout nmsl_layout(int N, int L, int C, int post_k) {
    NmslLayout l;
    size_t o = 0;
    l.sboxes = o; o += align256((size_t)N * C * 16);
    l.order = o; o += align256((size_t)N * C * 4);
    l.nvalid = o; o += align256((size_t)N * L * 4);
    l.mask = o; o += align256((size_t)N * L * NMSL_CAP * NMSL_WORDS * 8);
    l.keep_l = o; o += align256((size_t)N * L * post_k * 4);
    l.num_l = o; o += align256((size_t)N * L * 4);
    l.total = o;
    return l;
}


struct RpnWs { size_t tk_idx, tk_score, tk_cnt, boxes, scores, levels, keep, nms, total; };
inline RpnWs rpn_ws_layout(int N, int C, int L, int k, int post_k) {
    RpnWs w;
    size_t o = 0;
    w.tk_idx = o; o += align256((size_t)N * L * k * 4);
    w.tk_score = o; o += align256((size_t)N * L * k * 4);
    w.tk_cnt = o; o += align256((size_t)N * L * 4);
    w.boxes = o; o += align256((size_t)N * C * 16);
    w.scores = o; o += align256((size_t)N * C * 4);
    w.levels = o; o += align256((size_t)N * C * 4);
    w.keep = o; o += align256((size_t)N * post_k * 4);
    const size_t joint = nmsb_layout(N, C).total, per_level = nmsl_layout(N, L, C, post_k).total;
    w.nms = o; o += joint > per_level ? joint : per_level;
    w.total = o;
    return w;
}

inline int rpn_capacity(int L, const int32_t* lvl_pixels, int A, int pre_k) {
    int C = 0;
    for (int l = 0; l < L; ++l) C += lvl_pixels[l] * A < pre_k ? lvl_pixels[l] * A : pre_k;
    return C;
}

}  // namespace

extern "C" int bd_segment_topk(const void* scores, int is_bf16, int B, int64_t batch_stride, int A, int ldc, int coff, int nseg,
                               const int32_t* seg_start_host, const int32_t* seg_rows_host, int k, float min_score,
                               int use_min_score, int32_t* out_idx, float* out_score, int32_t* out_cnt, bd_stream_t stream) {
    BD_REQUIRE(scores && seg_start_host && seg_rows_host && out_idx && out_score && out_cnt, "segment_topk: null pointer");
    BD_REQUIRE(B > 0 && nseg > 0 && nseg <= BD_MAX_SEGS && A > 0 && ldc > 0 && coff >= 0, "segment_topk: bad sizes");
    BD_REQUIRE(k > 0 && k <= TOPK_MAX, "segment_topk: k=%d out of range (1..%d)", k, TOPK_MAX);
    TopkSegs segs{};
    segs.nseg = nseg;
    for (int s = 0; s < nseg; ++s) {
        segs.start[s] = seg_start_host[s];
        const long long c = (long long)seg_rows_host[s] * A;
        BD_REQUIRE(c < (1ll << 24), "segment_topk: segment of %lld items is too long", c);
        segs.count[s] = (int)c;
    }
    if (is_bf16)
        hipLaunchKernelGGL(segment_topk_kernel<true>, dim3(nseg, B), dim3(1024), 0, (hipStream_t)stream, scores,
                           (long long)batch_stride, A, ldc, coff, segs, k, min_score, use_min_score, out_idx, out_score, out_cnt);
    else
        hipLaunchKernelGGL(segment_topk_kernel<false>, dim3(nseg, B), dim3(1024), 0, (hipStream_t)stream, scores,
                           (long long)batch_stride, A, ldc, coff, segs, k, min_score, use_min_score, out_idx, out_score, out_cnt);
    BD_CHECK_LAUNCH("bd_segment_topk");
    return BD_OK;
}

extern "C" size_t bd_nms_batched_workspace_bytes(int B, int C) {
    if (B <= 0 || C <= 0) return 256;
    return nmsb_layout(B, C).total;
}

extern "C" int bd_nms_batched(const float* boxes, const float* scores, const int32_t* idxs, int B, int C, float iou_thresh,
                              int max_output, int keep_ld, int32_t* keep, int32_t* num_keep, void* ws, size_t ws_bytes,
                              bd_stream_t stream) {
    BD_REQUIRE(boxes && scores && keep && num_keep && ws, "nms_batched: null pointer");
    BD_REQUIRE(B > 0 && C > 0 && C <= NMSB_MAX, "nms_batched: C=%d out of range (1..%d)", C, NMSB_MAX);
    BD_REQUIRE(keep_ld >= (max_output > 0 ? (max_output < C ? max_output : C) : C), "nms_batched: keep_ld too small");
    if (ws_bytes < bd_nms_batched_workspace_bytes(B, C)) {
        bd_set_error("nms_batched: workspace %zu < %zu bytes", ws_bytes, bd_nms_batched_workspace_bytes(B, C));
        return BD_EWORKSPACE;
    }
    nmsb_run(boxes, scores, idxs, B, C, iou_thresh, max_output, keep_ld, keep, num_keep, (unsigned char*)ws, (hipStream_t)stream);
    BD_CHECK_LAUNCH("bd_nms_batched");
    return BD_OK;
}

extern "C" size_t bd_rpn_proposals_workspace_bytes(int N, int L, const int32_t* lvl_pixels_host, int A, int pre_k, int post_k) {
    if (N <= 0 || L <= 0 || L > BD_MAX_SEGS || !lvl_pixels_host) return 0;
    return rpn_ws_layout(N, rpn_capacity(L, lvl_pixels_host, A, pre_k), L, pre_k, post_k).total;
}

static int rpn_proposals_impl(const void* raw, int ldc, int A, int cls_off, int box_off, int N, int64_t pix_per_img, int L,
                              const int32_t* lvl_pix_off_host, const int32_t* lvl_pixels_host, const float* anchors,
                              const float* im_info, int info_ld, const float* mean4_host, const float* std4_host, int pre_k,
                              float nms_thresh, int post_k, float* rois, int32_t* num_rois, void* ws, size_t ws_bytes,
                              bool nms_per_level, bd_stream_t stream) {
    BD_REQUIRE(raw && lvl_pix_off_host && lvl_pixels_host && anchors && im_info && rois && num_rois && ws, "rpn_proposals: null pointer");
    BD_REQUIRE(N > 0 && L > 0 && L <= BD_MAX_SEGS && A > 0 && pre_k > 0 && pre_k <= TOPK_MAX && post_k > 0, "rpn_proposals: bad sizes");
    BD_REQUIRE(cls_off >= 0 && box_off >= 0 && cls_off + A <= ldc && box_off + 4 * A <= ldc, "rpn_proposals: bad channel layout");
    const int C = rpn_capacity(L, lvl_pixels_host, A, pre_k);
    BD_REQUIRE(C <= NMSB_MAX, "rpn_proposals: %d candidates per image exceed %d", C, NMSB_MAX);
    const RpnWs w = rpn_ws_layout(N, C, L, pre_k, post_k);
    if (ws_bytes < w.total) {
        bd_set_error("rpn_proposals: workspace %zu < %zu bytes", ws_bytes, w.total);
        return BD_EWORKSPACE;
    }
    unsigned char* wb = (unsigned char*)ws;
    hipStream_t st = (hipStream_t)stream;
    int* tk_idx = (int*)(wb + w.tk_idx);
    float* tk_score = (float*)(wb + w.tk_score);
    int* tk_cnt = (int*)(wb + w.tk_cnt);
    float* boxes = (float*)(wb + w.boxes);
    float* scores = (float*)(wb + w.scores);
    int* levels = (int*)(wb + w.levels);
    int* keep = (int*)(wb + w.keep);
    TopkSegs segs{};
    RpnLevels lv{};
    segs.nseg = L; lv.L = L;
    int co = 0;
    for (int l = 0; l < L; ++l) {
        segs.start[l] = lvl_pix_off_host[l];
        const long long c = (long long)lvl_pixels_host[l] * A;
        BD_REQUIRE(c < (1ll << 24), "rpn_proposals: level too large");
        segs.count[l] = (int)c;
        lv.pix_off[l] = lvl_pix_off_host[l];
        lv.cand_off[l] = co;
        co += c < pre_k ? (int)c : pre_k;
    }
    lv.cand_off[L] = co;
    hipLaunchKernelGGL(segment_topk_kernel<true>, dim3(L, N), dim3(1024), 0, st, raw, (long long)pix_per_img * ldc, A, ldc, cls_off,
                       segs, pre_k, 0.f, 0, tk_idx, tk_score, tk_cnt);
    hipLaunchKernelGGL(rpn_decode_kernel, dim3(cdiv(C, 256), N), dim3(256), 0, st, (const bf16_raw*)raw, (long long)pix_per_img, ldc,
                       A, box_off, anchors, lv, pre_k, tk_idx, tk_score, tk_cnt, im_info, info_ld, make_coder(mean4_host, std4_host),
                       C, boxes, scores, levels);
    if (nms_per_level) {
        const NmslLayout nl = nmsl_layout(N, L, C, post_k);
        unsigned char* nb = wb + w.nms;
        float* sboxes = (float*)(nb + nl.sboxes);
        int* order = (int*)(nb + nl.order);
        int* nvalid = (int*)(nb + nl.nvalid);
        unsigned long long* mask = (unsigned long long*)(nb + nl.mask);
        int* keep_l = (int*)(nb + nl.keep_l);
        int* num_l = (int*)(nb + nl.num_l);
        hipLaunchKernelGGL(nmsl_prepare_kernel, dim3(L, N), dim3(1024), 0, st, boxes, scores, lv, C, sboxes, order, nvalid);
        hipLaunchKernelGGL(nmsl_mask_kernel, dim3(NMSL_WORDS, NMSL_WORDS, N * L), dim3(64), 0, st, sboxes, order, nvalid, lv, C, nms_thresh, mask);
        hipLaunchKernelGGL(nmsl_scan_kernel, dim3(N * L), dim3(64), 0, st, mask, order, nvalid, lv, C, post_k, keep_l, num_l);
        hipLaunchKernelGGL(nmsl_merge_kernel, dim3(N), dim3(1024), 0, st, scores, keep_l, num_l, lv, C, post_k, keep, num_rois);
    } else {
        nmsb_run(boxes, scores, levels, N, C, nms_thresh, post_k, post_k, keep, num_rois, wb + w.nms, st);
    }
    hipLaunchKernelGGL(rpn_gather_kernel, dim3(cdiv(post_k, 256), N), dim3(256), 0, st, boxes, keep, num_rois, C, post_k, rois);
    BD_CHECK_LAUNCH("bd_rpn_proposals");
    return BD_OK;
}

extern "C" int bd_rpn_proposals(const void* raw, int ldc, int A, int cls_off, int box_off, int N, int64_t pix_per_img, int L,
                                const int32_t* lvl_pix_off_host, const int32_t* lvl_pixels_host, const float* anchors,
                                const float* im_info, int info_ld, const float* mean4_host, const float* std4_host, int pre_k,
                                float nms_thresh, int post_k, float* rois, int32_t* num_rois, void* ws, size_t ws_bytes,
                                bd_stream_t stream) {
    return rpn_proposals_impl(raw, ldc, A, cls_off, box_off, N, pix_per_img, L, lvl_pix_off_host, lvl_pixels_host, anchors, im_info, info_ld,
                              mean4_host, std4_host, pre_k, nms_thresh, post_k, rois, num_rois, ws, ws_bytes, true, stream);
}

// the batched NMS as ONE problem per image (rounds 1-4) instead of level by level + merge: the same proposals bit for bit (tests)
extern "C" int bd_rpn_proposals_joint(const void* raw, int ldc, int A, int cls_off, int box_off, int N, int64_t pix_per_img, int L,
                                      const int32_t* lvl_pix_off_host, const int32_t* lvl_pixels_host, const float* anchors,
                                      const float* im_info, int info_ld, const float* mean4_host, const float* std4_host, int pre_k,
                                      float nms_thresh, int post_k, float* rois, int32_t* num_rois, void* ws, size_t ws_bytes,
                                      bd_stream_t stream) {
    return rpn_proposals_impl(raw, ldc, A, cls_off, box_off, N, pix_per_img, L, lvl_pix_off_host, lvl_pixels_host, anchors, im_info, info_ld,
                              mean4_host, std4_host, pre_k, nms_thresh, post_k, rois, num_rois, ws, ws_bytes, false, stream);
}

extern "C" int bd_sample_labels(int32_t* labels, const float* keys_pos, const float* keys_neg, int N, int A, int num_pos_max,
                                int num_total, int32_t* num_valid, bd_stream_t stream) {
    BD_REQUIRE(labels && keys_pos && keys_neg && num_valid, "sample_labels: null pointer");
    BD_REQUIRE(N > 0 && A > 0 && num_pos_max >= 0 && num_total >= num_pos_max, "sample_labels: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(num_valid, 0, sizeof(int32_t), st);
    hipLaunchKernelGGL(sample_labels_kernel, dim3(N), dim3(1024), 0, st, labels, keys_pos, keys_neg, A, num_pos_max, num_total,
                       num_valid);
    BD_CHECK_LAUNCH("bd_sample_labels");
    return BD_OK;
}

extern "C" int bd_rcnn_sample_targets(const float* rois, const int32_t* num_rois, int post_k, const float* gt_boxes,
                                      const int32_t* num_gt, int N, int Gmax, const float* keys_fg, const float* keys_bg,
                                      int key_ld, int num_samples, int num_fg_max, float fg_thresh, float bg_thresh_hi,
                                      float bg_thresh_lo, const float* mean4_host, const float* std4_host, float* out_rois,
                                      int32_t* out_labels, float* out_targets, int32_t* out_count, int32_t* total_count,
                                      bd_stream_t stream) {
    BD_REQUIRE(rois && num_rois && gt_boxes && num_gt && keys_fg && keys_bg && out_rois && out_labels && out_targets && out_count &&
               total_count, "rcnn_sample_targets: null pointer");
    BD_REQUIRE(N > 0 && post_k > 0 && Gmax > 0 && post_k + Gmax <= RCNN_CAP && key_ld >= post_k + Gmax,
               "rcnn_sample_targets: post_k + Gmax = %d exceeds %d (or key_ld too small)", post_k + Gmax, RCNN_CAP);
    BD_REQUIRE(num_samples > 0 && num_fg_max >= 0 && num_fg_max <= num_samples, "rcnn_sample_targets: bad sample counts");
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(total_count, 0, sizeof(int32_t), st);
    hipLaunchKernelGGL(rcnn_sample_kernel, dim3(N), dim3(1024), 0, st, rois, num_rois, post_k, gt_boxes, num_gt, Gmax, keys_fg,
                       keys_bg, key_ld, num_samples, num_fg_max, fg_thresh, bg_thresh_hi, bg_thresh_lo,
                       make_coder(mean4_host, std4_host), out_rois, out_labels, out_targets, out_count, total_count);
    BD_CHECK_LAUNCH("bd_rcnn_sample_targets");
    return BD_OK;
}

static int fill_roi_levels(RoiLevels& lv, int L, const int32_t* pix_off, const int32_t* H, const int32_t* W, const int32_t* strides) {
    lv.L = L;
    for (int l = 0; l < L; ++l) {
        lv.pix_off[l] = pix_off[l]; lv.H[l] = H[l]; lv.W[l] = W[l];
        lv.scale[l] = 1.0f / (float)strides[l];
        int lg = 0;
        while ((1 << lg) < strides[l]) ++lg;
        if ((1 << lg) != strides[l]) return -1;
        if (l == 0) lv.min_level = lg;
        lv.max_level = lg;
    }
    return 0;
}

extern "C" int bd_roi_align_fwd(const void* feat, int64_t pix_per_img, int C, int L, const int32_t* lvl_pix_off_host,
                                const int32_t* lvl_h_host, const int32_t* lvl_w_host, const int32_t* strides_host,
                                const float* rois, const int32_t* labels, int R, int rois_per_img, int PH, int PW,
                                int sample_points, void* out, bd_stream_t stream) {
    BD_REQUIRE(feat && lvl_pix_off_host && lvl_h_host && lvl_w_host && strides_host && rois && out, "roi_align_fwd: null pointer");
    BD_REQUIRE(L > 0 && L <= BD_MAX_SEGS && C > 0 && C % 8 == 0 && PH > 0 && PW > 0 && sample_points > 0 && rois_per_img > 0,
               "roi_align_fwd: bad sizes");
    if (R == 0) return BD_OK;
    RoiLevels lv{};
    BD_REQUIRE(fill_roi_levels(lv, L, lvl_pix_off_host, lvl_h_host, lvl_w_host, strides_host) == 0, "roi_align_fwd: strides must be powers of two");
    hipLaunchKernelGGL(roi_align_fwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)feat, (long long)pix_per_img,
                       C, lv, rois, labels, rois_per_img, PH, PW, sample_points, (bf16_raw*)out);
    BD_CHECK_LAUNCH("bd_roi_align_fwd");
    return BD_OK;
}

extern "C" int bd_roi_align_bwd(const void* gout, int64_t pix_per_img, int C, int L, const int32_t* lvl_pix_off_host,
                                const int32_t* lvl_h_host, const int32_t* lvl_w_host, const int32_t* strides_host,
                                const float* rois, const int32_t* labels, int R, int rois_per_img, int PH, int PW,
                                int sample_points, float* gfeat, bd_stream_t stream) {
    BD_REQUIRE(gout && lvl_pix_off_host && lvl_h_host && lvl_w_host && strides_host && rois && gfeat, "roi_align_bwd: null pointer");
    BD_REQUIRE(L > 0 && L <= BD_MAX_SEGS && C > 0 && PH > 0 && PW > 0 && sample_points > 0 && rois_per_img > 0, "roi_align_bwd: bad sizes");
    if (R == 0) return BD_OK;
    RoiLevels lv{};
    BD_REQUIRE(fill_roi_levels(lv, L, lvl_pix_off_host, lvl_h_host, lvl_w_host, strides_host) == 0, "roi_align_bwd: strides must be powers of two");
    hipLaunchKernelGGL(roi_align_bwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)gout, (long long)pix_per_img,
                       C, lv, rois, labels, rois_per_img, PH, PW, sample_points, gfeat);
    BD_CHECK_LAUNCH("bd_roi_align_bwd");
    return BD_OK;
}

extern "C" int bd_subsample2x_fwd(const void* src, int64_t src_pix_per_img, int64_t src_off, int Hs, int Ws, void* dst,
                                  int64_t dst_pix_per_img, int64_t dst_off, int C, int N, bd_stream_t stream) {
    BD_REQUIRE(src && dst && Hs > 0 && Ws > 0 && C > 0 && C % 8 == 0 && N > 0, "subsample2x_fwd: bad arguments");
    const int Hd = (Hs - 1) / 2 + 1, Wd = (Ws - 1) / 2 + 1;
    const long long total = (long long)N * Hd * Wd * (C / 8);
    long long g = cdiv64(total, 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(subsample_fwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)src,
                       (long long)src_pix_per_img, (long long)src_off, Ws, (bf16_raw*)dst, (long long)dst_pix_per_img,
                       (long long)dst_off, Hd, Wd, C, N);
    BD_CHECK_LAUNCH("bd_subsample2x_fwd");
    return BD_OK;
}

extern "C" int bd_subsample2x_bwd_add(const void* gdst, int64_t dst_pix_per_img, int64_t dst_off, void* gsrc,
                                      int64_t src_pix_per_img, int64_t src_off, int Hs, int Ws, int C, int N, bd_stream_t stream) {
    BD_REQUIRE(gdst && gsrc && Hs > 0 && Ws > 0 && C > 0 && C % 8 == 0 && N > 0, "subsample2x_bwd_add: bad arguments");
    const int Hd = (Hs - 1) / 2 + 1, Wd = (Ws - 1) / 2 + 1;
    const long long total = (long long)N * Hd * Wd * (C / 8);
    long long g = cdiv64(total, 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(subsample_bwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)gdst,
                       (long long)dst_pix_per_img, (long long)dst_off, Hd, Wd, (bf16_raw*)gsrc, (long long)src_pix_per_img,
                       (long long)src_off, Ws, C, N);
    BD_CHECK_LAUNCH("bd_subsample2x_bwd_add");
    return BD_OK;
}

extern "C" int bd_f32_to_bf16(const float* src, void* dst, int64_t n, bd_stream_t stream) {
    BD_REQUIRE(src && dst && n >= 0 && n % 8 == 0, "f32_to_bf16: n must be a multiple of 8");
    if (n == 0) return BD_OK;
    long long g = cdiv64(n / 8, 256);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, src, (bf16_raw*)dst, (long long)(n / 8));
    BD_CHECK_LAUNCH("bd_f32_to_bf16");
    return BD_OK;
}

extern "C" int bd_f32_to_bf16_add(const float* src, void* dst, int64_t n, bd_stream_t stream) {
    BD_REQUIRE(src && dst && n >= 0 && n % 8 == 0, "f32_to_bf16_add: n must be a multiple of 8");
    if (n == 0) return BD_OK;
    long long g = cdiv64(n / 8, 256);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(f32_to_bf16_add_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, src, (bf16_raw*)dst, (long long)(n / 8));
    BD_CHECK_LAUNCH("bd_f32_to_bf16_add");
    return BD_OK;
}

static int roi_tiles_per_img(int L_all, const int32_t* H, const int32_t* W) {
    int ts = 0;
    for (int l = 0; l < L_all; ++l) ts += cdiv(W[l], RT) * cdiv(H[l], RT);
    return ts;
}
static int roi_tiles_max_level(int L_all, const int32_t* H, const int32_t* W) {
    int m = 1;
    for (int l = 0; l < L_all; ++l) m = std::max(m, cdiv(W[l], RT) * cdiv(H[l], RT));
    return m;
}

// workspace: tile counts | tile offsets (+1) | footprints (int4 per RoI slot) | tile lists (u16 slots) | totals of the 1024-tile scan chunks
extern "C" size_t bd_roi_align_bwd_bf16_workspace_bytes(int N, int L_all, const int32_t* lvl_h_host, const int32_t* lvl_w_host,
                                                        int rois_per_img) {
    if (N <= 0 || L_all <= 0 || L_all > BD_MAX_SEGS || rois_per_img <= 0 || !lvl_h_host || !lvl_w_host) return 256;
    const size_t ntiles = (size_t)N * roi_tiles_per_img(L_all, lvl_h_host, lvl_w_host);
    return 2 * align256((ntiles + 1) * 4) + align256((size_t)N * rois_per_img * 16) + align256((size_t)N * rois_per_img * roi_tiles_max_level(L_all, lvl_h_host, lvl_w_host) * 2) +
           align256((ntiles / 1024 + 1) * 4) + 256;
}

extern "C" int bd_roi_align_bwd_bf16(const void* gout, int64_t pix_per_img, int C, int L, int L_all, const int32_t* lvl_pix_off_host,
                                     const int32_t* lvl_h_host, const int32_t* lvl_w_host, const int32_t* strides_host,
                                     const float* rois, const int32_t* labels, int N, int rois_per_img, int PH, int PW,
                                     int sample_points, void* gfeat, int accumulate, void* ws, size_t ws_bytes, bd_stream_t stream) {
    BD_REQUIRE(gout && lvl_pix_off_host && lvl_h_host && lvl_w_host && strides_host && rois && gfeat && ws, "roi_align_bwd_bf16: null pointer");
    BD_REQUIRE(L > 0 && L <= L_all && L_all <= BD_MAX_SEGS && C > 0 && C % 2 == 0 && sample_points > 0 && rois_per_img > 0 && N > 0,
               "roi_align_bwd_bf16: bad sizes");
    BD_REQUIRE(PH == 7 && PW == 7, "roi_align_bwd_bf16: the pooled size is 7 x 7 (got %d x %d)", PH, PW);
    BD_REQUIRE(rois_per_img <= ROI_LIST_MAX, "roi_align_bwd_bf16: %d RoIs per image exceed %d", rois_per_img, ROI_LIST_MAX);
    if (ws_bytes < bd_roi_align_bwd_bf16_workspace_bytes(N, L_all, lvl_h_host, lvl_w_host, rois_per_img)) {
        bd_set_error("roi_align_bwd_bf16: workspace %zu < %zu bytes", ws_bytes,
                     bd_roi_align_bwd_bf16_workspace_bytes(N, L_all, lvl_h_host, lvl_w_host, rois_per_img));
        return BD_EWORKSPACE;
    }
    RoiLevels lv{};
    BD_REQUIRE(fill_roi_levels(lv, L, lvl_pix_off_host, lvl_h_host, lvl_w_host, strides_host) == 0, "roi_align_bwd_bf16: strides must be powers of two");
    PyrTiles pt{};
    pt.L = L_all;
    int ts = 0;
    for (int l = 0; l < L_all; ++l) {
        pt.pix_off[l] = lvl_pix_off_host[l]; pt.H[l] = lvl_h_host[l]; pt.W[l] = lvl_w_host[l];
        pt.tiles_x[l] = cdiv(lvl_w_host[l], RT);
        // (tile coordinates are packed two to a SIGNED int and unpacked with an arithmetic shift: 15 bits each)
        BD_REQUIRE(pt.tiles_x[l] < 32768 && cdiv(lvl_h_host[l], RT) < 32768, "roi_align_bwd_bf16: level too large");
        pt.tile_start[l] = ts;
        ts += pt.tiles_x[l] * cdiv(lvl_h_host[l], RT);
    }
    pt.tile_start[L_all] = ts;
    const int ntiles = N * ts;
    const int total = N * rois_per_img;
    unsigned char* wb = (unsigned char*)ws;
    int* tile_cnt = (int*)wb;
    int* tile_off = (int*)(wb + align256(((size_t)ntiles + 1) * 4));
    int4* foot = (int4*)(wb + 2 * align256(((size_t)ntiles + 1) * 4));
    unsigned short* entries = (unsigned short*)((unsigned char*)foot + align256((size_t)total * 16));
    const long long cap_ll = (long long)total * roi_tiles_max_level(L_all, lvl_h_host, lvl_w_host);
    BD_REQUIRE(cap_ll < 0x7fffffffll, "roi_align_bwd_bf16: tile lists too large");
    const int cap = (int)cap_ll;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(roi_foot_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, rois, labels, total, lv, pt, foot);
    hipLaunchKernelGGL(roi_tile_list_kernel, dim3(cdiv(ts, 16), N), dim3(1024), 0, st, (const int4*)foot, rois_per_img, pt, 0, tile_cnt,
                       (const int*)tile_off, entries, cap);
    const int chunks = cdiv(ntiles, 1024);
    int* chunk_tot = (int*)((unsigned char*)entries + align256((size_t)cap * 2));      // (behind the lists)
    hipLaunchKernelGGL(roi_tile_scan_local_kernel, dim3(chunks), dim3(1024), 0, st, (const int*)tile_cnt, tile_off, ntiles, chunk_tot);
    hipLaunchKernelGGL(roi_tile_scan_add_kernel, dim3(chunks), dim3(1024), 0, st, tile_off, ntiles, (const int*)chunk_tot);
    hipLaunchKernelGGL(roi_tile_list_kernel, dim3(cdiv(ts, 16), N), dim3(1024), 0, st, (const int4*)foot, rois_per_img, pt, 1, tile_cnt,
                       (const int*)tile_off, entries, cap);
    hipLaunchKernelGGL(roi_align_bwd_tile_kernel, dim3(ntiles * cdiv(C, 128)), dim3(64), 0, st, (const bf16_raw*)gout, (long long)pix_per_img,
                       C, lv, pt, rois, rois_per_img, sample_points, (const int*)tile_off, (const unsigned short*)entries, cap,
                       (bf16_raw*)gfeat, accumulate);
    BD_CHECK_LAUNCH("bd_roi_align_bwd_bf16");
    return BD_OK;
}
