// Block-wide selection helpers shared by the selection kernels (rcnn_ops.hip, freeanchor.hip): radix select of the k largest
// 32-bit keys of a segment by one workgroup, and the ordered rank of a flag within a 1024-thread workgroup.
#pragma once
#include "box_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------------------
// radix select: the k largest 32-bit keys among the valid items of one segment, one workgroup
// ------------------------------------------------------------------------------------------------------------
struct SelResult { unsigned int T; int need_eq; int take_all; int n_valid; };

__device__ __forceinline__ unsigned int f32_asc_key(float f) {
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_from_asc_key(unsigned int k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// After the call: items with key > T are all selected, and the first `need_eq` items (index order) with key == T.
// take_all: fewer than k valid items exist, all of them are selected.  hist: 256 words of LDS, sh: 8 ints of LDS.
template <class KeyFn>
__device__ SelResult radix_select_largest(int n, int k, int npass, KeyFn key, unsigned int* hist, int* sh) {
    const int tid = threadIdx.x;
    unsigned int prefix = 0, mask = 0;
    int remaining = k;
    SelResult r{0u, 0, 0, 0};
    for (int pass = 0; pass < npass; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int b = tid; b < 256; b += blockDim.x) hist[b] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += blockDim.x) {
            bool valid;
            const unsigned int kk = key(i, valid);
            if (valid && (kk & mask) == prefix) atomicAdd(&hist[(kk >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {
            unsigned int c[4], s = 0;      // lane l owns bins 255-4l .. 252-4l (descending key order)
#pragma unroll
            for (int j = 0; j < 4; ++j) { c[j] = hist[255 - 4 * tid - j]; s += c[j]; }
            unsigned int inc = s;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned int t = __shfl_up(inc, o, 64);
                if (tid >= o) inc += t;
            }
            const unsigned int total = __shfl(inc, 63, 64);
            if (pass == 0 && tid == 0) sh[3] = (int)total;
            if (total < (unsigned int)remaining) {
                if (tid == 0) sh[0] = -1;
            } else {
                const unsigned long long bal = __ballot(inc >= (unsigned int)remaining);
                const int first = __ffsll((long long)bal) - 1;
                if (tid == first) {
                    unsigned int before = inc - s;
                    int j = 0;
                    while (before + c[j] < (unsigned int)remaining) { before += c[j]; ++j; }
                    sh[0] = 255 - 4 * tid - j;
                    sh[1] = remaining - (int)before;
                }
            }
        }
        __syncthreads();
        const int bin = sh[0];
        r.n_valid = sh[3];
        if (bin < 0) { r.take_all = 1; __syncthreads(); return r; }
        remaining = sh[1];
        prefix |= (unsigned int)bin << shift;
        mask |= 255u << shift;
        __syncthreads();
    }
    r.T = prefix;
    r.need_eq = remaining;
    return r;
}

// Exclusive rank of this thread's flag among all set flags of the workgroup in thread order, plus the workgroup total.
// wcnt: 16 ints of LDS.  Two barriers.
__device__ __forceinline__ int block_rank_1024(bool flag, int* wcnt, int& total) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned long long bal = __ballot(flag);
    __syncthreads();                       // previous readers of wcnt are done
    if (lane == 0) wcnt[wave] = __popcll(bal);
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int c = wcnt[w];
        if (w < wave) before += c;
        tot += c;
    }
    total = tot;
    return before + __popcll(bal & ((1ull << lane) - 1ull));
}

}  // namespace
