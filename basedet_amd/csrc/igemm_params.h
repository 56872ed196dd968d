// Shared parameter block of the implicit-GEMM convolution kernels (conv_igemm.hip).
#pragma once
#include "common.h"

namespace igemm {

constexpr int TILE_C = 128;   // output channels per workgroup
constexpr int TILE_P = 128;   // pixels per workgroup
constexpr int MAX_SUB = 8;

struct SubSeg {
    int m_start;             // first GEMM column of this sub-segment
    int Hs, Ws;              // enumeration extent per image
    int y0, x0, step;        // dst pixel = (y0 + step*yy, x0 + step*xx)
    int Wd, dst_off;         // dst row pitch (pixels) and pixel offset inside one image
    int Hsrc, Wsrc, src_off; // src geometry
    float inv_per_img, inv_ws; // reciprocals of Hs*Ws and Ws for the float-based divmod (valid for m < 2^24)
};

__device__ __forceinline__ void fast_divmod(int n, int d, float inv, int& q, int& r) {
    q = (int)((float)n * inv);
    r = n - q * d;
    if (r < 0) { --q; r += d; }
    else if (r >= d) { ++q; r -= d; }
}

struct IgemmParams {
    const bf16_raw* src;
    const bf16_raw* w;
    const float* bias;
    const bf16_raw* add;
    const bf16_raw* mask;
    bf16_raw* dst;
    int CK, CO, R, S, stride, pad, mode, flags;
    int M, nsub;
    int src_pix_per_img, dst_pix_per_img;
    int m_tiles, n_tiles;
    int linear_dst;   // 1: destination pixel index == GEMM column m (single dense level, unit step)
    int linear_src;   // 1: additionally 1x1 / stride 1 with the same source geometry: source pixel index == m
    unsigned src_bytes, w_bytes;   // sizes of the source tensor and the packed weights when both are < 2 GB (buffer-load staging), else 0
    SubSeg sub[MAX_SUB];
};


// decode GEMM column m -> destination pixel index (+ optional source-side info)
__device__ __forceinline__ int find_sub(const IgemmParams& p, int m) {
    int s = 0;
#pragma unroll
    for (int k = 1; k < MAX_SUB; ++k)
        if (k < p.nsub && m >= p.sub[k].m_start) s = k;
    return s;
}

}  // namespace igemm
