// Device helpers shared by the box-operator translation units (boxops.hip, rcnn_ops.hip).  Both are compiled with
// -ffp-contract=off so that IoU / encode / decode values are bit-identical to the float32 numpy oracle.
#pragma once
#include "common.h"

namespace {

struct Box { float x1, y1, x2, y2; };

__device__ __forceinline__ Box ld_box(const float* p) {
    const f32x4_t v = *reinterpret_cast<const f32x4_t*>(p);
    return Box{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ float box_area(const Box& b) { return (b.x2 - b.x1) * (b.y2 - b.y1); }
__device__ __forceinline__ float box_inter(const Box& a, const Box& b) {
    const float iw = fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1);
    const float ih = fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1);
    return fmaxf(iw, 0.f) * fmaxf(ih, 0.f);
}
// op_patch.py:33-76: inter / (area1 + area2 - inter), max(., 0)   (fmaxf maps NaN -> 0)
__device__ __forceinline__ float box_iou_dev(const Box& a, float area_a, const Box& b, float area_b) {
    const float inter = box_inter(a, b);
    const float uni = (area_a + area_b) - inter;
    return fmaxf(inter / uni, 0.f);
}

struct Coder { float m0, m1, m2, m3, s0, s1, s2, s3; };

__device__ __forceinline__ f32x4_t encode_dev(const Box& a, const Box& g, const Coder& c) {
    const float aw = a.x2 - a.x1, ah = a.y2 - a.y1;
    const float acx = a.x1 + 0.5f * aw, acy = a.y1 + 0.5f * ah;
    const float gw = g.x2 - g.x1, gh = g.y2 - g.y1;
    const float gcx = g.x1 + 0.5f * gw, gcy = g.y1 + 0.5f * gh;
    f32x4_t t;
    t[0] = ((gcx - acx) / aw - c.m0) / c.s0;
    t[1] = ((gcy - acy) / ah - c.m1) / c.s1;
    t[2] = (logf(gw / aw) - c.m2) / c.s2;
    t[3] = (logf(gh / ah) - c.m3) / c.s3;
    return t;
}

__device__ __forceinline__ f32x4_t decode_dev(const Box& a, const f32x4_t d, const Coder& c) {
    const float d0 = d[0] * c.s0 + c.m0, d1 = d[1] * c.s1 + c.m1, d2 = d[2] * c.s2 + c.m2, d3 = d[3] * c.s3 + c.m3;
    const float aw = a.x2 - a.x1, ah = a.y2 - a.y1;
    const float acx = a.x1 + 0.5f * aw, acy = a.y1 + 0.5f * ah;
    const float cx = acx + d0 * aw, cy = acy + d1 * ah;
    const float w = aw * expf(d2), h = ah * expf(d3);
    f32x4_t o;
    o[0] = cx - 0.5f * w; o[1] = cy - 0.5f * h; o[2] = cx + 0.5f * w; o[3] = cy + 0.5f * h;
    return o;
}

__device__ __forceinline__ Box ld_gt(const float* p) { return Box{p[0], p[1], p[2], p[3]}; }

__device__ __forceinline__ unsigned int float_desc_key(float f) {
    unsigned int u = __float_as_uint(f);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending-orderable
    return ~u;                                         // descending score
}

inline Coder make_coder(const float* mean4, const float* std4) {
    Coder c{0, 0, 0, 0, 1, 1, 1, 1};
    if (mean4) { c.m0 = mean4[0]; c.m1 = mean4[1]; c.m2 = mean4[2]; c.m3 = mean4[3]; }
    if (std4) { c.s0 = std4[0]; c.s1 = std4[1]; c.s2 = std4[2]; c.s3 = std4[3]; }
    return c;
}


}  // namespace
