// Backward of a THIN 1x1 prediction layer (<= 16 output channels over every pixel of the pyramid: Faster R-CNN's RPN objectness + box
// deltas, rpn.py:60-68, 3 + 12 channels padded to 16) in ONE pass over its input.
//
// The generic kernels treat such a layer as three GEMMs with a 16-wide side: forward 0.23 ms, data gradient 0.45 ms, weight gradient
// 0.60 ms at C4's sizes (1.43 M pixels x 256 channels) for a layer whose whole traffic is one read of the activations t (733 MB), one
// write of dL/dt and 46 MB of prediction gradients.  This kernel reads t once and produces
//     dx[p][c] = (t[p][c] > 0) * sum_o g[p][o] * W[o][c]          the data gradient, gated by the ReLU that produced t
//     dW[o][c] = sum_p g[p][o] * t[p][c],   db[o] = sum_p g[p][o]  the weight / bias gradients
// from it: a workgroup of 8 waves walks groups of 16 pixels; wave m owns channels 32 m .. 32 m + 31, lane (p = lane % 16, q = lane / 16)
// holds the 8 channels 32 m + 8 q .. + 7 of pixel p -- ONE 16-byte load, which is at once
//   * the mask and the destination layout of the data gradient: two v_mfma_f32_16x16x16_bf16 (rows = channels, K = the 16 outputs,
//     columns = pixels) with the rows permuted so that lane (p, q) receives exactly its 8 channels;
//   * the t operand of the weight gradient on the vector pipe: 16 outputs x 8 channels = 128 fp32 sums per lane (64 v_pk_fma_f32 per
//     group), reduced over the 16 pixel lanes once at the end of the kernel.
// Loads run two groups ahead in registers.  The per-workgroup partial sums go to a slab and are added in workgroup order by a second
// kernel: no atomics, bitwise reproducible.
#include "common.h"

namespace {

constexpr int TO = 16;          // output channels (padded)
constexpr int TW = 8;           // waves per workgroup = 32-channel blocks of the input
constexpr int TCIN = TW * 32;   // 256

struct ThinParams {
    const bf16_raw* x;          // [M][256] the layer's input (a ReLU output: the gate of dx)
    const bf16_raw* g;          // [M][16]
    const float* w;             // [16][256] fp32 master weights
    bf16_raw* dx;               // [M][256]
    float* slab;                // [grid][16][256] weight-gradient partials, then [grid][16] bias partials
    long long M;
    int groups;                 // ceil(M / 16)
};

struct ThinStage { u32x4_t t; u32x4_t d0, d1; u32x2_t b; };

__global__ __launch_bounds__(TW * 64, 1) void conv1x1_thin_bwd_kernel(const ThinParams p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, q = lane >> 4;
    const int c0 = wave * 32 + q * 8;
    // A operands: block h, row r = lane % 16 -> channel wave * 32 + 8 * (r / 4) + 4 h + r % 4 (so that the D rows 4 q + i of the two blocks are
    // channels c0 + i and c0 + 4 + i); K index 4 q + i = output channel
    s16x4_t a[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ch = wave * 32 + 8 * (px >> 2) + 4 * h + (px & 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[h][i] = (short)f2bf(p.w[(4 * q + i) * TCIN + ch]);
    }
    f32x2_t acc[TO][4];
#pragma unroll
    for (int o = 0; o < TO; ++o)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[o][c] = (f32x2_t){0.f, 0.f};
    float sb[TO];
#pragma unroll
    for (int o = 0; o < TO; ++o) sb[o] = 0.f;

    auto load = [&](ThinStage& s, int grp) {
        long long row = (long long)grp * 16 + px;
        const bool ok = grp < p.groups && row < p.M;
        if (!ok) row = 0;                                            // (any valid row: its contribution is zeroed below)
        s.t = *reinterpret_cast<const u32x4_t*>(p.x + row * TCIN + c0);
        const u32x4_t* gr = reinterpret_cast<const u32x4_t*>(p.g + row * TO);
        s.d0 = gr[0]; s.d1 = gr[1];
        s.b = *reinterpret_cast<const u32x2_t*>(p.g + row * TO + 4 * q);
        if (!ok) { s.d0 = (u32x4_t){0u, 0u, 0u, 0u}; s.d1 = s.d0; s.b = (u32x2_t){0u, 0u}; }
    };
    auto compute = [&](const ThinStage& s, int grp) {
        if (grp >= p.groups) return;
        const long long row = (long long)grp * 16 + px;
        // data gradient
        const s16x4_t b = __builtin_bit_cast(s16x4_t, s.b);
        const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
        const f32x4_t r0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[0], b, z, 0, 0, 0);
        const f32x4_t r1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[1], b, z, 0, 0, 0);
        u32x4_t out;
        out[0] = pack_bf2(r0[0], r0[1]); out[1] = pack_bf2(r0[2], r0[3]); out[2] = pack_bf2(r1[0], r1[1]); out[3] = pack_bf2(r1[2], r1[3]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {                                // gate: t is a ReLU output, a channel passes where its bits are non-zero
            const unsigned w = s.t[k];
            const unsigned m = ((w & 0x7fffu) ? 0xffffu : 0u) | ((w & 0x7fff0000u) ? 0xffff0000u : 0u);
            out[k] &= m;
        }
        if (row < p.M) *reinterpret_cast<u32x4_t*>(p.dx + row * TCIN + c0) = out;
        // weight / bias gradient
        f32x2_t t2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t2[k] = (f32x2_t){bf_lo(s.t[k]), bf_hi(s.t[k])};
        float d[TO];
#pragma unroll
        for (int k = 0; k < 4; ++k) { d[2 * k] = bf_lo(s.d0[k]); d[2 * k + 1] = bf_hi(s.d0[k]); d[8 + 2 * k] = bf_lo(s.d1[k]); d[9 + 2 * k] = bf_hi(s.d1[k]); }
#pragma unroll
        for (int o = 0; o < TO; ++o) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[o][c] = __builtin_elementwise_fma(t2[c], (f32x2_t){d[o], d[o]}, acc[o][c]);
            if (wave == 0) sb[o] += d[o];
        }
    };

    ThinStage s0, s1, s2;
    const int G = gridDim.x;
    int grp = blockIdx.x;
    load(s0, grp); load(s1, grp + G);
    for (; grp < p.groups; grp += 3 * G) {
        load(s2, grp + 2 * G); compute(s0, grp);
        load(s0, grp + 3 * G); compute(s1, grp + G);
        load(s1, grp + 4 * G); compute(s2, grp + 2 * G);
    }
    // sum over the 16 pixel lanes of a row group; lane p == 0 of each q writes 16 x 8 values
    float* slab = p.slab + (long long)blockIdx.x * TO * TCIN;
#pragma unroll
    for (int o = 0; o < TO; ++o)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x2_t v = acc[o][c];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { v[0] += __shfl_xor(v[0], m, 64); v[1] += __shfl_xor(v[1], m, 64); }
            if (px == 0) *reinterpret_cast<f32x2_t*>(slab + o * TCIN + c0 + 2 * c) = v;
        }
    if (wave == 0) {
        float* sbias = p.slab + (long long)gridDim.x * TO * TCIN + (long long)blockIdx.x * TO;
#pragma unroll
        for (int o = 0; o < TO; ++o) {
            float v = q == 0 ? sb[o] : 0.f;                          // (the four q lanes of a pixel saw the same gradient row)
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m, 64);
            if (lane == 0) sbias[o] = v;
        }
    }
}

// dW[o][c] = sum over the workgroups' partials: eight lanes per element take every eighth workgroup each (in index order), then the eight
// partial sums are added in lane order -- a fixed tree, whatever the grid; db likewise
__global__ __launch_bounds__(256) void conv1x1_thin_reduce_kernel(const float* __restrict__ slab, int grid, float* __restrict__ dw,
                                                                  float* __restrict__ dbias, int cout_real) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int e = t >> 3, part = t & 7;
    float s = 0.f;
    if (e < TO * TCIN) {
        for (int b = part; b < grid; b += 8) s += slab[(long long)b * TO * TCIN + e];
    } else if (e < TO * TCIN + TO) {
        const float* sb = slab + (long long)grid * TO * TCIN;
        for (int b = part; b < grid; b += 8) s += sb[(long long)b * TO + (e - TO * TCIN)];
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) s += __shfl_xor(s, m, 64);
    if (part != 0) return;
    if (e < TO * TCIN) dw[e] = e / TCIN < cout_real ? s : 0.f;
    else if (e < TO * TCIN + TO && dbias) dbias[e - TO * TCIN] = e - TO * TCIN < cout_real ? s : 0.f;
}

// Forward of the same layer: y[p][o] = sum_c x[p][c] W[o][c] + bias[o].  One wave per group of 16 pixels and K step of 32 channels:
// v_mfma_f32_16x16x32_bf16 with the weights on the rows (all eight K steps of W in 32 registers) and the pixels on the columns -- the B
// operand of lane (p, q) is the 16 bytes x[p][32 k + 8 q ..], loaded straight from memory, one group ahead -- so that every lane ends up
// with 4 consecutive output channels of its pixel (an 8-byte store; a group's 16 x 32 bytes are contiguous).
__global__ __launch_bounds__(256) void conv1x1_thin_fwd_kernel(const bf16_raw* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                               bf16_raw* __restrict__ y, long long M, int groups) {
    const int lane = threadIdx.x & 63, px = lane & 15, q = lane >> 4;
    bf16x8_t a[TW];
#pragma unroll
    for (int k = 0; k < TW; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[k][i] = (__bf16)w[px * TCIN + 32 * k + 8 * q + i];
    f32x4_t b4 = {0.f, 0.f, 0.f, 0.f};
    if (bias) b4 = (f32x4_t){bias[4 * q], bias[4 * q + 1], bias[4 * q + 2], bias[4 * q + 3]};
    const int nw = gridDim.x * 4;
    int grp = blockIdx.x * 4 + (threadIdx.x >> 6);
    u32x4_t cur[TW], nxt[TW];
    auto load = [&](u32x4_t (&v)[TW], int g) {
        long long row = (long long)g * 16 + px;
        if (g >= groups || row >= M) row = 0;
        const u32x4_t* src = reinterpret_cast<const u32x4_t*>(x + row * TCIN + 8 * q);
#pragma unroll
        for (int k = 0; k < TW; ++k) v[k] = src[4 * k];
    };
    load(cur, grp);
    for (; grp < groups; grp += nw) {
        load(nxt, grp + nw);
        f32x4_t acc = b4;
#pragma unroll
        for (int k = 0; k < TW; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k], __builtin_bit_cast(bf16x8_t, cur[k]), acc, 0, 0, 0);
        const long long row = (long long)grp * 16 + px;
        if (row < M) *reinterpret_cast<u32x2_t*>(y + row * TO + 4 * q) = (u32x2_t){pack_bf2(acc[0], acc[1]), pack_bf2(acc[2], acc[3])};
#pragma unroll
        for (int k = 0; k < TW; ++k) cur[k] = nxt[k];
    }
}

int thin_grid() { return bd_num_cus(); }

}  // namespace

extern "C" int bd_conv1x1_thin_fwd(const void* x, const float* w, const float* bias, int64_t M, int Cin, int Cout, void* y, bd_stream_t stream) {
    BD_REQUIRE(x && w && y, "conv1x1_thin_fwd: null pointer");
    BD_REQUIRE(Cin == TCIN && Cout == TO && M > 0 && M < (1ll << 31) * 16, "conv1x1_thin_fwd: the layer is %d -> %d channels (supported: %d -> %d)",
               Cin, Cout, TCIN, TO);
    const int groups = (int)((M + 15) / 16);
    int grid = bd_num_cus() * 4;
    if (grid > (groups + 3) / 4) grid = (groups + 3) / 4;
    hipLaunchKernelGGL(conv1x1_thin_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)x, w, bias, (bf16_raw*)y, (long long)M, groups);
    BD_CHECK_LAUNCH("bd_conv1x1_thin_fwd");
    bd_note_kernel("conv1x1_thin_fwd_kernel");
    return BD_OK;
}

extern "C" size_t bd_conv1x1_thin_bwd_workspace_bytes(void) { return (size_t)thin_grid() * (TO * TCIN + TO) * 4 + 256; }

extern "C" int bd_conv1x1_thin_bwd(const void* x, const void* g, const float* w, int64_t M, int Cin, int Cout, void* dx, float* dw,
                                   float* dbias, int cout_real, void* ws, size_t ws_bytes, bd_stream_t stream) {
    BD_REQUIRE(x && g && w && dx && dw && ws, "conv1x1_thin_bwd: null pointer");
    BD_REQUIRE(Cin == TCIN && Cout == TO && M > 0 && M < (1ll << 31) * 16 && cout_real > 0 && cout_real <= TO,
               "conv1x1_thin_bwd: the layer is %d -> %d channels (supported: %d -> %d)", Cin, Cout, TCIN, TO);
    if (ws_bytes < bd_conv1x1_thin_bwd_workspace_bytes()) {
        bd_set_error("conv1x1_thin_bwd: workspace %zu < %zu bytes", ws_bytes, bd_conv1x1_thin_bwd_workspace_bytes());
        return BD_EWORKSPACE;
    }
    ThinParams p{};
    p.x = (const bf16_raw*)x; p.g = (const bf16_raw*)g; p.w = w; p.dx = (bf16_raw*)dx; p.slab = (float*)ws; p.M = M;
    p.groups = (int)((M + 15) / 16);
    const int grid = p.groups < thin_grid() ? p.groups : thin_grid();
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(conv1x1_thin_bwd_kernel, dim3(grid), dim3(TW * 64), 0, st, p);
    hipLaunchKernelGGL(conv1x1_thin_reduce_kernel, dim3(((TO * TCIN + TO) * 8 + 255) / 256), dim3(256), 0, st, (const float*)ws, grid, dw, dbias, cout_real);
    BD_CHECK_LAUNCH("bd_conv1x1_thin_bwd");
    bd_note_kernel("conv1x1_thin_bwd_kernel");
    return BD_OK;
}
