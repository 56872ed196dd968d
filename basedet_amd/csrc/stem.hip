// ResNet stem: 7x7 stride-2 pad-3 convolution 3 -> 64 with folded FrozenBN + ReLU (forward only: the stem is
// frozen, solver/default_solver.py:83-94).
//
// C_in = 3 is useless as an MFMA K dimension, so K is laid out per FILTER ROW: the input is stored NHWC with
// C padded to 4 and a zero halo (bd_pad_normalize), which makes the 7 taps x 4 channels of one filter row 56
// contiguous bytes; padded to 8 taps (64 bytes) that is exactly one K=32 MFMA step.  K = 7 rows x 32.
// The B fragment of a lane (pixel, 16-byte chunk) is read straight from global memory (16-byte aligned because
// windows start at even pixels); neighbouring windows overlap, which the vector L1 absorbs.  The 64x224 weight
// matrix lives in LDS for the lifetime of a persistent workgroup.
#include "common.h"

namespace {

constexpr int W_PITCH = 480;  // bytes per weight row in LDS: 448 data + 32 pad (conflict-free b128 reads)

__global__ __launch_bounds__(256) void stem_conv_kernel(const bf16_raw* __restrict__ x, const bf16_raw* __restrict__ w,
                                                        const float* __restrict__ bias, bf16_raw* __restrict__ y,
                                                        int N, int H, int W) {
    __shared__ __attribute__((aligned(16))) unsigned char wl[64 * W_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // weights: 64 rows x 28 chunks of 16 B.  LDS row (t*16 + rho) holds output channel 32*(t>>1) + 8*(rho>>2) + 4*(t&1) + (rho&3)
    // (the permutation of conv_igemm.hip): after the MFMAs lane group q owns channels 8q..8q+7 of each 32-channel half, so the
    // epilogue stores 16 bytes per lane = 64 contiguous bytes per pixel and instruction instead of 8 / 32
    for (int c = tid; c < 64 * 28; c += 256) {
        const int row = c / 28, ch = c - row * 28;
        const int rho = row & 15, t = row >> 4;
        const int co = 32 * (t >> 1) + 8 * (rho >> 2) + 4 * (t & 1) + (rho & 3);
        *reinterpret_cast<u32x4_t*>(wl + row * W_PITCH + ch * 16) =
            *reinterpret_cast<const u32x4_t*>(w + co * 224 + ch * 8);
    }
    __syncthreads();

    const int Ho = H / 2, Wo = W / 2;
    const int Hb = H + 6, Wb = W + 8;
    const int M = N * Ho * Wo;                    // < 2^31 (checked on the host): 32-bit index arithmetic
    const int ngroups = (M + 63) / 64;
    const int pix = lane & 15, q = lane >> 4;
    float bias_r[16];                              // this lane's 16 channels: 8q..8q+7 and 32+8q..32+8q+7
#pragma unroll
    for (int k = 0; k < 16; ++k) bias_r[k] = bias[32 * (k >> 3) + 8 * q + (k & 7)];

    for (int grp = blockIdx.x * 4 + wave; grp < ngroups; grp += gridDim.x * 4) {
        const int mbase = grp * 64;
        const bf16_raw* src[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int m = mbase + j * 16 + pix;
            if (m >= M) m = M - 1;   // clamp loads, stores are masked
            const int n = m / (Ho * Wo);
            const int rem = m - n * Ho * Wo;
            const int oy = rem / Wo, ox = rem - oy * Wo;
            src[j] = x + (((long long)n * Hb + 2 * oy) * Wb + 2 * ox + 2 * q) * 4;
        }
        f32x4_t acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            bf16x8_t a[4], b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(src[j] + (long long)r * Wb * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = *reinterpret_cast<const bf16x8_t*>(wl + (i * 16 + pix) * W_PITCH + (r * 4 + q) * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mbase + j * 16 + pix;
            if (m >= M) continue;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                u32x4_t o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e0 = 2 * k, e1 = 2 * k + 1;       // element e of the 8: tile 2*half + (e>>2), register e&3
                    o[k] = pack_bf2(fmaxf(acc[2 * half + (e0 >> 2)][j][e0 & 3] + bias_r[8 * half + e0], 0.f),
                                    fmaxf(acc[2 * half + (e1 >> 2)][j][e1 & 3] + bias_r[8 * half + e1], 0.f));
                }
                *reinterpret_cast<u32x4_t*>(y + (long long)m * 64 + 32 * half + 8 * q) = o;
            }
        }
    }
}

__global__ void stem_weight_pack_kernel(const float* __restrict__ w, const float* __restrict__ row_scale,
                                        bf16_raw* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over 64*7*8*4
    if (i >= 64 * 224) return;
    const int c = i & 3, pos = (i >> 2) & 7, r = (i >> 5) % 7, co = i / 224;
    float v = 0.f;
    if (c < 3 && pos >= 1) {
        v = w[((co * 7 + r) * 7 + (pos - 1)) * 3 + c];
        if (row_scale) v *= row_scale[co];
    }
    out[i] = f2bf(v);
}

}  // namespace

extern "C" int bd_stem_conv7x7_fwd(int N, int H, int W, const void* x_halo, const void* w_stem, const float* bias,
                                   void* y, bd_stream_t stream) {
    BD_REQUIRE(x_halo && w_stem && bias && y, "stem_conv7x7_fwd: null pointer");
    BD_REQUIRE(N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "stem_conv7x7_fwd: H=%d W=%d must be even", H, W);
    const long long M = (long long)N * (H / 2) * (W / 2);
    BD_REQUIRE(M < 0x7fffffffll, "stem_conv7x7_fwd: too many output pixels");
    long long groups = (M + 255) / 256;
    const int grid = (int)(groups < 2048 ? groups : 2048);
    hipLaunchKernelGGL(stem_conv_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)x_halo,
                       (const bf16_raw*)w_stem, bias, (bf16_raw*)y, N, H, W);
    BD_CHECK_LAUNCH("bd_stem_conv7x7_fwd");
    return BD_OK;
}

extern "C" int bd_stem_weight_pack(const float* w, const float* row_scale, void* w_stem, bd_stream_t stream) {
    BD_REQUIRE(w && w_stem, "stem_weight_pack: null pointer");
    hipLaunchKernelGGL(stem_weight_pack_kernel, dim3(cdiv(64 * 224, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       row_scale, (bf16_raw*)w_stem);
    BD_CHECK_LAUNCH("bd_stem_weight_pack");
    return BD_OK;
}
