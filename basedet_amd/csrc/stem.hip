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

// ---- stem + max-pool in one pass -------------------------------------------------------------------------------------------
// The stem is frozen and nothing but M.MaxPool2d(3, 2, 1) reads its output, so the 64-channel half-resolution tensor (550 MB at
// 16 x 800 x 1344) never needs to exist: this kernel writes the pooled quarter-resolution tensor only (-1.1 GB of HBM traffic).
// One WAVE is autonomous: it owns a strip of 16 stem columns (= 7 pooled columns: pooled column 7k + m takes stem columns
// 14k + 2m - 1 .. + 1, i.e. lanes 2m .. 2m + 2 of the 16-pixel MFMA column group) and walks down `rows` pooled rows, one stem row
// (16 pixels x 64 channels = 28 MFMAs) at a time.  Weights (64 x 224 bf16 = 28 KB) live in 112 VGPRs for the life of the wave; the
// input rows arrive by LDS-DMA into a per-wave ring of 16 row pairs (each input row is fetched once per strip instead of 3.5 times),
// eight stem rows ahead, under a counted vmcnt; the horizontal 3-max is two DPP row shifts, the vertical one a running register.
// Column halo: 16 computed stem columns per 14 used (x 1.14), row halo one stem row per chunk.
constexpr int SP_PAIRS = 16;       // ring: 16 row pairs x 1 KiB (a pair = two input rows x 64 pixels x 8 B) per wave
constexpr int SP_AHEAD = 8;        // DMA distance in stem rows (12 measured the same)
typedef __attribute__((address_space(3))) void sp_lds_void_t;
typedef __attribute__((ext_vector_type(2))) short i16x2_t;

__device__ __forceinline__ unsigned sp_max2(unsigned a, unsigned b) {      // packed bf16 max of non-negative values (or -0): signed 16-bit order
    const i16x2_t r = __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, a), __builtin_bit_cast(i16x2_t, b));
    return __builtin_bit_cast(unsigned, r);
}

__global__ __launch_bounds__(256, 2) void stem_pool_kernel(const bf16_raw* __restrict__ x, const bf16_raw* __restrict__ w,
                                                           const float* __restrict__ bias, bf16_raw* __restrict__ y,
                                                           int N, int H, int W, long long x_bytes, int rows, int nk, int nchunks) {
    __shared__ __attribute__((aligned(1024))) unsigned char ring_all[4 * SP_PAIRS * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* ring = ring_all + wave * (SP_PAIRS * 1024);
    const int pix = lane & 15, q = lane >> 4;
    const int Ho = H / 2, Wo = W / 2, Hb = H + 6, Wb = W + 8;
    const int Hq = (Ho - 1) / 2 + 1, Wq = (Wo - 1) / 2 + 1;

    // A fragments: tile i, row pix holds output channel 32 (i >> 1) + 8 (pix >> 2) + 4 (i & 1) + (pix & 3) (the permutation of
    // stem_conv_kernel), so lane group q ends up with channels 8q .. 8q + 7 of each 32-channel half
    bf16x8_t a[7][4];
#pragma unroll
    for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = 32 * (i >> 1) + 8 * (pix >> 2) + 4 * (i & 1) + (pix & 3);
            a[r][i] = *reinterpret_cast<const bf16x8_t*>(w + co * 224 + (r * 4 + q) * 8);
        }
    float bias_r[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) bias_r[k] = bias[32 * (k >> 3) + 8 * q + (k & 7)];

    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(x), 0, (unsigned)x_bytes, 0x00020000);
    const long long lane_off = (long long)(lane >> 5) * Wb * 8 + (lane & 31) * 16;     // lanes 0-31: first row of the pair, 32-63: second
    const long long pair_stride = 2ll * Wb * 8;
    const unsigned b_lane = (unsigned)((pix + q) * 16);                                 // B fragment: pixels 2 pix + 2q, + 1 of the strip row

    const int items = N * nk * nchunks;
    for (int it = blockIdx.x * 4 + wave; it < items; it += gridDim.x * 4) {
        const int c = it % nchunks, k = (it / nchunks) % nk, n = it / (nchunks * nk);
        const int p0 = c * rows, p1 = p0 + rows < Hq ? p0 + rows : Hq;
        const int s_begin = 2 * p0 - 1, nrows = 2 * (p1 - p0) + 1;
        const int c0 = 14 * k - 1;
        // byte offset of (image n, halo row 2 s_begin, halo column 2 c0): negative above / left of the tensor -> the range check
        // returns zeros (those stem rows / columns are padding and are zeroed below in any case)
        const long long base = (((long long)n * Hb + 2 * s_begin) * Wb + 2 * c0) * 8 + lane_off;
        // (a strip needs input pixels 2 c0 .. 2 c0 + 37 of a row; the ring row holds 64.  Requesting only the 40 measured SLOWER -- 174 vs
        // 145 us, with HBM traffic down from 708 to 518 MB per launch: the kernel waits on DMA latency, and the surplus pixels are the next
        // strip's, fetched ahead of it into L2)
        auto dma = [&](int j) {
            const long long o = base + j * pair_stride;
            const unsigned vo = (o < 0 || o + 16 > x_bytes) ? 0x80000000u : (unsigned)o;
            unsigned char* l = ring + (j & (SP_PAIRS - 1)) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (sp_lds_void_t*)l, 16, vo, 0, 0, 0);
        };
        bf16x8_t b[2][7];
        auto read_b = [&](int t, int buf) {                      // stem row t reads relative input rows 2t .. 2t + 6
#pragma unroll
            for (int r = 0; r < 7; ++r)
                b[buf][r] = *reinterpret_cast<const bf16x8_t*>(ring + (((2 * t + r) & (2 * SP_PAIRS - 1)) * 512) + b_lane);
        };
        const bool col_ok = (c0 + pix) >= 0 && (c0 + pix) < Wo;
        const int px = 7 * k + (pix >> 1);
        const bool st_lane = (pix & 1) == 0 && pix <= 12 && px < Wq;
        unsigned run[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) run[e] = 0u;

#pragma unroll
        for (int j = 0; j < SP_AHEAD; ++j) dma(j);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SP_AHEAD - 4) : "memory");        // pairs 0 .. 3 have landed
        read_b(0, 0);

        auto row = [&](int t, int cur) {
            dma(t + SP_AHEAD);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SP_AHEAD - 4) : "memory");    // pair t + 4 (the last one of stem row t + 1)
            read_b(t + 1, cur ^ 1);
            f32x4_t acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 7; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r][i], b[cur][r], acc[i], 0, 0, 0);
            const int srow = s_begin + t;
            const bool ok = col_ok && srow >= 0 && srow < Ho;
            unsigned h[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {               // packed pair e: channels 2 (e & 3), + 1 of half e >> 2
                const int half = e >> 2, e0 = 2 * (e & 3), e1 = e0 + 1;
                unsigned v = pack_bf2(fmaxf(acc[2 * half + (e0 >> 2)][e0 & 3] + bias_r[8 * half + e0], 0.f),
                                      fmaxf(acc[2 * half + (e1 >> 2)][e1 & 3] + bias_r[8 * half + e1], 0.f));
                v = ok ? v : 0u;
                const unsigned s1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true);   // row_shl:1: lane pix <- pix + 1
                const unsigned s2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x102, 0xf, 0xf, true);   // row_shl:2
                h[e] = sp_max2(v, sp_max2(s1, s2));
            }
            if (t & 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) run[e] = sp_max2(run[e], h[e]);
            } else {
                if (t > 0) {
                    const int p = p0 + (t >> 1) - 1;
                    if (st_lane) {
                        bf16_raw* dst = y + ((((long long)n * Hq + p) * Wq + px) * 64 + 8 * q);
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            u32x4_t o;
#pragma unroll
                            for (int kk = 0; kk < 4; ++kk) o[kk] = sp_max2(run[4 * half + kk], h[4 * half + kk]);
                            *reinterpret_cast<u32x4_t*>(dst + 32 * half) = o;
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) run[e] = h[e];
            }
        };
        for (int t = 0; t < nrows; t += 2) {       // nrows is odd: the second call of the last pass computes a row nobody reads
            row(t, 0);
            row(t + 1, 1);
        }
    }
    // the ring runs SP_AHEAD pairs ahead of the last row: let those DMA writes land before the wave ends and its LDS can be handed
    // to another workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

extern "C" int bd_stem_pool_fwd(int N, int H, int W, const void* x_halo, const void* w_stem, const float* bias, void* y_pool,
                               bd_stream_t stream) {
    BD_REQUIRE(x_halo && w_stem && bias && y_pool, "stem_pool_fwd: null pointer");
    BD_REQUIRE(N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "stem_pool_fwd: H=%d W=%d must be even", H, W);
    const long long x_bytes = (long long)N * (H + 6) * (W + 8) * 8;
    BD_REQUIRE(x_bytes < 0x7fffffffll, "stem_pool_fwd: input of %lld bytes (the buffer loads address < 2 GB)", x_bytes);
    const int Hq = (H / 2 - 1) / 2 + 1, Wq = (W / 2 - 1) / 2 + 1;
    const int rows = 10, nk = cdiv(Wq, 7), nchunks = cdiv(Hq, rows);
    const long long items = (long long)N * nk * nchunks;
    BD_REQUIRE(items < 0x7fffffffll, "stem_pool_fwd: too many strips");
    const long long blocks = (items + 3) / 4;
    const int grid = (int)(blocks < 512 ? blocks : 512);       // 2 workgroups (8 waves: the ring and the register budget) per CU
    hipLaunchKernelGGL(stem_pool_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)x_halo,
                       (const bf16_raw*)w_stem, bias, (bf16_raw*)y_pool, N, H, W, x_bytes, rows, nk, nchunks);
    BD_CHECK_LAUNCH("bd_stem_pool_fwd");
    return BD_OK;
}

extern "C" int bd_stem_weight_pack(const float* w, const float* row_scale, void* w_stem, bd_stream_t stream) {
    BD_REQUIRE(w && w_stem, "stem_weight_pack: null pointer");
    hipLaunchKernelGGL(stem_weight_pack_kernel, dim3(cdiv(64 * 224, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       row_scale, (bf16_raw*)w_stem);
    BD_CHECK_LAUNCH("bd_stem_weight_pack");
    return BD_OK;
}
