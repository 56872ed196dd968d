// Dense 1x1 convolution (forward and data gradient) = a plain GEMM  D[m][co] = sum_k X[m][k] * W[co][k]  over the pixels of ONE
// dense level (source pixel index == destination pixel index == m), with the fused epilogue of conv_igemm.hip.
//
// The ResNet bottleneck 1x1s and the FPN laterals: 40-225 FLOP per byte, the largest share of the step after the 3x3 convolutions.
// Same tile as the generic kernel (128 channels x 128 pixels, BK = 32, swizzled 64-byte LDS rows, 16x16x32 MFMA, channels on the MFMA
// row), specialised so that
//   * there is no tap / sub-segment / pixel-decode state: 121 VGPRs instead of 160 -> FOUR workgroups per CU instead of three;
//   * the ReLU mask of a data gradient may come BIT-PACKED (1 bit per element instead of a bf16: a 16x smaller stream), written by
//     the forward launch that produced the activation (ybits): bits[g * M + m], bit b = channel 32 g + b of pixel m is > 0.
// What was measured on the way (scripts/micro_1x1_step.py, bench.py --dense1x1): running the staging loads 2, 4
// or 6 K steps ahead through extra register sets -- at three or two workgroups per CU -- and requesting the epilogue operands before
// the last K step changed nothing on the 200x336 / 100x168 layers (4.2-4.8 / 3.5-3.9 TB/s either way) and LOST 10-35 % on the K >=
// 1024 layers, where the kernel is bound by its MFMA loop (one barrier per 16 MFMAs), not by memory: occupancy, not prefetch depth,
// is what fills that loop's bubbles.  (Those variants are gone; the DEPTH parameter below is what is left of them.)
#include "common.h"

namespace {

constexpr int TP = 128, TC = 128, BK = 32;
constexpr int TILE_BYTES = 128 * 64;

struct P1 {
    const bf16_raw* x;
    const bf16_raw* w;
    const float* bias;
    const bf16_raw* add;
    const bf16_raw* mask;
    const unsigned* maskbits;
    bf16_raw* y;
    unsigned* ybits;
    unsigned char* y8;        // optional one-byte twin of y (y * q_scale) for a following fp8 launch (conv3x3_pp8.hip):
    float q_scale;            // e4m3 for activations (y8_bf8 = 0), e5m2 for gradients (y8_bf8 = 1)
    int y8_bf8;
    int M, CK, CO, flags;
    unsigned x_bytes, w_bytes;
    int m_tiles, n_tiles;
};

__device__ __forceinline__ unsigned pack4_e4m3(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
    c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
    int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (unsigned)v;
}

__device__ __forceinline__ unsigned pack4_e5m2(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -57344.f), 57344.f); b = fminf(fmaxf(b, -57344.f), 57344.f);
    c = fminf(fmaxf(c, -57344.f), 57344.f); d = fminf(fmaxf(d, -57344.f), 57344.f);
    int v = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false);
    v = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, v, true);
    return (unsigned)v;
}

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 3)) << 4); }

template <int DEPTH, int WAVES>
__global__ __launch_bounds__(256, WAVES) void conv1x1_dense_kernel(const P1 p) {
    constexpr bool EARLY = false;       // (epilogue operands requested before the last K step: measured, no gain -- see the header)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wc = wave >> 1;   // channel half
    const int wp = wave & 1;    // pixel half

    // XCD-aware bijective remap: consecutive tile ids (the channel tiles of one pixel tile, then the next pixel tile) share an XCD's L2
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / p.n_tiles;
    const int tile_n = bid - tile_m * p.n_tiles;
    const int m0 = tile_m * TP;
    const int co0 = tile_n * TC;

    const int chunk = tid & 3;
    const int row0 = tid >> 2;          // rows row0 and row0 + 64 of both operand tiles

    constexpr unsigned X_NONE = 0x80000000u;          // >= num_records: the buffer load returns zeros
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.w), 0, p.w_bytes, 0x00020000);
    unsigned a_voff[2], b_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        // LDS row (h*64 + t*16 + rho) of the weight tile holds output channel h*64 + 32*(t>>1) + 8*(rho>>2) + 4*(t&1) + (rho&3): after the
        // MFMAs lane group cg = rho>>2 owns channels 8*cg..8*cg+7 of each 32-channel half (16-byte stores, 64 contiguous bytes per pixel)
        const int lrow = row0 + i * 64;
        const int rho = lrow & 15;
        const int co = co0 + (lrow & 64) + 32 * ((lrow >> 5) & 1) + 8 * (rho >> 2) + 4 * ((lrow >> 4) & 1) + (rho & 3);
        a_voff[i] = co < p.CO ? (unsigned)(co * p.CK + chunk * 8) * 2u : X_NONE;
        const int m = m0 + lrow;
        b_voff[i] = m < p.M ? (unsigned)(m * p.CK + chunk * 8) * 2u : X_NONE;
    }
    const int nsteps = (p.CK + BK - 1) / BK;

    u32x4_t ra[DEPTH][2], rb[DEPTH][2];
    auto stage_load = [&](int step, u32x4_t (&a)[2], u32x4_t (&b)[2]) {
        int so = step * BK * 2;
        asm volatile("" : "+s"(so));                               // keep the K offset in the scalar operand
        const bool dead = step * BK + chunk * 8 >= p.CK;           // channel tail (CK % 8 == 0): zero-fill
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[i] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, dead ? X_NONE : a_voff[i], so, 0);
            b[i] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, dead ? X_NONE : b_voff[i], so, 0);
        }
    };
    auto stage_write = [&](int buf, const u32x4_t (&a)[2], const u32x4_t (&b)[2]) {
        unsigned char* At = smem + buf * 2 * TILE_BYTES;
        unsigned char* Bt = At + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<u32x4_t*>(At + lds_off(row0 + i * 64, chunk)) = a[i];
            *reinterpret_cast<u32x4_t*>(Bt + lds_off(row0 + i * 64, chunk)) = b[i];
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int frag_row = lane & 15;
    const int frag_chunk = lane >> 4;
    auto compute = [&](int buf) {
        const unsigned char* At = smem + buf * 2 * TILE_BYTES;
        const unsigned char* Bt = At + TILE_BYTES;
        bf16x8_t a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(At + lds_off(wc * 64 + i * 16 + frag_row, frag_chunk));
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(Bt + lds_off(wp * 64 + j * 16 + frag_row, frag_chunk));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };

    // ---- epilogue operand addresses (lane (cg = lane>>4) holds channels cbase + 32*half + 0..7 of pixel m0 + wp*64 + j*16 + (lane&15))
    const int cg = lane >> 4;
    const int cbase = co0 + wc * 64 + 8 * cg;
    const bool do_relu = p.flags & BD_EPI_RELU;
    const bool add_before = (p.flags & BD_EPI_ADD_BEFORE) && p.add;
    const bool add_after = (p.flags & BD_EPI_ADD_AFTER) && p.add;
    const bool want_add = add_before || add_after;
    const bool mask_bf = (p.flags & BD_EPI_MASK) && p.mask;
    const bool mask_bits = (p.flags & BD_EPI_MASK) && p.maskbits && !p.mask;
    u32x4_t e_add[EARLY ? 8 : 1];
    unsigned e_bits[EARLY ? 8 : 1];
    auto epi_request = [&]() {
        if (!EARLY) return;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int m = m0 + wp * 64 + j * 16 + (lane & 15);
                const bool ok = m < p.M && cbase + 32 * half < p.CO;
                const long long idx = (long long)m * p.CO + cbase + 32 * half;
                u32x4_t a = {0u, 0u, 0u, 0u};
                unsigned bits = 0u;
                if (ok && want_add) a = *reinterpret_cast<const u32x4_t*>(p.add + idx);
                if (ok && mask_bits) bits = p.maskbits[(long long)((co0 + wc * 64 + 32 * half) >> 5) * p.M + m];
                e_add[j * 2 + half] = a; e_bits[j * 2 + half] = bits;
            }
    };

    // ---- main loop: step t is computed from LDS buffer t & 1 while the loads of steps t+1 .. t+DEPTH are in flight ------------------
#pragma unroll
    for (int u = 0; u < DEPTH; ++u)
        if (u < nsteps) stage_load(u, ra[u], rb[u]);
    stage_write(0, ra[0], rb[0]);
    __syncthreads();
    for (int t0 = 0; t0 < nsteps; t0 += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const int t = t0 + u;
            if (t < nsteps) {                                       // workgroup-uniform
                // set u held step t (written to LDS in the previous iteration): it is free for step t + DEPTH
                if (t + DEPTH < nsteps) stage_load(t + DEPTH, ra[u], rb[u]);
                if (t + 1 == nsteps) epi_request();                 // last step: the staging sets are dead, their registers take the epilogue operands
                compute(t & 1);
                if (t + 1 < nsteps) stage_write((t + 1) & 1, ra[(u + 1) % DEPTH], rb[(u + 1) % DEPTH]);
                __syncthreads();
            }
        }
    }

    // ---- epilogue -------------------------------------------------------------------------------------------------------------------
    float bias[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) bias[k] = 0.f;
    if (p.bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (cbase + 32 * (q >> 1) + 4 * (q & 1) < p.CO) {
                const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(p.bias + cbase + 32 * (q >> 1) + 4 * (q & 1));
                bias[4 * q] = bv[0]; bias[4 * q + 1] = bv[1]; bias[4 * q + 2] = bv[2]; bias[4 * q + 3] = bv[3];
            }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wp * 64 + j * 16 + (lane & 15);
#pragma unroll
        for (int half = 0; half < 2; ++half) {       // 8 channels = 16 bytes per half
            const bool ok = m < p.M && cbase + 32 * half < p.CO;      // CO % 8 == 0
            const long long idx = (long long)m * p.CO + cbase + 32 * half;
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = acc[2 * half + (k >> 2)][j][k & 3] + bias[8 * half + k];
            u32x4_t av = {0u, 0u, 0u, 0u};
            unsigned mbits = 0u;
            if (EARLY) { av = e_add[j * 2 + half]; mbits = e_bits[j * 2 + half]; }
            else {
                if (ok && want_add) av = *reinterpret_cast<const u32x4_t*>(p.add + idx);
                if (ok && mask_bits) mbits = p.maskbits[(long long)((co0 + wc * 64 + 32 * half) >> 5) * p.M + m];
            }
            if (add_before) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[2 * k] += bf_lo(av[k]); v[2 * k + 1] += bf_hi(av[k]); }
            }
            if (do_relu) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
            }
            if (mask_bf) {            // the bf16 form of the mask is requested here (the small-channel consumers; the wide ones come bit-packed)
                u32x4_t mv = {0u, 0u, 0u, 0u};
                if (ok) mv = *reinterpret_cast<const u32x4_t*>(p.mask + idx);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (!(bf_lo(mv[k]) > 0.f)) v[2 * k] = 0.f;
                    if (!(bf_hi(mv[k]) > 0.f)) v[2 * k + 1] = 0.f;
                }
            }
            if (mask_bits) {
                const unsigned byte = mbits >> (8 * cg);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (!((byte >> k) & 1u)) v[k] = 0.f;
            }
            if (add_after) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[2 * k] += bf_lo(av[k]); v[2 * k + 1] += bf_hi(av[k]); }
            }
            u32x4_t o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = pack_bf2(v[2 * k], v[2 * k + 1]);
            if (ok) *reinterpret_cast<u32x4_t*>(p.y + idx) = o;
            if (p.y8 && ok) {
                u32x2_t o8;
                if (p.y8_bf8) {
                    o8[0] = pack4_e5m2(v[0] * p.q_scale, v[1] * p.q_scale, v[2] * p.q_scale, v[3] * p.q_scale);
                    o8[1] = pack4_e5m2(v[4] * p.q_scale, v[5] * p.q_scale, v[6] * p.q_scale, v[7] * p.q_scale);
                } else {
                    o8[0] = pack4_e4m3(v[0] * p.q_scale, v[1] * p.q_scale, v[2] * p.q_scale, v[3] * p.q_scale);
                    o8[1] = pack4_e4m3(v[4] * p.q_scale, v[5] * p.q_scale, v[6] * p.q_scale, v[7] * p.q_scale);
                }
                *reinterpret_cast<u32x2_t*>(p.y8 + idx) = o8;
            }
            if (p.ybits) {
                // the stored bf16 value is > 0 exactly when the fp32 value is (rounding to nearest cannot reach 0 from a positive normal)
                unsigned byte = 0u;
#pragma unroll
                for (int k = 0; k < 8; ++k) byte |= (v[k] > 0.f ? 1u : 0u) << k;
                unsigned word = byte << (8 * cg);
                word |= __shfl_xor(word, 16, 64);
                word |= __shfl_xor(word, 32, 64);
                if (ok && cg == (j & 3)) p.ybits[(long long)((co0 + wc * 64 + 32 * half) >> 5) * p.M + m] = word;
            }
        }
    }
}

int g_conv1x1_depth = 1;        // bd_conv_set_dense1x1: 0 = off (the generic kernel takes the dense 1x1 launches: A/B), 1 = on

}  // namespace

extern "C" int bd_conv_set_dense1x1(int depth) {
    if (!(depth == 0 || depth == 1)) {
        bd_set_error("bd_conv_set_dense1x1: %d (0 or 1)", depth);
        return BD_EINVAL;
    }
    g_conv1x1_depth = depth;
    return BD_OK;
}

// Called by bd_conv2d_fwd / bd_conv2d_dgrad (conv_igemm.hip) for 1x1 / stride 1 / pad 0 launches over one dense level.
// Returns 0 when the launch was taken, 1 when the shape is left to the generic kernel.
int bd_conv1x1_dense_launch(const void* x, const void* w, const float* bias, const void* add, const void* mask, const unsigned* maskbits,
                            void* y, unsigned* ybits, void* y8, float q_scale, int y8_bf8, long long M, int CK, int CO, int flags,
                            hipStream_t stream) {
    if (g_conv1x1_depth == 0) return 1;
    const long long xb = M * CK * 2, wb = (long long)CO * CK * 2;
    if (xb >= 0x7fffffffll || wb >= 0x7fffffffll || M >= (1ll << 24)) return 1;
    if ((maskbits || ybits) && (CO % 32 != 0)) return 1;
    P1 p{};
    p.x = (const bf16_raw*)x; p.w = (const bf16_raw*)w; p.bias = bias; p.add = (const bf16_raw*)add; p.mask = (const bf16_raw*)mask;
    p.maskbits = maskbits; p.y = (bf16_raw*)y; p.ybits = ybits; p.y8 = (unsigned char*)y8; p.q_scale = q_scale; p.y8_bf8 = y8_bf8;
    p.M = (int)M; p.CK = CK; p.CO = CO; p.flags = flags;
    p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
    p.m_tiles = (int)cdiv64(M, TP); p.n_tiles = cdiv(CO, TC);
    const int grid = p.m_tiles * p.n_tiles;
    const size_t lds = 4 * TILE_BYTES;
    hipLaunchKernelGGL((conv1x1_dense_kernel<1, 4>), dim3(grid), dim3(256), lds, stream, p);
    return 0;
}
