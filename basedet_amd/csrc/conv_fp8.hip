// fp8 (OCP e4m3) forward convolution on the block-scaled CDNA4 MFMA, v_mfma_scale_f32_16x16x128_f8f6f4 (2x the bf16 matrix rate).
//
// BASELINE config 5 ("fp8 weights"): the reference's mixed-precision hook is fp16 autocast (solver/default_solver.py:66-76,
// tools/det_train.py:77-78); it has no fp8 counterpart, so the tolerance is stated against the fp32 oracle (tests/test_fp8_gpu.py).
//   weights      fp32 masters -> e4m3 with ONE SCALE PER OUTPUT CHANNEL (s_co = max |w_co * bn_scale| / 448), packed [Cout][R*S][Cin];
//   activations  bf16 in HBM (the backward pass and the 1x1 layers keep reading them); gfx950 has no mixed bf16 x fp8 MFMA, so a
//                3x3 convolution's input is cast to e4m3 by bd_quantize_fp8 (x * act_scale, clamped to +-448) into a scratch
//                tensor that its nine taps then read at half the bytes;
//   accumulate   fp32; the epilogue multiplies by s_co / act_scale, then bias / residual / ReLU as in the bf16 kernels; output bf16.
//
// Kernel = the generic implicit GEMM of conv_igemm.hip (128 channels x 128 pixels, 4 waves, weights on the MFMA row, range-checked
// buffer loads, two LDS buffers) with one-byte elements: a K step is 128 channels of one filter tap (128-byte LDS rows, 160-byte
// pitch), a lane's A / B fragment is 32 consecutive bytes of its row -- operand map of the 16x16x128 form, probed with exact data by
// scripts/exp/mfma_fp8_layout.hip: lane l holds row / column l & 15, k = 32 (l >> 4) .. + 31 -- and the E8M0 block scales are all
// 2^0 (the per-channel scale lives in the epilogue, where it costs one multiply per output).
#include "igemm_params.h"

using namespace igemm;

BD_KNOB unsigned g_fp8_sr_seed = 0;        // bd_conv_desc.sr_seed: read by every e5m2 quantiser's launcher (common.h)

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8_t;

constexpr int KB = 128;                 // channels (= bytes) per K step
constexpr int PITCH = 160;              // LDS row pitch
constexpr int TILE_BYTES = 128 * PITCH;
constexpr int PASSES = 4;               // 128 rows x 8 chunks / 256 threads
constexpr float FP8_MAX = 448.f;

__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -FP8_MAX), FP8_MAX); b = fminf(fmaxf(b, -FP8_MAX), FP8_MAX);
    c = fminf(fmaxf(c, -FP8_MAX), FP8_MAX); d = fminf(fmaxf(d, -FP8_MAX), FP8_MAX);
    int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (unsigned)v;
}

// x bf16 -> e4m3(x * scale); 16 elements per thread and trip (two 16-byte loads, one 16-byte store)
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const bf16_raw* __restrict__ x, long long n16, float scale,
                                                          u32x4_t* __restrict__ q) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
        const u32x4_t a = *reinterpret_cast<const u32x4_t*>(x + i * 16);
        const u32x4_t b = *reinterpret_cast<const u32x4_t*>(x + i * 16 + 8);
        u32x4_t o;
        o[0] = pack4_fp8(bf_lo(a[0]) * scale, bf_hi(a[0]) * scale, bf_lo(a[1]) * scale, bf_hi(a[1]) * scale);
        o[1] = pack4_fp8(bf_lo(a[2]) * scale, bf_hi(a[2]) * scale, bf_lo(a[3]) * scale, bf_hi(a[3]) * scale);
        o[2] = pack4_fp8(bf_lo(b[0]) * scale, bf_hi(b[0]) * scale, bf_lo(b[1]) * scale, bf_hi(b[1]) * scale);
        o[3] = pack4_fp8(bf_lo(b[2]) * scale, bf_hi(b[2]) * scale, bf_lo(b[3]) * scale, bf_hi(b[3]) * scale);
        q[i] = o;
    }
}

__device__ __forceinline__ unsigned pack4_bf8(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -57344.f), 57344.f); b = fminf(fmaxf(b, -57344.f), 57344.f);
    c = fminf(fmaxf(c, -57344.f), 57344.f); d = fminf(fmaxf(d, -57344.f), 57344.f);
    int v = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false);
    v = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, v, true);
    return (unsigned)v;
}

// x bf16 -> e5m2(x * scale): the gradient operand of the fp8 data-gradient kernel
__global__ __launch_bounds__(256) void quantize_bf8_kernel(const bf16_raw* __restrict__ x, long long n16, float scale,
                                                          u32x4_t* __restrict__ q, unsigned sr_seed) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
        const u32x4_t a = *reinterpret_cast<const u32x4_t*>(x + i * 16);
        const u32x4_t b = *reinterpret_cast<const u32x4_t*>(x + i * 16 + 8);
        u32x4_t o;
        if (sr_seed) {            // element group g = index / 4: the same random word as a producing launch's twin of this element
            const unsigned g0 = (unsigned)(i * 4);
            const unsigned r0 = bd_mix32(sr_seed ^ g0), r2 = bd_mix32(sr_seed ^ (g0 + 2));      // as the twin writers: one hash per eight elements
            o[0] = bd_pack4_e5m2_sr(bf_lo(a[0]) * scale, bf_hi(a[0]) * scale, bf_lo(a[1]) * scale, bf_hi(a[1]) * scale, r0);
            o[1] = bd_pack4_e5m2_sr(bf_lo(a[2]) * scale, bf_hi(a[2]) * scale, bf_lo(a[3]) * scale, bf_hi(a[3]) * scale, r0 * 0x9e3779b1u + 0x7f4a7c15u);
            o[2] = bd_pack4_e5m2_sr(bf_lo(b[0]) * scale, bf_hi(b[0]) * scale, bf_lo(b[1]) * scale, bf_hi(b[1]) * scale, r2);
            o[3] = bd_pack4_e5m2_sr(bf_lo(b[2]) * scale, bf_hi(b[2]) * scale, bf_lo(b[3]) * scale, bf_hi(b[3]) * scale, r2 * 0x9e3779b1u + 0x7f4a7c15u);
            q[i] = o;
            continue;
        }
        o[0] = pack4_bf8(bf_lo(a[0]) * scale, bf_hi(a[0]) * scale, bf_lo(a[1]) * scale, bf_hi(a[1]) * scale);
        o[1] = pack4_bf8(bf_lo(a[2]) * scale, bf_hi(a[2]) * scale, bf_lo(a[3]) * scale, bf_hi(a[3]) * scale);
        o[2] = pack4_bf8(bf_lo(b[0]) * scale, bf_hi(b[0]) * scale, bf_lo(b[1]) * scale, bf_hi(b[1]) * scale);
        o[3] = pack4_bf8(bf_lo(b[2]) * scale, bf_hi(b[2]) * scale, bf_lo(b[3]) * scale, bf_hi(b[3]) * scale);
        q[i] = o;
    }
}

// max |x| of a bf16 tensor into out[0] (atomic max on the bit pattern: non-negative floats order like unsigned integers; the caller
// zeroes out[0] first; NaNs are skipped).  Feeds the delayed scaling of the e5m2 gradients.
__global__ __launch_bounds__(256) void absmax_bf16_kernel(const bf16_raw* __restrict__ x, long long n8, unsigned* __restrict__ out) {
    unsigned m = 0u;                          // max over |x| as bf16 bit patterns (sign cleared): same order as the values
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const u32x4_t a = *reinterpret_cast<const u32x4_t*>(x + i * 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned lo = a[k] & 0x7fffu, hi = (a[k] >> 16) & 0x7fffu;
            if (lo <= 0x7f80u) m = max(m, lo);            // skip NaNs (exponent all ones, mantissa != 0); inf counts
            if (hi <= 0x7f80u) m = max(m, hi);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m << 16);       // fp32 bit pattern of the bf16 value
}

// data-gradient weights: one workgroup per INPUT channel ci: s = max over (tap, co) of |w[co][tap][ci] * row_scale[co]| / 448,
// wq_t[ci][tap][co] = e4m3(w * row_scale / s), out_scale[ci] = s / grad_scale
__global__ __launch_bounds__(256) void weight_pack_fp8_t_kernel(const float* __restrict__ w, const float* __restrict__ row_scale, int Cout,
                                                               int RS, int Cin, float grad_scale, unsigned char* __restrict__ wq_t,
                                                               float* __restrict__ out_scale) {
    __shared__ float red[4];
    const int ci = blockIdx.x;
    const int n = RS * Cout;
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int tap = i / Cout, co = i - tap * Cout;
        m = fmaxf(m, fabsf(w[((long long)co * RS + tap) * Cin + ci] * (row_scale ? row_scale[co] : 1.f)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float s = m > 0.f ? m / FP8_MAX : 1.f;
    const float inv = 1.f / s;
    for (int i = threadIdx.x * 4; i < n; i += 1024) {          // Cout % 4 == 0: the four entries share a tap
        const int tap = i / Cout, co = i - tap * Cout;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = w[((long long)(co + k) * RS + tap) * Cin + ci] * (row_scale ? row_scale[co + k] : 1.f) * inv;
        *reinterpret_cast<unsigned*>(wq_t + (long long)ci * n + i) = pack4_fp8(v[0], v[1], v[2], v[3]);
    }
    if (threadIdx.x == 0) out_scale[ci] = s / grad_scale;
}

// one workgroup per output channel: s = max |w * row_scale| / 448, wq = e4m3(w * row_scale / s), out_scale = s / act_scale
__global__ __launch_bounds__(256) void weight_pack_fp8_kernel(const float* __restrict__ w, const float* __restrict__ row_scale, int row_len,
                                                             float act_scale, unsigned char* __restrict__ wq, float* __restrict__ out_scale) {
    __shared__ float red[4];
    const int co = blockIdx.x;
    const float rs = row_scale ? row_scale[co] : 1.f;
    const float* src = w + (long long)co * row_len;
    float m = 0.f;
    for (int i = threadIdx.x; i < row_len; i += 256) m = fmaxf(m, fabsf(src[i] * rs));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float s = m > 0.f ? m / FP8_MAX : 1.f;
    const float inv = rs / s;
    for (int i = threadIdx.x * 4; i < row_len; i += 1024) {       // row_len % 4 == 0
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(src + i);
        *reinterpret_cast<unsigned*>(wq + (long long)co * row_len + i) = pack4_fp8(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
    }
    if (threadIdx.x == 0) out_scale[co] = s / act_scale;
}

struct F8Params {
    IgemmParams g;            // src = e4m3 activations, w = e4m3 weights (byte pointers carried as bf16_raw*), dst bf16
    const float* wscale;
    unsigned char* dst8;      // optional e4m3 twin of the output (dst * q_scale), read by a following fp8 convolution
    float q_scale;
};

__global__ __launch_bounds__(256, 2) void conv_fp8_kernel(const F8Params fp) {
    const IgemmParams& p = fp.g;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave >> 1, wp = wave & 1;

    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / p.n_tiles;
    const int tile_n = bid - tile_m * p.n_tiles;
    const int m0 = tile_m * TILE_P, co0 = tile_n * TILE_C;
    const int chunk = tid & 7;            // 16-byte chunk of the 128-byte row
    const int row0 = tid >> 3;            // rows row0 + 32 i

    // ---- per-row pixel decode (forward: pix = output pixel, source = conv input) ----
    int b_py[PASSES], b_px[PASSES], b_hs[PASSES], b_ws[PASSES], b_base[PASSES];
    const int RS = p.R * p.S;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
        const int m = m0 + row0 + i * 32;
        b_base[i] = -1; b_py[i] = 0; b_px[i] = 0; b_hs[i] = 0; b_ws[i] = 0;
        if (m < p.M) {
            const SubSeg ss = p.sub[find_sub(p, m)];
            const int local = m - ss.m_start;
            int n, rem, yy, xx;
            fast_divmod(local, ss.Hs * ss.Ws, ss.inv_per_img, n, rem);
            fast_divmod(rem, ss.Ws, ss.inv_ws, yy, xx);
            b_base[i] = n * p.src_pix_per_img + ss.src_off;
            b_hs[i] = ss.Hsrc; b_ws[i] = ss.Wsrc;
            b_py[i] = yy * p.stride - p.pad; b_px[i] = xx * p.stride - p.pad;
        }
    }
    const int kblocks = (p.CK + KB - 1) / KB;
    const int nsteps = RS * kblocks;

    constexpr unsigned X_NONE = 0x80000000u;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.src), 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.w), 0, p.w_bytes, 0x00020000);
    unsigned a_voff[PASSES], b_voff[PASSES];
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
        const int lrow = row0 + i * 32;
        const int rho = lrow & 15;
        const int co = co0 + (lrow & 64) + 32 * ((lrow >> 5) & 1) + 8 * (rho >> 2) + 4 * ((lrow >> 4) & 1) + (rho & 3);
        a_voff[i] = co < p.CO ? (unsigned)(co * RS * p.CK + chunk * 16) : X_NONE;
        b_voff[i] = X_NONE;
    }
    u32x4_t ra[PASSES], rb[PASSES];
    int cur_tap = -1, cur_kb = kblocks;
    auto stage_load = [&]() {
        if (cur_kb == kblocks) {           // next filter tap (workgroup-uniform): the source pixel of every staged row moves
            cur_kb = 0;
            ++cur_tap;
            const int tap_r = cur_tap / p.S, tap_s = cur_tap - tap_r * p.S;
#pragma unroll
            for (int i = 0; i < PASSES; ++i) {
                const int sy = b_py[i] + tap_r, sx = b_px[i] + tap_s;
                const bool ok = b_base[i] >= 0 && sy >= 0 && sx >= 0 && sy < b_hs[i] && sx < b_ws[i];
                b_voff[i] = ok ? (unsigned)((b_base[i] + sy * b_ws[i] + sx) * p.CK + chunk * 16) : X_NONE;
            }
        }
        int so_a = cur_tap * p.CK + cur_kb * KB, so_b = cur_kb * KB;
        asm volatile("" : "+s"(so_a), "+s"(so_b));
        const bool dead = cur_kb * KB + chunk * 16 >= p.CK;       // channel tail (CK % 16 == 0): zero-fill
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, dead ? X_NONE : a_voff[i], so_a, 0);
            rb[i] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, dead ? X_NONE : b_voff[i], so_b, 0);
        }
        ++cur_kb;
    };
    auto stage_write = [&](int buf) {
        unsigned char* At = smem + buf * 2 * TILE_BYTES;
        unsigned char* Bt = At + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int row = row0 + i * 32;
            *reinterpret_cast<u32x4_t*>(At + row * PITCH + chunk * 16) = ra[i];
            *reinterpret_cast<u32x4_t*>(Bt + row * PITCH + chunk * 16) = rb[i];
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int frag_row = lane & 15;
    const int frag_k = (lane >> 4) * 32;           // byte offset of the lane's 32 consecutive k
    const int one = 0x7f7f7f7f;                    // E8M0 block scales: 2^0
    auto compute = [&](int buf) {
        const unsigned char* At = smem + buf * 2 * TILE_BYTES;
        const unsigned char* Bt = At + TILE_BYTES;
        i32x8_t a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned char* r = At + (wc * 64 + i * 16 + frag_row) * PITCH + frag_k;
            const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(r), hi = *reinterpret_cast<const u32x4_t*>(r + 16);
            a[i] = (i32x8_t){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned char* r = Bt + (wp * 64 + j * 16 + frag_row) * PITCH + frag_k;
            const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(r), hi = *reinterpret_cast<const u32x4_t*>(r + 16);
            b[j] = (i32x8_t){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i][j], 0, 0, 0, one, 0, one);
    };

    if (nsteps > 0) { stage_load(); stage_write(0); }
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < nsteps; ++t) {
        const bool more = t + 1 < nsteps;
        if (more) stage_load();
        compute(cur);
        if (more) stage_write(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue (lane (cg = lane>>4) holds channels cbase + 32*half + 0..7 of pixel m0 + wp*64 + j*16 + (lane&15)) ----
    const int cg = lane >> 4;
    const bool do_relu = p.flags & BD_EPI_RELU;
    const bool add_before = (p.flags & BD_EPI_ADD_BEFORE) && p.add;
    const int cbase = co0 + wc * 64 + 8 * cg;
    float bias[16], scl[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { bias[k] = 0.f; scl[k] = 0.f; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = cbase + 32 * (q >> 1) + 4 * (q & 1);
        if (c < p.CO) {
            const f32x4_t sv = *reinterpret_cast<const f32x4_t*>(fp.wscale + c);
            scl[4 * q] = sv[0]; scl[4 * q + 1] = sv[1]; scl[4 * q + 2] = sv[2]; scl[4 * q + 3] = sv[3];
            if (p.bias) {
                const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(p.bias + c);
                bias[4 * q] = bv[0]; bias[4 * q + 1] = bv[1]; bias[4 * q + 2] = bv[2]; bias[4 * q + 3] = bv[3];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wp * 64 + j * 16 + (lane & 15);
        if (m >= p.M) continue;
        int dstpix;
        if (p.linear_dst) dstpix = m;
        else {
            const SubSeg ss = p.sub[find_sub(p, m)];
            const int local = m - ss.m_start;
            int n, rem, yy, xx;
            fast_divmod(local, ss.Hs * ss.Ws, ss.inv_per_img, n, rem);
            fast_divmod(rem, ss.Ws, ss.inv_ws, yy, xx);
            dstpix = n * p.dst_pix_per_img + ss.dst_off + yy * ss.Wd + xx;
        }
        const long long base = (long long)dstpix * p.CO + cbase;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (cbase + 32 * half >= p.CO) continue;
            const long long idx = base + 32 * half;
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = acc[2 * half + (k >> 2)][j][k & 3] * scl[8 * half + k] + bias[8 * half + k];
            if (add_before) {
                const u32x4_t av = *reinterpret_cast<const u32x4_t*>(p.add + idx);
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[2 * k] += bf_lo(av[k]); v[2 * k + 1] += bf_hi(av[k]); }
            }
            if (do_relu) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
            }
            u32x4_t o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = pack_bf2(v[2 * k], v[2 * k + 1]);
            *reinterpret_cast<u32x4_t*>(p.dst + idx) = o;
            if (fp.dst8) {
                u32x2_t o8;
                o8[0] = pack4_fp8(v[0] * fp.q_scale, v[1] * fp.q_scale, v[2] * fp.q_scale, v[3] * fp.q_scale);
                o8[1] = pack4_fp8(v[4] * fp.q_scale, v[5] * fp.q_scale, v[6] * fp.q_scale, v[7] * fp.q_scale);
                *reinterpret_cast<u32x2_t*>(fp.dst8 + idx) = o8;
            }
        }
    }
}

}  // namespace

BD_KNOB int g_fp8_patch = 1;        // bd_conv_desc.route[3] - 1: 0 = every shape through the generic per-tap kernel (A/B)

int bd_conv3x3_pp8_launch(const bd_conv_desc* d, int mode, const void* xq, const void* wq, const float* wscale, const float* bias,
                          const void* add, const void* mask, void* y, void* y8, float q_scale, int flags, hipStream_t stream);

extern "C" {

int bd_quantize_fp8(const void* x_bf16, int64_t n, float scale, void* q, bd_stream_t stream) {
    BD_REQUIRE(x_bf16 && q, "quantize_fp8: null pointer");
    BD_REQUIRE(n % 16 == 0 && scale > 0.f, "quantize_fp8: n %% 16 != 0 or scale <= 0");
    if (n == 0) return BD_OK;
    const long long n16 = n / 16;
    const int grid = (int)std::min<long long>(cdiv64(n16, 256), 256 * 16);
    hipLaunchKernelGGL(quantize_fp8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)x_bf16, n16, scale, (u32x4_t*)q);
    BD_CHECK_LAUNCH("bd_quantize_fp8");
    return BD_OK;
}

int bd_weight_pack_fp8(const float* w, const float* row_scale, int Cout, int RS, int Cin, float act_scale, void* wq, float* wscale,
                       bd_stream_t stream) {
    BD_REQUIRE(w && wq && wscale, "weight_pack_fp8: null pointer");
    BD_REQUIRE(Cout > 0 && RS > 0 && Cin > 0 && (RS * Cin) % 4 == 0 && act_scale > 0.f, "weight_pack_fp8: bad sizes");
    hipLaunchKernelGGL(weight_pack_fp8_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, w, row_scale, RS * Cin, act_scale,
                       (unsigned char*)wq, wscale);
    BD_CHECK_LAUNCH("bd_weight_pack_fp8");
    return BD_OK;
}

int bd_quantize_bf8(const void* x_bf16, int64_t n, float scale, void* q, uint32_t sr_seed, bd_stream_t stream) {
    BD_REQUIRE(x_bf16 && q, "quantize_bf8: null pointer");
    BD_REQUIRE(n % 16 == 0 && scale > 0.f, "quantize_bf8: n %% 16 != 0 or scale <= 0");
    if (n == 0) return BD_OK;
    const long long n16 = n / 16;
    const int grid = (int)std::min<long long>(cdiv64(n16, 256), 256 * 16);
    hipLaunchKernelGGL(quantize_bf8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)x_bf16, n16, scale, (u32x4_t*)q, sr_seed);
    BD_CHECK_LAUNCH("bd_quantize_bf8");
    return BD_OK;
}

int bd_absmax_bf16(const void* x_bf16, int64_t n, float* out, bd_stream_t stream) {
    BD_REQUIRE(x_bf16 && out && n >= 0 && n % 8 == 0, "absmax_bf16: n must be a multiple of 8");
    if (n == 0) return BD_OK;
    const long long n8 = n / 8;
    const int grid = (int)std::min<long long>(cdiv64(n8, 256), 256 * 8);
    hipLaunchKernelGGL(absmax_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)x_bf16, n8, (unsigned*)out);
    BD_CHECK_LAUNCH("bd_absmax_bf16");
    return BD_OK;
}

int bd_weight_pack_fp8_t(const float* w, const float* row_scale, int Cout, int RS, int Cin, float grad_scale, void* wq_t, float* wscale_t,
                         bd_stream_t stream) {
    BD_REQUIRE(w && wq_t && wscale_t, "weight_pack_fp8_t: null pointer");
    BD_REQUIRE(Cout > 0 && RS > 0 && Cin > 0 && Cout % 4 == 0 && grad_scale > 0.f, "weight_pack_fp8_t: bad sizes");
    hipLaunchKernelGGL(weight_pack_fp8_t_kernel, dim3(Cin), dim3(256), 0, (hipStream_t)stream, w, row_scale, Cout, RS, Cin, grad_scale,
                       (unsigned char*)wq_t, wscale_t);
    BD_CHECK_LAUNCH("bd_weight_pack_fp8_t");
    return BD_OK;
}

int bd_conv2d_dgrad_fp8(const bd_conv_desc* d, const void* g8, const void* wq_t, const float* wscale_t, const void* add, const void* mask,
                        void* dx, void* dx8, float q_scale, int flags, bd_stream_t stream) {
    BD_REQUIRE(d && g8 && wq_t && wscale_t && dx, "conv2d_dgrad_fp8: null pointer");
    BD_ROUTE(d);
    BD_REQUIRE(!(flags & BD_EPI_RELU), "conv2d_dgrad_fp8: BD_EPI_RELU is a forward-only flag");
    BD_REQUIRE(d->Cout % 16 == 0 && d->Cin % 8 == 0, "conv2d_dgrad_fp8: Cout %% 16 and Cin %% 8 must be 0 (got %d, %d)", d->Cout, d->Cin);
    if (bd_conv3x3_pp8_launch(d, 1, g8, wq_t, wscale_t, nullptr, add, mask, dx, dx8, q_scale, flags, (hipStream_t)stream) != 0) {
        bd_set_error("conv2d_dgrad_fp8: only 3x3 / stride 1 / pad 1 with Cin > 128 (the fp8 patch kernel); use bd_conv2d_dgrad");
        return BD_EINVAL;
    }
    BD_CHECK_LAUNCH("bd_conv2d_dgrad_fp8");
    return BD_OK;
}

int bd_conv2d_fwd_fp8_ex(const bd_conv_desc* d, const void* xq, const void* wq, const float* wscale, const float* bias, const void* add,
                         void* y, void* y8, float q_scale, int flags, bd_stream_t stream);

int bd_conv2d_fwd_fp8(const bd_conv_desc* d, const void* xq, const void* wq, const float* wscale, const float* bias, const void* add,
                      void* y, int flags, bd_stream_t stream) {
    return bd_conv2d_fwd_fp8_ex(d, xq, wq, wscale, bias, add, y, nullptr, 1.f, flags, stream);
}

int bd_conv2d_fwd_fp8_ex(const bd_conv_desc* d, const void* xq, const void* wq, const float* wscale, const float* bias, const void* add,
                         void* y, void* y8, float q_scale, int flags, bd_stream_t stream) {
    BD_REQUIRE(d && xq && wq && wscale && y, "conv2d_fwd_fp8: null pointer");
    BD_ROUTE(d);
    BD_REQUIRE(d->nseg >= 1 && d->nseg <= MAX_SUB && (d->stride == 1 || d->stride == 2) && d->R * d->S <= 32, "conv2d_fwd_fp8: bad descriptor");
    BD_REQUIRE(d->Cin % 16 == 0 && d->Cout % 8 == 0, "conv2d_fwd_fp8: Cin %% 16 and Cout %% 8 must be 0 (got %d, %d)", d->Cin, d->Cout);
    BD_REQUIRE(!(flags & (BD_EPI_MASK | BD_EPI_ADD_AFTER)), "conv2d_fwd_fp8: forward flags only");
    if (g_fp8_patch && bd_conv3x3_pp8_launch(d, 0, xq, wq, wscale, bias, add, nullptr, y, y8, q_scale, flags, (hipStream_t)stream) == 0) {
        BD_CHECK_LAUNCH("bd_conv2d_fwd_fp8(patch)");
        return BD_OK;
    }
    F8Params fp{};
    fp.dst8 = (unsigned char*)y8; fp.q_scale = q_scale;
    IgemmParams& p = fp.g;
    fp.wscale = wscale;
    p.src = (const bf16_raw*)xq; p.w = (const bf16_raw*)wq; p.bias = bias; p.add = (const bf16_raw*)add; p.mask = nullptr; p.dst = (bf16_raw*)y;
    p.CK = d->Cin; p.CO = d->Cout; p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad = d->pad; p.mode = 0; p.flags = flags;
    p.nsub = d->nseg;
    long long m = 0;
    for (int s = 0; s < d->nseg; ++s) {
        BD_REQUIRE((d->Hi[s] + 2 * d->pad - d->R) / d->stride + 1 == d->Ho[s] && (d->Wi[s] + 2 * d->pad - d->S) / d->stride + 1 == d->Wo[s],
                   "conv2d_fwd_fp8: level %d output size inconsistent", s);
        SubSeg& ss = p.sub[s];
        ss.m_start = (int)m;
        ss.Hs = d->Ho[s]; ss.Ws = d->Wo[s]; ss.y0 = 0; ss.x0 = 0; ss.step = 1;
        ss.Wd = d->Wo[s]; ss.dst_off = d->out_off[s];
        ss.Hsrc = d->Hi[s]; ss.Wsrc = d->Wi[s]; ss.src_off = d->in_off[s];
        ss.inv_per_img = 1.0f / (float)(ss.Hs * ss.Ws); ss.inv_ws = 1.0f / (float)ss.Ws;
        m += (long long)d->N * d->Ho[s] * d->Wo[s];
    }
    BD_REQUIRE(m < (1ll << 24), "conv2d_fwd_fp8: too many pixels (2^24 limit of the fast index decode)");
    p.linear_dst = (d->nseg == 1 && d->out_off[0] == 0 && d->out_pix_per_img == d->Ho[0] * d->Wo[0]) ? 1 : 0;
    p.M = (int)m;
    p.src_pix_per_img = d->in_pix_per_img; p.dst_pix_per_img = d->out_pix_per_img;
    const long long sb = (long long)d->N * d->in_pix_per_img * d->Cin, wb = (long long)d->Cout * d->R * d->S * d->Cin;
    BD_REQUIRE(sb < 0x7fffffffll && wb < 0x7fffffffll, "conv2d_fwd_fp8: tensors must be < 2 GB");
    p.src_bytes = (unsigned)sb; p.w_bytes = (unsigned)wb;
    p.m_tiles = cdiv(p.M, TILE_P); p.n_tiles = cdiv(p.CO, TILE_C);
    const size_t lds = 4 * TILE_BYTES;
    BD_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_fp8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    bd_note_kernel("conv_fp8_kernel");
    hipLaunchKernelGGL(conv_fp8_kernel, dim3(p.m_tiles * p.n_tiles), dim3(256), lds, (hipStream_t)stream, fp);
    BD_CHECK_LAUNCH("bd_conv2d_fwd_fp8");
    return BD_OK;
}

}  // extern "C"
