// Weight gradient of 3x3 / stride-1 / pad-1 convolutions on ONE-BYTE operands (BASELINE config 5): the nine-tap patch kernel of
// conv_wgrad3x3.hip on v_mfma_scale_f32_16x16x128_f8f6f4.
//
//   dW[co][tap][ci] = sum_pix G[pix][co] * X[pix + shift(tap)][ci]        X = e4m3 twin of the activations (x * act_scale),
//                                                                         G = e5m2 twin of the output gradient (g * grad_scale)
//
// The reduction index is the pixel, the slow index of both NHWC twins, and the MFMA wants 32 consecutive k per lane: both operands
// are staged [pixel][channel] as they lie in HBM and fetched with gfx950's byte transposing read ds_read_b64_tr_b8 (probed by
// scripts/exp/ds_read_tr8_layout.hip: in a 16-lane group lane 2q + p supplies the address of 8 bytes of row q -- channels 8p .. 8p + 7
// -- and lane i receives channel i of rows 0 .. 7: eight pixels of one channel, pixel-contiguous).  One K = 128 MFMA covers an
// 8 x 16 output patch; k = 32 g4 + 8 t + j  <->  patch pixel 32 t + 8 g4 + j (row 2t + (g4 >> 1), column 8 (g4 & 1) + j), the same
// permutation for both operands, chosen so that the two 16-lane groups of a half-wave read 16 consecutive LDS rows: with 64-byte rows
// at an 80-byte pitch that is every bank exactly once.  A tap only shifts the row of the X fragment by r * 18 + s rows -- an address
// immediate.  A workgroup stages, per K step, the patch of G (128 pixels x 32 co) and the 10 x 18 input patch of X (180 pixels x 64
// ci) -- since round 2's last revision a 64 ci x 32 co tile per FOUR-wave workgroup, two independent workgroups per CU; each wave 16 ci x 32 co of all nine taps (72 fp32 accumulators: the bf16 kernel's 16 x 64 wave tile plus eight-dword
// fragments spilled, and a reloaded spill waits in vmcnt order behind the prefetch): 18 MFMAs of K = 128 per wave and step.  Split over patches into fp32 slabs, summed
// by conv_wgrad.hip's fixed-order reduce; products of e4m3 x e5m2 are exact in fp32, the 1 / (act_scale * grad_scale) factor is
// applied when the slab is written.  The bias gradient is not fused here (it would be a sum of e5m2 values): bd_colsum_bf16 on the
// bf16 gradient.
#include "common.h"

namespace {

constexpr int F8_PH = 8, F8_PW = 16;                 // output patch = 128 pixels = one MFMA K
constexpr int F8_XW = F8_PW + 2, F8_XH = F8_PH + 2;  // input patch 10 x 18
constexpr int F8_XROWS = F8_XW * F8_XH;              // 180
constexpr int F8_GROWS = F8_PH * F8_PW;              // 128
constexpr int F8_TILE = 64;                          // input channels per tile
constexpr int F8_TCO = 32;                           // output channels per tile: FOUR waves of 16 ci x 32 co, two independent workgroups per CU
                                                     // (the eight-wave 64 x 64 form was one workgroup per CU: both waves of a SIMD at the same barrier)
#ifndef F8_PITCH_OVERRIDE
#define F8_PITCH_OVERRIDE 80
#endif
constexpr int F8_PITCH = F8_PITCH_OVERRIDE;          // 64 B of channels + 16 B pad: 16 consecutive rows x 16 B hit all 64 banks
constexpr int F8_X_BYTES = F8_XROWS * F8_PITCH;      // 14 400
constexpr int F8_GPITCH = 48;                         // 32 B of channels + 16 B pad (12 dwords: 16 consecutive rows x 16 B hit all 64 banks)
constexpr int F8_G_BYTES = F8_GROWS * F8_GPITCH;     // 6 144
constexpr int F8_BUF = F8_X_BYTES + F8_G_BYTES;      // 20 544
#ifndef F8_WGS
#define F8_WGS 2
#endif
constexpr int F8_XP = 3, F8_GP = 1;                  // staging passes: 720 / 256 chunks of 16 B over 256 threads

struct F8Seg { int patch_start, H, W, pw, in_off, out_off; };

struct WF8Params {
    const unsigned char* x;      // e4m3 [N][in_ppi][Cin]
    const unsigned char* g;      // e5m2 [N][out_ppi][Cout]
    float* slab;
    float inv_scale;
    int Cin, Cout, N, nseg;
    int in_ppi, out_ppi;
    unsigned x_bytes, g_bytes;
    int patches_per_img, total_patches, patches_per_split;
    int ci_tiles, co_tiles;
    F8Seg seg[BD_MAX_SEGS];
};

typedef __attribute__((ext_vector_type(8))) int i32x8_w8_t;
typedef __attribute__((ext_vector_type(2))) int i32x2_w8_t;

__global__ __launch_bounds__(256, F8_WGS) void conv_wgrad3x3_fp8_kernel(const WF8Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cih = wave;                             // this wave's 16 input channels of the 64 x 32 tile (all 32 output channels)
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap (conv_wgrad3x3.hip): the tiles of one split share its patches in one XCD's L2
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tiles = p.ci_tiles * p.co_tiles;
    const int split = bid / tiles;
    bid -= split * tiles;
    const int ci_tile = bid / p.co_tiles, co_tile = bid - ci_tile * p.co_tiles;
    const int ci0 = ci_tile * F8_TILE, co0 = co_tile * F8_TCO;
    const int pbeg = split * p.patches_per_split;
    int pend = pbeg + p.patches_per_split;
    if (pend > p.total_patches) pend = p.total_patches;

    // staging slots: X chunk c = tid + 256 k -> LDS row c >> 2, 16-channel chunk tid & 3; G: row tid >> 1, 16-channel chunk tid & 1
    const int chunk = tid & 3, gchunk = tid & 1;
    const bool x_cok = ci0 + chunk * 16 < p.Cin, g_cok = co0 + gchunk * 16 < p.Cout;     // Cin, Cout are multiples of 16
    u32x4_t rx[F8_XP], rg[F8_GP];
    constexpr unsigned X_NONE = 0x80000000u;          // past the end of either tensor (the host checks < 2 GB): reads as zeros
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.g), 0, p.g_bytes, 0x00020000);

    // Patch cursor + one-instruction loads for interior patches, as conv_wgrad3x3.hip: a split walks consecutive patch ids (increment
    // with carries instead of two divisions, a level search and a descriptor fetch per patch), the per-thread byte offsets are per-level
    // constants with X_NONE folded in, and the patch origin rides in the buffer instruction's scalar offset.  (Measured first on the
    // eight-wave form of this kernel -- one workgroup per CU, both waves of a SIMD at the same barrier, nothing to hide the decode
    // behind: +12-19 % -- and kept in the four-wave form: the decode was the larger part of a patch step of 18 MFMAs of 32 cycles.)
    int c_n = 0, c_s = 0, c_by = 0, c_bx = 0, c_rows = 1;
    F8Seg sg = p.seg[0];
    unsigned x_vec[F8_XP], g_vec[F8_GP];
    auto level_vectors = [&]() {
#pragma unroll
        for (int k = 0; k < F8_XP; ++k) {
            const int row = (tid + 256 * k) >> 2;
            const int iy = row / F8_XW, ix = row - iy * F8_XW;
            x_vec[k] = (row < F8_XROWS && x_cok) ? (unsigned)((iy * sg.W + ix) * p.Cin + chunk * 16) : X_NONE;
        }
#pragma unroll
        for (int k = 0; k < F8_GP; ++k) {
            const int row = tid >> 1;
            g_vec[k] = g_cok ? (unsigned)(((row >> 4) * sg.W + (row & 15)) * p.Cout + gchunk * 16) : X_NONE;
        }
    };
    auto seek = [&](int pid) {
        c_n = pid / p.patches_per_img;
        const int rem = pid - c_n * p.patches_per_img;
        c_s = 0;
#pragma unroll
        for (int k = 1; k < BD_MAX_SEGS; ++k)
            if (k < p.nseg && rem >= p.seg[k].patch_start) c_s = k;
        sg = p.seg[c_s];
        const int local = rem - sg.patch_start;
        c_by = local / sg.pw; c_bx = local - c_by * sg.pw;
        c_rows = (sg.H + F8_PH - 1) / F8_PH;
        level_vectors();
    };
    auto advance = [&]() {
        if (++c_bx < sg.pw) return;
        c_bx = 0;
        if (++c_by < c_rows) return;
        c_by = 0;
        if (p.nseg == 1) { ++c_n; return; }
        if (++c_s == p.nseg) { c_s = 0; ++c_n; }
        sg = p.seg[c_s];
        c_rows = (sg.H + F8_PH - 1) / F8_PH;
        level_vectors();
    };
    auto stage_load = [&]() {
        const int y0 = c_by * F8_PH, x0 = c_bx * F8_PW;
        const int ys = y0 - 1, xs = x0 - 1;
        const int xorg = (c_n * p.in_ppi + sg.in_off + ys * sg.W + xs) * p.Cin + ci0;       // may be negative; valid sums are not
        const int gorg = (c_n * p.out_ppi + sg.out_off + y0 * sg.W + x0) * p.Cout + co0;
        if (ys >= 0 && xs >= 0 && ys + F8_XH <= sg.H && xs + F8_XW <= sg.W) {                 // workgroup-uniform: every pixel of both patches exists
#pragma unroll
            for (int k = 0; k < F8_XP; ++k) rx[k] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, x_vec[k], xorg, 0);
#pragma unroll
            for (int k = 0; k < F8_GP; ++k) rg[k] = __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, g_vec[k], gorg, 0);
            return;
        }
#pragma unroll
        for (int k = 0; k < F8_XP; ++k) {
            const int row = (tid + 256 * k) >> 2;
            const int iy = row / F8_XW, ix = row - iy * F8_XW;
            const int y = ys + iy, x = xs + ix;
            const bool ok = y >= 0 && x >= 0 && y < sg.H && x < sg.W;       // unused slots / channel tails: x_vec is X_NONE already
            rx[k] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (ok && x_vec[k] != X_NONE) ? (unsigned)xorg + x_vec[k] : X_NONE, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < F8_GP; ++k) {
            const int row = tid >> 1;
            const bool ok = y0 + (row >> 4) < sg.H && x0 + (row & 15) < sg.W;
            rg[k] = __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, (ok && g_vec[k] != X_NONE) ? (unsigned)gorg + g_vec[k] : X_NONE, 0, 0);
        }
    };
    auto stage_write = [&](int buf) {
        unsigned char* Xt = smem + buf * F8_BUF;
        unsigned char* Gt = Xt + F8_X_BYTES;
#pragma unroll
        for (int k = 0; k < F8_XP; ++k) {
            const int row = (tid + 256 * k) >> 2;
            if (row < F8_XROWS) *reinterpret_cast<u32x4_t*>(Xt + row * F8_PITCH + chunk * 16) = rx[k];
        }
#pragma unroll
        for (int k = 0; k < F8_GP; ++k) {
            const int row = tid >> 1;
            *reinterpret_cast<u32x4_t*>(Gt + row * F8_GPITCH + gchunk * 16) = rg[k];
        }
    };

    f32x4_t acc[9][2];     // wave tile: 16 ci x 32 co per tap
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[t][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int g4 = lane >> 4, idx = lane & 15;
    const int tr_q = idx >> 1, tr_p = idx & 1;        // this lane supplies row tr_q, channels 8 tr_p .. + 7 of the group's 8 x 16 block
    // lane-constant parts of the fragment addresses: patch row (g4 >> 1) of the read's row pair, column 8 (g4 & 1) + tr_q
    const int x_lane_off = ((g4 >> 1) * F8_XW + 8 * (g4 & 1) + tr_q) * F8_PITCH + cih * 16 + 8 * tr_p;
    const int g_lane_off = ((g4 >> 1) * F8_PW + 8 * (g4 & 1) + tr_q) * F8_GPITCH + 8 * tr_p;
    const int one = 0x7f7f7f7f;                       // E8M0 block scales: 2^0

    // one operand fragment = the lane's 32 k = four transposing reads (t = 0 .. 3: patch rows 2t, 2t + 1), 8 bytes each
    auto frag = [&](const unsigned char* a0, int row_bytes) -> i32x8_w8_t {       // row_bytes = patch-row pitch in bytes
        i32x8_w8_t f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const i32x2_w8_t v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2_w8_t*)(a0 + 2 * t * row_bytes));
            f[2 * t] = v[0]; f[2 * t + 1] = v[1];
        }
        return f;
    };

    auto compute = [&](int buf) {
        const unsigned char* Xt = smem + buf * F8_BUF + x_lane_off;
        const unsigned char* Gt = smem + buf * F8_BUF + F8_X_BYTES + g_lane_off;
        __builtin_amdgcn_s_setprio(1);
        i32x8_w8_t b[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = frag(Gt + j * 16, F8_PW * F8_GPITCH);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int r = t / 3, s = t % 3;
            const i32x8_w8_t a = frag(Xt + (r * F8_XW + s) * F8_PITCH, F8_XW * F8_PITCH);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[t][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b[j], acc[t][j], 0, 1, 0, one, 0, one);   // A e4m3, B e5m2
        }
        __builtin_amdgcn_s_setprio(0);
    };

    if (pbeg < pend) {
        seek(pbeg);
        stage_load();
        advance();
        stage_write(0);
    }
    __syncthreads();
    int cur = 0;
    for (int pid = pbeg; pid < pend; ++pid) {
        const bool more = pid + 1 < pend;
        if (more) { stage_load(); advance(); }
        compute(cur);
        if (more) stage_write(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    float* slab = p.slab + (long long)split * p.Cout * 9 * p.Cin;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = co0 + j * 16 + idx;
        if (co >= p.Cout) continue;
        const int ci = ci0 + cih * 16 + g4 * 4;
        if (ci >= p.Cin) continue;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            f32x4_t v = acc[t][j];
            v *= p.inv_scale;
            *reinterpret_cast<f32x4_t*>(slab + ((long long)co * 9 + t) * p.Cin + ci) = v;
        }
    }
}

int f8_patches(const bd_conv_desc* d, int* ppi_out) {
    int ppi = 0;
    for (int s = 0; s < d->nseg; ++s) ppi += cdiv(d->Ho[s], F8_PH) * cdiv(d->Wo[s], F8_PW);
    if (ppi_out) *ppi_out = ppi;
    return ppi * d->N;
}

int f8_splits(const bd_conv_desc* d) {
    const int total = f8_patches(d, nullptr);
    const int tiles = cdiv(d->Cin, F8_TILE) * cdiv(d->Cout, F8_TCO);
    int splits = 512 / tiles;                          // workgroups per launch: two per CU (the bf16 kernel's measured optimum)
    if (splits < 1) splits = 1;
    if (splits > total) splits = total;
    const int per = cdiv(total, splits);
    return cdiv(total, per);
}

bool f8_ok(const bd_conv_desc* d) {
    if (!(d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1) || d->nseg < 1 || d->nseg > BD_MAX_SEGS) return false;
    for (int s = 0; s < d->nseg; ++s)
        if (d->Hi[s] != d->Ho[s] || d->Wi[s] != d->Wo[s]) return false;
    if (d->Cin % 16 != 0 || d->Cout % 16 != 0) return false;
    return (long long)d->N * d->in_pix_per_img * d->Cin < 0x7fffffffll && (long long)d->N * d->out_pix_per_img * d->Cout < 0x7fffffffll;
}

}  // namespace

void bd_wgrad_reduce_launch(const float* slab, int splits, long long n, int row_len, const float* row_scale, float* dw, int accumulate,
                            hipStream_t stream);       // conv_wgrad.hip

extern "C" size_t bd_conv2d_wgrad_fp8_workspace_bytes(const bd_conv_desc* d) {
    BdRouteScope rs__(d);
    if (rs__.rc != BD_OK) return 0;
    if (!d || !f8_ok(d)) return 0;
    return (size_t)f8_splits(d) * d->Cout * 9 * d->Cin * sizeof(float);
}

extern "C" int bd_conv2d_wgrad_fp8(const bd_conv_desc* d, const void* x8, const void* g8, float inv_scale, const float* row_scale,
                                   float* dw, int accumulate, void* ws, size_t ws_bytes, bd_stream_t stream) {
    BD_ROUTE(d);
    BD_REQUIRE(d && x8 && g8 && dw && ws, "conv2d_wgrad_fp8: null pointer");
    BD_REQUIRE(f8_ok(d), "conv2d_wgrad_fp8: 3x3 / stride 1 / pad 1 with Cin %% 16 == 0 and Cout %% 16 == 0 only (tensors < 2 GB)");
    const size_t need = bd_conv2d_wgrad_fp8_workspace_bytes(d);
    if (ws_bytes < need) {
        bd_set_error("conv2d_wgrad_fp8: workspace %zu < required %zu bytes", ws_bytes, need);
        return BD_EWORKSPACE;
    }
    WF8Params p{};
    p.x = (const unsigned char*)x8; p.g = (const unsigned char*)g8; p.slab = (float*)ws; p.inv_scale = inv_scale;
    p.Cin = d->Cin; p.Cout = d->Cout; p.N = d->N; p.nseg = d->nseg;
    p.in_ppi = d->in_pix_per_img; p.out_ppi = d->out_pix_per_img;
    p.x_bytes = (unsigned)((long long)d->N * d->in_pix_per_img * d->Cin);
    p.g_bytes = (unsigned)((long long)d->N * d->out_pix_per_img * d->Cout);
    int ppi;
    const int total = f8_patches(d, &ppi);
    const int splits = f8_splits(d);
    p.total_patches = total; p.patches_per_img = ppi; p.patches_per_split = cdiv(total, splits);
    p.ci_tiles = cdiv(d->Cin, F8_TILE); p.co_tiles = cdiv(d->Cout, F8_TCO);
    int ps = 0;
    for (int s = 0; s < d->nseg; ++s) {
        F8Seg& sg = p.seg[s];
        sg.patch_start = ps; sg.H = d->Ho[s]; sg.W = d->Wo[s]; sg.pw = cdiv(d->Wo[s], F8_PW);
        sg.in_off = d->in_off[s]; sg.out_off = d->out_off[s];
        ps += cdiv(d->Ho[s], F8_PH) * sg.pw;
    }
    BD_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad3x3_fp8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * F8_BUF));
    const int grid = splits * p.ci_tiles * p.co_tiles;
    bd_note_kernel("conv_wgrad3x3_fp8_kernel");
    hipLaunchKernelGGL(conv_wgrad3x3_fp8_kernel, dim3(grid), dim3(256), 2 * F8_BUF, (hipStream_t)stream, p);
    BD_CHECK_LAUNCH("bd_conv2d_wgrad_fp8");
    bd_wgrad_reduce_launch((const float*)ws, splits, (long long)d->Cout * 9 * d->Cin, 9 * d->Cin, row_scale, dw, accumulate, (hipStream_t)stream);
    BD_CHECK_LAUNCH("bd_conv2d_wgrad_fp8(reduce)");
    return BD_OK;
}
