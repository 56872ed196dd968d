// Weight gradient of 3x3 / stride-1 / pad-1 convolutions (RetinaNet head towers, FPN output convs, bottleneck
// conv2): all nine taps from ONE staged activation patch.
//
//   dW[co][tap][ci] = sum_pix G[pix][co] * X[pix + shift(tap)][ci]
//
// The generic wgrad kernel (conv_wgrad.hip) re-reads both operands once per tap and per opposite tile
// (64 FLOP per staged byte -> L2-bandwidth bound at ~250 TFLOP/s).  Here a workgroup stages, per K step, an
// 8x8 output patch of G (64 pixels x 64 co) and the 10x10 input patch of X around it (100 pixels x 64 ci) and
// accumulates the nine 64x64 tap products from them: ~225 FLOP per staged byte.  Each wave owns 16 ci x 64 co
// of all nine taps (144 fp32 accumulator registers), two workgroups per CU; the next patch is prefetched into
// registers under the 72 MFMAs of the current one.
//
// Both MFMA operands are pixel-major in LDS and are fetched with ds_read_b64_tr_b16 (see conv_wgrad.hip).  A tap
// only shifts the LDS row of the X fragment by (r*10 + s) rows -- a compile-time address immediate.
// MFMA k -> pixel map (same for A and B): k = 8*g4 + j  <->  patch row 4*kk + 2*(g4>>1) + (j>>2),
// column 4*(g4&1) + (j&3); a half-wave therefore reads 8 consecutive LDS rows = all 64 banks (conflict-free).
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int PATCH = 8;                 // output patch: 8 x 8 pixels (stride 1), 4 rows x 8 pixels (stride 2)
constexpr int TILE_CI = 64, TILE_CO = 64;
constexpr int X_PITCH = 160;             // 128 B data + 32 B pad  (40 dwords: 8 consecutive rows hit all 64 banks)
constexpr int G_PITCH = 160;             // 128 B data + 32 B pad

// Input-patch image in LDS.  Stride 1: one (PATCH+2)^2 image, tap (r, s) = row offset r*XW + s.  Stride 2 (the first 3x3 of
// layer2/3/4, P6/P7): the input patch is stored PHASE-MAJOR -- four images of the (row parity, column parity) classes -- so that
// the eight pixels a half-wave transposes are again eight CONSECUTIVE LDS rows; tap (r, s) selects the phase image (r&1, s&1) and
// the row offset (r>>1)*XW + (s>>1).  The stride-2 patch is 4 output rows x 8 (one 32-pixel K step, 9 x 17 input pixels): the 8 x 8
// patch's 17 x 17 input image would be 62 KB double-buffered twice = one four-wave workgroup per CU and nothing to overlap its
// staging with (that variant ran at 300-390 TFLOP/s); 2 x 34 KB keeps two workgroups per CU.
template <int STRIDE> struct XImg {
    static constexpr int PH = STRIDE == 1 ? PATCH : PATCH / 2, PW = PATCH;  // output patch rows / columns
    static constexpr int KSUB = PH / 4;                                    // 32-pixel MFMA K sub-steps per patch
    static constexpr int XW = STRIDE == 1 ? PW + 2 : PW + 1;               // rows per image line
    static constexpr int XH = STRIDE == 1 ? PH + 2 : PH + 1;               // lines per (phase) image
    static constexpr int IMG = XW * XH;                                    // rows per (phase) image
    static constexpr int ROWS = STRIDE == 1 ? IMG : 4 * IMG;
    static constexpr int X_BYTES = ROWS * X_PITCH;
    static constexpr int G_BYTES = PH * PW * G_PITCH;
    static constexpr int G_PASSES = PH * PW * 8 / 256;                     // 2 / 1
    static constexpr int BUF_BYTES = X_BYTES + G_BYTES;
    static constexpr int X_CHUNKS = ROWS * 8;
    static constexpr int X_PASSES = (X_CHUNKS + 255) / 256;
    static constexpr int EXT_H = STRIDE == 1 ? PH + 2 : 2 * PH + 1;        // input patch extent in pixels
    static constexpr int EXT_W = STRIDE == 1 ? PW + 2 : 2 * PW + 1;
    __host__ __device__ static constexpr int tap_row(int r, int s) {
        return STRIDE == 1 ? r * XW + s : ((r & 1) * 2 + (s & 1)) * IMG + (r >> 1) * XW + (s >> 1);
    }
};

struct PSeg { int patch_start, H, W, pw, in_off, out_off, Hi, Wi; };

#ifdef BD_W3_STAMP        // diagnostic build only (scripts/exp/w3_stamp.py): s_memtime cycles per loop phase, per wave of one workgroup
__device__ unsigned long long g_w3_stamp[4][8];
#endif

struct W3Params {
    const bf16_raw* x;
    const bf16_raw* g;
    float* slab;
    float* csum;             // optional [splits][Cout] partial column sums of g (bias gradient), written by the ci_tile 0 workgroups
    int Cin, Cout, N, nseg;
    int in_ppi, out_ppi;
    unsigned x_bytes, g_bytes;
    int patches_per_img, total_patches, patches_per_split;
    int ci_tiles, co_tiles;
    PSeg seg[BD_MAX_SEGS];
};

template <int STRIDE>
__global__ __launch_bounds__(256, 2) void conv_wgrad3x3_kernel(const W3Params p) {
    using XI = XImg<STRIDE>;
    constexpr int X_PASSES = XI::X_PASSES, X_CHUNKS = XI::X_CHUNKS, X_BYTES = XI::X_BYTES, BUF_BYTES = XI::BUF_BYTES, XW = XI::XW;
    constexpr int G_PASSES = XI::G_PASSES, PH = XI::PH, PW = XI::PW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap: the workgroups of one split (all ci x co tiles: they share the split's activation and gradient
        // patches) get consecutive ids on ONE XCD, so each patch is fetched from HBM once per XCD instead of once per tile
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tiles = p.ci_tiles * p.co_tiles;
    const int split = bid / tiles;
    bid -= split * tiles;
    const int ci_tile = bid / p.co_tiles, co_tile = bid - ci_tile * p.co_tiles;
    const int ci0 = ci_tile * TILE_CI, co0 = co_tile * TILE_CO;
    const int pbeg = split * p.patches_per_split;
    int pend = pbeg + p.patches_per_split;
    if (pend > p.total_patches) pend = p.total_patches;

    // per-thread staging slots
    int x_iy[X_PASSES], x_ix[X_PASSES];      // position inside the input patch (pixels), -1 = unused slot
    const int x_chunk = tid & 7;
    const bool x_cok = ci0 + x_chunk * 8 < p.Cin;
#pragma unroll
    for (int k = 0; k < X_PASSES; ++k) {
        const int c = tid + 256 * k;
        const int row = c >> 3;
        if (STRIDE == 1) {
            x_iy[k] = row < XI::ROWS ? row / XW : -1;
            x_ix[k] = row - (row / XW) * XW;
        } else {
            const int ph = row / XI::IMG, rr = row - ph * XI::IMG;
            const int a = rr / XW, b = rr - a * XW;
            const int u = 2 * a + (ph >> 1), v = 2 * b + (ph & 1);
            x_iy[k] = (row < XI::ROWS && u < XI::EXT_H && v < XI::EXT_W) ? u : -1;
            x_ix[k] = v;
        }
    }
    const int g_chunk = tid & 7;
    const bool g_cok = co0 + g_chunk * 8 < p.Cout;

    u32x4_t rx[X_PASSES], rg[G_PASSES];
    const bool do_cs = p.csum != nullptr && ci_tile == 0;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // column sums of this thread's 8 channels over the rows it stages

    // Range-checked buffer loads: a 32-bit per-thread byte offset, and X_NONE (past the end of the tensor: the host checks both
    // tensors are < 2 GB) for halo / out-of-image / channel-tail positions, which then read as zeros -- no predicated loads, no
    // 64-bit per-thread addresses, no zero-initialised staging registers.  The per-thread part of the offset depends on the pyramid
    // level (row pitch) only and is recomputed when the patch sequence crosses into another level.
    constexpr unsigned X_NONE = 0x80000000u;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.g), 0, p.g_bytes, 0x00020000);
    // byte offsets relative to the patch origin, for the current level, X_NONE for unused slots / channel tails
    int x_vec[X_PASSES], g_vec[G_PASSES];

    // Patch cursor (workgroup-uniform, lives in scalar registers): a split walks consecutive patch ids, so after the one decode of its
    // first id (two integer divisions, a level search, a descriptor fetch) the next patch is an increment with carries -- the decode per
    // patch was a serial chain of ~150 scalar instructions and a kernel-argument load in front of every MFMA block (stamped: 1 500
    // cycles of a 3 000-cycle patch step).
    int c_n = 0, c_s = 0, c_by = 0, c_bx = 0, c_rows = 1;
    PSeg sg = p.seg[0];
    auto level_vectors = [&]() {
#pragma unroll
        for (int k = 0; k < X_PASSES; ++k)
            x_vec[k] = (x_iy[k] >= 0 && x_cok) ? ((x_iy[k] * sg.Wi + x_ix[k]) * p.Cin + x_chunk * 8) * 2 : (int)X_NONE;
#pragma unroll
        for (int k = 0; k < G_PASSES; ++k) {
            const int row = (tid >> 3) + 32 * k;
            g_vec[k] = g_cok ? (((row >> 3) * sg.W + (row & 7)) * p.Cout + g_chunk * 8) * 2 : (int)X_NONE;
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
        c_rows = (sg.H + PH - 1) / PH;
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
        c_rows = (sg.H + PH - 1) / PH;
        level_vectors();
    };

    // Interior patches (every pixel of both patches inside the image: a workgroup-uniform test) take the short form: the per-thread
    // byte offset is a per-level constant (X_NONE already folded in for unused slots / channel tails) and the patch origin rides in the
    // buffer instruction's scalar offset -- one instruction per load.  A wave issues at most one instruction every four cycles, and the
    // checked form below is ~130 of them (stamped: 1 000 cycles in front of a 1 152-cycle MFMA block).
    auto stage_load = [&]() {
        const int n = c_n, by = c_by, bx = c_bx;
        const int y0 = by * PH, x0 = bx * PW;
        const int ys = STRIDE * y0 - 1, xs = STRIDE * x0 - 1;
        const int xorg = (n * p.in_ppi + sg.in_off + ys * sg.Wi + xs) * p.Cin + ci0;        // may be negative; valid sums are not
        const int gorg = (n * p.out_ppi + sg.out_off + y0 * sg.W + x0) * p.Cout + co0;
        if (ys >= 0 && xs >= 0 && ys + XI::EXT_H <= sg.Hi && xs + XI::EXT_W <= sg.Wi && y0 + PH <= sg.H && x0 + PW <= sg.W) {
#pragma unroll
            for (int k = 0; k < X_PASSES; ++k)
                rx[k] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (unsigned)x_vec[k], xorg * 2, 0);
#pragma unroll
            for (int k = 0; k < G_PASSES; ++k) rg[k] = __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, (unsigned)g_vec[k], gorg * 2, 0);
            return;
        }
#pragma unroll
        for (int k = 0; k < X_PASSES; ++k) {
            const int y = ys + x_iy[k], x = xs + x_ix[k];
            const bool ok = x_iy[k] >= 0 && x_cok && y >= 0 && x >= 0 && y < sg.Hi && x < sg.Wi;
            rx[k] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, ok ? (unsigned)xorg * 2u + (unsigned)x_vec[k] : X_NONE, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < G_PASSES; ++k) {
            const int row = (tid >> 3) + 32 * k;
            const bool ok = g_cok && y0 + (row >> 3) < sg.H && x0 + (row & 7) < sg.W;
            rg[k] = __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, ok ? (unsigned)gorg * 2u + (unsigned)g_vec[k] : X_NONE, 0, 0);
        }
    };
    auto stage_write = [&](int buf) {
        unsigned char* Xt = smem + buf * BUF_BYTES;
        unsigned char* Gt = Xt + X_BYTES;
#pragma unroll
        for (int k = 0; k < X_PASSES; ++k) {
            const int c = tid + 256 * k;
            if (256 * (k + 1) <= X_CHUNKS || c < X_CHUNKS) *reinterpret_cast<u32x4_t*>(Xt + (c >> 3) * X_PITCH + x_chunk * 16) = rx[k];
        }
#pragma unroll
        for (int k = 0; k < G_PASSES; ++k) {
            const int row = (tid >> 3) + 32 * k;
            *reinterpret_cast<u32x4_t*>(Gt + row * G_PITCH + g_chunk * 16) = rg[k];
        }
        if (do_cs) {            // bias gradient: every output pixel passes through exactly one patch (zeros where the patch overhangs)
#pragma unroll
            for (int k = 0; k < G_PASSES; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) { cs[2 * j] += bf_lo(rg[k][j]); cs[2 * j + 1] += bf_hi(rg[k][j]); }
        }
    };

    f32x4_t acc[9][4];     // wave tile: 16 ci x 64 co per tap
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int g4 = lane >> 4, idx = lane & 15;
    const int tr_q = idx >> 2, tr_p = idx & 3;
    // lane-constant parts of the fragment addresses
    const int prow_hi = 2 * (g4 >> 1);            // patch row inside the 4-row sub-step
    const int pcol = 4 * (g4 & 1) + tr_q;         // patch column supplied by this lane
    const int x_lane_off = (prow_hi * XW + pcol) * X_PITCH + (wave * 16 + 4 * tr_p) * 2;
    const int g_lane_off = (prow_hi * PW + pcol) * G_PITCH + (4 * tr_p) * 2;

    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    auto tr_frag = [&](const unsigned char* a0, int row_pitch_bytes) -> bf16x8_t {
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0 + row_pitch_bytes));
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, v);
    };

    // Fragment reads run one tap ahead of the MFMAs that consume them (two A register sets), and the next sub-step's four G fragments
    // are fetched under taps 1..4 (two B sets): only the first reads after a barrier are exposed.  The order is pinned with
    // sched_barrier -- left alone the compiler hoists 16 reads to the top of each sub-step and waits for all of them (lgkmcnt(0))
    // before the first MFMA.
    auto compute = [&](int buf) {
        const unsigned char* Xt = smem + buf * BUF_BYTES + x_lane_off;
        const unsigned char* Gt = smem + buf * BUF_BYTES + X_BYTES + g_lane_off;
        constexpr int NT = 9 * XI::KSUB;
        bf16x8_t a[2], b[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[0][j] = tr_frag(Gt + j * 32, PW * G_PITCH);
        a[0] = tr_frag(Xt + XI::tap_row(0, 0) * X_PITCH, XW * X_PITCH);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);      // two workgroups per CU: the wave in its MFMA block wins the issue arbitration (+2 %)
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int kk = i / 9, t = i % 9;
            if (i + 1 < NT) {
                const int k2 = (i + 1) / 9, t2 = (i + 1) % 9;
                a[(i + 1) & 1] = tr_frag(Xt + (4 * k2 * XW + XI::tap_row(t2 / 3, t2 % 3)) * X_PITCH, XW * X_PITCH);
            }
            if (kk + 1 < XI::KSUB && t >= 1 && t <= 4)
                b[(kk + 1) & 1][t - 1] = tr_frag(Gt + (4 * (kk + 1) * PW) * G_PITCH + (t - 1) * 32, PW * G_PITCH);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 1], b[kk & 1][j], acc[t][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
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
#ifdef BD_W3_STAMP
    unsigned long long st[5] = {0, 0, 0, 0, 0};
    const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
    const unsigned long long st_rbegin = __builtin_amdgcn_s_memrealtime();
#define W3_T() __builtin_amdgcn_s_memtime()
    for (int pid = pbeg; pid < pend; ++pid) {
        const bool more = pid + 1 < pend;
        const unsigned long long a0 = W3_T();
        if (more) { stage_load(); advance(); }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long a1 = W3_T();
        compute(cur);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long a2 = W3_T();
        if (more) stage_write(cur ^ 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long a3 = W3_T();
        __syncthreads();
        const unsigned long long a4 = W3_T();
        st[0] += a1 - a0; st[1] += a2 - a1; st[2] += a3 - a2; st[3] += a4 - a3; st[4] += 1;
        cur ^= 1;
    }
    if (blockIdx.x == 300 % gridDim.x && lane == 0) {
        for (int k = 0; k < 5; ++k) g_w3_stamp[wave][k] = st[k];
        g_w3_stamp[wave][5] = __builtin_amdgcn_s_memtime() - st_begin;
        g_w3_stamp[wave][6] = __builtin_amdgcn_s_memrealtime() - st_rbegin;       // 100 MHz
    }
#else
    for (int pid = pbeg; pid < pend; ++pid) {
        const bool more = pid + 1 < pend;
        if (more) { stage_load(); advance(); }
        compute(cur);
        if (more) stage_write(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
#endif

    if (do_cs) {            // 32 staging threads share a channel chunk: fixed-order sum through LDS (the loop's last barrier has passed)
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int j = 0; j < 8; ++j) red[tid * 8 + j] = cs[j];
        __syncthreads();
        if (tid < TILE_CO && co0 + tid < p.Cout) {
            float s = 0.f;
            for (int r = 0; r < 32; ++r) s += red[(r * 8 + (tid >> 3)) * 8 + (tid & 7)];
            p.csum[(long long)split * p.Cout + co0 + tid] = s;
        }
    }
    float* slab = p.slab + (long long)split * p.Cout * 9 * p.Cin;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co = co0 + j * 16 + idx;
        if (co >= p.Cout) continue;
        const int ci = ci0 + wave * 16 + g4 * 4;
        if (ci >= p.Cin) continue;
#pragma unroll
        for (int t = 0; t < 9; ++t)
            *reinterpret_cast<f32x4_t*>(slab + ((long long)co * 9 + t) * p.Cin + ci) = acc[t][j];
    }
}

}  // namespace

// plan shared with conv_wgrad.hip
int bd_wgrad3x3_splits(const bd_conv_desc* d, int* total_patches_out, int* patches_per_img_out) {
    int ppi = 0;
    const int ph = d->stride == 1 ? XImg<1>::PH : XImg<2>::PH;
    for (int s = 0; s < d->nseg; ++s) ppi += cdiv(d->Ho[s], ph) * cdiv(d->Wo[s], PATCH);
    const int total = ppi * d->N;
    const int tiles = cdiv(d->Cin, TILE_CI) * cdiv(d->Cout, TILE_CO);
    static const int target = bd_tune_env("BD_WGRAD3_TARGET", 512);   // workgroups per launch (measurement knob)
    int splits = target / tiles;
    if (splits < 1) splits = 1;
    if (splits > total) splits = total;
    const int per = cdiv(total, splits);
    splits = cdiv(total, per);
    if (total_patches_out) *total_patches_out = total;
    if (patches_per_img_out) *patches_per_img_out = ppi;
    return splits;
}

#ifdef BD_W3_STAMP
extern "C" int bd_debug_w3_stamp(unsigned long long* out32) {
    return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_w3_stamp), sizeof(g_w3_stamp)) == hipSuccess ? 0 : 1;
}
#endif

int bd_wgrad3x3_launch(const bd_conv_desc* d, const void* x, const void* g, float* slab, float* csum, int* splits_out, hipStream_t stream) {
    W3Params p{};
    p.x = (const bf16_raw*)x; p.g = (const bf16_raw*)g; p.slab = slab; p.csum = csum;
    p.Cin = d->Cin; p.Cout = d->Cout; p.N = d->N; p.nseg = d->nseg;
    p.in_ppi = d->in_pix_per_img; p.out_ppi = d->out_pix_per_img;
    p.x_bytes = (unsigned)((long long)d->N * d->in_pix_per_img * d->Cin * 2);
    p.g_bytes = (unsigned)((long long)d->N * d->out_pix_per_img * d->Cout * 2);
    int total, ppi;
    const int splits = bd_wgrad3x3_splits(d, &total, &ppi);
    p.total_patches = total; p.patches_per_img = ppi; p.patches_per_split = cdiv(total, splits);
    p.ci_tiles = cdiv(d->Cin, TILE_CI); p.co_tiles = cdiv(d->Cout, TILE_CO);
    int ps = 0;
    const int ph = d->stride == 1 ? XImg<1>::PH : XImg<2>::PH;
    for (int s = 0; s < d->nseg; ++s) {
        PSeg& sg = p.seg[s];
        sg.patch_start = ps; sg.H = d->Ho[s]; sg.W = d->Wo[s]; sg.pw = cdiv(d->Wo[s], PATCH);
        sg.in_off = d->in_off[s]; sg.out_off = d->out_off[s]; sg.Hi = d->Hi[s]; sg.Wi = d->Wi[s];
        ps += cdiv(d->Ho[s], ph) * sg.pw;
    }
    BD_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad3x3_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  2 * XImg<1>::BUF_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad3x3_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  2 * XImg<2>::BUF_BYTES));
    const int grid = splits * p.ci_tiles * p.co_tiles;
    if (d->stride == 1) hipLaunchKernelGGL(conv_wgrad3x3_kernel<1>, dim3(grid), dim3(256), 2 * XImg<1>::BUF_BYTES, stream, p);
    else hipLaunchKernelGGL(conv_wgrad3x3_kernel<2>, dim3(grid), dim3(256), 2 * XImg<2>::BUF_BYTES, stream, p);
    *splits_out = splits;
    return 0;
}
