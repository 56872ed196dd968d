// Weight gradient of 1x1 convolutions (stride 1 or 2): dW[co][ci] = sum_pix G[pix][co] * X[src(pix)][ci].
//
// A pure [Cout x Cin] = G^T X GEMM whose reduction (pixels, 16 800 .. 1 075 200) dwarfs its output, so the cost is
// re-reading the operands: with T_ci x T_co output tiles X is read Cout/T_co times and G Cin/T_ci times.  The generic
// kernel (128 x 128 tiles, 64 FLOP per staged byte) is L2/HBM-bound at ~180 TFLOP/s; this kernel gives each workgroup
// 65 536 outputs -- 256 ci x 256 co, or 128 ci x 512 co when Cin <= 128 -- i.e. 128 FLOP per staged byte and half the
// operand traffic.  8 waves, each 128 ci x 64 co (128 fp32 accumulators per lane); operands staged [pixel][channel]
// and transposed on the way to the MFMA by ds_read_b64_tr_b16 (conv_wgrad.hip); split over pixels into fp32 slabs.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int MAX_SEG = BD_MAX_SEGS;

struct WSeg1 { int m_start, Ho, Wo, per_img; float inv_per_img, inv_wo; int Hi, Wi, in_off, out_off; };

struct W1Params {
    const bf16_raw* x;
    const bf16_raw* g;
    float* slab;
    int Cin, Cout, stride, M, nseg;
    int linear;      // 1: single dense level, stride 1 -> source/destination pixel index == GEMM row (no decode)
    unsigned x_bytes, g_bytes;   // tensor sizes when both are < 2 GB (buffer-load staging), else 0
    int in_ppi, out_ppi;
    int ci_tiles, co_tiles, steps_per_split, total_steps;
    WSeg1 seg[MAX_SEG];
};

__device__ __forceinline__ void fast_divmod1(int n, int d, float inv, int& q, int& r) {
    q = (int)((float)n * inv);
    r = n - q * d;
    if (r < 0) { --q; r += d; }
    else if (r >= d) { ++q; r -= d; }
}

// BUF: operands staged with range-checked buffer loads (32-bit per-thread byte offsets, an offset past the buffer reads zeros).  On a
// dense stride-1 level the per-thread offset is a CONSTANT and the K step is a scalar offset: no per-load address arithmetic and no
// predication branches (the pointer path spent 50-125 VALU instructions and ~20 branches per 32 MFMAs).
template <int TCI, int TCO, int BKP, bool BUF>
__global__ __launch_bounds__(512, 2) void conv_wgrad1x1_kernel(const W1Params p) {
    constexpr int XP = TCI * 2 + 32, GP = TCO * 2 + 32;        // row pitches (== 8 dwords mod 64)
    constexpr int X_BYTES = BKP * XP, G_BYTES = BKP * GP;
    constexpr int XCH = TCI / 8, GCH = TCO / 8;                // 16-byte chunks per row
    constexpr int XPASS = BKP * XCH / 512, GPASS = BKP * GCH / 512;
    constexpr int WCO = TCO / 64;                              // waves along co
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wci = wave / WCO, wco = wave - wci * WCO;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap (as conv_wgrad3x3.hip): the tiles of one pixel split share its activation and gradient rows, so
        // they get consecutive ids on ONE XCD and the rows come from HBM once per split instead of once per tile
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tiles = p.ci_tiles * p.co_tiles;
    const int split = bid / tiles;
    bid -= split * tiles;
    const int ci_tile = bid / p.co_tiles, co_tile = bid - ci_tile * p.co_tiles;
    const int ci0 = ci_tile * TCI, co0 = co_tile * TCO;
    const int step_begin = split * p.steps_per_split;
    int step_end = step_begin + p.steps_per_split;
    if (step_end > p.total_steps) step_end = p.total_steps;

    u32x4_t rx[2][XPASS], rg[2][GPASS];      // two register stages: loads run two K steps ahead of the MFMAs

    auto decode = [&](int m, long long& xoff, long long& goff) {
        xoff = -1; goff = -1;
        if (m >= p.M) return;
        if (p.linear) { xoff = (long long)m * p.Cin; goff = (long long)m * p.Cout; return; }
        int s = 0;
#pragma unroll
        for (int k = 1; k < MAX_SEG; ++k)
            if (k < p.nseg && m >= p.seg[k].m_start) s = k;
        const WSeg1 sg = p.seg[s];
        int n, rem, oy, ox;
        fast_divmod1(m - sg.m_start, sg.per_img, sg.inv_per_img, n, rem);
        fast_divmod1(rem, sg.Wo, sg.inv_wo, oy, ox);
        goff = ((long long)n * p.out_ppi + sg.out_off + oy * sg.Wo + ox) * p.Cout;
        xoff = ((long long)n * p.in_ppi + sg.in_off + (long long)(oy * p.stride) * sg.Wi + ox * p.stride) * p.Cin;
    };
    // Strided / offset single-level layers (the stride-2 shortcut convolutions): the two float divisions of decode() per staged row
    // cost more VALU cycles than the step has MFMA cycles (4 rows x ~50 instructions x 4 cycles against 32 MFMAs x 16).  The steps
    // of a workgroup are consecutive, so each staged row keeps its (image, row, column) and advances it by BKP pixels per call.
    // (only in the 256 x 256 instance, which serves them: the 128 x 512 one has no registers to spare)
    const bool inc = TCI == 256 && TCO == 256 && !p.linear && p.nseg == 1;
    const WSeg1 sg0 = p.seg[0];
    const int n_img = p.M / (sg0.per_img > 0 ? sg0.per_img : 1);
    int ix_n[XPASS], ix_y[XPASS], ix_x[XPASS], ig_n[GPASS], ig_y[GPASS], ig_x[GPASS];
    if (inc) {
#pragma unroll
        for (int k = 0; k < XPASS; ++k) {
            int rem;
            fast_divmod1(step_begin * BKP + (tid + 512 * k) / XCH, sg0.per_img, sg0.inv_per_img, ix_n[k], rem);
            fast_divmod1(rem, sg0.Wo, sg0.inv_wo, ix_y[k], ix_x[k]);
        }
#pragma unroll
        for (int k = 0; k < GPASS; ++k) {
            int rem;
            fast_divmod1(step_begin * BKP + (tid + 512 * k) / GCH, sg0.per_img, sg0.inv_per_img, ig_n[k], rem);
            fast_divmod1(rem, sg0.Wo, sg0.inv_wo, ig_y[k], ig_x[k]);
        }
    }
    auto advance = [&](int& n, int& y, int& x) {
        x += BKP;
        while (x >= sg0.Wo) { x -= sg0.Wo; ++y; }
        while (y >= sg0.Ho) { y -= sg0.Ho; ++n; }
    };
    constexpr unsigned X_NONE = 0x80000000u;
    __amdgpu_buffer_rsrc_t x_rsrc, g_rsrc;
    unsigned xv[BUF ? XPASS : 1], gv[BUF ? GPASS : 1];      // dense level: constant byte offset of this thread's chunk in K step 0
    if (BUF) {
        x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.x), 0, p.x_bytes, 0x00020000);
        g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.g), 0, p.g_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < XPASS; ++k) {
            const int c = tid + 512 * k, row = c / XCH, ch = c - row * XCH;
            xv[k] = ci0 + ch * 8 < p.Cin ? (unsigned)(row * p.Cin + ci0 + ch * 8) * 2u : X_NONE;
        }
#pragma unroll
        for (int k = 0; k < GPASS; ++k) {
            const int c = tid + 512 * k, row = c / GCH, ch = c - row * GCH;
            gv[k] = co0 + ch * 8 < p.Cout ? (unsigned)(row * p.Cout + co0 + ch * 8) * 2u : X_NONE;
        }
    }
    auto stage_load = [&](int step, u32x4_t (&rx)[XPASS], u32x4_t (&rg)[GPASS]) {
        // x: chunk id c = tid + 512*k -> row c / XCH, chunk c % XCH ; g likewise
        if (BUF && p.linear) {
            int sx = step * BKP * p.Cin * 2, sg = step * BKP * p.Cout * 2;
            asm volatile("" : "+s"(sx), "+s"(sg));                 // the K step travels in the scalar offset
#pragma unroll
            for (int k = 0; k < XPASS; ++k)
                rx[k] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, step * BKP + (tid + 512 * k) / XCH < p.M ? xv[k] : X_NONE, sx, 0);
#pragma unroll
            for (int k = 0; k < GPASS; ++k)
                rg[k] = __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, step * BKP + (tid + 512 * k) / GCH < p.M ? gv[k] : X_NONE, sg, 0);
            return;
        }
#pragma unroll
        for (int k = 0; k < XPASS; ++k) {
            const int c = tid + 512 * k;
            const int row = c / XCH, ch = c - row * XCH;
            long long xo, go;
            if (inc) {
                xo = ix_n[k] < n_img ? (long long)(ix_n[k] * p.in_ppi + sg0.in_off + ix_y[k] * p.stride * sg0.Wi + ix_x[k] * p.stride) * p.Cin : -1;
                advance(ix_n[k], ix_y[k], ix_x[k]);
            } else {
                decode(step * BKP + row, xo, go);
            }
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (BUF) v = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (xo >= 0 && ci0 + ch * 8 < p.Cin) ? (unsigned)(xo + ci0 + ch * 8) * 2u : X_NONE, 0, 0);
            else if (xo >= 0 && ci0 + ch * 8 < p.Cin) v = *reinterpret_cast<const u32x4_t*>(p.x + xo + ci0 + ch * 8);
            rx[k] = v;
        }
#pragma unroll
        for (int k = 0; k < GPASS; ++k) {
            const int c = tid + 512 * k;
            const int row = c / GCH, ch = c - row * GCH;
            long long xo, go;
            if (inc) {
                go = ig_n[k] < n_img ? (long long)(ig_n[k] * p.out_ppi + sg0.out_off + ig_y[k] * sg0.Wo + ig_x[k]) * p.Cout : -1;
                advance(ig_n[k], ig_y[k], ig_x[k]);
            } else {
                decode(step * BKP + row, xo, go);
            }
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (BUF) v = __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, (go >= 0 && co0 + ch * 8 < p.Cout) ? (unsigned)(go + co0 + ch * 8) * 2u : X_NONE, 0, 0);
            else if (go >= 0 && co0 + ch * 8 < p.Cout) v = *reinterpret_cast<const u32x4_t*>(p.g + go + co0 + ch * 8);
            rg[k] = v;
        }
    };
    auto stage_write = [&](int buf, const u32x4_t (&rx)[XPASS], const u32x4_t (&rg)[GPASS]) {
        unsigned char* Xt = smem + buf * (X_BYTES + G_BYTES);
        unsigned char* Gt = Xt + X_BYTES;
#pragma unroll
        for (int k = 0; k < XPASS; ++k) {
            const int c = tid + 512 * k;
            const int row = c / XCH, ch = c - row * XCH;
            *reinterpret_cast<u32x4_t*>(Xt + row * XP + ch * 16) = rx[k];
        }
#pragma unroll
        for (int k = 0; k < GPASS; ++k) {
            const int c = tid + 512 * k;
            const int row = c / GCH, ch = c - row * GCH;
            *reinterpret_cast<u32x4_t*>(Gt + row * GP + ch * 16) = rg[k];
        }
    };

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // k -> pixel permutation as in conv_wgrad.hip: k = 8*g4 + j <-> pixel 16*(g4>>1) + 4*(g4&1) + (j&3) + 8*(j>>2)
    const int g4 = lane >> 4, idx = lane & 15;
    const int prow_base = 16 * (g4 >> 1) + 4 * (g4 & 1);
    const int tr_q = idx >> 2, tr_p = idx & 3;
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    auto tr_frag = [&](const unsigned char* a0, int pitch) -> bf16x8_t {
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0 + 8 * pitch));
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, v);
    };
    auto compute = [&](int buf) {
        const unsigned char* Xt = smem + buf * (X_BYTES + G_BYTES) + (prow_base + tr_q) * XP + (wci * 128 + 4 * tr_p) * 2;
        const unsigned char* Gt = smem + buf * (X_BYTES + G_BYTES) + X_BYTES + (prow_base + tr_q) * GP + (wco * 64 + 4 * tr_p) * 2;
#pragma unroll
        for (int kk = 0; kk < BKP / 32; ++kk) {
            bf16x8_t b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = tr_frag(Gt + kk * 32 * GP + j * 32, GP);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bf16x8_t a = tr_frag(Xt + kk * 32 * XP + i * 32, XP);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[j], acc[i][j], 0, 0, 0);
            }
        }
    };

    // stage s is issued at step s-2 into rX[s&1], written to LDS buffer s&1 at the end of step s-1, consumed at step s
    if (step_begin < step_end) stage_load(step_begin, rx[0], rg[0]);
    if (step_begin + 1 < step_end) stage_load(step_begin + 1, rx[1], rg[1]);
    if (step_begin < step_end) stage_write(0, rx[0], rg[0]);
    __syncthreads();
    for (int st = step_begin; st < step_end; st += 2) {
        // even relative step: buffer 0
        if (st + 2 < step_end) stage_load(st + 2, rx[0], rg[0]);
        compute(0);
        if (st + 1 < step_end) stage_write(1, rx[1], rg[1]);
        __syncthreads();
        if (st + 1 < step_end) {
            if (st + 3 < step_end) stage_load(st + 3, rx[1], rg[1]);
            compute(1);
            if (st + 2 < step_end) stage_write(0, rx[0], rg[0]);
            __syncthreads();
        }
    }

    float* slab = p.slab + (long long)split * p.Cout * p.Cin;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co = co0 + wco * 64 + j * 16 + idx;
        if (co >= p.Cout) continue;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ci = ci0 + wci * 128 + i * 16 + g4 * 4;
            if (ci >= p.Cin) continue;
            *reinterpret_cast<f32x4_t*>(slab + (long long)co * p.Cin + ci) = acc[i][j];
        }
    }
}

template <int TCI, int TCO, int BKP, bool BUF>
void launch_w1(const W1Params& p, int grid, hipStream_t stream) {
    constexpr size_t lds = 2 * (size_t)BKP * ((TCI * 2 + 32) + (TCO * 2 + 32));
    static_assert(lds <= 160 * 1024, "LDS budget");
    BD_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad1x1_kernel<TCI, TCO, BKP, BUF>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((conv_wgrad1x1_kernel<TCI, TCO, BKP, BUF>), dim3(grid), dim3(512), lds, stream, p);
}

int g_w1_buf = 1;         // 0 (BD_W1_PTR=1 in the environment): 64-bit pointer staging, for A/B runs
int g_w1_allow_512 = 0;   // the 512 x 128 tile measured 2.7x slower than its 128 x 512 mirror image (kept for study)
// (tile_ci, tile_co, pixels per K step): 128 x 512 needs the shorter step to fit LDS (2 x 32 x 1344 B = 84 KB)
void tile_shape(const bd_conv_desc* d, int& tci, int& tco, int& bkp) {
    if (d->Cin <= 128) { tci = 128; tco = 512; bkp = 32; }
    else if (d->Cout <= 128 && d->Cin >= 512 && g_w1_allow_512) { tci = 512; tco = 128; bkp = 32; }
    else { tci = 256; tco = 256; bkp = 32; }
}

}  // namespace

int bd_wgrad1x1_splits(const bd_conv_desc* d) {
    int tci, tco, bkp;
    tile_shape(d, tci, tco, bkp);
    long long M = 0;
    for (int s = 0; s < d->nseg; ++s) M += (long long)d->N * d->Ho[s] * d->Wo[s];
    const int total_steps = (int)cdiv64(M, bkp);
    const int tiles = cdiv(d->Cin, tci) * cdiv(d->Cout, tco);
    int splits = 256 / tiles;
    if (splits < 1) splits = 1;
    const int max_splits = total_steps / 4 > 0 ? total_steps / 4 : 1;
    if (splits > max_splits) splits = max_splits;
    const int per = cdiv(total_steps, splits);
    return cdiv(total_steps, per);
}

int bd_wgrad1x1_launch(const bd_conv_desc* d, const void* x, const void* g, float* slab, int* splits_out, hipStream_t stream) {
    int tci, tco, bkp;
    tile_shape(d, tci, tco, bkp);
    W1Params p{};
    p.x = (const bf16_raw*)x; p.g = (const bf16_raw*)g; p.slab = slab;
    p.Cin = d->Cin; p.Cout = d->Cout; p.stride = d->stride; p.nseg = d->nseg;
    p.in_ppi = d->in_pix_per_img; p.out_ppi = d->out_pix_per_img;
    long long m = 0;
    for (int s = 0; s < d->nseg; ++s) {
        WSeg1& sg = p.seg[s];
        sg.m_start = (int)m; sg.Ho = d->Ho[s]; sg.Wo = d->Wo[s]; sg.per_img = d->Ho[s] * d->Wo[s];
        sg.inv_per_img = 1.0f / (float)sg.per_img; sg.inv_wo = 1.0f / (float)sg.Wo;
        sg.Hi = d->Hi[s]; sg.Wi = d->Wi[s]; sg.in_off = d->in_off[s]; sg.out_off = d->out_off[s];
        m += (long long)d->N * sg.per_img;
    }
    p.M = (int)m;
    p.linear = (d->nseg == 1 && d->stride == 1 && d->in_off[0] == 0 && d->out_off[0] == 0 &&
                d->in_pix_per_img == d->Ho[0] * d->Wo[0] && d->out_pix_per_img == d->Ho[0] * d->Wo[0]) ? 1 : 0;
    p.total_steps = (int)cdiv64(m, bkp);
    const int splits = bd_wgrad1x1_splits(d);
    p.steps_per_split = cdiv(p.total_steps, splits);
    p.ci_tiles = cdiv(d->Cin, tci); p.co_tiles = cdiv(d->Cout, tco);
    const int grid = splits * p.ci_tiles * p.co_tiles;
    const long long xb = (long long)d->N * d->in_pix_per_img * d->Cin * 2, gb = (long long)d->N * d->out_pix_per_img * d->Cout * 2;
    static const bool env_ptr = bd_tune_env_str("BD_W1_PTR") != nullptr;
    const bool buf = g_w1_buf && !env_ptr && xb < 0x7fffffffll && gb < 0x7fffffffll;
    p.x_bytes = buf ? (unsigned)xb : 0u; p.g_bytes = buf ? (unsigned)gb : 0u;
    if (tci == 128) { if (buf) launch_w1<128, 512, 32, true>(p, grid, stream); else launch_w1<128, 512, 32, false>(p, grid, stream); }
    else if (tci == 512) launch_w1<512, 128, 32, false>(p, grid, stream);
    else { if (buf) launch_w1<256, 256, 32, true>(p, grid, stream); else launch_w1<256, 256, 32, false>(p, grid, stream); }
    *splits_out = splits;
    return 0;
}
