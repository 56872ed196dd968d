// Weight gradient of 3x3 / stride-1 / pad-1 convolutions, round-4 form: ONE persistent eight-wave workgroup per CU, operands
// streamed global -> LDS by LDS-DMA into a five-stage ring, 64 ci x 128 co x 9 taps of fp32 accumulators per workgroup.
//
//   dW[co][tap][ci] = sum_pix G[pix][co] * X[pix + shift(tap)][ci]
//
// conv_wgrad3x3.hip (round 1-3: 64 x 64 tile, four waves, two workgroups per CU, register-staged double buffer) ran at 0.42 of the
// MFMA roof with the matrix pipe busy half of the time: the next patch was requested ONE step (~1 150 MFMA cycles) ahead, less than a
// loaded chip's memory latency, and every step paid a vmcnt(0) + six ds_write_b128 per thread + a barrier.  Here
//   * a step's operands (10 x 10 input pixels x 64 ci, 8 x 8 output pixels x 128 co: 29 KB) are requested FOUR steps ahead with
//     `buffer_load ... lds` (no staging registers, no ds_write pass, out-of-image / channel-tail lanes read zeros through the
//     descriptor's range check), one counted vmcnt + one barrier per step;
//   * rows stay unpadded in LDS (a DMA instruction writes 1 KiB contiguously): the 32-byte pairs of a row are XOR-swizzled on the
//     SOURCE address so that the eight rows a half-wave transposes (ds_read_b64_tr_b16) fall on all 64 banks.  The swizzle key of the
//     input image is the pixel COLUMN only ((ix >> 1) & 3: any eight consecutive columns of the 10-wide image give four distinct keys
//     per row parity), so a tap's row shift stays an address immediate and only three lane addresses (tap column 0 / 1 / 2) are live;
//   * the workgroup walks its whole pixel range with one prologue; its partial sums leave as whole 1 KiB register rows
//     (slab[wg][wave][reg][lane], fully coalesced) and the fixed-order reduce (conv_wgrad.hip) un-permutes them;
//   * the bias gradient (column sums of G) is one extra MFMA per K sub-step against a fragment of ones, spread over the four waves that
//     share a co group: no pass over G, no VALU work.
#include <stdlib.h>
#include "common.h"
#include "wgrad_reduce.h"

namespace {

constexpr int RK_CI = 64, RK_CO = 128;           // workgroup tile
constexpr int XW = 10;                           // input image: 10 x 10 pixels around the 8 x 8 output patch
constexpr int X_PIECES = 13;                     // 13 x 8 = 104 rows of 128 B (rows 100..103: never read)
constexpr int X_BYTES = X_PIECES * 1024;
constexpr int G_PIECES = 16;                     // 64 rows of 256 B
constexpr int G_BYTES = G_PIECES * 1024;
constexpr int STAGE = X_BYTES + G_BYTES;         // 29 696 B
constexpr int DEPTH = 3;                         // steps requested ahead
constexpr int NSTAGE = 5;                        // (stage (t + 3) % 5 was last read in step t - 2: see the main loop)
constexpr int LDS_BYTES = NSTAGE * STAGE;        // 148 480 B: one workgroup per CU
constexpr int REGS = 36;                         // f32x4 accumulators per lane = 1 KiB rows per wave in the slab
constexpr unsigned X_NONE = 0x80000000u;

typedef __attribute__((address_space(3))) void lds_void_rk_t;

struct RSeg { int patch_start, H, W, pw, in_off, out_off, Hi, Wi; };

struct RParams {
    const bf16_raw* x;
    const bf16_raw* g;
    float* slab;             // [splits * tiles][8 waves][36][64 lanes] f32x4
    float* csum;             // optional [splits][Cout] partial column sums of g (bias gradient), written by the ci_tile 0 workgroups
    int Cin, Cout, N, nseg;
    int in_ppi, out_ppi;
    unsigned x_bytes, g_bytes;
    int patches_per_img, total_patches, patches_per_split;
    int ci_tiles, co_tiles;
    RSeg seg[BD_MAX_SEGS];
};

// 16 bytes per lane straight into LDS (1 KiB per wave, at the wave-uniform LDS byte address `lds_addr`), as INLINE ASSEMBLY: hipcc counts an
// LDS-DMA it knows about as a pending LDS write and puts `s_waitcnt vmcnt(0)` in front of the next ds_read_b64_tr_b16 builtin (it does not
// for a plain ds_read_b128: measured on a two-line kernel) -- that would drain the whole four-step ring at the top of every step.  Hidden
// from the compiler, the requests are retired by the loop's own counted vmcnt + barrier, and the transposing reads stay builtins whose
// lgkmcnt the compiler counts.  M0 is written in the statement that uses it (s_nop 0: the SALU-write-M0 -> LDS-DMA wait state).
#ifdef BD_RK_STAMP        // diagnostic build only (scripts/exp/rk_stamp.py): s_memtime cycles per loop phase, per wave of one workgroup
__device__ unsigned long long g_rk_stamp[8][8];
#endif

// -DBD_RK_ABLATE=<bits> (diagnostic builds, TIMING ONLY -- the results are wrong; scripts/exp/rk_power.sh): bit 0 = the second K half of a step
// re-uses the first half's fragments (half of the loop's transposing reads gone), bit 1 = no operand DMA inside the loop, bit 2 = only the X
// pieces 0 - 7 of 13 are requested (what a smaller input halo could save at most: -38 % of the X bytes).  Same MFMAs, same barriers.
#ifndef BD_RK_ABLATE
#define BD_RK_ABLATE 0
#endif
__device__ unsigned long long g_rk_clk[2];           // bd_probe_kernel_clock("conv_wgrad3x3_ring_kernel")

__device__ __forceinline__ void rk_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned lds_addr, unsigned voff, int soff = 0) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" : : "v"(voff), "s"(lds_addr), "s"(rsrc), "s"(soff) : "memory");
}

__global__ __launch_bounds__(512) void conv_wgrad3x3_ring_kernel(const RParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 3, wo = wave >> 2;          // 16-ci group, 64-co group
    if (blockIdx.x == 0 && threadIdx.x == 0) bd_clk_mark(g_rk_clk, false);
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap: the tiles of one split share its patches -> consecutive ids on ONE XCD (one HBM fetch per XCD)
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int wg = bid;                               // slab slot
    const int tiles = p.ci_tiles * p.co_tiles;
    const int split = bid / tiles;
    bid -= split * tiles;
    const int ci_tile = bid / p.co_tiles, co_tile = bid - ci_tile * p.co_tiles;
    const int ci0 = ci_tile * RK_CI, co0 = co_tile * RK_CO;
    const int pbeg = split * p.patches_per_split;
    int pend = pbeg + p.patches_per_split;
    if (pend > p.total_patches) pend = p.total_patches;
    const int nsteps = pend > pbeg ? pend - pbeg : 0;

    // ---- DMA lane constants.  X piece pc = LDS rows 8 pc .. 8 pc + 7 (row R = image pixel (R / 10, R % 10)), lane -> row lane >> 3,
    // 16-byte position lane & 7, source chunk = position ^ 2 * ((ix >> 1) & 3).  This wave owns pieces wave and wave + 8 (< 13).
    // G piece pc = rows 4 pc .. 4 pc + 3 (row R = patch pixel (R >> 3, R & 7)), lane -> row lane >> 4, position lane & 15, source chunk
    // pair = (position >> 1) ^ (R & 7).  This wave owns pieces wave and wave + 8.
    const bool two_x = wave + 8 < X_PIECES;
    int x_iy[2], x_ix[2], x_cb[2];
    bool x_ok[2], g_ok[2];                            // row exists and its channel chunk lies inside Cin / Cout
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int R = 8 * (wave + 8 * k) + (lane >> 3);
        const int iy = R / XW, ix = R - iy * XW;
        const int c = (lane & 7) ^ (2 * ((ix >> 1) & 3));
        x_ok[k] = R < XW * XW && ci0 + c * 8 < p.Cin;
        x_iy[k] = x_ok[k] ? iy : 0;
        x_ix[k] = ix;
        x_cb[k] = c * 16;
    }
    int g_ry[2], g_rx[2], g_cb[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int R = 4 * (wave + 8 * k) + (lane >> 4);
        const int pos = lane & 15;
        const int c = 2 * ((pos >> 1) ^ (R & 7)) + (pos & 1);
        g_ok[k] = co0 + c * 8 < p.Cout;
        g_ry[k] = R >> 3;
        g_rx[k] = R & 7;
        g_cb[k] = c * 16;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>((lds_void_rk_t*)smem));
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.g), 0, p.g_bytes, 0x00020000);

    // ---- patch cursor (workgroup-uniform): one decode of the first id, then increments with carries; it runs DEPTH steps ahead of the
    // MFMAs.  Per-level lane offsets are recomputed when the walk crosses into another pyramid level.
    int c_n = 0, c_s = 0, c_by = 0, c_bx = 0, c_rows = 1;
    RSeg sg = p.seg[0];
    int x_vec[2], g_vec[2];
    unsigned x_in[2], g_in[2];        // the same offsets with the static validity folded in (X_NONE): what an interior patch requests
    auto level_vectors = [&]() {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            x_vec[k] = ((x_iy[k] * sg.Wi + x_ix[k]) * p.Cin) * 2 + x_cb[k];
            g_vec[k] = ((g_ry[k] * sg.W + g_rx[k]) * p.Cout) * 2 + g_cb[k];
            x_in[k] = x_ok[k] ? (unsigned)x_vec[k] : X_NONE;
            g_in[k] = g_ok[k] ? (unsigned)g_vec[k] : X_NONE;
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
        c_rows = (sg.H + 7) >> 3;
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
        c_rows = (sg.H + 7) >> 3;
        level_vectors();
    };
    // request the cursor's patch into ring stage `stage`, X pieces and G pieces separately (dead: past the end of this workgroup's range --
    // the DMAs are still issued, with every lane out of range, so that each wave's vmcnt arithmetic is the same on every step)
    bool in_loop = false;
    auto issue_x = [&](int stage, bool dead) {
        if ((BD_RK_ABLATE & 2) && in_loop) return;
        const unsigned Xs = lds0 + stage * STAGE + wave * 1024;            // LDS byte address of this wave's first piece
        const int ys = c_by * 8 - 1, xs = c_bx * 8 - 1;
        const int xorg = ((c_n * p.in_ppi + sg.in_off + ys * sg.Wi + xs) * p.Cin + ci0) * 2;       // may be negative; valid sums are not
        // interior patch (a workgroup-uniform test; most patches): the lane offset is the per-level constant, the patch origin rides in the
        // scalar offset -- no per-lane arithmetic in the load segment
        if (!dead && ys >= 0 && xs >= 0 && ys + XW <= sg.Hi && xs + XW <= sg.Wi) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (k == 1 && (!two_x || (BD_RK_ABLATE & 4))) break;
                rk_dma16(x_rsrc, Xs + k * 8192, x_in[k], xorg);
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k == 1 && (!two_x || (BD_RK_ABLATE & 4))) break;
            const bool ok = !dead & x_ok[k] & ((unsigned)(ys + x_iy[k]) < (unsigned)sg.Hi) & ((unsigned)(xs + x_ix[k]) < (unsigned)sg.Wi);
            rk_dma16(x_rsrc, Xs + k * 8192, ok ? (unsigned)(xorg + x_vec[k]) : X_NONE);
        }
    };
    auto issue_g = [&](int stage, bool dead) {
        if ((BD_RK_ABLATE & 2) && in_loop) return;
        const unsigned Gs = lds0 + stage * STAGE + X_BYTES + wave * 1024;
        const int y0 = c_by * 8, x0 = c_bx * 8;
        const int gorg = ((c_n * p.out_ppi + sg.out_off + y0 * sg.W + x0) * p.Cout + co0) * 2;
        if (!dead && y0 + 8 <= sg.H && x0 + 8 <= sg.W) {
#pragma unroll
            for (int k = 0; k < 2; ++k) rk_dma16(g_rsrc, Gs + k * 8192, g_in[k], gorg);
            return;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool ok = !dead & g_ok[k] & (y0 + g_ry[k] < sg.H) & (x0 + g_rx[k] < sg.W);
            rk_dma16(g_rsrc, Gs + k * 8192, ok ? (unsigned)(gorg + g_vec[k]) : X_NONE);
        }
    };

    // ---- fragment addressing.  MFMA k -> pixel map (same for A and B): k = 8 g4 + j  <->  patch row 4 kk + 2 (g4 >> 1) + (j >> 2),
    // column 4 (g4 & 1) + (j & 3); a transposing read takes lane 4 q + p's address as row q, channels 4 p .. 4 p + 3 of its 16-lane group.
    const int g4 = lane >> 4, idx = lane & 15;
    const int tr_q = idx >> 2, tr_p = idx & 3;
    const int prow_hi = 2 * (g4 >> 1);
    const int pcol = 4 * (g4 & 1) + tr_q;
    int xa[3], ga[4];
#pragma unroll
    for (int s = 0; s < 3; ++s)
        xa[s] = (prow_hi * XW + pcol + s) * 128 + ((wi ^ (((pcol + s) >> 1) & 3)) * 32) + tr_p * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        ga[j] = X_BYTES + (prow_hi * 8 + pcol) * 256 + (((4 * wo + j) ^ pcol) * 32) + tr_p * 8;

    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    auto tr_frag = [&](const unsigned char* a0, int hi_bytes) -> bf16x8_t {
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0 + hi_bytes));
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, v);
    };

    f32x4_t acc[9][4];     // wave tile: 16 ci x 64 co per tap
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    f32x4_t cacc = {0.f, 0.f, 0.f, 0.f};               // ones x G: every row = the column sums of this wave's co fragment `wi`
    const bool do_cs = p.csum != nullptr && ci_tile == 0;
    bf16x8_t ones;
    {
        const s16x8_t o = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
        ones = __builtin_bit_cast(bf16x8_t, o);
    }

    // ---- main loop: two staggered wave groups (waves 0-3 / 4-7: one wave of each per SIMD), TWO phases per step (the K halves of the patch).
    // A phase = load segment (the 13 fragments of a K half: 26 transposing reads; two of the step's DMA requests), barrier, MFMA segment
    // (36 MFMAs + the bias row), barrier.  The second group runs one barrier behind, so one group's load segment -- LDS latency, the
    // ~100-cycle issue cost of a DMA piece, the counted vmcnt -- sits under the other group's MFMA segment on the same SIMDs.
    // (First form of this kernel: one barrier per step, every wave in the same phase: stamped 3 466 cycles per step against 2 368 of
    // MFMA work per SIMD -- 400 of DMA issue with the matrix pipe idle, the older wave of a SIMD done after 1 760 and parked for 1 190.)
    //
    // Ring discipline (NSTAGE = 5, DEPTH = 3): step t's K half 0 requests the X pieces of step t + 3, its K half 1 waits for THIS WAVE's
    // pieces of step t + 1 (everything younger stays in flight) and then requests the G pieces of step t + 3.  Stage (t + 3) % 5 was last
    // read in step t - 2.  A wave's wait in load segment (t, 1) is followed by two barriers (one for the group that runs behind) before any
    // wave reads step t + 1.
    // The "hi" rows of tap row r are the "lo" rows of tap row r + 1: a K half needs FOUR transposing reads per tap column (input rows
    // 4 kk .. 4 kk + 3 under this lane's two patch rows... i.e. image rows prow + 0 .. 3), not six.  They land in one 8-dword register
    // block per tap column; tap row r's fragment is dwords 2 r .. 2 r + 3 of it -- an even-aligned window, no copies.
    typedef __attribute__((ext_vector_type(8))) int i32x8_t;
    typedef __attribute__((ext_vector_type(4))) int i32x4_t;
    typedef __attribute__((ext_vector_type(2))) int i32x2_t;
    i32x8_t xr[3];
    bf16x8_t fb[4];
    auto tr64 = [&](const unsigned char* a0) -> i32x2_t {
        const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0));
        return __builtin_bit_cast(i32x2_t, v);
    };
    auto load_half = [&](int stage, int kk) {
        const unsigned char* base = smem + stage * STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = tr_frag(base + ga[j] + (4 * kk * 8) * 256, 8 * 256);
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) {
            const unsigned char* b0 = base + xa[sx] + (4 * kk * XW) * 128;
            const i32x2_t r0 = tr64(b0), r1 = tr64(b0 + XW * 128), r2 = tr64(b0 + 2 * XW * 128), r3 = tr64(b0 + 3 * XW * 128);
            xr[sx] = (i32x8_t){r0[0], r0[1], r1[0], r1[1], r2[0], r2[1], r3[0], r3[1]};
        }
    };
    auto mfma_taps = [&](int t0, int t1, bool with_bias) {
#pragma unroll
        for (int t = t0; t < t1; ++t) {
            const int r = t / 3, sx = t - 3 * (t / 3);
            const i32x4_t w = r == 0 ? __builtin_shufflevector(xr[sx], xr[sx], 0, 1, 2, 3)
                                     : (r == 1 ? __builtin_shufflevector(xr[sx], xr[sx], 2, 3, 4, 5) : __builtin_shufflevector(xr[sx], xr[sx], 4, 5, 6, 7));
            const bf16x8_t a = __builtin_bit_cast(bf16x8_t, w);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, fb[j], acc[t][j], 0, 0, 0);
        }
        if (with_bias && do_cs) {                     // wave-uniform branches: no register copies, one MFMA
            if (wi == 0) cacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fb[0], cacc, 0, 0, 0);
            else if (wi == 1) cacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fb[1], cacc, 0, 0, 0);
            else if (wi == 2) cacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fb[2], cacc, 0, 0, 0);
            else cacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fb[3], cacc, 0, 0, 0);
        }
    };
#define RK_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#define RK_FENCE() __builtin_amdgcn_sched_barrier(0)

#if defined(BD_PP_PRIO) && BD_PP_PRIO == 1
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    if (nsteps > 0) {
        seek(pbeg);
#pragma unroll 1
        for (int d = 0; d < DEPTH; ++d) {          // prologue: DEPTH whole steps in flight
            issue_x(d, d >= nsteps);
            issue_g(d, d >= nsteps);
            advance();
        }
        if (two_x) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // step 0 has landed (2 steps x 4 pieces stay in flight)
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        RK_BARRIER();
        if (wo == 1) RK_BARRIER();                 // stagger: the second group runs one barrier behind
        RK_FENCE();
        in_loop = true;
        int stage = 0, fill = DEPTH;               // stage of step t; stage the requests of step t + DEPTH go to
#ifdef BD_RK_STAMP
        unsigned long long st[4] = {0, 0, 0, 0};
        const unsigned long long st_begin = __builtin_amdgcn_s_memtime(), st_rbegin = __builtin_amdgcn_s_memrealtime();
        unsigned long long last__ = st_begin;
#define RK_T(i) do { const unsigned long long now__ = __builtin_amdgcn_s_memtime(); st[i] += now__ - last__; last__ = now__; } while (0)
#else
#define RK_T(i) do { } while (0)
#endif
#pragma unroll 1
        for (int t = 0; t < nsteps; ++t) {
            const bool dead = t + DEPTH >= nsteps;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                // ---------------- load segment ----------------
                if (!((BD_RK_ABLATE & 1) && kk == 1)) load_half(stage, kk);
                RK_FENCE();
                // (Round 6, measured and removed: the phase's requests issued behind tap 3 / tap 5 of the MFMA segment instead of here: the step
                // 3.0 % slower, profiles/r06_dma_pos.txt -- see conv3x3_pp.hip)
                if (kk == 0) {
                    issue_x(fill, dead);
                } else {
                    // this wave's pieces of step t + 1 have landed; step t + 2 and the X pieces of step t + 3 stay in flight
                    if (BD_RK_ABLATE & 6) { }                // (fewer / no requests in flight: the counts below would wait for younger ones)
                    else if (two_x) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    issue_g(fill, dead);
                    advance();
                }
                RK_FENCE();
                RK_T(0);
                RK_BARRIER();
                RK_T(1);
                // ---------------- MFMA segment ----------------
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                RK_FENCE();
                // (no s_setprio around the MFMA segment since round 5.  Rounds 2-4 raised the priority here and dropped it behind the
                // segment; alternating on one box, three repetitions: those flips 648.0 / 645.1 / 641.9 img/s, one static priority for
                // waves 4-7 649.1 / 647.7 / 647.0, none 649.5 / 649.2 / 647.6 -- profiles/r05_prio.txt; -DBD_PP_PRIO=0 brings the flips back)
#if defined(BD_PP_PRIO) && BD_PP_PRIO == 0
                __builtin_amdgcn_s_setprio(1);
#endif
                mfma_taps(0, 9, true);
#if defined(BD_PP_PRIO) && BD_PP_PRIO == 0
                __builtin_amdgcn_s_setprio(0);
#endif
                RK_FENCE();
                RK_T(2);
                RK_BARRIER();
                RK_FENCE();
                RK_T(3);
            }
            stage = stage + 1 == NSTAGE ? 0 : stage + 1;
            fill = fill + 1 == NSTAGE ? 0 : fill + 1;
        }
        if (wo == 0) RK_BARRIER();                 // un-stagger: every wave has passed the same number of barriers
#ifdef BD_RK_STAMP
        if (blockIdx.x == 100 % gridDim.x && lane == 0) {
            for (int k = 0; k < 4; ++k) g_rk_stamp[wave][k] = st[k];
            g_rk_stamp[wave][4] = (unsigned long long)nsteps;
            g_rk_stamp[wave][5] = __builtin_amdgcn_s_memtime() - st_begin;
            g_rk_stamp[wave][6] = __builtin_amdgcn_s_memrealtime() - st_rbegin;       // 100 MHz
        }
#endif
    }

    // ---- partial sums: whole register rows, 1 KiB per wave and instruction ----
    float* out = p.slab + ((size_t)(wg * 8 + wave) * REGS) * 256 + lane * 4;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4_t*>(out + (t * 4 + j) * 256) = acc[t][j];
    if (do_cs && g4 == 0) {
        const int co = co0 + wo * 64 + wi * 16 + idx;
        if (co < p.Cout) p.csum[(size_t)split * p.Cout + co] = cacc[0];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) bd_clk_mark(g_rk_clk, true);
}

struct RPlan { int ci_tiles, co_tiles, tiles, splits, total, ppi, per; };

RPlan ring_plan(const bd_conv_desc* d) {
    RPlan pl;
    pl.ppi = 0;
    for (int s = 0; s < d->nseg; ++s) pl.ppi += cdiv(d->Ho[s], 8) * cdiv(d->Wo[s], 8);
    pl.total = pl.ppi * d->N;
    pl.ci_tiles = cdiv(d->Cin, RK_CI);
    pl.co_tiles = cdiv(d->Cout, RK_CO);
    pl.tiles = pl.ci_tiles * pl.co_tiles;
    static const int target_env = bd_tune_env("BD_WGRAD3R_TARGET", 0);      // workgroups per launch (measurement knob)
    const int target = target_env > 0 ? target_env : bd_num_cus();
    int splits = target / pl.tiles;
    if (splits < 1) splits = 1;
    const int max_splits = pl.total / 8 > 0 ? pl.total / 8 : 1;        // at least 8 patches per workgroup: the prologue alone is 4 steps
    if (splits > max_splits) splits = max_splits;
    pl.per = cdiv(pl.total, splits);
    pl.splits = cdiv(pl.total, pl.per);
    return pl;
}

}  // namespace

int bd_rk_clk_read(unsigned long long* out2, int reset) {          // (probe.hip)
    if (hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_rk_clk), 16) != hipSuccess) return 1;
    if (reset) { const unsigned long long z[2] = {0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_rk_clk), z, 16) != hipSuccess) return 1; }
    return 0;
}

#ifdef BD_RK_STAMP
extern "C" int bd_debug_rk_stamp(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_rk_stamp), sizeof(g_rk_stamp)) == hipSuccess ? 0 : 1;
}
#endif

// 3x3 / stride 1 / pad 1, Cin >= 64, Cout >= 96 (narrower outputs waste most of the 128-channel tile: conv_wgrad3x3.hip keeps them)
bool bd_wgrad3x3r_eligible(const bd_conv_desc* d) {
    if (!(d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1)) return false;
    if (d->Cin < 64 || d->Cout < 96 || d->Cin % 8 || d->Cout % 8) return false;
    if ((long long)d->N * d->in_pix_per_img * d->Cin * 2 >= 0x7fffffffll || (long long)d->N * d->out_pix_per_img * d->Cout * 2 >= 0x7fffffffll)
        return false;
    for (int s = 0; s < d->nseg; ++s)
        if (d->Hi[s] != d->Ho[s] || d->Wi[s] != d->Wo[s]) return false;
    return true;
}

size_t bd_wgrad3x3r_slab_bytes(const bd_conv_desc* d, int* splits_out) {
    const RPlan pl = ring_plan(d);
    if (splits_out) *splits_out = pl.splits;
    return (size_t)pl.splits * pl.tiles * 8 * REGS * 1024;
}

// writes the slabs (and, with csum, [splits][Cout] partial column sums of g); the caller reduces with bd_wgrad3x3r_reduce
int bd_wgrad3x3r_launch(const bd_conv_desc* d, const void* x, const void* g, float* slab, float* csum, int* splits_out, hipStream_t stream) {
    RParams p{};
    const RPlan pl = ring_plan(d);
    p.x = (const bf16_raw*)x; p.g = (const bf16_raw*)g; p.slab = slab; p.csum = csum;
    p.Cin = d->Cin; p.Cout = d->Cout; p.N = d->N; p.nseg = d->nseg;
    p.in_ppi = d->in_pix_per_img; p.out_ppi = d->out_pix_per_img;
    p.x_bytes = (unsigned)((long long)d->N * d->in_pix_per_img * d->Cin * 2);
    p.g_bytes = (unsigned)((long long)d->N * d->out_pix_per_img * d->Cout * 2);
    p.total_patches = pl.total; p.patches_per_img = pl.ppi; p.patches_per_split = pl.per;
    p.ci_tiles = pl.ci_tiles; p.co_tiles = pl.co_tiles;
    int ps = 0;
    for (int s = 0; s < d->nseg; ++s) {
        RSeg& sg = p.seg[s];
        sg.patch_start = ps; sg.H = d->Ho[s]; sg.W = d->Wo[s]; sg.pw = cdiv(d->Wo[s], 8);
        sg.in_off = d->in_off[s]; sg.out_off = d->out_off[s]; sg.Hi = d->Hi[s]; sg.Wi = d->Wi[s];
        ps += cdiv(d->Ho[s], 8) * sg.pw;
    }
    BD_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad3x3_ring_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipLaunchKernelGGL(conv_wgrad3x3_ring_kernel, dim3(pl.splits * pl.tiles), dim3(512), LDS_BYTES, stream, p);
    *splits_out = pl.splits;
    return 0;
}

// the layout half of this layer's reduce descriptor (conv_wgrad.hip fills in pointers and launches / queues it)
void bd_wgrad3x3r_entry(const bd_conv_desc* d, BdRedEntry* e) {
    const RPlan pl = ring_plan(d);
    e->kind = 1; e->regs = REGS; e->co_tiles = pl.co_tiles; e->tci = RK_CI; e->tco = RK_CO; e->fi = 0; e->fj = 0;
    e->Cin = d->Cin; e->Cout = d->Cout; e->n4 = 8 * REGS * 64 * pl.tiles; e->row_len = 0;
}
