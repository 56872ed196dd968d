// 3x3 / stride-1 / pad-1 convolution forward and data-gradient: "patch" implicit GEMM.
//
// The generic kernel (conv_igemm.hip) stages the activation tile once per filter tap; with 128x128 tiles that is
// 64 FLOP per staged byte and the LDS store/load pipe, not the MFMA pipe, bounds it (~600 TFLOP/s).  Here a
// workgroup owns 256 output pixels = four 4x16 patches (one per pixel-quarter wave pair) and, per 64-channel K
// block, stages their four 6x18 input patches ONCE; the nine taps then read shifted rows of that image while only the 128x64 weight tile of each tap is
// re-staged:  ~225 FLOP per staged byte, LDS traffic per MFMA down ~1.7x, MFMA-bound.
//
// 512 threads = 8 waves = 2 (channel halves of 64) x 4 (pixel quarters of 64 = 4 patch rows of 16 px).
// LDS (unpadded, XOR-swizzled 128-byte rows: chunk position = chunk ^ (row & 7), conflict-free for ds_read_b128 of
// 16 consecutive rows): one 54 KB activation image + 2 x 48 KB weight buffers (three taps = one filter row per
// barrier step: 96 MFMAs per wave between barriers).
// dgrad = the same kernel on dY with the mirrored tap and the [Cin][tap][Cout] packed weights.
#include "common.h"

namespace {

constexpr int PH = 4, PW = 16;                 // output patch (4 rows of 16 px = the four pixel tiles of one wave)
constexpr int IH = PH + 2, IW = PW + 2;        // input patch
constexpr int NPATCH = 4;                      // patches per workgroup
constexpr int XROWS = NPATCH * IH * IW;        // 432 LDS rows
constexpr int X_PITCH = 144;                   // activation rows: 128 B of channels + 16 B pad -- 16 consecutive rows hit 16 distinct
                                               // bank quads (conflict-free ds_read_b128 / ds_write_b128) and a tap shift becomes a
                                               // compile-time byte offset: no swizzle arithmetic inside the MFMA loop
constexpr int X_BYTES = XROWS * X_PITCH;       // 62208
constexpr int TAPS_PER_STEP = 3;               // one filter row per barrier step
constexpr int W_TAP_BYTES = 128 * 128;         // 16384: one tap's 128 x 64 weight tile
constexpr int W_BYTES = TAPS_PER_STEP * W_TAP_BYTES;   // 49152
constexpr int WPASSES = TAPS_PER_STEP * 2;     // 16-byte chunks per thread and step
constexpr int XCHUNKS = XROWS * 8;             // 3456 16-byte chunks
constexpr int XPASSES = (XCHUNKS + 511) / 512; // 7
constexpr int TILE_CO = 128;
constexpr int MAX_SEG = BD_MAX_SEGS;

struct CSeg { int patch_start, H, W, pw, src_off, dst_off; };

struct C3Params {
    const bf16_raw* src;
    const bf16_raw* w;       // [CO][9][CK]
    const float* bias;
    const bf16_raw* add;
    const bf16_raw* mask;
    bf16_raw* dst;
    long long* dbg;          // optional timeline stamps of workgroup 0 (bd_conv3x3_set_debug)
    int CK, CO, mode, flags, N, nseg;
    int src_ppi, dst_ppi;
    int patches_per_img, total_patches, n_tiles;
    CSeg seg[MAX_SEG];
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// Bank layout of the activation image (rows of X_PITCH = 144 B, no row swizzle so that a tap shift is a compile-time offset).
// ds_read_b128 serves lanes {0-3, 12-15, 20-27} in one LDS cycle (MI355X_MICROARCH.md, LDS): pixel rows f = 0-3, 12-15 at K chunk c
// together with rows 4-11 at chunk c + 1.  With an odd pitch (in 16-B slots) those two row sets always collide on 7 of 8 slots,
// whatever the order.  Fix: MFMA column f computes patch column colperm(f) (rows 0-3, 12-15 -> even columns, 4-11 -> odd ones, so
// the two sets fall on even / odd slots) and chunks c, c + 1 are stored two slots apart (order 0 2 1 3): conflict-free.
__device__ __forceinline__ int colperm(int f) { return f < 4 ? 2 * f : (f < 12 ? 2 * (f - 4) + 1 : 2 * (f - 8)); }
__device__ __forceinline__ int xpos(int chunk) { return (chunk >> 2) * 64 + ((((chunk & 1) << 1) | ((chunk >> 1) & 1)) << 4); }

constexpr bool g_fence = true;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

// DMA = true: the weight tiles go global -> LDS with global_load_lds_dwordx4 (no staging VGPRs, no ds_write pass); a
// wave-instruction writes 1 KiB = 8 swizzled rows, the swizzle is applied on the per-lane SOURCE address.  Needs
// CK % 64 == 0 (no zero-filled K tail) -- other shapes use the register-staged variant.
template <bool DMA, int MODE>
__global__ __launch_bounds__(512, 2) void conv3x3_patch_kernel(const C3Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xbuf = smem;                       // [X_BYTES]      (single buffer, swapped between K blocks)
    unsigned char* wbuf = smem + X_BYTES;             // [2][W_BYTES]
    float* sbias = reinterpret_cast<float*>(smem + X_BYTES + 2 * W_BYTES);   // [128] epilogue vector of this channel tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave >> 2, wp = wave & 3;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int pt = bid / p.n_tiles;              // pixel tile (pair of patches)
    const int ct = bid - pt * p.n_tiles;
    const int co0 = ct * TILE_CO;

    // ---- geometry of the two patches (workgroup-uniform) ----
    int pn[NPATCH], py0[NPATCH], px0[NPATCH], pH[NPATCH], pWd[NPATCH];
    long long psrc[NPATCH], pdst[NPATCH];
#pragma unroll
    for (int k = 0; k < NPATCH; ++k) {
        const int pid = pt * NPATCH + k;
        pH[k] = 0; pWd[k] = 0; pn[k] = 0; py0[k] = 0; px0[k] = 0; psrc[k] = 0; pdst[k] = 0;
        if (pid < p.total_patches) {
            const int n = pid / p.patches_per_img;
            const int rem = pid - n * p.patches_per_img;
            int s = 0;
#pragma unroll
            for (int q = 1; q < MAX_SEG; ++q)
                if (q < p.nseg && rem >= p.seg[q].patch_start) s = q;
            const CSeg sg = p.seg[s];
            const int local = rem - sg.patch_start;
            const int by = local / sg.pw, bx = local - by * sg.pw;
            pn[k] = n; py0[k] = by * PH; px0[k] = bx * PW; pH[k] = sg.H; pWd[k] = sg.W;
            psrc[k] = (long long)n * p.src_ppi + sg.src_off;
            pdst[k] = (long long)n * p.dst_ppi + sg.dst_off;
        }
    }

    // ---- per-thread activation staging slots: chunk id c = tid + 512*k -> (row, chunk) ----
    long long x_off[XPASSES];   // element offset of the source pixel (without channel), -1 = zero-fill
    int x_lds[XPASSES];
    const int x_chunk = tid & 7;
#pragma unroll
    for (int k = 0; k < XPASSES; ++k) {
        const int c = tid + 512 * k;
        const int row = c >> 3;
        x_off[k] = -1; x_lds[k] = -1;
        if (row < XROWS) {
            const int pk = row / (IH * IW);
            const int rr = row - pk * (IH * IW);
            const int iy = rr / IW, ix = rr - iy * IW;
            int qy = py0[0], qx = px0[0], H = pH[0], W = pWd[0];
            long long qs = psrc[0];
#pragma unroll
            for (int q = 1; q < NPATCH; ++q)
                if (pk == q) { qy = py0[q]; qx = px0[q]; H = pH[q]; W = pWd[q]; qs = psrc[q]; }
            const int y = qy - 1 + iy, x = qx - 1 + ix;
            x_lds[k] = row * X_PITCH + xpos(x_chunk);
            if (y >= 0 && x >= 0 && y < H && x < W) x_off[k] = (qs + (long long)y * W + x) * p.CK;
        }
    }
    // weight staging: 128 rows x 8 chunks = 1024 chunks -> 2 per thread; LDS row lrow holds the permuted channel
    int w_co[2], w_lds[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = tid + 512 * k;
        const int lrow = c >> 3;
        const int rho = lrow & 15;
        w_co[k] = co0 + (lrow & 64) + 32 * ((lrow >> 5) & 1) + 8 * (rho >> 2) + 4 * ((lrow >> 4) & 1) + (rho & 3);
        w_lds[k] = swz(lrow, tid & 7);
    }

    // DMA pieces of this wave: piece pc = wave + 8k (k = 0..5) -> tap pc / 16, rows 8 (pc % 16) .. +7; lane -> (row, position)
    int dma_src[WPASSES];
    if (DMA) {
#pragma unroll
        for (int k = 0; k < WPASSES; ++k) {
            const int pc = wave + 8 * k;
            const int lrow = 8 * (pc & 15) + (lane >> 3);
            const int chunk = (lane & 7) ^ (lane >> 3);               // (lrow & 7) == lane >> 3
            const int rho = lrow & 15;
            int co = co0 + (lrow & 64) + 32 * ((lrow >> 5) & 1) + 8 * (rho >> 2) + 4 * ((lrow >> 4) & 1) + (rho & 3);
            if (co >= p.CO) co = p.CO - 1;                             // rows past CO are never stored: any finite data will do
            dma_src[k] = (co * 9 + (pc >> 4)) * p.CK + chunk * 8;
        }
    }

    const int kblocks = (p.CK + 63) / 64;
    const int nsteps = kblocks * 3;
    u32x4_t rw[DMA ? 1 : WPASSES], rx[XPASSES];
    auto dma_w = [&](int step, int buf, int k) {
        const int cb = step / 3, r = step - cb * 3;
        const int pc = wave + 8 * k;
        const bf16_raw* g = p.w + dma_src[k] + r * 3 * p.CK + cb * 64;
        unsigned char* l = wbuf + buf * W_BYTES + (pc >> 4) * W_TAP_BYTES + (pc & 15) * 1024;
        __builtin_amdgcn_global_load_lds((glb_void_t*)g, (lds_void_t*)l, 16, 0, 0);
    };

    auto load_w = [&](int step) {
        if (DMA) return;
        const int cb = step / 3, r = step - cb * 3;
        const int c0 = cb * 64 + (tid & 7) * 8;
        const bool cvalid = c0 + 8 <= p.CK;
#pragma unroll
        for (int t = 0; t < TAPS_PER_STEP; ++t)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                u32x4_t v = {0u, 0u, 0u, 0u};
                if (cvalid && w_co[k] < p.CO)
                    v = *reinterpret_cast<const u32x4_t*>(p.w + ((long long)w_co[k] * 9 + r * 3 + t) * p.CK + c0);
                rw[t * 2 + k] = v;
            }
    };
    auto write_w_tap = [&](int buf, int t) {
        if (DMA) return;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            *reinterpret_cast<u32x4_t*>(wbuf + buf * W_BYTES + t * W_TAP_BYTES + w_lds[k]) = rw[t * 2 + k];
    };
    auto load_x = [&](int cb) {
        const int c0 = cb * 64 + x_chunk * 8;
        const bool cvalid = c0 + 8 <= p.CK;
#pragma unroll
        for (int k = 0; k < XPASSES; ++k) {
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (cvalid && x_off[k] >= 0) v = *reinterpret_cast<const u32x4_t*>(p.src + x_off[k] + c0);
            rx[k] = v;
        }
    };
    auto write_x = [&]() {
#pragma unroll
        for (int k = 0; k < XPASSES; ++k)
            if (x_lds[k] >= 0) *reinterpret_cast<u32x4_t*>(xbuf + x_lds[k]) = rx[k];
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // fragment addressing
    const int frow = lane & 15, fchunk = lane >> 4;
    int a_off[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) a_off[i][kk] = swz(wc * 64 + i * 16 + frow, kk * 4 + fchunk);
    // B rows: patch wp, output row j, column frow; input row = (j + dy)*IW + frow + dx: lane base + compile-time offsets
    const unsigned char* b_base = xbuf + (wp * (IH * IW) + colperm(frow)) * X_PITCH + xpos(fchunk);

    // one barrier step = one filter row = 6 sub-steps (3 taps x 2 K halves) of 16 MFMAs.  Fragments are double
    // buffered: the ds_reads of sub-step u+1 are issued before the MFMAs of sub-step u, so LDS latency hides under the
    // MFMA pipe; after each tap the wave stores one tap of the NEXT step's weights (in registers since the end of the
    // previous step), so the LDS stores overlap the MFMAs as well.
    bf16x8_t fa[3][4], fb[3][4];      // DMA variant: three sets, prefetch distance 2 (the compiler then waits with lgkmcnt(8))
    auto load_frags = [&](int wb, int r, int u, bf16x8_t (&a)[4], bf16x8_t (&b)[4]) {
        const int t = u >> 1, kk = u & 1;
        int dy = r, dx = t;
        if (MODE == 1) { dy = 2 - dy; dx = 2 - dx; }       // dgrad: mirrored tap
        const unsigned char* Wt = wbuf + wb * W_BYTES + t * W_TAP_BYTES;
        const unsigned char* Bt = b_base + dy * (IW * X_PITCH);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(Wt + a_off[i][kk]);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(Bt + (j * IW + dx) * X_PITCH + kk * 64);
    };
    auto mfma16 = [&](const bf16x8_t (&a)[4], const bf16x8_t (&b)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    auto mfma_rows = [&](const bf16x8_t (&a)[4], const bf16x8_t (&b)[4], int i0, int i1) {
#pragma unroll
        for (int i = i0; i < i1; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    auto compute = [&](int wb, int r, bool write_next, int next_step) {
        if (DMA && write_next) {                                    // all six pieces at the start of the step: they land under the
#pragma unroll
            for (int k = 0; k < WPASSES; ++k) dma_w(next_step, wb ^ 1, k);      // MFMAs, long before the barrier's vmcnt(0)
            if (g_fence) __builtin_amdgcn_sched_barrier(0);
        }
        load_frags(wb, r, 0, fa[0], fb[0]);
        if (DMA && g_fence) load_frags(wb, r, 1, fa[1], fb[1]);
#pragma unroll
        for (int u = 0; u < 2 * TAPS_PER_STEP; ++u) {
            if (DMA && g_fence) {
                // fragments two sub-steps ahead: the ds_reads of sub-step u+2 are issued behind the first MFMAs of sub-step u, and the
                // wait in front of sub-step u+1 only covers reads that are a whole sub-step old
                __builtin_amdgcn_sched_barrier(0);
                mfma_rows(fa[u % 3], fb[u % 3], 0, 1);
                __builtin_amdgcn_sched_barrier(0);
                if (u + 2 < 2 * TAPS_PER_STEP) load_frags(wb, r, u + 2, fa[(u + 2) % 3], fb[(u + 2) % 3]);
                __builtin_amdgcn_sched_barrier(0);
                mfma_rows(fa[u % 3], fb[u % 3], 1, 4);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                if (u + 1 < 2 * TAPS_PER_STEP) load_frags(wb, r, u + 1, fa[(u + 1) & 1], fb[(u + 1) & 1]);
                mfma16(fa[u & 1], fb[u & 1]);
            }
            if ((u & 1) && write_next) write_w_tap(wb ^ 1, u >> 1);
        }
    };

#ifdef BD_PATCH_TIMELINE      // build with -DBD_PATCH_TIMELINE for scripts/patch_timeline.py (s_memtime stamps of one workgroup)
    long long stamps[64];
    int nst = 0;
    const bool dbg_on = p.dbg != nullptr && blockIdx.x == 8 && lane == 0;
#define STAMP() do { if (dbg_on && nst < 64) stamps[nst++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP() do { } while (0)
#endif
    STAMP();
    // prologue: activation image of K block 0, weights of step 0 (to LDS) and step 1 (in registers); the epilogue vector goes to
    // LDS now, so that its global-load latency is not exposed after the last MFMA
    if (tid < TILE_CO) sbias[tid] = (p.bias && co0 + tid < p.CO) ? p.bias[co0 + tid] : 0.f;
    load_x(0);
    if (DMA) {
#pragma unroll
        for (int k = 0; k < WPASSES; ++k) dma_w(0, 0, k);
    }
    load_w(0);
    write_x();
#pragma unroll
    for (int t = 0; t < TAPS_PER_STEP; ++t) write_w_tap(0, t);
    if (nsteps > 1) load_w(1);
    __syncthreads();
    STAMP();
    for (int step = 0; step < nsteps; ++step) {
        const int cb = step / 3, r = step - cb * 3;
        const bool next_x = (r == 2) && (cb + 1 < kblocks);
        if (next_x) load_x(cb + 1);          // in flight under this step's 96 MFMAs
        compute(step & 1, r, step + 1 < nsteps, step + 1);
        if (step + 2 < nsteps) load_w(step + 2);   // registers are free again: they were stored during this step
        STAMP();
        __syncthreads();                     // (DMA: the compiler drains vmcnt before the barrier, so the pieces have landed)
        STAMP();
        if (next_x) {                        // every wave is done with the old image: swap it
            write_x();
            __syncthreads();
            STAMP();
        }
    }

    // ---- epilogue (same permuted-channel scheme as conv_igemm.hip) ----
    const int cg = lane >> 4;
    const bool do_relu = p.flags & BD_EPI_RELU;
    const bool add_before = (p.flags & BD_EPI_ADD_BEFORE) && p.add;
    const bool add_after = (p.flags & BD_EPI_ADD_AFTER) && p.add;
    const bool do_mask = (p.flags & BD_EPI_MASK) && p.mask;
    const int cbase = co0 + wc * 64 + 8 * cg;      // + 32 * half below
    float bias[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) bias[k] = 0.f;
    {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(sbias + wc * 64 + 8 * cg + 32 * (q >> 1) + 4 * (q & 1));
            bias[4 * q] = bv[0]; bias[4 * q + 1] = bv[1]; bias[4 * q + 2] = bv[2]; bias[4 * q + 3] = bv[3];
        }
    }
    int oy0 = py0[0], ox = px0[0] + colperm(frow), H = pH[0], W = pWd[0];
    long long dbase = pdst[0];
#pragma unroll
    for (int q = 1; q < NPATCH; ++q)
        if (wp == q) { oy0 = py0[q]; ox = px0[q] + colperm(frow); H = pH[q]; W = pWd[q]; dbase = pdst[q]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int oy = oy0 + j;
        if (oy >= H || ox >= W) continue;
        const long long base = (dbase + (long long)oy * W + ox) * p.CO + cbase;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (cbase + 32 * half >= p.CO) continue;
            const long long idx = base + 32 * half;
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = acc[2 * half + (k >> 2)][j][k & 3] + bias[8 * half + k];
            if (add_before) {
                const u32x4_t av = *reinterpret_cast<const u32x4_t*>(p.add + idx);
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[2 * k] += bf_lo(av[k]); v[2 * k + 1] += bf_hi(av[k]); }
            }
            if (do_relu) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
            }
            if (do_mask) {
                const u32x4_t mv = *reinterpret_cast<const u32x4_t*>(p.mask + idx);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (!(bf_lo(mv[k]) > 0.f)) v[2 * k] = 0.f;
                    if (!(bf_hi(mv[k]) > 0.f)) v[2 * k + 1] = 0.f;
                }
            }
            if (add_after) {
                const u32x4_t av = *reinterpret_cast<const u32x4_t*>(p.add + idx);
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[2 * k] += bf_lo(av[k]); v[2 * k + 1] += bf_hi(av[k]); }
            }
            u32x4_t o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = pack_bf2(v[2 * k], v[2 * k + 1]);
            *reinterpret_cast<u32x4_t*>(p.dst + idx) = o;
        }
    }
    STAMP();
#ifdef BD_PATCH_TIMELINE
    if (dbg_on) {
        p.dbg[wave * 80] = nst;
        for (int i = 0; i < nst; ++i) p.dbg[wave * 80 + 1 + i] = stamps[i];
    }
#endif
#undef STAMP
}

}  // namespace

BD_KNOB int g_patch_pp = 2;       // 2 = wherever the shape allows (default), 1 = only where the makespan estimate favours it (bit 7), 0 = never (bit 6): staggered 256-channel-tile instance (conv3x3_pp.hip) for CO > 128, CK % 8 == 0
int bd_conv3x3_pp_launch(const bd_conv_desc* d, int mode, const void* src, const void* w, const float* bias, const void* add,
                         const void* mask, void* dst, int flags, hipStream_t stream);
BD_KNOB int g_patch_pp128 = 0;    // staggered 128 / 64-channel-tile instances (conv3x3_pp128.hip): 0 = the 64-channel tile for Cout <= 64 (default: it
                          // halves the matrix and LDS work of those layers), 1 = every remaining shape (bd_conv_desc.route[1] bit 8: faster per
                          // launch, not per step -- see DESIGN.md), -1 = never (bit 9)
int bd_conv3x3_pp128_launch(const bd_conv_desc* d, int mode, const void* src, const void* w, const float* bias, const void* add,
                            const void* mask, void* dst, int flags, hipStream_t stream);
BD_KNOB int g_patch_dma = 1;
static long long* g_patch_dbg = nullptr;
extern "C" int bd_conv3x3_set_debug(long long* buf) { g_patch_dbg = buf; return 0; }      // bd_conv_desc.route[1] bit 3 clears it (register-staged weights everywhere)

// called from conv_igemm.hip for 3x3 / stride 1 / pad 1 descriptors; mode 0 fwd (src = x, geometry in == out),
// mode 1 dgrad (src = dY).  CK = reduction channels, CO = produced channels.
int bd_conv3x3_patch_launch(const bd_conv_desc* d, int mode, const void* src, const void* w, const float* bias,
                            const void* add, const void* mask, void* dst, int flags, hipStream_t stream) {
    if (g_patch_pp && bd_conv3x3_pp_launch(d, mode, src, w, bias, add, mask, dst, flags, stream) == 0) return 0;
    if (g_patch_pp128 >= 0 && bd_conv3x3_pp128_launch(d, mode, src, w, bias, add, mask, dst, flags, stream) == 0) return 0;
    C3Params p{};
    p.src = (const bf16_raw*)src; p.w = (const bf16_raw*)w; p.bias = bias;
    p.add = (const bf16_raw*)add; p.mask = (const bf16_raw*)mask; p.dst = (bf16_raw*)dst;
    p.dbg = g_patch_dbg;
    p.CK = mode == 0 ? d->Cin : d->Cout;
    p.CO = mode == 0 ? d->Cout : d->Cin;
    p.mode = mode; p.flags = flags; p.N = d->N; p.nseg = d->nseg;
    p.src_ppi = mode == 0 ? d->in_pix_per_img : d->out_pix_per_img;
    p.dst_ppi = mode == 0 ? d->out_pix_per_img : d->in_pix_per_img;
    int ps = 0;
    for (int s = 0; s < d->nseg; ++s) {
        CSeg& sg = p.seg[s];
        sg.patch_start = ps; sg.H = d->Ho[s]; sg.W = d->Wo[s]; sg.pw = cdiv(d->Wo[s], PW);
        sg.src_off = mode == 0 ? d->in_off[s] : d->out_off[s];
        sg.dst_off = mode == 0 ? d->out_off[s] : d->in_off[s];
        ps += cdiv(d->Ho[s], PH) * sg.pw;
    }
    p.patches_per_img = ps;
    p.total_patches = ps * d->N;
    p.n_tiles = cdiv(p.CO, TILE_CO);
    const int grid = cdiv(p.total_patches, NPATCH) * p.n_tiles;
    const size_t lds = X_BYTES + 2 * W_BYTES + TILE_CO * sizeof(float);
    BD_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_patch_kernel<false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_patch_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_patch_kernel<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_patch_kernel<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const bool dma = g_patch_dma && p.CK % 64 == 0;
    bd_note_kernel("conv3x3_patch_kernel");
    if (dma && mode == 0) hipLaunchKernelGGL((conv3x3_patch_kernel<true, 0>), dim3(grid), dim3(512), lds, stream, p);
    else if (dma) hipLaunchKernelGGL((conv3x3_patch_kernel<true, 1>), dim3(grid), dim3(512), lds, stream, p);
    else if (mode == 0) hipLaunchKernelGGL((conv3x3_patch_kernel<false, 0>), dim3(grid), dim3(512), lds, stream, p);
    else hipLaunchKernelGGL((conv3x3_patch_kernel<false, 1>), dim3(grid), dim3(512), lds, stream, p);
    return 0;
}
