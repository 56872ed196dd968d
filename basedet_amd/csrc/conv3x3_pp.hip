// 3x3 / stride-1 / pad-1 convolution forward and data-gradient, 256-channel tile, two staggered wave groups ("ping-pong").
//
// Same patch implicit GEMM as conv3x3.hip (a workgroup owns four 4x16 output patches and keeps their 6x18 input patches of one
// 64-channel K block in LDS; the nine taps read shifted rows of that image), re-cut after the 256x256 eight-phase GEMM structure
// of cdna_hip_programming.md ("The 256^2 8-phase template"):
//   * workgroup tile 256 output channels x 256 pixels, 8 waves = 2 channel halves x 4 patches, wave tile 128 x 64 (acc[8][4]):
//     24 ds_read_b128 per 64 MFMAs instead of 32, and the activations are read once for all 256 output channels;
//     (v_mfma_f32_32x32x16_bf16 instead of 16x16x32 -- same reads, same barriers -- measured the same: the kernel runs at ~80 % of what
//     the matrix pipe delivers at the clock the chip holds under this load, MI355X_MICROARCH.md)
//   * one tap (K = 64) = two phases of 32 MFMAs (the K halves; round 2: four phases of 16 measured 1 - 3 % slower).  A phase is
//         [ds_read the K half's fragments + two weight DMA pieces]  s_barrier  [32 MFMAs]  s_barrier
//     and the two channel halves (waves 0-3 / 4-7 = the two waves of every SIMD) run ONE barrier apart, so that on each SIMD one
//     wave is in its MFMA segment while its partner is in its load segment: the matrix pipe never waits for LDS;
//   * weights: ring of three 32 KB tap slots filled by LDS-DMA two taps ahead (counted vmcnt, raw s_barrier: the pieces stay
//     in flight across barriers); activations: one padded image, re-staged through registers once per K block (the groups
//     re-synchronise for that swap);
//   * round 3: the grid is one PERSISTENT workgroup per CU walking up to 16 tiles (the next tile's image, first weight taps and patch
//     decode are requested inside the current tile's last K block), and the last r tiles of a 256 k + r grid run as 4 r workgroups of
//     the 64-channel tile (conv3x3_pp128_body.h) at the end of the same launch.
// Used for Cout > 128 and Cin % 8 == 0 (heads incl. the 720- and 40-channel gradients, FPN outputs, res4/res5) when its grid fills the
// chip; the rest stays on conv3x3.hip.
#include "conv3x3_pp128_body.h"

namespace {

constexpr int PH = 4, PW = 16, IH = PH + 2, IW = PW + 2;
constexpr int NPATCH = 4;
constexpr int XROWS = NPATCH * IH * IW;        // 432
constexpr int X_PITCH = 144;
constexpr int X_BYTES = XROWS * X_PITCH;       // 62208
constexpr int XPASSES = 7;                     // 432 rows x 8 chunks / 512 threads
constexpr int TILE_CO = 256;
constexpr int W_SLOT = TILE_CO * 128;          // 32768: one tap's 256 x 64 weight tile
constexpr int NSLOT = 3;
constexpr int W_BYTES = NSLOT * W_SLOT;        // 98304
constexpr int GTAB_TILES = 16;                 // patch geometry of this workgroup's next 16 tiles (8 ints per patch), rebuilt by wave 0
constexpr int GTAB_BYTES = GTAB_TILES * NPATCH * 32;          // 2048
constexpr int LDS_BYTES = W_BYTES + X_BYTES + TILE_CO * 4 + GTAB_BYTES;   // 163584 of 163840
constexpr int MAX_SEG = BD_MAX_SEGS;
constexpr int TAIL_CO = 64;                   // channel tile of the tail workgroups

struct PSeg { int patch_start, H, W, pw, src_off, dst_off; float inv_pw; };

struct PParams {
    const bf16_raw* src;
    const bf16_raw* w;       // [CO][9][CK]
    const float* bias;
    const bf16_raw* add;
    const bf16_raw* mask;
    bf16_raw* dst;
    int CK, CO, flags, nseg;
    int src_ppi, dst_ppi;
    unsigned src_bytes;
    int patches_per_img, total_patches, n_tiles;
    int main_grid;           // workgroups [0, main_grid) are PERSISTENT: each runs the 256-channel tiles pt = b / n_tiles + k * (main_grid / n_tiles)
    int px_tiles;            // (k = 0, 1, ...; pt < px_tiles) of channel tile b % n_tiles over patches [0, total_patches); the rest of the grid runs
    int tail_end;            // the 64-channel tile of conv3x3_pp128_body.h over patches [total_patches, tail_end)
    float inv_ppi;           // 1 / patches_per_img (the patch indices are < 2^24: exact quotients by a float multiply and one correction)
    PSeg seg[MAX_SEG];
    float* gn_part;          // GNS instance only: per (patch, group of 8 channels) the (sum, sum of squares) of the fp32 results, [patch][32][2]
};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

// n / d for 0 <= n < 2^24, d > 0, inv = 1.0f / d: the float product is within one of the quotient (an integer division is ~40 VALU instructions)
__device__ __forceinline__ int pp_div(int n, int d, float inv) {
    int q = (int)((float)n * inv);
    const int r = n - q * d;
    q += (r >= d) - (r < 0);
    return q;
}

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
// activation image bank layout: see conv3x3.hip (column permutation + chunk order 0 2 1 3 make the padded rows conflict-free)
__device__ __forceinline__ int colperm(int f) { return f < 4 ? 2 * f : (f < 12 ? 2 * (f - 4) + 1 : 2 * (f - 8)); }
__device__ __forceinline__ int xpos(int chunk) { return (chunk >> 2) * 64 + ((((chunk & 1) << 1) | ((chunk >> 1) & 1)) << 4); }

#ifdef BD_PP_STAMP        // diagnostic build only (scripts/exp/pp_clock.py): in-kernel clock of workgroup 0 = d(s_memtime) / d(s_memrealtime) x 100 MHz
__device__ unsigned long long g_pp_stamp[64];
__device__ unsigned long long g_pp_span[2048];      // [2 b], [2 b + 1]: s_memrealtime (100 MHz, chip-wide) at the start / end of workgroup b
#endif
// -DBD_PP_ABLATE=<bits> (diagnostic builds, TIMING ONLY -- the results are wrong; scripts/exp/pp_power.sh; bit 2 = no stores in the fast-path epilogue): bit 0 = the second K half of every
// tap re-uses the first half's fragments (half of the loop's ds_read_b128 gone), bit 1 = no weight DMA inside the K loop (the ring keeps the
// prologue's three taps).  Same MFMAs, same barriers: what moves is LDS / L2 traffic -- and with it launch time and the clock the chip holds.
#ifndef BD_PP_ABLATE
#define BD_PP_ABLATE 0
#endif
__device__ unsigned long long g_pp_clk[2];           // bd_probe_kernel_clock("conv3x3_pp_kernel"): sums of workgroup 0's (shader cycles, 100 MHz ticks)
#define PP_FENCE() __builtin_amdgcn_sched_barrier(0)
#define PP_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

// GNS (round 6): the forward instance that also leaves GroupNorm's statistics (layers/head/point_head.py:47-58: conv -> GroupNorm(32) -> ReLU in
// FCOS's towers): a lane's 8 channels of a pixel ARE one group of GroupNorm(32, 256), so the epilogue adds its fp32 results (before the bf16
// rounding) and their squares per channel group over the patch's pixels inside the image, the 16 lanes of a row are summed by a fixed butterfly,
// and lane 0 of the row stores the pair -- gn_stats_final_kernel (norm.hip) then sums a level's patches exactly as it sums 128-pixel slots.
// A separate instantiation: the plain forward / data-gradient instances are untouched.  (Round 5 built this on the 253-register kernel: the
// sums spilled INSIDE the K loop and the launch lost what the statistics pass saved; the 16 address registers found in round 6 made room.)
template <int MODE, bool GNS = false>
__global__ __launch_bounds__(512, 1) void conv3x3_pp_kernel(const PParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((int)blockIdx.x >= p.main_grid) {       // tail tiles of a grid of 256 k + r pixel tiles (see bd_conv3x3_pp_launch): 4 r short workgroups
        pp128::body<MODE, TAIL_CO>(p, smem, (int)blockIdx.x - p.main_grid, (int)gridDim.x - p.main_grid, p.total_patches, p.tail_end,
                                   (p.CO + TAIL_CO - 1) / TAIL_CO);
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) bd_clk_mark(g_pp_clk, false);
    unsigned char* wbuf = smem;                                   // [3][W_SLOT]
    unsigned char* xbuf = smem + W_BYTES;                         // [X_BYTES]
    float* sbias = reinterpret_cast<float*>(smem + W_BYTES + X_BYTES);      // [256]

#ifdef BD_PP_STAMP
    int st_n = 0;
    const bool st_on = blockIdx.x == 100 % gridDim.x && threadIdx.x == 0;
#define PP_STAMP() do { if (st_on && st_n < 62) g_pp_stamp[st_n] = __builtin_amdgcn_s_memtime(); ++st_n; } while (0)
    PP_STAMP();                    // 0: kernel start
    if (st_on) g_pp_stamp[62] = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_pp_span[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#else
#define PP_STAMP() do { } while (0)
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wp = wave & 3;          // channel half (= stagger group), patch
    int bid = blockIdx.x;
    {
        const int nwg = p.main_grid;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // Persistent workgroups (round 3): workgroup b keeps its channel tile (so its weight-DMA addresses and bias vector) and walks the pixel
    // tiles b / n_tiles + k * pt_step.  The NEXT tile's patch decode, its first activation image and its first two weight taps are requested
    // inside the current tile's last K block, exactly as a K-block swap would request them -- no extra registers -- and land under that
    // block's MFMAs and the epilogue: the 9 400-cycle prologue (two memory round trips in front of the first MFMA, ~10 % of a tile) is paid
    // once per workgroup instead of once per tile.
    const int ct = bid % p.n_tiles;          // (n_tiles is 1 .. 3: the compiler's division by a uniform value stays on the scalar path)
    int pt = bid / p.n_tiles;
    const int pt_step = p.main_grid / p.n_tiles;
    const int co0 = ct * TILE_CO;

    // ---- activation staging: chunk id c = tid + 512 k -> LDS row (tid >> 3) + 64 k, 16-byte chunk tid & 7 ----
    // buffer loads: 32-bit per-lane byte offset + scalar K-block offset, and an offset past the end of the tensor (X_NONE) returns
    // zeros: halo / out-of-image rows need no predication and no 64-bit per-lane addresses
    constexpr unsigned X_NONE = 0x80000000u;          // >= num_records (the host checks that the tensor is < 2 GB)
    const int x_lds0 = (tid >> 3) * X_PITCH + xpos(tid & 7);
    unsigned x_off[XPASSES];

    // ---- patch geometry ----
    // The level search and the two divisions of a patch decode (~250 instructions over a 60-register level table) run ONCE per workgroup:
    // wave 0 decodes the 64 patches of this workgroup's (at most 16: the host sizes the grid for that) tiles side by side, one per lane,
    // into a table in LDS.  What a tile change costs inside the K loop is then two LDS reads, 24 lane reads and the seven staging offsets.
    int* gtab = reinterpret_cast<int*>(smem + W_BYTES + X_BYTES + TILE_CO * 4);
    const int pt_first = pt;
    auto build_table = [&](int k0) {
        const int k = k0 + (lane >> 2);
        const int ptile = pt_first + k * pt_step;
        const int pid = ptile * NPATCH + (lane & (NPATCH - 1));
        int vH = 0, vW = 0, vy = 0, vx = 0, vs = 0, vd = 0;
        if (ptile < p.px_tiles && pid < p.total_patches) {
            const int n = pp_div(pid, p.patches_per_img, p.inv_ppi);
            const int rem = pid - n * p.patches_per_img;
            // level search as selects over the (scalar) level table: no per-lane table fetch
            int g_start = p.seg[0].patch_start, g_H = p.seg[0].H, g_W = p.seg[0].W, g_pw = p.seg[0].pw;
            int g_src = p.seg[0].src_off, g_dst = p.seg[0].dst_off;
            float g_inv = p.seg[0].inv_pw;
#pragma unroll
            for (int q = 1; q < MAX_SEG; ++q) {
                const bool in = q < p.nseg && rem >= p.seg[q].patch_start;
                g_start = in ? p.seg[q].patch_start : g_start; g_H = in ? p.seg[q].H : g_H; g_W = in ? p.seg[q].W : g_W;
                g_pw = in ? p.seg[q].pw : g_pw; g_src = in ? p.seg[q].src_off : g_src; g_dst = in ? p.seg[q].dst_off : g_dst;
                g_inv = in ? p.seg[q].inv_pw : g_inv;
            }
            const int local = rem - g_start;
            const int by = pp_div(local, g_pw, g_inv), bx = local - by * g_pw;
            vy = by * PH; vx = bx * PW; vH = g_H; vW = g_W;
            vs = n * p.src_ppi + g_src;
            vd = n * p.dst_ppi + g_dst;
        }
        typedef __attribute__((ext_vector_type(4))) int i32x4_t;
        i32x4_t* e = reinterpret_cast<i32x4_t*>(gtab + ((k & (GTAB_TILES - 1)) * NPATCH + (lane & (NPATCH - 1))) * 8);
        e[0] = (i32x4_t){vy, vx, vH, vW};
        e[1] = (i32x4_t){vs, vd, 0, 0};
    };
    // tile k of this workgroup -> x_off[] and this wave's epilogue geometry
    auto decode = [&](int k, int& e_oy0, int& e_px0, int& e_H, int& e_W, int& e_dst) {
        int py0[NPATCH], px0[NPATCH], pH[NPATCH], pWd[NPATCH];
        int psrc[NPATCH], pdst[NPATCH];
        // (an opaque copy of the thread index: the per-thread row / column constants below are recomputed per tile -- ~40 instructions --
        // instead of being hoisted out of the tile loop and kept, i.e. spilled, across the MFMA loop)
        int tq = tid;
        asm volatile("" : "+v"(tq));
        {
            typedef __attribute__((ext_vector_type(4))) int i32x4_t;
            const i32x4_t* e = reinterpret_cast<const i32x4_t*>(gtab + ((k & (GTAB_TILES - 1)) * NPATCH + (tq & (NPATCH - 1))) * 8);
            const i32x4_t e0 = e[0], e1 = e[1];
#pragma unroll
            for (int q = 0; q < NPATCH; ++q) {
                py0[q] = __builtin_amdgcn_readlane(e0[0], q); px0[q] = __builtin_amdgcn_readlane(e0[1], q);
                pH[q] = __builtin_amdgcn_readlane(e0[2], q); pWd[q] = __builtin_amdgcn_readlane(e0[3], q);
                psrc[q] = __builtin_amdgcn_readlane(e1[0], q); pdst[q] = __builtin_amdgcn_readlane(e1[1], q);
            }
        }
#pragma unroll
        for (int k = 0; k < XPASSES; ++k) {
            const int row = (tq >> 3) + 64 * k;
            unsigned off = X_NONE;
            if (row < XROWS) {
                // the 64 rows of pass k straddle at most two patches, both known at compile time: selects between scalar registers
                const int lo = (64 * k) / (IH * IW), hi = (64 * k + 63) / (IH * IW) < NPATCH ? (64 * k + 63) / (IH * IW) : NPATCH - 1;
                const bool up = row >= hi * (IH * IW);
                const int pk = up ? hi : lo;
                const int rr = row - pk * (IH * IW);
                const int iy = rr / IW, ix = rr - iy * IW;
                const int qy = up ? py0[hi] : py0[lo], qx = up ? px0[hi] : px0[lo], H = up ? pH[hi] : pH[lo];
                const int W = up ? pWd[hi] : pWd[lo], qs = up ? psrc[hi] : psrc[lo];
                const int y = qy - 1 + iy, x = qx - 1 + ix;
                if (y >= 0 && x >= 0 && y < H && x < W) off = (unsigned)((qs + y * W + x) * p.CK + (tq & 7) * 8) * 2u;
            }
            x_off[k] = off;
        }
        e_oy0 = py0[0]; e_px0 = px0[0]; e_H = pH[0]; e_W = pWd[0]; e_dst = pdst[0];
#pragma unroll
        for (int q = 1; q < NPATCH; ++q)
            if (wp == q) { e_oy0 = py0[q]; e_px0 = px0[q]; e_H = pH[q]; e_W = pWd[q]; e_dst = pdst[q]; }
    };
    int c_oy0, c_px0, c_H, c_W, c_dst;          // this wave's patch of the tile in flight (wave-uniform)
    int kt = 0;                                 // index of that tile in this workgroup's sequence
    if (wave == 0) build_table(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PP_BARRIER();
    decode(0, c_oy0, c_px0, c_H, c_W, c_dst);
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.src), 0, p.src_bytes, 0x00020000);
    u32x4_t rx[XPASSES];
    auto load_x = [&](int cb) {
        int so = cb * 128;
        asm volatile("" : "+s"(so));
        // K tail (CK % 64 != 0, e.g. the 720-channel class-score gradient): chunks past CK read as zeros, so whatever finite weights
        // the DMA picks up beyond a row's CK channels (the next tap's; zeros past the end of the buffer) contribute nothing
        const bool dead = cb * 64 + (tid & 7) * 8 >= p.CK;
#pragma unroll
        for (int k = 0; k < XPASSES; ++k) rx[k] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, dead ? X_NONE : x_off[k], so, 0);
    };
    auto write_x = [&]() {
#pragma unroll
        for (int k = 0; k < XPASSES; ++k)
            if (k < XPASSES - 1 || (tid >> 3) + 64 * k < XROWS)
                *reinterpret_cast<u32x4_t*>(xbuf + x_lds0 + k * (64 * X_PITCH)) = rx[k];
    };

    // ---- weight DMA: one tap = 256 rows x 128 B = 32 pieces of 1 KiB; this wave owns pieces wave + 8 k (k = 0..3) = LDS rows
    // 8 pc .. 8 pc + 7; lane -> row lane >> 3, position lane & 7 (source chunk = position ^ (row & 7): the swizzle sits on the
    // SOURCE address).  LDS row lrow holds channel co(lrow): the permutation that gives every lane 8 consecutive channels in the
    // epilogue (as conv_igemm.hip).
    unsigned dma_src[4];       // byte offsets
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int pc = wave + 8 * k;
        const int lrow = 8 * pc + (lane >> 3);
        const int chunk = (lane & 7) ^ (lane >> 3);
        const int rho = lrow & 15;
        int co = co0 + (lrow & 192) + 32 * ((lrow >> 5) & 1) + 8 * (rho >> 2) + 4 * ((lrow >> 4) & 1) + (rho & 3);
        if (co >= p.CO) co = p.CO - 1;                 // rows past CO are never stored: any finite data will do
        dma_src[k] = (unsigned)(co * 9 * p.CK + chunk * 8) * 2u;
    }
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.w), 0, (unsigned)(p.CO * 9 * p.CK) * 2u, 0x00020000);
    bool dma_on = true;          // (BD_PP_ABLATE bit 1 clears it behind the prologue)
    auto dma_piece = [&](int tap, int cb, int slot, int k) {       // buffer_load_dwordx4 ... offen lds: fixed VGPR offset + scalar offset
        if ((BD_PP_ABLATE & 2) && !dma_on) return;
        unsigned char* l = wbuf + slot * W_SLOT + (wave + 8 * k) * 1024;
        int so = (tap * p.CK + cb * 64) * 2;
        asm volatile("" : "+s"(so));          // keep the tap offset in the scalar operand (else 36 hoisted per-tap VGPR offsets)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void_t*)l, 16, dma_src[k], so, 0, 0);
    };

    f32x4_t acc[8][4];

    // ---- fragment addressing ----
    const int frow = lane & 15, fchunk = lane >> 4;
    const unsigned char* a_base[2];          // K half kk: row wm*128 + frow of slot 0; + slot * W_SLOT + i * 2048 at compile time
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) a_base[kk] = wbuf + swz(wm * 128 + frow, kk * 4 + fchunk);
    // slot 2 sits past the 16-bit ds_read offset: its base is a register of its own -- an OPAQUE 32-bit LDS address, or the compiler folds it back
    // into a_base + 0x10000 + i * 2048 and keeps one address register per (fragment row, K half): 16 VGPRs live through the K loop (round 6)
    typedef __attribute__((address_space(3))) const unsigned char lds_u8_t;
    typedef __attribute__((address_space(3))) const bf16x8_t lds_bf16x8_t;
    unsigned a_hi[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        a_hi[kk] = (unsigned)(size_t)(lds_u8_t*)(a_base[kk]) + 2 * W_SLOT;
        asm volatile("" : "+v"(a_hi[kk]));
    }
    const unsigned char* b_base = xbuf + (wp * (IH * IW) + colperm(frow)) * X_PITCH + xpos(fchunk);

    // one tap = four phases: (A rows 0-63, K half 0) (A rows 64-127, K half 0) (rows 0-63, half 1) (rows 64-127, half 1); the B
    // fragments of a K half (all four patch rows) are read in the first phase of the pair and stay for the second: 24 ds_read_b128 per
    // tap, 32 fragment registers
    bf16x8_t fa[8], fb[4];
    auto load_a = [&](int kk, int slot) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (slot == 2) fa[i] = *(lds_bf16x8_t*)(size_t)(a_hi[kk] + i * 2048);
            else fa[i] = *reinterpret_cast<const bf16x8_t*>(a_base[kk] + slot * W_SLOT + i * 2048);
        }
    };
    auto load_b = [&](int kk, int t) {
        int dy = t / 3, dx = t - 3 * (t / 3);
        if (MODE == 1) { dy = 2 - dy; dx = 2 - dx; }           // dgrad: mirrored tap
#pragma unroll
        for (int j = 0; j < 4; ++j)
            fb[j] = *reinterpret_cast<const bf16x8_t*>(b_base + ((j + dy) * IW + dx) * X_PITCH + kk * 64);
    };
    auto mfma_rows = [&](int i0, int i1) {
#pragma unroll
        for (int i = i0; i < i1; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                // tied accumulator (D = C) in inline asm: under this register pressure the allocator otherwise rotates the 32
                // accumulator quads through the whole file and ends up spilling the staged activations
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(fa[i]), "v"(fb[j]));
    };

    const int kblocks = (p.CK + 63) >> 6;

    PP_STAMP();                    // 1: first decode done
    // ---- prologue: activation image of K block 0, taps 0, 1 and 2 (the whole ring) ----
    if (tid < TILE_CO) sbias[tid] = (p.bias && co0 + tid < p.CO) ? p.bias[co0 + tid] : 0.f;
    load_x(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) dma_piece(0, 0, 0, k);
#pragma unroll
    for (int k = 0; k < 4; ++k) dma_piece(1, 0, 1, k);
#pragma unroll
    for (int k = 0; k < 4; ++k) dma_piece(2, 0, 2, k);
    write_x();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    PP_BARRIER();
    PP_FENCE();
    if (BD_PP_ABLATE & 2) dma_on = false;

    PP_STAMP();                    // 2: prologue done = K loop of tile 0 starts; then per tile: last K block (+ decode) starts, K loop done, epilogue issued, next K loop starts
    // Every tile enters its K loop with taps 0 and 1 of K block 0 complete in the ring and tap 2 requested (the prologue; for a following
    // tile: tap 1 is completed in phase (8, 1) of the previous tile's K loop and awaited in front of that tile's first store, tap 2 -- its
    // slot is tap 8's -- is requested at the head of the epilogue): phase (0, 1) awaits nothing, and the first counted wait, in phase (1, 1),
    // comes ~3 500 cycles after the epilogue's stores.  Loads and stores share one vmcnt: a counted wait for a load that is younger than the
    // stores waits for the stores' acknowledgements too, with the matrix pipe idle.
#if defined(BD_PP_PRIO) && BD_PP_PRIO == 1
    if (wm == 1) __builtin_amdgcn_s_setprio(1);      // (MI355X_MICROARCH.md, two waves per SIMD, item 4: static priority for the younger half)
#endif
    for (;;) {
    const int pt_next = pt + pt_step;
    const bool more = pt_next < p.px_tiles;
    int n_oy0 = 0, n_px0 = 0, n_H = 0, n_W = 0, n_dst = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int cb = 0; cb < kblocks; ++cb) {
        const bool last_kb = cb + 1 == kblocks;
        const bool cont = !last_kb || more;        // something follows this K block: the next one, or the first one of the next tile
        const int nb = last_kb ? 0 : cb + 1;       // its K block index
        // the next tile's patches: x_off[] was last read by the loads of THIS K block (issued in the previous one / the prologue)
        if (last_kb) PP_STAMP();
        if (last_kb && more) decode(kt + 1, n_oy0, n_px0, n_H, n_W, n_dst);
        if (wm == 1) PP_BARRIER();                 // stagger: the second channel half runs one barrier behind
        PP_FENCE();
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                // A tap is TWO phases of 32 MFMAs (the K halves of the 64-channel block; 12 ds_read_b128 and two weight DMA pieces each).
                // With four phases of 16 the load segment (6 reads + a DMA piece, whose issue costs 100 - 185 cycles next to the reads)
                // was as long as the other group's 256-cycle MFMA segment and every phase paid its two barriers: the matrix pipe was
                // busy 56 % of the time.
                // ---------------- load segment ----------------
                const int slot = t % 3;
                if (!((BD_PP_ABLATE & 1) && kk == 1)) {
                load_b(kk, t);
                PP_FENCE();
                load_a(kk, slot);
                PP_FENCE();
                }
                if (kk == 1) {
                    // retire the pieces of tap t+1 (the last of them was issued in phase (t, 0)); piece 0 of tap t+2 stays in flight
                    if (BD_PP_ABLATE & 2) { }           // (no DMA in flight: the counted waits would only catch the activation loads early)
                    else if (!cont && t >= 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (t >= 1 || cb > 0) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                    if (t == 6) {
                        if (cont) load_x(nb);                      // consumed by the swap after tap 8
                        else {                                     // (nothing follows: tell the allocator that the staging registers hold nothing
#pragma unroll
                            for (int k = 0; k < XPASSES; ++k) asm volatile("" : "=v"(rx[k]));      // from here on -- no instruction)
                        }
                    }
                    PP_FENCE();
                }
                // weight DMA, two pieces per phase: (t, 0) issues the last piece of tap t+1 and the first of tap t+2, (t, 1) pieces 1 and 2 of
                // tap t+2.  The slot of tap t+2 was last read (tap t-1) in the load segment of phase (t-1, 1), one barrier before this
                // group's (t, 0) -- by the other group, whose reads were issued before that barrier and return within ~100 cycles; the DMA
                // data needs a memory round trip to arrive
                // (Round 6, measured and removed: the two pieces requested INSIDE the MFMA segment, behind its 2nd / 4th row of four MFMAs -- the
                // guide prices a piece at ~60 cycles among bare MFMAs against 100 - 185 here -- made the step 2.0 % slower: a vector-memory
                // instruction in the MFMA stream holds up the wave that owns the matrix pipe, here the partner's MFMAs run meanwhile;
                // profiles/r06_dma_pos.txt)
                if (kk == 0) {
                    if (t + 1 < 9) { if (t > 1 || cb > 0) dma_piece(t + 1, cb, (t + 1) % 3, 3); }
                    else if (cont) dma_piece(0, nb, 0, 3);
                    if (t + 2 < 9) { if (t > 0 || cb > 0) dma_piece(t + 2, cb, (t + 2) % 3, 0); }
                    else if (cont) dma_piece(t - 7, nb, (t + 2) % 3, 0);
                } else {
#pragma unroll
                    for (int pc = 1; pc < 3; ++pc) {
                        if (t + 2 < 9) { if (t > 0 || cb > 0) dma_piece(t + 2, cb, (t + 2) % 3, pc); }
                        else if (cont) dma_piece(t - 7, nb, (t + 2) % 3, pc);
                    }
                    if (t == 8 && last_kb && more) dma_piece(1, 0, 1, 3);      // the next tile's tap 1 complete: see the head of the tile loop
                }
                PP_FENCE();
                PP_BARRIER();
                // ---------------- MFMA segment ----------------
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PP_FENCE();
                // (no s_setprio around the MFMA segment since round 5.  Rounds 2-4 raised the priority here and dropped it behind the
                // segment; alternating on one box, three repetitions: those flips 648.0 / 645.1 / 641.9 img/s, one static priority for
                // waves 4-7 649.1 / 647.7 / 647.0, none 649.5 / 649.2 / 647.6 -- profiles/r05_prio.txt; -DBD_PP_PRIO=0 brings the flips back)
#if defined(BD_PP_PRIO) && BD_PP_PRIO == 0
                __builtin_amdgcn_s_setprio(1);
#endif
                mfma_rows(0, 8);
#if defined(BD_PP_PRIO) && BD_PP_PRIO == 0
                __builtin_amdgcn_s_setprio(0);
#endif
                PP_FENCE();
                PP_BARRIER();
                PP_FENCE();
            }
        }
        if (wm == 0) PP_BARRIER();                 // un-stagger: every wave has passed the same number of barriers, all reads retired
        PP_FENCE();
        if (cont) {
            write_x();
            // (last K block: this publishes the NEXT tile's image, in front of this tile's epilogue, where the waves are still in step --
            // behind the epilogue every wave would wait here for the slowest one's stores to issue; now a wave that is done walks into the
            // next tile's first load segment and meets the others at that phase's barrier)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PP_BARRIER();
            PP_FENCE();
        }
    }

    PP_STAMP();                    // K loop done
    // ---- epilogue ----
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the inline-asm MFMAs are opaque to the hazard recogniser: let the last ones retire
    int lq = lane;                            // (opaque copy: the epilogue's per-lane constants are not kept across the MFMA loop)
    asm volatile("" : "+v"(lq));
    const int cg = lq >> 4, erow = lq & 15;
    // (the GNS instance is launched without residual, gate or ReLU -- GroupNorm follows: compile-time constants, so that the 64 gate registers
    // and the general path below do not exist in it)
    const bool do_relu = !GNS && (p.flags & BD_EPI_RELU);
    const bool add_before = !GNS && (p.flags & BD_EPI_ADD_BEFORE) && p.add;
    const bool add_after = !GNS && (p.flags & BD_EPI_ADD_AFTER) && p.add;
    const bool do_mask = !GNS && (p.flags & BD_EPI_MASK) && p.mask;
    const int cbase = co0 + wm * 128 + 8 * cg;          // + 32 h
    const int oy0 = c_oy0, ox = c_px0 + colperm(erow), H = c_H, W = c_W, dbase = c_dst;
    // Fast paths (no residual operand): the epilogue of a one-workgroup-per-CU kernel is pure issue time -- the general loop below, with
    // its run-time flag branches, 64-bit index arithmetic and per-element selects, took 8 100 cycles per tile (stamped), a sixth of the
    // 256-channel head launch's K loop.  Here: one address per patch row, packed adds / converts, ReLU and the gate as packed 16-bit
    // integer operations on the converted pairs (round-to-nearest keeps sign and zero, so max(bf16(x), 0) == bf16(max(x, 0))).
    if (!add_before && !add_after) {
        typedef __attribute__((ext_vector_type(2))) float f32x2_e;
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_e;
        typedef __attribute__((ext_vector_type(2))) short i16x2_e;
        const i16x2_e relu_floor = do_relu ? (i16x2_e){0, 0} : (i16x2_e){-32768, -32768};          // max(x, -32768) = x: no ReLU
        bf16_raw* drow[4];
        const bf16_raw* mrow[4];
        bool okj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oy = oy0 + j;
            okj[j] = oy < H && ox < W;
            const long long e = (long long)(dbase + oy * W + ox) * p.CO + cbase;
            drow[j] = p.dst + e;
            mrow[j] = p.mask + e;
        }
        // ALL of the tile's gate operands are requested before its first store (round 3).  gfx950 has one vmcnt for loads and stores: with
        // the loads of channel group h + 1 issued after the stores of group h, the compiler's wait for those loads was vmcnt(0) -- it also
        // waited for the stores' acknowledgements, four serialised round trips per tile in a kernel that has one workgroup per CU (seen in
        // the ISA; the same pattern cost bottleneck_fused.hip a third of its tile time).  The K loop's fragments and staging registers are
        // dead here, so the 16 operands (64 VGPRs) fit beside the accumulators.
        u32x4_t mv[4][4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const bool okh = cbase + 32 * h < p.CO;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mv[h][j] = (u32x4_t){0u, 0u, 0u, 0u};
                if (do_mask && okh && okj[j]) mv[h][j] = *reinterpret_cast<const u32x4_t*>(mrow[j] + 32 * h);
            }
        }
        PP_FENCE();
        if (more) {            // the next tile's tap 2 (its slot was tap 8's): the four youngest requests in front of the first store
#pragma unroll
            for (int k = 0; k < 4; ++k) dma_piece(2, 0, 2, k);
        }
        PP_FENCE();
        float gsum[4] = {0.f, 0.f, 0.f, 0.f}, gsq[4] = {0.f, 0.f, 0.f, 0.f};          // GNS: this lane's (sum, sum of squares) per channel group h
        auto unit = [&](int h, int j, const f32x4_t b0, const f32x4_t b1) {
            u32x4_t o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4_t a = acc[2 * h + (k >> 1)][j];
                const f32x4_t bb = (k >> 1) ? b1 : b0;
                f32x2_e v = {a[2 * (k & 1)], a[2 * (k & 1) + 1]};
                v += (f32x2_e){bb[2 * (k & 1)], bb[2 * (k & 1) + 1]};
                if constexpr (GNS) {
                    if (okj[j]) { gsum[h] += v[0] + v[1]; gsq[h] += v[0] * v[0] + v[1] * v[1]; }
                }
                i16x2_e w = __builtin_bit_cast(i16x2_e, __builtin_convertvector(v, bf16x2_e));
                if (do_mask) {          // keep where the stored activation is > 0: sat(0 - m) >> 15 is all ones exactly for m > 0 (-0.0 = 0x8000 saturates to 32767)
                    // (written as the two packed instructions: the vector-builtin form of this gate was compiled into selects that
                    // read the first mask register for every k -- caught by test_patch_instances_agree_bitwise; op_sel_hi:[0,1]: both halves shift
                    // by the low half of the inline constant, whose high half is 0)
                    unsigned gate;
                    asm("v_pk_sub_i16 %0, 0, %1 clamp\n\tv_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]" : "=v"(gate) : "v"(mv[h][j][k]));
                    w &= __builtin_bit_cast(i16x2_e, gate);
                }
                w = __builtin_elementwise_max(w, relu_floor);
                o[k] = __builtin_bit_cast(unsigned, w);
            }
            return o;
        };
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const bool okh = cbase + 32 * h < p.CO;
            const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(sbias + wm * 128 + 8 * cg + 32 * h);
            const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(sbias + wm * 128 + 8 * cg + 32 * h + 4);
            if (h == 0) {
                // the first channel group is converted BEFORE the wait that precedes the first store: everything requested before the next
                // tile's tap 2 -- the gates, that tile's tap 1 and activation image -- must have landed before a store is in flight (loads
                // return in order: with no store outstanding the count is exact)
                u32x4_t o0[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o0[j] = unit(0, j, b0, b1);
                PP_FENCE();
                PP_STAMP();            // first channel group converted
                if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                PP_STAMP();            // ... and everything older than the next tile's tap 2 has landed
                PP_FENCE();
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (BD_PP_ABLATE & 4) asm volatile("" :: "v"(o0[j]));           // (bit 2, timing only: the fast path converts but does not store)
                    else if (okh && okj[j]) *reinterpret_cast<u32x4_t*>(drow[j]) = o0[j];          // (non-temporal stores: -0.4 % per step, profiles/r06_pp_power.txt)
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32x4_t o = unit(h, j, b0, b1);
                    if (BD_PP_ABLATE & 4) asm volatile("" :: "v"(o));
                    else if (okh && okj[j]) *reinterpret_cast<u32x4_t*>(drow[j] + 32 * h) = o;
                }
            }
        }
        if constexpr (GNS) {
            // the 16 lanes of a row (same channel groups, 16 pixel columns): butterfly over lane bits 3..0 -- a fixed order; lane 0 of the row
            // stores the four (sum, sum of squares) pairs of the wave's patch: part[(patch * 32 + group) * 2], group = channel / 8
            const int pid = pt * NPATCH + wp;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                float a = gsum[h], b = gsq[h];
#pragma unroll
                for (int m = 8; m >= 1; m >>= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
                if (erow == 0 && pid < p.total_patches && cbase + 32 * h < p.CO) {
                    float* o = p.gn_part + ((long long)pid * 32 + ((cbase + 32 * h) >> 3)) * 2;
                    *reinterpret_cast<f32x2_t*>(o) = (f32x2_t){a, b};
                }
            }
        }
    } else {
    // General path (a residual / accumulate operand): the operands of TWO channel groups (eight 16-byte units: up to 64 VGPRs of residuals +
    // gates) are requested together, then consumed and stored -- two round trips per tile instead of one per unit (the plain loop compiled to
    // load -> s_waitcnt vmcnt(0) -> use -> store for each of the 16 units, each wait also covering the previous unit's store).
    if (more) {
#pragma unroll
        for (int k = 0; k < 4; ++k) dma_piece(2, 0, 2, k);
    }
    long long erow_off[4];          // element offset of this lane's 8-channel group in patch row j (+ 32 h)
#pragma unroll
    for (int j = 0; j < 4; ++j) erow_off[j] = (long long)(dbase + (oy0 + j) * W + ox) * p.CO + cbase;
#pragma unroll
    for (int hp = 0; hp < 2; ++hp) {
        u32x4_t av[2][4], mv[2][4];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int h = 2 * hp + hh;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int oy = oy0 + j;
                const bool ok = cbase + 32 * h < p.CO && oy < H && ox < W;
                const long long idx = erow_off[j] + 32 * h;
                av[hh][j] = (u32x4_t){0u, 0u, 0u, 0u}; mv[hh][j] = (u32x4_t){0u, 0u, 0u, 0u};
                if (ok && (add_before || add_after)) av[hh][j] = *reinterpret_cast<const u32x4_t*>(p.add + idx);
                if (ok && do_mask) mv[hh][j] = *reinterpret_cast<const u32x4_t*>(p.mask + idx);
            }
        }
        if (hp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (before the first store: see the fast path)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int h = 2 * hp + hh;
            if (cbase + 32 * h >= p.CO) continue;
            float bias[8];
            {
                const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(sbias + wm * 128 + 8 * cg + 32 * h);
                const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(sbias + wm * 128 + 8 * cg + 32 * h + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) { bias[k] = b0[k]; bias[4 + k] = b1[k]; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int oy = oy0 + j;
                if (oy >= H || ox >= W) continue;
                const long long idx = erow_off[j] + 32 * h;
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = acc[2 * h + (k >> 2)][j][k & 3] + bias[k];
                const u32x4_t a4 = av[hh][j], m4 = mv[hh][j];
                if (add_before) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v[2 * k] += bf_lo(a4[k]); v[2 * k + 1] += bf_hi(a4[k]); }
                }
                if (do_relu) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
                }
                if (do_mask) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (!(bf_lo(m4[k]) > 0.f)) v[2 * k] = 0.f;
                        if (!(bf_hi(m4[k]) > 0.f)) v[2 * k + 1] = 0.f;
                    }
                }
                if (add_after) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v[2 * k] += bf_lo(a4[k]); v[2 * k + 1] += bf_hi(a4[k]); }
                }
                u32x4_t o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = pack_bf2(v[2 * k], v[2 * k + 1]);
                *reinterpret_cast<u32x4_t*>(p.dst + idx) = o;
            }
        }
    }
    }
    PP_STAMP();                    // epilogue issued
    // ---- next tile: its image and first two taps are in LDS (written above / by DMA), its first loads were retired in the K loop ----
    if (!more) break;
    ++kt;
    pt = pt_next;
    c_oy0 = n_oy0; c_px0 = n_px0; c_H = n_H; c_W = n_W; c_dst = n_dst;
    PP_FENCE();
    PP_STAMP();                    // next K loop starts
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) bd_clk_mark(g_pp_clk, true);
#ifdef BD_PP_STAMP
    if (st_on) { g_pp_stamp[63] = __builtin_amdgcn_s_memrealtime(); g_pp_stamp[61] = __builtin_amdgcn_s_memtime(); }
    if (threadIdx.x == 0 && blockIdx.x < 1024) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); g_pp_span[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

}  // namespace

// 0 = launched, 1 = shape not handled here (caller falls back to conv3x3.hip)
BD_KNOB int g_pp_tail_split = 1;     // bd_conv_desc.route[1] bit 13 clears it
BD_KNOB int g_pp_persistent = 1;     // bit 14 clears it: one workgroup per tile
extern BD_KNOB int g_patch_pp;       // 0 = never, 1 = where the makespan estimate favours it, 2 = wherever the shape allows (default)
static int pp_launch(const bd_conv_desc* d, int mode, const void* src, const void* w, const float* bias, const void* add,
                     const void* mask, void* dst, int flags, float* gn_part, hipStream_t stream) {
    PParams p{};
    p.gn_part = gn_part;
    p.CK = mode == 0 ? d->Cin : d->Cout;
    p.CO = mode == 0 ? d->Cout : d->Cin;
    p.src_ppi = mode == 0 ? d->in_pix_per_img : d->out_pix_per_img;
    p.dst_ppi = mode == 0 ? d->out_pix_per_img : d->in_pix_per_img;
    if (p.CK % 8 != 0 || p.CO <= 128 || p.CO % 8 != 0) return 1;
    if ((long long)d->N * p.src_ppi * p.CK * 2 >= 0x7fffffffll || (long long)d->N * p.dst_ppi >= 0x7fffffffll ||
        (long long)p.CO * 9 * p.CK >= 0x7fffffffll) return 1;
    p.src = (const bf16_raw*)src; p.w = (const bf16_raw*)w; p.bias = bias;
    p.add = (const bf16_raw*)add; p.mask = (const bf16_raw*)mask; p.dst = (bf16_raw*)dst;
    p.flags = flags; p.nseg = d->nseg;
    p.src_bytes = (unsigned)((long long)d->N * p.src_ppi * p.CK * 2);
    int ps = 0;
    for (int s = 0; s < d->nseg; ++s) {
        PSeg& sg = p.seg[s];
        sg.patch_start = ps; sg.H = d->Ho[s]; sg.W = d->Wo[s]; sg.pw = cdiv(d->Wo[s], PW); sg.inv_pw = 1.0f / (float)sg.pw;
        sg.src_off = mode == 0 ? d->in_off[s] : d->out_off[s];
        sg.dst_off = mode == 0 ? d->out_off[s] : d->in_off[s];
        ps += cdiv(d->Ho[s], PH) * sg.pw;
    }
    p.patches_per_img = ps;
    p.inv_ppi = 1.0f / (float)ps;
    p.total_patches = ps * d->N;
    if (p.total_patches >= (1 << 24)) return 1;
    p.n_tiles = cdiv(p.CO, TILE_CO);
    int grid = cdiv(p.total_patches, NPATCH) * p.n_tiles;
    // Tail split (round 3).  One workgroup per CU: a grid of 256 k + r tiles with a small r pays a whole extra round for r tiles (res4's
    // 3x3 at 16 x 50x84: 312 tiles = 2 rounds, 56 CUs busy in the second).  The last r pixel tiles go to the 64- / 128-channel instance
    // of conv3x3_pp128.hip instead (same accumulation order: the same bits): the LAST 4 r workgroups of this launch's grid run that body
    // (~0.4 of a 256-channel tile's time each), one launch, and they start as soon as the first CUs come free.
    int tail_wgs = 0;
    int px_tiles = cdiv(p.total_patches, NPATCH);
    p.tail_end = p.total_patches;
    const int num_cus = bd_num_cus();
    if (g_pp_tail_split && !gn_part && p.n_tiles == 1 && px_tiles > num_cus) {          // (the tail body leaves no GroupNorm statistics)
        const int r = px_tiles % num_cus;
        if (r > 0 && r * (TILE_CO / TAIL_CO) <= num_cus) {
            px_tiles -= r;
            p.total_patches = px_tiles * NPATCH;
            tail_wgs = cdiv(p.tail_end - p.total_patches, NPATCH) * cdiv(p.CO, TAIL_CO);
        }
    }
    // persistent workgroups: one per CU (a multiple of the channel tiles, so that a workgroup keeps its channel tile), fewer if the tiles run out
    p.px_tiles = px_tiles;
    // ... and at most GTAB_TILES tiles per workgroup (the kernel's geometry table): larger launches get a whole multiple of that grid
    p.main_grid = px_tiles * p.n_tiles;
    if (g_pp_persistent) {
        const int g1 = (num_cus / p.n_tiles) * p.n_tiles;
        const int rounds = cdiv(px_tiles * p.n_tiles, g1 * GTAB_TILES);
        if (g1 * rounds < p.main_grid) p.main_grid = g1 * rounds;
    }
    {
        // A workgroup here does the work of two 128-channel workgroups of conv3x3.hip in ~1.5x their time, but small grids quantise
        // worse (one workgroup per CU, 256 CUs).  Mode 1 takes the instance with the shorter estimated makespan; the default takes
        // this one regardless: equal time on the small grids in isolation, fewer LDS and HBM bytes, and 0.2 % faster steps.
        const int grid128 = cdiv(p.total_patches, NPATCH) * cdiv(p.CO, 128);
        if (!gn_part && g_patch_pp < 2 && 3 * cdiv(grid, 256) >= 2 * cdiv(grid128, 256)) return 1;
    }
    BD_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pp_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pp_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    bd_note_kernel("conv3x3_pp_kernel");
    if (gn_part) {
        BD_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pp_kernel<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        hipLaunchKernelGGL((conv3x3_pp_kernel<0, true>), dim3(p.main_grid), dim3(512), LDS_BYTES, stream, p);
    } else if (mode == 0) hipLaunchKernelGGL((conv3x3_pp_kernel<0>), dim3(p.main_grid + tail_wgs), dim3(512), LDS_BYTES, stream, p);
    else hipLaunchKernelGGL((conv3x3_pp_kernel<1>), dim3(p.main_grid + tail_wgs), dim3(512), LDS_BYTES, stream, p);
    return 0;
}

// (probe.hip) sums since the last reset; reset != 0 clears them afterwards
int bd_pp_clk_read(unsigned long long* out2, int reset) {
    if (hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_pp_clk), 16) != hipSuccess) return 1;
    if (reset) { const unsigned long long z[2] = {0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_pp_clk), z, 16) != hipSuccess) return 1; }
    return 0;
}

#ifdef BD_PP_STAMP
extern "C" int bd_debug_pp_stamp(unsigned long long* out2) {
    return hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_pp_stamp), 512) == hipSuccess ? 0 : 1;
}
extern "C" int bd_debug_pp_span(unsigned long long* out2048) {
    return hipMemcpyFromSymbol(out2048, HIP_SYMBOL(g_pp_span), 2048 * 8) == hipSuccess ? 0 : 1;
}
#endif

int bd_conv3x3_pp_launch(const bd_conv_desc* d, int mode, const void* src, const void* w, const float* bias, const void* add,
                         const void* mask, void* dst, int flags, hipStream_t stream) {
    return pp_launch(d, mode, src, w, bias, add, mask, dst, flags, nullptr, stream);
}

// The patch layout of this kernel over a multi-level output, for the consumers of the GNS instance's statistics (norm.hip): starts[l] = first
// patch of level l inside one image, starts[nseg] = patches per image.
int bd_conv3x3_pp_patch_starts(const bd_conv_desc* d, int* starts) {
    int ps = 0;
    for (int s = 0; s < d->nseg; ++s) { starts[s] = ps; ps += cdiv(d->Ho[s], PH) * cdiv(d->Wo[s], PW); }
    starts[d->nseg] = ps;
    return ps;
}

extern "C" size_t bd_conv2d_fwd_gnstats_bytes(const bd_conv_desc* d) {
    if (!d || d->nseg < 1 || d->nseg > BD_MAX_SEGS) return 0;
    int starts[BD_MAX_SEGS + 1];
    return (size_t)d->N * bd_conv3x3_pp_patch_starts(d, starts) * 32 * 2 * sizeof(float);
}

extern "C" int bd_conv2d_fwd_gnstats(const bd_conv_desc* d, const void* x, const void* w_packed, const float* bias, void* y, float* part,
                                     size_t part_bytes, bd_stream_t stream) {
    BD_ROUTE(d);
    BD_REQUIRE(d && x && w_packed && y && part, "conv2d_fwd_gnstats: null pointer");
    BD_REQUIRE(d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1 && d->Cout == 256 && d->Cin % 64 == 0,
               "conv2d_fwd_gnstats: 3x3 / stride 1 / pad 1 into 256 channels (GroupNorm(32, 256)) only; got %dx%d s%d p%d %d -> %d", d->R, d->S,
               d->stride, d->pad, d->Cin, d->Cout);
    for (int s = 0; s < d->nseg; ++s)
        BD_REQUIRE(d->Hi[s] == d->Ho[s] && d->Wi[s] == d->Wo[s], "conv2d_fwd_gnstats: level %d changes size", s);
    if (part_bytes < bd_conv2d_fwd_gnstats_bytes(d)) { bd_set_error("conv2d_fwd_gnstats: statistics buffer too small"); return BD_EWORKSPACE; }
    if (pp_launch(d, 0, x, w_packed, bias, nullptr, nullptr, y, 0, part, (hipStream_t)stream) != 0) {
        bd_set_error("conv2d_fwd_gnstats: shape not taken by conv3x3_pp_kernel (tensor too large for 32-bit offsets?)");
        return BD_EINVAL;
    }
    BD_CHECK_LAUNCH("bd_conv2d_fwd_gnstats");
    return BD_OK;
}
