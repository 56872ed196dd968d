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
//     re-synchronise for that swap).
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
constexpr int LDS_BYTES = W_BYTES + X_BYTES + TILE_CO * 4;   // 161536
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
    int main_grid;           // workgroups [0, main_grid) run this kernel's 256-channel tiles over patches [0, total_patches); the rest of
    int tail_end;            // the grid runs the 64-channel tile of conv3x3_pp128_body.h over patches [total_patches, tail_end)
    float inv_ppi;           // 1 / patches_per_img (the patch indices are < 2^24: exact quotients by a float multiply and one correction)
    PSeg seg[MAX_SEG];
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
__device__ unsigned long long g_pp_stamp[8];
#endif
#define PP_FENCE() __builtin_amdgcn_sched_barrier(0)
#define PP_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

template <int MODE>
__global__ __launch_bounds__(512, 1) void conv3x3_pp_kernel(const PParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((int)blockIdx.x >= p.main_grid) {       // tail tiles of a grid of 256 k + r pixel tiles (see bd_conv3x3_pp_launch): 4 r short workgroups
        pp128::body<MODE, TAIL_CO>(p, smem, (int)blockIdx.x - p.main_grid, (int)gridDim.x - p.main_grid, p.total_patches, p.tail_end,
                                   (p.CO + TAIL_CO - 1) / TAIL_CO);
        return;
    }
    unsigned char* wbuf = smem;                                   // [3][W_SLOT]
    unsigned char* xbuf = smem + W_BYTES;                         // [X_BYTES]
    float* sbias = reinterpret_cast<float*>(smem + W_BYTES + X_BYTES);      // [256]

#ifdef BD_PP_STAMP
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wp = wave & 3;          // channel half (= stagger group), patch
    int bid = blockIdx.x;
    {
        const int nwg = p.main_grid;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int pt = bid / p.n_tiles;          // (n_tiles is 1 .. 3: the compiler's division by a uniform value stays on the scalar path)
    const int ct = bid - pt * p.n_tiles;
    const int co0 = ct * TILE_CO;

    // ---- geometry of the four patches (workgroup-uniform) ----
    // Lane k of every wave decodes patch k (the four decodes side by side instead of a four-times-longer serial chain in front of the
    // first load), then the results are broadcast into scalar registers, where they live through the MFMA loop.
    int py0[NPATCH], px0[NPATCH], pH[NPATCH], pWd[NPATCH];
    int psrc[NPATCH], pdst[NPATCH];
    {
        const int pid = pt * NPATCH + (lane & (NPATCH - 1));
        int vH = 0, vW = 0, vy = 0, vx = 0, vs = 0, vd = 0;
        if (pid < p.total_patches) {
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
#pragma unroll
        for (int k = 0; k < NPATCH; ++k) {
            py0[k] = __builtin_amdgcn_readlane(vy, k); px0[k] = __builtin_amdgcn_readlane(vx, k);
            pH[k] = __builtin_amdgcn_readlane(vH, k); pWd[k] = __builtin_amdgcn_readlane(vW, k);
            psrc[k] = __builtin_amdgcn_readlane(vs, k); pdst[k] = __builtin_amdgcn_readlane(vd, k);
        }
    }

    // ---- activation staging: chunk id c = tid + 512 k -> LDS row (tid >> 3) + 64 k, 16-byte chunk tid & 7 ----
    // buffer loads: 32-bit per-lane byte offset + scalar K-block offset, and an offset past the end of the tensor (X_NONE) returns
    // zeros: halo / out-of-image rows need no predication and no 64-bit per-lane addresses
    constexpr unsigned X_NONE = 0x80000000u;          // >= num_records (the host checks that the tensor is < 2 GB)
    const int x_lds0 = (tid >> 3) * X_PITCH + xpos(tid & 7);
    unsigned x_off[XPASSES];
#pragma unroll
    for (int k = 0; k < XPASSES; ++k) {
        const int row = (tid >> 3) + 64 * k;
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
            if (y >= 0 && x >= 0 && y < H && x < W) off = (unsigned)((qs + y * W + x) * p.CK + (tid & 7) * 8) * 2u;
        }
        x_off[k] = off;
    }
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
    auto dma_piece = [&](int tap, int cb, int slot, int k) {       // buffer_load_dwordx4 ... offen lds: fixed VGPR offset + scalar offset
        unsigned char* l = wbuf + slot * W_SLOT + (wave + 8 * k) * 1024;
        int so = (tap * p.CK + cb * 64) * 2;
        asm volatile("" : "+s"(so));          // keep the tap offset in the scalar operand (else 36 hoisted per-tap VGPR offsets)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void_t*)l, 16, dma_src[k], so, 0, 0);
    };

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // ---- fragment addressing ----
    const int frow = lane & 15, fchunk = lane >> 4;
    const unsigned char* a_base[2];          // K half kk: row wm*128 + frow of slot 0; + slot * W_SLOT + i * 2048 at compile time
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) a_base[kk] = wbuf + swz(wm * 128 + frow, kk * 4 + fchunk);
    const unsigned char* a_hi[2] = {a_base[0] + 2 * W_SLOT, a_base[1] + 2 * W_SLOT};     // slot 2 (ds_read offsets are 16 bit)
    const unsigned char* b_base = xbuf + (wp * (IH * IW) + colperm(frow)) * X_PITCH + xpos(fchunk);

    // one tap = four phases: (A rows 0-63, K half 0) (A rows 64-127, K half 0) (rows 0-63, half 1) (rows 64-127, half 1); the B
    // fragments of a K half (all four patch rows) are read in the first phase of the pair and stay for the second: 24 ds_read_b128 per
    // tap, 32 fragment registers
    bf16x8_t fa[8], fb[4];
    auto load_a = [&](int kk, int slot) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned char* base = slot == 2 ? a_hi[kk] : a_base[kk] + slot * W_SLOT;
            fa[i] = *reinterpret_cast<const bf16x8_t*>(base + i * 2048);
        }
    };
    auto load_b = [&](int kk, int t) {
        int dy = t / 3, dx = t - 3 * (t / 3);
        if (MODE == 1) { dy = 2 - dy; dx = 2 - dx; }           // dgrad: mirrored tap
#pragma unroll
        for (int j = 0; j < 4; ++j)
            fb[j] = *reinterpret_cast<const bf16x8_t*>(b_base + ((j + dy) * IW + dx) * X_PITCH + kk * 64);
    };
    auto mfma_khalf = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                // tied accumulator (D = C) in inline asm: under this register pressure the allocator otherwise rotates the 32
                // accumulator quads through the whole file and ends up spilling the staged activations
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(fa[i]), "v"(fb[j]));
    };

    const int kblocks = (p.CK + 63) >> 6;

#ifdef BD_PP_STAMP
    const unsigned long long st_t1 = __builtin_amdgcn_s_memtime();
#endif
    // ---- prologue: activation image of K block 0, taps 0 and 1 ----
    if (tid < TILE_CO) sbias[tid] = (p.bias && co0 + tid < p.CO) ? p.bias[co0 + tid] : 0.f;
    load_x(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) dma_piece(0, 0, 0, k);
#pragma unroll
    for (int k = 0; k < 4; ++k) dma_piece(1, 0, 1, k);
    write_x();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    PP_BARRIER();
    PP_FENCE();

#ifdef BD_PP_STAMP
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int cb = 0; cb < kblocks; ++cb) {
        const bool last_kb = cb + 1 == kblocks;
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
                load_b(kk, t);
                PP_FENCE();
                load_a(kk, slot);
                PP_FENCE();
                if (kk == 1) {
                    // retire the pieces of tap t+1 (the last of them was issued in phase (t, 0)); piece 0 of tap t+2 stays in flight
                    if (last_kb && t >= 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                    if (t == 6 && !last_kb) load_x(cb + 1);        // consumed by the swap after tap 8
                    PP_FENCE();
                }
                // weight DMA, two pieces per phase: (t, 0) issues the last piece of tap t+1 and the first of tap t+2, (t, 1) pieces 1 and 2 of
                // tap t+2.  The slot of tap t+2 was last read (tap t-1) in the load segment of phase (t-1, 1), one barrier before this
                // group's (t, 0) -- by the other group, whose reads were issued before that barrier and return within ~100 cycles; the DMA
                // data needs a memory round trip to arrive
                if (kk == 0) {
                    if (t + 1 < 9) { if (t > 0 || cb > 0) dma_piece(t + 1, cb, (t + 1) % 3, 3); }
                    else if (!last_kb) dma_piece(0, cb + 1, 0, 3);
                    if (t + 2 < 9) dma_piece(t + 2, cb, (t + 2) % 3, 0);
                    else if (!last_kb) dma_piece(t - 7, cb + 1, (t + 2) % 3, 0);
                } else {
#pragma unroll
                    for (int pc = 1; pc < 3; ++pc) {
                        if (t + 2 < 9) dma_piece(t + 2, cb, (t + 2) % 3, pc);
                        else if (!last_kb) dma_piece(t - 7, cb + 1, (t + 2) % 3, pc);
                    }
                }
                PP_FENCE();
                PP_BARRIER();
                // ---------------- MFMA segment ----------------
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PP_FENCE();
                __builtin_amdgcn_s_setprio(1);
                mfma_khalf();
                __builtin_amdgcn_s_setprio(0);
                PP_FENCE();
                PP_BARRIER();
                PP_FENCE();
            }
        }
        if (wm == 0) PP_BARRIER();                 // un-stagger: every wave has passed the same number of barriers, all reads retired
        PP_FENCE();
        if (!last_kb) {
            write_x();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PP_BARRIER();
            PP_FENCE();
        }
    }

#ifdef BD_PP_STAMP
    const unsigned long long st_t2 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 300 % gridDim.x && tid == 0) { g_pp_stamp[0] = st_t2 - st_c0; g_pp_stamp[1] = __builtin_amdgcn_s_memrealtime() - st_r0; g_pp_stamp[2] = st_c0 - st_t0; g_pp_stamp[5] = st_t1 - st_t0; }
#endif
    // ---- epilogue ----
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the inline-asm MFMAs are opaque to the hazard recogniser: let the last ones retire
    const int cg = lane >> 4;
    const bool do_relu = p.flags & BD_EPI_RELU;
    const bool add_before = (p.flags & BD_EPI_ADD_BEFORE) && p.add;
    const bool add_after = (p.flags & BD_EPI_ADD_AFTER) && p.add;
    const bool do_mask = (p.flags & BD_EPI_MASK) && p.mask;
    const int cbase = co0 + wm * 128 + 8 * cg;          // + 32 h
    int oy0 = py0[0], ox = px0[0] + colperm(frow), H = pH[0], W = pWd[0], dbase = pdst[0];
#pragma unroll
    for (int q = 1; q < NPATCH; ++q)
        if (wp == q) { oy0 = py0[q]; ox = px0[q] + colperm(frow); H = pH[q]; W = pWd[q]; dbase = pdst[q]; }
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
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const bool okh = cbase + 32 * h < p.CO;
            const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(sbias + wm * 128 + 8 * cg + 32 * h);
            const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(sbias + wm * 128 + 8 * cg + 32 * h + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x4_t o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4_t a = acc[2 * h + (k >> 1)][j];
                    const f32x4_t bb = (k >> 1) ? b1 : b0;
                    f32x2_e v = {a[2 * (k & 1)], a[2 * (k & 1) + 1]};
                    v += (f32x2_e){bb[2 * (k & 1)], bb[2 * (k & 1) + 1]};
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
                if (okh && okj[j]) *reinterpret_cast<u32x4_t*>(drow[j] + 32 * h) = o;
            }
        }
        return;
    }
    // General path (a residual / accumulate operand): the operands of TWO channel groups (eight 16-byte units: up to 64 VGPRs of residuals +
    // gates) are requested together, then consumed and stored -- two round trips per tile instead of one per unit (the plain loop compiled to
    // load -> s_waitcnt vmcnt(0) -> use -> store for each of the 16 units, each wait also covering the previous unit's store).
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
                const long long idx = (long long)(dbase + oy * W + ox) * p.CO + cbase + 32 * h;
                av[hh][j] = (u32x4_t){0u, 0u, 0u, 0u}; mv[hh][j] = (u32x4_t){0u, 0u, 0u, 0u};
                if (ok && (add_before || add_after)) av[hh][j] = *reinterpret_cast<const u32x4_t*>(p.add + idx);
                if (ok && do_mask) mv[hh][j] = *reinterpret_cast<const u32x4_t*>(p.mask + idx);
            }
        }
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
                const long long idx = (long long)(dbase + oy * W + ox) * p.CO + cbase + 32 * h;
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
#ifdef BD_PP_STAMP
    {
        const unsigned long long st_t3 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st_t4 = __builtin_amdgcn_s_memtime();
        if (blockIdx.x == 300 % gridDim.x && tid == 0) { g_pp_stamp[3] = st_t3 - st_t2; g_pp_stamp[4] = st_t4 - st_t3; }
    }
#endif
}

}  // namespace

// 0 = launched, 1 = shape not handled here (caller falls back to conv3x3.hip)
int g_pp_tail_split = 1;     // bd_conv_set_patch3x3 bit 13 clears it
extern int g_patch_pp;       // 0 = never, 1 = where the makespan estimate favours it, 2 = wherever the shape allows (default)
int bd_conv3x3_pp_launch(const bd_conv_desc* d, int mode, const void* src, const void* w, const float* bias, const void* add,
                         const void* mask, void* dst, int flags, hipStream_t stream) {
    PParams p{};
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
    p.main_grid = grid;
    p.tail_end = p.total_patches;
    static int num_cus = 0;
    if (!num_cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            num_cus <= 0) num_cus = 256;
    }
    if (g_pp_tail_split && p.n_tiles == 1 && grid > num_cus) {
        const int r = grid % num_cus;
        if (r > 0 && r * (TILE_CO / TAIL_CO) <= num_cus) {
            p.main_grid = grid - r;
            p.total_patches = p.main_grid * NPATCH;
            tail_wgs = cdiv(p.tail_end - p.total_patches, NPATCH) * cdiv(p.CO, TAIL_CO);
        }
    }
    {
        // A workgroup here does the work of two 128-channel workgroups of conv3x3.hip in ~1.5x their time, but small grids quantise
        // worse (one workgroup per CU, 256 CUs).  Mode 1 takes the instance with the shorter estimated makespan; the default takes
        // this one regardless: equal time on the small grids in isolation, fewer LDS and HBM bytes, and 0.2 % faster steps.
        const int grid128 = cdiv(p.total_patches, NPATCH) * cdiv(p.CO, 128);
        if (g_patch_pp < 2 && 3 * cdiv(grid, 256) >= 2 * cdiv(grid128, 256)) return 1;
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pp_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pp_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    if (mode == 0) hipLaunchKernelGGL((conv3x3_pp_kernel<0>), dim3(p.main_grid + tail_wgs), dim3(512), LDS_BYTES, stream, p);
    else hipLaunchKernelGGL((conv3x3_pp_kernel<1>), dim3(p.main_grid + tail_wgs), dim3(512), LDS_BYTES, stream, p);
    return 0;
}

#ifdef BD_PP_STAMP
extern "C" int bd_debug_pp_stamp(unsigned long long* out2) {
    return hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_pp_stamp), 64) == hipSuccess ? 0 : 1;
}
#endif
