// Device body of the staggered 128- / 64-channel-tile 3x3 patch kernel (conv3x3_pp128.hip), shared with conv3x3_pp.hip, whose launches
// run it in their last workgroups for the tail tiles of a grid that does not fill a whole round of CUs.
#pragma once
#include "common.h"

namespace pp128 {


constexpr int PH = 4, PW = 16, IH = PH + 2, IW = PW + 2;
constexpr int NPATCH = 4;
constexpr int XROWS = NPATCH * IH * IW;        // 432
constexpr int X_PITCH = 144;
constexpr int X_BYTES = XROWS * X_PITCH;       // 62208
constexpr int XPASSES = 7;                     // 432 rows x 8 chunks / 512 threads
constexpr int NSLOT = 4;
// TCO = channel tile: 128 (wave tile 64 x 64) or 64 (wave tile 32 x 64, for the <= 64-channel layers: no wasted matrix work)
template <int TCO> struct Tile {
    static constexpr int W_SLOT = TCO * 128;               // one tap's TCO x 64 weight tile: 16384 / 8192
    static constexpr int W_BYTES = NSLOT * W_SLOT;
    static constexpr int LDS_BYTES = W_BYTES + X_BYTES + TCO * 4;
    static constexpr int MREP = TCO / 32;                  // 16-row MFMA tiles per wave: 4 / 2
    static constexpr int NPIECE = TCO / 64;                // 1 KiB DMA pieces per wave and tap: 2 / 1
};
constexpr int MAX_SEG = BD_MAX_SEGS;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
// activation image bank layout: see conv3x3.hip (column permutation + chunk order 0 2 1 3 make the padded rows conflict-free)
__device__ __forceinline__ int colperm(int f) { return f < 4 ? 2 * f : (f < 12 ? 2 * (f - 4) + 1 : 2 * (f - 8)); }
__device__ __forceinline__ int xpos(int chunk) { return (chunk >> 2) * 64 + ((((chunk & 1) << 1) | ((chunk >> 1) & 1)) << 4); }

#define PP_FENCE() __builtin_amdgcn_sched_barrier(0)
#define PP_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)


// P: any parameter block with the fields of conv3x3_pp128.hip's PParams; the patch range [patch_begin, patch_end), the number of channel
// tiles and the workgroup index / count come as arguments (conv3x3_pp.hip runs this body in the LAST workgroups of its own grid)
template <int MODE, int TCO, class P>
__device__ __forceinline__ void body(const P& p, unsigned char* smem, int bid, const int nwg, const int patch_begin, const int patch_end, const int n_tiles) {
    constexpr int TILE_CO = TCO, W_SLOT = Tile<TCO>::W_SLOT, W_BYTES = Tile<TCO>::W_BYTES, MREP = Tile<TCO>::MREP, NPIECE = Tile<TCO>::NPIECE;
    constexpr int HALF = TCO / 2;                          // channels per wave group
    unsigned char* wbuf = smem;                                   // [4][W_SLOT]
    unsigned char* xbuf = smem + W_BYTES;                         // [X_BYTES]
    float* sbias = reinterpret_cast<float*>(smem + W_BYTES + X_BYTES);      // [128]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wp = wave & 3;          // channel half (= stagger group), patch
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int pt = bid / n_tiles;
    const int ct = bid - pt * n_tiles;
    const int co0 = ct * TILE_CO;

    // ---- geometry of the four patches (workgroup-uniform) ----
    int py0[NPATCH], px0[NPATCH], pH[NPATCH], pWd[NPATCH];
    int psrc[NPATCH], pdst[NPATCH];
#pragma unroll
    for (int k = 0; k < NPATCH; ++k) {
        const int pid = patch_begin + pt * NPATCH + k;
        pH[k] = 0; pWd[k] = 0; py0[k] = 0; px0[k] = 0; psrc[k] = 0; pdst[k] = 0;
        if (pid < patch_end) {
            const int n = pid / p.patches_per_img;
            const int rem = pid - n * p.patches_per_img;
            int s = 0;
#pragma unroll
            for (int q = 1; q < MAX_SEG; ++q)
                if (q < p.nseg && rem >= p.seg[q].patch_start) s = q;
            const auto sg = p.seg[s];
            const int local = rem - sg.patch_start;
            const int by = local / sg.pw, bx = local - by * sg.pw;
            py0[k] = by * PH; px0[k] = bx * PW; pH[k] = sg.H; pWd[k] = sg.W;
            psrc[k] = n * p.src_ppi + sg.src_off;
            pdst[k] = n * p.dst_ppi + sg.dst_off;
        }
        // the divisions run on the vector ALU: move the (uniform) results back to scalar registers, they live through the MFMA loop
        py0[k] = __builtin_amdgcn_readfirstlane(py0[k]); px0[k] = __builtin_amdgcn_readfirstlane(px0[k]);
        pH[k] = __builtin_amdgcn_readfirstlane(pH[k]); pWd[k] = __builtin_amdgcn_readfirstlane(pWd[k]);
        psrc[k] = __builtin_amdgcn_readfirstlane(psrc[k]); pdst[k] = __builtin_amdgcn_readfirstlane(pdst[k]);
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

    // ---- weight DMA: one tap = 128 rows x 128 B = 16 pieces of 1 KiB; this wave owns pieces wave + 8 k (k = 0, 1) = LDS rows
    // 8 pc .. 8 pc + 7; lane -> row lane >> 3, position lane & 7 (source chunk = position ^ (row & 7): the swizzle sits on the
    // SOURCE address).  LDS row lrow holds channel co(lrow): the permutation that gives every lane 8 consecutive channels in the
    // epilogue (as conv_igemm.hip).
    unsigned dma_src[2];            // byte offsets (fixed size: a template-dependent size here keeps clang from emitting the host stub)
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) {
        const int pc = wave + 8 * k;
        const int lrow = 8 * pc + (lane >> 3);
        const int chunk = (lane & 7) ^ (lane >> 3);
        const int rho = lrow & 15;
        int co = TCO == 128 ? co0 + (lrow & 64) + 32 * ((lrow >> 5) & 1) + 8 * (rho >> 2) + 4 * ((lrow >> 4) & 1) + (rho & 3)
                            : co0 + (lrow & 32) + 8 * (rho >> 2) + 4 * ((lrow >> 4) & 1) + (rho & 3);
        if (co >= p.CO) co = p.CO - 1;                 // rows past CO are never stored: any finite data will do
        dma_src[k] = (unsigned)(co * 9 * p.CK + chunk * 8) * 2u;
    }
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.w), 0, (unsigned)(p.CO * 9 * p.CK) * 2u, 0x00020000);
    // piece k of the tap with index `tap` (0..8) of K block `cb`; gtap = its running index over the whole tile (ring slot gtap & 3)
    auto dma_piece = [&](int tap, int cb, int gtap, int k) {
        if (k >= NPIECE) return;               // the 64-channel tile has one piece per wave and tap
        unsigned char* l = wbuf + (gtap & 3) * W_SLOT + (wave + 8 * k) * 1024;
        int so = (tap * p.CK + cb * 64) * 2;
        asm volatile("" : "+s"(so));          // keep the tap offset in the scalar operand
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void_t*)l, 16, dma_src[k < NPIECE ? k : 0], so, 0, 0);
    };

    f32x4_t acc[MREP][4];
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // ---- fragment addressing ----
    const int frow = lane & 15, fchunk = lane >> 4;
    const unsigned char* a_base[2];          // K half kk: row wm*64 + frow of slot 0; + slot * W_SLOT (run time) + i * 2048
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) a_base[kk] = wbuf + swz(wm * HALF + frow, kk * 4 + fchunk);
    const unsigned char* b_base = xbuf + (wp * (IH * IW) + colperm(frow)) * X_PITCH + xpos(fchunk);

    // one tap = two phases (K halves): A rows 0-63 of this wave's channel half + the B fragments of all four patch rows
    bf16x8_t fa[MREP], fb[4];
    auto load_a = [&](int kk, int gtap) {
        const unsigned char* base = a_base[kk] + (gtap & 3) * W_SLOT;
#pragma unroll
        for (int i = 0; i < MREP; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(base + i * 2048);
    };
    auto load_b = [&](int kk, int t) {
        int dy = t / 3, dx = t - 3 * (t / 3);
        if (MODE == 1) { dy = 2 - dy; dx = 2 - dx; }           // dgrad: mirrored tap
#pragma unroll
        for (int j = 0; j < 4; ++j)
            fb[j] = *reinterpret_cast<const bf16x8_t*>(b_base + ((j + dy) * IW + dx) * X_PITCH + kk * 64);
    };
    auto mfma_tile = [&]() {
        constexpr int i0 = 0;
#pragma unroll
        for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                // tied accumulator (D = C) in inline asm: under this register pressure the allocator otherwise rotates the 32
                // accumulator quads through the whole file and ends up spilling the staged activations
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i0 + i][j]) : "v"(fa[i]), "v"(fb[j]));
    };

    const int kblocks = (p.CK + 63) >> 6;

    // ---- prologue: activation image of K block 0; taps 0 and 1 and the first piece of tap 2 (what the steady-state schedule would
    // have issued before phase (0, 0)) ----
    if (tid < TILE_CO) sbias[tid] = (p.bias && co0 + tid < p.CO) ? p.bias[co0 + tid] : 0.f;
    load_x(0);
    dma_piece(0, 0, 0, 0); dma_piece(0, 0, 0, 1);
    dma_piece(1, 0, 1, 0); dma_piece(1, 0, 1, 1);
    dma_piece(2, 0, 2, 0);
    write_x();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    PP_BARRIER();
    PP_FENCE();

    for (int cb = 0; cb < kblocks; ++cb) {
        const bool last_kb = cb + 1 == kblocks;
        const int g0 = cb * 9;                     // running tap index of tap 0 of this K block
        if (wm == 1) PP_BARRIER();                 // stagger: the second channel half runs one barrier behind
        PP_FENCE();
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                // ---------------- load segment ----------------
                load_b(q, t);
                PP_FENCE();
                load_a(q, g0 + t);
                PP_FENCE();
                if (q == 0) {
                    // second piece of tap t+2
                    if (t + 2 < 9) dma_piece(t + 2, cb, g0 + t + 2, 1);
                    else if (!last_kb) dma_piece(t - 7, cb + 1, g0 + t + 2, 1);
                } else {
                    // retire tap t+1 (pieces of phases (t-2, 1) and (t-1, 0)); the two pieces of tap t+2 stay in flight -- and, in
                    // phase (7, 1), the seven activation loads of phase (6, 1)
                    if (last_kb && t >= 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (NPIECE == 2) {
                        if (t == 7) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    } else {                       // one piece per tap: one newer piece (tap t+2), plus the seven activation loads at t = 7
                        if (t == 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                    }
                    if (t == 6 && !last_kb) load_x(cb + 1);        // consumed by the swap after tap 8
                    PP_FENCE();
                    // first piece of tap t+3 (its slot's previous tenant, tap t-1, was last read in phase (t-1, 1))
                    if (t + 3 < 9) dma_piece(t + 3, cb, g0 + t + 3, 0);
                    else if (!last_kb) dma_piece(t - 6, cb + 1, g0 + t + 3, 0);
                }
                PP_FENCE();
                PP_BARRIER();
                // ---------------- MFMA segment ----------------
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PP_FENCE();
                __builtin_amdgcn_s_setprio(1);
                mfma_tile();
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

    // ---- epilogue ----
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the inline-asm MFMAs are opaque to the hazard recogniser: let the last ones retire
    const int cg = lane >> 4;
    const bool do_relu = p.flags & BD_EPI_RELU;
    const bool add_before = (p.flags & BD_EPI_ADD_BEFORE) && p.add;
    const bool add_after = (p.flags & BD_EPI_ADD_AFTER) && p.add;
    const bool do_mask = (p.flags & BD_EPI_MASK) && p.mask;
    const int cbase = co0 + wm * HALF + 8 * cg;         // + 32 h
    int oy0 = py0[0], ox = px0[0] + colperm(frow), H = pH[0], W = pWd[0], dbase = pdst[0];
#pragma unroll
    for (int q = 1; q < NPATCH; ++q)
        if (wp == q) { oy0 = py0[q]; ox = px0[q] + colperm(frow); H = pH[q]; W = pWd[q]; dbase = pdst[q]; }
#pragma unroll
    for (int h = 0; h < MREP / 2; ++h) {
        if (cbase + 32 * h >= p.CO) continue;
        float bias[8];
        {
            const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(sbias + wm * HALF + 8 * cg + 32 * h);
            const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(sbias + wm * HALF + 8 * cg + 32 * h + 4);
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
}


#undef PP_FENCE
#undef PP_BARRIER
}  // namespace pp128
