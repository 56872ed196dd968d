// fp8 (OCP e4m3) instance of the staggered 256-channel patch kernel (conv3x3_pp.hip): 3x3 / stride-1 / pad-1 FORWARD convolution on
// v_mfma_scale_f32_16x16x128_f8f6f4 (twice the bf16 matrix rate), for BASELINE config 5 ("fp8 weights").
//
// Byte for byte the bf16 kernel's LDS geometry: a K block is 128 one-byte channels = the 128-byte rows that 64 bf16 channels were
// (activation image of four 6x18 input patches at a 144-byte pitch, re-staged through registers once per K block; ring of three
// 32 KB weight-tap slots filled by LDS-DMA two taps ahead; two wave groups one barrier apart).  One MFMA now consumes a whole
// 128-channel row segment -- a lane's fragment is the 32 bytes at k = 32 (lane >> 4) (operand map probed by
// scripts/exp/mfma_fp8_layout.hip) = two 16-byte LDS reads -- so a tap's four phases are the four 32-row quarters of the wave's 128
// output channels (2 A fragments x 4 B fragments x K = 128 = 8 MFMAs of 32 cycles: the bf16 phase's 256 matrix cycles), and the B
// fragments of the tap are read once, in its first phase.  Per phase: the same LDS reads, the same DMA piece, the same waits as the
// bf16 kernel -- for twice the channels.
// Input: the e4m3 copy of the activation (bd_quantize_fp8, or the `y8` twin written by the producing launch of THIS kernel);
// weights: bd_weight_pack_fp8 (one scale per output channel, applied in the epilogue).  Output bf16 (+ optional e4m3 twin).
// MODE 1 is the data gradient (mirrored taps, e5m2 gradient operand, e5m2 twin).
//
// One workgroup per tile.  Round 6 rebuilt this kernel on conv3x3_pp.hip's persistent structure (one workgroup per CU walking its tiles, geometry
// table in LDS, the next tile's image and taps requested inside the last K block) and took it out again: with correct results it measured
// -0.8 % per R101-fp8 step against this form (head tower -2...-6 % per launch, res5 +11 %, the 720-channel score layer +6 %) -- two earlier readings
// of +8 % and +2 % came from builds that stored tiles to the wrong addresses (profiles/r06_pp8_ab.txt tells the whole story, with the per-shape
// table and the epilogue ablation: 32 % of a head-tower launch with a twin is epilogue, half of that the stores).  What stayed from the rebuild is
// the opaque slot-2 fragment base below: 256 -> 245 VGPRs, the last three spills gone, every shape 1-3 % faster, +0.4 % per step.
#include "common.h"

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
constexpr int LDS_BYTES = W_BYTES + X_BYTES + 2 * TILE_CO * 4;   // 162560
constexpr int MAX_SEG = BD_MAX_SEGS;

struct PSeg { int patch_start, H, W, pw, src_off, dst_off; };

struct PParams {
    const unsigned char* src;      // e4m3 [pix][CK]
    const unsigned char* w;        // e4m3 [CO][9][CK]
    const float* wscale;           // [CO]: s_co / act_scale
    unsigned char* dst8;           // optional e4m3 twin of the output (dst * q_scale)
    float q_scale;
    unsigned sr_seed;              // != 0: stochastic rounding of the e5m2 twin (common.h)
    const float* bias;
    const bf16_raw* add;
    const bf16_raw* mask;
    bf16_raw* dst;
    int CK, CO, flags, nseg;
    int src_ppi, dst_ppi;
    unsigned src_bytes;
    int patches_per_img, total_patches, n_tiles;
    PSeg seg[MAX_SEG];
};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
// activation image bank layout: see conv3x3.hip (column permutation + chunk order 0 2 1 3 make the padded rows conflict-free)
__device__ __forceinline__ int colperm(int f) { return f < 4 ? 2 * f : (f < 12 ? 2 * (f - 4) + 1 : 2 * (f - 8)); }
__device__ __forceinline__ int xpos(int chunk) { return (chunk >> 2) * 64 + ((((chunk & 1) << 1) | ((chunk >> 1) & 1)) << 4); }

__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
    c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
    int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (unsigned)v;
}

__device__ __forceinline__ unsigned pack4_bf8(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -57344.f), 57344.f); b = fminf(fmaxf(b, -57344.f), 57344.f);
    c = fminf(fmaxf(c, -57344.f), 57344.f); d = fminf(fmaxf(d, -57344.f), 57344.f);
    int v = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false);
    v = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, v, true);
    return (unsigned)v;
}

#define PP_FENCE() __builtin_amdgcn_sched_barrier(0)
#define PP_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

typedef __attribute__((ext_vector_type(8))) int i32x8_t;

// MODE 0: forward (activations e4m3).  MODE 1: data gradient (mirrored taps; the gradient operand is e5m2 = "bf8", scaled by the
// caller's static gradient scale; weights [Cin][9][Cout] e4m3 with one scale per INPUT channel).
template <int MODE>
__global__ __launch_bounds__(512, 1) void conv3x3_pp8_kernel(const PParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wbuf = smem;                                   // [3][W_SLOT]
    unsigned char* xbuf = smem + W_BYTES;                         // [X_BYTES]
    float* sbias = reinterpret_cast<float*>(smem + W_BYTES + X_BYTES);      // [256]
    float* sscale = sbias + TILE_CO;                                         // [256]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wp = wave & 3;          // channel half (= stagger group), patch
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int pt = bid / p.n_tiles;
    const int ct = bid - pt * p.n_tiles;
    const int co0 = ct * TILE_CO;

    // ---- geometry of the four patches (workgroup-uniform) ----
    int py0[NPATCH], px0[NPATCH], pH[NPATCH], pWd[NPATCH];
    int psrc[NPATCH], pdst[NPATCH];
#pragma unroll
    for (int k = 0; k < NPATCH; ++k) {
        const int pid = pt * NPATCH + k;
        pH[k] = 0; pWd[k] = 0; py0[k] = 0; px0[k] = 0; psrc[k] = 0; pdst[k] = 0;
        if (pid < p.total_patches) {
            const int n = pid / p.patches_per_img;
            const int rem = pid - n * p.patches_per_img;
            int s = 0;
#pragma unroll
            for (int q = 1; q < MAX_SEG; ++q)
                if (q < p.nseg && rem >= p.seg[q].patch_start) s = q;
            const PSeg sg = p.seg[s];
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
            if (y >= 0 && x >= 0 && y < H && x < W) off = (unsigned)((qs + y * W + x) * p.CK + (tid & 7) * 16);
        }
        x_off[k] = off;
    }
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.src), 0, p.src_bytes, 0x00020000);
    u32x4_t rx[XPASSES];
    auto load_x = [&](int cb) {
        int so = cb * 128;
        asm volatile("" : "+s"(so));
        // K tail (CK % 64 != 0, e.g. the 720-channel class-score gradient): chunks past CK read as zeros, so whatever finite weights
        // the DMA picks up beyond a row's CK channels (the next tap's; zeros past the end of the buffer) contribute nothing
        const bool dead = cb * 128 + (tid & 7) * 16 >= p.CK;
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
        dma_src[k] = (unsigned)(co * 9 * p.CK + chunk * 16);
    }
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.w), 0, (unsigned)(p.CO * 9 * p.CK), 0x00020000);
    auto dma_piece = [&](int tap, int cb, int slot, int k) {       // buffer_load_dwordx4 ... offen lds: fixed VGPR offset + scalar offset
        unsigned char* l = wbuf + slot * W_SLOT + (wave + 8 * k) * 1024;
        int so = tap * p.CK + cb * 128;
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
    // a lane's fragment = 16-byte chunks 2 fchunk and 2 fchunk + 1 of its row
    const unsigned char* a_base[2];          // chunk half: row wm*128 + frow of slot 0; + slot * W_SLOT + i * 2048 at compile time
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) a_base[hh] = wbuf + swz(wm * 128 + frow, 2 * fchunk + hh);
    // slot 2 sits past the 16-bit ds_read offset: its base is a register of its own -- an OPAQUE 32-bit LDS address, or the compiler folds it back
    // into a_base + 0x10000 + i * 2048 and keeps one address register per (fragment row, chunk half): 16 VGPRs live through the K loop
    typedef __attribute__((address_space(3))) const unsigned char lds_u8_t;
    typedef __attribute__((address_space(3))) const u32x4_t lds_u32x4_t;
    unsigned a_hi[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        a_hi[hh] = (unsigned)(size_t)(lds_u8_t*)(a_base[hh]) + 2 * W_SLOT;
        asm volatile("" : "+v"(a_hi[hh]));
    }
    const unsigned char* b_base[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) b_base[hh] = xbuf + (wp * (IH * IW) + colperm(frow)) * X_PITCH + xpos(2 * fchunk + hh);

    // one tap = four phases = the four 32-channel quarters of the wave's 128 output channels (A fragments i0, i0 + 1), K = 128 each;
    // the tap's four B fragments (the four patch rows) are read in its first phase
    // PP8_ROWS = 16-row channel fragments per phase: 4 (round 6) = TWO phases of 16 MFMAs per tap, as conv3x3_pp.hip since round 2; 2 = the
    // four phases of 8 of rounds 2-5.  A phase pays its two barriers and its lgkmcnt wait whatever it holds: ~100 cycles beside 256 (8 MFMAs of
    // 32 cycles) or beside 512.  Measured per R101-fp8 step, alternating: +0.2 % for two phases (profiles/r06_pp8_ab.txt, part 1).
    constexpr int PP8_ROWS = 4;
    i32x8_t fa[PP8_ROWS], fb[4];
    auto load_a = [&](int i0, int slot) {
#pragma unroll
        for (int i = 0; i < PP8_ROWS; ++i) {
            u32x4_t l, h;
            if (slot == 2) {
                l = *(lds_u32x4_t*)(size_t)(a_hi[0] + (i0 + i) * 2048);
                h = *(lds_u32x4_t*)(size_t)(a_hi[1] + (i0 + i) * 2048);
            } else {
                l = *reinterpret_cast<const u32x4_t*>(a_base[0] + slot * W_SLOT + (i0 + i) * 2048);
                h = *reinterpret_cast<const u32x4_t*>(a_base[1] + slot * W_SLOT + (i0 + i) * 2048);
            }
            fa[i] = (i32x8_t){(int)l[0], (int)l[1], (int)l[2], (int)l[3], (int)h[0], (int)h[1], (int)h[2], (int)h[3]};
        }
    };
    auto load_b = [&](int t) {
        int dy = t / 3, dx = t - 3 * (t / 3);
        if (MODE == 1) { dy = 2 - dy; dx = 2 - dx; }           // dgrad: mirrored tap
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4_t l = *reinterpret_cast<const u32x4_t*>(b_base[0] + ((j + dy) * IW + dx) * X_PITCH);
            const u32x4_t h = *reinterpret_cast<const u32x4_t*>(b_base[1] + ((j + dy) * IW + dx) * X_PITCH);
            fb[j] = (i32x8_t){(int)l[0], (int)l[1], (int)l[2], (int)l[3], (int)h[0], (int)h[1], (int)h[2], (int)h[3]};
        }
    };
    int one = 0x7f7f7f7f;                        // E8M0 block scales 2^0 (the per-channel weight scale is applied in the epilogue)
    asm volatile("" : "+v"(one));
    auto mfma_quarter = [&](int i0) {
#pragma unroll
        for (int i = 0; i < PP8_ROWS; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                // tied accumulator (D = C) in inline asm, as in the bf16 kernel
                if (MODE == 0)
                    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
                                 : "+v"(acc[i0 + i][j]) : "v"(fa[i]), "v"(fb[j]), "v"(one));
                else                                            // B operand (the gradient) in e5m2
                    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] blgp:1"
                                 : "+v"(acc[i0 + i][j]) : "v"(fa[i]), "v"(fb[j]), "v"(one));
    };

    const int kblocks = (p.CK + 127) >> 7;

    // ---- prologue: activation image of K block 0, taps 0 and 1 ----
    if (tid < TILE_CO) {
        sbias[tid] = (p.bias && co0 + tid < p.CO) ? p.bias[co0 + tid] : 0.f;
        sscale[tid] = co0 + tid < p.CO ? p.wscale[co0 + tid] : 0.f;
    }
    load_x(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) dma_piece(0, 0, 0, k);
#pragma unroll
    for (int k = 0; k < 4; ++k) dma_piece(1, 0, 1, k);
    write_x();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    PP_BARRIER();
    PP_FENCE();

    for (int cb = 0; cb < kblocks; ++cb) {
        const bool last_kb = cb + 1 == kblocks;
        if (wm == 1) PP_BARRIER();                 // stagger: the second channel half runs one barrier behind
        PP_FENCE();
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int q = 0; q < 4; q += PP8_ROWS / 2) {          // (q = the first 32-row quarter of the phase)
                // ---------------- load segment ----------------
                const int slot = t % 3;
                if (q == 0) { load_b(t); PP_FENCE(); }
                load_a(q * 2, slot);
                PP_FENCE();
                if (q == 2) {
                    // retire the pieces of tap t+1 (issued in the four phases up to (t, 0)); the piece of (t, 1) stays in flight
                    if (last_kb && t >= 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                    if (t == 6 && !last_kb) load_x(cb + 1);        // consumed by the swap after tap 8
                    PP_FENCE();
                }
                // weight DMA, one piece per phase: (t, 1..3) and (t+1, 0) fill the slot of tap t+2 -- its previous tenant, tap t-1, was
                // last read in phase (t-1, 3), two barriers before (t, 1)
#pragma unroll
                for (int qq = q; qq < q + PP8_ROWS / 2; ++qq) {          // (the pieces of the quarters this phase covers, in the old order)
                    if (qq == 0) {
                        if (t + 1 < 9) { if (t > 0 || cb > 0) dma_piece(t + 1, cb, (t + 1) % 3, 3); }
                        else if (!last_kb) dma_piece(0, cb + 1, 0, 3);
                    } else {
                        if (t + 2 < 9) dma_piece(t + 2, cb, (t + 2) % 3, qq - 1);
                        else if (!last_kb) dma_piece(t - 7, cb + 1, (t + 2) % 3, qq - 1);
                    }
                }
                PP_FENCE();
                PP_BARRIER();
                // ---------------- MFMA segment ----------------
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PP_FENCE();
                mfma_quarter(q * 2);
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
    const int cbase = co0 + wm * 128 + 8 * cg;          // + 32 h
    int oy0 = py0[0], ox = px0[0] + colperm(frow), H = pH[0], W = pWd[0], dbase = pdst[0];
#pragma unroll
    for (int q = 1; q < NPATCH; ++q)
        if (wp == q) { oy0 = py0[q]; ox = px0[q] + colperm(frow); H = pH[q]; W = pWd[q]; dbase = pdst[q]; }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        if (cbase + 32 * h >= p.CO) continue;
        float bias[8], scl[8];
        {
            const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(sbias + wm * 128 + 8 * cg + 32 * h);
            const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(sbias + wm * 128 + 8 * cg + 32 * h + 4);
            const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(sscale + wm * 128 + 8 * cg + 32 * h);
            const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(sscale + wm * 128 + 8 * cg + 32 * h + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { bias[k] = b0[k]; bias[4 + k] = b1[k]; scl[k] = s0[k]; scl[4 + k] = s1[k]; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oy = oy0 + j;
            if (oy >= H || ox >= W) continue;
            const long long idx = (long long)(dbase + oy * W + ox) * p.CO + cbase + 32 * h;
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = acc[2 * h + (k >> 2)][j][k & 3] * scl[k] + bias[k];
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
            if (p.dst8) {            // one-byte twin for a following fp8 launch (saves its cast pass): e4m3 activations, e5m2 gradients
                u32x2_t o8;
                if (MODE == 0) {
                    o8[0] = pack4_fp8(v[0] * p.q_scale, v[1] * p.q_scale, v[2] * p.q_scale, v[3] * p.q_scale);
                    o8[1] = pack4_fp8(v[4] * p.q_scale, v[5] * p.q_scale, v[6] * p.q_scale, v[7] * p.q_scale);
                } else {
                    if (p.sr_seed) {
                        const unsigned g0 = (unsigned)(idx >> 2);
                        const unsigned r0 = bd_mix32(p.sr_seed ^ g0);
                        o8[0] = bd_pack4_e5m2_sr(v[0] * p.q_scale, v[1] * p.q_scale, v[2] * p.q_scale, v[3] * p.q_scale, r0);
                        o8[1] = bd_pack4_e5m2_sr(v[4] * p.q_scale, v[5] * p.q_scale, v[6] * p.q_scale, v[7] * p.q_scale, r0 * 0x9e3779b1u + 0x7f4a7c15u);
                    } else {
                        o8[0] = pack4_bf8(v[0] * p.q_scale, v[1] * p.q_scale, v[2] * p.q_scale, v[3] * p.q_scale);
                        o8[1] = pack4_bf8(v[4] * p.q_scale, v[5] * p.q_scale, v[6] * p.q_scale, v[7] * p.q_scale);
                    }
                }
                *reinterpret_cast<u32x2_t*>(p.dst8 + idx) = o8;
            }
        }
    }
}

}  // namespace

// 0 = launched, 1 = shape not handled here (the caller falls back to the generic fp8 kernel / the bf16 path)
// mode 0: xq = e4m3 activations [pix][Cin], wq [Cout][9][Cin];  mode 1: xq = e5m2 gradients [pix][Cout], wq [Cin][9][Cout]
int bd_conv3x3_pp8_launch(const bd_conv_desc* d, int mode, const void* xq, const void* wq, const float* wscale, const float* bias,
                          const void* add, const void* mask, void* y, void* y8, float q_scale, int flags, hipStream_t stream) {
    if (!(d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1)) return 1;
    for (int s = 0; s < d->nseg; ++s)
        if (d->Hi[s] != d->Ho[s] || d->Wi[s] != d->Wo[s]) return 1;
    PParams p{};
    p.CK = mode == 0 ? d->Cin : d->Cout;
    p.CO = mode == 0 ? d->Cout : d->Cin;
    p.src_ppi = mode == 0 ? d->in_pix_per_img : d->out_pix_per_img;
    p.dst_ppi = mode == 0 ? d->out_pix_per_img : d->in_pix_per_img;
    if (p.CK % 16 != 0 || p.CO <= 128 || p.CO % 8 != 0) return 1;
    if ((long long)d->N * p.src_ppi * p.CK >= 0x7fffffffll || (long long)d->N * p.dst_ppi >= 0x7fffffffll ||
        (long long)p.CO * 9 * p.CK >= 0x7fffffffll) return 1;
    p.src = (const unsigned char*)xq; p.w = (const unsigned char*)wq; p.wscale = wscale; p.bias = bias;
    p.add = (const bf16_raw*)add; p.mask = (const bf16_raw*)mask; p.dst = (bf16_raw*)y; p.dst8 = (unsigned char*)y8; p.q_scale = q_scale; p.sr_seed = g_fp8_sr_seed;
    p.flags = flags; p.nseg = d->nseg;
    p.src_bytes = (unsigned)((long long)d->N * p.src_ppi * p.CK);
    int ps = 0;
    for (int s = 0; s < d->nseg; ++s) {
        PSeg& sg = p.seg[s];
        sg.patch_start = ps; sg.H = d->Ho[s]; sg.W = d->Wo[s]; sg.pw = cdiv(d->Wo[s], PW);
        sg.src_off = mode == 0 ? d->in_off[s] : d->out_off[s];
        sg.dst_off = mode == 0 ? d->out_off[s] : d->in_off[s];
        ps += cdiv(d->Ho[s], PH) * sg.pw;
    }
    p.patches_per_img = ps;
    p.total_patches = ps * d->N;
    p.n_tiles = cdiv(p.CO, TILE_CO);
    const int grid = cdiv(p.total_patches, NPATCH) * p.n_tiles;
    BD_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pp8_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pp8_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    bd_note_kernel("conv3x3_pp8_kernel");
    if (mode == 0) hipLaunchKernelGGL((conv3x3_pp8_kernel<0>), dim3(grid), dim3(512), LDS_BYTES, stream, p);
    else hipLaunchKernelGGL((conv3x3_pp8_kernel<1>), dim3(grid), dim3(512), LDS_BYTES, stream, p);
    return 0;
}
