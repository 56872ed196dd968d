// fp8 (OCP e4m3 / e5m2) instance of the staggered 256-channel patch kernel: 3x3 / stride-1 / pad-1 forward convolution and data gradient on
// v_mfma_scale_f32_16x16x128_f8f6f4 (twice the bf16 matrix rate), for BASELINE config 5 ("fp8 weights": models/cls/resnet.py:289-293,
// solver/default_solver.py:66-76).
//
// Round 6: this file is conv3x3_pp.hip's ROUND-3 STRUCTURE on one-byte operands -- one PERSISTENT workgroup per CU walking up to 16 tiles,
// the patch geometry of those tiles decoded once into an LDS table, the next tile's activation image and first weight taps requested inside
// the current tile's last K block, the epilogue's residual / gate operands requested eight units at a time -- where rounds 2-5 ran one
// workgroup per tile with a full prologue (two memory round trips in front of the first MFMA, ~9 400 cycles) in front of a K loop that is
// HALF as long per tile as the bf16 kernel's (a K block is 128 one-byte channels): a quarter of a tile's time.
//
// Byte for byte the bf16 kernel's LDS geometry: a K block is 128 one-byte channels = the 128-byte rows that 64 bf16 channels are (activation
// image of four 6x18 input patches at a 144-byte pitch, re-staged through registers once per K block; ring of three 32 KB weight-tap slots
// filled by LDS-DMA two taps ahead; two wave groups one barrier apart).  One MFMA consumes a whole 128-channel row segment -- a lane's
// fragment is the 32 bytes at k = 32 (lane >> 4) (operand map probed by scripts/exp/mfma_fp8_layout.hip) = two 16-byte LDS reads -- so a
// tap is four phases of 8 MFMAs of 32 cycles (the four 32-row quarters of the wave's 128 output channels: 2 A fragments x 4 B fragments x
// K = 128), and the tap's B fragments are read once, in its first phase.
// Input: the e4m3 copy of the activation (bd_quantize_fp8, or the `y8` twin written by the producing launch); MODE 1: the e5m2 gradient.
// Weights: bd_weight_pack_fp8 (one scale per produced channel, applied in the epilogue).  Output bf16 (+ optional one-byte twin).
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
constexpr int GTAB_TILES = 8;                  // patch geometry of this workgroup's (at most) 8 tiles (8 ints per patch), built by wave 0 (16 in the bf16 kernel: here the per-channel scales take 1 KB of its LDS)
constexpr int GTAB_BYTES = GTAB_TILES * NPATCH * 32;          // 1024
constexpr int LDS_BYTES = W_BYTES + X_BYTES + 2 * TILE_CO * 4 + GTAB_BYTES;   // 163584 of 163840
constexpr int MAX_SEG = BD_MAX_SEGS;

struct PSeg { int patch_start, H, W, pw, src_off, dst_off; float inv_pw; };

struct PParams {
    const unsigned char* src;      // e4m3 (MODE 1: e5m2) [pix][CK]
    const unsigned char* w;        // e4m3 [CO][9][CK]
    const float* wscale;           // [CO]: s_co / act_scale
    unsigned char* dst8;           // optional one-byte twin of the output (dst * q_scale): e4m3 forward, e5m2 data gradient
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
    int main_grid;           // the grid: PERSISTENT workgroups, each runs the 256-channel tiles pt = b / n_tiles + k * (main_grid / n_tiles)
    int px_tiles;            // (k = 0, 1, ...; pt < px_tiles) of channel tile b % n_tiles over patches [0, total_patches)
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

#define PP_FENCE() __builtin_amdgcn_sched_barrier(0)
#define PP_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

typedef __attribute__((ext_vector_type(8))) int i32x8_t;

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

// MODE 0: forward (activations e4m3).  MODE 1: data gradient (mirrored taps; the gradient operand is e5m2 = "bf8", scaled by the caller's
// gradient scale; weights [Cin][9][Cout] e4m3 with one scale per INPUT channel).
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
    int* gtab = reinterpret_cast<int*>(smem + W_BYTES + X_BYTES + 2 * TILE_CO * 4);
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
        if ((lane >> 2) < GTAB_TILES) {          // (64 lanes decode 16 tiles; this kernel's table holds 8)
            e[0] = (i32x4_t){vy, vx, vH, vW};
            e[1] = (i32x4_t){vs, vd, 0, 0};
        }
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
                if (y >= 0 && x >= 0 && y < H && x < W) off = (unsigned)((qs + y * W + x) * p.CK + (tq & 7) * 16);
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
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.src), 0, p.src_bytes, 0x00020000);
    u32x4_t rx[XPASSES];
    auto load_x = [&](int cb) {
        int so = cb * 128;
        asm volatile("" : "+s"(so));
        // K tail (CK % 128 != 0, e.g. the 720-channel class-score gradient): chunks past CK read as zeros, so whatever finite weights
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

    // ---- fragment addressing ----
    const int frow = lane & 15, fchunk = lane >> 4;
    // a lane's fragment = the 32 bytes at k = 32 fchunk of its row = 16-byte chunks 2 fchunk and 2 fchunk + 1
    const unsigned char* a_base[2];          // chunk half hh: row wm*128 + frow of slot 0; + slot * W_SLOT + i * 2048 at compile time
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) a_base[hh] = wbuf + swz(wm * 128 + frow, 2 * fchunk + hh);
    const unsigned char* a_hi[2] = {a_base[0] + 2 * W_SLOT, a_base[1] + 2 * W_SLOT};     // slot 2 (ds_read offsets are 16 bit)
    const unsigned char* b_base[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) b_base[hh] = xbuf + (wp * (IH * IW) + colperm(frow)) * X_PITCH + xpos(2 * fchunk + hh);

    // one tap = FOUR phases = the four 32-row quarters of the wave's 128 output channels (A fragments 2 q, 2 q + 1), K = 128 each: 8 MFMAs of
    // 32 cycles; the tap's four B fragments (the four patch rows) are read in its first phase and stay: 8 + 4 x 4 ds_read_b128 per tap, 48
    // fragment registers -- the bf16 kernel's budget.  (Two phases of 16 -- four A fragments, 64 registers -- put the staged activations
    // into scratch inside the K loop; on the one-tile-per-workgroup form of rounds 2-5, which had the registers, two phases measured
    // +0.2 %: 508.4 / 506.5 / 510.5 against 507.3 / 507.5 / 507.8 img/s on R101-fp8, alternating.)
    i32x8_t fa[2], fb[4];
    auto load_a = [&](int q, int slot) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned char* lo = slot == 2 ? a_hi[0] : a_base[0] + slot * W_SLOT;
            const unsigned char* hi = slot == 2 ? a_hi[1] : a_base[1] + slot * W_SLOT;
            const u32x4_t l = *reinterpret_cast<const u32x4_t*>(lo + (2 * q + i) * 2048), h = *reinterpret_cast<const u32x4_t*>(hi + (2 * q + i) * 2048);
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
    auto mfma_quarter = [&](int q) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                // tied accumulator (D = C) in inline asm, as in the bf16 kernel
                if (MODE == 0)
                    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
                                 : "+v"(acc[2 * q + i][j]) : "v"(fa[i]), "v"(fb[j]), "v"(one));
                else                                            // B operand (the gradient) in e5m2
                    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] blgp:1"
                                 : "+v"(acc[2 * q + i][j]) : "v"(fa[i]), "v"(fb[j]), "v"(one));
    };

    const int kblocks = (p.CK + 127) >> 7;
    // ---- prologue: activation image of K block 0, taps 0, 1 and 2 (the whole ring) ----
    if (tid < TILE_CO) {
        sbias[tid] = (p.bias && co0 + tid < p.CO) ? p.bias[co0 + tid] : 0.f;
        sscale[tid] = co0 + tid < p.CO ? p.wscale[co0 + tid] : 0.f;
    }
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
        if (last_kb && more) decode(kt + 1, n_oy0, n_px0, n_H, n_W, n_dst);
        if (wm == 1) PP_BARRIER();                 // stagger: the second channel half runs one barrier behind
        PP_FENCE();
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // ---------------- load segment ----------------
                const int slot = t % 3;
                if (q == 0) { load_b(t); PP_FENCE(); }
                load_a(q, slot);
                PP_FENCE();
                if (q == 2) {
                    // retire the pieces of tap t+1 (the last of them was issued in phase (t, 0)); piece 0 of tap t+2 stays in flight
                    if (!cont && t >= 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
                // weight DMA, one piece per phase, in the bf16 kernel's order: (t, 0) the last piece of tap t+1, (t, 1) the first of tap t+2,
                // (t, 2) and (t, 3) its pieces 1 and 2.  The slot of tap t+2 was last read (tap t-1) in phase (t-1, 3), two barriers before (t, 1)
                if (q == 0) {
                    if (t + 1 < 9) { if (t > 1 || cb > 0) dma_piece(t + 1, cb, (t + 1) % 3, 3); }
                    else if (cont) dma_piece(0, nb, 0, 3);
                } else {
                    if (t + 2 < 9) { if (t > 0 || cb > 0) dma_piece(t + 2, cb, (t + 2) % 3, q - 1); }
                    else if (cont) dma_piece(t - 7, nb, (t + 2) % 3, q - 1);
                    if (q == 3 && t == 8 && last_kb && more) dma_piece(1, 0, 1, 3);      // the next tile's tap 1 complete: see the head of the tile loop
                }
                PP_FENCE();
                PP_BARRIER();
                // ---------------- MFMA segment ----------------
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PP_FENCE();
                mfma_quarter(q);
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
    // ---- epilogue ----
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the inline-asm MFMAs are opaque to the hazard recogniser: let the last ones retire
    int lq = lane;                            // (opaque copy: the epilogue's per-lane constants are not kept across the MFMA loop)
    asm volatile("" : "+v"(lq));
    const int cg = lq >> 4, erow = lq & 15;
    const bool do_relu = p.flags & BD_EPI_RELU;
    const bool add_before = (p.flags & BD_EPI_ADD_BEFORE) && p.add;
    const bool add_after = (p.flags & BD_EPI_ADD_AFTER) && p.add;
    const bool do_mask = (p.flags & BD_EPI_MASK) && p.mask;
    const int cbase = co0 + wm * 128 + 8 * cg;          // + 32 h
    const int oy0 = c_oy0, ox = c_px0 + colperm(erow), H = c_H, W = c_W, dbase = c_dst;
    // The operands of TWO channel groups (eight 16-byte units: up to 64 VGPRs of residuals + gates) are requested together, then consumed and
    // stored -- two round trips per tile instead of one per unit (rounds 2-5: load -> wait -> use -> store for each of the 16 units, each
    // wait also covering the previous unit's store).  The next tile's tap 2 (its slot was tap 8's) is requested first.
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
        // before the first store: everything requested so far -- these operands, the next tile's image and taps -- has landed (loads and
        // stores share one vmcnt and retire in order: conv3x3_pp.hip)
        if (hp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int h = 2 * hp + hh;
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
                const long long idx = erow_off[j] + 32 * h;
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = acc[2 * h + (k >> 2)][j][k & 3] * scl[k] + bias[k];
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
    // ---- next tile: its image and first two taps are in LDS (written above / by DMA), its first loads were retired in the K loop ----
    if (!more) break;
    ++kt;
    pt = pt_next;
    c_oy0 = n_oy0; c_px0 = n_px0; c_H = n_H; c_W = n_W; c_dst = n_dst;
    PP_FENCE();
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
    const int px_tiles = cdiv(p.total_patches, NPATCH);
    p.px_tiles = px_tiles;
    // persistent workgroups: one per CU (a multiple of the channel tiles, so that a workgroup keeps its channel tile, its weight-DMA addresses
    // and its scale / bias vectors), fewer if the tiles run out; at most GTAB_TILES tiles per workgroup (the geometry table): larger launches
    // get a whole multiple of that grid
    p.main_grid = px_tiles * p.n_tiles;
    {
        const int num_cus = bd_num_cus();
        const int g1 = (num_cus / p.n_tiles) * p.n_tiles;
        const int rounds = cdiv(px_tiles * p.n_tiles, g1 * GTAB_TILES);
        if (g1 * rounds < p.main_grid) p.main_grid = g1 * rounds;
    }
    BD_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pp8_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pp8_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    bd_note_kernel("conv3x3_pp8_kernel");
    if (mode == 0) hipLaunchKernelGGL((conv3x3_pp8_kernel<0>), dim3(p.main_grid), dim3(512), LDS_BYTES, stream, p);
    else hipLaunchKernelGGL((conv3x3_pp8_kernel<1>), dim3(p.main_grid), dim3(512), LDS_BYTES, stream, p);
    return 0;
}
