// Implicit-GEMM convolution forward / data gradient, WIDE wave tile -- a round-5 EXPERIMENT for the stride-2 3x3 family (res3.0 / res4.0 /
// res5.0 conv2), opt-in (bd_conv_desc.route[1] bit 16), NOT on the default path: on those layers it runs exactly as fast as conv_igemm.hip's
// kernel, with either K order (forward 137 / 138 / 142 us against 140 / 126 / 123; data gradient 222 / 173 / 156 against 214 / 169 / 145:
// profiles/r05_s2_micro.txt), and 20-35 % slower than conv1x1.hip on the dense 1x1 launches (profiles/r05_dense1x1_wide.txt).  What both
// kernels share is what bounds them: ~96 KB of gathered operand bytes in flight per CU (LDS holds no more beside the tiles) over a loaded
// fabric round trip of ~2.5 us, i.e. ~40 GB/s per CU, and a gather that fetches every input pixel 2.25 times (the nine taps of a stride-2
// filter reach each pixel one, two, two or four times) with re-use distances of an L2's size.  Fewer LDS reads per MFMA (this tile) and
// shorter re-use distances (this K order) move neither; staging each input pixel ONCE per K block (phase-major patches in LDS, as
// conv3x3_pp.hip does for stride 1) would, and does not fit LDS beside a weight ring at a 256-pixel tile.
//
// Same GEMM and parameter block as conv_igemm.hip (32 channels per MFMA, fp32 accumulation; the K ORDER differs -- see `request` -- so the
// results agree to fp32 summation order, not bit for bit), re-cut for what bounded that kernel on these layers (rounds 3-4: 0.197 /
// 0.049 of their roofs, HBM traffic 2.2-2.9 x the algorithmic bytes):
//   * wave tile 128 channels x 64 pixels (acc[8][4]) instead of 64 x 64: 12 ds_read_b128 per 32 MFMAs instead of 8 per 16 -- a 128 x 128
//     workgroup tile of 64 x 64 wave tiles reads ~1 000 LDS cycles of fragments per 512 matrix cycles and is LDS-bound whatever feeds it
//     (round 4's gather-on-ring attempt ran exactly as fast as the generic kernel for that reason);
//   * workgroup tile 128 channels x 256 pixels, four waves side by side along the pixels: every gathered pixel row is fetched once per 128
//     output channels, and a tile is 1.5 - 5 whole output rows (neighbouring tiles share their halo rows in the XCD's L2);
//   * both operands by LDS-DMA (buffer_load ... lds) into a three-stage ring of 24 KB stages (two K steps in flight per workgroup, two
//     workgroups per CU): the per-lane source offset of a pixel row moves only when the filter tap changes; the K block rides in the
//     scalar offset; rows outside the image read zeros through an out-of-range offset (no predication, no staging registers).
// Stride-2 data gradients enumerate their pixels parity-class-major exactly as the generic kernel does (a tile visits only the taps that
// reach its class: 1 / 2 / 2 / 4 of the nine).
#include "igemm_params.h"

using namespace igemm;

namespace {

constexpr int WT_C = 128, WT_P = 256, WBK = 32;
constexpr int W_A_BYTES = WT_C * 64;            // weight tile of one stage: 128 rows x 64 B
constexpr int W_B_BYTES = WT_P * 64;            // pixel tile: 256 rows x 64 B
constexpr int W_STAGE = W_A_BYTES + W_B_BYTES;  // 24 576
constexpr int W_NSTAGE = 3;
constexpr int W_LDS = W_NSTAGE * W_STAGE;       // 73 728: two workgroups per CU
constexpr int W_NDMA = 6;                       // DMA instructions per wave and K step: 2 weight pieces + 4 pixel pieces

typedef __attribute__((address_space(3))) void lds_void_w_t;

__device__ __forceinline__ int w_lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 3)) << 4); }

__global__ __launch_bounds__(256, 2) void conv_igemm_wide_kernel(const IgemmParams p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned int* s_tapmask = reinterpret_cast<unsigned int*>(smem);       // borrowed before the ring starts
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    if (!(p.mode == 1 && p.stride > 1)) {          // (as conv_igemm.hip: the parity classes of a strided data gradient differ in work)
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / p.n_tiles;
    const int tile_n = bid - tile_m * p.n_tiles;
    const int m0 = tile_m * WT_P;
    const int co0 = tile_n * WT_C;
    const int RS = p.R * p.S;

    constexpr unsigned X_NONE = 0x80000000u;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.src), 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.w), 0, p.w_bytes, 0x00020000);

    // ---- DMA pieces: a piece = 16 LDS rows x 64 B = one wave-instruction; lane -> row lane >> 2, position lane & 3, source chunk =
    // position ^ ((row >> 1) & 3) (the read-side swizzle on the SOURCE address).  Weight tile: 8 pieces, this wave owns wave and wave + 4;
    // pixel tile: 16 pieces, this wave owns wave + 4 k.
    unsigned a_voff[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int lrow = 16 * (wave + 4 * k) + (lane >> 2);
        const int ch = (lane & 3) ^ ((lrow >> 1) & 3);
        const int rho = lrow & 15;
        // LDS row (h*64 + t*16 + rho) holds output channel h*64 + 32*(t>>1) + 8*(rho>>2) + 4*(t&1) + (rho&3) (conv_igemm.hip's permutation)
        const int co = co0 + (lrow & 64) + 32 * ((lrow >> 5) & 1) + 8 * (rho >> 2) + 4 * ((lrow >> 4) & 1) + (rho & 3);
        a_voff[k] = co < p.CO ? (unsigned)(co * RS * p.CK + ch * 8) * 2u : X_NONE;
    }
    int b_py[4], b_px[4], b_hs[4], b_ws[4], b_base[4], b_ch[4];
    unsigned int my_tapmask = 0;
    const bool class_taps = p.mode == 1 && p.stride > 1;          // (workgroup-uniform)
    if (class_taps && tid == 0) *s_tapmask = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int lrow = 16 * (wave + 4 * k) + (lane >> 2);
        b_ch[k] = ((lane & 3) ^ ((lrow >> 1) & 3)) * 8;
        const int m = m0 + lrow;
        b_base[k] = -1; b_py[k] = 0; b_px[k] = 0; b_hs[k] = 0; b_ws[k] = 0;
        if (m < p.M) {
            const int s = find_sub(p, m);
            const SubSeg ss = p.sub[s];
            const int local = m - ss.m_start;
            int n, rem, yy, xx;
            fast_divmod(local, ss.Hs * ss.Ws, ss.inv_per_img, n, rem);
            fast_divmod(rem, ss.Ws, ss.inv_ws, yy, xx);
            const int py = ss.y0 + ss.step * yy, px = ss.x0 + ss.step * xx;
            b_base[k] = n * p.src_pix_per_img + ss.src_off;
            b_hs[k] = ss.Hsrc; b_ws[k] = ss.Wsrc;
            if (p.mode == 0) { b_py[k] = py * p.stride - p.pad; b_px[k] = px * p.stride - p.pad; }
            else             { b_py[k] = py + p.pad;            b_px[k] = px + p.pad; }
            if (class_taps) {
                for (int t = 0; t < RS; ++t) {
                    const int r = t / p.S, sx = t - r * p.S;
                    const int ty = b_py[k] - r, tx = b_px[k] - sx;
                    const bool ok = ty >= 0 && tx >= 0 && ((ty | tx) & (p.stride - 1)) == 0 && (ty >> 1) < ss.Hsrc && (tx >> 1) < ss.Wsrc;
                    if (ok) my_tapmask |= 1u << t;
                }
            }
        }
    }
    unsigned int tapmask;
    if (class_taps) {
        __syncthreads();
        if (my_tapmask) atomicOr(s_tapmask, my_tapmask);
        __syncthreads();
        tapmask = *s_tapmask;
        __syncthreads();
    } else {
        tapmask = (RS >= 32) ? 0xffffffffu : ((1u << RS) - 1u);
    }
    tapmask = __builtin_amdgcn_readfirstlane(tapmask);
    const int kblocks = p.CK / WBK;                 // the host takes CK % 32 == 0 only
    const int nsteps = __popc(tapmask) * kblocks;

    // ---- producer state: the (K-block pair, tap, half) of the NEXT step to request.  K ORDER (what this kernel is about, measured round 5:
    // with the generic kernel's order -- tap outer, K block inner -- this kernel ran exactly as fast as the generic one, 141 vs 141 us on
    // res3.0's forward, at 2.2 x the algorithmic HBM traffic): a gathered input pixel is wanted again by a NEIGHBOURING tap (stride 2: tap
    // column 2 of output x and tap column 0 of output x + 1 are the same pixel), and the two 64-byte K blocks of a pixel share a 128-byte
    // line.  Tap-outer order puts 2 x kblocks steps x 16 KB per workgroup between those uses -- 8 MB per XCD with 64 workgroups in flight,
    // twice its L2 -- so every re-use went back to the fabric.  Here: pairs of K blocks outermost, the taps inside, the pair's two halves
    // innermost: a line's second half is the next step, a neighbouring tap's re-use two to four steps away.
    unsigned int rem_mask = tapmask;
    int i_tap = 0, i_g = 0, i_h = 0, i_hmax = 0;    // i_hmax = 0 forces a tap advance on the first request
    unsigned b_voff[4];
    auto request = [&](int stage) {
        if (i_h >= i_hmax) {                        // next filter tap (workgroup-uniform): the source pixel of every staged row moves
            i_h = 0;
            if (rem_mask == 0) { rem_mask = tapmask; ++i_g; }          // every tap of this K-block pair done: next pair
            i_hmax = kblocks - 2 * i_g < 2 ? kblocks - 2 * i_g : 2;
            i_tap = __ffs(rem_mask) - 1;
            rem_mask &= rem_mask - 1;
            const int tap_r = i_tap / p.S, tap_s = i_tap - tap_r * p.S;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int sy, sx;
                bool ok = b_base[k] >= 0;
                if (p.mode == 0) {
                    sy = b_py[k] + tap_r; sx = b_px[k] + tap_s;
                    ok = ok && sy >= 0 && sx >= 0 && sy < b_hs[k] && sx < b_ws[k];
                } else {
                    const int ty = b_py[k] - tap_r, tx = b_px[k] - tap_s;
                    ok = ok && ty >= 0 && tx >= 0;
                    if (p.stride == 2) { ok = ok && (((ty | tx) & 1) == 0); sy = ty >> 1; sx = tx >> 1; }
                    else { sy = ty; sx = tx; }
                    ok = ok && sy < b_hs[k] && sx < b_ws[k];
                }
                b_voff[k] = ok ? (unsigned)((b_base[k] + sy * b_ws[k] + sx) * p.CK + b_ch[k]) * 2u : X_NONE;
            }
        }
        const int i_kb = 2 * i_g + i_h;
        int so_a = (i_tap * p.CK + i_kb * WBK) * 2, so_b = i_kb * WBK * 2;
        asm volatile("" : "+s"(so_a), "+s"(so_b));            // keep the K-block offsets in the scalar operand
        unsigned char* At = smem + stage * W_STAGE;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void_w_t*)(At + (wave + 4 * k) * 1024), 16, a_voff[k], so_a, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void_w_t*)(At + W_A_BYTES + (wave + 4 * k) * 1024), 16, b_voff[k], so_b, 0, 0);
        ++i_h;
    };

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fchunk = lane >> 4;
    auto compute = [&](int stage) {
        const unsigned char* At = smem + stage * W_STAGE;
        const unsigned char* Bt = At + W_A_BYTES;
        bf16x8_t a[8], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(Bt + w_lds_off(wave * 64 + j * 16 + frow, fchunk));
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(At + w_lds_off(i * 16 + frow, fchunk));
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };

    // ---- ring: stages t and t + 1 in flight (W_NDMA instructions per wave each); stage t has landed when at most W_NDMA are outstanding
    if (nsteps > 0) request(0);
    if (nsteps > 1) request(1);
    int cs = 0, ps = 2;
    for (int t = 0; t < nsteps; ++t) {
        if (t + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W_NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");   // everyone's pieces; stage ps is free
        if (t + 2 < nsteps) request(ps);
        compute(cs);
        cs = cs == 2 ? 0 : cs + 1;
        ps = ps == 2 ? 0 : ps + 1;
    }

    // ---- epilogue (conv_igemm.hip's): lane group cg = lane >> 4 holds channels cbase + 32 q + 0..7 (q = 0..3) of pixel m0 + wave*64 + j*16 + (lane & 15)
    const int cg = lane >> 4;
    const bool do_relu = p.flags & BD_EPI_RELU;
    const bool add_before = (p.flags & BD_EPI_ADD_BEFORE) && p.add;
    const bool add_after = (p.flags & BD_EPI_ADD_AFTER) && p.add;
    const bool do_mask = (p.flags & BD_EPI_MASK) && p.mask;
    const int cbase = co0 + 8 * cg;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wave * 64 + j * 16 + (lane & 15);
        if (m >= p.M) continue;
        int dstpix;
        if (p.linear_dst) dstpix = m;
        else {
            const SubSeg ss = p.sub[find_sub(p, m)];
            const int local = m - ss.m_start;
            int n, rem, yy, xx;
            fast_divmod(local, ss.Hs * ss.Ws, ss.inv_per_img, n, rem);
            fast_divmod(rem, ss.Ws, ss.inv_ws, yy, xx);
            dstpix = n * p.dst_pix_per_img + ss.dst_off + (ss.y0 + ss.step * yy) * ss.Wd + ss.x0 + ss.step * xx;
        }
        const long long base = (long long)dstpix * p.CO + cbase;
        // operands of the pixel's four 16-byte units first, then compute and store
        u32x4_t av[4], mv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            av[q] = (u32x4_t){0u, 0u, 0u, 0u}; mv[q] = (u32x4_t){0u, 0u, 0u, 0u};
            if (cbase + 32 * q < p.CO) {
                if (add_before || add_after) av[q] = *reinterpret_cast<const u32x4_t*>(p.add + base + 32 * q);
                if (do_mask) mv[q] = *reinterpret_cast<const u32x4_t*>(p.mask + base + 32 * q);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {                 // 32-channel group q: accumulator rows 2 q, 2 q + 1
            if (cbase + 32 * q >= p.CO) continue;     // CO % 8 == 0
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = acc[2 * q + (k >> 2)][j][k & 3];
            if (p.bias) {
                const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(p.bias + cbase + 32 * q);
                const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(p.bias + cbase + 32 * q + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[k] += b0[k]; v[4 + k] += b1[k]; }
            }
            if (add_before) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[2 * k] += bf_lo(av[q][k]); v[2 * k + 1] += bf_hi(av[q][k]); }
            }
            if (do_relu) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
            }
            if (do_mask) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (!(bf_lo(mv[q][k]) > 0.f)) v[2 * k] = 0.f;
                    if (!(bf_hi(mv[q][k]) > 0.f)) v[2 * k + 1] = 0.f;
                }
            }
            if (add_after) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[2 * k] += bf_lo(av[q][k]); v[2 * k + 1] += bf_hi(av[q][k]); }
            }
            u32x4_t o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = pack_bf2(v[2 * k], v[2 * k + 1]);
            *reinterpret_cast<u32x4_t*>(p.dst + base + 32 * q) = o;
        }
    }
}

}  // namespace

// 0 = launched; 1 = not this kernel's shape (the caller keeps the generic kernel).  `everywhere`: every launch the kernel CAN take (tests / A-B),
// else the stride-2 3x3 launches whose grid fills the chip.
int bd_conv_igemm_wide_launch(IgemmParams p, bool everywhere, hipStream_t stream) {
    if (p.CK % WBK != 0 || p.CO % 8 != 0 || p.src_bytes == 0 || p.w_bytes == 0 || p.R * p.S > 32) return 1;
    if (!everywhere) return 1;
    p.m_tiles = cdiv(p.M, WT_P); p.n_tiles = cdiv(p.CO, WT_C);
    const int grid = p.m_tiles * p.n_tiles;
    BD_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS));
    bd_note_kernel("conv_igemm_wide_kernel");
    hipLaunchKernelGGL(conv_igemm_wide_kernel, dim3(grid), dim3(256), W_LDS, stream, p);
    return 0;
}
