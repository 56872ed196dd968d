// Implicit-GEMM convolution forward and data-gradient on CDNA4 MFMA (gfx950).
//
//   D[co][pix] = sum_{tap, ck}  W[co][tap][ck] * SRC[srcpix(pix, tap)][ck]
//
// MFMA A operand = packed weights (rows = output channels), B operand = NHWC activations
// (columns = pixels), both K-contiguous in memory, so fragments are plain 16-byte LDS reads.
// With the output channel on the MFMA row, every lane ends up holding 4 consecutive channels of one
// pixel -> 8-byte NHWC stores, and the per-channel bias is a per-register constant.
//
// Workgroup = 256 threads = 4 waves (2 channel halves x 2 pixel halves); tile = 128 channels x 128 pixels,
// K step = BK channels of one filter tap.  Global->register->LDS staging (loads of step t+1 are issued
// before the MFMAs of step t, written after them), two LDS buffers, one barrier per K step.
// LDS rows are padded (BK*2 + 32 bytes) so that ds_read_b128 fragment reads are bank-conflict free.
//
// mode 0 (forward):  pix = output pixel, src = conv input,  sy = oy*stride - pad + r
// mode 1 (dgrad):    pix = input pixel,  src = dY,          sy = (iy + pad - r)/stride when divisible.
//   For stride 2 the pixels are enumerated parity-class-major so that a tile only visits the taps
//   that can contribute to its class (no multiply-by-zero work).
#include "igemm_params.h"

using namespace igemm;

namespace {

template <int BK>
struct Cfg {
    static constexpr int CHUNKS = BK / 8;             // 16-byte chunks per row
    static constexpr int ROWS_PER_PASS = 256 / CHUNKS;
    static constexpr int PASSES = 128 / ROWS_PER_PASS;
    // BK = 64: 128-byte rows padded to 160 B (conflict-free ds_read_b128, measured SQ_LDS_BANK_CONFLICT = 0).
    // BK = 32: unpadded 64-byte rows with the chunk XOR-swizzled by (row >> 1) & 3: conflict-free reads AND
    //          stores (the 96-byte padded pitch showed 33 % conflict cycles), and only 32 KB per workgroup.
    static constexpr int LDS_STRIDE = BK == 64 ? 160 : 64;
    static constexpr int TILE_BYTES = 128 * LDS_STRIDE;
    __device__ static __forceinline__ int off(int row, int chunk) {
        if constexpr (BK == 64) return row * LDS_STRIDE + chunk * 16;
        else return row * 64 + ((chunk ^ ((row >> 1) & 3)) << 4);
    }
};

// PRE: the epilogue operands (residual / accumulated gradient `add`, ReLU `mask`) of a linear-destination tile are fetched at the
// START of the kernel into registers, so that for the short-K 1x1 layers (HBM-bound, 2..8 K steps) their HBM latency overlaps the
// operand loads and the MFMAs instead of forming a second serial round trip after the last MFMA.
// BUF: operands staged with range-checked buffer loads -- a 32-bit per-thread byte offset per staged row, recomputed only when the
// filter tap changes, plus a scalar K-block offset; rows outside the image / past the tile read as zeros through an offset beyond
// the buffer.  The 64-bit pointer arithmetic and the predicated loads of the general path are ~110 VALU instructions per K step and
// wave against 16 MFMAs: they, not the matrix pipe or the memory system, bounded the K >= 256 layers (tensors >= 2 GB keep that path).
template <int BK, bool PRE, bool BUF>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const IgemmParams p) {
    using C = Cfg<BK>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: [buf][A tile | B tile]; the stride-2 dgrad tap mask borrows the first word before staging starts
    unsigned char* tiles = smem;
    unsigned int* s_tapmask = reinterpret_cast<unsigned int*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wc = wave >> 1;   // channel half
    const int wp = wave & 1;    // pixel half

    // XCD-aware bijective remap: consecutive tile ids stay on one XCD (shared weights / halo rows in its L2)
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    if (!(p.mode == 1 && p.stride > 1)) {
        // (the stride-2 data-gradient enumerates its pixels parity-class-major: the classes differ in work -- for a 1x1 filter only
        // one class has any -- so there the tiles stay round-robin over the XCDs; the contiguous remap put all the work on two)
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / p.n_tiles;
    const int tile_n = bid - tile_m * p.n_tiles;
    const int m0 = tile_m * TILE_P;
    const int co0 = tile_n * TILE_C;

    const int chunk = tid % C::CHUNKS;
    const int row0 = tid / C::CHUNKS;

    u32x4_t pre_add[PRE ? 8 : 1], pre_mask[PRE ? 8 : 1];
    if (PRE) {
        const int cb_ = co0 + wc * 64 + 8 * (lane >> 4);
        const bool want_add = (p.flags & (BD_EPI_ADD_BEFORE | BD_EPI_ADD_AFTER)) && p.add;
        const bool want_mask = (p.flags & BD_EPI_MASK) && p.mask;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int m = m0 + wp * 64 + j * 16 + (lane & 15);
                const bool ok = m < p.M && cb_ + 32 * half < p.CO;
                const long long idx = (long long)m * p.CO + cb_ + 32 * half;
                u32x4_t a = {0u, 0u, 0u, 0u}, k = {0u, 0u, 0u, 0u};
                if (ok && want_add) a = *reinterpret_cast<const u32x4_t*>(p.add + idx);
                if (ok && want_mask) k = *reinterpret_cast<const u32x4_t*>(p.mask + idx);
                pre_add[j * 2 + half] = a; pre_mask[j * 2 + half] = k;
            }
    }

    // ---- per-row pixel decode (B operand rows handled by this thread) ----
    int b_py[C::PASSES], b_px[C::PASSES], b_hs[C::PASSES], b_ws[C::PASSES];
    long long b_base[C::PASSES];   // src pixel base of the image (+ level offset), -1 if row is out of range
    unsigned int my_tapmask = 0;
    const int RS = p.R * p.S;
    if (tid == 0) *s_tapmask = 0;
#pragma unroll
    for (int i = 0; i < C::PASSES; ++i) {
        const int row = row0 + i * C::ROWS_PER_PASS;
        const int m = m0 + row;
        b_base[i] = -1; b_py[i] = 0; b_px[i] = 0; b_hs[i] = 0; b_ws[i] = 0;
        if (p.linear_src) {
            // 1x1 / stride 1 on a dense level: the source pixel IS column m; encode it as row 0, column m of a 1 x M image
            if (m < p.M) { b_base[i] = 0; b_px[i] = m; b_hs[i] = 1; b_ws[i] = p.M; }
        } else if (m < p.M) {
            int s = 0;
#pragma unroll
            for (int k = 1; k < MAX_SUB; ++k)
                if (k < p.nsub && m >= p.sub[k].m_start) s = k;
            const SubSeg ss = p.sub[s];
            const int local = m - ss.m_start;
            int n, rem, yy, xx;
            fast_divmod(local, ss.Hs * ss.Ws, ss.inv_per_img, n, rem);
            fast_divmod(rem, ss.Ws, ss.inv_ws, yy, xx);
            const int py = ss.y0 + ss.step * yy;
            const int px = ss.x0 + ss.step * xx;
            b_base[i] = (long long)n * p.src_pix_per_img + ss.src_off;
            b_hs[i] = ss.Hsrc; b_ws[i] = ss.Wsrc;
            if (p.mode == 0) { b_py[i] = py * p.stride - p.pad; b_px[i] = px * p.stride - p.pad; }
            else             { b_py[i] = py + p.pad;            b_px[i] = px + p.pad; }
            if (p.mode == 1 && p.stride > 1) {
                for (int t = 0; t < RS; ++t) {
                    const int r = t / p.S, sx = t - r * p.S;
                    const int ty = b_py[i] - r, tx = b_px[i] - sx;
                    const bool ok = ty >= 0 && tx >= 0 && ((ty | tx) & (p.stride - 1)) == 0 &&
                                    (ty >> 1) < ss.Hsrc && (tx >> 1) < ss.Wsrc;
                    if (ok) my_tapmask |= 1u << t;
                }
            }
        }
    }
    unsigned int tapmask;
    if (p.mode == 1 && p.stride > 1) {
        __syncthreads();
        if (my_tapmask) atomicOr(s_tapmask, my_tapmask);
        __syncthreads();
        tapmask = *s_tapmask;
        __syncthreads();
    } else {
        tapmask = (RS >= 32) ? 0xffffffffu : ((1u << RS) - 1u);
    }
    const int kblocks = (p.CK + BK - 1) / BK;
    const int nsteps = __popc(tapmask) * kblocks;

    // ---- staging state ----
    u32x4_t ra[C::PASSES], rb[C::PASSES];
    unsigned int rem_mask = tapmask;
    int cur_tap = -1, cur_kb = kblocks;   // forces a tap advance on first call
    int tap_r = 0, tap_s = 0;

    constexpr unsigned X_NONE = 0x80000000u;          // >= num_records of either buffer: the load returns zeros
    unsigned a_voff[BUF ? C::PASSES : 1], b_voff[BUF ? C::PASSES : 1];
    __amdgpu_buffer_rsrc_t x_rsrc, w_rsrc;
    if (BUF) {
        x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.src), 0, p.src_bytes, 0x00020000);
        w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.w), 0, p.w_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < C::PASSES; ++i) {
            const int lrow = row0 + i * C::ROWS_PER_PASS;
            const int rho = lrow & 15;
            const int co = co0 + (lrow & 64) + 32 * ((lrow >> 5) & 1) + 8 * (rho >> 2) + 4 * ((lrow >> 4) & 1) + (rho & 3);
            a_voff[i] = co < p.CO ? (unsigned)(co * RS * p.CK + chunk * 8) * 2u : X_NONE;
        }
    }

    auto stage_load = [&]() {
        if (BUF) {
            if (cur_kb == kblocks) {          // next filter tap (workgroup-uniform): the source pixel of every staged row moves
                cur_kb = 0;
                cur_tap = __ffs(rem_mask) - 1;
                rem_mask &= rem_mask - 1;
                tap_r = cur_tap / p.S;
                tap_s = cur_tap - tap_r * p.S;
#pragma unroll
                for (int i = 0; i < C::PASSES; ++i) {
                    int sy, sx;
                    bool ok = b_base[i] >= 0;
                    if (p.mode == 0) {
                        sy = b_py[i] + tap_r; sx = b_px[i] + tap_s;
                        ok = ok && sy >= 0 && sx >= 0 && sy < b_hs[i] && sx < b_ws[i];
                    } else {
                        const int ty = b_py[i] - tap_r, tx = b_px[i] - tap_s;
                        ok = ok && ty >= 0 && tx >= 0;
                        if (p.stride == 2) { ok = ok && (((ty | tx) & 1) == 0); sy = ty >> 1; sx = tx >> 1; }
                        else { sy = ty; sx = tx; }
                        ok = ok && sy < b_hs[i] && sx < b_ws[i];
                    }
                    b_voff[i] = ok ? (unsigned)(((int)b_base[i] + sy * b_ws[i] + sx) * p.CK + chunk * 8) * 2u : X_NONE;
                }
            }
            int so_a = (cur_tap * p.CK + cur_kb * BK) * 2, so_b = cur_kb * BK * 2;
            asm volatile("" : "+s"(so_a), "+s"(so_b));           // keep the K-block offsets in the scalar operand
            const bool dead = cur_kb * BK + chunk * 8 >= p.CK;    // channel tail (CK % 8 == 0): zero-fill
#pragma unroll
            for (int i = 0; i < C::PASSES; ++i) {
                ra[i] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, dead ? X_NONE : a_voff[i], so_a, 0);
                rb[i] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, dead ? X_NONE : b_voff[i], so_b, 0);
            }
            ++cur_kb;
            return;
        }
        if (cur_kb == kblocks) {
            cur_kb = 0;
            cur_tap = __ffs(rem_mask) - 1;
            rem_mask &= rem_mask - 1;
            tap_r = cur_tap / p.S;
            tap_s = cur_tap - tap_r * p.S;
        }
        const int c0 = cur_kb * BK + chunk * 8;
        const bool cvalid = c0 + 8 <= p.CK;   // zero-fill the channel tail (CK % 8 == 0)
        // A: weights [co][tap][ck]
#pragma unroll
        for (int i = 0; i < C::PASSES; ++i) {
            // LDS row (h*64 + t*16 + rho) holds output channel h*64 + 32*(t>>1) + 8*(rho>>2) + 4*(t&1) + (rho&3): after
            // the MFMAs lane group cg = rho>>2 owns channels 8*cg..8*cg+7 of each 32-channel half, so one epilogue
            // instruction moves 16 B per lane = 64 contiguous bytes per pixel (4 lanes), two instructions per 128-B line
            const int lrow = row0 + i * C::ROWS_PER_PASS;
            const int rho = lrow & 15;
            const int co = co0 + (lrow & 64) + 32 * ((lrow >> 5) & 1) + 8 * (rho >> 2) + 4 * ((lrow >> 4) & 1) + (rho & 3);
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (cvalid && co < p.CO) {
                const bf16_raw* g = p.w + ((long long)co * RS + cur_tap) * p.CK + c0;
                v = *reinterpret_cast<const u32x4_t*>(g);
            }
            ra[i] = v;
        }
        // B: activations
#pragma unroll
        for (int i = 0; i < C::PASSES; ++i) {
            u32x4_t v = {0u, 0u, 0u, 0u};
            int sy, sx;
            bool ok = cvalid && b_base[i] >= 0;
            if (p.mode == 0) {
                sy = b_py[i] + tap_r; sx = b_px[i] + tap_s;
                ok = ok && sy >= 0 && sx >= 0 && sy < b_hs[i] && sx < b_ws[i];
            } else {
                const int ty = b_py[i] - tap_r, tx = b_px[i] - tap_s;
                ok = ok && ty >= 0 && tx >= 0;
                if (p.stride == 2) { ok = ok && (((ty | tx) & 1) == 0); sy = ty >> 1; sx = tx >> 1; }
                else { sy = ty; sx = tx; }
                ok = ok && sy < b_hs[i] && sx < b_ws[i];
            }
            if (ok) {
                const bf16_raw* g = p.src + (b_base[i] + (long long)sy * b_ws[i] + sx) * p.CK + c0;
                v = *reinterpret_cast<const u32x4_t*>(g);
            }
            rb[i] = v;
        }
        ++cur_kb;
    };
    auto stage_write = [&](int buf) {
        unsigned char* At = tiles + buf * 2 * C::TILE_BYTES;
        unsigned char* Bt = At + C::TILE_BYTES;
#pragma unroll
        for (int i = 0; i < C::PASSES; ++i) {
            const int row = row0 + i * C::ROWS_PER_PASS;
            *reinterpret_cast<u32x4_t*>(At + C::off(row, chunk)) = ra[i];
            *reinterpret_cast<u32x4_t*>(Bt + C::off(row, chunk)) = rb[i];
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int frag_row = lane & 15;
    const int frag_chunk = lane >> 4;

    auto compute = [&](int buf) {
        const unsigned char* At = tiles + buf * 2 * C::TILE_BYTES;
        const unsigned char* Bt = At + C::TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8_t a[4], b[4];
            const int ch = kk * 4 + frag_chunk;
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(At + C::off(wc * 64 + i * 16 + frag_row, ch));
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(Bt + C::off(wp * 64 + j * 16 + frag_row, ch));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };

    if (nsteps > 0) {
        stage_load();
        stage_write(0);
    }
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < nsteps; ++t) {
        const bool more = (t + 1 < nsteps);
        if (more) stage_load();
        compute(cur);
        if (more) stage_write(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: lane (cg = lane>>4) holds channels co0 + wc*64 + 16*cg + (4*i + r) of pixel (lane & 15) ----
    const int cg = lane >> 4;
    const bool do_relu = p.flags & BD_EPI_RELU;
    const bool add_before = (p.flags & BD_EPI_ADD_BEFORE) && p.add;
    const bool add_after = (p.flags & BD_EPI_ADD_AFTER) && p.add;
    const bool do_mask = (p.flags & BD_EPI_MASK) && p.mask;
    const int cbase = co0 + wc * 64 + 8 * cg;      // + 32 * half below
    float bias[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) bias[k] = 0.f;
    if (p.bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (cbase + 32 * (q >> 1) + 4 * (q & 1) < p.CO) {
                const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(p.bias + cbase + 32 * (q >> 1) + 4 * (q & 1));
                bias[4 * q] = bv[0]; bias[4 * q + 1] = bv[1]; bias[4 * q + 2] = bv[2]; bias[4 * q + 3] = bv[3];
            }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wp * 64 + j * 16 + (lane & 15);
        if (m >= p.M) continue;
        int dstpix;
        if (p.linear_dst) dstpix = m;
        else {
            int s = 0;
#pragma unroll
            for (int k = 1; k < MAX_SUB; ++k)
                if (k < p.nsub && m >= p.sub[k].m_start) s = k;
            const SubSeg ss = p.sub[s];
            const int local = m - ss.m_start;
            int n, rem, yy, xx;
            fast_divmod(local, ss.Hs * ss.Ws, ss.inv_per_img, n, rem);
            fast_divmod(rem, ss.Ws, ss.inv_ws, yy, xx);
            dstpix = n * p.dst_pix_per_img + ss.dst_off + (ss.y0 + ss.step * yy) * ss.Wd + ss.x0 + ss.step * xx;
        }
        const long long base = (long long)dstpix * p.CO + cbase;
#pragma unroll
        for (int half = 0; half < 2; ++half) {       // 8 channels = 16 bytes per half
            if (cbase + 32 * half >= p.CO) continue;  // CO % 8 == 0
            const long long idx = base + 32 * half;
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = acc[2 * half + (k >> 2)][j][k & 3] + bias[8 * half + k];
            if (add_before) {
                const u32x4_t av = PRE ? pre_add[j * 2 + half] : *reinterpret_cast<const u32x4_t*>(p.add + idx);
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[2 * k] += bf_lo(av[k]); v[2 * k + 1] += bf_hi(av[k]); }
            }
            if (do_relu) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
            }
            if (do_mask) {
                const u32x4_t mv = PRE ? pre_mask[j * 2 + half] : *reinterpret_cast<const u32x4_t*>(p.mask + idx);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (!(bf_lo(mv[k]) > 0.f)) v[2 * k] = 0.f;
                    if (!(bf_hi(mv[k]) > 0.f)) v[2 * k + 1] = 0.f;
                }
            }
            if (add_after) {
                const u32x4_t av = PRE ? pre_add[j * 2 + half] : *reinterpret_cast<const u32x4_t*>(p.add + idx);
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

BD_KNOB int g_igemm_prefetch_epi = 1;      // bd_conv_desc.route[1] bit 5 clears it
BD_KNOB int g_igemm_buf = 1;               // bd_conv_desc.route[1] bit 11 clears it: 64-bit pointer staging instead of buffer loads

template <int BK>
int launch_igemm(const IgemmParams& p, hipStream_t stream) {
    using C = Cfg<BK>;
    const size_t lds = 4 * C::TILE_BYTES;
    BD_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BK, false, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BK, true, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BK, false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BK, true, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = p.m_tiles * p.n_tiles;
    bd_note_kernel(BK == 32 ? "conv_igemm_kernel<32>" : "conv_igemm_kernel<64>");
    const bool epi_ops = ((p.flags & (BD_EPI_ADD_BEFORE | BD_EPI_ADD_AFTER)) && p.add) || ((p.flags & BD_EPI_MASK) && p.mask);
    const int ksteps = (p.CK + BK - 1) / BK * p.R * p.S;
    const bool pre = g_igemm_prefetch_epi && p.linear_dst && epi_ops && ksteps <= 4;     // (<= 8 / 16 / 64 measured: no difference)
    const bool buf = g_igemm_buf && p.src_bytes != 0 && p.w_bytes != 0;
    if (pre && buf) hipLaunchKernelGGL((conv_igemm_kernel<BK, true, true>), dim3(grid), dim3(256), lds, stream, p);
    else if (pre) hipLaunchKernelGGL((conv_igemm_kernel<BK, true, false>), dim3(grid), dim3(256), lds, stream, p);
    else if (buf) hipLaunchKernelGGL((conv_igemm_kernel<BK, false, true>), dim3(grid), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((conv_igemm_kernel<BK, false, false>), dim3(grid), dim3(256), lds, stream, p);
    return 0;
}

int check_desc(const bd_conv_desc* d) {
    BD_REQUIRE(d != nullptr, "conv: null descriptor");
    BD_REQUIRE(d->nseg >= 1 && d->nseg <= BD_MAX_SEGS, "conv: nseg=%d out of range", d->nseg);
    BD_REQUIRE(d->N >= 1 && d->Cin >= 1 && d->Cout >= 1, "conv: bad N/Cin/Cout");
    BD_REQUIRE(d->stride == 1 || d->stride == 2, "conv: stride %d unsupported (1 or 2)", d->stride);
    BD_REQUIRE(d->R * d->S <= 32, "conv: %dx%d kernel unsupported by the generic path", d->R, d->S);
    for (int s = 0; s < d->nseg; ++s) {
        BD_REQUIRE((d->Hi[s] + 2 * d->pad - d->R) / d->stride + 1 == d->Ho[s] &&
                   (d->Wi[s] + 2 * d->pad - d->S) / d->stride + 1 == d->Wo[s],
                   "conv: level %d output size inconsistent with input/stride/pad", s);
    }
    return 0;
}

// buffer-load staging needs 32-bit byte offsets: both operands < 2 GB (else the kernel keeps its 64-bit pointer path)
void set_buffer_sizes(IgemmParams& p, int N) {
    const long long sb = (long long)N * p.src_pix_per_img * p.CK * 2, wb = (long long)p.CO * p.R * p.S * p.CK * 2;
    p.src_bytes = (sb > 0 && sb < 0x7fffffffll) ? (unsigned)sb : 0u;
    p.w_bytes = (wb > 0 && wb < 0x7fffffffll) ? (unsigned)wb : 0u;
}

bool is_3x3s1(const bd_conv_desc* d) {
    if (!(d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1)) return false;
    for (int s = 0; s < d->nseg; ++s)
        if (d->Hi[s] != d->Ho[s] || d->Wi[s] != d->Wo[s]) return false;
    return true;
}
BD_KNOB int g_use_patch3x3 = 1;
#ifdef BD_AB_SKIP                // diagnostic build only (BD_EXTRA_FLAGS=-DBD_AB_SKIP next to BD_LIB_NAME): the shipped library cannot skip a launch
BD_KNOB int g_skip_s2_3x3 = 0;          // bd_conv_desc.route[1] bit 4: timing A/B only -- the stride-2 3x3 forward / data-gradient launches return at once
#endif
BD_KNOB int g_bk32_for_1x1 = 1;
// Stride-2 3x3 layers stay on the generic kernel, and there the tiles are short (the data gradient visits 1 / 2 / 2 / 4 taps per
// parity class) and bound by the latency of their few K steps: BK=32 tiles (32 KB of LDS instead of 80 KB: four workgroups per CU
// instead of two) run them 10-35 % faster as long as the grid still fills the chip (P6's 70-tile forward keeps BK=64).
// bd_conv_desc.route[1] bit 10 clears it.
BD_KNOB int g_bk32_s2 = 1;
// (Round 5 built conv_igemm_wide.hip -- a 128-channel x 256-pixel tile, 128 x 64 wave tiles, both operands by LDS-DMA -- for the stride-2 3x3
// layers and measured it no faster than this file's kernel; round 6 closed the family and deleted it: profiles/r06_s2_phase_major.txt.)

}  // namespace


int bd_conv3x3_patch_launch(const bd_conv_desc* d, int mode, const void* src, const void* w, const float* bias,
                            const void* add, const void* mask, void* dst, int flags, hipStream_t stream);

// debug/measurement knob: bit 0 clear forces the generic per-tap kernel for 3x3 stride-1 convolutions; bit 1: BK=32 tiles for 1x1
extern BD_KNOB int g_patch_dma;
extern BD_KNOB int g_patch_pp;
extern BD_KNOB int g_patch_pp128;
extern BD_KNOB int g_conv1x1_s2;
extern BD_KNOB int g_pp_tail_split;
extern BD_KNOB int g_pp_persistent;
// bd_conv_desc.route[1] - 1: the bit mask documented in include/basedet_hip.h
static int bd_route_patch3x3(int enable) {
#ifdef BD_AB_SKIP
    g_skip_s2_3x3 = (enable >> 4) & 1;
#else
    if ((enable >> 4) & 1) {         // (checked before anything is changed)
        bd_set_error("bd_conv_desc.route[1]: bit 4 (skip the 3x3 / stride-2 launches: a timing A/B that leaves stale outputs) exists in "
                     "-DBD_AB_SKIP diagnostic builds only");
        return BD_EINVAL;
    }
#endif
    g_use_patch3x3 = enable & 1; g_bk32_for_1x1 = (enable >> 1) & 1;
    g_patch_dma = ((enable >> 3) & 1) ^ 1;
    g_patch_pp = ((enable >> 6) & 1) ? 0 : (((enable >> 7) & 1) ? 1 : 2);
    g_patch_pp128 = ((enable >> 9) & 1) ? -1 : ((enable >> 8) & 1);
    g_bk32_s2 = ((enable >> 10) & 1) ^ 1;
    g_igemm_prefetch_epi = ((enable >> 5) & 1) ^ 1;
    g_igemm_buf = ((enable >> 11) & 1) ^ 1;
    g_conv1x1_s2 = ((enable >> 12) & 1) ^ 1;
    g_pp_tail_split = ((enable >> 13) & 1) ^ 1;
    g_pp_persistent = ((enable >> 14) & 1) ^ 1;
    if ((enable >> 16) & 1) {
        bd_set_error("bd_conv_desc.route[1]: bit 16 (conv_igemm_wide.hip) is gone since round 6");
        return BD_EINVAL;
    }
    return BD_OK;
}

int bd_route_dense1x1(int depth);          // conv1x1.hip
int bd_route_wgrad(int use_tr);            // conv_wgrad.hip
extern BD_KNOB int g_fp8_patch;            // conv_fp8.hip

// Every entry point that takes a descriptor: the routing variables are written from THAT descriptor, defaults where a word is 0
BdRouteScope::BdRouteScope(const bd_conv_desc* d) : rc(BD_OK) {
    const int r0 = d ? d->route[0] : 0, r1 = d ? d->route[1] : 0, r2 = d ? d->route[2] : 0, r3 = d ? d->route[3] : 0;
    if (r0 < 0 || r1 < 0 || r2 < 0 || r3 < 0) {
        bd_set_error("bd_conv_desc.route: negative word (0 = the library's choice, otherwise the route + 1)");
        rc = BD_EINVAL;
        return;
    }
    if ((rc = bd_route_dense1x1(r0 ? r0 - 1 : 1)) != BD_OK) return;
    if ((rc = bd_route_patch3x3(r1 ? r1 - 1 : 3)) != BD_OK) return;
    if ((rc = bd_route_wgrad(r2 ? r2 - 1 : 1)) != BD_OK) return;
    g_fp8_patch = r3 ? (r3 - 1 != 0) : 1;
    g_fp8_sr_seed = d ? d->sr_seed : 0u;
}

int bd_conv1x1_dense_launch(const void* x, const void* w, const float* bias, const void* add, const void* mask, const unsigned* maskbits,
                            void* y, unsigned* ybits, void* y8, float q_scale, int y8_bf8, long long M, int CK, int CO, int flags,
                            hipStream_t stream);

int bd_conv1x1_s2_launch(const bd_conv_desc* d, int mode, const void* src, const void* w, const float* bias, const void* add, const void* mask,
                         const unsigned* maskbits, void* dst, int flags, hipStream_t stream);
namespace {
// 1x1 / stride 1 / pad 0 over one dense level: source pixel index == destination pixel index (conv1x1.hip)
bool is_dense_1x1(const bd_conv_desc* d) {
    return d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0 && d->nseg == 1 && d->in_off[0] == 0 && d->out_off[0] == 0 &&
           d->in_pix_per_img == d->Hi[0] * d->Wi[0] && d->out_pix_per_img == d->Ho[0] * d->Wo[0];
}
}  // namespace

static int conv2d_fwd_impl(const bd_conv_desc* d, const void* x, const void* w_packed, const float* bias, const void* add, void* y,
                           unsigned* ybits, void* y8, float q_scale, int flags, bd_stream_t stream) {
    if (int e = check_desc(d)) return e;
    BD_REQUIRE(x && w_packed && y, "conv2d_fwd: null pointer");
    BD_REQUIRE(d->Cin % 8 == 0, "conv2d_fwd: Cin=%d must be a multiple of 8", d->Cin);
    BD_REQUIRE(d->Cout % 8 == 0, "conv2d_fwd: Cout=%d must be a multiple of 8", d->Cout);
    BD_REQUIRE(!(flags & BD_EPI_MASK), "conv2d_fwd: BD_EPI_MASK is a dgrad-only flag");
    if (is_dense_1x1(d) &&
        bd_conv1x1_dense_launch(x, w_packed, bias, add, nullptr, nullptr, y, ybits, y8, q_scale, 0, (long long)d->N * d->out_pix_per_img,
                                d->Cin, d->Cout, flags, (hipStream_t)stream) == 0) {
        BD_CHECK_LAUNCH("bd_conv2d_fwd(dense 1x1)");
        return BD_OK;
    }
    if (!ybits && !y8 && d->stride == 2 && d->R == 1 &&
        bd_conv1x1_s2_launch(d, 0, x, w_packed, bias, add, nullptr, nullptr, y, flags, (hipStream_t)stream) == 0) {
        BD_CHECK_LAUNCH("bd_conv2d_fwd(1x1 stride 2)");
        return BD_OK;
    }
    BD_REQUIRE(ybits == nullptr && y8 == nullptr, "conv2d_fwd_bits / _ex: the bit-packed ReLU mask and the e4m3 twin are written by the dense "
               "1x1 kernel only (1x1 / stride 1 over one dense level, Cout %% 32 == 0, tensors < 2 GB)");
#ifdef BD_AB_SKIP
    if (g_skip_s2_3x3 && d->R == 3 && d->S == 3 && d->stride == 2) return BD_OK;       // (A/B: "how much of the step are these launches?")
#endif
    if (g_use_patch3x3 && is_3x3s1(d)) {
        bd_conv3x3_patch_launch(d, 0, x, w_packed, bias, add, nullptr, y, flags, (hipStream_t)stream);
        BD_CHECK_LAUNCH("bd_conv2d_fwd(3x3 patch)");
        return BD_OK;
    }
    IgemmParams p{};
    p.src = (const bf16_raw*)x; p.w = (const bf16_raw*)w_packed; p.bias = bias;
    p.add = (const bf16_raw*)add; p.mask = nullptr; p.dst = (bf16_raw*)y;
    p.CK = d->Cin; p.CO = d->Cout; p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad = d->pad;
    p.mode = 0; p.flags = flags;
    p.nsub = d->nseg;
    long long m = 0;
    for (int s = 0; s < d->nseg; ++s) {
        SubSeg& ss = p.sub[s];
        ss.m_start = (int)m;
        ss.Hs = d->Ho[s]; ss.Ws = d->Wo[s]; ss.y0 = 0; ss.x0 = 0; ss.step = 1;
        ss.Wd = d->Wo[s]; ss.dst_off = d->out_off[s];
        ss.Hsrc = d->Hi[s]; ss.Wsrc = d->Wi[s]; ss.src_off = d->in_off[s];
        ss.inv_per_img = 1.0f / (float)(ss.Hs * ss.Ws); ss.inv_ws = 1.0f / (float)ss.Ws;
        m += (long long)d->N * d->Ho[s] * d->Wo[s];
    }
    BD_REQUIRE(m < (1ll << 24), "conv2d_fwd: too many pixels (2^24 limit of the fast index decode)");
    p.linear_dst = (d->nseg == 1 && d->out_off[0] == 0 && d->out_pix_per_img == d->Ho[0] * d->Wo[0]) ? 1 : 0;
    p.linear_src = (p.linear_dst && d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0 && d->in_off[0] == 0 &&
                    d->in_pix_per_img == d->Hi[0] * d->Wi[0]) ? 1 : 0;
    p.M = (int)m;
    p.src_pix_per_img = d->in_pix_per_img; p.dst_pix_per_img = d->out_pix_per_img;
    set_buffer_sizes(p, d->N);
    p.m_tiles = cdiv(p.M, TILE_P); p.n_tiles = cdiv(p.CO, TILE_C);
    if (p.CK > 32 && !(g_bk32_for_1x1 && p.R * p.S == 1) && !(g_bk32_s2 && p.stride == 2 && p.m_tiles * p.n_tiles >= 512))
        launch_igemm<64>(p, (hipStream_t)stream);
    else launch_igemm<32>(p, (hipStream_t)stream);
    BD_CHECK_LAUNCH("bd_conv2d_fwd");
    return BD_OK;
}

extern "C" int bd_conv2d_fwd(const bd_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                             const void* add, void* y, int flags, bd_stream_t stream) {
    BD_ROUTE(d);
    return conv2d_fwd_impl(d, x, w_packed, bias, add, y, nullptr, nullptr, 1.f, flags, stream);
}

extern "C" int bd_conv2d_fwd_ex(const bd_conv_desc* d, const void* x, const void* w_packed, const float* bias, const void* add, void* y,
                                uint32_t* ybits, void* y8, float q_scale, int flags, bd_stream_t stream) {
    BD_ROUTE(d);
    return conv2d_fwd_impl(d, x, w_packed, bias, add, y, ybits, y8, q_scale, flags, stream);
}

extern "C" int bd_conv2d_fwd_bits(const bd_conv_desc* d, const void* x, const void* w_packed, const float* bias, const void* add, void* y,
                                  uint32_t* ybits, int flags, bd_stream_t stream) {
    BD_ROUTE(d);
    BD_REQUIRE(ybits != nullptr, "conv2d_fwd_bits: null ybits");
    return conv2d_fwd_impl(d, x, w_packed, bias, add, y, ybits, nullptr, 1.f, flags, stream);
}

static int conv2d_dgrad_impl(const bd_conv_desc* d, const void* g, const void* w_packed_t, const void* add, const void* mask,
                             const unsigned* maskbits, void* dx, void* dx8, float q_scale, int flags, bd_stream_t stream) {
    if (int e = check_desc(d)) return e;
    BD_REQUIRE(g && w_packed_t && dx, "conv2d_dgrad: null pointer");
    if (d->stride == 1) flags &= ~BD_EPI_SPARSE;          // every pixel of a stride-1 data gradient is reached
    BD_REQUIRE(d->Cout % 8 == 0, "conv2d_dgrad: Cout=%d must be a multiple of 8 (pad the gradient)", d->Cout);
    BD_REQUIRE(d->Cin % 8 == 0, "conv2d_dgrad: Cin=%d must be a multiple of 8", d->Cin);
    BD_REQUIRE(!(flags & BD_EPI_RELU), "conv2d_dgrad: BD_EPI_RELU is a forward-only flag");
    if (is_dense_1x1(d) &&
        bd_conv1x1_dense_launch(g, w_packed_t, nullptr, add, mask, maskbits, dx, nullptr, dx8, q_scale, 1, (long long)d->N * d->in_pix_per_img,
                                d->Cout, d->Cin, flags, (hipStream_t)stream) == 0) {
        BD_CHECK_LAUNCH("bd_conv2d_dgrad(dense 1x1)");
        return BD_OK;
    }
    if (!dx8 && d->stride == 2 && d->R == 1 &&
        bd_conv1x1_s2_launch(d, 1, g, w_packed_t, nullptr, add, mask, maskbits, dx, flags, (hipStream_t)stream) == 0) {
        BD_CHECK_LAUNCH("bd_conv2d_dgrad(1x1 stride 2)");
        return BD_OK;
    }
    BD_REQUIRE(maskbits == nullptr && dx8 == nullptr, "conv2d_dgrad_bits / _ex: the bit-packed ReLU mask is read, and the e5m2 twin written, by "
               "the dense 1x1 kernel only (1x1 / stride 1 over one dense level, Cin %% 32 == 0, tensors < 2 GB)");
#ifdef BD_AB_SKIP
    if (g_skip_s2_3x3 && d->R == 3 && d->S == 3 && d->stride == 2) return BD_OK;
#endif
    if (g_use_patch3x3 && is_3x3s1(d)) {
        bd_conv3x3_patch_launch(d, 1, g, w_packed_t, nullptr, add, mask, dx, flags, (hipStream_t)stream);
        BD_CHECK_LAUNCH("bd_conv2d_dgrad(3x3 patch)");
        return BD_OK;
    }
    IgemmParams p{};
    p.src = (const bf16_raw*)g; p.w = (const bf16_raw*)w_packed_t; p.bias = nullptr;
    p.add = (const bf16_raw*)add; p.mask = (const bf16_raw*)mask; p.dst = (bf16_raw*)dx;
    p.CK = d->Cout; p.CO = d->Cin; p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad = d->pad;
    p.mode = 1; p.flags = flags;
    long long m = 0;
    int ns = 0;
    const int st = d->stride;
    const bool sparse = (flags & BD_EPI_SPARSE) && st > 1;
    BD_REQUIRE(!(flags & BD_EPI_SPARSE) || ((flags & BD_EPI_ADD_BEFORE) && add == dx),
               "conv2d_dgrad: BD_EPI_SPARSE needs an in-place accumulate (BD_EPI_ADD_BEFORE with add == dx)");
    p.flags = flags & ~BD_EPI_SPARSE;
    for (int s = 0; s < d->nseg; ++s) {
        for (int qy = 0; qy < st; ++qy)
            for (int qx = 0; qx < st; ++qx) {
                const int Hs = (d->Hi[s] - qy + st - 1) / st, Ws = (d->Wi[s] - qx + st - 1) / st;
                if (Hs <= 0 || Ws <= 0) continue;
                // BD_EPI_SPARSE: a parity class that no tap reaches (row parity: some r with (qy + pad - r) % st == 0, same for columns)
                // keeps what dx holds: the 1x1 / stride-2 shortcut then touches a quarter of the pixels
                if (sparse) {
                    bool ry = false, rx = false;
                    for (int r = 0; r < d->R; ++r) ry = ry || ((qy + d->pad - r) % st + st) % st == 0;
                    for (int c = 0; c < d->S; ++c) rx = rx || ((qx + d->pad - c) % st + st) % st == 0;
                    if (!(ry && rx)) continue;
                }
                BD_REQUIRE(ns < MAX_SUB, "conv2d_dgrad: too many sub-segments (levels x stride^2 > %d)", MAX_SUB);
                SubSeg& ss = p.sub[ns++];
                ss.m_start = (int)m;
                ss.Hs = Hs; ss.Ws = Ws; ss.y0 = qy; ss.x0 = qx; ss.step = st;
                ss.Wd = d->Wi[s]; ss.dst_off = d->in_off[s];
                ss.Hsrc = d->Ho[s]; ss.Wsrc = d->Wo[s]; ss.src_off = d->out_off[s];
                ss.inv_per_img = 1.0f / (float)(Hs * Ws); ss.inv_ws = 1.0f / (float)Ws;
                m += (long long)d->N * Hs * Ws;
            }
    }
    p.nsub = ns;
    BD_REQUIRE(m < (1ll << 24), "conv2d_dgrad: too many pixels (2^24 limit of the fast index decode)");
    p.linear_dst = (d->nseg == 1 && st == 1 && d->in_off[0] == 0 && d->in_pix_per_img == d->Hi[0] * d->Wi[0]) ? 1 : 0;
    p.linear_src = (p.linear_dst && d->R == 1 && d->S == 1 && d->pad == 0 && d->out_off[0] == 0 &&
                    d->out_pix_per_img == d->Ho[0] * d->Wo[0]) ? 1 : 0;
    p.M = (int)m;
    p.src_pix_per_img = d->out_pix_per_img; p.dst_pix_per_img = d->in_pix_per_img;
    set_buffer_sizes(p, d->N);
    p.m_tiles = cdiv(p.M, TILE_P); p.n_tiles = cdiv(p.CO, TILE_C);
    if (p.CK > 32 && !(g_bk32_for_1x1 && p.R * p.S == 1) && !(g_bk32_s2 && p.stride == 2 && p.m_tiles * p.n_tiles >= 512))
        launch_igemm<64>(p, (hipStream_t)stream);
    else launch_igemm<32>(p, (hipStream_t)stream);
    BD_CHECK_LAUNCH("bd_conv2d_dgrad");
    return BD_OK;
}

extern "C" int bd_conv2d_dgrad(const bd_conv_desc* d, const void* g, const void* w_packed_t, const void* add,
                               const void* mask, void* dx, int flags, bd_stream_t stream) {
    BD_ROUTE(d);
    return conv2d_dgrad_impl(d, g, w_packed_t, add, mask, nullptr, dx, nullptr, 1.f, flags, stream);
}

extern "C" int bd_conv2d_dgrad_ex(const bd_conv_desc* d, const void* g, const void* w_packed_t, const void* add, const void* mask,
                                  const uint32_t* maskbits, void* dx, void* dx8, float q_scale, int flags, bd_stream_t stream) {
    BD_ROUTE(d);
    return conv2d_dgrad_impl(d, g, w_packed_t, add, maskbits ? nullptr : mask, maskbits, dx, dx8, q_scale,
                             maskbits ? (flags | BD_EPI_MASK) : flags, stream);
}

extern "C" int bd_conv2d_dgrad_bits(const bd_conv_desc* d, const void* g, const void* w_packed_t, const void* add, const uint32_t* maskbits,
                                    void* dx, int flags, bd_stream_t stream) {
    BD_ROUTE(d);
    BD_REQUIRE(maskbits != nullptr, "conv2d_dgrad_bits: null maskbits");
    return conv2d_dgrad_impl(d, g, w_packed_t, add, nullptr, maskbits, dx, nullptr, 1.f, flags | BD_EPI_MASK, stream);
}
