// 1x1 convolution forward / dgrad for the HBM-bound bottleneck and FPN-lateral layers: persistent workgroups with a
// four-deep register prefetch ring that runs ACROSS output tiles.
//
// These layers have a short K (64..2048 channels, one tap): with one-tile workgroups the global-load latency of the
// first K step and the read-modify-write epilogue (residual / ReLU-mask) are fully exposed and the kernels reach only
// ~2 TB/s.  Here a workgroup walks a list of 128x128 output tiles; its loads form one continuous stream of K steps
// (32 channels each) that is always four steps ahead of the MFMAs, so the next tile's operands are in flight while
// the current tile's epilogue reads and writes HBM.  Two LDS buffers of 2 x 128 x 96 B, one barrier per step.
#include "igemm_params.h"

using namespace igemm;

namespace {

constexpr int BK = 32;
constexpr int DEPTH = 4;                 // register stages in flight
constexpr int LDS_STRIDE = 96;           // 64 B data + 32 B pad: conflict-free ds_read_b128
constexpr int TILE_BYTES = 128 * LDS_STRIDE;
constexpr int PASSES = 2;                // 128 rows / (256 threads / 4 chunks)

__global__ __launch_bounds__(256, 2) void conv1x1_stream_kernel(const IgemmParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char tiles[4 * TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave >> 1, wp = wave & 1;
    const int G = gridDim.x;
    const int total_tiles = p.m_tiles * p.n_tiles;
    const int my_tiles = (total_tiles - (int)blockIdx.x + G - 1) / G;
    const int kblocks = (p.CK + BK - 1) / BK;
    const int total_steps = my_tiles * kblocks;

    const int chunk = tid & 3;
    const int row0 = tid >> 2;

    // ---- load stream state ----
    int l_tile = (int)blockIdx.x - G, l_kb = kblocks;    // forces a tile advance on the first issue
    int l_co0 = 0;
    long long b_src[PASSES];     // source pixel element offset (pixel * CK) or -1
    int a_co[PASSES];
    // range-checked buffer loads (the host takes this kernel only when both operands are < 2 GB): 32-bit byte offsets per staged row,
    // recomputed once per tile; the K step travels in the scalar offset; rows outside the problem read zeros
    constexpr unsigned X_NONE = 0x80000000u;
    unsigned a_voff[PASSES], b_voff[PASSES];
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.src), 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.w), 0, p.w_bytes, 0x00020000);
    u32x4_t ra[DEPTH][PASSES], rb[DEPTH][PASSES];
    int ls = 0;

    auto decode_rows = [&]() {
        const int tile_m = l_tile / p.n_tiles;
        l_co0 = (l_tile - tile_m * p.n_tiles) * TILE_C;
        const int m0 = tile_m * TILE_P;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int lrow = row0 + i * 64;
            const int rho = lrow & 15;
            a_co[i] = l_co0 + (lrow & 64) + 32 * ((lrow >> 5) & 1) + 8 * (rho >> 2) + 4 * ((lrow >> 4) & 1) + (rho & 3);
            const int m = m0 + lrow;
            b_src[i] = -1;
            if (p.linear_src) {
                if (m < p.M) b_src[i] = (long long)m * p.CK;
            } else if (m < p.M) {
                const SubSeg ss = p.sub[find_sub(p, m)];
                const int local = m - ss.m_start;
                int n, rem, yy, xx;
                fast_divmod(local, ss.Hs * ss.Ws, ss.inv_per_img, n, rem);
                fast_divmod(rem, ss.Ws, ss.inv_ws, yy, xx);
                const int py = ss.y0 + ss.step * yy, px = ss.x0 + ss.step * xx;
                int sy, sx;
                bool ok = true;
                if (p.mode == 0) { sy = py * p.stride; sx = px * p.stride; }
                else if (p.stride == 2) { ok = ((py | px) & 1) == 0; sy = py >> 1; sx = px >> 1; }
                else { sy = py; sx = px; }
                ok = ok && sy < ss.Hsrc && sx < ss.Wsrc;
                if (ok) b_src[i] = ((long long)n * p.src_pix_per_img + ss.src_off + (long long)sy * ss.Wsrc + sx) * p.CK;
            }
            a_voff[i] = a_co[i] < p.CO ? (unsigned)(a_co[i] * p.CK + chunk * 8) * 2u : X_NONE;
            b_voff[i] = b_src[i] >= 0 ? (unsigned)((int)b_src[i] + chunk * 8) * 2u : X_NONE;
        }
    };
    auto issue_load = [&](u32x4_t (&xa)[PASSES], u32x4_t (&xb)[PASSES]) {
        if (l_kb == kblocks) { l_kb = 0; l_tile += G; decode_rows(); }
        const bool dead = l_kb * BK + chunk * 8 >= p.CK;      // channel tail (CK % 8 == 0): zero-fill
        int so = l_kb * BK * 2;
        asm volatile("" : "+s"(so));
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            xa[i] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, dead ? X_NONE : a_voff[i], so, 0);
            xb[i] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, dead ? X_NONE : b_voff[i], so, 0);
        }
        ++l_kb; ++ls;
    };
    auto write_lds = [&](int buf, const u32x4_t (&xa)[PASSES], const u32x4_t (&xb)[PASSES]) {
        unsigned char* At = tiles + buf * 2 * TILE_BYTES;
        unsigned char* Bt = At + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int row = row0 + i * 64;
            *reinterpret_cast<u32x4_t*>(At + row * LDS_STRIDE + chunk * 16) = xa[i];
            *reinterpret_cast<u32x4_t*>(Bt + row * LDS_STRIDE + chunk * 16) = xb[i];
        }
    };

    f32x4_t acc[4][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    const int frag_row = lane & 15, frag_chunk = lane >> 4;
    auto compute = [&](int buf) {
        const unsigned char* At = tiles + buf * 2 * TILE_BYTES + (wc * 64 + frag_row) * LDS_STRIDE + frag_chunk * 16;
        const unsigned char* Bt = tiles + buf * 2 * TILE_BYTES + TILE_BYTES + (wp * 64 + frag_row) * LDS_STRIDE + frag_chunk * 16;
        bf16x8_t a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(At + i * 16 * LDS_STRIDE);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(Bt + j * 16 * LDS_STRIDE);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };

    const int cg = lane >> 4;
    const bool do_relu = p.flags & BD_EPI_RELU;
    const bool add_before = (p.flags & BD_EPI_ADD_BEFORE) && p.add;
    const bool add_after = (p.flags & BD_EPI_ADD_AFTER) && p.add;
    const bool do_mask = (p.flags & BD_EPI_MASK) && p.mask;

    // epilogue operands of the tile being computed, requested when its first K step starts (linear destinations only)
    u32x4_t pre_add[8], pre_mask[8];
    const bool pre_ok = p.linear_dst && (((add_before || add_after)) || do_mask);
    auto prefetch_epi = [&](int tile) {
        const int tile_m = tile / p.n_tiles;
        const int co0 = (tile - tile_m * p.n_tiles) * TILE_C;
        const int cb_ = co0 + wc * 64 + 8 * cg;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int m = tile_m * TILE_P + wp * 64 + j * 16 + (lane & 15);
                const bool ok = m < p.M && cb_ + 32 * half < p.CO;
                const long long idx = (long long)m * p.CO + cb_ + 32 * half;
                u32x4_t a = {0u, 0u, 0u, 0u}, k = {0u, 0u, 0u, 0u};
                if (ok && (add_before || add_after)) a = *reinterpret_cast<const u32x4_t*>(p.add + idx);
                if (ok && do_mask) k = *reinterpret_cast<const u32x4_t*>(p.mask + idx);
                pre_add[j * 2 + half] = a; pre_mask[j * 2 + half] = k;
            }
    };

    auto epilogue = [&](int tile) {
        const int tile_m = tile / p.n_tiles;
        const int co0 = (tile - tile_m * p.n_tiles) * TILE_C;
        const int m0 = tile_m * TILE_P;
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
            int dstpix = m;
            if (!p.linear_dst) {
                const SubSeg ss = p.sub[find_sub(p, m)];
                const int local = m - ss.m_start;
                int n, rem, yy, xx;
                fast_divmod(local, ss.Hs * ss.Ws, ss.inv_per_img, n, rem);
                fast_divmod(rem, ss.Ws, ss.inv_ws, yy, xx);
                dstpix = n * p.dst_pix_per_img + ss.dst_off + (ss.y0 + ss.step * yy) * ss.Wd + ss.x0 + ss.step * xx;
            }
            const long long base = (long long)dstpix * p.CO + cbase;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (cbase + 32 * half >= p.CO) continue;
                const long long idx = base + 32 * half;
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = acc[2 * half + (k >> 2)][j][k & 3] + bias[8 * half + k];
                if (add_before) {
                    const u32x4_t av = pre_ok ? pre_add[j * 2 + half] : *reinterpret_cast<const u32x4_t*>(p.add + idx);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v[2 * k] += bf_lo(av[k]); v[2 * k + 1] += bf_hi(av[k]); }
                }
                if (do_relu) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
                }
                if (do_mask) {
                    const u32x4_t mv = pre_ok ? pre_mask[j * 2 + half] : *reinterpret_cast<const u32x4_t*>(p.mask + idx);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (!(bf_lo(mv[k]) > 0.f)) v[2 * k] = 0.f;
                        if (!(bf_hi(mv[k]) > 0.f)) v[2 * k + 1] = 0.f;
                    }
                }
                if (add_after) {
                    const u32x4_t av = pre_ok ? pre_add[j * 2 + half] : *reinterpret_cast<const u32x4_t*>(p.add + idx);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v[2 * k] += bf_lo(av[k]); v[2 * k + 1] += bf_hi(av[k]); }
                }
                u32x4_t o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = pack_bf2(v[2 * k], v[2 * k + 1]);
                *reinterpret_cast<u32x4_t*>(p.dst + idx) = o;
            }
        }
    };

    // ---- prologue: fill the ring ----
#pragma unroll
    for (int s = 0; s < DEPTH; ++s)
        if (ls < total_steps) issue_load(ra[s], rb[s]);

    int c_tile = (int)blockIdx.x, c_kb = 0;
    for (int cs = 0; cs < total_steps; cs += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const int step = cs + u;
            if (step < total_steps) {
                write_lds(step & 1, ra[u], rb[u]);
                __syncthreads();
                if (ls < total_steps) issue_load(ra[u], rb[u]);
                if (pre_ok && c_kb == 0) prefetch_epi(c_tile);
                compute(step & 1);
                if (++c_kb == kblocks) {
                    epilogue(c_tile);
                    zero_acc();
                    c_kb = 0;
                    c_tile += G;
                }
            }
        }
    }
}

}  // namespace

int bd_conv1x1_stream_launch(const IgemmParams& p, hipStream_t stream) {
    const int total_tiles = p.m_tiles * p.n_tiles;
    int grid = 512;                       // 2 persistent workgroups per CU
    if (grid > total_tiles) grid = total_tiles;
    hipLaunchKernelGGL(conv1x1_stream_kernel, dim3(grid), dim3(256), 0, stream, p);
    return 0;
}
