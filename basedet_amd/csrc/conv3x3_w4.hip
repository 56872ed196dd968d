// 3x3 / stride-1 / pad-1 convolution forward and data-gradient, four-wave instance of the "patch" implicit GEMM of conv3x3.hip.
//
// Same workgroup tile (128 output channels x 256 pixels = four 4x16 patches), same LDS image (54 KB activation patches + 2 x 48 KB
// weight rows, XOR-swizzled 128-byte rows, weights by global_load_lds), but 256 threads = ONE wave per SIMD with the whole 512-entry
// register file: a wave owns 64 channels x 128 pixels (32 MFMA tiles, 128 accumulator VGPRs), so a fragment set of 4 A + 8 B reads
// feeds 32 MFMAs (0.375 ds_read_b128 per MFMA instead of 0.5), and there is room for three fragment sets: the reads of sub-step
// u+2 are issued behind the first MFMAs of sub-step u, i.e. >= 1000 MFMA-pipe cycles before their use.  With one wave per SIMD there
// is no older/younger wave skew at the barrier (conv3x3.hip loses ~1500 cycles per step to it, profiles/r01_patch_timeline.txt).
// Needs CK % 64 == 0; other shapes stay on conv3x3.hip.
#include "common.h"

namespace {

constexpr int PH = 4, PW = 16;
constexpr int IH = PH + 2, IW = PW + 2;
constexpr int NPATCH = 4;
constexpr int XROWS = NPATCH * IH * IW;        // 432 LDS rows
constexpr int X_PITCH = 144;                   // 128 B of channels + 16 B pad: 16 consecutive rows hit 16 distinct bank quads, and a
                                               // tap shift is a compile-time byte offset (no swizzle arithmetic in the MFMA loop)
constexpr int X_BYTES = XROWS * X_PITCH;       // 62208
constexpr int TAPS_PER_STEP = 3;
constexpr int W_TAP_BYTES = 128 * 128;
constexpr int W_BYTES = TAPS_PER_STEP * W_TAP_BYTES;
constexpr int NT = 256;                        // threads
constexpr int DMA_PIECES = 48 / 4;             // 1-KiB pieces per wave and step
constexpr int XCHUNKS = XROWS * 8;
constexpr int XPASSES = (XCHUNKS + NT - 1) / NT;   // 14
constexpr int TILE_CO = 128;
constexpr int MAX_SEG = BD_MAX_SEGS;

struct CSeg4 { int patch_start, H, W, pw, src_off, dst_off; };

struct C3P4 {
    const bf16_raw* src;
    const bf16_raw* w;       // [CO][9][CK]
    const float* bias;
    const bf16_raw* add;
    const bf16_raw* mask;
    bf16_raw* dst;
    int CK, CO, mode, flags, N, nseg;
    int src_ppi, dst_ppi;
    int patches_per_img, total_patches, n_tiles;
    CSeg4 seg[MAX_SEG];
};

__device__ __forceinline__ int swz4(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

typedef __attribute__((address_space(3))) void lds_void4_t;
typedef __attribute__((address_space(1))) const void glb_void4_t;

template <int MODE>
__global__ __launch_bounds__(NT, 1) void conv3x3_patch4w_kernel(const C3P4 p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xbuf = smem;
    unsigned char* wbuf = smem + X_BYTES;
    float* sbias = reinterpret_cast<float*>(smem + X_BYTES + 2 * W_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave >> 1, wq = wave & 1;          // channel half, pixel half (patches 2wq, 2wq+1)
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int pt = bid / p.n_tiles;
    const int ct = bid - pt * p.n_tiles;
    const int co0 = ct * TILE_CO;

    int py0[NPATCH], px0[NPATCH], pH[NPATCH], pWd[NPATCH];
    long long psrc[NPATCH], pdst[NPATCH];
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
            const CSeg4 sg = p.seg[s];
            const int local = rem - sg.patch_start;
            const int by = local / sg.pw, bx = local - by * sg.pw;
            py0[k] = by * PH; px0[k] = bx * PW; pH[k] = sg.H; pWd[k] = sg.W;
            psrc[k] = (long long)n * p.src_ppi + sg.src_off;
            pdst[k] = (long long)n * p.dst_ppi + sg.dst_off;
        }
    }

    // activation staging slots: chunk id c = tid + 256 k -> (row, chunk)
    unsigned int x_off[XPASSES];      // element offset of the source pixel (32 bits: the launcher checks the tensor size), ~0u = zero-fill
    int x_lds[XPASSES];
    const int x_chunk = tid & 7;
#pragma unroll
    for (int k = 0; k < XPASSES; ++k) {
        const int c = tid + NT * k;
        const int row = c >> 3;
        x_off[k] = ~0u; x_lds[k] = -1;
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
            x_lds[k] = row * X_PITCH + x_chunk * 16;
            if (y >= 0 && x >= 0 && y < H && x < W) x_off[k] = (unsigned int)((qs + (long long)y * W + x) * p.CK);
        }
    }

    // DMA pieces of this wave: piece pc = wave + 4k (k = 0..11) -> tap pc / 16, rows 8 (pc % 16) .. +7; lane -> (row, position)
    int dma_src[DMA_PIECES];
#pragma unroll
    for (int k = 0; k < DMA_PIECES; ++k) {
        const int pc = wave + 4 * k;
        const int lrow = 8 * (pc & 15) + (lane >> 3);
        const int chunk = (lane & 7) ^ (lane >> 3);
        const int rho = lrow & 15;
        int co = co0 + (lrow & 64) + 32 * ((lrow >> 5) & 1) + 8 * (rho >> 2) + 4 * ((lrow >> 4) & 1) + (rho & 3);
        if (co >= p.CO) co = p.CO - 1;
        dma_src[k] = (co * 9 + (pc >> 4)) * p.CK + chunk * 8;
    }

    const int kblocks = p.CK / 64;
    const int nsteps = kblocks * 3;
    u32x4_t rx[XPASSES];
    auto dma_w = [&](int step, int buf, int k) {
        const int cb = step / 3, r = step - cb * 3;
        const int pc = wave + 4 * k;
        const bf16_raw* g = p.w + dma_src[k] + r * 3 * p.CK + cb * 64;
        unsigned char* l = wbuf + buf * W_BYTES + (pc >> 4) * W_TAP_BYTES + (pc & 15) * 1024;
        __builtin_amdgcn_global_load_lds((glb_void4_t*)g, (lds_void4_t*)l, 16, 0, 0);
    };
    auto load_x = [&](int cb) {
        const int c0 = cb * 64 + x_chunk * 8;
#pragma unroll
        for (int k = 0; k < XPASSES; ++k) {
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (x_off[k] != ~0u) v = *reinterpret_cast<const u32x4_t*>(p.src + (size_t)x_off[k] + c0);
            rx[k] = v;
        }
    };
    auto write_x = [&]() {
#pragma unroll
        for (int k = 0; k < XPASSES; ++k)
            if (x_lds[k] >= 0) *reinterpret_cast<u32x4_t*>(xbuf + x_lds[k]) = rx[k];
    };

    f32x4_t acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fchunk = lane >> 4;
    int a_off[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) a_off[i][kk] = swz4(wc * 64 + i * 16 + frow, kk * 4 + fchunk);
    // B fragments: lane base + compile-time offsets (patch, row, tap shift, K half)
    const unsigned char* b_base = xbuf + ((2 * wq) * (IH * IW) + frow) * X_PITCH + fchunk * 16;

    bf16x8_t fa[3][4], fb[3][8];
    auto load_frags = [&](int wb, int r, int u, bf16x8_t (&a)[4], bf16x8_t (&b)[8]) {
        const int t = u >> 1, kk = u & 1;
        int dy = r, dx = t;
        if (MODE == 1) { dy = 2 - dy; dx = 2 - dx; }        // dgrad: mirrored tap
        const unsigned char* Wt = wbuf + wb * W_BYTES + t * W_TAP_BYTES;
        const unsigned char* Bt = b_base + dy * (IW * X_PITCH);      // r is a runtime value: one multiply per step, hoisted
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(Wt + a_off[i][kk]);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            b[j] = *reinterpret_cast<const bf16x8_t*>(Bt + ((j >> 2) * (IH * IW) + (j & 3) * IW + dx) * X_PITCH + kk * 64);
    };
    auto mfma_rows = [&](const bf16x8_t (&a)[4], const bf16x8_t (&b)[8], int i0, int i1) {
#pragma unroll
        for (int i = i0; i < i1; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    auto compute = [&](int wb, int r, bool write_next, int next_step) {
        load_frags(wb, r, 0, fa[0], fb[0]);
        load_frags(wb, r, 1, fa[1], fb[1]);
#pragma unroll
        for (int u = 0; u < 2 * TAPS_PER_STEP; ++u) {
            __builtin_amdgcn_sched_barrier(0);
            mfma_rows(fa[u % 3], fb[u % 3], 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (u + 2 < 2 * TAPS_PER_STEP) load_frags(wb, r, u + 2, fa[(u + 2) % 3], fb[(u + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);
            mfma_rows(fa[u % 3], fb[u % 3], 1, 2);
            // the next step's weight rows: three 1-KiB pieces behind each of the first four sub-steps (one wave per SIMD: a burst of
            // twelve would hold the MFMA pipe idle), landed well before the barrier's vmcnt(0)
            if (write_next && u < 4) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 3; ++k) dma_w(next_step, wb ^ 1, 3 * u + k);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_rows(fa[u % 3], fb[u % 3], 2, 4);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (tid < TILE_CO) sbias[tid] = (p.bias && co0 + tid < p.CO) ? p.bias[co0 + tid] : 0.f;
    load_x(0);
#pragma unroll
    for (int k = 0; k < DMA_PIECES; ++k) dma_w(0, 0, k);
    write_x();
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        const int cb = step / 3, r = step - cb * 3;
        const bool next_x = (r == 2) && (cb + 1 < kblocks);
        if (next_x) load_x(cb + 1);
        compute(step & 1, r, step + 1 < nsteps, step + 1);
        __syncthreads();
        if (next_x) {
            write_x();
            __syncthreads();
        }
    }

    // ---- epilogue ----
    const int cg = lane >> 4;
    const bool do_relu = p.flags & BD_EPI_RELU;
    const bool add_before = (p.flags & BD_EPI_ADD_BEFORE) && p.add;
    const bool add_after = (p.flags & BD_EPI_ADD_AFTER) && p.add;
    const bool do_mask = (p.flags & BD_EPI_MASK) && p.mask;
    const int cbase = co0 + wc * 64 + 8 * cg;
    float bias[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(sbias + wc * 64 + 8 * cg + 32 * (q >> 1) + 4 * (q & 1));
        bias[4 * q] = bv[0]; bias[4 * q + 1] = bv[1]; bias[4 * q + 2] = bv[2]; bias[4 * q + 3] = bv[3];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int pk = 2 * wq + (j >> 2);
        int oy0 = py0[0], ox = px0[0] + frow, H = pH[0], W = pWd[0];
        long long dbase = pdst[0];
#pragma unroll
        for (int q = 1; q < NPATCH; ++q)
            if (pk == q) { oy0 = py0[q]; ox = px0[q] + frow; H = pH[q]; W = pWd[q]; dbase = pdst[q]; }
        const int oy = oy0 + (j & 3);
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
}

}  // namespace

// called from conv3x3.hip when the four-wave instance is selected (bd_conv_set_patch3x3 bit 4) and CK % 64 == 0
int bd_conv3x3_patch4w_launch(const bd_conv_desc* d, int mode, const void* src, const void* w, const float* bias, const void* add,
                              const void* mask, void* dst, int flags, hipStream_t stream) {
    C3P4 p{};
    p.src = (const bf16_raw*)src; p.w = (const bf16_raw*)w; p.bias = bias;
    p.add = (const bf16_raw*)add; p.mask = (const bf16_raw*)mask; p.dst = (bf16_raw*)dst;
    p.CK = mode == 0 ? d->Cin : d->Cout;
    p.CO = mode == 0 ? d->Cout : d->Cin;
    p.mode = mode; p.flags = flags; p.N = d->N; p.nseg = d->nseg;
    p.src_ppi = mode == 0 ? d->in_pix_per_img : d->out_pix_per_img;
    p.dst_ppi = mode == 0 ? d->out_pix_per_img : d->in_pix_per_img;
    int ps = 0;
    for (int s = 0; s < d->nseg; ++s) {
        CSeg4& sg = p.seg[s];
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
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_patch4w_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_patch4w_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (mode == 0) hipLaunchKernelGGL(conv3x3_patch4w_kernel<0>, dim3(grid), dim3(NT), lds, stream, p);
    else hipLaunchKernelGGL(conv3x3_patch4w_kernel<1>, dim3(grid), dim3(NT), lds, stream, p);
    return 0;
}
