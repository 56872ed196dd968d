// Convolution weight gradient on CDNA4 MFMA (gfx950).
//
//   dW[co][tap][ci] = sum_pix  G[pix][co] * X[srcpix(pix, tap)][ci]
//
// The reduction index (pixel) is the slow index of both NHWC operands, so both MFMA operands need a
// transpose.  Tiles are staged [pixel][channel] exactly as they lie in HBM (coalesced 16-byte loads,
// conflict-free ds_write_b128) and fragments are fetched with gfx950's transposing LDS read
// ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group, delivered pixel-contiguous per channel).
// The MFMA k index -> pixel mapping is permuted (same permutation for both operands) so that the two
// 16-lane groups of a half-wave read 8 distinct LDS rows = all 64 banks: conflict-free with the 288-byte
// row pitch.
//
// D'[ci][co]: A operand = X^T (rows = ci), B operand = G (cols = co)  => each lane owns 4 consecutive ci of one
// co, i.e. one 16-byte fp32 store into the [co][tap][ci] slab.
//
// Split over pixels: grid = splits x taps x ci-tiles x co-tiles; every workgroup writes its own fp32 slab,
// a second kernel sums the slabs in a fixed order (bitwise reproducible), applies the FrozenBN row scale and
// stores / accumulates into the fp32 gradient.
#include <vector>
#include "common.h"

namespace {

constexpr int TILE = 128;       // channels per tile side
constexpr int ROW_PITCH = 288;  // bytes per staged pixel row (256 data + 32 pad)
constexpr int MAX_SEG = BD_MAX_SEGS;

struct WSeg {
    int m_start;
    int Ho, Wo, per_img;
    float inv_per_img, inv_wo;
    int Hi, Wi, in_off, out_off;
};

struct WgradParams {
    const bf16_raw* x;
    const bf16_raw* g;
    float* slab;     // [splits][Cout][RS][Cin]
    int Cin, Cout, R, S, stride, pad;
    int M, nseg;
    int in_pix_per_img, out_pix_per_img;
    int ci_tiles, co_tiles, splits, steps_per_split, total_steps;
    WSeg seg[MAX_SEG];
};

__device__ __forceinline__ void fast_divmod(int n, int d, float inv, int& q, int& r) {
    q = (int)((float)n * inv);
    r = n - q * d;
    if (r < 0) { --q; r += d; }
    else if (r >= d) { ++q; r -= d; }
}

template <int BKP, bool USE_TR>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
    constexpr int PASSES = BKP / 16;
    constexpr int TILE_BYTES = BKP * ROW_PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wci = wave >> 1;  // ci half (MFMA rows)
    const int wco = wave & 1;   // co half (MFMA cols)

    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int RS = p.R * p.S;
    const int tiles_per_split = RS * p.ci_tiles * p.co_tiles;
    const int split = bid / tiles_per_split;
    int t = bid - split * tiles_per_split;
    const int tap = t / (p.ci_tiles * p.co_tiles);
    t -= tap * (p.ci_tiles * p.co_tiles);
    const int ci_tile = t / p.co_tiles;
    const int co_tile = t - ci_tile * p.co_tiles;
    const int ci0 = ci_tile * TILE, co0 = co_tile * TILE;
    const int tap_r = tap / p.S, tap_s = tap - tap_r * p.S;

    const int step_begin = split * p.steps_per_split;
    int step_end = step_begin + p.steps_per_split;
    if (step_end > p.total_steps) step_end = p.total_steps;

    const int chunk = tid & 15;
    const int row0 = tid >> 4;
    const bool ci_ok = ci0 + chunk * 8 < p.Cin;   // Cin % 8 == 0
    const bool co_ok = co0 + chunk * 8 < p.Cout;  // Cout % 8 == 0

    u32x4_t rx[PASSES], rg[PASSES];

    auto stage_load = [&](int step) {
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int m = step * BKP + row0 + i * 16;
            u32x4_t vx = {0u, 0u, 0u, 0u}, vg = {0u, 0u, 0u, 0u};
            if (m < p.M) {
                int s = 0;
#pragma unroll
                for (int k = 1; k < MAX_SEG; ++k)
                    if (k < p.nseg && m >= p.seg[k].m_start) s = k;
                const WSeg sg = p.seg[s];
                int n, rem, oy, ox;
                fast_divmod(m - sg.m_start, sg.per_img, sg.inv_per_img, n, rem);
                fast_divmod(rem, sg.Wo, sg.inv_wo, oy, ox);
                if (co_ok) {
                    const long long gp = (long long)n * p.out_pix_per_img + sg.out_off + oy * sg.Wo + ox;
                    vg = *reinterpret_cast<const u32x4_t*>(p.g + gp * p.Cout + co0 + chunk * 8);
                }
                const int sy = oy * p.stride - p.pad + tap_r, sx = ox * p.stride - p.pad + tap_s;
                if (ci_ok && sy >= 0 && sx >= 0 && sy < sg.Hi && sx < sg.Wi) {
                    const long long xp = (long long)n * p.in_pix_per_img + sg.in_off + sy * sg.Wi + sx;
                    vx = *reinterpret_cast<const u32x4_t*>(p.x + xp * p.Cin + ci0 + chunk * 8);
                }
            }
            rx[i] = vx; rg[i] = vg;
        }
    };
    auto stage_write = [&](int buf) {
        unsigned char* Xt = smem + buf * 2 * TILE_BYTES;
        unsigned char* Gt = Xt + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int row = row0 + i * 16;
            *reinterpret_cast<u32x4_t*>(Xt + row * ROW_PITCH + chunk * 16) = rx[i];
            *reinterpret_cast<u32x4_t*>(Gt + row * ROW_PITCH + chunk * 16) = rg[i];
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // k -> pixel permutation (identical for A and B): k = 8*g4 + j  <->  pixel 16*(g4>>1) + 4*(g4&1) + (j&3) + 8*(j>>2)
    const int g4 = lane >> 4, idx = lane & 15;
    const int prow_base = 16 * (g4 >> 1) + 4 * (g4 & 1);
    // transposing read: lane 4q+p of the group supplies &tile[row + q][col0 + 4p]
    const int tr_q = idx >> 2, tr_p = idx & 3;

    auto load_frag = [&](const unsigned char* tile, int kk, int cbase) -> bf16x8_t {
        bf16x8_t f;
        if constexpr (USE_TR) {
            const unsigned char* a0 = tile + (kk * 32 + prow_base + tr_q) * ROW_PITCH + (cbase + 4 * tr_p) * 2;
            s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0));
            s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0 + 8 * ROW_PITCH));
            typedef __attribute__((ext_vector_type(8))) short s16x8_t;
            s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            f = __builtin_bit_cast(bf16x8_t, v);
        } else {
            typedef __attribute__((ext_vector_type(8))) short s16x8_t;
            s16x8_t v;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int prow = kk * 32 + prow_base + (j & 3) + 8 * (j >> 2);
                v[j] = *reinterpret_cast<const short*>(tile + prow * ROW_PITCH + (cbase + idx) * 2);
            }
            f = __builtin_bit_cast(bf16x8_t, v);
        }
        return f;
    };

    auto compute = [&](int buf) {
        const unsigned char* Xt = smem + buf * 2 * TILE_BYTES;
        const unsigned char* Gt = Xt + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < BKP / 32; ++kk) {
            bf16x8_t a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = load_frag(Xt, kk, wci * 64 + i * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = load_frag(Gt, kk, wco * 64 + j * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };

    if (step_begin < step_end) {
        stage_load(step_begin);
        stage_write(0);
    }
    __syncthreads();
    int cur = 0;
    for (int st = step_begin; st < step_end; ++st) {
        const bool more = st + 1 < step_end;
        if (more) stage_load(st + 1);
        compute(cur);
        if (more) stage_write(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // D'[row = ci][col = co]: lane holds ci = (lane>>4)*4 + r (4 consecutive), co = lane & 15
    float* slab = p.slab + (long long)split * p.Cout * RS * p.Cin;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co = co0 + wco * 64 + j * 16 + idx;
        if (co >= p.Cout) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ci = ci0 + wci * 64 + i * 16 + g4 * 4;
            if (ci >= p.Cin) continue;   // Cin % 4 == 0
            *reinterpret_cast<f32x4_t*>(slab + ((long long)co * RS + tap) * p.Cin + ci) = acc[i][j];
        }
    }
}

// ---- fixed-order reduce of the per-split partial sums, BATCHED over layers ---------------------------------------------------------------
// One launch sums the slabs of up to BD_RED_MAX entries (a layer's weight gradient; its bias column sums are an entry of their own).
// Block = 64 f32x4 elements x 4 split chains: chain y sums splits y, y + 4, ... in order, the four partial sums are added in chain order ->
// bitwise reproducible whatever the grid looks like.  Rounds 1-3 launched one such kernel per layer (60 launches per RetinaNet-R50 step).
// kind 0: slab[split][n] floats in the result's own order (generic / register-staged kernels; bias column sums with row_len = 1, no scale);
// kind 1 / 2: the register-row slabs of conv_wgrad3x3_ring.hip / conv_wgrad1x1_ring.hip (slab[split][tile][wave][reg][lane] f32x4), un-permuted here.
}  // namespace
#include "wgrad_reduce.h"
namespace {

__global__ __launch_bounds__(256) void wgrad_batch_reduce_kernel(const BdRedBatch b) {
    __shared__ f32x4_t red[4][64];
    int ei = 0;
#pragma unroll 1
    for (int k = 1; k < b.count; ++k)
        if ((int)blockIdx.x >= b.e[k].block_begin) ei = k;
    const BdRedEntry& en = b.e[ei];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long long e = (long long)((int)blockIdx.x - en.block_begin) * 64 + tx;        // f32x4 element inside one split
    const long long n4 = en.n4;
    f32x4_t s = {0.f, 0.f, 0.f, 0.f};
    if (e < n4) {
        const long long stride = n4 * 4;
        const float* src = en.slab + e * 4;
        const int splits = en.splits;
        int k = ty;
        for (; k + 12 < splits; k += 16) {
            const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(src + (long long)k * stride);
            const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(src + (long long)(k + 4) * stride);
            const f32x4_t v2 = *reinterpret_cast<const f32x4_t*>(src + (long long)(k + 8) * stride);
            const f32x4_t v3 = *reinterpret_cast<const f32x4_t*>(src + (long long)(k + 12) * stride);
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; k < splits; k += 4) s += *reinterpret_cast<const f32x4_t*>(src + (long long)k * stride);
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty != 0 || e >= n4) return;
    f32x4_t t = red[0][tx];
    t += red[1][tx]; t += red[2][tx]; t += red[3][tx];
    long long idx;          // float index into dw
    int row;                // output channel (row scale)
    if (en.kind == 0) {
        idx = e * 4;
        row = (int)(idx / en.row_len);
    } else {
        const int regs = en.regs;                                     // f32x4 registers per lane: 36 (3x3: tap * 4 + j) / fi * fj (1x1: i * fj + j)
        const long long per_tile = 8ll * regs * 64;
        const int tile = (int)(e / per_tile);
        int r = (int)(e - tile * per_tile);
        const int wave = r / (regs * 64);
        r -= wave * (regs * 64);
        const int reg = r >> 6, lane = r & 63;
        const int ci_tile = tile / en.co_tiles, co_tile = tile - ci_tile * en.co_tiles;
        int co, ci, tap = 0, taps = 1;
        if (en.kind == 1) {                                           // wave = (16-ci group wave & 3, 64-co group wave >> 2), reg = tap * 4 + j
            tap = reg >> 2; taps = 9;
            co = co_tile * en.tco + (wave >> 2) * 64 + (reg & 3) * 16 + (lane & 15);
            ci = ci_tile * en.tci + (wave & 3) * 16 + (lane >> 4) * 4;
        } else {                                                      // wave = (wci, wco), reg = i * fj + j
            const int i = reg / en.fj, j = reg - i * en.fj;
            const int wco_n = en.tco / (16 * en.fj);
            const int wci = wave / wco_n, wco = wave - wci * wco_n;
            co = co_tile * en.tco + (wco * en.fj + j) * 16 + (lane & 15);
            ci = ci_tile * en.tci + (wci * en.fi + i) * 16 + (lane >> 4) * 4;
        }
        if (co >= en.Cout || ci >= en.Cin) return;                    // Cin % 4 == 0
        idx = ((long long)co * taps + tap) * en.Cin + ci;
        row = co;
    }
    float* d = en.dw + idx;
    if (en.row_scale) t *= en.row_scale[row];
    if (en.accumulate) t += *reinterpret_cast<const f32x4_t*>(d);
    *reinterpret_cast<f32x4_t*>(d) = t;
}

int launch_batch(const BdRedEntry* entries, int count, hipStream_t stream) {
    for (int base = 0; base < count; base += BD_RED_MAX) {
        BdRedBatch b{};
        b.count = count - base < BD_RED_MAX ? count - base : BD_RED_MAX;
        int blocks = 0;
        for (int k = 0; k < b.count; ++k) {
            b.e[k] = entries[base + k];
            b.e[k].block_begin = blocks;
            blocks += (int)cdiv64(b.e[k].n4, 64);
        }
        if (blocks > 0) hipLaunchKernelGGL(wgrad_batch_reduce_kernel, dim3(blocks), dim3(256), 0, stream, b);
    }
    return 0;
}

BdRedEntry plain_entry(const float* slab, int splits, long long n, int row_len, const float* row_scale, float* dw, int accumulate) {
    BdRedEntry e{};
    e.slab = slab; e.dw = dw; e.row_scale = row_scale; e.kind = 0; e.splits = splits; e.accumulate = accumulate;
    e.n4 = (int)(n / 4); e.row_len = row_len;
    return e;
}

struct Plan { int ci_tiles, co_tiles, splits, steps_per_split, total_steps; long long M; };

constexpr int BKP_DEFAULT = 64;
BD_KNOB int g_wgrad_use_tr = 1;
BD_KNOB int g_wgrad_use_3x3 = 1;
BD_KNOB int g_wgrad_use_ring = 1;       // conv_wgrad3x3_ring.hip for the wide stride-1 3x3 layers (bd_conv_desc.route[2] bit 2 set = off)

// 3x3 / pad 1 with stride 1 or 2: served by the nine-tap patch kernel (conv_wgrad3x3.hip)
bool is_3x3s1(const bd_conv_desc* d) {
    if (!(d->R == 3 && d->S == 3 && (d->stride == 1 || d->stride == 2) && d->pad == 1)) return false;
    // the patch kernel addresses both tensors with 32-bit byte offsets (range-checked buffer loads)
    if ((long long)d->N * d->in_pix_per_img * d->Cin * 2 >= 0x7fffffffll || (long long)d->N * d->out_pix_per_img * d->Cout * 2 >= 0x7fffffffll)
        return false;
    for (int s = 0; s < d->nseg; ++s)
        if ((d->Hi[s] - 1) / d->stride + 1 != d->Ho[s] || (d->Wi[s] - 1) / d->stride + 1 != d->Wo[s]) return false;
    return true;
}

Plan make_plan(const bd_conv_desc* d) {
    Plan pl;
    long long M = 0;
    for (int s = 0; s < d->nseg; ++s) M += (long long)d->N * d->Ho[s] * d->Wo[s];
    pl.M = M;
    pl.ci_tiles = cdiv(d->Cin, TILE);
    pl.co_tiles = cdiv(d->Cout, TILE);
    pl.total_steps = (int)cdiv64(M, BKP_DEFAULT);
    const int tiles = d->R * d->S * pl.ci_tiles * pl.co_tiles;
    int splits = 1024 / tiles;
    if (splits < 1) splits = 1;
    const int max_splits = pl.total_steps / 4 > 0 ? pl.total_steps / 4 : 1;
    if (splits > max_splits) splits = max_splits;
    pl.steps_per_split = cdiv(pl.total_steps, splits);
    pl.splits = cdiv(pl.total_steps, pl.steps_per_split);
    return pl;
}

}  // namespace

int bd_wgrad1x1_splits(const bd_conv_desc* d);
int bd_wgrad1x1_launch(const bd_conv_desc* d, const void* x, const void* g, float* slab, int* splits_out, hipStream_t stream);
int bd_wgrad3x3_splits(const bd_conv_desc* d, int* total_patches_out, int* patches_per_img_out);
int bd_wgrad3x3_launch(const bd_conv_desc* d, const void* x, const void* g, float* slab, float* csum, int* splits_out, hipStream_t stream);
bool bd_wgrad3x3r_eligible(const bd_conv_desc* d);
size_t bd_wgrad3x3r_slab_bytes(const bd_conv_desc* d, int* splits_out);
int bd_wgrad3x3r_launch(const bd_conv_desc* d, const void* x, const void* g, float* slab, float* csum, int* splits_out, hipStream_t stream);
void bd_wgrad3x3r_entry(const bd_conv_desc* d, BdRedEntry* e);
bool bd_wgrad1x1r_eligible(const bd_conv_desc* d);
size_t bd_wgrad1x1r_slab_bytes(const bd_conv_desc* d, int* splits_out);
int bd_wgrad1x1r_launch(const bd_conv_desc* d, const void* x, const void* g, float* slab, int* splits_out, hipStream_t stream);
void bd_wgrad1x1r_entry(const bd_conv_desc* d, BdRedEntry* e);

// use_tr: 1 = transposing LDS reads (default), 0 = scalar-read reference path of the generic kernel.
// bit 1 (value 2) additionally disables the nine-tap 3x3 kernels (forces the generic per-tap kernel);
// bit 2 (value 4) disables the ring-staged kernels only (conv_wgrad3x3_ring.hip, conv_wgrad1x1_ring.hip; their shapes then take
// conv_wgrad3x3.hip / conv_wgrad1x1.hip).
// bd_conv_desc.route[2] - 1 (BdRouteScope, conv_igemm.hip)
int bd_route_wgrad(int use_tr) {
    g_wgrad_use_tr = use_tr & 1;
    g_wgrad_use_3x3 = (use_tr & 2) ? 0 : 1;
    g_wgrad_use_ring = (use_tr & 4) ? 0 : 1;
    if (!(use_tr & 1)) g_wgrad_use_3x3 = 0;
    return BD_OK;
}

// the fixed-order slab reduce for the other translation units (conv_wgrad3x3_fp8.hip)
void bd_wgrad_reduce_launch(const float* slab, int splits, long long n, int row_len, const float* row_scale, float* dw, int accumulate,
                            hipStream_t stream) {
    const BdRedEntry e = plain_entry(slab, splits, n, row_len, row_scale, dw, accumulate);
    launch_batch(&e, 1, stream);
}

extern "C" size_t bd_conv2d_wgrad_workspace_bytes(const bd_conv_desc* d) {
    BdRouteScope rs__(d);
    if (rs__.rc != BD_OK) return 0;
    if (!d || d->nseg < 1 || d->nseg > BD_MAX_SEGS) return 0;
    const Plan pl = make_plan(d);
    size_t splits = (size_t)pl.splits;
    if (is_3x3s1(d)) {
        const size_t s3 = (size_t)bd_wgrad3x3_splits(d, nullptr, nullptr);
        if (s3 > splits) splits = s3;
    }
    if (d->R == 1 && d->S == 1 && d->pad == 0) {
        const size_t s1 = (size_t)bd_wgrad1x1_splits(d);
        if (s1 > splits) splits = s1;
    }
    size_t bytes = splits * d->Cout * d->R * d->S * d->Cin * sizeof(float);
    if (bd_wgrad3x3r_eligible(d)) {                 // register-row slabs of whole (padded) tiles
        const size_t r = bd_wgrad3x3r_slab_bytes(d, nullptr);
        if (r > bytes) bytes = r;
    }
    if (bd_wgrad1x1r_eligible(d)) {
        const size_t r = bd_wgrad1x1r_slab_bytes(d, nullptr);
        if (r > bytes) bytes = r;
    }
    return bytes;
}

namespace {
size_t align256(size_t v) { return (v + 255) & ~size_t(255); }
}  // namespace

extern "C" size_t bd_colsum_workspace_bytes(int C);
extern "C" int bd_colsum_bf16(const void* g, int N, int64_t pix_per_img, int64_t off, int64_t cnt, int C, float* out, int accumulate,
                              void* ws, size_t ws_bytes, bd_stream_t stream);
struct bd_wgrad_queue { std::vector<BdRedEntry> entries; };
static int wgrad_impl(const bd_conv_desc* d, const void* x, const void* g, const float* row_scale, float* dw, float* dbias,
                      int accumulate, void* ws, size_t ws_bytes, bd_stream_t stream, bd_wgrad_queue* q);

extern "C" size_t bd_conv2d_wgrad_bias_workspace_bytes(const bd_conv_desc* d) {
    BdRouteScope rs__(d);
    if (rs__.rc != BD_OK) return 0;
    const size_t w = bd_conv2d_wgrad_workspace_bytes(d);
    if (!w) return 0;
    size_t extra = bd_colsum_workspace_bytes(d->Cout);
    if (is_3x3s1(d)) {
        const size_t s3 = (size_t)bd_wgrad3x3_splits(d, nullptr, nullptr) * d->Cout * sizeof(float);
        if (s3 > extra) extra = s3;
    }
    if (bd_wgrad3x3r_eligible(d)) {
        int sr = 1;
        (void)bd_wgrad3x3r_slab_bytes(d, &sr);
        const size_t s3 = (size_t)sr * d->Cout * sizeof(float);
        if (s3 > extra) extra = s3;
    }
    return align256(w) + extra;
}

extern "C" int bd_conv2d_wgrad_bias(const bd_conv_desc* d, const void* x, const void* g, const float* row_scale, float* dw,
                                    float* dbias, int accumulate, void* ws, size_t ws_bytes, bd_stream_t stream) {
    BD_ROUTE(d);
    BD_REQUIRE(dbias, "conv2d_wgrad_bias: null bias gradient");
    BD_REQUIRE(d && d->Cout <= 2048, "conv2d_wgrad_bias: Cout must be <= 2048");
    const size_t need = bd_conv2d_wgrad_bias_workspace_bytes(d);
    if (ws_bytes < need) {
        bd_set_error("conv2d_wgrad_bias: workspace %zu < required %zu bytes", ws_bytes, need);
        return BD_EWORKSPACE;
    }
    return wgrad_impl(d, x, g, row_scale, dw, dbias, accumulate, ws, ws_bytes, stream, nullptr);
}

extern "C" int bd_conv2d_wgrad(const bd_conv_desc* d, const void* x, const void* g, const float* row_scale,
                               float* dw, int accumulate, void* ws, size_t ws_bytes, bd_stream_t stream) {
    BD_ROUTE(d);
    return wgrad_impl(d, x, g, row_scale, dw, nullptr, accumulate, ws, ws_bytes, stream, nullptr);
}

// ---- deferred reduces: the partial-sum kernels of several layers run back to back, ONE launch reduces them all ---------------------------
extern "C" int bd_wgrad_queue_create(bd_wgrad_queue_t* out) {
    BD_REQUIRE(out != nullptr, "bd_wgrad_queue_create: null argument");
    *out = new bd_wgrad_queue();
    return BD_OK;
}
extern "C" int bd_wgrad_queue_destroy(bd_wgrad_queue_t q) {
    delete q;
    return BD_OK;
}
extern "C" int bd_wgrad_queue_pending(bd_wgrad_queue_t q) { return q ? (int)q->entries.size() : 0; }
extern "C" int bd_conv2d_wgrad_queued(bd_wgrad_queue_t q, const bd_conv_desc* d, const void* x, const void* g, const float* row_scale, float* dw,
                                      float* dbias, int accumulate, void* ws, size_t ws_bytes, bd_stream_t stream) {
    BD_ROUTE(d);
    BD_REQUIRE(q != nullptr, "bd_conv2d_wgrad_queued: null queue");
    if (dbias) {
        BD_REQUIRE(d && d->Cout <= 2048, "conv2d_wgrad_queued: Cout must be <= 2048");
        const size_t need = bd_conv2d_wgrad_bias_workspace_bytes(d);
        if (ws_bytes < need) {
            bd_set_error("conv2d_wgrad_queued: workspace %zu < required %zu bytes", ws_bytes, need);
            return BD_EWORKSPACE;
        }
    }
    return wgrad_impl(d, x, g, row_scale, dw, dbias, accumulate, ws, ws_bytes, stream, q);
}
extern "C" int bd_wgrad_queue_flush(bd_wgrad_queue_t q, bd_stream_t stream) {
    BD_REQUIRE(q != nullptr, "bd_wgrad_queue_flush: null queue");
    if (q->entries.empty()) return BD_OK;
    launch_batch(q->entries.data(), (int)q->entries.size(), (hipStream_t)stream);
    q->entries.clear();
    BD_CHECK_LAUNCH("bd_wgrad_queue_flush");
    return BD_OK;
}

static int wgrad_impl(const bd_conv_desc* d, const void* x, const void* g, const float* row_scale, float* dw, float* dbias,
                      int accumulate, void* ws, size_t ws_bytes, bd_stream_t stream, bd_wgrad_queue* q) {
    BD_REQUIRE(d && x && g && dw && ws, "conv2d_wgrad: null pointer");
    BD_REQUIRE(d->nseg >= 1 && d->nseg <= BD_MAX_SEGS, "conv2d_wgrad: nseg out of range");
    BD_REQUIRE(d->Cin % 8 == 0 && d->Cout % 8 == 0, "conv2d_wgrad: Cin=%d / Cout=%d must be multiples of 8", d->Cin, d->Cout);
    const Plan pl = make_plan(d);
    BD_REQUIRE(pl.M < (1ll << 24), "conv2d_wgrad: %lld pixels exceed the 2^24 fast-division range", pl.M);
    const size_t need = bd_conv2d_wgrad_workspace_bytes(d);
    if (ws_bytes < need) {
        bd_set_error("conv2d_wgrad: workspace %zu < required %zu bytes", ws_bytes, need);
        return BD_EWORKSPACE;
    }
    // bias gradient (bd_conv2d_wgrad_bias): fused into the nine-tap kernel, which reads every row of g anyway; the other kernels are
    // followed by the stand-alone column-sum pass, level by level.  Its scratch sits behind the weight slabs.
    unsigned char* extra = (unsigned char*)ws + align256(need);
    auto bias_fallback = [&]() -> int {
        long long dense = 0;                      // levels packed back to back from offset 0: one pass over the whole tensor
        bool packed = true;
        for (int s = 0; s < d->nseg; ++s) { packed = packed && d->out_off[s] == dense; dense += (long long)d->Ho[s] * d->Wo[s]; }
        if (packed && dense == d->out_pix_per_img)
            return bd_colsum_bf16(g, 1, 0, 0, (int64_t)d->N * dense, d->Cout, dbias, accumulate, extra, bd_colsum_workspace_bytes(d->Cout), stream);
        for (int s = 0; s < d->nseg; ++s) {
            const int e = bd_colsum_bf16(g, d->N, d->out_pix_per_img, d->out_off[s], (int64_t)d->Ho[s] * d->Wo[s], d->Cout, dbias,
                                         (s > 0 || accumulate) ? 1 : 0, extra, bd_colsum_workspace_bytes(d->Cout), stream);
            if (e != BD_OK) return e;
        }
        return BD_OK;
    };
    // the reduce of this layer: its weight slabs and, where the kernel left per-split column sums of g behind (fused_csum), the bias
    // gradient -- launched at once, or handed to the queue (bd_conv2d_wgrad_queued) for the next bd_wgrad_queue_flush
    auto finish = [&](const BdRedEntry& we, int splits, bool fused_csum) -> int {
        BdRedEntry es[2] = {we, {}};
        int n = 1;
        if (dbias && fused_csum) es[n++] = plain_entry((const float*)extra, splits, d->Cout, 1, nullptr, dbias, accumulate);
        if (q) {
            for (int k = 0; k < n; ++k) q->entries.push_back(es[k]);
        } else {
            launch_batch(es, n, (hipStream_t)stream);
            BD_CHECK_LAUNCH("bd_conv2d_wgrad(reduce)");
        }
        return (dbias && !fused_csum) ? bias_fallback() : BD_OK;
    };
    if (g_wgrad_use_3x3 && g_wgrad_use_ring && bd_wgrad3x3r_eligible(d)) {
        int splitsr = 1;
        bd_note_kernel("conv_wgrad3x3_ring_kernel");
        bd_wgrad3x3r_launch(d, x, g, (float*)ws, dbias ? (float*)extra : nullptr, &splitsr, (hipStream_t)stream);
        BD_CHECK_LAUNCH("bd_conv2d_wgrad(3x3 ring)");
        BdRedEntry e{};
        bd_wgrad3x3r_entry(d, &e);
        e.slab = (const float*)ws; e.splits = splitsr; e.row_scale = row_scale; e.dw = dw; e.accumulate = accumulate;
        return finish(e, splitsr, true);
    }
    if (g_wgrad_use_3x3 && is_3x3s1(d)) {
        int splits3 = 1;
        bd_note_kernel("conv_wgrad3x3_kernel");
        bd_wgrad3x3_launch(d, x, g, (float*)ws, dbias ? (float*)extra : nullptr, &splits3, (hipStream_t)stream);
        BD_CHECK_LAUNCH("bd_conv2d_wgrad(3x3)");
        return finish(plain_entry((const float*)ws, splits3, (long long)d->Cout * 9 * d->Cin, 9 * d->Cin, row_scale, dw, accumulate), splits3, true);
    }
    if (g_wgrad_use_3x3 && g_wgrad_use_ring && bd_wgrad1x1r_eligible(d)) {
        int splitsr = 1;
        bd_note_kernel("conv_wgrad1x1_ring_kernel");
        bd_wgrad1x1r_launch(d, x, g, (float*)ws, &splitsr, (hipStream_t)stream);
        BD_CHECK_LAUNCH("bd_conv2d_wgrad(1x1 ring)");
        BdRedEntry e{};
        bd_wgrad1x1r_entry(d, &e);
        e.slab = (const float*)ws; e.splits = splitsr; e.row_scale = row_scale; e.dw = dw; e.accumulate = accumulate;
        return finish(e, splitsr, false);
    }
    if (g_wgrad_use_3x3 && d->R == 1 && d->S == 1 && d->pad == 0) {
        int splits1 = 1;
        bd_note_kernel("conv_wgrad1x1_kernel");
        bd_wgrad1x1_launch(d, x, g, (float*)ws, &splits1, (hipStream_t)stream);
        BD_CHECK_LAUNCH("bd_conv2d_wgrad(1x1)");
        return finish(plain_entry((const float*)ws, splits1, (long long)d->Cout * d->Cin, d->Cin, row_scale, dw, accumulate), splits1, false);
    }
    WgradParams p{};
    p.x = (const bf16_raw*)x; p.g = (const bf16_raw*)g; p.slab = (float*)ws;
    p.Cin = d->Cin; p.Cout = d->Cout; p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad = d->pad;
    p.M = (int)pl.M; p.nseg = d->nseg;
    p.in_pix_per_img = d->in_pix_per_img; p.out_pix_per_img = d->out_pix_per_img;
    p.ci_tiles = pl.ci_tiles; p.co_tiles = pl.co_tiles; p.splits = pl.splits;
    p.steps_per_split = pl.steps_per_split; p.total_steps = pl.total_steps;
    long long m = 0;
    for (int s = 0; s < d->nseg; ++s) {
        WSeg& sg = p.seg[s];
        sg.m_start = (int)m;
        sg.Ho = d->Ho[s]; sg.Wo = d->Wo[s]; sg.per_img = d->Ho[s] * d->Wo[s];
        sg.inv_per_img = 1.0f / (float)sg.per_img; sg.inv_wo = 1.0f / (float)sg.Wo;
        sg.Hi = d->Hi[s]; sg.Wi = d->Wi[s]; sg.in_off = d->in_off[s]; sg.out_off = d->out_off[s];
        m += (long long)d->N * sg.per_img;
    }
    const int grid = pl.splits * d->R * d->S * pl.ci_tiles * pl.co_tiles;
    const size_t lds = 4 * BKP_DEFAULT * ROW_PITCH;
    BD_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<BKP_DEFAULT, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<BKP_DEFAULT, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    bd_note_kernel("conv_wgrad_kernel");
    if (g_wgrad_use_tr)
        hipLaunchKernelGGL((conv_wgrad_kernel<BKP_DEFAULT, true>), dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<BKP_DEFAULT, false>), dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
    BD_CHECK_LAUNCH("bd_conv2d_wgrad");
    const long long n = (long long)d->Cout * d->R * d->S * d->Cin;
    return finish(plain_entry((const float*)ws, pl.splits, n, d->R * d->S * d->Cin, row_scale, dw, accumulate), pl.splits, false);
}
