// Weight gradient of 1x1 / stride-1 convolutions over one dense level, round-4 form: dW[co][ci] = sum_pix G[pix][co] * X[pix][ci] with
// the operands streamed global -> LDS by LDS-DMA into a four-stage ring (conv_wgrad3x3_ring.hip is the 3x3 sibling).
//
// The layer is HBM-bound (its reduction, 67 200 .. 268 800 pixels, dwarfs its output): what counts is bytes in flight per CU and how
// little else the loop does.  conv_wgrad1x1.hip stages through registers two K steps ahead (64 KB requested per CU, less in effect: each
// step's ds_write pass waits for the data) and ran at 0.37 of the HBM roof with the waves parked half of the time.  Here a K step (32
// pixels x (TCI + TCO) channels: 32 KB) is requested THREE steps ahead with `buffer_load ... lds` -- 96 KB in flight per CU, no staging
// registers, no ds_write pass -- and a step is one counted vmcnt, one barrier, 24 transposing reads and 32 MFMAs per wave.
//   * rows stay unpadded (a DMA instruction writes 1 KiB contiguously); the 32-byte pairs of a row are XOR-swizzled with (row & 7) on the
//     SOURCE address, so the eight rows a half-wave transposes fall on all 64 banks, and a fragment's address is (lane base) ^ (i << 5);
//   * one persistent eight-wave workgroup per CU walks its whole pixel range; partial sums leave as whole 1 KiB register rows
//     (slab[wg][wave][reg][lane]) and the fixed-order reduce below un-permutes them.
#include <stdlib.h>
#include "common.h"
#include "wgrad_reduce.h"

namespace {

constexpr int BKP = 32;                          // pixels per K step
constexpr int LDS_BUDGET = 160 * 1024;           // ring stages = as many as fit (at most 8); steps requested ahead = stages - 1
constexpr unsigned X_NONE = 0x80000000u;

struct R1Params {
    const bf16_raw* x;
    const bf16_raw* g;
    float* slab;             // [splits * tiles][8 waves][FI * FJ][64 lanes] f32x4
    int Cin, Cout, M;
    unsigned x_bytes, g_bytes;
    int ci_tiles, co_tiles, steps_per_split, total_steps;
};

// (conv_wgrad3x3_ring.hip: why the DMA is inline assembly -- hipcc would drain the ring in front of every ds_read_b64_tr_b16 builtin)
__device__ __forceinline__ void r1_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned lds_addr, unsigned voff, int soff) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" : : "v"(voff), "s"(lds_addr), "s"(rsrc), "s"(soff) : "memory");
}

// Tile TCI x TCO per workgroup, wave tile (16 FI ci) x (16 FJ co), eight waves.  The slabs are (number of workgroups) x (tile bytes) whatever the
// split: a launch of 256 workgroups with 256 x 256 tiles writes and re-reads 67 MB of partial sums next to 86 .. 344 MB of operands, with
// 128 x 128 tiles 17 MB -- at the price of re-reading each operand row from L2 by more tiles of the same split (the launcher picks per shape).
// (Tried and removed: an "L2 warm-up" -- one dword per 128-byte line of the rows 8 .. 32 steps ahead, requested into a landing pad nobody
// reads, to get past the ~100 KB the ring can keep in flight per CU.  Every shape got 20-70 % SLOWER (512->128 @ 100x168: 68 -> 115 us):
// a wave-instruction that touches 64 different lines is served line by line, and the rows are then fetched a second time.)
template <int TCI, int TCO, int FI, int FJ>
struct R1Cfg {
    static constexpr int XRB = TCI * 2, GRB = TCO * 2;                  // row bytes
    static constexpr int X_BYTES = BKP * XRB, G_BYTES = BKP * GRB, STAGE = X_BYTES + G_BYTES;
    static constexpr int XP = X_BYTES / 1024, GP = G_BYTES / 1024;      // 1 KiB pieces per step
    static constexpr int NP = (XP + GP) / 8;                            // pieces per wave and step
    static constexpr int NPX = XP / 8, NPG = GP / 8;
    static constexpr int WCI = TCI / (16 * FI), WCO = TCO / (16 * FJ);
    static constexpr int NSTAGE = LDS_BUDGET / STAGE < 8 ? LDS_BUDGET / STAGE : 8;
    static constexpr int DEPTH = NSTAGE - 1;                            // stage (t + DEPTH) % NSTAGE = the one step t - 1 just released
    static constexpr int REGS = FI * FJ;
    static_assert((XP + GP) % 8 == 0 && XP % 8 == 0, "X pieces and G pieces split evenly over the eight waves");
    static_assert(WCI * WCO == 8, "eight waves");
    static_assert((FI & (FI - 1)) == 0 && (FJ & (FJ - 1)) == 0 && FI <= 8 && FJ <= 8, "fragment counts: powers of two inside one 8-pair swizzle group");
    static_assert((DEPTH - 1) * NP < 64, "vmcnt field");
};

template <int TCI, int TCO, int FI, int FJ>
__global__ __launch_bounds__(512) void conv_wgrad1x1_ring_kernel(const R1Params p) {
    using C = R1Cfg<TCI, TCO, FI, FJ>;
    constexpr int XRB = C::XRB, GRB = C::GRB, X_BYTES = C::X_BYTES, STAGE = C::STAGE, NP = C::NP, NPX = C::NPX, NPG = C::NPG;
    constexpr int WCO = C::WCO, NSTAGE = C::NSTAGE, DEPTH = C::DEPTH, REGS = C::REGS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wci = wave / WCO, wco = wave - wci * WCO;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap: the tiles of one pixel split share its rows -> consecutive ids on ONE XCD
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int wg = bid;
    const int tiles = p.ci_tiles * p.co_tiles;
    const int split = bid / tiles;
    bid -= split * tiles;
    const int ci_tile = bid / p.co_tiles, co_tile = bid - ci_tile * p.co_tiles;
    const int ci0 = ci_tile * TCI, co0 = co_tile * TCO;
    const int step_begin = split * p.steps_per_split;
    int step_end = step_begin + p.steps_per_split;
    if (step_end > p.total_steps) step_end = p.total_steps;
    const int nsteps = step_end > step_begin ? step_end - step_begin : 0;

    // ---- DMA lane constants: piece pc of an operand = LDS rows (1024 / RB) pc ..., lane -> row, 16-byte position; the source chunk of
    // position pos in row r is (pos & ~15 ... ) : pair index (pos >> 1) with its low three bits XORed by (r & 7), half (pos & 1)
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) void*)smem));
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.g), 0, p.g_bytes, 0x00020000);
    unsigned xv[NPX], gv[NPG];       // byte offset of this lane's chunk in K step 0 of the tensor (X_NONE: channel tail)
    int xrow[NPX], grow[NPG];        // its row inside the step (the pixel bound check of the last step)
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
        constexpr int CPR = XRB / 16, RPP = 1024 / XRB;           // chunks per row, rows per piece
        const int pc = wave + 8 * k;
        const int r = pc * RPP + lane / CPR, pos = lane % CPR;
        const int c = 2 * (((pos >> 1) & ~7) | (((pos >> 1) & 7) ^ (r & 7))) + (pos & 1);
        xrow[k] = r;
        xv[k] = ci0 + c * 8 < p.Cin ? (unsigned)((r * p.Cin + ci0 + c * 8) * 2) : X_NONE;
    }
#pragma unroll
    for (int k = 0; k < NPG; ++k) {
        constexpr int CPR = GRB / 16, RPP = 1024 / GRB;
        const int pc = wave + 8 * k;
        int r, pos;
        if (RPP >= 1) { r = pc * RPP + lane / CPR; pos = lane % CPR; }
        else { r = pc / (GRB / 1024); pos = (pc % (GRB / 1024)) * 64 + lane; }       // rows longer than one piece (TCO = 1024: unused)
        const int c = 2 * (((pos >> 1) & ~7) | (((pos >> 1) & 7) ^ (r & 7))) + (pos & 1);
        grow[k] = r;
        gv[k] = co0 + c * 8 < p.Cout ? (unsigned)((r * p.Cout + co0 + c * 8) * 2) : X_NONE;
    }
    // request K step `step` into ring stage `stage` (dead: past this workgroup's range -- still issued, every lane out of range, so that
    // the vmcnt arithmetic is the same on every step)
    auto issue = [&](int stage, int step, bool dead) {
        const unsigned Xs = lds0 + stage * STAGE + wave * 1024, Gs = lds0 + stage * STAGE + X_BYTES + wave * 1024;
        const int m0 = step * BKP;
        const int sx = m0 * p.Cin * 2, sg = m0 * p.Cout * 2;       // the K step travels in the scalar offset
#pragma unroll
        for (int k = 0; k < NPX; ++k) r1_dma16(x_rsrc, Xs + k * 8192, (!dead & (m0 + xrow[k] < p.M)) ? xv[k] : X_NONE, sx);
#pragma unroll
        for (int k = 0; k < NPG; ++k) r1_dma16(g_rsrc, Gs + k * 8192, (!dead & (m0 + grow[k] < p.M)) ? gv[k] : X_NONE, sg);
    };

    f32x4_t acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // k -> pixel permutation (same for A and B): k = 8 g4 + j  <->  row 16 (g4 >> 1) + 4 (g4 & 1) + (j & 3) + 8 (j >> 2); a transposing read
    // takes lane 4 q + p's address as row q, channels 4 p .. 4 p + 3 of its 16-lane group
    const int g4 = lane >> 4, idx = lane & 15;
    const int tr_q = idx >> 2, tr_p = idx & 3;
    const int row_lo = 16 * (g4 >> 1) + 4 * (g4 & 1) + tr_q;
    const int r7 = row_lo & 7;                                  // (+ 8 for the second read: same key)
    // (pair index wci * FI + i: its low three bits are XORed with r7 -- the key goes in by XOR, not by addition; i occupies bits the wave
    // offset leaves clear, so fragment i's address is the lane base ^ (i << 5))
    const int xa0 = row_lo * XRB + (((wci * FI) ^ r7) << 5) + tr_p * 8;                          // fragment i: xa0 ^ (i << 5)
    const int ga0 = X_BYTES + row_lo * GRB + (((wco * FJ) ^ r7) << 5) + tr_p * 8;                // fragment j: ga0 ^ (j << 5)
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    auto tr_frag = [&](const unsigned char* a0, int hi_bytes) -> bf16x8_t {
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0 + hi_bytes));
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, v);
    };
    auto compute = [&](int stage) {
        const unsigned char* base = smem + stage * STAGE;
        bf16x8_t b[FJ], a[2];
#pragma unroll
        for (int j = 0; j < FJ; ++j) b[j] = tr_frag(base + (ga0 ^ (j << 5)), 8 * GRB);
        a[0] = tr_frag(base + xa0, 8 * XRB);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            if (i + 1 < FI) a[(i + 1) & 1] = tr_frag(base + (xa0 ^ ((i + 1) << 5)), 8 * XRB);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < FJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 1], b[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (nsteps > 0) {
#pragma unroll 1
        for (int d = 0; d < DEPTH; ++d) issue(d, step_begin + d, d >= nsteps);
        int stage = 0, fill = DEPTH;
#pragma unroll 1
        for (int t = 0; t < nsteps; ++t) {
            // this wave's pieces of step t have landed (DEPTH - 1 younger steps stay in flight); the barrier makes that true of every wave's
            // pieces, and every wave has retired its reads of step t - 1, whose stage is requested into next
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"((DEPTH - 1) * NP) : "memory");
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            issue(fill, step_begin + t + DEPTH, t + DEPTH >= nsteps);
            __builtin_amdgcn_sched_barrier(0);
            compute(stage);
            stage = stage + 1 == NSTAGE ? 0 : stage + 1;
            fill = fill + 1 == NSTAGE ? 0 : fill + 1;
        }
    }

    float* out = p.slab + ((size_t)(wg * 8 + wave) * REGS) * 256 + lane * 4;
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) *reinterpret_cast<f32x4_t*>(out + (i * FJ + j) * 256) = acc[i][j];
}

struct R1Plan { int tci, tco, fi, fj, ci_tiles, co_tiles, tiles, splits, total_steps, per; long long M; };

// Tile choice.  Estimated HBM bytes of a launch = operands once + 2 x slabs (written, then read by the reduce) with slabs = workgroups x tile
// bytes; the smaller tile wins until the re-reads of operand rows by the tiles of one split (served by that XCD's L2) outgrow it.
// BD_WGRAD1R_TILE = "tci,tco" forces a tile (measurement).
void ring1_tile(const bd_conv_desc* d, int& tci, int& tco, int& fi, int& fj) {
    static const char* env = bd_tune_env_str("BD_WGRAD1R_TILE");
    int et = 0, eo = 0;
    if (env && sscanf(env, "%d,%d", &et, &eo) == 2) { tci = et; tco = eo; }
    else {
        // measured per shape on the step's layers (scripts/micro_wgrad1x1_ring.py, kernel + reduce, us: tiles 128x128 / 128x256 / 256x256):
        //   512->128 @ 100x168: 68 / 99 / 93;  128->512: 69-76 / 75 / 99;  1024->256 @ 50x84: 56-59 / 52-56 / 56;  256->1024: 54-56 / 50-54 / 55;
        //   2048->512 @ 25x42: 53-55 / 49-53 / 53;  512->2048: 51-54 / 49-52 / 55;  512->256 @ 100x168: 105-108 / 101-109 / 104;
        //   1024->256 @ 50x84: 51-54 / 50-52 / 54;  2048->256 @ 25x42: 30-33 / 34-36 / 41
        const long long M = (long long)d->N * d->Ho[0] * d->Wo[0];
        const bool narrow = d->Cin <= 128 || d->Cout <= 128;
        const bool short_k = M < 32768 && d->Cout <= 256;            // few pixels, small result: the slabs are most of the traffic
        tci = 128; tco = (narrow || short_k) ? 128 : 256;
    }
    if (tci == 256 && tco == 256) { fi = 8; fj = 4; }
    else if (tci == 128 && tco == 512) { fi = 8; fj = 4; }
    else if (tci == 256 && tco == 128) { fi = 8; fj = 2; }
    else if (tci == 128 && tco == 256) { fi = 4; fj = 4; }
    else { tci = 128; tco = 128; fi = 4; fj = 2; }
}

R1Plan ring1_plan(const bd_conv_desc* d) {
    R1Plan pl;
    ring1_tile(d, pl.tci, pl.tco, pl.fi, pl.fj);
    pl.M = (long long)d->N * d->Ho[0] * d->Wo[0];
    pl.total_steps = (int)cdiv64(pl.M, BKP);
    pl.ci_tiles = cdiv(d->Cin, pl.tci);
    pl.co_tiles = cdiv(d->Cout, pl.tco);
    pl.tiles = pl.ci_tiles * pl.co_tiles;
    static const int target_env = bd_tune_env("BD_WGRAD1R_TARGET", 0);      // workgroups per launch (measurement knob)
    const int target = target_env > 0 ? target_env : bd_num_cus();
    int splits = target / pl.tiles;
    if (splits < 1) splits = 1;
    const int max_splits = pl.total_steps / 8 > 0 ? pl.total_steps / 8 : 1;
    if (splits > max_splits) splits = max_splits;
    pl.per = cdiv(pl.total_steps, splits);
    pl.splits = cdiv(pl.total_steps, pl.per);
    return pl;
}

template <int TCI, int TCO, int FI, int FJ>
void launch_r1(const R1Params& p, int grid, hipStream_t stream) {
    using C = R1Cfg<TCI, TCO, FI, FJ>;
    constexpr int lds = C::NSTAGE * C::STAGE;
    BD_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad1x1_ring_kernel<TCI, TCO, FI, FJ>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((conv_wgrad1x1_ring_kernel<TCI, TCO, FI, FJ>), dim3(grid), dim3(512), lds, stream, p);
}

}  // namespace

// 1x1 / stride 1 / pad 0 over ONE dense level (pixel index == GEMM row), tensors below 2 GB
bool bd_wgrad1x1r_eligible(const bd_conv_desc* d) {
    if (!(d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0 && d->nseg == 1)) return false;
    if (d->Cin % 8 || d->Cout % 8 || d->Cin < 64 || d->Cout < 64) return false;
    if (!(d->in_off[0] == 0 && d->out_off[0] == 0 && d->Hi[0] == d->Ho[0] && d->Wi[0] == d->Wo[0] &&
          d->in_pix_per_img == d->Ho[0] * d->Wo[0] && d->out_pix_per_img == d->Ho[0] * d->Wo[0])) return false;
    const long long M = (long long)d->N * d->Ho[0] * d->Wo[0];
    return M * d->Cin * 2 < 0x7fffffffll && M * d->Cout * 2 < 0x7fffffffll;
}

size_t bd_wgrad1x1r_slab_bytes(const bd_conv_desc* d, int* splits_out) {
    const R1Plan pl = ring1_plan(d);
    if (splits_out) *splits_out = pl.splits;
    return (size_t)pl.splits * pl.tiles * 8 * pl.fi * pl.fj * 1024;
}

int bd_wgrad1x1r_launch(const bd_conv_desc* d, const void* x, const void* g, float* slab, int* splits_out, hipStream_t stream) {
    const R1Plan pl = ring1_plan(d);
    R1Params p{};
    p.x = (const bf16_raw*)x; p.g = (const bf16_raw*)g; p.slab = slab;
    p.Cin = d->Cin; p.Cout = d->Cout; p.M = (int)pl.M;
    p.x_bytes = (unsigned)(pl.M * d->Cin * 2); p.g_bytes = (unsigned)(pl.M * d->Cout * 2);
    p.ci_tiles = pl.ci_tiles; p.co_tiles = pl.co_tiles; p.steps_per_split = pl.per; p.total_steps = pl.total_steps;
    const int grid = pl.splits * pl.tiles;
    if (pl.tci == 256 && pl.tco == 256) launch_r1<256, 256, 8, 4>(p, grid, stream);
    else if (pl.tci == 128 && pl.tco == 512) launch_r1<128, 512, 8, 4>(p, grid, stream);
    else if (pl.tci == 256 && pl.tco == 128) launch_r1<256, 128, 8, 2>(p, grid, stream);
    else if (pl.tci == 128 && pl.tco == 256) launch_r1<128, 256, 4, 4>(p, grid, stream);
    else launch_r1<128, 128, 4, 2>(p, grid, stream);
    *splits_out = pl.splits;
    return 0;
}

// the layout half of this layer's reduce descriptor (conv_wgrad.hip fills in pointers and launches / queues it)
void bd_wgrad1x1r_entry(const bd_conv_desc* d, BdRedEntry* e) {
    const R1Plan pl = ring1_plan(d);
    e->kind = 2; e->regs = pl.fi * pl.fj; e->co_tiles = pl.co_tiles; e->tci = pl.tci; e->tco = pl.tco; e->fi = pl.fi; e->fj = pl.fj;
    e->Cin = d->Cin; e->Cout = d->Cout; e->n4 = 8 * pl.fi * pl.fj * 64 * pl.tiles; e->row_len = 0;
}
