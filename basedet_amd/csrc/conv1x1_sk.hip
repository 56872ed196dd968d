// Dense 1x1 convolution (forward / data gradient) = GEMM  D[m][co] = sum_k X[m][k] * W[co][k]  on the 256 x 256 tile with a FIXED-ORDER
// K-SLICED schedule ("stream-K"), for the long-K / small-M launches whose tile count does not fill the chip: res5's conv1 (2048 -> 512 over
// 16 800 pixels: 132 tiles of 64 K steps for 256 CUs), conv3's data gradient, FPN lateral 5, res4's 263-tile layers
// (models/cls/resnet.py:70-113, layers/backbone/fpn_backbone.py:61-76).
//
// The K loop is conv1x1_big_kernel's (conv1x1.hip): 8 waves = 2 channel halves x 4 pixel quarters, wave tile 128 x 64 (acc[8][4] of
// v_mfma_f32_16x16x32_bf16), both operands by LDS-DMA into a four-stage ring of 32-channel K steps (swizzle and channel permutation on the
// SOURCE address), fragment reads of step s + 1 under the MFMAs of step s.  What is new is the schedule:
//   * the launch's work is the flat sequence of (tile, K step) UNITS, tile-major; workgroup g of G (one per CU) owns the contiguous range
//     [g U / G, (g + 1) U / G): every CU gets the same number of K steps (+- 1) whatever the tile count.  The ring runs on across the
//     range -- tile changes and slice ends cost no prologue;
//   * a range that covers a whole tile ends in the normal fused epilogue.  A tile cut by range boundaries is a set of SLICES, numbered by
//     their K position; a slice's fp32 partial sums (256 KB) either leave through a slab in the workspace or stay in registers:
//       - every slice takes a ticket (one agent-scope atomic on the tile's counter) when its K loop is done;
//       - a slice that is not the last to arrive writes its accumulators to its slab (16-byte sc1 stores, 1 KiB contiguous per wave
//         instruction), every wave waits for its stores (vmcnt(0)), the workgroup's barrier, then ONE lane adds to the tile's second
//         counter -- and the workgroup moves on: a producer never waits;
//       - the LAST arriver waits until the second counter says every other slice is written (it only ever waits for workgroups that
//         already run and have nothing left to do but store: no residency assumption, no deadlock), reads their slabs with sc1 loads and
//         adds the slices IN SLICE ORDER -- its own from registers at its own position -- so the sum does not depend on who arrived
//         last: bitwise reproducible.  Then the normal epilogue, and it clears the tile's two counters for the next launch.
//     (Hand-off form: MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility", measured row 3: sc1 payload
//     stores whole 128-byte lines per instruction, drained per wave, barrier, one lane's atomic add; sc1 poll, barrier, sc1 loads.)
// Workspace (caller-owned, zero-initialised ONCE by the caller; the kernel leaves the counters at zero): [2 x tiles] counters, then two
// slab slots per workgroup (a range starts with at most one tail slice and ends with at most one head slice).
#include "common.h"

namespace {

constexpr int SK_T = 256, SK_BK = 32;
constexpr int SK_HALF = SK_T * 64;             // one operand tile of a stage: 256 rows x 64 B
constexpr int SK_STAGE = 2 * SK_HALF;          // 32768
constexpr int SK_NSTAGE = 4;
constexpr int SK_RING = SK_NSTAGE * SK_STAGE;  // 131072
constexpr int SK_LDS = SK_RING + 64;           // + the ticket word
constexpr int SK_SLAB_FLOATS = SK_T * SK_T;    // 65536 fp32 = 256 KB per slice
constexpr int SK_MAX_TILES = 16384;            // counters: 2 x 16384 x 4 B = 128 KB at the head of the workspace
constexpr size_t SK_CNT_BYTES = 2ull * SK_MAX_TILES * 4;

struct SkParams {
    const bf16_raw* x;
    const bf16_raw* w;
    const float* bias;
    const bf16_raw* add;
    const bf16_raw* mask;
    const unsigned* maskbits;
    bf16_raw* y;
    unsigned* ybits;
    unsigned* counters;       // [2][SK_MAX_TILES]: tickets, slabs written
    float* slab;              // [2 * grid][SK_SLAB_FLOATS]
    int M, CK, CO, flags;
    unsigned x_bytes, w_bytes;
    int n_tiles;              // channel tiles per pixel tile (tile id = pixel tile * n_tiles + channel tile)
    int nsteps;               // K steps per tile
    int units;                // tiles * nsteps (< 2^31: checked by the launcher)
};

typedef __attribute__((address_space(3))) void lds_void_sk_t;

__device__ __forceinline__ int sk_lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 3)) << 4); }

// 16-byte sc1 (write-through / L1-bypassing) accesses of the slabs: the hand-off form the microarchitecture guide measured
__device__ __forceinline__ void sk_store_sc1(float* p, f32x4_t v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4_t sk_load_sc1(const float* p) {
    f32x4_t v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// the compiler does not know that an inline-assembly load is asynchronous: the wait names the loaded registers as read-write operands, so
// that every use of them is ordered behind it
__device__ __forceinline__ void sk_wait_loads(f32x4_t (&ld)[8]) {
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(ld[0]), "+v"(ld[1]), "+v"(ld[2]), "+v"(ld[3]), "+v"(ld[4]), "+v"(ld[5]), "+v"(ld[6]), "+v"(ld[7]) : : "memory");
}

__global__ __launch_bounds__(512, 1) void conv1x1_sk_kernel(const SkParams p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2;       // channel half (128 rows)
    const int wp = wave & 3;        // pixel quarter (64 pixels)
    const int G = gridDim.x;
    // consecutive logical ids on ONE XCD (blocks b and b + 8 share one): neighbouring ranges share tiles, and the channel tiles of a pixel
    // tile share its activation rows in that XCD's L2
    int lid = blockIdx.x;
    {
        const int q = G >> 3, r = G & 7, xcd = lid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lid >> 3);
    }
    const int S = p.nsteps;
    const int u0 = (int)((long long)lid * p.units / G), u1 = (int)((long long)(lid + 1) * p.units / G);
    int* ticket_lds = reinterpret_cast<int*>(smem + SK_RING);

    constexpr unsigned X_NONE = 0x80000000u;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.w), 0, p.w_bytes, 0x00020000);

    // ---- producer: unit pu (K step p_step of tile p_tile) goes to ring stage (flat index & 3).  An operand tile of a stage = 16 pieces of
    // 1 KiB (16 rows x 64 B); this wave owns pieces wave and wave + 8 of both; lane -> row lane >> 2, position lane & 3, source chunk =
    // position ^ ((row >> 1) & 3).  Past the end of the range the offsets are X_NONE (zeros land in a stage nobody reads): the number of DMA
    // instructions in flight stays what the counted waits assume.
    unsigned a_src[2], b_src[2];
    int pu = u0;
    int p_tile = u0 / S, p_step = u0 - p_tile * S, pflat = 0;
    auto producer_tile = [&](int t, bool live) {
        const int tm = t / p.n_tiles, tn = t - tm * p.n_tiles;
        int lq = lane;              // (an opaque copy: the row constants below are recomputed at a tile change -- ~20 instructions -- instead of
        asm volatile("" : "+v"(lq));        // being kept, i.e. spilled, across the MFMA loop)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int lrow = 16 * (wave + 8 * k) + (lq >> 2);
            const int chunk = (lq & 3) ^ ((lrow >> 1) & 3);
            const int rho = lrow & 15;
            const int co = tn * SK_T + (lrow & 192) + 32 * ((lrow >> 5) & 1) + 8 * (rho >> 2) + 4 * ((lrow >> 4) & 1) + (rho & 3);
            a_src[k] = (live && co < p.CO) ? (unsigned)(co * p.CK + chunk * 8) * 2u : X_NONE;
            const int m = tm * SK_T + lrow;
            b_src[k] = (live && m < p.M) ? (unsigned)(m * p.CK + chunk * 8) * 2u : X_NONE;
        }
    };
    auto produce = [&]() {
        int so = p_step * (SK_BK * 2);
        asm volatile("" : "+s"(so));
        unsigned char* At = smem + (pflat & (SK_NSTAGE - 1)) * SK_STAGE;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void_sk_t*)(At + (wave + 8 * k) * 1024), 16, a_src[k], so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void_sk_t*)(At + SK_HALF + (wave + 8 * k) * 1024), 16, b_src[k], so, 0, 0);
        }
        ++pflat;
        ++pu;
        if (++p_step == S) { p_step = 0; ++p_tile; producer_tile(p_tile, pu < u1); }
        else if (pu == u1) producer_tile(p_tile, false);
    };
    producer_tile(p_tile, pu < u1);
#pragma unroll
    for (int u = 0; u < SK_NSTAGE - 1; ++u) produce();

    const int frow = lane & 15, fchunk = lane >> 4;
    const int cg = lane >> 4;
    const bool do_relu = p.flags & BD_EPI_RELU;
    const bool add_before = (p.flags & BD_EPI_ADD_BEFORE) && p.add;
    const bool add_after = (p.flags & BD_EPI_ADD_AFTER) && p.add;
    const bool want_add = add_before || add_after;
    const bool mask_bf = (p.flags & BD_EPI_MASK) && p.mask;
    const bool mask_bits = (p.flags & BD_EPI_MASK) && p.maskbits && !p.mask;
    int cflat = 0;
    int u = u0;
    while (u < u1) {
    const int tile = u / S;
    const int s0 = u - tile * S;
    const int nst = (u1 - u) < (S - s0) ? (u1 - u) : (S - s0);       // K steps of this item
    const int tile_m = tile / p.n_tiles;
    const int tile_n = tile - tile_m * p.n_tiles;
    const int m0 = tile_m * SK_T;
    const int co0 = tile_n * SK_T;
    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // (K loop: see conv1x1_big_kernel -- the first half of a step's MFMAs runs over the reads of the second half's channel fragments, the
    // second over the reads of the next step's pixel fragments and first channel half)
    {
        bf16x8_t fa_lo[4], fa_hi[4], fb[2][4];
        auto stage_of = [&](int flat) { return smem + (flat & (SK_NSTAGE - 1)) * SK_STAGE; };
        auto read_b = [&](int flat, int set) {
            const unsigned char* Bt = stage_of(flat) + SK_HALF;
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[set][j] = *reinterpret_cast<const bf16x8_t*>(Bt + sk_lds_off(wp * 64 + j * 16 + frow, fchunk));
        };
        auto read_a = [&](int flat, int half, bf16x8_t (&f)[4]) {
            const unsigned char* At = stage_of(flat);
#pragma unroll
            for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const bf16x8_t*>(At + sk_lds_off(wm * 128 + (4 * half + i) * 16 + frow, fchunk));
        };
        auto step_sync = [&]() {
            // stages cflat', +1, +2 are in flight (4 DMA instructions each; slab / epilogue loads and stores are younger and only make the
            // wait stricter): the oldest has landed when at most 8 are outstanding.  lgkmcnt: this wave's earlier fragment reads have
            // returned, so after the barrier the stage they came from may be overwritten
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
            produce();
        };
        auto mfmas = [&](int half, const bf16x8_t (&f)[4], int set) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[4 * half + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[i], fb[set][j], acc[4 * half + i][j], 0, 0, 0);
        };
        auto one_step = [&](int t, int set) {             // set: compile-time at both call sites
            read_a(cflat + t, 1, fa_hi);
            mfmas(0, fa_lo, set);
            if (t + 1 < nst) {
                step_sync();
                read_b(cflat + t + 1, set ^ 1);
                read_a(cflat + t + 1, 0, fa_lo);
            }
            mfmas(1, fa_hi, set);
        };
        step_sync();
        read_b(cflat, 0);
        read_a(cflat, 0, fa_lo);
        for (int t = 0; t < nst; t += 2) {               // two steps per trip: the pixel-fragment sets are compile-time register names
            one_step(t, 0);
            if (t + 1 < nst) one_step(t + 1, 1);
        }
        cflat += nst;
    }
    u += nst;

    // ---- a slice of a cut tile: ticket, then either leave through the slab or gather the other slices in slice order ----
    if (nst != S) {
        // which workgroups hold this tile's slices: g(v) = the range that contains unit v = ceil((v + 1) G / U) - 1
        const long long t_lo = (long long)tile * S, t_hi = t_lo + S - 1;
        const int g_first = (int)(((t_lo + 1) * G + p.units - 1) / p.units) - 1;
        const int g_last = (int)(((t_hi + 1) * G + p.units - 1) / p.units) - 1;
        const int n_sl = g_last - g_first + 1;
        const int my_q = lid - g_first;
        unsigned* arrive = p.counters + tile;
        unsigned* written = p.counters + SK_MAX_TILES + tile;
        // slab slot of workgroup g for this tile: 2 g if the tile is the first of g's range, else 2 g + 1
        auto slot_of = [&](int g) {
            const int gu0 = (int)((long long)g * p.units / G);
            return 2 * g + (gu0 / S == tile ? 0 : 1);
        };
        if (tid == 0) ticket_lds[0] = (int)__hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        const int ticket = ticket_lds[0];
        __syncthreads();                       // (the word is rewritten by the next cut tile)
        if (ticket != n_sl - 1) {
            float* out = p.slab + ((size_t)slot_of(lid) * 8 + wave) * (32 * 256) + lane * 4;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) sk_store_sc1(out + (i * 4 + j) * 256, acc[i][j]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(written, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        // last arriver: every other slice is past its K loop and only has its stores left
        if (tid == 0) {
            while (__hip_atomic_load(written, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)(n_sl - 1)) __builtin_amdgcn_s_sleep(4);
        }
        __syncthreads();
        // sum in slice order, eight accumulator quads at a time -- four round trips per slice, the K loop's
        // fragment registers are dead here: prefix = s_0 + ... + s_{q-1} from the slabs, then + own, then + the rest
#ifndef BD_SK_ABLATE          // (-DBD_SK_ABLATE=1: timing only -- the last arriver skips the slab reads: what a free fix-up would be worth)
#pragma unroll                 // (compile-time accumulator names: a run-time index would put the accumulators in scratch)
        for (int b = 0; b < 4; ++b) {
            f32x4_t pre[8];
#pragma unroll 1
            for (int q = 0; q < my_q; ++q) {
                const float* in = p.slab + ((size_t)slot_of(g_first + q) * 8 + wave) * (32 * 256) + lane * 4 + b * 8 * 256;
                f32x4_t ld[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) ld[e] = sk_load_sc1(in + e * 256);
                sk_wait_loads(ld);
#pragma unroll
                for (int e = 0; e < 8; ++e) pre[e] = q == 0 ? ld[e] : pre[e] + ld[e];
            }
            if (my_q > 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[2 * b + (e >> 2)][e & 3] = pre[e] + acc[2 * b + (e >> 2)][e & 3];
            }
#pragma unroll 1
            for (int q = my_q + 1; q < n_sl; ++q) {
                const float* in = p.slab + ((size_t)slot_of(g_first + q) * 8 + wave) * (32 * 256) + lane * 4 + b * 8 * 256;
                f32x4_t ld[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) ld[e] = sk_load_sc1(in + e * 256);
                sk_wait_loads(ld);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[2 * b + (e >> 2)][e & 3] = acc[2 * b + (e >> 2)][e & 3] + ld[e];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        if (tid == 0) { *arrive = 0u; *written = 0u; }          // nobody touches them again in this launch: ready for the next one
    }

    // ---- epilogue (conv1x1_big_kernel's): lane group cg holds channels cbase + 32 h + 0..7 (h = 0..3) of pixel m0 + wp*64 + j*16 + (lane & 15).
    // Two halves (h pairs); a half first REQUESTS all of its operands, then computes and stores
    const int cbase = co0 + wm * 128 + 8 * cg;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        u32x4_t e_aux[8];             // the residual; or the bf16 mask when there is no residual (both: the mask is read in place)
        unsigned e_bits[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int h = 2 * hh + (q >> 2), jq = q & 3;
            const int m = m0 + wp * 64 + jq * 16 + (lane & 15);
            const bool ok = m < p.M && cbase + 32 * h < p.CO;
            const long long idx = (long long)m * p.CO + cbase + 32 * h;
            e_aux[q] = (u32x4_t){0u, 0u, 0u, 0u}; e_bits[q] = 0u;
            if (ok && want_add) e_aux[q] = *reinterpret_cast<const u32x4_t*>(p.add + idx);
            else if (ok && mask_bf) e_aux[q] = *reinterpret_cast<const u32x4_t*>(p.mask + idx);
            if (ok && mask_bits) e_bits[q] = p.maskbits[(long long)((co0 + wm * 128 + 32 * h) >> 5) * p.M + m];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int h = 2 * hh + (q >> 2), jq = q & 3;
            float bias[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) bias[k] = 0.f;
            if (p.bias && cbase + 32 * h < p.CO) {
                const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(p.bias + cbase + 32 * h);
                const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(p.bias + cbase + 32 * h + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) { bias[k] = b0[k]; bias[4 + k] = b1[k]; }
            }
            const int m = m0 + wp * 64 + jq * 16 + (lane & 15);
            const bool ok = m < p.M && cbase + 32 * h < p.CO;      // CO % 8 == 0
            const long long idx = (long long)m * p.CO + cbase + 32 * h;
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = acc[2 * h + (k >> 2)][jq][k & 3] + bias[k];
            const u32x4_t av = e_aux[q];
            if (add_before) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[2 * k] += bf_lo(av[k]); v[2 * k + 1] += bf_hi(av[k]); }
            }
            if (do_relu) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
            }
            if (mask_bf) {
                u32x4_t mv = e_aux[q];
                if (want_add) { mv = (u32x4_t){0u, 0u, 0u, 0u}; if (ok) mv = *reinterpret_cast<const u32x4_t*>(p.mask + idx); }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (!(bf_lo(mv[k]) > 0.f)) v[2 * k] = 0.f;
                    if (!(bf_hi(mv[k]) > 0.f)) v[2 * k + 1] = 0.f;
                }
            }
            if (mask_bits) {
                const unsigned byte = e_bits[q] >> (8 * cg);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (!((byte >> k) & 1u)) v[k] = 0.f;
            }
            if (add_after) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[2 * k] += bf_lo(av[k]); v[2 * k + 1] += bf_hi(av[k]); }
            }
            u32x4_t o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = pack_bf2(v[2 * k], v[2 * k + 1]);
            if (ok) *reinterpret_cast<u32x4_t*>(p.y + idx) = o;
            if (p.ybits) {
                unsigned byte = 0u;
#pragma unroll
                for (int k = 0; k < 8; ++k) byte |= (v[k] > 0.f ? 1u : 0u) << k;
                unsigned word = byte << (8 * cg);
                word |= __shfl_xor(word, 16, 64);
                word |= __shfl_xor(word, 32, 64);
                if (ok && cg == (jq & 3)) p.ybits[(long long)((co0 + wm * 128 + 32 * h) >> 5) * p.M + m] = word;
            }
        }
    }
    }   // items
    // the producer ran three (empty) stages past the end of the range: let those DMA writes land before the LDS can be handed to another workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int g_sk_mode = 1;          // 0 = never, 1 = where the schedule pays (sk_wanted below), 2 = every legal launch (tests / A-B)

// Where the K-sliced schedule is taken by default: long K, and a tile count that leaves a large part of the chip idle in its last round
// (profiles/r06_dense1x1_sk.txt: the launches it wins are the ones below; elsewhere the 128 x 128 kernels' occupancy hides more)
bool sk_wanted(long long M, int CK, int CO, int cus) {
    const long long tiles = cdiv64(M, SK_T) * cdiv(CO, SK_T);
    if (CK < 1024) return false;
    const long long rounds = cdiv64(tiles, cus);
    const double fill = (double)tiles / (double)(rounds * cus);          // how full the last-round-padded grid is
    return fill < 0.80;
}

}  // namespace

void bd_conv1x1_sk_set_mode(int mode) { g_sk_mode = mode; }       // conv1x1.hip: bd_conv_set_dense1x1(7) = 2, otherwise 1; (8) = 0

// would a launch of this shape take the K-sliced kernel, given a workspace (bit-mask side operands: Cout % 32 is the caller's business)
bool bd_conv1x1_sk_applies(long long M, int CK, int CO) {
    if (g_sk_mode == 0) return false;
    if (CK % SK_BK != 0 || CO < SK_T || CO % 8 != 0) return false;
    if (M * CK * 2 >= 0x7fffffffll || (long long)CO * CK * 2 >= 0x7fffffffll || M >= (1ll << 24)) return false;
    const long long tiles = cdiv64(M, SK_T) * cdiv(CO, SK_T);
    if (tiles > SK_MAX_TILES || tiles * (CK / SK_BK) >= 0x7fffffffll) return false;
    return g_sk_mode == 2 || sk_wanted(M, CK, CO, bd_num_cus());
}

size_t bd_conv1x1_sk_ws_bytes() {
    return SK_CNT_BYTES + (size_t)2 * bd_num_cus() * SK_SLAB_FLOATS * 4;
}

// 0 = launched, 1 = not taken (the caller goes on to the other dense 1x1 kernels)
int bd_conv1x1_sk_launch(const void* x, const void* w, const float* bias, const void* add, const void* mask, const unsigned* maskbits, void* y,
                         unsigned* ybits, long long M, int CK, int CO, int flags, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (g_sk_mode == 0 || ws == nullptr) return 1;
    const int cus = bd_num_cus();
    if (ws_bytes < SK_CNT_BYTES + (size_t)2 * cus * SK_SLAB_FLOATS * 4) return 1;
    if (CK % SK_BK != 0 || CO < SK_T || CO % 8 != 0) return 1;
    if ((maskbits || ybits) && CO % 32 != 0) return 1;
    const long long xb = M * CK * 2, wb = (long long)CO * CK * 2;
    if (xb >= 0x7fffffffll || wb >= 0x7fffffffll || M >= (1ll << 24)) return 1;
    const long long tiles = cdiv64(M, SK_T) * cdiv(CO, SK_T);
    if (tiles > SK_MAX_TILES) return 1;
    if (g_sk_mode == 1 && !sk_wanted(M, CK, CO, cus)) return 1;
    SkParams p{};
    p.x = (const bf16_raw*)x; p.w = (const bf16_raw*)w; p.bias = bias; p.add = (const bf16_raw*)add; p.mask = (const bf16_raw*)mask;
    p.maskbits = maskbits; p.y = (bf16_raw*)y; p.ybits = ybits;
    p.counters = (unsigned*)ws; p.slab = (float*)((unsigned char*)ws + SK_CNT_BYTES);
    p.M = (int)M; p.CK = CK; p.CO = CO; p.flags = flags;
    p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
    p.n_tiles = cdiv(CO, SK_T); p.nsteps = CK / SK_BK;
    if (tiles * p.nsteps >= 0x7fffffffll) return 1;
    p.units = (int)(tiles * p.nsteps);
    // one workgroup per CU; never more workgroups than there are pairs of K steps (a range of one step would be all hand-off)
    int grid = cus;
    if (grid > p.units / 2) grid = p.units / 2 > 0 ? p.units / 2 : 1;
    BD_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_sk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS));
    bd_note_kernel("conv1x1_sk_kernel");
    hipLaunchKernelGGL(conv1x1_sk_kernel, dim3(grid), dim3(512), SK_LDS, stream, p);
    return 0;
}
