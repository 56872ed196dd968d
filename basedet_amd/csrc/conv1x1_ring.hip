// Dense 1x1 / stride-1 convolution (forward and data gradient) over one level, round-4 form: y[m][co] = sum_k x[m][k] w[co][k] (+ epilogue)
// as persistent workgroups with every byte they touch requested ahead of its use.  The launches: conv1 / conv3 of the reference's
// Bottleneck (basedet/models/cls/resnet.py:70-113: conv + FrozenBN (+ residual) + ReLU, and their data gradients) and the FPN laterals
// (basedet/layers/backbone/fpn_backbone.py); entered through bd_conv2d_fwd / bd_conv2d_dgrad (include/basedet_hip.h) like every convolution.
//
// Why (scripts/exp/d1_stamp.py, profiles/r04_d1_*): in conv1x1_dense_kernel (conv1x1.hip: one 128^2 tile per workgroup, 3 - 4 workgroups per
// CU) a workgroup lives 18 - 45 us -- a K step is one round trip (0.9 - 1.3 us under load), the epilogue two more -- and the launch takes
// the SUM of what its parts take alone (256->1024 + residual @ 50x84: K loop alone 58 us, epilogue traffic alone 70 us, MFMA + epilogue
// arithmetic alone 27 us, together 111 us).  Requesting the epilogue operands before the K loop, or staging 2 - 4 K steps deep, shortens a
// workgroup's life and lowers the residency by the same factor.  Here:
//   * operands by LDS-DMA into a ring (32 KB stages: 128 output channels + 128 pixels x 64 input channels, 128-byte rows) that does not stop
//     at a tile boundary: while a tile's epilogue runs, the next tile's first K steps are already landing;
//   * the epilogue operands of tile j + 1 (residual, gate bytes) are requested at the START of tile j's epilogue, a whole tile ahead; the
//     bias vector sits in LDS;
//   * the epilogue arithmetic runs on packed bf16 pairs where it can (ReLU = max with 0 as int16, gate = AND, gate bits out = int16 > 0) and
//     the variants are template instantiations: the phase is bound by the instructions the waves issue, not by what they wait for;
//   * every vector-memory instruction is inline assembly with wave-uniform instruction counts (out-of-range lanes get an offset beyond
//     num_records, never an exec mask), so that each wait is an exact `s_waitcnt vmcnt(n)`: vmcnt retires in order, and a wait for "K step s
//     has landed" must let the epilogue loads and stores issued after that step's request stay in flight.
// Where a tile's ~11 000 cycles go (scripts/exp/r1x_stamp.py, 256->1024 + residual): requesting the K steps 2 200 - 3 200 (the CU's vector
// memory pipe takes 64 B per clock: 128 KB of operands per tile, and a wave that cannot issue its DMA stalls), fragment reads + MFMAs
// 1 600 - 1 900, epilogue 3 400 - 4 700 (two waves per SIMD issuing ~250 instructions each, and 64 KB more through the same pipe), waiting
// for the ring ~400, for the epilogue operands ~300, barrier skew the rest.  32-channel K steps (64-byte rows) were 5 % slower: twice the
// barriers and ring waits per tile, half the MFMAs behind each fragment read.
// Shape (template NW): eight waves, one workgroup per CU, four ring stages (three K steps = 96 KB in flight).  (Four waves, two workgroups
// per CU, two stages each = the dense kernel made persistent: measured no faster than it, not instantiated.)  Tile, fragment and epilogue
// layouts are conv1x1_dense_kernel's (16-byte chunks XOR-swizzled on the source side, weight rows permuted so that a lane ends up with 8
// consecutive channels; the K order of the MFMAs is the same): results are bit-identical to it.
// Two hardware facts met on the way: a buffer resource with num_records = 0 is NOT range-checked (absent operands get a small window and
// out-of-range offsets instead), and a store of more than 64 bits followed within two wait states by a VALU write of its data registers
// stores the new values (the compiler's hazard recogniser does not look into inline assembly).
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

constexpr int TP = 128, TC = 128;
constexpr unsigned X_NONE = 0x80000000u;
constexpr int MAX_CO = 2048;           // bias vector in LDS behind the ring (padded to whole channel tiles)

// epilogue variants (template bits)
constexpr int E_ADD = 1, E_RELU = 2, E_MASK = 4, E_YBITS = 8;

struct PR {
    const bf16_raw* x;
    const bf16_raw* w;
    const float* bias;
    const bf16_raw* add;
    const unsigned* maskbits;
    bf16_raw* y;
    unsigned* ybits;
    int M, CK, CO;
    unsigned x_bytes, w_bytes, y_bytes, bits_bytes;
    int n_tiles, tiles, nsteps;
    // one-byte operands (F8 != 0): per-produced-channel epilogue multiplier, optional one-byte twin of y (y * q_scale: e4m3 forward, e5m2 data
    // gradient, rounded stochastically when sr_seed != 0)
    const float* wscale;
    unsigned char* y8;
    float q_scale;
    unsigned sr_seed, y8_bytes;
};

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) short s16x2_t;

// BK = input channels per K step: 64 (128-byte rows, two MFMA k-halves per step, 32 KB stages: one workgroup per CU) or 32 (64-byte rows,
// 16 KB stages: TWO eight-wave workgroups per CU, four waves per SIMD)
template <int NW, int BK>
struct Shape {
    static constexpr int RB = BK * 2;                                  // row bytes
    static constexpr int CPR = RB / 16;                                // 16-byte chunks per row
    static constexpr int KH = BK / 32;                                 // MFMA k-halves per step
    static constexpr int A_BYTES = TC * RB, STAGE = A_BYTES + TP * RB; // weight rows, then pixel rows
    static constexpr int WPX = NW / 2;             // waves along the pixels (two along the channels: 64 each)
    static constexpr int FJ = 8 / WPX;             // 16-pixel fragments per wave
    static constexpr int PK = A_BYTES / 1024 / NW; // 1 KiB pieces of each operand per wave and K step
    static constexpr int NP = 2 * PK;              // DMA instructions per wave and K step
    static constexpr int NU = 2 * FJ;              // 16-byte epilogue units per lane and tile: unit q = 2 j + half
    static constexpr int NSTAGE = 4;
    static constexpr int DEPTH = NSTAGE - 1;
    static constexpr int RING_BYTES = NSTAGE * STAGE;
    static constexpr int WGS = RING_BYTES <= 64 * 1024 ? 2 : 1;        // workgroups per CU
    // rows XOR-swizzled in 16-byte chunks with bits 1.. of the row: the 16 rows a fragment read touches per chunk column fall on all 64 banks
    static __device__ __forceinline__ int lds_off(int row, int chunk) { return row * RB + ((chunk ^ ((row >> 1) & (CPR - 1))) << 4); }
};

template <int NU>
struct ESet {                // epilogue operands and addresses of one tile, per lane
    u32x4_t aux[NU];         // residual
    unsigned bits[NU];       // this lane's gate byte
    unsigned yv[NU / 2];     // byte offset of unit (j, 0) in y / add (unit (j, 1): + 64, as the instruction offset)
    unsigned tv[NU / 2];     // F8: byte offset of this lane's 16 bytes of pixel j in the one-byte twin (twin_pair, common.h)
    unsigned bv[2];          // byte offset of this lane's byte of unit (0, half) in the gate words (unit (j, half): + 64 j)
    bool full;               // whole tile in range: the offsets as they stand; else checked per unit
    int m, c;                // first pixel / channel of the lane
};

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned lds_addr, unsigned voff, int soff) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" : : "v"(voff), "s"(lds_addr), "s"(rsrc), "s"(soff) : "memory");
}
template <int OFF>
__device__ __forceinline__ void load16(u32x4_t& dst, __amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=&v"(dst) : "v"(voff), "s"(rsrc), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void load1(unsigned& dst, __amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
    asm volatile("buffer_load_ubyte %0, %1, %2, 0 offen offset:%3" : "=&v"(dst) : "v"(voff), "s"(rsrc), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void store16(u32x4_t v, __amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
    // (more than 64 bits of store data are read over several cycles: a VALU write to them within two wait states would be stored instead)
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen offset:%3\n\ts_nop 1" : : "v"(v), "v"(voff), "s"(rsrc), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void store1(unsigned v, __amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
    asm volatile("buffer_store_byte %0, %1, %2, 0 offen offset:%3" : : "v"(v), "v"(voff), "s"(rsrc), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { static_assert(N >= 0 && N < 64, "vmcnt field"); asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

#ifdef BD_R1X_STAMP          // diagnostic build only (scripts/exp/r1x_stamp.py): s_memtime cycles per phase, per wave of one workgroup
__device__ unsigned long long g_r1x_stamp[8][8];
#define R1X_T(i) do { const unsigned long long now__ = __builtin_amdgcn_s_memtime(); st[i] += now__ - last__; last__ = now__; } while (0)
#else
#define R1X_T(i) do { } while (0)
#endif


template <int J> using IC = std::integral_constant<int, J>;

// XRES (K = 128 or 256, i.e. 2 or 4 K steps): the pixel rows of a pixel tile stay in the pixel halves of the four stages while the workgroup
// walks that pixel tile's channel tiles -- only the weights stream -- and the rows of the pixel tile after (K = 256) / after next (K = 128)
// are requested into a slot as soon as the last channel tile has consumed it.  A third less through the CU's vector memory pipe per tile.
// F8 (round 6): 0 = bf16 operands; 1 / 2 = ONE-BYTE operands (BASELINE config 5: e4m3 weights with e4m3 activations / e5m2 gradients): the same
// 128-byte rows carry 128 channels, a K step is one v_mfma_scale_f32_16x16x128_f8f6f4 per fragment pair (all block scales 2^0; 2 declares the
// pixel operand e5m2), the epilogue multiplies by the per-channel weight scale and runs in fp32 like conv1x1_fp8_kernel's (conv1x1.hip) --
// same K order, same arithmetic: the same bits -- and ALWAYS issues the NU / 2 stores of the one-byte twin (beyond num_records when the caller
// wants none: the instruction counts behind the vmcnt waits stay compile-time constants).
template <int EPI, int BK, bool XRES, int F8 = 0>
__global__ __launch_bounds__(512, (Shape<8, BK>::WGS)) void conv1x1_ring_kernel(const PR p) {
    constexpr int NW = 8;
    using S = Shape<NW, BK>;
    constexpr int RB = S::RB, CPR = S::CPR, KH = S::KH, A_BYTES = S::A_BYTES, STAGE = S::STAGE;
    constexpr int WPX = S::WPX, FJ = S::FJ, PK = S::PK, NP = S::NP, NU = S::NU, NSTAGE = S::NSTAGE, DEPTH = S::DEPTH, RING_BYTES = S::RING_BYTES;
    constexpr bool ADD = EPI & E_ADD, RELU = EPI & E_RELU, MASK = EPI & E_MASK, YBITS = EPI & E_YBITS;
    constexpr int NLOAD = (ADD ? NU : 0) + (MASK ? NU : 0);     // vector-memory instructions of one epilogue: the NEXT tile's operands ...
    constexpr int NSTORE = NU + (YBITS ? NU : 0) + (F8 ? NU / 2 : 0);      // ... then this tile's stores
    constexpr unsigned ES = F8 ? 1u : 2u;                       // bytes per operand element
    constexpr int NE = NLOAD + NSTORE;
    typedef ESet<NU> E;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WPX;    // 64-channel half of the tile
    const int wp = wave % WPX;    // block of 16 FJ pixels
    int bid = blockIdx.x;
    const int nwg = gridDim.x;
    {   // XCD-aware bijective remap: consecutive ids (neighbouring tile ranges: the same pixel rows, the same weights) share an XCD's L2
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // this workgroup's tiles: [t_begin, t_end), tile t = pixel tile t / n_tiles, channel tile t % n_tiles
    const int per = p.tiles / nwg, rem = p.tiles - per * nwg;
    const int t_begin = bid * per + (bid < rem ? bid : rem);
    const int t_end = t_begin + per + (bid < rem ? 1 : 0);
    const int nsteps = p.nsteps;
    const int n_tiles = p.n_tiles;

    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) void*)smem));
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t add_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(ADD ? p.add : p.y), 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t mb_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(MASK ? p.maskbits : (const unsigned*)p.y), 0, MASK ? p.bits_bytes : 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t yb_rsrc = __builtin_amdgcn_make_buffer_rsrc(YBITS ? p.ybits : (unsigned*)p.y, 0, YBITS ? p.bits_bytes : 4, 0x00020000);
    const bool twin = F8 && p.y8;
    const __amdgpu_buffer_rsrc_t y8_rsrc = __builtin_amdgcn_make_buffer_rsrc(twin ? p.y8 : (unsigned char*)p.y, 0, twin ? p.y8_bytes : 4, 0x00020000);

    // ---- the bias vector, padded with zeros to whole channel tiles, behind the ring
    float* bias_lds = reinterpret_cast<float*>(smem + RING_BYTES);
    for (int i = tid; i < n_tiles * TC; i += 64 * NW) bias_lds[i] = (p.bias && i < p.CO) ? p.bias[i] : 0.f;
    float* scale_lds = bias_lds + n_tiles * TC;                  // F8: the weight scales behind the bias vector
    if constexpr (F8 != 0)
        for (int i = tid; i < n_tiles * TC; i += 64 * NW) scale_lds[i] = i < p.CO ? p.wscale[i] : 0.f;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // written before this wave's first barrier (the barriers below are bare s_barrier)

    // ---- producer: K step (ptile, pstep) -> ring stage `fill`; this wave owns pieces wave + NW k (8 rows x 128 B each) of both operands
    int lrow[PK], wperm[PK];
    unsigned a_lane[PK], b_lane[PK];
#pragma unroll
    for (int k = 0; k < PK; ++k) {
        lrow[k] = (1024 / RB) * (wave + NW * k) + lane / CPR;
        const int ch = (lane % CPR) ^ ((lrow[k] >> 1) & (CPR - 1));     // lds_off on the source side
        const int rho = lrow[k] & 15;
        wperm[k] = (lrow[k] & 64) + 32 * ((lrow[k] >> 5) & 1) + 8 * (rho >> 2) + 4 * ((lrow[k] >> 4) & 1) + (rho & 3);      // LDS row -> channel of the tile
        a_lane[k] = (unsigned)(wperm[k] * p.CK) * ES + (unsigned)ch * 16u;
        b_lane[k] = (unsigned)(lrow[k] * p.CK) * ES + (unsigned)ch * 16u;
    }
    const unsigned tile_stride = (unsigned)(128 * p.CK) * ES;           // bytes between the first rows of neighbouring tiles (either operand)
    unsigned a_src[PK], b_src[PK];
    int ptile = t_begin, pstep = 0;
    int ptm = t_begin / n_tiles, ptn = t_begin - ptm * n_tiles;
    auto producer_tile = [&]() {
        const bool live = ptile < t_end;
#pragma unroll
        for (int k = 0; k < PK; ++k) {
            a_src[k] = (live && ptn * TC + wperm[k] < p.CO) ? a_lane[k] + (unsigned)ptn * tile_stride : X_NONE;
            b_src[k] = (live && ptm * TP + lrow[k] < p.M) ? b_lane[k] + (unsigned)ptm * tile_stride : X_NONE;
        }
    };
    producer_tile();
    unsigned fill_addr = lds0 + wave * 1024;
    const int tm_last = (t_end - 1) / n_tiles;
    // XRES: K step s of pixel tile tmx -> the pixel half of stage `slot`
    auto issue_x = [&](int tmx, int s, int slot) {
        const bool live = tmx <= tm_last;
#pragma unroll
        for (int k = 0; k < PK; ++k)
            dma16(x_rsrc, lds0 + slot * STAGE + A_BYTES + (wave + k * NW) * 1024,
                  (live && tmx * TP + lrow[k] < p.M) ? b_lane[k] + (unsigned)tmx * tile_stride : X_NONE, s * RB);
    };
    auto issue = [&]() {
        const int so = pstep * RB;
#pragma unroll
        for (int k = 0; k < PK; ++k) dma16(w_rsrc, fill_addr + k * NW * 1024, a_src[k], so);
        if constexpr (!XRES) {
#pragma unroll
            for (int k = 0; k < PK; ++k) dma16(x_rsrc, fill_addr + A_BYTES + k * NW * 1024, b_src[k], so);
        }
        fill_addr = fill_addr + STAGE >= lds0 + RING_BYTES ? fill_addr + STAGE - RING_BYTES : fill_addr + STAGE;
        if (++pstep == nsteps) {
            pstep = 0; ++ptile;
            if (++ptn == n_tiles) { ptn = 0; ++ptm; }
            producer_tile();
        }
    };

    // ---- epilogue addressing: lane group cg holds channels c + 32 half + 0..7 of pixel m + 16 j
    const int cg = lane >> 4, l15 = lane & 15;
    auto address = [&](E& e, int tm, int tn, bool live) {
        e.m = tm * TP + wp * (16 * FJ) + l15;
        e.c = tn * TC + wc * 64 + 8 * cg;
        e.full = live && tm * TP + TP <= p.M && tn * TC + TC <= p.CO;
        const unsigned y0 = (unsigned)(e.m * p.CO + e.c) * 2u, b0 = (unsigned)((tn * 4 + wc * 2) * p.M + e.m) * 4u + cg;
#pragma unroll
        for (int j = 0; j < FJ; ++j) e.yv[j] = y0 + (unsigned)(32 * p.CO) * j;
        e.bv[0] = b0; e.bv[1] = b0 + (unsigned)(4 * p.M);
        if constexpr (F8 != 0) {
            // twin_pair: even cg stores bytes [8 cg, 8 cg + 16) of the wave's first 32 channels, odd cg bytes [8 (cg - 1), + 16) of the second 32
            const int toff = (cg & 1) ? 32 + 8 * (cg - 1) : 8 * cg;
            const unsigned t0 = (unsigned)(e.m * p.CO + tn * TC + wc * 64 + toff);
#pragma unroll
            for (int j = 0; j < FJ; ++j) e.tv[j] = twin ? t0 + (unsigned)(16 * p.CO) * j : X_NONE;
        }
    };
    auto twin_ok = [&](const E& e, int j) { return e.m + 16 * j < p.M && (int)(e.c - 8 * cg) + ((cg & 1) ? 32 + 8 * (cg - 1) : 8 * cg) < p.CO; };
    // edge tiles: unit (j, half) in range = pixel m + 16 j and channels c + 32 half .. + 7 (CO % 8 == 0)
    auto unit_ok = [&](const E& e, int j, int half) { return e.m + 16 * j < p.M && e.c + 32 * half < p.CO; };
    auto request = [&](E& e, int tm, int tn, bool live) {             // NLOAD instructions
        address(e, tm, tn, live);
        if (e.full) {
            auto one = [&](auto J) {
                constexpr int j = decltype(J)::value;
                if constexpr (j < FJ) {
                    if constexpr (ADD) { load16<0>(e.aux[2 * j], add_rsrc, e.yv[j]); load16<64>(e.aux[2 * j + 1], add_rsrc, e.yv[j]); }
                    if constexpr (MASK) { load1<64 * j>(e.bits[2 * j], mb_rsrc, e.bv[0]); load1<64 * j>(e.bits[2 * j + 1], mb_rsrc, e.bv[1]); }
                }
            };
            one(IC<0>{}); one(IC<1>{}); one(IC<2>{}); one(IC<3>{});
        } else {
#pragma unroll
            for (int q = 0; q < NU; ++q) {
                const int j = q >> 1, half = q & 1;
                const bool ok = live && unit_ok(e, j, half);
                if constexpr (ADD) load16<0>(e.aux[q], add_rsrc, ok ? e.yv[j] + 64u * half : X_NONE);
                if constexpr (MASK) load1<0>(e.bits[q], mb_rsrc, ok ? e.bv[half] + 64u * j : X_NONE);
            }
        }
    };
    // the loaded registers are defined from here on (the compiler must not read them before the wait in front of this)
    auto settle = [&](E& e) {
#pragma unroll
        for (int q = 0; q < NU; q += 4) {
            if constexpr (ADD) asm volatile("" : "+v"(e.aux[q]), "+v"(e.aux[q + 1]), "+v"(e.aux[q + 2]), "+v"(e.aux[q + 3]));
            if constexpr (MASK) asm volatile("" : "+v"(e.bits[q]), "+v"(e.bits[q + 1]), "+v"(e.bits[q + 2]), "+v"(e.bits[q + 3]));
        }
    };

    f32x4_t acc[4][FJ];
    const int frag_row = lane & 15, frag_chunk = lane >> 4;
    const int a_off = S::lds_off(wc * 64 + frag_row, frag_chunk), b_off = A_BYTES + S::lds_off(wp * (16 * FJ) + frag_row, frag_chunk);   // + 16 rows: same swizzle key
    // k-half kh of a row = chunks 4 kh + frag_chunk: the swizzle key is the row's, so the second half sits at (offset ^ 64)
    struct Frags { bf16x8_t a[KH][4], b[KH][FJ]; };
    auto read_frags = [&](Frags& f, const unsigned char* st, const unsigned char* stx) {
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
#pragma unroll
            for (int i = 0; i < 4; ++i) f.a[kh][i] = *reinterpret_cast<const bf16x8_t*>(st + (a_off ^ (kh << 6)) + i * 16 * RB);
#pragma unroll
            for (int j = 0; j < FJ; ++j) f.b[kh][j] = *reinterpret_cast<const bf16x8_t*>(stx + (b_off ^ (kh << 6)) + j * 16 * RB);
        }
    };
    auto mfmas = [&](const Frags& f) {
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[kh][i], f.b[kh][j], acc[i][j], 0, 0, 0);
    };
    // F8: one MFMA consumes a whole 128-channel row segment; a lane's fragment = the 32 bytes at k = 32 (lane >> 4) = chunks 2 frag_chunk and
    // 2 frag_chunk + 1 of its row (operand map: scripts/exp/mfma_fp8_layout.hip) -- under the swizzle the second one sits at (offset ^ 16)
    static_assert(F8 == 0 || BK == 64, "one-byte operands: 128-byte rows");
    typedef __attribute__((ext_vector_type(8))) int i32x8_t;
    struct Frags8 { i32x8_t a[4], b[FJ]; };
    const int a_off8 = S::lds_off(wc * 64 + frag_row, 2 * frag_chunk), b_off8 = A_BYTES + S::lds_off(wp * (16 * FJ) + frag_row, 2 * frag_chunk);
    auto read_frags8 = [&](Frags8& f, const unsigned char* st, const unsigned char* stx) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(st + a_off8 + i * 16 * RB), hi = *reinterpret_cast<const u32x4_t*>(st + (a_off8 ^ 16) + i * 16 * RB);
            f.a[i] = (i32x8_t){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        }
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(stx + b_off8 + j * 16 * RB), hi = *reinterpret_cast<const u32x4_t*>(stx + (b_off8 ^ 16) + j * 16 * RB);
            f.b[j] = (i32x8_t){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        }
    };
    auto mfmas8 = [&](const Frags8& f) {
        const int one = 0x7f7f7f7f;                  // E8M0 block scales 2^0
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(f.a[i], f.b[j], acc[i][j], 0, F8 == 2 ? 1 : 0, 0, one, 0, one);
    };

    // one 16-byte unit: 8 channels of one pixel.  The arithmetic after the fp32 sums runs on the packed bf16 pairs: ReLU = max with 0 as
    // int16 (a negative bf16 is a negative int16), the gate = AND with a mask spread from the gate byte, the gate bits written = (int16 > 0)
    // (the stored bf16 value is > 0 exactly when the fp32 value is: rounding to nearest cannot reach 0 from a positive normal)
    auto pack2 = [](float a, float b) -> unsigned {
        const bf16x2_t pk = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);           // v_cvt_pk_bf16_f32
        return __builtin_bit_cast(unsigned, pk);
    };
    auto unit = [&](const E& e, int q, const f32x4_t (&bias)[4], u32x4_t& out, unsigned& byte) {
        const int j = q >> 1, half = q & 1;
        f32x4_t lo = acc[2 * half][j] + bias[2 * half], hi = acc[2 * half + 1][j] + bias[2 * half + 1];
        if constexpr (ADD) {
            const u32x4_t av = e.aux[q];
            lo += (f32x4_t){bf_lo(av[0]), bf_hi(av[0]), bf_lo(av[1]), bf_hi(av[1])};
            hi += (f32x4_t){bf_lo(av[2]), bf_hi(av[2]), bf_lo(av[3]), bf_hi(av[3])};
        }
        unsigned o[4] = {pack2(lo[0], lo[1]), pack2(lo[2], lo[3]), pack2(hi[0], hi[1]), pack2(hi[2], hi[3])};
        if constexpr (RELU) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                o[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, o[k]), (s16x2_t){0, 0}));
        }
        if constexpr (MASK) {
            const int gate = (int)e.bits[q];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned mlo = (unsigned)__builtin_amdgcn_sbfe(gate, 2 * k, 1), mhi = (unsigned)__builtin_amdgcn_sbfe(gate, 2 * k + 1, 1);
                o[k] &= (mlo & 0xffffu) | (mhi & 0xffff0000u);
            }
        }
        if constexpr (YBITS) {
            unsigned t = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s16x2_t r = __builtin_elementwise_min(__builtin_bit_cast(s16x2_t, o[k]), (s16x2_t){1, 1});
                if constexpr (!RELU) r = __builtin_elementwise_max(r, (s16x2_t){0, 0});
                t |= __builtin_bit_cast(unsigned, r) << (2 * k);
            }
            byte = t | (t >> 15);          // low byte: bit 2 k = pair k's low half, bit 2 k + 1 = its high half
        }
        out = (u32x4_t){o[0], o[1], o[2], o[3]};
    };
    // F8: the unit in fp32, in conv1x1_fp8_kernel's order of operations (scale and bias, residual, ReLU, gate, round) -- the same bits -- plus
    // the unit's eight bytes of the one-byte twin
    auto unit8 = [&](const E& e, int q, const f32x4_t (&bias)[4], const f32x4_t (&scl)[4], u32x4_t& out, unsigned& byte, u32x2_t& o8) {
        const int j = q >> 1, half = q & 1;
        const f32x4_t lo = acc[2 * half][j] * scl[2 * half] + bias[2 * half], hi = acc[2 * half + 1][j] * scl[2 * half + 1] + bias[2 * half + 1];
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if constexpr (ADD) {
            const u32x4_t av = e.aux[q];
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[2 * k] += bf_lo(av[k]); v[2 * k + 1] += bf_hi(av[k]); }
        }
        if constexpr (RELU) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        if constexpr (MASK) {
            const unsigned gate = e.bits[q];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (!((gate >> k) & 1u)) v[k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = pack_bf2(v[2 * k], v[2 * k + 1]);
        if constexpr (YBITS) {
            byte = 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) byte |= (v[k] > 0.f ? 1u : 0u) << k;
        }
        if (twin) {
            if constexpr (F8 == 2) o8 = e5m2_pair(v, p.q_scale, p.sr_seed, (long long)(e.m + 16 * j) * p.CO + e.c + 32 * half);
            else {
                o8[0] = pack4_e4m3(v[0] * p.q_scale, v[1] * p.q_scale, v[2] * p.q_scale, v[3] * p.q_scale);
                o8[1] = pack4_e4m3(v[4] * p.q_scale, v[5] * p.q_scale, v[6] * p.q_scale, v[7] * p.q_scale);
            }
        }
    };
    auto epilogue = [&](const E& e) {          // NSTORE instructions
        f32x4_t bias[4], scl[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            bias[qq] = *reinterpret_cast<const f32x4_t*>(bias_lds + e.c + 32 * (qq >> 1) + 4 * (qq & 1));
            if constexpr (F8 != 0) scl[qq] = *reinterpret_cast<const f32x4_t*>(scale_lds + e.c + 32 * (qq >> 1) + 4 * (qq & 1));
        }
        if (e.full) {
            auto one = [&](auto J) {
                constexpr int j = decltype(J)::value;
                if constexpr (j < FJ) {
                    u32x4_t o0, o1;
                    unsigned y0 = 0u, y1 = 0u;
                    if constexpr (F8 != 0) {
                        u32x2_t h0 = {0u, 0u}, h1 = {0u, 0u};
                        unit8(e, 2 * j, bias, scl, o0, y0, h0);
                        unit8(e, 2 * j + 1, bias, scl, o1, y1, h1);
                        int off;
                        const u32x4_t t16 = twin_pair(h0, h1, cg, &off);
                        store16<0>(o0, y_rsrc, e.yv[j]); store16<64>(o1, y_rsrc, e.yv[j]);
                        store16<0>(t16, y8_rsrc, e.tv[j]);
                    } else {
                        unit(e, 2 * j, bias, o0, y0);
                        unit(e, 2 * j + 1, bias, o1, y1);
                        store16<0>(o0, y_rsrc, e.yv[j]); store16<64>(o1, y_rsrc, e.yv[j]);
                    }
                    if constexpr (YBITS) { store1<64 * j>(y0, yb_rsrc, e.bv[0]); store1<64 * j>(y1, yb_rsrc, e.bv[1]); }
                }
            };
            one(IC<0>{}); one(IC<1>{}); one(IC<2>{}); one(IC<3>{});
        } else {
            u32x2_t h0 = {0u, 0u};
#pragma unroll
            for (int q = 0; q < NU; ++q) {
                const int j = q >> 1, half = q & 1;
                const bool ok = unit_ok(e, j, half);
                u32x4_t o;
                unsigned yb = 0u;
                if constexpr (F8 != 0) {
                    u32x2_t h = {0u, 0u};
                    unit8(e, q, bias, scl, o, yb, h);
                    store16<0>(o, y_rsrc, ok ? e.yv[j] + 64u * half : X_NONE);
                    if (half == 0) h0 = h;
                    else {
                        int off;
                        const u32x4_t t16 = twin_pair(h0, h, cg, &off);
                        store16<0>(t16, y8_rsrc, twin_ok(e, j) ? e.tv[j] : X_NONE);
                    }
                } else {
                    unit(e, q, bias, o, yb);
                    store16<0>(o, y_rsrc, ok ? e.yv[j] + 64u * half : X_NONE);
                }
                if constexpr (YBITS) store1<0>(yb, yb_rsrc, ok ? e.bv[half] + 64u * j : X_NONE);
            }
        }
    };

    // ---- schedule -------------------------------------------------------------------------------------------------------------------
    // Instruction order of a wave: request(first tile); DMA of steps 0 .. DEPTH-1; consumer step c = [wait for DMA(c); barrier;
    // DMA(c + DEPTH); fragment reads and MFMAs of step c]; after a tile's last step its epilogue: NE instructions (request of the NEXT
    // tile, then this tile's stores).  vmcnt retires in order, so "DMA(c) has landed" = at most the instructions issued after it are
    // outstanding: the DMAs of the DEPTH - 1 consumer steps since, plus NE for every epilogue among those steps (`ends`: bit k = consumer
    // step c - 1 - k closed a tile).
    // (Measured and dropped: reading the fragments of step c + 1 under the MFMAs of step c -- the wait moves one step earlier and the
    // launches got 20 - 50 % slower.)
#ifdef BD_R1X_STAMP
    unsigned long long st[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long st_begin = __builtin_amdgcn_s_memtime(), st_rbegin = __builtin_amdgcn_s_memrealtime();
    unsigned long long last__ = st_begin;
#endif
    // XRES: within a step the pixel rows are requested BEFORE the weights, a refill three steps at the latest before its use -- the wait for
    // the weights of that step (requested after it) covers it.  W(c) was requested in step c - DEPTH behind that step's pixel rows: the
    // instructions issued after it = the weights of DEPTH - 1 steps, the pixel rows of those among them that refilled (`xhist`), the epilogues.
    constexpr int NPW = XRES ? PK : NP;                                 // instructions of a step that every step issues
    int tm = t_begin / n_tiles, tn = t_begin - tm * n_tiles;            // the consumer's tile
    E e0, e1;
    request(e0, tm, tn, true);
    const int xper = XRES ? NSTAGE / nsteps : 1;                        // pixel tiles resident (XRES: nsteps = 2 or 4)
    if constexpr (XRES) {
#pragma unroll 1
        for (int g = 0; g < NSTAGE; ++g) issue_x(tm + g / nsteps, g % nsteps, g);
    }
#pragma unroll 1
    for (int d = 0; d < DEPTH; ++d) issue();
    const unsigned char* stage = smem;
    int xslot = 0;                                                      // XRES: the slot of the consumer's K step
    bool p_last = false; int p_tm = 0, p_s = 0, p_slot = 0;             // XRES: the previous step closed a pixel tile's walk over that slot
    unsigned ends = 0, xhist = 0;
    auto ring_wait = [&]() {
        const int k = __builtin_popcount(ends & ((1u << DEPTH) - 1));
        if constexpr (XRES) {
            const int x = __builtin_popcount(xhist & ((1u << (DEPTH - 1)) - 1));
            constexpr int B = (DEPTH - 1) * NPW;
            static_assert(B + (DEPTH - 1) * PK + 2 * NE < 64, "vmcnt field");
            if (k == 0) { if (x == 0) wait_vm<B>(); else if (x == 1) wait_vm<B + PK>(); else wait_vm<B + 2 * PK>(); }
            else if (k == 1) { if (x == 0) wait_vm<B + NE>(); else if (x == 1) wait_vm<B + NE + PK>(); else wait_vm<B + NE + 2 * PK>(); }
            else { if (x == 0) wait_vm<B + 2 * NE>(); else if (x == 1) wait_vm<B + 2 * NE + PK>(); else wait_vm<B + 2 * NE + 2 * PK>(); }
        } else {
            if (k == 0) wait_vm<(DEPTH - 1) * NP>();
            else if (k == 1) wait_vm<(DEPTH - 1) * NP + NE>();
            else if constexpr ((DEPTH - 1) * NP + 2 * NE < 64) wait_vm<(DEPTH - 1) * NP + 2 * NE>();       // (three and more: waits for the oldest)
            else wait_vm<(DEPTH - 1) * NP + NE>();
        }
    };
    auto step = [&](int s, bool tile_end, bool last_co) {
        R1X_T(0);
        ring_wait();
        R1X_T(1);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        R1X_T(2);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (XRES) {
            if (p_last) issue_x(p_tm + xper, p_s, p_slot);
            xhist = (xhist << 1) | (p_last ? 1u : 0u);
        }
        issue();
        __builtin_amdgcn_sched_barrier(0);
#ifdef BD_R1X_STAMP
        R1X_T(0);        // (diagnostic build: slot 0 = the DMA requests + producer bookkeeping, slot 3 = fragment reads + MFMAs)
#endif
        if constexpr (F8 != 0) {
            Frags8 f;
            read_frags8(f, stage, XRES ? smem + xslot * STAGE : stage);
            mfmas8(f);
        } else {
            Frags f;
            read_frags(f, stage, XRES ? smem + xslot * STAGE : stage);
            mfmas(f);
        }
        stage = stage + STAGE == smem + RING_BYTES ? smem : stage + STAGE;
        ends = (ends << 1) | (tile_end ? 1u : 0u);
        if constexpr (XRES) { p_last = last_co; p_tm = tm; p_s = s; p_slot = xslot; }
        R1X_T(3);
    };
    bool first = true;
    auto run_tile = [&](E& cur, E& nxt, int tile) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const bool last_co = tn + 1 == n_tiles || tile + 1 == t_end;          // XRES: the pixel tile's rows are read for the last time
        const int xbase = xslot;
#pragma unroll 1
        for (int t = 0; t < nsteps; ++t) {
            if constexpr (XRES) xslot = xbase + t;
            step(t, t + 1 == nsteps, last_co);
        }
        if constexpr (XRES) xslot = last_co ? ((xbase + nsteps) & (NSTAGE - 1)) : xbase;      // the next pixel tile's slots / the same rows again
        // this tile's operands: requested one epilogue ago, before that epilogue's NSTORE stores and this tile's nsteps DMA groups (at least
        // NPW instructions each)
        if constexpr (NLOAD > 0) {
            if (first) wait_vm<(DEPTH - 1) * NPW>();
            else if (nsteps >= DEPTH) wait_vm<NSTORE + DEPTH * NPW>();
            else if (nsteps >= 2) wait_vm<NSTORE + 2 * NPW>();
            else wait_vm<NSTORE + NPW>();                                   // one K step per tile (K = 64 bf16 / 128 one-byte channels)
            settle(cur);
        }
        R1X_T(4);
        if (++tn == n_tiles) { tn = 0; ++tm; }
        request(nxt, tm, tn, tile + 1 < t_end);
        epilogue(cur);
        first = false;
        R1X_T(5);
    };
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; tile += 2) {
        run_tile(e0, e1, tile);
        if (tile + 1 < t_end) run_tile(e1, e0, tile + 1);
    }
    // the ring ran DEPTH (empty) steps past the last tile, and the last request's loads target registers: let all of it land before the
    // LDS and the registers are handed to another workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef BD_R1X_STAMP
    if (blockIdx.x == 100 % gridDim.x && lane == 0) {
        for (int k = 0; k < 6; ++k) g_r1x_stamp[wave][k] = st[k];
        g_r1x_stamp[wave][6] = __builtin_amdgcn_s_memtime() - st_begin;
        g_r1x_stamp[wave][7] = ((unsigned long long)(__builtin_amdgcn_s_memrealtime() - st_rbegin) << 24) | ((unsigned long long)(t_end - t_begin) << 8) | (unsigned)nsteps;
    }
#endif
}

template <int EPI, int BK, bool XRES, int F8 = 0>
void launch_ring(PR p, hipStream_t stream) {
    using S = Shape<8, BK>;
    constexpr int VECS = F8 ? 2 : 1;                             // bias (+ weight scales) behind the ring
    p.nsteps = p.CK / (F8 ? 2 * BK : BK);
    const int lds = S::RING_BYTES + p.n_tiles * TC * 4 * VECS;
    BD_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_ring_kernel<EPI, BK, XRES, F8>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, S::RING_BYTES + MAX_CO * 4 * VECS));
    const int slots = S::WGS * bd_num_cus();
    bd_note_kernel(F8 ? "conv1x1_ring_fp8_kernel" : "conv1x1_ring_kernel");
    hipLaunchKernelGGL((conv1x1_ring_kernel<EPI, BK, XRES, F8>), dim3(p.tiles < slots ? p.tiles : slots), dim3(512), lds, stream, p);
}

}  // namespace

#ifdef BD_R1X_STAMP
extern "C" int bd_debug_r1x_stamp(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_r1x_stamp), sizeof(g_r1x_stamp)) == hipSuccess ? 0 : 1;
}
#endif

static BD_KNOB bool g_ring_everywhere = false;
void bd_conv1x1_ring_everywhere(bool on) { g_ring_everywhere = on; }          // bd_conv_desc.route[0] mode 5: tests / A-B

// Called by bd_conv1x1_dense_launch (conv1x1.hip) before its own kernels.  0 = taken.
int bd_conv1x1_ring_launch(const void* x, const void* w, const float* bias, const void* add, const void* mask, const unsigned* maskbits, void* y,
                           unsigned* ybits, void* y8, long long M, int CK, int CO, int flags, hipStream_t stream) {
    // BD_DENSE1X1_RING: 0 = every launch stays on conv1x1_dense_kernel (A/B), 2 = every legal launch comes here (also bd_conv_desc.route[0] mode 5)
    static const int env_mode = bd_tune_env("BD_DENSE1X1_RING", 1);
    const int mode = g_ring_everywhere ? 2 : env_mode;
    if (!mode) return 1;
    // measured per launch class of the step (scripts/micro_1x1_step.py 5 3, profiles/r04_dense1x1_ring.txt): every launch over the 268 800
    // pixels of res3 gains 3 - 18 %, and so do the short-K launches into >= 256 channels at 67 200 pixels (conv3 forward, conv1's data
    // gradient: 14 - 16 %); the other launches at 67 200 / 16 800 pixels have 2 - 8 tiles per CU -- one persistent workgroup per CU pays the
    // rounding (5 tiles against 4.1) that three to four small workgroups per CU even out -- and stay on conv1x1_dense_kernel (2 - 6 % slower here)
    if (mode != 2 && !(M >= 131072 || (CK <= 256 && CO >= 256 && M >= 32768))) return 1;
    if (y8 || CK % 64 != 0 || CK < (g_ring_everywhere ? 64 : 128) || CO % 8 != 0 || CO > MAX_CO) return 1;      // (K = 64, one step per tile: legal, measured below)
    if ((flags & BD_EPI_MASK) && mask && !maskbits) return 1;                   // bf16 gates: the older kernel
    if ((maskbits || ybits) && CO % 32 != 0) return 1;
    if ((flags & BD_EPI_SPARSE) || ((flags & BD_EPI_ADD_AFTER) && add)) return 1;
    const long long xb = M * CK * 2, wb = (long long)CO * CK * 2, yb = M * CO * 2, bb = (long long)(CO / 32) * M * 4;
    if (xb >= 0x7fffffffll || wb >= 0x7fffffffll || yb >= 0x7fffffffll || M >= (1ll << 24)) return 1;
    PR p{};
    p.x = (const bf16_raw*)x; p.w = (const bf16_raw*)w; p.bias = bias; p.add = (const bf16_raw*)add; p.maskbits = maskbits;
    p.y = (bf16_raw*)y; p.ybits = ybits;
    p.M = (int)M; p.CK = CK; p.CO = CO;
    p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb; p.y_bytes = (unsigned)yb; p.bits_bytes = (unsigned)bb;
    p.n_tiles = cdiv(CO, TC);
    p.tiles = (int)cdiv64(M, TP) * p.n_tiles;
    const int epi = (((flags & BD_EPI_ADD_BEFORE) && add) ? E_ADD : 0) | ((flags & BD_EPI_RELU) ? E_RELU : 0) |
                    (((flags & BD_EPI_MASK) && maskbits) ? E_MASK : 0) | (ybits ? E_YBITS : 0);
    // (a four-wave shape, two workgroups per CU, measured no faster than conv1x1_dense_kernel either)
    // pixel rows resident over a pixel tile's channel tiles: K of exactly 2 or 4 steps and at least two channel tiles (BD_DENSE1X1_XRES=0: off, A/B)
    static const int xres_on = bd_tune_env("BD_DENSE1X1_XRES", 1);
    const bool xres = xres_on && (CK == 128 || CK == 256) && p.n_tiles >= 2;
    // (Shape<8, 32> -- 32-channel K steps, 64 KB rings, TWO eight-wave workgroups per CU, four waves per SIMD -- passes the same tests and is
    // 10 % slower over the step's launches than the one-workgroup form, 3 % slower than conv1x1_dense_kernel: more waves do not help, the
    // CU's LDS and vector-memory pipes are what the waves queue for.  Not instantiated.)
    switch (epi) {
#define BD_R1X_CASE(n) case n: if (xres) launch_ring<n, 64, true>(p, stream); else launch_ring<n, 64, false>(p, stream); break;
        BD_R1X_CASE(0) BD_R1X_CASE(1) BD_R1X_CASE(2) BD_R1X_CASE(3) BD_R1X_CASE(4) BD_R1X_CASE(5) BD_R1X_CASE(6) BD_R1X_CASE(7)
        BD_R1X_CASE(8) BD_R1X_CASE(9) BD_R1X_CASE(10) BD_R1X_CASE(11) BD_R1X_CASE(12) BD_R1X_CASE(13) BD_R1X_CASE(14) BD_R1X_CASE(15)
#undef BD_R1X_CASE
    }
    return 0;
}

// One-byte form (BASELINE config 5), called by bd_conv1x1_fp8 (conv1x1.hip) before conv1x1_fp8_kernel.  0 = taken.  mode 0: xq e4m3
// activations, no gates; mode 1: xq = e5m2 gradient, no ReLU.  Same launch classes as the bf16 rule above with K counted in 128-byte steps.
int bd_conv1x1_ring_fp8_launch(int mode, const void* xq, const void* wq, const float* wscale, const float* bias, const void* add,
                               const unsigned* maskbits, void* y, unsigned* ybits, void* y8, float q_scale, unsigned sr_seed, long long M, int CK,
                               int CO, int flags, hipStream_t stream) {
    // measured per launch class of R101 at batch 32 (scripts/exp/fp8_1x1_shapes.py, profiles/r06_fp8_1x1_ring.txt): up to four K steps this form is
    // 11 - 25 % faster than conv1x1_fp8_kernel (res4 conv3 forward 195 -> 156 us, its conv1 data gradient 187 -> 167, res3's K = 512 launches
    // 135 -> 102 / 119 -> 98); at eight and sixteen steps (K = 1 024 / 2 048) it is 3 - 9 % slower and those launches stay there
    if (!g_ring_everywhere && !(CK <= 512 && (M >= 131072 || (CO >= 256 && M >= 32768)))) return 1;
    if (CK % 128 != 0 || CO % 32 != 0 || CO > MAX_CO) return 1;          // (K = 128: one step per tile -- res3 conv3 forward 353 -> 314 us)
    if ((flags & BD_EPI_SPARSE) || ((flags & BD_EPI_ADD_AFTER) && add)) return 1;
    const long long xb = M * CK, wb = (long long)CO * CK, yb = M * CO * 2, bb = (long long)(CO / 32) * M * 4;
    if (xb >= 0x7fffffffll || wb >= 0x7fffffffll || yb >= 0x7fffffffll || M >= (1ll << 24)) return 1;
    PR p{};
    p.x = (const bf16_raw*)xq; p.w = (const bf16_raw*)wq; p.bias = bias; p.add = (const bf16_raw*)add; p.maskbits = maskbits;
    p.y = (bf16_raw*)y; p.ybits = ybits;
    p.wscale = wscale; p.y8 = (unsigned char*)y8; p.q_scale = q_scale; p.sr_seed = sr_seed; p.y8_bytes = (unsigned)(M * CO);
    p.M = (int)M; p.CK = CK; p.CO = CO;
    p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb; p.y_bytes = (unsigned)yb; p.bits_bytes = (unsigned)bb;
    p.n_tiles = cdiv(CO, TC);
    p.tiles = (int)cdiv64(M, TP) * p.n_tiles;
    const int epi = (((flags & BD_EPI_ADD_BEFORE) && add) ? E_ADD : 0) | ((flags & BD_EPI_RELU) ? E_RELU : 0) |
                    (((flags & BD_EPI_MASK) && maskbits) ? E_MASK : 0) | (ybits ? E_YBITS : 0);
    if (mode == 0 ? (epi & E_MASK) : (epi & E_RELU)) return 1;
    const bool xres = (CK == 256 || CK == 512) && p.n_tiles >= 2;         // K of exactly 2 or 4 steps: the pixel rows stay resident
    switch (epi | (mode << 4)) {
#define BD_R1X8_CASE(n, f8) case (n | ((f8 - 1) << 4)): if (xres) launch_ring<n, 64, true, f8>(p, stream); else launch_ring<n, 64, false, f8>(p, stream); break;
        BD_R1X8_CASE(0, 1) BD_R1X8_CASE(1, 1) BD_R1X8_CASE(2, 1) BD_R1X8_CASE(3, 1) BD_R1X8_CASE(8, 1) BD_R1X8_CASE(9, 1) BD_R1X8_CASE(10, 1) BD_R1X8_CASE(11, 1)
        BD_R1X8_CASE(0, 2) BD_R1X8_CASE(1, 2) BD_R1X8_CASE(4, 2) BD_R1X8_CASE(5, 2) BD_R1X8_CASE(8, 2) BD_R1X8_CASE(9, 2) BD_R1X8_CASE(12, 2) BD_R1X8_CASE(13, 2)
#undef BD_R1X8_CASE
        default: return 1;
    }
    return 0;
}
