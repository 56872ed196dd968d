// Dense 1x1 convolution (forward and data gradient) = a plain GEMM  D[m][co] = sum_k X[m][k] * W[co][k]  over the pixels of ONE
// dense level (source pixel index == destination pixel index == m), with the fused epilogue of conv_igemm.hip.
//
// The ResNet bottleneck 1x1s and the FPN laterals: 40-225 FLOP per byte, the largest share of the step after the 3x3 convolutions.
// Same tile as the generic kernel (128 channels x 128 pixels, BK = 32, swizzled 64-byte LDS rows, 16x16x32 MFMA, channels on the MFMA
// row), specialised so that
//   * there is no tap / sub-segment / pixel-decode state: 121 VGPRs instead of 160 -> FOUR workgroups per CU instead of three;
//   * the ReLU mask of a data gradient may come BIT-PACKED (1 bit per element instead of a bf16: a 16x smaller stream), written by
//     the forward launch that produced the activation (ybits): bits[g * M + m], bit b = channel 32 g + b of pixel m is > 0.
// What was measured on the way (scripts/micro_1x1_step.py, bench.py --dense1x1): running the staging loads 2, 4
// or 6 K steps ahead through extra register sets -- at three or two workgroups per CU -- and requesting the epilogue operands before
// the last K step changed nothing on the 200x336 / 100x168 layers (4.2-4.8 / 3.5-3.9 TB/s either way) and LOST 10-35 % on the K >=
// 1024 layers, where the kernel is bound by its MFMA loop (one barrier per 16 MFMAs), not by memory: occupancy, not prefetch depth,
// is what fills that loop's bubbles.  (Those variants are gone; the DEPTH parameter below is what is left of them.)
#include "common.h"

namespace {

constexpr int TP = 128, TC = 128, BK = 32;
constexpr int EPI_BATCH = 4;       // epilogue units (16 B per lane) whose operands are requested together
constexpr int TILE_BYTES = 128 * 64;

struct P1 {
    const bf16_raw* x;
    const bf16_raw* w;
    const float* bias;
    const bf16_raw* add;
    const bf16_raw* mask;
    const unsigned* maskbits;
    bf16_raw* y;
    unsigned* ybits;
    unsigned char* y8;        // optional one-byte twin of y (y * q_scale) for a following fp8 launch (conv3x3_pp8.hip):
    float q_scale;            // e4m3 for activations (y8_bf8 = 0), e5m2 for gradients (y8_bf8 = 1)
    int y8_bf8;
    int M, CK, CO, flags;
    unsigned x_bytes, w_bytes;
    int m_tiles, n_tiles;
    const float* wscale;      // fp8 kernel: per-output-channel epilogue multiplier (bd_weight_pack_fp8 / _t)
    // 1x1 / stride 2 (the bottleneck shortcuts): the GEMM rows are the pixels of the SMALL grid (dW x dHW / dW per image); s2_src: the
    // source rows lie at (2y, 2x) of the big grid (forward), s2_dst: the destination / residual / gate rows do (data gradient added in
    // place at the pixels it reaches).  Mbig = pixels of the big grid (bit-packed gates are indexed by it).
    int s2_src, s2_dst, dW, dHW, bW, bHW, Mbig;
    unsigned sr_seed;         // != 0: stochastic rounding of the e5m2 twin (common.h)
    float inv_dW, inv_dHW;
#ifdef BD_D1_STAMP
    int dbg;                  // BD_D1_ABLATE (timing only): 1 no stores, 2 no epilogue operand loads, 4 every tile reads pixel tile 0, 8 no K loop
#endif
};

// row of the big grid under small-grid pixel m (m < 2^24: the float quotients are within one of the integer ones)
__device__ __forceinline__ int s2_big_row(const P1& p, int m) {
    int n = (int)((float)m * p.inv_dHW);
    int r = m - n * p.dHW;
    if (r < 0) { --n; r += p.dHW; } else if (r >= p.dHW) { ++n; r -= p.dHW; }
    int y = (int)((float)r * p.inv_dW);
    int x = r - y * p.dW;
    if (x < 0) { --y; x += p.dW; } else if (x >= p.dW) { ++y; x -= p.dW; }
    return n * p.bHW + 2 * y * p.bW + 2 * x;
}

typedef __attribute__((address_space(3))) void lds_void_1x1_t;

// 16 bytes per lane straight into LDS.  (A plain function: called with template-dependent arguments from inside the kernel template, the
// target builtin made the HOST pass drop the instantiation without a diagnostic -- undefined kernel stubs at link time.)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_1x1_t*)lds, 16, voff, soff, 0, 0);
}

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 3)) << 4); }

// DMA: the operands go by LDS-DMA (no staging registers / LDS writes) into a THREE-stage ring, two K steps in flight per workgroup under a
// counted vmcnt and one raw barrier per step (48 KB of LDS: three workgroups per CU); taken for CK % 32 == 0 and long K (see the launcher).
// NC = 64-channel wave columns: 2 = the 128-channel tile (four waves); 4 = a 256-channel x 128-pixel tile of EIGHT waves (DMA form only):
// the activation rows are fetched once per 256 output channels instead of once per 128 -- the class is bound by the bytes it moves
// between L2 and the CUs -- while the pixel granularity of the grid stays 128.
#ifdef BD_D1_STAMP          // diagnostic build only (scripts/exp/d1_stamp.py): 100 MHz stamps of every workgroup's phases + where it ran
__device__ unsigned long long g_d1_stamp[32768][4];
#define D1_T(k) do { if (threadIdx.x == 0 && blockIdx.x < 32768) g_d1_stamp[blockIdx.x][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define D1_T(k) do { } while (0)
#endif

template <int DEPTH, int WAVES, bool DMA, int NC = 2, int EBATCH = EPI_BATCH, bool EARLY = false>
__global__ __launch_bounds__(128 * NC, WAVES) void conv1x1_dense_kernel(const P1 p) {
    static_assert(NC == 2 || (NC == 4 && DMA), "the 256-channel tile exists in the LDS-DMA form only");
    constexpr int TCW = 64 * NC;                    // channels per tile
    constexpr int A_BYTES = TCW * 64;               // weight tile of one stage (64-byte rows)
    constexpr int STAGE = A_BYTES + TILE_BYTES;     // + the pixel tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wc = wave >> 1;   // channel column (64 channels)
    const int wp = wave & 1;    // pixel half
    D1_T(0);
#ifdef BD_D1_STAMP
    if (threadIdx.x == 0 && blockIdx.x < 32768)
        g_d1_stamp[blockIdx.x][3] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif

    // XCD-aware bijective remap: consecutive tile ids (the channel tiles of one pixel tile, then the next pixel tile) share an XCD's L2
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / p.n_tiles;
    const int tile_n = bid - tile_m * p.n_tiles;
    const int m0 = tile_m * TP;
    const int co0 = tile_n * TCW;

    const int chunk = tid & 3;
    const int row0 = tid >> 2;          // rows row0 and row0 + 64 of both operand tiles

    constexpr unsigned X_NONE = 0x80000000u;          // >= num_records: the buffer load returns zeros
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.w), 0, p.w_bytes, 0x00020000);
    unsigned a_voff[2], b_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        // LDS row (h*64 + t*16 + rho) of the weight tile holds output channel h*64 + 32*(t>>1) + 8*(rho>>2) + 4*(t&1) + (rho&3): after the
        // MFMAs lane group cg = rho>>2 owns channels 8*cg..8*cg+7 of each 32-channel half (16-byte stores, 64 contiguous bytes per pixel)
        const int lrow = row0 + i * 64;
        const int rho = lrow & 15;
        const int co = co0 + (lrow & 64) + 32 * ((lrow >> 5) & 1) + 8 * (rho >> 2) + 4 * ((lrow >> 4) & 1) + (rho & 3);
        a_voff[i] = co < p.CO ? (unsigned)(co * p.CK + chunk * 8) * 2u : X_NONE;
#ifdef BD_D1_STAMP
        const int m = (p.dbg & 4) ? lrow : m0 + lrow;
#else
        const int m = m0 + lrow;
#endif
        b_voff[i] = m < p.M ? (unsigned)((p.s2_src ? s2_big_row(p, m) : m) * p.CK + chunk * 8) * 2u : X_NONE;
    }
#ifdef BD_D1_STAMP
    const int nsteps = (p.dbg & 8) ? 0 : (p.CK + BK - 1) / BK;
#else
    const int nsteps = (p.CK + BK - 1) / BK;
#endif

    u32x4_t ra[DEPTH][2], rb[DEPTH][2];
    auto stage_load = [&](int step, u32x4_t (&a)[2], u32x4_t (&b)[2]) {
        int so = step * BK * 2;
        asm volatile("" : "+s"(so));                               // keep the K offset in the scalar operand
        const bool dead = step * BK + chunk * 8 >= p.CK;           // channel tail (CK % 8 == 0): zero-fill
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[i] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, dead ? X_NONE : a_voff[i], so, 0);
            b[i] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, dead ? X_NONE : b_voff[i], so, 0);
        }
    };
    auto stage_write = [&](int buf, const u32x4_t (&a)[2], const u32x4_t (&b)[2]) {
        unsigned char* At = smem + buf * STAGE;
        unsigned char* Bt = At + A_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<u32x4_t*>(At + lds_off(row0 + i * 64, chunk)) = a[i];
            *reinterpret_cast<u32x4_t*>(Bt + lds_off(row0 + i * 64, chunk)) = b[i];
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
        const unsigned char* At = smem + buf * STAGE;
        const unsigned char* Bt = At + A_BYTES;
        bf16x8_t a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(At + lds_off(wc * 64 + i * 16 + frag_row, frag_chunk));
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(Bt + lds_off(wp * 64 + j * 16 + frag_row, frag_chunk));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };

    // ---- epilogue operand addresses (lane (cg = lane>>4) holds channels cbase + 32*half + 0..7 of pixel m0 + wp*64 + j*16 + (lane&15))
    const int cg = lane >> 4;
    const int cbase = co0 + wc * 64 + 8 * cg;
    const bool do_relu = p.flags & BD_EPI_RELU;
    const bool add_before = (p.flags & BD_EPI_ADD_BEFORE) && p.add;
    const bool add_after = (p.flags & BD_EPI_ADD_AFTER) && p.add;
    const bool want_add = add_before || add_after;
    const bool mask_bf = (p.flags & BD_EPI_MASK) && p.mask;
    const bool mask_bits = (p.flags & BD_EPI_MASK) && p.maskbits && !p.mask;
    constexpr int EB = NC == 4 ? 2 : (EARLY ? 8 : EBATCH);           // the eight-wave tile runs at a 128-register budget
    u32x4_t e_aux[EB];     // the residual; or the bf16 mask when there is no residual (both: the mask is read in place)
    unsigned e_bits[EB];
    auto request = [&](int part) {
#pragma unroll
        for (int qq = 0; qq < EB; ++qq) {
            const int q = part * EB + qq;
            const int j = q >> 1, half = q & 1;
            const int m = m0 + wp * 64 + j * 16 + (lane & 15);
#ifdef BD_D1_STAMP
            const bool ok = m < p.M && cbase + 32 * half < p.CO && !(p.dbg & 2);
#else
            const bool ok = m < p.M && cbase + 32 * half < p.CO;
#endif
            const int drow = (p.s2_dst && ok) ? s2_big_row(p, m) : m;
            const long long idx = (long long)drow * p.CO + cbase + 32 * half;
            e_aux[qq] = (u32x4_t){0u, 0u, 0u, 0u}; e_bits[qq] = 0u;
            if (ok && want_add) e_aux[qq] = *reinterpret_cast<const u32x4_t*>(p.add + idx);
            else if (ok && mask_bf) e_aux[qq] = *reinterpret_cast<const u32x4_t*>(p.mask + idx);
            if (ok && mask_bits) e_bits[qq] = p.maskbits[(long long)((co0 + wc * 64 + 32 * half) >> 5) * (p.s2_dst ? p.Mbig : p.M) + drow];
        }
    };
    // EARLY: the tile's epilogue operands are requested before the K loop (they are in flight under it; the first K step waits for them)
    if constexpr (EARLY) { request(0); __builtin_amdgcn_sched_barrier(0); }
    if constexpr (DMA) {
        // ---- main loop, LDS-DMA ring: a stage = two operand tiles of 8 pieces of 1 KiB (16 rows x 64 B); this wave owns pieces wave and
        // wave + 4 of both; lane -> row lane >> 2, position lane & 3, source chunk = position ^ ((row >> 1) & 3) (lds_off on the source side)
        // weight tile: TCW / 16 pieces, this wave owns pieces wave and wave + 2 NC; pixel tile: 8 pieces, pieces wave (+ 4 with four waves)
        constexpr int NWAVE = 2 * NC, B_PIECES = 8 / NWAVE;
        unsigned a_src[2], b_src[B_PIECES];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int lrow = 16 * (wave + NWAVE * k) + (lane >> 2);
            const int ch = (lane & 3) ^ ((lrow >> 1) & 3);
            const int rho = lrow & 15;
            const int co = co0 + (lrow & (TCW - 64)) + 32 * ((lrow >> 5) & 1) + 8 * (rho >> 2) + 4 * ((lrow >> 4) & 1) + (rho & 3);
            a_src[k] = co < p.CO ? (unsigned)(co * p.CK + ch * 8) * 2u : X_NONE;
        }
#pragma unroll
        for (int k = 0; k < B_PIECES; ++k) {
            const int lrow = 16 * (wave + NWAVE * k) + (lane >> 2);
            const int ch = (lane & 3) ^ ((lrow >> 1) & 3);
#ifdef BD_D1_STAMP
            const int m = (p.dbg & 4) ? lrow : m0 + lrow;
#else
            const int m = m0 + lrow;
#endif
            b_src[k] = m < p.M ? (unsigned)((p.s2_src ? s2_big_row(p, m) : m) * p.CK + ch * 8) * 2u : X_NONE;
        }
        const int uwave = __builtin_amdgcn_readfirstlane(wave);
        auto dma = [&](int step, int stage) {
            int so = step * BK * 2;
            asm volatile("" : "+s"(so));
            unsigned char* At = smem + stage * STAGE;
#pragma unroll
            for (int k = 0; k < 2; ++k)
                dma16(w_rsrc, At + (uwave + NWAVE * k) * 1024, a_src[k], so);
#pragma unroll
            for (int k = 0; k < B_PIECES; ++k)
                dma16(x_rsrc, At + A_BYTES + (uwave + NWAVE * k) * 1024, b_src[k], so);
        };
        if (nsteps > 0) dma(0, 0);
        if (nsteps > 1) dma(1, 1);
        int cs = 0, ps = 2;                       // consumer / producer stage
        for (int t = 0; t < nsteps; ++t) {
            // stages t and t + 1 are in flight (4 DMA instructions each; 3 with eight waves): stage t has landed when at most that many are outstanding
            if (t + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NC == 2 ? 4 : 3) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");   // everyone's pieces; stage ps is free
            if (t + 2 < nsteps) dma(t + 2, ps);
            compute(cs);
            cs = cs == 2 ? 0 : cs + 1;
            ps = ps == 2 ? 0 : ps + 1;
        }
    } else {
    // ---- main loop: step t is computed from LDS buffer t & 1 while the loads of steps t+1 .. t+DEPTH are in flight ------------------
#pragma unroll
    for (int u = 0; u < DEPTH; ++u)
        if (u < nsteps) stage_load(u, ra[u], rb[u]);
    if (nsteps > 0) stage_write(0, ra[0], rb[0]);
    __syncthreads();
    for (int t0 = 0; t0 < nsteps; t0 += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const int t = t0 + u;
            if (t < nsteps) {                                       // workgroup-uniform
                // set u held step t (written to LDS in the previous iteration): it is free for step t + DEPTH
                if (t + DEPTH < nsteps) stage_load(t + DEPTH, ra[u], rb[u]);
                compute(t & 1);
                if (t + 1 < nsteps) stage_write((t + 1) & 1, ra[(u + 1) % DEPTH], rb[(u + 1) % DEPTH]);
                __syncthreads();
            }
        }
    }

    }

    // ---- epilogue -------------------------------------------------------------------------------------------------------------------
    // All of the tile's residual / mask / gate operands are REQUESTED first (8 x 16 B per lane, into the registers the staging sets and
    // fragments have left), then consumed in order: as a load - use - store chain per 16 bytes (what the compiler makes of the plain
    // loop: every unit waits on vmcnt(0), i.e. also on the previous unit's store) a wave keeps 1 KB in flight and the epilogue -- half
    // of this class's bytes -- runs at the latency of 16 dependent round trips per tile.
    D1_T(1);
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
    for (int part = 0; part < 8 / EB; ++part) {
    if constexpr (!EARLY) request(part);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int qq = 0; qq < EB; ++qq) {
        const int q = part * EB + qq;
        const int j = q >> 1, half = q & 1;
        const int m = m0 + wp * 64 + j * 16 + (lane & 15);
        {
#ifdef BD_D1_STAMP
            const bool ok = m < p.M && cbase + 32 * half < p.CO && !((p.dbg & 1) && acc[0][0][0] != 12345.f);
#else
            const bool ok = m < p.M && cbase + 32 * half < p.CO;      // CO % 8 == 0
#endif
            const int drow = (p.s2_dst && ok) ? s2_big_row(p, m) : m;
            const long long idx = (long long)drow * p.CO + cbase + 32 * half;
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = acc[2 * half + (k >> 2)][j][k & 3] + bias[8 * half + k];
            const u32x4_t av = e_aux[qq];
            if (add_before) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[2 * k] += bf_lo(av[k]); v[2 * k + 1] += bf_hi(av[k]); }
            }
            if (do_relu) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
            }
            if (mask_bf) {
                u32x4_t mv = e_aux[qq];
                if (want_add) { mv = (u32x4_t){0u, 0u, 0u, 0u}; if (ok) mv = *reinterpret_cast<const u32x4_t*>(p.mask + idx); }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (!(bf_lo(mv[k]) > 0.f)) v[2 * k] = 0.f;
                    if (!(bf_hi(mv[k]) > 0.f)) v[2 * k + 1] = 0.f;
                }
            }
            if (mask_bits) {
                const unsigned byte = e_bits[qq] >> (8 * cg);
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
            if (p.y8 && ok) {
                u32x2_t o8;
                if (p.y8_bf8) {
                    o8 = e5m2_pair(v, p.q_scale, p.sr_seed, idx);
                } else {
                    o8[0] = pack4_e4m3(v[0] * p.q_scale, v[1] * p.q_scale, v[2] * p.q_scale, v[3] * p.q_scale);
                    o8[1] = pack4_e4m3(v[4] * p.q_scale, v[5] * p.q_scale, v[6] * p.q_scale, v[7] * p.q_scale);
                }
                *reinterpret_cast<u32x2_t*>(p.y8 + idx) = o8;
            }
            if (p.ybits) {
                // the stored bf16 value is > 0 exactly when the fp32 value is (rounding to nearest cannot reach 0 from a positive normal)
                unsigned byte = 0u;
#pragma unroll
                for (int k = 0; k < 8; ++k) byte |= (v[k] > 0.f ? 1u : 0u) << k;
                unsigned word = byte << (8 * cg);
                word |= __shfl_xor(word, 16, 64);
                word |= __shfl_xor(word, 32, 64);
                if (ok && cg == (j & 3)) p.ybits[(long long)((co0 + wc * 64 + 32 * half) >> 5) * p.M + m] = word;
            }
        }
    }
    }   // parts
#ifdef BD_D1_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the stores have been acknowledged
    D1_T(2);
#endif
}

#ifdef BD_D1_STAMP
}  // namespace
extern "C" int bd_debug_d1_stamp(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_d1_stamp), (size_t)n * 32) == hipSuccess ? 0 : 1;
}
namespace {
#endif

// ---- one-byte operands (BASELINE config 5) ----------------------------------------------------------------------------------------------
// The 1x1 class is bound by the bytes it moves between L2 and the CUs (see the 256^2 section below), so with e4m3 activations /
// e5m2 gradients and e4m3 weights a K step of the same 128-byte rows carries twice the channels: the 128^2 tile of the kernel above on
// v_mfma_scale_f32_16x16x128_f8f6f4 (all block scales 2^0; MODE 1 declares the pixel operand e5m2: a data gradient), both operands by
// LDS-DMA into a two-stage ring of 32 KB stages (two workgroups per CU), per-channel weight scale in the epilogue, and the epilogue of
// the bf16 kernel (residual / bit-packed or bf16 gates / ybits / one-byte twin) with all eight units requested up front.
constexpr int F8_TILE = 128 * 128;          // one operand tile of a stage: 128 rows x 128 B
typedef __attribute__((ext_vector_type(8))) int i32x8_1x1_t;

template <int MODE>
__global__ __launch_bounds__(256, 2) void conv1x1_fp8_kernel(const P1 p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 1, wp = wave & 1;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / p.n_tiles;
    const int tile_n = bid - tile_m * p.n_tiles;
    const int m0 = tile_m * TP;
    const int co0 = tile_n * TC;
    constexpr unsigned X_NONE = 0x80000000u;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.w), 0, p.w_bytes, 0x00020000);
    // DMA pieces: an operand tile = 16 pieces of 1 KiB (8 rows x 128 B); this wave owns pieces wave + 4k; lane -> row lane >> 3,
    // position lane & 7, source chunk = position ^ (row & 7)
    unsigned a_src[4], b_src[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int lrow = 8 * (wave + 4 * k) + (lane >> 3);
        const int ch = (lane & 7) ^ (lrow & 7);
        const int rho = lrow & 15;
        const int co = co0 + (lrow & 64) + 32 * ((lrow >> 5) & 1) + 8 * (rho >> 2) + 4 * ((lrow >> 4) & 1) + (rho & 3);
        a_src[k] = co < p.CO ? (unsigned)(co * p.CK + ch * 16) : X_NONE;
        const int m = m0 + lrow;
        b_src[k] = m < p.M ? (unsigned)(m * p.CK + ch * 16) : X_NONE;
    }
    auto dma = [&](int step, int stage) {
        int so = step * 128;
        asm volatile("" : "+s"(so));
        unsigned char* At = smem + stage * 2 * F8_TILE;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void_1x1_t*)(At + (wave + 4 * k) * 1024), 16, a_src[k], so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void_1x1_t*)(At + F8_TILE + (wave + 4 * k) * 1024), 16, b_src[k], so, 0, 0);
        }
    };
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fk = lane >> 4;        // the lane's 32 consecutive k = chunks 2 fk, 2 fk + 1 of the row
    const int one = 0x7f7f7f7f;
    const int nsteps = p.CK / 128;                     // the host takes CK % 128 == 0 only
    auto frag = [&](const unsigned char* T, int row) {
        const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(T + row * 128 + (((2 * fk) ^ (row & 7)) << 4));
        const u32x4_t hi = *reinterpret_cast<const u32x4_t*>(T + row * 128 + (((2 * fk + 1) ^ (row & 7)) << 4));
        return (i32x8_1x1_t){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    };
    // epilogue operands (residual / gates) are requested before the LAST K step's matrix work: two workgroups per CU leave little else
    // to overlap their latency with
    const int cg = lane >> 4;
    const int cbase = co0 + wc * 64 + 8 * cg;
    const bool do_relu = p.flags & BD_EPI_RELU;
    const bool add_before = (p.flags & BD_EPI_ADD_BEFORE) && p.add;
    const bool add_after = (p.flags & BD_EPI_ADD_AFTER) && p.add;
    const bool want_add = add_before || add_after;
    const bool mask_bf = (p.flags & BD_EPI_MASK) && p.mask;
    const bool mask_bits = (p.flags & BD_EPI_MASK) && p.maskbits && !p.mask;
    u32x4_t e_aux[8];             // the residual; or the bf16 mask when there is no residual (both: the mask is read in place)
    unsigned e_bits[8];
    auto epi_request = [&]() {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int j = q >> 1, half = q & 1;
        const int m = m0 + wp * 64 + j * 16 + (lane & 15);
        const bool ok = m < p.M && cbase + 32 * half < p.CO;
        const long long idx = (long long)m * p.CO + cbase + 32 * half;
        e_aux[q] = (u32x4_t){0u, 0u, 0u, 0u}; e_bits[q] = 0u;
        if (ok && want_add) e_aux[q] = *reinterpret_cast<const u32x4_t*>(p.add + idx);
        else if (ok && mask_bf) e_aux[q] = *reinterpret_cast<const u32x4_t*>(p.mask + idx);
        if (ok && mask_bits) e_bits[q] = p.maskbits[(long long)((co0 + wc * 64 + 32 * half) >> 5) * p.M + m];
    }
    };
    dma(0, 0);
    for (int t = 0; t < nsteps; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                        // this wave's pieces of step t
        asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");   // everyone's; the other stage is free
        if (t + 1 < nsteps) dma(t + 1, (t + 1) & 1);
        else epi_request();
        const unsigned char* At = smem + (t & 1) * 2 * F8_TILE;
        const unsigned char* Bt = At + F8_TILE;
        i32x8_1x1_t a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = frag(At, wc * 64 + i * 16 + frow);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = frag(Bt, wp * 64 + j * 16 + frow);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i][j], 0, MODE == 1 ? 1 : 0, 0, one, 0, one);
    }

    // ---- epilogue (the bf16 kernel's, with the per-channel weight scale) ----------------------------------------------------------------
    float bias[16], scl[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { bias[k] = 0.f; scl[k] = 0.f; }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (cbase + 32 * (q >> 1) + 4 * (q & 1) < p.CO) {
            const f32x4_t sv = *reinterpret_cast<const f32x4_t*>(p.wscale + cbase + 32 * (q >> 1) + 4 * (q & 1));
            scl[4 * q] = sv[0]; scl[4 * q + 1] = sv[1]; scl[4 * q + 2] = sv[2]; scl[4 * q + 3] = sv[3];
            if (p.bias) {
                const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(p.bias + cbase + 32 * (q >> 1) + 4 * (q & 1));
                bias[4 * q] = bv[0]; bias[4 * q + 1] = bv[1]; bias[4 * q + 2] = bv[2]; bias[4 * q + 3] = bv[3];
            }
        }
    __builtin_amdgcn_sched_barrier(0);
    u32x2_t twin_h0 = {0u, 0u};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int j = q >> 1, half = q & 1;
        const int m = m0 + wp * 64 + j * 16 + (lane & 15);
        const bool ok = m < p.M && cbase + 32 * half < p.CO;      // CO % 8 == 0
        const long long idx = (long long)m * p.CO + cbase + 32 * half;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = acc[2 * half + (k >> 2)][j][k & 3] * scl[8 * half + k] + bias[8 * half + k];
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
        if (p.y8) {              // (workgroup-uniform) both halves of the pixel are packed, then the lane pairs swap: 16-byte stores
            u32x2_t o8;
            if (p.y8_bf8) {
                o8 = e5m2_pair(v, p.q_scale, p.sr_seed, idx);
            } else {
                o8[0] = pack4_e4m3(v[0] * p.q_scale, v[1] * p.q_scale, v[2] * p.q_scale, v[3] * p.q_scale);
                o8[1] = pack4_e4m3(v[4] * p.q_scale, v[5] * p.q_scale, v[6] * p.q_scale, v[7] * p.q_scale);
            }
            if (half == 0) twin_h0 = o8;
            else {
                int off;
                const u32x4_t t16 = twin_pair(twin_h0, o8, cg, &off);
                const int cch = co0 + wc * 64 + off;               // first channel of this lane's 16 bytes
                if (m < p.M && cch < p.CO) *reinterpret_cast<u32x4_t*>(p.y8 + (long long)m * p.CO + cch) = t16;
            }
        }
        if (p.ybits) {
            unsigned byte = 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) byte |= (v[k] > 0.f ? 1u : 0u) << k;
            unsigned word = byte << (8 * cg);
            word |= __shfl_xor(word, 16, 64);
            word |= __shfl_xor(word, 32, 64);
            if (ok && cg == (j & 3)) p.ybits[(long long)((co0 + wc * 64 + 32 * half) >> 5) * p.M + m] = word;
        }
    }
}

// ---- 256 x 256 tile, persistent, four-stage LDS-DMA ring -----------------------------------------------------------------------------
// Across the step's sixteen launch classes the 128 x 128 kernel above moves a near-constant 6.3 - 9.6 TB/s BETWEEN L2 AND THE CUs (every
// activation tile is fetched Cout / 128 times, the weight matrix M / 128 times) while its HBM-level rate falls from 4.5 TB/s (res2: 2
// re-reads) to 1.5 - 2.5 TB/s (res5: 16): what bounds it is the bytes it keeps in flight per CU (4 workgroups x one 16 KB K step) over a
// ~2.5 us loaded round trip.  This kernel halves the L2 -> CU bytes per output (256 channels x 256 pixels per tile: 8 waves = 2
// channel halves x 4 pixel quarters, wave tile 128 x 64, acc[8][4] as in conv3x3_pp.hip) and keeps THREE 32 KB K steps in flight at
// all times: both operands go by LDS-DMA straight into a four-stage ring (no staging registers, no LDS writes, one raw barrier per 32
// MFMAs, swizzle and channel permutation on the SOURCE address), and the workgroup is persistent -- the ring runs on across tile
// boundaries, so the next tile's first three K steps are in flight while this tile's epilogue reads and writes its operands.
constexpr int BG_T = 256, BG_BK = 32;
constexpr int BG_HALF = BG_T * 64;             // one operand tile of a stage: 256 rows x 64 B
constexpr int BG_STAGE = 2 * BG_HALF;          // 32768
constexpr int BG_NSTAGE = 4;
constexpr int BG_LDS = BG_NSTAGE * BG_STAGE;   // 131072

__global__ __launch_bounds__(512, 1) void conv1x1_big_kernel(const P1 p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2;       // channel half (128 rows)
    const int wp = wave & 3;        // pixel quarter (64 pixels)
    const int total = p.m_tiles * p.n_tiles;
    const int grid = gridDim.x;
    // tile of (round i, workgroup b): with a grid that is a multiple of 8 the round's ids are dealt XCD-major, so that the channel tiles
    // of one pixel tile (consecutive ids) run on ONE XCD and share its L2
    auto tile_of = [&](int i) {
        const int b = blockIdx.x;
        const int id = (grid & 7) == 0 ? (b & 7) * (grid >> 3) + (b >> 3) : b;
        return i * grid + id;
    };

    constexpr unsigned X_NONE = 0x80000000u;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(p.w), 0, p.w_bytes, 0x00020000);
    const int nsteps = p.CK / BG_BK;               // the host takes CK % 32 == 0 only

    // ---- producer: K step ps of this workgroup's pi-th tile goes to ring stage (flat index & 3).  An operand tile of a stage = 16
    // pieces of 1 KiB (16 rows x 64 B); this wave owns pieces wave and wave + 8 of both; lane -> row lane >> 2, position lane & 3,
    // source chunk = position ^ ((row >> 1) & 3).  Past the last tile the offsets are X_NONE (zeros land in a stage nobody reads): the
    // number of DMA instructions in flight stays what the counted waits assume.
    unsigned a_src[2], b_src[2];
    int pi = 0, ps = 0, pflat = 0;
    auto producer_tile = [&]() {
        const int t = tile_of(pi);
        const bool live = t < total;
        const int tm = t / p.n_tiles, tn = t - tm * p.n_tiles;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int lrow = 16 * (wave + 8 * k) + (lane >> 2);
            const int chunk = (lane & 3) ^ ((lrow >> 1) & 3);
            const int rho = lrow & 15;
            const int co = tn * BG_T + (lrow & 192) + 32 * ((lrow >> 5) & 1) + 8 * (rho >> 2) + 4 * ((lrow >> 4) & 1) + (rho & 3);
            a_src[k] = (live && co < p.CO) ? (unsigned)(co * p.CK + chunk * 8) * 2u : X_NONE;
            const int m = tm * BG_T + lrow;
            b_src[k] = (live && m < p.M) ? (unsigned)(m * p.CK + chunk * 8) * 2u : X_NONE;
        }
    };
    auto produce = [&]() {
        int so = ps * (BG_BK * 2);
        asm volatile("" : "+s"(so));
        unsigned char* At = smem + (pflat & (BG_NSTAGE - 1)) * BG_STAGE;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void_1x1_t*)(At + (wave + 8 * k) * 1024), 16, a_src[k], so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void_1x1_t*)(At + BG_HALF + (wave + 8 * k) * 1024), 16, b_src[k], so, 0, 0);
        }
        ++pflat;
        if (++ps == nsteps) { ps = 0; ++pi; producer_tile(); }
    };
    producer_tile();
#pragma unroll
    for (int u = 0; u < BG_NSTAGE - 1; ++u) produce();

    const int frow = lane & 15, fchunk = lane >> 4;
    const int cg = lane >> 4;
    const bool do_relu = p.flags & BD_EPI_RELU;
    const bool add_before = (p.flags & BD_EPI_ADD_BEFORE) && p.add;
    const bool add_after = (p.flags & BD_EPI_ADD_AFTER) && p.add;
    const bool want_add = add_before || add_after;
    const bool mask_bf = (p.flags & BD_EPI_MASK) && p.mask;
    const bool mask_bits = (p.flags & BD_EPI_MASK) && p.maskbits && !p.mask;
    int cflat = 0;
    for (int ci = 0; tile_of(ci) < total; ++ci) {
    const int tile = tile_of(ci);
    const int tile_m = tile / p.n_tiles;
    const int tile_n = tile - tile_m * p.n_tiles;
    const int m0 = tile_m * BG_T;
    const int co0 = tile_n * BG_T;
    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // LDS reads run UNDER the matrix instructions: with one barrier per step all eight waves are in phase, and reading a step's
    // fragments and then issuing its MFMAs makes the CU alternate between an LDS phase (12 KB per wave: ~770 cycles) and a matrix
    // phase (32 MFMAs per wave, two waves per SIMD: 1 024 cycles).  (Measured: the same 6.5 ms over the step's launches with and
    // without this overlap -- the kernel's time per tile is set elsewhere: see the launcher.)  A step is two halves of 16
    // MFMAs (channel tiles 0-3 / 4-7): the first half runs over the reads of the second half's channel fragments, the second over
    // the reads of the next step's pixel fragments and first channel half (64 fragment registers: a full second set spilled, and a
    // reloaded spill waits in vmcnt order behind the ring's DMA).  The stage whose fragments sit in registers is free, so the
    // producer stays three stages ahead of the stage being read.
    bf16x8_t fa_lo[4], fa_hi[4], fb[2][4];
    auto stage_of = [&](int flat) { return smem + (flat & (BG_NSTAGE - 1)) * BG_STAGE; };
    auto read_b = [&](int flat, int set) {
        const unsigned char* Bt = stage_of(flat) + BG_HALF;
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[set][j] = *reinterpret_cast<const bf16x8_t*>(Bt + lds_off(wp * 64 + j * 16 + frow, fchunk));
    };
    auto read_a = [&](int flat, int half, bf16x8_t (&f)[4]) {
        const unsigned char* At = stage_of(flat);
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const bf16x8_t*>(At + lds_off(wm * 128 + (4 * half + i) * 16 + frow, fchunk));
    };
    auto step_sync = [&]() {
        // stages cflat', +1, +2 are in flight (4 DMA instructions each; epilogue loads / stores are younger and only make the wait
        // stricter): the oldest has landed when at most 8 are outstanding.  lgkmcnt: this wave's earlier fragment reads have returned,
        // so after the barrier the stage they came from may be overwritten
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
        if (t + 1 < nsteps) {
            step_sync();
            read_b(cflat + t + 1, set ^ 1);
            read_a(cflat + t + 1, 0, fa_lo);
        }
        mfmas(1, fa_hi, set);
    };
    step_sync();
    read_b(cflat, 0);
    read_a(cflat, 0, fa_lo);
    for (int t = 0; t < nsteps; t += 2) {            // two steps per trip: the pixel-fragment sets are compile-time register names
        one_step(t, 0);
        if (t + 1 < nsteps) one_step(t + 1, 1);
    }
    cflat += nsteps;

    // ---- epilogue: lane group cg holds channels cbase + 32 h + 0..7 (h = 0..3) of pixel m0 + wp*64 + j*16 + (lane & 15).  Two halves
    // (h pairs); a half first REQUESTS all of its operands (8 x residual / mask / gate words: up to 16 KB per wave in flight -- with
    // eight waves per CU a load-use-store chain per 16 bytes would leave the epilogue, half of this kernel's bytes, latency-bound),
    // then computes and stores
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
            if (p.y8 && ok) {
                u32x2_t o8;
                if (p.y8_bf8) {
                    o8 = e5m2_pair(v, p.q_scale, p.sr_seed, idx);
                } else {
                    o8[0] = pack4_e4m3(v[0] * p.q_scale, v[1] * p.q_scale, v[2] * p.q_scale, v[3] * p.q_scale);
                    o8[1] = pack4_e4m3(v[4] * p.q_scale, v[5] * p.q_scale, v[6] * p.q_scale, v[7] * p.q_scale);
                }
                *reinterpret_cast<u32x2_t*>(p.y8 + idx) = o8;
            }
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
    }   // tiles
    // the producer ran three (empty) stages past the last tile: let those DMA writes land before the LDS can be handed to another workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

BD_KNOB int g_conv1x1_dma_k = 512;      // BD_DENSE1X1_DMA_K: smallest CK that takes the LDS-DMA ring variant
BD_KNOB int g_conv1x1_depth = 1;        // bd_conv_desc.route[0]: 0 = off (generic kernel), 1 = the 128^2 tile (default), 2 = the 256^2 tile wherever legal (A/B), 3 = as 1,
                                // 4 = the 256-channel x 128-pixel eight-wave tile wherever legal (A/B)
BD_KNOB int g_conv1x1_wide_min_k = 0;   // BD_DENSE1X1_WIDE_K: smallest CK that takes the eight-wave tile in mode 4

}  // namespace

void bd_conv1x1_ring_everywhere(bool on);         // conv1x1_ring.hip

// bd_conv_desc.route[0] - 1 (BdRouteScope, conv_igemm.hip); the modes: include/basedet_hip.h
int bd_route_dense1x1(int depth) {
    if (depth < 0 || depth > 6) {
        bd_set_error("bd_conv_desc.route[0]: dense 1x1 mode %d (0 .. 6)", depth);
        return BD_EINVAL;
    }
    bd_conv1x1_ring_everywhere(depth == 5);          // 5 = as 1, with conv1x1_ring_kernel for every launch it can take (default: K <= 256 into >= 256 channels)
    // 6 = conv1x1_dense_kernel only (as 3) with its LDS-DMA ring variant for every K that allows it (default: 512 <= K <= 1024): test coverage
    g_conv1x1_dma_k = bd_tune_env("BD_DENSE1X1_DMA_K", depth == 6 ? 32 : 512);
    if (depth == 5) depth = 1;
    if (depth == 6) depth = 3;
    g_conv1x1_depth = depth;
    g_conv1x1_wide_min_k = bd_tune_env("BD_DENSE1X1_WIDE_K", g_conv1x1_wide_min_k);
    return BD_OK;
}

int bd_conv1x1_ring_launch(const void* x, const void* w, const float* bias, const void* add, const void* mask, const unsigned* maskbits, void* y,
                           unsigned* ybits, void* y8, long long M, int CK, int CO, int flags, hipStream_t stream);      // conv1x1_ring.hip
int bd_conv1x1_ring_fp8_launch(int mode, const void* xq, const void* wq, const float* wscale, const float* bias, const void* add,
                               const unsigned* maskbits, void* y, unsigned* ybits, void* y8, float q_scale, unsigned sr_seed, long long M, int CK,
                               int CO, int flags, hipStream_t stream);                                                   // conv1x1_ring.hip

// Called by bd_conv2d_fwd / bd_conv2d_dgrad (conv_igemm.hip) for 1x1 / stride 1 / pad 0 launches over one dense level.
// Returns 0 when the launch was taken, 1 when the shape is left to the generic kernel.
int bd_conv1x1_dense_launch(const void* x, const void* w, const float* bias, const void* add, const void* mask, const unsigned* maskbits,
                            void* y, unsigned* ybits, void* y8, float q_scale, int y8_bf8, long long M, int CK, int CO, int flags,
                            hipStream_t stream) {
    if (g_conv1x1_depth == 0) return 1;
    if (g_conv1x1_depth == 1 && bd_conv1x1_ring_launch(x, w, bias, add, mask, maskbits, y, ybits, y8, M, CK, CO, flags, stream) == 0) return 0;
    const long long xb = M * CK * 2, wb = (long long)CO * CK * 2;
    if (xb >= 0x7fffffffll || wb >= 0x7fffffffll || M >= (1ll << 24)) return 1;
    if ((maskbits || ybits) && (CO % 32 != 0)) return 1;
    P1 p{};
    p.sr_seed = g_fp8_sr_seed;
    p.x = (const bf16_raw*)x; p.w = (const bf16_raw*)w; p.bias = bias; p.add = (const bf16_raw*)add; p.mask = (const bf16_raw*)mask;
    p.maskbits = maskbits; p.y = (bf16_raw*)y; p.ybits = ybits; p.y8 = (unsigned char*)y8; p.q_scale = q_scale; p.y8_bf8 = y8_bf8;
    p.M = (int)M; p.CK = CK; p.CO = CO; p.flags = flags;
    p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
#ifdef BD_D1_STAMP
    p.dbg = bd_tune_env("BD_D1_ABLATE", 0);          // (a -DBD_D1_STAMP build is also a -DBD_TUNING build: scripts/exp/d1_stamp.py)
#endif
    {
        // the 256^2 tile: full 32-channel K steps and at least one full channel tile.  Measured (scripts/micro_1x1_step.py, the step's 16
        // launch classes): on par or slower than the 128^2 tile everywhere (6.2 - 6.5 vs 5.5 - 5.8 ms over the step's launches; its
        // one workgroup per CU quantises worse on the 263-tile res4 layers and its epilogue has 8 waves per CU to hide latency with,
        // not 16), so it runs on request only (bd_conv_desc.route[0] mode 2: A/B)
        const int mt = (int)cdiv64(M, BG_T), nt = cdiv(CO, BG_T);
        const bool legal = CK % BG_BK == 0 && CO >= BG_T && g_conv1x1_depth != 3;
        if (legal && g_conv1x1_depth == 2) {
            BD_ONCE_PER_DEVICE(
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BG_LDS));
            p.m_tiles = mt; p.n_tiles = nt;
            const long long tiles = (long long)mt * nt;
            bd_note_kernel("conv1x1_big_kernel");
            hipLaunchKernelGGL(conv1x1_big_kernel, dim3((int)(tiles < 256 ? tiles : 256)), dim3(512), BG_LDS, stream, p);
            return 0;
        }
    }
    if (g_conv1x1_depth == 4 && CK % BK == 0 && CO >= 256 && CK >= g_conv1x1_wide_min_k) {
        constexpr int WIDE_LDS = 3 * (256 * 64 + TILE_BYTES);          // 72 KB: two eight-wave workgroups per CU
        BD_ONCE_PER_DEVICE(
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_dense_kernel<1, 4, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, WIDE_LDS));
        p.m_tiles = (int)cdiv64(M, TP); p.n_tiles = cdiv(CO, 256);
        bd_note_kernel("conv1x1_dense_kernel");
        hipLaunchKernelGGL((conv1x1_dense_kernel<1, 4, true, 4>), dim3(p.m_tiles * p.n_tiles), dim3(512), WIDE_LDS, stream, p);
        return 0;
    }
    p.m_tiles = (int)cdiv64(M, TP); p.n_tiles = cdiv(CO, TC);
    const int grid = p.m_tiles * p.n_tiles;
    bd_note_kernel("conv1x1_dense_kernel");
    // 512 <= CK <= 1024 (res3 / res4 conv1 and conv3's data gradient, the laterals): the three-stage LDS-DMA ring (two K steps in flight
    // per workgroup, three workgroups per CU) is 3 - 11 % faster; shorter K (the epilogue is most of the tile) and the 16 800-pixel res5
    // layers (K = 2048: everything L2-resident, four workgroups per CU hide more) stay on the register-staged loop.  Measured per class
    // with scripts/micro_1x1_step.py; BD_DENSE1X1_DMA_K (read by bd_conv_desc.route[0]) moves the lower bound for A/B.
    static const int early = bd_tune_env("BD_DENSE1X1_EARLY", 0);     // measurement: epilogue operands requested before the K loop
    const bool has_ops = (add && (flags & (BD_EPI_ADD_BEFORE | BD_EPI_ADD_AFTER))) || ((flags & BD_EPI_MASK) && (mask || maskbits));
    if (CK % BK == 0 && CK >= g_conv1x1_dma_k && (CK <= 1024 || g_conv1x1_dma_k < 512)) {
        if (early && has_ops) hipLaunchKernelGGL((conv1x1_dense_kernel<1, 3, true, 2, 8, true>), dim3(grid), dim3(256), 6 * TILE_BYTES, stream, p);
        else hipLaunchKernelGGL((conv1x1_dense_kernel<1, 3, true>), dim3(grid), dim3(256), 6 * TILE_BYTES, stream, p);
        return 0;
    }
    const size_t lds = 4 * TILE_BYTES;
    if (early && has_ops) {
        if (early == 2) hipLaunchKernelGGL((conv1x1_dense_kernel<2, 3, false, 2, 8, true>), dim3(grid), dim3(256), lds, stream, p);
        else hipLaunchKernelGGL((conv1x1_dense_kernel<1, 3, false, 2, 8, true>), dim3(grid), dim3(256), lds, stream, p);
        return 0;
    }
    // (round 4, measured and removed: all eight epilogue operands requested at once -- EBATCH = 8 at three waves per SIMD -- 5.44 ms over
    // the step's launches against 5.41: the epilogue is not short of requests in flight)
    static const int reg_depth = bd_tune_env("BD_DENSE1X1_REGDEPTH", 1);     // register sets in flight (measurement)
    if (reg_depth == 2) hipLaunchKernelGGL((conv1x1_dense_kernel<2, 3, false>), dim3(grid), dim3(256), lds, stream, p);
    else if (reg_depth == 3) hipLaunchKernelGGL((conv1x1_dense_kernel<3, 3, false>), dim3(grid), dim3(256), lds, stream, p);
    else if (reg_depth == 4) hipLaunchKernelGGL((conv1x1_dense_kernel<4, 3, false>), dim3(grid), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((conv1x1_dense_kernel<1, 4, false>), dim3(grid), dim3(256), lds, stream, p);
    return 0;
}

// 1x1 / stride 2 / pad 0 over one dense level (the bottleneck shortcut convolutions) on the same kernel: mode 0 = forward (the source rows
// are the (2y, 2x) pixels of the input), mode 1 = the data gradient ADDED IN PLACE at the pixels it reaches (BD_EPI_SPARSE with
// BD_EPI_ADD_BEFORE and add == dx; the other pixels of dx keep what they hold).  Called from conv_igemm.hip; 0 = taken.
BD_KNOB int g_conv1x1_s2 = 1;        // bd_conv_desc.route[1] bit 12 clears it (A/B against the generic kernel)
int bd_conv1x1_s2_launch(const bd_conv_desc* d, int mode, const void* src, const void* w, const float* bias, const void* add, const void* mask,
                         const unsigned* maskbits, void* dst, int flags, hipStream_t stream) {
    if (!g_conv1x1_s2 || g_conv1x1_depth == 0) return 1;
    if (!(d->R == 1 && d->S == 1 && d->stride == 2 && d->pad == 0 && d->nseg == 1 && d->in_off[0] == 0 && d->out_off[0] == 0 &&
          d->in_pix_per_img == d->Hi[0] * d->Wi[0] && d->out_pix_per_img == d->Ho[0] * d->Wo[0])) return 1;
    if (d->Ho[0] != (d->Hi[0] - 1) / 2 + 1 || d->Wo[0] != (d->Wi[0] - 1) / 2 + 1) return 1;
    const long long M = (long long)d->N * d->out_pix_per_img, Mbig = (long long)d->N * d->in_pix_per_img;
    const int CK = mode == 0 ? d->Cin : d->Cout, CO = mode == 0 ? d->Cout : d->Cin;
    if (mode == 1 && !((flags & BD_EPI_SPARSE) && (flags & BD_EPI_ADD_BEFORE) && add == dst)) return 1;
    if (mode == 1 && maskbits && CO % 32 != 0) return 1;
    // 32-bit byte offsets: the source tensor and the weights below 2 GB, float-reciprocal row decode below 2^24 rows
    const long long src_bytes = (mode == 0 ? Mbig : M) * CK * 2, wb = (long long)CO * CK * 2;
    if (src_bytes >= 0x7fffffffll || wb >= 0x7fffffffll || M >= (1ll << 24) || Mbig >= (1ll << 30)) return 1;
    P1 p{};
    p.sr_seed = g_fp8_sr_seed;
    p.x = (const bf16_raw*)src; p.w = (const bf16_raw*)w; p.bias = bias; p.add = (const bf16_raw*)add; p.mask = maskbits ? nullptr : (const bf16_raw*)mask;
    p.maskbits = maskbits; p.y = (bf16_raw*)dst; p.q_scale = 1.f;
    p.M = (int)M; p.CK = CK; p.CO = CO; p.flags = flags & ~BD_EPI_SPARSE;
    p.x_bytes = (unsigned)src_bytes; p.w_bytes = (unsigned)wb;
    p.s2_src = mode == 0; p.s2_dst = mode == 1;
    p.dW = d->Wo[0]; p.dHW = d->Ho[0] * d->Wo[0]; p.bW = d->Wi[0]; p.bHW = d->Hi[0] * d->Wi[0]; p.Mbig = (int)Mbig;
    p.inv_dW = 1.0f / (float)p.dW; p.inv_dHW = 1.0f / (float)p.dHW;
    p.m_tiles = (int)cdiv64(M, TP); p.n_tiles = cdiv(CO, TC);
    const int grid = p.m_tiles * p.n_tiles;
    bd_note_kernel("conv1x1_dense_kernel");
    if (CK % BK == 0 && CK >= g_conv1x1_dma_k && (CK <= 1024 || g_conv1x1_dma_k < 512))
        hipLaunchKernelGGL((conv1x1_dense_kernel<1, 3, true>), dim3(grid), dim3(256), 6 * TILE_BYTES, stream, p);
    else
        hipLaunchKernelGGL((conv1x1_dense_kernel<1, 4, false>), dim3(grid), dim3(256), 4 * TILE_BYTES, stream, p);
    return 0;
}

// fp8 form of a dense 1x1 launch (1x1 / stride 1 / pad 0 over one dense level): mode 0 = forward (xq e4m3), mode 1 = data gradient
// (xq = e5m2 gradient of the conv output; "CK" is then Cout and "CO" Cin).  See include/basedet_hip.h.
extern "C" int bd_conv1x1_fp8(const bd_conv_desc* d, int mode, const void* xq, const void* wq, const float* wscale, const float* bias,
                              const void* add, const void* mask, const uint32_t* maskbits, void* y, uint32_t* ybits, void* y8, float q_scale,
                              int flags, bd_stream_t stream) {
    BD_ROUTE(d);
    BD_REQUIRE(d && xq && wq && wscale && y, "conv1x1_fp8: null pointer");
    BD_REQUIRE(mode == 0 || mode == 1, "conv1x1_fp8: mode %d (0 forward, 1 data gradient)", mode);
    BD_REQUIRE(d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0 && d->nseg == 1 && d->in_off[0] == 0 && d->out_off[0] == 0 &&
               d->in_pix_per_img == (long long)d->Hi[0] * d->Wi[0] && d->out_pix_per_img == (long long)d->Ho[0] * d->Wo[0],
               "conv1x1_fp8: 1x1 / stride 1 / pad 0 over one dense level only");
    const long long M = (long long)d->N * d->in_pix_per_img;
    const int CK = mode == 0 ? d->Cin : d->Cout, CO = mode == 0 ? d->Cout : d->Cin;
    BD_REQUIRE(CK % 128 == 0 && CO % 32 == 0, "conv1x1_fp8: K = %d must be a multiple of 128, the produced channels (%d) of 32", CK, CO);
    BD_REQUIRE(M * CK < 0x7fffffffll && M * CO * 2 < 0x7fffffffffll && M < (1ll << 24), "conv1x1_fp8: tensor too large for 32-bit offsets");
    BD_REQUIRE(!(mode == 0 && (flags & BD_EPI_MASK)) && !(mode == 1 && (flags & BD_EPI_RELU)), "conv1x1_fp8: flag / mode mismatch");
    P1 p{};
    p.sr_seed = g_fp8_sr_seed;
    p.x = (const bf16_raw*)xq; p.w = (const bf16_raw*)wq; p.wscale = wscale; p.bias = bias; p.add = (const bf16_raw*)add;
    p.mask = maskbits ? nullptr : (const bf16_raw*)mask; p.maskbits = maskbits; p.y = (bf16_raw*)y; p.ybits = ybits;
    p.y8 = (unsigned char*)y8; p.q_scale = q_scale; p.y8_bf8 = mode;
    p.M = (int)M; p.CK = CK; p.CO = CO; p.flags = flags | (maskbits ? BD_EPI_MASK : 0);
    p.x_bytes = (unsigned)(M * CK); p.w_bytes = (unsigned)((long long)CO * CK);
    p.m_tiles = (int)cdiv64(M, TP); p.n_tiles = cdiv(CO, TC);
    // round 6: the ring kernel's one-byte form for the launch classes its bf16 form takes (bd_conv_desc.route[0]: 1 default, 5 = every legal
    // launch, 3 = never); bf16 gates stay here
    if (g_conv1x1_depth == 1 && !p.mask &&
        bd_conv1x1_ring_fp8_launch(mode, xq, wq, wscale, bias, add, maskbits, y, ybits, y8, q_scale, p.sr_seed, M, CK, CO, p.flags, (hipStream_t)stream) == 0) {
        BD_CHECK_LAUNCH("bd_conv1x1_fp8");
        return BD_OK;
    }
    const int grid = p.m_tiles * p.n_tiles;
    BD_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_fp8_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * F8_TILE);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_fp8_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * F8_TILE));
    bd_note_kernel("conv1x1_fp8_kernel");
    if (mode == 0) hipLaunchKernelGGL((conv1x1_fp8_kernel<0>), dim3(grid), dim3(256), 4 * F8_TILE, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((conv1x1_fp8_kernel<1>), dim3(grid), dim3(256), 4 * F8_TILE, (hipStream_t)stream, p);
    BD_CHECK_LAUNCH("bd_conv1x1_fp8");
    return BD_OK;
}
