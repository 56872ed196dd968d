// Shared device/host helpers for libbasedet_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include <string>

#include "../../include/basedet_hip.h"

typedef unsigned short bf16_raw;  // storage type of a bf16 element
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

__device__ __forceinline__ float bf2f(bf16_raw h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ bf16_raw f2bf(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserved
    return __builtin_bit_cast(bf16_raw, b);
}
// ---- stochastic rounding of the e5m2 gradients (bd_conv_desc.sr_seed) ----------------------------------------------------
// Round-to-nearest e5m2 (two mantissa bits) makes the SAME error on the same value every step; on a repeated batch those errors add up as a
// bias and the run drifts.  With a seed set, the quantisers add a pseudo-random fraction below the kept bits before truncating
// (v_cvt_sr_bf8_f32): unbiased, and -- the random word being a hash of (seed, element index) -- independent of which kernel variant or
// tile writes the element.
extern thread_local unsigned g_fp8_sr_seed;          // host side, conv_fp8.hip (set per call from bd_conv_desc.sr_seed / bd_quantize_bf8's argument): 0 = round to nearest
__device__ __forceinline__ unsigned bd_mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned bd_pack4_e5m2_sr(float a, float b, float c, float d, unsigned r) {
    a = fminf(fmaxf(a, -57344.f), 57344.f); b = fminf(fmaxf(b, -57344.f), 57344.f);
    c = fminf(fmaxf(c, -57344.f), 57344.f); d = fminf(fmaxf(d, -57344.f), 57344.f);
    int v = 0;                        // one hashed word per four elements, rotated by a byte per element (v_alignbit: one instruction each)
    v = __builtin_amdgcn_cvt_sr_bf8_f32(a, r, v, 0);
    v = __builtin_amdgcn_cvt_sr_bf8_f32(b, __builtin_rotateright32(r, 8), v, 1);
    v = __builtin_amdgcn_cvt_sr_bf8_f32(c, __builtin_rotateright32(r, 16), v, 2);
    v = __builtin_amdgcn_cvt_sr_bf8_f32(d, __builtin_rotateright32(r, 24), v, 3);
    return (unsigned)v;
}

// ---- one-byte twins written by the 1x1 epilogues (conv1x1.hip, conv1x1_ring.hip) ---------------------------------------------------------
__device__ __forceinline__ unsigned pack4_e4m3(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
    c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
    int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (unsigned)v;
}

__device__ __forceinline__ unsigned pack4_e5m2(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -57344.f), 57344.f); b = fminf(fmaxf(b, -57344.f), 57344.f);
    c = fminf(fmaxf(c, -57344.f), 57344.f); d = fminf(fmaxf(d, -57344.f), 57344.f);
    int v = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false);
    v = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, v, true);
    return (unsigned)v;
}

// eight consecutive channels (element index idx, a multiple of 8) -> eight e5m2 bytes; seed != 0: stochastic rounding (common.h)
__device__ __forceinline__ u32x2_t e5m2_pair(const float* v, float qs, unsigned seed, long long idx) {
    u32x2_t o8;
    if (seed) {
        const unsigned g0 = (unsigned)(idx >> 2);
        const unsigned r0 = bd_mix32(seed ^ g0);           // one full hash per eight elements; the second word by a multiply-add
        o8[0] = bd_pack4_e5m2_sr(v[0] * qs, v[1] * qs, v[2] * qs, v[3] * qs, r0);
        o8[1] = bd_pack4_e5m2_sr(v[4] * qs, v[5] * qs, v[6] * qs, v[7] * qs, r0 * 0x9e3779b1u + 0x7f4a7c15u);
    } else {
        o8[0] = pack4_e5m2(v[0] * qs, v[1] * qs, v[2] * qs, v[3] * qs);
        o8[1] = pack4_e5m2(v[4] * qs, v[5] * qs, v[6] * qs, v[7] * qs);
    }
    return o8;
}

// One-byte twin of a pixel's 64 wave-channels: lane group cg holds 8 bytes of each 32-channel half; the lane pairs (cg, cg ^ 1) swap one
// piece so that every lane stores 16 contiguous bytes (64-byte runs per pixel and instruction instead of 32-byte ones): even cg ends up
// with bytes [8 cg, 8 cg + 16) of half 0, odd cg with bytes [8 (cg - 1), 8 (cg - 1) + 16) of half 1.  `keep` = this lane's piece of the
// half it stores, `send` = its piece of the other half.  Returns the 16 bytes; *off = byte offset inside the pixel's 64-byte run.
__device__ __forceinline__ u32x4_t twin_pair(u32x2_t h0, u32x2_t h1, int cg, int* off) {
    const bool odd = cg & 1;
    const u32x2_t send = odd ? h0 : h1;
    u32x2_t recv;
    recv[0] = (unsigned)__shfl_xor((int)send[0], 16, 64);
    recv[1] = (unsigned)__shfl_xor((int)send[1], 16, 64);
    *off = odd ? 32 + 8 * (cg - 1) : 8 * cg;
    return odd ? (u32x4_t){recv[0], recv[1], h1[0], h1[1]} : (u32x4_t){h0[0], h0[1], recv[0], recv[1]};
}

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// wave64 sum reduction (result valid in every lane)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- in-kernel clock probe (bd_probe_kernel_clock) -----------------------------------------------------------------------------------
// The chip lowers its clock under MFMA-dense load (MI355X_MICROARCH.md, DVFS give-back), so a TFLOP/s figure is (in-cycle efficiency) x
// (held clock).  The two dominant MFMA-bound kernels carry this probe in the SHIPPED build: lane 0 of workgroup 0 reads the shader-cycle
// and the 100 MHz counters when it starts and when it ends (four scalar reads and two atomics per LAUNCH; a persistent workgroup lives as
// long as the launch) and adds both differences to a pair of device counters nobody else reads; the host divides the sums.
// No register lives across the kernel for it (conv3x3_pp_kernel has none to spare: a held stamp pair spilled four VGPRs into its K loop):
// the START stamps are SUBTRACTED from the counters right away (mod 2^64), the END stamps added -- the sums are meaningful once the
// launches have finished, which is when the host reads them.
__device__ __forceinline__ void bd_clk_mark(unsigned long long* acc, bool end) {
#ifdef BD_NO_CLK_PROBE          // (A/B build: what the probe costs -- profiles/r06_clk_probe_ab.txt)
    return;
#endif
    const unsigned long long c = __builtin_amdgcn_s_memtime(), r = __builtin_amdgcn_s_memrealtime();
    atomicAdd(acc, end ? c : 0ull - c);
    atomicAdd(acc + 1, end ? r : 0ull - r);
}

// ---- kernel routing is PER CALL -------------------------------------------------------------------------------------------------------
// The routing knobs of rounds 1-5 (bd_conv_desc.route[0], bd_conv_desc.route[1], bd_conv_desc.route[2], bd_conv_desc.route[3],
// bd_conv_desc.sr_seed) were process-global setters.  Round 6: a caller that wants a route says so in the descriptor of the call
// (bd_conv_desc.route[], .sr_seed: include/basedet_hip.h); every entry point that takes a descriptor opens a BdRouteScope, which writes the
// routing variables below from THAT descriptor (library defaults where a word is 0) -- nothing survives the call.  The variables are
// thread_local: two host threads issuing calls with different routes do not see each other.
#define BD_KNOB thread_local
struct BdRouteScope {
    int rc;                    // BD_OK, or BD_EINVAL (message set) for a route word out of range
    explicit BdRouteScope(const bd_conv_desc* d);
};
#define BD_ROUTE(d)                                   \
    BdRouteScope route_scope__(d);                    \
    if (route_scope__.rc != BD_OK) return route_scope__.rc

void bd_set_error(const char* fmt, ...);
// bd_conv_last_kernel(): every convolution launch site names the kernel it dispatched to (a string literal; thread-local, host side only)
void bd_note_kernel(const char* name);

#define BD_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            bd_set_error(__VA_ARGS__);   \
            return BD_EINVAL;            \
        }                                \
    } while (0)

#define BD_CHECK_LAUNCH(name)                                                     \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            bd_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return BD_ELAUNCH;                                                    \
        }                                                                         \
    } while (0)

// Measurement switches from the ENVIRONMENT (tile targets, staging depths, A/B routes of scripts/) exist in -DBD_TUNING diagnostic builds
// only (BD_LIB_NAME / BD_EXTRA_FLAGS of basedet_amd/build.py): the shipped library reads no environment variable except BD_RCCL_LIB (where
// RCCL lives: deployment, not behaviour).  Nothing is process-global since round 6: kernel routing travels in bd_conv_desc.route[] per call.
#ifdef BD_TUNING
#include <stdlib.h>
static inline int bd_tune_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static inline const char* bd_tune_env_str(const char* name) { return getenv(name); }
#else
static inline int bd_tune_env(const char*, int dflt) { return dflt; }
static inline const char* bd_tune_env_str(const char*) { return nullptr; }
#endif

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- per-device launch state -------------------------------------------------------------------------------------------------------
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the CU count are properties of a DEVICE: a process-wide `static bool` set them for the
// first device only, and two threads racing through the first call could launch before the attribute was in place.
//   BD_ONCE_PER_DEVICE(stmts): runs stmts once per (call site, current device), under a lock the later callers of that site also take
//   bd_num_cus(): multiProcessorCount of the current device (256 if the query fails)
struct BdDeviceOnce {
    std::mutex mu;
    unsigned long long done[4] = {0, 0, 0, 0};      // 256 device ordinals
    struct Guard {
        BdDeviceOnce& o; int dev = 0; bool first_;
        explicit Guard(BdDeviceOnce& o_) : o(o_) {
            o.mu.lock();
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 255) dev = 0;
            first_ = !((o.done[dev >> 6] >> (dev & 63)) & 1ull);
        }
        bool first() const { return first_; }
        ~Guard() { if (first_) o.done[dev >> 6] |= 1ull << (dev & 63); o.mu.unlock(); }
    };
};
#define BD_ONCE_PER_DEVICE(...)                 \
    do {                                        \
        static BdDeviceOnce once__;             \
        BdDeviceOnce::Guard guard__(once__);    \
        if (guard__.first()) { __VA_ARGS__; }   \
    } while (0)
int bd_num_cus();
