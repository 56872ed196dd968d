// Measurement probe (not on the training path): what the matrix pipe of THIS device sustains on the register-level pattern of the
// convolution kernels -- 8 A fragments x 4 B fragments of full-entropy random bf16 rotating through 32 accumulators of
// v_mfma_f32_16x16x32_bf16, no LDS, no memory -- after `seconds` of back-to-back launches (the chip lowers its clock under this load:
// MI355X_MICROARCH.md "DVFS give-back"; round 2 read 2.05 PFLOP/s at 2.09-2.13 GHz, docs/HISTORY_r1_r2.md).  bench.py reports it as
// roofline.peak_measured beside the vendor peak.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void mfma_rotating_kernel(const unsigned* seed, int iters, float* out, unsigned long long* stamps) {
    unsigned s = seed[threadIdx.x & 63] ^ (threadIdx.x * 2654435761u);
    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    union U { bf16x8_t v; unsigned u[4]; } a[8], b[4];
    // sign, 7 mantissa bits, exponents 2^-3 .. 2^0
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; const unsigned m = s & 0x807f807fu, e = ((s >> 8) & 0x00030003u) << 7; return m | (0x3e003e00u + e); };
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) a[i].u[k] = rnd();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) b[j].u[k] = rnd();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)         // tied D = C (the builtin form lets the compiler rotate the accumulators through moves)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a[i].v), "v"(b[j].v));
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = r1 - r0; }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) t += acc[i][j][0];
    if (t == 12345.678f) out[0] = t;
}

}  // namespace

// SYNCHRONOUS (a probe, not an operator): runs on `stream` for about `seconds`, then times the last launches with HIP events.
// tflops_out: sustained dense bf16 TFLOP/s of the whole device; clock_mhz_out (may be NULL): in-kernel clock of workgroup 0.
extern "C" int bd_probe_mfma_rate(double seconds, double* tflops_out, double* clock_mhz_out, bd_stream_t stream) {
    BD_REQUIRE(tflops_out != nullptr && seconds > 0.0 && seconds <= 30.0, "bd_probe_mfma_rate: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    unsigned h[64];
    unsigned x = 0x9E3779B9u;
    for (int i = 0; i < 64; ++i) { x = x * 1664525u + 1013904223u; h[i] = x; }
    unsigned* d = nullptr; float* o = nullptr; unsigned long long* stamps = nullptr;
    hipEvent_t ea = nullptr, eb = nullptr;
    int rc = BD_ELAUNCH;
    do {
        if (hipMalloc(&d, sizeof(h)) != hipSuccess || hipMalloc(&o, 4) != hipSuccess || hipMalloc(&stamps, 16) != hipSuccess) break;
        if (hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, st) != hipSuccess) break;
        if (hipEventCreate(&ea) != hipSuccess || hipEventCreate(&eb) != hipSuccess) break;
        const int iters = 5000, grid = bd_num_cus() * 2;          // 8 waves per CU, two per SIMD
        const double flops = (double)grid * 4 * iters * 32 * 2.0 * 16 * 16 * 32;
        // one launch first: its duration sizes the run
        hipEventRecord(ea, st);
        hipLaunchKernelGGL(mfma_rotating_kernel, dim3(grid), dim3(256), 0, st, d, iters, o, stamps);
        hipEventRecord(eb, st);
        if (hipEventSynchronize(eb) != hipSuccess) break;
        float ms1 = 0.f;
        hipEventElapsedTime(&ms1, ea, eb);
        if (!(ms1 > 0.f)) break;
        int reps = (int)(seconds * 1e3 / ms1);
        reps = reps < 12 ? 12 : (reps > 100000 ? 100000 : reps);
        const int timed = 10;
        for (int r = 0; r < reps; ++r) {
            if (r == reps - timed) hipEventRecord(ea, st);
            hipLaunchKernelGGL(mfma_rotating_kernel, dim3(grid), dim3(256), 0, st, d, iters, o, stamps);
        }
        hipEventRecord(eb, st);
        if (hipEventSynchronize(eb) != hipSuccess) break;
        float ms = 0.f;
        hipEventElapsedTime(&ms, ea, eb);
        unsigned long long hs[2] = {0, 0};
        if (hipMemcpy(hs, stamps, 16, hipMemcpyDeviceToHost) != hipSuccess) break;
        *tflops_out = flops * timed / (ms * 1e-3) / 1e12;
        if (clock_mhz_out) *clock_mhz_out = hs[1] ? (double)hs[0] / (double)hs[1] * 100.0 : 0.0;
        rc = BD_OK;
    } while (0);
    if (ea) hipEventDestroy(ea);
    if (eb) hipEventDestroy(eb);
    if (d) hipFree(d);
    if (o) hipFree(o);
    if (stamps) hipFree(stamps);
    if (rc != BD_OK) bd_set_error("bd_probe_mfma_rate: %s", hipGetErrorString(hipGetLastError()));
    return rc;
}

int bd_pp_clk_read(unsigned long long* out2, int reset);          // conv3x3_pp.hip
int bd_rk_clk_read(unsigned long long* out2, int reset);          // conv_wgrad3x3_ring.hip

// SYNCHRONOUS with the device (hipMemcpyFromSymbol): the clock the chip held, averaged over every launch of `kernel` since the last reset.
extern "C" int bd_probe_kernel_clock(const char* kernel, int reset, double* clock_mhz_out, double* busy_ms_out) {
    BD_REQUIRE(kernel && clock_mhz_out, "bd_probe_kernel_clock: null pointer");
    unsigned long long v[2] = {0, 0};
    int rc;
    if (std::string(kernel) == "conv3x3_pp_kernel") rc = bd_pp_clk_read(v, reset);
    else if (std::string(kernel) == "conv_wgrad3x3_ring_kernel") rc = bd_rk_clk_read(v, reset);
    else { bd_set_error("bd_probe_kernel_clock: no probe in '%s' (conv3x3_pp_kernel, conv_wgrad3x3_ring_kernel)", kernel); return BD_EINVAL; }
    if (rc != 0) { bd_set_error("bd_probe_kernel_clock: %s", hipGetErrorString(hipGetLastError())); return BD_ELAUNCH; }
    *clock_mhz_out = v[1] ? (double)v[0] / (double)v[1] * 100.0 : 0.0;
    if (busy_ms_out) *busy_ms_out = (double)v[1] * 1e-5;
    return BD_OK;
}
