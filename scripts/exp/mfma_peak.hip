// What does the matrix pipe sustain on this part?  Back-to-back independent v_mfma_f32_16x16x32_bf16 (and the K = 128 fp8 form) from
// registers only -- no memory, no LDS -- on every SIMD, with all-zero and with random operands (data toggling moves the power-limited clock).
// hipcc --offload-arch=gfx950 -O3 scripts/exp/mfma_peak.hip -o /tmp/mp && /tmp/mp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(256) void k32(const unsigned* seed, int iters, float* out) {     // v_mfma_f32_32x32x16_bf16, 4 independent accumulators
    const unsigned s = seed[threadIdx.x & 63];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    union { bf16x8 v; unsigned u[4]; } a, b;
    for (int i = 0; i < 4; ++i) { a.u[i] = (s * (2 * i + 1)) & 0x3f803f80u; b.u[i] = (s * (2 * i + 3)) & 0x3f803f80u; }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc[i], 0, 0, 0);
    float t = 0;
    for (int i = 0; i < 4; ++i) t += acc[i][0];
    if (t == 12345.678f) out[0] = t;
}

template <int FP8>
__global__ __launch_bounds__(256) void k(const unsigned* seed, int iters, float* out) {
    const unsigned s = seed[threadIdx.x & 63];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    if (FP8) {
        i32x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (int)(s * (2 * i + 1)) & 0x3f3f3f3f; b[i] = (int)(s * (2 * i + 3)) & 0x3f3f3f3f; }   // small positive e4m3
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    } else {
        union { bf16x8 v; unsigned u[4]; } a, b;
        for (int i = 0; i < 4; ++i) { a.u[i] = (s * (2 * i + 1)) & 0x3f803f80u; b.u[i] = (s * (2 * i + 3)) & 0x3f803f80u; }       // bf16 values in [0, 2)
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a.v), "v"(b.v));   // tied D = C: the builtin form let the compiler rotate the accumulators through v_accvgpr moves (half the rate: an artefact)
    }
    float t = 0;
    for (int i = 0; i < 8; ++i) t += acc[i][0];
    if (t == 12345.678f) out[0] = t;
}

// sustained: the same 16x16x32 loop launched back to back for ~3 s on random operands; throughput of the last launches and the in-kernel
// clock = d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md)
__global__ __launch_bounds__(256) void ksus(const unsigned* seed, int iters, float* out, unsigned long long* stamps) {
    const unsigned s = seed[threadIdx.x & 63];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    union { bf16x8 v; unsigned u[4]; } a, b;
    for (int i = 0; i < 4; ++i) { a.u[i] = (s * (2 * i + 1)) & 0x3f803f80u; b.u[i] = (s * (2 * i + 3)) & 0x3f803f80u; }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a.v), "v"(b.v));
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = r1 - r0; }
    float t = 0;
    for (int i = 0; i < 8; ++i) t += acc[i][0];
    if (t == 12345.678f) out[0] = t;
}

// The register-level pattern of the real kernels, still without LDS or memory: 8 A fragments x 4 B fragments of full-entropy random bf16
// (sign, 7 mantissa bits, exponents 2^-3 .. 2^0), 32 accumulators -- every MFMA sees different operands.  ~3 s sustained.
__global__ __launch_bounds__(256) void krot(const unsigned* seed, int iters, float* out, unsigned long long* stamps) {
    unsigned s = seed[threadIdx.x & 63] ^ (threadIdx.x * 2654435761u);
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    union U { bf16x8 v; unsigned u[4]; } a[8], b[4];
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; const unsigned m = s & 0x807f807fu, e = ((s >> 8) & 0x00030003u) << 7; return m | (0x3e003e00u + e); };
    for (int i = 0; i < 8; ++i) for (int k = 0; k < 4; ++k) a[i].u[k] = rnd();
    for (int j = 0; j < 4; ++j) for (int k = 0; k < 4; ++k) b[j].u[k] = rnd();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a[i].v), "v"(b[j].v));
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = r1 - r0; }
    float t = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) t += acc[i][j][0];
    if (t == 12345.678f) out[0] = t;
}

int main() {
    unsigned h[64], *d; float* o;
    hipMalloc(&d, 256); hipMalloc(&o, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rnd = 0; rnd < 2; ++rnd) {
        for (int i = 0; i < 64; ++i) h[i] = rnd ? (unsigned)rand() * 2654435761u : 0u;
        hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
        for (int fp8 = 0; fp8 < 2; ++fp8)
            for (int wpc : {4, 8, 16}) {                 // waves per CU
                const int iters = 20000, grid = 256 * wpc / 4;
                float best = 1e9;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(a);
                    if (fp8) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, iters, o);
                    else hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d, iters, o);
                    hipEventRecord(b); hipEventSynchronize(b);
                    float ms; hipEventElapsedTime(&ms, a, b);
                    if (rep) best = ms < best ? ms : best;     // the first run includes clock ramp
                }
                const double flops = (double)grid * 4 * iters * 8 * 2.0 * 16 * 16 * (fp8 ? 128 : 32);
                printf("%s operands, %s, %2d waves/CU: %.0f TFLOP/s (%.1f ms)\n", rnd ? "random" : "zero  ", fp8 ? "fp8 K=128" : "bf16 K=32", wpc,
                       flops / (best * 1e-3) / 1e12, best);
            }
        for (int wpc : {4, 8, 16}) {
            const int iters = 20000, grid = 256 * wpc / 4;
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a);
                hipLaunchKernelGGL(k32, dim3(grid), dim3(256), 0, 0, d, iters, o);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (rep) best = ms < best ? ms : best;
            }
            const double flops = (double)grid * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
            printf("%s operands, bf16 32x32x16, %2d waves/CU: %.0f TFLOP/s (%.1f ms)\n", rnd ? "random" : "zero  ", wpc, flops / (best * 1e-3) / 1e12, best);
        }
    }
    {
        for (int i = 0; i < 64; ++i) h[i] = (unsigned)rand() * 2654435761u;
        hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
        unsigned long long* st; hipMalloc(&st, 16);
        const int iters = 20000, grid = 256 * 2;           // 8 waves per CU
        const double flops = (double)grid * 4 * iters * 8 * 2.0 * 16 * 16 * 32;
        float ms = 0;
        for (int rep = 0; rep < 1400; ++rep) {             // ~3 s
            const bool timed = rep >= 1390;
            if (timed) hipEventRecord(a);
            hipLaunchKernelGGL(ksus, dim3(grid), dim3(256), 0, 0, d, iters, o, st);
            if (timed) { hipEventRecord(b); hipEventSynchronize(b); float t; hipEventElapsedTime(&t, a, b); ms = t; }
        }
        unsigned long long hs[2]; hipMemcpy(hs, st, 16, hipMemcpyDeviceToHost);
        printf("sustained ~3 s, random operands, bf16 K=32, 8 waves/CU: %.0f TFLOP/s; in-kernel clock %.0f MHz\n", flops / (ms * 1e-3) / 1e12,
               (double)hs[0] / (double)hs[1] * 100.0);
    }
    {
        for (int i = 0; i < 64; ++i) h[i] = (unsigned)rand() * 2654435761u;
        hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
        unsigned long long* st; hipMalloc(&st, 16);
        for (int wpc : {4, 8}) {
            const int iters = 5000, grid = 256 * wpc / 4;
            const double flops = (double)grid * 4 * iters * 32 * 2.0 * 16 * 16 * 32;
            float ms = 0;
            for (int rep = 0; rep < 1400; ++rep) {
                const bool timed = rep >= 1390;
                if (timed) hipEventRecord(a);
                hipLaunchKernelGGL(krot, dim3(grid), dim3(256), 0, 0, d, iters, o, st);
                if (timed) { hipEventRecord(b); hipEventSynchronize(b); float t; hipEventElapsedTime(&t, a, b); ms = t; }
            }
            unsigned long long hs[2]; hipMemcpy(hs, st, 16, hipMemcpyDeviceToHost);
            printf("sustained, 8 x 4 rotating full-entropy fragments, %d waves/CU: %.0f TFLOP/s (%.2f ms per launch), in-kernel clock %.0f MHz\n", wpc,
                   flops / (ms * 1e-3) / 1e12, ms, (double)hs[0] / (double)hs[1] * 100.0);
        }
    }
    return 0;
}
