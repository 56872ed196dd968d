// How fast can the CUs pull operand tiles through LDS-DMA, as a function of bytes in flight per CU, piece shape and where the data
// lives (L2 / MALL / HBM)?  Every wave streams pieces of 1 KiB (64 lanes x 16 B: 1024 / ROWB rows of ROWB contiguous bytes, rows `pitch`
// bytes apart -- the staging pattern of the 1x1 kernels) into an LDS ring with STAGES pieces in flight under a counted vmcnt.
// hipcc --offload-arch=gfx950 -O3 scripts/exp/l2_stream.hip -o /tmp/l2s && /tmp/l2s
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void_t;

template <int STAGES, int ROWB>
__global__ __launch_bounds__(256) void stream_kernel(const char* buf, unsigned bytes, int iters, int pitch, unsigned span_mask, int* sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(buf), 0, bytes, 0x00020000);
    constexpr int LPR = ROWB / 16;                       // lanes per row
    const unsigned lane_off = (unsigned)((lane / LPR) * pitch + (lane % LPR) * 16);
    const unsigned rows = 64 / LPR;
    unsigned char* ring = smem + wave * (STAGES * 1024);
    // wave w of block b walks its own sequence of pieces (rows * pitch bytes of address space each), wrapped into the buffer span
    unsigned pos = (unsigned)((blockIdx.x * 4 + wave) * 977u);
    const unsigned step = rows * (unsigned)pitch;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < STAGES; ++s) {
            const unsigned base = ((pos * step) & span_mask);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(ring + s * 1024), 16, base + lane_off, 0, 0, 0);
            pos += gridDim.x * 4;
            if (it > 0 || s == STAGES - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STAGES - 1) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (smem[threadIdx.x] == 123 && iters < 0) sink[0] = 1;
}

template <int STAGES, int ROWB>
double run(const char* d, unsigned bytes, unsigned span, int wgs_per_cu, int pitch, int* sink) {
    const int iters = 64;
    // occupancy by LDS: 160 KB / wgs_per_cu each
    const int lds = (160 * 1024 / wgs_per_cu) & ~1023;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<STAGES, ROWB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int grid = 256 * wgs_per_cu * 4;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((stream_kernel<STAGES, ROWB>), dim3(grid), dim3(256), lds, 0, d, bytes, iters, pitch, span - 1, sink);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    const double moved = (double)grid * 4 * iters * STAGES * 1024.0;
    return moved / (ms * 1e-3) / 1e12;
}

int main() {
    const size_t big = 1ull << 31;                       // 2 GiB
    char* d; hipMalloc(&d, big); hipMemset(d, 1, big);
    int* sink; hipMalloc(&sink, 4);
    printf("TB/s delivered to LDS; bytes in flight per CU = wgs_per_cu x 4 waves x STAGES KiB\n");
    const unsigned spans[3] = {16u << 20, 128u << 20, 1u << 31};
    const char* names[3] = {"16 MB (L2-resident)", "128 MB (MALL)", "2 GB (HBM)"};
    for (int sp = 0; sp < 3; ++sp)
        for (int pitch : {512, 2048}) {
            printf("span %-20s row pitch %4d B\n", names[sp], pitch);
            for (int w : {1, 2, 4}) {
                printf("  %d WG/CU:  64-B rows: S=1 %.2f  S=2 %.2f  S=4 %.2f  S=8 %.2f | 128-B rows: S=2 %.2f  S=4 %.2f  S=8 %.2f | 256-B rows: S=4 %.2f  S=8 %.2f\n", w,
                       run<1, 64>(d, (unsigned)(big - 1), spans[sp], w, pitch, sink), run<2, 64>(d, (unsigned)(big - 1), spans[sp], w, pitch, sink),
                       run<4, 64>(d, (unsigned)(big - 1), spans[sp], w, pitch, sink), run<8, 64>(d, (unsigned)(big - 1), spans[sp], w, pitch, sink),
                       run<2, 128>(d, (unsigned)(big - 1), spans[sp], w, pitch, sink), run<4, 128>(d, (unsigned)(big - 1), spans[sp], w, pitch, sink),
                       run<8, 128>(d, (unsigned)(big - 1), spans[sp], w, pitch, sink),
                       run<4, 256>(d, (unsigned)(big - 1), spans[sp], w, pitch, sink), run<8, 256>(d, (unsigned)(big - 1), spans[sp], w, pitch, sink));
            }
        }
    return 0;
}
