// Re-measurement of the LDS-DMA ingest rate of a CU (a first version of this probe, with a run-time ring depth, read 34 GB/s where ingest_rate_mix.hip read 108-117 GB/s for what looked like
// the same stream): one kernel, compile-time in-flight depth, run-time number of issuing waves and shared-region size.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void_t;

template <int INFLIGHT>
__global__ __launch_bounds__(512) void dma(const unsigned char* shared, unsigned shared_bytes, int waves_on, int pieces, int stride_kb, int stagger) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= waves_on) return;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(shared), 0, shared_bytes, 0x00020000);
    unsigned off = ((wave * 37 + (stagger ? blockIdx.x * 61 : 0)) * 1024u) % (shared_bytes - 64 * 1024);
    const unsigned stride = stride_kb * 1024u;
    for (int i = 0; i < pieces; ++i) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(smem + (wave * INFLIGHT + (i % INFLIGHT)) * 1024), 16, off + lane * 16, 0, 0, 0);
        off += stride; if (off + 1024 > shared_bytes) off -= (shared_bytes - 8 * 1024);
        if (i >= INFLIGHT - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(INFLIGHT - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int INFLIGHT>
void run(const unsigned char* shared, unsigned shared_kb, int waves_on, int stride_kb, int stagger, int lds_kb) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int pieces = 4096;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dma<INFLIGHT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(dma<INFLIGHT>, dim3(256), dim3(512), lds_kb * 1024, 0, shared, shared_kb * 1024, waves_on, pieces, stride_kb, stagger);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    }
    printf("shared %6u KB, %d waves x %2d in flight, stride %2d KB, stagger %d, LDS %3d KB: %6.1f GB/s per CU (%5.1f TB/s chip)\n", shared_kb, waves_on, INFLIGHT,
           stride_kb, stagger, lds_kb, waves_on * (double)pieces * 1024 / (ms * 1e-3) / 1e9, 256.0 * waves_on * pieces * 1024 / (ms * 1e-3) / 1e12);
}

int main() {
    unsigned char* shared;
    hipMalloc(&shared, 256u << 20); hipMemset(shared, 1, 256u << 20);
    for (unsigned kb : {512u, 4096u, 32768u, 262144u - 64u})
        for (int waves : {4, 6, 8}) {
            run<8>(shared, kb, waves, 8, 1, 64);
            run<8>(shared, kb, waves, 8, 1, 160);
        }
    run<2>(shared, 4096, 8, 8, 1, 160); run<4>(shared, 4096, 8, 8, 1, 160); run<16>(shared, 4096, 8, 8, 1, 160);
    run<8>(shared, 4096, 8, 8, 0, 160); run<8>(shared, 4096, 8, 64, 1, 160); run<8>(shared, 4096, 8, 1, 1, 160);
    return 0;
}
