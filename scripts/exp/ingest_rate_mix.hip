// Is the ~34 GB/s LDS-DMA ceiling of a CU a limit of the DMA path alone?  8 waves per CU: waves 0..(8 - R - 1) stream L2-resident data by
// LDS-DMA, the last R waves by register loads (16 in flight each); both kinds together vs each alone.  build + run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__global__ __launch_bounds__(512) void mix(const unsigned char* shared, unsigned shared_bytes, int reg_waves, int dma_on, int reg_on, int pieces, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(shared), 0, shared_bytes, 0x00020000);
    unsigned off = ((wave * 37 + blockIdx.x * 61) * 1024u) % (shared_bytes - 64 * 1024);
    const bool is_reg = wave >= 8 - reg_waves;
    if (!is_reg) {
        if (!dma_on) return;
        for (int i = 0; i < pieces; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(smem + (wave * 8 + (i & 7)) * 1024), 16, off + lane * 16, 0, 0, 0);
            off += 8 * 1024; if (off + 1024 > shared_bytes) off -= (shared_bytes - 8 * 1024);
            if (i >= 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        if (!reg_on) return;
        u32x4_t acc = {0, 0, 0, 0};
        for (int i = 0; i < pieces; i += 16) {
            u32x4_t v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + lane * 16, 0, 0);
                off += 8 * 1024; if (off + 1024 > shared_bytes) off -= (shared_bytes - 8 * 1024);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) acc ^= v[k];
        }
        if (acc[0] == 0x12345 && acc[1] == 7) sink[0] = acc[2] + acc[3];
    }
}

int main() {
    unsigned char* shared; unsigned* sink;
    hipMalloc(&shared, 64 << 20); hipMalloc(&sink, 64); hipMemset(shared, 1, 64 << 20);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&mix), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int pieces = 4096;
    for (int reg_waves : {2, 4})
        for (int cfg = 0; cfg < 3; ++cfg) {
            const int dma_on = cfg != 1, reg_on = cfg != 0;
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                hipLaunchKernelGGL(mix, dim3(256), dim3(512), 64 * 1024, 0, shared, 4096u * 1024, reg_waves, dma_on, reg_on, pieces, sink);
                hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
            }
            const double dma_b = dma_on ? (8 - reg_waves) * (double)pieces * 1024 : 0, reg_b = reg_on ? reg_waves * (double)pieces * 1024 : 0;
            printf("%d DMA waves %s + %d register waves %s: %6.1f GB/s per CU in total (DMA share %5.1f, register share %5.1f); slower stream sets the time\n",
                   8 - reg_waves, dma_on ? "on " : "off", reg_waves, reg_on ? "on " : "off", (dma_b + reg_b) / (ms * 1e-3) / 1e9, dma_b / (ms * 1e-3) / 1e9, reg_b / (ms * 1e-3) / 1e9);
        }
    return 0;
}
