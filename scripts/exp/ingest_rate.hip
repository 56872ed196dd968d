// What can ONE CU take in through LDS-DMA?  256 persistent 512-thread workgroups, a ring of `depth` KiB in flight each (1 KiB pieces), no
// arithmetic: (a) every workgroup re-reads the same `shared_kb` (L2-resident: the weights of a 1x1 convolution), (b) every workgroup streams
// its own slice of a 2 GB buffer (HBM), (c) two shared pieces per private piece (the mix of conv1x1_dense_kernel's K loop at 128 x 256 tiles).
// build: hipcc -O3 --offload-arch=gfx950 scripts/exp/ingest_rate.hip -o /tmp/ingest_rate ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;

__global__ __launch_bounds__(512) void ingest(const unsigned char* shared, unsigned shared_bytes, const unsigned char* priv, unsigned priv_per_wg,
                                               int mode, int depth_pieces, int total_pieces, int stagger) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(shared), 0, shared_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(priv) + (size_t)blockIdx.x * priv_per_wg, 0, priv_per_wg, 0x00020000);
    // each wave issues its own pieces: piece i of this wave -> LDS slot (i % depth_per_wave)
    const int dpw = depth_pieces / 8;                 // ring slots per wave
    const int npw = total_pieces / 8;                 // pieces per wave
    unsigned s_off = ((wave * 37 + (stagger ? blockIdx.x * 61 : 0)) * 1024u) % (shared_bytes - 8 * 1024), p_off = wave * 1024;
    for (int i = 0; i < npw; ++i) {
        const unsigned lds = (unsigned)(wave * dpw + (i % dpw)) * 1024;
        const bool use_shared = mode == 0 || (mode == 2 && (i % 3) != 2);
        if (use_shared) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(s_rsrc, (lds_void_t*)(smem + lds), 16, s_off + lane * 16, 0, 0, 0);
            s_off += 8 * 1024; if (s_off + 1024 > shared_bytes) s_off -= (shared_bytes - 8 * 1024);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(p_rsrc, (lds_void_t*)(smem + lds), 16, p_off + lane * 16, 0, 0, 0);
            p_off += 8 * 1024; if (p_off + 1024 > priv_per_wg) p_off = wave * 1024;
        }
        // keep dpw - 1 pieces in flight per wave
        if (i + 1 >= dpw) {
            switch (dpw) {
                case 2: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                case 12: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
                case 16: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main() {
    const size_t priv_total = 2ull << 30;
    unsigned char *shared, *priv;
    hipMalloc(&shared, 64 << 20); hipMalloc(&priv, priv_total);
    hipMemset(shared, 1, 64 << 20); hipMemset(priv, 2, priv_total);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&ingest), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int grid = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int stagger = 0; stagger < 2; ++stagger)
    for (int mode = 0; mode < 3; mode += 2)
        for (unsigned shared_kb : {512u, 4096u})
            for (int depth_kb : {32, 64}) {
                if (mode == 1 && shared_kb != 512u) continue;
                const int depth_pieces = depth_kb, total_pieces = 8 * 1024 * 4;        // 32 MB per workgroup
                const unsigned priv_per_wg = (unsigned)(priv_total / grid);
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(a);
                    hipLaunchKernelGGL(ingest, dim3(grid), dim3(512), 160 * 1024, 0, shared, shared_kb * 1024, priv, priv_per_wg, mode, depth_pieces, total_pieces, stagger);
                    hipEventRecord(b); hipEventSynchronize(b);
                    float ms; hipEventElapsedTime(&ms, a, b);
                    if (rep == 1)
                        printf("stagger %d mode %d (%s) shared %4u KB in flight %3d KB per CU: %7.1f GB/s per CU, %6.2f TB/s chip\n", stagger, mode,
                               mode == 0 ? "all shared" : (mode == 1 ? "all private/HBM" : "2 shared : 1 private"), shared_kb, depth_kb,
                               total_pieces * 1024.0 / (ms * 1e-3) / 1e9, grid * total_pieces * 1024.0 / (ms * 1e-3) / 1e12);
                }
            }
    return 0;
}
