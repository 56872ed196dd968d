// Companion of ingest_rate.hip: the same streams taken into REGISTERS (buffer_load_dwordx4, 16 B per lane, 1 KiB per wave-instruction,
// `inflight` loads outstanding per wave) instead of LDS-DMA, with 4 / 8 / 16 waves per CU.  build + run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int INFLIGHT>
__global__ __launch_bounds__(512) void ingest_reg(const unsigned char* shared, unsigned shared_bytes, const unsigned char* priv, unsigned priv_per_wg,
                                                   int mode, int pieces_per_wave, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(shared), 0, shared_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(priv) + (size_t)blockIdx.x * priv_per_wg, 0, priv_per_wg, 0x00020000);
    unsigned s_off = ((wave * 37 + blockIdx.x * 61) * 1024u) % (shared_bytes - 64 * 1024), p_off = wave * 1024;
    u32x4_t acc = {0, 0, 0, 0};
    for (int i = 0; i < pieces_per_wave; i += INFLIGHT) {
        u32x4_t v[INFLIGHT];
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) {
            const bool use_shared = mode == 0 || (mode == 2 && ((i + k) % 3) != 2);
            if (use_shared) {
                v[k] = __builtin_amdgcn_raw_buffer_load_b128(s_rsrc, s_off + lane * 16, 0, 0);
                s_off += nw * 1024; if (s_off + 1024 > shared_bytes) s_off -= (shared_bytes - nw * 1024);
            } else {
                v[k] = __builtin_amdgcn_raw_buffer_load_b128(p_rsrc, p_off + lane * 16, 0, 0);
                p_off += nw * 1024; if (p_off + 1024 > priv_per_wg) p_off = wave * 1024;
            }
        }
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) acc ^= v[k];
    }
    if (acc[0] == 0x12345 && acc[1] == 7) sink[0] = acc[2] + acc[3];
}

int main() {
    const size_t priv_total = 2ull << 30;
    unsigned char *shared, *priv; unsigned* sink;
    hipMalloc(&shared, 64 << 20); hipMalloc(&priv, priv_total); hipMalloc(&sink, 64);
    hipMemset(shared, 1, 64 << 20); hipMemset(priv, 2, priv_total);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 3; ++mode)
        for (int waves : {4, 8, 16})
            for (int wgs_per_cu : {1, 2})
                for (int inflight : {8, 16}) {
                    const int grid = 256 * wgs_per_cu, threads = waves * 64 > 512 ? 512 : waves * 64;
                    if (waves == 16 && wgs_per_cu == 1) continue;
                    const int eff_waves = waves == 16 ? 8 : waves;            // 16 waves per CU = two 8-wave workgroups
                    if (waves == 16 && wgs_per_cu != 2) continue;
                    const int pieces_per_wave = 32 * 1024 / (eff_waves * wgs_per_cu) / 16 * 16;   // ~32 MB per CU
                    const unsigned priv_per_wg = (unsigned)(priv_total / grid);
                    float ms = 0;
                    for (int rep = 0; rep < 2; ++rep) {
                        hipEventRecord(a);
                        if (inflight == 8) hipLaunchKernelGGL(ingest_reg<8>, dim3(grid), dim3(threads), 0, 0, shared, 512u * 1024, priv, priv_per_wg, mode, pieces_per_wave, sink);
                        else hipLaunchKernelGGL(ingest_reg<16>, dim3(grid), dim3(threads), 0, 0, shared, 512u * 1024, priv, priv_per_wg, mode, pieces_per_wave, sink);
                        hipEventRecord(b); hipEventSynchronize(b);
                        hipEventElapsedTime(&ms, a, b);
                    }
                    const double bytes_per_cu = (double)pieces_per_wave * eff_waves * wgs_per_cu * 1024.0;
                    printf("mode %d (%s) %2d waves/WG x %d WG/CU, %2d loads in flight per wave: %7.1f GB/s per CU, %6.2f TB/s chip\n", mode,
                           mode == 0 ? "all shared 512 KB" : (mode == 1 ? "all private/HBM" : "2 shared : 1 private"), eff_waves, wgs_per_cu, inflight,
                           bytes_per_cu / (ms * 1e-3) / 1e9, 256 * bytes_per_cu / (ms * 1e-3) / 1e12);
                }
    return 0;
}
