// What does the SHAPE of a wave-instruction cost?  A 1 KiB buffer_load_dwordx4 is cut into 1024 / SEG rows of SEG bytes (SEG = 64: the K step
// of conv1x1_dense_kernel, 16 pixels x 32 channels; 128 / 256: longer K steps; 1024: one contiguous piece).  Consecutive instructions of a
// wave walk along the rows (row length RL bytes), as K steps do, then move to the next block of rows.  Source: a region every workgroup
// re-reads (L2) or private slices of 2 GB (HBM).   hipcc -O2 --offload-arch=gfx950 ingest_seg.hip -o ingest_seg && ./ingest_seg
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int INFLIGHT>
__global__ __launch_bounds__(256) void ingest_seg(const unsigned char* src, unsigned long long per_wg, unsigned region, int seg, int rl,
                                                  int pieces_per_wave, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src) + (size_t)blockIdx.x * per_wg, 0, region, 0x00020000);
    const int lanes_per_row = seg / 16, rows = 1024 / seg;
    const unsigned lane_off = (unsigned)(lane / lanes_per_row) * rl + (lane % lanes_per_row) * 16;
    const unsigned block_bytes = (unsigned)rows * rl;                 // one block of rows, all K steps
    const int steps = rl / seg;
    unsigned blk = (wave + (per_wg ? 0 : blockIdx.x * 7)) % (region / block_bytes);
    int step = 0;
    u32x4_t acc = {0, 0, 0, 0};
    for (int i = 0; i < pieces_per_wave; i += INFLIGHT) {
        u32x4_t v[INFLIGHT];
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) {
            v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, blk * block_bytes + step * seg + lane_off, 0, 0);
            if (++step == steps) { step = 0; blk += nw; if ((blk + 1) * block_bytes > region) blk = wave; }
        }
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) acc ^= v[k];
    }
    if (acc[0] == 0x12345 && acc[1] == 7) sink[0] = acc[2] + acc[3];
}

int main() {
    const size_t total = 2ull << 30;
    unsigned char* buf; unsigned* sink;
    hipMalloc(&buf, total); hipMalloc(&sink, 64);
    hipMemset(buf, 1, total);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int hbm = 0; hbm < 2; ++hbm)
        for (int rl : {512, 2048})
            for (int seg : {64, 128, 256, 1024})
                for (int wgs : {2, 4})
                    for (int inflight : {4, 8}) {
                        if (seg > rl) continue;
                        const int grid = 256 * wgs;
                        const unsigned long long per_wg = hbm ? total / grid : 0;
                        const unsigned region = hbm ? (unsigned)per_wg : (4u << 20);
                        const int pieces_per_wave = (hbm ? 4 : 16) * 1024 / (4 * wgs) / 8 * 8;       // 4 / 16 MB per CU
                        float ms = 0;
                        for (int rep = 0; rep < 2; ++rep) {
                            hipEventRecord(a);
                            if (inflight == 4) hipLaunchKernelGGL(ingest_seg<4>, dim3(grid), dim3(256), 0, 0, buf, per_wg, region, seg, rl, pieces_per_wave, sink);
                            else hipLaunchKernelGGL(ingest_seg<8>, dim3(grid), dim3(256), 0, 0, buf, per_wg, region, seg, rl, pieces_per_wave, sink);
                            hipEventRecord(b); hipEventSynchronize(b);
                            hipEventElapsedTime(&ms, a, b);
                        }
                        const double bytes_per_cu = (double)pieces_per_wave * 4 * wgs * 1024.0;
                        printf("%s rows of %4d B, segments of %4d B, %d WG/CU x 4 waves x %d in flight: %6.1f GB/s per CU, %5.2f TB/s chip\n",
                               hbm ? "HBM" : "L2 ", rl, seg, wgs, inflight, bytes_per_cu / (ms * 1e-3) / 1e9, 256 * bytes_per_cu / (ms * 1e-3) / 1e12);
                    }
    return 0;
}
