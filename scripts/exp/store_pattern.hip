// Store-pattern probe for the patch kernels' epilogue: one wave owns 64 pixels x 128 channels of an NHWC bf16 tensor (512 B per pixel row for 256
// channels) and writes them as the epilogue does -- lane = (pixel column l & 15, channel group l >> 4), 16 instructions of 16 B per lane.
//   pattern 0: the shipped layout: a lane's 8 channels at 8 cg + 32 h  -> per instruction 16 pixels x 64 contiguous bytes
//   pattern 1: 16 channels per lane at 16 cg + 64 h2, two adjacent 16-byte stores -> per instruction pair 16 pixels x 128 contiguous bytes
//   pattern 2: as 0 plus the one-byte twin (8 B per lane);  pattern 3: as 1 plus the twin as one 16-byte store
// Every byte of the tensor is written exactly once per pass.   hipcc -O3 --offload-arch=gfx950 store_pattern.hip -o store_pattern && ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int PAT>
__global__ __launch_bounds__(512) void k(unsigned short* dst, unsigned char* dst8, int tiles, int W) {
    // a workgroup = a 256-channel x (4 patches of 4 x 16 pixels) tile; wave = (channel half wm, patch wp); tiles walk the image rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 2, wp = wave & 3;
    const int cg = lane >> 4, col = lane & 15;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int patch = t * 4 + wp;
        const int pw = W / 16;
        const int by = patch / pw, bx = patch - by * pw;
        const u32x4 v = {(unsigned)t, (unsigned)lane, 3u, 4u};
        if (PAT == 0 || PAT == 2) {
#pragma unroll
            for (int h = 0; h < 4; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const long long idx = ((long long)(by * 4 + j) * W + bx * 16 + col) * 256 + wm * 128 + 8 * cg + 32 * h;
                    *reinterpret_cast<u32x4*>(dst + idx) = v;
                    if (PAT == 2) *reinterpret_cast<u32x2*>(dst8 + idx) = (u32x2){v[0], v[1]};
                }
        } else {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const long long idx = ((long long)(by * 4 + j) * W + bx * 16 + col) * 256 + wm * 128 + 16 * cg + 64 * h2;
                    *reinterpret_cast<u32x4*>(dst + idx) = v;
                    *reinterpret_cast<u32x4*>(dst + idx + 8) = v;
                    if (PAT == 3) *reinterpret_cast<u32x4*>(dst8 + idx) = v;
                }
        }
    }
}

int main() {
    const int W = 1024, Hrows = 704;                       // 720 896 pixels x 256 channels = 369 MB bf16 (+ 185 MB twin): the head tower at batch 32
    const int tiles = (Hrows / 4) * (W / 16) / 4;          // 2 816
    unsigned short* d; unsigned char* d8;
    hipMalloc(&d, (size_t)W * Hrows * 256 * 2); hipMalloc(&d8, (size_t)W * Hrows * 256);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep)
        for (int pat = 0; pat < 4; ++pat) {
            for (int grid : {256, 2816}) {
                float best = 1e9f;
                for (int it = 0; it < 5; ++it) {
                    hipEventRecord(a);
                    if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, d, d8, tiles, W);
                    if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, d, d8, tiles, W);
                    if (pat == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, d, d8, tiles, W);
                    if (pat == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(512), 0, 0, d, d8, tiles, W);
                    hipEventRecord(b); hipEventSynchronize(b);
                    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
                }
                const double bytes = (double)W * Hrows * 256 * (pat >= 2 ? 3 : 2);
                printf("pattern %d grid %4d: %7.1f us  %6.2f TB/s\n", pat, grid, best * 1e3, bytes / best / 1e9);
            }
        }
    return 0;
}
