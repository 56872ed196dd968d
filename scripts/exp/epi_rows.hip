// Does the dense 1x1 kernel's epilogue access shape cost bandwidth?  y[m][c] = relu(add[m][c]) over M x C bf16 with two lane -> address maps of
// a 128-pixel x 128-channel tile (256 threads, 4 waves as 2 channel halves x 2 pixel halves, like conv1x1_dense_kernel):
//   map 0 (today): a wave-instruction covers 16 pixels x 64 contiguous bytes (lane = 16 * cg + r: pixel r, 16-byte chunk cg) -- half lines;
//   map 1: a wave-instruction covers 4 pixels x 256 contiguous bytes (the tile's whole row: two full 128-byte lines per pixel).
// build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 scripts/exp/epi_rows.hip -o /tmp/epi_rows && /tmp/epi_rows
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int MAP>
__global__ __launch_bounds__(256) void epi(const unsigned short* add, unsigned short* y, int M, int C, int n_tiles) {
    const int tile_m = blockIdx.x / n_tiles, tile_n = blockIdx.x % n_tiles;
    const int m0 = tile_m * 128, c0 = tile_n * 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4_t v[8];
    long long idx[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        int m, c;
        if (MAP == 0) {
            const int wc = wave >> 1, wp = wave & 1, j = q >> 1, half = q & 1, cg = lane >> 4, r = lane & 15;
            m = m0 + wp * 64 + j * 16 + r; c = c0 + wc * 64 + 32 * half + 8 * cg;
        } else {
            m = m0 + wave * 32 + q * 4 + (lane >> 4); c = c0 + (lane & 15) * 8;
        }
        idx[q] = (long long)m * C + c;
        v[q] = (m < M) ? *reinterpret_cast<const u32x4_t*>(add + idx[q]) : (u32x4_t){0, 0, 0, 0};
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        u32x4_t o = v[q];
        o[0] &= 0x7fff7fffu; o[1] &= 0x7fff7fffu; o[2] &= 0x7fff7fffu; o[3] &= 0x7fff7fffu;
        const int m = (int)(idx[q] / C);
        if (m < M) *reinterpret_cast<u32x4_t*>(y + idx[q]) = o;
    }
}

int main() {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int C : {256, 512, 1024, 2048}) {
        const int M = C == 2048 ? 16800 : (C == 1024 ? 67200 : (C == 512 ? 268800 : 268800));
        unsigned short *add, *y;
        hipMalloc(&add, (size_t)M * C * 2); hipMalloc(&y, (size_t)M * C * 2);
        hipMemset(add, 1, (size_t)M * C * 2);
        const int n_tiles = C / 128, grid = (M + 127) / 128 * n_tiles;
        for (int map = 0; map < 2; ++map) {
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(a);
                if (map == 0) hipLaunchKernelGGL(epi<0>, dim3(grid), dim3(256), 0, 0, add, y, M, C, n_tiles);
                else hipLaunchKernelGGL(epi<1>, dim3(grid), dim3(256), 0, 0, add, y, M, C, n_tiles);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("M %6d C %4d map %d (%s): %7.1f us  %6.0f GB/s (read + write)\n", M, C, map, map ? "4 px x 256 B per instruction" : "16 px x 64 B per instruction",
                   best * 1e3, 4.0 * M * C / (best * 1e-3) / 1e9);
        }
        hipFree(add); hipFree(y);
    }
    return 0;
}
