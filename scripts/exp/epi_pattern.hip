// Does the ACCESS PATTERN of the generic kernel's epilogue (per wave instruction: 16 pixel rows x 64 bytes, 128-channel tiles of a
// wider row visited by different workgroups) limit its operand streams?  out = mask > 0 ? add : 0 over [M][CO] bf16 with
//   mode 0: the epilogue's lane -> element map      mode 1: same tiles, row-coalesced (4 rows x 256 B per wave instruction)
//   mode 2: flat streaming
// build: hipcc --offload-arch=gfx950 -O3 scripts/exp/epi_pattern.hip -o scripts/exp/epi_pattern ; run: epi_pattern M CO
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef unsigned short bf16_raw;

__device__ __forceinline__ u32x4_t f(u32x4_t a, u32x4_t m) {
    u32x4_t o;
    for (int k = 0; k < 4; ++k) {
        unsigned lo = (m[k] << 16) > 0 && !((m[k] << 16) >> 31) ? (a[k] & 0xffffu) : 0u;
        unsigned hi = (m[k] & 0xffff0000u) > 0 && !(m[k] >> 31) ? (a[k] & 0xffff0000u) : 0u;
        o[k] = lo | hi;
    }
    return o;
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const bf16_raw* add, const bf16_raw* mask, bf16_raw* out, int M, int CO, int n_tiles) {
    extern __shared__ unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wc = wave >> 1, wp = wave & 1;
    int bid = blockIdx.x;
    const int nwg = gridDim.x;
    { const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7; bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3); }
    if (MODE == 2) {
        const long long n16 = (long long)M * CO / 8;
        for (int k = 0; k < 8; ++k) {
            const long long i = (long long)bid * 2048 + k * 256 + tid;
            if (i < n16) {
                const u32x4_t a = *reinterpret_cast<const u32x4_t*>(add + i * 8), m = *reinterpret_cast<const u32x4_t*>(mask + i * 8);
                *reinterpret_cast<u32x4_t*>(out + i * 8) = f(a, m);
            }
        }
        return;
    }
    const int tile_m = bid / n_tiles, tile_n = bid - tile_m * n_tiles;
    const int m0 = tile_m * 128, co0 = tile_n * 128;
    if (MODE == 0) {
        const int cb = co0 + wc * 64 + 8 * (lane >> 4);
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wp * 64 + j * 16 + (lane & 15);
            if (m >= M) continue;
            for (int half = 0; half < 2; ++half) {
                const long long idx = (long long)m * CO + cb + 32 * half;
                const u32x4_t a = *reinterpret_cast<const u32x4_t*>(add + idx), mm = *reinterpret_cast<const u32x4_t*>(mask + idx);
                *reinterpret_cast<u32x4_t*>(out + idx) = f(a, mm);
            }
        }
    } else {
        for (int kk = 0; kk < 8; ++kk) {
            const int m = m0 + kk * 16 + (tid >> 4);
            if (m >= M) continue;
            const long long idx = (long long)m * CO + co0 + (tid & 15) * 8;
            const u32x4_t a = *reinterpret_cast<const u32x4_t*>(add + idx), mm = *reinterpret_cast<const u32x4_t*>(mask + idx);
            *reinterpret_cast<u32x4_t*>(out + idx) = f(a, mm);
        }
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 67200, CO = argc > 2 ? atoi(argv[2]) : 1024;
    const size_t n = (size_t)M * CO;
    bf16_raw *a, *m, *o;
    hipMalloc(&a, n * 2); hipMalloc(&m, n * 2); hipMalloc(&o, n * 2);
    hipMemset(a, 0x3f, n * 2); hipMemset(m, 0x3f, n * 2);
    const int m_tiles = (M + 127) / 128, n_tiles = CO / 128;
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int mode = 0; mode < 3; ++mode) {
        const int grid = mode == 2 ? (int)((n / 8 + 2047) / 2048) : m_tiles * n_tiles;
        auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 32768, 0, a, m, o, M, CO, n_tiles);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 32768, 0, a, m, o, M, CO, n_tiles);
            else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 32768, 0, a, m, o, M, CO, n_tiles);
        };
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(s);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e); ms /= 20;
        printf("M=%d CO=%d mode %d: %.1f us  %.2f TB/s\n", M, CO, mode, ms * 1e3, 3.0 * n * 2 / ms / 1e9);
    }
    return 0;
}
