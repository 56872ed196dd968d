// Probe of ds_read_b64_tr_b8 (gfx950): which LDS bytes does lane l receive?  The tile is 16 rows x 64 B with byte (row, col) = row * 16
// + (col & 15) | marker; every lane supplies the address given by a hypothesis and prints its 8 result bytes.
// hipcc --offload-arch=gfx950 scripts/exp/ds_read_tr8_layout.hip -o /tmp/tr8 && /tmp/tr8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) int i32x2;

__global__ void k(uint8_t* out, int hyp) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[32 * 64];
    const int l = threadIdx.x;
    for (int i = l; i < 32 * 64; i += 64) tile[i] = (uint8_t)(((i / 64) << 4) | ((i % 64) & 15));     // (row & 15) << 4 | col & 15
    __syncthreads();
    const int idx = l & 15, g = l >> 4;
    int row, col;
    if (hyp == 0) { row = idx >> 1; col = 8 * (idx & 1); }            // lane 2q+p -> row q, bytes 8p..8p+7
    else if (hyp == 1) { row = idx & 7; col = 8 * (idx >> 3); }       // lane 8p+q
    else { row = idx; col = 0; }                                      // 16 rows x 8 bytes
    const uint8_t* a = tile + (row + 0) * 64 + col + 16 * g;          // each 16-lane group its own 16-column block (marker = col&15 only)
    i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)a);
    uint8_t b[8];
    __builtin_memcpy(b, &v, 8);
    for (int j = 0; j < 8; ++j) out[l * 8 + j] = b[j];
}

int main() {
    uint8_t* d; hipMalloc(&d, 512);
    uint8_t h[512];
    for (int hyp = 0; hyp < 3; ++hyp) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, hyp);
        hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("hypothesis %d: lane -> 8 x (row,col)\n", hyp);
        for (int l = 0; l < 32; ++l) {
            printf("  lane %2d:", l);
            for (int j = 0; j < 8; ++j) printf(" (%d,%2d)", h[l * 8 + j] >> 4, h[l * 8 + j] & 15);
            printf("\n");
        }
    }
    return 0;
}
