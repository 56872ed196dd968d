// Probe of the operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands (gfx950): which (row, k) does byte j of lane l hold?
// One-hot probes: A = 1.0 at a single (lane, byte), B = per-(col, k) distinct small integers -> the output row/col/value identify (i, k).
// hipcc --offload-arch=gfx950 scripts/exp/mfma_fp8_layout.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void k(const uint8_t* a, const uint8_t* b, float* c, int sa, int sb) {
    const int l = threadIdx.x;
    i32x8 av, bv;
    memcpy(&av, a + l * 32, 32);
    memcpy(&bv, b + l * 32, 32);
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 4; ++r) c[l * 4 + r] = acc[r];
}

static uint8_t e4m3(float v) {   // exact for the small values used here: 0, 1, 2, 3, 4, 6, 8, 0.5 ...
    if (v == 0) return 0;
    int e = 0; float m = v;
    while (m >= 2) { m /= 2; ++e; }
    while (m < 1) { m *= 2; --e; }
    int mant = (int)((m - 1) * 8 + 0.5f);
    return (uint8_t)(((e + 7) << 3) | mant);
}

int main() {
    uint8_t *da, *db; float* dc;
    hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dc, 64 * 4 * 4);
    std::vector<uint8_t> A(64 * 32), B(64 * 32);
    std::vector<float> C(256);
    // hypothesis H1: lane l, byte j  <->  A[row l&15][k = 32*(l>>4) + j],  B[k = 32*(l>>4) + j][col l&15]
    // Check: A[i][k] = (i == i0 && k == k0), B[k][j] = 1 + (k % 7) for all j ... instead use a direct random-integer GEMM test:
    srand(1);
    float Af[16][128], Bf[128][16];
    const float vals[4] = {0.f, 1.f, 2.f, 0.5f};
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 128; ++kk) Af[i][kk] = vals[rand() & 3];
    for (int kk = 0; kk < 128; ++kk) for (int j = 0; j < 16; ++j) Bf[kk][j] = vals[rand() & 3];
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) {
        A[l * 32 + j] = e4m3(Af[l & 15][32 * (l >> 4) + j]);
        B[l * 32 + j] = e4m3(Bf[32 * (l >> 4) + j][l & 15]);
    }
    hipMemcpy(da, A.data(), A.size(), hipMemcpyHostToDevice);
    hipMemcpy(db, B.data(), B.size(), hipMemcpyHostToDevice);
    for (int sc = 0; sc < 2; ++sc) {
        const int s = sc == 0 ? 0x7f7f7f7f : 0x80808080;     // E8M0 127 = 2^0, 128 = 2^1
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, s, 0x7f7f7f7f);
        hipMemcpy(C.data(), dc, 1024, hipMemcpyDeviceToHost);
        double err = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            const int col = l & 15, row = (l >> 4) * 4 + r;       // C/D map of the 16x16 shapes
            double ref = 0;
            for (int kk = 0; kk < 128; ++kk) ref += (double)Af[row][kk] * Bf[kk][col];
            err += fabs(C[l * 4 + r] - ref * (sc ? 2.0 : 1.0));
        }
        printf("H1 (lane l byte j = k 32*(l>>4)+j), scale_a %s: sum |err| = %g  (C[0] = %g)\n", sc ? "2^1" : "2^0", err, C[0]);
    }
    return 0;
}
