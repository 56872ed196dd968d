// LDS read issue rates on gfx950: bytes per clock per CU for ds_read_b64, ds_read_b128, ds_read_b64_tr_b16 (conflict-free addresses),
// 4 / 8 / 16 waves per CU.   hipcc -O3 --offload-arch=gfx950 scripts/exp/lds_read_rate.hip -o /tmp/lds_rate && /tmp/lds_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

template <int KIND>
__global__ void rate_kernel(unsigned long long* out, int iters, int pitch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = i;
    __syncthreads();
    // KIND 0: b64 linear (lane * 8); 1: b128 linear (lane * 16); 2: tr_b16 with the wgrad pattern (8 rows x 32 B per half-wave, pitch bytes)
    int off;
    if (KIND == 0) off = lane * 8 + (wave & 3) * 512;
    else if (KIND == 1) off = lane * 16 + (wave & 3) * 1024;
    else { const int g4 = lane >> 4, idx = lane & 15; off = ((2 * (g4 >> 1)) * 10 + 4 * (g4 & 1) + (idx >> 2)) * pitch + ((wave & 3) * 16 + 4 * (idx & 3)) * 2; }
    unsigned acc = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const unsigned char* a = smem + off + (u & 7) * (KIND == 2 ? pitch : 2048);
            if (KIND == 0) { u32x2_t v = *reinterpret_cast<const volatile u32x2_t*>(a); acc += v[0] ^ v[1]; }
            else if (KIND == 1) { u32x4_t v = *reinterpret_cast<const volatile u32x4_t*>(a); acc += v[0] ^ v[3]; }
            else { s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a)); acc += (unsigned)v[0] ^ (unsigned)v[3]; }
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 0x12345u) out[1000 + blockIdx.x] = acc;
}

template <int KIND> void run(const char* name, int waves, int pitch) {
    unsigned long long* d; hipMalloc(&d, 8 * 4096);
    const int iters = 2000;
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256), dim3(64 * waves), 65536, 0, d, iters, pitch);
    hipDeviceSynchronize();
    unsigned long long h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0; for (int i = 0; i < 256; ++i) cyc += h[i]; cyc /= 256;
    const double bytes = (double)iters * 16 * waves * 64 * (KIND == 1 ? 16 : 8);
    // s_memtime counts at 100 MHz on gfx950 (constant clock): report per-ns, then per 2.4 GHz clock
    printf("%-28s waves %2d pitch %3d: %8.0f ticks  %7.1f B/tick per CU\n", name, waves, pitch, cyc, bytes / cyc);
    hipFree(d);
}

int main() {
    for (int w : {4, 8, 16}) {
        run<0>("ds_read_b64", w, 0);
        run<1>("ds_read_b128", w, 0);
        run<2>("ds_read_b64_tr_b16", w, 160);
        run<2>("ds_read_b64_tr_b16", w, 144);
        run<2>("ds_read_b64_tr_b16", w, 128);
    }
    return 0;
}
