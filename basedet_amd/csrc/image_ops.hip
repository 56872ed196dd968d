// HBM-bound image / tensor kernels: pad+normalise, max-pool, FPN bilinear merge (+ backward), elementwise
// bf16 helpers, weight packing, column sums (bias gradients), SGD.  All use 16-byte accesses per lane.
#include <stdarg.h>

#include <cmath>
#include "common.h"

// ---- error plumbing (shared by every translation unit) -----------------------------------------------------
static thread_local std::string g_bd_error = "";
// multiProcessorCount of the CURRENT device, cached per device ordinal (common.h)
int bd_num_cus() {
    static std::mutex mu;
    static int cached[256] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 255) return 256;
    std::lock_guard<std::mutex> g(mu);
    if (!cached[dev]) {
        int n = 0;
        cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cached[dev];
}

void bd_set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_bd_error = buf;
}
extern "C" const char* bd_last_error_string(void) { return g_bd_error.c_str(); }
// which kernel the calling thread's last convolution entry point dispatched to (set at every launch site: the dispatch itself, not a mirror)
static thread_local const char* g_bd_last_kernel = "";
void bd_note_kernel(const char* name) { g_bd_last_kernel = name; }
extern "C" const char* bd_conv_last_kernel(void) { return g_bd_last_kernel; }
extern "C" int bd_version(void) { return 100; }

namespace {

constexpr int HALO_Y = 3, HALO_X = 4;

// layers/common/pre_processing.py:11-19: pad with zeros first, then (x - mean) / std
__global__ void pad_normalize_kernel(const float* __restrict__ in, int N, int H, int W, int Hp, int Wp,
                                     float m0, float m1, float m2, float s0, float s1, float s2,
                                     bf16_raw* __restrict__ out) {
    const int Hb = Hp + 2 * HALO_Y, Wb = Wp + 2 * HALO_X;
    const long long total = (long long)N * Hb * Wb;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int xb = (int)(i % Wb);
        const long long t = i / Wb;
        const int yb = (int)(t % Hb);
        const int n = (int)(t / Hb);
        const int y = yb - HALO_Y, x = xb - HALO_X;
        u32x2_t o = {0u, 0u};
        if (y >= 0 && y < Hp && x >= 0 && x < Wp) {
            float v0 = 0.f, v1 = 0.f, v2 = 0.f;
            if (y < H && x < W) {
                const long long base = ((long long)n * 3 * H + y) * W + x;
                v0 = in[base]; v1 = in[base + (long long)H * W]; v2 = in[base + 2ll * H * W];
            }
            o[0] = pack_bf2((v0 - m0) / s0, (v1 - m1) / s1);
            o[1] = pack_bf2((v2 - m2) / s2, 0.f);
        }
        *reinterpret_cast<u32x2_t*>(out + i * 4) = o;
    }
}

__global__ void pad_normalize_nchw_kernel(const float* __restrict__ in, int N, int H, int W, int Hp, int Wp,
                                          float m0, float m1, float m2, float s0, float s1, float s2,
                                          float* __restrict__ out) {
    const long long total = (long long)N * 3 * Hp * Wp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wp);
        long long t = i / Wp;
        const int y = (int)(t % Hp);
        t /= Hp;
        const int c = (int)(t % 3);
        const int n = (int)(t / 3);
        float v = 0.f;
        if (y < H && x < W) v = in[(((long long)n * 3 + c) * H + y) * W + x];
        const float m = c == 0 ? m0 : (c == 1 ? m1 : m2);
        const float s = c == 0 ? s0 : (c == 1 ? s1 : s2);
        out[i] = (v - m) / s;
    }
}

__device__ __forceinline__ u32x4_t max_bf16x8(u32x4_t a, u32x4_t b) {
    u32x4_t r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = pack_bf2(fmaxf(bf_lo(a[k]), bf_lo(b[k])), fmaxf(bf_hi(a[k]), bf_hi(b[k])));
    return r;
}

// M.MaxPool2d(3, 2, 1): NHWC bf16, C % 8 == 0; out-of-range taps are ignored (-inf padding)
__global__ void maxpool3x3s2_kernel(const bf16_raw* __restrict__ x, int N, int H, int W, int C, bf16_raw* __restrict__ y) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int cv = C / 8;
    const long long total = (long long)N * Ho * Wo * cv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        long long t = i / cv;
        const int ox = (int)(t % Wo);
        t /= Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const unsigned int NEG = 0xff80ff80u;  // two bf16 -inf
        u32x4_t m = {NEG, NEG, NEG, NEG};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = 2 * oy - 1 + r;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int ix = 2 * ox - 1 + s;
                if (ix < 0 || ix >= W) continue;
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(x + (((long long)n * H + iy) * W + ix) * C + c * 8);
                m = max_bf16x8(m, v);
            }
        }
        *reinterpret_cast<u32x4_t*>(y + i * 8) = m;
    }
}

// bilinear x2, align_corners=False, border clamped: out index o -> source (o + 0.5)/2 - 0.5
//   even o = 2j : 0.25 * in[j-1] + 0.75 * in[j]     (j-1 clamped to 0  => weight 1 on in[0] at o = 0)
//   odd  o = 2j+1: 0.75 * in[j] + 0.25 * in[j+1]    (j+1 clamped to n-1)
__device__ __forceinline__ void up2_taps(int o, int n, int& i0, int& i1, float& w0, float& w1) {
    const int j = o >> 1;
    if (o & 1) { i0 = j; i1 = j + 1 < n ? j + 1 : n - 1; w0 = 0.75f; w1 = 0.25f; }
    else       { i0 = j - 1 >= 0 ? j - 1 : 0; i1 = j; w0 = 0.25f; w1 = 0.75f; }
}

__global__ void upsample2x_add_fwd_kernel(const bf16_raw* __restrict__ top, long long top_ppi, long long top_off,
                                          bf16_raw* __restrict__ lat, long long lat_ppi, long long lat_off,
                                          int N, int H, int W, int C) {
    const int cv = C / 8, H2 = 2 * H, W2 = 2 * W;
    const long long total = (long long)N * H2 * W2 * cv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        long long t = i / cv;
        const int x = (int)(t % W2);
        t /= W2;
        const int y = (int)(t % H2);
        const int n = (int)(t / H2);
        int y0, y1, x0, x1; float wy0, wy1, wx0, wx1;
        up2_taps(y, H, y0, y1, wy0, wy1);
        up2_taps(x, W, x0, x1, wx0, wx1);
        const bf16_raw* tb = top + ((long long)n * top_ppi + top_off) * C + c * 8;
        const u32x4_t a = *reinterpret_cast<const u32x4_t*>(tb + ((long long)y0 * W + x0) * C);
        const u32x4_t b = *reinterpret_cast<const u32x4_t*>(tb + ((long long)y0 * W + x1) * C);
        const u32x4_t cc = *reinterpret_cast<const u32x4_t*>(tb + ((long long)y1 * W + x0) * C);
        const u32x4_t d = *reinterpret_cast<const u32x4_t*>(tb + ((long long)y1 * W + x1) * C);
        bf16_raw* lp = lat + (((long long)n * lat_ppi + lat_off) + (long long)y * W2 + x) * C + c * 8;
        u32x4_t l = *reinterpret_cast<u32x4_t*>(lp);
        u32x4_t o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float lo = wy0 * (wx0 * bf_lo(a[k]) + wx1 * bf_lo(b[k])) + wy1 * (wx0 * bf_lo(cc[k]) + wx1 * bf_lo(d[k]));
            const float hi = wy0 * (wx0 * bf_hi(a[k]) + wx1 * bf_hi(b[k])) + wy1 * (wx0 * bf_hi(cc[k]) + wx1 * bf_hi(d[k]));
            o[k] = pack_bf2(bf_lo(l[k]) + lo, bf_hi(l[k]) + hi);
        }
        *reinterpret_cast<u32x4_t*>(lp) = o;
    }
}

// gather form of the transpose: every coarse pixel (y, x) collects from fine rows 2y-1 .. 2y+2 (clamped taps)
__device__ __forceinline__ float up2_bwd_weight(int o, int j, int n) {
    // weight of in[j] in out[o] under up2_taps (sums clamped duplicates)
    int i0, i1; float w0, w1;
    up2_taps(o, n, i0, i1, w0, w1);
    float w = 0.f;
    if (i0 == j) w += w0;
    if (i1 == j) w += w1;
    return w;
}

__global__ void upsample2x_add_bwd_kernel(const bf16_raw* __restrict__ dlat, long long lat_ppi, long long lat_off,
                                          bf16_raw* __restrict__ dtop, long long top_ppi, long long top_off,
                                          int N, int H, int W, int C, int accumulate) {
    const int cv = C / 8, H2 = 2 * H, W2 = 2 * W;
    const long long total = (long long)N * H * W * cv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        long long t = i / cv;
        const int x = (int)(t % W);
        t /= W;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const bf16_raw* lb = dlat + ((long long)n * lat_ppi + lat_off) * C + c * 8;
        for (int oy = 2 * y - 1; oy <= 2 * y + 2; ++oy) {
            if (oy < 0 || oy >= H2) continue;
            const float wy = up2_bwd_weight(oy, y, H);
            if (wy == 0.f) continue;
            for (int ox = 2 * x - 1; ox <= 2 * x + 2; ++ox) {
                if (ox < 0 || ox >= W2) continue;
                const float wgt = wy * up2_bwd_weight(ox, x, W);
                if (wgt == 0.f) continue;
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(lb + ((long long)oy * W2 + ox) * C);
#pragma unroll
                for (int k = 0; k < 4; ++k) { acc[2 * k] += wgt * bf_lo(v[k]); acc[2 * k + 1] += wgt * bf_hi(v[k]); }
            }
        }
        bf16_raw* tp = dtop + (((long long)n * top_ppi + top_off) + (long long)y * W + x) * C + c * 8;
        u32x4_t o;
        if (accumulate) {
            const u32x4_t old = *reinterpret_cast<u32x4_t*>(tp);
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = pack_bf2(bf_lo(old[k]) + acc[2 * k], bf_hi(old[k]) + acc[2 * k + 1]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = pack_bf2(acc[2 * k], acc[2 * k + 1]);
        }
        *reinterpret_cast<u32x4_t*>(tp) = o;
    }
}

__global__ void relu_kernel(const bf16_raw* __restrict__ x, bf16_raw* __restrict__ y, long long nv) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(x + i * 8);
        u32x4_t o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pack_bf2(fmaxf(bf_lo(v[k]), 0.f), fmaxf(bf_hi(v[k]), 0.f));
        *reinterpret_cast<u32x4_t*>(y + i * 8) = o;
    }
}

__global__ void relu_bwd_kernel(const bf16_raw* __restrict__ g, const bf16_raw* __restrict__ mask,
                                const bf16_raw* __restrict__ add, bf16_raw* __restrict__ y, long long nv) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        const u32x4_t gv = *reinterpret_cast<const u32x4_t*>(g + i * 8);
        const u32x4_t mv = *reinterpret_cast<const u32x4_t*>(mask + i * 8);
        u32x4_t av = {0u, 0u, 0u, 0u};
        if (add) av = *reinterpret_cast<const u32x4_t*>(add + i * 8);
        u32x4_t o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float lo = (bf_lo(mv[k]) > 0.f ? bf_lo(gv[k]) : 0.f) + bf_lo(av[k]);
            const float hi = (bf_hi(mv[k]) > 0.f ? bf_hi(gv[k]) : 0.f) + bf_hi(av[k]);
            o[k] = pack_bf2(lo, hi);
        }
        *reinterpret_cast<u32x4_t*>(y + i * 8) = o;
    }
}

__global__ void add_kernel(const bf16_raw* __restrict__ a, const bf16_raw* __restrict__ b, bf16_raw* __restrict__ y, long long nv) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        const u32x4_t av = *reinterpret_cast<const u32x4_t*>(a + i * 8);
        const u32x4_t bv = *reinterpret_cast<const u32x4_t*>(b + i * 8);
        u32x4_t o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pack_bf2(bf_lo(av[k]) + bf_lo(bv[k]), bf_hi(av[k]) + bf_hi(bv[k]));
        *reinterpret_cast<u32x4_t*>(y + i * 8) = o;
    }
}

// fp32 master [Cout][RS][Cin] -> bf16 [Cout][RS][Cin] and bf16 [Cin][RS][Cout]
__global__ void weight_pack_kernel(const float* __restrict__ w, const float* __restrict__ row_scale,
                                   bf16_raw* __restrict__ wf, bf16_raw* __restrict__ wd, int Cout, int RS, int Cin) {
    const long long total = (long long)Cout * RS * Cin;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        const long long t = i / Cin;
        const int tap = (int)(t % RS);
        const int co = (int)(t / RS);
        float v = w[i];
        if (row_scale) v *= row_scale[co];
        const bf16_raw b = f2bf(v);
        if (wf) wf[i] = b;
        if (wd) wd[((long long)ci * RS + tap) * Cout + co] = b;
    }
}

// many convs in one launch (after every optimizer step): 32 x 32 (co, ci) tiles per tap, transposed through LDS so that both
// packed copies are written with contiguous runs
__global__ __launch_bounds__(256) void weight_pack_multi_kernel(const bd_pack_desc* __restrict__ descs, int n) {
    __shared__ float tile[32][33];
    int lo = 0, hi = n - 1;
    const int b = blockIdx.x;
    while (lo < hi) {                       // last descriptor with block_start <= b
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block_start <= b) lo = mid; else hi = mid - 1;
    }
    const bd_pack_desc d = descs[lo];
    int t = b - d.block_start;
    const int ci_tiles = (d.Cin + 31) / 32, co_tiles = (d.Cout + 31) / 32;
    const int ci_t = t % ci_tiles; t /= ci_tiles;
    const int co_t = t % co_tiles;
    const int tap = t / co_tiles;
    const float* w = d.w;
    bf16_raw* wf = (bf16_raw*)d.w_fwd;
    bf16_raw* wd = (bf16_raw*)d.w_dgrad;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = co_t * 32 + ty + 8 * r, ci = ci_t * 32 + tx;
        float v = 0.f;
        if (co < d.Cout && ci < d.Cin) {
            const long long i = ((long long)co * d.RS + tap) * d.Cin + ci;
            v = w[i];
            if (d.row_scale) v *= d.row_scale[co];
            if (wf) wf[i] = f2bf(v);
        }
        tile[ty + 8 * r][tx] = v;
    }
    __syncthreads();
    if (wd) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = ci_t * 32 + ty + 8 * r, co = co_t * 32 + tx;
            if (co < d.Cout && ci < d.Cin) wd[((long long)ci * d.RS + tap) * d.Cout + co] = f2bf(tile[tx][ty + 8 * r]);
        }
    }
}

// column sums (bias gradients), atomics-free and reproducible:
// stage 1: block b sums rows b, b+nblocks*rif, ... into partial[b][C] (16-byte loads, LDS reduce over row lanes);
// stage 2 sums the partials in block order.
constexpr int COLSUM_BLOCKS = 1024;
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16_raw* __restrict__ g, long long rows, int C,
                                                             long long cnt, long long ppi, long long off,
                                                             float* __restrict__ partial) {
    __shared__ float red[256 * 8];
    const int cvec = C / 8;
    const int rif = 256 / cvec;           // rows in flight per block
    const int tx = threadIdx.x % cvec, ty = threadIdx.x / cvec;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (ty < rif) {
        for (long long r = (long long)blockIdx.x * rif + ty; r < rows; r += (long long)gridDim.x * rif) {
            const long long img = r / cnt;
            const long long row = img * ppi + off + (r - img * cnt);
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(g + row * C + tx * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc[2 * k] += bf_lo(v[k]); acc[2 * k + 1] += bf_hi(v[k]); }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] = acc[k];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const int vx = c >> 3, k = c & 7;
        float s = 0.f;
        for (int y = 0; y < rif; ++y) s += red[(y * cvec + vx) * 8 + k];
        partial[(long long)blockIdx.x * C + c] = s;
    }
}
// stage 2: block = 8 channels x 32 partial lanes; each lane sums every 32nd partial, LDS tree over the lanes
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int nblocks, int C,
                                                           float* __restrict__ out, int accumulate) {
    __shared__ float red[256];
    const int cl = threadIdx.x & 7, lane = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;
    float s = 0.f;
    if (c < C)
        for (int b = lane; b < nblocks; b += 32) s += partial[(long long)b * C + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (lane == 0 && c < C) {
        float t = 0.f;
        for (int k = 0; k < 32; ++k) t += red[k * 8 + cl];
        out[c] = accumulate ? out[c] + t : t;
    }
}

__global__ void sgd_kernel(float* __restrict__ w, float* __restrict__ v, const float* __restrict__ g, long long n,
                           float lr, float momentum, float wd, float grad_scale) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i * 4 < n; i += (long long)gridDim.x * blockDim.x) {
        const long long e = i * 4;
        if (e + 4 <= n) {
            f32x4_t wv = *reinterpret_cast<f32x4_t*>(w + e);
            f32x4_t vv = *reinterpret_cast<f32x4_t*>(v + e);
            const f32x4_t gv = *reinterpret_cast<const f32x4_t*>(g + e);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gg = gv[k] * grad_scale + wd * wv[k];
                vv[k] = momentum * vv[k] + gg;
                wv[k] -= lr * vv[k];
            }
            *reinterpret_cast<f32x4_t*>(w + e) = wv;
            *reinterpret_cast<f32x4_t*>(v + e) = vv;
        } else {
            for (long long k = e; k < n; ++k) {
                const float gg = g[k] * grad_scale + wd * w[k];
                v[k] = momentum * v[k] + gg;
                w[k] -= lr * v[k];
            }
        }
    }
}

inline int grid_for(long long n, int block = 256, int cap = 4096) {
    long long g = (n + block - 1) / block;
    if (g < 1) g = 1;
    return (int)(g < cap ? g : cap);
}

}  // namespace

extern "C" int bd_pad_normalize(const float* in, int N, int H, int W, int Hp, int Wp, const float* mean3,
                                const float* std3, void* out, bd_stream_t stream) {
    BD_REQUIRE(in && out && mean3 && std3, "pad_normalize: null pointer");
    BD_REQUIRE(Hp >= H && Wp >= W && N > 0, "pad_normalize: padded size smaller than input");
    const long long total = (long long)N * (Hp + 6) * (Wp + 8);
    hipLaunchKernelGGL(pad_normalize_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, N, H, W, Hp, Wp,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], (bf16_raw*)out);
    BD_CHECK_LAUNCH("bd_pad_normalize");
    return BD_OK;
}

extern "C" int bd_pad_normalize_nchw(const float* in, int N, int H, int W, int Hp, int Wp, const float* mean3,
                                     const float* std3, float* out, bd_stream_t stream) {
    BD_REQUIRE(in && out && mean3 && std3, "pad_normalize_nchw: null pointer");
    BD_REQUIRE(Hp >= H && Wp >= W && N > 0, "pad_normalize_nchw: padded size smaller than input");
    const long long total = (long long)N * 3 * Hp * Wp;
    hipLaunchKernelGGL(pad_normalize_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, N, H, W,
                       Hp, Wp, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], out);
    BD_CHECK_LAUNCH("bd_pad_normalize_nchw");
    return BD_OK;
}

extern "C" int bd_maxpool3x3s2_fwd(const void* x, int N, int H, int W, int C, void* y, bd_stream_t stream) {
    BD_REQUIRE(x && y, "maxpool: null pointer");
    BD_REQUIRE(C % 8 == 0, "maxpool: C=%d must be a multiple of 8", C);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)N * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_raw*)x, N, H, W, C, (bf16_raw*)y);
    BD_CHECK_LAUNCH("bd_maxpool3x3s2_fwd");
    return BD_OK;
}

extern "C" int bd_upsample2x_add_fwd(const void* top, int64_t top_ppi, int64_t top_off, void* lateral, int64_t lat_ppi,
                                     int64_t lat_off, int N, int H, int W, int C, bd_stream_t stream) {
    BD_REQUIRE(top && lateral, "upsample2x_add_fwd: null pointer");
    BD_REQUIRE(C % 8 == 0, "upsample2x_add_fwd: C=%d must be a multiple of 8", C);
    const long long total = (long long)N * 4 * H * W * (C / 8);
    hipLaunchKernelGGL(upsample2x_add_fwd_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_raw*)top, (long long)top_ppi, (long long)top_off, (bf16_raw*)lateral,
                       (long long)lat_ppi, (long long)lat_off, N, H, W, C);
    BD_CHECK_LAUNCH("bd_upsample2x_add_fwd");
    return BD_OK;
}

extern "C" int bd_upsample2x_add_bwd(const void* dlat, int64_t lat_ppi, int64_t lat_off, void* dtop, int64_t top_ppi,
                                     int64_t top_off, int N, int H, int W, int C, int accumulate, bd_stream_t stream) {
    BD_REQUIRE(dlat && dtop, "upsample2x_add_bwd: null pointer");
    BD_REQUIRE(C % 8 == 0, "upsample2x_add_bwd: C=%d must be a multiple of 8", C);
    const long long total = (long long)N * H * W * (C / 8);
    hipLaunchKernelGGL(upsample2x_add_bwd_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_raw*)dlat, (long long)lat_ppi, (long long)lat_off, (bf16_raw*)dtop,
                       (long long)top_ppi, (long long)top_off, N, H, W, C, accumulate);
    BD_CHECK_LAUNCH("bd_upsample2x_add_bwd");
    return BD_OK;
}

extern "C" int bd_relu_bf16(const void* x, void* y, int64_t n, bd_stream_t stream) {
    BD_REQUIRE(x && y && n % 8 == 0, "relu_bf16: null pointer or n %% 8 != 0");
    hipLaunchKernelGGL(relu_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)x, (bf16_raw*)y, (long long)(n / 8));
    BD_CHECK_LAUNCH("bd_relu_bf16");
    return BD_OK;
}
extern "C" int bd_relu_bwd_bf16(const void* g, const void* mask, const void* add, void* y, int64_t n, bd_stream_t stream) {
    BD_REQUIRE(g && mask && y && n % 8 == 0, "relu_bwd_bf16: null pointer or n %% 8 != 0");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)g,
                       (const bf16_raw*)mask, (const bf16_raw*)add, (bf16_raw*)y, (long long)(n / 8));
    BD_CHECK_LAUNCH("bd_relu_bwd_bf16");
    return BD_OK;
}
extern "C" int bd_add_bf16(const void* a, const void* b, void* y, int64_t n, bd_stream_t stream) {
    BD_REQUIRE(a && b && y && n % 8 == 0, "add_bf16: null pointer or n %% 8 != 0");
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)a,
                       (const bf16_raw*)b, (bf16_raw*)y, (long long)(n / 8));
    BD_CHECK_LAUNCH("bd_add_bf16");
    return BD_OK;
}

extern "C" int bd_weight_pack(const float* w, const float* row_scale, void* w_fwd, void* w_dgrad, int Cout, int RS,
                              int Cin, bd_stream_t stream) {
    BD_REQUIRE(w && (w_fwd || w_dgrad), "weight_pack: null pointer");
    const long long total = (long long)Cout * RS * Cin;
    hipLaunchKernelGGL(weight_pack_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w, row_scale,
                       (bf16_raw*)w_fwd, (bf16_raw*)w_dgrad, Cout, RS, Cin);
    BD_CHECK_LAUNCH("bd_weight_pack");
    return BD_OK;
}

extern "C" int bd_weight_pack_blocks(int Cout, int RS, int Cin) { return RS * ((Cout + 31) / 32) * ((Cin + 31) / 32); }

extern "C" int bd_weight_pack_multi(const bd_pack_desc* descs_dev, int n, int total_blocks, bd_stream_t stream) {
    BD_REQUIRE(descs_dev && n > 0 && total_blocks > 0, "weight_pack_multi: bad arguments");
    hipLaunchKernelGGL(weight_pack_multi_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, descs_dev, n);
    BD_CHECK_LAUNCH("bd_weight_pack_multi");
    return BD_OK;
}

extern "C" size_t bd_colsum_workspace_bytes(int C) { return (size_t)COLSUM_BLOCKS * (size_t)C * sizeof(float); }
extern "C" int bd_colsum_bf16(const void* g, int N, int64_t pix_per_img, int64_t off, int64_t cnt, int C, float* out,
                              int accumulate, void* ws, size_t ws_bytes, bd_stream_t stream) {
    BD_REQUIRE(g && out && ws && N >= 0 && cnt >= 0, "colsum: null pointer");
    const long long rows = (long long)N * cnt;
    BD_REQUIRE(C > 0 && C % 8 == 0 && C <= 2048, "colsum: C=%d must be a multiple of 8 and <= 2048", C);
    if (ws_bytes < bd_colsum_workspace_bytes(C)) {
        bd_set_error("colsum: workspace %zu < required %zu bytes", ws_bytes, bd_colsum_workspace_bytes(C));
        return BD_EWORKSPACE;
    }
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(COLSUM_BLOCKS), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)g,
                       rows, C, (long long)(cnt > 0 ? cnt : 1), (long long)pix_per_img, (long long)off, (float*)ws);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(C, 8)), dim3(256), 0, (hipStream_t)stream, (const float*)ws,
                       COLSUM_BLOCKS, C, out, accumulate);
    BD_CHECK_LAUNCH("bd_colsum_bf16");
    return BD_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Gradient clipping over the flat gradient arena (engine/trainer.py:57-61 -> basecore clip_grad -> megengine.optimizer.clip_grad_value /
// clip_grad_norm).  `pre_scale` is the reduce-mode factor (1 / world for MEAN) that the SGD launch otherwise applies: the reference clips
// the already averaged gradients, so it is folded in here and the optimizer step then runs with grad_scale = 1.
// ------------------------------------------------------------------------------------------------------------
__global__ void clip_value_kernel(float* __restrict__ g, long long n, float pre_scale, float lo, float hi) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i * 4 < n; i += (long long)gridDim.x * blockDim.x) {
        const long long e = i * 4;
        if (e + 4 <= n) {
            f32x4_t v = *reinterpret_cast<f32x4_t*>(g + e);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fminf(fmaxf(v[k] * pre_scale, lo), hi);
            *reinterpret_cast<f32x4_t*>(g + e) = v;
        } else {
            for (long long k = e; k < n; ++k) g[k] = fminf(fmaxf(g[k] * pre_scale, lo), hi);
        }
    }
}

constexpr int CLIP_BLOCKS = 1024;
// ord_mode 0: sum |g|^p (p = ord), 1: sum g^2, 2: max |g|.  Block b owns one contiguous chunk; fixed-order tree inside the block and a
// fixed-order final pass: bitwise reproducible.  Accumulation in double (37.7 M addends).
__global__ __launch_bounds__(256) void clip_norm_partial_kernel(const float* __restrict__ g, long long n, int ord_mode, float ord,
                                                                double* __restrict__ partial) {
    __shared__ double red[256];
    const long long chunk = ((n + CLIP_BLOCKS - 1) / CLIP_BLOCKS + 3) / 4 * 4;
    const long long lo = (long long)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    double acc = 0.0;
    for (long long e = lo + threadIdx.x * 4ll; e < hi; e += 1024) {
        float v[4];
        if (e + 4 <= hi) {
            const f32x4_t t = *reinterpret_cast<const f32x4_t*>(g + e);
            v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
        } else {
            for (int k = 0; k < 4; ++k) v[k] = e + k < hi ? g[e + k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = fabsf(v[k]);
            if (ord_mode == 2) acc = fmax(acc, (double)a);
            else if (ord_mode == 1) acc += (double)a * (double)a;
            else acc += (double)powf(a, ord);
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = ord_mode == 2 ? fmax(red[threadIdx.x], red[threadIdx.x + s]) : red[threadIdx.x] + red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// out[0] = norm of pre_scale * g, out[1] = the factor the arena is multiplied with: pre_scale * min(1, max_norm / (norm + 1e-6))
__global__ __launch_bounds__(256) void clip_norm_final_kernel(const double* __restrict__ partial, int ord_mode, float ord, float pre_scale,
                                                              float max_norm, float* __restrict__ out) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int b = threadIdx.x; b < CLIP_BLOCKS; b += 256) acc = ord_mode == 2 ? fmax(acc, partial[b]) : acc + partial[b];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = ord_mode == 2 ? fmax(red[threadIdx.x], red[threadIdx.x + s]) : red[threadIdx.x] + red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double nrm = red[0];
        if (ord_mode == 1) nrm = sqrt(nrm);
        else if (ord_mode == 0) nrm = pow(nrm, 1.0 / (double)ord);
        nrm *= (double)fabsf(pre_scale);
        const float nf = (float)nrm;
        out[0] = nf;
        out[1] = pre_scale * fminf(max_norm / (nf + 1e-6f), 1.f);
    }
}

__global__ void scale_by_dev_kernel(float* __restrict__ g, long long n, const float* __restrict__ factor) {
    const float f = factor[0];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i * 4 < n; i += (long long)gridDim.x * blockDim.x) {
        const long long e = i * 4;
        if (e + 4 <= n) {
            f32x4_t v = *reinterpret_cast<f32x4_t*>(g + e);
            v *= f;
            *reinterpret_cast<f32x4_t*>(g + e) = v;
        } else {
            for (long long k = e; k < n; ++k) g[k] *= f;
        }
    }
}

extern "C" int bd_clip_grad_value(float* g, int64_t n, float pre_scale, float lower, float upper, bd_stream_t stream) {
    BD_REQUIRE(g && n >= 0, "clip_grad_value: null pointer");
    BD_REQUIRE(lower <= upper, "clip_grad_value: lower %g > upper %g", (double)lower, (double)upper);
    if (n == 0) return BD_OK;
    hipLaunchKernelGGL(clip_value_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g, (long long)n, pre_scale,
                       lower, upper);
    BD_CHECK_LAUNCH("bd_clip_grad_value");
    return BD_OK;
}

extern "C" size_t bd_clip_grad_norm_workspace_bytes(void) { return CLIP_BLOCKS * sizeof(double) + 2 * sizeof(float); }

extern "C" int bd_clip_grad_norm(float* g, int64_t n, float pre_scale, float max_norm, float ord, float* norm_out, void* ws,
                                 size_t ws_bytes, bd_stream_t stream) {
    BD_REQUIRE(g && ws && n >= 0, "clip_grad_norm: null pointer");
    BD_REQUIRE(ord > 0.f, "clip_grad_norm: ord must be positive (inf allowed), got %g", (double)ord);
    if (ws_bytes < bd_clip_grad_norm_workspace_bytes()) {
        bd_set_error("clip_grad_norm: workspace %zu < required %zu bytes", ws_bytes, bd_clip_grad_norm_workspace_bytes());
        return BD_EWORKSPACE;
    }
    const int mode = std::isinf(ord) ? 2 : (ord == 2.f ? 1 : 0);
    double* partial = (double*)ws;
    float* out = (float*)(partial + CLIP_BLOCKS);
    hipLaunchKernelGGL(clip_norm_partial_kernel, dim3(CLIP_BLOCKS), dim3(256), 0, (hipStream_t)stream, (const float*)g, (long long)n,
                       mode, ord, partial);
    hipLaunchKernelGGL(clip_norm_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)partial, mode, ord, pre_scale,
                       max_norm, out);
    if (n > 0)
        hipLaunchKernelGGL(scale_by_dev_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g, (long long)n,
                           (const float*)(out + 1));
    if (norm_out) hipMemcpyAsync(norm_out, out, sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream);
    BD_CHECK_LAUNCH("bd_clip_grad_norm");
    return BD_OK;
}

extern "C" int bd_sgd_momentum_step(float* w, float* v, const float* g, int64_t n, float lr, float momentum, float wd,
                                    float grad_scale, bd_stream_t stream) {
    BD_REQUIRE(w && v && g && n >= 0, "sgd_momentum_step: null pointer");
    if (n == 0) return BD_OK;
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, w, v, g, (long long)n, lr,
                       momentum, wd, grad_scale);
    BD_CHECK_LAUNCH("bd_sgd_momentum_step");
    return BD_OK;
}
