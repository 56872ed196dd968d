// C++ caller of the C ABI (include/basedet_hip.h) with no Python and no torch in the process: the "C++ test / bench driver" of
// SURVEY section 8(b).  It owns its device memory (hipMalloc), calls the entry points the way a non-Python host would, checks
// the reference's own known answers (tests/structures/test_boxes.py:38-46 IoU, tests/layers/test_postprocess.py:13-28 NMS keep list)
// plus a convolution against a host loop, runs one communicator of world size 1 through bd_comm_* (RCCL resolved by dlopen), and
// times a head-shaped convolution.   Build: see basedet_amd/build.py (build_driver);  run: basedet_amd/lib/abi_driver
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "basedet_hip.h"

#define CHECK(x)                                                                              \
    do {                                                                                      \
        int rc__ = (x);                                                                       \
        if (rc__ != 0) {                                                                      \
            fprintf(stderr, "FAILED %s -> %d: %s\n", #x, rc__, bd_last_error_string());       \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)
#define EXPECT(c)                                                     \
    do {                                                              \
        if (!(c)) {                                                   \
            fprintf(stderr, "EXPECT failed: %s (line %d)\n", #c, __LINE__); \
            return 1;                                                 \
        }                                                             \
    } while (0)

static uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
template <typename T>
static T* dev(const std::vector<T>& h) {
    T* d = nullptr;
    hipMalloc(&d, h.size() * sizeof(T) + 16);
    hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}
template <typename T>
static std::vector<T> host(const T* d, size_t n) {
    std::vector<T> h(n);
    hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost);
    return h;
}

int main() {
    hipStream_t st;
    hipStreamCreate(&st);
    printf("bd_version %d\n", bd_version());

    // ---- IoU known answer (tests/structures/test_boxes.py:15-46) -------------------------------------------------------------------
    {
        std::vector<float> b1 = {0, 0, 1, 1, 0, 0, 1, 1};
        std::vector<float> b2 = {0, 0, 1, 1, 0, 0, .5f, 1, 0, 0, 1, .5f, 0, 0, .5f, .5f, .5f, .5f, 1, 1, .5f, .5f, 1.5f, 1.5f};
        float *d1 = dev(b1), *d2 = dev(b2), *out;
        hipMalloc(&out, 12 * 4);
        CHECK(bd_box_pairwise(d1, 2, d2, 6, 0, out, st));
        hipStreamSynchronize(st);
        auto r = host(out, 12);
        const float want[6] = {1.0f, 0.5f, 0.5f, 0.25f, 0.25f, 0.25f / (2 - 0.25f)};
        for (int i = 0; i < 12; ++i) EXPECT(fabsf(r[i] - want[i % 6]) < 1e-6f);
        printf("IoU known answer ok\n");
    }
    // ---- batched NMS known answer (tests/layers/test_postprocess.py:13-28): keep = [0, 3, 4, 2] -------------------------------------
    {
        std::vector<float> boxes = {0, 0, 100, 100, 0, 0, 100.5f, 100, 0, 0, 201, 200.5f, 0, 0, 200.5f, 200.5f, .5f, .5f, 100, 101, .5f, .5f, 120.5f, 120.5f};
        std::vector<float> scores = {0.9f, 0.8f, 0.3f, 0.7f, 0.6f, 0.4f};
        std::vector<int32_t> labels = {1, 1, 1, 2, 2, 2};
        float *db = dev(boxes), *ds = dev(scores);
        int32_t *dl = dev(labels), *keep, *num;
        hipMalloc(&keep, 6 * 4);
        hipMalloc(&num, 4);
        const size_t wsb = bd_nms_workspace_bytes(6);
        void* ws;
        hipMalloc(&ws, wsb + 16);
        CHECK(bd_batched_nms(db, ds, dl, 6, 0.4f, 0, keep, num, ws, wsb, st));
        hipStreamSynchronize(st);
        auto n = host(num, 1);
        auto k = host(keep, 6);
        EXPECT(n[0] == 4 && k[0] == 0 && k[1] == 3 && k[2] == 4 && k[3] == 2);
        printf("NMS known answer ok\n");
    }
    // ---- 1x1 convolution + bias + ReLU against a host loop ---------------------------------------------------------------------------
    {
        const int N = 2, H = 9, W = 13, Cin = 64, Cout = 72, M = N * H * W;
        std::vector<uint16_t> x(M * Cin), wp(Cout * Cin);
        std::vector<float> wf(Cout * Cin), bias(Cout);
        srand(7);
        for (auto& v : x) v = f2bf((rand() % 2001 - 1000) / 1000.f);
        for (auto& v : wf) v = (rand() % 2001 - 1000) / 8000.f;
        for (int i = 0; i < Cout; ++i) bias[i] = (rand() % 200 - 100) / 100.f;
        float* dwf = dev(wf);
        uint16_t *dx = dev(x), *dwp, *dwt, *dy;
        hipMalloc(&dwp, wp.size() * 2);
        hipMalloc(&dwt, wp.size() * 2);
        hipMalloc(&dy, (size_t)M * Cout * 2);
        float* dbias = dev(bias);
        CHECK(bd_weight_pack(dwf, nullptr, dwp, dwt, Cout, 1, Cin, st));
        bd_conv_desc d;
        memset(&d, 0, sizeof(d));
        d.N = N; d.Cin = Cin; d.Cout = Cout; d.R = d.S = 1; d.stride = 1; d.pad = 0; d.nseg = 1;
        d.Hi[0] = d.Ho[0] = H; d.Wi[0] = d.Wo[0] = W; d.in_pix_per_img = d.out_pix_per_img = H * W;
        CHECK(bd_conv2d_fwd(&d, dx, dwp, dbias, nullptr, dy, BD_EPI_RELU, st));
        hipStreamSynchronize(st);
        auto y = host(dy, (size_t)M * Cout);
        auto wq = host(dwp, wp.size());
        double num = 0, den = 0;
        for (int m = 0; m < M; ++m)
            for (int co = 0; co < Cout; ++co) {
                float acc = bias[co];
                for (int k = 0; k < Cin; ++k) acc += bf2f(x[m * Cin + k]) * bf2f(wq[co * Cin + k]);
                acc = acc > 0 ? acc : 0;
                const float got = bf2f(y[(size_t)m * Cout + co]);
                num += (got - acc) * (got - acc);
                den += acc * acc;
            }
        EXPECT(sqrt(num / den) < 5e-3);
        printf("conv 1x1 + bias + relu ok (rel-L2 %.2e)\n", sqrt(num / den));
    }
    // ---- collectives: a communicator of one rank (bd_comm_* over RCCL, no torch in the process) --------------------------------------
    {
        unsigned char id[BD_COMM_ID_BYTES];
        CHECK(bd_comm_unique_id(id));
        bd_comm_t comm = nullptr;
        CHECK(bd_comm_init(&comm, id, 0, 1, 0));
        std::vector<float> g(1 << 16);
        for (size_t i = 0; i < g.size(); ++i) g[i] = (float)(i % 97);
        float* dg = dev(g);
        bd_stream_t prod[1] = {st};
        CHECK(bd_comm_allreduce_async(comm, dg, g.size(), BD_COMM_F32, BD_COMM_SUM, prod, 1));
        CHECK(bd_comm_wait(comm, st));
        CHECK(bd_comm_bcast(comm, dg, g.size(), BD_COMM_F32, 0, st));
        CHECK(bd_comm_allreduce(comm, dg, 2, BD_COMM_F32, BD_COMM_AVG, st));
        hipStreamSynchronize(st);
        auto r = host(dg, g.size());
        for (size_t i = 0; i < g.size(); ++i) EXPECT(r[i] == g[i]);
        EXPECT(bd_comm_world(comm) == 1 && bd_comm_rank(comm) == 0);
        CHECK(bd_comm_destroy(comm));
        printf("bd_comm (world 1) ok\n");
    }
    // ---- timing: the RetinaNet head convolution (256 -> 256, 3x3, five levels, batch 16) ------------------------------------------------
    {
        const int N = 16, C = 256;
        const int Hs[5] = {100, 50, 25, 13, 7}, Ws[5] = {168, 84, 42, 21, 11};
        bd_conv_desc d;
        memset(&d, 0, sizeof(d));
        d.N = N; d.Cin = d.Cout = C; d.R = d.S = 3; d.stride = 1; d.pad = 1; d.nseg = 5;
        int off = 0;
        for (int i = 0; i < 5; ++i) {
            d.Hi[i] = d.Ho[i] = Hs[i]; d.Wi[i] = d.Wo[i] = Ws[i]; d.in_off[i] = d.out_off[i] = off;
            off += Hs[i] * Ws[i];
        }
        d.in_pix_per_img = d.out_pix_per_img = off;
        const size_t elems = (size_t)N * off * C;
        uint16_t *dx, *dy, *dw;
        hipMalloc(&dx, elems * 2); hipMalloc(&dy, elems * 2); hipMalloc(&dw, (size_t)C * 9 * C * 2);
        hipMemset(dx, 0x3c, elems * 2); hipMemset(dw, 0x3b, (size_t)C * 9 * C * 2);
        for (int i = 0; i < 3; ++i) CHECK(bd_conv2d_fwd(&d, dx, dw, nullptr, nullptr, dy, BD_EPI_RELU, st));
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, st);
        const int iters = 10;
        for (int i = 0; i < iters; ++i) CHECK(bd_conv2d_fwd(&d, dx, dw, nullptr, nullptr, dy, BD_EPI_RELU, st));
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * N * off * C * C * 9;
        printf("head conv 256->256 3x3 over 5 levels, batch 16: %.1f us, %.0f TFLOP/s\n", ms / iters * 1e3, flop / (ms / iters * 1e-3) / 1e12);
    }
    printf("abi_driver: all checks passed\n");
    return 0;
}
