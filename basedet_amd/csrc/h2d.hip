// Host batch -> fp32 device tensor (layers/common/pre_processing.py:13 `Tensor(image)`; the loaders of the reference hand over float64 /
// float32 / uint8 host arrays, utils/dummy.py:60 emits float64).  HOST code only: a persistent pool of worker threads converts the batch
// chunk by chunk into a pinned staging buffer and every chunk leaves with its own asynchronous DMA as soon as it is converted, so the
// conversion of chunk k + 1 runs under the transfer of chunk k.  The arithmetic of the path (pad, normalise) stays in bd_pad_normalize.
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"

namespace {

void cvt_f64(const double* __restrict__ s, float* __restrict__ d, size_t n) {
#pragma clang loop vectorize(enable) interleave(enable)
    for (size_t i = 0; i < n; ++i) d[i] = (float)s[i];
}
__attribute__((target("avx2"))) void cvt_f64_avx2(const double* __restrict__ s, float* __restrict__ d, size_t n) {
#pragma clang loop vectorize(enable) interleave(enable)
    for (size_t i = 0; i < n; ++i) d[i] = (float)s[i];
}
__attribute__((target("avx512f"))) void cvt_f64_avx512(const double* __restrict__ s, float* __restrict__ d, size_t n) {
#pragma clang loop vectorize(enable) interleave(enable)
    for (size_t i = 0; i < n; ++i) d[i] = (float)s[i];
}
void cvt_u8(const uint8_t* __restrict__ s, float* __restrict__ d, size_t n) {
#pragma clang loop vectorize(enable)
    for (size_t i = 0; i < n; ++i) d[i] = (float)s[i];
}

struct Job {
    const void* src = nullptr;
    int dtype = 0;
    size_t n = 0, chunk = 0, nchunks = 0;
    float* dst = nullptr;
    hipStream_t stream = nullptr;
};

}  // namespace

struct bd_h2d {
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    Job job;
    uint64_t generation = 0;
    std::atomic<size_t> next{0};
    size_t done = 0;
    int active = 0;
    bool stop = false;
    int first_error = 0;             // hipError_t of the first failed copy of the current job
    float* pinned = nullptr;
    size_t pinned_elems = 0;
    hipEvent_t drained = nullptr;    // recorded after the last copy of a submit: the staging buffer is free once it has fired
    bool pending = false;
    int device = 0;
    int simd = 0;                    // 0 scalar / autovectorised baseline, 1 avx2, 2 avx512f

    void run_chunks(const Job& j) {
        for (;;) {
            const size_t c = next.fetch_add(1);
            if (c >= j.nchunks) break;
            const size_t lo = c * j.chunk, len = (lo + j.chunk <= j.n) ? j.chunk : j.n - lo;
            float* stage = pinned + lo;
            if (j.dtype == BD_HOST_F64) {
                const double* s = (const double*)j.src + lo;
                if (simd == 2) cvt_f64_avx512(s, stage, len);
                else if (simd == 1) cvt_f64_avx2(s, stage, len);
                else cvt_f64(s, stage, len);
            } else if (j.dtype == BD_HOST_F32) {
                memcpy(stage, (const float*)j.src + lo, len * sizeof(float));
            } else {
                cvt_u8((const uint8_t*)j.src + lo, stage, len);
            }
            const hipError_t e = hipMemcpyAsync(j.dst + lo, stage, len * sizeof(float), hipMemcpyHostToDevice, j.stream);
            if (e != hipSuccess) {
                std::lock_guard<std::mutex> g(mu);
                if (!first_error) first_error = (int)e;
            }
        }
    }

    void worker_main() {
        (void)hipSetDevice(device);
        uint64_t seen = 0;
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
                j = job;
            }
            run_chunks(j);
            {
                std::lock_guard<std::mutex> g(mu);
                if (--active == 0) cv_done.notify_all();
            }
        }
    }
};

extern "C" {

int bd_h2d_create(bd_h2d_t* out, int device, int threads) {
    BD_REQUIRE(out != nullptr, "bd_h2d_create: null argument");
    if (threads <= 0) {
        const unsigned hw = std::thread::hardware_concurrency();
        threads = (int)(hw >= 32 ? 16 : (hw >= 8 ? hw / 2 : 2));
    }
    BD_REQUIRE(threads <= 256, "bd_h2d_create: %d threads", threads);
    // The caller's current device is left alone (a rank with LOCAL_RANK > 0 must not find itself on device 0 afterwards): the event
    // is created under a save / restore, and only the worker threads bind themselves to `device`.
    int ndev = 0, prev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || hipGetDevice(&prev) != hipSuccess) {
        bd_set_error("bd_h2d_create: device %d not available (%d devices)", device, ndev);
        return BD_ELAUNCH;
    }
    struct Restore { int prev, dev; ~Restore() { if (prev != dev) (void)hipSetDevice(prev); } } restore{prev, device};
    if (prev != device && hipSetDevice(device) != hipSuccess) {
        bd_set_error("bd_h2d_create: hipSetDevice(%d) failed", device);
        restore.dev = prev;
        return BD_ELAUNCH;
    }
    bd_h2d* h = new bd_h2d();
    h->device = device;
    h->simd = __builtin_cpu_supports("avx512f") ? 2 : (__builtin_cpu_supports("avx2") ? 1 : 0);
    if (hipEventCreateWithFlags(&h->drained, hipEventDisableTiming) != hipSuccess) {
        bd_set_error("bd_h2d_create: event creation failed");
        delete h;
        return BD_ELAUNCH;
    }
    for (int i = 0; i < threads - 1; ++i) h->workers.emplace_back([h] { h->worker_main(); });   // the submitting thread is the last worker
    *out = h;
    return BD_OK;
}

int bd_h2d_threads(bd_h2d_t h) { return h ? (int)h->workers.size() + 1 : 0; }

int bd_h2d_submit(bd_h2d_t h, const void* src_host, int src_dtype, int64_t n, float* dst_dev, int64_t chunk_elems, bd_stream_t stream) {
    BD_REQUIRE(h != nullptr && n >= 0, "bd_h2d_submit: null handle or negative count");
    BD_REQUIRE(src_dtype == BD_HOST_F64 || src_dtype == BD_HOST_F32 || src_dtype == BD_HOST_U8, "bd_h2d_submit: dtype %d", src_dtype);
    if (n == 0) return BD_OK;             // an empty batch: nothing to move (its pointers may be null)
    BD_REQUIRE(src_host && dst_dev, "bd_h2d_submit: null buffer");
    {   // the submitting thread issues copies too: it must sit on the handle's device (stream and destination belong to it)
        int cur = -1;
        BD_REQUIRE(hipGetDevice(&cur) == hipSuccess && cur == h->device, "bd_h2d_submit: current device %d is not the handle's device %d", cur,
                   h->device);
    }
    if (h->pending) {                 // the previous batch's copies still read the staging buffer
        if (hipEventSynchronize(h->drained) != hipSuccess) {
            bd_set_error("bd_h2d_submit: waiting for the previous transfer failed");
            return BD_ELAUNCH;
        }
        h->pending = false;
    }
    if ((size_t)n > h->pinned_elems) {
        if (h->pinned) (void)hipHostFree(h->pinned);
        h->pinned = nullptr;
        h->pinned_elems = 0;
        if (hipHostMalloc((void**)&h->pinned, (size_t)n * sizeof(float), hipHostMallocDefault) != hipSuccess) {
            bd_set_error("bd_h2d_submit: cannot pin %zu bytes of staging memory", (size_t)n * sizeof(float));
            return BD_EWORKSPACE;
        }
        h->pinned_elems = (size_t)n;
    }
    if (chunk_elems <= 0) chunk_elems = 1 << 20;          // 4 MB of fp32 per DMA: ~75 us on the link, conversion ~0.5 ms per thread
    Job j;
    j.src = src_host; j.dtype = src_dtype; j.n = (size_t)n; j.chunk = (size_t)chunk_elems;
    j.nchunks = (j.n + j.chunk - 1) / j.chunk;
    j.dst = dst_dev; j.stream = (hipStream_t)stream;
    {
        std::lock_guard<std::mutex> g(h->mu);
        h->job = j;
        h->next.store(0);
        h->first_error = 0;
        h->active = (int)h->workers.size();
        ++h->generation;
    }
    h->cv_work.notify_all();
    h->run_chunks(j);
    {
        std::unique_lock<std::mutex> lk(h->mu);
        h->cv_done.wait(lk, [&] { return h->active == 0; });
    }
    if (h->first_error) {
        bd_set_error("bd_h2d_submit: hipMemcpyAsync failed: %s", hipGetErrorString((hipError_t)h->first_error));
        return BD_ELAUNCH;
    }
    if (hipEventRecord(h->drained, (hipStream_t)stream) != hipSuccess) {
        bd_set_error("bd_h2d_submit: event record failed");
        return BD_ELAUNCH;
    }
    h->pending = true;
    return BD_OK;
}

int bd_h2d_destroy(bd_h2d_t h) {
    if (!h) return BD_OK;
    {
        std::lock_guard<std::mutex> g(h->mu);
        h->stop = true;
    }
    h->cv_work.notify_all();
    for (auto& t : h->workers) t.join();
    if (h->pending) (void)hipEventSynchronize(h->drained);
    if (h->drained) (void)hipEventDestroy(h->drained);
    if (h->pinned) (void)hipHostFree(h->pinned);
    delete h;
    return BD_OK;
}

}  // extern "C"
