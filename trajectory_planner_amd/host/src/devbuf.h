// devbuf.h — internal to the host facades: RAII device buffers for the staging copies around the C ABI calls.
#ifndef VIGO_HOST_DEVBUF_H
#define VIGO_HOST_DEVBUF_H
#include <hip/hip_runtime_api.h>

#include <cstddef>

namespace vigo_host {

// One non-blocking HIP stream per host thread: the facades bind the handle they drive to it (vigo_set_stream), stage
// their copies on it and wait on it alone, so two host threads planning two batches overlap on the device instead of
// meeting in hipDeviceSynchronize().  nullptr (the default stream) if the stream cannot be created.
// The stream belongs to the device that is CURRENT on the calling thread (the facades make their planner's device
// current first, setDevice()): one per (thread, device ordinal), so planners on different GPUs of one process never
// share a stream.
constexpr int kMaxDevices = 64;
inline int currentDevice() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) d = 0;
    return d;
}
inline hipStream_t threadStream() {
    static thread_local hipStream_t s[kMaxDevices] = {};
    static thread_local bool tried[kMaxDevices] = {};
    const int d = currentDevice();
    if (!tried[d]) {
        tried[d] = true;
        if (hipStreamCreateWithFlags(&s[d], hipStreamNonBlocking) != hipSuccess) s[d] = nullptr;
    }
    return s[d];
}
inline bool threadSync() { return hipStreamSynchronize(threadStream()) == hipSuccess; }

// RAII device buffer; every HIP failure is reported to the caller as `false`
struct DevBuf {
    void* p = nullptr;
    size_t n = 0;
    int dev = -1;        // device the allocation lives on: a buffer reused under another current device is re-made there
    bool keep = false;   // thread-lifetime staging buffers: left to the runtime's teardown, not freed after it
    ~DevBuf() { if (p && !keep) (void)hipFree(p); }
    bool upload(const void* src, size_t bytes) {
        const int cur = currentDevice();
        if (bytes > n || (p && dev != cur)) {
            if (p) (void)hipFree(p);
            p = nullptr;
            n = 0;
            if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) return false;
            n = bytes;
            dev = cur;
        }
        // (pageable source: staged by the runtime before the call returns; ordered with the thread's stream)
        return bytes == 0 || hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, threadStream()) == hipSuccess;
    }
    bool alloc(size_t bytes) {
        const int cur = currentDevice();
        if (bytes <= n && p && dev == cur) return true;
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
        if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) return false;
        n = bytes;
        dev = cur;
        return true;
    }
    bool download(void* dst, size_t bytes) const {
        return hipMemcpyAsync(dst, p, bytes, hipMemcpyDeviceToHost, threadStream()) == hipSuccess && threadSync();
    }
};

// thread-lifetime staging buffer (declare `static thread_local`): grows on demand, is reused by every later call
// of the thread — hipMalloc/hipFree per planning round cost more than the round's kernels — and is left to the
// runtime's teardown instead of being freed after it
struct StagingBuf : DevBuf {
    StagingBuf() { keep = true; }
};

}  // namespace vigo_host
#endif  /* VIGO_HOST_DEVBUF_H */
