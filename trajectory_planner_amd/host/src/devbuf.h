// devbuf.h — internal to the host facades: RAII device buffers for the staging copies around the C ABI calls.
#ifndef VIGO_HOST_DEVBUF_H
#define VIGO_HOST_DEVBUF_H
#include <hip/hip_runtime_api.h>

#include <cstddef>

namespace vigo_host {

// One non-blocking HIP stream per host thread: the facades bind the handle they drive to it (vigo_set_stream), stage
// their copies on it and wait on it alone, so two host threads planning two batches overlap on the device instead of
// meeting in hipDeviceSynchronize().  nullptr (the default stream) if the stream cannot be created.
inline hipStream_t threadStream() {
    static thread_local hipStream_t s = nullptr;
    static thread_local bool tried = false;
    if (!tried) {
        tried = true;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) s = nullptr;
    }
    return s;
}
inline bool threadSync() { return hipStreamSynchronize(threadStream()) == hipSuccess; }

// RAII device buffer; every HIP failure is reported to the caller as `false`
struct DevBuf {
    void* p = nullptr;
    size_t n = 0;
    bool keep = false;   // thread-lifetime staging buffers: left to the runtime's teardown, not freed after it
    ~DevBuf() { if (p && !keep) (void)hipFree(p); }
    bool upload(const void* src, size_t bytes) {
        if (bytes > n) {
            if (p) (void)hipFree(p);
            p = nullptr;
            n = 0;
            if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) return false;
            n = bytes;
        }
        // (pageable source: staged by the runtime before the call returns; ordered with the thread's stream)
        return bytes == 0 || hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, threadStream()) == hipSuccess;
    }
    bool alloc(size_t bytes) {
        if (bytes <= n && p) return true;
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
        if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) return false;
        n = bytes;
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
