// orbfe_pageable.h -- copies between CALLER memory (which may be pageable) and the device, for the few places where the
// library does not stage through page-locked memory of its own anyway (fallbacks when a thread's arena is too small, results
// larger than the arena's mirror, the vocabulary upload, orbfe_get_rays, the ring matching's download).
//
// A large copy between pageable memory and the device, handed to the HIP runtime as it is, makes the runtime pin the caller's
// pages on the fly and move the data through that mapping.  Round 6 saw a process die in exactly such a copy ("Memory access
// fault by GPU ... on address <a host heap address>"), and round 5's unexplained abort happened inside one made
// by another library.  Whatever the mechanism, the library does not depend on it: caller memory is only ever touched by the
// host, the device only by copies from / to page-locked memory this thread owns, in pieces of kChunk bytes.
#ifndef ORBFE_PAGEABLE_H
#define ORBFE_PAGEABLE_H
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>

namespace orbfe_pageable {
constexpr size_t kChunk = 1u << 20;
struct Bounce { // page-locked memory of one thread: allocated at first use, given back when the thread ends
    uint8_t* p = nullptr;
    ~Bounce()
    {
        if (p) (void)hipHostFree(p);
    }
};
inline uint8_t* bounce()
{
    static thread_local Bounce b;
    if (!b.p && hipHostMalloc((void**)&b.p, kChunk) != hipSuccess) {
        (void)hipGetLastError();
        b.p = nullptr;
    }
    return b.p;
}
// host -> device; returns when `host` has been read in full (the copies are queued on `s` and waited for)
inline hipError_t up(void* dev, const void* host, size_t bytes, hipStream_t s)
{
    if (!bytes) return hipSuccess;
    uint8_t* const b = bounce();
    if (!b) return hipErrorOutOfMemory;
    for (size_t o = 0; o < bytes; o += kChunk) {
        const size_t n = std::min(kChunk, bytes - o);
        std::memcpy(b, static_cast<const uint8_t*>(host) + o, n);
        hipError_t e = hipMemcpyAsync(static_cast<uint8_t*>(dev) + o, b, n, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s); // (the bounce buffer is refilled next)
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
// device -> host; returns when `host` is filled (everything queued on `s` before it has finished)
inline hipError_t down(void* host, const void* dev, size_t bytes, hipStream_t s)
{
    if (!bytes) return hipStreamSynchronize(s);
    uint8_t* const b = bounce();
    if (!b) return hipErrorOutOfMemory;
    for (size_t o = 0; o < bytes; o += kChunk) {
        const size_t n = std::min(kChunk, bytes - o);
        hipError_t e = hipMemcpyAsync(b, static_cast<const uint8_t*>(dev) + o, n, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return e;
        std::memcpy(static_cast<uint8_t*>(host) + o, b, n);
    }
    return hipSuccess;
}
} // namespace orbfe_pageable
#endif
