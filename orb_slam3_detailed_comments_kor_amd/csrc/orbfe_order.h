/*
 * orbfe_order.h -- ordering between the extractor's streams and the matcher's (internal to liborbfe.so).
 *
 * An extractor context works asynchronously on its own stream; a matcher call runs on the calling thread's private
 * stream.  A descriptor array that is still being written by the former must not be read by the latter without an
 * ordering between the two streams.  orbfe_get_device_outputs (the call that hands resident pointers out) records an
 * event on the producing stream and publishes the address ranges it covers; every matcher entry point that is given a
 * device pointer looks the pointer up and makes its stream wait for that event first.  Pointers nobody published
 * (a caller's own buffers) are the caller's to order, as with any HIP API.
 */
#ifndef ORBFE_ORDER_H
#define ORBFE_ORDER_H

#include <hip/hip_runtime.h>
#include <stddef.h>

#include <mutex>
#include <vector>

struct OrbfeProducedRange {
    const unsigned char* lo;
    const unsigned char* hi;
    hipEvent_t ev;
};
/* one registry per process (C++17 inline variables: the two translation units of the library share them) */
inline std::mutex g_orbfeProdMutex;
inline std::vector<OrbfeProducedRange> g_orbfeProduced;

/* [p, p + n) will be complete when `ev` (recorded on the producing stream) has fired.  Replaces ranges that overlap. */
inline void orbfe_producer_publish(const void* p, size_t n, hipEvent_t ev)
{
    if (!p || !n) return;
    const unsigned char* lo = (const unsigned char*)p;
    const unsigned char* hi = lo + n;
    std::lock_guard<std::mutex> lock(g_orbfeProdMutex);
    for (size_t i = 0; i < g_orbfeProduced.size();)
        if (g_orbfeProduced[i].lo < hi && lo < g_orbfeProduced[i].hi) g_orbfeProduced.erase(g_orbfeProduced.begin() + (long)i);
        else i++;
    g_orbfeProduced.push_back(OrbfeProducedRange{lo, hi, ev});
}
/* Forget every range published with `ev` (its context is going away). */
inline void orbfe_producer_retire(hipEvent_t ev)
{
    std::lock_guard<std::mutex> lock(g_orbfeProdMutex);
    for (size_t i = 0; i < g_orbfeProduced.size();)
        if (g_orbfeProduced[i].ev == ev) g_orbfeProduced.erase(g_orbfeProduced.begin() + (long)i);
        else i++;
}
/* If p lies inside a published range: hipStreamWaitEvent(consumer, its event).  Returns 1 when a wait was queued, 0 when the
 * pointer is nobody's published range (the caller's own buffer), < 0 when the wait could not be queued -- the read would then
 * be unordered, so callers hand the error on.  The wait is issued while the registry is locked:
 * orbfe_producer_retire (orbfe_destroy on another thread) takes the same lock before the event is destroyed, so the event is
 * alive for the duration of the call. */
inline int orbfe_producer_wait(const void* p, hipStream_t consumer)
{
    std::lock_guard<std::mutex> lock(g_orbfeProdMutex);
    for (const OrbfeProducedRange& r : g_orbfeProduced)
        if ((const unsigned char*)p >= r.lo && (const unsigned char*)p < r.hi) {
            const hipError_t e = hipStreamWaitEvent(consumer, r.ev, 0);
            return e == hipSuccess ? 1 : -(1000 + (int)e);
        }
    return 0;
}
/* The same for a call that is handed device pointers it cannot see one by one (orbfe_bfknn2_frames_device: the job table lives
 * on the device): wait for EVERY published event.  A handful of entries; each wait is a no-op on the device once its event
 * has fired. */
inline int orbfe_producer_wait_all(hipStream_t consumer)
{
    std::lock_guard<std::mutex> lock(g_orbfeProdMutex);
    hipEvent_t last = nullptr;
    for (const OrbfeProducedRange& r : g_orbfeProduced) {
        if (r.ev == last) continue; // (the three ranges of one context share an event)
        const hipError_t e = hipStreamWaitEvent(consumer, r.ev, 0);
        if (e != hipSuccess) return -(1000 + (int)e);
        last = r.ev;
    }
    return 0;
}

#endif
