// hostbench.cpp -- the PCIe-INCLUSIVE rates of the drop-in boundary, measured from a plain C++ caller through the
// C ABI (include/orbfe.h): what SURVEY.md section 8(d) / BASELINE.md section 3 define as the metric (wall time of
// orbfe_extract* including H2D of the images and D2H of keypoints + descriptors).  bench.py runs this program as
// a child process and embeds its one JSON line as the `pcie_inclusive` object; it is never bench.py's `value`.
//
//   hostbench <frames.raw> rows cols nframes nfeatures [device] [first | c5 | matcher | stream]
//
// frames.raw = nframes images of rows x cols bytes (bench.py writes its synthetic frames there).
//   single_pageable / single_pinned : orbfe_extract, one frame per call (mono protocol, reference src/Frame.cc:306)
//   batch_pageable / batch_pinned   : orbfe_extract_batch, all frames per call, blocking
//   batch_pipelined                 : orbfe_extract_batch_submit / _wait, two batches in flight, pinned buffers
//   pcie_floor                      : the same bytes moved with bare hipMemcpyAsync from / to pinned memory
//   stereo_pair                     : the reference's stereo protocol (src/Frame.cc:119-122): two extractors driven
//                                     from two threads started per frame, then Frame::ComputeStereoMatches on the
//                                     device-resident results (orbfe_compute_stereo_matches_resident)
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <functional>
#include <map>
#include <vector>

#include "../include/orbfe.h"

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Stat {
    double mean, p50, p99;
};
static Stat stat_of(std::vector<double>& v)
{
    std::sort(v.begin(), v.end());
    double s = 0;
    for (double x : v) s += x;
    return Stat{s / v.size(), v[v.size() / 2], v[std::min(v.size() - 1, (size_t)(v.size() * 0.99))]};
}

#define CHECK(expr)                                                          \
    do {                                                                     \
        long _r = (long)(expr);                                              \
        if (_r < 0) {                                                        \
            fprintf(stderr, "hostbench: %s failed: %ld\n", #expr, _r);       \
            return 2;                                                        \
        }                                                                    \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// Mode "stream" (round 5): a STREAM of EuRoC stereo frames with 1..4 frames in flight on ONE context
// (orbfe_extract_stereo_pair_submit / _wait on the context's batch lanes): sustained wall time per frame, host images in,
// host keypoints / descriptors / mvuRight / mvDepth out, for pageable and page-locked caller images.  in_flight = 1 is the
// blocking protocol (submit, wait); the blocking call orbfe_extract_stereo_pair is timed beside it.
static int run_stream(const std::vector<uint8_t>& frames, int rows, int cols, int B, int nF, int dev)
{
    const size_t imgBytes = (size_t)rows * cols;
    std::vector<uint8_t> right(imgBytes * B);
    for (int i = 0; i < B; i++)
        for (int y = 0; y < rows; y++) {
            const uint8_t* s = frames.data() + imgBytes * i + (size_t)y * cols;
            uint8_t* d = right.data() + imgBytes * i + (size_t)y * cols;
            memcpy(d, s + 12, cols - 12);
            memcpy(d + cols - 12, s, 12);
        }
    const float bf = 47.90639384423901f, fx = 435.2046959714599f; // Examples/Stereo/EuRoC.yaml
    const int lap2[4] = {0, 0, 0, 0};
    const int nFrames = 600;
    std::string out = "{\"frame\": \"" + std::to_string(cols) + "x" + std::to_string(rows) + "\", \"nfeatures\": " + std::to_string(nF) +
                      ", \"frames\": " + std::to_string(nFrames);
    double kpPerFrame = 0, matchesPerFrame = 0;
    for (int pinned = 0; pinned < 2; pinned++) {
        if (pinned) {
            CHECK(orbfe_host_register((void*)frames.data(), imgBytes * B));
            CHECK(orbfe_host_register(right.data(), imgBytes * B));
        }
        out += std::string(", \"") + (pinned ? "pinned" : "pageable") + "\": {";
        for (int depth = 0; depth <= ORBFE_MAX_LANES; depth++) { // depth 0: the blocking call
            orbfe_ctx* ex = nullptr;
            CHECK(orbfe_create(&ex, nF, 1.2f, 8, 20, 7, dev));
            CHECK(orbfe_set_lanes(ex, std::max(depth, 1)));
            const int cap = orbfe_max_keypoints(ex, rows, cols);
            CHECK(cap);
            struct Out {
                std::vector<uint8_t> k, d;
                std::vector<float> u, z;
                int n[2], mono[2];
            };
            std::vector<Out> ring((size_t)std::max(depth, 1));
            for (auto& o : ring) {
                o.k.resize((size_t)2 * cap * 28);
                o.d.resize((size_t)2 * cap * 32);
                o.u.resize(cap);
                o.z.resize(cap);
            }
            long kp = 0, matches = 0;
            double t0 = 0;
            int inFlight = 0;
            for (int r = -30; r < nFrames; r++) {
                if (r == 0) {
                    while (inFlight > 0) { // (the clock starts on an empty pipeline and stops on one)
                        CHECK(orbfe_extract_stereo_pair_wait(ex));
                        inFlight--;
                    }
                    t0 = now_s();
                }
                const int i = (r + 30) % B;
                Out& o = ring[(size_t)((r + 30) % (int)ring.size())];
                if (depth == 0) {
                    const int m = orbfe_extract_stereo_pair(ex, frames.data() + imgBytes * i, right.data() + imgBytes * i, rows, cols, cols,
                                                            lap2, (orbfe_kp*)o.k.data(), o.d.data(), cap, o.n, o.mono, bf / fx, bf,
                                                            o.u.data(), o.z.data());
                    CHECK(m);
                    if (r >= 0) {
                        kp += o.n[0] + o.n[1];
                        matches += m;
                    }
                    continue;
                }
                if (inFlight == depth) {
                    const int m = orbfe_extract_stereo_pair_wait(ex); // (the oldest frame: the ring entry about to be reused)
                    CHECK(m);
                    inFlight--;
                    if (r >= depth) {
                        kp += o.n[0] + o.n[1];
                        matches += m;
                    }
                }
                CHECK(orbfe_extract_stereo_pair_submit(ex, frames.data() + imgBytes * i, right.data() + imgBytes * i, rows, cols, cols,
                                                       lap2, (orbfe_kp*)o.k.data(), o.d.data(), cap, o.n, o.mono, bf / fx, bf,
                                                       o.u.data(), o.z.data()));
                inFlight++;
            }
            while (inFlight > 0) {
                CHECK(orbfe_extract_stereo_pair_wait(ex));
                inFlight--;
            }
            const double dt = now_s() - t0;
            char buf[256];
            snprintf(buf, sizeof buf, "%s\"%s\": {\"ms_per_frame\": %.4f}", depth ? ", " : "",
                     depth ? ("in_flight_" + std::to_string(depth)).c_str() : "blocking_call", 1e3 * dt / nFrames);
            out += buf;
            if (depth == 0) {
                kpPerFrame = (double)kp / nFrames;
                matchesPerFrame = (double)matches / nFrames;
            }
            orbfe_destroy(ex);
        }
        out += "}";
        if (pinned) {
            (void)orbfe_host_unregister((void*)frames.data());
            (void)orbfe_host_unregister(right.data());
        }
    }
    char tail[256];
    snprintf(tail, sizeof tail, ", \"keypoints_per_frame\": %.1f, \"matches_per_frame\": %.1f}", kpPerFrame, matchesPerFrame);
    out += tail;
    printf("%s\n", out.c_str());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Mode "c5" (BASELINE configs[4]): a fisheye stereo frame -- two 1024 x 1024 images, nFeatures 1500, KannalaBrandt8
// bearing rays fused into the extractor (orbfe_set_kb8), both images in the lapping area, then
// Frame::ComputeStereoFishEyeMatches (knn-2 brute force + ratio test + triangulation, orbfe_stereo_fisheye_matches).
// Two protocols per pair, like the rectified leg: the reference's (two extractors, two threads started per frame,
// src/Frame.cc:119-122 of the stereo-fisheye constructor) and one batched call on one context.
static int run_c5(const std::vector<uint8_t>& frames, int rows, int cols, int B, int nF, int dev)
{
    const size_t imgBytes = (size_t)rows * cols;
    // TUM-VI 512 parameters (Examples/Stereo-Inertial/TUM_512.yaml) scaled to the image size
    const float sc = (float)cols / 512.f;
    const float P[8] = {190.978477f * sc, 190.973307f * sc, 254.931706f * sc, 256.897442f * sc, 0.003482389f, 0.000715034f,
                        -0.002053236f, 0.000202937f};
    const float Rlr[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, tlr[3] = {0.101f, 0.0f, 0.0f}; // (a 10-cm rectified-like baseline)
    orbfe_ctx *exL = nullptr, *exR = nullptr;
    CHECK(orbfe_create(&exL, nF, 1.2f, 8, 20, 7, dev));
    CHECK(orbfe_create(&exR, nF, 1.2f, 8, 20, 7, dev));
    CHECK(orbfe_set_kb8(exL, P));
    CHECK(orbfe_set_kb8(exR, P));
    const int cap = orbfe_max_keypoints(exL, rows, cols);
    CHECK(cap);
    std::vector<float> sigma2(8);
    orbfe_get_scale_tables(exL, nullptr, nullptr, sigma2.data(), nullptr);
    std::vector<uint8_t> right(imgBytes * B);
    const int shift = 40;
    for (int i = 0; i < B; i++)
        for (int y = 0; y < rows; y++) {
            const uint8_t* s = frames.data() + imgBytes * i + (size_t)y * cols;
            uint8_t* d = right.data() + imgBytes * i + (size_t)y * cols;
            memcpy(d, s + shift, cols - shift);
            memcpy(d + cols - shift, s, shift);
        }
    std::vector<orbfe_kp> kps((size_t)2 * cap);
    std::vector<uint8_t> desc((size_t)2 * cap * 32);
    std::vector<float> xyL((size_t)2 * cap), xyR((size_t)2 * cap), depth(cap), p3d((size_t)3 * cap);
    std::vector<int32_t> octL(cap), octR(cap), l2r(cap), r2l(cap);
    const int lapX0 = 0, lapX1 = cols - 1; // everything lies in the lapping area (the usual TUM-VI setting is a sub-range)
    auto match = [&](const orbfe_kp* kL, const uint8_t* dL, int nL, int monoL, const orbfe_kp* kR, const uint8_t* dR, int nR,
                     int monoR) -> int {
        const int sL = nL - monoL, sR = nR - monoR;
        for (int i = 0; i < sL; i++) {
            xyL[2 * i] = kL[monoL + i].x;
            xyL[2 * i + 1] = kL[monoL + i].y;
            octL[i] = kL[monoL + i].octave;
        }
        for (int i = 0; i < sR; i++) {
            xyR[2 * i] = kR[monoR + i].x;
            xyR[2 * i + 1] = kR[monoR + i].y;
            octR[i] = kR[monoR + i].octave;
        }
        return orbfe_stereo_fisheye_matches(dev, dL + (size_t)monoL * 32, xyL.data(), octL.data(), sL, dR + (size_t)monoR * 32,
                                            xyR.data(), octR.data(), sR, P, P, Rlr, tlr, sigma2.data(), 8, l2r.data(), r2l.data(),
                                            depth.data(), p3d.data());
    };
    const int nPairs = 120;
    std::vector<double> lat2, latE2, lat1, latE1;
    double matches = 0;
    long kpTotal = 0;
    for (int r = -8; r < nPairs; r++) { // (a) two contexts, two threads per frame
        const int i = (r + 8) % B;
        int nL = 0, nR = 0, rcL = 0, rcR = 0;
        const double a = now_s();
        std::thread tl([&] { rcL = orbfe_extract(exL, frames.data() + imgBytes * i, rows, cols, cols, lapX0, lapX1, kps.data(), desc.data(), cap, &nL); });
        std::thread tr([&] { rcR = orbfe_extract(exR, right.data() + imgBytes * i, rows, cols, cols, lapX0, lapX1, kps.data() + cap, desc.data() + (size_t)cap * 32, cap, &nR); });
        tl.join();
        tr.join();
        const double b = now_s();
        if (rcL < 0 || rcR < 0) return 2;
        const int m = match(kps.data(), desc.data(), nL, rcL, kps.data() + cap, desc.data() + (size_t)cap * 32, nR, rcR);
        CHECK(m);
        if (r >= 0) {
            lat2.push_back(now_s() - a);
            latE2.push_back(b - a);
            matches += m;
            kpTotal += nL + nR;
        }
    }
    for (int r = -8; r < nPairs; r++) { // (b) one context, both images in one batched call
        const int i = (r + 8) % B;
        const uint8_t* two[2] = {frames.data() + imgBytes * i, right.data() + imgBytes * i};
        const int lap4[4] = {lapX0, lapX1, lapX0, lapX1};
        int n2[2] = {0, 0}, mono2[2] = {0, 0};
        const double a = now_s();
        CHECK(orbfe_extract_batch(exL, 2, two, rows, cols, cols, lap4, kps.data(), desc.data(), cap, n2, mono2));
        const double b = now_s();
        const int m = match(kps.data(), desc.data(), n2[0], mono2[0], kps.data() + cap, desc.data() + (size_t)cap * 32, n2[1], mono2[1]);
        CHECK(m);
        if (r >= 0) {
            lat1.push_back(now_s() - a);
            latE1.push_back(b - a);
        }
    }
    // (c) as (b), the matching on the descriptors where the extractor left them (orbfe_get_device_outputs): what travels to the
    // matching call is 12 bytes per keypoint
    std::vector<double> lat3, latE3;
    for (int r = -8; r < nPairs; r++) {
        const int i = (r + 8) % B;
        const uint8_t* two[2] = {frames.data() + imgBytes * i, right.data() + imgBytes * i};
        const int lap4[4] = {lapX0, lapX1, lapX0, lapX1};
        int n2[2] = {0, 0}, mono2[2] = {0, 0};
        const double a = now_s();
        CHECK(orbfe_extract_batch(exL, 2, two, rows, cols, cols, lap4, kps.data(), desc.data(), cap, n2, mono2));
        const double b = now_s();
        const uint8_t* dDesc = nullptr;
        int dcap = 0, dimgs = 0;
        CHECK(orbfe_get_device_outputs(exL, nullptr, &dDesc, nullptr, &dcap, &dimgs));
        if (dimgs != 2 || dcap != cap) return 2;
        const int m = match(kps.data(), dDesc, n2[0], mono2[0], kps.data() + cap, dDesc + (size_t)cap * 32, n2[1], mono2[1]);
        CHECK(m);
        if (r >= 0) {
            lat3.push_back(now_s() - a);
            latE3.push_back(b - a);
        }
    }
    const Stat s3 = stat_of(lat3), e3 = stat_of(latE3);
    // (d) as (b) with the caller's image buffers page-locked (orbfe_host_register once: a camera driver's ring), like the
    // rectified leg's pinned protocol: no staging copy in front of the upload
    std::vector<double> lat4, latE4;
    if (orbfe_host_register((void*)frames.data(), imgBytes * B) == 0 && orbfe_host_register(right.data(), imgBytes * B) == 0) {
        for (int r = -8; r < nPairs; r++) {
            const int i = (r + 8) % B;
            const uint8_t* two[2] = {frames.data() + imgBytes * i, right.data() + imgBytes * i};
            const int lap4[4] = {lapX0, lapX1, lapX0, lapX1};
            int n2[2] = {0, 0}, mono2[2] = {0, 0};
            const double a = now_s();
            CHECK(orbfe_extract_batch(exL, 2, two, rows, cols, cols, lap4, kps.data(), desc.data(), cap, n2, mono2));
            const double b = now_s();
            const int m = match(kps.data(), desc.data(), n2[0], mono2[0], kps.data() + cap, desc.data() + (size_t)cap * 32, n2[1], mono2[1]);
            CHECK(m);
            if (r >= 0) {
                lat4.push_back(now_s() - a);
                latE4.push_back(b - a);
            }
        }
    }
    (void)orbfe_host_unregister((void*)frames.data());
    (void)orbfe_host_unregister(right.data());
    if (lat4.empty()) {
        lat4.push_back(0);
        latE4.push_back(0);
    }
    const Stat s4 = stat_of(lat4), e4 = stat_of(latE4);
    const Stat s2 = stat_of(lat2), e2 = stat_of(latE2), s1 = stat_of(lat1), e1 = stat_of(latE1);
    printf("{\"config\": \"c5\", \"frame\": \"%dx%d\", \"nfeatures\": %d, \"pairs\": %d, \"keypoints_per_pair\": %.1f, "
           "\"matches_per_pair\": %.1f, "
           "\"two_threads\": {\"protocol\": \"2 contexts with orbfe_set_kb8, 2 threads started per frame, pageable images, then "
           "orbfe_stereo_fisheye_matches on host arrays\", \"ms_per_pair_mean\": %.4f, \"ms_per_pair_p50\": %.4f, "
           "\"ms_per_pair_p99\": %.4f, \"extract_ms_p50\": %.4f, \"keypoints_per_s\": %.0f}, "
           "\"one_call\": {\"protocol\": \"1 context, both images in one orbfe_extract_batch, then orbfe_stereo_fisheye_matches\", "
           "\"ms_per_pair_mean\": %.4f, \"ms_per_pair_p50\": %.4f, \"ms_per_pair_p99\": %.4f, \"extract_ms_p50\": %.4f, "
           "\"keypoints_per_s\": %.0f}, "
           "\"one_call_resident_matching\": {\"protocol\": \"as one_call, orbfe_stereo_fisheye_matches on the descriptors where the "
           "extractor left them (orbfe_get_device_outputs)\", \"ms_per_pair_mean\": %.4f, \"ms_per_pair_p50\": %.4f, "
           "\"ms_per_pair_p99\": %.4f, \"extract_ms_p50\": %.4f}, "
           "\"one_call_pinned\": {\"protocol\": \"as one_call with the caller's image buffers page-locked (orbfe_host_register)\", "
           "\"ms_per_pair_mean\": %.4f, \"ms_per_pair_p50\": %.4f, \"ms_per_pair_p99\": %.4f, \"extract_ms_p50\": %.4f}}\n",
           cols, rows, nF, nPairs, (double)kpTotal / nPairs, matches / nPairs, 1e3 * s2.mean, 1e3 * s2.p50, 1e3 * s2.p99,
           1e3 * e2.p50, kpTotal / (s2.mean * nPairs), 1e3 * s1.mean, 1e3 * s1.p50, 1e3 * s1.p99, 1e3 * e1.p50,
           kpTotal / (s1.mean * nPairs), 1e3 * s3.mean, 1e3 * s3.p50, 1e3 * s3.p99, 1e3 * e3.p50, 1e3 * s4.mean, 1e3 * s4.p50,
           1e3 * s4.p99, 1e3 * e4.p50);
    orbfe_destroy(exL);
    orbfe_destroy(exR);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Mode "matcher" (VERDICT r03 #5a): the matcher entry points from C++, so that the cost of a call is not mixed with
// ctypes marshalling.  Inputs are real: two frames are extracted (frame B = frame A shifted by 7 px), FeatureVectors are the
// nearest of 100 random 256-bit centroids.  For every call: p50 / p99 of the wall time with host arrays, with the
// descriptors left on the device by the extractor (device pointers), and with the keyframe side in a handle.
struct HostFv {
    std::vector<uint32_t> ids;
    std::vector<int32_t> off, ind;
    orbfe_fv view() const { return orbfe_fv{(int)ids.size(), ids.data(), off.data(), ind.data()}; }
};
static HostFv make_fv(const uint8_t* desc, int n, const std::vector<uint8_t>& cent, int ncent)
{
    std::vector<int> node(n);
    for (int i = 0; i < n; i++) {
        int best = 0, bd = 1 << 30;
        for (int c = 0; c < ncent; c++) {
            int d = 0;
            for (int w = 0; w < 4; w++) {
                unsigned long long a, b;
                memcpy(&a, desc + (size_t)i * 32 + 8 * w, 8);
                memcpy(&b, cent.data() + (size_t)c * 32 + 8 * w, 8);
                d += __builtin_popcountll(a ^ b);
            }
            if (d < bd) {
                bd = d;
                best = c;
            }
        }
        node[i] = best;
    }
    HostFv f;
    f.off.push_back(0);
    for (int c = 0; c < ncent; c++) {
        size_t before = f.ind.size();
        for (int i = 0; i < n; i++)
            if (node[i] == c) f.ind.push_back(i);
        if (f.ind.size() > before) {
            f.ids.push_back((uint32_t)c);
            f.off.push_back((int32_t)f.ind.size());
        }
    }
    return f;
}
template <class F>
static int timeit(const char* name, int reps, F body, std::string& out)
{
    std::vector<double> lat;
    for (int r = -10; r < reps; r++) {
        const double a = now_s();
        const long rc = (long)body();
        if (rc < 0) {
            fprintf(stderr, "hostbench: %s failed: %ld\n", name, rc);
            return 2;
        }
        if (r >= 0) lat.push_back(now_s() - a);
    }
    const Stat s = stat_of(lat);
    char buf[256];
    snprintf(buf, sizeof buf, "%s\"%s\": {\"ms_p50\": %.4f, \"ms_p99\": %.4f, \"ms_mean\": %.4f}", out.empty() ? "" : ", ", name,
             1e3 * s.p50, 1e3 * s.p99, 1e3 * s.mean);
    out += buf;
    return 0;
}
static int run_matcher(const std::vector<uint8_t>& frames, int rows, int cols, int B, int nF, int dev)
{
    const size_t imgBytes = (size_t)rows * cols;
    orbfe_ctx *exA = nullptr, *exB = nullptr;
    CHECK(orbfe_create(&exA, nF, 1.2f, 8, 20, 7, dev));
    CHECK(orbfe_create(&exB, nF, 1.2f, 8, 20, 7, dev));
    const int cap = orbfe_max_keypoints(exA, rows, cols);
    CHECK(cap);
    std::vector<uint8_t> imgB(imgBytes);
    for (int y = 0; y < rows; y++) {
        memcpy(imgB.data() + (size_t)y * cols, frames.data() + (size_t)y * cols + 7, cols - 7);
        memcpy(imgB.data() + (size_t)y * cols + cols - 7, frames.data() + (size_t)y * cols, 7);
    }
    std::vector<orbfe_kp> kA(cap), kB(cap);
    std::vector<uint8_t> dA((size_t)cap * 32), dB((size_t)cap * 32);
    int nA = 0, nB = 0;
    CHECK(orbfe_extract(exA, frames.data(), rows, cols, cols, 0, 0, kA.data(), dA.data(), cap, &nA) + 1);
    CHECK(orbfe_extract(exB, imgB.data(), rows, cols, cols, 0, 0, kB.data(), dB.data(), cap, &nB) + 1);
    const orbfe_kp *dkA, *dkB;
    const uint8_t *ddA, *ddB;
    CHECK(orbfe_get_device_outputs(exA, &dkA, &ddA, nullptr, nullptr, nullptr));
    CHECK(orbfe_get_device_outputs(exB, &dkB, &ddB, nullptr, nullptr, nullptr));
    CHECK(orbfe_sync(exA));
    CHECK(orbfe_sync(exB));
    // FeatureVectors over one "vocabulary" of 100 centroids.  The centroids are descriptors of frame A picked at a fixed
    // stride (a vocabulary is trained on ORB descriptors; uniformly random 256-bit strings are not descriptor-like and send
    // most features to a handful of nodes, which turns the node-sequential kernel into one long row loop)
    std::vector<uint8_t> cent(100 * 32);
    unsigned long long lcg = 88172645463325252ull;
    for (int c = 0; c < 100; c++) memcpy(cent.data() + (size_t)c * 32, dA.data() + (size_t)((c * 977) % nA) * 32, 32);
    const HostFv fA = make_fv(dA.data(), nA, cent, 100), fB = make_fv(dB.data(), nB, cent, 100);
    std::vector<uint8_t> maskA(nA), maskB(nB), hasA(nA), hasB(nB);
    std::vector<float> angA(nA), angB(nB), xyA((size_t)2 * nA), xyB((size_t)2 * nB), uA(nA, -1.f), uB(nB, -1.f);
    std::vector<int32_t> octA(nA), octB(nB);
    for (int i = 0; i < nA; i++) {
        maskA[i] = (i % 10) < 7;
        hasA[i] = (i % 10) < 4;
        angA[i] = kA[i].angle;
        xyA[2 * i] = kA[i].x;
        xyA[2 * i + 1] = kA[i].y;
        octA[i] = kA[i].octave;
        if (i % 3 == 0) uA[i] = kA[i].x - 20.f;
    }
    for (int i = 0; i < nB; i++) {
        maskB[i] = (i % 10) < 7;
        hasB[i] = (i % 10) < 4;
        angB[i] = kB[i].angle;
        xyB[2 * i] = kB[i].x;
        xyB[2 * i + 1] = kB[i].y;
        octB[i] = kB[i].octave;
        if (i % 3 == 0) uB[i] = kB[i].x - 20.f;
    }
    float sf[8], sig[8];
    orbfe_get_scale_tables(exA, sf, nullptr, sig, nullptr);
    int maxNodeA = 0, maxNodeB = 0;
    for (size_t i = 0; i + 1 < fA.off.size(); i++) maxNodeA = std::max(maxNodeA, fA.off[i + 1] - fA.off[i]);
    for (size_t i = 0; i + 1 < fB.off.size(); i++) maxNodeB = std::max(maxNodeB, fB.off[i + 1] - fB.off[i]);
    std::string out;
    std::vector<int32_t> match((size_t)std::max(nA, nB));
    // ---- SearchByBoW(KeyFrame*, Frame&): keyframe = A, frame = B
    orbfe_bow_args bow{};
    bow.desc1 = dA.data(); bow.n1 = nA; bow.mask1 = maskA.data(); bow.angle1 = angA.data(); bow.fv1 = fA.view(); bow.limit1 = -1;
    bow.desc2 = dB.data(); bow.n2 = nB; bow.mask2 = nullptr; bow.angle2 = angB.data(); bow.fv2 = fB.view(); bow.limit2 = -1;
    bow.Nleft = -1; bow.nnratio = 0.7f; bow.check_orientation = 1; bow.variant = 0;
    int nmHost = 0, nmDev = 0, nmKf = 0;
    if (timeit("search_bow_host_arrays", 300, [&] { return nmHost = orbfe_search_bow(dev, &bow, match.data()); }, out)) return 2;
    orbfe_bow_args bowDev = bow;
    bowDev.desc1 = ddA;
    bowDev.desc2 = ddB;
    if (timeit("search_bow_device_descriptors", 300, [&] { return nmDev = orbfe_search_bow(dev, &bowDev, match.data()); }, out)) return 2;
    orbfe_keyframe_args ka{};
    ka.desc = ddA; ka.n = nA; ka.mask = maskA.data(); ka.angle = angA.data(); ka.kp_xy = xyA.data(); ka.octave = octA.data();
    ka.uRight = uA.data(); ka.fv = fA.view();
    orbfe_keyframe* kfA = nullptr;
    if (timeit("keyframe_create_destroy", 100, [&] { orbfe_keyframe* k = nullptr; const int r = orbfe_keyframe_create(&k, dev, &ka); orbfe_keyframe_destroy(k); return r; }, out)) return 2;
    CHECK(orbfe_keyframe_create(&kfA, dev, &ka));
    orbfe_keyframe* kf1[1] = {kfA};
    int32_t* mp[1] = {match.data()};
    if (timeit("search_bow_keyframe_handle", 300, [&] { int nm = 0; const int r = orbfe_search_bow_keyframes(dev, 1, kf1, nullptr, &bowDev, mp, &nm); nmKf = nm; return r; }, out)) return 2;
    if (nmHost != nmDev || nmHost != nmKf) {
        fprintf(stderr, "hostbench: SearchByBoW forms disagree: %d %d %d\n", nmHost, nmDev, nmKf);
        return 2;
    }
    // (a library built with -DORBFE_BOW_TIMING exports the phase times of K-BOW's slowest wavefront: tuning only)
    if (auto bt = (int (*)(unsigned long long*, int))dlsym(RTLD_DEFAULT, "orbfe_debug_bow_times")) {
        unsigned long long t[16];
        bt(t, 1);
        for (int i = 0; i < 20; i++) {
            int nm = 0;
            CHECK(orbfe_search_bow_keyframes(dev, 1, kf1, nullptr, &bowDev, mp, &nm));
        }
        for (int i = 0; i < 3; i++) { // one launch at a time: when its wavefronts started and ended (100-MHz counter common to the chip)
            unsigned long long u[16];
            bt(u, 1);
            int nm = 0;
            const auto h0 = std::chrono::steady_clock::now();
            CHECK(orbfe_search_bow_keyframes(dev, 1, kf1, nullptr, &bowDev, mp, &nm));
            const double callUs = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
            (void)hipDeviceSynchronize();
            bt(u, 0);
            fprintf(stderr, "K-BOW launch %d: call %.1f us; first wavefront start -> last start %.2f us, -> last end %.2f us\n", i, callUs,
                    (u[13] - u[11]) * 0.01, (u[12] - u[11]) * 0.01);
        }
        if (bt(t, 0) == 0)
            fprintf(stderr, "K-BOW, first wavefront of a node, max over the nodes (us): records %.2f  prefetch %.2f  scan + merge %.2f  rounds %.2f  stores issued %.2f  stores acknowledged %.2f  counter %.2f  fence + flag %.2f; nodes %llu, rounds %llu\n",
                    t[0] * 0.01, t[1] * 0.01, t[2] * 0.01, t[3] * 0.01, t[7] * 0.01, t[9] * 0.01, t[10] * 0.01, t[4] * 0.01, t[5], t[6]);
    }
    float bowKernelMs = -1.f;
    {
        orbfe_matcher_time_kernels(1); // (events + a synchronisation per call: only for this one measurement)
        int nm = 0;
        for (int i = 0; i < 5; i++) CHECK(orbfe_search_bow_keyframes(dev, 1, kf1, nullptr, &bowDev, mp, &nm));
        bowKernelMs = orbfe_matcher_last_kernel_ms();
        orbfe_matcher_time_kernels(0);
    }
    // ---- 64 candidates per call (relocalisation, src/Tracking.cc:3784)
    const int NB = 64;
    std::vector<orbfe_bow_args> many(NB, bow), manyDev(NB, bowDev);
    std::vector<std::vector<int32_t>> outs(NB, std::vector<int32_t>((size_t)nB));
    std::vector<int32_t*> outp(NB);
    std::vector<int> nms(NB);
    std::vector<orbfe_keyframe*> kfs(NB, kfA);
    for (int i = 0; i < NB; i++) outp[i] = outs[i].data();
    if (timeit("search_bow_batch64_host_arrays", 60, [&] { return orbfe_search_bow_batch(dev, NB, many.data(), outp.data(), nms.data()); }, out)) return 2;
    if (timeit("search_bow_batch64_keyframe_handles", 60, [&] { return orbfe_search_bow_keyframes(dev, NB, kfs.data(), nullptr, manyDev.data(), outp.data(), nms.data()); }, out)) return 2;
    // ---- The relocalisation chain (src/Tracking.cc:3760-3790): the current frame B is extracted, ComputeBoW, then
    // SearchByBoW against 64 candidate keyframes in handles.  "resident": the frame's descriptors stay where the extractor left
    // them, orbfe_compute_bow leaves the FeatureVector on the device and the search reads it there (orbfe_bow_fv) -- no host round
    // trip between the three stages.  "host fold": round 5's form -- per-feature word / node / weight downloaded
    // (orbfe_vocab_transform), the maps folded on the host, the vector uploaded with the search.
    {
        // a k = 10, L = 3 vocabulary around frame A's descriptors (a trained tree is descriptor-like): 1000 words, 100 nodes one
        // level above the leaves (levelsup = 1, what levelsup = 4 is to ORBvoc's six levels)
        const int K = 10, L3 = 3;
        std::vector<uint8_t> vd(32, 0);
        std::vector<int32_t> voff, vids, vword;
        std::vector<double> vw;
        std::vector<int> parentOf(1, -1), depth(1, 0);
        unsigned long long g = 0x9E3779B97F4A7C15ull;
        auto rnd = [&] { g = g * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(g >> 33); };
        std::vector<std::vector<int>> kids(1);
        for (int lvl = 1; lvl <= L3; lvl++) {
            const int nPrev = (int)parentOf.size();
            for (int pnode = 0; pnode < nPrev; pnode++) {
                if (depth[pnode] != lvl - 1) continue;
                for (int c = 0; c < K; c++) {
                    uint8_t d[32];
                    if (lvl == 1) memcpy(d, dA.data() + (size_t)((c * 977 + 13) % nA) * 32, 32);
                    else {
                        memcpy(d, vd.data() + (size_t)pnode * 32, 32);
                        for (int f = 0; f < 22; f++) {
                            const unsigned b = rnd() % 256;
                            d[b >> 3] ^= (uint8_t)(1u << (b & 7));
                        }
                    }
                    const int id = (int)parentOf.size();
                    parentOf.push_back(pnode);
                    depth.push_back(lvl);
                    kids.push_back({});
                    kids[pnode].push_back(id);
                    vd.insert(vd.end(), d, d + 32);
                }
            }
        }
        const int nn = (int)parentOf.size();
        int nwords = 0;
        voff.assign(nn + 1, 0);
        for (int i = 0; i < nn; i++) {
            voff[i + 1] = voff[i] + (int)kids[i].size();
            for (int c : kids[i]) vids.push_back(c);
            vword.push_back(kids[i].empty() ? nwords++ : -1);
            vw.push_back(kids[i].empty() ? 0.5 + (rnd() % 1000) * 1e-3 : 0.0);
        }
        orbfe_vocab tree{nn, vd.data(), voff.data(), vids.data(), vword.data(), vw.data(), L3};
        orbfe_vocab_dev* voc = nullptr;
        CHECK(orbfe_vocab_upload(&voc, dev, &tree));
        orbfe_bow *bowA = nullptr, *bowB = nullptr;
        CHECK(orbfe_bow_create(&bowA, voc, cap));
        CHECK(orbfe_bow_create(&bowB, voc, cap));
        // the candidate keyframe (A): KeyFrame::ComputeBoW once, the handle keeps the vector
        CHECK(orbfe_compute_bow(bowA, ddA, nA, 1));
        orbfe_keyframe_args kv = ka;
        kv.mask = maskA.data();
        CHECK(orbfe_bow_fv(bowA, &kv.fv));
        orbfe_keyframe* kfV = nullptr;
        CHECK(orbfe_keyframe_create(&kfV, dev, &kv));
        std::vector<orbfe_keyframe*> kfsV(NB, kfV);
        orbfe_bow_args rb = bowDev; // set 2 = the frame B: device descriptors, angles from the host keypoints, the vector below
        std::vector<orbfe_bow_args> relRes(NB, rb), relHost(NB, rb);
        // host fold of round 5 (what binding.bow_from_transform does, in C++): FeatureVector only -- the BowVector is not needed by the search
        std::vector<int32_t> tw(cap), tn(cap);
        std::vector<double> twt(cap);
        HostFv hf;
        auto host_fold = [&]() -> int {
            const int r = orbfe_vocab_transform(voc, ddB, nB, 1, tw.data(), tn.data(), twt.data());
            if (r < 0) return r;
            std::map<uint32_t, std::vector<int32_t>> m;
            for (int i = 0; i < nB; i++)
                if (twt[i] > 0) m[(uint32_t)tn[i]].push_back(i);
            hf.ids.clear(); hf.off.assign(1, 0); hf.ind.clear();
            for (auto& e : m) {
                hf.ids.push_back(e.first);
                hf.ind.insert(hf.ind.end(), e.second.begin(), e.second.end());
                hf.off.push_back((int32_t)hf.ind.size());
            }
            return 0;
        };
        int nmRes = -1, nmFold = -1;
        auto chain_resident = [&](bool extract) -> int {
            if (extract) {
                int r = orbfe_extract(exB, imgB.data(), rows, cols, cols, 0, 0, kB.data(), dB.data(), cap, &nB);
                if (r < -1) return r;
                const uint8_t* dd = nullptr;
                if ((r = orbfe_get_device_outputs(exB, nullptr, &dd, nullptr, nullptr, nullptr)) < 0) return r;
                for (auto& q : relRes) q.desc2 = dd;
            }
            int r = orbfe_compute_bow(bowB, relRes[0].desc2, nB, 1); // asynchronous: three kernels queued
            if (r < 0) return r;
            orbfe_fv fv;
            if ((r = orbfe_bow_fv(bowB, &fv)) < 0) return r;
            for (auto& q : relRes) q.fv2 = fv;
            r = orbfe_search_bow_keyframes(dev, NB, kfsV.data(), nullptr, relRes.data(), outp.data(), nms.data());
            nmRes = nms[NB - 1];
            return r;
        };
        auto chain_host = [&](bool extract) -> int {
            if (extract) {
                int r = orbfe_extract(exB, imgB.data(), rows, cols, cols, 0, 0, kB.data(), dB.data(), cap, &nB);
                if (r < -1) return r;
            }
            int r = host_fold();
            if (r < 0) return r;
            for (auto& q : relHost) q.fv2 = hf.view();
            r = orbfe_search_bow_keyframes(dev, NB, kfsV.data(), nullptr, relHost.data(), outp.data(), nms.data());
            nmFold = nms[NB - 1];
            return r;
        };
        // (the relocalisation needs the FeatureVector on the device and nothing of the BowVector there: its normalisation -- one
        // lane's chain of dependent double additions -- is left to the host view, orbfe_bow_set_lazy_norm; the entry
        // "compute_bow_eager_then_host_copy" below times the fully device-resident form)
        CHECK(orbfe_bow_set_lazy_norm(bowB, 1));
        if (timeit("reloc_bow_search64_resident", 100, [&] { return chain_resident(false); }, out)) return 2;
        if (timeit("reloc_bow_search64_host_fold", 100, [&] { return chain_host(false); }, out)) return 2;
        if (timeit("reloc_extract_bow_search64_resident", 100, [&] { return chain_resident(true); }, out)) return 2;
        if (timeit("reloc_extract_bow_search64_host_fold", 100, [&] { return chain_host(true); }, out)) return 2;
        if (auto bv = (int (*)(unsigned long long*, int))dlsym(RTLD_DEFAULT, "orbfe_debug_bowvec_times")) { // (-DORBFE_BOWVEC_TIMING library)
            unsigned long long u[16];
            for (int rep = 0; rep < 3; rep++) {
                bv(u, 1);
                orbfe_bow_view v;
                CHECK(orbfe_compute_bow(bowB, ddB, nB, 1));
                CHECK(orbfe_bow_host(bowB, &v));
                bv(u, 0);
                fprintf(stderr, "k_bow_rank_fold (n %d, kept %d, nodes %d, words %d), us from the first wavefront: ranks done %.2f | fold begins %.2f  counted %.2f  "
                        "lists in LDS %.2f  heads scanned %.2f  tables written %.2f  values %.2f  norm %.2f  end %.2f\n", nB, v.n_kept, v.nn, v.nw,
                        (u[1] - u[0]) * 0.01, (u[2] - u[0]) * 0.01, (u[3] - u[0]) * 0.01, (u[4] - u[0]) * 0.01, (u[5] - u[0]) * 0.01,
                        (u[6] - u[0]) * 0.01, (u[7] - u[0]) * 0.01, (u[8] - u[0]) * 0.01, (u[9] - u[0]) * 0.01);
            }
        }
        if (timeit("compute_bow_lazy_norm_then_host_copy", 200, [&] { int r = orbfe_compute_bow(bowB, ddB, nB, 1); orbfe_bow_view v; return r < 0 ? r : orbfe_bow_host(bowB, &v); }, out)) return 2;
        CHECK(orbfe_bow_set_lazy_norm(bowB, 0));
        if (timeit("compute_bow_eager_then_host_copy", 200, [&] { int r = orbfe_compute_bow(bowB, ddB, nB, 1); orbfe_bow_view v; return r < 0 ? r : orbfe_bow_host(bowB, &v); }, out)) return 2;
        CHECK(orbfe_compute_bow(bowB, ddB, nB, 1));
        if (timeit("search_bow_batch64_handles_resident_vector_alone", 100, [&] { return orbfe_search_bow_keyframes(dev, NB, kfsV.data(), nullptr, relRes.data(), outp.data(), nms.data()); }, out)) return 2;
        if (nmRes != nmFold || nmRes < 0) {
            fprintf(stderr, "hostbench: relocalisation chains disagree: %d %d\n", nmRes, nmFold);
            return 2;
        }
        orbfe_keyframe_destroy(kfV);
        orbfe_bow_destroy(bowA);
        orbfe_bow_destroy(bowB);
        orbfe_vocab_free(voc);
    }
    // ---- SearchForTriangulation_: A against B (pure x-translation geometry)
    orbfe_tri_args tri{};
    tri.desc1 = dA.data(); tri.n1 = nA; tri.hasMP1 = hasA.data(); tri.kp1_xy = xyA.data(); tri.angle1 = angA.data(); tri.octave1 = octA.data();
    tri.uRight1 = uA.data(); tri.fv1 = fA.view();
    tri.desc2 = dB.data(); tri.n2 = nB; tri.hasMP2 = hasB.data(); tri.kp2_xy = xyB.data(); tri.angle2 = angB.data(); tri.octave2 = octB.data();
    tri.uRight2 = uB.data(); tri.fv2 = fB.view();
    const float fx = 458.654f, fy = 457.296f, tx = 0.11f;
    const float F12[9] = {0, 0, 0, 0, 0, -tx / fy, 0, tx / fy, 0}; // K^-T [t]x K^-1 for t = (tx, 0, 0), cy cancels in l . x2
    (void)fx;
    memcpy(tri.F12, F12, sizeof F12);
    tri.ep[0] = 5000.f; tri.ep[1] = 240.f;
    tri.scaleFactors2 = sf; tri.levelSigma2_2 = sig; tri.nlevels2 = 8;
    tri.only_stereo = 0; tri.coarse = 0; tri.check_orientation = 1;
    std::vector<int32_t> pairs((size_t)2 * nA);
    int npHost = 0;
    if (timeit("search_tri_host_arrays", 300, [&] { return npHost = orbfe_search_tri(dev, &tri, pairs.data()); }, out)) return 2;
    orbfe_keyframe_args kb{};
    kb.desc = ddB; kb.n = nB; kb.mask = hasB.data(); kb.angle = angB.data(); kb.kp_xy = xyB.data(); kb.octave = octB.data();
    kb.uRight = uB.data(); kb.fv = fB.view();
    orbfe_keyframe *kfB = nullptr, *kfAt = nullptr;
    CHECK(orbfe_keyframe_create(&kfB, dev, &kb));
    ka.mask = hasA.data(); // (the triangulation search reads "has a MapPoint", the BoW search "has a good MapPoint")
    CHECK(orbfe_keyframe_create(&kfAt, dev, &ka));
    orbfe_tri_pair tp{};
    memcpy(tp.F12, F12, sizeof F12);
    tp.ep[0] = tri.ep[0]; tp.ep[1] = tri.ep[1];
    tp.scaleFactors2 = sf; tp.levelSigma2_2 = sig; tp.nlevels2 = 8; tp.only_stereo = 0; tp.coarse = 0; tp.check_orientation = 1;
    const int NN = 20;
    std::vector<orbfe_keyframe*> neigh(NN, kfB);
    std::vector<orbfe_tri_pair> tps(NN, tp);
    std::vector<std::vector<int32_t>> pout(NN, std::vector<int32_t>((size_t)2 * nA));
    std::vector<int32_t*> pp(NN);
    std::vector<int> nps(NN);
    for (int i = 0; i < NN; i++) pp[i] = pout[i].data();
    if (timeit("search_tri_keyframe_handles_1", 300, [&] { return orbfe_search_tri_batch(kfAt, nullptr, 1, neigh.data(), tps.data(), pp.data(), nps.data()); }, out)) return 2;
    if (nps[0] != npHost) {
        fprintf(stderr, "hostbench: SearchForTriangulation forms disagree: %d %d\n", npHost, nps[0]);
        return 2;
    }
    float triKernelMs = -1.f;
    {
        orbfe_matcher_time_kernels(1);
        for (int i = 0; i < 5; i++) CHECK(orbfe_search_tri_batch(kfAt, nullptr, NN, neigh.data(), tps.data(), pp.data(), nps.data()));
        triKernelMs = orbfe_matcher_last_kernel_ms();
        orbfe_matcher_time_kernels(0);
    }
    if (timeit("search_tri_batch20_keyframe_handles", 100, [&] { return orbfe_search_tri_batch(kfAt, nullptr, NN, neigh.data(), tps.data(), pp.data(), nps.data()); }, out)) return 2;
    if (timeit("search_tri_20_calls_host_arrays", 30, [&] { int r = 0; for (int i = 0; i < NN && r >= 0; i++) r = orbfe_search_tri(dev, &tri, pairs.data()); return r; }, out)) return 2;
    // ---- SearchByProjection (Frame, local map points): 64 searches of nA map points into frame B
    orbfe_proj_args pr{};
    std::vector<float> kxB(nB), kyB(nB), qx(nA), qy(nA), qr(nA, 15.f);
    std::vector<int32_t> qlo(nA), qhi(nA);
    for (int i = 0; i < nB; i++) {
        kxB[i] = kB[i].x;
        kyB[i] = kB[i].y;
    }
    for (int i = 0; i < nA; i++) {
        qx[i] = kA[i].x - 7.f;
        qy[i] = kA[i].y;
        qlo[i] = std::max(0, kA[i].octave - 1);
        qhi[i] = std::min(7, kA[i].octave + 1);
    }
    pr.desc = dB.data(); pr.n = nB; pr.kx = kxB.data(); pr.ky = kyB.data(); pr.octave = octB.data(); pr.angle = angB.data();
    pr.Nleft = -1; pr.minX = 0; pr.minY = 0; pr.gridWInv = 64.f / cols; pr.gridHInv = 48.f / rows;
    pr.nq = nA; pr.qdesc = dA.data(); pr.qx = qx.data(); pr.qy = qy.data(); pr.qr = qr.data(); pr.qmin_level = qlo.data();
    pr.qmax_level = qhi.data(); pr.mode = 0; pr.nnratio = 0.8f; pr.th_high = 100; pr.check_orientation = 0;
    std::vector<int32_t> qm(nA), fm(nB);
    int nProj = 0;
    if (timeit("search_projection_host_arrays", 200, [&] { return nProj = orbfe_search_projection(dev, &pr, qm.data(), fm.data()); }, out)) return 2;
    {   // the frame side resident (orbfe_frame_create: what the adapter keeps per Frame; Tracking searches one Frame several
        // times -- src/Tracking.cc:2817-2827, :2927): only the queries travel
        orbfe_frame* fr = nullptr;
        // (what making the frame resident costs: once per Frame, before its first search)
        if (timeit("frame_create_destroy_host_arrays", 100, [&] { orbfe_frame* f = nullptr; const int r = orbfe_frame_create(&f, dev, &pr); orbfe_frame_destroy(f); return r; }, out)) return 2;
        {
            orbfe_proj_args prd = pr;
            prd.desc = ddB; // the descriptors where the extractor left them
            if (timeit("frame_create_destroy_device_descriptors", 100, [&] { orbfe_frame* f = nullptr; const int r = orbfe_frame_create(&f, dev, &prd); orbfe_frame_destroy(f); return r; }, out)) return 2;
        }
        CHECK(orbfe_frame_create(&fr, dev, &pr));
        std::vector<int32_t> qm2(nA), fm2(nB);
        int n2 = 0;
        if (timeit("search_projection_frame_handle", 300, [&] { return n2 = orbfe_search_projection_frame(fr, &pr, qm2.data(), fm2.data()); }, out)) return 2;
        if (auto pt = (int (*)(unsigned long long*))dlsym(RTLD_DEFAULT, "orbfe_debug_proj_times")) { // (-DORBFE_PROJ_TIMING library)
            unsigned long long t[16];
            (void)hipDeviceSynchronize();
            if (pt(t) == 0)
                fprintf(stderr, "K-PROJ sweeps workgroup (us since its start): init %.2f  cache %.2f  sweeps %.2f (%llu)  final %.2f  mirror %.2f\n"
                                "K-PROJ candidate wavefronts (mean over %llu, us since a wavefront's start): query + cell ranges %.2f  enumerated + scored %.2f  range reserved %.2f  sorted + written %.2f\n",
                        t[1] * 0.01, t[2] * 0.01, t[3] * 0.01, t[8], t[4] * 0.01, t[5] * 0.01, t[14], t[10] * 0.01 / (double)std::max<unsigned long long>(t[14], 1),
                        t[11] * 0.01 / (double)std::max<unsigned long long>(t[14], 1), t[12] * 0.01 / (double)std::max<unsigned long long>(t[14], 1),
                        t[13] * 0.01 / (double)std::max<unsigned long long>(t[14], 1));
        }
        {   // ... and a tracking-like search against the same resident frame: the queries ARE the frame's features seen again --
            // positions moved by a pixel or two, a dozen descriptor bits flipped, the last-frame form (mode 1, window 15 px
            // x the level's scale, src/Tracking.cc:2817) -- instead of another image's keypoints thrown at it (the problem
            // above: every query competes with a dozen others for the same features, eight resolution sweeps)
            std::vector<float> tx(nB), ty(nB), tr(nB);
            std::vector<int32_t> tlo(nB), thi(nB);
            std::vector<uint8_t> td(dB);
            uint32_t rs = 99u;
            auto rnd = [&] { return rs = rs * 1664525u + 1013904223u; };
            for (int i = 0; i < nB; i++) {
                tx[i] = kB[i].x + (float)((int)(rnd() >> 28) - 8) * 0.25f;
                ty[i] = kB[i].y + (float)((int)(rnd() >> 28) - 8) * 0.25f;
                float sc = 1.f;
                for (int l = 0; l < kB[i].octave; l++) sc *= 1.2f;
                tr[i] = 15.f * sc;
                tlo[i] = std::max(0, kB[i].octave - 1);
                thi[i] = std::min(7, kB[i].octave + 1);
                for (int b = 0; b < 12; b++) {
                    const uint32_t bit = rnd() >> 24;
                    td[(size_t)i * 32 + (bit >> 3)] ^= (uint8_t)(1u << (bit & 7u));
                }
            }
            orbfe_proj_args pt = pr;
            pt.nq = nB; pt.qdesc = td.data(); pt.qx = tx.data(); pt.qy = ty.data(); pt.qr = tr.data(); pt.qmin_level = tlo.data();
            pt.qmax_level = thi.data(); pt.mode = 1; pt.th_high = 100; pt.check_orientation = 0;
            std::vector<int32_t> qm3(nB), fm3(nB);
            int n3 = 0;
            if (timeit("search_projection_frame_handle_tracking_like", 300, [&] { return n3 = orbfe_search_projection_frame(fr, &pt, qm3.data(), fm3.data()); }, out)) return 2;
            fprintf(stderr, "hostbench: tracking-like projection search: %d of %d queries matched, %d sweeps\n", n3, nB, orbfe_search_projection_last_sweeps());
        }
        orbfe_frame_destroy(fr);
        if (n2 != nProj || qm2 != qm || fm2 != fm) {
            fprintf(stderr, "hostbench: SearchByProjection forms disagree: %d %d\n", nProj, n2);
            return 2;
        }
    }
    std::vector<orbfe_proj_args> prs(NB, pr);
    std::vector<std::vector<int32_t>> qms(NB, std::vector<int32_t>(nA)), fms(NB, std::vector<int32_t>(nB));
    std::vector<int32_t*> qmp(NB), fmp(NB);
    std::vector<int32_t> nmp(NB);
    for (int i = 0; i < NB; i++) {
        qmp[i] = qms[i].data();
        fmp[i] = fms[i].data();
    }
    if (timeit("search_projection_batch64_host_arrays", 40, [&] { return orbfe_search_projection_batch(dev, prs.data(), NB, qmp.data(), fmp.data(), nmp.data()); }, out)) return 2;
    orbfe_proj_args prDev = pr;
    prDev.desc = ddB;
    std::vector<orbfe_proj_args> prsDev(NB, prDev);
    if (timeit("search_projection_batch64_device_descriptors", 40, [&] { return orbfe_search_projection_batch(dev, prsDev.data(), NB, qmp.data(), fmp.data(), nmp.data()); }, out)) return 2;
    {   // ... and with the frame side in a handle that all 64 searches name (round 5: orbfe_search_projection_frames)
        orbfe_frame* frB = nullptr;
        orbfe_proj_args prf = pr;
        CHECK(orbfe_frame_create(&frB, dev, &prf));
        std::vector<orbfe_frame*> frs(NB, frB);
        if (timeit("search_projection_batch64_frame_handle", 40, [&] { return orbfe_search_projection_frames(frs.data(), prs.data(), NB, qmp.data(), fmp.data(), nmp.data()); }, out)) return 2;
        orbfe_frame_destroy(frB);
        if (nmp[0] != nProj || nmp[NB - 1] != nProj || qms[NB - 1] != qm) {
            fprintf(stderr, "hostbench: projection batch over a frame handle disagrees: %d %d\n", nmp[0], nProj);
            return 2;
        }
    }
    // ---- DBoW2 transform of one frame's descriptors: k = 10, L = 4 synthetic tree (11111 nodes)
    {
        const int k = 10, L = 4;
        int nn = 0;
        for (int l = 0, c = 1; l <= L; l++, c *= k) nn += c;
        std::vector<uint8_t> nd((size_t)nn * 32);
        for (auto& c : nd) {
            lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
            c = (uint8_t)(lcg >> 56);
        }
        std::vector<int32_t> coff(nn + 1), cids, word(nn, -1);
        std::vector<double> weight(nn, 0.0);
        int inner = 0;
        for (int l = 0, c = 1; l < L; l++, c *= k) inner += c;
        int wid = 0;
        for (int i = 0; i < nn; i++) {
            coff[i] = (int32_t)cids.size();
            if (i < inner)
                for (int c = 0; c < k; c++) cids.push_back(1 + k * i + c);
            else {
                word[i] = wid++;
                weight[i] = 1.0 + (i % 7);
            }
        }
        coff[nn] = (int32_t)cids.size();
        orbfe_vocab v{};
        v.nnodes = nn; v.node_desc = nd.data(); v.child_off = coff.data(); v.child_ids = cids.data(); v.node_word = word.data();
        v.node_weight = weight.data(); v.L = L;
        orbfe_vocab_dev* vd = nullptr;
        CHECK(orbfe_vocab_upload(&vd, dev, &v));
        std::vector<int32_t> wi(nB), ni(nB);
        std::vector<double> ww(nB);
        if (timeit("vocab_transform_host_descriptors", 300, [&] { return orbfe_vocab_transform(vd, dB.data(), nB, 4, wi.data(), ni.data(), ww.data()); }, out)) return 2;
        if (timeit("vocab_transform_device_descriptors", 300, [&] { return orbfe_vocab_transform(vd, ddB, nB, 4, wi.data(), ni.data(), ww.data()); }, out)) return 2;
        orbfe_vocab_free(vd);
    }
    printf("{\"config\": \"matcher\", \"frame\": \"%dx%d\", \"nA\": %d, \"nB\": %d, \"bow_matches\": %d, \"tri_pairs\": %d, "
           "\"projection_matches\": %d, \"feature_vector\": {\"nodes\": %d, \"largest_node\": [%d, %d]}, \"kernel_ms\": {\"search_bow\": %.4f, \"search_tri_batch20\": %.4f}, \"calls\": {%s}}\n",
           cols, rows, nA, nB, nmHost, npHost, nProj, (int)fA.ids.size(), maxNodeA, maxNodeB, bowKernelMs, triKernelMs, out.c_str());
    orbfe_keyframe_destroy(kfA);
    orbfe_keyframe_destroy(kfAt);
    orbfe_keyframe_destroy(kfB);
    orbfe_destroy(exA);
    orbfe_destroy(exB);
    (void)B;
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 6) {
        fprintf(stderr, "usage: hostbench frames.raw rows cols nframes nfeatures [device]\n");
        return 1;
    }
    const char* path = argv[1];
    const int rows = atoi(argv[2]), cols = atoi(argv[3]), B = atoi(argv[4]), nF = atoi(argv[5]);
    const int dev = argc > 6 ? atoi(argv[6]) : 0;
    const size_t imgBytes = (size_t)rows * cols;
    std::vector<uint8_t> frames(imgBytes * B);
    {
        FILE* f = fopen(path, "rb");
        if (!f || fread(frames.data(), 1, frames.size(), f) != frames.size()) {
            fprintf(stderr, "hostbench: cannot read %s\n", path);
            return 1;
        }
        fclose(f);
    }
    if (argc > 7 && !strcmp(argv[7], "c5")) return run_c5(frames, rows, cols, B, nF, dev);
    if (argc > 7 && !strcmp(argv[7], "matcher")) return run_matcher(frames, rows, cols, B, nF, dev);
    if (argc > 7 && !strcmp(argv[7], "stream")) return run_stream(frames, rows, cols, B, nF, dev);
    orbfe_ctx* ex = nullptr;
    const double tCreate0 = now_s();
    CHECK(orbfe_create(&ex, nF, 1.2f, 8, 20, 7, dev));
    const int cap = orbfe_max_keypoints(ex, rows, cols);
    CHECK(cap);
    const size_t kB = (size_t)cap * 28, dB = (size_t)cap * 32;

    // caller-side arrays: pageable and pinned twins
    std::vector<uint8_t> pgK(kB * B), pgD(dB * B);
    uint8_t* pinImg[2];
    uint8_t* pinK[2];
    uint8_t* pinD[2];
    for (int k = 0; k < 2; k++) {
        pinImg[k] = (uint8_t*)orbfe_host_alloc(imgBytes * B);
        pinK[k] = (uint8_t*)orbfe_host_alloc(kB * B);
        pinD[k] = (uint8_t*)orbfe_host_alloc(dB * B);
        if (!pinImg[k] || !pinK[k] || !pinD[k]) return 2;
        memcpy(pinImg[k], frames.data(), imgBytes * B);
    }
    std::vector<const uint8_t*> pPg(B), pPin0(B), pPin1(B);
    for (int i = 0; i < B; i++) {
        pPg[i] = frames.data() + imgBytes * i;
        pPin0[i] = pinImg[0] + imgBytes * i;
        pPin1[i] = pinImg[1] + imgBytes * i;
    }
    std::vector<int> lap(2 * B), n(B), mono(B), n2(B), mono2(B);
    for (int i = 0; i < B; i++) {
        lap[2 * i] = 0;
        lap[2 * i + 1] = 1000; // mono protocol, src/Frame.cc:306
    }

    // first call of the process: builds and uploads the libm trig table (ORBFE_TRIG_LIBM), allocates, uploads tables
    int n0 = 0;
    const double tFirst0 = now_s();
    CHECK(orbfe_extract(ex, pPg[0], rows, cols, cols, 0, 1000, (orbfe_kp*)pgK.data(), pgD.data(), cap, &n0) + 1);
    const double firstCallMs = 1e3 * (now_s() - tFirst0), createMs = 1e3 * (tFirst0 - tCreate0);
    if (argc > 7 && !strcmp(argv[7], "first")) { // start-up cost only (bench.py runs this with and without the table cache)
        printf("{\"create_ms\": %.2f, \"first_call_ms\": %.2f}\n", createMs, firstCallMs);
        orbfe_destroy(ex);
        return 0;
    }

    // ---- single frame per call
    auto single = [&](bool pinned, Stat* st, double* kpPerS, int ring = 0) -> int {
        std::vector<double> lat;
        long kp = 0;
        const int R = ring > 0 ? std::min(ring, B) : B; // the caller's buffers: all B frames, or a ring of a few (a camera driver)
        for (int w = 0; w < 20; w++) {
            int nn = 0;
            const int i = w % R;
            CHECK(orbfe_extract(ex, pinned ? pPin0[i] : pPg[i], rows, cols, cols, 0, 1000,
                                (orbfe_kp*)(pinned ? pinK[0] : pgK.data()), pinned ? pinD[0] : pgD.data(), cap, &nn) + 1);
        }
        const double t0 = now_s();
        for (int rep = 0; rep < 8; rep++)
            for (int j = 0; j < B; j++) {
                int nn = 0;
                const int i = j % R;
                const double a = now_s();
                CHECK(orbfe_extract(ex, pinned ? pPin0[i] : pPg[i], rows, cols, cols, 0, 1000,
                                    (orbfe_kp*)(pinned ? pinK[0] : pgK.data()), pinned ? pinD[0] : pgD.data(), cap, &nn) + 1);
                lat.push_back(now_s() - a);
                kp += nn;
            }
        *kpPerS = kp / (now_s() - t0);
        *st = stat_of(lat);
        return 0;
    };
    Stat sPg, sPin;
    double kpsPg, kpsPin;
    if (single(false, &sPg, &kpsPg)) return 2;
    if (single(true, &sPin, &kpsPin)) return 2;

    // ---- whole batch per call, blocking
    auto batch = [&](bool pinned, double* ms, double* kpPerS) -> int {
        const int reps = 20;
        long kp = 0;
        for (int w = 0; w < 3; w++)
            CHECK(orbfe_extract_batch(ex, B, pinned ? pPin0.data() : pPg.data(), rows, cols, cols, lap.data(),
                                      (orbfe_kp*)(pinned ? pinK[0] : pgK.data()), pinned ? pinD[0] : pgD.data(), cap,
                                      n.data(), mono.data()));
        const double t0 = now_s();
        for (int r = 0; r < reps; r++) {
            CHECK(orbfe_extract_batch(ex, B, pinned ? pPin0.data() : pPg.data(), rows, cols, cols, lap.data(),
                                      (orbfe_kp*)(pinned ? pinK[0] : pgK.data()), pinned ? pinD[0] : pgD.data(), cap,
                                      n.data(), mono.data()));
            for (int i = 0; i < B; i++) kp += n[i];
        }
        const double dt = now_s() - t0;
        *ms = 1e3 * dt / reps;
        *kpPerS = kp / dt;
        return 0;
    };
    double msPg, msPin, bkPg, bkPin;
    if (batch(false, &msPg, &bkPg)) return 2;
    if (batch(true, &msPin, &bkPin)) return 2;

    // ---- the unmodified caller with long-lived pageable buffers (orbfe_set_auto_register): the library page-locks a buffer
    // the second time it sees it; single frames from a ring of 8 buffers, and the whole batch from its one buffer
    Stat sAuto{0, 0, 0};
    double kpsAuto = 0, msAuto = 0, bkAuto = 0;
    CHECK(orbfe_set_auto_register(ex, 1));
    if (single(false, &sAuto, &kpsAuto, 8)) return 2;
    if (batch(false, &msAuto, &bkAuto)) return 2;
    CHECK(orbfe_set_auto_register(ex, 0));

    // ---- one batch at a time through the submit / wait entry points (copies on their own streams)
    double msSW = 0;
    {
        const int reps = 20;
        for (int w = 0; w < 3; w++) {
            CHECK(orbfe_extract_batch_submit(ex, B, pPin0.data(), rows, cols, cols, lap.data(), (orbfe_kp*)pinK[0], pinD[0], cap, n.data(), mono.data()));
            CHECK(orbfe_extract_batch_wait(ex));
        }
        const double t0 = now_s();
        for (int r = 0; r < reps; r++) {
            CHECK(orbfe_extract_batch_submit(ex, B, pPin0.data(), rows, cols, cols, lap.data(), (orbfe_kp*)pinK[0], pinD[0], cap, n.data(), mono.data()));
            CHECK(orbfe_extract_batch_wait(ex));
        }
        msSW = 1e3 * (now_s() - t0) / reps;
    }

    // ---- two batches in flight (pinned buffers, two sets)
    double msPipe = 0, bkPipe = 0;
    {
        const int reps = 60;
        long kp = 0;
        const uint8_t* const* P[2] = {pPin0.data(), pPin1.data()};
        int* N[2] = {n.data(), n2.data()};
        int* M[2] = {mono.data(), mono2.data()};
        CHECK(orbfe_extract_batch_submit(ex, B, P[0], rows, cols, cols, lap.data(), (orbfe_kp*)pinK[0], pinD[0], cap, N[0], M[0]));
        const double t0 = now_s();
        for (int r = 1; r <= reps; r++) {
            const int k = r & 1;
            CHECK(orbfe_extract_batch_submit(ex, B, P[k], rows, cols, cols, lap.data(), (orbfe_kp*)pinK[k], pinD[k], cap, N[k], M[k]));
            CHECK(orbfe_extract_batch_wait(ex)); // completes batch r-1
            for (int i = 0; i < B; i++) kp += N[k ^ 1][i];
        }
        const double dt = now_s() - t0;
        CHECK(orbfe_extract_batch_wait(ex));
        msPipe = 1e3 * dt / reps;
        bkPipe = kp / dt;
    }

    // ---- PCIe floor: the same bytes with bare copies
    double floorInMs = 0, floorOutMs = 0;
    {
        if (hipSetDevice(dev) != hipSuccess) return 2;
        uint8_t* d = nullptr;
        const size_t inB = imgBytes * B, outB = (kB + dB) * B;
        if (hipMalloc((void**)&d, std::max(inB, outB)) != hipSuccess) return 2;
        hipStream_t s;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 2;
        for (int w = 0; w < 3; w++) (void)hipMemcpyAsync(d, pinImg[0], inB, hipMemcpyHostToDevice, s);
        (void)hipStreamSynchronize(s);
        double t0 = now_s();
        for (int r = 0; r < 20; r++) (void)hipMemcpyAsync(d, pinImg[0], inB, hipMemcpyHostToDevice, s);
        (void)hipStreamSynchronize(s);
        floorInMs = 1e3 * (now_s() - t0) / 20;
        t0 = now_s();
        for (int r = 0; r < 20; r++) {
            (void)hipMemcpyAsync(pinK[1], d, kB * B, hipMemcpyDeviceToHost, s);
            (void)hipMemcpyAsync(pinD[1], d, dB * B, hipMemcpyDeviceToHost, s);
        }
        (void)hipStreamSynchronize(s);
        floorOutMs = 1e3 * (now_s() - t0) / 20;
        (void)hipStreamDestroy(s);
        (void)hipFree(d);
    }

    // ---- stereo pair: two extractors (nFeatures as given), two threads started per frame (src/Frame.cc:119-122),
    // then ComputeStereoMatches on the resident results.  The right image is the left one shifted by 12 px.
    Stat sStereo{0, 0, 0}, sStereoExtract{0, 0, 0}, sStereo1{0, 0, 0}, sStereoExtract1{0, 0, 0}, sStereo1P{0, 0, 0},
        sStereoExtract1P{0, 0, 0}, sStereoF{0, 0, 0}, sStereoFP{0, 0, 0};
    double stereoMatches = 0, stereoMatches1 = 0;
    long stereoKp = 0;
    const int nPairs = 200, nFstereo = 1200;
    {
        orbfe_ctx *exL = nullptr, *exR = nullptr;
        CHECK(orbfe_create(&exL, nFstereo, 1.2f, 8, 20, 7, dev)); // Examples/Stereo/EuRoC.yaml: ORBextractor.nFeatures: 1200
        CHECK(orbfe_create(&exR, nFstereo, 1.2f, 8, 20, 7, dev));
        const int capS = orbfe_max_keypoints(exL, rows, cols);
        CHECK(capS);
        std::vector<uint8_t> right(imgBytes * B);
        for (int i = 0; i < B; i++)
            for (int y = 0; y < rows; y++) {
                const uint8_t* s = frames.data() + imgBytes * i + (size_t)y * cols;
                uint8_t* d = right.data() + imgBytes * i + (size_t)y * cols;
                memcpy(d, s + 12, cols - 12);
                memcpy(d + cols - 12, s, 12);
            }
        std::vector<uint8_t> kL((size_t)capS * 28), dL((size_t)capS * 32), kR((size_t)capS * 28), dR((size_t)capS * 32);
        std::vector<float> uR(capS), depth(capS);
        const float bf = 47.90639384423901f, fx = 435.2046959714599f; // Examples/Stereo/EuRoC.yaml
        std::vector<double> lat, latE;
        for (int r = -10; r < nPairs; r++) {
            const int i = (r + 10) % B;
            int nL = 0, nR = 0, rcL = 0, rcR = 0;
            const double a = now_s();
            std::thread tl([&] { rcL = orbfe_extract(exL, frames.data() + imgBytes * i, rows, cols, cols, 0, 0, (orbfe_kp*)kL.data(), dL.data(), capS, &nL); });
            std::thread tr([&] { rcR = orbfe_extract(exR, right.data() + imgBytes * i, rows, cols, cols, 0, 0, (orbfe_kp*)kR.data(), dR.data(), capS, &nR); });
            tl.join();
            tr.join();
            const double b = now_s();
            if (rcL < 0 || rcR < 0) return 2;
            const int m = orbfe_compute_stereo_matches_resident(exL, 0, exR, 0, bf / fx, bf, uR.data(), depth.data(), nL);
            CHECK(m);
            if (r >= 0) {
                lat.push_back(now_s() - a);
                latE.push_back(b - a);
                stereoMatches += m;
                stereoKp += nL + nR;
            }
        }
        sStereo = stat_of(lat);
        sStereoExtract = stat_of(latE);
        // the same pair as ONE call: both images in one orbfe_extract_batch on one context (what a maintainer can put
        // in place of the two threads of Frame.cc:119-122), then ComputeStereoMatches between image 0 and image 1
        {
            std::vector<double> lat1, latE1;
            std::vector<uint8_t> kLR((size_t)2 * capS * 28), dLR((size_t)2 * capS * 32);
            int n2[2] = {0, 0}, mono2[2] = {0, 0};
            const int lap2[4] = {0, 0, 0, 0};
            for (int r = -10; r < nPairs; r++) {
                const int i = (r + 10) % B;
                const uint8_t* two[2] = {frames.data() + imgBytes * i, right.data() + imgBytes * i};
                const double a = now_s();
                CHECK(orbfe_extract_batch(exL, 2, two, rows, cols, cols, lap2, (orbfe_kp*)kLR.data(), dLR.data(), capS, n2, mono2));
                const double b = now_s();
                const int m = orbfe_compute_stereo_matches_resident(exL, 0, exL, 1, bf / fx, bf, uR.data(), depth.data(), n2[0]);
                CHECK(m);
                if (r >= 0) {
                    lat1.push_back(now_s() - a);
                    latE1.push_back(b - a);
                    stereoMatches1 += m;
                }
            }
            sStereo1 = stat_of(lat1);
            sStereoExtract1 = stat_of(latE1);
            // ... and extraction + matching behind ONE host wait (orbfe_extract_stereo_pair), pageable images
            {
                std::vector<double> lat3;
                std::vector<float> uR2(capS), dep2(capS);
                for (int r = -10; r < nPairs; r++) {
                    const int i = (r + 10) % B;
                    const double a = now_s();
                    const int m = orbfe_extract_stereo_pair(exL, frames.data() + imgBytes * i, right.data() + imgBytes * i, rows, cols, cols,
                                                            lap2, (orbfe_kp*)kLR.data(), dLR.data(), capS, n2, mono2, bf / fx, bf,
                                                            uR2.data(), dep2.data());
                    CHECK(m);
                    if (r >= 0) lat3.push_back(now_s() - a);
                }
                sStereoF = stat_of(lat3);
            }
            // ... and with the caller's image buffers page-locked (orbfe_host_register once: a camera driver's ring)
            if (orbfe_host_register(frames.data(), imgBytes * B) == 0 && orbfe_host_register(right.data(), imgBytes * B) == 0) {
                std::vector<double> lat2, latE2;
                for (int r = -10; r < nPairs; r++) {
                    const int i = (r + 10) % B;
                    const uint8_t* two[2] = {frames.data() + imgBytes * i, right.data() + imgBytes * i};
                    const double a = now_s();
                    CHECK(orbfe_extract_batch(exL, 2, two, rows, cols, cols, lap2, (orbfe_kp*)kLR.data(), dLR.data(), capS, n2, mono2));
                    const double b = now_s();
                    const int m = orbfe_compute_stereo_matches_resident(exL, 0, exL, 1, bf / fx, bf, uR.data(), depth.data(), n2[0]);
                    CHECK(m);
                    if (r >= 0) {
                        lat2.push_back(now_s() - a);
                        latE2.push_back(b - a);
                    }
                }
                sStereo1P = stat_of(lat2);
                sStereoExtract1P = stat_of(latE2);
                std::vector<double> lat4;
                std::vector<float> uR2(capS), dep2(capS);
                for (int r = -10; r < nPairs; r++) {
                    const int i = (r + 10) % B;
                    const double a = now_s();
                    const int m = orbfe_extract_stereo_pair(exL, frames.data() + imgBytes * i, right.data() + imgBytes * i, rows, cols, cols,
                                                            lap2, (orbfe_kp*)kLR.data(), dLR.data(), capS, n2, mono2, bf / fx, bf,
                                                            uR2.data(), dep2.data());
                    CHECK(m);
                    if (r >= 0) lat4.push_back(now_s() - a);
                }
                sStereoFP = stat_of(lat4);
                if (auto stt = (int (*)(unsigned long long*))dlsym(RTLD_DEFAULT, "orbfe_debug_stereo_times")) { // (-DORBFE_STEREO_TIMING library)
                    unsigned long long t[8];
                    const int sr = stt(t);
                    if (sr != 0 || !t[7]) fprintf(stderr, "K-STEREO timing: rc %d, wavefronts %llu\n", sr, t[7]);
                    if (sr == 0 && t[7])
                        fprintf(stderr, "K-STEREO, mean over %llu wavefronts (us since a wavefront's start): table staged %.2f  scanned %.2f  scored %.2f  "
                                        "SAD summed %.2f  stored %.2f\n",
                                t[7], t[1] * 0.01 / t[7], t[2] * 0.01 / t[7], t[3] * 0.01 / t[7], t[4] * 0.01 / t[7], t[5] * 0.01 / t[7]);
                }
                (void)orbfe_host_unregister(frames.data());
                (void)orbfe_host_unregister(right.data());
            }
        }
        orbfe_destroy(exL);
        orbfe_destroy(exR);
    }

    long kpBatch = 0;
    for (int i = 0; i < B; i++) kpBatch += n[i];
    const double inMB = imgBytes * B / 1e6, outMB = (double)kpBatch * 60 / 1e6;
    printf("{\"frame\": \"%dx%d\", \"nfeatures\": %d, \"batch\": %d, \"keypoints_per_batch\": %ld, "
           "\"create_ms\": %.2f, \"first_call_ms\": %.1f, "
           "\"single_pageable\": {\"ms_mean\": %.4f, \"ms_p50\": %.4f, \"ms_p99\": %.4f, \"keypoints_per_s\": %.0f}, "
           "\"single_pinned\": {\"ms_mean\": %.4f, \"ms_p50\": %.4f, \"ms_p99\": %.4f, \"keypoints_per_s\": %.0f}, "
           "\"batch_pageable\": {\"ms_per_batch\": %.4f, \"keypoints_per_s\": %.0f}, "
           "\"batch_pinned\": {\"ms_per_batch\": %.4f, \"keypoints_per_s\": %.0f}, "
           "\"single_pageable_autoreg\": {\"ms_mean\": %.4f, \"ms_p50\": %.4f, \"ms_p99\": %.4f, \"keypoints_per_s\": %.0f, "
           "\"note\": \"pageable ring of 8 caller buffers, orbfe_set_auto_register(ctx, 1)\"}, "
           "\"batch_pageable_autoreg\": {\"ms_per_batch\": %.4f, \"keypoints_per_s\": %.0f}, "
           "\"batch_submit_wait\": {\"in_flight\": 1, \"ms_per_batch\": %.4f}, "
           "\"batch_pipelined\": {\"in_flight\": 2, \"ms_per_batch\": %.4f, \"keypoints_per_s\": %.0f}, "
           "\"pcie_floor\": {\"h2d_ms\": %.4f, \"d2h_ms\": %.4f, \"h2d_GBps\": %.1f, \"d2h_GBps\": %.1f, "
           "\"in_MB\": %.2f, \"out_MB\": %.2f, \"note\": \"bare hipMemcpyAsync of the batch's images / full output slabs, pinned\"}, "
           "\"stereo_pair\": {\"nfeatures\": 1200, \"protocol\": \"2 contexts, 2 threads started per frame (src/Frame.cc:119-122), pageable images, "
           "then orbfe_compute_stereo_matches_resident\", \"pairs\": %d, \"ms_per_pair_mean\": %.4f, \"ms_per_pair_p50\": %.4f, "
           "\"ms_per_pair_p99\": %.4f, \"extract_ms_p50\": %.4f, \"keypoints_per_s\": %.0f, \"matches_per_pair\": %.1f}, "
           "\"stereo_pair_one_call\": {\"nfeatures\": 1200, \"protocol\": \"1 context, both images in one orbfe_extract_batch (pageable), "
           "then orbfe_compute_stereo_matches_resident between image 0 and image 1\", \"ms_per_pair_mean\": %.4f, "
           "\"ms_per_pair_p50\": %.4f, \"ms_per_pair_p99\": %.4f, \"extract_ms_p50\": %.4f, \"matches_per_pair\": %.1f}, "
           "\"stereo_pair_one_call_pinned\": {\"protocol\": \"the same with the caller's image buffers page-locked (orbfe_host_register)\", "
           "\"ms_per_pair_mean\": %.4f, \"ms_per_pair_p50\": %.4f, \"ms_per_pair_p99\": %.4f, \"extract_ms_p50\": %.4f}, "
           "\"stereo_pair_fused\": {\"protocol\": \"orbfe_extract_stereo_pair: both images and ComputeStereoMatches behind one host wait\", "
           "\"pageable\": {\"ms_per_pair_p50\": %.4f, \"ms_per_pair_p99\": %.4f}, \"pinned\": {\"ms_per_pair_p50\": %.4f, "
           "\"ms_per_pair_p99\": %.4f}}}\n",
           cols, rows, nF, B, kpBatch, createMs, firstCallMs, 1e3 * sPg.mean, 1e3 * sPg.p50, 1e3 * sPg.p99, kpsPg,
           1e3 * sPin.mean, 1e3 * sPin.p50, 1e3 * sPin.p99, kpsPin, msPg, bkPg, msPin, bkPin, 1e3 * sAuto.mean, 1e3 * sAuto.p50,
           1e3 * sAuto.p99, kpsAuto, msAuto, bkAuto, msSW, msPipe, bkPipe, floorInMs,
           floorOutMs, inMB / floorInMs, (kB + dB) * B / 1e6 / floorOutMs, inMB, outMB, nPairs, 1e3 * sStereo.mean,
           1e3 * sStereo.p50, 1e3 * sStereo.p99, 1e3 * sStereoExtract.p50, stereoKp / (sStereo.mean * nPairs),
           stereoMatches / nPairs, 1e3 * sStereo1.mean, 1e3 * sStereo1.p50, 1e3 * sStereo1.p99, 1e3 * sStereoExtract1.p50,
           stereoMatches1 / nPairs, 1e3 * sStereo1P.mean, 1e3 * sStereo1P.p50, 1e3 * sStereo1P.p99, 1e3 * sStereoExtract1P.p50,
           1e3 * sStereoF.p50, 1e3 * sStereoF.p99, 1e3 * sStereoFP.p50, 1e3 * sStereoFP.p99);
    for (int k = 0; k < 2; k++) {
        orbfe_host_free(pinImg[k]);
        orbfe_host_free(pinK[k]);
        orbfe_host_free(pinD[k]);
    }
    orbfe_destroy(ex);
    return 0;
}
