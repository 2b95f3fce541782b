// hostbench.cpp -- the PCIe-INCLUSIVE rates of the drop-in boundary, measured from a plain C++ caller through the
// C ABI (include/orbfe.h): what SURVEY.md section 8(d) / BASELINE.md section 3 define as the metric (wall time of
// orbfe_extract* including H2D of the images and D2H of keypoints + descriptors).  bench.py runs this program as
// a child process and embeds its one JSON line as the `pcie_inclusive` object; it is never bench.py's `value`.
//
//   hostbench <frames.raw> rows cols nframes nfeatures [device]
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

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
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
