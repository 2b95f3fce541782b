// orbfe_matcher_api_knn2.hip -- entry points: DescriptorDistance, knn-2 (host, device, frames).
// Part of the matcher's translation unit: included by orbfe_matcher.hip, in this order, behind the common device helpers
// (the text is the one translation unit it always was, cut at its family borders -- VERDICT r05 #6).

int orbfe_hamming_pairs(int device, const uint8_t* A, int nA, const uint8_t* B, int nB, uint16_t* D)
{
    if (nA < 0 || nB < 0 || (nA && !A) || (nB && !B) || (nA && nB && !D)) return ORBFE_ERR_ARGS;
    if (nA == 0 || nB == 0) return 0;
    int r;
    if ((r = select_device(device)) < 0) return r;
    Scratch s(device);
    uint8_t *dA, *dB;
    uint16_t* dD;
    if ((r = s.up_desc(&dA, A, (size_t)nA * 32)) < 0) return r;
    if ((r = s.up_desc(&dB, B, (size_t)nB * 32)) < 0) return r;
    if ((r = s.up<uint16_t>(&dD, nullptr, (size_t)nA * nB)) < 0) return r;
    {
        KernelTimer timer(s);
    hipLaunchKernelGGL(k_hamming_pairs, dim3((unsigned)((nB + 63) / 64), (unsigned)((nA + 63) / 64)), dim3(256), 0, g_ms, dA,
                       nA, dB, nB, dD);
    }
    HIP_TRY(hipGetLastError());
    INT_TRY(s.down(D, dD, (size_t)nA * nB * sizeof(uint16_t)));
    INT_TRY(s.fetch());
    return 0;
}

int orbfe_bfknn2(int device, const uint8_t* Q, int nQ, const uint8_t* T, int nT, int32_t* idx, int32_t* dist)
{
    if (nQ < 0 || nT < 0 || (nQ && (!Q || !idx || !dist)) || (nT && !T) || nT >= (1 << 20)) return ORBFE_ERR_ARGS;
    if (nQ == 0) return 0;
    int r;
    if ((r = select_device(device)) < 0) return r;
    Scratch s(device);
    uint8_t *dQ, *dT;
    int32_t *dI, *dD;
    if ((r = s.up_desc(&dQ, Q, (size_t)nQ * 32)) < 0) return r;
    if ((r = s.up_desc(&dT, T, (size_t)nT * 32)) < 0) return r;
    if ((r = s.up<int32_t>(&dI, nullptr, (size_t)nQ * 2)) < 0) return r;
    if ((r = s.up<int32_t>(&dD, nullptr, (size_t)nQ * 2)) < 0) return r;
    {
        KernelTimer timer(s);
    hipLaunchKernelGGL(k_bfknn2, dim3((unsigned)((nQ + 3) / 4)), dim3(256), 0, g_ms, dQ, nQ, dT, nT, dI, dD);
    }
    HIP_TRY(hipGetLastError());
    INT_TRY(s.down(idx, dI, (size_t)nQ * 2 * sizeof(int32_t)));
    INT_TRY(s.down(dist, dD, (size_t)nQ * 2 * sizeof(int32_t)));
    INT_TRY(s.fetch());
    return 0;
}

// ---- device-resident forms: every pointer is device memory, nothing is copied, nothing is waited for ----
static hipStream_t matcher_stream(int device, void* hip_stream)
{
    if (hip_stream) return (hipStream_t)hip_stream;
    Scratch s(device); // makes sure the calling thread's matcher stream exists
    return g_ms;
}

int orbfe_hamming_pairs_device(int device, void* hip_stream, const uint8_t* dA, int nA, const uint8_t* dB, int nB,
                               uint16_t* dD)
{
    if (nA < 0 || nB < 0 || (nA && !dA) || (nB && !dB) || (nA && nB && !dD)) return ORBFE_ERR_ARGS;
    if (nA == 0 || nB == 0) return 0;
    int r;
    if ((r = select_device(device)) < 0) return r;
    hipStream_t st = matcher_stream(device, hip_stream);
    if (int w = orbfe_producer_wait(dA, st); w < 0) return w;
    if (int w = orbfe_producer_wait(dB, st); w < 0) return w;
    hipLaunchKernelGGL(k_hamming_pairs, dim3((unsigned)((nB + 63) / 64), (unsigned)((nA + 63) / 64)), dim3(256), 0, st, dA, nA, dB, nB,
                       dD);
    HIP_TRY(hipGetLastError());
    return 0;
}

int orbfe_bfknn2_device(int device, void* hip_stream, const uint8_t* dQ, int nQ, const uint8_t* dT, int nT, int32_t* d_idx,
                        int32_t* d_dist)
{
    if (nQ < 0 || nT < 0 || (nQ && (!dQ || !d_idx || !d_dist)) || (nT && !dT) || nT >= (1 << 20)) return ORBFE_ERR_ARGS;
    if (nQ == 0) return 0;
    int r;
    if ((r = select_device(device)) < 0) return r;
    hipStream_t st = matcher_stream(device, hip_stream);
    if (int w = orbfe_producer_wait(dQ, st); w < 0) return w;
    if (int w = orbfe_producer_wait(dT, st); w < 0) return w;
    hipLaunchKernelGGL(k_bfknn2, dim3((unsigned)((nQ + 3) / 4)), dim3(256), 0, st, dQ, nQ, dT, nT, d_idx, d_dist);
    HIP_TRY(hipGetLastError());
    return 0;
}

// flags: bit 0 = the caller knows of other kernels in flight on the device (orbfe_mc_match_ring_async with extractions queued);
// bit 1 = rows between a query frame's count and `cap` are written too, as -1 | -1
extern "C" int orbfe_internal_bfknn2_frames(int device, void* hip_stream, const orbfe_knn2_job* d_jobs, int njobs, int cap,
                                            int32_t* d_idx, int32_t* d_dist, int flags);

int orbfe_bfknn2_frames_device(int device, void* hip_stream, const orbfe_knn2_job* d_jobs, int njobs, int cap,
                               int32_t* d_idx, int32_t* d_dist)
{
    return orbfe_internal_bfknn2_frames(device, hip_stream, d_jobs, njobs, cap, d_idx, d_dist, 0);
}

int orbfe_internal_bfknn2_frames(int device, void* hip_stream, const orbfe_knn2_job* d_jobs, int njobs, int cap,
                                 int32_t* d_idx, int32_t* d_dist, int flags)
{
    const int shared = flags & 1, fillTail = (flags >> 1) & 1;
    if (njobs < 0 || cap < 1 || cap >= (1 << 20) || (njobs && (!d_jobs || !d_idx || !d_dist))) return ORBFE_ERR_ARGS;
    if (njobs == 0) return 0;
    int r;
    if ((r = select_device(device)) < 0) return r;
    hipStream_t st = matcher_stream(device, hip_stream);
    // the job records live on the device, so the frames they name cannot be looked up one by one: this stream waits for
    // every extraction whose outputs were handed out (orbfe_get_device_outputs) -- a few events, fired long ago as a rule
    if ((r = orbfe_producer_wait_all(st)) < 0) return r;
    // The matrix-pipe form (k_bfknn2_frames_mfma: exact, keys of 11 index bits) for frames of up to 2048 keypoints; larger
    // frames take the vector-pipe kernel
    if (cap <= 2048) {
        size_t lds = 3 * KNN2M_TILE + (size_t)cap * 32; // three expanded tiles + the job's packed train rows (<= 90 KB)
        const dim3 mgrid((unsigned)((cap + KNN2M_QUERIES - 1) / KNN2M_QUERIES), (unsigned)njobs);
        // A grid that fits the chip once (64 jobs x 4 query blocks = 256 workgroups on 256 CUs) must not be packed two to a CU
        // with the rest of the chip idle -- the dispatcher does exactly that when two fit: 26.0 us per launch against 18.0
        // when each asks for more than half a CU's LDS (82 KB and more: 17.9-18.3 us; 81 KB still let two in).  Larger grids
        // keep their real size (two per CU then overlap), and so does a launch that shares the chip with extraction kernels
        // (`shared`): a workgroup that needs 96 KB waits longer for a CU there -- the cross-camera step of bench.py with two
        // extractions in flight: 0.196-0.199 ms padded, 0.190-0.193 not (three runs each).
        // (the last query block of every job does not count when it is the mostly empty one: it is dispatched last)
        const unsigned mainCols = (cap % KNN2M_QUERIES != 0 && mgrid.x > 1) ? mgrid.x - 1 : mgrid.x;
        if (!shared && (size_t)mainCols * mgrid.y <= 256) lds = std::max(lds, (size_t)96 * 1024);
        // (the attribute belongs to the CURRENT device's copy of the kernel: one high-water mark per device -- ADVICE r05)
        static std::atomic<size_t> ldsSet[kMaxDevices];
        if (lds > 64 * 1024 && ldsSet[device].load() < lds) {
            HIP_TRY(hipFuncSetAttribute((const void*)k_bfknn2_frames_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ldsSet[device].store(lds);
        }
        hipLaunchKernelGGL(k_bfknn2_frames_mfma, mgrid, dim3(KNN2M_THREADS), lds, st, d_jobs, cap, d_idx, d_dist, fillTail);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const dim3 grid((unsigned)((cap + 63) / 64), (unsigned)njobs);
    // few workgroups: more wavefronts per workgroup share the 64 queries (and fill the chip)
    if ((long)grid.x * njobs >= 2048)
        hipLaunchKernelGGL(k_bfknn2_frames<4>, grid, dim3(256), 0, st, d_jobs, cap, d_idx, d_dist, fillTail);
    else
        hipLaunchKernelGGL(k_bfknn2_frames<8>, grid, dim3(512), 0, st, d_jobs, cap, d_idx, d_dist, fillTail);
    HIP_TRY(hipGetLastError());
    return 0;
}

int orbfe_matcher_sync(int device)
{
    int r;
    if ((r = select_device(device)) < 0) return r;
    if (g_arena[device].stream) HIP_TRY(hipStreamSynchronize(g_arena[device].stream));
    return 0;
}

