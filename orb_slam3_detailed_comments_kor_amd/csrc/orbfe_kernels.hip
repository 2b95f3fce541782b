/*
 * orbfe_kernels.hip -- gfx950 (CDNA4) kernels of the ORB extractor.
 *
 *   K-PYR   k_pyr_fused (k_pyr_level0 / k_pyr_resize: unfused fallback)
 *                                         ComputePyramid            src/ORBextractor.cc:1152-1177
 *   K-FAST  k_fast_cells                  cell loop + cv::FAST      src/ORBextractor.cc:769-854
 *   K-QT    k_octree                      DistributeOctTree         src/ORBextractor.cc:537-761
 *                                         + the mono / stereo partition of each level (:1136-1143)
 *   K-DESC  k_orient_blur_desc            IC_Angle + GaussianBlur + computeOrbDescriptor + the output records
 *                                                                   src/ORBextractor.cc:75-145,1100-1147
 *   K-PACK  k_pack                        output slots + KannalaBrandt8 rays, only with orbfe_set_kb8
 *   K-UPLOAD k_upload                     image upload of the latency path (a frame or two per blocking call)
 *   K-BORDER k_border                     copyMakeBorder (only when mvImagePyramid is read)
 *
 * Integer / bitwise work, no MFMA.  64-lane wavefronts throughout (ballot masks are 64-bit).
 * All float expressions that must round like the reference's scalar code use the explicit
 * __f*_rn intrinsics (and the file is built with -ffp-contract=off), SURVEY.md Appendix D2.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "orbfe_geom.h"
#include "orbfe_sincos.h"
#include "orbfe_kb8.h"
#include "orb_pattern.inc"

#define WAVE 64

__device__ const int8_t ORB_PATTERN_31_DEV[256][4] = ORBFE_PATTERN_31_INIT;

// LDS hand-off between lanes of ONE wavefront (each wave owns its LDS region): DS operations of a
// wave are processed in order, so only the compiler must be kept from reordering.
#define WAVE_SYNC()                                          \
    do {                                                     \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                     \
    } while (0)

// ------------------------------------------------------------------ K-PYR
// Level 0: copy the caller's image into the aligned ROI of the pyramid slab.
__global__ __launch_bounds__(256) void k_pyr_level0(const uint8_t* __restrict__ src, size_t srcPitch,
                                                    size_t srcImgStride, uint8_t* __restrict__ pyr,
                                                    size_t pyrImgStride, OrbLevelGeom L0)
{
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int y = blockIdx.y;
    const int img = blockIdx.z;
    if (x4 >= L0.w) return;
    const uint8_t* s = src + (size_t)img * srcImgStride + (size_t)y * srcPitch + x4;
    uint8_t* d = pyr + (size_t)img * pyrImgStride + L0.roiOff + (size_t)y * L0.pitch + x4;
    if (x4 + 3 < L0.w) {
        uint32_t v = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)s[3] << 24);
        *reinterpret_cast<uint32_t*>(d) = v; // ROI rows are 64-B aligned
    } else {
        for (int k = 0; x4 + k < L0.w; k++) d[k] = s[k];
    }
}

// Level l from level l-1: cv::resize INTER_LINEAR 8UC1, 11-bit fixed point.
// The per-column / per-row source indices and weights come from host tables computed exactly
// as OpenCV computes them (double scale, float fractional part, cvRound to 1/2048).
__global__ __launch_bounds__(256) void k_pyr_resize(uint8_t* __restrict__ pyr, size_t pyrImgStride, OrbLevelGeom Ls,
                                                    OrbLevelGeom Ld, const OrbResizeX* __restrict__ xtab,
                                                    const OrbResizeY* __restrict__ ytab)
{
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int dy = blockIdx.y;
    const int img = blockIdx.z;
    if (x4 >= Ld.w) return;
    uint8_t* base = pyr + (size_t)img * pyrImgStride;
    const OrbResizeY ty = ytab[Ld.ytabOff + dy];
    const uint8_t* S0 = base + Ls.roiOff + (size_t)ty.sy0 * Ls.pitch;
    const uint8_t* S1 = base + Ls.roiOff + (size_t)ty.sy1 * Ls.pitch;
    uint8_t* D = base + Ld.roiOff + (size_t)dy * Ld.pitch + x4;
    uint32_t packed = 0;
    const int b0 = ty.b0, b1 = ty.b1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int dx = x4 + k;
        int v = 0;
        if (dx < Ld.w) {
            const OrbResizeX tx = xtab[Ld.xtabOff + dx];
            const int sx = tx.sx, sx1 = tx.pad;
            const int h0 = S0[sx] * tx.a0 + S0[sx1] * tx.a1;
            const int h1 = S1[sx] * tx.a0 + S1[sx1] * tx.a1;
            v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            v = min(max(v, 0), 255);
        }
        packed |= (uint32_t)v << (8 * k);
    }
    if (x4 + 3 < Ld.w) {
        *reinterpret_cast<uint32_t*>(D) = packed;
    } else {
        for (int k = 0; x4 + k < Ld.w; k++) D[k] = (uint8_t)(packed >> (8 * k));
    }
}

// x / d with a host-made reciprocal m = ceil(2^32 / d) (m == 0 encodes d == 1): exact while x*d < 2^32.
__device__ __forceinline__ int fast_div(unsigned x, unsigned m) { return m ? (int)__umulhi(x, m) : (int)x; }

// Whole pyramid in ONE launch.  A workgroup owns one ORBFE_PYR_TILE^2 tile of the coarsest level
// and walks the chain level 0 -> nlevels-1 through two LDS buffers: it loads the level-0 region its
// chain needs from the caller's image, and at every level interpolates the next region from LDS
// (cv::resize INTER_LINEAR fixed point, same tables as k_pyr_resize), writing the part it owns to
// the pyramid slab.  Halo pixels are recomputed by neighbouring workgroups from identical inputs,
// so every level is bit-identical to the level-by-level result; the image is read from HBM once
// and no level is ever read back.
__global__ __launch_bounds__(256) void k_pyr_fused(const uint8_t* __restrict__ src, size_t srcPitch,
                                                   size_t srcImgStride, uint8_t* __restrict__ pyr,
                                                   size_t pyrImgStride, const uint4* __restrict__ tileRecs,
                                                   int recBytes, int nlevels, int ntx, int nty, int bufBytes0,
                                                   int bufBytes1, int stageX, int imgCols, int imgBase,
                                                   int32_t* __restrict__ clearHdr /* 4 words or nullptr */,
                                                   int xcdAffine, uint32_t mPerImg /* reciprocals of ntx * nty */,
                                                   uint32_t mNtx /* and of ntx (fast_div) */)
{
    static_assert(sizeof(OrbPyrTileHdr) % 16 == 0, "the tables behind the header are read as uint4");
    // first kernel of a batch: clear the {fragile count, error flag, -, -} header the later kernels append to
    if (clearHdr && (blockIdx.x | blockIdx.y | blockIdx.z) == 0 && threadIdx.x < 4) clearHdr[threadIdx.x] = 0;
    // dynamic LDS: region buffer A | region buffer B | this tile's record (orbfe_geom.h: header, x groups, y entries)
    extern __shared__ __attribute__((aligned(16))) uint8_t pyr_lds[];
    uint8_t* bufA = pyr_lds;
    uint8_t* bufB = pyr_lds + bufBytes0;
    uint4* const tab = reinterpret_cast<uint4*>(pyr_lds + bufBytes0 + bufBytes1);
    const OrbPyrTileHdr* const H = reinterpret_cast<const OrbPyrTileHdr*>(tab);
    const uint4* const xsel = tab + sizeof(OrbPyrTileHdr) / 16;
    const uint4* const xa = xsel + stageX;
    const uint32_t* const xb = reinterpret_cast<const uint32_t*>(xa + stageX);
    const uint2* const yt = reinterpret_cast<const uint2*>(xb + ((stageX + 3) & ~3));
    const int tid = threadIdx.x;
    int ti = blockIdx.x, tj = blockIdx.y, img = (int)blockIdx.z;
    if (xcdAffine) {
        // whole images per XCD (the batch is a multiple of 8): the halo columns and rows neighbouring tiles
        // re-read, and the partial lines they write side by side, then meet in ONE L2
        const unsigned lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, k = lin >> 3;
        const unsigned perImg = gridDim.x * gridDim.y, j = (unsigned)fast_div(k, mPerImg), t = k - j * perImg;
        img = (int)(8u * j + (lin & 7u));
        tj = fast_div(t, mNtx);
        ti = (int)(t - (unsigned)tj * gridDim.x);
    }
    img += imgBase;
    uint8_t* base = pyr + (size_t)img * pyrImgStride;
    // the tile's record: one flat copy into LDS (it is complete at the barrier that ends the level-0 staging below, which
    // itself takes what it needs -- the level-0 ranges -- from the global copy through scalar loads)
    const uint4* const rec = tileRecs + (size_t)(tj * ntx + ti) * (size_t)(recBytes >> 4);
    for (int i = tid; i < (recBytes >> 4); i += 256) tab[i] = rec[i];
    const OrbPyrTileHdr* const G = reinterpret_cast<const OrbPyrTileHdr*>(rec); // (wave-uniform address)
    // level 0: stage the needed region of the input image, write the owned part
    int nW = G->xneed[0] - G->xlo[0], nH = G->yneed[0] - G->ylo[0];
    {
        const int sp = (nW + 3) & ~3; // LDS pitch of the level-0 region
        const int pitch0 = G->pitch[0];
        const int xlo = G->xlo[0], ylo = G->ylo[0];
        // (32-bit offsets from two wave-uniform bases: an image and a pyramid slab are below 4 GB)
        const uint8_t* const s0 = src + (size_t)img * srcImgStride;
        uint8_t* const d0 = base;
        const uint32_t sOrg = (uint32_t)ylo * (uint32_t)srcPitch + (uint32_t)xlo;
        const uint32_t dOrg = (uint32_t)G->roi[0] + (uint32_t)ylo * (uint32_t)pitch0 + (uint32_t)xlo;
        const int ownW = G->xown[0] - xlo, ownH = G->yown[0] - ylo;
        // dword granularity (global dword accesses may be unaligned).  A thread keeps one dword column
        // and walks down the rows, four loads in flight per step; no per-item division.
        const int ndw = (nW + 3) >> 2;           // dwords per region row (LDS pitch sp == 4*ndw)
        const int safeW = imgCols - xlo;         // bytes readable in a row without leaving the image row
        const unsigned rcp0 = G->recip[0];
        const int rr = rcp0 ? (int)__umulhi((unsigned)tid, rcp0) : tid; // tid / ndw
        const int c = 4 * (tid - rr * ndw);
        const int rpp = rcp0 ? (int)__umulhi(256u, rcp0) : 256;        // rows per pass
        if (rr < rpp) {
            const bool fullLoad = c + 4 <= safeW, colOwned = c < ownW, fullStore = c + 4 <= ownW;
            uint32_t po = sOrg + (uint32_t)rr * (uint32_t)srcPitch + (uint32_t)c;
            uint32_t qo = dOrg + (uint32_t)(rr * pitch0 + c);
            uint8_t* dq = bufA + rr * sp + c;
            const uint32_t pStep = (uint32_t)rpp * (uint32_t)srcPitch;
            const int qStep = rpp * pitch0, dStep = rpp * sp;
            for (int r = rr; r < nH; r += 4 * rpp) {
                uint32_t v[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    v[k] = 0;
                    if (r + k * rpp < nH) {
                        const uint8_t* pk = s0 + (po + (uint32_t)k * pStep);
                        if (fullLoad) {
                            __builtin_memcpy(&v[k], pk, 4);
                        } else if (safeW > c) { // the image's last 1..3 columns: one 16-bit and / or one 8-bit load
                            const int rem = safeW - c;
                            if (rem & 2) {
                                uint16_t h;
                                __builtin_memcpy(&h, pk, 2);
                                v[k] = h;
                            }
                            if (rem & 1) v[k] |= (uint32_t)pk[rem & 2] << (8 * (rem & 2));
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int rk = r + k * rpp;
                    if (rk < nH) {
                        *reinterpret_cast<uint32_t*>(dq + k * dStep) = v[k];
                        if (rk < ownH && colOwned) {
                            uint8_t* qk = d0 + (qo + (uint32_t)(k * qStep));
                            if (fullStore) {
                                __builtin_memcpy(qk, &v[k], 4);
                            } else {
                                const int rem = ownW - c; // 1..3
                                if (rem & 2) {
                                    const uint16_t h = (uint16_t)v[k];
                                    __builtin_memcpy(qk, &h, 2);
                                }
                                if (rem & 1) qk[rem & 2] = (uint8_t)(v[k] >> (8 * (rem & 2)));
                            }
                        }
                    }
                }
                po += 4u * pStep;
                qo += (uint32_t)(4 * qStep);
                dq += 4 * dStep;
            }
        }
    }
    __syncthreads();
    int xo = 0, yo = 0;
    const uint8_t* S = bufA;
    uint8_t* D = bufB;
    typedef unsigned short pyr_us2 __attribute__((ext_vector_type(2)));
    for (int l = 1; l < nlevels; l++) {
        const int xlo = H->xlo[l], ylo = H->ylo[l], gpitch = H->pitch[l];
        nW = H->xneed[l] - xlo;
        nH = H->yneed[l] - ylo;
        const int dp = (nW + 3) & ~3;
        const int ownW = H->xown[l] - xlo, ownH = H->yown[l] - ylo;
        // A thread keeps one group of 4 destination columns (its four x entries stay in registers) and
        // walks down the rows.  Per row: one 16-B y entry and, from each of the two source rows, three
        // ALIGNED dwords starting at the dword of the group's first source pixel (a misaligned LDS access
        // costs 64 cycles per wave on gfx950, tools/lds_rate.hip); two v_alignbyte_b32 turn them into
        // the 8 bytes that start at that pixel, and every destination pixel picks its source pair
        // (sx, sx+1) with one v_perm_b32 whose selector is a per-column constant (the host checks that the
        // four columns of any group span at most 7 source pixels).  Then two v_dot2_u32_u16 against the
        // packed (a0, a1) and two v_mul_hi_u32 against b << 16.  sx+1 instead of min(sx+1, w-1) is
        // harmless: a1 == 0 there.
        const int nG = dp >> 2;
        const unsigned rcp = H->recip[l];
        const int rr = rcp ? (int)__umulhi((unsigned)tid, rcp) : tid; // tid / nG
        const int g = tid - rr * nG;
        const int rowsPerPass = rcp ? (int)__umulhi(256u, rcp) : 256;  // 256 / nG (host guarantees nG <= 256)
        const int c0 = 4 * g;
        if (rr < rowsPerPass) {
            const uint4 selv = xsel[xo + g], av = xa[xo + g];
            const uint32_t xbw = xb[xo + g];
            const uint32_t xbase = xbw & 0xFFFFu, sh = xbw >> 16;
            const uint32_t sel[4] = {selv.x, selv.y, selv.z, selv.w}, aw[4] = {av.x, av.y, av.z, av.w};
            // destination addresses advance by whole passes (no per-row multiplies); the global one is a 32-bit offset from
            // the image's slab (a slab is below 4 GB: one scalar base + one vector offset per store, no 64-bit vector adds)
            uint32_t qo = (uint32_t)H->roi[l] + (uint32_t)((ylo + rr) * gpitch + xlo + c0);
            uint8_t* dq = D + rr * dp + c0;
            const int qStep = rowsPerPass * gpitch, dStep = rowsPerPass * dp;
            const bool colOwned = c0 < ownW, fullDword = c0 + 4 <= ownW;
            for (int r = rr; r < nH; r += rowsPerPass, qo += (uint32_t)qStep, dq += dStep) {
                uint8_t* const q = base + qo;
                const uint2 Yp = yt[yo + r];
                const uint4 Y = make_uint4(Yp.x & 0xFFFFu, Yp.x >> 16, Yp.y << 16, Yp.y & 0xFFFF0000u);
                const uint32_t* S0 = reinterpret_cast<const uint32_t*>(S + Y.x + xbase);
                const uint32_t* S1 = reinterpret_cast<const uint32_t*>(S + Y.y + xbase);
                const uint32_t d0 = S0[0], d1 = S0[1], d2 = S0[2];
                const uint32_t e0 = S1[0], e1 = S1[1], e2 = S1[2];
                const uint32_t W0 = __builtin_amdgcn_alignbyte(d1, d0, sh), W1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
                const uint32_t V0 = __builtin_amdgcn_alignbyte(e1, e0, sh), V1 = __builtin_amdgcn_alignbyte(e2, e1, sh);
                uint32_t t[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t p0 = __builtin_amdgcn_perm(W1, W0, sel[k]); // (I[sx], I[sx+1]) as two u16
                    const uint32_t p1 = __builtin_amdgcn_perm(V1, V0, sel[k]);
                    const pyr_us2 a = *reinterpret_cast<const pyr_us2*>(&aw[k]);
                    const uint32_t h0 = __builtin_amdgcn_udot2(*reinterpret_cast<const pyr_us2*>(&p0), a, 0u, false);
                    const uint32_t h1 = __builtin_amdgcn_udot2(*reinterpret_cast<const pyr_us2*>(&p1), a, 0u, false);
                    // cv::resize: (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2
                    t[k] = __umulhi(Y.z, h0 >> 4) + __umulhi(Y.w, h1 >> 4);
                }
                // (t + 2) >> 2 on two 16-bit fields at a time (t <= 1021: the host checks a0 + a1 <= 2050 and
                // b0 + b1 <= 2050, so the result is <= 255 and needs no saturation), then interleave the bytes
                const uint32_t u02 = (((t[0] | (t[2] << 16)) + 0x00020002u) >> 2) & 0x00FF00FFu;
                const uint32_t u13 = (((t[1] | (t[3] << 16)) + 0x00020002u) >> 2) & 0x00FF00FFu;
                const uint32_t packed = u02 | (u13 << 8);
                *reinterpret_cast<uint32_t*>(dq) = packed;
                if (r < ownH && colOwned) {
                    if (fullDword) {
                        __builtin_memcpy(q, &packed, 4);
                    } else { // the tile's last owned group: 1..3 bytes as at most one 16-bit and one 8-bit store
                        const int rem = ownW - c0;
                        if (rem & 2) {
                            const uint16_t h = (uint16_t)packed;
                            __builtin_memcpy(q, &h, 2);
                        }
                        if (rem & 1) q[rem & 2] = (uint8_t)(packed >> (8 * (rem & 2)));
                    }
                }
            }
        }
        __syncthreads();
        xo += nG;
        yo += nH;
        uint8_t* t = const_cast<uint8_t*>(S);
        S = D;
        D = t;
    }
}

// BORDER_REFLECT_101 frame of one level (19 px), written only when the caller asks for
// mvImagePyramid (nothing inside the extractor reads it).
__global__ __launch_bounds__(256) void k_border(uint8_t* __restrict__ pyr, size_t pyrImgStride, OrbLevelGeom L, int img)
{
    const int W = L.w + 2 * ORBFE_EDGE, H = L.h + 2 * ORBFE_EDGE;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= W * H) return;
    const int py = idx / W, px = idx - py * W;
    int x = px - ORBFE_EDGE, y = py - ORBFE_EDGE;
    if (x >= 0 && x < L.w && y >= 0 && y < L.h) return;
    int sx = x, sy = y;
    if (L.w == 1) sx = 0;
    else
        while (sx < 0 || sx >= L.w) sx = sx < 0 ? -sx : 2 * L.w - 2 - sx;
    if (L.h == 1) sy = 0;
    else
        while (sy < 0 || sy >= L.h) sy = sy < 0 ? -sy : 2 * L.h - 2 - sy;
    uint8_t* roi = pyr + (size_t)img * pyrImgStride + L.roiOff;
    roi[(ptrdiff_t)y * L.pitch + x] = roi[(ptrdiff_t)sy * L.pitch + sx];
}

// ----------------------------------------------------------------- K-FAST
// FAST-9/16 corner score = largest threshold for which the pixel is still a corner
// (cv::cornerScore<16>): max over the 16 arcs of 9 of min(r-v) and of min(v-r), minus 1.
// Every arc of 9 is {r[k]} + W or W + {r[k+9]} for one of the eight 8-windows W = r[k+1..k+8] with even k, and
// max(min(W, r[k]), min(W, r[k+9])) = min(W, max(r[k], r[k+9])): eight window minima by doubling (24 two-input
// minima), then three operations per window -- 47 per polarity instead of 16 + 16 three-input minima plus the
// reduction.  All of it on the raw 8-bit values (v is subtracted from the two results only) with the NON-packed
// 16-bit min/max, which issue at the fast rate on gfx950 (2.7 cycles per wave, tools/valu_rate.hip; v_min3_i32,
// v_min_i32 and the packed forms take 4.5).
__device__ __forceinline__ uint32_t min16(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t max16(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_max_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// P = tile pitch, a compile-time constant: the sixteen ring reads are ds_read_u8 with immediate offsets from one base
// (c - 3P - 3, so that every offset is non-negative and fits the instruction's offset field).
template <int P>
__device__ __forceinline__ int fast_score(const uint8_t* c)
{
    const int v = c[0];
    const uint8_t* b = c - 3 * P - 3;
#define ORBFE_RING(dy, dx) b[((dy) + 3) * P + (dx) + 3]
    uint32_t r[16];
    r[0] = ORBFE_RING(3, 0);
    r[1] = ORBFE_RING(3, 1);
    r[2] = ORBFE_RING(2, 2);
    r[3] = ORBFE_RING(1, 3);
    r[4] = ORBFE_RING(0, 3);
    r[5] = ORBFE_RING(-1, 3);
    r[6] = ORBFE_RING(-2, 2);
    r[7] = ORBFE_RING(-3, 1);
    r[8] = ORBFE_RING(-3, 0);
    r[9] = ORBFE_RING(-3, -1);
    r[10] = ORBFE_RING(-2, -2);
    r[11] = ORBFE_RING(-1, -3);
    r[12] = ORBFE_RING(0, -3);
    r[13] = ORBFE_RING(1, -3);
    r[14] = ORBFE_RING(2, -2);
    r[15] = ORBFE_RING(3, -1);
#undef ORBFE_RING
    uint32_t lo2[8], hi2[8], lo4[8], hi4[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { // r[2j+1], r[2j+2]
        lo2[j] = min16(r[2 * j + 1], r[(2 * j + 2) & 15]);
        hi2[j] = max16(r[2 * j + 1], r[(2 * j + 2) & 15]);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) { // r[2j+1 .. 2j+4]
        lo4[j] = min16(lo2[j], lo2[(j + 1) & 7]);
        hi4[j] = max16(hi2[j], hi2[(j + 1) & 7]);
    }
    uint32_t bright = 0, dark = 255; // max over arcs of the arc minimum / min over arcs of the arc maximum
#pragma unroll
    for (int j = 0; j < 8; j++) { // window r[2j+1 .. 2j+8], arcs starting at 2j and 2j+1
        const uint32_t lo8 = min16(lo4[j], lo4[(j + 2) & 7]), hi8 = max16(hi4[j], hi4[(j + 2) & 7]);
        const uint32_t a = r[2 * j], b2 = r[(2 * j + 9) & 15];
        bright = max16(bright, min16(lo8, max16(a, b2)));
        dark = min16(dark, max16(hi8, min16(a, b2)));
    }
    return max((int)bright - v, v - (int)dark) - 1;
}

// inclusive prefix sum over the 64 lanes with DPP (row shifts inside the 16-lane rows, row broadcasts across
// them): six dependent VALU instructions instead of six ds_bpermute round trips through the LDS
__device__ __forceinline__ int wave_incl_scan_i32(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true); // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true); // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true); // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true); // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, true); // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, true); // row_bcast:31 into rows 2 and 3
    return x;
}

// One LDS atomic per LANE (ds_add_rtn_u32 on a wave-uniform address, values differ per lane).  Written as
// inline asm because the compiler's atomic optimizer would otherwise turn it into a scalar loop over the
// active lanes (readlane / writelane per lane), which costs far more than the LDS serialising the adds.
__device__ __forceinline__ int lds_add_per_lane(int* p, int v)
{
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int*)p;
    int r;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(a), "v"(v) : "memory");
    return r;
}


// K-FAST: one 128-thread workgroup per FAST cell.  The reference makes one cv::FAST call per cell at iniThFAST and a second
// one at minThFAST when the first finds nothing (src/ORBextractor.cc:787-854) -- and the kernel does exactly that: a pass
// at iniThFAST and, only for a cell where it keeps nothing, a pass at minThFAST.  corner_at(t) <=> score >= t and the strict
// 8-neighbour NMS only ever loses to higher scores, so one score map serves both passes.  NMS does not see
// across the cell seam, exactly like per-cell cv::FAST.  Output: row-major ordered list per cell, packed x | y<<12 | s<<24.
//
// Round 3's form (round 2's kernel spent 0.53 scalar instructions per vector instruction, ten barriers per cell and a
// 190-instruction scalar prologue on per-workgroup overhead):
//  * the cell record is 32 B of host-folded values (OrbFastCell) fetched by one scalar load; the workgroup -> (image,
//    cell) mapping needs no division (the image group is the grid's y coordinate);
//  * the ROI is staged as aligned dwords (tile column 0 = level column iniX & ~3; ROI rows of the pyramid are 64-B aligned)
//    as a flat item list: tile dword i of the PD-pitched tile is LDS dword i, a thread's items are tid, tid + NT, ...,
//    five loads in flight, a clamp instead of a branch for the tail.  The tile pitch PD is a compile-time odd number of
//    dwords: every neighbour / ring / NMS read is an immediate offset from one address;
//  * phase A tests 4 pixels per lane from 5 dword LDS reads (SWAR on 6-bit-quantised bytes), phase B computes the exact
//    score of the queued survivors with all lanes busy;
//  * no corner queue and no survivor list: the NMS (phase C) walks the survivor queue itself (the score map says which
//    entries are corners at this threshold), marks the entries it keeps in place (bit 15) and sets their bit in a zone
//    bitmap; every wavefront then scans that bitmap for itself (no barrier) and writes its kept entries at their
//    row-major rank.  Four barriers per pass instead of ten, no LDS atomics except the queue counter;
//  * the pass at minThFAST has its own queue counter, so nothing has to be reset (and no barrier is needed) between
//    the passes.
// Measured and not kept: one wavefront per cell, a persistent grid with prefetched tiles, two pixels
// per lane with packed 16-bit min / max, a flat item list for phase A, 16 lanes per tile row for the staging.
// LDS: tile | score map | zone bitmap | its prefix sums | survivor queue (u16 per zone pixel).
#ifdef ORBFE_FAST_TIMING // tuning only (tools/ab_build.sh ft "-DORBFE_FAST_TIMING"): phase durations summed over all workgroups
// (spread over 4096 slots: atomics of all workgroups on one address would serialise and show up in the phases)
__device__ unsigned long long g_fastTimes[4096 * 16];
#define FT_BEGIN()                                \
    __shared__ unsigned ftL[16];                  \
    int ftPass = 0;                               \
    unsigned long long ftPrev = wall_clock64()
#define FT(k)                                                              \
    do {                                                                   \
        if (threadIdx.x == 0) {                                            \
            const unsigned long long ftNow = wall_clock64();               \
            ftL[k] = ((k) >= 3 && ftPass ? ftL[k] : 0u) + (unsigned)(ftNow - ftPrev); \
            ftPrev = ftNow;                                                \
        }                                                                  \
    } while (0)
#else
#define FT_BEGIN() do { } while (0)
#define FT(k) do { } while (0)
#endif
template <int NT, int PD>
__global__ __launch_bounds__(NT) void k_fast_cells(const uint8_t* __restrict__ pyr, size_t pyrImgStride,
                                              const OrbFastCell* __restrict__ cells, uint32_t* __restrict__ cand,
                                              size_t candImgStride, int32_t* __restrict__ cellCount, int nCellsTotal,
                                              int iniTh, int minTh, int tileBytes /* rows * 4 PD, multiple of 16 */,
                                              int bmWords /* multiple of 4 */, int gShift, int imgBase, int nImg,
                                              const uint2* __restrict__ pat /* phase A's per-thread constants, NT entries per
                                                                              pattern (cell width, alignment): .x = first centre
                                                                              dword | first zone row << 16 (all ones: no work),
                                                                              .y = column mask */)
{
    constexpr int P = 4 * PD;
    extern __shared__ __attribute__((aligned(16))) uint8_t fast_lds[];
    uint8_t* const tile = fast_lds;
    // the score map only has rows 2 .. rows-3 (scores live in rows 3 .. ch-4, the NMS looks one row further): its storage
    // starts where row 2 would be, `smap` is the virtual origin of row 0
    const int smapBytes = tileBytes - 4 * P;
    uint8_t* const smapStore = fast_lds + tileBytes;
    uint8_t* const smap = smapStore - 2 * P;
    uint32_t* const bm = reinterpret_cast<uint32_t*>(smapStore + smapBytes);
    int* const pre = reinterpret_cast<int*>(bm + bmWords);
    uint16_t* const queue = reinterpret_cast<uint16_t*>(pre + bmWords);
    __shared__ int qn[2]; // survivors of the pass at iniThFAST / at minThFAST

    const int tid = threadIdx.x, lane = tid & 63;
    // XCD-aware order (workgroups go round-robin over the 8 XCDs, each with its own L2).  gShift < 0: whole images per
    // XCD -- image xcd + 8 y; else groups of 1 << gShift row-neighbouring cells per XCD (small batches).
    const int xcd = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3);
    int cell, img;
    if (gShift < 0) {
        cell = slot;
        img = xcd + 8 * (int)blockIdx.y;
        if (img >= nImg) return;
    } else {
        cell = ((((slot >> gShift) << 3) + xcd) << gShift) + (slot & ((1 << gShift) - 1));
        img = (int)blockIdx.y;
        if (cell >= nCellsTotal) return;
    }
    img += imgBase;
    FT_BEGIN();
    const OrbFastCell c = cells[cell];
#ifdef ORBFE_FAST_TIMING
    asm volatile("" ::"s"(c.pitch)); // the record has arrived
    FT(0);
#endif
    const uint2 patE = pat[(size_t)c.pad * NT + tid]; // (arrives during the staging)
    const int cw = (int)(c.dims & 0xFFu), ch = (int)((c.dims >> 8) & 0xFFu), ox = (int)((c.dims >> 16) & 3u);
    const uint8_t* const gbase = pyr + (size_t)img * pyrImgStride + c.gOff;

    if (tid == 0) {
        qn[0] = 0;
        qn[1] = 0;
    }
    // clear the score map and the bitmap (contiguous), 16 B per store
    for (int o = tid * 16; o < smapBytes + 4 * bmWords; o += NT * 16) *reinterpret_cast<uint4*>(smapStore + o) = make_uint4(0u, 0u, 0u, 0u);
    // stage the ROI flat: tile dword i of the PD-pitched tile is LDS dword i (every row as PD dwords; the dwords past the
    // cell's own are never looked at), a thread's items are tid, tid + NT, ..., five loads in flight, a clamp instead of
    // a branch for the tail
    {
        const uint32_t nItems = (uint32_t)ch * PD, last = nItems - 1u;
        uint32_t* const T = reinterpret_cast<uint32_t*>(tile);
        // tile dword i sits in row i / PD: global byte offset = row * pitch + 4 (i - row PD) = 4 i + row (pitch - 4 PD); the
        // row by one multiply-high against ceil(2^32 / PD) (exact for the few thousand items of a tile)
        constexpr uint32_t MPD = 0xFFFFFFFFu / (uint32_t)PD + 1u;
        const uint32_t delta = c.pitch - 4u * (uint32_t)PD;
        for (uint32_t i0 = (uint32_t)tid; i0 < nItems; i0 += 5u * NT) {
            uint32_t v[5], idx[5];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                idx[k] = min(i0 + (uint32_t)(k * NT), last);
                const uint32_t row = __umulhi(idx[k], MPD);
                v[k] = *reinterpret_cast<const uint32_t*>(gbase + (__umul24(row, delta) + 4u * idx[k]));
            }
#pragma unroll
            for (int k = 0; k < 5; k++) T[idx[k]] = v[k];
        }
    }
    FT(1);
    __syncthreads();
    FT(2);

    const int zw = cw - 6, zh = ch - 6; // detection zone
    const int nz = (zw > 0 && zh > 0) ? zw * zh : 0;
    const int txLo = 3 + ox; // first zone column in tile coordinates (the last one is cw - 4 + ox)
    uint32_t* const out = cand + (size_t)img * candImgStride + c.slotBase;
    int th = iniTh, nk = 0;
    for (int pass = 0;; pass++) {
        // phase A: cheap necessary test.  Every arc of 9 contains one pixel of each opposite pair (k, k+8); test the pairs
        // (0,8) and (4,12).  The test is SWAR on all four pixels of a dword at once, on values quantised to 6 bits (x >> 2)
        // so that a byte field has room for the comparison: r > v + t implies (r >> 2) >= (v >> 2) + ((t + 1) >> 2) =:
        // q(v) + tb (floor((a + b) / 4) >= floor(a / 4) + floor(b / 4)), likewise for darker.  With H = q(v) + tb - 1 + 0x80
        // and L = q(v) - tb + 0x80 per byte, bit 7 of (H - q(r)) says "cannot be brighter" and bit 7 of (L - q(r)) "may be
        // darker"; no byte ever borrows from its neighbour (H - q(r) >= 0x7f - 63, L - q(r) >= 0x80 - 64 - 63 = 1, both
        // <= 254).  It lets ~a quarter more pixels through than the exact test would (they fail the exact score in phase
        // B), for 30 plain 32-bit operations per four pixels instead of 52.  At iniThFAST far fewer pixels survive it than
        // at minThFAST, so the usual cell scores a third of the pixels a single pass at min(iniTh, minTh) would.
        if (nz > 0) {
            const uint32_t Q = 0x3F3F3F3Fu;
            const int tb = (th + 1) >> 2;
            const uint32_t KH = (uint32_t)(tb - 1 + 0x80) * 0x01010101u, KL = (uint32_t)(0x80 - tb) * 0x01010101u;
            const uint32_t* const T = reinterpret_cast<const uint32_t*>(tile);
            // A thread keeps one dword column of the zone (4 pixels per row) and walks down the rows: addresses advance by
            // a constant and the column mask is a per-thread constant.  The order of the queue is irrelevant (scores go to
            // the map by position, the output is ranked by position): every lane reserves its own slots.
            // (r0 = tid / ndz: the thread's first zone row; its dword column d0 + tid % ndz; the mask of the zone columns
            // [txLo, txHi] among its four pixels: per-thread constants of the cell's pattern, folded on the host)
            const int rpp = fast_div((unsigned)NT, c.mNdz); // zone rows per pass
            const uint2 pe = patE;
            if (pe.x != 0xFFFFFFFFu) {
                const int r0 = (int)(pe.x >> 16);
                const uint32_t vM = pe.y;
                const int aStep = rpp * PD;
                int a = (int)(pe.x & 0xFFFFu); // dword index of the four centre pixels
                for (int r = r0; r < zh; r += rpp, a += aStep) {
                    const uint32_t C = T[a], Lf = T[a - 1], R = T[a + 1], U = T[a - 3 * PD], Dn = T[a + 3 * PD];
                    const uint32_t Cq = (C >> 2) & Q, Uq = (U >> 2) & Q, Dq = (Dn >> 2) & Q;
                    const uint32_t Lq = (__builtin_amdgcn_alignbyte(C, Lf, 1) >> 2) & Q;
                    const uint32_t Rq = (__builtin_amdgcn_alignbyte(R, C, 3) >> 2) & Q;
                    const uint32_t H = Cq + KH, L = Cq + KL;
                    const uint32_t notBright = ((H - Dq) & (H - Uq)) | ((H - Rq) & (H - Lq));
                    const uint32_t dark = ((L - Dq) | (L - Uq)) & ((L - Rq) | (L - Lq));
                    const uint32_t p = (~notBright | dark) & vM;
                    if (p) {
                        int s = lds_add_per_lane(&qn[pass], __popc(p));
                        const int pos0 = a << 2;
                        if (p & 0x80u) queue[s++] = (uint16_t)pos0;
                        if (p & 0x8000u) queue[s++] = (uint16_t)(pos0 + 1);
                        if (p & 0x800000u) queue[s++] = (uint16_t)(pos0 + 2);
                        if (p >> 31) queue[s] = (uint16_t)(pos0 + 3);
                    }
                }
            }
        }
        FT(3);
        __syncthreads();
        FT(4);
        // phase B: exact score of the survivors, all lanes busy; corners of this pass go to the score map
        const int nq = qn[pass];
        for (int qi = tid; qi < nq; qi += NT) {
            const int pos = queue[qi];
            const int sc = fast_score<P>(&tile[pos]);
            if (sc >= th) smap[pos] = (uint8_t)sc;
        }
        FT(5);
        __syncthreads();
        FT(6);
        const int nw = nq, wBase = 0, wDir = 1;
        // phase C: strict 8-neighbour NMS over the queue (an entry is a corner of this pass iff its score is >= th);
        // a kept entry is marked in place and sets its bit in the zone bitmap (row-major zone index)
        for (int qi = tid; qi < nw; qi += NT) {
            const int pos = queue[wBase + wDir * qi];
            const uint8_t* sp = smap + pos - P - 1; // immediate offsets below
            const int s0 = sp[P + 1];
            const int n0 = sp[P], n1 = sp[P + 2], n2 = sp[0], n3 = sp[1], n4 = sp[2], n5 = sp[2 * P], n6 = sp[2 * P + 1],
                      n7 = sp[2 * P + 2];
            const int mx = max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7)));
            if (s0 >= th && s0 > mx) {
                const int y = (int)((unsigned)pos / (unsigned)P), x = pos - y * P;
                const int z = (y - 3) * zw + (x - txLo);
                atomicOr(&bm[z >> 5], 1u << (z & 31));
                queue[wBase + wDir * qi] = (uint16_t)(pos | 0x8000);
            }
        }
        FT(7);
        __syncthreads();
        FT(8);
        // output: every wavefront scans the bitmap for itself (identical values, so the redundant stores to `pre` are
        // benign and no barrier is needed), then writes its kept entries at their rank
        const int nW = (nz + 31) >> 5;
        {
            int carry = 0;
            for (int base = 0; base < nW; base += 64) {
                const int i = base + lane;
                const int w = i < nW ? __popc(bm[i]) : 0;
                const int x = wave_incl_scan_i32(w);
                if (i < nW) pre[i] = x - w + carry;
                carry += __builtin_amdgcn_readlane(x, 63);
            }
            nk = carry;
        }
        if (nk > 0) {
            WAVE_SYNC();
            for (int qi = tid; qi < nw; qi += NT) {
                const int e = queue[wBase + wDir * qi];
                if (e & 0x8000) {
                    const int pos = e & 0x7FFF;
                    const int y = (int)((unsigned)pos / (unsigned)P), x = pos - y * P; // tile coordinates
                    const int z = (y - 3) * zw + (x - txLo);
                    const int rank = pre[z >> 5] + __popc(bm[z >> 5] & ((1u << (z & 31)) - 1u));
                    if (rank < (int)c.slotCap)
                        out[rank] = (uint32_t)(x - ox + (int)(c.off & 0xFFFFu)) | ((uint32_t)(y + (int)(c.off >> 16)) << 12) |
                                    ((uint32_t)smap[pos] << 24);
                }
            }
            break;
        }
        // nothing at iniThFAST: the same again with minThFAST (:825-828).  The queue may be overwritten at once (the
        // other wavefront is past its last read of it: nk == 0 skips the output loop), the bitmap is still clear.
        if (pass == 1 || minTh == iniTh) break;
        th = minTh;
#ifdef ORBFE_FAST_TIMING
        FT(9);
        ftPass = 1;
#endif
    }
    if (tid == 0) cellCount[(size_t)img * nCellsTotal + cell] = min(nk, (int)c.slotCap);
    FT(9);
#ifdef ORBFE_FAST_TIMING
    if (tid == 0) {
        unsigned long long* g = g_fastTimes + 16 * ((blockIdx.x + 977u * blockIdx.y) & 4095u);
        for (int k = 0; k < 10; k++) atomicAdd(&g[k], (unsigned long long)ftL[k]);
        atomicAdd(&g[15], 1ull);
    }
#endif
}

// ------------------------------------------------------------------- K-QT
#ifndef QT_THREADS
#define QT_THREADS 512 /* measured: 1024 -> 85 us, 512 -> 61 us, 256 -> 79 us (64 x 752x480) */
#endif
#define QT_WAVES (QT_THREADS / WAVE)
#ifndef ORBFE_QT_IMG_MAJOR
#define ORBFE_QT_IMG_MAJOR 1
#endif
#ifndef ORBFE_QT_FASTFWD
#define ORBFE_QT_FASTFWD 1 /* 0: always run the quadtree pass by pass (A/B, tools/ab_build.sh) */
#endif

// exclusive scan of a[0..n) in LDS, in place; returns the total to every thread.
// One barrier per 512-element chunk plus one at the end (every wave sums the <= 8 wave totals
// itself; the totals are double buffered so that a chunk never overwrites values still being read).
// F maps the stored element to the value that is scanned (identity for a plain scan): lets a caller fold the
// pass that would have prepared the scan input -- and its barrier -- into the scan itself.  G(i, exclusive
// prefix, value) runs where element i is written: a caller that only needs to look at (prefix, value) pairs once
// saves the pass and the barrier after the scan as well.
template <int NT, class T, class F, class G>
__device__ int qt_scan_map(const T* src, int* a, int n, int* wsum /* 2 * NT / 64 */, F f, G post)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int carry = 0, chunk = 0;
    for (int base = 0; base < n; base += NT, chunk ^= 1) {
        const int i = base + tid;
        const int v = i < n ? f(src[i]) : 0;
        const int x = wave_incl_scan_i32(v);
        int* ws = wsum + chunk * (NT / 64);
        if (lane == 63) ws[wave] = x;
        __syncthreads();
        int prefix = 0, total = 0;
#pragma unroll
        for (int w = 0; w < (NT / 64); w++) {
            const int t = ws[w];
            total += t;
            if (w < wave) prefix += t;
        }
        if (i < n) {
            const int e = x - v + prefix + carry;
            a[i] = e;
            post(i, e, v);
        }
        carry += total;
    }
    __syncthreads();
    return carry;
}
template <int NT, class T, class F>
__device__ int qt_scan_map(const T* src, int* a, int n, int* wsum, F f)
{
    return qt_scan_map<NT>(src, a, n, wsum, f, [](int, int, int) {});
}
template <int NT>
__device__ int qt_scan(int* a, int n, int* wsum /* 2 * NT / 64 */)
{
    return qt_scan_map<NT>(a, a, n, wsum, [](int v) { return v; });
}

// cc[idx] += 1 for every lane with idx >= 0, but with one LDS atomic per RUN of equal neighbouring indices: the
// keys of a level arrive in cell order, so in the first passes (a handful of nodes) whole stretches of a
// wavefront hit the same counter and plain per-lane atomics serialise 64 deep.
__device__ __forceinline__ void qt_hist_add(int* cc, int idx)
{
    const int lane = threadIdx.x & 63;
    const int prev = __builtin_amdgcn_update_dpp(-2, idx, 0x138, 0xF, 0xF, false); // wave_shr:1; lane 0 keeps -2
    const bool leader = idx >= 0 && idx != prev;
    const unsigned long long L = __ballot(leader), A = __ballot(idx >= 0);
    if (leader) {
        const unsigned long long rest = lane == 63 ? 0ull : (L >> (lane + 1));
        const int next = rest ? lane + 1 + (int)__builtin_ctzll(rest) : 64; // first lane of the next run
        const unsigned long long range = (next == 64 ? ~0ull : ((1ull << next) - 1ull)) & ~((1ull << lane) - 1ull);
        atomicAdd(&cc[idx], (int)__popcll(A & range)); // inactive lanes can only sit at the tail of a run
    }
}

__device__ __forceinline__ int qt_quadrant(int ul, int br, int x, int y)
{
    const int ulx = ul & 0xFFFF, uly = ul >> 16, brx = br & 0xFFFF, bry = br >> 16;
    const int hx = (brx - ulx + 1) >> 1, hy = (bry - uly + 1) >> 1; // ceil((float)d/2), d >= 0  (:481-482)
    return (x < ulx + hx ? 0 : 1) + (y < uly + hy ? 0 : 2);         // n1=0 n2=1 n3=2 n4=3 (:514-528)
}
__device__ __forceinline__ void qt_child(int ul, int br, int q, int& cul, int& cbr)
{
    const int ulx = ul & 0xFFFF, uly = ul >> 16, brx = br & 0xFFFF, bry = br >> 16;
    const int hx = (brx - ulx + 1) >> 1, hy = (bry - uly + 1) >> 1;
    const int x0 = (q & 1) ? ulx + hx : ulx, x1 = (q & 1) ? brx : ulx + hx;
    const int y0 = (q & 2) ? uly + hy : uly, y1 = (q & 2) ? bry : uly + hy;
    cul = x0 | (y0 << 16);
    cbr = x1 | (y1 << 16);
}

// Keys per thread that stay in REGISTERS for a whole level (key, list index of its node, child slot of the current
// pass): the usual level (one chunk of cells, <= QT_KPT * NT candidates) then walks registers, and a pass
// reads only the node tables from LDS -- kOf / UL / BR for the histogram, cpos / sidx for the relabelling -- instead
// of chasing keyNode -> kOf -> key -> UL / BR twice per key and pass through LDS.
#define QT_KPT 4
template <int NT, int J, class F>
__device__ __forceinline__ void qt_reg_step(int n, F& f)
{
    if (J * NT < n) f(std::integral_constant<int, J>(), (int)threadIdx.x + J * NT); // uniform guard
}
template <int NT, class F>
__device__ __forceinline__ void qt_each_key(bool regp, int n, F f)
{
    if (regp) {
        qt_reg_step<NT, 0>(n, f);
        qt_reg_step<NT, 1>(n, f);
        qt_reg_step<NT, 2>(n, f);
        qt_reg_step<NT, 3>(n, f);
    } else {
        for (int base = 0; base < n; base += NT) f(std::integral_constant<int, 0>(), base + (int)threadIdx.x);
    }
}

#if ORBFE_QT_IMG_MAJOR
#define QT_LEVEL_IDX levels.v[blockIdx.y]
#define QT_IMG_IDX blockIdx.x
#else
#define QT_LEVEL_IDX levels.v[blockIdx.x]
#define QT_IMG_IDX blockIdx.y
#endif
#ifdef ORBFE_QT_TIMING // tuning only (tools/ab_build.sh qtt "-DORBFE_QT_TIMING"): phase timestamps of one workgroup
__device__ unsigned long long g_qtTimes[64];
#ifndef ORBFE_QT_TIMING_LEVEL
#define ORBFE_QT_TIMING_LEVEL 0 /* the level whose workgroup (of image 0) leaves the stamps */
#endif
#define QT_STAMP(k)                                                                            \
    do {                                                                                       \
        if (threadIdx.x == 0 && QT_IMG_IDX == 0 && QT_LEVEL_IDX == ORBFE_QT_TIMING_LEVEL) g_qtTimes[k] = wall_clock64(); \
    } while (0)
// slowest workgroup per level since the last read: (duration in 10-ns ticks) << 16 | image, slots 30..37
#define QT_WG_BEGIN() const unsigned long long qtT0 = wall_clock64()
#define QT_WG_END()                                                                                                     \
    do {                                                                                                                \
        if (threadIdx.x == 0)                                                                                           \
            atomicMax(&g_qtTimes[30 + (QT_LEVEL_IDX & 7)], ((wall_clock64() - qtT0) << 16) | (QT_IMG_IDX & 0xFFFF));      \
    } while (0)
#else
#define QT_STAMP(k) do { } while (0)
#define QT_WG_BEGIN() do { } while (0)
#define QT_WG_END() do { } while (0)
#endif

// One workgroup per (image, level).  Level-synchronous restatement of DistributeOctTree
// (tests/qt_model.py is the executable specification, checked against the literal list
// transcription in the oracle): keys never move, every key carries the list index of its node,
// and each pass is a histogram + prefix sums.  Sort tie-break of :682 = creation order
// (SURVEY.md D1).
// GLOBAL (round 4): a level whose node tables do not fit a workgroup's LDS (24 ints per list entry: above ~1700 entries,
// i.e. nfeatures >~ 7800 at 8 levels -- the 5 x nFeatures initialisation extractor of src/Tracking.cc:1157 with KITTI's 2000)
// keeps them in a global scratch area of its own instead.  Same code: the arrays are reached through one base pointer, the
// atomics are generic, and __syncthreads orders a workgroup's global accesses as it orders its LDS accesses.  Slower (every
// table access is an L2 round trip), which only the few frames before the map is initialised pay.
struct OrbQtLevels {
    int32_t v[ORBFE_MAX_LEVELS]; // the levels this launch works on (grid coordinate -> level)
};
// The lapping ranges of a call of one or two images travel as kernel arguments: the host path keeps them in pinned HOST memory
// (no upload command for eight bytes), and reading them there was a PCIe round trip of ~2 us in every level's workgroup --
// behind its last barrier where it was first used (tools/qt_times.py: "output written" 1.96 us), in front of its first barrier
// when requested early (loads return in order).
struct OrbLapInline {
    int32_t n;    // images whose range is in v (0: read `lap`)
    int32_t v[4]; // lap0, lap1 of image 0, of image 1
};
template <bool GLOBAL, int NT>
__global__ __launch_bounds__(NT) void k_octree(const OrbLevelGeom* __restrict__ lg,
                                                       const OrbCellGeom* __restrict__ cg,
                                                       const uint32_t* __restrict__ cand, size_t candImgStride,
                                                       const int32_t* __restrict__ cellCount, int nCellsTotal,
                                                       uint32_t* __restrict__ keysAll, uint16_t* __restrict__ keyNodeAll,
                                                       size_t keyImgStride, uint32_t* __restrict__ lvlKp,
                                                       size_t kpImgStride, int32_t* __restrict__ lvlCount, int nlevels,
                                                       int32_t* __restrict__ errFlag, int imgBase, int keyLdsOff /* ints */,
                                                       int keyLdsCap /* keys */,
                                                       const int32_t* __restrict__ lap /* lapping range per image */,
                                                       uint32_t* __restrict__ lvlPre /* per keypoint slot: stereo flag << 15 |
                                                                                       stereo keypoints before it in its level */,
                                                       const OrbQtLevels levels, int* __restrict__ gScratch /* GLOBAL: node
                                                       tables, scratchStride ints per workgroup */, size_t scratchStride,
                                                       const OrbLapInline lapIn)
{
    extern __shared__ int lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#if ORBFE_QT_IMG_MAJOR
    // workgroup id = image + nimg * level: with batches that are a multiple of 8 an image's levels run on the XCD
    // (id mod 8) whose L2 K-FAST left the image's candidates in, and where K-PACK / K-DESC will read the result
    const int level = levels.v[blockIdx.y], img = (int)blockIdx.x + imgBase;
    const size_t wgLinear = (size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.y;
#else
    const int level = levels.v[blockIdx.x], img = (int)blockIdx.y + imgBase;
    const size_t wgLinear = (size_t)blockIdx.y + (size_t)gridDim.y * blockIdx.x;
#endif
    QT_STAMP(0);
    QT_WG_BEGIN();
    const OrbLevelGeom L = lg[level];
    const int LC = L.listCap;
    const int N = L.nFeat;

    // LDS: [0,64) scalars + scan scratch, then 24*LC ints of arrays; the first 1024 ints of the
    // array region double as the cell-count scan buffer of the gather (host sizes the allocation
    // as 64 + max(24*LC, 1024) ints).
    int* misc = lds;
    int* wsum = misc + 8;
    int* const A = GLOBAL ? gScratch + wgLinear * scratchStride : lds + 64;
    // (functions of the buffer index, not pointer arrays: indexing an array of pointers with the run-time `cur`
    // hides from the compiler that these are LDS addresses, and every access became a flat load / store)
    auto nodeUL = [A, LC](int b) { return A + b * LC; };
    auto nodeBR = [A, LC](int b) { return A + (2 + b) * LC; };
    auto nodeCnt = [A, LC](int b) { return A + (4 + b) * LC; };
    int* kOf = A + 6 * LC;       // list position -> expansion index k (or -1)
    int* sidx = A + 7 * LC;      // list position -> #expanded nodes before it
    int* cc = A + 8 * LC;        // [4*LC] child key counts, index 4k+q
    int* cpos = A + 12 * LC;     // [4*LC] child list positions, index 4k+q
    int* mpos = A + 16 * LC;     // [4*LC] packed scan: multi-key children (low 16) | non-empty children (high 16)
    auto multi = [A, LC](int b) { return A + (20 + b) * LC; }; // candidate list (list positions), creation order
    int* par = A + 22 * LC;      // expansion index k -> parent list position
    int* gpre = A + 23 * LC;     // growth prefix (final phase) / scratch
    int* gscan = GLOBAL ? lds + 64 : A; // gather only (always LDS)

    uint32_t* keys = keysAll + (size_t)img * keyImgStride + L.keyBase;
    uint16_t* keyNode = keyNodeAll + (size_t)img * keyImgStride + L.keyBase;
    const uint32_t* candImg = cand + (size_t)img * candImgStride;
    const int32_t* cnts = cellCount + (size_t)img * nCellsTotal + L.cellBase;
    const OrbCellGeom* cells = cg + L.cellBase;

    // (`keys` may point into LDS or into the global scratch, so the compiler reaches it with flat instructions; the two
    // accesses every level makes -- the gather's store and the retained key's look-up at the end -- go through keysL, the same
    // LDS words addressed as LDS, when keysInLds says so)
    uint32_t* const keysL = reinterpret_cast<uint32_t*>(lds + keyLdsOff);
    bool keysInLds = false; // (uniform)
    // Every pass walks the key arrays twice; keep them in LDS when the level's candidates fit (the
    // usual case), else in the global scratch arrays.  (Generic pointers: flat loads serve both.)
    // A level of up to NT cells (every level of a 752x480 frame) learns its total from the gather's own
    // scan below; only larger levels pay a separate pass over the cell counts (one more dependent global round trip).
    const bool oneChunk = L.nCells <= NT;
    if (!oneChunk) {
        int part = 0;
        for (int ci = tid; ci < L.nCells; ci += NT) part += cnts[ci];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
        if (lane == 0) wsum[wave] = part;
        __syncthreads();
        int total = 0;
        for (int w = 0; w < (NT / 64); w++) total += wsum[w];
        __syncthreads();
        if (total <= keyLdsCap) {
            keys = reinterpret_cast<uint32_t*>(lds + keyLdsOff);
            keyNode = reinterpret_cast<uint16_t*>(lds + keyLdsOff + keyLdsCap);
            keysInLds = true;
        }
    }

    // ---- gather the per-cell lists into one ordered key array (cells row-major, :787-853)
    // Flat over the output index: every thread finds the cell of its key by binary search in the
    // chunk's prefix array, so all global loads of the copy are independent (a per-cell copy loop
    // serialised three dependent global loads per cell and dominated this kernel).
    int n = 0;
    bool regp = false;
    uint32_t kReg[QT_KPT] = {0, 0, 0, 0};
    int nReg[QT_KPT] = {0, 0, 0, 0}, cReg[QT_KPT] = {-1, -1, -1, -1};
    int* gbase = gscan + NT; // slot base of the chunk's cells
    for (int cbase = 0; cbase < L.nCells; cbase += NT) {
        const int nc = min(NT, L.nCells - cbase);
        if (tid < nc) {
            gscan[tid] = cnts[cbase + tid];
            gbase[tid] = cells[cbase + tid].slotBase;
        }
        __syncthreads();
        QT_STAMP(5);
        const int tot = qt_scan<NT>(gscan, nc, wsum); // exclusive prefix of the counts
        QT_STAMP(6);
        if (oneChunk && tot <= keyLdsCap) {       // (uniform) the key arrays do not overlap gscan / gbase
            keys = reinterpret_cast<uint32_t*>(lds + keyLdsOff);
            keyNode = reinterpret_cast<uint16_t*>(lds + keyLdsOff + keyLdsCap);
            keysInLds = true;
        }
        regp = oneChunk && tot <= QT_KPT * NT; // (uniform)
        auto fetch = [&](int i) -> uint32_t {
            int lo = 0, hi = nc - 1; // last cell whose prefix <= i (cells with zero keys share a prefix; take the last)
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (gscan[mid] <= i) lo = mid;
                else hi = mid - 1;
            }
            return candImg[gbase[lo] + (i - gscan[lo])];
        };
        if (regp) {
            // the thread's four searches side by side: fixed steps 256 .. 1 (nc <= NT = 512), the four LDS reads of a
            // step independent of each other -- 9 LDS latencies in a row instead of 36 (four while-loops one after the other)
            static_assert(QT_KPT == 4 && NT >= 64 && (NT & (NT - 1)) == 0, "the interleaved search starts at NT / 2 (a power of two) and keeps four keys per thread");
            int lo4[QT_KPT] = {0, 0, 0, 0};
#pragma unroll
            for (int step = NT / 2; step >= 1; step >>= 1) {
                int m[QT_KPT], pv[QT_KPT];
#pragma unroll
                for (int j = 0; j < QT_KPT; j++) {
                    m[j] = lo4[j] + step;
                    pv[j] = gscan[min(m[j], nc - 1)];
                }
#pragma unroll
                for (int j = 0; j < QT_KPT; j++)
                    if (m[j] < nc && pv[j] <= tid + j * NT) lo4[j] = m[j];
            }
#pragma unroll
            for (int j = 0; j < QT_KPT; j++) {
                const int i = tid + j * NT;
                if (i < tot) {
                    kReg[j] = candImg[gbase[lo4[j]] + (i - gscan[lo4[j]])];
                    if (keysInLds) keysL[i] = kReg[j]; // the retained key of a node is looked up by index at the end
                    else keys[i] = kReg[j];
                }
            }
        } else {
            for (int i = tid; i < tot; i += NT) keys[n + i] = fetch(i);
        }
        n += tot;
        __syncthreads();
    }
    if (n > L.keyCap) n = L.keyCap; // cannot happen (keyCap = sum of slot caps)
    QT_STAMP(1);

    int cur = 0;
    int size = 0;
    // ---- roots (:540-583)
    const int nIni = L.nIni;
    if (n == 0 || nIni < 1) {
        if (tid == 0) lvlCount[(size_t)img * ORBFE_MAX_LEVELS + level] = 0;
        for (int p = tid; p < L.kpCap; p += NT) lvlPre[(size_t)img * kpImgStride + L.kpBase + p] = 0u; // no slot is valid
        return;
    }
    // ---- the first passes in closed form.  While 4 * (number of cells of a depth) <= N neither `size >= N` nor the
    // final-phase test `size + 3 * nToExpand > N` can fire there, so if in addition every node above depth FF holds
    // more than one key (a full pass divides every one of them) and every pass enlarges the list, the list after FF
    // full passes is known without running them: the non-empty depth-FF cells, in the order the push_fronts leave
    // them -- children of later parents first, n4 before n1, i.e. (parent position descending, quadrant descending),
    // which unrolls to the path digits compared in alternating direction.  One key walk bins the keys in that order
    // and the checks run on the bin counts; anything unusual falls through to the pass-by-pass code below.
    // (Replaying the loop's tests on the counts of depths 0..3 and entering the final phase directly skips one pass
    // more but costs as much as it saves: tried)
    int FF = 0;
#if ORBFE_QT_FASTFWD
    while (FF < 3 && 16 * nIni * (1 << (2 * FF)) <= N) FF++; // 4 * nIni * 4^d <= N for every depth d <= FF
#endif
    bool forwarded = false;
    if (FF > 0) {
        const int B = nIni << (2 * FF); // <= N / 4 < LC
        for (int i = tid; i < B; i += NT) cc[i] = 0;
        if (tid == 0) misc[1] = 0;
        __syncthreads();
        const int top = FF & 1; // the root digit runs backwards when FF is odd
        qt_each_key<NT>(regp, n, [&](auto J, int i) {
            constexpr int j = decltype(J)::value;
            int b = -1;
            if (i < n) {
                const uint32_t k = regp ? kReg[j] : keys[i];
                const int x = (int)(k & 0xFFF), y = (int)((k >> 12) & 0xFFF);
                int r = (int)__fdiv_rn((float)x, L.hX);
                r = min(r, nIni - 1);
                int ul = (int)__fmul_rn(L.hX, (float)r), br = (int)__fmul_rn(L.hX, (float)(r + 1)) | ((L.maxBY - ORBFE_MINB) << 16);
                b = top ? nIni - 1 - r : r;
                for (int d = 1; d <= FF; d++) {
                    const int q = qt_quadrant(ul, br, x, y);
                    int cul, cbr;
                    qt_child(ul, br, q, cul, cbr);
                    ul = cul;
                    br = cbr;
                    b = 4 * b + (((FF - d) & 1) ? q : 3 - q);
                }
                if (regp) cReg[j] = b;
                else keyNode[i] = (uint16_t)b;
            }
            qt_hist_add(cc, b); // (whole wavefronts)
        });
        __syncthreads();
        QT_STAMP(10);
        // checks: no node above depth FF with exactly one key; at every depth some node with two non-empty children.
        // One lane per (depth, node, child): the child's key count is a short run of bins, the node's total and its number of
        // non-empty children are sums over the quad of lanes (DPP).  (One thread per node summing its whole span -- 64 dependent
        // LDS reads for a root at FF = 3 -- was 0.8 of this kernel's 18 us for a single frame.)
        {
            int flags = 0;
            for (int d = 0; d < FF; d++) { // nodes of depth d = runs of 4^(FF-d) bins
                const int span = 1 << (2 * (FF - d)), quarter = span >> 2;
                const int nItems = (nIni << (2 * d)) * 4; // (a multiple of 4: the lanes of a quad are in range together)
                for (int it = tid; it < nItems; it += NT) {
                    const int g = it >> 2, c4 = it & 3;
                    int sub = 0;
                    for (int e = 0; e < quarter; e++) sub += cc[g * span + c4 * quarter + e];
                    int total = sub + __builtin_amdgcn_mov_dpp(sub, 0xB1, 0xF, 0xF, true); // quad_perm [1,0,3,2]
                    total += __builtin_amdgcn_mov_dpp(total, 0x4E, 0xF, 0xF, true);        // quad_perm [2,3,0,1]
                    int kids = sub > 0 ? 1 : 0;
                    kids += __builtin_amdgcn_mov_dpp(kids, 0xB1, 0xF, 0xF, true);
                    kids += __builtin_amdgcn_mov_dpp(kids, 0x4E, 0xF, 0xF, true);
                    if (total == 1) flags |= 1;
                    if (kids >= 2) flags |= 2 << d;
                }
            }
            // (one LDS atomic per wavefront: with a value per lane on one address the compiler's atomic optimizer emits a
            // scalar loop over the active lanes -- 1.0 of this kernel's 17 us for a single frame, tools/qt_times.py)
            flags |= __builtin_amdgcn_mov_dpp(flags, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
            flags |= __builtin_amdgcn_mov_dpp(flags, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
            flags |= __builtin_amdgcn_mov_dpp(flags, 0x141, 0xF, 0xF, true); // row_half_mirror
            flags |= __builtin_amdgcn_mov_dpp(flags, 0x140, 0xF, 0xF, true); // row_mirror
            flags = __builtin_amdgcn_readlane(flags, 0) | __builtin_amdgcn_readlane(flags, 16) | __builtin_amdgcn_readlane(flags, 32) |
                    __builtin_amdgcn_readlane(flags, 48);
            QT_STAMP(12);
            if (lane == 0 && flags) atomicOr(&misc[1], flags);
            QT_STAMP(13);
        }
        __syncthreads();
        QT_STAMP(11);
        forwarded = misc[1] == ((2 << FF) - 2); // every pass grew the list, no single-key node
        if (forwarded) {
            int* const ulW = nodeUL(0);
            int* const brW = nodeBR(0);
            int* const cntW = nodeCnt(0);
            const int* const ccR = cc;
            const float hX = L.hX;
            const int maxY = L.maxBY - ORBFE_MINB;
            size = qt_scan_map<NT>(cc, gpre, B, wsum, [](int c) { return c > 0 ? 1 : 0; },
                               [=](int b, int pos, int v) {
                                   if (!v) return;
                                   // the path of bin b: root digit, then FF quadrant digits (most significant first)
                                   const int rd = b >> (2 * FF);
                                   const int r = top ? nIni - 1 - rd : rd;
                                   int ul = (int)__fmul_rn(hX, (float)r), br = (int)__fmul_rn(hX, (float)(r + 1)) | (maxY << 16);
                                   for (int d = 1; d <= FF; d++) {
                                       const int dg = (b >> (2 * (FF - d))) & 3;
                                       int cul, cbr;
                                       qt_child(ul, br, ((FF - d) & 1) ? dg : 3 - dg, cul, cbr);
                                       ul = cul;
                                       br = cbr;
                                   }
                                   ulW[pos] = ul;
                                   brW[pos] = br;
                                   cntW[pos] = ccR[b];
                               });
            qt_each_key<NT>(regp, n, [&](auto J, int i) {
                constexpr int j = decltype(J)::value;
                if (i < n) {
                    if (regp) nReg[j] = gpre[cReg[j]];
                    else keyNode[i] = (uint16_t)gpre[keyNode[i]];
                }
            });
            __syncthreads();
        }
    }
    if (!forwarded) {
    if (tid < nIni) cc[tid] = 0;
    __syncthreads();
    qt_each_key<NT>(regp, n, [&](auto J, int i) { // whole wavefronts enter qt_hist_add
        constexpr int j = decltype(J)::value;
        int r = -1;
        if (i < n) {
            const uint32_t k = regp ? kReg[j] : keys[i];
            r = (int)__fdiv_rn((float)(k & 0xFFF), L.hX);
            r = min(r, nIni - 1);
            if (regp) nReg[j] = r;
            else keyNode[i] = (uint16_t)r;
        }
        qt_hist_add(cc, r);
    });
    __syncthreads();
    if (tid < nIni) gpre[tid] = cc[tid] > 0 ? 1 : 0;
    __syncthreads();
    size = qt_scan<NT>(gpre, nIni, wsum);
    if (tid < nIni && cc[tid] > 0) {
        const int p = gpre[tid];
        const int x0 = (int)__fmul_rn(L.hX, (float)tid), x1 = (int)__fmul_rn(L.hX, (float)(tid + 1));
        nodeUL(0)[p] = x0;                                 // UL = (x0, 0)
        nodeBR(0)[p] = x1 | ((L.maxBY - ORBFE_MINB) << 16); // BR = (x1, maxY-minY)
        nodeCnt(0)[p] = cc[tid];
    }
    __syncthreads();
    qt_each_key<NT>(regp, n, [&](auto J, int i) {
        constexpr int j = decltype(J)::value;
        if (i < n) {
            if (regp) nReg[j] = gpre[nReg[j]];
            else keyNode[i] = (uint16_t)gpre[keyNode[i]];
        }
    });
    __syncthreads();
    } // !forwarded

    // expansion step shared by the full passes and the final-phase rounds.
    // Preconditions: kOf/sidx valid for the current list; par[k] for k < nE; if !histDone, cc is
    // computed here for the nE expanded nodes, else cc[4k+q] already holds the counts.
    auto expand = [&](int nE, bool histDone, int& nMultiOut) -> int {
        const int* ul = nodeUL(cur);
        const int* br = nodeBR(cur);
        if (!histDone) { // cc[0 .. 4 nE) was cleared by the caller, before its last barrier
            qt_each_key<NT>(regp, n, [&](auto J, int i) {
                constexpr int j = decltype(J)::value;
                int c = -1;
                if (i < n) {
                    const int p = regp ? nReg[j] : (int)keyNode[i];
                    const int k = kOf[p];
                    if (k >= 0) {
                        const uint32_t key = regp ? kReg[j] : keys[i];
                        c = 4 * k + qt_quadrant(ul[p], br[p], key & 0xFFF, (key >> 12) & 0xFFF);
                    }
                }
                if (regp) cReg[j] = c; // the child slot is reused by the relabelling below
                qt_hist_add(cc, c);
            });
            __syncthreads();
        }
        // One packed scan over i = 4k+q: low half counts multi-key children (creation order), high
        // half counts non-empty children.  Children are push_front'ed, so their list order is the
        // REVERSE of i (n4,n3,n2,n1 of the last parent first): position = nChildren - 1 - (#non-empty before i).
        const int tot = qt_scan_map<NT>(cc, mpos, 4 * nE, wsum, [](int c) { return (c > 1 ? 1 : 0) | (c > 0 ? 0x10000 : 0); });
        const int nChildren = tot >> 16, nMulti = tot & 0xFFFF;
        const int nb = cur ^ 1;
        const int newSize = nChildren + (size - nE);
        if (newSize > LC) { // cannot happen; never write out of bounds
            if (tid == 0) atomicExch(errFlag, 1);
            nMultiOut = 0;
            return -1;
        }
        for (int i = tid; i < 4 * nE; i += NT) {
            const int k = i >> 2, q = i & 3;
            const int cnt = cc[i];
            if (cnt > 0) {
                const int pos = nChildren - 1 - (mpos[i] >> 16);
                cpos[i] = pos;
                const int p = par[k];
                int cul, cbr;
                qt_child(ul[p], br[p], q, cul, cbr);
                nodeUL(nb)[pos] = cul;
                nodeBR(nb)[pos] = cbr;
                nodeCnt(nb)[pos] = cnt;
                if (cnt > 1) multi(nb)[mpos[i] & 0xFFFF] = pos;
            }
        }
        __syncthreads();
        for (int p = tid; p < size; p += NT) {
            const int k = kOf[p];
            if (!(k >= 0 && k < nE)) {
                const int pos = nChildren + p - sidx[p];
                nodeUL(nb)[pos] = ul[p];
                nodeBR(nb)[pos] = br[p];
                nodeCnt(nb)[pos] = nodeCnt(cur)[p];
            }
        }
        qt_each_key<NT>(regp, n, [&](auto J, int i) {
            constexpr int j = decltype(J)::value;
            if (i >= n) return;
            if (regp) { // the child slot 4k+q is still in the register the histogram left it in
                const int c = cReg[j], p = nReg[j];
                nReg[j] = (c >= 0 && (c >> 2) < nE) ? cpos[c] : nChildren + p - sidx[p];
            } else {
                const int p = keyNode[i];
                const int k = kOf[p];
                int np;
                if (k >= 0 && k < nE) {
                    const uint32_t key = keys[i];
                    const int q = qt_quadrant(ul[p], br[p], key & 0xFFF, (key >> 12) & 0xFFF);
                    np = cpos[4 * k + q];
                } else {
                    np = nChildren + p - sidx[p];
                }
                keyNode[i] = (uint16_t)np;
            }
        });
        __syncthreads();
        cur = nb;
        nMultiOut = nMulti;
        return newSize;
    };

    QT_STAMP(2);
    int stampK = 3;
    bool finish = false;
    while (!finish) {
        QT_STAMP(stampK); stampK = min(stampK + 1, 40);
        // ---- full pass (:598-663): every node with more than one key is divided
        const int prevSize = size;
        // sidx[p] = number of divided nodes before p; kOf[p] = its own expansion index or -1 (written where the
        // scan writes, no separate pass).  The nodes were written before the barrier that ended the last pass.
        int* const kOfW = kOf;
        int* const parW = par;
        const int nE = qt_scan_map<NT>(nodeCnt(cur), sidx, size, wsum, [](int c) { return c > 1 ? 1 : 0; },
                                   [kOfW, parW](int p, int e, int v) {
                                       kOfW[p] = v ? e : -1;
                                       if (v) parW[e] = p;
                                   });
        if (nE == 0) break;
        for (int i = tid; i < 4 * nE; i += NT) cc[i] = 0; // histogram of expand(), cleared in this phase
        __syncthreads();
        int nMulti = 0;
        const int ns = expand(nE, false, nMulti);
        if (ns < 0) break;
        size = ns;
        if (size >= N || size == prevSize) {
            finish = true;
        } else if (size + 3 * nMulti > N) {
            // ---- final phase (:671-736): expand the largest nodes first until N is reached
            int m = nMulti;
            QT_STAMP(41);
            int stampF = 42;
            while (!finish) {
                QT_STAMP(stampF); stampF = min(stampF + 1, 58);
                const int prev2 = size;
                const int* mcur = multi(cur);
                // rank of candidate t in descending (count, creation index) order
                for (int p = tid; p < size; p += NT) kOf[p] = -1;
                // (count and creation index in one word where they fit: the rank below is then one comparison per pair)
                const bool packedRank = m <= 8 * 64 && n < 65536; // (uniform)
                for (int t = tid; t < m; t += NT) gpre[t] = packedRank ? (nodeCnt(cur)[mcur[t]] << 16) | t : nodeCnt(cur)[mcur[t]];
                for (int i = tid; i < 4 * m; i += NT) cc[i] = 0;
                for (int t = tid; t < m; t += NT) sidx[t] = 0; // rank accumulators (sidx is rebuilt below)
                if (tid == 0) misc[0] = m;
                __syncthreads();
                if (stampF == 43) QT_STAMP(20);
#ifdef ORBFE_QT_TIMING
                if (stampF == 43 && threadIdx.x == 0 && QT_IMG_IDX == 0 && QT_LEVEL_IDX == ORBFE_QT_TIMING_LEVEL) g_qtTimes[61] = ((unsigned long long)m << 32) | (unsigned)size;
#endif
                // rank = number of candidates ahead in (count descending, creation index descending) order.  All
                // eight wavefronts work on it: wavefront w compares every candidate t with its own slice of the
                // j range and adds the partial count (the plain loop -- one thread per t over all j, each LDS read
                // waited for -- took 4.4 of this kernel's 31 us at ~120 candidates).
                if (packedRank) {
                    // the wavefront's slice of j sits in one register (lane l: key of j0 + l), the candidates t it is compared
                    // with in up to eight more; per j one v_readlane and, per 64 candidates, a compare and an add-with-carry --
                    // no LDS read inside the loop (the form below: 1.4 us of this kernel's 16 for a single frame, this one 1.05)
                    const int per = (m + (NT / 64) - 1) / (NT / 64); // <= 64
                    const int j0 = wave * per, nj = min(m, j0 + per) - j0;
                    const unsigned mine = lane < nj ? (unsigned)gpre[j0 + lane] : 0u; // (0 is ahead of nobody: keys are >= 2 << 16)
                    // (one instantiation per number of 64-candidate groups: no test inside the loop)
                    auto rank_groups = [&](auto UC) {
                        constexpr int U = decltype(UC)::value;
                        unsigned tk[U];
                        int part[U];
#pragma unroll
                        for (int u = 0; u < U; u++) {
                            tk[u] = lane + 64 * u < m ? (unsigned)gpre[lane + 64 * u] : 0xFFFFFFFFu;
                            part[u] = 0;
                        }
                        if (stampF == 43) QT_STAMP(14);
                        // (four j per step, independent of each other; the lanes past the slice hold 0, which is ahead of nobody)
                        const int njU = __builtin_amdgcn_readfirstlane(nj);
                        for (int l = 0; l < njU; l += 4) {
                            const unsigned k0 = (unsigned)__builtin_amdgcn_readlane((int)mine, l),
                                           k1 = (unsigned)__builtin_amdgcn_readlane((int)mine, l + 1),
                                           k2 = (unsigned)__builtin_amdgcn_readlane((int)mine, l + 2),
                                           k3 = (unsigned)__builtin_amdgcn_readlane((int)mine, l + 3);
#pragma unroll
                            for (int u = 0; u < U; u++)
                                part[u] += (k0 > tk[u] ? 1 : 0) + (k1 > tk[u] ? 1 : 0) + (k2 > tk[u] ? 1 : 0) + (k3 > tk[u] ? 1 : 0);
                        }
                        if (stampF == 43) QT_STAMP(15);
#pragma unroll
                        for (int u = 0; u < U; u++)
                            if (lane + 64 * u < m && part[u]) atomicAdd(&sidx[lane + 64 * u], part[u]);
                        if (stampF == 43) QT_STAMP(16);
                    };
                    switch ((m + 63) >> 6) { // (uniform)
                    case 1: rank_groups(std::integral_constant<int, 1>()); break;
                    case 2: rank_groups(std::integral_constant<int, 2>()); break;
                    case 3: rank_groups(std::integral_constant<int, 3>()); break;
                    case 4: rank_groups(std::integral_constant<int, 4>()); break;
                    case 5: rank_groups(std::integral_constant<int, 5>()); break;
                    case 6: rank_groups(std::integral_constant<int, 6>()); break;
                    case 7: rank_groups(std::integral_constant<int, 7>()); break;
                    default: rank_groups(std::integral_constant<int, 8>()); break;
                    }
                } else {
                    const int per = (m + (NT / 64) - 1) / (NT / 64);
                    const int j0 = wave * per, j1 = min(m, j0 + per);
                    for (int t = lane; t < m; t += 64) {
                        const int ct = gpre[t];
                        int part = 0;
                        int j = j0;
                        for (; j + 4 <= j1; j += 4) {
                            const int c0 = gpre[j], c1 = gpre[j + 1], c2 = gpre[j + 2], c3 = gpre[j + 3];
                            part += (c0 > ct) || (c0 == ct && j > t);
                            part += (c1 > ct) || (c1 == ct && j + 1 > t);
                            part += (c2 > ct) || (c2 == ct && j + 2 > t);
                            part += (c3 > ct) || (c3 == ct && j + 3 > t);
                        }
                        for (; j < j1; j++) {
                            const int cj = gpre[j];
                            part += (cj > ct) || (cj == ct && j > t);
                        }
                        if (part) atomicAdd(&sidx[t], part);
                    }
                }
                __syncthreads();
                if (stampF == 43) QT_STAMP(21);
                for (int t = tid; t < m; t += NT) {
                    const int rank = sidx[t];
                    kOf[mcur[t]] = rank;
                    par[rank] = mcur[t];
                }
                __syncthreads();
                if (stampF == 43) QT_STAMP(22);
                {
                    const int* ul = nodeUL(cur);
                    const int* br = nodeBR(cur);
                    qt_each_key<NT>(regp, n, [&](auto J, int i) {
                        constexpr int j = decltype(J)::value;
                        int c = -1;
                        if (i < n) {
                            const int p = regp ? nReg[j] : (int)keyNode[i];
                            const int k = kOf[p];
                            if (k >= 0) {
                                const uint32_t key = regp ? kReg[j] : keys[i];
                                c = 4 * k + qt_quadrant(ul[p], br[p], key & 0xFFF, (key >> 12) & 0xFFF);
                            }
                        }
                        if (regp) cReg[j] = c;
                        qt_hist_add(cc, c);
                    });
                }
                __syncthreads();
                if (stampF == 43) QT_STAMP(23);
                // growth of candidate r = its non-empty children - 1; the exclusive prefix says where the list
                // size stands before r, and the one candidate at which it reaches N is the break of :728-729
                // (misc[0] was preset to m before the histogram's barrier)
                {
                    int* const cut = misc;
                    const int sz = size;
                    qt_scan_map<NT>(reinterpret_cast<const int4*>(cc), gpre, m, wsum,
                                [](int4 c) { return (c.x > 0) + (c.y > 0) + (c.z > 0) + (c.w > 0) - 1; },
                                [cut, sz, N](int r, int e, int g) {
                                    const int before = sz + e;
                                    if (before < N && before + g >= N) cut[0] = r + 1;
                                });
                }
                const int nE2 = misc[0];
                if (stampF == 43) QT_STAMP(24);
                qt_scan_map<NT>(kOf, sidx, size, wsum, [nE2](int k) { return (k >= 0 && k < nE2) ? 1 : 0; });
                if (stampF == 43) QT_STAMP(25);
                int nM2 = 0;
                const int ns2 = expand(nE2, true, nM2);
                if (stampF == 43) QT_STAMP(26);
                if (ns2 < 0) {
                    finish = true;
                    break;
                }
                size = ns2;
                m = nM2;
                if (size >= N || size == prev2) finish = true;
            }
        }
    }

    QT_STAMP(59);
    // ---- retain the best key of every node (:739-758): max response, first key wins ties
    unsigned* best = reinterpret_cast<unsigned*>(cc);
    for (int p = tid; p < size; p += NT) best[p] = 0;
    __syncthreads();
    qt_each_key<NT>(regp, n, [&](auto J, int i) {
        constexpr int j = decltype(J)::value;
        if (i >= n) return;
        const uint32_t key = regp ? kReg[j] : keys[i];
        atomicMax(&best[regp ? nReg[j] : (int)keyNode[i]], ((key >> 24) << 24) | (0xFFFFFFu - (unsigned)i));
    });
    __syncthreads();
    QT_STAMP(27);
    uint32_t* out = lvlKp + (size_t)img * kpImgStride + L.kpBase;
    const int nout = min(size, L.kpCap);
    // The mono / stereo partition of operator() (:1100-1147) is decided here, where the level's keypoints are final: a
    // keypoint whose level-0 x lies in the image's lapping range goes to the back of the output.  Its flag and the number of
    // such keypoints before it in the level go to lvlPre, the level's total into the high half of its count: K-DESC derives
    // every output slot from these (no K-PACK launch).  Bit 16 marks a slot that holds a keypoint of THIS batch (the slots
    // past the level's count are cleared), so that K-DESC's wavefronts know without the counts whether they have work.
    {
        const bool inl = img < lapIn.n; // (uniform)
        const float lap0 = (float)(inl ? lapIn.v[2 * img] : lap[2 * img]), lap1 = (float)(inl ? lapIn.v[2 * img + 1] : lap[2 * img + 1]),
                    scale = L.scale;
        uint32_t* const pre = lvlPre + (size_t)img * kpImgStride + L.kpBase;
        int run = 0, buf = 0;
        // the two usual ranges need no counting: (0, 0) of the rectified-stereo / RGB-D constructors holds no keypoint (a
        // keypoint's level-0 x is at least ORBFE_MINB), (0, 1000) of the monocular constructor (src/Frame.cc:306) holds
        // every keypoint of an image up to 1000 px wide
        const float sxMax = __fmul_rn((float)L.w, scale); // >= every keypoint's level-0 x (rounding is monotonic)
        const bool none = !(lap1 >= (float)ORBFE_MINB) || lap0 > lap1, all = !none && lap0 <= (float)ORBFE_MINB && lap1 >= sxMax;
        QT_STAMP(17);
        if (none || all) {
            const uint32_t flag = all ? 0x18000u : 0x10000u;
            for (int p = tid; p < nout; p += NT) {
                const uint32_t at = 0xFFFFFFu - (best[p] & 0xFFFFFFu);
                out[p] = keysInLds ? keysL[at] : keys[at];
                pre[p] = flag | (all ? (uint32_t)p : 0u);
            }
            run = all ? nout : 0;
            QT_STAMP(28);
        } else
        for (int base = 0; base < nout; base += NT, buf ^= 1) { // (uniform trip count; one barrier per round)
            const int p = base + tid;
            bool st = false;
            if (p < nout) {
                const uint32_t at = 0xFFFFFFu - (best[p] & 0xFFFFFFu);
                const uint32_t key = keysInLds ? keysL[at] : keys[at];
                out[p] = key;
                const float sx = __fmul_rn((float)((int)(key & 0xFFF) + ORBFE_MINB), scale); // keypoint->pt *= scale (:1131-1133)
                st = sx >= lap0 && sx <= lap1;
            }
            const unsigned long long m = __ballot(st);
            int* const wc = wsum + buf * (NT / 64);
            if (lane == 0) wc[wave] = __popcll(m);
            __syncthreads();
            int before = run;
#pragma unroll
            for (int w = 0; w < (NT / 64); w++) {
                const int c = wc[w];
                before += w < wave ? c : 0;
                run += c;
            }
            if (p < nout) pre[p] = 0x10000u | (st ? 0x8000u : 0u) | (uint32_t)(before + __popcll(m & ((1ull << lane) - 1ull)));
        }
        for (int p = nout + tid; p < L.kpCap; p += NT) pre[p] = 0u;
        QT_STAMP(29);
        if (tid == 0) lvlCount[(size_t)img * ORBFE_MAX_LEVELS + level] = nout | (run << 16);
    }
    QT_STAMP(60);
    QT_WG_END();
}

// ----------------------------------------------------------------- K-UPLOAD
// Latency path of a frame or two: the image comes out of page-locked HOST memory through this kernel (coalesced 16-byte
// reads over PCIe, every thread's requests in flight at once) instead of through a copy command: a copy engine between the
// host call and the first kernel costs a queue hand-over on each side of it.  src / dst are 16-byte aligned (the caller
// aligns the source down and gives the destination the same offset).
typedef unsigned int orbfe_u4v __attribute__((ext_vector_type(4)));
struct OrbUploadSegs { // up to two images per launch (a stereo pair): blockIdx.y selects the segment, so both stream at once
    const orbfe_u4v* src[2];
    orbfe_u4v* dst[2];
    unsigned n16[2];
};
__global__ __launch_bounds__(256) void k_upload(OrbUploadSegs segs)
{
    const orbfe_u4v* __restrict__ src = segs.src[blockIdx.y];
    orbfe_u4v* __restrict__ dst = segs.dst[blockIdx.y];
    const unsigned n16 = segs.n16[blockIdx.y];
    const unsigned stride = gridDim.x * 256u;
    unsigned i = blockIdx.x * 256u + threadIdx.x;
    for (; i + 3u * stride < n16; i += 4u * stride) {
        const orbfe_u4v a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride),
                        c = __builtin_nontemporal_load(src + i + 2u * stride),
                        d = __builtin_nontemporal_load(src + i + 3u * stride);
        dst[i] = a;
        dst[i + stride] = b;
        dst[i + 2u * stride] = c;
        dst[i + 3u * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}

// ----------------------------------------------------------------- K-PACK
// Output order and mono/stereo partition of ORBextractor::operator() (:1100-1147): level-major, list order inside a
// level; keypoints whose level-0 x lies in [lap0, lap1] fill the output from the back, the others from the front.
// Round 3: only launched when that partition is not the identity -- a lapping range that can hold a keypoint (fisheye
// rigs) -- or when bearing rays are wanted (orbfe_set_kb8).  It then writes the output slot of every keypoint slot
// (destMap, read by K-DESC), the counts and the rays; without it K-DESC derives everything itself (output slot = number of
// keypoints of the lower levels + index inside the level) and the pipeline is four launches.
#define PACK_THREADS 1024 /* one round for the usual <= 1024 keypoints per image */
__global__ __launch_bounds__(PACK_THREADS) void k_pack(const OrbLevelGeom* __restrict__ lg, int nlevels,
                                              const uint32_t* __restrict__ lvlKp, size_t kpImgStride,
                                              const int32_t* __restrict__ lvlCount, const int32_t* __restrict__ lap,
                                              int capPerImg, int32_t* __restrict__ destMap /* per keypoint slot */,
                                              int32_t* __restrict__ nOut, int32_t* __restrict__ monoOut,
                                              const float* __restrict__ kb8 /* or NULL */,
                                              float* __restrict__ raysOut /* 3 floats per kp, or NULL */, int imgBase,
                                              const int32_t* __restrict__ errIn /* the batch's error word or NULL */,
                                              int32_t* __restrict__ errOut /* where the host path reads it, or NULL */,
                                              int32_t* __restrict__ mirrorMeta /* [n | mono | err] in pinned HOST memory
                                                                                  (latency path of a frame or two), or NULL */,
                                              int mirrorImgs, const OrbLapInline lapIn)
{
    __shared__ int lvlOff[ORBFE_MAX_LEVELS + 1];
    __shared__ int waveCnt[PACK_THREADS / 64];
    __shared__ int runStereo;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int img = (int)blockIdx.x + imgBase;
    if (tid == 0) {
        int acc = 0;
        for (int l = 0; l < nlevels; l++) {
            lvlOff[l] = acc;
            acc += lvlCount[(size_t)img * ORBFE_MAX_LEVELS + l] & 0xFFFF; // (high half: K-QT's count of lapping-range keypoints)
        }
        lvlOff[nlevels] = acc;
        runStereo = 0;
    }
    __syncthreads();
    const int n = min(lvlOff[nlevels], capPerImg);
    const bool lapInl = img < lapIn.n; // (uniform; see OrbLapInline)
    const float lap0 = (float)(lapInl ? lapIn.v[2 * img] : lap[2 * img]), lap1 = (float)(lapInl ? lapIn.v[2 * img + 1] : lap[2 * img + 1]);
    for (int base = 0; base < n; base += PACK_THREADS) {
        const int g = base + tid;
        bool stereo = false;
        int level = 0;
        float sx = 0, sy = 0;
        if (g < n) {
            while (g >= lvlOff[level + 1]) level++;
            const uint32_t key = lvlKp[(size_t)img * kpImgStride + lg[level].kpBase + (g - lvlOff[level])];
            // keypoint->pt *= scale (:1131-1133); the scale of level 0 is 1
            sx = __fmul_rn((float)((int)(key & 0xFFF) + ORBFE_MINB), lg[level].scale);
            sy = __fmul_rn((float)((int)((key >> 12) & 0xFFF) + ORBFE_MINB), lg[level].scale);
            stereo = sx >= lap0 && sx <= lap1;
        }
        const unsigned long long m = __ballot(stereo);
        if (lane == 0) waveCnt[wave] = __popcll(m);
        __syncthreads();
        int sBefore = runStereo;
        for (int w = 0; w < wave; w++) sBefore += waveCnt[w];
        sBefore += __popcll(m & ((1ull << lane) - 1ull));
        if (g < n) {
            const int dest = stereo ? (n - 1 - sBefore) : (g - sBefore);
            destMap[(size_t)img * kpImgStride + lg[level].kpBase + (g - lvlOff[level])] = dest;
            // fisheye rigs: bearing ray of the keypoint, KannalaBrandt8::unproject fused into the pack
            if (kb8 && raysOut) orbfe_kb8_unproject_dev(kb8, sx, sy, raysOut + ((size_t)img * capPerImg + dest) * 3);
        }
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < PACK_THREADS / 64; w++) t += waveCnt[w];
            runStereo += t;
        }
        __syncthreads();
    }
    if (tid == 0) {
        nOut[img] = n;
        monoOut[img] = n - runStereo;
        // K-QT (the only kernel that raises the error word) has finished: hand the word to the host path's one
        // metadata transfer.  With sub-batches on several streams only the first sub-batch's K-QT is ordered before
        // this, which is why the host path runs on one stream.
        if (errOut && errIn && blockIdx.x == 0) errOut[0] = errIn[0];
        if (mirrorMeta) { // the caller reads these straight from pinned memory: no download command
            mirrorMeta[img] = n; // (the image's index in the CALL, not in the sub-batch: ADVICE r03)
            mirrorMeta[mirrorImgs + img] = n - runStereo;
            if (blockIdx.x == 0) mirrorMeta[2 * mirrorImgs] = errIn ? errIn[0] : 0;
        }
    }
}

// ----------------------------------------------------------------- K-DESC
__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// cv::fastAtan2: seven separately rounded single-precision operations.
// fmaHorner: the three inner Horner steps as fused multiply-adds -- what an OpenCV build evaluates whose AVX2
// translation unit of fastAtan2 was compiled with -mfma and contraction; default off.
__device__ __forceinline__ float fast_atan2_deg(float y, float x, bool fmaHorner = false)
{
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    // the two branches of the reference (ax >= ay: c = ay / (ax + eps); else c = ax / (ay + eps), a = 90 - a) differ
    // only in which of the two is the numerator: one division and one polynomial, then the branch's subtraction
    const bool steep = !(ax >= ay);
    const float num = steep ? ax : ay, den = steep ? ay : ax;
    const float c = __fdiv_rn(num, __fadd_rn(den, eps));
    const float c2 = __fmul_rn(c, c);
    float a;
    if (fmaHorner) a = __fmul_rn(__fmaf_rn(__fmaf_rn(__fmaf_rn(p7, c2, p5), c2, p3), c2, p1), c);
    else a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    if (steep) a = __fsub_rn(90.f, a);
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

__constant__ int8_t c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

// byte mask of the circular patch for IC_Angle: entry [|v|][d] covers patch columns 4(d+1) .. 4(d+1)+3
// (u = column - 21); a byte is 0xFF when |u| <= umax[|v|].
struct AngleMaskTab {
    uint32_t m[16][9];
};
__host__ __device__ constexpr AngleMaskTab make_angle_masks()
{
    const int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    AngleMaskTab t = {};
    for (int v = 0; v < 16; v++)
        for (int d = 0; d < 9; d++) {
            uint32_t m = 0;
            for (int k = 0; k < 4; k++) {
                const int u = 4 * (d + 1) + k - 21;
                if (u >= -umax[v] && u <= umax[v]) m |= 0xFFu << (8 * k);
            }
            t.m[v][d] = m;
        }
    return t;
}
__constant__ AngleMaskTab c_angleMask = make_angle_masks();

// IC_Angle as K-DESC evaluates it: item idx = lane + 64 k (k < 4) is one of the 213 dwords (row r, dword d of the 31 x 9
// that cover the circular patch's bounding box) that hold a pixel of the circle.  Everything about an item that does not depend on the pixels is in this
// table, one 16-byte entry per (k, lane): three tap words for v_dot4_u32_u8 -- byte j holds u + 21, v + 15 and 1 where the
// pixel (u = 4 (d + 1) + j - 21, v = r - 15) lies inside the circle, 0 elsewhere -- and the dword's byte offset in the
// wave's raw patch.  The moments are then three accumulating dot products per item (no masks, no multiplications, no
// index arithmetic): m10 = sum (u + 21) I - 21 sum I, m01 = sum (v + 15) I - 15 sum I.
struct IcTab {
    uint32_t t[4][64][4];
    int n;
};
__host__ __device__ constexpr IcTab make_ic_tab()
{
    const int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    IcTab t = {};
    // the 31 x 9 dwords that cover the circle's bounding box, row by row; the 66 that lie outside the circle altogether
    // are dropped: 213 items = four rounds of 64 lanes (unused entries: all-zero taps, offset 0)
    for (int r = 0; r < 31; r++)
        for (int d = 0; d < 9; d++) {
            const int v = r - 15, av = v < 0 ? -v : v;
            uint32_t tu = 0, tv = 0, t1 = 0;
            for (int j = 0; j < 4; j++) {
                const int u = 4 * (d + 1) + j - 21;
                if (u >= -umax[av] && u <= umax[av]) {
                    tu |= (uint32_t)(u + 21) << (8 * j);
                    tv |= (uint32_t)(v + 15) << (8 * j);
                    t1 |= 1u << (8 * j);
                }
            }
            if (t1 == 0u || t.n >= 256) continue;
            uint32_t* const e = t.t[t.n / 64][t.n % 64];
            e[0] = tu;
            e[1] = tv;
            e[2] = t1;
            e[3] = (uint32_t)((r + 6) * 44 + 4 * (d + 1));
            t.n++;
        }
    return t;
}
__constant__ IcTab c_icTab = make_ic_tab();
static_assert(make_ic_tab().n == 213, "every dword of the circle has a table entry");

// The items of K-DESC's horizontal pass that can matter.  A rotated tap lands within 18.385 (the largest norm of a pattern
// point) + 0.5 sqrt(2) (cvRound of both coordinates) of the patch centre, so the corners of the 37x37 blurred patch are
// never sampled: of the 19 x 10 (row pair, column group) items of the vertical pass 160 hold such a pixel, and of the
// 22 x 10 items of the horizontal pass the 190 that feed them (a vertical item reads the four row pairs below its first
// row) -- which fit three rounds of 64 lanes instead of four.  Entry = raw byte offset 88 pr + 4 hg | item index
// 10 pr + hg << 16, in ascending item order (the buffer-aliasing argument of the kernel's LDS layout needs ascending rows).
struct DescHItems {
    uint32_t t[192];
    uint32_t v[160]; // the vertical pass's items that hold a reachable pixel: H item index 10 q + g | blurred byte offset 80 q + 4 g << 16
    int nv;
    int n;
    int maxPr[3]; // last row pair of rounds 0..2
    int minPr[3]; // first row pair of rounds 0..2
};
__host__ __device__ constexpr DescHItems make_desc_h_items()
{
    DescHItems r = {};
    bool hneed[22][10] = {};
    for (int q = 0; q < 19; q++)
        for (int g = 0; g < 10; g++) {
            bool need = false;
            for (int y = 2 * q; y <= 2 * q + 1 && y < 37; y++)
                for (int x = 4 * g; x < 4 * g + 4 && x < 37; x++)
                    if ((x - 18) * (x - 18) + (y - 18) * (y - 18) <= 366) need = true; // (18.385 + 0.75)^2 = 366.15
            if (need) {
                for (int p = q; p <= q + 3; p++) hneed[p][g] = true;
                if (r.nv < 160) r.v[r.nv] = (uint32_t)(10 * q + g) | ((uint32_t)(80 * q + 4 * g) << 16);
                r.nv++;
            }
        }
    for (int k = 0; k < 3; k++) {
        r.maxPr[k] = -1;
        r.minPr[k] = 99;
    }
    for (int p = 0; p < 22; p++)
        for (int g = 0; g < 10; g++)
            if (hneed[p][g] && r.n < 192) {
                const int k = r.n / 64;
                r.t[r.n++] = (uint32_t)(88 * p + 4 * g) | ((uint32_t)(10 * p + g) << 16);
                if (p > r.maxPr[k]) r.maxPr[k] = p;
                if (p < r.minPr[k]) r.minPr[k] = p;
            }
    return r;
}
__constant__ DescHItems c_descHItems = make_desc_h_items();
constexpr DescHItems kDescHItems = make_desc_h_items();
static_assert(kDescHItems.n > 128 && kDescHItems.n <= 192, "three rounds of 64 lanes");
static_assert(kDescHItems.nv == 160, "vertical pass: two rounds of whole items and one of 32 items split into their two rows");

#define DESC_R 21    /* raw patch radius: 18 (rotated tap reach) + 3 (blur) */
// --------------------------------------------------------------- libm trig table
// ORBFE_TRIG_LIBM without a host round trip: for every float angle the IC_Angle can produce in
// [2^-7, 360] degrees the table holds how host libm's cosf/sinf (what the reference calls,
// src/ORBextractor.cc:110-111) differs from the correctly rounded value orbfe_sincos_cr yields on the
// device: 2 bits each for cos and sin (0 same, 1 next bit pattern up, 2 next bit pattern down, 3 other),
// one nibble per angle.  The host evaluates libm once per process for all ~1.3e8 angles and this kernel
// turns the values into codes; K-DESC then applies the code of its angle to its own (a, b).
#define ORBFE_TRIG_U0 0x3C000000u /* bits of 2^-7 deg: below, x = angle * pi/180 < 2^-12 and cos = 1, sin = x */
#define ORBFE_TRIG_U1 0x43B40000u /* bits of 360.0f (fastAtan2 can round up to it) */

// The device-side function the codes are relative to.  It does not have to be the correctly rounded value (that is
// orbfe_sincos_cr, ~64 double operations): any deterministic value within one bit pattern of libm's will do, and a value
// whose relative error stays below 0.4 float ulp always is (libm's is below 0.56 ulp, so both are one of the two floats
// that bracket the true value).  Double arithmetic with fused steps and Taylor series up to r^9 / r^10 on |r| <= pi/4
// (truncation < 2e-9 relative): ~25 instructions, which is what K-DESC pays per keypoint with the compact table.  The
// table build checks every angle (a code other than same / up / down raises `bad` and the mode falls back).
__device__ __forceinline__ void trig_tab_sincos(float angle, float* s_out, float* c_out)
{
    const double TWO_OVER_PI = 0.63661977236758134308;
    const double PIO2_HI = 1.57079632673412561417e+00;
    const double PIO2_LO = 6.07710050650619224932e-11;
    const double x = (double)angle;
    const double kd = floor(fma(x, TWO_OVER_PI, 0.5));
    const int ki = (int)kd;
    const double r = fma(-kd, PIO2_LO, fma(-kd, PIO2_HI, x));
    const double r2 = r * r;
    double ps = 1.0 / 362880.0;
    ps = fma(ps, r2, -1.0 / 5040.0);
    ps = fma(ps, r2, 1.0 / 120.0);
    ps = fma(ps, r2, -1.0 / 6.0);
    const float sr = (float)fma(r * r2, ps, r);
    double pc = -1.0 / 3628800.0;
    pc = fma(pc, r2, 1.0 / 40320.0);
    pc = fma(pc, r2, -1.0 / 720.0);
    pc = fma(pc, r2, 1.0 / 24.0);
    pc = fma(pc, r2, -0.5);
    const float cr = (float)fma(r2, pc, 1.0);
    // quadrant: (s, c) = (sr, cr), (cr, -sr), (-sr, -cr), (-cr, sr)
    const bool swap = ki & 1;
    const float s = swap ? cr : sr, c = swap ? sr : cr;
    *s_out = __uint_as_float(__float_as_uint(s) ^ ((ki & 2) ? 0x80000000u : 0u));
    *c_out = __uint_as_float(__float_as_uint(c) ^ (((ki + 1) & 2) ? 0x80000000u : 0u));
}

__device__ __forceinline__ unsigned trig_code(float libm, float cr)
{
    const int d = (int)__float_as_uint(libm) - (int)__float_as_uint(cr);
    return d == 0 ? 0u : d == 1 ? 1u : d == -1 ? 2u : 3u;
}
// one thread per PAIR of consecutive angles (one table byte); libmAB[i] = (cosf, sinf) of angle u0 + i
__global__ __launch_bounds__(256) void k_trig_codes(const float2* __restrict__ libmAB, uint32_t u0, uint32_t n,
                                                    uint8_t* __restrict__ table /* byte of angle u0 */,
                                                    int32_t* __restrict__ bad)
{
    const uint32_t i = 2u * (blockIdx.x * 256u + threadIdx.x);
    if (i >= n) return;
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    unsigned byte = 0;
    for (uint32_t k = 0; k < 2 && i + k < n; k++) {
        float sc, cc;
        trig_tab_sincos(__fmul_rn(__uint_as_float(u0 + i + k), factorPI), &sc, &cc);
        const float2 L = libmAB[i + k];
        const unsigned ca = trig_code(L.x, cc), cb = trig_code(L.y, sc);
        if (ca == 3u || cb == 3u) atomicAdd(bad, 1);
        byte |= (ca | (cb << 2)) << (4 * k);
    }
    table[i >> 1] = (uint8_t)byte; // u0 is even, so angle u0 + i is the low nibble
}
__device__ __forceinline__ float trig_apply(float v, unsigned code)
{
    return __uint_as_float(__float_as_uint(v) + (code == 1u ? 1u : code == 2u ? 0xFFFFFFFFu : 0u));
}
// The full table from the compact one, on the device: libm's (cosf, sinf) of angle u0 + i = the correctly rounded pair
// moved by its code.  1.03 GB written at HBM speed instead of being evaluated by the host and sent over PCIe.
__global__ __launch_bounds__(256) void k_trig_expand(const uint8_t* __restrict__ codes, uint32_t u0, uint32_t n,
                                                     float2* __restrict__ full)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float sc, cc;
    trig_tab_sincos(__fmul_rn(__uint_as_float(u0 + i), factorPI), &sc, &cc);
    const unsigned nib = (codes[i >> 1] >> (4u * (i & 1u))) & 0xFu;
    full[i] = make_float2(trig_apply(cc, nib & 3u), trig_apply(sc, nib >> 2));
}
// Checksum of the code table on the device (the cache file's payload is verified in full where it has just been uploaded:
// 65 MB at HBM speed instead of a host pass; the host has the same formula in trig_payload_sum).  Order-independent: the
// sum over the 8-byte words of mix(word + (index + 1) * C), so any number of threads may add their parts.
__host__ __device__ __forceinline__ unsigned long long trig_mix64(unsigned long long w, unsigned long long i)
{
    unsigned long long x = w + (i + 1ull) * 0x9E3779B97F4A7C15ull;
    x ^= x >> 32;
    x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 29;
    x *= 0xC2B2AE3D27D4EB4Full;
    x ^= x >> 32;
    return x;
}
__global__ __launch_bounds__(256) void k_trig_checksum(const unsigned long long* __restrict__ words, unsigned long long nWords,
                                                       unsigned long long* __restrict__ sum)
{
    unsigned long long acc = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x; i < nWords; i += (unsigned long long)gridDim.x * 256u)
        acc += trig_mix64(words[i], i);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(sum, acc);
}
// What K-DESC fetches for its angle as soon as the angle is known (the load then overlaps the blur): with the
// full table libm's (cosf, sinf) themselves, with the compact table the 4-bit code, else nothing.
struct TrigFetch {
    float2 ab;    // full table: libm values (valid when `have`)
    unsigned nib; // compact table: code
    bool have;
    bool tab;     // compact table in use: the codes are relative to trig_tab_sincos
};
__device__ __forceinline__ TrigFetch trig_fetch(const uint8_t* __restrict__ codes, const float2* __restrict__ full,
                                                float angleDeg)
{
    TrigFetch f;
    f.ab = make_float2(0.f, 0.f);
    f.nib = 0u;
    f.have = false;
    f.tab = false;
    const uint32_t idx = __float_as_uint(angleDeg) - ORBFE_TRIG_U0;
    const bool inTable = idx <= ORBFE_TRIG_U1 - ORBFE_TRIG_U0;
    if (full) {
        f.have = true;
        if (inTable) {
            f.ab = full[idx];
        } else { // below 2^-7 degrees: x < 2^-12 rad, cosf(x) == 1 and sinf(x) == x (checked when the table is built)
            f.ab = make_float2(1.0f, __fmul_rn(angleDeg, (float)(3.14159265358979323846 / 180.f)));
        }
    } else if (codes) {
        if (inTable) {
            f.tab = true;
            f.nib = ((unsigned)codes[idx >> 1] >> (4u * (idx & 1u))) & 15u;
        } else { // (as above)
            f.have = true;
            f.ab = make_float2(1.0f, __fmul_rn(angleDeg, (float)(3.14159265358979323846 / 180.f)));
        }
    }
    return f;
}
// a = cos, b = sin of the keypoint angle as the descriptor uses them (src/ORBextractor.cc:110-111)
__device__ __forceinline__ void trig_rotation(float angleDeg, const TrigFetch& f, float* a, float* b)
{
    if (f.have) { // wave-uniform in K-DESC
        *a = f.ab.x;
        *b = f.ab.y;
        return;
    }
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    if (f.tab) { // wave-uniform: compact table -- the cheap device value moved by its code = host libm's cosf / sinf, bit for bit
        trig_tab_sincos(__fmul_rn(angleDeg, factorPI), b, a);
        *a = trig_apply(*a, f.nib & 3u);
        *b = trig_apply(*b, f.nib >> 2);
    } else { // no table: the correctly rounded value (ORBFE_TRIG_CR, and the first guess of ORBFE_TRIG_LIBM_HOSTCHECK)
        orbfe_sincos_cr(__fmul_rn(angleDeg, factorPI), b, a);
    }
}
// test hook (orbfe_debug_trig): the rotation K-DESC would use for the given angles
__global__ __launch_bounds__(256) void k_debug_trig(const float* __restrict__ angles, int n,
                                                    const uint8_t* __restrict__ codes, const float2* __restrict__ full,
                                                    float* __restrict__ a, float* __restrict__ b)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    trig_rotation(angles[i], trig_fetch(codes, full, angles[i]), &a[i], &b[i]);
}

// Sum over the 64 lanes with DPP (quad permutes, half-row and row mirrors) and four v_readlane: no LDS
// traffic, unlike __shfl_xor (ds_bpermute); the result is wave-uniform.
__device__ __forceinline__ int wave_sum_i32(int v)
{
    v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true); // row_half_mirror
    v += __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true); // row_mirror: every lane holds its row's sum
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}

// one IEEE single multiplication, as an instruction the compiler cannot fuse or pair
__device__ __forceinline__ float fmul_single(float x, float y)
{
    float d;
    asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y));
    return d;
}

__device__ __forceinline__ float fma_single(float x, float k /* wave-uniform */, float z)
{
    float d;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "s"(k), "v"(z));
    return d;
}

#define DESC_RAW 43  /* raw patch side */
#define DESC_RAWP 44 /* raw pitch in bytes = 11 dwords */
#define DESC_BW 37   /* blurred patch side */
#define DESC_HP 40   /* pitch of the horizontal pass: 40 dwords per PAIR of rows (row 2p in the low, 2p+1 in the high half) */
#define DESC_HPAIRS 22 /* ceil(43/2) + 0: pair-rows 0..21 (row 43 is never used) */
#define DESC_BP 40   /* pitch of the blurred patch in bytes */
#define DESC_RAW_BYTES (DESC_RAW * DESC_RAWP + 28) /* + slack (last column group reads 12 B); multiple of 16 */
#define DESC_H_BYTES (DESC_HPAIRS * DESC_HP * 4)
#ifndef ORBFE_DESC_ALIAS
#define ORBFE_DESC_ALIAS 1 /* with four wavefronts per workgroup it made no difference (67.8 vs 67.9 us); with one it is what
                              lifts the kernel to eight wavefronts per SIMD: 62.4 -> 59.0 us */
#endif
#if ORBFE_DESC_ALIAS
// One region per wavefront holds all three buffers in turn (5440 -> 3520 bytes: eight wavefronts per SIMD instead of seven).
// The H buffer starts at the region's base and the raw patch DESC_RAW_OFF bytes into it.  Both passes work through their
// items in ascending row order, 64 items per round, and a round's LDS reads are issued before its writes (LDS executes a
// wavefront's instructions in order), so a buffer may overwrite rows of its source that no LATER round reads:
//  * horizontal pass, round k writes H pair-rows <= floor(6.4 k + 6.3) (160 B each) and later rounds read raw rows
//    >= 2 floor(6.4 (k + 1)) (44 B each): base + 160 (p + 1) <= base + OFF + 88 floor(6.4 (k + 1)) for k = 0, 1, 2 needs
//    OFF >= 592, 1024, 1528;
//  * vertical pass, round k writes blurred row pairs <= floor(6.4 k + 6.3) (80 B each) at the base and later rounds read H
//    pair-rows >= floor(6.4 (k + 1)) (160 B each): always below.
#define DESC_RAW_OFF 1536
// (the same condition for the compacted item list of the horizontal pass, whose rounds move through the row pairs faster
// where the corners are skipped: round k's last H row pair must end below the first raw rows round k + 1 reads)
static_assert(160 * (kDescHItems.maxPr[0] + 1) <= DESC_RAW_OFF + 88 * kDescHItems.minPr[1] &&
                  160 * (kDescHItems.maxPr[1] + 1) <= DESC_RAW_OFF + 88 * kDescHItems.minPr[2],
              "horizontal pass would overwrite raw rows it still has to read");
#define DESC_LDS_PER_WAVE (DESC_H_BYTES > DESC_RAW_OFF + DESC_RAW_BYTES ? DESC_H_BYTES : DESC_RAW_OFF + DESC_RAW_BYTES)
#else
#define DESC_LDS_PER_WAVE (DESC_RAW_BYTES + DESC_H_BYTES) /* blurred patch aliases the raw patch */
#endif

// Completion word of the latency path (a blocking call of a frame or two, results mirrored into page-locked host memory by
// this kernel): the host spins on a flag word the kernel publishes instead of waiting in hipStreamSynchronize for the
// end-of-kernel release and the runtime's signal (~5 us of a 12-us round trip, tools/latency_probe.hip).
// The flag may only land in host memory after every result store has.  A wavefront's own acknowledgements (s_waitcnt vmcnt(0))
// do not say that: they say that its stores have reached its XCD's L2, and the flag -- written by another wavefront, possibly on
// another XCD -- travels by a path of its own.  (Found by tests/test_gpu_keyframes.py's three host threads, which load the link:
// a search now and then read a row of its mirror before the row's stores had arrived.)  What does say it is a system-scope
// release (buffer_wbl2 sc0 sc1 + s_waitcnt: write-back of the XCD's L2 and a wait for it) issued on the SAME XCD after the
// acknowledgements -- and a thousand of those, one per K-DESC wavefront, queue behind each other: 0.066 -> 0.081 ms per frame.
// So the workgroups count per XCD: workgroups are dealt to the eight XCDs round-robin in linear-id order (tools/xcc_probe.hip
// reads the XCC_ID register: no exception on any grid), a workgroup counts itself -- after its own acknowledgements -- in the
// counter of the XCD its id says, the one that completes an XCD's count does ONE release there and counts the XCD, and the one
// that completes the eight writes the flag.  Every workgroup checks the register against its id: one that runs elsewhere
// releases its own stores before it counts; an XCD whose count is completed from elsewhere cannot be vouched for, and then the
// flag is not written at all -- the host's wait runs into its bound, synchronises the stream (always correct) and, after eight of
// those, stops using the word.
struct OrbDone {
    unsigned* ctr;  // 10 words of device memory, zero between calls (calls on one stream are ordered): [0..7] workgroups of the
                    // XCD done, [8] XCDs done, [9] "an XCD's count was completed from another XCD"
    unsigned* flag; // the kernel's address of the page-locked flag word; nullptr: none
    unsigned seq, nWaves /* reporting workgroups = the kernel's whole grid, in linear-id order */;
};
__device__ __forceinline__ unsigned orb_xcc_id()
{
    return (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; // hwreg(HW_REG_XCC_ID, 0, 4)
}
// By ONE wavefront of a reporting workgroup (all lanes), once every wavefront of the workgroup has waited for its own store
// acknowledgements.  L = the workgroup's linear id in the grid.
__device__ __forceinline__ void xcd_done(const OrbDone& d, unsigned L)
{
    const unsigned a = L & 7u;
    const bool off = orb_xcc_id() != a; // (uniform)
    if (off) __threadfence_system();    // not where the count assumes: its stores land by its own release
    unsigned closes = 0u;
    if ((threadIdx.x & 63) == 0) closes = atomicAdd(&d.ctr[a], 1u) + 1u == ((d.nWaves + 7u - a) >> 3) ? 1u : 0u;
    if (!__builtin_amdgcn_readfirstlane(closes)) return;
#ifndef ORBFE_DONE_ACK_ONLY // (defined: the acknowledgements alone, as first built -- to show the difference, never to ship)
    __threadfence_system(); // the XCD's last workgroup: everything its workgroups stored has reached this L2 -- now it lands
#endif
    if ((threadIdx.x & 63) == 0) {
        d.ctr[a] = 0u;
        if (off) { // (this L2 is not XCD a's)
            atomicExch(&d.ctr[9], 1u);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        }
        if (atomicAdd(&d.ctr[8], 1u) + 1u == min(d.nWaves, 8u)) {
            d.ctr[8] = 0u;
            if (atomicExch(&d.ctr[9], 0u) == 0u) {
                __threadfence_system();
                *(volatile unsigned*)d.flag = d.seq;
            } else {
                // (round 5) not vouched for: say so instead of saying nothing -- the host then synchronises the stream at once
                // instead of spinning into its bound.  Workgroups do leave the XCD their id names when kernels of OTHER queues
                // run beside this one (stereo frames in flight on several lanes: every second wait ran into the 400-us bound).
                // The marker carries no claim about the results, so it needs no ordering.
                *(volatile unsigned*)d.flag = d.seq | 0x80000000u;
            }
        }
    }
}
// The same word for a kernel of four-wavefront workgroups in which EVERY wavefront reports (K-STEREO): wavefronts count in
// LDS, the one that completes the workgroup's count reports for it; nWaves = number of workgroups (a one-dimensional grid).
// wg_done_begin before the first exit of any wavefront.
__device__ __forceinline__ void wg_done_begin(const OrbDone& d, unsigned* wgCnt)
{
    if (!d.flag) return; // (uniform)
    if (threadIdx.x == 0) *wgCnt = 0u;
    __syncthreads();
}
__device__ __forceinline__ void wg_done(const OrbDone& d, unsigned* wgCnt)
{
    if (!d.flag) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned closes = 0u;
    if ((threadIdx.x & 63) == 0) closes = atomicAdd(wgCnt, 1u) + 1u == 4u ? 1u : 0u;
    if (__builtin_amdgcn_readfirstlane(closes)) xcd_done(d, blockIdx.x);
}

// One wavefront per keypoint.  Stages the 43x43 raw neighbourhood in LDS, computes the
// intensity-centroid angle on the raw pixels (IC_Angle :75-102), blurs only the 37x37 patch the
// rotated taps can reach (GaussianBlur 7x7 sigma 2 fixed point, identical to blurring the whole
// level because the blur is local), then evaluates the 256 steered tests; the
// 32 descriptor bytes are four 64-bit ballots.
//   LDS traffic is kept wide: the patch is staged as dwords, the horizontal pass is
//   v_dot4_u32_u8 on (aligned / v_alignbyte-shifted) dwords producing 4 outputs per lane and one
//   ds_write_b64, the vertical pass reads 4 x u16 per ds_read_b64.
// MODE 0: trig from the libm table / orbfe_sincos_cr.
// MODE 2: the same, and keypoints whose sampling grid could differ under a 1-ulp change of sin/cos are appended to
//         fixList (ORBFE_TRIG_LIBM_HOSTCHECK; a mode of its own so that the hot instantiation does not carry the test).
// MODE 1: fix-up launch: one wave per fixList entry, trig (a, b) given in fixF.
template <bool SAT>
__device__ __forceinline__ uint32_t desc_hsat(uint32_t v)
{
    return SAT ? min(v, 65535u) : v;
}
// SAT: the taps sum to more than 256 (non-default taps only), so the horizontal pass can exceed 16 bits and
//      saturates like ufixedpoint16; with the default taps the sum is at most 255 * 256 and the min is dropped.
// DBG: the instantiation orbfe_debug_blurred_patch launches (the tap's loop would otherwise sit in the hot kernel).
#ifndef ORBFE_DESC_KPW
#define ORBFE_DESC_KPW 1 /* keypoint slots a wavefront works through in turn.  Two (half the wavefront launches, the item
                            tables loaded once for both): 58.1 against 54.6 us -- a fresh wavefront per keypoint lets the
                            hardware overlap one keypoint's patch loads with the others' arithmetic */
#endif
#ifndef ORBFE_DESC_WPW
#define ORBFE_DESC_WPW 1 /* wavefronts (= keypoint slots) per workgroup.  One: a workgroup's LDS is only released when its last
                            wavefront ends, so with four keypoints per workgroup a slow one (a border patch) keeps the LDS
                            of three finished ones; measured 64.7 (4) / 64.5 (2) / 62.4 (1) us, and 59.0 with one wavefront
                            per workgroup AND the aliased buffers below (eight wavefronts per SIMD) */
#endif
// (Round 6: K-DESC no longer reports to a call's completion word -- on the latency path the copy kernel behind it, k_mirror_out,
// is the call's last kernel and publishes it.)
template <int MODE, bool SAT, bool DBG = false>
__global__ __launch_bounds__(64 * ORBFE_DESC_WPW) void k_orient_blur_desc(const uint8_t* __restrict__ pyr, size_t pyrImgStride,
                                                          const OrbDescSlot* __restrict__ slots /* per keypoint slot */,
                                                          int nSlots,
                                                          const uint32_t* __restrict__ lvlKp /* K-QT's keypoints */,
                                                          size_t kpImgStride,
                                                          const int32_t* __restrict__ lvlCount /* 16 per image: count |
                                                                                  lapping-range keypoints << 16 */,
                                                          int nlevels,
                                                          const uint32_t* __restrict__ lvlPre /* K-QT's partition word per slot */,
                                                          const int32_t* __restrict__ destMap /* K-PACK's output slots, or
                                                                                               nullptr: identity partition */,
                                                          int capPerImg,
                                                          float* __restrict__ kpsOut, uint8_t* __restrict__ descOut,
                                                          const int* __restrict__ taps,
                                                          const float4* __restrict__ patternF,
                                                          int4* fixList /* MODE 0: out {img<<16|slot, angle, a, b} (float
                                                                           bits) after a 16-B header whose first word
                                                                           is the count; MODE 1: in {img<<16|slot, -, a, b} */,
                                                          int nFix, int listFragile, int imgBase, int xcdAffine,
                                                          const uint8_t* __restrict__ trigTab /* libm codes or nullptr */,
                                                          const float2* __restrict__ trigFull /* libm values or nullptr */,
                                                          int atanFma /* fused Horner steps in fastAtan2 (D2) */,
                                                          uint8_t* __restrict__ dbgPatch /* test tap: 37x37 blurred patch */,
                                                          int dbgDest /* ... of the keypoint with this output slot */,
                                                          int32_t* __restrict__ nOut = nullptr /* without K-PACK: counts, */,
                                                          int32_t* __restrict__ monoOut = nullptr /* error word and the    */,
                                                          const int32_t* __restrict__ errIn = nullptr /* mirrored metadata */,
                                                          int32_t* __restrict__ errOut = nullptr /* are written from here */,
                                                          int32_t* __restrict__ mirrorMeta = nullptr, int mirrorImgs = 0,
                                                          float* __restrict__ mirrorKps = nullptr /* the same outputs once more, */,
                                                          uint8_t* __restrict__ mirrorDesc = nullptr /* in pinned HOST memory   */)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_all[ORBFE_DESC_WPW][(DESC_LDS_PER_WAVE + 15) & ~15];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int img, g, imgLocal = 0;
    if (MODE != 1) {
        // XCD affinity (workgroups are dealt round-robin over the 8 XCDs in linear-id order): with a multiple of 8 images
        // the grid is (8 x workgroups per image, images / 8) and image = id mod 8 + 8 y, so every XCD keeps whole images to
        // itself and a pyramid is fetched into one L2 instead of eight (measured: 258 MB -> see DESIGN.md per 64 frames)
        if (xcdAffine) {
            imgLocal = (int)(blockIdx.x & 7u) + 8 * (int)blockIdx.y;
            g = ((int)(blockIdx.x >> 3) * ORBFE_DESC_WPW + wave) * ORBFE_DESC_KPW;
        } else {
            imgLocal = (int)blockIdx.y;
            g = ((int)blockIdx.x * ORBFE_DESC_WPW + wave) * ORBFE_DESC_KPW;
        }
        img = imgLocal + imgBase;
    } else {
        const int f = blockIdx.x * ORBFE_DESC_WPW + wave;
        if (f >= nFix) return;
        img = fixList[f].x >> 16;
        g = fixList[f].x & 0xFFFF;
    }
    // The work item is wave-uniform: three scalar loads, independent of each other (one round trip in front of the patch
    // loads): the slot's level geometry, K-QT's key in that slot and its partition word.
    img = __builtin_amdgcn_readfirstlane(img);
    auto desc_one = [&](int g) {
    g = __builtin_amdgcn_readfirstlane(g);
    if (g >= nSlots) return;
    typedef int i8v __attribute__((ext_vector_type(8)));
    i8v sv;
    uint32_t key, pre;
    const uint32_t slotIdx = (uint32_t)img * (uint32_t)kpImgStride + (uint32_t)g; // (a batch's slot arrays stay below 2^32 entries)
    {
        // one asm block so that the loads are in flight together (left to itself the compiler waits for each in turn)
        const OrbDescSlot* const sp = slots + g;
        const uint32_t* const kp = lvlKp + slotIdx;
        const uint32_t* const pp = lvlPre + slotIdx;
        asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dword %1, %4, 0x0\n\ts_load_dword %2, %5, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(sv), "=&s"(key), "=&s"(pre)
                     : "s"(sp), "s"(kp), "s"(pp)
                     : "memory");
    }
    // the image's level counts, one per lane (keypoints in the low half, lapping-range keypoints in the high half: neither
    // sum reaches 2^15): only the output slot needs them, at the end of the kernel
    const int cntL = lane < ORBFE_MAX_LEVELS ? lvlCount[(size_t)img * ORBFE_MAX_LEVELS + lane] : 0;
    const int level = sv[3] & 0xFF, kIn = (int)((unsigned)sv[3] >> 8);
    // sums over the levels below this one / over all levels: an inclusive scan across lanes 0..15
    auto level_sums = [&](int& below, int& total) {
        int inc = cntL;
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xF, 0xF, true); // row_shr:1
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xF, 0xF, true); // row_shr:2
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xF, 0xF, true); // row_shr:4
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xF, 0xF, true); // row_shr:8
        total = __builtin_amdgcn_readlane(inc, ORBFE_MAX_LEVELS - 1);
        below = __builtin_amdgcn_readlane(inc, level) - __builtin_amdgcn_readlane(cntL, level);
    };
    const bool speaks = MODE != 1 && !destMap && nOut && g == 0; // slot 0 of an image speaks for the image (K-PACK's
                                                                 // duties when it is not launched)
    auto image_outputs = [&](int total) {
        const int n = total & 0xFFFF, nStereo = (int)((unsigned)total >> 16);
        if (lane == 0) {
            nOut[img] = n;
            monoOut[img] = n - nStereo;
            if (errOut && errIn && imgLocal == 0) errOut[0] = errIn[0];
            if (mirrorMeta) { // the caller reads these straight from pinned memory: no download command
                mirrorMeta[img] = n; // (the image's index in the CALL, not in the sub-batch: ADVICE r03)
                mirrorMeta[mirrorImgs + img] = n - nStereo;
                if (imgLocal == 0) mirrorMeta[2 * mirrorImgs] = errIn ? errIn[0] : 0;
            }
        }
    };
    if (!(pre & 0x10000u)) { // wave-uniform: K-QT kept fewer keypoints at this level
        if (speaks) {
            int below, total;
            level_sums(below, total);
            image_outputs(total);
        }
        return;
    }
    struct {
        int x, y, dest;
    } w = {(int)(key & 0xFFF) + ORBFE_MINB, (int)((key >> 12) & 0xFFF) + ORBFE_MINB, 0};
    struct {
        int w, h, pitch;
    } L = {(int)(sv[2] & 0xFFFF), (int)((unsigned)sv[2] >> 16), sv[1]};
    const uint8_t* roi = pyr + (size_t)img * pyrImgStride + (uint32_t)sv[0];
#if ORBFE_DESC_ALIAS
    uint8_t* raw = s_all[wave] + DESC_RAW_OFF;
    uint16_t* hp = reinterpret_cast<uint16_t*>(s_all[wave]);
    uint8_t* bl = s_all[wave]; // the blurred patch overwrites the H rows the vertical pass has consumed
#else
    uint8_t* raw = s_all[wave];
    uint16_t* hp = reinterpret_cast<uint16_t*>(s_all[wave] + DESC_RAW_BYTES);
    uint8_t* bl = raw; // the blurred patch overwrites the raw patch once the horizontal pass is done
#endif

    // ---- IC_Angle table entries first: they do not depend on the patch, so their loads overlap the staging
    uint4 ict[4];
#pragma unroll
    for (int k = 0; k < 4; k++) ict[k] = *reinterpret_cast<const uint4*>(c_icTab.t[k][lane]);
    // ... and the horizontal pass's item list (three rounds; the test tap's instantiation computes all four)
    uint32_t hItem[3] = {0u, 0u, 0u}, vItem[3] = {0u, 0u, 0u};
    if (!DBG) {
#pragma unroll
        for (int k = 0; k < 3; k++) hItem[k] = c_descHItems.t[lane + 64 * k];
        vItem[0] = c_descHItems.v[lane];
        vItem[1] = c_descHItems.v[lane + 64];
        vItem[2] = c_descHItems.v[128 + (lane & 31)];
    }
    // ---- raw 43x43 patch (11 dwords per row; the 44th column is never used)
    const bool inside = w.x >= DESC_R && w.y >= DESC_R && w.x + DESC_R + 1 < L.w && w.y + DESC_R < L.h;
    if (inside) {
        const uint8_t* p0 = roi + (size_t)(w.y - DESC_R) * L.pitch + (w.x - DESC_R); // wave-uniform
        // Sixteen lanes per patch row, four rows per round, eleven rounds: the global offset advances by 4 rows per round
        // (one addition) and the LDS offset by a constant (an immediate); all loads in flight before the first LDS store;
        // global dword loads may be unaligned (the patch origin is arbitrary).  Lanes 11..15 of a row repeat column 10 and
        // the last round's fourth row repeats row 42 (identical stores): no branches.  (The flat item list this replaces --
        // item = lane + 64 k, 8 per lane -- needed a compare, two selects, two additions and a 64-bit address per item.)
        const uint32_t r4 = (uint32_t)lane >> 4, c4 = min((uint32_t)lane & 15u, 10u);
        const uint32_t pitch4 = 4u * (uint32_t)L.pitch;
        uint32_t off = r4 * (uint32_t)L.pitch + 4u * c4;
        const uint32_t rLast = min(40u + r4, 42u);
        uint32_t v[11];
#pragma unroll
        for (int k = 0; k < 10; k++) {
            __builtin_memcpy(&v[k], p0 + off, 4);
            off += pitch4;
        }
        __builtin_memcpy(&v[10], p0 + (rLast * (uint32_t)L.pitch + 4u * c4), 4);
        {
            uint32_t* dst = reinterpret_cast<uint32_t*>(raw + r4 * DESC_RAWP + 4u * c4);
#pragma unroll
            for (int k = 0; k < 10; k++) dst[k * DESC_RAWP] = v[k]; // (+ 4 rows = + 4 * 44 bytes = 44 dwords)
            *reinterpret_cast<uint32_t*>(raw + rLast * DESC_RAWP + 4u * c4) = v[10];
        }
    } else {
        // BORDER_REFLECT_101 at the level edges (keypoints within 21 px of an edge, ~7 % of them): the same 8 dword
        // items per lane, all loads in flight before the first LDS store.  Rows are reflected per item; a dword whose
        // four columns lie inside the level is one load, the few that straddle an edge are four reflected byte loads.
        // (A byte-by-byte loop over the 43x43 patch cost a border keypoint several times a normal one: 12 of this
        // kernel's 89 us.)
        const int x0 = w.x - DESC_R, y0 = w.y - DESC_R;
        uint32_t v[8];
        int c4 = lane - 11 * (lane / 11), row = lane / 11;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const bool valid = lane + 64 * k < DESC_RAW * 11;
            const int sy = reflect101(y0 + (valid ? row : 0), L.h);
            const uint8_t* rp = roi + (size_t)sy * L.pitch;
            const int xs = x0 + 4 * (valid ? c4 : 0);
            if (xs >= 0 && xs + 3 < L.w) {
                __builtin_memcpy(&v[k], rp + xs, 4);
            } else {
                v[k] = (uint32_t)rp[reflect101(xs, L.w)] | ((uint32_t)rp[reflect101(xs + 1, L.w)] << 8) |
                       ((uint32_t)rp[reflect101(xs + 2, L.w)] << 16) | ((uint32_t)rp[reflect101(xs + 3, L.w)] << 24);
            }
            const bool wrap = c4 >= 2;
            row += wrap ? 6 : 5;
            c4 += wrap ? -2 : 9;
        }
        uint32_t* dst = reinterpret_cast<uint32_t*>(raw) + lane;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (lane + 64 * k < DESC_RAW * 11) dst[64 * k] = v[k];
    }
    WAVE_SYNC();

    // ---- output slot (the counts were requested at the top; by now they have arrived)
    {
        int below, total;
        level_sums(below, total);
        if (speaks) image_outputs(total);
        if (destMap) {
            w.dest = destMap[slotIdx];
        } else {
            // output order of operator() (:1100-1147): level-major; lapping-range keypoints fill the output from the back
            const int n = total & 0xFFFF;
            const int gC = (below & 0xFFFF) + kIn, sBefore = (int)((unsigned)below >> 16) + (int)(pre & 0x7FFFu);
            w.dest = (pre & 0x8000u) ? n - 1 - sBefore : gC - sBefore;
        }
        if ((unsigned)w.dest >= (unsigned)capPerImg) return; // (cannot happen: cap >= the sum of the levels' capacities)
    }

    // ---- IC_Angle: m10 = sum u*I, m01 = sum v*I over the circular patch of radius 15.
    // items = (row, dword): rows 6..36, dwords 1..9 (columns 4..39 cover u = -15..15 = columns 6..36)
    int m10, m01;
    {
        uint32_t a10 = 0, a01 = 0, aS = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t px = *reinterpret_cast<const uint32_t*>(raw + ict[k].w);
            a10 = __builtin_amdgcn_udot4(px, ict[k].x, a10, false);
            a01 = __builtin_amdgcn_udot4(px, ict[k].y, a01, false);
            aS = __builtin_amdgcn_udot4(px, ict[k].z, aS, false);
        }
        m10 = (int)a10 - 21 * (int)aS;
        m01 = (int)a01 - 15 * (int)aS;
    }
    m10 = wave_sum_i32(m10);
    m01 = wave_sum_i32(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10, atanFma != 0);
    // libm codes of this angle (issued now, used after the blur)
    TrigFetch trigF;
    if (MODE != 1) trigF = trig_fetch(trigTab, trigFull, angle);

    // ---- separable 7-tap blur (8.8 taps; horizontal exact in u16, vertical 16.16 rounded)
    // (taps[0..6] are the seven taps; taps[8..25] the eighteen tap words below, packed by the host when the taps change:
    // forty-odd scalar shifts and ors per wavefront otherwise)
    const uint32_t* const tw = reinterpret_cast<const uint32_t*>(taps) + 8;
    // horizontal: item = (pair of rows 2p, 2p+1; group of 4 output columns) -> 22 x 10 items; the two
    // rows are packed as the low / high u16 of one dword so that the vertical pass can use
    // v_dot2_u32_u16 on vertically adjacent values.  Output j of a group needs source bytes j..j+6 of the
    // three aligned dwords (A, B, C) of its row: instead of shifting the DATA (v_alignbyte) the TAPS are
    // shifted -- ten v_dot4_u32_u8 per row against wave-uniform tap words, no byte shuffles.
    uint32_t* hp2 = reinterpret_cast<uint32_t*>(hp);
    {
        // A0 = t0 | t1 << 8 | t2 << 16 | t3 << 24, B0 = t4 | t5 << 8 | t6 << 16;  A1 = t0 << 8 | t1 << 16 | t2 << 24,
        // B1 = t3 | t4 << 8 | t5 << 16 | t6 << 24;  A2 = t0 << 16 | t1 << 24, B2 = t2 | t3 << 8 | t4 << 16 | t5 << 24, C2 = t6;
        // A3 = t0 << 24, B3 = t1 | t2 << 8 | t3 << 16 | t4 << 24, C3 = t5 | t6 << 8
        const uint32_t A0 = tw[0], B0 = tw[1], A1 = tw[2], B1 = tw[3], A2 = tw[4], B2 = tw[5], C2 = tw[6], A3 = tw[7], B3 = tw[8],
                       C3 = tw[9];
        // item idx = 10 * pr + hg = lane + 64 k: the H buffer is linear in idx (16 B per item); the raw offset
        // 88 * pr + 4 * hg advances by 544 (pr += 6, hg += 4) or, when hg wraps, by 592 (pr += 7, hg -= 6).
        // Pair-row 21 reads "row 43" past the patch: its values only ever meet a zero tap (row 43 is the
        // high half of the last pair), and the bytes lie inside this wave's LDS region.
        auto h_item = [&](int off, int idx) {
            const uint32_t* sa = reinterpret_cast<const uint32_t*>(raw + off);
            const uint32_t a0 = sa[0], a1 = sa[1], a2 = sa[2];
            const uint32_t b0 = sa[11], b1 = sa[12], b2 = sa[13]; // next row: + DESC_RAWP bytes
#define ORBFE_H0(A, B, C) desc_hsat<SAT>(__builtin_amdgcn_udot4(B, B0, __builtin_amdgcn_udot4(A, A0, 0u, false), false))
#define ORBFE_H1(A, B, C) desc_hsat<SAT>(__builtin_amdgcn_udot4(B, B1, __builtin_amdgcn_udot4(A, A1, 0u, false), false))
#define ORBFE_H2(A, B, C) \
    desc_hsat<SAT>(__builtin_amdgcn_udot4(C, C2, __builtin_amdgcn_udot4(B, B2, __builtin_amdgcn_udot4(A, A2, 0u, false), false), false))
#define ORBFE_H3(A, B, C) \
    desc_hsat<SAT>(__builtin_amdgcn_udot4(C, C3, __builtin_amdgcn_udot4(B, B3, __builtin_amdgcn_udot4(A, A3, 0u, false), false), false))
            uint4 o;
            // (low halves of two sums into one dword: one v_perm_b32 instead of a shift and an or)
            o.x = __builtin_amdgcn_perm(ORBFE_H0(b0, b1, b2), ORBFE_H0(a0, a1, a2), 0x05040100u);
            o.y = __builtin_amdgcn_perm(ORBFE_H1(b0, b1, b2), ORBFE_H1(a0, a1, a2), 0x05040100u);
            o.z = __builtin_amdgcn_perm(ORBFE_H2(b0, b1, b2), ORBFE_H2(a0, a1, a2), 0x05040100u);
            o.w = __builtin_amdgcn_perm(ORBFE_H3(b0, b1, b2), ORBFE_H3(a0, a1, a2), 0x05040100u);
#undef ORBFE_H0
#undef ORBFE_H1
#undef ORBFE_H2
#undef ORBFE_H3
            reinterpret_cast<uint4*>(hp2)[idx] = o;
        };
        if (!DBG) {
            // the 190 items whose output a rotated tap can reach (c_descHItems), three rounds
#pragma unroll
            for (int k = 0; k < 3; k++)
                if (lane + 64 * k < kDescHItems.n) h_item((int)(hItem[k] & 0xFFFFu), (int)(hItem[k] >> 16));
        } else {
            int hg = lane - 10 * (lane / 10);
            int off = 88 * (lane / 10) + 4 * hg;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (lane + 64 * k < DESC_HPAIRS * 10) h_item(off, lane + 64 * k);
                const bool wrap = hg >= 6;
                off += wrap ? 592 : 544;
                hg += wrap ? -6 : 4;
            }
        }
    }
    WAVE_SYNC();
    // vertical: item = (pair of output rows 2k, 2k+1; group of 4 columns) -> 19 x 10 items = three nearly
    // full rounds.  Both rows read the same four row pairs of the H buffer (H rows 2k..2k+7): the even row
    // weighs them (t0,t1)(t2,t3)(t4,t5)(t6,0), the odd one (0,t0)(t1,t2)(t3,t4)(t5,t6).  Output row 37 (second
    // row of the last pair) does not exist; it lands in unused bytes of the patch buffer.
    {
        typedef __attribute__((ext_vector_type(2))) unsigned short us2;
        union U2 {
            uint32_t u;
            us2 v;
        };
        U2 e0, e1, e2, e3, o0, o1, o2, o3;
        // e: t0 | t1 << 16, t2 | t3 << 16, t4 | t5 << 16, t6;   o: t0 << 16, t1 | t2 << 16, t3 | t4 << 16, t5 | t6 << 16
        e0.u = tw[10];
        e1.u = tw[11];
        e2.u = tw[12];
        e3.u = tw[13];
        o0.u = tw[14];
        o1.u = tw[15];
        o2.u = tw[16];
        o3.u = tw[17];
        // one output row of an item: four columns x four u16-pair dot products against the row's tap pairs, packed to four bytes
        auto v_row = [&](const uint4& p0, const uint4& p1, const uint4& p2, const uint4& p3, uint32_t w0, uint32_t w1, uint32_t w2,
                         uint32_t w3) -> uint32_t {
            U2 t0, t1, t2, t3;
            t0.u = w0;
            t1.u = w1;
            t2.u = w2;
            t3.u = w3;
            uint32_t a[4]; // 16.16 sums of the columns x y z w
#define ORBFE_VCOL(F, I)                                                              \
    {                                                                                  \
        U2 q0, q1, q2, q3;                                                             \
        q0.u = p0.F;                                                                   \
        q1.u = p1.F;                                                                   \
        q2.u = p2.F;                                                                   \
        q3.u = p3.F;                                                                   \
        uint32_t acc = __builtin_amdgcn_udot2(q0.v, t0.v, 32768u, false);              \
        acc = __builtin_amdgcn_udot2(q1.v, t1.v, acc, false);                          \
        acc = __builtin_amdgcn_udot2(q2.v, t2.v, acc, false);                          \
        a[I] = __builtin_amdgcn_udot2(q3.v, t3.v, acc, false);                         \
    }
            ORBFE_VCOL(x, 0)
            ORBFE_VCOL(y, 1)
            ORBFE_VCOL(z, 2)
            ORBFE_VCOL(w, 3)
#undef ORBFE_VCOL
            if (SAT) // taps that sum to more than 256: the result can exceed 255 and saturates (ufixedpoint16 -> uchar)
                return min(a[0] >> 16, 255u) | (min(a[1] >> 16, 255u) << 8) | (min(a[2] >> 16, 255u) << 16) | (min(a[3] >> 16, 255u) << 24);
            // <= 255 * 256 * 256 + 32768: byte 2 of each sum IS the pixel; three v_perm_b32 gather four of them
            // (ds_write_b8_d16_hi -- which stores exactly that byte -- instead of the v_perm_b32 and a dword store: measured,
            // 64.7 -> 65.5 us)
            return __builtin_amdgcn_perm(__builtin_amdgcn_perm(a[3], a[2], 0x0C0C0602u), __builtin_amdgcn_perm(a[1], a[0], 0x0C0C0602u),
                                         0x05040100u);
        };
        const uint4* const H4 = reinterpret_cast<const uint4*>(hp2);
        if (!DBG) {
            // The 160 items that hold a pixel a rotated tap can reach (c_descHItems.v, ascending): two rounds of whole items,
            // and the last 32 split into their two rows over the 64 lanes (lane >= 32: the odd row), so that the third round
            // costs half of a whole one.
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const uint4* sp4 = H4 + (vItem[k] & 0xFFFFu);
                const uint32_t boff = vItem[k] >> 16;
                const uint4 p0 = sp4[0], p1 = sp4[DESC_HP / 4], p2 = sp4[2 * (DESC_HP / 4)], p3 = sp4[3 * (DESC_HP / 4)];
                const uint32_t outE = v_row(p0, p1, p2, p3, e0.u, e1.u, e2.u, e3.u), outO = v_row(p0, p1, p2, p3, o0.u, o1.u, o2.u, o3.u);
                *reinterpret_cast<uint32_t*>(bl + boff) = outE;
                *reinterpret_cast<uint32_t*>(bl + boff + DESC_BP) = outO;
            }
            {
                const bool odd = lane >= 32;
                const uint4* sp4 = H4 + (vItem[2] & 0xFFFFu);
                const uint32_t boff = (vItem[2] >> 16) + (odd ? DESC_BP : 0);
                const uint4 p0 = sp4[0], p1 = sp4[DESC_HP / 4], p2 = sp4[2 * (DESC_HP / 4)], p3 = sp4[3 * (DESC_HP / 4)];
                *reinterpret_cast<uint32_t*>(bl + boff) =
                    v_row(p0, p1, p2, p3, odd ? o0.u : e0.u, odd ? o1.u : e1.u, odd ? o2.u : e2.u, odd ? o3.u : e3.u);
            }
        } else {
            int hg = lane - 10 * (lane / 10);
            int boff = 2 * DESC_BP * (lane / 10) + 4 * hg; // byte offset of output row 2k in the blurred patch
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (lane + 64 * k < 19 * 10) {
                    const uint4* sp4 = H4 + lane + 64 * k;
                    const uint4 p0 = sp4[0], p1 = sp4[DESC_HP / 4], p2 = sp4[2 * (DESC_HP / 4)], p3 = sp4[3 * (DESC_HP / 4)];
                    *reinterpret_cast<uint32_t*>(bl + boff) = v_row(p0, p1, p2, p3, e0.u, e1.u, e2.u, e3.u);
                    *reinterpret_cast<uint32_t*>(bl + boff + DESC_BP) = v_row(p0, p1, p2, p3, o0.u, o1.u, o2.u, o3.u);
                }
                const bool wrap = hg >= 6;
                boff += wrap ? (7 * 2 * DESC_BP - 24) : (6 * 2 * DESC_BP + 16);
                hg += wrap ? -6 : 4;
            }
        }
    }
    WAVE_SYNC();

    if (DBG && dbgPatch && w.dest == dbgDest) { // test tap (orbfe_debug_blurred_patch): GaussianBlur's output under this keypoint
#pragma nounroll
        for (int i = lane; i < DESC_BW * DESC_BW; i += 64) dbgPatch[i] = bl[(i / DESC_BW) * DESC_BP + i % DESC_BW];
    }
    // ---- steered BRIEF (:106-145)
    float a, b;
    if (MODE != 1) {
        trig_rotation(angle, trigF, &a, &b);
    } else {
        const int f = blockIdx.x * ORBFE_DESC_WPW + wave;
        a = __int_as_float(fixList[f].z);
        b = __int_as_float(fixList[f].w);
    }
    const uint8_t* center = bl + 18 * DESC_BP + 18;
    unsigned long long word[4];
    bool frag = false;
    // A tap is fragile when its pre-rounding coordinate v = fl(fl(x*b) + fl(y*a)) is within FR of a
    // half-integer.  With a, b each off by at most 1 ulp (2^-24, |a|,|b| <= 1) and |x|,|y| <= 13,
    // |v| < 18.4:  |dv| <= (|x|+|y|) 2^-24 + ulp(x*b) + ulp(y*a) + ulp(v)
    //                   <= 1.55e-6 + 2 * 9.5e-7 + 1.9e-6 = 5.4e-6  ->  FR = 6e-6.
    const float FR = 6e-6f;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float4 pt = patternF[q * 64 + lane]; // (x0, y0, x1, y1) of test bit q*64+lane
        // (single v_mul_f32: the compiler would pair these into v_pk_mul_f32, which issues at a third of the rate)
        const float fy0 = __fadd_rn(fmul_single(pt.x, b), fmul_single(pt.y, a));
        const float fx0 = __fsub_rn(fmul_single(pt.x, a), fmul_single(pt.y, b));
        const float fy1 = __fadd_rn(fmul_single(pt.z, b), fmul_single(pt.w, a));
        const float fx1 = __fsub_rn(fmul_single(pt.z, a), fmul_single(pt.w, b));
        // cvRound (:113-118) = round-half-even: v_rndne_f32, reused by the fragility test below
        const float ry0 = rintf(fy0), rx0 = rintf(fx0), ry1 = rintf(fy1), rx1 = rintf(fx1);
        // byte offset iy * 40 + ix in float (small integers: exact), one conversion per tap
        const int v0 = center[(int)fma_single(ry0, (float)DESC_BP, rx0)];
        const int v1 = center[(int)fma_single(ry1, (float)DESC_BP, rx1)];
        word[q] = __ballot(v0 < v1);
        if (MODE == 2) { // (compile-time: only the host-check instantiation carries the test)
            // |f - round(f)| > 0.5 - FR  <=>  f is within FR of a half-integer
            const float TH = 0.5f - FR;
            frag |= fmaxf(fmaxf(fabsf(fy0 - ry0), fabsf(fx0 - rx0)), fmaxf(fabsf(fy1 - ry1), fabsf(fx1 - rx1))) > TH;
        }
    }
    const size_t slot = (size_t)img * capPerImg + w.dest;
    if (lane < 4) {
        const unsigned long long v = lane == 0 ? word[0] : lane == 1 ? word[1] : lane == 2 ? word[2] : word[3];
        reinterpret_cast<unsigned long long*>(descOut + slot * 32)[lane] = v;
        // latency path of a frame or two: the caller's copy is written from here (posted writes over PCIe to pinned memory,
        // same layout as the device arrays) instead of by a download command after the kernel
        if (mirrorDesc) reinterpret_cast<unsigned long long*>(mirrorDesc + slot * 32)[lane] = v;
    }
    if (MODE != 1) {
        const bool anyFrag = MODE == 2 && __ballot(frag) != 0ull;
        if (lane == 0) {
            // the 28-byte cv::KeyPoint record (:1100-1147): pt scaled to level 0 (the scale of level 0 is 1), size of the
            // level, the angle, the FAST response, the octave, class_id -1
            struct __attribute__((packed, aligned(4))) Rec {
                float x, y, size, angle, response;
                int32_t octave, classId;
            } r;
            r.x = __fmul_rn((float)w.x, __int_as_float(sv[4]));
            r.y = __fmul_rn((float)w.y, __int_as_float(sv[4]));
            r.size = __int_as_float(sv[5]);
            r.angle = angle;
            r.response = (float)(key >> 24);
            r.octave = level;
            r.classId = -1;
            *reinterpret_cast<Rec*>(kpsOut + slot * 7) = r;
            if (mirrorKps) *reinterpret_cast<Rec*>(mirrorKps + slot * 7) = r;
            if (MODE == 2 && anyFrag) {
                const int idx = atomicAdd(reinterpret_cast<int*>(fixList), 1);
                fixList[1 + idx] = make_int4((img << 16) | g, __float_as_int(angle), __float_as_int(a), __float_as_int(b));
            }
        }
    }
    }; // desc_one
    desc_one(g);
    if (MODE != 1 && ORBFE_DESC_KPW > 1) {
#pragma unroll
        for (int rep = 1; rep < ORBFE_DESC_KPW; rep++) {
            WAVE_SYNC();
            desc_one(g + rep);
        }
    }
}

// --------------------------------------------------------------- K-MIRROR
// Results of a frame or two from the device slab [meta | keypoints | descriptors] into its page-locked twin (the same layout),
// behind K-DESC on the same stream: the metadata block, then per image its n keypoint records and n descriptor rows, as dwords
// across all lanes (256 contiguous bytes per wavefront store: full-line writes over PCIe).  A few workgroups (one CU's stores
// cross the link at ~7 GB/s).  Completion word: every thread's stores have LANDED (system-scope release) before its workgroup
// counts itself; the workgroup that completes the count publishes the flag -- whatever CU or XCD it ran on, every other
// workgroup's data is in host memory by then.
#define ORBFE_MIRROR_WGS 8
__global__ __launch_bounds__(256) void k_mirror_out(const uint8_t* __restrict__ dMeta, const uint8_t* __restrict__ dKps,
                                                    const uint8_t* __restrict__ dDesc, uint8_t* __restrict__ mirror,
                                                    unsigned metaBytes, int nimg, int cap, const OrbDone done)
{
    const unsigned t = blockIdx.x * 256u + threadIdx.x, T = gridDim.x * 256u;
    const int32_t* const n = reinterpret_cast<const int32_t*>(dMeta);
    for (unsigned i = t; i < metaBytes / 4u; i += T) reinterpret_cast<uint32_t*>(mirror)[i] = reinterpret_cast<const uint32_t*>(dMeta)[i];
    uint8_t* const mK = mirror + metaBytes;
    uint8_t* const mD = mK + (size_t)nimg * cap * 28;
    for (int im = 0; im < nimg; im++) {
        int cnt = n[im];
        cnt = cnt < 0 ? 0 : (cnt > cap ? cap : cnt); // (never a copy length out of a bad count: the host refuses such a count too)
        // keypoint records: 28 bytes each, the image's block starts at a multiple of 4 bytes -> dwords
        const uint32_t* sk = reinterpret_cast<const uint32_t*>(dKps + (size_t)im * cap * 28);
        uint32_t* dk = reinterpret_cast<uint32_t*>(mK + (size_t)im * cap * 28);
        for (unsigned i = t; i < (unsigned)cnt * 7u; i += T) dk[i] = sk[i];
        // descriptor rows: 32 bytes each (the block behind nimg * cap * 28 bytes of records is 4-byte aligned for any cap)
        const uint32_t* sd = reinterpret_cast<const uint32_t*>(dDesc + (size_t)im * cap * 32);
        uint32_t* dd = reinterpret_cast<uint32_t*>(mD + (size_t)im * cap * 32);
        for (unsigned i = t; i < (unsigned)cnt * 8u; i += T) dd[i] = sd[i];
    }
    if (!done.flag) return; // (uniform)
    __threadfence_system(); // this thread's stores have landed in host memory
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(done.ctr, 1u) + 1u == gridDim.x) {
            *done.ctr = 0u; // for the next call (calls on one stream are ordered)
            __threadfence_system();
            *(volatile unsigned*)done.flag = done.seq;
        }
    }
}

// --------------------------------------------------------------- K-STEREO
// Frame::ComputeStereoMatches (reference src/Frame.cc:797-967), rectified stereo: one wavefront per
// left keypoint.  The row table of the reference (vRowIndices) only restricts and orders candidates
// (ascending right index inside a row), and the scan keeps the first minimum, so the best candidate
// is argmin (distance, right index) over ALL right keypoints whose row band covers (int)vL -- a
// brute-force pass across the lanes.  The 11x11 SAD refinement reads both image pyramids where the
// extractors left them in HBM (no mvImagePyramid download).  The median-based outlier cut
// (:952-966) needs all matches and stays O(N) host code.
#define STEREO_CHUNK 2048 /* right keypoints staged per round: 8 + 8 + 16 KB of LDS */
#ifdef ORBFE_STEREO_TIMING // tuning only (tools/ab_build.sh stt "-DORBFE_STEREO_TIMING"): stage times, summed over the live wavefronts
__device__ unsigned long long g_stereoTimes[8]; // [1..5] sums of (stage end - wavefront start) in 100-MHz ticks, [7] wavefronts
#define ST_BEGIN() unsigned long long stS[6] = {(unsigned long long)wall_clock64(), 0, 0, 0, 0, 0}
#define ST(k) stS[k] = (unsigned long long)wall_clock64()
#define ST_END()                                                                                  \
    do {                                                                                          \
        if (lane == 0 && live) {                                                                  \
            unsigned long long prev_ = stS[0];                                                    \
            for (int k_ = 1; k_ <= 5; k_++) {                                                     \
                if (stS[k_]) prev_ = stS[k_];                                                     \
                atomicAdd(&g_stereoTimes[k_], prev_ - stS[0]);                                    \
            }                                                                                     \
            atomicAdd(&g_stereoTimes[7], 1ull);                                                   \
        }                                                                                         \
    } while (0)
#else
#define ST_BEGIN() do { } while (0)
#define ST(k) do { } while (0)
#define ST_END() do { } while (0)
#endif
__global__ __launch_bounds__(256) void k_stereo_match(const uint8_t* __restrict__ pyrL,
                                                      const uint8_t* __restrict__ pyrR,
                                                      const OrbLevelGeom* __restrict__ lg, int nlevels,
                                                      const float* __restrict__ kpsL, const uint8_t* __restrict__ descL,
                                                      int nL, const float* __restrict__ kpsR,
                                                      const uint8_t* __restrict__ descR, int nR, float mb, float mbf,
                                                      float* __restrict__ uRight, float* __restrict__ depth,
                                                      int32_t* __restrict__ sadOut,
                                                      const int32_t* __restrict__ nLdev /* counts still on the device */,
                                                      const int32_t* __restrict__ nRdev /* (resident form), or NULL   */,
                                                      const OrbDone done = OrbDone{nullptr, nullptr, 0u, 0u})
{
    __shared__ unsigned wgCnt;
    __shared__ float sScale[ORBFE_MAX_LEVELS];
    __shared__ float sU[STEREO_CHUNK];
    __shared__ uint32_t sBand[STEREO_CHUNK];
    __shared__ uint16_t sCand[4][STEREO_CHUNK];
    __shared__ uint32_t sWinR[4][66], sWinL[4][36]; // the SAD windows of a wavefront's keypoint: 11 rows x 6 / x 3 dwords
    wg_done_begin(done, &wgCnt);
    ST_BEGIN();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int iL = blockIdx.x * 4 + wave;
    const int rowsL = nL; // rows of the output arrays
    if (nLdev) nL = min(nL, nLdev[0]);
    if (nRdev) nR = min(nR, nRdev[0]);
    if (blockIdx.x * 4 >= nL) { // (uniform over the workgroup) rows past the keypoint count: "no match", so that a caller that
        if (iL < rowsL && lane == 0) { // asks for more rows never reads stale data
            uRight[iL] = -1.0f;
            depth[iL] = -1.0f;
            sadOut[iL] = -1;
        }
        wg_done(done, &wgCnt);
        return;
    }
    if (threadIdx.x < ORBFE_MAX_LEVELS) sScale[threadIdx.x] = (int)threadIdx.x < nlevels ? lg[threadIdx.x].scale : 0.f;
    // (a workgroup that straddles the count: its wavefronts past it take part in the staging and the barriers with a left
    // keypoint that matches nothing, and write "no match")
    const bool live = iL < nL;
    if (!live) {
        if (iL < rowsL && lane == 0) {
            uRight[iL] = -1.0f;
            depth[iL] = -1.0f;
            sadOut[iL] = -1;
        }
    }
    const int iLs = live ? iL : 0; // (what a wavefront past the count reads; it writes nothing more)
    const float uL = kpsL[iLs * 7 + 0], vL = kpsL[iLs * 7 + 1];
    const int levelL = reinterpret_cast<const int32_t*>(kpsL)[iLs * 7 + 5];
    const int vLi = (int)vL;
    const float maxD = __fdiv_rn(mbf, mb); // mbf / minZ, minZ = mb (:825-827)
    const float minU = __fsub_rn(uL, maxD), maxU = uL;
    const OrbLevelGeom G = lg[(levelL >= 0 && levelL < nlevels) ? levelL : 0]; // (for the SAD search: requested now, needed after the scan)
    const unsigned long long* dl = reinterpret_cast<const unsigned long long*>(descL + (size_t)iLs * 32);
    const unsigned long long l0 = dl[0], l1 = dl[1], l2 = dl[2], l3 = dl[3];
    unsigned best = 0xFFFFFFFFu;
    // Round 4: the scan over the right keypoints in two phases.  It used to load a right keypoint's x, y and octave (three
    // strided loads from the 28-byte records) and its level's scale inside the loop, with the descriptor load behind the row /
    // level / disparity tests: ~19 rounds of dependent global round trips per wavefront.  Now (a) the workgroup stages the right
    // keypoints ONCE per chunk as (uR, row band | octave) in LDS -- coalesced loads, the band computed as the reference does
    // (:812-822) --, (b) every wavefront scans the table from LDS and collects the indices that pass the three tests, and (c)
    // loads the descriptors of those (a dozen) with all loads in flight.
    for (int base = 0; base < nR; base += STEREO_CHUNK) { // (uniform over the workgroup)
        const int cn = min(STEREO_CHUNK, nR - base);
        __syncthreads(); // (the previous chunk's table and lists are no longer read)
        for (int j = threadIdx.x; j < cn; j += 256) {
            const int iR = base + j;
            const float uR = kpsR[iR * 7 + 0], kpY = kpsR[iR * 7 + 1];
            const int octR = reinterpret_cast<const int32_t*>(kpsR)[iR * 7 + 5];
            uint32_t band = 0xFFFFFFFFu; // (an octave outside the table: never a candidate)
            if (octR >= 0 && octR < nlevels) {
                const float r = __fmul_rn(2.0f, sScale[octR]);
                const int maxr = (int)ceilf(__fadd_rn(kpY, r)), minr = (int)floorf(__fsub_rn(kpY, r));
                // rows fit 13 bits each (image side <= 4096; minr may be -1..: clamp keeps the comparison with vLi >= 0 intact)
                band = (uint32_t)max(minr, 0) | ((uint32_t)min(max(maxr, -1) + 1, 8191) << 13) | ((uint32_t)octR << 26);
            }
            sU[j] = uR;
            sBand[j] = band;
        }
        __syncthreads();
        ST(1); // table staged
        int nc = 0;
        for (int j0 = 0; j0 < cn; j0 += 256) { // (uniform) four rows of 64 entries per step: their LDS reads are in flight together
            uint32_t bb[4];
            float uu[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int j = j0 + 64 * t + lane;
                bb[t] = j < cn ? sBand[j] : 0xFFFFFFFFu;
                uu[t] = j < cn ? sU[j] : 0.f;
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t b = bb[t];
                const int minr = (int)(b & 8191u), maxr1 = (int)((b >> 13) & 8191u), octR = (int)(b >> 26);
                const bool ok = live && b != 0xFFFFFFFFu && vLi >= minr && vLi < maxr1 && !(octR < levelL - 1 || octR > levelL + 1) &&
                                (uu[t] >= minU && uu[t] <= maxU);
                const unsigned long long m = __ballot(ok);
                if (ok) sCand[wave][nc + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(j0 + 64 * t + lane);
                nc += __popcll(m);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        ST(2); // table scanned
        for (int k0 = 0; k0 < nc; k0 += 64) {
            const int k = k0 + lane;
            if (k < nc) {
                const int iR = base + (int)sCand[wave][k];
                const unsigned long long* dr = reinterpret_cast<const unsigned long long*>(descR + (size_t)iR * 32);
                const int dist = __popcll(l0 ^ dr[0]) + __popcll(l1 ^ dr[1]) + __popcll(l2 ^ dr[2]) + __popcll(l3 ^ dr[3]);
                if (dist < 100) best = min(best, ((unsigned)dist << 20) | (unsigned)iR); // TH_HIGH
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, off));
    ST(3); // candidates scored, best found
    float outU = -1.0f, outD = -1.0f;
    int outS = -1;
    const int bestDist = best == 0xFFFFFFFFu ? 100 : (int)(best >> 20);
    if (bestDist < 75 && levelL >= 0 && levelL < nlevels) { // thOrbDist = (TH_HIGH+TH_LOW)/2
        const int bestIdxR = (int)(best & 0xFFFFF);
        const float uR0 = kpsR[bestIdxR * 7 + 0];
        const float sf = __fdiv_rn(1.0f, G.scale); // mvInvScaleFactors[octave]
        const float scaleduL = roundf(__fmul_rn(uL, sf)), scaledvL = roundf(__fmul_rn(vL, sf));
        const float scaleduR0 = roundf(__fmul_rn(uR0, sf));
        const int w = 5, Lw = 5;
        const float iniu = scaleduR0 + Lw - w, endu = scaleduR0 + Lw + w + 1;
        const int su = (int)scaleduL, sv = (int)scaledvL, sr = (int)scaleduR0;
        const bool inside = su - w >= 0 && su + w < G.w && sv - w >= 0 && sv + w < G.h && sr - Lw - w >= 0;
        if (!(iniu < 0 || endu >= (float)G.w) && inside) {
            const uint8_t* PL = pyrL + G.roiOff;
            const uint8_t* PR = pyrR + G.roiOff;
            // Round 4: the two windows (left 11 x 11, right 11 x 21) come into LDS as dwords first -- 99 loads, all in flight
            // together -- and the 121 x 11 absolute differences read bytes from there.  (One byte load per pixel and shift,
            // twelve per lane and round, from the pyramids in global memory: 8.5 of this wavefront's 15.5 us.)
            {
                const uint8_t* const rowL = PL + (size_t)(sv - w) * G.pitch + (su - w);
                const uint8_t* const rowR = PR + (size_t)(sv - w) * G.pitch + (sr - Lw - w);
                uint32_t v0 = 0, v1 = 0;
                { // item i < 66: right window, row i / 6, dword i % 6; item 66 + j: left window, row j / 3, dword j % 3
                    const int r = lane / 6, c = lane - 6 * r; // (lanes 0..63: right items 0..63)
                    __builtin_memcpy(&v0, rowR + (size_t)r * G.pitch + 4 * c, 4);
                    const int i = 64 + lane;
                    if (i < 66) {
                        __builtin_memcpy(&v1, rowR + (size_t)10 * G.pitch + 4 * (i - 60), 4);
                    } else if (i < 99) {
                        const int j = i - 66, rl = j / 3, cl = j - 3 * rl;
                        __builtin_memcpy(&v1, rowL + (size_t)rl * G.pitch + 4 * cl, 4);
                    }
                    sWinR[wave][lane] = v0;
                    if (i < 66) sWinR[wave][i] = v1;
                    else if (i < 99) sWinL[wave][i - 66] = v1;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            const uint8_t* const wL = reinterpret_cast<const uint8_t*>(sWinL[wave]); // pitch 12
            const uint8_t* const wR = reinterpret_cast<const uint8_t*>(sWinR[wave]); // pitch 24
            int sad[11];
#pragma unroll
            for (int k = 0; k < 11; k++) sad[k] = 0;
            for (int p = lane; p < 121; p += 64) {
                const int dyi = p / 11, dxi = p - dyi * 11;
                const int vl = wL[dyi * 12 + dxi];
                const uint8_t* rr = wR + dyi * 24 + dxi; // column sr + dx + (k - Lw) of the level = window column dxi + k
#pragma unroll
                for (int k = 0; k < 11; k++) sad[k] += abs(vl - (int)rr[k]);
            }
#pragma unroll
            for (int k = 0; k < 11; k++) sad[k] = wave_sum_i32(sad[k]); // (DPP + v_readlane: no LDS round trips)
            ST(4); // SAD windows loaded and summed
            int bestS = 0x7fffffff, bestinc = 0;
#pragma unroll
            for (int k = 0; k < 11; k++)
                if (sad[k] < bestS) {
                    bestS = sad[k];
                    bestinc = k - Lw;
                }
            if (!(bestinc == -Lw || bestinc == Lw)) {
                float d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
                for (int k = 1; k < 10; k++)
                    if (k - Lw == bestinc) {
                        d1 = (float)sad[k - 1];
                        d2 = (float)sad[k];
                        d3 = (float)sad[k + 1];
                    }
                const float deltaR = __fdiv_rn(__fsub_rn(d1, d3),
                                               __fmul_rn(2.0f, __fsub_rn(__fadd_rn(d1, d3), __fmul_rn(2.0f, d2))));
                if (!(deltaR < -1 || deltaR > 1)) {
                    float bestuR = __fmul_rn(G.scale, __fadd_rn(__fadd_rn(scaleduR0, (float)bestinc), deltaR));
                    float disparity = __fsub_rn(uL, bestuR);
                    if (disparity >= 0 && disparity < maxD) {
                        if (disparity <= 0) {
                            disparity = (float)0.01;
                            bestuR = (float)((double)uL - 0.01);
                        }
                        outD = __fdiv_rn(mbf, disparity);
                        outU = bestuR;
                        outS = bestS;
                    }
                }
            }
        }
    }
    if (lane == 0 && live) {
        uRight[iL] = outU;
        depth[iL] = outD;
        sadOut[iL] = outS;
    }
    ST(5); // results stored
    wg_done(done, &wgCnt);
    ST_END();
}
