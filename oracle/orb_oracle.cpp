/*
 * orb_oracle.cpp -- CPU oracle for the ORB front-end (TEST INFRASTRUCTURE, not product).
 *
 * A restatement, in dependency-free C++17, of
 *   - ORB_SLAM3::ORBextractor            reference src/ORBextractor.cc, include/ORBextractor.h
 *   - the Hamming brute-force searches   reference src/ORBmatcher.cc:269-471, 823-963, 1208-1449, 2545-2607
 *   - Frame::ComputeStereoFishEyeMatches' knn-2 brute force   reference src/Frame.cc:1119-1159
 *   - KannalaBrandt8::unproject          reference src/CameraModels/KannalaBrandt8.cpp:96-123
 * and of the OpenCV primitives they call (cv::resize, cv::copyMakeBorder, cv::FAST,
 * cv::GaussianBlur, cv::fastAtan2, cvRound, cv::BFMatcher), whose arithmetic lives in
 * OpenCV ("4.0, else >= 3.0", reference CMakeLists.txt:36-42) -- an un-vendored
 * dependency that is absent from this image.  Their published algorithms are restated
 * here as SURVEY.md Appendix B describes them.
 *
 * PARITY UNPINNED: the reference holds no tests, fixtures or golden vectors for this
 * path, and neither the reference nor OpenCV can be built here.  The oracle is pinned
 * by closed-form known-answer tests (tests/test_oracle_*.py) only.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this
 * library.  Build: make -C oracle   (g++ -O2 -ffp-contract=off; no FMA contraction so
 * float expressions round exactly as written).
 */
#include "orb_oracle.h"

#include <algorithm>
#include <chrono>
#include <thread>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <list>
#include <map>
#include <utility>
#include <vector>

#include "orb_pattern.inc"
#include "orb_sincos_cr.h"

namespace {

typedef orb_oracle_kp KP;

const int PATCH_SIZE = 31;      // reference src/ORBextractor.cc:70
const int HALF_PATCH_SIZE = 15; // :71
const int EDGE_THRESHOLD = 19;  // :72

// cvRound: round-half-to-even (SSE cvtss2si / cvtsd2si), SURVEY.md B.6
inline int cv_round(float v) { return (int)lrintf(v); }
inline int cv_round(double v) { return (int)lrint(v); }
inline int cv_floor(double v)
{
    int i = (int)v;
    return i - (i > v);
}
inline int cv_ceil(double v)
{
    int i = (int)v;
    return i + (i < v);
}
inline int16_t sat_s16(int v) { return (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
inline int reflect101(int i, int n)
{
    // cv::borderInterpolate(BORDER_REFLECT_101), SURVEY.md B.2
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * n - 2 - i;
    }
    return i;
}

// ---------------------------------------------------------------- cv::resize
// INTER_LINEAR, 8UC1, generic fixed-point path.
void resize_linear(const uint8_t* src, int sh, int sw, size_t sstride, uint8_t* dst, int dh, int dw, size_t dstride)
{
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    static thread_local std::vector<int> xofs, yofs, H0, H1;
    static thread_local std::vector<int16_t> ialpha, ibeta;
    xofs.resize(dw);
    yofs.resize(dh);
    ialpha.resize(2 * dw);
    ibeta.resize(2 * dh);
    H0.resize(dw);
    H1.resize(dw);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx] = sat_s16(cv_round((1.f - fx) * 2048));
        ialpha[2 * dx + 1] = sat_s16(cv_round(fx * 2048));
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[2 * dy] = sat_s16(cv_round((1.f - fy) * 2048));
        ibeta[2 * dy + 1] = sat_s16(cv_round(fy * 2048));
    }
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = std::min(std::max(yofs[dy], 0), sh - 1);
        int sy1 = std::min(std::max(yofs[dy] + 1, 0), sh - 1);
        const uint8_t* S0 = src + (size_t)sy0 * sstride;
        const uint8_t* S1 = src + (size_t)sy1 * sstride;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx], sx1 = std::min(sx + 1, sw - 1);
            int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
            H0[dx] = S0[sx] * a0 + S0[sx1] * a1;
            H1[dx] = S1[sx] * a0 + S1[sx1] * a1;
        }
        int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++) {
            int v = (((b0 * (H0[dx] >> 4)) >> 16) + ((b1 * (H1[dx] >> 4)) >> 16) + 2) >> 2;
            D[dx] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

// cv::copyMakeBorder(REFLECT_101) in place: buf is the padded buffer (rows x cols incl. border);
// the interior [border, rows-border) x [border, cols-border) is the source.
void border_reflect101(uint8_t* buf, int rows, int cols, size_t stride, int border)
{
    const int h = rows - 2 * border, w = cols - 2 * border;
    for (int y = 0; y < rows; y++) {
        int sy = reflect101(y - border, h) + border;
        uint8_t* drow = buf + (size_t)y * stride;
        const uint8_t* srow = buf + (size_t)sy * stride;
        const bool interior_row = (y >= border && y < border + h);
        for (int x = 0; x < cols; x++) {
            if (interior_row && x >= border && x < border + w) continue;
            int sx = reflect101(x - border, w) + border;
            drow[x] = srow[sx];
        }
    }
}

// ------------------------------------------------------------------ cv::FAST
// TYPE_9_16 ring, SURVEY.md B.3.
const int RING_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
const int RING_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

inline bool has_run9(unsigned m16)
{
    unsigned x = m16 | (m16 << 16);
    x &= x >> 1; // runs >= 2
    x &= x >> 2; // >= 4
    x &= x >> 4; // >= 8
    x &= x >> 1; // >= 9
    return (x & 0xFFFFu) != 0;
}

inline bool fast_is_corner(const uint8_t* c, size_t stride, int t)
{
    const int v = c[0];
    {
        // every arc of 9 holds one pixel of each opposite pair (k, k+8): cheap rejection first,
        // as cv::FAST does with its threshold table
        const ptrdiff_t s = (ptrdiff_t)stride;
        const int hi = v + t, lo = v - t;
        const int r0 = c[3 * s], r8 = c[-3 * s];
        unsigned d = ((r0 > hi) | (r8 > hi) ? 1u : 0u) | ((r0 < lo) | (r8 < lo) ? 2u : 0u);
        if (!d) return false;
        const int r4 = c[3], r12 = c[-3];
        d &= ((r4 > hi) | (r12 > hi) ? 1u : 0u) | ((r4 < lo) | (r12 < lo) ? 2u : 0u);
        if (!d) return false;
        const int r2 = c[2 * s + 2], r10 = c[-2 * s - 2];
        d &= ((r2 > hi) | (r10 > hi) ? 1u : 0u) | ((r2 < lo) | (r10 < lo) ? 2u : 0u);
        if (!d) return false;
        const int r6 = c[-2 * s + 2], r14 = c[2 * s - 2];
        d &= ((r6 > hi) | (r14 > hi) ? 1u : 0u) | ((r6 < lo) | (r14 < lo) ? 2u : 0u);
        if (!d) return false;
    }
    unsigned bright = 0, dark = 0;
    for (int k = 0; k < 16; k++) {
        int r = c[(ptrdiff_t)RING_DY[k] * (ptrdiff_t)stride + RING_DX[k]];
        if (r > v + t) bright |= 1u << k;
        if (r < v - t) dark |= 1u << k;
    }
    return has_run9(bright) || has_run9(dark);
}

// closed form: largest t' for which the pixel is still a FAST-9 corner.
int fast_score_closed(const uint8_t* c, size_t stride)
{
    int d[25];
    const int v = c[0];
    for (int k = 0; k < 16; k++) d[k] = (int)c[(ptrdiff_t)RING_DY[k] * (ptrdiff_t)stride + RING_DX[k]] - v;
    for (int k = 16; k < 25; k++) d[k] = d[k - 16];
    int best = -1000;
    for (int k = 0; k < 16; k++) {
        int mn = 1000, mx = -1000;
        for (int j = 0; j < 9; j++) {
            mn = std::min(mn, d[k + j]);
            mx = std::max(mx, d[k + j]);
        }
        best = std::max(best, mn);  // bright arc: min(r - v)
        best = std::max(best, -mx); // dark arc:   min(v - r)
    }
    return best - 1;
}

// OpenCV's cornerScore<16> two-loop form (early-outs, seeded with the threshold); kept as a
// cross-check of the closed form (tests/test_oracle_fast.py).
int fast_score_2loop(const uint8_t* c, size_t stride, int threshold)
{
    const int N = 25;
    int d[N];
    const int v = c[0];
    for (int k = 0; k < N; k++) {
        int kk = k & 15;
        d[k] = v - (int)c[(ptrdiff_t)RING_DY[kk] * (ptrdiff_t)stride + RING_DX[kk]];
    }
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], d[k + 2]);
        a = std::min(a, d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, d[k + 4]);
        a = std::min(a, d[k + 5]);
        a = std::min(a, d[k + 6]);
        a = std::min(a, d[k + 7]);
        a = std::min(a, d[k + 8]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k + 1], d[k + 2]);
        b = std::max(b, d[k + 3]);
        b = std::max(b, d[k + 4]);
        b = std::max(b, d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, d[k + 6]);
        b = std::max(b, d[k + 7]);
        b = std::max(b, d[k + 8]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
    }
    return -b0 - 1;
}

// cv::FAST(img, kps, threshold, nonmaxSuppression).  Row-major output; size=7, angle=-1.
void fast_detect(const uint8_t* img, int rows, int cols, size_t stride, int threshold, bool nms, std::vector<KP>& out)
{
    out.clear();
    threshold = std::min(std::max(threshold, 0), 255);
    if (rows < 7 || cols < 7) return;
    static thread_local std::vector<int> score;
    static thread_local std::vector<uint8_t> corner;
    score.assign((size_t)rows * cols, 0);
    corner.assign((size_t)rows * cols, 0);
    for (int y = 3; y < rows - 3; y++)
        for (int x = 3; x < cols - 3; x++) {
            const uint8_t* c = img + (size_t)y * stride + x;
            if (fast_is_corner(c, stride, threshold)) {
                corner[(size_t)y * cols + x] = 1;
                if (nms) score[(size_t)y * cols + x] = fast_score_closed(c, stride);
            }
        }
    for (int y = 3; y < rows - 3; y++)
        for (int x = 3; x < cols - 3; x++) {
            if (!corner[(size_t)y * cols + x]) continue;
            int s = score[(size_t)y * cols + x];
            if (nms) {
                bool keep = true;
                for (int dy = -1; dy <= 1 && keep; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        if (!dx && !dy) continue;
                        if (!(s > score[(size_t)(y + dy) * cols + (x + dx)])) { keep = false; break; }
                    }
                if (!keep) continue;
            }
            KP k;
            k.x = (float)x;
            k.y = (float)y;
            k.size = 7.f;
            k.angle = -1.f;
            k.response = nms ? (float)s : 0.f;
            k.octave = 0;
            k.class_id = -1;
            out.push_back(k);
        }
}

// -------------------------------------------------------------- GaussianBlur
// 7x7, sigma 2, BORDER_REFLECT_101, 8UC1, OpenCV 4.x fixed-point path:
// taps in 8.8, horizontal pass exact in u16, vertical pass 16.16, round-to-nearest.
void gaussian_blur7(const uint8_t* src, int rows, int cols, size_t sstride, uint8_t* dst, size_t dstride,
                    const int* taps)
{
    // horizontal pass into H (u16, saturating like ufixedpoint16), rows padded by REFLECT_101
    // scratch is kept per thread: large per-call allocations go through mmap/munmap, which
    // serialises many-thread runs on the process-wide mm lock
    static thread_local std::vector<uint16_t> H;
    static thread_local std::vector<uint8_t> prow;
    H.resize((size_t)rows * cols);
    prow.resize((size_t)cols + 6);
    for (int y = 0; y < rows; y++) {
        const uint8_t* s = src + (size_t)y * sstride;
        for (int x = -3; x < cols + 3; x++) prow[x + 3] = s[reflect101(x, cols)];
        uint16_t* h = &H[(size_t)y * cols];
        for (int x = 0; x < cols; x++) {
            uint32_t acc = 0;
            for (int i = 0; i < 7; i++) acc += (uint32_t)taps[i] * prow[x + i];
            h[x] = (uint16_t)(acc > 65535u ? 65535u : acc);
        }
    }
    for (int y = 0; y < rows; y++) {
        const uint16_t* r[7];
        for (int j = 0; j < 7; j++) r[j] = &H[(size_t)reflect101(y + j - 3, rows) * cols];
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < cols; x++) {
            uint32_t acc = 0;
            for (int j = 0; j < 7; j++) acc += (uint32_t)taps[j] * r[j][x];
            const uint32_t v = (acc + 32768u) >> 16;
            d[x] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
}

// ---- timing-only fast path ("an honest CPU leg") ----------------------------------------------------------
// The functions above are the PARITY oracle: scalar, one pixel at a time.  OpenCV's own cv::FAST and GaussianBlur are SIMD
// code, so a CPU baseline timed on the scalar port flatters the GPU.  orb_oracle_set_fastpath(o, 1) swaps in the two below for
// the TIMED CPU leg of bench.py only: the same results (bench.py asserts equality with the scalar path on the timed frames
// before it times anything; tests/test_oracle_fastpath.py does the same on the CPU), computed the way a SIMD library computes
// them.  Never used by a parity test as the checker.
#if defined(__AVX2__)
#include <immintrin.h>
// 32 centre pixels at once through the "one pixel of each opposite pair" rejection test (what cv::FAST does with its vector
// path before it counts arcs); returns the bit mask of the pixels that survive it
static inline unsigned fast_prefilter32(const uint8_t* c, ptrdiff_t s, int t)
{
    const __m256i v = _mm256_loadu_si256((const __m256i*)c);
    const __m256i tt = _mm256_set1_epi8((char)t);
    const __m256i hi = _mm256_adds_epu8(v, tt), lo = _mm256_subs_epu8(v, tt), z = _mm256_setzero_si256();
    auto pair = [&](const uint8_t* a, const uint8_t* b, __m256i& B, __m256i& D) {
        const __m256i ra = _mm256_loadu_si256((const __m256i*)a), rb = _mm256_loadu_si256((const __m256i*)b);
        // r > hi  <=>  r -sat hi != 0;   r < lo  <=>  lo -sat r != 0
        const __m256i nb = _mm256_cmpeq_epi8(_mm256_or_si256(_mm256_subs_epu8(ra, hi), _mm256_subs_epu8(rb, hi)), z);
        const __m256i nd = _mm256_cmpeq_epi8(_mm256_or_si256(_mm256_subs_epu8(lo, ra), _mm256_subs_epu8(lo, rb)), z);
        B = _mm256_andnot_si256(nb, B);
        D = _mm256_andnot_si256(nd, D);
    };
    __m256i B = _mm256_set1_epi8((char)0xFF), D = B;
    pair(c + 3 * s, c - 3 * s, B, D);
    if (_mm256_testz_si256(_mm256_or_si256(B, D), _mm256_or_si256(B, D))) return 0u;
    pair(c + 3, c - 3, B, D);
    if (_mm256_testz_si256(_mm256_or_si256(B, D), _mm256_or_si256(B, D))) return 0u;
    pair(c + 2 * s + 2, c - 2 * s - 2, B, D);
    pair(c - 2 * s + 2, c + 2 * s - 2, B, D);
    return (unsigned)_mm256_movemask_epi8(_mm256_or_si256(B, D));
}
#define ORB_ORACLE_HAVE_SIMD 1
#else
#define ORB_ORACLE_HAVE_SIMD 0
#endif

// resize_linear with OpenCV's row reuse: the horizontal interpolation of a source row is computed once and serves every
// destination row that reads it (at scale 1.2 most source rows serve two); the vertical pass is a plain row loop the compiler
// vectorises.  The same fixed-point arithmetic per pixel, hence the same bytes.
void resize_linear_fastpath(const uint8_t* src, int sh, int sw, size_t sstride, uint8_t* dst, int dh, int dw, size_t dstride)
{
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    static thread_local std::vector<int> xofs, xofs1, Hb[2];
    static thread_local std::vector<int16_t> a0v, a1v;
    xofs.resize(dw);
    xofs1.resize(dw);
    a0v.resize(dw);
    a1v.resize(dw);
    Hb[0].resize(dw);
    Hb[1].resize(dw);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        xofs1[dx] = std::min(sx + 1, sw - 1);
        a0v[dx] = sat_s16(cv_round((1.f - fx) * 2048));
        a1v[dx] = sat_s16(cv_round(fx * 2048));
    }
    int tag[2] = {-1, -1};
    auto hrow = [&](int sy, int other) -> const int* {
        for (int k = 0; k < 2; k++)
            if (tag[k] == sy) return Hb[k].data();
        const int k = (tag[0] == other) ? 1 : 0; // not the buffer the other row of this step lives in
        const uint8_t* S = src + (size_t)sy * sstride;
        int* __restrict H = Hb[k].data();
        const int *__restrict x0 = xofs.data(), *__restrict x1 = xofs1.data();
        const int16_t *__restrict a0 = a0v.data(), *__restrict a1 = a1v.data();
        for (int dx = 0; dx < dw; dx++) H[dx] = S[x0[dx]] * a0[dx] + S[x1[dx]] * a1[dx];
        tag[k] = sy;
        return H;
    };
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        const int b0 = sat_s16(cv_round((1.f - fy) * 2048)), b1 = sat_s16(cv_round(fy * 2048));
        const int sy0 = std::min(std::max(sy, 0), sh - 1), sy1 = std::min(std::max(sy + 1, 0), sh - 1);
        const int* __restrict H0 = hrow(sy0, sy1);
        const int* __restrict H1 = hrow(sy1, sy0);
        uint8_t* __restrict D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++) {
            int v = (((b0 * (H0[dx] >> 4)) >> 16) + ((b1 * (H1[dx] >> 4)) >> 16) + 2) >> 2;
            D[dx] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

// fast_detect with the vector prefilter (scalar finish: exact arc test + score on the few survivors) and an NMS pass over the
// list of corners instead of over every pixel.  Same output as fast_detect, element for element.
void fast_detect_fastpath(const uint8_t* img, int rows, int cols, size_t stride, int threshold, bool nms, std::vector<KP>& out)
{
#if ORB_ORACLE_HAVE_SIMD
    out.clear();
    threshold = std::min(std::max(threshold, 0), 255);
    if (rows < 7 || cols < 7) return;
    if (cols - 6 < 32) return fast_detect(img, rows, cols, stride, threshold, nms, out); // narrower than one vector
    static thread_local std::vector<int> score;
    static thread_local std::vector<int> pos;
    score.assign((size_t)rows * cols, 0);
    pos.clear();
    const ptrdiff_t s = (ptrdiff_t)stride;
    for (int y = 3; y < rows - 3; y++) {
        const uint8_t* row = img + (size_t)y * stride;
        int x = 3;
        auto finish = [&](int x0, unsigned m, int from) { // survivors of one vector, ascending x; `from`: first x not yet done
            while (m) {
                const int b = __builtin_ctz(m);
                m &= m - 1;
                const int xx = x0 + b;
                if (xx < from) continue;
                const uint8_t* c = row + xx;
                if (fast_is_corner(c, stride, threshold)) {
                    pos.push_back(y * cols + xx);
                    if (nms) score[(size_t)y * cols + xx] = fast_score_closed(c, stride);
                }
            }
        };
        for (; x + 32 <= cols - 3; x += 32) finish(x, fast_prefilter32(row + x, s, threshold), x);
        if (x < cols - 3) { // the tail: one more vector that ends at the last interior pixel (overlaps the previous one)
            const int x0 = cols - 3 - 32;
            finish(x0, fast_prefilter32(row + x0, s, threshold), x);
        }
    }
    for (int p : pos) {
        const int y = p / cols, x = p - y * cols;
        const int sc = score[(size_t)p];
        if (nms) {
            const int* r0 = &score[(size_t)(y - 1) * cols + x];
            const int* r1 = &score[(size_t)y * cols + x];
            const int* r2 = &score[(size_t)(y + 1) * cols + x];
            if (!(sc > r0[-1] && sc > r0[0] && sc > r0[1] && sc > r1[-1] && sc > r1[1] && sc > r2[-1] && sc > r2[0] && sc > r2[1])) continue;
        }
        KP k;
        k.x = (float)x;
        k.y = (float)y;
        k.size = 7.f;
        k.angle = -1.f;
        k.response = nms ? (float)sc : 0.f;
        k.octave = 0;
        k.class_id = -1;
        out.push_back(k);
    }
#else
    fast_detect(img, rows, cols, stride, threshold, nms, out);
#endif
}

// gaussian_blur7 written as whole-row passes per tap, which the compiler vectorises (-O3 -march=native: vpmullw / vpaddw rows).
// The u16 accumulator of the horizontal pass is exact while the taps sum to <= 257 (255 * 257 = 65535: ufixedpoint16's
// saturation cannot trigger); other taps take the scalar function.
void gaussian_blur7_fastpath(const uint8_t* src, int rows, int cols, size_t sstride, uint8_t* dst, size_t dstride, const int* taps)
{
    int sum = 0;
    for (int i = 0; i < 7; i++) sum += taps[i];
    if (sum > 257) return gaussian_blur7(src, rows, cols, sstride, dst, dstride, taps);
    static thread_local std::vector<uint16_t> H;
    static thread_local std::vector<uint8_t> prow;
    static thread_local std::vector<uint32_t> acc;
    H.resize((size_t)rows * cols);
    prow.resize((size_t)cols + 6);
    acc.resize((size_t)cols);
    uint16_t t16[7];
    for (int i = 0; i < 7; i++) t16[i] = (uint16_t)taps[i];
    for (int y = 0; y < rows; y++) {
        const uint8_t* sr = src + (size_t)y * sstride;
        for (int x = -3; x < 0; x++) prow[x + 3] = sr[reflect101(x, cols)];
        memcpy(&prow[3], sr, (size_t)cols);
        for (int x = cols; x < cols + 3; x++) prow[x + 3] = sr[reflect101(x, cols)];
        uint16_t* __restrict h = &H[(size_t)y * cols];
        const uint8_t* __restrict p = prow.data();
        for (int x = 0; x < cols; x++)
            h[x] = (uint16_t)(t16[0] * p[x] + t16[1] * p[x + 1] + t16[2] * p[x + 2] + t16[3] * p[x + 3] + t16[4] * p[x + 4] +
                              t16[5] * p[x + 5] + t16[6] * p[x + 6]);
    }
    for (int y = 0; y < rows; y++) {
        const uint16_t* r[7];
        for (int j = 0; j < 7; j++) r[j] = &H[(size_t)reflect101(y + j - 3, rows) * cols];
        uint8_t* __restrict d = dst + (size_t)y * dstride;
        uint32_t* __restrict a = acc.data();
        const uint32_t t0 = (uint32_t)taps[0], t1 = (uint32_t)taps[1], t2 = (uint32_t)taps[2], t3 = (uint32_t)taps[3],
                       t4 = (uint32_t)taps[4], t5 = (uint32_t)taps[5], t6 = (uint32_t)taps[6];
        const uint16_t *r0 = r[0], *r1 = r[1], *r2 = r[2], *r3 = r[3], *r4 = r[4], *r5 = r[5], *r6 = r[6];
        for (int x = 0; x < cols; x++)
            a[x] = t0 * r0[x] + t1 * r1[x] + t2 * r2[x] + t3 * r3[x] + t4 * r4[x] + t5 * r5[x] + t6 * r6[x];
        for (int x = 0; x < cols; x++) {
            const uint32_t v = (a[x] + 32768u) >> 16;
            d[x] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
}

// ---------------------------------------------------------------- fastAtan2
// cv::fastAtan2 (degrees); single precision, no FMA.
// fma_horner: the three inner Horner steps fused (an OpenCV AVX2 build with -mfma)
float fast_atan2(float y, float x, bool fma_horner = false)
{
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16; // (float)DBL_EPSILON
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = fma_horner ? std::fmaf(std::fmaf(std::fmaf(p7, c2, p5), c2, p3), c2, p1) * c
                       : (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (fma_horner ? std::fmaf(std::fmaf(std::fmaf(p7, c2, p5), c2, p3), c2, p1) * c
                                : (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c);
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// ------------------------------------------------- ORBextractor restatement
struct Pt2i {
    int x, y;
};

// reference include/ORBextractor.h:30-41 (ExtractorNode) + src/ORBextractor.cc:479-535 (DivideNode)
struct Node {
    std::vector<KP> vKeys;
    Pt2i UL, UR, BL, BR;
    std::list<Node>::iterator lit;
    bool bNoMore = false;
    long seq = 0; // creation sequence number: canonical replacement for the heap address

    void Divide(Node& n1, Node& n2, Node& n3, Node& n4) const
    {
        const int halfX = (int)std::ceil(static_cast<float>(UR.x - UL.x) / 2);
        const int halfY = (int)std::ceil(static_cast<float>(BR.y - UL.y) / 2);
        n1.UL = UL;
        n1.UR = {UL.x + halfX, UL.y};
        n1.BL = {UL.x, UL.y + halfY};
        n1.BR = {UL.x + halfX, UL.y + halfY};
        n2.UL = n1.UR;
        n2.UR = UR;
        n2.BL = n1.BR;
        n2.BR = {UR.x, UL.y + halfY};
        n3.UL = n1.BL;
        n3.UR = n1.BR;
        n3.BL = BL;
        n3.BR = {n1.BR.x, BL.y};
        n4.UL = n3.UR;
        n4.UR = n2.BR;
        n4.BL = n3.BR;
        n4.BR = BR;
        for (size_t i = 0; i < vKeys.size(); i++) {
            const KP& kp = vKeys[i];
            if (kp.x < n1.UR.x) {
                if (kp.y < n1.BR.y) n1.vKeys.push_back(kp);
                else n3.vKeys.push_back(kp);
            } else if (kp.y < n1.BR.y)
                n2.vKeys.push_back(kp);
            else
                n4.vKeys.push_back(kp);
        }
        if (n1.vKeys.size() == 1) n1.bNoMore = true;
        if (n2.vKeys.size() == 1) n2.bNoMore = true;
        if (n3.vKeys.size() == 1) n3.bNoMore = true;
        if (n4.vKeys.size() == 1) n4.bNoMore = true;
    }
};

typedef std::pair<int, long> SizeSeq; // (size, seq) -- sort key of :682 with seq in place of the pointer

// reference src/ORBextractor.cc:537-761, literal list semantics.
std::vector<KP> DistributeOctTree(const std::vector<KP>& vToDistributeKeys, int minX, int maxX, int minY, int maxY,
                                  int N)
{
    std::vector<KP> vResultKeys;
    const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    if (nIni < 1) return vResultKeys; // reference divides by zero here (portrait ratio < 0.5)
    const float hX = static_cast<float>(maxX - minX) / nIni;

    std::list<Node> lNodes;
    std::vector<Node*> vpIniNodes(nIni);
    long seq = 0;
    for (int i = 0; i < nIni; i++) {
        Node ni;
        ni.UL = {(int)(hX * static_cast<float>(i)), 0};
        ni.UR = {(int)(hX * static_cast<float>(i + 1)), 0};
        ni.BL = {ni.UL.x, maxY - minY};
        ni.BR = {ni.UR.x, maxY - minY};
        ni.seq = seq++;
        lNodes.push_back(ni);
        vpIniNodes[i] = &lNodes.back();
    }
    for (size_t i = 0; i < vToDistributeKeys.size(); i++) {
        const KP& kp = vToDistributeKeys[i];
        int r = (int)(kp.x / hX);
        if (r >= nIni) r = nIni - 1; // cannot happen for x < maxX-minX; guards UB of the reference
        vpIniNodes[r]->vKeys.push_back(kp);
    }
    std::list<Node>::iterator lit = lNodes.begin();
    while (lit != lNodes.end()) {
        if (lit->vKeys.size() == 1) {
            lit->bNoMore = true;
            lit++;
        } else if (lit->vKeys.empty())
            lit = lNodes.erase(lit);
        else
            lit++;
    }

    bool bFinish = false;
    std::vector<std::pair<SizeSeq, Node*>> vSizeAndPointerToNode;

    auto push_children = [&](Node (&n)[4], int* nToExpand) {
        for (int c = 0; c < 4; c++) {
            if (n[c].vKeys.size() > 0) {
                n[c].seq = seq++;
                lNodes.push_front(n[c]);
                if (n[c].vKeys.size() > 1) {
                    if (nToExpand) (*nToExpand)++;
                    vSizeAndPointerToNode.push_back(
                        std::make_pair(SizeSeq((int)n[c].vKeys.size(), lNodes.front().seq), &lNodes.front()));
                    lNodes.front().lit = lNodes.begin();
                }
            }
        }
    };

    while (!bFinish) {
        int prevSize = (int)lNodes.size();
        lit = lNodes.begin();
        int nToExpand = 0;
        vSizeAndPointerToNode.clear();
        while (lit != lNodes.end()) {
            if (lit->bNoMore) {
                lit++;
                continue;
            }
            Node n[4];
            lit->Divide(n[0], n[1], n[2], n[3]);
            push_children(n, &nToExpand);
            lit = lNodes.erase(lit);
        }
        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
            bFinish = true;
        } else if (((int)lNodes.size() + nToExpand * 3) > N) {
            while (!bFinish) {
                prevSize = (int)lNodes.size();
                std::vector<std::pair<SizeSeq, Node*>> vPrev = vSizeAndPointerToNode;
                vSizeAndPointerToNode.clear();
                std::sort(vPrev.begin(), vPrev.end(),
                          [](const std::pair<SizeSeq, Node*>& a, const std::pair<SizeSeq, Node*>& b) {
                              return a.first < b.first;
                          });
                for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
                    Node n[4];
                    vPrev[j].second->Divide(n[0], n[1], n[2], n[3]);
                    push_children(n, nullptr);
                    lNodes.erase(vPrev[j].second->lit);
                    if ((int)lNodes.size() >= N) break;
                }
                if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
            }
        }
    }

    vResultKeys.reserve(lNodes.size());
    for (std::list<Node>::iterator it = lNodes.begin(); it != lNodes.end(); it++) {
        std::vector<KP>& vNodeKeys = it->vKeys;
        KP* pKP = &vNodeKeys[0];
        float maxResponse = pKP->response;
        for (size_t k = 1; k < vNodeKeys.size(); k++) {
            if (vNodeKeys[k].response > maxResponse) {
                pKP = &vNodeKeys[k];
                maxResponse = vNodeKeys[k].response;
            }
        }
        vResultKeys.push_back(*pKP);
    }
    return vResultKeys;
}

struct Level {
    int rows = 0, cols = 0;    // level size (without border)
    size_t stride = 0;         // of the padded buffer
    std::vector<uint8_t> buf;  // (rows+38) x (cols+38)
    std::vector<uint8_t> blur; // rows x cols (only when the level has keypoints)
    uint8_t* roi() { return buf.data() + (size_t)EDGE_THRESHOLD * stride + EDGE_THRESHOLD; }
};

} // namespace

struct orb_oracle {
    int nfeatures;
    double scaleFactor; // reference include/ORBextractor.h:96 -- a double initialised from a float
    int nlevels, iniThFAST, minThFAST;
    std::vector<int> mnFeaturesPerLevel, umax;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    int taps[7] = {18, 34, 48, 56, 48, 34, 18};
    int trig_mode = ORB_ORACLE_TRIG_LIBM;
    bool atan_fma = false;
    std::vector<Level> pyr;
    std::vector<std::vector<KP>> cands, allKeypoints;
    bool fastpath = false; // timing-only SIMD FAST + vector-friendly blur (orb_oracle_set_fastpath)
    double stage_s[6] = {0, 0, 0, 0, 0, 0}; // accumulated: pyramid, FAST (cell loop), quadtree, orientation, blur, descriptors
    long stage_calls = 0;
    static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    // reference src/ORBextractor.cc:408-468
    orb_oracle(int _nfeatures, float _scaleFactor, int _nlevels, int _ini, int _min)
        : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_ini), minThFAST(_min)
    {
        mvScaleFactor.resize(nlevels);
        mvLevelSigma2.resize(nlevels);
        mvScaleFactor[0] = 1.0f;
        mvLevelSigma2[0] = 1.0f;
        for (int i = 1; i < nlevels; i++) {
            mvScaleFactor[i] = (float)(mvScaleFactor[i - 1] * scaleFactor);
            mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
        }
        mvInvScaleFactor.resize(nlevels);
        mvInvLevelSigma2.resize(nlevels);
        for (int i = 0; i < nlevels; i++) {
            mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
            mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
        }
        mnFeaturesPerLevel.resize(nlevels);
        float factor = (float)(1.0f / scaleFactor);
        float nDesiredFeaturesPerScale =
            nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
        int sumFeatures = 0;
        for (int level = 0; level < nlevels - 1; level++) {
            mnFeaturesPerLevel[level] = cv_round(nDesiredFeaturesPerScale);
            sumFeatures += mnFeaturesPerLevel[level];
            nDesiredFeaturesPerScale *= factor;
        }
        mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);

        umax.resize(HALF_PATCH_SIZE + 1);
        int v, v0, vmax = cv_floor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
        int vmin = cv_ceil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
        const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
        for (v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt(hp2 - v * v));
        for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
        pyr.resize(nlevels);
    }

    // reference :1152-1177
    void ComputePyramid(const uint8_t* img, int rows, int cols, size_t stride)
    {
        for (int level = 0; level < nlevels; ++level) {
            float scale = mvInvScaleFactor[level];
            Level& L = pyr[level];
            L.cols = cv_round((float)cols * scale);
            L.rows = cv_round((float)rows * scale);
            L.stride = (size_t)L.cols + EDGE_THRESHOLD * 2;
            L.buf.assign((size_t)(L.rows + EDGE_THRESHOLD * 2) * L.stride, 0);
            L.blur.clear();
            if (level != 0) {
                Level& P = pyr[level - 1];
                if (fastpath) resize_linear_fastpath(P.roi(), P.rows, P.cols, P.stride, L.roi(), L.rows, L.cols, L.stride);
                else resize_linear(P.roi(), P.rows, P.cols, P.stride, L.roi(), L.rows, L.cols, L.stride);
            } else {
                for (int y = 0; y < rows; y++) memcpy(L.roi() + (size_t)y * L.stride, img + (size_t)y * stride, cols);
            }
            border_reflect101(L.buf.data(), L.rows + 2 * EDGE_THRESHOLD, L.cols + 2 * EDGE_THRESHOLD, L.stride,
                              EDGE_THRESHOLD);
        }
    }

    // reference :75-102
    float IC_Angle(Level& L, float ptx, float pty)
    {
        int m_01 = 0, m_10 = 0;
        const uint8_t* center = L.roi() + (size_t)cv_round(pty) * L.stride + cv_round(ptx);
        for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
        int step = (int)L.stride;
        for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
            int v_sum = 0;
            int d = umax[v];
            for (int u = -d; u <= d; ++u) {
                int val_plus = center[u + v * step], val_minus = center[u - v * step];
                v_sum += (val_plus - val_minus);
                m_10 += u * (val_plus + val_minus);
            }
            m_01 += v * v_sum;
        }
        return fast_atan2((float)m_01, (float)m_10, atan_fma);
    }

    // reference :763-878.  Returns false when a level is too small for the 35-px cell grid.
    bool ComputeKeyPointsOctTree()
    {
        allKeypoints.assign(nlevels, std::vector<KP>());
        cands.assign(nlevels, std::vector<KP>());
        const float W = 35;
        for (int level = 0; level < nlevels; ++level) {
            const double tLevel = now_s();
            Level& L = pyr[level];
            const int minBorderX = EDGE_THRESHOLD - 3;
            const int minBorderY = minBorderX;
            const int maxBorderX = L.cols - EDGE_THRESHOLD + 3;
            const int maxBorderY = L.rows - EDGE_THRESHOLD + 3;
            std::vector<KP>& vToDistributeKeys = cands[level];
            const float width = (float)(maxBorderX - minBorderX);
            const float height = (float)(maxBorderY - minBorderY);
            const int nCols = (int)(width / W);
            const int nRows = (int)(height / W);
            if (nCols < 1 || nRows < 1) return false; // reference divides by zero
            const int wCell = (int)std::ceil(width / nCols);
            const int hCell = (int)std::ceil(height / nRows);
            for (int i = 0; i < nRows; i++) {
                const float iniY = (float)(minBorderY + i * hCell);
                float maxY = iniY + hCell + 6;
                if (iniY >= maxBorderY - 3) continue;
                if (maxY > maxBorderY) maxY = (float)maxBorderY;
                for (int j = 0; j < nCols; j++) {
                    const float iniX = (float)(minBorderX + j * wCell);
                    float maxX = iniX + wCell + 6;
                    if (iniX >= maxBorderX - 6) continue;
                    if (maxX > maxBorderX) maxX = (float)maxBorderX;
                    std::vector<KP> vKeysCell;
                    const uint8_t* roi = L.roi() + (size_t)(int)iniY * L.stride + (int)iniX;
                    const int rr = (int)maxY - (int)iniY, cc = (int)maxX - (int)iniX;
                    if (fastpath) {
                        fast_detect_fastpath(roi, rr, cc, L.stride, iniThFAST, true, vKeysCell);
                        if (vKeysCell.empty()) fast_detect_fastpath(roi, rr, cc, L.stride, minThFAST, true, vKeysCell);
                    } else {
                        fast_detect(roi, rr, cc, L.stride, iniThFAST, true, vKeysCell);
                        if (vKeysCell.empty()) fast_detect(roi, rr, cc, L.stride, minThFAST, true, vKeysCell);
                    }
                    for (KP& k : vKeysCell) {
                        k.x += j * wCell;
                        k.y += i * hCell;
                        vToDistributeKeys.push_back(k);
                    }
                }
            }
            const double tq = now_s();
            stage_s[1] += tq - tLevel;
            std::vector<KP>& keypoints = allKeypoints[level];
            keypoints = DistributeOctTree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                          mnFeaturesPerLevel[level]);
            stage_s[2] += now_s() - tq;
            const int scaledPatchSize = (int)(PATCH_SIZE * mvScaleFactor[level]);
            for (KP& k : keypoints) {
                k.x += minBorderX;
                k.y += minBorderY;
                k.octave = level;
                k.size = (float)scaledPatchSize;
            }
        }
        const double ta = now_s();
        for (int level = 0; level < nlevels; ++level)
            for (KP& k : allKeypoints[level]) k.angle = IC_Angle(pyr[level], k.x, k.y);
        stage_s[3] += now_s() - ta;
        return true;
    }

    // reference :104-145
    void computeOrbDescriptor(const KP& kpt, const uint8_t* img, int step, uint8_t* desc)
    {
        const float factorPI = (float)(3.14159265358979323846 / 180.f);
        float angle = (float)kpt.angle * factorPI;
        float a, b;
        if (trig_mode == ORB_ORACLE_TRIG_LIBM) {
            a = cosf(angle);
            b = sinf(angle);
        } else {
            orb_sincos_cr_impl(angle, &b, &a);
        }
        const uint8_t* center = img + (size_t)cv_round(kpt.y) * step + cv_round(kpt.x);
        for (int i = 0; i < 32; ++i) {
            int val = 0;
            for (int j = 0; j < 8; j++) {
                const signed char* p = ORB_PATTERN_31[i * 8 + j];
                int t0 = center[cv_round(p[0] * b + p[1] * a) * step + cv_round(p[0] * a - p[1] * b)];
                int t1 = center[cv_round(p[2] * b + p[3] * a) * step + cv_round(p[2] * a - p[3] * b)];
                val |= (t0 < t1) << j;
            }
            desc[i] = (uint8_t)val;
        }
    }

    // reference :1068-1150
    int extract(const uint8_t* img, int rows, int cols, size_t stride, int lap0, int lap1, KP* kps, uint8_t* desc,
                int cap, int* n_out)
    {
        if (n_out) *n_out = 0;
        if (!img || rows <= 0 || cols <= 0) return -1;
        const double tp = now_s();
        ComputePyramid(img, rows, cols, stride);
        stage_s[0] += now_s() - tp;
        stage_calls++;
        if (!ComputeKeyPointsOctTree()) return -2;
        int nkeypoints = 0;
        for (int level = 0; level < nlevels; ++level) nkeypoints += (int)allKeypoints[level].size();
        if (nkeypoints > cap) return -2;
        if (n_out) *n_out = nkeypoints;
        int monoIndex = 0, stereoIndex = nkeypoints - 1;
        for (int level = 0; level < nlevels; ++level) {
            std::vector<KP>& keypoints = allKeypoints[level];
            if (keypoints.empty()) continue;
            Level& L = pyr[level];
            L.blur.resize((size_t)L.rows * L.cols);
            const double tb = now_s();
            if (fastpath) gaussian_blur7_fastpath(L.roi(), L.rows, L.cols, L.stride, L.blur.data(), (size_t)L.cols, taps);
            else gaussian_blur7(L.roi(), L.rows, L.cols, L.stride, L.blur.data(), (size_t)L.cols, taps);
            const double td = now_s();
            stage_s[4] += td - tb;
            float scale = mvScaleFactor[level];
            for (KP& kp0 : keypoints) {
                uint8_t d[32];
                computeOrbDescriptor(kp0, L.blur.data(), L.cols, d);
                KP kp = kp0;
                if (level != 0) {
                    kp.x *= scale;
                    kp.y *= scale;
                }
                int dst;
                if (kp.x >= lap0 && kp.x <= lap1) dst = stereoIndex--;
                else dst = monoIndex++;
                kps[dst] = kp;
                memcpy(desc + (size_t)dst * 32, d, 32);
            }
            stage_s[5] += now_s() - td;
        }
        return monoIndex;
    }
};

// ------------------------------------------------------------------ matcher
namespace {

const int TH_LOW = 50;       // reference src/ORBmatcher.cc:37
const int HISTO_LENGTH = 30; // :38

// reference src/ORBmatcher.cc:2591-2607
int DescriptorDistance(const uint8_t* a, const uint8_t* b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        unsigned int v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

// reference :2545-2586 (on bin sizes)
void ComputeThreeMaxima(const int* histo, int L, int& ind1, int& ind2, int& ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        if (s > max1) {
            max3 = max2;
            max2 = max1;
            max1 = s;
            ind3 = ind2;
            ind2 = ind1;
            ind1 = i;
        } else if (s > max2) {
            max3 = max2;
            max2 = s;
            ind3 = ind2;
            ind2 = i;
        } else if (s > max3) {
            max3 = s;
            ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) {
        ind2 = -1;
        ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
        ind3 = -1;
    }
}

inline int rot_bin(float a1, float a2)
{
    const float factor = 1.0f / HISTO_LENGTH; // sic: 1/30, reference :282,394
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)std::round(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

// merge-join of two CSR feature vectors by node id (std::map walk + lower_bound, :285-448)
template <class F>
void for_each_shared_node(const orb_oracle_fv* a, const orb_oracle_fv* b, F f)
{
    int i = 0, j = 0;
    while (i < a->nn && j < b->nn) {
        if (a->node_ids[i] == b->node_ids[j]) {
            f(i, j);
            i++;
            j++;
        } else if (a->node_ids[i] < b->node_ids[j]) {
            while (i < a->nn && a->node_ids[i] < b->node_ids[j]) i++;
        } else {
            while (j < b->nn && b->node_ids[j] < a->node_ids[i]) j++;
        }
    }
}

int cull_rotation(std::vector<int>* rotHist, int32_t* match, int nmatches)
{
    int counts[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) counts[i] = (int)rotHist[i].size();
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(counts, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
        if (i == ind1 || i == ind2 || i == ind3) continue;
        for (size_t j = 0; j < rotHist[i].size(); j++) {
            match[rotHist[i][j]] = -1;
            nmatches--;
        }
    }
    return nmatches;
}

} // namespace

extern "C" {

orb_oracle* orb_oracle_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
{
    if (nfeatures < 0 || nlevels < 1 || nlevels > 32 || !(scaleFactor > 1.0f)) return nullptr;
    return new orb_oracle(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST);
}
void orb_oracle_destroy(orb_oracle* o) { delete o; }
void orb_oracle_set_gauss_taps(orb_oracle* o, const int* t)
{
    for (int i = 0; i < 7; i++) o->taps[i] = t[i];
}
void orb_oracle_set_trig_mode(orb_oracle* o, int mode) { o->trig_mode = mode; }
int orb_oracle_set_fastpath(orb_oracle* o, int on)
{
    o->fastpath = on != 0;
    return ORB_ORACLE_HAVE_SIMD; // 1: the FAST prefilter is AVX2 code in this build; 0: only the blur differs
}
void orb_oracle_get_stage_seconds(orb_oracle* o, double* s6, long* calls, int reset)
{
    for (int i = 0; i < 6; i++) s6[i] = o->stage_s[i];
    if (calls) *calls = o->stage_calls;
    if (reset) {
        for (int i = 0; i < 6; i++) o->stage_s[i] = 0;
        o->stage_calls = 0;
    }
}
void orb_oracle_set_atan_fma(orb_oracle* o, int on) { o->atan_fma = on != 0; }

int orb_oracle_extract(orb_oracle* o, const uint8_t* img, int rows, int cols, size_t stride, int lap0, int lap1,
                       orb_oracle_kp* kps, uint8_t* desc, int cap, int* n_out)
{
    return o->extract(img, rows, cols, stride, lap0, lap1, kps, desc, cap, n_out);
}

void orb_oracle_get_scale_tables(orb_oracle* o, float* sf, float* inv, float* s2, float* is2)
{
    for (int i = 0; i < o->nlevels; i++) {
        if (sf) sf[i] = o->mvScaleFactor[i];
        if (inv) inv[i] = o->mvInvScaleFactor[i];
        if (s2) s2[i] = o->mvLevelSigma2[i];
        if (is2) is2[i] = o->mvInvLevelSigma2[i];
    }
}
void orb_oracle_get_features_per_level(orb_oracle* o, int* n)
{
    for (int i = 0; i < o->nlevels; i++) n[i] = o->mnFeaturesPerLevel[i];
}
void orb_oracle_get_umax(orb_oracle* o, int* u)
{
    for (int i = 0; i < 16; i++) u[i] = o->umax[i];
}

int orb_oracle_get_level(orb_oracle* o, int level, const uint8_t** data, int* rows, int* cols, size_t* stride)
{
    if (level < 0 || level >= o->nlevels || o->pyr[level].buf.empty()) return -1;
    Level& L = o->pyr[level];
    *data = L.buf.data();
    *rows = L.rows + 2 * EDGE_THRESHOLD;
    *cols = L.cols + 2 * EDGE_THRESHOLD;
    *stride = L.stride;
    return 0;
}
int orb_oracle_get_blurred(orb_oracle* o, int level, const uint8_t** data, int* rows, int* cols, size_t* stride)
{
    if (level < 0 || level >= o->nlevels || o->pyr[level].blur.empty()) return -1;
    Level& L = o->pyr[level];
    *data = L.blur.data();
    *rows = L.rows;
    *cols = L.cols;
    *stride = (size_t)L.cols;
    return 0;
}
int orb_oracle_get_candidates(orb_oracle* o, int level, const orb_oracle_kp** kps)
{
    if (level < 0 || level >= (int)o->cands.size()) return -1;
    *kps = o->cands[level].data();
    return (int)o->cands[level].size();
}
int orb_oracle_get_level_keypoints(orb_oracle* o, int level, const orb_oracle_kp** kps)
{
    if (level < 0 || level >= (int)o->allKeypoints.size()) return -1;
    *kps = o->allKeypoints[level].data();
    return (int)o->allKeypoints[level].size();
}

void orb_oracle_resize_linear(const uint8_t* src, int sh, int sw, size_t sstride, uint8_t* dst, int dh, int dw,
                              size_t dstride)
{
    resize_linear(src, sh, sw, sstride, dst, dh, dw, dstride);
}
void orb_oracle_border_reflect101(uint8_t* buf, int rows, int cols, size_t stride, int border)
{
    border_reflect101(buf, rows, cols, stride, border);
}
int orb_oracle_fast(const uint8_t* img, int rows, int cols, size_t stride, int threshold, int nms, orb_oracle_kp* out,
                    int cap)
{
    std::vector<KP> v;
    fast_detect(img, rows, cols, stride, threshold, nms != 0, v);
    int n = (int)v.size();
    for (int i = 0; i < n && i < cap; i++) out[i] = v[i];
    return n;
}
int orb_oracle_fast_score_closed(const uint8_t* c, size_t stride) { return fast_score_closed(c, stride); }
int orb_oracle_fast_score_2loop(const uint8_t* c, size_t stride, int t) { return fast_score_2loop(c, stride, t); }
void orb_oracle_gaussian_blur7(const uint8_t* src, int rows, int cols, size_t sstride, uint8_t* dst, size_t dstride,
                               const int* taps7)
{
    const int def[7] = {18, 34, 48, 56, 48, 34, 18};
    gaussian_blur7(src, rows, cols, sstride, dst, dstride, taps7 ? taps7 : def);
}
float orb_oracle_fast_atan2(float y, float x) { return fast_atan2(y, x); }
float orb_oracle_fast_atan2_fma(float y, float x) { return fast_atan2(y, x, true); }
void orb_oracle_sincos_cr(float a, float* s, float* c) { orb_sincos_cr_impl(a, s, c); }

int orb_oracle_distribute_octree(const orb_oracle_kp* cands, int n, int minX, int maxX, int minY, int maxY, int N,
                                 orb_oracle_kp* out, int cap)
{
    std::vector<KP> v(cands, cands + n);
    std::vector<KP> r = DistributeOctTree(v, minX, maxX, minY, maxY, N);
    for (int i = 0; i < (int)r.size() && i < cap; i++) out[i] = r[i];
    return (int)r.size();
}

// CPU-baseline helper: `nthreads` threads, each with its own extractor (the reference's
// one-extractor-per-thread protocol, src/Frame.cc:119-122), each running `reps` extractions over the
// given frames.  Returns the total number of keypoints; *seconds = wall time.
long orb_oracle_extract_many2(int nthreads, int reps, int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh,
                              const uint8_t* imgs, int nimg, int rows, int cols, int lap0, int lap1, int fastpath, double* seconds,
                              double* stage_s6 /* summed over the threads; may be NULL */);
long orb_oracle_extract_many(int nthreads, int reps, int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh,
                             const uint8_t* imgs, int nimg, int rows, int cols, int lap0, int lap1, double* seconds)
{
    return orb_oracle_extract_many2(nthreads, reps, nfeatures, scaleFactor, nlevels, iniTh, minTh, imgs, nimg, rows, cols, lap0, lap1, 0,
                                    seconds, nullptr);
}
long orb_oracle_extract_many2(int nthreads, int reps, int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh,
                              const uint8_t* imgs, int nimg, int rows, int cols, int lap0, int lap1, int fastpath, double* seconds,
                              double* stage_s6)
{
    std::vector<orb_oracle*> ex(nthreads);
    for (auto& e : ex) {
        e = new orb_oracle(nfeatures, scaleFactor, nlevels, iniTh, minTh);
        e->fastpath = fastpath != 0;
    }
    std::vector<long> counts(nthreads, 0);
    const int cap = nfeatures + 64 + 16 * nlevels;
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++)
        th.emplace_back([&, t]() {
            std::vector<KP> kps(cap);
            std::vector<uint8_t> desc((size_t)cap * 32);
            long c = 0;
            for (int r = 0; r < reps; r++) {
                int n = 0;
                const uint8_t* im = imgs + (size_t)((t + r) % nimg) * rows * cols;
                ex[t]->extract(im, rows, cols, (size_t)cols, lap0, lap1, kps.data(), desc.data(), cap, &n);
                c += n;
            }
            counts[t] = c;
        });
    for (auto& t : th) t.join();
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    long total = 0;
    for (long c : counts) total += c;
    if (stage_s6)
        for (int i = 0; i < 6; i++) {
            stage_s6[i] = 0;
            for (auto& e : ex) stage_s6[i] += e->stage_s[i];
        }
    for (auto& e : ex) delete e;
    return total;
}

int orb_oracle_descriptor_distance(const uint8_t* a, const uint8_t* b) { return DescriptorDistance(a, b); }

void orb_oracle_hamming_matrix(const uint8_t* A, int nA, const uint8_t* B, int nB, uint16_t* D)
{
    for (int i = 0; i < nA; i++)
        for (int j = 0; j < nB; j++) D[(size_t)i * nB + j] = (uint16_t)DescriptorDistance(A + 32 * i, B + 32 * j);
}

// cv::BFMatcher(NORM_HAMMING).knnMatch(k=2), SURVEY.md B.7: two smallest, ties -> lower train index.
void orb_oracle_bfknn2(const uint8_t* Q, int nQ, const uint8_t* T, int nT, int32_t* idx, int32_t* dist)
{
    for (int q = 0; q < nQ; q++) {
        int d0 = 1 << 30, d1 = 1 << 30, i0 = -1, i1 = -1;
        for (int t = 0; t < nT; t++) {
            int d = DescriptorDistance(Q + 32 * q, T + 32 * t);
            if (d < d0) {
                d1 = d0;
                i1 = i0;
                d0 = d;
                i0 = t;
            } else if (d < d1) {
                d1 = d;
                i1 = t;
            }
        }
        idx[2 * q] = i0;
        idx[2 * q + 1] = i1;
        dist[2 * q] = i0 >= 0 ? d0 : -1;
        dist[2 * q + 1] = i1 >= 0 ? d1 : -1;
    }
}

void orb_oracle_three_maxima(const int* histo, int L, int* i1, int* i2, int* i3)
{
    int a = -1, b = -1, c = -1;
    ComputeThreeMaxima(histo, L, a, b, c);
    *i1 = a;
    *i2 = b;
    *i3 = c;
}

int orb_oracle_search_bow_kf_f(const uint8_t* descKF, int nKF, const uint8_t* maskKF, const float* angKF,
                               const orb_oracle_fv* fvKF, const uint8_t* descF, int nF, const float* angF,
                               const orb_oracle_fv* fvF, int Nleft, float nnratio, int checkOri, int32_t* match)
{
    (void)nKF;
    for (int i = 0; i < nF; i++) match[i] = -1;
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    for_each_shared_node(fvKF, fvF, [&](int a, int b) {
        for (int iKF = fvKF->offsets[a]; iKF < fvKF->offsets[a + 1]; iKF++) {
            const int realIdxKF = fvKF->indices[iKF];
            if (!maskKF[realIdxKF]) continue;
            const uint8_t* dKF = descKF + 32 * (size_t)realIdxKF;
            int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
            int bestDist1R = 256, bestIdxFR = -1, bestDist2R = 256;
            for (int iF = fvF->offsets[b]; iF < fvF->offsets[b + 1]; iF++) {
                const int realIdxF = fvF->indices[iF];
                if (match[realIdxF] >= 0) continue;
                const int dist = DescriptorDistance(dKF, descF + 32 * (size_t)realIdxF);
                if (Nleft == -1) {
                    if (dist < bestDist1) {
                        bestDist2 = bestDist1;
                        bestDist1 = dist;
                        bestIdxF = realIdxF;
                    } else if (dist < bestDist2) {
                        bestDist2 = dist;
                    }
                } else {
                    if (realIdxF < Nleft && dist < bestDist1) {
                        bestDist2 = bestDist1;
                        bestDist1 = dist;
                        bestIdxF = realIdxF;
                    } else if (realIdxF < Nleft && dist < bestDist2) {
                        bestDist2 = dist;
                    }
                    if (realIdxF >= Nleft && dist < bestDist1R) {
                        bestDist2R = bestDist1R;
                        bestDist1R = dist;
                        bestIdxFR = realIdxF;
                    } else if (realIdxF >= Nleft && dist < bestDist2R) {
                        bestDist2R = dist;
                    }
                }
            }
            if (bestDist1 <= TH_LOW) {
                if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                    match[bestIdxF] = realIdxKF;
                    if (checkOri) rotHist[rot_bin(angKF[realIdxKF], angF[bestIdxF])].push_back(bestIdxF);
                    nmatches++;
                }
                if (bestDist1R <= TH_LOW) {
                    // ratio test is "|| true" in the reference (:405)
                    match[bestIdxFR] = realIdxKF;
                    if (checkOri) rotHist[rot_bin(angKF[realIdxKF], angF[bestIdxFR])].push_back(bestIdxFR);
                    nmatches++;
                }
            }
        }
    });
    if (checkOri) nmatches = cull_rotation(rotHist, match, nmatches);
    return nmatches;
}

int orb_oracle_search_bow_kf_kf(const uint8_t* desc1, int n1, const uint8_t* mask1, const float* ang1,
                                const orb_oracle_fv* fv1, int lim1, const uint8_t* desc2, int n2,
                                const uint8_t* mask2, const float* ang2, const orb_oracle_fv* fv2, int lim2,
                                float nnratio, int checkOri, int32_t* match12)
{
    for (int i = 0; i < n1; i++) match12[i] = -1;
    std::vector<uint8_t> vbMatched2(n2, 0);
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    for_each_shared_node(fv1, fv2, [&](int a, int b) {
        for (int i1 = fv1->offsets[a]; i1 < fv1->offsets[a + 1]; i1++) {
            const int idx1 = fv1->indices[i1];
            if (lim1 != -1 && idx1 >= lim1) continue;
            if (!mask1[idx1]) continue;
            const uint8_t* d1 = desc1 + 32 * (size_t)idx1;
            int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
            for (int i2 = fv2->offsets[b]; i2 < fv2->offsets[b + 1]; i2++) {
                const int idx2 = fv2->indices[i2];
                if (lim2 != -1 && idx2 >= lim2) continue;
                if (vbMatched2[idx2] || !mask2[idx2]) continue;
                int dist = DescriptorDistance(d1, desc2 + 32 * (size_t)idx2);
                if (dist < bestDist1) {
                    bestDist2 = bestDist1;
                    bestDist1 = dist;
                    bestIdx2 = idx2;
                } else if (dist < bestDist2) {
                    bestDist2 = dist;
                }
            }
            if (bestDist1 < TH_LOW) {
                if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                    match12[idx1] = bestIdx2;
                    vbMatched2[bestIdx2] = 1;
                    if (checkOri) rotHist[rot_bin(ang1[idx1], ang2[bestIdx2])].push_back(idx1);
                    nmatches++;
                }
            }
        }
    });
    if (checkOri) nmatches = cull_rotation(rotHist, match12, nmatches);
    return nmatches;
}

int orb_oracle_search_triangulation(const uint8_t* desc1, int n1, const uint8_t* hasMP1, const float* kp1xy,
                                    const float* ang1, const int32_t* oct1, const float* uRight1,
                                    const orb_oracle_fv* fv1, const uint8_t* desc2, int n2, const uint8_t* hasMP2,
                                    const float* kp2xy, const float* ang2, const int32_t* oct2, const float* uRight2,
                                    const orb_oracle_fv* fv2, const float* F12, float epx, float epy,
                                    const float* scaleFactors2, const float* levelSigma2_2, int bOnlyStereo,
                                    int bCoarse, int checkOri, int32_t* pairs)
{
    (void)oct1;
    (void)n2;
    std::vector<int32_t> vMatches12(n1, -1);
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    for_each_shared_node(fv1, fv2, [&](int a, int b) {
        for (int i1 = fv1->offsets[a]; i1 < fv1->offsets[a + 1]; i1++) {
            const int idx1 = fv1->indices[i1];
            if (hasMP1[idx1]) continue;
            const bool bStereo1 = uRight1[idx1] >= 0;
            if (bOnlyStereo && !bStereo1) continue;
            const float k1x = kp1xy[2 * idx1], k1y = kp1xy[2 * idx1 + 1];
            const uint8_t* d1 = desc1 + 32 * (size_t)idx1;
            int bestDist = TH_LOW;
            int bestIdx2 = -1;
            for (int i2 = fv2->offsets[b]; i2 < fv2->offsets[b + 1]; i2++) {
                const int idx2 = fv2->indices[i2];
                if (hasMP2[idx2]) continue; // vbMatched2 is never set in the reference
                const bool bStereo2 = uRight2[idx2] >= 0;
                if (bOnlyStereo && !bStereo2) continue;
                const int dist = DescriptorDistance(d1, desc2 + 32 * (size_t)idx2);
                if (dist > TH_LOW || dist > bestDist) continue;
                const float k2x = kp2xy[2 * idx2], k2y = kp2xy[2 * idx2 + 1];
                if (!bStereo1 && !bStereo2) {
                    const float distex = epx - k2x;
                    const float distey = epy - k2y;
                    if (distex * distex + distey * distey < 100 * scaleFactors2[oct2[idx2]]) continue;
                }
                // Pinhole::epipolarConstrain_ (src/CameraModels/Pinhole.cpp:159-181) with F12 given
                bool ok = false;
                {
                    const float la = k1x * F12[0] + k1y * F12[3] + F12[6];
                    const float lb = k1x * F12[1] + k1y * F12[4] + F12[7];
                    const float lc = k1x * F12[2] + k1y * F12[5] + F12[8];
                    const float num = la * k2x + lb * k2y + lc;
                    const float den = la * la + lb * lb;
                    if (den != 0) {
                        const float dsqr = num * num / den;
                        ok = dsqr < 3.84 * levelSigma2_2[oct2[idx2]];
                    }
                }
                if (ok || bCoarse) {
                    bestIdx2 = idx2;
                    bestDist = dist;
                }
            }
            if (bestIdx2 >= 0) {
                vMatches12[idx1] = bestIdx2;
                nmatches++;
                if (checkOri) rotHist[rot_bin(ang1[idx1], ang2[bestIdx2])].push_back(idx1);
            }
        }
    });
    if (checkOri) nmatches = cull_rotation(rotHist, vMatches12.data(), nmatches);
    int np = 0;
    for (int i = 0; i < n1; i++) {
        if (vMatches12[i] < 0) continue;
        pairs[2 * np] = i;
        pairs[2 * np + 1] = vMatches12[i];
        np++;
    }
    return np;
}

// ORBmatcher::SearchByProjection over flattened inputs.  mode 0 = (Frame&, vector<MapPoint*>&, th, ...)
// src/ORBmatcher.cc:44-197: one query per camera a map point is tracked in (qflags bit0 = right camera,
// bit1 = "this is the right-camera search of the previous query's map point", which the `continue` at :128
// skips when the left search was rejected by the ratio test).  mode 1 = best-only search of
// (Frame& CurrentFrame, const Frame& LastFrame, th, bMono) :2193-2419 and of the relocalisation overload
// :2421-2541, with the rotation histogram.  The grid is Frame::AssignFeaturesToGrid (src/Frame.cc:380-410),
// the window Frame::GetFeaturesInArea (:643-708).  Occupancy follows F.mvpMapPoints: `taken` = non-null and
// Observations()>0 on entry; a query's map point blocks a feature when qblocks[q] (NULL = all block).
// ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:706-821), literal: F1 keypoints of level 0 search a
// window around vbPrevMatched in F2's grid (Frame::GetFeaturesInArea, src/Frame.cc:643-708, levels [0, 0]);
// a better match steals an F2 feature from an earlier one (vMatchedDistance / vnMatches21).
int orb_oracle_search_initialization(const orb_oracle_init_args* a, int32_t* vnMatches12)
{
    const int GC = 64, GR = 48;
    std::vector<std::vector<size_t>> mGrid((size_t)GC * GR);
    for (int i = 0; i < a->n2; i++) {
        const int posX = (int)std::round((a->kx2[i] - a->minX) * a->gridWInv);
        const int posY = (int)std::round((a->ky2[i] - a->minY) * a->gridHInv);
        if (posX < 0 || posX >= GC || posY < 0 || posY >= GR) continue;
        mGrid[(size_t)posX * GR + posY].push_back((size_t)i);
    }
    auto GetFeaturesInArea = [&](float x, float y, float r, int minLevel, int maxLevel) {
        std::vector<size_t> vIndices;
        const float fx0 = std::floor((x - a->minX - r) * a->gridWInv);
        if (!(fx0 < (float)GC)) return vIndices;
        const int nMinCellX = fx0 > 0.f ? (int)fx0 : 0;
        const float fx1 = std::ceil((x - a->minX + r) * a->gridWInv);
        if (!(fx1 >= 0.f)) return vIndices;
        const int nMaxCellX = fx1 < (float)(GC - 1) ? (int)fx1 : GC - 1;
        const float fy0 = std::floor((y - a->minY - r) * a->gridHInv);
        if (!(fy0 < (float)GR)) return vIndices;
        const int nMinCellY = fy0 > 0.f ? (int)fy0 : 0;
        const float fy1 = std::ceil((y - a->minY + r) * a->gridHInv);
        if (!(fy1 >= 0.f)) return vIndices;
        const int nMaxCellY = fy1 < (float)(GR - 1) ? (int)fy1 : GR - 1;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const std::vector<size_t>& vCell = mGrid[(size_t)ix * GR + iy];
                for (size_t j = 0; j < vCell.size(); j++) {
                    const size_t g = vCell[j];
                    if (bCheckLevels) {
                        if (a->octave2[g] < minLevel) continue;
                        if (maxLevel >= 0)
                            if (a->octave2[g] > maxLevel) continue;
                    }
                    const float distx = a->kx2[g] - x, disty = a->ky2[g] - y;
                    if (std::fabs(distx) < r && std::fabs(disty) < r) vIndices.push_back(g);
                }
            }
        return vIndices;
    };
    int nmatches = 0;
    for (int i = 0; i < a->n1; i++) vnMatches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    std::vector<int> vMatchedDistance(a->n2, INT_MAX);
    std::vector<int> vnMatches21(a->n2, -1);
    for (int i1 = 0; i1 < a->n1; i1++) {
        const int level1 = a->octave1[i1];
        if (level1 > 0) continue;
        const std::vector<size_t> vIndices2 =
            GetFeaturesInArea(a->prev_xy[2 * i1], a->prev_xy[2 * i1 + 1], (float)a->window_size, level1, level1);
        if (vIndices2.empty()) continue;
        const uint8_t* d1 = a->desc1 + 32 * (size_t)i1;
        int bestDist = INT_MAX;
        int bestDist2 = INT_MAX;
        int bestIdx2 = -1;
        for (size_t k = 0; k < vIndices2.size(); k++) {
            const size_t i2 = vIndices2[k];
            const int dist = DescriptorDistance(d1, a->desc2 + 32 * i2);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestIdx2 = (int)i2;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * a->nnratio) {
                if (vnMatches21[bestIdx2] >= 0) {
                    vnMatches12[vnMatches21[bestIdx2]] = -1;
                    nmatches--;
                }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (a->check_orientation) rotHist[rot_bin(a->angle1[i1], a->angle2[bestIdx2])].push_back(i1);
            }
        }
    }
    if (a->check_orientation) {
        int counts[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) counts[i] = (int)rotHist[i].size();
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(counts, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) {
                const int idx1 = rotHist[i][j];
                if (vnMatches12[idx1] >= 0) {
                    vnMatches12[idx1] = -1;
                    nmatches--;
                }
            }
        }
    }
    return nmatches;
}

int orb_oracle_search_projection(const orb_oracle_proj_args* a, int32_t* q_match, int32_t* feat_match)
{
    const int GC = 64, GR = 48; // FRAME_GRID_COLS / FRAME_GRID_ROWS, include/Frame.h
    const int N = a->n, Nleft = a->Nleft;
    std::vector<std::vector<size_t>> mGrid((size_t)GC * GR), mGridRight((size_t)GC * GR);
    for (int i = 0; i < N; i++) {
        const int posX = (int)std::round((a->kx[i] - a->minX) * a->gridWInv);
        const int posY = (int)std::round((a->ky[i] - a->minY) * a->gridHInv);
        if (posX < 0 || posX >= GC || posY < 0 || posY >= GR) continue;
        if (Nleft == -1 || i < Nleft) mGrid[(size_t)posX * GR + posY].push_back((size_t)i);
        else mGridRight[(size_t)posX * GR + posY].push_back((size_t)(i - Nleft));
    }
    auto GetFeaturesInArea = [&](float x, float y, float r, int minLevel, int maxLevel, bool bRight) {
        std::vector<size_t> vIndices;
        const float factorX = r, factorY = r;
        const float fx0 = std::floor((x - a->minX - factorX) * a->gridWInv);
        if (!(fx0 < (float)GC)) return vIndices;
        const int nMinCellX = fx0 > 0.f ? (int)fx0 : 0;
        const float fx1 = std::ceil((x - a->minX + factorX) * a->gridWInv);
        if (!(fx1 >= 0.f)) return vIndices;
        const int nMaxCellX = fx1 < (float)(GC - 1) ? (int)fx1 : GC - 1;
        const float fy0 = std::floor((y - a->minY - factorY) * a->gridHInv);
        if (!(fy0 < (float)GR)) return vIndices;
        const int nMinCellY = fy0 > 0.f ? (int)fy0 : 0;
        const float fy1 = std::ceil((y - a->minY + factorY) * a->gridHInv);
        if (!(fy1 >= 0.f)) return vIndices;
        const int nMaxCellY = fy1 < (float)(GR - 1) ? (int)fy1 : GR - 1;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const std::vector<size_t>& vCell = !bRight ? mGrid[(size_t)ix * GR + iy] : mGridRight[(size_t)ix * GR + iy];
                for (size_t j = 0; j < vCell.size(); j++) {
                    const size_t g = (Nleft == -1 || !bRight) ? vCell[j] : vCell[j] + (size_t)Nleft;
                    if (bCheckLevels) {
                        if (a->octave[g] < minLevel) continue;
                        if (maxLevel >= 0)
                            if (a->octave[g] > maxLevel) continue;
                    }
                    const float distx = a->kx[g] - x, disty = a->ky[g] - y;
                    if (std::fabs(distx) < factorX && std::fabs(disty) < factorY) vIndices.push_back(vCell[j]);
                }
            }
        return vIndices;
    };
    // occupant of F.mvpMapPoints[i]: -2 NULL or obs==0 on entry, -1 blocking on entry, q >= 0 written by query q
    std::vector<int> occupant(N);
    for (int i = 0; i < N; i++) {
        occupant[i] = (a->taken && a->taken[i]) ? -1 : -2;
        feat_match[i] = -1;
    }
    auto blocked = [&](size_t i) {
        const int o = occupant[i];
        if (o == -2) return false;
        if (o == -1) return true;
        return a->qblocks ? a->qblocks[o] != 0 : true;
    };
    auto write = [&](size_t i, int q) {
        occupant[i] = q;
        feat_match[i] = q;
    };
    std::vector<int> rotHist[HISTO_LENGTH];
    int nmatches = 0;
    bool prevRatioRejected = false, prevAreaEmpty = false;
    for (int q = 0; q < a->nq; q++) {
        q_match[q] = -1;
        const bool bRight = a->qflags && (a->qflags[q] & 1);
        const bool linked = a->qflags && (a->qflags[q] & 2);
        // bit 2: the right-camera search of a point of SearchByProjection(CurrentFrame, LastFrame) sits behind the
        // `if(vIndices2.empty()) continue;` of its left-camera search (src/ORBmatcher.cc:2255-2256, :2326)
        const bool behindArea = a->qflags && (a->qflags[q] & 4);
        const bool skip = (linked && prevRatioRejected) || (behindArea && prevAreaEmpty);
        prevRatioRejected = false;
        prevAreaEmpty = false;
        if (skip) continue;
        const size_t base = bRight ? (size_t)Nleft : 0;
        const float r = a->qr[q];
        const std::vector<size_t> vIndices = GetFeaturesInArea(a->qx[q], a->qy[q], r, a->qmin_level[q], a->qmax_level[q], bRight);
        if (vIndices.empty()) {
            prevAreaEmpty = true;
            continue;
        }
        const uint8_t* MPdescriptor = a->qdesc + 32 * (size_t)q;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (size_t k = 0; k < vIndices.size(); k++) {
            const size_t idx = vIndices[k];
            if (blocked(idx + base)) continue;
            if (a->chi2_gate) {
                // Fuse, src/ORBmatcher.cc:1773-1799.  pKF->mvuRight[idx] is read with the index BEFORE
                // `idx += pKF->NLeft` (:1801), also for right-camera candidates.
                const float kpx = a->kx[idx + base], kpy = a->ky[idx + base];
                const int kpLevel = a->octave[idx + base];
                if (a->uright && a->uright[idx] >= 0) {
                    const float kpr = a->uright[idx];
                    const float ex = a->qx[q] - kpx;
                    const float ey = a->qy[q] - kpy;
                    const float er = a->qxr[q] - kpr;
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * a->inv_level_sigma2[kpLevel] > 7.8) continue;
                } else {
                    const float ex = a->qx[q] - kpx;
                    const float ey = a->qy[q] - kpy;
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * a->inv_level_sigma2[kpLevel] > 5.99) continue;
                }
            } else if (!bRight && Nleft == -1 && a->uright && a->uright[idx] > 0) {
                const float er = std::fabs(a->qxr[q] - a->uright[idx]);
                if (er > r) continue;
            }
            const int dist = DescriptorDistance(MPdescriptor, a->desc + 32 * (idx + base));
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestLevel2 = bestLevel;
                bestLevel = a->octave[idx + base];
                bestIdx = (int)idx;
            } else if (a->mode == 0 && dist < bestDist2) {
                bestLevel2 = a->octave[idx + base];
                bestDist2 = dist;
            }
        }
        if (bestDist <= a->th_high) {
            if (a->mode == 0) {
                if (bestLevel == bestLevel2 && bestDist > a->nnratio * bestDist2) {
                    prevRatioRejected = true;
                    continue;
                }
                if (!bRight) {
                    write((size_t)bestIdx, q);
                    if (Nleft != -1 && a->left_to_right && a->left_to_right[bestIdx] != -1) {
                        write((size_t)(a->left_to_right[bestIdx] + Nleft), q);
                        nmatches++;
                    }
                    nmatches++;
                } else {
                    if (Nleft != -1 && a->right_to_left && a->right_to_left[bestIdx] != -1) {
                        write((size_t)a->right_to_left[bestIdx], q);
                        nmatches++;
                    }
                    write((size_t)bestIdx + base, q);
                    nmatches++;
                }
                q_match[q] = bestIdx + (int)base;
            } else {
                write((size_t)bestIdx + base, q);
                nmatches++;
                q_match[q] = bestIdx + (int)base;
                if (a->check_orientation) rotHist[rot_bin(a->qangle[q], a->angle[bestIdx + base])].push_back(bestIdx + (int)base);
            }
        }
    }
    if (a->mode == 1 && a->check_orientation) nmatches = cull_rotation(rotHist, feat_match, nmatches);
    return nmatches;
}

// MapPoint::ComputeDistinctiveDescriptors, reference src/MapPoint.cc:387-419, for `npts` map points whose
// observation descriptors are pooled: point p owns rows offsets[p] .. offsets[p+1).  best[p] = row (relative
// to the point) with the least median distance to the rest, -1 for a point without descriptors.
void orb_oracle_distinctive_descriptors(const uint8_t* pool, const int32_t* offsets, int npts, int32_t* best)
{
    for (int p = 0; p < npts; p++) {
        const int N = offsets[p + 1] - offsets[p];
        const uint8_t* D = pool + 32 * (size_t)offsets[p];
        if (N <= 0) {
            best[p] = -1;
            continue;
        }
        std::vector<std::vector<int>> Distances(N, std::vector<int>(N, 0));
        for (int i = 0; i < N; i++)
            for (int j = i + 1; j < N; j++) {
                const int d = DescriptorDistance(D + 32 * (size_t)i, D + 32 * (size_t)j);
                Distances[i][j] = d;
                Distances[j][i] = d;
            }
        int BestMedian = 0x7fffffff, BestIdx = 0;
        for (int i = 0; i < N; i++) {
            std::vector<int> vDists(Distances[i]);
            std::sort(vDists.begin(), vDists.end());
            const int median = vDists[(size_t)(0.5 * (N - 1))];
            if (median < BestMedian) {
                BestMedian = median;
                BestIdx = i;
            }
        }
        best[p] = BestIdx;
    }
}

// TemplatedVocabulary::transform for one feature, reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1217-1259
// (F::distance = FORB::distance = Hamming, FORB.cpp:81-101).  The tree is given in CSR form.
void orb_oracle_vocab_transform(int nnodes, const uint8_t* node_desc, const int32_t* child_off, const int32_t* child_ids,
                                const int32_t* node_word, const double* node_weight, int L, const uint8_t* feats, int n,
                                int levelsup, int32_t* word_id, int32_t* node_id, double* weight)
{
    (void)nnodes;
    for (int f = 0; f < n; f++) {
        const uint8_t* feature = feats + 32 * (size_t)f;
        const int nid_level = L - levelsup;
        int nid = 0; // root when nid_level <= 0
        int final_id = 0, current_level = 0;
        do {
            ++current_level;
            const int c0 = child_off[final_id], c1 = child_off[final_id + 1];
            if (c0 >= c1) break; // malformed: inner node without children
            final_id = child_ids[c0];
            double best_d = DescriptorDistance(feature, node_desc + 32 * (size_t)final_id);
            for (int k = c0 + 1; k < c1; k++) {
                const int id = child_ids[k];
                const double d = DescriptorDistance(feature, node_desc + 32 * (size_t)id);
                if (d < best_d) {
                    best_d = d;
                    final_id = id;
                }
            }
            if (current_level == nid_level) nid = final_id;
        } while (child_off[final_id] < child_off[final_id + 1]);
        word_id[f] = node_word[final_id];
        weight[f] = node_weight[final_id];
        node_id[f] = nid;
    }
}

// TemplatedVocabulary::transform(features, v, fv, levelsup), reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1192,
// on the containers the reference uses: BowVector = std::map<WordId, WordValue> (BowVector.h:59-60), FeatureVector =
// std::map<NodeId, std::vector<unsigned int>> (FeatureVector.h:23-24).
int orb_oracle_compute_bow(int nnodes, const uint8_t* node_desc, const int32_t* child_off, const int32_t* child_ids,
                           const int32_t* node_word, const double* node_weight, int L, const uint8_t* feats, int n, int levelsup,
                           int weighting, int scoring, uint32_t* bow_ids, double* bow_vals, uint32_t* node_ids, int32_t* offsets,
                           int32_t* indices, int* nn_out)
{
    std::map<unsigned, double> v;                 // v.clear()  :1131
    std::map<unsigned, std::vector<unsigned>> fv; // fv.clear() :1132
    // m_scoring_object->mustNormalize(norm) (ScoringObject.h:73-89): every scoring but DOT_PRODUCT normalises, L2_NORM with L2
    const bool must = scoring != 5;
    const bool normL2 = scoring == 1;
    std::vector<int32_t> word((size_t)std::max(n, 1)), nid((size_t)std::max(n, 1));
    std::vector<double> wt((size_t)std::max(n, 1));
    orb_oracle_vocab_transform(nnodes, node_desc, child_off, child_ids, node_word, node_weight, L, feats, n, levelsup, word.data(),
                               nid.data(), wt.data());
    const bool tf = weighting == 0 || weighting == 1; // TF_IDF || TF :1145
    for (int i = 0; i < n; i++) {                     // i_feature :1147 / :1174
        const unsigned id = (unsigned)word[(size_t)i];
        const double w = wt[(size_t)i];
        if (w > 0) { // not stopped :1157 / :1184
            auto vit = v.lower_bound(id);
            if (tf) { // BowVector::addWeight (BowVector.cpp:34-46)
                if (vit != v.end() && !(v.key_comp()(id, vit->first))) vit->second += w;
                else v.insert(vit, std::make_pair(id, w));
            } else { // BowVector::addIfNotExist (BowVector.cpp:50-58)
                if (vit == v.end() || v.key_comp()(id, vit->first)) v.insert(vit, std::make_pair(id, w));
            }
            // FeatureVector::addFeature (FeatureVector.cpp:31-45)
            const unsigned node = (unsigned)nid[(size_t)i];
            auto fit = fv.lower_bound(node);
            if (fit != fv.end() && fit->first == node) fit->second.push_back((unsigned)i);
            else {
                fit = fv.insert(fit, std::make_pair(node, std::vector<unsigned>()));
                fit->second.push_back((unsigned)i);
            }
        }
    }
    if (tf && !v.empty() && !must) { // unnecessary when normalizing :1164-1170
        const double nd = (double)v.size();
        for (auto& e : v) e.second /= nd;
    }
    if (must) { // BowVector::normalize (BowVector.cpp:62-86)
        double norm = 0.0;
        if (!normL2) {
            for (auto& e : v) norm += std::fabs(e.second);
        } else {
            for (auto& e : v) norm += e.second * e.second;
            norm = std::sqrt(norm);
        }
        if (norm > 0.0)
            for (auto& e : v) e.second /= norm;
    }
    int k = 0;
    for (auto& e : v) {
        bow_ids[k] = e.first;
        bow_vals[k] = e.second;
        k++;
    }
    int s = 0, at = 0;
    for (auto& e : fv) {
        node_ids[s] = e.first;
        offsets[s] = at;
        for (unsigned f : e.second) indices[at++] = (int32_t)f;
        s++;
    }
    offsets[s] = at;
    if (nn_out) *nn_out = s;
    return k;
}

// Frame::ComputeStereoMatches, reference src/Frame.cc:797-967.  L / R are the oracle extractors that
// processed the left / right image (their mvImagePyramid is read for the SAD refinement).
int orb_oracle_compute_stereo_matches(orb_oracle* L, orb_oracle* R, const orb_oracle_kp* mvKeys,
                                      const uint8_t* mDescriptors, int N, const orb_oracle_kp* mvKeysRight,
                                      const uint8_t* mDescriptorsRight, int Nr, float mb, float mbf, float* mvuRight,
                                      float* mvDepth)
{
    const int TH_HIGH = 100;
    for (int i = 0; i < N; i++) {
        mvuRight[i] = -1.0f;
        mvDepth[i] = -1.0f;
    }
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = L->pyr[0].rows;
    std::vector<std::vector<size_t>> vRowIndices(nRows);
    for (int iR = 0; iR < Nr; iR++) {
        const KP& kp = mvKeysRight[iR];
        const float kpY = kp.y;
        const float r = 2.0f * R->mvScaleFactor[kp.octave];
        const int maxr = (int)std::ceil(kpY + r);
        const int minr = (int)std::floor(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR); // the reference indexes unchecked
    }
    const float minZ = mb;
    const float minD = 0;
    const float maxD = mbf / minZ;
    std::vector<std::pair<int, int>> vDistIdx;
    for (int iL = 0; iL < N; iL++) {
        const KP& kpL = mvKeys[iL];
        const int levelL = kpL.octave;
        const float vL = kpL.y;
        const float uL = kpL.x;
        if ((int)vL < 0 || (int)vL >= nRows) continue;
        const std::vector<size_t>& vCandidates = vRowIndices[(size_t)vL];
        if (vCandidates.empty()) continue;
        const float minU = uL - maxD;
        const float maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH;
        size_t bestIdxR = 0;
        const uint8_t* dL = mDescriptors + 32 * (size_t)iL;
        for (size_t iC = 0; iC < vCandidates.size(); iC++) {
            const size_t iR = vCandidates[iC];
            const KP& kpR = mvKeysRight[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = DescriptorDistance(dL, mDescriptorsRight + 32 * iR);
                if (dist < bestDist) {
                    bestDist = dist;
                    bestIdxR = iR;
                }
            }
        }
        if (bestDist < thOrbDist) {
            const float uR0 = mvKeysRight[bestIdxR].x;
            const float scaleFactor = L->mvInvScaleFactor[kpL.octave];
            const float scaleduL = std::round(kpL.x * scaleFactor);
            const float scaledvL = std::round(kpL.y * scaleFactor);
            const float scaleduR0 = std::round(uR0 * scaleFactor);
            const int w = 5;
            Level& PL = L->pyr[kpL.octave];
            Level& PR = R->pyr[kpL.octave];
            int bestDistS = 0x7fffffff;
            int bestincR = 0;
            const int Lw = 5;
            std::vector<float> vDists(2 * Lw + 1);
            const float iniu = scaleduR0 + Lw - w;
            const float endu = scaleduR0 + Lw + w + 1;
            if (iniu < 0 || endu >= PR.cols) continue;
            for (int incR = -Lw; incR <= +Lw; incR++) {
                // cv::norm(IL, IR, NORM_L1) over the 11x11 windows
                double nrm = 0;
                for (int dy = -w; dy <= w; dy++) {
                    const uint8_t* rl = PL.roi() + (size_t)((int)scaledvL + dy) * PL.stride + ((int)scaleduL - w);
                    const uint8_t* rr = PR.roi() + (size_t)((int)scaledvL + dy) * PR.stride + ((int)scaleduR0 + incR - w);
                    for (int dx = 0; dx <= 2 * w; dx++) nrm += std::abs((int)rl[dx] - (int)rr[dx]);
                }
                float dist = (float)nrm;
                if (dist < bestDistS) {
                    bestDistS = (int)dist;
                    bestincR = incR;
                }
                vDists[Lw + incR] = dist;
            }
            if (bestincR == -Lw || bestincR == Lw) continue;
            const float dist1 = vDists[Lw + bestincR - 1];
            const float dist2 = vDists[Lw + bestincR];
            const float dist3 = vDists[Lw + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = L->mvScaleFactor[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) {
                    disparity = 0.01;
                    bestuR = uL - 0.01;
                }
                mvDepth[iL] = mbf / disparity;
                mvuRight[iL] = bestuR;
                vDistIdx.push_back(std::pair<int, int>(bestDistS, iL));
            }
        }
    }
    if (vDistIdx.empty()) return 0; // the reference reads vDistIdx[0] of an empty vector here
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    int kept = (int)vDistIdx.size();
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
        if (vDistIdx[i].first < thDist) break;
        mvuRight[vDistIdx[i].second] = -1;
        mvDepth[vDistIdx[i].second] = -1;
        kept--;
    }
    return kept;
}

// ---------------------------------------------------------------------------------------------------
// KannalaBrandt8 gate of SearchForTriangulation_ for fisheye rigs: KannalaBrandt8::epipolarConstrain_
// (src/CameraModels/KannalaBrandt8.cpp:239-242) = TriangulateMatches_ (:409-480) > 0.0001f, with project
// (:25-41), unproject (:96-123), Triangulate_ (:514-535) and, inside it, cv::SVD::compute of a 4x4 float
// matrix.  OpenCV is absent here, so the SVD is restated from its published algorithm (one-sided Jacobi,
// modules/core/src/lapack.cpp JacobiSVDImpl_<float>, scalar path; eps = 2 FLT_EPSILON; double accumulators):
// like every OpenCV primitive of this oracle it is unpinned.  cv::Matx arithmetic is float with sequential
// accumulation (MatxMul, dot), cv::norm accumulates in double, Matx / scalar multiplies by 1.f / scalar.
namespace {

void kb8_project_pt(const float* P, float x, float y, float z, float* u, float* v)
{
    const float x2_plus_y2 = x * x + y * y;
    const float theta = atan2f(sqrtf(x2_plus_y2), z);
    const float psi = atan2f(y, x);
    const float theta2 = theta * theta;
    const float theta3 = theta * theta2;
    const float theta5 = theta3 * theta2;
    const float theta7 = theta5 * theta2;
    const float theta9 = theta7 * theta2;
    const float r = theta + P[4] * theta3 + P[5] * theta5 + P[6] * theta7 + P[7] * theta9;
    // `cos(psi)` on a float with <cmath>: the float overload
    *u = P[0] * r * cosf(psi) + P[2];
    *v = P[1] * r * sinf(psi) + P[3];
}

// rows of vt = right singular vectors, by descending singular value
void jacobi_svd_vt_4x4(const float A[16], float vt[16])
{
    const int m = 4, n = 4;
    float At[16];
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 4; k++) At[i * 4 + k] = A[k * 4 + i]; // transpose(src, temp_a)
    double W[4];
    const float eps = FLT_EPSILON * 2;
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) {
            const float t = At[i * 4 + k];
            sd += (double)t * t;
        }
        W[i] = sd;
        for (int k = 0; k < n; k++) vt[i * 4 + k] = 0;
        vt[i * 4 + i] = 1;
    }
    const int max_iter = 30; // std::max(m, 30)
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
        for (int i = 0; i < n - 1; i++)
            for (int j = i + 1; j < n; j++) {
                float *Ai = At + i * 4, *Aj = At + j * 4;
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < m; k++) p += (double)Ai[k] * Aj[k];
                if (std::abs(p) <= eps * std::sqrt((double)a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = hypot((double)p, beta);
                float c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = (float)std::sqrt(delta / gamma);
                    c = (float)(p / (gamma * s * 2));
                } else {
                    c = (float)std::sqrt((gamma + beta) / (gamma * 2));
                    s = (float)(p / (gamma * c * 2));
                }
                a = b = 0;
                for (int k = 0; k < m; k++) {
                    const float t0 = c * Ai[k] + s * Aj[k];
                    const float t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0;
                    Aj[k] = t1;
                    a += (double)t0 * t0;
                    b += (double)t1 * t1;
                }
                W[i] = a;
                W[j] = b;
                changed = true;
                float *Vi = vt + i * 4, *Vj = vt + j * 4;
                for (int k = 0; k < n; k++) {
                    const float t0 = c * Vi[k] + s * Vj[k];
                    const float t1 = -s * Vi[k] + c * Vj[k];
                    Vi[k] = t0;
                    Vj[k] = t1;
                }
            }
        if (!changed) break;
    }
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) {
            const float t = At[i * 4 + k];
            sd += (double)t * t;
        }
        W[i] = std::sqrt(sd);
    }
    for (int i = 0; i < n - 1; i++) {
        int j = i;
        for (int k = i + 1; k < n; k++)
            if (W[j] < W[k]) j = k;
        if (i != j) {
            std::swap(W[i], W[j]);
            for (int k = 0; k < m; k++) std::swap(At[i * 4 + k], At[j * 4 + k]);
            for (int k = 0; k < n; k++) std::swap(vt[i * 4 + k], vt[j * 4 + k]);
        }
    }
}

void matx33_mul31(const float* R, const float* v, float* out)
{
    for (int i = 0; i < 3; i++) {
        float s = 0;
        for (int k = 0; k < 3; k++) s += R[i * 3 + k] * v[k];
        out[i] = s;
    }
}
float matx_dot3(const float* a, const float* b)
{
    float s = 0;
    for (int i = 0; i < 3; i++) s += a[i] * b[i];
    return s;
}
double matx_norm3(const float* a)
{
    double s = 0;
    for (int i = 0; i < 3; i++) {
        const double v = a[i];
        s += v * v;
    }
    return std::sqrt(s);
}

} // namespace

float orb_oracle_kb8_triangulate_matches(const float* P1, const float* P2, const float* kp1, const float* kp2,
                                         const float* R12, const float* t12, float sigmaLevel, float unc, float* p3D)
{
    float r1[3], r2[3];
    orb_oracle_kb8_unproject(P1, kp1, 1, r1);
    orb_oracle_kb8_unproject(P2, kp2, 1, r2);
    // Check parallax
    float r21[3];
    matx33_mul31(R12, r2, r21);
    const float cosParallaxRays = (float)(matx_dot3(r1, r21) / (matx_norm3(r1) * matx_norm3(r21)));
    if (cosParallaxRays > 0.9998) return -1;
    // Parallax is good, so we try to triangulate
    const float p11x = r1[0], p11y = r1[1], p22x = r2[0], p22y = r2[1];
    float Tcw1[16] = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float R21[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R21[i * 3 + j] = R12[j * 3 + i];
    float nR21[9], t21[3];
    for (int i = 0; i < 9; i++) nR21[i] = R21[i] * -1.f; // -R21 (Matx unary minus = scale by -1)
    matx33_mul31(nR21, t12, t21);
    float Tcw2[16] = {R21[0], R21[1], R21[2], t21[0], R21[3], R21[4], R21[5], t21[1],
                      R21[6], R21[7], R21[8], t21[2], 0.f, 0.f, 0.f, 1.f};
    // Triangulate_
    float A[16];
    for (int k = 0; k < 4; k++) {
        A[0 * 4 + k] = p11x * Tcw1[2 * 4 + k] - Tcw1[0 * 4 + k];
        A[1 * 4 + k] = p11y * Tcw1[2 * 4 + k] - Tcw1[1 * 4 + k];
        A[2 * 4 + k] = p22x * Tcw2[2 * 4 + k] - Tcw2[0 * 4 + k];
        A[3 * 4 + k] = p22y * Tcw2[2 * 4 + k] - Tcw2[1 * 4 + k];
    }
    float vt[16];
    jacobi_svd_vt_4x4(A, vt);
    const float inv = 1.f / vt[3 * 4 + 3]; // Matx / float
    const float x3D[3] = {vt[3 * 4 + 0] * inv, vt[3 * 4 + 1] * inv, vt[3 * 4 + 2] * inv};
    const float z1 = x3D[2];
    if (z1 <= 0) return -1;
    const float z2 = matx_dot3(R21 + 6, x3D) + t21[2];
    if (z2 <= 0) return -1;
    // Check reprojection error
    float u1, v1;
    kb8_project_pt(P1, x3D[0], x3D[1], x3D[2], &u1, &v1);
    const float errX1 = u1 - kp1[0];
    const float errY1 = v1 - kp1[1];
    if ((errX1 * errX1 + errY1 * errY1) > 5.991 * sigmaLevel) return -1;
    float x3D2[3];
    matx33_mul31(R21, x3D, x3D2);
    for (int i = 0; i < 3; i++) x3D2[i] = x3D2[i] + t21[i];
    float u2, v2;
    kb8_project_pt(P2, x3D2[0], x3D2[1], x3D2[2], &u2, &v2);
    const float errX2 = u2 - kp2[0];
    const float errY2 = v2 - kp2[1];
    if ((errX2 * errX2 + errY2 * errY2) > 5.991 * unc) return -1;
    if (p3D) {
        p3D[0] = x3D[0];
        p3D[1] = x3D[1];
        p3D[2] = x3D[2];
    }
    return z1;
}

// KannalaBrandt8::matchAndtriangulate (src/CameraModels/KannalaBrandt8.cpp:244-335) and the cv::Mat Triangulate it
// calls (:498-512).  cv::Mat arithmetic on CV_32F: a matrix product (gemm) and Mat::dot accumulate in double and
// round once; `s*row - row` is float (addWeighted); `m / s` scales by (float)(1.0 / s); cv::norm is double.
int orb_oracle_kb8_match_and_triangulate(const float* P1, const float* P2, const float* kp1, const float* kp2,
                                         const float* Tcw1, const float* Tcw2, float sigmaLevel1, float sigmaLevel2,
                                         float* x3Dout)
{
    auto Rc = [](const float* T, int i, int j) { return T[i * 4 + j]; };
    float r1[3], r2[3];
    orb_oracle_kb8_unproject(P1, kp1, 1, r1); // ray1c, ray2c
    orb_oracle_kb8_unproject(P2, kp2, 1, r2);
    // Check parallax between rays: ray = Rwc * r (Rwc = Rcw^T)
    float ray1[3], ray2[3];
    for (int i = 0; i < 3; i++) {
        double s1 = 0, s2 = 0;
        for (int k = 0; k < 3; k++) {
            s1 += (double)Rc(Tcw1, k, i) * (double)r1[k];
            s2 += (double)Rc(Tcw2, k, i) * (double)r2[k];
        }
        ray1[i] = (float)s1;
        ray2[i] = (float)s2;
    }
    double dot = 0;
    for (int i = 0; i < 3; i++) dot += (double)ray1[i] * (double)ray2[i];
    const float cosParallaxRays = (float)(dot / (matx_norm3(ray1) * matx_norm3(ray2)));
    if (cosParallaxRays > 0.9998) return 0;
    // Triangulate(p11, p22, Tcw1, Tcw2, x3D)
    float A[16];
    for (int k = 0; k < 4; k++) {
        A[0 * 4 + k] = r1[0] * Tcw1[2 * 4 + k] - Tcw1[0 * 4 + k];
        A[1 * 4 + k] = r1[1] * Tcw1[2 * 4 + k] - Tcw1[1 * 4 + k];
        A[2 * 4 + k] = r2[0] * Tcw2[2 * 4 + k] - Tcw2[0 * 4 + k];
        A[3 * 4 + k] = r2[1] * Tcw2[2 * 4 + k] - Tcw2[1 * 4 + k];
    }
    float vt[16];
    jacobi_svd_vt_4x4(A, vt);
    const float inv = (float)(1.0 / (double)vt[3 * 4 + 3]);
    const float x3D[3] = {vt[3 * 4 + 0] * inv, vt[3 * 4 + 1] * inv, vt[3 * 4 + 2] * inv};
    // Check triangulation in front of cameras
    auto depth = [&](const float* T) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Rc(T, 2, k) * (double)x3D[k];
        return (float)(s + (double)T[2 * 4 + 3]);
    };
    if (depth(Tcw1) <= 0) return 0;
    if (depth(Tcw2) <= 0) return 0;
    // Check reprojection errors: x3Dc = Rcw * x3D + tcw (one gemm)
    auto reproj_ok = [&](const float* P, const float* T, const float* kp, float sigma) {
        float xc[3];
        for (int i = 0; i < 3; i++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += (double)Rc(T, i, k) * (double)x3D[k];
            xc[i] = (float)(s + (double)T[i * 4 + 3]);
        }
        float u, v;
        kb8_project_pt(P, xc[0], xc[1], xc[2], &u, &v);
        const float errX = u - kp[0];
        const float errY = v - kp[1];
        return !((errX * errX + errY * errY) > 5.991 * sigma);
    };
    if (!reproj_ok(P1, Tcw1, kp1, sigmaLevel1)) return 0;
    if (!reproj_ok(P2, Tcw2, kp2, sigmaLevel2)) return 0;
    for (int i = 0; i < 3; i++) x3Dout[i] = x3D[i];
    return 1;
}

// ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo, vMatchedPoints), src/ORBmatcher.cc
// :1452-1641 (bOnlyStereo and F12 are not read by it; no epipole gate; the gate is matchAndtriangulate).
int orb_oracle_search_triangulation_3d(const orb_oracle_tri3d_args* a, int32_t* pairs, float* points)
{
    std::vector<int32_t> vMatches12(a->n1, -1);
    std::vector<float> vMatchesPoints12(3 * (size_t)a->n1, 0.f);
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    for_each_shared_node(a->fv1, a->fv2, [&](int na, int nb) {
        for (int i1 = a->fv1->offsets[na]; i1 < a->fv1->offsets[na + 1]; i1++) {
            const int idx1 = a->fv1->indices[i1];
            if (a->hasMP1[idx1]) continue;
            const float* kp1 = a->kp1xy + 2 * idx1;
            const bool bRight1 = !(a->Nleft1 == -1 || idx1 < a->Nleft1);
            const uint8_t* d1 = a->desc1 + 32 * (size_t)idx1;
            int bestDist = TH_LOW;
            int bestIdx2 = -1;
            float bestPoint[3] = {0.f, 0.f, 0.f};
            for (int i2 = a->fv2->offsets[nb]; i2 < a->fv2->offsets[nb + 1]; i2++) {
                const int idx2 = a->fv2->indices[i2];
                if (a->hasMP2[idx2]) continue; // (vbMatched2 is never set, :1506)
                const int dist = DescriptorDistance(d1, a->desc2 + 32 * (size_t)idx2);
                if (dist > TH_LOW || dist > bestDist) continue;
                const float* kp2 = a->kp2xy + 2 * idx2;
                const bool bRight2 = !(a->Nleft2 == -1 || idx2 < a->Nleft2);
                const float* P1 = bRight1 ? a->kb8_1R : a->kb8_1L;
                const float* P2 = bRight2 ? a->kb8_2R : a->kb8_2L;
                if (!P1) continue; // Pinhole::matchAndtriangulate returns false
                float x3D[3];
                if (orb_oracle_kb8_match_and_triangulate(P1, P2, kp1, kp2, bRight1 ? a->Tcw1R : a->Tcw1L,
                                                         bRight2 ? a->Tcw2R : a->Tcw2L, a->levelSigma2_1[a->oct1[idx1]],
                                                         a->levelSigma2_2[a->oct2[idx2]], x3D)) {
                    bestIdx2 = idx2;
                    bestDist = dist;
                    for (int k = 0; k < 3; k++) bestPoint[k] = x3D[k];
                }
            }
            if (bestIdx2 >= 0) {
                vMatches12[idx1] = bestIdx2;
                for (int k = 0; k < 3; k++) vMatchesPoints12[3 * (size_t)idx1 + k] = bestPoint[k];
                nmatches++;
                if (a->check_orientation) rotHist[rot_bin(a->ang1[idx1], a->ang2[bestIdx2])].push_back(idx1);
            }
        }
    });
    if (a->check_orientation) nmatches = cull_rotation(rotHist, vMatches12.data(), nmatches);
    int np = 0;
    for (int i = 0; i < a->n1; i++) {
        if (vMatches12[i] < 0) continue;
        pairs[2 * np] = i;
        pairs[2 * np + 1] = vMatches12[i];
        for (int k = 0; k < 3; k++) points[3 * np + k] = vMatchesPoints12[3 * (size_t)i + k];
        np++;
    }
    return np;
}

// Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1119-1159): knn-2 brute force between the lapping-area
// descriptors of the two fisheye images, Lowe ratio 0.7, then KannalaBrandt8::TriangulateMatches per survivor.
// Inputs are the lapping-area slices (the caller adds monoLeft / monoRight to the indices).
int orb_oracle_stereo_fisheye_matches(const uint8_t* descL, const float* kpL_xy, const int32_t* octL, int nL,
                                      const uint8_t* descR, const float* kpR_xy, const int32_t* octR, int nR,
                                      const float* P1, const float* P2, const float* Rlr, const float* tlr,
                                      const float* levelSigma2, int32_t* leftToRight, int32_t* rightToLeft,
                                      float* depth, float* p3D)
{
    for (int i = 0; i < nL; i++) {
        leftToRight[i] = -1;
        depth[i] = -1.0f;
        p3D[3 * i] = p3D[3 * i + 1] = p3D[3 * i + 2] = 0.f;
    }
    for (int i = 0; i < nR; i++) rightToLeft[i] = -1;
    if (nL == 0 || nR == 0) return 0;
    std::vector<int32_t> idx(2 * (size_t)nL), dist(2 * (size_t)nL);
    orb_oracle_bfknn2(descL, nL, descR, nR, idx.data(), dist.data());
    int nMatches = 0;
    for (int q = 0; q < nL; q++) {
        if (nR < 2) continue;                                                        // (*it).size() >= 2
        if (!((float)dist[2 * q] < (float)dist[2 * q + 1] * 0.7)) continue;          // float distance * double 0.7
        const int t = idx[2 * q];
        const float sigma1 = levelSigma2[octL[q]], sigma2 = levelSigma2[octR[t]];
        float X[3] = {0.f, 0.f, 0.f};
        const float d = orb_oracle_kb8_triangulate_matches(P1, P2, kpL_xy + 2 * q, kpR_xy + 2 * t, Rlr, tlr, sigma1, sigma2, X);
        if (d > 0.0001f) {
            leftToRight[q] = t;
            rightToLeft[t] = q;
            p3D[3 * q] = X[0];
            p3D[3 * q + 1] = X[1];
            p3D[3 * q + 2] = X[2];
            depth[q] = d;
            nMatches++;
        }
    }
    return nMatches;
}

// SearchForTriangulation_ (src/ORBmatcher.cc:1208-1449) with KannalaBrandt8 cameras: a monocular fisheye
// keyframe pair (Nleft == -1, one camera each) or a two-camera rig (features [0, Nleft) from the left camera,
// the rest from the right one, :1293-1297, and the four relative poses ll / lr / rl / rr of :1238-1248).
int orb_oracle_search_triangulation_kb8(const orb_oracle_tri_kb8_args* a, int32_t* pairs)
{
    std::vector<int32_t> vMatches12(a->n1, -1);
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const bool rig = a->Nleft1 != -1 && a->Nleft2 != -1; // pKF1->mpCamera2 && pKF2->mpCamera2
    for_each_shared_node(a->fv1, a->fv2, [&](int na, int nb) {
        for (int i1 = a->fv1->offsets[na]; i1 < a->fv1->offsets[na + 1]; i1++) {
            const int idx1 = a->fv1->indices[i1];
            if (a->hasMP1[idx1]) continue;
            const bool bStereo1 = !rig && a->uRight1 && a->uRight1[idx1] >= 0; // :1286
            if (a->only_stereo && !bStereo1) continue;
            const float* kp1 = a->kp1xy + 2 * idx1;
            const bool bRight1 = !(a->Nleft1 == -1 || idx1 < a->Nleft1);
            const uint8_t* d1 = a->desc1 + 32 * (size_t)idx1;
            int bestDist = TH_LOW;
            int bestIdx2 = -1;
            for (int i2 = a->fv2->offsets[nb]; i2 < a->fv2->offsets[nb + 1]; i2++) {
                const int idx2 = a->fv2->indices[i2];
                if (a->hasMP2[idx2]) continue;
                const bool bStereo2 = !rig && a->uRight2 && a->uRight2[idx2] >= 0;
                if (a->only_stereo && !bStereo2) continue;
                const int dist = DescriptorDistance(d1, a->desc2 + 32 * (size_t)idx2);
                if (dist > TH_LOW || dist > bestDist) continue;
                const float* kp2 = a->kp2xy + 2 * idx2;
                const bool bRight2 = !(a->Nleft2 == -1 || idx2 < a->Nleft2);
                if (!bStereo1 && !bStereo2 && !rig) {
                    const float distex = a->ep[0] - kp2[0];
                    const float distey = a->ep[1] - kp2[1];
                    if (distex * distex + distey * distey < 100 * a->scaleFactors2[a->oct2[idx2]]) continue;
                }
                int sel = 0; // ll
                const float *P1 = a->kb8_1L, *P2 = a->kb8_2L;
                if (rig) {
                    if (bRight1 && bRight2) { sel = 3; P1 = a->kb8_1R; P2 = a->kb8_2R; }
                    else if (bRight1 && !bRight2) { sel = 2; P1 = a->kb8_1R; P2 = a->kb8_2L; }
                    else if (!bRight1 && bRight2) { sel = 1; P1 = a->kb8_1L; P2 = a->kb8_2R; }
                }
                const bool ok = orb_oracle_kb8_triangulate_matches(P1, P2, kp1, kp2, a->R12 + 9 * sel, a->t12 + 3 * sel,
                                                                   a->levelSigma2_1[a->oct1[idx1]],
                                                                   a->levelSigma2_2[a->oct2[idx2]], nullptr) > 0.0001f;
                if (ok || a->coarse) {
                    bestIdx2 = idx2;
                    bestDist = dist;
                }
            }
            if (bestIdx2 >= 0) {
                vMatches12[idx1] = bestIdx2;
                nmatches++;
                if (a->check_orientation) rotHist[rot_bin(a->ang1[idx1], a->ang2[bestIdx2])].push_back(idx1);
            }
        }
    });
    if (a->check_orientation) nmatches = cull_rotation(rotHist, vMatches12.data(), nmatches);
    int np = 0;
    for (int i = 0; i < a->n1; i++) {
        if (vMatches12[i] < 0) continue;
        pairs[2 * np] = i;
        pairs[2 * np + 1] = vMatches12[i];
        np++;
    }
    return np;
}

// cv::SVD::compute of a 4 x 4 float matrix as restated above (rows of vt = right singular vectors by descending singular
// value): exported for the OpenCV differential harness (adapters/diff_opencv.cpp)
void orb_oracle_svd_vt_4x4(const float* A16, float* vt16) { jacobi_svd_vt_4x4(A16, vt16); }

void orb_oracle_kb8_unproject(const float* P, const float* uv, int n, float* rays)
{
    const float precision = 1e-6f; // reference include/CameraModels/KannalaBrandt8.h (precision member)
    for (int i = 0; i < n; i++) {
        float pwx = (uv[2 * i] - P[2]) / P[0], pwy = (uv[2 * i + 1] - P[3]) / P[1];
        float scale = 1.f;
        float theta_d = sqrtf(pwx * pwx + pwy * pwy);
        theta_d = fminf(fmaxf((float)(-3.14159265358979323846 / 2.f), theta_d), (float)(3.14159265358979323846 / 2.f));
        if (theta_d > 1e-8) {
            float theta = theta_d;
            for (int j = 0; j < 10; j++) {
                float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2,
                      theta8 = theta4 * theta4;
                float k0_theta2 = P[4] * theta2, k1_theta4 = P[5] * theta4;
                float k2_theta6 = P[6] * theta6, k3_theta8 = P[7] * theta8;
                float theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                                  (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
                theta = theta - theta_fix;
                if (fabsf(theta_fix) < precision) break;
            }
            scale = std::tan(theta) / theta_d;
        }
        rays[3 * i] = pwx * scale;
        rays[3 * i + 1] = pwy * scale;
        rays[3 * i + 2] = 1.f;
    }
}

} // extern "C"
