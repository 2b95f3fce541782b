/*
 * orbfe_kb8.h -- KannalaBrandt8::unproject (reference src/CameraModels/KannalaBrandt8.cpp:96-123) on the device.
 * P = fx, fy, cx, cy, k0..k3.  Every float operation is separately rounded like the reference's scalar
 * code; sqrt goes through double (v_sqrt_f32 is 1-ulp) and tan through double (no libm on the device:
 * the only difference to the host is tanf's < 1 ulp, see tests/test_gpu_matcher.py::test_kb8_unproject).
 */
#ifndef ORBFE_KB8_H
#define ORBFE_KB8_H
#ifdef ORBFE_KB8_TIMING // tuning only: where a thread of the triangulation spends its time (sums over threads, 100-MHz ticks)
static __device__ unsigned long long g_kb8Times[8];
#define KT_BEGIN() unsigned long long ktPrev = (unsigned long long)wall_clock64()
#define KT(k)                                                              \
    do {                                                                   \
        const unsigned long long n_ = (unsigned long long)wall_clock64();  \
        atomicAdd(&g_kb8Times[k], n_ - ktPrev);                            \
        ktPrev = n_;                                                       \
    } while (0)
#else
#define KT_BEGIN() do { } while (0)
#define KT(k) do { } while (0)
#endif
__device__ __forceinline__ void orbfe_kb8_unproject_dev(const float* __restrict__ P, float u, float v, float* ray)
{
    const float pwx = __fdiv_rn(__fsub_rn(u, P[2]), P[0]);
    const float pwy = __fdiv_rn(__fsub_rn(v, P[3]), P[1]);
    float scale = 1.f;
    // v_sqrt_f32 is only 1-ulp accurate: take the (correctly rounded) float sqrt through double
    float theta_d = (float)__dsqrt_rn((double)__fadd_rn(__fmul_rn(pwx, pwx), __fmul_rn(pwy, pwy)));
    const float hp = (float)(3.14159265358979323846 / 2.0);
    theta_d = fminf(fmaxf(-hp, theta_d), hp);
    if (theta_d > 1e-8) {
        float theta = theta_d;
        for (int j = 0; j < 10; j++) {
            const float t2 = __fmul_rn(theta, theta), t4 = __fmul_rn(t2, t2), t6 = __fmul_rn(t4, t2),
                        t8 = __fmul_rn(t4, t4);
            const float k0 = __fmul_rn(P[4], t2), k1 = __fmul_rn(P[5], t4), k2 = __fmul_rn(P[6], t6),
                        k3 = __fmul_rn(P[7], t8);
            const float num = __fsub_rn(
                __fmul_rn(theta, __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(1.f, k0), k1), k2), k3)), theta_d);
            const float den = __fadd_rn(
                __fadd_rn(__fadd_rn(__fadd_rn(1.f, __fmul_rn(3.f, k0)), __fmul_rn(5.f, k1)), __fmul_rn(7.f, k2)),
                __fmul_rn(9.f, k3));
            const float fix = __fdiv_rn(num, den);
            theta = __fsub_rn(theta, fix);
            if (fabsf(fix) < 1e-6f) break;
        }
        scale = __fdiv_rn((float)tan((double)theta), theta_d); // correctly rounded tan; host libm tanf is < 1 ulp
    }
    ray[0] = __fmul_rn(pwx, scale);
    ray[1] = __fmul_rn(pwy, scale);
    ray[2] = 1.f;
}

#ifdef ORBFE_SINCOS_H
/* KannalaBrandt8::project (KannalaBrandt8.cpp:25-41).  atan2f through double atan2 and cosf/sinf through the
 * correctly rounded routine: the host's libm values are within 1 ulp of these. */
__device__ __forceinline__ void orbfe_kb8_project_dev(const float* __restrict__ P, float x, float y, float z, float* u,
                                                      float* v)
{
    const float x2y2 = __fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y));
    const float theta = (float)atan2((double)(float)__dsqrt_rn((double)x2y2), (double)z);
    const float psi = (float)atan2((double)y, (double)x);
    const float t2 = __fmul_rn(theta, theta), t3 = __fmul_rn(theta, t2), t5 = __fmul_rn(t3, t2), t7 = __fmul_rn(t5, t2),
                t9 = __fmul_rn(t7, t2);
    const float r = __fadd_rn(
        __fadd_rn(__fadd_rn(__fadd_rn(theta, __fmul_rn(P[4], t3)), __fmul_rn(P[5], t5)), __fmul_rn(P[6], t7)),
        __fmul_rn(P[7], t9));
    float sn, cs;
    orbfe_sincos_cr(psi, &sn, &cs);
    *u = __fadd_rn(__fmul_rn(__fmul_rn(P[0], r), cs), P[2]);
    *v = __fadd_rn(__fmul_rn(__fmul_rn(P[1], r), sn), P[3]);
}

/* cv::SVD::compute on a 4x4 float matrix as Triangulate_ uses it (KannalaBrandt8.cpp:514-535): one-sided Jacobi
 * (OpenCV modules/core/src/lapack.cpp, JacobiSVDImpl_<float>), returns the right singular vector of the
 * smallest singular value (row 3 of vt).  Same operation order as the oracle's restatement; sqrt / hypot in
 * double. */
__device__ inline void orbfe_svd4_last_vt(const float* A, float* h)
{
    float At[16], Vt[16];
    double W[4];
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 4; k++) At[i * 4 + k] = A[k * 4 + i];
    const float eps = 2.384185791015625e-07f; // FLT_EPSILON * 2
    for (int i = 0; i < 4; i++) {
        double sd = 0;
        for (int k = 0; k < 4; k++) sd = __dadd_rn(sd, __dmul_rn((double)At[i * 4 + k], (double)At[i * 4 + k]));
        W[i] = sd;
        for (int k = 0; k < 4; k++) Vt[i * 4 + k] = (i == k) ? 1.f : 0.f;
    }
#ifndef ORBFE_SVD_MAXIT
#define ORBFE_SVD_MAXIT 30 /* OpenCV: max(m, 30) sweeps; other values are for timing experiments only (tools/r05_svd.sh) */
#endif
    // (Round 5, measured on tools/hostbench c5, whose matrices run all 30 sweeps -- sweeps 4..30 are 70 of the call's 300 us --
    // and not kept, each bit-identical on 180 000 triangulations, tools/kb8_ab.py: v_fma_f64 for the exact float x float products
    // and a v_sqrt_f64 filter in front of the convergence test's square root, -15 % instructions: no change; rotations on
    // disjoint rows -- (0,3) with (1,2), (2,3) with the next sweep's (0,1) -- as two branch-free chains: +50 us, because the
    // branch-free form pays hypot, two divisions and two square roots for every pair the `continue` below skips; leaving the loop
    // after a sweep that rotated and yet left At, Vt and W bit for bit as they were -- an exact shortcut: the next sweep would
    // repeat it -- no change either: the matrices that take all 30 sweeps keep moving in their last bits.)
    for (int iter = 0; iter < ORBFE_SVD_MAXIT; iter++) {
        bool changed = false;
        for (int i = 0; i < 3; i++)
            for (int j = i + 1; j < 4; j++) {
                float *Ai = At + i * 4, *Aj = At + j * 4;
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < 4; k++) p = __dadd_rn(p, __dmul_rn((double)Ai[k], (double)Aj[k]));
                if (fabs(p) <= __dmul_rn((double)eps, __dsqrt_rn(__dmul_rn(a, b)))) continue;
                p = __dmul_rn(p, 2.0);
                const double beta = __dadd_rn(a, -b), gamma = hypot(p, beta);
                float c, s;
                if (beta < 0) {
                    const double delta = __dmul_rn(__dadd_rn(gamma, -beta), 0.5);
                    s = (float)__dsqrt_rn(__ddiv_rn(delta, gamma));
                    c = (float)__ddiv_rn(p, __dmul_rn(__dmul_rn(gamma, (double)s), 2.0));
                } else {
                    c = (float)__dsqrt_rn(__ddiv_rn(__dadd_rn(gamma, beta), __dmul_rn(gamma, 2.0)));
                    s = (float)__ddiv_rn(p, __dmul_rn(__dmul_rn(gamma, (double)c), 2.0));
                }
                a = b = 0;
                for (int k = 0; k < 4; k++) {
                    const float t0 = __fadd_rn(__fmul_rn(c, Ai[k]), __fmul_rn(s, Aj[k]));
                    const float t1 = __fadd_rn(__fmul_rn(-s, Ai[k]), __fmul_rn(c, Aj[k]));
                    Ai[k] = t0;
                    Aj[k] = t1;
                    a = __dadd_rn(a, __dmul_rn((double)t0, (double)t0));
                    b = __dadd_rn(b, __dmul_rn((double)t1, (double)t1));
                }
                W[i] = a;
                W[j] = b;
                changed = true;
                float *Vi = Vt + i * 4, *Vj = Vt + j * 4;
                for (int k = 0; k < 4; k++) {
                    const float t0 = __fadd_rn(__fmul_rn(c, Vi[k]), __fmul_rn(s, Vj[k]));
                    const float t1 = __fadd_rn(__fmul_rn(-s, Vi[k]), __fmul_rn(c, Vj[k]));
                    Vi[k] = t0;
                    Vj[k] = t1;
                }
            }
        if (!changed) break;
    }
    for (int i = 0; i < 4; i++) {
        double sd = 0;
        for (int k = 0; k < 4; k++) sd = __dadd_rn(sd, __dmul_rn((double)At[i * 4 + k], (double)At[i * 4 + k]));
        W[i] = __dsqrt_rn(sd);
    }
    // the descending selection sort of the reference moves the row with the least W to position 3: simulate it.  (On scalars
    // with selects: indexing W / perm / Vt with a run-time index put them into scratch memory -- 80 bytes per thread and a
    // memory round trip per access at the end of every triangulation.)
    double w0 = W[0], w1 = W[1], w2 = W[2], w3 = W[3];
    int q0 = 0, q1 = 1, q2 = 2, q3 = 3;
    { // i = 0: j = first maximum among 0..3 by the scan `if (W[j] < W[k]) j = k`
        double mv = w0;
        int mi = 0, mq = q0;
        if (mv < w1) { mv = w1; mi = 1; mq = q1; }
        if (mv < w2) { mv = w2; mi = 2; mq = q2; }
        if (mv < w3) { mv = w3; mi = 3; mq = q3; }
        const double ow = w0;
        const int oq = q0;
        w0 = mv; q0 = mq;
        if (mi == 1) { w1 = ow; q1 = oq; }
        if (mi == 2) { w2 = ow; q2 = oq; }
        if (mi == 3) { w3 = ow; q3 = oq; }
    }
    { // i = 1
        double mv = w1;
        int mi = 1, mq = q1;
        if (mv < w2) { mv = w2; mi = 2; mq = q2; }
        if (mv < w3) { mv = w3; mi = 3; mq = q3; }
        const double ow = w1;
        const int oq = q1;
        w1 = mv; q1 = mq;
        if (mi == 2) { w2 = ow; q2 = oq; }
        if (mi == 3) { w3 = ow; q3 = oq; }
    }
    { // i = 2
        if (w2 < w3) {
            const int oq = q2;
            q2 = q3;
            q3 = oq;
        }
    }
    (void)q0; (void)q1; (void)q2;
#pragma unroll
    for (int k = 0; k < 4; k++) h[k] = q3 == 0 ? Vt[k] : q3 == 1 ? Vt[4 + k] : q3 == 2 ? Vt[8 + k] : Vt[12 + k];
}

/* KannalaBrandt8::TriangulateMatches_ (KannalaBrandt8.cpp:409-480): z1 of the triangulated point or -1. */
__device__ inline float orbfe_kb8_triangulate_dev(const float* __restrict__ P1, const float* __restrict__ P2, float k1x,
                                                  float k1y, float k2x, float k2y, const float* __restrict__ R12,
                                                  const float* __restrict__ t12, float sigmaLevel, float unc,
                                                  float* p3D = nullptr)
{
    float r1[3], r2[3], r21[3];
    KT_BEGIN();
    orbfe_kb8_unproject_dev(P1, k1x, k1y, r1);
    orbfe_kb8_unproject_dev(P2, k2x, k2y, r2);
    KT(0); // two unprojections
    for (int i = 0; i < 3; i++) {
        float s = 0.f;
        for (int k = 0; k < 3; k++) s = __fadd_rn(s, __fmul_rn(R12[i * 3 + k], r2[k]));
        r21[i] = s;
    }
    float dot = 0.f;
    double n1 = 0, n2 = 0;
    for (int i = 0; i < 3; i++) {
        dot = __fadd_rn(dot, __fmul_rn(r1[i], r21[i]));
        n1 = __dadd_rn(n1, __dmul_rn((double)r1[i], (double)r1[i]));
        n2 = __dadd_rn(n2, __dmul_rn((double)r21[i], (double)r21[i]));
    }
    const float cosParallax = (float)__ddiv_rn((double)dot, __dmul_rn(__dsqrt_rn(n1), __dsqrt_rn(n2)));
    if ((double)cosParallax > 0.9998) return -1.f;
    float R21[9], t21[3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R21[i * 3 + j] = R12[j * 3 + i];
    for (int i = 0; i < 3; i++) {
        float s = 0.f;
        for (int k = 0; k < 3; k++) s = __fadd_rn(s, __fmul_rn(__fmul_rn(R21[i * 3 + k], -1.f), t12[k]));
        t21[i] = s;
    }
    // rows of A = p.x * T.row(2) - T.row(0), p.y * T.row(2) - T.row(1) with Tcw1 = [I | 0] (last row zero)
    float A[16];
    const float T1[12] = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f};
    const float T2[12] = {R21[0], R21[1], R21[2], t21[0], R21[3], R21[4], R21[5], t21[1], R21[6], R21[7], R21[8], t21[2]};
    for (int k = 0; k < 4; k++) {
        A[0 * 4 + k] = __fsub_rn(__fmul_rn(r1[0], T1[8 + k]), T1[k]);
        A[1 * 4 + k] = __fsub_rn(__fmul_rn(r1[1], T1[8 + k]), T1[4 + k]);
        A[2 * 4 + k] = __fsub_rn(__fmul_rn(r2[0], T2[8 + k]), T2[k]);
        A[3 * 4 + k] = __fsub_rn(__fmul_rn(r2[1], T2[8 + k]), T2[4 + k]);
    }
    float h[4];
    KT(1); // parallax, A
    orbfe_svd4_last_vt(A, h);
    KT(2); // SVD
    const float inv = __fdiv_rn(1.f, h[3]);
    const float X[3] = {__fmul_rn(h[0], inv), __fmul_rn(h[1], inv), __fmul_rn(h[2], inv)};
    const float z1 = X[2];
    if (!(z1 > 0.f)) return -1.f; // also NaN
    float z2 = 0.f;
    for (int k = 0; k < 3; k++) z2 = __fadd_rn(z2, __fmul_rn(R21[6 + k], X[k]));
    z2 = __fadd_rn(z2, t21[2]);
    if (!(z2 > 0.f)) return -1.f;
    float u1, v1;
    orbfe_kb8_project_dev(P1, X[0], X[1], X[2], &u1, &v1);
    const float ex1 = __fsub_rn(u1, k1x), ey1 = __fsub_rn(v1, k1y);
    if ((double)__fadd_rn(__fmul_rn(ex1, ex1), __fmul_rn(ey1, ey1)) > __dmul_rn(5.991, (double)sigmaLevel)) return -1.f;
    float X2[3];
    for (int i = 0; i < 3; i++) {
        float s = 0.f;
        for (int k = 0; k < 3; k++) s = __fadd_rn(s, __fmul_rn(R21[i * 3 + k], X[k]));
        X2[i] = __fadd_rn(s, t21[i]);
    }
    float u2, v2;
    orbfe_kb8_project_dev(P2, X2[0], X2[1], X2[2], &u2, &v2);
    const float ex2 = __fsub_rn(u2, k2x), ey2 = __fsub_rn(v2, k2y);
    if ((double)__fadd_rn(__fmul_rn(ex2, ex2), __fmul_rn(ey2, ey2)) > __dmul_rn(5.991, (double)unc)) return -1.f;
    KT(3); // two projections
    if (p3D) {
        p3D[0] = X[0];
        p3D[1] = X[1];
        p3D[2] = X[2];
    }
    return z1;
}

/* KannalaBrandt8::matchAndtriangulate (KannalaBrandt8.cpp:244-335) with the cv::Mat Triangulate (:498-512): true and
 * the world point, or false.  T = rows 0..2 of the camera pose (3x4 row-major).  cv::Mat products and dots
 * accumulate in double and round once; the rows of A are float; `/ w` scales by (float)(1.0 / w). */
__device__ inline bool orbfe_kb8_match_triangulate_dev(const float* __restrict__ P1, const float* __restrict__ P2, float k1x,
                                                       float k1y, float k2x, float k2y, const float* __restrict__ T1,
                                                       const float* __restrict__ T2, float sigma1, float sigma2, float* X)
{
    float r1[3], r2[3], ray1[3], ray2[3];
    orbfe_kb8_unproject_dev(P1, k1x, k1y, r1);
    orbfe_kb8_unproject_dev(P2, k2x, k2y, r2);
    for (int i = 0; i < 3; i++) {
        double s1 = 0, s2 = 0;
        for (int k = 0; k < 3; k++) {
            s1 = __dadd_rn(s1, __dmul_rn((double)T1[k * 4 + i], (double)r1[k]));
            s2 = __dadd_rn(s2, __dmul_rn((double)T2[k * 4 + i], (double)r2[k]));
        }
        ray1[i] = (float)s1;
        ray2[i] = (float)s2;
    }
    double dot = 0, n1 = 0, n2 = 0;
    for (int i = 0; i < 3; i++) {
        dot = __dadd_rn(dot, __dmul_rn((double)ray1[i], (double)ray2[i]));
        n1 = __dadd_rn(n1, __dmul_rn((double)ray1[i], (double)ray1[i]));
        n2 = __dadd_rn(n2, __dmul_rn((double)ray2[i], (double)ray2[i]));
    }
    const float cosParallax = (float)__ddiv_rn(dot, __dmul_rn(__dsqrt_rn(n1), __dsqrt_rn(n2)));
    if ((double)cosParallax > 0.9998) return false;
    float A[16];
    for (int k = 0; k < 4; k++) {
        A[0 * 4 + k] = __fsub_rn(__fmul_rn(r1[0], T1[8 + k]), T1[k]);
        A[1 * 4 + k] = __fsub_rn(__fmul_rn(r1[1], T1[8 + k]), T1[4 + k]);
        A[2 * 4 + k] = __fsub_rn(__fmul_rn(r2[0], T2[8 + k]), T2[k]);
        A[3 * 4 + k] = __fsub_rn(__fmul_rn(r2[1], T2[8 + k]), T2[4 + k]);
    }
    float h[4];
    orbfe_svd4_last_vt(A, h);
    const float inv = (float)__ddiv_rn(1.0, (double)h[3]);
    const float Xw[3] = {__fmul_rn(h[0], inv), __fmul_rn(h[1], inv), __fmul_rn(h[2], inv)};
    const float* Ts[2] = {T1, T2};
    const float* Ps[2] = {P1, P2};
    const float kx[2] = {k1x, k2x}, ky[2] = {k1y, k2y}, sg[2] = {sigma1, sigma2};
    float xc[2][3];
    for (int c = 0; c < 2; c++) // both depths are tested before either reprojection (:295-306)
        for (int i = 0; i < 3; i++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s = __dadd_rn(s, __dmul_rn((double)Ts[c][i * 4 + k], (double)Xw[k]));
            xc[c][i] = (float)__dadd_rn(s, (double)Ts[c][i * 4 + 3]);
        }
    if (xc[0][2] <= 0.f || xc[1][2] <= 0.f) return false;
    for (int c = 0; c < 2; c++) {
        float u, v;
        orbfe_kb8_project_dev(Ps[c], xc[c][0], xc[c][1], xc[c][2], &u, &v);
        const float ex = __fsub_rn(u, kx[c]), ey = __fsub_rn(v, ky[c]);
        if ((double)__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)) > __dmul_rn(5.991, (double)sg[c])) return false;
    }
    X[0] = Xw[0];
    X[1] = Xw[1];
    X[2] = Xw[2];
    return true;
}
#endif /* ORBFE_SINCOS_H */
#endif
