/*
 * orbfe_sincos.h -- float sin/cos as a fixed sequence of IEEE double operations.
 *
 * The reference rotates the rBRIEF pattern with host libm cosf/sinf
 * (src/ORBextractor.cc:110-111); libm is neither available on the device nor
 * correctly rounded.  This routine is evaluated identically
 * on the device (K-DESC) and on the host (trig fix-up): every operation goes through
 * ORBFE_DMUL / ORBFE_DADD so that neither compiler can contract it into an FMA.
 * Its float result is the correctly rounded sin/cos of the float argument unless the
 * true value lies within ~1e-17 (relative) of a rounding boundary.
 */
#ifndef ORBFE_SINCOS_H
#define ORBFE_SINCOS_H

#if defined(__HIP_DEVICE_COMPILE__)
#define ORBFE_HD __host__ __device__
#define ORBFE_DMUL(a, b) __dmul_rn((a), (b))
#define ORBFE_DADD(a, b) __dadd_rn((a), (b))
#else
#if defined(__HIPCC__)
#define ORBFE_HD __host__ __device__
#else
#define ORBFE_HD
#endif
static inline double orbfe_dmul_host(double a, double b)
{
    volatile double r = a * b; /* volatile store: no contraction with a following add */
    return r;
}
#define ORBFE_DMUL(a, b) orbfe_dmul_host((a), (b))
#define ORBFE_DADD(a, b) ((a) + (b))
#endif

ORBFE_HD static inline void orbfe_sincos_cr(float angle, float* s_out, float* c_out)
{
    const double TWO_OVER_PI = 0.63661977236758134308;
    const double PIO2_HI = 1.57079632673412561417e+00;
    const double PIO2_LO = 6.07710050650619224932e-11;
    const double S1 = -1.0 / 6.0, S2 = 1.0 / 120.0, S3 = -1.0 / 5040.0, S4 = 1.0 / 362880.0,
                 S5 = -1.0 / 39916800.0, S6 = 1.0 / 6227020800.0, S7 = -1.0 / 1307674368000.0,
                 S8 = 1.0 / 355687428096000.0;
    const double C1 = -0.5, C2 = 1.0 / 24.0, C3 = -1.0 / 720.0, C4 = 1.0 / 40320.0, C5 = -1.0 / 3628800.0,
                 C6 = 1.0 / 479001600.0, C7 = -1.0 / 87178291200.0, C8 = 1.0 / 20922789888000.0;
    const double x = (double)angle;
    const double t = ORBFE_DADD(ORBFE_DMUL(x, TWO_OVER_PI), 0.5);
    long long ki = (long long)t;
    if ((double)ki > t) ki -= 1;
    const double kd = (double)ki;
    const double r = ORBFE_DADD(ORBFE_DADD(x, -ORBFE_DMUL(kd, PIO2_HI)), -ORBFE_DMUL(kd, PIO2_LO));
    const double r2 = ORBFE_DMUL(r, r);
    double ps = S8;
    ps = ORBFE_DADD(ORBFE_DMUL(ps, r2), S7);
    ps = ORBFE_DADD(ORBFE_DMUL(ps, r2), S6);
    ps = ORBFE_DADD(ORBFE_DMUL(ps, r2), S5);
    ps = ORBFE_DADD(ORBFE_DMUL(ps, r2), S4);
    ps = ORBFE_DADD(ORBFE_DMUL(ps, r2), S3);
    ps = ORBFE_DADD(ORBFE_DMUL(ps, r2), S2);
    ps = ORBFE_DADD(ORBFE_DMUL(ps, r2), S1);
    const double sr = ORBFE_DADD(r, ORBFE_DMUL(ORBFE_DMUL(r, r2), ps));
    double pc = C8;
    pc = ORBFE_DADD(ORBFE_DMUL(pc, r2), C7);
    pc = ORBFE_DADD(ORBFE_DMUL(pc, r2), C6);
    pc = ORBFE_DADD(ORBFE_DMUL(pc, r2), C5);
    pc = ORBFE_DADD(ORBFE_DMUL(pc, r2), C4);
    pc = ORBFE_DADD(ORBFE_DMUL(pc, r2), C3);
    pc = ORBFE_DADD(ORBFE_DMUL(pc, r2), C2);
    pc = ORBFE_DADD(ORBFE_DMUL(pc, r2), C1);
    const double cr = ORBFE_DADD(1.0, ORBFE_DMUL(r2, pc));
    double s, c;
    switch ((int)(ki & 3)) {
    case 0: s = sr; c = cr; break;
    case 1: s = cr; c = -sr; break;
    case 2: s = -sr; c = -cr; break;
    default: s = -cr; c = sr; break;
    }
    *s_out = (float)s;
    *c_out = (float)c;
}

#endif
