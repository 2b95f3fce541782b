/*
 * orb_sincos_cr.h -- oracle copy of the "canonical" float sin/cos (TEST INFRASTRUCTURE).
 *
 * The reference rotates the rBRIEF pattern with libm cosf/sinf
 * (src/ORBextractor.cc:110-111).  glibc's cosf/sinf are not correctly rounded
 * (measured here: 1.3 % of angles differ by 1 ulp from the correctly rounded value)
 * and have CPU-dispatched FMA variants, so the reference's own trig is platform
 * dependent.  This routine is a fixed sequence of IEEE
 * double operations (no FMA, no libm) whose float result is the correctly rounded
 * sin/cos except when the true value lies within ~1e-17 (relative) of a rounding
 * boundary.  The product carries an identical copy (csrc/orbfe_sincos.h) used on
 * the device and in the host-side trig fix-up; both must be compiled without FP
 * contraction.
 */
#ifndef ORB_SINCOS_CR_H
#define ORB_SINCOS_CR_H

static inline void orb_sincos_cr_impl(float angle, float* s_out, float* c_out)
{
    const double TWO_OVER_PI = 0.63661977236758134308;
    const double PIO2_HI = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
    const double PIO2_LO = 6.07710050650619224932e-11; /* pi/2 - PIO2_HI         */
    const double S1 = -1.0 / 6.0, S2 = 1.0 / 120.0, S3 = -1.0 / 5040.0, S4 = 1.0 / 362880.0,
                 S5 = -1.0 / 39916800.0, S6 = 1.0 / 6227020800.0, S7 = -1.0 / 1307674368000.0,
                 S8 = 1.0 / 355687428096000.0;
    const double C1 = -0.5, C2 = 1.0 / 24.0, C3 = -1.0 / 720.0, C4 = 1.0 / 40320.0, C5 = -1.0 / 3628800.0,
                 C6 = 1.0 / 479001600.0, C7 = -1.0 / 87178291200.0, C8 = 1.0 / 20922789888000.0;
    double x = (double)angle;
    double t = x * TWO_OVER_PI + 0.5;
    /* floor() for |t| < 2^31 without libm */
    long long ki = (long long)t;
    if ((double)ki > t) ki -= 1;
    double kd = (double)ki;
    double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
    double r2 = r * r;
    double ps = S8;
    ps = ps * r2 + S7;
    ps = ps * r2 + S6;
    ps = ps * r2 + S5;
    ps = ps * r2 + S4;
    ps = ps * r2 + S3;
    ps = ps * r2 + S2;
    ps = ps * r2 + S1;
    double sr = r + (r * r2) * ps;
    double pc = C8;
    pc = pc * r2 + C7;
    pc = pc * r2 + C6;
    pc = pc * r2 + C5;
    pc = pc * r2 + C4;
    pc = pc * r2 + C3;
    pc = pc * r2 + C2;
    pc = pc * r2 + C1;
    double cr = 1.0 + r2 * pc;
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
