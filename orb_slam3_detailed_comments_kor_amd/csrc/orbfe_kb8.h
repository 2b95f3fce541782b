/*
 * orbfe_kb8.h -- KannalaBrandt8::unproject (reference src/CameraModels/KannalaBrandt8.cpp:96-123) on the device.
 * P = fx, fy, cx, cy, k0..k3.  Every float operation is separately rounded like the reference's scalar
 * code; sqrt goes through double (v_sqrt_f32 is 1-ulp) and tan through double (no libm on the device:
 * the only difference to the host is tanf's < 1 ulp, see tests/test_gpu_matcher.py::test_kb8_unproject).
 */
#ifndef ORBFE_KB8_H
#define ORBFE_KB8_H
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
#endif
