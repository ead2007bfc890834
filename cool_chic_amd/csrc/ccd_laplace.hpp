// ccd_laplace.hpp - the leaky quantised Laplace model's left cumulative (constriction 0.4.2 QuantizedLaplace(-64, 63),
// 24-bit precision) as BOTH entropy kernels evaluate it: f64 exp the way glibc does it + the host's RN(1 / b) table.  What must
// agree with the encoder's libm is the 24-bit integer boundary, and that is proven on the whole reachable domain (32 768 mu
// indices x 2 561 scale indices x 127 symbols = 1.0658e10 boundaries: tools/cdf_sweep.py, profiles/r03/cdf_sweep.log;
// tests/test_gpu_parity.py::test_laplace_boundaries_sweep keeps every 64th scale index in the GPU suite).
#pragma once

#include "ccd_device.hpp"

namespace ccd {

// exp(x) / 2 for x <= 0 in f64, the way glibc does it: x = (128 k' + j) ln2 / 128 + r with |r| <= ln2 / 256, e^x = 2^k' * T[j] *
// (1 + r + r^2/2 + .. + r^5/120) with T[j] = 2^(j/128) from a 1 KB table (LDS in the kernel), two-part ln2 / 128.  13 f64
// operations and a Horner chain of depth 4 instead of 21 and depth 7 for the table-free degree-13 version it replaced
// (-DCCD_EXP_POLY13 keeps that one for A/B).  Error ~1 ulp; what matters is floor(16777088 * cdf), and THAT is proven on the
// whole reachable domain: tools/cdf_sweep.py compares all 1.0658e10 boundaries with libm (profiles/r03/cdf_sweep.log).
#ifndef CCD_EXP_LOG
#define CCD_EXP_LOG 7
#endif
#ifndef CCD_EXP_DEG
#define CCD_EXP_DEG 5
#endif
#ifndef CCD_EXP_INC
#define CCD_EXP_INC "ccd_exp_table.inc"
#endif
static __device__ const double kExpTab[1 << CCD_EXP_LOG] = {  // (one copy per translation unit: the two kernels do not link device code)
#include CCD_EXP_INC
};
#ifdef CCD_EXP_POLY13
__device__ __forceinline__ double exp_nonpos(double x, const double*) {
    // branch-free on purpose: four of these chains are interleaved by the table builder
    const bool tiny = x < -60.0;  // below 2^-86: contributes nothing to a 24-bit cumulative, and 1 - e/2 == 1
    x = tiny ? -60.0 : x;
    const double k = rint(x * 1.44269504088896338700e+00);
    double r = fma(k, -6.93147180369123816490e-01, x);
    r = fma(k, -1.90821492927058770002e-10, r);
    // e^r = E(r^2) + r * O(r^2): two independent Horner chains of 7 instead of one of 14
    const double r2 = r * r;
    double pe = 1.1470745597729725e-11;           // 1/14!
    double po = 1.6059043836821613e-10;           // 1/13!
    pe = fma(pe, r2, 2.08767569878681e-09);       // 1/12!
    po = fma(po, r2, 2.505210838544172e-08);      // 1/11!
    pe = fma(pe, r2, 2.755731922398589e-07);      // 1/10!
    po = fma(po, r2, 2.7557319223985893e-06);     // 1/9!
    pe = fma(pe, r2, 2.48015873015873e-05);       // 1/8!
    po = fma(po, r2, 1.984126984126984e-04);      // 1/7!
    pe = fma(pe, r2, 1.388888888888889e-03);      // 1/6!
    po = fma(po, r2, 8.333333333333333e-03);      // 1/5!
    pe = fma(pe, r2, 4.1666666666666664e-02);     // 1/4!
    po = fma(po, r2, 1.6666666666666666e-01);     // 1/3!
    pe = fma(pe, r2, 0.5);                        // 1/2!
    po = fma(po, r2, 1.0);                        // 1/1!
    pe = fma(pe, r2, 1.0);                        // 1/0!
    const double e = ldexp(fma(po, r, pe), static_cast<int>(k) - 1);
    return tiny ? 0.0 : e;
}
#else
constexpr int kExpLog = CCD_EXP_LOG, kExpN = 1 << kExpLog;
__device__ __forceinline__ double exp_nonpos(double x, const double* tab /* 2^(j/N): LDS in the kernel */) {
    // No clamp for very negative x: kd stays finite, 2^(ki >> log N) underflows to an exact 0 in v_ldexp_f64 (and 1 - 0 == 1), which
    // is what the 24-bit cumulative needs; |x| <= 128 / min scale = 1.9e4 here, far from where ki could overflow.
    const double kd = rint(x * ldexp(0x1.71547652b82fep+0, kExpLog));            // N / ln 2
    double r = fma(kd, -ldexp(0x1.62e42fef00000p-1, -kExpLog), x);               // ln 2 / N, leading 33 bits: kd * hi is exact below 2^20
    r = fma(kd, -ldexp(0x1.473de6af278edp-34, -kExpLog), r);
    const int ki = static_cast<int>(kd);
    const double t = tab[ki & (kExpN - 1)];
    const double r2 = r * r;
#if CCD_EXP_DEG == 5
    double p = fma(r, 8.333333333333333e-03, 4.1666666666666664e-02);  // 1/5!, 1/4!
    p = fma(p, r, 1.6666666666666666e-01);
    p = fma(p, r, 0.5);
#elif CCD_EXP_DEG == 4
    double p = fma(r, 4.1666666666666664e-02, 1.6666666666666666e-01);
    p = fma(p, r, 0.5);
#elif CCD_EXP_DEG == 3
    double p = fma(r, 1.6666666666666666e-01, 0.5);
#else
    double p = 0.5;
#endif
    p = fma(p, r2, r);                                            // e^r - 1
    return ldexp(fma(t, p, t), (ki >> kExpLog) - 1);              // e^x / 2 (the caller's 0.5 *, folded into the exponent)
}
#endif

// Left cumulative of symbol s under (mu, b) with rcp = RN(1 / b) from the host.  The quotient (x - mu) / b is formed as
// (x - mu) * rcp WITHOUT the Newton step that would make it the correctly rounded quotient: with or without it, with or
// without a clamp of very negative arguments, all 1.0658e10 reachable boundaries equal libm's (tools/cdf_sweep.py) - the
// proof is the enumeration, not the error analysis.
__device__ __forceinline__ uint32_t window_left(double mu, double rcp, int s, const double* exp_tab) {
    const double x = static_cast<double>(s) - 0.5;
    const double d = x - mu;
    const double e = exp_nonpos(-fabs(d) * rcp, exp_tab);  // e^(-|x - mu| / b) / 2
    const double cdf = d <= 0.0 ? e : 1.0 - e;
    const uint32_t v = static_cast<uint32_t>(16777088.0 * cdf) + static_cast<uint32_t>(s - kAcLo);
    return s <= kAcLo ? 0u : (s > kAcLo + kAlphabet - 1 ? (1u << kRcPrecision) : v);
}

}  // namespace ccd
