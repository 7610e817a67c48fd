// prl_device_math.h — the literal float64 arithmetic of PRLib's binarizers as device functions.
//
// These follow the reference source operation by operation (one IEEE rounding per written
// operation; the translation unit is compiled with -ffp-contract=off so hipcc never fuses a
// multiply-add).  Citations are to the PRLib tree; [upstream] marks OpenCV semantics taken from
// SURVEY.md Appendix B.
#pragma once

#include <hip/hip_runtime.h>

#include "prl_internal.h"

namespace prl_hip {

// cv::filter2D with the four non-zero taps (0,0)=+f (0,w-1)=-f (w-1,0)=-f (w-1,w-1)=+f visited in
// row-major order, accumulator starting at delta = 0  — binarizeSauvola.cpp:83-90 [upstream Filter2D].
__device__ __forceinline__ double box4_literal(double A, double B, double C, double D, double f)
{
    const double nf = -f;
    double s0 = 0.0;
    s0 += f * A;
    s0 += nf * B;
    s0 += nf * C;
    s0 += f * D;
    return s0;
}

// m.mul(m); q -= m2; cv::sqrt(q)  — binarizeSauvola.cpp:93,109-110.  sqrt(double) is correctly
// rounded on gfx950 (checked against the host in tests/test_gpu_math.py); negative -> NaN.
__device__ __forceinline__ double dev_from(double m, double q)
{
    const double m2 = m * m;
    const double v = q - m2;
    return sqrt(v);
}

// [upstream] cv::pow for Feng's only possible inputs r in {0, 1, NaN}  — binarizeFeng.cpp:126
__device__ __forceinline__ double feng_pow(double r, double gamma)
{
    if (gamma == 2.0) return r * r;
    if (gamma == 0.0) return 1.0;
    if (r == 0.0) return gamma > 0.0 ? 0.0 : __builtin_inf();
    return r;  // 1 -> 1, NaN -> NaN
}

// Threshold before the u8 cast, from the literal mean m and deviation s.
__device__ __forceinline__ double threshold_literal(const ThrParams& tp, double m, double s,
                                                    double imin, double coeff)
{
    switch (tp.method) {
    case PRL_SAUVOLA: {
        // s.convertTo(s, f64, k*RBack, 1-k) ; T = m.mul(s)       binarizeSauvola.cpp:115-118
        const double d = s * tp.a + tp.b;
        return m * d;
    }
    case PRL_NIBLACK:
        // localMeanValues + k * localDevianceValues               binarizeNiblack.cpp:108
        // [upstream] lowers to scaleAdd(s, k, m) = s*k + m
        return s * tp.k + m;
    case PRL_WOLFJOLION: {
        // s.convertTo(s, f64, coeff, -k); s = s.mul(m - imageMin); T = m + s   binarizeWolfJolion.cpp:128-130
        const double d = s * coeff + (-tp.k);
        const double e = m * 1.0 + (-imin);
        const double g = d * e;
        return m + g;
    }
    case PRL_NICK: {
        // C = m.mul(m); s = s.mul(s); C = C + s; sqrt(C); addWeighted(m,1,C,k,0)  binarizeNICK.cpp:121-126
        double C = m * m;
        const double s2 = s * s;
        C = C + s2;
        C = sqrt(C);
        return (m * 1.0 + C * tp.k) + 0.0;
    }
    case PRL_FENG: {
        // binarizeFeng.cpp:118-142 with Rs aliasing s; [upstream <=3.x] divide: x/0 -> 0
        const double r = (s != 0.0) ? (s / s) : 0.0;
        const double r2 = feng_pow(r, tp.gamma);
        const double a3 = r2 * tp.k2 + 0.0;
        const double c2 = r2 * r;
        const double c3 = (a3 * imin + c2 * (-imin)) + 0.0;
        double T = c2 * 1.0 + tp.c1;
        T = T * m;
        T = T + c3;
        return T;
    }
    default:
        return __builtin_nan("");
    }
}

// saturate_cast<uchar>(double) of convertTo(CV_8UC1), binarizeSauvola.cpp:119:
// cvRound = round-half-even; NaN, +-inf and values outside int32 become INT_MIN -> 0.
__device__ __forceinline__ unsigned sat_u8_literal(double T)
{
    if (T != T) return 0u;
    const double r = rint(T);
    if (!(r >= -2147483648.0 && r <= 2147483647.0)) return 0u;
    return r < 0.0 ? 0u : (r > 255.0 ? 255u : (unsigned)r);
}

// in(rect) > T8  — binarizeSauvola.cpp:122
__device__ __forceinline__ unsigned char decide_literal(unsigned p, double T)
{
    return (p > sat_u8_literal(T)) ? 255 : 0;
}

}  // namespace prl_hip
