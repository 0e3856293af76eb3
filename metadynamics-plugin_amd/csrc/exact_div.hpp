// exact_div.hpp — a / b, CORRECTLY ROUNDED, for a divisor that is loop-invariant (a box length, a mesh dimension): the
// reciprocal y = RN(1 / b) is formed once on the host (an IEEE division), the device spends one multiply and four fused
// multiply-adds instead of the ~25-instruction expansion of a double division (v_div_scale x 2, v_rcp_f64, refinement,
// v_div_fmas, v_div_fixup).  mesh.hip::locate needs the reference's quotients bit for bit, twelve per particle.
//
// Markstein's theorem (P. Markstein, IBM J. Res. Dev. 34 (1990); Muller et al., Handbook of Floating-Point Arithmetic, ch. 5):
// with y = RN(1 / b) and q a FAITHFUL rounding of a / b, the residual r = a - b q is exact in one FMA and RN(q + r y) is a / b
// correctly rounded — provided the significand of b is not all ones, no underflow / overflow.  q0 = RN(a y) can be 1.5 ulp off
// (not always faithful), so one refinement makes it faithful first:
//     q0 = RN(a y);  r0 = fma(-b, q0, a);  q1 = fma(r0, y, q0)      (|q1 - a/b| < 1 ulp)
//     r1 = fma(-b, q1, a);                 q2 = fma(r1, y, q1)      (= RN(a / b))
// The host refuses the fast form for a divisor with an all-ones significand (exact_div_ok): the kernels then divide.
// Checked on the device against the hardware division: tools/probe_div.hip (2^32 random pairs per divisor class, 0 differences).
#pragma once

#include <cstdint>
#include <cstring>

struct ExactDivisor
    {
    double b, y;          // divisor and RN(1 / b)
    int fast;             // 1: the FMA form is proven for this divisor
    };

inline ExactDivisor make_exact_divisor(double b)
    {
    ExactDivisor d;
    d.b = b;
    d.y = 1.0 / b;
    uint64_t bits;
    std::memcpy(&bits, &b, sizeof(bits));
    const uint64_t frac = bits & 0x000fffffffffffffull;
    const unsigned int expo = (unsigned int)((bits >> 52) & 0x7ff);
    d.fast = (frac != 0x000fffffffffffffull && expo > 64 && expo < 1983) ? 1 : 0;     // not all ones; far from underflow / overflow
    return d;
    }

#ifdef __HIPCC__
__device__ __forceinline__ double exact_div(const double a, const ExactDivisor &d)
    {
    if (!d.fast) return a / d.b;                                     // (uniform)
    const double q0 = a * d.y;
    const double r0 = __builtin_fma(-d.b, q0, a);
    const double q1 = __builtin_fma(r0, d.y, q0);
    const double r1 = __builtin_fma(-d.b, q1, a);
    return __builtin_fma(r1, d.y, q1);
    }
#endif
