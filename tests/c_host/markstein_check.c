/* markstein_check.c -- CPU check that the 3-op sequence the autocast kernels use for  y = idx / t2  is
 * bit-identical to the IEEE fp32 division, signed zeros included:
 *     r  = RN(1/t2)            (once per row)
 *     q0 = RN(idx * r) ; e = fma(-q0, t2, idx) ; q = (e == 0 || |q0| == inf) ? q0 : fma(e, r, q0)
 * (Markstein's theorem: with a correctly rounded reciprocal and a faithful q0, the correction step yields the
 * correctly rounded quotient.)  Exhaustive over idx in [-300, 300] (and wider 8-bit-significand values) for
 * tens of millions of divisors spread over the exponent range t2 can take.      gcc -O2 -mfma markstein_check.c -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline float mark_div(float a, float b, float r) {
    const float q0 = a * r;
    const float e = fmaf(-q0, b, a);
    return (e == 0.0f || fabsf(q0) == INFINITY) ? q0 : fmaf(e, r, q0);
}

int main(void) {
    uint64_t checked = 0, bad = 0;
    uint32_t s = 2463534242u;
    for (int rep = 0; rep < 60000; ++rep) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        /* divisors: exponent from 2^-24 .. 2^40 (t2 = s + 1e-6 lives in [1e-6, ~3.3e10]), random significand */
        const int ex = (int)(s % 65u) - 24;
        const float b = ldexpf(1.0f + (float)(s >> 9) * (1.0f / 8388608.0f), ex);
        volatile float rv = 1.0f / b;
        const float r = rv;
        for (int i = -300; i <= 300; ++i) {
            float a = (float)i;
            if (i == 0 && (rep & 1)) a = -0.0f;
            volatile float want = a / b;
            const float got = mark_div(a, b, r);
            ++checked;
            if (f2u(got) != f2u(want)) { if (bad++ < 5) printf("MISMATCH a=%g b=%a got=%a want=%a\n", a, b, got, (float)want); }
        }
        for (int j = 9; j <= 15; ++j) {          /* wider quantizers: indices m * 2^j, m <= 255 */
            const float a = ldexpf((float)(128 + (s >> (j + 3)) % 128), j - 7) * ((s >> 5) & 1 ? -1.f : 1.f);
            volatile float want = a / b;
            ++checked;
            if (f2u(mark_div(a, b, r)) != f2u((float)want)) { if (bad++ < 5) printf("MISMATCH wide a=%g b=%a\n", a, b); }
        }
    }
    /* all divisors of one binade, a fixed set of numerators */
    for (uint32_t m = 0; m < (1u << 23); m += 3) {
        const float b = u2f(0x3F800000u | m) * 64.0f;
        volatile float rv = 1.0f / b;
        for (int i = 1; i <= 127; i += 7) {
            volatile float want = (float)i / b;
            ++checked;
            if (f2u(mark_div((float)i, b, rv)) != f2u((float)want)) { if (bad++ < 5) printf("MISMATCH binade a=%d b=%a\n", i, b); }
        }
    }
    /* general numerators (AsymQuantizer's (x - min) / (alpha + 1e-8) and the fp16 chains): divisor anywhere in the
     * range the kernels accept for this sequence [2^-60, 2^100] -- incl. all-ones and power-of-two significands --,
     * numerator any fp32 with |a| >= 2^-101 (below that the correction term underflows; the kernels never depend on
     * such a quotient) and a finite quotient */
    for (int rep = 0; rep < 4000000; ++rep) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        const int eb = (int)(s % 161u) - 60;
        uint32_t mb = (s >> 9);
        if ((rep & 15) == 0) mb = 0x7FFFFFu;          /* 1.11...1 */
        else if ((rep & 15) == 1) mb = 0;              /* power of two */
        else if ((rep & 15) == 2) mb &= 0x7FE000u;     /* fp16-valued significand */
        const float b = ldexpf(1.0f + (float)mb * (1.0f / 8388608.0f), eb);
        volatile float rv = 1.0f / b;
        const float r = rv;
        for (int k = 0; k < 6; ++k) {
            s ^= s << 13; s ^= s >> 17; s ^= s << 5;
            int ea = eb - 40 + (int)(s % 45u);         /* quotient in [2^-41, 2^5) */
            if (ea < -101) ea = -101;
            if (ea > 126) ea = 126;
            uint32_t ma = (s >> 9);
            if (k == 4) ma &= 0x7FE000u;
            float a = ldexpf(1.0f + (float)ma * (1.0f / 8388608.0f), ea);
            if (s & 256u) a = -a;
            volatile float want = a / b;
            const float got = mark_div(a, b, r);
            ++checked;
            if (f2u(got) != f2u((float)want)) { if (bad++ < 5) printf("MISMATCH general a=%a b=%a got=%a want=%a\n", a, b, got, (float)want); }
        }
    }
    {   /* special numerators: +-inf (fp16 bins beyond 65504), NaN, +-0 */
        const float bs[4] = {1e-6f, 0.37f, 65535.0f, 3.3e10f};
        const float as[5] = {INFINITY, -INFINITY, NAN, 0.0f, -0.0f};
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 5; ++j) {
                volatile float rv = 1.0f / bs[i];
                volatile float want = as[j] / bs[i];
                const float got = mark_div(as[j], bs[i], rv);
                ++checked;
                const int same = (got != got) ? ((float)want != (float)want) : f2u(got) == f2u((float)want);
                if (!same) { if (bad++ < 5) printf("MISMATCH special a=%g b=%g got=%g want=%g\n", as[j], bs[i], got, (float)want); }
            }
    }
    /* divide by S = 2^bits - 1 (AsymQuantizer's .div(s)): every bin index for bits <= 12, sampled above */
    for (int bits = 2; bits <= 31; ++bits) {
        const float S = (float)((1ull << bits) - 1);
        volatile float rv = 1.0f / S;
        const uint64_t top = (1ull << bits) - 1, step = bits <= 12 ? 1 : (top / 4099u) | 1u;
        for (uint64_t i = 0; i <= top; i += step) {
            const float a = (float)i;
            volatile float want = a / S;
            ++checked;
            if (f2u(mark_div(a, S, rv)) != f2u((float)want)) { if (bad++ < 5) printf("MISMATCH divS bits=%d i=%llu\n", bits, (unsigned long long)i); }
        }
    }
    printf("markstein check: %llu quotients, %llu mismatches\n", (unsigned long long)checked, (unsigned long long)bad);
    return bad ? 1 : 0;
}
