/* The rolling-mean kernel (csrc/sk_prep.hip, k_roll_one) forms t = RN(S / w) for an integer window sum |S| < 2^31 and an
 * integer window 1 <= w < 65 536 as   q = S * RN(1/w);  r = fma(-q, w, S);  t = fma(r, RN(1/w), q)   (three FP64
 * operations instead of the division's thirty).  Why that is the correctly rounded quotient: r is exact (a multiple of
 * ulp(q) below 2^18 ulp(q)); q + r RN(1/w) differs from S / w by at most 2^-52 ulp; and S / w cannot lie within
 * ulp / (2 w) >= 2^-17 ulp of a rounding boundary without being one (numerator of the difference is a non-zero integer),
 * which a 31-bit S over a 16-bit w never is.  This program checks the claim against the C division: every w, with
 * S = random, S near multiples of w, powers of two and the extremes.      gcc -O2 -o check_intdiv check_intdiv.c -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

static inline double quick(double s, double dw, double inv)
{
    const double q = s * inv;
    const double r = fma(-q, dw, s);
    return fma(r, inv, q);
}

int main(int argc, char **argv)
{
    const long per_w = argc > 1 ? atol(argv[1]) : 20000;
    uint64_t x = 88172645463325252ull, bad = 0, n = 0;
    for (int w = 1; w < 65536; w++) {
        const double dw = (double)w, inv = 1.0 / dw;
        for (long k = 0; k < per_w; k++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            int64_t s;
            switch (k & 7) {
            case 0: s = (int32_t)(x >> 32); break;
            case 1: s = (int64_t)w * (int32_t)((x >> 40) % 32768) + (int)(x % 5) - 2; break;
            case 2: s = ((int64_t)1 << (x % 31)) + (int)((x >> 8) % 7) - 3; break;
            case 3: s = (int64_t)((x >> 20) % (600ull * w + 1)); break;          /* what a window of samples ~ 600 gives */
            case 4: s = -(int64_t)((x >> 20) % (32768ull * w + 1)); break;
            case 5: s = 2147483647ll - (int64_t)(x % 4096); break;
            case 6: s = -2147483648ll + (int64_t)(x % 4096); break;
            default: s = (int64_t)(x % 4096) - 2048; break;
            }
            if (s > 2147483647ll) s = 2147483647ll;
            if (s < -2147483648ll) s = -2147483648ll;
            const double d = (double)s;
            n++;
            if (quick(d, dw, inv) != d / dw) { if (bad++ < 10) printf("MISMATCH S=%lld w=%d\n", (long long)s, w); }
        }
    }
    printf("%llu quotients, %llu mismatches\n", (unsigned long long)n, (unsigned long long)bad);
    return bad != 0;
}
