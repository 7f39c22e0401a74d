/* Exhaustive-style check of the three-operation x / 3.0 used by the engine's getPolygonCenter (fpe_kernels.hip::div3):
   q = x * c, r = fma(-3, q, x), y = fma(r, c, q) with c = RN(1/3) must equal x / 3.0 bit for bit for every x the
   engine sends through it (|x| in (1e-280, 1e300); zero, NaN and the extremes take the true division).
   Prints the number of mismatches.  Built and run by tests/test_cpu_abi_and_host.py. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static uint64_t s = 88172645463325252ull;
static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static int check(double x) {
    const double c = 0x1.5555555555555p-2;
    const double ax = fabs(x);
    if (!(ax > 1e-280 && ax < 1e300)) return 0;
    const double q = x * c, r = fma(-3.0, q, x), y = fma(r, c, q), z = x / 3.0;
    return memcmp(&y, &z, 8) != 0;
}
int main(void) {
    long bad = 0;
    if (0x1.5555555555555p-2 != 1.0 / 3.0) bad++;
    for (long i = 0; i < 20000000L; i++) {
        uint64_t b = rnd();
        double x;
        if (i & 1) b = (b & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(1023 - 40 + (rnd() % 80)) << 52);
        else b = (b & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(100 + (rnd() % 1800)) << 52);
        memcpy(&x, &b, 8);
        bad += check(x);
    }
    /* mantissa patterns near the rounding boundaries of the quotient: 3k, 3k +- 1 around powers of two, all-ones, ... */
    for (int e = -60; e <= 60; e++)
        for (uint64_t m = 0; m < 4096; m++) {
            const uint64_t pats[6] = {m, 0xFFFFFFFFFFFFFull - m, 0x8000000000000ull + m, 0x8000000000000ull - m,
                                      0x5555555555555ull + m, 0xAAAAAAAAAAAAAull - m};
            for (int k = 0; k < 6; k++) {
                uint64_t b = ((uint64_t)(1023 + e) << 52) | (pats[k] & 0xFFFFFFFFFFFFFull);
                double x;
                memcpy(&x, &b, 8);
                bad += check(x) + check(-x);
            }
        }
    printf("%ld\n", bad);
    return bad != 0;
}
