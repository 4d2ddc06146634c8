// Checks that x*r + fma residual correction == IEEE x/d bit for bit for every box-window area d
// (products of two integers in [1,19]) and |x| >= 2^-100.  Build: gcc -O2 -mfma -ffp-contract=off
// tools/check_fastdiv.c -lm ; run: ./a.out 40000000  (samples per area).  Used to justify
// div_small_int() in stereo_matching_cuda_amd/csrc/smx_agg_v2.hip.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
static inline float fastdiv(float x, float d, float r) {
    float q = x * r;
    float e = fmaf(-q, d, x);
    return fmaf(e, r, q);
}
static uint64_t s = 88172645463325252ULL;
static inline uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main(int argc, char** argv) {
    long bad = 0, total = 0;
    int areas[400], na = 0;
    for (int a = 1; a <= 19; ++a) for (int b = a; b <= 19; ++b) { int v = a*b, dup = 0; for (int i = 0; i < na; ++i) if (areas[i]==v) dup=1; if(!dup) areas[na++]=v; }
    long per = atol(argv[1]);
    for (int i = 0; i < na; ++i) {
        float d = (float)areas[i], r = 1.0f / d;
        for (long k = 0; k < per; ++k) {
            uint32_t u = (uint32_t)rnd();
            // exponent range: keep |x| in [2^-60, 2^60] mostly, plus some full-range
            if (k & 7) { uint32_t e = 67 + (rnd() % 120); u = (u & 0x807FFFFFu) | (e << 23); }
            float x; memcpy(&x, &u, 4);
            if (!isfinite(x)) continue;
            float a = x / d, b = fastdiv(x, d, r);
            uint32_t ua, ub; memcpy(&ua,&a,4); memcpy(&ub,&b,4);
            total++;
            if (ua != ub && !(a == 0 && b == 0) && fabsf(x) >= 0x1p-100f) { if (bad < 10) printf("bad x=%a d=%g : %a vs %a\n", x, d, a, b); bad++; }
        }
    }
    printf("areas %d total %ld bad %ld\n", na, total, bad);
    return 0;
}
