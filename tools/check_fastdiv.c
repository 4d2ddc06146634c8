// Exhaustive check of div_small_int() (stereo_matching_cuda_amd/csrc/smx_agg_v3.hip): for EVERY box
// window area d (products of two integers in [1, 19]) and EVERY f32 significand and sign, at the
// exponents that matter (the extremes of the admitted range |x| in [2^-100, FLT_MAX] and the middle),
//      q = x*r;  e = fma(-q, d, x);  q' = fma(e, r, q)      with r = RN(1/d)
// equals the IEEE division x / d bit for bit.  Scaling x by a power of two scales every intermediate
// exactly as long as nothing under- or overflows, so the significands at a mid exponent cover the whole
// interior of the range; the lowest and highest binades are walked in full as well.
// Build + run (tests/test_capi.py does this): gcc -O2 -mfma -ffp-contract=off -fopenmp tools/check_fastdiv.c -lm
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline float fastdiv(float x, float d, float r) {
    float q = x * r;
    float e = fmaf(-q, d, x);
    return fmaf(e, r, q);
}

int main(void) {
    int areas[400], na = 0;
    for (int a = 1; a <= 19; ++a)
        for (int b = a; b <= 19; ++b) {
            int v = a * b, dup = 0;
            for (int i = 0; i < na; ++i) if (areas[i] == v) dup = 1;
            if (!dup) areas[na++] = v;
        }
    // biased exponents: 27 = 2^-100 (lowest admitted), 28, 127 (1.0), 150, 253, 254 (top binade, the quotient
    // cannot overflow because d >= 1)
    const uint32_t exps[] = {27, 28, 100, 127, 150, 200, 253, 254};
    const int ne = (int)(sizeof(exps) / sizeof(exps[0]));
    long bad = 0, total = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : bad, total)
    for (int i = 0; i < na; ++i) {
        const float d = (float)areas[i], r = 1.0f / d;
        for (int ei = 0; ei < ne; ++ei) {
            // full significand walk only at three exponents; a stride elsewhere keeps the run in seconds
            const uint32_t step = (exps[ei] == 27 || exps[ei] == 127 || exps[ei] == 254) ? 1u : 257u;
            for (uint32_t m = 0; m < (1u << 23); m += step)
                for (uint32_t sgn = 0; sgn < 2; ++sgn) {
                    const uint32_t u = (sgn << 31) | (exps[ei] << 23) | m;
                    float x;
                    memcpy(&x, &u, 4);
                    const float a = x / d, b = fastdiv(x, d, r);
                    uint32_t ua, ub;
                    memcpy(&ua, &a, 4);
                    memcpy(&ub, &b, 4);
                    ++total;
                    if (ua != ub) {
                        if (bad < 10) printf("bad x=%a d=%g : %a vs %a\n", x, d, a, b);
                        ++bad;
                    }
                }
        }
    }
    printf("areas %d total %ld bad %ld\n", na, total, bad);
    return bad != 0;
}
