/*
 * smx_oracle.c -- CPU restatement of the stereo-pair -> disparity-map path of
 * hamza1030/stereo_matching_cuda.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product (the HIP library in
 * stereo_matching_cuda_amd/csrc) never links, loads or falls back to it.
 *
 * Pinning: the restatement is checked (tests/test_oracle_golden.py) against the
 * 12 output images the reference's authors committed next to their two Tsukuba
 * inputs (reference stereo_matching_cuda/data/ *.png, copied as data fixtures to
 * tests/golden/), and against the sha256 manifest of raw f32/u8 dumps recorded in
 * SURVEY.md Appendix C.  The reference itself is CUDA-only (needs cuda_runtime.h
 * and nvcc, neither in this image), so no oracle/_ref build exists.
 *
 * Arithmetic contract: IEEE f32, every operation rounded
 * individually in source order, no FMA contraction -> build with
 *   gcc -O2 -ffp-contract=off   (never -ffast-math)
 *
 * All sizes are 64-bit safe and D (number of disparity slices) is a runtime
 * parameter; the reference fixes it at compile time (SystemIncludes.h:11-12).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/stereo_matching_cuda/).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ---- constants: SystemIncludes.h:7-24 --------------------------------- */
typedef struct orc_params {
    double r_w, g_w, b_w; /* R_W 0.299, G_W 0.587, B_W 0.0721 (sic)      :7-9  */
    double alpha;         /* ALPHA 0.9                                    :10   */
    int th_color;         /* TH_color 7                                   :14   */
    int th_grad;          /* TH_grad 2                                    :13   */
    int radius;           /* RADIUS 9                                     :21   */
    double eps;           /* EPS 6.5025                                   :23   */
    int d_lr;             /* D_LR 0                                       :24   */
} orc_params;

ORC_API void orc_default_params(orc_params* p) {
    p->r_w = 0.299; p->g_w = 0.587; p->b_w = 0.0721;
    p->alpha = 0.9; p->th_color = 7; p->th_grad = 2;
    p->radius = 9; p->eps = 6.5025; p->d_lr = 0;
}

/* ---- rgb_to_grayscale.cu:14-23 (sumArraysOnGPU) -------------------------
 * double val = R_W*r + G_W*g + B_W*b (left to right, double), truncation. */
ORC_API void orc_gray(const orc_params* P, const uint8_t* rgb, int64_t n, int ch,
                      uint8_t* gray) {
    for (int64_t k = 0; k < n; ++k) {
        const uint8_t* px = rgb + (int64_t)ch * k;
        double val = P->r_w * px[0] + P->g_w * px[1] + P->b_w * px[2];
        gray[k] = (uint8_t)val;
    }
}

/* ---- costVolume.cu:358-381 (x_derivativeOnGPU) ---------------------------
 * out = (in[x-1] - in[x+1]) / 2 ; one-sided at both row ends. */
ORC_API void orc_xderiv(const uint8_t* in, float* out, int w, int h) {
    for (int y = 0; y < h; ++y) {
        const uint8_t* r = in + (int64_t)y * w;
        float* o = out + (int64_t)y * w;
        for (int x = 0; x < w; ++x) {
            float c1 = 0, c2 = 0;
            if (x - 1 >= 0 && x + 1 < w) { c1 = (int)r[x + 1]; c2 = (int)r[x - 1]; }
            else if (x + 1 >= w)         { c1 = (int)r[x];     c2 = (int)r[x - 1]; }
            else if (x - 1 <= -1)        { c1 = (int)r[x + 1]; c2 = (int)r[x];     }
            o[x] = 1.0f * (c2 - c1) / 2;
        }
    }
}

/* ---- costVolume.cu:163-190 (costVolumOnGPU2) -----------------------------
 * One slice z (d = dmin + z) of the truncated SAD + gradient cost.
 * i1/g1 have width w1, i2/g2 width w2, both height h. */
ORC_API void orc_cost_slice(const orc_params* P, const uint8_t* i1, const uint8_t* i2,
                            const float* g1, const float* g2, float* cost_slice,
                            int w1, int w2, int h, int d) {
    const float alpha = 1.0f * P->alpha;      /* costVolume.cu:169 */
    const float th_color = 1.0f * P->th_color;
    const float th_grad = 1.0f * P->th_grad;
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w1; ++x) {
            int64_t id1 = (int64_t)y * w1 + x;
            /* the reference indexes image 2 as i2[x_lin + d] with x_lin the linear
             * index in image 1 (costVolume.cu:186); identical to row-wise indexing
             * when w1 == w2, which is the only case main.cu exercises. */
            int64_t id2 = (int64_t)y * w1 + x + d;
            float c = (1 - alpha) * th_color + 1.0f * alpha * th_grad; /* :184 */
            if (((x + d) < w2) && ((x + d) >= 0)) {
                float t1 = 1.0f * (abs((int)i1[id1] - (int)i2[id2]));
                float t2 = 1.0f * fabsf(g1[id1] - g2[id2]);
                float m1 = t1 < th_color ? t1 : th_color;
                float m2 = t2 < th_grad ? t2 : th_grad;
                float a = (1.0f - alpha) * m1;
                float b = alpha * m2;
                c = a + b;                                             /* :187 */
            }
            cost_slice[id1] = c;
        }
    }
}

/* compute_cost wrapper (costVolume.cu:4-84) with runtime D. volume [z][y][x]. */
ORC_API void orc_cost_volume(const orc_params* P, const uint8_t* i1, const uint8_t* i2,
                             float* cost, int w1, int w2, int h, int size_d, int dmin) {
    int64_t n1 = (int64_t)w1 * h, n2 = (int64_t)w2 * h;
    float* g1 = (float*)malloc(sizeof(float) * n1);
    float* g2 = (float*)malloc(sizeof(float) * n2);
    orc_xderiv(i1, g1, w1, h);
    orc_xderiv(i2, g2, w2, h);
    for (int z = 0; z < size_d; ++z)
        orc_cost_slice(P, i1, i2, g1, g2, cost + (int64_t)z * n1, w1, w2, h, dmin + z);
    free(g1); free(g2);
}

/* ---- integral.cu:78-90 (rowSum) + :121-131 (colSum) ----------------------
 * Sequential f32 prefix sums, x first then y.  Order is part of the contract. */
ORC_API void orc_integral(const float* in, float* out, int w, int h) {
    for (int y = 0; y < h; ++y) {
        const float* r = in + (int64_t)y * w;
        float* o = out + (int64_t)y * w;
        o[0] = r[0];
        for (int x = 1; x < w; ++x) o[x] = r[x] + o[x - 1];
    }
    for (int y = 1; y < h; ++y) {
        float* o = out + (int64_t)y * w;
        const float* up = out + (int64_t)(y - 1) * w;
        for (int x = 0; x < w; ++x) o[x] = o[x] + up[x];
    }
}

/* ---- guidedFilter.cu:305-318 (computeMeanOnGPU) --------------------------- */
static inline float box_at(const float* S, int x, int y, int w, int h, int R) {
    int ymin = y - R - 1 > -1 ? y - R - 1 : -1;
    int ymax = y + R < h - 1 ? y + R : h - 1;
    int xmin = x - R - 1 > -1 ? x - R - 1 : -1;
    int xmax = x + R < w - 1 ? x + R : w - 1;
    float val = S[(int64_t)ymax * w + xmax];
    if (xmin >= 0) val -= S[(int64_t)ymax * w + xmin];
    if (ymin >= 0) val -= S[(int64_t)ymin * w + xmax];
    if (xmin >= 0 && ymin >= 0) val += S[(int64_t)ymin * w + xmin];
    return (1.0f * val / ((xmax - xmin) * (ymax - ymin)));
}

ORC_API void orc_box_mean(const float* S, float* mean, int w, int h, int radius) {
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            mean[(int64_t)y * w + x] = box_at(S, x, y, w, h, radius);
}

/* ---- guidance statistics: guidedFilter.cu:58-123 --------------------------
 * im = float(I); mean_I = box(integral(im)); mean u8 = trunc/clamp(mean_I);
 * var = box(integral(im*im)) - mean_I*mean_I.  */
ORC_API void orc_guidance(const orc_params* P, const uint8_t* I, float* im, float* mean_im,
                          float* var_im, uint8_t* mean_u8, int w, int h) {
    int64_t n = (int64_t)w * h;
    float* S = (float*)malloc(sizeof(float) * n);
    float* sq = (float*)malloc(sizeof(float) * n);
    for (int64_t k = 0; k < n; ++k) { int c = (int)I[k]; im[k] = 1.0f * c; } /* :442-449 */
    orc_integral(im, S, w, h);
    orc_box_mean(S, mean_im, w, h, P->radius);
    if (mean_u8)
        for (int64_t k = 0; k < n; ++k) {                                     /* :451-458 */
            int c = mean_im[k];
            mean_u8[k] = (c > 255) ? 255 : (uint8_t)c;
        }
    for (int64_t k = 0; k < n; ++k) sq[k] = im[k] * im[k];                    /* :111 */
    orc_integral(sq, S, w, h);
    orc_box_mean(S, sq, w, h, P->radius);                                     /* d_temp */
    for (int64_t k = 0; k < n; ++k) {
        float m2 = mean_im[k] * mean_im[k];                                   /* :112 */
        var_im[k] = sq[k] - m2;                                               /* :121 */
    }
    free(S); free(sq);
}

/* ---- one slice of the guided filter: guidedFilter.cu:198-233 -------------- */
static void agg_slice(const orc_params* P, const float* im, const float* mean_im,
                      const float* var_im, const float* p, float* q, float* t0, float* t1,
                      float* t2, float* t3, int w, int h) {
    int64_t n = (int64_t)w * h;
    const int R = P->radius;
    float* S = t0; float* mp = t1; float* mIp = t2; float* a = t2; float* b = t1;
    /* mean(p) */
    orc_integral(p, S, w, h);
    orc_box_mean(S, mp, w, h, R);
    /* mean(I*p) */
    for (int64_t k = 0; k < n; ++k) t3[k] = im[k] * p[k];                     /* :209 */
    orc_integral(t3, S, w, h);
    orc_box_mean(S, mIp, w, h, R);
    /* a_k, b_k: compute_ak_and_bk guidedFilter.cu:345-354
     *   c = 1.0f / (var + EPS)   -- EPS is a double literal: double add+div, then
     *                                rounded to float on assignment            */
    for (int64_t k = 0; k < n; ++k) {
        float c = 1.0f / (var_im[k] + P->eps);
        float mm = mean_im[k] * mp[k];
        float ak = 1.0f * (mIp[k] - mm) * c;
        float mb = 1.0f * mean_im[k] * ak;
        float bk = 1.0f * mp[k] - mb;
        a[k] = ak; b[k] = bk;
    }
    /* mean(a), mean(b) */
    orc_integral(a, S, w, h);
    orc_box_mean(S, t3, w, h, R);            /* abar -> t3 */
    orc_integral(b, S, w, h);
    orc_box_mean(S, a, w, h, R);             /* bbar -> t2 (a no longer needed) */
    /* q = abar*I + bbar : compute_q guidedFilter.cu:363-369 */
    for (int64_t k = 0; k < n; ++k) {
        float m = t3[k] * im[k];
        q[k] = m + a[k];
    }
}

/* ---- compute_guided_filter: guidedFilter.cu:4-295 -------------------------
 * Slices [s_begin, s_end) of `cost` (volume [z][y][x] holding size_d slices,
 * indexed by absolute slice number) are aggregated and folded into the running
 * WTA (dispSelectOnGPU guidedFilter.cu:403-411: `best >= q` -> later slice wins
 * ties).  best/dmap are in/out exactly as in the reference (main.cu:112-118 sets
 * them to 0x7F7F7F7F / 0).  agg (optional) receives q for the processed slices at
 * agg[(s - s_begin)*n].  mean_u8 optional. */
ORC_API void orc_guided_filter(const orc_params* P, const uint8_t* I, const float* cost,
                               float* best, float* dmap, uint8_t* mean_u8, float* agg,
                               int w, int h, int dmin, int s_begin, int s_end) {
    int64_t n = (int64_t)w * h;
    float* buf = (float*)malloc(sizeof(float) * n * 8);
    float *im = buf, *mean_im = buf + n, *var_im = buf + 2 * n, *q = buf + 3 * n;
    float *t0 = buf + 4 * n, *t1 = buf + 5 * n, *t2 = buf + 6 * n, *t3 = buf + 7 * n;
    orc_guidance(P, I, im, mean_im, var_im, mean_u8, w, h);
    for (int s = s_begin; s < s_end; ++s) {
        const float* p = cost + (int64_t)s * n;                               /* :198 */
        float* qs = agg ? agg + (int64_t)(s - s_begin) * n : q;
        agg_slice(P, im, mean_im, var_im, p, qs, t0, t1, t2, t3, w, h);
        int label = dmin + s;                                                 /* :234 */
        for (int64_t k = 0; k < n; ++k) {
            if (1.0f * best[k] >= 1.0f * qs[k]) { dmap[k] = label; best[k] = qs[k]; }
        }
    }
    free(buf);
}

/* main.cu:112-118: memset(best, 9999999.0f, ...) -> every byte 0x7F. */
ORC_API void orc_init_wta(float* best, float* dmap, int64_t n) {
    memset(best, 0x7F, sizeof(float) * n);
    memset(dmap, 0, sizeof(float) * n);
}

/* ---- occlusion.cu:3-15 (detect_occlusionOnGPU) ---------------------------- */
ORC_API void orc_detect_occlusion(const orc_params* P, float* dL, const float* dR,
                                  int dOcclusion, int w, int h) {
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int64_t id = (int64_t)y * w + x;
            int d = (int)dL[id];
            if (x + d < 0 || x + d >= w || fabsf(d + dR[id + d]) > P->d_lr)
                dL[id] = dOcclusion;
        }
}

/* ---- occlusion.cu:134-176 (fill_occlusionOnGPU1) --------------------------
 * Snapshot semantics (the in-place race of the reference is benign: an already
 * filled neighbour holds max(L,R) of the same run). */
ORC_API void orc_fill_occlusion(float* disp, int w, int h, float vMin) {
    float* src = (float*)malloc(sizeof(float) * w);
    for (int y = 0; y < h; ++y) {
        float* row = disp + (int64_t)y * w;
        memcpy(src, row, sizeof(float) * w);
        for (int x = 0; x < w; ++x) {
            int dX = src[x];
            if (dX >= vMin) continue;
            float dLeft = vMin, dRight = vMin;
            for (int xl = x; xl >= 0; --xl)
                if (src[xl] >= vMin) { dLeft = src[xl]; break; }
            for (int xr = x; xr < w; ++xr)
                if (src[xr] >= vMin) { dRight = src[xr]; break; }
            row[x] = dLeft > dRight ? dLeft : dRight;
        }
    }
    free(src);
}

/* ---- main.cu:13-35 (write_mat) : min/max normalisation to u8 --------------
 * Note the `else if`: min is only updated by elements that did not raise max. */
ORC_API void orc_write_mat_u8(const float* mat, uint8_t* out, int64_t n) {
    float max = -150000000.0f, min = 150000000.0f;
    for (int64_t i = 0; i < n; ++i) {
        if (mat[i] > max) max = mat[i];
        else if (mat[i] <= min) min = mat[i];
    }
    for (int64_t i = 0; i < n; ++i) {
        int c = (mat[i] - min) * 255.0f / (max - min);
        out[i] = (uint8_t)c;
    }
}

/* ---- the whole pair path: main.cu:65-155 ----------------------------------
 * Left volume labels dminl..dminl+D-1 (dminl = -(D-1)-dmax_off ... main.cu:79),
 * right volume labels dminr.. ; outputs are optional (NULL to skip). */
ORC_API void orc_stereo_pair(const orc_params* P, const uint8_t* Il, const uint8_t* Ir,
                             int w, int h, int size_d, int dminl, int dminr,
                             float* costl_out, float* costr_out, float* aggl_out,
                             float* aggr_out, float* bestl, float* bestr, float* dmapl,
                             float* dmapr, uint8_t* meanl, uint8_t* meanr, float* occl,
                             float* filled) {
    int64_t n = (int64_t)w * h;
    float* costl = costl_out ? costl_out : (float*)malloc(sizeof(float) * n * size_d);
    float* costr = costr_out ? costr_out : (float*)malloc(sizeof(float) * n * size_d);
    orc_cost_volume(P, Il, Ir, costl, w, w, h, size_d, dminl);               /* main.cu:80 */
    orc_cost_volume(P, Ir, Il, costr, w, w, h, size_d, dminr);               /* main.cu:82 */
    orc_init_wta(bestl, dmapl, n);
    orc_init_wta(bestr, dmapr, n);
    orc_guided_filter(P, Il, costl, bestl, dmapl, meanl, aggl_out, w, h, dminl, 0, size_d);
    orc_guided_filter(P, Ir, costr, bestr, dmapr, meanr, aggr_out, w, h, dminr, 0, size_d);
    if (occl) {
        memcpy(occl, dmapl, sizeof(float) * n);                              /* main.cu:141 */
        orc_detect_occlusion(P, occl, dmapr, dminl - 100, w, h);             /* main.cu:149-150 */
        if (filled) {
            memcpy(filled, occl, sizeof(float) * n);                         /* main.cu:153 */
            orc_fill_occlusion(filled, w, h, (float)dminl);                  /* main.cu:154-155 */
        }
    }
    if (!costl_out) free(costl);
    if (!costr_out) free(costr);
}

/* ---- filter.cu:117-207 (filter(): dead code in the reference) ---------------
 * Direct zero-padded (2R+1)^2 box sums in f32, x offset outer / y offset inner
 * (filter.cu:57-61), truncated means (:63-64), var = mean(I*I) - mean*mean with
 * the u8 mean (:143-181). */
ORC_API void orc_filter(const orc_params* P, const uint8_t* I, uint8_t* mean, float* var,
                        int w, int h) {
    const int R = P->radius;
    const int area = (2 * R + 1) * (2 * R + 1);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float s1 = 0.0f, s2 = 0.0f;
            for (int ix = -R; ix <= R; ++ix)
                for (int iy = -R; iy <= R; ++iy) {
                    int xx = x + ix, yy = y + iy;
                    float v = 0.0f, v2 = 0.0f;
                    if (xx >= 0 && xx < w && yy >= 0 && yy < h) {
                        int c = (int)I[(int64_t)yy * w + xx];
                        v = (float)c;
                        v2 = (float)(c * c);
                    }
                    s1 += v;
                    s2 += v2;
                }
            int m = (int)(s1 / area);
            int m2 = (int)(s2 / area);
            uint8_t mu = (uint8_t)m;
            mean[(int64_t)y * w + x] = mu;
            var[(int64_t)y * w + x] = (float)m2 - (float)((int)mu * (int)mu);
        }
}

/* ---- packed WTA key (no reference counterpart) --------------
 * key = (sord(best) << 32) | (0xFFFFFFFF - slice) compared as SIGNED 64-bit
 * integers; min key == min cost and, on equal cost, the LARGEST slice, i.e. the
 * sequential `>=` rule of dispSelect.  sord: order-preserving f32 -> i32 map
 * (-0 canonicalised to +0).  Identity (and NaN, which never wins) = INT64_MAX. */
ORC_API int64_t orc_pack_key(float best, uint32_t slice) {
    uint32_t u;
    if (best != best) return INT64_MAX; /* NaN never wins (`best >= q` is false) */
    if (best == 0.0f) best = 0.0f; /* -0 -> +0 */
    memcpy(&u, &best, 4);
    if (u & 0x80000000u) u = ~u ^ 0x80000000u; /* negative floats: reverse their order */
    return (int64_t)(((uint64_t)u << 32) | (uint64_t)(0xFFFFFFFFu - slice));
}

ORC_API void orc_unpack_key(int64_t key, float* best, uint32_t* slice) {
    uint32_t u = (uint32_t)((uint64_t)key >> 32);
    if (u & 0x80000000u) u = ~(u ^ 0x80000000u);
    memcpy(best, &u, 4);
    *slice = 0xFFFFFFFFu - (uint32_t)((uint64_t)key & 0xFFFFFFFFu);
}
