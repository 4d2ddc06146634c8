// cpu_twins.cpp -- host-side twins of the device stages for the `host_gpu_compare` self-check mode of the
// reference (main.cu:40, costVolume.cu:56-74, rgb_to_grayscale.cu:60-65, guidedFilter.cu:75-78).
// The names and argument lists are the reference's (costVolume.cuh:8-14, guidedFilter.cuh:9-39,
// integral.cuh:7, occlusion.cuh:12,19, rgb_to_grayscale.cuh:5,8); the bodies are this repository's own
// sequential restatements of what the DEVICE kernels compute -- in particular guided_filter_onCpu is a
// correct twin (the reference's version writes pixel indices as labels and divides in double,
// SURVEY.md section 4), so that check_errors() can hold it against the GPU result bit for bit.
// Everything is plain f32 arithmetic in source order; build with -ffp-contract=off.
#include <vector>

#include "costVolume.cuh"
#include "filter.cuh"
#include "guidedFilter.cuh"
#include "integral.cuh"
#include "occlusion.cuh"
#include "rgb_to_grayscale.cuh"

using std::vector;

// ---- filter.cuh:6 (see the note there: a correct twin of the device path's mean, not the reference's broken body) --------
void boxFilterOnCPU(unsigned char* image, unsigned char* mean, int width, int height) {
    const int R = smx_config().params.radius, area = (2 * R + 1) * (2 * R + 1);
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            float sum = 0.0f;
            for (int ix = -R; ix <= R; ++ix)
                for (int iy = -R; iy <= R; ++iy) {
                    const int xx = x + ix, yy = y + iy;
                    sum += (xx >= 0 && xx < width && yy >= 0 && yy < height) ? (float)image[(size_t)yy * width + xx] : 0.0f;
                }
            mean[(size_t)y * width + x] = (unsigned char)(int)(sum / area);
        }
}

// ---- rgb_to_grayscale.cuh ------------------------------------------------------------------
void sumArraysOnHost(unsigned char* image, unsigned char* gray, const int N, int channels) {
    const smx_params& P = smx_config().params;
    for (int k = 0; k < N; ++k) {
        const unsigned char* px = image + (size_t)channels * k;
        const double v = P.r_w * px[0] + P.g_w * px[1] + P.b_w * px[2];
        gray[k] = (unsigned char)v;
    }
}

bool check_errors_grayscale(unsigned char* host, unsigned char* gpu, int len) {
    return check_errors(host, gpu, len);
}

// ---- costVolume.cuh ------------------------------------------------------------------------
int iDivUp(int a, int b) { return (a + b - 1) / b; }

float x_derivativeCPU(unsigned char* im, int col_index, int index, int width) {
    int right, left;
    if (col_index - 1 >= 0 && col_index + 1 < width) { right = im[index + 1]; left = im[index - 1]; }
    else if (col_index + 1 >= width)                 { right = im[index];     left = im[index - 1]; }
    else                                             { right = im[index + 1]; left = im[index];     }
    return 1.0f * (float)(left - right) / 2;
}

void x_derivativeOnCpu(unsigned char* in, float* out, int w, int h) {
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) out[(size_t)y * w + x] = x_derivativeCPU(in, x, y * w + x, w);
}

void compute_costVolumeOnCpu(unsigned char* i1, unsigned char* i2, float* cost, float* derivative1,
                             float* derivative2, int w1, int w2, int h1, int h2, int size_d, int dmin) {
    (void)h2;
    const smx_params& P = smx_config().params;
    const float alpha = 1.0f * P.alpha, th_color = 1.0f * P.th_color, th_grad = 1.0f * P.th_grad;
    const float border = (1 - alpha) * th_color + 1.0f * alpha * th_grad;
    const size_t n = (size_t)w1 * h1;
    for (int z = 0; z < size_d; ++z) {
        const int d = dmin + z;
        for (int y = 0; y < h1; ++y)
            for (int x = 0; x < w1; ++x) {
                const size_t a = (size_t)y * w1 + x;
                float c = border;
                if (x + d < w2 && x + d >= 0) {
                    const size_t b = a + d;
                    const int di = (int)i1[a] - (int)i2[b];
                    const float t1 = 1.0f * (float)(di < 0 ? -di : di);
                    const float t2 = 1.0f * fabsf(derivative1[a] - derivative2[b]);
                    const float m1 = t1 < th_color ? t1 : th_color;
                    const float m2 = t2 < th_grad ? t2 : th_grad;
                    const float p1 = (1.0f - alpha) * m1;
                    const float p2 = alpha * m2;
                    c = p1 + p2;
                }
                cost[(size_t)z * n + a] = c;
            }
    }
}

void costVolumeOnCPU(unsigned char* i1, unsigned char* i2, float* cost, int w1, int w2, int h1, int h2,
                     int size_d, int dmin) {
    vector<float> g1((size_t)w1 * h1), g2((size_t)w2 * h2);
    x_derivativeOnCpu(i1, g1.data(), w1, h1);
    x_derivativeOnCpu(i2, g2.data(), w2, h2);
    compute_costVolumeOnCpu(i1, i2, cost, g1.data(), g2.data(), w1, w2, h1, h2, size_d, dmin);
}

// ---- integral.cuh --------------------------------------------------------------------------
void integralOnCPU(float* in, float* out, const int w, const int h) {
    for (int y = 0; y < h; ++y) {
        float acc = in[(size_t)y * w];
        out[(size_t)y * w] = acc;
        for (int x = 1; x < w; ++x) {
            acc = in[(size_t)y * w + x] + acc;
            out[(size_t)y * w + x] = acc;
        }
    }
    for (int y = 1; y < h; ++y)
        for (int x = 0; x < w; ++x) out[(size_t)y * w + x] = out[(size_t)y * w + x] + out[(size_t)(y - 1) * w + x];
}

// ---- guidedFilter.cuh ----------------------------------------------------------------------
float computeMeanOnCPU(float* I, float* S, int idx, int idy, const int w, const int h) {
    (void)I;
    const int R = smx_config().params.radius;
    const int y0 = std::max(-1, idy - R - 1), y1 = std::min(h - 1, idy + R);
    const int x0 = std::max(-1, idx - R - 1), x1 = std::min(w - 1, idx + R);
    float val = S[(size_t)y1 * w + x1];
    if (x0 >= 0) val -= S[(size_t)y1 * w + x0];
    if (y0 >= 0) val -= S[(size_t)y0 * w + x1];
    if (x0 >= 0 && y0 >= 0) val += S[(size_t)y0 * w + x0];
    return 1.0f * val / (float)((x1 - x0) * (y1 - y0));
}

void computeBoxFilterOnCPU(float* image, float* integral, float* mean, const int w, const int h) {
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) mean[(size_t)y * w + x] = computeMeanOnCPU(image, integral, x, y, w, h);
}

void chToFlOnCPU(unsigned char* image, float* result, int len) {
    for (int k = 0; k < len; ++k) result[k] = 1.0f * (float)(int)image[k];
}
void flToChOnCPU(float* image, unsigned char* result, int len) {
    for (int k = 0; k < len; ++k) {
        const int c = (int)image[k];
        result[k] = c > 255 ? 255 : (unsigned char)c;
    }
}
void pixelMultOnCPU(float* a, float* b, float* r, int len) { for (int k = 0; k < len; ++k) r[k] = a[k] * b[k]; }
void pixelSousOnCPU(float* a, float* b, float* r, int len) { for (int k = 0; k < len; ++k) r[k] = a[k] - b[k]; }
void pixelAddOnCPU(float* a, float* b, float* r, int len) { for (int k = 0; k < len; ++k) r[k] = a[k] + b[k]; }
void pixelDivOnCPU(float* a, float* b, float* r, int len) { for (int k = 0; k < len; ++k) r[k] = a[k] / b[k]; }

void dispSelectOnCPU(float* q, float* filter_cost, float* dmap, const int n, int label) {
    for (int k = 0; k < n; ++k)
        if (1.0f * filter_cost[k] >= 1.0f * q[k]) { dmap[k] = (float)label; filter_cost[k] = q[k]; }
}

// What compute_guided_filter computes on the device (guidedFilter.cu:58-238), sequentially.
void guided_filter_onCpu(unsigned char* im1, float* cost, float* filtered_cost, float* dmap,
                         unsigned char* mean, const int w, const int h, const int size_d, int dmin) {
    const int n = w * h;
    const double eps = smx_config().params.eps;
    vector<float> im(n), S(n), mI(n), var(n), t(n), mp(n), mIp(n), a(n), b(n), q(n);
    chToFlOnCPU(im1, im.data(), n);
    integralOnCPU(im.data(), S.data(), w, h);
    computeBoxFilterOnCPU(im.data(), S.data(), mI.data(), w, h);
    if (mean) flToChOnCPU(mI.data(), mean, n);
    pixelMultOnCPU(im.data(), im.data(), t.data(), n);
    integralOnCPU(t.data(), S.data(), w, h);
    computeBoxFilterOnCPU(t.data(), S.data(), var.data(), w, h);
    pixelMultOnCPU(mI.data(), mI.data(), t.data(), n);
    pixelSousOnCPU(var.data(), t.data(), var.data(), n);
    for (int s = 0; s < size_d; ++s) {
        float* p = cost + (size_t)s * n;
        integralOnCPU(p, S.data(), w, h);
        computeBoxFilterOnCPU(p, S.data(), mp.data(), w, h);
        pixelMultOnCPU(im.data(), p, t.data(), n);
        integralOnCPU(t.data(), S.data(), w, h);
        computeBoxFilterOnCPU(t.data(), S.data(), mIp.data(), w, h);
        for (int k = 0; k < n; ++k) {                    // compute_ak_and_bk, guidedFilter.cu:345-354
            const float c = (float)(1.0f / ((double)var[k] + eps));
            const float mm = mI[k] * mp[k];
            a[k] = 1.0f * (mIp[k] - mm) * c;
            const float mb = 1.0f * mI[k] * a[k];
            b[k] = 1.0f * mp[k] - mb;
        }
        integralOnCPU(a.data(), S.data(), w, h);
        computeBoxFilterOnCPU(a.data(), S.data(), t.data(), w, h);      // mean(a)
        integralOnCPU(b.data(), S.data(), w, h);
        computeBoxFilterOnCPU(b.data(), S.data(), mp.data(), w, h);     // mean(b)
        for (int k = 0; k < n; ++k) {                    // compute_q, :363-369
            const float m = t[k] * im[k];
            q[k] = m + mp[k];
        }
        dispSelectOnCPU(q.data(), filtered_cost, dmap, n, dmin + s);
    }
}

// ---- occlusion.cuh -------------------------------------------------------------------------
void detect_occlusionOnCPU(float* disparityLeft, float* disparityRight, const int dOcclusion, const int w,
                           const int h) {
    const int d_lr = smx_config().params.d_lr;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t id = (size_t)y * w + x;
            const int d = (int)disparityLeft[id];
            if (x + d < 0 || x + d >= w || fabsf((float)d + disparityRight[id + d]) > (float)d_lr)
                disparityLeft[id] = (float)dOcclusion;
        }
}

void fill_occlusionOnCPU(float* disparity, const int w, const int h, const float vMin) {
    vector<float> row(w);
    for (int y = 0; y < h; ++y) {
        float* r = disparity + (size_t)y * w;
        std::copy(r, r + w, row.begin());       // snapshot: every pixel sees the unfilled row
        for (int x = 0; x < w; ++x) {
            if ((float)(int)row[x] >= vMin) continue;
            float left = vMin, right = vMin;
            for (int k = x; k >= 0; --k) if (row[k] >= vMin) { left = row[k]; break; }
            for (int k = x; k < w; ++k) if (row[k] >= vMin) { right = row[k]; break; }
            r[x] = left > right ? left : right;
        }
    }
}
