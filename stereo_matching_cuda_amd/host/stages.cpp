// stages.cpp -- the reference's per-stage host functions (same names, arguments and in/out
// behaviour) as thin forwards to the C-ABI of include/smx.h.  Host pointers in, host pointers out,
// synchronous, errors abort like the reference's CHECK macro.
#include "costVolume.cuh"
#include "filter.cuh"
#include "guidedFilter.cuh"
#include "occlusion.cuh"
#include "rgb_to_grayscale.cuh"

using namespace std;

smx_host_config& smx_config() {
    static smx_host_config c = [] {
        smx_host_config k;
        smx_default_params(&k.params);
        k.params.r_w = R_W; k.params.g_w = G_W; k.params.b_w = B_W;
        k.params.alpha = ALPHA; k.params.th_color = TH_color; k.params.th_grad = TH_grad;
        k.params.radius = RADIUS; k.params.eps = EPS; k.params.d_lr = D_LR;
        k.d_min = D_MIN; k.d_max = D_MAX;
        return k;
    }();
    return c;
}

// helpers.cu:3-25 -- exact-equality compare that reports every mismatching element
namespace {
template <class T>
bool report_mismatches(const T* expected, const T* got, int len) {
    int mismatches = 0;
    for (int k = 0; k < len; ++k) {
        if (expected[k] == got[k]) continue;
        ++mismatches;
        cout << "error at element: " << k << " ResultGPU = " << +got[k] << " and ResultCPU= " << +expected[k]
             << endl;
    }
    return mismatches == 0;
}
}  // namespace

bool check_errors(float* resCPU, float* resGPU, int len) { return report_mismatches(resCPU, resGPU, len); }
bool check_errors(unsigned char* resCPU, unsigned char* resGPU, int len) {
    return report_mismatches(resCPU, resGPU, len);
}

// rgb_to_grayscale.cu:25-73
unsigned char* rgb_to_grayscale(unsigned char* h_rgb, const int n, int channels, bool) {
    unsigned char* h_gray = (unsigned char*)malloc(n);
    memset(h_gray, 0, n);
    CHECK(smx_rgb_to_grayscale(&smx_config().params, h_rgb, n, channels, h_gray));
    return h_gray;
}

// costVolume.cu:4-84
void compute_cost(unsigned char* i1, unsigned char* i2, float* cost, int w1, int w2, int h1, int h2,
                  int dmin, bool) {
    const int size_d = smx_config().d_max - smx_config().d_min + 1;   // costVolume.cu:5
    CHECK(smx_compute_cost(&smx_config().params, i1, i2, cost, w1, w2, h1, h2, size_d, dmin));
}

// integral.cu:3-51
void integral(float* image, float* integral, int width, int height) {
    CHECK(smx_integral(image, integral, width, height));
}

// guidedFilter.cu:4-295
void compute_guided_filter(unsigned char* i, float* cost, float* filter_cost, float* disp_map,
                           unsigned char* mean, const int w, const int h, const int size_d, int dmin,
                           bool) {
    CHECK(smx_compute_guided_filter(&smx_config().params, i, cost, filter_cost, disp_map, mean,
                                    nullptr, w, h, size_d, dmin));
}

// occlusion.cu:17-85
void detect_occlusion(float* disparityLeft, float* disparityRight, const int dOcclusion,
                      unsigned char*, unsigned char*, const int w, const int h) {
    CHECK(smx_detect_occlusion(&smx_config().params, disparityLeft, disparityRight, dOcclusion, w, h));
}

// occlusion.cu:111-132
void fill_occlusion(float* disparity, const int w, const int h, const float vMin) {
    CHECK(smx_fill_occlusion(disparity, w, h, vMin));
}

// filter.cu:117-207 is dead code in the reference and not part of the stereo path.
void filter(unsigned char*, int, int, unsigned char*, float*, bool) {
    fprintf(stderr, "filter(): dead code in the reference (never called from main.cu); "
                    "not on the stereo path and not provided.\n");
    exit(1);
}
