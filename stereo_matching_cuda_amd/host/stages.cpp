// stages.cpp -- the reference's per-stage host functions (same names, arguments and in/out
// behaviour) as thin forwards to the C-ABI of include/smx.h.  Host pointers in, host pointers out,
// synchronous, errors abort like the reference's CHECK macro.
#include "costVolume.cuh"
#include "filter.cuh"
#include "guidedFilter.cuh"
#include "occlusion.cuh"
#include "rgb_to_grayscale.cuh"

#include <vector>

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

// rgb_to_grayscale.cu:25-73 (host_gpu_compare: :60-65)
unsigned char* rgb_to_grayscale(unsigned char* h_rgb, const int n, int channels, bool host_gpu_compare) {
    unsigned char* h_gray = (unsigned char*)malloc(n);
    memset(h_gray, 0, n);
    CHECK(smx_rgb_to_grayscale(&smx_config().params, h_rgb, n, channels, h_gray));
    if (host_gpu_compare) {
        std::vector<unsigned char> twin(n);
        sumArraysOnHost(h_rgb, twin.data(), n, channels);
        if (check_errors_grayscale(twin.data(), h_gray, n)) cout << "Grayscale ok!" << endl;
    }
    return h_gray;
}

// costVolume.cu:4-84 (host_gpu_compare: :56-74)
void compute_cost(unsigned char* i1, unsigned char* i2, float* cost, int w1, int w2, int h1, int h2,
                  int dmin, bool host_gpu_compare) {
    const int size_d = smx_config().d_max - smx_config().d_min + 1;   // costVolume.cu:5
    CHECK(smx_compute_cost(&smx_config().params, i1, i2, cost, w1, w2, h1, h2, size_d, dmin));
    if (host_gpu_compare) {
        std::vector<float> twin((size_t)size_d * w1 * h1);
        costVolumeOnCPU(i1, i2, twin.data(), w1, w2, h1, h2, size_d, dmin);
        // (the reference-signature check_errors takes an int count: volumes beyond 2^31 cells are compared in planes)
        bool ok = true;
        for (int z = 0; z < size_d; ++z)
            ok = check_errors(twin.data() + (size_t)z * w1 * h1, cost + (size_t)z * w1 * h1, w1 * h1) && ok;
        if (ok) cout << "Cost volume ok!" << endl;
    }
}

// integral.cu:3-51
void integral(float* image, float* integral, int width, int height) {
    CHECK(smx_integral(image, integral, width, height));
}

// guidedFilter.cu:4-295.  host_gpu_compare: the correct CPU twin runs on copies of the in/out arrays
// and every output is compared exactly (the reference runs its twin from main.cu:136-139 and never
// looks at the result).
void compute_guided_filter(unsigned char* i, float* cost, float* filter_cost, float* disp_map,
                           unsigned char* mean, const int w, const int h, const int size_d, int dmin,
                           bool host_gpu_compare) {
    const int n = w * h;
    std::vector<float> best_twin, dmap_twin;
    if (host_gpu_compare) {
        best_twin.assign(filter_cost, filter_cost + n);
        dmap_twin.assign(disp_map, disp_map + n);
    }
    CHECK(smx_compute_guided_filter(&smx_config().params, i, cost, filter_cost, disp_map, mean,
                                    nullptr, w, h, size_d, dmin));
    if (host_gpu_compare) {
        std::vector<unsigned char> mean_twin(n);
        guided_filter_onCpu(i, cost, best_twin.data(), dmap_twin.data(), mean_twin.data(), w, h, size_d, dmin);
        bool ok = check_errors(best_twin.data(), filter_cost, n);
        ok = check_errors(dmap_twin.data(), disp_map, n) && ok;
        if (mean) ok = check_errors(mean_twin.data(), mean, n) && ok;
        if (ok) cout << "Guided filter ok!" << endl;
    }
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

// filter.cu:117-207: dead code in the reference (never called from main.cu); a standalone device op here.
void filter(unsigned char* image, int width, int height, unsigned char* mean, float* var, bool) {
    CHECK(smx_filter(&smx_config().params, image, width, height, mean, var));
}
