// Sanitizer leg ("host tests under -fsanitize=address,undefined"), built and run by
// tests/test_host_mirror.py::test_host_layer_and_oracle_under_sanitizers on the CPU only:
//   * the CPU twins of stereo_matching_cuda_amd/host/cpu_twins.cpp against the oracle (oracle/smx_oracle.c,
//     compiled into this program with the same sanitizers) on a small seeded pair, bit for bit;
//   * the PNG reader / writers of host/png_io.cpp on well-formed and malformed files given on the command line.
// No GPU and no libsmx_hip.so: the one symbol the twins need from the host layer (smx_config) is defined here.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "costVolume.cuh"
#include "filter.cuh"
#include "guidedFilter.cuh"
#include "helpers.cuh"
#include "integral.cuh"
#include "occlusion.cuh"
#include "png_io.h"
#include "rgb_to_grayscale.cuh"

// helpers.cuh (defined in stages.cpp next to the GPU wrappers): exact compare, like helpers.cu:3-25
bool check_errors(unsigned char* a, unsigned char* b, int len) { return std::memcmp(a, b, (size_t)len) == 0; }
bool check_errors(float* a, float* b, int len) { return std::memcmp(a, b, (size_t)len * 4) == 0; }

smx_host_config& smx_config() {
    static smx_host_config c = {{0.299, 0.587, 0.0721, 0.9, 7, 2, 9, 6.5025, 0}, -5, 0};
    return c;
}

extern "C" {
typedef struct orc_params {
    double r_w, g_w, b_w, alpha;
    int th_color, th_grad, radius;
    double eps;
    int d_lr;
} orc_params;
void orc_default_params(orc_params* p);
void orc_gray(const orc_params* P, const uint8_t* rgb, int64_t n, int ch, uint8_t* gray);
void orc_cost_volume(const orc_params* P, const uint8_t* i1, const uint8_t* i2, float* cost, int w1, int w2, int h, int size_d, int dmin);
void orc_integral(const float* in, float* out, int w, int h);
void orc_init_wta(float* best, float* dmap, int64_t n);
void orc_filter(const orc_params* P, const uint8_t* I, uint8_t* mean, float* var, int w, int h);
void orc_guided_filter(const orc_params* P, const uint8_t* I, const float* cost, float* best, float* dmap, uint8_t* mean,
                       float* agg, int w, int h, int dmin, int s_begin, int s_end);
void orc_detect_occlusion(const orc_params* P, float* dL, const float* dR, int dOcc, int w, int h);
void orc_fill_occlusion(float* disp, int w, int h, float vMin);
}

static int fails = 0;
static void same(const char* what, const void* a, const void* b, size_t bytes) {
    if (std::memcmp(a, b, bytes) != 0) { std::printf("MISMATCH %s\n", what); ++fails; }
    else std::printf("ok %s\n", what);
}

int main(int argc, char** argv) {
    const int w = 61, h = 47, D = 6, dmin = -5;
    const size_t n = (size_t)w * h;
    std::vector<unsigned char> rgb1(3 * n), rgb2(3 * n);
    unsigned s = 12345u;
    for (size_t k = 0; k < 3 * n; ++k) {
        s = s * 1664525u + 1013904223u;
        rgb1[k] = (unsigned char)(s >> 24);
        rgb2[k] = (unsigned char)((k >= 9 ? rgb1[k - 9] : rgb1[k]) + ((s >> 13) & 3));
    }
    orc_params P;
    orc_default_params(&P);
    std::vector<unsigned char> g1(n), g2(n), o1(n), o2(n);
    sumArraysOnHost(rgb1.data(), g1.data(), (int)n, 3);
    sumArraysOnHost(rgb2.data(), g2.data(), (int)n, 3);
    orc_gray(&P, rgb1.data(), (int64_t)n, 3, o1.data());
    orc_gray(&P, rgb2.data(), (int64_t)n, 3, o2.data());
    same("gray", g1.data(), o1.data(), n);
    std::vector<float> cost(n * D), ocost(n * D);
    costVolumeOnCPU(g1.data(), g2.data(), cost.data(), w, w, h, h, D, dmin);
    orc_cost_volume(&P, g1.data(), g2.data(), ocost.data(), w, w, h, D, dmin);
    same("cost volume", cost.data(), ocost.data(), n * D * 4);
    std::vector<float> S(n), oS(n);
    integralOnCPU(cost.data() + 2 * n, S.data(), w, h);
    orc_integral(cost.data() + 2 * n, oS.data(), w, h);
    same("integral", S.data(), oS.data(), n * 4);
    std::vector<float> best(n), dmap(n), obest(n), odmap(n);
    std::vector<unsigned char> mean(n), omean(n);
    orc_init_wta(best.data(), dmap.data(), (int64_t)n);
    orc_init_wta(obest.data(), odmap.data(), (int64_t)n);
    guided_filter_onCpu(g1.data(), cost.data(), best.data(), dmap.data(), mean.data(), w, h, D, dmin);
    orc_guided_filter(&P, g1.data(), ocost.data(), obest.data(), odmap.data(), omean.data(), nullptr, w, h, dmin, 0, D);
    same("guided filter best", best.data(), obest.data(), n * 4);
    same("guided filter dmap", dmap.data(), odmap.data(), n * 4);
    same("guided filter mean", mean.data(), omean.data(), n);
    {   // boxFilterOnCPU (filter.cuh:6) against the oracle's restatement of filter()'s mean
        std::vector<unsigned char> bm(n), om(n);
        std::vector<float> ov(n);
        boxFilterOnCPU(g1.data(), bm.data(), w, h);
        orc_filter(&P, g1.data(), om.data(), ov.data(), w, h);
        same("boxFilterOnCPU", bm.data(), om.data(), n);
    }
    std::vector<float> dr(n), occ(dmap), oocc(dmap);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) dr[(size_t)y * w + x] = std::fabs(dmap[(size_t)y * w + (w - 1 - x)]);
    detect_occlusionOnCPU(occ.data(), dr.data(), dmin - 100, w, h);
    orc_detect_occlusion(&P, oocc.data(), dr.data(), dmin - 100, w, h);
    same("detect occlusion", occ.data(), oocc.data(), n * 4);
    fill_occlusionOnCPU(occ.data(), w, h, (float)dmin);
    orc_fill_occlusion(oocc.data(), w, h, (float)dmin);
    same("fill occlusion", occ.data(), oocc.data(), n * 4);

    // PNG / PFM writers and the reader on what they wrote
    if (argc > 1) {
        const std::string dir = argv[1];
        const std::string f8 = dir + "/g.png", f16 = dir + "/g16.png", fp = dir + "/d.pfm";
        if (!smx_png_write(f8.c_str(), w, h, 1, g1.data())) { std::printf("MISMATCH png write\n"); ++fails; }
        int rw = 0, rh = 0, rc = 0;
        unsigned char* back = smx_png_load(f8.c_str(), &rw, &rh, &rc);
        if (!back || rw != w || rh != h || rc != 1 || std::memcmp(back, g1.data(), n) != 0) { std::printf("MISMATCH png round trip\n"); ++fails; }
        else std::printf("ok png round trip\n");
        std::free(back);
        std::vector<unsigned short> d16(n);
        for (size_t k = 0; k < n; ++k) d16[k] = (unsigned short)(k * 7);
        if (!smx_png_write_gray16(f16.c_str(), w, h, d16.data()) || !smx_pfm_write(fp.c_str(), w, h, S.data())) {
            std::printf("MISMATCH png16 / pfm write\n");
            ++fails;
        }
        for (int a = 2; a < argc; ++a) {      // malformed files: must be rejected (or loaded) without any report
            int mw, mh, mc;
            unsigned char* p = smx_png_load(argv[a], &mw, &mh, &mc);
            std::printf("%s %s\n", p ? "loaded" : "rejected", argv[a]);
            std::free(p);
        }
    }
    return fails ? 1 : 0;
}
