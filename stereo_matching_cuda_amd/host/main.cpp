// main.cpp -- drop-in for the reference's entry point (stereo_matching_cuda/main.cu:37-214).
// Same stage order, same progress lines on stdout, same 12 output images, and every stage goes
// through the reference-named host functions of this directory (which forward to the HIP kernels
// behind include/smx.h).  The code itself is organised differently: buffers are std::vectors, the
// outputs are table driven, and the image normaliser is a two-pass restatement of write_mat.
//
//   smx_main                         reference behaviour: ./data/tsukuba0.png, ./data/tsukuba1.png,
//                                    D_MIN..D_MAX from the macros, outputs into ./data/
//   smx_main L.png R.png [dmin dmax [outdir]]
#include <vector>

#include "costVolume.cuh"
#include "filter.cuh"
#include "guidedFilter.cuh"
#include "helpers.cuh"
#include "occlusion.cuh"
#include "png_io.h"
#include "rgb_to_grayscale.cuh"
#include "winner_take_all.cuh"

namespace {

// Float map -> 8-bit image exactly like the reference's write_mat (main.cu:13-35): the maximum is
// the true maximum, but the minimum only considers elements that did NOT raise the running maximum
// at their position (the reference's `else if`); values map through (v - min) * 255 / (max - min)
// in f32 and are truncated.
std::vector<unsigned char> normalise_like_reference(const float* v, size_t n) {
    float hi = -150000000.0f, lo = 150000000.0f;
    for (size_t i = 0; i < n; ++i) {
        const bool raises_max = v[i] > hi;
        if (raises_max) hi = v[i];
        if (!raises_max && v[i] <= lo) lo = v[i];
    }
    std::vector<unsigned char> out(n);
    const float span = hi - lo;
    for (size_t i = 0; i < n; ++i) {
        const int level = (v[i] - lo) * 255.0f / span;
        out[i] = (unsigned char)level;
    }
    return out;
}

struct Pair {
    int w = 0, h = 0;
    unsigned char* rgb[2] = {nullptr, nullptr};
    int channels[2] = {0, 0};
};

bool load_pair(const std::string& left, const std::string& right, Pair& p) {
    int w2 = 0, h2 = 0;
    p.rgb[0] = smx_png_load(left.c_str(), &p.w, &p.h, &p.channels[0]);
    p.rgb[1] = smx_png_load(right.c_str(), &w2, &h2, &p.channels[1]);
    return p.rgb[0] && p.rgb[1] && p.channels[0] >= 3 && p.channels[1] >= 3 && p.w == w2 && p.h == h2;
}

}  // namespace

int main(int argc, char** argv) {
    const bool host_compare = false;          // main.cu:40 (the reference hard-codes false too)
    std::printf("Starting...\n");
    if (smx_device_count() < 1) {
        std::fprintf(stderr, "no HIP device available\n");
        return 1;
    }
    std::printf("Using Device %d: %s\n", 0, smx_version());

    std::string left = "./data/tsukuba0.png", right = "./data/tsukuba1.png", outdir = "./data";
    if (argc >= 3) { left = argv[1]; right = argv[2]; }
    if (argc >= 5) { smx_config().d_min = std::atoi(argv[3]); smx_config().d_max = std::atoi(argv[4]); }
    if (argc >= 6) outdir = argv[5];
    const int d_lo = smx_config().d_min, d_hi = smx_config().d_max;

    const std::clock_t t_begin = std::clock();
    Pair in;
    if (!load_pair(left, right, in)) {
        std::fprintf(stderr, "cannot load an RGB pair of equal size from %s / %s\n", left.c_str(),
                     right.c_str());
        return 1;
    }
    const int w = in.w, h = in.h, n = w * h;
    std::cout << "Resolution : " << w << "x" << h << std::endl;

    std::cout << "RGB to grayscale ..." << std::endl;
    unsigned char* gray[2] = {rgb_to_grayscale(in.rgb[0], n, in.channels[0], host_compare),
                              rgb_to_grayscale(in.rgb[1], n, in.channels[1], host_compare)};

    // left volume: labels d_lo .. d_hi; right volume: labels -d_hi .. -d_lo   (main.cu:79-82)
    const int size_d = d_hi - d_lo + 1;
    const int dmin[2] = {d_lo, -d_hi};
    std::vector<float> cost[2] = {std::vector<float>((size_t)n * size_d), std::vector<float>((size_t)n * size_d)};
    std::cout << "Cost Volume ..." << std::endl;
    compute_cost(gray[0], gray[1], cost[0].data(), w, w, h, h, dmin[0], host_compare);
    compute_cost(gray[1], gray[0], cost[1].data(), w, w, h, h, dmin[1], host_compare);

    // WTA presets of main.cu:112-118: memset(best, 9999999.0f) stores byte 0x7F everywhere
    std::vector<float> best[2], dmap[2];
    std::vector<unsigned char> mean[2], unused_u8[2];
    for (int v = 0; v < 2; ++v) {
        best[v].resize(n);
        std::memset(best[v].data(), 0x7F, sizeof(float) * n);
        dmap[v].assign(n, 0.0f);
        mean[v].assign(n, 0);
        unused_u8[v].assign(n, 0);
    }
    std::cout << "guided filter ..." << std::endl;
    for (int v = 0; v < 2; ++v)
        compute_guided_filter(gray[v], cost[v].data(), best[v].data(), dmap[v].data(), mean[v].data(), w, h,
                              size_d, dmin[v], host_compare);
    std::cout << "guided filter ok" << std::endl;

    // left-right check on a copy of the left map, then scan-line filling on a copy of that
    std::vector<float> occlusion(dmap[0]);
    detect_occlusion(occlusion.data(), dmap[1].data(), dmin[0] - 100, unused_u8[0].data(),
                     unused_u8[1].data(), w, h);                                   // main.cu:149-150
    std::vector<float> filled(occlusion);
    fill_occlusion(filled.data(), w, h, (float)d_lo);                              // main.cu:154-155
    const double duration = (std::clock() - t_begin) / (double)CLOCKS_PER_SEC;

    std::cout << "writing images ..." << std::endl;
    struct U8Out { const char* name; const unsigned char* data; };
    struct F32Out { const char* name; const float* data; };
    const U8Out u8_outputs[] = {{"image_left.png", gray[0]}, {"image_right.png", gray[1]},
                                {"image_mean_left.png", mean[0].data()},
                                {"image_mean_right.png", mean[1].data()}};
    // file names of the reference (its cost images are named after its default range)
    const F32Out f32_outputs[] = {{"best_costl.png", best[0].data()},       {"best_costr.png", best[1].data()},
                                  {"cost_lminus15.png", cost[0].data()},    {"cost_rminus15.png", cost[1].data()},
                                  {"occlu_mapl.png", occlusion.data()},     {"disparity_mapl.png", dmap[0].data()},
                                  {"disparity_mapr.png", dmap[1].data()},   {"occlu_mapl_filled.png", filled.data()}};
    for (const U8Out& o : u8_outputs) smx_png_write((outdir + "/" + o.name).c_str(), w, h, 1, o.data);
    for (const F32Out& o : f32_outputs) {
        const std::vector<unsigned char> img = normalise_like_reference(o.data, (size_t)n);
        smx_png_write((outdir + "/" + o.name).c_str(), w, h, 1, img.data());
    }

    std::cout << "duration: " << duration << std::endl;
    std::cout << "Free the memory ..." << std::endl;
    for (int v = 0; v < 2; ++v) { std::free(gray[v]); std::free(in.rgb[v]); }
    return 0;
}
