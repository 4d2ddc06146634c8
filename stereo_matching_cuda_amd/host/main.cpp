// main.cpp -- drop-in for the reference's entry point (stereo_matching_cuda/main.cu:37-214):
// same flow, same progress lines, same 12 output images, every stage through the reference-named
// host functions of this directory (which forward to the HIP kernels behind include/smx.h).
//
//   smx_main                         reference behaviour: ./data/tsukuba0.png, ./data/tsukuba1.png,
//                                    D_MIN..D_MAX from the macros, outputs into ./data/
//   smx_main L.png R.png [dmin dmax [outdir]]
#include "costVolume.cuh"
#include "filter.cuh"
#include "guidedFilter.cuh"
#include "helpers.cuh"
#include "occlusion.cuh"
#include "png_io.h"
#include "rgb_to_grayscale.cuh"
#include "winner_take_all.cuh"

using namespace std;

// main.cu:13-35
void write_mat(float* mat, const char* filename, int w, int h, int start) {
    unsigned char* matchar = (unsigned char*)malloc((size_t)w * h);
    memset(matchar, 0, (size_t)w * h);
    float max = -150000000.0f;
    float min = 150000000.0f;
    for (int i = start; i < start + w * h; i++) {
        if (mat[i] > max) {
            max = mat[i];
        } else if (mat[i] <= min) {
            min = mat[i];
        }
    }
    for (int i = 0; i < w * h; i++) {
        int c = (mat[i + start] - min) * 255.0f / (max - min);
        matchar[i] = (unsigned char)c;
    }
    smx_png_write(filename, w, h, 1, matchar);
    free(matchar);
}

int main(int argc, char** argv) {
    bool host_compare = false;
    printf("Starting...\n");
    if (smx_device_count() < 1) {
        fprintf(stderr, "no HIP device available\n");
        return 1;
    }
    printf("Using Device %d: %s\n", 0, smx_version());

    string left = "./data/tsukuba0.png", right = "./data/tsukuba1.png", outdir = "./data";
    if (argc >= 3) { left = argv[1]; right = argv[2]; }
    if (argc >= 5) { smx_config().d_min = atoi(argv[3]); smx_config().d_max = atoi(argv[4]); }
    if (argc >= 6) outdir = argv[5];
    const int d_min = smx_config().d_min, d_max = smx_config().d_max;

    std::clock_t start = std::clock();
    int w1, h1, ch1, w2, h2, ch2;
    unsigned char* data1 = smx_png_load(left.c_str(), &w1, &h1, &ch1);
    unsigned char* data2 = smx_png_load(right.c_str(), &w2, &h2, &ch2);
    if (!data1 || !data2 || ch1 < 3 || ch2 < 3 || w1 != w2 || h1 != h2) {
        fprintf(stderr, "cannot load an RGB pair of equal size from %s / %s\n", left.c_str(), right.c_str());
        return 1;
    }
    int n1 = w1 * h1, n2 = w2 * h2;
    cout << "Resolution : " << w1 << "x" << h1 << endl;
    cout << "RGB to grayscale ..." << endl;
    unsigned char* I_l = rgb_to_grayscale(data1, n1, ch1, host_compare);
    unsigned char* I_r = rgb_to_grayscale(data2, n2, ch2, host_compare);

    int size_d = d_max - d_min + 1;
    size_t totalSize1 = (size_t)n1 * size_d, totalSize2 = (size_t)n2 * size_d;
    float* costl = (float*)malloc(sizeof(float) * totalSize1);
    float* costr = (float*)malloc(sizeof(float) * totalSize2);
    cout << "Cost Volume ..." << endl;
    const int dminl = d_min;
    compute_cost(I_l, I_r, costl, w1, w2, h1, h2, dminl, host_compare);
    const int dminr = -d_max;
    compute_cost(I_r, I_l, costr, w2, w1, h2, h1, dminr, host_compare);

    unsigned char* mean1 = (unsigned char*)malloc(n1);
    unsigned char* mean2 = (unsigned char*)malloc(n2);
    float* best_costl = (float*)malloc(n1 * sizeof(float));
    float* best_costr = (float*)malloc(n2 * sizeof(float));
    memset(best_costl, 9999999.0f, n1 * sizeof(float));   // main.cu:112: every byte 0x7F
    memset(best_costr, 9999999.0f, n2 * sizeof(float));
    float* dmapl = (float*)malloc(n1 * sizeof(float));
    float* dmapr = (float*)malloc(n2 * sizeof(float));
    memset(dmapl, 0, n1 * sizeof(float));
    memset(dmapr, 0, n2 * sizeof(float));
    unsigned char* dmaplChar = (unsigned char*)calloc(n1, 1);
    unsigned char* dmaprChar = (unsigned char*)calloc(n2, 1);

    cout << "guided filter ..." << endl;
    compute_guided_filter(I_l, costl, best_costl, dmapl, mean1, w1, h1, size_d, dminl, host_compare);
    compute_guided_filter(I_r, costr, best_costr, dmapr, mean2, w2, h2, size_d, dminr, host_compare);

    float* occlusion = (float*)malloc(n1 * sizeof(float));
    memcpy(occlusion, dmapl, n1 * sizeof(float));
    float* occlusion_filled = (float*)malloc(n1 * sizeof(float));
    cout << "guided filter ok" << endl;

    const int dOcclusion = (dminl - 100);
    detect_occlusion(occlusion, dmapr, dOcclusion, dmaplChar, dmaprChar, w1, h1);
    memcpy(occlusion_filled, occlusion, n1 * sizeof(float));
    int vMin = d_min;
    fill_occlusion(occlusion_filled, w1, h1, vMin);
    double duration = (std::clock() - start) / (double)CLOCKS_PER_SEC;

    cout << "writing images ..." << endl;
    auto out = [&](const char* name) { return outdir + "/" + name; };
    smx_png_write(out("image_left.png").c_str(), w1, h1, 1, I_l);
    smx_png_write(out("image_right.png").c_str(), w2, h2, 1, I_r);
    smx_png_write(out("image_mean_left.png").c_str(), w1, h1, 1, mean1);
    smx_png_write(out("image_mean_right.png").c_str(), w2, h2, 1, mean2);
    write_mat(best_costl, out("best_costl.png").c_str(), w1, h1, 0);
    write_mat(best_costr, out("best_costr.png").c_str(), w2, h2, 0);
    // the reference names these after its default range (cost_lminus15.png); same names kept
    write_mat(costl, out("cost_lminus15.png").c_str(), w1, h1, 0);
    write_mat(costr, out("cost_rminus15.png").c_str(), w2, h2, 0);
    write_mat(occlusion, out("occlu_mapl.png").c_str(), w1, h1, 0);
    write_mat(dmapl, out("disparity_mapl.png").c_str(), w1, h1, 0);
    write_mat(dmapr, out("disparity_mapr.png").c_str(), w2, h2, 0);
    write_mat(occlusion_filled, out("occlu_mapl_filled.png").c_str(), w1, h1, 0);

    std::cout << "duration: " << duration << std::endl;
    cout << "Free the memory ..." << endl;
    free(occlusion); free(occlusion_filled); free(I_l); free(I_r); free(data1); free(data2);
    free(mean1); free(mean2); free(costl); free(costr); free(dmapl); free(dmapr);
    free(best_costr); free(best_costl); free(dmaprChar); free(dmaplChar);
    return 0;
}
