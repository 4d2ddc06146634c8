// Test program (built by tests/test_host_mirror.py): drives the reference-signature C++ wrappers of
// stereo_matching_cuda_amd/host/ -- integral(), filter(), check_errors(), the occlusion pair -- on a GPU
// and holds them against the host-side twins.  Prints one "ok <name>" line per check; exit code = failures.
#include <vector>

#include "costVolume.cuh"
#include "filter.cuh"
#include "guidedFilter.cuh"
#include "occlusion.cuh"

static unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

int main() {
    int failures = 0;
    unsigned seed = 12345;
    const int w = 131, h = 77, n = w * h;
    // check_errors itself: equal arrays pass, one differing element is reported
    {
        std::vector<float> a(16, 1.5f), b(16, 1.5f);
        bool same = check_errors(a.data(), b.data(), 16);
        b[7] = 2.0f;
        bool diff = check_errors(a.data(), b.data(), 16);
        std::vector<unsigned char> c(8, 3), d(8, 3);
        bool same8 = check_errors(c.data(), d.data(), 8);
        if (same && !diff && same8) std::cout << "ok check_errors" << std::endl; else ++failures;
    }
    // integral(): device wrapper vs integralOnCPU (integral.cu:3-51 vs :92-119)
    {
        std::vector<float> img(n), dev(n), cpu(n);
        for (int i = 0; i < n; ++i) img[i] = (float)(lcg(seed) % 2001) / 7.0f - 100.0f;
        integral(img.data(), dev.data(), w, h);
        integralOnCPU(img.data(), cpu.data(), w, h);
        if (check_errors(cpu.data(), dev.data(), n)) std::cout << "ok integral" << std::endl; else ++failures;
    }
    // filter(): device wrapper vs a direct restatement of filter.cu:39-115,143-181
    {
        std::vector<unsigned char> img(n), mean(n), mean_ref(n);
        std::vector<float> var(n), var_ref(n);
        for (int i = 0; i < n; ++i) img[i] = (unsigned char)(lcg(seed) & 0xFF);
        filter(img.data(), w, h, mean.data(), var.data(), true);
        const int R = RADIUS, area = (2 * R + 1) * (2 * R + 1);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                float s1 = 0, s2 = 0;
                for (int ix = -R; ix <= R; ++ix)
                    for (int iy = -R; iy <= R; ++iy) {
                        int xx = x + ix, yy = y + iy;
                        float v = 0, v2 = 0;
                        if (xx >= 0 && xx < w && yy >= 0 && yy < h) { int c = img[yy * w + xx]; v = (float)c; v2 = (float)(c * c); }
                        s1 += v; s2 += v2;
                    }
                unsigned char m = (unsigned char)(int)(s1 / area);
                mean_ref[y * w + x] = m;
                var_ref[y * w + x] = (float)(int)(s2 / area) - (float)((int)m * (int)m);
            }
        bool ok = check_errors(mean_ref.data(), mean.data(), n);
        ok = check_errors(var_ref.data(), var.data(), n) && ok;
        if (ok) std::cout << "ok filter" << std::endl; else ++failures;
        // boxFilterOnCPU (filter.cuh:6): the CPU twin of the device path's mean
        std::vector<unsigned char> mean_cpu(n);
        boxFilterOnCPU(img.data(), mean_cpu.data(), w, h);
        if (check_errors(mean_cpu.data(), mean.data(), n)) std::cout << "ok boxFilterOnCPU" << std::endl; else ++failures;
    }
    // detect_occlusion / fill_occlusion vs their twins
    {
        std::vector<float> dl(n), dr(n);
        for (int i = 0; i < n; ++i) { dl[i] = -(float)(lcg(seed) % 16); dr[i] = (float)(lcg(seed) % 16); }
        std::vector<float> a(dl), b(dl);
        std::vector<unsigned char> u(n);
        detect_occlusion(a.data(), dr.data(), -115, u.data(), u.data(), w, h);
        detect_occlusionOnCPU(b.data(), dr.data(), -115, w, h);
        bool ok = check_errors(b.data(), a.data(), n);
        fill_occlusion(a.data(), w, h, -15.0f);
        fill_occlusionOnCPU(b.data(), w, h, -15.0f);
        ok = check_errors(b.data(), a.data(), n) && ok;
        if (ok) std::cout << "ok occlusion" << std::endl; else ++failures;
    }
    return failures;
}
