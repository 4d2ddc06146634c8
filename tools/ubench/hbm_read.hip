// HBM read ceiling on gfx950 for the access shape of the WTA pass (k_v5_wta): every lane walks `planes` planes of a
// 715 MB volume with 16-byte nt loads, 8 in flight, one lane per 16 bytes of a plane -- and the same bytes as one flat stream.
// Question (round 5): the WTA pass reads q at 5.6 TB/s; is that the machine's read ceiling for a volume that was written just
// before, or the kernel's?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/hbm_read.hip -o tools/ubench/hbm_read && tools/ubench/hbm_read
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// planes-major: lane e reads plane z at q[z * np + 4 e] for z = 0 .. planes-1 (the WTA pass)
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_planes(const float* __restrict__ q, float* out, size_t np, int planes) {
    const size_t e0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e0 >= np) return;
    const float* p = q + e0;
    f4 acc = {0, 0, 0, 0};
    for (int z = 0; z + U <= planes; z += U) {
        f4 v[U];
#pragma unroll
        for (int t = 0; t < U; ++t) v[t] = NT ? __builtin_nontemporal_load((const f4*)&p[(size_t)(z + t) * np]) : *(const f4*)&p[(size_t)(z + t) * np];
#pragma unroll
        for (int t = 0; t < U; ++t) acc += v[t];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}
// flat: a grid-stride stream over the whole volume
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_flat(const float* __restrict__ q, float* out, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    f4 acc = {0, 0, 0, 0};
    const f4* p = (const f4*)q;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride * U) {
        f4 v[U];
#pragma unroll
        for (int t = 0; t < U; ++t) { const size_t j = i + t * stride; v[t] = j < n4 ? (NT ? __builtin_nontemporal_load(&p[j]) : p[j]) : (f4){0, 0, 0, 0}; }
#pragma unroll
        for (int t = 0; t < U; ++t) acc += v[t];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}
__global__ void k_fill(float* q, size_t n) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(1.0f, &q[i]); }

int main() {
    const size_t np = (size_t)9 * 375 * 152;        // floats per plane (KITTI: 9 strips x 375 rows x 152 columns)
    const int planes = 192 * 2;                     // both views
    const size_t n = np * planes;
    float *q, *out;
    CK(hipMalloc(&q, n * 4)); CK(hipMalloc(&out, 64));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char* name, auto launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, q, n);      // (written just before, like q)
            hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
        }
        printf("%-44s %7.3f ms  %6.2f TB/s\n", name, best, (double)n * 4 / best / 1e9);
        return 0;
    };
    const unsigned gp = (unsigned)((np / 4 + 255) / 256);
    run("planes-major, 8 nt loads in flight", [&] { hipLaunchKernelGGL((k_planes<8, true>), dim3(gp), dim3(256), 0, 0, q, out, np, planes); });
    run("planes-major, 16 nt loads in flight", [&] { hipLaunchKernelGGL((k_planes<16, true>), dim3(gp), dim3(256), 0, 0, q, out, np, planes); });
    run("planes-major, 8 plain loads in flight", [&] { hipLaunchKernelGGL((k_planes<8, false>), dim3(gp), dim3(256), 0, 0, q, out, np, planes); });
    for (unsigned g : {1024u, 2048u, 4096u, 8192u}) {
        char nm[64]; snprintf(nm, sizeof nm, "flat stream, %u workgroups, 8 nt loads", g);
        run(nm, [&] { hipLaunchKernelGGL((k_flat<8, true>), dim3(g), dim3(256), 0, 0, q, out, n / 4); });
    }
    run("flat stream, 4096 workgroups, 8 plain loads", [&] { hipLaunchKernelGGL((k_flat<8, false>), dim3(4096), dim3(256), 0, 0, q, out, n / 4); });
    run("flat stream, 4096 workgroups, 4 nt loads", [&] { hipLaunchKernelGGL((k_flat<4, true>), dim3(4096), dim3(256), 0, 0, q, out, n / 4); });
    return 0;
}
