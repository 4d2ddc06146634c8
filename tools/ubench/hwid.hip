// Where do the waves of co-resident workgroups run?  768 workgroups of 512 threads with 49.5 KB of LDS (three per
// CU, like k_v4_walk): prints, per CU, the SIMD of every wave of its workgroups (HW_REG_HW_ID: wave 3:0, simd 5:4,
// cu 11:8, sh 12, se 14:13, tg 19:16; HW_REG_XCC_ID).   hipcc --offload-arch=gfx950 -O2 hwid.hip -o hwid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(512) void k(unsigned* out) {
    __shared__ float pad[12600];
    pad[threadIdx.x] = 1.0f;
    __syncthreads();
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 8 + threadIdx.x / 64) * 2] = hw;
        out[(blockIdx.x * 8 + threadIdx.x / 64) * 2 + 1] = xcc;
    }
    // stay resident long enough for the whole grid to be dispatched
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 200000) __builtin_amdgcn_s_sleep(8);
    if (pad[threadIdx.x] < 0) out[0] = 0;
}
int main() {
    const int nb = 768;
    unsigned* d;
    hipMalloc(&d, nb * 8 * 2 * 4);
    hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, 0, d);
    std::vector<unsigned> h(nb * 16);
    hipMemcpy(h.data(), d, nb * 64, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;   // (xcc, se, sh, cu) -> blocks
    for (int b = 0; b < nb; ++b) {
        const unsigned hw = h[b * 16], x = h[b * 16 + 1] & 15;
        cu[(x << 16) | (((hw >> 13) & 3) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)].push_back(b);
    }
    printf("CUs seen: %zu\n", cu.size());
    int shown = 0, same0 = 0, total = 0;
    for (auto& kv : cu) {
        std::vector<int> simd0;
        for (int b : kv.second) simd0.push_back((h[b * 16] >> 4) & 3);
        ++total;
        bool all_same = true;
        for (int s : simd0) all_same = all_same && s == simd0[0];
        same0 += all_same && simd0.size() > 1;
        if (shown++ < 6) {
            printf("cu %05x:", kv.first);
            for (int b : kv.second) {
                printf("  block %3d tg %2u simd of waves 0..7:", b, (h[b * 16] >> 16) & 15);
                for (int w = 0; w < 8; ++w) printf(" %u", (h[(b * 8 + w) * 2] >> 4) & 3);
            }
            printf("\n");
        }
    }
    printf("CUs whose workgroups all have wave 0 on the same SIMD: %d of %d\n", same0, total);
    return 0;
}
