// Rate of 64-bit signed atomic min on a key map, the access shape of a fused winner-take-all: every wave
// instruction covers 64 consecutive 8-byte keys (512 B) of a row.  Variants: device scope (what a cross-XCD
// combine needs) and wave scope on a per-XCD copy of the map (the atomic executes in the XCD's own L2; the 8
// copies would be min-combined afterwards).  Also a plain 8-byte store pass for comparison.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(256) void k(long long* maps, size_t n, int rounds) {
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    long long* map = MODE == 1 ? maps + (size_t)xcc * n : maps;
    const size_t nrow = n / 64;
    size_t row = ((size_t)blockIdx.x * 4 + threadIdx.x / 64) * 2654435761u % nrow;
    const int lane = threadIdx.x & 63;
    long long key = ((long long)(threadIdx.x + blockIdx.x) << 32) | lane;
    for (int r = 0; r < rounds; ++r) {
        long long* p = map + row * 64 + lane;
        if (MODE == 0) __hip_atomic_fetch_min(p, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 1) __hip_atomic_fetch_min(p, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (MODE == 2) __builtin_nontemporal_store(key, p);
        row = (row + 9973) % nrow;
        key -= 7;
    }
}
template <int MODE> void run(const char* name, long long* d, size_t n) {
    const int rounds = 256, blocks = 2048;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, n, rounds); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, n, rounds); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 256 * rounds * 8;
    printf("%-34s %.3f ms  %.2f TB/s of keys (%.1f M wave-instructions/s)\n", name, ms, bytes / ms / 1e9, blocks * 4.0 * rounds / ms / 1e3);
}
int main() {
    const size_t n = 2 * 1242 * 375;   // keys of one KITTI pair
    long long* d; hipMalloc(&d, 8 * n * 8); hipMemset(d, 0x7f, 8 * n * 8);
    run<0>("atomic smin64, agent scope", d, n);
    run<1>("atomic smin64, wave scope, per XCD", d, n);
    run<2>("plain nt store 8 B", d, n);
    return 0;
}
