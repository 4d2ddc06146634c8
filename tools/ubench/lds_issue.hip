// dev microbenchmark: issue cost (one wave, back to back, independent) of LDS instructions
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(X) X X X X X X X X X X X X X X X X
__global__ void k(float* out, unsigned long long* cyc, int n, int nact) {
    __shared__ f4 buf[2048];
    f4 v = {1, 2, 3, 4};
    f2 w = {1, 2};
    unsigned a128 = threadIdx.x * 16, a64 = threadIdx.x * 8, arow = threadIdx.x * 688;
    unsigned long long t[12];
    int s = 0;
    const bool act = (int)threadIdx.x < nact;
    auto T = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t[s++] = __builtin_amdgcn_s_memtime(); };
    T();
    if (act) for (int i = 0; i < n; ++i) { REP16(asm volatile("ds_write_b128 %0, %1" :: "v"(a128), "v"(v) : "memory");) }
    T();
    if (act) for (int i = 0; i < n; ++i) { REP16(asm volatile("ds_write_b64 %0, %1" :: "v"(a64), "v"(w) : "memory");) }
    T();
    if (act) for (int i = 0; i < n; ++i) { REP16(asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a128) : "memory");) }
    T();
    if (act) for (int i = 0; i < n; ++i) { REP16(asm volatile("ds_read_b64 %0, %1" : "=v"(w) : "v"(a64) : "memory");) }
    T();
    // LANE = ROW addressing (stride 688 B) as in the row scan
    if (act) for (int i = 0; i < n; ++i) { REP16(asm volatile("ds_write_b128 %0, %1" :: "v"(arow), "v"(v) : "memory");) }
    T();
    if (act) for (int i = 0; i < n; ++i) { REP16(asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(arow) : "memory");) }
    T();
    // the row-scan pattern: read, 2 dependent pk adds, write, per column pair; reads 4 pairs ahead
    f2 acc = {0, 0};
    if (act) for (int i = 0; i < n; ++i) {
        auto step = [&]() {
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(arow) : "memory");
            f2 lo, hi; lo.x = v.x; lo.y = v.y; hi.x = v.z; hi.y = v.w;
            f2 s0 = lo + acc;
            acc = hi + s0;
            f4 o; o.x = s0.x; o.y = s0.y; o.z = acc.x; o.w = acc.y;
            asm volatile("ds_write_b128 %0, %1" :: "v"(arow), "v"(o) : "memory");
        };
        REP16(step();)
    }
    T();
    out[threadIdx.x] = v.x + w.x + acc.x + buf[threadIdx.x].x;
    if (threadIdx.x == 0) for (int j = 0; j + 1 < s; ++j) cyc[j] = t[j + 1] - t[j];
}
int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256); (void)hipMalloc(&cyc, 128);
    const int n = 500;
    for (int nact : {64, 26}) {
        k<<<1, 64>>>(out, cyc, n, nact); k<<<1, 64>>>(out, cyc, n, nact);
        unsigned long long h[8]; (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        const char* nm[] = {"w128", "w64", "r128", "r64", "w128 rowstride", "r128 rowstride", "serial read-2add-write"};
        printf("active lanes %d:", nact);
        for (int j = 0; j < 7; ++j) printf("  %s %.1f", nm[j], h[j] / (16.0 * n));
        printf("\n");
    }
    return 0;
}
