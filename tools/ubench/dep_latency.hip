// dev microbenchmark: dependent-chain latency of v_add_f32 / v_pk_add_f32 and LDS b128 round trips, one wave
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, unsigned long long* cyc, int n) {
    __shared__ f4 buf[64 * 8];
    float a = threadIdx.x, b = 1.0f;
    f2 p = {a, a}, q = {1.0f, 2.0f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(a) : "v"(b));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q));
    }
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    // two independent v_add chains interleaved (x and y as separate scalars)
    float c = a, d = a;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(c) : "v"(b));
            asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(d) : "v"(b));
        }
    }
    unsigned long long t3 = __builtin_amdgcn_s_memtime();
    // LDS: write b128 then read it back dependent
    f4 v = {a, a, a, a};
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            buf[threadIdx.x] = v;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            v = buf[threadIdx.x ^ 1];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    unsigned long long t4 = __builtin_amdgcn_s_memtime();
    // issue cost of back-to-back independent ds_write_b128 (26 lanes active)
    if (threadIdx.x < 26) {
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) buf[threadIdx.x + 64 * (u & 7)] = v;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t5 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) buf[threadIdx.x + 64 * (u & 7)] = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t6 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a + p.x + p.y + c + d + v.x + buf[threadIdx.x].y;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; cyc[5] = t6 - t5; }
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256); hipMalloc(&cyc, 64);
    const int n = 1000;
    for (int r = 0; r < 2; ++r) k<<<1, 64>>>(out, cyc, n);
    unsigned long long h[6]; hipMemcpy(h, cyc, 48, hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz on gfx9? report raw per-op units too
    printf("v_add dep: %.2f  v_pk_add dep: %.2f  2x v_add interleaved (per pair): %.2f  lds w128+r128 round trip: %.2f  ds_write_b128 26 lanes: %.2f  64 lanes: %.2f  (memtime ticks per op)\n",
           h[0] / (16.0 * n), h[1] / (16.0 * n), h[2] / (16.0 * n), h[3] / (4.0 * n), h[4] / (16.0 * n), h[5] / (16.0 * n));
    return 0;
}
