// VALU issue rate on gfx950: cycles per wave64 instruction per SIMD for v_fma_f32, v_pk_fma_f32, v_add_f32,
// v_pk_add_f32, s_add and (round 4) the DPP, mixed-precision and packed-half instructions of the comb walker with 1, 2, 4
// waves per SIMD (independent instructions, 8 accumulators).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters) {
    float a[8]; f2 p[8]; int s = __builtin_amdgcn_readfirstlane(threadIdx.x);
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.5f + i; p[i] = (f2){a[i], a[i] + 1}; }
    const float m = 1.0001f, c = 0.5f; const f2 m2 = {m, m}, c2 = {c, c};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
                if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
                if (KIND == 4) asm volatile("s_add_i32 %0, %0, 1" : "+s"(s));
                if (KIND == 5) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(a[(i + 4) & 7]));
                if (KIND == 6) asm volatile("v_fma_mix_f32 %0, %0, %1, 0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(m));
                if (KIND == 7) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 8) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[i]) : "v"(m2));
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = s;
    for (int i = 0; i < 8; ++i) acc += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND> void run(const char* name) {
    float* out; unsigned long long* cyc; hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8192);
    const int iters = 2000;
    for (int wps = 1; wps <= 4; wps *= 2) {       // waves per SIMD: block of 256*wps threads, one block per CU
        k<KIND><<<256, 256 * wps>>>(out, cyc, iters); hipDeviceSynchronize();
        k<KIND><<<256, 256 * wps>>>(out, cyc, iters); hipDeviceSynchronize();
        unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
        printf("%-14s %d waves/SIMD: %.2f cycles per wave-instruction per SIMD (%.2f per instruction of one wave)\n", name, wps,
               avg / (iters * 32.0 * wps), avg / (iters * 32.0));
    }
}
int main() { run<0>("v_fma_f32"); run<1>("v_pk_fma_f32"); run<2>("v_add_f32"); run<3>("v_pk_add_f32"); run<4>("s_add_i32");
    run<5>("v_add_f32_dpp"); run<6>("v_fma_mix_f32"); run<7>("v_pk_add_f16"); run<8>("v_pk_mul_f32"); return 0; }
