// SIMD THROUGHPUT of the VALU instructions the comb walker is made of, on gfx950: wave64 instructions per SIMD and
// cycle with 1, 2, 4, 8 waves resident per SIMD, measured over the WHOLE launch (hipEvents) -- not the duration of
// one wave, which is what valu_rate.hip times (the oldest wave of a SIMD wins the arbiter and finishes as if alone,
// so that figure says nothing about what four waves cost each other).  Resolves DESIGN.md section 6's contradiction:
// does a v_pk_*_f32 occupy a SIMD for one pass (4 cycles per wave64) or two?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/simd_rate.hip -o tools/ubench/simd_rate && tools/ubench/simd_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int iters) {
    float a[8]; f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.5f + i; p[i] = (f2){a[i], a[i] + 1}; }
    const float m = 1.0001f, c = 0.5f; const f2 m2 = {m, m}, c2 = {c, c};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
                if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[i]) : "v"(m2));
                if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
                if (KIND == 5) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(a[(i + 4) & 7]));
                if (KIND == 6) asm volatile("v_fma_mix_f32 %0, %0, %1, 0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(m));
                if (KIND == 7) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 8) asm volatile("v_min_f32 %0, |%0|, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 10) {       // the mix of a comb row: 5 pk, 4 dpp, 5 plain of 14
                    if (i < 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
                    else if (i < 5) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(a[(i + 4) & 7]));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                }
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += a[i] + p[i].x + p[i].y;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) { clk[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = t0; clk[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = t1; }
}
template <int KIND> void run(const char* name) {
    int ncu = 256; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    float* out; unsigned long long* clk; hipMalloc(&out, (size_t)ncu * 16 * 256 * 4); hipMalloc(&clk, (size_t)ncu * 16 * 4 * 16);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 8; wps *= 2) {       // waves per SIMD = workgroups of 4 waves per CU
        const int grid = ncu * wps;
        k<KIND><<<grid, 256>>>(out, clk, iters); hipDeviceSynchronize();
        hipEventRecord(e0); k<KIND><<<grid, 256>>>(out, clk, iters); hipEventRecord(e1); hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        static unsigned long long h[256 * 16 * 4 * 2];
        hipMemcpy(h, clk, sizeof(unsigned long long) * grid * 8, hipMemcpyDeviceToHost);
        unsigned long long lo = ~0ull, hi = 0; double own = 0;
        for (int i = 0; i < grid * 4; ++i) { if (h[2 * i] < lo) lo = h[2 * i]; if (h[2 * i + 1] > hi) hi = h[2 * i + 1]; own += (double)(h[2 * i + 1] - h[2 * i]); }
        own /= grid * 4;
        const double winst = (double)grid * 4 * iters * 32, simds = ncu * 4.0;
        const double span = (double)(hi - lo);      // s_memtime ticks from the first wave's start to the last wave's end
        printf("%-16s %d waves/SIMD: %8.3f ms  %.3f wave-instr per SIMD per ns | span %.0f ticks: %.2f ticks per wave-instr per SIMD; one wave's own loop: %.2f ticks per instr\n",
               name, wps, ms, winst / simds / (ms * 1e6), span, span / (winst / simds), own / (iters * 32.0));
    }
    printf("   (clock rate attribute %d kHz)\n", khz);
    hipFree(out); hipFree(clk);
}
int main() {
    run<0>("v_add_f32"); run<1>("v_fma_f32"); run<9>("v_mul_f32"); run<8>("v_min_f32 |a|"); run<2>("v_pk_add_f32"); run<3>("v_pk_mul_f32");
    run<4>("v_pk_fma_f32"); run<5>("v_add_f32_dpp"); run<6>("v_fma_mix_f32"); run<7>("v_pk_add_f16"); run<10>("comb-row mix");
    return 0;
}
