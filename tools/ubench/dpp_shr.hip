// Micro-check (GPU box): semantics of the DPP forms smx_agg_v5.hip relies on.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/dpp_shr.hip -o tools/ubench/dpp_shr && tools/ubench/dpp_shr
// row_shr:1 with bound_ctrl (zero fill) fused into v_subrev_f32 / v_add_f32; wave_shr:1 (whole-wave shift).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* in, float* out) {
    const int l = threadIdx.x;
    float a = in[l], b = in[64 + l];
    float d0, d1, d2 = b;
    asm volatile("s_nop 1\n\tv_subrev_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d0) : "v"(a), "v"(b));
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d1) : "v"(a), "v"(b));
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(d2) : "v"(a));
    out[l] = d0; out[64 + l] = d1; out[128 + l] = d2;
}
int main() {
    float h[128], o[192], *di, *dout;
    for (int i = 0; i < 64; ++i) { h[i] = 100.0f + i; h[64 + i] = 1000.0f * (i + 1); }
    hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0, badw = 0;
    for (int i = 0; i < 64; ++i) {
        const float left = (i % 16) ? h[i - 1] : 0.0f;
        if (o[i] != h[64 + i] - left) ++bad;
        if (o[64 + i] != h[64 + i] + left) ++bad;
        const float wl = i ? h[64 + i] + h[i - 1] : h[64 + i];   // wave_shr:1, lane 0 keeps its value
        if (o[128 + i] != wl) ++badw;
    }
    printf("row_shr:1 zero-fill fused sub/add: %s; wave_shr:1: %s (lane 16 got %g, want %g)\n", bad ? "WRONG" : "ok",
           badw ? "WRONG/unsupported" : "ok", o[128 + 16], h[64 + 16] + h[15]);
    return bad ? 1 : 0;
}
