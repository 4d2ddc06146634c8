// Does buffer_load_dwordx4 honour a 4-byte aligned (not 16-byte aligned) offset on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* in, unsigned* out) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 4096, 0x00020000);
    const int lane = threadIdx.x;
    u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 4, 0, 0);   // lane l loads dwords l .. l+3
    out[4 * lane + 0] = v.x; out[4 * lane + 1] = v.y; out[4 * lane + 2] = v.z; out[4 * lane + 3] = v.w;
}
int main() {
    unsigned h[1024], o[256];
    for (int i = 0; i < 1024; ++i) h[i] = i;
    unsigned *d, *e;
    hipMalloc(&d, 4096); hipMalloc(&e, 1024);
    hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, e);
    hipMemcpy(o, e, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (o[4 * l + j] != (unsigned)(l + j)) ++bad;
    printf("unaligned b128 buffer loads: %d wrong dwords; lane 1 got %u %u %u %u (want 1 2 3 4)\n", bad, o[4], o[5], o[6], o[7]);
    return 0;
}
