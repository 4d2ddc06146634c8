// Vector-memory INSTRUCTION throughput of a CU on gfx950: wave64 buffer loads / stores per CU and cycle for 4-, 8- and
// 16-byte accesses (contiguous lanes, a per-wave window that stays in L2), 4 / 8 / 16 waves per CU, whole-launch timing.
// Question behind it (round 5): the comb walker issues ~200 vector-memory instructions per CU and slot (q rows as 4-byte
// stores, guidance / image quads as 16-byte loads) -- does an instruction cost the CU's address path the same whatever
// its width (then q rows should leave as 8- or 16-byte stores), or do bytes count?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/vmem_rate.hip -o tools/ubench/vmem_rate && tools/ubench/vmem_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef int rsrc_t __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rs_t;
__device__ __forceinline__ rs_t mk_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// KIND 0/1/2: load b32 / b64 / b128; 3/4/5: store b32 / b64 / b128 (nt); 6: store b32 plain
template <int KIND>
__global__ __launch_bounds__(256) void k(char* buf, unsigned* out, int iters, unsigned win) {
    const unsigned wave = (blockIdx.x * 4u + (threadIdx.x >> 6));
    char* base = buf + (size_t)wave * win;
    const rs_t r = mk_rsrc(base, win);
    const unsigned lane = threadIdx.x & 63;
    constexpr unsigned W = KIND % 3 == 0 ? 4u : KIND % 3 == 1 ? 8u : 16u;
    unsigned acc = 0;
    const unsigned step = 64u * W, nst = win / step;
    unsigned pos = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned off = pos * step + lane * W;
            pos = pos + 1 == nst ? 0 : pos + 1;
            if (KIND == 0) acc += __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0);
            if (KIND == 1) { const u2 v = __builtin_bit_cast(u2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0)); acc += v.x ^ v.y; }
            if (KIND == 2) { const u4 v = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0)); acc += v.x ^ v.w; }
            if (KIND == 3) __builtin_amdgcn_raw_buffer_store_b32(acc + u, r, (int)off, 0, 2);
            if (KIND == 4) __builtin_amdgcn_raw_buffer_store_b64((u2){acc, (unsigned)u}, r, (int)off, 0, 2);
            if (KIND == 5) __builtin_amdgcn_raw_buffer_store_b128((u4){acc, (unsigned)u, lane, 1u}, r, (int)off, 0, 2);
            if (KIND == 6) __builtin_amdgcn_raw_buffer_store_b32(acc + u, r, (int)off, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int KIND> void run(const char* name, char* buf, unsigned* out) {
    int ncu = 256; (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const int iters = 500; const unsigned win = 4096;       // 4 KB per wave, reused: 16 waves x 32 CUs = 2 MB per XCD, inside its L2
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wpc = 4; wpc <= 16; wpc *= 2) {
        const int grid = ncu * wpc / 4;
        k<KIND><<<grid, 256>>>(buf, out, iters, win); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); k<KIND><<<grid, 256>>>(buf, out, iters, win); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double winst = (double)grid * 4 * iters * 8;
        const double per_cu_ns = winst / ncu / (ms * 1e6);
        const unsigned W = KIND % 3 == 0 ? 4u : KIND % 3 == 1 ? 8u : 16u;
        printf("%-18s %2d waves/CU: %7.3f ms  %.4f wave-instr per CU per ns = %.1f cycles per instr at 2.4 GHz, %.1f B per CU per cycle, %.2f TB/s chip\n", name, wpc, ms,
               per_cu_ns, 2.4 / per_cu_ns, per_cu_ns * 64 * W / 2.4, winst * 64 * W / (ms * 1e-3) / 1e12);
    }
}
int main() {
    char* buf; unsigned* out;
    (void)hipMalloc(&buf, (size_t)256 * 16 * 16384 + 65536); (void)hipMalloc(&out, 256 * 4 * 256 * 4);
    (void)hipMemset(buf, 1, (size_t)256 * 16 * 16384);
    run<0>("load b32", buf, out); run<1>("load b64", buf, out); run<2>("load b128", buf, out);
    run<6>("store b32", buf, out); run<3>("store b32 nt", buf, out); run<4>("store b64 nt", buf, out); run<5>("store b128 nt", buf, out);
    return 0;
}
