// Probe of the 16-byte global->LDS DMA forms (gfx950 global_load_lds_dwordx4):
//  (a) per-lane 64-bit address, source only 4-byte aligned: LDS gets 16 B per lane, lane-linear;
//  (b) SGPR base + 32-bit VGPR byte offset (saddr form), unaligned base;
//  (c) lanes switched off by EXEC write nothing.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/glds_x4_probe.hip -o build/glds_x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds16_s(const void *sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

__global__ void probe(const float *in, float *out, int shift) {
    __shared__ __attribute__((aligned(16))) float lds[256 * 3];
    const int l = threadIdx.x;
    for (int i = l; i < 256 * 3; i += 64) lds[i] = -1.f;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)lds;
    // (a) lane l fetches the 4 words at in + 1 + 5*l  (4-byte aligned only)
    glds16(in + 1 + 5 * l, base);
    // (b) uniform base in + shift (shift odd), lane offset 16*l bytes
    glds16_s(in + shift, 16u * l, base + 256 * 4);
    // (c) only lanes < 10
    if (l < 10) glds16_s(in + shift, 16u * l, base + 512 * 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = l; i < 256 * 3; i += 64) out[i] = lds[i];
}

int main() {
    const int n = 2048;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *din, *dout;
    if (hipMalloc(&din, n * 4) != hipSuccess || hipMalloc(&dout, 768 * 4) != hipSuccess) return 2;
    if (hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess) return 2;
    probe<<<1, 64>>>(din, dout, 3);
    std::vector<float> o(768);
    if (hipMemcpy(o.data(), dout, 768 * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int bad_a = 0, bad_b = 0, bad_c = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            bad_a += o[4 * l + r] != (float)(1 + 5 * l + r);
            bad_b += o[256 + 4 * l + r] != (float)(3 + 4 * l + r);
            bad_c += o[512 + 4 * l + r] != (l < 10 ? (float)(3 + 4 * l + r) : -1.f);
        }
    printf("(a) x4, per-lane 4-byte-aligned source: %s\n", bad_a ? "MISMATCH" : "ok");
    printf("(b) x4, SGPR base + VGPR offset, unaligned base: %s\n", bad_b ? "MISMATCH" : "ok");
    printf("(c) EXEC-masked lanes write nothing: %s\n", bad_c ? "MISMATCH" : "ok");
    if (bad_a) { printf("    a: "); for (int i = 0; i < 12; ++i) printf("%g ", o[i]); printf("\n"); }
    if (bad_b) { printf("    b: "); for (int i = 256; i < 268; ++i) printf("%g ", o[i]); printf("\n"); }
    return (bad_a || bad_b || bad_c) ? 1 : 0;
}
