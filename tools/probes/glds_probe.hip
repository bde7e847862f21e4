// Probe of the gfx950 global->LDS DMA semantics the quad kernel's stage prefetch relies on:
//  (1) per-lane source address, lane-linear LDS destination (M0 base + lane*4);
//  (2) the instruction's immediate offset: applies to the global address, the LDS address or both;
//  (3) an unrolled sequence with a different LDS base per instruction.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/glds_probe.hip -o build/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// one DMA wave-instruction: lane's dword at gsrc -> LDS byte address lds_dst + 4*lane
__device__ __forceinline__ void glds4(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4_off16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off offset:16\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ void probe(const float *in, float *out) {
    __shared__ float lds[64 * 8];
    const int l = threadIdx.x;
    for (int i = l; i < 64 * 8; i += 64) lds[i] = -1.f;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)lds;
    glds4(in + (l * 7) % 64, base);
    glds4_off16(in + (l * 7) % 64, base + 64 * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds4(in + 100 + 4 * i + (l & 3) + 16 * (l >> 2), base + (192 + 64 * i) * 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = l; i < 64 * 8; i += 64) out[i] = lds[i];
}

int main() {
    const int n = 1024;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *din, *dout;
    if (hipMalloc(&din, n * 4) != hipSuccess || hipMalloc(&dout, 512 * 4) != hipSuccess) return 2;
    if (hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess) return 2;
    probe<<<1, 64>>>(din, dout);
    std::vector<float> o(512);
    if (hipMemcpy(o.data(), dout, 512 * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int bad1 = 0, bad3 = 0;
    for (int l = 0; l < 64; ++l) bad1 += o[l] != (float)((l * 7) % 64);
    printf("(1) per-lane source, lane-linear dest: %s\n", bad1 ? "MISMATCH" : "ok");
    printf("(2) offset:16, M0 base = word 64; non-empty words of 64..191:");
    for (int l = 64; l < 192; ++l) if (o[l] != -1.f && (l < 72 || l > 124)) printf(" [%d]=%g", l, o[l]);
    printf("\n");
    for (int i = 0; i < 4; ++i)
        for (int l = 0; l < 64; ++l) bad3 += o[192 + 64 * i + l] != (float)(100 + 4 * i + (l & 3) + 16 * (l >> 2));
    printf("(3) unrolled sequence with moving LDS base: %s\n", bad3 ? "MISMATCH" : "ok");
    return (bad1 || bad3) ? 1 : 0;
}
