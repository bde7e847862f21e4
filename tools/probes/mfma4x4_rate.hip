// Issue rate of v_mfma_f32_4x4x1_16b_f32 for ONE wave per SIMD (the quad kernel's regime):
//  (a) 16 independent accumulators round-robin, MFMA only;
//  (b) the same with one independent VALU FMA between consecutive MFMAs;
//  (c) a dependent chain on one accumulator.
// Prints cycles per MFMA (s_memtime, wave 0 of a 256-CU x 4-wave launch so that every SIMD is busy).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0)

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

__global__ __launch_bounds__(64) void rate(float *out, unsigned long long *cyc, int iters) {
    const int l = threadIdx.x;
    v4f acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = v4f{(float)l, 1.f, 2.f, 3.f};
    float a = 1.0f + l * 1e-3f, b = 0.5f, x = 0.25f, y = 1.0f;
    unsigned long long t0 = now();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = MF(a, b, acc[i]);
    }
    unsigned long long t1 = now();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[i] = MF(a, b, acc[i]);
            y = __builtin_fmaf(y, x, a);
        }
    }
    unsigned long long t2 = now();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[0] = MF(a, b, acc[0]);
    }
    unsigned long long t3 = now();
    float s = y;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 64 + l] = s;
    if (blockIdx.x == 0 && l == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; }
}

int main() {
    float *out; unsigned long long *cyc, h[3];
    if (hipMalloc(&out, 1024 * 64 * 4) != hipSuccess || hipMalloc(&cyc, 24) != hipSuccess) return 2;
    const int iters = 2000;
    rate<<<1024, 64>>>(out, cyc, iters);
    rate<<<1024, 64>>>(out, cyc, iters);
    if (hipMemcpy(h, cyc, 24, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    const double n = 16.0 * iters;
    printf("cycles per v_mfma_f32_4x4x1 (one wave per SIMD): independent %.2f | with one VALU FMA each %.2f | dependent chain %.2f\n",
           h[0] / n, h[1] / n, h[2] / n);
    return 0;
}
