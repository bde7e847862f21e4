// Probe: raw buffer b128 / b32 loads and stores through a buffer resource on gfx950 (interleaved record layout)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k(float *ws, float *out, int T) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(ws + (size_t)blockIdx.x * 1024 * T, 0, 1024 * T * 4, 0x00020000);
    const unsigned lane = threadIdx.x, inst = lane >> 2, q = lane & 3;
    const unsigned lo = inst * 16;
    for (int t = 0; t < T; ++t) {
        const int so = t * 4096;
        // chunk row c = q: store (inst, q, t) pattern
        u4 v = {__builtin_bit_cast(unsigned, float(1000 * t + 10 * inst + q)), __builtin_bit_cast(unsigned, 1.f), __builtin_bit_cast(unsigned, 2.f), __builtin_bit_cast(unsigned, 3.f)};
        __builtin_amdgcn_raw_buffer_store_b128(v, r, lo + q * 256, so, 0);
    }
    float acc = 0, acc2 = 0;
    for (int t = 0; t < T; ++t) {
        const int so = t * 4096;
        u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lo + q * 256, so, 0);
        acc += __builtin_bit_cast(float, v.x) + __builtin_bit_cast(float, v.y) + __builtin_bit_cast(float, v.z) + __builtin_bit_cast(float, v.w);
        acc2 += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lo + q * 256, so, 0)) + 6.f;
    }
    out[blockIdx.x * 128 + lane] = acc;
    out[blockIdx.x * 128 + 64 + lane] = acc2;
}
int main() {
    const int T = 3, NB = 2;
    float *ws, *out;
    hipMalloc(&ws, NB * 1024 * T * 4);
    hipMalloc(&out, NB * 128 * 4);
    hipMemset(ws, 0, NB * 1024 * T * 4);
    hipLaunchKernelGGL(k, dim3(NB), dim3(64), 0, 0, ws, out, T);
    std::vector<float> h(NB * 128);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < NB; ++b)
        for (int l = 0; l < 64; ++l) {
            float want = 0;
            for (int t = 0; t < T; ++t) want += 1000 * t + 10 * (l >> 2) + (l & 3) + 6;
            if (h[b * 128 + l] != want || h[b * 128 + 64 + l] != want) { if (bad < 8) printf("lane %d: b128 %g b32 %g want %g\n", l, h[b * 128 + l], h[b * 128 + 64 + l], want); ++bad; }
        }
    printf("bad=%d\n", bad);
    return bad != 0;
}
