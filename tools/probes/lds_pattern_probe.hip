// lds_pattern_probe.hip - cycles per ds_read_b64 for the address patterns of the resident interior-point kernel's
// sweeps (one wavefront per CU, batches of 40 independent reads followed by one wait, like a sweep step's prefetch).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_pattern_probe.hip -o lds_pattern_probe && ./lds_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int NL = 40, REP = 2000;

template <int WIDTH>   // 1: ds_read_b64, 2: ds_read2_b64 (pairs), 4: 32-bit
__global__ __launch_bounds__(64) void k(const int *offs, long long *out, double *sink, int pattern) {
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 6656; i += 64) lds[i] = i;
    __syncthreads();
    int off[NL];
    for (int j = 0; j < NL; ++j) off[j] = offs[(pattern * NL + j) * 64 + threadIdx.x];
    double acc = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
        double v[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) v[j] = lds[off[j]];
#pragma unroll
        for (int j = 0; j < NL; ++j) asm volatile("" :: "v"(v[j]));
        acc += v[0];
        asm volatile("" ::: "memory");
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main() {
    const char *names[] = {"contiguous lane*1 (+j*64)", "row r*17+k, 4 groups alike", "column k*17+r, 4 groups alike",
                           "halves: top row / bottom column (other block)", "packed M rows, 4 groups alike",
                           "packed M rows, halves in slots 0 / 10", "all lanes one address", "16 distinct, stride 2 words"};
    const int NP = 8;
    std::vector<int> h(NP * NL * 64);
    for (int p = 0; p < NP; ++p)
        for (int j = 0; j < NL; ++j)
            for (int l = 0; l < 64; ++l) {
                const int r = l & 15, rc = r < 13 ? r : 12, bot = l >= 32, k = j % 13;
                int o = 0;
                if (p == 0) o = l + j * 64;
                if (p == 1) o = rc * 17 + k + (j / 13) * 221;
                if (p == 2) o = k * 17 + rc + (j / 13) * 221;
                if (p == 3) o = bot ? 9 * 221 + k * 17 + rc : rc * 17 + k;
                if (p == 4 || p == 5) {
                    const int kk = j % 12;
                    o = (rc > kk) ? rc * (rc - 1) / 2 + kk : 91;
                    o += 4199 + ((p == 5 && bot) ? 10 * 92 : 0) + (j / 12) * 92;
                }
                if (p == 6) o = j;
                if (p == 7) o = r * 2 + j * 32;
                h[(p * NL + j) * 64 + l] = o;
            }
    int *d; long long *out; double *sink;
    hipMalloc(&d, h.size() * 4); hipMalloc(&out, 8 * 1024); hipMalloc(&sink, 8 * 64 * 1024);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int waves = 1; waves <= 3; waves += 2)
        for (int p = 0; p < NP; ++p) {
            hipLaunchKernelGGL(k<1>, dim3(256 * waves), dim3(64), 53248, 0, d, out, sink, p);
            hipDeviceSynchronize();
            long long t; hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
            printf("%d wave(s)/CU  %-48s %6.2f cycles per ds_read_b64 (batch of %d, then wait)\n", waves, names[p], (double)t / REP / NL, NL);
        }
    return 0;
}
