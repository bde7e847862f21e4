// HBM throughput of the quad kernel's access pattern against an instance-interleaved one, in the
// kernel's regime: 1024 wavefronts (one per SIMD), each serving 16 instances x 4 lanes, walking
// T stages of RECW-word records; per stage every quad reads its record in 64-byte pieces
// (16 bytes per lane) and writes 6 of the 22 pieces back (WR=1: every 4th piece; WR=2: the first
// six, contiguous, like the factor's slot of the record).
//   layout 0 "per instance":  word w of record (i, t) at ((i*T + t)*RECW + w): a wave-wide
//                             instruction touches 16 pieces 28 KB apart (what ships today);
//   layout 2 "stage-major":   record (i, t) at ((t*B + i)*RECW): the 16 records a wave touches in a
//                             stage are one contiguous 22.5 KB block;
//   layout 1 "interleaved":   the 16 instances of a wave interleaved at 64-byte granularity:
//                             a wave-wide instruction touches 1 KB of contiguous memory.
// Prints GB/s (bytes read + written over the event time) for both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int T = 20, RECW = 352, PIECES = RECW / 16;

template <int LAYOUT, int WR>
__global__ __launch_bounds__(64) void walk(float *ws, int B, int passes, float *sink) {
    const int lane = threadIdx.x, q = lane & 3, iq = lane >> 2;
    const size_t wave = blockIdx.x;
    const size_t i = wave * 16 + iq;
    if (i >= (size_t)B) return;
    float acc = 0;
    for (int p = 0; p < passes; ++p) {
        for (int tt = 0; tt < T; ++tt) {
            const int t = (p & 1) ? T - 1 - tt : tt;   // forward then backward, as the solver sweeps
            float4 v[PIECES];
#pragma unroll
            for (int c = 0; c < PIECES; ++c) {
                const size_t w = LAYOUT == 0 ? (i * T + t) * RECW + 16 * c + 4 * q
                               : LAYOUT == 2 ? ((size_t)t * B + i) * RECW + 16 * c + 4 * q
                                             : ((wave * T + t) * PIECES + c) * 256 + iq * 16 + 4 * q;
                v[c] = *reinterpret_cast<const float4 *>(ws + w);
            }
#pragma unroll
            for (int c = 0; c < PIECES; ++c) {
                acc += v[c].x + v[c].y + v[c].z + v[c].w;
                if ((WR == 1 && (c & 3) == 0) || (WR == 2 && c < 6)) {
                    const size_t w = LAYOUT == 0 ? (i * T + t) * RECW + 16 * c + 4 * q
                                   : LAYOUT == 2 ? ((size_t)t * B + i) * RECW + 16 * c + 4 * q
                                                 : ((wave * T + t) * PIECES + c) * 256 + iq * 16 + 4 * q;
                    float4 o = v[c];
                    o.x += 1.0f;
                    *reinterpret_cast<float4 *>(ws + w) = o;
                }
            }
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int LAYOUT, int WR>
static double run(float *ws, int B, int passes, float *sink) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int grid = (B + 15) / 16;
    walk<LAYOUT, WR><<<grid, 64>>>(ws, B, 2, sink);   // warm-up
    CK(hipEventRecord(a));
    walk<LAYOUT, WR><<<grid, 64>>>(ws, B, passes, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)B * T * RECW * 4 * passes * (1.0 + (WR ? 0.25 * (double)((PIECES + 3) / 4 * 4) / PIECES : 0.0));
    return bytes / (ms * 1e-3) / 1e9;
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 16384, passes = 16;
    const size_t words = (size_t)((B + 15) / 16) * 16 * T * RECW;   // both layouts fit
    float *ws, *sink;
    CK(hipMalloc(&ws, words * 4));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(ws, 0, words * 4));
    printf("B=%d T=%d RECW=%d (%.0f MB per pass), %d passes\n", B, T, RECW, words * 4 / 1e6, passes);
    printf("read only : per-instance %.0f GB/s   interleaved %.0f GB/s\n", run<0, 0>(ws, B, passes, sink), run<1, 0>(ws, B, passes, sink));
    printf("read + write every 4th piece : per-instance %.0f GB/s   interleaved %.0f GB/s\n", run<0, 1>(ws, B, passes, sink), run<1, 1>(ws, B, passes, sink));
    printf("read + write pieces 0..5     : per-instance %.0f GB/s   interleaved %.0f GB/s   stage-major %.0f GB/s\n", run<0, 2>(ws, B, passes, sink), run<1, 2>(ws, B, passes, sink), run<2, 2>(ws, B, passes, sink));
    printf("read only                    : stage-major %.0f GB/s\n", run<2, 0>(ws, B, passes, sink));
    CK(hipFree(ws));
    CK(hipFree(sink));
    return 0;
}
