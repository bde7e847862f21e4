// Probe: issue cost and dependent latency of the instructions the register-resident interior-point kernel is made of,
// for ONE wavefront alone on its SIMD (the situation that kernel runs in). Prints cycles per instruction.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/dp_latency_probe.hip -o build/dp_latency_probe && gpurun -- ./build/dp_latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 256
#define STR2(x) #x
#define STR(x) STR2(x)

#define TIMED(NAME, IDX, BODY)                                                             \
    {                                                                                      \
        long long t0 = __builtin_amdgcn_s_memtime();                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                 \
        t0 = __builtin_amdgcn_s_memtime();                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)\n\t.rept " STR(REP) "\n\t" BODY "\n\t.endr\n\ts_nop 0" \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(f0), "+v"(f1), "+v"(i0) : "v"(x), "v"(m), "v"(addr) : "memory"); \
        long long t1 = __builtin_amdgcn_s_memtime();                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                 \
        t1 = __builtin_amdgcn_s_memtime();                                                 \
        if (threadIdx.x == 0) out[IDX] = t1 - t0;                                          \
    }

__global__ void probe(long long *out, double *sink) {
    __shared__ double lds[1024];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    double a0 = 1, a1 = 2, a2 = 3, a3 = 4, x = 0.5 + threadIdx.x * 1e-9, m = 1e-9;
    float f0 = 1, f1 = 2;
    int i0 = threadIdx.x * 4;
    int addr = threadIdx.x * 8;
    // 0: dependent v_fmac_f64_dpp chain on one accumulator
    TIMED("dep dpp f64", 0, "v_fmac_f64_dpp %0, %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf")
    // 1: two accumulators alternating (2 instructions per rep)
    TIMED("2acc dpp f64", 1, "v_fmac_f64_dpp %0, %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %7, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf")
    // 2: four accumulators (4 per rep)
    TIMED("4acc dpp f64", 2, "v_fmac_f64_dpp %0, %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %7, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %7, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %7, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf")
    // 3: dependent plain v_fma_f64
    TIMED("dep fma f64", 3, "v_fma_f64 %0, %7, %8, %0")
    // 4: four independent plain v_fma_f64
    TIMED("4acc fma f64", 4, "v_fma_f64 %0, %7, %8, %0\n\tv_fma_f64 %1, %7, %8, %1\n\tv_fma_f64 %2, %7, %8, %2\n\tv_fma_f64 %3, %7, %8, %3")
    // 5: dependent v_rcp_f64 (a0 = rcp(a0))
    TIMED("dep rcp f64", 5, "v_rcp_f64 %0, %0")
    // 6: dependent v_fmac_f32_dpp; 7: four accumulators f32 (approximated with 2 regs + 2 more f64 halves unused)
    TIMED("dep dpp f32", 6, "v_fmac_f32_dpp %4, %5, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf")
    TIMED("2acc dpp f32", 7, "v_fmac_f32_dpp %4, %6, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %5, %6, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf")
    // 8: dependent ds_read_b64 (address from the loaded low word is not possible for f64: use i0 = lds index chain with ds_read_b32)
    TIMED("dep ds_read_b32", 8, "ds_read_b32 %6, %6\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %6, 0xffc, %6")
    // 9: ds_read_b64 issue (independent, no wait)
    TIMED("indep ds_read_b64", 9, "ds_read_b64 %0, %9")
    // 10: dependent ds_bpermute_b32
    TIMED("dep ds_bpermute", 10, "ds_bpermute_b32 %6, %6, %6\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %6, 0xfc, %6")
    // 11: s_nop 1 + dpp pairs (cost of the hazard pad): nop + one dpp fmac on 4 rotating accumulators
    TIMED("nop+dpp", 11, "s_nop 1\n\tv_fmac_f64_dpp %0, %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_fmac_f64_dpp %1, %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_fmac_f64_dpp %2, %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_fmac_f64_dpp %3, %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf")
    // 12: v_mov_b64_dpp dependent on a fmac result then used (bcast round trip): mov_dpp a1 <- a0 ; fmac a0 += a1*m
    TIMED("bcast+fma dep", 12, "s_nop 1\n\tv_mov_b64_dpp %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %0, %1, %8, %0")
    // 13: v_cndmask pair (fp64 select) dependent
    TIMED("dep mul f64", 13, "v_mul_f64 %0, %0, %7")
    sink[threadIdx.x] = a0 + a1 + a2 + a3 + f0 + f1 + i0;
}

int main() {
    long long *out; double *sink;
    hipMalloc(&out, 64 * sizeof(long long)); hipMalloc(&sink, 64 * sizeof(double));
    hipMemset(out, 0, 64 * sizeof(long long));
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, sink); hipDeviceSynchronize(); }
    long long h[64]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[] = {"dep v_fmac_f64_dpp", "2-acc v_fmac_f64_dpp (x2)", "4-acc v_fmac_f64_dpp (x4)", "dep v_fma_f64", "4-acc v_fma_f64 (x4)",
                           "dep v_rcp_f64", "dep v_fmac_f32_dpp", "2-acc v_fmac_f32_dpp (x2)", "dep ds_read_b32 round trip", "indep ds_read_b64 issue",
                           "dep ds_bpermute round trip", "s_nop1 + dpp fmac (x4)", "bcast mov_dpp + fma round trip", "dep v_mul_f64"};
    const int per[] = {1, 2, 4, 1, 4, 1, 1, 2, 1, 1, 1, 4, 1, 1};
    for (int i = 0; i < 14; ++i) printf("%-34s %8.2f cycles per rep, %6.2f per instruction-group member\n", names[i], (double)h[i] / REP, (double)h[i] / REP / per[i]);
    return 0;
}
