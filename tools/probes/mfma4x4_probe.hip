// Probe of v_mfma_f32_4x4x1_16b_f32 operand layout (gfx950): 16 independent 4x4 blocks,
// block = lane / 4. Claim used by the quad kernel: for lane q of a block, result register i
//   D_i(q) = C_i(q) + A(lane i of the block) * B(lane q)
// i.e. B is the lane's own operand, A is broadcast across the 4 lanes by the register index.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma4x4_probe.hip -o build/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void probe(const float *a, const float *b, float *d) {
    const int l = threadIdx.x;
    v4f c = {100.f, 200.f, 300.f, 400.f};
    v4f r = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[4 * l + i] = r[i];
}

int main() {
    std::vector<float> a(64), b(64), d(256);
    for (int l = 0; l < 64; ++l) { a[l] = 1.f + l; b[l] = 0.5f + 0.25f * l; }
    float *da, *db, *dd;
    if (hipMalloc(&da, 256) != hipSuccess || hipMalloc(&db, 256) != hipSuccess || hipMalloc(&dd, 1024) != hipSuccess) return 2;
    hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(da, db, dd);
    if (hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            const float want = 100.f * (i + 1) + a[(l & ~3) + i] * b[l];
            if (d[4 * l + i] != want) { if (bad < 6) printf("lane %d reg %d: got %g want %g\n", l, i, d[4 * l + i], want); ++bad; }
        }
    printf("D_i(q) = C_i(q) + A(lane i) * B(lane q), block = lane/4: %s\n", bad ? "MISMATCH" : "ok");
    return bad ? 1 : 0;
}
