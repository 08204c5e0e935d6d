// Practical fp32 MFMA peak on this box: waves doing nothing but v_mfma_f32_32x32x2_f32.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 123.456f) out[0] = s;
}
template <int NACC> void run(int wgs, int iters, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(wgs), dim3(256), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(wgs), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfmas = (double)wgs * 4 * iters * 8 * NACC;
    printf("NACC=%d wgs=%d: %.3f ms, %.1f TFLOP/s, %.1f cycles/MFMA/SIMD at 2.4 GHz (waves per SIMD %.1f)\n", NACC, wgs, ms,
           mfmas * 4096 / ms / 1e9, ms * 1e-3 * 2.4e9 / (mfmas / 1024.0), wgs / 256.0);
}
int main() {
    float* d; hipMalloc(&d, 4);
    run<1>(256, 4096, d); run<2>(256, 2048, d); run<4>(256, 1024, d);
    run<1>(512, 4096, d); run<2>(512, 2048, d); run<1>(1024, 2048, d);
    return 0;
}
