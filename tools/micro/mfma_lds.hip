// fp32 MFMA rate with the multi-tap kernels' operand pattern: A from LDS (ds_read_b128 feeding 4 MFMAs), optional mask multiply,
// B from registers; no global memory traffic.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MASK, int LDSA, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float mkin) {
    __shared__ __attribute__((aligned(16))) float img[98 * 132];
    for (int i = threadIdx.x; i < 98 * 132; i += 256) img[i] = i * 1e-4f;
    __syncthreads();
    const int lane = threadIdx.x & 63, li = lane & 31, kq = lane >> 5, wave = threadIdx.x >> 6;
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float4 b[8];
    for (int q = 0; q < 8; ++q) b[q] = make_float4(1.f + q, 2.f, 3.f, 4.f + lane);
    const float mk = mkin;
    for (int it = 0; it < iters; ++it) {
        const float* ar = img + ((wave >> 1) * 32 + li + 9 + (it % 3 - 1) * 8 + (it % 5 & 1)) * 132 + 64 * (wave & 1) + 4 * kq;
        float4 a[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = LDSA ? *(const float4*)(ar + 8 * q) : make_float4(mk + q, mk, 1.f, 2.f);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(MASK ? a[q].x * mk : a[q].x, b[q].x, acc[0], 0, 0, 0);
            acc[NACC - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(MASK ? a[q].y * mk : a[q].y, b[q].y, acc[NACC - 1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(MASK ? a[q].z * mk : a[q].z, b[q].z, acc[0], 0, 0, 0);
            acc[NACC - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(MASK ? a[q].w * mk : a[q].w, b[q].w, acc[NACC - 1], 0, 0, 0);
        }
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 123.456f) out[0] = s;
}
template <int MASK, int LDSA, int NACC> void run(int wgs, int iters, float* d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MASK, LDSA, NACC>), dim3(wgs), dim3(256), 0, 0, d, iters, 1.0f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MASK, LDSA, NACC>), dim3(wgs), dim3(256), 0, 0, d, iters, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double mfmas = (double)wgs * 4 * iters * 32;
    printf("mask=%d ldsA=%d nacc=%d wgs=%d: %.3f ms, %.1f TFLOP/s\n", MASK, LDSA, NACC, wgs, ms, mfmas * 4096 / ms / 1e9);
}
int main() {
    float* d; (void)hipMalloc(&d, 4);
    run<0, 0, 2>(512, 1024, d); run<1, 0, 2>(512, 1024, d); run<0, 1, 2>(512, 1024, d); run<1, 1, 2>(512, 1024, d); run<1, 1, 1>(512, 1024, d);
    run<1, 1, 2>(256, 1024, d); run<1, 1, 2>(768, 1024, d);
    return 0;
}
