// 1x1x1 convolution of the dense layers (MONAI _DenseLayer.layers.{norm1,relu1,conv1}, oracle/densenet3d.py) on launches with FEW rows:
// dense blocks 2-4 of a fold group's step (1024 / 128 / 16 rows per model on 64x64x32 volumes).
//
// The tile-GEMM forms give such launches enough workgroups by splitting the K (input-channel) loop over workgroups and having the last
// arriver of a tile sum the published partials (Conv1FwdOp<.., KSPLIT>: publish, ticket, read back -- three dependent memory round trips
// behind the MFMAs), or run 32x32 tiles whose K loop is a chain of round trips.  Here a workgroup owns a 16-row x 16-column output tile and
// the WHOLE K range: both operand panels (16 x K each, <= 64 KB at K = 1024) are requested in ONE batch of loads, staged in LDS (the
// activation panel through BatchNorm1 + ReLU), and the K range is then consumed from LDS by the four waves (16-channel groups dealt round
// robin, v_mfma_f32_16x16x4_f32, two accumulators each) and summed through LDS.  One memory round trip before the MFMAs, none between
// them, no cross-workgroup hand-off; statistics as everywhere (fp64 column sums, one atomic pair per column and workgroup).
#include "dn_ops.h"
#include <stdlib.h>

namespace {

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
constexpr int C1S_NP = 16;          // operand pieces (float4) per thread and panel at K = 1024

__global__ __launch_bounds__(256) void conv1s_fwd_kernel(const Grp<Conv1FwdP> grp) {
    const Conv1FwdP& p = grp.p[blockIdx.z];
    const float* __restrict__ x = p.x;                 // kernel arguments read once (dn_c3s.hip: left in the kernarg segment they are
    const float* __restrict__ w = p.w;                 // re-read inside every predicated block)
    const int M = p.M, K = p.K, N = p.N, ldx = p.ldx;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, h = lane >> 4;
    const int m0 = blockIdx.x * 16, n0 = blockIdx.y * 16;
    const int P = K + 4;                               // LDS row pitch (floats)
    float* As = smem;                                  // [16][P]  relu(bn1(x)) panel
    float* Bs = As + 16 * P;                           // [16][P]  weight panel
    float* cmean = Bs + 16 * P;                        // [K] BatchNorm1 constants
    float* csc = cmean + K;
    float* cbeta = csc + K;

    // ---- one batch of loads: both panels (clamped addresses, branch-free) + the BatchNorm statistics / parameters
    const int kq = K >> 2, total = 16 * kq;            // float4 pieces per panel
    float4 ra[C1S_NP], rb[C1S_NP];
#pragma unroll
    for (int i = 0; i < C1S_NP; ++i) {
        if (256 * i < total) {                         // workgroup-uniform
            const int idx = tid + 256 * i, ic = idx < total ? idx : total - 1;
            const int r = ic / kq, k4 = (ic - r * kq) * 4;
            const int mr = m0 + r < M ? m0 + r : M - 1, nr = n0 + r < N ? n0 + r : N - 1;
            ra[i] = *(const float4*)(x + (size_t)mr * ldx + k4);
            rb[i] = *(const float4*)(w + (size_t)nr * K + k4);
        }
    }
    bn_consts_to_lds<4>(p.bn, K, tid, cmean, csc, cbeta);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < C1S_NP; ++i) {
        if (256 * i < total) {
            const int idx = tid + 256 * i;
            if (idx < total) {
                const int r = idx / kq, k4 = (idx - r * kq) * 4;
                const float z = m0 + r < M ? 1.f : 0.f, zb = n0 + r < N ? 1.f : 0.f;
                const float4 v = ra[i];
                *(float4*)&As[r * P + k4] = make_float4(z * fmaxf(bn_apply(v.x, cmean[k4], csc[k4], cbeta[k4]), 0.f),
                                                        z * fmaxf(bn_apply(v.y, cmean[k4 + 1], csc[k4 + 1], cbeta[k4 + 1]), 0.f),
                                                        z * fmaxf(bn_apply(v.z, cmean[k4 + 2], csc[k4 + 2], cbeta[k4 + 2]), 0.f),
                                                        z * fmaxf(bn_apply(v.w, cmean[k4 + 3], csc[k4 + 3], cbeta[k4 + 3]), 0.f));
                *(float4*)&Bs[r * P + k4] = make_float4(zb * rb[i].x, zb * rb[i].y, zb * rb[i].z, zb * rb[i].w);
            }
        }
    }
    __syncthreads();

    // ---- the K range from LDS: 16-channel groups g = wave, wave + 4, ...; element e of lane (row, h) is k = 16 g + 4 h + e for both operands
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const float* ar = As + li * P + 4 * h;
    const float* br = Bs + li * P + 4 * h;
    const int ng16 = K >> 4;
    for (int g = wave; g < ng16; g += 8) {
        const float4 a0 = *(const float4*)(ar + 16 * g), b0 = *(const float4*)(br + 16 * g);
        const bool two = g + 4 < ng16;
        const float4 a1 = two ? *(const float4*)(ar + 16 * (g + 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 b1 = two ? *(const float4*)(br + 16 * (g + 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
        acc0 = MFMA16(a0.x, b0.x, acc0); acc1 = MFMA16(a1.x, b1.x, acc1);
        acc0 = MFMA16(a0.y, b0.y, acc0); acc1 = MFMA16(a1.y, b1.y, acc1);
        acc0 = MFMA16(a0.z, b0.z, acc0); acc1 = MFMA16(a1.z, b1.z, acc1);
        acc0 = MFMA16(a0.w, b0.w, acc0); acc1 = MFMA16(a1.w, b1.w, acc1);
    }
    __syncthreads();                                   // the panels are dead: Cs aliases them
    float* Cs = smem;                                  // [4 waves][16][17]
#pragma unroll
    for (int r = 0; r < 4; ++r) Cs[(wave * 16 + 4 * h + r) * 17 + li] = acc0[r] + acc1[r];      // C/D: column = lane & 15, row = 4 (lane >> 4) + r
    __syncthreads();
    double* red = (double*)(smem + 4 * 16 * 17);       // [2][16][16] (offset 4352 bytes: 8-byte aligned)
    const int r = tid >> 4, c = tid & 15, m = m0 + r, n = n0 + c;
    const float v = (Cs[r * 17 + c] + Cs[(16 + r) * 17 + c]) + (Cs[(32 + r) * 17 + c] + Cs[(48 + r) * 17 + c]);
    const bool ok = m < M && n < N;
    if (ok) p.y[(size_t)m * p.ldy + n] = v;
    if (p.osum == nullptr) return;
    red[r * 16 + c] = ok ? (double)v : 0.0;
    red[256 + r * 16 + c] = ok ? (double)v * v : 0.0;
    __syncthreads();
    if (tid < 16 && n0 + tid < N) {
        double s = 0, q = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s += red[i * 16 + tid]; q += red[256 + i * 16 + tid]; }
        atomicAdd(&stat_rep(p.osum, p.srep, p.sstride)[n0 + tid], s);
        atomicAdd(&stat_rep(p.osumsq, p.srep, p.sstride)[n0 + tid], q);
    }
}

}  // namespace

// Driver-internal launcher (argument checks are mms_conv1_fwd_group's).
int mms_c1s_fwd(const Conv1FwdP* pp, int ng, hipStream_t s) {
    const Conv1FwdP& p = *pp;
    int smem = (2 * 16 * (p.K + 4) + 3 * p.K) * (int)sizeof(float);
    if (smem < 4 * 16 * 17 * 4 + 2 * 256 * 8) smem = 4 * 16 * 17 * 4 + 2 * 256 * 8;
    static std::once_flag attr_once;
    std::call_once(attr_once, [] { hipFuncSetAttribute((const void*)conv1s_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (2 * 16 * 1028 + 3 * 1024) * (int)sizeof(float)); });
    Grp<Conv1FwdP> a;
    if (!grp_fill(a, pp, ng, 1)) return MMS_ERR_ARG;
    MMS_LAUNCH(conv1s_fwd_kernel, dim3((p.M + 15) / 16, (p.N + 15) / 16, ng), dim3(256), smem, s, a);
    return mms_check_launch();
}
