// 1x1x1 convolution of the dense layers (MONAI _DenseLayer.layers.{norm1,relu1,conv1}, oracle/densenet3d.py) on launches with FEW rows:
// dense blocks 2-4 of a fold group's step (1024 / 128 / 16 rows per model on 64x64x32 volumes).
//
// The tile-GEMM forms give such launches enough workgroups by splitting the K (input-channel) loop over workgroups and having the last
// arriver of a tile sum the published partials (Conv1FwdOp<.., KSPLIT>: publish, ticket, read back -- three dependent memory round trips
// behind the MFMAs), or run 32x32 tiles whose K loop is a chain of round trips.  Here a workgroup owns a 16-row x 16-column output tile and
// the WHOLE K range: both operand panels (16 x K each, <= 64 KB at K = 1024) are requested in ONE batch of loads -- the activation panel
// staged in LDS through BatchNorm1 + ReLU, the weights straight into the MFMA register layout -- and the K range is then consumed by the
// four waves (16-channel groups dealt round robin, v_mfma_f32_16x16x4_f32, two accumulators each) and summed through LDS.  One memory round trip before the MFMAs, none between
// them, no cross-workgroup hand-off; statistics as everywhere (fp64 column sums, one atomic pair per column and workgroup).
#include "dn_ops.h"
#include <stdlib.h>

namespace {

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// C1S_NP: operand pieces (float4) per thread and panel -- 16 at K <= 1024 (228 VGPRs), 8 at K <= 512 (156 VGPRs: such a wave still fits
// on a SIMD that holds two waves of another stream's block-1 kernels, see mms_c3s_fwd)
template <int C1S_NP>
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
    float* cmean = As + 16 * P;                        // [K] BatchNorm1 constants
    float* csc = cmean + K;
    float* cbeta = csc + K;

    // ---- one batch of loads: the activation panel (clamped addresses, branch-free), this wave's weight fragments, the BatchNorm
    // statistics / parameters.  The weights go straight into the MFMA register layout (lane (li, h), group g: W[n0 + li][16 g + 4 h ..
    // + 3]) -- only the activation panel, which needs the BatchNorm transform, is staged: half the LDS of a two-panel layout, so a
    // second workgroup -- or the other streams' workgroups -- fits beside this one on a CU (the step-ablation runs of
    // profiles/r03_step_ablation.txt: the kernels with the largest LDS footprints cost the step the most per microsecond of their own).
    const int kq = K >> 2, total = 16 * kq;            // float4 pieces of the panel
    const int ng16 = K >> 4;
    float4 ra[C1S_NP], rb[C1S_NP];
#pragma unroll
    for (int i = 0; i < C1S_NP; ++i) {
        if (256 * i < total) {                         // workgroup-uniform
            const int idx = tid + 256 * i, ic = idx < total ? idx : total - 1;
            const int r = ic / kq, k4 = (ic - r * kq) * 4;
            const int mr = m0 + r < M ? m0 + r : M - 1;
            ra[i] = *(const float4*)(x + (size_t)mr * ldx + k4);
        }
    }
    const float* wr = w + (size_t)(n0 + li < N ? n0 + li : N - 1) * K + 4 * h;
#pragma unroll
    for (int i = 0; i < C1S_NP; ++i) {
        if (wave + 4 * i < ng16) rb[i] = *(const float4*)(wr + 16 * (wave + 4 * i));      // wave-uniform
    }
    bn_consts_to_lds<4>(p.bn, K, tid, cmean, csc, cbeta);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < C1S_NP; ++i) {
        if (256 * i < total) {
            const int idx = tid + 256 * i;
            if (idx < total) {
                const int r = idx / kq, k4 = (idx - r * kq) * 4;
                const float z = m0 + r < M ? 1.f : 0.f;
                const float4 v = ra[i];
                *(float4*)&As[r * P + k4] = make_float4(z * fmaxf(bn_apply(v.x, cmean[k4], csc[k4], cbeta[k4]), 0.f),
                                                        z * fmaxf(bn_apply(v.y, cmean[k4 + 1], csc[k4 + 1], cbeta[k4 + 1]), 0.f),
                                                        z * fmaxf(bn_apply(v.z, cmean[k4 + 2], csc[k4 + 2], cbeta[k4 + 2]), 0.f),
                                                        z * fmaxf(bn_apply(v.w, cmean[k4 + 3], csc[k4 + 3], cbeta[k4 + 3]), 0.f));
            }
        }
    }
    __syncthreads();

    // ---- the K range: 16-channel groups g = wave, wave + 4, ...; element e of lane (row, h) is k = 16 g + 4 h + e for both operands
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const float* ar = As + li * P + 4 * h;
    const float zb = n0 + li < N ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < C1S_NP; i += 2) {
        const int g = wave + 4 * i;
        if (g < ng16) {                                // wave-uniform
            const bool two = g + 4 < ng16;
            const float4 a0 = *(const float4*)(ar + 16 * g);
            const float4 a1 = two ? *(const float4*)(ar + 16 * (g + 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 b0 = rb[i];
            const float4 b1 = two ? rb[i + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
            acc0 = MFMA16(a0.x, zb * b0.x, acc0); acc1 = MFMA16(a1.x, zb * b1.x, acc1);
            acc0 = MFMA16(a0.y, zb * b0.y, acc0); acc1 = MFMA16(a1.y, zb * b1.y, acc1);
            acc0 = MFMA16(a0.z, zb * b0.z, acc0); acc1 = MFMA16(a1.z, zb * b1.z, acc1);
            acc0 = MFMA16(a0.w, zb * b0.w, acc0); acc1 = MFMA16(a1.w, zb * b1.w, acc1);
        }
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

// ---- backward-data of the same layers when ONE workgroup can own every row of its channels (M <= 128: dense block 3 of 64x64x32
// volumes, block 4's per-layer path): norm2 backward folded into the gradient operand, conv1 backward-data, relu1 mask, and -- the two
// BatchNorm1-backward sums being complete inside the workgroup -- norm1's backward applied in place on the slab gradient.  Replaces the
// tile-GEMM launch (32-row tiles, fp64 atomics, dbn scratch) AND the mms_bn_bwd_apply launch behind it (Conv1BwdP.fuse_dx semantics,
// same arithmetic).  A workgroup owns 16 input channels: the whole gradient panel dy'[M][128] (norm2 backward applied at LDS-store
// time), its 128 x 16 weight slice, its x / dx values -- every global load of the kernel is issued in one batch before the first use.
constexpr int C1SB_P = 132;         // LDS pitch of the [rows][128] gradient panel and the [16][128] transposed weight slice

__global__ __launch_bounds__(256) void conv1s_bwd_kernel(const Grp<Conv1BwdP> grp) {
    const Conv1BwdP& p = grp.p[blockIdx.z];
    const float* __restrict__ dyraw = p.dyraw;         // kernel arguments read once (see conv1s_fwd_kernel)
    const float* __restrict__ yf = p.y;
    const float* __restrict__ x = p.x;
    const float* __restrict__ w = p.w;
    float* __restrict__ dx = p.fuse_dx;
    const int M = p.M, K = p.K, lddy = p.lddy, ldy = p.ldy, ldx = p.ldx, lddx = p.fuse_lddx, accum = p.fuse_accumulate;
    const float inv_in = p.bn_in.inv_count, inv_out = p.bn_out.inv_count;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, h = lane >> 4;
    const int k0 = blockIdx.y * 16;
    const int rows16 = (M + 15) & ~15;
    float* Ds = smem;                                  // [rows16][C1SB_P] dy' panel
    float* Wt = Ds + rows16 * C1SB_P;                  // [16][C1SB_P]     W[n][k0 + c] transposed: Wt[c][n]
    float* dcA = Wt + 16 * C1SB_P;                     // [128] x 5: norm2-backward constants per mid channel n
    float* dcB = dcA + 128; float* dcC = dcB + 128; float* dcM = dcC + 128; float* dcR = dcM + 128;
    float* ein = dcR + 128;                            // [4][16] norm1 (mean, rstd, gamma, beta) of the 16 owned channels
    double* red = (double*)(ein + 64);                 // [4 waves][2][16]

    // ---- one batch of loads
    float4 rg[16], ry[16];                             // piece i: rows 8 i .. 8 i + 7 of the panel
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (8 * i < rows16) {                          // workgroup-uniform
            const int idx = tid + 256 * i, r = idx >> 5, c4 = (idx & 31) * 4, mr = r < M ? r : M - 1;
            rg[i] = *(const float4*)(dyraw + (size_t)mr * lddy + c4);
            ry[i] = *(const float4*)(yf + (size_t)mr * ldy + c4);
        }
    }
    float4 rw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;
        rw[i] = *(const float4*)(w + (size_t)(idx >> 2) * K + k0 + (idx & 3) * 4);
    }
    float xv[8], ov[8];                                // this lane's accumulator positions: row = 16 (2 wave + t) + 4 h + r, channel k0 + li
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = 16 * (2 * wave + (j >> 2)) + 4 * h + (j & 3), mr = r < M ? r : M - 1;
        xv[j] = x[(size_t)mr * ldx + k0 + li];
        ov[j] = dx[(size_t)mr * lddx + k0 + li];
    }
    if (tid < 128) {
        float mu, rs, ga;
        double t1, t2;
        bn_bwd_consts(p.bn_out, p.bb_out, tid, mu, rs, ga, t1, t2);
        dcA[tid] = ga * rs; dcB[tid] = (float)(t1 * (double)inv_out); dcC[tid] = (float)(t2 * (double)inv_out);
        dcM[tid] = mu; dcR[tid] = rs;
    } else if (tid < 144) {
        const int c = tid - 128;
        float mu, rs, ga, be;
        bn_consts1(p.bn_in, k0 + c, mu, rs, ga, be);
        ein[c] = mu; ein[16 + c] = rs; ein[32 + c] = ga; ein[48 + c] = be;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (8 * i < rows16) {
            const int idx = tid + 256 * i, r = idx >> 5, c4 = (idx & 31) * 4;
            const float z = r < M ? 1.f : 0.f;
            const float4 g = rg[i], yv = ry[i];
            // dy' = A (g - B - (y - mean) rstd C): DyConsts::dy of the tile-GEMM form (dn_bwd.hip)
            *(float4*)&Ds[r * C1SB_P + c4] =
                make_float4(z * dcA[c4] * (g.x - dcB[c4] - (yv.x - dcM[c4]) * dcR[c4] * dcC[c4]),
                            z * dcA[c4 + 1] * (g.y - dcB[c4 + 1] - (yv.y - dcM[c4 + 1]) * dcR[c4 + 1] * dcC[c4 + 1]),
                            z * dcA[c4 + 2] * (g.z - dcB[c4 + 2] - (yv.z - dcM[c4 + 2]) * dcR[c4 + 2] * dcC[c4 + 2]),
                            z * dcA[c4 + 3] * (g.w - dcB[c4 + 3] - (yv.w - dcM[c4 + 3]) * dcR[c4 + 3] * dcC[c4 + 3]));
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i, n = idx >> 2, c = (idx & 3) * 4;
        Wt[c * C1SB_P + n] = rw[i].x; Wt[(c + 1) * C1SB_P + n] = rw[i].y; Wt[(c + 2) * C1SB_P + n] = rw[i].z; Wt[(c + 3) * C1SB_P + n] = rw[i].w;
    }
    __syncthreads();

    // ---- da[m][k0 + c] = sum_n dy'[m][n] W[n][k0 + c]: a wave owns two 16-row tiles; element e of lane (li, h) in group g is n = 16 g + 4 h + e
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int t0 = 2 * wave;
    if (16 * t0 < M) {                                 // wave-uniform
        const bool two = 16 * (t0 + 1) < M;
        const float* a0p = Ds + (16 * t0 + li) * C1SB_P + 4 * h;
        const float* a1p = two ? a0p + 16 * C1SB_P : a0p;
        const float* bp = Wt + li * C1SB_P + 4 * h;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 a0 = *(const float4*)(a0p + 16 * g), a1 = *(const float4*)(a1p + 16 * g), b = *(const float4*)(bp + 16 * g);
            acc0 = MFMA16(a0.x, b.x, acc0); acc1 = MFMA16(a1.x, b.x, acc1);
            acc0 = MFMA16(a0.y, b.y, acc0); acc1 = MFMA16(a1.y, b.y, acc1);
            acc0 = MFMA16(a0.z, b.z, acc0); acc1 = MFMA16(a1.z, b.z, acc1);
            acc0 = MFMA16(a0.w, b.w, acc0); acc1 = MFMA16(a1.w, b.w, acc1);
        }
    }

    // ---- relu1 mask, the two norm1-backward sums over ALL rows (lanes li + 16 h, then the four waves), norm1 backward in place
    const float mu = ein[li], rs = ein[16 + li], ga = ein[32 + li], be = ein[48 + li];
    float gv[8];
    double s1 = 0, s2 = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = 16 * (t0 + (j >> 2)) + 4 * h + (j & 3);
        const float xh = (xv[j] - mu) * rs;
        const float a = j < 4 ? acc0[j & 3] : acc1[j & 3];
        const float g = (r < M && fmaf(ga, xh, be) > 0.f) ? a : 0.f;
        gv[j] = g;
        s1 += g; s2 += (double)g * xh;
    }
    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    if (h == 0) { red[(wave * 2) * 16 + li] = s1; red[(wave * 2 + 1) * 16 + li] = s2; }
    __syncthreads();
    double a = 0, b = 0;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) { a += red[(wv * 2) * 16 + li]; b += red[(wv * 2 + 1) * 16 + li]; }
    const float gr = ga * rs, m1 = (float)(a * (double)inv_in), m2 = rs * (float)(b * (double)inv_in);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = 16 * (t0 + (j >> 2)) + 4 * h + (j & 3);
        const float o = accum ? ov[j] : 0.f;
        if (r < M) dx[(size_t)r * lddx + k0 + li] = o + gr * (gv[j] - m1 - (xv[j] - mu) * m2);
    }
    if (wave == 0 && h == 0 && p.fuse_dgamma) { p.fuse_dgamma[k0 + li] += (float)b; p.fuse_dbeta[k0 + li] += (float)a; }
}

}  // namespace

// Driver-internal launcher (argument checks beyond these are mms_conv1_bwd_data_group's).
bool mms_conv1_small_bwd_ok(const Conv1BwdP& p, const MmsDnOpts& o) {
    if (o.conv1_small_bwd < 0) return false;
    return p.fuse_dx && !p.pool && p.has_bn_out && p.N == 128 && p.M >= 1 && p.M <= 128 && p.K % 16 == 0 && p.lddy % 4 == 0 && p.ldy % 4 == 0 &&
           (((uintptr_t)p.dyraw | (uintptr_t)p.y | (uintptr_t)p.w) & 15) == 0 && p.bn_in.train && p.bn_out.train;
}
int mms_c1s_bwd(const Conv1BwdP* pp, int ng, hipStream_t s) {
    const Conv1BwdP& p = *pp;
    for (int g = 0; g < ng; ++g) if (!mms_conv1_small_bwd_ok(pp[g], MmsDnOpts{})) return MMS_ERR_ARG;
    const int rows16 = (p.M + 15) & ~15;
    const int smem = ((rows16 + 16) * C1SB_P + 5 * 128 + 64) * (int)sizeof(float) + 4 * 2 * 16 * (int)sizeof(double);
    static std::once_flag attr_once;
    std::call_once(attr_once, [] { hipFuncSetAttribute((const void*)conv1s_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       ((128 + 16) * C1SB_P + 5 * 128 + 64) * (int)sizeof(float) + 4 * 2 * 16 * (int)sizeof(double)); });
    Grp<Conv1BwdP> a;
    if (!grp_fill(a, pp, ng, 1)) return MMS_ERR_ARG;
    MMS_LAUNCH(conv1s_bwd_kernel, dim3(1, p.K / 16, ng), dim3(256), smem, s, a);
    return mms_check_launch();
}

// Driver-internal launcher (argument checks are mms_conv1_fwd_group's).
int mms_c1s_fwd(const Conv1FwdP* pp, int ng, hipStream_t s) {
    const Conv1FwdP& p = *pp;
    int smem = (16 * (p.K + 4) + 3 * p.K) * (int)sizeof(float);
    if (smem < 4 * 16 * 17 * 4 + 2 * 256 * 8) smem = 4 * 16 * 17 * 4 + 2 * 256 * 8;
    static std::once_flag attr_once;
    std::call_once(attr_once, [] { hipFuncSetAttribute((const void*)(void (*)(const Grp<Conv1FwdP>))conv1s_fwd_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (16 * 1028 + 3 * 1024) * (int)sizeof(float)); });
    Grp<Conv1FwdP> a;
    if (!grp_fill(a, pp, ng, 1)) return MMS_ERR_ARG;
    const dim3 grid((p.M + 15) / 16, (p.N + 15) / 16, ng);
    if (p.K <= 512) {
        const auto kern = conv1s_fwd_kernel<8>;        // 16 x 516 + 3 x 512 floats = 39 KB: below the default dynamic-LDS limit
        MMS_LAUNCH(kern, grid, dim3(256), smem, s, a);
    } else {
        const auto kern = conv1s_fwd_kernel<16>;
        MMS_LAUNCH(kern, grid, dim3(256), smem, s, a);
    }
    return mms_check_launch();
}
