// Preprocessing upstream of the hot path, on the GPU:
//   * CT volume: min-max normalisation + order-1 resample to the network grid -- the reference does this per item on the
//     CPU with numpy + scipy.ndimage.zoom(order=1) (R/scripts/training/partial_modality_training.py:94-109,
//     simple_fusion.py:117-134, flexible_multimodal.py:118-128);
//   * RNA-seq counts: log2(count + 1) then per-gene z-score over the cohort (sklearn StandardScaler, ddof 0)
//     (R/scripts/preprocessing/preprocess_genomic.py:108-117).
#include "common.h"
#include <float.h>

// ---- per-volume min / max: stage 1 -> [blocks][2] partials, stage 2 (inside the resample kernel's prologue) ----
__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* __restrict__ x, long long n, float* __restrict__ part) {
    __shared__ float smin[4], smax[4];
    float lo = FLT_MAX, hi = -FLT_MAX;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) { const float v = x[i]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
        part[2 * blockIdx.x + 1] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    }
}

// scipy.ndimage.zoom(order=1, grid_mode=False): output index o maps to input coordinate o * (in - 1) / (out - 1)
// (0 when out == 1); linear interpolation between the two neighbours.  Normalisation (x - min) / (max - min + 1e-8) is
// affine, so it commutes with the interpolation and is applied to the interpolated value.
__global__ __launch_bounds__(256) void ct_resample_norm_kernel(const float* __restrict__ x, Dims3 in, float* __restrict__ out, Dims3 o,
                                                               const float* __restrict__ part, int nparts) {
    __shared__ float mm[2];
    if (threadIdx.x < 64) {
        float lo = FLT_MAX, hi = -FLT_MAX;
        for (int i = threadIdx.x; i < nparts; i += 64) { lo = fminf(lo, part[2 * i]); hi = fmaxf(hi, part[2 * i + 1]); }
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) { lo = fminf(lo, __shfl_xor(lo, s, 64)); hi = fmaxf(hi, __shfl_xor(hi, s, 64)); }
        if (threadIdx.x == 0) { mm[0] = lo; mm[1] = hi; }
    }
    __syncthreads();
    const float lo = mm[0], inv = 1.0f / (mm[1] - mm[0] + 1e-8f);
    const long long n = (long long)o.D * o.H * o.W;
    const double sd = o.D > 1 ? (double)(in.D - 1) / (o.D - 1) : 0.0, sh = o.H > 1 ? (double)(in.H - 1) / (o.H - 1) : 0.0,
                 sw = o.W > 1 ? (double)(in.W - 1) / (o.W - 1) : 0.0;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256) {
        const int ow = (int)(idx % o.W), oh = (int)((idx / o.W) % o.H), od = (int)(idx / ((long long)o.W * o.H));
        const double cd = od * sd, ch = oh * sh, cw = ow * sw;
        const int d0 = (int)cd, h0 = (int)ch, w0 = (int)cw;
        const int d1 = d0 + 1 < in.D ? d0 + 1 : d0, h1 = h0 + 1 < in.H ? h0 + 1 : h0, w1 = w0 + 1 < in.W ? w0 + 1 : w0;
        const float fd = (float)(cd - d0), fh = (float)(ch - h0), fw = (float)(cw - w0);
        auto at = [&](int d, int h, int w) { return x[((size_t)d * in.H + h) * in.W + w]; };
        const float c00 = at(d0, h0, w0) * (1 - fw) + at(d0, h0, w1) * fw, c01 = at(d0, h1, w0) * (1 - fw) + at(d0, h1, w1) * fw;
        const float c10 = at(d1, h0, w0) * (1 - fw) + at(d1, h0, w1) * fw, c11 = at(d1, h1, w0) * (1 - fw) + at(d1, h1, w1) * fw;
        const float v = (c00 * (1 - fh) + c01 * fh) * (1 - fd) + (c10 * (1 - fh) + c11 * fh) * fd;
        out[idx] = (v - lo) * inv;
    }
}

// x: one volume [inD][inH][inW] (device), out: [oD][oH][oW], scratch: >= 2 * 256 floats
extern "C" int mms_ct_preprocess(const float* x, int inD, int inH, int inW, float* out, int oD, int oH, int oW, float* scratch, hipStream_t s) {
    if (!x || !out || !scratch || inD <= 0 || inH <= 0 || inW <= 0 || oD <= 0 || oH <= 0 || oW <= 0) return MMS_ERR_ARG;
    const long long n = (long long)inD * inH * inW;
    int nb = (int)((n + 256 * 16 - 1) / (256 * 16));
    if (nb < 1) nb = 1;
    if (nb > 256) nb = 256;
    MMS_LAUNCH(minmax_partial_kernel, dim3(nb), dim3(256), 0, s, x, n, scratch);
    const long long no = (long long)oD * oH * oW;
    int ob = (int)((no + 255) / 256);
    if (ob > 2048) ob = 2048;
    MMS_LAUNCH(ct_resample_norm_kernel, dim3(ob), dim3(256), 0, s, x, Dims3{inD, inH, inW}, out, Dims3{oD, oH, oW}, (const float*)scratch, nb);
    return mms_check_launch();
}

// counts [n][g] row-major -> out [n][g] = zscore_over_rows(log2(count + 1)); zero-variance genes keep scale 1 (sklearn).
// one workgroup per 64 genes: 4 row-groups x 64 columns, fp64 accumulation
__global__ __launch_bounds__(256) void rna_log_zscore_kernel(const float* __restrict__ c, float* __restrict__ out, int n, int g) {
    __shared__ double ssum[4][64], ssq[4][64];
    __shared__ float smean[64], sinv[64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    double s1 = 0, s2 = 0;
    if (col < g)
        for (int r = rg; r < n; r += 4) { const double v = log2((double)c[(size_t)r * g + col] + 1.0); s1 += v; s2 += v * v; }
    ssum[rg][threadIdx.x & 63] = s1; ssq[rg][threadIdx.x & 63] = s2;
    __syncthreads();
    if (rg == 0) {
        const int j = threadIdx.x;
        const double a = ssum[0][j] + ssum[1][j] + ssum[2][j] + ssum[3][j], b = ssq[0][j] + ssq[1][j] + ssq[2][j] + ssq[3][j];
        const double m = a / n;
        double var = b / n - m * m;
        if (var < 0) var = 0;
        const double sd = sqrt(var);
        smean[j] = (float)m;
        sinv[j] = sd < 1e-12 ? 1.0f : (float)(1.0 / sd);       // (sklearn: scale_ == 0 -> 1)
    }
    __syncthreads();
    if (col < g)
        for (int r = rg; r < n; r += 4) {
            const float v = (float)log2((double)c[(size_t)r * g + col] + 1.0);
            out[(size_t)r * g + col] = (v - smean[threadIdx.x & 63]) * sinv[threadIdx.x & 63];
        }
}
extern "C" int mms_rna_log_zscore(const float* counts, float* out, int n, int g, hipStream_t s) {
    if (!counts || !out || n <= 0 || g <= 0) return MMS_ERR_ARG;
    MMS_LAUNCH(rna_log_zscore_kernel, dim3((g + 63) / 64), dim3(256), 0, s, counts, out, n, g);
    return mms_check_launch();
}
