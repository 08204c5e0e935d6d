// Fallback CT encoder of the reference (final_multimodal.py:75-86, partial_modality_training.py:179-190,
// simple_fusion.py:191-202): 3 x [Conv3d(k3,s2,p1)+BN3d+ReLU] + global average pool.  Correctness-first kernels
// (VALU, LDS-staged activations); this path is 15x fewer FLOPs than DenseNet121 and is not the headline config.
#include "common.h"
#include <string.h>

// activation a(m_in, cin) = has_bn ? relu(bn(x)) : x
struct FbAct {
    float mu, sc, be; int has;
    __device__ __forceinline__ float operator()(float x) const { return has ? fmaxf(bn_apply(x, mu, sc, be), 0.f) : x; }
};
__device__ __forceinline__ FbAct fb_act(const FbConvP& p, int cin) {
    FbAct a; a.has = p.has_bn; a.mu = 0; a.sc = 1; a.be = 0;
    if (p.has_bn) { float rs; bn_mean_rstd(p.bn, cin, a.mu, rs); a.sc = p.bn.gamma[cin] * rs; a.be = p.bn.beta[cin]; }
    return a;
}

// ---- forward: one workgroup = VB output voxels x Cout; the 27*Cin activation patch of each voxel is staged in LDS
__global__ __launch_bounds__(256) void fb_conv_fwd_kernel(const FbConvP p) {
    extern __shared__ float patch[];            // [VB][27*Cin]
    __shared__ double red[2][256];
    const int VB = 256 / p.Cout, K = 27 * p.Cin;
    const int vox_out = p.out.D * p.out.H * p.out.W, Mout = p.B * vox_out;
    const int m0 = blockIdx.x * VB;
    for (int idx = threadIdx.x; idx < VB * K; idx += 256) {
        const int v = idx / K, k = idx % K, cin = k / 27, tap = k % 27, m = m0 + v;
        float a = 0.f;
        if (m < Mout) {
            const int b = m / vox_out, r = m % vox_out;
            const int od = r / (p.out.H * p.out.W), oh = (r / p.out.W) % p.out.H, ow = r % p.out.W;
            const int id = 2 * od - 1 + tap / 9, ih = 2 * oh - 1 + (tap / 3) % 3, iw = 2 * ow - 1 + tap % 3;
            if ((unsigned)id < (unsigned)p.in.D && (unsigned)ih < (unsigned)p.in.H && (unsigned)iw < (unsigned)p.in.W) {
                const size_t src = ((size_t)(b * p.in.D + id) * p.in.H + ih) * p.in.W + iw;
                a = fb_act(p, cin)(p.x[src * p.Cin + cin]);      // zero padding applies to the activated input
            }
        }
        patch[idx] = a;
    }
    __syncthreads();
    const int co = threadIdx.x % p.Cout, v = threadIdx.x / p.Cout, m = m0 + v;
    float acc = 0.f;
    if (m < Mout) {
        const float* wr = p.w + (size_t)co * K;
        const float* pr = patch + v * K;
        for (int k = 0; k < K; ++k) acc = fmaf(wr[k], pr[k], acc);
        acc += p.bias[co];
        p.y[(size_t)m * p.Cout + co] = acc;
    }
    if (p.osum) {
        red[0][threadIdx.x] = m < Mout ? acc : 0.0; red[1][threadIdx.x] = m < Mout ? (double)acc * acc : 0.0;
        __syncthreads();
        if (v == 0) {
            double s = 0, q = 0;
            for (int j = 0; j < VB; ++j) { s += red[0][j * p.Cout + co]; q += red[1][j * p.Cout + co]; }
            atomicAdd(&p.osum[co], s); atomicAdd(&p.osumsq[co], q);
        }
    }
}
extern "C" int mms_fb_conv_fwd(const FbConvP* pp, hipStream_t s) {
    const FbConvP& p = *pp;
    if (p.Cout <= 0 || 256 % p.Cout || p.Cin <= 0) return MMS_ERR_ARG;
    const int VB = 256 / p.Cout, Mout = p.B * p.out.D * p.out.H * p.out.W;
    MMS_LAUNCH(fb_conv_fwd_kernel, dim3((Mout + VB - 1) / VB), dim3(256), (size_t)VB * 27 * p.Cin * 4, s, p);
    return mms_check_launch();
}

// ---- backward, weights: thread per (cout, cin, tap), rows split over grid.y
__global__ __launch_bounds__(256) void fb_conv_bwd_w_kernel(const FbConvP p) {
    const int K = 27 * p.Cin, idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.Cout * K) return;
    const int co = idx / K, k = idx % K, cin = k / 27, tap = k % 27;
    const int vox_out = p.out.D * p.out.H * p.out.W, Mout = p.B * vox_out;
    const int mc = (Mout + p.msplit - 1) / p.msplit, mb = blockIdx.y * mc, me = mb + mc < Mout ? mb + mc : Mout;
    const FbAct act = fb_act(p, cin);
    float acc = 0.f, accb = 0.f;
    for (int m = mb; m < me; ++m) {
        const float g = p.dy[(size_t)m * p.Cout + co];
        accb += g;
        const int b = m / vox_out, r = m % vox_out;
        const int od = r / (p.out.H * p.out.W), oh = (r / p.out.W) % p.out.H, ow = r % p.out.W;
        const int id = 2 * od - 1 + tap / 9, ih = 2 * oh - 1 + (tap / 3) % 3, iw = 2 * ow - 1 + tap % 3;
        if ((unsigned)id < (unsigned)p.in.D && (unsigned)ih < (unsigned)p.in.H && (unsigned)iw < (unsigned)p.in.W) {
            const size_t src = ((size_t)(b * p.in.D + id) * p.in.H + ih) * p.in.W + iw;
            acc = fmaf(g, act(p.x[src * p.Cin + cin]), acc);
        }
    }
    atomicAdd(&p.dw[idx], acc);
    if (k == 0) atomicAdd(&p.dbias[co], accb);
}
extern "C" int mms_fb_conv_bwd_w(const FbConvP* pp, hipStream_t s) {
    const FbConvP& p = *pp;
    if (p.msplit <= 0) return MMS_ERR_ARG;
    MMS_LAUNCH(fb_conv_bwd_w_kernel, dim3((p.Cout * 27 * p.Cin + 255) / 256, p.msplit), dim3(256), 0, s, p);
    return mms_check_launch();
}

// ---- backward, input: thread per (m_in, cin); gather over the <= 8 (out voxel, tap) pairs that read this input voxel
__global__ __launch_bounds__(256) void fb_conv_bwd_x_kernel(const FbConvP p) {
    __shared__ double red[2][256];
    const int vox_in = p.in.D * p.in.H * p.in.W, Min = p.B * vox_in;
    const int VB = 256 / p.Cin, cin = threadIdx.x % p.Cin, v = threadIdx.x / p.Cin, m = blockIdx.x * VB + v;
    float g = 0.f, xh = 0.f;
    if (m < Min) {
        const int b = m / vox_in, r = m % vox_in;
        const int id = r / (p.in.H * p.in.W), ih = (r / p.in.W) % p.in.H, iw = r % p.in.W;
        float da = 0.f;
        for (int td = 0; td < 3; ++td) {
            const int nd = id + 1 - td;
            if (nd < 0 || (nd & 1) || (nd >> 1) >= p.out.D) continue;
            for (int th = 0; th < 3; ++th) {
                const int nh = ih + 1 - th;
                if (nh < 0 || (nh & 1) || (nh >> 1) >= p.out.H) continue;
                for (int tw = 0; tw < 3; ++tw) {
                    const int nw = iw + 1 - tw;
                    if (nw < 0 || (nw & 1) || (nw >> 1) >= p.out.W) continue;
                    const size_t mo = ((size_t)(b * p.out.D + (nd >> 1)) * p.out.H + (nh >> 1)) * p.out.W + (nw >> 1);
                    const int tap = (td * 3 + th) * 3 + tw;
                    const float* dyr = p.dy + mo * p.Cout;
                    for (int co = 0; co < p.Cout; ++co) da = fmaf(dyr[co], p.w[((size_t)co * p.Cin + cin) * 27 + tap], da);
                }
            }
        }
        float mu, rs;
        bn_mean_rstd(p.bn, cin, mu, rs);
        xh = (p.x[(size_t)m * p.Cin + cin] - mu) * rs;
        g = fmaf(p.bn.gamma[cin], xh, p.bn.beta[cin]) > 0.f ? da : 0.f;
        p.dbn_in[(size_t)m * p.Cin + cin] = g;
    }
    red[0][threadIdx.x] = g; red[1][threadIdx.x] = (double)g * xh;
    __syncthreads();
    if (v == 0) {
        double a = 0, c = 0;
        for (int j = 0; j < VB; ++j) { a += red[0][j * p.Cin + cin]; c += red[1][j * p.Cin + cin]; }
        atomicAdd(&p.s1[cin], a); atomicAdd(&p.s2[cin], c);
    }
}
extern "C" int mms_fb_conv_bwd_x(const FbConvP* pp, hipStream_t s) {
    const FbConvP& p = *pp;
    if (!p.has_bn || p.Cin <= 0 || 256 % p.Cin) return MMS_ERR_ARG;
    const int VB = 256 / p.Cin, Min = p.B * p.in.D * p.in.H * p.in.W;
    MMS_LAUNCH(fb_conv_bwd_x_kernel, dim3((Min + VB - 1) / VB), dim3(256), 0, s, p);
    return mms_check_launch();
}

// ---- BN + ReLU + global average pool (thread per (b, c)) and its backward
__global__ void fb_pool_fwd_kernel(const FbPoolP p) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.B * p.C) return;
    const int b = idx / p.C, c = idx % p.C;
    float mu, rs;
    bn_mean_rstd(p.bn, c, mu, rs);
    const float sc = p.bn.gamma[c] * rs, be = p.bn.beta[c];
    float a = 0.f;
    for (int v = 0; v < p.V; ++v) a += fmaxf(bn_apply(p.y[((size_t)b * p.V + v) * p.C + c], mu, sc, be), 0.f);
    p.out[(size_t)b * p.ldo + c] = a / (float)p.V;
}
__global__ void fb_pool_bwd_kernel(const FbPoolP p) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.B * p.C) return;
    const int b = idx / p.C, c = idx % p.C;
    float mu, rs;
    bn_mean_rstd(p.bn, c, mu, rs);
    const float ga = p.bn.gamma[c], be = p.bn.beta[c], d = p.dout[(size_t)b * p.lddout + c] / (float)p.V;
    double s1 = 0, s2 = 0;
    for (int v = 0; v < p.V; ++v) {
        const size_t o = ((size_t)b * p.V + v) * p.C + c;
        const float xh = (p.y[o] - mu) * rs;
        const float g = fmaf(ga, xh, be) > 0.f ? d : 0.f;
        p.dbn[o] = g;
        s1 += g; s2 += (double)g * xh;
    }
    atomicAdd(&p.s1[c], s1); atomicAdd(&p.s2[c], s2);
}
extern "C" int mms_fb_pool_fwd(const FbPoolP* pp, hipStream_t s) {
    MMS_LAUNCH(fb_pool_fwd_kernel, dim3((pp->B * pp->C + 255) / 256), dim3(256), 0, s, *pp);
    return mms_check_launch();
}
extern "C" int mms_fb_pool_bwd(const FbPoolP* pp, hipStream_t s) {
    MMS_LAUNCH(fb_pool_bwd_kernel, dim3((pp->B * pp->C + 255) / 256), dim3(256), 0, s, *pp);
    return mms_check_launch();
}

// ================================= whole-encoder driver =================================
namespace {
constexpr int FC[4] = {1, 32, 64, 128};
struct FbPlan {
    int B; Dims3 g[4]; int M[4];
    size_t y[4], dy[4], dbn[4], st[4], bb[4], tab_bn, stats_begin, stats_end, total;
};
inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
bool fb_plan(FbPlan& P, int B, int D, int H, int W) {
    if (B <= 0 || D < 1 || H < 1 || W < 1) return false;
    P.B = B; P.g[0] = Dims3{D, H, W};
    for (int l = 1; l < 4; ++l) P.g[l] = Dims3{(P.g[l - 1].D + 1) / 2, (P.g[l - 1].H + 1) / 2, (P.g[l - 1].W + 1) / 2};
    for (int l = 0; l < 4; ++l) P.M[l] = B * P.g[l].D * P.g[l].H * P.g[l].W;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = al(o + n); return r; };
    for (int l = 1; l < 4; ++l) { P.y[l] = take((size_t)P.M[l] * FC[l] * 4); P.dy[l] = take((size_t)P.M[l] * FC[l] * 4); P.dbn[l] = take((size_t)P.M[l] * FC[l] * 4); }
    P.tab_bn = take(sizeof(BnRunEntry) * 3);
    P.stats_begin = o;
    for (int l = 1; l < 4; ++l) { P.st[l] = take(2 * 128 * 8); P.bb[l] = take(2 * 128 * 8); }
    P.stats_end = o; P.total = o;
    return true;
}
template <class T> inline T* at(void* ws, size_t off) { return (T*)((char*)ws + off); }
inline BnSrc fb_bn(void* ws, const FbPlan& P, int l, const float* const* prm, const void* const* buf, int train) {
    BnSrc b;
    b.nrep = 0; b.rep_stride = 0;
    b.sum = at<double>(ws, P.st[l]); b.sumsq = b.sum + 128;
    b.rmean = buf ? (const float*)buf[3 * (l - 1)] : nullptr; b.rvar = buf ? (const float*)buf[3 * (l - 1) + 1] : nullptr;
    b.gamma = prm[4 * (l - 1) + 2]; b.beta = prm[4 * (l - 1) + 3];
    b.inv_count = 1.f / (float)P.M[l]; b.eps = 1e-5f; b.train = train;
    return b;
}
}  // namespace

extern "C" int mms_bn_running_update(const void*, int, float, hipStream_t);
extern "C" int mms_zero_regions_group(void* const*, int, size_t, hipStream_t);
extern "C" int mms_bn_bwd_apply(const BnBwdApplyP*, hipStream_t);
#define TRY(x) do { int rc_ = (x); if (rc_ != MMS_OK) return rc_; } while (0)

extern "C" int mms_fb_workspace_bytes(int B, int D, int H, int W, size_t* bytes) {
    FbPlan P;
    if (!fb_plan(P, B, D, H, W) || !bytes) return MMS_ERR_ARG;
    *bytes = P.total;
    return MMS_OK;
}
extern "C" int mms_fb_init(void* ws, int B, int D, int H, int W, const void* const* buffers, hipStream_t s) {
    FbPlan P;
    if (!fb_plan(P, B, D, H, W) || !ws || !buffers) return MMS_ERR_ARG;
    (void)hipGetLastError();
    BnRunEntry bn[3];
    for (int l = 1; l < 4; ++l) {
        bn[l - 1].sum = at<double>(ws, P.st[l]); bn[l - 1].sumsq = bn[l - 1].sum + 128;
        bn[l - 1].rmean = (float*)buffers[3 * (l - 1)]; bn[l - 1].rvar = (float*)buffers[3 * (l - 1) + 1];
        bn[l - 1].nbt = (long long*)buffers[3 * (l - 1) + 2]; bn[l - 1].C = FC[l]; bn[l - 1].count = (float)P.M[l];
        bn[l - 1].nrep = 0; bn[l - 1].rep_stride = 0;
    }
    if (hipMemcpyAsync(at<void>(ws, P.tab_bn), bn, sizeof(bn), hipMemcpyHostToDevice, s) != hipSuccess) return MMS_ERR_LAUNCH;
    if (hipStreamSynchronize(s) != hipSuccess) return MMS_ERR_LAUNCH;
    return MMS_OK;
}
extern "C" int mms_fb_forward(void* ws, int B, int D, int H, int W, const float* x, const void* const* params_,
                              const void* const* buffers, float* out, int ldo, int train, hipStream_t s) {
    FbPlan P;
    if (!fb_plan(P, B, D, H, W) || !ws || !x || !params_ || !out) return MMS_ERR_ARG;
    const float* const* prm = (const float* const*)params_;
    if (train) {    // (a zero-fill kernel, not hipMemsetAsync: consecutive memset nodes of a captured graph were observed to misorder)
        void* reg = at<void>(ws, P.stats_begin);
        TRY(mms_zero_regions_group(&reg, 1, P.stats_end - P.stats_begin, s));
    }
    for (int l = 1; l < 4; ++l) {
        FbConvP c{};
        c.x = l == 1 ? x : at<float>(ws, P.y[l - 1]); c.Cin = FC[l - 1]; c.in = P.g[l - 1]; c.out = P.g[l]; c.B = B;
        c.has_bn = l > 1; if (l > 1) c.bn = fb_bn(ws, P, l - 1, prm, buffers, train);
        c.w = prm[4 * (l - 1)]; c.bias = prm[4 * (l - 1) + 1]; c.Cout = FC[l];
        c.y = at<float>(ws, P.y[l]);
        c.osum = train ? at<double>(ws, P.st[l]) : nullptr; c.osumsq = train ? at<double>(ws, P.st[l]) + 128 : nullptr;
        TRY(mms_fb_conv_fwd(&c, s));
    }
    FbPoolP pl{};
    pl.y = at<float>(ws, P.y[3]); pl.C = 128; pl.V = P.M[3] / B; pl.B = B; pl.bn = fb_bn(ws, P, 3, prm, buffers, train);
    pl.out = out; pl.ldo = ldo;
    TRY(mms_fb_pool_fwd(&pl, s));
    if (train && buffers) TRY(mms_bn_running_update(at<void>(ws, P.tab_bn), 3, 0.1f, s));
    return MMS_OK;
}
extern "C" int mms_fb_backward(void* ws, int B, int D, int H, int W, const float* x, const void* const* params_,
                               const float* dout, int lddout, void* const* grads_, hipStream_t s) {
    FbPlan P;
    if (!fb_plan(P, B, D, H, W) || !ws || !x || !params_ || !dout || !grads_) return MMS_ERR_ARG;
    const float* const* prm = (const float* const*)params_;
    float* const* grd = (float* const*)grads_;
    FbPoolP pl{};
    pl.y = at<float>(ws, P.y[3]); pl.C = 128; pl.V = P.M[3] / B; pl.B = B; pl.bn = fb_bn(ws, P, 3, prm, nullptr, 1);
    pl.dout = dout; pl.lddout = lddout; pl.dbn = at<float>(ws, P.dbn[3]);
    pl.s1 = at<double>(ws, P.bb[3]); pl.s2 = pl.s1 + 128;
    TRY(mms_fb_pool_bwd(&pl, s));
    for (int l = 3; l >= 1; --l) {
        // BN_l backward: dbn_l -> dy_l (gradient w.r.t. the raw conv output), BN parameter grads
        BnBwdApplyP ap{at<float>(ws, P.dbn[l]), FC[l], at<float>(ws, P.y[l]), FC[l], at<float>(ws, P.dy[l]), FC[l], P.M[l], FC[l],
                       fb_bn(ws, P, l, prm, nullptr, 1), BnBwd{at<double>(ws, P.bb[l]), at<double>(ws, P.bb[l]) + 128, 0, 0}, 0,
                       grd[4 * (l - 1) + 2], grd[4 * (l - 1) + 3]};
        TRY(mms_bn_bwd_apply(&ap, s));
        FbConvP c{};
        c.x = l == 1 ? x : at<float>(ws, P.y[l - 1]); c.Cin = FC[l - 1]; c.in = P.g[l - 1]; c.out = P.g[l]; c.B = B;
        c.has_bn = l > 1; if (l > 1) c.bn = fb_bn(ws, P, l - 1, prm, nullptr, 1);
        c.w = prm[4 * (l - 1)]; c.Cout = FC[l]; c.dy = at<float>(ws, P.dy[l]);
        c.dw = grd[4 * (l - 1)]; c.dbias = grd[4 * (l - 1) + 1];
        c.msplit = P.M[l] >= 4096 ? 16 : (P.M[l] >= 512 ? 4 : 1);
        TRY(mms_fb_conv_bwd_w(&c, s));
        if (l > 1) {
            c.dbn_in = at<float>(ws, P.dbn[l - 1]); c.s1 = at<double>(ws, P.bb[l - 1]); c.s2 = c.s1 + 128;
            TRY(mms_fb_conv_bwd_x(&c, s));
        }
    }
    return MMS_OK;
}
