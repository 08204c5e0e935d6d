// Heads of the survival networks for gfx950: small-batch Linear layers with the neighbouring
// BatchNorm1d / ReLU / Dropout fused in, the softmax gate, the Cox partial likelihood, Harrell's C and the
// clip + Adam update.  Replaces the torch op sequences at final_multimodal.py:93-120,137-148,171-186,259-260;
// partial_modality_training.py:213-218,257-275,322-331; simple_fusion.py:167-178,206-215,47-73.
//
// Batch rows M <= 32: every lane keeps all M rows of "its" input column in registers, so BatchNorm1d batch
// statistics, their backward and the per-row dropout are lane-local (no atomics, no extra launches); weights
// are streamed once with coalesced loads (these layers are HBM/L2-bound: 2.7 M parameters, 4 rows).
#include "common.h"

// ------------------------------------------------------------------------------------------------------
// input prologue  x' = dropout(relu(bn1d(x)))  on the M values of one column, lane-local
// ------------------------------------------------------------------------------------------------------
template <int MM>
struct ColProlog {
    float xhat[MM], pre[MM], scale[MM];   // saved for the backward
    float mean, rstd, gamma;
    template <bool UPDATE_RUNNING>
    __device__ __forceinline__ void apply(const InProlog& pr, float (&x)[MM], int M, int k, int K) {
        if (pr.bn) {
            float mu, var;
            if (pr.train) {
                mu = 0;
#pragma unroll
                for (int m = 0; m < MM; ++m) if (m < M) mu += x[m];
                mu /= (float)M;
                var = 0;
#pragma unroll
                for (int m = 0; m < MM; ++m) if (m < M) { float d = x[m] - mu; var = fmaf(d, d, var); }
                var /= (float)M;
                if (UPDATE_RUNNING) {
                    const float unb = M > 1 ? var * (float)M / (float)(M - 1) : var;
                    pr.rmean[k] = (1.f - pr.momentum) * pr.rmean[k] + pr.momentum * mu;
                    pr.rvar[k] = (1.f - pr.momentum) * pr.rvar[k] + pr.momentum * unb;
                }
            } else {
                mu = pr.rmean[k]; var = pr.rvar[k];
            }
            mean = mu; rstd = 1.0f / sqrtf(var + pr.eps); gamma = pr.gamma[k];
            const float be = pr.beta[k];
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                xhat[m] = (x[m] - mu) * rstd;
                pre[m] = fmaf(gamma, xhat[m], be);
                x[m] = fmaxf(pre[m], 0.f);
            }
        }
#pragma unroll
        for (int m = 0; m < MM; ++m) scale[m] = 1.f;
        if (pr.train && (pr.drop_mask || pr.drop_p > 0.f)) {
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                if (m < M) {
                    scale[m] = pr.drop_mask ? pr.drop_mask[(size_t)m * K + k]
                                            : dropout_scale(pr.rng[0] + 0x9E3779B9u * pr.rng[1], pr.stream_id, (uint32_t)(m * K + k), pr.drop_p);
                    x[m] *= scale[m];
                }
            }
        }
    }
};

// One wave per output feature n (KS4 = false: four features per workgroup), or -- wide layers, K >= 2048: the 5005-wide RNA-seq input
// layers -- one WORKGROUP per feature whose four waves split the row (KS4: a wave's 20 KB weight row was a chain of K / (64 NU) memory round
// trips on 128 workgroups; 25.3 us per launch).  Columns are taken NU at a time: every (w, x[0..M)) load of the NU groups is requested
// before the first prologue / FMA.
template <int MM, bool KS4>
__global__ __launch_bounds__(256) void linear_fwd_kernel(const Grp<LinearFwdP> grp) {
    const LinearFwdP& p = grp.p[blockIdx.z];
    __shared__ float red[4][MM];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = KS4 ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
    if (n >= p.N) return;                                            // (KS4: workgroup-uniform)
    constexpr int KSTEP = KS4 ? 256 : 64;
    const bool upd = (n == 0) && p.pro.bn && p.pro.train;
    float acc[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) acc[m] = 0.f;
    int kstart = lane + (KS4 ? 64 * wave : 0);
    const float* __restrict__ wr = p.w + (size_t)n * p.K;
    const float* __restrict__ xr = p.x;
    const bool plain = !p.pro.bn && !(p.pro.train && (p.pro.drop_mask || p.pro.drop_p > 0.f));
    constexpr int NU = MM <= 4 ? 8 : (MM <= 8 ? 4 : 2);
    for (; kstart + KSTEP * (NU - 1) < p.K; kstart += KSTEP * NU) {
        float w4[NU], x4[NU][MM];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            w4[u] = wr[kstart + KSTEP * u];
#pragma unroll
            for (int m = 0; m < MM; ++m) x4[u][m] = m < p.M ? xr[(size_t)m * p.ldx + kstart + KSTEP * u] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (!plain) {
                ColProlog<MM> cp;
                if (upd) cp.template apply<true>(p.pro, x4[u], p.M, kstart + KSTEP * u, p.K); else cp.template apply<false>(p.pro, x4[u], p.M, kstart + KSTEP * u, p.K);
            }
#pragma unroll
            for (int m = 0; m < MM; ++m) acc[m] = fmaf(w4[u], x4[u][m], acc[m]);
        }
    }
    for (int k = kstart; k < p.K; k += KSTEP) {
        float x[MM];
#pragma unroll
        for (int m = 0; m < MM; ++m) x[m] = m < p.M ? xr[(size_t)m * p.ldx + k] : 0.f;
        ColProlog<MM> cp;
        if (upd) cp.template apply<true>(p.pro, x, p.M, k, p.K); else cp.template apply<false>(p.pro, x, p.M, k, p.K);
        const float w = wr[k];
#pragma unroll
        for (int m = 0; m < MM; ++m) acc[m] = fmaf(w, x[m], acc[m]);
    }
    if (upd && lane == 0 && (!KS4 || wave == 0) && p.pro.nbt) *p.pro.nbt += 1;
    const float b = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int m = 0; m < MM; ++m) acc[m] = wave_sum(acc[m]);
    if constexpr (KS4) {
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < MM; ++m) red[wave][m] = acc[m];
        }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int m = 0; m < MM; ++m) acc[m] = red[0][m] + red[1][m] + red[2][m] + red[3][m];
    }
    if (lane == 0) {
#pragma unroll
        for (int m = 0; m < MM; ++m)
            if (m < p.M) {
                const float v = acc[m] + b;
                p.y[(size_t)m * p.ldy + n] = p.out_relu ? fmaxf(v, 0.f) : v;
            }
    }
}

template <int MM>
static int launch_linear_fwd(const Grp<LinearFwdP>& a, int ng, hipStream_t s) {
    if (a.p[0].K >= 2048) MMS_LAUNCH((linear_fwd_kernel<MM, true>), dim3(a.p[0].N, 1, ng), dim3(256), 0, s, a);
    else MMS_LAUNCH((linear_fwd_kernel<MM, false>), dim3((a.p[0].N + 3) / 4, 1, ng), dim3(256), 0, s, a);
    return mms_check_launch();
}
extern "C" int mms_linear_fwd_group(const LinearFwdP* pp, int ng, hipStream_t s) {
    Grp<LinearFwdP> a;
    if (!grp_fill(a, pp, ng, 1)) return MMS_ERR_ARG;
    const LinearFwdP& p = *pp;
    if (p.M <= 0 || p.M > 32 || p.N <= 0 || p.K <= 0) return MMS_ERR_ARG;
    if (p.pro.bn && p.pro.train && p.M < 2) return MMS_ERR_ARG;   // torch: "Expected more than 1 value per channel"
    for (int g = 1; g < ng; ++g) {
        const LinearFwdP& q = pp[g];
        if (q.M != p.M || q.N != p.N || q.K != p.K || q.pro.bn != p.pro.bn || q.pro.train != p.pro.train) return MMS_ERR_ARG;
    }
    if (p.M <= 4) return launch_linear_fwd<4>(a, ng, s);
    if (p.M <= 8) return launch_linear_fwd<8>(a, ng, s);
    if (p.M <= 16) return launch_linear_fwd<16>(a, ng, s);
    return launch_linear_fwd<32>(a, ng, s);
}
MMS_SINGLE(mms_linear_fwd, LinearFwdP)

// ---- backward: weight/bias (one wave per output feature n) ---------------------------------------------------
template <int MM, bool KS4>
__global__ __launch_bounds__(256) void linear_bwd_w_kernel(const Grp<LinearBwdP> grp) {
    const LinearBwdP& p = grp.p[blockIdx.z];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = KS4 ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;      // KS4: as linear_fwd_kernel
    if (n >= p.N) return;
    constexpr int KSTEP = KS4 ? 256 : 64;
    float dz[MM], db = 0.f;
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        float g = 0.f;
        if (m < p.M) {
            g = p.dy[(size_t)m * p.lddy + n];
            if (p.out_relu && !(p.y[(size_t)m * p.ldy + n] > 0.f)) g = 0.f;
        }
        dz[m] = g; db += g;
    }
    if (lane == 0 && (!KS4 || wave == 0) && p.dbias) p.dbias[n] += db;
    // the gradient row is read-modify-written: NU independent (dw, x[0..M)) load groups are requested before the first store (one
    // group per trip was a chain of K / 64 dependent round trips: 56 us for the 5005-wide first layer)
    constexpr int NU = MM <= 4 ? 8 : 2;
    float* __restrict__ dwr = p.dw + (size_t)n * p.K;
    const float* __restrict__ xr = p.x;
    int k0 = lane + (KS4 ? 64 * wave : 0);
    for (; k0 + KSTEP * (NU - 1) < p.K; k0 += KSTEP * NU) {
        float d0[NU], xx[NU][MM];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            d0[u] = dwr[k0 + KSTEP * u];
#pragma unroll
            for (int m = 0; m < MM; ++m) xx[u][m] = m < p.M ? xr[(size_t)m * p.ldx + k0 + KSTEP * u] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            ColProlog<MM> cp;
            cp.template apply<false>(p.pro, xx[u], p.M, k0 + KSTEP * u, p.K);
            float a = 0.f;
#pragma unroll
            for (int m = 0; m < MM; ++m) a = fmaf(dz[m], xx[u][m], a);
            dwr[k0 + KSTEP * u] = d0[u] + a;
        }
    }
    for (int k = k0; k < p.K; k += KSTEP) {
        float x[MM];
#pragma unroll
        for (int m = 0; m < MM; ++m) x[m] = m < p.M ? p.x[(size_t)m * p.ldx + k] : 0.f;
        ColProlog<MM> cp;
        cp.template apply<false>(p.pro, x, p.M, k, p.K);
        float a = 0.f;
#pragma unroll
        for (int m = 0; m < MM; ++m) a = fmaf(dz[m], x[m], a);
        p.dw[(size_t)n * p.K + k] += a;
    }
}

// ---- backward: input.  A workgroup owns 64 input columns; its four waves split the output features (wave w takes 32 of every 128) and
// meet in LDS; the prologue backward is lane-local (wave 0, one lane per column).  A wave's 32 weight rows per chunk are requested 16 at a
// time before their first use: one thread per column over ALL N rows was a chain of N dependent-latency loads on two CUs (K = 512, N = 256:
// 18.6 us per launch; 47 us with an 8-deep unroll hint).
template <int MM>
__global__ __launch_bounds__(256) void linear_bwd_x_kernel(const Grp<LinearBwdP> grp) {
    const LinearBwdP& p = grp.p[blockIdx.z];
    __shared__ float dzs[MM][128];
    __shared__ float red[3][MM][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + lane;
    const int N = p.N, K = p.K;
    const float* __restrict__ wk = p.w + (k < K ? k : K - 1);        // clamped: branch-free loads
    float acc[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) acc[m] = 0.f;
    for (int nb = 0; nb < N; nb += 128) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < MM * 128; idx += 256) {
            const int m = idx >> 7, n = nb + (idx & 127);
            float g = 0.f;
            if (m < p.M && n < N) {
                g = p.dy[(size_t)m * p.lddy + n];
                if (p.out_relu && !(p.y[(size_t)m * p.ldy + n] > 0.f)) g = 0.f;
            }
            dzs[m][idx & 127] = g;                                   // rows n >= N: zero, so their (clamped) weight loads add nothing
        }
        __syncthreads();
        if (nb + 32 * wave < N) {                                    // wave-uniform
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float wv[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int n = nb + 32 * wave + 16 * h + j;
                    wv[j] = wk[(size_t)(n < N ? n : N - 1) * K];
                }
#pragma unroll
                for (int j = 0; j < 16; ++j)
#pragma unroll
                    for (int m = 0; m < MM; ++m) acc[m] = fmaf(dzs[m][32 * wave + 16 * h + j], wv[j], acc[m]);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int m = 0; m < MM; ++m) red[wave - 1][m][lane] = acc[m];
    }
    __syncthreads();
    if (wave > 0 || k >= K) return;
#pragma unroll
    for (int m = 0; m < MM; ++m) acc[m] += red[0][m][lane] + red[1][m][lane] + red[2][m][lane];
    float x[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) x[m] = m < p.M ? p.x[(size_t)m * p.ldx + k] : 0.f;
    ColProlog<MM> cp;
    cp.template apply<false>(p.pro, x, p.M, k, p.K);
    float dxv[MM];
    if (p.pro.bn) {
        float s1 = 0.f, s2 = 0.f, dpre[MM];
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            dpre[m] = (m < p.M && cp.pre[m] > 0.f) ? acc[m] * cp.scale[m] : 0.f;
            s1 += dpre[m]; s2 = fmaf(dpre[m], cp.xhat[m], s2);
        }
        if (p.dgamma) { p.dgamma[k] += s2; p.dbeta[k] += s1; }
        const float gr = cp.gamma * cp.rstd, m1 = s1 / (float)p.M, m2 = s2 / (float)p.M;
#pragma unroll
        for (int m = 0; m < MM; ++m)
            dxv[m] = p.pro.train ? gr * (dpre[m] - m1 - cp.xhat[m] * m2) : gr * dpre[m];
    } else {
#pragma unroll
        for (int m = 0; m < MM; ++m) dxv[m] = acc[m] * cp.scale[m];
    }
#pragma unroll
    for (int m = 0; m < MM; ++m) if (m < p.M) p.dx[(size_t)m * p.lddx + k] = dxv[m];
}

template <int MM>
static int launch_linear_bwd(const Grp<LinearBwdP>& a, int ng, hipStream_t s) {
    const LinearBwdP& p = a.p[0];
    if (p.dw) {
        if (p.K >= 2048) MMS_LAUNCH((linear_bwd_w_kernel<MM, true>), dim3(p.N, 1, ng), dim3(256), 0, s, a);
        else MMS_LAUNCH((linear_bwd_w_kernel<MM, false>), dim3((p.N + 3) / 4, 1, ng), dim3(256), 0, s, a);
    }
    if (p.dx) MMS_LAUNCH(linear_bwd_x_kernel<MM>, dim3((p.K + 63) / 64, 1, ng), dim3(256), 0, s, a);
    return mms_check_launch();
}
extern "C" int mms_linear_bwd_group(const LinearBwdP* pp, int ng, hipStream_t s) {
    Grp<LinearBwdP> a;
    if (!grp_fill(a, pp, ng, 1)) return MMS_ERR_ARG;
    const LinearBwdP& p = *pp;
    if (p.M <= 0 || p.M > 32 || p.N <= 0 || p.K <= 0) return MMS_ERR_ARG;
    for (int g = 1; g < ng; ++g) {
        const LinearBwdP& q = pp[g];
        if (q.M != p.M || q.N != p.N || q.K != p.K || (q.dw == nullptr) != (p.dw == nullptr) || (q.dx == nullptr) != (p.dx == nullptr)) return MMS_ERR_ARG;
    }
    if (p.M <= 4) return launch_linear_bwd<4>(a, ng, s);
    if (p.M <= 8) return launch_linear_bwd<8>(a, ng, s);
    if (p.M <= 16) return launch_linear_bwd<16>(a, ng, s);
    return launch_linear_bwd<32>(a, ng, s);
}
MMS_SINGLE(mms_linear_bwd, LinearBwdP)

// ------------------------------------------------------------------------------------------------------
// gated fusion: one workgroup per patient row
// ------------------------------------------------------------------------------------------------------
#define GATE_F 288
#define GATE_IN 291
__device__ __forceinline__ int gate_seg(int t) { return t < 128 ? 0 : (t < 256 ? 1 : 2); }

__global__ __launch_bounds__(256) void gate_fwd_kernel(const Grp<GateP> grp) {
    const GateP& p = grp.p[blockIdx.z];
    __shared__ float G[GATE_IN + 1], h[64], g[3];
    const int m = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < GATE_IN; i += 256)
        G[i] = i < GATE_F ? p.feats[(size_t)m * GATE_F + i] * p.mask[m * 3 + gate_seg(i)] : p.mask[m * 3 + (i - GATE_F)];
    __syncthreads();
    for (int j = wave * 16; j < wave * 16 + 16; ++j) {
        float a = 0.f;
        for (int k = lane; k < GATE_IN; k += 64) a = fmaf(p.w1[j * GATE_IN + k], G[k], a);
        a = wave_sum(a);
        if (lane == 0) { h[j] = fmaxf(a + p.b1[j], 0.f); p.hidden[m * 64 + j] = h[j]; }
    }
    __syncthreads();
    if (wave == 0) {
        float l[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) l[c] = wave_sum(p.w2[c * 64 + lane] * h[lane]) + p.b2[c];
        if (lane == 0) {
            const float mx = fmaxf(l[0], fmaxf(l[1], l[2]));
            const float e0 = expf(l[0] - mx), e1 = expf(l[1] - mx), e2 = expf(l[2] - mx), inv = 1.f / (e0 + e1 + e2);
            g[0] = e0 * inv; g[1] = e1 * inv; g[2] = e2 * inv;
            float ent = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) { p.gate[m * 3 + c] = g[c]; ent += g[c] * logf(g[c] + 1e-8f); }
            if (p.entropy) atomicAdd(p.entropy, ent / (float)p.M);     // gate_entropy_loss = -mean(entropy)
        }
    }
    __syncthreads();
    for (int i = t; i < GATE_F; i += 256) p.fused[(size_t)m * GATE_F + i] = G[i] * g[gate_seg(i)];
}

__global__ __launch_bounds__(256) void gate_bwd_kernel(const Grp<GateP> grp) {
    const GateP& p = grp.p[blockIdx.z];
    __shared__ float G[GATE_IN + 1], h[64], g[3], dgs[3], dl[3], dhp[64], red[4][3];
    const int m = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < GATE_IN; i += 256)
        G[i] = i < GATE_F ? p.feats[(size_t)m * GATE_F + i] * p.mask[m * 3 + gate_seg(i)] : p.mask[m * 3 + (i - GATE_F)];
    if (t < 64) h[t] = p.hidden[m * 64 + t];
    if (t < 3) g[t] = p.gate[m * 3 + t];
    __syncthreads();
    // dgate[c] = sum over segment c of dfused * masked
    float part[3] = {0.f, 0.f, 0.f};
    for (int i = t; i < GATE_F; i += 256) part[gate_seg(i)] += p.dfused[(size_t)m * GATE_F + i] * G[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) { float v = wave_sum(part[c]); if (lane == 0) red[wave][c] = v; }
    __syncthreads();
    if (t == 0) {
        float dg[3], dot = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            dg[c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
            if (p.ent_weight != 0.f) dg[c] += p.ent_weight / (float)p.M * (logf(g[c] + 1e-8f) + g[c] / (g[c] + 1e-8f));
            if (p.dgate_ext) dg[c] += p.dgate_ext[m * 3 + c];
            dot += g[c] * dg[c];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { dl[c] = g[c] * (dg[c] - dot); atomicAdd(&p.db2[c], dl[c]); }
    }
    __syncthreads();
    if (t < 64) {
        float dh = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) { dh = fmaf(dl[c], p.w2[c * 64 + t], dh); atomicAdd(&p.dw2[c * 64 + t], dl[c] * h[t]); }
        dhp[t] = h[t] > 0.f ? dh : 0.f;
        atomicAdd(&p.db1[t], dhp[t]);
    }
    __syncthreads();
    for (int idx = t; idx < 64 * GATE_IN; idx += 256) {
        const int j = idx / GATE_IN, k = idx % GATE_IN;
        atomicAdd(&p.dw1[idx], dhp[j] * G[k]);
    }
    for (int i = t; i < GATE_F; i += 256) {
        float dG = 0.f;
        for (int j = 0; j < 64; ++j) dG = fmaf(dhp[j], p.w1[j * GATE_IN + i], dG);
        const int sg = gate_seg(i);
        p.dfeats[(size_t)m * GATE_F + i] = (p.dfused[(size_t)m * GATE_F + i] * g[sg] + dG) * p.mask[m * 3 + sg];
    }
}
// gate_entropy_loss(gate) = mean_b sum_k g log(g + 1e-8)  (partial_modality_training.py:322-331), value and gradient
__global__ void gate_entropy_kernel(const float* gate, int M, float scale, float* loss, float* dgate) {
    const int m = blockIdx.x * 64 + threadIdx.x;
    float e = 0.f;
    if (m < M) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g = gate[m * 3 + c];
            e += g * logf(g + 1e-8f);
            if (dgate) dgate[m * 3 + c] = scale / (float)M * (logf(g + 1e-8f) + g / (g + 1e-8f));
        }
    }
    e = wave_sum(e);
    if (threadIdx.x == 0 && loss) atomicAdd(loss, e / (float)M);
}
extern "C" int mms_gate_entropy(const float* gate, int M, float scale, float* loss, float* dgate, hipStream_t s) {
    if (M <= 0) return MMS_ERR_ARG;
    MMS_LAUNCH(gate_entropy_kernel, dim3((M + 63) / 64), dim3(64), 0, s, gate, M, scale, loss, dgate);
    return mms_check_launch();
}

static bool gate_group(Grp<GateP>& a, const GateP* pp, int ng) {
    if (!grp_fill(a, pp, ng, 1) || pp->M <= 0) return false;
    for (int g = 1; g < ng; ++g) if (pp[g].M != pp->M) return false;
    return true;
}
extern "C" int mms_gate_fwd_group(const GateP* pp, int ng, hipStream_t s) {
    Grp<GateP> a;
    if (!gate_group(a, pp, ng)) return MMS_ERR_ARG;
    MMS_LAUNCH(gate_fwd_kernel, dim3(pp->M, 1, ng), dim3(256), 0, s, a);
    return mms_check_launch();
}
extern "C" int mms_gate_bwd_group(const GateP* pp, int ng, hipStream_t s) {
    Grp<GateP> a;
    if (!gate_group(a, pp, ng)) return MMS_ERR_ARG;
    MMS_LAUNCH(gate_bwd_kernel, dim3(pp->M, 1, ng), dim3(256), 0, s, a);
    return mms_check_launch();
}
MMS_SINGLE(mms_gate_fwd, GateP)
MMS_SINGLE(mms_gate_bwd, GateP)

// ------------------------------------------------------------------------------------------------------
// Cox partial likelihood, O(n^2) risk-set form.  One wave per row i, lanes over j (wavefront-shuffle reductions).
//   pass 1: lse_i = log sum_{j valid, t_j >= t_i} exp(h_j)          pass 2: loss, dh
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool cox_valid(const CoxP& p, int i) { return p.valid == nullptr || p.valid[i] != 0.f; }

__global__ __launch_bounds__(256) void cox_lse_kernel(const Grp<CoxP> grp) {
    const CoxP& p = grp.p[blockIdx.z];
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= p.n) return;
    if (!cox_valid(p, i)) { if (lane == 0) { p.lse[i] = 0.f; if (p.tie_mode == 1) p.tie_frac[i] = -1.f; } return; }
    const float ti = p.time[i];
    float mx = -INFINITY;
    for (int j = lane; j < p.n; j += 64)
        if (cox_valid(p, j) && p.time[j] >= ti) mx = fmaxf(mx, p.h[(size_t)j * p.ldh]);
    mx = wave_max(mx);
    float s = 0.f;
    if (p.tie_mode != 1) {
        for (int j = lane; j < p.n; j += 64)
            if (cox_valid(p, j) && p.time[j] >= ti) s += expf(p.h[(size_t)j * p.ldh] - mx);
        s = wave_sum(s);
        if (lane == 0) p.lse[i] = mx + logf(s);
        return;
    }
    // Efron (torchsurv _partial_likelihood_efron): the m events tied at t_i take the denominators D - (l/m) T, l = 0..m-1,
    // D = sum over the risk set, T = sum over the tied events; event i takes l = its rank (by index) among them.
    float st = 0.f, m = 0.f, l = 0.f;
    for (int j = lane; j < p.n; j += 64)
        if (cox_valid(p, j)) {
            const float tj = p.time[j];
            if (tj >= ti) {
                const float e = expf(p.h[(size_t)j * p.ldh] - mx);
                s += e;
                if (tj == ti && p.event[j] != 0.f) { st += e; m += 1.f; l += j < i ? 1.f : 0.f; }
            }
        }
    s = wave_sum(s); st = wave_sum(st); m = wave_sum(m); l = wave_sum(l);
    if (lane == 0) {
        const bool ev = p.event[i] != 0.f;
        const float f = ev ? l / m : 0.f;            // (censored rows: lse unused)
        p.lse[i] = mx + logf(s - f * st);
        p.tie_frac[i] = ev ? f : -1.f;
    }
}

__global__ __launch_bounds__(256) void cox_grad_kernel(const Grp<CoxP> grp) {
    const CoxP& p = grp.p[blockIdx.z];
    __shared__ float red[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = blockIdx.x * 4 + wave;
    const bool efron = p.tie_mode == 1;
    // batch counts (every block recomputes them: n is small).  ne = events (Breslow / untied: mean over events);
    // Efron: nd = distinct event times = events of tie rank 0 (torchsurv averages its per-time terms)
    float nv = 0.f, ne = 0.f, ls = 0.f, nd = 0.f;
    for (int j = threadIdx.x; j < p.n; j += 256)
        if (cox_valid(p, j)) {
            nv += 1.f;
            if (p.event[j] != 0.f) {
                ne += 1.f; ls += p.h[(size_t)j * p.ldh] - p.lse[j];
                if (efron && p.tie_frac[j] == 0.f) nd += 1.f;
            }
        }
    nv = wave_sum(nv); ne = wave_sum(ne); ls = wave_sum(ls); nd = wave_sum(nd);
    if (lane == 0) { red[wave] = nv; red[4 + wave] = ne; }
    __syncthreads();
    nv = red[0] + red[1] + red[2] + red[3]; ne = red[4] + red[5] + red[6] + red[7];
    __syncthreads();
    if (lane == 0) { red[wave] = ls; red[4 + wave] = nd; }
    __syncthreads();
    ls = red[0] + red[1] + red[2] + red[3]; nd = red[4] + red[5] + red[6] + red[7];
    const bool usable = nv >= 2.f && ne > 0.f;      // final_multimodal.py:173-176
    const float denom = efron ? nd : ne + 1e-8f;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        p.out[0] = usable ? -ls / denom : 0.f;
        p.out[1] = usable ? 1.f : 0.f;
    }
    if (k >= p.n || p.dh == nullptr) return;
    float gsum = 0.f;
    if (usable && cox_valid(p, k)) {
        const float tk = p.time[k], hk = p.h[(size_t)k * p.ldh];
        const bool ek = p.event[k] != 0.f;
        for (int i = lane; i < p.n; i += 64)
            if (cox_valid(p, i) && p.event[i] != 0.f && p.time[i] <= tk) {
                float w = 1.f;
                if (efron && ek && p.time[i] == tk) w -= p.tie_frac[i];      // d/dh_k of D_i - f_i T_i
                gsum += w * expf(hk - p.lse[i]);
            }
        gsum = wave_sum(gsum);
        if (lane == 0) p.dh[(size_t)k * p.lddh] = -p.scale * ((ek ? 1.f : 0.f) - gsum) / denom;
    } else if (lane == 0) {
        p.dh[(size_t)k * p.lddh] = 0.f;
    }
}
extern "C" int mms_cox_fwd_bwd_group(const CoxP* pp, int ng, hipStream_t s) {
    Grp<CoxP> a;
    if (!grp_fill(a, pp, ng, 1)) return MMS_ERR_ARG;
    const CoxP& p = *pp;
    for (int g = 0; g < ng; ++g)
        if (pp[g].n != p.n || pp[g].n <= 0 || !pp[g].lse || !pp[g].out || pp[g].tie_mode < 0 || pp[g].tie_mode > 1 || (pp[g].tie_mode == 1 && !pp[g].tie_frac)) return MMS_ERR_ARG;
    MMS_LAUNCH(cox_lse_kernel, dim3((p.n + 3) / 4, 1, ng), dim3(256), 0, s, a);
    MMS_LAUNCH(cox_grad_kernel, dim3((p.n + 3) / 4, 1, ng), dim3(256), 0, s, a);
    return mms_check_launch();
}
MMS_SINGLE(mms_cox_fwd_bwd, CoxP)

// ------------------------------------------------------------------------------------------------------
// Harrell's C pair counts: one wave per event row i
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cindex_kernel(const CindexP p) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= p.n || p.event[i] != 1.f) return;
    const float ti = p.time[i], hi = p.h[i];
    unsigned conc = 0, tied = 0, perm = 0;
    for (int j = lane; j < p.n; j += 64)
        if (p.time[j] > ti) {
            ++perm;
            const float hj = p.h[j];
            conc += hi > hj; tied += hi == hj;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { conc += __shfl_xor(conc, o, 64); tied += __shfl_xor(tied, o, 64); perm += __shfl_xor(perm, o, 64); }
    if (lane == 0) {
        atomicAdd(&p.counts[0], (unsigned long long)conc);
        atomicAdd(&p.counts[1], (unsigned long long)tied);
        atomicAdd(&p.counts[2], (unsigned long long)perm);
    }
}
extern "C" int mms_cindex_counts(const CindexP* pp, hipStream_t s) {
    if (pp->n <= 0) return MMS_ERR_ARG;
    MMS_LAUNCH(cindex_kernel, dim3((pp->n + 3) / 4), dim3(256), 0, s, *pp);
    return mms_check_launch();
}

// ------------------------------------------------------------------------------------------------------
// clip_grad_norm_ + Adam / AdamW over a flat buffer
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grad_sumsq_kernel(const Grp<AdamP> grp) {
    const AdamP& p = grp.p[blockIdx.z];
    __shared__ double red[4];
    const long long n4 = p.n >> 2, stride = (long long)gridDim.x * 256;
    float a = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {          // 4 independent 16-B loads in flight per thread
        const float4 g0 = ((const float4*)p.g)[i], g1 = ((const float4*)p.g)[i + stride], g2 = ((const float4*)p.g)[i + 2 * stride],
                     g3 = ((const float4*)p.g)[i + 3 * stride];
        a = fmaf(g0.x, g0.x, a); a = fmaf(g0.y, g0.y, a); a = fmaf(g0.z, g0.z, a); a = fmaf(g0.w, g0.w, a);
        a1 = fmaf(g1.x, g1.x, a1); a1 = fmaf(g1.y, g1.y, a1); a1 = fmaf(g1.z, g1.z, a1); a1 = fmaf(g1.w, g1.w, a1);
        a2 = fmaf(g2.x, g2.x, a2); a2 = fmaf(g2.y, g2.y, a2); a2 = fmaf(g2.z, g2.z, a2); a2 = fmaf(g2.w, g2.w, a2);
        a3 = fmaf(g3.x, g3.x, a3); a3 = fmaf(g3.y, g3.y, a3); a3 = fmaf(g3.z, g3.z, a3); a3 = fmaf(g3.w, g3.w, a3);
    }
    for (; i < n4; i += stride) {
        const float4 g = ((const float4*)p.g)[i];
        a = fmaf(g.x, g.x, a); a = fmaf(g.y, g.y, a); a = fmaf(g.z, g.z, a); a = fmaf(g.w, g.w, a);
    }
    a += a1 + a2 + a3;
    if (blockIdx.x == 0 && threadIdx.x < (p.n & 3)) { const float g = p.g[(n4 << 2) + threadIdx.x]; a = fmaf(g, g, a); }
    const double t = block_sum_d((double)a, red);
    if (threadIdx.x == 0) {
        atomicAdd(p.sumsq, t);
        if (blockIdx.x == 0) {
            if (p.skip_flag == nullptr || *p.skip_flag != 0.f) p.step[0] += 1.f;
            // per-step bookkeeping of the training loop (R/final_multimodal.py:262-264: total_loss += loss.item(), n_batches += 1),
            // kept on the device: [sum loss*usable, n usable, sum gate entropy, n batches]; dropout stream counter
            if (p.acc) {
                if (p.cox_out) { p.acc[0] += p.cox_out[0] * p.cox_out[1]; p.acc[1] += p.cox_out[1]; }
                if (p.entropy) p.acc[2] += p.entropy[0];
                p.acc[3] += 1.f;
            }
            if (p.rng) p.rng[1] += 1;
        }
    }
}
constexpr int W2N = 32 * 27 * 128;       // floats per conv2 tensor
__global__ __launch_bounds__(256) void clip_adam_kernel(const Grp<AdamP> grp) {
    const AdamP& p = grp.p[blockIdx.z];
    __shared__ float c[6];
    if (p.skip_flag != nullptr && *p.skip_flag == 0.f) return;
    if (threadIdx.x == 0) {
        const double lr = p.hyper[0], b1 = p.hyper[1], b2 = p.hyper[2], wd = p.hyper[4], maxn = p.hyper[5];
        const double t = p.step[0];
        const double norm = sqrt(*p.sumsq);
        double coef = maxn / (norm + 1e-6);
        if (coef > 1.0) coef = 1.0;
        if (maxn <= 0.0) coef = 1.0;
        const double bc1 = 1.0 - pow(b1, t), bc2 = 1.0 - pow(b2, t);
        c[0] = (float)coef; c[1] = (float)(lr / bc1); c[2] = (float)(1.0 / sqrt(bc2));
        c[3] = (float)(1.0 - lr * wd); c[4] = (float)wd;
    }
    __syncthreads();
    const float coef = c[0], step_size = c[1], inv_sqrt_bc2 = c[2], decay = c[3], wd = c[4];
    const float b1 = p.hyper[1], b2 = p.hyper[2], eps = p.hyper[3];
    const long long stride = (long long)gridDim.x * 256;
    // The flat buffer minus the packed-primary conv2 tensors (w2_adam_pack_kernel updates those).  The kernel walks a VIRTUAL index over
    // the remaining elements; tensor k sits at virtual position vp[k] = w2_off[k] - k * W2N, so element j of the walk is the real element
    // j + W2N * #{k : vp[k] <= j} (binary search in LDS -- a loop over the 59 gaps with their offsets read from memory cost 59 serial
    // round trips per workgroup: 108 us instead of 40).
    __shared__ long long vp[64];
    const int nw = p.n_w2;
    if ((int)threadIdx.x < nw) vp[threadIdx.x] = p.w2_off[threadIdx.x] - (long long)threadIdx.x * W2N;
    if (nw) __syncthreads();
    const long long nv = p.n - (long long)nw * W2N;
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < nv; j += stride) {
        int lo = 0, hi = nw;               // #{k : vp[k] <= j}
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (vp[mid] <= j) lo = mid + 1; else hi = mid; }
        const long long i = j + (long long)lo * W2N;
        float w = p.p[i], g = p.g[i] * coef;
        if (p.adamw) w *= decay; else g = fmaf(wd, w, g);
        const float m = fmaf(b1, p.m[i], (1.f - b1) * g);
        const float v = fmaf(b2, p.v[i], (1.f - b2) * g * g);
        p.m[i] = m; p.v[i] = v;
        p.p[i] = w - step_size * m / (sqrtf(v) * inv_sqrt_bc2 + eps);
    }
}

// ---- conv2 (3x3x3, 128 -> 32) weights of the dense layers in PACKED PRIMARY storage (MmsDnOpts.w2_packed, round 4) -------------------
// MONAI's _DenseLayer.layers.conv2.weight is a torch tensor [32 co][128 cin][3][3][3].  The kernels never read that layout: the forward
// wants [co][tap][cin] (or an MFMA-fragment order), the backward-data [cin][tap][co] (or its fragment order), and the weight-gradient
// kernels flush 512-byte runs per (co, tap).  Rounds 1-3 kept the torch layout as primary storage and paid, every step, a pack launch
// (canonical -> two packs, 51 MB per model) and an unpack launch (tap-major gradient scratch -> canonical gradient).  Now the host stores
// weight, gradient and both Adam moments of such a tensor as [co][tap][cin] inside its flat buffers (the nn.Parameter is a strided VIEW of
// that storage, so state_dict() / load_state_dict() / .grad keep their torch shapes), and the optimiser step -- which touches every
// weight anyway -- emits the derived packs:
//   * forward, layers whose launches take the classic pack: the primary storage IS that pack -- nothing to do;
//   * backward-data pack [cin][tap][co], or both MFMA-fragment orders for the layers of w2_fragmask: written here through an LDS transpose.
// One workgroup = (layer, tap): the tap's 32 x 128 tile of w, g, m, v is one 512-byte run per output channel.
constexpr int W2P = 132;                 // LDS pitch of the 32 x 128 tile (floats): float4-aligned rows, 4 banks of skew per row

template <bool UPDATE>
__global__ __launch_bounds__(256) void w2_adam_pack_kernel(const Grp<AdamP> grp) {
    const AdamP& p = grp.p[blockIdx.z];
    const int tap = blockIdx.x, layer = blockIdx.y, tid = threadIdx.x;
    __shared__ __attribute__((aligned(16))) float t[32 * W2P];
    __shared__ float c[6];
    if (UPDATE) {
        if (p.skip_flag != nullptr && *p.skip_flag == 0.f) return;      // unusable batch: no update, the packs stay valid
        if (tid == 0) {      // (the constants of clip_adam_kernel, same arithmetic)
            const double lr = p.hyper[0], b1 = p.hyper[1], b2 = p.hyper[2], wd = p.hyper[4], maxn = p.hyper[5];
            const double st = p.step[0];
            const double norm = sqrt(*p.sumsq);
            double coef = maxn / (norm + 1e-6);
            if (coef > 1.0) coef = 1.0;
            if (maxn <= 0.0) coef = 1.0;
            const double bc1 = 1.0 - pow(b1, st), bc2 = 1.0 - pow(b2, st);
            c[0] = (float)coef; c[1] = (float)(lr / bc1); c[2] = (float)(1.0 / sqrt(bc2));
            c[3] = (float)(1.0 - lr * wd); c[4] = (float)wd;
        }
        __syncthreads();
    }
    const size_t base = (size_t)p.w2_off[layer];
    // thread -> float4 j of the tile: co = idx4 >> 5, cin = 4 * (idx4 & 31); 32 consecutive threads cover one 512-byte run
    float4 w4[4];
    size_t off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx4 = tid + 256 * j, co = idx4 >> 5, c4 = idx4 & 31;
        off[j] = base + ((size_t)co * 27 + tap) * 128 + 4 * c4;
        w4[j] = *(const float4*)(p.p + off[j]);
    }
    if (UPDATE) {
        float4 g4[4], m4[4], v4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { g4[j] = *(const float4*)(p.g + off[j]); m4[j] = *(const float4*)(p.m + off[j]); v4[j] = *(const float4*)(p.v + off[j]); }
        const float coef = c[0], step_size = c[1], inv_sqrt_bc2 = c[2], decay = c[3], wd = c[4];
        const float b1 = p.hyper[1], b2 = p.hyper[2], eps = p.hyper[3];
        auto upd = [&](float& w, float g, float& m, float& v) {
            g *= coef;
            if (p.adamw) w *= decay; else g = fmaf(wd, w, g);
            m = fmaf(b1, m, (1.f - b1) * g);
            v = fmaf(b2, v, (1.f - b2) * g * g);
            w = w - step_size * m / (sqrtf(v) * inv_sqrt_bc2 + eps);
        };
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            upd(w4[j].x, g4[j].x, m4[j].x, v4[j].x); upd(w4[j].y, g4[j].y, m4[j].y, v4[j].y);
            upd(w4[j].z, g4[j].z, m4[j].z, v4[j].z); upd(w4[j].w, g4[j].w, m4[j].w, v4[j].w);
            *(float4*)(p.p + off[j]) = w4[j]; *(float4*)(p.m + off[j]) = m4[j]; *(float4*)(p.v + off[j]) = v4[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx4 = tid + 256 * j, co = idx4 >> 5, c4 = idx4 & 31;
        *(float4*)(t + co * W2P + 4 * c4) = w4[j];
    }
    __syncthreads();
    const bool frag = (p.w2_fragmask >> layer) & 1ull;
    float4* pb = (float4*)p.w2_pack_b[layer];
    if (frag) {
        // backward fragment order [tap][cin/16][co/16][lane = ((co/4)%4)*16 + cin%16][co%4]: float4 index q * 64 + lane, q = (cin/16)*2 + co/16
        float4* dst = pb + (size_t)tap * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx4 = tid + 256 * j, q = idx4 >> 6, lane = idx4 & 63;
            const int cin = ((q >> 1) << 4) | (lane & 15), co0 = ((q & 1) << 4) | ((lane >> 4) << 2);
            dst[idx4] = make_float4(t[co0 * W2P + cin], t[(co0 + 1) * W2P + cin], t[(co0 + 2) * W2P + cin], t[(co0 + 3) * W2P + cin]);
        }
        // forward fragment order [tap][cin/32][(cin/16)%2][co/16][lane = ((cin/4)%4)*16 + co%16][cin%4]: float4 index q * 64 + lane,
        // q = (cin/16)*2 + co/16 (cin/16 = 2*(cin/32) + (cin/16)%2)
        float4* df = (float4*)p.w2_pack_f[layer] + (size_t)tap * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx4 = tid + 256 * j, q = idx4 >> 6, lane = idx4 & 63;
            const int c4 = ((q >> 1) << 2) | (lane >> 4), co = ((q & 1) << 4) | (lane & 15);
            df[idx4] = *(const float4*)(t + co * W2P + 4 * c4);
        }
    } else {
        // classic backward-data pack [cin][tap][co]: one 128-byte run of 32 output channels per input channel
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx4 = tid + 256 * j, cin = idx4 >> 3, co0 = (idx4 & 7) << 2;
            pb[((size_t)cin * 27 + tap) * 8 + (idx4 & 7)] =
                make_float4(t[co0 * W2P + cin], t[(co0 + 1) * W2P + cin], t[(co0 + 2) * W2P + cin], t[(co0 + 3) * W2P + cin]);
        }
    }
}



static bool adam_group(Grp<AdamP>& a, const AdamP* pp, int ng) {
    if (!grp_fill(a, pp, ng, 1) || pp->n <= 0) return false;
    for (int g = 0; g < ng; ++g) {
        const AdamP& p = pp[g];
        if (p.n != pp->n || p.n_w2 != pp->n_w2 || p.w2_fragmask != pp->w2_fragmask || p.n_w2 < 0 || p.n_w2 > 64) return false;
        if (p.n_w2 && (!p.w2_off || !p.w2_pack_b || (p.w2_fragmask && !p.w2_pack_f) || (((uintptr_t)p.p | (uintptr_t)p.g | (uintptr_t)p.m | (uintptr_t)p.v) & 15))) return false;
    }
    return true;
}
extern "C" int mms_grad_sumsq_group(const AdamP* pp, int ng, hipStream_t s) {
    Grp<AdamP> a;
    if (!adam_group(a, pp, ng)) return MMS_ERR_ARG;
    long long blocks = (pp->n / 16 + 255) / 256;          // 4 float4 per thread and trip
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    MMS_LAUNCH(grad_sumsq_kernel, dim3((unsigned)blocks, 1, ng), dim3(256), 0, s, a);
    return mms_check_launch();
}
extern "C" int mms_clip_adam_group(const AdamP* pp, int ng, hipStream_t s) {
    Grp<AdamP> a;
    if (!adam_group(a, pp, ng)) return MMS_ERR_ARG;
    long long blocks = (pp->n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    MMS_LAUNCH(clip_adam_kernel, dim3((unsigned)blocks, 1, ng), dim3(256), 0, s, a);
    if (pp->n_w2 > 0) MMS_LAUNCH(w2_adam_pack_kernel<true>, dim3(27, pp->n_w2, ng), dim3(256), 0, s, a);
    return mms_check_launch();
}
extern "C" int mms_w2_pack_group(const AdamP* pp, int ng, hipStream_t s) {
    Grp<AdamP> a;
    if (!adam_group(a, pp, ng)) return MMS_ERR_ARG;
    if (pp->n_w2 > 0) MMS_LAUNCH(w2_adam_pack_kernel<false>, dim3(27, pp->n_w2, ng), dim3(256), 0, s, a);
    return mms_check_launch();
}
MMS_SINGLE(mms_grad_sumsq, AdamP)
MMS_SINGLE(mms_clip_adam, AdamP)
MMS_SINGLE(mms_w2_pack, AdamP)

// ------------------------------------------------------------------------------------------------------
// batch assembly: gather cohort rows into the step's static input buffers (all sources, all models, one launch)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const Grp<GatherP> grp) {
    const GatherP& p = grp.p[blockIdx.z];
    const int b = blockIdx.y;
    if (b >= p.B) return;
    const long long row = p.idx[b];
    const int t0 = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    for (int s = 0; s < p.nsrc; ++s) {
        const float* src = p.src[s] + (size_t)row * p.src_ld[s];
        float* dst = p.dst[s] + (size_t)b * p.dst_ld[s];
        const int w = p.width[s];
        const bool absent = p.present[s] && p.present[s][(size_t)row * p.present_ld[s]] == 0.f;     // all-zero row by contract: not read
        if ((w & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
            if (absent) { for (int i = t0; i < (w >> 2); i += stride) ((float4*)dst)[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
            else {
                int i = t0;
                for (; i + 3 * stride < (w >> 2); i += 4 * stride) {        // four 16-B loads in flight per thread (host-resident rows: PCIe latency)
                    const float4 a = ((const float4*)src)[i], b = ((const float4*)src)[i + stride], c = ((const float4*)src)[i + 2 * stride],
                                 d = ((const float4*)src)[i + 3 * stride];
                    ((float4*)dst)[i] = a; ((float4*)dst)[i + stride] = b; ((float4*)dst)[i + 2 * stride] = c; ((float4*)dst)[i + 3 * stride] = d;
                }
                for (; i < (w >> 2); i += stride) ((float4*)dst)[i] = ((const float4*)src)[i];
            }
        } else {
            for (int i = t0; i < w; i += stride) dst[i] = absent ? 0.f : src[i];
        }
    }
}
extern "C" int mms_gather_rows_group(const GatherP* pp, int ng, hipStream_t s) {
    Grp<GatherP> a;
    if (!grp_fill(a, pp, ng, 1)) return MMS_ERR_ARG;
    int wmax = 0;
    for (int g = 0; g < ng; ++g) {
        const GatherP& q = pp[g];
        if (q.B != pp->B || q.B <= 0 || !q.idx || q.nsrc < 1 || q.nsrc > 8) return MMS_ERR_ARG;
        for (int i = 0; i < q.nsrc; ++i) {
            if (!q.src[i] || !q.dst[i] || q.width[i] <= 0) return MMS_ERR_ARG;
            if (q.width[i] > wmax) wmax = q.width[i];
        }
    }
    int blocks = (wmax / 4 + 1023) / 1024;       // ~4 float4 per thread on the widest source
    if (blocks < 1) blocks = 1;
    if (blocks > 64) blocks = 64;
    MMS_LAUNCH(gather_rows_kernel, dim3(blocks, pp->B, ng), dim3(256), 0, s, a);
    return mms_check_launch();
}

// ------------------------------------------------------------------------------------------------------
// learnable missing-modality bias (flexible_multimodal.py:243-250): one thread per feature column
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void missing_mix_kernel(const Grp<MixP> grp, int bwd) {
    const MixP& p = grp.p[blockIdx.z];
    const int j = blockIdx.x * 256 + threadIdx.x;
    int s = -1;
    for (int t = 0; t < p.nseg; ++t) if (j >= p.seg_begin[t] && j < p.seg_begin[t] + p.seg_width[t]) s = t;
    if (s < 0) return;
    const int jl = j - p.seg_begin[s];
    if (!bwd) {
        const float b = p.bias[s][jl];
        for (int m = 0; m < p.M; ++m) {
            const float mk = p.mask[(size_t)m * p.ldm + s];
            float* f = &p.feats[(size_t)m * p.ld + j];
            *f = *f * mk + b * (1.f - mk);
        }
    } else {
        float acc = 0.f;
        for (int m = 0; m < p.M; ++m) {
            const float mk = p.mask[(size_t)m * p.ldm + s];
            float* d = &p.dfeats[(size_t)m * p.ldd + j];
            acc = fmaf(*d, 1.f - mk, acc);
            *d = *d * mk;
        }
        if (p.dbias[s]) p.dbias[s][jl] += acc;
    }
}
static int mix_launch(const MixP* pp, int ng, int bwd, hipStream_t s) {
    Grp<MixP> a;
    if (!grp_fill(a, pp, ng, 1)) return MMS_ERR_ARG;
    int wmax = 0;
    for (int g = 0; g < ng; ++g) {
        const MixP& q = pp[g];
        if (q.M != pp->M || q.M <= 0 || q.nseg != pp->nseg || q.nseg < 1 || q.nseg > 4 || !q.mask) return MMS_ERR_ARG;
        if (bwd ? !q.dfeats : !q.feats) return MMS_ERR_ARG;
        for (int t = 0; t < q.nseg; ++t) {
            if (q.seg_begin[t] != pp->seg_begin[t] || q.seg_width[t] != pp->seg_width[t] || !q.bias[t]) return MMS_ERR_ARG;
            if (q.seg_begin[t] + q.seg_width[t] > wmax) wmax = q.seg_begin[t] + q.seg_width[t];
        }
    }
    MMS_LAUNCH(missing_mix_kernel, dim3((wmax + 255) / 256, 1, ng), dim3(256), 0, s, a, bwd);
    return mms_check_launch();
}
extern "C" int mms_missing_mix_fwd_group(const MixP* pp, int ng, hipStream_t s) { return mix_launch(pp, ng, 0, s); }
extern "C" int mms_missing_mix_bwd_group(const MixP* pp, int ng, hipStream_t s) { return mix_launch(pp, ng, 1, s); }
MMS_SINGLE(mms_missing_mix_fwd, MixP)
MMS_SINGLE(mms_missing_mix_bwd, MixP)
