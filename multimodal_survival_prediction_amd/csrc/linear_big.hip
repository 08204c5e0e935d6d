// Large-batch Linear layers (rows M > 32) on the fp32-MFMA tile core: BASELINE config 5 (RNA-seq-only model at batch 2048,
// R/scripts/training/train_rnaseq_only.py:126-176).  The small-batch kernels of heads.hip keep all rows of a column in one
// lane (M <= 32); here the layers are ordinary GEMMs -- [M x K] x [K x N] with K up to 5005 -- and the neighbouring
// BatchNorm1d / ReLU / Dropout are fused the way the DenseNet ops fuse BatchNorm3d: batch statistics of a layer's output are
// accumulated by its epilogue (fp64 sum / sumsq), the next layer normalises in its operand prologue.
#include "dn_ops.h"
#include "tile_gemm.h"

namespace {

// prologue constants of the K input columns in LDS (K <= 1024 whenever a BatchNorm1d precedes: 1024 / 512 / 256 / 128)
struct Prolog {
    const float *mean, *sc, *beta;
    bool has_bn, drop;
    uint32_t seed, stream;
    float dp;
    const float* mask; int K;
    __device__ void init(const LinBigP& p, float* lds, int tid) {
        mean = lds; sc = lds + 1024; beta = lds + 2048;
        has_bn = p.has_bn != 0;
        if (has_bn) bn_consts_to_lds<4>(p.bn, p.K, tid, lds, lds + 1024, lds + 2048);
        drop = p.train && (p.drop_mask != nullptr || p.drop_p > 0.f);
        mask = p.drop_mask; dp = p.drop_p; K = p.K; stream = p.stream_id;
        seed = (drop && !mask) ? p.rng[0] + 0x9E3779B9u * p.rng[1] : 0u;
    }
    __device__ __forceinline__ float scale(int m, int k) const {
        if (!drop) return 1.f;
        return mask ? mask[(size_t)m * K + k] : dropout_scale(seed, stream, (uint32_t)(m * K + k), dp);
    }
    // value of P(x)[m][k]
    __device__ __forceinline__ float act(float v, int m, int k) const {
        if (has_bn) v = fmaxf(bn_apply(v, mean[k], sc[k], beta[k]), 0.f);
        return v * scale(m, k);
    }
};

// 16-byte load from a 4-byte aligned address (rows of a [N][5005] weight): one global_load_dwordx4, the hardware splits it
typedef float f4u_t __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ float4 ld4u(const float* p) {
    const f4u_t v = *(const f4u_t*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
// elements k..k+3 of a K-long row (zero beyond the row end; the last partial quad is read element-wise: no over-read)
__device__ __forceinline__ float4 ld4_row(const float* row, int k, int K) {
    if (k + 3 < K) return ld4u(row + k);
    return make_float4(row[k], k + 1 < K ? row[k + 1] : 0.f, k + 2 < K ? row[k + 2] : 0.f, 0.f);
}

// zero the lanes of a float4 that lie beyond column K (pad columns of a padded row carry no data)
__device__ __forceinline__ float4 tail4(float4 v, int k, int K) {
    if (k + 3 < K) return v;
    return make_float4(v.x, k + 1 < K ? v.y : 0.f, k + 2 < K ? v.z : 0.f, 0.f);
}

// ---------------------------------------------------------------------------------------------------------------------
// forward:  y[m][n] = sum_k P(x)[m][k] * W[n][k] + bias[n]
// ---------------------------------------------------------------------------------------------------------------------
// XA: rows of x are 16-B aligned and padded to a multiple of 4 floats (ldx >= roundup4(K), pad columns readable) -> float4
// loads; WA: 16-byte loads of the weight rows from 4-byte aligned addresses (the [N][5005] weight of the first layer has
// unaligned rows; the hardware splits such a load, still one instruction per 4 elements instead of four).
template <bool XA, bool WA>
struct LinBigFwdOp {
    typedef LinBigP Params;
    static constexpr int WM = 2, WN = 2, WK = 1, AMODE = XA ? LD_K4 : LD_K1, BMODE = WA ? LD_K4 : LD_K1;
    static constexpr int TM = 64, TN = 64;
    static constexpr int EXTRA = 3 * 1024;
    typedef typename std::conditional<XA, float4, float>::type ARaw;
    typedef typename std::conditional<WA, float4, float>::type BRaw;
    Prolog pr;
    __device__ void step(const Params&, int) {}
    __device__ void setup(const Params& p, int m0, int n0, int, float* extra, int tid) {
        pr.init(p, extra, tid);
        // running statistics of the BatchNorm1d in front of this layer (torch: momentum update with the unbiased variance)
        if (p.has_bn && p.train && p.rmean && m0 == 0 && n0 == 0) {
            for (int k = tid; k < p.K; k += 256) {
                const double cnt = 1.0 / (double)p.bn.inv_count;
                const double m = rep_sum(p.bn.sum, k, p.bn.nrep, p.bn.rep_stride) / cnt;
                double v = rep_sum(p.bn.sumsq, k, p.bn.nrep, p.bn.rep_stride) / cnt - m * m;
                if (v < 0) v = 0;
                const double unb = cnt > 1.0 ? v * cnt / (cnt - 1.0) : v;
                p.rmean[k] = (1.f - p.momentum) * p.rmean[k] + p.momentum * (float)m;
                p.rvar[k] = (1.f - p.momentum) * p.rvar[k] + p.momentum * (float)unb;
            }
            if (tid == 0 && p.nbt) *p.nbt += 1;
        }
    }
    __device__ void krange(const Params& p, int, int& kb, int& ke) { kb = 0; ke = p.K; }
    __device__ ARaw a_ld(const Params& p, int, int m, int k, bool& ok) const {
        ok = m < p.M && k < p.K;
        if constexpr (XA) return ok ? *(const float4*)(p.x + (size_t)m * p.ldx + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        else return ok ? p.x[(size_t)m * p.ldx + k] : 0.f;
    }
    __device__ ARaw a_tx(const Params& p, int, const ARaw& v, int m, int k, bool ok) const {
        if constexpr (XA) {
            if (!ok) return v;
            return tail4(make_float4(pr.act(v.x, m, k), pr.act(v.y, m, k + 1), pr.act(v.z, m, k + 2), pr.act(v.w, m, k + 3)), k, p.K);
        } else {
            return ok ? pr.act(v, m, k) : 0.f;
        }
    }
    __device__ BRaw b_ld(const Params& p, int, int n, int k, bool& ok) const {
        ok = n < p.N && k < p.K;
        if constexpr (WA) return ok ? ld4_row(p.w + (size_t)n * p.K, k, p.K) : make_float4(0.f, 0.f, 0.f, 0.f);
        else return ok ? p.w[(size_t)n * p.K + k] : 0.f;
    }
    __device__ BRaw b_tx(const Params&, int, const BRaw& v, int, int, bool) const { return v; }
    __device__ void epilogue(const Params& p, int m0, int n0, int, const float* Cs, int tid, bool) {
        float* C = const_cast<float*>(Cs);
        for (int idx = tid; idx < TM * TN; idx += 256) {
            const int r = idx / TN, c = idx % TN, m = m0 + r, n = n0 + c;
            if (m < p.M && n < p.N) {
                float v = C[r * (TN + 1) + c] + (p.bias ? p.bias[n] : 0.f);
                if (p.out_relu) v = fmaxf(v, 0.f);
                C[r * (TN + 1) + c] = v;
                p.y[(size_t)m * p.ldy + n] = v;
            }
        }
        __syncthreads();
        if (p.osum && tid < TN && n0 + tid < p.N) {
            double s = 0, q = 0;
            const int rows = p.M - m0 < TM ? p.M - m0 : TM;
            for (int r = 0; r < rows; ++r) { const double v = C[r * (TN + 1) + tid]; s += v; q += v * v; }
            atomicAdd(&p.osum[n0 + tid], s);
            atomicAdd(&p.osumsq[n0 + tid], q);
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// forward of a layer WITHOUT input prologue (the first layer), large shapes: 128 x 128 output tiles, split-K.  A wave owns
// all 128 rows of the tile x 32 columns (4 accumulators); per 16 MFMAs it reads 4 x-fragments + 1 weight fragment
// (ds_read_b128) and the workgroup loads half the operand bytes per MFMA of the 64 x 64 form.  2048 x 1024 outputs are only
// 128 such tiles, so K is split over grid.z and the partial products are added into a zeroed y with fp32 atomics (128-byte
// runs; split 0 adds the bias); output ReLU and the BatchNorm column statistics of y follow in lin_colstats_kernel.
// ---------------------------------------------------------------------------------------------------------------------
#define LFW_P 36
#define LFW_STAGE (2 * 128 * LFW_P)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void lin_fwd_wide_kernel(const LinBigP p, const int kc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
    const int kb = blockIdx.z * kc, ke = kb + kc < p.K ? kb + kc : p.K;
    if (kb >= ke) return;
    const int k4 = (tid & 7) * 4, r32 = tid >> 3;
    float4 ra[4], rb[4];
    auto gload = [&](int k0) __attribute__((always_inline)) {
        const int k = k0 + k4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + r32 + 32 * i, n = n0 + r32 + 32 * i;
            ra[i] = (m < p.M && k + 3 < p.ldx) ? tail4(*(const float4*)(p.x + (size_t)m * p.ldx + k), k, p.K) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[i] = (n < p.N && k < p.K) ? ld4_row(p.w + (size_t)n * p.K, k, p.K) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto sstore = [&](float* st) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(float4*)&st[(r32 + 32 * i) * LFW_P + k4] = ra[i];
            *(float4*)&st[128 * LFW_P + (r32 + 32 * i) * LFW_P + k4] = rb[i];
        }
    };
    f32x16 acc0, acc1, acc2, acc3;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
    const int T = (ke - kb + 31) / 32;
    gload(kb);
    sstore(smem);
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        const float* cur = smem + (t & 1) * LFW_STAGE;
        float* nxt = smem + ((t + 1) & 1) * LFW_STAGE;
        if (t + 1 < T) gload(kb + 32 * (t + 1));
        const float* at = cur + li * LFW_P + 4 * h;
        const float* bt = cur + 128 * LFW_P + (32 * wave + li) * LFW_P + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b = *(const float4*)(bt + 8 * q);
            const float4 a0 = *(const float4*)(at + 8 * q), a1 = *(const float4*)(at + 32 * LFW_P + 8 * q);
            const float4 a2 = *(const float4*)(at + 64 * LFW_P + 8 * q), a3 = *(const float4*)(at + 96 * LFW_P + 8 * q);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.x, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, b.x, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.x, b.x, acc3, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.y, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, b.y, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.y, b.y, acc3, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b.z, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.z, b.z, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.z, b.z, acc3, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b.w, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.w, b.w, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.w, b.w, acc3, 0, 0, 0);
        }
        if (t + 1 < T) sstore(nxt);
        __syncthreads();
    }
    const int n = n0 + 32 * wave + li;
    if (n < p.N) {
        const float bv = (blockIdx.z == 0 && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x16& acc = e == 0 ? acc0 : (e == 1 ? acc1 : (e == 2 ? acc2 : acc3));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * e + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < p.M) atomicAdd(&p.y[(size_t)m * p.ldy + n], acc[r] + bv);
            }
        }
    }
}
__global__ __launch_bounds__(256) void lin_zero_y_kernel(const LinBigP p) {
    const int n4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, r0 = blockIdx.y * 64 + (threadIdx.x >> 6);
    if (n4 >= p.N) return;
    for (int m = r0; m < blockIdx.y * 64 + 64 && m < p.M; m += 4) {
        float* d = p.y + (size_t)m * p.ldy + n4;
        if (n4 + 3 < p.N && (p.ldy & 3) == 0 && ((uintptr_t)p.y & 15) == 0) *(float4*)d = make_float4(0.f, 0.f, 0.f, 0.f);
        else for (int j = 0; j < 4 && n4 + j < p.N; ++j) d[j] = 0.f;
    }
}
// output ReLU (in place) and the BatchNorm column statistics of y, after all K splits have landed
__global__ __launch_bounds__(256) void lin_colstats_kernel(const LinBigP p) {
    __shared__ double red[2][4][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6, n = blockIdx.x * 64 + c, r0 = blockIdx.y * 64;
    double s = 0, q = 0;
    if (n < p.N)
        for (int m = r0 + rg; m < r0 + 64 && m < p.M; m += 4) {
            float v = p.y[(size_t)m * p.ldy + n];
            if (p.out_relu) { v = fmaxf(v, 0.f); p.y[(size_t)m * p.ldy + n] = v; }
            s += v; q += (double)v * v;
        }
    if (!p.osum) return;
    red[0][rg][c] = s; red[1][rg][c] = q;
    __syncthreads();
    if (rg == 0 && n < p.N) {
        atomicAdd(&p.osum[n], red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
        atomicAdd(&p.osumsq[n], red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
    }
}
// K split of the wide forward: smallest split whose grid fills >= 90 % of a whole number of rounds (2 workgroups x 256 CUs)
static int lfw_kc(const LinBigP& p, int& ks_out) {
    const long tiles = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    int best = 1;
    for (int ks = 1; ks <= 8 && p.K / ks >= 256; ++ks) {
        const long w = tiles * ks, rounds = (w + 511) / 512;
        best = ks;
        if (w * 10 >= rounds * 512 * 9) break;
    }
    ks_out = best;
    return (((p.K + best - 1) / best) + 31) & ~31;
}

// gradient wrt the pre-activation output: dy masked by the output ReLU
__device__ __forceinline__ float dpre_of(const LinBigP& p, float dy, float y) { return (p.out_relu && !(y > 0.f)) ? 0.f : dy; }

struct DpreRaw { float4 g, y; };
// dpre[m][n..n+3] (zero beyond column N); the 1-wide Cox head and unaligned rows take the scalar branch
__device__ __forceinline__ DpreRaw dpre_ld(const LinBigP& p, int m, int n) {
    DpreRaw r; r.g = make_float4(0.f, 0.f, 0.f, 0.f); r.y = make_float4(1.f, 1.f, 1.f, 1.f);
    if (n + 3 < p.N && ((p.lddy | p.ldy) & 3) == 0 && ((((uintptr_t)p.dy) | ((uintptr_t)p.y)) & 15) == 0) {
        r.g = *(const float4*)(p.dy + (size_t)m * p.lddy + n);
        if (p.out_relu) r.y = *(const float4*)(p.y + (size_t)m * p.ldy + n);
    } else {
        float g[4] = {0.f, 0.f, 0.f, 0.f}, y[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (n + j < p.N) { g[j] = p.dy[(size_t)m * p.lddy + n + j]; if (p.out_relu) y[j] = p.y[(size_t)m * p.ldy + n + j]; }
        r.g = make_float4(g[0], g[1], g[2], g[3]); r.y = make_float4(y[0], y[1], y[2], y[3]);
    }
    return r;
}
__device__ __forceinline__ float4 dpre_tx(const LinBigP& p, const DpreRaw& r) {
    return make_float4(dpre_of(p, r.g.x, r.y.x), dpre_of(p, r.g.y, r.y.y), dpre_of(p, r.g.z, r.y.z), dpre_of(p, r.g.w, r.y.w));
}

// ---------------------------------------------------------------------------------------------------------------------
// backward-weight:  dW[n][k] += sum_m dpre[m][n] * P(x)[m][k];  dbias[n] += sum_m dpre[m][n]
// tile rows = n, cols = k, reduction over the rows m (split over grid.z, fp32 atomics)
// ---------------------------------------------------------------------------------------------------------------------
template <bool XA>
struct LinBigBwdWOp {
    typedef LinBigP Params;
    static constexpr int WM = 2, WN = 2, WK = 1, AMODE = LD_R4, BMODE = XA ? LD_R4 : LD_K1;
    static constexpr int TM = 64, TN = 64;
    static constexpr int EXTRA = 3 * 1024;
    typedef DpreRaw ARaw;
    typedef typename std::conditional<XA, float4, float>::type BRaw;
    Prolog pr;
    int mb, me;
    __device__ void step(const Params&, int) {}
    __device__ void setup(const Params& p, int, int, int z, float* extra, int tid) {
        pr.init(p, extra, tid);
        int mc = (p.M + p.msplit - 1) / p.msplit;
        mc = (mc + 31) & ~31;
        mb = z * mc;
        me = mb + mc < p.M ? mb + mc : p.M;
    }
    __device__ void krange(const Params&, int, int& kb, int& ke) { kb = mb; ke = me; }
    __device__ ARaw a_ld(const Params& p, int, int n, int m, bool& ok) const {        // A(rows n..n+3, m) = dpre[m][n..n+3]
        ok = m < me && n < p.N;
        if (!ok) { ARaw r; r.g = make_float4(0.f, 0.f, 0.f, 0.f); r.y = r.g; return r; }
        return dpre_ld(p, m, n);
    }
    __device__ float4 a_tx(const Params& p, int, const ARaw& r, int, int, bool ok) const { return ok ? dpre_tx(p, r) : r.g; }
    __device__ BRaw b_ld(const Params& p, int, int k, int m, bool& ok) const {        // B(cols k.., m) = P(x)[m][k..]
        ok = m < me && k < p.K;
        if constexpr (XA) return ok ? *(const float4*)(p.x + (size_t)m * p.ldx + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        else return ok ? p.x[(size_t)m * p.ldx + k] : 0.f;
    }
    __device__ BRaw b_tx(const Params& p, int, const BRaw& v, int k, int m, bool ok) const {
        if constexpr (XA) {
            if (!ok) return v;
            return tail4(make_float4(pr.act(v.x, m, k), pr.act(v.y, m, k + 1), pr.act(v.z, m, k + 2), pr.act(v.w, m, k + 3)), k, p.K);
        } else {
            return ok ? pr.act(v, m, k) : 0.f;
        }
    }
    __device__ void epilogue(const Params& p, int n0r, int k0c, int, const float* Cs, int tid, bool) {
        __shared__ float red[4][64];
        if (mb >= me) return;
        // (a plain read-modify-write for msplit == 1 -- one writer per element -- measured slower than the fire-and-forget atomic)
        for (int idx = tid; idx < TM * TN; idx += 256) {
            const int r = idx / TN, c = idx % TN, n = n0r + r, k = k0c + c;
            if (n < p.N && k < p.K) atomicAdd(&p.dw[(size_t)n * p.K + k], Cs[r * (TN + 1) + c]);
        }
        if (k0c != 0 || !p.dbias) return;            // bias gradient: the workgroups of column tile 0 sum their rows of dpre
        const int c = tid & 63, rg = tid >> 6, n = n0r + c;
        float s = 0.f;
        if (n < p.N)
            for (int m = mb + rg; m < me; m += 4) s += dpre_of(p, p.dy[(size_t)m * p.lddy + n], p.out_relu ? p.y[(size_t)m * p.ldy + n] : 1.f);
        red[rg][c] = s;
        __syncthreads();
        if (rg == 0 && n < p.N) atomicAdd(&p.dbias[n], red[0][c] + red[1][c] + red[2][c] + red[3][c]);
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// backward-weight of a layer WITHOUT input prologue (the first layer: x is the raw input), large shapes: 128 x 128 output tiles.
// Both operands are row-contiguous (dpre[m][n..], x[m][k..]), so their LDS images are m-major.  Instead of one ds_read_b32 per
// MFMA operand, a lane reads FOUR consecutive n of one row with a single ds_read_b128 and feeds them to four MFMAs whose
// 32 output rows are the interleaved sets {4i + e}: a wave owns all 128 n of the tile x 32 k columns (4 accumulators), the
// x fragment (one ds_read_b32) is shared by the four.  Per 64 MFMAs: 16 b128 + 16 b32 LDS reads (the GEMM-core form: 128 b32)
// and half the global loads.  Rows are split over grid.z (fp32 atomics, 128-byte runs along k).
// ---------------------------------------------------------------------------------------------------------------------
#define LBW_P 132
#define LBW_STAGE (2 * 32 * LBW_P)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void lin_bwdw_wide_kernel(const LinBigP p, const int msplit) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 128, k0 = blockIdx.y * 128;
    int mc = (p.M + msplit - 1) / msplit;
    mc = (mc + 31) & ~31;
    const int mb = blockIdx.z * mc, me = mb + mc < p.M ? mb + mc : p.M;
    if (mb >= me) return;
    const int c4 = (tid & 31) * 4, r8 = tid >> 5;
    const bool nok = n0 + c4 < p.N, kok = k0 + c4 + 3 < p.ldx;
    DpreRaw ra[4];
    float4 rb[4];
    auto gload = [&](int r0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = r0 + r8 + 8 * i;
            const bool ok = m < me;
            if (ok && nok) ra[i] = dpre_ld(p, m, n0 + c4);
            else { ra[i].g = make_float4(0.f, 0.f, 0.f, 0.f); ra[i].y = ra[i].g; }
            rb[i] = ok && kok ? *(const float4*)(p.x + (size_t)m * p.ldx + k0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto sstore = [&](float* st) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(float4*)&st[(r8 + 8 * i) * LBW_P + c4] = dpre_tx(p, ra[i]);
            *(float4*)&st[32 * LBW_P + (r8 + 8 * i) * LBW_P + c4] = tail4(rb[i], k0 + c4, p.K);
        }
    };
    f32x16 acc0, acc1, acc2, acc3;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
    const bool do_bias = blockIdx.y == 0 && p.dbias != nullptr;
    float bsum = 0.f;
    const int T = (me - mb + 31) / 32;
    gload(mb);
    sstore(smem);
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        const float* cur = smem + (t & 1) * LBW_STAGE;
        float* nxt = smem + ((t + 1) & 1) * LBW_STAGE;
        if (t + 1 < T) gload(mb + 32 * (t + 1));
        const float* at = cur + h * LBW_P + 4 * li;
        const float* bt = cur + 32 * LBW_P + h * LBW_P + 32 * wave + li;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float4 a = *(const float4*)(at + 2 * q * LBW_P);
            const float b = bt[2 * q * LBW_P];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b, acc3, 0, 0, 0);
        }
        if (do_bias && tid < 128) {
#pragma unroll 8
            for (int mm = 0; mm < 32; ++mm) bsum += cur[mm * LBW_P + tid];
        }
        if (t + 1 < T) sstore(nxt);
        __syncthreads();
    }
    const int k = k0 + 32 * wave + li;
    if (k < p.K) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x16& acc = e == 0 ? acc0 : (e == 1 ? acc1 : (e == 2 ? acc2 : acc3));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + 4 * ((r & 3) + 8 * (r >> 2) + 4 * h) + e;
                if (n < p.N) atomicAdd(&p.dw[(size_t)n * p.K + k], acc[r]);
            }
        }
    }
    if (do_bias && tid < 128 && n0 + tid < p.N) atomicAdd(&p.dbias[n0 + tid], bsum);
}
// rows split of the wide kernel: the smallest split whose grid fills >= 90 % of a whole number of rounds (2 workgroups x 256 CUs)
static int lbw_msplit(const LinBigP& p) {
    const long tiles = (long)((p.N + 127) / 128) * ((p.K + 127) / 128);
    int best = 1;
    for (int ms = 1; ms <= 8 && p.M / ms >= 256; ++ms) {
        const long w = tiles * ms, rounds = (w + 511) / 512;
        best = ms;
        if (w * 10 >= rounds * 512 * 9) break;
    }
    return best;
}

// ---------------------------------------------------------------------------------------------------------------------
// backward-data:  dP[m][k] = sum_n dpre[m][n] * W[n][k]  ->  through dropout and ReLU:  dbn[m][k], plus the BatchNorm-backward
// column sums s1 = sum_m dbn, s2 = sum_m dbn * (x - mean)      (K % 4 == 0: every hidden width)
// ---------------------------------------------------------------------------------------------------------------------
struct LinBigBwdXOp {
    typedef LinBigP Params;
    static constexpr int WM = 2, WN = 2, WK = 1, AMODE = LD_K4, BMODE = LD_R4;
    static constexpr int TM = 64, TN = 64;
    static constexpr int EXTRA = 3 * 1024;
    typedef DpreRaw ARaw;
    typedef float4 BRaw;
    Prolog pr;
    __device__ void step(const Params&, int) {}
    __device__ void setup(const Params& p, int, int, int, float* extra, int tid) { pr.init(p, extra, tid); }
    __device__ void krange(const Params& p, int, int& kb, int& ke) { kb = 0; ke = p.N; }
    __device__ ARaw a_ld(const Params& p, int, int m, int n, bool& ok) const {        // A(m, n..n+3) = dpre[m][n..n+3]
        ok = m < p.M && n < p.N;
        if (!ok) { ARaw r; r.g = make_float4(0.f, 0.f, 0.f, 0.f); r.y = r.g; return r; }
        return dpre_ld(p, m, n);
    }
    __device__ float4 a_tx(const Params& p, int, const ARaw& r, int, int, bool ok) const { return ok ? dpre_tx(p, r) : r.g; }
    __device__ float4 b_ld(const Params& p, int, int k, int n, bool& ok) const {      // B(cols k..k+3, n) = W[n][k..k+3]
        ok = n < p.N && k < p.K;
        return ok ? *(const float4*)(p.w + (size_t)n * p.K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __device__ float4 b_tx(const Params&, int, const float4& v, int, int, bool) const { return v; }
    __device__ void epilogue(const Params& p, int m0, int k0, int, const float* Cs, int tid, bool) {
        __shared__ double red[2][4][64];
        const int c = tid & 63, rg = tid >> 6, k = k0 + c;
        double s1 = 0, s2 = 0;
        if (k < p.K) {
            const int rows = p.M - m0 < TM ? p.M - m0 : TM;
            for (int r = rg; r < rows; r += 4) {
                const int m = m0 + r;
                float g = Cs[r * (TN + 1) + c] * pr.scale(m, k);
                if (pr.has_bn) {
                    const float xc = p.x[(size_t)m * p.ldx + k] - pr.mean[k];
                    g = fmaf(xc, pr.sc[k], pr.beta[k]) > 0.f ? g : 0.f;
                    s1 += g; s2 += (double)g * xc;
                }
                p.dbn[(size_t)m * p.lddbn + k] = g;
            }
        }
        if (!pr.has_bn) return;
        red[0][rg][c] = s1; red[1][rg][c] = s2;
        __syncthreads();
        if (rg == 0 && k < p.K) {
            atomicAdd(&p.s1[k], red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
            atomicAdd(&p.s2[k], red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
        }
    }
};

// BatchNorm1d backward (training mode):  dx = gamma * rstd * (dbn - mean_m(dbn) - xhat * mean_m(dbn * xhat));
// dgamma += sum_m dbn * xhat, dbeta += sum_m dbn          (s2 holds sum_m dbn * (x - mean): xhat's rstd is applied here)
__global__ __launch_bounds__(256) void bn1d_bwd_apply_kernel(const LinBigP p) {
    const int k = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    if (k >= p.K) return;
    float mu, rstd;
    bn_mean_rstd(p.bn, k, mu, rstd);
    const float ga = p.bn.gamma[k];
    const float t2 = (float)(p.s2[k] * (double)rstd);
    const float m1 = (float)(p.s1[k] * (double)p.bn.inv_count), m2 = t2 * p.bn.inv_count;
    if (blockIdx.y == 0 && rg == 0) {
        if (p.dgamma) p.dgamma[k] += t2;
        if (p.dbeta) p.dbeta[k] += (float)p.s1[k];
    }
    const int r0 = blockIdx.y * 64;
    for (int r = r0 + rg; r < r0 + 64 && r < p.M; r += 4) {
        const float xh = (p.x[(size_t)r * p.ldx + k] - mu) * rstd;
        p.dx[(size_t)r * p.lddx + k] = ga * rstd * (p.dbn[(size_t)r * p.lddbn + k] - m1 - xh * m2);
    }
}

bool args_ok(const LinBigP& p) {
    return p.x && p.w && p.y && p.M > 0 && p.K > 0 && p.N > 0 && p.ldx >= p.K && p.ldy >= p.N &&
           (!p.has_bn || (p.K <= 1024 && p.bn.gamma && p.bn.beta && (p.bn.train ? (p.bn.sum && p.bn.sumsq) : (p.bn.rmean && p.bn.rvar))));
}
// float4 loads of the rows of x: 16-B aligned, row pitch a multiple of 4 floats that covers roundup4(K)
bool x_aligned(const LinBigP& p) { return (p.ldx & 3) == 0 && p.ldx >= ((p.K + 3) & ~3) && ((uintptr_t)p.x & 15) == 0; }
bool w_aligned(const LinBigP& p) { return (p.K & 3) == 0 && ((uintptr_t)p.w & 15) == 0; }      // backward-data: float4 along k from k-major images
bool w_vec(const LinBigP& p) { return ((uintptr_t)p.w & 3) == 0; }                                 // forward: 4-byte aligned 16-byte loads

}  // namespace

extern "C" int mms_linear_big_fwd(const LinBigP* pp, hipStream_t s) {
    if (!pp || !args_ok(*pp)) return MMS_ERR_ARG;
    const LinBigP& p = *pp;
    const bool xa = x_aligned(p), wa = w_vec(p);
    const bool plain = !p.has_bn && !(p.train && (p.drop_mask || p.drop_p > 0.f));
    if (plain && xa && wa && p.M >= 256 && p.N >= 128 && p.K >= 512 && !p.core_only) {
        constexpr int smem = 2 * LFW_STAGE * (int)sizeof(float);             // 73.7 KB: 2 workgroups per CU
        static std::once_flag attr_once;
        std::call_once(attr_once, [&] { hipFuncSetAttribute((const void*)lin_fwd_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem); });
        int ks = 1;
        const int kc = lfw_kc(p, ks);
        MMS_LAUNCH(lin_zero_y_kernel, dim3((p.N + 255) / 256, (p.M + 63) / 64), dim3(256), 0, s, p);
        MMS_LAUNCH(lin_fwd_wide_kernel, dim3((p.M + 127) / 128, (p.N + 127) / 128, ks), dim3(256), smem, s, p, kc);
        if (p.osum || p.out_relu) MMS_LAUNCH(lin_colstats_kernel, dim3((p.N + 63) / 64, (p.M + 63) / 64), dim3(256), 0, s, p);
        return mms_check_launch();
    }
    const dim3 g((p.M + 63) / 64, (p.N + 63) / 64, 1);
    if (xa && wa) return launch_tile_gemm<LinBigFwdOp<true, true>>(pp, 1, g, s);
    if (xa) return launch_tile_gemm<LinBigFwdOp<true, false>>(pp, 1, g, s);
    if (wa) return launch_tile_gemm<LinBigFwdOp<false, true>>(pp, 1, g, s);
    return launch_tile_gemm<LinBigFwdOp<false, false>>(pp, 1, g, s);
}

extern "C" int mms_linear_big_bwd_w(const LinBigP* pp, hipStream_t s) {
    if (!pp || !args_ok(*pp) || !pp->dy || !pp->dw || pp->msplit <= 0 || pp->lddy < pp->N) return MMS_ERR_ARG;
    const LinBigP& p = *pp;
    const bool plain = !p.has_bn && !(p.train && (p.drop_mask || p.drop_p > 0.f));
    if (plain && x_aligned(p) && p.N >= 128 && p.K >= 128 && p.M >= 256 && !p.core_only) {
        constexpr int smem = 2 * LBW_STAGE * (int)sizeof(float);             // 67.6 KB: 2 workgroups per CU
        static std::once_flag attr_once;
        std::call_once(attr_once, [&] { hipFuncSetAttribute((const void*)lin_bwdw_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem); });
        const int ms = lbw_msplit(p);
        MMS_LAUNCH(lin_bwdw_wide_kernel, dim3((p.N + 127) / 128, (p.K + 127) / 128, ms), dim3(256), smem, s, p, ms);
        return mms_check_launch();
    }
    const dim3 g((p.N + 63) / 64, (p.K + 63) / 64, p.msplit);
    return x_aligned(p) ? launch_tile_gemm<LinBigBwdWOp<true>>(pp, 1, g, s) : launch_tile_gemm<LinBigBwdWOp<false>>(pp, 1, g, s);
}

extern "C" int mms_linear_big_bwd_x(const LinBigP* pp, hipStream_t s) {
    if (!pp || !args_ok(*pp) || !pp->dy || !pp->dbn || pp->lddy < pp->N || pp->lddbn < pp->K) return MMS_ERR_ARG;
    const LinBigP& p = *pp;
    if (!w_aligned(p) || (p.has_bn && (!p.s1 || !p.s2))) return MMS_ERR_ARG;
    return launch_tile_gemm<LinBigBwdXOp>(pp, 1, dim3((p.M + 63) / 64, (p.K + 63) / 64, 1), s);
}

extern "C" int mms_bn1d_bwd_apply(const LinBigP* pp, hipStream_t s) {
    if (!pp || !pp->has_bn || !pp->bn.train || !pp->dbn || !pp->dx || !pp->s1 || !pp->s2 || !pp->x || pp->M <= 0 || pp->K <= 0) return MMS_ERR_ARG;
    const LinBigP& p = *pp;
    MMS_LAUNCH(bn1d_bwd_apply_kernel, dim3((p.K + 63) / 64, (p.M + 63) / 64), dim3(256), 0, s, p);
    return mms_check_launch();
}
