// DenseNet121-3D forward ops for gfx950: fused BN+ReLU prologues, fp32 MFMA contractions, batch-statistic
// epilogues.  Replaces the torch/MIOpen op sequence behind MONAI's DenseNet121 at the reference call sites
// final_multimodal.py:66-71, partial_modality_training.py:171-176, simple_fusion.py:182-187.
#include "dn_ops.h"
#include "tile_gemm.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------------------------------
// shared epilogue pieces
// ------------------------------------------------------------------------------------------------------
template <int TM, int TN>
__device__ __forceinline__ void store_tile(float* y, int ldy, int M, int N, int m0, int n0, const float* Cs, int tid) {
    for (int idx = tid; idx < TM * TN; idx += 256) {
        int r = idx / TN, c = idx % TN, m = m0 + r, n = n0 + c;
        if (m < M && n < N) y[(size_t)m * ldy + n] = Cs[r * (TN + 1) + c];
    }
}
template <int TM, int TN>
__device__ __forceinline__ void tile_col_stats(double* osum, double* osumsq, int M, int N, int m0, int n0,
                                               const float* Cs, int tid) {
#ifdef MMS_ABLATE_STATS
    return;
#endif
    if (osum == nullptr || tid >= TN) return;
    int n = n0 + tid;
    if (n >= N) return;
    double s = 0, q = 0;
    int rows = M - m0 < TM ? M - m0 : TM;
    for (int r = 0; r < rows; ++r) {
        double v = Cs[r * (TN + 1) + tid];
        s += v; q += v * v;
    }
    atomicAdd(&osum[n], s);
    atomicAdd(&osumsq[n], q);
}

// ------------------------------------------------------------------------------------------------------
// 1x1x1 conv
// ------------------------------------------------------------------------------------------------------
template <int WM_, int WN_, int WK_, bool POOL, bool KSPLIT = false>
struct Conv1FwdOp {
    typedef Conv1FwdP Params;
    static constexpr int WM = WM_, WN = WN_, WK = WK_, AMODE = LD_K4, BMODE = LD_K4;
    static constexpr bool SINGLE_BUF = WK_ == 4;       // 128-deep tiles: one LDS buffer (tile_gemm.h)
    __device__ void step(const Params&, int) {}
    static constexpr int TM = 32 * WM, TN = 32 * WN;
    static constexpr int EXTRA = 3 * 1024 + TM;
    const float *mean, *sc, *beta;
    const int* srcbase;
    int m0;
    __device__ void setup(const Params& p, int m0_, int, int, float* extra, int tid) {
        mean = extra; sc = extra + 1024; beta = extra + 2048; srcbase = (const int*)(extra + 3072); m0 = m0_;
        bn_consts_to_lds<4>(p.bn, p.K, tid, extra, extra + 1024, extra + 2048);
        if (POOL && tid < TM) {
            int m = m0 + tid, base = -1;
            if (m < p.M) {
                int D2 = p.in.D >> 1, H2 = p.in.H >> 1, W2 = p.in.W >> 1, vox2 = D2 * H2 * W2;
                int b = m / vox2, r = m % vox2, d = r / (H2 * W2), h = (r / W2) % H2, w = r % W2;
                base = ((b * p.in.D + 2 * d) * p.in.H + 2 * h) * p.in.W + 2 * w;
            }
            ((int*)extra)[3072 + tid] = base;
        }
    }
    __device__ void krange(const Params& p, int z, int& kb, int& ke) {
        if (KSPLIT) {
            const int per = (((p.K + 127) / 128 + p.ksplit - 1) / p.ksplit) * 128;   // whole K-steps (TK = 128) per workgroup
            kb = z * per; ke = kb + per < p.K ? kb + per : p.K;
        } else { kb = 0; ke = p.K; }
    }
    __device__ float4 act4(const float4 v, int k) const {
        float4 r;
        r.x = fmaxf(bn_apply(v.x, mean[k], sc[k], beta[k]), 0.f);
        r.y = fmaxf(bn_apply(v.y, mean[k + 1], sc[k + 1], beta[k + 1]), 0.f);
        r.z = fmaxf(bn_apply(v.z, mean[k + 2], sc[k + 2], beta[k + 2]), 0.f);
        r.w = fmaxf(bn_apply(v.w, mean[k + 3], sc[k + 3], beta[k + 3]), 0.f);
        return r;
    }
    typedef float4 ARaw;
    typedef float4 BRaw;
    // Loaders are BRANCH-FREE: a predicated load compiles to an exec-masked block that re-reads its kernel arguments (s_load + wait) --
    // one serial scalar round trip per piece and K-step.  Out-of-range pieces load from a clamped (valid) address and are zeroed in *_tx.
    __device__ float4 a_ld(const Params& p, int, int m, int k, bool& ok) const {
        ok = m < p.M && k < p.K;
        if (!POOL) return *(const float4*)(p.x + (size_t)(m < p.M ? m : p.M - 1) * p.ldx + (k < p.K ? k : p.K - 4));
        if (!ok) return make_float4(0, 0, 0, 0);
        // transition: 8 source voxels per pooled row -- activation applied while loading (6 launches per step)
        const int base = srcbase[m - m0], HW = p.in.H * p.in.W, W = p.in.W;
        float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            int src = base + (o >> 2) * HW + ((o >> 1) & 1) * W + (o & 1);
            float4 v = act4(*(const float4*)(p.x + (size_t)src * p.ldx + k), k);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        return make_float4(s.x * 0.125f, s.y * 0.125f, s.z * 0.125f, s.w * 0.125f);
    }
    __device__ float4 a_tx(const Params& p, int, const float4& v, int, int k, bool ok) const {
        if (POOL) return v;
        const float4 r = act4(v, k < p.K ? k : p.K - 4);
        const float z = ok ? 1.f : 0.f;
        return make_float4(z * r.x, z * r.y, z * r.z, z * r.w);
    }
    __device__ float4 b_ld(const Params& p, int, int n, int k, bool& ok) const {
        ok = n < p.N && k < p.K;
        return *(const float4*)(p.w + (size_t)(n < p.N ? n : p.N - 1) * p.K + (k < p.K ? k : p.K - 4));
    }
    __device__ float4 b_tx(const Params&, int, const float4& v, int, int, bool ok) const { return ok ? v : make_float4(0, 0, 0, 0); }
    __device__ void epilogue(const Params& p, int m0_, int n0, int z, const float* Cs, int tid, bool active) {
        if (!active) return;
        if (KSPLIT) {        // publish the partial tile; the last-arriving workgroup of the tile sums them (fixed order)
            for (int idx = tid; idx < TM * TN; idx += 256) {
                const int r = idx / TN, c = idx % TN, m = m0_ + r, n = n0 + c;
                if (m < p.M && n < p.N) pstore(&p.partial[((size_t)z * p.M + m) * p.N + n], Cs[r * (TN + 1) + c]);
            }
            if (!tile_last_arriver(p.counters + (m0_ / TM) * ((p.N + TN - 1) / TN) + n0 / TN, (unsigned)p.ksplit, tid)) return;
            float* C = const_cast<float*>(Cs);
            for (int idx = tid; idx < TM * TN; idx += 256) {
                const int r = idx / TN, c = idx % TN, m = m0_ + r, n = n0 + c;
                float v[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = (t < p.ksplit && m < p.M && n < p.N) ? pload(&p.partial[((size_t)t * p.M + m) * p.N + n]) : 0.f;
                float a = 0.f;
#pragma unroll
                for (int t = 0; t < 8; ++t) a += v[t];
                C[r * (TN + 1) + c] = a;
            }
            __syncthreads();
        }
        store_tile<TM, TN>(p.y, p.ldy, p.M, p.N, m0_, n0, Cs, tid);
        tile_col_stats<TM, TN>(stat_rep(p.osum, p.srep, p.sstride), stat_rep(p.osumsq, p.srep, p.sstride), p.M, p.N, m0_, n0, Cs, tid);
    }
};

// ------------------------------------------------------------------------------------------------------
// transition pre-pass: pooled[m'][k] = mean of relu(bn(x)) over the 2x2x2 source voxels of pooled row m' (summed in the order the
// pooled GEMM loaders use, then x 0.125).  A thread owns 4 channels (constants in registers) and walks PA_ROWS pooled rows: 8 float4 loads
// per row, all rows' loads requested before the first use.  Grid: (K / 4 / 64 column groups, row groups, models).
// ------------------------------------------------------------------------------------------------------
#define PA_ROWS 2
__global__ __launch_bounds__(256) void pool_act_kernel(const Grp<PoolActP> grp) {
    const PoolActP& p = grp.p[blockIdx.z];
    const int tid = threadIdx.x, c4 = ((int)blockIdx.x * 64 + (tid & 63)) * 4;
    if (c4 >= p.K) return;
    float mean[4], sc[4], beta[4];
    bn_consts4(p.bn, c4, mean, sc, beta);
    const int D2 = p.in.D >> 1, H2 = p.in.H >> 1, W2 = p.in.W >> 1, vox2 = D2 * H2 * W2, HW = p.in.H * p.in.W, W = p.in.W;
    const int r0 = ((int)blockIdx.y * 4 + (tid >> 6)) * PA_ROWS;
    float4 v[PA_ROWS][8];
#pragma unroll
    for (int i = 0; i < PA_ROWS; ++i) {
        const int m = r0 + i < p.Mout ? r0 + i : p.Mout - 1;
        const int b = m / vox2, r = m % vox2, d = r / (H2 * W2), h = (r / W2) % H2, w = r % W2;
        const int base = ((b * p.in.D + 2 * d) * p.in.H + 2 * h) * p.in.W + 2 * w;
#pragma unroll
        for (int o = 0; o < 8; ++o) v[i][o] = *(const float4*)(p.x + (size_t)(base + (o >> 2) * HW + ((o >> 1) & 1) * W + (o & 1)) * p.ldx + c4);
    }
#pragma unroll
    for (int i = 0; i < PA_ROWS; ++i) {
        if (r0 + i >= p.Mout) break;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            a.x += fmaxf(bn_apply(v[i][o].x, mean[0], sc[0], beta[0]), 0.f); a.y += fmaxf(bn_apply(v[i][o].y, mean[1], sc[1], beta[1]), 0.f);
            a.z += fmaxf(bn_apply(v[i][o].z, mean[2], sc[2], beta[2]), 0.f); a.w += fmaxf(bn_apply(v[i][o].w, mean[3], sc[3], beta[3]), 0.f);
        }
        *(float4*)(p.y + (size_t)(r0 + i) * p.ldy + c4) = make_float4(a.x * 0.125f, a.y * 0.125f, a.z * 0.125f, a.w * 0.125f);
    }
}
extern "C" int mms_pool_act_group(const PoolActP* pp, int ng, hipStream_t s) {
    Grp<PoolActP> a;
    if (!grp_fill(a, pp, ng, 1)) return MMS_ERR_ARG;
    const PoolActP& p = *pp;
    if (p.K <= 0 || p.K % 4 != 0 || p.Mout <= 0 || (p.in.D | p.in.H | p.in.W) & 1 || p.in.D <= 0 || p.in.H <= 0 || p.in.W <= 0 ||
        (long)p.Mout * 8 % ((long)p.in.D * p.in.H * p.in.W) != 0) return MMS_ERR_ARG;
    for (int g = 0; g < ng; ++g) {
        const PoolActP& q = pp[g];
        if (q.K != p.K || q.Mout != p.Mout || q.in.D != p.in.D || q.in.H != p.in.H || q.in.W != p.in.W || q.ldx % 4 != 0 || q.ldy % 4 != 0 ||
            (((uintptr_t)q.x | (uintptr_t)q.y) & 15) != 0 || !mms_bn_aligned16(q.bn)) return MMS_ERR_ARG;
    }
    MMS_LAUNCH(pool_act_kernel, dim3((p.K / 4 + 63) / 64, (p.Mout + 4 * PA_ROWS - 1) / (4 * PA_ROWS), ng), dim3(256), 0, s, a);
    return mms_check_launch();
}

extern "C" int mms_conv1_fwd_group(const Conv1FwdP* pp, int ng, const MmsDnOpts* opts, hipStream_t s) {
    if (!pp || ng < 1 || ng > MMS_MAX_GROUP) return MMS_ERR_ARG;
    const MmsDnOpts o = mms_opts(opts);
    const Conv1FwdP& p = *pp;
    if (p.K % 32 != 0 || p.K > 1024 || p.ldx % 4 != 0 || p.M <= 0) return MMS_ERR_ARG;
    if (p.pool && ((p.in.D | p.in.H | p.in.W) & 1)) return MMS_ERR_ARG;
    for (int g = 1; g < ng; ++g) {
        const Conv1FwdP& q = pp[g];
        if (q.M != p.M || q.N != p.N || q.K != p.K || q.ldx % 4 != 0 || q.pool != p.pool || q.in.D != p.in.D || q.in.H != p.in.H ||
            q.in.W != p.in.W) return MMS_ERR_ARG;
    }
    if (mms_conv1_small_ok(p, ng, o)) {      // few rows: 16 x 16 tiles over the whole K range from LDS-resident panels (dn_c1s.hip)
        for (int g = 0; g < ng; ++g) if (((uintptr_t)pp[g].x | (uintptr_t)pp[g].w) & 15) return MMS_ERR_ARG;
        return mms_c1s_fwd(pp, ng, s);
    }
    // big M: 64x64 tiles, no in-workgroup K split; small M: 32x32 tiles with the 4 waves splitting K
    // tile shape from the whole group's work (MmsDnOpts.big_ng = -1: from one model's -- tests that need ng-independent arithmetic)
    const bool big = (long)p.M * p.N * (o.big_ng < 0 ? 1 : ng) >= 256L * 64 * 64;
    if (big) {
        dim3 g((p.M + 63) / 64, (p.N + 63) / 64, 1);
        return p.pool ? launch_tile_gemm<Conv1FwdOp<2, 2, 1, true>>(pp, ng, g, s)
                      : launch_tile_gemm<Conv1FwdOp<2, 2, 1, false>>(pp, ng, g, s);
    }
    dim3 g((p.M + 31) / 32, (p.N + 31) / 32, 1);
    if (p.partial && p.counters && p.ksplit > 1 && !p.pool) {
        if (p.ksplit > 8 || (long)(p.ksplit - 1) * ((((p.K + 127) / 128 + p.ksplit - 1) / p.ksplit) * 128) >= p.K) return MMS_ERR_ARG;   // every slice owns >= 1 channel
        for (int i = 1; i < ng; ++i) if (pp[i].ksplit != p.ksplit || !pp[i].partial || !pp[i].counters) return MMS_ERR_ARG;
        g.z = p.ksplit;
        return launch_tile_gemm<Conv1FwdOp<1, 1, 4, false, true>>(pp, ng, g, s);
    }
    return p.pool ? launch_tile_gemm<Conv1FwdOp<1, 1, 4, true>>(pp, ng, g, s)
                  : launch_tile_gemm<Conv1FwdOp<1, 1, 4, false>>(pp, ng, g, s);
}
MMS_SINGLE_O(mms_conv1_fwd, Conv1FwdP)

// ------------------------------------------------------------------------------------------------------
// 3x3x3 conv (pad 1): implicit GEMM, K = (tap, cin) = 27*128, one K-step per tap
// ------------------------------------------------------------------------------------------------------
// per-axis tap validity of a voxel, 3 bits per axis (bit t <-> tap offset t-1): [0..2] d, [3..5] h, [6..8] w
__device__ __forceinline__ unsigned tap_mask9(int c, Dims3 g, bool mirror) {
    int d, h, w;
    unpack_dhw(c, d, h, w);
    const unsigned lo_d = d > 0, hi_d = d + 1 < g.D, lo_h = h > 0, hi_h = h + 1 < g.H, lo_w = w > 0, hi_w = w + 1 < g.W;
    // forward reads voxel + (t-1): t=0 needs the low neighbour; backward-data reads voxel - (t-1): mirrored
    const unsigned dm = mirror ? (hi_d | 2u | (lo_d << 2)) : (lo_d | 2u | (hi_d << 2));
    const unsigned hm = mirror ? (hi_h | 2u | (lo_h << 2)) : (lo_h | 2u | (hi_h << 2));
    const unsigned wm = mirror ? (hi_w | 2u | (lo_w << 2)) : (lo_w | 2u | (hi_w << 2));
    return dm | (hm << 3) | (wm << 6);
}

template <bool SPLIT>
struct Conv3FwdOp {
    typedef Conv3FwdP Params;
    static constexpr int WM = 1, WN = 1, WK = 4, AMODE = LD_K4, BMODE = LD_K4;
    static constexpr int TM = 32, TN = 32;
    static constexpr int EXTRA = 4;
    typedef float4 ARaw;
    typedef float4 BRaw;
    // TK = 128 = all input channels of one tap: a thread's 4 channels ((tid & 31) * 4 ...) and its 4 tile rows
    // (tid/32 + 8i) are the same at every K-step.  Everything per-thread is computed once: BatchNorm constants,
    // byte offsets into y1 / the packed weights, and a 9-bit tap-validity mask per row.  A K-step's loader is then
    // 8 buffer loads + ~3 VALU each; the per-tap part of every address is wave-uniform (SGPR).
    float mean[4], sc[4], beta[4];
    int voff[4], woff[4];
    unsigned m9[4];
    buf_rsrc_t ry, rw;
    int tapoff_b, wsoff, p_nsplit;
    unsigned sel;
    __device__ void setup(const Params& p, int m0, int, int, float*, int tid) {
        p_nsplit = p.nsplit > 0 ? p.nsplit : 27;
        const int c0 = (tid & 31) * 4;
        bn_consts4(p.bn, c0, mean, sc, beta);
        ry = make_rsrc(p.y1, (unsigned)p.M * 512u);
        rw = make_rsrc(p.wp, 32u * 27u * 128u * 4u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 5) + 8 * i, m = m0 + row;
            const bool valid = m < p.M;
            m9[i] = valid ? tap_mask9(p.coords[valid ? m : m0], p.g, false) : 0u;
            voff[i] = ((valid ? m : m0) * 128 + c0) * 4;
            woff[i] = (row * (27 * 128) + c0) * 4;
        }
    }
    __device__ void krange(const Params&, int z, int& kb, int& ke) {
        if (SPLIT) {
            const int tpw = (27 + p_nsplit - 1) / p_nsplit;
            kb = z * tpw * 128; ke = kb + tpw * 128; if (ke > 27 * 128) ke = 27 * 128;
        } else { kb = 0; ke = 27 * 128; }
    }
    __device__ void step(const Params& p, int k0) {
        const int tap = k0 >> 7, kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        sel = (1u << kd) | (8u << kh) | (64u << kw);
        tapoff_b = (((kd - 1) * p.g.H + (kh - 1)) * p.g.W + (kw - 1)) * 512;
        wsoff = k0 * 4;
    }
    __device__ float4 a_ld(const Params&, int i, int, int, bool& ok) const {
        ok = (m9[i] & sel) == sel;               // zero padding is applied AFTER bn+relu (a_tx)
        return buf_load4(ry, voff[i] + (ok ? tapoff_b : 0), 0);
    }
    __device__ float4 a_tx(const Params&, int, const float4& v, int, int, bool ok) const {   // branch-free
        const float z = ok ? 1.f : 0.f;   // relu output * {0,1}: exact
        return make_float4(z * fmaxf(bn_apply(v.x, mean[0], sc[0], beta[0]), 0.f), z * fmaxf(bn_apply(v.y, mean[1], sc[1], beta[1]), 0.f),
                           z * fmaxf(bn_apply(v.z, mean[2], sc[2], beta[2]), 0.f), z * fmaxf(bn_apply(v.w, mean[3], sc[3], beta[3]), 0.f));
    }
    __device__ float4 b_ld(const Params&, int i, int, int, bool& ok) const {
        ok = true;
        return buf_load4(rw, woff[i], wsoff);
    }
    __device__ float4 b_tx(const Params&, int, const float4& v, int, int, bool) const { return v; }
    __device__ void epilogue(const Params& p, int m0_, int n0, int z, const float* Cs, int tid, bool active) {
        if (!active) return;
        if (SPLIT) {      // (a last-arriver fixup as in Conv1FwdOp was measured here too: slower than the reduce launch -- one
                          // workgroup reading up to 27 partial tiles is a longer serial tail than the launch it saves)
            store_tile<TM, TN>(p.partial + (size_t)z * p.M * 32, 32, p.M, 32, m0_, n0, Cs, tid);
            return;
        }
        store_tile<TM, TN>(p.out, p.ldo, p.M, 32, m0_, n0, Cs, tid);
        tile_col_stats<TM, TN>(stat_rep(p.osum, p.srep, p.sstride), stat_rep(p.osumsq, p.srep, p.sstride), p.M, 32, m0_, n0, Cs, tid);
    }
};

// sums the 27 tap partials (fixed order: deterministic), writes the slab columns, accumulates the batch statistics
__global__ __launch_bounds__(256) void conv3_fwd_reduce_kernel(const Grp<Conv3FwdP> grp) {
    const Conv3FwdP& p = grp.p[blockIdx.z];
    __shared__ double red[2][8][32];
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    double s = 0, q = 0;
    for (int m = blockIdx.x * 8 + rg; m < p.M && m < blockIdx.x * 8 + 8; m += 8) {
        float v[27];
#pragma unroll
        for (int t = 0; t < 27; ++t) v[t] = t < p.nsplit ? p.partial[((size_t)t * p.M + m) * 32 + c] : 0.f;    // loads in flight
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < 27; ++t) a += v[t];
        p.out[(size_t)m * p.ldo + c] = a;
        s += a; q += (double)a * a;
    }
    if (p.osum == nullptr) return;
    red[0][rg][c] = s; red[1][rg][c] = q;
    __syncthreads();
    if (rg == 0) {
        double a = 0, b = 0;
#pragma unroll
        for (int g = 0; g < 8; ++g) { a += red[0][g][c]; b += red[1][g][c]; }
        atomicAdd(&stat_rep(p.osum, p.srep, p.sstride)[c], a);
        atomicAdd(&stat_rep(p.osumsq, p.srep, p.sstride)[c], b);
    }
}

// ------------------------------------------------------------------------------------------------------
// 3x3x3 conv, multi-tap form for launches with many rows (>= 512 tiles of 64 rows over the group, W <= 16).
// Conv3FwdOp loads and BN+ReLU-transforms a tile's A rows once PER TAP (27x).  Rows are voxels in (d, h, w) order, so the
// 9 taps of one kd read rows m + (kh-1)*W + (kw-1): a window of TM + 2(W+1) consecutive rows.  This kernel stages that
// window of relu(bn(y1)) in LDS once per kd (3x per tile instead of 27x) and reads the A operand of each tap from the
// shifted slot; the weight operand of a tap goes from L1/L2 straight into the MFMA register layout (prefetched one tap
// ahead), so there is no weight tile in LDS and no barrier per tap.  Zero padding = a per-(row, tap) 0/1 factor on the A registers (4 VALU per 4 MFMAs).
// Workgroup = 64 rows x 32 output channels; 4 waves = 2 row tiles x 2 halves of the 128 input channels (summed through
// LDS at the end); 27 taps x 32 MFMAs per wave.  LDS: window (64 + 2*17) x 132 floats + 2 weight tiles = 85.5 KB.
// ------------------------------------------------------------------------------------------------------
// Two tile heights: 64 rows (2 row tiles x 2 channel halves) when the launch has >= 512 such tiles, else 32 rows (1 row tile x 4
// channel quarters, window 32 + 2(W+1) rows = 26 KB at W = 8): the block-1 launches of 2-3 fold models (512 / 768 tiles, 2-3 per CU).
#define C3M_PITCH 132
#define C3M_MAXHALO 17
// -DC3M_TIMING (tools/c3m_timing.py): shader-clock stamps of the kernel's phases, one record of 8 words per workgroup
#ifdef C3M_TIMING
__device__ unsigned long long* c3m_ts_buf = nullptr;
extern "C" int mms_c3m_timing_buffer(void* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(c3m_ts_buf), &buf, sizeof(buf)) == hipSuccess ? MMS_OK : MMS_ERR_LAUNCH; }
#define C3M_TS_DECL unsigned long long ts_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define C3M_STAMP(i) do { asm volatile("" ::: "memory"); ts_[i] = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); } while (0)
#define C3M_STAMPW(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); ts_[i] = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); } while (0)
#define C3M_TS_FLUSH() do { if (threadIdx.x == 0 && c3m_ts_buf) { unsigned long long* o_ = c3m_ts_buf + 8 * (size_t)(blockIdx.x + gridDim.x * blockIdx.z); \
    for (int i_ = 0; i_ < 8; ++i_) o_[i_] = ts_[i_]; } } while (0)
#else
#define C3M_TS_DECL
#define C3M_STAMP(i)
#define C3M_STAMPW(i)
#define C3M_TS_FLUSH()
#endif
// Program order of the tap loop is the schedule: the empty asm keeps the IR passes, the scheduling barrier the machine scheduler from
// moving memory operations across (left alone the compiler sinks every LDS read next to its first use: the wave then idles for the LDS
// latency twice per tap -- with one or two waves per SIMD, launches of 1-2 models, nothing else covers it; measured per tap and workgroup,
// one model: 1.63 k cycles against 1.02 k of MFMA issue, tools/c3m_timing.py).
// With three workgroups per CU (launches of >= 4 models) the other waves cover those latencies and the free schedule is the faster one
// (5 models: 107.5 vs 112.0 us per launch; 1 / 2 / 3 models: 36.2 / 46.5 / 68.4 free vs 29.6 / 44.1 / 65.0 us pinned): PIN is a template flag.
#define C3M_PIN() do { if constexpr (PIN) { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } } while (0)
#ifndef C3M_NTAPS
#define C3M_NTAPS 27          // timing-only ablation: fewer taps (tools/build_variant.sh)
#endif
template <int C3M_TM, bool PIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(C3M_TM == 64 ? 1 : 3, C3M_TM == 64 ? 2 : 3))) void conv3_fwd_mt_kernel(const Grp<Conv3FwdP> grp) {
    constexpr int RT = C3M_TM / 32, KS = 4 / RT, CW = 128 / KS, NQ = CW / 8, NH = NQ / 2;   // row tiles, channel splits, channels / float4 K-groups per wave, per half tap
    int gi, bx;
    C3M_TS_DECL;
    C3M_STAMP(0);
    xcd_place(gi, bx);           // XCD-contiguous row ranges; a fold group of 2 / 4 / 8 models: each model on its own XCDs
    const Conv3FwdP& p = grp.p[gi];
    // every kernel argument the loops need, read ONCE into registers (left as references into the kernarg segment the compiler re-loads them,
    // s_load + wait, inside each predicated load)
    const float* __restrict__ y1 = p.y1;
    const float* __restrict__ wp = p.wp;
    const int M = p.M, W = p.g.W, HW = p.g.H * p.g.W;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* img = smem;                                               // [nrows + 1][132]: the window and one row of zeros
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kq = lane >> 5;
    const int rt = wave / KS, kh2 = wave % KS;
    const int m0 = bx * C3M_TM;
    const int halo = W + 1, nrows = C3M_TM + 2 * halo;
    const int c4 = (tid & 31) * 4;
    float mean[4], sc[4], beta[4];
    bn_consts4(p.bn, c4, mean, sc, beta);
    const int myrow = m0 + rt * 32 + li;
    const unsigned m9 = myrow < M ? tap_mask9(p.coords[myrow < M ? myrow : 0], p.g, false) : 0u;
    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }

    // B operand: straight from global memory (L1/L2-resident packed weights) into the MFMA register layout, one tap ahead:
    // lane (kq, co = li) needs W[co][tap][CW*kh2 + 8q + 4kq .. +3], q = 0..NQ-1.  No LDS tile, hence no per-tap barrier: the window is
    // read-only during a kd phase and the waves drift freely.
    // (Measured and dropped: a tap-major lane order [tap][cin / 8][kq][cout][4], in which the 64 lanes of a load read one contiguous KB
    // instead of 32 bytes of 32 lines -- 43.0 vs 44.6 us per 2-model launch, 110.8 vs 113.3 at 5 models, 30.6 vs 30.1 at one: not worth a
    // third derived weight pack.)
    const float* wlane = wp + (size_t)li * (27 * 128) + CW * kh2 + 4 * kq;
    constexpr int wts = 128, wqs = 8;
    float4 b[2][NQ];
    auto bload = [&](float4 (&bb)[NQ], int tap) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) bb[q] = *(const float4*)(wlane + tap * wts + q * wqs);
    };
    // window rows: loaded into registers ahead of the kd phase that needs them (wload: branch-free, invalid rows read row 0 and are zeroed
    // when staged), transformed + stored at its start (wstore)
    constexpr int NI = ((C3M_TM + 2 * C3M_MAXHALO) * 32 + 255) / 256;      // 9 / 13
    float4 wv[NI];
    unsigned wok = 0;
    auto wload = [&](int kd) __attribute__((always_inline)) {
        const int base = m0 - halo + (kd - 1) * HW;
        wok = 0;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (8 * i < nrows) {                                       // workgroup-uniform
                const int s = (tid >> 5) + 8 * i, src = base + s;
                const bool ok = s < nrows && src >= 0 && src < M;
                wok |= (ok ? 1u : 0u) << i;
                wv[i] = *(const float4*)(y1 + (size_t)(ok ? src : 0) * 128 + c4);
            }
        }
    };
    auto wstore = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int s = (tid >> 5) + 8 * i;
            if (8 * i < nrows && s < nrows) {
                const float z = (wok >> i) & 1u ? 1.f : 0.f;
                *(float4*)&img[s * C3M_PITCH + c4] =
                    make_float4(z * fmaxf(bn_apply(wv[i].x, mean[0], sc[0], beta[0]), 0.f), z * fmaxf(bn_apply(wv[i].y, mean[1], sc[1], beta[1]), 0.f),
                                z * fmaxf(bn_apply(wv[i].z, mean[2], sc[2], beta[2]), 0.f), z * fmaxf(bn_apply(wv[i].w, mean[3], sc[3], beta[3]), 0.f));
            }
        }
    };
    // A operand of a tap: the window slot shifted by (kh - 1, kw - 1) -- or, for a row the tap's zero padding excludes, the row of zeros
    // (one address select per tap instead of 4 NQ multiplies by a 0/1 factor)
    const float* arow = img + (rt * 32 + li + halo) * C3M_PITCH + CW * kh2 + 4 * kq;
    const float* azero = img + nrows * C3M_PITCH + CW * kh2 + 4 * kq;
    float4 aL[NH], aH[NH];
    auto aread = [&](float4 (&a)[NH], int tap, int half) __attribute__((always_inline)) {
        const int kd = tap / 9, t9 = tap - 9 * kd, kh = t9 / 3, kw = t9 - 3 * kh;
        const unsigned sel = (1u << kd) | (8u << kh) | (64u << kw);
        const float* ar = (m9 & sel) == sel ? arow + ((kh - 1) * W + (kw - 1)) * C3M_PITCH : azero;
#pragma unroll
        for (int q = 0; q < NH; ++q) a[q] = *(const float4*)(ar + 8 * (half * NH + q));
    };
    auto mma = [&](const float4 (&a)[NH], const float4 (&bb)[NQ], int half) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NH; ++q) {        // two accumulators: consecutive MFMAs never wait for each other's result
            const float4& bq = bb[half * NH + q];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, bq.x, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, bq.y, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, bq.z, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, bq.w, acc2, 0, 0, 0);
        }
    };
    bload(b[0], 0);
    wload(0);
    if (tid < C3M_PITCH) img[nrows * C3M_PITCH + tid] = 0.f;
    C3M_STAMP(1);                 // prologue loads issued (constants, mask, first window, first tap's weights)
    wstore();
    __syncthreads();
    C3M_STAMP(2);                 // first window staged
    aread(aL, 0, 0);
    // Half-tap software pipeline over the same 4 NQ A registers: the LDS reads of half h + 1 are issued BEFORE the MFMAs of half h.
    static_for<C3M_NTAPS>([&](auto T) __attribute__((always_inline)) {
        constexpr int tap = decltype(T)::value, t9 = tap % 9;
#ifdef C3M_TIMING
        if constexpr (tap == 10) C3M_STAMP(3);      // kd = 0 done, second window staged, tap 9 done
        if constexpr (tap == 18) C3M_STAMP(4);      // kd = 1 done
#endif
        if constexpr (tap + 1 < C3M_NTAPS) bload(b[(tap + 1) & 1], tap + 1);     // next tap's weights: in flight during this tap
        if constexpr (t9 == 5 && tap + 4 < C3M_NTAPS) wload(tap / 9 + 1);        // next kd's rows: in flight during taps 5..8
        aread(aH, tap, 1);
        C3M_PIN();
        mma(aL, b[tap & 1], 0);
        C3M_PIN();
        if constexpr (t9 != 8 && tap + 1 < C3M_NTAPS) aread(aL, tap + 1, 0);
        C3M_PIN();
        mma(aH, b[tap & 1], 1);
        C3M_PIN();
        if constexpr (t9 == 8 && tap + 1 < C3M_NTAPS) {                          // all waves are done with this window
            __syncthreads();
            wstore();
            __syncthreads();
            aread(aL, tap + 1, 0);
        }
    });
    C3M_STAMP(5);                 // 27 taps
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
    __syncthreads();                                                 // window -> Cs alias
    // ---- epilogue: every channel split's partial tile into its own LDS region, summed in split order; 32 slab columns, batch statistics
    float* Cs = smem;                                                // [KS][C3M_TM][33], aliases the window (<= C3M_TM x 132 floats)
    {
        float* mine = Cs + kh2 * (C3M_TM * 33) + (rt * 32 + 4 * kq) * 33 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[((r & 3) + 8 * (r >> 2)) * 33] = acc[r];
    }
    __syncthreads();
    for (int idx = tid; idx < C3M_TM * 33; idx += 256) {
        float v = Cs[idx];
#pragma unroll
        for (int h = 1; h < KS; ++h) v += Cs[idx + h * (C3M_TM * 33)];
        Cs[idx] = v;
    }
    __syncthreads();
    C3M_STAMP(6);                 // channel splits added
    store_tile<C3M_TM, 32>(p.out, p.ldo, M, 32, m0, 0, Cs, tid);
    tile_col_stats<C3M_TM, 32>(stat_rep(p.osum, p.srep, p.sstride), stat_rep(p.osumsq, p.srep, p.sstride), M, 32, m0, 0, Cs, tid);
    C3M_STAMPW(7);                // tile written, statistic atomics acknowledged
    C3M_TS_FLUSH();
}
// Tile height by how well the launch fills whole rounds of the chip: 32-row tiles run 3 workgroups per CU (768 per round), 64-row tiles
// 2 per CU (512 per round).  Measured on block 1 (8192 rows per model), G models per launch, us per launch, per-tap GEMM form / 32 / 64:
// G=1 36/31/48, G=2 51/44/50, G=3 77/57/73, G=4 105/85/80, G=5 116/90/112, G=8 188/145/145, G=10 223/172/180 -- the 32-row form unless the
// 64-row tiles fill their rounds clearly better (G = 4).  Below MmsDnOpts.conv3_mt32_min (256) 32-row tiles: the per-tap GEMM form (Conv3FwdOp).
// MmsDnOpts.conv3_mt: -1 = never, 2 = 64-row form whenever it applies, 3 = 32-row form whenever it applies (tests); returns the tile height or 0.
static inline int conv3_mt_tile(int M, int ng, const Dims3& g, const MmsDnOpts& o) {
    if (g.W + 1 > C3M_MAXHALO || o.conv3_mt < 0) return 0;
    if (o.conv3_mt == 2) return M >= 1024 ? 64 : 0;
    if (o.conv3_mt == 3) return M >= 64 ? 32 : 0;
    const int min32 = o.conv3_mt32_min > 0 ? o.conv3_mt32_min : 256;
    const long n32 = (long)((M + 31) / 32) * ng, n64 = (long)((M + 63) / 64) * ng;
    if (M < 1024 || n32 < min32) return 0;
    const double f32 = (double)n32 / (double)((n32 + 767) / 768 * 768), f64 = (double)n64 / (double)((n64 + 511) / 512 * 512);
    return n64 >= 512 && f64 > f32 + 0.1 ? 64 : 32;
}
template <int TM, bool PIN>
static int launch_conv3_fwd_mt(const Conv3FwdP* pp, int ng, hipStream_t s) {
    const Conv3FwdP& p = *pp;
    constexpr int smem_max = (TM + 2 * C3M_MAXHALO + 1) * C3M_PITCH * (int)sizeof(float);
    constexpr int smem_epi = 4 / (TM / 32) * TM * 33 * (int)sizeof(float);                // the epilogue's partial tiles: 16.9 KB
    const int smem_win = (TM + 2 * (p.g.W + 1) + 1) * C3M_PITCH * (int)sizeof(float);  // window + the row of zeros; W = 8: 43.8 KB (64 rows) / 26.9 KB (32 rows)
    const int smem = smem_win > smem_epi ? smem_win : smem_epi;
    static std::once_flag attr_once;
    std::call_once(attr_once, [&] { hipFuncSetAttribute((const void*)(conv3_fwd_mt_kernel<TM, PIN>), hipFuncAttributeMaxDynamicSharedMemorySize, smem_max); });
    Grp<Conv3FwdP> a;
    grp_fill(a, pp, ng, 1);
    MMS_LAUNCH((conv3_fwd_mt_kernel<TM, PIN>), dim3((p.M + TM - 1) / TM, 1, ng), dim3(256), smem, s, a);
    return mms_check_launch();
}

extern "C" int mms_conv3_fwd_group(const Conv3FwdP* pp, int ng, const MmsDnOpts* opts, hipStream_t s) {
    if (!pp || ng < 1 || ng > MMS_MAX_GROUP) return MMS_ERR_ARG;
    const MmsDnOpts o = mms_opts(opts);
    const Conv3FwdP& p = *pp;
    if (p.M <= 0 || p.ldo % 4 != 0) return MMS_ERR_ARG;
    for (int g = 1; g < ng; ++g) {
        const Conv3FwdP& q = pp[g];
        if (q.M != p.M || q.ldo % 4 != 0 || q.g.D != p.g.D || q.g.H != p.g.H || q.g.W != p.g.W || (q.partial == nullptr) != (p.partial == nullptr) ||
            q.nsplit != p.nsplit) return MMS_ERR_ARG;
    }
    for (int g = 0; g < ng; ++g) if (!mms_bn_aligned16(pp[g].bn) || pp[g].wfrag != p.wfrag) return MMS_ERR_ARG;   // BatchNorm blocks are read with 16-byte vector loads
    if (p.wfrag) return (!p.partial && mms_conv3_small_jn(p.M, ng, p.g, o)) ? mms_c3s_fwd(pp, ng, o, s) : MMS_ERR_ARG;       // fragment-ordered weights: the small-grid kernel only
    if (const int tm = p.partial ? 0 : conv3_mt_tile(p.M, ng, p.g, o)) {
        if (tm == 64) return launch_conv3_fwd_mt<64, false>(pp, ng, s);
        // pinned schedule while the launch is at most one round of three 32-row workgroups per CU
        return (long)((p.M + 31) / 32) * ng <= 768 ? launch_conv3_fwd_mt<32, true>(pp, ng, s) : launch_conv3_fwd_mt<32, false>(pp, ng, s);
    }
    if (!p.partial && mms_conv3_small_jn(p.M, ng, p.g, o)) return mms_c3s_fwd(pp, ng, o, s);      // small grids: 16-row tiles, all taps, no reduce launch
    if (p.partial) {
        if (p.nsplit < 1 || p.nsplit > 27) return MMS_ERR_ARG;
        const int tpw = (27 + p.nsplit - 1) / p.nsplit;
        if ((p.nsplit - 1) * tpw >= 27) return MMS_ERR_ARG;       // every workgroup must own at least one tap
        int rc = launch_tile_gemm<Conv3FwdOp<true>>(pp, ng, dim3((p.M + 31) / 32, 1, p.nsplit), s);
        if (rc != MMS_OK) return rc;
        Grp<Conv3FwdP> a;
        grp_fill(a, pp, ng, 1);
        MMS_LAUNCH(conv3_fwd_reduce_kernel, dim3((p.M + 7) / 8, 1, ng), dim3(256), 0, s, a);
        return mms_check_launch();
    }
    return launch_tile_gemm<Conv3FwdOp<false>>(pp, ng, dim3((p.M + 31) / 32, 1, 1), s);
}
MMS_SINGLE_O(mms_conv3_fwd, Conv3FwdP)

// ------------------------------------------------------------------------------------------------------
// conv0: Conv3d(1, 64, k7, s2, p3) as implicit GEMM, K = 343 taps
// ------------------------------------------------------------------------------------------------------
struct Conv0FwdOp {
    typedef Conv0FwdP Params;
    static constexpr int WM = 2, WN = 2, WK = 1, AMODE = LD_K1, BMODE = LD_K1;
    __device__ void step(const Params&, int) {}
    static constexpr int TM = 64, TN = 64;
    static constexpr int EXTRA = 4 * 64;
    const int* info;   // per tile row: sample offset, id0, ih0, iw0
    int m0;
    __device__ void setup(const Params& p, int m0_, int, int, float* extra, int tid) {
        info = (const int*)extra; m0 = m0_;
        if (tid < 64) {
            int* o = (int*)extra + 4 * tid;
            int m = m0 + tid;
            if (m < p.M) {
                int od, oh, ow;
                unpack_dhw(p.coords[m], od, oh, ow);
                int b = m / (p.out.D * p.out.H * p.out.W);
                o[0] = b * p.in.D * p.in.H * p.in.W; o[1] = 2 * od - 3; o[2] = 2 * oh - 3; o[3] = 2 * ow - 3;
            } else {
                o[0] = -1; o[1] = o[2] = o[3] = 0;
            }
        }
    }
    __device__ void krange(const Params&, int, int& kb, int& ke) { kb = 0; ke = 343; }
    typedef float ARaw;
    typedef float BRaw;
    __device__ float a_ld(const Params& p, int, int m, int k, bool& ok) const {
        ok = true;
        if (k >= 343) return 0.f;
        const int* o = info + 4 * (m - m0);
        if (o[0] < 0) return 0.f;
        const int kd = k / 49, kh = (k / 7) % 7, kw = k % 7;
        const int id = o[1] + kd, ih = o[2] + kh, iw = o[3] + kw;
        if ((unsigned)id >= (unsigned)p.in.D || (unsigned)ih >= (unsigned)p.in.H || (unsigned)iw >= (unsigned)p.in.W)
            return 0.f;
        return p.x[(size_t)o[0] + ((size_t)id * p.in.H + ih) * p.in.W + iw];
    }
    __device__ float a_tx(const Params&, int, float v, int, int, bool) const { return v; }
    __device__ float b_ld(const Params& p, int, int n, int k, bool& ok) const { ok = true; return k < 343 ? p.w[n * 343 + k] : 0.f; }
    __device__ float b_tx(const Params&, int, float v, int, int, bool) const { return v; }
    __device__ void epilogue(const Params& p, int m0_, int n0, int, const float* Cs, int tid, bool active) {
        if (!active) return;
        store_tile<TM, TN>(p.y, 64, p.M, 64, m0_, n0, Cs, tid);
        tile_col_stats<TM, TN>(stat_rep(p.osum, p.srep, p.sstride), stat_rep(p.osumsq, p.srep, p.sstride), p.M, 64, m0_, n0, Cs, tid);
    }
};

// ------------------------------------------------------------------------------------------------------
// conv0, LDS-staged form (default; the tile-GEMM op above remains for grids that are not a multiple of 4x4x4).
// The GEMM form gathers x[patch(m, tap)] with one predicated scalar load per (voxel, tap).  Here a workgroup walks boxes
// of 4x4x4 output voxels: the box's 13^3 input region is staged in LDS once (double-buffered) and the A operand of every
// MFMA is read straight from it -- A[voxel][tap] = region[moff(voxel) + koff(tap)], one ds_read_b32, no im2col image.
// The 343 x 32 weight slice of a wave's channel tile lives in 172 VGPRs for the whole launch (B operand, loaded once per
// model), so the steady state is: 9 global loads per thread and box, then 172 MFMAs per wave fed by LDS only.
// 4 waves = 2 voxel tiles (d-slices {0,1} / {2,3} of the box) x 2 channel tiles.  Boxes of all models are pooled over
// gridDim.x workgroups (grp.zdim = number of models) as in conv0_bwd_weight_kernel.
// ------------------------------------------------------------------------------------------------------
#define C0F_REG (13 * 13 * 13)      // 2197
__host__ __device__ constexpr int c0f_koff(int tap) { return (tap / 49) * 169 + ((tap / 7) % 7) * 13 + tap % 7; }
__global__ __launch_bounds__(256) void conv0_fwd_box_kernel(const Grp<Conv0FwdP> grp) {
    __shared__ float xs[2][C0F_REG + 3];
    __shared__ float Cs[4][32 * 33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kq = lane >> 5;
    const int vt = wave >> 1, ct = wave & 1;                       // voxel tile, channel tile
    const int D0 = grp.p[0].out.D, H0 = grp.p[0].out.H, W0 = grp.p[0].out.W, Di = grp.p[0].in.D, Hi = grp.p[0].in.H, Wi = grp.p[0].in.W;
    const int bh = H0 >> 2, bw = W0 >> 2, bps = (D0 >> 2) * bh * bw;
    const int nbox = (grp.p[0].M / (D0 * H0 * W0)) * bps;
    const int total = nbox * grp.zdim;
    const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int g0 = blockIdx.x * per, g1 = g0 + per < total ? g0 + per : total;
    const int dl = 2 * vt + (li >> 4), hl = (li >> 2) & 3, wl = li & 3;
    const int moff = dl * 2 * 169 + hl * 2 * 13 + wl * 2;
    float* cs = Cs[wave];
    __syncthreads();

    for (int gs = g0; gs < g1;) {
        const int model = gs / nbox, seg_end = (model + 1) * nbox < g1 ? (model + 1) * nbox : g1;
        const int b0 = gs - model * nbox, b1 = seg_end - model * nbox;
        gs = seg_end;
        const Conv0FwdP& p = grp.p[model];
        float breg[172];                                       // W[32*ct + li][2s + kq], s = 0..171 (tap 343 = padding)
        {
            // through LDS: the 88 KB weight block is copied with coalesced loads, all in flight together, and each lane picks its row from
            // there.  (Straight from global memory a lane's row sits 1372 B from its neighbour's: 172 load instructions of 32 lines each,
            // ~15 us of every workgroup's life -- a third of a one-model launch of four boxes per workgroup.)
            extern __shared__ __attribute__((aligned(16))) float wst[];      // [64][343]
            __syncthreads();                                   // previous segment's rows are in registers
            constexpr int NW = 64 * 343, NPASS = (NW + 255) / 256;           // 86 words per thread
            float t[NPASS];
#pragma unroll
            for (int i = 0; i < NPASS; ++i) { const int e = tid + 256 * i; t[i] = p.w[e < NW ? e : NW - 1]; }
#pragma unroll
            for (int i = 0; i < NPASS; ++i) { const int e = tid + 256 * i; if (e < NW) wst[e] = t[i]; }
            __syncthreads();
            const float* wrow = wst + (32 * ct + li) * 343;
#pragma unroll
            for (int s = 0; s < 172; ++s) breg[s] = (2 * s + kq < 343) ? wrow[2 * s + kq] : 0.f;
        }
        double ssum = 0, ssq = 0;
        float xr[9];
        auto gload = [&](int bx) {
            const int b = bx / bps, r = bx - b * bps, bz = r / (bh * bw), r2 = r - bz * (bh * bw), by = r2 / bw, bxw = r2 - by * bw;
            const int id0 = 8 * bz - 3, ih0 = 8 * by - 3, iw0 = 8 * bxw - 3;
            const float* xb = p.x + (size_t)b * Di * Hi * Wi;
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const int e = tid + 256 * i, rd = e / 169, e2 = e - rd * 169, rh = e2 / 13, rw = e2 - rh * 13;
                const int id = id0 + rd, ih = ih0 + rh, iw = iw0 + rw;
                const bool ok = e < C0F_REG && (unsigned)id < (unsigned)Di && (unsigned)ih < (unsigned)Hi && (unsigned)iw < (unsigned)Wi;
                xr[i] = ok ? xb[((size_t)id * Hi + ih) * Wi + iw] : 0.f;
            }
        };
        auto sstore = [&](int buf) {
#pragma unroll
            for (int i = 0; i < 9; ++i) { const int e = tid + 256 * i; if (e < C0F_REG) xs[buf][e] = xr[i]; }
        };
        __syncthreads();                                       // previous segment done with xs
        gload(b0);
        sstore(0);
        __syncthreads();
        int buf = 0;
        for (int bx = b0; bx < b1; ++bx) {
            const bool more = bx + 1 < b1;
            if (more) gload(bx + 1);
            // tap offsets are compile-time constants of the unrolled step: the two lane halves differ by koff(2s+1) - koff(2s)
            // (1, 7 or 85), so the address is one of three per-lane bases plus an immediate -- no index arithmetic per MFMA.
            // Two accumulators alternate so that consecutive MFMAs do not wait on each other's result.
            const float* xb = xs[buf] + moff;
            f32x16 acc, acc2;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
            // The A operand of MFMA s is requested C0F_PF MFMAs ahead and the order is pinned: left alone the compiler puts each ds_read right
            // before its two MFMAs and the wave -- the only one on its SIMD: 343 VGPRs -- idles for the LDS latency once per pair.
            constexpr int C0F_PF = 8;
            float ring[C0F_PF];
            auto aread = [&](int s_) __attribute__((always_inline)) {
                const int k0 = c0f_koff(2 * s_), k1 = c0f_koff(2 * s_ + 1 < 343 ? 2 * s_ + 1 : 342);
                return xb[k0 + kq * (k1 - k0)];
            };
#pragma unroll
            for (int s_ = 0; s_ < C0F_PF; ++s_) ring[s_] = aread(s_);
            static_for<172>([&](auto S) __attribute__((always_inline)) {
                constexpr int s_ = decltype(S)::value;
                const float a = ring[s_ % C0F_PF];
                if constexpr (s_ + C0F_PF < 172) ring[s_ % C0F_PF] = aread(s_ + C0F_PF);
                asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
                if constexpr ((s_ & 1) == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, breg[s_], acc, 0, 0, 0);
                else acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, breg[s_], acc2, 0, 0, 0);
                asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
            });
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
            // statistics (channel 32*ct + li, this lane's 16 voxel rows) and the store through a per-wave LDS transpose
            float fs = 0.f, fq = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { fs += acc[r]; fq = fmaf(acc[r], acc[r], fq); }
            ssum += fs; ssq += fq;
#pragma unroll
            for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * kq) * 33 + li] = acc[r];
            __builtin_amdgcn_wave_barrier();
            {
                const int b = bx / bps, r = bx - b * bps, bz = r / (bh * bw), r2 = r - bz * (bh * bw), by = r2 / bw, bxw = r2 - by * bw;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int v = 2 * i + kq;                  // voxel of this wave's tile
                    const int od = 4 * bz + 2 * vt + (v >> 4), oh = 4 * by + ((v >> 2) & 3), ow = 4 * bxw + (v & 3);
                    const size_t m = ((size_t)(b * D0 + od) * H0 + oh) * W0 + ow;
                    p.y[m * 64 + 32 * ct + li] = cs[v * 33 + li];
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (more) sstore(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
        if (p.osum) {
            ssum += __shfl_xor(ssum, 32, 64); ssq += __shfl_xor(ssq, 32, 64);
            if (kq == 0) {
                atomicAdd(&stat_rep(p.osum, p.srep, p.sstride)[32 * ct + li], ssum);
                atomicAdd(&stat_rep(p.osumsq, p.srep, p.sstride)[32 * ct + li], ssq);
            }
        }
    }
}

extern "C" int mms_conv0_fwd_group(const Conv0FwdP* pp, int ng, const MmsDnOpts* opts, hipStream_t s) {
    if (!pp || ng < 1 || ng > MMS_MAX_GROUP) return MMS_ERR_ARG;
    const Conv0FwdP& p = *pp;
    if (p.M <= 0) return MMS_ERR_ARG;
    for (int g = 1; g < ng; ++g) {
        const Conv0FwdP& q = pp[g];
        if (q.M != p.M || q.in.D != p.in.D || q.in.H != p.in.H || q.in.W != p.in.W || q.out.D != p.out.D || q.out.H != p.out.H ||
            q.out.W != p.out.W || (q.osum == nullptr) != (p.osum == nullptr)) return MMS_ERR_ARG;
    }
    const long vox = (long)p.out.D * p.out.H * p.out.W;
    const bool boxed = (p.out.D % 4 == 0) && (p.out.H % 4 == 0) && (p.out.W % 4 == 0) && vox > 0 && p.M % vox == 0 &&
                       p.in.D == 2 * p.out.D && p.in.H == 2 * p.out.H && p.in.W == 2 * p.out.W;
    if (!boxed) return launch_tile_gemm<Conv0FwdOp>(pp, ng, dim3((p.M + 63) / 64, 1, 1), s);
    Grp<Conv0FwdP> a;
    grp_fill(a, pp, ng, ng);
    // one workgroup per CU at most: 342 VGPRs leave room for one wave per SIMD, so a second round would only pay the per-workgroup set-up
    // (weight rows into registers) again
    const long boxes = (long)ng * (p.M / 64);
    int nwg = (opts && opts->c0f_nwg > 0) ? opts->c0f_nwg : (boxes >= 256 ? 256 : (int)boxes);
    if (nwg < 1) nwg = 1;
    constexpr int smem = 64 * 343 * (int)sizeof(float);            // 87.8 KB of dynamic LDS (the weight block) + 34.5 KB static
    static std::once_flag attr_once;
    std::call_once(attr_once, [&] { hipFuncSetAttribute((const void*)conv0_fwd_box_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem); });
    MMS_LAUNCH(conv0_fwd_box_kernel, dim3(nwg, 1, 1), dim3(256), smem, s, a);
    return mms_check_launch();
}
MMS_SINGLE_O(mms_conv0_fwd, Conv0FwdP)

// ------------------------------------------------------------------------------------------------------
// bn0 + relu + maxpool(3,2,1): one workgroup = 32 pooled voxels x 64 channels
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_fwd_kernel(const Grp<PoolFwdP> grp) {
    const PoolFwdP& p = grp.p[blockIdx.z];
    // workgroup = 16 pooled voxels x 64 channels (4 voxel rows x 4 iterations).  The 27 window loads of a voxel are
    // issued together from clamped (always valid) addresses and selected afterwards: branch-free, 27 loads in flight.
    __shared__ double red[2][4][64];
    const int c = threadIdx.x & 63, vr = threadIdx.x >> 6;
    float mu, rstd, ga_, be;
    bn_consts1(p.bn, c, mu, rstd, ga_, be);
    const float sc = ga_ * rstd;
    const int vox_out = p.out.D * p.out.H * p.out.W, Mout = p.B * vox_out;
    double s = 0, q = 0;
    for (int it = 0; it < 4; ++it) {
        const int m = blockIdx.x * 16 + it * 4 + vr;
        if (m >= Mout) break;
        const int b = m / vox_out, r = m % vox_out;
        const int od = r / (p.out.H * p.out.W), oh = (r / p.out.W) % p.out.H, ow = r % p.out.W;
        const float* base = p.y0 + (size_t)b * p.in.D * p.in.H * p.in.W * 64 + c;
        float v[27];
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const int id = 2 * od - 1 + t / 9, ih = 2 * oh - 1 + (t / 3) % 3, iw = 2 * ow - 1 + t % 3;
            const int cd = min(max(id, 0), p.in.D - 1), ch = min(max(ih, 0), p.in.H - 1), cw = min(max(iw, 0), p.in.W - 1);
            v[t] = base[((size_t)(cd * p.in.H + ch) * p.in.W + cw) * 64];
        }
        float best = -INFINITY;
        int bi = 0;
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const int id = 2 * od - 1 + t / 9, ih = 2 * oh - 1 + (t / 3) % 3, iw = 2 * ow - 1 + t % 3;
            const bool ok = (unsigned)id < (unsigned)p.in.D && (unsigned)ih < (unsigned)p.in.H && (unsigned)iw < (unsigned)p.in.W;
            const float a = ok ? fmaxf(bn_apply(v[t], mu, sc, be), 0.f) : -INFINITY;
            if (a > best) { best = a; bi = t; }   // strict >: first maximum in scan order wins (torch)
        }
        p.slab[(size_t)m * p.ld + c] = best;
        if (p.argmax) p.argmax[(size_t)m * 64 + c] = (uint8_t)bi;
        s += best; q += (double)best * best;
    }
    if (p.osum) {
        red[0][vr][c] = s; red[1][vr][c] = q;
        __syncthreads();
        if (vr == 0) {
            atomicAdd(&stat_rep(p.osum, p.srep, p.sstride)[c], red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
            atomicAdd(&stat_rep(p.osumsq, p.srep, p.sstride)[c], red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
        }
    }
}

extern "C" int mms_pool_fwd_group(const PoolFwdP* pp, int ng, hipStream_t s) {
    Grp<PoolFwdP> a;
    if (!grp_fill(a, pp, ng, 1)) return MMS_ERR_ARG;
    const PoolFwdP& p = *pp;
    int Mout = p.B * p.out.D * p.out.H * p.out.W;
    if (Mout <= 0) return MMS_ERR_ARG;
    for (int g = 1; g < ng; ++g) {
        const PoolFwdP& q = pp[g];
        if (q.B != p.B || q.in.D != p.in.D || q.in.H != p.in.H || q.in.W != p.in.W || q.out.D != p.out.D || q.out.H != p.out.H ||
            q.out.W != p.out.W) return MMS_ERR_ARG;
    }
    MMS_LAUNCH(pool_fwd_kernel, dim3((Mout + 15) / 16, 1, ng), dim3(256), 0, s, a);
    return mms_check_launch();
}
MMS_SINGLE(mms_pool_fwd, PoolFwdP)

// ------------------------------------------------------------------------------------------------------
// head: norm5 + relu + global average pool + Linear(C, N)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_fwd_kernel(const Grp<HeadFwdP> grp) {
    const HeadFwdP& p = grp.p[blockIdx.z];
    extern __shared__ float pooled[];   // [B][C]
    const int tid = threadIdx.x;
    // pooled[b][c] = mean over the sample's V voxels of relu(norm5(x)).  A thread owns 4 channels (tid + 256 j) of a 1024-channel chunk:
    // their BatchNorm constants are requested together, then the rows in batches of 8 -- 1 + rows / 8 memory round trips per chunk
    // (one (b, c) element per trip was B * C / 256 x (constants + V rows) serial round trips: 31 us per launch).
    const float invV = 1.f / (float)p.V;
    const int rows = p.B * p.V;
    for (int c0 = 0; c0 < p.C; c0 += 1024) {
        float mu[4], sc[4], be[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + tid + 256 * j, cc = c < p.C ? c : p.C - 1;
            float rstd, ga_;
            bn_consts1(p.bn, cc, mu[j], rstd, ga_, be[j]);
            sc[j] = ga_ * rstd;
        }
        float accb[4] = {0.f, 0.f, 0.f, 0.f};          // running sum of the current sample, per owned channel
        for (int r0 = 0; r0 < rows; r0 += 8) {
            float x[8][4];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = c0 + tid + 256 * j, r = r0 + i;
                    x[i][j] = (r < rows && c < p.C) ? p.slab[(size_t)r * p.ld + c] : 0.f;
                }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = r0 + i;
                if (r < rows) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) accb[j] += fmaxf(bn_apply(x[i][j], mu[j], sc[j], be[j]), 0.f);
                    if ((r + 1) % p.V == 0) {          // the sample's last voxel: its means are complete
                        const int b = r / p.V;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int c = c0 + tid + 256 * j;
                            if (c < p.C) {
                                const float a = accb[j] * invV;
                                pooled[b * p.C + c] = a;
                                if (blockIdx.x == 0 && p.pooled) p.pooled[b * p.C + c] = a;
                            }
                            accb[j] = 0.f;
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    const int n = blockIdx.x * 4 + (tid >> 6), lane = tid & 63;
    if (n >= p.N) return;
    const float bias = p.bias[n];
    for (int b0 = 0; b0 < p.B; b0 += 4) {          // the weight row is read once per 4 samples
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int c = lane; c < p.C; c += 64) {
            const float wv = p.w[(size_t)n * p.C + c];
#pragma unroll
            for (int i = 0; i < 4; ++i) if (b0 + i < p.B) a[i] = fmaf(wv, pooled[(b0 + i) * p.C + c], a[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = wave_sum(a[i]);
            if (lane == 0 && b0 + i < p.B) p.out[(b0 + i) * p.ldo + n] = v + bias;
        }
    }
}

extern "C" int mms_head_fwd_group(const HeadFwdP* pp, int ng, hipStream_t s) {
    Grp<HeadFwdP> a;
    if (!grp_fill(a, pp, ng, 1)) return MMS_ERR_ARG;
    const HeadFwdP& p = *pp;
    size_t smem = (size_t)p.B * p.C * sizeof(float);
    if (smem > 64 * 1024 || p.B <= 0) return MMS_ERR_ARG;
    for (int g = 1; g < ng; ++g) {
        const HeadFwdP& q = pp[g];
        if (q.B != p.B || q.C != p.C || q.N != p.N || q.V != p.V) return MMS_ERR_ARG;
    }
    MMS_LAUNCH(head_fwd_kernel, dim3((p.N + 3) / 4, 1, ng), dim3(256), smem, s, a);
    return mms_check_launch();
}
MMS_SINGLE(mms_head_fwd, HeadFwdP)

// ------------------------------------------------------------------------------------------------------
// helpers: voxel coordinate tables, conv2 weight packing, BN running-stat update
// ------------------------------------------------------------------------------------------------------
__global__ void init_coords_kernel(int* coords, int M, Dims3 g) {
    int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    int r = m % (g.D * g.H * g.W);
    coords[m] = pack_dhw(r / (g.H * g.W), (r / g.W) % g.H, r % g.W);
}
extern "C" int mms_init_coords(int* coords, int B, int D, int H, int W, hipStream_t s) {
    if (D > 1023 || H > 1023 || W > 1023) return MMS_ERR_ARG;
    int M = B * D * H * W;
    Dims3 g{D, H, W};
    MMS_LAUNCH(init_coords_kernel, dim3((M + 255) / 256), dim3(256), 0, s, coords, M, g);
    return mms_check_launch();
}

// canonical torch conv2 weight [32][128][27] -> fwd pack [32][27][128] and bwd-data pack [128][27][32]
__global__ void pack_conv3_kernel(const float* __restrict__ w, float* __restrict__ wpf, float* __restrict__ wpb) {
    int idx = blockIdx.x * 256 + threadIdx.x;      // over [cout][tap][cin]
    if (idx >= 32 * 27 * 128) return;
    int cin = idx & 127, tap = (idx >> 7) % 27, cout = idx / (27 * 128);
    float v = w[(cout * 128 + cin) * 27 + tap];
    wpf[idx] = v;
    wpb[(cin * 27 + tap) * 32 + cout] = v;
}
extern "C" int mms_pack_conv3(const float* w, float* wpf, float* wpb, hipStream_t s) {
    MMS_LAUNCH(pack_conv3_kernel, dim3(32 * 27 * 128 / 256), dim3(256), 0, s, w, wpf, wpb);
    return mms_check_launch();
}

// single layer, MFMA-fragment orders (tests / tools; the step uses the table kernel below)
__global__ void pack_conv3_frag_kernel(const float* __restrict__ w, float* __restrict__ wff, float* __restrict__ wfb) {
    const int idx = blockIdx.x * 256 + threadIdx.x;      // over the canonical [cout][cin][tap]
    if (idx >= 32 * 128 * 27) return;
    const int tap = idx % 27, cin = (idx / 27) & 127, co = idx / (27 * 128);
    const float v = w[idx];
    wff[(((((size_t)tap * 4 + (cin >> 5)) * 2 + ((cin >> 4) & 1)) * 2 + (co >> 4)) * 64 + ((cin >> 2) & 3) * 16 + (co & 15)) * 4 + (cin & 3)] = v;
    wfb[((((size_t)tap * 8 + (cin >> 4)) * 2 + (co >> 4)) * 64 + ((co >> 2) & 3) * 16 + (cin & 15)) * 4 + (co & 3)] = v;
}
extern "C" int mms_pack_conv3_frag(const float* w, float* wff, float* wfb, hipStream_t s) {
    if (!w || !wff || !wfb) return MMS_ERR_ARG;
    MMS_LAUNCH(pack_conv3_frag_kernel, dim3(32 * 27 * 128 / 256), dim3(256), 0, s, w, wff, wfb);
    return mms_check_launch();
}

// batched variant over device tables of layer pointers (one launch per forward, all models of the fold group).
// One workgroup = (model, layer, 32 input channels): the [32 co][32 cin][27] slice is staged in LDS (odd strides: every
// phase is bank-conflict free) so that the canonical reads (1728-B runs), the backward pack (the slice is one contiguous
// 110 KB run of wpb) and the forward pack (128-B runs = whole cache lines) are all coalesced.
struct TabPtrs { const void* t[MMS_MAX_GROUP]; };
#define PACK_CO_STRIDE 865      // 32 * 27 + 1
// fragmask bit l: layer l's packs in MFMA-fragment order (consumed by the small-grid kernels of dn_c3s.hip, whose weight loads are then one
// contiguous 1 KB per wave instruction instead of 64 separate 16-byte pieces):
//   forward  [tap][cin/32][(cin/16)%2][co/16][lane = ((cin/4)%4)*16 + co%16][cin%4]
//   backward [tap][cin/16][co/16][lane = ((co/4)%4)*16 + cin%16][co%4]
__global__ __launch_bounds__(256) void pack_conv3_table_kernel(const TabPtrs tabs, uint64_t fragmask) {
    extern __shared__ float t[];         // [32 co][PACK_CO_STRIDE]: element (co, cin_l, tap) at co * 865 + cin_l * 27 + tap
    const PackEntry e = ((const PackEntry*)tabs.t[blockIdx.z])[blockIdx.y];
    const int cin0 = blockIdx.x * 32;
    for (int idx = threadIdx.x; idx < 32 * 864; idx += 256) {
        const int co = idx / 864, r = idx - co * 864;
        t[co * PACK_CO_STRIDE + r] = e.w[((size_t)co * 128 + cin0) * 27 + r];
    }
    __syncthreads();
    if ((fragmask >> blockIdx.y) & 1ull) {
        for (int idx = threadIdx.x; idx < 32 * 864; idx += 256) {        // backward: this slice = cin tiles 2x, 2x+1 -> 4 KB runs per tap
            const int el = idx & 3, lane = (idx >> 2) & 63, q = (idx >> 8) & 1, ntl = (idx >> 9) & 1, tap = idx >> 10;
            const int cin_l = 16 * ntl + (lane & 15), co = 16 * q + 4 * (lane >> 4) + el;
            e.wpb[((((size_t)tap * 8 + 2 * blockIdx.x + ntl) * 2 + q) * 64 + lane) * 4 + el] = t[co * PACK_CO_STRIDE + cin_l * 27 + tap];
        }
        for (int idx = threadIdx.x; idx < 32 * 864; idx += 256) {        // forward: this slice = channel quarter x -> 4 KB runs per tap
            const int el = idx & 3, lane = (idx >> 2) & 63, j = (idx >> 8) & 1, q = (idx >> 9) & 1, tap = idx >> 10;
            const int co = 16 * j + (lane & 15), cin_l = 16 * q + 4 * (lane >> 4) + el;
            e.wpf[(((((size_t)tap * 4 + blockIdx.x) * 2 + q) * 2 + j) * 64 + lane) * 4 + el] = t[co * PACK_CO_STRIDE + cin_l * 27 + tap];
        }
        return;
    }
    float* wpb = e.wpb + (size_t)cin0 * 27 * 32;
    for (int idx = threadIdx.x; idx < 32 * 864; idx += 256) {            // wpb[cin][tap][co]
        const int co = idx & 31, ct = idx >> 5;                           // ct = cin_l * 27 + tap
        wpb[idx] = t[co * PACK_CO_STRIDE + ct];
    }
    for (int idx = threadIdx.x; idx < 32 * 864; idx += 256) {            // wpf[co][tap][cin]
        const int cl = idx & 31, q = idx >> 5, tap = q % 27, co = q / 27;
        e.wpf[((size_t)co * 27 + tap) * 128 + cin0 + cl] = t[co * PACK_CO_STRIDE + cl * 27 + tap];
    }
}
extern "C" int mms_pack_conv3_table_group_ex(const void* const* tables_dev, int ng, int nlayers, uint64_t fragmask, hipStream_t s) {
    if (!tables_dev || ng < 1 || ng > MMS_MAX_GROUP || nlayers <= 0 || nlayers > 64) return MMS_ERR_ARG;
    TabPtrs tp;
    for (int g = 0; g < ng; ++g) tp.t[g] = tables_dev[g];
    constexpr int smem = 32 * PACK_CO_STRIDE * sizeof(float);
    static std::once_flag attr_once;
    std::call_once(attr_once, [&] { hipFuncSetAttribute((const void*)pack_conv3_table_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem); });
    MMS_LAUNCH(pack_conv3_table_kernel, dim3(4, nlayers, ng), dim3(256), smem, s, tp, fragmask);
    return mms_check_launch();
}
extern "C" int mms_pack_conv3_table_group(const void* const* tables_dev, int ng, int nlayers, hipStream_t s) {
    return mms_pack_conv3_table_group_ex(tables_dev, ng, nlayers, 0ull, s);
}
extern "C" int mms_pack_conv3_table(const void* table_dev, int nlayers, hipStream_t s) {
    return mms_pack_conv3_table_group(&table_dev, 1, nlayers, s);
}

// running_mean/var momentum update (torch: running = 0.9*running + 0.1*batch, unbiased var; nbt += 1)
__global__ void bn_running_update_kernel(const TabPtrs tabs, float momentum) {
    const BnRunEntry e = ((const BnRunEntry*)tabs.t[blockIdx.z])[blockIdx.x];
    for (int c = threadIdx.x; c < e.C; c += blockDim.x) {
        double m = rep_sum(e.sum, c, e.nrep, e.rep_stride) / e.count, v = rep_sum(e.sumsq, c, e.nrep, e.rep_stride) / e.count - m * m;
        if (v < 0) v = 0;
        double unb = e.count > 1.f ? v * e.count / (e.count - 1.0) : v;
        e.rmean[c] = (1.f - momentum) * e.rmean[c] + momentum * (float)m;
        e.rvar[c] = (1.f - momentum) * e.rvar[c] + momentum * (float)unb;
    }
    if (threadIdx.x == 0 && e.nbt) *e.nbt += 1;
}
// zero-fill of one region per model (the per-step statistic accumulators + gradient scratch): one launch for the group
__global__ __launch_bounds__(256) void zero_regions_kernel(const TabPtrs regions, size_t n16) {
    float4* dst = (float4*)regions.t[blockIdx.z];
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
extern "C" int mms_zero_regions_group(void* const* regions_dev, int ng, size_t bytes, hipStream_t s) {
    if (!regions_dev || ng < 1 || ng > MMS_MAX_GROUP || (bytes & 15)) return MMS_ERR_ARG;
    if (bytes == 0) return MMS_OK;
    TabPtrs tp;
    for (int g = 0; g < ng; ++g) { if (((uintptr_t)regions_dev[g]) & 15) return MMS_ERR_ARG; tp.t[g] = regions_dev[g]; }
    size_t n16 = bytes >> 4, blocks = (n16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    MMS_LAUNCH(zero_regions_kernel, dim3((unsigned)blocks, 1, ng), dim3(256), 0, s, tp, n16);
    return mms_check_launch();
}
extern "C" int mms_bn_running_update_group(const void* const* tables_dev, int ng, int n, float momentum, hipStream_t s) {
    if (n <= 0) return MMS_OK;
    if (!tables_dev || ng < 1 || ng > MMS_MAX_GROUP) return MMS_ERR_ARG;
    TabPtrs tp;
    for (int g = 0; g < ng; ++g) tp.t[g] = tables_dev[g];
    MMS_LAUNCH(bn_running_update_kernel, dim3(n, 1, ng), dim3(256), 0, s, tp, momentum);
    return mms_check_launch();
}
extern "C" int mms_bn_running_update(const void* table_dev, int n, float momentum, hipStream_t s) {
    return mms_bn_running_update_group(&table_dev, 1, n, momentum, s);
}
