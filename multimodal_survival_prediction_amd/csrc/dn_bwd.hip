// DenseNet121-3D backward ops for gfx950 (what torch autograd would run for MONAI's DenseNet121 under
// loss.backward() at final_multimodal.py:258, partial_modality_training.py:426, simple_fusion.py:272).
// BatchNorm backward (training mode):  dx = g*rstd*(dbn - mean_m(dbn) - xhat*mean_m(dbn*xhat));
// the two per-channel sums are accumulated (fp64 atomics) by the kernel that produces dbn and consumed by
// the next kernel's operand prologue, so dx of the inner BN (norm2) is never materialised.
#include "dn_ops.h"
#include "tile_gemm.h"
#include <string.h>
#include <stdlib.h>

#define Z4 make_float4(0, 0, 0, 0)

// ------------------------------------------------------------------------------------------------------
// conv3 backward-data:  dbn2[m][cin] = [a2>0] * sum_{tap,cout} dz[m - off(tap)][cout] * W[cout][cin][tap]
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned tap_mask9(int c, Dims3 g, bool mirror) {
    int d, h, w;
    unpack_dhw(c, d, h, w);
    const unsigned lo_d = d > 0, hi_d = d + 1 < g.D, lo_h = h > 0, hi_h = h + 1 < g.H, lo_w = w > 0, hi_w = w + 1 < g.W;
    const unsigned dm = mirror ? (hi_d | 2u | (lo_d << 2)) : (lo_d | 2u | (hi_d << 2));
    const unsigned hm = mirror ? (hi_h | 2u | (lo_h << 2)) : (lo_h | 2u | (hi_h << 2));
    const unsigned wm = mirror ? (hi_w | 2u | (lo_w << 2)) : (lo_w | 2u | (hi_w << 2));
    return dm | (hm << 3) | (wm << 6);
}

__device__ __forceinline__ void store_tile_bwd(float* y, int M, int m0, const float* Cs, int tid) {
    for (int idx = tid; idx < 32 * 128; idx += 256) {
        const int r = idx >> 7, c = idx & 127, m = m0 + r;
        if (m < M) y[(size_t)m * 128 + c] = Cs[r * 129 + c];
    }
}

template <bool SPLIT>
struct Conv3BwdDataOp {
    typedef Conv3BwdDataP Params;
    static constexpr int WM = 1, WN = 4, WK = 1, AMODE = LD_K4, BMODE = LD_K4;
    static constexpr int TM = 32, TN = 128;
    static constexpr int EXTRA = 4 * 128 + 32 + 2 * 2 * 128 * 2;   // bn consts, (unused), fp64 reduction scratch
    typedef float4 ARaw;
    typedef float4 BRaw;
    float* ex;
    // TK = 32 = the 32 output channels of one tap; per-thread constants as in Conv3FwdOp (taps mirrored)
    int voff, woff[4], tapoff_b, wsoff, p_nsplit;
    unsigned m9, sel;
    buf_rsrc_t rz, rw;
    __device__ void setup(const Params& p, int m0, int, int, float* extra, int tid) {
        ex = extra;
        p_nsplit = p.nsplit > 0 ? p.nsplit : 27;
        if (tid < 128) {
            float mu, rstd, ga_, be_;
            bn_consts1(p.bn, tid, mu, rstd, ga_, be_);
            extra[tid] = mu; extra[128 + tid] = rstd; extra[256 + tid] = ga_; extra[384 + tid] = be_;
        }
        rz = make_rsrc(p.dz, (unsigned)(p.M - 1) * (unsigned)p.lddz * 4u + 128u);
        rw = make_rsrc(p.wpb, 128u * 27u * 32u * 4u);
        const int row = tid >> 3, co = (tid & 7) * 4, m = m0 + row;
        const bool valid = m < p.M;
        m9 = valid ? tap_mask9(p.coords[valid ? m : m0], p.g, true) : 0u;
        voff = ((valid ? m : m0) * p.lddz + co) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) woff[i] = (((tid >> 3) + 32 * i) * 864 + co) * 4;
    }
    __device__ void krange(const Params&, int z, int& kb, int& ke) {
        if (SPLIT) {
            const int tpw = (27 + p_nsplit - 1) / p_nsplit;
            kb = z * tpw * 32; ke = kb + tpw * 32; if (ke > 27 * 32) ke = 27 * 32;
        } else { kb = 0; ke = 27 * 32; }
    }
    __device__ void step(const Params& p, int k0) {
        const int tap = k0 >> 5, kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        sel = (1u << kd) | (8u << kh) | (64u << kw);
        tapoff_b = -(((kd - 1) * p.g.H + (kh - 1)) * p.g.W + (kw - 1)) * p.lddz * 4;
        wsoff = k0 * 4;
    }
    __device__ float4 a_ld(const Params&, int, int, int, bool& ok) const {
        ok = (m9 & sel) == sel;
        return buf_load4(rz, voff + (ok ? tapoff_b : 0), 0);
    }
    __device__ float4 a_tx(const Params&, int, const float4& v, int, int, bool ok) const {
        const float z = ok ? 1.f : 0.f;
        return make_float4(z * v.x, z * v.y, z * v.z, z * v.w);
    }
    __device__ float4 b_ld(const Params&, int i, int, int, bool& ok) const { ok = true; return buf_load4(rw, woff[i], wsoff); }
    __device__ float4 b_tx(const Params&, int, const float4& v, int, int, bool) const { return v; }
    __device__ void epilogue(const Params& p, int m0_, int, int z, const float* Cs, int tid, bool active) {
        if (SPLIT) {       // raw partial tile; mask + BN sums happen in conv3_bwd_data_reduce_kernel
            if (active) store_tile_bwd(p.partial + (size_t)z * p.M * 128, p.M, m0_, Cs, tid);
            return;
        }
        const int c = tid & 127, rg = tid >> 7;
        const float mu = ex[c], rstd = ex[128 + c], ga = ex[256 + c], be = ex[384 + c];
        double s1 = 0, s2 = 0;
        const int rows = !active ? 0 : (p.M - m0_ < TM ? p.M - m0_ : TM);
        float yv[TM / 2];                          // this thread's TM/2 rows: every y1 load in flight before the first use
#pragma unroll
        for (int i = 0; i < TM / 2; ++i) {
            const int r = rg + 2 * i;
            yv[i] = r < rows ? p.y1[(size_t)(m0_ + r) * 128 + c] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < TM / 2; ++i) {
            const int r = rg + 2 * i;
            if (r < rows) {
                const float xh = (yv[i] - mu) * rstd;
                const float pre = fmaf(ga, xh, be);
                const float g = pre > 0.f ? Cs[r * (TN + 1) + c] : 0.f;
                p.dbn[(size_t)(m0_ + r) * 128 + c] = g;
                s1 += g; s2 += (double)g * xh;
            }
        }
        double* red = (double*)(ex + 544);     // [2][2][128], 8-byte aligned (544*4 = 2176)
        if (active) { red[(rg * 2 + 0) * 128 + c] = s1; red[(rg * 2 + 1) * 128 + c] = s2; }
        __syncthreads();
        if (rg == 0 && active) {
            atomicAdd(&stat_rep(p.s1, p.srep, p.sstride)[c], red[c] + red[2 * 128 + c]);
            atomicAdd(&stat_rep(p.s2, p.srep, p.sstride)[c], red[128 + c] + red[3 * 128 + c]);
        }
    }
};

// tap-split variant: sum the 27 partials, apply the relu2 mask, write dbn2 and the BN2-backward sums
__global__ __launch_bounds__(256) void conv3_bwd_data_reduce_kernel(const Grp<Conv3BwdDataP> grp) {
    const Conv3BwdDataP& p = grp.p[blockIdx.z];
    __shared__ double red[2][2][128];
    const int c = threadIdx.x & 127, rg = threadIdx.x >> 7;
    float mu, rstd;
    bn_mean_rstd(p.bn, c, mu, rstd);
    const float ga = p.bn.gamma[c], be = p.bn.beta[c];
    double s1 = 0, s2 = 0;
    const int mend = blockIdx.x * 4 + 4 < p.M ? blockIdx.x * 4 + 4 : p.M;
    for (int m = blockIdx.x * 4 + rg; m < mend; m += 2) {
        float v[27];
#pragma unroll
        for (int t = 0; t < 27; ++t) v[t] = t < p.nsplit ? p.partial[((size_t)t * p.M + m) * 128 + c] : 0.f;    // loads in flight
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < 27; ++t) a += v[t];
        const size_t o = (size_t)m * 128 + c;
        const float xh = (p.y1[o] - mu) * rstd;
        const float g = fmaf(ga, xh, be) > 0.f ? a : 0.f;
        p.dbn[o] = g;
        s1 += g; s2 += (double)g * xh;
    }
    red[0][rg][c] = s1; red[1][rg][c] = s2;
    __syncthreads();
    if (rg == 0) {
        atomicAdd(&stat_rep(p.s1, p.srep, p.sstride)[c], red[0][0][c] + red[0][1][c]);
        atomicAdd(&stat_rep(p.s2, p.srep, p.sstride)[c], red[1][0][c] + red[1][1][c]);
    }
}

// same for launches with many rows and few partials (block 2: 1024 rows per model, 3-4 partials): 16 rows per workgroup, a thread's 8 rows
// and their <= 4 partials all in flight before the first use, a quarter of the statistic atomics (13.1 -> 11.2 us per launch of 3 models; K = 5 epoch +1 %.  32 rows per workgroup measured slower)
template <int RT>      // rows per thread; 2 RT rows per workgroup
__global__ __launch_bounds__(256) void conv3_bwd_data_reduce16_kernel(const Grp<Conv3BwdDataP> grp) {
    const Conv3BwdDataP& p = grp.p[blockIdx.z];
    __shared__ double red[2][2][128];
    const int c = threadIdx.x & 127, rg = threadIdx.x >> 7;
    float mu, rstd;
    bn_mean_rstd(p.bn, c, mu, rstd);
    const float ga = p.bn.gamma[c], be = p.bn.beta[c];
    const int m0 = blockIdx.x * (2 * RT) + rg;
    float v[RT][4], y[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int m = m0 + 2 * i;
        const bool ok = m < p.M;
#pragma unroll
        for (int t = 0; t < 4; ++t) v[i][t] = (ok && t < p.nsplit) ? p.partial[((size_t)t * p.M + m) * 128 + c] : 0.f;
        y[i] = ok ? p.y1[(size_t)m * 128 + c] : 0.f;
    }
    double s1 = 0, s2 = 0;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int m = m0 + 2 * i;
        if (m < p.M) {
            float a = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) a += v[i][t];             // same order as the 27-slot loop of the general kernel (zeros beyond nsplit)
            const float xh = (y[i] - mu) * rstd;
            const float g = fmaf(ga, xh, be) > 0.f ? a : 0.f;
            p.dbn[(size_t)m * 128 + c] = g;
            s1 += g; s2 += (double)g * xh;
        }
    }
    red[0][rg][c] = s1; red[1][rg][c] = s2;
    __syncthreads();
    if (rg == 0) {
        atomicAdd(&stat_rep(p.s1, p.srep, p.sstride)[c], red[0][0][c] + red[0][1][c]);
        atomicAdd(&stat_rep(p.s2, p.srep, p.sstride)[c], red[1][0][c] + red[1][1][c]);
    }
}

// ------------------------------------------------------------------------------------------------------
// conv2 backward-data, multi-tap form (block 1: launches with >= 256 tiles of 32 rows; the mirror of conv3_fwd_mt_kernel, dn_fwd.hip).
// dy1[m][cin] = sum_tap sum_co dz[m - off(tap)][co] * W[co][tap][cin].  A workgroup owns 32 rows x 128 input channels, wave w the channels
// 32w..32w+31 (no split of the reduction: K = 32 output channels per tap, 16 MFMAs per tap and wave).  dz is only 32 channels wide, so the
// windows of ALL three kd planes -- rows [m0 - (kd-1) HW - (W+1), .. + 32 + 2(W+1)) x 32 floats, 21.7 KB at W = 8 -- are staged once: no
// barrier inside the tap loop.  A operand: the window slot shifted by -((kh-1) W + (kw-1)), or a row of zeros where the tap's (mirrored)
// zero padding excludes the row; B operand: straight from the [cin][tap][cout] pack into the MFMA register layout, two taps ahead (ring of
// three).  Half-tap software pipeline and pinned schedule as in the forward kernel.  Epilogue from the accumulator registers (a lane holds
// 16 rows of ONE channel): relu2 mask from y1, dbn2 store, BatchNorm-backward sums -- no LDS staging, no barrier.
// ------------------------------------------------------------------------------------------------------
#define C3D_BP 36
#define C3D_MAXHALO 17
#define C3D_PIN() do { if constexpr (PIN) { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } } while (0)
template <bool PIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void conv3_bwd_data_mt_kernel(const Grp<Conv3BwdDataP> grp) {
    int gi, bx;
    xcd_place(gi, bx);
    const Conv3BwdDataP& p = grp.p[gi];
    const float* __restrict__ dz = p.dz;
    const float* __restrict__ wpb = p.wpb;
    const float* __restrict__ y1 = p.y1;
    float* __restrict__ dbn = p.dbn;
    const int M = p.M, lddz = p.lddz, W = p.g.W, HW = p.g.H * p.g.W;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [3][nrows][36] + one row of zeros
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kq = lane >> 5;
    const int m0 = bx * 32, halo = W + 1, nrows = 32 + 2 * halo;
    const int cin = 32 * wave + li;
    // ---- prologue: every load issued before the first use (constants of this lane's channel, the row's tap mask, the three windows)
    float mu, rstd, ga, be;
    bn_consts1(p.bn, cin, mu, rstd, ga, be);
    const int myrow = m0 + li;
    const int mycoord = p.coords[myrow < M ? myrow : 0];
    constexpr int NR = (32 + 2 * C3D_MAXHALO + 31) / 32;               // row passes per window: 3
    float4 wv[3][NR];
    const int wr = tid >> 3, wc = (tid & 7) * 4;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = wr + 32 * i, src = m0 - (kd - 1) * HW - halo + r;
            const bool ok = r < nrows && src >= 0 && src < M;
            const float4 v = *(const float4*)(dz + (size_t)(ok ? src : 0) * lddz + wc);
            wv[kd][i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    const float* wl = wpb + (size_t)cin * 864 + 4 * kq;
    float4 b[3][4];
    auto bload = [&](float4 (&bb)[4], int tap) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bb[q] = *(const float4*)(wl + tap * 32 + 8 * q);
    };
    bload(b[0], 0);
    bload(b[1], 1);
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = wr + 32 * i;
            if (r < nrows) *(float4*)&smem[(kd * nrows + r) * C3D_BP + wc] = wv[kd][i];
        }
    if (tid < C3D_BP) smem[3 * nrows * C3D_BP + tid] = 0.f;
    const unsigned m9 = myrow < M ? tap_mask9(mycoord, p.g, true) : 0u;
    __syncthreads();
    const float* arow = smem + (li + halo) * C3D_BP + 4 * kq;
    const float* azero = smem + 3 * nrows * C3D_BP + 4 * kq;
    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    float4 aL[2], aH[2];
    auto aread = [&](float4 (&a)[2], int tap, int half) __attribute__((always_inline)) {
        const int kd = tap / 9, t9 = tap - 9 * kd, kh = t9 / 3, kw = t9 - 3 * kh;
        const unsigned sel = (1u << kd) | (8u << kh) | (64u << kw);
        const float* ar = (m9 & sel) == sel ? arow + (kd * nrows - ((kh - 1) * W + (kw - 1))) * C3D_BP : azero;
#pragma unroll
        for (int q = 0; q < 2; ++q) a[q] = *(const float4*)(ar + 8 * (half * 2 + q));
    };
    auto mma = [&](const float4 (&a)[2], const float4 (&bb)[4], int half) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {          // two accumulators: consecutive MFMAs never wait for each other's result
            const float4& bq = bb[half * 2 + q];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, bq.x, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, bq.y, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, bq.z, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, bq.w, acc2, 0, 0, 0);
        }
    };
    aread(aL, 0, 0);
    static_for<27>([&](auto T) __attribute__((always_inline)) {
        constexpr int tap = decltype(T)::value;
        if constexpr (tap + 2 < 27) bload(b[(tap + 2) % 3], tap + 2);
        aread(aH, tap, 1);
        C3D_PIN();
        mma(aL, b[tap % 3], 0);
        C3D_PIN();
        if constexpr (tap + 1 < 27) aread(aL, tap + 1, 0);
        C3D_PIN();
        mma(aH, b[tap % 3], 1);
        C3D_PIN();
    });
    // ---- epilogue: lane (channel cin, row half kq) holds rows 4 kq + (r & 3) + 8 (r >> 2), r = 0..15
    float yv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + 4 * kq + (r & 3) + 8 * (r >> 2);
        yv[r] = y1[(size_t)(m < M ? m : 0) * 128 + cin];
    }
    double s1 = 0, s2 = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + 4 * kq + (r & 3) + 8 * (r >> 2);
        if (m < M) {
            const float xh = (yv[r] - mu) * rstd;
            const float pre = fmaf(ga, xh, be);
            const float g = pre > 0.f ? acc[r] + acc2[r] : 0.f;
            dbn[(size_t)m * 128 + cin] = g;
            s1 += g; s2 += (double)g * xh;
        }
    }
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (kq == 0) {
        atomicAdd(&stat_rep(p.s1, p.srep, p.sstride)[cin], s1);
        atomicAdd(&stat_rep(p.s2, p.srep, p.sstride)[cin], s2);
    }
}
// Taken where the forward takes its multi-tap form (conv3_mt_tile, dn_fwd.hip): un-split launches of >= MmsDnOpts.conv3_mt32_min (256) tiles of 32
// rows on levels with >= 1024 rows and W <= 16; MmsDnOpts.conv3_mt: -1 = never, 2 / 3 = whenever it applies (tests).
static inline bool conv3_bwd_data_mt_ok(int M, int ng, const Dims3& g, const MmsDnOpts& o) {
    if (g.W + 1 > C3D_MAXHALO || o.conv3_mt < 0) return false;
    if (o.conv3_mt == 2 || o.conv3_mt == 3) return M >= 64;
    const int min32 = o.conv3_mt32_min > 0 ? o.conv3_mt32_min : 256;
    return M >= 1024 && (long)((M + 31) / 32) * ng >= min32;
}
template <bool PIN>
static int launch_conv3_bwd_data_mt(const Conv3BwdDataP* pp, int ng, hipStream_t s) {
    const Conv3BwdDataP& p = *pp;
    const int smem = (3 * (32 + 2 * (p.g.W + 1)) + 1) * C3D_BP * (int)sizeof(float);      // W = 8: 21.7 KB, W = 16: 28.7 KB
    Grp<Conv3BwdDataP> a;
    grp_fill(a, pp, ng, 1);
    MMS_LAUNCH((conv3_bwd_data_mt_kernel<PIN>), dim3((p.M + 31) / 32, 1, ng), dim3(256), smem, s, a);
    return mms_check_launch();
}

extern "C" int mms_conv3_bwd_data_group(const Conv3BwdDataP* pp, int ng, const MmsDnOpts* opts, hipStream_t s) {
    if (!pp || ng < 1 || ng > MMS_MAX_GROUP) return MMS_ERR_ARG;
    const MmsDnOpts o = mms_opts(opts);
    const Conv3BwdDataP& p = *pp;
    if (p.M <= 0 || p.lddz % 4 != 0) return MMS_ERR_ARG;
    for (int g = 1; g < ng; ++g) {
        const Conv3BwdDataP& q = pp[g];
        if (q.M != p.M || q.lddz % 4 != 0 || q.g.D != p.g.D || q.g.H != p.g.H || q.g.W != p.g.W ||
            (q.partial == nullptr) != (p.partial == nullptr) || q.nsplit != p.nsplit) return MMS_ERR_ARG;
    }
    for (int g = 1; g < ng; ++g) if (pp[g].wfrag != p.wfrag) return MMS_ERR_ARG;
    if (p.wfrag) return (!p.partial && mms_conv3_small_jn(p.M, ng, p.g, o)) ? mms_c3s_bwd_data(pp, ng, o, s) : MMS_ERR_ARG;   // fragment-ordered weights: the small-grid kernel only
    if (p.partial) {
        if (p.nsplit < 1 || p.nsplit > 27 || (p.nsplit - 1) * ((27 + p.nsplit - 1) / p.nsplit) >= 27) return MMS_ERR_ARG;
        int rc = launch_tile_gemm<Conv3BwdDataOp<true>>(pp, ng, dim3((p.M + 31) / 32, 1, p.nsplit), s);
        if (rc != MMS_OK) return rc;
        Grp<Conv3BwdDataP> a;
        grp_fill(a, pp, ng, 1);
        if (p.nsplit <= 4 && p.M >= 512) MMS_LAUNCH(conv3_bwd_data_reduce16_kernel<8>, dim3((p.M + 15) / 16, 1, ng), dim3(256), 0, s, a);
        else MMS_LAUNCH(conv3_bwd_data_reduce_kernel, dim3((p.M + 3) / 4, 1, ng), dim3(256), 0, s, a);
        return mms_check_launch();
    }
    if (conv3_bwd_data_mt_ok(p.M, ng, p.g, o)) {
        for (int g = 0; g < ng; ++g) if (pp[g].lddz % 4 != 0 || ((uintptr_t)pp[g].dz & 15) != 0) return MMS_ERR_ARG;     // dz rows are read with 16-byte loads
        // pinned schedule while the launch is at most one round of three workgroups per CU
        return (long)((p.M + 31) / 32) * ng <= 768 ? launch_conv3_bwd_data_mt<true>(pp, ng, s) : launch_conv3_bwd_data_mt<false>(pp, ng, s);
    }
    if (mms_conv3_small_jn(p.M, ng, p.g, o)) return mms_c3s_bwd_data(pp, ng, o, s);      // small grids: 16-row tiles, all taps, no reduce launch
    return launch_tile_gemm<Conv3BwdDataOp<false>>(pp, ng, dim3((p.M + 31) / 32, 1, 1), s);
}
MMS_SINGLE_O(mms_conv3_bwd_data, Conv3BwdDataP)

// ------------------------------------------------------------------------------------------------------
// conv3 backward-weight: dW[cout][cin][tap] += sum_m a2[m + off(tap)][cin] * dz[m][cout]
// tile: rows = cin (128), cols = cout (32), reduction index = voxel m (split over grid.z with the tap)
// ------------------------------------------------------------------------------------------------------
struct Conv3BwdWOp {
    typedef Conv3BwdWP Params;
    static constexpr int WM = 4, WN = 1, WK = 1, AMODE = LD_R4, BMODE = LD_R4;
    static constexpr int TM = 128, TN = 32;
    static constexpr int EXTRA = 1024;       // tap-validity mask of every voxel of this workgroup's row chunk
    typedef float4 ARaw;
    typedef float4 BRaw;
    // TM/4 = 32 row groups: a thread's 4 input channels ((tid & 31) * 4 ...) never change -> BN constants in registers
    float mean[4], sc[4], beta[4];
    const unsigned* vm;
    int tap, mb, me, avoff[4], bvoff, tapoff_b, asoff, bsoff, k0s;
    unsigned sel;
    buf_rsrc_t ry, rz;
    // XCD placement: the 27 tap workgroups of one (model, row chunk) pair read the same y1 / dz rows.  Dealt round-robin
    // they land on all 8 XCDs and every L2 streams every chunk (measured: 12x the algorithmic bytes at the fabric);
    // with the pairs partitioned over the XCDs (pair % 8) each L2 only sees its own chunks.  z = chunk * 27 + tap.
    static __device__ void zremap(int flat, int zdim, int nflat, int& gi, int& z) {
        const int pairs = nflat / 27, ms = zdim / 27;
        if ((pairs & 7) == 0) {
            const int x = flat & 7, slot = flat >> 3, pair = (slot / 27) * 8 + x;
            gi = pair / ms;
            z = (pair - gi * ms) * 27 + slot % 27;
        } else { gi = flat / zdim; z = flat - gi * zdim; }
    }
    __device__ void setup(const Params& p, int, int, int z, float* extra, int tid) {
        const int c0 = (tid & 31) * 4;
        bn_consts4(p.bn, c0, mean, sc, beta);
        tap = z % 27;
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        sel = (1u << kd) | (8u << kh) | (64u << kw);
        tapoff_b = (((kd - 1) * p.g.H + (kh - 1)) * p.g.W + (kw - 1)) * 512;
        const int chunk = z / 27;
        int mc = (p.M + p.msplit - 1) / p.msplit;
        mc = (mc + 31) & ~31;
        mb = chunk * mc;
        me = mb + mc < p.M ? mb + mc : p.M;
        vm = (const unsigned*)extra;
        for (int j = tid; j < mc; j += 256) ((unsigned*)extra)[j] = (mb + j < me) ? tap_mask9(p.coords[mb + j], p.g, false) : 0u;
        ry = make_rsrc(p.y1, (unsigned)p.M * 512u);
        rz = make_rsrc(p.dz, (unsigned)(p.M - 1) * (unsigned)p.lddz * 4u + 128u);
#pragma unroll
        for (int i = 0; i < 4; ++i) avoff[i] = (((tid >> 5) + 8 * i) * 128 + c0) * 4;
        bvoff = ((tid >> 3) * p.lddz + (tid & 7) * 4) * 4;
    }
    __device__ void krange(const Params&, int, int& kb, int& ke) { kb = mb; ke = me; }
    __device__ void step(const Params& p, int k0) { k0s = k0; asoff = k0 * 512; bsoff = k0 * p.lddz * 4; }
    __device__ float4 a_ld(const Params&, int i, int, int, bool& ok) const {      // A(row = cin.., k = voxel m)
        const unsigned mk = vm[k0s - mb + (threadIdx.x >> 5 & 7) + 8 * i];
        ok = (mk & sel) == sel;
        // the voffset operand must stay non-negative on its own (offsets are unsigned, the range check sees the
        // un-wrapped sum), so the tile offset is added in the VALU rather than passed as soffset
        return buf_load4(ry, avoff[i] + asoff + (ok ? tapoff_b : 0), 0);     // rows >= M: hardware range check -> 0
    }
    __device__ float4 a_tx(const Params&, int, const float4& v, int, int, bool ok) const {   // branch-free
        const float z = ok ? 1.f : 0.f;
        return make_float4(z * fmaxf(bn_apply(v.x, mean[0], sc[0], beta[0]), 0.f), z * fmaxf(bn_apply(v.y, mean[1], sc[1], beta[1]), 0.f),
                           z * fmaxf(bn_apply(v.z, mean[2], sc[2], beta[2]), 0.f), z * fmaxf(bn_apply(v.w, mean[3], sc[3], beta[3]), 0.f));
    }
    __device__ float4 b_ld(const Params&, int, int, int m, bool& ok) const {      // B(col = cout.., k = voxel m)
        ok = m < me;
        return buf_load4(rz, bvoff, bsoff);
    }
    __device__ float4 b_tx(const Params&, int, const float4& v, int, int, bool ok) const {
        const float z = ok ? 1.f : 0.f;
        return make_float4(z * v.x, z * v.y, z * v.z, z * v.w);
    }
    __device__ void epilogue(const Params& p, int, int, int, const float* Cs, int tid, bool active) {
        if (mb >= me || !active) return;
#ifdef MMS_ABLATE_FLUSH
        if (p.M > 0) return;
#endif
        for (int idx = tid; idx < TM * TN; idx += 256) {
            const int cin = idx & 127, co = idx >> 7;
            const size_t dst = p.dw_layout == 2 ? ((size_t)co * 27 + tap) * 128 + cin : (p.dw_layout == 1 ? ((size_t)tap * 32 + co) * 128 + cin : ((size_t)co * 128 + cin) * 27 + tap);
            atomicAdd(&p.dw[dst], Cs[cin * (TN + 1) + co]);
        }
    }
};

// tap-major gradient scratch -> canonical torch layout, through LDS so both sides are coalesced.
// one workgroup per (layer, cout): 27 x 128 floats
struct UnpackTable { UnpackEntry e[64]; };      // passed BY VALUE as a kernel argument (1 KB): capture-safe, no device table
__global__ __launch_bounds__(256) void unpack_conv3_grads_kernel(const UnpackTable tab) {
    __shared__ float t[27 * 129];
    const UnpackEntry e = tab.e[blockIdx.y];
    const int co = blockIdx.x;
    for (int idx = threadIdx.x; idx < 27 * 128; idx += 256) {
        const int tap = idx >> 7, cin = idx & 127;
        t[tap * 129 + cin] = e.scratch[((size_t)tap * 32 + co) * 128 + cin];
    }
    __syncthreads();
    float* dst = e.dw + (size_t)co * 128 * 27;
    for (int idx = threadIdx.x; idx < 27 * 128; idx += 256) {
        const int cin = idx / 27, tap = idx % 27;
        dst[idx] += t[tap * 129 + cin];
    }
}
// fold-group form: the scratch of layer i of model g is scratch[g] + i * 27*32*128 (how the driver lays it out), so a table
// entry is just the destination pointer: 8 models x 58 layers x 8 B = 3.7 KB of kernarg, one launch for the group
#define UNPACK_MAXG 7        // 7 x 58 x 8 B + 56 B = 3.3 KB of kernarg; larger groups take two launches
struct UnpackGroup { const float* scratch[UNPACK_MAXG]; float* dw[UNPACK_MAXG][58]; };
__global__ __launch_bounds__(256) void unpack_conv3_grads_group_kernel(const UnpackGroup tab) {
    __shared__ float t[27 * 129];
    const int co = blockIdx.x, layer = blockIdx.y, g = blockIdx.z;
    const float* scratch = tab.scratch[g] + (size_t)layer * 27 * 32 * 128;
    for (int idx = threadIdx.x; idx < 27 * 128; idx += 256) {
        const int tap = idx >> 7, cin = idx & 127;
        t[tap * 129 + cin] = scratch[((size_t)tap * 32 + co) * 128 + cin];
    }
    __syncthreads();
    float* dst = tab.dw[g][layer] + (size_t)co * 128 * 27;
    for (int idx = threadIdx.x; idx < 27 * 128; idx += 256) {
        const int cin = idx / 27, tap = idx % 27;
        dst[idx] += t[tap * 129 + cin];
    }
}
extern "C" int mms_unpack_conv3_grads_group(const float* const* scratch, float* const* const* dw, int ng, int nlayers, hipStream_t s) {
    if (nlayers <= 0) return MMS_OK;
    if (!scratch || !dw || ng < 1 || ng > MMS_MAX_GROUP || nlayers > 58) return MMS_ERR_ARG;
    for (int g0 = 0; g0 < ng; g0 += UNPACK_MAXG) {
        const int n = ng - g0 < UNPACK_MAXG ? ng - g0 : UNPACK_MAXG;
        UnpackGroup t;
        for (int g = 0; g < n; ++g) {
            t.scratch[g] = scratch[g0 + g];
            for (int i = 0; i < nlayers; ++i) t.dw[g][i] = dw[g0 + g][i];
        }
        MMS_LAUNCH(unpack_conv3_grads_group_kernel, dim3(32, nlayers, n), dim3(256), 0, s, t);
        const int rc = mms_check_launch();
        if (rc != MMS_OK) return rc;
    }
    return MMS_OK;
}
extern "C" int mms_unpack_conv3_grads(const void* table_host, int nlayers, hipStream_t s) {
    if (nlayers <= 0) return MMS_OK;
    if (nlayers > 64) return MMS_ERR_ARG;
    UnpackTable t;
    memcpy(t.e, table_host, sizeof(UnpackEntry) * nlayers);
    MMS_LAUNCH(unpack_conv3_grads_kernel, dim3(32, nlayers), dim3(256), 0, s, t);
    return mms_check_launch();
}

// ------------------------------------------------------------------------------------------------------
// conv3 backward-weight, multi-tap form (launches with enough row chunks to fill the chip with a third of the workgroups):
// a workgroup owns one (model, row chunk, kd, kh) and produces the gradients of its THREE kw taps at once.  Their a2 operands
// are the same rows shifted by one voxel, so per 32-row step the a2 rows [r0-1, r0+33) + off(kd,kh) are normalised and staged
// ONCE (34 x 128, k-major) and every fragment read from LDS feeds three MFMAs (three independent accumulators per wave,
// wave w = input channels 32w..32w+31); the tap validity (zero padding) rides on the small operand: dz[r] * valid(r, tap),
// applied in registers from the chunk's 9-bit masks.  Versus the one-tap GEMM form: a third of the a2 loads / BN transforms /
// LDS stores per MFMA and 81 instead of 96 LDS fragment reads per 48 MFMAs.
// ------------------------------------------------------------------------------------------------------
#define C3W_AP 132                      // a2 image pitch (floats): [34][132]
#define C3W_BP 36                       // dz image pitch: [32][36]
#define C3W_STAGE (34 * C3W_AP + 32 * C3W_BP)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void conv3_bwdw_mt_kernel(const Grp<Conv3BwdWP> grp) {
    // blockIdx.z = flat over (model, chunk, kd*3+kh); the 9 workgroups of a (model, chunk) pair share their rows: one XCD per pair
    int gi, z;
    {
        const int flat = blockIdx.z, nflat = gridDim.z, zdim = grp.zdim, pairs = nflat / 9, ms = zdim / 9;
        if ((pairs & 7) == 0) {
            const int x = flat & 7, slot = flat >> 3, pair = (slot / 9) * 8 + x;
            gi = pair / ms;
            z = (pair - gi * ms) * 9 + slot % 9;
        } else { gi = flat / zdim; z = flat - gi * zdim; }
    }
    const Conv3BwdWP& p = grp.p[gi];
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float2* vf = (float2*)(smem + 2 * C3W_STAGE);                      // per chunk row: 1/0 for "kw = 0 valid", "kw = 2 valid"
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    const int kdh = z % 9, chunk = z / 9, kd = kdh / 3, kh = kdh - 3 * kd;
    const int offdh = ((kd - 1) * p.g.H + (kh - 1)) * p.g.W;
    int mc = (p.M + p.msplit - 1) / p.msplit;
    mc = (mc + 31) & ~31;
    const int mb = chunk * mc, me = mb + mc < p.M ? mb + mc : p.M;
    if (mb >= me) return;
    const int c0 = (tid & 31) * 4;
    float mean[4], sc[4], beta[4];
    bn_consts4(p.bn, c0, mean, sc, beta);
    // zero padding: the (kd, kh) part of a row's tap validity is common to the three taps and is folded into the staged dz
    // row; the kw part (kw = 1 is always valid) is a per-row factor pair read next to the dz fragment
    for (int j = tid; j < mc; j += 256) {
        const unsigned mk = (mb + j < me) ? tap_mask9(p.coords[mb + j], p.g, false) : 0u;
        vf[j] = make_float2(mk & 64u ? 1.f : 0.f, mk & 256u ? 1.f : 0.f);
    }
    const unsigned selb = (1u << kd) | (8u << kh);

    float4 ra[5], rb;
    unsigned oka = 0;
    bool okb = false;
    auto gload = [&](int r0) __attribute__((always_inline)) {
        oka = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int j = (tid >> 5) + 8 * i, src = r0 + j - 1 + offdh;
            const bool ok = j < 34 && src >= 0 && src < p.M;
            oka |= (ok ? 1u : 0u) << i;
            ra[i] = ok ? *(const float4*)(p.y1 + (size_t)src * 128 + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int m = r0 + (tid >> 3);
        rb = m < me ? *(const float4*)(p.dz + (size_t)m * p.lddz + (tid & 7) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        okb = m < me && (tap_mask9(p.coords[m], p.g, false) & selb) == selb;
    };
    auto sstore = [&](float* st) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int j = (tid >> 5) + 8 * i;
            if (j < 34) {
                const float zf = (oka >> i) & 1u ? 1.f : 0.f;
                *(float4*)&st[j * C3W_AP + c0] =
                    make_float4(zf * fmaxf(bn_apply(ra[i].x, mean[0], sc[0], beta[0]), 0.f), zf * fmaxf(bn_apply(ra[i].y, mean[1], sc[1], beta[1]), 0.f),
                                zf * fmaxf(bn_apply(ra[i].z, mean[2], sc[2], beta[2]), 0.f), zf * fmaxf(bn_apply(ra[i].w, mean[3], sc[3], beta[3]), 0.f));
            }
        }
        const float zb = okb ? 1.f : 0.f;
        *(float4*)&st[34 * C3W_AP + (tid >> 3) * C3W_BP + (tid & 7) * 4] = make_float4(zb * rb.x, zb * rb.y, zb * rb.z, zb * rb.w);
    };
    f32x16 acc0, acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; }
    auto mma = [&](const float* st, int t) __attribute__((always_inline)) {
        const float* at = st + h * C3W_AP + 32 * wave + li;          // lane (i = li, k-parity h): a2 image rows j + h
        const float* bt = st + 34 * C3W_AP + h * C3W_BP + li;         // dz rows k + h, column cout = li
        const float2* vt = vf + 32 * t + h;
        float a[33];
#ifdef C3W_NO_READ
#pragma unroll
        for (int j = 0; j < 33; ++j) a[j] = (float)(j + t);
#else
#pragma unroll
        for (int j = 0; j < 33; ++j) a[j] = at[j * C3W_AP];
#endif
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int kk = 2 * q;
#ifdef C3W_NO_READ
            const float b = (float)(q + t);
            const float2 f = make_float2((float)(t & 1), 1.f);
#else
            const float b = bt[kk * C3W_BP];
            const float2 f = vt[kk];
#endif
#ifdef C3W_NO_MFMA
            acc0[q] += a[kk] * (b * f.x);
            acc1[q] += a[kk + 1] * b;
            acc2[q] += a[kk + 2] * (b * f.y);
#else
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b * f.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk + 1], b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk + 2], b * f.y, acc2, 0, 0, 0);
#endif
        }
    };
    const int T = (me - mb + 31) / 32;
    gload(mb);
    sstore(smem);
    __syncthreads();                                                   // also publishes vf
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        float* cur = smem + (t & 1) * C3W_STAGE;
        float* nxt = smem + ((t + 1) & 1) * C3W_STAGE;
#ifndef C3W_NO_LOAD
        if (t + 1 < T) gload(mb + 32 * (t + 1));                     // in flight during the matrix phase
#endif
        mma(cur, t);
#ifndef C3W_NO_LOAD
        if (t + 1 < T) sstore(nxt);                                   // the other buffer: last read before the previous barrier
#endif
#ifndef C3W_NO_BARRIER
        __syncthreads();
#endif
    }
    // ---- flush: one tap at a time through LDS so that the atomics run along cin (contiguous in the tap-major scratch)
    float* Cs = smem;                                                  // [128][33]
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const f32x16& acc = kw == 0 ? acc0 : (kw == 1 ? acc1 : acc2);
#pragma unroll
        for (int r = 0; r < 16; ++r) Cs[(32 * wave + 4 * h + (r & 3) + 8 * (r >> 2)) * 33 + li] = acc[r];
        __syncthreads();
        const int tap = kdh * 3 + kw;
        for (int idx = tid; idx < 128 * 32; idx += 256) {
            const int cin = idx & 127, co = idx >> 7;
            const size_t dst = p.dw_layout == 2 ? ((size_t)co * 27 + tap) * 128 + cin : (p.dw_layout == 1 ? ((size_t)tap * 32 + co) * 128 + cin : ((size_t)co * 128 + cin) * 27 + tap);
            atomicAdd(&p.dw[dst], Cs[cin * 33 + co]);
        }
        __syncthreads();
    }
}
// MmsDnOpts.conv3w_mt: -1 = never, 2 = always (tests); default: chunks of >= 512 rows (16 steps to amortise the three-tap flush) whose
// 9-per-chunk grid fills >= 90 % of a whole number of rounds of the chip (3 workgroups x 256 CUs).  Measured per launch,
// block 1 of a fold group (tools/run_mt.sh): 10 models x 8192 rows, 720 workgroups: 247 -> 216 us; 5 models on 512-row chunks,
// 720: 135 -> 119 us; but 8 models (576 workgroups = 0.75 round): 199 -> 210 us, 4 models on 512-row chunks (576): 108 -> 116 us.
static inline bool conv3w_mt_ok(int rows_per_chunk, int msplit, int ng, const MmsDnOpts& o) {
    if (o.conv3w_mt < 0) return false;
    if (o.conv3w_mt == 2) return true;
    return rows_per_chunk >= 512 && mms_conv3w_mt_fills((long)msplit * ng * 9);
}

extern "C" int mms_conv3_bwd_weight_group(const Conv3BwdWP* pp, int ng, const MmsDnOpts* opts, hipStream_t s) {
    if (!pp || ng < 1 || ng > MMS_MAX_GROUP) return MMS_ERR_ARG;
    const MmsDnOpts o = mms_opts(opts);
    const Conv3BwdWP& p = *pp;
    if (p.M <= 0 || p.msplit <= 0 || p.lddz % 4 != 0 || p.dw_layout < 0 || p.dw_layout > 2) return MMS_ERR_ARG;
    if ((((p.M + p.msplit - 1) / p.msplit + 31) & ~31) > 1024) return MMS_ERR_ARG;    // row chunk must fit the LDS mask table
    for (int g = 0; g < ng; ++g) if (!mms_bn_aligned16(pp[g].bn)) return MMS_ERR_ARG;   // BatchNorm blocks are read with 16-byte vector loads
    for (int g = 1; g < ng; ++g) {
        const Conv3BwdWP& q = pp[g];
        if (q.M != p.M || q.msplit != p.msplit || q.lddz % 4 != 0 || q.g.D != p.g.D || q.g.H != p.g.H || q.g.W != p.g.W ||
            q.dw_layout != p.dw_layout) return MMS_ERR_ARG;
    }
    if (conv3w_mt_ok(((p.M + p.msplit - 1) / p.msplit + 31) & ~31, p.msplit, ng, o)) {
        constexpr int smem = (2 * C3W_STAGE + 2 * 1024) * (int)sizeof(float);        // 53.3 KB: 3 workgroups per CU
        Grp<Conv3BwdWP> a;
        if (!grp_fill(a, pp, ng, 9 * p.msplit)) return MMS_ERR_ARG;
        MMS_LAUNCH(conv3_bwdw_mt_kernel, dim3(1, 1, 9 * p.msplit * ng), dim3(256), smem, s, a);
        return mms_check_launch();
    }
    return launch_tile_gemm<Conv3BwdWOp>(pp, ng, dim3(1, 1, 27 * p.msplit), s);
}
MMS_SINGLE_O(mms_conv3_bwd_weight, Conv3BwdWP)
extern "C" int mms_conv3_bwd_weight_msplit(int M, int members, const MmsDnOpts* opts) {
    if (M <= 0 || members < 1 || members > MMS_MAX_GROUP) return 0;
    return mms_conv3w_msplit(M, members, mms_opts(opts));
}

// ------------------------------------------------------------------------------------------------------
// 1x1 conv backward.  Shared pieces: dy(m, n) with the output-side BN backward folded in, a(m, k) recompute.
// ------------------------------------------------------------------------------------------------------
struct DyConsts {   // per output channel n (LDS): dy = A*(dbn - B - yhat*C), yhat = (y - mean)*rstd
    float *A, *Bc, *Cc, *mean, *rstd;
    __device__ void init(const Conv1BwdP& p, float* e, int n0, int cnt, int tid) {
        A = e; Bc = e + cnt; Cc = e + 2 * cnt; mean = e + 3 * cnt; rstd = e + 4 * cnt;
        if (!p.has_bn_out) return;
        for (int i = tid; i < cnt; i += 256) {
            const int n = n0 + i;
            if (n < p.N) {
                float mu, rs, ga;
                double t1, t2;
                bn_bwd_consts(p.bn_out, p.bb_out, n, mu, rs, ga, t1, t2);      // five loads, one round trip
                A[i] = ga * rs;
                Bc[i] = (float)(t1 * (double)p.bn_out.inv_count);
                Cc[i] = (float)(t2 * (double)p.bn_out.inv_count);
                mean[i] = mu; rstd[i] = rs;
            } else {
                A[i] = Bc[i] = Cc[i] = mean[i] = rstd[i] = 0.f;
            }
        }
    }
    __device__ __forceinline__ float dy(const Conv1BwdP& p, float g, float y, int i) const {
        if (!p.has_bn_out) return g;
        return A[i] * (g - Bc[i] - (y - mean[i]) * rstd[i] * Cc[i]);
    }
};

// ---- data: dbn_in[m][k] = [a>0] * sum_n dy[m][n] * W[n][k]  (+ un-pool) ; rows m, cols k, reduce over n
template <int WM_, int WN_, int WK_, bool POOL, bool ONE_ = false>
struct Conv1BwdDataOp {
    typedef Conv1BwdP Params;
    static constexpr bool ONE_TILE = ONE_;       // N <= 32 WK: the whole reduction is one tile (dense layers: N = 128 = TK of <1, 1, 4>)
    static constexpr bool SINGLE_BUF = WK_ == 4;   // 128-deep tiles: one LDS buffer (tile_gemm.h)
    static constexpr int WM = WM_, WN = WN_, WK = WK_, AMODE = LD_K4, BMODE = LD_R4;
    __device__ void step(const Params&, int) {}
    static constexpr int TM = 32 * WM, TN = 32 * WN;
    // dy consts for all N (<=512 without bn_out, 128 with), bn_in consts for TN columns, srcbase[TM], fp64 scratch
    static constexpr int EXTRA = 5 * 128 + 4 * TN + TM + 1024;   // + fp64 [256/TN][2][TN] reduction scratch
    DyConsts dc;
    float* ein;
    const int* srcbase;
    int m0;
    __device__ void setup(const Params& p, int m0_, int n0, int, float* extra, int tid) {
        m0 = m0_;
        dc.init(p, extra, 0, 128, tid);
        ein = extra + 640;
        srcbase = (const int*)(extra + 640 + 4 * TN);
        if (tid < TN) {
            const int k = n0 + tid;
            float mu = 0, rs = 0, ga = 0, be = 0;
            if (k < p.K) bn_consts1(p.bn_in, k, mu, rs, ga, be);
            ein[tid] = mu; ein[TN + tid] = rs; ein[2 * TN + tid] = ga; ein[3 * TN + tid] = be;
        }
        if (POOL && tid < TM) {
            int m = m0 + tid, base = -1;
            if (m < p.M) {
                int D2 = p.in.D >> 1, H2 = p.in.H >> 1, W2 = p.in.W >> 1, vox2 = D2 * H2 * W2;
                int b = m / vox2, r = m % vox2, d = r / (H2 * W2), h = (r / W2) % H2, w = r % W2;
                base = ((b * p.in.D + 2 * d) * p.in.H + 2 * h) * p.in.W + 2 * w;
            }
            ((int*)extra)[640 + 4 * TN + tid] = base;
        }
    }
    __device__ void krange(const Params& p, int, int& kb, int& ke) { kb = 0; ke = p.N; }
    struct ARaw { float4 g, y; };
    typedef float4 BRaw;
    // branch-free loaders (see Conv1FwdOp): clamped addresses, out-of-range pieces zeroed in *_tx
    __device__ ARaw a_ld(const Params& p, int, int m, int n, bool& ok) const {   // A(m, n) = dy[m][n..n+3]
        ARaw r; r.y = Z4;
        ok = m < p.M && n < p.N;
        const size_t mm = m < p.M ? m : p.M - 1;
        const int nn = n < p.N ? n : p.N - 4;
        r.g = *(const float4*)(p.dyraw + mm * p.lddy + nn);
        if (p.has_bn_out) r.y = *(const float4*)(p.y + mm * p.ldy + nn);
        return r;
    }
    __device__ float4 a_tx(const Params& p, int, const ARaw& r, int, int n, bool ok) const {
        if (!ok) return Z4;
        if (!p.has_bn_out) return r.g;
        return make_float4(dc.dy(p, r.g.x, r.y.x, n), dc.dy(p, r.g.y, r.y.y, n + 1), dc.dy(p, r.g.z, r.y.z, n + 2), dc.dy(p, r.g.w, r.y.w, n + 3));
    }
    __device__ float4 b_ld(const Params& p, int, int k, int n, bool& ok) const {   // B(col k..k+3, n) = W[n][k..k+3]
        ok = n < p.N && k < p.K;
        return *(const float4*)(p.w + (size_t)(n < p.N ? n : p.N - 1) * p.K + (k < p.K ? k : p.K - 4));
    }
    __device__ float4 b_tx(const Params&, int, const float4& v, int, int, bool ok) const { return ok ? v : Z4; }
    __device__ void epilogue(const Params& p, int m0_, int n0, int, const float* Cs, int tid, bool active) {
        constexpr int RG = 256 / TN;              // row groups
        const int c = tid % TN, rg = tid / TN, k = n0 + c;
        const float mu = ein[c], rs = ein[TN + c], ga = ein[2 * TN + c], be = ein[3 * TN + c];
        double s1 = 0, s2 = 0;
        const int rows = !active ? 0 : (p.M - m0_ < TM ? p.M - m0_ : TM);
        constexpr int NI = TM / RG;               // rows per thread
        float xv[NI], gv[NI];                     // (non-pool) this thread's x values and masked gradients, kept for the fused apply
        if (!POOL) {
            // every global load of the pass is issued before its first use: ONE memory round trip for the thread's NI rows
            // (as a `for (r = rg; r < rows; r += RG)` loop the loads ran one after the other: 16 dependent trips at 128 rows)
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int r = rg + i * RG;
                xv[i] = (r < rows && k < p.K) ? p.x[(size_t)(m0_ + r) * p.ldx + k] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int r = rg + i * RG;
                const bool ok = r < rows && k < p.K;
                const float xh = (xv[i] - mu) * rs;
                const float g = ok && fmaf(ga, xh, be) > 0.f ? Cs[r * (TN + 1) + c] : 0.f;
                gv[i] = g;
                if (ok && !p.fuse_dx) p.dbn[(size_t)(m0_ + r) * p.lddbn + k] = g;
                if (ok) { s1 += g; s2 += (double)g * xh; }
            }
        } else if (k < p.K) {
            for (int r = rg; r < rows; r += RG) {
                const float da = Cs[r * (TN + 1) + c];
                {
                    const int base = srcbase[r], HW = p.in.H * p.in.W, W = p.in.W;
                    const float d8 = da * 0.125f;
#pragma unroll
                    for (int o = 0; o < 8; ++o) {
                        const size_t src = base + (o >> 2) * HW + ((o >> 1) & 1) * W + (o & 1);
                        const float xh = (p.x[src * p.ldx + k] - mu) * rs;
                        const float g = fmaf(ga, xh, be) > 0.f ? d8 : 0.f;
                        p.dbn[src * p.lddbn + k] = g;
                        s1 += g; s2 += (double)g * xh;
                    }
                }
            }
        }
        double* red = (double*)(ein + 4 * TN + TM);     // offset (640 + 4TN + TM)*4 bytes: multiple of 8
        if (active) { red[(rg * 2 + 0) * TN + c] = s1; red[(rg * 2 + 1) * TN + c] = s2; }
        __syncthreads();
        if (!POOL && p.fuse_dx) {
            // This workgroup holds ALL rows of its channels (launcher: M <= TM, one row tile): the BN1-backward sums are complete
            // right here, so norm1's backward is applied in place -- dx[:, k] (+)= gamma*rstd*(g - s1/M - xhat*s2/M), dgamma, dbeta --
            // without the dbn scratch, the atomics and the separate mms_bn_bwd_apply launch (same arithmetic as that kernel).
            if (k < p.K && active) {
                double a = 0, b = 0;
                for (int g = 0; g < RG; ++g) { a += red[(g * 2) * TN + c]; b += red[(g * 2 + 1) * TN + c]; }
                const float gr = ga * rs, m1 = (float)(a * (double)p.bn_in.inv_count), m2 = rs * (float)(b * (double)p.bn_in.inv_count);
                float ov[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) {            // the read half of the read-modify-write, all rows in flight together
                    const int r = rg + i * RG;
                    ov[i] = (r < rows && p.fuse_accumulate) ? p.fuse_dx[(size_t)(m0_ + r) * p.fuse_lddx + k] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int r = rg + i * RG;
                    if (r < rows) p.fuse_dx[(size_t)(m0_ + r) * p.fuse_lddx + k] = ov[i] + gr * (gv[i] - m1 - (xv[i] - mu) * m2);
                }
                if (rg == 0 && p.fuse_dgamma) { p.fuse_dgamma[k] += (float)b; p.fuse_dbeta[k] += (float)a; }
            }
            return;
        }
        if (rg == 0 && k < p.K && active) {
            double a = 0, b = 0;
            for (int g = 0; g < RG; ++g) { a += red[(g * 2) * TN + c]; b += red[(g * 2 + 1) * TN + c]; }
            atomicAdd(&stat_rep(p.s1, p.srep, p.sstride)[k], a);
            atomicAdd(&stat_rep(p.s2, p.srep, p.sstride)[k], b);
        }
    }
};

static bool conv1_bwd_same(const Conv1BwdP* pp, int ng, bool same_k = true) {
    const Conv1BwdP& p = *pp;
    for (int g = 1; g < ng; ++g) {
        const Conv1BwdP& q = pp[g];
        if (q.M != p.M || q.N != p.N || (same_k && q.K != p.K) || q.K % 32 != 0 || q.ldx % 4 != 0 || q.lddy % 4 != 0 || q.pool != p.pool || q.has_bn_out != p.has_bn_out ||
            q.msplit != p.msplit || q.in.D != p.in.D || q.in.H != p.in.H || q.in.W != p.in.W) return false;
    }
    return true;
}
extern "C" int mms_conv1_bwd_data_group(const Conv1BwdP* pp, int ng, const MmsDnOpts* opts, hipStream_t s) {
    if (!pp || ng < 1 || ng > MMS_MAX_GROUP) return MMS_ERR_ARG;
    const MmsDnOpts o = mms_opts(opts);
    const Conv1BwdP& p = *pp;
    if (p.M <= 0 || p.K % 32 != 0 || p.N % 32 != 0 || p.ldx % 4 != 0 || p.lddy % 4 != 0) return MMS_ERR_ARG;
    if (p.has_bn_out && p.N != 128) return MMS_ERR_ARG;
    if (!conv1_bwd_same(pp, ng)) return MMS_ERR_ARG;
    for (int g = 0; g < ng; ++g) if ((pp[g].fuse_dx != nullptr) != (p.fuse_dx != nullptr)) return MMS_ERR_ARG;
    if (p.fuse_dx) {      // norm1 backward fused into the epilogue: one workgroup must own every row of its 32 channels
        if (p.pool || p.M > 128 || p.fuse_lddx % 4 != 0) return MMS_ERR_ARG;
        bool small = true;
        for (int g = 0; g < ng; ++g) small = small && mms_conv1_small_bwd_ok(pp[g], o);
        if (small) return mms_c1s_bwd(pp, ng, s);
        dim3 g(1, (p.K + 31) / 32, 1);
        return p.M <= 32 ? launch_tile_gemm<Conv1BwdDataOp<1, 1, 4, false>>(pp, ng, g, s)
                         : launch_tile_gemm<Conv1BwdDataOp<4, 1, 1, false>>(pp, ng, g, s);
    }
    // tile shape from the whole group's work (MmsDnOpts.big_ng = -1: from one model's -- tests that need ng-independent arithmetic)
    const bool big = (long)p.M * p.K * (o.big_ng < 0 ? 1 : ng) >= 256L * 64 * 64;
    if (big) {
        dim3 g((p.M + 63) / 64, (p.K + 63) / 64, 1);
        return p.pool ? launch_tile_gemm<Conv1BwdDataOp<2, 2, 1, true>>(pp, ng, g, s)
                      : launch_tile_gemm<Conv1BwdDataOp<2, 2, 1, false>>(pp, ng, g, s);
    }
    dim3 g((p.M + 31) / 32, (p.K + 31) / 32, 1);
    if (!p.pool && p.N <= 128) return launch_tile_gemm<Conv1BwdDataOp<1, 1, 4, false, true>>(pp, ng, g, s);     // one K tile: single LDS buffer
    return p.pool ? launch_tile_gemm<Conv1BwdDataOp<1, 1, 4, true>>(pp, ng, g, s)
                  : launch_tile_gemm<Conv1BwdDataOp<1, 1, 4, false>>(pp, ng, g, s);
}
MMS_SINGLE_O(mms_conv1_bwd_data, Conv1BwdP)

// ---- weight: dW[n][k] += sum_m dy[m][n] * a[m][k] ; rows n, cols k, reduce over m (split over grid.z)
template <bool POOL>
struct Conv1BwdWOp {
    typedef Conv1BwdP Params;
    static constexpr int WM = 2, WN = 2, WK = 1, AMODE = LD_R4, BMODE = LD_R4;
    __device__ void step(const Params&, int) {}
    static constexpr int TM = 64, TN = 64;
    static constexpr int EXTRA = 5 * 64 + 3 * 64;
    DyConsts dc;
    const float *mean, *sc, *beta;
    int n0r, k0c, mb, me;
    __device__ void setup(const Params& p, int m0_, int n0, int z, float* extra, int tid) {
        n0r = m0_; k0c = n0;
        dc.init(p, extra, n0r, 64, tid);
        mean = extra + 320; sc = extra + 384; beta = extra + 448;
        if (tid < 64) {
            const int k = k0c + tid;
            float mu = 0, rs = 0, ga = 0, be = 0;
            if (k < p.K) bn_consts1(p.bn_in, k, mu, rs, ga, be);
            extra[320 + tid] = mu; extra[384 + tid] = ga * rs; extra[448 + tid] = be;
        }
        int mc = (p.M + p.msplit - 1) / p.msplit;
        mc = (mc + 31) & ~31;
        mb = z * mc;
        me = mb + mc < p.M ? mb + mc : p.M;
        // BN2 parameter grads: dgamma = sum dbn*xhat, dbeta = sum dbn
        if (p.has_bn_out && p.dgamma_out && blockIdx.y == 0 && z == 0 && tid < 64 && n0r + tid < p.N) {
            p.dgamma_out[n0r + tid] += (float)rep_sum(p.bb_out.s2, n0r + tid, p.bb_out.nrep, p.bb_out.rep_stride);
            p.dbeta_out[n0r + tid] += (float)rep_sum(p.bb_out.s1, n0r + tid, p.bb_out.nrep, p.bb_out.rep_stride);
        }
    }
    __device__ void krange(const Params& p, int, int& kb, int& ke) {
        kb = mb; ke = me;
        if (k0c >= p.K) ke = kb;        // members of one launch may differ in K (layers of a dense block): nothing to do here
    }
    struct ARaw { float4 g, y; };
    typedef float4 BRaw;
    __device__ ARaw a_ld(const Params& p, int, int n, int m, bool& ok) const {   // A(row n..n+3, m) = dy[m][n..n+3]; branch-free (see Conv1FwdOp)
        ARaw r; r.y = Z4;
        ok = m < me && n < p.N;
        const size_t mm = m < me ? m : me - 1;        // (the K loop only runs when mb < me)
        const int nn = n < p.N ? n : p.N - 4;
        r.g = *(const float4*)(p.dyraw + mm * p.lddy + nn);
        if (p.has_bn_out) r.y = *(const float4*)(p.y + mm * p.ldy + nn);
        return r;
    }
    __device__ float4 a_tx(const Params& p, int, const ARaw& r, int n, int, bool ok) const {
        if (!ok) return Z4;
        if (!p.has_bn_out) return r.g;
        const int i = n - n0r;
        return make_float4(dc.dy(p, r.g.x, r.y.x, i), dc.dy(p, r.g.y, r.y.y, i + 1), dc.dy(p, r.g.z, r.y.z, i + 2), dc.dy(p, r.g.w, r.y.w, i + 3));
    }
    __device__ float4 act4(const float4 v, int i) const {
        return make_float4(fmaxf(bn_apply(v.x, mean[i], sc[i], beta[i]), 0.f), fmaxf(bn_apply(v.y, mean[i + 1], sc[i + 1], beta[i + 1]), 0.f),
                           fmaxf(bn_apply(v.z, mean[i + 2], sc[i + 2], beta[i + 2]), 0.f), fmaxf(bn_apply(v.w, mean[i + 3], sc[i + 3], beta[i + 3]), 0.f));
    }
    __device__ float4 b_ld(const Params& p, int, int k, int m, bool& ok) const {   // B(col k..k+3, m) = a[m][k..k+3]
        ok = m < me && k < p.K;
        if (!POOL) return *(const float4*)(p.x + (size_t)(m < me ? m : me - 1) * p.ldx + (k < p.K ? k : p.K - 4));
        if (!ok) return Z4;
        const int i = k - k0c;
        const int D2 = p.in.D >> 1, H2 = p.in.H >> 1, W2 = p.in.W >> 1, vox2 = D2 * H2 * W2;
        const int b = m / vox2, r = m % vox2, d = r / (H2 * W2), h = (r / W2) % H2, w = r % W2;
        const int base = ((b * p.in.D + 2 * d) * p.in.H + 2 * h) * p.in.W + 2 * w, HW = p.in.H * p.in.W, W = p.in.W;
        float4 s = Z4;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const size_t src = base + (o >> 2) * HW + ((o >> 1) & 1) * W + (o & 1);
            const float4 v = act4(*(const float4*)(p.x + src * p.ldx + k), i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        return make_float4(s.x * 0.125f, s.y * 0.125f, s.z * 0.125f, s.w * 0.125f);
    }
    __device__ float4 b_tx(const Params& p, int, const float4& v, int k, int, bool ok) const {
        if (POOL) return v;
        if (!ok) return Z4;
        return act4(v, k - k0c);
    }
    __device__ void epilogue(const Params& p, int, int, int, const float* Cs, int tid, bool active) {
        if (mb >= me || !active || k0c >= p.K) return;
        for (int idx = tid; idx < TM * TN; idx += 256) {
            const int r = idx / TN, c = idx % TN, n = n0r + r, k = k0c + c;
            if (n < p.N && k < p.K) atomicAdd(&p.dw[(size_t)n * p.K + k], Cs[r * (TN + 1) + c]);
        }
    }
};

extern "C" int mms_conv1_bwd_weight_group(const Conv1BwdP* pp, int ng, hipStream_t s) {
    if (!pp || ng < 1 || ng > MMS_MAX_GROUP) return MMS_ERR_ARG;
    const Conv1BwdP& p = *pp;
    if (p.M <= 0 || p.msplit <= 0 || p.K % 32 != 0 || p.N % 32 != 0) return MMS_ERR_ARG;
    if (!conv1_bwd_same(pp, ng, false)) return MMS_ERR_ARG;
    int kmax = p.K;                     // members may differ in K (input channels): the grid covers the widest, the others' surplus
    for (int g = 1; g < ng; ++g) if (pp[g].K > kmax) kmax = pp[g].K;      // workgroups return at once (Conv1BwdWOp::krange)
    dim3 g((p.N + 63) / 64, (kmax + 63) / 64, p.msplit);
    return p.pool ? launch_tile_gemm<Conv1BwdWOp<true>>(pp, ng, g, s) : launch_tile_gemm<Conv1BwdWOp<false>>(pp, ng, g, s);
}
MMS_SINGLE(mms_conv1_bwd_weight, Conv1BwdP)

// ------------------------------------------------------------------------------------------------------
// BN backward apply into the gradient slab: dx[:, 0:C] (+)= g*rstd*(dbn - s1/M - xhat*s2/M)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const Grp<BnBwdApplyP> grp, const int rows) {
    const BnBwdApplyP& p = grp.p[blockIdx.z];
    // workgroup = `rows` rows x one chunk of <= 256 channels.  The per-channel constants (fp64 replica sums of four
    // accumulators) are computed ONCE per workgroup, one channel per thread, and shared through LDS; the threads then map
    // densely onto (channel quad, row): a 64-channel layer keeps all 256 lanes busy (16 quads x 16 rows).
    __shared__ float cm[4][256];            // mean | gamma*rstd | m1 | rstd*m2
    const int tid = threadIdx.x, c0 = blockIdx.y * 256;
    const int nc = p.C - c0 < 256 ? p.C - c0 : 256;          // channels of this chunk (multiple of 4)
    if (tid < nc) {
        const int c = c0 + tid;
        float mu, rs, ga;
        double t1, t2;
        bn_bwd_consts(p.bn, p.bb, c, mu, rs, ga, t1, t2);                       // five loads, one round trip
        cm[0][tid] = mu; cm[1][tid] = ga * rs;
        cm[2][tid] = (float)(t1 * (double)p.bn.inv_count);
        cm[3][tid] = rs * (float)(t2 * (double)p.bn.inv_count);
        if (blockIdx.x == 0 && p.dgamma) {
            p.dgamma[c] += (float)t2;
            p.dbeta[c] += (float)t1;
        }
    }
    __syncthreads();
    const int nq = nc >> 2, rpar = 256 / nq;                  // row lanes per pass
    const int q = tid % nq, rl = tid / nq;
    if (rl >= rpar) return;
    const int c = 4 * q;
    const float4 mu = *(const float4*)&cm[0][c], gr = *(const float4*)&cm[1][c], m1 = *(const float4*)&cm[2][c], m2 = *(const float4*)&cm[3][c];
    const int r0 = blockIdx.x * rows, rend = r0 + rows < p.M ? r0 + rows : p.M;
    // rows of this thread: r0 + rl + j * rpar; processed four at a time with all twelve loads issued before the first use
    for (int rb = r0 + rl; rb < rend; rb += 4 * rpar) {
        float4 g[4], x[4], o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = rb + j * rpar;
            const size_t m = r < rend ? r : rb;
            g[j] = *(const float4*)(p.dbn + m * p.lddbn + c0 + c);
            x[j] = *(const float4*)(p.x + m * p.ldx + c0 + c);
            o[j] = p.accumulate ? *(const float4*)(p.dx + m * p.lddx + c0 + c) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = rb + j * rpar;
            if (r < rend) {
                float4 v = o[j];
                v.x += gr.x * (g[j].x - m1.x - (x[j].x - mu.x) * m2.x);
                v.y += gr.y * (g[j].y - m1.y - (x[j].y - mu.y) * m2.y);
                v.z += gr.z * (g[j].z - m1.z - (x[j].z - mu.z) * m2.z);
                v.w += gr.w * (g[j].w - m1.w - (x[j].w - mu.w) * m2.w);
                *(float4*)(p.dx + (size_t)r * p.lddx + c0 + c) = v;
            }
        }
    }
}

extern "C" int mms_bn_bwd_apply_group(const BnBwdApplyP* pp, int ng, hipStream_t s) {
    Grp<BnBwdApplyP> a;
    if (!grp_fill(a, pp, ng, 1)) return MMS_ERR_ARG;
    const BnBwdApplyP& p = *pp;
    if (p.M <= 0 || p.C <= 0 || p.C % 4 != 0 || p.lddbn % 4 || p.ldx % 4 || p.lddx % 4) return MMS_ERR_ARG;
    for (int g = 1; g < ng; ++g) {
        const BnBwdApplyP& q = pp[g];
        if (q.M != p.M || q.C != p.C || q.lddbn % 4 || q.ldx % 4 || q.lddx % 4 || q.accumulate != p.accumulate) return MMS_ERR_ARG;
    }
    const int rows = (long)p.M * ng >= 16384 ? 128 : 32;     // big launches amortise the constants over more rows; small ones keep their workgroup count
    MMS_LAUNCH(bn_bwd_apply_kernel, dim3((p.M + rows - 1) / rows, (p.C + 255) / 256, ng), dim3(256), 0, s, a, rows);
    return mms_check_launch();
}
MMS_SINGLE(mms_bn_bwd_apply, BnBwdApplyP)

// ------------------------------------------------------------------------------------------------------
// head backward: Linear(C,N) + global-avg-pool + relu + norm5.  M = B*V rows is small: one thread per channel
// walks all rows, so the BN sums need no atomics.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_bwd_feat_kernel(const Grp<HeadBwdP> grp) {
    // workgroup = 64 channels x 4 sample lanes: thread (c, bl) does the samples b = bl, bl + 4, ... of channel c (the Linear
    // backward dot over N outputs, the ReLU mask, the BN sums of its rows); the four partial sums per channel meet in LDS.
    // (One thread per channel walking every sample was a 512-load serial chain on 4 workgroups: 101 us per launch.)
    const HeadBwdP& p = grp.p[blockIdx.z];
    __shared__ double red[2][4][64];
    const int cl = threadIdx.x & 63, bl = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    const bool ok = c < p.C;
    float mu = 0.f, rs = 0.f, ga = 0.f, be = 0.f;
    if (ok) bn_consts1(p.bn, c, mu, rs, ga, be);
    const int M = p.B * p.V;
    const float invV = 1.f / (float)p.V;
    double s1 = 0, s2 = 0;
    if (ok)
        for (int b = bl; b < p.B; b += 4) {
            float dp = 0;
#pragma unroll 16
            for (int n = 0; n < p.N; ++n) dp = fmaf(p.dout[b * p.lddout + n], p.w[(size_t)n * p.C + c], dp);      // (16 weight loads in flight)
            dp *= invV;
            for (int v = 0; v < p.V; ++v) {
                const size_t m = (size_t)b * p.V + v;
                const float xh = (p.slab[m * p.ld + c] - mu) * rs;
                const float g = fmaf(ga, xh, be) > 0.f ? dp : 0.f;
                p.dslab[m * p.ldd + c] = g;            // stash dbn; finalised below by the same thread
                s1 += g; s2 += (double)g * xh;
            }
        }
    red[0][bl][cl] = s1; red[1][bl][cl] = s2;
    __syncthreads();
    if (!ok) return;
    const double t1 = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
    const double t2 = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
    const float m1 = (float)(t1 / M), m2 = (float)(t2 / M);
    for (int b = bl; b < p.B; b += 4)
        for (int v = 0; v < p.V; ++v) {
            const size_t m = (size_t)b * p.V + v;
            const float xh = (p.slab[m * p.ld + c] - mu) * rs;
            const float g = p.dslab[m * p.ldd + c];
            p.dslab[m * p.ldd + c] = ga * rs * (g - m1 - xh * m2);
        }
    if (bl == 0) {
        p.dgamma[c] += (float)t2;
        p.dbeta[c] += (float)t1;
    }
}
__global__ __launch_bounds__(256) void head_bwd_w_kernel(const Grp<HeadBwdP> grp) {
    const HeadBwdP& p = grp.p[blockIdx.z];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.N * p.C) return;
    const int n = idx / p.C, c = idx % p.C;
    float a = 0;
    for (int b = 0; b < p.B; ++b) a = fmaf(p.dout[b * p.lddout + n], p.pooled[b * p.C + c], a);
    p.dw[idx] += a;
    if (c == 0) {
        float d = 0;
        for (int b = 0; b < p.B; ++b) d += p.dout[b * p.lddout + n];
        p.dbias[n] += d;
    }
}
extern "C" int mms_head_bwd_group(const HeadBwdP* pp, int ng, hipStream_t s) {
    Grp<HeadBwdP> a;
    if (!grp_fill(a, pp, ng, 1)) return MMS_ERR_ARG;
    const HeadBwdP& p = *pp;
    if (p.B <= 0) return MMS_ERR_ARG;
    for (int g = 1; g < ng; ++g) {
        const HeadBwdP& q = pp[g];
        if (q.B != p.B || q.C != p.C || q.N != p.N || q.V != p.V) return MMS_ERR_ARG;
    }
    MMS_LAUNCH(head_bwd_feat_kernel, dim3((p.C + 63) / 64, 1, ng), dim3(256), 0, s, a);
    MMS_LAUNCH(head_bwd_w_kernel, dim3((p.N * p.C + 255) / 256, 1, ng), dim3(256), 0, s, a);
    return mms_check_launch();
}
MMS_SINGLE(mms_head_bwd, HeadBwdP)

// SyncBN split of the head backward: the two norm5-backward sums leave the kernel (ext_sums, all-reduced over the ranks by the
// caller between the two launches); p.bn carries the GLOBAL statistics and count.  One thread per channel; not a hot path.
__global__ __launch_bounds__(256) void head_bwd_sums_kernel(const HeadBwdP p) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= p.C) return;
    float mu, rs, ga, be;
    bn_consts1(p.bn, c, mu, rs, ga, be);
    const float invV = 1.f / (float)p.V;
    double s1 = 0, s2 = 0;
    for (int b = 0; b < p.B; ++b) {
        float dp = 0;
        for (int n = 0; n < p.N; ++n) dp = fmaf(p.dout[b * p.lddout + n], p.w[(size_t)n * p.C + c], dp);
        dp *= invV;
        for (int v = 0; v < p.V; ++v) {
            const size_t m = (size_t)b * p.V + v;
            const float xh = (p.slab[m * p.ld + c] - mu) * rs;
            const float g = fmaf(ga, xh, be) > 0.f ? dp : 0.f;
            p.dslab[m * p.ldd + c] = g;
            s1 += g; s2 += (double)g * xh;
        }
    }
    p.ext_sums[c] = s1; p.ext_sums[p.C + c] = s2;
}
__global__ __launch_bounds__(256) void head_bwd_apply_kernel(const HeadBwdP p) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= p.C) return;
    float mu, rs;
    bn_mean_rstd(p.bn, c, mu, rs);
    const float ga = p.bn.gamma[c];
    const double t1 = p.ext_sums[c], t2 = p.ext_sums[p.C + c];
    const float m1 = (float)(t1 * (double)p.bn.inv_count), m2 = (float)(t2 * (double)p.bn.inv_count);
    for (int m = 0; m < p.B * p.V; ++m) {
        const float xh = (p.slab[(size_t)m * p.ld + c] - mu) * rs;
        const float g = p.dslab[(size_t)m * p.ldd + c];
        p.dslab[(size_t)m * p.ldd + c] = ga * rs * (g - m1 - xh * m2);
    }
    // (like every BatchNorm parameter gradient under SyncBN these are the GLOBAL sums on every rank: the caller scales them by
    // 1/ranks before its SUM all-reduce of the gradients)
    p.dgamma[c] += (float)t2;
    p.dbeta[c] += (float)t1;
}
extern "C" int mms_head_bwd_sums(const HeadBwdP* pp, hipStream_t s) {
    if (!pp || pp->B <= 0 || !pp->ext_sums) return MMS_ERR_ARG;
    MMS_LAUNCH(head_bwd_sums_kernel, dim3((pp->C + 255) / 256), dim3(256), 0, s, *pp);
    return mms_check_launch();
}
extern "C" int mms_head_bwd_apply(const HeadBwdP* pp, hipStream_t s) {
    if (!pp || pp->B <= 0 || !pp->ext_sums) return MMS_ERR_ARG;
    Grp<HeadBwdP> a;
    if (!grp_fill(a, pp, 1, 1)) return MMS_ERR_ARG;
    MMS_LAUNCH(head_bwd_apply_kernel, dim3((pp->C + 255) / 256), dim3(256), 0, s, *pp);
    MMS_LAUNCH(head_bwd_w_kernel, dim3((pp->N * pp->C + 255) / 256, 1, 1), dim3(256), 0, s, a);
    return mms_check_launch();
}

// ------------------------------------------------------------------------------------------------------
// maxpool(3,2,1) backward (gather form, no atomics on the gradient) + relu0 mask -> dbn0, BN0 sums
// one workgroup = 256 conv0-grid voxels x 64 channels
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_bwd_kernel(const Grp<PoolBwdP> grp) {
    const PoolBwdP& p = grp.p[blockIdx.z];
    // workgroup = 64 conv0-grid voxels x 64 channels.  Each input voxel is covered by 1 or 2 windows per axis
    // (od in {id>>1, (id+1)>>1}; they coincide for even id): all 8 candidates are loaded unconditionally from clamped
    // addresses and selected afterwards (the duplicate candidate of an even coordinate is masked out).
    __shared__ double red[2][4][64];
    const int c = threadIdx.x & 63, vr = threadIdx.x >> 6;
    float mu, rs, ga, be;
    bn_consts1(p.bn, c, mu, rs, ga, be);
    const int vox_in = p.in.D * p.in.H * p.in.W, Min = p.B * vox_in, vox_out = p.out.D * p.out.H * p.out.W;
    double s1 = 0, s2 = 0;
    for (int it = 0; it < 16; ++it) {
        const int m = blockIdx.x * 64 + it * 4 + vr;
        if (m >= Min) break;
        int b, id, ih, iw;
        if (p.coords) {
            unpack_dhw(p.coords[m], id, ih, iw);
            b = (m - ((id * p.in.H + ih) * p.in.W + iw)) / vox_in;
        } else {
            const int r = m % vox_in;
            b = m / vox_in; id = r / (p.in.H * p.in.W); ih = (r / p.in.W) % p.in.H; iw = r % p.in.W;
        }
        const float y = p.y0[(size_t)m * 64 + c];
        uint8_t am[8];
        float gv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int od = min((id + (k >> 2)) >> 1, p.out.D - 1), oh = min((ih + ((k >> 1) & 1)) >> 1, p.out.H - 1),
                      ow = min((iw + (k & 1)) >> 1, p.out.W - 1);
            const size_t mo = (size_t)b * vox_out + ((size_t)od * p.out.H + oh) * p.out.W + ow;
            am[k] = p.argmax[mo * 64 + c];
            gv[k] = p.dslab[mo * p.ld + c];
        }
        float g = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int kd = k >> 2, kh = (k >> 1) & 1, kw = k & 1;
            const int od = (id + kd) >> 1, oh = (ih + kh) >> 1, ow = (iw + kw) >> 1;
            // second candidate of an axis exists only for odd coordinates and inside the pooled grid
            const bool ok = (kd == 0 || (id & 1)) && (kh == 0 || (ih & 1)) && (kw == 0 || (iw & 1)) &&
                            od < p.out.D && oh < p.out.H && ow < p.out.W;
            const int tap = ((id - 2 * od + 1) * 3 + (ih - 2 * oh + 1)) * 3 + (iw - 2 * ow + 1);
            g += (ok && am[k] == tap) ? gv[k] : 0.f;
        }
        const float xh = (y - mu) * rs;
        g = fmaf(ga, xh, be) > 0.f ? g : 0.f;
        p.dbn[(size_t)m * 64 + c] = g;
        s1 += g; s2 += (double)g * xh;
    }
    red[0][vr][c] = s1; red[1][vr][c] = s2;
    __syncthreads();
    if (vr == 0) {
        atomicAdd(&stat_rep(p.s1, p.srep, p.sstride)[c], red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
        atomicAdd(&stat_rep(p.s2, p.srep, p.sstride)[c], red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
    }
}
// Brick form (default when the conv0 grid is a multiple of 2 x 4 x WT): a workgroup owns a 2 x 4 x WT brick of conv0-grid
// voxels; the <= 2 x 3 x (WT/2+1) pooled voxels whose windows can reach it are staged in LDS once (argmax byte + gradient,
// coalesced 64-channel rows), so a voxel's 8 candidates cost 16 LDS reads instead of 16 global loads (8 of them 1-byte).
template <int WT>
__global__ __launch_bounds__(256) void pool_bwd_brick_kernel(const Grp<PoolBwdP> grp) {
    const PoolBwdP& p = grp.p[blockIdx.z];
    constexpr int WO = WT / 2 + 1, NC = 2 * 3 * WO;
    __shared__ float cg[NC][64];
    __shared__ uint8_t ca[NC][64];
    __shared__ double red[2][4][64];
    const int c = threadIdx.x & 63, vr = threadIdx.x >> 6;
    const int Di = p.in.D, Hi = p.in.H, Wi = p.in.W, Do = p.out.D, Ho = p.out.H, Wo = p.out.W;
    const int nwt = Wi / WT, nhb = Hi / 4, ndb = Di / 2;
    int r = blockIdx.x;
    const int wt = r % nwt; r /= nwt;
    const int hb = r % nhb; r /= nhb;
    const int db = r % ndb, b = r / ndb;
    const int w0 = wt * WT, ow0 = w0 >> 1;
    for (int cand = vr; cand < NC; cand += 4) {
        const int dd = cand / (3 * WO), hh = (cand / WO) % 3, ww = cand % WO;
        const int od = db + dd, oh = 2 * hb + hh, ow = ow0 + ww;
        const bool ok = od < Do && oh < Ho && ow < Wo;
        const size_t mo = ok ? ((size_t)(b * Do + od) * Ho + oh) * Wo + ow : 0;
        cg[cand][c] = ok ? p.dslab[mo * p.ld + c] : 0.f;
        ca[cand][c] = ok ? p.argmax[mo * 64 + c] : (uint8_t)255;
    }
    float mu, rs, ga, be;
    bn_consts1(p.bn, c, mu, rs, ga, be);
    __syncthreads();
    double s1 = 0, s2 = 0;
    for (int v = vr; v < 8 * WT; v += 4) {
        const int di = v / (4 * WT), hi = (v / WT) & 3, wi = v % WT;
        const int id = 2 * db + di, ih = 4 * hb + hi, iw = w0 + wi;
        const size_t m = ((size_t)(b * Di + id) * Hi + ih) * Wi + iw;
        const float y = p.y0[m * 64 + c];
        float g = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int kd = k >> 2, kh = (k >> 1) & 1, kw = k & 1;
            const int dd = (di + kd) >> 1, hh = (hi + kh) >> 1, ww = (wi + kw) >> 1;
            const int od = db + dd, oh = 2 * hb + hh, ow = ow0 + ww;
            const bool ok = (kd == 0 || (id & 1)) && (kh == 0 || (ih & 1)) && (kw == 0 || (iw & 1));
            const int tap = ((id - 2 * od + 1) * 3 + (ih - 2 * oh + 1)) * 3 + (iw - 2 * ow + 1);
            const int cand = (dd * 3 + hh) * WO + ww;
            g += (ok && ca[cand][c] == tap) ? cg[cand][c] : 0.f;
        }
        const float xh = (y - mu) * rs;
        g = fmaf(ga, xh, be) > 0.f ? g : 0.f;
        p.dbn[m * 64 + c] = g;
        s1 += g; s2 += (double)g * xh;
    }
    red[0][vr][c] = s1; red[1][vr][c] = s2;
    __syncthreads();
    if (vr == 0) {
        atomicAdd(&stat_rep(p.s1, p.srep, p.sstride)[c], red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
        atomicAdd(&stat_rep(p.s2, p.srep, p.sstride)[c], red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
    }
}

extern "C" int mms_pool_bwd_group(const PoolBwdP* pp, int ng, hipStream_t s) {
    Grp<PoolBwdP> a;
    if (!grp_fill(a, pp, ng, 1)) return MMS_ERR_ARG;
    const PoolBwdP& p = *pp;
    const int Min = p.B * p.in.D * p.in.H * p.in.W;
    if (Min <= 0) return MMS_ERR_ARG;
    for (int g = 1; g < ng; ++g) {
        const PoolBwdP& q = pp[g];
        if (q.B != p.B || q.in.D != p.in.D || q.in.H != p.in.H || q.in.W != p.in.W || q.out.D != p.out.D || q.out.H != p.out.H ||
            q.out.W != p.out.W) return MMS_ERR_ARG;
    }
    if (p.in.D % 2 == 0 && p.in.H % 4 == 0 && p.in.W % 16 == 0) {
        if (p.in.W % 32 == 0) {
            MMS_LAUNCH(pool_bwd_brick_kernel<32>, dim3(p.B * (p.in.D / 2) * (p.in.H / 4) * (p.in.W / 32), 1, ng), dim3(256), 0, s, a);
        } else {
            MMS_LAUNCH(pool_bwd_brick_kernel<16>, dim3(p.B * (p.in.D / 2) * (p.in.H / 4) * (p.in.W / 16), 1, ng), dim3(256), 0, s, a);
        }
        return mms_check_launch();
    }
    MMS_LAUNCH(pool_bwd_kernel, dim3((Min + 63) / 64, 1, ng), dim3(256), 0, s, a);
    return mms_check_launch();
}
MMS_SINGLE(mms_pool_bwd, PoolBwdP)

// ------------------------------------------------------------------------------------------------------
// conv0 backward-weight: dW0[n][k] += sum_m dy0[m][n] * x[patch(m,k)], dy0 = BN0-backward(dbn0)
// tile: rows = tap k (343), cols = n (64), reduction over output voxels m (split over grid.z)
// ------------------------------------------------------------------------------------------------------
struct Conv0BwdWOp {
    typedef Conv0BwdWP Params;
    static constexpr int WM = 2, WN = 2, WK = 1, AMODE = LD_K1, BMODE = LD_R4;
    __device__ void step(const Params&, int) {}
    static constexpr int TM = 64, TN = 64;
    static constexpr int EXTRA = 5 * 64;
    float *A, *Bc, *Cc, *mean, *rstd;
    int k0r, mb, me;
    __device__ void setup(const Params& p, int m0_, int, int z, float* extra, int tid) {
        k0r = m0_;
        A = extra; Bc = extra + 64; Cc = extra + 128; mean = extra + 192; rstd = extra + 256;
        if (tid < 64) {
            float mu, rs, ga_;
            double t1, t2;
            bn_bwd_consts(p.bn, p.bb, tid, mu, rs, ga_, t1, t2);
            A[tid] = ga_ * rs;
            Bc[tid] = (float)(t1 * (double)p.bn.inv_count);
            Cc[tid] = (float)(t2 * (double)p.bn.inv_count);
            mean[tid] = mu; rstd[tid] = rs;
            if (blockIdx.x == 0 && z == 0 && p.dgamma) {
                p.dgamma[tid] += (float)t2;
                p.dbeta[tid] += (float)t1;
            }
        }
        int mc = (p.M + p.msplit - 1) / p.msplit;
        mc = (mc + 31) & ~31;
        mb = z * mc;
        me = mb + mc < p.M ? mb + mc : p.M;
    }
    __device__ void krange(const Params&, int, int& kb, int& ke) { kb = mb; ke = me; }
    typedef float ARaw;
    struct BRaw { float4 g, y; };
    __device__ float a_ld(const Params& p, int, int k, int m, bool& ok) const {    // A(row = tap k, m) = x[patch(m, k)]
        ok = true;
        if (k >= 343 || m >= me) return 0.f;
        int od, oh, ow;
        unpack_dhw(p.coords[m], od, oh, ow);
        const int b = m / (p.out.D * p.out.H * p.out.W);
        const int id = 2 * od - 3 + k / 49, ih = 2 * oh - 3 + (k / 7) % 7, iw = 2 * ow - 3 + k % 7;
        if ((unsigned)id >= (unsigned)p.in.D || (unsigned)ih >= (unsigned)p.in.H || (unsigned)iw >= (unsigned)p.in.W)
            return 0.f;
        return p.x[((size_t)(b * p.in.D + id) * p.in.H + ih) * p.in.W + iw];
    }
    __device__ float a_tx(const Params&, int, float v, int, int, bool) const { return v; }
    __device__ __forceinline__ float dy(float g, float y, int n) const {
        return A[n] * (g - Bc[n] - (y - mean[n]) * rstd[n] * Cc[n]);
    }
    __device__ BRaw b_ld(const Params& p, int, int n, int m, bool& ok) const {   // B(col n..n+3, m) = dy0[m][n..n+3]
        BRaw r; r.g = Z4; r.y = Z4;
        ok = m < me;
        if (!ok) return r;
        r.g = *(const float4*)(p.dbn + (size_t)m * 64 + n);
        r.y = *(const float4*)(p.y0 + (size_t)m * 64 + n);
        return r;
    }
    __device__ float4 b_tx(const Params&, int, const BRaw& r, int n, int, bool ok) const {
        if (!ok) return Z4;
        return make_float4(dy(r.g.x, r.y.x, n), dy(r.g.y, r.y.y, n + 1), dy(r.g.z, r.y.z, n + 2), dy(r.g.w, r.y.w, n + 3));
    }
    __device__ void epilogue(const Params& p, int, int, int, const float* Cs, int tid, bool active) {
        if (mb >= me || !active) return;
        for (int idx = tid; idx < TM * TN; idx += 256) {
            const int r = idx & 63, c = idx >> 6, k = k0r + r;     // r fastest: dW0[n][k] contiguous in k
            if (k < 343) atomicAdd(&p.dw[c * 343 + k], Cs[r * (TN + 1) + c]);
        }
    }
};
// ------------------------------------------------------------------------------------------------------
// conv0 backward-weight, LDS-staged form (the default; the tile-GEMM op above remains for grids that are not a
// multiple of the 2x4x4 box).  The GEMM form gathers x[patch(m, k)] with one scalar load per (voxel, tap) and re-reads
// dbn0/y0 once per 64-tap tile (6x); here a workgroup walks boxes of 2x4x4 = 32 output voxels, stages the box's input
// region (9 x 13 x 13 voxels of the single input channel) and its 32 x 64 dy0 values in LDS once, and feeds the MFMAs
// straight from LDS: A[tap][voxel] = region[koff(tap) + moff(voxel)] (one ds_read_b32 per operand, no im2col image),
// B[voxel][ch] = dy0.  Each wave owns 3 of the 11 tap tiles x both channel tiles (96 accumulator VGPRs), so the whole
// 343 x 64 gradient of the workgroup's voxel range lives in registers and dbn0/y0/x are read exactly once.
// Algorithmic bytes per output voxel: 64 ch x 4 B x 2 (dbn0, y0) + ~8 x 4 B of x = 544 B; FLOPs: 2 * 343 * 64.
// ------------------------------------------------------------------------------------------------------
#define C0_REG (9 * 13 * 13)        // 1521
#define C0_DYP 68
// Work distribution: the boxes of all models of the group form one pool that is dealt evenly over gridDim.x workgroups
// (a multiple of the CU count: the kernel is MFMA-bound, so an uneven deal costs its full imbalance); a workgroup whose
// range crosses a model boundary flushes its accumulators there (grp.zdim = number of models).
__global__ __launch_bounds__(256) void conv0_bwd_weight_kernel(const Grp<Conv0BwdWP> grp) {
    __shared__ float xs[2][C0_REG + 7];
    __shared__ float dys[2][32 * C0_DYP];
    __shared__ float cst[5 * 64];
    __shared__ float Cs[4][32 * 65];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kq = lane >> 5;
    float *cA = cst, *cB = cst + 64, *cC = cst + 128, *cM = cst + 192, *cR = cst + 256;
    const int D0 = grp.p[0].out.D, H0 = grp.p[0].out.H, W0 = grp.p[0].out.W, Di = grp.p[0].in.D, Hi = grp.p[0].in.H, Wi = grp.p[0].in.W;
    const int bh = H0 >> 2, bw = W0 >> 2, bps = (D0 >> 1) * bh * bw;          // boxes per sample
    const int nbox = (grp.p[0].M / (D0 * H0 * W0)) * bps;                      // boxes per model
    const int total = nbox * grp.zdim;
    const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int g0 = blockIdx.x * per, g1 = g0 + per < total ? g0 + per : total;

    // per-lane operand offsets: tap tiles T = wave + 4t; voxel pairs q (voxel = 2q + kq)
    int koff[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        int tap = 32 * (wave + 4 * t) + li;
        tap = tap < 343 ? tap : 342;                 // padded taps compute a duplicate that the epilogue drops
        koff[t] = (tap / 49) * 169 + ((tap / 7) % 7) * 13 + tap % 7;
    }
    int moff[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int v = 2 * q + kq;
        moff[q] = (v >> 4) * 2 * 169 + ((v >> 2) & 3) * 2 * 13 + (v & 3) * 2;
    }
    const bool t2ok = wave + 8 < 11;                // wave 3 owns tap tiles 3 and 7 only

  for (int gs = g0; gs < g1;) {
    const int model = gs / nbox, seg_end = (model + 1) * nbox < g1 ? (model + 1) * nbox : g1;
    const int b0 = gs - model * nbox, b1 = seg_end - model * nbox;
    gs = seg_end;
    const Conv0BwdWP& p = grp.p[model];
    __syncthreads();                          // the previous segment is done with cst / xs / dys
    if (tid < 64) {
        float mu, rs, ga_;
        double t1, t2;
        bn_bwd_consts(p.bn, p.bb, tid, mu, rs, ga_, t1, t2);
        cA[tid] = ga_ * rs;
        cB[tid] = (float)(t1 * (double)p.bn.inv_count);
        cC[tid] = (float)(t2 * (double)p.bn.inv_count);
        cM[tid] = mu; cR[tid] = rs;
        if (b0 == 0 && p.dgamma) {            // exactly one workgroup owns a model's first box
            p.dgamma[tid] += (float)t2;
            p.dbeta[tid] += (float)t1;
        }
    }
    __syncthreads();
    // staging roles: 6 region elements per thread; dy: voxel tid >> 3, channels (tid & 7) * 8 .. + 7
    const int sv = tid >> 3, sc0 = (tid & 7) * 8;
    float xr[6];
    float4 rg0, rg1, ry0, ry1;
    auto gload = [&](int bx) {
        const int b = bx / bps, r = bx - b * bps, bz = r / (bh * bw), r2 = r - bz * (bh * bw), by = r2 / bw, bxw = r2 - by * bw;
        const int od0 = 2 * bz, oh0 = 4 * by, ow0 = 4 * bxw, id0 = 2 * od0 - 3, ih0 = 2 * oh0 - 3, iw0 = 2 * ow0 - 3;
        const float* xb = p.x + (size_t)b * Di * Hi * Wi;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int e = tid + 256 * i, rd = e / 169, e2 = e - rd * 169, rh = e2 / 13, rw = e2 - rh * 13;
            const int id = id0 + rd, ih = ih0 + rh, iw = iw0 + rw;
            const bool ok = e < C0_REG && (unsigned)id < (unsigned)Di && (unsigned)ih < (unsigned)Hi && (unsigned)iw < (unsigned)Wi;
            xr[i] = ok ? xb[((size_t)id * Hi + ih) * Wi + iw] : 0.f;
        }
        const int m = ((b * D0 + od0 + (sv >> 4)) * H0 + oh0 + ((sv >> 2) & 3)) * W0 + ow0 + (sv & 3);
        const float* gp = p.dbn + (size_t)m * 64 + sc0;
        const float* yp = p.y0 + (size_t)m * 64 + sc0;
        rg0 = *(const float4*)gp; rg1 = *(const float4*)(gp + 4);
        ry0 = *(const float4*)yp; ry1 = *(const float4*)(yp + 4);
    };
    auto dyv = [&](float g, float y, int n) { return cA[n] * (g - cB[n] - (y - cM[n]) * cR[n] * cC[n]); };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 6; ++i) { const int e = tid + 256 * i; if (e < C0_REG) xs[buf][e] = xr[i]; }
        float* d = &dys[buf][sv * C0_DYP + sc0];
        *(float4*)d = make_float4(dyv(rg0.x, ry0.x, sc0), dyv(rg0.y, ry0.y, sc0 + 1), dyv(rg0.z, ry0.z, sc0 + 2), dyv(rg0.w, ry0.w, sc0 + 3));
        *(float4*)(d + 4) = make_float4(dyv(rg1.x, ry1.x, sc0 + 4), dyv(rg1.y, ry1.y, sc0 + 5), dyv(rg1.z, ry1.z, sc0 + 6), dyv(rg1.w, ry1.w, sc0 + 7));
    };

    f32x16 acc[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][c][r] = 0.f;

    if (b0 < b1) {
        gload(b0);
        sstore(0);
        __syncthreads();
        int buf = 0;
        for (int bx = b0; bx < b1; ++bx) {
#ifdef C0_NO_STAGE
            const bool more = false;
#else
            const bool more = bx + 1 < b1;
#endif
            if (more) gload(bx + 1);
            const float* xb = xs[buf];
            const float* db = dys[buf];
#ifndef C0_NO_MMA
            // operands of voxel pair q + 1 are requested before the MFMAs of pair q, order pinned (one wave per SIMD: nothing else covers
            // the LDS latency; left alone the compiler exposes it twice per pair)
            float bv0[2], bv1[2], a0[2], a1[2], a2[2];
            auto oread = [&](int q, int j) __attribute__((always_inline)) {
                bv0[j] = db[(2 * q + kq) * C0_DYP + li]; bv1[j] = db[(2 * q + kq) * C0_DYP + 32 + li];
                a0[j] = xb[koff[0] + moff[q]]; a1[j] = xb[koff[1] + moff[q]]; a2[j] = xb[koff[2] + moff[q]];
            };
            oread(0, 0);
            static_for<16>([&](auto Q) __attribute__((always_inline)) {
                constexpr int q = decltype(Q)::value, j = q & 1;
                if constexpr (q + 1 < 16) oread(q + 1, j ^ 1);
                asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bv0[j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bv1[j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], bv0[j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], bv1[j], acc[1][1], 0, 0, 0);
                if (t2ok) {
                    acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[j], bv0[j], acc[2][0], 0, 0, 0);
                    acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[j], bv1[j], acc[2][1], 0, 0, 0);
                }
                asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
            });
#endif
            if (more) sstore(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    // epilogue: per wave, tile by tile through LDS so that the atomics run along k (dW0[n][k] is contiguous in k)
    float* cs = Cs[wave];
    float* dwdst = p.dw_rep ? p.dw_rep + (size_t)(blockIdx.x % (unsigned)p.nrep) * (64 * 343) : p.dw;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        if (t == 2 && !t2ok) break;
        const int T = wave + 4 * t;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * kq) * 65 + 32 * c + li] = acc[t][c][r];
        __builtin_amdgcn_wave_barrier();
#ifndef C0_NO_EPI
        for (int idx = lane; idx < 32 * 64; idx += 64) {
            const int tl = idx & 31, ch = idx >> 5, tap = 32 * T + tl;
            if (tap < 343) atomicAdd(&dwdst[ch * 343 + tap], cs[tl * 65 + ch]);
        }
#endif
        __builtin_amdgcn_wave_barrier();
    }
  }
}
__global__ __launch_bounds__(256) void conv0_dw_reduce_kernel(const Grp<Conv0BwdWP> grp) {
    const Conv0BwdWP& p = grp.p[blockIdx.z];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 64 * 343) return;
    float v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = p.dw_rep[(size_t)(r < p.nrep ? r : 0) * (64 * 343) + i];
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) if (r < p.nrep) { a += v[r]; p.dw_rep[(size_t)r * (64 * 343) + i] = 0.f; }      // (left zeroed for the next call)
    p.dw[i] += a;
}

extern "C" int mms_conv0_bwd_weight_group(const Conv0BwdWP* pp, int ng, const MmsDnOpts* opts, hipStream_t s) {
    if (!pp || ng < 1 || ng > MMS_MAX_GROUP) return MMS_ERR_ARG;
    const Conv0BwdWP& p = *pp;
    if (p.M <= 0 || p.msplit <= 0) return MMS_ERR_ARG;
    for (int g = 1; g < ng; ++g) {
        const Conv0BwdWP& q = pp[g];
        if (q.M != p.M || q.msplit != p.msplit || q.in.D != p.in.D || q.in.H != p.in.H || q.in.W != p.in.W || q.out.D != p.out.D ||
            q.out.H != p.out.H || q.out.W != p.out.W) return MMS_ERR_ARG;
    }
    const long vox = (long)p.out.D * p.out.H * p.out.W;
    const bool boxed = (p.out.D % 2 == 0) && (p.out.H % 4 == 0) && (p.out.W % 4 == 0) && vox > 0 && p.M % vox == 0 &&
                       p.in.D == 2 * p.out.D && p.in.H == 2 * p.out.H && p.in.W == 2 * p.out.W;
    if (!boxed) return launch_tile_gemm<Conv0BwdWOp>(pp, ng, dim3(6, 1, p.msplit), s);
    Grp<Conv0BwdWP> a;
    grp_fill(a, pp, ng, ng);
    // workgroups: 2 per CU (the second hides the first one's staging), but at least 8 boxes of 32 voxels each
    const long boxes = (long)ng * (p.M / 32);
    int nwg = (opts && opts->c0_nwg > 0) ? opts->c0_nwg : (boxes >= 256L * 8 ? 256 : (int)((boxes + 7) / 8));
    if (nwg < 1) nwg = 1;
    for (int g = 0; g < ng; ++g) if ((pp[g].dw_rep != nullptr) != (p.dw_rep != nullptr) || pp[g].nrep != p.nrep || (p.dw_rep && (p.nrep < 1 || p.nrep > 8))) return MMS_ERR_ARG;
    MMS_LAUNCH(conv0_bwd_weight_kernel, dim3(nwg, 1, 1), dim3(256), 0, s, a);
    if (p.dw_rep) {
        const int rc = mms_check_launch();
        if (rc != MMS_OK) return rc;
        Grp<Conv0BwdWP> b;
        grp_fill(b, pp, ng, 1);
        MMS_LAUNCH(conv0_dw_reduce_kernel, dim3((64 * 343 + 255) / 256, 1, ng), dim3(256), 0, s, b);
    }
    return mms_check_launch();
}
MMS_SINGLE_O(mms_conv0_bwd_weight, Conv0BwdWP)
