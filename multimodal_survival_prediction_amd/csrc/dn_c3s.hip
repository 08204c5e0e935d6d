// 3x3x3 convolution of the dense layers on SMALL grids (dense blocks 2-4 of 64x64x32 volumes: 8x8x4, 4x4x2, 2x2x1 voxels per sample;
// MONAI _DenseLayer.layers.conv2 as restated in oracle/densenet3d.py), forward and backward-data, WITHOUT a tap split.
//
// The tile-GEMM forms (dn_fwd.hip Conv3FwdOp, dn_bwd.hip Conv3BwdDataOp) give a block with few rows enough workgroups by spreading
// the 27 taps over workgroups and summing their partial tiles in a second launch -- one more kernel on the step's critical chain per
// layer and pass, and 2.2-2.6x the algorithmic bytes.  Here a workgroup owns 16 consecutive voxel rows and ALL 27 taps:
//   * rows are voxels in (b, d, h, w) order, so every tap of a row tile reads rows inside [m0 - halo, m0 + 16 + halo) with
//     halo = H*W + W + 1.  That window (90 rows in block 2, 38 in block 3) is staged in LDS ONCE -- for the forward after BatchNorm2 +
//     ReLU, i.e. the transform runs once per element instead of once per tap -- and every tap reads its A operand from a shifted slot;
//   * v_mfma_f32_16x16x4_f32 (16-row tiles: twice the workgroups of the 32-row forms), operands as float4 of 4 consecutive k per lane
//     (element e of lane (i, h) is k = 16q + 4h + e for both operands, so MFMA e sums k in {e, 4+e, 8+e, 12+e} of the group);
//   * the weight operand goes from L2 straight into the MFMA register layout through a register ring 6-9 taps deep; with the weights
//     packed in fragment order (wfrag: what the network driver does for these layers) each of those loads is one contiguous 1 KB per wave --
//     from the classic [co][tap][cin] pack a wave instruction gathers 64 separate 16-byte pieces and the launch is bound by the texture path;
//   * zero padding = a 0/1 factor per (row, tap) on the A registers, from the row's 9-bit tap-validity mask;
//   * forward: the 4 waves split the 128 input channels and are summed through LDS; backward-data: they split the 128 output
//     (= conv input) channels, nothing to sum.  With few row tiles in the launch the output channels are split over blockIdx.y too (JN = 1).
// Summation order is fixed: results are deterministic and independent of the group size.
#include "dn_ops.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ unsigned c3s_mask9(int c, Dims3 g, bool mirror) {
    int d, h, w;
    unpack_dhw(c, d, h, w);
    const unsigned lo_d = d > 0, hi_d = d + 1 < g.D, lo_h = h > 0, hi_h = h + 1 < g.H, lo_w = w > 0, hi_w = w + 1 < g.W;
    const unsigned dm = mirror ? (hi_d | 2u | (lo_d << 2)) : (lo_d | 2u | (hi_d << 2));
    const unsigned hm = mirror ? (hi_h | 2u | (lo_h << 2)) : (lo_h | 2u | (hi_h << 2));
    const unsigned wm = mirror ? (hi_w | 2u | (lo_w << 2)) : (lo_w | 2u | (hi_w << 2));
    return dm | (hm << 3) | (wm << 6);
}

constexpr int C3S_TM = 16;
constexpr int C3S_FP = 132;         // forward window pitch (floats): 128 channels + 4
constexpr int C3S_BP = 36;          // backward window pitch: 32 channels + 4
constexpr int C3S_NI = (MMS_C3S_MAXROWS + 7) / 8;

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// -DC3S_TIMING (tools/c3s_timing.py): shader-clock stamps of the kernel's phases, one record of 8 words per workgroup
#ifdef C3S_TIMING
__device__ unsigned long long* c3s_ts_buf = nullptr;
#define C3S_TS_DECL unsigned long long ts_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define C3S_STAMP(i) do { asm volatile("" ::: "memory"); ts_[i] = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); } while (0)
#define C3S_STAMPW(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); ts_[i] = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); } while (0)
#define C3S_TS_FLUSH() do { if (threadIdx.x == 0 && c3s_ts_buf) { unsigned long long* o = c3s_ts_buf + 8 * (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)); \
    for (int i_ = 0; i_ < 8; ++i_) o[i_] = ts_[i_]; } } while (0)
#else
#define C3S_TS_DECL
#define C3S_STAMP(i)
#define C3S_STAMPW(i)
#define C3S_TS_FLUSH()
#endif

// Weight stream: one CU cannot hide an L2/MALL round trip (~1600 cycles with every workgroup of the launch asking for the same
// lines) behind one tap's MFMAs (256-512 cycles), and left alone the compiler sinks each tap's loads next to their use (measured:
// 0.8 us per tap, fully exposed -- 21.8 us per block-3 launch).  The taps' weights therefore sit in a register RING of D taps filled
// in program order (tap t + D is requested right after tap t's MFMAs) and pinned there with scheduling barriers.
// (the empty asm with a memory clobber keeps the IR-level passes from sinking the loads, the scheduling barrier the machine scheduler:
// ALU, MFMA and LDS instructions may cross it, global loads may not)
#define C3S_PIN() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0x38F); } while (0)

// Shared tap loop: A fragments of tap t+1 are read from LDS before tap t's MFMAs (the pins keep LDS reads where they are written, so the
// prefetch is explicit), the MFMAs rotate over the 2 JN accumulators (a dependent v_mfma_f32_16x16x4_f32 issues every 40 cycles, an
// independent one every 32), tap t + D's weights are requested right behind tap t's MFMAs.  SIGN = +1 forward, -1 backward-data (mirrored taps).
#define C3S_TAP_LOOP(SIGN, PITCH)                                                                                                     \
    f32x4 acc[JN][2];                                                                                                                 \
    _Pragma("unroll") for (int j = 0; j < JN; ++j) _Pragma("unroll") for (int q = 0; q < 2; ++q) acc[j][q] = f32x4{0.f, 0.f, 0.f, 0.f}; \
    float4 an[2];                                                                                                                     \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) an[q] = *(const float4*)(arow + (SIGN) * (-HW - W - 1) * (PITCH) + 16 * q);           \
    _Pragma("unroll") for (int t = 0; t < 27; ++t) {                                                                                  \
        const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;                                                                           \
        const unsigned sel = (1u << kd) | (8u << kh) | (64u << kw);                                                                   \
        const float mk = (m9 & sel) == sel ? 1.f : 0.f;                                                                               \
        float4 a[2];                                                                                                                  \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) a[q] = make_float4(an[q].x * mk, an[q].y * mk, an[q].z * mk, an[q].w * mk);     \
        if (t + 1 < 27) {                                                                                                             \
            const int t1 = t + 1, off1 = (t1 / 9 - 1) * HW + ((t1 / 3) % 3 - 1) * W + (t1 % 3 - 1);                                   \
            _Pragma("unroll") for (int q = 0; q < 2; ++q) an[q] = *(const float4*)(arow + (SIGN) * off1 * (PITCH) + 16 * q);          \
        }                                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);      /* tap t's MFMAs stay BEHIND the LDS reads of tap t + 1 (they cover their latency) */  \
        const float4 (&bt)[JN][2] = b[t % D];                                                                                         \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) _Pragma("unroll") for (int j = 0; j < JN; ++j) acc[j][q] = MFMA16(a[q].x, bt[j][q].x, acc[j][q]); \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) _Pragma("unroll") for (int j = 0; j < JN; ++j) acc[j][q] = MFMA16(a[q].y, bt[j][q].y, acc[j][q]); \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) _Pragma("unroll") for (int j = 0; j < JN; ++j) acc[j][q] = MFMA16(a[q].z, bt[j][q].z, acc[j][q]); \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) _Pragma("unroll") for (int j = 0; j < JN; ++j) acc[j][q] = MFMA16(a[q].w, bt[j][q].w, acc[j][q]); \
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);                                                             \
        if (t + D < 27) bload(b[t % D], t + D);                                                                                       \
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);                                                             \
    }

// Placement (speed only, results unchanged): workgroups are dealt round-robin over the 8 XCDs by flat id, and WHICH weights a workgroup
// reads is decided by its (model, output-channel half) pair.  The pair is therefore taken from flat % pairs and the row tile from
// flat / pairs: an XCD's L2 then serves ONE pair when the pair count divides 8 (1, 2, 4, 8: one or two models per launch) and half of them
// otherwise -- with (tile, half, model) = blockIdx every L2 fetched every model's 442 KB per layer (PMC: 2.9x the algorithmic bytes of the forward).
__device__ __forceinline__ void c3s_place(int& tile, int& half, int& model) {
    const int gy = gridDim.y, pairs = gy * (int)gridDim.z;
    const int flat = blockIdx.x + gridDim.x * (blockIdx.y + gy * blockIdx.z);
    const int pr = flat % pairs;
    tile = flat / pairs; model = pr / gy; half = pr - model * gy;
}

// ---- forward: out[m][co] = sum_tap sum_cin relu(bn2(y1))[m + off(tap)][cin] * W[co][tap][cin] --------------------------------------
template <int JN, int D, bool FRAG>
__global__ __launch_bounds__(256) void conv3s_fwd_kernel(const Grp<Conv3FwdP> grp) {
    // every kernel argument the prologue needs, read ONCE into registers: left as references into the kernarg segment the compiler
    // re-loads them (s_load + wait) inside each predicated load below -- 30 serial scalar round trips, 3.3 us before the first MFMA
    int tile_, half_, model_;
    c3s_place(tile_, half_, model_);
    const Conv3FwdP& p = grp.p[model_];
    const float* __restrict__ y1 = p.y1;
    const float* __restrict__ wp = p.wp;
    const int* __restrict__ coords = p.coords;
    const int M = p.M;
    const Dims3 g = p.g;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, h = lane >> 4;
    const int m0 = tile_ * C3S_TM, co0 = JN == 1 ? 16 * half_ : 0;
    const int W = g.W, HW = g.H * g.W, halo = HW + W + 1, nrows = C3S_TM + 2 * halo;
    C3S_TS_DECL;
    C3S_STAMP(0);
#ifdef C3S_TIMING
    { int probe_ = M + g.W; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 0" : "+s"(probe_) :: "memory"); ts_[6] = __builtin_amdgcn_s_memtime(); }
#endif

    // window rows (raw y1) -> registers: branch-free, from clamped (always valid) addresses; rows outside [0, M) are zeroed when staged
    const int c4 = (tid & 31) * 4;
    float4 wv[C3S_NI];
    unsigned wok = 0;
#pragma unroll
    for (int i = 0; i < C3S_NI; ++i) {
        if (8 * i < nrows) {                 // workgroup-uniform
            const int s = (tid >> 5) + 8 * i, src = m0 - halo + s;
            wok |= ((s < nrows && src >= 0 && src < M) ? 1u : 0u) << i;
            const int sc_ = src < 0 ? 0 : (src < M ? src : M - 1);
            wv[i] = *(const float4*)(y1 + (size_t)sc_ * 128 + c4);
        }
    }
    const int myrow = m0 + li;
    const int mycoord = coords[myrow < M ? myrow : M - 1];
    // the first D taps' weights, behind the window in the memory queue (vmcnt retires in order: the window is needed first)
    // classic pack [co][tap][cin]: a wave instruction gathers 64 separate 16-byte pieces (16 rows x 64 B); fragment order: one contiguous 1 KB
    const float* wlane = FRAG ? wp + (size_t)wave * 1024 + (co0 >> 4) * 256 + lane * 4
                              : wp + (size_t)(co0 + li) * (27 * 128) + 32 * wave + 4 * h;
    float4 b[D][JN][2];
    auto bload = [&](float4 (&bt)[JN][2], int tap) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < JN; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                bt[j][q] = FRAG ? *(const float4*)(wlane + tap * 4096 + q * 512 + j * 256)
                                : *(const float4*)(wlane + (size_t)j * 16 * (27 * 128) + tap * 128 + 16 * q);
    };
    C3S_PIN();
#pragma unroll
    for (int t = 0; t < D; ++t) bload(b[t], t);
    C3S_PIN();
    // BatchNorm2 constants of this thread's 4 channels; transform -> LDS
    float mean[4], sc[4], beta[4];
    C3S_STAMP(7);
    bn_consts4(p.bn, c4, mean, sc, beta);
    C3S_STAMPW(1);
    const unsigned m9 = myrow < M ? c3s_mask9(mycoord, g, false) : 0u;
#pragma unroll
    for (int i = 0; i < C3S_NI; ++i) {
        const int s = (tid >> 5) + 8 * i;
        if (8 * i < nrows && s < nrows) {
            const float z = (wok >> i) & 1u ? 1.f : 0.f;
            *(float4*)&smem[s * C3S_FP + c4] =
                make_float4(z * fmaxf(bn_apply(wv[i].x, mean[0], sc[0], beta[0]), 0.f), z * fmaxf(bn_apply(wv[i].y, mean[1], sc[1], beta[1]), 0.f),
                            z * fmaxf(bn_apply(wv[i].z, mean[2], sc[2], beta[2]), 0.f), z * fmaxf(bn_apply(wv[i].w, mean[3], sc[3], beta[3]), 0.f));
        }
    }
    __syncthreads();
    C3S_STAMP(2);

    const float* arow = smem + (li + halo) * C3S_FP + 32 * wave + 4 * h;
    C3S_TAP_LOOP(1, C3S_FP)
    C3S_STAMP(3);
    __syncthreads();                                      // the window is dead: Cs aliases it
    // ---- epilogue: add the four channel quarters, write the slab columns, batch statistics
    constexpr int NC = 16 * JN, CP = NC + 1;
    float* Cs = smem;                                     // [4 waves][16][CP]
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cs[(wave * 16 + 4 * h + r) * CP + 16 * j + li] = acc[j][0][r] + acc[j][1][r];
    __syncthreads();
    double* red = (double*)(smem + 4 * 16 * CP + (4 * 16 * CP & 1));      // [2][8][NC], 8-byte aligned
    const int c = tid % NC, rg = tid / NC;                // JN = 2: 8 row groups x 2 rows; JN = 1: 16 row groups x 1 row
    constexpr int NRG = 256 / NC, RPT = 16 / NRG;
    double s = 0, q2 = 0;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = rg * RPT + i, m = m0 + r;
        const float v = (Cs[r * CP + c] + Cs[(16 + r) * CP + c]) + (Cs[(32 + r) * CP + c] + Cs[(48 + r) * CP + c]);
        if (m < p.M) {
            p.out[(size_t)m * p.ldo + co0 + c] = v;
            s += v; q2 += (double)v * v;
        }
    }
    C3S_STAMP(4);
    if (p.osum == nullptr) { C3S_TS_FLUSH(); return; }
    red[rg * NC + c] = s; red[(NRG + rg) * NC + c] = q2;
    __syncthreads();
    if (tid < NC) {
        double a = 0, b = 0;
#pragma unroll
        for (int g = 0; g < NRG; ++g) { a += red[g * NC + tid]; b += red[(NRG + g) * NC + tid]; }
        atomicAdd(&stat_rep(p.osum, p.srep, p.sstride)[co0 + tid], a);
        atomicAdd(&stat_rep(p.osumsq, p.srep, p.sstride)[co0 + tid], b);
    }
    C3S_STAMPW(5);
    C3S_TS_FLUSH();
}

// ---- backward-data: dbn2[m][cin] = [a2 > 0] * sum_tap sum_co dz[m - off(tap)][co] * W[cin][tap][co]; BatchNorm2-backward sums -------
template <int JN, int D, bool FRAG>
__global__ __launch_bounds__(256) void conv3s_bwd_data_kernel(const Grp<Conv3BwdDataP> grp) {
    int tile_, half_, model_;
    c3s_place(tile_, half_, model_);
    const Conv3BwdDataP& p = grp.p[model_];
    const float* __restrict__ dz = p.dz;               // kernel arguments read once (see conv3s_fwd_kernel)
    const float* __restrict__ wpb = p.wpb;
    const float* __restrict__ y1 = p.y1;
    const int* __restrict__ coords = p.coords;
    const int M = p.M, lddz = p.lddz;
    const Dims3 g = p.g;
    const BnSrc bn = p.bn;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, h = lane >> 4;
    const int m0 = tile_ * C3S_TM;
    const int cin0 = (JN == 1 ? 64 * half_ : 0) + 16 * JN * wave;      // this wave's first output column
    const int W = g.W, HW = g.H * g.W, halo = HW + W + 1, nrows = C3S_TM + 2 * halo;

    // dz window: 8 threads per row (32 channels), 32 rows per pass; clamped addresses, rows outside [0, M) zeroed when staged
    constexpr int NP = (MMS_C3S_MAXROWS + 31) / 32;
    float4 wv[NP];
    unsigned wok = 0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (32 * i < nrows) {
            const int s = (tid >> 3) + 32 * i, src = m0 - halo + s;
            wok |= ((s < nrows && src >= 0 && src < M) ? 1u : 0u) << i;
            const int sc_ = src < 0 ? 0 : (src < M ? src : M - 1);
            wv[i] = *(const float4*)(dz + (size_t)sc_ * lddz + (tid & 7) * 4);
        }
    }
    const int myrow = m0 + li;
    const int mycoord = coords[myrow < M ? myrow : M - 1];
    const float* wlane = FRAG ? wpb + (size_t)(cin0 >> 4) * 512 + lane * 4
                              : wpb + (size_t)(cin0 + li) * (27 * 32) + 4 * h;
    float4 b[D][JN][2];
    auto bload = [&](float4 (&bt)[JN][2], int tap) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < JN; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                bt[j][q] = FRAG ? *(const float4*)(wlane + tap * 4096 + j * 512 + q * 256)
                                : *(const float4*)(wlane + (size_t)j * 16 * (27 * 32) + tap * 32 + 16 * q);
    };
    C3S_PIN();
#pragma unroll
    for (int t = 0; t < D; ++t) bload(b[t], t);
    C3S_PIN();
    // epilogue operands of this lane's JN columns x 4 rows, requested now: y1, BatchNorm2 statistics and parameters (all loads first)
    float yv[JN][4], mu[JN], rstd[JN], ga[JN], be[JN];
    double bs[JN], bq[JN];
#pragma unroll
    for (int j = 0; j < JN; ++j) {
        const int c = cin0 + 16 * j + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 4 * h + r;
            yv[j][r] = y1[(size_t)(m < M ? m : M - 1) * 128 + c];
        }
        bs[j] = bn.sum[c]; bq[j] = bn.sumsq[c];
        ga[j] = bn.gamma[c]; be[j] = bn.beta[c];
    }
#pragma unroll
    for (int j = 0; j < JN; ++j) {
        const int c = cin0 + 16 * j + li;
        for (int r = 1; r < bn.nrep; ++r) { bs[j] += bn.sum[c + (size_t)r * bn.rep_stride]; bq[j] += bn.sumsq[c + (size_t)r * bn.rep_stride]; }
        const double m = bs[j] * (double)bn.inv_count;
        double v = bq[j] * (double)bn.inv_count - m * m;
        v = v > 0.0 ? v : 0.0;
        mu[j] = (float)m; rstd[j] = 1.0f / sqrtf((float)v + bn.eps);
    }
    const unsigned m9 = myrow < M ? c3s_mask9(mycoord, g, true) : 0u;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int s = (tid >> 3) + 32 * i;
        if (32 * i < nrows && s < nrows) {
            const float z = (wok >> i) & 1u ? 1.f : 0.f;
            *(float4*)&smem[s * C3S_BP + (tid & 7) * 4] = make_float4(z * wv[i].x, z * wv[i].y, z * wv[i].z, z * wv[i].w);
        }
    }
    __syncthreads();

    const float* arow = smem + (li + halo) * C3S_BP + 4 * h;
    C3S_TAP_LOOP(-1, C3S_BP)
    // ---- epilogue (wave-local: column = cin0 + 16 j + li, rows 4h .. 4h+3): relu2 mask, dbn2, the two BatchNorm-backward sums
#pragma unroll
    for (int j = 0; j < JN; ++j) {
        const int c = cin0 + 16 * j + li;
        double s1 = 0, s2 = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 4 * h + r;
            if (m < p.M) {
                const float xh = (yv[j][r] - mu[j]) * rstd[j];
                const float g = fmaf(ga[j], xh, be[j]) > 0.f ? acc[j][0][r] + acc[j][1][r] : 0.f;
                p.dbn[(size_t)m * 128 + c] = g;
                s1 += g; s2 += (double)g * xh;
            }
        }
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (h == 0) {
            atomicAdd(&stat_rep(p.s1, p.srep, p.sstride)[c], s1);
            atomicAdd(&stat_rep(p.s2, p.srep, p.sstride)[c], s2);
        }
    }
}

template <int JN, int D, bool FRAG>
int launch_fwd(const Conv3FwdP* pp, int ng, hipStream_t s) {
    const Conv3FwdP& p = *pp;
    const int nrows = C3S_TM + 2 * (p.g.H * p.g.W + p.g.W + 1);
    int smem = nrows * C3S_FP * (int)sizeof(float);
    if (smem < 12800) smem = 12800;                                  // the epilogue's 4 x 16 x 33 floats + 2 x 8 x 32 doubles alias the window
    static std::once_flag attr_once;
    std::call_once(attr_once, [] { hipFuncSetAttribute((const void*)(void (*)(const Grp<Conv3FwdP>))conv3s_fwd_kernel<JN, D, FRAG>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       MMS_C3S_MAXROWS * C3S_FP * (int)sizeof(float)); });
    Grp<Conv3FwdP> a;
    grp_fill(a, pp, ng, 1);
    const auto kern = conv3s_fwd_kernel<JN, D, FRAG>;
    MMS_LAUNCH(kern, dim3((p.M + C3S_TM - 1) / C3S_TM, 2 / JN, ng), dim3(256), smem, s, a);
    return mms_check_launch();
}
template <int JN, int D, bool FRAG>
int launch_bwd(const Conv3BwdDataP* pp, int ng, hipStream_t s) {
    const Conv3BwdDataP& p = *pp;
    const int nrows = C3S_TM + 2 * (p.g.H * p.g.W + p.g.W + 1);
    const int smem = nrows * C3S_BP * (int)sizeof(float);
    Grp<Conv3BwdDataP> a;
    grp_fill(a, pp, ng, 1);
    const auto kern = conv3s_bwd_data_kernel<JN, D, FRAG>;
    MMS_LAUNCH(kern, dim3((p.M + C3S_TM - 1) / C3S_TM, 2 / JN, ng), dim3(256), smem, s, a);
    return mms_check_launch();
}

}  // namespace

#ifdef C3S_TIMING
extern "C" int mms_c3s_timing_buffer(void* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(c3s_ts_buf), &buf, sizeof(buf)) == hipSuccess ? MMS_OK : MMS_ERR_LAUNCH; }
#endif
// Driver-internal launchers (argument checks are the callers': mms_conv3_fwd_group / mms_conv3_bwd_data_group).
int mms_c3s_fwd(const Conv3FwdP* pp, int ng, const MmsDnOpts& o, hipStream_t s) {
    const bool two = mms_conv3_small_jn(pp->M, ng, pp->g, o) == 2;
    // Ring depth of the one-tile form (MmsDnOpts.c3s_ring = 9 / 6 / 4, fragment-ordered weights only).  Alone on the GPU a launch is as fast with
    // 9 taps in flight as with 6 (block 3: 9.6 us either way), but the 9-deep kernel holds 200 VGPRs against 168: beside two resident
    // workgroups of another stream's block-1 forward (164 VGPRs each) a SIMD has 176 registers left, so only the 6-deep one can be
    // placed there.  K = 5 epoch on three streams: 2709-2717 (9) / 2736-2751 (6) / 2736-2742 (4) patients/s.
    const int ring = o.c3s_ring > 0 ? o.c3s_ring : 6;
    if (pp->wfrag && !two && ring == 6) return launch_fwd<1, 6, true>(pp, ng, s);
    if (pp->wfrag && !two && ring == 4) return launch_fwd<1, 4, true>(pp, ng, s);
    if (pp->wfrag) return two ? launch_fwd<2, 6, true>(pp, ng, s) : launch_fwd<1, 9, true>(pp, ng, s);
    return two ? launch_fwd<2, 6, false>(pp, ng, s) : launch_fwd<1, 9, false>(pp, ng, s);
}
int mms_c3s_bwd_data(const Conv3BwdDataP* pp, int ng, const MmsDnOpts& o, hipStream_t s) {
    const bool two = mms_conv3_small_jn(pp->M, ng, pp->g, o) == 2;
    const int ring = o.c3s_ring > 0 ? o.c3s_ring : 6;      // 140 / 112 / 96 VGPRs at 9 / 6 / 4
    if (pp->wfrag && !two && ring == 6) return launch_bwd<1, 6, true>(pp, ng, s);
    if (pp->wfrag && !two && ring == 4) return launch_bwd<1, 4, true>(pp, ng, s);
    if (pp->wfrag) return two ? launch_bwd<2, 6, true>(pp, ng, s) : launch_bwd<1, 9, true>(pp, ng, s);
    return two ? launch_bwd<2, 6, false>(pp, ng, s) : launch_bwd<1, 9, false>(pp, ng, s);
}
