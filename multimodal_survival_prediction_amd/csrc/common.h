// Shared device helpers for the mmsurv gfx950 kernels (CDNA4, wave64, fp32-input MFMA).
#pragma once
#include <utility>
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mmsurv.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Fold groups: one launch advances up to MMS_MAX_GROUP independent models of identical shape (the K-fold models of the
// reference's cross-validation loop, final_multimodal.py:316-402).  The kernel argument is the array of the models'
// parameter blocks BY VALUE (<= 10 x 352 B, kernarg segment); the model index is an extra grid dimension: blockIdx.z for
// plain kernels, blockIdx.z / zdim for the tile-GEMM core (whose own z = blockIdx.z % zdim).  Per-model work is exactly
// the single-model kernel's: grouping changes placement only, never results.
template <class P> struct Grp { P p[MMS_MAX_GROUP]; int zdim; };
template <class P> static inline bool grp_fill(Grp<P>& a, const P* pp, int ng, int zdim) {
    if (!pp || ng < 1 || ng > MMS_MAX_GROUP) return false;
    for (int g = 0; g < ng; ++g) a.p[g] = pp[g];
    a.zdim = zdim;
    return true;
}
#define MMS_SINGLE(name, T) extern "C" int name(const T* p, hipStream_t s) { return name##_group(p, 1, s); }
#define MMS_SINGLE_O(name, T) extern "C" int name(const T* p, hipStream_t s) { return name##_group(p, 1, nullptr, s); }      // (launch-shape options: defaults)
// launch-shape options by value: a NULL pointer means "all defaults" (include/mmsurv.h: MmsDnOpts)
static inline MmsDnOpts mms_opts(const MmsDnOpts* o) { return o ? *o : MmsDnOpts{}; }


// XCD-aware placement of an M-tiled launch whose grid is (tiles, *, models) (cdna_hip_programming.md T1; placement affects speed only,
// never results).  Workgroups are dealt round-robin over the 8 XCDs by linear id, so with gridDim.x a multiple of 8 the workgroups
// with equal blockIdx.x % 8 share an XCD (and its L2) whatever their y / z.  (a) One model: XCD k gets the CONTIGUOUS tile range
// [k n/8, (k+1) n/8) -- neighbouring tiles share halo rows.  (b) A fold group of 2, 4 or 8 models: model g gets the 8 / ng XCDs
// g * 8/ng ..: a layer's weights are then fetched by 4 / 2 / 1 L2s instead of by all 8 (round 3, sub-groups of 2: forward 2.96x the
// algorithmic bytes at the fabric, each of the 8 L2s holding its own copy of both models' 442 KB).  -> model index, tile index.
__device__ __forceinline__ void xcd_place(int& gi, int& bx) {
    gi = blockIdx.z; bx = blockIdx.x;
    if ((gridDim.x & 7) != 0) return;
    const int k = blockIdx.x & 7, per = gridDim.x >> 3, ng = gridDim.z;
    if (ng > 1 && ng <= 8 && (8 % ng) == 0) {
        const int X = 8 / ng, slots = per * ng, slot = (blockIdx.x >> 3) + per * blockIdx.z;
        gi = k / X; bx = (k % X) * slots + slot;
    } else {
        bx = k * per + (blockIdx.x >> 3);
    }
}

// Compile-time loop: f(std::integral_constant<int, 0>{}) ... f(<N - 1>) in order.  (A `#pragma unroll` loop whose body holds rarely taken
// branches is costed at full size per iteration and left rolled; arrays it indexes by the loop counter then live in scratch.)
template <class F, int... T>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, T...>) { (f(std::integral_constant<int, T>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// Statistic accumulators may be replicated (fp64 atomics on one address serialise: 256 workgroups adding to the same 64
// words cost ~5 us): a producer workgroup adds to replica blockIdx.x % nrep, readers add the replicas up.
__device__ __forceinline__ double* stat_rep(double* base, int nrep, int stride) {
    return (base && nrep > 1) ? base + (size_t)(blockIdx.x % nrep) * stride : base;
}
// Replicas 1 .. nrep-1 of NA accumulators (V = double, or a 2-vector of doubles) whose replica-0 values the caller has already loaded into s[]:
// three replicas at a time, every load of a chunk requested before the first add and branch-free (absent replicas re-read replica 0 and are
// not added) -- a plain loop over the runtime replica count is one dependent memory round trip PER REPLICA in every consumer's prologue,
// which is what made more than one replica per 8192 rows a net loss in rounds 2-3.  Added in replica order.  a[i]: replica 0's address.
template <int NA, class V>
__device__ __forceinline__ void rep_add(const V* const (&a)[NA], int nrep, size_t stride_doubles, V (&s)[NA]) {
    for (int r0 = 1; r0 < nrep; r0 += 3) {                           // uniform
        V t[NA][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const size_t o = (size_t)(r0 + j < nrep ? r0 + j : 0) * stride_doubles;
#pragma unroll
            for (int i = 0; i < NA; ++i) t[i][j] = *(const V*)((const double*)a[i] + o);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (r0 + j < nrep) {
#pragma unroll
                for (int i = 0; i < NA; ++i) s[i] += t[i][j];
            }
    }
}
__device__ __forceinline__ double rep_sum(const double* a, int c, int nrep, int stride) {
    double s[1] = {a[c]};
    const double* const p[1] = {a + c};
    rep_add(p, nrep, (size_t)stride, s);
    return s[0];
}

// Split fixup without a second launch: every workgroup of an output tile publishes its partial, then takes a ticket; the
// one that draws the last ticket returns true -- it may then read all partials of the tile -- and re-arms the counter for
// the next launch.  No workgroup ever waits for another, so there is nothing to deadlock.
// Cross-XCD visibility WITHOUT fences: a device-scope fence is a whole-L2 write-back + invalidate per workgroup (measured:
// 1.5x slower step).  Instead the partials are written with agent-scope relaxed atomic stores (global_store ... sc1:
// write-through to the memory side) and read back with agent-scope atomic loads (sc1: never served from this XCD's L2);
// the only ordering needed is "my stores have been acknowledged before my ticket", i.e. s_waitcnt vmcnt(0) + the barrier.
__device__ __forceinline__ void pstore(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float pload(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool tile_last_arriver(unsigned* counter, unsigned nsplit, int tid) {
    __shared__ unsigned s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's partial stores are acknowledged
    __syncthreads();
    if (tid == 0) {
        const unsigned old = atomicAdd(counter, 1u);
        s_last = (old == nsplit - 1u) ? 1u : 0u;
        if (old == nsplit - 1u) atomicExch(counter, 0u);
    }
    __syncthreads();
    return s_last != 0u;
}

// Two accumulators of one channel with their replicas: BOTH base loads are issued before the replica loop, so the pair costs one
// memory round trip (two rep_sum calls in a row cost two: the first call's loop block separates the loads and its result is consumed
// before the second load is issued).
__device__ __forceinline__ void rep_sum2(const double* a, const double* b, int c, int nrep, int stride, double& sa, double& sb) {
    double s[2] = {a[c], b[c]};
    const double* const p[2] = {a + c, b + c};
    rep_add(p, nrep, (size_t)stride, s);
    sa = s[0]; sb = s[1];
}

// mean / rstd of channel c.  The batch variance is the biased one (what torch normalises with).
__device__ __forceinline__ void bn_mean_rstd(const BnSrc& b, int c, float& mean, float& rstd) {
    if (!b.gamma) { mean = 0.f; rstd = 1.f; return; }             // identity (BnSrc, mmsurv.h)
    if (b.train) {
        double s, q;
        rep_sum2(b.sum, b.sumsq, c, b.nrep, b.rep_stride, s, q);
        const double m = s * (double)b.inv_count;
        double v = q * (double)b.inv_count - m * m;
        v = v > 0.0 ? v : 0.0;
        mean = (float)m;
        rstd = 1.0f / sqrtf((float)v + b.eps);      // the fp64 part is the cancellation-prone E[x^2]-E[x]^2 only
    } else {
        mean = b.rmean[c];
        rstd = 1.0f / sqrtf(b.rvar[c] + b.eps);
    }
}

// (mean, rstd, gamma, beta) of channel c with gamma / beta requested BEFORE the statistics' replica loop: one round trip.
__device__ __forceinline__ void bn_consts1(const BnSrc& b, int c, float& mean, float& rstd, float& gamma, float& beta) {
    if (!b.gamma) { mean = 0.f; rstd = 1.f; gamma = 1.f; beta = 0.f; return; }
    gamma = b.gamma[c]; beta = b.beta[c];
    bn_mean_rstd(b, c, mean, rstd);
}

// Everything a BatchNorm-backward consumer needs for channel c of a training-mode BatchNorm -- (mean, rstd, gamma) and the two backward
// sums (s1 = sum dy, s2 = sum dy * xhat) -- with all five base loads issued together (one round trip instead of five).
__device__ __forceinline__ void bn_bwd_consts(const BnSrc& b, const BnBwd& bb, int c, float& mean, float& rstd, float& gamma, double& t1, double& t2) {
    double sq[2] = {b.sum[c], b.sumsq[c]}, tt[2] = {bb.s1[c], bb.s2[c]};
    gamma = b.gamma[c];
    { const double* const p[2] = {b.sum + c, b.sumsq + c}; rep_add(p, b.nrep, (size_t)b.rep_stride, sq); }
    { const double* const p[2] = {bb.s1 + c, bb.s2 + c}; rep_add(p, bb.nrep, (size_t)bb.rep_stride, tt); }
    t1 = tt[0]; t2 = tt[1];
    const double s = sq[0], q = sq[1];
    const double m = s * (double)b.inv_count;
    double v = q * (double)b.inv_count - m * m;
    v = v > 0.0 ? v : 0.0;
    mean = (float)m;
    rstd = 1.0f / sqrtf((float)v + b.eps);
}

// BatchNorm constants of the 4 consecutive channels c .. c+3 (c % 4 == 0) as (mean, gamma*rstd, beta).  A thread's four bn_mean_rstd
// calls compile to four SERIAL memory round trips (the replica loop's control flow separates them); here the six vector loads are
// issued together: one round trip.  All six arrays must be 16-byte aligned (mms_bn_aligned16: checked by the launchers).
__device__ __forceinline__ void bn_consts4(const BnSrc& b, int c, float (&mean)[4], float (&sc)[4], float (&beta)[4]) {
    if (!b.gamma) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { mean[i] = 0.f; sc[i] = 1.f; beta[i] = 0.f; }
        return;
    }
    const float4 g = *(const float4*)(b.gamma + c), be = *(const float4*)(b.beta + c);
    float4 mu4, rs4;
    if (b.train) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 sv[4] = {*(const d2*)(b.sum + c), *(const d2*)(b.sum + c + 2), *(const d2*)(b.sumsq + c), *(const d2*)(b.sumsq + c + 2)};
        { const d2* const p[4] = {(const d2*)(b.sum + c), (const d2*)(b.sum + c + 2), (const d2*)(b.sumsq + c), (const d2*)(b.sumsq + c + 2)};
          rep_add(p, b.nrep, (size_t)b.rep_stride, sv); }
        const d2 s01 = sv[0], s23 = sv[1], q01 = sv[2], q23 = sv[3];
        const double ic = (double)b.inv_count;
        const double m0 = s01.x * ic, m1 = s01.y * ic, m2 = s23.x * ic, m3 = s23.y * ic;
        const double v0 = q01.x * ic - m0 * m0, v1 = q01.y * ic - m1 * m1, v2 = q23.x * ic - m2 * m2, v3 = q23.y * ic - m3 * m3;
        mu4 = make_float4((float)m0, (float)m1, (float)m2, (float)m3);
        rs4 = make_float4(1.0f / sqrtf((float)(v0 > 0.0 ? v0 : 0.0) + b.eps), 1.0f / sqrtf((float)(v1 > 0.0 ? v1 : 0.0) + b.eps),
                          1.0f / sqrtf((float)(v2 > 0.0 ? v2 : 0.0) + b.eps), 1.0f / sqrtf((float)(v3 > 0.0 ? v3 : 0.0) + b.eps));
    } else {
        const float4 rv = *(const float4*)(b.rvar + c);
        mu4 = *(const float4*)(b.rmean + c);
        rs4 = make_float4(1.0f / sqrtf(rv.x + b.eps), 1.0f / sqrtf(rv.y + b.eps), 1.0f / sqrtf(rv.z + b.eps), 1.0f / sqrtf(rv.w + b.eps));
    }
    mean[0] = mu4.x; mean[1] = mu4.y; mean[2] = mu4.z; mean[3] = mu4.w;
    sc[0] = g.x * rs4.x; sc[1] = g.y * rs4.y; sc[2] = g.z * rs4.z; sc[3] = g.w * rs4.w;
    beta[0] = be.x; beta[1] = be.y; beta[2] = be.z; beta[3] = be.w;
}
// host-side check for the kernels that read BatchNorm parameter blocks with 16-byte vector loads (bn_consts4)
static inline bool mms_bn_aligned16(const BnSrc& b) {
    const uintptr_t a = b.train ? ((uintptr_t)b.sum | (uintptr_t)b.sumsq | (uintptr_t)((size_t)b.rep_stride * 8 * (b.nrep > 1)))
                                : ((uintptr_t)b.rmean | (uintptr_t)b.rvar);
    return ((a | (uintptr_t)b.gamma | (uintptr_t)b.beta) & 15) == 0;
}

// BN constants of channels tid, tid+256, ... (< C <= 256*NJ) into LDS arrays; all global loads are issued before any
// dependent math so a workgroup pays ONE memory round trip for its constants instead of NJ serialized ones.
template <int NJ>
__device__ __forceinline__ void bn_consts_to_lds(const BnSrc& b, int C, int tid, float* o_mean, float* o_sc, float* o_beta) {
#ifdef MMS_ABLATE_SETUP
    return;
#endif
    if (!b.gamma) {                            // identity
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const int c = tid + 256 * j; if (c < C) { o_mean[c] = 0.f; o_sc[c] = 1.f; o_beta[c] = 0.f; } }
        return;
    }
    double s[NJ], q[NJ];
    float g[NJ], be[NJ], rm[NJ], rv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {             // every base load first (no loop between them: one round trip) ...
        const int c = tid + 256 * j, cc = c < C ? c : C - 1;
        if (b.train) { s[j] = b.sum[cc]; q[j] = b.sumsq[cc]; rm[j] = 0.f; rv[j] = 0.f; }
        else { rm[j] = b.rmean[cc]; rv[j] = b.rvar[cc]; s[j] = 0; q[j] = 0; }
        g[j] = b.gamma[cc]; be[j] = b.beta[cc];
    }
    if (b.train && b.nrep > 1) {               // ... then the replicas, if any (three at a time, rep_add)
        double sq[2 * NJ];
        const double* pp[2 * NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = tid + 256 * j, cc = c < C ? c : C - 1;
            sq[2 * j] = s[j]; sq[2 * j + 1] = q[j]; pp[2 * j] = b.sum + cc; pp[2 * j + 1] = b.sumsq + cc;
        }
        const double* const (&pr)[2 * NJ] = pp;
        rep_add(pr, b.nrep, (size_t)b.rep_stride, sq);
#pragma unroll
        for (int j = 0; j < NJ; ++j) { s[j] = sq[2 * j]; q[j] = sq[2 * j + 1]; }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = tid + 256 * j;
        float mu, rstd;
        if (b.train) {
            const double m = s[j] * (double)b.inv_count;
            double v = q[j] * (double)b.inv_count - m * m;
            v = v > 0.0 ? v : 0.0;
            mu = (float)m; rstd = 1.0f / sqrtf((float)v + b.eps);
        } else {
            mu = rm[j]; rstd = 1.0f / sqrtf(rv[j] + b.eps);
        }
        if (c < C) { o_mean[c] = mu; o_sc[c] = g[j] * rstd; o_beta[c] = be[j]; }
    }
}

// y = (x - mean) * (gamma*rstd) + beta, the centred form torch uses (no large-mean cancellation).
__device__ __forceinline__ float bn_apply(float x, float mean, float sc, float beta) {
    return fmaf(x - mean, sc, beta);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum of a double over 256 threads; result valid in every thread. red: >= 4 doubles of LDS.
__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += red[i];
    return r;
}

// Voxel coordinate table entry: (d, h, w) packed 10 bits each + batch index in the high bits is not needed
// (neighbours never cross a sample: bounds are checked per axis).
__device__ __forceinline__ int pack_dhw(int d, int h, int w) { return (d << 20) | (h << 10) | w; }
__device__ __forceinline__ void unpack_dhw(int c, int& d, int& h, int& w) {
    d = c >> 20; h = (c >> 10) & 1023; w = c & 1023;
}


// dropout keep-mask from a counter hash (perf mode).  Parity mode passes an explicit mask instead.
__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float dropout_scale(uint32_t seed, uint32_t stream, uint32_t idx, float p) {
    // returns 0 (dropped) or 1/(1-p) (kept)
    uint32_t h = hash_u32(idx * 0x9E3779B9U + hash_u32(seed ^ (stream * 0x85ebca6bU)));
    float u = (float)(h >> 8) * (1.0f / 16777216.0f);
    return u < p ? 0.0f : 1.0f / (1.0f - p);
}

// raw buffer loads: 128-bit resource in SGPRs + 32-bit byte offsets -> no 64-bit per-lane address arithmetic, and
// a wave-uniform scalar offset (soffset) for the per-tile part of the address
typedef __amdgpu_buffer_rsrc_t buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(buf_rsrc_t r, int voff, int soff) {
    auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return *(float4*)&v;
}

// Launch with a clean error slate: hipGetLastError() is sticky per thread and the host framework may leave benign
// errors behind (e.g. attribute probes), which must not be reported as OUR launch failing -- but an error that was already
// pending is not dropped silently either: it is logged once per distinct code (it may be a previous kernel's asynchronous fault).
static inline void mms_note_pending_error(const char* where) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return;
    static std::atomic<int> last{0};
    if (last.exchange((int)e) != (int)e)
        fprintf(stderr, "mmsurv: note: HIP error '%s' was pending before the launch at %s (not raised by this launch)\n", hipGetErrorString(e), where);
}
#define MMS_STR2(x) #x
#define MMS_STR(x) MMS_STR2(x)
#define MMS_LAUNCH(...) do { mms_note_pending_error(__FILE__ ":" MMS_STR(__LINE__)); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

static inline int mms_check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) fprintf(stderr, "mmsurv: kernel launch failed: %s\n", hipGetErrorString(e));
    return e == hipSuccess ? MMS_OK : MMS_ERR_LAUNCH;
}

// Device tables shared by the drivers (dn_net.hip, fallback.hip) and the kernels that walk them.
struct PackEntry { const float* w; float* wpf; float* wpb; };
struct UnpackEntry { const float* scratch; float* dw; };
struct BnRunEntry { const double* sum; const double* sumsq; float* rmean; float* rvar; long long* nbt; int C; float count; int nrep; int rep_stride; };
