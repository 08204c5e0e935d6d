// Dense blocks with few voxels per sample as ONE launch per pass: "cluster" kernels (round 4; generalises round 2's block-4 kernel).
//
// Why: at <= 32 voxels per sample a dense layer is 5-80 MFLOP, yet its per-layer launch sequence costs 18-24 us of dependent-launch
// latency per layer and pass (profiles/r03_d_group1_step_breakdown.txt: block 3 = 24 x (9.5 + 9.1) us forward).  Here a CLUSTER of 8
// workgroups owns a few whole samples (16 * RT rows: RT = 1 -> block 4 of 64x64x32 volumes, all 4 samples x 4 voxels in one cluster;
// RT = 2 -> block 3, one sample of 32 voxels per cluster, 4 clusters per model) and walks the block's layers inside one launch:
//   * every workgroup keeps its cluster's whole activation slab on chip: the first 16 rows IN REGISTERS, already in the MFMA A-fragment
//     layout (lane (r, k4) of wave v holds row r, channels 16 (v + 4 i) + 4 k4 .. + 3: 64 VGPRs, no operand reads for conv1), the second
//     row tile (RT = 2) in LDS (66 KB; 128 more VGPRs for it spilled ~100 registers);
//   * work is split over OUTPUT CHANNELS (conv1: 16 of the 128 per workgroup) and TAPS (conv2: the live taps dealt over the
//     workgroups), never over a cluster's rows;
//   * per layer two hand-offs inside the cluster -- A: all gather relu(bn2(y1)) (16 RT x 128), B: all gather the 8 partial conv2
//     outputs (16 RT x 32 each) and add them in fixed order -- and, when the batch spans several clusters, two small exchanges of
//     BatchNorm partial sums between the workgroups that own the same channels in the other clusters (summed in cluster order: every
//     workgroup computes bit-identical statistics).
// Hand-offs are GRANULES (cdna_hip_programming.md, Guideline 16, R2): every float travels as ONE aligned 8-byte {tag, value} word
// written by an agent-scope (sc1) store; a consumer lane re-reads its granules with agent-scope loads until every tag is the phase's
// tag -- no counter, no drain, no producer-side barrier, one memory round trip (measured on the block-4 kernel with -DB4_TIMING:
// 6.0 + 1.8 us per layer in the two counter hand-offs of round 3 -> 1.25 + 1.25 us).  STATE: all granule buffers are zeroed before
// every launch (the forward driver's per-step zero-fill); tags are >= 1 and unique per (layer, hand-off) within a launch.  Buffer
// reuse is safe because a workgroup publishes hand-off k + 2 into a buffer only after it has consumed hand-off k + 1 from ALL
// producers, each of which published that only after consuming hand-off k.  Every sweep is BOUNDED (wall clock ~0.3 s, or another
// workgroup's raised error word): a cluster whose workgroups never became co-resident leaves the sticky error word set and returns.
// Weights never depend on activations: a layer's conv1 slice, conv2 tap slices and BatchNorm parameters are requested right behind
// the previous layer's second hand-off and land under that layer's statistics; no bulk load is in flight while a wave sweeps
// (a wave's loads return in order).
#include "dn_ops.h"

namespace {

constexpr int CLW = 8;            // workgroups per cluster
constexpr int CLA2P = 132;        // LDS pitch of the gathered relu(bn2(y1)) rows (floats)
constexpr int CLZP = 36;          // LDS pitch of the summed conv2 output rows
constexpr int CLXP = 1028;        // LDS pitch of the second row tile's slab image: rows 16 B apart -> conflict-free ds_read_b128

typedef unsigned long long cl_u64;
__device__ __forceinline__ void g_store(cl_u64* g, unsigned tag, unsigned bits) {
    __hip_atomic_store(g, ((cl_u64)tag << 32) | (cl_u64)bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void g_store(cl_u64* g, unsigned tag, float v) { g_store(g, tag, __float_as_uint(v)); }
__device__ __forceinline__ void g_store_d(cl_u64* g, unsigned tag, double v) {      // a double = two granules
    const cl_u64 b = (cl_u64)__double_as_longlong(v);
    g_store(g, tag, (unsigned)b); g_store(g + 1, tag, (unsigned)(b >> 32));
}
// The lanes of the calling WAVE with `active` set sweep their N granules g[(k / INNER) * S + (k % INNER) * SI] until all tags match.
// -> false on time-out / raised error word (the word is raised on time-out), wave-uniform.
template <int N, int INNER = 1>
__device__ __forceinline__ bool g_sweep(const cl_u64* g, int S, unsigned tag, unsigned (&v)[N], unsigned* err, bool active = true, int SI = 0) {
    const unsigned long long t0 = wall_clock64();
    for (unsigned spins = 1;; ++spins) {
        bool ok = true;
        if (active) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const cl_u64 x = __hip_atomic_load(g + (size_t)(k / INNER) * S + (size_t)(k % INNER) * SI, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v[k] = (unsigned)x;
                ok = ok && (unsigned)(x >> 32) == tag;
            }
        }
        if (__all(ok)) return true;
        if ((spins & 31u) == 0u) {
            const bool late = wall_clock64() - t0 > 30000000ull;          // 100 MHz ticks
            if (late || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                if (late) atomicExch(err, 1u);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// Sum of per-channel (sum, sumsq) pairs over the clusters of a model.  On entry EVERY thread holds its channel's (tid % NCH) cluster
// totals; on exit the totals over all clusters, added in cluster order (bit-identical in every workgroup and cluster).  slots: this
// channel group's [ncl][NCH][4] granules (a double = two granules); scratch: 2 * NCH * ncl doubles of LDS.  Two barriers.
template <int NCH>
__device__ __forceinline__ bool cluster_sum(double& s, double& q, cl_u64* slots, int cl, int ncl, unsigned tag, double* scratch, unsigned* err, int tid) {
    if (tid < NCH) { g_store_d(slots + ((size_t)cl * NCH + tid) * 4, tag, s); g_store_d(slots + ((size_t)cl * NCH + tid) * 4 + 2, tag, q); }
    unsigned v[4];
    const int c2 = tid / NCH, ch = tid % NCH;
    const bool ok = g_sweep<4>(slots + ((size_t)c2 * NCH + ch) * 4, 1, tag, v, err, c2 < ncl);
    if (c2 < ncl) {
        scratch[(c2 * NCH + ch) * 2] = __longlong_as_double((long long)(((cl_u64)v[1] << 32) | v[0]));
        scratch[(c2 * NCH + ch) * 2 + 1] = __longlong_as_double((long long)(((cl_u64)v[3] << 32) | v[2]));
    }
    __syncthreads();
    s = 0.0; q = 0.0;
    for (int c = 0; c < ncl; ++c) { s += scratch[(c * NCH + ch) * 2]; q += scratch[(c * NCH + ch) * 2 + 1]; }
    __syncthreads();          // scratch may be rewritten
    return ok;
}

template <int RT>
__global__ __launch_bounds__(256) void cl_fwd_kernel(const Grp<ClFwdP> grp) {
    const ClFwdP& p = grp.p[blockIdx.z];
    constexpr int R = 16 * RT;
    const int w = blockIdx.x, cl = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, k4 = lane >> 4;
    const int ncl = p.ncl, row0 = cl * p.rpc;
    const int nrows = p.M - row0 < p.rpc ? p.M - row0 : p.rpc;            // rows of this cluster (<= R); the statistics span all p.M rows
    extern __shared__ __attribute__((aligned(16))) float smem[];
    B4Layer* tabs = (B4Layer*)smem;            // [<= 24] the block's layer table (pointer reads from LDS, not through a dependent global load)
    float* mu = smem + 768;                    // [1024] batch mean of every slab channel (train)
    float* rs = mu + 1024;                     // [1024] batch rstd
    float* mn1 = rs + 1024;                    // [1024] norm1 of the current layer: mean | gamma * rstd | beta
    float* sc1 = mn1 + 1024;
    float* be1 = sc1 + 1024;
    float* a2s = be1 + 1024;                   // [R][CLA2P] gathered relu(bn2(y1))
    float* red = a2s + R * CLA2P;              // [RT][4][256] cross-wave sums
    float* zs = red + RT * 1024;               // [R][CLZP] the layer's summed conv2 output, rows 0 .. 15 (on their way into the register slab)
    double* dred = (double*)(zs + R * CLZP);   // [1024] statistic partials: [0, 256) the four waves, [256, 768) the clusters
    int* nbt = (int*)(dred + 1024);             // [27][R] neighbour row (cluster-local) of (tap, row), -1 = zero padding
    int* live = nbt + 27 * R;                  // [0] = number of live taps, [1..27] = their indices, [31] = "a sweep timed out"
    int& s_fail = live[31];
    float* xs1 = (float*)(live + 32);          // RT = 2: [16][CLXP] raw slab rows 16 .. 31 of the cluster (rows beyond the cluster's stay zero)
    const int M = p.M, C0 = p.C0, ld = p.ld;
    const float inv_m = 1.0f / (float)M;
    const double inv_md = (double)inv_m;
    const int row = tid >> 4, col = tid & 15;
    const int n0 = 16 * w;                      // this workgroup's conv1 output channels
    // hand-off buffers of this cluster / of this workgroup's channel peers in the other clusters
    cl_u64* xa = p.xa + (size_t)cl * (CLW * RT * 256);
    cl_u64* xb = p.xb + (size_t)cl * (CLW * RT * 512);
    cl_u64* gsy = p.gst + (size_t)w * ncl * 64;                           // [ncl][16 channels][sum, sumsq as 2 granules each]
    cl_u64* gsz = p.gst + (size_t)CLW * ncl * 64 + (size_t)w * ncl * 128; // [ncl][32 channels][4]

    // ---- set-up -----------------------------------------------------------------------------------------------------------
    static_assert(sizeof(B4Layer) == 120, "layer table entry");
    for (int i = tid; i < p.nlayers * 30; i += 256) ((unsigned*)tabs)[i] = ((const unsigned*)p.tab)[i];
    if (tid == 0) s_fail = 0;
    // the on-chip slab: channels [0, C0) from the global slab (rows beyond the cluster's stay zero)
    float4 xreg[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int t = wave + 4 * i;
        xreg[i] = (16 * t < C0 && r16 < nrows) ? *(const float4*)(p.slab + (size_t)(row0 + r16) * ld + 16 * t + 4 * k4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (RT == 2) {      // every load of a batch of 8 float4 per thread is issued before the first LDS store
        const int n4 = C0 >> 2, tot4 = 16 * n4;
        for (int base = 0; base < tot4; base += 2048) {
            float4 r[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i4 = base + tid + 256 * j, m = i4 / n4, k = (i4 - m * n4) << 2;
                r[j] = (i4 < tot4 && 16 + m < nrows) ? *(const float4*)(p.slab + (size_t)(row0 + 16 + m) * ld + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i4 = base + tid + 256 * j, m = i4 / n4, k = (i4 - m * n4) << 2;
                if (i4 < tot4) *(float4*)(xs1 + m * CLXP + k) = r[j];
            }
        }
    }
    if (p.train) {
        double sv[4], qv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int k = tid + 256 * j, kk = k < C0 ? k : 0; sv[j] = p.st_slab[kk]; qv[j] = p.st_slab[ld + kk]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = tid + 256 * j;
            const double m_ = sv[j] * inv_md;
            double v = qv[j] * inv_md - m_ * m_;
            v = v > 0.0 ? v : 0.0;
            if (k < C0) { mu[k] = (float)m_; rs[k] = 1.0f / sqrtf((float)v + p.eps); }
        }
    }
    for (int idx = tid; idx < 27 * R; idx += 256) {
        const int tap = idx / R, m = idx - tap * R;
        int nb = -1;
        if (m < nrows) {
            int d, h, x;
            unpack_dhw(p.coords[row0 + m], d, h, x);
            const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
            const int nd = d + kd - 1, nh = h + kh - 1, nw = x + kw - 1;
            if ((unsigned)nd < (unsigned)p.g.D && (unsigned)nh < (unsigned)p.g.H && (unsigned)nw < (unsigned)p.g.W)
                nb = m + ((kd - 1) * p.g.H + (kh - 1)) * p.g.W + (kw - 1);
        }
        nbt[idx] = nb;
    }
    __syncthreads();
    if (tid == 0) {
        int n = 0;
        for (int tap = 0; tap < 27; ++tap) {
            bool any = false;
            for (int m = 0; m < nrows; ++m) any = any || nbt[tap * R + m] >= 0;
            if (any) live[1 + n++] = tap;
        }
        live[0] = n;
    }
    __syncthreads();
    const int nlive = live[0];

    const int nt = wave & 1, half = wave >> 1;
    float4 wreg[16];
    float4 treg[4][4];
    int mytap[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int li = w + CLW * i; mytap[i] = li < nlive ? live[1 + li] : -1; }
    auto load_w1 = [&](int l) __attribute__((always_inline)) {
        const int C = C0 + 32 * l, nT = C >> 4;
        const float* w1 = tabs[l].w1;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int t = wave + 4 * i;
            wreg[i] = t < nT ? *(const float4*)(w1 + (size_t)(n0 + r16) * C + 16 * t + 4 * k4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto load_taps = [&](int l) __attribute__((always_inline)) {
        const float* wpf = tabs[l].wpf;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                treg[i][t] = mytap[i] >= 0 ? *(const float4*)(wpf + ((size_t)(16 * nt + r16) * 27 + mytap[i]) * 128 + 64 * half + 16 * t + 4 * k4)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // norm1 parameters of the layer's C channels (<= 4 per thread): gamma | beta [| running mean | running var]; norm2's of this
    // workgroup's 16 channels (the thread's column): gamma | beta [| running mean | running var]
    float cg[4], cb[4], cm[4], cv[4], g2v, b2v, m2v, v2v;
    auto load_c1 = [&](int l) __attribute__((always_inline)) {
        const int C = C0 + 32 * l;
        const B4Layer& T = tabs[l];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = tid + 256 * j, kk = k < C ? k : C - 1;
            cg[j] = T.g1[kk]; cb[j] = T.b1[kk];
            if (!p.train) { cm[j] = T.rm1[kk]; cv[j] = T.rv1[kk]; } else { cm[j] = 0.f; cv[j] = 1.f; }
        }
        g2v = T.g2[n0 + col]; b2v = T.b2[n0 + col];
        if (!p.train) { m2v = T.rm2[n0 + col]; v2v = T.rv2[n0 + col]; } else { m2v = 0.f; v2v = 1.f; }
    };
    load_w1(0);
    load_c1(0);
    load_taps(0);

#ifdef B4_TIMING
    unsigned long long tacc[7] = {0, 0, 0, 0, 0, 0, 0}, tl = wall_clock64();
#define CL_T(i) do { if (tid == 0) { const unsigned long long n_ = wall_clock64(); tacc[i] += n_ - tl; tl = n_; } } while (0)
#else
#define CL_T(i)
#endif
    for (int l = 0; l < p.nlayers; ++l) {
        const int C = C0 + 32 * l, nT = C >> 4;                 // K super-steps of 16 channels
        const B4Layer& L = tabs[l];
        const unsigned tag0 = 4u * (unsigned)l + 1u;            // + 0: y1 statistics | 1: hand-off A | 2: hand-off B | 3: z statistics
        CL_T(0);
        // a. norm1 constants of this layer for the C input channels (train: the channels' batch statistics, cached since they were
        //    produced; eval: this layer's running statistics); the transform itself rides in the MFMA loop
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = tid + 256 * j;
            if (k < C) {
                float m_, r_;
                if (p.train) { m_ = mu[k]; r_ = rs[k]; } else { m_ = cm[j]; r_ = 1.0f / sqrtf(cv[j] + p.eps); }
                mn1[k] = m_; sc1[k] = cg[j] * r_; be1[k] = cb[j];
            }
        }
        __syncthreads();
        CL_T(1);
        // b. conv1: RT x 16 rows x 16 channels, K split over the waves; A = relu(bn1(x)) built from the register slab on the fly
        f32x4 acc[RT];
        float rowz[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) { acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f}; rowz[rt] = 16 * rt + r16 < nrows ? 1.f : 0.f; }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int t = wave + 4 * i;
            if (t < nT) {
                const int k = 16 * t + 4 * k4;
                const float4 m4 = *(const float4*)(mn1 + k), s4 = *(const float4*)(sc1 + k), b4 = *(const float4*)(be1 + k);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const float4 x = rt == 0 ? xreg[i] : *(const float4*)(xs1 + r16 * CLXP + k);
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(rowz[rt] * fmaxf(bn_apply(x.x, m4.x, s4.x, b4.x), 0.f), wreg[i].x, acc[rt], 0, 0, 0);
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(rowz[rt] * fmaxf(bn_apply(x.y, m4.y, s4.y, b4.y), 0.f), wreg[i].y, acc[rt], 0, 0, 0);
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(rowz[rt] * fmaxf(bn_apply(x.z, m4.z, s4.z, b4.z), 0.f), wreg[i].z, acc[rt], 0, 0, 0);
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(rowz[rt] * fmaxf(bn_apply(x.w, m4.w, s4.w, b4.w), 0.f), wreg[i].w, acc[rt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[rt * 1024 + wave * 256 + (4 * k4 + r) * 16 + r16] = acc[rt][r];
        __syncthreads();
        float y[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
            y[rt] = red[rt * 1024 + tid] + red[rt * 1024 + 256 + tid] + red[rt * 1024 + 512 + tid] + red[rt * 1024 + 768 + tid];   // y1[16 rt + row][n0 + col]
        // c. BatchNorm2 statistics of the 16 channels: the cluster's rows by lane shuffles + LDS, the other clusters' partial sums by a
        //    granule exchange with the workgroups that own these channels there; y1 + statistics saved for the backward
        float m_, r_;
        if (p.train) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) { const double v = 16 * rt + row < nrows ? (double)y[rt] : 0.0; s += v; q += v * v; }
            s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
            s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
            if (lane < 16) { dred[wave * 32 + lane] = s; dred[128 + wave * 32 + lane] = q; }
            __syncthreads();
            s = ((dred[col] + dred[32 + col]) + dred[64 + col]) + dred[96 + col];
            q = ((dred[128 + col] + dred[160 + col]) + dred[192 + col]) + dred[224 + col];
            if (ncl > 1 && !cluster_sum<16>(s, q, gsy, cl, ncl, tag0, dred + 256, p.err, tid)) s_fail = 1;
            const double mm = s * inv_md;
            double var = q * inv_md - mm * mm;
            var = var > 0.0 ? var : 0.0;
            m_ = (float)mm; r_ = 1.0f / sqrtf((float)var + p.eps);
            if (tid < 16 && cl == 0) { L.st_y1[n0 + tid] = s; L.st_y1[128 + n0 + tid] = q; }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                if (16 * rt + row < nrows) L.y1[(size_t)(row0 + 16 * rt + row) * 128 + n0 + col] = y[rt];
        } else {
            m_ = m2v; r_ = 1.0f / sqrtf(v2v + p.eps);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
            g_store(xa + (w * RT + rt) * 256 + tid, tag0 + 1u, 16 * rt + row < nrows ? fmaxf(bn_apply(y[rt], m_, g2v * r_, b2v), 0.f) : 0.f);
        // ---- hand-off A: gather the cluster's R x 128 relu(bn2(y1)) -----------------------------------------------------------
        CL_T(2);
        {
            unsigned v[8 * RT];
            const bool ok = g_sweep<8 * RT>(xa + tid, 256, tag0 + 1u, v, p.err);      // granule k = (j * RT + rt): workgroup j, row tile rt
#pragma unroll
            for (int k = 0; k < 8 * RT; ++k) a2s[(16 * (k % RT) + row) * CLA2P + 16 * (k / RT) + col] = __uint_as_float(v[k]);
            if (!ok) s_fail = 1;
        }
        CL_T(3);
        __syncthreads();
        if (s_fail) return;
        // e. conv2: this workgroup's live taps; wave = (output-channel tile nt, input-channel half)
        f32x4 zc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) zc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (mytap[i] >= 0) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int nb = nbt[mytap[i] * R + 16 * rt + r16];
                    const float* ar = a2s + (nb >= 0 ? nb : 0) * CLA2P + 64 * half + 4 * k4;
                    const float z = nb >= 0 ? 1.f : 0.f;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float4 av = *(const float4*)(ar + 16 * t);
                        zc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x * z, treg[i][t].x, zc[rt], 0, 0, 0);
                        zc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y * z, treg[i][t].y, zc[rt], 0, 0, 0);
                        zc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z * z, treg[i][t].z, zc[rt], 0, 0, 0);
                        zc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w * z, treg[i][t].w, zc[rt], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[rt * 1024 + wave * 256 + (4 * k4 + r) * 16 + r16] = zc[rt][r];      // [rt][wave][row][co within the tile]
        __syncthreads();
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int j = 0; j < 2; ++j)          // partial z[16 rt + row][16 j + col]: co tile nt = j, input-channel halves summed (waves j and j + 2)
                g_store(xb + (w * RT + rt) * 512 + row * 32 + 16 * j + col, tag0 + 2u, red[rt * 1024 + j * 256 + tid] + red[rt * 1024 + (j + 2) * 256 + tid]);
        // ---- hand-off B: gather the 8 partial R x 32 outputs ------------------------------------------------------------------
        CL_T(4);
        float zsum[RT][2];
        {   // one sweep per row tile (16 granules per lane each: 32 at once cost ~100 spilled registers at RT = 2); the second one
            // normally finds its granules there
            // granule (workgroup q, row tile rt, element e = tid + 256 j) sits at ((q RT + rt) * 2 + j) * 256 + tid
            bool ok = true;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                unsigned v[16];
                ok = g_sweep<16, 2>(xb + rt * 512 + tid, 512 * RT, tag0 + 2u, v, p.err, true, 256) && ok;      // k = 2 q + j: q * (RT * 512) + j * 256
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float z = __uint_as_float(v[j]);
#pragma unroll
                    for (int q = 1; q < 8; ++q) z += __uint_as_float(v[2 * q + j]);      // fixed order: deterministic, identical in every workgroup
                    zsum[rt][j] = z;
                }
            }
            if (!ok) s_fail = 1;
        }
        CL_T(5);
        if (l + 1 < p.nlayers) { load_w1(l + 1); load_c1(l + 1); load_taps(l + 1); }      // next layer's weights: ~2 us / ~6 us ahead of their use
        {
            double s = 0.0, q = 0.0;          // the thread's 2 RT elements all belong to channel tid & 31
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int e = tid + 256 * j, zr = 16 * rt + (e >> 5), co = e & 31;
                    const float z = zr < nrows ? zsum[rt][j] : 0.f;
                    if (rt == 0) zs[zr * CLZP + co] = z; else xs1[(zr - 16) * CLXP + C + co] = z;
                    if (w == 0 && zr < nrows) p.slab[(size_t)(row0 + zr) * ld + C + co] = z;
                    s += (double)z; q += (double)z * (double)z;
                }
            if (p.train) {
                s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
                if (lane < 32) { dred[wave * 32 + lane] = s; dred[128 + wave * 32 + lane] = q; }
            }
        }
        __syncthreads();
        if (s_fail) return;
        {   // the 32 new channels enter the register slab (rows 0 .. 15): columns C .. C + 31 = K super-steps t0 = C / 16 (even) and
            // t0 + 1, held by waves t0 & 3 and (t0 & 3) + 1 in fragment i = t0 >> 2
            const int t0 = C >> 4, i0 = t0 >> 2, h = wave - (t0 & 3);
            if (h == 0 || h == 1) {
                const float4 v = *(const float4*)(zs + r16 * CLZP + 16 * h + 4 * k4);
#pragma unroll
                for (int i = 0; i < 16; ++i) if (i == i0) xreg[i] = v;
            }
        }
        if (p.train) {
            const int co = tid & 31;
            double s = ((dred[co] + dred[32 + co]) + dred[64 + co]) + dred[96 + co];
            double q = ((dred[128 + co] + dred[160 + co]) + dred[192 + co]) + dred[224 + co];
            if (ncl > 1 && !cluster_sum<32>(s, q, gsz, cl, ncl, tag0 + 3u, dred + 256, p.err, tid)) s_fail = 1;
            if (tid < 32) {
                const double mm = s * inv_md;
                double v = q * inv_md - mm * mm;
                v = v > 0.0 ? v : 0.0;
                mu[C + tid] = (float)mm; rs[C + tid] = 1.0f / sqrtf((float)v + p.eps);
                if (w == 0 && cl == 0) { p.st_slab[C + tid] = s; p.st_slab[ld + C + tid] = q; }
            }
        }
        __syncthreads();          // mu / rs of the new channels feed the next layer's norm1 constants; zs / dred are free again
        if (s_fail) return;
        CL_T(6);
    }
#ifdef B4_TIMING
    if (tid == 0 && cl == 0) for (int i = 0; i < 7; ++i) p.err[8 + 64 * (RT - 1) + 8 * w + i] = (unsigned)tacc[i];      // 100 MHz ticks summed over the layers, per workgroup (RT = 2: second table)
#endif
}


// ---------------------------------------------------------------------------------------------------------------------------
// Backward data path of a single-cluster block (dense block 4 with <= 16 rows): the chain dslab -> relu2/norm2/conv2 -> relu1/norm1/conv1
// -> dslab of the layers nl-1 .. 0 as one launch; the weight gradients stay with the batched launches the network driver issues at the
// end of the block (they read what this kernel leaves per layer: the masked gradient at norm2's output `dmid`, its BatchNorm-backward
// sums, the final dz columns of dslab).  Round 4 form: granule hand-offs and FIXED column ownership --
//   * workgroup w owns slab columns [128 w, 128 w + 128) for the whole launch: their saved activations, batch statistics and the running
//     gradient dslab[:, own] stay in REGISTERS (8 rows per thread), so norm1's in-place update never leaves the chip; the gradient slab is
//     read once at the start and written once at the end (round 3: a per-layer column split of C/8, the slab exchanged through memory
//     behind a counter barrier every layer);
//   * per layer: (dz) the layer's 16 x 32 output gradient = the owner's current dslab columns [C, C + 32), broadcast as granules by the
//     workgroup that owns them (the last layer reads the incoming slab directly); (A) conv2 backward-data + relu2 mask + norm2 backward
//     for the own 16 mid channels (all rows are here: sums are local), dy1 published, all gather 16 x 128; (B) conv1 backward-data for
//     the own columns below C (one float4 of W1 = 4 consecutive columns feeds 4 MFMAs with interleaved output columns), relu1 mask,
//     norm1 backward applied to the register gradient.
// Hand-off A's buffer is double-buffered by layer parity (the dz broadcast has ONE producer, so it does not order the other workgroups'
// reads of the previous A); tags: A = l + 1, dz of layer l = l + 1, in separate buffers, zeroed with the step's statistics.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int CLDZP = 36, CLDYP = 132;

__global__ __launch_bounds__(256) void cl_bwd_kernel(const Grp<ClBwdP> grp) {
    const ClBwdP& p = grp.p[blockIdx.z];
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, k4 = lane >> 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    B4Layer* tabs = (B4Layer*)smem;            // [<= 24] the block's layer table
    float* dzs = smem + 768;                   // [16][CLDZP] dz of the layer (rows >= M zero)
    float* dys = dzs + 16 * CLDZP;             // [16][CLDYP] gathered dy1
    float* da = dys + 16 * CLDYP;              // [2][16][CLDYP] d(a1) partials of the two halves of the reduction
    float* red = da + 2 * 16 * CLDYP;          // [4][256] cross-wave sums
    double* dred = (double*)(red + 1024);      // [1024] column-sum partials
    float* c2 = (float*)(dred + 1024);         // [64]: norm2 mean | rstd | s1/M | s2/M of this workgroup's 16 channels
    int* nbm = (int*)(c2 + 64);                // [27][16] row whose output tap `tap` reads this row (row - off(tap)), -1 = outside the grid
    int* live = nbm + 27 * 16;                 // [0] = number of live taps, [1..27] = their indices, [31] = "a sweep timed out"
    int& s_fail = live[31];
    const int M = p.M, C0 = p.C0, ld = p.ld;
    const float inv_m = 1.0f / (float)M;
    const double inv_md = (double)inv_m;

    for (int i = tid; i < p.nlayers * 30; i += 256) ((unsigned*)tabs)[i] = ((const unsigned*)p.tab)[i];
    if (tid == 0) s_fail = 0;
    const int row = tid >> 4, col = tid & 15;                 // phase A: element (row, mid channel 16 w + col)
    const int bcol = tid & 127, brg = tid >> 7;                // phase B: column 128 w + bcol, rows 8 brg .. 8 brg + 7
    const int c = 128 * w + bcol;
    const int T = wave & 1, kh2 = wave >> 1;                   // phase B MFMA: 64-column half, half of the 128-long reduction
    // the own columns: saved activations, incoming gradient, batch statistics -- loaded once, resident for the launch
    float xv[8], dv[8], mu1, r1;
    {
        const double s_ = p.st_slab[c], q_ = p.st_slab[ld + c];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = 8 * brg + i;
            xv[i] = m < M ? p.slab[(size_t)m * ld + c] : 0.f;
            dv[i] = m < M ? p.dslab[(size_t)m * ld + c] : 0.f;
        }
        const double mm = s_ * inv_md;
        double v = q_ * inv_md - mm * mm;
        v = v > 0.0 ? v : 0.0;
        mu1 = (float)mm; r1 = 1.0f / sqrtf((float)v + p.eps);
    }
    for (int idx = tid; idx < 27 * 16; idx += 256) {
        const int tap = idx >> 4, m = idx & 15;
        int nb = -1;
        if (m < M) {
            int d, h, x;
            unpack_dhw(p.coords[m], d, h, x);
            const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
            const int nd = d - (kd - 1), nh = h - (kh - 1), nw = x - (kw - 1);
            if ((unsigned)nd < (unsigned)p.g.D && (unsigned)nh < (unsigned)p.g.H && (unsigned)nw < (unsigned)p.g.W)
                nb = m - (((kd - 1) * p.g.H + (kh - 1)) * p.g.W + (kw - 1));
        }
        nbm[idx] = nb;
    }
    {   // dz of the last layer: straight from the incoming gradient slab (nothing in this launch has written it)
        const int Cl = C0 + 32 * (p.nlayers - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) { const int e = tid + 256 * j, m = e >> 5; dzs[m * CLDZP + (e & 31)] = m < M ? p.dslab[(size_t)m * ld + Cl + (e & 31)] : 0.f; }
    }
    __syncthreads();
    if (tid == 0) {
        int n = 0;
        for (int tap = 0; tap < 27; ++tap) {
            bool any = false;
            for (int m = 0; m < M; ++m) any = any || nbm[tap * 16 + m] >= 0;
            if (any) live[1 + n++] = tap;
        }
        live[0] = n;
    }
    __syncthreads();
    const int nlive = live[0];
    int mytap[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) { const int ti = wave + 4 * j; mytap[j] = ti < nlive ? live[1 + ti] : -1; }

    // registers loaded one phase ahead of their use (weights and saved activations never depend on the chain)
    float4 treg[7][2];
    float yv, g2v, b2v; double sy, qy;
    auto load_A = [&](int l) __attribute__((always_inline)) {
        const B4Layer& L = tabs[l];
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                treg[j][t] = mytap[j] >= 0 ? *(const float4*)(L.wpb + ((size_t)(16 * w + r16) * 27 + mytap[j]) * 32 + 16 * t + 4 * k4)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
        yv = row < M ? L.y1[(size_t)row * 128 + 16 * w + col] : 0.f;
        g2v = L.g2[16 * w + col]; b2v = L.b2[16 * w + col];
        sy = L.st_y1[16 * w + col]; qy = L.st_y1[128 + 16 * w + col];
    };
    float4 wreg[16];
    float g1v, b1v;
    auto load_B = [&](int l) __attribute__((always_inline)) {
        const B4Layer& L = tabs[l];
        const int C = C0 + 32 * l;
        const bool okw = 128 * w + 64 * T + 4 * r16 < C;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int n = 4 * (16 * kh2 + i) + k4;
            wreg[i] = okw ? *(const float4*)(L.w1 + (size_t)n * C + 128 * w + 64 * T + 4 * r16) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        g1v = c < C ? L.g1[c] : 0.f; b1v = c < C ? L.b1[c] : 0.f;
    };
    load_A(p.nlayers - 1);

    for (int l = p.nlayers - 1; l >= 0; --l) {
        const int C = C0 + 32 * l;
        const B4Layer& L = tabs[l];
        const unsigned tag = (unsigned)l + 1u;
        cl_u64* ga = p.ga + (size_t)(l & 1) * (CLW * 256);
        // ---- dz of this layer: broadcast by the owner of columns [C, C + 32) after the previous layer's phase B -------------------
        if (l != p.nlayers - 1) {
            unsigned v[2];
            const bool ok = g_sweep<2>(p.gz + tid, 256, tag, v, p.err);
#pragma unroll
            for (int j = 0; j < 2; ++j) { const int e = tid + 256 * j; dzs[(e >> 5) * CLDZP + (e & 31)] = __uint_as_float(v[j]); }
            if (!ok) s_fail = 1;
        }
        load_B(l);                                // conv1 weights / norm1 parameters of this layer: used after hand-off A
        __syncthreads();
        if (s_fail) return;
        // ---- A1. conv2 backward-data for the 16 own mid channels, this wave's taps --------------------------------------------------
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            if (mytap[j] >= 0) {
                const int nb = nbm[mytap[j] * 16 + r16];
                const float* ar = dzs + (nb >= 0 ? nb : 0) * CLDZP + 4 * k4;
                const float z = nb >= 0 ? 1.f : 0.f;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float4 av = *(const float4*)(ar + 16 * t);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x * z, treg[j][t].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y * z, treg[j][t].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z * z, treg[j][t].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w * z, treg[j][t].w, acc, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave * 256 + (4 * k4 + r) * 16 + r16] = acc[r];
        if (tid < 16) {                               // norm2 batch statistics of the own channels
            const double mm = sy * inv_md;
            double v = qy * inv_md - mm * mm;
            v = v > 0.0 ? v : 0.0;
            c2[tid] = (float)mm; c2[16 + tid] = 1.0f / sqrtf((float)v + p.eps);
        }
        __syncthreads();
        // ---- A2. relu2 mask, norm2 backward sums, dy1 ----------------------------------------------------------------------------------
        const float dval = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid];
        const float mu2 = c2[col], r2 = c2[16 + col];
        const float xh2 = (yv - mu2) * r2;
        const float g = (row < M && fmaf(g2v, xh2, b2v) > 0.f) ? dval : 0.f;
        if (row < M) L.dmid[(size_t)row * 128 + 16 * w + col] = g;
        {
            double a = (double)g, b = (double)g * xh2;          // rows 4 wave .. 4 wave + 3 by shuffles, the waves through LDS
            a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
            a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
            if (lane < 16) { dred[wave * 32 + lane] = a; dred[128 + wave * 32 + lane] = b; }
        }
        __syncthreads();
        const double s1 = ((dred[col] + dred[32 + col]) + dred[64 + col]) + dred[96 + col];
        const double s2 = ((dred[128 + col] + dred[160 + col]) + dred[192 + col]) + dred[224 + col];
        if (tid < 16) { L.bb_y1[16 * w + tid] = s1; L.bb_y1[128 + 16 * w + tid] = s2; }
        g_store(ga + w * 256 + tid, tag, row < M ? (g2v * r2) * (g - (float)(s1 * inv_md) - (yv - mu2) * r2 * (float)(s2 * inv_md)) : 0.f);
        // ---- hand-off A: gather dy1 (16 x 128) ------------------------------------------------------------------------------------------
        {
            unsigned v[8];
            const bool ok = g_sweep<8>(ga + tid, 256, tag, v, p.err);
#pragma unroll
            for (int j = 0; j < 8; ++j) dys[row * CLDYP + 16 * j + col] = __uint_as_float(v[j]);
            if (!ok) s_fail = 1;
        }
        if (l > 0) load_A(l - 1);                     // next layer's conv2 weights / y1 / norm2 parameters: used after the dz broadcast
        __syncthreads();
        if (s_fail) return;
        if (128 * w < C) {          // (workgroup-uniform: the barriers below are taken by all of its threads or by none)
            // ---- B1. conv1 backward-data: d(a1)[16][own columns], 4 interleaved 16-column sets per wave, half the reduction per wave ------
            f32x4 ac4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) ac4[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float a = dys[r16 * CLDYP + 4 * (16 * kh2 + i) + k4];
                ac4[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[i].x, ac4[0], 0, 0, 0);
                ac4[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[i].y, ac4[1], 0, 0, 0);
                ac4[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[i].z, ac4[2], 0, 0, 0);
                ac4[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[i].w, ac4[3], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *(float4*)(da + kh2 * (16 * CLDYP) + (4 * k4 + r) * CLDYP + 64 * T + 4 * r16) = make_float4(ac4[0][r], ac4[1][r], ac4[2][r], ac4[3][r]);
            __syncthreads();
            // ---- B2. relu1 mask, column sums, norm1 backward applied to the register gradient ---------------------------------------------
            const bool okc = c < C;
            float gv[8];
            double t1 = 0, t2 = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = 8 * brg + i;
                const float xh = (xv[i] - mu1) * r1;
                const float gg = (okc && m < M && fmaf(g1v, xh, b1v) > 0.f) ? da[m * CLDYP + bcol] + da[16 * CLDYP + m * CLDYP + bcol] : 0.f;
                gv[i] = gg;
                t1 += gg; t2 += (double)gg * xh;
            }
            dred[256 + (brg * 2) * 128 + bcol] = t1; dred[256 + (brg * 2 + 1) * 128 + bcol] = t2;
            __syncthreads();
            if (okc) {
                const double a = dred[256 + bcol] + dred[256 + 256 + bcol], b = dred[256 + 128 + bcol] + dred[256 + 384 + bcol];
                const float gr = g1v * r1, m1 = (float)(a * inv_md), m2 = r1 * (float)(b * inv_md);
#pragma unroll
                for (int i = 0; i < 8; ++i) dv[i] += gr * (gv[i] - m1 - (xv[i] - mu1) * m2);
                if (brg == 0) { p.dg1[l][c] += (float)b; p.db1[l][c] += (float)a; }
            }
        }
        // ---- the next layer's dz = the (now final) gradient of columns [C - 32, C), broadcast by their owner ------------------------------
        if (l > 0) {
            const int Cn = C - 32, j = c - Cn;
            if (j >= 0 && j < 32) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { const int m = 8 * brg + i; g_store(p.gz + m * 32 + j, tag - 1u, m < M ? dv[i] : 0.f); }
            }
        }
    }
    // the block's gradient slab: columns [0, C0) = the block input's gradient, [C_l, C_l + 32) = layer l's final dz
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int m = 8 * brg + i; if (m < M) p.dslab[(size_t)m * ld + c] = dv[i]; }
}

}  // namespace

extern "C" int mms_cl_fwd_group(const ClFwdP* pp, int ng, hipStream_t s) {
    Grp<ClFwdP> a;
    static_assert(sizeof(Grp<ClFwdP>) <= 4096, "kernel argument block");
    if (!grp_fill(a, pp, ng, 1)) return MMS_ERR_ARG;
    const int rt = pp->rpc <= 16 ? 1 : 2;
    for (int g = 0; g < ng; ++g) {
        const ClFwdP& p = pp[g];
        if (p.M < 1 || p.rpc < 1 || p.rpc > 32 || p.ncl < 1 || p.ncl > 8 || (long)p.ncl * p.rpc < p.M || (long)(p.ncl - 1) * p.rpc >= p.M ||
            p.ld != 1024 || p.C0 % 32 != 0 || p.nlayers < 1 || p.nlayers > 24 || p.C0 + 32 * p.nlayers > p.ld || !p.tab || !p.slab || !p.xa ||
            !p.xb || !p.gst || !p.err || !p.coords || (p.train && !p.st_slab) || p.M != pp->M || p.rpc != pp->rpc || p.ncl != pp->ncl ||
            p.nlayers != pp->nlayers || p.C0 != pp->C0 || (((uintptr_t)p.slab | (uintptr_t)p.xa | (uintptr_t)p.xb | (uintptr_t)p.gst) & 15))
            return MMS_ERR_ARG;
    }
    const int R = 16 * rt;
    const int smem = (768 + 5 * 1024 + R * CLA2P + rt * 1024 + R * CLZP) * 4 + 1024 * 8 + (27 * R + 32) * 4 + (rt == 2 ? 16 * CLXP * 4 : 0);
    static std::once_flag attr_once;
    std::call_once(attr_once, [&] {
        hipFuncSetAttribute((const void*)cl_fwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        hipFuncSetAttribute((const void*)cl_fwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    });
    if (rt == 1) MMS_LAUNCH(cl_fwd_kernel<1>, dim3(CLW, pp->ncl, ng), dim3(256), smem, s, a);
    else MMS_LAUNCH(cl_fwd_kernel<2>, dim3(CLW, pp->ncl, ng), dim3(256), smem, s, a);
    return mms_check_launch();
}

extern "C" int mms_cl_bwd_group(const ClBwdP* pp, int ng, hipStream_t s) {
    Grp<ClBwdP> a;
    static_assert(sizeof(Grp<ClBwdP>) <= 4096, "kernel argument block");
    if (!grp_fill(a, pp, ng, 1)) return MMS_ERR_ARG;
    for (int g = 0; g < ng; ++g) {
        const ClBwdP& p = pp[g];
        if (p.M < 1 || p.M > 16 || p.ld != 1024 || p.C0 % 32 != 0 || p.nlayers < 1 || p.nlayers > 16 || p.C0 + 32 * p.nlayers > p.ld || !p.tab || !p.slab ||
            !p.dslab || !p.st_slab || !p.ga || !p.gz || !p.err || !p.coords || p.M != pp->M || p.nlayers != pp->nlayers || p.C0 != pp->C0 ||
            (((uintptr_t)p.ga | (uintptr_t)p.gz) & 15)) return MMS_ERR_ARG;
        for (int l = 0; l < p.nlayers; ++l) if (!p.dg1[l] || !p.db1[l]) return MMS_ERR_ARG;
    }
    constexpr int smem = (768 + 16 * CLDZP + 3 * 16 * CLDYP + 1024) * 4 + 1024 * 8 + 64 * 4 + (27 * 16 + 32) * 4;
    MMS_LAUNCH(cl_bwd_kernel, dim3(CLW, 1, ng), dim3(256), smem, s, a);
    return mms_check_launch();
}
