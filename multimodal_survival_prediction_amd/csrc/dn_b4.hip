// Dense block 4 of DenseNet121-3D as ONE launch per pass (forward here) for launches with <= 16 rows per model
// (batch 4 on 64x64x32 volumes: 4 samples x 2x2x1 voxels).
//
// Why: at 16 rows a dense layer is ~5 MFLOP, yet the per-layer launch sequence (conv1 with K-split fixup, conv2 taps, conv2
// reduce) costs ~21 us of dependent-launch latency per layer and pass (profiles/r02_b_group*_step_breakdown.txt).  Here a
// CLUSTER of 8 workgroups per model walks the 16 layers inside one launch.  Every workgroup keeps the block's whole activation
// slab (16 x 1024 fp32 = 64 KB) and the batch statistics of its channels in LDS (100 KB in all); because all rows of the block sit in every
// workgroup, work is split over OUTPUT CHANNELS / TAPS and never over rows, so each BatchNorm statistic is complete inside the
// workgroup that needs it.  Per layer two hand-offs between the 8 workgroups (tools/micro/cluster_handoff.hip: ~2.1-2.4 us each):
//   A. conv1 (norm1-relu-1x1x1 conv): workgroup w computes y1[:, 16w..16w+15] over the full K (v_mfma_f32_16x16x4_f32, K split
//      over its 4 waves), its BatchNorm2 statistics, writes y1 + statistics for the backward, and PUBLISHES relu(bn2(y1)) (1 KB);
//      all gather the 16 x 128 result.
//   B. conv2 (3x3x3, pad 1): the LIVE taps (9 of 27 on a 2x2x1 grid) are dealt over the workgroups; each PUBLISHES its partial
//      16 x 32 output (2 KB); all gather and add the 8 partials in fixed order (deterministic), append the 32 new channels to their
//      LDS slab and compute their statistics; workgroup 0 also writes them to the global slab for the rest of the network.
// Hand-off protocol (MI355X_MICROARCH.md, "Valid forms"): payload with agent-scope relaxed atomic stores (sc1 write-through), every
// wave s_waitcnt vmcnt(0), workgroup barrier, ONE lane adds to the cluster's monotonic counter and polls it (sc1 loads + s_sleep),
// workgroup barrier, payload read with agent-scope atomic loads.  The poll is BOUNDED: a workgroup that does not see its cluster
// arrive within ~2^22 polls raises the error word and leaves, so the grid always drains.  All 8 x ng workgroups are resident at once
// (<= 80 on 256 CUs); workgroups that start late only make the others wait.
#include "dn_ops.h"

namespace {

constexpr int B4W = 8;            // workgroups per model
constexpr int B4R = 16;           // rows (MFMA M)
constexpr int B4P = 1028;         // LDS row pitch of the slab images (floats): 16 rows x 16 B apart -> conflict-free ds_read_b128
constexpr int B4A2P = 132;

// arrive: my published stores are acknowledged, then one lane signals.  wait: one lane polls (bounded), everybody learns the outcome.
// (Counter form of the hand-off: the backward kernel's second hand-off per layer, whose payload is the gradient slab itself.)
__device__ __forceinline__ void b4_arrive(unsigned* counter, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's published stores are acknowledged
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool b4_wait(unsigned* counter, unsigned* err, unsigned target, int tid, int* s_ok) {
    if (tid == 0) {
        int spins = 0, good = 1;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) { good = 0; atomicExch(err, 1u); break; }
        }
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) good = 0;
        *s_ok = good;
    }
    __syncthreads();
    return *s_ok != 0;
}

// Granule form of a hand-off (round 4; cdna_hip_programming.md, Guideline 16, R2 "the data IS the flag"): every handed-off float travels
// as ONE naturally aligned 8-byte {tag, value} word written by one agent-scope (sc1, write-through) store; a consumer lane re-reads ITS
// granules with agent-scope loads until every tag equals the phase's tag.  No counter, no vmcnt drain, no barrier on the producer side and
// one memory round trip on the consumer side -- against store-ack -> barrier -> atomic add -> poll -> payload load of the counter form
// (measured with -DB4_TIMING, round 3 kernels: 6.0 + 1.8 us of the 14.3 us per layer sat in the two counter hand-offs).
// STATE: the granule buffers are zeroed before every launch (tags are 1.. within a step, never 0): the forward driver's per-step
// zero-fill covers them.  A buffer is rewritten with the next phase's tag only after every consumer has finished reading the previous
// one (a workgroup publishes phase k + 2 into a buffer only after it has consumed phase k + 1 from ALL workgroups, which each of them
// publishes only after consuming phase k).
typedef unsigned long long b4_u64;
__device__ __forceinline__ void g_store(b4_u64* g, unsigned tag, float v) {
    __hip_atomic_store(g, ((b4_u64)tag << 32) | (b4_u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Every lane of the calling WAVE sweeps its N granules g[k * S] until all of the wave's tags match.  Bounded: ~0.3 s of wall clock
// (100 MHz counter) or another workgroup's raised error word end the sweep with `false` (and raise the word), so the grid always drains.
template <int N>
__device__ __forceinline__ bool g_sweep(const b4_u64* g, int S, unsigned tag, float (&v)[N], unsigned* err) {
    const unsigned long long t0 = wall_clock64();
    for (unsigned spins = 1;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const b4_u64 x = __hip_atomic_load(g + (size_t)k * S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[k] = __uint_as_float((unsigned)x);
            ok = ok && (unsigned)(x >> 32) == tag;
        }
        if (__all(ok)) return true;
        if ((spins & 31u) == 0u) {
            const bool late = wall_clock64() - t0 > 30000000ull;
            if (late || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                if (late) atomicExch(err, 1u);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

__global__ __launch_bounds__(256) void b4_fwd_kernel(const Grp<B4FwdP> grp) {
    const B4FwdP& p = grp.p[blockIdx.z];
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, k4 = lane >> 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    B4Layer* tabs = (B4Layer*)smem;            // [16] the block's layer table (pointer reads from LDS, not through a dependent global load)
    float* xs = smem + 512;                    // [16][B4P] raw slab (rows >= M stay zero)
    float* mu = xs + B4R * B4P;                // [1024] batch mean of every slab channel (train)
    float* rs = mu + 1024;                     // [1024] batch rstd
    float* mn1 = rs + 1024;                    // [1024] norm1 of the current layer: mean | gamma * rstd | beta
    float* sc1 = mn1 + 1024;
    float* be1 = sc1 + 1024;
    float* a2s = be1 + 1024;                   // [16][B4A2P] gathered relu(bn2(y1))
    float* red = a2s + B4R * B4A2P;            // [4][256] cross-wave sums, then scratch
    double* dred = (double*)(red + 1024);      // [512] column-statistic partials of the four waves
    int* nbt = (int*)(dred + 512);             // [27][16] neighbour row of (tap, row), -1 = zero padding
    int* live = nbt + 27 * 16;                 // [0] = number of live taps, [1..27] = their indices, [31] = "a sweep timed out"
    int& s_fail = live[31];
    const int M = p.M, C0 = p.C0, ld = p.ld;
    const float inv_m = 1.0f / (float)M;
    const int row = tid >> 4, col = tid & 15;
    const int n0 = 16 * w;                      // this workgroup's conv1 output channels

    // ---- set-up: layer table, slab columns [0, C0) and their statistics, neighbour table, live taps ------------------------
    static_assert(sizeof(B4Layer) == 120, "layer table entry");
    for (int i = tid; i < p.nlayers * 30; i += 256) ((unsigned*)tabs)[i] = ((const unsigned*)p.tab)[i];
    if (tid == 0) s_fail = 0;
    {   // every load of a batch of 8 float4 per thread is issued before the first LDS store (one memory round trip per batch)
        const int n4 = C0 >> 2, tot4 = B4R * n4;
        for (int base = 0; base < tot4; base += 2048) {
            float4 r[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i4 = base + tid + 256 * j, m = i4 / n4, k = (i4 - m * n4) << 2;
                r[j] = (i4 < tot4 && m < M) ? *(const float4*)(p.slab + (size_t)m * ld + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i4 = base + tid + 256 * j, m = i4 / n4, k = (i4 - m * n4) << 2;
                if (i4 < tot4) *(float4*)(xs + m * B4P + k) = r[j];
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
            const double m_ = sv[j] * (double)inv_m;
            double v = qv[j] * (double)inv_m - m_ * m_;
            v = v > 0.0 ? v : 0.0;
            if (k < C0) { mu[k] = (float)m_; rs[k] = 1.0f / sqrtf((float)v + p.eps); }
        }
    }
    for (int idx = tid; idx < 27 * B4R; idx += 256) {
        const int tap = idx >> 4, m = idx & 15;
        int nb = -1;
        if (m < M) {
            int d, h, x;
            unpack_dhw(p.coords[m], d, h, x);
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
            for (int m = 0; m < M; ++m) any = any || nbt[tap * 16 + m] >= 0;
            if (any) live[1 + n++] = tap;
        }
        live[0] = n;
    }
    __syncthreads();
    const int nlive = live[0];

    // Weights never depend on activations: a layer's conv1 slice (<= 16 float4 per lane), its conv2 tap slices (16 float4 per lane) and
    // its BatchNorm parameters are requested right after the PREVIOUS layer's second hand-off -- their latency hides under that layer's
    // statistics and this layer's norm1 set-up, and no load of theirs is in flight while a wave sweeps granules (a wave's loads return
    // in order: a sweep issued behind 64 KB of weight loads would wait for all of them).
    const int nt = wave & 1, half = wave >> 1;
    float4 wreg[16];
    float4 treg[4][4];
    int mytap[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int li = w + B4W * i; mytap[i] = li < nlive ? live[1 + li] : -1; }
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
    // workgroup's 16 channels (thread's column): gamma | beta [| running mean | running var]
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
#define B4_T(i) do { if (tid == 0) { const unsigned long long n_ = wall_clock64(); tacc[i] += n_ - tl; tl = n_; } } while (0)
#else
#define B4_T(i)
#endif
    for (int l = 0; l < p.nlayers; ++l) {
        const int C = C0 + 32 * l, nT = C >> 4;                 // K super-steps of 16 channels
        const B4Layer& L = tabs[l];
        B4_T(0);
        // a. norm1 constants of this layer for the C input channels (train: the channels' batch statistics, cached since they were
        //    produced; eval: this layer's running statistics); the transform itself rides in the MFMA loop's operand reads
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
        B4_T(1);
        // b. conv1: 16 rows x 16 channels, K split over the waves; A = relu(bn1(x)) built from the LDS slab on the fly
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float rowz = r16 < M ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int t = wave + 4 * i;
            if (t < nT) {
                const int k = 16 * t + 4 * k4;
                const float4 x = *(const float4*)(xs + r16 * B4P + k), m4 = *(const float4*)(mn1 + k), s4 = *(const float4*)(sc1 + k),
                             b4 = *(const float4*)(be1 + k);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rowz * fmaxf(bn_apply(x.x, m4.x, s4.x, b4.x), 0.f), wreg[i].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rowz * fmaxf(bn_apply(x.y, m4.y, s4.y, b4.y), 0.f), wreg[i].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rowz * fmaxf(bn_apply(x.z, m4.z, s4.z, b4.z), 0.f), wreg[i].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rowz * fmaxf(bn_apply(x.w, m4.w, s4.w, b4.w), 0.f), wreg[i].w, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave * 256 + (4 * k4 + r) * 16 + r16] = acc[r];
        __syncthreads();
        const float y = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid];       // y1[row][n0 + col]
        // c. BatchNorm2 statistics of the 16 channels (all rows are here): rows 4 wave .. 4 wave + 3 by lane shuffles, the four waves
        //    through LDS; every thread then holds its column's complete sums.  y1 + statistics saved for the backward
        float m_, r_;
        if (p.train) {
            double s = row < M ? (double)y : 0.0, q = s * s;
            s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
            s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
            if (lane < 16) { dred[wave * 32 + lane] = s; dred[128 + wave * 32 + lane] = q; }
            __syncthreads();
            s = ((dred[col] + dred[32 + col]) + dred[64 + col]) + dred[96 + col];
            q = ((dred[128 + col] + dred[160 + col]) + dred[192 + col]) + dred[224 + col];
            const double mm = s * (double)inv_m;
            double v = q * (double)inv_m - mm * mm;
            v = v > 0.0 ? v : 0.0;
            m_ = (float)mm; r_ = 1.0f / sqrtf((float)v + p.eps);
            if (tid < 16) { L.st_y1[n0 + tid] = s; L.st_y1[128 + n0 + tid] = q; }
            if (row < M) L.y1[(size_t)row * 128 + n0 + col] = y;
        } else {
            m_ = m2v; r_ = 1.0f / sqrtf(v2v + p.eps);
        }
        const unsigned tagA = 2u * (unsigned)l + 1u, tagB = tagA + 1u;
        g_store(p.xa + w * 256 + tid, tagA, row < M ? fmaxf(bn_apply(y, m_, g2v * r_, b2v), 0.f) : 0.f);
        // ---- hand-off A: gather the 16 x 128 relu(bn2(y1)) ------------------------------------------------------------------
        B4_T(2);
        {
            float v[8];
            const bool ok = g_sweep<8>(p.xa + tid, 256, tagA, v, p.err);
#pragma unroll
            for (int j = 0; j < 8; ++j) a2s[row * B4A2P + 16 * j + col] = v[j];
            if (!ok) s_fail = 1;
        }
        B4_T(3);
        __syncthreads();
        if (s_fail) return;
        // e. conv2: this workgroup's live taps; wave = (output-channel tile nt, input-channel half)
        f32x4 zc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (mytap[i] >= 0) {
                const int nb = nbt[mytap[i] * 16 + r16];
                const float* ar = a2s + (nb >= 0 ? nb : 0) * B4A2P + 64 * half + 4 * k4;
                const float z = nb >= 0 ? 1.f : 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 av = *(const float4*)(ar + 16 * t);
                    zc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x * z, treg[i][t].x, zc, 0, 0, 0);
                    zc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y * z, treg[i][t].y, zc, 0, 0, 0);
                    zc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z * z, treg[i][t].z, zc, 0, 0, 0);
                    zc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w * z, treg[i][t].w, zc, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave * 256 + (4 * k4 + r) * 16 + r16] = zc[r];      // [wave][row][co within the tile]
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j)              // partial z[row][co]: co tile nt = j, halves summed (waves j and j + 2)
            g_store(p.xb + w * 512 + row * 32 + 16 * j + col, tagB, red[j * 256 + tid] + red[(j + 2) * 256 + tid]);
        // ---- hand-off B: gather the 8 partial 16 x 32 outputs -----------------------------------------------------------------
        B4_T(4);
        float zv[2][8];
        {
            float v[16];
            const bool ok = g_sweep<16>(p.xb + tid, 256, tagB, v, p.err);      // granule (q, e = tid + 256 j) at q * 512 + e: k = 2 q + j
#pragma unroll
            for (int k = 0; k < 16; ++k) zv[k & 1][k >> 1] = v[k];
            if (!ok) s_fail = 1;
        }
        B4_T(5);
        if (l + 1 < p.nlayers) { load_w1(l + 1); load_c1(l + 1); load_taps(l + 1); }      // next layer's weights: ~2 us / ~6 us ahead of their use
        {
            double sq[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int e = tid + 256 * j, zr = e >> 5, co = e & 31;
                const float (&v)[8] = zv[j];
                float z = v[0];
#pragma unroll
                for (int q = 1; q < 8; ++q) z += v[q];                    // fixed order: deterministic, identical in every workgroup
                if (zr >= M) z = 0.f;
                xs[zr * B4P + C + co] = z;
                if (w == 0 && zr < M) p.slab[(size_t)zr * ld + C + co] = z;
                // column statistics of the 32 new channels: thread e covers (row e >> 5, channel e & 31); a wave = 2 rows x 32 channels
                double s = (double)z, q2 = s * s;
                s += __shfl_xor(s, 32, 64); q2 += __shfl_xor(q2, 32, 64);
                sq[j][0] = s; sq[j][1] = q2;
            }
            if (p.train) {
                // rows {2 wave, 2 wave + 1} (j = 0) and {8 + 2 wave, 9 + 2 wave} (j = 1): 8 partials per channel
                if (lane < 32) {
                    dred[wave * 32 + lane] = sq[0][0]; dred[128 + wave * 32 + lane] = sq[0][1];
                    dred[256 + wave * 32 + lane] = sq[1][0]; dred[384 + wave * 32 + lane] = sq[1][1];
                }
            }
        }
        __syncthreads();
        if (s_fail) return;
        if (tid < 32 && p.train) {
            double s = 0, q = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) { s += dred[u * 32 + tid]; q += dred[128 + u * 32 + tid]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { s += dred[256 + u * 32 + tid]; q += dred[384 + u * 32 + tid]; }
            const double mm = s * (double)inv_m;
            double v = q * (double)inv_m - mm * mm;
            v = v > 0.0 ? v : 0.0;
            mu[C + tid] = (float)mm; rs[C + tid] = 1.0f / sqrtf((float)v + p.eps);
            if (w == 0) { p.st_slab[C + tid] = s; p.st_slab[ld + C + tid] = q; }
        }
        __syncthreads();          // mu / rs of the new channels feed the next layer's norm1 constants
        B4_T(6);
    }
#ifdef B4_TIMING
    if (tid == 0) for (int i = 0; i < 7; ++i) p.err[8 + 8 * w + i] = (unsigned)tacc[i];      // 100 MHz ticks summed over the layers, per workgroup
#endif
}


// ---------------------------------------------------------------------------------------------------------------------------
// Backward data path of the block (the chain dslab -> relu2/norm2/conv2 -> relu1/norm1/conv1 -> dslab of layers 15 .. 0) as one launch;
// the weight gradients stay with the batched launches the network driver issues at the end of the block (they read what this kernel
// leaves per layer: the masked gradient at norm2's output `dmid`, its BatchNorm-backward sums, the final dz columns of dslab).
// Same cluster of 8 workgroups and the same hand-off protocol as the forward; per layer two hand-offs:
//   A. conv2 backward-data + norm2 backward: workgroup w owns the 16 mid channels [16w, 16w + 16): d(a2)[:, own] = sum over the live
//      taps of dz[row - off(tap)][32] x W2[own][tap][32] (taps dealt over the 4 waves), ReLU mask and BatchNorm-backward sums from the
//      saved y1 (all rows are here), writes dmid + sums, PUBLISHES dy1[:, own] = gamma*rstd*(g - s1/M - xhat*s2/M); all gather 16 x 128.
//   B. conv1 backward-data + norm1 backward: the layer's C input channels are dealt evenly (C/8 columns per workgroup); d(a1)[:, own] =
//      dy1[16][128] x W1[128][own] with one float4 of W1 (4 consecutive columns) feeding 4 MFMAs whose 16 output columns are the
//      interleaved sets {4j + e}; ReLU mask from the saved slab, column sums, norm1's backward applied in place on dslab[:, own]
//      (agent-scope stores: the next layer's dz columns and column split belong to other workgroups), dgamma / dbeta accumulated.
// Arithmetic (mask tests, sums in fp64, the dy / dx formulas) is the per-layer kernels' (Conv3BwdDataOp, DyConsts, Conv1BwdDataOp's fused
// epilogue), so the two paths agree to fp32 summation order.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int B4DZP = 36, B4DYP = 132;

__global__ __launch_bounds__(256) void b4_bwd_kernel(const Grp<B4BwdP> grp) {
    const B4BwdP& p = grp.p[blockIdx.z];
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, k4 = lane >> 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* mu = smem;                          // [1024] batch mean of every slab channel
    float* rs = mu + 1024;                     // [1024] batch rstd
    float* dzs = rs + 1024;                    // [16][B4DZP] dz of the layer (rows >= M zero)
    float* dys = dzs + B4R * B4DZP;            // [16][B4DYP] gathered dy1
    float* da = dys + B4R * B4DYP;             // [2][16][B4DYP] d(a1) partials of the two halves of the reduction
    float* red = da + 2 * B4R * B4DYP;         // [4][256] cross-wave sums
    double* dred = (double*)(red + 1024);      // [1024] column-sum partials
    float* c2 = (float*)(dred + 1024);         // [64]: norm2 mean | rstd | s1/M | s2/M of this workgroup's 16 channels
    int* nbm = (int*)(c2 + 64);                // [27][16] row whose output tap `tap` reads this row (row - off(tap)), -1 = outside the grid
    int* live = nbm + 27 * 16;                 // [0] = number of live taps, [1..] = their indices
    __shared__ int s_ok;
    const int M = p.M, C0 = p.C0, ld = p.ld;
    const float inv_m = 1.0f / (float)M;
    const double inv_md = (double)inv_m;

    for (int k = tid; k < ld; k += 256) {
        const double s = p.st_slab[k], q = p.st_slab[ld + k];
        const double m_ = s * inv_md;
        double v = q * inv_md - m_ * m_;
        v = v > 0.0 ? v : 0.0;
        mu[k] = (float)m_; rs[k] = 1.0f / sqrtf((float)v + p.eps);
    }
    for (int idx = tid; idx < 27 * B4R; idx += 256) {
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
    const int row = tid >> 4, col = tid & 15;                 // phase A: element (row, mid channel 16w + col)
    const int bcol = tid & 127, brg = tid >> 7;                // phase B: column bcol of this workgroup's slice, rows 8 brg .. 8 brg + 7
    const int T = wave & 1, kh2 = wave >> 1;                   // phase B MFMA: 64-column half, half of the 128-long reduction
    unsigned phase = 0;
    unsigned* counter = p.counter;

    // registers loaded one phase ahead of their use (weights and saved activations never depend on the chain)
    float4 treg[7][2];
    float yv, g2v, b2v; double sy, qy;
    auto load_A = [&](int l) __attribute__((always_inline)) {
        const B4Layer& L = p.tab[l];
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
    float xv[8], ov[8], g1v, b1v;
    auto load_B = [&](int l) __attribute__((always_inline)) {
        const B4Layer& L = p.tab[l];
        const int C = C0 + 32 * l, nc = C >> 3, c0 = w * nc;
        const bool okc = 64 * T + 4 * r16 < nc;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int n = 4 * (16 * kh2 + i) + k4;
            wreg[i] = okc ? *(const float4*)(L.w1 + (size_t)n * C + c0 + 64 * T + 4 * r16) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int c = c0 + (bcol < nc ? bcol : 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = 8 * brg + i;
            xv[i] = m < M ? p.slab[(size_t)m * ld + c] : 0.f;
            ov[i] = m < M ? pload(p.dslab + (size_t)m * ld + c) : 0.f;     // the read half of norm1's in-place update: final since the last hand-off
        }
        g1v = L.g1[c]; b1v = L.b1[c];
    };
    load_A(p.nlayers - 1);

    for (int l = p.nlayers - 1; l >= 0; --l) {
        const int C = C0 + 32 * l, nc = C >> 3, c0 = w * nc;
        const B4Layer L = p.tab[l];
        // ---- dz of this layer: final since the previous layer's phase B (hand-off at the bottom of the loop) ----------------------
        {
            float v[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) { const int e = tid + 256 * j, m = e >> 5; v[j] = m < M ? pload(p.dslab + (size_t)m * ld + C + (e & 31)) : 0.f; }
            asm volatile("" ::: "memory");            // the prefetch goes out BEHIND the dz loads (vmcnt retires in issue order)
            load_B(l);                                // conv1 weights / slab values / norm1 parameters of this layer: used after hand-off A
#pragma unroll
            for (int j = 0; j < 2; ++j) { const int e = tid + 256 * j; dzs[(e >> 5) * B4DZP + (e & 31)] = v[j]; }
        }
        __syncthreads();
        // ---- A1. conv2 backward-data for the 16 own mid channels, this wave's taps --------------------------------------------------
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            if (mytap[j] >= 0) {
                const int nb = nbm[mytap[j] * 16 + r16];
                const float* ar = dzs + (nb >= 0 ? nb : 0) * B4DZP + 4 * k4;
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
        dred[tid] = (double)g; dred[256 + tid] = (double)g * xh2;
        __syncthreads();
        if (tid < 16) {
            double a = 0, b = 0;
            for (int m = 0; m < B4R; ++m) { a += dred[m * 16 + tid]; b += dred[256 + m * 16 + tid]; }
            L.bb_y1[16 * w + tid] = a; L.bb_y1[128 + 16 * w + tid] = b;
            c2[32 + tid] = (float)(a * inv_md); c2[48 + tid] = (float)(b * inv_md);
        }
        __syncthreads();
        {
            const float dy = row < M ? (g2v * r2) * (g - c2[32 + col] - (yv - mu2) * r2 * c2[48 + col]) : 0.f;
            pstore(p.xa + w * 256 + tid, dy);
        }
        // ---- hand-off A ---------------------------------------------------------------------------------------------------------------
        b4_arrive(counter, tid);
        if (!b4_wait(counter, p.err, (++phase) * B4W, tid, &s_ok)) return;
        {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = pload(p.xa + j * 256 + tid);
            asm volatile("" ::: "memory");
            if (l > 0) load_A(l - 1);                 // next layer's conv2 weights / y1 / norm2 parameters: used after hand-off B
#pragma unroll
            for (int j = 0; j < 8; ++j) dys[row * B4DYP + 16 * j + col] = v[j];
        }
        __syncthreads();
        // ---- B1. conv1 backward-data: d(a1)[16][own columns], 4 interleaved 16-column sets per wave, half the reduction per wave ------
        f32x4 ac4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) ac4[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float a = dys[r16 * B4DYP + 4 * (16 * kh2 + i) + k4];
            ac4[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[i].x, ac4[0], 0, 0, 0);
            ac4[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[i].y, ac4[1], 0, 0, 0);
            ac4[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[i].z, ac4[2], 0, 0, 0);
            ac4[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[i].w, ac4[3], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *(float4*)(da + kh2 * (B4R * B4DYP) + (4 * k4 + r) * B4DYP + 64 * T + 4 * r16) = make_float4(ac4[0][r], ac4[1][r], ac4[2][r], ac4[3][r]);
        __syncthreads();
        // ---- B2. relu1 mask, column sums, norm1 backward applied in place ---------------------------------------------------------------
        {
            const bool okc = bcol < nc;
            const int c = c0 + (okc ? bcol : 0);
            const float mu1 = mu[c], r1 = rs[c];
            float gv[8];
            double s1 = 0, s2 = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = 8 * brg + i;
                const float xh = (xv[i] - mu1) * r1;
                const float gg = (okc && m < M && fmaf(g1v, xh, b1v) > 0.f) ? da[m * B4DYP + bcol] + da[B4R * B4DYP + m * B4DYP + bcol] : 0.f;
                gv[i] = gg;
                s1 += gg; s2 += (double)gg * xh;
            }
            dred[(brg * 2) * 128 + bcol] = s1; dred[(brg * 2 + 1) * 128 + bcol] = s2;
            __syncthreads();
            if (okc) {
                const double a = dred[bcol] + dred[256 + bcol], b = dred[128 + bcol] + dred[384 + bcol];
                const float gr = g1v * r1, m1 = (float)(a * inv_md), m2 = r1 * (float)(b * inv_md);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int m = 8 * brg + i;
                    if (m < M) pstore(p.dslab + (size_t)m * ld + c, ov[i] + gr * (gv[i] - m1 - (xv[i] - mu1) * m2));
                }
                if (brg == 0) { p.dg1[l][c] += (float)b; p.db1[l][c] += (float)a; }
            }
        }
        // ---- hand-off B: this layer's dslab updates are visible to the cluster before the next layer reads its dz / its columns --------
        if (l > 0) {
            b4_arrive(counter, tid);
            if (!b4_wait(counter, p.err, (++phase) * B4W, tid, &s_ok)) return;
        }
    }
}

}  // namespace

extern "C" int mms_b4_fwd_group(const B4FwdP* pp, int ng, hipStream_t s) {
    Grp<B4FwdP> a;
    if (!grp_fill(a, pp, ng, 1)) return MMS_ERR_ARG;
    for (int g = 0; g < ng; ++g) {
        const B4FwdP& p = pp[g];
        if (p.M < 1 || p.M > 16 || p.ld != 1024 || p.C0 % 32 != 0 || p.C0 + 32 * p.nlayers > p.ld || !p.tab || !p.slab || !p.xa || !p.xb ||
            !p.err || !p.coords || (p.train && !p.st_slab) || p.M != pp->M || p.nlayers != pp->nlayers || p.nlayers < 1 || p.nlayers > 16 ||
            (((uintptr_t)p.slab | (uintptr_t)p.xa | (uintptr_t)p.xb) & 15)) return MMS_ERR_ARG;
    }
    constexpr int smem = 2048 + (B4R * B4P + 5 * 1024 + B4R * B4A2P + 1024) * 4 + 512 * 8 + (27 * 16 + 32) * 4;
    static std::once_flag attr_once;
    std::call_once(attr_once, [&] { hipFuncSetAttribute((const void*)b4_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem); });
    MMS_LAUNCH(b4_fwd_kernel, dim3(B4W, 1, ng), dim3(256), smem, s, a);
    return mms_check_launch();
}

extern "C" int mms_b4_bwd_group(const B4BwdP* pp, int ng, hipStream_t s) {
    Grp<B4BwdP> a;
    static_assert(sizeof(Grp<B4BwdP>) <= 4096, "kernel argument block");
    if (!grp_fill(a, pp, ng, 1)) return MMS_ERR_ARG;
    for (int g = 0; g < ng; ++g) {
        const B4BwdP& p = pp[g];
        if (p.M < 1 || p.M > 16 || p.ld != 1024 || p.C0 % 32 != 0 || p.nlayers < 1 || p.nlayers > 16 || p.C0 + 32 * p.nlayers > p.ld || !p.tab || !p.slab ||
            !p.dslab || !p.st_slab || !p.xa || !p.counter || !p.err || !p.coords || p.M != pp->M || p.nlayers != pp->nlayers) return MMS_ERR_ARG;
        for (int l = 0; l < p.nlayers; ++l) if (!p.dg1[l] || !p.db1[l]) return MMS_ERR_ARG;
    }
    constexpr int smem = (2 * 1024 + B4R * B4DZP + 3 * B4R * B4DYP + 1024) * 4 + 1024 * 8 + 64 * 4 + (27 * 16 + 32) * 4;
    MMS_LAUNCH(b4_bwd_kernel, dim3(B4W, 1, ng), dim3(256), smem, s, a);
    return mms_check_launch();
}
