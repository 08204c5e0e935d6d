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
