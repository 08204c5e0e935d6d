// Generic LDS-tiled fp32 GEMM core on v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain).
//
//   C[m][n] = sum_k A(m,k) * B(n,k)          (both operands presented "k-contiguous" in LDS)
//
// One 256-thread workgroup (4 wave64) owns a (32*WM) x (32*WN) output tile; WK waves split each
// K-step (TK = 32*WK) between them and are summed through LDS in the epilogue.  Every wave owns exactly
// one 32x32 accumulator (16 VGPRs/lane).  Operands are produced by an Op's loader methods, which may
// gather (conv taps, pooling windows) and transform (BatchNorm+ReLU prologues, BN-backward) on the way
// from HBM/L2 to LDS -- so the fused ops of the hot path are all instances of this one core.
//
// LDS images (K4/K1 loaders): As[2][TM][TK+4], Bs[2][TN][TK+4] floats (row pitch = TK+4 dwords keeps 16-B
// alignment for ds_read_b128 and makes the 16-lane read groups conflict-free); R4 loaders: k-major images, see
// TileGemmCfg.
// MFMA operand map (cdna_hip_programming.md section 3): lane l holds A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// one ds_read_b128 per operand feeds 4 MFMAs: element e of lane (i,h) is k = kk + 4h + e for both
// operands, so MFMA e sums k in {kk+e, kk+4+e}.  C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
//
// Op interface (all static-polymorphic):
//   typedef Params;  static constexpr int WM, WN, WK, AMODE, BMODE, EXTRA (floats of extra LDS);
//   AMODE/BMODE: 0 = K4 (float4 of 4 consecutive k for one row), 1 = R4 (4 consecutive rows at one k),
//                2 = K1 (scalar, arbitrary alignment)
//   __device__ void setup(const Params&, int m0, int n0, int z, float* extra, int tid);   (barrier follows)
//   __device__ void krange(const Params&, int z, int& kb, int& ke);
//   typedef ARaw, BRaw: what one loader call leaves in registers (raw global data, no dependent math);
//   __device__ void step(const Params&, int k0);   once per tile, wave-uniform k0: scalar per-tile state
//   __device__ ARaw a_ld(const Params&, int i, int m, int k, bool& ok);   issue the global load(s) only; i = the
//       thread's piece index (ops may keep per-piece state -- byte offsets, validity masks -- in registers)
//   __device__ float4|float a_tx(const Params&, int i, const ARaw&, int m, int k, bool ok);   transform -> LDS value
//   (same for b_ld / b_tx with n instead of m).  The core issues every *_ld of tile t+1 BEFORE the MFMAs of
//   tile t and runs *_tx after them, at LDS-store time, so HBM/L2 latency hides under the matrix phase.
//   __device__ void epilogue(const Params&, int m0, int n0, int z, const float* Cs /*[TM][TN+1]*/, int tid, bool active);
//   (active is always true; a wave-specialised 512-thread variant of this core -- producer waves loading, consumer waves
//   issuing MFMAs -- was measured on the conv3 forward and gave nothing over the register-prefetch pipeline, and was removed)
#pragma once
#include "common.h"

enum { LD_K4 = 0, LD_R4 = 1, LD_K1 = 2 };

// LDS image of one operand tile: K4/K1 loaders -> [row][TK+4] (read with ds_read_b128);
// R4 loaders (source contiguous along the row index) -> k-major [TK][rows+4]: written with conflict-free
// ds_write_b128 along the rows, read with one ds_read_b32 per MFMA operand (32 consecutive dwords per half-wave).
// Optional Op trait: static constexpr bool ONE_TILE = true -- the launcher guarantees a K range of at most one tile (ke - kb <= TK):
// no second LDS buffer (a third workgroup fits a CU, and the kernel leaves room for the other streams' workgroups), no prefetch set.
template <class Op, class = void> struct op_one_tile { static constexpr bool value = false; };
template <class Op> struct op_one_tile<Op, decltype((void)Op::ONE_TILE)> { static constexpr bool value = Op::ONE_TILE; };

// Optional Op trait: static constexpr bool SINGLE_BUF = true -- one LDS tile buffer for a multi-tile K range: the next tile waits in
// registers (the prefetch sets are unchanged) and is stored after a barrier behind the current tile's MFMAs.  One more barrier per K
// step for half the LDS: for the 128-deep tiles of the <1, 1, 4> shapes (80 KB double-buffered = 2 workgroups per CU and no room for
// another stream's workgroups beside them) the footprint, not the barrier, is what costs the step (profiles/r03_step_ablation.txt).
template <class Op, class = void> struct op_single_buf { static constexpr bool value = false; };
template <class Op> struct op_single_buf<Op, decltype((void)Op::SINGLE_BUF)> { static constexpr bool value = Op::SINGLE_BUF; };

template <class Op>
struct TileGemmCfg {
    static constexpr int TM = 32 * Op::WM, TN = 32 * Op::WN, TK = 32 * Op::WK;
    static constexpr int APITCH = Op::AMODE == 1 ? TM + 4 : TK + 4, BPITCH = Op::BMODE == 1 ? TN + 4 : TK + 4;
    static constexpr int ATILE = Op::AMODE == 1 ? TK * APITCH : TM * APITCH;
    static constexpr int BTILE = Op::BMODE == 1 ? TK * BPITCH : TN * BPITCH;
    static constexpr size_t main_floats() {
        size_t tiles = (size_t)((op_one_tile<Op>::value || op_single_buf<Op>::value) ? 1 : 2) * (ATILE + BTILE);
        size_t cs = (size_t)TM * (TN + 1);
        return tiles > cs ? tiles : cs;
    }
    static constexpr size_t smem_bytes() { return (main_floats() + Op::EXTRA) * sizeof(float); }
};

// Optional Op hook: static void zremap(int flat, int zdim, int nflat, int& gi, int& z) -- decodes blockIdx.z (flat over
// models x zdim) itself, e.g. to keep the workgroups that share operand rows on one XCD (workgroups are dealt to the 8
// XCDs round-robin by linear id).  Default: gi = flat / zdim, z = flat % zdim.
template <class Op, class = void> struct has_zremap { static constexpr bool value = false; };
template <class Op> struct has_zremap<Op, decltype((void)&Op::zremap)> { static constexpr bool value = true; };

#ifndef MMS_TGK_ATTR
#define MMS_TGK_ATTR            // register-budget experiments: -DMMS_TGK_ATTR='__attribute__((amdgpu_waves_per_eu(4, 4)))'
#endif
template <class Op>
__global__ __launch_bounds__(256) MMS_TGK_ATTR void tile_gemm_kernel(const Grp<typename Op::Params> grp) {
    int gi, z, bx = blockIdx.x;                            // model of the fold group, the op's own z index, the M tile
    if constexpr (has_zremap<Op>::value) Op::zremap((int)blockIdx.z, grp.zdim, (int)gridDim.z, gi, z);
    else if (grp.zdim == 1) { xcd_place(gi, bx); z = 0; }  // M-tiled launch: XCD-contiguous tile ranges, each model of a 2 / 4 / 8 group on its own XCDs
    else { gi = blockIdx.z / grp.zdim; z = blockIdx.z - gi * grp.zdim; }
    const typename Op::Params& p = grp.p[gi];
    constexpr int WM = Op::WM, WN = Op::WN, WK = Op::WK;
    static_assert(WM * WN * WK == 4, "4 waves per workgroup");
    typedef TileGemmCfg<Op> Cfg;
    constexpr int TM = Cfg::TM, TN = Cfg::TN, TK = Cfg::TK;
    constexpr int APITCH = Cfg::APITCH, BPITCH = Cfg::BPITCH, ATILE = Cfg::ATILE, BTILE = Cfg::BTILE;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    constexpr bool ONE = op_one_tile<Op>::value, SINGLE = op_single_buf<Op>::value && !ONE;
    float* Bs = smem + ((ONE || SINGLE) ? 1 : 2) * ATILE;
    float* extra = smem + Cfg::main_floats();

    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    constexpr bool consumer = true;
    const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
    // XCD-aware tile order (cdna_hip_programming.md T1): workgroups are dealt round-robin over the 8 XCDs, so give the
    // blocks that share an XCD (equal blockIdx.x % 8) a CONTIGUOUS range of M tiles -- neighbouring tiles share halo rows /
    // operand panels, and each XCD's L2 then holds one compact slice of the activations instead of a scatter of all of them.
    // Placement only affects speed, never results.  (zdim == 1: done by xcd_place above, with the models of a fold group apart.)
    if ((has_zremap<Op>::value || grp.zdim != 1) && (gridDim.x & 7) == 0) bx = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int m0 = bx * TM, n0 = blockIdx.y * TN;

    Op op;
    op.setup(p, m0, n0, z, extra, tid);
    __syncthreads();
    int kb, ke;
    op.krange(p, z, kb, ke);

    constexpr int NA = (Op::AMODE == LD_K1) ? TM * TK / 256 : TM * TK / 4 / 256;
    constexpr int NB = (Op::BMODE == LD_K1) ? TN * TK / 256 : TN * TK / 4 / 256;
    struct RegSet {                 // one tile's worth of raw operand data held by a loader thread
        typename Op::ARaw ra[NA];
        typename Op::BRaw rb[NB];
        unsigned oka, okb;
        int kcur;
    };
    RegSet R0, R1;

    // thread -> (row, k) of its i-th piece of the A / B tile
    auto a_pos = [&](int i, int& row, int& k) {
        const int idx = tid + i * 256;
        if constexpr (Op::AMODE == LD_K4) { row = idx / (TK / 4); k = (idx % (TK / 4)) * 4; }
        else if constexpr (Op::AMODE == LD_R4) { row = (idx % (TM / 4)) * 4; k = idx / (TM / 4); }
        else { row = idx / TK; k = idx % TK; }
    };
    auto b_pos = [&](int i, int& row, int& k) {
        const int idx = tid + i * 256;
        if constexpr (Op::BMODE == LD_K4) { row = idx / (TK / 4); k = (idx % (TK / 4)) * 4; }
        else if constexpr (Op::BMODE == LD_R4) { row = (idx % (TN / 4)) * 4; k = idx / (TN / 4); }
        else { row = idx / TK; k = idx % TK; }
    };
    auto gload = [&](RegSet& R, int k0) {
        R.kcur = k0; R.oka = 0; R.okb = 0;
#ifdef MMS_ABLATE_GLOAD
        return;
#endif
        op.step(p, k0);                      // wave-uniform per-tile state (tap offsets, scalar load offsets)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            int row, k; a_pos(i, row, k);
            bool ok;
#ifdef MMS_ABLATE_ALOAD
            ok = false; R.ra[i] = typename Op::ARaw{};
#else
            R.ra[i] = op.a_ld(p, i, m0 + row, k0 + k, ok);
#endif
            R.oka |= (ok ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int row, k; b_pos(i, row, k);
            bool ok;
#ifdef MMS_ABLATE_BLOAD
            ok = false; R.rb[i] = typename Op::BRaw{};
#else
            R.rb[i] = op.b_ld(p, i, n0 + row, k0 + k, ok);
#endif
            R.okb |= (ok ? 1u : 0u) << i;
        }
    };
    auto sstore = [&](const RegSet& R, int buf) {
#ifdef MMS_ABLATE_SSTORE
        return;
#endif
        float* a = As + buf * ATILE;
        float* b = Bs + buf * BTILE;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            int row, k; a_pos(i, row, k);
            const auto v = op.a_tx(p, i, R.ra[i], m0 + row, R.kcur + k, (R.oka >> i) & 1u);
            if constexpr (Op::AMODE == LD_K4) { *(float4*)&a[row * APITCH + k] = v; }
            else if constexpr (Op::AMODE == LD_R4) { *(float4*)&a[k * APITCH + row] = v; }
            else { a[row * APITCH + k] = v; }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int row, k; b_pos(i, row, k);
            const auto v = op.b_tx(p, i, R.rb[i], n0 + row, R.kcur + k, (R.okb >> i) & 1u);
            if constexpr (Op::BMODE == LD_K4) { *(float4*)&b[row * BPITCH + k] = v; }
            else if constexpr (Op::BMODE == LD_R4) { *(float4*)&b[k * BPITCH + row] = v; }
            else { b[row * BPITCH + k] = v; }
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

    auto mma = [&](int buf) {
#ifdef MMS_ABLATE_MMA
        return;
#endif
        const float* at = As + buf * ATILE;
        const float* bt = Bs + buf * BTILE;
        const int li = lane & 31, kh = wk * 32 + 4 * (lane >> 5);
        float4 a[4], b[4];      // all LDS reads of the K-step are issued before its 16 dependent MFMAs
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kk = 8 * q;
            if constexpr (Op::AMODE == LD_R4) {
                const float* r = at + (kh + kk) * APITCH + wm * 32 + li;
                a[q] = make_float4(r[0], r[APITCH], r[2 * APITCH], r[3 * APITCH]);
            } else {
                a[q] = *(const float4*)(at + (wm * 32 + li) * APITCH + kh + kk);
            }
            if constexpr (Op::BMODE == LD_R4) {
                const float* r = bt + (kh + kk) * BPITCH + wn * 32 + li;
                b[q] = make_float4(r[0], r[BPITCH], r[2 * BPITCH], r[3 * BPITCH]);
            } else {
                b[q] = *(const float4*)(bt + (wn * 32 + li) * BPITCH + kh + kk);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, b[q].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, b[q].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, b[q].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, b[q].w, acc, 0, 0, 0);
        }
    };

    if (kb < ke) {
        if constexpr (ONE) {
            gload(R0, kb);
            sstore(R0, 0);
            __syncthreads();
            mma(0);
            __syncthreads();
        } else if constexpr (SINGLE) {
            gload(R0, kb);
            if (kb + TK < ke) gload(R1, kb + TK);
            sstore(R0, 0);
            __syncthreads();
            for (int k0 = kb;; k0 += 2 * TK) {           // same tiles, same prefetch distance and accumulation order as the two-buffer loop
                if (k0 + 2 * TK < ke) gload(R0, k0 + 2 * TK);
                mma(0);
                if (k0 + TK >= ke) break;
                __syncthreads();
                sstore(R1, 0);
                __syncthreads();
                if (k0 + 3 * TK < ke) gload(R1, k0 + 3 * TK);
                mma(0);
                if (k0 + 2 * TK >= ke) break;
                __syncthreads();
                sstore(R0, 0);
                __syncthreads();
            }
            __syncthreads();
        } else {
            // two register sets: the global loads of tile t+2 are issued before the MFMAs of tile t and consumed (transform
            // + LDS store) one step later, so L2/MALL latency has a full step to hide (vmcnt retires in order)
            gload(R0, kb);
            if (kb + TK < ke) gload(R1, kb + TK);
            sstore(R0, 0);
            __syncthreads();
            int buf = 0;
            for (int k0 = kb; k0 + TK < ke; k0 += 2 * TK) {
                if (k0 + 2 * TK < ke) gload(R0, k0 + 2 * TK);
                mma(buf);
                sstore(R1, buf ^ 1);
                __syncthreads();
                buf ^= 1;
                if (k0 + 2 * TK >= ke) break;
                if (k0 + 3 * TK < ke) gload(R1, k0 + 3 * TK);
                mma(buf);
                sstore(R0, buf ^ 1);
                __syncthreads();
                buf ^= 1;
            }
            mma(buf);            // last tile: nothing left to prefetch
            __syncthreads();
        }
    }

    // ---- epilogue: sum the WK partial accumulators into Cs[TM][TN+1] (aliases the tile buffers) ----
    float* Cs = smem;
    const int crow = wm * 32 + 4 * (lane >> 5), ccol = wn * 32 + (lane & 31);
#pragma unroll
    for (int w = 0; w < WK; ++w) {
        if (wk == w && consumer) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* c = &Cs[(crow + (r & 3) + 8 * (r >> 2)) * (TN + 1) + ccol];
                if (w == 0) *c = acc[r]; else *c += acc[r];
            }
        }
        __syncthreads();
    }
    op.epilogue(p, m0, n0, z, Cs, tid, consumer);
}

template <class Op>
static inline int launch_tile_gemm(const typename Op::Params* pp, int ng, dim3 grid, hipStream_t s) {
    constexpr size_t smem = TileGemmCfg<Op>::smem_bytes();
    if (smem > 64 * 1024) {      // once per instantiation, safe under concurrent host threads (the ABI is documented re-entrant)
        static std::once_flag attr_once;
        std::call_once(attr_once, [] { hipFuncSetAttribute((const void*)tile_gemm_kernel<Op>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); });
    }
    if (grid.x == 0 || grid.y == 0 || grid.z == 0) return MMS_OK;
    Grp<typename Op::Params> a;
    if (!grp_fill(a, pp, ng, (int)grid.z)) return MMS_ERR_ARG;
    grid.z *= ng;
    MMS_LAUNCH(tile_gemm_kernel<Op>, grid, dim3(256), smem, s, a);
    return mms_check_launch();
}
