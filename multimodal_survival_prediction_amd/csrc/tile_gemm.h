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
// LDS images: As[2][TM][TK+4], Bs[2][TN][TK+4] floats (row pitch = TK+4 dwords keeps 16-B alignment for
// ds_read_b128 and makes the 16-lane read groups conflict-free: pitch mod 64 dwords = 4*odd or 36).
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
//   __device__ float4 a_k4 / a_r4 (const Params&, int m, int k);  float a_k1(...);   zero-fill out of range
//   __device__ float4 b_k4 / b_r4 (const Params&, int n, int k);  float b_k1(...);
//   __device__ void epilogue(const Params&, int m0, int n0, int z, const float* Cs /*[TM][TN+1]*/, int tid);
#pragma once
#include "common.h"

enum { LD_K4 = 0, LD_R4 = 1, LD_K1 = 2 };

template <class Op>
struct TileGemmCfg {
    static constexpr int TM = 32 * Op::WM, TN = 32 * Op::WN, TK = 32 * Op::WK, PITCH = TK + 4;
    static constexpr size_t smem_bytes() {
        size_t tiles = (size_t)2 * (TM + TN) * PITCH;
        size_t cs = (size_t)TM * (TN + 1);
        return ((tiles > cs ? tiles : cs) + Op::EXTRA) * sizeof(float);
    }
};

template <class Op>
__global__ __launch_bounds__(256) void tile_gemm_kernel(const typename Op::Params p) {
    constexpr int WM = Op::WM, WN = Op::WN, WK = Op::WK;
    static_assert(WM * WN * WK == 4, "4 waves per workgroup");
    constexpr int TM = 32 * WM, TN = 32 * WN, TK = 32 * WK, PITCH = TK + 4;
    constexpr size_t TILE_FLOATS = (size_t)2 * (TM + TN) * PITCH;
    constexpr size_t CS_FLOATS = (size_t)TM * (TN + 1);
    constexpr size_t MAIN_FLOATS = TILE_FLOATS > CS_FLOATS ? TILE_FLOATS : CS_FLOATS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * TM * PITCH;
    float* extra = smem + MAIN_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN, z = blockIdx.z;

    Op op;
    op.setup(p, m0, n0, z, extra, tid);
    __syncthreads();
    int kb, ke;
    op.krange(p, z, kb, ke);

    constexpr int NA = (Op::AMODE == LD_K1) ? TM * TK / 256 : TM * TK / 4 / 256;
    constexpr int NB = (Op::BMODE == LD_K1) ? TN * TK / 256 : TN * TK / 4 / 256;
    float4 ra[Op::AMODE == LD_K1 ? 1 : NA];
    float4 rb[Op::BMODE == LD_K1 ? 1 : NB];
    float sa[Op::AMODE == LD_K1 ? NA : 1];
    float sb[Op::BMODE == LD_K1 ? NB : 1];

    auto gload = [&](int k0) {
        if constexpr (Op::AMODE == LD_K4) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                int idx = tid + i * 256, row = idx / (TK / 4), kq = (idx % (TK / 4)) * 4;
                ra[i] = op.a_k4(p, m0 + row, k0 + kq);
            }
        } else if constexpr (Op::AMODE == LD_R4) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                int idx = tid + i * 256, r4 = (idx % (TM / 4)) * 4, k = idx / (TM / 4);
                ra[i] = op.a_r4(p, m0 + r4, k0 + k);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                int idx = tid + i * 256, row = idx / TK, k = idx % TK;
                sa[i] = op.a_k1(p, m0 + row, k0 + k);
            }
        }
        if constexpr (Op::BMODE == LD_K4) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                int idx = tid + i * 256, row = idx / (TK / 4), kq = (idx % (TK / 4)) * 4;
                rb[i] = op.b_k4(p, n0 + row, k0 + kq);
            }
        } else if constexpr (Op::BMODE == LD_R4) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                int idx = tid + i * 256, r4 = (idx % (TN / 4)) * 4, k = idx / (TN / 4);
                rb[i] = op.b_r4(p, n0 + r4, k0 + k);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                int idx = tid + i * 256, row = idx / TK, k = idx % TK;
                sb[i] = op.b_k1(p, n0 + row, k0 + k);
            }
        }
    };
    auto sstore = [&](int buf) {
        float* a = As + buf * TM * PITCH;
        float* b = Bs + buf * TN * PITCH;
        if constexpr (Op::AMODE == LD_K4) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                int idx = tid + i * 256, row = idx / (TK / 4), kq = (idx % (TK / 4)) * 4;
                *(float4*)&a[row * PITCH + kq] = ra[i];
            }
        } else if constexpr (Op::AMODE == LD_R4) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                int idx = tid + i * 256, r4 = (idx % (TM / 4)) * 4, k = idx / (TM / 4);
                a[(r4 + 0) * PITCH + k] = ra[i].x; a[(r4 + 1) * PITCH + k] = ra[i].y;
                a[(r4 + 2) * PITCH + k] = ra[i].z; a[(r4 + 3) * PITCH + k] = ra[i].w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                int idx = tid + i * 256, row = idx / TK, k = idx % TK;
                a[row * PITCH + k] = sa[i];
            }
        }
        if constexpr (Op::BMODE == LD_K4) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                int idx = tid + i * 256, row = idx / (TK / 4), kq = (idx % (TK / 4)) * 4;
                *(float4*)&b[row * PITCH + kq] = rb[i];
            }
        } else if constexpr (Op::BMODE == LD_R4) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                int idx = tid + i * 256, r4 = (idx % (TN / 4)) * 4, k = idx / (TN / 4);
                b[(r4 + 0) * PITCH + k] = rb[i].x; b[(r4 + 1) * PITCH + k] = rb[i].y;
                b[(r4 + 2) * PITCH + k] = rb[i].z; b[(r4 + 3) * PITCH + k] = rb[i].w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                int idx = tid + i * 256, row = idx / TK, k = idx % TK;
                b[row * PITCH + k] = sb[i];
            }
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

    if (kb < ke) {
        gload(kb);
        sstore(0);
        __syncthreads();
        int buf = 0;
        for (int k0 = kb; k0 < ke; k0 += TK) {
            const bool more = (k0 + TK) < ke;
            if (more) gload(k0 + TK);
            const float* ap = As + (buf * TM + wm * 32 + (lane & 31)) * PITCH + wk * 32 + 4 * (lane >> 5);
            const float* bp = Bs + (buf * TN + wn * 32 + (lane & 31)) * PITCH + wk * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int kk = 0; kk < 32; kk += 8) {
                const float4 a = *(const float4*)(ap + kk);
                const float4 b = *(const float4*)(bp + kk);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            }
            if (more) sstore(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }

    // ---- epilogue: sum the WK partial accumulators into Cs[TM][TN+1] (aliases the tile buffers) ----
    float* Cs = smem;
    const int crow = wm * 32 + 4 * (lane >> 5), ccol = wn * 32 + (lane & 31);
#pragma unroll
    for (int w = 0; w < WK; ++w) {
        if (wk == w) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* c = &Cs[(crow + (r & 3) + 8 * (r >> 2)) * (TN + 1) + ccol];
                if (w == 0) *c = acc[r]; else *c += acc[r];
            }
        }
        __syncthreads();
    }
    op.epilogue(p, m0, n0, z, Cs, tid);
}

template <class Op>
static inline int launch_tile_gemm(const typename Op::Params& p, dim3 grid, hipStream_t s) {
    constexpr size_t smem = TileGemmCfg<Op>::smem_bytes();
    static bool attr_set = false;
    if (smem > 64 * 1024 && !attr_set) {
        hipFuncSetAttribute((const void*)tile_gemm_kernel<Op>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    if (grid.x == 0 || grid.y == 0 || grid.z == 0) return MMS_OK;
    MMS_LAUNCH(tile_gemm_kernel<Op>, grid, dim3(256), smem, s, p);
    return mms_check_launch();
}
