/* mmsurv.h -- C ABI of libmmsurv_hip.so: the MI355X (gfx950) implementation of the training hot path of
 * baek0203/multimodal_survival_prediction (per-batch forward/backward of the multimodal survival networks +
 * Cox partial likelihood).
 *
 * The reference has no FFI of its own: its boundary is the Python surface of scripts/training/<name>.py
 * (SURVEY.md section 8b).  Every entry point below replaces a torch/MONAI/torchsurv op sequence invoked from
 * that surface; the replaced call site is cited per function as R/<file>:<line> (R = the reference repo).
 *
 * Conventions
 *   - plain C: raw device pointers, sizes, POD parameter blocks; no torch/HIP C++ types in signatures
 *     (hipStream_t is an opaque pointer);
 *   - the CALLER owns every buffer (PyTorch caching allocator in the shipped host code); nothing here
 *     allocates, frees or synchronises; all work is enqueued on the given stream, so the calls are legal
 *     inside hipStreamBeginCapture/EndCapture;
 *   - return 0 on success, negative on error (MMS_ERR_*); never throws;
 *   - activations are channels-last fp32 matrices [rows = B*D*H*W][channels] ("slabs"); a dense block's
 *     torch.cat is a column offset into its slab;
 *   - BatchNorm batch statistics travel as fp64 (sum, sumsq) accumulators that the producer kernel's
 *     epilogue fills with atomics; they must be zeroed (hipMemsetAsync) once per step by the caller.
 */
#ifndef MMSURV_H
#define MMSURV_H
#include <stdint.h>
#include <stddef.h>

#ifndef HIP_INCLUDE_HIP_HIP_RUNTIME_API_H
typedef struct ihipStream_t* hipStream_t;
typedef struct ihipEvent_t* hipEvent_t;
#endif

#define MMS_OK 0
#define MMS_ERR_ARG (-1)
#define MMS_ERR_LAUNCH (-2)
#define MMS_MAX_GROUP 10     /* models per fold-group launch (the *_group entry points): 10 x the largest block (Conv1BwdP) stays under the 4 KB kernarg limit */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct Dims3 { int D; int H; int W; } Dims3;

/* Launch-shape options of the DenseNet121 drivers (mms_dn121_*) and of the convolution entry points they launch.  Kernel selection is a
 * function of the ARGUMENTS: the library reads no environment variable and keeps no per-process or per-thread setting.  A zero-filled
 * block (or a NULL pointer wherever an `opts` argument is taken) selects the defaults -- what the shipped entry points and bench.py run;
 * the other values exist for A/B measurements and for the parity tests that must reach one particular kernel form.  Never stored: the
 * block is read during the call only.  Forward and backward calls on one workspace must pass the same values. */
typedef struct MmsDnOpts {
    int out_features;      /* width N of class_layers.out ([N][1024] weight, [N] bias; out / dout carry N columns), 0 = 128
                              (R/scripts/training/simple_fusion.py:163 img_feature_dim); 1 <= N <= 4096 */
    int persist_b4;        /* dense block 4 with <= 16 rows as ONE launch per pass (csrc/dn_b4.hip): 0 = both passes, 1 = forward only,
                              -1 = per-layer launches.  The persistent launches need their 8 workgroups per model co-resident: a caller
                              that may have more of them in flight than the chip has CUs passes -1 */
    int persist_b3;        /* dense block 3's forward as one persistent launch when a sample has <= 32 voxels there (csrc/dn_cl.hip: clusters
                              of 8 workgroups per sample, BatchNorm partial sums exchanged between them): 1 = on; 0 (default) / -1 = per-layer
                              launches -- measured faster at 32 voxels per sample (profiles/r04_cluster_kernels.txt) */
    int split_wgs;         /* target workgroups of a tap-split conv2 launch, 0 = 256 */
    int conv1_ksplit;      /* -1 = never split the conv1 K loop over workgroups (0 = by launch size) */
    int conv1_small;       /* whole-K 16x16-tile conv1 forward (csrc/dn_c1s.hip): 0 = launches of <= c1s_max_wgs tiles, 1 = whenever the
                              shape allows, -1 = never */
    int c1s_max_wgs;       /* 0 = 640 */
    int conv1_small_bwd;   /* whole-M conv1 backward-data with norm1's backward fused (csrc/dn_c1s.hip): 0 = up to 128 rows, -1 = never */
    int fuse_apply_rows;   /* most rows for which norm1's backward rides in the conv1 backward-data launch, 0 = 128 (32 with
                              conv1_small_bwd = -1) */
    int conv3_small;       /* all-tap 16-row conv2 kernels for small grids (csrc/dn_c3s.hip): 0 = whenever the grid fits (one or two
                              16-column output tiles per wave by launch size), 1 / 2 = force that number, -1 = never */
    int c3s_ring;          /* weight-ring depth of their one-tile form: 0 = 6; 4, 6 or 9 */
    int conv3_mt;          /* multi-tap conv2 forward: 0 = by launch size, 2 / 3 = force it on 64- / 32-row tiles, -1 = never */
    int conv3_mt32_min;    /* fewest 32-row tiles for it, 0 = 256 */
    int conv3w_mt;         /* multi-tap conv2 weight gradient: 0 = by launch size, 2 = force, -1 = never */
    int big_ng;            /* -1 = tile shapes of the 1x1x1 kernels from ONE model's work (arithmetic independent of the group size) */
    int batch_w;           /* weight-gradient launches deferred to the end of their dense block and batched over layers: 0 = every block,
                              1 = blocks 2-4 only (rounds 2-3), -1 = one launch pair per layer */
    int ms3_rows;          /* rows per conv2 weight-gradient workgroup, 0 = by launch size */
    int ms1_div;           /* divisor of the conv1 weight-gradient row chunks, 0 = by launch size */
    int c0_nwg;            /* workgroups of the pooled conv0 weight-gradient kernel, 0 = default */
    int c0f_nwg;           /* workgroups of the pooled conv0 forward kernel, 0 = default */
    int w2_packed;         /* 1: the 58 conv2 weight tensors (and their gradients) are stored [cout][tap][cin] -- "packed primary" -- instead
                              of torch's [cout][cin][taps]: the forward reads them as they are, the weight-gradient kernels write the
                              gradient in place (no pack launch, no gradient scratch, no unpack launch in the step), and the parameter
                              table carries 116 more pointers: params[364 + 2 l] = layer l's backward-data pack, params[364 + 2 l + 1] =
                              its forward MFMA-fragment pack (layers of mms_dn121_w2_fragmask; else unused) -- kept current by the caller:
                              mms_clip_adam writes them with the update (AdamP.w2_*), mms_w2_pack after any other change of the weights.
                              0: torch layout, packs and unpack inside the drivers (rounds 1-3) */
    int trans_prepass;     /* transitions: 0 = AvgPool3d(relu(norm(x))) by its own launch into the workspace (mms_pool_act), the 1x1x1 convolution
                              and its weight gradient read that pooled operand; -1 = pooled while loading, inside both GEMMs (each of the N / 16-32
                              column tiles of a row tile re-reads and re-normalises the 8 source voxels: rounds 1-3) */
} MmsDnOpts;

/* BatchNorm parameter source. train=1: batch statistics from the fp64 accumulators; train=0: running stats.
 * (torch BatchNorm3d/1d, eps 1e-5, momentum 0.1: R/scripts/training/final_multimodal.py:77-96; MONAI norm="batch")
 * The 3x3x3-convolution entry points (mms_conv3_fwd*, mms_conv3_bwd_weight*) read the block with 16-byte vector loads: every array
 * they use (and rep_stride * 8 when nrep > 1) must be 16-byte aligned, else MMS_ERR_ARG.
 * gamma == NULL: the IDENTITY (mean 0, scale 1, shift 0; every other field ignored) -- for an operand that is already normalised and
 * non-negative (the pooled transition operand of mms_pool_act): relu(identity(x)) == x exactly. */
typedef struct BnSrc {
    const double* sum;     /* [C] batch sum      (train) */
    const double* sumsq;   /* [C] batch sum x^2  (train) */
    const float* rmean;    /* [C] running mean   (eval)  */
    const float* rvar;     /* [C] running var    (eval)  */
    const float* gamma;    /* [C] */
    const float* beta;     /* [C] */
    float inv_count;       /* 1 / rows the statistics were taken over */
    float eps;
    int train;
    int nrep;              /* accumulator replicas (0/1 = one): sum[c + r*rep_stride], r < nrep, are added up by the reader */
    int rep_stride;        /* in doubles */
} BnSrc;


// ---- 1x1x1 conv (dense-layer conv1, transition conv): y = relu(bn(x)) [avg-pooled 2x2x2] @ W^T -------
typedef struct Conv1FwdP {
    const float* x; int ldx;        // input slab [Min][ldx], first K columns used
    int M;                          // output rows (pooled rows when pool=1)
    int K;                          // input channels (multiple of 32)
    const float* w;                 // [N][K] (torch Conv3d weight (N,K,1,1,1))
    int N;
    float* y; int ldy;              // output [M][ldy] (column offset already applied)
    BnSrc bn;                       // over the K input channels
    double* osum; double* osumsq;   // [N] batch-stat accumulators of y (nullptr in eval)
    int pool;                       // 1: AvgPool3d(2,2) of relu(bn(x)) before the GEMM (== conv then pool)
    Dims3 in;                       // input grid (pool=1)
    int srep; int sstride;          // statistic-accumulator replicas written by this op: workgroup x adds to replica x % srep (stride in doubles)
    /* optional split of the K (input-channel) loop over ksplit workgroups per 32x32 tile (small M: the loop is a chain of
       dependent memory round trips).  Each publishes its partial tile in `partial` [ksplit][M][N] and takes a ticket in
       `counters` [tiles]; the LAST one to arrive sums the partials in fixed order (deterministic), writes y and the
       statistics -- no second launch.  counters must be zero on entry (the fixing workgroup re-arms them). */
    float* partial; int ksplit; unsigned* counters;
} Conv1FwdP;

// ---- 3x3x3 conv, pad 1 (dense-layer conv2): slab[:, coff:coff+32] = conv3(relu(bn(y1)), W) ---------------
typedef struct Conv3FwdP {
    const float* y1;                // [M][128]
    const int* coords;              // [M] packed (d,h,w)
    Dims3 g; int M;
    const float* wp;                // packed [32][27][128]  (cout, tap, cin)
    float* out; int ldo;            // slab + coff, row pitch
    BnSrc bn;                       // over 128 channels of y1
    double* osum; double* osumsq;   // [32] (nullptr in eval)
    float* partial;                 // optional scratch [nsplit][M][32]: split the 27 taps over nsplit workgroups per tile,
                                    // then a reduce kernel sums them, writes the slab columns and the statistics
    int nsplit;                     // 1..27 (used when partial != null); workgroup z handles taps [z*ceil(27/nsplit), ...)
    int srep; int sstride;          // statistic-accumulator replicas written by this op: workgroup x adds to replica x % srep (stride in doubles)
    int wfrag;                      // 1: wp is in MFMA-fragment order [tap][cin/32][(cin/16)%2][cout/16][lane = ((cin/4)%4)*16 + cout%16][cin%4]
                                    // (mms_pack_conv3_frag): every weight load of the small-grid kernel is one contiguous 1 KB per wave.
                                    // Only with partial == null on grids that kernel takes (H*W + W + 1 <= 52), else MMS_ERR_ARG
} Conv3FwdP;

// ---- conv0: Conv3d(1,64,k7,s2,p3,no bias) -------------------------------------------------------------
typedef struct Conv0FwdP {
    const float* x;                 // [B][D][H][W]
    Dims3 in; Dims3 out;            // out = ceil(in/2)
    const int* coords;              // [M0] packed (od,oh,ow)
    int M;                          // B*out voxels
    const float* w;                 // [64][343]
    float* y;                       // [M][64]
    double* osum; double* osumsq;   // [64]
    int srep; int sstride;          // statistic-accumulator replicas written by this op: workgroup x adds to replica x % srep (stride in doubles)
} Conv0FwdP;

// ---- bn0 + relu + MaxPool3d(3,2,1) -> slab1[:, 0:64] -----------------------------------------------------
typedef struct PoolFwdP {
    const float* y0; Dims3 in;      // [B*in][64]
    Dims3 out; int B;
    float* slab; int ld;            // [B*out][ld]
    uint8_t* argmax;                // [B*out][64] tap index 0..26 of the first maximum (torch scan order)
    BnSrc bn;
    double* osum; double* osumsq;   // [64] stats of the pooled output (nullptr in eval)
    int srep; int sstride;          // statistic-accumulator replicas written by this op: workgroup x adds to replica x % srep (stride in doubles)
} PoolFwdP;

// ---- transition pre-pass: y[m'][k] = AvgPool3d(2,2)(relu(bn(x)))[m'][k] (MONAI _Transition: norm, relu, conv, pool == norm, relu, pool, conv) ----
typedef struct PoolActP {
    const float* x; int ldx;        // [B*in][ldx], first K columns used
    int K;                          // channels (multiple of 4)
    BnSrc bn;                       // over the K channels
    Dims3 in; int Mout;             // input grid (even dims); Mout = B * in / 8 pooled rows
    float* y; int ldy;              // [Mout][ldy]
} PoolActP;

// ---- norm5 + relu + global-avg-pool + Linear(1024,128) ---------------------------------------------------
typedef struct HeadFwdP {
    const float* slab; int ld; int C; int B; int V;   // [B*V][ld], C channels, V voxels per sample
    BnSrc bn;
    const float* w; const float* bias; int N;          // [N][C]
    float* pooled;                                     // [B][C] saved for backward
    float* out; int ldo;                               // [B][ldo] first N columns written
} HeadFwdP;

// =========================== backward ==================================================================
// BatchNorm backward constants of one layer: sums accumulated by the producer of dbn.
typedef struct BnBwd {
    const double* s1;      // [C] sum_m dbn
    const double* s2;      // [C] sum_m dbn * xhat
    int nrep; int rep_stride;   // replicas as in BnSrc
} BnBwd;

// conv3 backward-data: dbn2 = (dz (*) W^T) * [a2 > 0], plus BN2-backward sums
typedef struct Conv3BwdDataP {
    const float* dz; int lddz;      // dslab + coff  [M][lddz], 32 columns
    const int* coords; Dims3 g; int M;
    const float* wpb;               // packed [128][27][32] (cin, tap, cout)
    const float* y1;                // [M][128] forward pre-BN activations
    BnSrc bn;                       // bn2
    float* dbn;                     // [M][128] out
    double* s1; double* s2;         // [128] out (atomics)
    float* partial;                 // optional scratch [nsplit][M][128]: tap split as in Conv3FwdP
    int nsplit;
    int srep; int sstride;          // statistic-accumulator replicas written by this op: workgroup x adds to replica x % srep (stride in doubles)
    int wfrag;                      // 1: wpb is in MFMA-fragment order [tap][cin/16][cout/16][lane = ((cout/4)%4)*16 + cin%16][cout%4] (mms_pack_conv3_frag); as Conv3FwdP.wfrag
} Conv3BwdDataP;

// conv3 backward-weight: dW[cout][cin][tap] += sum_m relu(bn(y1))[nbr(m,tap)][cin] * dz[m][cout]
typedef struct Conv3BwdWP {
    const float* y1; const int* coords; Dims3 g; int M;
    BnSrc bn;
    const float* dz; int lddz;
    float* dw;                      // gradient, accumulated with atomics; layout by dw_layout
    int msplit;                     // grid.z = 27 * msplit
    int dw_layout;                  // 0: canonical torch layout [32 cout][128 cin][27 taps] (every atomic on its own cache line);
                                    // 1: a zeroed scratch in [27][32][128] (tap, cout, cin) order: 512-byte contiguous runs = full atomic
                                    //    rate; mms_unpack_conv3_grads adds it into the canonical gradient;
                                    // 2: [32][27][128] (cout, tap, cin) -- the PACKED PRIMARY layout of MmsDnOpts.w2_packed: the same
                                    //    512-byte runs, and the buffer IS the parameter's gradient (no scratch, no unpack)
} Conv3BwdWP;

// 1x1 conv backward (dense conv1 and transition conv).
//   dy[m][n]  = g2[n]*rstd2[n]*(dbn2[m][n] - s1[n]/M - yhat[m][n]*s2[n]/M)     (has_bn_out=1)
//             = dyraw[m][n]                                                     (has_bn_out=0, transition)
//   dW[n][k] += sum_m dy[m][n] * a[m][k],   a = relu(bn_in(x)) [avg-pooled]
//   da[m][k]  = sum_n dy[m][n] * W[n][k];   dbn_in = da * [a>0] (un-pooled /8 when pool) ; sums s1,s2 over k
typedef struct Conv1BwdP {
    // output-side gradient
    const float* dyraw; int lddy;   // [M][lddy]: dbn2 (has_bn_out) or dslab_next (transition)
    const float* y; int ldy;        // forward output of this conv (pre-BN2), needed when has_bn_out
    BnSrc bn_out; BnBwd bb_out; int has_bn_out;
    int M; int N;                   // rows of dy (pooled rows when pool), N output channels
    // input side
    const float* x; int ldx; int K; BnSrc bn_in;
    const float* w;                 // [N][K]
    int pool; Dims3 in;             // transition: x grid dims (rows of x = M*8)
    float* dw;                      // [N][K] accumulated with atomics (weight kernel)
    float* dbn; int lddbn;          // [Mx][lddbn] out (data kernel), Mx = M*(pool?8:1)
    double* s1; double* s2;         // [K] out (data kernel)
    int msplit;
    float* dgamma_out; float* dbeta_out;   // [N] BN2 parameter grads (= s2_out, s1_out), written by the weight kernel
    int srep; int sstride;          // statistic-accumulator replicas written by this op: workgroup x adds to replica x % srep (stride in doubles)
    /* optional (data kernel, pool = 0, M <= 128 rows): norm1's backward fused into the epilogue -- a workgroup then owns all rows of
       its channels, so the two BN-backward sums are complete locally: fuse_dx[:, 0:K] (+)= gamma*rstd*(dbn - s1/M - xhat*s2/M),
       fuse_dgamma += s2, fuse_dbeta += s1; dbn / s1 / s2 are not written and no mms_bn_bwd_apply launch follows. */
    float* fuse_dx; int fuse_lddx; int fuse_accumulate;
    float* fuse_dgamma; float* fuse_dbeta;
} Conv1BwdP;

// dslab[:, 0:C] (+)= g*rstd*(dbn - s1/M - xhat*s2/M)
typedef struct BnBwdApplyP {
    const float* dbn; int lddbn;
    const float* x; int ldx;
    float* dx; int lddx;
    int M; int C; BnSrc bn; BnBwd bb; int accumulate;
    float* dgamma; float* dbeta;    // [C] written by block 0 (dgamma = s2, dbeta = s1)
} BnBwdApplyP;

typedef struct HeadBwdP {
    const float* dout; int lddout;  // [B][lddout] first N columns read
    const float* pooled;            // [B][C]
    const float* slab; int ld; int C; int B; int V; BnSrc bn;
    const float* w; int N;
    float* dw; float* dbias;        // [N][C], [N]
    float* dgamma; float* dbeta;    // [C]
    float* dslab; int ldd;          // [B*V][ldd] first C columns written
    double* ext_sums;               // mms_head_bwd_sums / _apply only: [2][C] norm5-backward sums (s1 | s2), zeroed by the caller
} HeadBwdP;

typedef struct PoolBwdP {                   // maxpool backward + relu0 mask -> dbn0, sums
    const float* dslab; int ld;     // [B*out][ld] first 64 columns
    const uint8_t* argmax; Dims3 out; Dims3 in; int B;
    const float* y0; BnSrc bn;
    float* dbn;                     // [B*in][64]
    double* s1; double* s2;         // [64]
    const int* coords;              // optional [B*in] packed (d,h,w) of the conv0 grid (saves the per-voxel integer divisions)
    int srep; int sstride;          // statistic-accumulator replicas written by this op: workgroup x adds to replica x % srep (stride in doubles)
} PoolBwdP;

typedef struct Conv0BwdWP {                 // dW0[64][343] += sum_m bn0bwd(dbn0)[m][n] * x[patch(m,k)]
    const float* dbn; const float* y0; BnSrc bn; BnBwd bb;
    const float* x; Dims3 in; Dims3 out; const int* coords; int M;
    float* dw; int msplit;
    float* dgamma; float* dbeta;    // [64]
    float* dw_rep; int nrep;        // optional: nrep zeroed replicas [nrep][64*343] -- workgroup x adds to replica x % nrep and a second launch adds the
                                    // replicas into dw (256 workgroups x 22 k atomics on the 686 lines of ONE gradient serialise at the memory side);
                                    // NULL / 0: atomics straight into dw
} Conv0BwdWP;

/* =========================== fallback CT encoder (R/scripts/training/final_multimodal.py:75-86) ==================
 * 3 x [Conv3d(k3, s2, p1, bias) + BatchNorm3d + ReLU] (1->32->64->128) + AdaptiveAvgPool3d(1): what the reference
 * runs when MONAI is absent.  Channels-last activations; each conv stores its raw output (+bias) and batch
 * statistics, BN+ReLU of layer l is applied while layer l+1 (or the pooling) reads it. */
typedef struct FbConvP {
    const float* x; int Cin; Dims3 in; Dims3 out; int B;     // x: [B*in][Cin] raw output of the previous conv (or the volume, Cin=1)
    int has_bn; BnSrc bn;                                    // BN+ReLU of the previous layer (has_bn=0 for the first conv)
    const float* w; const float* bias; int Cout;             // torch layout [Cout][Cin][27], [Cout]
    float* y;                                                // [B*out][Cout]
    double* osum; double* osumsq;                            // [Cout] (nullptr in eval)
    /* backward (fb_conv_bwd_w / fb_conv_bwd_x) */
    const float* dy;                                         // [B*out][Cout] gradient w.r.t. y
    float* dw; float* dbias;                                 // accumulated (atomics)
    float* dbn_in; double* s1; double* s2;                   // [B*in][Cin] masked input gradient + BN-backward sums (bwd_x)
    int msplit;
} FbConvP;

typedef struct FbPoolP {     /* BN+ReLU+global average pool and its backward */
    const float* y; int C; int V; int B; BnSrc bn;           // [B*V][C]
    float* out; int ldo;                                     // [B][ldo] first C columns
    const float* dout; int lddout;
    float* dbn; double* s1; double* s2;
} FbPoolP;

/* =========================== heads: small-batch MLP, gate, Cox, optimizer ============================== */
/* Input prologue of a Linear layer: x' = dropout(relu(bn1d(x))) -- the BatchNorm1d/ReLU/Dropout that FOLLOW the
 * previous Linear in the reference's nn.Sequential (R/scripts/training/final_multimodal.py:93-117) are applied
 * while this layer reads its input.  All M rows of a column sit in one lane, so batch statistics need no atomics. */
typedef struct InProlog {
    int bn;                         // 1: BatchNorm1d (+ReLU) on the input columns
    const float* gamma; const float* beta;
    float* rmean; float* rvar; long long* nbt;   // running stats (read in eval, momentum-updated in train by fwd)
    float eps; float momentum;
    int train;
    float drop_p;                   // dropout probability applied after bn+relu (0 = none; ignored in eval)
    const float* drop_mask;         // optional explicit multiplicative mask [M][K] (already scaled by 1/(1-p)); parity mode
    const uint32_t* rng;            // device {seed, step} for the hash RNG when drop_mask is null
    uint32_t stream_id;             // distinguishes dropout sites
} InProlog;

typedef struct LinearFwdP {
    const float* x; int ldx; int M; int K;
    InProlog pro;
    const float* w; const float* bias; int N;     // torch Linear: w [N][K]
    float* y; int ldy; int out_relu;              // y = [relu](x' W^T + b)
} LinearFwdP;

typedef struct LinearBwdP {
    const float* dy; int lddy;      // gradient w.r.t. y (post-activation) [M][lddy]
    const float* y; int ldy;        // forward output (relu mask when out_relu)
    int out_relu;
    const float* x; int ldx; int M; int K; InProlog pro;
    const float* w; int N;
    float* dw; float* dbias;        // accumulated (+=)
    float* dx; int lddx;            // gradient w.r.t. the RAW input x (before the prologue) [M][lddx]; null = skip
    float* dgamma; float* dbeta;    // prologue BN parameter grads (+=), null when pro.bn == 0
} LinearBwdP;

/* Gated late fusion (R/scripts/training/partial_modality_training.py:257-271): feats [M][288] = ct|rna|clin,
 * mask [M][3]; masked = feats*mask; gate = softmax(W2 relu(W1 [masked|mask] + b1) + b2); fused = masked*gate. */
typedef struct GateP {
    const float* feats; const float* mask; int M;
    const float* w1; const float* b1; const float* w2; const float* b2;   // [64][291],[64],[3][64],[3]
    float* hidden;                  // [M][64] saved
    float* gate;                    // [M][3]  saved / output
    float* fused;                   // [M][288]
    // backward
    const float* dfused;            // [M][288]
    float ent_weight;               // lambda * upstream of gate_entropy_loss (R/...:322-331); 0 = none
    const float* dgate_ext;         // optional [M][3]: external gradient w.r.t. the gate weights (autograd path)
    float* dfeats;                  // [M][288]
    float* dw1; float* db1; float* dw2; float* db2;   // accumulated (atomics)
    float* entropy;                 // optional scalar out (fwd): -mean_b(entropy_b)
} GateP;

/* Cox negative partial log-likelihood (R/scripts/training/final_multimodal.py:158-186): Breslow risk sets
 * loss = -(1/n_e) sum_{i:e_i} (h_i - log sum_{j: t_j>=t_i} exp h_j), or torchsurv's Efron tie handling (tie_mode).
 * valid: optional per-sample 0/1 (has_survival); excluded samples get dh = 0.
 * out[0] = loss, out[1] = 1 if the batch is usable (>=2 valid samples and >=1 event) else 0 (loss 0, dh 0). */
typedef struct CoxP {
    const float* h; int ldh;        // log-hazards, element i at h[i*ldh]
    const float* time; const float* event; const float* valid; int n;
    float scale;                    // upstream gradient
    float* lse;                     // [n] workspace (multi-block path)
    float* dh; int lddh;            // dL/dh (null = forward only)
    float* out;                     // [2]
    int tie_mode;                   // 0: Breslow risk sets (= every formulation of the reference on distinct times);
                                    // 1: Efron tie correction as torchsurv's neg_partial_log_likelihood applies it when times
                                    //    repeat (R/scripts/training/final_multimodal.py:158-162): the m events tied at a time
                                    //    take the denominators D - (l/m) T, l = 0..m-1, and the loss is the MEAN over the
                                    //    distinct event times (on distinct times both modes give the same value)
    float* tie_frac;                // [n] workspace (tie_mode 1): l/m of each event
} CoxP;

/* Harrell C, pair counting (R/scripts/training/simple_fusion.py:59-73): counts[0]=concordant (h_i>h_j),
 * counts[1]=tied hazards, counts[2]=permissible pairs (e_i==1, t_j>t_i). */
typedef struct CindexP {
    const float* h; const float* time; const float* event; int n;
    unsigned long long* counts;     // [3], zeroed by the caller
} CindexP;

/* clip_grad_norm_(max_norm) + Adam/AdamW over one flat fp32 parameter buffer (R/...final_multimodal.py:259-260,350;
 * simple_fusion.py:273-274,391).  hyper (device): {lr, beta1, beta2, eps, weight_decay, max_norm}; state (device):
 * {step (as float), sumsq scratch (double as 2 floats)}; skip: optional device flag, 0 => no-op (degenerate batch). */
typedef struct AdamP {
    float* p; float* g; float* m; float* v; long long n;
    const float* hyper;             // [6]
    double* sumsq;                  // [1] zeroed by mms_grad_sumsq's caller each step
    float* step;                    // [1] step counter (incremented by the update kernel when not skipped)
    const float* skip_flag;         // null or device scalar: update only if *skip_flag != 0
    int adamw;                      // 0: Adam with L2-coupled weight decay; 1: AdamW (decoupled)
    /* optional per-step bookkeeping done by mms_grad_sumsq (all nullable): acc[4] += {loss*usable, usable, entropy, 1} */
    float* acc; const float* cox_out; const float* entropy;
    int* rng;                       // [2] dropout (seed, step counter): counter += 1 after the step's kernels have used it
    /* conv2 weights in packed primary storage (MmsDnOpts.w2_packed; n_w2 = 0: none).  The n_w2 tensors of 32 x 27 x 128 floats at
       offsets w2_off[i] (ascending) of p / g / m / v are stored [cout][tap][cin]; mms_clip_adam updates them with a kernel of their own
       that also writes the derived packs the convolution kernels read: w2_pack_b[i] = backward-data pack ([cin][tap][cout], or the
       backward MFMA-fragment order when bit i of w2_fragmask is set), w2_pack_f[i] = forward MFMA-fragment order (bit i set; else unused:
       the primary storage is the forward pack).  All three tables live in device memory. */
    const long long* w2_off; float* const* w2_pack_b; float* const* w2_pack_f; int n_w2; uint64_t w2_fragmask;
} AdamP;

/* Large-batch Linear (rows M > 32: BASELINE config 5, RNA-seq-only B = 2048; R/scripts/training/train_rnaseq_only.py:126-151).
 * y = P(x) W^T + bias [-> ReLU], P = Dropout(ReLU(BatchNorm1d(x))) of the PRECEDING nn.Sequential entries (has_bn = 0: P = identity
 * or dropout only).  fp32 MFMA GEMMs on the tile core; BatchNorm1d batch statistics travel like the BatchNorm3d ones: the
 * producing launch accumulates (sum, sumsq) of its output in fp64 (osum/osumsq), the consuming launch normalises in its
 * operand prologue (bn: sum/sumsq over the M rows).  Backward of one layer: mms_linear_big_bwd_w (dW, db), mms_linear_big_bwd_x
 * (gradient wrt P's output, through dropout/ReLU -> dbn + the two BN-backward column sums), mms_bn1d_bwd_apply (dx, dgamma, dbeta). */
typedef struct LinBigP {
    const float* x; int ldx; int M; int K;
    const float* w; const float* bias; int N;      // torch Linear: weight [N][K], bias [N]
    float* y; int ldy; int out_relu;
    int has_bn; BnSrc bn;                            // BatchNorm1d over the K input columns (train: batch sums of x; eval: running)
    float* rmean; float* rvar; long long* nbt; float momentum;   // running statistics, updated by the forward in train mode (nullable)
    float drop_p; const float* drop_mask; const uint32_t* rng; uint32_t stream_id; int train;
    double* osum; double* osumsq;                   // [N] statistics of y (nullable)
    /* backward */
    const float* dy; int lddy;                      // gradient wrt y (after out_relu)
    float* dw; float* dbias; int msplit;            // accumulated (atomics); rows split over msplit workgroups
    int core_only;                                  // 1: the 64x64 GEMM-core forms only, never the wide first-layer kernels (A/B measurements)
    float* dbn; int lddbn;                          // [M][K] gradient wrt BN output (has_bn) or wrt x (no BN)
    double* s1; double* s2;                         // [K] sum dbn, sum dbn * xhat (has_bn)
    float* dx; int lddx;                            // mms_bn1d_bwd_apply output
    float* dgamma; float* dbeta;
} LinBigP;

/* Learnable missing-modality bias (R/scripts/training/flexible_multimodal.py:205-206,243-250): per feature segment s
 * (image 128 | RNA 256 columns of the fused vector)   feats[m][j] = feats[m][j]*mask[m][s] + bias_s[j]*(1 - mask[m][s]),
 * in place; backward: dbias_s[j] += sum_m dfeats[m][j]*(1 - mask[m][s]);  dfeats[m][j] *= mask[m][s]. */
typedef struct MixP {
    float* feats; int ld; int M;
    const float* mask; int ldm;     // [M][ldm], column s = has-modality flag of segment s
    int nseg;                       // <= 4
    int seg_begin[4]; int seg_width[4];
    const float* bias[4];
    float* dfeats; int ldd;         // backward only
    float* dbias[4];
} MixP;

/* Batch assembly (the DataLoader collate + `.to(device)` of R/scripts/training/final_multimodal.py:228-247): row idx[b] of each
 * source array -> row b of its destination, all sources of all models of a fold group in ONE launch.  The sources live in HBM
 * (device-resident cohort) or in PINNED HOST memory (hipHostMalloc / torch pin_memory: the kernel then reads the rows over PCIe --
 * the host-to-device copy of the batch IS this launch).  present[i] != NULL: per-patient availability flag of source i's modality
 * (the cohort's mask column; R/scripts/training/partial_modality_training.py:96-141 yields all-zero tensors for a missing modality):
 * rows whose flag is 0 are zero-filled without being read (4 of 5 CT rows of BASELINE config 3's cohort). */
typedef struct GatherP {
    const long long* idx; int B;    // [B] patient indices (device)
    int nsrc;                       // sources used (<= 8)
    const float* src[8]; float* dst[8];
    int src_ld[8]; int dst_ld[8];   // row pitches in floats
    int width[8];                   // floats copied per row
    const float* present[8]; int present_ld[8];   // optional availability flag of row r: present[i][r * present_ld[i]]
} GatherP;

/* ---- ABI self-description ---- */
int mms_abi_sizeof(const char* name);      /* sizeof(struct <name>) as compiled, -1 if unknown */
int mms_abi_version(void);

/* ---- DenseNet121-3D ops (MONAI DenseNet121(spatial_dims=3,in_channels=1,out_channels=128) at
 *      R/scripts/training/final_multimodal.py:66-71, partial_modality_training.py:171-176, simple_fusion.py:182-187) */
int mms_init_coords(int* coords, int B, int D, int H, int W, hipStream_t s);
int mms_conv0_fwd(const Conv0FwdP* p, hipStream_t s);          /* features.conv0 */
int mms_pool_fwd(const PoolFwdP* p, hipStream_t s);            /* features.norm0/relu0/pool0 */
int mms_conv1_fwd(const Conv1FwdP* p, hipStream_t s);          /* denselayer norm1/relu1/conv1; transition norm/relu/conv/pool */
int mms_conv3_fwd(const Conv3FwdP* p, hipStream_t s);          /* denselayer norm2/relu2/conv2 + torch.cat */
int mms_head_fwd(const HeadFwdP* p, hipStream_t s);            /* features.norm5 + class_layers */
int mms_conv3_bwd_data(const Conv3BwdDataP* p, hipStream_t s);  /* autograd of conv2 wrt its input + relu2 mask */
int mms_conv3_bwd_weight(const Conv3BwdWP* p, hipStream_t s);    /* autograd of conv2 wrt weight */
int mms_conv1_bwd_data(const Conv1BwdP* p, hipStream_t s);       /* norm2 backward + conv1 wrt input + relu1 mask */
int mms_conv1_bwd_weight(const Conv1BwdP* p, hipStream_t s);     /* norm2 backward + conv1 wrt weight */
int mms_bn_bwd_apply(const BnBwdApplyP* p, hipStream_t s);       /* norm1 backward into the block's gradient slab */
int mms_head_bwd(const HeadBwdP* p, hipStream_t s);              /* class_layers + norm5 backward */
int mms_pool_bwd(const PoolBwdP* p, hipStream_t s);              /* pool0 + relu0 backward */
int mms_conv0_bwd_weight(const Conv0BwdWP* p, hipStream_t s);    /* norm0 backward + conv0 wrt weight */
int mms_unpack_conv3_grads(const void* table_host, int nlayers, hipStream_t s);   /* host array of {scratch [27][32][128], dw canonical}, <= 64 layers */
int mms_pack_conv3(const float* w, float* wpf, float* wpb, hipStream_t s);
int mms_pack_conv3_frag(const float* w, float* wff, float* wfb, hipStream_t s);   /* canonical [32][128][27] -> the two MFMA-fragment orders (Conv3FwdP.wfrag / Conv3BwdDataP.wfrag) */
int mms_pack_conv3_table(const void* table_dev, int nlayers, hipStream_t s);
int mms_bn_running_update(const void* table_dev, int n, float momentum, hipStream_t s);


/* ---- fallback encoder ops + whole-encoder driver (params: 12 pointers in named_parameters() order of the
 *      nn.Sequential -- {0,1,3,4,6,7}.{weight,bias}; buffers: 3 x {running_mean, running_var, num_batches_tracked}) ---- */
int mms_fb_conv_fwd(const FbConvP* p, hipStream_t s);
int mms_fb_conv_bwd_w(const FbConvP* p, hipStream_t s);
int mms_fb_conv_bwd_x(const FbConvP* p, hipStream_t s);
int mms_fb_pool_fwd(const FbPoolP* p, hipStream_t s);
int mms_fb_pool_bwd(const FbPoolP* p, hipStream_t s);
int mms_fb_workspace_bytes(int B, int D, int H, int W, size_t* bytes);
int mms_fb_init(void* ws, int B, int D, int H, int W, const void* const* buffers, hipStream_t s);
int mms_fb_forward(void* ws, int B, int D, int H, int W, const float* x, const void* const* params,
                   const void* const* buffers, float* out, int ldo, int train, hipStream_t s);
int mms_fb_backward(void* ws, int B, int D, int H, int W, const float* x, const void* const* params,
                    const float* dout, int lddout, void* const* grads, hipStream_t s);

/* ---- heads ---- */
int mms_linear_fwd(const LinearFwdP* p, hipStream_t s);        /* nn.Linear (+ preceding BN1d/ReLU/Dropout, + following ReLU) */
int mms_linear_bwd(const LinearBwdP* p, hipStream_t s);
int mms_gate_fwd(const GateP* p, hipStream_t s);
int mms_gate_bwd(const GateP* p, hipStream_t s);
int mms_gate_entropy(const float* gate, int M, float scale, float* loss, float* dgate, hipStream_t s);  /* gate_entropy_loss value (+= into *loss) and gradient */
int mms_linear_big_fwd(const LinBigP* p, hipStream_t s);
int mms_linear_big_bwd_w(const LinBigP* p, hipStream_t s);
int mms_linear_big_bwd_x(const LinBigP* p, hipStream_t s);
int mms_bn1d_bwd_apply(const LinBigP* p, hipStream_t s);
int mms_missing_mix_fwd(const MixP* p, hipStream_t s);        /* R/scripts/training/flexible_multimodal.py:243-250 */
int mms_missing_mix_bwd(const MixP* p, hipStream_t s);
int mms_cox_fwd_bwd(const CoxP* p, hipStream_t s);
int mms_cindex_counts(const CindexP* p, hipStream_t s);
int mms_grad_sumsq(const AdamP* p, hipStream_t s);             /* sum of squares of the flat gradient (fp64 atomics) */
int mms_clip_adam(const AdamP* p, hipStream_t s);              /* clip by global norm + Adam/AdamW update (+ the derived conv2 packs, AdamP.w2_*) */
int mms_w2_pack(const AdamP* p, hipStream_t s);                /* the derived conv2 packs alone, from the current weights (after load_state_dict / any external update) */

/* ---- whole-encoder driver: replaces `self.ct_encoder(ct)` / `self.image_encoder(image)` and its autograd
 *      (R/scripts/training/final_multimodal.py:124, partial_modality_training.py:245, simple_fusion.py:226).
 *      params: 364 device pointers in torch named_parameters() order of MONAI DenseNet121;
 *      buffers: 121 x {running_mean, running_var, num_batches_tracked} in module order;
 *      grads: 364 device pointers, ACCUMULATED into (caller zeroes).  D,H,W multiples of 32.            */
int mms_ablation_build(void);      /* 1: the library was built with a timing-ablation flag (MMS_CXXFLAGS=-DMMS_ABLATE_..., tools/ablate*.sh): launches
                                      may be left out of the step -- diagnostics only; bench.py refuses to report a value from such a build */
int mms_dn121_workspace_bytes(int B, int D, int H, int W, size_t* bytes);
int mms_dn121_region(int B, int D, int H, int W, const char* name, int index, size_t* off, size_t* bytes);
int mms_dn121_init(void* ws, int B, int D, int H, int W, const void* const* params, const void* const* buffers, const MmsDnOpts* opts, hipStream_t s);
/* which dense layers (bit l, l < 58) get their conv2 packs in MFMA-fragment order for this problem and these options (the layers whose
   launches take the small-grid kernels of csrc/dn_c3s.hip) */
int mms_dn121_w2_fragmask(int B, int D, int H, int W, const MmsDnOpts* opts, uint64_t* mask);
/* opts: launch-shape options incl. the width of class_layers.out (MmsDnOpts above; NULL = defaults). */
int mms_dn121_forward(void* ws, int B, int D, int H, int W, const float* x, const void* const* params,
                      const void* const* buffers, float* out, int ldo, int train, const MmsDnOpts* opts, hipStream_t s);
int mms_dn121_backward(void* ws, int B, int D, int H, int W, const float* x, const void* const* params,
                       const float* dout, int lddout, void* const* grads, const MmsDnOpts* opts, hipStream_t s);
/* ---- data-parallel (one process per GPU) variants of the single-model drivers --------------------------------------------
 * The reference is single-process (R/scripts/training/final_multimodal.py:52,345); BASELINE config 4 shards a global batch over
 * ranks.  Two things then leave the rank: (a) the gradients, all-reduced bucket by bucket while the backward still runs -- the
 * backward is issued in STAGES, dense blocks block_hi..block_lo (3 = denseblock4 + norm5/class_layers + transition3, ..., 0 =
 * denseblock1 + stem), each finalising one contiguous range of the parameter table (incl. its conv2 gradient unpack);
 * (b) with SyncBN, every BatchNorm statistic: the drivers call `hook` right after each kernel that produced accumulator words and
 * before the first kernel that consumes them; the hook all-reduces (SUM over ranks, in place) the words
 *     base[r * rep_stride + j * pair_stride + c],  r < nrep, j < 2, c < ncols        (fp64)
 * on stream s and returns 0.  bn_world = ranks the statistics span (counts become bn_world * rows); hook may be null (bn_world 1).
 * mms_dn121_init_sync: as mms_dn121_init, running-statistics table built for bn_world * rows. */
typedef int (*mms_sync_fn)(void* user, double* base, int nrep, long rep_stride, int ncols, long pair_stride, hipStream_t s);
int mms_dn121_init_sync(void* ws, int B, int D, int H, int W, const void* const* params, const void* const* buffers, int bn_world,
                        const MmsDnOpts* opts, hipStream_t s);
int mms_dn121_forward_sync(void* ws, int B, int D, int H, int W, const float* x, const void* const* params,
                           const void* const* buffers, float* out, int ldo, int bn_world, mms_sync_fn hook, void* user,
                           const MmsDnOpts* opts, hipStream_t s);
int mms_dn121_backward_stage(void* ws, int B, int D, int H, int W, const float* x, const void* const* params,
                             const float* dout, int lddout, void* const* grads, int block_hi, int block_lo,
                             int bn_world, mms_sync_fn hook, void* user, const MmsDnOpts* opts, hipStream_t s);
int mms_head_bwd_sums(const HeadBwdP* p, hipStream_t s);         /* SyncBN split of mms_head_bwd: masked gradient stash + local sums */
int mms_head_bwd_apply(const HeadBwdP* p, hipStream_t s);        /* ... norm5 backward with the (all-reduced) sums + class_layers.out gradients */
/* same, with the per-layer weight-gradient kernels forked onto `side` (caller-created stream and two events): they
 * are off the critical path dslab -> dbn2 -> dbn1 -> dslab, so under graph capture they become parallel branches. */
int mms_dn121_backward_mt(void* ws, int B, int D, int H, int W, const float* x, const void* const* params,
                          const float* dout, int lddout, void* const* grads, const MmsDnOpts* opts, hipStream_t s, hipStream_t side,
                          hipEvent_t ev_fork, hipEvent_t ev_join);

/* ---- preprocessing upstream of the path, on the GPU (replaces per-item numpy/scipy on the CPU) ----
 * mms_ct_preprocess: (x - min) / (max - min + 1e-8) and scipy.ndimage.zoom(order=1) to the network grid
 *   (R/scripts/training/partial_modality_training.py:94-109, simple_fusion.py:117-134); scratch >= 512 floats.
 * mms_rna_log_zscore: log2(count + 1), then per-gene StandardScaler over the n samples
 *   (R/scripts/preprocessing/preprocess_genomic.py:108-117). */
int mms_ct_preprocess(const float* x, int inD, int inH, int inW, float* out, int oD, int oH, int oW, float* scratch, hipStream_t s);
int mms_rna_log_zscore(const float* counts, float* out, int n, int g, hipStream_t s);

/* =========================== fold groups =====================================================================
 * The reference trains its K fold models one after the other (R/scripts/training/final_multimodal.py:316-402,
 * partial_modality_training.py:482-560, simple_fusion.py:318-436); they are independent, so this library can advance
 * ng <= MMS_MAX_GROUP of them in lock-step: every *_group entry point takes an ARRAY of ng parameter blocks (one per
 * model, identical shapes, any pointers) and issues ONE launch whose grid carries the model index as an extra
 * dimension.  Per-model arithmetic is exactly that of the single-model entry point (which is the ng = 1 case of the
 * same kernel); a batch-4 step of one model cannot fill 256 CUs, a group of them can.
 * The convolution entry points that have several kernel forms take the launch-shape options (MmsDnOpts; NULL = defaults). */
int mms_conv0_fwd_group(const Conv0FwdP* p, int ng, const MmsDnOpts* opts, hipStream_t s);
int mms_pool_fwd_group(const PoolFwdP* p, int ng, hipStream_t s);
int mms_pool_act_group(const PoolActP* p, int ng, hipStream_t s);       /* transition.norm / relu / pool (before transition.conv) */
int mms_conv1_fwd_group(const Conv1FwdP* p, int ng, const MmsDnOpts* opts, hipStream_t s);
int mms_conv3_fwd_group(const Conv3FwdP* p, int ng, const MmsDnOpts* opts, hipStream_t s);
int mms_head_fwd_group(const HeadFwdP* p, int ng, hipStream_t s);
int mms_conv3_bwd_data_group(const Conv3BwdDataP* p, int ng, const MmsDnOpts* opts, hipStream_t s);
int mms_conv3_bwd_weight_group(const Conv3BwdWP* p, int ng, const MmsDnOpts* opts, hipStream_t s);
/* Conv3BwdWP.msplit the whole-encoder drivers use for a launch of `members` (model, layer) members of M rows each (0: bad arguments) --
   so that a caller timing the launch alone (bench.py's roofline leg) shapes it exactly as the step does */
int mms_conv3_bwd_weight_msplit(int M, int members, const MmsDnOpts* opts);
int mms_conv1_bwd_data_group(const Conv1BwdP* p, int ng, const MmsDnOpts* opts, hipStream_t s);
int mms_conv1_bwd_weight_group(const Conv1BwdP* p, int ng, hipStream_t s);
int mms_bn_bwd_apply_group(const BnBwdApplyP* p, int ng, hipStream_t s);
int mms_head_bwd_group(const HeadBwdP* p, int ng, hipStream_t s);
int mms_pool_bwd_group(const PoolBwdP* p, int ng, hipStream_t s);
int mms_conv0_bwd_weight_group(const Conv0BwdWP* p, int ng, const MmsDnOpts* opts, hipStream_t s);
int mms_pack_conv3_table_group(const void* const* tables_dev, int ng, int nlayers, hipStream_t s);
int mms_pack_conv3_table_group_ex(const void* const* tables_dev, int ng, int nlayers, uint64_t fragmask, hipStream_t s);   /* bit l set: layer l's two packs in MFMA-fragment order */
int mms_bn_running_update_group(const void* const* tables_dev, int ng, int n, float momentum, hipStream_t s);
int mms_missing_mix_fwd_group(const MixP* p, int ng, hipStream_t s);
int mms_missing_mix_bwd_group(const MixP* p, int ng, hipStream_t s);
int mms_gather_rows_group(const GatherP* p, int ng, hipStream_t s);
int mms_unpack_conv3_grads_group(const float* const* scratch, float* const* const* dw, int ng, int nlayers, hipStream_t s);  /* scratch[g]: model g's [nlayers][27][32][128] tap-major scratch; dw[g][i]: canonical gradient of layer i; nlayers <= 58 */
int mms_zero_regions_group(void* const* regions_dev, int ng, size_t bytes, hipStream_t s);   /* 16-B aligned regions of equal size, zero-filled by one launch */
int mms_linear_fwd_group(const LinearFwdP* p, int ng, hipStream_t s);
int mms_linear_bwd_group(const LinearBwdP* p, int ng, hipStream_t s);
int mms_gate_fwd_group(const GateP* p, int ng, hipStream_t s);
int mms_gate_bwd_group(const GateP* p, int ng, hipStream_t s);
int mms_cox_fwd_bwd_group(const CoxP* p, int ng, hipStream_t s);
int mms_grad_sumsq_group(const AdamP* p, int ng, hipStream_t s);
int mms_clip_adam_group(const AdamP* p, int ng, hipStream_t s);
int mms_w2_pack_group(const AdamP* p, int ng, hipStream_t s);
/* whole-encoder drivers: entry g of every array describes model g (arguments as mms_dn121_forward / _backward) */
int mms_dn121_forward_group(int ng, void* const* ws, int B, int D, int H, int W, const float* const* x,
                            const void* const* const* params, const void* const* const* buffers, float* const* out,
                            int ldo, int train, const MmsDnOpts* opts, hipStream_t s);
int mms_dn121_backward_group(int ng, void* const* ws, int B, int D, int H, int W, const float* const* x,
                             const void* const* const* params, const float* const* dout, int lddout,
                             void* const* const* grads, const MmsDnOpts* opts, hipStream_t s);

#ifdef __cplusplus
}
#endif
#endif /* MMSURV_H */
