// DenseNet121-3D network driver: the whole encoder forward / backward as one C-ABI call each.
// Host-only code (no kernels besides the ones it launches): computes the workspace layout for a (B, D, H, W)
// problem, enqueues the fused ops of dn_fwd.hip / dn_bwd.hip on the caller's stream in dependency order.
// No allocation, no synchronisation (graph-capturable); mms_dn121_init is the only call that copies tables.
//
// Topology restated from MONAI DenseNet121(spatial_dims=3, in_channels=1, out_channels=128):
// init_features 64, growth 32, bn_size 4, block_config (6,12,24,16) -- see oracle/densenet3d.py.
// Parameter table order == torch named_parameters() order of that module (364 tensors):
//   conv0.w, norm0.{w,b}, per dense layer {norm1.w, norm1.b, conv1.w, norm2.w, norm2.b, conv2.w},
//   after blocks 1-3 transition{norm.w, norm.b, conv.w}, norm5.{w,b}, class_layers.out.{w,b}
// Buffer table order: per BatchNorm in module order {running_mean, running_var, num_batches_tracked} (121 BNs).
#include "dn_ops.h"
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

namespace {

constexpr int NB = 4;
constexpr int LAYERS[NB] = {6, 12, 24, 16};
constexpr int C0[NB] = {64, 128, 256, 512};
constexpr int CTOT[NB] = {256, 512, 1024, 1024};
constexpr int NLAYER = 58;
constexpr int NBN = 121;
constexpr int NPARAM = 364;


struct Plan {
    int B; Dims3 in, g0, g[NB];
    int M0, M[NB];
    long partial_rows;  // capacity of the tap-split scratch in [128]-float rows
    int R0, R[NB];     // statistic-accumulator replicas of the stem level / of each block (common.h: stat_rep)
    // byte offsets into the workspace
    size_t coords0, coords[NB], y0, argmax, slab[NB], dslab[NB], y1[NLAYER], wpf[NLAYER], wpb[NLAYER];
    size_t dbn_mid_l[NLAYER], dbn_in, dbn0, pooled, tab_pack, tab_bn, partial, dwp[NLAYER];
    // fp64 statistic accumulators (one contiguous region, zeroed once per step)
    size_t dw0_rep;                 // 8 replicas of the conv0 weight gradient (inside the zeroed region)
    size_t tpool[3];                // pooled transition operands [M[t + 1]][CTOT[t]] (forward pre-pass; read again by the weight gradient)
    size_t stats_begin, stats_begin_packed, stats_end;      // (packed primary conv2 storage: the gradient scratch dwp at the head of the region is unused)
    size_t st_y0, st_slab[NB], st_y1[NLAYER];          // forward (sum | sumsq), each 2*C doubles
    size_t counters;                                   // split-fixup tickets (zeroed at init, re-armed by their users)
    // persistent per-block ("cluster") kernels of dense blocks 3 / 4 (dn_cl.hip): geometry (row tiles per cluster: 0 = the block does not
    // qualify; rows per cluster; clusters), granule hand-off buffers of the forward and of block 4's backward (inside the per-step zeroed
    // region), layer tables; sticky error word
    int cl_rt[NB], cl_rpc[NB], cl_ncl[NB];
    size_t cl_xa[NB], cl_xb[NB], cl_gst[NB], cl_tab[NB], cl_zero_begin, cl_ga, cl_gz;
    size_t b4_err;
    size_t bb_y0, bb_y1[NLAYER], bb_in[NLAYER], bb_tr[3], bb_head;   // backward (s1 | s2)
    size_t total;
};

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

bool make_plan(Plan& P, int B, int D, int H, int W) {
    if (B <= 0 || D < 32 || H < 32 || W < 32 || (D % 32) || (H % 32) || (W % 32)) return false;   // 5 halvings, even dims at each
    if (D > 1023 * 2 || H > 1023 * 2 || W > 1023 * 2) return false;
    P.B = B; P.in = Dims3{D, H, W};
    P.g0 = Dims3{D / 2, H / 2, W / 2};
    P.g[0] = Dims3{D / 4, H / 4, W / 4};
    for (int b = 1; b < NB; ++b) P.g[b] = Dims3{P.g[b - 1].D / 2, P.g[b - 1].H / 2, P.g[b - 1].W / 2};
    P.M0 = B * P.g0.D * P.g0.H * P.g0.W;
    for (int b = 0; b < NB; ++b) P.M[b] = B * P.g[b].D * P.g[b].H * P.g[b].W;
    // statistic-accumulator replicas of a level: one per 2048 rows, at most 8 -- the stem (65536 rows) -> 8, block 1 of 64x64x32 volumes
    // (8192 rows) -> 4, blocks 2-4 -> 1.  Replicas relieve the producers' fp64 atomics: 256 workgroups adding to the same four cache lines
    // serialise at the memory side, ~23 ns per line request -- 6 us of a 30 us block-1 conv2 launch of one model (tools/build_variant.sh
    // with -DMMS_ABLATE_STATS); every consumer workgroup re-adds them in its prologue, three replicas per memory round trip (rep_add,
    // common.h).  History: with a dependent round trip PER replica in the consumers, one replica per 2048 rows measured 2283-2296 patients/s on
    // the K = 5 epoch against 2322-2325 for one per 8192 rows (rounds 2-3).  (Part of the workspace LAYOUT, so a constant.)
    constexpr int rep_rows = 2048;
    auto reps = [](int M) { int r = 1; while (r < 8 && M / (2 * r) >= rep_rows) r *= 2; return r; };
    P.R0 = reps(P.M0);
    for (int b = 0; b < NB; ++b) P.R[b] = reps(P.M[b]);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = al(o + bytes); return r; };
    P.coords0 = take((size_t)P.M0 * 4);
    for (int b = 0; b < NB; ++b) P.coords[b] = take((size_t)P.M[b] * 4);
    P.y0 = take((size_t)P.M0 * 64 * 4);
    P.argmax = take((size_t)P.M[0] * 64);
    for (int b = 0; b < NB; ++b) P.slab[b] = take((size_t)P.M[b] * CTOT[b] * 4);
    for (int b = 0; b < NB; ++b) P.dslab[b] = take((size_t)P.M[b] * CTOT[b] * 4);
    int l = 0;
    for (int b = 0; b < NB; ++b)
        for (int i = 0; i < LAYERS[b]; ++i, ++l) {
            P.y1[l] = take((size_t)P.M[b] * 128 * 4);
            P.wpf[l] = take((size_t)32 * 27 * 128 * 4);
            P.wpb[l] = take((size_t)32 * 27 * 128 * 4);
        }
    {   // conv2's input gradient of a layer: one buffer per layer, so that the layers' weight-gradient kernels (off the backward's
        // critical chain) can run batched over layers at the end of their block (round 4: block 1 too -- 6 x 4 MB per model at batch 4)
        int l2 = 0;
        for (int b = 0; b < NB; ++b)
            for (int i = 0; i < LAYERS[b]; ++i, ++l2) P.dbn_mid_l[l2] = take((size_t)P.M[b] * 128 * 4);
    }
    size_t mx = 0;
    for (int b = 0; b < NB; ++b) { size_t v = (size_t)P.M[b] * CTOT[b] * 4; if (v > mx) mx = v; }
    P.dbn_in = take(mx);
    P.dbn0 = take((size_t)P.M0 * 64 * 4);
    P.pooled = take((size_t)B * 1024 * 4);
    {   // tap-split scratch: 27 partials for blocks with M <= 1024, NSPLIT_BIG for larger ones
        size_t a = (size_t)27 * (P.M[1] > 1024 ? 1024 : P.M[1]) * 128 * 4, b2 = (size_t)3 * P.M[0] * 128 * 4;
        P.partial = take(a > b2 ? a : b2);
        P.partial_rows = (long)((a > b2 ? a : b2) / (128 * 4));
    }
    P.tab_pack = take(sizeof(PackEntry) * NLAYER);
    P.tab_bn = take(sizeof(BnRunEntry) * NBN);
    P.stats_begin = o;
    for (int i = 0; i < NLAYER; ++i) P.dwp[i] = take((size_t)27 * 32 * 128 * 4);    // tap-major conv2 gradient scratch (zeroed with the stats)
    P.stats_begin_packed = o;
    int blk_of[NLAYER];
    { int l2 = 0; for (int b = 0; b < NB; ++b) for (int i = 0; i < LAYERS[b]; ++i) blk_of[l2++] = b; }
    P.st_y0 = take((size_t)P.R0 * 2 * 64 * 8);
    for (int b = 0; b < NB; ++b) P.st_slab[b] = take((size_t)P.R[b] * 2 * CTOT[b] * 8);
    for (int i = 0; i < NLAYER; ++i) P.st_y1[i] = take((size_t)P.R[blk_of[i]] * 2 * 128 * 8);
    P.bb_y0 = take((size_t)P.R0 * 2 * 64 * 8);
    for (int i = 0; i < NLAYER; ++i) P.bb_y1[i] = take((size_t)P.R[blk_of[i]] * 2 * 128 * 8);
    for (int i = 0; i < NLAYER; ++i) P.bb_in[i] = take((size_t)P.R[blk_of[i]] * 2 * 1024 * 8);
    for (int i = 0; i < 3; ++i) P.bb_tr[i] = take((size_t)P.R[i] * 2 * 1024 * 8);
    P.bb_head = take((size_t)2 * 1024 * 8);
    P.cl_zero_begin = o;
    P.cl_ga = take((size_t)2 * 8 * 256 * 8);          // backward of a single-cluster block 4: hand-off A by layer parity, the dz broadcast
    P.cl_gz = take((size_t)512 * 8);
    for (int b = 0; b < NB; ++b) {
        // a block qualifies for the cluster kernels when a sample has <= 32 voxels: clusters of whole samples with <= 16 * RT rows
        const int vox = P.g[b].D * P.g[b].H * P.g[b].W;
        int rt = b >= 2 ? (vox <= 16 ? 1 : (vox <= 32 ? 2 : 0)) : 0;
        int spc = rt ? (16 * rt) / vox : 0;
        if (spc > B) spc = B;
        int ncl = rt ? (B + spc - 1) / spc : 0;
        if (ncl > 8 || P.R[b] != 1) rt = 0;
        P.cl_rt[b] = rt; P.cl_rpc[b] = rt ? spc * vox : 0; P.cl_ncl[b] = rt ? ncl : 0;
        P.cl_xa[b] = P.cl_xb[b] = P.cl_gst[b] = 0;
        if (rt) {      // {tag, value} granules of the hand-offs: zero before every launch
            P.cl_xa[b] = take((size_t)ncl * 8 * rt * 256 * 8);
            P.cl_xb[b] = take((size_t)ncl * 8 * rt * 512 * 8);
            P.cl_gst[b] = take((size_t)ncl * 8 * 192 * 8);
        }
    }
    P.dw0_rep = take((size_t)8 * 64 * 343 * 4);
    P.stats_end = o;
    for (int t = 0; t < 3; ++t) P.tpool[t] = take((size_t)P.M[t + 1] * CTOT[t] * 4);
    P.counters = take(4096 * 4);
    P.b4_err = take(1024);
    for (int b = 0; b < NB; ++b) P.cl_tab[b] = take(sizeof(B4Layer) * LAYERS[b]);
    P.total = o;
    return true;
}

// parameter-table indexing
struct Idx {
    int conv0 = 0, n0w = 1, n0b = 2;
    int layer[NLAYER];          // first of the 6 tensors of dense layer l
    int trans[3];               // first of the 3 tensors of transition t
    int n5w, n5b, outw, outb;
    int bn_layer1[NLAYER], bn_layer2[NLAYER], bn_trans[3], bn5, bn0 = 0;   // BatchNorm ordinal (buffer table)
    Idx() {
        int p = 3, q = 1, l = 0;
        for (int b = 0; b < NB; ++b) {
            for (int i = 0; i < LAYERS[b]; ++i, ++l) { layer[l] = p; p += 6; bn_layer1[l] = q++; bn_layer2[l] = q++; }
            if (b < 3) { trans[b] = p; p += 3; bn_trans[b] = q++; }
        }
        n5w = p++; n5b = p++; bn5 = q++; outw = p++; outb = p++;
    }
};
const Idx IDX;

template <class T> inline T* at(void* ws, size_t off) { return (T*)((char*)ws + off); }

inline BnSrc mk_bn(void* ws, size_t st_off, int Ctot_, const float* const* prm, int iw, const void* const* buf, int bn_ord,
                   int count, int train, int nrep) {
    BnSrc b;
    b.nrep = nrep; b.rep_stride = 2 * Ctot_;
    b.sum = at<double>(ws, st_off);
    b.sumsq = at<double>(ws, st_off) + Ctot_;
    b.rmean = buf ? (const float*)buf[3 * bn_ord] : nullptr;
    b.rvar = buf ? (const float*)buf[3 * bn_ord + 1] : nullptr;
    b.gamma = prm[iw]; b.beta = prm[iw + 1];
    b.inv_count = 1.0f / (float)count; b.eps = 1e-5f; b.train = train;
    return b;
}

}  // namespace

extern "C" int mms_init_coords(int*, int, int, int, int, hipStream_t);
extern "C" int mms_pack_conv3_table_group_ex(const void* const*, int, int, uint64_t, hipStream_t);
extern "C" int mms_unpack_conv3_grads(const void*, int, hipStream_t);
extern "C" int mms_unpack_conv3_grads_group(const float* const*, float* const* const*, int, int, hipStream_t);
extern "C" int mms_bn_running_update_group(const void* const*, int, int, float, hipStream_t);
extern "C" int mms_zero_regions_group(void* const*, int, size_t, hipStream_t);
extern "C" int mms_conv0_fwd_group(const Conv0FwdP*, int, const MmsDnOpts*, hipStream_t);
extern "C" int mms_pool_fwd_group(const PoolFwdP*, int, hipStream_t);
extern "C" int mms_pool_act_group(const PoolActP*, int, hipStream_t);
extern "C" int mms_conv1_fwd_group(const Conv1FwdP*, int, const MmsDnOpts*, hipStream_t);
extern "C" int mms_conv3_fwd_group(const Conv3FwdP*, int, const MmsDnOpts*, hipStream_t);
extern "C" int mms_head_fwd_group(const HeadFwdP*, int, hipStream_t);
extern "C" int mms_conv3_bwd_data_group(const Conv3BwdDataP*, int, const MmsDnOpts*, hipStream_t);
extern "C" int mms_conv3_bwd_weight_group(const Conv3BwdWP*, int, const MmsDnOpts*, hipStream_t);
extern "C" int mms_conv1_bwd_data_group(const Conv1BwdP*, int, const MmsDnOpts*, hipStream_t);
extern "C" int mms_conv1_bwd_weight_group(const Conv1BwdP*, int, hipStream_t);
extern "C" int mms_bn_bwd_apply_group(const BnBwdApplyP*, int, hipStream_t);
extern "C" int mms_head_bwd_group(const HeadBwdP*, int, hipStream_t);
extern "C" int mms_head_bwd_sums(const HeadBwdP*, hipStream_t);
extern "C" int mms_head_bwd_apply(const HeadBwdP*, hipStream_t);
extern "C" int mms_pool_bwd_group(const PoolBwdP*, int, hipStream_t);
extern "C" int mms_conv0_bwd_weight_group(const Conv0BwdWP*, int, const MmsDnOpts*, hipStream_t);

// Width of class_layers.out (MONAI DenseNet121 `out_channels`): 128 in the three hot-path models (final_multimodal.py:66-71), a free
// constructor argument (img_feature_dim) in simple_fusion.py:163 / flexible_multimodal.py -- MmsDnOpts.out_features (0 = 128).
static inline int out_features_of(const MmsDnOpts& o) { return o.out_features > 0 ? o.out_features : 128; }

#ifdef MMS_ABLATE_STEP
// TIMING-ABLATION BUILDS ONLY (MMS_CXXFLAGS=-DMMS_ABLATE_STEP, tools/ablate_step.sh; never in the shipped library): MMS_DEBUG_SKIP (64-bit
// mask, strtoull base 0) leaves the named launches out of the step -- results are then wrong; what the step gains without a kernel class
// bounds what optimising that class can gain.
// bits 0-3 conv1 fwd of dense block 1-4 | 4-7 conv2 fwd | 8-11 conv2 bwd-data | 12-15 conv2 bwd-weight | 16-19 conv1 bwd-weight |
// 20-23 conv1 bwd-data | 24-27 bn_bwd_apply | 28 stem fwd | 29 stem bwd | 30 transitions fwd | 31 transitions bwd | 32 / 33 block-4
// persistent fwd / bwd
static inline bool dbg_skip(int bit) {
    const char* e = getenv("MMS_DEBUG_SKIP");
    if (!e) return false;
    const unsigned long long m = strtoull(e, nullptr, 0);
    static std::once_flag once;
    if (m) std::call_once(once, [m] { fprintf(stderr, "mmsurv: MMS_DEBUG_SKIP=0x%llx -- TIMING ABLATION BUILD, launches are left out of every step: RESULTS ARE WRONG\n", m); });
    return (m >> bit) & 1ull;
}
#define TRYS(bit, x) do { if (!dbg_skip(bit)) TRY(x); } while (0)
#else
#define TRYS(bit, x) TRY(x)
#endif
#define TRY(x) do { int rc_ = (x); if (rc_ != MMS_OK) { fprintf(stderr, "mmsurv: %s -> %d (dn_net.hip:%d)\n", #x, rc_, __LINE__); return rc_; } } while (0)

extern "C" int mms_dn121_workspace_bytes(int B, int D, int H, int W, size_t* bytes) {
    Plan P;
    if (!make_plan(P, B, D, H, W) || !bytes) return MMS_ERR_ARG;
    *bytes = P.total;
    return MMS_OK;
}

// Named workspace regions, for tests/diagnostics: returns byte offset and size.
extern "C" int mms_dn121_region(int B, int D, int H, int W, const char* name, int index, size_t* off, size_t* bytes) {
    Plan P;
    if (!make_plan(P, B, D, H, W)) return MMS_ERR_ARG;
    auto set = [&](size_t o, size_t n) { *off = o; *bytes = n; return MMS_OK; };
    if (!strcmp(name, "y0")) return set(P.y0, (size_t)P.M0 * 64 * 4);
    if (!strcmp(name, "slab") && index >= 0 && index < NB) return set(P.slab[index], (size_t)P.M[index] * CTOT[index] * 4);
    if (!strcmp(name, "dslab") && index >= 0 && index < NB) return set(P.dslab[index], (size_t)P.M[index] * CTOT[index] * 4);
    if (!strcmp(name, "y1") && index >= 0 && index < NLAYER) {
        int b = 0, l = index;
        while (l >= LAYERS[b]) { l -= LAYERS[b]; ++b; }
        return set(P.y1[index], (size_t)P.M[b] * 128 * 4);
    }
    if (!strcmp(name, "tpool") && index >= 0 && index < 3) return set(P.tpool[index], (size_t)P.M[index + 1] * CTOT[index] * 4);
    if (!strcmp(name, "stats")) return set(P.stats_begin, P.stats_end - P.stats_begin);
    if (!strcmp(name, "b4_err")) return set(P.b4_err, 1024);
    return MMS_ERR_ARG;
}

// One-time (per workspace / per parameter-pointer set) initialisation: coordinate tables + device tables.
static int dn121_init_impl(void* ws, int B, int D, int H, int W, const void* const* params,
                           const void* const* buffers, int bn_world, const MmsDnOpts* opts, hipStream_t s) {
    Plan P;
    const bool packed = opts && opts->w2_packed;      // conv2 weights in packed primary storage: params[364 + 2 l (+ 1)] = the derived packs
    if (!make_plan(P, B, D, H, W) || !ws || !params || !buffers || bn_world < 1) return MMS_ERR_ARG;
    if (packed) for (int l2 = 0; l2 < NLAYER; ++l2) if (!params[NPARAM + 2 * l2]) return MMS_ERR_ARG;
    TRY(mms_init_coords(at<int>(ws, P.coords0), B, P.g0.D, P.g0.H, P.g0.W, s));
    for (int b = 0; b < NB; ++b) TRY(mms_init_coords(at<int>(ws, P.coords[b]), B, P.g[b].D, P.g[b].H, P.g[b].W, s));
    (void)hipGetLastError();
    PackEntry pk[NLAYER];
    BnRunEntry bn[NBN];
    auto set_bn = [&](int ord, size_t st, int Ctot_, int C, int count, int nrep) {
        bn[ord].nrep = nrep; bn[ord].rep_stride = 2 * Ctot_;
        bn[ord].sum = at<double>(ws, st); bn[ord].sumsq = at<double>(ws, st) + Ctot_;
        bn[ord].rmean = (float*)buffers[3 * ord]; bn[ord].rvar = (float*)buffers[3 * ord + 1];
        bn[ord].nbt = (long long*)buffers[3 * ord + 2]; bn[ord].C = C; bn[ord].count = (float)count * (float)bn_world;
    };
    set_bn(0, P.st_y0, 64, 64, P.M0, P.R0);
    int l = 0;
    for (int b = 0; b < NB; ++b) {
        int C = C0[b];
        for (int i = 0; i < LAYERS[b]; ++i, ++l, C += 32) {
            pk[l].w = (const float*)params[IDX.layer[l] + 5];
            pk[l].wpf = at<float>(ws, P.wpf[l]); pk[l].wpb = at<float>(ws, P.wpb[l]);
            set_bn(IDX.bn_layer1[l], P.st_slab[b], CTOT[b], C, P.M[b], P.R[b]);
            set_bn(IDX.bn_layer2[l], P.st_y1[l], 128, 128, P.M[b], P.R[b]);
        }
        if (b < 3) set_bn(IDX.bn_trans[b], P.st_slab[b], CTOT[b], CTOT[b], P.M[b], P.R[b]);
        else set_bn(IDX.bn5, P.st_slab[b], CTOT[b], CTOT[b], P.M[b], P.R[b]);
    }
    if (hipMemsetAsync(at<void>(ws, P.counters), 0, 4096 * 4, s) != hipSuccess) return MMS_ERR_LAUNCH;
    if (hipMemsetAsync(at<void>(ws, P.b4_err), 0, 1024, s) != hipSuccess) return MMS_ERR_LAUNCH;
    B4Layer clt[NLAYER];            // layer tables of the cluster kernels (every block: the table is tiny)
    for (int li = 0; li < NLAYER; ++li) {
        const int ip = IDX.layer[li], o1 = IDX.bn_layer1[li], o2 = IDX.bn_layer2[li];
        clt[li] = B4Layer{(const float*)params[ip], (const float*)params[ip + 1], (const float*)params[ip + 2],
                          (const float*)params[ip + 3], (const float*)params[ip + 4],
                          packed ? (const float*)params[ip + 5] : at<float>(ws, P.wpf[li]),              // (the cluster kernels read the classic packs:
                          packed ? (const float*)params[NPARAM + 2 * li] : at<float>(ws, P.wpb[li]),     //  [co][tap][cin] = the packed primary storage itself)
                          (const float*)buffers[3 * o1], (const float*)buffers[3 * o1 + 1], (const float*)buffers[3 * o2], (const float*)buffers[3 * o2 + 1],
                          at<float>(ws, P.y1[li]), at<double>(ws, P.st_y1[li]), at<float>(ws, P.dbn_mid_l[li]), at<double>(ws, P.bb_y1[li])};
    }
    hipError_t e0 = hipSuccess;
    {
        int l0 = 0;
        for (int b = 0; b < NB; ++b) {
            const hipError_t e = hipMemcpyAsync(at<void>(ws, P.cl_tab[b]), clt + l0, sizeof(B4Layer) * LAYERS[b], hipMemcpyHostToDevice, s);
            if (e != hipSuccess) e0 = e;
            l0 += LAYERS[b];
        }
    }
    if (e0 != hipSuccess) return MMS_ERR_LAUNCH;
    hipError_t e1 = hipMemcpyAsync(at<void>(ws, P.tab_pack), pk, sizeof(pk), hipMemcpyHostToDevice, s);
    hipError_t e2 = hipMemcpyAsync(at<void>(ws, P.tab_bn), bn, sizeof(bn), hipMemcpyHostToDevice, s);
    hipError_t e3 = hipStreamSynchronize(s);   // pk/bn are stack arrays
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
        fprintf(stderr, "mms_dn121_init: memcpy/sync failed: %s / %s / %s\n", hipGetErrorString(e1), hipGetErrorString(e2), hipGetErrorString(e3));
        return MMS_ERR_LAUNCH;
    }
    return MMS_OK;
}

extern "C" int mms_dn121_init(void* ws, int B, int D, int H, int W, const void* const* params,
                              const void* const* buffers, const MmsDnOpts* opts, hipStream_t s) {
    return dn121_init_impl(ws, B, D, H, W, params, buffers, 1, opts, s);
}
extern "C" int mms_dn121_init_sync(void* ws, int B, int D, int H, int W, const void* const* params,
                                   const void* const* buffers, int bn_world, const MmsDnOpts* opts, hipStream_t s) {
    return dn121_init_impl(ws, B, D, H, W, params, buffers, bn_world, opts, s);
}

// Tap split of the 3x3x3 convolutions (forward and backward-data): a launch should put about `target` workgroups on the
// 256 CUs.  One model's block with M <= 1024 rows has <= 32 row tiles, so its 27 taps are spread over workgroups and
// summed by a reduce kernel; a fold group multiplies the tiles by ng and needs less (or no) splitting.
static int split_target(const MmsDnOpts& o) {        // 256 workgroups.  (Round 1 used 864 for a single model: its blocks 2-4 took the 27-way split; they now run on the
                                                     // small-grid kernels, and block 1 of ONE model is faster unsplit on the multi-tap kernel: 890 -> 934 patients/s)
    return o.split_wgs > 0 ? o.split_wgs : 256;      // (MmsDnOpts.split_wgs: tuning / test override)
}
static int conv3_nsplit(int M, int ng, long cap_rows, const Dims3& g, const MmsDnOpts& o) {
    if (mms_conv3_small_jn(M, ng, g, o)) return 1;       // small grids: the all-tap kernels of dn_c3s.hip (no tap split, no reduce launch)
    const long tiles = (long)((M + 31) / 32) * ng;
    if (tiles >= 256 && tiles >= split_target(o)) return 1;
    long ns = (split_target(o) + tiles - 1) / tiles;
    if (ns > 27) ns = 27;
    if (ns < 1) ns = 1;
    int tpw = (int)((27 + ns - 1) / ns);
    while (tpw < 27 && (long)((27 + tpw - 1) / tpw) * M > cap_rows) ++tpw;   // the partials must fit the scratch
    return (27 + tpw - 1) / tpw;                                         // every workgroup owns >= 1 tap
}

// Which dense layers get their conv2 weights packed in MFMA-fragment order (Conv3FwdP.wfrag): those of blocks 1-3 whose launches go to the
// small-grid kernels of dn_c3s.hip.  Block 4 keeps the classic packs (its persistent kernels, dn_cl.hip, read those; its per-layer fallback
// path runs the small-grid kernels on them).  A function of the plan and MmsDnOpts.conv3_small only, so forward and backward agree.
// Whether dense block b runs as ONE launch per pass on the cluster kernels (dn_cl.hip): pass 0 = forward, 1 = backward.
// Block 4 -- MmsDnOpts.persist_b4: 0 = both passes, 1 = forward only, -1 = per-layer launches.  Block 3 -- persist_b3: 1 = forward
// (opt-in: measured SLOWER than its per-layer launches at 32 voxels per sample, 522 vs 446 us per model, profiles/r04_cluster_kernels.txt),
// 0 / -1 = per-layer launches.  (The caller decides: the clusters' workgroups must be co-resident.)
static bool cluster_block(const Plan& P, int b, const MmsDnOpts& o, int pass) {
    if (b < 2 || !P.cl_rt[b]) return false;
    if (b == 2) return pass == 0 && o.persist_b3 > 0;
    if (pass == 1 && P.M[3] > 16) return false;          // (backward: a single 16-row cluster only)
    return pass == 0 ? o.persist_b4 >= 0 : o.persist_b4 == 0;
}
static bool conv3_frag_block(const Plan& P, int b, int ng, const MmsDnOpts& o) {
    // (the cluster kernels read the classic forward pack: with packed primary storage that is the parameter itself, always there)
    return b < NB - 1 && mms_conv3_small_jn(P.M[b], ng, P.g[b], o) != 0 && (o.w2_packed || !cluster_block(P, b, o, 0));
}
static uint64_t conv3_fragmask(const Plan& P, int ng, const MmsDnOpts& o) {
    uint64_t m = 0;
    int l = 0;
    for (int b = 0; b < NB; ++b)
        for (int i = 0; i < LAYERS[b]; ++i, ++l) if (conv3_frag_block(P, b, ng, o)) m |= 1ull << l;
    return m;
}

extern "C" int mms_dn121_w2_fragmask(int B, int D, int H, int W, const MmsDnOpts* opts, uint64_t* mask) {
    Plan P;
    if (!make_plan(P, B, D, H, W) || !mask) return MMS_ERR_ARG;
    *mask = conv3_fragmask(P, 1, mms_opts(opts));
    return MMS_OK;
}

// One model of a fold group as the drivers see it.
struct Ctx {
    void* ws; const float* x; const float* const* prm; const void* const* buf; float* out;      // forward
    const float* dout; float* const* grd;                                                        // backward
};
// Data-parallel extras of the single-model drivers (mms_dn121_*_sync / _stage): BatchNorm statistics over bn_world ranks
// (hook = the caller's all-reduce of freshly written accumulator words) and the dense-block range [b_lo, b_hi] of a backward stage.
struct Dp {
    int bn_world = 1; mms_sync_fn hook = nullptr; void* user = nullptr;
    int b_hi = NB - 1, b_lo = 0;
};
#define SYNC(ptr, nrep, rstride, ncols, pstride) do { if (dp.hook) { int rc_ = dp.hook(dp.user, (ptr), (nrep), (long)(rstride), (ncols), (long)(pstride), s); \
    if (rc_ != MMS_OK) { fprintf(stderr, "mmsurv: statistics all-reduce hook failed (dn_net.hip:%d)\n", __LINE__); return MMS_ERR_LAUNCH; } } } while (0)
#define FOR_G for (int g = 0; g < ng; ++g)

// Forward of ng models of identical shape in lock-step: every launch below carries all ng parameter blocks.
static int dn121_forward_impl(const Ctx* cx, int ng, int B, int D, int H, int W, int ldo, int train, const MmsDnOpts* opts, hipStream_t s, const Dp& dp = Dp()) {
    const MmsDnOpts o = mms_opts(opts);
    const int nout = out_features_of(o);
    Plan P;
    if (!make_plan(P, B, D, H, W) || ng < 1 || ng > MMS_MAX_GROUP || nout > 4096 || ldo < nout) return MMS_ERR_ARG;
    if ((dp.hook || dp.bn_world > 1) && (ng != 1 || !train)) return MMS_ERR_ARG;
    const int bw = dp.bn_world;        // BatchNorm statistics are taken over bw * M rows (SyncBN: the hook has summed them over the ranks)
    FOR_G if (!cx[g].ws || !cx[g].x || !cx[g].prm || !cx[g].out) return MMS_ERR_ARG;
    const void* tabs[MMS_MAX_GROUP];
    const bool packed = o.w2_packed != 0;
    if (train) {
        void* regs[MMS_MAX_GROUP];
        const size_t zb = packed ? P.stats_begin_packed : P.stats_begin;
        FOR_G regs[g] = at<void>(cx[g].ws, zb);
        TRY(mms_zero_regions_group(regs, ng, P.stats_end - zb, s));
    }
    if (!packed) {        // torch-layout weights: both packs are rebuilt from them every forward
        FOR_G tabs[g] = at<void>(cx[g].ws, P.tab_pack);
        TRY(mms_pack_conv3_table_group_ex(tabs, ng, NLAYER, conv3_fragmask(P, ng, o), s));
    }
    auto st = [&](void* ws, size_t off, int Ctot_, int coff, bool sq) -> double* {
        return train ? at<double>(ws, off) + (sq ? Ctot_ : 0) + coff : nullptr;
    };
    {   // stem
        Conv0FwdP c0[MMS_MAX_GROUP];
        PoolFwdP pf[MMS_MAX_GROUP];
        FOR_G {
            const Ctx& c = cx[g];
            c0[g] = Conv0FwdP{c.x, P.in, P.g0, at<int>(c.ws, P.coords0), P.M0, c.prm[IDX.conv0], at<float>(c.ws, P.y0),
                              st(c.ws, P.st_y0, 64, 0, false), st(c.ws, P.st_y0, 64, 0, true)};
            c0[g].srep = P.R0; c0[g].sstride = 2 * 64;
            pf[g] = PoolFwdP{at<float>(c.ws, P.y0), P.g0, P.g[0], B, at<float>(c.ws, P.slab[0]), CTOT[0], at<uint8_t>(c.ws, P.argmax),
                             mk_bn(c.ws, P.st_y0, 64, c.prm, IDX.n0w, c.buf, IDX.bn0, P.M0 * bw, train, P.R0),
                             st(c.ws, P.st_slab[0], CTOT[0], 0, false), st(c.ws, P.st_slab[0], CTOT[0], 0, true)};
            pf[g].srep = P.R[0]; pf[g].sstride = 2 * CTOT[0];
        }
        TRYS(28, mms_conv0_fwd_group(c0, ng, &o, s));
        SYNC(at<double>(cx[0].ws, P.st_y0), P.R0, 2 * 64, 64, 64);
        TRYS(28, mms_pool_fwd_group(pf, ng, s));
        SYNC(at<double>(cx[0].ws, P.st_slab[0]), P.R[0], 2 * CTOT[0], 64, CTOT[0]);
    }
    // dense blocks with <= 32 voxels per sample as ONE launch (dn_cl.hip): block 4 (and block 3) of 64x64x32 volumes
    const bool cl_ok = !dp.hook && dp.bn_world == 1;
    if (!train && cl_ok && (cluster_block(P, 2, o, 0) || cluster_block(P, 3, o, 0))) {      // granules + counters (training: the statistics zero-fill above covers them)
        void* regs[MMS_MAX_GROUP];
        FOR_G regs[g] = at<void>(cx[g].ws, P.cl_zero_begin);
        TRY(mms_zero_regions_group(regs, ng, P.stats_end - P.cl_zero_begin, s));
    }
    int l = 0;
    for (int b = 0; b < NB; ++b) {
        int C = C0[b];
        if (cl_ok && cluster_block(P, b, o, 0)) {
            ClFwdP q[MMS_MAX_GROUP];
            FOR_G {
                const Ctx& c = cx[g];
                q[g] = ClFwdP{at<B4Layer>(c.ws, P.cl_tab[b]), LAYERS[b], C0[b], at<float>(c.ws, P.slab[b]), CTOT[b], at<double>(c.ws, P.st_slab[b]),
                              at<int>(c.ws, P.coords[b]), P.g[b], P.M[b], P.cl_rpc[b], P.cl_ncl[b], train, 1e-5f,
                              at<unsigned long long>(c.ws, P.cl_xa[b]), at<unsigned long long>(c.ws, P.cl_xb[b]), at<unsigned long long>(c.ws, P.cl_gst[b]),
                              at<unsigned>(c.ws, P.b4_err)};
            }
            TRYS(32, mms_cl_fwd_group(q, ng, s));
            l += LAYERS[b];
            C += 32 * LAYERS[b];
        } else
        for (int i = 0; i < LAYERS[b]; ++i, ++l, C += 32) {
            const int ip = IDX.layer[l];
            Conv1FwdP c1[MMS_MAX_GROUP];
            Conv3FwdP c3[MMS_MAX_GROUP];
            const int ns3 = conv3_nsplit(P.M[b], ng, P.partial_rows, P.g[b], o);
            // conv1 at small M is a chain of dependent K-steps on a handful of workgroups: one K-step per workgroup instead
            int ks1 = 1;
            if (o.conv1_ksplit >= 0 && (long)((P.M[b] + 31) / 32) * 4 * ng <= 128 && C >= 256 && (long)((C + 127) / 128) * P.M[b] <= P.partial_rows)
                ks1 = (C + 127) / 128;
            FOR_G {
                const Ctx& c = cx[g];
                float* slab = at<float>(c.ws, P.slab[b]);
                c1[g] = Conv1FwdP{slab, CTOT[b], P.M[b], C, c.prm[ip + 2], 128, at<float>(c.ws, P.y1[l]), 128,
                                  mk_bn(c.ws, P.st_slab[b], CTOT[b], c.prm, ip, c.buf, IDX.bn_layer1[l], P.M[b] * bw, train, P.R[b]),
                                  st(c.ws, P.st_y1[l], 128, 0, false), st(c.ws, P.st_y1[l], 128, 0, true), 0, Dims3{0, 0, 0}};
                c1[g].srep = P.R[b]; c1[g].sstride = 2 * 128;
                if (ks1 > 1) { c1[g].partial = at<float>(c.ws, P.partial); c1[g].ksplit = ks1; c1[g].counters = at<unsigned>(c.ws, P.counters); }
                const bool frag3 = conv3_frag_block(P, b, ng, o);
                const float* wp3 = !packed ? at<float>(c.ws, P.wpf[l]) : (frag3 ? c.prm[NPARAM + 2 * l + 1] : c.prm[ip + 5]);
                c3[g] = Conv3FwdP{at<float>(c.ws, P.y1[l]), at<int>(c.ws, P.coords[b]), P.g[b], P.M[b], wp3,
                                  slab + C, CTOT[b], mk_bn(c.ws, P.st_y1[l], 128, c.prm, ip + 3, c.buf, IDX.bn_layer2[l], P.M[b] * bw, train, P.R[b]),
                                  st(c.ws, P.st_slab[b], CTOT[b], C, false), st(c.ws, P.st_slab[b], CTOT[b], C, true),
                                  ns3 > 1 ? at<float>(c.ws, P.partial) : nullptr, ns3};
                c3[g].srep = P.R[b]; c3[g].sstride = 2 * CTOT[b]; c3[g].wfrag = conv3_frag_block(P, b, ng, o) ? 1 : 0;
            }
            TRYS(b, mms_conv1_fwd_group(c1, ng, &o, s));
            SYNC(at<double>(cx[0].ws, P.st_y1[l]), P.R[b], 2 * 128, 128, 128);
            TRYS(4 + b, mms_conv3_fwd_group(c3, ng, &o, s));
            SYNC(at<double>(cx[0].ws, P.st_slab[b]) + C, P.R[b], 2 * CTOT[b], 32, CTOT[b]);
        }
        if (b < 3) {
            const int ip = IDX.trans[b];
            Conv1FwdP t[MMS_MAX_GROUP];
            PoolActP pa[MMS_MAX_GROUP];
            const bool pre = o.trans_prepass >= 0;       // norm / relu / pool by their own launch; the convolution reads the pooled rows
            FOR_G {
                const Ctx& c = cx[g];
                const BnSrc bnt = mk_bn(c.ws, P.st_slab[b], CTOT[b], c.prm, ip, c.buf, IDX.bn_trans[b], P.M[b] * bw, train, P.R[b]);
                pa[g] = PoolActP{at<float>(c.ws, P.slab[b]), CTOT[b], CTOT[b], bnt, P.g[b], P.M[b + 1], at<float>(c.ws, P.tpool[b]), CTOT[b]};
                t[g] = Conv1FwdP{pre ? at<float>(c.ws, P.tpool[b]) : at<float>(c.ws, P.slab[b]), CTOT[b], P.M[b + 1], CTOT[b], c.prm[ip + 2], CTOT[b] / 2,
                                 at<float>(c.ws, P.slab[b + 1]), CTOT[b + 1], pre ? BnSrc{} : bnt,
                                 st(c.ws, P.st_slab[b + 1], CTOT[b + 1], 0, false), st(c.ws, P.st_slab[b + 1], CTOT[b + 1], 0, true), pre ? 0 : 1, pre ? Dims3{0, 0, 0} : P.g[b]};
                t[g].srep = P.R[b + 1]; t[g].sstride = 2 * CTOT[b + 1];
            }
            if (pre) TRYS(30, mms_pool_act_group(pa, ng, s));
            TRYS(30, mms_conv1_fwd_group(t, ng, &o, s));
            SYNC(at<double>(cx[0].ws, P.st_slab[b + 1]), P.R[b + 1], 2 * CTOT[b + 1], CTOT[b] / 2, CTOT[b + 1]);
        }
    }
    HeadFwdP hd[MMS_MAX_GROUP];
    FOR_G {
        const Ctx& c = cx[g];
        hd[g] = HeadFwdP{at<float>(c.ws, P.slab[3]), CTOT[3], 1024, B, P.M[3] / B,
                         mk_bn(c.ws, P.st_slab[3], CTOT[3], c.prm, IDX.n5w, c.buf, IDX.bn5, P.M[3] * bw, train, P.R[3]),
                         c.prm[IDX.outw], c.prm[IDX.outb], nout, at<float>(c.ws, P.pooled), c.out, ldo};
    }
    TRY(mms_head_fwd_group(hd, ng, s));
    if (train) {
        bool all = true;
        FOR_G { all = all && cx[g].buf; tabs[g] = at<void>(cx[g].ws, P.tab_bn); }
        if (all) TRY(mms_bn_running_update_group(tabs, ng, NBN, 0.1f, s));
    }
    return MMS_OK;
}

// Backward of the training-mode forward that last ran on these workspaces.  grads are ACCUMULATED into
// (caller zeroes them, e.g. one hipMemsetAsync over a flat gradient buffer).  dout: [B][128] per model.
static int dn121_backward_impl(const Ctx* cx, int ng, int B, int D, int H, int W, int lddout, const MmsDnOpts* opts, hipStream_t s, hipStream_t side,
                               hipEvent_t ev_fork, hipEvent_t ev_join, const Dp& dp = Dp()) {
    const MmsDnOpts o = mms_opts(opts);
    const int nout = out_features_of(o);
    if (nout > 4096 || lddout < nout) return MMS_ERR_ARG;
    hipStream_t sw = side ? side : s;       // stream of the weight-gradient kernels
    bool side_pending = false;
    Plan P;
    if (!make_plan(P, B, D, H, W) || ng < 1 || ng > MMS_MAX_GROUP) return MMS_ERR_ARG;
    FOR_G if (!cx[g].ws || !cx[g].x || !cx[g].prm || !cx[g].dout || !cx[g].grd) return MMS_ERR_ARG;
    if ((dp.hook || dp.bn_world > 1) && ng != 1) return MMS_ERR_ARG;
    if (dp.b_hi < dp.b_lo || dp.b_hi >= NB || dp.b_lo < 0) return MMS_ERR_ARG;
    const int bnw = dp.bn_world;
    const bool sync = dp.hook != nullptr || bnw > 1;
    const bool packed = o.w2_packed != 0;
    auto bbsrc = [&](void* ws, size_t off, int stride, int nrep) { return BnBwd{at<double>(ws, off), at<double>(ws, off) + stride, nrep, 2 * stride}; };
    if (dp.b_hi == NB - 1) {
        HeadBwdP hb[MMS_MAX_GROUP];
        FOR_G {
            const Ctx& c = cx[g];
            hb[g] = HeadBwdP{c.dout, lddout, at<float>(c.ws, P.pooled), at<float>(c.ws, P.slab[3]), CTOT[3], 1024, B, P.M[3] / B,
                             mk_bn(c.ws, P.st_slab[3], CTOT[3], c.prm, IDX.n5w, nullptr, 0, P.M[3] * bnw, 1, P.R[3]), c.prm[IDX.outw], nout,
                             c.grd[IDX.outw], c.grd[IDX.outb], c.grd[IDX.n5w], c.grd[IDX.n5b], at<float>(c.ws, P.dslab[3]), CTOT[3]};
        }
        if (sync) {     // norm5's backward sums must span all ranks: sums kernel | all-reduce | apply kernel
            hb[0].ext_sums = at<double>(cx[0].ws, P.bb_head);
            TRY(mms_head_bwd_sums(hb, s));
            SYNC(at<double>(cx[0].ws, P.bb_head), 1, 2 * 1024, 1024, 1024);
            TRY(mms_head_bwd_apply(hb, s));
        } else {
            TRY(mms_head_bwd_group(hb, ng, s));
        }
    }
    // Weight gradients are off the backward's critical chain (nothing reads them before the optimiser).  In blocks 2-4 a
    // layer's two weight-gradient launches are far too small to fill the chip (27-216 workgroups per model), so they are
    // deferred and issued batched over layers -- as many (model, layer) members per launch as the group entry points carry --
    // once the block's chain is through: dz_l = dslab[:, C_l:C_l+32] is final from the moment layer l has been processed (earlier
    // layers only add into columns < C_l), y1 / the statistic accumulators are per layer, and dbn_mid is per layer there.
    // Round 4: block 1 too.  Its launches do fill the chip, but at 1-2 models per launch badly (weight gradient 32 / 37 % of the MFMA
    // peak at 1 / 2 models, 48-53 % at 5-10): its six layers are issued as launches of 6 (one model), 3 + 3 layers x 2 (two models), ...
    // members.  MmsDnOpts.batch_w = -1 restores one launch pair per layer, 1 = blocks 2-4 only (rounds 2-3).
    const bool batch_w = o.batch_w >= 0 && !side;      // (fine under SyncBN too: the sums the weight kernels read are all-reduced by then)
    Conv3BwdWP bwq[MMS_MAX_GROUP];
    Conv1BwdP c1q[MMS_MAX_GROUP];
    int nq = 0;
    int nq_limit = MMS_MAX_GROUP;
    auto flush_w = [&](int b) -> int {
        if (nq == 0) return MMS_OK;
        TRYS(12 + b, mms_conv3_bwd_weight_group(bwq, nq, &o, s));
        TRYS(16 + b, mms_conv1_bwd_weight_group(c1q, nq, s));
        nq = 0;
        return MMS_OK;
    };
    int l = NLAYER;
    for (int b = NB - 1; b > dp.b_hi; --b) l -= LAYERS[b];
    for (int b = dp.b_hi; b >= dp.b_lo; --b) {
        int C = CTOT[b];
        const int M = P.M[b];
        const bool defer = batch_w && (b > 0 || o.batch_w == 0) && 2 * ng <= MMS_MAX_GROUP;
        // (model, layer) members per weight-gradient launch: blocks 2-4 fill the launches greedily; block 1 -- where a launch is large
        // and its kernel form depends on its size -- splits its layers evenly over the fewest launches
        int ngw_blk = ng;
        if (defer) {
            if (b > 0) ngw_blk = MMS_MAX_GROUP / ng * ng;
            else { const int nl = (LAYERS[b] * ng + MMS_MAX_GROUP - 1) / MMS_MAX_GROUP; ngw_blk = (LAYERS[b] + nl - 1) / nl * ng; }
        }
        nq_limit = ngw_blk;
        // block 4 (<= 32 rows, one MFMA row tile): norm1's backward rides in conv1_bwd_data's epilogue (Conv1BwdP.fuse_dx), no
        // mms_bn_bwd_apply launch.  (Measured at 128 rows -- block 3, 128 x 32 tiles -- the fused form is slower than the two
        // launches it replaces: 25 us against 8.8 + 6.7 us, rocprofv3 kernel stats; MmsDnOpts.fuse_apply_rows = 128 selects it anyway.)
        // Round 3: up to 128 rows the whole-M kernel of dn_c1s.hip (one workgroup per 16 channels, every row) takes the fused form --
        // one launch instead of conv1_bwd_data + bn_bwd_apply, no statistic atomics; MmsDnOpts.conv1_small_bwd = -1 restores the rule above.
        const int fuse_rows = o.fuse_apply_rows > 0 ? o.fuse_apply_rows : (o.conv1_small_bwd < 0 ? 32 : 128);
        const bool fuse_apply = M <= fuse_rows && M <= 128 && !sync;      // SyncBN: the sums leave the workgroup (all-reduce) before they are applied
        // block 4 with <= 16 rows: the whole data path of the block's backward as ONE launch (dn_cl.hip); the loop below then only queues
        // the layers' weight-gradient members.  MmsDnOpts.persist_b4: -1 = off (both passes), 1 = forward only; default both.
        const bool b4_bwd = b == 3 && !sync && dp.bn_world == 1 && defer && fuse_apply && cluster_block(P, 3, o, 1);
        if (b4_bwd) {
            ClBwdP q[MMS_MAX_GROUP];
            FOR_G {
                const Ctx& c = cx[g];
                q[g] = ClBwdP{at<B4Layer>(c.ws, P.cl_tab[3]), LAYERS[3], C0[3], at<float>(c.ws, P.slab[3]), at<float>(c.ws, P.dslab[3]), CTOT[3],
                              at<double>(c.ws, P.st_slab[3]), at<int>(c.ws, P.coords[3]), P.g[3], M, 1e-5f,
                              at<unsigned long long>(c.ws, P.cl_ga), at<unsigned long long>(c.ws, P.cl_gz), at<unsigned>(c.ws, P.b4_err), {}, {}};
                for (int i = 0; i < LAYERS[3]; ++i) {
                    const int ip = IDX.layer[NLAYER - LAYERS[3] + i];
                    q[g].dg1[i] = (float*)c.grd[ip]; q[g].db1[i] = (float*)c.grd[ip + 1];
                }
            }
            TRYS(33, mms_cl_bwd_group(q, ng, s));
        }
        for (int i = LAYERS[b] - 1; i >= 0; --i) {
            --l; C -= 32;
            const int ip = IDX.layer[l];
            Conv3BwdDataP bd[MMS_MAX_GROUP];
            Conv3BwdWP bw[MMS_MAX_GROUP];
            Conv1BwdP c1[MMS_MAX_GROUP];
            BnBwdApplyP ap[MMS_MAX_GROUP];
            const int ns3 = conv3_nsplit(M, ng, P.partial_rows, P.g[b], o);
            const int ngw = ngw_blk;                                      // (model, layer) members per weight-gradient launch
            const int ms3 = mms_conv3w_msplit(M, ngw, o);
            int ms1 = M > 1024 ? M / 256 : M / 128; if (ms1 < 1) ms1 = 1; if (ms1 > 32) ms1 = 32;
            { const int dv = o.ms1_div > 0 ? o.ms1_div : (ngw >= 4 ? 2 : 1); if (dv > 1) { ms1 = ms1 / dv; if (ms1 < 1) ms1 = 1; } }   // groups: half the chunks (fewer atomic flushes)
            FOR_G {
                const Ctx& c = cx[g];
                float* slab = at<float>(c.ws, P.slab[b]);
                float* dslab = at<float>(c.ws, P.dslab[b]);
                float* dmid = at<float>(c.ws, P.dbn_mid_l[l]);
                const BnSrc bn1 = mk_bn(c.ws, P.st_slab[b], CTOT[b], c.prm, ip, nullptr, 0, M * bnw, 1, P.R[b]);
                const BnSrc bn2 = mk_bn(c.ws, P.st_y1[l], 128, c.prm, ip + 3, nullptr, 0, M * bnw, 1, P.R[b]);
                bd[g] = Conv3BwdDataP{dslab + C, CTOT[b], at<int>(c.ws, P.coords[b]), P.g[b], M, packed ? c.prm[NPARAM + 2 * l] : at<float>(c.ws, P.wpb[l]),
                                      at<float>(c.ws, P.y1[l]), bn2, dmid,
                                      at<double>(c.ws, P.bb_y1[l]), at<double>(c.ws, P.bb_y1[l]) + 128,
                                      ns3 > 1 ? at<float>(c.ws, P.partial) : nullptr, ns3};
                bd[g].srep = P.R[b]; bd[g].sstride = 2 * 128; bd[g].wfrag = conv3_frag_block(P, b, ng, o) ? 1 : 0;
                bw[g] = Conv3BwdWP{at<float>(c.ws, P.y1[l]), at<int>(c.ws, P.coords[b]), P.g[b], M, bn2, dslab + C, CTOT[b],
                                   packed ? c.grd[ip + 5] : at<float>(c.ws, P.dwp[l]), ms3, packed ? 2 : 1};
                Conv1BwdP& q = c1[g];
                q = Conv1BwdP{};
                q.dyraw = dmid; q.lddy = 128;
                q.y = at<float>(c.ws, P.y1[l]); q.ldy = 128;
                q.bn_out = bn2; q.bb_out = bbsrc(c.ws, P.bb_y1[l], 128, P.R[b]); q.has_bn_out = 1;
                q.M = M; q.N = 128;
                q.x = slab; q.ldx = CTOT[b]; q.K = C; q.bn_in = bn1;
                q.w = c.prm[ip + 2]; q.pool = 0; q.in = Dims3{0, 0, 0};
                q.dw = c.grd[ip + 2];
                q.dbn = at<float>(c.ws, P.dbn_in); q.lddbn = CTOT[b];
                q.s1 = at<double>(c.ws, P.bb_in[l]); q.s2 = at<double>(c.ws, P.bb_in[l]) + 1024;
                q.srep = P.R[b]; q.sstride = 2 * 1024;
                q.msplit = ms1; q.dgamma_out = c.grd[ip + 3]; q.dbeta_out = c.grd[ip + 4];
                if (fuse_apply) { q.fuse_dx = dslab; q.fuse_lddx = CTOT[b]; q.fuse_accumulate = 1; q.fuse_dgamma = c.grd[ip]; q.fuse_dbeta = c.grd[ip + 1]; }
                ap[g] = BnBwdApplyP{at<float>(c.ws, P.dbn_in), CTOT[b], slab, CTOT[b], dslab, CTOT[b], M, C, bn1,
                                    bbsrc(c.ws, P.bb_in[l], 1024, P.R[b]), 1, c.grd[ip], c.grd[ip + 1]};
            }
            if (side && side_pending) {       // the previous layer's weight kernels read dbn_mid: join before overwriting it
                if (hipStreamWaitEvent(s, ev_join, 0) != hipSuccess) return MMS_ERR_LAUNCH;
                side_pending = false;
            }
            if (!b4_bwd) {
                TRYS(8 + b, mms_conv3_bwd_data_group(bd, ng, &o, s));
                SYNC(at<double>(cx[0].ws, P.bb_y1[l]), P.R[b], 2 * 128, 128, 128);
            }
            if (defer) {
                if (nq + ng > nq_limit) TRY(flush_w(b));
                FOR_G { bwq[nq] = bw[g]; c1q[nq] = c1[g]; ++nq; }
            } else {
                if (side) {
                    if (hipEventRecord(ev_fork, s) != hipSuccess || hipStreamWaitEvent(side, ev_fork, 0) != hipSuccess) return MMS_ERR_LAUNCH;
                }
                TRYS(12 + b, mms_conv3_bwd_weight_group(bw, ng, &o, sw));
                TRYS(16 + b, mms_conv1_bwd_weight_group(c1, ng, sw));
                if (side) {
                    if (hipEventRecord(ev_join, side) != hipSuccess) return MMS_ERR_LAUNCH;
                    side_pending = true;
                }
            }
            if (b4_bwd) continue;
            TRYS(20 + b, mms_conv1_bwd_data_group(c1, ng, &o, s));
            SYNC(at<double>(cx[0].ws, P.bb_in[l]), P.R[b], 2 * 1024, C, 1024);
            if (!fuse_apply) TRYS(24 + b, mms_bn_bwd_apply_group(ap, ng, s));
        }
        TRY(flush_w(b));
        if (side && side_pending) {
            if (hipStreamWaitEvent(s, ev_join, 0) != hipSuccess) return MMS_ERR_LAUNCH;
            side_pending = false;
        }
        if (b > 0) {   // transition b-1 -> b
            const int t = b - 1, ip = IDX.trans[t], Kp = CTOT[t], Mp = P.M[t];
            int ms1 = M / 256; if (ms1 < 1) ms1 = 1; if (ms1 > 32) ms1 = 32;
            Conv1BwdP c1[MMS_MAX_GROUP];
            BnBwdApplyP ap[MMS_MAX_GROUP];
            FOR_G {
                const Ctx& c = cx[g];
                const BnSrc bnt = mk_bn(c.ws, P.st_slab[t], CTOT[t], c.prm, ip, nullptr, 0, Mp * bnw, 1, P.R[t]);
                Conv1BwdP& q = c1[g];
                q = Conv1BwdP{};
                q.dyraw = at<float>(c.ws, P.dslab[b]); q.lddy = CTOT[b]; q.y = nullptr; q.ldy = 0; q.has_bn_out = 0;
                q.bn_out = bnt; q.bb_out = BnBwd{nullptr, nullptr, 0, 0};
                q.M = M; q.N = Kp / 2;
                q.x = at<float>(c.ws, P.slab[t]); q.ldx = CTOT[t]; q.K = Kp; q.bn_in = bnt;
                q.w = c.prm[ip + 2]; q.pool = 1; q.in = P.g[t];
                q.dw = c.grd[ip + 2];
                q.dbn = at<float>(c.ws, P.dbn_in); q.lddbn = CTOT[t];
                q.s1 = at<double>(c.ws, P.bb_tr[t]); q.s2 = at<double>(c.ws, P.bb_tr[t]) + 1024;
                q.srep = P.R[t]; q.sstride = 2 * 1024;
                q.msplit = ms1; q.dgamma_out = nullptr; q.dbeta_out = nullptr;
                ap[g] = BnBwdApplyP{at<float>(c.ws, P.dbn_in), CTOT[t], at<float>(c.ws, P.slab[t]), CTOT[t], at<float>(c.ws, P.dslab[t]), CTOT[t],
                                    Mp, Kp, bnt, bbsrc(c.ws, P.bb_tr[t], 1024, P.R[t]), 0, c.grd[ip], c.grd[ip + 1]};
            }
            if (o.trans_prepass >= 0) {       // the weight gradient reads the pooled operand the forward left in the workspace
                Conv1BwdP cw[MMS_MAX_GROUP];
                FOR_G { cw[g] = c1[g]; cw[g].x = at<float>(cx[g].ws, P.tpool[t]); cw[g].pool = 0; cw[g].in = Dims3{0, 0, 0}; cw[g].bn_in = BnSrc{}; }
                TRYS(31, mms_conv1_bwd_weight_group(cw, ng, s));
            } else TRYS(31, mms_conv1_bwd_weight_group(c1, ng, s));
            TRYS(31, mms_conv1_bwd_data_group(c1, ng, &o, s));
            SYNC(at<double>(cx[0].ws, P.bb_tr[t]), P.R[t], 2 * 1024, Kp, 1024);
            TRYS(31, mms_bn_bwd_apply_group(ap, ng, s));
        } else {       // stem
            int ms0 = P.M0 / 1024; if (ms0 < 1) ms0 = 1; if (ms0 > 64) ms0 = 64;
            PoolBwdP pb[MMS_MAX_GROUP];
            Conv0BwdWP cw[MMS_MAX_GROUP];
            FOR_G {
                const Ctx& c = cx[g];
                const BnSrc bn0 = mk_bn(c.ws, P.st_y0, 64, c.prm, IDX.n0w, nullptr, 0, P.M0 * bnw, 1, P.R0);
                pb[g] = PoolBwdP{at<float>(c.ws, P.dslab[0]), CTOT[0], at<uint8_t>(c.ws, P.argmax), P.g[0], P.g0, B, at<float>(c.ws, P.y0), bn0,
                                 at<float>(c.ws, P.dbn0), at<double>(c.ws, P.bb_y0), at<double>(c.ws, P.bb_y0) + 64, at<int>(c.ws, P.coords0)};
                pb[g].srep = P.R0; pb[g].sstride = 2 * 64;
                cw[g] = Conv0BwdWP{at<float>(c.ws, P.dbn0), at<float>(c.ws, P.y0), bn0, bbsrc(c.ws, P.bb_y0, 64, P.R0), c.x, P.in, P.g0,
                                   at<int>(c.ws, P.coords0), P.M0, c.grd[IDX.conv0], ms0, c.grd[IDX.n0w], c.grd[IDX.n0b],
                                   at<float>(c.ws, P.dw0_rep), 8};
            }
            TRYS(29, mms_pool_bwd_group(pb, ng, s));
            SYNC(at<double>(cx[0].ws, P.bb_y0), P.R0, 2 * 64, 64, 64);
            TRYS(29, mms_conv0_bwd_weight_group(cw, ng, &o, s));
        }
    }
    if (side && side_pending) {
        if (hipStreamWaitEvent(s, ev_join, 0) != hipSuccess) return MMS_ERR_LAUNCH;
    }
    {   // tap-major scratch -> canonical conv2 gradients of the layers this call processed: one launch for the group
        // (the dwp regions are consecutive takes)
        static_assert(NLAYER == 58, "UnpackGroup is sized for DenseNet121");
        int l0 = 0, nl = 0;
        for (int b = 0; b < NB; ++b) { if (b < dp.b_lo) l0 += LAYERS[b]; else if (b <= dp.b_hi) nl += LAYERS[b]; }
        if (nl == 0 || packed) return MMS_OK;        // (packed primary storage: the weight-gradient kernels wrote the gradient in place)
        const float* scr[MMS_MAX_GROUP];
        float* dwt[MMS_MAX_GROUP][NLAYER];
        float* const* dwp_[MMS_MAX_GROUP];
        if (P.dwp[1] - P.dwp[0] != (size_t)27 * 32 * 128 * 4) return MMS_ERR_ARG;
        FOR_G {
            scr[g] = at<float>(cx[g].ws, P.dwp[l0]);
            for (int i = 0; i < nl; ++i) dwt[g][i] = cx[g].grd[IDX.layer[l0 + i] + 5];
            dwp_[g] = dwt[g];
        }
        TRY(mms_unpack_conv3_grads_group(scr, dwp_, ng, nl, s));
    }
    return MMS_OK;
}

extern "C" int mms_dn121_forward(void* ws, int B, int D, int H, int W, const float* x, const void* const* params_,
                                 const void* const* buffers, float* out, int ldo, int train, const MmsDnOpts* opts, hipStream_t s) {
    Ctx c{ws, x, (const float* const*)params_, buffers, out, nullptr, nullptr};
    return dn121_forward_impl(&c, 1, B, D, H, W, ldo, train, opts, s);
}
extern "C" int mms_dn121_backward(void* ws, int B, int D, int H, int W, const float* x, const void* const* params,
                                  const float* dout, int lddout, void* const* grads, const MmsDnOpts* opts, hipStream_t s) {
    Ctx c{ws, x, (const float* const*)params, nullptr, nullptr, dout, (float* const*)grads};
    return dn121_backward_impl(&c, 1, B, D, H, W, lddout, opts, s, nullptr, nullptr, nullptr);
}
extern "C" int mms_dn121_backward_mt(void* ws, int B, int D, int H, int W, const float* x, const void* const* params,
                                     const float* dout, int lddout, void* const* grads, const MmsDnOpts* opts, hipStream_t s, hipStream_t side,
                                     hipEvent_t ev_fork, hipEvent_t ev_join) {
    if (!side || !ev_fork || !ev_join) return MMS_ERR_ARG;
    Ctx c{ws, x, (const float* const*)params, nullptr, nullptr, dout, (float* const*)grads};
    return dn121_backward_impl(&c, 1, B, D, H, W, lddout, opts, s, side, ev_fork, ev_join);
}

// Data-parallel variants of the single-model drivers (one process per GPU; include/mmsurv.h).
extern "C" int mms_dn121_forward_sync(void* ws, int B, int D, int H, int W, const float* x, const void* const* params_,
                                      const void* const* buffers, float* out, int ldo, int bn_world, mms_sync_fn hook, void* user,
                                      const MmsDnOpts* opts, hipStream_t s) {
    if (bn_world < 1) return MMS_ERR_ARG;
    Ctx c{ws, x, (const float* const*)params_, buffers, out, nullptr, nullptr};
    Dp dp; dp.bn_world = bn_world; dp.hook = hook; dp.user = user;
    return dn121_forward_impl(&c, 1, B, D, H, W, ldo, 1, opts, s, dp);
}
extern "C" int mms_dn121_backward_stage(void* ws, int B, int D, int H, int W, const float* x, const void* const* params,
                                        const float* dout, int lddout, void* const* grads, int block_hi, int block_lo,
                                        int bn_world, mms_sync_fn hook, void* user, const MmsDnOpts* opts, hipStream_t s) {
    if (bn_world < 1) return MMS_ERR_ARG;
    Ctx c{ws, x, (const float* const*)params, nullptr, nullptr, dout, (float* const*)grads};
    Dp dp; dp.bn_world = bn_world; dp.hook = hook; dp.user = user; dp.b_hi = block_hi; dp.b_lo = block_lo;
    return dn121_backward_impl(&c, 1, B, D, H, W, lddout, opts, s, nullptr, nullptr, nullptr, dp);
}

// Fold-group drivers: model g of the group is described by the g-th entry of each array (all models share B, D, H, W).
extern "C" int mms_dn121_forward_group(int ng, void* const* ws, int B, int D, int H, int W, const float* const* x,
                                       const void* const* const* params, const void* const* const* buffers, float* const* out,
                                       int ldo, int train, const MmsDnOpts* opts, hipStream_t s) {
    if (ng < 1 || ng > MMS_MAX_GROUP || !ws || !x || !params || !out) return MMS_ERR_ARG;
    Ctx c[MMS_MAX_GROUP];
    FOR_G c[g] = Ctx{ws[g], x[g], (const float* const*)params[g], buffers ? buffers[g] : nullptr, out[g], nullptr, nullptr};
    return dn121_forward_impl(c, ng, B, D, H, W, ldo, train, opts, s);
}
extern "C" int mms_dn121_backward_group(int ng, void* const* ws, int B, int D, int H, int W, const float* const* x,
                                        const void* const* const* params, const float* const* dout, int lddout,
                                        void* const* const* grads, const MmsDnOpts* opts, hipStream_t s) {
    if (ng < 1 || ng > MMS_MAX_GROUP || !ws || !x || !params || !dout || !grads) return MMS_ERR_ARG;
    Ctx c[MMS_MAX_GROUP];
    FOR_G c[g] = Ctx{ws[g], x[g], (const float* const*)params[g], nullptr, nullptr, dout[g], (float* const*)grads[g]};
    return dn121_backward_impl(c, ng, B, D, H, W, lddout, opts, s, nullptr, nullptr, nullptr);
}
