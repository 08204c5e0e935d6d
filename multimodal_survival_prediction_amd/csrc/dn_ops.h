// Parameter blocks live in the public C header.
#pragma once
#include "common.h"
#include <stdlib.h>

// Launch-shape rule shared by the weight-gradient launcher (dn_bwd.hip) and the network driver (dn_net.hip): the multi-tap conv3
// weight-gradient kernel holds 3 workgroups per CU; it only pays when its grid fills >= 90 % of a whole number of such rounds.
static inline bool mms_conv3w_mt_fills(long workgroups) {
    const long slots = 3 * 256, rounds = (workgroups + slots - 1) / slots;
    return workgroups > 0 && workgroups * 10 >= rounds * slots * 9;
}

// Row chunks of a conv2 weight-gradient launch with `members` (model, layer) members of M rows each (the msplit the driver passes;
// MmsDnOpts.ms3_rows > 0 fixes the rows per chunk).  Every chunk flushes 27 x 16 KB of fp32 atomics, so launches with >= 4 members
// (which bring their own parallelism) take chunks twice as long -- half the fabric writes (PMC WRITE_SIZE) for the same FLOPs.  Large-M
// launches of >= 4 members look for a chunk count that puts them on the multi-tap kernel with a well-filled grid: among the counts with
// chunks of 512..1024 rows whose 9-per-chunk grid passes mms_conv3w_mt_fills, the one with the least rounds x rows per chunk (ties: fewest
// chunks): 10 members x 8192 rows -> 8 chunks (720 workgroups), 5 -> 16 (720), 6 -> 14 (756), 9 -> 9 (729), 8 -> 10 (720).
static inline int mms_conv3w_msplit(int M, int members, const MmsDnOpts& o) {
    const bool e3 = o.ms3_rows > 0;
    const int rows3 = e3 ? o.ms3_rows : (members >= 4 ? 1024 : 512), rows3s = e3 ? 128 : (members >= 4 ? 256 : 128);
    if (M <= 1024) { const int ms = (M + rows3s - 1) / rows3s; return ms < 1 ? 1 : ms; }
    int best = 0; long best_cost = 0;
    if (!e3 && members >= 4 && o.conv3w_mt >= 0) {
        for (int ms = (M + 1023) / 1024; ms <= M / 512; ++ms) {
            const int chunk = ((M + ms - 1) / ms + 31) & ~31;
            const long wgs = (long)ms * members * 9;
            if (chunk < 512 || chunk > 1024 || !mms_conv3w_mt_fills(wgs)) continue;
            const long cost = (wgs + 767) / 768 * chunk;
            if (!best || cost < best_cost) { best = ms; best_cost = cost; }
        }
    }
    return best ? best : (M + rows3 - 1) / rows3;
}

// Small-grid 3x3x3 convolution kernels (dn_c3s.hip): a workgroup owns 16 voxel rows and all 27 taps, the rows' whole neighbourhood
// [m0 - halo, m0 + 16 + halo), halo = H*W + W + 1, staged in LDS once -- no tap split, no reduce launch.  Applies when that window fits
// (dense blocks 2-4 of 64x64x32 volumes, blocks 3-4 of 128x128x64 volumes).  Returns 0 (not applicable / MMS_CONV3_SMALL=0) or the
// number of 16-column output tiles per wave: 2, or 1 (output channels split over blockIdx.y) when the launch has few row tiles
// (MmsDnOpts.conv3_small = 1 / 2 force that number, -1 = never: tests).
#define MMS_C3S_MAXROWS 120
static inline int mms_conv3_small_jn(int M, int ng, const Dims3& g, const MmsDnOpts& o) {
    if (o.conv3_small < 0 || M <= 0) return 0;
    const long halo = (long)g.H * g.W + g.W + 1;
    if (16 + 2 * halo > MMS_C3S_MAXROWS) return 0;
    if (o.conv3_small == 1 || o.conv3_small == 2) return o.conv3_small;
    const long tiles = (long)((M + 15) / 16) * ng;
    return tiles > 128 ? 2 : 1;
}
// Small-launch 1x1x1 convolution forward (dn_c1s.hip): 16 x 16 output tiles, the whole K range per workgroup, both operand panels in LDS -- no
// K split over workgroups, no ticket.  Taken for un-pooled launches with at most MmsDnOpts.c1s_max_wgs tiles (default 640: dense block 3 of every
// fold group, block 2 of a single model -- measured per launch at 3 models: block 3 (192 tiles) 13.5 -> 8.8 us, block 2 (1536 tiles: three
// rounds of two workgroups per CU) 14.9 -> 20.7 us; one model: block 2 (512 tiles) 12.3 -> 8.9 us); MmsDnOpts.conv1_small = -1 disables it (A/B, the
// K-split tests), 1 forces it whenever the shape allows.
static inline bool mms_conv1_small_ok(const Conv1FwdP& p, int ng, const MmsDnOpts& o) {
    if (o.conv1_small < 0 || p.pool || p.K % 32 != 0 || p.K > 1024 || p.K < 32 || p.ldx % 4 != 0) return false;
    if (o.conv1_small > 0) return true;
    const long maxwg = o.c1s_max_wgs > 0 ? o.c1s_max_wgs : 640;
    return (long)((p.M + 15) / 16) * ((p.N + 15) / 16) * ng <= maxwg;
}
int mms_c1s_fwd(const Conv1FwdP* pp, int ng, hipStream_t s);
// whole-M backward-data + fused norm1 backward (dn_c1s.hip): MmsDnOpts.conv1_small_bwd = -1 keeps the tile-GEMM forms
bool mms_conv1_small_bwd_ok(const Conv1BwdP& p, const MmsDnOpts& o);
int mms_c1s_bwd(const Conv1BwdP* pp, int ng, hipStream_t s);
int mms_c3s_fwd(const Conv3FwdP* pp, int ng, const MmsDnOpts& o, hipStream_t s);
int mms_c3s_bwd_data(const Conv3BwdDataP* pp, int ng, const MmsDnOpts& o, hipStream_t s);

// ---- dense blocks 3 / 4 as one launch per pass (dn_cl.hip, dn_b4.hip); internal to the network drivers ---------------------------
struct B4Layer {               // device table entry, one per dense layer (built by mms_dn121_init)
    const float *g1, *b1, *w1;                 // norm1 gamma / beta [C], conv1 weight [128][C]
    const float *g2, *b2, *wpf, *wpb;          // norm2 gamma / beta [128], packed conv2 weights [32][27][128] / [128][27][32]
    const float *rm1, *rv1, *rm2, *rv2;        // running statistics (eval-mode forward)
    float* y1; double* st_y1;                  // pre-BatchNorm2 activations [M][128] and their (sum | sumsq) [2][128], saved for the backward
    float* dmid; double* bb_y1;                // backward: masked gradient at norm2's output [M][128] and its BatchNorm-backward sums (s1 | s2) [2][128]
};
// Forward of a dense block with <= 32 voxels per sample as one launch (dn_cl.hip): clusters of 8 workgroups, each owning whole samples.
struct ClFwdP {
    const B4Layer* tab; int nlayers; int C0;   // the block's layers, its first layer's input channels
    float* slab; int ld;                       // [M][ld = 1024]: columns [0, C0) in, the rest out
    double* st_slab;                           // (sum | sumsq) [2][ld], one replica (train)
    const int* coords; Dims3 g; int M;         // M = rows of the whole batch (the BatchNorm statistics span them)
    int rpc; int ncl;                          // rows per cluster (whole samples; <= 16: one MFMA row tile, <= 32: two), clusters (<= 8)
    int train; float eps;
    unsigned long long* xa; unsigned long long* xb; unsigned long long* gst;     // {tag, value} granule buffers, ZERO on entry:
                                               // [ncl][8][rt * 256], [ncl][8][rt * 512], [8][ncl][64] + [8][ncl][128] (rt = row tiles)
    unsigned* err;                             // sticky time-out flag
};
extern "C" int mms_cl_fwd_group(const ClFwdP* pp, int ng, hipStream_t s);
struct ClBwdP {                // the data path of a single-cluster block's backward (dslab -> norm2/conv2 -> norm1/conv1 -> dslab, last layer .. first) as one launch
    const B4Layer* tab; int nlayers; int C0;
    const float* slab; float* dslab; int ld;   // saved activations / their gradient [M][ld = 1024]; on entry dslab holds d(loss)/d(slab) from norm5,
                                               // on exit columns [0, C0) are the block input's gradient and [C_l, C_l + 32) layer l's final dz
    const double* st_slab;                     // (sum | sumsq) [2][ld] of the slab channels, one replica
    const int* coords; Dims3 g; int M;         // M <= 16 rows
    float eps;
    unsigned long long* ga; unsigned long long* gz;      // {tag, value} granule buffers, ZERO on entry: [2][8][256] (hand-off A, by layer parity), [512] (dz broadcast)
    unsigned* err;                             // sticky time-out flag
    float* dg1[16]; float* db1[16];            // norm1 gamma / beta gradients of the layers (accumulated into)
};
extern "C" int mms_cl_bwd_group(const ClBwdP* pp, int ng, hipStream_t s);
