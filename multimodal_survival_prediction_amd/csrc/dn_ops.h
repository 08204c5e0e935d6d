// Parameter blocks live in the public C header.
#pragma once
#include "common.h"

// Launch-shape rule shared by the weight-gradient launcher (dn_bwd.hip) and the network driver (dn_net.hip): the multi-tap conv3
// weight-gradient kernel holds 3 workgroups per CU; it only pays when its grid fills >= 90 % of a whole number of such rounds.
static inline bool mms_conv3w_mt_fills(long workgroups) {
    const long slots = 3 * 256, rounds = (workgroups + slots - 1) / slots;
    return workgroups > 0 && workgroups * 10 >= rounds * slots * 9;
}

// ---- dense block 4 as one launch per pass (dn_b4.hip); internal to the network drivers ------------------------------------------
struct B4Layer {               // device table entry, one per dense layer of block 4 (built by mms_dn121_init)
    const float *g1, *b1, *w1;                 // norm1 gamma / beta [C], conv1 weight [128][C]
    const float *g2, *b2, *wpf, *wpb;          // norm2 gamma / beta [128], packed conv2 weights [32][27][128] / [128][27][32]
    const float *rm1, *rv1, *rm2, *rv2;        // running statistics (eval-mode forward)
    float* y1; double* st_y1;                  // pre-BatchNorm2 activations [M][128] and their (sum | sumsq) [2][128], saved for the backward
    float* dmid; double* bb_y1;                // backward: masked gradient at norm2's output [M][128] and its BatchNorm-backward sums (s1 | s2) [2][128]
};
struct B4FwdP {
    const B4Layer* tab; int nlayers; int C0;   // 16 layers, 512 input channels
    float* slab; int ld;                       // [M][ld = 1024]: columns [0, C0) in, the rest out
    double* st_slab;                           // (sum | sumsq) [2][ld], one replica (train)
    const int* coords; Dims3 g; int M;         // M <= 16 rows
    int train; float eps;
    float* xa; float* xb;                      // hand-off buffers [8][256], [8][512]
    unsigned* counter; unsigned* err;          // counter: zero on entry; err: sticky time-out flag
};
extern "C" int mms_b4_fwd_group(const B4FwdP* pp, int ng, hipStream_t s);
struct B4BwdP {                // the data path of block 4's backward (dslab -> norm2/conv2 -> norm1/conv1 -> dslab, layer 15 .. 0) as one launch
    const B4Layer* tab; int nlayers; int C0;
    const float* slab; float* dslab; int ld;   // saved activations / their gradient [M][ld = 1024]; on entry dslab holds d(loss)/d(slab) from norm5,
                                               // on exit columns [0, C0) are the block input's gradient and [C_l, C_l + 32) layer l's final dz
    const double* st_slab;                     // (sum | sumsq) [2][ld] of the slab channels, one replica
    const int* coords; Dims3 g; int M;         // M <= 16 rows
    float eps;
    float* xa;                                 // hand-off buffer [8][256]
    unsigned* counter; unsigned* err;          // counter: zero on entry (its own word); err: sticky time-out flag
    float* dg1[16]; float* db1[16];            // norm1 gamma / beta gradients of the layers (accumulated into)
};
extern "C" int mms_b4_bwd_group(const B4BwdP* pp, int ng, hipStream_t s);
