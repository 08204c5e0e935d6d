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
