// Parameter blocks live in the public C header.
#pragma once
#include "common.h"

// Launch-shape rule shared by the weight-gradient launcher (dn_bwd.hip) and the network driver (dn_net.hip): the multi-tap conv3
// weight-gradient kernel holds 3 workgroups per CU; it only pays when its grid fills >= 90 % of a whole number of such rounds.
static inline bool mms_conv3w_mt_fills(long workgroups) {
    const long slots = 3 * 256, rounds = (workgroups + slots - 1) / slots;
    return workgroups > 0 && workgroups * 10 >= rounds * slots * 9;
}
