// ABI self-description: lets the host check that its view of include/mmsurv.h matches the compiled library.
#include "common.h"
#include <string.h>

#define SZ(T) if (strcmp(name, #T) == 0) return (int)sizeof(T);
extern "C" int mms_abi_sizeof(const char* name) {
    SZ(Dims3) SZ(MmsDnOpts) SZ(BnSrc) SZ(BnBwd) SZ(Conv1FwdP) SZ(Conv3FwdP) SZ(Conv0FwdP) SZ(PoolFwdP) SZ(PoolActP) SZ(HeadFwdP)
    SZ(Conv3BwdDataP) SZ(Conv3BwdWP) SZ(Conv1BwdP) SZ(BnBwdApplyP) SZ(HeadBwdP) SZ(PoolBwdP) SZ(Conv0BwdWP) SZ(InProlog) SZ(LinearFwdP) SZ(LinearBwdP) SZ(GateP) SZ(CoxP) SZ(CindexP) SZ(AdamP) SZ(FbConvP) SZ(FbPoolP) SZ(GatherP) SZ(MixP) SZ(LinBigP)
    return -1;
}
extern "C" int mms_abi_version(void) { return 3; }
// nonzero = a timing-ablation build (MMS_CXXFLAGS=-DMMS_ABLATE_* / -D*_TIMING): its numbers are diagnostics, its results may be wrong
extern "C" int mms_ablation_build(void) {
#if defined(MMS_ABLATE_STEP) || defined(MMS_ABLATE_SETUP) || defined(MMS_ABLATE_FLUSH) || defined(MMS_ABLATE_STATS) || defined(B4_TIMING) || defined(C3S_TIMING) || defined(C3M_TIMING)
    return 1;
#else
    return 0;
#endif
}
