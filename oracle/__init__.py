"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

A CPU restatement (plain PyTorch fp32 / numpy fp64) of the reference's training hot path, used as the
checker for the HIP path.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this package; the product package `multimodal_survival_prediction_amd` never does (its ops raise if
the HIP library is missing -- there is no CPU fallback).

Pinning status ("Oracle"):
  * heads, fallback 3-conv CT encoder, custom Cox loss, C-index, gate entropy, train_epoch/validate:
    PINNED against golden vectors produced by executing the reference's own definitions in the build
    container (tests/golden/generate_golden.py -> tests/golden/g*.npz; tests/test_oracle_golden.py).
  * MONAI DenseNet121(spatial_dims=3) and torchsurv's Efron tie handling: PARITY UNPINNED -- neither
    library is vendored in /root/reference nor installed in the image; `densenet3d.py` restates MONAI's
    published topology (monai>=1.3, monai/networks/nets/densenet.py) and is sanity-pinned only by the
    parameter count 11,373,824 and the state_dict key set.
"""
