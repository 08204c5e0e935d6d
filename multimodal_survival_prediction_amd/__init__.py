"""MI355X-native (gfx950) training hot path of baek0203/multimodal_survival_prediction.

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed only); every numeric op of the
hot path runs in hand-written HIP kernels behind the C ABI in include/mmsurv.h (libmmsurv_hip.so).
There is no CPU fallback: ops raise if the library is missing.

Modules: models (the reference's five nn.Modules), losses, training (train_epoch_* / validate_* of each script, lock-step
K-fold variants), engine (fused HIP-graph step of one model), fold_group (K fold models advanced by one launch sequence),
data / cohort_io (synthetic cohorts, on-disk contract, GPU preprocessing), distributed (fold sharding, DDP helpers).
"""
from ._lib import lib_path, load_library  # noqa: F401

__all__ = ["load_library", "lib_path"]
