"""CPU restatement of the reference's per-item preprocessing (test infrastructure only).

ct_preprocess: partial_modality_training.py:94-109 / simple_fusion.py:117-134 / flexible_multimodal.py:118-128 --
min-max normalise, then scipy.ndimage.zoom(order=1) with zoom factors target/native (scipy IS the library the reference
calls, so this function is the reference's own arithmetic, not a re-derivation).
rna_log_zscore: preprocess_genomic.py:108-117 -- log2(count + 1) then sklearn StandardScaler."""
import numpy as np


def ct_preprocess(img_np, target_size):
    from scipy.ndimage import zoom
    img_np = (img_np - img_np.min()) / (img_np.max() - img_np.min() + 1e-8)
    D, H, W = img_np.shape
    td, th, tw = target_size
    return zoom(img_np, (td / D, th / H, tw / W), order=1).astype(np.float32)


def rna_log_zscore(counts):
    from sklearn.preprocessing import StandardScaler
    return StandardScaler().fit_transform(np.log2(counts.astype(np.float64) + 1)).astype(np.float32)
