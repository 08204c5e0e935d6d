"""The reference's survival networks as drop-in nn.Modules whose forward/backward run on the HIP engine.

Constructor signatures, forward signatures, sub-module names (state_dict keys) and parameter creation order are
the reference's (final_multimodal.py:59-150, partial_modality_training.py:165-277, simple_fusion.py:160-236,
flexible_multimodal.py:157-256, train_rnaseq_only.py:126-151; MONAI branch).  The nn.Sequential containers only own parameters; they are never called.  Parameters stay ordinary
autograd leaves: `torch.optim.Adam(model.parameters())`, `clip_grad_norm_`, `state_dict()/load_state_dict()` keep
working (the autograd-compatible path), while `training.train_epoch_*` drive the fused HIP-graph step.
"""
import torch
import torch.nn as nn

from .densenet import DenseNet121
from .engine import engine_of

# The reference picks its CT encoder with a module-level switch set by `try: import monai`
# (final_multimodal.py:34-42).  True -> DenseNet121-3D (MONAI's topology), False -> the in-file 3-conv fallback
# (final_multimodal.py:75-86).  Both run as HIP kernels here; flip it before constructing a model.
USE_MONAI = True


def _ct_encoder(out_dim=128):
    if USE_MONAI:
        return DenseNet121(spatial_dims=3, in_channels=1, out_channels=out_dim, pretrained=False)
    return nn.Sequential(        # parameter holder with the reference's keys (0,1,3,4,6,7); never called
        nn.Conv3d(1, 32, 3, stride=2, padding=1), nn.BatchNorm3d(32), nn.ReLU(),
        nn.Conv3d(32, 64, 3, stride=2, padding=1), nn.BatchNorm3d(64), nn.ReLU(),
        nn.Conv3d(64, out_dim, 3, stride=2, padding=1), nn.BatchNorm3d(out_dim), nn.ReLU(),
        nn.AdaptiveAvgPool3d(1))


def _rna_encoder(rna_dim):
    return nn.Sequential(nn.Linear(rna_dim, 512), nn.BatchNorm1d(512), nn.ReLU(), nn.Dropout(0.3),
                         nn.Linear(512, 128), nn.ReLU())


def _fusion(d):
    return nn.Sequential(nn.Linear(d, 256), nn.BatchNorm1d(256), nn.ReLU(), nn.Dropout(0.3), nn.Linear(256, 128), nn.ReLU())


class _Net(torch.autograd.Function):
    """forward/backward of a whole network on the engine's eager path."""

    @staticmethod
    def forward(ctx, anchor, model, ct, rna, clinical, mask):
        eng = engine_of(model)
        P = eng.plan(rna.shape[0], tuple(ct.shape[-3:]) if ct is not None else None)
        eng.load_batch(P, ct, rna, clinical, mask)
        eng._forward(P, model.training)
        ctx.eng, ctx.P, ctx.train = eng, P, model.training
        hz = P.buf["hz"][:, 0].clone()
        if P.gate is not None:
            return hz, P.gatew.clone()
        return hz, None

    @staticmethod
    def backward(ctx, dhz, dgate):
        if not ctx.train:
            raise RuntimeError("backward through an eval-mode forward is not supported (BN batch statistics)")
        eng, P = ctx.eng, ctx.P
        if eng.params[0].grad is None:
            eng.gflat.zero_()
            eng.attach_grads()
        if dhz is None:
            P.dbuf["hz"].zero_()
        else:
            P.dbuf["hz"][:, 0].copy_(dhz)
        if P.gate is not None:
            ext = dgate.contiguous() if dgate is not None else None
            P.gate.dgate_ext = ext.data_ptr() if ext is not None else None
            old, P.gate.ent_weight = P.gate.ent_weight, 0.0      # the entropy term arrives through dgate here
            eng._backward_from_dhz(P)
            P.gate.ent_weight, P.gate.dgate_ext = old, None
        else:
            eng._backward_from_dhz(P)
        return None, None, None, None, None, None


def _run(model, ct, rna, clinical, mask):
    if not rna.is_cuda or (ct is not None and not ct.is_cuda):
        raise RuntimeError("%s (HIP): inputs must be on an MI355X device; there is no CPU fallback" % type(model).__name__)
    anchor = next(model.parameters())
    if torch.is_grad_enabled() and model.training:
        return _Net.apply(anchor, model, ct, rna, clinical, mask)
    eng = engine_of(model)
    P = eng.plan(rna.shape[0], tuple(ct.shape[-3:]) if ct is not None else None)
    eng.load_batch(P, ct, rna, clinical, mask)
    eng._forward(P, model.training)
    return P.buf["hz"][:, 0].clone(), (P.gatew.clone() if P.gate is not None else None)


class MultiModalSurvivalNet(nn.Module):
    def __init__(self, rna_dim=5005, clinical_dim=1):
        super().__init__()
        self.ct_encoder = _ct_encoder(128)
        self.use_monai = USE_MONAI
        self.ct_pool = nn.AdaptiveAvgPool3d(1)
        self.rna_encoder = _rna_encoder(rna_dim)
        self.clinical_encoder = nn.Sequential(nn.Linear(clinical_dim, 32), nn.ReLU())
        self.fusion = _fusion(128 + 128 + 32)
        self.cox_head = nn.Linear(128, 1)

    def forward(self, ct, rna, clinical):
        return _run(self, ct, rna, clinical, None)[0]


class PartialModalityNet(nn.Module):
    def __init__(self, rna_dim=5005, clinical_dim=1):
        super().__init__()
        self.ct_encoder = _ct_encoder(128)
        self.use_monai = USE_MONAI
        self.ct_pool = nn.AdaptiveAvgPool3d(1)
        self.rna_encoder = _rna_encoder(rna_dim)
        self.clinical_encoder = nn.Sequential(nn.Linear(clinical_dim, 32), nn.ReLU())
        self.gate = nn.Sequential(nn.Linear(128 + 128 + 32 + 3, 64), nn.ReLU(), nn.Linear(64, 3), nn.Softmax(dim=1))
        self.fusion = _fusion(128 + 128 + 32)
        self.cox_head = nn.Linear(128, 1)

    def forward(self, ct, rna, clinical, mask):
        hz, gate = _run(self, ct, rna, clinical, mask)
        return hz, gate


class SimpleFusionModel(nn.Module):
    def __init__(self, rna_dim=5005, img_feature_dim=128, rna_feature_dim=256):
        super().__init__()
        # both widths are free constructor arguments (simple_fusion.py:163); the heads read feature rows with 16-byte loads
        if rna_feature_dim % 4 != 0 or rna_feature_dim <= 0 or img_feature_dim % 4 != 0 or img_feature_dim <= 0:
            raise ValueError("rna_feature_dim and img_feature_dim must be positive multiples of 4 (16-byte aligned feature columns)")
        self.rna_encoder = nn.Sequential(
            nn.Linear(rna_dim, 1024), nn.BatchNorm1d(1024), nn.ReLU(), nn.Dropout(0.3),
            nn.Linear(1024, 512), nn.BatchNorm1d(512), nn.ReLU(), nn.Dropout(0.3),
            nn.Linear(512, rna_feature_dim), nn.ReLU())
        self.image_encoder = _ct_encoder(img_feature_dim)
        self.use_monai = USE_MONAI
        if USE_MONAI:
            self.image_pool = nn.AdaptiveAvgPool3d(1)
        self.fusion = nn.Sequential(
            nn.Linear(rna_feature_dim + img_feature_dim, 256), nn.BatchNorm1d(256), nn.ReLU(), nn.Dropout(0.3),
            nn.Linear(256, 128), nn.ReLU(), nn.Dropout(0.2), nn.Linear(128, 1))

    def forward(self, image, rnaseq):
        return _run(self, image, rnaseq, None, None)[0]


class FlexibleMultimodalModel(nn.Module):
    """flexible_multimodal.py:157-256: SimpleFusion heads on [image | rna] features with a LEARNABLE bias standing in for a
    missing modality (mask (B,2) = [has_image, has_rnaseq]).  Parameter creation order (image_encoder, rna_encoder, the two
    biases via torch.randn, fusion) is the reference's, so a seeded construction draws the same initial weights."""

    def __init__(self, rna_dim=5005, img_feature_dim=128, rna_feature_dim=256):
        super().__init__()
        # both widths are free constructor arguments (simple_fusion.py:163); the heads read feature rows with 16-byte loads
        if rna_feature_dim % 4 != 0 or rna_feature_dim <= 0 or img_feature_dim % 4 != 0 or img_feature_dim <= 0:
            raise ValueError("rna_feature_dim and img_feature_dim must be positive multiples of 4 (16-byte aligned feature columns)")
        self.image_encoder = _ct_encoder(img_feature_dim)
        self.use_monai = USE_MONAI
        if USE_MONAI:
            self.image_pool = nn.AdaptiveAvgPool3d(1)
        self.rna_encoder = nn.Sequential(
            nn.Linear(rna_dim, 1024), nn.BatchNorm1d(1024), nn.ReLU(), nn.Dropout(0.3),
            nn.Linear(1024, 512), nn.BatchNorm1d(512), nn.ReLU(), nn.Dropout(0.3),
            nn.Linear(512, rna_feature_dim), nn.ReLU())
        self.missing_image_bias = nn.Parameter(torch.randn(img_feature_dim))
        self.missing_rna_bias = nn.Parameter(torch.randn(rna_feature_dim))
        self.fusion = nn.Sequential(
            nn.Linear(img_feature_dim + rna_feature_dim, 256), nn.BatchNorm1d(256), nn.ReLU(), nn.Dropout(0.3),
            nn.Linear(256, 128), nn.ReLU(), nn.Dropout(0.2), nn.Linear(128, 1))

    def forward(self, image, rnaseq, mask):
        return _run(self, image, rnaseq, None, mask)[0]


class RNASeqSurvivalModel(nn.Module):
    """train_rnaseq_only.py:126-151: MLP 5005 -> 1024 -> 512 -> 256 -> 1 ([Linear, BatchNorm1d, ReLU, Dropout(0.3)] per hidden
    layer); forward returns the (B, 1) log-hazard like the reference."""

    def __init__(self, input_dim=5005, hidden_dims=[1024, 512, 256]):
        super().__init__()
        layers, in_dim = [], input_dim
        for h in hidden_dims:
            layers.extend([nn.Linear(in_dim, h), nn.BatchNorm1d(h), nn.ReLU(), nn.Dropout(0.3)])
            in_dim = h
        layers.append(nn.Linear(in_dim, 1))
        self.mlp = nn.Sequential(*layers)

    def forward(self, rnaseq):
        return _run(self, None, rnaseq, None, None)[0].unsqueeze(1)
