"""CPU restatement of the reference's three survival networks (plain torch.nn, fp32).

Each class keeps the reference's constructor signature, forward signature, sub-module names (state_dict keys)
and parameter creation ORDER (so `torch.manual_seed(s)` construction reproduces the reference's weights
bit-for-bit; pinned by tests/golden/g3_models.npz).  `use_monai` selects the CT encoder the reference's
`USE_MONAI` import switch would select: True -> DenseNet121-3D (oracle/densenet3d.py, parity unpinned),
False -> the in-file 3-conv fallback (pinned).
"""
import torch
import torch.nn as nn

from .densenet3d import DenseNet121


def _fallback_encoder(out_dim=128):
    # final_multimodal.py:75-86, partial_modality_training.py:179-190, simple_fusion.py:191-202
    return nn.Sequential(
        nn.Conv3d(1, 32, 3, stride=2, padding=1), nn.BatchNorm3d(32), nn.ReLU(),
        nn.Conv3d(32, 64, 3, stride=2, padding=1), nn.BatchNorm3d(64), nn.ReLU(),
        nn.Conv3d(64, out_dim, 3, stride=2, padding=1), nn.BatchNorm3d(out_dim), nn.ReLU(),
        nn.AdaptiveAvgPool3d(1),
    )


def _rna_encoder(rna_dim):
    # final_multimodal.py:93-100 / partial_modality_training.py:196-203
    return nn.Sequential(nn.Linear(rna_dim, 512), nn.BatchNorm1d(512), nn.ReLU(), nn.Dropout(0.3),
                         nn.Linear(512, 128), nn.ReLU())


def _fusion(fusion_dim):
    # final_multimodal.py:110-117 / partial_modality_training.py:222-229
    return nn.Sequential(nn.Linear(fusion_dim, 256), nn.BatchNorm1d(256), nn.ReLU(), nn.Dropout(0.3),
                         nn.Linear(256, 128), nn.ReLU())


class MultiModalSurvivalNet(nn.Module):
    """final_multimodal.py:59-150 -- plain concat late fusion."""

    def __init__(self, rna_dim=5005, clinical_dim=1, use_monai=True):
        super().__init__()
        if use_monai:
            self.ct_encoder = DenseNet121(spatial_dims=3, in_channels=1, out_channels=128, pretrained=False)
        else:
            self.ct_encoder = _fallback_encoder(128)
        self.use_monai = use_monai
        self.ct_pool = nn.AdaptiveAvgPool3d(1)
        self.rna_encoder = _rna_encoder(rna_dim)
        self.clinical_encoder = nn.Sequential(nn.Linear(clinical_dim, 32), nn.ReLU())
        self.fusion = _fusion(128 + 128 + 32)
        self.cox_head = nn.Linear(128, 1)

    def forward(self, ct, rna, clinical):
        ct_feat = self.ct_encoder(ct)
        ct_feat = ct_feat.view(ct_feat.size(0), -1)           # :126-135 (both branches end as (B,128))
        rna_feat = self.rna_encoder(rna)                       # :138
        clin_feat = self.clinical_encoder(clinical)            # :141
        fused = torch.cat([ct_feat, rna_feat, clin_feat], dim=1)   # :144
        fused = self.fusion(fused)
        return self.cox_head(fused).squeeze(1)                 # :148


class PartialModalityNet(nn.Module):
    """partial_modality_training.py:165-277 -- modality masks + softmax gate."""

    def __init__(self, rna_dim=5005, clinical_dim=1, use_monai=True):
        super().__init__()
        if use_monai:
            self.ct_encoder = DenseNet121(spatial_dims=3, in_channels=1, out_channels=128, pretrained=False)
        else:
            self.ct_encoder = _fallback_encoder(128)
        self.use_monai = use_monai
        self.ct_pool = nn.AdaptiveAvgPool3d(1)
        self.rna_encoder = _rna_encoder(rna_dim)
        self.clinical_encoder = nn.Sequential(nn.Linear(clinical_dim, 32), nn.ReLU())
        self.gate = nn.Sequential(nn.Linear(128 + 128 + 32 + 3, 64), nn.ReLU(), nn.Linear(64, 3), nn.Softmax(dim=1))
        self.fusion = _fusion(128 + 128 + 32)
        self.cox_head = nn.Linear(128, 1)

    def forward(self, ct, rna, clinical, mask):
        B = ct.size(0)
        ct_feat = self.ct_encoder(ct).view(B, -1)              # :245-251
        rna_feat = self.rna_encoder(rna)
        clin_feat = self.clinical_encoder(clinical)
        ct_feat = ct_feat * mask[:, 0:1]                       # :257-259
        rna_feat = rna_feat * mask[:, 1:2]
        clin_feat = clin_feat * mask[:, 2:3]
        gate_weights = self.gate(torch.cat([ct_feat, rna_feat, clin_feat, mask], dim=1))   # :262-263
        fused = torch.cat([ct_feat * gate_weights[:, 0:1], rna_feat * gate_weights[:, 1:2],
                           clin_feat * gate_weights[:, 2:3]], dim=1)                       # :266-271
        fused = self.fusion(fused)
        return self.cox_head(fused).squeeze(1), gate_weights


class SimpleFusionModel(nn.Module):
    """simple_fusion.py:160-236 -- 3-layer RNA MLP + CT encoder, concat, 3-layer head."""

    def __init__(self, rna_dim=5005, img_feature_dim=128, rna_feature_dim=256, use_monai=True):
        super().__init__()
        self.rna_encoder = nn.Sequential(
            nn.Linear(rna_dim, 1024), nn.BatchNorm1d(1024), nn.ReLU(), nn.Dropout(0.3),
            nn.Linear(1024, 512), nn.BatchNorm1d(512), nn.ReLU(), nn.Dropout(0.3),
            nn.Linear(512, rna_feature_dim), nn.ReLU())
        if use_monai:
            self.image_encoder = DenseNet121(spatial_dims=3, in_channels=1, out_channels=img_feature_dim,
                                             pretrained=False)
            self.image_pool = nn.AdaptiveAvgPool3d(1)
        else:
            self.image_encoder = _fallback_encoder(img_feature_dim)
        self.use_monai = use_monai
        self.fusion = nn.Sequential(
            nn.Linear(rna_feature_dim + img_feature_dim, 256), nn.BatchNorm1d(256), nn.ReLU(), nn.Dropout(0.3),
            nn.Linear(256, 128), nn.ReLU(), nn.Dropout(0.2), nn.Linear(128, 1))

    def forward(self, image, rnaseq):
        B = image.size(0)
        rna_feat = self.rna_encoder(rnaseq)
        img_feat = self.image_encoder(image).view(B, -1)
        fused = torch.cat([rna_feat, img_feat], dim=1)         # :233
        return self.fusion(fused).squeeze(1)


class FlexibleMultimodalModel(nn.Module):
    """flexible_multimodal.py:157-256 -- SimpleFusion-style heads, features ordered [image | rna] (:252), a learnable
    bias replaces a missing modality's features (:205-206, :243-250).  Creation order: image_encoder, rna_encoder,
    missing_image_bias, missing_rna_bias (torch.randn), fusion."""

    def __init__(self, rna_dim=5005, img_feature_dim=128, rna_feature_dim=256, use_monai=True):
        super().__init__()
        if use_monai:
            self.image_encoder = DenseNet121(spatial_dims=3, in_channels=1, out_channels=img_feature_dim,
                                             pretrained=False)
            self.use_monai = True
            self.image_pool = nn.AdaptiveAvgPool3d(1)
        else:
            self.image_encoder = _fallback_encoder(img_feature_dim)
            self.use_monai = False
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
        B = image.size(0)
        img_feat = self.image_encoder(image).view(B, -1)
        rna_feat = self.rna_encoder(rnaseq)
        img_mask, rna_mask = mask[:, 0:1], mask[:, 1:2]
        img_feat = img_feat * img_mask + self.missing_image_bias.unsqueeze(0) * (1 - img_mask)      # :249
        rna_feat = rna_feat * rna_mask + self.missing_rna_bias.unsqueeze(0) * (1 - rna_mask)        # :250
        return self.fusion(torch.cat([img_feat, rna_feat], dim=1)).squeeze(1)


class RNASeqSurvivalModel(nn.Module):
    """train_rnaseq_only.py:126-151 -- MLP over RNA-seq only; returns (B, 1)."""

    def __init__(self, input_dim=5005, hidden_dims=[1024, 512, 256]):
        super().__init__()
        layers, in_dim = [], input_dim
        for h in hidden_dims:
            layers.extend([nn.Linear(in_dim, h), nn.BatchNorm1d(h), nn.ReLU(), nn.Dropout(0.3)])
            in_dim = h
        layers.append(nn.Linear(in_dim, 1))
        self.mlp = nn.Sequential(*layers)

    def forward(self, rnaseq):
        return self.mlp(rnaseq)
