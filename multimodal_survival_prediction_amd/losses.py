"""Loss / metric surface of the reference's training scripts, backed by the HIP kernels.

Same names, argument meaning and degenerate-batch behaviour as final_multimodal.py:158-194,
partial_modality_training.py:285-331, simple_fusion.py:47-73:
  cox_loss(hazard, event, time), neg_partial_log_likelihood(log_hazard, event, time), gate_entropy_loss(gate),
  calculate_cindex(hazard, event, time) -> float, ConcordanceIndex()(log_hazard, event, time) -> 0-d tensor.

The reference picks its loss/metric implementation with a module-level switch set by `try: import torchsurv`
(final_multimodal.py:24-32, simple_fusion.py:22-29).  `USE_TORCHSURV` below is that switch:
  True  (the reference's documented install, requirements.txt:28-36): torchsurv semantics -- Efron tie correction in the
        partial likelihood when survival times repeat, 0.5 credit for tied risk scores in the C-index;
  False: the in-file fallbacks -- Breslow-equivalent sorted-logcumsumexp loss, tied scores counted discordant
        (simple_fusion.py:59-73); final/partial `calculate_cindex` falls back to lifelines there (0.5 credit).
On distinct times and distinct scores (all synthetic cohorts) both settings give identical numbers.
"""
import torch

from . import _lib, ops

USE_TORCHSURV = True


def default_ties():
    return "efron" if USE_TORCHSURV else "breslow"


class _Cox(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hazard, event, time, ties):
        if not hazard.is_cuda:
            raise RuntimeError("cox_loss (HIP): tensors must be on the GPU; there is no CPU fallback")
        h = hazard.detach().reshape(-1).contiguous().float()
        out, dh = ops.cox_fwd_bwd(h, time.reshape(-1).contiguous().float(), event.reshape(-1).float().contiguous(), ties=ties)
        ctx.save_for_backward(dh)
        ctx.shape = hazard.shape
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        (dh,) = ctx.saved_tensors
        return (dh * g).reshape(ctx.shape), None, None, None


def cox_loss(hazard, event, time, ties=None):
    """Cox loss of the reference (final_multimodal.py:158-186): 0 for n<2 or no events.  ties: "breslow" | "efron"
    (default: what the USE_TORCHSURV switch selects)."""
    return _Cox.apply(hazard, event, time, ties or default_ties())


def neg_partial_log_likelihood(log_hazard, event, time, ties=None):
    """simple_fusion.py:47-57 / torchsurv call signature (event may be bool)."""
    return _Cox.apply(log_hazard, event, time, ties or default_ties())


class _GateEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gate):
        if not gate.is_cuda:
            raise RuntimeError("gate_entropy_loss (HIP): tensor must be on the GPU")
        g = gate.detach().contiguous().float()
        loss = torch.zeros(1, device=g.device)
        dg = torch.empty_like(g)
        _lib.check(_lib.load_library().mms_gate_entropy(g.data_ptr(), g.shape[0], 1.0, loss.data_ptr(), dg.data_ptr(),
                                                        ops.stream()), "mms_gate_entropy")
        ctx.save_for_backward(dg)
        return loss[0]

    @staticmethod
    def backward(ctx, go):
        (dg,) = ctx.saved_tensors
        return dg * go


def gate_entropy_loss(gate_weights):
    """partial_modality_training.py:322-331."""
    return _GateEntropy.apply(gate_weights)


def cindex_counts(hazard, event, time):
    h = hazard.detach().reshape(-1).contiguous().float()
    return ops.cindex_counts(h, time.reshape(-1).contiguous().float(), event.reshape(-1).float().contiguous())


def calculate_cindex(hazard, event, time, tie_credit=0.5):
    """Harrell C over a validation fold, final_multimodal.py:164-169,188-194 / partial_modality_training.py:290-319: torchsurv's
    ConcordanceIndex, else lifelines.concordance_index -- both give 0.5 credit to pairs with tied risk scores (a collapsed model
    scores 0.5, which is what ReduceLROnPlateau / best-checkpoint selection see).  tie_credit=0 is the counting rule of the
    in-file fallback class of simple_fusion.py:59-73 (tied scores discordant)."""
    if not hazard.is_cuda:
        hazard, event, time = hazard.cuda(), event.cuda(), time.cuda()
    c = cindex_counts(hazard, event, time).tolist()
    return (c[0] + tie_credit * c[1]) / c[2] if c[2] > 0 else 0.5


class ConcordanceIndex:
    """simple_fusion.py:30-31,59-73 / train_rnaseq_only.py:55-70: torchsurv's metric (0.5 tie credit) when USE_TORCHSURV, else the
    in-file fallback (tied scores discordant); `tie_credit` overrides."""

    def __init__(self, tie_credit=None):
        self.tie_credit = tie_credit

    def __call__(self, log_hazard, event, time):
        tc = self.tie_credit if self.tie_credit is not None else (0.5 if USE_TORCHSURV else 0.0)
        return torch.tensor(calculate_cindex(log_hazard, event, time, tie_credit=tc))
