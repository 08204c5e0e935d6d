"""Loss / metric surface of the reference's training scripts, backed by the HIP kernels.

Same names, argument meaning and degenerate-batch behaviour as final_multimodal.py:158-194,
partial_modality_training.py:285-331, simple_fusion.py:47-73:
  cox_loss(hazard, event, time), neg_partial_log_likelihood(log_hazard, event, time), gate_entropy_loss(gate),
  calculate_cindex(hazard, event, time) -> float, ConcordanceIndex()(log_hazard, event, time) -> 0-d tensor.
Ties: Breslow risk sets / fallback C-index counting (tied hazards discordant); synthetic times are distinct.
"""
import torch

from . import _lib, ops


class _Cox(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hazard, event, time):
        if not hazard.is_cuda:
            raise RuntimeError("cox_loss (HIP): tensors must be on the GPU; there is no CPU fallback")
        h = hazard.detach().reshape(-1).contiguous().float()
        out, dh = ops.cox_fwd_bwd(h, time.reshape(-1).contiguous().float(), event.reshape(-1).float().contiguous())
        ctx.save_for_backward(dh)
        ctx.shape = hazard.shape
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        (dh,) = ctx.saved_tensors
        return (dh * g).reshape(ctx.shape), None, None


def cox_loss(hazard, event, time):
    """Custom Cox loss of the reference (final_multimodal.py:171-186): 0 for n<2 or no events."""
    return _Cox.apply(hazard, event, time)


def neg_partial_log_likelihood(log_hazard, event, time):
    """simple_fusion.py:47-57 / torchsurv call signature (event may be bool)."""
    return _Cox.apply(log_hazard, event, time)


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


def calculate_cindex(hazard, event, time, tie_credit=0.0):
    """Harrell C over a validation fold.  tie_credit=0 reproduces the reference's in-file fallback
    (simple_fusion.py:59-73); 0.5 is what torchsurv/lifelines give for tied risk scores."""
    if not hazard.is_cuda:
        hazard, event, time = hazard.cuda(), event.cuda(), time.cuda()
    c = cindex_counts(hazard, event, time).tolist()
    return (c[0] + tie_credit * c[1]) / c[2] if c[2] > 0 else 0.5


class ConcordanceIndex:
    def __call__(self, log_hazard, event, time):
        return torch.tensor(calculate_cindex(log_hazard, event, time))
