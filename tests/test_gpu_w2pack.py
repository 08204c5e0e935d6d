"""Packed primary storage of the conv2 weights (MmsDnOpts.w2_packed, round 4): the 3x3x3 weights of the 58 dense layers live as
[cout][tap][cin] inside the engine's flat buffers (strided nn.Parameter views), the weight-gradient kernels write the gradient in place,
and the optimiser step emits the derived packs -- no pack / unpack launch in the step.  Checked here through the C ABI:
the derived packs against the stand-alone pack kernels on the torch-layout weights; the fused update against the flat Adam kernel on the
same numbers; the module surface (state_dict round trip through the ORACLE class, .grad shapes, a torch optimiser on the strided views)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import DEV, assert_close

W2N = 32 * 27 * 128


def _setup(n_w2, fragmask, seed=0, gap=(37, 260, 8)):
    """A flat buffer with n_w2 conv2 tensors in packed primary layout between gaps of ordinary parameters -> (flat, g, m, v, canon, w2)."""
    from multimodal_survival_prediction_amd import ops
    g_ = torch.Generator().manual_seed(seed)
    offs, o = [], 0
    for i in range(n_w2):
        o += gap[i % len(gap)] * 4
        offs.append(o)
        o += W2N
    n = o + 64
    mk = lambda s: (torch.randn(n, generator=g_) * s).to(DEV)
    flat, g, m, v = mk(0.05), mk(1e-3), mk(1e-3), (mk(1e-3) ** 2)
    packs = torch.zeros(n_w2, 2, W2N, device=DEV)
    base = packs.data_ptr()
    w2 = dict(off=torch.tensor(offs, dtype=torch.int64, device=DEV),
              pack_b=torch.tensor([base + 2 * i * W2N * 4 for i in range(n_w2)], dtype=torch.int64, device=DEV),
              pack_f=torch.tensor([base + (2 * i + 1) * W2N * 4 for i in range(n_w2)], dtype=torch.int64, device=DEV), fragmask=fragmask)
    return flat, g, m, v, offs, packs, w2


def _canon(flat, off):
    """torch-layout [32][128][27] copy of the packed primary tensor at `off`."""
    return flat[off:off + W2N].view(32, 27, 128).permute(0, 2, 1).contiguous()


@pytest.mark.parametrize("fragmask", [0b0000, 0b0110])
def test_w2_pack_equals_standalone_pack_kernels(fragmask):
    from multimodal_survival_prediction_amd import _lib, ops
    flat, g, m, v, offs, packs, w2 = _setup(4, fragmask)
    hyper = torch.tensor([1e-4, 0.9, 0.999, 1e-8, 1e-4, 1.0], device=DEV)
    sumsq, step = torch.zeros(1, dtype=torch.float64, device=DEV), torch.zeros(1, device=DEV)      # (kept alive: the block holds raw pointers)
    a = ops.adam_params(flat, g, m, v, hyper, sumsq, step, w2=w2)
    before = flat.clone()
    ops.call("mms_w2_pack", a)
    torch.cuda.synchronize()
    assert torch.equal(flat, before)                                   # packing never touches the weights
    for i, off in enumerate(offs):
        w = _canon(flat, off)
        wpf, wpb = ops.pack_conv3(w.view(32, 128, 3, 3, 3))
        assert torch.equal(wpf, flat[off:off + W2N])                   # the primary storage IS the classic forward pack
        if (fragmask >> i) & 1:
            wff, wfb = ops.pack_conv3_frag(w.view(32, 128, 3, 3, 3))
            assert torch.equal(packs[i, 0], wfb) and torch.equal(packs[i, 1], wff)
        else:
            assert torch.equal(packs[i, 0], wpb)


@pytest.mark.parametrize("adamw,skip", [(False, None), (True, None), (False, 0.0), (False, 1.0)])
def test_clip_adam_with_packed_tensors_equals_flat_update(adamw, skip):
    """mms_clip_adam on a buffer with packed conv2 tensors (gap kernel + the (layer, tap) kernel) == the flat kernel on the same numbers,
    bit for bit; the packs it leaves == the packs of the updated weights; a skipped step changes nothing."""
    from multimodal_survival_prediction_amd import ops
    fragmask = 0b101
    flat, g, m, v, offs, packs, w2 = _setup(3, fragmask, seed=3)
    hyper = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 1e-2, 0.5], device=DEV)
    sk = torch.tensor([skip], device=DEV) if skip is not None else None
    ref = [t.clone() for t in (flat, g, m, v)]
    keep = []
    for bufs, w2_ in ((ref, None), ((flat, g, m, v), w2)):
        sumsq, step = torch.zeros(1, dtype=torch.float64, device=DEV), torch.zeros(1, device=DEV)      # (kept alive: the block holds raw pointers)
        a = ops.adam_params(bufs[0], bufs[1], bufs[2], bufs[3], hyper, sumsq, step, sk, adamw, w2=w2_)
        keep.append((a, sumsq, step))
        ops.call("mms_w2_pack", a) if w2_ is not None else None
        ops.call("mms_grad_sumsq", a)
        ops.call("mms_clip_adam", a)
    torch.cuda.synchronize()
    for got, want, name in zip((flat, m, v), (ref[0], ref[2], ref[3]), "pmv"):
        assert_close(got, want, 1e-6, name)                              # (same formula; the two kernels may contract an fma differently)
    assert float((flat - _setup(3, fragmask, seed=3)[0]).abs().max()) > 0 or skip == 0.0
    if skip == 0.0:
        assert torch.equal(flat, _setup(3, fragmask, seed=3)[0])
    for i, off in enumerate(offs):
        w = _canon(flat, off).view(32, 128, 3, 3, 3)
        if (fragmask >> i) & 1:
            wff, wfb = ops.pack_conv3_frag(w)
            assert torch.equal(packs[i, 0], wfb) and torch.equal(packs[i, 1], wff)
        else:
            assert torch.equal(packs[i, 0], ops.pack_conv3(w)[1])


def test_packed_model_surface():
    """The module surface with packed storage: conv2 parameters are strided views with torch shapes; a state_dict round trip through the
    ORACLE class reproduces the eval hazards; .grad after an engine step has the parameter's shape and matches the oracle's gradient;
    an in-place change of the weights by torch (what a torch optimiser or load_state_dict does) is picked up (packs rebuilt)."""
    from oracle import models as OM
    from oracle import losses as OL
    from multimodal_survival_prediction_amd import models as HM
    from multimodal_survival_prediction_amd.engine import engine_of
    from test_gpu_densenet import structured_volumes
    torch.manual_seed(5)
    ref = OM.MultiModalSurvivalNet(rna_dim=64, use_monai=True)
    for mod in ref.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    net = HM.MultiModalSurvivalNet(rna_dim=64)
    net.load_state_dict(ref.state_dict())
    for mod in net.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    net.to(DEV)
    eng = engine_of(net)
    w = net.ct_encoder.features.denseblock2.denselayer3.layers.conv2.weight
    assert tuple(w.shape) == (32, 128, 3, 3, 3) and not w.is_contiguous() and w.stride() == (27 * 128, 1, 9 * 128, 3 * 128, 128)
    assert eng.dn_opts.w2_packed == 1 and len(eng.w2_offsets) == 58
    B, dims = 4, (64, 64, 32)
    ct, rna, clin = structured_volumes(B, dims, 3), torch.randn(B, 64), torch.rand(B, 1)
    t, e = torch.tensor([5.0, 3.0, 9.0, 1.0]), torch.tensor([1.0, 0.0, 1.0, 1.0])
    ref.eval(); net.eval()
    with torch.no_grad():
        assert_close(net(ct.to(DEV), rna.to(DEV), clin.to(DEV)), ref(ct, rna, clin), 1e-4, "eval hazards")
    # state_dict of the HIP model -> a fresh ORACLE model
    ref2 = OM.MultiModalSurvivalNet(rna_dim=64, use_monai=True)
    ref2.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
    ref2.eval()
    with torch.no_grad():
        assert_close(ref2(ct, rna, clin), ref(ct, rna, clin), 1e-6, "round trip through state_dict")
    # gradients: autograd-compatible path
    ref.train(); net.train()
    OL.cox_loss(ref(ct, rna, clin), e, t).backward()
    from multimodal_survival_prediction_amd import losses as HL
    HL.cox_loss(net(ct.to(DEV), rna.to(DEV), clin.to(DEV)), e.to(DEV), t.to(DEV)).backward()
    torch.cuda.synchronize()
    wr = dict(ref.named_parameters())["ct_encoder.features.denseblock4.denselayer16.layers.conv2.weight"]
    wh = dict(net.named_parameters())["ct_encoder.features.denseblock4.denselayer16.layers.conv2.weight"]
    assert wh.grad.shape == wr.grad.shape
    assert_close(wh.grad, wr.grad, 2e-4, "conv2 weight gradient (last layer: no ReLU flip downstream)")
    # torch changes the weights in place -> the derived packs follow
    with torch.no_grad():
        for p, q in zip(net.parameters(), ref.parameters()):
            p.mul_(0.5); q.mul_(0.5)
    ref.eval(); net.eval()
    with torch.no_grad():
        assert_close(net(ct.to(DEV), rna.to(DEV), clin.to(DEV)), ref(ct, rna, clin), 1e-4, "eval hazards after an in-place update by torch")
