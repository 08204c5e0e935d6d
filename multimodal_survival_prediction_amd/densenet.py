"""DenseNet121 (3-D) CT encoder: drop-in for `monai.networks.nets.DenseNet121(spatial_dims=3, in_channels=1,
out_channels=128, pretrained=False)` as the reference instantiates it (final_multimodal.py:66-71,
partial_modality_training.py:171-176, simple_fusion.py:182-187).

The module tree below exists only to own parameters/buffers under MONAI's state_dict keys (so `state_dict()` /
`load_state_dict()` / `torch.optim` keep working); none of its sub-modules is ever called.  `forward` hands raw
device pointers to the C ABI (`mms_dn121_forward/backward`, include/mmsurv.h) and fails loudly without the HIP
library or on a non-GPU tensor -- there is no torch fallback.
"""
import ctypes
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib

_BLOCKS = (6, 12, 24, 16)
_GROWTH, _BN_SIZE, _INIT = 32, 4, 64


def _holder(**mods):
    m = nn.Module()
    for k, v in mods.items():
        m.add_module(k, v)
    return m


def _build_features():
    f = nn.Module()
    f.add_module("conv0", nn.Conv3d(1, _INIT, 7, stride=2, padding=3, bias=False))
    f.add_module("norm0", nn.BatchNorm3d(_INIT))
    f.add_module("relu0", nn.ReLU(inplace=True))
    f.add_module("pool0", nn.MaxPool3d(3, 2, 1))
    c = _INIT
    for b, n in enumerate(_BLOCKS):
        block = nn.Module()
        for i in range(n):
            mid = _BN_SIZE * _GROWTH
            layers = _holder(norm1=nn.BatchNorm3d(c), relu1=nn.ReLU(inplace=True),
                             conv1=nn.Conv3d(c, mid, 1, bias=False), norm2=nn.BatchNorm3d(mid),
                             relu2=nn.ReLU(inplace=True), conv2=nn.Conv3d(mid, _GROWTH, 3, padding=1, bias=False))
            block.add_module("denselayer%d" % (i + 1), _holder(layers=layers))
            c += _GROWTH
        f.add_module("denseblock%d" % (b + 1), block)
        if b == len(_BLOCKS) - 1:
            f.add_module("norm5", nn.BatchNorm3d(c))
        else:
            f.add_module("transition%d" % (b + 1),
                         _holder(norm=nn.BatchNorm3d(c), relu=nn.ReLU(inplace=True),
                                 conv=nn.Conv3d(c, c // 2, 1, bias=False), pool=nn.AvgPool3d(2, 2)))
            c //= 2
    return f, c


class _Encode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, net):
        ctx.net = net
        ctx.save_for_backward(x)
        ctx.train = net.training
        return net._run_forward(x)

    @staticmethod
    def backward(ctx, dout):
        if not ctx.train:
            raise RuntimeError("DenseNet121 (HIP): backward needs a training-mode forward (BN batch statistics)")
        (x,) = ctx.saved_tensors
        ctx.net._run_backward(x, dout)
        return None, None, None


class DenseNet121(nn.Module):
    def __init__(self, spatial_dims=3, in_channels=1, out_channels=128, pretrained=False):
        super().__init__()
        if spatial_dims != 3 or in_channels != 1 or pretrained:
            raise ValueError("only DenseNet121(spatial_dims=3, in_channels=1, pretrained=False) is implemented "
                             "(the reference's call signature)")
        self.features, c = _build_features()
        self.class_layers = _holder(relu=nn.ReLU(inplace=True), pool=nn.AdaptiveAvgPool3d(1), flatten=nn.Flatten(1),
                                    out=nn.Linear(c, out_channels))
        if not 1 <= out_channels <= 4096 or out_channels % 4:
            raise ValueError("out_channels must be a multiple of 4 in [4, 4096] (16-byte aligned feature columns)")
        for m in self.modules():   # MONAI's init
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm3d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.constant_(m.bias, 0)
        self._eng = None      # engine state: (key, workspace, tables)
        self._gflat = None
        self.dn_opts = {}     # launch-shape options of this module's eager path (fields of MmsDnOpts, include/mmsurv.h; {} = defaults)

    def _opts(self):
        from . import ops
        return ops.dn_opts(self.dn_opts, out_features=self.class_layers.out.out_features)      # (incl. the width of class_layers.out)

    # ---- engine plumbing -------------------------------------------------------------------------
    def _tables(self, x):
        B, _, D, H, W = x.shape
        params = list(self.parameters())
        bufs = list(self.buffers())
        key = (x.device, B, D, H, W, params[0].data_ptr(), params[-1].data_ptr(), bufs[0].data_ptr())
        if self._eng is not None and self._eng["key"] == key:
            return self._eng
        lib = _lib.load_library()
        assert len(params) == 364 and len(bufs) == 363
        for p in params:
            if p.device != x.device or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("DenseNet121 (HIP): parameters must be contiguous fp32 on the input's device")
        nbytes = ctypes.c_size_t(0)
        _lib.check(lib.mms_dn121_workspace_bytes(B, D, H, W, ctypes.byref(nbytes)), "mms_dn121_workspace_bytes")
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=x.device)
        ptab = (ctypes.c_void_p * 364)(*[p.data_ptr() for p in params])
        btab = (ctypes.c_void_p * 363)(*[b.data_ptr() for b in bufs])
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.mms_dn121_init(ws.data_ptr(), B, D, H, W, ptab, btab, None, st), "mms_dn121_init")      # (torch-layout conv2 weights)
        self._eng = dict(key=key, ws=ws, ptab=ptab, btab=btab, dims=(B, D, H, W), lib=lib)
        return self._eng

    def _check_input(self, x):
        if not x.is_cuda:
            raise RuntimeError("DenseNet121 (HIP): input must live on an MI355X device; there is no CPU fallback")
        if x.dim() != 5 or x.shape[1] != 1:
            raise ValueError("expected (B, 1, D, H, W)")
        return x.contiguous().float()

    def _run_forward(self, x):
        e = self._tables(x)
        B, D, H, W = e["dims"]
        nout = self.class_layers.out.out_features
        out = torch.empty(B, nout, device=x.device, dtype=torch.float32)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(e["lib"].mms_dn121_forward(e["ws"].data_ptr(), B, D, H, W, x.data_ptr(), e["ptab"], e["btab"],
                                              out.data_ptr(), out.stride(0), 1 if self.training else 0, ctypes.byref(self._opts()), st),
                   "mms_dn121_forward")
        return out

    def _grad_table(self):
        params = list(self.parameters())
        n = sum(p.numel() for p in params)
        if self._gflat is None or self._gflat.device != params[0].device or self._gflat.numel() != n:
            self._gflat = torch.zeros(n, device=params[0].device)
            self._gviews = None
        if self._gviews is None:
            views, o = [], 0
            for p in params:
                views.append(self._gflat[o:o + p.numel()].view_as(p))
                o += p.numel()
            self._gviews = views
        fresh = params[0].grad is None
        if fresh:
            self._gflat.zero_()
        gt = []
        for p, v in zip(params, self._gviews):
            if p.grad is None:
                if not fresh:
                    v.zero_()
                p.grad = v
            gt.append(p.grad.data_ptr())
        return (ctypes.c_void_p * 364)(*gt)

    def _run_backward(self, x, dout):
        e = self._tables(x)
        B, D, H, W = e["dims"]
        gtab = self._grad_table()
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        dout = dout.contiguous().float()
        _lib.check(e["lib"].mms_dn121_backward(e["ws"].data_ptr(), B, D, H, W, x.data_ptr(), e["ptab"],
                                               dout.data_ptr(), dout.stride(0), gtab, ctypes.byref(self._opts()), st), "mms_dn121_backward")

    def workspace_region(self, name, index=0, dtype=torch.float32):
        """Diagnostic view of a named workspace region (tests)."""
        e = self._eng
        B, D, H, W = e["dims"]
        off, nb = ctypes.c_size_t(0), ctypes.c_size_t(0)
        _lib.check(e["lib"].mms_dn121_region(B, D, H, W, name.encode(), index, ctypes.byref(off), ctypes.byref(nb)),
                   "mms_dn121_region")
        return e["ws"][off.value:off.value + nb.value].view(dtype)

    def forward(self, x):
        x = self._check_input(x)
        anchor = self.features.conv0.weight
        if torch.is_grad_enabled() and anchor.requires_grad:
            return _Encode.apply(x, anchor, self)
        return self._run_forward(x)
