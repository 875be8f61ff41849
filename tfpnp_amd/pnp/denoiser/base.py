"""UNetDenoiser2D on MI355X -- drop-in for tfpnp/pnp/denoiser/base.py:7-32.

Same call surface: `denoiser(x[B,1,H,W], sigma[B]) -> clamp(UNet(cat[x, sigma*1]), 0, 1)`, weights frozen,
loaded from a state_dict with the reference's key names (`inc.conv.conv-0.conv2d.weight` ...).  The forward
pass is one libpnpx.so call (hand-written gfx950 kernels); there is no PyTorch/CPU fallback.
"""
import os

import torch

from ... import ops
from ... import torch_ops as T
from ..._lib import PnpxError

CURRENT_DIR = os.path.dirname(os.path.abspath(__file__))


class UNetDenoiser2D(torch.nn.Module):
    def __init__(self, ckpt_path=None, state_dict=None, conv_mode=None):
        """conv_mode: None/0 = fp32 MFMA convolutions, the reference's arithmetic (default since r6); 1 = the fast mode:
        half-split f16 MFMA convolutions (22-bit significand, ~1.55x the episode rate)."""
        super().__init__()
        self.conv_mode = conv_mode
        if state_dict is None:
            if ckpt_path is None:
                ckpt_path = os.path.join(CURRENT_DIR, 'pretrained', 'unet-nm.pt')
                if not os.path.exists(ckpt_path):
                    # same error contract as the reference (denoiser/base.py:12-13)
                    raise ValueError('Default ckpt not found, you have to provide a ckpt path')
            state_dict = torch.load(ckpt_path, map_location='cpu')
        self._state = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else torch.as_tensor(v))
                       for k, v in state_dict.items()}
        self._ctx = {}  # device index -> ops.Context holding the packed weights

    def state_dict(self, *a, **k):  # the frozen weights, reference key names
        return dict(self._state)

    def context(self, device):
        """The native context (packed weights + workspaces) for `device`; created on first use."""
        device = torch.device(device)
        if device.type != 'cuda':
            raise PnpxError(f'UNetDenoiser2D runs on MI355X only (tensor on {device}); there is no CPU path')
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if idx not in self._ctx:
            ctx = ops.Context(torch.device('cuda', idx))
            ctx.load_unet(self._state)
            if self.conv_mode is not None:
                ctx.set_option('conv_mode', self.conv_mode)
            self._ctx[idx] = ctx
        return self._ctx[idx]

    def forward(self, x, sigma):
        # x: [B,1,H,W]; sigma: [B]      (denoiser/base.py:23-32)
        # one dispatcher op; its registered autograd formula is the native VJP wrt x and sigma.  Under autograd the
        # training variant runs: same kernels, and the activations stay in the context's ring for that VJP
        cid = self.context(x.device).cid
        if torch.is_grad_enabled() and (x.requires_grad or sigma.requires_grad):
            return T.call("unet_denoise_train", x, sigma, cid)[0]
        return T.call("unet_denoise", x, sigma, cid)

    def forward_preclamp(self, x, sigma):
        """(clamped, pre-clamp) outputs -- the pre-clamp UNet output is what the parity tests compare."""
        return T.call("unet_denoise_preclamp", x, sigma, self.context(x.device).cid)


class DRUNetDenoiser2D(UNetDenoiser2D):
    """DRUNet (KAIR UNetRes: bias-free, 64-128-256-512 channels, `nb` ResBlocks per scale, strided / transposed 2x2
    convolutions) behind the same call surface as UNetDenoiser2D: `denoiser(x[B,1,H,W], sigma[B]) ->
    clamp(DRUNet(cat[x, sigma*1]), 0, 1)`.  The reference ships only the building blocks
    (tfpnp/pnp/denoiser/models/basicblock.py:61-101,211-227,413-419,437-446) and no checkpoint: a state_dict with KAIR's
    key names (`m_head.weight`, `m_down1.0.res.0.weight`, ... -- what drunet_gray.pth holds) or a path to one is required.
    H and W must be multiples of 8.  Under autograd the registered VJP re-computes the forward natively keeping every
    ResBlock's ReLU output and back-propagates on the same kernels (csrc/drunet.hip::drunet_denoise_backward)."""

    def __init__(self, ckpt_path=None, state_dict=None, nb=4, conv_mode=None):
        """conv_mode: None/0 = fp32 arithmetic throughout (default since r6; csrc/drunet_f32.hip: Winograd / direct fp32 MFMA
        kernels, forward and VJP); 1 = the fast mode: half-split f16 MFMA convolutions (forward and VJP)."""
        torch.nn.Module.__init__(self)
        self.conv_mode = conv_mode
        self.nb = nb
        if state_dict is None:
            if ckpt_path is None:
                raise ValueError('DRUNet: no default ckpt exists, you have to provide a ckpt path or a state_dict')
            state_dict = torch.load(ckpt_path, map_location='cpu')
        self._state = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else torch.as_tensor(v))
                       for k, v in state_dict.items()}
        self._ctx = {}

    def context(self, device):
        device = torch.device(device)
        if device.type != 'cuda':
            raise PnpxError(f'DRUNetDenoiser2D runs on MI355X only (tensor on {device}); there is no CPU path')
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if idx not in self._ctx:
            ctx = ops.Context(torch.device('cuda', idx))
            ctx.load_drunet(self._state, nb=self.nb)
            if self.conv_mode is not None:
                ctx.set_option('conv_mode', self.conv_mode)
            self._ctx[idx] = ctx
        return self._ctx[idx]

    def forward(self, x, sigma):
        # no activation ring for the DRUNet: the op's registered autograd formula is the re-computing native VJP
        return T.call("unet_denoise", x, sigma, self.context(x.device).cid)
