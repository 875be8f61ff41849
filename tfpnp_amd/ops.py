"""Tensor-level wrappers over the C ABI (include/pnpx.h): torch tensors in, torch tensors out.

PyTorch is plumbing here (device memory, streams); every op below is one libpnpx.so call on the caller's
current HIP stream.  All ops require contiguous fp32 tensors on a ROCm device; anything else raises.
"""
import ctypes as C
import itertools
import threading
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import PnpxError, check


_ctx_ids = itertools.count(1)
_ctx_by_id = weakref.WeakValueDictionary()


def context_by_id(cid):
    """The live Context with integer handle `cid` (how contexts travel through torch.ops.pnpx.* schemas)."""
    try:
        return _ctx_by_id[int(cid)]
    except KeyError:
        raise PnpxError(f"no live native context with id {cid}") from None


class Context:
    """One pnpx_ctx: a device, (optionally) packed denoiser weights, and the native workspaces."""

    def __init__(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise PnpxError(f"tfpnp_amd runs on MI355X (ROCm 'cuda' devices) only, got {device}; there is no CPU path")
        self.device = torch.device("cuda", device.index if device.index is not None else torch.cuda.current_device())
        h = C.c_void_p()
        check(_lib.lib().pnpx_ctx_create(self.device.index, C.byref(h)))
        self._h = h
        self._has_weights = False
        self.is_drunet = False
        self._policy = None
        self.cid = next(_ctx_ids)          # integer handle for the dispatcher-registered ops (torch_ops.py)
        _ctx_by_id[self.cid] = self

    @property
    def handle(self):
        if self._h is None:
            raise PnpxError("context already destroyed")
        return self._h

    def load_unet(self, state_dict):
        """state_dict: mapping with the reference's 56 key names -> tensors/ndarrays (any device)."""
        from .synth import unet_param_specs
        chunks = []
        for key, shape in unet_param_specs():
            if key not in state_dict:
                raise PnpxError(f"denoiser state_dict is missing '{key}'")
            v = state_dict[key]
            v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            if tuple(v.shape) != tuple(shape):
                raise PnpxError(f"'{key}' has shape {tuple(v.shape)}, expected {tuple(shape)}")
            chunks.append(np.ascontiguousarray(v, dtype=np.float32).reshape(-1))
        flat = np.concatenate(chunks)
        check(_lib.lib().pnpx_unet_load(self.handle, flat.ctypes.data_as(C.c_void_p), flat.size))
        self._has_weights = True
        self.is_drunet = False

    def load_drunet(self, state_dict, nb=4):
        """state_dict with KAIR's UNetRes / DRUNet key names (synth.drunet_param_specs) -> the context's denoiser."""
        from .synth import drunet_param_specs
        chunks = []
        for key, shape in drunet_param_specs(nb=nb):
            if key not in state_dict:
                raise PnpxError(f"DRUNet state_dict is missing '{key}'")
            v = state_dict[key]
            v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            if tuple(v.shape) != tuple(shape):
                raise PnpxError(f"'{key}' has shape {tuple(v.shape)}, expected {tuple(shape)}")
            chunks.append(np.ascontiguousarray(v, dtype=np.float32).reshape(-1))
        flat = np.concatenate(chunks)
        check(_lib.lib().pnpx_drunet_load(self.handle, flat.ctypes.data_as(C.c_void_p), flat.size, int(nb)))
        self._has_weights = True
        self.is_drunet = True

    def load_policy(self, state_dict, num_inputs, n_det, spi_head=False):
        """state_dict with the reference's ResNetActor_* key names (tfpnp/policy/network.py) -> native actor."""
        from .synth import policy_param_specs
        chunks = []
        for key, shape in policy_param_specs(num_inputs, n_det, spi_head):
            if key not in state_dict:
                raise PnpxError(f"policy state_dict is missing '{key}'")
            v = state_dict[key]
            v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            if tuple(v.shape) != tuple(shape):
                raise PnpxError(f"'{key}' has shape {tuple(v.shape)}, expected {tuple(shape)}")
            chunks.append(np.ascontiguousarray(v, dtype=np.float32).reshape(-1))
        flat = np.concatenate(chunks)
        check(_lib.lib().pnpx_policy_load(self.handle, flat.ctypes.data_as(C.c_void_p), flat.size, int(num_inputs),
                                          int(n_det), int(bool(spi_head))))
        self._policy = (int(num_inputs), int(n_det))

    def set_option(self, key, value):
        """e.g. set_option('conv_mode', 1) selects the fast half-split f16 MFMA convolutions (default 0 = fp32 arithmetic)."""
        check(_lib.lib().pnpx_ctx_set_option(self.handle, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_int(0)
        check(_lib.lib().pnpx_ctx_get_option(self.handle, key.encode(), C.byref(v)))
        return int(v.value)

    def status(self):
        """Raises PnpxError once the half-split range guard has tripped (an earlier call's output was invalid; the
        context has switched itself to conv_mode 0).  Does not synchronise: call it after the stream has drained,
        e.g. right after reading a result back.  Re-arm with set_option('range_guard', 1)."""
        check(_lib.lib().pnpx_ctx_status(self.handle))

    def range_tripped(self):
        """True once the half-split range guard has tripped (see status()); other failures raise.  Like status() it does
        not synchronise -- ask after a point where the stream has drained (PnPEnv.step does, after its one host read)."""
        st = _lib.lib().pnpx_ctx_status(self.handle)
        if st == _lib.PNPX_ERR_RANGE:
            return True
        check(st)
        return False

    def release_train_cache(self):
        """Give the training path's activation ring (raw device memory outside PyTorch's caching allocator) back to the
        device now; it re-grows on the next denoiser forward under autograd.  Outstanding tickets re-compute."""
        check(_lib.lib().pnpx_ctx_set_option(self.handle, b"train_cache_release", 1))

    def reserve(self, B, H, W):
        check(_lib.lib().pnpx_ctx_reserve(self.handle, B, H, W))

    def bytes(self):
        return int(_lib.lib().pnpx_ctx_bytes(self.handle))

    def close(self):
        if self._h is not None:
            _lib.lib().pnpx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}
_default_lock = threading.Lock()


def default_context(device):
    """Weight-less per-device context for the stateless transforms (fft2, cdp, psnr ...)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise PnpxError(f"tfpnp_amd has no CPU path (tensor on {device})")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    with _default_lock:
        if idx not in _default_ctx:
            _default_ctx[idx] = Context(torch.device("cuda", idx))
        return _default_ctx[idx]


# ------------------------------------------------------------------------------------------------- helpers
def _f32(t, name):
    if not isinstance(t, torch.Tensor):
        raise PnpxError(f"{name}: expected a torch.Tensor")
    if t.device.type != "cuda":
        raise PnpxError(f"{name}: tensor is on {t.device}; tfpnp_amd has no CPU path")
    if t.dtype != torch.float32:
        raise PnpxError(f"{name}: expected float32, got {t.dtype}")
    return t.detach().contiguous()


def _mask_u8(mask, name="mask"):
    if mask.device.type != "cuda":
        raise PnpxError(f"{name}: tensor is on {mask.device}; tfpnp_amd has no CPU path")
    m = mask.detach()
    if m.dtype == torch.bool:
        return m.contiguous().view(torch.uint8)
    return (m != 0).contiguous().view(torch.uint8)


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr())


def _params(B, *ps):
    """Hyper-parameter tensors [B,T] (or [B]) -> contiguous fp32 [B,T], common T and row stride."""
    out = []
    T = None
    for i, p in enumerate(ps):
        p = _f32(p, f"parameter {i}")
        if p.dim() == 1:
            p = p.view(-1, 1)
        if p.dim() != 2 or p.shape[0] != B:
            raise PnpxError(f"hyper-parameter {i} must be [B, iter_num] with B={B}, got {tuple(p.shape)}")
        T = p.shape[1] if T is None else T
        if p.shape[1] != T:
            raise PnpxError("hyper-parameters disagree on iter_num")
        out.append(p)
    return out, T


# ------------------------------------------------------------------------------------------------- denoiser
def unet_denoise(ctx, x, sigma, return_preclamp=False):
    x = _f32(x, "x")
    sigma = _f32(sigma, "sigma").reshape(-1)
    if x.dim() != 4 or x.shape[1] != 1:
        raise PnpxError(f"denoiser input must be [B,1,H,W], got {tuple(x.shape)}")
    B, _, H, W = x.shape
    if sigma.numel() != B:
        raise PnpxError(f"sigma must have {B} entries, got {sigma.numel()}")
    out = torch.empty_like(x)
    pre = torch.empty_like(x) if return_preclamp else None
    if B == 0:                       # every item already stopped: empty in, empty out (as the reference's modules)
        return (out, pre) if return_preclamp else out
    with torch.cuda.device(x.device):
        check(_lib.lib().pnpx_unet_denoise(ctx.handle, _p(x), _p(sigma), _p(out), _p(pre) if pre is not None else None,
                                           B, H, W, _stream(x)))
    return (out, pre) if return_preclamp else out


def unet_denoise_train(ctx, x, sigma):
    """Denoiser forward for autograd -> (out [B,1,H,W], ticket): the activations are parked in the context's training
    ring under `ticket` (0: not parked) so that unet_denoise_backward(..., ticket=) need not re-compute them."""
    x = _f32(x, "x")
    sigma = _f32(sigma, "sigma").reshape(-1)
    if x.dim() != 4 or x.shape[1] != 1 or sigma.numel() != x.shape[0]:
        raise PnpxError(f"denoiser input must be [B,1,H,W] with sigma [B], got {tuple(x.shape)} / {tuple(sigma.shape)}")
    B, _, H, W = x.shape
    out = torch.empty_like(x)
    ticket = C.c_ulonglong(0)
    if B:
        with torch.cuda.device(x.device):
            check(_lib.lib().pnpx_unet_denoise_train(ctx.handle, _p(x), _p(sigma), _p(out), B, H, W, C.byref(ticket),
                                                     _stream(x)))
    return out, int(ticket.value)


def unet_denoise_backward(ctx, x, sigma, grad_out, ticket=0):
    """(grad_x [B,1,H,W], grad_sigma [B]) = J^T grad_out of unet_denoise; the forward pass is re-computed natively unless
    `ticket` names activations the context's training ring still holds."""
    x = _f32(x, "x")
    sigma = _f32(sigma, "sigma").reshape(-1)
    grad_out = _f32(grad_out, "grad_out")
    B, _, H, W = x.shape
    if grad_out.shape != x.shape or sigma.numel() != B:
        raise PnpxError("unet_denoise_backward: shape mismatch")
    gx = torch.empty_like(x)
    gs = torch.empty((B,), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        check(_lib.lib().pnpx_unet_denoise_backward_ticket(ctx.handle, _p(x), _p(sigma), _p(grad_out), _p(gx), _p(gs), B, H,
                                                           W, int(ticket), _stream(x)))
    return gx, gs


def policy_forward(ctx, ob):
    """ob [B,C,H,W] -> (probs [B,2], det [B,n_det]) of the loaded actor (eval mode)."""
    ob = _f32(ob, "ob")
    if ob.dim() != 4 or getattr(ctx, "_policy", None) is None or ob.shape[1] != ctx._policy[0]:
        raise PnpxError("policy_forward: no policy loaded or observation has the wrong channel count")
    B, _, H, W = ob.shape
    probs = torch.empty((B, 2), device=ob.device, dtype=torch.float32)
    det = torch.empty((B, ctx._policy[1]), device=ob.device, dtype=torch.float32)
    if B == 0:
        return probs, det
    with torch.cuda.device(ob.device):
        check(_lib.lib().pnpx_policy_forward(ctx.handle, _p(ob), _p(probs), _p(det), B, H, W, _stream(ob)))
    return probs, det


def unet_profile(ctx, x, sigma):
    """[(name, ms, flops)] per kernel launch of one denoiser forward (HIP events on the current stream)."""
    x = _f32(x, "x")
    sigma = _f32(sigma, "sigma").reshape(-1)
    B, _, H, W = x.shape
    out = torch.empty_like(x)
    cap = 1000
    ms = (C.c_float * cap)()
    fl = (C.c_double * cap)()
    names = (C.c_char_p * cap)()
    n = C.c_int(0)
    with torch.cuda.device(x.device):
        check(_lib.lib().pnpx_unet_profile(ctx.handle, _p(x), _p(sigma), _p(out), B, H, W, _stream(x), cap, ms, fl,
                                           names, C.byref(n)))
    return [(names[i].decode(), float(ms[i]), float(fl[i])) for i in range(n.value)]


# ------------------------------------------------------------------------------------------------- transforms
def fft2(x, inverse=False, centered=True, ctx=None):
    x = _f32(x, "data")
    if x.dim() < 3 or x.shape[-1] != 2:
        raise AssertionError("fft2 expects [..., H, W, 2]")  # the reference asserts data.size(-1) == 2
    H, W = x.shape[-3], x.shape[-2]
    n_img = x.numel() // (H * W * 2)
    out = torch.empty_like(x)
    ctx = ctx or default_context(x.device)
    with torch.cuda.device(x.device):
        check(_lib.lib().pnpx_fft2(ctx.handle, _p(x), _p(out), n_img, H, W, int(inverse), int(centered), _stream(x)))
    return out


def cdp_forward(x, mask, ctx=None):
    mask = _f32(mask, "mask")
    if mask.shape[-1] != 2:
        raise AssertionError("mask must be complex [..., 2]")
    x = _f32(x, "data")
    if x.dim() == 4:
        x = torch.stack([x, torch.zeros_like(x)], -1)
    B, S, H, W, _ = mask.shape
    out = torch.empty_like(mask)
    ctx = ctx or default_context(x.device)
    with torch.cuda.device(x.device):
        check(_lib.lib().pnpx_cdp_forward(ctx.handle, _p(x), _p(mask), _p(out), B, S, H, W, _stream(x)))
    return out


def cdp_backward(y, mask, ctx=None):
    mask = _f32(mask, "mask")
    y = _f32(y, "data")
    B, S, H, W, _ = mask.shape
    out = torch.empty((B, 1, H, W, 2), device=y.device, dtype=torch.float32)
    ctx = ctx or default_context(y.device)
    with torch.cuda.device(y.device):
        check(_lib.lib().pnpx_cdp_backward(ctx.handle, _p(y), _p(mask), _p(out), B, S, H, W, _stream(y)))
    return out


def spi_inverse(ztilde, K1, K, mu, ctx=None):
    ztilde = _f32(ztilde, "ztilde")
    B, _, H, W = ztilde.shape
    K1 = _f32(K1, "K1").expand_as(ztilde).contiguous()
    K = _f32(K, "K").reshape(B, -1)[:, 0].contiguous()
    mu = _f32(mu, "mu").reshape(B, -1)[:, 0].contiguous()
    out = torch.empty_like(ztilde)
    ctx = ctx or default_context(ztilde.device)
    with torch.cuda.device(ztilde.device):
        check(_lib.lib().pnpx_spi_inverse(ctx.handle, _p(ztilde), _p(K1), _p(K), _p(mu), _p(out), B, H, W,
                                          _stream(ztilde)))
    return out


def psnr(output, gt, ctx=None):
    output = _f32(output, "output")
    gt = _f32(gt, "gt")
    B = output.shape[0]
    if gt.numel() != output.numel():
        raise PnpxError("psnr: output and gt differ in size")
    out = torch.empty((B, 1), device=output.device, dtype=torch.float32)
    if B == 0:
        return out
    n = output.numel() // B
    ctx = ctx or default_context(output.device)
    with torch.cuda.device(output.device):
        check(_lib.lib().pnpx_psnr(ctx.handle, _p(output), _p(gt), _p(out), B, n, _stream(output)))
    return out


# ------------------------------------------------------------------------------------------------- solver loops
def _vars(variables, nvar, complex_):
    v = _f32(variables, "variables")
    want = 5 if complex_ else 4
    if v.dim() != want or v.shape[1] != nvar or (complex_ and v.shape[-1] != 2):
        raise PnpxError(f"solver state must be [B,{nvar},H,W{',2' if complex_ else ''}], got {tuple(v.shape)}")
    return v


def _csmri_common(fn_name, nvar, ctx, variables, y0, mask, params, iter_num):
    v = _vars(variables, nvar, True)
    B, _, H, W, _ = v.shape
    y0 = _f32(y0, "y0")
    m = _mask_u8(mask)
    if y0.numel() != B * H * W * 2 or m.numel() != B * H * W:
        raise PnpxError("y0/mask do not match the state's [B,H,W]")
    ps, T = _params(B, *params)
    if iter_num is not None:
        if iter_num > T:
            raise PnpxError(f"iter_num {iter_num} exceeds the {T} hyper-parameter columns provided")
        T = iter_num
    out = torch.empty_like(v)
    if B == 0:
        return out
    stride = ps[0].shape[1]
    fn = getattr(_lib.lib(), fn_name)
    with torch.cuda.device(v.device):
        check(fn(ctx.handle, _p(v), _p(out), _p(y0), _p(m), *[_p(p) for p in ps], stride, B, H, W, T, _stream(v)))
    return out


def csmri_admm(ctx, variables, y0, mask, sigma_d, mu, iter_num=None):
    return _csmri_common("pnpx_csmri_admm", 3, ctx, variables, y0, mask, (sigma_d, mu), iter_num)


def csmri_admm_train(ctx, variables, y0, mask, sigma_d, mu, iter_num=None, _entry="pnpx_csmri_admm_train", _nvar=3,
                     _saved_per=3):
    """pnpx_csmri_admm_train: the ADMM forward that also returns what its VJP needs -> (next state, saved [3*T*B*H*W],
    ticket of the context's activation cache (int, 0 = not cached))."""
    v = _vars(variables, _nvar, True)
    B, _, H, W, _ = v.shape
    y0, m = _f32(y0, "y0"), _mask_u8(mask)
    if y0.numel() != B * H * W * 2 or m.numel() != B * H * W:
        raise PnpxError("y0/mask do not match the state's [B,H,W]")
    ps, T = _params(B, sigma_d, mu)
    if iter_num is not None:
        if iter_num > T:
            raise PnpxError(f"iter_num {iter_num} exceeds the {T} hyper-parameter columns provided")
        T = iter_num
    out = torch.empty_like(v)
    saved = torch.empty(_saved_per * T * B * H * W, dtype=torch.float32, device=v.device)
    if B == 0:
        return out, saved, 0
    ticket = C.c_ulonglong(0)
    with torch.cuda.device(v.device):
        check(getattr(_lib.lib(), _entry)(ctx.handle, _p(v), _p(out), _p(y0), _p(m), _p(ps[0]), _p(ps[1]),
                                          ps[0].shape[1], B, H, W, T, _p(saved), C.byref(ticket), _stream(v)))
    return out, saved, int(ticket.value)


def csmri_hqs_train(ctx, variables, y0, mask, sigma_d, mu, iter_num=None):
    """pnpx_csmri_hqs_train: HQSSolver_CSMRI.forward for autograd (state [B,2,H,W,2]); same returns as csmri_admm_train."""
    return csmri_admm_train(ctx, variables, y0, mask, sigma_d, mu, iter_num, "pnpx_csmri_hqs_train", 2)


def csmri_hqs_backward(ctx, y0, mask, sigma_d, mu, saved, grad_out, iter_num=None, ticket=0):
    """pnpx_csmri_hqs_backward -> (grad variables [B,2,H,W,2], grad sigma_d [B,T], grad mu [B,T])."""
    return csmri_admm_backward(ctx, y0, mask, sigma_d, mu, saved, grad_out, iter_num, ticket, "pnpx_csmri_hqs_backward", 2)


def csmri_pg_train(ctx, variables, y0, mask, sigma_d, tau, iter_num=None):
    """pnpx_csmri_pg_train: PGSolver_CSMRI.forward for autograd (state [B,1,H,W,2]); saved = 2*T*B*H*W floats."""
    return csmri_admm_train(ctx, variables, y0, mask, sigma_d, tau, iter_num, "pnpx_csmri_pg_train", 1, 2)


def csmri_pg_backward(ctx, y0, mask, sigma_d, tau, saved, grad_out, iter_num=None, ticket=0):
    """pnpx_csmri_pg_backward -> (grad x [B,1,H,W,2], grad sigma_d [B,T], grad tau [B,T])."""
    return csmri_admm_backward(ctx, y0, mask, sigma_d, tau, saved, grad_out, iter_num, ticket, "pnpx_csmri_pg_backward", 1, 2)


def csmri_admm_backward(ctx, y0, mask, sigma_d, mu, saved, grad_out, iter_num=None, ticket=0,
                        _entry="pnpx_csmri_admm_backward", _nvar=3, _saved_per=3):
    """pnpx_csmri_admm_backward -> (grad variables [B,3,H,W,2], grad sigma_d [B,T], grad mu [B,T])."""
    g = _vars(grad_out, _nvar, True)
    B, _, H, W, _ = g.shape
    y0, m = _f32(y0, "y0"), _mask_u8(mask)
    ps, T = _params(B, sigma_d, mu)
    T = T if iter_num is None else iter_num
    if saved.numel() != _saved_per * T * B * H * W:
        raise PnpxError("saved does not belong to a forward of this shape / iteration count")
    gin = torch.empty_like(g)
    gs = torch.zeros(T, B, dtype=torch.float32, device=g.device)
    gm = torch.zeros(T, B, dtype=torch.float32, device=g.device)
    if B and T:
        work = torch.empty(3 * B * H * W, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            check(getattr(_lib.lib(), _entry)(ctx.handle, _p(y0), _p(m), _p(ps[0]), _p(ps[1]), ps[0].shape[1],
                                              _p(saved), _p(g), _p(gin), _p(gs), _p(gm), _p(work), B, H, W, T,
                                              int(ticket), _stream(g)))
    elif B:
        gin.copy_(g)
    return gin, gs.t().contiguous(), gm.t().contiguous()


def _csmri_train3(entry, nvar, per, ctx, variables, y0, mask, params, iter_num):
    """Training forward of a CS-MRI solver with three hyper-parameter rows -> (next state, saved [per*T*B*H*W], ticket)."""
    v = _vars(variables, nvar, True)
    B, _, H, W, _ = v.shape
    y0, m = _f32(y0, "y0"), _mask_u8(mask)
    if y0.numel() != B * H * W * 2 or m.numel() != B * H * W:
        raise PnpxError("y0/mask do not match the state's [B,H,W]")
    ps, T = _params(B, *params)
    if iter_num is not None:
        if iter_num > T:
            raise PnpxError(f"iter_num {iter_num} exceeds the {T} hyper-parameter columns provided")
        T = iter_num
    out = torch.empty_like(v)
    saved = torch.empty(per * T * B * H * W, dtype=torch.float32, device=v.device)
    if B == 0:
        return out, saved, 0
    ticket = C.c_ulonglong(0)
    with torch.cuda.device(v.device):
        check(getattr(_lib.lib(), entry)(ctx.handle, _p(v), _p(out), _p(y0), _p(m), _p(ps[0]), _p(ps[1]), _p(ps[2]),
                                         ps[0].shape[1], B, H, W, T, _p(saved), C.byref(ticket), _stream(v)))
    return out, saved, int(ticket.value)


def _csmri_backward3(entry, nvar, per, ctx, y0, mask, params, saved, grad_out, iter_num, ticket):
    """Fused VJP of the same -> (grad variables, three hyper-parameter gradients, each [B,T])."""
    g = _vars(grad_out, nvar, True)
    B, _, H, W, _ = g.shape
    y0, m = _f32(y0, "y0"), _mask_u8(mask)
    ps, T = _params(B, *params)
    T = T if iter_num is None else iter_num
    if saved.numel() != per * T * B * H * W:
        raise PnpxError("saved does not belong to a forward of this shape / iteration count")
    gin = torch.empty_like(g)
    g0, g1, g2 = (torch.zeros(T, B, dtype=torch.float32, device=g.device) for _ in range(3))
    if B and T:
        work = torch.empty(4 * B * H * W, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            check(getattr(_lib.lib(), entry)(ctx.handle, _p(y0), _p(m), _p(ps[0]), _p(ps[1]), _p(ps[2]), ps[0].shape[1],
                                             _p(saved), _p(g), _p(gin), _p(g0), _p(g1), _p(g2), _p(work), B, H, W, T,
                                             int(ticket), _stream(g)))
    elif B:
        gin.copy_(g)
    return gin, g0.t().contiguous(), g1.t().contiguous(), g2.t().contiguous()


def csmri_apg_train(ctx, variables, y0, mask, sigma_d, tau, beta, iter_num=None):
    """pnpx_csmri_apg_train: APGSolver_CSMRI.forward for autograd -> (next state [B,2,H,W,2], saved [4*T*B*H*W], ticket)."""
    return _csmri_train3("pnpx_csmri_apg_train", 2, 4, ctx, variables, y0, mask, (sigma_d, tau, beta), iter_num)


def csmri_apg_backward(ctx, y0, mask, sigma_d, tau, beta, saved, grad_out, iter_num=None, ticket=0):
    """pnpx_csmri_apg_backward -> (grad variables [B,2,H,W,2], grad sigma_d, grad tau, grad beta, each [B,T])."""
    return _csmri_backward3("pnpx_csmri_apg_backward", 2, 4, ctx, y0, mask, (sigma_d, tau, beta), saved, grad_out, iter_num,
                            ticket)


def csmri_redadmm_train(ctx, variables, y0, mask, sigma_d, mu, lamda, iter_num=None):
    """pnpx_csmri_redadmm_train: REDADMMSolver_CSMRI.forward for autograd -> (next state [B,3,H,W,2], saved [7*T*B*H*W],
    ticket)."""
    return _csmri_train3("pnpx_csmri_redadmm_train", 3, 7, ctx, variables, y0, mask, (sigma_d, mu, lamda), iter_num)


def csmri_redadmm_backward(ctx, y0, mask, sigma_d, mu, lamda, saved, grad_out, iter_num=None, ticket=0):
    """pnpx_csmri_redadmm_backward -> (grad variables [B,3,H,W,2], grad sigma_d, grad mu, grad lamda, each [B,T])."""
    return _csmri_backward3("pnpx_csmri_redadmm_backward", 3, 7, ctx, y0, mask, (sigma_d, mu, lamda), saved, grad_out,
                            iter_num, ticket)


def csmri_hqs(ctx, variables, y0, mask, sigma_d, mu, iter_num=None):
    return _csmri_common("pnpx_csmri_hqs", 2, ctx, variables, y0, mask, (sigma_d, mu), iter_num)


def csmri_pg(ctx, variables, y0, mask, sigma_d, tau, iter_num=None):
    return _csmri_common("pnpx_csmri_pg", 1, ctx, variables, y0, mask, (sigma_d, tau), iter_num)


def csmri_apg(ctx, variables, y0, mask, sigma_d, tau, beta, iter_num=None):
    return _csmri_common("pnpx_csmri_apg", 2, ctx, variables, y0, mask, (sigma_d, tau, beta), iter_num)


def csmri_redadmm(ctx, variables, y0, mask, sigma_d, mu, lamda, iter_num=None):
    return _csmri_common("pnpx_csmri_redadmm", 3, ctx, variables, y0, mask, (sigma_d, mu, lamda), iter_num)


def pr_iadmm(ctx, variables, y0, mask, sigma_d, mu, tau, iter_num=None):
    v = _vars(variables, 3, True)
    B, _, H, W, _ = v.shape
    y0 = _f32(y0, "y0")
    mask = _f32(mask, "mask")
    S = mask.shape[1]
    if tuple(mask.shape) != (B, S, H, W, 2) or tuple(y0.shape) != (B, S, H, W):
        raise PnpxError("pr_iadmm: y0 must be [B,S,H,W] and mask [B,S,H,W,2]")
    ps, T = _params(B, sigma_d, mu, tau)
    T = T if iter_num is None else iter_num
    out = torch.empty_like(v)
    if B == 0:
        return out
    with torch.cuda.device(v.device):
        check(_lib.lib().pnpx_pr_iadmm(ctx.handle, _p(v), _p(out), _p(y0), _p(mask), *[_p(p) for p in ps],
                                       ps[0].shape[1], B, S, H, W, T, _stream(v)))
    return out


def _pr_args(variables, y0, mask):
    v = _vars(variables, 3, True)
    B, _, H, W, _ = v.shape
    y0, mask = _f32(y0, "y0"), _f32(mask, "mask")
    S = mask.shape[1]
    if tuple(mask.shape) != (B, S, H, W, 2) or tuple(y0.shape) != (B, S, H, W):
        raise PnpxError("pr_iadmm: y0 must be [B,S,H,W] and mask [B,S,H,W,2]")
    return v, y0, mask, B, S, H, W


def pr_iadmm_train(ctx, variables, y0, mask, sigma_d, mu, tau, iter_num=None):
    """pnpx_pr_iadmm_train: IADMMSolver_PR.forward for autograd -> (next state [B,3,H,W,2], saved [(2S+5)*T*B*H*W], ticket)."""
    v, y0, mask, B, S, H, W = _pr_args(variables, y0, mask)
    ps, T = _params(B, sigma_d, mu, tau)
    if iter_num is not None:
        if iter_num > T:
            raise PnpxError(f"iter_num {iter_num} exceeds the {T} hyper-parameter columns provided")
        T = iter_num
    out = torch.empty_like(v)
    saved = torch.empty((2 * S + 5) * T * B * H * W, dtype=torch.float32, device=v.device)
    if B == 0:
        return out, saved, 0
    ticket = C.c_ulonglong(0)
    with torch.cuda.device(v.device):
        check(_lib.lib().pnpx_pr_iadmm_train(ctx.handle, _p(v), _p(out), _p(y0), _p(mask), *[_p(p) for p in ps],
                                             ps[0].shape[1], B, S, H, W, T, _p(saved), C.byref(ticket), _stream(v)))
    return out, saved, int(ticket.value)


def pr_iadmm_backward(ctx, y0, mask, sigma_d, mu, tau, saved, grad_out, iter_num=None, ticket=0):
    """pnpx_pr_iadmm_backward -> (grad variables [B,3,H,W,2], grad sigma_d, grad mu, grad tau, each [B,T])."""
    g, y0, mask, B, S, H, W = _pr_args(grad_out, y0, mask)
    ps, T = _params(B, sigma_d, mu, tau)
    T = T if iter_num is None else iter_num
    if saved.numel() != (2 * S + 5) * T * B * H * W:
        raise PnpxError("saved does not belong to a forward of this shape / iteration count")
    gin = torch.empty_like(g)
    g0, g1, g2 = (torch.zeros(T, B, dtype=torch.float32, device=g.device) for _ in range(3))
    if B and T:
        work = torch.empty(4 * B * H * W, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            check(_lib.lib().pnpx_pr_iadmm_backward(ctx.handle, _p(y0), _p(mask), *[_p(p) for p in ps], ps[0].shape[1],
                                                    _p(saved), _p(g), _p(gin), _p(g0), _p(g1), _p(g2), _p(work), B, S, H, W,
                                                    T, int(ticket), _stream(g)))
    elif B:
        gin.copy_(g)
    return gin, g0.t().contiguous(), g1.t().contiguous(), g2.t().contiguous()


def spi_admm(ctx, variables, x0, Kmap, sigma_d, mu, iter_num=None):
    v = _vars(variables, 3, False)
    B, _, H, W = v.shape
    x0 = _f32(x0, "x0")
    Kmap = _f32(Kmap, "K")
    if x0.numel() != B * H * W or Kmap.numel() != B * H * W:
        raise PnpxError("spi_admm: x0 and K must be [B,1,H,W]")
    ps, T = _params(B, sigma_d, mu)
    T = T if iter_num is None else iter_num
    out = torch.empty_like(v)
    if B == 0:
        return out
    with torch.cuda.device(v.device):
        check(_lib.lib().pnpx_spi_admm(ctx.handle, _p(v), _p(out), _p(x0), _p(Kmap), *[_p(p) for p in ps],
                                       ps[0].shape[1], B, H, W, T, _stream(v)))
    return out


def _train_T(B, params, iter_num):
    ps, T = _params(B, *params)
    if iter_num is not None:
        if iter_num > T:
            raise PnpxError(f"iter_num {iter_num} exceeds the {T} hyper-parameter columns provided")
        T = iter_num
    return ps, T


def _hyper_grads(T, B, k, device):
    return [torch.zeros(T, B, dtype=torch.float32, device=device) for _ in range(k)]


def spi_admm_train(ctx, variables, x0, Kmap, sigma_d, mu, iter_num=None):
    """pnpx_spi_admm_train: ADMMSolver_SPI.forward for autograd -> (next state [B,3,H,W], saved [2*T*B*H*W], ticket)."""
    v = _vars(variables, 3, False)
    B, _, H, W = v.shape
    x0, Kmap = _f32(x0, "x0"), _f32(Kmap, "K")
    if x0.numel() != B * H * W or Kmap.numel() != B * H * W:
        raise PnpxError("spi_admm: x0 and K must be [B,1,H,W]")
    ps, T = _train_T(B, (sigma_d, mu), iter_num)
    out = torch.empty_like(v)
    saved = torch.empty(2 * T * B * H * W, dtype=torch.float32, device=v.device)
    if B == 0:
        return out, saved, 0
    ticket = C.c_ulonglong(0)
    with torch.cuda.device(v.device):
        check(_lib.lib().pnpx_spi_admm_train(ctx.handle, _p(v), _p(out), _p(x0), _p(Kmap), *[_p(p) for p in ps],
                                             ps[0].shape[1], B, H, W, T, _p(saved), C.byref(ticket), _stream(v)))
    return out, saved, int(ticket.value)


def spi_admm_backward(ctx, x0, Kmap, sigma_d, mu, saved, grad_out, iter_num=None, ticket=0):
    """pnpx_spi_admm_backward -> (grad variables [B,3,H,W], grad sigma_d, grad mu, each [B,T])."""
    g = _vars(grad_out, 3, False)
    B, _, H, W = g.shape
    x0, Kmap = _f32(x0, "x0"), _f32(Kmap, "K")
    ps, T = _params(B, sigma_d, mu)
    T = T if iter_num is None else iter_num
    if saved.numel() != 2 * T * B * H * W:
        raise PnpxError("saved does not belong to a forward of this shape / iteration count")
    gin = torch.empty_like(g)
    gs = _hyper_grads(T, B, 2, g.device)
    if B and T:
        work = torch.empty(3 * B * H * W, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            check(_lib.lib().pnpx_spi_admm_backward(ctx.handle, _p(x0), _p(Kmap), _p(ps[0]), _p(ps[1]), ps[0].shape[1],
                                                    _p(saved), _p(g), _p(gin), _p(gs[0]), _p(gs[1]), _p(work), B, H, W, T,
                                                    int(ticket), _stream(g)))
    elif B:
        gin.copy_(g)
    return (gin, *[x.t().contiguous() for x in gs])


# ------------------------------------------------------------------------------------------------- CT
def radon_det_count(R):
    return int(_lib.lib().pnpx_radon_det_count(int(R)))


def radon_forward(img, n_view, ctx=None):
    img = _f32(img, "img")
    B, _, R, R2 = img.shape
    if R != R2:
        raise PnpxError("radon_forward: square images only")
    out = torch.empty((B, 1, n_view, radon_det_count(R)), device=img.device, dtype=torch.float32)
    ctx = ctx or default_context(img.device)
    with torch.cuda.device(img.device):
        check(_lib.lib().pnpx_radon_forward(ctx.handle, _p(img), _p(out), B, R, n_view, _stream(img)))
    return out


def radon_backprojection(sino, R, ctx=None):
    sino = _f32(sino, "sino")
    B, _, V, det = sino.shape
    if det != radon_det_count(R):
        raise PnpxError(f"sinogram has {det} detectors, resolution {R} needs {radon_det_count(R)}")
    out = torch.empty((B, 1, R, R), device=sino.device, dtype=torch.float32)
    ctx = ctx or default_context(sino.device)
    with torch.cuda.device(sino.device):
        check(_lib.lib().pnpx_radon_backprojection(ctx.handle, _p(sino), _p(out), B, R, V, _stream(sino)))
    return out


def _ct_sino(y0, B, R, n_view):
    """The native CT loops index the sinogram with a (n_view, det) stride: a mismatching tensor must not get through."""
    y0 = _f32(y0, "y0")
    want = (B, 1, int(n_view), radon_det_count(R))
    if tuple(y0.shape) != want:
        raise PnpxError(f"y0: sinogram of shape {tuple(y0.shape)} does not match (B, 1, n_view, det) = {want}")
    return y0


def ct_iadmm(ctx, variables, y0, n_view, opnorm, sigma_d, mu, tau, iter_num=None):
    v = _vars(variables, 3, False)
    B, _, R, _ = v.shape
    y0 = _ct_sino(y0, B, R, n_view)
    ps, T = _params(B, sigma_d, mu, tau)
    T = T if iter_num is None else iter_num
    out = torch.empty_like(v)
    if B == 0:
        return out
    with torch.cuda.device(v.device):
        check(_lib.lib().pnpx_ct_iadmm(ctx.handle, _p(v), _p(out), _p(y0), int(n_view), float(opnorm),
                                       *[_p(p) for p in ps], ps[0].shape[1], B, R, T, _stream(v)))
    return out


def ct_pg(ctx, variables, y0, n_view, opnorm, sigma_d, tau, iter_num=None):
    v = _vars(variables, 1, False)
    B, _, R, _ = v.shape
    y0 = _ct_sino(y0, B, R, n_view)
    ps, T = _params(B, sigma_d, tau)
    T = T if iter_num is None else iter_num
    out = torch.empty_like(v)
    if B == 0:
        return out
    with torch.cuda.device(v.device):
        check(_lib.lib().pnpx_ct_pg(ctx.handle, _p(v), _p(out), _p(y0), int(n_view), float(opnorm),
                                    *[_p(p) for p in ps], ps[0].shape[1], B, R, T, _stream(v)))
    return out


def ct_iadmm_train(ctx, variables, y0, n_view, opnorm, sigma_d, mu, tau, iter_num=None):
    """pnpx_ct_iadmm_train: IADMMSolver_CT.forward for autograd -> (next state [B,3,R,R], saved [3*T*B*R*R], ticket)."""
    v = _vars(variables, 3, False)
    B, _, R, _ = v.shape
    y0 = _ct_sino(y0, B, R, n_view)
    ps, T = _train_T(B, (sigma_d, mu, tau), iter_num)
    out = torch.empty_like(v)
    saved = torch.empty(3 * T * B * R * R, dtype=torch.float32, device=v.device)
    if B == 0:
        return out, saved, 0
    ticket = C.c_ulonglong(0)
    with torch.cuda.device(v.device):
        check(_lib.lib().pnpx_ct_iadmm_train(ctx.handle, _p(v), _p(out), _p(y0), int(n_view), float(opnorm),
                                             *[_p(p) for p in ps], ps[0].shape[1], B, R, T, _p(saved), C.byref(ticket),
                                             _stream(v)))
    return out, saved, int(ticket.value)


def ct_iadmm_backward(ctx, n_view, opnorm, sigma_d, mu, tau, saved, grad_out, iter_num=None, ticket=0):
    """pnpx_ct_iadmm_backward -> (grad variables [B,3,R,R], grad sigma_d, grad mu, grad tau, each [B,T])."""
    g = _vars(grad_out, 3, False)
    B, _, R, _ = g.shape
    ps, T = _params(B, sigma_d, mu, tau)
    T = T if iter_num is None else iter_num
    if saved.numel() != 3 * T * B * R * R:
        raise PnpxError("saved does not belong to a forward of this shape / iteration count")
    gin = torch.empty_like(g)
    gs = _hyper_grads(T, B, 3, g.device)
    if B and T:
        work = torch.empty(6 * B * R * R + 2 * int(n_view), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            check(_lib.lib().pnpx_ct_iadmm_backward(ctx.handle, int(n_view), float(opnorm), *[_p(p) for p in ps],
                                                    ps[0].shape[1], _p(saved), _p(g), _p(gin), *[_p(x) for x in gs], _p(work),
                                                    B, R, T, int(ticket), _stream(g)))
    elif B:
        gin.copy_(g)
    return (gin, *[x.t().contiguous() for x in gs])


def ct_pg_train(ctx, variables, y0, n_view, opnorm, sigma_d, tau, iter_num=None):
    """pnpx_ct_pg_train: PGSolver_CT.forward for autograd -> (next x [B,1,R,R], saved [2*T*B*R*R], ticket)."""
    v = _vars(variables, 1, False)
    B, _, R, _ = v.shape
    y0 = _ct_sino(y0, B, R, n_view)
    ps, T = _train_T(B, (sigma_d, tau), iter_num)
    out = torch.empty_like(v)
    saved = torch.empty(2 * T * B * R * R, dtype=torch.float32, device=v.device)
    if B == 0:
        return out, saved, 0
    ticket = C.c_ulonglong(0)
    with torch.cuda.device(v.device):
        check(_lib.lib().pnpx_ct_pg_train(ctx.handle, _p(v), _p(out), _p(y0), int(n_view), float(opnorm),
                                          *[_p(p) for p in ps], ps[0].shape[1], B, R, T, _p(saved), C.byref(ticket),
                                          _stream(v)))
    return out, saved, int(ticket.value)


def ct_pg_backward(ctx, n_view, opnorm, sigma_d, tau, saved, grad_out, iter_num=None, ticket=0):
    """pnpx_ct_pg_backward -> (grad x [B,1,R,R], grad sigma_d, grad tau, each [B,T])."""
    g = _vars(grad_out, 1, False)
    B, _, R, _ = g.shape
    ps, T = _params(B, sigma_d, tau)
    T = T if iter_num is None else iter_num
    if saved.numel() != 2 * T * B * R * R:
        raise PnpxError("saved does not belong to a forward of this shape / iteration count")
    gin = torch.empty_like(g)
    gs = _hyper_grads(T, B, 2, g.device)
    if B and T:
        work = torch.empty(3 * B * R * R + 1 + 2 * int(n_view), dtype=torch.float32, device=g.device)   # +1: float2 table at an even offset
        with torch.cuda.device(g.device):
            check(_lib.lib().pnpx_ct_pg_backward(ctx.handle, int(n_view), float(opnorm), *[_p(p) for p in ps], ps[0].shape[1],
                                                 _p(saved), _p(g), _p(gin), *[_p(x) for x in gs], _p(work), B, R, T,
                                                 int(ticket), _stream(g)))
    elif B:
        gin.copy_(g)
    return (gin, *[x.t().contiguous() for x in gs])


# ------------------------------------------------------------------------------------- episode orchestration (env.hip)
def _dense(t, name):
    if not isinstance(t, torch.Tensor) or t.device.type != "cuda":
        raise PnpxError(f"{name}: expected a tensor on a ROCm device (there is no CPU path)")
    return t if t.is_contiguous() else t.contiguous()


def _idx64(idx, name):
    if not isinstance(idx, torch.Tensor) or idx.dtype != torch.int64 or idx.device.type != "cuda":
        raise PnpxError(f"{name}: expected an int64 tensor on a ROCm device")
    return idx if idx.is_contiguous() else idx.contiguous()


def _rows_call(fn, srcs, dsts, small, idx, n_rows, dev):
    n = len(srcs)
    S = (C.c_void_p * n)(*[s.data_ptr() for s in srcs])
    D = (C.c_void_p * n)(*[d.data_ptr() for d in dsts])
    RB = (C.c_size_t * n)(*[(t.numel() // max(t.shape[0], 1)) * t.element_size() for t in small])
    ctx = default_context(dev)
    with torch.cuda.device(dev):
        check(fn(ctx.handle, n, S, D, RB, _p(idx), int(n_rows), _stream(idx)))


def rows_gather(tensors, idx, n_rows):
    """[t[idx[:n_rows]] for t in tensors] in ONE launch (tfpnp/env/base.py:162-166).  idx: device int64 (capacity >=
    n_rows); dtypes are preserved (fp32 state, bool masks)."""
    idx = _idx64(idx, "idx")
    srcs = [_dense(t, "tensor") for t in tensors]
    outs = [torch.empty((n_rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in srcs]
    if n_rows > 0 and srcs:
        _rows_call(_lib.lib().pnpx_rows_gather, srcs, outs, outs, idx, n_rows, idx.device)
    return outs


def rows_scatter(values, targets, idx, n_rows):
    """targets[k][idx[:n_rows]] = values[k] for every k, in ONE launch (tfpnp/env/base.py:171-172).  In place."""
    idx = _idx64(idx, "idx")
    vals = []
    for v, t in zip(values, targets):
        if not t.is_contiguous() or t.device.type != "cuda":
            raise PnpxError("rows_scatter: targets must be contiguous device tensors")
        v = _dense(v, "value")
        if v.dtype != t.dtype or tuple(v.shape[1:]) != tuple(t.shape[1:]) or v.shape[0] < n_rows:
            raise PnpxError(f"rows_scatter: value {tuple(v.shape)}/{v.dtype} does not fit target {tuple(t.shape)}/{t.dtype}")
        vals.append(v)
    if n_rows > 0 and vals:
        _rows_call(_lib.lib().pnpx_rows_scatter, vals, list(targets), vals, idx, n_rows, idx.device)


def live_compact(idx_left, idx_stop, n):
    """(idx_left[:n][idx_stop == 0] in a buffer of capacity n, number of survivors as a Python int).
    Synchronises the current stream: the one host read of an env step (tfpnp/env/base.py:180-182)."""
    idx_left = _idx64(idx_left, "idx_left")
    idx_stop = _idx64(idx_stop, "idx_stop")
    if idx_stop.numel() < n or idx_left.numel() < n:
        raise PnpxError("live_compact: idx_left / idx_stop shorter than n")
    out = torch.empty((max(n, 1),), dtype=torch.int64, device=idx_left.device)
    cnt = C.c_int(0)
    ctx = default_context(idx_left.device)
    with torch.cuda.device(idx_left.device):
        check(_lib.lib().pnpx_live_compact(ctx.handle, _p(idx_left), _p(idx_stop), int(n), _p(out), C.byref(cnt),
                                           _stream(idx_left)))
    return out, int(cnt.value)


_PACK_KINDS = {"raw": 0, "real": 1, "channel": 2, "u8": 3}


def policy_ob_pack(entries, idx=None, n_rows=None):
    """Fused get_policy_ob (tasks/*/env.py): entries = [(tensor, 'raw' | 'real' | 'channel'), ...] in channel order;
    'raw' fp32/bool [B,c,H,W], 'real' / 'channel' complex [B,c,H,W,2].  idx/n_rows: gather rows idx[:n_rows] on the fly."""
    srcs, kinds, chans = [], [], []
    H = W = None
    for t, kind in entries:
        t = _dense(t, "observation entry")
        if kind == "raw":
            if t.dim() != 4:
                raise PnpxError(f"policy_ob_pack: raw entry must be [B,c,H,W], got {tuple(t.shape)}")
            if t.dtype in (torch.bool, torch.uint8):
                k = "u8"
            elif t.dtype == torch.float32:
                k = "raw"
            else:
                raise PnpxError(f"policy_ob_pack: unsupported dtype {t.dtype}")
        elif kind in ("real", "channel"):
            if t.dim() != 5 or t.shape[-1] != 2 or t.dtype != torch.float32:
                raise PnpxError(f"policy_ob_pack: {kind} entry must be fp32 [B,c,H,W,2], got {tuple(t.shape)}")
            k = kind
        else:
            raise PnpxError(f"policy_ob_pack: unknown kind {kind}")
        if H is None:
            H, W = t.shape[2], t.shape[3]
        elif (t.shape[2], t.shape[3]) != (H, W):
            raise PnpxError("policy_ob_pack: entries disagree on H, W")
        srcs.append(t)
        kinds.append(_PACK_KINDS[k])
        chans.append(t.shape[1])
    n = len(srcs)
    if n == 0:
        raise PnpxError("policy_ob_pack: no entries")
    rows = srcs[0].shape[0] if n_rows is None else int(n_rows)
    C_out = sum(c * (2 if k == 2 else 1) for c, k in zip(chans, kinds))
    out = torch.empty((rows, C_out, H, W), dtype=torch.float32, device=srcs[0].device)
    if rows == 0:
        return out
    if idx is not None:
        idx = _idx64(idx, "idx")
    S = (C.c_void_p * n)(*[s.data_ptr() for s in srcs])
    K = (C.c_int * n)(*kinds)
    CH = (C.c_int * n)(*chans)
    ctx = default_context(out.device)
    with torch.cuda.device(out.device):
        check(_lib.lib().pnpx_policy_ob_pack(ctx.handle, n, S, K, CH, _p(idx) if idx is not None else None, rows, H, W,
                                             _p(out), _stream(out)))
    return out
