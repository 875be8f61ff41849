"""`torch.ops.pnpx.*` -- the native hot path registered with the PyTorch dispatcher (torch.library custom ops).

Every operator is a thin shim over one C-ABI entry of libpnpx.so (include/pnpx.h, bound in tfpnp_amd/ops.py): schema
+ a ROCm ("cuda") implementation + a fake-tensor (meta) implementation, so the ops trace under FakeTensorMode /
torch.compile / torch.export, and autograd formulas for the ops the training path differentiates (denoiser VJP, the
unitary FFTs, the Radon pair, PSNR).  There is no CPU implementation: dispatching a CPU tensor raises
NotImplementedError from the dispatcher itself.

A native context (packed denoiser weights + workspaces) cannot travel through an operator schema, so ops take an
integer handle `ctx` (tfpnp_amd.ops.Context.cid; 0 = the weight-less default context of the tensors' device).
`iter_num = -1` means "all columns of the hyper-parameter tensors" (the reference's iter_num=None).
"""
from typing import Tuple

import torch
from torch import Tensor

from . import ops

_lib_def = torch.library.custom_op


def _ctx(cid, t):
    return ops.default_context(t.device) if cid == 0 else ops.context_by_id(cid)


def _pin(node, cid, t):
    """setup_context helper: the autograd node keeps the native context ALIVE until its backward has run (the id -> context
    table holds weak references only; a denoiser collected between forward and backward must not strand the node)."""
    node.cid = cid
    node.native = _ctx(cid, t)


def _it(iter_num):
    return None if iter_num < 0 else iter_num


# ------------------------------------------------------------------------------------------------- denoiser prox
@_lib_def("pnpx::unet_denoise", mutates_args=(), device_types="cuda")
def unet_denoise(x: Tensor, sigma: Tensor, ctx: int) -> Tensor:
    """UNetDenoiser2D.forward (tfpnp/pnp/denoiser/base.py:23-32): x [B,1,H,W], sigma [B] -> [B,1,H,W]."""
    return ops.unet_denoise(_ctx(ctx, x), x, sigma)


@unet_denoise.register_fake
def _(x, sigma, ctx):
    return torch.empty_like(x, memory_format=torch.contiguous_format)


@_lib_def("pnpx::unet_denoise_preclamp", mutates_args=(), device_types="cuda")
def unet_denoise_preclamp(x: Tensor, sigma: Tensor, ctx: int) -> Tuple[Tensor, Tensor]:
    """(clamped, pre-clamp) denoiser outputs."""
    return ops.unet_denoise(_ctx(ctx, x), x, sigma, return_preclamp=True)


@unet_denoise_preclamp.register_fake
def _(x, sigma, ctx):
    e = lambda: torch.empty_like(x, memory_format=torch.contiguous_format)
    return e(), e()


@_lib_def("pnpx::unet_denoise_backward", mutates_args=(), device_types="cuda")
def unet_denoise_backward(x: Tensor, sigma: Tensor, grad_out: Tensor, ctx: int) -> Tuple[Tensor, Tensor]:
    """VJP of unet_denoise wrt (x, sigma); the forward pass is re-computed inside (gradient checkpointing)."""
    return ops.unet_denoise_backward(_ctx(ctx, x), x, sigma.reshape(-1), grad_out)


@unet_denoise_backward.register_fake
def _(x, sigma, grad_out, ctx):
    return (torch.empty_like(x, memory_format=torch.contiguous_format),
            torch.empty((x.shape[0],), dtype=x.dtype, device=x.device))


def _denoise_setup(ctx, inputs, output):
    x, sigma, cid = inputs
    ctx.save_for_backward(x, sigma)
    _pin(ctx, cid, x)


def _denoise_bwd(ctx, g):
    x, sigma = ctx.saved_tensors
    gx, gs = torch.ops.pnpx.unet_denoise_backward(x, sigma, g.contiguous(), ctx.cid)
    return gx, gs.view_as(sigma), None


unet_denoise.register_autograd(_denoise_bwd, setup_context=_denoise_setup)


@_lib_def("pnpx::unet_denoise_train", mutates_args=(), device_types="cuda")
def unet_denoise_train(x: Tensor, sigma: Tensor, ctx: int) -> Tuple[Tensor, Tensor]:
    """unet_denoise for autograd graphs: also parks the activations in the context's training ring; the second output is
    the ticket (one-element int64 CPU tensor) its VJP presents to skip the re-computation (pnpx_unet_denoise_train)."""
    out, ticket = ops.unet_denoise_train(_ctx(ctx, x), x, sigma)
    return out, torch.tensor([ticket], dtype=torch.int64)


@unet_denoise_train.register_fake
def _(x, sigma, ctx):
    return torch.empty_like(x, memory_format=torch.contiguous_format), torch.empty((1,), dtype=torch.int64)


@_lib_def("pnpx::unet_denoise_backward_ticket", mutates_args=(), device_types="cuda")
def unet_denoise_backward_ticket(x: Tensor, sigma: Tensor, grad_out: Tensor, ticket: Tensor, ctx: int) -> Tuple[Tensor, Tensor]:
    """VJP of unet_denoise_train: served from the training ring if it still holds `ticket`, else by re-computation."""
    return ops.unet_denoise_backward(_ctx(ctx, x), x, sigma.reshape(-1), grad_out, ticket=int(ticket[0]))


@unet_denoise_backward_ticket.register_fake
def _(x, sigma, grad_out, ticket, ctx):
    return (torch.empty_like(x, memory_format=torch.contiguous_format),
            torch.empty((x.shape[0],), dtype=x.dtype, device=x.device))


def _denoise_train_setup(ctx, inputs, output):
    x, sigma, cid = inputs
    _pin(ctx, cid, x)
    ctx.save_for_backward(x, sigma, output[1])


def _denoise_train_bwd(ctx, g, _g_ticket):
    x, sigma, ticket = ctx.saved_tensors
    gx, gs = torch.ops.pnpx.unet_denoise_backward_ticket(x, sigma, g.contiguous(), ticket, ctx.cid)
    return gx, gs.view_as(sigma), None


unet_denoise_train.register_autograd(_denoise_train_bwd, setup_context=_denoise_train_setup)


@_lib_def("pnpx::policy_forward", mutates_args=(), device_types="cuda")
def policy_forward(ob: Tensor, ctx: int) -> Tuple[Tensor, Tensor]:
    """ResNetActorBase.forward in eval mode (tfpnp/policy/network.py:129-147): ob [B,C,H,W] -> (probs [B,2], det [B,n])."""
    return ops.policy_forward(ops.context_by_id(ctx), ob)


@policy_forward.register_fake
def _(ob, ctx):
    n_det = ops.context_by_id(ctx)._policy[1]
    return (torch.empty((ob.shape[0], 2), dtype=ob.dtype, device=ob.device),
            torch.empty((ob.shape[0], n_det), dtype=ob.dtype, device=ob.device))


# ------------------------------------------------------------------------------------------------- transforms
@_lib_def("pnpx::fft2", mutates_args=(), device_types="cuda")
def fft2(x: Tensor, inverse: bool, centered: bool) -> Tensor:
    """fft2 / ifft2 (tfpnp/utils/transforms.py:68-103), orthonormal, over dims (-3,-2) of [...,H,W,2]."""
    return ops.fft2(x, inverse=inverse, centered=centered)


@fft2.register_fake
def _(x, inverse, centered):
    return torch.empty_like(x, memory_format=torch.contiguous_format)


def _fft_setup(ctx, inputs, output):
    _, ctx.inverse, ctx.centered = inputs


def _fft_bwd(ctx, g):      # orthonormal transform (+ index permutations): adjoint == inverse
    return torch.ops.pnpx.fft2(g.contiguous(), not ctx.inverse, ctx.centered), None, None


fft2.register_autograd(_fft_bwd, setup_context=_fft_setup)


@_lib_def("pnpx::cdp_forward", mutates_args=(), device_types="cuda")
def cdp_forward(x: Tensor, mask: Tensor) -> Tensor:
    """cdp_forward (transforms.py:282-301): x [B,1,H,W,2], mask [B,S,H,W,2] -> [B,S,H,W,2]."""
    return ops.cdp_forward(x, mask)


@cdp_forward.register_fake
def _(x, mask):
    return torch.empty_like(mask, memory_format=torch.contiguous_format)


@_lib_def("pnpx::cdp_backward", mutates_args=(), device_types="cuda")
def cdp_backward(y: Tensor, mask: Tensor) -> Tensor:
    """cdp_backward (transforms.py:304-320): y, mask [B,S,H,W,2] -> [B,1,H,W,2]."""
    return ops.cdp_backward(y, mask)


@cdp_backward.register_fake
def _(y, mask):
    B, S, H, W, _ = y.shape
    return torch.empty((B, 1, H, W, 2), dtype=y.dtype, device=y.device)


@_lib_def("pnpx::spi_inverse", mutates_args=(), device_types="cuda")
def spi_inverse(ztilde: Tensor, K1: Tensor, K: Tensor, mu: Tensor) -> Tensor:
    """spi_inverse (transforms.py:404-439)."""
    return ops.spi_inverse(ztilde, K1, K, mu)


@spi_inverse.register_fake
def _(ztilde, K1, K, mu):
    return torch.empty_like(ztilde, memory_format=torch.contiguous_format)


@_lib_def("pnpx::psnr", mutates_args=(), device_types="cuda")
def psnr(output: Tensor, gt: Tensor) -> Tensor:
    """torch_psnr (tfpnp/env/base.py:237-242): [B,...] x 2 -> [B,1]."""
    return ops.psnr(output, gt)


@psnr.register_fake
def _(output, gt):
    return torch.empty((output.shape[0], 1), dtype=output.dtype, device=output.device)


def _psnr_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _psnr_bwd(ctx, g):
    # d psnr_b / d out = -(20 / ln 10) * (clamp(out) - gt) / sse_b on the closed interval [0, 1] (torch.clamp's VJP)
    output, gt = ctx.saved_tensors
    B = output.shape[0]
    o = output.reshape(B, -1)
    diff = o.clamp(0, 1) - gt.reshape(B, -1)
    sse = (diff * diff).sum(dim=1, keepdim=True)
    inside = (o >= 0) & (o <= 1)
    go = (-20.0 / 2.302585092994046) * g.reshape(B, 1) * diff / sse * inside
    return go.view_as(output), None


psnr.register_autograd(_psnr_bwd, setup_context=_psnr_setup)


@_lib_def("pnpx::radon_forward", mutates_args=(), device_types="cuda")
def radon_forward(img: Tensor, n_view: int) -> Tensor:
    """Parallel-beam Radon transform standing in for torch_radon (transforms.py:465-508): [B,1,R,R] -> [B,1,V,det]."""
    return ops.radon_forward(img, n_view)


@radon_forward.register_fake
def _(img, n_view):
    return torch.empty((img.shape[0], 1, n_view, ops.radon_det_count(img.shape[-1])), dtype=img.dtype, device=img.device)


@_lib_def("pnpx::radon_backprojection", mutates_args=(), device_types="cuda")
def radon_backprojection(sino: Tensor, R: int) -> Tensor:
    return ops.radon_backprojection(sino, R)


@radon_backprojection.register_fake
def _(sino, R):
    return torch.empty((sino.shape[0], 1, R, R), dtype=sino.dtype, device=sino.device)


# each is the other's VJP (the unmatched ray-/pixel-driven pair torch_radon also uses)
radon_forward.register_autograd(
    lambda ctx, g: (torch.ops.pnpx.radon_backprojection(g.contiguous(), ctx.R), None),
    setup_context=lambda ctx, inputs, output: setattr(ctx, "R", inputs[0].shape[-1]))
radon_backprojection.register_autograd(
    lambda ctx, g: (torch.ops.pnpx.radon_forward(g.contiguous(), ctx.V), None),
    setup_context=lambda ctx, inputs, output: setattr(ctx, "V", inputs[0].shape[2]))


# ------------------------------------------------------------------------------------------------- solver loops
# ------------------------------------------------------------------------------------------------- solver training ops
def _solver_train_ops(name, aux, hypers, train_fn, backward_fn, saved_per_pixel, bwd_takes_aux=True):
    """Register `pnpx::<name>_train` (+ `_backward` and the autograd formula) for a solver whose native training entry follows the
    common contract: (variables, *aux, *hyper-parameters, iter_num, ctx) -> (next state, saved, ticket) -- the same result as the
    inference op plus what the native VJP replays (`saved`: per-iteration denoiser inputs and data-step intermediates, include/pnpx.h)
    and the ticket of the context's activation cache (a one-element int64 CPU tensor).  This is what PnPEnv.forward reaches under
    autograd (tfpnp/env/base.py:193-206).  `aux` = [(name, schema type)]; non-tensor aux entries (view count, operator norm)
    travel on the autograd node; `saved_per_pixel` = floats per pixel and iteration (a number, or a function of the aux tuple).
    The backward op returns gradients for the hyper-parameter columns that were iterated; the rest of [B, action_pack] gets zeros."""
    n_aux, n_h = len(aux), len(hypers)
    aux_sig = [f"{t} {n}" for n, t in aux]
    hyp_sig = [f"Tensor {h}" for h in hypers]
    fwd_schema = "(" + ", ".join(["Tensor variables"] + aux_sig + hyp_sig + ["int iter_num", "int ctx"]) + ") -> (Tensor, Tensor, Tensor)"
    bwd_schema = ("(" + ", ".join(aux_sig + hyp_sig + ["Tensor saved", "Tensor ticket", "Tensor grad_out", "int iter_num", "int ctx"]) +
                  ") -> (" + ", ".join(["Tensor"] * (1 + n_h)) + ")")

    def n_iter(h0, iter_num):
        return (h0.shape[1] if h0.dim() == 2 else 1) if iter_num < 0 else iter_num

    def fwd(*a):
        variables, auxv, hv, (iter_num, cid) = a[0], a[1:1 + n_aux], a[1 + n_aux:1 + n_aux + n_h], a[-2:]
        out, saved, ticket = train_fn(_ctx(cid, variables), variables, *auxv, *hv, _it(iter_num))
        return out, saved, torch.tensor([ticket], dtype=torch.int64)

    def fwd_fake(*a):
        variables, hv, iter_num = a[0], a[1 + n_aux:1 + n_aux + n_h], a[-2]
        n_px = variables.numel() // variables.shape[1] // (2 if variables.dim() == 5 else 1)
        per = saved_per_pixel(a[1:1 + n_aux]) if callable(saved_per_pixel) else saved_per_pixel
        return (torch.empty_like(variables, memory_format=torch.contiguous_format),
                torch.empty((per * n_iter(hv[0], iter_num) * n_px,), dtype=variables.dtype, device=variables.device),
                torch.empty((1,), dtype=torch.int64))

    def bwd(*a):
        auxv, hv = a[:n_aux], a[n_aux:n_aux + n_h]
        saved, ticket, grad_out, iter_num, cid = a[n_aux + n_h:]
        lead = auxv if bwd_takes_aux else tuple(v for v, (_, t) in zip(auxv, aux) if t != "Tensor")
        return backward_fn(_ctx(cid, grad_out), *lead, *hv, saved, grad_out, _it(iter_num), ticket=int(ticket[0]))

    def bwd_fake(*a):
        hv, grad_out, iter_num = a[n_aux:n_aux + n_h], a[-3], a[-2]
        e = lambda: torch.empty((grad_out.shape[0], n_iter(hv[0], iter_num)), dtype=grad_out.dtype, device=grad_out.device)
        return (torch.empty_like(grad_out, memory_format=torch.contiguous_format), *[e() for _ in range(n_h)])

    op_f = _lib_def(f"pnpx::{name}_train", fwd, mutates_args=(), device_types="cuda", schema=fwd_schema)
    op_f.register_fake(fwd_fake)
    op_b = _lib_def(f"pnpx::{name}_backward", bwd, mutates_args=(), device_types="cuda", schema=bwd_schema)
    op_b.register_fake(bwd_fake)
    tensor_aux = [k for k, (_, t) in enumerate(aux) if t == "Tensor"]

    def setup(ctx, inputs, output):
        auxv, hv = inputs[1:1 + n_aux], inputs[1 + n_aux:1 + n_aux + n_h]
        ctx.iter_num, cid = inputs[-2:]
        _pin(ctx, cid, inputs[0])
        ctx.plain_aux = {k: v for k, v in enumerate(auxv) if k not in tensor_aux}
        ctx.save_for_backward(*[auxv[k] for k in tensor_aux], *hv, output[1], output[2])

    def backward(ctx, g_out, _g_saved, _g_ticket):
        t = ctx.saved_tensors
        auxv = [None] * n_aux
        for j, k in enumerate(tensor_aux):
            auxv[k] = t[j]
        for k, v in ctx.plain_aux.items():
            auxv[k] = v
        hv = t[len(tensor_aux):len(tensor_aux) + n_h]
        res = getattr(torch.ops.pnpx, f"{name}_backward")(*auxv, *hv, t[-2], t[-1], g_out.contiguous(), ctx.iter_num, ctx.cid)

        def like(g, p):
            full = torch.zeros_like(p)
            if p.numel():
                full.view(p.shape[0], -1)[:, :g.shape[1]] = g
            return full

        return (res[0], *[None] * n_aux, *[like(g, p) for g, p in zip(res[1:], hv)], None, None)

    op_f.register_autograd(backward, setup_context=setup)
    return op_f, op_b


_MRI_AUX = [("y0", "Tensor"), ("mask", "Tensor")]
csmri_admm_train, csmri_admm_backward = _solver_train_ops(
    "csmri_admm", _MRI_AUX, ("sigma_d", "mu"), ops.csmri_admm_train, ops.csmri_admm_backward, 3)
csmri_hqs_train, csmri_hqs_backward = _solver_train_ops(
    "csmri_hqs", _MRI_AUX, ("sigma_d", "mu"), ops.csmri_hqs_train, ops.csmri_hqs_backward, 3)
csmri_pg_train, csmri_pg_backward = _solver_train_ops(
    "csmri_pg", _MRI_AUX, ("sigma_d", "tau"), ops.csmri_pg_train, ops.csmri_pg_backward, 2)
csmri_apg_train, csmri_apg_backward = _solver_train_ops(
    "csmri_apg", _MRI_AUX, ("sigma_d", "tau", "beta"), ops.csmri_apg_train, ops.csmri_apg_backward, 4)
csmri_redadmm_train, csmri_redadmm_backward = _solver_train_ops(
    "csmri_redadmm", _MRI_AUX, ("sigma_d", "mu", "lamda"), ops.csmri_redadmm_train, ops.csmri_redadmm_backward, 7)
pr_iadmm_train, pr_iadmm_backward = _solver_train_ops(
    "pr_iadmm", _MRI_AUX, ("sigma_d", "mu", "tau"), ops.pr_iadmm_train, ops.pr_iadmm_backward,
    lambda aux: 2 * aux[1].shape[1] + 5)        # mask [B,S,H,W,2]: S k-space images per iteration
spi_admm_train, spi_admm_backward = _solver_train_ops(
    "spi_admm", [("x0", "Tensor"), ("Kmap", "Tensor")], ("sigma_d", "mu"), ops.spi_admm_train, ops.spi_admm_backward, 2)
ct_iadmm_train, ct_iadmm_backward = _solver_train_ops(
    "ct_iadmm", [("y0", "Tensor"), ("n_view", "int"), ("opnorm", "float")], ("sigma_d", "mu", "tau"), ops.ct_iadmm_train,
    ops.ct_iadmm_backward, 3, bwd_takes_aux=False)
ct_pg_train, ct_pg_backward = _solver_train_ops(
    "ct_pg", [("y0", "Tensor"), ("n_view", "int"), ("opnorm", "float")], ("sigma_d", "tau"), ops.ct_pg_train, ops.ct_pg_backward, 2,
    bwd_takes_aux=False)



def _same(variables, *a):
    return torch.empty_like(variables, memory_format=torch.contiguous_format)


@_lib_def("pnpx::csmri_admm", mutates_args=(), device_types="cuda")
def csmri_admm(variables: Tensor, y0: Tensor, mask: Tensor, sigma_d: Tensor, mu: Tensor, iter_num: int, ctx: int) -> Tensor:
    """ADMMSolver_CSMRI.forward (tasks/csmri/solver.py:29-57), all inner iterations in one native call."""
    return ops.csmri_admm(_ctx(ctx, variables), variables, y0, mask, sigma_d, mu, _it(iter_num))


csmri_admm.register_fake(_same)


@_lib_def("pnpx::csmri_hqs", mutates_args=(), device_types="cuda")
def csmri_hqs(variables: Tensor, y0: Tensor, mask: Tensor, sigma_d: Tensor, mu: Tensor, iter_num: int, ctx: int) -> Tensor:
    """HQSSolver_CSMRI.forward (tasks/csmri/solver.py:64-89)."""
    return ops.csmri_hqs(_ctx(ctx, variables), variables, y0, mask, sigma_d, mu, _it(iter_num))


csmri_hqs.register_fake(_same)


@_lib_def("pnpx::csmri_pg", mutates_args=(), device_types="cuda")
def csmri_pg(variables: Tensor, y0: Tensor, mask: Tensor, sigma_d: Tensor, tau: Tensor, iter_num: int, ctx: int) -> Tensor:
    """PGSolver_CSMRI.forward (tasks/csmri/solver.py:96-120)."""
    return ops.csmri_pg(_ctx(ctx, variables), variables, y0, mask, sigma_d, tau, _it(iter_num))


csmri_pg.register_fake(_same)


@_lib_def("pnpx::csmri_apg", mutates_args=(), device_types="cuda")
def csmri_apg(variables: Tensor, y0: Tensor, mask: Tensor, sigma_d: Tensor, tau: Tensor, beta: Tensor, iter_num: int,
              ctx: int) -> Tensor:
    """APGSolver_CSMRI.forward (tasks/csmri/solver.py:127-165)."""
    return ops.csmri_apg(_ctx(ctx, variables), variables, y0, mask, sigma_d, tau, beta, _it(iter_num))


csmri_apg.register_fake(_same)


@_lib_def("pnpx::csmri_redadmm", mutates_args=(), device_types="cuda")
def csmri_redadmm(variables: Tensor, y0: Tensor, mask: Tensor, sigma_d: Tensor, mu: Tensor, lamda: Tensor, iter_num: int,
                  ctx: int) -> Tensor:
    """REDADMMSolver_CSMRI.forward (tasks/csmri/solver.py:172-204)."""
    return ops.csmri_redadmm(_ctx(ctx, variables), variables, y0, mask, sigma_d, mu, lamda, _it(iter_num))


csmri_redadmm.register_fake(_same)


@_lib_def("pnpx::pr_iadmm", mutates_args=(), device_types="cuda")
def pr_iadmm(variables: Tensor, y0: Tensor, mask: Tensor, sigma_d: Tensor, mu: Tensor, tau: Tensor, iter_num: int,
             ctx: int) -> Tensor:
    """IADMMSolver_PR.forward (tasks/pr/solver.py:37-76)."""
    return ops.pr_iadmm(_ctx(ctx, variables), variables, y0, mask, sigma_d, mu, tau, _it(iter_num))


pr_iadmm.register_fake(_same)


@_lib_def("pnpx::spi_admm", mutates_args=(), device_types="cuda")
def spi_admm(variables: Tensor, x0: Tensor, Kmap: Tensor, sigma_d: Tensor, mu: Tensor, iter_num: int, ctx: int) -> Tensor:
    """ADMMSolver_SPI.forward (tasks/spi/solver.py:17-52)."""
    return ops.spi_admm(_ctx(ctx, variables), variables, x0, Kmap, sigma_d, mu, _it(iter_num))


spi_admm.register_fake(_same)


@_lib_def("pnpx::ct_iadmm", mutates_args=(), device_types="cuda")
def ct_iadmm(variables: Tensor, y0: Tensor, n_view: int, opnorm: float, sigma_d: Tensor, mu: Tensor, tau: Tensor,
             iter_num: int, ctx: int) -> Tensor:
    """IADMMSolver_CT.forward (tasks/ct/solver.py:17-53)."""
    return ops.ct_iadmm(_ctx(ctx, variables), variables, y0, n_view, opnorm, sigma_d, mu, tau, _it(iter_num))


ct_iadmm.register_fake(_same)


@_lib_def("pnpx::ct_pg", mutates_args=(), device_types="cuda")
def ct_pg(variables: Tensor, y0: Tensor, n_view: int, opnorm: float, sigma_d: Tensor, tau: Tensor, iter_num: int,
          ctx: int) -> Tensor:
    """PGSolver_CT.forward (tasks/ct/solver.py:61-87)."""
    return ops.ct_pg(_ctx(ctx, variables), variables, y0, n_view, opnorm, sigma_d, tau, _it(iter_num))


ct_pg.register_fake(_same)

def call(name, *args):
    """torch.ops.pnpx.<name>(*args) with the package's error contract: tensors that are not on a ROCm device raise
    PnpxError (there is no CPU kernel to dispatch to) instead of the dispatcher's NotImplementedError."""
    for a in args:
        if isinstance(a, Tensor) and not a.is_cuda:
            raise ops.PnpxError(f"pnpx::{name}: tensor on {a.device}; tfpnp_amd runs on MI355X only, there is no CPU path")
    return getattr(torch.ops.pnpx, name)(*args)


ALL_OPS = ("unet_denoise", "unet_denoise_preclamp", "unet_denoise_backward", "unet_denoise_train",
           "unet_denoise_backward_ticket", "policy_forward", "fft2", "cdp_forward",
           "cdp_backward", "spi_inverse", "psnr", "radon_forward", "radon_backprojection", "csmri_admm", "csmri_admm_train",
           "csmri_admm_backward", "csmri_hqs",
           "csmri_pg", "csmri_apg", "csmri_redadmm", "pr_iadmm", "spi_admm", "ct_iadmm", "ct_pg")
