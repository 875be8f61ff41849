"""GPU tests of the dispatcher registration (tfpnp_amd/torch_ops.py): `torch.ops.pnpx.*` exist, carry schemas, fake
implementations and autograd formulas (torch.library.opcheck), and the package's own classes go through them."""
import numpy as np
import pytest
import torch

from tests.golden_inputs import denoiser_inputs
from tfpnp_amd import synth

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def g(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def test_every_op_is_registered():
    from tfpnp_amd import torch_ops as T
    for name in T.ALL_OPS:
        op = getattr(torch.ops.pnpx, name)
        assert "pnpx::" + name in str(op.default._schema)


def test_opcheck_transforms():
    from tfpnp_amd import torch_ops  # noqa: F401
    x = torch.randn(2, 1, 16, 32, 2, device=dev(), requires_grad=True)
    for inv, cen in ((False, True), (True, False)):
        torch.library.opcheck(torch.ops.pnpx.fft2, (x, inv, cen))
    img = torch.rand(2, 1, 32, 32, device=dev(), requires_grad=True)
    torch.library.opcheck(torch.ops.pnpx.radon_forward, (img, 12))
    from tfpnp_amd import ops
    sino = torch.rand(2, 1, 12, ops.radon_det_count(32), device=dev(), requires_grad=True)
    torch.library.opcheck(torch.ops.pnpx.radon_backprojection, (sino, 32))
    out = torch.rand(3, 1, 16, 16, device=dev(), requires_grad=True)
    gt = torch.rand(3, 1, 16, 16, device=dev())
    torch.library.opcheck(torch.ops.pnpx.psnr, (out, gt))
    m = torch.randn(2, 4, 16, 16, 2, device=dev())
    torch.library.opcheck(torch.ops.pnpx.cdp_forward, (torch.randn(2, 1, 16, 16, 2, device=dev()), m))
    torch.library.opcheck(torch.ops.pnpx.cdp_backward, (torch.randn(2, 4, 16, 16, 2, device=dev()), m))


def test_opcheck_denoiser_and_solver(unet_params):
    from tfpnp_amd.pnp import UNetDenoiser2D
    den = UNetDenoiser2D(state_dict=unet_params)
    cid = den.context(dev()).cid
    x, s = denoiser_inputs(2, 32, 32, 5)
    x, s = g(x).requires_grad_(True), g(s).requires_grad_(True)
    # the VJP is piecewise (LeakyReLU / max-pool / clamp kinks): opcheck compares eager with AOT-traced runs of the SAME
    # kernels, which are deterministic, so exact agreement is expected
    torch.library.opcheck(torch.ops.pnpx.unet_denoise, (x, s, cid))
    torch.library.opcheck(torch.ops.pnpx.unet_denoise_preclamp, (x.detach(), s.detach(), cid))
    d = synth.make_csmri_batch(2, 32, 32, ratio=4, seed=5)
    a = synth.make_actions(2)[0]
    v0 = torch.cat([g(d["x0"]), g(d["x0"]), torch.zeros_like(g(d["x0"]))], 1)
    torch.library.opcheck(torch.ops.pnpx.csmri_admm, (v0, g(d["y0"]), g(d["mask"]), g(a["sigma_d"]), g(a["mu"]), -1, cid))
    # the differentiable solver op: schema, fake tensors, and its registered (native) VJP under eager and AOT autograd
    lv, ls, lm = (t_.clone().requires_grad_(True) for t_ in (v0, g(a["sigma_d"]), g(a["mu"])))
    # (their ticket output is a fresh number per call by design, so the eager-vs-AOT output comparison does not apply;
    #  schema, fake-tensor and autograd-registration checks do, and the gradients are compared in test_gpu_backward.py)
    parts = ("test_schema", "test_faketensor", "test_autograd_registration")
    torch.library.opcheck(torch.ops.pnpx.csmri_admm_train, (lv, g(d["y0"]), g(d["mask"]), ls, lm, 3, cid), test_utils=parts)
    torch.library.opcheck(torch.ops.pnpx.unet_denoise_train, (x, s, cid), test_utils=parts)
    out, ticket = torch.ops.pnpx.unet_denoise_train(x, s, cid)
    assert ticket.device.type == "cpu" and int(ticket[0]) > 0
    assert (out - torch.ops.pnpx.unet_denoise(x, s, cid)).abs().max() < 1e-6
    gx, gs = torch.autograd.grad(out.sum(), (x, s))
    gx2, gs2 = torch.ops.pnpx.unet_denoise_backward(x.detach(), s.detach(), torch.ones_like(out), cid)   # re-computation
    assert torch.equal(gx, gx2) and torch.equal(gs, gs2)


def test_fake_tensor_propagation_and_eager_compile(unet_params):
    from torch._subclasses.fake_tensor import FakeTensorMode
    from tfpnp_amd.pnp import UNetDenoiser2D
    den = UNetDenoiser2D(state_dict=unet_params)
    cid = den.context(dev()).cid
    with FakeTensorMode():
        x = torch.empty(4, 1, 64, 64, device="cuda")
        s = torch.empty(4, device="cuda")
        y = torch.ops.pnpx.unet_denoise(x, s, cid)
        k = torch.ops.pnpx.fft2(torch.empty(4, 1, 64, 64, 2, device="cuda"), False, True)
        r = torch.ops.pnpx.radon_forward(x, 30)
    assert y.shape == (4, 1, 64, 64) and k.shape == (4, 1, 64, 64, 2) and r.shape[:3] == (4, 1, 30)

    def step(x, s):
        return torch.ops.pnpx.unet_denoise(x, s, cid) * 2.0

    x, s = denoiser_inputs(2, 32, 32, 6)
    x, s = g(x), g(s)
    compiled = torch.compile(step, backend="eager", fullgraph=True)     # no graph break at the custom op
    assert torch.equal(compiled(x, s), step(x, s))


def test_package_classes_dispatch_through_torch_ops(unet_params, monkeypatch):
    """UNetDenoiser2D / the solvers / transforms call torch.ops.pnpx.*, not the ctypes layer directly."""
    from tfpnp_amd import torch_ops as T
    from tfpnp_amd.pnp import UNetDenoiser2D
    from tfpnp_amd.tasks.csmri import ADMMSolver_CSMRI
    from tfpnp_amd.utils import transforms
    seen = []
    real = T.call
    monkeypatch.setattr(T, "call", lambda name, *a: (seen.append(name), real(name, *a))[1])
    den = UNetDenoiser2D(state_dict=unet_params)
    x, s = denoiser_inputs(2, 32, 32, 7)
    den(g(x), g(s))
    transforms.fft2(torch.randn(1, 1, 16, 16, 2, device=dev()))
    d = synth.make_csmri_batch(2, 32, 32, ratio=4, seed=5)
    a = synth.make_actions(2)[0]
    sol = ADMMSolver_CSMRI(den)
    sol((sol.reset({"x0": g(d["x0"])}), (g(d["y0"]), g(d["mask"]))), (g(a["sigma_d"]), g(a["mu"])))
    assert seen == ["unet_denoise", "fft2", "csmri_admm"]
