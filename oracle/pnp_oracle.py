"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU (PyTorch fp32, modern torch.fft API) restatement of the reference's PnP hot path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module, and only as the checker / the timed CPU baseline.  tfpnp_amd/ never imports it.

Pinning: every function here is checked against outputs of the real reference
(imported in the build container through oracle/ref_shim.py) by
tests/test_oracle_golden.py using the fixtures under tests/golden/ written by
oracle/make_goldens.py.  Exception: CT (Radon).  The reference delegates it to the
third-party CUDA package torch_radon (matteo-ronchetti/torch-radon, un-vendored, no
version pinned anywhere in the reference; v1.0.0-era API per
tfpnp/utils/transforms.py:465-481), which cannot be built or run here, and the
reference holds no test vectors for it => "CT: parity unpinned".  The Radon pair below
is this project's own documented discretisation (see radon_forward/radon_backprojection).

All file:line citations are into /root/reference.
Layouts follow the reference: complex tensors are real tensors with a trailing dim of 2.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- complex helpers
def real2complex(x):
    """tfpnp/utils/transforms.py:12-13"""
    return torch.stack([x, torch.zeros_like(x)], dim=-1)


def complex2real(x):
    """tfpnp/utils/transforms.py:16-17"""
    return x[..., 0]


def complex_abs(x):
    """tfpnp/utils/transforms.py:106-118"""
    return (x ** 2).sum(dim=-1).sqrt()


def complex_mul(a, b):
    """tfpnp/utils/transforms.py:260-270"""
    return torch.stack((a[..., 0] * b[..., 0] - a[..., 1] * b[..., 1],
                        a[..., 0] * b[..., 1] + a[..., 1] * b[..., 0]), -1)


def conjugate(x):
    """tfpnp/utils/transforms.py:273-274"""
    return torch.stack([x[..., 0], -x[..., 1]], -1)


def _c(x):
    return torch.view_as_complex(x.contiguous())


def _r(x):
    return torch.view_as_real(x)


def fft2c(x):
    """Centered orthonormal 2-D FFT over dims (-3,-2) of [...,H,W,2].
    tfpnp/utils/transforms.py:68-84 (ifftshift -> legacy torch.fft(.,2,normalized=True) -> fftshift;
    shifts by (n+1)//2 and n//2, transforms.py:232-257)."""
    c = _c(x)
    c = torch.fft.ifftshift(c, dim=(-2, -1))
    c = torch.fft.fft2(c, dim=(-2, -1), norm="ortho")
    c = torch.fft.fftshift(c, dim=(-2, -1))
    return _r(c)


def ifft2c(x):
    """tfpnp/utils/transforms.py:87-103"""
    c = _c(x)
    c = torch.fft.ifftshift(c, dim=(-2, -1))
    c = torch.fft.ifft2(c, dim=(-2, -1), norm="ortho")
    c = torch.fft.fftshift(c, dim=(-2, -1))
    return _r(c)


def cdp_forward(data, mask):
    """A x = FFT_ortho(mask_s * x), un-centered.  tfpnp/utils/transforms.py:282-301"""
    if data.dim() == 4:
        data = real2complex(data)
    S = mask.shape[1]
    x = data.repeat(1, S, 1, 1, 1)
    md = complex_mul(x, mask)
    return _r(torch.fft.fft2(_c(md), dim=(-2, -1), norm="ortho"))


def cdp_backward(data, mask):
    """A^H y = mean_s(conj(mask_s) * IFFT_ortho(y_s)).  tfpnp/utils/transforms.py:304-320"""
    i = _r(torch.fft.ifft2(_c(data), dim=(-2, -1), norm="ortho"))
    return complex_mul(i, conjugate(mask)).mean(1, keepdim=True)


def spi_inverse(ztilde, K1, K, mu):
    """Poisson prox by 10-step bisection.  tfpnp/utils/transforms.py:404-439
    (same masked-assignment semantics, written with torch.where)."""
    K0 = K ** 2 - K1
    is0 = (K1 == 0)
    z_lin = ztilde - (K0 / mu)
    done = is0.expand_as(ztilde).clone()
    bmin = 1e-5 * torch.ones_like(ztilde)
    bmax = 1.1 * torch.ones_like(ztilde)
    bave = (bmin + bmax) / 2.0
    for _ in range(10):
        tmp = K1 / (torch.exp(bave) - 1) - mu * bave - K0 + mu * ztilde
        live = ~done
        pos = (tmp > 0) & live
        neg = (tmp < 0) & live
        zero = (tmp == 0) & live
        done = done | zero
        live = ~done
        bmin = torch.where(pos, bave, bmin)
        bmax = torch.where(neg, bave, bmax)
        bave = torch.where(live, (bmin + bmax) / 2.0, bave)
    z = torch.where(K1 != 0, bave, z_lin)
    return torch.clamp(z, 0.0, 1.0)


# ----------------------------------------------------------------------------- denoiser
def _to_t(params):
    return {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))) for k, v in params.items()}


def _block(x, p, name):
    """ConvBlock = 3 x [Conv2d 3x3 pad 1 + bias, LeakyReLU(0.2)].  models/unet.py:8-31"""
    for j in range(3):
        x = F.conv2d(x, p[f"{name}.conv-{j}.conv2d.weight"], p[f"{name}.conv-{j}.conv2d.bias"], padding=1)
        x = F.leaky_relu(x, 0.2)
    return x


def _up(x1, x2, p, name):
    """up.forward: bilinear x2 align_corners=True, pad to skip size, cat([skip, up]).  models/unet.py:105-121"""
    x1 = F.interpolate(x1, scale_factor=2, mode="bilinear", align_corners=True)
    dY = x2.shape[2] - x1.shape[2]
    dX = x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, (dX // 2, dX - dX // 2, dY // 2, dY - dY // 2))
    return _block(torch.cat([x2, x1], dim=1), p, name)


def unet_forward(x, params):
    """UNet(2,1).forward.  tfpnp/pnp/denoiser/models/unet.py:52-66.  Returns the PRE-clamp output."""
    p = _to_t(params)
    x1 = _block(x, p, "inc.conv")
    x2 = _block(F.max_pool2d(x1, 2), p, "down1.mpconv.1")
    x3 = _block(F.max_pool2d(x2, 2), p, "down2.mpconv.1")
    x4 = _block(F.max_pool2d(x3, 2), p, "down3.mpconv.1")
    x5 = _block(F.max_pool2d(x4, 2), p, "down4.mpconv.1")
    y = _up(x5, x4, p, "up1.conv")
    y = _up(y, x3, p, "up2.conv")
    y = _up(y, x2, p, "up3.conv")
    y = _up(y, x1, p, "up4.conv")
    res = F.conv2d(y, p["outc.conv.weight"], p["outc.conv.bias"])
    return x[:, :1] + res


def denoise(x, sigma, params):
    """UNetDenoiser2D.forward.  tfpnp/pnp/denoiser/base.py:23-32"""
    N, C, H, W = x.shape
    noise_map = torch.ones(N, 1, H, W, dtype=x.dtype) * sigma.view(N, 1, 1, 1)
    return torch.clamp(unet_forward(torch.cat([x, noise_map], dim=1), params), 0, 1)


def drunet_forward(x, params):
    """DRUNet (KAIR UNetRes) on a [B, 2, H, W] input (image + noise-level map); returns the PRE-clamp output.
    Parts as in tfpnp/pnp/denoiser/models/basicblock.py: bias-free Conv2d 3x3 pad 1 (conv :61-66), ResBlock =
    x + conv(relu(conv(x))) (:211-227), strided Conv2d k2 s2 p0 (downsample_strideconv :437-446), ConvTranspose2d k2 s2 p0
    (upsample_convtranspose :413-419); additive skips; H, W multiples of 8."""
    p = _to_t(params)
    nb = sum(1 for k in p if k.startswith("m_body.") and k.endswith(".res.0.weight"))

    def resblocks(v, prefix, first):
        for i in range(first, first + nb):
            r = F.conv2d(F.relu(F.conv2d(v, p[f"{prefix}.{i}.res.0.weight"], padding=1)), p[f"{prefix}.{i}.res.2.weight"], padding=1)
            v = v + r
        return v

    x1 = F.conv2d(x, p["m_head.weight"], padding=1)
    x2 = F.conv2d(resblocks(x1, "m_down1", 0), p[f"m_down1.{nb}.weight"], stride=2)
    x3 = F.conv2d(resblocks(x2, "m_down2", 0), p[f"m_down2.{nb}.weight"], stride=2)
    x4 = F.conv2d(resblocks(x3, "m_down3", 0), p[f"m_down3.{nb}.weight"], stride=2)
    v = resblocks(x4, "m_body", 0)
    v = resblocks(F.conv_transpose2d(v + x4, p["m_up3.0.weight"], stride=2), "m_up3", 1)
    v = resblocks(F.conv_transpose2d(v + x3, p["m_up2.0.weight"], stride=2), "m_up2", 1)
    v = resblocks(F.conv_transpose2d(v + x2, p["m_up1.0.weight"], stride=2), "m_up1", 1)
    return F.conv2d(v + x1, p["m_tail.weight"], padding=1)


def drunet_denoise(x, sigma, params):
    """DRUNet behind the denoiser contract of UNetDenoiser2D.forward (tfpnp/pnp/denoiser/base.py:23-32): noise-level map
    concatenated as a second channel, output clamped to [0, 1]."""
    N, C, H, W = x.shape
    noise_map = torch.ones(N, 1, H, W, dtype=x.dtype) * sigma.view(N, 1, 1, 1)
    return torch.clamp(drunet_forward(torch.cat([x, noise_map], dim=1), params), 0, 1)


class DRUNetDenoiser:
    def __init__(self, params, dtype=None):
        self.params = _to_t(params)
        if dtype is not None:
            self.params = {k: v.to(dtype) for k, v in self.params.items()}

    def __call__(self, x, sigma):
        return drunet_denoise(x, sigma, self.params)


class Denoiser:
    def __init__(self, params, dtype=None):
        """dtype=torch.float64 gives the double-precision yardstick used by the drift tests."""
        self.params = _to_t(params)
        if dtype is not None:
            self.params = {k: v.to(dtype) for k, v in self.params.items()}

    def __call__(self, x, sigma):
        return denoise(x, sigma, self.params)


# ----------------------------------------------------------------------------- solvers: state packing
def admm_reset(x0):
    """ADMMSolver.reset.  tfpnp/pnp/solver/base.py:95-99"""
    x = x0.clone()
    return torch.cat((x, x.clone(), torch.zeros_like(x)), dim=1)


def _v5(t, B):
    return t.view(B, 1, 1, 1, 1)


def csmri_admm(den, variables, y0, mask, sigma_d, mu, iter_num=None):
    """ADMMSolver_CSMRI.forward.  tasks/csmri/solver.py:29-57"""
    x, z, u = torch.split(variables, variables.shape[1] // 3, dim=1)
    B = x.shape[0]
    T = sigma_d.shape[-1] if iter_num is None else iter_num
    m = mask.bool().unsqueeze(-1)
    for i in range(T):
        x = real2complex(den(complex2real(z - u), sigma_d[:, i]))
        k = fft2c(x + u)
        _mu = _v5(mu[:, i], B)
        temp = (_mu * k + y0) / (1 + _mu)
        k = torch.where(m, temp, k)
        z = ifft2c(k)
        u = u + x - z
    return torch.cat((x, z, u), dim=1)


def csmri_hqs(den, variables, y0, mask, sigma_d, mu, iter_num=None):
    """HQSSolver_CSMRI.forward.  tasks/csmri/solver.py:64-89"""
    x, z = torch.split(variables, variables.shape[1] // 2, dim=1)
    B = x.shape[0]
    T = sigma_d.shape[-1] if iter_num is None else iter_num
    m = mask.bool().unsqueeze(-1)
    for i in range(T):
        x = real2complex(den(complex2real(z), sigma_d[:, i]))
        k = fft2c(x)
        _mu = _v5(mu[:, i], B)
        temp = (_mu * k + y0) / (1 + _mu)
        k = torch.where(m, temp, k)
        z = ifft2c(k)
    return torch.cat([x, z], dim=1)


def csmri_pg(den, variables, y0, mask, sigma_d, tau, iter_num=None):
    """PGSolver_CSMRI.forward.  tasks/csmri/solver.py:96-120"""
    x = variables
    B = x.shape[0]
    T = sigma_d.shape[-1] if iter_num is None else iter_num
    m = mask.bool().unsqueeze(-1)
    for i in range(T):
        temp = fft2c(x) - y0
        temp = torch.where(m, temp, torch.zeros_like(temp))
        z = x - _v5(tau[:, i], B) * ifft2c(temp)
        x = real2complex(den(complex2real(z), sigma_d[:, i]))
    return x


def csmri_apg(den, variables, y0, mask, sigma_d, tau, beta, iter_num=None):
    """APGSolver_CSMRI.forward.  tasks/csmri/solver.py:127-165"""
    x, s = torch.split(variables, variables.shape[1] // 2, dim=1)
    B = x.shape[0]
    T = sigma_d.shape[-1] if iter_num is None else iter_num
    m = mask.bool().unsqueeze(-1)
    for i in range(T):
        temp = fft2c(s) - y0
        temp = torch.where(m, temp, torch.zeros_like(temp))
        z = s - _v5(tau[:, i], B) * ifft2c(temp)
        x_prev = x
        x = real2complex(den(complex2real(z), sigma_d[:, i]))
        s = x + _v5(beta[:, i], B) * (x - x_prev)
    return torch.cat([x, s], dim=1)


def csmri_redadmm(den, variables, y0, mask, sigma_d, mu, lamda, iter_num=None):
    """REDADMMSolver_CSMRI.forward.  tasks/csmri/solver.py:172-204"""
    x, z, u = torch.split(variables, variables.shape[1] // 3, dim=1)
    B = x.shape[0]
    T = sigma_d.shape[-1] if iter_num is None else iter_num
    m = mask.bool().unsqueeze(-1)
    for i in range(T):
        _mu = _v5(mu[:, i], B)
        _la = _v5(lamda[:, i], B)
        x_half = real2complex(den(complex2real(x), sigma_d[:, i]))
        x = (_la * x_half + _mu * (z - u)) / (_mu + _la)
        k = fft2c(x + u)
        temp = (_mu * k + y0) / (1 + _mu)
        k = torch.where(m, temp, k)
        z = ifft2c(k)
        u = u + x - z
    return torch.cat([x, z, u], dim=1)


def pr_reset(x0):
    """IADMMSolver_PR.reset.  tasks/pr/solver.py:29-35"""
    x = real2complex(x0.clone())
    return torch.cat([x, x.clone(), torch.zeros_like(x)], dim=1)


def pr_iadmm(den, variables, y0, mask, sigma_d, mu, tau, iter_num=None):
    """IADMMSolver_PR.forward.  tasks/pr/solver.py:37-76 (no epsilon in the division, as the reference)."""
    x, z, u = torch.split(variables, variables.shape[1] // 3, dim=1)
    B = x.shape[0]
    T = sigma_d.shape[-1] if iter_num is None else iter_num
    for i in range(T):
        x = real2complex(den(complex2real(z - u), sigma_d[:, i]))
        _tau = _v5(tau[:, i], B)
        _mu = _v5(mu[:, i], B)
        Az = cdp_forward(z, mask)
        y_hat = complex_abs(Az)
        meas_err = y_hat - y0
        gf = torch.stack((meas_err / y_hat * Az[..., 0], meas_err / y_hat * Az[..., 1]), -1)
        g = cdp_backward(gf, mask)
        z = z - _tau * (g + _mu * (z - (x + u)))
        u = u + x - z
    return torch.cat([x, z, u], dim=1)


def spi_admm(den, variables, x0, Kmap, sigma_d, mu, iter_num=None):
    """ADMMSolver_SPI.forward.  tasks/spi/solver.py:17-52 (order: z, u, x)."""
    x, z, u = torch.split(variables, variables.shape[1] // 3, dim=1)
    B = x.shape[0]
    T = sigma_d.shape[-1] if iter_num is None else iter_num
    K = Kmap[:, 0, 0, 0].view(B, 1, 1, 1) * 10
    K1 = x0 * (K ** 2)
    for i in range(T):
        _mu = mu[:, i].view(B, 1, 1, 1)
        z = spi_inverse(x + u, K1, K, _mu)
        u = u + x - z
        x = den(z - u, sigma_d[:, i])
    return torch.cat([x, z, u], dim=1)


# ----------------------------------------------------------------------------- CT (parity unpinned)
def radon_geometry(res, n_view):
    """create_radon: angles = linspace(0, 179/180*pi, view), det_count = ceil(sqrt(2)*res).
    tfpnp/utils/transforms.py:487-491"""
    angles = np.linspace(0, 179.0 / 180.0 * math.pi, n_view).astype(np.float32)
    det = int(np.ceil(np.sqrt(2) * res))
    return angles, det


def radon_forward(img, angles, det):
    """Own discretisation (documented in DESIGN.md): ray-driven, unit detector spacing, n_steps = det
    unit steps along each ray, bilinear sampling with zero outside.  img [B,1,R,R] -> sino [B,1,V,det].
    Sample k of ray (v,s):  (px,py) = s_pos*(cos,sin) + t_k*(-sin,cos),  s_pos = s-det/2+0.5,
    t_k = k-det/2+0.5; pixel coords  (px + R/2 - 0.5, py + R/2 - 0.5)."""
    B, _, R, _ = img.shape
    V = len(angles)
    s = torch.arange(det, dtype=torch.float32) - det / 2 + 0.5
    t = torch.arange(det, dtype=torch.float32) - det / 2 + 0.5
    out = torch.zeros(B, 1, V, det)
    for v in range(V):
        c, sn = float(np.float32(math.cos(float(angles[v])))), float(np.float32(math.sin(float(angles[v]))))
        px = s[:, None] * c - t[None, :] * sn + (R / 2 - 0.5)   # [det, steps]
        py = s[:, None] * sn + t[None, :] * c + (R / 2 - 0.5)
        x0 = torch.floor(px)
        y0 = torch.floor(py)
        fx = px - x0
        fy = py - y0
        acc = torch.zeros(B, det, det)
        for dy in (0, 1):
            for dx in (0, 1):
                xi = (x0 + dx).long()
                yi = (y0 + dy).long()
                w = (fx if dx else 1 - fx) * (fy if dy else 1 - fy)
                ok = (xi >= 0) & (xi < R) & (yi >= 0) & (yi < R)
                xi = xi.clamp(0, R - 1)
                yi = yi.clamp(0, R - 1)
                vals = img[:, 0][:, yi, xi]  # [B, det, steps]
                acc = acc + vals * (w * ok)[None]
        out[:, 0, v] = acc.sum(-1)
    return out


def radon_backprojection(sino, angles, res):
    """Pixel-driven adjoint-like backprojection: for each pixel, sum over views of the
    linearly interpolated sinogram at s = x*cos + y*sin (zero outside the detector)."""
    B, _, V, det = sino.shape
    R = res
    ys, xs = torch.meshgrid(torch.arange(R, dtype=torch.float32) - (R / 2 - 0.5),
                            torch.arange(R, dtype=torch.float32) - (R / 2 - 0.5), indexing="ij")
    out = torch.zeros(B, 1, R, R)
    for v in range(V):
        c, sn = float(np.float32(math.cos(float(angles[v])))), float(np.float32(math.sin(float(angles[v]))))
        sp = xs * c + ys * sn + (det / 2 - 0.5)
        s0 = torch.floor(sp)
        f = sp - s0
        for d in (0, 1):
            si = (s0 + d).long()
            w = f if d else 1 - f
            ok = (si >= 0) & (si < det)
            si = si.clamp(0, det - 1)
            out[:, 0] += sino[:, 0, v][:, si] * (w * ok)[None]
    return out


def radon_opnorm(res, n_view, n_iter=10, seed=0):
    """power_method_opnorm on backward(forward(.)).  tfpnp/utils/transforms.py:447-462.
    The reference starts from an unseeded randn on the GPU (non-deterministic); here the start
    vector is seeded so that oracle and product agree."""
    angles, det = radon_geometry(res, n_view)
    x = torch.from_numpy(np.random.RandomState(seed).standard_normal((1, 1, res, res)).astype(np.float32))
    x = x / x.norm()
    v = 1.0
    for _ in range(n_iter):
        nx = radon_backprojection(radon_forward(x, angles, det), angles, res)
        v = float(nx.norm())
        x = nx / v
    return v ** 0.5


def ct_iadmm(den, variables, y0, n_view, opnorm, sigma_d, mu, tau, iter_num=None):
    """IADMMSolver_CT.forward.  tasks/ct/solver.py:17-53"""
    x, z, u = torch.split(variables, variables.shape[1] // 3, dim=1)
    B, _, R, _ = x.shape
    angles, det = radon_geometry(R, n_view)
    T = sigma_d.shape[-1] if iter_num is None else iter_num
    for i in range(T):
        x = den(z - u, sigma_d[:, i])
        _tau = tau[:, i].view(B, 1, 1, 1)
        _mu = mu[:, i].view(B, 1, 1, 1)
        g = radon_backprojection(radon_forward(z, angles, det) - y0, angles, R) / opnorm ** 2
        z = z - _tau * (g + _mu * (z - (x + u)))
        u = u + x - z
    return torch.cat([x, z, u], dim=1)


def ct_pg(den, variables, y0, n_view, opnorm, sigma_d, tau, iter_num=None):
    """PGSolver_CT.forward.  tasks/ct/solver.py:61-87"""
    x = variables
    B, _, R, _ = x.shape
    angles, det = radon_geometry(R, n_view)
    T = sigma_d.shape[-1] if iter_num is None else iter_num
    for i in range(T):
        _tau = tau[:, i].view(B, 1, 1, 1)
        z = x - _tau * radon_backprojection(radon_forward(x, angles, det) - y0, angles, R) / opnorm ** 2
        x = den(z, sigma_d[:, i])
    return x


# ----------------------------------------------------------------------------- policy actor (eval mode)
def policy_forward(params, state, spi_head=False):
    """ResNetActorBase.forward up to the head activations, eval-mode BatchNorm.  tfpnp/policy/network.py:87-147
    (ResNetEncoder :87-125, BasicBlock :33-58, heads :137-147, SPI head :262-268).  Returns (probs [B,2], det [B,n])."""
    p = {k: v.to(state.dtype) for k, v in _to_t(params).items()}

    def bn(x, pre):
        return F.batch_norm(x, p[pre + ".running_mean"], p[pre + ".running_var"], p[pre + ".weight"], p[pre + ".bias"],
                            False, 0.1, 1e-5)

    x = F.relu(bn(F.conv2d(state, p["actor_encoder.conv1.weight"], stride=2, padding=1), "actor_encoder.bn1"))
    for li in range(1, 5):
        for blk in range(2):
            pre = f"actor_encoder.layer{li}.{blk}"
            stride = 2 if blk == 0 else 1
            out = F.relu(bn(F.conv2d(x, p[pre + ".conv1.weight"], stride=stride, padding=1), pre + ".bn1"))
            out = bn(F.conv2d(out, p[pre + ".conv2.weight"], padding=1), pre + ".bn2")
            sc = x
            if blk == 0:
                sc = bn(F.conv2d(x, p[pre + ".shortcut.0.weight"], stride=stride), pre + ".shortcut.1")
            x = F.relu(out + sc)
    x = F.adaptive_avg_pool2d(x, 1).view(x.shape[0], -1)
    probs = torch.softmax(F.linear(x, p["fc_softmax.0.weight"], p["fc_softmax.0.bias"]), dim=1)
    h = F.linear(x, p["fc_deterministic.0.weight"], p["fc_deterministic.0.bias"])
    if spi_head:
        h = F.linear(F.relu(h), p["fc_deterministic.2.weight"], p["fc_deterministic.2.bias"])
    return probs, torch.sigmoid(h)


# ----------------------------------------------------------------------------- metric + env step contract
def torch_psnr(output, gt):
    """tfpnp/env/base.py:237-242"""
    N = output.shape[0]
    output = torch.clamp(output, 0, 1)
    mse = torch.mean(F.mse_loss(output.reshape(N, -1), gt.reshape(N, -1), reduction="none"), dim=1)
    return (10 * torch.log10(1.0 / mse)).unsqueeze(1)


class CSMRIEnvOracle:
    """The call contract of PnPEnv.reset/step for CS-MRI ADMM (tfpnp/env/base.py:121-191,
    tasks/csmri/env.py:28-56): live-row gather, solver call, write-back, delta-PSNR reward,
    idx_left shrink.  Observation packing / policy are out of scope."""

    def __init__(self, den, max_episode_step):
        self.den = den
        self.max_episode_step = max_episode_step

    def reset(self, data):
        self.state = {k: v.clone() for k, v in data.items()}
        self.state["solver"] = admm_reset(data["x0"])
        B = data["gt"].shape[0]
        self.idx_left = torch.arange(B)
        self.cur_step = 0
        self.last_metric = torch_psnr(self.state["output"], self.state["gt"])

    def step(self, action):
        self.cur_step += 1
        il = self.idx_left
        st = csmri_admm(self.den, self.state["solver"][il], self.state["y0"][il], self.state["mask"][il],
                        action["sigma_d"], action["mu"])
        self.state["output"][il] = complex2real(st[:, :1])
        self.state["solver"][il] = st
        metric = torch_psnr(self.state["output"], self.state["gt"])
        reward = metric - self.last_metric
        self.last_metric = metric
        idx_stop = action["idx_stop"]
        self.idx_left = il[idx_stop == 0]
        all_done = len(self.idx_left) == 0
        done = idx_stop.clone()
        if self.cur_step == self.max_episode_step:
            all_done = True
            done = torch.ones_like(idx_stop)
        return reward, all_done, done
