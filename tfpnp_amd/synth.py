"""Deterministic synthetic weights and measurements (host side, numpy only).

The reference ships no weights, images, masks or checkpoints
(/root/reference/.gitignore:4-7,13,15; tfpnp/pnp/denoiser/base.py:10-13), so every
test, the smoke run and bench.py regenerate identical inputs from seeds here.
Formulas for the measurements follow the reference datasets:
  CS-MRI  tasks/csmri/dataset.py:56-64  (y0 = mask * (fft2c(gt) + sigma*N), x0 = ifft2c(y0))
  PR      tasks/pr/dataset.py:53-57     (y0 = |A gt| + Poisson-like noise, x0 = ones)
  SPI     tasks/spi/dataset.py:49-50    (binary photon counts averaged over K x K)
  CT      tasks/ct/dataset.py:92-98     (sinogram + percentage Gaussian noise)
Nothing in this file touches the GPU or the oracle.
"""
import math
import numpy as np

# (name, cin, cout) of the 27 3x3 convs of UNet(2,1) in state_dict order,
# /root/reference/tfpnp/pnp/denoiser/models/unet.py:37-46.
UNET_BLOCKS = [
    ("inc.conv", 2, 32),
    ("down1.mpconv.1", 32, 64),
    ("down2.mpconv.1", 64, 128),
    ("down3.mpconv.1", 128, 256),
    ("down4.mpconv.1", 256, 512),
    ("up1.conv", 512 + 256, 256),
    ("up2.conv", 256 + 128, 128),
    ("up3.conv", 128 + 64, 64),
    ("up4.conv", 64 + 32, 32),
]


def unet_param_specs():
    """[(key, shape)] in the reference's state_dict order (56 tensors)."""
    specs = []
    for name, cin, cout in UNET_BLOCKS:
        for j in range(3):
            ci = cin if j == 0 else cout
            specs.append((f"{name}.conv-{j}.conv2d.weight", (cout, ci, 3, 3)))
            specs.append((f"{name}.conv-{j}.conv2d.bias", (cout,)))
    specs.append(("outc.conv.weight", (1, 32, 1, 1)))
    specs.append(("outc.conv.bias", (1,)))
    return specs


def unet_num_params():
    return sum(int(np.prod(s)) for _, s in unet_param_specs())


def make_unet_params(seed=0):
    """He-normal synthetic weights: std = sqrt(2 / (1.04 * fan_in)), bias ~ N(0, 0.05^2).

    Returns an ordered dict key -> float32 ndarray (reference key names, so the
    reference's UNet.load_state_dict accepts it unchanged).
    """
    rs = np.random.RandomState(seed)
    out = {}
    for key, shape in unet_param_specs():
        if key.endswith("weight"):
            fan_in = shape[1] * shape[2] * shape[3]
            std = math.sqrt(2.0 / (1.04 * fan_in))
            out[key] = (rs.standard_normal(shape) * std).astype(np.float32)
        else:
            out[key] = (rs.standard_normal(shape) * 0.05).astype(np.float32)
    return out


def make_unet_params_default(seed=0):
    """A second synthetic weight set at PyTorch's DEFAULT initialisation scale (nn.Conv2d.reset_parameters:
    kaiming_uniform(a=sqrt(5)) = U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights and biases): a contractive network like
    the one SURVEY section 6 probed (fp32-vs-fp64 drift 6.9e-7 over 30 iterations), next to the He-scaled set of
    make_unet_params whose gain > 1 amplifies round-off x1.5-2 per call.  Shows the real margin to the 1e-4 bar."""
    rs = np.random.RandomState(seed + 7919)
    out = {}
    fan_in = 1
    for key, shape in unet_param_specs():
        if key.endswith("weight"):
            fan_in = shape[1] * shape[2] * shape[3]
        b = 1.0 / math.sqrt(fan_in)
        out[key] = rs.uniform(-b, b, shape).astype(np.float32)
    return out


def flatten_params(params):
    """Concatenate in state_dict order -> 1-D float32 (the C-ABI's weight blob)."""
    return np.concatenate([params[k].reshape(-1) for k, _ in unet_param_specs()]).astype(np.float32)


# --------------------------------------------------------------------------- DRUNet parameters
# DRUNet (Zhang et al., "Plug-and-Play Image Restoration with Deep Denoiser Prior", KAIR models/network_unet.py::UNetRes):
# the reference ships only its building blocks (tfpnp/pnp/denoiser/models/basicblock.py: conv :61-101, ResBlock
# :211-227, upsample_convtranspose :413-419, downsample_strideconv :437-446) and no model or checkpoint.  Topology:
# head conv3x3(in_nc+1 -> 64); three scales of [nb ResBlocks + strided 2x2 conv]; body of nb ResBlocks at 512 channels;
# three scales of [transposed 2x2 conv + nb ResBlocks] with additive skips; tail conv3x3(64 -> 1); everything bias-free,
# ReLU inside the ResBlocks only.  Key names are those of KAIR's released drunet_gray.pth.
DRUNET_NC = (64, 128, 256, 512)
DRUNET_NB = 4


def drunet_param_specs(in_nc=2, out_nc=1, nc=DRUNET_NC, nb=DRUNET_NB):
    """[(key, shape)] in state_dict order of UNetRes(in_nc, out_nc, nc, nb, act_mode='R', 'strideconv', 'convtranspose')."""
    specs = [("m_head.weight", (nc[0], in_nc, 3, 3))]

    def res(prefix, c):
        return [(f"{prefix}.res.0.weight", (c, c, 3, 3)), (f"{prefix}.res.2.weight", (c, c, 3, 3))]

    for lvl in range(3):
        for i in range(nb):
            specs += res(f"m_down{lvl + 1}.{i}", nc[lvl])
        specs.append((f"m_down{lvl + 1}.{nb}.weight", (nc[lvl + 1], nc[lvl], 2, 2)))
    for i in range(nb):
        specs += res(f"m_body.{i}", nc[3])
    for lvl in (2, 1, 0):
        specs.append((f"m_up{lvl + 1}.0.weight", (nc[lvl + 1], nc[lvl], 2, 2)))      # ConvTranspose2d: [cin, cout, 2, 2]
        for i in range(nb):
            specs += res(f"m_up{lvl + 1}.{i + 1}", nc[lvl])
    specs.append(("m_tail.weight", (out_nc, nc[0], 3, 3)))
    return specs


def drunet_num_params(**kw):
    return sum(int(np.prod(s)) for _, s in drunet_param_specs(**kw))


def make_drunet_params(seed=0, **kw):
    """Synthetic DRUNet weights: N(0, 1/fan_in) convolutions (variance preserving through head / strided / transposed /
    tail layers), He-normal first and 0.25 x second convolution inside every ResBlock (the residual branch adds ~6 % of
    the trunk's variance per block, so 32 blocks stay O(1): activations of the synthetic net peak at a few units)."""
    rs = np.random.RandomState(seed)
    out = {}
    for key, shape in drunet_param_specs(**kw):
        if ".res.0." in key:
            std = math.sqrt(2.0 / (shape[1] * 9))
        elif ".res.2." in key:
            std = 0.25 * math.sqrt(1.0 / (shape[1] * 9))
        elif key.startswith("m_up"):           # ConvTranspose2d [cin, cout, 2, 2]: every output pixel sees cin inputs
            std = math.sqrt(1.0 / shape[0])
        elif key == "m_tail.weight":           # output of order 0.3: a fair share of pixels inside [0, 1] before the clamp
            std = 0.25 * math.sqrt(1.0 / (shape[1] * 9))
        else:
            std = math.sqrt(1.0 / (shape[1] * shape[2] * shape[3]))
        out[key] = (rs.standard_normal(shape) * std).astype(np.float32)
    return out


# --------------------------------------------------------------------------- policy actor parameters
def policy_param_specs(num_inputs, n_det, spi_head=False):
    """(key, shape) of the fp32 entries of ResNetActorBase.state_dict() in registration order
    (tfpnp/policy/network.py:87-147; integer num_batches_tracked entries are not listed)."""
    specs = []

    def bn(prefix, c):
        for k in ("weight", "bias", "running_mean", "running_var"):
            specs.append((f"{prefix}.{k}", (c,)))

    specs.append(("actor_encoder.conv1.weight", (64, num_inputs, 3, 3)))
    bn("actor_encoder.bn1", 64)
    in_planes = 64
    for li, planes in enumerate((64, 128, 256, 512), start=1):
        for blk in range(2):
            pre = f"actor_encoder.layer{li}.{blk}"
            cin = in_planes if blk == 0 else planes
            specs.append((f"{pre}.conv1.weight", (planes, cin, 3, 3)))
            bn(f"{pre}.bn1", planes)
            specs.append((f"{pre}.conv2.weight", (planes, planes, 3, 3)))
            bn(f"{pre}.bn2", planes)
            if blk == 0:
                specs.append((f"{pre}.shortcut.0.weight", (planes, in_planes, 1, 1)))
                bn(f"{pre}.shortcut.1", planes)
        in_planes = planes
    specs += [("fc_softmax.0.weight", (2, 512)), ("fc_softmax.0.bias", (2,))]
    if spi_head:
        specs += [("fc_deterministic.0.weight", (64, 512)), ("fc_deterministic.0.bias", (64,)),
                  ("fc_deterministic.2.weight", (n_det, 64)), ("fc_deterministic.2.bias", (n_det,))]
    else:
        specs += [("fc_deterministic.0.weight", (n_det, 512)), ("fc_deterministic.0.bias", (n_det,))]
    return specs


def make_policy_params(num_inputs, n_det, spi_head=False, seed=0):
    """Synthetic actor weights with non-trivial BatchNorm statistics (no trained actor checkpoint ships with the
    reference): He-normal convolutions, gamma ~ U(0.5,1.5), beta, running_mean ~ N(0,0.1^2), running_var ~ U(0.5,1.5),
    heads ~ N(0, 0.01/fan_in) (keeps softmax / sigmoid away from saturation)."""
    rs = np.random.RandomState(seed)
    out = {}
    for key, shape in policy_param_specs(num_inputs, n_det, spi_head):
        if len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            out[key] = (rs.standard_normal(shape) * math.sqrt(2.0 / fan_in)).astype(np.float32)
        elif len(shape) == 2:
            out[key] = (rs.standard_normal(shape) * 0.1 * math.sqrt(1.0 / shape[1])).astype(np.float32)
        elif key.endswith("running_var") or (key.endswith(".weight") and ".bn" in key or ".shortcut.1.weight" in key):
            out[key] = rs.uniform(0.5, 1.5, shape).astype(np.float32)
        else:
            out[key] = (rs.standard_normal(shape) * 0.1).astype(np.float32)
    return out


# --------------------------------------------------------------------------- images
def phantom(H, W, seed):
    """Smooth ellipse phantom in [0,1], float32 [H,W]."""
    rs = np.random.RandomState(seed)
    yy, xx = np.meshgrid(np.linspace(-1, 1, H, dtype=np.float32),
                         np.linspace(-1, 1, W, dtype=np.float32), indexing="ij")
    img = np.zeros((H, W), np.float32)
    n = rs.randint(4, 9)
    for _ in range(n):
        cx, cy = rs.uniform(-0.5, 0.5, 2)
        a, b = rs.uniform(0.1, 0.6, 2)
        th = rs.uniform(0, math.pi)
        val = rs.uniform(0.1, 0.5)
        xr = (xx - cx) * math.cos(th) + (yy - cy) * math.sin(th)
        yr = -(xx - cx) * math.sin(th) + (yy - cy) * math.cos(th)
        r = (xr / a) ** 2 + (yr / b) ** 2
        img += (val * np.clip(1.5 - 1.5 * r, 0, 1)).astype(np.float32)
    img = np.clip(img, 0, 1)
    return img.astype(np.float32)


def phantom_batch(B, H, W, seed=1234):
    return np.stack([phantom(H, W, seed + b) for b in range(B)])[:, None]  # [B,1,H,W]


def radial_mask(H, W, ratio, seed=0):
    """Radial-line k-space mask through the centre, bool [H,W], sampling ~ 1/ratio."""
    target = H * W / float(ratio)
    yy, xx = np.meshgrid(np.arange(H) - H // 2, np.arange(W) - W // 2, indexing="ij")
    n_lines = max(2, int(target / max(H, W)))
    rs = np.random.RandomState(seed)
    off = rs.uniform(0, math.pi)
    while True:
        mask = np.zeros((H, W), bool)
        L = int(math.hypot(H, W)) + 2
        t = np.arange(-L, L + 1) * 0.5
        for i in range(n_lines):
            th = off + math.pi * i / n_lines
            ys = np.round(t * math.sin(th)).astype(int) + H // 2
            xs = np.round(t * math.cos(th)).astype(int) + W // 2
            ok = (ys >= 0) & (ys < H) & (xs >= 0) & (xs < W)
            mask[ys[ok], xs[ok]] = True
        if mask.sum() >= target or n_lines > 4 * max(H, W):
            return mask
        n_lines += 1


def _fft2c_np(x):
    return np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(x, axes=(-2, -1)), norm="ortho"), axes=(-2, -1))


def _ifft2c_np(x):
    return np.fft.fftshift(np.fft.ifft2(np.fft.ifftshift(x, axes=(-2, -1)), norm="ortho"), axes=(-2, -1))


def c2ri(x):
    """complex ndarray -> float32 [...,2]."""
    return np.stack([x.real, x.imag], -1).astype(np.float32)


def make_csmri_batch(B, H, W, ratio=4, sigma_n=15.0, seed=1234):
    """dict of float32/bool ndarrays shaped like the reference's CS-MRI items (batched)."""
    gt = phantom_batch(B, H, W, seed)
    rs = np.random.RandomState(seed + 77)
    mask = np.stack([radial_mask(H, W, ratio, seed=seed + 13 * b) for b in range(B)])[:, None]
    k = _fft2c_np(gt.astype(np.complex64))
    noise = (rs.standard_normal(k.shape) + 1j * rs.standard_normal(k.shape)) * (sigma_n / 255.0)
    y0 = (k + noise) * mask
    x0 = _ifft2c_np(y0)
    return {
        "gt": gt.astype(np.float32),
        "mask": mask,
        "y0": c2ri(y0),
        "x0": c2ri(x0),
        "ATy0": c2ri(x0),
        "output": x0.real.astype(np.float32),
        "sigma_n": np.full((B, 1, H, W, 2), sigma_n / 255.0, np.float32),
    }


def make_actions(B, n_steps=6, pack=5, sig_hi=50.0 / 255, sig_lo=5.0 / 255, mu_lo=0.1, mu_hi=0.9,
                 tau=None):
    """Per-step action dicts: sigma_d log-spaced, mu linear over n_steps*pack iterations."""
    T = n_steps * pack
    sig = np.exp(np.linspace(math.log(sig_hi), math.log(sig_lo), T)).astype(np.float32)
    mu = np.linspace(mu_lo, mu_hi, T).astype(np.float32)
    acts = []
    for s in range(n_steps):
        a = {
            "sigma_d": np.tile(sig[s * pack:(s + 1) * pack][None], (B, 1)),
            "mu": np.tile(mu[s * pack:(s + 1) * pack][None], (B, 1)),
        }
        if tau is not None:
            a["tau"] = np.full((B, pack), tau, np.float32)
        acts.append(a)
    return acts


def make_pr_batch(B, H, W, S=4, alpha=9.0, seed=1234):
    """Phase retrieval with S coded-diffraction unit-modulus masks."""
    gt = phantom_batch(B, H, W, seed)
    rs = np.random.RandomState(seed + 5)
    phi = rs.uniform(0, 2 * math.pi, (B, S, H, W))
    m = np.exp(1j * phi).astype(np.complex64)
    Az = np.fft.fft2(m * gt.astype(np.complex64), norm="ortho")
    z = np.abs(Az)
    inoise = alpha / 255.0 * np.abs(z) * rs.standard_normal(z.shape)
    y0 = np.sqrt(np.clip(z ** 2 + inoise, 0, None))
    return {
        "gt": gt.astype(np.float32),
        "mask": c2ri(m),
        "y0": y0.astype(np.float32),
        "x0": np.ones((B, 1, H, W), np.float32),
        "output": np.ones((B, 1, H, W), np.float32),
    }


def make_spi_batch(B, H, W, K=6, seed=1234):
    """Single-photon imaging: x0 = avg_pool(binary counts, K) at image size H x W."""
    gt = phantom_batch(B, H, W, seed)
    rs = np.random.RandomState(seed + 9)
    # spi_forward(x, K, alpha=K**2, q=1): theta = alpha * kron(x, 1_KxK) / K**2 = kron(x, 1)
    theta = np.repeat(np.repeat(gt, K, axis=2), K, axis=3)
    y = rs.poisson(theta)
    ob = (y >= 1).astype(np.float32)
    x0 = ob.reshape(B, 1, H, K, W, K).mean(axis=(3, 5)).astype(np.float32)
    return {
        "gt": gt.astype(np.float32),
        "x0": x0,
        "output": x0.copy(),
        "K": np.full((B, 1, H, W), K / 10.0, np.float32),
    }
