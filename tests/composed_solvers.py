"""The reference's solver loops composed from the differentiable native building blocks (tfpnp_amd/autograd.py) -- the
independent path the fused native training loops (pnpx_*_train / *_backward) are tested against.  TEST INFRASTRUCTURE: the
product classes carry only their fused native forward; `install()` attaches these compositions to them as
`_forward_autograd` for the GPU tests and tools/time_train.py.  Each function cites the reference loop it restates."""
import torch

from tfpnp_amd import autograd as A
from tfpnp_amd import torch_ops as T
from tfpnp_amd.tasks import csmri, ct, pr, spi


def _v5(t, B):
    return t.reshape(B, 1, 1, 1, 1)


def _blend(k, y0, m, mu):
    """k[mask] = ((mu*k) + y0)[mask] / (1 + mu)   (tasks/csmri/solver.py:49-51) without in-place writes."""
    return torch.where(m, (mu * k + y0) / (1 + mu), k)

def csmri_ADMMSolver_CSMRI(self, variables, y0, mask, sigma_d, mu, iter_num):
    """The reference's loop (tasks/csmri/solver.py:43-55) from differentiable building blocks: the composition the
    fused native VJP (pnpx_csmri_admm_backward) is tested against; the other solvers' training paths look like this."""
    x, z, u = torch.split(variables, variables.shape[1] // 3, dim=1)
    B = x.shape[0]
    m = (mask != 0).unsqueeze(-1)
    for i in range(sigma_d.shape[-1] if iter_num is None else iter_num):
        x = A.r2c(self.prox_mapping(A.c2r(z - u), sigma_d[:, i]))
        z = A.fft2(_blend(A.fft2(x + u), y0, m, _v5(mu[:, i], B)), inverse=True)
        u = u + x - z
    return torch.cat((x, z, u), dim=1)

def csmri_HQSSolver_CSMRI(self, variables, y0, mask, sigma_d, mu, iter_num):
    """The reference's loop (tasks/csmri/solver.py:76-85) from differentiable building blocks: what the fused native VJP
    (pnpx_csmri_hqs_backward) is tested against."""
    x, z = torch.split(variables, variables.shape[1] // 2, dim=1)
    B, m = x.shape[0], (mask != 0).unsqueeze(-1)
    for i in range(sigma_d.shape[-1] if iter_num is None else iter_num):
        x = A.r2c(self.prox_mapping(A.c2r(z), sigma_d[:, i]))
        z = A.fft2(_blend(A.fft2(x), y0, m, _v5(mu[:, i], B)), inverse=True)
    return torch.cat([x, z], dim=1)

def csmri_PGSolver_CSMRI(self, variables, y0, mask, sigma_d, tau, iter_num):
    """The reference's loop (tasks/csmri/solver.py:107-116) from differentiable building blocks: what the fused native
    VJP (pnpx_csmri_pg_backward) is tested against."""
    x, B, m = variables, variables.shape[0], (mask != 0).unsqueeze(-1)
    for i in range(sigma_d.shape[-1] if iter_num is None else iter_num):
        temp = torch.where(m, A.fft2(x) - y0, torch.zeros_like(y0))
        z = x - _v5(tau[:, i], B) * A.fft2(temp, inverse=True)
        x = A.r2c(self.prox_mapping(A.c2r(z), sigma_d[:, i]))
    return x

def csmri_APGSolver_CSMRI(self, variables, y0, mask, sigma_d, tau, beta, iter_num):
    """The reference's loop (tasks/csmri/solver.py:141-159) from differentiable building blocks: what the fused native VJP
    (pnpx_csmri_apg_backward) is tested against."""
    x, s = torch.split(variables, variables.shape[1] // 2, dim=1)
    B, m = x.shape[0], (mask != 0).unsqueeze(-1)
    for i in range(sigma_d.shape[-1] if iter_num is None else iter_num):
        temp = torch.where(m, A.fft2(s) - y0, torch.zeros_like(y0))
        z = s - _v5(tau[:, i], B) * A.fft2(temp, inverse=True)
        x_prev = x
        x = A.r2c(self.prox_mapping(A.c2r(z), sigma_d[:, i]))
        s = x + _v5(beta[:, i], B) * (x - x_prev)
    return torch.cat([x, s], dim=1)

def csmri_REDADMMSolver_CSMRI(self, variables, y0, mask, sigma_d, mu, lamda, iter_num):
    """The reference's loop (tasks/csmri/solver.py:183-200) from differentiable building blocks: what the fused native VJP
    (pnpx_csmri_redadmm_backward) is tested against."""
    x, z, u = torch.split(variables, variables.shape[1] // 3, dim=1)
    B, m = x.shape[0], (mask != 0).unsqueeze(-1)
    for i in range(sigma_d.shape[-1] if iter_num is None else iter_num):
        _mu, _la = _v5(mu[:, i], B), _v5(lamda[:, i], B)
        x_half = A.r2c(self.prox_mapping(A.c2r(x), sigma_d[:, i]))
        x = (_la * x_half + _mu * (z - u)) / (_mu + _la)
        z = A.fft2(_blend(A.fft2(x + u), y0, m, _mu), inverse=True)
        u = u + x - z
    return torch.cat([x, z, u], dim=1)

def pr_IADMMSolver_PR(self, variables, y0, mask, sigma_d, mu, tau, iter_num):
    """The reference's loop (tasks/pr/solver.py:49-72) from differentiable building blocks: what the fused native VJP
    (pnpx_pr_iadmm_backward) is tested against."""
    x, z, u = torch.split(variables, variables.shape[1] // 3, dim=1)
    B, S = x.shape[0], mask.shape[1]
    for i in range(sigma_d.shape[-1] if iter_num is None else iter_num):
        x = A.r2c(self.prox_mapping(A.c2r(z - u), sigma_d[:, i]))
        _tau, _mu = tau[:, i].reshape(B, 1, 1, 1, 1), mu[:, i].reshape(B, 1, 1, 1, 1)
        Az = A.fft2(A.cmul(z.repeat(1, S, 1, 1, 1), mask), centered=False)                  # cdp_forward
        y_hat = (Az ** 2).sum(dim=-1).sqrt()
        q = ((y_hat - y0) / y_hat).unsqueeze(-1)
        g = A.cmul(A.fft2(q * Az, inverse=True, centered=False), A.conj(mask)).mean(1, keepdim=True)  # cdp_backward
        z = z - _tau * (g + _mu * (z - (x + u)))
        u = u + x - z
    return torch.cat([x, z, u], dim=1)

def spi_ADMMSolver_SPI(self, variables, x0, K, sigma_d, mu, iter_num):
    """The reference's loop (tasks/spi/solver.py:36-47) from differentiable building blocks: what the fused native VJP
    (pnpx_spi_admm_backward) is tested against."""
    x, z, u = torch.split(variables, variables.shape[1] // 3, dim=1)
    B = x.shape[0]
    Kv = K[:, 0, 0, 0].reshape(B, 1, 1, 1) * 10
    K1 = x0 * (Kv ** 2)
    for i in range(sigma_d.shape[-1] if iter_num is None else iter_num):
        _mu = mu[:, i].reshape(B, 1, 1, 1)
        zt = x + u
        # spi_inverse (transforms.py:404-439): the bisection result carries no gradient in the reference
        # either (bmin/bmax/bave are built by masked assignment of constants); only the K1 == 0 branch does.
        bis = T.call("spi_inverse", zt.detach(), K1, Kv, _mu.detach())
        z = torch.clamp(torch.where(K1 == 0, zt - (Kv ** 2 - K1) / _mu, bis), 0.0, 1.0)
        u = u + x - z
        x = self.prox_mapping(z - u, sigma_d[:, i])
    return torch.cat([x, z, u], dim=1)

def ct_IADMMSolver_CT(self, variables, y0, sigma_d, mu, tau, iter_num):
    """The reference's loop (tasks/ct/solver.py:32-49) from differentiable building blocks: what the fused native VJP
    (pnpx_ct_iadmm_backward) is tested against."""
    n_view = int(y0.shape[2])
    radon = self.radon_generator(variables.shape[-1], n_view, device=variables.device)
    x, z, u = torch.split(variables, variables.shape[1] // 3, dim=1)
    B, R = x.shape[0], x.shape[-1]
    for i in range(sigma_d.shape[-1] if iter_num is None else iter_num):
        x = self.prox_mapping(z - u, sigma_d[:, i])
        _tau, _mu = tau[:, i].reshape(B, 1, 1, 1), mu[:, i].reshape(B, 1, 1, 1)
        g = A.radon_backprojection(A.radon_forward(z, n_view) - y0, R) / radon.opnorm ** 2
        z = z - _tau * (g + _mu * (z - (x + u)))
        u = u + x - z
    return torch.cat([x, z, u], dim=1)

def ct_PGSolver_CT(self, variables, y0, sigma_d, tau, iter_num):
    """The reference's loop (tasks/ct/solver.py:73-83) from differentiable building blocks."""
    n_view = int(y0.shape[2])
    radon = self.radon_generator(variables.shape[-1], n_view, device=variables.device)
    x, B, R = variables, variables.shape[0], variables.shape[-1]
    for i in range(sigma_d.shape[-1] if iter_num is None else iter_num):
        g = A.radon_backprojection(A.radon_forward(x, n_view) - y0, R) / radon.opnorm ** 2
        x = self.prox_mapping(x - tau[:, i].reshape(B, 1, 1, 1) * g, sigma_d[:, i])
    return x

def install():
    """Attach the compositions above to the product solver classes as `_forward_autograd` (tests / tools only)."""
    csmri.ADMMSolver_CSMRI._forward_autograd = csmri_ADMMSolver_CSMRI
    csmri.HQSSolver_CSMRI._forward_autograd = csmri_HQSSolver_CSMRI
    csmri.PGSolver_CSMRI._forward_autograd = csmri_PGSolver_CSMRI
    csmri.APGSolver_CSMRI._forward_autograd = csmri_APGSolver_CSMRI
    csmri.REDADMMSolver_CSMRI._forward_autograd = csmri_REDADMMSolver_CSMRI
    pr.IADMMSolver_PR._forward_autograd = pr_IADMMSolver_PR
    spi.ADMMSolver_SPI._forward_autograd = spi_ADMMSolver_SPI
    ct.IADMMSolver_CT._forward_autograd = ct_IADMMSolver_CT
    ct.PGSolver_CT._forward_autograd = ct_PGSolver_CT
