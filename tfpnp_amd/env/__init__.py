from .base import PnPEnv, torch_psnr  # noqa: F401
