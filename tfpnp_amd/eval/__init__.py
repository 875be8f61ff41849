from .evaluator import Evaluator, eval_batch, eval_single, psnr_qrnn3d  # noqa: F401
