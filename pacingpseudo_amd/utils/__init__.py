from .utils import AvgMeter, cosine_lr_decay, gaussian_ramp_up, linear_lr_decay, poly_lr_decay  # noqa: F401
