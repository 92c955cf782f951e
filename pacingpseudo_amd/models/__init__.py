from .unet import UNet  # noqa: F401
from .aux_path_memory import AuxPath  # noqa: F401
from .consistency_reglur_memory import ConsistencyRegulr  # noqa: F401
