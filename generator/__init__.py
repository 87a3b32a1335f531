"""Import-path shim: ``generator.*`` of the reference maps onto ``dgdm_amd.generator.*``."""
import sys as _sys
from dgdm_amd.generator import dataloader, diffusion, diffusion_utils, train  # noqa: F401
for _n in ("dataloader", "diffusion", "diffusion_utils"):
    _sys.modules[__name__ + "." + _n] = getattr(_sys.modules[__name__], _n)
