"""Import-path shim: ``assets.finger_sampler`` / ``assets.finger_3d`` of the reference map onto ``dgdm_amd.assets.*``
(only the sampler-adjacent decode functions exist here, see dgdm_amd/assets/__init__.py)."""
import sys as _sys
from dgdm_amd.assets import finger_3d, finger_sampler  # noqa: F401
for _n in ("finger_3d", "finger_sampler"):
    _sys.modules[__name__ + "." + _n] = getattr(_sys.modules[__name__], _n)
