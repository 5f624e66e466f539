"""mt355: MI355X-native DFPN / CHN training-step hot path.

Exports the names the reference package exposes for this path
(``master_thesis/__init__.py:1-12``): the models call each other through
``mt.DFPN``, ``mt.FlowsUtils`` ... exactly as upstream.  Importing the package
does not need a GPU; running any op does, and needs ``libmt355.so`` (built by
``__graft_entry__.build()``): there is no CPU fallback.
"""
from .model_vgg import VGGFeatures  # noqa: F401
from .model_dfpn import DFPN  # noqa: F401
from .model_chn import CHN  # noqa: F401
from .model_cpn import CPN  # noqa: F401
from .utils import FlowsUtils, LossesUtils, MeasuresUtils, MovementsUtils, TransformsUtils  # noqa: F401
from .optim import FlatAdam  # noqa: F401

__all__ = ['CHN', 'CPN', 'DFPN', 'VGGFeatures', 'FlowsUtils', 'LossesUtils', 'MeasuresUtils', 'MovementsUtils', 'TransformsUtils',
           'FlatAdam']
