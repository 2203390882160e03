"""MI355X-native pool-scoring path of alfrunesiq/SemanticSegmentationActiveLearning.

Scope: ENet forward inference over the unlabelled pool + per-pixel
softmax-entropy / margin / confidence acquisition score + float64 per-image mean + top-k selection,
behind the reference's ``models.ENet`` operator API.  The arithmetic lives in hand-written HIP
kernels for gfx950 (``csrc/``) exported through a C ABI (``include/ssal_enet.h``, ``include/ssal_icnet.h``
for the ICNet row, ICNET_SPEC.md); this package is
the Python host side (weights, sequencing, ``torch.distributed`` sharding).  No CPU fallback.
"""
from . import models  # noqa: F401
from .models import ENet, ICNet  # noqa: F401

__all__ = ["models", "ENet", "ICNet"]
__version__ = "0.2.0"
