"""Mirror of the reference's ``tensortools`` package: the TFRecord front-end of the scoring path
(``input``, ``tfrecord``).  losses / metrics / checkpoint_manager are training-side and out of scope."""
from . import input, tfrecord  # noqa: F401
from .input import InputStage, NumpyCapsule, generate_mask  # noqa: F401
