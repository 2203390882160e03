"""Mirror of the reference's ``tensortools`` package: the TFRecord front-end of the scoring path
(``input``, ``tfrecord``) and the forward value of ``losses``.  metrics / checkpoint_manager are out of scope."""
from . import input, losses, tfrecord  # noqa: F401
from .input import InputStage, NumpyCapsule, generate_mask  # noqa: F401
