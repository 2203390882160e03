"""ICNet placeholder.

The reference's ``models/icnet/icnet.py:1-7`` is an empty class (a paper URL in the docstring, an
``__init__`` that does nothing) and is not exported by ``models/__init__.py``.  There is therefore no
reference computation, layout or parity target for ICNet: BASELINE config C4
is *undefined* on the reference side.  This class keeps the name importable and fails loudly instead
of inventing behaviour; the margin acquisition measure that C4 asks for is implemented for ENet
(``ENet.score(measure="margin")``).
"""


class ICNet:
    """https://arxiv.org/abs/1704.08545 (reference: empty stub)"""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "ICNet has no reference implementation to match (reference models/icnet/icnet.py is an "
            "empty stub); use models.ENet")
