"""Mirror of the reference's ``models`` package (models/__init__.py:1-3): exports ENet -- and ICNet, which the
reference leaves as an empty class (models/icnet/icnet.py:1-7; architecture pinned in ICNET_SPEC.md)."""
from .enet.enet import ENet
from .icnet.icnet import ICNet

__all__ = ["ENet", "ICNet"]
