"""Mirror of the reference's ``models`` package (models/__init__.py:1-3): exports ENet."""
from .enet.enet import ENet

__all__ = ["ENet"]
