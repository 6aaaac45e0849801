"""`verl.utils.logger` — the reference's package surface (verl/utils/logger/__init__.py: Tracker)."""
from .logger import Tracker

__all__ = ["Tracker"]
