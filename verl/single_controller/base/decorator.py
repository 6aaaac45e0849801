"""The reference's import path `verl.single_controller.base.decorator` (verl/workers/fsdp_workers.py:42 imports `Dispatch, register`
from it): the objects live in `verl/single_controller/decorator.py`; third-party worker code written against the reference resolves here."""
from ..decorator import MAGIC_ATTR, Dispatch, Execute, register  # noqa: F401

__all__ = ["MAGIC_ATTR", "Dispatch", "Execute", "register"]
