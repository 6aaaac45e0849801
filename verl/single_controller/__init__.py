from .decorator import Dispatch, Execute, register  # noqa: F401
from .worker_group import SPMDWorkerGroup  # noqa: F401
