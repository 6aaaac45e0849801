"""`@register(dispatch_mode=...)` tags of worker methods — the RPC contract of the reference
(verl/single_controller/base/decorator.py:33-40,198-213).  Ray is replaced by one process per GPU launched with
torchrun: every rank runs the same driver loop on ITS shard of the batch (SPMD), so DP_COMPUTE_PROTO's
chunk -> workers -> concat (decorator.py:106-123) degenerates to a direct call on the local shard; the attribute
bookkeeping is kept so tools that introspect `MAGIC_ATTR` keep working."""
from enum import Enum
from functools import wraps

MAGIC_ATTR = "attrs_3141562937"


class Dispatch(Enum):
    RANK_ZERO = 0
    ONE_TO_ALL = 1
    ALL_TO_ALL = 2
    DP_COMPUTE = 3
    DP_COMPUTE_PROTO = 4
    DP_COMPUTE_PROTO_WITH_FUNC = 5
    DP_COMPUTE_METRIC = 6


class Execute(Enum):
    ALL = 0
    RANK_ZERO = 1


def register(dispatch_mode=Dispatch.ALL_TO_ALL, execute_mode=Execute.ALL, blocking=True, materialize_futures=True):
    """(materialize_futures: the reference resolves DataProtoFuture arguments before the call, decorator.py:171-196 — there are no futures in a
    one-process-per-GPU run; accepted and ignored)"""
    def decorator(func):
        @wraps(func)
        def inner(*args, **kwargs):
            return func(*args, **kwargs)
        setattr(inner, MAGIC_ATTR, {"dispatch_mode": dispatch_mode, "execute_mode": execute_mode, "blocking": blocking})
        return inner
    return decorator
