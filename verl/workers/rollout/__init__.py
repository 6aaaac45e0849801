from .hip_rollout import assemble_rollout_batch  # noqa: F401
