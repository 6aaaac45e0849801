from .custom import CustomRewardManager  # noqa: F401
