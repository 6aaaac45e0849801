"""`Tracker` — the metric sinks behind `trainer.logger` (reference: verl/utils/logger/logger.py:40-155): console, tensorboard, wandb, mlflow,
swanlab, with the reference's calls into each package (project / experiment names from config["trainer"], the flattened config as
hyper-parameters, `log(data, step)`, `finish()`).  Differences, on purpose: a backend whose package is not installed is SKIPPED with one
message (the reference dies with a NameError at start-up — the shipped scripts ask for wandb); only rank 0 of a multi-process run logs
(the reference has one driver process); the console backend prints one line per step (`step N: key:value - ...`, parsed by
`bench.py --through-api`) instead of a YAML block."""
from __future__ import annotations

import os
from abc import ABC, abstractmethod
from typing import Any, Dict, List, Optional, Tuple, Union

from ..py_functional import convert_dict_to_str, flatten_dict, is_package_available
from .gen_logger import AggregateGenerationsLogger


class Logger(ABC):
    @abstractmethod
    def __init__(self, config: Dict[str, Any]) -> None: ...

    @abstractmethod
    def log(self, data: Dict[str, Any], step: int) -> None: ...

    def finish(self) -> None:
        pass


class ConsoleLogger(Logger):
    def __init__(self, config: Dict[str, Any]) -> None:
        if config and os.environ.get("ST_LOG_CONFIG", "0") == "1":
            print("Config\n" + convert_dict_to_str(config))

    def log(self, data: Dict[str, Any], step: int) -> None:
        print(f"step {step}: " + " - ".join(f"{k}:{v:.4g}" if isinstance(v, (int, float)) else f"{k}:{v}" for k, v in sorted(data.items())), flush=True)


class TensorBoardLogger(Logger):
    def __init__(self, config: Dict[str, Any]) -> None:
        from torch.utils.tensorboard import SummaryWriter
        log_dir = os.getenv("TENSORBOARD_DIR", "tensorboard_log")
        os.makedirs(log_dir, exist_ok=True)
        print(f"Saving tensorboard log to {log_dir}.")
        self.writer = SummaryWriter(log_dir)
        hparams = {k: (v if isinstance(v, (int, float, str, bool)) else str(v)) for k, v in flatten_dict(config or {}).items()}
        self.writer.add_hparams(hparams, {})

    def log(self, data: Dict[str, Any], step: int) -> None:
        for key, value in data.items():
            self.writer.add_scalar(key, value, step)

    def finish(self) -> None:
        self.writer.close()


class WandbLogger(Logger):
    def __init__(self, config: Dict[str, Any]) -> None:
        import wandb
        wandb.init(project=config["trainer"]["project_name"], name=config["trainer"]["experiment_name"], config=config)

    def log(self, data: Dict[str, Any], step: int) -> None:
        import wandb
        wandb.log(data=data, step=step)

    def finish(self) -> None:
        import wandb
        wandb.finish()


class MlflowLogger(Logger):
    def __init__(self, config: Dict[str, Any]) -> None:
        import mlflow
        mlflow.start_run(run_name=config["trainer"]["experiment_name"])
        mlflow.log_params(flatten_dict(config))

    def log(self, data: Dict[str, Any], step: int) -> None:
        import mlflow
        mlflow.log_metrics(metrics=data, step=step)


class SwanlabLogger(Logger):
    def __init__(self, config: Dict[str, Any]) -> None:
        import swanlab
        key = os.getenv("SWANLAB_API_KEY")
        if key:
            swanlab.login(key)
        swanlab.init(project=config["trainer"]["project_name"], experiment_name=config["trainer"]["experiment_name"],
                     config={"UPPERFRAMEWORK": "EasyR1", "FRAMEWORK": "veRL", **config}, logdir=os.getenv("SWANLAB_DIR", "swanlab_log"),
                     mode=os.getenv("SWANLAB_MODE", "cloud"))

    def log(self, data: Dict[str, Any], step: int) -> None:
        import swanlab
        swanlab.log(data=data, step=step)

    def finish(self) -> None:
        import swanlab
        swanlab.finish()


LOGGERS = {"wandb": WandbLogger, "mlflow": MlflowLogger, "tensorboard": TensorBoardLogger, "console": ConsoleLogger, "swanlab": SwanlabLogger}
_PACKAGE = {"wandb": "wandb", "mlflow": "mlflow", "tensorboard": "tensorboard", "swanlab": "swanlab"}


class Tracker:
    def __init__(self, loggers: Union[str, List[str], Tuple[str, ...]] = "console", config: Optional[Dict[str, Any]] = None):
        names = [loggers] if isinstance(loggers, str) else list(loggers)
        for name in names:
            if name not in LOGGERS:
                raise ValueError(f"{name} is not supported.")
        self.rank = int(os.environ.get("RANK", 0))
        self.loggers: List[Logger] = []
        active: List[str] = []
        if self.rank == 0:
            for name in names:
                if name in _PACKAGE and not is_package_available(_PACKAGE[name]):
                    print(f"[logger] `{name}` is requested by trainer.logger but the package is not installed: skipped")
                    continue
                self.loggers.append(LOGGERS[name](config or {}))
                active.append(name)
        self.gen_logger = AggregateGenerationsLogger(active)

    def log(self, data: Dict[str, Any], step: int) -> None:
        for lg in self.loggers:
            lg.log(data=data, step=step)

    def log_generation(self, samples: List[Tuple[str, str, str, float]], step: int) -> None:
        self.gen_logger.log(samples, step)

    def finish(self) -> None:
        for lg in self.loggers:
            lg.finish()
        self.loggers = []

    def __del__(self):
        try:
            self.finish()
        except Exception:                                   # interpreter shutdown: the backends may be gone already
            pass
