"""Validation-generation loggers (reference: verl/utils/logger/gen_logger.py:32-104): rows of (input, output, label, score) per validation
step to the console, a growing wandb table, or swanlab text cards.  Backends whose package is missing are skipped."""
from __future__ import annotations

import os
from abc import ABC, abstractmethod
from typing import List, Tuple

from ..py_functional import is_package_available

Sample = Tuple[str, str, str, float]


class GenerationLogger(ABC):
    @abstractmethod
    def log(self, samples: List[Sample], step: int) -> None: ...


class ConsoleGenerationLogger(GenerationLogger):
    def log(self, samples: List[Sample], step: int) -> None:
        if int(os.environ.get("RANK", 0)) != 0:
            return
        for inp, out, lab, score in samples:
            print(f"[val generation @ step {step}] score={score:.4g}\n  prompt: {inp[:200]!r}\n  output: {out[:400]!r}\n  label : {str(lab)[:200]!r}", flush=True)


class WandbGenerationLogger(GenerationLogger):
    """one table row per validation step, columns step, input_1, output_1, label_1, score_1, input_2, ...; the table is re-created with
    the old rows every time (wandb shows only tables logged as new objects — the reference's workaround, gen_logger.py:58-69)"""

    def __init__(self):
        self.rows: list = []

    def log(self, samples: List[Sample], step: int) -> None:
        import wandb
        columns = ["step"]
        for i in range(len(samples)):
            columns += [f"input_{i + 1}", f"output_{i + 1}", f"label_{i + 1}", f"score_{i + 1}"]
        row = [step]
        for s in samples:
            row.extend(s)
        self.rows.append(row)
        wandb.log({"val/generations": wandb.Table(columns=columns, data=[list(r) for r in self.rows])}, step=step)


class SwanlabGenerationLogger(GenerationLogger):
    def log(self, samples: List[Sample], step: int) -> None:
        import swanlab
        cards = []
        for i, (inp, out, lab, score) in enumerate(samples):
            text = "\n\n---\n\n".join((f"input: {inp}", f"output: {out}", f"label: {lab}", f"score: {score}"))
            cards.append(swanlab.Text(text, caption=f"sample {i + 1}"))
        swanlab.log({"val/generations": cards}, step=step)


GEN_LOGGERS = {"console": ConsoleGenerationLogger, "wandb": WandbGenerationLogger, "swanlab": SwanlabGenerationLogger}


class AggregateGenerationsLogger:
    def __init__(self, loggers: List[str]):
        self.loggers: List[GenerationLogger] = [GEN_LOGGERS[name]() for name in loggers
                                                if name in GEN_LOGGERS and (name == "console" or is_package_available(name))]

    def log(self, samples: List[Sample], step: int) -> None:
        for lg in self.loggers:
            lg.log(samples, step)
