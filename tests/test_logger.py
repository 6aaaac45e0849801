"""verl.utils.logger.Tracker / verl.utils.py_functional (reference: verl/utils/logger/logger.py:40-155, gen_logger.py:32-104,
py_functional.py:50-103).  wandb / mlflow / swanlab / tensorboard are not in this image: stand-in modules that record their calls are put
into sys.modules, so what is checked is the call pattern the reference makes into each package."""
import importlib
import sys
import types

import pytest


def test_py_functional_helpers():
    from verl.utils.py_functional import append_to_dict, convert_dict_to_str, flatten_dict, is_package_available, unflatten_dict, union_two_dict
    flat = {"actor/pg_loss": 0.5, "actor/kl": 1e-5, "perf/mfu": 0.123456, "step": 3}
    nested = unflatten_dict(flat)
    assert nested == {"actor": {"pg_loss": 0.5, "kl": 1e-5}, "perf": {"mfu": 0.123456}, "step": 3}
    assert flatten_dict(nested) == flat and flatten_dict({"a": {"b": {"c": 1}}}, sep=".") == {"a.b.c": 1}
    d = {"x": [1]}
    append_to_dict(d, {"x": 2, "y": 3})
    assert d == {"x": [1, 2], "y": [3]}
    assert union_two_dict({"a": 1}, {"a": 1, "b": 2}) == {"a": 1, "b": 2}
    with pytest.raises(AssertionError):
        union_two_dict({"a": 1}, {"a": 2})
    text = convert_dict_to_str(nested)
    assert "mfu: 0.123" in text and "kl: 1.0e-05" in text and "pg_loss: 0.5" in text          # 3 decimals, scientific notation kept
    assert is_package_available("torch") and not is_package_available("no_such_package_xyz")


class _Recorder(types.ModuleType):
    def __init__(self, name):
        super().__init__(name)
        self.calls = []

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)

        def fn(*a, **k):
            self.calls.append((item, a, k))
            return (item, a, k)
        return fn


def test_tracker_calls_each_backend_as_the_reference_does(monkeypatch, capsys):
    import verl.utils.py_functional as pf
    fakes = {n: _Recorder(n) for n in ("wandb", "mlflow", "swanlab")}
    for n, m in fakes.items():
        monkeypatch.setitem(sys.modules, n, m)
    monkeypatch.setattr(pf, "is_package_available", lambda name: name in fakes)
    import verl.utils.logger.gen_logger as G
    import verl.utils.logger.logger as L
    monkeypatch.setattr(L, "is_package_available", pf.is_package_available)
    monkeypatch.setattr(G, "is_package_available", pf.is_package_available)
    monkeypatch.delenv("RANK", raising=False)
    cfg = {"trainer": {"project_name": "proj", "experiment_name": "exp", "logger": ["console", "wandb"]}, "data": {"seed": 1}}
    tr = L.Tracker(["console", "wandb", "mlflow", "swanlab", "tensorboard"], cfg)          # tensorboard: not "installed" -> skipped
    assert "tensorboard" in capsys.readouterr().out and len(tr.loggers) == 4
    tr.log({"actor/pg_loss": 0.25, "val/acc": 1}, step=7)
    assert "step 7: actor/pg_loss:0.25 - val/acc:1" in capsys.readouterr().out
    tr.log_generation([("q1", "a1", "l1", 0.5), ("q2", "a2", "l2", 1.0)], step=7)
    tr.log_generation([("q3", "a3", "l3", 0.0), ("q4", "a4", "l4", 1.0)], step=8)
    assert "score=0.5" in capsys.readouterr().out
    tr.finish()
    w = fakes["wandb"].calls
    assert w[0] == ("init", (), {"project": "proj", "name": "exp", "config": cfg})
    assert w[1] == ("log", (), {"data": {"actor/pg_loss": 0.25, "val/acc": 1}, "step": 7})
    cols = ["step", "input_1", "output_1", "label_1", "score_1", "input_2", "output_2", "label_2", "score_2"]
    assert w[2] == ("Table", (), {"columns": cols, "data": [[7, "q1", "a1", "l1", 0.5, "q2", "a2", "l2", 1.0]]})
    assert w[3][0] == "log" and w[3][2]["step"] == 7 and list(w[3][1][0]) == ["val/generations"]
    assert w[4] == ("Table", (), {"columns": cols, "data": [[7, "q1", "a1", "l1", 0.5, "q2", "a2", "l2", 1.0], [8, "q3", "a3", "l3", 0.0, "q4", "a4", "l4", 1.0]]})
    assert w[-1][0] == "finish"
    m = fakes["mlflow"].calls
    assert m[0] == ("start_run", (), {"run_name": "exp"}) and m[1][0] == "log_params" and m[1][1][0]["trainer/project_name"] == "proj"
    assert m[2] == ("log_metrics", (), {"metrics": {"actor/pg_loss": 0.25, "val/acc": 1}, "step": 7})
    s = fakes["swanlab"].calls
    assert s[0][0] == "init" and s[0][2]["project"] == "proj" and s[0][2]["experiment_name"] == "exp" and s[0][2]["config"]["FRAMEWORK"] == "veRL"
    assert s[1] == ("log", (), {"data": {"actor/pg_loss": 0.25, "val/acc": 1}, "step": 7}) and s[-1][0] == "finish"
    assert [c[0] for c in s].count("Text") == 4
    with pytest.raises(ValueError):
        L.Tracker(["console", "nonsense"], cfg)


def test_tracker_is_silent_on_other_ranks_and_skips_missing_packages(monkeypatch, capsys):
    from verl.utils.logger import Tracker
    monkeypatch.setenv("RANK", "1")
    tr = Tracker(("console", "wandb"), {"trainer": {"project_name": "p", "experiment_name": "e"}})
    tr.log({"a": 1.0}, 1)
    tr.log_generation([("q", "a", "l", 1.0)], 1)
    assert capsys.readouterr().out == "" and tr.loggers == []
    monkeypatch.setenv("RANK", "0")
    tr = Tracker(("console", "wandb"), {"trainer": {"project_name": "p", "experiment_name": "e"}})          # the shipped scripts' list; wandb absent
    out = capsys.readouterr().out
    assert "`wandb`" in out and "skipped" in out and len(tr.loggers) == 1
    tr.log({"a": 1.0}, 2)
    assert "step 2: a:1" in capsys.readouterr().out
