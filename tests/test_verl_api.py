"""CPU tests of the API mirror: config parsing of the shipped scripts' keys, DataProto semantics, balancing, reward manager."""
import json
import os

import numpy as np
import pytest
import torch

from verl.protocol import DataProto, pad_dataproto_to_divisor, unpad_dataproto
from verl.trainer.config import PPOConfig, load_config
from verl.utils.seqlen_balancing import get_seqlen_balanced_partitions, log_seqlen_unbalance

REF_YAML = """
data: {train_files: a@train, val_files: a@val, prompt_key: problem, answer_key: answer, image_key: image, max_prompt_length: 2048,
       max_response_length: 2048, rollout_batch_size: 512, val_batch_size: -1, shuffle: true, seed: 1, max_pixels: 4194304, min_pixels: 262144}
algorithm: {adv_estimator: grpo, disable_kl: false, use_kl_loss: true, kl_penalty: low_var_kl, kl_coef: 1.0e-2}
worker:
  actor:
    global_batch_size: 128
    micro_batch_size_per_device_for_update: 4
    micro_batch_size_per_device_for_experience: 16
    padding_free: true
    model: {model_path: Qwen/Qwen2.5-7B-Instruct, enable_gradient_checkpointing: true, trust_remote_code: false}
    optim: {lr: 1.0e-6, weight_decay: 1.0e-2, strategy: adamw, lr_warmup_ratio: 0.0}
    fsdp: {enable_full_shard: true, enable_rank0_init: true}
    offload: {offload_params: true, offload_optimizer: true}
  rollout: {temperature: 1.0, n: 5, tensor_parallel_size: 2, val_override_config: {temperature: 0.5, n: 1}}
  ref: {fsdp: {enable_cpu_offload: true}, offload: {offload_params: false}}
  reward: {reward_type: function, score_function: r1v, skip_special_tokens: true}
trainer: {total_episodes: 15, logger: [console, wandb], n_gpus_per_node: 4, val_freq: 5, save_freq: 5, save_limit: 1, save_checkpoint_path: null}
"""


def test_config_merge_like_the_launch_scripts(tmp_path):
    y = tmp_path / "config.yaml"
    y.write_text(REF_YAML)
    cfg = load_config([f"config={y}", "data.train_files=hub/STVQA-7K@train", "worker.actor.model.model_path=Qwen/Qwen2.5-VL-7B-Instruct",
                       "worker.reward.score_function=spatial_sgg", "trainer.experiment_name=spatialthinker10k_7B", "trainer.n_gpus_per_node=4",
                       "trainer.save_checkpoint_path=ckpts/x", "worker.actor.fsdp.torch_dtype=bf16", "worker.actor.optim.strategy=adamw_bf16",
                       "worker.rollout.n=8", "trainer.max_steps=75", "trainer.total_episodes=75", "data.answer_key=answer_option_text",
                       "data.image_key=images", "data.val_batch_size=8", "data.max_prompt_length=6144", "data.max_response_length=2048",
                       "worker.rollout.max_num_batched_tokens=8192", "data.format_prompt=<image> You FIRST think\n about it.\n Q. ",
                       "trainer.logger=['console','swanlab']"])
    cfg.deep_post_init()
    assert cfg.worker.rollout.n == 8 and cfg.worker.rollout.prompt_length == 6144 and cfg.worker.rollout.response_length == 2048
    assert cfg.worker.actor.use_kl_loss and cfg.worker.actor.kl_penalty == "low_var_kl" and cfg.worker.actor.kl_coef == 1e-2
    assert cfg.worker.ref.padding_free is True and cfg.worker.ref.micro_batch_size_per_device_for_experience == 16
    assert cfg.worker.actor.model.tokenizer_path == "Qwen/Qwen2.5-VL-7B-Instruct"
    assert cfg.trainer.logger == ("console", "swanlab") and cfg.trainer.max_steps == 75 and cfg.trainer.save_checkpoint_path == "ckpts/x"
    assert cfg.data.format_prompt == "<image> You FIRST think\n about it.\n Q. "          # multi-line strings survive verbatim
    assert cfg.worker.rollout.val_override_config == {"temperature": 0.5, "n": 1}
    assert PPOConfig().trainer.save_checkpoint_path is None
    d = PPOConfig(); d.deep_post_init()
    assert d.trainer.save_checkpoint_path == os.path.join("checkpoints", "easy_r1", "demo")
    with pytest.raises(KeyError):
        load_config(["data.no_such_key=1"])
    with pytest.raises(ValueError):
        load_config(["worker.rollout.n=abc"])
    json.dumps(cfg.to_dict())


def _proto(n=8):
    return DataProto.from_single_dict({"a": torch.arange(n * 2).view(n, 2), "b": torch.arange(n).float(),
                                       "s": np.array([f"s{i}" for i in range(n)], dtype=object)}, meta_info={"t": 1.0})


def test_dataproto_semantics():
    d = _proto(8)
    assert len(d) == 8 and d[2].batch["b"].item() == 2.0 and d[2].non_tensor_batch["s"] == "s2"
    parts = d.chunk(4)
    assert [len(p) for p in parts] == [2, 2, 2, 2] and parts[3].non_tensor_batch["s"].tolist() == ["s6", "s7"] and parts[0].meta_info == {"t": 1.0}
    with pytest.raises(AssertionError):
        d.chunk(3)
    back = DataProto.concat(parts)
    assert torch.equal(back.batch["a"], d.batch["a"]) and back.non_tensor_batch["s"].tolist() == d.non_tensor_batch["s"].tolist()
    r = d.repeat(3, interleave=True)
    assert r.batch["b"][:4].tolist() == [0, 0, 0, 1] and r.non_tensor_batch["s"][:4].tolist() == ["s0", "s0", "s0", "s1"]
    r2 = d.repeat(2, interleave=False)
    assert r2.batch["b"][8].item() == 0 and r2.non_tensor_batch["s"][9] == "s1"
    p = d.pop(batch_keys=["b"], non_tensor_batch_keys=["s"])
    assert "b" not in d.batch and "s" not in d.non_tensor_batch and len(p) == 8
    d.union(p)
    assert "b" in d.batch
    with pytest.raises(ValueError):
        d.union(DataProto.from_dict({"b": torch.zeros(8)}))
    d.reorder(torch.tensor([7, 6, 5, 4, 3, 2, 1, 0]))
    assert d.batch["b"][0].item() == 7 and d.non_tensor_batch["s"][0] == "s7"
    sel = d.select(batch_keys=["a"], non_tensor_batch_keys=[])
    assert list(sel.batch.keys()) == ["a"] and sel.non_tensor_batch == {}
    padded, pad = pad_dataproto_to_divisor(_proto(5), 4)
    assert len(padded) == 8 and pad == 3 and len(unpad_dataproto(padded, pad)) == 5
    with pytest.raises(AssertionError):
        DataProto.from_dict({"a": torch.zeros(3), "b": torch.zeros(4)})


def test_balanced_partitions_golden(golden_dir):
    for case in json.load(open(os.path.join(golden_dir, "balance.json"))):
        parts = get_seqlen_balanced_partitions(case["lens"], case["k"], equal_size=True)
        assert parts == case["parts"]
        stats = log_seqlen_unbalance(case["lens"], parts, "global_seqlen")
        assert stats["global_seqlen/balanced_max"] - stats["global_seqlen/balanced_min"] <= stats["global_seqlen/minmax_diff"] + max(case["lens"])


class _Tok:
    def decode(self, ids, skip_special_tokens=True):
        return "".join(chr(int(i)) for i in ids)


def test_reward_manager_places_score_on_last_valid_token():
    from verl.workers.reward import CustomRewardManager
    from verl.trainer.config import RewardConfig
    texts = ["<think>a</think> <answer>cat</answer>", "junk"]
    R = 48
    resp = torch.zeros(2, R, dtype=torch.long)
    mask = torch.zeros(2, R, dtype=torch.long)
    for i, t in enumerate(texts):
        resp[i, :len(t)] = torch.tensor([ord(c) for c in t]); mask[i, :len(t)] = 1
    data = DataProto.from_dict({"responses": resp, "response_mask": mask},
                               non_tensors={"ground_truth": np.array(["cat", "dog"], dtype=object), "problem": np.array(["p", "p"], dtype=object)})
    rm = CustomRewardManager(_Tok(), RewardConfig(score_function="r1v"))
    reward, metrics = rm(data)
    assert reward[0, len(texts[0]) - 1].item() == 1.0 and reward[0].sum().item() == 1.0 and reward[1].sum().item() == 0.0
    assert metrics["overall"] == [1.0, 0.0] and metrics["format"] == [1.0, 0.0]
    with pytest.raises(NotImplementedError):
        CustomRewardManager(_Tok(), RewardConfig(score_function="nope"))


def test_reference_import_paths_of_the_single_controller_resolve():
    """Third-party worker code written against the reference imports `verl.single_controller.base.Worker` and
    `verl.single_controller.base.decorator.{Dispatch, register}` (reference verl/workers/fsdp_workers.py:41-42)."""
    from verl.single_controller.base import ClassWithInitArgs, ResourcePool, Worker, WorkerGroup
    from verl.single_controller.base.decorator import Dispatch, Execute, register
    from verl.single_controller import decorator as D
    from verl.workers.fsdp_workers import FSDPWorker
    assert Dispatch is D.Dispatch and register is D.register and Execute is D.Execute
    assert issubclass(FSDPWorker, Worker)

    class Mine(Worker):
        @register(dispatch_mode=Dispatch.DP_COMPUTE_PROTO)
        def twice(self, x):
            return 2 * x

    wg = WorkerGroup(ResourcePool([1]), ClassWithInitArgs(Mine))
    assert wg.twice(21) == 42 and wg.world_size == 1 and wg.worker.rank == 0
    assert wg.execute_func_rank_zero(lambda a, b: a + b, 1, 2) == 3
    assert wg.worker.get_master_addr_port() == (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"))


def test_synthetic_lengths_travel_as_dataset_rows_not_as_a_worker_hook():
    """bench.py --through-api: `synthetic:stvqa:len=mu,sd@train` rows carry the forced response lengths; FSDPWorker has no benchmark hook."""
    import inspect
    from spatialthinker_amd.pretrained import synthetic_config
    from verl.utils.dataset import SyntheticSTVQADataset
    from verl.utils.tokenizer import get_tokenizer
    from verl.workers import fsdp_workers
    assert "ST_SYNTH" not in inspect.getsource(fsdp_workers)
    mcfg, _ = synthetic_config("random:tiny")
    ds = SyntheticSTVQADataset(mcfg, get_tokenizer("random:tiny"), size=8, max_prompt_length=128, grid=(1, 8, 8), text_tokens=(8, 12),
                               response_lengths=(16.0, 4.0, 4, 24))
    row = ds[3]
    lens = row["synthetic_response_lengths"]
    assert lens.shape == (4,) and lens.dtype == np.int64 and (lens >= 1).all() and (lens <= 24).all()
    assert np.array_equal(lens, ds[3]["synthetic_response_lengths"])                  # deterministic per row
    assert "synthetic_response_lengths" not in SyntheticSTVQADataset(mcfg, get_tokenizer("random:tiny"), size=8, max_prompt_length=128,
                                                                      grid=(1, 8, 8), text_tokens=(8, 12))[3]


def test_reference_worker_subpackage_import_paths_resolve():
    """The names the reference's workers import from their sub-packages (verl/workers/fsdp_workers.py:43-60: `from .actor import
    DataParallelPPOActor`, `.critic`, `.rollout`, `.config.WorkerConfig`, reward `CustomRewardManager`; each package's __init__ exports)
    exist under the same paths and are the objects the trainer uses."""
    import verl.trainer.config as C
    from verl.workers.actor import ActorConfig, BasePPOActor, DataParallelPPOActor, FSDPConfig, ModelConfig, OptimConfig, RefConfig
    from verl.workers.actor.config import OffloadConfig
    from verl.workers.config import WorkerConfig
    from verl.workers.critic import BasePPOCritic, CriticConfig, DataParallelPPOCritic
    from verl.workers.reward import CustomRewardManager, RewardConfig
    from verl.workers.rollout import BaseRollout, RolloutConfig
    assert ActorConfig is C.ActorConfig and RefConfig is C.RefConfig and CriticConfig is C.CriticConfig and WorkerConfig is C.WorkerConfig
    assert RolloutConfig is C.RolloutConfig and RewardConfig is C.RewardConfig and FSDPConfig is C.FSDPConfig and ModelConfig is C.ModelConfig
    assert OptimConfig is C.OptimConfig and OffloadConfig is C.OffloadConfig and CustomRewardManager is not None and BaseRollout is not None
    assert issubclass(DataParallelPPOActor, BasePPOActor) and issubclass(DataParallelPPOCritic, BasePPOCritic)
    for cls, methods in ((BasePPOActor, ("compute_log_prob", "update_policy")), (BasePPOCritic, ("compute_values", "update_critic"))):
        assert set(cls.__abstractmethods__) == set(methods)
    import pytest
    with pytest.raises(TypeError):                                       # the module argument is this build's engine, not an nn.Module
        DataParallelPPOActor(C.ActorConfig(), torch.nn.Linear(2, 2))
    with pytest.raises(TypeError):
        DataParallelPPOCritic(C.CriticConfig(), torch.nn.Linear(2, 2))


def test_reference_util_import_paths_resolve(tmp_path, capsys):
    """verl.utils.checkpoint.{CHECKPOINT_TRACKER, remove_obsolete_ckpt}, checkpoint_manager.find_latest_ckpt_path, verl.utils.torch_dtypes.
    PrecisionType, verl.utils.model_utils — the names the reference's trainer and workers import (ray_trainer.py:43, fsdp_workers.py:53-55)."""
    from verl.utils.checkpoint import CHECKPOINT_TRACKER, remove_obsolete_ckpt
    from verl.utils.checkpoint.checkpoint_manager import find_latest_ckpt_path, get_checkpoint_tracker_filename
    from verl.utils.model_utils import is_rank0, print_model_size
    from verl.utils.torch_dtypes import PrecisionType
    assert CHECKPOINT_TRACKER == "latest_global_step.txt" and find_latest_ckpt_path(None) is None and find_latest_ckpt_path(str(tmp_path)) is None
    for s in (2, 4, 6, 8):
        (tmp_path / f"global_step_{s}").mkdir()
    (tmp_path / "notes").mkdir()
    with open(get_checkpoint_tracker_filename(str(tmp_path)), "w") as f:
        f.write("8")
    assert find_latest_ckpt_path(str(tmp_path)) == str(tmp_path / "global_step_8")
    remove_obsolete_ckpt(str(tmp_path), 8, save_limit=2)                 # the step being written + ONE older
    assert sorted(p.name for p in tmp_path.iterdir() if p.is_dir()) == ["global_step_6", "global_step_8", "notes"]
    with open(get_checkpoint_tracker_filename(str(tmp_path)), "w") as f:
        f.write("10")
    assert find_latest_ckpt_path(str(tmp_path)) is None                 # tracker ahead of the directories
    assert PrecisionType.to_dtype("bf16") is torch.bfloat16 and PrecisionType.to_dtype(32) is torch.float32 and PrecisionType.to_dtype("fp16") is torch.float16
    assert PrecisionType.to_str(torch.bfloat16) == "bfloat16" and PrecisionType.is_bf16("bfloat16") and not PrecisionType.is_fp32("bf16")
    with pytest.raises(RuntimeError):
        PrecisionType.to_dtype("fp8")
    capsys.readouterr()
    assert is_rank0()
    print_model_size(torch.nn.Linear(1000, 2000), name="toy")
    assert "toy contains 2.00M parameters." in capsys.readouterr().out


def test_checkpoint_manager_classes_and_remaining_reference_names():
    """FSDPCheckpointManager / BaseCheckpointManager (fsdp_checkpoint_manager.py:34, checkpoint_manager.py:34), Role (ray_trainer.py:53-64),
    the r1v_scene module path and the config re-exports of verl.workers.config."""
    from verl.trainer.ray_trainer import Role
    from verl.utils.checkpoint import BaseCheckpointManager, FSDPCheckpointManager
    from verl.utils.reward_score.r1v_scene import r1v_scene_compute_score
    from verl.workers.config import FSDPConfig, ModelConfig, OptimConfig  # noqa: F401
    assert issubclass(FSDPCheckpointManager, BaseCheckpointManager) and set(BaseCheckpointManager.__abstractmethods__) == {"load_checkpoint", "save_checkpoint"}
    assert [r.name for r in Role] == ["Actor", "Rollout", "ActorRollout", "Critic", "RefPolicy", "RewardModel", "ActorRolloutRef"] and int(Role.ActorRolloutRef) == 7
    with pytest.raises(TypeError):
        FSDPCheckpointManager(torch.nn.Linear(2, 2))
    assert r1v_scene_compute_score("x", "y")["overall"] == 0.0
    st = BaseCheckpointManager.get_rng_state()
    a = torch.rand(3)
    BaseCheckpointManager.load_rng_state(st)
    assert torch.equal(torch.rand(3), a)


def test_dataproto_iterator_collate_fold_and_print(capsys):
    """DataProto.make_iterator / print_size and protocol.{collate_fn, batch_collate, fold_batch_dim, union_tensor_dict, union_numpy_dict}
    (reference verl/protocol.py:84-155,224-238,447-486)."""
    from verl.protocol import DataProtoItem, batch_collate, collate_fn, fold_batch_dim, union_numpy_dict, union_tensor_dict
    n = 12
    d = DataProto.from_dict({"x": torch.arange(n * 3).view(n, 3), "y": torch.arange(n).float()},
                            non_tensors={"s": np.array([f"r{i}" for i in range(n)], dtype=object), "l": np.array([[i, i] for i in range(n)], dtype=object)},
                            meta_info={"temperature": 0.7})
    # mini-batches in order, twice
    seen = list(d.make_iterator(mini_batch_size=4, epochs=2))
    assert len(seen) == 6 and all(len(b) == 4 and b.meta_info == {"temperature": 0.7} for b in seen)
    assert torch.equal(torch.cat([b.batch["y"] for b in seen[:3]]), d.batch["y"]) and list(seen[4].non_tensor_batch["s"]) == ["r4", "r5", "r6", "r7"]
    assert seen[0].non_tensor_batch["l"].dtype == object and seen[0].non_tensor_batch["l"].shape[0] == 4
    # shuffled: a permutation, reproducible from the seed, different between seeds
    perm = lambda seed: torch.cat([b.batch["y"] for b in d.make_iterator(4, 1, seed=seed, dataloader_kwargs={"shuffle": True})])
    a, b, c = perm(3), perm(3), perm(4)
    assert torch.equal(a, b) and not torch.equal(a, c) and sorted(a.tolist()) == list(range(n)) and not torch.equal(a, d.batch["y"])
    with pytest.raises(AssertionError):
        d.make_iterator(5, 1)
    # collate of items = the rows again
    back = collate_fn([d[i] for i in (2, 5)])
    assert isinstance(d[2], DataProtoItem) and torch.equal(back.batch["x"], d.batch["x"][[2, 5]]) and list(back.non_tensor_batch["s"]) == ["r2", "r5"]
    assert batch_collate([{"a": 1, "b": 2}, {"a": 3}]) == {"a": [1, 3], "b": [2]} and batch_collate([]) == {}
    f = fold_batch_dim(d, 3)
    assert len(f) == 3 and f.batch["x"].shape == (3, 4, 3) and f.non_tensor_batch["s"].shape == (3, 4) and f.non_tensor_batch["s"][1, 0] == "r4"
    d.print_size(prefix="batch")
    out = capsys.readouterr().out
    assert out.startswith("batch Size of tensordict:") and "non_tensor_batch" in out
    u = union_tensor_dict(tensor_dict1=d.batch.select("x"), tensor_dict2=d.batch.select("y"))
    assert set(u.keys()) == {"x", "y"}
    with pytest.raises(ValueError):
        union_numpy_dict(tensor_dict1={"s": np.array(["a"], dtype=object)}, tensor_dict2={"s": np.array(["b"], dtype=object)})
