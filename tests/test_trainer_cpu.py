"""The driver loop (verl/trainer/ray_trainer.py, mirror of the reference's RayPPOTrainer.fit :565-706) on CPU with a stub worker:
row alignment of uid / repeat / union, the balance reorder, the resumable rank-sharded dataloader, checkpoint retention +
resume, data-parallel validation and — on two gloo ranks — the cross-rank metric gathering.  adv_estimator=rloo keeps the step
on host math (GRPO's advantage kernel needs the GPU; it has its own -m gpu tests)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch.utils.data import Dataset

from verl.protocol import DataProto
from verl.trainer.config import load_config
from verl.trainer.ray_trainer import RayPPOTrainer, remove_obsolete_ckpt
from verl.utils.dataloader import ResumableDataLoader
from verl.utils.dataset import collate_fn
from verl.utils.seqlen_balancing import get_seqlen_balanced_partitions, log_seqlen_unbalance
from verl.workers.rollout import assemble_rollout_batch

P, R, EOS, PAD = 6, 5, 2, 0


class Rows(Dataset):
    """Row i: prompt tokens 100+i (length 2 + i % 4, left-padded to P); ground truth names the row."""

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        L = 2 + i % 4
        ids = torch.full((P,), PAD, dtype=torch.long); ids[P - L:] = 100 + i
        am = torch.zeros(P, dtype=torch.long); am[P - L:] = 1
        pos = torch.clip(am.cumsum(0) - 1, min=0)[None, :].repeat(3, 1)
        return {"input_ids": ids, "attention_mask": am, "position_ids": pos, "raw_prompt_ids": [100 + i] * L,
                "ground_truth": f"gt{i}", "problem": f"p{i}", "row": i}


class Tok:
    pad_token_id = PAD

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(str(int(t)) for t in ids if int(t) != PAD)


class StubWorkerGroup:
    """generate: sample j of prompt row i answers with j+1 copies of token (100+i)%50+10, then EOS; log-probs are functions of the
    ids; update_actor records what it was handed."""

    def __init__(self):
        self.world_size = dist.get_world_size() if dist.is_initialized() else 1
        self.updates, self.saved, self.loaded, self.gen_meta, self.logprob_rows, self.gen_rows = [], [], [], [], [], []

    def gather_objects(self, obj):
        if self.world_size == 1:
            return [obj]
        out = [None] * self.world_size
        dist.all_gather_object(out, obj)
        return out

    def init_model(self):
        pass

    def generate_sequences(self, prompts: DataProto) -> DataProto:
        n = int(prompts.meta_info.get("n", self.n))
        self.gen_meta.append(dict(prompts.meta_info))
        ids = prompts.batch["input_ids"]
        self.gen_rows.append([int(ids[i, -1]) - 100 for i in range(len(prompts)) for _ in range(n)])
        rows = []
        for i in range(len(prompts)):
            tok = int(ids[i, -1]) % 50 + 10
            for j in range(n):
                k = min(j + 1, R - 1)
                rows.append([tok] * k + [EOS] + [PAD] * (R - k - 1))
        resp = torch.tensor(rows)
        return DataProto.from_dict(assemble_rollout_batch(ids, prompts.batch["attention_mask"], prompts.batch["position_ids"], resp, n, EOS))

    def compute_log_probs(self, data):
        self.logprob_rows.append(list(data.non_tensor_batch["ground_truth"]))
        return DataProto.from_dict({"old_log_probs": -data.batch["responses"].float() / 100.0},
                                   meta_info={"temperature": 1.0, "prompt_cache_hit": True})

    def compute_ref_log_probs(self, data):
        return DataProto.from_dict({"ref_log_probs": -data.batch["responses"].float() / 90.0})

    def update_actor(self, data):
        self.updates.append(data)
        k = len(data) // 2
        return DataProto(non_tensor_batch={"actor/pg_loss": np.array([float(data.batch["advantages"].sum())] * 2),
                                           "actor/grad_norm": np.array([float(len(data))])})

    def save_checkpoint(self, path):
        self.saved.append(path)
        if int(os.environ.get("RANK", 0)) == 0:
            open(os.path.join(path, "weights.bin"), "w").write("w")

    def load_checkpoint(self, path):
        self.loaded.append(path)


def reward_fn(data: DataProto):
    """score = number of non-EOS response tokens / 10 at the last valid token."""
    m = data.batch["response_mask"]
    n = m.sum(-1)
    rew = torch.zeros(m.shape, dtype=torch.float32)
    rew[torch.arange(len(data)), n - 1] = (n - 1).float() / 10.0
    return rew, {"overall": ((n - 1).float() / 10.0).tolist(), "format": [1.0] * len(data)}


def make_trainer(tmp, extra=(), n_rows=16, val_rows=0, n=2):
    cfg = load_config(["data.rollout_batch_size=4", "data.max_prompt_length=6", "data.max_response_length=5", "data.shuffle=true", "data.seed=3",
                       "worker.actor.global_batch_size=2", "worker.actor.micro_batch_size_per_device_for_update=2",
                       "worker.actor.micro_batch_size_per_device_for_experience=2", f"worker.rollout.n={n}", "algorithm.adv_estimator=rloo",
                       "trainer.total_episodes=2", "trainer.val_before_train=false", "trainer.logger=['console']",
                       f"trainer.save_checkpoint_path={tmp}/ckpt"] + list(extra))
    cfg.deep_post_init()
    world = int(os.environ.get("WORLD_SIZE", 1))
    cfg.worker.actor.global_batch_size_per_device = cfg.worker.actor.global_batch_size * n // world     # FSDPWorker._init_batch_sizes
    wg = StubWorkerGroup(); wg.n = n
    tr = RayPPOTrainer(cfg, Tok(), None, wg, wg, reward_fn, reward_fn, Rows(n_rows), Rows(val_rows) if val_rows else None)
    return tr, wg, cfg


# ------------------------------------------------------------------------------------------------ dataloader
def test_dataloader_order_is_random_sampler_order_and_shards_are_dataproto_chunks():
    ds = Rows(10)
    g = torch.Generator().manual_seed(5)
    want = [torch.randperm(10, generator=g).tolist() for _ in range(2)]                 # what RandomSampler(generator=g) yields per epoch
    for world in (1, 2):
        got = [[], []]
        loaders = [ResumableDataLoader(ds, 4, True, 5, collate_fn, True, 0, r, world) for r in range(world)]
        for ep in range(2):
            per_rank = [[b["row"].tolist() for b in ld] for ld in loaders]
            assert all(len(x) == 2 for x in per_rank)                                   # drop_last: 10 // 4
            for bi in range(2):
                glob = want[ep][bi * 4:(bi + 1) * 4]
                chunks = DataProto.from_single_dict({"row": torch.tensor(glob)}).chunk(world)
                for r in range(world):
                    assert per_rank[r][bi] == chunks[r].batch["row"].tolist()


def test_dataloader_resume_continues_with_the_next_batch():
    ds = Rows(12)
    ref = ResumableDataLoader(ds, 4, True, 9, collate_fn)
    full = [b["row"].tolist() for _ in range(3) for b in ref]                           # 3 epochs x 3 batches
    for stop in (1, 2, 3, 4, 7):                                                        # mid-epoch, epoch end, later epochs
        a = ResumableDataLoader(ds, 4, True, 9, collate_fn)
        seen, state = [], None
        for _ in range(3):
            for b in a:
                seen.append(b["row"].tolist())
                if len(seen) == stop:
                    state = a.state_dict()
                    break
            if state is not None:
                break
        b_ = ResumableDataLoader(ds, 4, True, 9, collate_fn)
        b_.load_state_dict(state)
        rest = []
        while len(seen) + len(rest) < len(full):
            for item in b_:
                rest.append(item["row"].tolist())
                if len(seen) + len(rest) == len(full):
                    break
        assert seen + rest == full, stop
    with pytest.raises(ValueError):
        ResumableDataLoader(Rows(11), 4, True, 9, collate_fn).load_state_dict(state)


# ------------------------------------------------------------------------------------------------ one step, single rank
def test_step_row_alignment_balance_and_metrics(tmp_path, capsys):
    tr, wg, cfg = make_trainer(tmp_path, ["trainer.max_steps=1", "worker.actor.global_batch_size=1"], n=2)
    cfg.worker.actor.global_batch_size_per_device = 2                       # 1 prompt x n=2 rows per optimizer step: 4 mini-batches
    tr.fit()
    assert len(wg.updates) == 1
    d = wg.updates[0]
    b, nt = d.batch, d.non_tensor_batch
    assert len(d) == 8
    for i in range(0, 8, 2):                                                # groups stay contiguous after the reorder
        assert nt["uid"][i] == nt["uid"][i + 1] and nt["ground_truth"][i] == nt["ground_truth"][i + 1]
        row = int(nt["ground_truth"][i][2:])
        for j in (0, 1):                                                    # prompt, ground truth and BOTH responses belong to the same row
            assert int(b["prompts"][i + j, -1]) == 100 + row
            tok = (100 + row) % 50 + 10
            assert b["responses"][i + j, 0].item() == tok and int(b["response_mask"][i + j].sum()) == j + 2
            assert torch.equal(b["input_ids"][i + j], torch.cat([b["prompts"][i + j], b["responses"][i + j]]))
    assert len(set(nt["uid"])) == 4
    # the old / ref log-prob passes see the rows in GENERATION order (row r <-> prompt r // n: what the rollout's prompt K/V cache is
    # keyed by); the mini-batch balance permutation is applied afterwards, for update_actor only
    assert [int(g[2:]) for g in wg.logprob_rows[0]] == wg.gen_rows[0]
    assert [int(g[2:]) for g in nt["ground_truth"]] != wg.gen_rows[0]       # ... and update_actor got the Karmarkar-Karp group order
    # default config: use_kl_loss off -> the KL penalty branch (ray_trainer.py:658-665) shapes the rewards; RLOO inside each group
    # of 2 is then A = r_i - r_other on the PENALISED rewards
    assert not torch.equal(b["token_level_rewards"], b["token_level_scores"])
    sc = b["token_level_rewards"].sum(-1)
    for i in range(0, 8, 2):
        assert abs(float(b["advantages"][i, 0]) - float(sc[i] - sc[i + 1])) < 1e-6
    # the group order handed to update_actor is the Karmarkar-Karp order over the group token sums (4 mini-batches of 1 group)
    lens = b["attention_mask"].sum(-1).tolist()
    out = capsys.readouterr().out
    line = [l for l in out.splitlines() if l.startswith("step 1:")][0]
    assert "perf/prompt_cache_hit:1" in line
    for key in ("global_seqlen/balanced_max", "global_seqlen/minmax_diff", "minibatch_seqlen/balanced_max", "reward/overall", "actor/pg_loss",
                "critic/score/mean", "response_length/mean", "timing_s/gen", "timing_s/update_actor", "perf/throughput", "perf/total_num_tokens"):
        assert key in line, key
    assert f"perf/total_num_tokens:{sum(lens)}" in line


def test_minibatch_balance_reorder_is_karmarkar_karp_over_groups(tmp_path):
    tr, wg, cfg = make_trainer(tmp_path, n=2)
    cfg.worker.actor.global_batch_size_per_device = 4                       # 2 groups per mini-batch, 2 mini-batches
    lens_g = [(9, 3), (4, 4), (7, 7), (2, 3)]                              # (rows of group g)
    am = torch.zeros(8, 12, dtype=torch.long)
    for g, pair in enumerate(lens_g):
        for j, L in enumerate(pair):
            am[2 * g + j, :L] = 1
    batch = DataProto.from_dict({"attention_mask": am, "tag": torch.arange(8)}, non_tensors={"uid": np.repeat(np.array(list("abcd"), dtype=object), 2)})
    metrics = {}
    tr._balance_batch(batch, metrics)
    sums = [sum(p) for p in lens_g]
    parts = get_seqlen_balanced_partitions(sums, 2, True)
    want = [2 * g + j for p in parts for g in p for j in (0, 1)]
    assert batch.batch["tag"].tolist() == want
    assert batch.non_tensor_batch["uid"].tolist() == [("abcd"[t // 2]) for t in want]
    flat = [L for p in lens_g for L in p]
    assert {k: v for k, v in metrics.items() if k.startswith("global_seqlen")} == log_seqlen_unbalance(flat, [list(range(8))], "global_seqlen")
    assert metrics["minibatch_seqlen/balanced_max"] == max(sum(sums[g] for g in p) for p in parts)


# ------------------------------------------------------------------------------------------------ checkpoints
def test_remove_obsolete_ckpt_keeps_save_limit(tmp_path):
    for s in (2, 4, 6, 8):
        os.makedirs(tmp_path / f"global_step_{s}")
    os.makedirs(tmp_path / "other")
    remove_obsolete_ckpt(str(tmp_path), 10, save_limit=3)                   # room for the new one: keep the 2 newest older steps
    assert sorted(os.listdir(tmp_path)) == ["global_step_6", "global_step_8", "other"]
    remove_obsolete_ckpt(str(tmp_path), 10, save_limit=-1)
    assert sorted(os.listdir(tmp_path)) == ["global_step_6", "global_step_8", "other"]


def test_save_limit_and_resume_replays_nothing(tmp_path, capsys):
    tr, wg, cfg = make_trainer(tmp_path, ["trainer.max_steps=5", "trainer.save_freq=2", "trainer.save_limit=1"], n_rows=16)
    tr.fit()
    seen_a = [sorted(set(int(g[2:]) for g in u.non_tensor_batch["ground_truth"])) for u in wg.updates]
    root = tmp_path / "ckpt"
    # saves at steps 2 and 4; like the reference the counter ends at max_steps + 1 = 6, a multiple of save_freq, so no final save
    assert (root / "latest_global_step.txt").read_text() == "4"
    assert sorted(p for p in os.listdir(root) if p.startswith("global_step_")) == ["global_step_4"]       # save_limit=1
    assert os.path.exists(root / "global_step_4" / "dataloader.pt") and os.path.exists(root / "global_step_4" / "actor" / "weights.bin")
    # a run stopped after step 2, resumed from its checkpoint, must see exactly the batches 3, 4, 5 of the uninterrupted run
    tr1, wg1, _ = make_trainer(tmp_path / "b", ["trainer.max_steps=2", "trainer.save_freq=2"], n_rows=16)
    tr1.fit()
    ck = tmp_path / "b" / "ckpt" / "global_step_2"
    assert os.path.exists(ck / "dataloader.pt")
    tr2, wg2, _ = make_trainer(tmp_path / "c", ["trainer.max_steps=5", f"trainer.load_checkpoint_path={ck}"], n_rows=16)
    tr2.fit()
    assert wg2.loaded == [str(ck / "actor")]
    seen_c = [sorted(set(int(g[2:]) for g in u.non_tensor_batch["ground_truth"])) for u in wg2.updates]
    assert seen_c == seen_a[2:] and len(seen_c) == 3
    out = capsys.readouterr().out
    assert "step 3:" in out and "step 5:" in out
    with pytest.raises(ValueError):
        make_trainer(tmp_path / "d", ["trainer.load_checkpoint_path=/tmp/not_a_step_dir"])[0].fit()
    # a state of THIS loader saved for another dataset size must STOP the resume (it used to restart the data order with a
    # print, after kl_coef had already been overwritten); only another implementation's dataloader.pt restarts the order
    tr3, _, _ = make_trainer(tmp_path / "e", ["trainer.max_steps=5", f"trainer.load_checkpoint_path={ck}"], n_rows=32)
    kl_before = tr3.kl_ctrl.kl_coef
    with pytest.raises(ValueError, match="different dataset size"):
        tr3.fit()
    assert tr3.kl_ctrl.kl_coef == kl_before
    import shutil
    ck_f = tmp_path / "f" / "global_step_2"
    shutil.copytree(ck, ck_f)
    torch.save({"_snapshot": {"_main_snapshot": {}}, "_steps_since_snapshot": 0}, ck_f / "dataloader.pt")      # StatefulDataLoader-shaped
    tr4, wg4, _ = make_trainer(tmp_path / "g", ["trainer.max_steps=3", f"trainer.load_checkpoint_path={ck_f}"], n_rows=16)
    tr4.fit()
    assert "the data order starts from scratch" in capsys.readouterr().out and len(wg4.updates) == 1


# ------------------------------------------------------------------------------------------------ validation
def test_validate_metric_names_override_and_final_validation(tmp_path, capsys):
    tr, wg, cfg = make_trainer(tmp_path, ["trainer.max_steps=1", "trainer.val_before_train=true", "data.val_batch_size=2", "trainer.val_generations_to_log=2",
                                          "worker.rollout.val_override_config={'temperature': 0.5, 'n': 1}"], val_rows=5)
    os.environ["ST_SKIP_FINAL_SAVE"] = "1"
    try:
        tr.fit()
    finally:
        del os.environ["ST_SKIP_FINAL_SAVE"]
    out = capsys.readouterr().out
    val_calls = [m for m in wg.gen_meta if m.get("temperature") == 0.5]
    assert len(val_calls) == 2 * 3 and all(m["n"] == 1 for m in val_calls)            # before training + after training, 3 batches of <= 2 rows
    # n = 1 completion per row: 1 token + EOS -> score 0.1 for each of the 5 rows
    assert "step 0: val/format_reward:1 - val/overall_reward:0.1 - val/reward_score:0.1" in out
    assert "Final validation metrics: " in out and out.count("[val generation @ step") == 4


# ------------------------------------------------------------------------------------------------ two ranks (gloo)
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _two_rank_worker(rank, world, port, tmp):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ST_SKIP_FINAL_SAVE="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import contextlib, io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        tr, wg, cfg = make_trainer(tmp, ["trainer.max_steps=1", "trainer.val_before_train=true", "data.val_batch_size=-1",
                                         "worker.rollout.val_override_config={'n': 1}"], n_rows=16, val_rows=5, n=2)
        tr.fit()
    d = wg.updates[0]
    torch.save({"out": buf.getvalue(), "rows": sorted(set(int(g[2:]) for g in d.non_tensor_batch["ground_truth"])),
                "lens": d.batch["attention_mask"].sum(-1).tolist(), "scores": d.batch["token_level_scores"].sum(-1).tolist(),
                "val_gen_rows": [len(m) for m in wg.gen_meta]}, os.path.join(tmp, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_two_ranks_shard_rows_and_rank0_logs_global_metrics(tmp_path):
    world = 2
    mp.spawn(_two_rank_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    g = torch.Generator().manual_seed(3)
    first = torch.randperm(16, generator=g).tolist()[:4]
    assert r0["rows"] == sorted(first[:2]) and r1["rows"] == sorted(first[2:])        # rank r gets chunk r of the global batch
    assert r1["out"].count("step 1:") == 0                                            # only rank 0 logs
    line = [l for l in r0["out"].splitlines() if l.startswith("step 1:")][0]
    kv = dict(item.split(":", 1) for item in line[len("step 1: "):].split(" - "))
    lens = r0["lens"] + r1["lens"]
    # balance statistics over the GLOBAL list with k = world (the balanced sums are order-independent; min/max are rank sums)
    parts = get_seqlen_balanced_partitions(lens, 2, True)
    bal = [sum(lens[i] for i in p) for p in parts]
    assert float(kv["global_seqlen/balanced_max"]) == max(bal) and float(kv["global_seqlen/balanced_min"]) == min(bal)
    assert float(kv["global_seqlen/max"]) == max(sum(r0["lens"]), sum(r1["lens"]))
    assert float(kv["perf/total_num_tokens"]) == sum(lens)
    scores = r0["scores"] + r1["scores"]
    assert abs(float(kv["critic/score/mean"]) - float(np.mean(scores))) < 1e-4 and abs(float(kv["reward/overall"]) - float(np.mean(scores))) < 1e-4
    assert abs(float(kv["actor/grad_norm"]) - 4.0) < 1e-6                             # every rank reported its 4 rows; the mean of the gathered list
    # validation: 5 rows -> one batch padded to 6, 3 per rank; the padded duplicate is dropped from the score
    v = [l for l in r0["out"].splitlines() if l.startswith("step 0:")][0]
    assert "val/reward_score:0.1" in v and "val/overall_reward:0.1" in v


def _eight_rank_worker(rank, world, port, tmp):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ST_SKIP_FINAL_SAVE="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import contextlib, io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        tr, wg, cfg = make_trainer(tmp, ["trainer.max_steps=2", "data.rollout_batch_size=16", "worker.actor.global_batch_size=8"], n_rows=64, n=2)
        tr.fit()
    res = {"out": buf.getvalue(), "steps": []}
    for d in wg.updates:
        res["steps"].append({"rows": [int(g[2:]) for g in d.non_tensor_batch["ground_truth"]], "lens": d.batch["attention_mask"].sum(-1).tolist(),
                             "scores": d.batch["token_level_scores"].sum(-1).tolist(), "adv": d.batch["advantages"].clone()})
    torch.save(res, os.path.join(tmp, f"e{rank}.pt"))
    dist.destroy_process_group()


def test_eight_ranks_shard_rows_as_dataproto_chunks_and_rank0_logs_global_metrics(tmp_path):
    """BASELINE config #4's partitioning (DP = 8; reference verl/single_controller/base/decorator.py:106-123 `dispatch_dp_compute_data_proto`
    = DataProto.chunk(world), fsdp_workers.py:130-136 batch arithmetic) without the hardware: 8 gloo ranks drive RayPPOTrainer.fit for two
    steps; rank r works on chunk r of the step's global batch (2 prompts x n = 2 rollouts), the groups of a prompt never straddle ranks,
    and rank 0's log line carries the GLOBAL statistics."""
    world = 8
    mp.spawn(_eight_rank_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"e{k}.pt", weights_only=False) for k in range(world)]
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(3)).tolist()
    for step in range(2):
        batch = perm[16 * step:16 * (step + 1)]
        lens, scores = [], []
        for k in range(world):
            st = r[k]["steps"][step]
            assert sorted(set(st["rows"])) == sorted(batch[2 * k:2 * k + 2]) and len(st["rows"]) == 4      # chunk k, both rollouts of each prompt
            lens += st["lens"]; scores += st["scores"]
            assert st["adv"].shape[0] == 4
        line = [l for l in r[0]["out"].splitlines() if l.startswith(f"step {step + 1}:")][0]
        kv = dict(item.split(":", 1) for item in line[len(f"step {step + 1}: "):].split(" - "))
        assert float(kv["perf/total_num_tokens"]) == sum(lens)
        parts = get_seqlen_balanced_partitions(lens, world, True)
        bal = [sum(lens[i] for i in p) for p in parts]
        assert float(kv["global_seqlen/balanced_max"]) == max(bal) and float(kv["global_seqlen/balanced_min"]) == min(bal)
        assert abs(float(kv["critic/score/mean"]) - float(np.mean(scores))) < 1e-4
        assert abs(float(kv["actor/grad_norm"]) - 4.0) < 1e-6
    for k in range(1, world):
        assert r[k]["out"].count("step 1:") == 0


def _migrate_worker(rank, world, port, tmp):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ST_SKIP_FINAL_SAVE="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        tr, wg, cfg = make_trainer(tmp, ["trainer.max_steps=1", "trainer.balance_mode=migrate", "data.rollout_batch_size=8",
                                         "worker.actor.global_batch_size=8", "algorithm.adv_estimator=rloo", "algorithm.use_kl_loss=true"], n_rows=16, n=2)
        tr.fit()
    d = wg.updates[0]
    torch.save({"gt": list(d.non_tensor_batch["ground_truth"]), "lens": d.batch["attention_mask"].sum(-1).tolist(),
                "resp_len": d.batch["response_mask"].sum(-1).tolist(), "adv": d.batch["advantages"][:, 0].tolist(),
                "scores": d.batch["token_level_scores"].sum(-1).tolist(), "old": d.batch["old_log_probs"].clone(),
                "resp": d.batch["responses"].clone(), "gen_rows": wg.gen_rows[0]}, os.path.join(tmp, f"m{rank}.pt"))
    dist.destroy_process_group()


def test_balance_mode_migrate_reproduces_the_reference_row_migration(tmp_path):
    """trainer.balance_mode=migrate on two gloo ranks: the global (rank-major) row list is cut into world_size equal-size sets by the
    golden-pinned Karmarkar-Karp partitioner (tests/golden/balance.json pins it against the reference's own function) and rank r trains
    on set r in partition order — verl/trainer/ray_trainer.py:526-541.  Group statistics (RLOO here) still cover whole groups although
    their members now sit on different ranks, and every per-row tensor travelled with its row."""
    world = 2
    mp.spawn(_migrate_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"m{i}.pt") for i in range(world)]
    # global order before the migration: rank 0's generated rows, then rank 1's (prompt-major, n = 2 rollouts each)
    glob_rows = r[0]["gen_rows"] + r[1]["gen_rows"]
    assert len(glob_rows) == 16
    P_len = {row: 2 + row % 4 for row in set(glob_rows)}
    resp_len = [2 + (j % 2) for j in range(16)]                              # sample j of a prompt answers with j + 1 tokens + EOS
    lens = [P_len[row] + resp_len[j] for j, row in enumerate(glob_rows)]
    parts = get_seqlen_balanced_partitions(lens, 2, True)
    for rank in range(world):
        want = [(glob_rows[j], resp_len[j]) for j in parts[rank]]
        got = [(int(g[2:]), int(L)) for g, L in zip(r[rank]["gt"], r[rank]["resp_len"])]
        assert got == want, rank
        assert r[rank]["lens"] == [lens[j] for j in parts[rank]]
        assert torch.equal(r[rank]["old"], -r[rank]["resp"].float() / 100.0)          # old_log_probs belong to the migrated rows
    # the two rollouts of a prompt may now live on different ranks; RLOO (k = 2): A_i = r_i - r_other over the WHOLE group
    by_row = {}
    for rank in range(world):
        for g, sc, adv in zip(r[rank]["gt"], r[rank]["scores"], r[rank]["adv"]):
            by_row.setdefault(g, []).append((sc, adv))
    split_groups = 0
    for g, items in by_row.items():
        assert len(items) == 2
        (s0, a0), (s1, a1) = items
        assert abs(a0 - (s0 - s1)) < 1e-5 and abs(a1 - (s1 - s0)) < 1e-5
    for rank in range(world):
        split_groups += sum(1 for g in set(r[rank]["gt"]) if r[rank]["gt"].count(g) == 1)
    assert split_groups > 0                                                  # the migration really scattered at least one group


def test_dataloader_state_saved_after_an_epoch_end_draws_a_new_permutation():
    """A checkpoint written after the loops end naturally (training_steps = len x episodes) stores the state BETWEEN epochs: a
    resume must continue with the NEXT epoch's permutation (StatefulDataLoader semantics), not replay the finished one."""
    ds = Rows(12)
    ref = ResumableDataLoader(ds, 4, True, 9, collate_fn)
    full = [b["row"].tolist() for _ in range(3) for b in ref]
    a = ResumableDataLoader(ds, 4, True, 9, collate_fn)
    first = [b["row"].tolist() for b in a]                              # one whole epoch, exhausted naturally
    state = a.state_dict()
    assert state["batches_yielded"] == 0 and state["epochs_done"] == 1
    b_ = ResumableDataLoader(ds, 4, True, 9, collate_fn)
    b_.load_state_dict(state)
    second = [b["row"].tolist() for b in b_]
    assert first == full[:3] and second == full[3:6] and second != first


@pytest.mark.parametrize("rows,world", [(1, 8), (3, 8), (5, 4), (7, 8), (2, 2)])
def test_ragged_last_batch_is_padded_cyclically_to_the_world_size(rows, world):
    """drop_last=False validation loaders: a last batch of 1..world-1 rows is padded by cycling its rows until every rank gets the
    same number (>= 1) — pad_dataproto_to_divisor of the reference (verl/protocol.py:48-66)."""
    ds = Rows(rows)
    shards = []
    for r in range(world):
        ld = ResumableDataLoader(ds, -(-rows // world) * world, False, 1, collate_fn, False, 0, r, world)
        items = [b["row"].tolist() for b in ld]
        assert len(items) == 1 and len(items[0]) >= 1
        shards.append(items[0])
    assert len({len(x) for x in shards}) == 1
    flat = [x for sh in shards for x in sh]
    want = (list(range(rows)) * (len(flat) // rows + 1))[:len(flat)]
    assert flat == want


# ------------------------------------------------------------------------------------------------ adv_estimator = gae (critic)
class StubCriticGroup(StubWorkerGroup):
    """values = 0.1 * (response slot index + 1) on valid slots; update_critic records the batch it was handed."""

    def __init__(self):
        super().__init__()
        self.inits, self.value_calls, self.critic_updates = 0, 0, []

    def init_model(self):
        self.inits += 1

    def compute_values(self, data):
        self.value_calls += 1
        m = data.batch["response_mask"].float()
        v = 0.1 * torch.arange(1, m.shape[1] + 1, dtype=torch.float32)[None, :] * m
        return DataProto.from_dict({"values": v})

    def update_critic(self, data):
        self.critic_updates.append(data)
        return DataProto(non_tensor_batch={"critic/vf_loss": np.array([0.25, 0.75]), "critic/grad_norm": np.array([1.5])})


def test_gae_runs_values_advantages_critic_update_then_actor_update(tmp_path, capsys):
    """ray_trainer.py:230-233, 248-257, 428-434, 644-675, 483-517: with adv_estimator=gae the step is gen -> reward -> old (-> ref) -> values
    -> GAE (core_algos.compute_gae_advantage_return on the critic's values) -> update_critic -> update_actor, the critic is initialised
    first and checkpointed next to the actor; without a critic worker group the trainer refuses to start."""
    from verl.trainer import core_algos
    extra = ["algorithm.adv_estimator=gae", "algorithm.gamma=1.0", "algorithm.lam=0.9", "algorithm.disable_kl=true", "worker.critic.global_batch_size=2",
             "worker.critic.micro_batch_size_per_device_for_update=2", "worker.critic.micro_batch_size_per_device_for_experience=2", "trainer.max_steps=2",
             "trainer.save_freq=1"]
    tr, wg, cfg = make_trainer(tmp_path, extra=extra)
    assert tr.use_critic and cfg.worker.critic.optim.training_steps == tr.training_steps
    with pytest.raises(ValueError, match="critic worker group"):
        tr.set_worker_groups(wg, wg)
    cg = StubCriticGroup(); cg.n = wg.n
    tr.set_worker_groups(wg, wg, cg)
    tr.init_workers()
    assert cg.inits == 1
    tr.fit()
    out = capsys.readouterr().out
    assert cg.value_calls == 2 and len(cg.critic_updates) == 2 and len(wg.updates) == 2
    b = cg.critic_updates[-1]
    assert {"values", "returns", "advantages", "token_level_rewards"} <= set(b.batch.keys())
    adv, ret = core_algos.compute_gae_advantage_return(b.batch["token_level_rewards"], b.batch["values"], b.batch["response_mask"], 1.0, 0.9)
    assert torch.allclose(b.batch["returns"], ret) and torch.allclose(b.batch["advantages"], adv)
    assert torch.equal(wg.updates[-1].batch["advantages"], b.batch["advantages"])            # the actor trains on the same advantages
    step = [l for l in out.splitlines() if l.startswith("step 2:")][0]
    for key in ("critic/vf_loss:0.5", "critic/grad_norm:1.5", "critic/values/mean", "critic/vf_explained_var", "timing_s/values", "timing_s/update_critic"):
        assert key in step, (key, step)
    assert any(p.endswith(os.path.join("global_step_2", "critic")) for p in cg.saved) and any(p.endswith(os.path.join("global_step_2", "actor")) for p in wg.saved)
    # batch-size validation of the critic's config (ray_trainer.py:248-257)
    with pytest.raises(ValueError, match="critic global batch size"):
        make_trainer(tmp_path, extra=extra + ["worker.critic.global_batch_size=3"])
