"""Actor / reference-policy compute on one MI355X: the counterpart of DataParallelPPOActor
(verl/workers/actor/dp_actor.py:45-292) + AnyPrecisionAdamW + the constant-with-warmup schedule
(verl/utils/torch_functional.py:187-329), built on the explicit engine in model.py.

Data-parallel semantics: weights are replicated; each rank accumulates fp32 gradients over
its micro-batches and ONE sum all-reduce over RCCL/xGMI (bucketed slices of the flat buffer) + division by
the world size reproduces FSDP's averaged reduce-scatter; the optimizer then runs identically on every rank.
The slices of the LM layers are reduced while the LAST backward pass of the optimizer step is still running
(GradReducer: a layer's slice is final as soon as that pass has left the layer), the rest right after it.
"""
from __future__ import annotations

from collections import defaultdict
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import ops
from .model import F32, I64, ParamStore, Qwen25VL, VLConfig, pixels_on_device


@dataclass
class ActorHyper:
    """Mirror of the fields of verl/workers/actor/config.py ActorConfig / OptimConfig that the path reads."""
    micro_batch_size_per_device_for_update: int = 4
    micro_batch_size_per_device_for_experience: int = 16
    global_batch_size_per_device: int = 16
    max_grad_norm: float = 1.0
    clip_ratio_low: float = 0.2
    clip_ratio_high: float = 0.3
    clip_ratio_dual: float = 3.0
    ppo_epochs: int = 1
    use_kl_loss: bool = True
    disable_kl: bool = False
    kl_penalty: str = "low_var_kl"
    kl_coef: float = 1e-2
    lr: float = 1e-6
    betas: tuple = (0.9, 0.999)
    eps: float = 1e-8
    weight_decay: float = 1e-2
    lr_warmup_steps: int = 0
    allreduce_bucket_mb: int = 512
    grad_exchange: str = "allreduce"        # "allreduce" | "reduce_scatter" (direct reduce-scatter + all-gather over all xGMI links)
    grad_exchange_dtype: str = "fp32"       # payload of the exchange: "fp32" (FSDP mp_reduce_dtype default) | "bf16"
    optim_strategy: str = "adamw_bf16"      # adamw_bf16 = AnyPrecisionAdamW (bf16 states + Kahan); adamw = torch.optim.AdamW(fused) semantics
    freeze_vision_tower: bool = False       # fsdp_workers.py:226-232: the ViT gets no gradient and no optimizer update
    cliprange_value: float = 0.5            # critic only (verl/workers/critic/config.py:29)


class _StagePool:
    """Staging buffers of the gradient exchange, allocated once and recycled (round 4).  Round 3's reducer allocated a zero-filled send
    buffer + a receive buffer per 512-MB bucket (direct schedule) or a bf16 copy per bucket (bf16 payload) on EVERY optimizer step — tens
    of GB of caching-allocator traffic per step at 7B next to a pool already at ~220 GB reserved.  A buffer goes back to the pool when the
    collective that used it has been waited for; take() hands out a free buffer of the right size / dtype or allocates one (only during
    the first optimizer step: the slices of a step are the same every time)."""

    def __init__(self, device):
        self.device = device
        self.free: Dict[tuple, list] = defaultdict(list)
        self.allocated_bytes = 0
        self.allocations = 0

    def take(self, numel: int, dtype) -> torch.Tensor:
        lst = self.free[(numel, dtype)]
        if lst:
            return lst.pop()
        self.allocations += 1
        self.allocated_bytes += numel * torch.empty((), dtype=dtype).element_size()
        return torch.empty(numel, dtype=dtype, device=self.device)

    def give(self, buf: torch.Tensor):
        self.free[(buf.numel(), buf.dtype)].append(buf)


class GradReducer:
    """Sum-exchange of the flat fp32 gradient buffer in slices, some of them early.

    ready(lo, hi) — the caller guarantees grad[lo:hi] will not be written again before finish(): the slice goes out at once as
    an asynchronous collective (RCCL runs it on its own stream, ordered after the kernels already queued on the current one),
    so it overlaps the rest of the backward pass.  finish() exchanges whatever was NOT announced (in buckets of bucket_elems),
    waits for everything and divides by the world size.  Correctness therefore never depends on ready() being called at all;
    every rank must announce the same slices in the same order (they do: the order is the backward order of the layers).

    mode (ActorHyper.grad_exchange):
      "allreduce"       one ncclAllReduce(sum) per bucket — RCCL's ring / tree schedule.
      "reduce_scatter"  SURVEY.md §5.8's topology-native schedule for a fully connected xGMI node: DIRECT reduce-scatter +
                        all-gather.  A bucket is cut into `world` shards; an all-to-all hands shard j of every rank to rank j over all
                        peer links at once (each of the 7 xGMI links carries 1/8 of the bucket, no ring hops), rank j adds the
                        `world` contributions in rank order (fixed order: every rank ends up with bit-identical sums), and an
                        all-gather returns the reduced shards.  Each byte crosses a link once per phase: 2 x (W-1)/W of the
                        bucket in total per GPU, spread over W-1 links in parallel.
    payload "bf16" (ActorHyper.grad_exchange_dtype; the reference's FSDP knob is `mp_reduce_dtype`, fp32 by default) sends bf16-rounded
    contributions (half the bytes).  Where the sum is formed depends on the mode: "reduce_scatter" upcasts the received shards and adds
    them in fp32 (then rounds the sum to bf16 for the all-gather); "allreduce" hands the bf16 buffer to RCCL, which accumulates IN bf16 —
    the same arithmetic as FSDP's reduce_dtype=bf16.  The AdamW kernels round the gradient to bf16 anyway (the dtype the reference's
    optimizer sees), so only the clip-norm and the rounding before / inside the sum differ.

    Timing (round 4, bench.py `allreduce_s` / `allreduce_exposed_s`): device events on the compute stream bracket (a) the whole exchange —
    first slice sent .. gradients final — and (b) the part of it the compute stream spends inside finish() (what the overlap did NOT
    hide); stats() reads the events of the steps finished so far."""

    def __init__(self, grad: torch.Tensor, world: int, group=None, bucket_elems: int = 1 << 27, mode: str = "allreduce", payload: str = "fp32",
                 mean_over: Optional[int] = None):
        assert mode in ("allreduce", "reduce_scatter") and payload in ("fp32", "bf16"), (mode, payload)
        self.grad, self.world, self.pg, self.bucket = grad, world, group, max(1, int(bucket_elems))
        # the exchange SUMS over all `world` ranks and divides by `mean_over`: the data-parallel replicas (= world, FSDP's averaged
        # reduce-scatter) — or world / sp under Ulysses sequence parallelism, where the sp ranks of a group hold partial sums over their
        # token slices of the SAME rows (what the reference reaches with Gather's grad_scaler, verl/utils/ulysses.py:227-235)
        self.mean_over = int(mean_over) if mean_over else world
        self.mode, self.payload = mode, payload
        self.sent: List[tuple] = []
        self.works: list = []
        self.pending: list = []           # reduce_scatter mode: (lo, hi, send, recv, work) of all-to-alls in flight
        self.early_elems = 0
        # the post-processing of the direct schedule (shard sums, all-gathers) runs on a side stream so that a collective still in
        # flight never stalls the backward kernels queued on the compute stream
        self.side = torch.cuda.Stream(device=grad.device) if grad.is_cuda else None
        self.pool = _StagePool(grad.device)
        self._t_first = None              # event (CUDA) / perf_counter (CPU) of the first slice sent in the current exchange
        self._timers: list = []           # (first, finish_begin, finish_end) per finished exchange, not yet read
        self.totals = {"exchanges": 0, "allreduce_s": 0.0, "allreduce_exposed_s": 0.0, "early_fraction": 0.0}

    def _mark(self):
        if self.grad.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            return ev
        import time
        return time.perf_counter()

    def stats(self) -> Dict[str, float]:
        """Totals over the exchanges finished so far (synchronises on their events)."""
        for first, b, e, early in self._timers:
            if self.grad.is_cuda:
                e.synchronize()
                tot, exp = first.elapsed_time(e) * 1e-3, b.elapsed_time(e) * 1e-3
            else:
                tot, exp = e - first, e - b
            self.totals["exchanges"] += 1
            self.totals["allreduce_s"] += tot
            self.totals["allreduce_exposed_s"] += exp
            self.totals["early_fraction"] += early
        self._timers = []
        return dict(self.totals)

    # ---- plain all-reduce
    def _send_allreduce(self, lo: int, hi: int):
        for o in range(lo, hi, self.bucket):
            sl = self.grad[o:min(hi, o + self.bucket)]
            if self.payload == "bf16":
                buf = self.pool.take(sl.numel(), torch.bfloat16)
                buf.copy_(sl)                                            # fp32 -> bf16 into the recycled staging buffer
                self.works.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), sl, buf, (buf,)))
            else:
                self.works.append((dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), None, None, ()))

    # ---- direct reduce-scatter + all-gather
    def _ctx(self):
        import contextlib
        return torch.cuda.stream(self.side) if self.side is not None else contextlib.nullcontext()

    def _send_direct(self, lo: int, hi: int):
        W = self.world
        dt = torch.bfloat16 if self.payload == "bf16" else torch.float32
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream())        # the slice's last gradient kernels are queued on the compute stream
        with self._ctx():
            for o in range(lo, hi, self.bucket):
                e = min(hi, o + self.bucket)
                n = e - o
                per = -(-n // W)
                send = self.pool.take(W * per, dt)
                send[:n].copy_(self.grad[o:e])
                if W * per > n:
                    send[n:].zero_()                                    # the ragged tail of the last shard (recycled buffer: stale data)
                recv = self.pool.take(W * per, dt)
                work = dist.all_to_all_single(recv, send, group=self.pg, async_op=True)      # recv[j*per:(j+1)*per] = rank j's shard for me
                self.pending.append((o, e, per, send, recv, work))

    def _drain_direct(self):
        """Shard sums + all-gathers of every all-to-all issued so far (on the side stream)."""
        W = self.world
        with self._ctx():
            for (o, e, per, send, recv, work) in self.pending:
                work.wait()
                acc = self.pool.take(per, torch.float32)
                acc.copy_(recv.view(W, per)[0])
                for j in range(1, W):                                   # rank order: the same fp32 sum on every rank
                    acc.add_(recv.view(W, per)[j])
                staged = [send, recv, acc]
                mine = acc
                if send.dtype != torch.float32:
                    mine = self.pool.take(per, send.dtype)
                    mine.copy_(acc)
                    staged.append(mine)
                full = send                                             # reuse as the gather target
                self.works.append((dist.all_gather_into_tensor(full, mine, group=self.pg, async_op=True), self.grad[o:e], full, tuple(staged)))
        self.pending = []

    def _send(self, lo: int, hi: int):
        if self.mode == "reduce_scatter":
            self._send_direct(lo, hi)
        else:
            self._send_allreduce(lo, hi)

    def ready(self, lo: int, hi: int):
        lo, hi = max(0, int(lo)), min(int(hi), self.grad.numel())
        if hi <= lo:
            return
        for a, b in self.sent:
            assert hi <= a or lo >= b, f"gradient slice [{lo},{hi}) announced twice (overlaps [{a},{b}))"
        self.sent.append((lo, hi))
        self.early_elems += hi - lo
        if self._t_first is None:
            self._t_first = self._mark()
        if self.mode == "reduce_scatter" and self.pending:
            self._drain_direct()                                        # earlier slices: sum + all-gather while backward continues
        self._send(lo, hi)

    def finish(self):
        t_begin = self._mark()
        if self._t_first is None:
            self._t_first = t_begin
        early = self.early_elems / max(1, self.grad.numel())
        pos = 0
        for a, b in sorted(self.sent):
            if a > pos:
                self._send(pos, a)
            pos = b
        if pos < self.grad.numel():
            self._send(pos, self.grad.numel())
        if self.pending:
            self._drain_direct()
        with self._ctx():
            for w, dst, buf, staged in self.works:
                w.wait()
                if dst is not None:
                    dst.copy_(buf[:dst.numel()])                        # bf16 payload / gathered shards back into the fp32 buffer
                for b_ in staged:                                       # the next user's copy is ordered behind this one on the same stream
                    self.pool.give(b_)
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        self.sent, self.works, self.early_elems = [], [], 0
        self.grad.mul_(1.0 / self.mean_over)                      # FSDP reduce-scatter averages over the data-parallel ranks
        self._timers.append((self._t_first, t_begin, self._mark(), early))
        self._t_first = None


def release_cached_blocks():
    """End of a phase (rollout / a log-prob pass / update): hand the allocator's cached blocks back to the driver.  The phases
    allocate differently shaped tensors (KV caches, 150k-wide logits, per-layer activations of passes whose packed length changes
    every step); carried over, the split blocks of one phase fragment the next one and the reserved pool creeps up step after step
    (round 2: 202 -> 235 GB over 25 steps).  OPT-IN (ST_EMPTY_CACHE=1): measured on the bench it costs more than it saves — the
    update phase then re-requests ~70 GB from the driver every step, peak reserved went 202 -> 287 GB and update_actor 14.9 -> 17.1 s
    (profiles/r03_notes.md; torch's expandable-segments mode, the usual cure, is "not supported on this platform")."""
    if not torch.cuda.is_available():
        return
    if os.environ.get("ST_EMPTY_CACHE", "0") == "1":
        torch.cuda.empty_cache()
        return
    # default: only when the pool has crept close to the device's capacity (the caching allocator would otherwise start to
    # free-and-retry inside the next phase, which synchronises the device per block) — a rare, bounded clean-up
    free_b, total_b = torch.cuda.mem_get_info()
    if torch.cuda.memory_reserved() > 0.80 * total_b and torch.cuda.memory_allocated() < 0.60 * total_b:
        torch.cuda.empty_cache()


def _rows(x, sl):
    return x[sl] if x is not None else None


def _rollout_rows(prompt_ids, prompt_mask, prompt_pos, responses, eos_token_id) -> Dict[str, torch.Tensor]:
    """The rows `assemble_rollout_batch` (verl/workers/rollout/hip_rollout.py; reference vllm_rollout_spmd.py:144-188) builds for these
    samples: response mask = 1 through the first EOS (torch_functional.py:23-33), response position ids last + 1 .. last + R on every
    M-RoPE row.  Restated here so that the engine does not import the API layer above it; tests compare the two."""
    R = responses.shape[1]
    eos = torch.as_tensor([eos_token_id] if isinstance(eos_token_id, int) else list(eos_token_id), dtype=responses.dtype)
    hit = torch.isin(responses, eos).long()
    rmask = (hit.cumsum(1) - hit).eq(0).to(prompt_mask.dtype)
    delta = torch.arange(1, R + 1).view(1, 1, -1)
    return {"input_ids": torch.cat([prompt_ids, responses], -1), "attention_mask": torch.cat([prompt_mask, rmask], -1), "response_mask": rmask,
            "position_ids": torch.cat([prompt_pos, prompt_pos[..., -1:] + delta], -1)}


class EarlyLogProb:
    """The old-policy log-prob pass of a GRPO step, started before the rollout has ended (round 5).

    The reference runs rollout and `compute_log_probs` one after the other (verl/trainer/ray_trainer.py:585-640).  Here the last ~400 of a
    rollout's ~900 decode iterations run <= 128 live rows — weight streaming with the matrix pipes idle — while the log-prob pass of the
    samples that have already finished is MFMA-bound and needs nothing but their tokens and the prompt K/V of the prefill.  The generator
    reports the finishers of every decode phase (Generator.generate(on_finished=self.feed)); `feed` stages their rows exactly as
    `assemble_rollout_batch` will (same response mask, same position ids) and issues Qwen25VL.log_probs_cached for them on `side_stream`, a
    CU-range stream complementary to the decode tail's (ops.cu_range_stream).  EXPERIMENTAL — measured slower end to end on MI355X
    (gen + old 12.35-12.40 s vs 11.99 s serial, profiles/r05_notes.md §2): the streams keep to their compute units, but the passes contend
    for every XCD's L2 and the fabric, so the mode is off by default.  `finish` computes the samples of the last phase, waits for the side stream and returns the
    (N, R) tensor compute_log_prob would have returned.

    Same arithmetic, different grouping: a pass holds the finishers of ONE phase (cut by the token budget), not consecutive rows.  The
    log-prob of a row does not depend on the rows it is packed with up to the summation order of the GEMM tail split, so
    `PolicyEngine.compute_log_prob_in_sets` — the same sets, processed serially after the rollout — is the bit-identical serial form
    (tests/test_gpu_rollout.py::test_early_old_log_probs_are_bit_identical_to_the_serial_order)."""

    def __init__(self, engine: "PolicyEngine", input_ids, attention_mask, position_ids, n: int, response_length: int, temperature: float,
                 eos_token_id, side_stream=None):
        self.e, self.n, self.R, self.t, self.eos, self.side = engine, int(n), int(response_length), float(temperature), eos_token_id, side_stream
        self.ids = torch.as_tensor(_to_np(input_ids))
        self.mask = torch.as_tensor(_to_np(attention_mask))
        pos = torch.as_tensor(_to_np(position_ids))
        self.pos = pos if pos.dim() == 3 else pos[:, None, :].repeat(1, 3, 1)
        self.done: List[tuple] = []          # (sample ids, responses used (cpu), log-probs (device), stream it was issued on)
        self.sets: List[np.ndarray] = []     # the sets in processing order (the serial re-run of the bit-identity test)
        self.tokens_per_pass = int(os.environ.get("ST_TOKENS_EARLY", "32768"))
        self.rows_per_pass = 256
        self.fed_s = 0.0                     # host seconds spent inside feed (staging + launches)
        self._keep: list = []

    # -- one set of samples: rows as assemble_rollout_batch builds them, passes by the token budget, log_probs_cached per pass
    def _process(self, sample_ids: np.ndarray, resp_cpu: torch.Tensor, cache: dict, compute_stream=None, copy_stream=None):
        """compute_stream / copy_stream (the hook's case): every host -> device copy of the staging runs on `copy_stream`, where it waits
        for nothing but the copies before it — on the compute stream a copy from pageable memory would hold the HOST until the passes
        queued there have finished (first version: 2.3 s of host time per rollout inside the hook, the decode loop starved meanwhile);
        the passes themselves are then queued on `compute_stream` in one go."""
        pr = torch.as_tensor(sample_ids // self.n)
        rows = _rollout_rows(self.ids[pr], self.mask[pr], self.pos[pr], resp_cpu, self.eos)
        r_len = rows["response_mask"].sum(1).numpy().astype(np.int64)
        cuts, a = [], 0
        while a < len(sample_ids):
            b, tok = a + 1, int(r_len[a])
            while b < len(sample_ids) and b - a < self.rows_per_pass and tok + int(r_len[b]) <= self.tokens_per_pass:
                tok += int(r_len[b]); b += 1
            cuts.append(slice(a, b))
            a = b
        cur = torch.cuda.current_stream()
        cs, ks = copy_stream or cur, compute_stream or cur
        with torch.cuda.stream(cs):
            dbs = [self.e.model.stage_responses(rows["input_ids"][sl], rows["attention_mask"][sl], rows["position_ids"][sl], self.R,
                                                (sample_ids[sl] // self.n), cache["p_off"]) for sl in cuts]
        if ks is not cs:
            ks.wait_stream(cs)
        with torch.cuda.stream(ks):
            lp = torch.cat([self.e.model.log_probs_cached(db, cache, self.t) for db in dbs], 0)
        self._keep.append(dbs)               # staged on another stream than the one that reads them: alive until finish()
        return lp

    def feed(self, sample_ids: np.ndarray, out: torch.Tensor, event, cache: dict):
        """Generator hook: rows `sample_ids` of `out` are final once `event` has passed."""
        t0 = time.perf_counter()
        sample_ids = np.asarray(sample_ids, dtype=np.int64)
        st = self.side if self.side is not None else torch.cuda.current_stream()
        cs = self._copy_stream() if self.side is not None else st
        cs.wait_event(event)
        with torch.cuda.stream(cs):
            resp = out[torch.as_tensor(sample_ids, device=out.device)].cpu()       # holds the host until `event` only (passed long ago)
        with ops.scratch_slot(1 if self.side is not None else 0):
            lp = self._process(sample_ids, resp, cache, compute_stream=st, copy_stream=cs)
        self.done.append((sample_ids, resp, lp, st))
        self.sets.append(sample_ids)
        self.fed_s += time.perf_counter() - t0

    _copy = {}

    def _copy_stream(self):
        dev = torch.cuda.current_device()
        if dev not in EarlyLogProb._copy:
            EarlyLogProb._copy[dev] = torch.cuda.Stream()
        return EarlyLogProb._copy[dev]

    @torch.no_grad()
    def finish(self, data: Dict[str, Any], cache: Optional[dict]) -> torch.Tensor:
        """(N, R) old-policy log-probs for `data` (the assembled rollout batch).  Rows the hook never saw (the last phase's finishers;
        everything when the generator ran without hooks) are computed here; fed rows are checked against data["responses"] and
        recomputed through the ordinary pass on any mismatch (then nothing of the early work is used)."""
        e = self.e
        N = data["input_ids"].shape[0]
        resp_all = torch.as_tensor(data["responses"]).cpu()
        ok = cache is not None and "kp" in cache and e._cache_matches(data, cache, self.R)
        for ids_, resp, _, _ in self.done:
            ok = ok and bool(torch.equal(resp_all[torch.as_tensor(ids_)], resp))
        if not ok:
            self.done, self.sets = [], []
            return e.compute_log_prob(data, self.t, prompt_cache=cache)
        seen = np.zeros(N, dtype=bool)
        for ids_, *_ in self.done:
            seen[ids_] = True
        rest = np.nonzero(~seen)[0].astype(np.int64)
        cur = torch.cuda.current_stream()
        if len(rest):
            lp = self._process(rest, resp_all[torch.as_tensor(rest)], cache)
            self.done.append((rest, None, lp, cur))
            self.sets.append(rest)
        out = torch.zeros(N, self.R, dtype=F32, device=e.store.device)
        for ids_, _, lp, st in self.done:
            if st is not cur:
                cur.wait_stream(st)
            out.index_copy_(0, torch.as_tensor(ids_, device=out.device), lp)
        e.last_prompt_cache_hit, e.last_log_prob_source = True, "forward (finished samples during the decode tail)"
        e.last_plan["experience"] = [(int(len(s_)),) for s_ in self.sets]
        for st in {id(d[3]): d[3] for d in self.done}.values():
            st.synchronize()                 # the staged batches were allocated on the copy stream and read on the side stream: nothing of
        self.done, self._keep = [], []       # them may return to the allocator while a pass still runs
        release_cached_blocks()
        return out


def _to_np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


class PolicyEngine:
    """Holds one model replica (actor with optimizer, or frozen reference when hyper is None)."""

    def __init__(self, cfg: VLConfig, store: ParamStore, hyper: Optional[ActorHyper] = None, process_group=None, sp_group=None):
        self.cfg, self.store, self.h = cfg, store, hyper
        self.model = Qwen25VL(cfg, store)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        # Ulysses sequence parallelism (worker.actor.ulysses_sequence_parallel_size): the ranks of sp_group run the SAME rows, each on
        # its slice of every packed pass (Qwen25VL.set_sequence_parallel); gradients add up over sp_group and average over world / sp
        self.sp_group = sp_group
        self.sp = dist.get_world_size(sp_group) if sp_group is not None else 1
        self.model.set_sequence_parallel(sp_group if self.sp > 1 else None)
        self.sync_grads = self.world > 1                                       # tests force it on a 1-rank group
        self.overlap_allreduce = os.environ.get("ST_OVERLAP_ALLREDUCE", "1") != "0"
        self._reducer: Optional[GradReducer] = None
        self.share_prompts = True     # pack the prompt of a rollout group once (see _stage)
        # Several reference micro-batches ride in one forward(/backward) pass, as many as a PACKED-TOKEN budget admits (_plan_passes):
        # the reference's contract is that micro_batch_size_per_device_* holds for any sequence length up to max_prompt_length +
        # max_response_length (dp_actor.py:169-292, scripts/spatialthinker_7b_grpo.sh:33-34), so the fusion factor is derived from the
        # rows at hand, never a constant.  With gradients a packed token keeps ~3.0 MB of activations at 7B (28 layers x (x0, qkv, attn
        # out, x1, gate|up) bf16, the RMSNorm / SwiGLU outputs recomputed in the backward): 24k tokens = ~72 GB next to the 131 GB of
        # weights, gradients, optimizer state and the frozen reference.  The bench's 8 x 4 rows (~21k tokens) fit one pass; 32 rows at
        # the 2048-token response cap (~70k tokens) are cut into passes of 8 rows.  A single reference micro-batch is never split.
        self.fuse_micro_batches = 8                                                      # upper bound on micro-batches per update pass
        self.tokens_per_pass_grad = int(os.environ.get("ST_TOKENS_GRAD", "24576"))
        if hyper is not None:
            self.model.recompute_light = True
        # no-grad log-prob passes: rows are independent there (no loss normalisation), the result does not depend on the grouping,
        # the GEMMs see more rows; activations are transient (~0.2 MB per packed token + 0.3 MB per response row of logits at 7B)
        self.fuse_experience = int(os.environ.get("ST_FUSE_EXPERIENCE", "16"))         # upper bound on micro-batches per no-grad pass
        self.tokens_per_pass_nograd = int(os.environ.get("ST_TOKENS_NOGRAD", "65536"))
        self.last_plan: Dict[str, list] = {}          # row ranges of the passes of the latest compute_log_prob / update_policy call
        self.last_prompt_cache_hit = False   # did the latest compute_log_prob run on the rollout's prompt K/V? (perf/prompt_cache_hit)
        self.opt_steps = 0            # t of AdamW (state["step"])
        self.sched_steps = 0          # lr_scheduler.step() calls so far: once per update_policy call (fsdp_workers.py:453)
        self._norm_buf = torch.zeros(1, dtype=F32, device=store.device) if hyper is not None else None
        self._coef = torch.ones(1, dtype=F32, device=store.device) if hyper is not None else None

    # ------------------------------------------------------------------ schedule
    def current_lr(self) -> float:
        """get_constant_schedule_with_warmup (torch_functional.py:187-197): lr * min(1, s / max(1, warmup)); s = 0 on
        the first update call, so that call trains at lr = 0 — reproduced on purpose."""
        return self.h.lr * min(1.0, float(self.sched_steps) / float(max(1, self.h.lr_warmup_steps)))

    # ------------------------------------------------------------------ micro-batch plumbing
    def _stage(self, data: Dict[str, Any], sl: slice):
        """Stage one micro-batch.  Rows that are rollouts of the same prompt (identical prompt columns AND the same image
        object — DataProto.repeat / np.repeat keep references) form a group: the prompt (and its image) is packed and computed
        once per group (SURVEY 8a11's per-sequence packing computes it G times); single-member groups are plain sequences."""
        ids, am = data["input_ids"][sl], data["attention_mask"][sl]
        R = data["responses"].shape[1]
        ids_np, am_np = _to_np(ids), _to_np(am)
        n, Pc = ids_np.shape[0], ids_np.shape[1] - R
        mm = data.get("multi_modal_inputs")
        items = list(mm[sl]) if mm is not None else [None] * n
        img_key = [id(it["pixel_values"]) if (it is not None and "pixel_values" in it) else None for it in items]
        groups, keys = [], {}
        for r in range(n):
            key = (img_key[r], ids_np[r, :Pc].tobytes(), am_np[r, :Pc].tobytes()) if self.share_prompts else r
            groups.append(keys.setdefault(key, len(keys)))
        px = gr = None
        multi = any(it is not None and "image_grid_thw" in it and len(np.asarray(it["image_grid_thw"]).reshape(-1, 3)) != 1 for it in items)
        first = {}
        for r, gid in enumerate(groups):
            first.setdefault(gid, r)
        lead = sorted(first.values())                           # pack_batch orders the groups by first appearance
        owners = [r for r in (lead if not multi else range(n)) if items[r] is not None and "pixel_values" in items[r]]
        if multi:                                               # several images per sample: no sharing, the per-sample path
            groups = list(range(n))
        if owners:                                              # dp_actor.py:78-83 concatenates over the samples
            px = torch.cat([pixels_on_device(items[r]["pixel_values"], self.store.device) for r in owners], 0)      # one H2D copy per image and step
            gr = np.concatenate([np.asarray(items[r]["image_grid_thw"]).reshape(-1, 3) for r in owners], 0)
        return self.model.stage(ids, am, data["position_ids"][sl], R, px, gr, groups=groups)

    # ------------------------------------------------------------------ token-budgeted passes
    def _row_stats(self, data: Dict[str, Any], R: int):
        """Per row: valid prompt tokens, valid response tokens and an identity of (prompt, image) — what one packed pass holds is
        every response + each DISTINCT prompt once (shared-prompt packing, _stage)."""
        ids, am = _to_np(data["input_ids"]), _to_np(data["attention_mask"])
        Pc = ids.shape[1] - R
        p_len, r_len = am[:, :Pc].sum(1).astype(np.int64), am[:, Pc:].sum(1).astype(np.int64)
        mm = data.get("multi_modal_inputs")
        keys = []
        for r in range(ids.shape[0]):
            it = mm[r] if mm is not None else None
            img = id(it["pixel_values"]) if (it is not None and "pixel_values" in it) else None
            keys.append(hash((img, ids[r, :Pc].tobytes(), am[r, :Pc].tobytes())) if self.share_prompts else r)
        return p_len, r_len, keys

    @staticmethod
    def _plan_passes(lo: int, hi: int, unit: int, max_units: int, budget: int, p_len, r_len, keys, prompts_cached: bool = False) -> List[tuple]:
        """Cut rows [lo, hi) into passes of whole `unit`-row blocks (the reference's micro-batches): a pass grows block by block while
        it stays within `max_units` blocks and `budget` packed tokens.  One block always forms a pass, whatever its size."""
        def tokens(a, b):
            t = int(r_len[a:b].sum())
            if not prompts_cached:
                seen = {}
                for r in range(a, b):
                    seen.setdefault(keys[r], int(p_len[r]))
                t += sum(seen.values())
            return t
        out, s0 = [], lo
        while s0 < hi:
            e = min(hi, s0 + unit)
            while e < hi and (e - s0) // unit < max_units and tokens(s0, min(hi, e + unit)) <= budget:
                e = min(hi, e + unit)
            out.append((s0, e))
            s0 = e
        return out

    def rollout_log_probs(self, data: Dict[str, Any], temperature: float, cache: Optional[dict]) -> Optional[torch.Tensor]:
        """The log-probs the rollout recorded while sampling (Generator.generate(emit_log_probs=True)), IF they are the old-policy
        log-probs of exactly these rows: same weights (version counter), same temperature, the same response tokens row for row.
        Masked to the response mask like every compute_log_prob result; None when anything differs (the caller recomputes)."""
        if not cache or cache.get("log_probs") is None:
            return None
        if cache.get("weights_version", 0) != getattr(self.store, "version", 0) or abs(float(cache.get("temperature", -1.0)) - float(temperature)) > 0:
            return None
        resp = torch.as_tensor(data["responses"])
        lp, gen = cache["log_probs"], cache["responses"]
        if tuple(resp.shape) != tuple(gen.shape) or not torch.equal(resp.to(gen.device), gen):
            return None
        R = resp.shape[1]
        return lp * torch.as_tensor(data["attention_mask"])[:, -R:].to(lp.device, lp.dtype)

    @torch.no_grad()
    def compute_log_prob(self, data: Dict[str, Any], temperature: float, micro_batch_size: Optional[int] = None,
                         prompt_cache: Optional[dict] = None, use_rollout_log_probs: bool = False) -> torch.Tensor:
        """dp_actor.py:169-210: (N, R) fp32 log-probs of the responses, micro-batched.
        use_rollout_log_probs (opt-in, round 4): when the rollout recorded the log-probs of the tokens it sampled and `data` is exactly
        that rollout under the current weights, they ARE the old-policy log-probs — returned without a second forward
        (self.last_log_prob_source = "rollout"); any mismatch falls back to the pass below.
        prompt_cache (from Generator.generate(return_prompt_cache=True) with THESE weights, rows = its prompts x n, prompt-major):
        the pass then runs on the response tokens only, on top of the cached prompt K/V — the old-policy log-probs of a GRPO step
        need no second pass over the prompts and images."""
        self.last_log_prob_source = "forward"
        if use_rollout_log_probs:
            got = self.rollout_log_probs(data, temperature, prompt_cache)
            if got is not None:
                self.last_log_prob_source, self.last_prompt_cache_hit = "rollout", False
                self.last_plan["experience"] = []
                return got
        dbg = os.environ.get("ST_STAGE_DEBUG", "0") == "1"
        tl = [time.perf_counter()]
        N = data["input_ids"].shape[0]
        mb = micro_batch_size or (self.h.micro_batch_size_per_device_for_experience if self.h else 16)
        R = data["responses"].shape[1]
        self.last_prompt_cache_hit = bool(prompt_cache is not None and "kp" in prompt_cache and getattr(self, "sp", 1) == 1
                                          and self._cache_matches(data, prompt_cache, R))   # (sequence-parallel passes take whole rows)
        p_len, r_len, keys = self._row_stats(data, R)
        passes = self._plan_passes(0, N, mb, max(1, int(self.fuse_experience)), self.tokens_per_pass_nograd, p_len, r_len, keys,
                                   prompts_cached=self.last_prompt_cache_hit)
        self.last_plan["experience"] = passes
        tl.append(time.perf_counter())
        outs = []
        if self.last_prompt_cache_hit:
            n = prompt_cache["n"]
            for (a, b_) in passes:
                sl = slice(a, b_)
                rows = np.arange(a, b_)
                b = self.model.stage_responses(data["input_ids"][sl], data["attention_mask"][sl], data["position_ids"][sl], R,
                                               rows // n, prompt_cache["p_off"])
                tl.append(time.perf_counter())
                outs.append(self.model.log_probs_cached(b, prompt_cache, temperature))
                tl.append(time.perf_counter())
        else:
            for (a, b_) in passes:
                b = self._stage(data, slice(a, b_))
                tl.append(time.perf_counter())
                outs.append(self.model.log_probs(b, temperature))
                tl.append(time.perf_counter())
        if dbg:
            print(f"[stage debug] compute_log_prob host ms: plan {1e3 * (tl[1] - tl[0]):.1f}; (stage, launch) per pass: "
                  + ", ".join(f"({1e3 * (tl[i + 1] - tl[i]):.0f}, {1e3 * (tl[i + 2] - tl[i + 1]):.0f})" for i in range(1, len(tl) - 2, 2)), flush=True)
        out = torch.cat(outs, 0)
        del outs, b
        release_cached_blocks()
        return out

    def early_log_prob(self, input_ids, attention_mask, position_ids, n: int, response_length: int, temperature: float, eos_token_id,
                       side_stream=None) -> "EarlyLogProb":
        """Old-policy log-probs of the rollouts that have FINISHED, computed while the decode tail of the same rollout still runs
        (EarlyLogProb): hand its .feed to Generator.generate(on_finished=...), call .finish(data, prompt_cache) where compute_log_prob
        would be called."""
        return EarlyLogProb(self, input_ids, attention_mask, position_ids, n, response_length, temperature, eos_token_id, side_stream)

    @torch.no_grad()
    def compute_log_prob_in_sets(self, data: Dict[str, Any], temperature: float, prompt_cache: dict, sets, eos_token_id, n: int) -> torch.Tensor:
        """The serial form of EarlyLogProb: the same sets of rows, one after the other on the current stream, after the rollout."""
        R = data["responses"].shape[1]
        P = data["input_ids"].shape[1] - R
        early = EarlyLogProb(self, data["input_ids"][::n, :P], data["attention_mask"][::n, :P], data["position_ids"][::n, ..., :P], n, R,
                             temperature, eos_token_id, None)
        resp_all = torch.as_tensor(data["responses"]).cpu()
        out = torch.zeros(data["input_ids"].shape[0], R, dtype=F32, device=self.store.device)
        for s_ in sets:
            s_ = np.asarray(s_, dtype=np.int64)
            if len(s_):
                out.index_copy_(0, torch.as_tensor(s_, device=out.device), early._process(s_, resp_all[torch.as_tensor(s_)], prompt_cache))
        return out

    def _cache_matches(self, data: Dict[str, Any], cache: dict, R: int) -> bool:
        """The cache is usable only for exactly the prompts it was built from (row r <-> prompt r // n) and the current weights."""
        ids, am = _to_np(data["input_ids"]), _to_np(data["attention_mask"])
        n, Pc = cache["n"], ids.shape[1] - R
        if cache.get("weights_version", 0) != getattr(self.store, "version", 0):
            return False
        if ids.shape[0] != cache["prompt_ids"].shape[0] * n or Pc != cache["prompt_ids"].shape[1]:
            return False
        return bool(np.array_equal(ids[::n, :Pc], cache["prompt_ids"]) and np.array_equal(am[::n, :Pc], cache["prompt_mask"])
                    and np.array_equal(ids[:, :Pc], np.repeat(cache["prompt_ids"], n, 0)))

    # ------------------------------------------------------------------ optimizer
    def zero_grad(self):
        self.store.grad.zero_()

    def grad_reducer(self) -> Optional[GradReducer]:
        if not getattr(self, "sync_grads", self.world > 1):
            return None
        if getattr(self, "_reducer", None) is None or self._reducer.grad is not self.store.grad:
            self._reducer = GradReducer(self.store.grad, self.world, self.pg, self.h.allreduce_bucket_mb * (1 << 20) // 4,
                                        mode=os.environ.get("ST_GRAD_EXCHANGE", getattr(self.h, "grad_exchange", "allreduce")),
                                        payload=os.environ.get("ST_GRAD_EXCHANGE_DTYPE", getattr(self.h, "grad_exchange_dtype", "fp32")),
                                        mean_over=max(1, self.world // max(1, getattr(self, "sp", 1))))
        return self._reducer

    def all_reduce_grads(self):
        """Everything update_policy's last backward pass has not already sent (all of it when overlap is off)."""
        red = self.grad_reducer()
        if red is not None:
            red.finish()

    def optimizer_step(self) -> float:
        """dp_actor.py:155-167: global grad-norm clip (max_grad_norm), skip on a non-finite norm, AdamW-Kahan step."""
        h, st = self.h, self.store
        self.all_reduce_grads()
        ops.sumsq(st.grad, out=self._norm_buf)
        norm = float(self._norm_buf.sqrt().item())                # one host sync per optimizer step
        if not np.isfinite(norm):
            print("Gradient norm is not finite. Skip update.")
            self.zero_grad()
            return norm
        self._coef.fill_(min(1.0, h.max_grad_norm / (norm + 1e-6)))   # clip_grad_norm_: coef = max_norm/(norm+1e-6), clamped to 1
        self.opt_steps += 1
        # frozen parameters have grad None in the reference, so its optimizers skip them entirely (no decay either): the ViT
        # slice leads the flat buffer (model.param_layout), the update starts behind it
        lo = st.offsets["embed"] if h.freeze_vision_tower else 0
        if st.master is not None:                               # fp32 master weights + fp32 moments: torch.optim.AdamW(fused) on the master
            if h.optim_strategy != "adamw":
                raise NotImplementedError("fp32 master weights are built for optim.strategy=adamw (the reference's default pair)")
            ops.adamw_master_step_(st.master[lo:], st.flat[lo:], st.grad[lo:], st.m[lo:], st.v[lo:], t=self.opt_steps, lr=self.current_lr(),
                                   betas=h.betas, eps=h.eps, weight_decay=h.weight_decay, grad_scale=self._coef)
        elif h.optim_strategy == "adamw_bf16":
            ops.adamw_kahan_step_(st.flat[lo:], st.grad[lo:], st.m[lo:], st.v[lo:], st.c[lo:], t=self.opt_steps, lr=self.current_lr(),
                                  betas=h.betas, eps=h.eps, weight_decay=h.weight_decay, grad_scale=self._coef)
        elif h.optim_strategy == "adamw":
            ops.adamw_step_(st.flat[lo:], st.grad[lo:], st.m[lo:], st.v[lo:], t=self.opt_steps, lr=self.current_lr(), betas=h.betas,
                            eps=h.eps, weight_decay=h.weight_decay, grad_scale=self._coef)
        else:
            raise NotImplementedError(f"Optimizer {h.optim_strategy} not supported.")
        st.refresh_transposes()
        st.version = getattr(st, "version", 0) + 1          # invalidates prompt caches built from the previous weights
        self.zero_grad()
        return norm

    # ------------------------------------------------------------------ update
    def update_policy(self, data: Dict[str, Any], temperature: float) -> Dict[str, List[float]]:
        """dp_actor.py:212-292.  data: input_ids, attention_mask, position_ids, responses, old_log_probs, advantages,
        [ref_log_probs], [multi_modal_inputs]; rows = this rank's share of the rollout batch."""
        h = self.h
        N = data["input_ids"].shape[0]
        R = data["responses"].shape[1]
        dev = self.store.device
        use_ref = h.use_kl_loss and not h.disable_kl and data.get("ref_log_probs") is not None
        metrics: Dict[str, List[float]] = defaultdict(list)
        pending = []
        mini, micro = h.global_batch_size_per_device, h.micro_batch_size_per_device_for_update
        assert N % mini == 0 and mini % micro == 0, (N, mini, micro)
        accum = mini // micro
        # several reference micro-batches per forward/backward pass (each keeps its own loss normalisation, see
        # Qwen25VL.forward_backward): with micro = 4 and G = 8 a pass then holds a whole rollout group behind ONE prompt copy.  How
        # many is decided per pass from the packed-token count of the rows at hand (_plan_passes)
        p_len, r_len, keys = self._row_stats(data, R)
        plan = self.last_plan["update"] = []
        for _ in range(h.ppo_epochs):
            for m0 in range(0, N, mini):
                passes = self._plan_passes(m0, m0 + mini, micro, max(1, int(self.fuse_micro_batches)), self.tokens_per_pass_grad, p_len, r_len, keys)
                plan.extend(passes)
                for (s, e) in passes:
                    sl = slice(s, e)
                    # the last pass of the optimizer step: gradient slices go out to the other ranks as backward leaves them
                    red = self.grad_reducer() if (self.overlap_allreduce and e >= m0 + mini) else None
                    b = self._stage(data, sl)
                    to = lambda k, dt=F32: ops.h2d(torch.as_tensor(data[k][sl]), dt, dev)
                    loss_in = dict(old_log_probs=to("old_log_probs"), advantages=to("advantages"),
                                   ref_log_probs=to("ref_log_probs") if use_ref else None,
                                   response_mask=ops.h2d(torch.as_tensor(data["attention_mask"][sl])[:, -R:], I64, dev))
                    _, met = self.model.forward_backward(b, loss_in, temperature, clip_low=h.clip_ratio_low, clip_high=h.clip_ratio_high,
                                                         clip_dual=h.clip_ratio_dual, kl_kind=h.kl_penalty, kl_coef=h.kl_coef,
                                                         grad_accum=float(accum), loss_rows=micro,
                                                         on_final=red.ready if red is not None else None,
                                                         train_vision=not h.freeze_vision_tower)
                    pending.extend(met if met.dim() == 2 else [met])
                norm = self.optimizer_step()
                metrics["actor/grad_norm"].append(norm)
        for met in torch.stack(pending).cpu().tolist():           # one device->host transfer for all micro-batches
            metrics["actor/pg_loss"].append(met[0]); metrics["actor/pg_clipfrac_higher"].append(met[1])
            metrics["actor/pg_clipfrac_lower"].append(met[2]); metrics["actor/ppo_kl"].append(met[3])
            metrics["actor/entropy_loss"].append(met[4])
            if use_ref:
                metrics["actor/kl_loss"] = met[5]                 # the reference overwrites these two (dp_actor.py:273-274)
                metrics["actor/kl_coef"] = h.kl_coef
        self.sched_steps += 1
        metrics["actor/lr"] = self.current_lr()                   # lr AFTER scheduler.step(), as fsdp_workers.py:453-455
        release_cached_blocks()
        return dict(metrics)


class CriticEngine(PolicyEngine):
    """DataParallelPPOCritic (verl/workers/critic/dp_critic.py) on the same engine: the backbone with a score head (VLConfig.value_head),
    the same pass planner / packing / gradient exchange / optimizer as the actor; only the head, the loss and the metric names differ."""

    @staticmethod
    def action_mask(attention_mask: torch.Tensor, R: int) -> torch.Tensor:
        """attention_mask[:, -R-1:-1] (dp_critic.py:174,193): the mask of the INPUT token at each value position, shifted left by one."""
        return torch.as_tensor(attention_mask)[:, -R - 1:-1]

    def compute_values(self, data: Dict[str, Any], micro_batch_size: Optional[int] = None) -> torch.Tensor:
        """dp_critic.py:140-175: (N, R) fp32 values of the states before each response token, times the action mask."""
        N = data["input_ids"].shape[0]
        mb = micro_batch_size or (self.h.micro_batch_size_per_device_for_experience if self.h else 16)
        R = data["responses"].shape[1]
        p_len, r_len, keys = self._row_stats(data, R)
        passes = self._plan_passes(0, N, mb, max(1, int(self.fuse_experience)), self.tokens_per_pass_nograd, p_len, r_len, keys)
        self.last_plan["experience"] = passes
        outs = []
        for (a, b_) in passes:
            b = self._stage(data, slice(a, b_))
            outs.append(self.model.values(b))
        out = torch.cat(outs, 0)
        del outs, b
        release_cached_blocks()
        return out * self.action_mask(data["attention_mask"], R).to(out.device, out.dtype)

    def update_critic(self, data: Dict[str, Any]) -> Dict[str, List[float]]:
        """dp_critic.py:177-225.  data: input_ids, attention_mask, position_ids, responses, values, returns, [multi_modal_inputs]."""
        h = self.h
        N = data["input_ids"].shape[0]
        R = data["responses"].shape[1]
        dev = self.store.device
        metrics: Dict[str, List[float]] = defaultdict(list)
        pending = []
        mini, micro = h.global_batch_size_per_device, h.micro_batch_size_per_device_for_update
        assert N % mini == 0 and mini % micro == 0, (N, mini, micro)
        accum = mini // micro
        p_len, r_len, keys = self._row_stats(data, R)
        plan = self.last_plan["update"] = []
        for _ in range(h.ppo_epochs):
            for m0 in range(0, N, mini):
                passes = self._plan_passes(m0, m0 + mini, micro, max(1, int(self.fuse_micro_batches)), self.tokens_per_pass_grad, p_len, r_len, keys)
                plan.extend(passes)
                for (s, e) in passes:
                    sl = slice(s, e)
                    red = self.grad_reducer() if (self.overlap_allreduce and e >= m0 + mini) else None
                    b = self._stage(data, sl)
                    to = lambda k, dt=F32: ops.h2d(torch.as_tensor(data[k][sl]), dt, dev)
                    loss_in = dict(values=to("values"), returns=to("returns"),
                                   action_mask=ops.h2d(self.action_mask(data["attention_mask"][sl], R), I64, dev))
                    _, met = self.model.value_forward_backward(b, loss_in, cliprange_value=h.cliprange_value, grad_accum=float(accum), loss_rows=micro,
                                                               on_final=red.ready if red is not None else None,
                                                               train_vision=not h.freeze_vision_tower)
                    pending.extend(met if met.dim() == 2 else [met])
                norm = self.optimizer_step()
                metrics["critic/grad_norm"].append(norm)
        for met in torch.stack(pending).cpu().tolist():
            metrics["critic/vf_loss"].append(met[0]); metrics["critic/vf_clipfrac"].append(met[1]); metrics["critic/vpred_mean"].append(met[2])
        self.sched_steps += 1
        metrics["critic/lr"] = self.current_lr()
        release_cached_blocks()
        return dict(metrics)

