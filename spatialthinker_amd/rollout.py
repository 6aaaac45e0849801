"""Rollout generator on MI355X: packed prefill + KV-cache decode, n samples per prompt.

Replaces vLLMRollout.generate_sequences (verl/workers/rollout/vllm_rollout_spmd.py:115-188) and the
FSDP->vLLM weight hand-off (verl/workers/sharding_manager/fsdp_vllm.py:76-116): the generator reads the
actor's own flat weight buffer, so it always samples from the post-update policy with zero copies.

Design: the n rollouts of a prompt share the prompt's KV (prefilled once); each decode step runs the
MFMA attention kernel twice per layer — one tile per (prompt, kv head) over the shared prompt keys for all
n x (n_q/n_kv) query rows, one per sample over its own generated keys — and merges the two partials.
Sampling is exact multinomial (Gumbel-max) with a counter RNG keyed by (seed, step, row).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import os
import time

import numpy as np
import torch

from . import indexing as ix
from . import ops
from .model import BF16, F32, I32, I64, Qwen25VL, drop_pixel_cache, pixels_on_device

# per-sample decode attention on the one-wave-per-item kernel (st_attn_decode_rows; same partials up to fp32 rounding of the online softmax).
# Round 6, measured and left OFF (profiles/r06_notes.md §2): equal at 512 rows (per layer 34 + 29 us vs 62 us in one launch at 100-token mean
# contexts, 34 + 65 vs 101 at 300), 7 % SLOWER per iteration at <= 64 rows — both kernels sit on the CU's LDS-DMA issue rate (16 one-KiB
# copies per 16-KiB tile), not on workgroup lifetime or occupancy, which is what the one-wave kernel changes.  ST_DECODE_ROWS=1 selects it.
DECODE_ROWS_DEFAULT = os.environ.get("ST_DECODE_ROWS", "0") == "1"


class Generator:
    def __init__(self, model: Qwen25VL, prefill_chunk_tokens: int = 32768, autotune: bool = False, fused_decode: bool = True):
        self.m = model
        # sequences decoded together (rows of the decode GEMMs), <= 512.  An iteration streams every weight once whatever the row count, so
        # the rollout time is (iterations) x (cost of an iteration): one 512-row wave needs ~920 iterations at 16.5 -> 7.9 ms (two row
        # tiles until half of the rows have finished, then one) where two 256-row waves + pooled survivors need ~1430 at 7.9 ms:
        # 11.9 s instead of 13.3 s for the bench's 512 rollouts (tools/gen_phases.py, round 2)
        self.max_decode_batch = int(os.environ.get("ST_MAX_DECODE", "512"))
        self.decode_rows_kernel = DECODE_ROWS_DEFAULT      # per-sample decode partials on st_attn_decode_rows (opt-in, ST_DECODE_ROWS=1: measured no faster)
        self.compact = True               # restart the decode graph on the survivors once half of a phase's rows have finished
        self.fused_decode = fused_decode  # fused decode epilogues (bit-identical to the unfused launch chain; tests compare both)
        self.prefill_chunk_tokens = prefill_chunk_tokens
        self.autotune = autotune          # time the decode GEMM tile/split-K candidates once per (batch, weight shape)
        # decode-loop accounting for the HBM roofline of the rollout (bench.py `roofline_decode`): seconds inside the replayed
        # decode iterations (device events), iterations, and the ALGORITHMIC bytes one iteration must read — every LM weight once
        # + the K/V of the live context (prompt K/V once per prompt and kv-head group, generated K/V per row)
        self.stats = {"decode_s": 0.0, "decode_steps": 0, "decode_bytes": 0.0, "decode_row_steps": 0, "decode_flops": 0.0, "prefill_s": 0.0, "phases": 0}
        self._timers: list = []
        self.tap = None                   # test hook, see iteration(); needs use_graph=False

    def _tune_decode(self, B: int):
        w = self.m.p.w
        head = w["lm_head"] if "lm_head" in w else w["embed"]
        L = self.m.cfg.num_layers
        for name in ("qkv_w", "o_w", "gu_w", "down_w"):
            ops.autotune_decode_gemm(B, [w[f"l.{i}.{name}"] for i in range(L)])
        ops.autotune_decode_gemm(B, head)                      # single weight: only tuned when it alone exceeds the cache

    # ------------------------------------------------------------------ memory plan of one call
    def rollout_bytes(self, prompt_tokens: int, n_prompts: int, n: int, R: int) -> float:
        """HBM one generate() call holds at its peak for `n_prompts` prompts of `prompt_tokens` valid tokens in total: prompt K/V of
        every layer, the per-sample generated K/V [layer][sample][R], the pending logits of every sample, the decode wave's working set
        (logits + attention partials of <= max_decode_batch rows) and the prefill activations of one chunk (~0.2 MB per packed token)."""
        c = self.m.cfg
        width, L, B = c.num_kv_heads * c.head_dim, c.num_layers, n_prompts * n
        kv_prompt = 2.0 * L * (prompt_tokens + 128) * width * 2
        kv_gen = 2.0 * L * B * R * width * 2
        logits = 2.0 * B * c.vocab_size
        wave = min(B, self.max_decode_batch)
        work = wave * c.vocab_size * 2.0 * 3 + wave * c.num_heads * c.head_dim * 2.0 * 16
        prefill = 0.2e6 * min(prompt_tokens, self.prefill_chunk_tokens) * (c.hidden_size / 3584.0)
        return kv_prompt + kv_gen + logits + work + prefill

    def plan_prompt_chunks(self, lens: np.ndarray, n: int, R: int, budget_bytes: Optional[float] = None) -> List[tuple]:
        """Cut the prompts of one call into consecutive chunks whose rollouts fit the free HBM (the K/V caches of 128 prompts x 8 rollouts
        at the shipped scripts' limits — 6144-token prompts, 2048-token responses, scripts/spatialthinker_7b_grpo.sh:23,33-34 — are
        45 + 120 GB at 7B, next to 131 GB of weights, gradients, optimizer state and the frozen reference).  One chunk = the whole call
        whenever it fits.  Raises a RuntimeError with the numbers when not even a single prompt's rollouts fit — never an OOM from the
        middle of the decode loop."""
        if budget_bytes is None:
            if not torch.cuda.is_available():
                return [(0, len(lens))]
            free_b, _ = torch.cuda.mem_get_info()
            cached = torch.cuda.memory_reserved() - torch.cuda.memory_allocated()          # blocks the allocator can hand out again
            budget_bytes = float(os.environ.get("ST_ROLLOUT_MEM_FRACTION", "0.85")) * (free_b + cached)
        chunks, a = [], 0
        while a < len(lens):
            b = a + 1
            if self.rollout_bytes(int(lens[a:b].sum()), 1, n, R) > budget_bytes:
                raise RuntimeError(
                    f"rollout does not fit: prompt {a} ({int(lens[a])} tokens) x {n} rollouts x {R} response tokens needs "
                    f"{self.rollout_bytes(int(lens[a]), 1, n, R) / 2 ** 30:.1f} GB of K/V cache and logits, {budget_bytes / 2 ** 30:.1f} GB are free "
                    f"(lower data.max_response_length / worker.rollout.n, or ST_MAX_DECODE)")
            while b < len(lens) and self.rollout_bytes(int(lens[a:b + 1].sum()), b + 1 - a, n, R) <= budget_bytes:
                b += 1
            chunks.append((a, b))
            a = b
        return chunks

    @torch.no_grad()
    def generate(self, input_ids, attention_mask, position_ids, *, n: int, max_new_tokens: int, return_prompt_cache: bool = False,
                 pixel_values: Optional[Sequence] = None, image_grid_thw: Optional[Sequence] = None,
                 forced_lengths: Optional[np.ndarray] = None, **kw):
        """generate_chunk() over the whole call when its caches fit the free HBM, otherwise over consecutive chunks of prompts
        (plan_prompt_chunks).  Samples keep their GLOBAL row ids for the counter RNG, so the tokens do not depend on the chunking.  A
        chunked call returns no prompt cache (it would have to outlive the chunks): the old-policy pass then recomputes the prompts."""
        drop_pixel_cache()                                       # a new rollout = new prompts: every image pays its one host -> device copy again
        mask_np = np.asarray(attention_mask.cpu() if torch.is_tensor(attention_mask) else attention_mask)
        lens = mask_np.sum(1).astype(np.int64)
        chunks = self.plan_prompt_chunks(lens, n, max_new_tokens)
        self.last_chunks = chunks
        if len(chunks) == 1:
            return self.generate_chunk(input_ids, attention_mask, position_ids, n=n, max_new_tokens=max_new_tokens,
                                       return_prompt_cache=return_prompt_cache, pixel_values=pixel_values, image_grid_thw=image_grid_thw,
                                       forced_lengths=forced_lengths, **kw)
        outs, lps = [], []
        emit = bool(kw.get("emit_log_probs"))
        for k_ in ("on_finished", "tail_stream", "tail_rows"):   # the hook's sample ids and prompt cache are per call: a chunked call runs without
            kw.pop(k_, None)
        self.last_finish_sets = []
        for (a, b) in chunks:
            outs.append(self.generate_chunk(input_ids[a:b], attention_mask[a:b], position_ids[a:b], n=n, max_new_tokens=max_new_tokens,
                                            return_prompt_cache=False, pixel_values=None if pixel_values is None else pixel_values[a:b],
                                            image_grid_thw=None if image_grid_thw is None else image_grid_thw[a:b],
                                            forced_lengths=None if forced_lengths is None else forced_lengths[a * n:b * n], rng_row_offset=a * n, **kw))
            if emit:                                               # (responses, {log_probs, ...}) per chunk
                outs[-1], d = outs[-1]
                lps.append(d["log_probs"])
        out = torch.cat(outs, 0)
        self.last_finish_sets = []          # (the chunks' sets hold chunk-local ids)
        if emit:                            # same contract as generate_chunk: (responses, dict) whenever emit is set, with or without
            # return_prompt_cache — no prompt K/V outlives the chunks, but the rollout's own log-probs do
            return out, dict(log_probs=torch.cat(lps, 0), responses=out, temperature=float(kw.get("temperature", 1.0)),
                             weights_version=getattr(self.m.p, "version", 0))
        return (out, None) if return_prompt_cache else out

    @torch.no_grad()
    def generate_chunk(self, input_ids, attention_mask, position_ids, *, n: int, max_new_tokens: int, temperature: float = 1.0,
                       eos_token_id=(151645,), pad_token_id: int = 151643, seed: int = 0, pixel_values: Optional[Sequence] = None,
                       image_grid_thw: Optional[Sequence] = None, forced_lengths: Optional[np.ndarray] = None, ignore_eos: bool = False,
                       sync_every: int = 4, use_graph: bool = True, top_k: int = -1, top_p: float = 1.0,
                       return_prompt_cache: bool = False, rng_row_offset: int = 0, emit_log_probs: bool = False,
                       on_finished=None, tail_stream=None, tail_rows: int = 128):
        """input_ids / attention_mask (b, P) left-padded, position_ids (b, 3, P) (or (b, P) text-only); per-prompt lists
        pixel_values[i] (N_i, 1176) / image_grid_thw[i] (1, 3).  Returns responses (b*n, max_new_tokens) int64 on the
        device, prompt-major, padded with pad_token_id after the first EOS (vllm_rollout_spmd.py:144-147).
        forced_lengths (b*n,): synthetic-benchmark mode — EOS is forced at exactly that response length.
        return_prompt_cache: also return {kp, vp, last_h, p_off, prompt_ids, prompt_mask, n} — the prompt K/V and last hidden
        states of the prefill; PolicyEngine.compute_log_prob re-uses them for the old-policy log-probs (same weights).
        emit_log_probs (round 4, opt-in): the decode loop also records log softmax(logits / T)[token] of every token it samples —
        log pi_old(a_t | s_t) of the policy that drew it, from the logits the sampler reads anyway (vLLM's `logprobs`).  Returned as
        cache["log_probs"] (b*n, R) fp32 (0 behind the end of a response) with cache["responses"] / ["temperature"] for the identity
        check of PolicyEngine.compute_log_prob(use_rollout_log_probs=True).
        on_finished / tail_stream / tail_rows (round 5): the decode tail runs beside other work.  Every time a decode phase ends,
        on_finished(sample_ids, out, event, prompt_cache) is called for the samples that finished in it — `out` the (b*n, R) response
        tensor whose rows `sample_ids` are final once `event` (recorded on the phase's stream) has passed, `prompt_cache` the prefill's
        K/V — after the NEXT phase's first iterations are queued, so the hook's host work (staging a log-prob pass for another stream)
        hides behind them.  Phases of <= tail_rows rows run on `tail_stream` (a CU-range stream, ops.cu_range_stream): their kernels
        keep to its compute units and a pass on the complementary range runs at its own speed (tools/probes/cu_mask_probe.hip).
        self.last_finish_sets records the sets in hook order (the last phase's finishers included, for which no hook runs)."""
        m, c, w = self.m, self.m.cfg, self.m.p.w
        dev = self.m.p.device
        dbg_t = []
        def mark(what):                                        # ST_GEN_DEBUG: host wall-clock of the sections of a call
            if os.environ.get("ST_GEN_DEBUG"):
                torch.cuda.synchronize(); dbg_t.append((what, time.perf_counter()))
        mark("entry")
        ids_np, mask_np, pos_np = (np.asarray(x.cpu() if torch.is_tensor(x) else x) for x in (input_ids, attention_mask, position_ids))
        nb, P = ids_np.shape
        if pos_np.ndim == 2:
            pos_np = np.repeat(pos_np[:, None, :], 3, 1)
        B, R = nb * n, max_new_tokens
        D, nq, nkv, L = c.head_dim, c.num_heads, c.num_kv_heads, c.num_layers
        g, width = nq // nkv, nkv * D
        eos = [eos_token_id] if isinstance(eos_token_id, int) else list(eos_token_id)
        lens = mask_np.sum(1).astype(np.int64)
        # ---------------- prefill (chunks of prompts), prompt KV per layer: (Tp, n_kv*D)
        p_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        Tp = int(p_off[-1])
        kp = torch.empty(L, Tp + 128, width, dtype=BF16, device=dev)
        vp = torch.empty(L, Tp + 128, width, dtype=BF16, device=dev)
        last_h = torch.empty(nb, c.hidden_size, dtype=BF16, device=dev)
        mark("prompt K/V buffers")
        ev_p0, ev_p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev_p0.record()
        i0 = 0
        while i0 < nb:
            i1, tok = i0, 0
            while i1 < nb and (i1 == i0 or tok + lens[i1] <= self.prefill_chunk_tokens):
                tok += int(lens[i1]); i1 += 1
            px = gr = None
            if pixel_values is not None:
                px = torch.cat([pixels_on_device(pixel_values[i], dev) for i in range(i0, i1)], 0)
                gr = np.concatenate([np.asarray(image_grid_thw[i]).reshape(-1, 3) for i in range(i0, i1)], 0)
            b = m.stage(ids_np[i0:i1], mask_np[i0:i1], pos_np[i0:i1], 0, px, gr)
            T, base = b.pk.T, int(p_off[i0])

            def keep_kv(layer, k, v, T=T, base=base):
                kp[layer, base:base + T].copy_(k[:T]); vp[layer, base:base + T].copy_(v[:T])

            x = m._embed(b, None)
            for layer in range(L):
                x = m._lm_layer_fwd(layer, x, b, None, kv_out=keep_kv)
            rows = torch.from_numpy((b.pk.cu_seqlens[1:] - 1).astype(np.int32)).to(dev)
            ops.rows_gather(x, rows, out=last_h[i0:i1])
            i0 = i1
        ev_p1.record()
        mark("prefill (host staging + forward)")
        head = w["embed"] if c.tie_word_embeddings else w["lm_head"]
        hook_cache = dict(kp=kp, vp=vp, last_h=last_h, p_off=p_off.astype(np.int64), prompt_ids=ids_np, prompt_mask=mask_np, n=n,
                          weights_version=getattr(self.m.p, "version", 0)) if on_finished is not None else None
        # ---------------- decode state (per SAMPLE, global over the phases below)
        rep = torch.arange(nb, device=dev, dtype=I32).repeat_interleave(n)
        hn, _ = ops.rmsnorm_fwd(ops.rows_gather(last_h, rep), w["final_norm"], c.rms_eps, want_rstd=False)
        out = torch.full((B, R), pad_token_id, dtype=I64, device=dev)
        logp_g = torch.zeros(B, R, dtype=F32, device=dev) if emit_log_probs else None
        kg = torch.empty(L, B, R, width, dtype=BF16, device=dev)
        vg = torch.empty(L, B, R, width, dtype=BF16, device=dev)
        gen_len_g = torch.zeros(B, dtype=I32, device=dev)      # tokens generated so far = response index of the next token
        last_pos = torch.from_numpy(np.ascontiguousarray(pos_np[:, :, -1].T)).to(dev, I32).repeat_interleave(n, dim=1)   # (3, B)
        pos_g = (last_pos + 1).contiguous()                    # position of the token sampled at response index 0
        eos_t = torch.tensor(eos, device=dev, dtype=I64)
        forced_len_g = None if forced_lengths is None else torch.as_tensor(forced_lengths, device=dev, dtype=I32)
        # split-KV chunk sizes of the decode attention (keys per workgroup): prompt keys / generated keys
        CK = int(os.environ.get("ST_DECODE_CKP", "576"))          # prompt K/V keys per partial: 256 -> 576 measured -2 % decode time (fewer partials to merge)
        CKG = int(os.environ.get("ST_DECODE_CKG", "512"))   # same-box A/B at the bench workload: gen 12.04 s (256) -> 11.74 s (512), 11.79 s (1024)
        C = max(1, int(-(-int(lens.max()) // CK)))
        Cg = max(1, -(-R // CKG))
        NP = C + Cg
        pb, pe = p_off[:-1].astype(np.int64), p_off[1:].astype(np.int64)
        ti = lambda a_: torch.from_numpy(np.ascontiguousarray(a_)).to(dev, I32)
        logits_g = torch.empty(B, c.vocab_size, dtype=BF16, device=dev)        # pending logits of every sample
        ops.gemm_nt(hn, head, out=logits_g)
        w_bytes = 2.0 * (sum(w[f"l.{i}.{nm}"].numel() for i in range(L) for nm in ("qkv_w", "o_w", "gu_w", "down_w")) + head.numel())
        can_fuse = self.fused_decode and c.hidden_size <= 4096 and c.hidden_size % 8 == 0
        assert can_fuse or not emit_log_probs, "emit_log_probs lives in the fused decode step (st_decode_step)"
        wave = max(1, self.max_decode_batch)
        if self.autotune and wave <= ops.DECODE_MAX_ROWS:
            self._tune_decode(ix.round_up(min(B, wave), 32))

        pending_hooks: list = []                               # (sample ids, event) of finished phases whose on_finished call is still due
        self.last_finish_sets = []

        hooks_live = [tail_stream is None]                     # with a tail stream the hooks only fire while a phase runs ON it: a pass
        # on the complementary CU range next to a full-chip phase would push that phase onto the few CUs the pass leaves free

        def run_hooks():
            while hooks_live[0] and pending_hooks:
                fin, ev = pending_hooks.pop(0)
                if on_finished is not None and len(fin):
                    on_finished(fin, out, ev, hook_cache)

        def decode_phase(S_np: np.ndarray, until_half: bool):
            """Decode the samples S_np (sorted ids; they may sit at different response indices) until none is live or — with
            until_half — at most half of the phase's rows are (finished rows would otherwise keep occupying the GEMM tiles; the
            caller re-batches the survivors, possibly together with those of other waves).  Returns the surviving ids."""
            if debug:
                torch.cuda.synchronize(); t_ph = [time.perf_counter()]
            Ba = len(S_np)
            Bp = ix.round_up(Ba, 32) if Ba <= 256 else ix.round_up(Ba, 128)
            fused = can_fuse and Bp <= ops.DECODE_MAX_ROWS
            if emit_log_probs and not fused:
                # the log-probability of a sampled token is written by st_decode_step (fused path only): an unfused wave (ST_MAX_DECODE above
                # DECODE_MAX_ROWS) would hand back zeros that PolicyEngine.rollout_log_probs cannot tell from real values
                raise RuntimeError(f"emit_log_probs needs the fused decode step, but a wave of {Bp} rows exceeds DECODE_MAX_ROWS = {ops.DECODE_MAX_ROWS} "
                                   f"(ST_MAX_DECODE = {self.max_decode_batch}): lower ST_MAX_DECODE or turn old_log_probs_from_rollout off")
            S_t = ti(S_np)
            S_l = S_t.long()
            S_rng = S_t + int(rng_row_offset) if rng_row_offset else S_t       # the counter RNG is keyed by the sample's row in the WHOLE call
            rows_all = Ba * g
            gen_len = gen_len_g[S_l].contiguous()
            pos = pos_g[:, S_l].contiguous()
            active = torch.ones(Ba, dtype=I32, device=dev)
            forced_len = None if forced_len_g is None else forced_len_g[S_l].contiguous()
            out_l = out[S_l].contiguous()
            logp_l = logp_g[S_l].contiguous() if emit_log_probs else None
            lse_scratch = torch.empty(Ba * 32, dtype=F32, device=dev) if emit_log_probs else None
            ar = torch.arange(Ba, device=dev, dtype=I32)
            # prompt partial: one "sequence" per (key chunk c, prompt p present in this phase) -> slab c (flash-decoding split-KV);
            # the samples of a prompt are consecutive local rows
            prom = S_np // n
            pids, first, cnt = np.unique(prom, return_index=True, return_counts=True)
            kb1_np = np.concatenate([np.minimum(pb[pids] + c_ * CK, pe[pids]) for c_ in range(C)])
            ke1_np = np.concatenate([np.minimum(pb[pids] + (c_ + 1) * CK, pe[pids]) for c_ in range(C)])
            qb1_np = np.tile(first * g, C)
            qe1_np = np.tile((first + cnt) * g, C)
            ob1_np = np.concatenate([c_ * rows_all + first * g for c_ in range(C)])
            qb1, qe1, kb1, ke1, ob1 = ti(qb1_np), ti(qe1_np), ti(kb1_np), ti(ke1_np), ti(ob1_np)
            max_q1 = int(cnt.max()) * g
            # generated partial: one "sequence" per (key chunk c, local row); chunks beyond the current length are empty ranges
            qb2 = (ar * g).repeat(Cg).contiguous(); qe2 = qb2 + g
            kbase = (S_t * R).repeat(Cg)
            kb2 = (kbase + torch.arange(Cg, device=dev, dtype=I32).repeat_interleave(Ba) * CKG).contiguous()
            ob2 = ((C + torch.arange(Cg, device=dev, dtype=I32).repeat_interleave(Ba)) * rows_all + (ar * g).repeat(Cg)).contiguous()
            parts = torch.empty(NP * rows_all, width, dtype=BF16, device=dev)
            lse_parts = torch.empty(nkv, NP * rows_all, dtype=F32, device=dev)
            # ONE attention launch per layer: prompt partials (keys = prefix range in the prompt cache, own range empty) and
            # generated partials (own range in the sample's cache, prefix empty) are items of the same grid
            n1, n2 = qb1.numel(), qb2.numel()
            z1, z2 = torch.zeros(n1, dtype=I32, device=dev), torch.zeros(n2, dtype=I32, device=dev)
            qb_all, qe_all, ob_all = torch.cat([qb1, qb2]), torch.cat([qe1, qe2]), torch.cat([ob1, ob2])
            kb_all, ke_all = torch.cat([z1, kb2]), torch.cat([z1, kb2])        # ke_all[n1:] is refreshed every step
            pb_all, pe_all = torch.cat([kb1, z2]), torch.cat([ke1, z2])
            max_q_all = max(max_q1, g)
            xbuf = torch.zeros(Bp, c.hidden_size, dtype=BF16, device=dev)
            abuf = torch.zeros(Bp, nq * D, dtype=BF16, device=dev)          # attention output, pad rows stay zero
            qbuf = torch.zeros(Bp, nq * D, dtype=BF16, device=dev)          # roped queries of the fused path (pad rows stay zero)
            logits = torch.empty(Bp if Bp <= ops.DECODE_MAX_ROWS else Ba, c.vocab_size, dtype=BF16, device=dev)
            logits[:Ba].copy_(logits_g[S_l])
            tok32 = torch.zeros(Ba, dtype=I32, device=dev)
            pad_t = torch.full((Ba,), pad_token_id, dtype=I64, device=dev)
            kgv, vgv = kg.view(L, B * R, width), vg.view(L, B * R, width)
            s_first = int(S_np[0])
            assert fused or np.array_equal(S_np, np.arange(s_first, s_first + Ba)), "the unfused path decodes contiguous waves only"

            # state of the one-launch step bookkeeping (st_decode_step) of the fused path
            kbase_row = (S_t * R).contiguous()                               # first cache row of every local row's sample
            slot = torch.zeros(Ba, dtype=I32, device=dev)                    # cache slot of the current token's K/V
            cos_b = torch.empty(Ba, D // 2, dtype=F32, device=dev); sin_b = torch.empty_like(cos_b)
            samp_scratch = torch.empty(Ba * 33, dtype=F32, device=dev)
            ke_gen = ke_all[n1:]                                            # view: the generated-key range ends inside the launch arrays
            rows_kernel = self.decode_rows_kernel and D == 128 and g <= 32
            fp8_gu = bool(getattr(m, "fp8", False) and getattr(m.p, "wq", None) and Bp > 256 and c.hidden_size % 128 == 0 and c.intermediate_size % 128 == 0
                          and os.environ.get("ST_FP8_DECODE", "1") != "0")
            # ring depth of the one-wave kernel: 2 slots (32 KiB, five items per CU) while the items alone fill the chip, 4 slots (three
            # tiles in flight per item) for the late phases whose few items are latency chains (ST_DECODE_ROWS_SLOTS overrides)
            rows_slots = int(os.environ.get("ST_DECODE_ROWS_SLOTS", "0")) or (2 if Ba * nkv >= 1024 else 4)

            def iteration():
                """sample -> record -> one decode forward for the phase's rows -> next logits.  Device state only (graph-capturable).
                Finished rows keep computing on their last token until the phase is re-batched; their output is ignored."""
                if fused:
                    # two launches: the split sampler, then ONE kernel for everything between two forwards (sampler finish + forced
                    # EOS, token record, live flags, response index, cache slot, key-range ends, M-RoPE table rows + position
                    # advance, embedding gather) — the unfused branch below is the same sequence as ~25 torch kernels
                    live = active.bool() if self.tap is not None else None
                    ops.sample_partials(logits[:Ba], temperature, seed, samp_scratch, row_steps=gen_len, row_ids=S_rng, top_k=top_k, top_p=top_p,
                                        lse_partials=lse_scratch)
                    ops.decode_step(samp_scratch, forced_len=forced_len, forced_token=int(eos[0]), eos_ids=eos_t, ignore_eos=ignore_eos,
                                    gen_len=gen_len, active=active, out_tokens=out_l, tok_out=tok32, slot_out=slot, k_base=kbase_row,
                                    kb_gen=kb2, ke_gen=ke_gen, n_chunks=Cg, chunk_keys=CKG, pos=pos, inv_freq=m.inv_freq, D=D,
                                    section=c.mrope_section, cos_out=cos_b, sin_out=sin_b, embed=w["embed"], x_out=xbuf,
                                    lse_partials=lse_scratch, logits=logits if emit_log_probs else None, temperature=temperature, logp_out=logp_l)
                    cos, sin, glen, x = cos_b, sin_b, slot, xbuf
                else:
                    forced = None
                    if forced_len is not None:
                        forced = torch.where(forced_len == gen_len + 1, int(eos[0]), -1).to(I32)
                    ops.sample(logits[:Ba], temperature, seed, forced=forced, row_steps=gen_len, out=tok32, row_ids=S_rng, top_k=top_k, top_p=top_p)
                    tok = tok32.to(I64)
                    live = active.bool()
                    col = gen_len.clamp(max=R - 1).long()[:, None]
                    out_l.scatter_(1, col, torch.where(live, tok, out_l.gather(1, col)[:, 0])[:, None])
                    stop = gen_len + 1 >= R                                             # the length cap ends a sample like an EOS
                    if not ignore_eos:
                        stop = stop | (tok[:, None] == eos_t[None, :]).any(1)
                    active.copy_((live & ~stop).to(I32))
                    cos, sin = ops.mrope_table(pos, m.inv_freq, D, c.mrope_section)
                    ops.embed_gather(w["embed"], tok32, out=xbuf[:Ba])
                    x = xbuf
                    glen = gen_len.clamp(max=R - 1)                                     # finished rows at the cap rewrite their last slot
                    ke2 = torch.maximum(torch.minimum(kb2 + CKG, kbase + (glen + 1).repeat(Cg)), kb2).contiguous()
                    ke_all[n1:].copy_(ke2)
                if fused:
                    # 10 launches per layer: the split-K slabs of the projections are consumed by fused epilogues (bias + RoPE +
                    # cache append; residual + RMSNorm of the NEXT op) and the SwiGLU lives in the gate/up GEMM epilogue
                    H = c.hidden_size
                    h1, _ = ops.rmsnorm_fwd(x, w["l.0.in_norm"], c.rms_eps, want_rstd=False)
                    for layer in range(L):
                        p = f"l.{layer}."
                        slabs, sp = ops.gemm_nt_decode_slabs(h1, w[p + "qkv_w"])
                        ops.decode_finish_qkv(slabs, sp, Bp, w[p + "qkv_b"], cos, sin, qbuf, kg[layer], vg[layer], glen, Ba, nq, nkv, D,
                                              row_map=S_t)
                        if rows_kernel:
                            # round 6: the per-sample partials (7 query rows against the sample's own keys) run one WAVE per item
                            # (st_attn_decode_rows: five items per CU instead of two 256-thread workgroups with one computing wave
                            # each); the prompt partials (56 rows against 576 shared keys) stay on the workgroup kernel
                            ops.attn_fwd_ranges(qbuf, kgv[layer], vgv[layer], qb1, qe1, z1, z1, max_q1, nkv, nkv, D, m.scale, parts, lse_parts,
                                                o_beg=ob1, q_group=g, pre_beg=kb1, pre_end=ke1, k_pre=kp[layer], v_pre=vp[layer])
                            ops.attn_decode_rows(qbuf, kgv[layer], vgv[layer], qb2, qe2, kb2, ke_gen, g, nkv, D, m.scale, parts, lse_parts,
                                                 o_beg=ob2, q_group=g, slots=rows_slots)
                        else:
                            ops.attn_fwd_ranges(qbuf, kgv[layer], vgv[layer], qb_all, qe_all, kb_all, ke_all, max_q_all, nkv, nkv, D, m.scale,
                                                parts, lse_parts, o_beg=ob_all, q_group=g, pre_beg=pb_all, pre_end=pe_all,
                                                k_pre=kp[layer], v_pre=vp[layer])
                        ops.attn_merge(parts, lse_parts, NP, nkv, D, out=abuf, q_group=g)
                        slabs, sp = ops.gemm_nt_decode_slabs(abuf, w[p + "o_w"])
                        x1 = torch.empty(Bp, H, dtype=BF16, device=dev); h2 = torch.empty(Bp, H, dtype=BF16, device=dev)
                        ops.decode_finish_norm(slabs, sp, Bp, H, residual=x, x_out=x1, norm_w=w[p + "post_norm"], eps=c.rms_eps, h_out=h2)
                        if fp8_gu:
                            # fp8 mode (config #5's arithmetic; not a parity mode): at 257..512 rows the gate/up product is MFMA-bound (1.1
                            # PF/s on the bf16 tile) — the training path's MX-fp8 tile with the SwiGLU epilogue runs it from the fp8 weight
                            # copy the forward passes already keep (round 6); the narrow projections stay on the bf16 split-K tiles
                            hq, hs = ops.mxfp8_quantize(h2)
                            mm = ops.gemm_mxfp8_swiglu(hq, hs, *m.p.wq[p + "gu_w"])
                        else:
                            mm = ops.gemm_swiglu_decode(h2, w[p + "gu_w"])
                        slabs, sp = ops.gemm_nt_decode_slabs(mm, w[p + "down_w"])
                        x = torch.empty(Bp, H, dtype=BF16, device=dev); h1 = torch.empty(Bp, H, dtype=BF16, device=dev)
                        nxt = w[f"l.{layer + 1}.in_norm"] if layer + 1 < L else w["final_norm"]
                        ops.decode_finish_norm(slabs, sp, Bp, H, residual=x1, x_out=x, norm_w=nxt, eps=c.rms_eps, h_out=h1)
                    hn2 = h1
                else:
                    for layer in range(L):
                        p = f"l.{layer}."
                        h1, _ = ops.rmsnorm_fwd(x, w[p + "in_norm"], c.rms_eps, want_rstd=False)
                        qkv = ops.gemm_nt(h1, w[p + "qkv_w"], bias=w[p + "qkv_b"], decode=True)
                        ops.rope_apply_(qkv[:Ba], cos, sin, nq + nkv, D)
                        ops.kv_append_(qkv[:Ba], nq * D, nq * D + width, width, kg[layer][s_first:s_first + Ba], vg[layer][s_first:s_first + Ba], glen)
                        ops.attn_fwd_ranges(qkv, kp[layer], vp[layer], qb1, qe1, kb1, ke1, max_q1, nkv, nkv, D, m.scale, parts, lse_parts,
                                            o_beg=ob1, q_group=g)
                        ops.attn_fwd_ranges(qkv, kgv[layer], vgv[layer], qb2, qe2, kb2, ke2, g, nkv, nkv, D, m.scale, parts, lse_parts,
                                            o_beg=ob2, q_group=g)
                        ops.attn_merge(parts, lse_parts, NP, nkv, D, out=abuf, q_group=g)      # writes the (B, n_q*D) layout directly
                        x1 = ops.gemm_nt(abuf, w[p + "o_w"], residual=x, decode=True)
                        h2, _ = ops.rmsnorm_fwd(x1, w[p + "post_norm"], c.rms_eps, want_rstd=False)
                        mm = ops.swiglu_fwd(ops.gemm_nt(h2, w[p + "gu_w"], decode=True))
                        x = ops.gemm_nt(mm, w[p + "down_w"], residual=x1, decode=True)
                    hn2, _ = ops.rmsnorm_fwd(x, w["final_norm"], c.rms_eps, want_rstd=False)
                ops.gemm_nt(hn2[:logits.shape[0]], head, out=logits, decode=True)
                if not fused:
                    gen_len.add_(1); pos.add_(1)              # (st_decode_step advances both itself)
                if self.tap is not None:                      # test hook (eager iterations only): this iteration's tokens and next logits
                    self.tap(S_np, live.clone(), gen_len - 1, tok32.clone(), logits[:Ba].clone())

            # the decode iteration is launch-bound (~10 launches x layers): capture it once per phase into a hipGraph and replay
            graph, done = None, 0
            if debug:
                torch.cuda.synchronize(); t_ph.append(time.perf_counter())
            if use_graph:
                iteration()                                        # eager warm-up iteration
                torch.cuda.synchronize()
                if debug:
                    t_ph.append(time.perf_counter())
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    iteration()
                done = 1                                           # capture itself does not execute
            if debug:
                torch.cuda.synchronize(); t_ph.append(time.perf_counter())
            n_live = Ba
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            len0 = gen_len.sum()
            len0_rows = gen_len.clone() if fused else None       # fused path: a finished row's length freezes (st_decode_step), see below
            ev0.record()
            done0 = done
            # The live count is read every `sync_every` iterations (a blocking read: the queue drains, ~0.1 ms of idle GPU per read).  A phase
            # overshoots its re-batch point by sync_every / 2 iterations on average, at the price of the phase's WIDEST row count: with 32
            # (rounds 1-4) the five phases of the bench's rollout wasted ~16 x (11.9 + 7.7 + 6.3 + 5.2 + 4.3) ms = 0.57 s per step, with 4
            # ~0.07 s + ~0.03 s of reads (round 5, profiles/r05_notes.md)
            while True:
                if done % sync_every == 0 or done >= R:
                    if done > done0:
                        run_hooks()                                # the iterations queued above cover the hook's host work
                    n_live = int(active.sum().item())
                    if n_live == 0 or (until_half and Bp > 32 and n_live <= Bp // 2):
                        break
                if graph is not None:
                    graph.replay()
                else:
                    iteration()
                done += 1
            ev1.record()
            if debug:
                torch.cuda.synchronize(); t_ph.append(time.perf_counter())
                print(f"[gen]   phase of {Ba} rows: setup {1e3 * (t_ph[1] - t_ph[0]):.1f} ms, eager iteration {1e3 * (t_ph[2] - t_ph[1]):.1f} ms, "
                      f"capture + instantiate {1e3 * (t_ph[3] - t_ph[2]):.1f} ms, replay loop {1e3 * (t_ph[4] - t_ph[3]):.1f} ms wall for {done - done0} iterations", flush=True)
            steps = done - done0
            if steps > 0:
                kv_row = 2 * width * 2 * L                                   # K and V bytes of one cached position over all layers
                prompt_ctx = float(sum(int(pe[p_] - pb[p_]) for p_ in pids))
                if fused:
                    # generated K/V the iterations HAD to read: a row attends to its own context only while it is live (finished rows have
                    # empty key ranges): sum over rows of (mean context while live) x (iterations it was live), per iteration of the phase
                    l0, l1 = len0_rows.double(), gen_len.double()
                    gen_ctx = float(((l0 + l1) * 0.5 * (l1 - l0)).sum().item()) / steps
                else:
                    gen_ctx = float(len0.item() + gen_len.sum().item()) / 2.0    # mean generated context per iteration, summed over the rows
                # algorithmic flops of the iterations: every live-or-not row of the phase runs the projections + lm_head (2 flop per weight)
                # and attends to its prompt + generated context (4 * D flop per (query head, key))
                row_ctx = float(sum(int(pe[p_] - pb[p_]) * int(c_) for p_, c_ in zip(pids, cnt))) + gen_ctx
                flops = steps * (Ba * w_bytes + 4.0 * D * nq * L * row_ctx)
                self._timers.append((ev0, ev1, steps, steps * (w_bytes + (prompt_ctx + gen_ctx) * kv_row), steps * Ba, flops))
            out[S_l] = out_l
            if emit_log_probs:
                logp_g[S_l] = logp_l
            gen_len_g[S_l] = gen_len
            pos_g[:, S_l] = pos
            if n_live == 0:
                return np.zeros(0, dtype=S_np.dtype)
            keep = active.bool()
            logits_g[S_l[keep]] = logits[:Ba][keep]
            return S_np[keep.cpu().numpy()]

        # ---------------- scheduler: fresh waves of <= max_decode_batch samples run until half of their rows have finished; the
        # survivors of all waves are then decoded TOGETHER (they sit at different response indices: every row carries its own
        # step), re-batched again each time half of them are done — one short tail for the whole rollout batch instead of one
        # per wave.  (The unfused path has no sample-indexed cache append: its waves simply run to completion.)
        compact = self.compact and can_fuse and wave <= ops.DECODE_MAX_ROWS
        debug = bool(os.environ.get("ST_GEN_DEBUG"))
        pool = np.zeros(0, dtype=np.int64)

        main_stream = torch.cuda.current_stream()
        last_stream = [main_stream]

        def run(S, until_half):
            if debug:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            st = tail_stream if (tail_stream is not None and len(S) <= tail_rows) else main_stream
            if st is not last_stream[0]:
                st.wait_stream(last_stream[0])                  # the sample-indexed state (out, K/V, pending logits) of the earlier phases
            hooks_live[0] = tail_stream is None or st is tail_stream
            with torch.cuda.stream(st):
                r = decode_phase(S, until_half)
                ev = torch.cuda.Event()
                ev.record(st)                                   # `out` holds the final tokens of this phase's finishers from here on
            last_stream[0] = st
            fin = np.setdiff1d(S, r)
            self.last_finish_sets.append(fin)
            pending_hooks.append((fin, ev))
            if debug:
                torch.cuda.synchronize()
                print(f"[gen] phase rows={len(S)} -> survivors={len(r)} in {time.perf_counter() - t0:.3f}s" + (" (tail stream)" if st is not main_stream else ""), flush=True)
            return r
        mark("decode state (K/V buffers of the samples, first logits)")
        for s0 in range(0, B, wave):
            S = np.arange(s0, min(B, s0 + wave), dtype=np.int64)
            pool = np.concatenate([pool, run(S, compact)])
        while len(pool):
            S, pool = np.sort(pool[:wave]), pool[wave:]
            pool = np.concatenate([pool, run(S, len(S) > 32)])
        if len(pending_hooks) > 1:                              # all but the LAST phase's finishers (those are the caller's, after the rollout)
            last = pending_hooks.pop()
            hooks_live[0] = True
            run_hooks()
        pending_hooks.clear()
        if last_stream[0] is not main_stream:
            main_stream.wait_stream(last_stream[0])
        torch.cuda.synchronize()
        mark("decode phases")
        if dbg_t:
            print("[gen] sections: " + ", ".join(f"{w_} {1e3 * (t_ - dbg_t[i_][1]):.0f} ms" for i_, (w_, t_) in enumerate(dbg_t[1:])), flush=True)
        self.stats["prefill_s"] += ev_p0.elapsed_time(ev_p1) * 1e-3
        self.stats["phases"] += len(self._timers)
        for ev0, ev1, steps, nbytes, rows, flops in self._timers:
            self.stats["decode_s"] += ev0.elapsed_time(ev1) * 1e-3
            self.stats["decode_steps"] += steps
            self.stats["decode_bytes"] += nbytes
            self.stats["decode_row_steps"] += rows
            self.stats["decode_flops"] = self.stats.get("decode_flops", 0.0) + flops
        self._timers = []
        del kg, vg, logits_g
        from .actor import release_cached_blocks
        release_cached_blocks()                       # the generated-token K/V and the logits of this rollout are gone: see actor.py
        if emit_log_probs and not return_prompt_cache:
            return out, dict(log_probs=logp_g, responses=out, temperature=float(temperature), weights_version=getattr(self.m.p, "version", 0))
        if return_prompt_cache:
            cache = dict(kp=kp, vp=vp, last_h=last_h, p_off=p_off.astype(np.int64), prompt_ids=ids_np, prompt_mask=mask_np, n=n,
                         weights_version=getattr(self.m.p, "version", 0))
            if emit_log_probs:
                cache.update(log_probs=logp_g, responses=out, temperature=float(temperature))
            return out, cache
        return out
