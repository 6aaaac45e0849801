"""Host-side integer bookkeeping of the packed (padding-free) Qwen2.5-VL layout.

Product code (numpy, runs on the driver / dataloader side like the reference's
`get_rope_index` does — verl/models/transformers/qwen2_vl.py:36-136).  The GPU never sees
padded (B, S) tensors: everything is converted here into row indices of the packed stream.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import os

import numpy as np


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# ------------------------------------------------------------------ M-RoPE position ids
def get_rope_index(input_ids: np.ndarray, image_grid_thw: Optional[np.ndarray], attention_mask: Optional[np.ndarray], *,
                   image_token_id: int, vision_start_token_id: int, spatial_merge_size: int = 2,
                   video_grid_thw: Optional[np.ndarray] = None, second_per_grid_ts: Optional[np.ndarray] = None,
                   video_token_id: Optional[int] = None, tokens_per_second: int = 2) -> np.ndarray:
    """(3, S) M-RoPE ids of ONE sequence; same contract as the reference function
    (verl/models/transformers/qwen2_vl.py:36-136): text tokens advance all three rows together, a vision block contributes a
    (t, h, w) grid shifted by the running offset, the text after it resumes at max+1, masked positions hold 1.  Images keep the temporal
    row at 0 (second_per_grid_t = 0, :83-87); a VIDEO block's temporal index is trunc(frame * second_per_grid_t * tokens_per_second)
    with second_per_grid_t = second_per_grid_ts[video] or 1.0 (:96-101, :117-118).  Blocks are taken in the order they appear in the
    sequence; the k-th image block uses image_grid_thw[k], the k-th video block video_grid_thw[k]."""
    ids = np.asarray(input_ids).astype(np.int64)
    S = ids.shape[0]
    mask = np.ones(S, dtype=np.int64) if attention_mask is None else np.asarray(attention_mask).astype(np.int64)
    no_img = image_grid_thw is None or len(image_grid_thw) == 0
    no_vid = video_grid_thw is None or len(video_grid_thw) == 0
    if no_img and no_vid:
        pos = np.cumsum(mask) - 1
        pos[mask == 0] = 1
        return np.tile(pos, (3, 1))
    if not no_vid and video_token_id is None:
        raise ValueError("video_grid_thw given without video_token_id")
    keep = np.nonzero(mask == 1)[0]
    toks = ids[keep]
    after_start = np.concatenate([[False], toks[:-1] == vision_start_token_id])
    is_vis = (toks == image_token_id) | ((toks == video_token_id) if video_token_id is not None else False)
    # a vision block = the run of image / video tokens that follows a vision-start token
    starts = np.nonzero(is_vis & after_start)[0]
    packed = np.empty((3, toks.shape[0]), dtype=np.int64)
    cursor, nxt = 0, 0                     # cursor: token index, nxt: next position value
    n_image = n_video = 0
    for st in starts:
        if toks[st] == image_token_id:
            t, h, w = (int(v) for v in image_grid_thw[n_image])
            n_image += 1
            sec = 0.0
        else:
            t, h, w = (int(v) for v in video_grid_thw[n_video])
            sec = float(second_per_grid_ts[n_video]) if second_per_grid_ts is not None else 1.0
            n_video += 1
        gh, gw = h // spatial_merge_size, w // spatial_merge_size
        n_txt = int(st) - cursor
        packed[:, cursor:st] = nxt + np.arange(n_txt)
        base = nxt + n_txt
        n_vis = t * gh * gw
        grid = np.indices((t, gh, gw)).reshape(3, -1)
        # float32 product truncated toward zero, as torch's `(t_index * second_per_grid_t * tokens_per_second).long()`
        grid[0] = (grid[0].astype(np.float32) * np.float32(sec) * np.float32(tokens_per_second)).astype(np.int64)
        packed[:, st:st + n_vis] = grid + base
        nxt = int(packed[:, st:st + n_vis].max()) + 1
        cursor = int(st) + n_vis
    packed[:, cursor:] = nxt + np.arange(toks.shape[0] - cursor)
    out = np.ones((3, S), dtype=np.int64)
    out[:, keep] = packed
    return out


# ------------------------------------------------------------------ vision tower indices
def vision_window_index(grid_thw: np.ndarray, merge: int, window: int, patch: int):
    """Window-major order of merged tokens + cumulative window lengths in patches (duplicates
    dropped) — HF get_vision_window_index (transformers/vision_utils.py:124-188)."""
    side = window // merge // patch
    unit = merge * merge
    order: List[np.ndarray] = []
    cu = [0]
    offset = 0
    for t, h, w in np.asarray(grid_thw).tolist():
        gh, gw = h // merge, w // merge
        # HF pads by a FULL window when the size is already a multiple (side - x % side)
        ph, pw = side - gh % side, side - gw % side
        canvas = -np.ones((t, gh + ph, gw + pw), dtype=np.int64)
        canvas[:, :gh, :gw] = np.arange(t * gh * gw).reshape(t, gh, gw)
        nh, nw = (gh + ph) // side, (gw + pw) // side
        tiles = canvas.reshape(t, nh, side, nw, side).swapaxes(2, 3).reshape(t * nh * nw, side * side)
        for tile in tiles:
            members = tile[tile >= 0]
            if members.size:
                order.append(members + offset)
                cu.append(cu[-1] + members.size * unit)
        offset += t * gh * gw
    return np.concatenate(order).astype(np.int64), np.asarray(cu, dtype=np.int32)


def vision_position_ids(grid_thw: np.ndarray, merge: int) -> np.ndarray:
    """(N, 2) (h, w) coordinates in pixel_values row order (merge-block-major) — HF
    get_vision_position_ids (transformers/vision_utils.py:81-121)."""
    out = []
    for t, h, w in np.asarray(grid_thw).tolist():
        hh = np.arange(h)[:, None].repeat(w, 1)
        ww = np.arange(w)[None, :].repeat(h, 0)
        blk = lambda a: a.reshape(h // merge, merge, w // merge, merge).swapaxes(1, 2).reshape(-1)
        out.append(np.tile(np.stack([blk(hh), blk(ww)], -1), (t, 1)))
    return np.concatenate(out, 0).astype(np.int64)


@dataclass
class VisionPlan:
    n_patches: int
    patch_gather: np.ndarray        # (N,) int32: window-ordered row -> pixel_values row
    merged_inverse: np.ndarray      # (N/4,) int32: original merged token -> window-ordered merged row
    cu_window: np.ndarray           # int32
    cu_image: np.ndarray            # int32 (per frame)
    max_window: int
    max_image: int
    cos: np.ndarray                 # (N, hd/2) fp32, window-ordered
    sin: np.ndarray


_VISION_PLANS: dict = {}


def plan_vision(grid_thw: np.ndarray, *, merge: int, window: int, patch: int, head_dim: int) -> VisionPlan:
    """Window order, inverse order, window / image boundaries and the 2-D rotary table of a list of image grids.  A function of the grids
    alone (48 ms of numpy for four 1344-patch images — with the GPU idle behind it at the head of every packed pass until round 6), so the
    plans of the last 256 distinct grid lists are kept (callers treat a plan's arrays as read-only: model.stage copies them to the device)."""
    import torch
    g = np.asarray(grid_thw).reshape(-1, 3)
    key = (g.astype(np.int64).tobytes(), merge, window, patch, head_dim)
    hit = _VISION_PLANS.get(key)
    if hit is not None:
        return hit
    plan = _plan_vision(g, merge=merge, window=window, patch=patch, head_dim=head_dim)
    if len(_VISION_PLANS) >= 256:
        _VISION_PLANS.pop(next(iter(_VISION_PLANS)))
    _VISION_PLANS[key] = plan
    return plan


def _plan_vision(g: np.ndarray, *, merge: int, window: int, patch: int, head_dim: int) -> VisionPlan:
    import torch
    unit = merge * merge
    widx, cu_win = vision_window_index(g, merge, window, patch)
    n = int((g[:, 0] * g[:, 1] * g[:, 2]).sum())
    patch_gather = (widx[:, None] * unit + np.arange(unit)[None, :]).reshape(-1)
    inv = np.empty_like(widx)
    inv[widx] = np.arange(widx.shape[0])                # == argsort(widx)
    lens = np.repeat(g[:, 1] * g[:, 2], g[:, 0])
    cu_img = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    # rotary table: HF Qwen2_5_VisionRotaryEmbedding (modeling_qwen2_5_vl.py:125-135), dim = head_dim/2, theta 1e4
    dim = head_dim // 2
    inv_freq = (1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))).numpy()
    pos = vision_position_ids(g, merge)[patch_gather]                                    # window order
    ang = torch.from_numpy((pos.astype(np.float32)[:, :, None] * inv_freq[None, None, :]).reshape(n, -1))
    return VisionPlan(n, patch_gather.astype(np.int32), inv.astype(np.int32), cu_win, cu_img,
                      int(np.diff(cu_win).max()), int(lens.max()), ang.cos().numpy(), ang.sin().numpy())


# ------------------------------------------------------------------ packing of (B, S) batches
@dataclass
class PackedBatch:
    T: int                          # valid tokens
    T_pad: int
    ids: np.ndarray                 # (T_pad,) int32, pad rows = 0
    pos: np.ndarray                 # (3, T_pad) int32
    cu_seqlens: np.ndarray          # (B+1,) int32
    max_seqlen: int
    image_rows: np.ndarray          # (n_img_tokens,) int32 rows of the packed stream holding image features
    embed_ids: np.ndarray           # (T_pad,) int32: token id, or -1 on image rows / pad rows (no embedding grad)
    logit_rows: np.ndarray          # (Tr,) int32 packed rows whose NEXT token is a valid response token
    labels: np.ndarray              # (Tr,) int64
    out_index: np.ndarray           # (Tr,) int64 flat index into (B, R) where each log-prob lands
    B: int
    R: int
    # segment view of the packed stream for the shared-prefix attention (st_attn_fwd_seg / st_attn_bwd_seg)
    seg_b: np.ndarray = None         # (n_seg,) int32 own rows [seg_b, seg_e)
    seg_e: np.ndarray = None
    pre_b: np.ndarray = None         # prefix rows every query of the segment sees (the shared prompt), empty when pre_b == pre_e
    pre_e: np.ndarray = None
    dep_e: np.ndarray = None         # rows [seg_e, dep_e) are queries outside the segment that see all of its keys
    max_seg: int = 0
    logit_dup: Optional[np.ndarray] = None     # (n_distinct, kmax) int32: positions in logit_rows sharing one packed row, -1 padded
    logit_distinct: Optional[np.ndarray] = None  # (n_distinct,) int32 the distinct packed rows
    first_prompt: Optional[np.ndarray] = None    # responses-only packing: prompt index whose cached last hidden state predicts slot 0


def pack_batch(input_ids: np.ndarray, attention_mask: np.ndarray, position_ids: np.ndarray, response_length: int, *,
               image_token_id: int, pad_multiple: int = 128, groups: Optional[Sequence[int]] = None, value_rows: bool = False) -> PackedBatch:
    """The padding-free transformation of verl/workers/actor/dp_actor.py:86-104,136-139 done once on
    the host: valid tokens are concatenated; log-probs are only needed at [:, -R-1:-1] of every row,
    and only where the response mask is set, so the lm_head runs on exactly those rows.

    groups (optional, one id per row): rows with the same id are rollouts of ONE prompt (identical prompt columns — verified
    here).  Such a group is packed as [prompt][response_1]...[response_k] with the prompt stored once; the segment arrays tell
    the attention which rows each query sees.  Every other row-wise op (norms, projections, MLP) then simply runs on ~half the
    tokens.  Without groups every row is its own stand-alone sequence (one segment, the reference's layout)."""
    ids = np.asarray(input_ids)
    mask = np.asarray(attention_mask).astype(bool)
    B, S = ids.shape
    R = response_length
    Pc = S - R                                           # prompt columns
    pos = np.asarray(position_ids)
    if pos.ndim == 2:                                   # text-only (B, S) -> replicate on the 3 rows
        pos = np.repeat(pos[:, None, :], 3, axis=1)
    # ---- group structure: members (row lists) in order of first appearance; only rows with identical prompts may share
    if groups is None:
        members = [[b] for b in range(B)]
    else:
        by_id = {}
        for b, gid in enumerate(groups):
            by_id.setdefault(gid, []).append(b)
        members = []
        for rows in by_id.values():
            r0 = rows[0]
            same = [r for r in rows if np.array_equal(ids[r, :Pc], ids[r0, :Pc]) and np.array_equal(mask[r, :Pc], mask[r0, :Pc])
                    and np.array_equal(pos[r, :, :Pc], pos[r0, :, :Pc])]
            members.append(same)
            members.extend([r] for r in rows if r not in same)       # a row that does not really share the prompt stands alone
        members.sort(key=lambda m: m[0])
    packed_row = -np.ones((B, S), dtype=np.int64)
    p_ids_l, p_pos_l = [], []
    seg = []                                            # (b, e, pre_b, pre_e, dep_e)
    lens_full = mask.sum(1)
    cur = 0
    for rows in members:
        r0 = rows[0]
        pcols = np.nonzero(mask[r0, :Pc])[0]
        P = len(pcols)
        shared = len(rows) > 1 and P > 0
        g_start = cur
        p_ids_l.append(ids[r0, pcols]); p_pos_l.append(pos[r0][:, pcols])
        for r in rows:
            packed_row[r, pcols] = cur + np.arange(P)
        cur += P
        resp_ranges = []
        for r in rows:
            rcols = Pc + np.nonzero(mask[r, Pc:])[0]
            n = len(rcols)
            p_ids_l.append(ids[r, rcols]); p_pos_l.append(pos[r][:, rcols])
            packed_row[r, rcols] = cur + np.arange(n)
            resp_ranges.append((cur, cur + n))
            cur += n
        if shared:
            seg.append((g_start, g_start + P, 0, 0, cur))
            for b_, e_ in resp_ranges:
                if e_ > b_:
                    seg.append((b_, e_, g_start, g_start + P, e_))
        elif cur > g_start:                              # stand-alone sequence: one causal segment over prompt + response
            seg.append((g_start, cur, 0, 0, cur))
    T = cur
    T_pad = max(round_up(T, pad_multiple), pad_multiple)
    p_ids = np.zeros(T_pad, dtype=np.int32)
    p_pos = np.zeros((3, T_pad), dtype=np.int32)
    if T:
        p_ids[:T] = np.concatenate(p_ids_l)
        p_pos[:, :T] = np.concatenate(p_pos_l, axis=1)
    # per-row cumulative lengths keep their historical meaning (full sequence lengths): metrics / FLOP counters use them
    cu = np.concatenate([[0], np.cumsum(lens_full)]).astype(np.int32)
    # response slot j of row b sits at column S-R+j; its log-prob comes from the logits of column S-R+j-1
    # value_rows (critic, dp_critic.py:113,174,193): every slot whose INPUT token is valid — attention_mask[:, -R-1:-1] — i.e. one more than
    # the response mask per row: the state after the last response token, whose value the GAE recursion reads (core_algos.py:100-110)
    cols = np.arange(Pc, S)
    valid = mask[:, cols - 1] if value_rows else (mask[:, cols] & mask[:, cols - 1])
    bb, jj = np.nonzero(valid)
    logit_rows = packed_row[bb, cols[jj] - 1]
    labels = ids[bb, cols[jj]].astype(np.int64)
    # image placeholders live in the prompt part only; a sampled response token that happens to equal the placeholder id is
    # ordinary text (HF's masked_scatter would raise on such a row; random-weight synthetic models do sample it)
    is_prompt_row = np.zeros(T_pad, dtype=bool)
    prow = packed_row[:, :Pc]
    is_prompt_row[prow[prow >= 0]] = True
    img_rows = np.nonzero((p_ids[:T] == image_token_id) & is_prompt_row[:T])[0].astype(np.int32)
    embed_ids = p_ids.copy()
    embed_ids[img_rows] = -1
    embed_ids[T:] = -1
    sa = np.asarray(seg, dtype=np.int32).reshape(-1, 5)
    # duplicates in logit_rows: the last prompt row of a group predicts the first response token of every member
    dup = distinct = None
    if len(logit_rows) and len(np.unique(logit_rows)) != len(logit_rows):
        distinct, inv = np.unique(logit_rows, return_inverse=True)
        counts = np.bincount(inv)
        dup = -np.ones((len(distinct), int(counts.max())), dtype=np.int32)
        fill = np.zeros(len(distinct), dtype=np.int64)
        for i_, u in enumerate(inv):
            dup[u, fill[u]] = i_; fill[u] += 1
        distinct = distinct.astype(np.int32)
    return PackedBatch(T, T_pad, p_ids, p_pos, cu, int((sa[:, 1] - sa[:, 0]).max()) if len(sa) else 0, img_rows, embed_ids,
                       logit_rows.astype(np.int32), labels, (bb * R + jj).astype(np.int64), B, R,
                       seg_b=sa[:, 0].copy(), seg_e=sa[:, 1].copy(), pre_b=sa[:, 2].copy(), pre_e=sa[:, 3].copy(), dep_e=sa[:, 4].copy(),
                       max_seg=int((sa[:, 1] - sa[:, 0]).max()) if len(sa) else 0, logit_dup=dup, logit_distinct=distinct)


def pack_responses(input_ids: np.ndarray, attention_mask: np.ndarray, position_ids: np.ndarray, response_length: int,
                   prompt_of_row: Sequence[int], prompt_offsets: np.ndarray, *, pad_multiple: int = 128) -> PackedBatch:
    """Responses-only packing for a log-prob pass that re-uses the prompt K/V cache of the rollout prefill (same weights): only
    the valid RESPONSE tokens are packed; row r's segment sees the cached keys [prompt_offsets[p], prompt_offsets[p+1]) of its
    prompt p = prompt_of_row[r] as prefix.  The logits of response slot 0 come from the prompt's cached last hidden state
    (`first_prompt`), all other slots from the packed row of the previous response token; log-prob rows are ordered
    [slot-0 rows..., the rest] and `logit_rows` holds the packed rows of "the rest" only."""
    ids = np.asarray(input_ids)
    mask = np.asarray(attention_mask).astype(bool)
    B, S = ids.shape
    R = response_length
    Pc = S - R
    pos = np.asarray(position_ids)
    if pos.ndim == 2:
        pos = np.repeat(pos[:, None, :], 3, axis=1)
    packed_row = -np.ones((B, S), dtype=np.int64)
    p_ids_l, p_pos_l, seg = [], [], []
    cur = 0
    for r in range(B):
        rcols = Pc + np.nonzero(mask[r, Pc:])[0]
        n = len(rcols)
        if n == 0:
            continue
        p_ids_l.append(ids[r, rcols]); p_pos_l.append(pos[r][:, rcols])
        packed_row[r, rcols] = cur + np.arange(n)
        pr = int(prompt_of_row[r])
        seg.append((cur, cur + n, int(prompt_offsets[pr]), int(prompt_offsets[pr + 1]), cur + n))
        cur += n
    T = cur
    T_pad = max(round_up(T, pad_multiple), pad_multiple)
    p_ids = np.zeros(T_pad, dtype=np.int32)
    p_pos = np.zeros((3, T_pad), dtype=np.int32)
    if T:
        p_ids[:T] = np.concatenate(p_ids_l)
        p_pos[:, :T] = np.concatenate(p_pos_l, axis=1)
    cols = np.arange(Pc, S)
    valid = mask[:, cols] & mask[:, cols - 1]
    b0 = np.nonzero(valid[:, 0])[0]                                  # rows whose slot 0 is predicted by the cached prompt state
    bb, jj = np.nonzero(valid[:, 1:])
    jj = jj + 1
    logit_rows = packed_row[bb, cols[jj] - 1]
    labels = np.concatenate([ids[b0, Pc], ids[bb, cols[jj]]]).astype(np.int64)
    out_index = np.concatenate([b0 * R, bb * R + jj]).astype(np.int64)
    embed_ids = p_ids.copy(); embed_ids[T:] = -1
    sa = np.asarray(seg, dtype=np.int32).reshape(-1, 5)
    lens_full = mask.sum(1)
    return PackedBatch(T, T_pad, p_ids, p_pos, np.concatenate([[0], np.cumsum(lens_full)]).astype(np.int32),
                       int((sa[:, 1] - sa[:, 0]).max()) if len(sa) else 0, np.zeros(0, dtype=np.int32), embed_ids,
                       logit_rows.astype(np.int32), labels, out_index, B, R,
                       seg_b=sa[:, 0].copy(), seg_e=sa[:, 1].copy(), pre_b=sa[:, 2].copy(), pre_e=sa[:, 3].copy(), dep_e=sa[:, 4].copy(),
                       max_seg=int((sa[:, 1] - sa[:, 0]).max()) if len(sa) else 0,
                       first_prompt=np.asarray([prompt_of_row[r] for r in b0], dtype=np.int32))
