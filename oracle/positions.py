"""Oracle (test infrastructure): integer index math of the Qwen2.5-VL path — M-RoPE
position ids, vision window order, vision 2-D position ids, rollout-side position
continuation, packed-sequence boundaries, sequence-length balancing.  numpy only.
"""
from __future__ import annotations

import heapq

import numpy as np


def mrope_position_ids(input_ids, image_grid_thw, attention_mask, *, image_token_id: int,
                       vision_start_token_id: int, merge_size: int = 2) -> np.ndarray:
    """verl/models/transformers/qwen2_vl.py:36-136 (`get_rope_index`, image-only branch).

    One un-batched sequence: input_ids (S,), attention_mask (S,), image_grid_thw (n,3).
    Valid text tokens get t=h=w=running index; an image block of (t, h/m, w/m) merged tokens
    gets t=const 0.., h/w = grid coordinates, all offset by st_idx + text_len; the text after
    it resumes at max(previous)+1.  Padded positions keep the value 1 (:58).  Without images:
    cumsum(mask)-1 with masked positions set to 1 (:128-131)."""
    ids = np.asarray(input_ids, dtype=np.int64)
    mask = np.ones_like(ids) if attention_mask is None else np.asarray(attention_mask, dtype=np.int64)
    if image_grid_thw is None or len(image_grid_thw) == 0:
        pos = np.cumsum(mask) - 1
        pos[mask == 0] = 1
        return np.broadcast_to(pos, (3, ids.shape[0])).copy()
    out = np.ones((3, ids.shape[0]), dtype=np.int64)
    toks = ids[mask == 1].tolist()
    n_img = sum(1 for i, t in enumerate(toks[:-1]) if t == vision_start_token_id and toks[i + 1] == image_token_id)
    chunks, st, base = [], 0, 0          # base = position the next chunk starts at (:109)
    for k in range(n_img):
        ed = toks.index(image_token_id, st)
        t, h, w = (int(x) for x in image_grid_thw[k])
        gh, gw = h // merge_size, w // merge_size
        text_len = ed - st
        chunks.append(np.broadcast_to(np.arange(text_len) + base, (3, text_len)))
        tt = np.repeat(np.arange(t), gh * gw)            # second_per_grid_t = 0 for images (:87)
        tt = tt * 0
        hh = np.tile(np.repeat(np.arange(gh), gw), t)
        ww = np.tile(np.arange(gw), t * gh)
        img = np.stack([tt, hh, ww]) + text_len + base
        chunks.append(img)
        base = int(img.max()) + 1
        st = ed + t * gh * gw
    if st < len(toks):
        tail = len(toks) - st
        chunks.append(np.broadcast_to(np.arange(tail) + base, (3, tail)))
    out[:, mask == 1] = np.concatenate(chunks, axis=1)
    return out


def mrope_position_ids_with_video(input_ids, image_grid_thw, video_grid_thw, second_per_grid_ts, attention_mask, *, image_token_id: int,
                                  video_token_id: int, vision_start_token_id: int, merge_size: int = 2, tokens_per_second: int = 2) -> np.ndarray:
    """verl/models/transformers/qwen2_vl.py:36-136 with video blocks (:88-101, :117-118), restated as a token walk: every vision-start
    token opens a block of t * (h/m) * (w/m) tokens; an image keeps its temporal row at the block's base, a video's frame f sits at
    trunc(f * seconds_per_grid * tokens_per_second); text continues at max(previous block) + 1 on all three rows."""
    ids = np.asarray(input_ids, dtype=np.int64)
    mask = np.ones_like(ids) if attention_mask is None else np.asarray(attention_mask, dtype=np.int64)
    toks = ids[mask == 1].tolist()
    cols, nxt, i, ni, nv = [], 0, 0, 0, 0
    while i < len(toks):
        opens = toks[i] == vision_start_token_id and i + 1 < len(toks) and toks[i + 1] in (image_token_id, video_token_id)
        cols.append((nxt, nxt, nxt))
        nxt += 1
        i += 1
        if not opens:
            continue
        if toks[i] == image_token_id:
            t, h, w = (int(x) for x in image_grid_thw[ni]); ni += 1
            sec = 0.0
        else:
            t, h, w = (int(x) for x in video_grid_thw[nv])
            sec = float(second_per_grid_ts[nv]) if second_per_grid_ts is not None and len(second_per_grid_ts) else 1.0
            nv += 1
        gh, gw = h // merge_size, w // merge_size
        top = nxt
        for f in range(t):
            tt = int(np.float32(f) * np.float32(sec) * np.float32(tokens_per_second))
            for y in range(gh):
                for x in range(gw):
                    cols.append((nxt + tt, nxt + y, nxt + x))
                    top = max(top, nxt + tt, nxt + y, nxt + x)
        i += t * gh * gw
        nxt = top + 1
    out = np.ones((3, ids.shape[0]), dtype=np.int64)
    out[:, mask == 1] = np.asarray(cols, dtype=np.int64).T
    return out


def continue_position_ids(prompt_position_ids: np.ndarray, response_length: int) -> np.ndarray:
    """verl/workers/rollout/vllm_rollout_spmd.py:159-170: pos[..., P+j] = pos[..., P-1]+1+j
    on every row (all three M-RoPE rows alike), continued past EOS."""
    p = np.asarray(prompt_position_ids)
    delta = np.arange(1, response_length + 1, dtype=p.dtype)
    resp = p[..., -1:] + delta
    return np.concatenate([p, resp], axis=-1)


def packed_cu_seqlens_from_positions(temporal_positions: np.ndarray) -> np.ndarray:
    """verl/models/transformers/flash_attention_utils.py:43-58 (`prepare_fa2_from_position_ids`):
    a new packed sequence starts wherever the temporal position row equals 0."""
    pos = np.asarray(temporal_positions).reshape(-1)
    starts = np.nonzero(pos == 0)[0]
    return np.concatenate([starts, [pos.shape[0]]]).astype(np.int32)


def vision_window_index(grid_thw, *, merge_size: int = 2, window_size: int = 112, patch_size: int = 14):
    """HF vision_utils.py:124-188 (`get_vision_window_index`): order of merged (2x2) tokens so
    that each window_size-pixel window is contiguous, and cumulative window boundaries in
    PATCH units (merged count * merge_size**2), consecutive duplicates removed."""
    win = window_size // merge_size // patch_size
    unit = merge_size * merge_size
    order, cu, base = [], [0], 0
    for t, h, w in np.asarray(grid_thw).tolist():
        gh, gw = h // merge_size, w // merge_size
        idx = np.arange(t * gh * gw).reshape(t, gh, gw)
        ph, pw = win - gh % win, win - gw % win
        nh, nw = (gh + ph) // win, (gw + pw) // win
        padded = np.full((t, gh + ph, gw + pw), -100, dtype=np.int64)
        padded[:, :gh, :gw] = idx
        padded = padded.reshape(t, nh, win, nw, win).transpose(0, 1, 3, 2, 4).reshape(t, nh * nw, win, win)
        counts = (padded != -100).sum(axis=(2, 3)).reshape(-1)
        flat = padded.reshape(-1)
        order.append(flat[flat != -100] + base)
        cu.extend((np.cumsum(counts) * unit + cu[-1]).tolist())
        base += t * gh * gw
    cu = np.asarray(cu, dtype=np.int32)
    keep = np.concatenate([[True], cu[1:] != cu[:-1]])
    return np.concatenate(order).astype(np.int64), cu[keep]


def vision_position_ids(grid_thw, merge_size: int = 2) -> np.ndarray:
    """HF vision_utils.py:81-121 (`get_vision_position_ids`): (h, w) patch coordinates laid out
    merge-block-major, i.e. in the same row order as `pixel_values`."""
    out = []
    for t, h, w in np.asarray(grid_thw).tolist():
        hh, ww = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        shape = (h // merge_size, merge_size, w // merge_size, merge_size)
        hh = hh.reshape(shape).transpose(0, 2, 1, 3).reshape(-1)
        ww = ww.reshape(shape).transpose(0, 2, 1, 3).reshape(-1)
        out.append(np.tile(np.stack([hh, ww], axis=-1), (t, 1)))
    return np.concatenate(out, axis=0).astype(np.int64)


def vision_cu_seqlens(grid_thw) -> np.ndarray:
    """HF vision_utils.py:55-65: per-frame patch counts, cumulative, int32."""
    g = np.asarray(grid_thw)
    lens = np.repeat(g[:, 1] * g[:, 2], g[:, 0])
    return np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)


# --------------------------------------------------------------------------- balancing
def balanced_partitions(seqlens, k: int):
    """verl/utils/seqlen_balancing.py:97-181 — Karmarkar–Karp largest-differencing with
    equal_size=True, exact tie-breaking of the reference's heap ordering.

    A set is (sum, [(idx, len), ...]); a state is k sets kept in DEcreasing order.  Set
    ordering (:38-43): by sum, then item count, then the item list.  State heap ordering
    (:76-83): larger spread first, then larger first-set first.  Merge (:66-69): set i of the
    popped state absorbs set k-1-i of the next."""
    seqlens = list(seqlens)
    assert len(seqlens) % k == 0 and len(seqlens) >= k

    def set_key(s):
        return (s[0], len(s[1]), s[1])

    class _State:
        __slots__ = ("sets",)

        def __init__(self, sets):
            self.sets = sorted(sets, key=set_key, reverse=True)

        @property
        def spread(self):
            return self.sets[0][0] - self.sets[-1][0]

        def __lt__(self, other):
            if self.spread != other.spread:
                return self.spread > other.spread
            return set_key(self.sets[0]) > set_key(other.sets[0])

    ordered = sorted((length, i) for i, length in enumerate(seqlens))
    heap = []
    for off in range(0, len(ordered), k):
        sets = [(length, [(i, length)]) for length, i in ordered[off:off + k]]
        heapq.heappush(heap, _State(sets))
    while len(heap) > 1:
        a, b = heapq.heappop(heap), heapq.heappop(heap)
        merged = [(a.sets[i][0] + b.sets[k - 1 - i][0], a.sets[i][1] + b.sets[k - 1 - i][1]) for i in range(k)]
        heapq.heappush(heap, _State(merged))
    return [sorted(i for i, _ in s[1]) for s in heap[0].sets]
