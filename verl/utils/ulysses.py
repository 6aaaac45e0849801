"""Ulysses sequence parallelism: the all-to-all pair around attention (reference: verl/utils/ulysses.py:30-298; DeepSpeed-Ulysses,
arXiv 2309.14509) — same function names, argument meaning and autograd behaviour, own implementation.

A sequence-parallel group of `sp` ranks holds the packed token stream cut into `sp` equal slices (padded first:
`ulysses_pad_and_slice_inputs`).  Every row-wise operation (embedding, norms, projections, MLP, lm_head, log-prob) runs on the local slice.
Attention needs whole sequences, so just before it `gather_seq_scatter_heads` trades the cut: every rank receives ALL tokens of
heads / sp heads; `gather_heads_scatter_seq` trades it back afterwards.  Both are one all-to-all and each is the other's gradient
(`SeqAllToAll`).  `gather_outputs_and_unpad` collects the per-slice log-probs; its backward hands every rank the gradient of its own
slice, scaled by `sp` when the data-parallel gradient average also runs over the sequence-parallel ranks (`Gather`).

MI355X note: a sequence-parallel group is `sp` GPUs of one node, so the all-to-all is `sp - 1` direct xGMI transfers per rank (RCCL
all-to-all; the engine calls it through all_to_all_tensor).  On 288 GB parts no shipped script needs SP > 1; the
engine's own use is spatialthinker_amd.model.Qwen25VL(sp_group=...)."""
from __future__ import annotations

from typing import Any, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor
from torch.distributed import ProcessGroup

_SP_GROUP: Optional[ProcessGroup] = None


def set_ulysses_sequence_parallel_group(group: Optional[ProcessGroup]) -> None:
    global _SP_GROUP
    _SP_GROUP = group


def get_ulysses_sequence_parallel_group() -> Optional[ProcessGroup]:
    return _SP_GROUP


def _group(group: Optional[ProcessGroup]) -> Optional[ProcessGroup]:
    return _SP_GROUP if group is None else group


def get_ulysses_sequence_parallel_world_size(group: Optional[ProcessGroup] = None) -> int:
    g = _group(group)
    return dist.get_world_size(g) if g is not None else 1


def get_ulysses_sequence_parallel_rank(group: Optional[ProcessGroup] = None) -> int:
    g = _group(group)
    return dist.get_rank(g) if g is not None else 0


# ------------------------------------------------------------------ padding helpers
def _pad_tensor(x: Tensor, dim: int, padding_size: int) -> Tensor:
    if padding_size <= 0:
        return x
    shape = list(x.shape)
    shape[dim] = padding_size
    return torch.cat([x, x.new_zeros(shape)], dim=dim)


def _unpad_tensor(x: Tensor, dim: int, padding_size: int) -> Tensor:
    return x if padding_size <= 0 else x.narrow(dim, 0, x.size(dim) - padding_size)


def slice_input_tensor(x: Tensor, dim: int, padding: bool = True, group: Optional[ProcessGroup] = None) -> Tensor:
    """This rank's slice of `x` along `dim` (zero-padded to a multiple of the group size first when `padding`)."""
    g = _group(group)
    sp, rank = dist.get_world_size(g), dist.get_rank(g)
    if padding and x.size(dim) % sp:
        x = _pad_tensor(x, dim, sp - x.size(dim) % sp)
    per = x.size(dim) // sp
    return x.narrow(dim, rank * per, per).contiguous()


# ------------------------------------------------------------------ collectives
def all_to_all_tensor(local_input: Tensor, scatter_dim: int, gather_dim: int, group: Optional[ProcessGroup] = None, async_op: bool = False):
    """Cut `local_input` into `sp` pieces along scatter_dim, send piece j to rank j, concatenate the received pieces (rank order) along
    gather_dim.  async_op: returns a callable that waits and assembles."""
    g = _group(group)
    sp = dist.get_world_size(g)
    if local_input.size(scatter_dim) % sp:
        raise ValueError(f"all_to_all_tensor: dimension {scatter_dim} of size {local_input.size(scatter_dim)} is not divisible by the group size {sp}")
    # equal pieces -> ONE buffer of sp slabs and the single-tensor all-to-all (RCCL: one grouped send/recv; gloo has no list form at all)
    send = torch.stack([p.contiguous() for p in torch.tensor_split(local_input, sp, dim=scatter_dim)], 0)
    recv = torch.empty_like(send)
    work = None
    if dist.get_backend(g) == "gloo" and local_input.is_cuda:
        # test mode only (ranks sharing one GPU exchange over gloo): gloo's all-to-all takes host tensors
        host = torch.empty(send.shape, dtype=send.dtype)
        dist.all_to_all_single(host, send.cpu(), group=g)
        recv.copy_(host)
    else:
        work = dist.all_to_all_single(recv, send, group=g, async_op=async_op)

    def assemble():
        return torch.cat(list(recv.unbind(0)), dim=gather_dim).contiguous()
    if not async_op:
        return assemble()

    def wait():
        if work is not None:
            work.wait()
        return assemble()
    return wait


def all_gather_tensor(local_tensor: Tensor, group: Optional[ProcessGroup] = None, async_op: bool = False) -> Tensor:
    """Rank-major concatenation along dim 0."""
    g = _group(group)
    sp = dist.get_world_size(g)
    out = local_tensor.new_empty((local_tensor.shape[0] * sp,) + tuple(local_tensor.shape[1:]))
    dist.all_gather_into_tensor(out, local_tensor.contiguous(), group=g, async_op=async_op)
    return out


class SeqAllToAll(torch.autograd.Function):
    """all_to_all_tensor with its own transpose as gradient: backward scatters along what forward gathered and vice versa."""

    @staticmethod
    def forward(ctx: Any, group: ProcessGroup, local_input: Tensor, scatter_dim: int, gather_dim: int, async_op: bool = False) -> Tensor:
        ctx.group, ctx.scatter_dim, ctx.gather_dim, ctx.async_op = group, scatter_dim, gather_dim, async_op
        return all_to_all_tensor(local_input, scatter_dim, gather_dim, group, async_op)

    @staticmethod
    def backward(ctx: Any, *grad_output: Tensor) -> Tuple[None, Tensor, None, None, None]:
        g = torch.cat(grad_output[1:], dim=ctx.gather_dim).contiguous() if ctx.async_op else grad_output[0]
        return None, all_to_all_tensor(g, ctx.gather_dim, ctx.scatter_dim, ctx.group, False), None, None, None


def gather_seq_scatter_heads(x: Tensor, seq_dim: int, head_dim: int, unpadded_dim_size: int = 0, group: Optional[ProcessGroup] = None) -> Tensor:
    """[.., seq / sp, .., heads, ..] -> [.., seq, .., heads / sp, ..]; with unpadded_dim_size the sequence padding added by
    gather_heads_scatter_seq / ulysses_pad_and_slice_inputs is dropped again."""
    g = _group(group)
    if g is None:
        return x
    sp = dist.get_world_size(g)
    x = SeqAllToAll.apply(g, x, head_dim, seq_dim)
    if unpadded_dim_size and unpadded_dim_size % sp:
        x = _unpad_tensor(x, seq_dim, x.size(seq_dim) - unpadded_dim_size)
    return x


def gather_heads_scatter_seq(x: Tensor, head_dim: int, seq_dim: int, group: Optional[ProcessGroup] = None) -> Tensor:
    """[.., seq, .., heads / sp, ..] -> [.., seq / sp, .., heads, ..] (the sequence is zero-padded to a multiple of sp first)."""
    g = _group(group)
    if g is None:
        return x
    sp = dist.get_world_size(g)
    if x.size(seq_dim) % sp:
        x = _pad_tensor(x, seq_dim, sp - x.size(seq_dim) % sp)
    return SeqAllToAll.apply(g, x, seq_dim, head_dim, False)


class Gather(torch.autograd.Function):
    """all-gather along gather_dim; backward = this rank's slice of the gradient, times sp when grad_scaler (the gradient average of the
    data-parallel reduction also runs over the sp ranks, which all hold the SAME loss)."""

    @staticmethod
    def forward(ctx: Any, group: ProcessGroup, local_tensor: Tensor, gather_dim: int, grad_scaler: bool = True, async_op: bool = False) -> Tensor:
        ctx.group, ctx.gather_dim, ctx.grad_scaler = group, gather_dim, grad_scaler
        ctx.sp, ctx.rank, ctx.part = dist.get_world_size(group), dist.get_rank(group), local_tensor.size(gather_dim)
        rows = local_tensor.size(0)
        stacked = all_gather_tensor(local_tensor, group, async_op)                 # rank-major along dim 0
        return torch.cat(stacked.split(rows, dim=0), dim=gather_dim)

    @staticmethod
    def backward(ctx: Any, grad_output: Tensor) -> Any:
        if ctx.grad_scaler:
            grad_output = grad_output * ctx.sp
        return None, grad_output.split(ctx.part, dim=ctx.gather_dim)[ctx.rank].contiguous(), None, None, None


def gather_outputs_and_unpad(x: Tensor, gather_dim: int, unpad_dim: Optional[int] = None, padding_size: int = 0, grad_scaler: bool = True,
                             group: Optional[ProcessGroup] = None) -> Tensor:
    g = _group(group)
    if g is None:
        return x
    x = Gather.apply(g, x, gather_dim, grad_scaler)
    if unpad_dim is not None:
        assert isinstance(padding_size, int), "padding size is not given or is not an integer"
        x = _unpad_tensor(x, unpad_dim, padding_size)
    return x


def ulysses_pad_and_slice_inputs(input_ids_rmpad: Tensor, position_ids_rmpad: Optional[Tensor] = None, sp_size: int = 1):
    """input_ids (1, T) -> zero-padded to a multiple of sp_size and cut to this rank's slice; position_ids (.., T) padded with
    0, 1, .. pad-1 (a fresh "sequence" for the pad tokens) but NOT sliced (attention sees the whole stream).  Returns
    (ids_slice, padded position ids, pad_size)."""
    if position_ids_rmpad is not None:
        assert position_ids_rmpad.size(0) == 1 or position_ids_rmpad.dim() == 3
        assert input_ids_rmpad.size(1) == position_ids_rmpad.size(-1)
    if sp_size <= 1:
        return input_ids_rmpad, position_ids_rmpad, 0
    total = input_ids_rmpad.shape[1]
    pad = (-total) % sp_size
    if pad:
        input_ids_rmpad = torch.nn.functional.pad(input_ids_rmpad, (0, pad), value=0)
        if position_ids_rmpad is not None:
            extra = torch.arange(pad, device=position_ids_rmpad.device, dtype=position_ids_rmpad.dtype)
            extra = extra.expand(*position_ids_rmpad.shape[:-1], pad)
            position_ids_rmpad = torch.cat([position_ids_rmpad, extra], dim=-1)
    return slice_input_tensor(input_ids_rmpad, dim=1, padding=False), position_ids_rmpad, pad
