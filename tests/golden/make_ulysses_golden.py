"""Golden vectors for verl/utils/ulysses.py from the reference's own module, run HERE on two gloo ranks (the reference cannot travel).
    python tests/golden/make_ulysses_golden.py          ->  tests/golden/ulysses.npz
The reference file is loaded by path (its package name collides with this repo's `verl`).  gloo has no list-form all_to_all, which
the reference's all_to_all_tensor calls: for the run it is emulated with all_gather — rank r's output j = rank j's input r — so what the
fixture pins is the reference's own split / pad / concatenate / gradient-scale arithmetic around the collective, which is the part a
re-implementation can get wrong; the transport is torch's.  Inputs are a pure function of (seed, rank); every output is stored per rank."""
import importlib.util
import os
import sys

sys.dont_write_bytecode = True          # the reference tree is read-only: no __pycache__ next to its sources (spawned workers re-run this line)

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/verl/utils/ulysses.py"
W = 2


def inputs(rank):
    """what rank `rank` feeds (also used by tests/test_ulysses_gloo.py)"""
    g = torch.Generator().manual_seed(100 + rank)
    return dict(
        ids=torch.randint(0, 1000, (1, 37), generator=torch.Generator().manual_seed(7)),            # same on all ranks: 37 tokens -> pad 1
        pos=torch.arange(37).unsqueeze(0),
        x_seq=torch.randn(1, 19, 4, 6, generator=g, dtype=torch.float64),                            # (b, local seq, heads, d): heads % W == 0
        x_head=torch.randn(1, 38, 2, 6, generator=g, dtype=torch.float64),                           # (b, full padded seq, local heads, d)
        y=torch.randn(19, 5, generator=g, dtype=torch.float64),                                      # local rows of a padded 38-row result
        gy=torch.randn(37, 5, generator=torch.Generator().manual_seed(9), dtype=torch.float64),      # upstream gradient of the gathered result
    )


def _emulated_all_to_all(output_list, input_list, group=None, async_op=False):
    n = dist.get_world_size(group)
    stacked = torch.stack([t.contiguous() for t in input_list])
    got = [torch.empty_like(stacked) for _ in range(n)]
    dist.all_gather(got, stacked, group=group)
    r = dist.get_rank(group)
    for j in range(n):
        output_list[j].copy_(got[j][r])


def worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=W)
    spec = importlib.util.spec_from_file_location("ref_ulysses", REF)
    U = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(U)
    dist.all_to_all = _emulated_all_to_all
    U.set_ulysses_sequence_parallel_group(dist.group.WORLD)
    i = inputs(rank)
    res = {}
    ids, pos, pad = U.ulysses_pad_and_slice_inputs(i["ids"], i["pos"], sp_size=W)
    res.update(pad_ids=ids, pad_pos=pos, pad_size=torch.tensor(pad))
    res["slice_pad"] = U.slice_input_tensor(i["pos"].double(), dim=1, padding=True)
    res["gather_seq"] = U.gather_seq_scatter_heads(i["x_seq"], seq_dim=1, head_dim=2)
    res["gather_seq_unpad"] = U.gather_seq_scatter_heads(i["x_seq"], seq_dim=1, head_dim=2, unpadded_dim_size=37)
    res["gather_heads"] = U.gather_heads_scatter_seq(i["x_head"], head_dim=2, seq_dim=1)
    for scaler in (True, False):
        y = i["y"].clone().requires_grad_(True)
        full = U.gather_outputs_and_unpad(y, gather_dim=0, unpad_dim=0, padding_size=1, grad_scaler=scaler)
        (full * i["gy"]).sum().backward()
        res[f"gather_out_{int(scaler)}"] = full.detach()
        res[f"gather_out_grad_{int(scaler)}"] = y.grad
    xs = i["x_seq"].clone().requires_grad_(True)
    o = U.gather_seq_scatter_heads(xs, seq_dim=1, head_dim=2)
    (o * torch.arange(o.numel(), dtype=torch.float64).view_as(o)).sum().backward()
    res["gather_seq_grad"] = xs.grad
    np.savez(os.path.join(out_dir, f"ulysses_rank{rank}.npz"), **{k: v.numpy() for k, v in res.items()})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(worker, args=(29631, d), nprocs=W, join=True)
        out = {}
        for r in range(W):
            z = np.load(os.path.join(d, f"ulysses_rank{r}.npz"))
            out.update({f"r{r}_{k}": z[k] for k in z.files})
    np.savez_compressed(os.path.join(HERE, "ulysses.npz"), **out)
    print("wrote", os.path.join(HERE, "ulysses.npz"), len(out), "arrays")
