"""Multi-GPU plumbing of the episodic inference path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The path shards naturally: support classes and query images are independent units
split across ranks in contiguous blocks (the reference uses detectron2 InferenceSampler,
sylph/data/build.py:578-592,749-763); the ONLY exchange step is the class-code gather
(MetaFCOSRunner._gather_class_code, sylph/runner/meta_fcos_runner.py:381-439, which pickles
Python dicts through all_gather_object).  Here the codes travel as one dense fp32 block per rank
([n, 260]: cls_conv 256 | cls_bias | acc_weight | class id | valid) in a single all_gather; the payload
is <= 0.9 MB (866 classes), i.e. latency bound, so one collective with padded equal-size blocks is
the right shape for the point-to-point xGMI fabric (no ring, no bucketing).
"""
from typing import List, Tuple

import torch
import torch.distributed as dist

CODE_DIM = 256
F_BIAS, F_ACC, F_CID, F_VALID, ROW = 256, 257, 258, 259, 260


def get_world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def inference_shard(n: int, rank: int = None, world: int = None) -> Tuple[int, int]:
    """Contiguous block [rank*ceil(n/W), ...) of n items (InferenceSampler sharding)."""
    rank = get_rank() if rank is None else rank
    world = get_world_size() if world is None else world
    shard = (n + world - 1) // world if n > 0 else 0
    begin = min(shard * rank, n)
    return begin, min(begin + shard, n)


def pack_codes(cls_conv: torch.Tensor, cls_bias: torch.Tensor, class_ids, acc_weight=None) -> torch.Tensor:
    """(n,256[,1,1]), (n,), ids -> (n, 260) fp32 rows."""
    n = cls_conv.shape[0]
    out = torch.zeros(n, ROW, dtype=torch.float32, device=cls_conv.device)
    out[:, :CODE_DIM] = cls_conv.reshape(n, CODE_DIM)
    out[:, F_BIAS] = cls_bias.reshape(n)
    out[:, F_ACC] = 1.0 if acc_weight is None else torch.as_tensor(acc_weight, dtype=torch.float32, device=out.device)
    out[:, F_CID] = torch.as_tensor(class_ids, dtype=torch.float32, device=out.device)
    out[:, F_VALID] = 1.0
    return out


def gather_packed_codes(local: torch.Tensor) -> torch.Tensor:
    """All ranks' (n_r, 260) rows concatenated in rank order (every rank gets the same result).
    Blocks are padded to the max n_r so a single fixed-size all_gather suffices."""
    world = get_world_size()
    if world == 1:
        return local
    dev = local.device
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    nmax = max(int(c.item()) for c in counts)
    padded = torch.zeros(max(nmax, 1), ROW, dtype=torch.float32, device=dev)
    padded[: local.shape[0]] = local
    blocks = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(blocks, padded)
    return torch.cat([b[: int(c.item())] for b, c in zip(blocks, counts)], dim=0)


def reduce_packed_codes(rows: torch.Tensor) -> torch.Tensor:
    """reduce_class_code (sylph/modeling/code_generator/utils.py:397-427) on packed rows: sum the
    (already len/total_len-weighted) chunk codes of each class id in order of first appearance,
    divide by acc_weight when |1 - acc| > 1e-6; returned rows carry acc_weight = 1."""
    if rows.shape[0] == 0:
        return rows
    cids = rows[:, F_CID].round().to(torch.int64).tolist()
    order, index = [], {}
    for i, c in enumerate(cids):
        if c not in index:
            index[c] = len(order)
            order.append(c)
    out = torch.zeros(len(order), ROW, dtype=torch.float32, device=rows.device)
    for i, c in enumerate(cids):  # fixed order -> deterministic sums
        out[index[c], : F_ACC + 1] += rows[i, : F_ACC + 1]
    for j, c in enumerate(order):
        acc = float(out[j, F_ACC])
        if abs(1.0 - acc) > 1e-6:
            out[j, : F_BIAS + 1] /= acc
        out[j, F_ACC] = 1.0
        out[j, F_CID] = float(c)
        out[j, F_VALID] = 1.0
    return out


def order_by_class_id(rows: torch.Tensor, num_classes: int) -> torch.Tensor:
    """format_class_codes_shared ordering (meta_learn_evaluation.py:71-103): row i <- class id i."""
    cids = rows[:, F_CID].round().to(torch.int64)
    assert sorted(cids.tolist()) == list(range(num_classes)), \
        f"Got {rows.shape[0]} class codes for prediction, but expect to be {num_classes}."
    out = torch.empty_like(rows)
    out[cids] = rows
    return out
