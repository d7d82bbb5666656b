"""Multi-GPU plumbing of the episodic inference path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The path shards naturally: support classes and query images are independent units
split across ranks in contiguous blocks (the reference uses detectron2 InferenceSampler,
sylph/data/build.py:578-592,749-763); the ONLY exchange step is the class-code gather
(MetaFCOSRunner._gather_class_code, sylph/runner/meta_fcos_runner.py:381-439, which pickles
Python dicts through all_gather_object).  Here everything a class code carries travels in ONE dense
fp32 block per rank and ONE collective (all_gather_into_tensor): no pickle, no count exchange, no
host read-back.  Row layout (ROW = 280 floats = 1120 B):

    [0, 256) cls_conv | 256 cls_bias | 257 acc_weight | 258 class id | 259 valid | 260 cls_weight_norm |
    261 has_weight_norm | 262 has_acc_weight (the record carried an "acc_weight" key) |
    263 overflow (row 0 of a rank's block: rows that rank had to drop because they did not fit the block; 0 otherwise) |
    [264, 280) class name: 16 lanes x 3 UTF-8 bytes (longer names are cut at a character boundary, with a warning)

Every rank contributes a block of the same, statically known capacity (the InferenceSampler shard size
ceil(n / world), or num_classes rows indexed by class id for the base-class path); unused rows have
valid = 0.  The payload is <= 1 MB per rank (866 classes), i.e. latency bound: one collective with
equal-size blocks is the right shape for the point-to-point xGMI fabric (no ring, no bucketing).
"""
from typing import List, Optional, Sequence, Tuple

import warnings

import numpy as np
import torch
import torch.distributed as dist

CODE_DIM = 256
F_BIAS, F_ACC, F_CID, F_VALID, F_WNORM, F_HAS_WNORM, F_HAS_ACC, F_OVERFLOW, F_NAME, NAME_FLOATS = 256, 257, 258, 259, 260, 261, 262, 263, 264, 16
ROW = F_NAME + NAME_FLOATS
NAME_BYTES = 3 * NAME_FLOATS  # 3 name bytes per fp32 lane (exact integers < 2^24)


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


def shard_capacity(n: int, world: int = None) -> int:
    """Rows every rank reserves in the gather block for n sharded items: the shard size ceil(n / world)."""
    world = get_world_size() if world is None else world
    return max((n + world - 1) // world, 1)


def _name_floats(names: Optional[Sequence[Optional[str]]], n: int) -> torch.Tensor:
    """Class names as fp32 lanes that survive ANY fp32 hop, not only bit copies: one byte per lane would waste the block, so
    each lane carries 3 name bytes as an integer < 2^24 (exactly representable; no NaN / denormal bit patterns)."""
    vals = np.zeros((n, NAME_FLOATS), dtype=np.float32)
    if names is not None:
        for i, s in enumerate(names):
            if s:
                b = str(s).encode("utf-8")
                if len(b) > NAME_BYTES:
                    # never raise here: this runs on ONE rank right before a collective, and an exception would leave the other
                    # ranks waiting in all_gather_into_tensor.  Cut at a character boundary and say so.
                    cut = b[:NAME_BYTES].decode("utf-8", "ignore").encode("utf-8")
                    warnings.warn(f"class name {s!r} is longer than {NAME_BYTES} bytes; it travels as {cut.decode('utf-8')!r}")
                    b = cut
                pad = b + b"\0" * (3 * NAME_FLOATS - len(b))
                arr = np.frombuffer(pad, dtype=np.uint8).reshape(NAME_FLOATS, 3).astype(np.uint32)
                vals[i] = (arr[:, 0] | (arr[:, 1] << 8) | (arr[:, 2] << 16)).astype(np.float32)
    return torch.from_numpy(vals)


def pack_codes(cls_conv: torch.Tensor, cls_bias: torch.Tensor, class_ids, acc_weight=None, weight_norm=None,
               names: Optional[Sequence[Optional[str]]] = None, has_acc=None) -> torch.Tensor:
    """(n,256[,1,1]), (n,), ids -> (n, ROW) fp32 rows (layout above) on the device of cls_conv."""
    n = cls_conv.shape[0]
    dev = cls_conv.device
    out = torch.zeros(n, ROW, dtype=torch.float32, device=dev)
    if n == 0:
        return out
    out[:, :CODE_DIM] = cls_conv.reshape(n, CODE_DIM)
    out[:, F_BIAS] = cls_bias.reshape(n)
    out[:, F_ACC] = 1.0 if acc_weight is None else torch.as_tensor(acc_weight, dtype=torch.float32, device=dev)
    if has_acc is None:
        has_acc = acc_weight is not None
    out[:, F_HAS_ACC] = torch.as_tensor(has_acc, dtype=torch.float32, device=dev)  # explicit flag: a weight of exactly 1.0 is still a weight
    out[:, F_CID] = torch.as_tensor(class_ids, dtype=torch.float32, device=dev)
    out[:, F_VALID] = 1.0
    if weight_norm is not None:
        out[:, F_WNORM] = torch.as_tensor(weight_norm, dtype=torch.float32, device=dev).reshape(n)
        out[:, F_HAS_WNORM] = 1.0
    if names is not None:
        out[:, F_NAME:] = _name_floats(names, n).to(dev)
    return out


def unpack_names(rows: torch.Tensor) -> List[str]:
    """The name lanes of host rows -> strings (3 bytes per fp32 lane, see _name_floats)."""
    v = rows[:, F_NAME:].contiguous().cpu().numpy().round().astype(np.uint32)
    raw = np.stack([v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF], axis=2).astype(np.uint8).reshape(rows.shape[0], -1)
    return [bytes(r).split(b"\0", 1)[0].decode("utf-8", "replace") for r in raw]


class GatherOverflow(RuntimeError):
    """Some rank held more rows than the gather block reserves.  Raised AFTER the collective, from the gathered rows, i.e. on every
    rank with the same message."""


def fit_block(local: torch.Tensor, capacity: int) -> torch.Tensor:
    """Rows of one rank cut to the block's capacity.  Never raises: this runs on ONE rank right in front of a collective, and an
    exception here would leave the other ranks blocked in it.  The number of dropped rows travels in the overflow lane of
    the block's first row; `check_overflow` turns it into the same error on every rank once the gathered rows are on the host."""
    n = int(local.shape[0])
    capacity = max(int(capacity), 1)  # a block of zero rows could not carry the overflow lane; gather_packed_codes clamps alike
    if n <= capacity:
        return local
    local = local[:capacity].clone()
    local[0, F_OVERFLOW] = float(n - capacity)
    warnings.warn(f"{n} class-code rows on this rank do not fit the gather block of {capacity}: {n - capacity} dropped, every rank will raise")
    return local


def check_overflow(host_rows: torch.Tensor, capacity: Optional[int] = None):
    """Gathered rows on the HOST -> raises GatherOverflow if any rank flagged dropped rows (identical on all ranks)."""
    lost = host_rows[:, F_OVERFLOW]
    if bool((lost != 0).any()):
        cap = int(capacity) if capacity else None
        where = [(int(i) // cap if cap else int(i), int(v)) for i, v in zip(lost.nonzero().reshape(-1).tolist(), lost[lost != 0].tolist())]
        raise GatherOverflow("class-code gather: " + ", ".join(f"{'rank' if cap else 'row'} {r} dropped {v} row(s)" for r, v in where) +
                             (f" that did not fit the per-rank block of {cap} rows" if cap else " that did not fit its block") +
                             " -- the capacity passed to _gather_class_code is smaller than a rank's row count")


def pad_block(local: torch.Tensor, capacity: int) -> torch.Tensor:
    """(n, ROW) -> (capacity, ROW), unused rows zero (valid = 0); n > capacity: see fit_block."""
    capacity = max(int(capacity), 1)
    local = fit_block(local, capacity)
    if local.shape[0] == capacity:
        return local.contiguous()
    block = torch.zeros(capacity, ROW, dtype=torch.float32, device=local.device)
    block[: local.shape[0]] = local
    return block


class CAbiCodeGather:
    """The same collective WITHOUT torch.distributed on the data path: sylph_allgather_codes of libsylph_hip.so (one in-place
    ncclAllGather on the engine's stream through the library's own RCCL communicator) -- what a non-Python host of the C ABI calls.
    The 128-byte communicator id is made by rank 0 and must reach the other ranks over some side channel: `id_bytes` (e.g. from a
    launcher), else one torch.distributed broadcast at construction.  Opt-in (install_c_abi_gather); torch.distributed (backend
    "nccl" = RCCL) stays the default."""

    def __init__(self, engine, rank: Optional[int] = None, world: Optional[int] = None, id_bytes: Optional[bytes] = None):
        import ctypes
        from ._lib import check
        self.engine, self.L = engine, engine.L
        self.rank = get_rank() if rank is None else int(rank)
        self.world = get_world_size() if world is None else int(world)
        if id_bytes is None:
            buf = ctypes.create_string_buffer(128)
            if self.rank == 0:
                check(self.L.sylph_comm_unique_id(buf), "comm_unique_id")
            box = [bytes(buf.raw)]
            if self.world > 1:
                dist.broadcast_object_list(box, src=0)  # bootstrap only: 128 bytes, once per process group
            id_bytes = box[0]
        assert len(id_bytes) == 128
        self.comm = ctypes.c_void_p(0)
        check(self.L.sylph_comm_init_rank(engine._ctx, id_bytes, self.world, self.rank, ctypes.byref(self.comm)), "comm_init_rank")

    def gather(self, local: torch.Tensor, capacity: int) -> torch.Tensor:
        """(n, ROW) device rows of this rank -> (world * capacity, ROW), rank order, unused rows zero."""
        import ctypes
        from ._lib import check
        assert local.is_cuda and local.dtype == torch.float32 and local.dim() == 2 and local.shape[1] == ROW
        local = local.contiguous()
        self.engine._stream()
        out = torch.empty(self.world * capacity, ROW, dtype=torch.float32, device=local.device)
        check(self.L.sylph_allgather_codes(self.engine._ctx, self.comm, ctypes.c_void_p(local.data_ptr()), int(local.shape[0]),
                                           int(capacity), ctypes.c_void_p(out.data_ptr())), "allgather_codes")
        return out

    def close(self):
        if self.comm:
            self.L.sylph_comm_destroy(self.comm)
            self.comm = None


_c_abi_gather: Optional[CAbiCodeGather] = None


def install_c_abi_gather(g: Optional[CAbiCodeGather]):
    """Route gather_packed_codes through the C-ABI collective (None: back to torch.distributed)."""
    global _c_abi_gather
    _c_abi_gather = g


def gather_code_blocks(block: torch.Tensor) -> torch.Tensor:
    """ONE collective: every rank's (capacity, ROW) block -> (world * capacity, ROW) in rank order on every rank."""
    if _c_abi_gather is not None and block.is_cuda:
        return _c_abi_gather.gather(block, block.shape[0])
    if not (dist.is_available() and dist.is_initialized()):
        return block
    world = get_world_size()  # a single-rank group still goes through the collective (RCCL init + call are exercised)
    out = torch.empty(world * block.shape[0], ROW, dtype=torch.float32, device=block.device)
    dist.all_gather_into_tensor(out, block.contiguous())
    return out


def gather_packed_codes(local: torch.Tensor, capacity: Optional[int] = None) -> torch.Tensor:
    """All ranks' rows in rank order, padded: (world * capacity, ROW) with valid flags (no compaction: that would need a
    host read-back; consumers select by the valid column).  capacity defaults to the local row count, which is only
    correct when every rank holds the same number of rows.  A rank with more rows than `capacity` does not raise here (the others
    would hang in the collective): see fit_block / check_overflow."""
    cap = max(int(capacity), 1) if capacity is not None else max(int(local.shape[0]), 1)  # same clamp on every rank
    if _c_abi_gather is not None and local.is_cuda:
        return _c_abi_gather.gather(fit_block(local, cap), cap)  # pads inside the call (memset of the block tail on the stream)
    return gather_code_blocks(pad_block(local, cap))


def scatter_by_class_id(rows: torch.Tensor, num_classes: int) -> torch.Tensor:
    """format_class_codes_shared ordering (meta_learn_evaluation.py:71-103) without a host sync: out[c] <- the valid row
    whose class id is c; invalid rows land in a scratch slot.  out[:, F_VALID] tells which classes arrived."""
    cid = rows[:, F_CID].round().to(torch.int64)
    ok = (rows[:, F_VALID] > 0) & (cid >= 0) & (cid < num_classes)  # an out-of-range id must not overwrite another class's code
    idx = torch.where(ok, cid, torch.full_like(cid, num_classes))
    out = torch.zeros(num_classes + 1, rows.shape[1], dtype=rows.dtype, device=rows.device)
    out.index_copy_(0, idx, rows)
    return out[:num_classes]


def order_by_class_id(rows: torch.Tensor, num_classes: int, check: bool = True) -> torch.Tensor:
    """scatter_by_class_id + (optionally, one host sync) the reference's completeness assertion."""
    out = scatter_by_class_id(rows, num_classes)
    if check:
        cid = rows[:, F_CID].round()
        bad = int(((rows[:, F_VALID] > 0) & ((cid < 0) | (cid >= num_classes))).sum().item())
        assert bad == 0, f"{bad} class code(s) carry a class id outside [0, {num_classes})"
        got = int(out[:, F_VALID].sum().item())
        assert got == num_classes, f"Got {got} class codes for prediction, but expect to be {num_classes}."
    return out


def reduce_packed_codes(rows: torch.Tensor, divide_by_acc: bool = True) -> torch.Tensor:
    """reduce_class_code (sylph/modeling/code_generator/utils.py:397-427) on HOST rows (CPU tensors: the gloo tests and the
    dict-level API); the device path is Engine.reduce_codes / sylph_reduce_codes with the same arithmetic.  Sums the
    (already len/total_len-weighted) chunk codes of each class id in row order, first-appearance class order, divides by
    acc_weight (accumulated in double, like the reference's Python floats) when |1 - acc| > 1e-6."""
    rows = rows[rows[:, F_VALID] > 0]
    if rows.shape[0] == 0:
        return rows
    cids = rows[:, F_CID].round().to(torch.int64).tolist()
    order, index = [], {}
    for c in cids:
        if c not in index:
            index[c] = len(order)
            order.append(c)
    out = torch.zeros(len(order), rows.shape[1], dtype=torch.float32, device=rows.device)
    acc = [0.0] * len(order)
    first = [-1] * len(order)
    for i, c in enumerate(cids):  # fixed order -> deterministic sums
        j = index[c]
        out[j, : F_BIAS + 1] += rows[i, : F_BIAS + 1]
        out[j, F_WNORM] += rows[i, F_WNORM]
        out[j, F_HAS_WNORM] = max(float(out[j, F_HAS_WNORM]), float(rows[i, F_HAS_WNORM]))
        acc[j] += float(rows[i, F_ACC])
        if first[j] < 0:
            first[j] = i
    for j, c in enumerate(order):
        if divide_by_acc and abs(1.0 - acc[j]) > 1e-6:
            a = torch.tensor(acc[j], dtype=torch.float32)
            out[j, : F_BIAS + 1] /= a
            out[j, F_WNORM] /= a
        out[j, F_ACC] = 1.0 if divide_by_acc else acc[j]
        out[j, F_CID] = float(c)
        out[j, F_VALID] = 1.0
        out[j, F_HAS_ACC:] = rows[first[j], F_HAS_ACC:]  # flag + name lanes of the class's first row (as sylph_reduce_codes)
    return out
