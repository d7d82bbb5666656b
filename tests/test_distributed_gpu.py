"""GPU tests of the class-code exchange: device-side segmented reduce (sylph_reduce_codes) against the reference golden,
the base-class "use all ground truths" episode against the oracle, and the RCCL path itself (backend "nccl",
world size 1: process-group init + the one all_gather_into_tensor run on the real GPU)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _chunks(g, with_wn=True):
    chunks = []
    for i in range(5):
        keys = ("cls_conv", "cls_bias", "cls_weight_norm") if with_wn else ("cls_conv", "cls_bias")
        cc = {k: torch.as_tensor(g[f"chunk{i}_{k}"]) for k in keys}
        cc["acc_weight"] = float(g[f"chunk{i}_acc_weight"])
        chunks.append({"support_set_target": int(g[f"chunk{i}_cid"]), "class_name": f"k{int(g[f'chunk{i}_cid'])}", "class_code": cc})
    return chunks


def test_reduce_codes_device_matches_reference_golden(golden_dir):
    """reduce_class_code of the reference (utils.py:397-427) incl. cls_weight_norm: class 0 weights sum to 1 (no division),
    class 1 to 0.7 (divided by the accumulated weight)."""
    from sylph_amd import distributed as D
    from sylph_amd.engine import Engine
    from sylph_amd.runner import MetaFCOSRunner, _rows_from_codes
    g = np.load(os.path.join(golden_dir, "g5_reduce_condblock.npz"))
    eng = Engine(None, dtype="f32")
    rows = _rows_from_codes(_chunks(g), eng.device)
    red = eng.reduce_codes(rows.contiguous(), 3).cpu()
    assert red[:, D.F_VALID].tolist() == [1.0, 1.0, 0.0]
    for cid in (0, 1):
        np.testing.assert_allclose(red[cid, :256].numpy(), g[f"reduced{cid}_cls_conv"].reshape(-1), atol=1e-6)
        np.testing.assert_allclose(red[cid, D.F_BIAS].item(), g[f"reduced{cid}_cls_bias"].reshape(-1)[0], atol=1e-6)
        np.testing.assert_allclose(red[cid, D.F_WNORM].item(), g[f"reduced{cid}_cls_weight_norm"].reshape(-1)[0], atol=1e-6)
    assert D.unpack_names(red[:2]) == ["k0", "k1"]
    # the runner entry with an engine takes the same device path and returns the reference's dict form
    out = MetaFCOSRunner._gather_class_code(_chunks(g), reduce=True, engine=eng)
    assert [r["support_set_target"] for r in out] == [0, 1] and "acc_weight" not in out[0]["class_code"]
    np.testing.assert_allclose(out[1]["class_code"]["cls_conv"].numpy(), g["reduced1_cls_conv"], atol=1e-6)
    np.testing.assert_allclose(out[1]["class_code"]["cls_weight_norm"].numpy(), g["reduced1_cls_weight_norm"], atol=1e-6)
    # plain accumulation (per-rank step): sums only, accumulated weight kept
    acc = eng.reduce_codes(rows.contiguous(), 2, divide_by_acc=False).cpu()
    assert abs(acc[1, D.F_ACC].item() - 0.7) < 1e-6
    np.testing.assert_allclose(acc[1, :256].numpy() / 0.7, g["reduced1_cls_conv"].reshape(-1), atol=1e-5)


def test_base_class_support_path_matches_oracle():
    """inference_on_support_set_dataset_base + reduce + replace (meta_learn_evaluation.py:118-254) driven through the
    runner with a base_support_loader: class 0 has 13 shots (chunks 10 + 3), class 1 has 4; both replace the few-shot codes."""
    from oracle import codegen as CG, episode as E
    from sylph_amd import synthetic as W
    from sylph_amd.data import SyntheticBaseSupportLoader, SyntheticQueryLoader, SyntheticSupportSetLoader
    from sylph_amd.runner import MetaFCOSRunner, create_cfg
    runner = MetaFCOSRunner()
    cfg = create_cfg(runner.get_default_cfg(), "sylph://COCO-Detection/Meta-FCOS/Meta-FCOS-finetune.yaml")
    sd = W.synthetic_state_dict(0, depth=50)
    model = runner.build_model(cfg, dtype="f32")
    model.load_state_dict(sd)
    model.eval()
    sup = SyntheticSupportSetLoader(3, 1, 96, 128, seed=3)
    base = SyntheticBaseSupportLoader([13, 4], 96, 128, chunk=10, seed=9)
    assert len(base) == 3
    qry = SyntheticQueryLoader(1, 96, 128, batch_size=1, seed=4)
    _, codes = runner._do_test_meta_learning(cfg, model, sup, qry, None, base_support_loader=base, num_classes=3)
    # oracle: weighted sum of the chunk codes per base class, then normalisation
    acc = {}
    for item in base:
        it = item[0]
        imgs = [r["image"].cpu() for r in it["support_set"]]
        boxes = torch.cat([r["instances"].gt_boxes.tensor for r in it["support_set"]])
        c = E.forward_class_code(imgs, boxes, sd)
        wgt = it["len"] / it["total_len"]
        cid = int(it["support_set_target"])
        a = acc.setdefault(cid, {"cls_conv": 0, "cls_bias": 0})
        a["cls_conv"] = a["cls_conv"] + c["cls_conv"] * wgt
        a["cls_bias"] = a["cls_bias"] + c["cls_bias"] * wgt
    recs = [{"support_set_target": torch.tensor(k), "class_name": str(k), "class_code": v} for k, v in acc.items()]
    ref = CG.forward_normalize_code(recs, sd)
    for r in ref:
        k = int(r["support_set_target"])
        np.testing.assert_allclose(codes["cls_conv"][k].reshape(-1).cpu().numpy(), r["class_code"]["cls_conv"].reshape(-1).numpy(), atol=1e-3)
        np.testing.assert_allclose(codes["cls_bias"][k].item(), r["class_code"]["cls_bias"].item(), atol=1e-3)
    assert codes["cls_conv"].shape == (3, 256, 1, 1)


_NCCL_SCRIPT = r"""
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "sylph-few-shot-detection_amd"))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from sylph_amd import distributed as D
from sylph_amd.engine import Engine
g = torch.Generator().manual_seed(3)
conv, bias = torch.randn(5, 256, generator=g).cuda(), torch.randn(5, generator=g).cuda()
local = D.pack_codes(conv, bias, [4, 0, 2, 4, 1], acc_weight=[0.5, 1.0, 1.0, 0.25, 1.0], names=["e", "a", "c", "e", "b"])
rows = D.gather_packed_codes(local, capacity=8)          # RCCL all_gather_into_tensor on the GPU
assert rows.shape == (8, D.ROW) and rows.is_cuda
torch.cuda.synchronize()
assert torch.equal(rows[:5], local) and float(rows[5:, D.F_VALID].abs().sum()) == 0.0
eng = Engine(None, dtype="f32")
red = eng.reduce_codes(rows.contiguous(), 5).cpu()
host = D.scatter_by_class_id(D.reduce_packed_codes(rows.cpu()), 5)
ok = host[:, D.F_VALID] > 0
assert ok.tolist() == [True, True, True, False, True] and torch.equal(red[:, D.F_VALID] > 0, ok)
assert torch.allclose(red[ok][:, :262], host[ok][:, :262], atol=1e-6), (red[ok][:, :262] - host[ok][:, :262]).abs().max()
assert D.unpack_names(red) == ["a", "b", "c", "", "e"]
dist.barrier()
dist.destroy_process_group()
print("NCCL_WORLD1_OK")
"""


def test_rccl_backend_world1_gather_reduce():
    """Nothing else in the suite touches RCCL (the driver has no multi-GPU node): initialise backend "nccl" with one rank
    and run the episode's single collective + the device reduce on the real GPU."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _NCCL_SCRIPT % {"root": ROOT}], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "NCCL_WORLD1_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


_CABI_SCRIPT = r"""
import sys, torch
sys.path.insert(0, "%(root)s/sylph-few-shot-detection_amd")
from sylph_amd import distributed as D
from sylph_amd.engine import Engine
eng = Engine(None, dtype="f32")
gat = D.CAbiCodeGather(eng, rank=0, world=1)           # sylph_comm_unique_id + sylph_comm_init_rank: no torch.distributed at all
g = torch.Generator().manual_seed(4)
conv, bias = torch.randn(5, 256, generator=g).cuda(), torch.randn(5, generator=g).cuda()
local = D.pack_codes(conv, bias, [4, 0, 2, 4, 1], acc_weight=[0.5, 1.0, 1.0, 0.25, 1.0], names=["e", "a", "c", "e", "b"])
rows = gat.gather(local, 8)                              # sylph_allgather_codes: ONE in-place ncclAllGather on the engine's stream
torch.cuda.synchronize()
assert rows.shape == (8, D.ROW) and torch.equal(rows[:5], local) and float(rows[5:].abs().sum()) == 0.0
full = gat.gather(local, 5)                              # n_local == capacity: no padding memset
torch.cuda.synchronize()
assert torch.equal(full, local)
D.install_c_abi_gather(gat)                              # ... and as the transport of the reference-shaped API
rows2 = D.gather_packed_codes(local, capacity=8)
torch.cuda.synchronize()
assert torch.equal(rows2, rows)
D.install_c_abi_gather(None)
try:
    gat.gather(local, 3)
    raise SystemExit("an over-full block must be refused")
except RuntimeError as e:
    assert "do not fit" in str(e), e
red = eng.reduce_codes(rows.contiguous(), 5).cpu()
assert D.unpack_names(red) == ["a", "b", "c", "", "e"]
gat.close()
print("CABI_GATHER_OK")
"""


def test_c_abi_allgather_codes_world1():
    """the episode's one collective is reachable WITHOUT torch: sylph_comm_unique_id / sylph_comm_init_rank /
    sylph_allgather_codes (RCCL resolved with dlopen inside libsylph_hip.so).  World 1 on the one GPU of the test box: the id
    hand-off, the communicator, the in-place ncclAllGather, padding and the capacity check are exercised; the N > 1 behaviour is
    ncclAllGather's own."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _CABI_SCRIPT % {"root": ROOT}], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "CABI_GATHER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def _run_bench(nranks, tmp_path, tag, extra=()):
    """bench.py under torch.distributed.run with `nranks` ranks sharing the ONE GPU of the test box over gloo
    (SYLPH_BENCH_BACKEND=gloo SYLPH_BENCH_ONE_DEVICE=1): the multi-rank control flow of the script -- class / query shards incl.
    empty ones, the single code collective, barrier + max-over-ranks timing, one JSON line from rank 0."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    codes = str(tmp_path / f"codes_{tag}.pt")
    args = ["bench.py", "--gpus", str(nranks), "--steps", "2", "--warmup", "1", "--batch", "2", "--height", "128", "--width", "160", "--dtype", "f32",
            "--no-sweep", "--no-parity", "--no-cpu-baseline", "--dump-codes", codes, *extra]
    env = dict(os.environ, SYLPH_BENCH_BACKEND="gloo", SYLPH_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if nranks <= 2:  # 2 ranks: plain `python bench.py --gpus 2` -- the script launches its own ranks when WORLD_SIZE is unset
        cmd = [sys.executable] + args
    else:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + args
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line from rank 0, got {len(lines)}: {res.stdout[-2000:]}"
    return json.loads(lines[0]), torch.load(codes)


def test_bench_multi_rank_control_flow_on_one_device(tmp_path):
    """2 ranks (3 + 2 classes) and 8 ranks (5 classes -> three empty class shards) against the world-1 run: one JSON
    line, n_gpus, whole-job value, and the gathered + normalised class codes equal to the single-rank episode (fp32 mode: per-class
    arithmetic does not depend on which rank computed it)."""
    one, c1 = _run_bench(1, tmp_path, "w1")
    assert one["n_gpus"] == 1 and one["config"]["ways"] == 5 and c1["valid"].tolist() == [1.0] * 5
    for n in (2, 8):
        out, cn = _run_bench(n, tmp_path, f"w{n}")
        assert out["n_gpus"] == n and out["steps"] == 2 and out["scaling"] == "weak" and out["value"] > 0
        assert abs(out["value"] - n * out["images_per_sec_per_gpu"]) <= 0.006 * n + 0.01  # both are rounded to 2 decimals
        assert out["episode_setup"]["code_gather_is_collective"] is True
        assert cn["valid"].tolist() == [1.0] * 5
        assert torch.equal(cn["codes"], c1["codes"]), f"class codes of the {n}-rank episode differ from the single-rank episode"


_EPISODE_GATHER_SCRIPT = r"""
import os, sys
root = %(root)r
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "sylph-few-shot-detection_amd"))
import torch, torch.distributed as dist
world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
torch.cuda.set_device(0)                      # every rank on the ONE GPU of the test box
if world > 1:
    dist.init_process_group("gloo")
from sylph_amd import distributed as D, synthetic as W
from sylph_amd.data import SyntheticSupportSetLoader
from sylph_amd.evaluation import format_class_codes_shared, inference_normalization, inference_on_support_set_dataset
from sylph_amd.runner import MetaFCOSRunner, MetaFCOSROIEncoderRunner, create_cfg
kind, ways, shots, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
if kind == "c4":     # BASELINE configs[3]: R-101-FPN LVISv1 Meta-FCOS, 866-way 5-shot, 8 ranks
    runner = MetaFCOSRunner()
    cfg = create_cfg(runner.get_default_cfg(), "sylph://LVISv1-Detection/Meta-FCOS/Meta-FCOS-finetune.yaml",
                     ["MODEL.RESNETS.DEPTH", 101, "MODEL.META_LEARN.EVAL_SHOT", shots])
    sd = W.synthetic_state_dict(0, depth=101, num_classes=866)
else:                # BASELINE configs[4]: ROIEncoder code generator, LVIS rare 337-way 5-shot, 4 ranks
    runner = MetaFCOSROIEncoderRunner()
    cfg = create_cfg(runner.get_default_cfg(), "sylph://LVISv1-Detection/Meta-FCOS/Meta-FCOS-ROI-Encoder-finetune.yaml",
                     ["MODEL.META_LEARN.EVAL_SHOT", shots])
    sd = {}
    sd.update(W.backbone_state_dict(0, depth=50)); sd.update(W.head_state_dict(1, num_classes=60)); sd.update(W.roi_encoder_state_dict(seed=4))
model = runner.build_model(cfg, dtype="f32")
model.load_state_dict(sd)
model.eval()
sup = SyntheticSupportSetLoader(ways, shots, 64, 96, seed=13)      # sharded in contiguous blocks (InferenceSampler rule)
sub = inference_on_support_set_dataset(model, sup, output_dir=None)
cap = D.shard_capacity(ways)
assert len(sub) == len(sup) <= cap
codes = runner._gather_class_code(sub, capacity=cap)                # THE collective: [cap][280] block per rank
assert len(codes) == ways and [int(c["support_set_target"]) for c in codes] == list(range(ways))
assert [c["class_name"] for c in codes] == [f"class_{i}" for i in range(ways)]          # names travel in the block
assert not any("acc_weight" in c["class_code"] for c in codes)                        # HAS_ACC lane: few-shot records carry none
if kind == "c4":
    codes = inference_normalization(model, codes)
fm = format_class_codes_shared(codes, device="cpu")
if rank == 0:
    torch.save({"cls_conv": fm["cls_conv"].cpu(), "cls_bias": fm["cls_bias"].cpu(), "cap": cap, "world": world}, out)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
print("EPISODE_GATHER_OK")
"""


@pytest.mark.parametrize("kind,ways,shots,ranks", [("c4", 866, 5, 8), ("c5", 337, 5, 4)], ids=["C4_866way_8ranks", "C5_337way_4ranks"])
def test_runner_code_gather_at_real_sizes_on_one_device(tmp_path, kind, ways, shots, ranks):
    """the 8-rank half of BASELINE configs[3] (R-101 LVIS yaml, 866 classes x 5 shots -> 109-row blocks, the last
    rank's block partly empty) and the 4-rank half of configs[4] (ROIEncoder yaml, 337 classes -> 85-row blocks) through
    MetaFCOSRunner: support loop on every rank's class shard -> ONE all_gather_into_tensor of [capacity][280] blocks (names, HAS_ACC
    lane) -> normalise -> format.  All ranks share the test box's one GPU over gloo; the formatted class codes must be bit-identical
    to the single-rank episode (fp32: a class's arithmetic does not depend on the rank or batch it was computed in)."""
    import socket
    script = str(tmp_path / "episode_gather.py")
    with open(script, "w") as f:
        f.write(_EPISODE_GATHER_SCRIPT % {"root": ROOT})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = {}
    for n in (1, ranks):
        out = str(tmp_path / f"codes_w{n}.pt")
        if n == 1:
            cmd = [sys.executable, script, kind, str(ways), str(shots), out]
        else:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                   "--master-port", str(port), script, kind, str(ways), str(shots), out]
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0 and "EPISODE_GATHER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        outs[n] = torch.load(out)
    one, many = outs[1], outs[ranks]
    assert many["world"] == ranks and many["cap"] == -(-ways // ranks) and many["cap"] * ranks >= ways
    assert tuple(many["cls_conv"].shape) == (ways, 256, 1, 1) and tuple(many["cls_bias"].shape) == (ways,)
    assert torch.isfinite(many["cls_conv"]).all() and float(many["cls_conv"].abs().sum()) > 0
    assert torch.equal(many["cls_conv"], one["cls_conv"]) and torch.equal(many["cls_bias"], one["cls_bias"])
