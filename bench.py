#!/usr/bin/env python
"""bench.py -- query images/sec of the Sylph MetaOneStageDetector inference path on MI355X.

Workload = BASELINE.json configs[1]: R-50-FPN, 5-way 5-shot, 800x1333 synthetic queries, bf16.
One "step" = one pass of the hot path over one batch of B query images already resident in HBM:
normalise+pad -> ResNet-50-FPN -> FCOS towers + class-conditional classifier -> decode + NMS +
postprocess (device outputs; one 4*(B+1)-byte count read-back per step).  The episode set-up (support
images -> class codes -> RCCL all-gather -> normalise) runs once before the timed region and is
reported separately (it is amortised over all queries of an episode).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (the MFMA
implicit-GEMM conv kernel, timed with HIP events on its launch stream inside the timed region) and
`cpu_baseline` (the CPU oracle = fp32 torch restatement of the reference, timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "sylph-few-shot-detection_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# algorithmic work per 800x1333 (padded 800x1344) query image, R-50-FPN, N = 5
GFLOP_PER_IMAGE_TOTAL = 411.45
GFLOP_PER_IMAGE_MFMA_CONV = GFLOP_PER_IMAGE_TOTAL  # every conv of the path (stem included) runs on the MFMA kernel
PEAK_BF16_TFLOPS = 2500.0            # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def make_cfg():
    from sylph_amd.config import get_default_cfg
    cfg = get_default_cfg()
    cfg.MODEL.META_LEARN.EPISODIC_LEARNING = True
    cg = cfg.MODEL.META_LEARN.CODE_GENERATOR
    cg.CONV_L2_NORM = True
    cg.TOWER_LAYERS = [["GN", "ReLU"], ["GN", "ReLU"]]
    cg.CLS_LAYER = ["", "", 1]
    cg.BIAS_LAYER = ["", "", 1]
    cfg.MODEL.META_LEARN.EVAL_SHOT = 5
    cfg.MODEL.META_LEARN.CLASS = 5
    return cfg


def dev_images(n, h, w, seed, device):
    g = torch.Generator(device=device).manual_seed(seed)
    return [torch.randint(0, 256, (3, h, w), generator=g, device=device).float() for _ in range(n)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=192,
                    help="query images per GPU per step.  Round 6: 192 -- the tower launches are 92 tiles per image, so 192 images are exactly 69 "
                         "rounds of the 256 CUs, and every launch's ramp / tail is amortised over 1.6 x the work: on one box, alternating, "
                         "2 234-2 235 img/s at 120, 2 251-2 253 at 176, 2 255-2 267 at 192, 2 254-2 255 at 208 (profiles/r6_batch_sweep.txt); "
                         "from 225 images on conv_igemm refuses res2-sized inputs (31-bit element offsets): a clean error, not a slow path")
    ap.add_argument("--ways", type=int, default=5)
    ap.add_argument("--shots", type=int, default=5)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--code-scale", type=float, default=3.0,
                    help="multiplier on the normalised class codes so the random-weight detector fires (SURVEY 8d)")
    ap.add_argument("--inflight", type=int, default=2, help="query steps enqueued before the oldest one's counts are read back")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket conv launches with HIP events (roofline fields become 0)")
    ap.add_argument("--cpu-images", type=int, default=3)
    ap.add_argument("--no-sweep", action="store_true", help="skip the untimed batch-size sweep / fp32 legs")
    ap.add_argument("--no-parity", action="store_true", help="skip the bf16-vs-oracle agreement leg")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not collect roofline.traffic in this run (two rocprofv3 --pmc passes of a short child run of this script); "
                         "the committed profiles/r6_pmc_hbm_traffic.json is used instead if its source fingerprint matches")
    ap.add_argument("--dump-codes", default=None, help="rank 0 writes the gathered, normalised class codes (N x 257) to this .pt file (tests)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: launch the N ranks ourselves (one process per GPU, rendezvous on 127.0.0.1)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.run(cmd, env=env).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # SYLPH_BENCH_BACKEND=gloo + SYLPH_BENCH_ONE_DEVICE=1: N ranks on ONE GPU, to exercise the multi-rank control flow of this script
        # (shards, the code gather, the barrier / max-over-ranks timing) on a single-GPU box; the real run is one rank per GPU on RCCL
        if os.environ.get("SYLPH_BENCH_ONE_DEVICE"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        backend = os.environ.get("SYLPH_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from sylph_amd import synthetic as W  # seeded synthetic weights / inputs (pure data generation)
    from sylph_amd import distributed as D
    from sylph_amd.engine import Engine

    cfg = make_cfg()
    sd = W.synthetic_state_dict(0, depth=50)
    eng = Engine(cfg, dtype=args.dtype, device=local_rank)
    eng.load_state_dict(sd)
    H, Wd, B, N, S = args.height, args.width, args.batch, args.ways, args.shots

    # ---- episode set-up: this rank's classes -> codes -> all-gather -> normalise (untimed) ----------
    # The classes of this rank share ONE backbone / code-generator batch (sylph_codegen_classes); the gather + ordering is
    # timed on its second call (the first pays one-off torch kernel / RCCL initialisation, reported separately).
    c0, c1 = D.inference_shard(N, rank, world)

    def support_codes():
        if c1 <= c0:
            return torch.zeros(0, D.ROW, device=device)
        sup = [im for c in range(c0, c1) for im in dev_images(S, H, Wd, 1000 + c, device)]
        boxes = torch.cat([W.synthetic_boxes(S, H, Wd, seed=2000 + c) for c in range(c0, c1)])
        eng.preprocess(sup)
        eng.backbone()
        lc = eng.codegen_classes(boxes, S)
        return D.pack_codes(lc[:, :256], lc[:, 256], list(range(c0, c1)))

    def gather(packed):
        return D.scatter_by_class_id(D.gather_packed_codes(packed, D.shard_capacity(N, world)), N)  # ONE collective, no host sync

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    packed = support_codes()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    rows = gather(packed)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows = gather(packed)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    codes = eng.normalize_codes(rows[:, :257].contiguous())
    cls_conv = (codes[:, :256] * args.code_scale).reshape(N, 256, 1, 1).contiguous()
    cls_bias = codes[:, 256].contiguous()
    torch.cuda.synchronize()
    if args.dump_codes and rank == 0:
        torch.save({"codes": codes.cpu(), "valid": rows[:, D.F_VALID].cpu()}, args.dump_codes)
    setup = {"codegen_s_first_call": t1 - t0, "code_gather_and_order_s_first_call": t2 - t1, "code_gather_and_order_s": t3 - t2,
             "code_gather_is_collective": world > 1, "support_images": (c1 - c0) * S}

    # ---- query steps -----------------------------------------------------------------------------
    queries = dev_images(B, H, Wd, 7 + rank, device)

    # One step = one batch through preprocess -> backbone+FPN -> head -> decode+NMS.  With --inflight 2 (default) the
    # host enqueues step k+1 before it reads back the detection counts of step k (same stream, same engine: the
    # device work is unchanged and strictly ordered; only the host-side readback gap is hidden).
    def launch():
        eng.preprocess(queries)
        eng.backbone()
        eng.head(cls_conv, cls_bias)
        return eng.decode_launch()

    def run_steps(n):
        pending, out = [], None
        for _ in range(n):
            pending.append(launch())
            if len(pending) >= args.inflight:
                out = eng.decode_fetch(pending.pop(0))
        while pending:
            out = eng.decode_fetch(pending.pop(0))
        return out

    dets = run_steps(args.warmup)
    eng.profile_enable(not args.no_kernel_events)
    eng.profile_read()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    dets = run_steps(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    prof = eng.profile_read()
    # Split of the conv time into backbone+FPN and head (BASELINE's second metric: "backbone MFMA %peak"): three
    # extra UNTIMED, unpipelined steps with a sync between the two stages (the split needs a readback per stage).
    split = {"backbone_ms": 0.0, "backbone_flops": 0.0, "head_ms": 0.0, "head_flops": 0.0}
    if not args.no_kernel_events:
        for _ in range(3):
            eng.preprocess(queries)
            eng.backbone()
            torch.cuda.synchronize()
            p1 = eng.profile_read()
            eng.head(cls_conv, cls_bias)
            torch.cuda.synchronize()
            p2 = eng.profile_read()
            split["backbone_ms"] += p1["conv_ms"]; split["backbone_flops"] += p1["conv_flops"]
            split["head_ms"] += p2["conv_ms"]; split["head_flops"] += p2["conv_flops"]
    eng.profile_enable(False)
    # ---- untimed extra legs (rank 0, one GPU): batch-size sweep in the reference protocol, fp32 mode, bf16 agreement ----
    sweep, fp32_img_s, parity_mode_img_s, parity, host_u8_img_s, large_batch = None, None, None, None, None, None
    if rank == 0 and world == 1 and not args.no_sweep:
        sweep = {}
        for b in (1, 8, 16, 64, 96, 120):
            if b == B:
                continue
            qs = queries[:b] if b <= B else dev_images(b, H, Wd, 7, device)

            def step_b():
                eng.preprocess(qs); eng.backbone(); eng.head(cls_conv, cls_bias)
                return eng.decode()
            for _ in range(5):  # the reference excludes 5 warm-up iterations (meta_learn_evaluation.py:392-417)
                step_b()
            torch.cuda.synchronize()
            n = 40 if b == 1 else (24 if b <= 16 else 6)  # small batches: enough steps that rounds can be compared (a step is 1.6 - 9 ms there)
            ts = time.perf_counter()
            for _ in range(n):
                step_b()  # synchronous: decode() ends on the count read-back, as the reference loop ends on cuda.synchronize
            torch.cuda.synchronize()
            sweep[f"B{b}"] = round(b * n / (time.perf_counter() - ts), 1)
        # other batch sizes in the HEADLINE protocol (steps in flight), so that the figures compare with `value`: 120 = rounds 5-6's headline
        # batch (like-for-like with BENCH_r05), 160 = past the former 124-image limit of the 32-bit whole-tensor offsets
        large_batch = {"unit": "images/s", "protocol": f"{args.inflight} steps in flight, 8 timed steps after 3 warm-up steps"}
        for b in (120, 160):
            if b == B:
                continue
            qs = queries[:b] if b <= B else dev_images(b, H, Wd, 7, device)

            def launch_b():
                eng.preprocess(qs); eng.backbone(); eng.head(cls_conv, cls_bias)
                return eng.decode_launch()

            def run_b(n):
                pend = []
                for _ in range(n):
                    pend.append(launch_b())
                    if len(pend) >= args.inflight:
                        eng.decode_fetch(pend.pop(0))
                while pend:
                    eng.decode_fetch(pend.pop(0))
            run_b(3)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            run_b(8)
            torch.cuda.synchronize()
            large_batch[f"B{b}"] = round(b * 8 / (time.perf_counter() - ts), 1)
            del qs
        # input pipeline from HOST memory (the serving shape of SylphPredictor): B uint8 HWC 480x640 camera frames in pinned
        # memory -> async H2D -> ONE kernel: PIL-exact BILINEAR resize to 800x1067 + normalise + pad -> the same step.
        # PCIe-inclusive; never the headline value (inputs of the timed region are resident in HBM).
        g8 = torch.Generator().manual_seed(5)
        frames = [torch.randint(0, 256, (480, 640, 3), dtype=torch.uint8, generator=g8).pin_memory() for _ in range(B)]
        sizes = [(800, 1067)] * B

        def step_u8():
            eng.preprocess_u8(frames, sizes); eng.backbone(); eng.head(cls_conv, cls_bias)
            return eng.decode_launch()
        pend = [step_u8() for _ in range(2)]
        for h_ in pend:
            eng.decode_fetch(h_)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        pend = []
        for _ in range(6):
            pend.append(step_u8())
            if len(pend) >= args.inflight:
                eng.decode_fetch(pend.pop(0))
        while pend:
            eng.decode_fetch(pend.pop(0))
        torch.cuda.synchronize()
        host_u8_img_s = round(B * 6 / (time.perf_counter() - ts), 1)
        # the two modes with <= 1e-3 parity against the oracle: "f32" = exact-fp32 MFMA (v_mfma_f32_32x32x2_f32); "f32s" = the same fp32
        # storage with every conv product as three bf16 MFMAs on operands split into bf16 hi + lo parts (conv_igemm.hip MmaSplit)
        def mode_leg(mode, b, n):
            e = Engine(cfg, dtype=mode, device=local_rank)
            e.load_state_dict(sd)
            qs = queries[:b] if b <= B else dev_images(b, H, Wd, 7, device)
            for _ in range(2):
                e.preprocess(qs); e.backbone(); e.head(cls_conv, cls_bias); e.decode()
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(n):
                e.preprocess(qs); e.backbone(); e.head(cls_conv, cls_bias); e.decode()
            torch.cuda.synchronize()
            r = round(b * n / (time.perf_counter() - ts), 1)
            e.close()
            return r
        fp32_img_s = mode_leg("f32", 64, 3)
        parity_mode_img_s = mode_leg("f32s", 64, 4)  # 64 per step: 522 / 560 / 571 / 580 / 585 img/s at 8 / 16 / 32 / 48 / 64 (one box)
    support_leg = None
    if rank == 0 and world == 1 and not args.no_sweep:
        # steady-state SUPPORT path: classes x shots support images of 800x1333 per batch through preprocess ->
        # ResNet-FPN -> ROIAlign -> code-generator tower -> per-class codes (sylph_codegen_classes), after one warm-up batch
        support_leg = {"unit": "support images/s", "image": [H, Wd], "note": "batched classes share the launches; codes per class as in the one-class-per-call path"}
        for shots, ncls in ((5, 12), (10, 6)):
            sup = dev_images(shots * ncls, H, Wd, 4000 + shots, device)
            bxs = torch.cat([W.synthetic_boxes(shots, H, Wd, seed=5000 + c) for c in range(ncls)])

            def sup_step():
                eng.preprocess(sup); eng.backbone()
                return eng.codegen_classes(bxs, shots)
            sup_step()
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(5):
                out_codes = sup_step()
            torch.cuda.synchronize()
            support_leg[f"S{shots}_batch{shots * ncls}"] = round(5 * shots * ncls / (time.perf_counter() - ts), 1)
        # the reference protocol: one class (S images) per call
        sup1 = dev_images(5, H, Wd, 4100, device)
        bx1 = W.synthetic_boxes(5, H, Wd, seed=5100)
        for _ in range(2):
            eng.preprocess(sup1); eng.backbone(); eng.codegen(bx1)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(10):
            eng.preprocess(sup1); eng.backbone(); eng.codegen(bx1)
        torch.cuda.synchronize()
        support_leg["S5_one_class_per_call"] = round(50 / (time.perf_counter() - ts), 1)
    many_way = None
    if rank == 0 and world == 1 and not args.no_sweep:
        # many-way episodes on the same backbone (BASELINE configs[2] / [3] head loads: 20 and 866 classes, 16 query images per step,
        # synchronous steps): the class-conditional conv + score scan are one kernel there (logits_scan_kernel), top-k by histogram
        many_way = {"unit": "images/s", "batch": 16, "code_scale": 1.5,
                    "note": "R-50, synchronous steps; 866 classes at code scale 1.5 = ~5 % of the 14.5 M P3 scores above the threshold"}
        q16 = queries[:16] if B >= 16 else dev_images(16, H, Wd, 7, device)
        cfg_m = make_cfg()
        cfg_m.MODEL.FCOS.POST_NMS_TOPK_TEST = 300  # the LVIS setting
        eng_m = Engine(cfg_m, dtype=args.dtype, device=local_rank)
        eng_m.load_state_dict(sd)
        for nway in (20, 866):
            cm = W.synthetic_codes(nway, seed=3, scale=1.5)
            cwm, cbm = cm["cls_conv"].to(device), cm["cls_bias"].to(device)

            def step_m():
                eng_m.preprocess(q16); eng_m.backbone(); eng_m.head(cwm, cbm)
                return eng_m.decode()
            for _ in range(3):
                step_m()
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(8):
                step_m()
            torch.cuda.synchronize()
            many_way[f"N{nway}"] = round(16 * 8 / (time.perf_counter() - ts), 1)
        eng_m.close()
    config_legs = None
    if rank == 0 and world == 1 and not args.no_sweep:
        eng.close()  # the headline engine's plans (about 10 GB) are not needed any more
        config_legs = baseline_config_legs(args, device, local_rank)
    if rank == 0 and world == 1 and not args.no_parity:
        parity = parity_bf16(sd, queries[:2], cls_conv, cls_bias, dets[:2])
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ndet = [int(d["scores"].numel()) for d in dets]

    if rank == 0:
        total_images = world * B * args.steps
        value = total_images / elapsed
        conv_s = prof["conv_ms"] / 1e3
        launches = max(prof["conv_launches"], 1)
        images_timed = B * args.steps
        alg_flops = GFLOP_PER_IMAGE_MFMA_CONV * 1e9 * images_timed  # all conv launches of the region, this rank
        achieved = alg_flops / conv_s / 1e12 if conv_s > 0 else 0.0
        if not args.no_live_pmc and world == 1 and args.dtype == "bf16":
            collect_live_pmc(B)  # two short rocprofv3 --pmc passes of this command, after the timed region (N = 1 only)
        traffic, traffic_src = pmc_traffic_per_launch(B, launches // max(args.steps, 1))
        roofline = {
            "kernel": "conv_hpipe_kernel + conv_igemm_kernel + bottleneck64[p]_kernel + stem_pool_kernel + gn_logits / gn_taps: the MFMA conv launches (and the fused passes that replace convs) of the timed region",
            "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3,
            "unit": "TFLOP/s", "frac": round(achieved / (PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3), 4),
            "traffic": traffic, "traffic_source": traffic_src,
            "launches": launches, "avg_launch_us": round(conv_s / launches * 1e6, 2),
            "algorithmic_gflop_per_launch": round(alg_flops / launches / 1e9, 3),
            "gflop_counted_by_library_per_image": round(prof["conv_flops"] / images_timed / 1e9, 2),
            "conv_share_of_step_time": round(conv_s / elapsed, 3),
        }
        peakv = PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3
        kern = prof.get("kernels", {})
        if kern:
            hbm = pmc_per_kernel_bytes(B)
            per = []
            for name, k in sorted(kern.items(), key=lambda kv: -kv[1]["ms"]):
                tf = k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0
                e = {"kernel": name, "ms_per_step": round(k["ms"] / args.steps, 3), "launches_per_step": k["launches"] // max(args.steps, 1),
                     "gflop_per_step": round(k["flops"] / args.steps / 1e9, 1), "achieved_tflops": round(tf, 1), "frac": round(tf / peakv, 4)}
                hb = hbm.get(name.split("+")[0])
                if hb:
                    e["hbm_bytes_per_step"] = int(hb)
                    e["hbm_frac"] = round(hb / (k["ms"] / args.steps * 1e-3) / 8e12, 4)
                per.append(e)
            roofline["per_kernel"] = per[:8]
            dom = per[0]
            roofline["dominant"] = {"kernel": dom["kernel"], "frac": dom["frac"], "achieved": dom["achieved_tflops"], "bound": "mfma",
                                    "share_of_conv_time": round(kern[dom["kernel"]]["ms"] / prof["conv_ms"], 3),
                                    "note": "per-kernel: algorithmic FLOPs of its launches / their summed HIP-event durations in the timed region"}
        if split["backbone_ms"] > 0 and split["head_ms"] > 0:
            peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3
            bb = split["backbone_flops"] / (split["backbone_ms"] * 1e-3) / 1e12
            hd = split["head_flops"] / (split["head_ms"] * 1e-3) / 1e12
            roofline["backbone_fpn"] = {"achieved": round(bb, 2), "frac": round(bb / peak, 4), "gflop_per_image": round(split["backbone_flops"] / (3 * B) / 1e9, 2)}
            roofline["head"] = {"achieved": round(hd, 2), "frac": round(hd / peak, 4), "gflop_per_image": round(split["head_flops"] / (3 * B) / 1e9, 2)}
        out = {
            "metric": "query images/sec, R50-FPN 5-way 5-shot 800x1333 (whole job)", "value": round(value, 2),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: R-50-FPN 5-way 5-shot, 800x1333 synthetic queries",
                       "batch_per_gpu": B, "steps_in_flight": args.inflight, "ways": N, "shots": S, "image": [H, Wd], "code_scale": args.code_scale,
                       "parallelism": f"dp{world} (queries sharded, codes all-gathered once per episode)",
                       "detections_last_step": {"images": len(ndet), "total": sum(ndet), "min": min(ndet), "max": max(ndet)}},
            "images_per_sec_per_gpu": round(value / world, 2),
            "tflops_sustained_per_gpu": round(GFLOP_PER_IMAGE_TOTAL * value / world / 1e3, 1),
            "episode_setup": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in setup.items()},
            "roofline": roofline,
        }
        if sweep is not None:
            sweep[f"B{B}"] = round(value, 1)
            out["sweep"] = {"unit": "images/s", "protocol": "synchronous steps (decode read-back per step), 5 warm-up steps; the headline value "
                            f"keeps {args.inflight} steps in flight", **sweep}
            out["large_batch"] = large_batch
            out["fp32_img_s"] = fp32_img_s
            out["parity_mode_img_s"] = parity_mode_img_s
            out["parity_mode"] = {"dtype": "f32s", "img_s": parity_mode_img_s, "batch": 64, "north_star_target_img_s": 300,
                                  "note": "fp32 storage, conv products as three bf16 MFMAs on bf16 hi + lo operand parts; head outputs <= 1e-3 of the "
                                          "fp32 CPU oracle and the oracle's NMS indices at 800x1333 (tests/test_split_mode_gpu.py); the exact-fp32-MFMA "
                                          "mode is fp32_img_s"}
            out["input_pipeline"] = {"from_host_u8_img_s": host_u8_img_s, "frames": "480x640x3 uint8, pinned host memory",
                                     "resized_to": [800, 1067], "note": "PCIe-inclusive: async H2D + fused PIL-exact resize/normalise/pad kernel + "
                                     "the same step; never the headline value"}
        if support_leg is not None:
            out["support_path"] = support_leg
        if many_way is not None:
            out["many_way"] = many_way
        if config_legs is not None:
            out.update(config_legs)
        if parity is not None:
            out["parity_bf16"] = parity
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, queries, cls_conv, cls_bias, args.cpu_images)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def baseline_config_legs(args, device, local_rank):
    """Untimed extras (same JSON line): the OTHER BASELINE.json configurations on one GPU, each with its own work count --
      c2_r50_20way_10shot : configs[2]  R-50-FPN, COCO 20 novel classes, 10-shot, batch 16 queries of 800x1333
      c4_r101_866way      : configs[3]  R-101-FPN, LVIS freq + common = 866 classes, 5-shot (4 330 support images per episode),
                                         192 queries of 800x1333 per step (one rank's share of the 8-GPU job; the code all-gather is not in it; rounds 4-5: 64, early round 6: 120 per step)
      c5_roi_encoder_337way: configs[4] ROI-Encoder code generator + CondConvBlock head, LVIS rare = 337 classes, 5-shot, 800x1200 queries
    Per leg: img_s (query steps, two in flight, codes resident), support_img_s (steady-state support batches of 12 classes through
    backbone -> code generator), gflop_per_image = the library's own 2 M N K count over the conv launches of a query step,
    roofline_frac = that work / the summed HIP-event time of those launches / the dense bf16 peak.  Synthetic codes at scale 1.5
    (random-weight generators give no usable score distribution): ~5 % of the scores pass the 0.05 threshold."""
    from sylph_amd import synthetic as W
    from sylph_amd.engine import Engine
    from sylph_amd.runner import MetaFCOSROIEncoderRunner, create_cfg

    def run_leg(cfg, sd, B, H, Wd, nway, shots, topk, code_scale=1.5, steps=6):
        cfg.MODEL.FCOS.POST_NMS_TOPK_TEST = topk
        e = Engine(cfg, dtype=args.dtype, device=local_rank)
        e.load_state_dict(sd)
        q = dev_images(B, H, Wd, 11, device)
        cw = cb = None

        def launch():
            e.preprocess(q); e.backbone(); e.head(cw, cb)
            return e.decode_launch()

        def run(n):
            pend, out = [], None
            for _ in range(n):
                pend.append(launch())
                if len(pend) >= 2:
                    out = e.decode_fetch(pend.pop(0))
            while pend:
                out = e.decode_fetch(pend.pop(0))
            return out
        # synthetic codes: the largest of a few scales whose candidate count fits the decode buffers (1 / 8 of all scores per level)
        for scale in (code_scale, 1.0, 0.7, 0.5):
            cm = W.synthetic_codes(nway, seed=3, scale=scale)
            cw, cb = cm["cls_conv"].to(device), cm["cls_bias"].to(device)
            try:
                run(3)
                code_scale = scale
                break
            except RuntimeError as ex:
                if "candidate capacity" not in str(ex):
                    raise
        else:
            raise RuntimeError("no synthetic code scale fits the candidate buffers")
        e.profile_enable(True); e.profile_read()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        dets = run(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - ts
        prof = e.profile_read()
        e.profile_enable(False)
        res = {"img_s": round(B * steps / dt, 1), "batch": B, "image": [H, Wd], "ways": nway, "shots": shots, "code_scale": code_scale,
               "gflop_per_image": round(prof["conv_flops"] / (B * steps) / 1e9, 2),
               "roofline_frac": round(prof["conv_flops"] / (prof["conv_ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4) if prof["conv_ms"] > 0 else None,
               "detections_per_image": round(sum(int(d["scores"].numel()) for d in dets) / len(dets), 1)}
        # support path of the same model: 12 classes x shots images per batch (fewer when the images are few)
        ncls = max(1, 60 // shots)
        sup = dev_images(ncls * shots, H, Wd, 4200 + shots, device)
        bxs = torch.cat([W.synthetic_boxes(shots, H, Wd, seed=5200 + c) for c in range(ncls)])

        def sup_step():
            e.preprocess(sup); e.backbone()
            return e.codegen_classes(bxs, shots)
        sup_step()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(4):
            sup_step()
        torch.cuda.synchronize()
        res["support_img_s"] = round(4 * ncls * shots / (time.perf_counter() - ts), 1)
        res["support_images_per_episode"] = nway * shots
        res["episode_support_s"] = round(nway * shots / res["support_img_s"], 3)
        e.close()
        return res

    legs = {}
    leg_batch = int(os.environ.get("SYLPH_BENCH_LEG_BATCH", "192"))  # query images per step of the C4 / C5 legs
    legs["c2_r50_20way_10shot"] = run_leg(make_cfg(), W.synthetic_state_dict(0, depth=50), 16, args.height, args.width, 20, 10, 100, steps=12)
    legs["c2_r50_20way_10shot"]["config"] = "BASELINE configs[2]: R-50-FPN COCO 20 novel classes, 10-shot, batch 16 queries"
    cfg4 = make_cfg()
    cfg4.MODEL.RESNETS.DEPTH = 101
    legs["c4_r101_866way"] = run_leg(cfg4, W.synthetic_state_dict(0, depth=101), leg_batch, args.height, args.width, 866, 5, 300)
    legs["c4_r101_866way"]["config"] = ("BASELINE configs[3]: R-101-FPN LVISv1 Meta-FCOS, 866-way 5-shot; one rank's query share of the 8-GPU job "
                                        "(the code all-gather over xGMI is not part of a query step)")
    runner = MetaFCOSROIEncoderRunner()
    cfg5 = create_cfg(runner.get_default_cfg(), "sylph://LVISv1-Detection/Meta-FCOS/Meta-FCOS-ROI-Encoder-finetune.yaml", ["MODEL.META_LEARN.EVAL_SHOT", 5])
    sd5 = {}
    sd5.update(W.backbone_state_dict(0, depth=50)); sd5.update(W.head_state_dict(1, num_classes=60)); sd5.update(W.roi_encoder_state_dict(seed=4))
    legs["c5_roi_encoder_337way"] = run_leg(cfg5, sd5, leg_batch, 800, 1200, 337, 5, 300)
    legs["c5_roi_encoder_337way"]["config"] = ("BASELINE configs[4]: ROI-Encoder code generator + CondConvBlock head (Meta-FCOS-ROI-Encoder-finetune.yaml), "
                                               "LVIS rare 337-way 5-shot, 800x1200 queries; one rank's share of the 4-GPU job")
    return legs


_LIVE_PMC = None  # the summary collected by collect_live_pmc() in THIS run


def collect_live_pmc(batch):
    """roofline.traffic collected in the run itself: two rocprofv3 passes (--pmc FETCH_SIZE, then --pmc WRITE_SIZE,
    each with --kernel-trace only, as MI355X_MICROARCH.md prescribes: separate passes) of a short child run of this same script at the
    same batch, summarised by tools/rocpd_pmc.py (last query step; FETCH_SIZE doubled on gfx950).  Any failure leaves the committed
    profiles/r6_pmc_hbm_traffic.json (fingerprint-checked) as the source."""
    global _LIVE_PMC
    import shutil
    import subprocess
    import tempfile
    if os.environ.get("SYLPH_BENCH_PMC_CHILD") or shutil.which("rocprofv3") is None:
        return
    tmp = tempfile.mkdtemp(prefix="sylph_pmc_", dir="/tmp")
    env = dict(os.environ, SYLPH_BENCH_PMC_CHILD="1", TMPDIR="/tmp")
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(batch), "--steps", "2", "--warmup", "2", "--no-cpu-baseline",
             "--no-kernel-events", "--no-sweep", "--no-parity", "--no-live-pmc"]
    try:
        for counter, tag in (("FETCH_SIZE", "f"), ("WRITE_SIZE", "w")):
            r = subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", os.path.join(tmp, tag), "-o", tag, "--"] + child,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=420)
            if r.returncode != 0 or not os.path.exists(os.path.join(tmp, tag, f"{tag}_results.db")):
                print(f"bench.py: live PMC pass {counter} failed (rc {r.returncode}): {r.stderr[-300:]}", file=sys.stderr)
                return
        out = os.path.join(tmp, "pmc.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_pmc.py"), os.path.join(tmp, "f", "f_results.db"),
                            os.path.join(tmp, "w", "w_results.db"), str(batch), out], capture_output=True, text=True, timeout=120)
        if r.returncode != 0:
            print(f"bench.py: tools/rocpd_pmc.py failed: {r.stderr[-300:]}", file=sys.stderr)
            return
        with open(out) as f:
            _LIVE_PMC = json.load(f)
    except Exception as e:  # noqa: BLE001 -- measurement aid: never fail the bench line
        print(f"bench.py: live PMC collection failed: {e}", file=sys.stderr)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _pmc_file():
    """The committed PMC summary of THIS build (tools/collect_profiles.sh -> tools/rocpd_pmc.py), or (None, reason): a summary whose
    kernel-source fingerprint differs from the tree is measured on other kernels and is refused -- loudly, no fall-back to older rounds."""
    if _LIVE_PMC is not None:
        return _LIVE_PMC, "collected in this run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of a 2-step child run)"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from rocpd_pmc_fingerprint import csrc_fingerprint
    path = os.path.join(ROOT, "profiles", "r6_pmc_hbm_traffic.json")
    if not os.path.exists(path):
        return None, "profiles/r6_pmc_hbm_traffic.json is missing"
    with open(path) as f:
        d = json.load(f)
    if d.get("csrc_fingerprint") != csrc_fingerprint():
        return None, (f"profiles/r6_pmc_hbm_traffic.json was collected on other kernel sources (fingerprint {d.get('csrc_fingerprint')} != "
                      f"{csrc_fingerprint()}): re-run tools/collect_profiles.sh")
    return d, "profiles/r6_pmc_hbm_traffic.json"


def pmc_per_kernel_bytes(batch):
    """HBM bytes per step and kernel from the committed PMC passes (same command; scaled by batch), keyed by kernel name."""
    d, _ = _pmc_file()
    if d is None:
        return {}
    out = {}
    for k, v in d.get("per_kernel_hbm_bytes_per_image", {}).items():
        kk = k
        for tag in ("conv_igemm_kernel", "conv_pw_kernel", "conv_spw_kernel", "conv_hpipe_kernel<true>", "conv_hpipe_kernel<false>", "bottleneck64p_kernel",
                    "bottleneck64_kernel", "stem_pool_kernel", "gn_logits_kernel", "gn_taps_kernel"):
            if tag in k:
                kk = tag
                break
        out[kk] = out.get(kk, 0) + v * batch
    return out


def pmc_traffic_per_launch(batch, launches_per_step):
    """HBM bytes per conv launch from the committed rocprofv3 PMC passes (separate runs of this same command, as the
    guide prescribes: FETCH_SIZE doubled on gfx950, WRITE_SIZE calibrated on the first kernel of the step; tools/rocpd_pmc.py).
    Scaled from the profiled batch to this run's batch (traffic is per image).  Returns (bytes or None, source label)."""
    d, src = _pmc_file()
    if d is None or launches_per_step <= 0:
        print(f"bench.py: roofline.traffic = null: {src}", file=sys.stderr)
        return None, {"stale_or_missing": src}
    return round(d["hbm_bytes_per_image"] * batch / launches_per_step), {
        "file": src, "profiled_batch": d.get("batch"), "hbm_bytes_per_image": round(d["hbm_bytes_per_image"]),
        "csrc_fingerprint": d.get("csrc_fingerprint"),
        "note": "separate rocprofv3 --pmc passes of this command on these kernel sources"}


def parity_bf16(sd, queries, cls_conv, cls_bias, dets):
    """Agreement of the timed (bf16) configuration with the fp32 CPU oracle on the first query images: fraction of the
    oracle's detections that the HIP path reproduces (same class, IoU >= 0.9) and the largest score difference over
    those.  The oracle is the checker here, never the thing measured."""
    from oracle import episode as E
    codes = {"cls_conv": cls_conv.cpu(), "cls_bias": cls_bias.cpu()}
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))  # (the oracle is fastest at 16 threads on the 256-CPU boxes)
    matched, total, dmax = 0, 0, 0.0
    with torch.no_grad():
        for q, d in zip(queries, dets):
            w = E.forward_instances([q.cpu()], codes, sd)[0]
            gb, gc, gs = d["pred_boxes"].float().cpu(), d["pred_classes"].cpu(), d["scores"].float().cpu()
            wb, wc, ws = w["pred_boxes"], w["pred_classes"], w["scores"]
            total += int(ws.numel())
            if wb.numel() == 0 or gb.numel() == 0:
                continue
            lt = torch.max(wb[:, None, :2], gb[None, :, :2])
            rb = torch.min(wb[:, None, 2:], gb[None, :, 2:])
            inter = (rb - lt).clamp(min=0).prod(-1)
            aw, ag = (wb[:, 2] - wb[:, 0]) * (wb[:, 3] - wb[:, 1]), (gb[:, 2] - gb[:, 0]) * (gb[:, 3] - gb[:, 1])
            iou = inter / (aw[:, None] + ag[None, :] - inter)
            iou = torch.where(wc[:, None] == gc[None, :], iou, torch.zeros_like(iou))
            best, idx = iou.max(dim=1)
            ok = best >= 0.9
            matched += int(ok.sum())
            if ok.any():
                dmax = max(dmax, float((gs[idx] - ws).abs()[ok].max()))
    return {"matched_frac": round(matched / max(total, 1), 4), "max_abs_dscore": round(dmax, 5), "n_ref": total,
            "images": len(queries), "criterion": "same class and IoU >= 0.9 against the fp32 CPU oracle's detections"}


def cpu_baseline(sd, queries, cls_conv, cls_bias, n_images):
    """The CPU oracle (fp32 torch restatement of the reference path) on the host cores, batch 1,
    first image excluded as warm-up (the reference protocol, meta_learn_evaluation.py:392-417)."""
    from oracle import episode as E
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # pick the thread count that is actually fastest on this host (oversubscribed pods run the oracle ~50x slower at cpu_count()
    # threads; a single tower conv once picked 128 threads on a 128-core host where the whole oracle then ran 4 x slower than at 32, and
    # three conv shapes picked 64 where the whole path is faster at 32): time the ORACLE ITSELF, on a quarter-size image, per candidate
    codes = {"cls_conv": cls_conv.cpu(), "cls_bias": cls_bias.cpu()}
    probe = [queries[0][:, :400, :672].cpu().contiguous()]
    best, best_t = 1, float("inf")
    with torch.no_grad():
        torch.set_num_threads(min(16, avail))
        E.forward_instances(probe, codes, sd)  # warm-up (allocator, weight layout caches)
        for nt in sorted({t for t in (8, 16, 32, 64, min(avail, 64)) if t <= avail}):  # (more than 64 threads never won and cost seconds each)
            torch.set_num_threads(nt)
            t0 = time.perf_counter()
            E.forward_instances(probe, codes, sd)
            dt = time.perf_counter() - t0
            if dt < best_t:
                best, best_t = nt, dt
    torch.set_num_threads(best)
    imgs = [q.cpu() for q in queries[: n_images + 1]]
    times = []
    with torch.no_grad():
        for i, im in enumerate(imgs):
            t0 = time.perf_counter()
            E.forward_instances([im], codes, sd)
            dt = time.perf_counter() - t0
            if i > 0 or len(imgs) == 1:
                times.append(dt)
    v = len(times) / sum(times)
    return {"value": round(v, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(times)} query image(s) 800x1333 through the fp32 CPU oracle (batch 1, 1 warm-up image)"}


if __name__ == "__main__":
    main()
