#!/usr/bin/env python
"""HBM traffic of the conv kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; rocpd dbs).

    python tools/rocpd_pmc.py <fetch.db> <write.db> <batch> [out.json]

Sums the counters over the conv launches (conv_igemm / conv_hpipe / conv_pw / bottleneck64[p] / stem_pool / gn_logits / gn_taps + tap_gather) of the LAST query
step (from its first kernel on: preprocess_kernel, or stem_pool_kernel when the normalisation is fused into it); also reports the sum over EVERY kernel of that step.  Units/corrections as MI355X_MICROARCH.md prescribes: the counters are
KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads -> doubled; WRITE_SIZE is
used as is (checked here against the step's first kernel, whose write volume is known exactly)."""
import json
import os
import sqlite3
import sys


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocpd_pmc_fingerprint import csrc_fingerprint  # noqa: E402


def last_step(dbfile, counter):
    db = sqlite3.connect(dbfile)
    rows = db.execute("select kernel_name, value, start from counters_collection where counter_name=? order by start",
                      (counter,)).fetchall()
    # a query step starts with preprocess_kernel, or -- bf16, normalisation fused into the stem -- with stem_pool_kernel
    first = [i for i, r in enumerate(rows) if "preprocess_kernel" in r[0]] or [i for i, r in enumerate(rows) if "stem_pool_kernel" in r[0]]
    idx = first[-1]
    step = rows[idx:]
    # every launch bench.py times as conv work (sylph_profile_*): the conv kernels proper and the fused passes that replace convs
    CONV = ("conv_igemm_kernel", "conv_hpipe_kernel", "conv_pw_kernel", "conv_spw_kernel", "bottleneck64", "stem_pool_kernel", "stem_conv_kernel", "gn_logits_kernel",
            "gn_taps_kernel", "tap_gather_kernel")
    conv = [r for r in step if any(k in r[0] for k in CONV)]
    pre = step[0][1]
    by = {}
    for r in step:
        k = r[0].split("(")[0].replace("sylph::", "").replace("void ", "")[:48]
        by[k] = by.get(k, 0.0) + r[1] * 1024.0
    return sum(r[1] for r in conv) * 1024.0, len(conv), pre * 1024.0, sum(r[1] for r in step) * 1024.0, by


fetch, n1, pre_f, fetch_all, fetch_by = last_step(sys.argv[1], "FETCH_SIZE")
write, n2, pre_w, write_all, write_by = last_step(sys.argv[2], "WRITE_SIZE")
B = int(sys.argv[3])
assert n1 == n2, (n1, n2)
out = {
    "batch": B, "conv_launches_per_step": n1, "csrc_fingerprint": csrc_fingerprint(),
    "fetch_bytes_raw_per_step": fetch, "fetch_bytes_corrected_per_step": 2.0 * fetch, "write_bytes_per_step": write,
    "hbm_bytes_per_image": (2.0 * fetch + write) / B,
    "hbm_bytes_per_launch": (2.0 * fetch + write) / n1,
    "all_kernels_hbm_bytes_per_image": (2.0 * fetch_all + write_all) / B,
    "per_kernel_hbm_bytes_per_image": {k: round((2.0 * fetch_by.get(k, 0.0) + write_by.get(k, 0.0)) / B) for k in sorted(set(fetch_by) | set(write_by))},
    # first kernel of the step: preprocess_kernel writes B x 800 x 1344 x 4 bf16, the fused stem + pool kernel B x 200 x 336 x 64 bf16
    # (the same byte count) and both read the B x 3 x 800 x 1333 fp32 input (the stem re-reads its tile halos: > expected)
    "calibration": {"first_kernel_write_bytes": pre_w, "first_kernel_write_expected": B * 800 * 1344 * 4 * 2,
                    "first_kernel_fetch_bytes_raw": pre_f, "first_kernel_fetch_expected": B * 3 * 800 * 1333 * 4},
    "note": "FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads); WRITE_SIZE uncorrected.  The fused stem reads its fp32 input with 4-byte loads: "
            "its raw FETCH_SIZE already matches the expected bytes x the tile-halo overlap (see calibration), so the doubled per-kernel figure overstates it (~16 MB/img); "
            "its writes include one 16-byte trash store per thread and tile (constant store count for the vmcnt bookkeeping)",
}
print(json.dumps(out, indent=1))
if len(sys.argv) > 4:
    json.dump(out, open(sys.argv[4], "w"), indent=1)
