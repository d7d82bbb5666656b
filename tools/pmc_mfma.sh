#!/bin/bash
# MFMA-utilisation / stall / LDS counters of the conv kernels, per layer shape (rocprofv3 --pmc passes, kernel-trace only).
#   tools/pmc_mfma.sh <outdir> <batch> <layer> [<layer> ...]      layers: names of tools/pmc_layer.py
# Writes <outdir>/pmc_mfma.json: {layer: {kernel: {counter: per-launch value, ..., derived: {...}}}}
OUT=$1; B=$2; shift 2
mkdir -p $OUT; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
SETS=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
      "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
      "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE SQ_CYCLES")
for L in "$@"; do
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace -d $ROOT/$OUT/$L/p$i -o p -- python $ROOT/tools/pmc_layer.py $L $B > $ROOT/$OUT/$L.p$i.log 2>&1) || echo "pass $i of $L failed"
  done
done
python - "$OUT" "$B" "$@" <<'PY'
import glob, json, sqlite3, sys
out, B, layers = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
res = {}
for L in layers:
    per = {}
    for f in sorted(glob.glob(f"{out}/{L}/p*/**/*_results.db", recursive=True)):
        db = sqlite3.connect(f)
        tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
        if "counters_collection" not in tabs:
            continue
        for k, c, v in db.execute("select kernel_name, counter_name, value from counters_collection where kernel_name like '%conv_%'"):
            k = k.split("(")[0].replace("sylph::", "")[:60]
            a = per.setdefault(k, {}).setdefault(c, [0, 0.0]); a[0] += 1; a[1] += v
    res[L] = {}
    for k, cs in per.items():
        d = {c: a[1] / a[0] for c, a in cs.items()}
        d["launches_seen"] = max(a[0] for a in cs.values())
        der = {}
        if d.get("SQ_BUSY_CYCLES") and d.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 256 x 4 SIMDs (= 32 cycles x SQ_INSTS_MFMA for the 8-pass bf16 MFMA);
            # SQ_BUSY_CYCLES is summed over the 32 shader engines (8 XCDs x 4) -> / 32 = the launch's duration in shader clocks
            der["mfma_busy_cycles_per_simd"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (256 * 4)
            der["kernel_cycles"] = d["SQ_BUSY_CYCLES"] / 32
            der["mfma_util"] = round(der["mfma_busy_cycles_per_simd"] / der["kernel_cycles"], 4)
        if d.get("SQ_WAVE_CYCLES"):
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if c in d:
                    der[c + "/WAVE_CYCLES"] = round(d[c] / d["SQ_WAVE_CYCLES"], 4)
        if d.get("SQ_LDS_IDX_ACTIVE"):
            der["lds_bank_conflict_frac"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"], 4)
        d["derived"] = der
        res[L][k] = d
json.dump({"batch": B, "layers": res}, open(f"{out}/pmc_mfma.json", "w"), indent=1)
for L in layers:
    for k, d in res[L].items():
        print(L, k, json.dumps(d["derived"]))
PY
