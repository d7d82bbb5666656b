#!/bin/bash
# Round-2 evidence, run on the GPU box through gpurun: kernel trace + timeline + HBM PMC passes of the default bench command,
# MFMA / stall / LDS counters of the hot conv shapes, and a full default bench line.  Outputs land in gpurun_out/r2/.
OUT=gpurun_out/r2; mkdir -p $OUT
bash tools/collect_profiles.sh r2 > $OUT/collect.log 2>&1
bash tools/pmc_mfma.sh $OUT/pmc 64 tower tower.gnin fpn.out3 res4.conv2 > $OUT/pmc_mfma.log 2>&1
cp $OUT/pmc/pmc_mfma.json $OUT/pmc_mfma.json
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/pmc/*/p*
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/collect.log; tail -6 $OUT/pmc_mfma.log; cut -c1-400 $OUT/bench.json
