#!/bin/bash
# conv_hpipe ablation builds (tools/build_variant.sh hp_* conv_hpipe.hip ...) on the 3x3 layer shapes
python tools/bench_3x3.py 64 20
for v in $@; do SYLPH_LIB_PATH=$GRAFT_REPO_ROOT/sylph-few-shot-detection_amd/lib/variants/libsylph_$v.so python tools/bench_3x3.py 64 20; done
