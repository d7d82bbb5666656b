#!/bin/bash
# lib/variants/libsylph_hq.so: the product objects + conv_hpipe.hip rebuilt with the dispatch to the four-wave probe kernel
# (tools/probes/conv_hpipe4.hip, SYLPH_CONV_HQ=1 selects it at run time).  Measurement only.
set -e
cd "$(dirname "$0")/../../sylph-few-shot-detection_amd/csrc"
mkdir -p ../lib/variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DSYLPH_ABLATE -DSYLPH_HQ_PROBE"
/opt/rocm/bin/hipcc $F -c conv_hpipe.hip -o /tmp/hq_conv_hpipe.o
/opt/rocm/bin/hipcc $F -c ../../tools/probes/conv_hpipe4.hip -o /tmp/hq_conv_hpipe4.o
OBJS=""
for o in conv_igemm.o conv_pw.o collective.o bottleneck.o head_fused.o stem_conv.o elementwise.o detect.o codegen.o roi_encoder.o api_core.o api_weights.o api_conv.o api_backbone.o api_head.o api_codegen.o api_parity.o; do OBJS="$OBJS $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/hq_conv_hpipe.o /tmp/hq_conv_hpipe4.o -ldl -o ../lib/variants/libsylph_hq.so
echo built ../lib/variants/libsylph_hq.so
