#!/bin/bash
# tools/build_variant.sh <name> <file.hip> "<extra -D flags>": links lib/variants/libsylph_<name>.so with one translation
# unit rebuilt under -DSYLPH_ABLATE + extra defines (kernel ablations / A-B builds; select with SYLPH_LIB_PATH).
set -e
NAME=$1; SRC=$2; FLAGS=$3
cd "$(dirname "$0")/../sylph-few-shot-detection_amd/csrc"
mkdir -p ../lib/variants
OBJ=/tmp/variant_${NAME}_$(basename $SRC .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DSYLPH_ABLATE $FLAGS -c $SRC -o $OBJ
OBJS=""
for o in conv_igemm.o conv_hpipe.o conv_pw.o conv_spw.o conv_rw3.o collective.o bottleneck.o head_fused.o stem_conv.o elementwise.o detect.o codegen.o roi_encoder.o api_core.o api_weights.o api_conv.o api_backbone.o api_head.o api_codegen.o api_parity.o; do
  if [ "$o" == "$(basename $SRC .hip).o" ]; then OBJS="$OBJS $OBJ"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o ../lib/variants/libsylph_${NAME}.so
echo built ../lib/variants/libsylph_${NAME}.so
