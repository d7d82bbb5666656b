#!/bin/bash
# main-loop-only timing (epilogue compiled out) of the eight-wave and four-wave 256x256 halo kernels
V=$GRAFT_REPO_ROOT/sylph-few-shot-detection_amd/lib/variants
python tools/bench_3x3.py 64 20
SYLPH_CONV_HQ=1 python tools/bench_3x3.py 64 20
SYLPH_LIB_PATH=$V/libsylph_hp_noepi2.so python tools/bench_3x3.py 64 20
SYLPH_CONV_HQ=1 SYLPH_LIB_PATH=$V/libsylph_hq_noepi.so python tools/bench_3x3.py 64 20
