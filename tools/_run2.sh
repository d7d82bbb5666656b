set -x
SYLPH_CONV_PW=2 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "conv2d or backbone or full_size_f32" > gpurun_out/r3_pw2.log 2>&1; tail -3 gpurun_out/r3_pw2.log
python -m pytest tests/test_bf16_pinned_gpu.py -m gpu -q -s -k "bottleneck or stagewise" > gpurun_out/r3_bf16pin2.log 2>&1; tail -3 gpurun_out/r3_bf16pin2.log
for v in "" "SYLPH_PW_TILE=1" "SYLPH_PW_TILE=2" "SYLPH_PW_TILE=1 SYLPH_PW_NST=3" "SYLPH_CONV_PW=0"; do echo "== $v"; env $v python tools/bench_layers.py 64 2>&1 | grep -v amdgpu.ids | grep "1x1"; done > gpurun_out/r3_layers_pw2.txt 2>&1
cat gpurun_out/r3_layers_pw2.txt
