set -x
python -m pytest tests/test_bf16_pinned_gpu.py -m gpu -q -s > gpurun_out/r3_bf16pin.log 2>&1; tail -3 gpurun_out/r3_bf16pin.log
SYLPH_CONV_PW=2 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "conv2d or backbone or full_size_f32" > gpurun_out/r3_pw2.log 2>&1; tail -3 gpurun_out/r3_pw2.log
python tools/bench_layers.py 64 > gpurun_out/r3_layers_pw.txt 2>&1; cat gpurun_out/r3_layers_pw.txt
