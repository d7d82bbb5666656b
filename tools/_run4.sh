python -m pytest tests -m gpu -q -x > gpurun_out/r3_gpu_all.log 2>&1; tail -15 gpurun_out/r3_gpu_all.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_a.json 2> gpurun_out/r3_bench_a.err; tail -c 1500 gpurun_out/r3_bench_a.json
