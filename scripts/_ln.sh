timeout 600 python -m pytest tests/test_gpu_producers.py -x -q -m gpu 2>&1 | tail -15
timeout 200 python bench.py --steps 3 --warmup 1 --cpu-baseline off 2> gpurun_out/ln_bench.err | cut -c1-400; tail -5 gpurun_out/ln_bench.err
timeout 200 python bench.py --steps 3 --warmup 1 --cpu-baseline off --producers stock 2>/dev/null | cut -c1-300
