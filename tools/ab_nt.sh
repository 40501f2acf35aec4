python tools/bench_kernels.py --iters 10 --only nt 2>&1 | grep -E "fprop|dgrad"
