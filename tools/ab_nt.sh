for a in 0 1; do echo "== SMALL $a"; SISS_NT_SMALL=$a python tools/bench_kernels.py --iters 20 --only nt 2>&1 | grep -E "fprop|dgrad" | tail -6; done
