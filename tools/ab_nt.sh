for a in 0 8; do echo "== ABLATE $a"; SISS_NT_ABLATE=$a python tools/bench_kernels.py --iters 10 --only nt 2>&1 | grep -E "fprop|dgrad" | head -8; done
