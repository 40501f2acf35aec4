for a in 0 1 2 4 6 7 16 17 22; do echo ABLATE $a; SISS_NT_ABLATE=$a python tools/bench_kernels.py --iters 10 --only nt 2>&1 | grep -E "fprop +256\^2 +128|dgrad +256\^2 +128-> 128"; done
