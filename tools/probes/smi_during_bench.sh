# clocks / power / temperature of the GPU WHILE the CelebA-HQ step runs (300 replays), one rocm-smi sample per second
mkdir -p gpurun_out
( for i in $(seq 1 14); do sleep 1; rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "sclk|mclk|Power \(W\)|junction|memory\)" | tr -s '\t ' ' ' | tr '\n' ';'; echo; done ) > gpurun_out/smi_samples.txt &
SMI=$!
timeout -k 10 300 python bench.py --steps 300 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('ms_per_step', d['ms_per_step'], d['step_ms'])"
wait $SMI
cat gpurun_out/smi_samples.txt
