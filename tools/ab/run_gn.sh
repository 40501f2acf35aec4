for lib in head tools/ab/libsiss_gn1535.so tools/ab/libsiss_gn1151.so head; do
  if [ $lib = head ]; then unset SISS_LIB_PATH; else export SISS_LIB_PATH=$PWD/$lib; fi
  echo "== $lib"; python tools/bench_gn.py 2>/dev/null | head -4 | cut -c1-110
done
unset SISS_LIB_PATH
for i in 1 2 3; do for cfg in "subpixel_up=0" "subpixel_up=1"; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --engine-attr $cfg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['step_ms']['p50'])"
done; done
