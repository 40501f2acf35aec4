# Ablation builds of flash_attn32.hip (FA32_ABL = 1..4: see the source) linked into libsiss_hip_ablN.so; run: for n in 0 1 2 3 4 ...
# Usage (build container): bash tools/probes/fa32_ablate.sh build;  (GPU box): bash tools/probes/fa32_ablate.sh run
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  mkdir -p tools/probes/_probe_build
  for n in 1 2 3 4; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize -DFA32_ABL=$n -c siss_amd/csrc/flash_attn32.hip -o tools/probes/_probe_build/fa32_abl$n.o || exit 1
    objs=$(ls siss_amd/build/*.o | grep -v flash_attn32.o)
    hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/probes/_probe_build/fa32_abl$n.o -o tools/probes/_probe_build/libsiss_hip_abl$n.so || exit 1
  done
else
  PRE=1 python tools/probes/flash_time.py 2>&1 | tail -1
  for n in 1 2 3 4; do
    echo "ablation $n"; PRE=1 SISS_LIB_PATH=tools/probes/_probe_build/libsiss_hip_abl$n.so python tools/probes/flash_time.py 2>&1 | tail -1
  done
fi
