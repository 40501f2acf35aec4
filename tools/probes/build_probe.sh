#!/bin/bash
# Probe build of the C-ABI library: the product sources compiled with -DSISS_PROBE, which adds the in-kernel phase timers
# ($SISS_NT_DEBUG_PTR) and the ablation switches ($SISS_NT_ABLATE: 1 no stores, 2 no DMA after the first two groups,
# 4 no MFMAs) of the persistent 3x3 kernel.  Output: tools/probes/libsiss_hip_probe.so -- load it with
#   SISS_LIB_PATH=tools/probes/libsiss_hip_probe.so python tools/probes/c3p_ticks.py
# The product library (siss_amd/libsiss_hip.so) contains none of this.
# Also here, as records of measured-and-rejected kernels (docs/experiments.md; they are NOT part of either build):
#   gemm_nt_c3.hip     one-tile-per-block predecessor of gemm_nt_c3p      gemm_nt_conv3.hip  A tile shared by three taps in the generic kernel
#   groupnorm2p.hip    two-phase on-chip GroupNorm (2x slower than two passes)
#   gemm_tn_wide.hip   one-tap wgrad with a 128 x 384 tile (all-role waves)        gemm_tn_pc.hip     the same as producer / consumer, 128 x 256
set -e
cd "$(dirname "$0")/../.."
out=tools/probes/_probe_build; mkdir -p $out
for f in siss_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  # the per-file flags of the PRODUCT build (siss_amd/build.py: EXACT, EXTRA), so that the probe library's kernels are the product's
  extra=$(python -c "import sys; sys.path.insert(0, '.'); from siss_amd import build as B; f = '$b.hip'; print(' '.join((['-ffp-contract=off'] if f in B.EXACT else []) + B.EXTRA.get(f, [])))")
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSISS_PROBE $extra -c $f -o $out/$b.o &
  if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $out/*.o -o tools/probes/libsiss_hip_probe.so
echo built tools/probes/libsiss_hip_probe.so
