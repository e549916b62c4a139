#!/bin/bash
# Kernel time of the any-channel (8-channel) dense kernels against the number of streams around multiples of 1024 waves
# (one dense wave = 64 lanes = 8 eight-channel streams; the chip has 1024 SIMDs): where does BASELINE config 4
# (10 000 streams = 1250 waves) sit?  Usage (GPU box): bash tools/wave_quantisation_sweep.sh > gpurun_out/r03_wave_quantisation.txt
set -u
cd "$(dirname "$0")/.."
for bits in 3 2; do
  echo "== 8 channels, ${bits}-bit, one block per stream (HIP events, mean of 20)"
  for streams in 4096 8000 8192 8200 8704 9216 10000 12288 16384 16392 20000 24576; do
    line=$(python3 tools/saturated_probe.py --bits $bits --channels 8 --streams $streams --reps 20 2>/dev/null | tail -1)
    python3 - "$streams" "$line" <<'PY'
import json, sys
s, d = int(sys.argv[1]), json.loads(sys.argv[2])
waves = -(-s * 8 // 64)
print("  streams %6d  waves %5d (%.3f per SIMD)  encode %.4f ms  decode %.4f ms  -> %.0f / %.0f Gsamples/s, %.3f / %.3f of 8 TB/s"
      % (s, waves, waves / 1024, d["encode_ms"], d["decode_ms"], d["encode_gsps"], d["decode_gsps"], d["encode_tbs"] / 8, d["decode_tbs"] / 8))
PY
  done
done
