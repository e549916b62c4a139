#!/bin/bash
# Copy the summaries of the last tools/collect_profiles.sh run (merged back under gpurun_out/prof) into profiles/ (tracked),
# render the configs summary and check that the stamp belongs to the kernel sources in this tree.  usage: bash tools/install_profiles.sh [tag]
set -e
TAG=${1:-r04}
cd "$(dirname "$0")/.."
cp gpurun_out/prof/${TAG}_pmc_stamp.json profiles/${TAG}_pmc_stamp.json
for f in bench_full_kernel_stats_rocprofv3.csv bench_kernel_stats_rocprofv3.csv bench_line.json pmc_summary_bench.txt saturated_kernel_stats_rocprofv3.csv saturated_pmc_summary.txt; do
  cp gpurun_out/prof/${TAG}_$f profiles/${TAG}_$f
done
python tools/config_pmc_summary.py profiles/${TAG}_pmc_stamp.json > profiles/${TAG}_pmc_summary_configs.txt
python - <<PY
import json, sys
sys.path.insert(0, ".")
import bench
d = json.load(open("profiles/${TAG}_pmc_stamp.json"))
ok = d["kernel_source_sha256"] == bench.kernel_source_digest()
print(sorted(d["workloads"]), "stamp matches this tree's kernel sources:", ok)
b = json.load(open("profiles/${TAG}_bench_line.json"))
print("value", b["value"], "enc/dec kernel ms", b["roofline"]["kernel_ms"], b["roofline"]["decode_kernel"]["kernel_ms"], "trials2", b["trials2"]["value"],
      "saturated", b["saturated"]["encode_frac"], b["saturated"]["decode_frac"], "cpu", b["cpu_baseline"]["value"])
sys.exit(0 if ok else 1)
PY
