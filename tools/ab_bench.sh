for rep in 1 2; do
for v in old new; do
  if [ $v = old ]; then export AAD_HIP_LIBRARY=$PWD/build/libaad_hip_old.so; else unset AAD_HIP_LIBRARY; fi
  python bench.py --no-saturated --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['encode_msps'], d['decode_msps'], d['roofline']['kernel_ms'], d['roofline']['decode_kernel']['kernel_ms'])"
done; done
