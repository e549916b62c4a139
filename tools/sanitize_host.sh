#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over everything that runs on the HOST (GPU sanitizers are not available on this
# pool): the oracle's C restatement under the crafted-bitstream / golden / reference-comparison suites, the library's host C
# (aad_format.c, aad_legacy_api.c, aad_wav.c, aad_synth.c, linked with the regular device objects) under the host-API suite, and
# aad_batch under the CLI refusal suite.  Runs in the build container (no GPU needed); restores the regular binaries afterwards.
# usage: bash tools/sanitize_host.sh        (after `make -C aad_amd/csrc`)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined"
PRE="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0   # the interpreter's own allocations are not ours to report
cd $R
echo "== oracle"
gcc -O1 -g -std=c11 -fPIC -Wall -Wextra -fno-strict-overflow -fwrapv $SAN -shared -o oracle/libaad_oracle.so oracle/aad_oracle.c -lm
LD_PRELOAD="$PRE" python -m pytest tests/test_bitstream_fuzz.py tests/test_oracle_golden.py tests/test_trials_high.py tests/test_oracle_vs_ref.py \
    tests/test_oracle_extremes.py -q -x -p no:cacheprovider || { rm -f oracle/libaad_oracle.so; make -s -C oracle libaad_oracle.so; exit 1; }
rm -f oracle/libaad_oracle.so && make -s -C oracle libaad_oracle.so
echo "== library host C"
for f in aad_format aad_legacy_api aad_wav aad_synth; do
  gcc -std=c99 -O1 -g -fPIC -Wall -Wextra -Iinclude $SAN -c aad_amd/csrc/$f.c -o $T/$f.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $T/libaad_hip_asan.so $T/*.o build/csrc/aad_hip_engine.o build/csrc/aad_decode_split.o \
    build/csrc/aad_decode_tiled.o -Wl,-rpath,/opt/rocm/lib $SAN
LD_PRELOAD="$PRE" AAD_HIP_LIBRARY=$T/libaad_hip_asan.so python -m pytest tests/test_host_api.py -q -x -p no:cacheprovider
echo "== aad_batch"
cp aad_amd/aad_batch $T/aad_batch.orig
gcc -std=c99 -O1 -g -Wall -Wextra -Iinclude -pthread $SAN -o aad_amd/aad_batch aad_amd/cli/aad_batch.c -Laad_amd -laad_hip -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib
python -m pytest tests/test_cli_host.py -q -x -p no:cacheprovider; rc=$?
cp $T/aad_batch.orig aad_amd/aad_batch
rm -rf $T
exit $rc
