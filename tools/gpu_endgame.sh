#!/bin/bash
# endgame refill: streaming parity tests, the launch stamps, same-box step time against libptshim_old.so
set -o pipefail
mkdir -p gpurun_out/r04
out=gpurun_out/r04/endgame.txt; : > $out
timeout -k 10 900 python -m pytest tests/test_gpu_streaming.py -m gpu -x -q > gpurun_out/r04/endgame_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r04/endgame_pytest.log
[ $rc -ne 0 ] && { tail -40 gpurun_out/r04/endgame_pytest.log; exit $rc; }
PT_SHIM_LIB=$(pwd)/oclpathtracer_amd/libptshim_lstamps.so timeout -k 10 200 python tools/launch_stamps.py 1 >> $out 2>&1 || { tail $out; exit 1; }
run() { echo "== $1" >> $out; shift; env "$@" timeout -k 10 200 python tools/step_time.py >> $out 2>&1 || { tail -5 $out; exit 1; }; }
OLD=PT_SHIM_LIB=$(pwd)/oclpathtracer_amd/libptshim_old.so
for rep in 1 2; do
run "old, ring" $OLD X=1
run "new, ring" X=1
run "old, one launch" $OLD PT_STAGING_MB=100000
run "new, one launch" PT_STAGING_MB=100000
done
cat $out
