#!/bin/bash
# single-chunk renders without a checkpoint: tests, then same-box A/B against libptshim_old.so on configs[1] and on a one-launch configs[2]
set -o pipefail
mkdir -p gpurun_out/r04
out=gpurun_out/r04/onechunk.txt; : > $out
timeout -k 10 900 python -m pytest tests/test_gpu_streaming.py tests/test_gpu_lbvh_robust.py -m gpu -x -q > gpurun_out/r04/onechunk_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r04/onechunk_pytest.log
[ $rc -ne 0 ] && { tail -40 gpurun_out/r04/onechunk_pytest.log; exit $rc; }
run() { echo "== $1" >> $out; shift; env "$@" >> $out 2>&1 || { tail -5 $out; exit 1; }; }
OLD=PT_SHIM_LIB=$(pwd)/oclpathtracer_amd/libptshim_old.so
for rep in 1 2; do
run "committed library, configs[1]" $OLD timeout -k 10 200 python tools/step_time.py 512 512 64 2 400
run "this one, configs[1]" X=1 timeout -k 10 200 python tools/step_time.py 512 512 64 2 400
run "committed library, 256 x 256 x 1 frame" $OLD timeout -k 10 200 python tools/step_time.py 256 256 1 16 2000
run "this one, 256 x 256 x 1 frame" X=1 timeout -k 10 200 python tools/step_time.py 256 256 1 16 2000
run "committed library, configs[2] one launch" $OLD PT_STAGING_MB=100000 timeout -k 10 200 python tools/step_time.py
run "this one, configs[2] one launch" PT_STAGING_MB=100000 timeout -k 10 200 python tools/step_time.py
done
cat $out
