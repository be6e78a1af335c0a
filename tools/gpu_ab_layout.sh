#!/bin/bash
# same-box A/B: the committed layout (libptshim_old.so) against the new one, ring and one launch
set -o pipefail
mkdir -p gpurun_out/r04
out=gpurun_out/r04/ab_layout.txt; : > $out
run() { echo "== $1" >> $out; shift; env "$@" timeout -k 10 200 python tools/step_time.py >> $out 2>&1 || { tail -5 $out; exit 1; }; }
OLD=PT_SHIM_LIB=$(pwd)/oclpathtracer_amd/libptshim_old.so
for rep in 1 2; do
run "old, ring 2 x 192 MiB" $OLD PT_STAGING_MB=384
run "old, one launch" $OLD PT_STAGING_MB=100000
run "new, chains 1, 2 slots, 384 MiB" PT_SHIM_CHAINS=1 PT_SHIM_RING_SLOTS=2 PT_STAGING_MB=384
run "new, chains 1, 2 slots, 384 MiB, lanes 1" PT_SHIM_CHAINS=1 PT_SHIM_RING_SLOTS=2 PT_STAGING_MB=384 PT_LANES=1
run "new, chains 2, 3 slots, 384 MiB" PT_SHIM_CHAINS=2 PT_SHIM_RING_SLOTS=3 PT_STAGING_MB=384
run "new, chains 2, 4 slots, 384 MiB" PT_SHIM_CHAINS=2 PT_SHIM_RING_SLOTS=4 PT_STAGING_MB=384
run "new, one launch" PT_STAGING_MB=100000
done
cat $out
