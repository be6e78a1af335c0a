#!/bin/bash
# round-2 first GPU session: the whole GPU test tier (new full-size + N-rank tests), smoke, default bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu --durations=15 -s > gpurun_out/r02a_pytest.log 2>&1
rc=$?; echo "pytest -m gpu rc=$rc"; tail -30 gpurun_out/r02a_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02a_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r02a_smoke.log
timeout -k 10 600 python bench.py > gpurun_out/r02a_bench.log 2> gpurun_out/r02a_bench.err
echo "bench rc=$?"; tail -c 3000 gpurun_out/r02a_bench.log; tail -5 gpurun_out/r02a_bench.err
