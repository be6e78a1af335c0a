#!/bin/bash
# round-4 record run, part 2: smoke, the pass-1 filter validator (diagnostic build), the randomised parity campaign, all PMC passes of the
# LBVH kernel on configs[4]'s scene.  Everything under gpurun_out/r04rec/.
set -o pipefail
O=gpurun_out/r04rec
mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
rm -f gpurun_out/filter_validation.txt
timeout -k 10 1500 tools/gpu_validate.sh 1024 1024 64 > $O/validate.log 2>&1; echo "validate rc=$?"
cp gpurun_out/filter_validation.txt $O/filter_validation_r04.txt 2>/dev/null
grep -c "VIOLATIONS: 0" $O/filter_validation_r04.txt; grep -v "VIOLATIONS: 0" $O/filter_validation_r04.txt | grep VIOLATIONS | head
timeout -k 10 1200 python tools/gpu_fuzz.py 30000 1500 > $O/fuzz_r04.txt 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz_r04.txt
tools/gpu_pmc_soup.sh r04f 8 > $O/pmcsoup_full.txt 2>&1; tail -4 $O/pmcsoup_full.txt
