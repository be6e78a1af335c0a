#!/bin/bash
# same-box A/B of two libraries on the headline render (tools/step_time.py): libptshim_old.so against libptshim.so, alternating
set -o pipefail
for rep in 1 2 3; do
  for lib in libptshim_old.so libptshim.so; do
    echo -n "$lib  "; PT_SHIM_LIB=$(pwd)/oclpathtracer_amd/$lib timeout -k 10 200 python tools/step_time.py || exit 1
  done
done
