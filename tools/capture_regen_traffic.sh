#!/bin/bash
# HBM traffic of the regen kernels (separate PMC passes, kernel-trace only): FETCH_SIZE and WRITE_SIZE per dispatch.
set -u
R=$(pwd); O=$R/gpurun_out/traffic; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -- python3 $R/tools/prof_regen.py sdobj 4 3 > $O/$c.stdout 2> $O/$c.stderr; echo "$c rc=$?"
done
find $O -name '*_kernel_trace.csv' -delete
