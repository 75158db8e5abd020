#!/bin/bash
# Captures every rocprofv3 pass profiles/README.md lists, on the GPU box, into gpurun_out/prof/.
# usage (from the repo root on the box): bash tools/capture_profiles.sh
# PMC passes are separate runs with --kernel-trace only (no sys/hip trace domains), one counter group per pass.
set -u
R=$(pwd); O=$R/gpurun_out/prof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P=python3
run() { name=$1; shift; echo "== $name"; timeout -k 10 300 rocprofv3 "$@" > $O/$name.stdout 2> $O/$name.stderr; echo "rc=$?"; }
run regen_trace --kernel-trace --stats --output-format csv -d $O/regen_trace -- $P $R/tools/prof_regen.py sdobj 4 40 || exit 1
run regen_pmc_sq --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/regen_pmc_sq -- $P $R/tools/prof_regen.py sdobj 4 5 || exit 1
run regen_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/regen_fetch -- $P $R/tools/prof_regen.py sdobj 4 3 || exit 1
run regen_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/regen_write -- $P $R/tools/prof_regen.py sdobj 4 3 || exit 1
run dense_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/dense_write -- $P $R/tools/dense_once.py || exit 1
run dense_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/dense_fetch -- $P $R/tools/dense_once.py || exit 1
run bench_trace --kernel-trace --stats --output-format csv -d $O/bench_trace -- $P $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-sizes --no-pipelined --no-anim --no-ref-abi || exit 1
# the one-mesh steps on a one-rank RCCL communicator (what one GPU can run of the multi-GPU protocol): kernel traces + wall times
for k in 4 5; do
  run zslab_trace_$k --kernel-trace --stats --output-format csv -d $O/zslab_trace_$k -- $P $R/tools/prof_sharded.py zslab sdobj $k 40 || exit 1
  run xslab_trace_$k --kernel-trace --stats --output-format csv -d $O/xslab_trace_$k -- $P $R/tools/prof_sharded.py xslab sdobj $k 40 || exit 1
done
( for f in plain zslab xslab; do for k in 3 4 5; do $P $R/tools/prof_sharded.py $f sdobj $k 200 2>/dev/null | grep ms_per_step; done; done ) > $O/one_mesh_one_rank.txt
cd $R
for u in valu_rate valu_issue; do      # (built in-tree by the builder; a fresh clone compiles them here)
  [ -x tools/ubench/$u ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $([ $u = valu_rate ] && echo -ffp-contract=off) tools/ubench/$u.hip -o tools/ubench/$u
done
echo "== ubench"; timeout -k 10 60 ./tools/ubench/valu_rate > $O/valu_rate.jsonl 2> $O/valu_rate.stderr; echo "rc=$?"
echo "== ubench (machine instructions)"; timeout -k 10 200 ./tools/ubench/valu_issue > $O/valu_issue.jsonl 2> $O/valu_issue.stderr; echo "rc=$?"
echo "== projection timeline"; timeout -k 10 100 $P tools/project_timeline.py sdobj 4 $O/project_timeline.json > $O/project_timeline.stdout 2>&1; echo "rc=$?"
echo "== dedupe phases"; timeout -k 10 100 $P tools/tile_phases.py sdobj 4 > $O/tile_phases.txt 2>&1; echo "rc=$?"
echo "== plain bench"; timeout -k 10 500 $P bench.py > $O/bench_plain.json 2> $O/bench_plain.stderr; echo "rc=$?"
$P $R/tools/streaming_window.py $O/bench_trace $O/streaming_window.csv
# keep only the small summaries (the raw traces are large)
find $O -name '*_kernel_trace.csv' -size +4M -delete
find $O -name '*_counter_collection.csv' -size +8M -delete
du -sh $O
