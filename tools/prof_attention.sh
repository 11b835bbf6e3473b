#!/bin/bash
# rocprofv3 passes of the headline path (bench.py --no-extra): kernel trace + FETCH_SIZE + WRITE_SIZE, each its own run
set -e
export TMPDIR=/tmp
out=$1
mkdir -p $out
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-extra --steps 100 > $out/trace_bench.json 2> $out/trace.log
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmcf -- python3 bench.py --no-extra --steps 10 > $out/pmcf_bench.json 2> $out/pmcf.log
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmcw -- python3 bench.py --no-extra --steps 10 > $out/pmcw_bench.json 2> $out/pmcw.log
python3 tools/rocprof_summary.py $out/trace $out/pmcf $out/pmcw > $out/summary.txt 2>&1
