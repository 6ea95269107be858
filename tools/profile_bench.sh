#!/bin/bash
# rocprofv3 passes over bench.py (GPU box, through gpurun): kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in passes of
# their own (no tracing domains beside --pmc).  usage: tools/profile_bench.sh <tag>
set -e
tag=$1
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/stats -o stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/stats_bench.json 2> $out/stats.log || { tail -5 $out/stats.log; exit 1; }
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/fetch_bench.json 2> $out/fetch.log || { tail -5 $out/fetch.log; exit 1; }
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE -d $out/write -o write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/write_bench.json 2> $out/write.log || { tail -5 $out/write.log; exit 1; }
echo "write pass done"
python3 tools/rocprof_db_summary.py stats $(ls $out/stats/*/*.db $out/stats/*.db 2>/dev/null | head -1) $out/kernel_stats.csv
python3 tools/rocprof_db_summary.py hbm $(ls $out/fetch/*/*.db $out/fetch/*.db 2>/dev/null | head -1) $(ls $out/write/*/*.db $out/write/*.db 2>/dev/null | head -1) $out/pmc_hbm.json \
  "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs of python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline (1 image); HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB"
