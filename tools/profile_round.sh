#!/bin/bash
# Round profile set (run on the GPU box from the repo root): kernel stats of bench.py and PMC HBM traffic per
# kernel class, for the library that is in the tree.  Usage: tools/profile_round.sh r02 f16x2
tag=${1:-r02}; dt=${2:-f16x2}
root=$PWD; out=$root/gpurun_out/prof_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ks_$dt -o k --output-format csv -- python3 $root/bench.py --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline --no-other-modes > $out/${tag}_bench_${dt}.json 2> /dev/null
cp $(ls /tmp/ks_$dt/*kernel_stats.csv /tmp/ks_$dt/*/*kernel_stats.csv 2>/dev/null | head -1) $out/${tag}_bench_${dt}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE -d /tmp/pf_$dt -o p --output-format csv -- python3 $root/tools/one_forward.py $dt --per-op > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pw_$dt -o p --output-format csv -- python3 $root/tools/one_forward.py $dt --per-op > /dev/null 2>&1
cd $root && UDP_POSE_NO_GROUPS=1 python3 tools/pmc_by_op.py /tmp/pf_$dt /tmp/pw_$dt $dt > $out/${tag}_traffic_${dt}.json
head -12 $out/${tag}_bench_${dt}_kernel_stats.csv
python3 -c "
import json; d=json.load(open('$out/${tag}_traffic_${dt}.json'))
for k,v in d['classes'].items(): print(k, v['launches'], round(v['hbm_bytes_per_launch']/1e6,1), 'MB/launch', v['traffic_over_algorithmic'])"
