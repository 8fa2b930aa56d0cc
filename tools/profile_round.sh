#!/bin/bash
# Round profile set (run on the GPU box from the repo root) for the library that is in the tree:
#   kernel stats (rocprofv3 --kernel-trace --stats) of bench.py as the driver runs it (two sub-batch lanes) and with
#   UDP_POSE_LANES=1 (the configuration bench.py's roofline pass measures), of the W48 / RSN-18 benches and of the
#   training step; PMC HBM traffic per kernel class (separate FETCH_SIZE / WRITE_SIZE passes); stall counters of the
#   dominant kernel.   Usage: tools/profile_round.sh r03 [f16x2]
tag=${1:-r03}; dt=${2:-f16x2}
root=$PWD; out=$root/gpurun_out/prof_$tag; mkdir -p "$out"
run=/tmp/prof_$$; mkdir -p $run
cd /tmp && export TMPDIR=/tmp
stats() {   # name, then the python command line
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -d $run/ks_$name -o k --output-format csv -- python3 "$@" > "$out/${tag}_${name}.json" 2> /dev/null
  cp "$(ls $run/ks_$name/*kernel_stats.csv $run/ks_$name/*/*kernel_stats.csv 2>/dev/null | head -1)" "$out/${tag}_${name}_kernel_stats.csv"
  echo "== $name"; head -6 "$out/${tag}_${name}_kernel_stats.csv" | cut -c1-150
}
B="--steps 20 --warmup 5 --no-cpu-baseline --no-other-modes --no-other-configs"
stats bench_${dt} $root/bench.py --dtype $dt $B
export UDP_POSE_LANES=1
stats bench_${dt}_onelane $root/bench.py --dtype $dt $B
unset UDP_POSE_LANES
stats bench_w48_${dt} $root/bench.py --model w48 --dtype $dt $B
stats bench_rsn18_${dt} $root/bench.py --model rsn18 --dtype $dt $B
stats train_bf16 $root/tools/bench_train.py --dtype bf16 --steps 5 --warmup 3
stats train_f32 $root/tools/bench_train.py --dtype f32 --steps 5 --warmup 3
rocprofv3 --pmc FETCH_SIZE -d $run/pf_$dt -o p --output-format csv -- python3 $root/tools/one_forward.py $dt --per-op > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $run/pw_$dt -o p --output-format csv -- python3 $root/tools/one_forward.py $dt --per-op > /dev/null 2>&1
cd $root && UDP_POSE_NO_GROUPS=1 python3 tools/pmc_by_op.py $run/pf_$dt $run/pw_$dt $dt > "$out/${tag}_traffic_${dt}.json"
python3 -c "
import json; d=json.load(open('$out/${tag}_traffic_${dt}.json'))
for k,v in d['classes'].items(): print(k, v['launches'], round(v['hbm_bytes_per_launch']/1e6,1), 'MB/launch', v['traffic_over_algorithmic'])"
tools/pmc_diag.sh gpurun_out/prof_$tag/${tag}_pmc_conv_ws_multi.txt "conv_ws_multi<3>" > /dev/null 2>&1
cat "$out/${tag}_pmc_conv_ws_multi.txt"
