#!/bin/bash
# Everything the judged profiles/ artifacts are made from, in one GPU-box call:
#   gpurun --timeout 3000 -- 'bash scripts/make_profiles.sh round6'
# writes gpurun_out/profiles_<tag>/ ; copy what is to be judged into profiles/ and commit.
# `bash scripts/make_profiles.sh round6 pmc` repeats the counter passes (section 3) only.
TAG=${1:-round6}
ONLY=${2:-all}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PY="python"
if [ "$ONLY" = all ]; then
# 1. the bench lines (default flags), fp32 and bf16
timeout 900 $PY $R/bench.py > $O/${TAG}_bench.json 2> $O/bench.err
timeout 300 $PY $R/bench.py --dtype bf16 --no-cpu-baseline > $O/${TAG}_bench_bf16.json 2>> $O/bench.err
# 1b. 400-step soaks of the same two commands (a 20-step mean is 0.8 s of work)
timeout 600 $PY $R/bench.py --steps 400 --no-cpu-baseline > $O/${TAG}_bench_400_steps.json 2>> $O/bench.err
timeout 600 $PY $R/bench.py --steps 400 --dtype bf16 --no-cpu-baseline > $O/${TAG}_bench_bf16_400_steps.json 2>> $O/bench.err
timeout 600 $PY $R/bench.py --steps 2000 --timing-only > $O/${TAG}_bench_2000_steps.json 2>> $O/bench.err
timeout 600 $PY $R/bench.py --steps 2000 --timing-only --dtype bf16 > $O/${TAG}_bench_bf16_2000_steps.json 2>> $O/bench.err
# 2. kernel traces: the default (concurrent graphs) command and the serial one whose averages the roofline object quotes
rm -rf /tmp/kt1 /tmp/kt2
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt1 -- $PY $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.json 2>/dev/null
$PY $R/scripts/prof_summary.py $(ls /tmp/kt1/*/*.db | head -1) > $O/${TAG}_bench_kernel_trace.txt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt2 -- $PY $R/bench.py --serial --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_serial_under_rocprof.json 2>/dev/null
$PY $R/scripts/prof_summary.py $(ls /tmp/kt2/*/*.db | head -1) > $O/${TAG}_bench_serial_kernel_trace.txt
$PY $R/scripts/torch_share.py /tmp/kt2 $O/${TAG}_torch_share.json > /dev/null
# 2b. the generator step alone (eager, one stream: the kernel composition of the iteration's backbone) and the device timeline
#     of the pipelined loop (phase lengths, end of every line)
rm -rf /tmp/kt3
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt3 -- $PY $R/scripts/g_step_trace.py 20 > /dev/null 2>&1
$PY $R/scripts/prof_summary.py $(ls /tmp/kt3/*/*.db | head -1) > $O/${TAG}_generator_step_kernel_trace.txt
timeout 300 $PY $R/scripts/dev/iter_gpu_timeline.py 2>/dev/null | tail -26 > $O/${TAG}_iteration_timeline.txt
fi
# 3. PMC passes (each in its own run, --kernel-trace only).  A pass that leaves no database (rocprofv3 has died at exit on
#    this pool now and then) is repeated, up to three times.
pmc_pass() {   # pmc_pass <out dir> <stdout file> <bench flags...> -- <counters...>
  local out=$1 line=$2; shift 2
  local flags=(); while [ "$1" != "--" ]; do flags+=("$1"); shift; done; shift
  for try in 1 2 3; do
    rm -rf $out
    timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $out -- $PY $R/bench.py "${flags[@]}" > $line 2>/dev/null
    if ls $out/*/*.db > /dev/null 2>&1; then return 0; fi
    echo "pmc pass $out: no database (try $try)" >> $O/bench.err
  done
  return 1
}
CMD="bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial"
pmc_pass /tmp/pf /tmp/pf_line.json --steps 2 --warmup 1 --no-cpu-baseline --serial -- FETCH_SIZE
pmc_pass /tmp/pw /dev/null --steps 2 --warmup 1 --no-cpu-baseline --serial -- WRITE_SIZE
$PY $R/scripts/pmc_traffic_json.py /tmp/pf /tmp/pw $O/${TAG}_pmc_traffic.json "python $CMD" > /dev/null
$PY $R/scripts/pmc_traffic_by_kernel.py /tmp/pf /tmp/pw /tmp/pf_line.json $O/${TAG}_pmc_traffic_by_kernel.txt $O/${TAG}_pmc_traffic_by_kernel.json > /dev/null
pmc_pass /tmp/pm /dev/null --steps 2 --warmup 1 --no-cpu-baseline --serial -- SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES
$PY $R/scripts/pmc_mfma.py /tmp/pm $O/${TAG}_pmc_mfma.json > $O/${TAG}_pmc_mfma.txt
# 3a. one table per kernel: launches | us | MFMA-busy | t_mfma | t_rest | traffic ratio (the overlap claim, checkable in one file)
$PY $R/scripts/overlap_table.py /tmp/pm $O/${TAG}_pmc_traffic_by_kernel.json $O/${TAG}_overlap_table.txt > /dev/null
# 3b. the same counters for the bf16 path (configs[2]'s dtype)
CMDB="bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial --dtype bf16"
pmc_pass /tmp/pfb /tmp/pfb_line.json --steps 2 --warmup 1 --no-cpu-baseline --serial --dtype bf16 -- FETCH_SIZE
pmc_pass /tmp/pwb /dev/null --steps 2 --warmup 1 --no-cpu-baseline --serial --dtype bf16 -- WRITE_SIZE
$PY $R/scripts/pmc_traffic_json.py /tmp/pfb /tmp/pwb $O/${TAG}_pmc_traffic_bf16.json "python $CMDB" > /dev/null
$PY $R/scripts/pmc_traffic_by_kernel.py /tmp/pfb /tmp/pwb /tmp/pfb_line.json $O/${TAG}_pmc_traffic_by_kernel_bf16.txt $O/${TAG}_pmc_traffic_by_kernel_bf16.json > /dev/null
pmc_pass /tmp/pmb /dev/null --steps 2 --warmup 1 --no-cpu-baseline --serial --dtype bf16 -- SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES
$PY $R/scripts/pmc_mfma.py /tmp/pmb $O/${TAG}_pmc_mfma_bf16.json > $O/${TAG}_pmc_mfma_bf16.txt
$PY $R/scripts/overlap_table.py /tmp/pmb $O/${TAG}_pmc_traffic_by_kernel_bf16.json $O/${TAG}_overlap_table_bf16.txt > /dev/null
if [ "$ONLY" = all ]; then
# 4. per-shape tables
cd $R
timeout 600 $PY scripts/conv_shapes_bench.py 16 f32 > $O/${TAG}_conv_shapes.txt 2>/dev/null
timeout 600 $PY scripts/conv_shapes_bench.py 16 bf16 > $O/${TAG}_conv_shapes_bf16.txt 2>/dev/null
timeout 300 $PY scripts/ew_shapes_bench.py > $O/${TAG}_elementwise_shapes.txt 2>/dev/null
timeout 300 $PY scripts/predict_latency.py > $O/${TAG}_predict_latency.txt 2>/dev/null
timeout 900 $PY scripts/bench_configs.py > $O/${TAG}_secondary_configs.json 2>/dev/null
timeout 600 $PY scripts/wgrad_bench.py 16 > $O/${TAG}_wgrad_shapes.txt 2>/dev/null
timeout 300 $PY scripts/dev/wino_bench.py > $O/${TAG}_winograd_f4x4.txt 2>/dev/null
fi
ls -la $O
