#!/bin/bash
# Round-end evidence, collected on the GPU box into gpurun_out/prof_r05/ (copied to profiles/ by hand afterwards):
#   bash tools/collect_profiles.sh [quick]
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$PWD; O=$R/gpurun_out/prof_r05; mkdir -p $O
python bench.py 2>/dev/null | tail -1 > $O/r05_bench_line.json
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/e2e -o e2e -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-stages > /dev/null 2>&1 )
f=$(find $O/e2e -name "*kernel_stats.csv" | head -1); cp "$f" $O/r05_e2e_kernel_stats.csv; python tools/prof_summary.py $O/r05_e2e_kernel_stats.csv 3 45 > $O/r05_e2e_summary.txt
rm -rf $O/e2e
python tools/pmc_bench.py $O/pmc_gemm > /dev/null 2>&1; cp $O/pmc_gemm/gemm_pmc.json $O/r05_gemm_pmc.json; rm -rf $O/pmc_gemm
python tools/unet_bench.py > $O/r05_unet_bench.json 2>&1
python tools/train_bench.py --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/r05_train_bench_line.json
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o tr -- python $R/tools/train_bench.py --steps 2 --warmup 1 > /dev/null 2>&1 )
f=$(find $O/tr -name "*kernel_stats.csv" | head -1); cp "$f" $O/r05_train_kernel_stats.csv; python tools/prof_summary.py $O/r05_train_kernel_stats.csv 3 45 > $O/r05_train_summary.txt
rm -rf $O/tr
python tools/aten_audit.py $O/r05_aten_audit.txt > /dev/null 2>&1
python tools/aten_audit_train.py $O/r05_aten_audit_train.txt > /dev/null 2>&1
python tools/train_calls.py "" 60 > $O/r05_train_calls.txt 2>/dev/null
# the level-0 weight gradient alone, 1 500 launches back to back, with the socket's clock / power sampled beside it
SMI=1 python tools/wgrad_time.py 128 16 8 1500 2>/dev/null | tail -2 > $O/r05_wgrad3_power.txt
if [ "$1" != "quick" ]; then
python tools/pmc_run.py $O/pmc_train "" -- python $R/tools/train_bench.py --steps 2 --warmup 1 > /dev/null 2>&1; cp $O/pmc_train/pmc_summary.json $O/r05_train_pmc_kernels.json; rm -rf $O/pmc_train
python tools/pmc_table.py $O/r05_train_pmc_kernels.json 32 > $O/r05_train_pmc_table.txt
python tools/pmc_run.py $O/pmc_all k_ -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-stages > $O/r05_pmc_kernels.txt 2>&1; cp $O/pmc_all/pmc_summary.json $O/r05_pmc_kernels.json; rm -rf $O/pmc_all
fi
ls -la $O
if [ -f $O/r05_pmc_kernels.json ]; then python tools/pmc_table.py $O/r05_pmc_kernels.json 24 > $O/r05_pmc_table.txt; fi
ls -la $O
