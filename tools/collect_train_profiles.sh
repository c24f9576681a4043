set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$PWD; O=$R/gpurun_out/prof_r05; mkdir -p $O
python tools/train_bench.py --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/r05_train_bench_line.json
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o tr -- python $R/tools/train_bench.py --steps 2 --warmup 1 > /dev/null 2>&1 )
f=$(find $O/tr -name "*kernel_stats.csv" | head -1); cp "$f" $O/r05_train_kernel_stats.csv; python tools/prof_summary.py $O/r05_train_kernel_stats.csv 3 45 > $O/r05_train_summary.txt
rm -rf $O/tr
python tools/aten_audit_train.py $O/r05_aten_audit_train.txt > /dev/null 2>&1
python tools/train_calls.py "" 60 > $O/r05_train_calls.txt 2>/dev/null
python tools/pmc_run.py $O/pmc_train "" -- python $R/tools/train_bench.py --steps 2 --warmup 1 > /dev/null 2>&1; cp $O/pmc_train/pmc_summary.json $O/r05_train_pmc_kernels.json; rm -rf $O/pmc_train
python tools/pmc_table.py $O/r05_train_pmc_kernels.json 32 > $O/r05_train_pmc_table.txt
python bench.py --workload train --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/r05_bench_train_workload_line.json
