#!/bin/bash
# Round-end evidence, collected on the GPU box into gpurun_out/prof_r04/ (copied to profiles/ by hand afterwards):
#   bash tools/collect_profiles.sh
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$PWD; O=$R/gpurun_out/prof_r04; mkdir -p $O
python bench.py 2>/dev/null | tail -1 > $O/r04_bench_line.json
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/e2e -o e2e -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-stages > /dev/null 2>&1 )
f=$(find $O/e2e -name "*kernel_stats.csv" | head -1); cp "$f" $O/r04_e2e_kernel_stats.csv; python tools/prof_summary.py $O/r04_e2e_kernel_stats.csv 3 40 > $O/r04_e2e_summary.txt
rm -rf $O/e2e
python tools/pmc_bench.py $O/pmc_gemm > /dev/null 2>&1; cp $O/pmc_gemm/gemm_pmc.json $O/r04_gemm_pmc.json; rm -rf $O/pmc_gemm
python tools/pmc_run.py $O/pmc_all k_ -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-stages > $O/r04_pmc_kernels.txt 2>&1; cp $O/pmc_all/pmc_summary.json $O/r04_pmc_kernels.json; rm -rf $O/pmc_all
hipcc --offload-arch=gfx950 -O3 tools/mfma_skeleton.cpp -o /tmp/sk 2>/dev/null && /tmp/sk > $O/r04_skeleton.txt
( hipcc --offload-arch=gfx950 -O3 -w tools/valu_rate.cpp -o /tmp/vr && /tmp/vr; hipcc --offload-arch=gfx950 -O3 -w tools/coissue_probe.cpp -o /tmp/ci && /tmp/ci ) > $O/r04_valu_probes.txt 2>&1
python tools/gemm_probe.py abltrace > $O/r04_abltrace.txt 2>&1
SEMABS_TUNE_LIB=0 python tools/gemm_probe.py deep > $O/r04_gemm_ab.txt 2>&1
python tools/gemm_probe.py v3trace > $O/r04_persist_trace.txt 2>&1
python tools/unet_bench.py > $O/r04_unet_bench.json 2>&1
python tools/train_bench.py --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/r04_train_bench_line.json
ls -la $O
