# round-3 profiles: kernel trace of the bench command (steady-state tail + one centre-bond timeline), then the PMC passes of
# the apply kernel (each in its own run, program directly after `--`).  Usage on the GPU box: bash tools/profile_r03.sh [tag]
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o $TAG --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}.err
KT=$(find $R/gpurun_out/prof_$TAG -name "*kernel_trace.csv")
python $R/tools/trace_tail.py $KT 0.12 > $R/gpurun_out/${TAG}_trace_tail_steady_state.txt
python $R/tools/bond_timeline.py $KT 40 > $R/gpurun_out/${TAG}_bond_timeline.txt
cp $(find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv") $R/gpurun_out/${TAG}_kernel_stats.csv
rm -f $KT
echo trace done
if [ "$2" = "pmc" ]; then
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_grouped_gemm_z" -d $R/gpurun_out/pmc_${TAG}_fetch -o f --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc1.err
python $R/tools/pmc_summary.py $(find $R/gpurun_out/pmc_${TAG}_fetch -name "*counter_collection.csv") FETCH_SIZE 0.1 > $R/gpurun_out/${TAG}_pmc_gemm.txt
rm -rf $R/gpurun_out/pmc_${TAG}_fetch
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_grouped_gemm_z" -d $R/gpurun_out/pmc_${TAG}_write -o w --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc2.err
python $R/tools/pmc_summary.py $(find $R/gpurun_out/pmc_${TAG}_write -name "*counter_collection.csv") WRITE_SIZE 0.1 >> $R/gpurun_out/${TAG}_pmc_gemm.txt
rm -rf $R/gpurun_out/pmc_${TAG}_write
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES -d $R/gpurun_out/pmc_${TAG}_sq -o s --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc3.err
python $R/tools/pmc_by_kernel.py $(find $R/gpurun_out/pmc_${TAG}_sq -name "*counter_collection.csv") 0.1 > $R/gpurun_out/${TAG}_pmc_sq_by_kernel.txt
rm -rf $R/gpurun_out/pmc_${TAG}_sq
echo pmc done
fi
