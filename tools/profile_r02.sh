set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02c -o r02c --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02c_bench_under_rocprof.json 2> $R/gpurun_out/r02c.err
KT=$(find $R/gpurun_out/prof_r02c -name "*kernel_trace.csv")
python $R/tools/trace_tail.py $KT 0.12 > $R/gpurun_out/r02c_trace_tail_steady_state.txt
python $R/tools/bond_timeline.py $KT 40 > $R/gpurun_out/r02c_bond_timeline.txt
rm -f $KT
echo trace done
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_grouped_gemm_z" -d $R/gpurun_out/pmc_r02_fetch -o f --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc1.err
python $R/tools/pmc_summary.py $(find $R/gpurun_out/pmc_r02_fetch -name "*counter_collection.csv") FETCH_SIZE 0.1 > $R/gpurun_out/r02_pmc_gemm.txt
rm -rf $R/gpurun_out/pmc_r02_fetch
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_grouped_gemm_z" -d $R/gpurun_out/pmc_r02_write -o w --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc2.err
python $R/tools/pmc_summary.py $(find $R/gpurun_out/pmc_r02_write -name "*counter_collection.csv") WRITE_SIZE 0.1 >> $R/gpurun_out/r02_pmc_gemm.txt
rm -rf $R/gpurun_out/pmc_r02_write
echo write done
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES -d $R/gpurun_out/pmc_r02_sq -o s --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc3.err
python $R/tools/pmc_by_kernel.py $(find $R/gpurun_out/pmc_r02_sq -name "*counter_collection.csv") 0.1 > $R/gpurun_out/r02_pmc_sq_by_kernel.txt
rm -rf $R/gpurun_out/pmc_r02_sq
echo sq done
