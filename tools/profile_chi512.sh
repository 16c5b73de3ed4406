set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r03c512 -o c512 --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --chi 512 > $R/gpurun_out/r03c512_bench.json 2> $R/gpurun_out/r03c512.err
KT=$(find $R/gpurun_out/prof_r03c512 -name "*kernel_trace.csv")
python $R/tools/trace_tail.py $KT 0.12 > $R/gpurun_out/r03c512_trace_tail.txt
python $R/tools/bond_timeline.py $KT 40 > $R/gpurun_out/r03c512_bond_timeline.txt
rm -f $KT
