R=$GRAFT_REPO_ROOT
for g in 0 1; do
HTN_GEMM_GROUPS=$g python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r03_grp$g.json 2> $R/gpurun_out/r03_grp$g.err
python -c "
import json
d=json.loads(open('$R/gpurun_out/r03_grp$g.json').read().strip().splitlines()[-1])
print('groups=$g', d['value'], d['roofline']['avg_launch_us'], round(d['roofline']['frac'],4), d['stage_s_per_sweep']['lanczos'])"
done
cd /tmp && export TMPDIR=/tmp
for g in 0 1; do
HTN_GEMM_GROUPS=$g rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_grouped_gemm_z" -d $R/gpurun_out/pmc_grp$g -o f --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_grp$g.err
echo groups=$g; python $R/tools/pmc_summary.py $(find $R/gpurun_out/pmc_grp$g -name "*counter_collection.csv") FETCH_SIZE 0.1
rm -rf $R/gpurun_out/pmc_grp$g
done
