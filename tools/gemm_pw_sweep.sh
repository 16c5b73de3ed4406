R=$GRAFT_REPO_ROOT
for pw in 1 3 5 8; do
HTN_GEMM_PW=$pw python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r03_pw$pw.json 2> $R/gpurun_out/r03_pw$pw.err
python -c "
import json
d=json.loads(open('$R/gpurun_out/r03_pw$pw.json').read().strip().splitlines()[-1])
print('pw=$pw', d['value'], d['roofline']['avg_launch_us'], round(d['roofline']['frac'],4), d['stage_s_per_sweep']['lanczos'])"
done
HTN_GEMM_GROUPS=0 python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r03_pwoff.json 2> $R/gpurun_out/r03_pwoff.err
python -c "
import json
d=json.loads(open('$R/gpurun_out/r03_pwoff.json').read().strip().splitlines()[-1])
print('groups off', d['value'], d['roofline']['avg_launch_us'], round(d['roofline']['frac'],4), d['stage_s_per_sweep']['lanczos'])"
