R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python bench.py --workload cfg5 --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline --verbose 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['device_ms_per_step'], d['hip_kernel_share'], d['whole_step'])
for k,v in list(d['per_kernel_ms_per_step'].items())[:40]: print(f'{v:8.3f}  {k[:90]}')
print('--- other')
for k,v in d['other_kernels_ms_per_step'].items(): print(v, k[:110])"
