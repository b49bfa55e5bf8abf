R=$GRAFT_REPO_ROOT
cd $R
python tools/train_time.py --precision bf16 --steps 10 2>&1 | grep -v amdgpu.ids | tail -2
for i in 1 2; do
timeout 600 python bench.py --workload cfg5 --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d.get('hip_kernel_share'))"
done
nproc; python -c "import os; print(os.cpu_count())"; cat /proc/cpuinfo | grep "model name" | head -1
