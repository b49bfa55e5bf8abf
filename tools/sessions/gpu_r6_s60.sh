R=$GRAFT_REPO_ROOT; cd $R
for i in $(seq 1 6); do
timeout 600 python -m pytest tests/test_gpu_bench.py -x -q -k "two_ranks_on_one_device" > /tmp/t_$i.log 2>&1; rc=$?
echo "run $i rc=$rc $(tail -1 /tmp/t_$i.log)"
if [ $rc -ne 0 ]; then grep -v "^$" /tmp/t_$i.log | tail -80; fi
done
