# quick GPU check of the tier-b path: pit tests, three seeds, timeline of one step (-> gpurun_out/)
timeout 600 python -m pytest tests/test_gpu_pit.py -x -q 2>&1 | tail -3
timeout 1200 python scripts/pit_exp.py --seeds ${SEEDS:-1000,1001,1002} --variants ${VARIANTS:-default} ${EXPARGS} > gpurun_out/exp.txt 2>&1; grep "^##" gpurun_out/exp.txt | cut -c1-420
cd /tmp && export TMPDIR=/tmp; rm -rf /root/repo/gpurun_out/tl; rocprofv3 --kernel-trace -d /root/repo/gpurun_out/tl -o tl -- python /root/repo/bench.py --workload ${WL:-c3} --bank 0 --no-cpu-baseline --exact-steps 0 --steps 2 --warmup 1 > /dev/null 2>&1; cd /root/repo; python scripts/rocpd_timeline.py $(find gpurun_out/tl -name "*.db" | head -1) > gpurun_out/timeline.txt; grep -n "unwrap_apply" gpurun_out/timeline.txt | head -2
