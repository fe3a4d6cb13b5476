cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
timeout -k 10 600 python3 tools/long_reads.py --no-seed --routes auto 2>&1 | grep -v "^#\|amdgpu.ids"
timeout -k 10 300 python3 bench.py --no-e2e --cpu-seconds 3 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','verified_vs_oracle','verified_reads')}, d['roofline']['kernel_ms_avg'], d.get('lone_call'), d.get('host_streamed_pinned',{}).get('value'), d.get('secondary_error'))"
