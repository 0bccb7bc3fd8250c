cd $GRAFT_REPO_ROOT
python bench.py --batch 1 --state-dim 203 --features 30 --steps 200 --warmup 20 --no-cpu-baseline --no-mixed --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('profiled', d['ms_per_step'], d['stage_ms']); print(d.get('dropin'))"
python bench.py --batch 1 --state-dim 203 --features 30 --steps 200 --warmup 20 --no-cpu-baseline --no-mixed --no-configs --no-dropin --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unprofiled', d['ms_per_step'])"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_b1 -o b1 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batch 1 --state-dim 203 --features 30 --steps 50 --warmup 5 --no-cpu-baseline --no-mixed --no-configs --no-dropin --no-profile > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/prof_b1/**/b1_kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 3 steps
tail=rows[-24:]
t0=int(tail[0]['Start_Timestamp'])
for r in tail:
    print(r['Kernel_Name'][:60].ljust(60), (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
PY
