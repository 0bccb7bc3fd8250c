#!/bin/bash
# SQ counter passes (wave cycles / waits / instruction mix / LDS conflicts) of the bench command, one counter group per run;
# summary printed per kernel. DESIGN.md 3.2 quotes these for trsm_lds_f64_kernel<10,4>.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_sq; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-mixed --no-parity-check --no-configs --no-dropin ${BENCH_ARGS:-}"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES -d $O -o p1 --output-format csv -- $BENCH > $O/p1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O -o p2 --output-format csv -- $BENCH > $O/p2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH -d $O -o p3 --output-format csv -- $BENCH > $O/p3.log 2>&1
cd $R
python - <<'PY'
import csv, collections, glob, os
O="gpurun_out/pmc_sq"
for p in ("p1","p2","p3"):
    f=glob.glob(f"{O}/**/{p}_counter_collection.csv", recursive=True)
    if not f: print(p,"no file", open(f"{O}/{p}.log").read()[-600:]); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"].split("(")[0][-50:]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
    for k,v in agg.items():
        if "trsm" in k or "ell_tile" in k or "chol" in k or "fused" in k or "propagate" in k:
            print(p, k, {c: round(val/max(1,n[(k,c)])/1e6,2) for c,val in v.items()}, "(M per dispatch)")
PY
