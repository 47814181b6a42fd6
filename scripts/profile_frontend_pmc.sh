#!/bin/bash
# rocprofv3 --pmc passes on the fused log-mel kernel alone (target scripts/fe_prof.py): HBM traffic, wave states, LDS array (profiles/r01_frontend_counters.txt)
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/fe_fetch -- python3 $R/scripts/fe_prof.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/fe_write -- python3 $R/scripts/fe_prof.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/fe_sq1 -- python3 $R/scripts/fe_prof.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/fe_sq2 -- python3 $R/scripts/fe_prof.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bench_stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-small-batch > $R/gpurun_out/bench_prof_line.json 2>/dev/null
cd $R
python3 bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/bench_line.json
python3 scripts/fe_bench.py > gpurun_out/fe_bench.txt 2>/dev/null
python3 - <<'P'
import csv, glob, collections
for d in ("fe_fetch","fe_write","fe_sq1","fe_sq2"):
    dur=[]
    for f in glob.glob(f"gpurun_out/{d}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "logmel" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "logmel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in acc: print(d, k, acc[k] / n[k], "launches", n[k], "median_us", sorted(dur)[len(dur)//2])
P
