#!/bin/bash
# round-1 recipe: MFMA-pipe / wave-state counters of the conv micro-benchmark, one --pmc pass per layer (round 2: scripts/profile_mfma_counters.sh)
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for L in 3 4 5; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/cpmc_$L -- python3 $R/scripts/conv_bench.py 10240 bf16 3 $L > /dev/null 2>&1
done
cd $R
python3 - <<'P'
import csv, glob, collections
for L in (3,4,5):
    d=f"cpmc_{L}"; dur=[]
    for f in glob.glob(f"gpurun_out/{d}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv3x3" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "conv3x3" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        print("layer", L, "median_us", sorted(dur)[len(dur)//2], {k: acc[k]/n[k] for k in acc})
P
