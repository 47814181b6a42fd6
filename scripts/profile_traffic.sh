#!/bin/bash
# HBM traffic per launch of the pipeline's kernels from rocprofv3 PMC counters, collected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), --kernel-trace only;
# corrections for gfx950: FETCH_SIZE (KB) counts 64 B per 128-B request of a wide coalesced read -> x2; WRITE_SIZE (KB) as read.
# Writes profiles-style JSON (bytes per clip per kernel) to gpurun_out/pmc_traffic.json; copy to profiles/r02_pmc_traffic.json.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline --no-small-batch --no-parity-mode --no-h2d --no-train-leg"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/traffic_fetch -- $B > /dev/null 2>$R/gpurun_out/traffic_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/traffic_write -- $B > /dev/null 2>$R/gpurun_out/traffic_write.err
cd $R
python3 - <<'P'
import collections, csv, glob, json, re
CLIPS = 10240
NAMES = {"Cfg<bf16, 64, 128, 48, 32, true, ": "conv2", "Cfg<bf16, 128, 256, 24, 16, false, 4": "conv3", "Cfg<bf16, 256, 256, 24, 16, true, 4": "conv4",
         "Cfg<bf16, 256, 512, 12, 8, false, 4": "conv5", "Cfg<bf16, 512, 512, 12, 8, true, 4": "conv6"}
def collect(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("mma::bf16_t", "bf16")
            if "logmel_dyn_kernel" in k or "logmel_kernel" in k:
                acc["logmel/bf16" if "__hip_bfloat16" in k or "bf16" in k.split("_kernel")[1][:40] else "logmel/f32"].append(float(r["Counter_Value"]))
            for pat, name in NAMES.items():
                if "conv3x3_kernel<" + pat in k:
                    acc[name + "/bf16"].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
fetch, write = collect("traffic_fetch", "FETCH_SIZE"), collect("traffic_write", "WRITE_SIZE")
out = {}
for k in sorted(fetch):
    rd, wr = fetch[k] * 1024 * 2, write.get(k, 0.0) * 1024
    out[k] = {"bytes_per_clip": (rd + wr) / CLIPS, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "clips_per_launch": CLIPS,
              "source": "rocprofv3 --pmc FETCH_SIZE (KB x 1024 x 2: gfx950 tallies 128-B requests at 64 B) + --pmc WRITE_SIZE (KB x 1024), separate passes, "
                        "scripts/profile_traffic.sh, mean over the launches of bench.py --steps 3 --warmup 1"}
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
for k, v in out.items():
    print(k, "read %.4g B write %.4g B per launch -> %.0f B per clip" % (v["read_bytes_per_launch"], v["write_bytes_per_launch"], v["bytes_per_clip"]))
P
