#!/bin/bash
# HBM traffic per launch of the pipeline's kernels from rocprofv3 PMC counters, collected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), --kernel-trace only;
# corrections for gfx950: FETCH_SIZE (KB) counts 64 B per 128-B request of a wide coalesced read -> x2; WRITE_SIZE (KB) as read.
# Writes profiles-style JSON (bytes per clip per kernel) to gpurun_out/pmc_traffic.json; copy to profiles/rNN_pmc_traffic.json (bench.py reads the newest).
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline --no-small-batch --no-parity-mode --no-h2d --no-train-leg"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/traffic_fetch -- $B > /dev/null 2>$R/gpurun_out/traffic_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/traffic_write -- $B > /dev/null 2>$R/gpurun_out/traffic_write.err
# third pass: what actually binds the log-mel kernel -- vector instructions issued and LDS-array cycles (bench.py: roofline_frontend.issue)
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/traffic_issue -- $B > /dev/null 2>$R/gpurun_out/traffic_issue.err
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
# issue-side counters of the log-mel kernel (per clip): vector instructions, LDS instructions, LDS-array busy cycles, clock while profiled
issue = {}
for name in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_SALU", "GRBM_GUI_ACTIVE"):
    issue[name] = collect("traffic_issue", name)
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/traffic_issue/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "logmel" in r["Kernel_Name"]:
            dur["logmel"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in [k for k in out if k.startswith("logmel/")]:
    if k in issue["SQ_INSTS_VALU"]:
        us = sum(dur["logmel"]) / max(len(dur["logmel"]), 1)
        out[k]["issue"] = {"valu_insts_per_clip": issue["SQ_INSTS_VALU"][k] / CLIPS, "lds_insts_per_clip": issue["SQ_INSTS_LDS"].get(k, 0.0) / CLIPS,
                           "salu_insts_per_clip": issue["SQ_INSTS_SALU"].get(k, 0.0) / CLIPS,
                           "lds_array_cycles_per_clip": issue["SQ_LDS_IDX_ACTIVE"].get(k, 0.0) / CLIPS,
                           "lds_conflict_cycles_per_clip": issue["SQ_LDS_BANK_CONFLICT"].get(k, 0.0) / CLIPS,
                           "clock_GHz_while_profiled": issue["GRBM_GUI_ACTIVE"].get(k, 0.0) / 8.0 / (us * 1e3) if us else None,
                           "launch_us_while_profiled": us,
                           "source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE "
                                     "(one pass, --kernel-trace only), scripts/profile_traffic.sh; wave-instructions / LDS-array cycles summed over the chip"}
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
for k, v in out.items():
    print(k, "read %.4g B write %.4g B per launch -> %.0f B per clip" % (v["read_bytes_per_launch"], v["write_bytes_per_launch"], v["bytes_per_clip"]))
P
