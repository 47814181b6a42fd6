"""Aggregate rocprofv3 --pmc passes (counter_collection.csv + kernel_trace.csv under the given directories) into one
table per kernel: mean counter value per launch and mean duration. Usage: pmc_summary.py OUT.txt DIR [DIR ...]
Kernel names are shortened to the template configuration; only conv / gemm / logmel kernels of libmla_hip.so are kept."""
import collections, csv, glob, re, sys


def short(name):
    m = re.search(r"(conv3x3_kernel|conv1_kernel|conv1_patch_kernel|gemm_ring_kernel|gemm_kernel|logmel_kernel|logmel_dyn_kernel|wgrad_bf16_kernel|wgrad_kernel)<(.*?)>\s*\(", name)
    if not m:
        return None
    cfg = m.group(2).replace("(anonymous namespace)::", "").replace("mma::bf16_t", "bf16").replace("__hip_bfloat16", "bf16")
    return "%s<%s>" % (m.group(1), cfg.rstrip(" >") )


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k:
                    ctr[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k:
                    dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    with open(out, "w") as fo:
        for k in sorted(ctr):
            c = {n: sum(v) / len(v) for n, v in ctr[k].items()}
            n = max(len(v) for v in ctr[k].values())
            us = sum(dur[k]) / max(len(dur[k]), 1)
            fo.write("%s\n  launches %d  mean_us(profiled) %.1f\n" % (k, n, us))
            for name in sorted(c):
                fo.write("  %-28s %.6g\n" % (name, c[name]))
            g = c.get("GRBM_GUI_ACTIVE")
            if g:
                cyc = g / 8.0                                   # GRBM_GUI_ACTIVE sums over the 8 XCDs
                fo.write("  -> kernel cycles %.4g, clock %.3f GHz\n" % (cyc, cyc / (us * 1e3) if us else 0))
                if "SQ_VALU_MFMA_BUSY_CYCLES" in c:             # counts cycles, summed over 256 CUs x 4 SIMDs
                    fo.write("  -> MFMA pipe busy %.1f %% of SIMD-cycles\n" % (100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)))
                if "SQ_BUSY_CYCLES" in c:
                    fo.write("  -> SQ busy / kernel cycles (per SE sum) %.3g\n" % (c["SQ_BUSY_CYCLES"] / cyc))
            if "SQ_WAVE_CYCLES" in c:
                w = c["SQ_WAVE_CYCLES"]
                for n2 in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
                    if n2 in c:
                        fo.write("  -> %s / SQ_WAVE_CYCLES %.1f %%\n" % (n2, 100 * c[n2] / w))
            if "SQ_LDS_IDX_ACTIVE" in c and "SQ_LDS_BANK_CONFLICT" in c and c["SQ_LDS_IDX_ACTIVE"]:
                fo.write("  -> LDS bank-conflict cycles / LDS active %.2f %%\n" % (100 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]))
            fo.write("\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
