"""Register / spill / LDS report of the kernels of one source (hipcc -Rpass-analysis=kernel-resource-usage).
    python scripts/kernel_resources.py conv.hip [name-filter] [extra -D flags]"""
import importlib, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("audio-classification-using-a-deep-cnn-combined-with-multi-level-attention_amd.build")
src, filt, flags = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else ""), sys.argv[3:]
r = subprocess.run([b.HIPCC] + b.FLAGS + b.PER_FILE_FLAGS.get(src, []) + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(b.CSRC, src), "-o", "/tmp/_ru.o"],
                   capture_output=True, text=True)
for blk in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
    name = blk.split("\n")[0].strip()
    dm = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").replace("mma::", "")
    if filt not in dm:
        continue
    g = lambda k: (re.search(k + r": (\d+)", blk) or [None, "?"])[1]
    print("%-120s VGPR %s AGPR %s spill %s scratch %s LDS %s occ %s" % (dm[:120], g("VGPRs"), g("AGPRs"), g("VGPR Spill"), g("ScratchSize \[bytes/lane\]"), g("LDS Size \[bytes/block\]"), g("Occupancy \[waves/SIMD\]")))
