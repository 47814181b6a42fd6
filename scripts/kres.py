"""Compile one .hip for gfx950 and print per-kernel register / scratch / LDS usage."""
import re, subprocess, sys
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I/root/repo/include", "-c",
       "-Rpass-analysis=kernel-resource-usage", src, "-o", "/tmp/kres.o"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
for line in out.splitlines():
    if "error" in line:
        print(line)
    m = re.search(r"remark: (?:.*?:\d+:\d+: )?\s*(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        if cur:
            print(cur)
        name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = {"fn": name[:110]}
    else:
        cur[k.split()[0] if k != "VGPRs Spill" else "spill"] = v
if cur:
    print(cur)
