"""Interleaved A/B of library variants in ONE gpurun call (same device): runs `rounds` rounds of the given micro-benchmark command per
variant, alternating, and prints per-variant medians.   python scripts/ab_bench.py <rounds> <libA> <libB> [...] -- <cmd...>
Each <lib> is a path to a libmla_hip.so build ("main" = the in-tree one)."""
import json, os, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
i = sys.argv.index("--")
rounds, libs, cmd = int(sys.argv[1]), sys.argv[2:i], sys.argv[i + 1:]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ)
        if lib != "main":
            env["MLA_HIP_LIB"] = os.path.join(ROOT, lib)
        out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=ROOT).stdout.decode()
        for line in out.splitlines():
            if line.startswith("{"):
                d = json.loads(line)
                key = str(d.get("layer", d.get("K", d.get("n_wave", ""))))
                if "N" in d: key += "x%s" % d["N"]
                if "n_samples" in d: key += "x%s/%s" % (d["n_samples"], d.get("out", ""))
                res[lib][key].append(d.get("TFLOPs", d.get("GBps")))
for key in next(iter(res.values())):
    print(key, {lib: "%.1f (min %.1f max %.1f)" % (sorted(res[lib][key])[len(res[lib][key]) // 2], min(res[lib][key]), max(res[lib][key])) for lib in libs})
