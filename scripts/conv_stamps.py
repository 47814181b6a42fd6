"""Diagnostic: per-phase cycles of one tap of the conv kernel from s_memtime stamps (build: scripts/build_variant.py stamps
--only=conv.hip -DMLA_CONV_STAMPS=1 -DMLA_CONV_STAGGER=0; run with MLA_HIP_LIB=build/variants/libmla_stamps.so)."""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
PKG = "audio-classification-using-a-deep-cnn-combined-with-multi-level-attention_amd"
ops = importlib.import_module(PKG + ".ops"); W = importlib.import_module(PKG + ".weights"); L = importlib.import_module(PKG + "._lib")
layer = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = 10240
(h, w, cin), (ho, wo, cout) = ops.CONV_SHAPES[layer]
x = (torch.rand((n, h, w, cin), device="cuda") * 2 - 0.5).clamp_min(0).to(torch.bfloat16)
wt = torch.from_numpy(W.uniform(1, layer, cout * cin * 9)).reshape(cout, cin, 3, 3).cuda() * (6.0 / (9 * cin)) ** 0.5
wp = ops.repack_conv_weight(wt, torch.bfloat16)
b = torch.zeros(cout, device="cuda")
for _ in range(300):                                 # warm clocks: ~1 s of back-to-back launches
    ops.conv(layer, x, wp, b)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 2 * 8))()
assert L.lib().mla_debug_conv_stamps(buf) == 0
names = ["rd0 issue", "rd0 wait", "mm0", "rd1 issue+wait", "mm1", "to vmcnt(0)", "barrier"]
late_names = ["mm(prev ks1)", "rd0 issue+wait", "mm0", "rd1 issue", "rd1 wait", "to vmcnt(0)", "barrier"]
for blk in range(8):
    for wv in range(2):
        s = [buf[(blk * 2 + wv) * 8 + k] for k in range(8)]
        if s[0] == 0:
            continue
        d = [s[k + 1] - s[k] for k in range(7)]
        print("wg %d wave %d: " % (blk, wv * 4) + ", ".join("%s %d" % (nm, v) for nm, v in zip(late_names if (wv and os.environ.get("STAGGERED")) else names, d)) + " | tap total %d (t0 %d)" % (s[7] - s[0], s[0] % 100000))
