"""Diagnostic: tile-boundary cycles of the conv kernel (flush of the carried-over burst, epilogue) from s_memtime stamps
(build: scripts/build_variant.py tstamps --only=conv.hip -DMLA_CONV_STAMPS=2 -DMLA_STAMP_TAP=4 -DMLA_STAMP_CHUNK=5; run with MLA_HIP_LIB)."""
import ctypes, importlib, os, sys
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
import torch
PKG = "audio-classification-using-a-deep-cnn-combined-with-multi-level-attention_amd"
ops = importlib.import_module(PKG + ".ops"); W = importlib.import_module(PKG + ".weights"); L = importlib.import_module(PKG + "._lib")
layer = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = 10240
(h, w, cin), (ho, wo, cout) = ops.CONV_SHAPES[layer]
x = (torch.rand((n, h, w, cin), device="cuda") * 2 - 0.5).clamp_min(0).to(torch.bfloat16)
wt = torch.from_numpy(W.uniform(1, layer, cout * cin * 9)).reshape(cout, cin, 3, 3).cuda() * (6.0 / (9 * cin)) ** 0.5
wp = ops.repack_conv_weight(wt, torch.bfloat16); b = torch.zeros(cout, device="cuda")
for _ in range(200): ops.conv(layer, x, wp, b)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
assert L.lib().mla_debug_conv_stamps(buf) == 0
for blk in range(8):
    for wv in range(2):
        s = [buf[(blk * 2 + wv) * 8 + k] for k in range(8)]
        print(blk, wv, s[:3]) if not s[0] else print("wg %d wave %d: flush-mm %d, epilogue %d  (t0 %d)" % (blk, wv * 4, s[1] - s[0], s[2] - s[1], s[0] % 1000000))
