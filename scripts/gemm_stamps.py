"""Diagnostic: per-phase cycles of one K stage of the FC1 GEMM from s_memtime stamps (build: scripts/build_variant.py gstamps
--only=gemm.hip -DMLA_GEMM_STAMPS=1 [...]; run with MLA_HIP_LIB=build/variants/libmla_gstamps.so)."""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
PKG = "audio-classification-using-a-deep-cnn-combined-with-multi-level-attention_amd"
ops = importlib.import_module(PKG + ".ops"); L = importlib.import_module(PKG + "._lib")
M, N, K = 10240, 4096, 12288
a = (torch.rand((M, K), device="cuda") - 0.3).clamp_min(0).to(torch.bfloat16)
w = ((torch.rand((N, K), device="cuda") - 0.5) * (6.0 / K) ** 0.5).to(torch.bfloat16)
b = torch.zeros(N, device="cuda")
for _ in range(300):
    ops.linear(a, w, b, relu=True, out_dtype=torch.bfloat16)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 2 * 8))()
assert L.lib().mla_debug_gemm_stamps(buf) == 0
early = ["rd0 issue", "dma issue", "mm0", "rd1", "mm1", "to vmcnt(0)", "barrier"]
late = ["mm(prev ks1)", "dma issue", "rd0+mm0", "rd1 issue", "rd1 wait", "to vmcnt(0)", "barrier"]
for blk in range(8):
    for wv in range(2):
        s = [buf[(blk * 2 + wv) * 8 + k] for k in range(8)]
        if s[0] == 0:
            continue
        d = [s[k + 1] - s[k] for k in range(7)]
        nm = late if (wv and os.environ.get("STAGGERED")) else early
        print("wg %d wave %d: " % (blk, wv * 4) + ", ".join("%s %d" % (n_, v) for n_, v in zip(nm, d)) + " | stage total %d (t0 %d)" % (s[7] - s[0], s[0] % 100000))
