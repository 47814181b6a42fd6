"""Workgroup start / end wall-clock stamps of the fused log-mel kernel (diagnostic build -DMLA_LOGMEL_STAMPS=1): how evenly the
persistent workgroups finish.   MLA_HIP_LIB=build/variants/libmla_fe_stamps.so python scripts/logmel_stamps.py [n_wave]"""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
PKG = "audio-classification-using-a-deep-cnn-combined-with-multi-level-attention_amd"
fe = importlib.import_module(PKG + ".frontend")
W = importlib.import_module(PKG + ".weights")
L = importlib.import_module(PKG + "._lib")
n_wave = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
base = torch.from_numpy(W.waveform(1, 160000, 16)).cuda()
pcm = base.repeat(n_wave // 16, 1).contiguous()
out = torch.empty((n_wave * 10, 96, 64), dtype=torch.bfloat16, device="cuda")
for _ in range(5):
    fe.waveforms_to_examples(pcm, torch.bfloat16, out)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (256 * 10))()
lib = L.lib()
lib.mla_debug_logmel_stamps.argtypes = [ctypes.c_void_p]
assert lib.mla_debug_logmel_stamps(buf) == 0
s = np.array(buf, dtype=np.uint64).reshape(256, 10).astype(np.int64)
loop0 = s[:, 9]
s = s[:, :9]
t0 = s[:, 0].min()
start, end = (s[:, 0] - t0) / 100.0, (s[:, 1:] - t0) / 100.0          # microseconds; end[wg][wave]
span = end.max()
print("n_wave %d: kernel span %.1f us; workgroup start min/max %.1f / %.1f us; wave end min/median/max %.1f / %.1f / %.1f us"
      % (n_wave, span, start.min(), start.max(), end.min(), np.median(end), span))
print("idle at the end, mean over waves: %.1f us = %.1f %% of the span; waves 0-3 end median %.1f us, waves 4-7 %.1f us"
      % ((span - end).mean(), 100 * (span - end).mean() / span, np.median(end[:, :4]), np.median(end[:, 4:])))
print("prologue (kernel entry -> first item fetched): min/median/max %.1f / %.1f / %.1f us" % tuple(np.percentile((loop0 - s[:, 0]) / 100.0, [0, 50, 100])))
print("per-workgroup last wave: min/median/max %.1f / %.1f / %.1f us" % (end.max(1).min(), np.median(end.max(1)), end.max(1).max()))
