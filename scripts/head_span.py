"""Reads a rocprofv3 kernel_trace.csv of scripts/train_prof.py (frozen step) and prints, per step: the CNN forward's span, the span
of everything after it (head forward / backward, Adam), the sum of that part's kernel durations and the GPU idle time inside it.
    python scripts/head_span.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda k: k[0])
starts = [i for i, k in enumerate(ks) if "conv1_patch_kernel" in k[2]]
for a, b in list(zip(starts, starts[1:]))[-4:]:
    step = ks[a:b]
    last_cnn = max(i for i, k in enumerate(step) if "gemm_kernel<mma::bf16_t" in k[2] or "conv3x3_kernel" in k[2])
    head = step[last_cnn + 1:]
    cnn_span = (step[last_cnn][1] - step[0][0]) / 1e3
    head_span = (max(k[1] for k in head) - step[last_cnn][1]) / 1e3
    busy, cur_end, idle = 0.0, step[last_cnn][1], 0.0
    for s, e, _ in head:
        busy += (e - s) / 1e3
        if s > cur_end:
            idle += (s - cur_end) / 1e3
        cur_end = max(cur_end, e)
    print("step: %d kernels, CNN span %.0f us, head span %.0f us (%d kernels, sum of durations %.0f us, GPU idle inside %.0f us); step total %.0f us"
          % (len(step), cnn_span, head_span, len(head), busy, idle, (ks[b][0] - step[0][0]) / 1e3))
