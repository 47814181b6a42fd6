"""In-tree build of libmla_hip.so (hipcc, gfx950 only).

``python -m <pkg>.build`` or ``__graft_entry__.build()``. The shared object is written next
to this file so that it travels with the source tree to the GPU box; objects are cached
under ``csrc/_obj`` and rebuilt when a source or header is newer.
"""

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libmla_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-value", "-Wno-pass-failed",
         "-I", INCLUDE, "-I", CSRC]
# logmel.hip: hipcc's SLP vectoriser packs the FFT butterflies into v_pk_*_f32 and pays for it with ~100
# v_mov register shuffles per frame; packed f32 has no throughput advantage on CDNA4 (MI355X_MICROARCH.md,
# per-instruction constants). Measured on the previous kernel: 1.81 -> 1.56 ms per 40 960 clips.
PER_FILE_FLAGS = {"logmel.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newest_header():
    ts = [os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h")]
    ts.append(os.path.getmtime(os.path.join(INCLUDE, "mla_hip.h")))
    return max(ts)


def _compile(src, force, hdr_ts):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    path = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(path), hdr_ts)):
        return obj, False
    subprocess.run([HIPCC] + FLAGS + PER_FILE_FLAGS.get(src, []) + ["-c", path, "-o", obj], check=True)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hdr_ts = _newest_header()
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        results = list(ex.map(lambda s: _compile(s, force, hdr_ts), sources()))
    objs = [o for o, _ in results]
    if any(changed for _, changed in results) or not os.path.exists(LIB):
        link = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-ldl"]
        # no -lrccl: csrc/collective.hip binds the RCCL already loaded in the process (the build PyTorch ships, or the C
        # host's own) with dlopen(RTLD_NOLOAD) at first use; linking ROCm's here would put a second RCCL beside it
        subprocess.run(link, check=True)
        if verbose:
            print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
