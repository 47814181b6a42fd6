"""A/B builds: compile the library with extra -D flags into build/variants/libmla_<name>.so (objects under build/variants/<name>/).
Select it at run time with MLA_HIP_LIB=<path> (the ctypes binding honours it). Deltas are only meaningful between variants run
interleaved in ONE process / one gpurun call on one device (MI355X devices differ by up to 12 % on the same binary).
    python scripts/build_variant.py nostagger -DMLA_CONV_STAGGER=0"""
import importlib, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("audio-classification-using-a-deep-cnn-combined-with-multi-level-attention_amd.build")


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    only = None
    if flags and flags[0].startswith("--only="):
        only, flags = flags[0][7:].split(","), flags[1:]
    out_dir = os.path.join(ROOT, "build", "variants", name)
    os.makedirs(out_dir, exist_ok=True)

    def comp(src):
        obj = os.path.join(out_dir, src[:-4] + ".o")
        if only and src not in only:                       # unchanged sources: reuse the main build's object
            return os.path.join(b.OBJ, src[:-4] + ".o")
        subprocess.run([b.HIPCC] + b.FLAGS + b.PER_FILE_FLAGS.get(src, []) + flags + ["-c", os.path.join(b.CSRC, src), "-o", obj], check=True)
        return obj
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(comp, b.sources()))
    lib = os.path.join(ROOT, "build", "variants", "libmla_%s.so" % name)
    subprocess.run([b.HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib] + objs + ["-ldl"], check=True)
    print(lib)


if __name__ == "__main__":
    main()
