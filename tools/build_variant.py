#!/usr/bin/env python3
"""Fast library variant for same-box A/B runs: compile ONLY the given replacement sources (against the product headers, or
a directory of replacement headers) and link them with the product's cached objects of every other source.

    python tools/build_variant.py OUT.so path/to/ssal_bottleneck_mfma.hip [more.hip ...] [-- extra hipcc flags]

The replacement's basename selects which product object it replaces."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from semanticsegmentationactivelearning_amd import build as B  # noqa: E402


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        k = args.index("--")
        args, extra = args[:k], args[k + 1:]
    out, repl = args[0], args[1:]
    B.build(verbose=False)  # product objects up to date
    hh = B._header_hash()
    objs = []
    names = {os.path.basename(r): r for r in repl}
    for src in B.sources(measure="-DSSAL_MEASURE" in extra):
        base = os.path.basename(src)
        if base in names:
            obj = os.path.join(B.OBJ, "variant_%s_%s.o" % (os.path.basename(out), base[:-4]))
            cmd = ["hipcc"] + B.CFLAGS + extra + ["-I", B.CSRC, "-I", os.path.join(ROOT, "include"), "-c", names[base], "-o", obj]
            print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        else:
            obj = os.path.join(B.OBJ, "%s.%s.o" % (base[:-4], B._source_digest(src, hh)[:16]))
            assert os.path.exists(obj), obj
        objs.append(obj)
    subprocess.check_call(["hipcc", "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", out] + objs)
    print("built", out)


if __name__ == "__main__":
    main()
