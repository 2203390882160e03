"""Build recipe of libssal_hip.so: hipcc, gfx950 only, in-tree (the .so travels with the repo
snapshot to the GPU box).  Every csrc/*.hip is compiled to its own object (in parallel, cached by a
content hash of the source, the headers and the flags) and the objects are linked into one shared
library."""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, ".obj")
OUT = os.path.join(HERE, "libssal_hip.so")

# -ffp-contract=off: every fused multiply-add in the kernels is an explicit fmaf(), so results are
# bit-comparable with the parity oracle (which is built the same way).
# gfx950:xnack-: the pool runs with XNACK off (XNACK-on is not available on it); telling the compiler so lets it reuse the
# address registers of a vector load for its result (no replay to protect): every fused kernel 2-4 % faster, identical bits
# (profiles/r03_ab_xnack_off.txt)
ARCH = "gfx950:xnack-"
CFLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
          "-fvisibility=hidden", "-Wall"]
FLAGS = CFLAGS + ["-shared"]  # one-shot form (tools/phase_trace.py builds its measurement variant with it)


MEASURE_ONLY = ("ssal_probe.hip", "ssal_split_probe.hip")  # copy / MFMA probes, the bf16x3 split-operand bottleneck: measurement libraries only (tools/phase_trace.py)


def sources(measure=False):
    """product sources; measure=True adds the probe kernels of the -DSSAL_MEASURE libraries"""
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                  if f.endswith(".hip") and (measure or f not in MEASURE_ONLY))


def _headers():
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    inc = os.path.join(HERE, "..", "include")
    deps += sorted(os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h"))
    return deps


STAMP = OUT + ".stamp"


def _header_hash():
    h = hashlib.sha256(" ".join(CFLAGS).encode())
    for d in _headers():
        h.update(os.path.basename(d).encode())
        h.update(open(d, "rb").read())
    return h


def _source_digest(src, hh):
    h = hh.copy()
    h.update(os.path.basename(src).encode())
    h.update(open(src, "rb").read())
    return h.hexdigest()


def _digest():
    """content hash of every input of the build (mtimes do not survive a repo snapshot copy)"""
    hh = _header_hash()
    h = hashlib.sha256()
    for s in sources():
        h.update(_source_digest(s, hh).encode())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(OUT) or not os.path.exists(STAMP):
        return True
    return open(STAMP).read().strip() != _digest()


def build(force=False, verbose=True, jobs=None):
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJ, exist_ok=True)
    hh = _header_hash()
    objs, todo = [], []
    for src in sources():
        base = os.path.basename(src)[:-4]
        obj = os.path.join(OBJ, "%s.%s.o" % (base, _source_digest(src, hh)[:16]))
        objs.append(obj)
        if force or not os.path.exists(obj):
            todo.append((src, obj, base))

    def compile_one(item):
        src, obj, base = item
        for old in os.listdir(OBJ):  # drop stale objects of this source
            if old.startswith(base + ".") and os.path.join(OBJ, old) != obj:
                os.remove(os.path.join(OBJ, old))
        cmd = [hipcc] + CFLAGS + ["-c", src, "-o", obj + ".tmp"]
        if verbose:
            print("[ssal build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(obj + ".tmp", obj)

    if todo:
        with concurrent.futures.ThreadPoolExecutor(max_workers=jobs or min(len(todo), 6)) as ex:
            list(ex.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT + ".tmp"] + objs
    if verbose:
        print("[ssal build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(OUT + ".tmp", OUT)
    with open(STAMP, "w") as f:
        f.write(_digest())
    return OUT


if __name__ == "__main__":
    import sys
    build(force="--force" in sys.argv)
