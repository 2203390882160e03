"""Build recipe of libssal_hip.so: hipcc, gfx950 only, in-tree (the .so travels with the repo
snapshot to the GPU box)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libssal_hip.so")

# -ffp-contract=off: every fused multiply-add in the kernels is an explicit fmaf(), so results are
# bit-comparable with the parity oracle (which is built the same way).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fvisibility=hidden", "-Wall"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


STAMP = OUT + ".stamp"


def _digest():
    """content hash of every input of the build (mtimes do not survive a repo snapshot copy)"""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    deps = sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    deps.append(os.path.join(HERE, "..", "include", "ssal_enet.h"))
    for d in deps:
        h.update(os.path.basename(d).encode())
        h.update(open(d, "rb").read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(OUT) or not os.path.exists(STAMP):
        return True
    return open(STAMP).read().strip() != _digest()


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + ["-o", OUT + ".tmp"] + sources()
    if verbose:
        print("[ssal build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(OUT + ".tmp", OUT)
    with open(STAMP, "w") as f:
        f.write(_digest())
    return OUT


if __name__ == "__main__":
    build(force=True)
