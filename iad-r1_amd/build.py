"""Build libiadr1_hip.so (gfx950) in-tree with hipcc.  `python -m iadr1_amd.build` or via
__graft_entry__.build().  hipcc cross-compiles without a GPU; the .so travels to the GPU box."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libiadr1_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result"]
# per-file additions.  attn.hip: MFMA results in ordinary VGPRs -- the softmax / dS arithmetic works on the accumulators, and with them in AGPRs
# every such operation is a v_accvgpr_read + op + v_accvgpr_write (2-3 moves per MFMA in these kernels); none of them needs more than 256 registers
EXTRA_FLAGS = {"attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
if os.environ.get("IADR1_NO_EXTRA_FLAGS"):
    EXTRA_FLAGS = {}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256()
    for f in sources() + [os.path.join(CSRC, "common.h")]:
        h.update(open(f, "rb").read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "build.sha256")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        procs.append((src, subprocess.Popen([hipcc, *FLAGS, *EXTRA_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out.strip():
            print(out.decode())
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    open(stamp, "w").write(dig)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
