#!/usr/bin/env python3
"""Builds a VARIANT of libiadr1_hip.so with extra preprocessor defines for A/B probes of kernel variants on the GPU box:
    python tools/build_variant.py NAME -DIADR1_X=1 [-D...]     ->  iad-r1_amd/lib/variants/libiadr1_hip_NAME.so
Select it with IADR1_HIP_LIB=<path> (iadr1_amd.hip).  The product build (iadr1_amd.build) is untouched."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iadr1_amd  # noqa
from iadr1_amd import build as b

name, defs = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(b.LIBDIR, "variants", name)
os.makedirs(out_dir, exist_ok=True)
objs, procs = [], []
for src in b.sources():
    obj = os.path.join(out_dir, os.path.basename(src)[:-4] + ".o")
    objs.append(obj)
    procs.append((src, subprocess.Popen(["/opt/rocm/bin/hipcc", *b.FLAGS, *b.EXTRA_FLAGS.get(os.path.basename(src), []), *defs, "-c", src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
for src, pr in procs:
    o, _ = pr.communicate()
    if pr.returncode:
        raise SystemExit(f"hipcc failed on {src}:\n{o.decode()}")
lib = os.path.join(b.LIBDIR, "variants", f"libiadr1_hip_{name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
print(lib)
