"""ctypes binding of libiadr1_hip.so (include/iadr1_hip.h).  The prototypes are parsed from the header
itself so the Python side cannot drift from the C ABI.  There is NO fallback: if the library is missing
or a call fails, this raises."""
from __future__ import annotations

import ctypes
import os
import re

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "iadr1_hip.h")
# IADR1_HIP_LIB: an alternative build of the same library (A/B probes of kernel variants: tools/build_variant.py); still no fallback of any kind
LIB_PATH = os.environ.get("IADR1_HIP_LIB") or os.path.join(HERE, "lib", "libiadr1_hip.so")

_CTYPE = {
    "int": ctypes.c_int, "unsigned": ctypes.c_uint, "float": ctypes.c_float, "long long": ctypes.c_longlong,
    "unsigned long long": ctypes.c_ulonglong, "iadr1_stream_t": ctypes.c_void_p,
}


def parse_header(path: str = HEADER) -> dict[str, tuple[str, list[tuple[str, str]]]]:
    """name -> (return type, [(ctype-name, arg-name)])"""
    txt = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"(int|long long|const char\*)\s+(iadr1_\w+)\s*\(([^)]*)\)\s*;", txt):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        parsed = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    parsed.append(("ptr", a.split("*")[-1].strip()))
                else:
                    parts = a.rsplit(" ", 1)
                    parsed.append((parts[0].replace("const ", "").strip(), parts[1]))
        protos[name] = (ret, parsed)
    return protos


PROTOS = parse_header()
_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (ret, args) in PROTOS.items():
            fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = {"int": ctypes.c_int, "long long": ctypes.c_longlong}.get(ret, ctypes.c_char_p)
            fn.argtypes = [ctypes.c_void_p if t == "ptr" else _CTYPE[t] for t, _ in args]
        _lib = L
    return _lib


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        return x.data_ptr()
    return int(x)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr() -> int:
    """hipStream_t of torch's current stream.  Through the raw accessor when torch has it: building a torch.cuda.Stream object per kernel launch
    costs ~9 us of host time, 6000 launches a step."""
    if _raw_stream is None or _get_device is None:
        return torch.cuda.current_stream().cuda_stream
    return _raw_stream(_get_device())


def call(name: str, *args):
    """Call iadr1_<name>(*args, stream) on torch's current stream.  Tensors -> device pointers."""
    full = "iadr1_" + name
    proto = PROTOS[full][1]
    if len(args) != len(proto) - 1:
        raise TypeError(f"{full}: expected {len(proto) - 1} args (+stream), got {len(args)}")
    conv = []
    for (t, an), a in zip(proto, args):
        conv.append(_ptr(a) if t == "ptr" else a)
    rc = getattr(lib(), full)(*conv, stream_ptr())
    if rc != 0:
        raise RuntimeError(f"{full} failed ({rc}): {lib().iadr1_last_error().decode()}")


def version() -> int:
    return lib().iadr1_version()


def cu_mask_stream(first_cu: int, n_cus: int, total_cus: int | None = None, cus=None) -> "torch.cuda.ExternalStream":
    """A HIP stream confined to CUs [first_cu, first_cu + n_cus) of the driver's numbering (include/iadr1_hip.h iadr1_stream_create_cu_mask: consecutive
    bits go round-robin over the XCDs, so a multiple of 8 CUs is the same share of every XCD), or to the explicit list `cus` (e.g. every CU of some XCDs:
    xcd_cus).  The stream is owned by the caller for the life of the process."""
    import numpy as np
    total = total_cus or torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    if cus is None:
        if not (0 <= first_cu and n_cus > 0 and first_cu + n_cus <= total):
            raise ValueError(f"CU range [{first_cu}, {first_cu + n_cus}) outside the device's {total} CUs")
        cus = range(first_cu, first_cu + n_cus)
    cus = sorted(set(int(c) for c in cus))
    if not cus or cus[0] < 0 or cus[-1] >= total:
        raise ValueError(f"CU list outside the device's {total} CUs")
    words = np.zeros((total + 31) // 32, dtype=np.uint32)
    for cu in cus:
        words[cu // 32] |= np.uint32(1 << (cu % 32))
    out = np.zeros(1, dtype=np.uint64)
    rc = lib().iadr1_stream_create_cu_mask(words.ctypes.data, len(words), out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"iadr1_stream_create_cu_mask failed ({rc}): {lib().iadr1_last_error().decode()}")
    _CU_SHARE[int(out[0])] = len(cus) / total
    return torch.cuda.ExternalStream(int(out[0]))


def xcd_cus(xcds, total_cus: int | None = None, n_xcd: int = 8) -> list:
    """Mask bits of every CU of the given XCDs (bit i of a CU mask is CU i // n_xcd of XCD i % n_xcd)."""
    total = total_cus or torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    xs = set(int(x) for x in xcds)
    return [i for i in range(total) if i % n_xcd in xs]


_CU_SHARE: dict = {}      # raw stream handle -> fraction of the device's CUs a CU-masked stream owns


def cu_share(stream=None) -> float:
    """Fraction of the device's CUs the given (default: torch's current) stream may use: 1.0 unless it came from cu_mask_stream."""
    st = stream if stream is not None else torch.cuda.current_stream()
    return _CU_SHARE.get(int(st.cuda_stream), 1.0)


def set_decode_cus(n_cus: int):
    """CUs the decode-step launchers size their persistent grids for (0 = the device's); set before the decode graph is captured."""
    rc = lib().iadr1_set_decode_cus(int(n_cus))
    if rc != 0:
        raise RuntimeError(f"iadr1_set_decode_cus failed ({rc}): {lib().iadr1_last_error().decode()}")
