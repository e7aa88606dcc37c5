"""ctypes binding of libsempyr.so (the C ABI declared in include/sempyr.h).

The prototypes are parsed from the header itself, so the Python side cannot drift from the ABI and a
CPU-only test can check that every declared symbol is exported.  There is NO fallback: if the shared
object is missing or a symbol is absent, importing/using the ops raises - the product path never
silently runs on something else.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "sempyr.h")
LIB_PATH = os.environ.get("SEMPYR_LIB") or os.path.join(_HERE, "libsempyr.so")     # SEMPYR_LIB: A/B runs of two builds

SP_F32, SP_BF16, SP_F8, SP_F16, SP_F8_F16 = 0, 1, 2, 3, 4
ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH = 0, 1, 2, 3

# sp_set_tuning keys (include/sempyr.h).  The library reads no environment variables itself; an environment variable of
# the same name (SP_CONV_TALL=0 ...) is forwarded ONCE, when the library is loaded - that is how A/B runs flip a switch.
TUNE_KEYS = {"SP_CONV_TALL": 0, "SP_IGEMM_DMA": 1, "SP_WGRAD_ROWS": 2, "SP_DETERMINISTIC": 3, "SP_SPLITK_TARGET": 4,
             "SP_SPLITK_MINSTEPS": 5, "SP_CONV1X1_DIRECT": 6, "SP_CONV_SHORT": 7, "SP_WGRAD9_BLOCKS": 8, "SP_WGRAD_BLOCKS": 9,
             "SP_WGRAD_MINSTEPS": 10, "SP_WGRAD_SMALL_M": 11, "SP_WGRAD_K1_TILE64": 12, "SP_WGRAD_ROWS_THIN": 13,
             "SP_WGRAD_ROWS_BLOCKS": 14, "SP_WGRAD_ROWS_SLABS": 15, "SP_CONV_STAGGER": 16, "SP_CONV1X1_SPLITK": 17, "SP_WGRAD1X1": 18, "SP_CONV_CIN8": 19, "SP_CONV_THINCO": 20,
             "SP_CONV_PP": 21, "SP_CONV_PP_PRIO": 22, "SP_WGRAD_PP": 23, "SP_BN_ITERS": 24, "SP_IGEMM_TILE": 25, "SP_CONV_PPW": 26, "SP_LINEAR_KS": 27, "SP_CONV_PP_SPLIT": 28}


SP_CONV_SPLIT_SYNC_BYTES = 8192          # include/sempyr.h


class SpConvParams(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("w", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("y", ctypes.c_void_p),
                ("res1", ctypes.c_void_p), ("res2", ctypes.c_void_p), ("mask_src", ctypes.c_void_p),
                ("mask_neg_slope", ctypes.c_float),
                ("n", ctypes.c_int32), ("h", ctypes.c_int32), ("w_", ctypes.c_int32), ("cin_p", ctypes.c_int32),
                ("cout", ctypes.c_int32), ("ldy", ctypes.c_int32), ("ksize", ctypes.c_int32), ("act", ctypes.c_int32),
                ("dtype", ctypes.c_int32), ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_int64),
                ("pool2", ctypes.c_int32), ("in_up2", ctypes.c_int32),
                ("x_scale", ctypes.c_void_p), ("w_scale", ctypes.c_void_p), ("y8", ctypes.c_void_p),
                ("y8_inv_scale", ctypes.c_void_p), ("y8_amax", ctypes.c_void_p),
                ("img_scale", ctypes.c_void_p), ("img_split", ctypes.c_int32), ("reserved_", ctypes.c_int32),
                ("split_pix_", ctypes.c_int64),
                ("tail_w", ctypes.c_void_p), ("tail_bias", ctypes.c_void_p), ("tail_y", ctypes.c_void_p),
                ("tail_cout", ctypes.c_int32), ("tail_act", ctypes.c_int32), ("tail_ld", ctypes.c_int32), ("reserved2_", ctypes.c_int32),
                ("pool_idx", ctypes.c_void_p), ("split_sync", ctypes.c_void_p)]


class SpSnLayer(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("u", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("scratch_off", ctypes.c_int64), ("part_off", ctypes.c_int64), ("fwd_off", ctypes.c_int64),
                ("dgrad_off", ctypes.c_int64),
                ("rows", ctypes.c_int32), ("cols", ctypes.c_int32), ("cin", ctypes.c_int32), ("taps", ctypes.c_int32),
                ("cin_p", ctypes.c_int32), ("cout_p", ctypes.c_int32), ("kind", ctypes.c_int32),
                ("pack_block0", ctypes.c_int32)]


class SpSnBwdLayer(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("dw_off", ctypes.c_int64), ("dot_off", ctypes.c_int64),
                ("scratch_off", ctypes.c_int64), ("grad_off", ctypes.c_int64),
                ("rows", ctypes.c_int32), ("cols", ctypes.c_int32), ("cin", ctypes.c_int32), ("taps", ctypes.c_int32),
                ("cin_p", ctypes.c_int32), ("plain", ctypes.c_int32), ("db_off", ctypes.c_int32),
                ("bias_off", ctypes.c_int32)]


class SpRecLevel(ctypes.Structure):
    _fields_ = [("real", ctypes.c_void_p), ("fake", ctypes.c_void_p), ("mask", ctypes.c_void_p), ("dfake", ctypes.c_void_p),
                ("ld_real", ctypes.c_int32), ("ld_fake", ctypes.c_int32), ("ld_dfake", ctypes.c_int32), ("n", ctypes.c_int32),
                ("h", ctypes.c_int32), ("w_", ctypes.c_int32), ("c", ctypes.c_int32), ("reserved_", ctypes.c_int32)]


_CTYPE = {"int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "double": ctypes.c_double, "int": ctypes.c_int,
          "sp_stream_t": ctypes.c_void_p, "uint64_t": ctypes.c_uint64}


def parse_header(path: str = HEADER) -> Dict[str, Tuple[object, List[object]]]:
    """name -> (restype, argtypes) for every `int sp_*(...)` / `const char* sp_*(...)` prototype."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|const char\*)\s+(sp_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    ty = a.rsplit(" ", 1)[0].replace("const ", "").strip()
                    argtypes.append(_CTYPE[ty])
        protos[name] = (ctypes.c_char_p if "char" in ret else ctypes.c_int, argtypes)
    return protos


class SempyrError(RuntimeError):
    pass


_lib = None
_protos = None


def lib():
    """Loads libsempyr.so once; raises if it is missing (no CPU / torch fallback exists)."""
    global _lib, _protos
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SempyrError("libsempyr.so not found at %s - build it with __graft_entry__.build() "
                              "(semantic_pyramid_for_image_generation_amd/csrc/build.sh); there is no fallback path"
                              % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        _protos = parse_header()
        for name, (ret, argtypes) in _protos.items():
            fn = getattr(handle, name)         # AttributeError if the symbol is not exported
            fn.restype = ret
            fn.argtypes = argtypes
        for env_name, key in TUNE_KEYS.items():
            if os.environ.get(env_name) not in (None, ""):
                handle.sp_set_tuning(key, int(os.environ[env_name]))
        _lib = handle
    return _lib


def call(name: str, *args) -> None:
    """Calls an int-returning entry point and raises SempyrError (with the library's message) on failure."""
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise SempyrError("%s failed (%d): %s" % (name, rc, lib().sp_last_error_string().decode()))
