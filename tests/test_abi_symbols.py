"""The C-ABI library builds for gfx950 without a GPU, loads, and exports every symbol include/sempyr.h declares.
No compute entry point is called here (no GPU in this container)."""
import ctypes
import os
import re

from semantic_pyramid_for_image_generation_amd import _lib


def test_header_declares_entry_points_with_reference_citations():
    text = open(_lib.HEADER).read()
    protos = _lib.parse_header()
    assert len(protos) >= 40
    for needle in ("models.py", "lossfunction.py", "model_wrapper.py"):
        assert needle in text
    assert "extern \"C\"" in text


def test_library_loads_and_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in _lib.parse_header():
        assert hasattr(handle, name), name
    lib = _lib.lib()
    assert lib.sp_version() == 1
    assert isinstance(lib.sp_last_error_string(), bytes)


def test_struct_layouts_match_header():
    text = re.sub(r"/\*.*?\*/", "", open(_lib.HEADER).read(), flags=re.S)
    conv = re.search(r"typedef struct sp_conv_params \{(.*?)\}", text, flags=re.S).group(1)
    fields = [f.strip().split()[-1].rstrip(";").lstrip("*") for f in conv.split(";") if f.strip()]
    flat = []
    for f in fields:
        flat += [x.strip().lstrip("*") for x in f.split(",")]
    decl = re.findall(r"(\w+)\s*[;,]", conv)
    assert [n for n, _ in _lib.SpConvParams._fields_] == decl
    sn = re.search(r"typedef struct sp_sn_layer \{(.*?)\}", text, flags=re.S).group(1)
    assert [n for n, _ in _lib.SpSnLayer._fields_] == re.findall(r"(\w+)\s*[;,]", sn)


def test_tuning_keys_match_header_and_are_accepted():
    """Every SP_TUNE_* key of the header has its SP_* environment name in the binding (same index), the library accepts all of
    them and rejects the first index past SP_TUNE_COUNT (sp_set_tuning touches no device state: callable without a GPU)."""
    text = re.sub(r"/\*.*?\*/", "", open(_lib.HEADER).read(), flags=re.S)
    enum = re.search(r"enum \{\s*(SP_TUNE_CONV_TALL.*?)\};", text, flags=re.S).group(1)
    keys = {k: int(v) for k, v in re.findall(r"(SP_TUNE_\w+)\s*=\s*(\d+)", enum)}
    count = keys.pop("SP_TUNE_COUNT")
    assert sorted(keys.values()) == list(range(count))
    assert {k.replace("SP_TUNE_", "SP_"): v for k, v in keys.items()} == _lib.TUNE_KEYS
    lib = _lib.lib()
    for v in keys.values():
        assert lib.sp_set_tuning(v, -1) == 0
    assert lib.sp_set_tuning(count, 0) != 0


def test_linear_split_rule_and_reduce_queue_host_side():
    """Host-only entry points (no device state): the split-K linear kernel's K range per block - the widest of 1024 / 512 / 256 / 128
    that yields 256 blocks, one split for small layers (csrc/linear.hip: linear_ks) - seen through sp_linear_workspace (floats =
    splits x batch x n), a forced width through SP_TUNE_LINEAR_KS; the deferred-reduction queue starts empty, defer / drop are
    accepted without a GPU."""
    lib = _lib.lib()
    SP_BF16 = _lib.SP_BF16

    def splits(b, k, n):
        out = ctypes.c_int64(-1)
        assert lib.sp_linear_workspace(b, k, n, SP_BF16, ctypes.byref(out)) == 0
        assert out.value % (b * n) == 0
        return out.value // (b * n)

    # (batch, k, n) -> K range: VGG FC1 forward 1024 (25 splits), its input gradient 1024 (4), FC2 512 (8), 4096 -> 2048 256 (16),
    # 1000 -> 4096 128 (8), the 365 -> 128 class mapping one split of 512
    assert splits(40, 25088, 4096) == 25
    assert splits(20, 4096, 25088) == 4
    assert splits(40, 4096, 4096) == 8
    assert splits(20, 4096, 2048) == 16
    assert splits(20, 1000, 4096) == 8
    assert splits(20, 365, 128) == 1
    key = _lib.TUNE_KEYS["SP_LINEAR_KS"]
    try:
        assert lib.sp_set_tuning(key, 128) == 0
        assert splits(40, 4096, 4096) == 32
    finally:
        lib.sp_set_tuning(key, -1)
    assert lib.sp_wgrad_reduce_pending() == 0
    assert lib.sp_wgrad_reduce_defer(1) == 0 and lib.sp_wgrad_reduce_defer(0) == 0
    assert lib.sp_wgrad_reduce_flush(0, None) == 0 and lib.sp_wgrad_reduce_pending() == 0
