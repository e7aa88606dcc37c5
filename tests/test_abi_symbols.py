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
