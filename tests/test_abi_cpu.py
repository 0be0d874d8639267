"""CPU-side checks of the drop-in boundary: the shared library loads and exports every
symbol include/ccv.h declares, the ctypes structs match the header field for field, and
the product path refuses to run without a GPU (no silent fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    from camc2v_amd import lib
    return lib


def _header():
    return open(os.path.join(ROOT, "include", "ccv.h")).read()


def test_library_exports_every_declared_symbol(built):
    declared = set(re.findall(r"\b(ccv_[a-z0-9_]+)\s*\(", _header()))
    declared = {d for d in declared if not d.endswith("_")}
    assert {"ccv_gemm", "ccv_attn_fwd", "ccv_groupnorm", "ccv_layernorm", "ccv_ddim_cfg_step"} <= declared
    handle = ctypes.CDLL(built.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(handle, name), f"{name} declared in include/ccv.h but not exported"
    assert set(built.SIGNATURES) == declared
    assert built.lib().ccv_version() == 100


def _struct_fields(name):
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), _header(), re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = [n.strip().lstrip("*") for n in decl.split(",")]
        first = names[0].split()[-1].lstrip("*")
        fields.append(first)
        fields.extend(names[1:])
    return fields


@pytest.mark.parametrize("name", ["CcvGemm", "CcvAttn"])
def test_ctypes_structs_mirror_header(built, name):
    assert [f for f, _ in getattr(built, name)._fields_] == _struct_fields(name)


def test_argument_errors_are_reported_not_thrown(built):
    p = built.CcvGemm()
    rc = built.lib().ccv_gemm(ctypes.byref(p), None)
    assert rc == -1 and b"null" in built.lib().ccv_last_error()


def test_product_path_refuses_cpu_tensors(built):
    from camc2v_amd import ops
    from camc2v_amd.lib import CcvError
    with pytest.raises(CcvError):
        ops.layernorm(torch.zeros(4, 64), torch.ones(64), torch.zeros(64))
