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


def test_gemm_refuses_operands_of_2_gib_or_more(built):
    """The LDS-DMA kernels address both operands with 32-bit byte offsets from their base (buffer descriptors): ccv_gemm reports
    an operand that spans 2 GiB or more as a shape error before any device work (the pointers below are never dereferenced)."""
    p = built.CcvGemm()
    p.A = p.W = p.C = 0x1000
    p.M, p.N, p.K, p.taps = 1 << 21, 320, 1024, 1
    p.lda, p.ldc = 1024, 320                       # activations: 2^21 rows x 2 KiB = 4 GiB
    p.alpha = 1.0
    rc = built.lib().ccv_gemm(ctypes.byref(p), None)
    assert rc < 0 and b"2 GiB" in built.lib().ccv_last_error()


def test_streams_in_flight_hint_round_trips(built):
    assert built.lib().ccv_set_streams_in_flight(2) == 1       # default: one launch stream
    assert built.lib().ccv_set_streams_in_flight(0) == 2       # clamped to >= 1
    assert built.lib().ccv_set_streams_in_flight(1) == 1


def test_product_path_refuses_cpu_tensors(built):
    from camc2v_amd import ops
    from camc2v_amd.lib import CcvError
    with pytest.raises(CcvError):
        ops.layernorm(torch.zeros(4, 64), torch.ones(64), torch.zeros(64))


def _plan(built, M, N, K, taps=1, gather=0, geglu=0, a_f32=0):
    p = built.CcvGemm()
    p.M, p.N, p.K, p.taps, p.gather, p.geglu, p.a_f32 = M, N, K, taps, gather, geglu, a_f32
    tile, split = ctypes.c_int32(-9), ctypes.c_int32(-9)
    rc = built.lib().ccv_gemm_plan(ctypes.byref(p), ctypes.byref(tile), ctypes.byref(split))
    assert rc == 0, built.lib().ccv_last_error()
    ws = built.lib().ccv_gemm_ws_bytes(ctypes.byref(p))
    assert ws == (split.value * M * N * 4 if split.value > 1 else 0)   # workspace request and plan agree
    return tile.value, split.value


def _gn_slots(built, M, N, K, rpi, taps=1, gather=0, out_f32=0, geglu=0):
    p = built.CcvGemm()
    p.M, p.N, p.K, p.taps, p.gather, p.geglu, p.out_f32, p.ldc = M, N, K, taps, gather, geglu, out_f32, N
    return built.lib().ccv_gemm_gn_slots(ctypes.byref(p), rpi)


def test_gemm_epilogue_groupnorm_slots_host_logic(built):
    """Which problems get their GroupNorm statistics from the GEMM epilogue is host code: slots per instance = row tiles inside the
    instance x column tiles of the kernel the planner picks, 0 where that kernel cannot produce them."""
    assert _gn_slots(built, 32768, 320, 320, 1024, taps=9, gather=1) == 8 * 2        # ResBlock conv at 32x32 latents: 128x160 tiles, a frame = 8 row tiles
    assert _gn_slots(built, 32768, 320, 320, 16384, taps=3, gather=2) > 128           # temporal conv, clip-wide norm: every tile of the clip a slot (<= 512)
    assert _gn_slots(built, 32768, 320, 320, 16384, taps=3, gather=2) <= 512
    assert _gn_slots(built, 8192, 640, 1280, 256, taps=9, gather=1) == 8              # split-K plan: the reduce pass emits them, 32 rows per workgroup (256 workgroups)
    assert _gn_slots(built, 8192, 640, 1280, 4096, taps=9, gather=1) == 128           # ... clip-wide instance of the same layer
    assert _gn_slots(built, 512, 1280, 1280, 256, taps=9, gather=1) == 128            # 4x4 latents: 2 rows per workgroup
    assert _gn_slots(built, 32768, 320, 320, 1024, taps=9, gather=1, out_f32=2) == 8 * 2   # fp16 (stream) output: same tiles
    assert _gn_slots(built, 8192, 640, 640, 4096) == 32 * 4                           # transformer proj_in / proj_out at 16x16 latents, clip-wide consumer: 128x160 tiles (plan table)
    assert _gn_slots(built, 32768, 320, 320, 1000, taps=9, gather=1) == 0             # instance rows not whole tiles
    assert _gn_slots(built, 32768, 960, 320, 1024) == 0                               # A-stationary kernel
    assert _gn_slots(built, 8192, 5120, 640, 256, geglu=1) == 0                       # GEGLU epilogue


def test_gemm_planner_host_logic(built):
    """Kernel selection is host code (no device work): the model's layer classes map to the kernels the sweep
    in profiles/ found fastest, and the workspace request always matches the planned split."""
    assert _plan(built, 1, 21120, 1280) == (-5, 1) and _plan(built, 4, 1280, 320) == (-5, 1)   # <= 4 rows: the row-vector kernel
    assert _plan(built, 5, 1280, 320)[0] != -5
    assert _plan(built, 32768, 960, 320) == (-4, 1)                      # QKV projection at 32x32 latents (K = 320, M = 256 x 128): A-stationary kernel
    assert _plan(built, 32768, 2560, 320, geglu=1) == (-4, 1)            # GEGLU up-projection there: the same
    assert _plan(built, 16384, 960, 320) == (-1, 1)                      # a single (not CFG-paired) forward: 128 row tiles would idle half the chip
    assert _plan(built, 8192, 5120, 640, geglu=1) == (5, 1)              # GEGLU up-projection at 16x16 latents: 2-stage 128x320 tile
    assert _plan(built, 8192, 1920, 640) == (6, 1)                       # QKV projection at 16x16 latents: 2-deep 128x160 ring (cold-operand sweep table)
    assert _plan(built, 512, 10240, 1280, geglu=1) == (-1, 1)            # ... unless its 128 tiles would leave CUs idle
    assert _plan(built, 2048, 1280, 5120) == (-1, 2)                     # FF down-projection at 8x8 latents: 128x160 family tile, 3 stages, split 2 (table)
    assert _plan(built, 512, 1280, 5120) == (-1, 4)                      # ... at 4x4 latents: 160 tiles, split-K 4
    assert _plan(built, 32768, 320, 320, taps=9, gather=1) == (-2, 1)    # conv3x3 at 32x32 latents: 128x160 family tile (128-byte rows)
    assert _plan(built, 32768, 320, 960, taps=9, gather=1) == (-2, 1)
    assert _plan(built, 8192, 640, 1280, taps=9, gather=1) == (-1, 2)    # 16x16 latents: 128x160 family tile, split 2 (table)
    assert _plan(built, 8192, 640, 640, taps=9, gather=1) == (-1, 2)
    assert _plan(built, 2048, 1280, 2560, taps=9, gather=1) == (-2, 4)   # 8x8 latents: 128 tiles of 128x160 x split 4
    assert _plan(built, 512, 1280, 1280, taps=9, gather=1) == (-1, 8)    # 4x4 latents: 32 tiles of 128x160 (3 stages) x 8 splits (table)
    assert _plan(built, 32768, 320, 320, taps=3, gather=2) == (-2, 1)
    assert _plan(built, 2048, 1280, 1280, taps=3, gather=2) == (-1, 1)   # temporal conv at 8x8 latents: 64x160 tile, 3 stages, unsplit (table)
    assert _plan(built, 512, 1280, 1280, taps=3, gather=2) == (-1, 3)
    assert _plan(built, 32768, 512, 2048) == (-1, 1)                     # N not a multiple of 160: never a ring tile
    assert _plan(built, 2048, 1280, 2560, a_f32=1)[0] == -1              # fp32 activations: register-staged kernel
    p = built.CcvGemm()
    tile, split = ctypes.c_int32(0), ctypes.c_int32(0)
    assert built.lib().ccv_gemm_plan(ctypes.byref(p), ctypes.byref(tile), ctypes.byref(split)) != 0


@pytest.mark.parametrize("nbh,ngroups", [(10, 256), (20, 64), (40, 16), (8, 100), (3, 17), (1, 5), (13, 33), (64, 7)])
def test_sparse_attention_xcd_queues_cover_every_item_once(built, nbh, ngroups):
    """Host logic of the sparse attention kernel's work distribution (`ccv_attn_sparse_queue_item`, the same inline function the
    kernel decodes its counter values with): the 8 per-XCD queues hold every (slice, rank) pair exactly once, ranks never
    decrease within a queue (longest-first order survives the merge), home slices stay on their XCD, a leftover slice is
    shared only by the XCDs assigned to it, and the queues are balanced to within one block of ranks."""
    L = built.lib()
    bh, rank = ctypes.c_int32(), ctypes.c_int32()
    seen, lens = {}, []
    for xq in range(8):
        n = L.ccv_attn_sparse_queue_item(nbh, ngroups, xq, -1, None, None)
        last, valid = -1, 0
        for i in range(n):
            assert L.ccv_attn_sparse_queue_item(nbh, ngroups, xq, i, ctypes.byref(bh), ctypes.byref(rank)) == n
            assert 0 <= bh.value < nbh and rank.value >= last
            last = rank.value
            if rank.value >= ngroups:
                continue                      # padding item: skipped by the kernel
            valid += 1
            assert (bh.value, rank.value) not in seen
            seen[(bh.value, rank.value)] = xq
            if bh.value < 8 * (nbh // 8):
                assert bh.value % 8 == xq     # home slices are private to their XCD
        lens.append(valid)
    assert len(seen) == nbh * ngroups
    n_extra = nbh % 8
    for s in range(8 * (nbh // 8), nbh):      # leftover slice k is worked on by the XCDs with xq % n_extra == k only
        owners = {x for (b, _), x in seen.items() if b == s}
        assert owners <= {x for x in range(8) if x % n_extra == s - 8 * (nbh // 8)}
    if nbh == 10 and ngroups == 256:          # the 32x32-latent launch of the benchmark: 320 items per XCD
        assert lens == [320] * 8
