"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol include/smx.h declares with the signatures the ctypes binding assumes.  No compute
call is made here (no GPU in the authoring container)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import stereo_matching_cuda_amd as smx
from stereo_matching_cuda_amd import _lib


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.SO_PATH):
        smx.build()
    return _lib.lib()


def _declared_functions():
    src = open(_lib.HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(smx_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree(lib):
    declared = _declared_functions()
    assert declared, "no functions parsed from include/smx.h"
    assert sorted(_lib.SIGNATURES) == declared
    for name in declared:
        assert hasattr(lib, name), f"libsmx_hip.so lacks {name}"


def test_default_params_match_reference_macros(lib):
    p = smx.default_params()  # SystemIncludes.h:7-24
    assert (p.r_w, p.g_w, p.b_w) == (0.299, 0.587, 0.0721)
    assert (p.alpha, p.th_color, p.th_grad, p.radius, p.eps, p.d_lr) == (0.9, 7, 2, 9, 6.5025, 0)


def test_struct_layout_matches_oracle_params(orc):
    assert C.sizeof(_lib.Params) == C.sizeof(orc.Params)
    assert [f[0] for f in _lib.Params._fields_] == [f[0] for f in orc.Params._fields_]


def test_pack_key_host_matches_oracle(lib, orc):
    rng = np.random.default_rng(1)
    vals = np.concatenate([rng.normal(size=100).astype(np.float32),
                           np.array([0.0, -0.0, 2.5, 3.3961514e38], np.float32)])
    for v in vals:
        for s in (0, 1, 191, 511):
            k = lib.smx_pack_key(float(v), s)
            assert k == int(orc.lib().orc_pack_key(float(v), s))
            c, sl = C.c_float(), C.c_uint32()
            lib.smx_unpack_key(k, C.byref(c), C.byref(sl))
            assert sl.value == s and np.float32(c.value) == (np.float32(0) if v == 0 else v)


def test_nan_cost_never_wins(lib, orc):
    nan_pos = np.float32(np.nan)
    nan_neg = np.array([0xFFC00000], np.uint32).view(np.float32)[0]
    for v in (nan_pos, nan_neg):
        assert lib.smx_pack_key(float(v), 3) == orc.KEY_IDENTITY
        assert int(orc.lib().orc_pack_key(float(v), 3)) == orc.KEY_IDENTITY
    assert int(orc.pack_keys(np.array([nan_neg, 1.0], np.float32), [3, 4])[0]) == orc.KEY_IDENTITY


def test_argument_errors_do_not_need_a_gpu(lib):
    p = smx.default_params()
    rc = lib.smx_compute_cost(C.byref(p), None, None, None, 4, 4, 4, 4, 1, 0)
    assert rc == -1 and b"bad argument" in lib.smx_last_error()
    buf = np.zeros(16, np.uint8)
    ptr = buf.ctypes.data_as(C.c_void_p)
    rc = lib.smx_compute_cost(C.byref(p), ptr, ptr, ptr, 4, 5, 4, 4, 1, 0)  # w1 != w2
    assert rc == -1
    assert lib.smx_agg_workspace_bytes(0, 4, 1) == 0
    assert lib.smx_agg_workspace_bytes(1242, 375, 192) > 5 * 192 * 1242 * 375 * 4


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "SO_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.lib()


@pytest.mark.skipif(smx._lib.lib().smx_device_count() > 0, reason="a GPU is present")
def test_compute_without_gpu_raises_not_falls_back():
    img = np.zeros((8, 8), np.uint8)
    with pytest.raises(smx.SmxError):
        smx.compute_cost(img, img, 2, 0)


def test_write_mat_matches_reference_normaliser(orc):
    rng = np.random.default_rng(3)
    for _ in range(5):
        m = rng.normal(size=(17, 23)).astype(np.float32) * 10
        assert np.array_equal(smx.write_mat(m), orc.write_mat_u8(m))
    m = np.array([[3, 1, 2], [0, 5, -1]], np.float32)  # min seen only via the else-branch
    assert np.array_equal(smx.write_mat(m), orc.write_mat_u8(m))
    m = np.arange(12, dtype=np.float32).reshape(3, 4)   # strictly increasing: min never updated
    assert np.array_equal(smx.write_mat(m), orc.write_mat_u8(m))


def test_header_is_plain_c_and_links(lib, tmp_path):
    """include/smx.h must be consumable from C (the boundary is a C-ABI, not a C++ API)."""
    import subprocess
    src = tmp_path / "c_abi.c"
    src.write_text('#include "smx.h"\n#include <stdio.h>\n'
                   'int main(void) { smx_params p; smx_default_params(&p);\n'
                   '  printf("%s %d %g\\n", smx_version(), p.radius, p.eps);\n'
                   '  return smx_device_count() < 0; }\n')
    exe = tmp_path / "c_abi"
    inc = os.path.dirname(_lib.HEADER_PATH)
    libdir = os.path.dirname(_lib.SO_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc,
                           str(src), "-o", str(exe), "-L", libdir, "-lsmx_hip",
                           "-Wl,-rpath," + libdir])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and "9 6.5025" in out.stdout


def test_fast_division_is_exhaustively_bit_identical_to_ieee(tmp_path):
    """div_small_int() of the fused kernel (x*r + one fma residual step, r = RN(1/area)) against the IEEE
    division for all 142 window areas x every f32 significand and sign at the extreme and middle
    exponents of the admitted range (7.2e9 cases, tools/check_fastdiv.c); smaller / non-finite x take
    the true division in the kernel (div_needs_exact)."""
    import subprocess
    exe = str(tmp_path / "check_fastdiv")
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-fopenmp",
                           os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_fastdiv.c"), "-lm", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " bad 0" in r.stdout, r.stdout


def test_integration_md_c_snippets_compile_against_the_headers(tmp_path):
    """Every ```cpp block of INTEGRATION.md that includes smx.h / smx_rccl.h is real code: it must pass
    g++ -fsyntax-only against include/ (the boundary document cannot drift from the headers again)."""
    import re
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.join(os.path.dirname(__file__), "..")
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = [b for b in re.findall(r"```cpp\n(.*?)```", text, re.S) if '#include "smx' in b]
    assert len(blocks) >= 4, "expected the context, stream, one-rank and sharded snippets"
    for k, b in enumerate(blocks):
        if '#include "smx.h"' in b and "CHECK(" in b:
            continue                      # the stages.cpp excerpt: compiled as part of host/stages.cpp
        src = tmp_path / f"snippet{k}.cpp"
        src.write_text("#include <stdint.h>\n#include <stddef.h>\n" + b)
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                            str(src)], capture_output=True, text=True)
        assert r.returncode == 0, f"snippet {k}:\n{b}\n{r.stderr}"


def test_comb_walker_descriptor_bound(lib):
    """The comb walker addresses both image planes, the guidance planes and their comb-ordered copies through ONE buffer
    descriptor with 32-bit offsets (0x80000000 = "outside the image"): aggregate_v4 picks it only while that region stays
    below 2 GiB.  The bound is the exact sum of the carved parts, not a per-pixel estimate (round-4 advisor finding: a
    24 B/pixel guard let 8192x5460 through at ~49 B/pixel).  Pure arithmetic: no GPU."""
    fn = C.CDLL(_lib.SO_PATH).smx_debug_v5_fix_bytes
    fn.restype, fn.argtypes = C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64)]

    def fix(w, h, nv):
        b = C.c_uint64()
        assert fn(w, h, nv, C.byref(b)) == 0
        return b.value
    # BASELINE shapes: far inside
    assert fix(1242, 375, 2) < 64 << 20 and fix(3840, 2160, 2) < 1 << 30
    # 44.7 Mpix pair: outside (the old guard, h * (w + 8) * 24 < 2^31, passed it)
    assert 5460 * (8192 + 8) * 24 < 1 << 31 <= fix(8192, 5460, 2)
    # per-pixel size of the region for a pair: two 4-byte image planes, two 8-byte guidance planes and the comb-ordered
    # copies (100 B per comb lane slot and band: ~25 B per pixel) -> well above 24 B
    assert 45 < fix(8192, 5460, 2) / (8192 * 5460) < 60
    # monotone in both dimensions around the limit, and a single view needs less
    assert fix(8192, 5000, 2) < fix(8192, 5460, 2) < fix(8400, 5460, 2) and fix(8192, 5460, 1) < fix(8192, 5460, 2)


def test_wta_run_equals_the_min_of_the_packed_keys(lib):
    """The WTA kernels pick the winner of a call's slices in the float domain (`take = q <= m`, smx_common.h WtaRun) and pack it
    once; that must be the integer min of the packed keys of every candidate (pack_key: cost ascending, -0 == +0, among equal
    costs the larger slice, NaN = the identity) -- dispSelectOnGPU's `best >= q` rule (guidedFilter.cu:403-411)."""
    so = C.CDLL(_lib.SO_PATH)
    so.smx_debug_wta_run.restype = C.c_int
    so.smx_debug_wta_run.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_uint32, C.POINTER(C.c_int64)]
    so.smx_pack_key.restype, so.smx_pack_key.argtypes = C.c_int64, [C.c_float, C.c_uint32]
    ident = 0x7FFFFFFFFFFFFFFF
    rng = np.random.default_rng(11)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, -np.nan, 1.0, -1.0, 3.4e38, -3.4e38, 1e-45, -1e-45], np.float32)
    for it in range(400):
        n = int(rng.integers(0, 20))
        if it % 3 == 0:
            q = rng.choice(special, size=n)
        elif it % 3 == 1:
            q = rng.choice(np.concatenate([special, rng.integers(-3, 4, size=6).astype(np.float32)]), size=n)
        else:
            q = rng.standard_normal(n).astype(np.float32)
        q = np.ascontiguousarray(q, np.float32)
        s0 = int(rng.integers(0, 1000))
        want = min([so.smx_pack_key(float(v), s0 + i) for i, v in enumerate(q)], default=ident)
        got = C.c_int64()
        assert so.smx_debug_wta_run(q.ctypes.data_as(C.POINTER(C.c_float)), n, s0, C.byref(got)) == 0
        assert got.value == want, (q, s0, got.value, want)


def test_max_slices_per_launch_knob(lib):
    assert lib.smx_set_max_slices_per_launch(-1) == -1
    assert lib.smx_set_max_slices_per_launch(7) == 0 and lib.smx_set_max_slices_per_launch(0) == 0
    c, n = C.c_int(-1), C.c_int(-1)
    assert lib.smx_last_agg_chunk(C.byref(c), C.byref(n)) == 0 and (c.value, n.value) == (0, 0)   # no aggregation yet on this thread
