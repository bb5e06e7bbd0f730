"""CPU-side checks of the boundary: the C-ABI library builds, loads and exports every symbol that
include/kmerseek_amd.h declares; host-only entry points agree with the oracle.  No GPU compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import kmerseek_amd as ks
from kmerseek_amd import _lib, build as ks_build
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    ks_build.build()
    return _lib.load()


def declared_functions(header="kmerseek_amd.h", prefix="ks_"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(lib):
    names = declared_functions()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/kmerseek_amd.h but not exported"
    # and the ctypes table covers exactly the header
    assert sorted(_lib.SIGNATURES) == names


def test_host_shim_symbols_all_exported(lib):
    from kmerseek_amd import host
    names = declared_functions("kmerseek_host_c.h", "ksh_")
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/kmerseek_host_c.h but not exported"
    assert sorted(host.HOST_SIGNATURES) == names


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.ks_params) == 24
    assert C.sizeof(_lib.ks_kernel_time) == 64
    assert _lib.ks_params.seed.offset == 16


def test_version_and_strings(lib):
    assert lib.ks_abi_version() == 1
    assert lib.ks_status_string(0) == b"ok"
    assert b"amino acid" in lib.ks_status_string(_lib.KS_ERR_INVALID_RESIDUE)


def test_moltype_and_max_hash_agree_with_oracle(lib):
    for name, want in (("protein", 0), ("raw", 0), ("dayhoff", 1), ("hp", 2)):
        assert ks.moltype_id(name) == want == oracle.moltype_id(name)
    with pytest.raises(ks.KmerseekError) as e:
        ks.moltype_id("dna")
    assert str(e.value) == "Invalid moltype: dna, only 'protein', 'hp', or 'dayhoff' are supported"
    for scaled in (0, 1, 2, 3, 5, 7, 10, 100, 1000, 2**31, 2**32 - 1):
        assert ks.max_hash(scaled) == oracle.max_hash(scaled)
    assert ks.max_hash(5) == 3689348814741910528


def test_validate_and_resolve_matches_oracle(index_kats):
    rng = np.random.default_rng(1)
    alphabet = list(b"ACDEFGHIKLMNPQRSTVWYXUO*BZJ") + list(b"acdxbzj1$@ ")
    for _ in range(300):
        seq = bytes(rng.choice(alphabet, size=int(rng.integers(0, 40))).tolist())
        for upper in (False, True):
            src = seq.upper() if upper else seq
            try:
                want = oracle.validate_and_resolve(src, b"")
                werr = None
            except oracle.InvalidAminoAcid as e:
                want, werr = None, (e.char, e.pos)
            try:
                got = ks.validate_and_resolve(seq, upper=upper, rng_seed=7)
                gerr = None
            except ks.InvalidAminoAcid as e:
                got, gerr = None, (e.char, e.position)
            assert werr == gerr
            if want is not None:
                assert len(got) == len(want)
                for g, w, s in zip(got, want, src):
                    if s in b"BZJ":
                        assert bytes([g]) in {b"B": (b"D", b"N"), b"Z": (b"E", b"Q"), b"J": (b"I", b"L")}[bytes([s])]
                    else:
                        assert g == w
    for case in index_kats["invalid"]:
        with pytest.raises(ks.InvalidAminoAcid) as e:
            ks.validate_and_resolve(case["sequence"].encode())
        assert case["message"] in str(e.value) and e.value.position == 18
    # seeded resolution is reproducible and uses both candidates
    a = ks.validate_and_resolve(b"B" * 200, rng_seed=3)
    assert a == ks.validate_and_resolve(b"B" * 200, rng_seed=3) and set(a) == set(b"DN")
    assert ks.validate_and_resolve(b"mAaGg*cc", upper=True) == b"MAAGG*"


def test_no_device_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ks.KmerseekError) as e:
        ks.Context(0)
    assert e.value.status == _lib.KS_ERR_NO_DEVICE and "no CPU fallback" in str(e.value)


def test_nothing_throws_across_the_boundary(lib):
    """include/kmerseek_amd.h: every entry point that can allocate runs behind an exception guard — std::bad_alloc comes out
    as KS_ERR_OOM, anything else as KS_ERR_HIP, never as std::terminate (errors are values: src/rust/errors.rs:8-24)."""
    assert lib.ks_debug_guard_selftest(b"nothing") == _lib.KS_OK
    assert lib.ks_debug_guard_selftest(b"bad_alloc") in (_lib.KS_ERR_OOM, _lib.KS_ERR_HIP)  # (bad_alloc, or length_error from the reservation)
    assert lib.ks_debug_guard_selftest(b"runtime") == _lib.KS_ERR_HIP
    assert lib.ks_debug_guard_selftest(b"other") == _lib.KS_ERR_HIP


def test_every_entry_point_with_a_context_is_guarded():
    """Source check: an extern "C" int function that takes a ks_ctx and spans more than one statement hands its body to ks_guard."""
    import re
    csrc = os.path.join(ROOT, "kmerseek_amd", "csrc")
    unguarded = []
    for f in ("ks_api.hip", "ks_ctx.hip", "ks_copy.hip"):
        src = open(os.path.join(csrc, f)).read()
        for m in re.finditer(r'^extern "C" int (\w+)\(([^)]*)\)\s*\{', src, re.M):
            name, params = m.group(1), m.group(2)
            if "ks_ctx *ctx" not in params or name in ("ks_ctx_reload_debug_env",):
                continue
            head = src[m.end():m.end() + 120]
            if "ks_guard(" not in head:
                unguarded.append(name)
    assert not unguarded, unguarded


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "kmerseek_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src, f"{f} mentions the oracle"


def test_synth_is_deterministic():
    from kmerseek_amd import synth
    r1, o1 = synth.proteome(500, stream=9)
    r2, o2 = synth.proteome(500, stream=9)
    assert np.array_equal(r1, r2) and np.array_equal(o1, o2)
    q1 = synth.queries(300, r1, o1, stream=10)
    q2 = synth.queries(300, r1, o1, stream=10)
    assert np.array_equal(q1[0], q2[0]) and np.array_equal(q1[1], q2[1])
    lens = (o1[1:] - o1[:-1])
    assert lens.min() >= 30 and lens.max() <= 3000 and 200 < lens.mean() < 400
    assert set(np.unique(r1).tolist()) <= set(b"ACDEFGHIKLMNPQRSTVWY")
    # related queries really share k-mers with the index
    qo, qm, _ = oracle.sketch_batch(q1[0], q1[1], 10, 1, "protein")
    to, tm, ta = oracle.sketch_batch(r1, o1, 10, 1, "protein")
    qid, tid, isect, nw = oracle.manysearch(qo, qm, to, tm, ta, n_threads=4)
    assert len(set(qid.tolist())) >= 55 and isect.max() > 20


def test_ranks_that_do_not_build_wait_for_the_library(monkeypatch):
    """bench.py: rank 0 builds before it joins the process group, the other ranks wait for an up-to-date library (not inside
    init_process_group, whose timeout is short on purpose)."""
    from kmerseek_amd import build as ks_build
    so = ks_build.build()
    assert ks_build.wait_until_built(timeout_s=1.0) == so  # up to date: returns at once
    state = {"n": 0}

    def stale_twice():
        state["n"] += 1
        return state["n"] <= 2
    monkeypatch.setattr(ks_build, "stale", stale_twice)
    assert ks_build.wait_until_built(timeout_s=5.0, poll_s=0.01) == so and state["n"] == 3
    monkeypatch.setattr(ks_build, "stale", lambda: True)
    with pytest.raises(TimeoutError):
        ks_build.wait_until_built(timeout_s=0.05, poll_s=0.01)
