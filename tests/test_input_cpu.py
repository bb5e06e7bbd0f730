"""CPU-side checks of the host layer around the hot path (no GPU): the ingest's input layer (plain / gzip / zstd / bzip2 / xz by magic
number — the reference's needletail auto-detection, src/rust/index.rs:907-961, tested there at :1734-1845 with
tests/testdata/fasta/test_compression.fasta{,.zst}), thread safety of validate_and_resolve on a cold library, the bounded
reader of the flat index file, and the per-row parameter checks of the .sig.zip reader."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

from kmerseek_amd import build as ks_build, host, wire
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module", autouse=True)
def built():
    ks_build.build()


def test_plain_gzip_zstd_decode_to_the_same_bytes(tmp_path):
    plain = open(os.path.join(GOLDEN, "test_compression.fasta"), "rb").read()
    got, fmt = host.decompress(os.path.join(GOLDEN, "test_compression.fasta"))
    assert (got, fmt) == (plain, "plain")
    # the reference's own zstd fixture (index.rs:1734-1788 reads it through process_fasta)
    got, fmt = host.decompress(os.path.join(GOLDEN, "test_compression.fasta.zst"))
    assert (got, fmt) == (plain, "zstd")
    gz = tmp_path / "t.fasta.gz"
    gz.write_bytes(gzip.compress(plain))
    got, fmt = host.decompress(gz)
    assert (got, fmt) == (plain, "gzip")
    # multi-megabyte streams cross the reader's chunk boundaries; two gzip members back to back are one file
    big = (b">r\n" + bytes(np.random.default_rng(1).integers(65, 90, 3_000_000, dtype=np.uint8)) + b"\n") * 2
    gz2 = tmp_path / "big.fasta.gz"
    gz2.write_bytes(gzip.compress(big[:len(big) // 2]) + gzip.compress(big[len(big) // 2:]))
    got, fmt = host.decompress(gz2)
    assert got == big and fmt == "gzip"
    # bzip2 and xz: needletail's default features read them too (the reference's process_fasta goes through it, index.rs:920)
    import bz2, lzma
    for data, name in ((bz2.compress(big), "bzip2"), (bz2.compress(big[:1000]) + bz2.compress(big[1000:]), "bzip2"),
                       (lzma.compress(big), "xz"), (lzma.compress(big[:1000]) + lzma.compress(big[1000:]), "xz")):
        f = tmp_path / ("big." + name)
        f.write_bytes(data)
        got, fmt = host.decompress(f)
        assert got == big and fmt == name
    got, _ = host.decompress(os.path.join(GOLDEN, "bcl2_first25_uniprotkb_accession_O43236_OR_accession_2025_02_06.fasta.gz"))
    assert got == gzip.open(os.path.join(GOLDEN, "bcl2_first25_uniprotkb_accession_O43236_OR_accession_2025_02_06.fasta.gz")).read()


def test_truncated_archives_are_errors_not_shorter_files(tmp_path):
    plain = b">a\n" + b"ACDEFGHIKLMNPQRSTVWY" * 5000 + b"\n"
    gz = gzip.compress(plain)
    cut = tmp_path / "cut.fasta.gz"
    cut.write_bytes(gz[:len(gz) // 2])
    with pytest.raises(host.IndexError_) as e:
        host.decompress(cut)
    assert e.value.kind == "ParseError" and "truncated" in str(e.value)
    zst = open(os.path.join(GOLDEN, "test_compression.fasta.zst"), "rb").read()
    cutz = tmp_path / "cut.fasta.zst"
    cutz.write_bytes(zst[:-9])
    with pytest.raises(host.IndexError_) as e:
        host.decompress(cutz)
    assert e.value.kind == "ParseError" and ("truncated" in str(e.value) or "corrupt" in str(e.value))
    for magic, name in ((b"BZh91AY&SY", "bzip2"), (b"\xfd7zXZ\x00\x00\x04", "xz")):
        f = tmp_path / ("x." + name)
        f.write_bytes(magic + b"\0" * 32)
        with pytest.raises(host.IndexError_) as e:
            host.decompress(f)
        assert name in str(e.value)
    import bz2, lzma
    for data, name in ((bz2.compress(plain), "bzip2"), (lzma.compress(plain), "xz")):
        f = tmp_path / ("cut." + name)
        f.write_bytes(data[:len(data) * 2 // 3])
        with pytest.raises(host.IndexError_) as e:
            host.decompress(f)
        assert e.value.kind == "ParseError" and name in str(e.value)
    # a gzip file that ends exactly where a member ends is a complete (shorter) file, not a truncated one; one that ends inside
    # its second member is an error even though the first member was whole
    a, b = gzip.compress(plain), gzip.compress(b">b\n" + b"WYVT" * 9000 + b"\n")
    whole = tmp_path / "member_boundary.fasta.gz"
    whole.write_bytes(a)
    assert host.decompress(whole) == (plain, "gzip")
    both = tmp_path / "two_members.fasta.gz"
    both.write_bytes(a + b)
    assert host.decompress(both)[0] == plain + b">b\n" + b"WYVT" * 9000 + b"\n"
    half = tmp_path / "second_member_cut.fasta.gz"
    half.write_bytes(a + b[:len(b) // 2])
    with pytest.raises(host.IndexError_) as e:
        host.decompress(half)
    assert e.value.kind == "ParseError" and "truncated" in str(e.value)
    with pytest.raises(host.IndexError_):
        host.decompress(tmp_path / "does_not_exist.fasta")


_COLD = r"""
import ctypes as C, sys, threading
sys.path.insert(0, %r)
from kmerseek_amd import _lib
L = _lib.load()                      # fresh process: the class table of ks_validate_and_resolve has never been built
seq = (b"ACDEFGHIKLMNPQRSTVWYXUO" * 40) + b"BZJ" + b"acdefghiklmnpqrstvwy"
N = 32
barrier = threading.Barrier(N)
bad = []
def work(i):
    out = C.create_string_buffer(len(seq) + 1)
    n = C.c_uint64()
    err = _lib.ks_residue_error()
    barrier.wait()                   # all threads hit the cold table together (ctypes drops the GIL during the call)
    for _ in range(50):
        rc = L.ks_validate_and_resolve(seq, len(seq), 1, 7 + i, out, C.byref(n), C.byref(err))
        if rc != 0 or n.value != len(seq):
            bad.append((i, rc, n.value, chr(err.residue)))
            return
        got = out.raw[:n.value]
        if got[:920] != seq[:920] or got[923:] != seq[923:].upper():
            bad.append((i, "bytes"))
            return
        if got[920] not in b"DN" or got[921] not in b"EQ" or got[922] not in b"IL":
            bad.append((i, "ambiguity", got[920:923]))
            return
ts = [threading.Thread(target=work, args=(i,)) for i in range(N)]
[t.start() for t in ts]; [t.join() for t in ts]
print("BAD", bad) if bad else print("OK")
"""


def test_validate_and_resolve_is_thread_safe_on_a_cold_library():
    # the class table used to be filled lazily behind a plain `static bool`; 16 packer threads call this at once
    for _ in range(3):
        out = subprocess.run([sys.executable, "-c", _COLD % ROOT], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert out.stdout.strip() == "OK", out.stdout


def test_sig_zip_reader_checks_every_row(tmp_path):
    rng = np.random.default_rng(5)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    res, offs = oracle.pack([bytes(aa[rng.integers(0, 20, 400)]), bytes(aa[rng.integers(0, 20, 300)])])
    o5, m5, a5 = oracle.sketch_batch(res, offs, 5, 5, "protein")
    o1, m1, a1 = oracle.sketch_batch(res, offs, 5, 1, "protein")
    good = tmp_path / "good.sig.zip"
    wire.write_sig_zip(str(good), ["a", "b"], o5, m5, a5, 5, 5, "protein", "x.fasta")
    assert wire.read_sig_zip(str(good))[4:] == (5, 5, "protein")
    # a scaled=1 sketch filed under scaled=5: its hashes exceed max_hash(5)
    bad = tmp_path / "bad.sig.zip"
    wire.write_sig_zip(str(bad), ["a", "b"], o1, m1, a1, 5, 5, "protein", "x.fasta")
    assert (m1 > oracle.max_hash(5)).any()
    with pytest.raises(ValueError, match="outside"):
        wire.read_sig_zip(str(bad))
    # rows that disagree on their parameters (the last row alone used to decide)
    import csv, io, json, zipfile
    mixed = tmp_path / "mixed.sig.zip"
    za, zb = zipfile.ZipFile(good), None
    other = tmp_path / "other.sig.zip"
    wire.write_sig_zip(str(other), ["c"], o1[:2], m1[:int(o1[1])], a1[:int(o1[1])], 5, 1, "protein", "x.fasta")
    zb = zipfile.ZipFile(other)
    rows_a = [l for l in za.read("SOURMASH-MANIFEST.csv").decode().splitlines()]
    rows_b = [l for l in zb.read("SOURMASH-MANIFEST.csv").decode().splitlines()]
    with zipfile.ZipFile(mixed, "w") as z:
        for zz in (za, zb):
            for n in zz.namelist():
                if n != "SOURMASH-MANIFEST.csv":
                    z.writestr(n, zz.read(n))
        z.writestr("SOURMASH-MANIFEST.csv", "\n".join(rows_a + rows_b[2:]) + "\n")
    with pytest.raises(ValueError, match="one search takes one set of parameters"):
        wire.read_sig_zip(str(mixed))
