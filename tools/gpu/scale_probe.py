"""Beyond-2^31 check (manual, not part of the suite): 8M synthetic proteins = 2.35 G residues / windows in ONE batch —
64-bit offsets everywhere, 640k tiles in one look-back chain.  Sketches checked against the oracle on sequences sampled
across the batch (first, last, around the 2^31-th residue) and by CSR invariants."""
import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, '.')
import kmerseek_amd as ks
from kmerseek_amd import synth
from oracle import oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
k, scaled, mol = 10, 1, "protein"
t0 = time.time()
parts = [synth.proteome(1_000_000, stream=300 + i) for i in range(n // 1_000_000)]
res = np.concatenate([p[0] for p in parts])
offs = np.zeros(n + 1, np.uint64)
at, base = 1, np.uint64(0)
for r, o in parts:
    offs[at:at + len(o) - 1] = o[1:] + base
    at += len(o) - 1; base += o[-1]
del parts
print(f"generated {n} proteins, {len(res)} residues in {time.time() - t0:.1f} s", flush=True)
ctx = ks.Context(0)
d_res, d_off = ctx.to_device(res), ctx.to_device(offs)
lens = (offs[1:] - offs[:-1]).astype(np.int64)
t0 = time.time()
S = ctx.sketch_batch_device(d_res.ptr, d_off.ptr, n, len(res), k, scaled, mol, max_seq_len=int(lens.max()))
ctx.synchronize()
print(f"sketched in {(time.time() - t0) * 1e3:.1f} ms: {S.n_hashes} hashes, {S.n_windows} windows, stats {ctx.sketch_stats()}", flush=True)
assert S.n_windows == int(np.maximum(lens - k + 1, 0).sum())
L = ctx._L
csr = np.empty(n + 1, np.uint64)
ctx._check(L.ks_dev_download(ctx._h, csr.ctypes.data_as(C.c_void_p), C.c_void_p(L.ks_sketches_device_offsets(S._h)), csr.nbytes))
assert csr[0] == 0 and int(csr[-1]) == S.n_hashes and np.all(csr[1:] >= csr[:-1])
mid = int(np.searchsorted(offs, np.uint64(1 << 31)))
ids = np.unique(np.concatenate([np.arange(200), np.arange(n - 200, n), np.arange(mid - 100, mid + 100), np.linspace(0, n - 1, 500).astype(np.int64)]))
hp, ap = L.ks_sketches_device_hashes(S._h), L.ks_sketches_device_abunds(S._h)
bad = 0
for i in ids.tolist():
    b, e = int(csr[i]), int(csr[i + 1])
    h = np.empty(e - b, np.uint64); a = np.empty(e - b, np.uint32)
    if e > b:
        ctx._check(L.ks_dev_download(ctx._h, h.ctypes.data_as(C.c_void_p), C.c_void_p(hp + 8 * b), h.nbytes))
        ctx._check(L.ks_dev_download(ctx._h, a.ctypes.data_as(C.c_void_p), C.c_void_p(ap + 4 * b), a.nbytes))
    wm, wa = oracle.sketch_protein(bytes(res[int(offs[i]):int(offs[i + 1])]), k, scaled, mol)
    if not (np.array_equal(h, wm) and np.array_equal(a, wa)):
        bad += 1; print("MISMATCH at sequence", i)
print(f"{len(ids)} sampled sequences vs oracle: {bad} mismatches; residue 2^31 lies in sequence {mid}")
assert bad == 0
