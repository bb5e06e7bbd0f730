"""Randomised differential test of the HIP path against the oracle (sketch, k-mer positions, search).

    python tools/fuzz_parity.py [--cases N] [--seed S]

Every case draws k, scaled, moltype, a length distribution (peptides / proteome-like / long / degenerate), an alphabet
(full, 2-letter, single residue, with ambiguity codes and lower case) and a batch size, sketches it through the C ABI and
compares with the oracle bit for bit; every third case also builds an index and searches it.  Prints one line per failure.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import kmerseek_amd as ks
from oracle import oracle

ALPHABETS = [b"ACDEFGHIKLMNPQRSTVWY", b"AC", b"A", b"ACDEFGHIKLMNPQRSTVWYXUO*BZJ", b"acdefghiklmnpqrstvwyACDEFG", b"LLLLLLLLLS"]


def draw_batch(rng):
    kind = rng.integers(0, 6)
    n = int(rng.integers(1, 400))
    if kind == 0:
        lens = rng.integers(0, 40, n)
    elif kind == 1:
        lens = np.clip(np.rint(rng.lognormal(np.log(260.0), 0.55, n)), 0, 3000)
    elif kind == 2:
        lens = rng.integers(700, 4200, max(1, n // 8))
    elif kind == 3:
        lens = rng.choice([0, 1, 5, 4079, 4080, 4081, 8200, 300], size=max(1, n // 10))
    elif kind == 4:
        lens = rng.integers(0, 12, n * 4)
    else:
        lens = np.concatenate([rng.integers(0, 300, n), [int(rng.integers(5000, 30000))]])
        rng.shuffle(lens)
    lens = lens.astype(np.uint64)
    offs = np.zeros(len(lens) + 1, np.uint64)
    np.cumsum(lens, out=offs[1:])
    alpha = np.frombuffer(ALPHABETS[int(rng.integers(0, len(ALPHABETS)))], np.uint8)
    res = rng.choice(alpha, size=int(offs[-1])).astype(np.uint8)
    if rng.random() < 0.3 and len(res) > 50:  # planted repeats: abundances > 1, heavy buckets
        unit = res[:int(rng.integers(3, 40))]
        reps = np.tile(unit, len(res) // len(unit) + 1)[:len(res)]
        mask = rng.random(len(res)) < 0.5
        res = np.where(mask, reps, res).astype(np.uint8)
    return res, offs


def run(cases: int, seed: int) -> int:
    """Runs `cases` random cases; prints one line per failure; returns the number of failures."""
    rng = np.random.default_rng(seed)
    ctx = ks.Context(0, follow_debug_env=True)
    bad = 0
    for case in range(cases):
        k = int(rng.choice([1, 2, 3, 5, 7, 8, 9, 10, 15, 16, 17, 21, 24, 31, 32, 33, 48, 64, 100, 128]))
        scaled = int(rng.choice([1, 1, 1, 2, 5, 10, 50, 1000]))
        mol = str(rng.choice(["protein", "dayhoff", "hp"]))
        res, offs = draw_batch(rng)
        tag = f"case {case}: k={k} scaled={scaled} {mol} n_seqs={len(offs) - 1} n_res={len(res)}"
        try:
            S = ctx.sketch_batch(res, offs, k, scaled, mol)
            got = S.to_host()
            want = oracle.sketch_batch(res, offs, k, scaled, mol, n_threads=8)
            if not all(np.array_equal(g, w) for g, w in zip(got, want)):
                print("SKETCH MISMATCH", tag); bad += 1; continue
            # (process_kmers hashes the validated, already upper-case sequence as given — src/rust/index.rs:749-786 — so the
            # oracle's position pass does not fold case; lower-case input never reaches it in the reference)
            if case % 4 == 1 and len(res) < 200000 and not np.any((res >= 97) & (res <= 122)):
                ps, pst, ph = ctx.kmer_positions(res, offs, k, scaled, mol)
                o, mins, _ = want
                ws, wst, wh = [], [], []
                for i in range(len(offs) - 1):
                    st, hh = oracle.kmer_positions(bytes(res[int(offs[i]):int(offs[i + 1])]), k, mol, mins[int(o[i]):int(o[i + 1])])
                    ws.append(np.full(len(st), i, np.uint32)); wst.append(st); wh.append(hh)
                if not (np.array_equal(ps, np.concatenate(ws)) and np.array_equal(pst, np.concatenate(wst)) and np.array_equal(ph, np.concatenate(wh))):
                    ws_, wst_, wh_ = np.concatenate(ws), np.concatenate(wst), np.concatenate(wh)
                    m = min(len(ps), len(ws_))
                    d = np.nonzero((ps[:m] != ws_[:m]) | (pst[:m] != wst_[:m]) | (ph[:m] != wh_[:m]))[0]
                    j = int(d[0]) if len(d) else m
                    ctxt = ""
                    if j < len(ws_):
                        sq, st0 = int(ws_[j]), int(wst_[j])
                        ctxt = f" want(seq={sq},start={st0},h={int(wh_[j])}) window={bytes(res[int(offs[sq]) + st0:int(offs[sq]) + st0 + k])!r} seqlen={int(offs[sq + 1] - offs[sq])} seqoff={int(offs[sq])}"
                    if j < len(ps):
                        ctxt += f" got(seq={int(ps[j])},start={int(pst[j])},h={int(ph[j])})"
                    print("POSITIONS MISMATCH", tag, f"n_got={len(ps)} n_want={len(ws_)} first_diff={j}" + ctxt); bad += 1; continue
            if case % 3 == 0:
                res2, offs2 = draw_batch(rng)
                if rng.random() < 0.5 and len(offs) > 2:  # make the queries overlap the targets
                    cut = int(offs[len(offs) // 2])
                    res2 = np.concatenate([res[:cut], res2]).astype(np.uint8)
                    offs2 = np.concatenate([offs[:len(offs) // 2], offs2 + np.uint64(cut)]).astype(np.uint64)
                Q = ctx.sketch_batch(res2, offs2, k, scaled, mol)
                ix = ctx.index_build(S)
                h = ctx.search(ix, Q).to_host()
                qo, qm, _ = Q.to_host()
                w = oracle.manysearch(qo, qm, want[0], want[1], want[2], n_threads=8)
                if not all(np.array_equal(x, y) for x, y in zip(h, w)):
                    print("SEARCH MISMATCH", tag, f"n_q={len(offs2) - 1} hits={len(h[0])}/{len(w[0])}"); bad += 1; continue
                d_res, d_off = ctx.to_device(res2), ctx.to_device(offs2)
                Q2 = ctx.sketch_queries_device(ix, d_res.ptr, d_off.ptr, len(offs2) - 1, len(res2))
                h2 = ctx.search(ix, Q2).to_host()
                if not all(np.array_equal(x, y) for x, y in zip(h2, w)):
                    print("FUSED SEARCH MISMATCH", tag); bad += 1; continue
                # one call (ks_sketch_search_device), with the bound that lets it defer the sketch's read-back
                mx = int((offs2[1:] - offs2[:-1]).max()) if len(offs2) > 1 else 0
                Q3, H3 = ctx.sketch_search_device(ix, d_res.ptr, d_off.ptr, len(offs2) - 1, len(res2), max_seq_len=mx)
                if not (all(np.array_equal(x, y) for x, y in zip(H3.to_host(), w)) and
                        all(np.array_equal(x, y) for x, y in zip(Q3.to_host(), Q.to_host()))):
                    print("ONE-CALL SEARCH MISMATCH", tag); bad += 1; continue
        except Exception as e:  # noqa: BLE001
            print("ERROR", tag, repr(e)); bad += 1
    ctx.close()
    return bad


def run_big(cases: int, seed: int) -> int:
    """Batches of 5k-60k proteins (many tiles, look-back chains, full partition paths): sketch vs oracle; search by the
    fused path (postings from the sketch kernel) vs the plain path, and partitioned vs LSD index build — GPU vs GPU,
    the pairwise oracle search is quadratic."""
    from kmerseek_amd import synth
    rng = np.random.default_rng(seed)
    ctx = ks.Context(0, follow_debug_env=True)
    bad = 0
    for case in range(cases):
        k = int(rng.choice([5, 7, 10, 16, 21, 24, 32]))
        scaled = int(rng.choice([1, 1, 2, 5, 20]))
        mol = str(rng.choice(["protein", "dayhoff", "hp"]))
        if mol == "hp" and k < 16:
            k = 16  # 2-letter alphabet: below ~16 every protein shares every k-mer and the match list exceeds its 2^32 cap
        nt, nq = int(rng.integers(5000, 60000)), int(rng.integers(5000, 60000))
        t_res, t_off = synth.proteome(nt, stream=5000 + case, hi=int(rng.choice([300, 3000, 9000])))
        q_res, q_off = synth.queries(nq, t_res, t_off, stream=6000 + case, frac_related=float(rng.choice([0.0, 0.2, 0.9])))
        tag = f"big case {case}: k={k} scaled={scaled} {mol} nt={nt} nq={nq}"
        if case % 4 == 3:
            # a big batch of PEPTIDES (>= 262,144 sequences: packed tiles then take up to 1024 sequences each — more than a
            # tile's LDS tables hold — and neighbouring peptides share k-mers)
            n = int(rng.integers(270000, 400000))
            lens = rng.integers(0, int(rng.choice([12, 20, 40])), n).astype(np.uint64)
            offs = np.zeros(n + 1, np.uint64)
            np.cumsum(lens, out=offs[1:])
            alpha = np.frombuffer(ALPHABETS[int(rng.integers(0, len(ALPHABETS)))], np.uint8)
            res = rng.choice(alpha, size=int(offs[-1])).astype(np.uint8)
            kk = int(rng.choice([2, 3, 5, 7]))
            tag = f"big case {case}: peptides k={kk} scaled={scaled} {mol} n={n}"
            try:
                got = ctx.sketch_batch(res, offs, kk, scaled, mol).to_host()
                if not all(np.array_equal(g, w) for g, w in zip(got, oracle.sketch_batch(res, offs, kk, scaled, mol, n_threads=16))):
                    print("SKETCH MISMATCH", tag); bad += 1
            except Exception as e:  # noqa: BLE001
                print("ERROR", tag, repr(e)); bad += 1
            continue
        try:
            T = ctx.sketch_batch(t_res, t_off, k, scaled, mol)
            if not all(np.array_equal(g, w) for g, w in zip(T.to_host(), oracle.sketch_batch(t_res, t_off, k, scaled, mol, n_threads=16))):
                print("SKETCH MISMATCH", tag); bad += 1; continue
            ix = ctx.index_build(T)
            Q = ctx.sketch_batch(q_res, q_off, k, scaled, mol)
            plain = ctx.search(ix, Q).to_host()
            d_res, d_off = ctx.to_device(q_res), ctx.to_device(q_off)
            Qf = ctx.sketch_queries_device(ix, d_res.ptr, d_off.ptr, nq, len(q_res))
            fused = ctx.search(ix, Qf).to_host()
            os.environ["KS_DEBUG_INDEX_LSD"] = "1"
            lsd = ctx.search(ctx.index_build(T), Q).to_host()
            del os.environ["KS_DEBUG_INDEX_LSD"]
            if not all(np.array_equal(x, y) for x, y in zip(plain, fused)):
                print("FUSED != PLAIN", tag); bad += 1; continue
            one = ctx.sketch_search_device(ix, d_res.ptr, d_off.ptr, nq, len(q_res), max_seq_len=int((q_off[1:] - q_off[:-1]).max()),
                                           want_sketches=False)[1].to_host()
            if not all(np.array_equal(x, y) for x, y in zip(plain, one)):
                print("ONE CALL != PLAIN", tag); bad += 1; continue
            if not all(np.array_equal(x, y) for x, y in zip(plain, lsd)):
                print("PARTITIONED INDEX != LSD INDEX", tag); bad += 1; continue
            # oracle on a few queries
            qo, qm, _ = Q.to_host()
            to, tm, ta = T.to_host()
            pick = rng.choice(nq, size=6, replace=False)
            for qi in pick.tolist():
                w = oracle.manysearch(np.array([0, qo[qi + 1] - qo[qi]], np.uint64), qm[int(qo[qi]):int(qo[qi + 1])], to, tm, ta, n_threads=16)
                sel = plain[0] == qi
                if not (np.array_equal(plain[1][sel], w[1]) and np.array_equal(plain[2][sel], w[2]) and np.array_equal(plain[3][sel], w[3])):
                    print("SEARCH MISMATCH", tag, f"query {qi}"); bad += 1; break
        except Exception as e:  # noqa: BLE001
            print("ERROR", tag, repr(e)); bad += 1
    ctx.close()
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--big", action="store_true", help="few large batches instead of many small ones")
    a = ap.parse_args()
    bad = run_big(a.cases, a.seed) if a.big else run(a.cases, a.seed)
    print(f"{a.cases} cases, {bad} failures")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
