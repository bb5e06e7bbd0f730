"""Deterministic synthetic proteomes for parity tests and bench.py.

Lengths: log-normal(mu = ln 260, sigma = 0.55) clipped to [30, 3000] (mean ~300 aa).
Residues: i.i.d. from UniProt-like frequencies; no B/Z/J/X/U/O/*/lower-case.
Queries: 20 % mutated copies of random index proteins (10 % substitutions, 1 % indels) come first,
then 80 % independent proteins.
Generator: numpy PCG64 seeded with 0x6b6d6572 + stream id.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

BASE_SEED = 0x6B6D6572
AA = np.frombuffer(b"LSPAEGRVKTQDINFHCYMW", dtype=np.uint8)
FREQ = np.array([9.6, 9.0, 7.1, 6.9, 6.9, 6.7, 6.2, 5.6, 5.6, 5.4, 4.9, 4.5, 3.9, 3.4, 3.3, 2.8, 2.5, 2.3, 2.1, 1.4])
CDF = np.cumsum(FREQ / FREQ.sum())
CDF[-1] = 1.0
# 16-bit inverse-CDF table: residue for a uniform u16 draw (frequencies quantised to 1/65536)
_TABLE = AA[np.searchsorted(CDF, (np.arange(65536) + 0.5) / 65536.0, side="right").clip(0, 19)]


def _residues(rng: np.random.Generator, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.uint8)
    step = 1 << 24
    for i in range(0, n, step):
        m = min(step, n - i)
        out[i:i + m] = _TABLE[rng.integers(0, 65536, m, dtype=np.uint16)]
    return out


def lengths(rng: np.random.Generator, n: int, lo: int = 30, hi: int = 3000) -> np.ndarray:
    return np.clip(np.rint(rng.lognormal(np.log(260.0), 0.55, n)), lo, hi).astype(np.uint64)


def proteome(n_seqs: int, stream: int = 0, lo: int = 30, hi: int = 3000) -> Tuple[np.ndarray, np.ndarray]:
    """(residues u8, offsets u64[n+1]) of n_seqs independent proteins."""
    rng = np.random.Generator(np.random.PCG64(BASE_SEED + stream))
    lens = lengths(rng, n_seqs, lo, hi)
    offs = np.zeros(n_seqs + 1, dtype=np.uint64)
    np.cumsum(lens, out=offs[1:])
    return _residues(rng, int(offs[-1])), offs


def queries(n_queries: int, index_res: np.ndarray, index_offs: np.ndarray, stream: int = 1,
            frac_related: float = 0.2, p_sub: float = 0.10, p_indel: float = 0.01, shuffle: bool = True
            ) -> Tuple[np.ndarray, np.ndarray]:
    """Query set: frac_related mutated copies of random index proteins, the rest independent, in random order
    (shuffle=False leaves the related ones first — which flatters the join: all matches then sit in the same few
    registers of a workgroup)."""
    rng = np.random.Generator(np.random.PCG64(BASE_SEED + stream))
    n_rel = int(round(n_queries * frac_related))
    n_ind = n_queries - n_rel
    n_index = len(index_offs) - 1
    # --- mutated copies (vectorised over the concatenation of the chosen proteins)
    src = rng.integers(0, n_index, n_rel)
    s_b = index_offs[src].astype(np.int64)
    s_len = (index_offs[src + 1] - index_offs[src]).astype(np.int64)
    tot = int(s_len.sum())
    starts = np.zeros(n_rel + 1, dtype=np.int64)
    np.cumsum(s_len, out=starts[1:])
    gather = np.arange(tot, dtype=np.int64) - np.repeat(starts[:-1], s_len) + np.repeat(s_b, s_len)
    res = index_res[gather].copy()
    sub = rng.random(tot) < p_sub
    res[sub] = _residues(rng, int(sub.sum()))
    u = rng.random(tot)
    rep = np.ones(tot, dtype=np.int64)
    rep[u < p_indel / 2] = 0                       # deletion
    rep[(u >= p_indel / 2) & (u < p_indel)] = 2    # insertion after this residue
    first = starts[:-1][s_len > 0]
    rep[first] = np.maximum(rep[first], 1)         # never delete a whole 1-residue protein's only residue
    new_len = np.add.reduceat(rep, starts[:-1]) if tot else np.zeros(n_rel, np.int64)
    if tot:
        new_len[s_len == 0] = 0
    out_res = np.repeat(res, rep)
    # the second copy of a doubled residue becomes a random insertion
    ins_pos = np.cumsum(rep)[rep == 2] - 1
    out_res[ins_pos] = _residues(rng, len(ins_pos))
    # --- independent proteins
    lens_ind = lengths(rng, n_ind).astype(np.int64)
    ind_res = _residues(rng, int(lens_ind.sum()))
    all_len = np.concatenate([new_len, lens_ind]).astype(np.uint64)
    offs = np.zeros(n_queries + 1, dtype=np.uint64)
    np.cumsum(all_len, out=offs[1:])
    all_res = np.concatenate([out_res, ind_res])
    if not shuffle or n_queries < 2:
        return all_res, offs
    perm = rng.permutation(n_queries)
    p_len = all_len[perm].astype(np.int64)
    p_offs = np.zeros(n_queries + 1, dtype=np.uint64)
    np.cumsum(p_len.astype(np.uint64), out=p_offs[1:])
    # gather in slabs of sequences to bound the index arrays (a 1M-protein set is 300M residues)
    out = np.empty(int(p_offs[-1]), dtype=np.uint8)
    step = 100_000
    for lo in range(0, n_queries, step):
        hi = min(lo + step, n_queries)
        ln = p_len[lo:hi]
        tot_l = int(ln.sum())
        if tot_l == 0:
            continue
        dst0 = int(p_offs[lo])
        starts_l = np.zeros(hi - lo, dtype=np.int64)
        np.cumsum(ln[:-1], out=starts_l[1:])
        src0 = offs[perm[lo:hi]].astype(np.int64)
        idx = np.arange(tot_l, dtype=np.int64) - np.repeat(starts_l, ln) + np.repeat(src0, ln)
        out[dst0:dst0 + tot_l] = all_res[idx]
    return out, p_offs
