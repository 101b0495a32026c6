"""Deterministic synthetic corpora shared by tests, golden generation and bench.py.

Every generator is a pure function of (kind, n, seed) built on numpy's PCG64 with
fixed algorithms, so the GPU box regenerates exactly the inputs the goldens were
made from (tests/golden/manifest.json stores the input SHA-256 as a guard).

Kinds mirror the reference's own test data where it has any:
  runs    genRandomData()  test/main.c:293-310  (run length U[0,99], byte U[90,154])
  mod200  bt.c corpus 1    test/bt.c:109-122     (j % 200)
  allA    bt.c corpus 3    ('A' * n)
  rand    bt.c corpus 2-ish / incompressible => stored blocks
  text    Zipf words over a 4000-word lexicon (SURVEY.md §8d "Silesia-like" part 1)
  records structured binary: LE counters + repeated 16-64 B records
  lzmix   literal bursts + copies from history at all distances/lengths (stress)
  silesia 55% text / 15% runs / 20% records / 10% rand in 4-64 KB segments
"""
import hashlib

import numpy as np

KINDS = ("rand", "allA", "mod200", "runs", "text", "records", "lzmix", "silesia")


def _rng(seed, kind):
    return np.random.Generator(np.random.PCG64([seed, KINDS.index(kind) + 1]))


def _runs(n, g):
    if n == 0:
        return np.zeros(0, np.uint8)
    cnt = n // 40 + 16
    out = []
    tot = 0
    while tot < n:
        lens = g.integers(0, 100, cnt)
        vals = g.integers(90, 155, cnt).astype(np.uint8)
        out.append(np.repeat(vals, lens))
        tot += int(lens.sum())
    return np.concatenate(out)[:n]


_LEX = None


def _lexicon():
    """4000 pseudo-words (2-10 letters, English-like letter frequencies) as a padded matrix."""
    global _LEX
    if _LEX is None:
        lex_g = np.random.Generator(np.random.PCG64(4000))
        nwords = 4000
        wl = lex_g.integers(2, 11, nwords)
        letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", np.uint8)
        pl = np.arange(1, 27, dtype=np.float64) ** -1.0
        pl /= pl.sum()
        mat = letters[lex_g.choice(26, (nwords, 11), p=pl)]
        pz = np.arange(1, nwords + 1, dtype=np.float64) ** -1.1
        pz /= pz.sum()
        _LEX = (mat, wl.astype(np.int64), np.cumsum(pz))
    return _LEX


def _text(n, g):
    """Zipf-distributed words separated by space/newline/punctuation (vectorised)."""
    if n == 0:
        return np.zeros(0, np.uint8)
    mat, wl, cdf = _lexicon()
    seps = np.frombuffer(b"     \n,.", np.uint8)
    out = []
    tot = 0
    while tot < n:
        need = (n - tot) // 5 + 64
        idx = np.searchsorted(cdf, g.random(need)).clip(0, len(wl) - 1)
        lens = wl[idx] + 1                                  # word + separator
        starts = np.cumsum(lens) - lens
        total = int(lens.sum())
        owner = np.repeat(np.arange(need), lens)
        off = np.arange(total) - starts[owner]
        m = mat.copy()
        a = m[idx[owner], np.minimum(off, 10)]
        sep_pos = off == (lens[owner] - 1)
        a[sep_pos] = seps[g.integers(0, len(seps), int(sep_pos.sum()))]
        out.append(a)
        tot += total
    return np.concatenate(out)[:n]


def _records(n, g):
    if n == 0:
        return np.zeros(0, np.uint8)
    out = []
    tot = 0
    ctr = int(g.integers(0, 1 << 20))
    while tot < n:
        rl = int(g.integers(16, 65))
        reps = int(g.integers(8, 200))
        base = g.integers(0, 256, rl).astype(np.uint8)
        blk = np.tile(base, (reps, 1))
        c = (np.arange(reps, dtype=np.uint32) + ctr).view(np.uint8).reshape(reps, 4)
        blk[:, :4] = c
        nmut = max(1, rl // 8)
        cols = g.integers(4, rl, nmut)
        blk[:, cols] = g.integers(0, 256, (reps, nmut)).astype(np.uint8)
        ctr += reps
        out.append(blk.reshape(-1))
        tot += blk.size
    return np.concatenate(out)[:n]


def _lzmix(n, g):
    out = np.zeros(n, np.uint8)
    if n == 0:
        return out
    pos = 0
    while pos < n:
        if pos > 0 and g.random() < 0.55:
            maxd = min(pos, 70000)
            # favour the interesting distances
            d = int(g.choice([1, 2, 3, 4, int(g.integers(1, maxd + 1)), int(g.integers(1, min(maxd, 300) + 1)),
                              min(maxd, 32506), min(maxd, 32505), min(maxd, 32507), min(maxd, 32768)]))
            ln = int(g.choice([3, 4, 5, 6, 7, 8, 9, int(g.integers(3, 40)), int(g.integers(3, 600)), 258, 259, 257]))
            ln = min(ln, n - pos)
            for i in range(ln):                       # overlapping copies need the byte loop
                k = pos + i - d                       # (the fixed distances 1..4 may reach before the start when pos < 4: numpy
                out[pos + i] = out[k] if k >= -n else 0   #  then reads from the end - kept as it is, every seed's bytes are what
                                                      #  they were - except where that index does not exist: n = 2, 3)
            pos += ln
        else:
            ln = min(int(g.integers(1, 24)), n - pos)
            out[pos:pos + ln] = g.integers(0, 256, ln).astype(np.uint8) if g.random() < 0.7 \
                else g.integers(97, 101, ln).astype(np.uint8)
            pos += ln
    return out


def _silesia(n, g, seed):
    if n == 0:
        return np.zeros(0, np.uint8)
    out = []
    tot = 0
    k = 0
    while tot < n:
        seg = int(g.integers(4096, 65537))
        r = g.random()
        kind = "text" if r < 0.55 else "runs" if r < 0.70 else "records" if r < 0.90 else "rand"
        out.append(gen(kind, seg, seed * 1000003 + k))
        tot += seg
        k += 1
    return np.concatenate(out)[:n]


def gen(kind: str, n: int, seed: int = 1) -> np.ndarray:
    """uint8 array of length n."""
    g = _rng(seed, kind)
    if kind == "rand":
        return g.integers(0, 256, n).astype(np.uint8)
    if kind == "allA":
        return np.full(n, 65, np.uint8)
    if kind == "mod200":
        return (np.arange(n, dtype=np.int64) % 200).astype(np.uint8)
    if kind == "runs":
        return _runs(n, g)
    if kind == "text":
        return _text(n, g)
    if kind == "records":
        return _records(n, g)
    if kind == "lzmix":
        return _lzmix(n, g)
    if kind == "silesia":
        return _silesia(n, g, seed)
    raise ValueError(kind)


def gen_bytes(kind, n, seed=1) -> bytes:
    return gen(kind, n, seed).tobytes()


def sha(b) -> str:
    return hashlib.sha256(bytes(b)).hexdigest()
