#!/usr/bin/env python3
"""Developer model (CPU only, no GPU, no library): how serial is zlib's level-1 parse really, window by window?

K1 today hides the serial parse behind 64-position windows and pays for it with a 1 MiB candidate table per chunk in
flight (4096 of them: HBM-resident, one random 64-byte fetch per position).  Keeping the table on chip (head[] in LDS,
prev[] in L2) means ONE chunk per CU, so the parse inside a chunk must be found in parallel over much wider windows.
The obstacle is that which positions zlib inserts into its hash chains depends on the parse itself.  This model measures
the obvious way round it: assume an inserted set for the window, find every position's match against the candidates that
set implies (all positions at once), parse greedily, derive the inserted set that parse implies, repeat until it stops
changing.  Positions before the window are exact (earlier windows are final).  It reports, for windows of W positions:
how many rounds until the fixpoint, and how many positions changed their match after round 1.

usage: k1_fixpoint_model.py [W=1024] [chunks=6] [kind=silesia]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402

MAXDIST, NICE, MAXINS, CHAIN = 32506, 8, 4, 4


def hash3(d, p):
    return ((d[p] << 12) ^ (d[p + 1] << 6) ^ d[p + 2]) & 0xffff


def best_match(d, n, p, cands):
    """zlib longest_match at level 1 for the candidate list (newest first, already cut to the chain length)"""
    best, bpos = 2, -1
    maxlen = min(258, n - p)
    nice = min(NICE, maxlen)
    for q in cands:
        ln = 0
        while ln < maxlen and d[q + ln] == d[p + ln]:
            ln += 1
        if ln > best:
            best, bpos = ln, q
            if ln >= nice:
                break
    return (best, bpos) if best >= 3 else (0, -1)


def run(W, nchunks, kind):
    data = datagen.gen_bytes(kind, nchunks * 65536, 4242)
    tot_rounds = tot_windows = tot_changed = tot_pos = 0
    hist = {}
    for c in range(nchunks):
        d = data[c * 65536:(c + 1) * 65536]
        n = len(d)
        chains = {}                       # hash -> list of inserted positions, oldest first (exact, for everything before the window)
        pos = 0
        while pos < n:
            wend = min(pos + W, n)
            H = [hash3(d, p) if p + 3 <= n else -1 for p in range(pos, wend)]
            ins = set(range(pos, wend))   # round 0: every position of the window inserted
            prev_parse = None
            rounds = 0
            first_m = None
            while True:
                rounds += 1
                # in-window inserted positions by hash, in order
                local = {}
                for p in sorted(ins):
                    h = H[p - pos]
                    if h >= 0:
                        local.setdefault(h, []).append(p)
                mlen = {}
                for p in range(pos, wend):
                    h = H[p - pos]
                    if h < 0:
                        mlen[p] = 0
                        continue
                    cand = [q for q in local.get(h, []) if q < p][-CHAIN:][::-1]
                    if len(cand) < CHAIN:
                        cand += chains.get(h, [])[::-1][:CHAIN - len(cand)]
                    cand = [q for q in cand if 0 < p - q <= MAXDIST and q != 0]      # NIL = position 0
                    mlen[p] = best_match(d, n, p, cand)[0]
                # greedy parse of the window from its (exact) start
                parse, newins, p = [], set(), pos
                while p < wend:
                    parse.append(p)
                    m = mlen[p]
                    if p + 3 <= n:
                        newins.add(p)
                    if m:
                        if m <= MAXINS:
                            for k in range(1, m):
                                if p + k + 3 <= n and p + k < wend:
                                    newins.add(p + k)
                        p += m
                    else:
                        p += 1
                if first_m is None:
                    first_m = dict(mlen)
                if newins == ins and parse == prev_parse:
                    break
                prev_parse, ins = parse, newins
                if rounds > 40:
                    break
            changed = sum(1 for q in parse if mlen[q] != first_m[q])
            tot_rounds += rounds; tot_windows += 1; tot_changed += changed; tot_pos += wend - pos
            hist[rounds] = hist.get(rounds, 0) + 1
            # commit the exact insertions of this window, advance to the parse point behind it
            last = parse[-1]
            nxt = last + (mlen[last] if mlen[last] else 1)
            for q in sorted(ins):
                if q < nxt and q + 3 <= n:
                    chains.setdefault(H[q - pos], []).append(q)
            pos = max(nxt, wend) if nxt >= wend else wend
    print("W=%d kind=%s: %d windows, %.2f rounds per window on average (to the fixpoint, the last one only confirms), "
          "%.1f %% of the parse points changed their match after round 1" % (W, kind, tot_windows, tot_rounds / tot_windows,
                                                                             100.0 * tot_changed / max(1, tot_pos)))
    print("   rounds histogram:", dict(sorted(hist.items())))


if __name__ == "__main__":
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    nch = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    kind = sys.argv[3] if len(sys.argv) > 3 else "silesia"
    run(W, nch, kind)
