"""k_band_tiles, when may an idle reserve tile leave?  (csrc/lqr_hip.hip, the wait loop of k_band_tiles; DESIGN.md 4.15 lesson vi.)
An exhaustive interleaving check of the header protocol on a small model -- CPU only, no engine code involved:

  * hdr[1] = tickets drawn (fetch-add by a RUNNING tile that wants a neighbour), then the request word for that ticket is
    written (a second step); hdr[0] = tiles that have ended (base tiles and reserves that ran);
  * reserve r polls: its request word; else k0 = hdr[1]; if k0 <= r: fin = hdr[0]; k1 = hdr[1]; it leaves iff k1 == k0 and
    fin >= t_base + k0.  The three reads are separate steps (other tiles move in between).

Property: no ticket is ever drawn for a reserve tile that has left (its asker would wait for a hand-over that never comes).
The first version's rule -- leave when all BASE tiles have ended -- is checked too and must FAIL the property: that is the
time-out the fuzz run found (a reserve tile two hops out asks after the base tiles are done)."""
import sys


def explore(t_base, n_rsv, asks_per_tile, rule):
    """DFS over all interleavings; returns (property holds everywhere, some reserve tile did leave somewhere, states seen)."""
    # runner state: (asks_left, pending_word_ticket or -1, ended)
    # reserve state: ("wait", phase, k0, fin) | ("run", asks_left, pending, ended) | ("left",)
    init = (0, 0, tuple((asks_per_tile, -1, False) for _ in range(t_base)), tuple(("wait", 0, 0, 0) for _ in range(n_rsv)), tuple([False] * n_rsv))
    seen, stack = {init}, [init]
    left_seen = False
    while stack:
        K, F, base, rsv, words = stack.pop()
        succ = []

        def runner_moves(st, put):
            asks, pend, ended = st
            if ended:
                return
            if pend >= 0:                                   # the request word follows its ticket
                w = list(words); w[pend] = True
                succ.append(put((asks, -1, False), K, F, tuple(w)))
                return
            if asks > 0:
                if K < n_rsv:                               # ticket K is reserve K's
                    if rsv[K][0] == "left":
                        return "VIOLATION"
                    succ.append(put((asks - 1, K, False), K + 1, F, words))
                else:
                    succ.append(put((asks - 1, -1, False), K + 1, F, words))      # no reserve left: the counter still moves
            succ.append(put((asks, -1, True), K, F + 1, words))                   # the tile ends (and counts itself)

        for i, st in enumerate(base):
            def put(new, K2, F2, w2, i=i):
                b = list(base); b[i] = new
                return (K2, F2, tuple(b), rsv, w2)
            if runner_moves(st, put) == "VIOLATION":
                return False, left_seen, len(seen)
        for r, st in enumerate(rsv):
            def putr(new, K2=K, F2=F, w2=words, r=r):
                x = list(rsv); x[r] = new
                return (K2, F2, base, tuple(x), w2)
            if st[0] == "run":
                def put(new, K2, F2, w2, r=r):
                    return putr(("run",) + new, K2, F2, w2, r)
                if runner_moves(st[1:], put) == "VIOLATION":
                    return False, left_seen, len(seen)
            elif st[0] == "wait":
                _, phase, k0, fin = st
                if phase == 0:                              # look at the request word, then at the ticket count
                    if words[r]:
                        succ.append(putr(("run", asks_per_tile, -1, False)))
                    elif rule == "base_done":
                        if F >= t_base and all(b[2] for b in base):
                            succ.append(putr(("left",)))    # (the first version looked once more at its word: same step here)
                    else:
                        succ.append(putr(("wait", 1, K, 0)))
                elif phase == 1:
                    succ.append(putr(("wait", 0, 0, 0)) if k0 > r else putr(("wait", 2, k0, F)))
                elif phase == 2:
                    succ.append(putr(("left",)) if (K == k0 and fin >= t_base + k0) else putr(("wait", 0, 0, 0)))
        for s in succ:
            left_seen = left_seen or any(x[0] == "left" for x in s[3])
            if s not in seen:
                seen.add(s); stack.append(s)
    return True, left_seen, len(seen)


def test_leaving_when_every_started_tile_has_ended_is_safe():
    for t_base in (1, 2):
        for n_rsv in (1, 2, 3):
            ok, left, states = explore(t_base, n_rsv, 1, "all_started_done")
            assert ok and left and states > 20, (t_base, n_rsv, ok, left, states)      # safe, and reserves do get away
    assert explore(1, 2, 2, "all_started_done")[0]


def test_leaving_when_the_base_tiles_have_ended_is_not():
    """a woken reserve tile may still ask after the last base tile ended (in the kernel: two hops out, second-to-last block)"""
    assert not explore(1, 2, 1, "base_done")[0]


if __name__ == "__main__":
    print(explore(2, 3, 1, "all_started_done"), explore(1, 2, 1, "base_done"))
    sys.exit(0)
