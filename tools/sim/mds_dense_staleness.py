"""Replay of the sampler's DENSE regime (cut ball covers the cloud: every pick changes every density): how many
consecutive picks could a bounded-staleness scheme take per exchange?

Scheme: every member publishes its K lowest candidates (density, coordinates); a leader replays
picks among the G x K candidates -- exact, since a candidate's new density needs only its old density and the picks'
coordinates -- for as long as the picked minimum stays below the smallest STALE density outside the set plus the
smallest increment any outside point can have received (densities only grow).  This script measures, on the data of
the dense regime (uniform cloud, n = 19384, mml 0.05 / 0.0853, float64 arithmetic -- statistics only):
  (a) run length of consecutive true picks that lie inside the set of the C lowest points at the time of the exchange
      (an upper bound for ANY scheme that exchanges C candidates), and
  (b) the picks the exact staleness test accepts.
  (c) the same with the candidates as the TEAM kernel would have them: G members own contiguous stretches of the
      Morton-sorted cloud (spatially coherent chunks), each publishes its K lowest; the bound of what a member did
      not publish is its (K + 1)-th lowest stale density (+ the smallest possible increments).
Usage: python tools/sim/mds_dense_staleness.py [rounds]
"""
import sys
import numpy as np

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(7)
n = 19384
x = rng.random((n, 3))
wgt = np.where(np.arange(n) < 8192, 1.0, 2.0)
for mml in (0.05, 0.0853, 0.03):
    t = 5.0 * mml * mml
    dens = np.zeros(n)
    picked = np.zeros(n, bool)
    last = 0
    picked[0] = True
    seq = [0]
    snaps = {}
    # plain sequential replay, remembering the density vector at every round
    for j in range(1, rounds):
        d = ((x - x[last]) ** 2).sum(1)
        dens = dens + wgt * np.exp(-d / t)
        dd = np.where(picked, 1e9, dens)
        last = int(np.argmin(dd))
        if j % 50 == 0 and j + 40 < rounds:
            snaps[j] = (dens.copy(), picked.copy())   # the state in which pick j is taken (not yet marked)
        picked[last] = True
        seq.append(last)
    for C in (16, 64, 256):
        runs, accepted = [], []
        for j, (dn, pk) in snaps.items():
            # state AFTER pick j was chosen (seq[j]) but before its update: the exchange happens here
            dd = np.where(pk, 1e9, dn)
            order = np.argsort(dd)
            cand = order[:C]
            stale_out = dd[order[C]]                       # smallest stale density outside the set
            inset = set(cand.tolist())
            run = 0
            cd = {int(c): dd[c] for c in cand}             # replayed candidate densities
            acc = 0
            ok = True
            floor_out = stale_out
            for q in range(0, 32):                         # picks j, j+1, ... (seq[j] is the minimum of dd)
                pk_id = seq[j + q]
                if pk_id not in inset:
                    break
                run += 1
                if ok:
                    # exact test: the candidates' minimum must be below the lower bound of everything outside
                    cur = min(cd.items(), key=lambda kv: kv[1])
                    if cur[0] == pk_id and cur[1] < floor_out:
                        acc += 1
                    else:
                        ok = False
                # apply the pick to the candidates and to the outside bound
                p = x[pk_id]
                for c in list(cd):
                    if c == pk_id:
                        cd[c] = 1e9
                    else:
                        cd[c] += wgt[c] * np.exp(-((x[c] - p) ** 2).sum() / t)
                # outside points gained at least the smallest possible increment: the farthest corner of the unit cube
                far2 = (np.maximum(p, 1 - p) ** 2).sum()
                floor_out += np.exp(-far2 / t)
            runs.append(run)
            accepted.append(acc)
        print(f"mml {mml}: C = {C:3d} candidates per exchange: consecutive true picks inside the set: mean {np.mean(runs):.2f} "
              f"(max {max(runs)}); picks the exact staleness test accepts: mean {np.mean(accepted):.2f} (max {max(accepted)})"
              f"   [{len(runs)} exchanges sampled over {rounds} rounds]")


    # (c) per-member candidates: G members, contiguous chunks of the Morton order
    def morton(q):
        v = np.zeros(len(q), np.int64)
        for b in range(10):
            for a in range(3):
                v |= ((q[:, a] >> b) & 1) << (3 * b + a)
        return v
    order_m = np.argsort(morton((x * 1023).astype(np.int64)), kind="stable")
    for G in (16, 32):
        member = np.empty(n, np.int64)
        member[order_m] = np.arange(n) * G // n
        for K in (1, 2, 4, 8):
            accepted = []
            for j, (dn, pk) in snaps.items():
                dd = np.where(pk, 1e9, dn)
                cd, bound = {}, np.inf
                for g_ in range(G):
                    idx = np.nonzero(member == g_)[0]
                    o = idx[np.argsort(dd[idx])]
                    for c in o[:K]:
                        cd[int(c)] = dd[c]
                    bound = min(bound, dd[o[K]])
                acc = 0
                for q in range(0, 64):
                    cur = min(cd.items(), key=lambda kv: kv[1])
                    if not (cur[1] < bound) or cur[0] != seq[j + q]:
                        break
                    acc += 1
                    p = x[cur[0]]
                    for c in list(cd):
                        cd[c] = 1e9 if c == cur[0] else cd[c] + wgt[c] * np.exp(-((x[c] - p) ** 2).sum() / t)
                    bound += np.exp(-(np.maximum(p, 1 - p) ** 2).sum() / t)
                    if j + q + 1 >= len(seq):
                        break
                accepted.append(acc)
            print(f"mml {mml}: team of G = {G} members, K = {K} candidates each: picks per exchange mean {np.mean(accepted):.2f} "
                  f"(min {min(accepted)}, max {max(accepted)})")
