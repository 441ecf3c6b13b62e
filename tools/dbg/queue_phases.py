"""The three-phase structure of pytorch3d's unsorted K-queue, checked against a direct simulation of the queue
on random candidate sequences with many equal depths (CPU only):

    python tools/dbg/queue_phases.py [trials]

queue_sim   the reference's loop as oracle/raster_oracle.c restates it: the first K candidates fill the array in arrival order; afterwards a
            candidate strictly nearer than the array's farthest entry replaces it, and the new farthest is the FIRST slot holding the
            largest depth.
phase_pred  with T the K-th smallest depth: (1) fill; (2) while far entries (depth > T) remain, every near or tied arrival replaces the
            farthest far entry (lowest slot among equal depths) and inherits its slot, a far arrival only if nearer than that entry; (3) once
            none is left - after exactly K near-or-tied arrivals in all - tied arrivals are refused and each near arrival evicts the tied
            entry in the lowest slot.  Survivors: every near candidate + the tied entries present after phase 2 minus the lowest slots.
"""
import random
import sys


def queue_sim(cands, K):
    q, qmax, qidx = [], -1.0, -1
    for i, z in cands:
        if len(q) < K:
            q.append((z, i))
            if z > qmax:
                qmax, qidx = z, len(q) - 1
        elif z < qmax:
            q[qidx] = (z, i)
            qmax = -1.0
            for k, (zz, _) in enumerate(q):
                if zz > qmax:
                    qmax, qidx = zz, k
    return sorted(i for _, i in q)


def phase_pred(cands, K):
    if len(cands) <= K:
        return sorted(i for i, _ in cands)
    T = sorted(z for _, z in cands)[K - 1]
    near, tied, far, near_or_tied = [], {}, [], 0
    for s, (i, z) in enumerate(cands[:K]):
        if z < T:
            near.append(i)
        elif z == T:
            tied[i] = s
        else:
            far.append([z, s])
        near_or_tied += z <= T
    evictions = 0
    for i, z in cands[K:]:
        if far:
            far.sort(key=lambda t: (-t[0], t[1]))
            if z > T:
                if z < far[0][0]:
                    far[0][0] = z
            else:
                _, slot = far.pop(0)
                near_or_tied += 1
                if z < T:
                    near.append(i)
                else:
                    tied[i] = slot
        else:
            assert near_or_tied == K  # phase 3 begins after exactly K near-or-tied arrivals
            if z < T:
                near.append(i)
                evictions += 1
    order = sorted(tied.items(), key=lambda t: t[1])
    return sorted(near + [i for i, _ in order[evictions:]])


if __name__ == "__main__":
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    random.seed(1)
    bad = 0
    for _ in range(trials):
        K = random.choice([3, 5, 8, 16, 100])
        n = random.randint(K, 4 * K)
        levels = random.randint(2, 8)
        cands = [(i, float(random.randint(0, levels))) for i in range(n)]
        bad += queue_sim(cands, K) != phase_pred(cands, K)
    print(f"{trials} random candidate sequences: {bad} mismatches between the queue and its three-phase form")
    sys.exit(1 if bad else 0)
