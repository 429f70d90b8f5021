"""What capping the traversal per launch and regrouping the unfinished rays would buy: from a dump of node visits per
closest-hit ray in queue order (TRHIP_DUMP_VIS=file, TRHIP_LANES=1), compare the wave phases of the present kernel
(sum over 64-ray chunks of the longest ray) with rounds of K phases after which survivors are compacted into new waves."""
import sys
import numpy as np
v = np.fromfile(sys.argv[1], dtype=np.uint32)
bounces = int(sys.argv[2]) if len(sys.argv) > 2 else 4
v = v.reshape(bounces, -1)
for b in range(bounces):
    L = v[b][v[b] > 0].astype(np.int64) - 1
    n = len(L)
    if n == 0: continue
    pad = (-n) % 64
    Lp = np.concatenate([L, np.zeros(pad, np.int64)]).reshape(-1, 64)
    now = Lp.max(1).sum()
    ideal = L.sum() / 64.0
    out = [f"bounce {b}: rays {n}, mean {L.mean():.1f}, p50 {np.median(L):.0f}, p90 {np.percentile(L, 90):.0f}, p99 {np.percentile(L, 99):.0f}, max {L.max()}",
           f"   wave phases now {now} ({now / (n / 64):.1f} per chunk), ideal {ideal:.0f} ({ideal / (n / 64):.1f} per chunk)"]
    for caps in ((16, 16, 32, 10**9), (24, 24, 10**9), (32, 10**9), (12, 12, 12, 24, 10**9), (8, 8, 8, 8, 16, 32, 10**9)):
        rem = L.copy(); total = 0; rounds = []
        for K in caps:
            if len(rem) == 0: break
            pad = (-len(rem)) % 64
            R = np.concatenate([rem, np.zeros(pad, np.int64)]).reshape(-1, 64)
            total += np.minimum(R.max(1), K).sum()
            rounds.append(len(rem))
            rem = rem[rem > K] - K
        out.append(f"   caps {caps[:-1]}+rest: phases {total} ({now / total:.2f}x fewer), rays per round {rounds}")
    print("\n".join(out))
