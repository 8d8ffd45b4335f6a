"""Developer experiment (CPU, numpy): how tight is a Lagrangian bound on the exact solver's model? Reads the model that
`python3 scripts/exact_probe.py --dump <pairs>` leaves in gpurun_out/exact_model.txt (one chain: free reads, windows, depth table),
dualises the window counts and minimises the bound by subgradient steps. At 10 000 read pairs x 8 alleles: gap 1.6e-2 -> 6.2e-4."""
import numpy as np, sys, time
lines = open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/exact_model.txt').read().split('\n')
n, tw, ld, a, dc, aln_fixed = lines[0].split(); n, tw, ld = int(n), int(tw), int(ld); a, dc, aln_fixed = float(a), float(dc), float(aln_fixed)
lo = np.zeros(tw, dtype=np.int64); cap = np.zeros(tw, dtype=np.int64); ww = np.zeros(tw); gc = np.zeros(tw, dtype=np.int64)
reads = []; lut = {}
wi = 0
for ln in lines[1:]:
    if not ln: continue
    p = ln.split()
    if p[0] == 'W':
        lo[wi], cap[wi], ww[wi], gc[wi] = int(p[1]), int(p[2]), float(p[3]), int(p[4]); wi += 1
    elif p[0] == 'R':
        fixed, nl = int(p[1]), int(p[2])
        locs = [(float(p[3 + 3 * t]), int(p[4 + 3 * t]), int(p[5 + 3 * t])) for t in range(nl)]
        reads.append((fixed, locs))
    elif p[0] == 'L':
        lut[int(p[1])] = np.array([float(x) for x in p[2:]])
    elif p[0] == 'I':
        incumbent = float(p[1])
free = [locs for f, locs in reads if not f]
print("free reads", len(free), "windows", tw, "incumbent", incumbent, "max locs", max(len(l) for l in free))
# v_w(k) for k in 0..cap_w
V = []
for w in range(tw):
    if ww[w] == 0.0: V.append(np.zeros(cap[w] + 1))
    else: V.append(dc * ww[w] * lut[gc[w]][lo[w]:lo[w] + cap[w] + 1])
# arrays for 2-location reads (the bulk); handle general with loops over t
T = max(len(l) for l in free)
nf = len(free)
LP = np.full((nf, T), -1e300); WA = np.zeros((nf, T), dtype=np.int64); WB = np.zeros((nf, T), dtype=np.int64)
for i, locs in enumerate(free):
    for t, (lp, wa, wb) in enumerate(locs):
        LP[i, t], WA[i, t], WB[i, t] = a * lp, wa, wb
const = a * aln_fixed
def dual(lam):
    val = LP + lam[WA] + lam[WB]
    tstar = np.argmax(val, axis=1)
    rsum = val[np.arange(nf), tstar].sum()
    cnt = np.zeros(tw)
    np.add.at(cnt, WA[np.arange(nf), tstar], 1); np.add.at(cnt, WB[np.arange(nf), tstar], 1)
    wsum = 0.0; kstar = np.zeros(tw)
    for w in range(tw):
        x = V[w] - lam[w] * np.arange(cap[w] + 1)
        k = int(np.argmax(x)); kstar[w] = k; wsum += x[k]
    return const + rsum + wsum, cnt - kstar, tstar
def primal(tstar):
    cnt = np.zeros(tw, dtype=np.int64)
    np.add.at(cnt, WA[np.arange(nf), tstar], 1); np.add.at(cnt, WB[np.arange(nf), tstar], 1)
    return const + LP[np.arange(nf), tstar].sum() + sum(V[w][cnt[w]] for w in range(tw))
lam = np.zeros(tw)
best_ub = np.inf; best_lb = incumbent
t0 = time.time()
theta = 1.0; stall = 0
for it in range(4000):
    ub, g, tstar = dual(lam)
    if ub < best_ub - 1e-9: best_ub = ub; stall = 0
    else:
        stall += 1
        if stall >= 20: theta *= 0.7; stall = 0
    if it % 50 == 0:
        lb = primal(tstar); best_lb = max(best_lb, lb)
    if it % 200 == 0:
        print(it, "ub", round(best_ub, 4), "lb", round(best_lb, 4), "gap", (best_ub - best_lb) / abs(best_lb), "theta", round(theta, 4), "t", round(time.time() - t0, 1))
    nrm = (g * g).sum()
    if nrm == 0: break
    lam = lam - theta * (ub - best_lb) / nrm * g
print("final ub", best_ub, "lb", best_lb, "gap", (best_ub - best_lb) / abs(best_lb))
