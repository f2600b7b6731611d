"""Lane utilisation of the owner phase of the tile-sort Phi kernel under run-length balancing policies: every wave runs
its owner loops as long as its longest run, so utilisation = points served / (64 x sum over waves of the longest service count).
Cells get Poisson / multinomial arrivals of lam points per tile; a policy serves min(queue, C) points per cell and tile and carries the rest.
usage: python tools/owner_balance_sim.py"""
import numpy as np

rng = np.random.default_rng(0)


def run(lam, serve_fn, ntiles=600, ncell=2044):
    q = np.zeros(2048, int)
    slots = pts = 0
    left = []
    state = {}
    for t in range(ntiles):
        arr = np.zeros(2048, int)
        arr[:ncell] = rng.multinomial(int(lam * 2048), np.ones(ncell) / ncell)
        n = q + arr
        serve = serve_fn(n, q, arr, state)
        sA = serve[:1024].reshape(16, 64).max(1)
        sB = serve[1024:].reshape(16, 64).max(1)
        if t > 100:
            slots += (sA.sum() + sB.sum()) * 64
            pts += serve.sum()
            left.append((n - serve).sum())
        q = n - serve
    return pts / slots, float(np.mean(left)), int(np.max(left))


def uncapped(n, q, arr, st):
    return n


def capped(C, Lmax):
    """serve up to C per cell and tile; at most Lmax leftover points per cell may be carried (register-held leftovers)"""
    def f(n, q, arr, st):
        return np.maximum(np.minimum(n, C), np.minimum(np.maximum(n - Lmax, 0), n))
    return f


def two_buffers(C):
    """two alternating sort buffers: what a tile leaves over must be consumed in the NEXT tile (q = leftover of the older buffer)"""
    def f(n, q, arr, st):
        return np.maximum(q, np.minimum(n, C))
    return f


if __name__ == "__main__":
    for lam in (2, 3):
        u, _, _ = run(lam, uncapped)
        print("lambda %d: uncapped (the product)                       utilisation %.3f" % (lam, u))
        for C in (lam + 1, lam + 2):
            for Lmax in (1, 2, 3, 10 ** 6):
                u, l, lm = run(lam, capped(C, Lmax))
                print("lambda %d: cap %d, at most %7s carried per cell        utilisation %.3f  leftover points per tile: mean %.0f max %d"
                      % (lam, C, "any" if Lmax > 100 else str(Lmax), u, l, lm))
            u, l, lm = run(lam, two_buffers(C))
            print("lambda %d: cap %d, two alternating sort buffers             utilisation %.3f" % (lam, C, u))
