"""Would double-double arithmetic in the CROSS-thread stages make the refinement's Schur sums independent of the tiling?  

The streaming passes give every thread ~7 inliers (inlier i of a shard goes to thread i mod 131072 of that shard's launch) and add them in
plain fp64; the per-thread sums are then combined across waves, workgroups and ranks.  This script takes the per-inlier terms of ONE Schur
slot on realistic data (F_a F_b products of the refinement's Jacobian at 1280x720), forms the per-thread partial sums under two
tilings -- one rank, and eight ranks (each rank's threads stride over its own contiguous shard) -- and combines them EXACTLY (math.fsum:
the limit of what a double-double combination can deliver).  Result: the per-thread sums are ~1e5 times smaller than the total, so their
own rounding errors amount to ~4e-19 of it and the exactly combined totals agree to the last bit in most slots -- but not in all: a total
whose exact value lies within ~1/250 ulp of a rounding boundary comes out one ulp apart.  With 54 / 70 slots per pass and two passes per LM
iteration a solve forms a few hundred such sums, so error-free combination stages would shrink the tiled-vs-single differences from
~1e-13 relative to "at most one ulp, in a few sums per solve" -- not to zero: bit-for-bit equality between tilings needs every per-inlier
addition in double-double as well (6 flops instead of 1 on each accumulator of a pass that is bound by fp64 issue: about twice its time),
and a refinement whose trajectory splits on the last bit of a sum ("chaotic refinements") splits on one ulp too."""
import math
import sys

import numpy as np


def per_thread_partials(x, nthreads):
    """plain fp64 accumulation in index order, inlier i -> thread i mod nthreads"""
    n = len(x)
    pad = (-n) % nthreads
    xp = np.concatenate([x, np.zeros(pad)]).reshape(-1, nthreads)
    acc = np.zeros(nthreads)
    for row in xp:  # sequential adds per thread, like the kernel's loop
        acc = acc + row
    return acc


def main():
    rng = np.random.default_rng(4)
    n = 921600
    # F_a F_b-like terms: products of two smooth image-coordinate polynomials times beta^2, positive and negative (off-diagonal slots)
    x_ = rng.uniform(-0.64, 0.64, n)
    y_ = rng.uniform(-0.37, 0.37, n)
    beta = 1.0 + 0.02 * rng.standard_normal(n)
    slots = {"F3.F3": (beta * (1 + y_ * y_)) ** 2, "F3.F4": -(beta * beta) * (1 + y_ * y_) * (x_ * y_), "F0.F2": -(beta * beta) * x_,
             "F5.F5": (beta * x_) ** 2 + (beta * y_) ** 2}
    nthreads = 131072
    print("# per-thread fp64 partial sums combined EXACTLY (math.fsum), 1 rank vs 8 ranks, n = %d inliers, %d threads per launch" % (n, nthreads))
    for name, x in slots.items():
        exact = math.fsum(x.tolist())
        one = math.fsum(per_thread_partials(x, nthreads).tolist())
        shards = np.array_split(x, 8)
        eight = math.fsum(sum((per_thread_partials(s_, nthreads).tolist() for s_ in shards), []))
        ulp = np.spacing(abs(exact))
        print("%-6s exact %.17g   1 rank: %+.1f ulp   8 ranks: %+.1f ulp   1 rank vs 8 ranks: %+.1f ulp (%.1e relative)"
              % (name, exact, (one - exact) / ulp, (eight - exact) / ulp, (one - eight) / ulp, abs(one - eight) / abs(exact)))
    print("# => error-free combination stages make the tilings agree in most sums and leave single ulps in the others (the per-thread sums differ)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
