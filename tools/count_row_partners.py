#!/usr/bin/env python3
"""Gate for the on-chip pre-summing of row-adjacent paired sources: of the sources of
BASELINE configs[3] (256^3, 1000 on the densest cells) and configs[4] (512^3, 1e5), how many have a partner on the
same grid ROW within delta <= 7 cells -- same (i, j) and |dk| <= 7 for the units of the x- and y-faces (rows along k),
same (j, k) and |di| <= 7 for the units of the z-faces (rows along i, the [k][j][i] twin).  Greedy pairing in row order,
as a pairing pass on the host would do it.  Host only (numpy).

  python tools/count_row_partners.py [--delta 7]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def greedy_pairs(pos, row_axes, along, delta):
    """pos (3, n) 0-based.  Sources on one row (equal coordinates on row_axes), sorted along `along`; a source pairs with
    the next unpaired one if it lies within delta (and is not on the same cell... equal cells pair too: delta = 0)."""
    key = np.lexsort((pos[along], pos[row_axes[1]], pos[row_axes[0]]))
    p = pos[:, key]
    n = p.shape[1]
    paired = 0
    hist = np.zeros(delta + 1, dtype=np.int64)
    q = 0
    while q + 1 < n:
        same_row = p[row_axes[0], q] == p[row_axes[0], q + 1] and p[row_axes[1], q] == p[row_axes[1], q + 1]
        d = int(p[along, q + 1] - p[along, q])
        if same_row and d <= delta:
            paired += 2
            hist[d] += 1
            q += 2
        else:
            q += 1
    return paired, hist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--delta", type=int, default=7)
    args = ap.parse_args()
    out = []
    for label, N, ns in (("configs[3]", 256, 1000), ("configs[4]", 512, 100000)):
        _, _, _, _, pos, _ = bench.make_workload("cosmo", N, ns)
        pos0 = np.asarray(pos) - 1
        pk, hk = greedy_pairs(pos0, (0, 1), 2, args.delta)
        pi, hi = greedy_pairs(pos0, (1, 2), 0, args.delta)
        rec = {"config": label, "N": N, "sources": ns, "delta_max": args.delta,
               "paired_along_k_xy_face_units": pk, "frac_k": pk / ns, "pairs_by_delta_k": hk.tolist(),
               "paired_along_i_z_face_units": pi, "frac_i": pi / ns, "pairs_by_delta_i": hi.tolist()}
        out.append(rec)
        print(json.dumps(rec))
    return out


if __name__ == "__main__":
    main()
