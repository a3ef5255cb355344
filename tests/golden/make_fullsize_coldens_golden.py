"""Column densities at the FULL benchmark size, from the REFERENCE ITSELF (north_star: "ionized fraction AND column
density within 1e-5").

Run in the build container (where /root/reference exists):

    make -C oracle
    python tests/golden/make_fullsize_coldens_golden.py

What the reference's CPU function hands back in `coldensh_out` is the outgoing column density of every cell of the LAST
source's sub-box (ref: src/c2ray/raytracing.f90:181 zeroes the grid per source, :488 stores the cell's value): the cube of
+-r_RT cells around that source for the call of raytracing_benchmark/run_test.py:88 (one sub-box of r_RT cells, R_max_LLS =
r_RT, loss_fraction = 0).  Because the grid is zeroed per source, the result does not depend on the sources before the last:
the fixture is made from a call with the last source alone, and for the configs[2] workload at r_RT = 32 the whole
1000-source call is made as well and must return the same bits (asserted here).

Three cases, all 256^3 with the Teff = 1e5 K table of bench.py:
  u32, u64   BASELINE configs[2], uniform medium, last of the 1000 RandomState(100) sources, r_RT = 32 / 64
  c32        BASELINE configs[3], log-normal density, last of the 1000 sources on the densest cells, r_RT = 32
Stored per case (the cube has (2 r + 1)^3 cells; offsets are source-relative, periodic wrap as the reference's):
  <case>_vals        column density at 20 000 seeded offsets of the cube (regenerated from the seed at test time)
  <case>_plane_sums  sum over every di-plane of the cube (2 r + 1 values)      -- with block_sums a checksum over ALL its cells
  <case>_row_sums    sum over every (di, dj) row of the cube ((2 r + 1)^2 values)
  <case>_total, <case>_nonzero, <case>_src  (the last source's 1-based position)
Data only: inputs are regenerated from seeds, nothing of the reference's source text is kept.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, ".."))

N, NS, NSAMPLE = 256, 1000, 20000
CASES = {"u32": ("uniform", 32), "u64": ("uniform", 64), "c32": ("cosmo", 32)}


def sample_offsets(case):
    """Seeded (3, NSAMPLE) offsets in [-r, r]^3 the fixture holds values for."""
    r = CASES[case][1]
    rng = np.random.default_rng(20261000 + sum(map(ord, case)))
    return rng.integers(-r, r + 1, size=(3, NSAMPLE))


def cube_of(grid, src, r):
    """The (2r+1)^3 cube of `grid` (logical [i,j,k], any storage order) around the 1-based position `src`, periodic."""
    n = grid.shape[0]
    rng1 = np.arange(-r, r + 1)
    ii, jj, kk = ((int(src[a]) - 1 + rng1) % n for a in range(3))
    return np.ascontiguousarray(grid[np.ix_(ii, jj, kk)])


def digest(cube, case):
    """cube: (2r+1,)*3 indexed by offset + r -> what the fixture stores / the tests compare."""
    r = CASES[case][1]
    o = sample_offsets(case) + r
    return {f"{case}_vals": cube[o[0], o[1], o[2]], f"{case}_plane_sums": cube.sum(axis=(1, 2)),
            f"{case}_row_sums": cube.sum(axis=2).ravel(), f"{case}_total": np.array(float(cube.sum())),
            f"{case}_nonzero": np.array(int(np.count_nonzero(cube)))}


def main():
    import bench
    from oracle import ref_fortran as F
    assert F.available(), "build oracle/_ref first: make -C oracle"
    thin, thick, dlog = bench.make_tables()
    out = dict(table_sums=np.array([thin.sum(), thick.sum()]))
    workloads = {}
    for case, (kind, r) in CASES.items():
        if kind not in workloads:
            workloads[kind] = bench.make_workload(kind, N, NS)
        ndens, xh, temp, dr, pos, flux = workloads[kind]
        call = lambda p, f: F.do_all_sources(f, p, max_subbox=r, subboxsize=r, sig=bench.SIG, dr=dr, ndens=ndens, xh_av=xh,
                                             loss_fraction=0.0, thin=thin, thick=thick, minlogtau=bench.MINLOGTAU, dlogtau=dlog,
                                             R_max_LLS=float(r), NumTau=thin.shape[0] - 1)
        t0 = time.time()
        res = call(pos[:, -1:], flux[-1:])
        assert res["nsubbox"] == 1
        cd = res["coldens"]
        cube = cube_of(cd, pos[:, -1], r)
        assert np.count_nonzero(cd) == np.count_nonzero(cube) == (2 * r + 1) ** 3      # nothing outside the cube, every cell inside
        print(f"{case}: last source alone {time.time() - t0:.0f} s, total {cube.sum():.6e}", flush=True)
        if case == "u32":       # the whole call, as the test makes it: same bits
            t0 = time.time()
            full = call(pos, flux)["coldens"]
            assert np.array_equal(full, cd), "coldensh_out of the 1000-source call differs from the last source alone"
            print(f"{case}: 1000-source call {time.time() - t0:.0f} s, identical", flush=True)
        out.update(digest(cube, case))
        out[f"{case}_src"] = pos[:, -1].astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "fullsize_coldens.npz"), **out)


if __name__ == "__main__":
    main()
