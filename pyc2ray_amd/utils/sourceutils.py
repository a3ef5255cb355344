"""Source-list helpers with the reference's names and conventions
(pyc2ray/utils/sourceutils.py:7-112)."""
import numpy as np

__all__ = ["format_sources", "generate_test_sourcefile", "read_test_sources"]


def format_sources(source_pos, source_flux):
    """(3,Ns) 1-based positions -> flat int32 0-based [x0,y0,z0,x1,...]; flux -> float64.
    Same layout as pyc2ray/utils/sourceutils.py:30-31."""
    pos = np.asarray(source_pos)
    if pos.ndim != 2 or pos.shape[0] != 3:
        raise ValueError(f"source_pos must have shape (3, numsrc), got {pos.shape}")
    source_pos_flat = np.ravel((pos - 1).astype('int32'), order='F')
    source_flux_flat = np.ascontiguousarray(np.asarray(source_flux).astype('float64'))
    if source_flux_flat.shape[0] != pos.shape[1]:
        raise ValueError("source_flux and source_pos disagree on the number of sources")
    return source_pos_flat, source_flux_flat


def generate_test_sourcefile(filename, N, numsrc, strength, seed=100):
    """Equal-strength sources at random 1-based grid positions, in the text format of the original
    C2-Ray (pyc2ray/utils/sourceutils.py:35-68): a count line, then `x y z flux 0.0` rows."""
    rng = np.random.RandomState(seed)
    srcpos = 1 + rng.randint(0, N, size=3 * numsrc)
    srcpos = srcpos.reshape((numsrc, 3), order='C')
    rows = np.hstack((srcpos, strength * np.ones((numsrc, 1)), np.zeros((numsrc, 1))))
    with open(filename, 'w') as f:
        f.write(f"{numsrc:n}\n")
        np.savetxt(f, rows, ("%i %i %i %.0e %.1f"))


def read_test_sources(file, numsrc, S_star_ref=1e48):
    """Read `numsrc` sources of a Test-C2Ray source file (pyc2ray/utils/sourceutils.py:70-112).
    Returns (src_pos (3,numsrc), src_flux normalised by S_star_ref)."""
    with open(file, "r") as f:
        inp = np.loadtxt(f, skiprows=1, usecols=(0, 1, 2, 3), ndmin=2)
    max_n = inp.shape[0]
    if numsrc > max_n:
        raise ValueError(f"Number of sources given ({numsrc:n}) is larger than that of the file ({max_n:n})")
    return np.transpose(inp[:numsrc, 0:3]), inp[:numsrc, 3] / S_star_ref
