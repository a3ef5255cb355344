"""The two files INTEGRATION.md section 2 tells a maintainer of the reference to add -- `pyc2ray/lib/libasora.py` and
`pyc2ray/lib/libc2ray.py` -- are EXECUTED here as printed: the code blocks are cut out of INTEGRATION.md, loaded as modules
and driven by the loop of the reference's evolve3D (ref: pyc2ray/evolve.py:139-245, restated below call for call: same
array preparation, same arguments in the same order, both branches) against tests/golden/evolve.npz, the outputs of the
reference's OWN evolve.py over the compiled reference Fortran.  A typo in an `argtypes` list or an argument order of the
printed stubs fails here.

CPU (not gpu): the blocks load, bind every symbol they name, and their argtypes lists have the arity of the prototypes
in include/asora_hip.h.   GPU: the loop.
"""
import os
import re
import types

import numpy as np
import pytest

import cases

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
G = os.path.join(HERE, "golden")
LIB = os.path.join(ROOT, "pyc2ray_amd", "lib", "libasora_hip.so")


def _stub_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    out = {}
    for b in blocks:
        m = re.match(r"# pyc2ray/lib/(libasora|libc2ray)\.py", b)
        if m:
            assert m.group(1) not in out, "one block per file"
            out[m.group(1)] = b
    assert set(out) == {"libasora", "libc2ray"}, "INTEGRATION.md section 2 must print both stub files"
    return out


@pytest.fixture(scope="module")
def stubs():
    """{name: module} -- each printed file executed as a module of its own, as `from .lib import libasora` would."""
    if not os.path.exists(LIB):
        pytest.fail("pyc2ray_amd/lib/libasora_hip.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    old = os.environ.get("ASORA_HIP_LIB")
    os.environ["ASORA_HIP_LIB"] = LIB
    mods = {}
    try:
        for name, src in _stub_blocks().items():
            mod = types.ModuleType(f"pyc2ray.lib.{name}")
            exec(compile(src, f"INTEGRATION.md:{name}.py", "exec"), mod.__dict__)
            mods[name] = mod
    finally:
        if old is None:
            os.environ.pop("ASORA_HIP_LIB", None)
        else:
            os.environ["ASORA_HIP_LIB"] = old
    return mods


def _prototype_arity(name):
    """Number of parameters of `name` in include/asora_hip.h."""
    header = open(os.path.join(ROOT, "include", "asora_hip.h")).read()
    m = re.search(r"\b" + name + r"\s*\(([^;{]*?)\)\s*;", header, flags=re.S)
    assert m, f"{name} is not declared in include/asora_hip.h"
    params = m.group(1).strip()
    return 0 if params in ("", "void") else len(params.split(","))


def test_printed_stubs_load_and_match_the_header(stubs):
    a, c = stubs["libasora"], stubs["libc2ray"]
    # the reference's call sites: asora_core.py:34,44,56, evolve.py:147,155,187, raytracing.py:63,71,89
    for fn in ("device_init", "device_close", "density_to_device", "photo_table_to_device", "source_data_to_device", "do_all_sources"):
        assert callable(getattr(a, fn))
    # evolve.py:190,210, raytracing.py:92, chemistry.py:91
    assert callable(c.chemistry.global_pass) and callable(c.raytracing.do_all_sources)
    for mod in (a, c):
        text = _stub_blocks()["libasora" if mod is a else "libc2ray"]
        named = set(re.findall(r"_l\.((?:asora|c2ray)_\w+)", text))
        assert named
        for sym in named:
            f = getattr(mod._l, sym)            # AttributeError: the printed stub names a symbol the library does not export
            if f.argtypes is not None:
                assert len(f.argtypes) == _prototype_arity(sym), sym
        # every call in the text passes as many arguments as the prototype has parameters
        for sym, args in re.findall(r"_ck\(_l\.((?:asora|c2ray)_\w+)\((.*?)\)\)\s*(?:#|$)", text, flags=re.S | re.M):
            depth, n, cur = 0, 0, ""
            for ch in args:
                if ch in "([":
                    depth += 1
                elif ch in ")]":
                    depth -= 1
                if ch == "," and depth == 0:
                    n += 1
                    cur = ""
                else:
                    cur += ch
            n += 1 if cur.strip() else 0
            assert n == _prototype_arity(sym), (sym, n)
    # intent(inout) arguments are checked like f2py checks them
    x = np.zeros((4, 4, 4), order="F")
    with pytest.raises(ValueError):
        c.chemistry.global_pass(1.0, x, x, x, x.astype(np.float32), x, x, 1.0, 1.0, 1.0, 1.0, 1.0)


def _reference_loop(libasora, libc2ray, dt, dr, src_flux, src_pos, use_gpu, max_subbox, subboxsize, loss_fraction, temp, ndens, xh,
                    photo_thin_table, photo_thick_table, minlogtau, dlogtau, R_max_LLS, convergence_fraction, sig, bh00,
                    albpow, colh0, temph0, abu_c):
    """The control flow and the extension-module calls of ref: pyc2ray/evolve.py:119-245, in its order (test infrastructure:
    what the stubs are called BY when they sit in the reference package)."""
    NumSrc = src_flux.shape[0]
    N = temp.shape[0]
    NumCells = N * N * N
    NumTau = photo_thin_table.shape[0]                                   # evolve.py:124
    conv_criterion = min(int(convergence_fraction * NumCells), (NumSrc - 1) / 3)
    prev1 = prev0 = 2 * NumCells
    converged, niter = False, 0
    xh_av, xh_intermed = np.copy(xh), np.copy(xh)                         # evolve.py:136-137 (keeps the order of xh)
    if use_gpu:
        xh_av_flat = np.ravel(xh).astype("float64", copy=True)            # evolve.py:142-143
        ndens_flat = np.ravel(ndens).astype("float64", copy=True)
        srcpos_flat = np.ravel((src_pos - 1).astype("int32"), order="F")  # sourceutils.py:30
        normflux_flat = src_flux.astype("float64")
        libasora.source_data_to_device(srcpos_flat, normflux_flat, NumSrc)
        coldensh_out_flat = np.ravel(np.zeros((N, N, N), dtype="float64"))
        phi_ion_flat = np.ravel(np.zeros((N, N, N), dtype="float64"))
        libasora.density_to_device(ndens_flat, N)
    rows = []
    while not converged:
        niter += 1
        if not use_gpu:
            phi_ion = np.zeros((N, N, N), order="F")
            phi_heat = np.zeros((N, N, N), order="F")
            coldensh_out = np.zeros((N, N, N), order="F")
        if use_gpu:
            libasora.do_all_sources(R_max_LLS, coldensh_out_flat, sig, dr, ndens_flat, xh_av_flat, phi_ion_flat, NumSrc, N, minlogtau,
                                    dlogtau, NumTau)
            phi_ion = np.reshape(phi_ion_flat, (N, N, N))
        else:
            nsubbox, photonloss = libc2ray.raytracing.do_all_sources(src_flux, src_pos, max_subbox, subboxsize, coldensh_out, sig, dr, ndens,
                                                                     xh_av, phi_ion, phi_heat, loss_fraction, photo_thin_table,
                                                                     photo_thick_table, np.zeros(NumTau), np.zeros(NumTau), minlogtau,
                                                                     dlogtau, R_max_LLS)
            assert nsubbox >= NumSrc and np.isfinite(photonloss)
        conv_flag = libc2ray.chemistry.global_pass(dt, ndens, temp, xh, xh_av, xh_intermed, phi_ion, bh00, albpow, colh0, temph0, abu_c)
        s1, s0 = np.sum(xh_intermed), np.sum(1.0 - xh_intermed)
        rel1 = np.abs((s1 - prev1) / s1) if s1 > 0.0 else 1.0
        rel0 = np.abs((s0 - prev0) / s0) if s0 > 0.0 else 1.0
        rows.append((conv_flag, rel1))
        converged = (conv_flag < conv_criterion) or ((rel1 < convergence_fraction) and (rel0 < convergence_fraction))
        prev1, prev0 = s1, s0
        if use_gpu and not converged:
            xh_av_flat = np.ravel(xh_av)
        assert niter < 200
    return xh_intermed, phi_ion, rows


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["l16_gpu_F", "l24_gpu_F_37src", "l16_cpu_F", "cfg0_64"])
def test_reference_loop_through_the_printed_stubs_reproduces_the_reference(stubs, name):
    import pyc2ray_amd as p
    golden = np.load(os.path.join(G, "evolve.npz"))
    libasora, libc2ray = stubs["libasora"], stubs["libc2ray"]
    c = cases.evolve_case(name)
    N = c["N"]
    if p.cuda_is_init():            # (the package's own binding shares the process-global state of the library)
        p.device_close()
    if c["use_gpu"]:
        libasora.device_init(N, 8)                                              # asora_core.py:34
        libasora.photo_table_to_device(c["thin"], c["thick"], c["thin"].shape[0])   # asora_core.py:54-56: NumTau = len(table)
    try:
        xh = c["xh"]
        for step in range(c["steps"]):
            x, phi, rows = _reference_loop(libasora, libc2ray, c["dt"], c["dr"], c["flux"], c["pos"], c["use_gpu"], c["max_subbox"],
                                           c["subboxsize"], c["loss_fraction"], c["temp"], c["ndens"], xh, c["thin"], c["thick"],
                                           cases.MINLOGTAU, c["dlogtau"], c["R"], c["convergence_fraction"], cases.SIG, cases.BH00,
                                           cases.ALBPOW, cases.COLH0, cases.TEMPH0, cases.ABU_C)
            want = golden[f"{name}__rows{step}"]
            assert len(rows) == len(want), f"step {step}: outer iterations"
            for (flag, rel), (flag_ref, rel_ref) in zip(rows, want):
                assert abs(int(flag) - int(flag_ref)) <= max(2, int(2e-4 * flag_ref))      # cells sitting on the 1e-3 threshold
                assert float(rel) == pytest.approx(rel_ref, rel=2e-2, abs=1e-12)           # the fixture holds the log's three digits
            np.testing.assert_allclose(x, golden[f"{name}__xh{step}"], rtol=1e-8, atol=0)
            ref_phi = golden[f"{name}__phi{step}"]
            w = ref_phi != 0
            assert np.array_equal(phi != 0, w)
            np.testing.assert_allclose(phi[w], ref_phi[w], rtol=1e-7, atol=0)
            x_is_f, phi_is_f = golden[f"{name}__orders{step}"]                 # evolve.py:178,200: the arrays the loop hands back
            assert (x.flags.f_contiguous and not x.flags.c_contiguous) == bool(x_is_f)
            assert (phi.flags.f_contiguous and not phi.flags.c_contiguous) == bool(phi_is_f)
            xh = x
    finally:
        if c["use_gpu"]:
            libasora.device_close()
