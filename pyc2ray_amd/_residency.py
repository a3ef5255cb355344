"""Who holds the device copies of the grids (C2Ray.device_resident).

The library's device state is process-global (as the reference's, src/asora/memory.cu:20-29).  A C2Ray object that keeps
its grids on the device between time steps trusts that state until its next step -- so every other way into the library
that overwrites the grids (the module-level evolve3D / evolve3D_MPI / do_raytracing, another C2Ray object's step,
device_init / device_close) first asks the resident objects to TAKE THEIR DATA HOME: results that so far exist only on the
device are downloaded, and every input grid is marked for upload at the object's next step.  Callers of the bare C-ABI
(``load_asora()`` methods) bypass this, as they bypass the class.
"""
import weakref

_resident = weakref.WeakSet()


def register(sim):
    _resident.add(sim)


def reclaim(except_for=None):
    """Called before anything but `except_for` overwrites the device grids."""
    for sim in list(_resident):
        if sim is not except_for:
            sim._leave_device()
