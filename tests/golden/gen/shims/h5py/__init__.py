"""import-only stand-in so NuRadioMC.simulation.simulation can be imported (no file I/O is used)."""


class File:
    def __init__(self, *a, **k):
        raise RuntimeError("h5py stand-in: file I/O is not available")


class Group:
    pass


class Dataset:
    pass
