"""ctypes declarations for include/folve_host.h (SoundProcessor / ProcessorPool mirror)."""
HOST_SYMBOLS = []


def declare(L):
    for name, res, args in HOST_SYMBOLS:
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
