"""ctypes declarations for the MoE-adapter section of the ABI (filled in as the ABI grows)."""


def declare(L):
    return
