"""Lowering of the compiled model into the tables the HIP kernels consume (placeholder
until the kernel-side layout is fixed; see lower())."""


def lower(cm):
    return cm
