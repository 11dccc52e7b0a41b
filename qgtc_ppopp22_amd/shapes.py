"""Padding / shape algebra of the packed layouts (reference utility.h:33-45, QGTC_device.cu:83,97,115)."""


def S8(x: int) -> int:
    return (x + 7) >> 3


def S128(x: int) -> int:
    return (x + 127) >> 7


def P8(x: int) -> int:
    return S8(x) << 3


def P128(x: int) -> int:
    return S128(x) << 7


def rows_shape(H: int, W: int, nbits: int):
    """Tensor shape of the rows layout of an HxW matrix (QGTC_device.cu:115,223)."""
    return (nbits * P8(H), S128(W) * 4)


def cols_shape(H: int, W: int, nbits: int, output_layer: bool = False):
    """Tensor shape of the cols layout of an HxW matrix (QGTC_device.cu:83,97,456)."""
    return (nbits * S128(H) * 4, P8(W) if output_layer else P128(W))
