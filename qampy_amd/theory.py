"""
Constellation tables and equaliser constants that feed the hot path.

Only the ~30 lines of QAMpy's ``theory`` / ``core.equalisation`` modules whose *values* are inputs of the
equaliser and BPS kernels are re-stated here.  All tables are pinned bit-for-bit by
``tests/golden/constants.npz`` which was captured from the reference.

Reference behaviour restated (file:line in /root/reference):
  * ``cal_symbols_qam``            qampy/theory.py:111-118, 150-177
  * ``cal_scaling_factor_qam``     qampy/theory.py:139-148
  * ``gray_code_qam``              qampy/theory.py:180-192, qampy/core/utils.py:195-200
  * Gray-label ordered alphabet    qampy/signals.py:831-845 (``coded_symbols``)
  * CMA / MCMA constants           qampy/core/equalisation/equalisation.py:271-281
  * RDE / MRDE codes + partitions  qampy/core/equalisation/equalisation.py:311-359
"""
import numpy as np


def _is_cross(M):
    nbits = int(round(np.log2(M)))
    if 2 ** nbits != M:
        raise ValueError("M must be a power of two, got %r" % (M,))
    return bool(nbits % 2)


def _square_points(M):
    side = int(round(np.sqrt(M)))
    lev = np.arange(-(side - 1), side, 2, dtype=np.float64)
    # raster order of the reference: real part is the slow axis, imaginary part the fast axis
    re = np.repeat(lev, side)
    im = np.tile(lev, side)
    return re + 1j * im


def _cross_points(M):
    nbits = int(round(np.log2(M)))
    n = (nbits - 1) // 2
    if n < 1:
        raise ValueError("cross QAM needs M >= 8")
    s = 2.0 ** (n - 1)
    wide = 2 ** (n + 1)
    tall = 2 ** n
    re = np.repeat(np.arange(-(wide - 1), wide, 2, dtype=np.float64), tall)
    im = np.tile(np.arange(-(tall - 1), tall, 2, dtype=np.float64), wide)
    are, aim = np.abs(re), np.abs(im)
    outer = are > 3 * s
    corner = outer & (aim > s)
    edge = outer & (aim <= s)
    new_re = np.where(corner, np.sign(re) * (are - 2 * s), np.where(edge, np.sign(re) * (4 * s - are), re))
    new_im = np.where(corner, np.sign(im) * (4 * s - aim), np.where(edge, np.sign(im) * (aim + 2 * s), im))
    return new_re + 1j * new_im


def cal_symbols_qam(M):
    """Un-normalised M-QAM constellation in the reference's raster order (qampy/theory.py:111-118)."""
    return _cross_points(M) if _is_cross(M) else _square_points(M)


def cal_scaling_factor_qam(M):
    """Mean power of the un-normalised constellation (qampy/theory.py:139-148)."""
    if _is_cross(M):
        return (np.abs(cal_symbols_qam(M)) ** 2).mean()
    return 2 / 3 * (M - 1)


def normalised_symbols_qam(M):
    """``cal_symbols_qam(M)/sqrt(scale)`` exactly as equalisation.py:117-125 evaluates it (complex / real)."""
    syms = cal_symbols_qam(M)
    syms /= np.sqrt(cal_scaling_factor_qam(M))
    return syms


def gray_code_qam(M):
    """Gray labels of the raster-ordered constellation (qampy/theory.py:180-192)."""
    nbits = int(round(np.log2(M)))
    n_im = nbits // 2
    n_re = nbits - n_im
    a = np.repeat(np.arange(2 ** n_re), 2 ** n_im)
    b = np.tile(np.arange(2 ** n_im), 2 ** n_re)
    ga = a ^ (a >> 1)
    gb = b ^ (b >> 1)
    return (ga << n_im) | gb


def coded_symbols_qam(M, dtype=np.complex128):
    """
    Unit-power alphabet in Gray-label order, i.e. what ``SignalQAMGrayCoded.coded_symbols`` holds
    (qampy/signals.py:831-845, :662).  Entry ``g`` is the point whose Gray label is ``g``.
    """
    syms = cal_symbols_qam(M).astype(dtype)
    syms /= np.sqrt(cal_scaling_factor_qam(M))
    code = gray_code_qam(M)
    inv = np.zeros_like(code)
    inv[code] = np.arange(code.size)
    return syms[inv]


# --------------------------------------------------------------------------- equaliser constants
def cal_Rconstant(M):
    """CMA radius <|s|^4>/<|s|^2> (equalisation.py:271-275)."""
    s = normalised_symbols_qam(M)
    return np.mean(abs(s) ** 4) / np.mean(abs(s) ** 2)


def cal_Rconstant_complex(M):
    """MCMA per-quadrature radius (equalisation.py:277-281)."""
    s = normalised_symbols_qam(M)
    return np.mean(s.real ** 4) / np.mean(s.real ** 2) + 1.j * np.mean(s.imag ** 4) / np.mean(s.imag ** 2)


def generate_partition_codes_complex(M):
    """MRDE codes then partitions, real axis in .real and imaginary axis in .imag (equalisation.py:311-336)."""
    s = normalised_symbols_qam(M)
    # NB: the reference forms |x|^4/|x|^2 (not |x|^2) before np.unique, which leaves float-duplicates for
    # some M; the tables are pinned by golden constants, so the same expression is evaluated here.
    cr = np.unique(abs(s.real) ** 4 / abs(s.real) ** 2)
    ci = np.unique(abs(s.imag) ** 4 / abs(s.imag) ** 2)
    codes = cr + 1.j * ci
    pr = cr[:-1] + np.diff(cr) / 2
    pi = ci[:-1] + np.diff(ci) / 2
    return np.hstack([codes, pr + 1.j * pi])


def generate_partition_codes_radius(M):
    """RDE codes then partitions on |s|^2 (equalisation.py:338-359)."""
    s = normalised_symbols_qam(M)
    codes = np.unique(abs(s) ** 4 / abs(s) ** 2)
    parts = codes[:-1] + np.diff(codes) / 2
    return np.hstack([codes, parts])
