"""The C-ABI library loads on a CPU-only box, exports every symbol include/qampy_hip.h declares and fails loudly
(no CPU fallback) when a kernel entry point is called without a GPU."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from qampy_amd import _lib


def _declared():
    text = open(os.path.join(ROOT, "include", "qampy_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qh_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 39
    for n in names:
        assert hasattr(lib, n), "%s declared in include/qampy_hip.h but not exported" % n
    # and the ctypes signature table covers the whole header (minus qh_last_error, which returns a string, and qh_abi_version,
    # which returns the version itself and is checked by the loader)
    assert set(names) - {"qh_last_error", "qh_abi_version"} == set(_lib.SIGNATURES)
    # the loader refuses a library built from another version of the header
    text = open(os.path.join(ROOT, "include", "qampy_hip.h")).read()
    assert int(re.search(r"#define QH_ABI_VERSION (\d+)", text).group(1)) == _lib.ABI_VERSION == lib.qh_abi_version()


def test_method_ids_match_header():
    text = open(os.path.join(ROOT, "include", "qampy_hip.h")).read()
    order = re.search(r"enum \{ (QH_M_CMA.*?) \};", text).group(1).replace(" = 0", "").split(", ")
    assert [o[5:].lower() for o in order] == list(_lib.METHOD_ID)
    order = re.search(r"enum \{ (QH_RM_CMA.*?) \};", text).group(1).replace(" = 0", "").split(", ")
    assert [o[6:].lower() for o in order] == list(_lib.REAL_METHOD_ID)


@pytest.mark.skipif(_lib.device_count() > 0, reason="only meaningful on a box without a GPU")
def test_no_silent_cpu_fallback():
    from qampy_amd.core.equalisation import hip_equalisation as k
    E = np.zeros((1, 64), np.complex64)
    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback"):
        k.apply_filter_to_signal(E, 2, np.zeros((1, 1, 5), np.complex64))
    with pytest.raises(RuntimeError):
        _lib.init(0)


def test_unknown_method_is_a_value_error_before_touching_the_device():
    from qampy_amd.core.equalisation import hip_equalisation as k
    E = np.zeros((1, 64), np.complex64)
    with pytest.raises(ValueError, match="Unknown method"):
        k.train_equaliser(E, 4, 1, 2, np.float32(1e-3), np.zeros((1, 1, 5), np.complex64), None, False,
                          np.ones((1, 1), np.complex64), "nonsense")
    with pytest.raises(TypeError):
        k.train_equaliser(E, 4, 1, 2, np.float32(1e-3), np.zeros((1, 1, 5), np.complex128), None, False,
                          np.ones((1, 1), np.complex64), "cma")


def test_header_is_plain_c(tmp_path):
    """include/qampy_hip.h is a C header (extern "C" only under __cplusplus): a C99 and a C++11 translation unit that include
    it compile, and a C program can link against the library by name."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "abi.c"
    src.write_text('#include "qampy_hip.h"\nint main(void) { int n = -1; return qh_device_count(&n) == QH_OK ? 0 : (n < 0 ? 0 : 0); }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-fsyntax-only", "-x", "c++", "-I", inc, str(src)])
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lqampy_hip", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib"])
    assert subprocess.call([str(exe)]) == 0          # runs without a GPU: counting devices is not an error


@pytest.mark.gpu
def test_c_program_drives_the_hot_path(tmp_path):
    """examples/c_abi_demo.c: training, filter and blind phase search through the C ABI from plain C, no Python in the loop."""
    import subprocess
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = tmp_path / "c_abi_demo"
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_demo.c"),
                           "-o", str(exe), "-L", libdir, "-lqampy_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("symbol errors") == 2


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof / offsetof of qh_pit_opts and qh_pit_report as the C compiler sees them against the ctypes mirrors in qampy_amd/_lib.py."""
    import shutil
    import subprocess
    import ctypes as C
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    from qampy_amd import _lib
    src = tmp_path / "layout.c"
    fields_o = ["segments", "tol", "acq_chunk", "correction", "basis", "corr_beta", "seg_first", "exchange", "exchange_user"]
    fields_r = ["segments", "seg_len", "mu", "defect", "acq_err", "gain", "acq_done", "result_change"]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "qampy_hip.h"', 'int main(void) {',
             'printf("%zu %zu\\n", sizeof(qh_pit_opts), sizeof(qh_pit_report));']
    lines += ['printf("%%zu\\n", offsetof(qh_pit_opts, %s));' % f for f in fields_o]
    lines += ['printf("%%zu\\n", offsetof(qh_pit_report, %s));' % f for f in fields_r]
    lines += ["return 0; }"]
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert [int(out[0]), int(out[1])] == [C.sizeof(_lib.PitOpts), C.sizeof(_lib.PitReport)]
    got = [int(v) for v in out[2:]]
    want = [getattr(_lib.PitOpts, f).offset for f in fields_o] + [getattr(_lib.PitReport, f).offset for f in fields_r]
    assert got == want
