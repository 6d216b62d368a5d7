"""CPU: the C-ABI library loads and exports every symbol include/xna_basecaller.h declares (no compute calls)."""
import ctypes
import os
import re

from conftest import ROOT
from xna_basecaller_amd import _lib


def _declared():
    text = open(os.path.join(ROOT, "include", "xna_basecaller.h")).read()
    return sorted(set(re.findall(r"XB_API\s+[\w\s\*]+?\b(xb_\w+)\s*\(", text)))


def test_header_symbols_exported():
    names = _declared()
    assert len(names) >= 18 and "xb_basecall_chunks_dev" in names
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.EXPORTS) == names


def test_version_and_config_struct():
    lib = _lib.load()
    assert b"gfx950" in lib.xb_version()
    assert ctypes.sizeof(_lib.XbConfig) == 44
    # argument validation happens before any device work
    h = ctypes.c_void_p()
    cfg = _lib.XbConfig(7, 3, 768, 19, 5, 5.0, 2.0, 4000, 4, 0, 0)
    assert lib.xb_ctx_create(ctypes.byref(h), 0, ctypes.byref(cfg)) == -1
    assert b"n_base" in lib.xb_last_error(None)
    cfg = _lib.XbConfig(6, 3, 100, 19, 5, 5.0, 2.0, 4000, 4, 0, 0)
    assert lib.xb_ctx_create(ctypes.byref(h), 0, ctypes.byref(cfg)) == -1
    assert b"features" in lib.xb_last_error(None)


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU, never compute on the CPU."""
    import pytest
    from conftest import make_config
    from xna_basecaller_amd.crf.model import Model
    import numpy as np
    m = Model(make_config(32))
    with pytest.raises(RuntimeError):
        m.to("cpu")
    if _lib.device_count() == 0:
        with pytest.raises(Exception):
            m(np.zeros((1, 1, 400), np.float32))


def test_product_does_not_import_oracle():
    import subprocess, sys
    code = "import sys; import xna_basecaller_amd, xna_basecaller_amd.cli.basecaller, xna_basecaller_amd.dist; " \
           "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'"
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "xna_basecaller_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "libxna_oracle" not in src, f
                assert not [l for l in src.splitlines() if l.lstrip().startswith("#include") and "oracle" in l], f


def test_gemm4p_main_loop_isa_audit():
    """The product GEMM's weight loads are inline asm with hand-counted waits: tools/check_gemm_isa.py compiles the kernels for
    gfx950 (no GPU needed) and checks every instantiation's main loop for what would break that -- register copies between load
    and wait, compiler-inserted vmcnt waits or branches, spills."""
    import subprocess
    import sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_gemm_isa.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count(": ok ") == 9, out.stdout
