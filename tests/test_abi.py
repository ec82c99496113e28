"""CPU-only checks of the drop-in boundary: the shared library loads, exports every symbol that
include/simpledet_ops.h declares, and validates arguments without touching a GPU."""
import ctypes
import os

import pytest

from simpledet_amd import _lib


def test_header_parses_and_library_exports_every_symbol():
    protos = _lib.parse_header()
    assert len(protos) >= 20
    l = _lib.lib()
    for name in protos:
        assert hasattr(l.cdll, name), name
    # layout / size contracts are versioned too: library and header must agree exactly
    assert l.cdll.sd_abi_version() == _lib.header_abi_version() >= 3


def test_argument_validation_needs_no_gpu():
    l = _lib.lib()
    with pytest.raises(_lib.SimpleDetOpsError, match="pooled_size"):
        l.call("sd_roi_align_v2_fwd", None, None, None, None, None, 1, 1, 4, 4, 1, 0, 7, 0.5, None)
    with pytest.raises(_lib.SimpleDetOpsError, match="kWriteInplace"):
        l.call("sd_roi_align_v2_bwd", None, None, None, None, None, None, 2, 1, 1, 1, 4, 4, 1, 7, 7,
               0.5, None)
    with pytest.raises(_lib.SimpleDetOpsError, match="unknown tuning key"):
        l.call("sd_set_tuning", b"no_such_knob", 1)
    with pytest.raises(_lib.SimpleDetOpsError, match="method"):
        l.call("sd_soft_nms_batched", None, None, 1, 10, 0.5, 0.3, 0.001, 7, None, None, None, None)
    # empty problems are accepted without touching the device
    assert l.call("sd_nms", None, 0, 0, -1, 10, 0.5, 0, 0, None, None, None, None, 0, None) == 0
    assert l.call("sd_gen_anchor", None, 0, 0, 16, (ctypes.c_double * 1)(8.0), 1,
                  (ctypes.c_double * 1)(1.0), 1, None) == 0


def test_glibc_srand_host_matches_oracle(oracle):
    l = _lib.lib()
    for seed in (1, 0, 12345):
        st = (ctypes.c_int32 * 33)()
        l.call("sd_glibc_srand_host", ctypes.c_uint32(seed), st)
        g = oracle.GlibcRand(seed)
        assert list(st) == list(g.state_words())


def test_no_product_code_touches_the_oracle():
    """The product path must never import/link the oracle (voids parity claims otherwise)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dp, _, files in os.walk(os.path.join(root, "simpledet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dp, f)).read()
                for bad in ("pyoracle", "liboracle", "import oracle", "from oracle", "-loracle",
                            "oracle.h", "orc_"):
                    assert bad not in text, (f, bad)
