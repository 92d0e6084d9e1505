"""CPU tier: the ISA lint `make` runs on the product library (tools/isa_lint.py) -- here once more on whatever library is in
the tree, and its detector on a synthetic listing (a lint that cannot fail proves nothing)."""
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_lint_detects_what_it_is_for():
    import isa_lint
    ks = isa_lint.kernels("0000000000001000 <_ZN3gqp9ky_factorILi8ELi15ELi0EEEv6GqpDev7GqpOptsi>:\n\ts_load_dwordx2 s[0:1], s[4:5], 0x0 // 000\n"
                          "\tglobal_load_lds_dwordx4 v1, s[2:3] offset:1024 // 004\n\ts_waitcnt vmcnt(3) // 008\n\ts_waitcnt vmcnt(0) // 00c\n")
    (sym, ins), = ks.items()
    assert sym.startswith("_ZN3gqp9ky_factor") and len(ins) == 4
    assert [t for t in ins if t.startswith("global_load_lds")] and "vmcnt(3)" in ins[2]


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "acados_amd", "csrc", "libacados_amd_qp.so")), reason="product library not built")
def test_product_library_passes_the_lint():
    import isa_lint
    bad, report = isa_lint.lint(os.path.join(ROOT, "acados_amd", "csrc", "libacados_amd_qp.so"))
    assert not bad, bad
    # the two-rows family is what uses LDS-DMA: factor + the two forward sweeps of every compiled shape
    assert len(report) >= 11 and all("ky_" in r or "kt_" in r for r in report), report   # (kt_: the factor sweep on MFMA tiles)
    assert all("scratch 0 B" in r for r in report), report
    strict_bad, _ = isa_lint.lint(os.path.join(ROOT, "acados_amd", "csrc", "libacados_amd_qp.so"), strict=True)
    assert strict_bad and all(b.startswith("rule 1") for b in strict_bad)      # the listing exists: see profiles/r04_vmcnt_probe.txt for why it is not a fault
