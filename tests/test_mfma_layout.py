"""CPU tier: the operand / result layout of v_mfma_f64_4x4x4_4b_f64 that mfma4.hpp documents and its host emulation implements
(block = (lane >> 2) & 3; A[i][k] in lane i + 4 b + 16 k, B[k][j] in lane j + 4 b + 16 k, D[i][j] in lane j + 4 b + 16 i) against
what the instruction did on the device: the 256 (A-lane, B-lane, D-lane) triples tools/mfma_f64_probe/probe3.hip recorded from
all 64 x 64 unit-vector pairs (profiles/r04_mfma4x4x4_layout.txt)."""
import os
import re

from conftest import ROOT


def test_documented_layout_matches_the_probe():
    triples = set()
    for ln in open(os.path.join(ROOT, "profiles", "r04_mfma4x4x4_layout.txt")):
        m = re.match(r"A lane\s+(\d+):(.*)", ln)
        if not m:
            continue
        la = int(m.group(1))
        for lb, lds in re.findall(r"B\s+(\d+) -> D((?:\s+\d+)+)", m.group(2)):
            for ld in lds.split():
                triples.add((la, int(lb), int(ld)))
    assert len(triples) == 256
    want = {(i + 4 * b + 16 * k, j + 4 * b + 16 * k, j + 4 * b + 16 * i)
            for b in range(4) for i in range(4) for j in range(4) for k in range(4)}
    assert triples == want


def test_host_emulation_formula_is_the_documented_one():
    """the emulation in mfma4.hpp: lane l = (blk, j = l & 3, i = l >> 4) sums a[i + 4 blk + 16 k] * b[j + 4 blk + 16 k] over k"""
    src = open(os.path.join(ROOT, "acados_amd", "csrc", "mfma4.hpp")).read()
    assert "m4_a[w0 + i + 4 * blk + 16 * k] * m4_b[w0 + j + 4 * blk + 16 * k]" in src
    assert "blk = (l >> 2) & 3, j = l & 3, i = l >> 4" in src
