"""Pins the oracle's anchor-merge restatement against the REAL reference binary
(oracle/_ref/anchor_merge = src/merge_candidates.cpp compiled unmodified);
fixtures by tests/golden/make_golden.py."""
import glob
import os

import numpy as np
import pytest

import pyoracle as O
from mumsfile import format_mums, parse_mums

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "anchor_merge")
CASES = sorted(os.listdir(G))


def load_parts(case):
    parts = []
    for p in sorted(glob.glob(os.path.join(G, case, "p*.mums"))):
        L, off, st = parse_mums(open(p, "rb").read())
        nb = np.fromfile(p[:-5] + ".athresh", np.uint16)
        parts.append((L, off, st, nb))
    return parts


@pytest.mark.parametrize("case", CASES)
def test_fold_matches_reference_binary(case):
    L, off, st, nb = O.anchor_merge(load_parts(case))
    assert format_mums(L, off, st) == open(os.path.join(G, case, "merged.mums"), "rb").read()
    assert nb.tobytes() == open(os.path.join(G, case, "merged.athresh"), "rb").read()


@pytest.mark.parametrize("case", CASES)
def test_merged_resorted_equals_direct_run(case):
    # re-sorting merged rows by match string (bytes of the anchor
    # at off_0) reproduces the direct run byte for byte; merged .athresh too.
    anchor = np.load(os.path.join(G, case, "anchor.npy")).tobytes().upper()
    L, off, st = parse_mums(open(os.path.join(G, case, "merged.mums"), "rb").read())
    order = sorted(range(len(L)), key=lambda i: anchor[off[i, 0]: off[i, 0] + L[i]])
    resorted = format_mums(L[order], off[order], st[order])
    assert resorted == open(os.path.join(G, case, "direct.mums"), "rb").read()
    assert open(os.path.join(G, case, "merged.athresh"), "rb").read() == \
        open(os.path.join(G, case, "direct.athresh"), "rb").read()
