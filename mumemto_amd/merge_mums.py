"""`merge` -- fold the multi-MUMs of independent runs into the multi-MUMs of the union of their inputs.

Host-side twin of the reference's merge driver (mumemto/merge_mums.py; command line `mumemto merge p1.mums p2.mums ...
-o out`, README.md:124-141), over this package's GPU tools:

* every PREFIX.athresh present  -> anchor-based merge: `anchor_merge` (GPU fold, csrc/merge.cpp);
* every PREFIX.thresh present   -> string-based merge:
    1. `extract_mums` writes each partition's MUM strings as a multi-FASTA, one `#`-terminated record per row;
    2. `mumemto_exec` (the GPU hot path) finds the multi-MUMs of those collections ("MUMs of MUMs");
    3. `string_merge` below maps every such match back through the partitions' rows and thresholds.

Step 3 is the reference's per-row Python loop (merge_mums.py:226-287) restated over whole NumPy columns; inputs and
outputs are the same files, byte for byte (tests/golden/string_merge holds the reference's own outputs).

Row order contract (as in the reference): PREFIX.thresh / .thresh_rev are laid out MUM after MUM in the order of the
rows' offsets in the partition's first document (include/mem_finder.hpp:126-130), and step 3 indexes them by position
in the extracted multi-FASTA, so the .mums / .bumbl handed to a string merge must be sorted by first-document offset
(the files this tool writes are; `sort_partition` converts a raw run's output).
"""
import argparse
import os
import shutil
import subprocess
import sys

import numpy as np

from . import mumsio

BIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin")
MIN_SEGMENT = 20          # merge_mums.py:134,141: pieces of a match cut at `#` shorter than this are dropped


def parse_arguments(argv=None):
    ap = argparse.ArgumentParser(description="Merge MUMs files")
    ap.add_argument("--merged_mums", "-m", help="Path to MUMs of MUMs file (only for string merging)")
    ap.add_argument("mum_files", metavar="MUM_FILES", nargs="+", help="Paths to MUMs files to merge")
    ap.add_argument("--output", "-o", help="Path to output merged MUMs file", default="merged")
    ap.add_argument("--verbose", "-v", action="store_true", help="Print verbose output")
    args = ap.parse_args(argv)
    if len(args.mum_files) < 2:
        ap.error("At least two MUMs files are required for merging")
    if not args.output.endswith((".bumbl", ".mums")):
        args.output += ".mums"
    args.output_base = args.output[: args.output.rindex(".")]
    args.paths = []
    for f in args.mum_files:
        if not f.endswith((".mums", ".bumbl")):
            ap.error("Invalid input: %s. Input must explicitly end with .mums or .bumbl." % f)
        args.paths.append(f[: f.rindex(".")])
    if args.merged_mums is not None:
        if not args.merged_mums.endswith(".mums"):
            args.merged_mums += ".mums"
        if not os.path.exists(args.merged_mums):
            print("Error: MUMs of MUMs file %s does not exist. Omit -m to run merge from start." % args.merged_mums,
                  file=sys.stderr)
            sys.exit(1)
    return args


# ---- PREFIX.lengths -------------------------------------------------------------------------------------------------
def _write_length_lines(path, lines):
    """All two-column or all three-column lines pass through; a mix is lifted to the three-column (multi-FASTA) form
    (merge_mums.py:78-91)."""
    widths = {len(ln) for ln in lines}
    if widths <= {2} or widths <= {3}:
        out = lines
    else:
        out = []
        for ln in lines:
            if len(ln) == 3:
                out.append(ln)
            else:
                out.append([ln[0], "*", ln[1]])
                out.append([ln[0], os.path.basename(ln[0]), ln[1]])
    with open(path, "w") as f:
        f.write("\n".join(" ".join(ln) for ln in out))


def _length_lines(prefix):
    with open(prefix + ".lengths") as f:
        return [ln.split() for ln in f.read().splitlines()]


def merge_lengths(args):
    _write_length_lines(args.output_base + ".lengths", [ln for p in args.paths for ln in _length_lines(p)])


def merge_anchor_lengths(args):
    per_file = [_length_lines(p) for p in args.paths]
    anchor = os.path.basename(per_file[0][0][0])
    for lines in per_file:
        if os.path.basename(lines[0][0]) != anchor:
            print("Error: Cannot perform anchor-merge. Anchor sequence is not identical in each partition. Ensure paths "
                  "are identical in the first line of each lengths file.", file=sys.stderr)
            sys.exit(1)
    keep = list(per_file[0])
    for lines in per_file[1:]:
        keep += [ln for ln in lines if os.path.basename(ln[0]) != anchor]
    _write_length_lines(args.output_base + ".lengths", keep)


def record_lengths(lengths_file):
    """Per-document lists of record lengths of a three-column lengths file (mumemto/utils.py:180-192)."""
    docs, cur = [], None
    with open(lengths_file) as f:
        for ln in f:
            t = ln.split()
            if not t:
                continue
            if len(t) < 3:
                raise ValueError("Multi-FASTA lengths not available in " + lengths_file)
            if t[1] == "*":
                if cur is not None:
                    docs.append(cur)
                cur = []
            else:
                cur.append(int(t[2]))
    docs.append(cur if cur is not None else [])
    return [np.array(d, np.int64) for d in docs]


# ---- the string merge proper ----------------------------------------------------------------------------------------
def _sorted_by_first_offset(rows):
    lengths, starts, strands = rows
    if len(lengths) and np.any(np.diff(starts[:, 0]) < 0):
        order = np.argsort(starts[:, 0], kind="stable")
        return lengths[order], starts[order], strands[order]
    return rows


def split_at_terminators(mom, terminators):
    """Cut every MUM-of-MUMs row at the `#` characters it spans in the first collection (merge_mums.py:121-146).
    mom = (lengths, starts [n, S], strands [n, S]); terminators = sorted positions of `#` in collection 0.
    Pieces keep row order, then left-to-right order; a piece shorter than MIN_SEGMENT is dropped."""
    lengths, starts, strands = mom
    lengths = lengths.astype(np.int64)
    s0 = starts[:, 0]
    a = np.searchsorted(terminators, s0, "left")
    b = np.searchsorted(terminators, s0 + lengths, "left")
    pieces = b - a + 1
    row = np.repeat(np.arange(len(lengths)), pieces)
    k = np.arange(int(pieces.sum())) - np.repeat(np.cumsum(pieces) - pieces, pieces)      # piece number within its row
    term = np.concatenate((terminators, [0]))                                              # padded for the tail piece
    left = np.where(k == 0, 0, term[np.maximum(a[row] + k - 1, 0)] - s0[row] + 1)
    end = np.where(k == pieces[row] - 1, lengths[row], term[np.minimum(a[row] + k, len(terminators))] - s0[row])
    new_len = end - left
    keep = (new_len >= MIN_SEGMENT) | (pieces[row] == 1)
    row, left, end, new_len = row[keep], left[keep], end[keep], new_len[keep]
    fwd = strands[row]
    new_starts = np.where(fwd, starts[row] + left[:, None], starts[row] + (lengths[row] - end)[:, None])
    return new_len, new_starts, fwd


def string_merge(parts, mom, mom_records):
    """parts: per partition (lengths, starts, strands, thresh, thresh_rev), rows in first-offset order.
    mom: rows of the multi-MUMs of the extracted collections; mom_records: per collection the record lengths
    (MUM length + 1 for the `#`).  Returns (lengths u32, starts i64, strands bool, thresh u16, thresh_rev u16) of the
    merged run, rows sorted by offset in the first document (merge_mums.py:211-307)."""
    S = len(parts)
    if not (len(mom_records) == S and mom[1].shape[1] == S):
        raise ValueError("input # of MUM files does not match merged MUM input file")
    rec_end = [np.cumsum(r) for r in mom_records]                       # one past each record (after its `#`)
    rec_begin = [np.concatenate(([0], e)) for e in rec_end]
    seg_len, seg_start, seg_fwd = split_at_terminators(mom, rec_end[0] - 1)

    ok = np.ones(len(seg_len), bool)
    rec, left_off, right_off = [], [], []
    for i in range(S):
        r = np.searchsorted(rec_end[i], seg_start[:, i], "right")       # the MUM of partition i the piece lies in
        rc = np.minimum(r, len(rec_end[i]) - 1)
        rec.append(rc)
        left_off.append(seg_start[:, i] - rec_begin[i][rc])
        right_off.append(rec_begin[i][rc + 1] - seg_start[:, i] - seg_len - 1)
        th = np.take(parts[i][3], seg_start[:, i], mode="clip").astype(np.int64)
        ok &= (r < len(rec_end[i])) & (th != 0) & (seg_len > th)        # still unique inside partition i
    sel = np.nonzero(ok)[0]

    cols_start, cols_strand = [], []
    for i in range(S):
        _, p_starts, p_strands = parts[i][:3]
        r = rec[i][sel]
        fwd = p_strands[r]
        cols_start.append(np.where(fwd, p_starts[r] + left_off[i][sel, None], p_starts[r] + right_off[i][sel, None]))
        cols_strand.append(np.where(seg_fwd[sel, i, None], fwd, ~fwd))
    n_cols = sum(p[1].shape[1] for p in parts)
    starts = np.concatenate(cols_start, axis=1) if len(sel) else np.zeros((0, n_cols), np.int64)
    strands = np.concatenate(cols_strand, axis=1) if len(sel) else np.zeros((0, n_cols), bool)
    order = np.argsort(starts[:, 0], kind="stable")
    sel, starts, strands = sel[order], starts[order], strands[order]
    lengths = seg_len[sel]

    # thresholds of the merged rows: position by position the maximum over the partitions where every partition
    # still has one (> 0), else 0; forward and reverse tables swap for a partition whose MUM matched reversed
    total = int(lengths.sum())
    which = np.repeat(np.arange(len(sel)), lengths)
    within = np.arange(total) - np.repeat(np.cumsum(lengths) - lengths, lengths)
    fwd_max = np.zeros(total, np.uint16)
    rev_max = np.zeros(total, np.uint16)
    fwd_all = np.ones(total, bool)
    rev_all = np.ones(total, bool)
    for i in range(S):
        th, th_rev = parts[i][3], parts[i][4]
        a = np.take(th, seg_start[sel, i][which] + within, mode="clip")
        b = np.take(th_rev, (rec_begin[i][rec[i][sel]] + right_off[i][sel])[which] + within, mode="clip")
        same = seg_fwd[sel, i][which]
        f, r = np.where(same, a, b), np.where(same, b, a)
        fwd_max, rev_max = np.maximum(fwd_max, f), np.maximum(rev_max, r)
        fwd_all &= f > 0
        rev_all &= r > 0
    out_pos = np.arange(total) + which                                  # one 0 after every row
    thresh = np.zeros(total + len(sel), np.uint16)
    thresh_rev = np.zeros(total + len(sel), np.uint16)
    thresh[out_pos] = np.where(fwd_all, fwd_max, 0)
    thresh_rev[out_pos] = np.where(rev_all, rev_max, 0)
    return lengths.astype(np.uint32), starts, strands, thresh, thresh_rev


# ---- drivers --------------------------------------------------------------------------------------------------------
def _tool(name):
    path = os.path.join(BIN, name)
    if not os.path.exists(path):
        path = shutil.which(name)
    if not path:
        print("Error: %s not built. Run `python -m mumemto_amd.build` first" % name, file=sys.stderr)
        sys.exit(1)
    return path


def run_merger(args):
    """extract the MUM strings of every partition and find their multi-MUMs on the GPU (merge_mums.py:138-168)"""
    for f in args.mum_files:
        if subprocess.run([_tool("extract_mums"), "-m", f]).returncode != 0:
            print("Error: Partial MUMs detected. Aborting merge. Cleaning up...", file=sys.stderr)
            for p in args.paths:
                if os.path.exists(p + "_mums.fa"):
                    os.remove(p + "_mums.fa")
            sys.exit(1)
    cmd = [_tool("mumemto_exec")] + [p + "_mums.fa" for p in args.paths] + ["-o", args.output_base + "_temp_merged"]
    if args.verbose:
        print("Running command: " + " ".join(cmd), file=sys.stderr)
    if subprocess.run(cmd).returncode != 0:
        print("Error: mumemto_exec failed on the extracted MUM sequences", file=sys.stderr)
        sys.exit(1)
    args.merged_mums = args.output_base + "_temp_merged.mums"


def run_anchor_merger(args):
    cmd = [_tool("anchor_merge")] + args.mum_files + ["-o", args.output] + (["-v"] if args.verbose else [])
    if args.verbose:
        print("Running command: " + " ".join(cmd), file=sys.stderr)
    return subprocess.run(cmd).returncode


def sort_partition(prefix, out_prefix=None):
    """Rewrite a raw run's PREFIX.mums in first-document offset order (the order PREFIX.thresh is already in)."""
    lengths, starts, strands = _sorted_by_first_offset(mumsio.read_mums(prefix + ".mums"))
    mumsio.write_mums((out_prefix or prefix) + ".mums", lengths, starts, strands)


def main(args):
    if all(os.path.exists(p + ".athresh") for p in args.paths):
        if args.merged_mums is not None:
            print("Error: -m is only for string merging, but anchor-based merging detected. Ignoring -m.", file=sys.stderr)
        merge_anchor_lengths(args)
        sys.exit(run_anchor_merger(args))
    if not all(os.path.exists(p + ".thresh") for p in args.paths):
        print("Error: *.thresh or *.athresh files required for all inputs for merging.", file=sys.stderr)
        sys.exit(1)
    merge_lengths(args)
    cleanup = args.merged_mums is None
    if cleanup:
        run_merger(args)

    parts = []
    for f, p in zip(args.mum_files, args.paths):
        rows = _sorted_by_first_offset(mumsio.read_rows(f))
        parts.append(rows + (np.fromfile(p + ".thresh", np.uint16), np.fromfile(p + ".thresh_rev", np.uint16)))
    mom_prefix = os.path.splitext(args.merged_mums)[0]
    lengths, starts, strands, thresh, thresh_rev = string_merge(parts, mumsio.read_mums(args.merged_mums),
                                                                record_lengths(mom_prefix + ".lengths"))
    if args.output.endswith(".bumbl"):
        mumsio.write_bumbl(args.output, lengths, starts, strands)
    else:
        mumsio.write_mums(args.output, lengths, starts, strands)
    thresh.tofile(args.output_base + ".thresh")
    thresh_rev.tofile(args.output_base + ".thresh_rev")
    if cleanup:
        for p in args.paths:
            os.remove(p + "_mums.fa")
        os.remove(args.merged_mums)
        os.remove(args.output_base + "_temp_merged.lengths")
    return len(lengths)


if __name__ == "__main__":
    main(parse_arguments())
