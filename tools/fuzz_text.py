#!/usr/bin/env python
"""The GPU text reader against pandas on random files -- well-formed ones of every layout and MUTATED ones (a byte replaced,
inserted or dropped; a field emptied, quoted, replaced by NA / text / a long integer; a line dropped, doubled, cut or blanked):
whatever textio.read_frame RETURNS must be the frame pd.read_csv returns (values bit for bit, labels, dtypes); refusing is
always allowed, accepting something pandas reads differently never is.  Not part of the suite (minutes); a failure prints the
file and exits non-zero.      SEED=1 CASES=2000 [BIG=1] python tools/fuzz_text.py"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import textcases  # noqa: E402
from prmf_amd import textio  # noqa: E402
from prmf_amd.runner import check_header, check_row_names  # noqa: E402

rs = np.random.RandomState(int(os.environ.get("SEED", 1)))
ncases = int(os.environ.get("CASES", 1000))
pool = textcases.spellings(20000, seed=int(os.environ.get("SEED", 1)) + 100)
if os.environ.get("BIG"):       # 20000 cells per file: keep the spellings the reader takes (integers of > 15 digits are refused), or none would pass
    pool = [x for x in pool if not (x.lstrip("+-").isdigit() and len(x.lstrip("+-")) > 15)]
ints = [str(int(v)) for v in rs.randint(-1000, 100000, size=2000)]
d = tempfile.mkdtemp(prefix="prmf_fuzz_text_")
path = os.path.join(d, "f.txt")
accepted = refused = mutated_accepted = 0


def build():
    big = int(os.environ.get("BIG", 0))              # BIG=1: files of many 4 KiB tiles (tile and 16-byte boundaries everywhere)
    m, n = (int(rs.randint(1, 80)), int(rs.randint(1, 260))) if big else (int(rs.randint(1, 12)), int(rs.randint(1, 15)))
    delim = str(rs.choice([",", "\t", ";", "|", " "]))
    header, names = rs.rand() < 0.7, rs.rand() < 0.6
    kind = rs.rand()
    src = ints if kind < 0.15 else pool
    cells = [[src[rs.randint(len(src))] for _ in range(n)] for _ in range(m)]
    if 0.15 <= kind < 0.3:
        for row in cells:                       # one integer-only column among floats
            row[0] = ints[rs.randint(len(ints))]
    quote = rs.rand() < 0.3
    q = (lambda s: '"%s"' % s) if quote else (lambda s: s)
    lines = []
    if header:
        lines.append(delim.join(([q("id")] if names else []) + [q("g%d" % j) for j in range(n)]))
    for i, row in enumerate(cells):
        lines.append(delim.join(([q("r%d" % i)] if names else []) + row))
    eol = "\r\n" if rs.rand() < 0.2 else "\n"
    return eol.join(lines) + (eol if rs.rand() < 0.85 else ""), delim


def mutate(text, delim):
    t = rs.randint(18)
    b = bytearray(text.encode())
    if not b:
        return text
    pos = int(rs.randint(len(b)))
    if t == 0:
        b[pos] = rs.choice(list(b'0123456789.,;-+eE"x \t\n\r' + delim.encode()))
    elif t == 1:
        b.insert(pos, int(rs.choice(list(b'0123456789.,;-+eE"x \n' + delim.encode()))))
    elif t == 2:
        del b[pos]
    else:
        lines = text.split("\n")
        li = int(rs.randint(len(lines)))
        fields = lines[li].split(delim)
        fi = int(rs.randint(len(fields)))
        if t == 3:
            fields[fi] = ""
        elif t == 4:
            fields[fi] = '"%s"' % fields[fi]
        elif t == 5:
            fields[fi] = str(rs.choice(["NA", "nan", "inf", "-inf", "NULL", "abc", "1e400", "0x1A", "1,5", "12345678901234567890", "1234567890123456", "7" * 300]))
        elif t == 6:
            fields.append(pool[rs.randint(len(pool))])
        elif t == 7 and len(fields) > 1:
            del fields[fi]
        lines[li] = delim.join(fields)
        if t == 8:
            del lines[li]
        elif t == 9:
            lines.insert(li, lines[li])
        elif t == 10:
            lines.insert(li, "")
        elif t == 11:
            lines[li] = lines[li][:max(0, len(lines[li]) // 2)]
        elif t == 12:                              # white space around a field
            fields[fi] = str(rs.choice([" ", "  ", "\t"])) + fields[fi] + str(rs.choice(["", " ", "\t "]))
            lines[li] = delim.join(fields)
        elif t == 13 and len(fields) > 1:          # two equal labels / values in a line
            fields[fi] = fields[(fi + 1) % len(fields)]
            lines[li] = delim.join(fields)
        elif t == 14:                              # a first field that pandas types itself: number, NA, empty, quoted number
            fields[0] = str(rs.choice(["7", "3.5", "NA", "", '"12"', "nan", "-", "1e3", "#N/A", "None", "<NA>"]))
            lines[li] = delim.join(fields)
        elif t == 15:
            lines[li] = lines[li].replace(delim, delim + delim, 1)
        elif t == 16:
            lines[li] = lines[li] + str(rs.choice([delim, " ", "\r", "#c", ";"]))
        elif t == 17:
            fields[fi] = str(rs.choice(["+", "-", "00", "-0", "+0.", "1.", ".5", "5e", "5e+", "e5", "1e+05", "1E5", "1d5", "1_000", "٣", "１"])) + \
                ("" if rs.rand() < 0.5 else fields[fi])
            lines[li] = delim.join(fields)
        return "\n".join(lines)
    return b.decode("utf-8", "replace")


def pandas_frame(delim, has_header, has_names):
    return pd.read_csv(path, sep=delim, header="infer" if has_header else None, index_col=0 if has_names else None)


for case in range(ncases):
    text, delim = build()
    is_mutant = rs.rand() < 0.6
    if is_mutant:
        for _ in range(int(rs.randint(1, 3))):
            text = mutate(text, delim)
    with open(path, "w", newline="", encoding="utf-8") as fh:
        fh.write(text)
    try:
        has_header = check_header(path, delim)
        has_names = check_row_names(path, delim, has_header)
    except Exception:
        continue                                   # the reference's own probes choke on it: it never reaches the reader
    why = []
    got = textio.read_frame(path, delim, has_header, has_names, why=why)
    if got is None:
        refused += 1
        continue
    accepted += 1
    mutated_accepted += is_mutant
    try:
        want = pandas_frame(delim, has_header, has_names)
        pd.testing.assert_frame_equal(got, want, check_exact=True, check_dtype=True, check_names=True)
        a, b = got.to_numpy(), want.to_numpy()
        assert a.dtype == b.dtype
        if a.dtype == np.float64:
            assert (a.view(np.int64) == b.view(np.int64)).all()
    except BaseException as e:
        print("CASE %d: the reader ACCEPTED a file pandas reads differently (or not at all): %r" % (case, e))
        print(repr(text)[:3000], "delimiter %r header %s row names %s" % (delim, has_header, has_names))
        sys.exit(1)
print("%d files: %d accepted and equal to pandas' frame (%d of them mutated), %d refused" % (ncases, accepted, mutated_accepted, refused))
import shutil  # noqa: E402
shutil.rmtree(d, ignore_errors=True)
