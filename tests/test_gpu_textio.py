"""--data as delimited text through the GPU (prmf_text_* / prmf_amd.textio) against the reader it replaces:
pd.read_csv as the reference calls it (script/prmf_runner.py:942) -- the same DataFrame, values bit for bit."""
import csv
import ctypes as C
import os

import numpy as np
import pandas as pd
import pytest

import synth
import textcases
from prmf_amd import _capi, textio

pytestmark = pytest.mark.gpu


def write_matrix(path, cells, delimiter=",", header=None, row_names=None, eol="\n", final_newline=True, quote_names=False):
    q = (lambda s: '"%s"' % s) if quote_names else (lambda s: s)
    lines = []
    if header is not None:
        lines.append(delimiter.join(([q(header[0])] if row_names is not None else []) + [q(h) for h in header[1:]]))
    for i, row in enumerate(cells):
        lines.append(delimiter.join(([q(row_names[i])] if row_names is not None else []) + list(row)))
    text = eol.join(lines) + (eol if final_newline else "")
    with open(path, "w", newline="") as fh:
        fh.write(text)


def pandas_frame(path, delimiter, has_header, has_row_names):
    return pd.read_csv(path, sep=delimiter, header="infer" if has_header else None, index_col=0 if has_row_names else None)


def same_frame(got, want):
    assert got is not None
    pd.testing.assert_frame_equal(got, want, check_exact=True, check_dtype=True, check_names=True)
    a, b = got.to_numpy(), want.to_numpy()
    assert a.dtype == b.dtype
    if a.dtype == np.float64:
        assert (a.view(np.int64) == b.view(np.int64)).all()          # signed zeros too


def cells_of(m, n, seed):
    # integers of more than 15 digits get a decimal point: without one the reader refuses them (pandas' int64 is exact there,
    # a double is not) -- test_files_that_are_not_plain_are_left_to_pandas covers that
    s = [x + "." if x.lstrip("+-").isdigit() and len(x.lstrip("+-")) > 15 else x for x in textcases.spellings(m * n, seed)]
    s[0] = "1.5"            # the first line types the columns: a float on top, as in a data file
    return [s[i * n:(i + 1) * n] for i in range(m)]


@pytest.mark.parametrize("delimiter,header,row_names,eol,final_newline,quote_names", [
    (",", True, True, "\n", True, False),
    (",", True, True, "\n", True, True),          # names as csv.QUOTE_NONNUMERIC writes them (the CLI's own U.csv / V.csv)
    ("\t", True, False, "\n", True, False),       # the reference's test data: tab-separated, header, no row names
    ("\t", False, False, "\n", False, False),     # ... and without a header, last line not terminated
    (",", False, True, "\r\n", True, False),
    (";", True, True, "\r\n", False, True),
    (" ", False, False, "\n", True, False),
    ("|", True, True, "\n", True, False),
])
def test_same_frame_as_pandas(tmp_path, delimiter, header, row_names, eol, final_newline, quote_names):
    m, n = 37, 211
    cells = cells_of(m, n, seed=len(delimiter) + 3 * m + ord(delimiter))
    path = str(tmp_path / "data.txt")
    write_matrix(path, cells, delimiter, header=["id"] + ["ENSP%d" % j for j in range(n)] if header else None,
                 row_names=["sample %d" % i if quote_names or delimiter != " " else "s%d" % i for i in range(m)] if row_names else None,
                 eol=eol, final_newline=final_newline, quote_names=quote_names)
    why = []
    got = textio.read_frame(path, delimiter, header, row_names, why=why)
    assert got is not None, why
    same_frame(got, pandas_frame(path, delimiter, header, row_names))


def test_a_million_spellings_through_the_kernel(tmp_path):
    """The conversion itself, at volume: 1000 x 1000 fields of every spelling (17+ digit mantissas, subnormals, leading zeros)."""
    cells = cells_of(1000, 1000, seed=11)
    path = str(tmp_path / "big.csv")
    write_matrix(path, cells, ",")
    got = textio.read_frame(path, ",", False, False)
    want = pandas_frame(path, ",", False, False)
    same_frame(got, want)
    exact = np.array([[float(c) for c in row] for row in cells])
    assert (want.to_numpy() != exact).mean() > 0.05          # not the correctly rounded conversion: pandas' own


def test_pieces_of_any_size(tmp_path, monkeypatch):
    cells = cells_of(50, 40, seed=4)
    path = str(tmp_path / "data.csv")
    write_matrix(path, cells, ",", header=["id"] + ["g%d" % j for j in range(40)], row_names=["r%d" % i for i in range(50)])
    want = pandas_frame(path, ",", True, True)
    for piece in (1, 7, 4096, 4097, 100000):
        monkeypatch.setattr(textio, "CHUNK_BYTES", piece)
        same_frame(textio.read_frame(path, ",", True, True), want)


def test_integers(tmp_path):
    rs = np.random.RandomState(3)
    counts = rs.randint(0, 100000, size=(30, 50))
    path = str(tmp_path / "counts.tsv")
    write_matrix(path, [[str(v) for v in row] for row in counts], "\t", header=["gene"] + ["g%d" % j for j in range(50)],
                 row_names=["s%d" % i for i in range(30)])
    got = textio.read_frame(path, "\t", True, True)
    same_frame(got, pandas_frame(path, "\t", True, True))
    assert got.to_numpy().dtype == np.int64
    # some integer-only columns among float columns (an all-zero gene written "0"): pandas types them int64; the command line
    # only takes .to_numpy() of the frame, and that matrix is what mixed_as_float returns
    cells = [["%.4f" % v for v in row] for row in rs.rand(30, 50)]
    for i in range(30):
        cells[i][7] = "0"
        cells[i][20] = "-0" if i == 3 else str(i)
    path = str(tmp_path / "mixed.csv")
    write_matrix(path, cells, ",")
    why = []
    assert textio.read_frame(path, ",", False, False, why=why) is None and "integers only" in why[0]
    got = textio.read_frame(path, ",", False, False, mixed_as_float=True)
    want = pandas_frame(path, ",", False, False)
    assert list(want.dtypes[[7, 20]]) == [np.int64, np.int64]
    a, b = got.to_numpy(), want.to_numpy()
    assert a.dtype == b.dtype == np.float64 and (a.view(np.int64) == b.view(np.int64)).all()


NOT_PLAIN = {
    "missing value": "a,b,c\n1,,3\n4,5,6\n",
    "NA": "a,b,c\n1,NA,3\n4,5,6\n",
    "nan": "a,b,c\n1,nan,3\n4,5,6\n",
    "inf": "a,b,c\n1,inf,3\n4,5,6\n",
    "quoted number": 'a,b,c\n1,"2",3\n4,5,6\n',
    "short line": "a,b,c\n1,2,3\n4,5\n7,8,9\n",
    "long line": "a,b,c\n1,2,3\n4,5,6,7\n",
    "blank line": "a,b,c\n1,2,3\n\n4,5,6\n",
    "blank line at the end": "a,b,c\n1,2,3\n4,5,6\n\n",
    "duplicate names": "a,b,a\n1,2,3\n4,5,6\n",
    "empty name": "a,,c\n1,2,3\n4,5,6\n",
    "header one field short": "b,c\nx,2,3\ny,5,6\n",
    "hex": "a,b,c\n1,0x10,3\n4,5,6\n",
    "exponent of five digits": "a,b,c\n1,1e-00005,3\n4,5,6\n",
    "overflow": "a,b,c\n1,1e400,3\n4,5,6\n",
    "sixteen-digit integer": "a,b,c\n1,1234567890123456,3\n4,5,6\n",
    "integer part beyond uint64": "a,b,c\n1.5,2.5,3.5\n4,74233762471773093888.000,6\n",
    "old Mac line ends": "a,b,c\r1,2,3\r4,5,6\r",
    "carriage return inside a row name": "id,a,b\nx,1.5,2.5\ny\rz,3.5,4.5\n",
    "carriage return inside a quoted header": '"id","a\rb","c"\nx,1.5,2.5\n',
    "corner cell equal to a column name": "a,a,b\nx,1.5,2.5\ny,3.5,4.5\n",
    "unbalanced quote in the header": '"id,a,b\nx,1.5,2.5\n',
    "byte-order mark": "﻿a,b,c\n1,2,3\n4,5,6\n",
    "text": "a,b,c\n1,two,3\n4,5,6\n",
    "thousands": 'a,b,c\n1,"2,000",3\n4,5,6\n',
    "header only": "a,b,c\n",
    "one line without end": "1,2,3",
    "a field of 5000 digits": "a,b,c\n1,%s,3\n4,5,6\n" % ("7" * 5000),
}


@pytest.mark.parametrize("what", sorted(NOT_PLAIN))
def test_files_that_are_not_plain_are_left_to_pandas(tmp_path, what):
    path = str(tmp_path / "data.csv")
    with open(path, "w", newline="", encoding="utf-8") as fh:
        fh.write(NOT_PLAIN[what])
    has_header = what != "one line without end"
    why = []
    row_names = what in ("header one field short", "carriage return inside a row name", "carriage return inside a quoted header",
                         "unbalanced quote in the header", "corner cell equal to a column name")
    assert textio.read_frame(path, ",", has_header, row_names, why=why) is None
    assert why and why[0]
    assert textio.read_frame(path + ".gz", ",", True, False, why=why) is None


def test_quoted_row_name_holding_the_delimiter(tmp_path):
    path = str(tmp_path / "data.csv")
    with open(path, "w") as fh:
        fh.write('id,a,b\n"x,1",1.5,2.5\ny,3.5,4.5\n')
    assert textio.read_frame(path, ",", True, True) is None          # one separator too many in that line
    with open(path, "w") as fh:
        fh.write('id,a,b\n"x;1",1.5,2.5\n"NA",3.5,4.5\n7,1.0,2.0\n')
    same_frame(textio.read_frame(path, ",", True, True), pandas_frame(path, ",", True, True))


def test_call_order_and_arguments():
    lib = _capi.load()
    h = C.c_void_p()
    assert lib.prmf_text_create(C.byref(h), 0, -1) == _capi.PRMF_ERR_ARG
    assert lib.prmf_text_create(C.byref(h), 0, 8) == _capi.PRMF_OK
    rows, flags = C.c_int64(), C.c_int()
    out = np.zeros(4)
    isint = np.zeros(2, dtype=np.int32)
    assert lib.prmf_text_parse(h, 0, out.ctypes.data_as(C.c_void_p), None, isint.ctypes.data_as(C.c_void_p), C.byref(flags), None) == _capi.PRMF_ERR_STATE
    assert b"before prmf_text_index" in lib.prmf_last_error(None)
    assert lib.prmf_text_append(h, b"1,2\n3,4\n5", 9) == _capi.PRMF_ERR_ARG          # more than the 8 bytes created for
    assert lib.prmf_text_append(h, b"1,2\n", 4) == _capi.PRMF_OK
    assert lib.prmf_text_append(h, b"3,4", 3) == _capi.PRMF_OK                      # no final newline: supplied
    for bad in (b"e", b"5", b".", b'"', b"\n", b"\x00"):
        assert lib.prmf_text_index(h, bad, 2, C.byref(rows), C.byref(flags)) == _capi.PRMF_ERR_ARG
    assert lib.prmf_text_index(h, b",", 0, C.byref(rows), C.byref(flags)) == _capi.PRMF_ERR_ARG
    assert lib.prmf_text_index(h, b",", 2, C.byref(rows), C.byref(flags)) == _capi.PRMF_OK
    assert (rows.value, flags.value) == (2, 0)
    assert lib.prmf_text_index(h, b",", 2, C.byref(rows), C.byref(flags)) == _capi.PRMF_ERR_STATE
    assert lib.prmf_text_append(h, b"1", 1) == _capi.PRMF_ERR_STATE
    assert lib.prmf_text_parse(h, 1, out.ctypes.data_as(C.c_void_p), None, isint.ctypes.data_as(C.c_void_p), C.byref(flags), None) == _capi.PRMF_ERR_ARG
    bad_at = C.c_int64()
    assert lib.prmf_text_parse(h, 0, out.ctypes.data_as(C.c_void_p), None, isint.ctypes.data_as(C.c_void_p), C.byref(flags), C.byref(bad_at)) == _capi.PRMF_OK
    assert flags.value == 0 and bad_at.value == -1 and out.tolist() == [1.0, 2.0, 3.0, 4.0] and isint.tolist() == [1, 1]
    assert lib.prmf_text_destroy(h) == _capi.PRMF_OK
    assert lib.prmf_text_destroy(None) == _capi.PRMF_OK
    # an empty body, and where the first refused field is
    assert lib.prmf_text_create(C.byref(h), 0, 0) == _capi.PRMF_OK
    assert lib.prmf_text_index(h, b",", 3, C.byref(rows), C.byref(flags)) == _capi.PRMF_OK and rows.value == 0
    lib.prmf_text_destroy(h)
    body = b"1,2\n3,x\n5,\n"
    assert lib.prmf_text_create(C.byref(h), 0, len(body)) == _capi.PRMF_OK
    assert lib.prmf_text_append(h, body, len(body)) == _capi.PRMF_OK
    assert lib.prmf_text_index(h, b",", 2, C.byref(rows), C.byref(flags)) == _capi.PRMF_OK and rows.value == 3
    out = np.zeros(6)
    assert lib.prmf_text_parse(h, 0, out.ctypes.data_as(C.c_void_p), None, isint.ctypes.data_as(C.c_void_p), C.byref(flags), C.byref(bad_at)) == _capi.PRMF_OK
    assert flags.value == _capi.PRMF_TEXT_NOT_NUMERIC and bad_at.value == body.index(b"x")
    lib.prmf_text_destroy(h)


def test_command_line_gives_the_same_files_with_either_reader(tmp_path, monkeypatch):
    """The whole CLI on the reference's test generator: U.csv, V.csv and obj.txt byte for byte whichever reader took --data."""
    from prmf_amd import runner
    case = synth.cli_case(str(tmp_path / "in"), variant=1)
    outs = {}
    for reader in ("pandas", "gpu"):
        monkeypatch.setenv("PRMF_TEXT_READER", reader)
        out = str(tmp_path / ("out_" + reader))
        os.makedirs(out)
        runner.main(["--data", case["data_header"], "--manifolds"] + case["manifolds"] + ["--outdir", out, "--seed", "1", "--delimiter", "\t"])
        outs[reader] = {name: open(os.path.join(out, name), "rb").read() for name in ("U.csv", "V.csv", "obj.txt")}
    assert outs["gpu"] == outs["pandas"]
    # and the reader did take it (the test data is plain)
    assert textio.read_frame(case["data_header"], "\t", True, False, mixed_as_float=True) is not None


def test_typical_size(tmp_path):
    """500 samples x 20000 genes would be 187 MB; a fifth of the rows is enough to cross several 4 KiB tiles per line and the
    64 MB piece size."""
    rs = np.random.RandomState(0)
    X = rs.gamma(5.0, size=(100, 6)).dot(rs.gamma(5.0, size=(20000, 6)).T)
    path = str(tmp_path / "data.csv")
    with open(path, "w") as fh:
        fh.write("," + ",".join("ENSP%d" % j for j in range(20000)) + "\n")
        for i in range(100):
            fh.write("s%d," % i + ",".join(map(repr, X[i].tolist())) + "\n")
    got = textio.read_frame(path, ",", True, True)
    want = pandas_frame(path, ",", True, True)
    same_frame(got, want)
    assert (got.to_numpy() == X).mean() > 0.5
