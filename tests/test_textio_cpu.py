"""The text reader's host side without a GPU: the number conversion (prmf_amd/csrc/xstrtod.h, compiled for the host --
the same function the GPU kernel calls) against pandas' reader bit for bit, the header rules, the reader switch."""
import ctypes
import io
import os
import subprocess

import numpy as np
import pandas as pd
import pytest

import textcases
from prmf_amd import textio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_parser(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("xstrtod") / "libxstrtod_host.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-I", os.path.join(ROOT, "prmf_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "xstrtod_host.cpp"), "-o", so], check=True)
    lib = ctypes.CDLL(so)

    def parse(strings):
        text = ("\n".join(strings) + "\n").encode()
        off = np.zeros(len(strings) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(s.encode()) + 1 for s in strings])
        out = np.zeros(len(strings))
        verdict = np.zeros(len(strings), dtype=np.int32)
        lib.parse_many(text, off.ctypes.data_as(ctypes.c_void_p), ctypes.c_longlong(len(strings)), out.ctypes.data_as(ctypes.c_void_p),
                       verdict.ctypes.data_as(ctypes.c_void_p))
        return out, verdict
    return parse


def test_conversion_equals_pandas_bit_for_bit(host_parser):
    strings = textcases.spellings(1000000, seed=5)
    want = pd.read_csv(io.StringIO("x\n" + "\n".join(strings) + "\n"))["x"].to_numpy()
    assert want.dtype == np.float64
    got, verdict = host_parser(strings)
    assert (verdict & 1).all()
    assert (got.view(np.int64) == want.view(np.int64)).all()
    # the comparison has teeth: pandas' converter is not the correctly rounded one on this mix
    exact = np.array([float(s) for s in strings])
    assert (want != exact).mean() > 0.05


def test_what_the_grammar_refuses(host_parser):
    refused = ["", " ", "NA", "nan", "NaN", "inf", "-inf", "Infinity", "0x10", "1e", "1e+", "e5", ".", "-", "+", "1.2.3", "1,5", '"1.5"', "1 2",
               "1e99999", "1e400", "2e308", "1.5f", "--1", "1_000", "٣"]
    _, verdict = host_parser(refused)
    assert not (verdict & 1).any(), [s for s, v in zip(refused, verdict) if v & 1]
    taken = ["1", " 1", "1 ", "\t1.5\t", "+1", "-0", "007", "1.", ".5", "1e5", "1E-5", "1e+05", "1.7976931348623157e308", "4.9e-324", "1e-400",
             "123456789012345", "1234567890123456", "0.1e-3"]
    values, verdict = host_parser(taken)
    assert (verdict & 1).all()
    want = pd.read_csv(io.StringIO("x\n" + "\n".join(s.replace("\t", " ") for s in taken) + "\n"), dtype=np.float64, skipinitialspace=False)["x"].to_numpy()
    assert (values.view(np.int64) == want.view(np.int64)).all()
    is_int = {s: bool(v & 2) for s, v in zip(taken, verdict)}
    assert is_int["1"] and is_int["-0"] and is_int["007"] and is_int[" 1"] and not is_int["1."] and not is_int["1e5"]
    wide = {s: bool(v & 4) for s, v in zip(taken, verdict)}
    assert wide["1234567890123456"] and not wide["123456789012345"]
    _, verdict = host_parser(textcases.wide_spellings() + ["123456789012345678.5", "000000000000000000001.5", "1e30"])
    assert [bool(v & 4) for v in verdict] == [True] * len(textcases.wide_spellings()) + [False, False, False]


def test_why_long_integer_parts_are_refused():
    """pandas types a column by position when an integer probe overflows: the same field makes it float64 below a
    float and object below an integer.  The reader does not repeat that; it refuses the field (PRMF_TEXT_WIDE_INT)."""
    big = "74233762471773093888.000"
    assert pd.read_csv(io.StringIO("x\n1.5\n%s\n" % big))["x"].dtype == np.float64
    assert pd.read_csv(io.StringIO("x\n%s\n1.5\n" % big))["x"].dtype == object
    assert pd.read_csv(io.StringIO("x\n3\n%s\n1.5\n" % big))["x"].dtype == object


def test_header_rules():
    assert textio._split_header(b"a,b,c\n", ",") == ["a", "b", "c"]
    assert textio._split_header(b'"a","b,1","c"\r\n', ",") == ["a", "b,1", "c"]
    assert textio._split_header(b"\tg1\tg2\n", "\t") == ["", "g1", "g2"]
    with pytest.raises(textio.NotPlain):
        textio._split_header(b"\xef\xbb\xbfa,b\n", ",")
    with pytest.raises(textio.NotPlain):
        textio._split_header(b"\xff\xfe,b\n", ",")


def test_reader_switch(tmp_path, monkeypatch):
    small = tmp_path / "small.csv"
    small.write_text("1,2\n")
    big = tmp_path / "big.csv"
    big.write_bytes(b"1,2\n" * (textio.AUTO_MIN_BYTES // 4 + 1))
    monkeypatch.delenv("PRMF_TEXT_READER", raising=False)
    assert not textio.wanted(str(small)) and textio.wanted(str(big))
    monkeypatch.setenv("PRMF_TEXT_READER", "gpu")
    assert textio.wanted(str(small))
    monkeypatch.setenv("PRMF_TEXT_READER", "pandas")
    assert not textio.wanted(str(big))
    monkeypatch.setenv("PRMF_TEXT_READER", "sometimes")
    with pytest.raises(ValueError):
        textio.wanted(str(big))


@pytest.mark.parametrize("delimiter", ["ab", "é", "e", "5", ".", '"', "-"])
def test_delimiters_that_cannot_separate_numbers_are_left_to_pandas(tmp_path, delimiter):
    f = tmp_path / "x.txt"
    f.write_text("1\n2\n")
    why = []
    assert textio.read_frame(str(f), delimiter, False, False, why=why) is None and why


def _pandas_bytes(tmp_path, values, index, columns):
    import csv
    p = str(tmp_path / "want.csv")
    pd.DataFrame(values, index=index, columns=columns).to_csv(p, sep=",", index=index is not None, quoting=csv.QUOTE_NONNUMERIC)
    return open(p, "rb").read()


def test_written_matrix_equals_pandas_byte_for_byte(tmp_path):
    """prmf_write_csv (host code of the library: runs without a GPU) against DataFrame.to_csv(quoting=QUOTE_NONNUMERIC), the
    call that writes U.csv / V.csv (prmf_runner.py:1068-1071): repr() of every float, quoted labels."""
    rs = np.random.RandomState(8)
    special = [0.0, -0.0, 1.0, -1.0, 0.1, 1e16, 1e15, 9999999999999998.0, 1e-4, 1e-5, 0.00012345, 123456789012345678.0, 5e-324, 2.2250738585072014e-308,
               1.7976931348623157e308, float("inf"), float("-inf"), float("nan"), 1e22, 1e23, 1.5e300, 2.5e-300, 100.0, 1e100, 123456.789,
               float(np.float32(0.3)), float(np.float32(1e-7)), float(np.finfo(np.float32).eps)]
    a = np.concatenate([np.array(special), rs.rand(300), rs.gamma(2.0, size=300) * 10.0 ** rs.randint(-30, 30, 300),
                        rs.rand(300).astype(np.float32).astype(np.float64), np.frombuffer(rs.bytes(8 * 372), dtype=np.float64)])
    a = a[:1300].reshape(100, 13)
    cols = ["LV%d" % i for i in range(13)]
    cases = [(None, "no index"), (["s%d" % i for i in range(100)], "strings"), (list(range(100, 200)), "integers"),
             (['he said "x"', "a,b", "naïve", "", " lead"] + ["r%d" % i for i in range(95)], "awkward strings"),
             (list(np.arange(100, dtype=np.int64)), "numpy integers")]
    for index, what in cases:
        out = str(tmp_path / "got.csv")
        assert textio.write_frame(out, a, index, cols), what
        assert open(out, "rb").read() == _pandas_bytes(tmp_path, a, index, cols), what
    # the threaded path (>= 65536 cells), a matrix as the solver returns it (float32 values held in float64)
    big = (3 * rs.rand(3000, 64)).astype(np.float32).astype(np.float64)
    out = str(tmp_path / "big.csv")
    assert textio.write_frame(out, big, ["g%d" % i for i in range(3000)], ["LV%d" % i for i in range(64)])
    assert open(out, "rb").read() == _pandas_bytes(tmp_path, big, ["g%d" % i for i in range(3000)], ["LV%d" % i for i in range(64)])


def test_writer_leaves_what_is_not_plain_to_pandas(tmp_path):
    a = np.ones((3, 2))
    out = str(tmp_path / "x.csv")
    assert not textio.write_frame(out, a.astype(np.float32), None, ["a", "b"])
    assert not textio.write_frame(out, a, [1.5, 2.5, 3.5], ["a", "b"])
    assert not textio.write_frame(out, a, ["a", 1, "c"], ["a", "b"])
    assert not textio.write_frame(out, a, ["a", "line\nbreak", "c"], ["a", "b"])
    assert not textio.write_frame(out, a, [True, False, True], ["a", "b"])
    assert not textio.write_frame(out, a, None, ["a", 2])
    assert not os.path.exists(out)
    # an I/O failure is the exception DataFrame.to_csv raises for it (PRMF_ERR_IO -> OSError with the errno)
    with pytest.raises(OSError) as ei:
        textio.write_frame(str(tmp_path / "no" / "such" / "dir.csv"), a, None, ["a", "b"])
    import errno
    assert ei.value.errno == errno.ENOENT
