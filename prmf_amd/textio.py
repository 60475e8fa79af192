"""--data as delimited text, read through the GPU.

The reference reads its data matrix with `pd.read_csv(args.data, sep=delimiter, header='infer' | None, index_col=0 | None)`
(script/prmf_runner.py:942).  At its typical size (500 samples x 20000 genes, 187 MB of text) that parse is 1.7 s of a 2.0 s
run (profiles/r05_cli_end_to_end.txt).  `read_frame` returns the SAME DataFrame -- values bit for bit (the GPU repeats
pandas' own float64 operations, prmf_amd/csrc/xstrtod.h), index, column labels, their names and dtypes -- with the numeric
body parsed by `prmf_text_*` (prmf_amd/csrc/textio.hip):

  * the header line (if any) is split here; the row names (if any) are cut out of the text at the offsets the GPU reports
    and read by pandas itself, so their type inference (strings, NA, ...) is pandas';
  * the body goes to the device in 64 MB pieces and comes back as one float64 matrix.

It is an accelerator for the plain case, not a second CSV dialect: whatever is not plain -- quotes in the body, missing
values, ragged or blank lines, duplicate or empty column names, compressed files, multi-character delimiters, columns of
which only some are integers written without a decimal point -- makes `read_frame` return None (with the reason in
`why`), and the caller reads the file with pandas as before.  No value is ever produced by a second interpretation.
PRMF_TEXT_READER = "auto" (default: files of at least 4 MB) | "gpu" (always try) | "pandas" (never)."""
import csv
import ctypes as C
import io
import os

import numpy as np

from . import _capi

CHUNK_BYTES = 64 << 20
AUTO_MIN_BYTES = 4 << 20
_COMPRESSED = (".gz", ".bz2", ".zip", ".xz", ".zst", ".tar")


class NotPlain(Exception):
    """The file is outside what the GPU reader takes; the message says why."""


def _split_header(line, delimiter):
    """Column labels of a header line the way pandas' C parser names them, for the plain case only: unique, non-empty
    labels (quoted or not).  Anything pandas would rename ("Unnamed: 3", "x.1") is refused."""
    text = line.rstrip(b"\n")
    if text.endswith(b"\r"):
        text = text[:-1]
    try:
        text = text.decode("utf-8")
    except UnicodeDecodeError:
        raise NotPlain("header is not UTF-8")
    if "\r" in text:
        raise NotPlain("carriage return inside the header")
    if text.startswith("﻿"):
        raise NotPlain("byte-order mark")
    if '"' in text:
        try:
            rows = list(csv.reader(io.StringIO(text), delimiter=delimiter, strict=True))      # strict: an open quote is an error
        except csv.Error as e:
            raise NotPlain("header: %s" % e)
        if len(rows) != 1:
            raise NotPlain("header spans several lines")
        names = rows[0]
    else:
        names = text.split(delimiter)
    return names


def read_frame(path, delimiter, has_header, has_row_names, device=0, why=None, mixed_as_float=False):
    """-> DataFrame equal to pd.read_csv(path, sep=delimiter, header='infer' if has_header else None,
    index_col=0 if has_row_names else None), or None if the file is not plain (why: a list that receives the reason).
    mixed_as_float: when only SOME columns hold nothing but integers (pandas types those int64 and the others float64),
    return one float64 frame anyway -- `.to_numpy()` of pandas' frame, which is all the command line uses, has exactly
    these values -- instead of refusing."""
    try:
        return _read_frame(path, delimiter, has_header, has_row_names, device, mixed_as_float)
    except NotPlain as e:
        if why is not None:
            why.append(str(e))
        return None
    except _capi.PrmfError as e:
        # a HIP failure on the way (text + 16 B per field must fit the free HBM; every rank of a multi-process run reads):
        # a file pandas can read is still read -- by pandas.  Argument / call-order errors are bugs and surface.
        if e.code != _capi.PRMF_ERR_HIP:
            raise
        if why is not None:
            why.append("the GPU reader failed (%s)" % e)
        return None


def wanted(path):
    mode = os.environ.get("PRMF_TEXT_READER", "auto")
    if mode not in ("auto", "gpu", "pandas"):
        raise ValueError("PRMF_TEXT_READER must be auto, gpu or pandas")
    if mode == "pandas":
        return False
    if mode == "gpu":
        return True
    try:
        return os.path.getsize(path) >= AUTO_MIN_BYTES
    except OSError:
        return False


def _read_frame(path, delimiter, has_header, has_row_names, device, mixed_as_float):
    import pandas as pd
    if not isinstance(delimiter, str) or len(delimiter) != 1 or ord(delimiter) > 127:
        raise NotPlain("delimiter is not one ASCII character")
    if delimiter in '"\r\n.+-eE0123456789' or delimiter == "\0":
        raise NotPlain("delimiter can be part of a number")
    if str(path).lower().endswith(_COMPRESSED):
        raise NotPlain("compressed file")
    rn = 1 if has_row_names else 0
    size = os.path.getsize(path)
    with open(path, "rb") as fh:
        first = fh.readline()
        if not first.endswith(b"\n"):
            raise NotPlain("one line only")
        if has_header:
            names = _split_header(first, delimiter)
            body_off = len(first)
            fields = len(names)
            labels = names[rn:]
            # (pandas renames duplicates over ALL names of the line, the corner cell above the row names included)
            if any(lab == "" for lab in labels) or len(set(names)) != len(names):
                raise NotPlain("empty or duplicate column names")
            columns = pd.Index(labels, dtype=object)
            index_name = (names[0] or None) if rn else None
        else:
            if b'"' in first:
                raise NotPlain("quotes in the first line")
            body_off = 0
            fields = first.count(delimiter.encode()) + 1
            # header=None: integer labels; with index_col=0 column 0 became the index, the others keep their numbers
            columns = pd.Index(range(rn, fields), dtype=np.int64)
            index_name = 0 if rn else None
        if fields - rn < 1:
            raise NotPlain("no data column")
        lib = _capi.load()
        handle = C.c_void_p()
        _capi.check(lib, None, lib.prmf_text_create(C.byref(handle), int(device), size - body_off))
        try:
            fh.seek(body_off)
            buf = bytearray(min(CHUNK_BYTES, max(1, size - body_off)))
            cbuf = (C.c_char * len(buf)).from_buffer(buf)
            while True:
                got = fh.readinto(buf)
                if not got:
                    break
                _capi.check(lib, None, lib.prmf_text_append(handle, cbuf, got))
            rows, flags = C.c_int64(), C.c_int()
            _capi.check(lib, None, lib.prmf_text_index(handle, delimiter.encode(), fields, C.byref(rows), C.byref(flags)))
            if flags.value:
                raise NotPlain("lines of different lengths or a bare carriage return (flags %d)" % flags.value)
            m, n = rows.value, fields - rn
            if m == 0:
                raise NotPlain("no data line")
            body = np.empty((m, n), dtype=np.float64)
            spans = np.empty((m, 2), dtype=np.int64)
            col_is_int = np.empty(n, dtype=np.int32)
            bad = C.c_int64()
            _capi.check(lib, None, lib.prmf_text_parse(handle, rn, body.ctypes.data_as(C.c_void_p), spans.ctypes.data_as(C.c_void_p),
                                                       col_is_int.ctypes.data_as(C.c_void_p), C.byref(flags), C.byref(bad)))
            if flags.value:
                raise NotPlain("field at byte %d of the body is not a plain decimal number (flags %d)" % (bad.value, flags.value))
        finally:
            lib.prmf_text_destroy(handle)
    n_int = int(col_is_int.sum())
    if n_int == n:
        body = body.astype(np.int64)              # pandas: every column int64 (<= 15 digits each: exact)
    elif n_int:
        if not mixed_as_float:
            raise NotPlain("%d of %d columns hold integers only (pandas would type them int64)" % (n_int, n))
        cols = np.flatnonzero(col_is_int)
        body[:, cols] = body[:, cols].astype(np.int64)      # through int64 as pandas' upcast goes: "-0" is +0.0
    if rn:
        # the names, cut out of the file at the GPU's offsets, read by pandas: its quoting rules and its type inference
        import mmap
        with open(path, "rb") as fh, mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ) as mm:
            lines = [mm[body_off + a:body_off + b] for a, b in spans.tolist()]
        try:
            sub = pd.read_csv(io.BytesIO(b"\n".join(lines) + b"\n"), sep=delimiter, header=None, index_col=0, skip_blank_lines=False)
        except (pd.errors.ParserError, pd.errors.EmptyDataError, UnicodeDecodeError, ValueError) as e:
            raise NotPlain("row names: %s" % e)
        if len(sub.index) != m or sub.shape[1] != 0:
            raise NotPlain("row names that pandas reads as something else than one field each")
        index = sub.index
        index.name = index_name
    else:
        index = pd.RangeIndex(m)
    return pd.DataFrame(body, index=index, columns=columns, copy=False)


def write_frame(path, values, index, columns):
    """pd.DataFrame(values, index=index, columns=columns).to_csv(path, sep=",", index=index is not None,
    quoting=csv.QUOTE_NONNUMERIC) -- the call that writes U.csv / V.csv (script/prmf_runner.py:1068-1071) -- byte for byte,
    with the floats formatted by `prmf_write_csv` (repr digits, on all cores) instead of one Python object per cell:
    3.0 s -> 0.2 s for 50000 x 64 + 20000 x 64.  Takes the plain case only -- a 2-D float64 matrix, labels that are all
    `str` or all `int` -- and returns False otherwise (the caller then writes with pandas)."""
    a = np.asarray(values)
    if a.dtype != np.float64 or a.ndim != 2 or a.shape[1] < 1:
        return False
    columns = list(columns)
    if len(columns) != a.shape[1] or not all(type(c) is str for c in columns):
        return False
    buf = io.StringIO()
    w = csv.writer(buf, delimiter=",", quoting=csv.QUOTE_NONNUMERIC, lineterminator="\n")
    prefixes = offsets = None
    keep = []
    if index is not None:
        index = list(index)
        if len(index) != a.shape[0]:
            return False
        if all(type(x) is str for x in index):
            pass
        elif all(isinstance(x, (int, np.integer)) and not isinstance(x, (bool, np.bool_)) for x in index):
            index = [int(x) for x in index]
        else:
            return False
        w.writerow([""] + columns)                  # the index has no name: pandas writes the empty label, quoted
        header = buf.getvalue().encode("utf-8")
        buf = io.StringIO()
        w = csv.writer(buf, delimiter=",", quoting=csv.QUOTE_NONNUMERIC, lineterminator="\n")
        w.writerows([x] for x in index)
        cells = buf.getvalue().encode("utf-8").split(b"\n")[:-1]
        if len(cells) != len(index):                # a label with a line break inside: not plain
            return False
        blob = b",".join(cells) + b","
        offsets = np.zeros(len(cells) + 1, dtype=np.int64)
        np.cumsum([len(c) + 1 for c in cells], out=offsets[1:])
        prefixes = blob
        keep = [blob, offsets]
    else:
        w.writerow(columns)
        header = buf.getvalue().encode("utf-8")
    a = np.ascontiguousarray(a)
    lib = _capi.load()
    rc = lib.prmf_write_csv(os.fsencode(path), header, prefixes, offsets.ctypes.data_as(C.c_void_p) if offsets is not None else None,
                            a.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1], a.shape[1], b",", 0)
    del keep
    if rc == _capi.PRMF_ERR_IO:      # what DataFrame.to_csv raises for the same failure (script/prmf_runner.py:1068-1071)
        import re
        msg = lib.prmf_last_error(None).decode("utf-8", "replace")
        hit = re.search(r"errno (\d+): (.*)$", msg)
        raise OSError(int(hit.group(1)) if hit else 5, hit.group(2) if hit else msg, os.fspath(path))
    _capi.check(lib, None, rc)
    return True
