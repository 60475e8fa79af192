"""Spellings of decimal numbers for the text-reader tests (CPU: the host build of xstrtod.h; GPU: the parse kernel),
both compared with pandas' own reader."""
import random
import struct


def spellings(n, seed):
    """n strings pandas reads as finite floats, of every shape a data file holds: repr, %g, %e, %f with up to 25 digits,
    integers with and without a trailing point, leading zeros, 17+ digit mantissas (where pandas' converter is NOT the
    correctly rounded one), subnormals, explicit signs."""
    rnd = random.Random(seed)
    out = []
    while len(out) < n:
        t = rnd.random()
        if t < 0.3:
            s = repr(rnd.random() * 10 ** rnd.randint(-5, 8))
        elif t < 0.45:
            s = "%.*e" % (rnd.randint(0, 20), rnd.random() * 10 ** rnd.randint(-320, 307))
        elif t < 0.6:
            s = "%.*f" % (rnd.randint(0, 25), rnd.random() * 10 ** rnd.randint(-8, 20))
        elif t < 0.7:
            s = "-" + repr(rnd.random())
        elif t < 0.8:
            s = repr(struct.unpack("d", struct.pack("Q", rnd.getrandbits(62)))[0])
        elif t < 0.85:
            s = "0.000" + str(rnd.getrandbits(70))
        elif t < 0.9:
            s = str(rnd.getrandbits(rnd.randint(1, 49))) + rnd.choice([".", ".0", "e0", "E+2", "e-3", ".5"])
        elif t < 0.95:
            s = rnd.choice(["+", ""]) + "%.6g" % (rnd.random() * 10 ** rnd.randint(-10, 10))
        else:
            s = "." + str(rnd.getrandbits(40))
        if "inf" in s or "nan" in s:
            continue
        # an integer part of 19+ digits: pandas' int64 / uint64 probes overflow on it and the column's dtype then depends on
        # the rows above (object if they were all integers) -- the reader refuses such fields, see wide_spellings
        if len(s.lstrip("+-").split(".")[0].split("e")[0].lstrip("0")) >= 19:
            continue
        out.append(s)
    return out


def wide_spellings():
    """Fields the reader must refuse although they are decimal numbers: pandas types their column by position."""
    return ["74233762471773093888.000", "18446744073709551616.0", "9223372036854775808", "1234567890123456789.5", "-9223372036854775809.0",
            "0001234567890123456789.0", "1234567890123456", "99999999999999999999e-5"]
