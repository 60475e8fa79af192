// Decimal text -> float64 exactly as pandas' C parser does it -- the conversion behind `pd.read_csv(args.data, ...)` at
// script/prmf_runner.py:942 (pandas >= 1.2: float_precision "high" is the default, its converter is `precise_xstrtod`).
// pandas is a dependency of the reference, not part of it; its published algorithm, restated:
//   up to 17 digits (leading zeros included) are accumulated in a double as number * 10 + digit -- two rounded
//   operations, never a fused one --, further integer digits raise the decimal exponent, further fraction digits are
//   dropped; then ONE multiplication or division by an exactly represented power of ten from a table (two divisions
//   below 1e-308).  The result is within an ulp of the correctly rounded value, not always equal to it (about a third
//   of 17-digit inputs differ from strtod) -- so equality with the reference's X needs these operations, in this order.
// Pinned against pandas itself: tests/test_textio_cpu.py compiles this header for the host and compares 10^6 random
// spellings with pd.read_csv bit for bit; tests/test_gpu_textio.py does the same through the GPU kernel.
//
// The grammar accepted here is the plain one --  [ \t]* [+-]? digits* ( . digits* )? ( [eE] [+-]? digits{1,4} )? [ \t]*
// with at least one mantissa digit.  Anything else (empty fields, NA / nan / inf, quotes, hex, thousands separators,
// a result that overflows) is REFUSED, and the caller hands the whole file to pandas: no second opinion on what such a
// field means is formed here.
#pragma once
#include <cstdint>

#ifndef PRMF_HD
#ifdef __HIPCC__
#define PRMF_HD __host__ __device__
#else
#define PRMF_HD
#endif
#endif

namespace prmf {

#ifdef __HIP_DEVICE_COMPILE__
__device__
#endif
static const double TEXT_POW10[309] = {
    1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7,
    1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
    1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22, 1e23,
    1e24, 1e25, 1e26, 1e27, 1e28, 1e29, 1e30, 1e31,
    1e32, 1e33, 1e34, 1e35, 1e36, 1e37, 1e38, 1e39,
    1e40, 1e41, 1e42, 1e43, 1e44, 1e45, 1e46, 1e47,
    1e48, 1e49, 1e50, 1e51, 1e52, 1e53, 1e54, 1e55,
    1e56, 1e57, 1e58, 1e59, 1e60, 1e61, 1e62, 1e63,
    1e64, 1e65, 1e66, 1e67, 1e68, 1e69, 1e70, 1e71,
    1e72, 1e73, 1e74, 1e75, 1e76, 1e77, 1e78, 1e79,
    1e80, 1e81, 1e82, 1e83, 1e84, 1e85, 1e86, 1e87,
    1e88, 1e89, 1e90, 1e91, 1e92, 1e93, 1e94, 1e95,
    1e96, 1e97, 1e98, 1e99, 1e100, 1e101, 1e102, 1e103,
    1e104, 1e105, 1e106, 1e107, 1e108, 1e109, 1e110, 1e111,
    1e112, 1e113, 1e114, 1e115, 1e116, 1e117, 1e118, 1e119,
    1e120, 1e121, 1e122, 1e123, 1e124, 1e125, 1e126, 1e127,
    1e128, 1e129, 1e130, 1e131, 1e132, 1e133, 1e134, 1e135,
    1e136, 1e137, 1e138, 1e139, 1e140, 1e141, 1e142, 1e143,
    1e144, 1e145, 1e146, 1e147, 1e148, 1e149, 1e150, 1e151,
    1e152, 1e153, 1e154, 1e155, 1e156, 1e157, 1e158, 1e159,
    1e160, 1e161, 1e162, 1e163, 1e164, 1e165, 1e166, 1e167,
    1e168, 1e169, 1e170, 1e171, 1e172, 1e173, 1e174, 1e175,
    1e176, 1e177, 1e178, 1e179, 1e180, 1e181, 1e182, 1e183,
    1e184, 1e185, 1e186, 1e187, 1e188, 1e189, 1e190, 1e191,
    1e192, 1e193, 1e194, 1e195, 1e196, 1e197, 1e198, 1e199,
    1e200, 1e201, 1e202, 1e203, 1e204, 1e205, 1e206, 1e207,
    1e208, 1e209, 1e210, 1e211, 1e212, 1e213, 1e214, 1e215,
    1e216, 1e217, 1e218, 1e219, 1e220, 1e221, 1e222, 1e223,
    1e224, 1e225, 1e226, 1e227, 1e228, 1e229, 1e230, 1e231,
    1e232, 1e233, 1e234, 1e235, 1e236, 1e237, 1e238, 1e239,
    1e240, 1e241, 1e242, 1e243, 1e244, 1e245, 1e246, 1e247,
    1e248, 1e249, 1e250, 1e251, 1e252, 1e253, 1e254, 1e255,
    1e256, 1e257, 1e258, 1e259, 1e260, 1e261, 1e262, 1e263,
    1e264, 1e265, 1e266, 1e267, 1e268, 1e269, 1e270, 1e271,
    1e272, 1e273, 1e274, 1e275, 1e276, 1e277, 1e278, 1e279,
    1e280, 1e281, 1e282, 1e283, 1e284, 1e285, 1e286, 1e287,
    1e288, 1e289, 1e290, 1e291, 1e292, 1e293, 1e294, 1e295,
    1e296, 1e297, 1e298, 1e299, 1e300, 1e301, 1e302, 1e303,
    1e304, 1e305, 1e306, 1e307, 1e308
};

struct ParsedField {
  double value;
  bool ok;        // the plain grammar, finite result
  bool is_int;    // no '.', no exponent: pandas tries int64 on such a column first
  bool wide_int;  // an integer of more than 15 digits (exact in pandas' int64, not in a double), or an integer PART of 19
                  // or more significant digits: pandas probes a column as int64, then uint64, before float64, and a probe that
                  // overflows (instead of meeting the '.') types the column by what stood above that field -- not repeated here
};

// [p, end): one field without its separator
PRMF_HD inline ParsedField parse_decimal(const unsigned char* p, const unsigned char* end) {
#pragma clang fp contract(off)
  ParsedField r{0.0, false, true, false};
  while (p < end && (*p == ' ' || *p == '\t')) ++p;
  bool negative = false;
  if (p < end && (*p == '-' || *p == '+')) {
    negative = *p == '-';
    ++p;
  }
  double number = 0.0;
  int exponent = 0, num_digits = 0, num_decimals = 0, int_digits = 0, significant = 0;
  while (p < end && *p >= '0' && *p <= '9') {
    if (significant || *p != '0') ++significant;
    if (num_digits < 17) {
      number = number * 10.0;
      number = number + (double)(*p - '0');
      ++num_digits;
    } else {
      ++exponent;
    }
    ++int_digits;
    ++p;
  }
  if (p < end && *p == '.') {
    r.is_int = false;
    ++p;
    while (num_digits < 17 && p < end && *p >= '0' && *p <= '9') {
      number = number * 10.0;
      number = number + (double)(*p - '0');
      ++p;
      ++num_digits;
      ++num_decimals;
    }
    if (num_digits >= 17)
      while (p < end && *p >= '0' && *p <= '9') ++p;
    exponent -= num_decimals;
  }
  if (num_digits == 0) return r;
  if (negative) number = -number;
  if (p < end && (*p == 'e' || *p == 'E')) {
    r.is_int = false;
    ++p;
    bool eneg = false;
    if (p < end && (*p == '-' || *p == '+')) {
      eneg = *p == '-';
      ++p;
    }
    int n = 0, nd = 0;
    while (p < end && *p >= '0' && *p <= '9') {
      if (++nd > 4) return r;
      n = n * 10 + (*p - '0');
      ++p;
    }
    if (nd == 0) return r;
    exponent += eneg ? -n : n;
  }
  while (p < end && (*p == ' ' || *p == '\t')) ++p;
  if (p != end) return r;
  if (exponent > 308) return r;                       // pandas: ERANGE -> the column is not numeric
  if (exponent > 0) {
    number = number * TEXT_POW10[exponent];
  } else if (exponent < -308) {
    if (exponent < -616) {
      number = 0.0;
    } else {
      number = number / TEXT_POW10[-308 - exponent];
      number = number / TEXT_POW10[308];
    }
  } else {
    number = number / TEXT_POW10[-exponent];
  }
  if (number > 1.7976931348623157e308 || number < -1.7976931348623157e308) return r;   // HUGE_VAL: ERANGE
  r.value = number;
  r.ok = true;
  r.wide_int = (r.is_int && int_digits > 15) || significant >= 19;
  return r;
}

}  // namespace prmf
