"""The per-family lines of `report` are formatted with std::to_chars (general, precision 6) instead of printf("%g") --
2.4x cheaper, and specified to give the same text; this pins it on the values a report prints (p-values k/1000, products of
them, tiny and exact-halfway decimals) and on random doubles.  Host only."""
import ctypes as C

import numpy as np


def test_report_number_formatting_equals_printf():
    from cafe_amd import _lib
    L = _lib.load()
    r = np.random.default_rng(7)
    parts = [
        np.arange(0, 1001) / 1000.0,                                        # Monte-Carlo p-values
        r.random(400000),                                                   # Viterbi / branch p-values
        r.random(200000) * 10.0 ** r.integers(-12, 3, 200000),              # small and large magnitudes (exponent form below 1e-4)
        np.nextafter(np.round(r.random(200000), 6), r.choice([-1.0, 2.0], 200000)),   # next to a 6-digit decimal: rounding cases
        np.round(r.random(100000), 5) + 5e-7,                               # near halfway between two 6-digit outputs
        np.array([0.0, 1.0, 0.5, 1e-4, 9.99999e-5, 0.000123456789, 999999.5, 1e6, 123456.7, 1e-300, 5e-324, 1.7976931348623157e308]),
        -r.random(1000),
    ]
    v = np.ascontiguousarray(np.concatenate(parts), np.float64)
    bad_value = C.c_double(0.0)
    bad = L.cafehost_format_selftest(v.ctypes.data_as(C.POINTER(C.c_double)), len(v), C.byref(bad_value))
    assert bad == 0, "first value formatted differently: %r" % bad_value.value
