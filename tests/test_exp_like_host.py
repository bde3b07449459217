"""exp() as this host's libm computes it, restated for the device (cafe_amd/csrc/exp_like_host.hpp): the HOST build of the
same function against the host's exp(), bit for bit, on ten million arguments (the whole range, the range the terms of a
transition matrix live in, the special range below -512, around zero).  One of the two forms -- compiled with / without fused
multiply-add, as x86-64 glibc selects at load time -- must match EXACTLY on a glibc host; the library then builds the
report phase's matrices with it (tests/test_gpu_parity.py checks those against the oracle's, entry by entry).  CPU only."""
import ctypes as C


def test_one_restated_form_is_this_hosts_exp_bit_for_bit():
    from cafe_amd import _lib
    L = _lib.load()
    fused, plain = C.c_long(-1), C.c_long(-1)
    variant = L.cafehip_exp_like_host_selftest(10_000_000, 20261005, C.byref(fused), C.byref(plain))
    print("host exp(): fused form misses %d, plain form misses %d of 10,000,000 -> variant %d" % (fused.value, plain.value, variant))
    assert variant in (1, 2), "neither restated form is this host's exp(): the device library's exp will be used (not an error on a non-glibc host)"
    assert (fused.value if variant == 1 else plain.value) == 0
    # the two forms are different functions: the other one must miss some arguments, or the detection proves nothing
    assert (plain.value if variant == 1 else fused.value) > 0
    # another seed, same verdict
    assert L.cafehip_exp_like_host_selftest(1_000_000, 7, C.byref(fused), C.byref(plain)) == variant
