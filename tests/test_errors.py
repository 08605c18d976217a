"""-m gpu: the error paths of the C-ABI — status codes instead of undefined behaviour (include/bnpk.h: "every function
returns 0 or a negative bnpk_status; nothing throws across the ABI")."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OK, ERR_ARG, ERR_ALIGN, ERR_HIP, ERR_NOMEM, ERR_NODEVICE, ERR_RANGE = 0, -1, -2, -3, -4, -5, -6


@pytest.fixture(scope="module")
def env():
    import torch
    from bionumpy_amd._native import lib
    from bionumpy_amd.device import Device, ptr
    return lib, Device.get(), ptr, torch


def test_sizes_beyond_the_supported_range_are_refused_before_anything_is_launched(env):
    lib, dev, ptr, torch = env
    keys = torch.zeros(16, dtype=torch.int64, device="cuda")
    out = torch.zeros(16, dtype=torch.int64, device="cuda")
    child = torch.zeros(1025, dtype=torch.int64, device="cuda")
    too_many = 1 << 35
    assert lib.bnpk_radix_partition(dev.ctx, ptr(keys), too_many, None, 1, 52, 10, ptr(out), ptr(child), None) == ERR_RANGE
    assert lib.bnpk_kmers_partition(dev.ctx, ptr(keys), ptr(keys), too_many, 31, 0, 52, 10, ptr(out), ptr(child), None) == ERR_RANGE
    assert torch.count_nonzero(out).item() == 0                      # untouched
    state = torch.zeros(64, dtype=torch.int64, device="cuda")
    nu, ov = C.c_int64(0), C.c_int(0)
    assert lib.bnpk_finish_sorted(dev.ctx, ptr(keys), 16, ptr(child), 1 << 31, 42, ptr(out), ptr(out), ptr(state), None, 0,
                                  None, None, C.byref(nu), C.byref(ov), None) in (ERR_RANGE, ERR_ARG)


def test_bad_arguments(env):
    lib, dev, ptr, torch = env
    keys = torch.zeros(16, dtype=torch.int64, device="cuda")
    out = torch.zeros(16, dtype=torch.int64, device="cuda")
    child = torch.zeros(4096, dtype=torch.int64, device="cuda")
    assert lib.bnpk_radix_partition(dev.ctx, ptr(keys), 16, None, 1, 52, 12, ptr(out), ptr(child), None) == ERR_ARG   # > 11 bits
    assert lib.bnpk_radix_partition(dev.ctx, ptr(keys), 16, None, 1, 60, 10, ptr(out), ptr(child), None) == ERR_ARG   # beyond bit 63
    assert lib.bnpk_radix_partition(dev.ctx, ptr(keys), 16, None, 1, 52, 10, ptr(keys), ptr(child), None) == ERR_ARG  # in place
    assert lib.bnpk_radix_partition(None, ptr(keys), 16, None, 1, 52, 10, ptr(out), ptr(child), None) == ERR_ARG
    assert lib.bnpk_kmers_generic(dev.ctx, ptr(keys), ptr(keys), ptr(keys), 1, 1, 32, 4, ptr(out), None) == ERR_ARG   # k > 31
    assert lib.bnpk_set_option(dev.ctx, b"no such knob", 1) == ERR_ARG
    assert lib.bnpk_set_option(dev.ctx, b"finish_mode", 7) == ERR_ARG
    assert lib.bnpk_set_option(dev.ctx, b"fastq_encoder", 2) == ERR_ARG
    text = torch.zeros(64, dtype=torch.uint8, device="cuda")
    cell = torch.zeros(1, dtype=torch.int64, device="cuda")
    lut = np.zeros(256, dtype=np.uint8)
    assert lib.bnpk_lut_bytes(dev.ctx, C.c_void_p(text.data_ptr() + 1), 16, lut.ctypes.data_as(C.c_void_p),
                              C.c_void_p(text.data_ptr() + 1), ptr(cell), None) == ERR_ALIGN
    assert lib.bnpk_strerror(ERR_RANGE).decode() == "size out of supported range"


def test_out_of_memory_is_a_status(env):
    lib, dev, ptr, torch = env
    p = C.c_void_p()
    assert lib.bnpk_host_alloc(1 << 50, C.byref(p)) == ERR_NOMEM        # a petabyte of pinned memory


def test_finish_reports_a_bucket_over_capacity_that_nobody_counted(env):
    """*h_overflow: a bucket larger than bnpk_finish_capacity() without a pre-counted entry — the outputs are to be discarded"""
    lib, dev, ptr, torch = env
    cap = int(lib.bnpk_finish_capacity())
    n = cap + 5
    keys = torch.arange(n, dtype=torch.int64, device="cuda")
    off = torch.tensor([0, n], dtype=torch.int64, device="cuda")
    out_k, out_c = torch.empty(n, dtype=torch.int64, device="cuda"), torch.empty(n, dtype=torch.int64, device="cuda")
    state = torch.empty(lib.bnpk_finish_state_words(1), dtype=torch.int64, device="cuda")
    for mode in (0, 1, 2, 3, 4, 6):
        assert lib.bnpk_set_option(dev.ctx, b"finish_mode", mode) == OK
        nu, ov = C.c_int64(0), C.c_int(0)
        assert lib.bnpk_finish_sorted(dev.ctx, ptr(keys), n, ptr(off), 1, 40, ptr(out_k), ptr(out_c), ptr(state), None, 0, None,
                                      None, C.byref(nu), C.byref(ov), None) == OK
        assert ov.value == 1, mode
    assert lib.bnpk_set_option(dev.ctx, b"finish_mode", 0) == OK
