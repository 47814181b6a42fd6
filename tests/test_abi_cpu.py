"""CPU tests of the C-ABI library and of the kernel arithmetic simulated on the host.

No compute entry point is called here (no GPU in this container): the library must load,
export every symbol include/mla_hip.h declares, and its host-side helpers (frame counts,
constant tables) must agree with the oracle. The per-lane kernel math (csrc/logmel_core.h)
is compiled with g++ and run lane by lane against the oracle.
"""

import ctypes
import importlib
import os
import subprocess

import numpy as np
import pytest

from conftest import PKG, ROOT
from oracle import frontend as ofe


@pytest.fixture(scope="module")
def L():
    build = importlib.import_module(PKG + ".build")
    build.build(verbose=False)
    return importlib.import_module(PKG + "._lib")


def test_library_exports_every_declared_symbol(L):
    lib = L.lib()
    names = L.declared_symbols()
    assert "mla_logmel_examples" in names and len(names) >= 7
    for n in names:
        assert hasattr(lib, n), n
    assert lib.mla_abi_version() >= 1


def test_counts_match_reference_table(L, golden):
    lib = L.lib()
    f, e = ctypes.c_int64(), ctypes.c_int64()
    for n, n_ex, raised in golden("frontend")["count_table"]:
        rc = lib.mla_logmel_counts(int(n), ctypes.byref(f), ctypes.byref(e))
        if raised:
            assert rc == L.E_SHORT and b"ValueError" in lib.mla_last_error()
        else:
            assert rc == 0 and e.value == n_ex and f.value == ofe.num_frames(int(n), 400, 160)
    for n in range(0, 40000, 37):
        rc = lib.mla_logmel_counts(n, ctypes.byref(f), ctypes.byref(e))
        if n < 240:
            assert rc == L.E_SHORT
        else:
            assert rc == 0 and (f.value, e.value) == (ofe.num_frames(n, 400, 160), ofe.num_examples(n))


def test_tables_match_oracle(L):
    lib = L.lib()
    win, mel = np.zeros(400), np.zeros((257, 64))
    assert lib.mla_logmel_reference_tables(win.ctypes.data_as(ctypes.c_void_p), mel.ctypes.data_as(ctypes.c_void_p)) == 0
    assert np.array_equal(win, ofe.periodic_hann(400))
    ref = ofe.mel_matrix(64, 257, 16000, 125.0, 7500.0)
    np.testing.assert_allclose(mel, ref, rtol=0, atol=1e-15)
    assert np.array_equal(mel != 0, ref != 0)
    n = lib.mla_logmel_table_floats()
    tab = np.zeros(n, dtype=np.float32)
    assert lib.mla_logmel_build_tables(tab.ctypes.data_as(ctypes.c_void_p)) == 0
    assert np.array_equal(tab[:400], win.astype(np.float32)) and not tab[400:512].any()
    m = np.arange(256)
    np.testing.assert_allclose(tab[512:1024:2], np.cos(2 * np.pi * m / 256), atol=1e-7)
    np.testing.assert_allclose(tab[513:1024:2], -np.sin(2 * np.pi * m / 256), atol=1e-7)
    # sparse per-lane mel rows reproduce the dense matrix (weights are stored halved)
    starts = tab[1536:1600].view(np.int32).reshape(16, 4)
    rows = tab[1600:1600 + 16 * 52].reshape(16, 52)
    dense = np.zeros((257, 64))
    counts = (8, 8, 12, 20)
    for lane in range(16):
        bands = (lane, 31 - lane, 32 + lane, 63 - lane)
        first = 0
        for s in range(4):
            assert 4 <= starts[lane, s] and starts[lane, s] % 4 == 0 and starts[lane, s] + counts[s] <= 256
            dense[starts[lane, s]:starts[lane, s] + counts[s], bands[s]] = 2.0 * rows[lane, first:first + counts[s]]
            first += counts[s]
    np.testing.assert_allclose(dense, ref, rtol=1e-7, atol=0)


@pytest.fixture(scope="module")
def hostsim(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("hostsim") / "hostsim.so")
    src = os.path.join(ROOT, PKG, "csrc", "logmel_hostsim.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src], check=True)
    lib = ctypes.CDLL(so)
    lib.hostsim_examples.restype = ctypes.c_int64
    lib.hostsim_examples.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    return lib


def mel_domain_close(got, ref, rel=2e-5, floor=2e-6):
    """|mel - mel_ref| <= rel*mel_ref + floor*max(mel_ref of the frame): float32 dynamic range
    (about -115 dB below the frame's strongest band), see DESIGN.md "front-end tolerance"."""
    g, r = np.exp(got.astype(np.float64)) - 0.01, np.exp(ref) - 0.01
    bound = rel * r + floor * r.max(axis=-1, keepdims=True) + 1e-9
    return bool(np.all(np.abs(g - r) <= bound)), float(np.max(np.abs(g - r) / bound))


def test_kernel_math_on_host_matches_oracle(hostsim, mk):
    for name, wav in mk.test_waveforms().items():
        if wav.ndim > 1:
            wav = wav.mean(axis=1)
        x = np.ascontiguousarray(wav.astype(np.float32))
        ref = ofe.waveform_to_examples(wav)
        out = np.zeros(ref.shape, dtype=np.float32)
        n = hostsim.hostsim_examples(x.ctypes.data_as(ctypes.c_void_p), len(x), out.ctypes.data_as(ctypes.c_void_p))
        assert n == ref.shape[0], name
        ok, worst = mel_domain_close(out, ref)
        assert ok, (name, worst)
        if name.startswith(("noise", "stereo", "silence")):       # broadband: log-domain bound
            assert np.abs(out - ref).max() <= 1e-4, name


def test_argument_errors_are_reported_before_any_launch(L):
    """Error behaviour of the C ABI (include/mla_hip.h): negative MLA_E_* codes + a message in mla_last_error(), decided on
    the host before any HIP call -- so it is checkable without a GPU. Zero-sized work returns MLA_OK without touching
    its (null) buffers, like the reference's functions on empty inputs."""
    lib = L.lib()
    lib.mla_last_error.restype = ctypes.c_char_p
    vp = ctypes.c_void_p
    fake = vp(0x1000)                                # aligned, never dereferenced: every call below fails validation first
    E_ARG, E_SHAPE, E_SHORT, E_DTYPE = -1, -2, -3, -5

    def expect(code, rc, needle=None):
        assert rc == code, (rc, lib.mla_last_error())
        if needle:
            assert needle in lib.mla_last_error().decode(), lib.mla_last_error()

    # front-end: reference raises ValueError for n < 240 (negative frame count)
    expect(E_SHORT, lib.mla_logmel_examples(fake, L.F32, 1, 239, 239, fake, fake, L.F32, None), "239")
    expect(E_DTYPE, lib.mla_logmel_examples(fake, 7, 1, 16000, 16000, fake, fake, L.F32, None))
    expect(E_DTYPE, lib.mla_logmel_examples(fake, L.F32, 1, 16000, 16000, fake, fake, L.I16, None))
    expect(E_ARG, lib.mla_logmel_examples(None, L.F32, 1, 16000, 16000, fake, fake, L.F32, None), "null")
    expect(E_ARG, lib.mla_logmel_examples(fake, L.F32, 1, 16000, 15999, fake, fake, L.F32, None))      # stride < n_samples
    expect(E_ARG, lib.mla_logmel_examples(fake, L.F32, 1, 16000, 16000, fake, vp(0x1004), L.F32, None))  # out misaligned
    assert lib.mla_logmel_examples(None, L.F32, 0, 16000, 16000, None, None, L.F32, None) == 0          # no waveforms
    assert lib.mla_logmel_examples(None, L.F32, 4, 15599, 15599, None, None, L.F32, None) == 0          # 0 examples each
    expect(E_ARG, lib.mla_mono_mix(fake, L.F32, 100, 0, fake, None))
    expect(E_DTYPE, lib.mla_mono_mix(fake, L.BF16, 100, 2, fake, None))
    assert lib.mla_mono_mix(None, L.F32, 0, 2, None, None) == 0
    # GEMM / conv
    expect(E_ARG, lib.mla_linear(fake, 64, fake, 64, None, fake, 64, -1, 64, 64, L.F32, L.F32, 0, None))
    expect(E_SHAPE, lib.mla_linear(fake, 66, fake, 66, None, fake, 64, 8, 64, 66, L.F32, L.F32, 0, None), "16-byte rows")
    expect(E_DTYPE, lib.mla_linear(fake, 64, fake, 64, None, fake, 64, 8, 64, 64, L.F32, L.BF16, 0, None))
    expect(E_ARG, lib.mla_linear(vp(0x1008), 64, fake, 64, None, fake, 64, 8, 64, 64, L.F32, L.F32, 0, None), "aligned")
    assert lib.mla_linear(None, 64, None, 64, None, None, 64, 0, 64, 64, L.F32, L.F32, 0, None) == 0
    expect(E_DTYPE, lib.mla_vggish_conv(2, fake, fake, fake, fake, 4, 9, None))
    expect(E_ARG, lib.mla_vggish_conv(2, None, fake, fake, fake, 4, L.BF16, None))
    assert lib.mla_vggish_conv(2, None, None, None, None, 0, L.BF16, None) == 0
    rc = lib.mla_vggish_conv(9, fake, fake, fake, fake, 4, L.BF16, None)
    assert rc in (E_ARG, E_SHAPE) and b"layer" in lib.mla_last_error()


def _build_c_example(L, tmp_path):
    """gcc -std=c99 on examples/c_abi_logmel.c against include/mla_hip.h + libmla_hip.so: the header is plain C."""
    L.lib()
    exe = str(tmp_path / "c_abi_logmel")
    pkg = os.path.join(ROOT, PKG)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "c_abi_logmel.c"), "-L" + pkg, "-lmla_hip", "-L/opt/rocm/lib", "-lamdhip64",
                    "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    return exe


def test_c_host_compiles_against_the_header(L, tmp_path):
    assert os.path.exists(_build_c_example(L, tmp_path))


@pytest.mark.gpu
def test_c_host_runs(L, tmp_path):
    """The plain-C host (no Python, no torch in the process) produces examples and sees the reference's short-input error."""
    out = subprocess.run([_build_c_example(L, tmp_path)], check=True, capture_output=True, text=True).stdout
    assert "examples (40, 96, 64)" in out and "rc -3" in out, out
