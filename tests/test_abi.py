"""The C-ABI library loads without a GPU and exports exactly what include/pacingpseudo_hip.h and (16-bit storage modes)
include/pacingpseudo_hip_h16.h / include/pacingpseudo_hip_bf16.h declare."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'pacingpseudo_hip.h')
HEADER_H16 = os.path.join(ROOT, 'include', 'pacingpseudo_hip_h16.h')
HEADER_BF16 = os.path.join(ROOT, 'include', 'pacingpseudo_hip_bf16.h')


def declared():
    """{name: number of parameters} parsed from the three headers."""
    txt = open(HEADER).read() + open(HEADER_H16).read() + open(HEADER_BF16).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    out = {}
    for m in re.finditer(r'\b(?:int|size_t|const char\*)\s+(pp_\w+)\s*\(([^;]*?)\)\s*;', txt, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ('', 'void') else args.count(',') + 1
    return out


def test_header_and_binding_agree():
    from pacingpseudo_amd import _lib
    d = declared()
    assert len(d) >= 30
    assert set(d) == set(_lib.EXPORTED_SYMBOLS), set(d) ^ set(_lib.EXPORTED_SYMBOLS)
    for name, n in d.items():
        assert len(_lib._PROTOS[name][1]) == n, f'{name}: header has {n} parameters, ctypes binding {len(_lib._PROTOS[name][1])}'


def test_h16_header_is_generated_from_the_sources():
    """include/pacingpseudo_hip_h16.h is the text scripts/gen_h16_header.py derives from the PP_FN definitions, and the ctypes
    table routes exactly those entry points to their _h16 twins."""
    import subprocess
    import sys
    from pacingpseudo_amd import _lib
    assert subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'gen_h16_header.py'), '--check']).returncode == 0, \
        'run python scripts/gen_h16_header.py'
    txt = re.sub(r'/\*.*?\*/', '', open(HEADER_H16).read(), flags=re.S)
    names = set(re.findall(r'\b(pp_\w+_h16)\s*\(', txt))
    assert names == {n + '_h16' for n in _lib.H16_ENTRIES}
    assert 'pp_h16_t' in txt and 'float* in,' not in txt and 'float* dz,' not in txt     # activation pointers are fp16
    # round 6: the same entry points a third time with bfloat16 tensors
    txt = re.sub(r'/\*.*?\*/', '', open(HEADER_BF16).read(), flags=re.S)
    assert set(re.findall(r'\b(pp_\w+_bf16)\s*\(', txt)) == {n + '_bf16' for n in _lib.H16_ENTRIES}
    assert 'pp_bf16_t' in txt and 'float* in,' not in txt and 'float* dz,' not in txt
    assert _lib.lib_for('bf16') is _lib.lib_bf16 and _lib.lib_for('fp16') is _lib.lib_h16 and _lib.lib_for(4) is _lib.lib


def test_library_exports_every_symbol():
    from pacingpseudo_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('library not built (run __graft_entry__.build())')
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared():
        assert hasattr(dll, name), name
    assert _lib.lib.pp_version() >= 600
    assert _lib.lib.pp_last_error() is not None


def test_every_entry_cites_the_reference():
    txt = open(HEADER).read()
    for needle in ('models/unet.py', 'aux_path_memory.py', 'losses/losses.py', 'train_chaos.py',
                   'consistency_reglur_memory.py', 'utils/metrics.py'):
        assert needle in txt, needle


def test_missing_library_fails_loudly(monkeypatch):
    from pacingpseudo_amd import _lib
    fresh = _lib._Lib()
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libpacingpseudo_hip.so')
    with pytest.raises(_lib.HipLibraryError):
        fresh.load()


def test_stale_library_fails_loudly(monkeypatch, tmp_path):
    """A library built from an older tree (pp_version below what this host side binds, or no pp_version at all) is refused at
    load time instead of being called with this round's argument lists."""
    import shutil
    import subprocess
    from pacingpseudo_amd import _lib
    if shutil.which('gcc') is None:
        pytest.skip('gcc is not installed')
    for body in ('int pp_version(void) { return %d; }' % (_lib.MIN_LIB_VERSION - 1), 'int pp_other(void) { return 0; }'):
        src = tmp_path / 'stale.c'
        src.write_text(body)
        so = tmp_path / ('stale%d.so' % len(body))
        subprocess.run(['gcc', '-shared', '-fPIC', '-o', str(so), str(src)], check=True)
        fresh = _lib._Lib()
        monkeypatch.setattr(_lib, 'LIB_PATH', str(so))
        with pytest.raises(_lib.HipLibraryError):
            fresh.load()


def test_model_refuses_cpu():
    """No CPU fallback: the product path raises instead of computing on the host."""
    import torch
    from oracle import pacing_oracle as O
    from tests.test_gpu_step import build_model  # noqa: F401  (import only; it calls .cuda())
    from pacingpseudo_amd.models import ConsistencyRegulr
    args = O.default_args(init_ch=4, max_ch=32, hid_ch=8, feat_ch=[32, 32])
    m = ConsistencyRegulr(
        kwargs_unet=dict(input_ch=1, init_ch=4, max_ch=32, num_classes=5, output_stride=8, is_stride_conv=False,
                         is_trans_conv=False, elab_end_points=True),
        kwargs_aux_path=dict(num_classes=5, feat_stage=args.feat_stage, feat_ch=args.feat_ch, hid_ch=8,
                             aux_drop_prob=0.0, do_memory=False, max_step=400, update_momentum=0.9,
                             ensemble_mode='cosine_similarity'),
        args_parser=args)
    b = O.synthetic_batch(1, 32, 32)
    with pytest.raises(RuntimeError):
        m(b, mode='val')
    with pytest.raises(RuntimeError):
        m.backbone.enc_block1.conv_block.conv_layer1(torch.zeros(1, 1, 8, 8))


def test_host_side_under_address_sanitizer():
    """`make asan` instruments the host half of every source (pp_runtime.cpp, the launch wrappers' argument
    checks) and runs tests/native/asan_args.cpp -- invalid arguments for a representative entry point of every source file.
    No GPU needed: every call must be rejected before anything is launched, without a sanitizer report."""
    import shutil
    import subprocess
    if shutil.which('hipcc') is None or not os.path.exists('/opt/rocm/lib/llvm/bin/clang++'):
        pytest.skip('hipcc / the ROCm clang++ are not installed: the sanitizer build cannot be made here')
    r = subprocess.run(['make', '-C', ROOT, '-j', '6', 'asan'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'all argument checks rejected their input' in r.stdout and 'AddressSanitizer' not in r.stdout + r.stderr
