"""Where do the cycles of conv3x3_halo_f16x3_kernel go?  Runs the benchmark's narrow-layer shapes through a library
built with -DPP_HALO_TRACE (pacingpseudo_amd/lib/trace/, see HT_TRK in csrc/pp_conv.hip): wave 0 of block (0, 0) adds
up the shader-clock cycles of each phase of a stage.

    make trace && PP_LIB_PATH=$PWD/pacingpseudo_amd/lib/trace/libpacingpseudo_hip.so python tests/studies/halo_phase_trace.py
phases: 0 first barrier (waiting for the other waves / the previous stage), 1 patch -> LDS (incl. the wait for its
loads), 2 second barrier, 3 pending-tile stores + accumulator reset, 4 address arithmetic + issue of the prefetch,
5 the 54 MFMAs with their fragment reads, 6 tile finalisation."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pacingpseudo_amd import _lib  # noqa: E402

lib = _lib.lib
dll = lib.load()
dll.pp_debug_halo_trace.restype = C.c_int
dll.pp_debug_halo_trace.argtypes = [C.c_void_p, C.c_int]
NAMES1 = ['barrier1', 'patch_to_lds', 'barrier2', 'pending_stores', 'prefetch_issue', 'mfma_loop', 'finalize', '-', '-', '-']
# two-half kernel: P phase 0-4, M phase 5-9 (5 = its first barrier, 6 = steps 0-7, 7 = its second barrier, 8 = steps 8-17)
NAMES2 = ['P.barrier1', 'P.patch_to_lds', 'P.barrier2', 'P.pending_stores', 'P.prefetch_issue', 'M.barrier1', 'M.steps0_7',
          'M.barrier2', 'M.steps8_17', 'M.finalize']
dev = 'cuda'
st = torch.cuda.current_stream().cuda_stream
out_rows = []
for (Cin, Cout, H, B, acc) in [(32, 32, 256, 64, 0), (64, 64, 128, 64, 0), (96, 32, 256, 64, 0), (64, 192, 128, 64, 0),
                               (64, 192, 128, 64, 1), (32, 96, 256, 64, 1)]:
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    wf = torch.zeros(Cout, 9, Cin, device=dev)
    lib.pp_pack_conv3x3_weights_f16x3(w.data_ptr(), Cout, Cin, Cin, wf.data_ptr(), None, st)
    y = torch.zeros(B, H, H, Cout, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(3):
        if it == 2:
            ev[0].record()
        lib.pp_conv3x3_fwd_f16x3(x.data_ptr(), Cin, Cin, wf.data_ptr(), None, y.data_ptr(), Cout, Cout, B, H, H, 1, acc, None, st)
    ev[1].record()
    torch.cuda.synchronize()
    buf = (C.c_longlong * 16)()
    assert dll.pp_debug_halo_trace(buf, 16) == 0
    tr = list(buf)
    stages = max(tr[10], 1)
    names = NAMES2 if tr[13] == 2 else NAMES1
    mhz = tr[11] / (tr[12] / 100.0) if tr[12] else 0.0          # shader cycles per microsecond
    row = dict(shape=f'{Cin}->{Cout} @{H}^2 x{B} acc={acc}', kernel='two-half' if tr[13] == 2 else 'one-half',
               launch_us=round(ev[0].elapsed_time(ev[1]) * 1e3, 1),
               block0_us=round(tr[12] / 100.0, 1), shader_clock_mhz=round(mhz), stages_or_rounds=stages,
               cycles_per_stage={n: round(tr[i] / stages) for i, n in enumerate(names) if n != '-'},
               total_cycles_per_stage=round(tr[11] / stages))
    out_rows.append(row)
    print(json.dumps(row))
json.dump(out_rows, open('gpurun_out/halo_phase_trace.json', 'w'), indent=1)
