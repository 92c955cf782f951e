import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pacingpseudo_amd._lib import lib, stream_ptr
st = stream_ptr()
for blocks in (256, 512, 1024):
    out = torch.empty(blocks * 256, device='cuda')
    fl = ctypes.c_double()
    lib.pp_mfma_probe(out.data_ptr(), blocks, 2000, ctypes.byref(fl), st); torch.cuda.synchronize()
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.pp_mfma_probe(out.data_ptr(), blocks, 20000, ctypes.byref(fl), st); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(f'blocks {blocks}: {fl.value / ms / 1e9:.1f} TFLOP/s ({ms:.2f} ms)')
