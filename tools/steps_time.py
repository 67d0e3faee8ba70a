"""timing only (no verification): K asynchronous 2^log2n-point MSMs in overlap mode, three repetitions.
    python tools/steps_time.py [log2n] [K] [bracket the accumulation with events: 0|1]
Used with H2AGG_DBG_SKIP=1|3|7 (skip bucket reduction / + window sums / + Horner tail: WRONG results, timing only; needs a
library built with -DH2AGG_MEASURE_KNOBS) to price the tails."""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package()
_early = None
_tearly = None
if os.environ.get('TORCH_STREAM_EARLY'):   # a torch-made stream that exists before the context's own streams do
    _tearly = torch.cuda.Stream(torch.device('cuda', 0))
if os.environ.get('HIP_STREAM_EARLY'):   # the caller's stream made BEFORE the context (and its tail streams) exists
    import ctypes
    torch.zeros(1, device='cuda')
    _hip0 = ctypes.CDLL("libamdhip64.so")
    _early = ctypes.c_void_p()
    assert _hip0.hipStreamCreateWithFlags(ctypes.byref(_early), 1) == 0
eng = pkg.H2Agg(0)
if _early is not None:
    eng.set_stream(_early.value)
if _tearly is not None:
    torch.cuda.set_stream(_tearly); eng.set_stream(_tearly.cuda_stream)
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
PROF = int(sys.argv[3]) if len(sys.argv) > 3 else 0
from bench import gen_scalars
_, k = gen_scalars(1, n)
_, s = gen_scalars(2, n)
k = k.copy(); s = s.copy()
dev = torch.device('cuda', 0)
dk = torch.from_numpy(k).to(dev); ds = torch.from_numpy(s).to(dev)
out = torch.zeros(96 * K, dtype=torch.uint8, device=dev)
t = eng.bases_generate(dk.data_ptr(), n)
import os
if os.environ.get('HIP_STREAM'):     # a stream made with the HIP API directly: "blocking" (hipStreamCreate) or "nonblocking"
    import ctypes
    _hip = ctypes.CDLL("libamdhip64.so")
    _h = ctypes.c_void_p()
    rc = _hip.hipStreamCreateWithFlags(ctypes.byref(_h), 1 if os.environ['HIP_STREAM'] == 'nonblocking' else 0)
    assert rc == 0 and _h.value
    eng.set_stream(_h.value)
if os.environ.get('TORCH_STREAM'):   # the caller's own stream as the context's main stream (a torch-made one: torch keeps a pool of them)
    _st = torch.cuda.Stream(dev); torch.cuda.set_stream(_st); eng.set_stream(_st.cuda_stream)
eng.msm_set_tail_overlap(int(os.environ.get('LEVEL', '2')))
if os.environ.get('WINDOW'):
    eng.msm_configure(int(os.environ['WINDOW']), 0, 0)
if PROF:
    eng.profile_enable(True, only_stage=4)
for rep in range(3):
    for i in range(5):
        eng.g1_msm_device_async(t, ds.data_ptr(), n, out.data_ptr() + 96 * i)
    eng.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        eng.g1_msm_device_async(t, ds.data_ptr(), n, out.data_ptr() + 96 * i)
    eng.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("ms/step %.3f  %.1f Mpts/s" % (dt * 1e3, n / dt / 1e6), flush=True)
