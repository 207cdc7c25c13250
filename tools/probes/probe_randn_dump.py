"""GPU box: dump torch.randn(2^20) and the per-thread raw material of its Box-Muller (probe_randn_dump.hip) to gpurun_out/."""
import ctypes, os, subprocess, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "probe_randn_dump.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "probe_randn_dump.hip")])
lib = ctypes.CDLL(so)
lib.probe_dump.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int64, ctypes.c_void_p]
dev = torch.device("cuda:0"); torch.zeros(1, device=dev)
n = 1 << 20
torch.manual_seed(4242)
g = torch.cuda.default_generators[0]
seed, off = g.initial_seed(), g.get_offset()
want = torch.randn(n, device=dev)
G = 256 * min(2048, n // 256)
dump = torch.empty(G, 8, device=dev)
lib.probe_dump(ctypes.c_void_p(dump.data_ptr()), seed, off, G, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
out = sys.argv[1]
sel = slice(0, 40000)
np.savez_compressed(out, dump=dump[sel].cpu().numpy(), want0=want[:G][sel].cpu().numpy(), want1=want[G:2 * G][sel].cpu().numpy(), G=G, n=n)
print("saved", out, G)
