"""Run on the GPU box: which Box-Muller arithmetic reproduces torch.randn bit for bit (tools/probes/probe_randn.hip)?"""
import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "probe_randn.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "probe_randn.hip")])
lib = ctypes.CDLL(so)
lib.probe_randn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int64, ctypes.c_void_p]
dev = torch.device("cuda:0")
torch.cuda.init()
torch.zeros(1, device=dev)
n = 1024 * 64
os.path.exists(so) and os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "probe_randn.hip")) and os.remove(so)
torch.manual_seed(4242)
g = torch.cuda.default_generators[0]
seed, off = g.initial_seed(), g.get_offset()
want = torch.randn(n, device=dev)
out = torch.empty(n, device=dev)
names = {"VF": ["mul+add", "fma", "(y+1)*c", "y*c"], "LG": ["logf", "__logf", "__log2f*ln2"], "SQ": ["sqrtf", "__fsqrt_rn", "amdgcn_sqrt"], "SC": ["__sincosf", "sincosf"]}
for v in [18, 30, 36, 48, 52, 40, 54, 66, 70]:
    vf, lg, sq, sc = v // 18, (v // 6) % 3, (v // 2) % 3, v % 2
    rc = lib.probe_randn(v, ctypes.c_void_p(out.data_ptr()), seed, off, n, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    bad = int((out != want).sum())
    print("variant %2d v=%-8s log=%-12s sqrt=%-12s sincos=%-10s mismatches %6d  max|d| %.3e" % (
        v, names["VF"][vf], names["LG"][lg], names["SQ"][sq], names["SC"][sc], bad, float((out - want).abs().max())), flush=True)
# ---- the first mismatches of variant 30 with every intermediate, next to float64 references
import numpy as np
lib.probe_dump.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int64, ctypes.c_void_p]
lib.probe_randn(30, ctypes.c_void_p(out.data_ptr()), seed, off, n, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
dump = torch.empty(n, 8, device=dev)
lib.probe_dump(ctypes.c_void_p(dump.data_ptr()), seed, off, n, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
bad = (out != want).nonzero().flatten()[:24].cpu()
D = dump.cpu().numpy(); W = want.cpu().numpy(); O = out.cpu().numpy()
f32 = np.float32
for i in bad.tolist():
    x, y = D[i, 0:2].view(np.uint32)
    u, v, l, m, s, sn = [f32(t) for t in D[i, 2:8]]
    lu = np.log(np.float64(u)); s64 = np.sqrt(-2 * lu); sin64 = np.sin(np.float64(v))
    # what s would have to be for want == fl(sn * s_t): the neighbours of s
    cands = [np.nextafter(s, f32(0)), s, np.nextafter(s, f32(9))]
    fit_s = [bool(f32(sn * c) == W[i]) for c in cands]
    sns = [np.nextafter(sn, f32(-9)), sn, np.nextafter(sn, f32(9))]
    fit_sn = [bool(f32(c * s) == W[i]) for c in sns]
    print("i %6d x %08x y %08x got %s want %s | s %s (exact %.9g, fl %s) m %s (exact -2ln u %.9g) | sn %s (exact %.9g) | want fits s-1,s,s+1: %s  sn-1,sn,sn+1: %s"
          % (i, x, y, O[i].view(np.uint32).item().__format__('08x'), W[i].view(np.uint32).item().__format__('08x'), s.view(np.uint32).item().__format__('08x'),
             s64, f32(s64).view(np.uint32).item().__format__('08x'), m.view(np.uint32).item().__format__('08x'), -2 * lu, sn.view(np.uint32).item().__format__('08x'), sin64, fit_s, fit_sn))
