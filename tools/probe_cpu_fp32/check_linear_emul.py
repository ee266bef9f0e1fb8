"""mklgemm_emul.c vs torch-CPU F.linear on random data at the Q-Former encoder's Linear shapes (fp32)"""
import ctypes, sys, numpy as np, torch, torch.nn.functional as F, os
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmklgemm.so"))
def model(x, w, b, variant, tail):
    M, K = x.shape; N = w.shape[0]
    out = np.empty((M, N), np.float32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.linear_model(p(x), p(w), p(b) if b is not None else None, p(out), M, N, K, variant, tail)
    return out
torch.manual_seed(0)
shapes = [(64, 192), (64, 1024), (512, 1536), (64, 64), (64, 256), (256, 64), (512, 512), (512, 2048), (2048, 512), (512, 16), (256, 512), (1024, 512), (1536, 4608), (6144, 1536)]
Ms = [int(a) for a in sys.argv[1:]] or [512, 4096]
for K, N in shapes:
    for M in Ms:
        x = torch.randn(M, K); w = torch.randn(N, K) / K ** 0.5; b = torch.randn(N)
        y = F.linear(x, w, b).numpy()
        res = []
        for variant in (0, 1, 2):
            for tail in (0, 1):
                o = model(x.numpy(), w.numpy(), b.numpy(), variant, tail)
                res.append((variant, tail, int((o.view(np.uint32) != y.view(np.uint32)).sum())))
        print(f"K={K} N={N} M={M}: mismatching outputs of {M*N} per (variant, tail rule): {res}", flush=True)
