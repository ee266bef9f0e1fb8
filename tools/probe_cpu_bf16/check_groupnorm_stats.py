import ctypes, numpy as np, torch
lib = ctypes.CDLL("./libgn.so")
fp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
def bits(t): return t.contiguous().view(torch.int16).numpy().view(np.uint16)
torch.manual_seed(8)
for (B,C,HW) in [(64,32,512),(64,32,1024),(64,32,8192),(8,128,65536)]:
    x = (torch.randn(B,C,HW,1)*1.7+0.3).bfloat16(); g = torch.ones(C).bfloat16(); b = torch.zeros(C).bfloat16()
    out, mean, rstd = torch.native_group_norm(x, g.float(), b.float(), B, C, HW, 32, 1e-6)
    m = mean.reshape(-1).numpy(); r = rstd.reshape(-1).numpy()
    xb, gb, bb = bits(x), bits(g), bits(b)
    for var in (6,):
        lib.set_var(var)
        o = np.zeros(xb.shape, dtype=np.uint16); stats = np.zeros((B*32,2), dtype=np.float32)
        lib.group_norm_bf16(fp(xb), fp(gb), fp(bb), fp(o), B, C, ctypes.c_int64(HW), 32, ctypes.c_double(1e-6), 0, fp(stats))
        print(f"group elems {C//32*HW}: VAR={var} mean equal {(m == stats[:,0]).mean():.3f} rstd equal {(r == stats[:,1]).mean():.3f}")
