import ctypes, numpy as np, torch, sys
lib = ctypes.CDLL("./libgn.so")
fp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
def bits(t): return t.contiguous().view(torch.int16).numpy().view(np.uint16)
torch.manual_seed(8)
for (B,C,H) in [(1,128,256),(2,128,256),(1,256,128),(1,512,64),(2,512,32),(1,128,128),(1,256,64)]:
    x = (torch.randn(B,C,H,H)*1.7+0.3).bfloat16(); g = (torch.rand(C)*0.2+0.9).bfloat16(); b = (torch.randn(C)*0.1).bfloat16()
    y = torch.nn.functional.group_norm(x, 32, g, b, eps=1e-6)
    xb, gb, bb = bits(x), bits(g), bits(b)
    lib.set_var(6)
    for variant in range(8):
        out = np.zeros(xb.shape, dtype=np.uint16)
        lib.group_norm_bf16(fp(xb), fp(gb), fp(bb), fp(out), B, C, ctypes.c_int64(H*H), 32, ctypes.c_double(1e-6), variant, None)
        mism = int((out != bits(y)).sum())
        print(f"B={B} C={C} H={H} variant={variant}: mismatches {mism} / {out.size}", flush=True)
