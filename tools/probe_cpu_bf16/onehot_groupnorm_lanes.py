import numpy as np, torch, ctypes
lib = ctypes.CDLL("./libgn.so")
fp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
def bits(t): return t.contiguous().view(torch.int16).numpy().view(np.uint16)
HW=32; C=32; B=1
x = torch.eye(32).reshape(1,32,32,1).bfloat16()*3.0   # channel c: one-hot at position c
out, mean, rstd = torch.native_group_norm(x, torch.ones(C), torch.zeros(C), B, C, HW, 32, 1e-6)
m = mean.reshape(-1).numpy()
np.set_printoptions(precision=10, linewidth=200)
print("torch mean*32/3 - 1 (in units of 2^-24):", ((m.astype(np.float64)*32/3 - 1)*2**24).round(2))
xb = bits(x); o = np.zeros(xb.shape, dtype=np.uint16); stats = np.zeros((B*32,2), dtype=np.float32)
for var in (0,7):
    lib.set_var(var)
    lib.group_norm_bf16(fp(xb), fp(bits(torch.ones(C).bfloat16())), fp(bits(torch.zeros(C).bfloat16())), fp(o), B, C, ctypes.c_int64(HW), 32, ctypes.c_double(1e-6), 0, fp(stats))
    print("emul var",var, ((stats[:,0].astype(np.float64)*32/3 - 1)*2**24).round(2))
