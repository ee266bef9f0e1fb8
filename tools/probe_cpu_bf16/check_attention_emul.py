import ctypes, numpy as np, torch, time, sys
lib = ctypes.CDLL("./libattn.so")
fp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
def bits(t): return t.contiguous().view(torch.int16).numpy().view(np.uint16)
torch.manual_seed(14)
B, T, C = int(sys.argv[1]) if len(sys.argv)>1 else 2, 1024, 512
q = (torch.randn(B,1,T,C)*1.5).bfloat16(); k = (torch.randn(B,1,T,C)*1.5).bfloat16(); v = torch.randn(B,1,T,C).bfloat16()
ref = torch.nn.functional.scaled_dot_product_attention(q,k,v)
rb = bits(ref)
for lanes in (16,):
    for pv in (0, 2):
        o = np.zeros(rb.shape, dtype=np.uint16); t0=time.time()
        lib.attn_emul(fp(bits(q)), fp(bits(k)), fp(bits(v)), fp(o), B, T, C, lanes, pv, 256, 512)
        print(f"lanes={lanes} pv_mode={pv}: mismatches {(o != rb).sum()} / {o.size}  ({time.time()-t0:.1f}s)", flush=True)
